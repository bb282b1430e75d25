"""Launcher: run an unmodified reference entry point (train.py / eval.py) on the HIP kernels.

    cd <MegReader checkout> && python -m megreader_amd.run train.py experiments/recognition/crnn.yaml --batch_size 256

The optimizer named by the YAML (torch.optim.Adam / SGD) resolves to the fused flat-buffer optimizers and `Trainer.train_step`
replays one captured hipGraph per step -- with `-d` the gradient all-reduce of the apex shim is captured inside it
(MEGREADER_DDP_GRAPH=auto|capture|graph2|off) -- MEGREADER_FAST=0 switches both off: the reference's own eager step on the HIP
modules.
"""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = sys.argv[1]
    root = os.path.dirname(os.path.abspath(script))
    from . import dropin
    fast = os.environ.get("MEGREADER_FAST", "1") != "0"
    dropin.install(root, fused_optimizer=fast, graph_step=fast)     # distributed (-d) too: dropin._GraphedTrainStep._capture
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
