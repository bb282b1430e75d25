"""Launcher: run an unmodified reference entry point (train.py / eval.py) on the HIP kernels.

    cd <MegReader checkout> && python -m megreader_amd.run train.py experiments/recognition/crnn.yaml --batch_size 256
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
    dropin.install(root)
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
