"""Which torch (ATen) ops of one eager training step launch GPU work, and which line of megreader_amd issued them -- forward AND
backward (a TorchDispatchMode travels with the autograd engine's thread-local state, unlike the stack filter of
tools/trace_glue.py).  View / allocation ops are ignored.  usage: python tools/trace_ops.py --workload crnn|res50ppm|fpn_attention|db"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

IGNORE = {"empty", "empty_like", "empty_strided", "view", "_unsafe_view", "permute", "slice", "select", "as_strided", "expand",
          "detach", "alias", "t", "transpose", "unsqueeze", "squeeze", "reshape", "_reshape_alias", "set_", "unbind", "split",
          "narrow", "view_as", "is_same_size", "_local_scalar_dense", "lift_fresh", "new_empty", "new_empty_strided", "unfold",
          "result_type", "sym_size", "sym_stride", "sym_numel", "is_pinned", "record_stream", "resize_", "chunk",
          "split_with_sizes", "_to_copy_view"}


class Recorder(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        out = func(*args, **(kwargs or {}))
        if name not in IGNORE:
            frames = [f for f in traceback.extract_stack() if "megreader_amd" in f.filename and "_lib.py" not in f.filename]
            where = " < ".join("%s:%d %s" % (f.filename.split("megreader_amd/")[-1], f.lineno, f.name) for f in frames[-2:][::-1])
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            desc = "%s%s" % (str(t.dtype).replace("torch.", ""), list(t.shape)) if t is not None else ""
            self.counts[(name, where or "(outside megreader_amd)", desc)] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="crnn")
    ap.add_argument("--top", type=int, default=120)
    args = ap.parse_args()
    from trace_glue import build
    step = build(args.workload, torch.device("cuda", 0))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rec = Recorder()
    with rec:
        step()
    torch.cuda.synchronize()
    total = sum(rec.counts.values())
    print("%s: %d kernel-launching ATen calls in one eager step (views / allocations ignored)" % (args.workload, total))
    for (name, where, desc), n in rec.counts.most_common(args.top):
        print("%4d  %-22s %-34s %s" % (n, name, desc[:34], where[:150]))


if __name__ == "__main__":
    main()
