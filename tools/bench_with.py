#!/usr/bin/env python
"""Run bench.py with library tuning setters applied first (in-step A/B of kernel options):
    python tools/bench_with.py mr_set_tn_taps_group=2 mr_set_tn_group=1 -- --no-cpu-baseline --no-secondary"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megreader_amd import _lib  # noqa: E402

args = sys.argv[1:]
sep = args.index("--") if "--" in args else len(args)
lib = _lib.load()
for kv in args[:sep]:
    k, v = kv.split("=")
    getattr(lib, k)(*[int(x) for x in v.split(",")])
sys.argv = [os.path.join(ROOT, "bench.py")] + args[sep + 1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
