#!/bin/bash
# the whole GPU suite, as the driver runs it
cd /root/repo
mkdir -p gpurun_out/r05f
( time timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/r05f/pytest_gpu_full_run.log 2>&1
tail -8 gpurun_out/r05f/pytest_gpu_full_run.log
