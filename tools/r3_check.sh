#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3chk; mkdir -p $O
timeout 600 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py tests/test_seg_detector_gpu.py tests/test_ddp_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
timeout 300 python bench.py --workload db --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_db.log 2>&1; tail -1 $O/bench_db.log | cut -c1-200
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 300 python bench.py --force-ddp --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/bench_ddp.log 2>&1; tail -1 $O/bench_ddp.log | cut -c1-330
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_ddp -- python bench.py --force-ddp --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 5 > $O/trace_ddp.log 2>&1
db=$(find $O/trace_ddp -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/ddp_kernel_stats.csv 2>&1; grep -i "nccl\|rccl\|TOTAL" $O/ddp_kernel_stats.csv | cut -c1-200; fi
rm -rf $O/trace_ddp
