cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06dec; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/tr -- python bench.py --workload fpn_attention --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace.log 2>&1
db=$(find $O/tr -name "*.db" | head -1)
python tools/rocpd_sequence.py "$db" > $O/fpn_seq.txt 2>&1
rm -rf $O/tr
grep -o '"ms_per_step": [0-9.]*' $O/trace.log | head -1
