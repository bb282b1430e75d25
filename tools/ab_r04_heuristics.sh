cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4z; mkdir -p $O
run() { # workload tuning
  MEGREADER_TUNING="$2" timeout 200 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --workload $1 --steps 40 --warmup 5 > $O/b.log 2>&1
  echo "$1 [$2]: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1)" | tee -a $O/ab.txt
}
for w in res50ppm fpn_attention db; do
  run $w ""
  run $w "tn_taps=0"
  run $w "nt_big_min_k=256"
  run $w "nt_big_min_k=128"
  run $w "tn_group=1"
  run $w "tn_group=8"
  run $w "nt_deep=0"
  run $w ""
done
