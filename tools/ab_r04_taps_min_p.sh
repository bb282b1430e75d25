# A/B of mr_tuning.tn_taps_min_p (all-taps wgrad kernel only from this many output pixels up) inside the four steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4aa; mkdir -p $O
run() { # tuning, bench args...
  t="$1"; shift
  MEGREADER_TUNING="$t" timeout 200 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer "$@" --steps 40 --warmup 5 > $O/b.log 2>&1
  echo "$* [$t]: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1)" | tee -a $O/ab.txt
}
for p in 0 3000 10000 20000 0; do
  run "tn_taps_min_p=$p" --workload fpn_attention
  run "tn_taps_min_p=$p" --workload res50ppm
  run "tn_taps_min_p=$p" --workload db
  run "tn_taps_min_p=$p" --workload crnn --batch 32
done
