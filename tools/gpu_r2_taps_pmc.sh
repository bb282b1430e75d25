#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02c; mkdir -p $O
for cfg in "0 1" "0 4" "1 2" "1 4"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/p -- python tools/pmc_case_taps.py $1 $2 > $O/p.log 2>&1
    f=$(find $O/p -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then echo "w8=$1 group=$2 $c"; python tools/pmc_summary.py "$f" | grep -A1 "igemm_tn_taps"; fi
    rm -rf $O/p
  done
done 2>&1 | tee $O/taps_traffic_by_config.txt
