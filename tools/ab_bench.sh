#!/bin/bash
# A/B of RL_TUNE settings on the headline bench. usage: tools/ab_bench.sh <out> "<tune1>" ...   ("-" = default)
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUT
for t in "$@"; do
  if [ "$t" = "-" ]; then unset RL_TUNE; else export RL_TUNE=$t; fi
  python3 $R/bench.py --steps 4 --warmup 2 --configs "" --no-cpu-baseline --no-live-pmc --check-rows 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tune=$t', round(d['value'],1), round(d['ms_per_step'],2), d['check']['timed_frame_equals_counting_frame'])" | tee -a $R/gpurun_out/$OUT/ab.txt
done
