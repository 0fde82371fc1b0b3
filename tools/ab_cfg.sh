#!/bin/bash
# A/B of RL_TUNE settings on one config's timed kernel. usage: tools/ab_cfg.sh <out> <cfg> <spp> "<tune1>" "<tune2>" ...   ("-" = default)
OUT=$1; CFG=$2; SPP=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUT
for t in "$@"; do
  if [ "$t" = "-" ]; then unset RL_TUNE; else export RL_TUNE=$t; fi
  for rep in 1 2; do
    python3 $R/tools/cfg_workload.py $CFG $SPP 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$CFG tune=$t', round(d['Mrays_s'],1), 'Mrays/s', round(d['timed_ms'],1), 'ms')" | tee -a $R/gpurun_out/$OUT/ab.txt
  done
done
