#!/bin/bash
# A/B of the headline kernel on the GPU box: full frame and the emulated 1/8 shard, for each RL_TUNE value given (or the default).
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --no-cpu-baseline --no-check --steps ${STEPS:-3} "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mrays/s %8.2f ms' % (d['value'], d['ms_per_step']))"; }
if [ $# -eq 0 ]; then set -- default; fi
for t in "$@"; do
  if [ "$t" = default ]; then unset RL_TUNE; else export RL_TUNE=$t; fi
  echo "tune=$t  full: $(run)   shard8: $(run --emulate-shard 8)"
done
