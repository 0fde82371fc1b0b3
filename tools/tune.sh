#!/bin/bash
# sweep RL_TUNE settings; usage: tools/tune.sh <variant> <spp> "a,b" "c,d" ...
export RL_RTIOW_KERNEL=$1; SPP=$2; shift 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in "$@"; do
  RL_TUNE=$t python bench.py --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$t', round(j['value'],1))"
done
