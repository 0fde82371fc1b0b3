#!/bin/bash
# which kernel variant / scheduler setting suits a small shard (latency-bound tail)? usage: tools/shard_ab.sh <N>
N=$1
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --no-cpu-baseline --emulate-shard $N 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],1), 'ms')"; }
for v in auto wave1024ops wave768 wave512 wave256; do RL_RTIOW_KERNEL=$v run "kernel=$v"; done
for t in "24,6" "8,6" "4,8" "48,2" "24,12"; do RL_TUNE=$t run "tune=$t"; done
RL_LPT=0 run "lpt=0"
