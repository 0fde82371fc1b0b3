#!/bin/bash
# A/B of RTIOW kernel variants on the GPU box: parity tests + a short bench per variant.
# usage: tools/ab.sh <tag> <spp> variant...
TAG=$1; SPP=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in "$@"; do
  export RL_RTIOW_KERNEL=$v
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k rtiow > $OUT/test_$v.log 2>&1; echo "$v tests exit=$? $(tail -1 $OUT/test_$v.log)"
  timeout -k 10 300 python bench.py --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err || { echo "$v bench failed"; tail -3 $OUT/bench_$v.err; continue; }
  python - <<PY
import json
j=json.load(open("$OUT/bench_$v.json"))
print("$v", round(j["value"],1), "Mrays/s", round(j["ms_per_step"],1), "ms/step  frac", round(j["roofline"]["frac"],3))
PY
done
