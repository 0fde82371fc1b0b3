#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/wfprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RL_RTIOW_KERNEL=wavefront
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --spp ${1:-32} > $OUT/log.txt 2>&1
cat $OUT/*/*kernel_stats.csv | cut -c1-160
