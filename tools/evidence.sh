#!/bin/bash
# End-of-round evidence on the GPU box (through gpurun): bench line, rocprofv3 kernel-trace summary of THE SAME command, shard rehearsal,
# cfg 5 at its stated 4096 spp.  usage: tools/evidence.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-r03z}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"; python3 -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['check'], [(c['baseline_config'], round(c['Mrays_s'],1), c['check']['timed_frame_equals_counting_frame'], c['check']['max_abs_err']) for c in d['configs']])"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-live-pmc > $OUT/kt.json 2> $OUT/kt.err || { echo "kernel-trace failed"; tail -5 $OUT/kt.err; exit 1; }
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); grep -E "rtiow_wave_kernel|rtiow_fast_general|rtc_kernel|Name" $f | cut -c1-220 > $OUT/bench_kernel_stats.csv; cat $OUT/bench_kernel_stats.csv | cut -c1-160
find $OUT/kt -name "*kernel_trace.csv" -delete
for n in 2 4 8; do python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --configs "" --emulate-shard $n 2>/dev/null | tail -1 >> $OUT/shards.jsonl; done
python3 -c "
import json
for l in open('$OUT/shards.jsonl'): d=json.loads(l); print('shard', d['config'].get('emulated_shard_of'), round(d['ms_per_step'],1), d['check']['timed_frame_equals_counting_frame'])"
CFG5_SPP=4096 CFG_COUNTING_SPP=8 python3 $R/tools/bench_configs.py cfg5 2>/dev/null | tail -1 > $OUT/cfg5_4096.json; cut -c1-400 $OUT/cfg5_4096.json
