#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench run (on the GPU box through gpurun).
# usage: tools/kernel_stats.sh <out-prefix under gpurun_out/> [bench args...]
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-in-library-multi --sustain-s 0 --no-pmc "$@" > $R/gpurun_out/$OUT/bench_under_rocprof.jsonl 2> $R/gpurun_out/$OUT/err.log || { tail -5 $R/gpurun_out/$OUT/err.log; exit 1; }
cp $(find $R/gpurun_out/$OUT/trace -name '*kernel_stats.csv' | head -1) $R/gpurun_out/$OUT/kernel_stats.csv
find $R/gpurun_out/$OUT/trace -name '*.csv' -size +1M -delete
head -12 $R/gpurun_out/$OUT/kernel_stats.csv | cut -c1-220
