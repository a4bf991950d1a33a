#!/bin/bash
# rocprofv3 kernel stats of one full-Reddit-scale step (config 5 sizes on one GPU) -> gpurun_out/s5prof/
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/s5prof
rm -rf $OUT; mkdir -p $OUT
S5_STEPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/s5_probe.py > $OUT/run.log 2>&1
python3 tools/stats_top.py $OUT/kt 40 > $OUT/top.txt
grep -v "^[EWI]2026" $OUT/run.log | tail -8
head -45 $OUT/top.txt
