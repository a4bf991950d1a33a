#!/bin/bash
# per-kernel timeline of the captured step segments -> gpurun_out/trace/timeline.txt
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 tools/g1_trace.py > $OUT/run.log 2>&1
python3 tools/g1_trace_analyze.py $OUT/kt > $OUT/timeline.txt
tail -3 $OUT/run.log
