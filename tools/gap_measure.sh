#!/bin/bash
# GPU idle time inside two whole epochs of the profiled bench run -> gpurun_out/r03/gap_report.txt (see tools/gap_report.py)
set -e
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/gapkt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --epochs 2 --diag-steps 0 --alts 0 --s5 0 > $O/gap.log 2>&1
cd $R
python tools/gap_report.py $O/gapkt 460 > $O/gap_report.txt
rm -rf $O/gapkt
head -30 $O/gap_report.txt
