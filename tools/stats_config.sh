#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration; the top kernels (whole process) on stdout and as CSV.
#   tools/stats_config.sh <S2|S4|S5|S3> <out.csv> [bench args...]
R=$PWD
C=$1; OUT=$R/$2; shift 2
O=$R/gpurun_out/stats_$C
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --config $C --no-cpu-baseline "$@" > $O/run.log 2>&1
cd $R
cp $O/*/*kernel_stats.csv $OUT
python3 tools/stats_top.py $O 14
tail -c 600 $O/run.log
rm -rf $O/*/
