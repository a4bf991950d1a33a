#!/bin/bash
# Three rocprofv3 --pmc passes (two SQ sets + FETCH_SIZE, then WRITE_SIZE + TCC hits) over a python script; per-kernel averages as JSON lines.
#   tools/pmc_kernels.sh <out-prefix> <kernel-name-filter (regex)> <script.py> [args...]
# (--pmc only with --kernel-trace; the program itself follows `--`)
R=$PWD
P=$R/$1; F=$2; S=$R/$3; shift 3
O=$R/gpurun_out/pmck
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $S "$@" > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq2 -- python3 $S "$@" > $O/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $S "$@" > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/write -- python3 $S "$@" > $O/write.log 2>&1
cd $R
python3 tools/pmc_report.py $O "$F" > $P.jsonl
rm -rf $O/sq $O/sq2 $O/fetch $O/write
cat $P.jsonl
