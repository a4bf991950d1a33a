#!/bin/bash
# one SQ-counter pass over the scorer forward (default variant); prints per-dispatch averages
R=$PWD
O=$R/gpurun_out/pmcq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/prof_scorer.py 351194 6 > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/prof_scorer.py 351194 6 > $O/sq2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, json
for d in ("sq", "sq2"):
    fs = glob.glob("$O/%s/*/*counter_collection.csv" % d)
    if not fs:
        print(d, "no output"); continue
    per = collections.defaultdict(dict); dur = {}
    for r in csv.DictReader(open(fs[0])):
        if "edge_score_bf16x6" not in r["Kernel_Name"]: continue
        k = int(r["Dispatch_Id"]); per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ks = sorted(per)[1:]
    avg = collections.defaultdict(float)
    for k in ks:
        for n, v in per[k].items(): avg[n] += v / len(ks)
    print(d, "dur_us", round(sum(dur[k] for k in ks) / max(len(ks), 1), 1), json.dumps({k: round(v) for k, v in avg.items()}))
PY
rm -rf $O/sq $O/sq2
