#!/bin/bash
# Round-end measurement set (run on the GPU box from the repo root): bench line, rocprofv3 kernel stats of the same
# command, three --pmc passes over the dominant kernel, per-segment timeline of the captured graphs.
set -e
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --hipgraph 0 --fused-adam 0 --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err
python bench.py --no-cpu-baseline --steps 240 > $O/bench_240.json 2> $O/bench_240.err
python tools/graph_timing.py > $O/graph_segment_times.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/prof_scorer.py 351194 6 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/prof_scorer.py 351194 6 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/prof_scorer.py 351194 6 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/seg -- python3 $R/tools/g1_trace.py > $O/seg.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_sq $O/pmc_fetch $O/pmc_write edge_score_bf16x6 $O/scorer_pmc.json > /dev/null
python tools/g1_trace_analyze.py $O/seg > $O/graph_segments_timeline.txt
cp $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv
python - <<PY
import csv, glob, json
rows = [r for r in csv.DictReader(open(glob.glob("$O/stats/*/*kernel_trace.csv")[0])) if "edge_score_bf16x6_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-20:]                      # bench.py's roofline loop: 20 timed launches on the largest partition, issued last
avg = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last) / 1e3
line = json.loads([l for l in open("$O/stats.log") if l.startswith("{")][-1])
json.dump({"rocprofv3_kernel_trace_avg_us_last20_bf16x6": round(avg, 1), "bench_hip_events_ms_per_launch_same_run": line["roofline"]["ms_per_launch"],
           "note": "events time sgs_edge_score_fwd = W1a split/pack launch (~5 us) + this kernel"}, open("$O/scorer_agreement.json", "w"), indent=1)
PY
# keep only the small files (the merge-back limit is 64 MiB)
for d in pmc_sq pmc_fetch pmc_write; do mkdir -p $O/keep_$d; python - <<PY
import csv, glob
f = glob.glob("$O/$d/*/*counter_collection.csv")[0]
rows = [r for r in csv.reader(open(f))]
keep = [rows[0]] + [r for r in rows[1:] if "edge_score" in r[8]]
csv.writer(open("$O/keep_$d/counter_collection_edge_score.csv", "w")).writerows(keep)
PY
done
rm -rf $O/stats $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/seg
ls -la $O
