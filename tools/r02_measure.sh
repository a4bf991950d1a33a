#!/bin/bash
# Round-2 measurement set (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats of the driver's
# command, three --pmc passes over the dominant kernel at the bench's largest partition, per-segment timeline of the captured step.
# Every GPU step is chained with && : after a fault nothing else runs.
set -e
R=$PWD
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
EBIG=${EBIG:-494652}
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_window.json 2> $O/bench_driver_window.err
python bench.py --hipgraph 0 --fused-adam 0 --no-cpu-baseline --epochs 1 --diag-steps 0 > $O/bench_eager.json 2> $O/bench_eager.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --epochs 2 --diag-steps 0 --alts 0 > $O/stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/prof_scorer.py $EBIG 6 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/prof_scorer.py $EBIG 6 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/prof_scorer.py $EBIG 6 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/ppmc_sq -- python3 $R/tools/prof_scorer.py $EBIG 6 paired > $O/ppmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ppmc_fetch -- python3 $R/tools/prof_scorer.py $EBIG 6 paired > $O/ppmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/ppmc_write -- python3 $R/tools/prof_scorer.py $EBIG 6 paired > $O/ppmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/seg -- python3 $R/tools/g1_trace.py > $O/seg.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_sq $O/pmc_fetch $O/pmc_write "edge_score_bf16x6_kernel<8, 4, 0>" $O/scorer_pmc.json $EBIG > /dev/null
python tools/pmc_summary.py $O/ppmc_sq $O/ppmc_fetch $O/ppmc_write "edge_score_bf16x6_kernel<8, 4, 3>" $O/scorer_paired_pmc.json $EBIG > /dev/null
python tools/g1_trace_analyze.py $O/seg > $O/graph_segments_timeline.txt
python tools/g1_trace.py 2>/dev/null | tail -3 > $O/graph_segment_times.txt          # HIP-event segment times WITHOUT the profiler attached
cp $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats_whole_process.csv
python tools/gap_report.py $O/stats 460 > $O/gap_report.txt          # GPU idle time inside the two epochs of the profiled run, by preceding kernel
# per-kernel totals of the STEPS only: dispatches from the first staging launch on (everything before it builds the synthetic pool)
python - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$O/stats/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = next(int(r["Start_Timestamp"]) for r in rows if "stage_segments_kernel" in r["Kernel_Name"])
agg = collections.defaultdict(lambda: [0, 0, 10**18, 0])
for r in rows:
    if int(r["Start_Timestamp"]) < t0:
        continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[r["Kernel_Name"]]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
w = csv.writer(open("$O/bench_kernel_stats.csv", "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 2), a[2], a[3]])
PY
python - <<PY
import csv, glob, json
rows = [r for r in csv.DictReader(open(glob.glob("$O/stats/*/*kernel_trace.csv")[0])) if "edge_score_bf16x6_kernel<8, 4, 3>" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-20:]                      # bench.py's roofline loop: 20 timed launches on the largest partition, issued last
avg = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last) / 1e3
line = json.loads([l for l in open("$O/stats.log") if l.startswith("{")][-1])
json.dump({"rocprofv3_kernel_trace_avg_us_last20_bf16x6": round(avg, 1), "bench_hip_events_ms_per_launch_same_run": line["roofline"]["ms_per_launch"],
           "edges_per_launch": line["roofline"]["edges_per_launch"],
           "note": "events time sgs_edge_score_fwd = W1a split/pack launch (~5 us) + this kernel"}, open("$O/scorer_agreement.json", "w"), indent=1)
PY
for d in pmc_sq pmc_fetch pmc_write ppmc_sq ppmc_fetch ppmc_write; do mkdir -p $O/keep_$d; python - <<PY
import csv, glob
f = glob.glob("$O/$d/*/*counter_collection.csv")[0]
rows = [r for r in csv.reader(open(f))]
keep = [rows[0]] + [r for r in rows[1:] if "edge_score" in r[8]]
csv.writer(open("$O/keep_$d/counter_collection_edge_score.csv", "w")).writerows(keep)
PY
done
rm -rf $O/stats $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/ppmc_sq $O/ppmc_fetch $O/ppmc_write $O/seg
ls -la $O
