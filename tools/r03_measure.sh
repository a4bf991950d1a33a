#!/bin/bash
# Round-3 measurement set (run on the GPU box from the repo root; results under gpurun_out/r03, copied into profiles/ afterwards):
#   bench lines (default, the driver's window), rocprofv3 kernel stats of the bench, agreement of bench.py's HIP-event time of the dominant
#   kernel with the rocprofv3 trace of the same run, --pmc passes over the dominant kernel and over the scorer backward's kernels,
#   per-segment timeline of the captured step.  Every GPU step is chained with && : after a fault nothing else runs.
set -e
R=$PWD
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --s5 0 > $O/bench_driver_window.json 2> $O/bench_driver_window.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --epochs 2 --diag-steps 0 --alts 0 --s5 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/seg -- python3 $R/tools/g1_trace.py > $O/seg.log 2>&1
cd $R
tools/pmc_kernels.sh gpurun_out/r03/scorer_fwd_pmc "edge_score_bf16x6_kernel<8, 4, 3>" tools/prof_scorer.py 494652 6 paired > /dev/null
tools/pmc_kernels.sh gpurun_out/r03/scorer_bwd_pmc "edge_score_bf16x6_kernel|gemm_tn|scorer_bwd|endpoint_reduce" tools/bwd_chain_probe.py > /dev/null
python tools/g1_trace_analyze.py $O/seg > $O/graph_segments_timeline.txt
python tools/g1_trace.py 2>/dev/null | tail -3 > $O/graph_segment_times.txt          # HIP-event segment times WITHOUT the profiler attached
python tools/gap_report.py $O/stats 460 > $O/gap_report.txt || true
python - <<PY
import csv, glob, collections, json
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
sc = [r for r in rows if "edge_score_bf16x6_kernel<8, 4, 3>" in r["Kernel_Name"]]
last = sc[-20:]                      # bench.py's roofline loop: 20 timed launches on the largest partition, issued last
avg = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last) / 1e3
line = json.loads([l for l in open("$O/stats.log") if l.startswith("{")][-1])
ev = line["roofline"]["ms_per_launch"] * 1e3
json.dump({"rocprofv3_kernel_trace_avg_us_last20_bf16x6": round(avg, 1), "bench_hip_events_us_per_launch_same_run": round(ev, 1),
           "pack_launch_us_inside_the_events": 5.0, "relative_difference_after_the_pack": round(abs(ev - 5.0 - avg) / avg, 4),
           "edges_per_launch": line["roofline"]["edges_per_launch"],
           "note": "events time sgs_edge_score_fwd_mask = W1a split/pack launch (~5 us) + this kernel; same process, same 20 launches"},
          open("$O/scorer_agreement.json", "w"), indent=1)
PY
rm -rf $O/stats $O/seg
ls -la $O
