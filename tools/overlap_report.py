"""How much of the prefetch stream's work (staging + G0) runs WHILE the main stream has a kernel on the GPU?

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --alts 0 --epochs 2 --no-cpu-baseline --diag-steps 0
    python tools/overlap_report.py OUT [steps]
"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 460
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
stage = [r[0] for r in rows if "stage_segments_kernel" in r[2]]
ticks = [r[1] for r in rows if "loss_tick" in r[2]]
lo, hi = stage[-steps], ticks[-1]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
byq = defaultdict(list)
for r in rows:
    byq[(r[3], r[4])].append(r)
print("queues/streams:", {k: len(v) for k, v in byq.items()})
preq = min(byq, key=lambda k: len(byq[k]))          # the prefetch stream issues one staging launch + 12 prefix kernels per step: the short one
pre = byq[preq]
main = [r for k, v in byq.items() if k != preq for r in v]
main.sort()
# overlap of each pre kernel with the union of main kernels
import bisect
starts = [m[0] for m in main]
tot = ov = 0
for s, e, n, _, _ in pre:
    tot += e - s
    i = max(0, bisect.bisect_left(starts, s) - 4)
    while i < len(main) and main[i][0] < e:
        a, b = max(s, main[i][0]), min(e, main[i][1])
        if b > a:
            ov += b - a
        i += 1
print(f"prefetch-stream kernels: {len(pre)}, busy {tot/1e6:.2f} ms, of which while a main-stream kernel runs: {ov/1e6:.2f} ms ({100*ov/max(tot,1):.0f} %)")
