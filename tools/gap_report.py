"""Idle time on the GPU inside the bench's steady state, by the kernel that precedes each gap.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --alts 0 --epochs 2
    python tools/gap_report.py OUT [steps]

Kernels of ALL streams are merged into one busy/idle timeline (a gap = no kernel running on any stream).  The window is the last
`steps` (default 460 = two 230-partition epochs) training steps: from the start of the staging kernel of the first of them to the end of
the last loss_tick (round 3: the last adam_multi launch, which carries the tick)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 460
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
stage = [r[0] for r in rows if "stage_segments_kernel" in r[2]]
ticks = [r[1] for r in rows if "loss_tick" in r[2]] or [r[1] for r in rows if "adam_multi" in r[2]]      # (round 3: the tick rides in adam_multi)
lo, hi = stage[-steps], ticks[-1]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy_end, prev = rows[0][0], None
gaps = defaultdict(lambda: [0, 0.0])
busy = 0.0
for s, e, n in rows:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        key = (prev or "?")[:60]
        gaps[key][0] += 1
        gaps[key][1] += g
        busy_end = s
    if e > busy_end:
        busy += (e - busy_end) / 1e3
        busy_end = e
        prev = n
wall = (rows[-1][1] - rows[0][0]) / 1e3
idle = sum(v[1] for v in gaps.values())
print(f"window {wall/1e3:.2f} ms, busy {busy/1e3:.2f} ms, idle {idle/1e3:.2f} ms ({100*idle/wall:.1f} %), kernels {len(rows)}")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t/1e3:8.2f} ms  n={c:5d}  mean {t/c:7.1f} us  after {k}")
