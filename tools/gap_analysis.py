"""Idle time between consecutive kernels in a rocprofv3 kernel trace of bench.py (graph mode): where the GPU waits."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f'{d}/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    return n.replace('sgs::(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')[:48]
# timed region = last 60% of the trace by kernel count (after warm-up / captures)
rows = rows[int(len(rows) * 0.5):]
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
gaps = collections.Counter(); cnt = collections.Counter()
big = 0
for a, b in zip(rows, rows[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if g > 2000:
        key = f"{short(a['Kernel_Name'])} -> {short(b['Kernel_Name'])}"
        gaps[key] += g; cnt[key] += 1; big += g
print(f"span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / (t1 - t0):.1f} %), gaps > 2 us: {big / 1e6:.2f} ms")
for k, v in gaps.most_common(14):
    print(f"{v / 1e3:9.1f} us total  n={cnt[k]:4d}  avg={v / cnt[k] / 1e3:7.1f} us   {k}")
