"""Per-step breakdown of a rocprofv3 kernel trace of bench.py (windows delimited by prior draws)."""
import csv, glob, collections, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f'{d}/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    return n.replace('sgs::(anonymous namespace)::', '').replace('void ', '')[:64]
starts = [i for i, r in enumerate(rows) if 'reduce_partial<1>' in r['Kernel_Name']]
def window(j):
    s = starts[j]; e = starts[j + 1] if j + 1 < len(starts) else len(rows)
    return rows[s:e]
L, R = [], []
for j in range(len(starts) // 2, len(starts) - 1):
    w = window(j)
    span = (int(w[-1]['End_Timestamp']) - int(w[0]['Start_Timestamp'])) / 1e3
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in w) / 1e3
    learned = any('edge_score_kernel<8, true' in r['Kernel_Name'] for r in w)
    (L if learned else R).append((len(w), span, busy))
for name, g in (("learned-win", L), ("random-win", R)):
    if g:
        print(f"{name}: n={len(g)} kernels={sum(x[0] for x in g)/len(g):.0f} span_us={sum(x[1] for x in g)/len(g):.0f} busy_us={sum(x[2] for x in g)/len(g):.0f}")
agg = collections.Counter(); cnt = collections.Counter(); nwin = 0
for j in range(len(starts) // 2, len(starts) - 1):
    nwin += 1
    for r in window(j):
        agg[short(r['Kernel_Name'])] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        cnt[short(r['Kernel_Name'])] += 1
tot = sum(agg.values())
print(f"windows={nwin} total busy us={tot:.0f}")
for k, v in agg.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    print(f"{v/nwin:8.1f}us/win {100*v/tot:5.1f}% n/win={cnt[k]/nwin:5.1f} avg={v/cnt[k]:7.1f}us  {k}")
