import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f'{d}/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    return n.replace('sgs::(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')[:70]
# segments = maximal runs separated by > 10 ms of idle time; the last 12 are the replays (3 each of G1, G2R, G2L, G small)
segs, cur = [], [rows[0]]
for prev, r in zip(rows, rows[1:]):
    if int(r['Start_Timestamp']) - int(prev['End_Timestamp']) > 10_000_000:
        segs.append(cur); cur = []
    cur.append(r)
segs.append(cur)
names = ["G0 (parameter-independent prefix: prior draw, CSR of the random graph, unit norm) big",
         "G1 (scores, learned draw, encoders, gate counts) big", "G2R (random backward + Adam) big",
         "G2L (learned backward + 2 Adam) big", "G (unsampled step) small"]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # trailing segments that are not traced replays (g1_trace.py's HIP-event loop)
segs = segs[:len(segs) - skip] if skip else segs
if len(segs) < 3 * len(names):       # capture without a prefix graph: no G0 segment
    names = names[1:]
segs = segs[-3 * len(names):]
for gi in range(len(names)):
    seg = segs[gi * 3 + 2]
    t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    print(f"== {names[gi]}: kernels={len(seg)} span={(t1 - t0) / 1e3:.1f}us busy={busy / 1e3:.1f}us")
    prev = t0
    for r in seg:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"  +{(st - t0) / 1e3:8.1f} gap={(st - prev) / 1e3:6.1f} dur={(en - st) / 1e3:7.1f}  {short(r['Kernel_Name'])}")
        prev = en
