import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    name = r["Name"].replace("sgs::(anonymous namespace)::", "").replace("void ", "")[:64]
    print(f"{name:64s} calls={int(r['Calls']):4d} avg={float(r['AverageNs']) / 1e3:10.1f} us total={float(r['TotalDurationNs']) / 1e6:8.2f} ms")
