"""Per-kernel averages of the four --pmc passes made by tools/pmc_kernels.sh: python tools/pmc_report.py <dir> <kernel regex>."""
import collections, csv, glob, json, re, sys

root, needle = sys.argv[1], re.compile(sys.argv[2])


def load(d):
    fs = glob.glob(f"{root}/{d}/*/*counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(dict))
    dur = collections.defaultdict(dict)
    if not fs:
        return out, dur
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]
        if not needle.search(name):
            continue
        clean = name.replace("(anonymous namespace)::", "").replace("void ", "")
        m = re.match(r"([\w:]+(?:<[^(]*>)?)", clean)
        short = m.group(1) if m else clean[:60]
        k = int(r["Dispatch_Id"])
        out[short][k][r["Counter_Name"]] = out[short][k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        dur[short][k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return out, dur


def avg(per, dur):
    res = {}
    for short, disp in per.items():
        ks = sorted(disp)[1:] or sorted(disp)          # drop the first (cold) dispatch
        a = collections.defaultdict(float)
        for k in ks:
            for n, v in disp[k].items():
                a[n] += v / len(ks)
        res[short] = (dict(a), sum(dur[short][k] for k in ks) / len(ks), len(ks))
    return res


sq, sq2, fe, wr = (avg(*load(d)) for d in ("sq", "sq2", "fetch", "write"))
for short in sorted(sq):
    c, d, n = sq[short]
    rec = {"kernel": short, "dispatches": n, "dur_us_under_pmc": round(d, 1)}
    if "GRBM_GUI_ACTIVE" in c:
        gui = c["GRBM_GUI_ACTIVE"] / 8
        rec["clock_GHz"] = round(gui / (d * 1e-6) / 1e9, 3)
        rec["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024), 4)
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        rec["wave_split"] = {"wait_any": round(c.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_any": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                             "active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}
        rec["lds_idx_active_frac_of_cu_cycles"] = round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (gui * 256), 4)
        rec["lds_bank_conflict_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0))
    if short in sq2:
        c2 = sq2[short][0]
        rec["insts"] = {k.replace("SQ_INSTS_", "").lower(): round(v) for k, v in c2.items() if k.startswith("SQ_INSTS_")}
        rec["active_inst_valu_quadcycles"] = round(c2.get("SQ_ACTIVE_INST_VALU", 0.0))
        rec["wait_inst_lds_quadcycles"] = round(c2.get("SQ_WAIT_INST_LDS", 0.0))
    if short in fe:
        rec["FETCH_SIZE_KB_raw"] = round(fe[short][0].get("FETCH_SIZE", 0.0), 1)
    if short in wr:
        w = wr[short][0]
        rec["WRITE_SIZE_KB"] = round(w.get("WRITE_SIZE", 0.0), 1)
        h, m = w.get("TCC_HIT_sum", 0.0), w.get("TCC_MISS_sum", 0.0)
        rec["l2_hit_rate"] = round(h / (h + m), 4) if h + m else None
    if "FETCH_SIZE_KB_raw" in rec and "WRITE_SIZE_KB" in rec:
        rec["hbm_bytes_per_launch"] = int(2 * rec["FETCH_SIZE_KB_raw"] * 1024 + rec["WRITE_SIZE_KB"] * 1024)      # FETCH_SIZE doubled (gfx950, 16-B/lane reads)
    print(json.dumps(rec))
