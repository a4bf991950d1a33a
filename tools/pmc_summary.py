"""Summarise the three rocprofv3 --pmc passes over tools/prof_scorer.py into profiles/r01_scorer_pmc.json.
   usage: python tools/pmc_summary.py <dir_sq> <dir_fetch> <dir_write> <kernel-name-substring> <out.json>"""
import csv, glob, json, sys, collections

def load(d, needle):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    per = collections.defaultdict(dict)
    dur = {}
    for r in csv.DictReader(open(f)):
        if needle not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ks = sorted(per)[1:]                    # drop the first (cold) dispatch
    avg = collections.defaultdict(float)
    for k in ks:
        for n, v in per[k].items():
            avg[n] += v / len(ks)
    return dict(avg), sum(dur[k] for k in ks) / len(ks), len(ks)

d_sq, d_f, d_w, needle, out = sys.argv[1:6]
E_ARG = int(sys.argv[6]) if len(sys.argv) > 6 else 351194
sq, dur, n = load(d_sq, needle)
fe, _, _ = load(d_f, needle)
wr, _, _ = load(d_w, needle)
E, N, H = E_ARG, 1013, 256
clock = sq["GRBM_GUI_ACTIVE"] / 8 / (dur * 1e-6) / 1e9                 # guide: counter is the sum over the 8 XCDs
simds = 256 * 4
mfma_busy = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * simds)      # pipe-busy cycles / SIMD cycles
wave_cyc = sq["SQ_WAVE_CYCLES"]
fetch_kb, write_kb = fe["FETCH_SIZE"], wr["WRITE_SIZE"]
hit, miss = wr.get("TCC_HIT_sum", 0.0), wr.get("TCC_MISS_sum", 0.0)
rec = {
    "kernel": f"{needle} ({'sgs_edge_score_fwd_mask: the training forward, paired + mask of every scored edge kept' if ', 3>' in needle else 'sgs_edge_score_fwd'}), E={E}, N={N}, H={H}, dropout 0.3",
    "command": f"rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 tools/prof_scorer.py {E} 6  (three separate passes)",
    "dispatches_averaged": n,
    "avg_duration_us_under_pmc": round(dur, 1),
    "effective_clock_GHz": round(clock, 3),
    "mfma_busy_fraction": round(mfma_busy, 4),
    "wave_time_split": {"wait_inst_any": round(sq["SQ_WAIT_INST_ANY"] / wave_cyc, 3), "wait_any": round(sq["SQ_WAIT_ANY"] / wave_cyc, 3),
                        "active_inst_any": round(sq["SQ_ACTIVE_INST_ANY"] / wave_cyc, 3)},
    "lds_bank_conflict_cycles": sq.get("SQ_LDS_BANK_CONFLICT", 0.0),
    "FETCH_SIZE_KB_raw": round(fetch_kb, 2),
    "WRITE_SIZE_KB": round(write_kb, 2),
    "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
    "hbm_traffic_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "traffic_note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane reads); gathers are 16 B/lane. "
                    "Algorithmic HBM bytes per launch: 16 B x E edge ids + 4 B x E output + node tables (codes, U: 2 x 1 MB) + W1a (256 KB fp32, 384 KB as bf16 pieces) = 20 B x E + 2.4 MB; "
                    "the table re-reads are L2 hits.",
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
