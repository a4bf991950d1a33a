"""Where a workgroup of the paired forward spends its life: shader-clock stamps (start, main loop, epilogue, end) of every workgroup
of one launch, with and without the start-up stagger."""
import ctypes, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "sgs-gnn_amd", "libsgs_hip_probe.so")        # SGS_PHASE_PROBE=1 python sgs-gnn_amd/build.py
if os.path.exists(PROBE):
    os.environ["SGS_LIB_PATH"] = PROBE
import sgs_gnn_amd as S
ops = S.ops
L = S._lib.lib()
dev = "cuda:0"
N, H = 1013, 256
sizes = S.reddit_partition_sizes(230, seed=1000, q=100_000)
idx = max(range(len(sizes)), key=lambda i: sizes[i])
ei = S.reddit_partition_stream(num_parts=230, seed=1000, nfeat=602, ncls=41, n=N, q=100_000, device=dev, only={idx})[idx].edge_index
pairs = ops.get_pairs(ei, N, build=True)
g = torch.Generator(device=dev).manual_seed(0)
codes = torch.relu(torch.randn(N, H, device=dev, generator=g)).requires_grad_(True)
fc1 = torch.nn.Linear(2 * H, H).to(dev)
fc2 = torch.nn.Linear(H, 1).to(dev)
def run(reps):
    for _ in range(reps):
        ops.edge_score(codes, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ei, p=0.3, seed=1, site=2, pairs=pairs)
L.sgs_edge_score_probe_trace.argtypes = [ctypes.c_void_p]
nwg = 1933
buf = torch.zeros(nwg * 8 * 5, dtype=torch.int64, device=dev)
out = {}
for stagger, prio in ((0, 0), (640, 0), (0, 64), (0, 68)):
    S._lib.check(L.sgs_edge_score_probe_set(stagger, prio, 9))
    run(5)
    torch.cuda.synchronize()
    buf.zero_()
    S._lib.check(L.sgs_edge_score_probe_trace(buf.data_ptr()))
    run(1)
    torch.cuda.synchronize()
    S._lib.check(L.sgs_edge_score_probe_trace(None))
    full = buf.cpu().numpy()
    t = full[:nwg * 8].reshape(nwg, 8)
    ps = full[nwg * 8:].reshape(nwg * 4, 8)
    ps = ps[ps[:, 0] > 0]
    used = t[:, 3] > 0
    t = t[used]
    t0 = t[:, 0].min()
    st, ml, ep, en = (t[:, k] - t0 for k in range(4))
    d = {"workgroups": int(used.sum()), 
         "prologue": [float(np.median(ml - st)), float(np.percentile(ml - st, 90))],
         "main_loop": [float(np.median(ep - ml)), float(np.percentile(ep - ml, 90))],
         "epilogue": [float(np.median(en - ep)), float(np.percentile(en - ep, 90))],
         "life": [float(np.median(en - st)), float(np.percentile(en - st, 90))]}
    # per CU: how the two resident workgroups' phases overlap (fraction of a main loop that runs while the CU's other workgroup is in ITS main loop)
    cu = (t[:, 4] >> 32) * 65536 + (t[:, 4] & 0xFF00)          # XCC, then SE / SH / CU of HW_ID
    for x in np.unique(t[:, 4] >> 32):                             # every XCD has its own clock: stamps relative to its first
        m = (t[:, 4] >> 32) == x
        base = t[m, 0].min()
        for arr in (st, ml, ep, en):
            arr[m] = arr[m] + t0 - base
    ov = []
    for c in np.unique(cu):
        m = np.where(cu == c)[0]
        for a in m:
            tot = 0
            for b in m:
                if a == b: continue
                tot += max(0, min(ep[a], ep[b]) - max(ml[a], ml[b]))
            ov.append(tot / max(1, ep[a] - ml[a]))
    if len(ps):
        names = ["top->before last tile (NT-1 tiles of MFMAs, split, loads)", "wait for the DMA", "barrier", "next fragments + last tile"]
        d["phase_breakdown_median_p90"] = {n: [float(np.median(ps[:, k + 1] - ps[:, k])), float(np.percentile(ps[:, k + 1] - ps[:, k], 90))] for k, n in enumerate(names)}
    d["phase_total"] = float(np.median(ps[:, 4] - ps[:, 0])) if len(ps) else None
    d["main_loop_overlap_frac_with_cu_mates"] = float(np.mean(ov))
    d["workgroups_per_cu"] = float(len(cu) / len(np.unique(cu)))
    d["clock_span"] = int(en.max())
    out[f"stagger={stagger} prio={prio}"] = d
S._lib.check(L.sgs_edge_score_probe_set(-1, 0, 9))
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r3_stagger_trace.json", "w"), indent=1)
