"""Build libsgs_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

    python sgs-gnn_amd/build.py [--force]

Each csrc/*.hip is compiled to an object (in parallel, only when stale) and linked into
sgs-gnn_amd/libsgs_hip.so, which travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsgs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -fno-slp-vectorize: under plain -O3 hipcc packs adjacent scalar fp32 adds / multiplies of the kernels' epilogues into v_pk_add_f32 /
# v_pk_mul_f32.  Beside MFMAs those are slower than the scalar forms (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever"), and a
# round-3 build of the paired scorer forward that contained them was NOT run-to-run deterministic at N = 33 869 (a few lanes of a wave
# off by one hidden unit's bias term in ~half of the launches; tools/dbg_det2.py: 27-60 of 60 runs with, 0 of 60 without the packing;
# tests/test_gpu_edge_score.py::test_paired_forward_is_run_to_run_deterministic_at_arxiv_size).
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"]
if os.environ.get("SGS_PHASE_PROBE") == "1":       # tools/stagger_trace.py's in-phase stamps (see csrc/edge_score.hip): a library of its own,
    FLAGS.append("-DSGS_PHASE_PROBE=1")            # loaded only through SGS_LIB_PATH, never the shipped build
    OBJ = os.path.join(HERE, "build_probe")
    LIB = os.path.join(HERE, "libsgs_hip_probe.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "sgs_hip.h"))
    return hs


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if _stale(obj, [src] + headers()):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[sgs build] linked {LIB}")
    elif verbose:
        print(f"[sgs build] {LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
