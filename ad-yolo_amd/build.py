"""Build libadyolo_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libadyolo_hip.so")
SOURCES = ["conv.hip", "wino.hip", "wino4.hip", "wino4w.hip", "wino4p_e0.hip", "wino4p_e1.hip", "wino4p_e2.hip", "wino4p_e9.hip", "wino4p_e15.hip", "wino4p_e27.hip", "wino4p_e31.hip", "gemm.hip", "norm.hip", "seq.hip", "loss.hip", "losses.hip", "features.hip", "features_mic.hip", "conformer.hip", "wino1d.hip", "attention.hip", "aug.hip", "optim.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file additions.  wino4.hip: the depth of its B-fragment register ring (9, 12 or 18; 9 = 2304 matrix cycles ahead; 12 measured the same)
# wino.hip: no SLP vectoriser -- it pairs the float4 transform arithmetic into v_pk_fma_f32 across DIFFERENT ds_read results and gathers
# the operands with four v_mov per packed instruction (the weight-gradient loop: 223 vector instructions per 64 MFMAs with it, 159
# without; every one of them costs MFMA issue time, DESIGN section 5 "Round 4")
EXTRA_FLAGS = {"wino4.hip": ["-DW4_BRING=9"], "wino.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    common = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join(HERE, "..", "include", "adyolo_hip.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + common):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return o

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built %s (%d objects recompiled)" % (OUT, len(jobs)))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
