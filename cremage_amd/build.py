"""In-tree build of libcrg_hip.so (hipcc, --offload-arch=gfx950 only).

The built library lives next to this file (cremage_amd/libcrg_hip.so): it is git-ignored but
travels to the GPU box with the source snapshot.  Objects are cached under cremage_amd/csrc/_obj
keyed on source mtime so that a rebuild only recompiles what changed.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcrg_hip.so")
LIB_F16 = os.path.join(HERE, "libcrg_hip_f16.so")  # the same sources with -DCRG_F16_BUILD: fp16 operands (crg_common.h)
SOURCES = ["crg_api.hip", "gemm_conv.hip", "conv_pp.hip", "gemm_ring.hip", "lngemm.hip", "norms.hip", "attention.hip", "small_ops.hip"]
HEADERS = [os.path.join(CSRC, "crg_common.h"), os.path.join(CSRC, "gemm_shared.h"), os.path.join(HERE, "..", "include", "crg_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
# attention.hip: without NaN-honouring semantics fmaxf lowers to plain v_max_f32 / v_max3_f32 instead of a canonicalising
# v_max per MFMA output (32 extra VALU per 64-key tile in a VALU-bound loop).  Infinities (the -inf key mask) are kept.
EXTRA_FLAGS = {"attention.hip": ["-fno-honor-nans"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(job):
    src, f16 = job
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace(".hip", "_f16.o" if f16 else ".o"))
    if not (_newer(s, o) or any(_newer(h, o) for h in HEADERS)):
        return o, False
    cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-DCRG_F16_BUILD"] if f16 else []) + ["-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return o, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    jobs = [(s, False) for s in SOURCES] + [(s, True) for s in SOURCES]
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(_compile, jobs))
    for lib, part in ((LIB, res[:len(SOURCES)]), (LIB_F16, res[len(SOURCES):])):
        if any(ch for _, ch in part) or not os.path.exists(lib):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for o, _ in part]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
            if verbose:
                print(f"[cremage_amd.build] linked {lib}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
