"""Build libkvq_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot; nothing is
JIT-compiled at import time on the GPU box unless the library is missing or stale.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# KVQ_BUILD_TAG=<tag> (diagnostics: same-box A/B runs of a compile-time kernel variant, tools/ab_bench.sh) builds and loads
# libkvq_hip_<tag>.so from build_<tag>/ with the extra flags recorded in build_<tag>/flags.txt (KVQ_EXTRA_HIPCC_FLAGS when the
# variant is first built); unset = the product library.
TAG = os.environ.get("KVQ_BUILD_TAG", "")
LIB = os.path.join(PKG, f"libkvq_hip_{TAG}.so" if TAG else "libkvq_hip.so")
HEADER = os.path.join(os.path.dirname(PKG), "include", "kvq_hip.h")
SOURCES = ["common.cpp", "gemm.hip", "gemm256.hip", "ln.hip", "attn.hip", "attn32.hip", "misc.hip", "plan.hip", "conv.hip", "tail.hip", "tailmm.hip", "embed.hip", "merge.hip", "vit.hip", "convnet.hip", "bottleneck.hip", "slowneck.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]
# attn.hip is VALU-bound: SLP packing of adjacent f32 ops into v_pk_* costs more v_mov than it saves, and
# NaN-honouring fmaxf inserts a canonicalising v_max per MFMA output (no NaN can arise: -inf only).
EXTRA = {"attn.hip": ["-fno-slp-vectorize", "-fno-honor-nans"], "attn32.hip": ["-fno-slp-vectorize", "-fno-honor-nans"], "tail.hip": ["-fno-slp-vectorize", "-fno-honor-nans"], "tailmm.hip": ["-fno-slp-vectorize", "-fno-honor-nans"], "embed.hip": ["-fno-slp-vectorize"], "merge.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libkvq_hip.so)")


# sources compiled several times with -D<macro>=<part> (their instantiations split over translation units: tailmm.hip's 42 kernels take ~5
# minutes in one unit, ~1.5 in four)
PARTS = {"tailmm.hip": ("KVQ_TAILMM_PART", 4)}


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def units():
    """(source path, object file name, extra flags) of every translation unit"""
    out = []
    for src in sources():
        base = os.path.basename(src)
        if base in PARTS:
            macro, n = PARTS[base]
            out += [(src, f"{base}.p{k}.o", [f"-D{macro}={k}"]) for k in range(n)]
        else:
            out.append((src, base + ".o", []))
    return out


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [HEADER] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".hpp")]
    if any(os.path.getmtime(d) > t for d in deps if os.path.exists(d)):
        return True
    # a source edited WHILE a build ran is newer than its object yet older than the library that build linked: the objects are checked too
    objdir = os.path.join(PKG, f"build_{TAG}" if TAG else "build")
    for src, oname, _ in units():
        obj = os.path.join(objdir, oname)
        if os.path.exists(obj) and os.path.getmtime(src) > os.path.getmtime(obj):
            return True
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile + link under an inter-process file lock: N ranks of one torch.distributed.run job (or a prefetch thread and
    the main thread) that all find the library missing must not write the same objects at once — the first builds, the
    others wait on the lock and then find the library fresh."""
    if not force and not is_stale():
        return LIB
    import fcntl
    objdir = os.path.join(PKG, f"build_{TAG}" if TAG else "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():      # another process built it while this one waited
                return LIB
            return _build_locked(force, verbose, objdir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool, objdir: str) -> str:
    cc = _hipcc()
    extra_flags = os.environ.get("KVQ_EXTRA_HIPCC_FLAGS", "").split()
    if TAG:                                     # a variant remembers its flags: a rebuild elsewhere reproduces the same variant
        fpath = os.path.join(objdir, "flags.txt")
        if extra_flags or not os.path.exists(fpath):
            with open(fpath, "w") as f:
                f.write(" ".join(extra_flags))
        extra_flags = open(fpath).read().split()

    def compile_one(unit):
        src, oname, uflags = unit
        obj = os.path.join(objdir, oname)
        hdrs = [HEADER] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".hpp")]
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + hdrs)):
            return obj
        tmp = f"{obj}.{os.getpid()}.tmp"          # never leave a half-written object under the final name
        cmd = ([cc] + FLAGS + EXTRA.get(os.path.basename(src), []) + uflags + extra_flags
               + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", tmp])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr[-2000:])
        os.replace(tmp, obj)
        return obj

    todo = units()
    with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
        objs = list(ex.map(compile_one, todo))
    tmp = f"{LIB}.{os.getpid()}.tmp"
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    os.replace(tmp, LIB)                          # atomic: a concurrent dlopen sees the old or the new file, never half
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
