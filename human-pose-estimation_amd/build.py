"""Build libhpe_hip.so (the C-ABI library of include/hpe.h) for gfx950 with hipcc, in-tree.

    python -m hpe_amd.build          # or: from hpe_amd.build import build; build()

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhpe_hip.so")
SOURCES = ["conv_gemm.hip", "conv_gemm_bf16.hip", "conv_gemm_bf16_p8.hip", "conv_chain_bf16.hip", "conv_chain_f32.hip", "conv3_halo_bf16.hip", "conv_wino.hip", "conv_wino4.hip", "stem_fused.hip", "encoder_ops.hip", "smpl.hip", "losses.hip", "prepost.hip", "hpe_api.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-fast-math"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "hpe.h")]


def _obj_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + _deps() if os.path.exists(d))


def _flags_now():
    return " ".join(CFLAGS + os.environ.get("HPE_EXTRA_FLAGS", "").split())


def built_flags():
    """compile flags of the library on disk ('' if unknown): lets tools that need a diagnostics build (-DHPE_ABLATION) check for it"""
    tag = os.path.join(LIBDIR, "obj", "flags.txt")
    return open(tag).read() if os.path.exists(tag) else ""


def _stale():
    if not os.path.exists(LIB):
        return True
    # a library built with other flags (HPE_EXTRA_FLAGS, e.g. -DHPE_ABLATION) is stale whatever its age: without this check an
    # ablation run silently measured the production build, and an ablation build stayed loaded for later production runs
    if built_flags() != _flags_now():
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "hpe.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source (one object per file, in parallel; only the stale ones) and link lib/libhpe_hip.so.
    Returns the library path."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    extra = os.environ.get("HPE_EXTRA_FLAGS", "").split()
    tag = os.path.join(objdir, "flags.txt")
    flags_now = _flags_now()
    if force or not os.path.exists(tag) or open(tag).read() != flags_now:
        for f in os.listdir(objdir):
            os.remove(os.path.join(objdir, f))
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        if not _obj_stale(src, obj):
            return None
        cmd = [hipcc] + CFLAGS + extra + ["-c", "-o", obj + ".tmp%d" % os.getpid(), src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return "hipcc failed on %s:\n%s%s" % (os.path.basename(src), r.stdout, r.stderr)
        os.replace(obj + ".tmp%d" % os.getpid(), obj)
        return None

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
        errs = [e for e in ex.map(compile_one, jobs) if e]
    if errs:
        raise RuntimeError("\n".join(errs))
    open(tag, "w").write(flags_now)
    tmp = LIB + ".tmp%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [o for _, o in jobs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
