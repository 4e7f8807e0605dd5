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
SOURCES = ["conv_gemm.hip", "conv_gemm_bf16.hip", "conv_wino.hip", "stem_fused.hip", "encoder_ops.hip", "smpl.hip", "losses.hip", "prepost.hip", "hpe_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-fno-fast-math"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "hpe.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source into lib/libhpe_hip.so.  Returns the library path."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    tmp = LIB + ".tmp%d" % os.getpid()
    extra = os.environ.get("HPE_EXTRA_FLAGS", "").split()
    cmd = [_hipcc()] + FLAGS + extra + ["-o", tmp] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
