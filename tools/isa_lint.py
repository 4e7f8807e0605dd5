#!/usr/bin/env python3
"""ISA lint of the LDS-DMA kernels: no path from a `global_load_lds` to an `s_barrier` without an `s_waitcnt` that names vmcnt.

Why: `global_load_lds_dwordx4` writes LDS asynchronously and is tracked by vmcnt only.  A barrier that is meant to publish DMA'd data to
the other waves needs `s_waitcnt vmcnt(N)` in front of it; hipcc's `__syncthreads()` emits that wait behind some DMA patterns and only
`lgkmcnt(0)` behind others (the round-3 race in conv_wino4.hip: whole tile blocks wrong once >= 200 workgroups stretched the DMA latency).
The kernels therefore write their waits out; this lint disassembles the device code of every object (llvm-objdump, no GPU needed) and walks
the control-flow graph of each kernel, so that a compiler upgrade or an edit that drops a wait fails in the build container.

The rule is deliberately syntactic: ANY `s_waitcnt` with a vmcnt field between the DMA and the barrier satisfies it -- counted waits
(`vmcnt(8)`: "everything but the newest slab") are how the deeper rings work, and whether the count is right is what the GPU parity
tests check.  What the lint catches is the wait that is not there at all.

    python tools/isa_lint.py [objects...]        (default: human-pose-estimation_amd/lib/obj/*.o)
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP_CANDIDATES = ["/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")]

_INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)$")
_LABEL = re.compile(r"^([0-9A-Fa-f]+)\s+<([^>]+)>:")


def objdump():
    for c in OBJDUMP_CANDIDATES:
        if c and os.path.exists(c):
            return c
    raise RuntimeError("llvm-objdump not found")


def device_disassembly(obj):
    """disassembly text of the gfx950 code object bundled in a hipcc object / shared library"""
    tmp = tempfile.mkdtemp(prefix="isa_lint_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([objdump(), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        parts = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not parts:
            return ""
        out = []
        for f in parts:
            out.append(subprocess.run([objdump(), "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout)
        return "\n".join(out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def parse_kernels(text):
    """{symbol: [(addr, size_bytes, mnemonic, operands)]}"""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = _LABEL.match(line)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            continue
        m = _INSN.match(line)
        if m and cur is not None:
            words = m.group(4).split()
            cur.append((int(m.group(3), 16), 4 * len(words), m.group(1), m.group(2)))
    return kernels


def _branch_target(addr, size, ops):
    # s_branch / s_cbranch_*: simm16 counted in dwords from the next instruction
    try:
        off = int(ops.split()[0], 0)
    except (ValueError, IndexError):
        return None
    if off >= 0x8000:
        off -= 0x10000
    return addr + size + 4 * off


def lint_kernel(insns):
    """addresses of s_barrier instructions reachable from a global_load_lds with no vmcnt wait in between"""
    if not any(i[2].startswith("global_load_lds") or (i[2].startswith("buffer_load") and " lds" in i[3]) for i in insns):
        return []
    index = {a: k for k, (a, _, _, _) in enumerate(insns)}
    n = len(insns)
    succ = [[] for _ in range(n)]
    for k, (a, sz, mn, ops) in enumerate(insns):
        if mn == "s_endpgm":
            continue
        if mn == "s_branch" or mn.startswith("s_cbranch"):
            t = _branch_target(a, sz, ops)
            if t in index:
                succ[k].append(index[t])
            if mn == "s_branch":
                continue
        if k + 1 < n:
            succ[k].append(k + 1)
    # forward dataflow: pending[k] = a DMA may be outstanding (no vmcnt wait since) when instruction k starts
    pending = [False] * n
    work = list(range(n))
    bad = set()
    while work:
        k = work.pop()
        _, _, mn, ops = insns[k]
        state = pending[k]
        if mn == "s_barrier" and state:
            bad.add(insns[k][0])
        if mn.startswith("global_load_lds") or (mn.startswith("buffer_load") and " lds" in ops):
            out = True
        elif mn == "s_waitcnt" and "vmcnt(" in ops:
            out = False
        else:
            out = state
        for s in succ[k]:
            if out and not pending[s]:
                pending[s] = True
                work.append(s)
    return sorted(bad)


def lint_object(obj):
    """[(kernel symbol, [barrier addresses])] for the kernels of one object that violate the rule; also returns how many kernels use LDS-DMA"""
    kernels = parse_kernels(device_disassembly(obj))
    bad, n_dma = [], 0
    for name, insns in kernels.items():
        if any(i[2].startswith("global_load_lds") for i in insns):
            n_dma += 1
        b = lint_kernel(insns)
        if b:
            bad.append((name, b))
    return bad, n_dma


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "human-pose-estimation_amd", "lib", "obj", "*.o")))
    rc = 0
    for o in objs:
        bad, n_dma = lint_object(o)
        print("%-28s %3d LDS-DMA kernels, %d with a DMA -> s_barrier path that has no vmcnt wait" % (os.path.basename(o), n_dma, len(bad)))
        for name, addrs in bad:
            rc = 1
            print("    %s: s_barrier at %s" % (name, ", ".join("0x%x" % a for a in addrs[:8])))
    return rc


if __name__ == "__main__":
    sys.exit(main())
