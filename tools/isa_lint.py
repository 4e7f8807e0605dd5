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


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_lgkm_hazard(insns):
    """Replay the lgkmcnt queue of a wave over the instruction stream (program order of the text: the software-pipelined loops here
    carry the same registers in flight over the back edge as their prologue puts in flight): an LDS read returns its data when an
    `s_waitcnt lgkmcnt(N)` leaves at most N younger operations outstanding.  Any instruction that names a destination register of a read
    that is still outstanding uses stale data (or is overtaken by the returning data).  hipcc keeps this invariant for the reads it
    manages; the inline-asm reads with counted waits of conv3_halo_bf16.hip are invisible to it -- a register copy that the allocator
    puts between such a read and its wait is what this rule catches.  Returns [(address, mnemonic, register)]."""
    queue = []  # outstanding lgkm operations in issue order: set of destination VGPRs (empty: ds_write, LDS atomics without return ...)
    smem = 0    # scalar loads outstanding: they return out of order, only lgkmcnt(0) retires them for sure
    bad = []
    for addr, _, mn, ops in insns:
        if mn == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ops)
            if m:
                n = int(m.group(1))
                if n == 0:
                    queue, smem = [], 0
                elif smem == 0:
                    del queue[: max(0, len(queue) - n)]
            continue
        pending = set().union(*queue) if queue else set()
        is_ds = mn.startswith("ds_")
        parts = [x.strip() for x in ops.split(",")] if ops else []
        dest = _vregs(parts[0]) if (is_ds and mn.startswith("ds_read") and parts) else set()
        used = _vregs(",".join(parts[1:])) if dest else _vregs(ops)
        hit = (used | dest) & pending
        if hit:
            bad.append((addr, mn, min(hit)))
        if is_ds:
            queue.append(dest)
        elif mn.startswith("s_load") or mn.startswith("s_buffer_load"):
            smem += 1
        elif mn.startswith("flat_") and not mn.startswith("flat_store"):
            queue.append(set())  # FLAT loads count in lgkmcnt too; none in this library
    return bad


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
        h = lint_lgkm_hazard(insns)
        if h:
            bad.append((name + " [register of an outstanding LDS read named before its lgkmcnt wait: %s v%d]" % (h[0][1], h[0][2]), [x[0] for x in h]))
    return bad, n_dma


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "human-pose-estimation_amd", "lib", "obj", "*.o")))
    rc = 0
    for o in objs:
        bad, n_dma = lint_object(o)
        print("%-28s %3d LDS-DMA kernels, %d findings (DMA -> s_barrier path without a vmcnt wait / LDS read used before its lgkmcnt wait)"
              % (os.path.basename(o), n_dma, len(bad)))
        for name, addrs in bad:
            rc = 1
            print("    %s: s_barrier at %s" % (name, ", ".join("0x%x" % a for a in addrs[:8])))
    return rc


if __name__ == "__main__":
    sys.exit(main())
