"""CPU-side ISA lint (tools/isa_lint.py): every kernel of the in-tree objects that uses LDS-DMA (`global_load_lds`) has an `s_waitcnt`
naming vmcnt on every control-flow path from a DMA to an `s_barrier`, and no instruction names the destination of an LDS read that its
lgkmcnt waits have not retired yet (the inline-asm fragment reads of conv3_halo_bf16.hip).  A compiler upgrade (or an edit) that drops such a wait -- the
round-3 race of conv_wino4.hip -- fails here, in the build container, not on the GPU box."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

from hpe_amd import build as hbuild  # noqa: E402


def _mk(seq):
    """[(mnemonic, operands)] -> instruction list with 4-byte encodings at consecutive addresses"""
    return [(0x1000 + 4 * i, 4, mn, ops) for i, (mn, ops) in enumerate(seq)]


def test_lint_flags_missing_wait_and_accepts_counted_wait():
    dma = ("global_load_lds_dwordx4", "v[2:3], off")
    bad = _mk([dma, ("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", ""), ("s_endpgm", "")])
    assert isa_lint.lint_kernel(bad) == [0x1008]
    ok = _mk([dma, dma, ("s_waitcnt", "vmcnt(1)"), ("s_barrier", ""), ("v_mfma_f32_32x32x16_bf16", "..."), ("s_barrier", ""), ("s_endpgm", "")])
    assert isa_lint.lint_kernel(ok) == []
    merged = _mk([dma, ("s_waitcnt", "vmcnt(0) lgkmcnt(0)"), ("s_barrier", ""), ("s_endpgm", "")])
    assert isa_lint.lint_kernel(merged) == []
    no_dma = _mk([("ds_read_b128", "v[0:3], v4"), ("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", ""), ("s_endpgm", "")])
    assert isa_lint.lint_kernel(no_dma) == []


def test_lint_follows_branches():
    dma = ("global_load_lds_dwordx4", "v[2:3], off")
    # loop: [top] barrier ... DMA at the bottom, backward branch to the top: the loop-carried path DMA -> barrier has no wait
    loop = _mk([("s_waitcnt", "vmcnt(0)"), ("s_barrier", ""), ("v_mfma_f32_32x32x2_f32", "..."), dma, ("s_cbranch_scc1", "65531"), ("s_endpgm", "")])
    # 65531 = -5 dwords from the instruction after the branch (index 5) -> index 0: the wait is seen, no finding
    assert isa_lint.lint_kernel(loop) == []
    loop2 = _mk([("s_waitcnt", "vmcnt(0)"), ("s_barrier", ""), ("v_mfma_f32_32x32x2_f32", "..."), dma, ("s_cbranch_scc1", "65532"), ("s_endpgm", "")])
    # -4 dwords -> index 1, the barrier itself: reached with a DMA pending
    assert isa_lint.lint_kernel(loop2) == [0x1004]
    # the wait sits in only one arm of a branch (the round-3 pattern: wave-uniform DMA issue, wait emitted on one path)
    arm = _mk([dma, ("s_cbranch_vccz", "1"), ("s_waitcnt", "vmcnt(0)"), ("s_barrier", ""), ("s_endpgm", "")])
    assert isa_lint.lint_kernel(arm) == [0x100C]


def test_lgkm_replay_flags_a_fragment_register_named_before_its_wait():
    """The second rule: inline-asm LDS reads with counted lgkmcnt waits (conv3_halo_bf16.hip).  LDS reads return in order; a register is
    safe once the wait leaves fewer younger operations outstanding than were issued after its read."""
    rd = lambda d, a: ("ds_read_b128", "v[%d:%d], v%d" % (d, d + 3, a))  # noqa: E731
    mfma = ("v_mfma_f32_32x32x16_bf16", "v[40:55], v[0:3], v[4:7], v[40:55]")
    # two groups in flight, lgkmcnt(2) retires the older one: the multiply may use v[0:7]
    ok = _mk([rd(0, 30), rd(4, 31), rd(8, 30), rd(12, 31), ("s_waitcnt", "lgkmcnt(2)"), mfma, ("s_waitcnt", "lgkmcnt(0)"), ("s_endpgm", "")])
    assert isa_lint.lint_lgkm_hazard(ok) == []
    # the same with a wait that leaves three outstanding: v[4:7] has not landed
    late = _mk([rd(0, 30), rd(4, 31), rd(8, 30), rd(12, 31), ("s_waitcnt", "lgkmcnt(3)"), mfma, ("s_endpgm", "")])
    assert [(a, m) for a, m, _ in isa_lint.lint_lgkm_hazard(late)] == [(0x1014, "v_mfma_f32_32x32x16_bf16")]
    # a register copy between the read and its wait (what the allocator did on a loop back edge)
    copy = _mk([rd(0, 30), ("v_mov_b64_e32", "v[20:21], v[2:3]"), ("s_waitcnt", "lgkmcnt(0)"), ("s_endpgm", "")])
    assert [(a, m, r) for a, m, r in isa_lint.lint_lgkm_hazard(copy)] == [(0x1004, "v_mov_b64_e32", 2)]
    # address registers of a pending read may be reused; an outstanding scalar load makes counted waits inexact (only lgkmcnt(0) retires)
    addr = _mk([rd(0, 30), ("v_add_u32_e32", "v30, 1, v30"), ("s_waitcnt", "lgkmcnt(0)"), ("v_mov_b32_e32", "v9, v0"), ("s_endpgm", "")])
    assert isa_lint.lint_lgkm_hazard(addr) == []
    smem = _mk([("s_load_dwordx2", "s[4:5], s[0:1], 0x0"), rd(0, 30), rd(4, 31), ("s_waitcnt", "lgkmcnt(1)"), ("v_mov_b32_e32", "v9, v0"), ("s_endpgm", "")])
    assert len(isa_lint.lint_lgkm_hazard(smem)) == 1


def test_objects_have_a_vmcnt_wait_between_every_dma_and_barrier():
    hbuild.build()
    objs = sorted(glob.glob(os.path.join(ROOT, "human-pose-estimation_amd", "lib", "obj", "*.o")))
    assert len(objs) >= 12
    total = 0
    findings = []
    for o in objs:
        src = os.path.join(ROOT, "human-pose-estimation_amd", "csrc", os.path.basename(o).replace(".o", ".hip"))
        if "global_load_lds" not in open(src).read():
            continue  # nothing to disassemble
        bad, n_dma = isa_lint.lint_object(o)
        total += n_dma
        findings += [(os.path.basename(o), name, ["0x%x" % a for a in addrs]) for name, addrs in bad]
    assert total >= 80, total  # conv_gemm (fp32 / bf16 / p8), both Winograd files, the chain kernel: the parser found them
    assert not findings, findings
