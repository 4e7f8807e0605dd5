"""One row per 3x3-path kernel of a bench step: launches, total ms (kernel trace) and MfmaUtil (PMC pass) -- tools/wino_variants.sh."""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import kernel_rows, pmc_rows, short  # noqa: E402

KEYS = ("wino_gemm", "wino_fused", "wino_input", "conv_gemm_f32_dma_kernel<1,")  # <1, = GEMM_CONV3 (direct 3x3)


def main():
    t = {}
    for name, s, e, _d in kernel_rows(sys.argv[1]):
        k = short(name)
        a = t.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    busy, act = {}, {}
    for name, _d, cname, v in pmc_rows(sys.argv[2]):
        d = busy if cname == "SQ_VALU_MFMA_BUSY_CYCLES" else (act if cname == "GRBM_GUI_ACTIVE" else None)
        if d is not None:
            d[short(name)] = d.get(short(name), 0.0) + v
    tot = sum(v[1] for v in t.values())
    conv = sum(v[1] for k, v in t.items() if k.startswith(("conv_gemm", "wino_", "stem_fused")))
    sel = 0.0
    for k in sorted(t):
        if not any(k.startswith(p) or p in k for p in KEYS):
            continue
        util = 100.0 * busy[k] / (act[k] / 8 * 1024) if act.get(k) else float("nan")
        print("  %-46s %3d launches %8.3f ms   MfmaUtil %5.1f %%" % (k, t[k][0], t[k][1], util))
        sel += t[k][1]
    print("  3x3 path total %.3f ms, all conv kernels %.3f ms, step (serial, all kernels) %.3f ms" % (sel, conv, tot))


if __name__ == "__main__":
    main()
