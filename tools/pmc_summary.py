"""Summarise rocprofv3 outputs of one bench step into profiles/: per-kernel time (kernel trace) and HBM traffic (PMC passes).

    python tools/pmc_summary.py <kernel_trace.db> <fetch.db> <write.db> [<mfma.db>] > summary.md

All inputs are rocprofv3 result databases (rocpd sqlite, the default output of this ROCm).  The PMC passes are separate
runs (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) of `bench.py --steps 1 --warmup 0 --no-roofline
--cpu-sample 0` with HPE_STREAMS=1; only dispatches of the LAST forward pass are counted (the one-time initialisation
forward with 2 images comes first).  FETCH_SIZE is reported in KB and doubled (gfx950 correction, MI355X_MICROARCH.md).
"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


FIRST_KERNELS = ("stem_fused_", "pad_input")  # the first launch of a forward pass (fused stem; pad pass with HPE_STEM_FUSED=0)


def last_pass(rows):
    """rows sorted by start; keep the dispatches from the last forward's first kernel on (HPE_STREAMS=1: one such launch per forward)"""
    idx = [i for i, r in enumerate(rows) if any(k in r[0] for k in FIRST_KERNELS)]
    return rows[idx[-1]:] if idx else rows


def kernel_rows(db):
    cur = sqlite3.connect(db).cursor()
    return last_pass(list(cur.execute("select name, start, end, dispatch_id from kernels order by start")))


def pmc_rows(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        raise SystemExit("no counters_collection view in %s: %s" % (db, tabs))
    cols = [d[1] for d in cur.execute("pragma table_info(%s)" % view)]
    need = {"kernel_name", "counter_name", "value", "dispatch_id"}
    if not need <= set(cols):
        raise SystemExit("unexpected columns %s" % cols)
    rows = list(cur.execute("select kernel_name, dispatch_id, counter_name, sum(value) from %s group by dispatch_id, counter_name order by dispatch_id" % view))
    # split at the last forward's first kernel
    pads = [r[1] for r in rows if any(k in r[0] for k in FIRST_KERNELS)]
    first = max(pads) if pads else 0
    return [r for r in rows if r[1] >= first]


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    kt, fdb, wdb = pos[:3]
    mdb = pos[3] if len(pos) > 3 else None
    t = {}
    for name, s, e, _d in kernel_rows(kt):
        k = short(name)
        a = t.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    fetch, write, mfma, active = {}, {}, {}, {}
    for name, _d, cname, v in pmc_rows(fdb):
        if cname == "FETCH_SIZE":
            fetch[short(name)] = fetch.get(short(name), 0.0) + v * 1024 * 2
    for name, _d, cname, v in pmc_rows(wdb):
        if cname == "WRITE_SIZE":
            write[short(name)] = write.get(short(name), 0.0) + v * 1024
    if mdb:
        for name, _d, cname, v in pmc_rows(mdb):
            d = mfma if cname == "SQ_VALU_MFMA_BUSY_CYCLES" else (active if cname == "GRBM_GUI_ACTIVE" else None)
            if d is not None:
                d[short(name)] = d.get(short(name), 0.0) + v
    print("| kernel | launches | time ms | HBM read GB | HBM write GB | HBM TB/s | MfmaUtil % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    tot_t = tot_b = 0.0
    for k, (n, ms) in sorted(t.items(), key=lambda kv: -kv[1][1]):
        fb, wb = fetch.get(k, 0.0), write.get(k, 0.0)
        util = ""
        if k in mfma and active.get(k):
            util = "%.1f" % (100.0 * mfma[k] / (active[k] / 8 * 1024))
        print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %s |" % (k, n, ms, fb / 1e9, wb / 1e9, (fb + wb) / ms / 1e9 if ms else 0, util))
        tot_t += ms
        tot_b += fb + wb
    print("\ntotal %.2f ms, %.2f GB" % (tot_t, tot_b / 1e9))
    # ---- the conv family = the encoder's launches: everything dispatched before the global average pool of the pass (the regressor's
    #      Dense layers run through conv_gemm_f32_dma_kernel too, after it).  The fraction is GENERATED here so that DESIGN.md quotes it.
    rows = kernel_rows(kt)
    pool = [i for i, r in enumerate(rows) if "avgpool" in r[0]]
    enc_rows = rows[:pool[-1]] if pool else rows
    enc_ms = sum((e - s0) / 1e6 for _n, s0, e, _d in enc_rows)
    batch = 256
    for a in sys.argv[1:]:
        if a.startswith("--batch="):
            batch = int(a.split("=", 1)[1])
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
    import hpe_amd.resnet_spec as rs

    bf16 = any("bf16" in r[0] for r in enc_rows)
    if enc_ms > 0:
        if bf16:
            gb = rs.encoder_min_bytes_per_image(2) * batch / 1e9
            print("\nconv family (encoder launches before the average pool, profiler attached, chunk streams off): %d launches, %.3f ms; "
                  "algorithmic %.2f GB (every conv output written and read once, bf16) -> %.2f TB/s = %.3f of 8 TB/s"
                  % (len(enc_rows), enc_ms, gb, gb / enc_ms, gb / enc_ms / 8.0))
        else:
            gf = 2.0 * rs.encoder_macs_per_image() * batch / 1e9
            print("\nconv family (encoder launches before the average pool, profiler attached, chunk streams off): %d launches, %.3f ms; "
                  "algorithmic %.1f GFLOP (direct-convolution FLOPs) -> %.1f TFLOP/s = %.3f of 157.3 TFLOP/s"
                  % (len(enc_rows), enc_ms, gf, gf / enc_ms, gf / enc_ms / 157.3))
    conv = [k for k in t if k.startswith("conv_gemm") or k.startswith("wino_") or k.startswith("w4_") or k.startswith("stem_fused") or k.startswith("conv1x1_stream") or k.startswith("chain_") or k.startswith("conv3_halo")]
    cf = sum(fetch.get(k, 0.0) for k in conv)
    cw = sum(write.get(k, 0.0) for k in conv)
    print("\nJSON " + json.dumps({"fetch_bytes_corrected": cf, "write_bytes": cw, "total_bytes": cf + cw, "kernels": sorted(conv)}))


if __name__ == "__main__":
    main()
