import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
q = f"""select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc} e join {disp} d on e.event_id = d.event_id
        join {sym} s on d.kernel_id = s.id join {info} i on e.pmc_id = i.id group by s.kernel_name, i.name"""
agg = collections.defaultdict(dict)
n = {}
for k, c, v, cnt in cur.execute(q):
    agg[k][c] = v; n[k] = cnt
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:14]:
    a = agg[k]
    wc = a.get("SQ_WAVE_CYCLES", 1)
    print("%-90s n=%3d" % (k[:90], n[k]))
    print("    " + "  ".join("%s=%.3g" % (c, v) for c, v in sorted(a.items())))
    print("    wait_any/wave %.2f  wait_inst/wave %.2f  active/wave %.2f  lds_conflict/lds_active %.3f  mfma_busy/(wave_cycles*4/2waves) %.3f" % (
        a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1), a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (wc * 4.0)))
