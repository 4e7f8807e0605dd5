"""GPU busy / idle time of the timed steps of a `rocprofv3 --kernel-trace` run of bench.py (rocpd sqlite database):
union of the kernel intervals, idle gaps, mean number of kernels in flight, per stream when the table has one.

    python tools/timeline_gaps.py <kt_results.db> [chunks_per_step [first_kernel_substring]]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # batch chunks per step (HPE_STREAMS)
    first = sys.argv[3] if len(sys.argv) > 3 else "stem_fused_"
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    # the first kernel is launched once per batch chunk: every `per_step`-th launch opens a step
    steps = [starts[i:i + per_step] for i in range(0, len(starts) - per_step + 1, per_step)]
    if len(steps) < 3:
        raise SystemExit("need >= 3 steps in the trace")
    lo, hi = steps[1][0], steps[-1][0]  # from the 2nd step's first kernel to the last step's first kernel
    t0, t1 = rows[lo][1], rows[hi][1]
    seg = [r for r in rows if r[1] >= t0 and r[1] < t1]
    n_steps = len(steps) - 2
    busy, cur_end, gaps = 0, t0, []
    for _n, s, e in seg:
        e = min(e, t1)
        if s > cur_end:
            gaps.append(s - cur_end)
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    total = sum(min(e, t1) - s for _n, s, e in seg)
    wall = t1 - t0
    print("steps %d  wall %.3f ms/step  busy (union) %.3f ms/step = %.1f %%  sum of kernel times %.3f ms/step (%.2f in flight)"
          % (n_steps, wall / n_steps / 1e6, busy / n_steps / 1e6, 100.0 * busy / wall, total / n_steps / 1e6, total / busy))
    gaps.sort(reverse=True)
    print("idle gaps: %d, total %.3f ms/step, largest %s us" % (len(gaps), sum(gaps) / n_steps / 1e6, [round(g / 1e3, 1) for g in gaps[:8]]))


if __name__ == "__main__":
    main()
