"""Per-kernel statistics of a rocprofv3 --kernel-trace result database (rocpd sqlite), in the column layout of
`rocprofv3 --stats` (kernel_stats.csv): Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs.

    python tools/kernel_stats.py <kt_results.db> > kernel_stats.csv
"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = list(cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                            "from kernels group by name order by 3 desc"))
    total = float(sum(r[2] for r in rows)) or 1.0
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, n, tot, avg, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name.replace('"', "'"), n, tot, avg, 100.0 * tot / total, mn, mx))


if __name__ == "__main__":
    main()
