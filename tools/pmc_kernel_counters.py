"""Sum the counters of a rocprofv3 --pmc result database over the dispatches of the kernels whose name contains a substring.

    python tools/pmc_kernel_counters.py <results.db> <kernel substring>
"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    pat = "%" + sys.argv[2] + "%"
    n = cur.execute("select count(distinct dispatch_id) from counters_collection where kernel_name like ?", (pat,)).fetchone()[0]
    print("dispatches %d" % n)
    for name, v in cur.execute("select counter_name, sum(value) from counters_collection where kernel_name like ? group by counter_name", (pat,)):
        print("%-36s total %.4g   per dispatch %.4g" % (name, v, v / max(n, 1)))


if __name__ == "__main__":
    main()
