"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd database (a `--pmc` run).
usage: python tools/rocpd_pmc.py results.db [kernel-name-substring ...]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    want = sys.argv[2:]
    rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    acc = {}
    for name, counter, value in rows:
        short = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
        if want and not any(w in short for w in want):
            continue
        a = acc.setdefault((short, counter), [0, 0.0])
        a[0] += 1
        a[1] += value
    print("kernel,counter,launches,mean_value")
    for (k, c), (n, s) in sorted(acc.items()):
        print(f"\"{k}\",{c},{n},{s / n:.1f}")


if __name__ == "__main__":
    main()
