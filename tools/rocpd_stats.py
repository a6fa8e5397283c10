"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table.
usage: python tools/rocpd_stats.py results.db [out.csv] [--skip-first N]"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)
        d = (e - s) / 1e3
        st = stats.setdefault(short, [0, 0.0, 1e30, 0.0])
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{k}\",{v[0]},{v[1]:.1f},{v[1]/v[0]:.2f},{v[2]:.2f},{v[3]:.2f},{100*v[1]/total:.2f}")
    lines.append(f"\"TOTAL\",{sum(v[0] for v in stats.values())},{total:.1f},,,,100")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
