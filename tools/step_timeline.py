"""Print the kernel timeline of the last full train step found in a rocprofv3 rocpd database
(rocprofv3 --kernel-trace -d <dir> -- python bench.py ...): start, duration, gap to the previous kernel on the
same stream, grid, name.  Usage: python tools/step_timeline.py <results.db> [--stats]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name,start,end,stream_id,grid_x,grid_y,grid_z from kernels order by start"))
    first = [i for i, r in enumerate(rows) if "conv0_fwd_kernel" in r[0]]
    last = [i for i, r in enumerate(rows) if "adam_kernel" in r[0] or "multi_tensor_apply" in r[0]]
    end = last[-1]
    begin = max(i for i in first if i < end)
    if "--period" in sys.argv:
        # distance between the optimiser kernels of consecutive steps = the true step period, idle gaps included
        ad = [rows[i][1] for i in last]
        d = [(b - a) / 1e3 for a, b in zip(ad, ad[1:])]
        tail = d[-8:]
        print(f"step period over the last {len(tail)} steps: mean {sum(tail) / len(tail):.1f} us, min {min(tail):.1f}, max {max(tail):.1f}")
        # what runs between the optimiser kernel and the next conv0 forward
        prev_end = max(i for i in last if i < begin)
        for i in range(prev_end, begin + 1):
            n, s0, e0, st, *_ = rows[i]
            print(f"  s{st} t={(s0 - rows[prev_end][1]) / 1e3:8.1f} dur={(e0 - s0) / 1e3:7.1f} {n[:70]}")
        return
    if "--stats" in sys.argv:
        tot = {}
        for n, s, e, *_ in rows[begin:end + 1]:
            k = n.split("(")[0][:80]
            c, t = tot.get(k, (0, 0.0))
            tot[k] = (c + 1, t + (e - s) / 1e3)
        for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            print(f"{t:9.1f} us {c:4d}  {k}")
        print(f"step span {(rows[end][2] - rows[begin][1]) / 1e3:.1f} us, {end - begin + 1} kernels")
        return
    t0 = rows[begin][1]
    prev = {}
    for i in range(begin, end + 1):
        n, s, e, st, gx, gy, gz = rows[i]
        gap = (s - prev.get(st, s)) / 1e3
        prev[st] = e
        print(f"s{st} t={(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:7.1f} gap={gap:6.1f} g=({gx},{gy},{gz}) {n[:80]}")


if __name__ == "__main__":
    main()
