"""Idle gaps of the GPU inside the steady-state steps of a rocprofv3 kernel trace (rocpd database):
python tools/rocprof_gaps.py DB --marker SUBSTR --last N [--top 25]. Prints the busy union, the idle total and the largest
gaps with the kernels on either side — where a step loses time to launch latency / host work rather than to kernels."""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    marker = sys.argv[sys.argv.index("--marker") + 1]
    last = int(sys.argv[sys.argv.index("--last") + 1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    starts = [r[1] for r in rows if marker in r[0]]
    first = starts[-last]
    rows = [r for r in rows if r[1] >= first]
    short = lambda n: re.sub(r"^void ", "", re.sub(r"\(.*", "", n))[:60]
    span = rows[-1][2] - rows[0][1]
    busy_end, idle, gaps = rows[0][1], 0, []
    prev = rows[0][0]
    for name, s, e in rows:
        if s > busy_end:
            idle += s - busy_end
            gaps.append((s - busy_end, short(prev), short(name)))
        if e > busy_end:
            busy_end, prev = e, name
    print(f"span {span / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %), {len(gaps)} gaps, "
          f"{sum(1 for g in gaps if g[0] > 20000)} above 20 us")
    hist = {}
    for g, a, b in gaps:
        k = (a, b)
        c = hist.setdefault(k, [0, 0])
        c[0] += 1; c[1] += g
    print("gap time by (kernel before -> kernel after):")
    for (a, b), (n, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  {t / 1e6:8.3f} ms in {n:5d} gaps (avg {t / n / 1e3:7.1f} us)  {a} -> {b}")


if __name__ == "__main__":
    main()
