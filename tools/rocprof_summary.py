"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel CSV: python tools/rocprof_summary.py DB OUT.csv
[--marker SUBSTR --last N]: only dispatches from the N-th last occurrence of a kernel whose name contains SUBSTR."""
import collections
import csv
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    marker, last = None, 0
    if "--marker" in sys.argv:
        marker = sys.argv[sys.argv.index("--marker") + 1]
        last = int(sys.argv[sys.argv.index("--last") + 1])
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    first = 0
    if marker:
        starts = [r[1] for r in rows if marker in r[0]]
        first = starts[-last]
    agg = collections.defaultdict(lambda: [0, 0.0, 1e18, 0])
    for name, s, e in rows:
        if s < first:
            continue
        k = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
        a = agg[k]
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    tot = sum(v[1] for v in agg.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, v[0], int(v[1]), int(v[1] / v[0]), round(100 * v[1] / tot, 3), int(v[2]), int(v[3])])
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(v[0] for v in agg.values())} dispatches")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{v[1] / 1e6:10.3f} ms {v[0]:6d} calls avg {v[1] / v[0] / 1e3:9.1f} us  {k[:90]}")


if __name__ == "__main__":
    main()
