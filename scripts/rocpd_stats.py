"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / average duration (like --stats csv)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(path, top=40):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, n, tot, avg, mn, mx in rows[:top]:
        print(f"{short(name):110s} {n:7d} {tot / 1e6:10.3f} {avg / 1e3:9.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {100 * tot / total:6.2f}")
    print(f"TOTAL kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
