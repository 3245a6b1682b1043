"""Kernels of all queues around the n-th occurrence of a main-queue gap between two named kernels (rocprofv3 rocpd database):
python scripts/rocpd_window.py DB "prev kernel substring" "next kernel substring" [occurrence] [margin_us]"""
import re, sqlite3, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n)
    return n[:70]


db, a, b = sys.argv[1:4]
occ = int(sys.argv[4]) if len(sys.argv) > 4 else 5
margin = float(sys.argv[5]) * 1e3 if len(sys.argv) > 5 else 60e3
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
step = rows[ad[-2] + 1:ad[-1] + 1]
busy = {}
for r in step:
    busy[r[3]] = busy.get(r[3], 0) + r[2] - r[1]
mainq = max(busy, key=busy.get)
mq = [r for r in step if r[3] == mainq]
hits = [(x, y) for x, y in zip(mq, mq[1:]) if a in x[0] and b in y[0]]
x, y = hits[min(occ, len(hits) - 1)]
t0 = x[2]
print(f"gap {1e-3 * (y[1] - x[2]):.1f} us between main-queue kernels; times relative to the end of the first, us")
for r in step:
    if r[2] >= x[1] - margin and r[1] <= y[2] + margin:
        print(f"  q{r[3]} {'MAIN' if r[3] == mainq else 'side'}  {1e-3 * (r[1] - t0):9.1f} .. {1e-3 * (r[2] - t0):9.1f}   {short(r[0])}")
