"""Which kernels run next to a given kernel in the last train step of a rocprofv3 rocpd database, and how its duration depends on them:
python scripts/rocpd_corun.py <results.db> <kernel name substring>
Per launch: duration, the kernels of the OTHER queues that overlap it and for how long; then the average duration by co-runner."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n)
    return n[:48]


def main(path, pat):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    step = rows[ad[-2] + 1:ad[-1] + 1]
    by = {}
    for n, s, e, q in step:
        if pat not in n:
            continue
        co = {}
        for n2, s2, e2, q2 in step:
            if q2 == q:
                continue
            ov = min(e, e2) - max(s, s2)
            if ov > 0:
                co[short(n2)] = co.get(short(n2), 0) + ov
        key = max(co, key=co.get) if co else "(alone)"
        frac = sum(co.values()) / (e - s)
        print(f"{1e-3 * (e - s):8.1f} us  overlapped {100 * frac:5.1f} %  " + ", ".join(f"{k} {1e-3 * v:.0f}" for k, v in sorted(co.items(), key=lambda kv: -kv[1])[:3]))
        by.setdefault(key, []).append(e - s)
    print("average duration by main co-runner:")
    for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print(f"  {k:50s} {len(v):3d} x {1e-3 * sum(v) / len(v):7.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
