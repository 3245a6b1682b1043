"""Timeline of the last train step in a rocprofv3 rocpd database: wall time between two optimizer kernels, busy time (union over
queues), per-queue busy time, and the largest idle gaps with the kernels around them."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n)
    return n[:60]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    a, b = ad[-2], ad[-1]
    step = rows[a + 1:b + 1]
    t0, t1 = rows[a][2], rows[b][2]
    print(f"step wall {1e-6 * (t1 - t0):.3f} ms, {len(step)} dispatches")
    ev = sorted((r[1], r[2]) for r in step)
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for s, e in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e, s))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"union busy {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms in {len(gaps)} gaps")
    qs = {}
    for n, s, e, q in step:
        qs.setdefault(q, [0, 0]); qs[q][0] += e - s; qs[q][1] += 1
    for q, (t, n) in qs.items():
        print(f"  queue {q}: {1e-6 * t:.3f} ms busy, {n} dispatches")
    hist = {}
    for g, _, _ in gaps:
        k = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else ">=100us"
        hist.setdefault(k, [0, 0]); hist[k][0] += 1; hist[k][1] += g
    print("  gaps:", {k: (v[0], f"{1e-6 * v[1]:.3f} ms") for k, v in hist.items()})
    for g, ge, gs in sorted(gaps, reverse=True)[:12]:
        before = [short(r[0]) for r in step if r[2] == ge][:1]
        after = [short(r[0]) for r in step if r[1] == gs][:1]
        print(f"  gap {1e-3 * g:8.1f} us  after {before}  before {after}")
    # per-queue idle: gaps between consecutive kernels of a queue, by (previous -> next) kernel pair; the busiest queue is the main stream
    for rank, q in enumerate(sorted(qs, key=lambda q: -qs[q][0])):
        mq = sorted((r[1], r[2], r[0]) for r in step if r[3] == q)
        pair, tot = {}, 0
        for (s0, e0, n0), (s1, e1, n1) in zip(mq, mq[1:]):
            g = s1 - e0
            if g > 0:
                tot += g
                k = (short(n0)[:34], short(n1)[:34])
                pair.setdefault(k, [0, 0]); pair[k][0] += 1; pair[k][1] += g
        span = mq[-1][1] - mq[0][0]
        print(f"{'main' if rank == 0 else 'side'} queue {q}: idle between its own kernels {1e-6 * tot:.3f} ms in {len(mq) - 1} gaps "
              f"(first kernel to last: {1e-6 * span:.3f} ms)")
        for k, (n, g) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:16 if rank == 0 else 10]:
            print(f"  {1e-3 * g:8.1f} us total, {n:3d} x {1e-3 * g / n:6.1f} us   {k[0]}  ->  {k[1]}")
    # overlap: time with >= 2 kernels running
    pts = sorted([(s, 1) for s, e in ev] + [(e, -1) for s, e in ev])
    depth, last, two = 0, None, 0
    for t, d in pts:
        if depth >= 2: two += t - last
        depth += d; last = t
    print(f"time with >= 2 kernels in flight: {1e-6 * two:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
