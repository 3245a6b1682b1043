"""Serial sections of the last train step in a rocprofv3 rocpd database: every kernel of both queues between the last forward block
and the first backward block (the prototype / loss head), and around the optimizer, with start offsets and durations."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n)
    return n[:70]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    a, b = ad[-2], ad[-1]
    step = rows[a:b + 1]
    t0 = step[0][1]
    # head: from the last attn_fwd to the first attn_bwd
    i0 = max(i for i, r in enumerate(step) if "attn_fwd_kernel" in r[0])
    i1 = min(i for i, r in enumerate(step) if "attn_bwd" in r[0])
    print(f"== head section: {1e-3 * (step[i1][1] - step[i0][1]):.1f} us between the last attention forward and the first attention backward")
    for n, s, e, q in step[i0:i1 + 1]:
        print(f"  q{q} +{1e-3 * (s - step[i0][1]):8.1f} us  {1e-3 * (e - s):7.1f} us  {short(n)}")
    print("== step boundary (optimizer .. first attention forward)")
    j1 = min(i for i, r in enumerate(step) if "attn_fwd_kernel" in r[0])
    for n, s, e, q in step[:j1 + 1]:
        print(f"  q{q} +{1e-3 * (s - t0):8.1f} us  {1e-3 * (e - s):7.1f} us  {short(n)}")
    print("== end of backward (last 14 kernels before the optimizer)")
    for n, s, e, q in step[-14:]:
        print(f"  q{q} {1e-3 * (s - step[-1][1]):9.1f} us  {1e-3 * (e - s):7.1f} us  {short(n)}")


if __name__ == "__main__":
    main(sys.argv[1])
