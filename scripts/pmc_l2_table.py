"""Per-kernel L2 (TCC) load from one rocprofv3 --pmc pass (TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_MISS_sum) over bench.py:
requests and channel-busy cycles per train step, ranked by busy cycles.  usage: pmc_l2_table.py <db> <steps profiled>"""
import re
import sqlite3
import sys


def main(path, steps):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    by = {}
    for k, cn, v, n in rows:
        d = by.setdefault(k, {})
        d[cn] = v
        d["n"] = n
    tot_busy = sum(d.get("TCC_BUSY_sum", 0) for d in by.values())
    tot_req = sum(d.get("TCC_REQ_sum", 0) for d in by.values())
    tot_cyc = sum(d.get("TCC_CYCLE_sum", 0) for d in by.values())
    print(f"per train step ({steps:g} steps profiled): TCC requests {tot_req / steps / 1e6:.1f} M, channel-busy cycles {tot_busy / steps / 1e6:.1f} M "
          f"(= {tot_busy / steps / 128 / 2.1e3:.0f} us of all 128 channels at 2.1 GHz), busy / cycle over the kernels' own durations {tot_busy / max(tot_cyc, 1):.2f}")
    print(f"{'kernel':84s} {'launch/step':>11s} {'Mreq/step':>10s} {'busy Mcyc/step':>14s} {'busy%':>6s} {'busy/cycle':>10s} {'miss%':>6s}")
    for k, d in sorted(by.items(), key=lambda kv: -kv[1].get("TCC_BUSY_sum", 0))[:40]:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:84]
        print(f"{name:84s} {d['n'] / steps:11.1f} {d.get('TCC_REQ_sum', 0) / steps / 1e6:10.2f} {d.get('TCC_BUSY_sum', 0) / steps / 1e6:14.2f} "
              f"{100 * d.get('TCC_BUSY_sum', 0) / max(tot_busy, 1):6.1f} {d.get('TCC_BUSY_sum', 0) / max(d.get('TCC_CYCLE_sum', 1), 1):10.2f} "
              f"{100 * d.get('TCC_MISS_sum', 0) / max(d.get('TCC_REQ_sum', 1), 1):6.1f}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0)
