"""Per-kernel averages of the SQ counters in a rocprofv3 --pmc rocpd database.  usage: rocpd_sq.py <db> [name filter]"""
import re
import sqlite3
import sys


def main(path, flt=""):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    by = {}
    for k, cn, v, n in rows:
        if flt and flt not in k:
            continue
        by.setdefault(k, {})[cn] = (v, n)
    for k, d in sorted(by.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0))[0] * kv[1].get("SQ_WAVE_CYCLES", (0, 1))[1])[:24]:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", k)[:70]
        wc = d.get("SQ_WAVE_CYCLES", (0, 0))[0] or 1
        print(f"{name:70s} x{list(d.values())[0][1]:<4d} " + " ".join(f"{cn[3:]}={v:.3g}" + (f"({100 * v / wc:.0f}%)" if cn != "SQ_WAVE_CYCLES" else "") for cn, (v, _) in sorted(d.items())))


if __name__ == "__main__":
    main(*sys.argv[1:3])
