"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs of the same command).
Applies MI355X_MICROARCH.md's gfx950 correction: FETCH_SIZE reports half of a wide coalesced read stream (unit: KB).
usage: rocpd_pmc.py <fetch.db> <write.db> <out.json>"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {n: (v, k) for n, v, k in rows}


def main(fdb, wdb, out):
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    kern = {}
    for name in sorted(set(f) | set(w), key=lambda n: -(2 * f.get(n, (0, 0))[0] + w.get(n, (0, 0))[0]) * max(f.get(n, (0, 0))[1], w.get(n, (0, 0))[1])):
        fk, wk = f.get(name, (None, 0))[0], w.get(name, (None, 0))[0]
        kern[name] = {"launches": max(f.get(name, (0, 0))[1], w.get(name, (0, 0))[1]), "fetch_kb": fk, "write_kb": wk,
                      "hbm_bytes_per_launch": None if fk is None or wk is None else (2 * fk + wk) * 1024}
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`. "
            "Units: KB per launch (average over launches). Correction per MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by "
            "exactly 2x on gfx950; WRITE_SIZE matches algorithmic byte counts. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; "
            "memory-side requests include Infinity-Cache hits. Kernels of the two HIP streams run concurrently in these passes.")
    json.dump({"note": note, "kernels": dict(list(kern.items())[:40])}, open(out, "w"), indent=1)
    for n, v in list(kern.items())[:16]:
        print(f"{n[:80]:80s} fetch {2 * (v['fetch_kb'] or 0) / 1024:8.1f} MB write {(v['write_kb'] or 0) / 1024:8.1f} MB x{v['launches']}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
