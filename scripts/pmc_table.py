"""Text table of per-kernel HBM traffic per train step from scripts/rocpd_pmc.py's JSON: pmc_table.py <traffic.json> [steps profiled]
The number of train steps in the profiled run is the launch count of adamw_kernel (one per step: timed steps, warm-up, the recording
steps of engine.ReplayedTrainStep and bench.py's single-step host measurements); the optional argument is the fallback."""
import json
import sys


def main(path, steps):
    d = json.load(open(path))["kernels"]
    for name, v in d.items():
        if "adamw_kernel" in name and v.get("launches"):
            steps = float(v["launches"])
    rows = []
    for name, v in d.items():
        b = v["hbm_bytes_per_launch"]
        if b is None:
            continue
        rows.append((b * v["launches"] / steps, name, v["launches"] / steps, b))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"HBM traffic per train step (FETCH_SIZE x2 + WRITE_SIZE, {steps:g} steps profiled incl. warm-up): {tot / 1e9:.2f} GB")
    print(f"{'kernel':96s} {'launches/step':>13s} {'MB/launch':>10s} {'GB/step':>8s} {'pct':>6s}")
    for gb, name, n, b in rows:
        print(f"{name[:96]:96s} {n:13.1f} {b / 1e6:10.1f} {gb / 1e9:8.2f} {100 * gb / tot:6.1f}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0)
