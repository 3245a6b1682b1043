"""Every switch DESIGN.md section 9 lists still RUNS at its non-default value: two train steps of a reference-fixture
micro model in a child process per value (the switches are read once per process), held against the default run -- bit-identical where the
switch only moves work between streams / libraries, to the stated rounding where it changes the arithmetic."""
import json
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CACHE = {}


def _run(arch, **env):
    key = (arch, tuple(sorted(env.items())))
    if key not in _CACHE:
        e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **env)
        for k in [k for k in e if k.startswith("PPF_") and k not in env]:
            del e[k]                                                  # the default run is the DEFAULT run whatever the caller's shell exports
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu", "switch_check.py"), arch], capture_output=True, text=True, timeout=900,
                           cwd=ROOT, env=e)
        line = [l for l in r.stdout.splitlines() if l.startswith("SWITCH_CHECK ")]
        assert r.returncode == 0 and line, (env, r.stdout[-1500:], r.stderr[-3000:])
        _CACHE[key] = json.loads(line[0][len("SWITCH_CHECK "):])
        assert _CACHE[key]["finite"], _CACHE[key]
    return _CACHE[key]


def _close(a, b, tol):
    return all(abs(x - y) <= tol * abs(y) for x, y in zip(a["losses"], b["losses"]))


# (switch value, architecture, relative tolerance on both losses; 0 = bit-identical losses AND parameter checksum)
CASES = [
    ("PPF_WGRAD_STREAM", "0", "deit", 0.0),            # everything on one stream: same kernels, same order of every reduction
    ("PPF_WGRAD_STREAM", "0", "cait", 0.0),
    ("PPF_COMPACT_RESERVED", "0", "deit", 2e-3),        # the reference's masked full-length last blocks: bf16 rounding of other row counts
    ("PPF_PROTO_KEEP_DIST", "1", "deit", 1e-5),         # backward from the saved distance map instead of the activation map
    ("PPF_TH_FUSED", "0", "cait", 3e-3),                # materialising talking-heads kernels
    ("PPF_TH_FUSED", "pv0", "cait", 3e-3),              # fused up to A / dS, per-head products as batched GEMMs
    ("PPF_PRECISE", "1", "deit", 2e-2),                 # fp32 verification mode against the bf16 product path
    ("PPF_PRECISE", "1", "cait", 2e-2),
]


@pytest.mark.parametrize("name,value,arch,tol", CASES, ids=[f"{n}={v}-{a}" for n, v, a, _ in CASES])
def test_switch_runs_at_non_default_value(name, value, arch, tol):
    base, got = _run(arch), _run(arch, **{name: value})
    if name == "PPF_WGRAD_STREAM":
        assert base["side_stream"] and not got["side_stream"]
    if name == "PPF_PRECISE":
        assert got["precise"] and not base["precise"]
    if tol == 0.0:
        assert got["losses"] == base["losses"] and got["checksum"] == base["checksum"], (base, got)
    else:
        assert _close(got, base, tol), (base["losses"], got["losses"])
        assert got["losses"][1] < got["losses"][0]                    # and it trains: same batch, AdamW


def test_gradsync_switches_single_rank_rccl():
    """PPF_FORCE_GRADSYNC=1 (RCCL collectives with one rank = identity: bit-identical), PPF_GRADSYNC_CUTS (another partition: still bit-identical),
    PPF_GRADSYNC_BF16=1 (the gradients make one bf16 round trip: 2^-9 relative on every element -> the SECOND step's loss moves at 1e-3)."""
    base = _run("deit")
    f = _run("deit", PPF_FORCE_GRADSYNC="1")
    assert f["collectives"] >= 2 * f["chunks"] and f["payload"] == "fp32" and f["losses"] == base["losses"] and f["checksum"] == base["checksum"]
    c = _run("deit", PPF_FORCE_GRADSYNC="1", PPF_GRADSYNC_CUTS="1+2+3")
    assert c["chunks"] == 5 and c["chunks"] != f["chunks"] and c["losses"] == base["losses"] and c["checksum"] == base["checksum"]
    b = _run("deit", PPF_FORCE_GRADSYNC="1", PPF_GRADSYNC_BF16="1")
    assert b["payload"] == "bf16" and b["losses"][0] == base["losses"][0] and _close(b, base, 2e-3) and b["checksum"] != base["checksum"]


def test_lib_path_switch(tmp_path):
    """PPF_LIB_PATH: another build of the library (here a byte copy) is the one that gets loaded, results unchanged."""
    from protopformer_amd import _lib
    other = str(tmp_path / "libppf_hip_copy.so")
    shutil.copy(os.path.join(ROOT, "protopformer_amd", "lib", "libppf_hip.so"), other)
    base, got = _run("deit"), _run("deit", PPF_LIB_PATH=other)
    assert got["lib"] == other and base["lib"] != other and got["losses"] == base["losses"] and got["checksum"] == base["checksum"]


def test_bench_probe_switch():
    """PPF_BENCH_PROBE=0: bench.py without its event probes (what the records cost): the line has no live roofline timing and no named path."""
    env = dict(os.environ, PPF_BENCH_PROBE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "deit_tiny", "--batch", "16", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(line[0])
    assert d["value"] > 0 and d["roofline"].get("named_path") is None, d["roofline"]
