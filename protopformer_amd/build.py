"""Builds protopformer_amd/lib/libppf_hip.so (the C-ABI HIP library) for gfx950 with hipcc.

    python -m protopformer_amd.build [--force]

hipcc cross-compiles without a GPU; objects are cached under protopformer_amd/lib/obj and rebuilt only
when a source or header is newer."""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libppf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
         "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", SRC]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj):
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if "warning:" in r.stderr:                       # (round 6: a reserved-register clobber warning went unseen because stderr was dropped)
        print(f"[ppf build] warnings from {os.path.basename(src)}:\n{r.stderr[:2000]}", flush=True)
    return src


def build_variant(tag, defines, verbose=True):
    """A second library lib/libppf_hip_<tag>.so compiled with extra -D switches (measurement builds for same-box A/Bs through PPF_LIB_PATH)."""
    objdir = os.path.join(LIBDIR, "obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(SRC, "*.hip")))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    def one(a):
        r = subprocess.run([HIPCC] + FLAGS + ["-D" + d for d in defines] + ["-c", a[0], "-o", a[1]], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {a[0]}:\n{r.stderr}")
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(one, zip(srcs, objs)))
    lib = os.path.join(LIBDIR, f"libppf_hip_{tag}.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print("[ppf build] linked", lib, flush=True)
    return lib


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(SRC, "*.hip")))
    hdrs = glob.glob(os.path.join(SRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append((s, o))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for done in ex.map(lambda a: _compile(*a), jobs):
                if verbose:
                    print("[ppf build] compiled", os.path.basename(done), flush=True)
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("[ppf build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":          # python -m protopformer_amd.build --variant TAG DEFINE [DEFINE ...]
        build_variant(sys.argv[2], sys.argv[3:])
    else:
        build(force="--force" in sys.argv)
