"""L2 -> CU read bandwidth of an MI355X as seen by 16-byte loads (and global_load_lds): working sets from L2-resident to HBM-sized.
Builds scripts/gpu/micro/l2bw.hip next to itself (hipcc must be on the box)."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "l2bw.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-Wno-unused-value", os.path.join(here, "l2bw.hip"), "-o", so])
L = ctypes.CDLL(so)
L.l2bw_run.restype = ctypes.c_float
L.l2bw_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sink = torch.zeros(16, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for wg_per_cu in (1, 2, 4):
    grid = 256 * wg_per_cu
    for ws_mb in (1, 3, 16, 64, 512):
        nbytes = ws_mb << 20
        buf = torch.randint(0, 2**31 - 1, (nbytes // 4,), dtype=torch.int32, device="cuda")
        per_wg = 256 << 10                               # every workgroup streams 256 KiB per repetition
        reps = 8
        for mode, name in ((0, "load b128"), (1, "load_lds ")):
            ms = L.l2bw_run(mode, buf.data_ptr(), nbytes, reps, per_wg, grid, 10, sink.data_ptr())
            tb = grid * per_wg * reps / (ms * 1e-3) / 1e12
            print(f"{wg_per_cu} WG/CU  working set {ws_mb:4d} MB  {name}: {tb:6.2f} TB/s  ({ms * 1e3:7.1f} us)")
