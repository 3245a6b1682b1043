"""Reference point only (never on the product path): what the vendor GEMM library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the plain
(no fused epilogue) bf16 GEMM shapes of the train step.  Tells how far the hand-written kernels are from a tuned library."""
import torch

M, D = 256 * 197, 384
dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).bfloat16()


x = rnd(M, D); wq = rnd(3 * D, D); wp = rnd(D, D); w1 = rnd(4 * D, D); w2 = rnd(D, 4 * D)
g = rnd(M, 4 * D); dy = rnd(M, D); dqkv = rnd(M, 3 * D); dh = rnd(M, 4 * D)
cases = [
    ("fwd qkv   x Wq^T      N=1152 K=384", lambda: x @ wq.t(), 2 * M * 3 * D * D),
    ("fwd proj  x Wp^T      N=384  K=384", lambda: x @ wp.t(), 2 * M * D * D),
    ("fwd fc1   x W1^T      N=1536 K=384", lambda: x @ w1.t(), 2 * M * 4 * D * D),
    ("fwd fc2   g W2^T      N=384  K=1536", lambda: g @ w2.t(), 2 * M * 4 * D * D),
    ("dgrad fc2 dy W2       N=1536 K=384", lambda: dy @ w2, 2 * M * 4 * D * D),
    ("dgrad fc1 dh W1       N=384  K=1536", lambda: dh @ w1, 2 * M * 4 * D * D),
    ("dgrad qkv dqkv Wq     N=384  K=1152", lambda: dqkv @ wq, 2 * M * 3 * D * D),
    ("dgrad prj dy Wp       N=384  K=384", lambda: dy @ wp, 2 * M * D * D),
    ("wgrad fc1 dh^T x      1536x384 K=50432", lambda: dh.t() @ x, 2 * M * 4 * D * D),
    ("wgrad fc2 dy^T g      384x1536 K=50432", lambda: dy.t() @ g, 2 * M * 4 * D * D),
    ("wgrad qkv dqkv^T x    1152x384 K=50432", lambda: dqkv.t() @ x, 2 * M * 3 * D * D),
    ("wgrad prj dy^T x      384x384  K=50432", lambda: dy.t() @ x, 2 * M * D * D),
]
tot = 0
for name, fn, fl in cases:
    t = timeit(fn)
    tot += t
    print(f"{name:40s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s")
print(f"sum: {tot * 1e3:.3f} ms")
