"""How much of the forward pass is kernel tails?  One no-grad forward of 256 images on one stream against two concurrent forwards of 128
images on two streams (same weights; the side lane is switched off so that each forward is a single in-order chain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PPF_WGRAD_STREAM"] = "0"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench

cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "deit_small"])
dev = torch.device("cuda", 0)
model, opt, crit, sync = bench.build(cfg, dev, 1028)
model.eval()
B = cfg["batch"]
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(B, 3, 224, 224, device=dev, generator=g)
halves = (img[: B // 2].contiguous(), img[B // 2:].contiguous())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    with torch.no_grad():
        model(img)


def two():
    with torch.no_grad():
        for s, x in ((s1, halves[0]), (s2, halves[1])):
            with torch.cuda.stream(s):
                model(x)


def two_interleaved_threads():
    import threading
    def run(s, x):
        with torch.no_grad(), torch.cuda.stream(s):
            model(x)
    ts = [threading.Thread(target=run, args=(s, x)) for s, x in ((s1, halves[0]), (s2, halves[1]))]
    for t in ts: t.start()
    for t in ts: t.join()


for name, fn in (("one stream, 256 images", one), ("two streams, 2 x 128 (enqueued one after the other)", two), ("two streams, two host threads", two_interleaved_threads)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name:55s} {1e3 * (time.perf_counter() - t0) / 10:7.3f} ms per 256 images")
