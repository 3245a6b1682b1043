"""Side measurement (not the headline): train-step throughput of BASELINE.json's CaiT configuration -- cait_xxs24_224,
1960x192 prototypes, 196 classes, k = 121, 5 global prototypes per class, batch 128 -- on the same engine."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from protopformer_amd.engine import FlatAdamW, train_one_step
from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet
dev = torch.device("cuda", 0)
torch.manual_seed(1028)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = construct_PPNet("cait_xxs24_224", pretrained=False, img_size=224, prototype_shape=(1960, 192, 1, 1), num_classes=196, reserve_layers=[1],
                    reserve_token_nums=[121], use_global=True, use_ppc_loss=True, global_proto_per_class=5, add_on_layers_type="regular").to(dev)
m.train()
opt = FlatAdamW(m, weight_decay=0.05, ema_decay=0.99996); crit = CrossEntropyLoss()
img = torch.randn(B, 3, 224, 224, device=dev); lab = torch.randint(0, 196, (B,), device=dev)
for _ in range(3): train_one_step(m, crit, img, lab, opt, epoch=20)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): loss, _, _ = train_one_step(m, crit, img, lab, opt, epoch=20)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"cait_xxs24_224 + 1960x192 prototypes, batch {B}: {B * 10 / dt:.0f} images/s, {dt * 100:.2f} ms/step, loss {float(loss):.3f}")
