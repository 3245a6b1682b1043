"""One rank over the **nccl** (= RCCL) backend with PPF_FORCE_GRADSYNC=1: the chunked all-reduce of engine.GradSync is issued on the
communication stream behind both compute streams although world == 1 (a sum over one rank is the identity), so three train steps must
leave bit-identical parameters, moments and EMA weights to three steps without any gradient exchange.
Prints one line: NCCL_SINGLE_RANK {json}."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ["PPF_FORCE_GRADSYNC"] = "1"
import torch
import torch.distributed as dist

from protopformer_amd import backbone
from protopformer_amd.engine import FlatAdamW, ReplayedTrainStep, make_grad_sync, train_one_step
from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)


def run(sync_on, steps=3, replay=False):
    backbone._KEEP_CACHE.clear()                       # DropPath stream restarts: identical draws in both runs
    torch.manual_seed(7)
    m = construct_PPNet("deit_tiny_patch16_224", pretrained=False, img_size=224, prototype_shape=(200, 64, 1, 1), num_classes=20, reserve_layers=[11],
                        reserve_token_nums=[81], use_global=True, use_ppc_loss=True, global_proto_per_class=5, add_on_layers_type="regular").to(dev)
    m.train()
    opt = FlatAdamW(m, weight_decay=0.05, ema_decay=0.999)
    sync = make_grad_sync(m, opt) if sync_on else None
    g = torch.Generator(device=dev).manual_seed(100)
    img = torch.randn(8, 3, 224, 224, device=dev, generator=g)
    lab = torch.randint(0, 20, (8,), device=dev, generator=g)
    crit = CrossEntropyLoss()
    losses = []
    step = ReplayedTrainStep(m, crit, opt, epoch=20, grad_sync=sync, warmup=1) if replay else None      # eager, recorded, then replayed steps
    for _ in range(steps):
        loss, _, _ = step(img, lab) if replay else train_one_step(m, crit, img, lab, opt, epoch=20, grad_sync=sync)
        losses.append(float(loss))
    torch.cuda.synchronize()
    launched = 0 if sync is None else sync.launched
    return m.flat_store().params.clone(), opt.exp_avg.clone(), opt.ema.clone(), losses, launched


p0, m0, e0, l0, _ = run(False)
dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1, device_id=dev)
p1, m1, e1, l1, launched = run(True)
# the bench's default execution: the recorded command list, with the collectives as live entries of the list
p2, m2, e2, l2, launched2 = run(True, steps=4, replay=True)
p3, _, _, l3, _ = run(False, steps=4)
out = dict(backend=dist.get_backend(), world=dist.get_world_size(), collectives=launched, params_equal=bool(torch.equal(p0, p1)),
           moments_equal=bool(torch.equal(m0, m1)), ema_equal=bool(torch.equal(e0, e1)), losses_equal=l0 == l1, loss=l1[-1],
           replay_collectives=launched2, replay_params_equal=bool(torch.equal(p2, p3)), replay_losses_equal=l2 == l3)
print("NCCL_SINGLE_RANK " + json.dumps(out), flush=True)
dist.destroy_process_group()
