"""Two data-parallel ranks on ONE GPU over gloo (RCCL refuses duplicate devices): exercises GradSync + the weight-gradient lane
end to end and checks that the ranks hold identical parameters after a few steps and that the averaged gradient equals the
single-process gradient of the concatenated batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist
import bench
from protopformer_amd.engine import FlatAdamW, ReplayedTrainStep, make_grad_sync, train_one_step
from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)

def build(seed=7):
    torch.manual_seed(seed)
    m = construct_PPNet("deit_tiny_patch16_224", pretrained=False, img_size=224, prototype_shape=(200, 64, 1, 1), num_classes=20, reserve_layers=[11],
                        reserve_token_nums=[81], use_global=True, use_ppc_loss=True, global_proto_per_class=5, add_on_layers_type="regular").to(dev)
    m.train()
    for blk in m.features.blocks:
        blk.drop_path_rate = 0.0
    return m

g = torch.Generator(device=dev).manual_seed(100)
img_all = torch.randn(2 * 8, 3, 224, 224, device=dev, generator=g); lab_all = torch.randint(0, 20, (16,), device=dev, generator=g)
img, lab = img_all[rank * 8:(rank + 1) * 8], lab_all[rank * 8:(rank + 1) * 8]
# different seeds per rank, as the reference (main.py:254 seed + rank): make_grad_sync() must make the replicas identical
m = build(7 + rank); opt = FlatAdamW(m, weight_decay=0.05, ema_decay=0.999); sync = make_grad_sync(m, opt); crit = CrossEntropyLoss()
p0 = m.flat_store().params
lo, hi = p0.clone(), p0.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert bool(torch.equal(lo, hi)), "rank-0 broadcast did not make the replicas identical"
assert bool(torch.equal(opt.ema, p0)), "EMA copy must follow the broadcast parameters"
for it in range(3):
    loss, cov, mean = train_one_step(m, crit, img, lab, opt, epoch=20, grad_sync=sync)
# three more steps through the recorded command list (bench.py's default execution): eager, recorded, replayed
rstep = ReplayedTrainStep(m, crit, opt, epoch=20, grad_sync=sync, warmup=1)
for it in range(3):
    loss, cov, mean = rstep(img, lab)
torch.cuda.synchronize()
p = m.flat_store().params
lo, hi = p.clone(), p.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
same = bool(torch.equal(lo, hi))
# gradient check: one more backward with sync vs a single-process backward over the concatenated batch
opt.zero_grad(); m._grad_sync = sync
out, aux = m(img); l = crit(out, lab); c, mu = m.get_PPC_loss(aux[2], aux[3], aux[4], lab); (l + 0.1 * c + 0.5 * mu).backward()
m._grad_sync = None; scale = sync.finish(); torch.cuda.synchronize()
g_sync = m.flat_store().grads.clone() * scale
opt.zero_grad()
out, aux = m(img_all); l = crit(out, lab_all); c, mu = m.get_PPC_loss(aux[2], aux[3], aux[4], lab_all); (l + 0.1 * c + 0.5 * mu).backward()
torch.cuda.synchronize()
g_full = m.flat_store().grads
cos = float(torch.dot(g_sync, g_full) / (g_sync.norm() * g_full.norm()))
if rank == 0:
    print(f"ranks identical after 3 steps: {same}; loss {float(loss):.4f}; cos(avg of per-rank grads, full-batch grad) = {cos:.5f}", flush=True)
dist.destroy_process_group()
