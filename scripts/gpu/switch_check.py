"""Two train steps of a reference-fixture micro model (tests/golden/micro_{deit,cait}.npz) under whatever PPF_* switches the parent set
(tests/test_gpu_switches.py runs it once per non-default value of every switch DESIGN.md section 9 lists).  Prints SWITCH_CHECK {json}:
the two losses, a parameter checksum and which optional paths really ran.   python scripts/gpu/switch_check.py deit|cait"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29583")
import torch
import torch.distributed as dist

from helpers import build_micro, micro
from protopformer_amd import _lib, backbone
from protopformer_amd.engine import FlatAdamW, make_grad_sync, train_one_step
from protopformer_amd.protopformer import CrossEntropyLoss

arch = sys.argv[1] if len(sys.argv) > 1 else "deit"
sd, cfg, z = micro(f"micro_{arch}.npz")
torch.cuda.set_device(0)
forced = os.environ.get("PPF_FORCE_GRADSYNC", "0") != "0"
if forced:
    dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1, device_id=torch.device("cuda", 0))
backbone._KEEP_CACHE.clear()
torch.manual_seed(0)
m = build_micro(cfg, sd).train()
opt = FlatAdamW(m, weight_decay=0.05, ema_decay=0.999)
sync = make_grad_sync(m, opt) if forced else None
img, label = torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["label"]).cuda()
crit = CrossEntropyLoss()
losses = [float(train_one_step(m, crit, img, label, opt, epoch=20, grad_sync=sync)[0]) for _ in range(2)]
torch.cuda.synchronize()
st = m.flat_store()
out = dict(arch=arch, losses=losses, checksum=float(st.params.double().abs().sum()), finite=bool(torch.isfinite(st.params).all()),
           lib=_lib.LIB_PATH, precise=bool(m.precise or os.environ.get("PPF_PRECISE", "0") != "0"), collectives=0 if sync is None else sync.launched,
           chunks=0 if sync is None else len(sync.bounds) - 1, payload=None if sync is None else sync.payload,
           side_stream=backbone.wgrad_lane(st).enabled)
print("SWITCH_CHECK " + json.dumps(out), flush=True)
if forced:
    dist.destroy_process_group()
