#!/bin/bash
# round 6, GPU call 15: fp32-MFMA ppf_sgemm for the bottleneck head's tail -- parity tests + step time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_precise.py tests/test_gpu_train_state.py tests/test_gpu_head.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r6o_tests.log 2>&1; tail -6 gpurun_out/r6o_tests.log
python - <<'PY' 2>&1 | grep -v amdgpu
import torch, sys
sys.path.insert(0, '.')
from protopformer_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n*1e3
g=torch.Generator().manual_seed(0)
M,K,N=20992,384,384
x=torch.randn(M,K,generator=g).cuda(); w=torch.randn(N,K,generator=g).cuda()*0.05; dy=torch.randn(M,N,generator=g).cuda()
y=ops.linear_f32(x,w,None)
ref=(x.double()@w.double().t())
print("fwd  rel err", float((y.double()-ref).abs().max()/ref.abs().max()), "us", round(t(lambda: ops.linear_f32(x,w,None)),1), "TF/s", round(2*M*N*K/t(lambda: ops.linear_f32(x,w,None))/1e6,1))
dx=ops.linear_dgrad_f32(dy,w); refd=dy.double()@w.double()
print("dgrad rel err", float((dx.double()-refd).abs().max()/refd.abs().max()), "us", round(t(lambda: ops.linear_dgrad_f32(dy,w)),1))
gw=torch.zeros(N,K,device='cuda'); ops.linear_wgrad_f32(dy,x,gw); refw=dy.double().t()@x.double()
print("wgrad rel err", float((gw.double()-refw).abs().max()/refw.abs().max()), "us", round(t(lambda: ops.linear_wgrad_f32(dy,x,gw)),1))
for (M2,K2,N2) in [(1000,200,130),(777,96,64),(20992,96,192)]:
    x2=torch.randn(M2,K2,generator=g).cuda(); w2=torch.randn(N2,K2,generator=g).cuda(); y2=ops.linear_f32(x2,w2,None); r2=x2.double()@w2.double().t()
    print((M2,K2,N2), "rel err", float((y2.double()-r2).abs().max()/r2.abs().max()))
PY
{ for a in "--addon regular" "--addon bottleneck" "--addon bottleneck --proto-dim 64"; do echo "== bench.py --no-cpu-baseline --no-secondary $a"; python bench.py --no-cpu-baseline --no-secondary $a | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')"; done; } > gpurun_out/r6o_bottleneck_step.txt 2>&1; grep -v amdgpu gpurun_out/r6o_bottleneck_step.txt
