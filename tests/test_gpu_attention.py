"""GPU parity of the attention kernels vs the CPU oracle (oracle/ppf_oracle.py) on the same bf16-rounded q/k/v."""
import pytest
import torch

from helpers import assert_close
from oracle import ppf_oracle as O

pytestmark = pytest.mark.gpu


def _qkv(B, N, D, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(B * N, 3 * D, generator=g)).bfloat16()


def _oracle_attn(qkv, B, H, N, D, policy, self_keep):
    hd = D // H
    t = qkv.float().reshape(B, N, 3, H, hd)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * hd ** -0.5
    p = O.policy_softmax(s, policy, self_keep=self_keep)
    return (p @ v).transpose(1, 2).reshape(B * N, D), p, (q, k, v)


def _policy(B, N, keep, seed):
    g = torch.Generator().manual_seed(seed)
    pol = torch.zeros(B, N)
    pol[:, 0] = 1
    for b in range(B):
        pol[b, torch.randperm(N - 1, generator=g)[:keep] + 1] = 1
    return pol


@pytest.mark.parametrize("B,H,N,D,use_policy", [(2, 6, 197, 384, False), (3, 3, 197, 192, True), (2, 4, 196, 192, False),
                                                (4, 2, 17, 128, True), (2, 2, 65, 96, True)])
def test_attn_fwd_and_headmean(B, H, N, D, use_policy):
    from protopformer_amd import ops
    qkv = _qkv(B, N, D, 3, 1.5)
    pol = _policy(B, N, max(2, N // 3), 5) if use_policy else torch.ones(B, N)
    ref_o, ref_p, _ = _oracle_attn(qkv, B, H, N, D, pol, True)
    pol_d = pol.cuda() if use_policy else None
    out, rowmax, zinv = ops.attn_fwd(qkv.cuda(), B, H, N, D, policy=pol_d)
    # P is rounded to bf16 before P.V (fp32 accumulate), output stored in bf16: tolerance = bf16 rounding
    assert_close(out.float(), ref_o, rtol=1e-2, atol=1e-2, what="attention out")
    hm = ops.attn_headmean(qkv.cuda(), rowmax, zinv, B, H, N, D, policy=pol_d)
    assert_close(hm[:, :, :N], ref_p.mean(1), rtol=1e-3, atol=1e-6, what="head-mean probabilities")     # fp32 path
    if hm.shape[-1] > N:
        assert float(hm[:, :, N:].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,N,D,use_policy", [(2, 6, 197, 384, False), (3, 3, 197, 192, True), (2, 2, 17, 128, True), (2, 6, 82, 384, True),
                                                (1, 2, 208, 128, False), (2, 3, 96, 192, True), (2, 12, 197, 768, False)])
@pytest.mark.parametrize("with_map", [True, False], ids=["with-map", "no-map"])
def test_attn_fwd_one_launch_with_headmean(B, H, N, D, use_policy, with_map):
    """attn_fwd16_kernel: forward pass, softmax statistics and the head-mean map from ONE launch, against the oracle and against the two
    kernels it replaces (identical statistics contract for the backward kernel)."""
    from protopformer_amd import ops
    assert ops.attn_fwd_hm_ok(H, N, D)
    qkv = _qkv(B, N, D, 3, 1.5)
    pol = _policy(B, N, max(2, N // 3), 5) if use_policy else torch.ones(B, N)
    ref_o, ref_p, _ = _oracle_attn(qkv, B, H, N, D, pol, True)
    pol_d = pol.cuda() if use_policy else None
    NP = (N + 3) // 4 * 4
    hm = torch.full((B, N, NP), float("nan"), device="cuda") if with_map else None
    if with_map:
        out, rowmax, zinv = ops.attn_fwd(qkv.cuda(), B, H, N, D, policy=pol_d, headmean=hm)
    else:
        out = torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda")
        rowmax = torch.empty((B, H, N), device="cuda"); zinv = torch.empty_like(rowmax)
        ops._lib.call("ppf_attn_fwd_hm", qkv.cuda(), out, pol_d, rowmax, zinv, None, NP, B, H, N, D, 1, 0)
    assert_close(out.float(), ref_o, rtol=1e-2, atol=1e-2, what="attention out")
    out2, rowmax2, zinv2 = torch.empty_like(out), torch.empty_like(rowmax), torch.empty_like(zinv)
    ops._lib.call("ppf_attn_fwd", qkv.cuda(), out2, pol_d, rowmax2, zinv2, B, H, N, D, 1, 0)            # the 32-row two-pass kernel
    assert_close(rowmax, rowmax2, rtol=1e-5, atol=1e-5, what="row max")
    assert_close(zinv, zinv2, rtol=1e-4, atol=1e-7, what="1 / (sum + eps)")
    if with_map:
        assert_close(hm[:, :, :N], ref_p.mean(1), rtol=1e-3, atol=1e-6, what="head-mean probabilities")
        if NP > N:
            assert float(hm[:, :, N:].abs().max()) == 0.0


def test_attn_masked_rows_known_answer():
    """Policy keeping only cls: every query's mass sits on {cls, itself}."""
    from protopformer_amd import ops
    B, H, N, D = 1, 2, 40, 128
    qkv = _qkv(B, N, D, 9)
    pol = torch.zeros(B, N); pol[:, 0] = 1
    out, rowmax, zinv = ops.attn_fwd(qkv.cuda(), B, H, N, D, policy=pol.cuda())
    hm = ops.attn_headmean(qkv.cuda(), rowmax, zinv, B, H, N, D, policy=pol.cuda())[0, :, :N].cpu()
    mass = hm[:, 0] + torch.diagonal(hm)
    mass[0] = hm[0, 0]
    # (e + eps/N)/(sum + eps): when the unmasked scores sit far below the row max, eps shows up at ~eps/e
    assert float((mass - 1).abs().max()) < 1e-3


@pytest.mark.parametrize("N,mode", [(197, "no-policy"), (82, "no-policy"), (150, "no-policy"), (197, "all-masked-but-self")])
def test_attn_eps_term_known_answer(N, mode):
    """The +eps/N term of the policy softmax (deit:42) as a KNOWN ANSWER -- it is 5e-9, invisible to every 1e-3 comparison above, and round 6 moved it
    out of the per-element arithmetic (one per-row constant of the head-mean map; dropped from the bf16 output of the no-policy kernel).
    no-policy: every query scores 0 on ONE key and -200 on all others (exp underflows to exactly 0): p = (1 + c) / (1 + eps) there, c / (1 + eps)
    elsewhere with c = eps / N.  all-masked-but-self: policy 0 everywhere, q = k so that a query's own score is its row maximum: the same two values
    on / off the diagonal.  N = 197 / 82 take the packed (TAIL) instantiation, 150 the per-element one, the policy case the exact per-element one."""
    from protopformer_amd import ops
    B, H, D = 2, 6, 384
    hd, eps = D // H, 1e-6
    c = eps / N
    g = torch.Generator().manual_seed(3)
    qkv = torch.zeros(B * N, 3 * D)
    v = torch.randn(B * N, D, generator=g)
    qkv[:, 2 * D:] = v
    if mode == "no-policy":
        u = torch.zeros(hd); u[0] = 40.0                                   # q = 40 e_0; k = -40 e_0 (score -200) except key j* (k = 0: score 0)
        qkv[:, :D] = u.repeat(H)
        qkv[:, D:2 * D] = (-u).repeat(H)
        jstar = 5
        for b in range(B):
            qkv[b * N + jstar, D:2 * D] = 0.0
        pol = None
    else:
        k = torch.zeros(B * N, D)
        k[:, ::hd] = 8.0                                                   # per head: k = 8 e_0 for every token -> every score 8, ties: the self key is kept
        qkv[:, :D] = k; qkv[:, D:2 * D] = k
        pol = torch.zeros(B, N).cuda()
    qkv = qkv.bfloat16()
    NP = (N + 3) // 4 * 4
    hm = torch.full((B, N, NP), float("nan"), device="cuda")
    out, rowmax, zinv = ops.attn_fwd(qkv.cuda(), B, H, N, D, policy=pol, headmean=hm)
    hm = hm.cpu()
    big, small = (1 + c) / (1 + eps), c / (1 + eps)
    exp = torch.full((B, N, N), small)
    if mode == "no-policy":
        exp[:, :, jstar] = big
        ref_out = v.bfloat16().float().reshape(B, N, D)[:, jstar:jstar + 1].expand(B, N, D).reshape(B * N, D)
    else:
        exp[:, torch.arange(N), torch.arange(N)] = big
        ref_out = v.bfloat16().float()
    assert_close(hm[:, :, :N], exp, rtol=1e-4, atol=0.0, what=f"head-mean map, eps term ({mode}, N={N})")
    assert float(hm[:, :, N:].abs().max() if NP > N else 0.0) == 0.0
    assert_close(zinv.cpu(), torch.full((B, H, N), 1 / (1 + eps)), rtol=1e-6, what="1 / (sum + eps)")
    # output: the selected value row up to bf16 rounding (the eps-weighted sum of the other rows is 1e-6 of max |v|: below the rounding of `out`)
    assert_close(out.float().cpu(), ref_out, rtol=8e-3, atol=1e-5 * float(v.abs().max()) + 1e-3, what="attention output")


@pytest.mark.parametrize("B,H,N,D,use_policy", [(2, 6, 197, 384, False), (2, 3, 197, 192, True), (2, 4, 196, 192, False), (2, 2, 17, 128, True),
                                                 (2, 6, 82, 384, True), (2, 3, 50, 96, True), (3, 4, 128, 128, False)])
def test_attn_bwd(B, H, N, D, use_policy):
    from protopformer_amd import ops
    hd = D // H
    qkv = _qkv(B, N, D, 11, 1.2)
    g = torch.Generator().manual_seed(12)
    dout = torch.randn(B * N, D, generator=g).bfloat16()
    pol = _policy(B, N, max(2, N // 3), 5) if use_policy else torch.ones(B, N)
    x = qkv.float().clone().requires_grad_(True)
    t = x.reshape(B, N, 3, H, hd)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    p = O.policy_softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, pol, self_keep=True)
    o = (p @ v).transpose(1, 2).reshape(B * N, D)
    o.backward(dout.float())
    ref = x.grad
    pol_d = pol.cuda() if use_policy else None
    out, rowmax, zinv = ops.attn_fwd(qkv.cuda(), B, H, N, D, policy=pol_d)
    dqkv = ops.attn_bwd(qkv.cuda(), out, dout.cuda(), rowmax, zinv, B, H, N, D, policy=pol_d).float().cpu()
    scale = float(ref.abs().max())
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert_close(dqkv[:, sl], ref[:, sl], rtol=2e-2, atol=1.5e-2 * scale, what=name)   # bf16 P/dS operands + bf16 output


def test_attn_bwd_bit_identical_runs():
    """The one-pass backward adds the key tiles' dQ contributions in LDS in a fixed rotated order: no run-to-run differences."""
    from protopformer_amd import ops
    B, H, N, D = 8, 6, 197, 384
    qkv = _qkv(B, N, D, 21, 1.0).cuda()
    dout = torch.randn(B * N, D, generator=torch.Generator().manual_seed(22)).bfloat16().cuda()
    pol = _policy(B, N, 80, 7).cuda()
    out, rowmax, zinv = ops.attn_fwd(qkv, B, H, N, D, policy=pol)
    first = ops.attn_bwd(qkv, out, dout, rowmax, zinv, B, H, N, D, policy=pol).clone()
    for _ in range(5):
        again = ops.attn_bwd(qkv, out, dout, rowmax, zinv, B, H, N, D, policy=pol)
        assert torch.equal(first.view(torch.int16), again.view(torch.int16))


def test_attn_bwd_streaming_many_items_per_workgroup():
    """More (batch, head) items than the chip holds workgroups: the persistent backward kernel walks several items per workgroup, prefetching
    item i + 1 (LDS-DMA images, K / V rows) under the stores of item i.  Against the fp32 oracle on every item, and bit-identical twice."""
    from protopformer_amd import ops
    B, H, N, D = 48, 6, 197, 384                       # 288 items on <= 256 one-workgroup-per-CU slots
    hd = D // H
    qkv = _qkv(B, N, D, 31, 1.2)
    dout = torch.randn(B * N, D, generator=torch.Generator().manual_seed(32)).bfloat16()
    pol = _policy(B, N, 60, 9)
    x = qkv.float().clone().requires_grad_(True)
    t = x.reshape(B, N, 3, H, hd)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    p = O.policy_softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, pol, self_keep=True)
    (p @ v).transpose(1, 2).reshape(B * N, D).backward(dout.float())
    ref = x.grad
    out, rowmax, zinv = ops.attn_fwd(qkv.cuda(), B, H, N, D, policy=pol.cuda())
    got = ops.attn_bwd(qkv.cuda(), out, dout.cuda(), rowmax, zinv, B, H, N, D, policy=pol.cuda())
    again = ops.attn_bwd(qkv.cuda(), out, dout.cuda(), rowmax, zinv, B, H, N, D, policy=pol.cuda())
    assert torch.equal(got.view(torch.int16), again.view(torch.int16))
    dqkv = got.float().cpu()
    scale = float(ref.abs().max())
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert_close(dqkv[:, sl], ref[:, sl], rtol=2e-2, atol=1.5e-2 * scale, what=name)


def test_path_probe_counts_launches_and_algorithmic_work():
    """ppf_path_probe / ppf_path_probe_read (bench.py's roofline.named_path): HIP events around the attention forward / backward launches,
    the algorithmic flops and bytes DESIGN.md section 4 states, nothing recorded while the probe is off."""
    import ctypes
    from protopformer_amd import _lib, ops
    B, H, N, D = 4, 6, 197, 384
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).bfloat16()

    def read(tag):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.call("ppf_path_probe_read", tag, ctypes.addressof(ms), ctypes.addressof(n), ctypes.addressof(fl), ctypes.addressof(by))
        return ms.value, int(n.value), fl.value, by.value

    ao, rowmax, zinv = ops.attn_fwd(qkv, B, H, N, D)                  # probe off
    _lib.call("ppf_path_probe", 1)
    assert read(0)[1] == 0 and read(1)[1] == 0
    for _ in range(3):
        ao, rowmax, zinv = ops.attn_fwd(qkv, B, H, N, D)
    ops.attn_bwd(qkv, ao, torch.randn_like(ao), rowmax, zinv, B, H, N, D)
    torch.cuda.synchronize()
    _lib.call("ppf_path_probe", 0)
    ops.attn_fwd(qkv, B, H, N, D)                                     # probe off again: not counted
    ms, n, fl, by = read(0)
    assert n == 3 and ms > 0 and fl == 3 * 4.0 * B * H * N * N * (D // H) and by >= 3 * 8.0 * B * N * D
    ms, n, fl, _ = read(1)
    assert n == 1 and ms > 0 and fl == 10.0 * B * H * N * N * (D // H)
    assert read(2)[1] == 0
    _lib.call("ppf_path_probe", 2)
