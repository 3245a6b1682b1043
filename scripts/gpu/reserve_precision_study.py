"""Where does the bf16 product path's rollout-map error (5 % of the map's maximum -> 3-6 of 162 reserved tokens differ from the fp32
reference, logits_own_reservation 1.1e-3) come from?  

The fp32 verification mode (PPNet.precise: == the reference to 1e-6) is run with exactly ONE stage of the product path's bf16 roundings
switched on ("one-in"), and with all of them but one ("leave-one-out"), at the BASELINE shapes; reported against the un-rounded run:
  map   max |cls_token_attn - ref| / max ref          tok   reserved tokens differing (of B x k)
  logit max |logits - ref| / max |ref|  (each run follows ITS OWN reservation: what a user switching from the reference sees)
Stages: ln (LayerNorm outputs), w (Linear weights = the bf16 shadow), qk / v (columns of the qkv GEMM output), ao (attention output),
h (MLP hidden after GELU), cols (im2col patches).  The last line per shape is the bf16 product path itself.
Measurement tooling: patches attributes of protopformer_amd.ops for the duration of a run; nothing here is on the product path.
    python scripts/gpu/reserve_precision_study.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from protopformer_amd import ops
from protopformer_amd.protopformer import construct_PPNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STAGES = ["ln", "w", "qk", "v", "ao", "h", "cols"]
CFG = {
    "deit_small": dict(arch="deit_small_patch16_224", P=2000, Dp=384, C=200, k=81, layer=11, gpc=10),
    "deit_tiny": dict(arch="deit_tiny_patch16_224", P=2000, Dp=192, C=200, k=81, layer=11, gpc=10),
    "cait_xxs24": dict(arch="cait_xxs24_224", P=1960, Dp=192, C=196, k=121, layer=1, gpc=5),
}
r16 = lambda t: t.bfloat16().float()
ORIG = {n: getattr(ops, n) for n in ("linear_f32", "layernorm_fwd_f32", "attn_fwd_f32", "th_attn_fwd_f32", "im2col_patch_f32")}


def patch(on):
    def linear_f32(x, w, bias=None, kind=0, **kw):
        y = ORIG["linear_f32"](x, r16(w) if "w" in on else w, bias, kind=kind, **kw)
        return r16(y) if ("h" in on and kind == 1) else y

    def layernorm_fwd_f32(*a, **kw):
        y = ORIG["layernorm_fwd_f32"](*a, **kw)
        return r16(y) if "ln" in on else y

    def round_qkv(qkv, D):
        if "qk" in on or "v" in on:
            qkv = qkv.clone()
            if "qk" in on:
                qkv[:, :2 * D] = r16(qkv[:, :2 * D])
            if "v" in on:
                qkv[:, 2 * D:] = r16(qkv[:, 2 * D:])
        return qkv

    def attn_fwd_f32(qkv, B_, H, N, D, **kw):
        y = ORIG["attn_fwd_f32"](round_qkv(qkv, D), B_, H, N, D, **kw)
        return r16(y) if "ao" in on else y

    def th_attn_fwd_f32(qkv, wl, bl, ww, bw, B_, H, N, D, headmean):
        y = ORIG["th_attn_fwd_f32"](round_qkv(qkv, D), wl, bl, ww, bw, B_, H, N, D, headmean)
        return r16(y) if "ao" in on else y

    def im2col_patch_f32(*a, **kw):
        y = ORIG["im2col_patch_f32"](*a, **kw)
        return r16(y) if "cols" in on else y

    for n, f in dict(linear_f32=linear_f32, layernorm_fwd_f32=layernorm_fwd_f32, attn_fwd_f32=attn_fwd_f32, th_attn_fwd_f32=th_attn_fwd_f32,
                     im2col_patch_f32=im2col_patch_f32).items():
        setattr(ops, n, f)


def unpatch():
    for n, f in ORIG.items():
        setattr(ops, n, f)


def run(m, img, on=None, precise=True):
    m.precise = precise
    if on is not None:
        patch(set(on))
    try:
        with torch.no_grad():
            logits, aux = m(img)
        return logits.float().clone(), aux[3].float().clone(), m._ppc_cache[1].long().clone()
    finally:
        unpatch()


def main():
    for name, c in CFG.items():
        torch.manual_seed(0)
        m = construct_PPNet(c["arch"], pretrained=False, img_size=224, prototype_shape=(c["P"], c["Dp"], 1, 1), num_classes=c["C"],
                            reserve_layers=[c["layer"]], reserve_token_nums=[c["k"]], use_global=True, use_ppc_loss=True,
                            global_proto_per_class=c["gpc"], add_on_layers_type="regular").cuda().train()
        for blk in m.features.blocks:
            blk.drop_path_rate = 0.0
        g = torch.Generator(device="cuda").manual_seed(77)
        img = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
        ref_logits, ref_map, ref_idx = run(m, img)

        def line(tag, res):
            logits, cmap, idx = res
            sel = lambda i: torch.zeros_like(ref_map, dtype=torch.bool).scatter_(1, i, True)
            ntok = int((sel(idx) != sel(ref_idx)).sum()) // 2
            print(f"{name:11s} {tag:22s} map {float((cmap - ref_map).abs().max() / ref_map.abs().max()):9.2e}   tok {ntok:3d} / {idx.numel():4d}"
                  f"   logit {float((logits - ref_logits).abs().max() / ref_logits.abs().max()):9.2e}", flush=True)

        line("fp32 (repeat)", run(m, img))
        for s in STAGES:
            line("one-in  " + s, run(m, img, [s]))
        line("all stages", run(m, img, STAGES))
        for s in STAGES:
            line("all but " + s, run(m, img, [t for t in STAGES if t != s]))
        line("all but qk+ln", run(m, img, [t for t in STAGES if t not in ("qk", "ln")]))
        line("all but qk+ln+w", run(m, img, [t for t in STAGES if t not in ("qk", "ln", "w")]))
        line("bf16 product path", run(m, img, None, precise=False))
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
