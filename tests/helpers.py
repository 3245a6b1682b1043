"""Shared test helpers (fixtures loading, tolerances)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def micro(name):
    """Returns (state_dict, cfg, npz) for tests/golden/micro_{deit,cait}.npz."""
    z = load_npz(name)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta/")}
    cfg = dict(arch=m["arch"], dim=m["dim"], depth=m["depth"], heads=m["heads"], reserve_layer=m["reserve_layer"],
               reserve_k=m["reserve_k"], global_coe=m["global_coe"], num_prototypes=m["num_prototypes"],
               proto_dim=m["proto_dim"], num_classes=m["num_classes"], global_per_class=m["global_per_class"],
               protos_per_class=m["num_prototypes"] // m["num_classes"], img=m["img"], patch=16, add_on=m.get("add_on", "regular"))
    return sd, cfg, z


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_elementwise(a, b, tol, what=""):
    """The element-wise form of a `rel_err(a, b) < tol` gate: every element within tol * |b| + tol * max|b|.
    rel_err is max-abs-error over max-abs-reference (one number for the tensor); this states the same bound per element, with the
    relative term on top, so that a report of `rel_err` can never hide an element that is off by more than tol of the tensor's scale."""
    b_ = torch.as_tensor(b).detach().double().cpu()
    assert_close(torch.as_tensor(a).detach(), b_, rtol=tol, atol=tol * float(b_.abs().max()), what=what)


def assert_close(a, b, rtol, atol=0.0, what=""):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} out of tol; max abs err {float(err.max()):.3e}, "
                                 f"max rel-to-max {float(err.max() / b.abs().max().clamp_min(1e-30)):.3e}")


def check_param_tensors(z, tag, named, rtol, atol, grad_floor=None):
    """Compare {name: tensor} with the 'grad'/'step' entries of a micro fixture (full or sampled).

    grad_floor: for the post-AdamW 'step' comparison only entries whose reference gradient magnitude
    exceeds it are compared -- AdamW's first step is lr*g/(|g|+1e-8), i.e. it amplifies rounding noise of
    mathematically-zero gradients (e.g. the key bias of a softmax) to a full +-lr step."""
    n = 0
    for name, t in named.items():
        t = t.detach().double().cpu()
        if f"{tag}/{name}" in z.files:
            ref = torch.from_numpy(z[f"{tag}/{name}"]).double()
            if grad_floor is not None:
                m = torch.from_numpy(z[f"grad/{name}"]).abs() > grad_floor
                t, ref = t[m], ref[m]
            assert_close(t, ref, rtol, atol, f"{tag}/{name}")
        elif f"{tag}_idx/{name}" in z.files:
            idx = torch.from_numpy(z[f"{tag}_idx/{name}"])
            tv, ref = t.reshape(-1)[idx], torch.from_numpy(z[f"{tag}_val/{name}"]).double()
            if grad_floor is not None:
                m = torch.from_numpy(z[f"grad_val/{name}"]).abs() > grad_floor
                assert_close(tv[m], ref[m], rtol, atol, f"{tag}/{name} (sample)")
                n += 1
                continue
            assert_close(tv, ref, rtol, atol, f"{tag}/{name} (sample)")
            s = float(z[f"{tag}_sum/{name}"])
            assert abs(float(t.sum()) - s) <= 1e-3 * max(1.0, float(t.abs().sum())), f"{tag}/{name} checksum"
        else:
            raise AssertionError(f"fixture has no {tag} entry for {name}")
        n += 1
    return n


def build_micro(cfg, sd):
    """The drop-in PPNet for a micro fixture (tests/golden/micro_{deit,cait}.npz) with the reference's state dict loaded."""
    from protopformer_amd.cait import MyCait
    from protopformer_amd.deit import MyVisionTransformer
    from protopformer_amd.protopformer import PPNet
    cls = MyVisionTransformer if cfg["arch"] == "deit" else MyCait
    feats = cls(img_size=cfg["img"], patch_size=16, embed_dim=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"], drop_path_rate=0.0)
    m = PPNet(features=feats, img_size=cfg["img"], prototype_shape=[cfg["num_prototypes"], cfg["proto_dim"], 1, 1], proto_layer_rf_info=None,
              num_classes=cfg["num_classes"], reserve_layers=[cfg["reserve_layer"]], reserve_token_nums=[cfg["reserve_k"]], use_global=True,
              use_ppc_loss=True, ppc_cov_thresh=1., ppc_mean_thresh=2., global_coe=cfg["global_coe"],
              global_proto_per_class=cfg["global_per_class"], add_on_layers_type=cfg.get("add_on", "regular"))
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def report(name, **values):
    """Append measured errors to gpurun_out/tol_report.jsonl (scratch, merged back from the GPU box): the gates in the tests are
    set at <= 3x these measurements."""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "tol_report.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if v is not None else None) for k, v in values.items()}}) + "\n")
    except OSError:
        pass


# gelu'(x) as the fc1 epilogue saves it for backward: 8-bit linear codes (csrc/gemm_common.h gelu8_*)
GELU8_ZERO, GELU8_STEP = 26.0, 1.0 / 196.0        # 0 -> code 26, 1 -> code 222: both exact


def gelu8_decode(q):
    return (q.float() - GELU8_ZERO) * GELU8_STEP


def gelu8_encode(d):
    return torch.clamp(torch.floor(d.float() * 196.0 + GELU8_ZERO + 0.5), 0, 255).to(torch.uint8)
