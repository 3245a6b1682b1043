"""CPU-side checks of the C-ABI library: it loads without a GPU and exports exactly what include/ppf_hip.h declares;
the Python binding table agrees with the header; missing-GPU use fails loudly (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ppf_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppf_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from protopformer_amd.build import build
    return ctypes.CDLL(build(verbose=False))


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in ppf_hip.h but not exported: {missing}"


def test_binding_table_matches_header(lib):
    from protopformer_amd import _lib
    declared = set(_declared())
    assert set(_lib.SIGS) <= declared
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, spec in _lib.SIGS.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        assert nargs == len(spec), f"{name}: header has {nargs} parameters, binding spec {len(spec)}"


def test_error_channel_without_gpu(lib):
    lib.ppf_last_error.restype = ctypes.c_char_p
    from protopformer_amd import _lib
    want = int(re.search(r"#define\s+PPF_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.ppf_abi_version() == want == _lib.EXPECTED_ABI
    # argument validation happens before any device call: a bad shape is reported through the error channel
    rc = lib.ppf_cast_f32_bf16(None, None, ctypes.c_int64(7), None)
    assert rc == -1 and b"multiple of 8" in lib.ppf_last_error()


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from protopformer_amd.protopformer import construct_PPNet
    m = construct_PPNet("deit_tiny_patch16_224", pretrained=False, prototype_shape=(20, 32, 1, 1), num_classes=10, reserve_layers=[11],
                        reserve_token_nums=[81], use_global=True, use_ppc_loss=True, global_proto_per_class=2, add_on_layers_type="regular")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 224, 224))


def test_state_dict_keys_match_reference_fixture():
    """Drop-in checkpoint compatibility: key set and shapes equal the reference's state dict."""
    from helpers import micro
    from protopformer_amd.deit import MyVisionTransformer
    from protopformer_amd.protopformer import PPNet
    sd, cfg, _ = micro("micro_deit.npz")
    feats = MyVisionTransformer(img_size=64, patch_size=16, embed_dim=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"], drop_path_rate=0.0)
    m = PPNet(features=feats, img_size=64, prototype_shape=[20, 32, 1, 1], proto_layer_rf_info=None, num_classes=10, reserve_layers=[cfg['reserve_layer']],
              reserve_token_nums=[9], use_global=True, use_ppc_loss=True, global_proto_per_class=2, add_on_layers_type="regular")
    mine = m.state_dict()
    assert set(mine) == set(sd)
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    w = m.last_layer.weight
    assert float(w[3, 6:8].min()) == 1.0 and float(w[3, :6].max()) == -0.5
