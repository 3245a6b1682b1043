"""GPU side of SURVEY 8(f)3 / 8(f)4: the one-pass input finisher kernel (uint8 HWC -> normalised fp32 NCHW + RandomErasing) against
numpy, the DeviceLoader feeding the train / eval loop from a miniature CUB tree, and the visualisation / consistency consumers
driven by the HIP model's eval outputs."""
import os
import random
import types

import numpy as np
import pytest
import torch

from helpers import build_micro, micro
from test_data_cpu import _jpeg

pytestmark = pytest.mark.gpu


def test_image_finish_kernel_matches_numpy_and_erases():
    from protopformer_amd.data import IMAGENET_DEFAULT_MEAN as MEAN, IMAGENET_DEFAULT_STD as STD, GpuFinisher
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, (6, 224, 224, 3), dtype=np.uint8)
    ref = (u8.astype(np.float32) / 255.0 - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)
    ref = np.transpose(ref, (0, 3, 1, 2))
    out = GpuFinisher(re_prob=0.0)(torch.from_numpy(u8).cuda()).cpu().numpy()
    assert out.shape == (6, 3, 224, 224) and np.allclose(out, ref, rtol=1e-6, atol=1e-6)
    # RandomErasing 'pixel': inside the rectangle N(0,1) noise, outside untouched; fresh noise per call
    fin = GpuFinisher(re_prob=1.0, seed=11, rng=random.Random(2))
    a = fin(torch.from_numpy(u8).cuda()).cpu().numpy()
    fin2 = GpuFinisher(re_prob=1.0, seed=11, rng=random.Random(2))
    b = fin2(torch.from_numpy(u8).cuda()).cpu().numpy()
    assert np.array_equal(a, b)                                       # same seed / step / rectangles -> same batch
    from protopformer_amd.data import random_erasing_rects
    rects = random_erasing_rects(6, 224, 224, 1.0, rng=random.Random(2))
    n_in = 0
    vals = []
    for i, (y, x, h, w) in enumerate(rects):
        assert h > 0
        mask = np.zeros((224, 224), bool); mask[y:y + h, x:x + w] = True
        assert np.allclose(a[i][:, ~mask], ref[i][:, ~mask], rtol=1e-6, atol=1e-6)
        assert not np.allclose(a[i][:, mask], ref[i][:, mask])
        vals.append(a[i][:, mask].reshape(-1)); n_in += int(mask.sum())
    v = np.concatenate(vals)
    assert abs(v.mean()) < 0.02 and abs(v.std() - 1.0) < 0.02 and np.isfinite(v).all()
    c = fin(torch.from_numpy(u8).cuda()).cpu().numpy()                # next step: new rectangles / new noise
    assert not np.array_equal(a, c)


def test_device_loader_feeds_train_and_eval(tmp_path):
    from protopformer_amd import data as D
    from protopformer_amd.engine import FlatAdamW, evaluate, train_one_epoch
    from protopformer_amd.protopformer import CrossEntropyLoss
    meta = tmp_path / "CUB_200_2011"
    os.makedirs(meta / "images")
    rows = []
    for i in range(1, 17):
        cls = (i - 1) % 10 + 1
        fp = f"{cls:03d}.B/{i:04d}.jpg"
        _jpeg(str(meta / "images" / fp), 90 + i, 70 + i, i)
        rows.append((i, fp, cls, 1 if i <= 12 else 0))
    (meta / "images.txt").write_text("".join(f"{i} {fp}\n" for i, fp, _, _ in rows))
    (meta / "image_class_labels.txt").write_text("".join(f"{i} {c}\n" for i, _, c, _ in rows))
    (meta / "train_test_split.txt").write_text("".join(f"{i} {t}\n" for i, _, _, t in rows))
    sd, cfg, z = micro("micro_deit.npz")                              # 64x64 inputs, 10 classes
    args = types.SimpleNamespace(input_size=64, aa="rand-m9-mstd0.5-inc1", train_interpolation="bicubic", data_set="CUB2011U",
                                 data_path=str(tmp_path), batch_size=4, num_workers=0, reprob=0.25)
    train, val, nb = D.build_loaders(args, torch.device("cuda"))
    assert nb == 200 and len(train) == 3 and len(val) == 1
    x, y = next(iter(train))
    assert x.is_cuda and x.dtype == torch.float32 and x.shape == (4, 3, 64, 64) and y.dtype == torch.int64 and y.is_cuda
    m = build_micro(cfg, sd)
    opt = FlatAdamW(m, weight_decay=0.05)
    stats = train_one_epoch(m, CrossEntropyLoss(), train, opt, torch.device("cuda"), epoch=20, log_every=1, logger=lambda s: None)
    assert np.isfinite(stats["loss"])
    acc = evaluate(val, m, torch.device("cuda"))
    assert 0.0 <= acc["acc1"] <= 100.0 and np.isfinite(acc["loss"])


def test_visualisation_and_consistency_on_hip_eval_outputs(tmp_path):
    from protopformer_amd import interpret as I
    sd, cfg, z = micro("micro_deit.npz")
    m = build_micro(cfg, sd).eval()
    img, label = torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["label"])
    out = I.collect_eval_outputs(m, [(img, label)])
    # the scatter back onto the 4x4 patch grid agrees with the reference fixture's distances
    acts = torch.from_numpy(I.proto_acts_from_distances(out["distances"], m.epsilon))
    grid = I.expand_to_grid(acts, torch.from_numpy(out["token_attn"]), cfg["reserve_k"])
    ref_acts = torch.from_numpy(I.proto_acts_from_distances(z["eval/distances"], m.epsilon))
    ref_grid = I.expand_to_grid(ref_acts, torch.from_numpy(z["eval/cls_token_attn"]), cfg["reserve_k"])
    assert grid.shape == (4, 20, 4, 4)
    assert torch.equal(grid != 0, ref_grid != 0)                      # same reserved cells
    far = torch.from_numpy(z["eval/distances"]).reshape(4, 20, -1) >= 0.05
    assert float((acts.reshape(4, 20, -1)[far] - ref_acts.reshape(4, 20, -1)[far]).abs().max()) < 4e-3 * float(ref_acts.abs().max())
    view = (np.random.default_rng(0).integers(0, 256, (4, 64, 64, 3))).astype(np.uint8)
    files = I.visualize_category(m, [(img, label)], view, str(tmp_path), category_id=3, proto_per_category=2, input_size=64, use_gauss=True)
    assert len(files) == 2 * (1 + 2 * 2) and all(os.path.getsize(f) > 0 for f in files)      # label 3 occurs twice in the fixture
    # consistency score plumbing: two test images of class 3 with one visible part placed on the arg-max of prototype 0's map
    ids = torch.tensor([1, 2, 3, 4])
    _, pa = m.push_forward(img)
    g = I.expand_to_grid(pa[:, 6:8].float().cpu(), torch.from_numpy(out["token_attn"]), cfg["reserve_k"]).numpy()
    parts = types.SimpleNamespace(id_to_part_loc={})
    for j in (0, 3):
        up = I.resize_cubic(g[j, 0], 64)
        ys, xs = np.where(up == up.max())
        parts.id_to_part_loc[int(ids[j])] = [[1, int(xs[0]), int(ys[0])]]
    sizes = {int(i): (64, 64) for i in ids}
    score = I.consistency_score(m, [(img, label, ids)], parts, sizes, num_classes=10, half_size=10)
    assert 0.0 < score <= 1.0
