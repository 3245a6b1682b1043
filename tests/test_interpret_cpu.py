"""Visualisation / interpretability post-processing (main_visualize.py, eval_interpretability.py), CPU side:
known-answer and property tests of the restated helpers."""
import os

import numpy as np
import pytest
import torch

from protopformer_amd import interpret as I


def test_expand_to_grid_matches_reference_scatter():
    g = torch.Generator().manual_seed(0)
    B, P, k, n = 3, 5, 9, 16
    attn = torch.rand(B, n, generator=g)
    acts = torch.rand(B, P, 3, 3, generator=g)
    out = I.expand_to_grid(acts, attn, k)
    # main_visualize.py:343-350 verbatim semantics
    idx = torch.topk(attn, k=k, dim=-1)[1].sort(dim=-1)[0][:, None, :].repeat(1, P, 1)
    ref = torch.zeros(B, P, n).scatter_(2, idx, acts.flatten(start_dim=2)).reshape(B, P, 4, 4)
    assert torch.equal(out, ref)
    assert int((out != 0).sum()) == B * P * k


def test_resize_cubic_properties():
    # constants stay constant, a ramp stays monotone and within the a = -0.75 kernel's ripple of the line (only a = -0.5 is exactly
    # linear-preserving), shape, and the interpolation weights are the Keys kernel's
    assert np.allclose(I.resize_cubic(np.full((14, 14), 3.5), 224), 3.5)
    ramp = np.tile(np.arange(14, dtype=np.float64), (14, 1))
    up = I.resize_cubic(ramp, 224)
    assert up.shape == (224, 224)
    x = (np.arange(224) + 0.5) * (14 / 224) - 0.5
    interior = (x > 1.5) & (x < 11.5)
    assert np.allclose(up[100, interior], x[interior], atol=0.06) and (np.diff(up[100, interior]) > 0).all()
    w = I._cubic_weights(np.array([0.0, 0.5]))
    assert np.allclose(w[0], [0, 1, 0, 0]) and np.allclose(w[1], [-0.09375, 0.59375, 0.59375, -0.09375])     # a = -0.75
    # identity at equal size
    a = np.random.default_rng(0).random((7, 7))
    assert np.allclose(I.resize_cubic(a, 7), a, atol=1e-6)


def test_colormap_jet_and_crop_and_rect():
    jet = I.colormap_jet(np.array([0, 64, 128, 191, 255], dtype=np.uint8))
    assert jet.shape == (5, 3) and jet.dtype == np.uint8
    assert tuple(jet[0]) == (128, 0, 0) and tuple(jet[-1]) == (0, 0, 128)          # BGR: dark blue ... dark red
    assert jet[2][1] == 255                                                         # green plateau in the middle
    m = np.zeros((20, 30)); m[5:11, 10:20] = 1.0                  # 10 % of the map: the 95th percentile is inside the block
    assert I.find_high_activation_crop(m, 95) == (5, 11, 10, 20)
    img = np.zeros((20, 30, 3), dtype=np.uint8)
    r = I.draw_rect(img, (10, 5), (14, 8), (0, 255, 255), thickness=1)
    assert (r[5, 10:15] == (0, 255, 255)).all() and (r[6, 11:14] == 0).all() and img.sum() == 0
    d = I.get_discard_img(np.full((32, 32, 3), 9, dtype=np.uint8), [0, 3], fea_size=2, patch_size=16, replace_color=[0, 0, 0])
    assert d[:16, :16].sum() == 0 and d[16:, 16:].sum() == 0 and (d[:16, 16:] == 9).all()


def test_gaussian_params_known_answer():
    """Uniform weights on the 14x14 grid: mean (6.5, 6.5), covariance diag(16.25 * 196 / 195) -- the PPC estimator."""
    mean, cov = I.get_gaussian_params(np.ones((14, 14)))
    assert np.allclose(mean, [6.5, 6.5]) and np.allclose(cov, np.diag([16.25 * 196 / 195] * 2))
    pos = np.zeros((1, 1, 2)) + mean
    assert np.isclose(I.multivariate_gaussian(pos, mean, cov)[0, 0], 1 / (2 * np.pi * np.sqrt(np.linalg.det(cov))))
    assert np.isclose(I.proto_acts_from_distances(np.array([0.0]))[0], np.log(1 / 1e-4))


def test_consistency_tables_and_cub_parts(tmp_path):
    root = tmp_path
    os.makedirs(root / "parts")
    (root / "images.txt").write_text("1 001.A/a.jpg\n2 001.A/b.jpg\n")
    (root / "bounding_boxes.txt").write_text("1 10.0 20.0 100.0 50.0\n2 1.0 2.0 3.0 4.0\n")
    (root / "parts" / "parts.txt").write_text("1 back\n2 beak\n")
    (root / "parts" / "part_locs.txt").write_text("1 1 30.0 40.0 1\n1 2 0.0 0.0 0\n2 2 5.5 6.5 1\n")
    p = I.CubParts(str(root))
    assert p.id_to_path[1] == ("001.A", "a.jpg") and p.id_to_bbox[1] == (10, 20, 110, 70)
    assert p.id_to_part_loc[1] == [[1, 30, 40]] and p.id_to_part_loc[2] == [[2, 5, 6]] and p.part_names["2"] == "beak"
    # a prototype whose peak sits on part 0 in 4 of 5 images (>= 0.8) is consistent; one that wanders is not
    acts = np.zeros((2, 14, 14)); acts[0, 3, 3] = 1.0; acts[1, 10, 10] = 1.0
    t = I.prototype_part_table(acts, [(0, 56, 56)], img_size=224)
    assert t[0, 0] == 1 and t[1, 0] == 0
    tables = [t, t, t, t, np.zeros_like(t)]
    masks = [np.eye(15)[0]] * 5
    effect, mx = I.consistency_from_tables(tables, masks, 0.8)
    assert effect == [1, 0] and mx[0] == pytest.approx(0.8)


def test_consistency_pipeline_equals_the_reference_run():
    """tests/golden/interp_consistency.npz holds what /root/reference/eval_interpretability.py:152-290 computed (with tools/local_parts.py's
    tables) from mini_trees.interp_inputs() on the miniature CUB tree (tests/golden/make_golden_data.py): grid scatter, per-(class,
    prototype) consistency flags, best part fractions and the score must be reproduced; CubParts must parse the same tables."""
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import mini_trees as M
    z = np.load(os.path.join(here, "golden", "interp_consistency.npz"))
    root = tempfile.mkdtemp(prefix="ppf_mini_")
    meta = M.build_cub(root)
    parts = I.CubParts(meta)
    assert np.array_equal(np.array([[i, *parts.id_to_bbox[i]] for i in sorted(parts.id_to_bbox)]), z["parts_bbox"])
    assert np.array_equal(np.array([[i, *p] for i in sorted(parts.id_to_part_loc) for p in parts.id_to_part_loc[i]]), z["parts_locs"])
    attn, acts, targets, ids = M.interp_inputs(k=int(z["k"]), ppc=int(z["ppc"]))
    assert np.array_equal(attn, z["attn"]) and np.array_equal(acts, z["acts"])          # the generator is deterministic
    sizes = {int(i): M.cub_size(int(i)) for i in ids}
    score, effects, max_parts, grid = I.consistency_from_outputs(attn, acts, targets, ids, parts, sizes, int(z["k"]), int(z["img_size"]),
                                                                 num_classes=int(targets.max()) + 1)
    assert np.array_equal(grid, z["grid_acts"])
    assert effects == z["class_proto_effect"].tolist()
    assert np.allclose(max_parts, z["class_max_part"], atol=1e-12)
    assert score == pytest.approx(float(z["score"]), abs=1e-12) and 0.3 < score < 0.8     # a non-trivial mix of consistent / inconsistent
