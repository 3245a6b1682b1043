"""Input pipeline on synthetic miniature datasets laid out like CUB-200-2011 / Stanford Cars / Stanford Dogs:
index parsing, splits, label shifts (tools/datasets.py:402-474, 477-589, 662-907) and the PIL-space transforms
(tools/datasets.py:280-336) that run in the CPU workers."""
import os
import random
import types

import numpy as np
import pytest
from PIL import Image


def _jpeg(path, w, h, seed):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path, quality=95)


@pytest.fixture(scope="module")
def cub_root(tmp_path_factory):
    root = tmp_path_factory.mktemp("cub")
    meta = root / "CUB_200_2011"
    os.makedirs(meta / "images")
    rows = []
    for i in range(1, 13):
        cls = (i - 1) // 3 + 1
        fp = f"{cls:03d}.Bird_{cls}/Bird_{cls}_{i:04d}.jpg"
        _jpeg(str(meta / "images" / fp), 96 + 8 * i, 80 + 4 * i, i)
        rows.append((i, fp, cls, 1 if i % 3 else 0))
    (meta / "images.txt").write_text("".join(f"{i} {fp}\n" for i, fp, _, _ in rows))
    (meta / "image_class_labels.txt").write_text("".join(f"{i} {c}\n" for i, _, c, _ in rows))
    (meta / "train_test_split.txt").write_text("".join(f"{i} {t}\n" for i, _, _, t in rows))
    return str(root)


def test_cub2011_index_split_and_labels(cub_root):
    from protopformer_amd.data import Cub2011
    tr, te = Cub2011(cub_root, train=True), Cub2011(cub_root, train=False, return_id=True)
    assert len(tr) == 8 and len(te) == 4
    img, target = tr[0]
    assert img.mode == "RGB" and target == 0                          # targets start at 1 in the files -> 0-based
    assert sorted({t for _, t in (tr[i] for i in range(len(tr)))}) == [0, 1, 2, 3]
    img, target, img_id = te[3]
    assert img_id == 12 and target == 3
    with pytest.raises(RuntimeError, match="not found"):
        Cub2011(os.path.join(cub_root, "nope"))


def test_stanford_cars_and_dogs_indexes(tmp_path):
    import scipy.io as sio
    from protopformer_amd.data import Dogs, StanfordCars
    base = tmp_path / "stanford_cars"
    os.makedirs(base / "devkit")
    ann = np.zeros(3, dtype=[("fname", "O"), ("class", "i4")])
    for i in range(3):
        _jpeg(str(base / "cars_train" / f"{i:05d}.jpg"), 64, 48, i)
        ann[i] = (f"{i:05d}.jpg", i + 1)
    sio.savemat(str(base / "devkit" / "cars_train_annos.mat"), {"annotations": ann})
    sio.savemat(str(base / "devkit" / "cars_meta.mat"), {"class_names": np.array(["a", "b", "c"], dtype=object)})
    cars = StanfordCars(str(tmp_path), split="train")
    assert len(cars) == 3 and cars[2][1] == 2 and cars.classes == ["a", "b", "c"]
    with pytest.raises(RuntimeError):
        StanfordCars(str(tmp_path), split="test")
    droot = tmp_path / "stanford_dogs"
    names = ["n01-Chihuahua/n01_1", "n01-Chihuahua/n01_2", "n02-Maltese/n02_1"]
    for k, n in enumerate(names):
        _jpeg(str(droot / "Images" / (n + ".jpg")), 60, 50, k)
        os.makedirs(droot / "Annotation" / os.path.dirname(n), exist_ok=True)
        (droot / "Annotation" / n).write_text("<annotation><object><bndbox><xmin>5</xmin><ymin>6</ymin><xmax>40</xmax><ymax>30</ymax></bndbox></object></annotation>")
    cell = np.empty((len(names), 1), dtype=object)                       # a MATLAB cell array of strings, as the real lists
    for k, n in enumerate(names):
        cell[k, 0] = n
    lst = {"annotation_list": cell, "labels": np.array([[1], [1], [2]])}
    sio.savemat(str(droot / "train_list.mat"), lst)
    dogs = Dogs(str(droot), train=True)
    assert len(dogs) == 3 and dogs[2][1] == 1 and dogs.stats() == {0: 2, 1: 1}
    crop = Dogs(str(droot), train=True, cropped=True)
    assert crop[0][0].size == (35, 24)


def test_transforms_geometry_and_randaugment(cub_root):
    from protopformer_amd import data as D
    args = types.SimpleNamespace(input_size=224, aa="rand-m9-mstd0.5-inc1", train_interpolation="bicubic", data_set="CUB2011U", data_path=cub_root)
    random.seed(3)
    ds, nb = D.build_dataset(True, args)
    assert nb == 200
    for i in range(len(ds)):
        a, t = ds[i]
        assert a.dtype == np.uint8 and a.shape == (224, 224, 3) and a.flags["C_CONTIGUOUS"]
    ev, _ = D.build_dataset(False, args)
    a, _ = ev[0]
    assert a.shape == (224, 224, 3)
    # eval geometry: shorter side -> 256 (bicubic), centre crop 224
    img = Image.new("RGB", (400, 300), (10, 20, 30))
    r = D.Resize(256, "bicubic")(img)
    assert r.size == (341, 256) and D.CenterCrop(224)(r).size == (224, 224)
    # RandomResizedCrop parameters stay inside the image and respect the scale / ratio bounds
    rrc = D.RandomResizedCrop(224, rng=random.Random(0))
    for _ in range(200):
        top, left, ch, cw = rrc.get_params(400, 300)
        assert 0 <= top and top + ch <= 300 and 0 <= left and left + cw <= 400
        assert 0.08 * 0.9 <= ch * cw / (400 * 300) <= 1.0 and 0.74 <= cw / ch <= 1.34 or (ch, cw) == (300, 400)
    # every RandAugment op runs, keeps the size, and magnitude 0 of the enhancement ops is the identity
    ra = D.RandAugment("rand-m9-mstd0.5-inc1", rng=random.Random(1))
    src = Image.fromarray(np.random.default_rng(0).integers(0, 256, (224, 224, 3), dtype=np.uint8))
    for op in D.RandAugment.OPS:
        out = ra._apply(op, src, 9.0)
        assert out.size == src.size and out.mode == "RGB", op
    for op in ("ColorIncreasing", "BrightnessIncreasing", "Rotate", "ShearX", "TranslateYRel", "SolarizeAdd"):
        assert np.array_equal(np.asarray(ra._apply(op, src, 0.0)), np.asarray(src)), op
    assert (ra.m, ra.mstd, ra.n, ra.inc) == (9.0, 0.5, 2, True)
    flips = sum(np.array_equal(np.asarray(D.RandomHorizontalFlip(0.5, rng=random.Random(k))(src)), np.asarray(src)[:, ::-1]) for k in range(200))
    assert 70 < flips < 130


def test_random_erasing_rects_distribution():
    from protopformer_amd.data import random_erasing_rects
    r = random_erasing_rects(4000, 224, 224, prob=0.25, rng=random.Random(5))
    hit = r[:, 2] > 0
    assert 0.21 < hit.mean() < 0.29
    area = (r[hit, 2] * r[hit, 3]) / (224 * 224)
    assert area.min() >= 0.015 and area.max() <= 0.34
    assert (r[hit, 0] + r[hit, 2] <= 224).all() and (r[hit, 1] + r[hit, 3] <= 224).all()
    asp = r[hit, 2] / r[hit, 3]
    assert asp.min() > 0.25 and asp.max() < 4.0


def test_preprocess_round_trip():
    import torch
    from protopformer_amd.data import preprocess_input_function, undo_preprocess_input_function
    x = torch.rand(2, 3, 8, 8)
    y = preprocess_input_function(x)
    assert torch.allclose(y[:, 1], (x[:, 1] - 0.456) / 0.224)
    assert torch.allclose(undo_preprocess_input_function(y), x, atol=1e-6)


# ---------------------------------------------------------------------------------------------- pinned to the reference
def _golden_index():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_index.json")))


@pytest.fixture(scope="module")
def mini_root(tmp_path_factory):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import mini_trees as M
    root = str(tmp_path_factory.mktemp("mini"))
    M.build_cub(root); M.build_cars(root); M.build_dogs(os.path.join(root, "dogs"))
    return root


def test_index_classes_equal_the_reference_tables(mini_root):
    """tests/golden/data_index.json holds what /root/reference/tools/datasets.py's Cub2011 / StanfordCars / Dogs returned on the same
    miniature trees (tests/golden/make_golden_data.py): sample order, relative paths, 0-based labels, image sizes after the dataset's
    own cropping, class names and per-class counts must all agree."""
    from protopformer_amd.data import Cub2011, Dogs, StanfordCars
    gold = _golden_index()
    for train in (True, False):
        ds = Cub2011(mini_root, train=train, return_id=True)
        ref = gold[f"cub_{'train' if train else 'test'}"]
        assert len(ds) == len(ref)
        for j, r in enumerate(ref):
            img, target, img_id = ds[j]
            assert (ds.data[j][1], target, img_id, list(img.size)) == (r["path"], r["target"], r["img_id"], r["size"]), (j, r)
    for split in ("train", "test"):
        ds = StanfordCars(mini_root, split=split)
        ref = gold[f"cars_{split}"]
        assert len(ds) == len(ref) and ds.classes == gold["cars_classes"]
        for j, r in enumerate(ref):
            img, target = ds[j]
            assert (os.path.relpath(ds._samples[j][0], mini_root), target, list(img.size)) == (r["path"], r["target"], r["size"]), (j, r)
    droot = os.path.join(mini_root, "dogs")
    for train in (True, False):
        for cropped in (False, True):
            ds = Dogs(droot, train=train, cropped=cropped)
            key = f"dogs_{'train' if train else 'test'}"
            ref = gold[key + ("_cropped" if cropped else "")]
            assert len(ds) == len(ref)
            for j, r in enumerate(ref):
                img, target = ds[j]
                assert (ds._flat_breed_images[j][0], target, list(img.size)) == (r["path"], r["target"], r["size"]), (j, r)
                if cropped:
                    assert list(ds._flat_breed_annotations[j][1]) == r["box"]
            if not cropped:
                assert {str(k): v for k, v in ds.stats().items()} == gold[key + "_stats"]
