"""Deterministic miniature CUB-200-2011 / Stanford Cars / Stanford Dogs trees (a few tiny JPEGs + index files in the datasets'
real on-disk formats).  Test infrastructure: make_golden_data.py runs the REFERENCE's dataset classes and interpretability loop on
these trees and commits what they return (tests/golden/data_index.json, interp_consistency.npz); tests/test_data_cpu.py and
tests/test_interpret_cpu.py rebuild the same trees and hold protopformer_amd.data / .interpret to those tables.

The index files are deliberately awkward where the real ones are regular: images.txt is NOT sorted by id, one image has no class
label, Cars annotations carry the devkit's bounding-box fields, Dogs annotations hold one or two boxes."""
import os

import numpy as np
from PIL import Image


def _jpeg(path, w, h, seed):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path, quality=95)


# (img_id, class (1-based), is_training_img): 4 classes x 6 images, every other image is a test image; images.txt lists them in the
# (fixed) shuffled order below, not sorted by id
_ORDER = [7, 1, 20, 2, 12, 3, 24, 4, 5, 17, 6, 8, 9, 22, 10, 11, 13, 14, 19, 15, 16, 18, 21, 23, 25]
CUB_ROWS = [(i, min((i - 1) // 6 + 1, 4), i % 2) for i in _ORDER]
CUB_NO_LABEL = {25}          # present in images.txt and the split file, absent from image_class_labels.txt


def cub_path(i, c):
    return f"{c:03d}.Bird_{c}/Bird_{c}_{i:04d}.jpg"


def cub_size(i):
    return 96 + 4 * i, 80 + 2 * i          # (width, height)


def build_cub(root):
    """root/CUB_200_2011/{images.txt, image_class_labels.txt, train_test_split.txt, bounding_boxes.txt, parts/, images/}."""
    meta = os.path.join(root, "CUB_200_2011")
    os.makedirs(os.path.join(meta, "images"), exist_ok=True)
    os.makedirs(os.path.join(meta, "parts"), exist_ok=True)
    for i, c, _ in CUB_ROWS:
        w, h = cub_size(i)
        _jpeg(os.path.join(meta, "images", cub_path(i, c)), w, h, i)
    with open(os.path.join(meta, "images.txt"), "w") as f:
        f.writelines(f"{i} {cub_path(i, c)}\n" for i, c, _ in CUB_ROWS)
    with open(os.path.join(meta, "image_class_labels.txt"), "w") as f:
        f.writelines(f"{i} {c}\n" for i, c, _ in sorted(CUB_ROWS) if i not in CUB_NO_LABEL)
    with open(os.path.join(meta, "train_test_split.txt"), "w") as f:
        f.writelines(f"{i} {t}\n" for i, _, t in sorted(CUB_ROWS))
    with open(os.path.join(meta, "bounding_boxes.txt"), "w") as f:
        f.writelines(f"{i} {10.0 + i} {5.0 + i} {60.0 + 2 * i} {40.0 + i}\n" for i, _, _ in sorted(CUB_ROWS))
    with open(os.path.join(meta, "parts", "parts.txt"), "w") as f:
        f.writelines(f"{p} part name {p}\n" for p in range(1, 16))
    # part locations: every class has its own relative layout of the 15 parts (+- 2 px of jitter per image, ~75 % visible), fractional
    # coordinates as in the real file -- so that prototypes which fire at a fixed place of a class's images ARE part-consistent
    rng = np.random.default_rng(2011)
    with open(os.path.join(meta, "parts", "part_locs.txt"), "w") as f:
        for i, c, _ in sorted(CUB_ROWS):
            w, h = cub_size(i)
            rel = np.random.default_rng(1000 + c).random((15, 2))
            for p in range(1, 16):
                vis = int(rng.random() < 0.75)
                x = min(max(rel[p - 1, 0] * (w - 1) + rng.uniform(-2, 2), 0.0), w - 1.0) if vis else 0.0
                y = min(max(rel[p - 1, 1] * (h - 1) + rng.uniform(-2, 2), 0.0), h - 1.0) if vis else 0.0
                f.write(f"{i} {p} {x:.1f} {y:.1f} {vis}\n")
    return meta


CARS_TRAIN = [("00001.jpg", 3), ("00002.jpg", 1), ("00003.jpg", 2), ("00004.jpg", 3)]
CARS_TEST = [("00001.jpg", 2), ("00002.jpg", 2), ("00003.jpg", 1)]
CARS_CLASSES = ["AM General Hummer SUV 2000", "Acura RL Sedan 2012", "Acura TL Sedan 2012"]


def build_cars(root):
    """root/stanford_cars/{devkit/cars_train_annos.mat, devkit/cars_meta.mat, cars_test_annos_withlabels.mat, cars_train/, cars_test/}."""
    import scipy.io as sio
    base = os.path.join(root, "stanford_cars")
    os.makedirs(os.path.join(base, "devkit"), exist_ok=True)
    dt = [("bbox_x1", "O"), ("bbox_y1", "O"), ("bbox_x2", "O"), ("bbox_y2", "O"), ("class", "O"), ("fname", "O")]
    for rows, folder, path in ((CARS_TRAIN, "cars_train", os.path.join(base, "devkit", "cars_train_annos.mat")),
                               (CARS_TEST, "cars_test", os.path.join(base, "cars_test_annos_withlabels.mat"))):
        ann = np.zeros((1, len(rows)), dtype=dt)
        for k, (fn, c) in enumerate(rows):
            _jpeg(os.path.join(base, folder, fn), 64 + 4 * k, 48 + 2 * k, 100 + k + len(folder))
            ann[0, k] = (np.array([[3 + k]], dtype=np.uint8), np.array([[4 + k]], dtype=np.uint8), np.array([[50 + k]], dtype=np.uint8),
                         np.array([[40 + k]], dtype=np.uint8), np.array([[c]], dtype=np.uint8), np.array([fn]))
        sio.savemat(path, {"annotations": ann})
    names = np.empty((1, len(CARS_CLASSES)), dtype=object)
    for k, n in enumerate(CARS_CLASSES):
        names[0, k] = np.array([n])
    sio.savemat(os.path.join(base, "devkit", "cars_meta.mat"), {"class_names": names})
    return base


DOGS_TRAIN = [("n02085620-Chihuahua/n02085620_10", 1), ("n02085620-Chihuahua/n02085620_11", 1), ("n02085936-Maltese_dog/n02085936_7", 3),
              ("n02085782-Japanese_spaniel/n02085782_2", 2)]
DOGS_TEST = [("n02085936-Maltese_dog/n02085936_9", 3), ("n02085620-Chihuahua/n02085620_12", 1)]
DOGS_TWO_BOXES = {"n02085620-Chihuahua/n02085620_11", "n02085936-Maltese_dog/n02085936_9"}


def build_dogs(root):
    """root/{Images/, Annotation/, train_list.mat, test_list.mat} (file_list / annotation_list cell arrays, labels column)."""
    import scipy.io as sio
    os.makedirs(root, exist_ok=True)
    for rows, fname in ((DOGS_TRAIN, "train_list.mat"), (DOGS_TEST, "test_list.mat")):
        cell_a = np.empty((len(rows), 1), dtype=object)
        cell_f = np.empty((len(rows), 1), dtype=object)
        for k, (n, _) in enumerate(rows):
            _jpeg(os.path.join(root, "Images", n + ".jpg"), 70 + 3 * k, 60 + 2 * k, 200 + k + len(fname))
            os.makedirs(os.path.join(root, "Annotation", os.path.dirname(n)), exist_ok=True)
            boxes = [(5 + k, 6, 40 + k, 30)] + ([(20, 10 + k, 60, 55)] if n in DOGS_TWO_BOXES else [])
            objs = "".join(f"<object><name>dog</name><bndbox><xmin>{a}</xmin><ymin>{b}</ymin><xmax>{c}</xmax><ymax>{d}</ymax></bndbox></object>"
                           for a, b, c, d in boxes)
            with open(os.path.join(root, "Annotation", n), "w") as f:
                f.write(f"<annotation><folder>{os.path.dirname(n)}</folder>{objs}</annotation>")
            cell_a[k, 0] = n
            cell_f[k, 0] = n + ".jpg"
        sio.savemat(os.path.join(root, fname), {"file_list": cell_f, "annotation_list": cell_a,
                                                "labels": np.array([[c] for _, c in rows], dtype=np.uint8)})
    return root


def interp_inputs(seed=5, k=9, ppc=10, grid=14):
    """Synthetic push_forward outputs for the CUB test split of build_cub (the labelled is_training_img == 0 images, in dataset
    order): rollout scores (B, grid*grid), the class's own prototype activations on the reserved tokens (B, ppc, s, s), targets, ids."""
    rng = np.random.default_rng(seed)
    rows = [(i, c) for i, c, t in CUB_ROWS if t == 0 and i not in CUB_NO_LABEL]
    B, s = len(rows), int(round(k ** 0.5))
    # rollout scores: one well-separated pattern per class (the reserved cells of a class's images coincide) + tiny per-image noise
    base = {c: np.random.default_rng(50 + c).permutation(grid * grid).astype(np.float32) for c in {c for _, c in rows}}
    attn = np.stack([base[c] + 0.01 * rng.random(grid * grid).astype(np.float32) for _, c in rows])
    acts = rng.random((B, ppc, s, s)).astype(np.float32) * 5.0
    # every third prototype peaks at a class-specific reserved token; the others wander
    for j, (i, c) in enumerate(rows):
        for p in range(0, ppc, 3):
            acts[j, p, (c + p) % s, (2 * c + p) % s] = 9.0
    targets = np.array([c - 1 for _, c in rows], dtype=np.int64)
    ids = np.array([i for i, _ in rows], dtype=np.int64)
    return attn, acts, targets, ids
