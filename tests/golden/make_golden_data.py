"""Golden tables for SURVEY 8(f)3 / 8(f)4 from the REFERENCE's own code, run in the build container on the miniature trees of
mini_trees.py:

    python tests/golden/make_golden_data.py

* tests/golden/data_index.json      what /root/reference/tools/datasets.py's Cub2011 / StanfordCars / Dogs return on those trees:
                                    per sample (relative path, label, image size after the dataset's own crop) for every split
* tests/golden/interp_consistency.npz   what /root/reference/eval_interpretability.py:152-290 (grid scatter, part tables, consistency
                                    score) computes from mini_trees.interp_inputs() and tools/local_parts.py's tables

Needs /root/reference, pandas and scipy (installed).  Absent third-party modules are replaced by stand-ins for the NON-arithmetic
calls only: torchvision.datasets (VisionDataset base class, default_loader = PIL open + convert, list_dir, verify_str_arg), timm.data
names, cv2.imread / cvtColor (PIL) and cv2.resize -- whose INTER_CUBIC arithmetic is delegated to torch.nn.functional.interpolate
(bicubic, align_corners=False: the same Keys a = -0.75 kernel with replicated borders), NOT to protopformer_amd.interpret.
eval_interpretability.py is a script (argparse + model + GPU at import), so its analysis section is executed from its source text
with the variables the earlier part of the script would have produced; no reference text is stored, the outputs are numbers."""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import mini_trees as M  # noqa: E402
from PIL import Image  # noqa: E402


def _install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def default_loader(path):
        with open(path, "rb") as f:
            return Image.open(f).convert("RGB")

    class VisionDataset:                      # torchvision.datasets.VisionDataset: stores root / transforms, nothing else used here
        def __init__(self, root=None, transforms=None, transform=None, target_transform=None):
            self.root = os.path.expanduser(root) if isinstance(root, str) else root
            self.transform, self.target_transform = transform, target_transform

    def verify_str_arg(value, arg=None, valid_values=None, custom_msg=None):
        if valid_values is not None and value not in valid_values:
            raise ValueError(f"unknown value {value!r} for {arg}")
        return value

    def list_dir(root, prefix=False):
        return [d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))]

    def no_network(*a, **k):
        raise RuntimeError("no network")

    class ImageFolder(VisionDataset):
        pass

    folder = mod("torchvision.datasets.folder", ImageFolder=ImageFolder, default_loader=default_loader)
    utils = mod("torchvision.datasets.utils", download_url=no_network, extract_archive=no_network, list_dir=list_dir,
                download_and_extract_archive=no_network, verify_str_arg=verify_str_arg)
    datasets = mod("torchvision.datasets", VisionDataset=VisionDataset, folder=folder, utils=utils, ImageFolder=ImageFolder)
    transforms = mod("torchvision.transforms")
    mod("torchvision", datasets=datasets, transforms=transforms)
    mod("timm")
    mod("timm.data", create_transform=None)
    mod("timm.data.constants", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))

    INTER_CUBIC = 2

    def imread(path):
        return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()

    def resize(a, dsize, interpolation=1):
        a = np.asarray(a)
        if a.ndim == 3:                       # the photograph: only its shape is used by the analysis
            return np.asarray(Image.fromarray(a[:, :, ::-1]).resize(dsize, Image.BILINEAR))[:, :, ::-1].copy()
        assert interpolation == INTER_CUBIC
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None]
        return torch.nn.functional.interpolate(t, size=(dsize[1], dsize[0]), mode="bicubic", align_corners=False)[0, 0].numpy()

    mod("cv2", transform=None, imread=imread, resize=resize, INTER_CUBIC=INTER_CUBIC, COLOR_BGR2RGB=4,
        cvtColor=lambda img, code: img[:, :, ::-1].copy())


def dataset_tables(root):
    sys.path.insert(0, REF)
    import tools.datasets as RD               # the reference module itself
    out = {}

    def table(ds, rel_to):
        rows = []
        for j in range(len(ds)):
            img, target = ds[j][0], ds[j][1]
            rows.append(dict(size=list(img.size), target=int(target)))
        return rows

    for train in (True, False):
        ds = RD.Cub2011(root, train=train)
        rows = table(ds, root)
        for r, (_, rec) in zip(rows, ds.data.iterrows()):
            r.update(path=str(rec.filepath), img_id=int(rec.img_id))
        out[f"cub_{'train' if train else 'test'}"] = rows
    for split in ("train", "test"):
        ds = RD.StanfordCars(root, split=split)
        rows = table(ds, root)
        for r, (p, _) in zip(rows, ds._samples):
            r.update(path=os.path.relpath(p, root))
        out[f"cars_{split}"] = rows
        out["cars_classes"] = [str(c) for c in ds.classes]
    droot = os.path.join(root, "dogs")
    for train in (True, False):
        for cropped in (False, True):
            ds = RD.Dogs(droot, train=train, cropped=cropped)
            rows = table(ds, droot)
            for j, (r, (name, _)) in enumerate(zip(rows, ds._flat_breed_images)):
                r.update(path=name)
                if cropped:
                    r.update(box=[int(v) for v in ds._flat_breed_annotations[j][1]])
            out[f"dogs_{'train' if train else 'test'}{'_cropped' if cropped else ''}"] = rows
            if not cropped:
                out[f"dogs_{'train' if train else 'test'}_stats"] = {str(int(k)): int(v) for k, v in ds.stats().items()}
    return out


def interp_tables(root, k=9, ppc=10, img_size=224):
    """Runs eval_interpretability.py's analysis section (its lines from `if args.reserve_token_nums[0] != 196:` to the consistency
    score) on synthetic push_forward outputs."""
    cwd = os.getcwd()
    os.makedirs(os.path.join(root, "datasets"), exist_ok=True)
    if not os.path.exists(os.path.join(root, "datasets", "CUB_200_2011")):
        os.symlink(os.path.join(root, "CUB_200_2011"), os.path.join(root, "datasets", "CUB_200_2011"))
    os.chdir(root)                            # tools/local_parts.py reads the relative path datasets/CUB_200_2011 at import
    try:
        import tools.local_parts as LP
    finally:
        os.chdir(cwd)
    src = open(os.path.join(REF, "eval_interpretability.py")).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.startswith("if args.reserve_token_nums[0] != 196:"))
    stop = next(i for i, l in enumerate(src) if l.startswith("class_proto_effect_score = "))
    attn, acts, targets, ids = M.interp_inputs(k=k, ppc=ppc)
    import cv2
    ns = dict(np=np, torch=torch, os=os, cv2=cv2, tqdm=lambda x, *a, **kw: x, plt=None,
              args=types.SimpleNamespace(reserve_token_nums=[k], data_path=os.path.join(root, "CUB_200_2011"), vis_image=False),
              all_token_attn=torch.from_numpy(attn), all_proto_acts=acts.copy(), all_targets=targets, all_img_ids=ids,
              token_reserve_num=k, num_prototypes_per_class=ppc, num_classes=int(targets.max()) + 1, img_size=img_size, half_size=36,
              part_thresh=0.8, id_to_path=LP.id_to_path, id_to_part_loc=LP.id_to_part_loc, id_to_bbox=LP.id_to_bbox, in_bbox=LP.in_bbox)
    exec(compile("\n".join(src[start:stop + 1]), "eval_interpretability.py[analysis]", "exec"), ns)
    return dict(attn=attn, acts=acts, targets=targets, ids=ids, grid_acts=np.asarray(ns["all_proto_acts"], dtype=np.float32),
                class_proto_effect=np.asarray(ns["class_proto_effect"], dtype=np.int64), class_max_part=np.asarray(ns["class_max_part"], dtype=np.float64),
                class_mean_part=np.asarray(ns["class_mean_part"], dtype=np.int64), score=np.float64(ns["class_proto_effect_score"]),
                k=np.int64(k), ppc=np.int64(ppc), img_size=np.int64(img_size),
                parts_bbox=np.array([[i, *LP.id_to_bbox[i]] for i in sorted(LP.id_to_bbox)], dtype=np.int64),
                parts_locs=np.array([[i, *p] for i in sorted(LP.id_to_part_loc) for p in LP.id_to_part_loc[i]], dtype=np.int64))


def main():
    _install_standins()
    root = tempfile.mkdtemp(prefix="ppf_mini_")
    M.build_cub(root); M.build_cars(root); M.build_dogs(os.path.join(root, "dogs"))
    tables = dataset_tables(root)
    with open(os.path.join(HERE, "data_index.json"), "w") as f:
        json.dump(tables, f, indent=0, sort_keys=True)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in tables.items()})
    it = interp_tables(root)
    np.savez_compressed(os.path.join(HERE, "interp_consistency.npz"), **it)
    print("consistency score", float(it["score"]), "effects", it["class_proto_effect"].tolist(), "max_part", np.round(it["class_max_part"], 3).tolist())


if __name__ == "__main__":
    main()
