"""Real-data input pipeline of the train / eval loop (reference: tools/datasets.py:167-335 build_dataset /
build_transform, :402-474 Cub2011, :477-589 StanfordCars, :662-907 Dogs, tools/preprocess.py:1-33).

MI355X-first split of the work:
  * CPU worker processes do what only a CPU library can do here -- JPEG decode (PIL) and the PIL-space geometric / photometric
    augmentations (RandomResizedCrop, flip, RandAugment) -- and hand over **uint8 HWC frames** (150 KB per 224x224 image: a
    quarter of the fp32 tensor the reference's ToTensor produces, so the host-to-device copy of a 256-image batch is 38 MB);
  * the GPU does the rest in ONE kernel pass (`ppf_image_finish_u8`, csrc/elementwise.hip): uint8 -> fp32, HWC -> NCHW,
    mean/std normalisation and timm's RandomErasing ('pixel' mode: N(0,1) noise from Philox) written straight into the batch
    buffer the train step reads.
Dataset index parsing follows the reference's classes line by line in behaviour (same files, same label shifts, same splits).

Third-party boundary: the train transform the reference builds is timm 0.5.4's `create_transform(..., auto_augment=
'rand-m9-mstd0.5-inc1', interpolation='bicubic', re_prob=0.25, re_mode='pixel')` (tools/datasets.py:280-309); timm is not
vendored and not installed here, so RandAugment / RandomResizedCrop / RandomErasing below are written from timm's public API
("parity unpinned" at that boundary, as for PatchEmbed / Mlp in the oracle) and tested for their invariants.
"""
import math
import os
import random
import xml.etree.ElementTree

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageOps

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def default_loader(path):
    with open(path, "rb") as f:
        return Image.open(f).convert("RGB")


# ------------------------------------------------------------------------------------------------ datasets
class Cub2011:
    """CUB-200-2011 (tools/datasets.py:402-474): root/CUB_200_2011/{images.txt, image_class_labels.txt, train_test_split.txt,
    images/}; targets shifted to 0..199; `train` selects is_training_img == 1 / 0.  return_id=True also yields the image id
    (eval_interpretability.py:140 iterates (data, targets, img_ids))."""
    base_folder = "CUB_200_2011/images"

    def __init__(self, root, train=True, transform=None, loader=default_loader, download=False, return_id=False):
        self.root = os.path.expanduser(root)
        self.transform, self.loader, self.train, self.return_id = transform, loader, train, return_id
        if download:
            raise RuntimeError("no network access: place CUB_200_2011 under the data path")
        meta = os.path.join(self.root, "CUB_200_2011")
        try:
            paths = dict(self._pairs(os.path.join(meta, "images.txt"), str))
            labels = dict(self._pairs(os.path.join(meta, "image_class_labels.txt"), int))
            split = dict(self._pairs(os.path.join(meta, "train_test_split.txt"), int))
        except OSError as e:
            raise RuntimeError("Dataset not found or corrupted. (no download possible here)") from e
        want = 1 if train else 0
        # pandas inner merges on img_id (tools/datasets.py:427-434): rows stay in the order of images.txt, ids missing from either of
        # the other two files drop out
        self.data = [(i, paths[i], labels[i]) for i in paths if split.get(i) == want and i in labels]
        for _, fp, _ in self.data:
            if not os.path.isfile(os.path.join(self.root, self.base_folder, fp)):
                raise RuntimeError("Dataset not found or corrupted: missing " + fp)

    @staticmethod
    def _pairs(path, conv):
        with open(path) as f:
            for line in f:
                a, b = line.rstrip("\n").split(" ", 1)
                yield int(a), conv(b)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        img_id, fp, target = self.data[idx]
        img = self.loader(os.path.join(self.root, self.base_folder, fp))
        if self.transform is not None:
            img = self.transform(img)
        return (img, target - 1, img_id) if self.return_id else (img, target - 1)


class StanfordCars:
    """Stanford Cars (tools/datasets.py:477-589): root/stanford_cars/{devkit/cars_train_annos.mat, devkit/cars_meta.mat,
    cars_test_annos_withlabels.mat, cars_train/, cars_test/}; class ids shifted to 0..195."""

    def __init__(self, root, split="train", transform=None, target_transform=None, download=False):
        import scipy.io as sio
        if split not in ("train", "test"):
            raise ValueError("split must be 'train' or 'test'")
        base = os.path.join(root, "stanford_cars")
        devkit = os.path.join(base, "devkit")
        annos = os.path.join(devkit, "cars_train_annos.mat") if split == "train" else os.path.join(base, "cars_test_annos_withlabels.mat")
        images = os.path.join(base, "cars_train" if split == "train" else "cars_test")
        if not (os.path.exists(annos) and os.path.isdir(images)):
            raise RuntimeError("Dataset not found. (no download possible here)")
        self.transform, self.target_transform = transform, target_transform
        ann = np.atleast_1d(sio.loadmat(annos, squeeze_me=True)["annotations"])
        self._samples = [(os.path.join(images, str(a["fname"])), int(a["class"]) - 1) for a in ann]
        self.classes = [str(c) for c in np.atleast_1d(sio.loadmat(os.path.join(devkit, "cars_meta.mat"), squeeze_me=True)["class_names"]).tolist()]
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}

    def __len__(self):
        return len(self._samples)

    def __getitem__(self, idx):
        path, target = self._samples[idx]
        img = Image.open(path).convert("RGB")
        if self.transform is not None:
            img = self.transform(img)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return img, target


class Dogs:
    """Stanford Dogs (tools/datasets.py:662-907): root/{Images/, Annotation/, train_list.mat, test_list.mat}; `cropped` cuts every
    annotated bounding box out as its own sample."""

    def __init__(self, root, train=True, cropped=False, transform=None, target_transform=None, download=False):
        import scipy.io as sio
        self.root, self.train, self.cropped, self.transform, self.target_transform = root, train, cropped, transform, target_transform
        lst = sio.loadmat(os.path.join(root, "train_list.mat" if train else "test_list.mat"))
        split = [(str(item[0][0]), int(lab[0]) - 1) for item, lab in zip(lst["annotation_list"], lst["labels"])]
        self.images_folder, self.annotations_folder = os.path.join(root, "Images"), os.path.join(root, "Annotation")
        self._breeds = sorted(d for d in os.listdir(self.images_folder) if os.path.isdir(os.path.join(self.images_folder, d)))
        if cropped:
            self._flat_breed_annotations = [(a, box, idx) for a, idx in split for box in self.get_boxes(os.path.join(self.annotations_folder, a))]
            self._flat_breed_images = [(a + ".jpg", idx) for a, _, idx in self._flat_breed_annotations]
        else:
            self._flat_breed_images = [(a + ".jpg", idx) for a, idx in split]

    @staticmethod
    def get_boxes(path):
        e = xml.etree.ElementTree.parse(path).getroot()
        return [[int(o.find("bndbox").find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")] for o in e.iter("object")]

    def __len__(self):
        return len(self._flat_breed_images)

    def __getitem__(self, index):
        name, target = self._flat_breed_images[index]
        img = Image.open(os.path.join(self.images_folder, name)).convert("RGB")
        if self.cropped:
            img = img.crop(self._flat_breed_annotations[index][1])
        if self.transform:
            img = self.transform(img)
        if self.target_transform:
            target = self.target_transform(target)
        return img, target

    def stats(self):
        counts = {}
        for _, t in self._flat_breed_images:
            counts[t] = counts.get(t, 0) + 1
        return counts


# ------------------------------------------------------------------------------------------------ PIL-space transforms (CPU workers)
_INTERP = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}


class Resize:
    """torchvision Resize: int -> shorter side to `size` keeping the aspect ratio; (h, w) -> exact."""

    def __init__(self, size, interpolation="bicubic"):
        self.size, self.interp = size, _INTERP[interpolation] if isinstance(interpolation, str) else interpolation

    def __call__(self, img):
        if isinstance(self.size, int):
            w, h = img.size
            if (w <= h and w == self.size) or (h <= w and h == self.size):
                return img
            if w < h:
                return img.resize((self.size, int(self.size * h / w)), self.interp)
            return img.resize((int(self.size * w / h), self.size), self.interp)
        return img.resize((self.size[1], self.size[0]), self.interp)


class CenterCrop:
    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        w, h = img.size
        left, top = int(round((w - self.size) / 2.0)), int(round((h - self.size) / 2.0))
        return img.crop((left, top, left + self.size, top + self.size))


class RandomResizedCrop:
    """timm RandomResizedCropAndInterpolation / torchvision RandomResizedCrop: scale (0.08, 1), ratio (3/4, 4/3), 10 attempts."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation="bicubic", rng=random):
        self.size, self.scale, self.ratio, self.interp, self.rng = size, scale, ratio, _INTERP[interpolation], rng

    def get_params(self, w, h):
        area = w * h
        for _ in range(10):
            target = self.rng.uniform(*self.scale) * area
            ar = math.exp(self.rng.uniform(math.log(self.ratio[0]), math.log(self.ratio[1])))
            cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                return self.rng.randint(0, h - ch), self.rng.randint(0, w - cw), ch, cw
        in_ratio = w / h
        if in_ratio < self.ratio[0]:
            cw, ch = w, int(round(w / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            ch, cw = h, int(round(h * self.ratio[1]))
        else:
            cw, ch = w, h
        return (h - ch) // 2, (w - cw) // 2, ch, cw

    def __call__(self, img):
        top, left, ch, cw = self.get_params(*img.size)
        return img.resize((self.size, self.size), self.interp, box=(left, top, left + cw, top + ch))


class RandomHorizontalFlip:
    def __init__(self, p=0.5, rng=random):
        self.p, self.rng = p, rng

    def __call__(self, img):
        return img.transpose(Image.FLIP_LEFT_RIGHT) if self.rng.random() < self.p else img


class RandAugment:
    """timm rand_augment_transform('rand-m9-mstd0.5-inc1'): 2 ops per image drawn uniformly from the 'increasing' set, each applied
    with probability 0.5 at magnitude ~ N(9, 0.5) clipped to [0, 10]."""
    OPS = ("AutoContrast", "Equalize", "Invert", "Rotate", "PosterizeIncreasing", "SolarizeIncreasing", "SolarizeAdd", "ColorIncreasing",
           "ContrastIncreasing", "BrightnessIncreasing", "SharpnessIncreasing", "ShearX", "ShearY", "TranslateXRel", "TranslateYRel")

    def __init__(self, config="rand-m9-mstd0.5-inc1", img_size=224, mean=IMAGENET_DEFAULT_MEAN, rng=random):
        parts = config.split("-")
        assert parts[0] == "rand"
        self.m, self.mstd, self.n, self.inc = 10.0, 0.0, 2, False
        for c in parts[1:]:
            if c.startswith("mstd"):
                self.mstd = float(c[4:])
            elif c.startswith("inc"):
                self.inc = bool(int(c[3:]))
            elif c.startswith("m"):
                self.m = float(c[1:])
            elif c.startswith("n"):
                self.n = int(c[1:])
        if not self.inc:
            raise NotImplementedError("only the 'increasing' op set of the reference's default policy is implemented")
        self.fill = tuple(int(round(255 * x)) for x in mean)
        self.translate_pct = 0.45
        self.rng = rng

    def _neg(self, v):
        return -v if self.rng.random() > 0.5 else v

    def _apply(self, name, img, mag):
        lv = mag / 10.0
        resample = self.rng.choice((Image.BILINEAR, Image.BICUBIC))
        if name == "AutoContrast":
            return ImageOps.autocontrast(img)
        if name == "Equalize":
            return ImageOps.equalize(img)
        if name == "Invert":
            return ImageOps.invert(img)
        if name == "Rotate":
            return img.rotate(self._neg(lv * 30.0), resample=resample, fillcolor=self.fill)
        if name == "PosterizeIncreasing":
            return ImageOps.posterize(img, max(1, 4 - int(lv * 4))) if 4 - int(lv * 4) < 8 else img
        if name == "SolarizeIncreasing":
            return ImageOps.solarize(img, 256 - int(lv * 256))
        if name == "SolarizeAdd":
            add, thresh = int(lv * 110), 128
            lut = [min(255, i + add) if i < thresh else i for i in range(256)]
            return img.point(lut * 3)
        if name in ("ColorIncreasing", "ContrastIncreasing", "BrightnessIncreasing", "SharpnessIncreasing"):
            f = max(0.1, 1.0 + self._neg(lv * 0.9))
            enh = {"Color": ImageEnhance.Color, "Contrast": ImageEnhance.Contrast, "Brightness": ImageEnhance.Brightness,
                   "Sharpness": ImageEnhance.Sharpness}[name[:-len("Increasing")]]
            return enh(img).enhance(f)
        if name == "ShearX":
            return img.transform(img.size, Image.AFFINE, (1, self._neg(lv * 0.3), 0, 0, 1, 0), resample=resample, fillcolor=self.fill)
        if name == "ShearY":
            return img.transform(img.size, Image.AFFINE, (1, 0, 0, self._neg(lv * 0.3), 1, 0), resample=resample, fillcolor=self.fill)
        if name == "TranslateXRel":
            return img.transform(img.size, Image.AFFINE, (1, 0, self._neg(lv * self.translate_pct) * img.size[0], 0, 1, 0), resample=resample, fillcolor=self.fill)
        if name == "TranslateYRel":
            return img.transform(img.size, Image.AFFINE, (1, 0, 0, 0, 1, self._neg(lv * self.translate_pct) * img.size[1]), resample=resample, fillcolor=self.fill)
        raise KeyError(name)

    def __call__(self, img):
        for name in [self.rng.choice(self.OPS) for _ in range(self.n)]:
            if self.rng.random() > 0.5:
                continue
            mag = self.m if self.mstd <= 0 else self.rng.gauss(self.m, self.mstd)
            img = self._apply(name, img, min(10.0, max(0.0, mag)))
        return img


class ToUint8HWC:
    """The hand-over format to the GPU finisher: a contiguous uint8 [H, W, 3] array (no float conversion on the CPU)."""

    def __call__(self, img):
        a = np.asarray(img, dtype=np.uint8)
        if a.ndim == 2:
            a = np.repeat(a[:, :, None], 3, axis=2)
        return np.ascontiguousarray(a[:, :, :3])


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


def build_transform(is_train, args):
    """tools/datasets.py:280-336 for input_size > 32, up to (not including) ToTensor / Normalize / RandomErasing, which run on the
    GPU (GpuFinisher).  Train: RandomResizedCrop(bicubic) + flip + RandAugment(args.aa); eval: Resize(256/224 * size, bicubic) +
    CenterCrop."""
    size = args.input_size
    if size <= 32:
        raise NotImplementedError("the CIFAR-sized (input_size <= 32) branch is not on the ProtoPFormer path")
    if is_train:
        t = [RandomResizedCrop(size, interpolation=getattr(args, "train_interpolation", "bicubic")), RandomHorizontalFlip(0.5)]
        aa = getattr(args, "aa", "rand-m9-mstd0.5-inc1")
        if aa and aa != "none":
            t.append(RandAugment(aa, img_size=size))
        return Compose(t + [ToUint8HWC()])
    return Compose([Resize(int((256 / 224) * size), "bicubic"), CenterCrop(size), ToUint8HWC()])


def build_view_transform(args, square=False):
    """build_dataset_view / build_dataset_noaug geometry (tools/datasets.py:77-166): Resize(256/224 * size, bicubic) + CenterCrop,
    or a plain (size, size) resize when `square` (the reference's 'adam' model branch and eval_interpretability.py:128-132)."""
    size = args.input_size
    if square:
        return Compose([Resize((size, size), "bilinear"), ToUint8HWC()])
    return Compose([Resize(int((256 / 224) * size), "bicubic"), CenterCrop(size), ToUint8HWC()])


def build_dataset(is_train, args, transform=None):
    """tools/datasets.py:167-277 for the fine-grained sets ProtoPFormer is trained on.  Returns (dataset, nb_classes)."""
    transform = transform if transform is not None else build_transform(is_train, args)
    if args.data_set == "CUB2011U":
        return Cub2011(args.data_path, train=is_train, transform=transform), 200
    if args.data_set == "Car":
        return StanfordCars(args.data_path, split="train" if is_train else "test", transform=transform), 196
    if args.data_set == "Dogs":
        return Dogs(root=os.path.join(args.data_path, "stanford_dogs"), train=is_train, cropped=False, transform=transform), 120
    raise NotImplementedError(f"data_set {args.data_set!r}: only CUB2011U / Car / Dogs (scripts/train_{{cub,car,dog}}.sh) are on the path")


# ------------------------------------------------------------------------------------------------ GPU finisher + loader
def random_erasing_rects(B, H, W, prob=0.25, min_area=0.02, max_area=1 / 3, min_aspect=0.3, rng=random):
    """timm RandomErasing(probability, mode='pixel', max_count=1) rectangle draw per sample: int32 [B, 4] = (y, x, h, w), h = 0: none."""
    rects = np.zeros((B, 4), dtype=np.int32)
    log_r = (math.log(min_aspect), math.log(1 / min_aspect))
    for b in range(B):
        if rng.random() > prob:
            continue
        for _ in range(10):
            target = rng.uniform(min_area, max_area) * H * W
            ar = math.exp(rng.uniform(*log_r))
            h, w = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if w < W and h < H:
                rects[b] = (rng.randint(0, H - h), rng.randint(0, W - w), h, w)
                break
    return rects


class GpuFinisher:
    """uint8 [B,H,W,3] (device) -> fp32 [B,3,H,W] normalised (+ RandomErasing when training) in one HIP kernel pass."""

    def __init__(self, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, re_prob=0.0, seed=None, rng=random):
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)
        self.re_prob, self.rng = float(re_prob), rng
        self.seed = torch.initial_seed() if seed is None else int(seed)
        self.step = 0

    def __call__(self, frames_u8, out=None):
        from . import _lib
        if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8:
            raise RuntimeError("GpuFinisher needs a uint8 CUDA tensor [B, H, W, 3] (no CPU fallback path)")
        B, H, W, C = frames_u8.shape
        assert C == 3 and frames_u8.is_contiguous()
        if out is None:
            out = torch.empty((B, 3, H, W), dtype=torch.float32, device=frames_u8.device)
        rects = state = None
        if self.re_prob > 0:
            rects = torch.from_numpy(random_erasing_rects(B, H, W, self.re_prob, rng=self.rng)).to(frames_u8.device, non_blocking=True)
            state = torch.tensor([self.step], dtype=torch.int64, device=frames_u8.device)
            self.step += 1
        _lib.call("ppf_image_finish_u8", frames_u8, out, B, H, W, self.mean.ctypes.data, self.std.ctypes.data, rects, self.seed & 0xFFFFFFFFFFFFFFFF, state)
        return out


def _collate_u8(batch):
    frames = torch.from_numpy(np.stack([b[0] for b in batch]))
    rest = [torch.as_tensor([b[i] for b in batch]) for i in range(1, len(batch[0]))]
    return (frames, *rest)


class DeviceLoader:
    """DataLoader (CPU decode / augmentation workers, uint8 collate, pinned) + GpuFinisher: yields (fp32 NCHW cuda, labels cuda[, ids])
    -- what engine.train_one_epoch / evaluate iterate over (main.py:302-316)."""

    def __init__(self, dataset, batch_size, device, finisher, shuffle=False, sampler=None, num_workers=0, drop_last=False, pin_memory=True):
        self.loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle if sampler is None else False, sampler=sampler,
                                                  num_workers=num_workers, drop_last=drop_last, pin_memory=pin_memory and torch.cuda.is_available(),
                                                  collate_fn=_collate_u8)
        self.device, self.finisher, self.sampler = device, finisher, sampler

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for frames, target, *rest in self.loader:
            x = self.finisher(frames.to(self.device, non_blocking=True))
            yield (x, target.to(self.device, non_blocking=True).long(), *rest)


def build_loaders(args, device):
    """main.py:281-316: train loader (shuffled / DistributedSampler, drop_last) and validation loader (batch 1.5x, sequential)."""
    import torch.distributed as dist
    ds_train, nb = build_dataset(True, args)
    ds_val, _ = build_dataset(False, args)
    sampler = None
    if dist.is_initialized() and dist.get_world_size() > 1:
        sampler = torch.utils.data.DistributedSampler(ds_train, num_replicas=dist.get_world_size(), rank=dist.get_rank(), shuffle=True)
    train = DeviceLoader(ds_train, args.batch_size, device, GpuFinisher(re_prob=getattr(args, "reprob", 0.25)), shuffle=sampler is None, sampler=sampler,
                         num_workers=getattr(args, "num_workers", 10), drop_last=True)
    val = DeviceLoader(ds_val, int(1.5 * args.batch_size), device, GpuFinisher(re_prob=0.0), num_workers=getattr(args, "num_workers", 10))
    return train, val, nb


# ------------------------------------------------------------------------------------------------ tools/preprocess.py
def preprocess(x, mean, std):
    assert x.size(1) == 3
    m = torch.as_tensor(mean, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    s = torch.as_tensor(std, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return (x - m) / s


def preprocess_input_function(x):
    return preprocess(x, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)


def undo_preprocess(x, mean, std):
    assert x.size(1) == 3
    m = torch.as_tensor(mean, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    s = torch.as_tensor(std, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return x * s + m


def undo_preprocess_input_function(x):
    return undo_preprocess(x, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)
