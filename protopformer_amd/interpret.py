"""Consumers of the eval-branch outputs: prototype activation visualisation (reference main_visualize.py:273-474)
and the part-consistency interpretability score on CUB (reference eval_interpretability.py:100-336, tools/local_parts.py).

The model side runs on the HIP kernels (`PPNet.forward` in eval mode / `push_forward`); everything here is the reference's
CPU-side post-processing restated on numpy / PIL.  OpenCV is not installed in this image, so the three cv2 primitives the
reference uses are written out from their documented definitions (unpinned against cv2 itself, tested for their properties):
`resize_cubic` = cv2.resize(..., interpolation=INTER_CUBIC) (Keys kernel a = -0.75, half-pixel centres, replicated border),
`colormap_jet` = cv2.applyColorMap(..., COLORMAP_JET) (BGR), `draw_rect` = cv2.rectangle outline."""
import os

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------ activation maps
def reserved_indices(token_attn, k):
    """topk(k) + ascending sort of the rollout scores (main_visualize.py:345-346): the grid cells the reserved tokens came from."""
    return torch.topk(token_attn.reshape(token_attn.shape[0], -1), k=k, dim=-1)[1].sort(dim=-1)[0]


def expand_to_grid(acts, token_attn, k):
    """(B, P, s, s) activations of the k = s*s reserved tokens -> (B, P, g, g) on the full patch grid, zeros elsewhere
    (main_visualize.py:343-350, eval_interpretability.py:156-167)."""
    B, P = acts.shape[:2]
    n = token_attn.reshape(B, -1).shape[-1]
    g = int(round(n ** 0.5))
    idx = reserved_indices(token_attn, k)[:, None, :].expand(B, P, k)
    out = torch.zeros(B, P, n, dtype=acts.dtype, device=acts.device)
    out.scatter_(2, idx, acts.reshape(B, P, -1))
    return out.reshape(B, P, g, g)


def proto_acts_from_distances(distances, epsilon=1e-4):
    """main_visualize.py:326: log((d + 1) / (d + eps))."""
    return np.log((distances + 1) / (distances + epsilon))


@torch.no_grad()
def collect_eval_outputs(ppnet, loader, category_id=None, min_count=20):
    """main_visualize.py:306-327: run the eval branch over `loader` (until more than `min_count` samples of `category_id` were
    seen, if given).  Returns dict(token_attn (B, Np), distances (B, P, s, s), labels, pred)."""
    ppnet.eval()
    attn, dist, labels, pred = [], [], [], []
    for x, y, *_ in loader:
        logits, aux = ppnet(x.cuda() if not x.is_cuda else x)
        attn.append(aux[0].float().cpu().numpy()); dist.append(aux[1].float().cpu().numpy())
        labels.append(np.asarray(y.cpu())); pred.append(logits.argmax(1).cpu().numpy())
        if category_id is not None and int((np.concatenate(labels) == category_id).sum()) > min_count:
            break
    return dict(token_attn=np.concatenate(attn), distances=np.concatenate(dist), labels=np.concatenate(labels), pred=np.concatenate(pred))


# ------------------------------------------------------------------------------------------------ cv2 primitives, restated
def _cubic_weights(f, a=-0.75):
    w0 = ((a * (f + 1) - 5 * a) * (f + 1) + 8 * a) * (f + 1) - 4 * a
    w1 = ((a + 2) * f - (a + 3)) * f * f + 1
    w2 = ((a + 2) * (1 - f) - (a + 3)) * (1 - f) * (1 - f) + 1
    return np.stack([w0, w1, w2, 1.0 - w0 - w1 - w2], axis=-1)


def _resize_axis(a, size, axis):
    n = a.shape[axis]
    s = (np.arange(size) + 0.5) * (n / size) - 0.5
    i0 = np.floor(s).astype(np.int64)
    w = _cubic_weights(s - i0)                                         # (size, 4)
    idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n - 1)   # replicated border
    taken = np.take(a, idx.reshape(-1), axis=axis)
    shape = list(a.shape); shape[axis:axis + 1] = [size, 4]
    taken = taken.reshape(shape)
    wshape = [1] * len(shape); wshape[axis] = size; wshape[axis + 1] = 4
    return (taken * w.reshape(wshape)).sum(axis=axis + 1)


def resize_cubic(a, size):
    """cv2.resize(a, (size, size), interpolation=cv2.INTER_CUBIC) for a 2-D float array."""
    a = np.asarray(a, dtype=np.float64)
    return _resize_axis(_resize_axis(a, size, 0), size, 1).astype(np.float32)


def colormap_jet(gray_u8):
    """cv2.applyColorMap(gray, cv2.COLORMAP_JET): uint8 (...,) -> uint8 (..., 3) in B, G, R order."""
    x = np.asarray(gray_u8, dtype=np.float64) / 255.0
    r = np.clip(1.5 - np.abs(4 * x - 3), 0, 1)
    g = np.clip(1.5 - np.abs(4 * x - 2), 0, 1)
    b = np.clip(1.5 - np.abs(4 * x - 1), 0, 1)
    return np.stack([b, g, r], axis=-1).__mul__(255).round().astype(np.uint8)


def draw_rect(img, start_xy, end_xy, color, thickness=2):
    """cv2.rectangle outline on a copy of an (H, W, 3) image; start / end are (x, y) corners."""
    out = np.array(img, copy=True)
    (x0, y0), (x1, y1) = start_xy, end_xy
    H, W = out.shape[:2]
    x0, x1, y0, y1 = max(0, min(x0, x1)), min(W - 1, max(x0, x1)), max(0, min(y0, y1)), min(H - 1, max(y0, y1))
    t = thickness
    out[y0:y0 + t, x0:x1 + 1] = color; out[max(y0, y1 - t + 1):y1 + 1, x0:x1 + 1] = color
    out[y0:y1 + 1, x0:x0 + t] = color; out[y0:y1 + 1, max(x0, x1 - t + 1):x1 + 1] = color
    return out


# ------------------------------------------------------------------------------------------------ main_visualize.py helpers
def get_discard_img(view_img, discard_indices, fea_size, patch_size, replace_color):
    """main_visualize.py:36-41: paint the patches of the discarded tokens."""
    res = np.copy(view_img)
    for d in discard_indices:
        h, w = int(d) // fea_size, int(d) % fea_size
        res[h * patch_size:(h + 1) * patch_size, w * patch_size:(w + 1) * patch_size] = replace_color
    return res


def find_high_activation_crop(activation_map, percentile=95):
    """main_visualize.py:44-66: bounding box (y0, y1, x0, x1) of the entries >= the given percentile."""
    mask = activation_map >= np.percentile(activation_map, percentile)
    rows, cols = np.nonzero(mask.any(axis=1))[0], np.nonzero(mask.any(axis=0))[0]
    if rows.size == 0:
        return 0, 1, 0, 1
    return int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1


def get_gaussian_params(proto_act):
    """main_visualize.py:69-84: weighted mean (2,) and covariance (2, 2) of the grid coordinates, weights rescaled to sum n^2
    (the same estimator as the PPC loss, protopformer.py:249-257)."""
    n = proto_act.shape[-1]
    pts = np.array([[x, y] for x in range(n) for y in range(n)], dtype=np.float64).T        # (2, n*n)
    w = proto_act.reshape(1, -1).astype(np.float64)
    w = w / w.sum(axis=-1) * (n * n)
    mean = np.mean(pts * w, axis=-1)
    cut = pts - mean[:, None]
    return mean, np.dot(cut * w, cut.T) / (n * n - 1)


def multivariate_gaussian(pos, mu, sigma):
    """main_visualize.py:87-98."""
    k = mu.shape[0]
    fac = np.einsum("...k,kl,...l->...", pos - mu, np.linalg.inv(sigma), pos - mu)
    return np.exp(-fac / 2) / np.sqrt((2 * np.pi) ** k * np.linalg.det(sigma))


def activation_overlay(view_img_bgr, proto_act, input_size):
    """main_visualize.py:381-391, 398: upsample one (g, g) activation map to the image, min-max normalise, JET heat map, 0.7 / 0.3
    blend; returns (overlay uint8 BGR, top-5 % box (y0, y1, x0, x1), arg-max grid cell)."""
    up = resize_cubic(proto_act, input_size)
    up = up - up.min()
    up = up / max(float(up.max()), 1e-12)
    heat = colormap_jet(np.uint8(255 * up))
    cell = tuple(int(t[0]) for t in np.where(proto_act == proto_act.max()))
    return (view_img_bgr * 0.7 + heat * 0.3).astype(np.uint8), find_high_activation_crop(up), cell


def visualize_category(ppnet, loader, view_images_bgr, out_dir, category_id, proto_per_category=10, input_size=224, patch_size=16,
                       use_gauss=False, max_images=None):
    """main_visualize.py:273-474 for one category: for every test image of the class and each of its prototypes, write the
    activation overlay, the top-5 % box image and the discarded-token mask (JPEG, via PIL).  `view_images_bgr`: uint8 (B, H, W, 3)
    un-normalised images aligned with `loader`'s order.  Returns the list of written files."""
    from PIL import Image
    out = collect_eval_outputs(ppnet, loader, category_id)
    k = ppnet.reserve_token_nums[-1]
    sel = np.nonzero(out["labels"] == category_id)[0]
    if max_images:
        sel = sel[:max_images]
    acts = torch.from_numpy(proto_acts_from_distances(out["distances"][sel], ppnet.epsilon))
    attn = torch.from_numpy(out["token_attn"][sel])
    grid = expand_to_grid(acts, attn, k).numpy()
    n_patches = attn.shape[-1]
    fea = int(round(n_patches ** 0.5))
    discard = torch.topk(attn, k=n_patches - k, dim=-1, largest=False)[1].numpy()
    cat_dir = os.path.join(out_dir, f"category_{category_id}")
    written = []
    for j, b in enumerate(sel):
        img_dir = os.path.join(cat_dir, f"img_{j}")
        os.makedirs(img_dir, exist_ok=True)
        img = view_images_bgr[b]
        path = os.path.join(img_dir, f"catch_img_reserve{k}_mask.jpg")
        Image.fromarray(get_discard_img(img, discard[j], fea, patch_size, [0, 0, 0])[:, :, ::-1]).save(path); written.append(path)
        for pi in range(proto_per_category):
            act = grid[j, category_id * proto_per_category + pi]
            over, (y0, y1, x0, x1), _ = activation_overlay(img, act, input_size)
            path = os.path.join(img_dir, f"proto{pi}_reserve{k}.jpg")
            Image.fromarray(over[:, :, ::-1]).save(path); written.append(path)
            path = os.path.join(img_dir, f"proto{pi}_reserve{k}_bnd.jpg")
            Image.fromarray(draw_rect(img, (x0, y0), (x1, y1), (0, 255, 255))[:, :, ::-1]).save(path); written.append(path)
            if use_gauss:
                mean, cov = get_gaussian_params(np.maximum(act, 0) + 1e-12)
                np.save(os.path.join(img_dir, f"gaussian_{pi}.npy"), np.concatenate([mean, cov.reshape(-1)]))
    return written


# ------------------------------------------------------------------------------------------------ eval_interpretability.py
class CubParts:
    """tools/local_parts.py: image paths, bounding boxes and the visible part locations of CUB-200-2011."""

    def __init__(self, data_root):
        def lines(*p):
            with open(os.path.join(data_root, *p)) as f:
                return [l.rstrip("\n") for l in f if l.strip()]
        self.id_to_path = {}
        for l in lines("images.txt"):
            i, p = l.split(" ", 1)
            self.id_to_path[int(i)] = tuple(p.split("/", 1))
        self.id_to_bbox = {}
        for l in lines("bounding_boxes.txt"):
            c = l.split(" ")
            x, y, w, h = (int(v.split(".")[0]) for v in c[1:5])
            self.id_to_bbox[int(c[0])] = (x, y, x + w, y + h)
        self.part_names = {l.split(" ", 1)[0]: l.split(" ", 1)[1] for l in lines("parts", "parts.txt")}
        self.id_to_part_loc = {}
        for l in lines("parts", "part_locs.txt"):
            c = l.split(" ")
            i, pid, x, y, vis = int(c[0]), int(c[1]), int(float(c[2])), int(float(c[3])), int(c[4])
            self.id_to_part_loc.setdefault(i, [])
            if vis == 1:
                self.id_to_part_loc[i].append([pid, x, y])


def in_bbox(loc, bbox):
    return bbox[0] <= loc[0] <= bbox[1] and bbox[2] <= loc[1] <= bbox[3]


def prototype_part_table(acts_grid, part_labels, img_size, half_size=36, n_parts=15):
    """eval_interpretability.py:206-226 for one image: (n_proto, n_parts) 0/1 table -- does the 2*half_size box around the arg-max of
    the up-sampled activation contain the part?  part_labels: [(part_id0, x, y)] in resized-image pixels."""
    table = np.zeros((acts_grid.shape[0], n_parts))
    for pi in range(acts_grid.shape[0]):
        up = resize_cubic(acts_grid[pi], img_size)
        ys, xs = np.where(up == up.max())
        my, mx = int(ys[0]), int(xs[0])
        box = (max(0, my - half_size), min(img_size, my + half_size), max(0, mx - half_size), min(img_size, mx + half_size))
        for pid, x, y in part_labels:
            if in_bbox((y, x), box):
                table[pi, pid] = 1
    return table


def consistency_from_tables(tables, masks, part_thresh=0.8):
    """eval_interpretability.py:262-286 for one class: tables (n_img, n_proto, n_parts), masks (n_img, n_parts) of annotated parts.
    A prototype is consistent if some part falls inside its box in >= part_thresh of the images where that part is visible."""
    tables = np.transpose(np.asarray(tables), (1, 0, 2))
    masks = np.asarray(masks)
    denom = masks.sum(axis=0)
    denom = np.where(denom == 0, 1, denom)
    effect, max_part = [], []
    for t in tables:
        assert ((1.0 - masks) * t).sum() == 0
        frac = t.sum(axis=0) / denom
        effect.append(int((frac >= part_thresh).any()))
        max_part.append(float(frac.max()))
    return effect, max_part


@torch.no_grad()
def consistency_score(ppnet, loader, parts, image_sizes, num_classes=200, part_thresh=0.8, half_size=36, n_parts=15):
    """eval_interpretability.py:136-290: push_forward over the test set, the class's own prototypes expanded to the patch grid, the
    part table per image, and the fraction of prototypes that are part-consistent.  loader yields (x, targets, img_ids);
    image_sizes: {img_id: (width, height)} of the original files (the reference reads them with cv2.imread)."""
    ppnet.eval()
    ppc, k, img_size = ppnet.num_prototypes_per_class, ppnet.reserve_token_nums[0], ppnet.img_size
    attn, acts, targets, ids = [], [], [], []
    for x, t, i in loader:
        ta, pa = ppnet.push_forward(x.cuda() if not x.is_cuda else x)
        t = torch.as_tensor(t)
        cols = (t.to(pa.device) * ppc)[:, None] + torch.arange(ppc, device=pa.device)[None, :]
        acts.append(torch.gather(pa, 1, cols[:, :, None, None].expand(-1, -1, pa.shape[-2], pa.shape[-1])).cpu())
        attn.append(ta.cpu()); targets.append(t.cpu()); ids.append(torch.as_tensor(i).cpu())
    return consistency_from_outputs(torch.cat(attn), torch.cat(acts), torch.cat(targets).numpy(), torch.cat(ids).numpy(), parts, image_sizes, k,
                                    img_size, num_classes, part_thresh, half_size, n_parts)[0]


def consistency_from_outputs(attn, acts, targets, ids, parts, image_sizes, k, img_size, num_classes=200, part_thresh=0.8, half_size=36, n_parts=15):
    """eval_interpretability.py:152-290 on collected push_forward outputs: attn (B, Np) rollout scores, acts (B, ppc, s, s) the class's
    own prototype activations on the k = s*s reserved tokens, targets / ids (B,).  Returns (score, effect per (class, prototype),
    best part fraction per (class, prototype), activations on the patch grid).  Classes without a test image are skipped (the
    reference's loop assumes every class has one)."""
    attn, acts = torch.as_tensor(attn), torch.as_tensor(acts)
    targets, ids = np.asarray(targets), np.asarray(ids)
    grid = expand_to_grid(acts.float(), attn.float(), k).numpy() if k != attn.reshape(attn.shape[0], -1).shape[-1] else acts.numpy()
    effects, max_parts = [], []
    for c in range(num_classes):
        sel = np.nonzero(targets == c)[0]
        if sel.size == 0:
            continue
        tables, masks = [], []
        for j in sel:
            w, h = image_sizes[int(ids[j])]
            mask, labels = np.zeros(n_parts), []
            for pid, x, y in parts.id_to_part_loc.get(int(ids[j]), []):
                mask[pid - 1] = 1
                labels.append((pid - 1, int(img_size * (x / w)), int(img_size * (y / h))))
            tables.append(prototype_part_table(grid[j], labels, img_size, half_size, n_parts)); masks.append(mask)
        e, m = consistency_from_tables(tables, masks, part_thresh)
        effects.extend(e); max_parts.extend(m)
    return (float(np.mean(effects)) if effects else 0.0), effects, max_parts, grid
