"""CPU restatement of the host-side pre/post-processing — TEST INFRASTRUCTURE.

Pillow (the reference's own dependency: `image.resize`, `image.crop` in src/inference/any_res.py) IS installed here and
on the GPU box, so the resampling oracle is Pillow itself. torchvision (src/processer/transforms.py) is not installed:
its three transforms on PIL input are restated from the published torchvision 0.15 semantics [ext]:
  Resize((h, w) | int, interpolation)  → PIL `img.resize((w, h), resample)` (int: shorter side → size, longer =
                                          int(size * long / short)); antialiasing is Pillow's
  CenterCrop(s)                        → crop at (int(round((h - s) / 2.0)), int(round((w - s) / 2.0)))
  ToTensor()                           → uint8 HWC → float32 CHW, `.div(255)`
  Normalize(mean, std)                 → `(x - mean) / std` in float32
`install_torchvision_shim()` exposes them as `torchvision.transforms` so that the reference's own get_transform() and
process_anyres_image() can be executed for the golden fixtures (oracle/gen_golden.py).
"""
import numpy as np
import torch
from PIL import Image

_RES = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC}


class InterpolationMode:
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class Resize:
    def __init__(self, size, interpolation=InterpolationMode.BILINEAR):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        w, h = img.size
        if isinstance(self.size, int):
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = self.size, int(self.size * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        else:
            nh, nw = self.size
        return img.resize((nw, nh), _RES[self.interpolation])


class CenterCrop:
    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        w, h = img.size
        top, left = int(round((h - self.size) / 2.0)), int(round((w - self.size) / 2.0))
        return img.crop((left, top, left + self.size, top + self.size))


class Lambda:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, img):
        return self.fn(img)


class ToTensor:
    def __call__(self, img):
        a = torch.from_numpy(np.array(img, dtype=np.uint8, copy=True))
        if a.ndim == 2:
            a = a[:, :, None]
        return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return t.clone().sub_(mean).div_(std)


def install_torchvision_shim(force=False):
    import importlib.machinery
    import sys
    import types
    if "torchvision" in sys.modules and not force and getattr(sys.modules["torchvision"], "_seedx_real_shim", False):
        return
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    for m, n in ((tv, "torchvision"), (tvt, "torchvision.transforms")):
        m.__spec__ = importlib.machinery.ModuleSpec(n, None)
    for k in ("Compose", "Resize", "CenterCrop", "Lambda", "ToTensor", "Normalize", "InterpolationMode"):
        setattr(tvt, k, globals()[k])
    tv.transforms, tv._seedx_real_shim = tvt, True
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt


# ---- direct restatements used by the tests on machines without /root/reference ----------------------------------------
CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def clip_transform(image_size=448):
    """get_transform('clip', keep_ratio=False, image_size) of src/processer/transforms.py:5-20."""
    return Compose([Resize((image_size, image_size)), ToTensor(), Normalize(CLIP_MEAN, CLIP_STD)])


def postprocess_pil(image):
    """VaeImageProcessor.postprocess(output_type='pil') [ext]: [B,3,H,W] in [-1,1] → list of PIL images."""
    arr = (image / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).float().numpy()
    return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]


def marker_mask(input_ids, boi, eoi, bop, eop):
    """eval_img2text_seed_x_i.py:153-160."""
    ids_cmp_mask = torch.zeros_like(input_ids, dtype=torch.bool)
    boi_indices = torch.where(torch.logical_or(input_ids == boi, input_ids == bop))[0].tolist()
    eoi_indices = torch.where(torch.logical_or(input_ids == eoi, input_ids == eop))[0].tolist()
    for b, e in zip(boi_indices, eoi_indices):
        ids_cmp_mask[b + 1:e] = True
    return ids_cmp_mask


def resample_u8_fixed_point(img, size, kk_w, bounds_w, kk_h, bounds_h):
    """numpy emulation of csrc/preproc.hip's two integer passes with caller-supplied coefficient tables: lets the CPU
    suite check the product's host-side table builder (image_ops.pil_coeffs) against Pillow without a GPU."""
    a = np.asarray(img, dtype=np.int64)
    H, W, C = a.shape
    ow, oh = size
    half = 1 << 21

    def one_pass(src, kk, bounds, n_out):                      # resample along axis 1
        out = np.empty((src.shape[0], n_out, src.shape[2]), dtype=np.uint8)
        for o in range(n_out):
            lo, cnt = int(bounds[o, 0]), int(bounds[o, 1])
            acc = half + np.tensordot(src[:, lo:lo + cnt, :], kk[o, :cnt].astype(np.int64), axes=([1], [0]))
            out[:, o, :] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
        return out
    if ow != W:
        a = one_pass(a, kk_w, bounds_w, ow).astype(np.int64)
    if oh != H:
        a = one_pass(a.transpose(1, 0, 2), kk_h, bounds_h, oh).transpose(1, 0, 2).astype(np.int64)
    return a.astype(np.uint8)


# ---- any-res tiling restated with Pillow (src/inference/any_res.py) --------------------------------------------------
def _select_best_resolution(original_size, possible_resolutions):          # any_res.py:9-37
    ow, oh = original_size
    best, max_eff, min_waste = None, 0, float('inf')
    for w, h in possible_resolutions:
        scale = min(w / ow, h / oh)
        eff = min(int(ow * scale) * int(oh * scale), ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (w, h)
    return best


def _select_best_resolution_v2(original_size, possible_resolutions):       # any_res.py:40-72
    ow, oh = original_size
    oar, oarea = oh / ow, ow * oh
    best, min_ard, min_ar = None, float('inf'), float('inf')
    for w, h in possible_resolutions:
        ar, area = h / w, w * h
        ard = max(ar, oar) / min(ar, oar)
        arr = max(area, oarea) / min(area, oarea)
        if ard < min_ard or (ard == min_ard and arr < min_ar):
            min_ard, min_ar, best = ard, arr, (w, h)
    return best


def process_anyres_image(image, image_transform, grid_pinpoints, base):     # any_res.py:158-201
    w1, h1 = _select_best_resolution(image.size, grid_pinpoints)
    w2, h2 = _select_best_resolution_v2(image.size, grid_pinpoints)
    bw, bh = (w2, h2) if w1 * h1 > w2 * h2 else (w1, h1)
    padded = image.resize((bw, bh))                                         # Pillow default resample: BICUBIC
    patches = [padded.crop((j, i, j + base, i + base)) for i in range(0, bh, base) for j in range(0, bw, base)]
    patches.append(image.resize((base, base)))
    out = torch.stack([image_transform(p) for p in patches], dim=0)
    gx, gy = bw // base, bh // base
    x_index = (torch.arange(gx).repeat(gy, 1) + 0.5) / gx
    y_index = (torch.arange(gy).unsqueeze(1).repeat(1, gx) + 0.5) / gy
    pos = torch.cat([torch.stack([x_index, y_index], dim=-1).flatten(0, 1), torch.tensor([[0.5, 0.5]])], dim=0)
    return out, pos
