"""GPU image pre/post-processing around the dense paths — host side of csrc/preproc.hip.

Reference interfaces mirrored (same names, argument meaning and results):
  * ``get_transform(type, keep_ratio, image_size)``                    src/processer/transforms.py:5-86
  * ``process_anyres_image(image, image_transform, grid_pinpoints, base_image_size)``   src/inference/any_res.py:158-201
    (+ ``select_best_resolution`` :9-37, ``select_best_resolution_v2`` :40-72, ``resize_and_pad_image`` :75-114,
    ``divide_to_patches`` :117-136, ``get_anyres_image_grid_shape`` :139-155)
  * marker-mask construction                                           src/inference/eval_img2text_seed_x_i.py:153-160
  * latents → PIL                                                      pipeline_stable_diffusion_xl_t2i_edit.py:986

The reference does this on the host with Pillow + torchvision; here the uint8 image is uploaded once and everything —
Pillow's antialiased bicubic / bilinear resample (bit exact: Pillow's own 22-bit fixed-point scheme), tiling, ToTensor,
Normalize — runs as HIP kernels, producing the ``[n_crops, 3, 448, 448]`` tensor path A consumes directly in HBM.
Only Pillow's coefficient tables (a few KB, double precision, cached per size pair) are computed on the host.
"""
import ast
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import check

BILINEAR, BICUBIC = "bilinear", "bicubic"
_PRECISION_BITS = 32 - 8 - 2
_COEFF_CACHE = {}
_LUT_CACHE = {}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bilinear(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {BILINEAR: (_bilinear, 1.0), BICUBIC: (_bicubic, 2.0)}


def pil_coeffs(in_size, out_size, resample):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (Resample.c [ext]) for a full-axis resize: returns
    (kk int32 [out][ksize], bounds int32 [out][2], ksize). Python floats are the C doubles; int() is the C cast."""
    key = (in_size, out_size, resample)
    if key in _COEFF_CACHE:
        return _COEFF_CACHE[key]
    filt, fsupport = _FILTERS[resample]
    in0, in1 = 0.0, float(in_size)
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        w = []
        for x in range(xmax):
            v = filt((x + xmin - center + 0.5) * ss)
            w.append(v)
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
    _COEFF_CACHE[key] = (kk, bounds, ksize)
    return _COEFF_CACHE[key]


def _dev_tables(in_size, out_size, resample, device):
    key = (in_size, out_size, resample, str(device))
    if key not in _COEFF_CACHE:
        kk, bounds, ksize = pil_coeffs(in_size, out_size, resample)
        _COEFF_CACHE[key] = (torch.from_numpy(kk).to(device), torch.from_numpy(bounds).to(device), ksize, bounds)
    return _COEFF_CACHE[key]


def to_device_u8(image, device):
    """PIL image (converted to RGB) | HxWx3 uint8 ndarray | uint8 tensor → contiguous uint8 [H, W, 3] on `device`."""
    if torch.is_tensor(image):
        t = image
    else:
        if hasattr(image, "convert"):
            image = image.convert("RGB")
        t = torch.from_numpy(np.array(image, dtype=np.uint8, copy=True))
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, "expected an HxWx3 uint8 image"
    return t.to(device).contiguous()


def resize_u8(img, size, resample=BICUBIC):
    """``PIL.Image.resize(size, resample)`` on a device image: img uint8 [H, W, 3] (cuda), size = (width, height) like PIL.
    Bit-exact with Pillow (antialiased, 8-bit intermediate after the horizontal pass). Returns uint8 [height, width, 3]."""
    if not img.is_cuda:
        raise RuntimeError("seedx_amd.image_ops: images must live on the GPU (no CPU fallback)")
    lib = _lib.load()
    Hin, Win, Cc = img.shape
    Wout, Hout = int(size[0]), int(size[1])
    if (Wout, Hout) == (Win, Hin):
        return img.clone()                                           # Pillow returns a copy
    need_h, need_v = Wout != Win, Hout != Hin
    out = torch.empty((Hout, Wout, Cc), dtype=torch.uint8, device=img.device)
    kh = bh = kv = bv = None
    ksh = ksv = y_first = y_rows = 0
    tmp = None
    if need_h:
        kh, bh, ksh, _ = _dev_tables(Win, Wout, resample, img.device)
    if need_v:
        kv, bv, ksv, bnp = _dev_tables(Hin, Hout, resample, img.device)
        if need_h:                                                   # Pillow: only the rows the vertical pass reads
            y_first = int(bnp[0, 0])
            y_rows = int(bnp[-1, 0] + bnp[-1, 1]) - y_first
            tmp = torch.empty((y_rows, Wout, Cc), dtype=torch.uint8, device=img.device)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    check(lib.sx_resample_u8(p(img), Hin, Win, Cc, img.stride(0), p(out), Hout, Wout, p(kh), p(bh), ksh, p(kv), p(bv), ksv,
                             y_first, y_rows, p(tmp), _stream()), "sx_resample_u8")
    return out


def _lut(mean, std, device):
    """lut[c][v] = (float32(v) / 255 - mean[c]) / std[c], evaluated in float32 exactly like ToTensor() + Normalize()."""
    key = (tuple(mean), tuple(std), str(device))
    if key not in _LUT_CACHE:
        v = (np.arange(256, dtype=np.float32) / np.float32(255.0))[None, :]
        m = np.asarray(mean, dtype=np.float32).reshape(-1, 1)
        s = np.asarray(std, dtype=np.float32).reshape(-1, 1)
        lut = ((v - m) / s).astype(np.float32)
        if lut.shape[0] == 1:
            lut = np.repeat(lut, 3, axis=0)
        _LUT_CACHE[key] = torch.from_numpy(np.ascontiguousarray(lut)).to(device)
    return _LUT_CACHE[key]


def crop_to_tensor(img, box, mean, std, out=None):
    """crop(box = (left, upper, right, lower)) → ToTensor → Normalize(mean, std): uint8 [H,W,3] → fp32 [3, h, w]."""
    lib = _lib.load()
    H, W, _ = img.shape
    x0, y0, x1, y1 = box
    if out is None:
        out = torch.empty((3, y1 - y0, x1 - x0), dtype=torch.float32, device=img.device)
    assert out.is_contiguous() and out.shape == (3, y1 - y0, x1 - x0)
    check(lib.sx_u8_to_chw_lut(C.c_void_p(img.data_ptr()), H, W, img.stride(0), x0, y0, y1 - y0, x1 - x0,
                               C.c_void_p(_lut(mean, std, img.device).data_ptr()), C.c_void_p(out.data_ptr()), _stream()),
          "sx_u8_to_chw_lut")
    return out


def images_to_u8(images):
    """fp32 [B, 3, H, W] decoder output in [-1, 1] → uint8 [B, H, W, 3]: (x/2+0.5).clamp(0,1)·255 rounded half-even."""
    lib = _lib.load()
    x = images.contiguous()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3
    B, _, H, W = x.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=x.device)
    for b in range(B):
        check(lib.sx_chw_to_u8_image(C.c_void_p(x[b].data_ptr()), H, W, C.c_void_p(out[b].data_ptr()), _stream()),
              "sx_chw_to_u8_image")
    return out


def images_to_pil(images):
    from PIL import Image
    return [Image.fromarray(a) for a in images_to_u8(images).cpu().numpy()]


def marker_mask(input_ids, boi_token_id, eoi_token_id, bop_token_id, eop_token_id):
    """ids_cmp_mask of eval_img2text_seed_x_i.py:153-160 for a 1-D int64 id tensor on the GPU → bool [T]."""
    lib = _lib.load()
    ids = input_ids.reshape(-1).to(torch.int64).contiguous()
    if not ids.is_cuda:
        raise RuntimeError("seedx_amd.image_ops: tensors must live on the GPU (no CPU fallback)")
    mask = torch.empty(ids.shape[0], dtype=torch.uint8, device=ids.device)
    check(lib.sx_marker_mask(C.c_void_p(ids.data_ptr()), ids.shape[0], boi_token_id, bop_token_id, eoi_token_id,
                             eop_token_id, C.c_void_p(mask.data_ptr()), _stream()), "sx_marker_mask")
    return mask.view(torch.bool) if hasattr(mask, "view") else mask.bool()


def l2norm_dim1(x, eps=1e-12):
    lib = _lib.load()
    x = x.float().contiguous()
    B, T, D = x.shape
    y = torch.empty_like(x)
    check(lib.sx_l2norm_dim1(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, T, D, eps, _stream()), "sx_l2norm_dim1")
    return y


# ---------------------------------------------------------------------------------------------------------------
# src/processer/transforms.py
# ---------------------------------------------------------------------------------------------------------------
_CLIP_MEAN, _CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


class GpuTransform:
    """Callable stand-in for the torchvision ``Compose([Resize | Resize+CenterCrop, ToTensor, Normalize])`` the reference
    builds (transforms.py:5-86). ``transform(pil_image)`` → fp32 [3, S, S] tensor ON THE GPU (the reference returns a CPU
    tensor that the scripts then move with ``.to(device)``; that move becomes a no-op)."""

    def __init__(self, image_size, keep_ratio, resample, mean, std, expand_to_square=False, device="cuda"):
        self.image_size, self.keep_ratio, self.resample = image_size, keep_ratio, resample
        self.mean, self.std, self.expand_to_square = mean, std, expand_to_square
        self.device = torch.device(device)

    def to(self, device):
        self.device = torch.device(device)
        return self

    def __call__(self, image, out=None):
        S = self.image_size
        img = to_device_u8(image, self.device)
        H, W, _ = img.shape
        if self.expand_to_square and H != W:                                   # 'clipb' expand2square (:37-52)
            side = max(H, W)
            bg = torch.tensor([int(x * 255) for x in _CLIP_MEAN], dtype=torch.uint8, device=self.device)
            sq = bg.view(1, 1, 3).expand(side, side, 3).contiguous()
            if W > H:
                sq[(W - H) // 2:(W - H) // 2 + H] = img
            else:
                sq[:, (H - W) // 2:(H - W) // 2 + W] = img
            img, H, W = sq, side, side
        if self.keep_ratio and not self.expand_to_square:                      # Resize(int) + CenterCrop(int) [ext torchvision]
            if W <= H:
                nw, nh = S, int(S * H / W)
            else:
                nh, nw = S, int(S * W / H)
            img = resize_u8(img, (nw, nh), self.resample)
            top, left = int(round((nh - S) / 2.0)), int(round((nw - S) / 2.0))
            box = (left, top, left + S, top + S)
        else:                                                                  # Resize((S, S))
            img = resize_u8(img, (S, S), self.resample)
            box = (0, 0, S, S)
        return crop_to_tensor(img, box, self.mean, self.std, out=out)


def get_transform(type='clip', keep_ratio=True, image_size=224, device="cuda"):
    if type == 'clip':
        return GpuTransform(image_size, keep_ratio, BILINEAR, _CLIP_MEAN, _CLIP_STD, device=device)
    if type == 'clipa':
        return GpuTransform(image_size, keep_ratio, BILINEAR, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), device=device)
    if type == 'clipb':
        return GpuTransform(image_size, keep_ratio, BILINEAR, _CLIP_MEAN, _CLIP_STD, expand_to_square=keep_ratio, device=device)
    if type == 'sd':
        return GpuTransform(image_size, keep_ratio, BICUBIC, (0.5,), (0.5,), device=device)
    raise NotImplementedError


# ---------------------------------------------------------------------------------------------------------------
# src/inference/any_res.py (integer grid selection is host logic, the pixels never leave the GPU)
# ---------------------------------------------------------------------------------------------------------------
def select_best_resolution(original_size, possible_resolutions):
    ow, oh = original_size
    best_fit, max_eff, min_waste = None, 0, float('inf')
    for width, height in possible_resolutions:
        scale = min(width / ow, height / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = (width * height) - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best_fit = eff, waste, (width, height)
    return best_fit


def select_best_resolution_v2(original_size, possible_resolutions):
    ow, oh = original_size
    oar, oarea = oh / ow, ow * oh
    best_fit, min_ard, min_area_ratio = None, float('inf'), float('inf')
    for width, height in possible_resolutions:
        ar, area = height / width, width * height
        ard = max(ar, oar) / min(ar, oar)
        area_ratio = max(area, oarea) / min(area, oarea)
        if ard < min_ard or (ard == min_ard and area_ratio < min_area_ratio):
            min_ard, min_area_ratio, best_fit = ard, area_ratio, (width, height)
    return best_fit


def _best_resolution(size, grid_pinpoints):
    res = grid_pinpoints if type(grid_pinpoints) is list else ast.literal_eval(grid_pinpoints)
    w1, h1 = select_best_resolution(size, res)
    w2, h2 = select_best_resolution_v2(size, res)
    return (w2, h2) if w1 * h1 > w2 * h2 else (w1, h1)


def get_anyres_image_grid_shape(image_size, grid_pinpoints, patch_size):
    w, h = _best_resolution(image_size, grid_pinpoints)
    return w // patch_size, h // patch_size


def process_anyres_image(image, image_transform, grid_pinpoints, base_image_size):
    """any_res.py:158-201 on the GPU. `image`: PIL image / HxWx3 uint8 array / uint8 tensor; `image_transform`: a
    ``GpuTransform`` (its Resize is the identity on the base_image_size tiles, as in the reference).
    Returns (fp32 [n_tiles + 1, 3, S, S] on the GPU — tiles row-major, then the global view — and patch_pos [n+1, 2])."""
    S = base_image_size
    dev = image_transform.device
    img = to_device_u8(image, dev)
    H, W, _ = img.shape
    bw, bh = _best_resolution((W, H), grid_pinpoints)
    padded = resize_u8(img, (bw, bh), BICUBIC)                                  # resize_and_pad_image(keep_ratio=False) :111
    gx, gy = bw // S, bh // S
    out = torch.empty((gx * gy + 1, 3, image_transform.image_size, image_transform.image_size), dtype=torch.float32,
                      device=dev)
    tile_is_identity = image_transform.image_size == S and not image_transform.keep_ratio
    k = 0
    for i in range(gy):                                                         # divide_to_patches :128-134
        for j in range(gx):
            if tile_is_identity:
                crop_to_tensor(padded, (j * S, i * S, (j + 1) * S, (i + 1) * S), image_transform.mean, image_transform.std,
                               out=out[k])
            else:
                image_transform(padded[i * S:(i + 1) * S, j * S:(j + 1) * S].contiguous(), out=out[k])
            k += 1
    image_transform(resize_u8(img, (S, S), BICUBIC), out=out[k])                # global view, always appended (:187-189)
    x_index = (torch.arange(gx).repeat(gy, 1) + 0.5) / gx
    y_index = (torch.arange(gy).unsqueeze(1).repeat(1, gx) + 0.5) / gy
    patch_pos = torch.stack([x_index, y_index], dim=-1).flatten(0, 1)
    patch_pos = torch.cat([patch_pos, torch.tensor([[0.5, 0.5]])], dim=0)
    return out, patch_pos
