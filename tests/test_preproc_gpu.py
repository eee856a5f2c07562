"""GPU pre/post-processing (csrc/preproc.hip + seedx_amd/image_ops.py) — all BIT-EXACT checks:
against Pillow itself (the reference's resampler), against fixtures produced by the reference's own
process_anyres_image / get_transform (tests/golden/anyres_mini.npz, oracle/gen_golden.py) and against the restated
torchvision / VaeImageProcessor arithmetic of oracle/restated_preproc.py."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import restated_preproc as rp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRIDS = ['1x1', '1x2', '1x3', '2x1', '3x1', '1x4', '4x1', '2x2']


def _pins(S):
    return [[int(s.split('x')[0]) * S, int(s.split('x')[1]) * S] for s in GRIDS]


def _img(rng, H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) * 7 % 256)], -1)
    return ((base.astype(np.int32) + rng.integers(-60, 61, base.shape)).clip(0, 255)).astype(np.uint8)


@pytest.mark.parametrize("hw,size,rs", [((600, 800), (448, 448), "bicubic"), ((896, 896), (448, 448), "bicubic"),
                                        ((333, 517), (448, 448), "bilinear"), ((100, 120), (448, 448), "bicubic"),
                                        ((1792, 1000), (448, 1344), "bicubic"), ((448, 700), (448, 448), "bilinear"),
                                        ((700, 448), (448, 448), "bicubic"), ((1, 5), (7, 3), "bicubic"),
                                        ((2048, 2048), (896, 896), "bicubic")])
def test_resize_is_bit_exact_with_pillow(dev, hw, size, rs):
    from seedx_amd import image_ops as io
    img = _img(np.random.default_rng(hw[0] * 7 + hw[1]), *hw)
    ref = np.asarray(Image.fromarray(img).resize(size, {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}[rs]))
    out = io.resize_u8(torch.from_numpy(img).to(dev), size, rs).cpu().numpy()
    assert out.shape == ref.shape and np.array_equal(out, ref)


def test_anyres_and_transforms_vs_reference_executed_golden(dev):
    """process_anyres_image + get_transform('clip'|'sd'|'clipa'|'clipb') against what the reference's own functions
    produced: every float bit-identical, patch positions identical."""
    from seedx_amd import image_ops as io
    g = np.load(os.path.join(GOLD, "anyres_mini.npz"))
    S = 64
    tf = io.get_transform(type='clip', image_size=S, keep_ratio=False, device=dev)
    for i in range(5):
        out, pos = io.process_anyres_image(Image.fromarray(g[f"img{i}"]), tf, _pins(S), S)
        assert out.is_cuda and tuple(out.shape) == g[f"out{i}"].shape
        assert np.array_equal(out.cpu().numpy(), g[f"out{i}"]), f"image {i}"
        assert np.array_equal(pos.numpy(), g[f"pos{i}"])
    im = Image.fromarray(g["img1"])
    for name, kw in (("clip_keep", dict(type='clip', keep_ratio=True)), ("sd", dict(type='sd', keep_ratio=False)),
                     ("clipb_keep", dict(type='clipb', keep_ratio=True)), ("clipa", dict(type='clipa', keep_ratio=False))):
        t = io.get_transform(image_size=S, device=dev, **kw)(im)
        assert np.array_equal(t.cpu().numpy(), g["tf_" + name]), name


@pytest.mark.parametrize("hw", [(896, 896), (448, 448), (600, 1500), (1400, 500), (1000, 1000)])
def test_anyres_full_size_vs_pillow_pipeline(dev, hw):
    """BASELINE config 2 / 5 input sizes at the real 448 base: [n_crops, 3, 448, 448] bit-identical to the Pillow +
    restated-torchvision pipeline (any_res.py:158-201 restated in oracle/restated_preproc.py)."""
    from seedx_amd import image_ops as io
    img = _img(np.random.default_rng(hw[0] + hw[1]), *hw)
    ref, ref_pos = rp.process_anyres_image(Image.fromarray(img), rp.clip_transform(448), _pins(448), 448)
    tf = io.get_transform(type='clip', image_size=448, keep_ratio=False, device=dev)
    out, pos = io.process_anyres_image(torch.from_numpy(img), tf, _pins(448), 448)
    assert tuple(out.shape) == tuple(ref.shape) and torch.equal(out.cpu(), ref) and torch.equal(pos, ref_pos)
    gx, gy = io.get_anyres_image_grid_shape((hw[1], hw[0]), _pins(448), 448)
    assert gx * gy + 1 == out.shape[0]


def test_latents_to_pil_bit_exact(dev):
    from seedx_amd import image_ops as io
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 96, 160, generator=g) * 0.8
    x[0, :, :4, :4] = torch.tensor([-1.0, 1.0, 0.0, 1e-3])[None, None, :]        # exact edges + a .5 rounding case
    x[1, 0, 0, 0] = (0.5 / 255.0 - 0.5) * 2
    ref = rp.postprocess_pil(x)
    out = io.images_to_pil(x.to(dev))
    for a, b in zip(out, ref):
        assert a.size == b.size and np.array_equal(np.asarray(a), np.asarray(b))


def test_marker_mask_matches_script_logic(dev):
    from seedx_amd import image_ops as io
    boi, eoi, bop, eop = 11, 12, 13, 14
    rng = np.random.default_rng(3)
    cases = []
    well = [1, 5, 5] + [bop] + [20] * 4 + [eop] + [boi] + [21] * 4 + [eoi] + [7, 8]
    cases.append(well)
    cases.append([boi, eoi])
    cases.append([5, 6, 7])
    cases.append([eoi, 3, boi, 4, 4, eop, boi, 9])                                  # closer first, dangling opener
    cases.append([boi, bop, 3, eoi, 4, eop, 5])                                     # nested openers
    for _ in range(20):
        cases.append(rng.choice([boi, eoi, bop, eop, 1, 2, 3, 4, 5, 6], size=int(rng.integers(1, 700))).tolist())
    for ids in cases:
        t = torch.tensor(ids, dtype=torch.long)
        ref = rp.marker_mask(t, boi, eoi, bop, eop)
        out = io.marker_mask(t.to(dev), boi, eoi, bop, eop).cpu()
        assert torch.equal(out, ref), ids[:40]


def test_l2norm_dim1(dev):
    from seedx_amd import image_ops as io
    x = torch.randn(3, 24, 256, generator=torch.Generator().manual_seed(1))
    x[1, :, 7] = 0
    ref = torch.nn.functional.normalize(x)
    out = io.l2norm_dim1(x.to(dev)).cpu()
    assert torch.allclose(out, ref, rtol=2e-6, atol=1e-7)
