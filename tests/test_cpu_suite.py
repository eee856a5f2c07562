"""CPU-only suite (runs everywhere, no GPU, no /root/reference):
  * the oracle (oracle/restated*.py) against the committed golden fixtures generated from the reference's modules
  * the C-ABI library loads and exports every symbol include/seedx_hip.h declares (no compute calls)
  * host-side logic: weight packing helpers, scheduler tables, tile/shape inventories, the no-fallback guarantees
  * the N > 1 aggregation path with a world_size-2 gloo group
"""
import ctypes
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import restated, restated_unet as ru, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _rel(a, b):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return ((a - b).norm() / b.norm()).item()


def _gold(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


# ---------------------------------------------------------------------------------------------------------
# oracle vs golden vectors (golden = outputs of the reference's own modules, oracle/gen_golden.py)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,cfg", [("vit_hd128", weights.MINI_VIT), ("vit_hd104", weights.MINI_VIT_104)])
def test_oracle_vit_matches_golden(tag, cfg):
    g = _gold(tag + ".npz")
    assert _rel(restated.vit_forward(weights.vit_sd(cfg), cfg, g["x"]), g["y"]) < 2e-5


def test_oracle_llama_matches_golden():
    cfg = weights.MINI_LLM
    sd = weights.llama_sd(cfg)
    g = _gold("llama_mini.npz")
    logits, past, hn = restated.llama_forward(sd, cfg, g["x"])
    assert _rel(logits, g["logits"]) < 2e-5 and _rel(hn, g["hidden"]) < 2e-5
    l2, _, h2 = restated.llama_forward(sd, cfg, sd["model.embed_tokens.weight"][g["tok"].long()], past)
    assert _rel(l2, g["logits2"]) < 2e-5 and _rel(h2, g["hidden2"]) < 2e-5


@pytest.mark.parametrize("tag,dt", [("fp16", torch.float16), ("bf16", torch.bfloat16)])
def test_oracle_llama_matches_ckpt16_golden(tag, dt):
    """tests/golden/llama_mini_ckpt16.npz = the reference's LlamaForCausalLM on 16-bit-representable weights with the RoPE tables of a 16-bit
    run, executed in fp32 (oracle/gen_golden.py run_reference_llama_ckpt16): the oracle with table_dtype reproduces it — prefill of 21
    positions and 3 cached decode steps."""
    gold = np.load(os.path.join(GOLD, "llama_mini_ckpt16.npz"))
    cfg = weights.MINI_LLM
    sd = {k: v.to(dt).float() for k, v in weights.llama_sd(cfg).items()}
    x = torch.from_numpy(gold["x"])
    logits, past, hn = restated.llama_forward(sd, cfg, x, table_dtype=dt)
    assert _rel(logits, torch.from_numpy(gold[tag + ".logits"])) < 2e-5 and _rel(hn, torch.from_numpy(gold[tag + ".hidden"])) < 2e-5
    for i, t in enumerate(gold["toks"].tolist()):
        l2, past, _ = restated.llama_forward(sd, cfg, sd["model.embed_tokens.weight"][torch.tensor([[t]])], past, table_dtype=dt)
        assert _rel(l2[0, -1], torch.from_numpy(gold[tag + ".step_logits"][i])) < 2e-5


def test_oracle_logits_rule_matches_golden():
    g = _gold("logits_rule.npz")
    ids = list(range(400, 466))
    for i, last in enumerate(g["last"].tolist()):
        mine = restated.logits_rule(int(last), g["scores_in"][i].clone(), ids)
        assert torch.equal(mine, g["scores_out"][i])


def test_oracle_resamplers_match_golden():
    g = _gold("resampler_mini.npz")
    sd = weights.resampler_sd(weights._g(7), "", 4, 320, 256)
    assert _rel(restated.resampler_forward(sd, "", g["x"], 2, 1e-5), g["y"]) < 2e-5
    g = _gold("xlv2_mini.npz")
    cfg = weights.MINI_XLV2
    pe, pooled = restated.resampler_xlv2_forward(weights.xlv2_sd(cfg, pre=""), cfg, g["x"], pre="")
    assert _rel(pe, g["prompt"]) < 2e-5 and _rel(pooled, g["pooled"]) < 2e-5


def test_oracle_modules_match_ckpt16_golden():
    """tests/golden/modules_mini_ckpt16.npz (the reference's ViT / Resampler / ResamplerXLV2 in fp32 on fp16-representable weights): the oracle
    on the same rounded weights reproduces it."""
    g = _gold("modules_mini_ckpt16.npz")
    r16 = lambda sd: {k: v.to(torch.float16).float() for k, v in sd.items()}
    for tag, cfg in (("vit_hd128", weights.MINI_VIT), ("vit_hd104", weights.MINI_VIT_104)):
        assert _rel(restated.vit_forward(r16(weights.vit_sd(cfg)), cfg, g[tag + ".x"]), g[tag + ".y"]) < 2e-5
    sd = r16(weights.resampler_sd(weights._g(7), "", 4, 320, 256))
    assert _rel(restated.resampler_forward(sd, "", g["resampler.x"], 2, 1e-5), g["resampler.y"]) < 2e-5
    cfg = weights.MINI_XLV2
    pe, pooled = restated.resampler_xlv2_forward(r16(weights.xlv2_sd(cfg, pre="")), cfg, g["xlv2.x"], pre="")
    assert _rel(pe, g["xlv2.prompt"]) < 2e-5 and _rel(pooled, g["xlv2.pooled"]) < 2e-5


@pytest.mark.parametrize("ckpt16", [False, True], ids=["fp32-weights", "ckpt16"])
@pytest.mark.parametrize("name", ["comp2", "t2i", "anyres5", "truncated"])
def test_oracle_lvlm_generate_matches_reference_executed_golden(name, ckpt16):
    """oracle/restated.lvlm_generate (+ greedy_generate, logits_rule) against tests/golden/lvlm_generate_mini.npz = the
    reference's OWN ContinuousLVLM.generate / prepare_inputs_for_generation / AutoImageTokenGenerationProcessor executed over
    the HF-4.30.2 greedy stand-in (oracle/hf_generate_shim.py): ids, per-step final hidden states, text, image features.
    ckpt16: the second fixture (weights a 16-bit checkpoint holds, RoPE tables of the reference's fp16 runs), oracle with table_dtype."""
    from oracle import gen_golden as gg, hf_generate_shim as hs
    gold = np.load(os.path.join(GOLD, "lvlm_generate_mini_ckpt16.npz" if ckpt16 else "lvlm_generate_mini.npz"))
    cfg, sd_llm, sd_agent = gg.lvlm_case_weights(name, gold, ckpt16)
    kw = gg.lvlm_case_inputs(name, gold)
    tok = hs.StubTokenizer()
    nimg = kw["num_img_gen_tokens"]
    img_ids = tok.encode("".join(["<img>"] + [f"<img_{i:05d}>" for i in range(nimg)] + ["</img>"]))
    ids = tok(kw["prompt"]).input_ids[0].tolist() if "prompt" in kw else kw["input_ids"].reshape(-1).tolist()
    assert ids == gold[f"{name}.input_ids"].reshape(-1).tolist()
    out = restated.lvlm_generate(sd_llm, sd_agent, cfg, {"in_heads": 2, "out_heads": 2}, ids, kw.get("image_embeds"),
                                 kw.get("embeds_cmp_mask"), kw.get("ids_cmp_mask"), kw.get("patch_positions"), img_ids,
                                 img_ids[0], img_ids[-1], kw["max_new_tokens"], nimg, eos_id=tok.eos_token_id, tokenizer=tok,
                                 table_dtype=torch.float16 if ckpt16 else None)
    assert out["ids"] == gold[f"{name}.generate_ids"].tolist()
    assert out["text"] == str(gold[f"{name}.text"])
    assert out["has_img_output"] == bool(gold[f"{name}.has_img_output"]) and out["num_gen_imgs"] == int(gold[f"{name}.num_gen_imgs"])
    assert out["last_hidden"].shape == gold[f"{name}.last_hidden_states"].shape
    assert _rel(out["last_hidden"], gold[f"{name}.last_hidden_states"]) < 2e-5
    if out["has_img_output"]:
        assert out["img_gen_feat"].shape == gold[f"{name}.img_gen_feat"].shape
        assert _rel(out["img_gen_feat"], gold[f"{name}.img_gen_feat"]) < 2e-5
    else:
        assert out["img_gen_feat"] is None and gold[f"{name}.img_gen_feat"].size == 0


def test_oracle_detok_matches_reference_executed_golden():
    """tests/golden/{t2i,edit}_mini.npz were produced by EXECUTING the reference's adapter_modules.py and
    pipeline_stable_diffusion_xl_t2i_edit.py (oracle/diffusers_shim.py supplies the third-party base classes); the
    restated adapter / CFG loops used as the oracle of every GPU test must reproduce them."""
    from oracle import restated_adapter as ra, restated_unet as ru, restated_vae as rv
    t2i, edit = _gold("t2i_mini.npz"), _gold("edit_mini.npz")
    V, X = weights.DETOK_VIT, weights.DETOK_XLV2
    sd_vit, sd_x = weights.vit_sd(V), weights.xlv2_sd(X)
    out = ra.get_image_embeds(sd_vit, V, sd_x, X, image_tensor=t2i["image_tensor"])
    for o, k in zip(out, ("tensor_prompt", "tensor_prompt_neg", "tensor_pooled", "tensor_pooled_neg")):
        assert _rel(o, t2i[k]) < 2e-5, k
    u4, u8 = weights.detok_unet_cfg(4), weights.detok_unet_cfg(8)
    kw = dict(height=128, width=128)
    lat = ra.adapter_generate(sd_vit, V, sd_x, X, ru.unet_sd(u4), u4, t2i["noise"], 5, image_embeds=t2i["feats"], **kw)
    assert _rel(lat, t2i["latents"]) < 5e-5
    # SDXLAdapter.forward (adapter_modules.py:39-52), executed by the reference class → fwd_* arrays
    loss, npred = ra.adapter_forward(sd_x, X, ru.unet_sd(u4), u4, t2i["fwd_noisy"], t2i["fwd_t"], t2i["fwd_feats"],
                                     t2i["fwd_noise"], t2i["fwd_time_ids"])
    assert _rel(npred, t2i["fwd_noise_pred"]) < 2e-5 and abs(float(loss) - float(t2i["fwd_loss"][0])) < 1e-5
    sd8 = ru.unet_sd(u8)
    lat = ra.adapter_generate(sd_vit, V, sd_x, X, sd8, u8, edit["noise"], 5, image_embeds=edit["feats"],
                              image_latents=edit["image_latents"], **kw)
    assert _rel(lat, edit["latents"]) < 5e-5
    il = ra.edit_image_latents(rv.vae_encoder_sd(weights.DETOK_VAE), weights.DETOK_VAE, edit["src_image"])
    lat = ra.adapter_generate(sd_vit, V, sd_x, X, sd8, u8, edit["noise"], 2, image_embeds=edit["feats"],
                              image_latents=il, **kw)
    assert _rel(lat, edit["latents_from_rgb"]) < 5e-5


def test_pillow_coefficient_tables_and_anyres_oracle():
    """Host logic of the GPU preprocessing: image_ops.pil_coeffs (Pillow's precompute_coeffs / normalize_coeffs_8bpc
    restated) fed to a numpy emulation of the two integer kernel passes reproduces Pillow's resize bit for bit; the
    Pillow-based any-res oracle reproduces the fixtures the reference's own process_anyres_image produced; the grid
    selection of the product equals the oracle's."""
    import numpy as np
    from PIL import Image
    from oracle import restated_preproc as rp
    from seedx_amd import image_ops as io
    rng = np.random.default_rng(0)
    for (H, W), (ow, oh), rs in [((300, 400), (224, 224), "bicubic"), ((100, 120), (448, 448), "bicubic"),
                                 ((333, 517), (448, 448), "bilinear"), ((448, 700), (448, 448), "bilinear"),
                                 ((900, 450), (448, 1344), "bicubic"), ((1, 5), (7, 3), "bicubic")]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}[rs]))
        kw, bw, _ = io.pil_coeffs(W, ow, rs)
        kh, bh, _ = io.pil_coeffs(H, oh, rs)
        assert np.array_equal(rp.resample_u8_fixed_point(img, (ow, oh), kw, bw, kh, bh), ref), ((H, W), (ow, oh), rs)
    g = np.load(os.path.join(GOLD, "anyres_mini.npz"))
    S = 64
    pins = [[int(s.split('x')[0]) * S, int(s.split('x')[1]) * S] for s in ['1x1', '1x2', '1x3', '2x1', '3x1', '1x4', '4x1', '2x2']]
    for i in range(5):
        o, p = rp.process_anyres_image(Image.fromarray(g[f"img{i}"]), rp.clip_transform(S), pins, S)
        assert np.array_equal(o.numpy(), g[f"out{i}"]) and np.array_equal(p.numpy(), g[f"pos{i}"])
        H, W, _ = g[f"img{i}"].shape
        gx, gy = io.get_anyres_image_grid_shape((W, H), pins, S)
        assert gx * gy + 1 == o.shape[0]


def test_unet_inventory_matches_sdxl_param_count():
    """The only structural pin available for the diffusers UNet: exact parameter counts."""
    assert ru.unet_param_count(ru.FULL_UNET) == 2_567_463_684
    assert ru.unet_param_count(dict(ru.FULL_UNET, in_channels=8)) == 2_567_475_204
    from seedx_amd import synthetic
    from seedx_amd.unet import SDXL_BASE_CONFIG
    prod = synthetic.unet_param_shapes(SDXL_BASE_CONFIG)
    assert prod == {k: tuple(v) for k, v in ru.unet_param_shapes(ru.FULL_UNET).items()}   # product inventory == oracle's


def test_euler_tables():
    ts, sig, init = ru.euler_tables(50)
    assert ts[0] == 981 and ts[1] == 961 and ts[-1] == 1 and len(sig) == 51 and sig[-1] == 0
    assert abs(init - math.sqrt(float(sig[0]) ** 2 + 1)) < 1e-6
    from seedx_amd.detokenizer import EulerDiscreteScheduler
    s = EulerDiscreteScheduler()
    s.set_timesteps(50)
    assert torch.equal(s.timesteps, ts) and torch.allclose(s.sigmas, sig) and abs(s.init_noise_sigma - init) < 1e-5
    # known answers of the scaled-linear 0.00085 → 0.012 schedule, published with the Stable Diffusion samplers
    # (k-diffusion's sigma_min / sigma_max for SD: 0.0292 / 14.6146): the un-interpolated 1000-step table
    _, full, _ = ru.euler_tables(1000, steps_offset=0)
    assert abs(float(full[0]) - 14.6146) < 1e-3 and abs(float(full[-2]) - 0.0292) < 1e-4
    s.config["steps_offset"] = 0
    s.set_timesteps(1000)
    assert abs(float(s.sigmas[0]) - 14.6146) < 1e-3 and abs(float(s.sigmas[-2]) - 0.0292) < 1e-4


def test_oracle_cfg_loops_are_consistent():
    """t2i loop == edit loop when image guidance is neutral (igs = gs on identical branches collapses the formula)."""
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, 4, 8, 8, generator=g) * 10
    w = torch.randn(4, 4, generator=g) * 0.1

    def unet4(s, t, e, p, ti):
        return torch.einsum("oc,bchw->bohw", w, s[:, :4]) + e.mean(dim=(1, 2))[:, None, None, None]
    pe, ne = torch.randn(1, 4, 8, generator=g), torch.randn(1, 4, 8, generator=g)
    pool = torch.zeros(1, 8)
    tid = torch.zeros(1, 6)
    a = ru.t2i_loop(unet4, lat, pe, ne, pool, pool, tid, 6, 7.5)
    il3 = torch.zeros(3, 4, 8, 8)
    b = ru.edit_loop(unet4, lat, il3, pe, ne, pool, pool, tid, 6, 7.5, 7.5)   # x0_i == x0_u → u + gs*(t - u)
    assert _rel(b, a) < 1e-4


# ---------------------------------------------------------------------------------------------------------
# C-ABI
# ---------------------------------------------------------------------------------------------------------
def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "seedx_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sx_[a-z0-9_]+)\s*\(", src)))


def test_c_abi_library_exports_every_declared_symbol():
    from seedx_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/seedx_hip.h but not exported"
    bound = _lib.load()
    assert set(_lib.SIGNATURES) | {"sx_last_error"} == set(syms), "ctypes SIGNATURES out of sync with the header"
    assert bound.sx_version() >= 1


def test_ctypes_struct_layout_matches_header():
    """Field order/count of the args structs must mirror the header (a mismatch would silently corrupt launches)."""
    from seedx_amd import _lib
    src = open(os.path.join(ROOT, "include", "seedx_hip.h")).read()
    for cname, cls in (("sx_gemm_args", _lib.GemmArgs), ("sx_gemv_args", _lib.GemvArgs), ("sx_attn_args", _lib.AttnArgs),
                       ("sx_attn_small_args", _lib.AttnSmallArgs), ("sx_oneshot_args", _lib.OneshotArgs),
                       ("sx_attn_decode_args", _lib.AttnDecodeArgs), ("sx_gemm_ln_args", _lib.GemmLnArgs),
                       ("sx_attn_f32_args", _lib.AttnF32Args)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(void|float|double|int32_t|int64_t|uint32_t|uint64_t)\s*\*?", "", decl)
            names += [n.strip().lstrip("*") for n in decl.split(",")]
        assert names == [f[0] for f in cls._fields_], (cname, names)


def test_product_path_has_no_cpu_fallback():
    from seedx_amd import ops
    a = torch.zeros(64, 64, dtype=torch.float16)
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm(a, a)          # CPU tensors must be rejected, never silently computed
    for f in os.listdir(os.path.join(ROOT, "seed-x_amd")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "seed-x_amd", f)).read()
            assert "oracle" not in txt.replace("oracle's", "").replace("oracle/", "").replace("the oracle", "") or f == "synthetic.py", \
                f"{f} must not import the oracle"
            assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


# ---------------------------------------------------------------------------------------------------------
# host logic
# ---------------------------------------------------------------------------------------------------------
def test_tiled16_layout_contract():
    """ops.Tiled16 = the SX_TILED16 layout of include/seedx_hip.h: element (m, k) of a [rows <= 32, cols] activation sits in block
    m // 16, tile k // 32, row m % 16, column k % 32 of [rows/16][cols/32][16][32] (what sx_gemv reads with x_layout = 1 and what the
    decode step's producers write). Host-side index arithmetic only."""
    from seedx_amd import ops
    t = ops.Tiled16(5, 96, torch.float32, "cpu")
    t.t.zero_()
    dense = torch.arange(5 * 96, dtype=torch.float32).view(5, 96)
    t.t[0, :, :5] = dense.view(5, 3, 32).permute(1, 0, 2)
    assert t.t.shape == (1, 3, 16, 32) and torch.equal(t.dense(), dense)
    assert t.t[0, 2, 4, 7].item() == dense[4, 2 * 32 + 7].item()
    t2 = ops.Tiled16(20, 64, torch.float32, "cpu")                   # lock-step batches above 16: a second block of tiles
    t2.t.zero_()
    d2 = torch.arange(20 * 64, dtype=torch.float32).view(20, 64)
    pad = torch.zeros(32, 64)
    pad[:20] = d2
    t2.t.copy_(pad.view(2, 16, 2, 32).permute(0, 2, 1, 3))
    assert t2.t.shape == (2, 2, 16, 32) and torch.equal(t2.dense(), d2) and t2.t[1, 1, 3, 5].item() == d2[19, 37].item()
    hdr = open(os.path.join(ROOT, "include", "seedx_hip.h")).read()
    from seedx_amd import _lib
    assert "#define SX_TILED16 0x100" in hdr and _lib.SX_TILED16 == 0x100
    with pytest.raises(AssertionError):
        ops.Tiled16(33, 64, torch.float32, "cpu")


def test_decode_attention_policy_is_tp_invariant():
    """The decode-attention form (one fused launch with one KV split per head, or RoPE + split-KV + combine with just enough
    splits) is chosen from the lock-step batch and the GLOBAL head count: every tensor-parallel degree must make the same
    choice as a single rank, otherwise per-head results would differ between TP degrees in the last bit."""
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.parallel import Comm

    class FakeComm(Comm):
        def __init__(self, rank, world):
            self.rank, self.world = rank, world
    cfg = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=2, num_attention_heads=40, vocab_size=32330,
               rms_norm_eps=1e-5, max_position_embeddings=2048)
    seen = {}
    for G in (1, 4, 13, 16):
        for tp in (1, 2, 4, 8):
            m = LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=G, comm=FakeComm(0, tp))
            seen.setdefault(G, set()).add((m.fused_decode_attention, m.decode_nsplit))
    assert all(len(v) == 1 for v in seen.values()), seen
    assert seen[16] == {(True, 1)} and seen[13] == {(True, 1)}          # 520+ (sequence, head) pairs: no split needed
    assert seen[1] == {(False, 8)} and seen[4] == {(False, 7)}         # few pairs: ~1024 workgroups through KV splits


def test_glu_pack_rows_contract():
    from seedx_amd.llama import glu_pack_rows
    lin = torch.arange(64 * 3).float().view(64, 3)
    gate = -torch.arange(64 * 3).float().view(64, 3)
    w = glu_pack_rows(lin, gate)
    assert w.shape == (128, 3)
    for g in range(4):
        assert torch.equal(w[32 * g:32 * g + 16], lin[16 * g:16 * g + 16])
        assert torch.equal(w[32 * g + 16:32 * g + 32], gate[16 * g:16 * g + 16])


def test_pos_tables_match_oracle():
    from seedx_amd import visual_encoder as ve
    t = torch.randn(256, 32, generator=torch.Generator().manual_seed(1))
    assert torch.equal(ve.get_abs_pos(t, 1024), restated.get_abs_pos(t, 1024))
    assert torch.equal(ve.get_2d_sincos_pos_embed(64, 8), restated.sincos_2d(64, 8))


def test_modules_fail_loudly_without_gpu_or_weights():
    from seedx_amd.visual_encoder import VisionTransformerWithAttnPool
    m = VisionTransformerWithAttnPool(**weights.MINI_VIT)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 112, 112))                    # no weights loaded
    m.load_state_dict(weights.vit_sd(weights.MINI_VIT))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 112, 112))                    # not on a GPU: no CPU fallback


# ---------------------------------------------------------------------------------------------------------
# N > 1 path: world_size-2 gloo group on CPU
# ---------------------------------------------------------------------------------------------------------
_WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from seedx_amd import dist_utils as du
ctx = du.init(backend="gloo")
assert ctx.world == 2 and ctx.rank == int(os.environ["RANK"])
seeds = du.shard_seeds(ctx, steps=3)
du.barrier(ctx)
t = du.max_over_ranks(ctx, 1.0 + ctx.rank)          # rank 1 is slower → everyone sees 2.0
total = du.total_units(ctx, 3)
if ctx.rank == 0:
    print(json.dumps({"t": t, "total": total, "seeds": seeds}))
else:
    print(json.dumps({"seeds": seeds}), file=sys.stderr)
du.finalize(ctx)
"""


def test_two_rank_gloo_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["t"] == 2.0 and rec["total"] == 6 and rec["seeds"] == [0, 2, 4]


def test_bench_gpus2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no external launcher must start two ranks (re-exec under torch.distributed.run) and
    report n_gpus 2; a rank count that does not match --gpus is an error. gloo + the stub workload: no GPU, no model."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--backend", "gloo", "--stub", "--batch", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["generations_per_step"] == 4
    assert abs(rec["value"] - 2 * 3 * 4 / (rec["ms_per_step"] * 3 / 1e3)) < 1e-6 * rec["value"]
    assert [g["rank"] for g in rec["ranks_seen"]] == [0, 1] and len({g["device_id"] for g in rec["ranks_seen"]}) == 2
    assert out.stderr.count("[bench rank ") == 2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub"],
                         capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "rank(s) came up" in (bad.stderr + bad.stdout)


def test_bench_gpus8_launch_path_and_ranks_seen():
    """The driver's N = 8 invocation without 8 GPUs: eight gloo ranks of the stub workload started by bench.py
    itself — rank start-up, request sharding, the barrier / max-over-ranks clock and a JSON line that names all eight ranks."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--backend", "gloo", "--stub", "--batch", "16"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak" and rec["generations_per_step"] == 16
    assert abs(rec["value"] - 8 * 2 * 16 / (rec["ms_per_step"] * 2 / 1e3)) < 1e-6 * rec["value"]
    assert sorted(g["rank"] for g in rec["ranks_seen"]) == list(range(8)) and len({g["device_id"] for g in rec["ranks_seen"]}) == 8
    assert out.stderr.count("[bench rank ") == 8


def test_ranks_seen_flags_shared_devices(monkeypatch):
    """world = 1 returns early; the duplicate-device / missing-rank logic is exercised with a mocked all_gather_object:
    two ranks reporting one device id are flagged (fatal only with strict_devices), a group that returns a wrong rank set raises."""
    from seedx_amd import dist_utils as du
    ctx = du.Ctx(0, 1, 0, "gloo")
    assert du.ranks_seen(ctx)[0]["rank"] == 0
    ctx4 = du.Ctx(0, 4, 0, "gloo")
    me = du.device_identity(ctx4)

    def fake_gather(ids):
        def f(out, obj):
            for i, did in enumerate(ids):
                out[i] = dict(obj, rank=i, local_rank=i, device_id=did)
        return f
    monkeypatch.setattr(du.dist, "all_gather_object", fake_gather(["gpu-a", "gpu-b", "gpu-c", "gpu-d"]))
    got = du.ranks_seen(ctx4)
    assert [g["rank"] for g in got] == [0, 1, 2, 3] and not any(g.get("shared_device") for g in got)
    monkeypatch.setattr(du.dist, "all_gather_object", fake_gather(["gpu-a", "gpu-b", "gpu-a", "gpu-d"]))
    got = du.ranks_seen(ctx4)
    assert [bool(g.get("shared_device")) for g in got] == [True, False, True, False]
    with pytest.raises(RuntimeError, match="distinct device ids"):
        du.ranks_seen(ctx4, strict_devices=True)

    def wrong_ranks(out, obj):
        for i in range(4):
            out[i] = dict(obj, rank=min(i, 2), device_id="gpu-%d" % i)
    monkeypatch.setattr(du.dist, "all_gather_object", wrong_ranks)
    with pytest.raises(RuntimeError, match="expected ranks"):
        du.ranks_seen(ctx4)
    assert me["rank"] == 0


def test_merge_peft_lora_matches_peft_merge():
    """The optional LoRA route (configs/clm_models/llm_seed_x_lora.yaml): a PeftModel-style state dict merges at load to
    W + (B A) alpha / r, norms from modules_to_save — checked against the formula and, where the reference tree exists, against the
    reference's own peft Linear.merge() (proj/peft/src/peft/tuners/lora.py:779-806)."""
    from seedx_amd.llama import LlamaForCausalLM, merge_peft_lora
    g = torch.Generator().manual_seed(0)
    H, I, r, alpha = 32, 64, 4, 8
    base = {"model.embed_tokens.weight": torch.randn(50, H, generator=g), "model.norm.weight": torch.ones(H),
            "lm_head.weight": torch.randn(50, H, generator=g)}
    peft_sd = {}
    p = "model.layers.0."
    shapes = {"self_attn.q_proj": (H, H), "self_attn.k_proj": (H, H), "self_attn.v_proj": (H, H), "self_attn.o_proj": (H, H),
              "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}
    want = dict(base)
    for n, (o, i) in shapes.items():
        w, A, B = torch.randn(o, i, generator=g), torch.randn(r, i, generator=g), torch.randn(o, r, generator=g)
        peft_sd[f"base_model.model.{p}{n}.weight"] = w
        peft_sd[f"base_model.model.{p}{n}.lora_A.default.weight"] = A
        peft_sd[f"base_model.model.{p}{n}.lora_B.default.weight"] = B
        want[p + n + ".weight"] = w + (B @ A) * (alpha / r)
    for n in ("input_layernorm", "post_attention_layernorm"):
        peft_sd[f"base_model.model.{p}{n}.original_module.weight"] = torch.ones(H)
        peft_sd[f"base_model.model.{p}{n}.modules_to_save.default.weight"] = want.setdefault(p + n + ".weight", torch.rand(H, generator=g) + 0.5)
    peft_sd.update({"base_model.model." + k: v for k, v in base.items()})
    got = merge_peft_lora(peft_sd, lora_alpha=alpha)
    assert set(got) == set(want)
    for k in want:
        assert torch.allclose(got[k], want[k], atol=1e-6), k
    m = LlamaForCausalLM(dict(hidden_size=H, intermediate_size=I, num_hidden_layers=1, num_attention_heads=2, vocab_size=50))
    missing, _ = m.load_state_dict(peft_sd, lora_alpha=alpha)            # auto-detected and merged
    assert missing == [] and torch.allclose(m._sd[p + "mlp.down_proj.weight"], want[p + "mlp.down_proj.weight"], atol=1e-6)
    ref_lora = os.path.join("/root/reference", "proj", "peft", "src")
    if os.path.isdir(ref_lora):
        from oracle import refshim
        refshim.install()                                                 # stand-ins for transformers.deepspeed & co (absent here)
        sys.path.insert(0, ref_lora)
        try:
            from peft.tuners.lora import Linear as PeftLinear
        except Exception:                                                 # the reference's peft needs packages this image may lack
            PeftLinear = None
        finally:
            sys.path.remove(ref_lora)
        if PeftLinear is None:      # the formula check above stands on its own; the cross-check needs the reference's peft to import
            pytest.skip("the reference's peft Linear did not import here: merge checked against the formula only")
        if PeftLinear is not None:
            lin = PeftLinear("default", H, I, r=r, lora_alpha=alpha, lora_dropout=0.0)
            n = "mlp.gate_proj"
            with torch.no_grad():
                lin.weight.copy_(peft_sd[f"base_model.model.{p}{n}.weight"])
                lin.lora_A["default"].weight.copy_(peft_sd[f"base_model.model.{p}{n}.lora_A.default.weight"])
                lin.lora_B["default"].weight.copy_(peft_sd[f"base_model.model.{p}{n}.lora_B.default.weight"])
                lin.merge()
            assert torch.allclose(lin.weight.data, got[p + n + ".weight"], atol=1e-5)


# ---- tensor-parallel host logic (parallel.py) ------------------------------------------------------------------------
def test_llama_tp_shard_reconstructs_full_layer():
    """Megatron slices of one decoder layer: column-parallel outputs concatenate, row-parallel partial products sum, to
    the unsharded result (pure tensor arithmetic on the CPU; the GPU path consumes exactly these slices)."""
    from seedx_amd.parallel import llama_tp_shard
    torch.manual_seed(0)
    nh, hd, I = 4, 16, 96
    H = nh * hd
    p = "model.layers.0."
    sd = {p + f"self_attn.{n}.weight": torch.randn(H, H, dtype=torch.float64) for n in ("q_proj", "k_proj", "v_proj", "o_proj")}
    sd.update({p + "mlp.gate_proj.weight": torch.randn(I, H, dtype=torch.float64),
               p + "mlp.up_proj.weight": torch.randn(I, H, dtype=torch.float64),
               p + "mlp.down_proj.weight": torch.randn(H, I, dtype=torch.float64)})
    x = torch.randn(5, H, dtype=torch.float64)
    full_q = x @ sd[p + "self_attn.q_proj.weight"].T
    att = torch.randn(5, H, dtype=torch.float64)
    full_o = att @ sd[p + "self_attn.o_proj.weight"].T
    act = torch.nn.functional.silu(x @ sd[p + "mlp.gate_proj.weight"].T) * (x @ sd[p + "mlp.up_proj.weight"].T)
    full_d = act @ sd[p + "mlp.down_proj.weight"].T
    for tp in (1, 2):
        shards = [llama_tp_shard(sd, p, r, tp, nh, hd) for r in range(tp)]
        assert torch.allclose(torch.cat([x @ s_["q"].T for s_ in shards], dim=1), full_q)
        hl = H // tp
        assert torch.allclose(sum(att[:, r * hl:(r + 1) * hl] @ shards[r]["o"].T for r in range(tp)), full_o)
        part = [torch.nn.functional.silu(x @ s_["gate"].T) * (x @ s_["up"].T) for s_ in shards]
        assert torch.allclose(sum(part[r] @ shards[r]["down"].T for r in range(tp)), full_d)
    with pytest.raises(AssertionError):
        llama_tp_shard(sd, p, 0, 3, nh, hd)          # 4 heads do not split 3 ways


def _tp_comm_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seedx_amd.parallel import TorchDistComm
    c = TorchDistComm()
    t = torch.full((3, 4), float(rank + 1))
    c.all_reduce(t)
    g = c.all_gather(torch.tensor([rank * 10.0, rank * 10.0 + 1]))
    c.barrier()
    q.put((rank, t.tolist(), g.tolist()))
    dist.destroy_process_group()


def test_torchdist_comm_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_tp_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in ps:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p_ in ps:
        p_.join(60)
    for rank, red, gat in res:
        assert red == [[3.0] * 4] * 3
        assert gat == [[0.0, 1.0], [10.0, 11.0]]


def _emul_conv(x_nhwc, w, stride=1, upsample=False, pad_mode=0):
    """torch emulation of ops.conv3x3's geometry (NHWC in, weights [Co, 9*Ci] in (ky, kx, ci) order): pad_mode 0 pads 1 on
    every side, pad_mode 1 pads only bottom/right; upsample = nearest 2x first."""
    import torch.nn.functional as F
    Co, Ci = w.shape[0], w.shape[1] // 9
    x = x_nhwc.permute(0, 3, 1, 2)
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    x = F.pad(x, (0, 1, 0, 1)) if pad_mode else F.pad(x, (1, 1, 1, 1))
    y = F.conv2d(x, w.view(Co, 3, 3, Ci).permute(0, 3, 1, 2), stride=stride)
    return y.permute(0, 2, 3, 1)


def _seqpar_check(comm):
    """Shard arithmetic of the pixel-row-parallel UNet (seedx_amd/seqpar.py + the slab geometry of unet._conv): convolving
    the halo-extended slab and dropping the rows that saw the halo's own padding reproduces the rows of the full conv, for
    stride 1, stride 2 (halo above + zero column left + bottom/right padding) and the fused 2x up-sampling; K|V and row
    gathers restore image order; split GroupNorm statistics add up."""
    from seedx_amd import seqpar
    tp, r = comm.world, comm.rank
    g = torch.Generator().manual_seed(11)
    B, H, W, Ci, Co = 2, 8 * tp, 6, 4, 5
    x = torch.randn(B, H * W, Ci, generator=g)
    w = torch.randn(Co, 9 * Ci, generator=g)
    xl = seqpar.local_rows(x, r, tp, H, W)
    Hl = H // tp
    assert torch.equal(seqpar.gather_rows(xl, comm), x)
    full = _emul_conv(x.view(B, H, W, Ci), w)
    got = _emul_conv(seqpar.with_halo(xl, comm, Hl, W), w)[:, 1:Hl + 1]
    assert torch.allclose(got, full[:, r * Hl:(r + 1) * Hl], atol=1e-5)
    full2 = _emul_conv(x.view(B, H, W, Ci), w, stride=2)
    got2 = _emul_conv(seqpar.with_halo(xl, comm, Hl, W, left_col=True, bottom=False), w, stride=2, pad_mode=1)
    assert got2.shape[1:3] == (Hl // 2, W // 2)
    assert torch.allclose(got2, full2[:, r * Hl // 2:(r + 1) * Hl // 2], atol=1e-5)
    fullu = _emul_conv(x.view(B, H, W, Ci), w, upsample=True)
    gotu = _emul_conv(seqpar.with_halo(xl, comm, Hl, W), w, upsample=True)[:, 2:2 * (Hl + 1)]
    assert torch.allclose(gotu, fullu[:, 2 * r * Hl:2 * (r + 1) * Hl], atol=1e-5)
    kv = torch.randn(B, H * W, 2, 3, 4, generator=g)
    kvl = kv.view(B, H, W, 2, 3, 4)[:, r * Hl:(r + 1) * Hl].reshape(B, Hl * W, 2, 3, 4)
    assert torch.equal(seqpar.gather_kv(kvl, comm), kv)
    assert torch.equal(comm.all_gather_async(kvl).wait(), comm.all_gather(kvl))
    st = torch.stack([xl.double().sum((1, 2)), (xl.double() ** 2).sum((1, 2))], -1)        # per-sample partial sums
    comm.all_reduce(st)
    assert torch.allclose(st[:, 0], x.double().sum((1, 2))) and torch.allclose(st[:, 1], (x.double() ** 2).sum((1, 2)))
    return True


def _seqpar_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seedx_amd.parallel import TorchDistComm
    ok = _seqpar_check(TorchDistComm())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_seqpar_shard_arithmetic_two_ranks_gloo_and_virtual_ranks():
    import torch.multiprocessing as mp
    from seedx_amd.parallel import run_virtual_ranks
    for tp in (2, 4, 8):
        assert all(run_virtual_ranks(tp, _seqpar_check))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_seqpar_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in ps:
        p_.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p_ in ps:
        p_.join(60)
    assert res == [(0, True), (1, True)]


def test_causal_lm_output_access_forms():
    """The forward() return object offers the access forms of transformers' CausalLMOutputWithPast that reference-side callers
    use (modeling_llama_xformer.py:736-746): attribute, key, integer index / slice over the non-None fields, tuple form."""
    from seedx_amd.llama import CausalLMOutputWithPast
    lg, pkv, hs = torch.zeros(1, 3, 5), ((torch.zeros(1), torch.zeros(1)),), (None, torch.ones(1, 3, 2))
    out = CausalLMOutputWithPast(loss=None, logits=lg, past_key_values=pkv, hidden_states=hs, attentions=None)
    assert out.logits is lg and out["logits"] is lg and out[0] is lg
    assert out.past_key_values is pkv and out[1] is pkv and out.hidden_states[-1] is hs[-1]
    assert out.to_tuple() == (lg, pkv, hs) and out[:2] == (lg, pkv)
    assert out.loss is None and out.attentions is None
    with pytest.raises(AttributeError):
        out.no_such_field


def test_split_batch_forward_virtual_ranks_cpu():
    """ViT crop split of the latency mode (parallel.split_batch_forward): every rank encodes its share of the batch, the
    all-gathered result equals the unsplit forward in item order — ragged shares (B not a multiple of the world), B < world."""
    from seedx_amd.parallel import run_virtual_ranks, split_batch_forward
    fn = lambda t: torch.cat([t * 2 + 1, t.flip(-1)], dim=-1)
    for world in (2, 4, 8):
        for B in (1, 2, 5, 20):                       # 2 crops (one 448 px image), 5 (896 px), 20 (BASELINE config 5)
            x = torch.arange(B * 6, dtype=torch.float32).view(B, 2, 3)
            calls = []

            def counted(t):
                calls.append(t.shape[0])
                return fn(t)
            outs = run_virtual_ranks(world, lambda c: split_batch_forward(counted, x, c))
            assert all(torch.equal(o, fn(x)) for o in outs), (world, B)
            assert B == 1 or max(calls) == -(-B // world)          # no rank encodes more than its share


def test_thread_comm_virtual_ranks_cpu():
    from seedx_amd.parallel import run_virtual_ranks

    def fn(comm):
        t = torch.full((2,), float(comm.rank + 1))
        comm.all_reduce(t)
        return t.tolist(), comm.all_gather(torch.tensor([float(comm.rank)])).flatten().tolist()
    out = run_virtual_ranks(3, fn)
    assert all(o == ([6.0, 6.0], [0.0, 1.0, 2.0]) for o in out)


# ---- VAE decoder host logic ------------------------------------------------------------------------------------------
def test_decode_tile_weight_layout_cpu():
    """ops.pack_decode_tiles (host-side re-ordering of a weight for the batched-decode skinny GEMM): element (n, k) of the
    row-major weight lands in tile (n // 16, k // 32) at [n % 16][k % 32], tiles of one 16-row group are consecutive along K,
    every tile is 1 KB contiguous, and the lane that feeds the 16x16x32 MFMA — row r = lane & 15, k-slots [8g, 8g + 8) with
    g = lane >> 4 — reads 16 contiguous bytes at byte offset r*64 + g*16 of its tile (what csrc/decode.hip computes)."""
    from seedx_amd import ops
    N, K = 64, 128
    w = torch.arange(N * K, dtype=torch.int32).to(torch.float32).view(N, K).to(torch.bfloat16)   # values are not unique in
    idx = torch.arange(N * K, dtype=torch.int64).view(N, K)                                       # bf16: track indices too
    t = ops.pack_decode_tiles(w)
    ti = idx.view(N // 16, 16, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(-1)
    assert t.shape == w.shape and t.is_contiguous()
    flat = w.reshape(-1)
    assert torch.equal(t.view(-1), flat[ti])
    for n, k in ((0, 0), (17, 5), (33, 64), (63, 127)):
        tile, r, c = (n // 16) * (K // 32) + k // 32, n % 16, k % 32
        assert ti[tile * 512 + r * 32 + c] == n * K + k
    for lane in (0, 5, 16, 37, 63):                       # operand slice of one lane for tile (group 2, k-block 3)
        r, g = lane & 15, lane >> 4
        base = (2 * (K // 32) + 3) * 512 + r * 32 + 8 * g
        assert ti[base:base + 8].tolist() == [(2 * 16 + r) * K + 3 * 32 + 8 * g + e for e in range(8)]
        assert (base * 2) % 1024 == r * 64 + g * 16


def test_vae_fp32_grade_mode_selection_and_weight_planes():
    """Host logic of the VAE's fp32-grade mode: it is selected exactly when the reference's pipeline would upcast the VAE
    (fp16 + force_upcast, pipeline_stable_diffusion_xl_t2i_edit.py:509-511 / :967-970) or on an explicit fp32 / precision
    request; weight rows carry [hi | lo | hi] per conv tap so that a tripled-K product with [hi | hi | lo] activation rows
    is Ah·Wh + Ah·Wl + Al·Wh ≈ A·W to 16 mantissa bits."""
    from seedx_amd.vae import AutoencoderKL, pack_planes
    cases = [(torch.float16, True, "auto", True), (torch.float16, False, "auto", False), (torch.bfloat16, True, "auto", False),
             (torch.float32, False, "auto", True), (torch.float16, True, "fast", False), (torch.bfloat16, True, "fp32", True)]
    for dt, upcast, prec, want in cases:
        m = AutoencoderKL(block_out_channels=(64, 128), layers_per_block=1, force_upcast=upcast)
        m.dtype, m.precision = dt, prec
        assert m.split == want, (dt, upcast, prec)
        assert m.operand_dtype == (torch.bfloat16 if (want or dt == torch.float32) else dt)
    g = torch.Generator().manual_seed(3)
    Co, taps, Cc, M = 8, 9, 16, 5
    w = torch.randn(Co, taps * Cc, generator=g) * torch.logspace(-3, 3, taps * Cc)
    ww, wa = pack_planes(w, taps, "w"), pack_planes(w, taps, "a")
    assert ww.shape == wa.shape == (Co, taps * 3 * Cc) and ww.dtype == torch.bfloat16
    w3, hi = ww.view(Co, taps, 3, Cc), w.view(Co, taps, Cc).to(torch.bfloat16)
    assert torch.equal(w3[:, :, 0], hi) and torch.equal(w3[:, :, 2], hi)
    assert torch.equal(wa.view(Co, taps, 3, Cc)[:, :, 2], w3[:, :, 1]) and torch.equal(wa.view(Co, taps, 3, Cc)[:, :, 1], hi)
    rec = w3[:, :, 0].double() + w3[:, :, 1].double()
    assert ((rec - w.view(Co, taps, Cc).double()).abs() / w.view(Co, taps, Cc).double().abs()).max() < 2.0 ** -16
    a = torch.randn(M, taps * Cc, generator=g)                       # activation rows, same per-tap plane layout, role "a"
    prod = pack_planes(a, taps, "a").double() @ ww.double().t()      # what the tripled-K GEMM accumulates
    ref = a.double() @ w.double().t()
    single = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t()
    e3, e1 = ((prod - ref).norm() / ref.norm()).item(), ((single - ref).norm() / ref.norm()).item()
    assert e3 < 2e-5 < 1e-3 < e1, (e3, e1)


def test_vae_inventory_matches_oracle_and_param_count():
    from oracle import restated_vae as rv
    from seedx_amd.vae import AutoencoderKL
    assert rv.vae_decoder_param_count(rv.FULL_VAE) == 49_490_179 + 20          # SDXL decoder + post_quant_conv
    m = AutoencoderKL()
    assert m.param_shapes() == rv.vae_decoder_param_shapes(rv.FULL_VAE)
    mini = AutoencoderKL(block_out_channels=(64, 128), layers_per_block=1)
    assert mini.param_shapes() == rv.vae_decoder_param_shapes(rv.MINI_VAE)
    # pre-0.19 diffusers attention names are mapped; the encoder is optional but must be complete; a missing decoder key raises
    sd = rv.vae_sd(rv.MINI_VAE)
    a = "decoder.mid_block.attentions.0."
    for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
        sd[a + old + ".weight"] = sd.pop(a + new + ".weight")[:, :, None, None]
        sd[a + old + ".bias"] = sd.pop(a + new + ".bias")
    missing, _ = mini.load_state_dict(sd)
    assert missing == [] and mini._sd[a + "to_q.weight"].dim() == 2 and not mini.has_encoder
    with pytest.raises(KeyError):                                   # a partial encoder is a broken checkpoint
        mini.load_state_dict(dict(sd, **{"encoder.conv_in.weight": torch.zeros(1)}))
    mini.load_state_dict(dict(sd, **rv.vae_encoder_sd(rv.MINI_VAE)))
    assert mini.has_encoder
    sd.pop("decoder.conv_in.bias")
    with pytest.raises(KeyError):
        mini.load_state_dict(sd)
    # encoder inventory: 34 163 592 + quant_conv 72; whole AutoencoderKL = the published 83 653 863
    assert rv.vae_encoder_param_count(rv.FULL_VAE) == 34_163_592 + 72
    assert rv.vae_encoder_param_count(rv.FULL_VAE) + rv.vae_decoder_param_count(rv.FULL_VAE) == 83_653_863
    assert m.encoder_param_shapes() == rv.vae_encoder_param_shapes(rv.FULL_VAE)


def test_vae_oracle_decode_shapes_cpu():
    from oracle import restated_vae as rv
    sd = rv.vae_sd(rv.MINI_VAE)
    y = rv.vae_decode(sd, rv.MINI_VAE, torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(0)))
    assert y.shape == (1, 3, 16, 16) and torch.isfinite(y).all()


# ---- on-disk checkpoint formats: every from_pretrained() against files written in the layout the
# ---- reference's configs point at (tiny synthetic weights; packing to the GPU is exercised by the -m gpu tests) --------
def test_from_pretrained_reads_reference_checkpoint_layouts(tmp_path):
    import json
    from safetensors.torch import save_file
    from oracle import restated_vae as rv
    from seedx_amd.detokenizer import EulerDiscreteScheduler, ResamplerXLV2, SDXLAdapter
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.seed_x import ContinuousLVLM
    from seedx_amd.unet import UNet2DConditionModel
    from seedx_amd.vae import AutoencoderKL
    from seedx_amd.visual_encoder import Resampler, VisionTransformerWithAttnPool

    # (1) pretrained/QwenViT/qwen_vit_G.pt: torch pickle of the visual state dict (reload_qwen_vit.py:9-13)
    vcfg = weights.MINI_VIT
    vsd = weights.vit_sd(vcfg)
    torch.save(vsd, tmp_path / "qwen_vit_G.pt")
    vit = VisionTransformerWithAttnPool.from_pretrained(pretrained_model_path=str(tmp_path / "qwen_vit_G.pt"), **vcfg)
    assert set(vit.expected_keys()) - {"attn_pool.pos_embed"} <= set(vsd)

    # (2) HF Llama directory: config.json + safetensors shard (llm_seed_x_i.yaml:1-3)
    lcfg = weights.MINI_LLM
    ldir = tmp_path / "llm"
    ldir.mkdir()
    json.dump(dict(lcfg, architectures=["LlamaForCausalLM"], model_type="llama"), open(ldir / "config.json", "w"))
    lsd = {k: v.contiguous() for k, v in weights.llama_sd(lcfg).items()}
    save_file(lsd, str(ldir / "model-00001-of-00001.safetensors"))
    llm = LlamaForCausalLM.from_pretrained(str(ldir), torch_dtype=torch.bfloat16, max_cache_len=64)
    assert llm.H == lcfg["hidden_size"] and llm.L == lcfg["num_hidden_layers"] and llm.dtype == torch.bfloat16
    assert set(llm.expected_keys()) <= set(llm._sd)

    # (3) agent/pytorch_model.bin: input_resampler.* / output_resampler.* / patch_pos_embed (seed_x.py:231-233)
    asd = weights.agent_sd(lcfg, 128, in_grid=4, out_grid=4)
    torch.save(asd, tmp_path / "agent.bin")
    H = lcfg["hidden_size"]
    agent = ContinuousLVLM.from_pretrained(llm, Resampler(4, H, 2, kv_dim=128), Resampler(4, 128, 2, kv_dim=H),
                                           pretrained_model_path=str(tmp_path / "agent.bin"), add_patch_pos=True)
    assert agent.patch_pos_embed is not None

    # (4) diffusers unet/ directory: config.json + diffusion_pytorch_model.safetensors (eval_seed_x_detokenizer.py:30-36)
    ucfg = ru.MINI_UNET
    udir = tmp_path / "sdxl" / "unet"
    udir.mkdir(parents=True)
    boc = ucfg["block_out_channels"]
    json.dump(dict(in_channels=ucfg["in_channels"], out_channels=ucfg["out_channels"], block_out_channels=list(boc),
                   layers_per_block=ucfg["layers_per_block"], transformer_layers_per_block=list(ucfg["transformer_layers"]),
                   attention_head_dim=list(ucfg["heads"]), cross_attention_dim=ucfg["cross_attention_dim"],
                   addition_time_embed_dim=ucfg["addition_time_embed_dim"],
                   projection_class_embeddings_input_dim=ucfg["pooled_dim"] + 6 * ucfg["addition_time_embed_dim"],
                   norm_num_groups=ucfg["norm_groups"], sample_size=16,
                   down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                   up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"]), open(udir / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in ru.unet_sd(ucfg).items()}, str(udir / "diffusion_pytorch_model.safetensors"))
    unet = UNet2DConditionModel.from_pretrained(str(tmp_path / "sdxl"), subfolder="unet")
    assert unet.cfg["block_out_channels"] == tuple(boc) and unet.cfg["pooled_dim"] == ucfg["pooled_dim"]

    # (5) scheduler/scheduler_config.json
    sdir = tmp_path / "sdxl" / "scheduler"
    sdir.mkdir()
    json.dump(dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                   steps_offset=1, timestep_spacing="leading", prediction_type="epsilon"), open(sdir / "scheduler_config.json", "w"))
    sch = EulerDiscreteScheduler.from_pretrained(str(tmp_path / "sdxl"), subfolder="scheduler")
    sch.set_timesteps(50)
    assert float(sch.timesteps[0]) == 981.0 and len(sch.sigmas) == 51

    # (6) vae/ directory (decoder + encoder), .bin fallback
    vdir = tmp_path / "sdxl" / "vae"
    vdir.mkdir()
    mv = rv.MINI_VAE
    json.dump(dict(in_channels=3, out_channels=3, block_out_channels=list(mv["block_out_channels"]),
                   layers_per_block=mv["layers_per_block"], latent_channels=4, norm_num_groups=32,
                   scaling_factor=0.13025, force_upcast=True, act_fn="silu"), open(vdir / "config.json", "w"))
    torch.save(dict(rv.vae_sd(mv), **rv.vae_encoder_sd(mv)), vdir / "diffusion_pytorch_model.bin")
    vae = AutoencoderKL.from_pretrained(str(tmp_path / "sdxl"), subfolder="vae")
    assert vae.has_encoder and vae.config.scaling_factor == 0.13025 and vae.config.force_upcast

    # (7) seed_detokenizer/*/pytorch_model.bin: resampler.* + unet.* (adapter_modules.py:62-65)
    xcfg = weights.MINI_XLV2
    ck = dict(weights.xlv2_sd(xcfg))
    ck.update({"unet." + k: v for k, v in ru.unet_sd(ucfg).items()})
    torch.save(ck, tmp_path / "detok.bin")
    ad = SDXLAdapter.from_pretrained(UNet2DConditionModel(**ucfg), ResamplerXLV2(normalize=False, **xcfg),
                                     pretrained_model_path=str(tmp_path / "detok.bin"))
    assert ad.unet._sd is not None and ad.resampler is not None
    ck.pop("unet.conv_in.weight")
    torch.save(ck, tmp_path / "detok_bad.bin")
    with pytest.raises(KeyError):
        SDXLAdapter.from_pretrained(UNet2DConditionModel(**ucfg), ResamplerXLV2(normalize=False, **xcfg),
                                    pretrained_model_path=str(tmp_path / "detok_bad.bin"))


def test_gemm_tile_picker_host_logic():
    """pick_tile() is pure host code (cost model fitted on the MI355X sweep): check the decisions the bench relies on,
    through the C-ABI query, without launching anything."""
    from seedx_amd import _lib
    lib = _lib.load()
    # 7 / 8 = the ping-pong schedule (csrc/gemm_pp.hip) of the 256x256 / 256x320 tiles
    names = ["128x128", "128x80", "64x128", "64x64", "256x256", "256x320", "256x160", "256x256", "256x320"]
    pick = lambda M, N, K, glu=0, conv=0: names[lib.sx_gemm_pick_tile(M, N, K, glu, conv)]
    assert lib.sx_gemm_pick_tile(32768, 3840, 1280, 0, 0) == 8 and lib.sx_gemm_pick_tile(32768, 10240, 1280, 1, 0) == 7
    # SDXL channel counts are k*320: whole rounds of 256 tiles beat the ragged 256x256 grid
    assert pick(16384, 1280, 1280) == "256x320"          # 64 x 4 = 256 tiles = one round
    assert pick(8192, 1280, 1280) == "256x160"           # 32 x 8 = 256 tiles
    assert pick(65536, 640, 2560) == "256x320"
    assert pick(8192, 8192, 8192) == "256x256"
    # GEGLU projections may only use GLU-capable tiles (32-row [linear|gate] groups inside one wave)
    assert pick(16384, 10240, 1280, glu=1) in ("256x256", "128x128", "64x128", "64x64")
    assert pick(16384, 10240, 1280, glu=1) == "256x256"
    # small problems (batch-1 UNet, LLM prefill) stay on the 4-wave tiles
    assert pick(2048, 1280, 1280) in ("64x64", "64x128", "128x80")
    assert pick(165, 15360, 5120) in ("64x128", "64x64")
    # convs: N = 320 is a single 256x320 column at CFG batch 16+, 256x160 at batch 1
    assert pick(262144, 320, 2880, conv=1) == "256x320"
    assert pick(32768, 320, 2880, conv=1) == "256x160"
    for M, N, K in [(1, 64, 64), (7, 5120, 5120), (1000000, 128, 1152)]:
        assert 0 <= lib.sx_gemm_pick_tile(M, N, K, 0, 0) <= 8


def test_layernorm_fold_host_algebra_and_tile_gate():
    """ops.fold_layernorm: rstd (x W'^T - mu colsum) + bias' == LayerNorm(x) W^T + bias (fp32 weights: exact algebra up to rounding);
    GLU-packed rows fold row by row. ops.ln_fold_ok / sx_gemm_ln: the fold exists only where every neighbour GEMM runs on a
    ping-pong tile — the C-ABI call refuses a lock-step shape before any launch."""
    import ctypes as C
    from seedx_amd import _lib, ops
    g = torch.Generator().manual_seed(8)
    M, K, N = 64, 96, 48
    x = torch.randn(M, K, generator=g, dtype=torch.float64) * 2 + 0.7
    w, b = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    wf, cs, bf = ops.fold_layernorm(w, b, gamma, beta)
    mu, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    ref = ((x - mu) * rstd * gamma.double() + beta.double()) @ w.double().T + b.double()
    got = rstd * (x @ wf.double().T - mu * cs.double()) + bf.double()
    assert _rel(got, ref) < 1e-6
    # the UNet's shapes: folded at the bench batch (32 samples) and at 16, separate LayerNorm launches at batch 1 (2 samples)
    shapes = lambda C_: dict(consumers=[(3 * C_, False), (C_, False), (8 * C_, True)], producers=[C_, 4 * C_])
    assert ops.ln_fold_ok(32 * 1024, 1280, **shapes(1280)) and ops.ln_fold_ok(32 * 4096, 640, **shapes(640))
    assert ops.ln_fold_ok(16 * 1024, 1280, **shapes(1280))
    assert not ops.ln_fold_ok(2 * 1024, 1280, **shapes(1280))
    lib = _lib.load()
    args, la = _lib.GemmArgs(), _lib.GemmLnArgs()
    dummy = C.create_string_buffer(64)
    args.A = args.W = args.C = C.addressof(dummy)
    args.M, args.N, args.K, args.ldc = 2048, 1280, 1280, 1280
    args.dtype, args.out_dtype = _lib.SX_F16, _lib.SX_F32
    la.x16_out = la.row_stats_out = C.addressof(dummy)
    la.ld_x16 = 1280
    assert lib.sx_gemm_ln(C.byref(args), C.byref(la), None) != 0
    assert "ping-pong" in lib.sx_last_error().decode()


def test_from_pretrained_with_peft_adapter_directory(tmp_path):
    """`LlamaForCausalLM.from_pretrained(base_dir, peft_adapter=adapter_dir)`: the on-disk form PeftModel.save_pretrained
    writes — adapter_config.json carrying lora_alpha, `….lora_A.weight` / `….lora_B.weight` keys WITHOUT the adapter name, the
    modules_to_save copies as plain `….weight` — merged into a base checkpoint directory (config.json + safetensors shard)."""
    import json
    from safetensors.torch import save_file
    from seedx_amd.llama import LlamaForCausalLM
    g = torch.Generator().manual_seed(3)
    H, I, V, r, alpha = 32, 64, 50, 4, 16
    cfg = dict(hidden_size=H, intermediate_size=I, num_hidden_layers=1, num_attention_heads=2, vocab_size=V)
    p = "model.layers.0."
    shapes = {"self_attn.q_proj": (H, H), "self_attn.k_proj": (H, H), "self_attn.v_proj": (H, H), "self_attn.o_proj": (H, H),
              "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}
    base = {"model.embed_tokens.weight": torch.randn(V, H, generator=g), "model.norm.weight": torch.ones(H),
            "lm_head.weight": torch.randn(V, H, generator=g),
            p + "input_layernorm.weight": torch.ones(H), p + "post_attention_layernorm.weight": torch.ones(H)}
    adapter, want = {}, {}
    for n, (o, i) in shapes.items():
        w, A, B = torch.randn(o, i, generator=g), torch.randn(r, i, generator=g), torch.randn(o, r, generator=g)
        base[p + n + ".weight"] = w
        adapter[f"base_model.model.{p}{n}.lora_A.weight"] = A
        adapter[f"base_model.model.{p}{n}.lora_B.weight"] = B
        want[p + n + ".weight"] = w + (B @ A) * (alpha / r)
    new_norm = torch.rand(H, generator=g) + 0.5
    adapter[f"base_model.model.{p}input_layernorm.weight"] = new_norm                    # a modules_to_save copy, as saved
    bdir, adir = tmp_path / "base", tmp_path / "adapter"
    bdir.mkdir(); adir.mkdir()
    json.dump(cfg, open(bdir / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in base.items()}, str(bdir / "model.safetensors"))
    json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "target_modules": [n.split(".")[-1] for n in shapes]}, open(adir / "adapter_config.json", "w"))
    save_file({k: v.contiguous() for k, v in adapter.items()}, str(adir / "adapter_model.safetensors"))
    m = LlamaForCausalLM.from_pretrained(str(bdir), peft_adapter=str(adir))
    for k, v in want.items():
        assert torch.allclose(m._sd[k], v, atol=1e-5), k
    assert torch.equal(m._sd[p + "input_layernorm.weight"], new_norm)
    assert torch.equal(m._sd[p + "post_attention_layernorm.weight"], base[p + "post_attention_layernorm.weight"])
    plain = LlamaForCausalLM.from_pretrained(str(bdir))
    assert torch.equal(plain._sd[p + "mlp.up_proj.weight"], base[p + "mlp.up_proj.weight"])


def test_attn64_accumulators_are_private(tmp_path):
    """`attn64_kernel` (csrc/attn.hip; lab-only experiment of round 6) keeps its O^T accumulators in a[0:95] behind the compiler's back:
    every asm statement that touches them names the registers literally and clobbers all 96. That is only sound while the compiler
    itself never allocates one of them — checked here on the ISA it generates (cross-compiles without a GPU)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seed-x_amd", "csrc", "attn.hip")
    out = tmp_path / "attn.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffast-math", "-fno-finite-math-only", "-mllvm",
                    "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", src, "-o", str(out)], check=True, capture_output=True)
    lines = out.read_text().splitlines()
    kernels, cur, in_asm, bad = 0, None, False, []
    for ln in lines:
        m = re.match(r"^(_ZN8sxk_attn13attn64_kernel\w+):", ln)
        if m:
            cur, kernels = m.group(1), kernels + 1
            continue
        if cur is None:
            continue
        if "s_endpgm" in ln:
            cur = None
            continue
        if "ASMSTART" in ln:
            in_asm = True
        elif "ASMEND" in ln:
            in_asm = False
        elif not in_asm:
            for r in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):\d+\]", ln.split(";")[0]):
                if int(r.group(1) or r.group(2)) < 96:
                    bad.append((cur, ln.strip()))
    assert kernels >= 2, "attn64_kernel instantiations not found in the ISA"
    assert not bad, f"compiler-generated use of a private accumulator register: {bad[:3]}"


def test_llm_mode_selection_and_memory_footprint(monkeypatch):
    """Host logic of LlamaForCausalLM's numerics modes (no GPU): precise mode is the default for EVERY lock-step batch size up to 32 (round 6:
    four operand blocks above 16 sequences), SX_LLM_PRECISE32=0 sends 17..32 sequences to the plain flow, SX_LLM_PRECISE=0 / precise=False
    everything; the fp32 decode attention's key splits follow the GLOBAL (sequence, head) count; memory_footprint() prices weights, decode
    tiles and the KV cache of the chosen mode before anything is allocated."""
    from seedx_amd.llama import LlamaForCausalLM
    cfg = dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40, vocab_size=32330,
               rms_norm_eps=1e-5, max_position_embeddings=4096)
    monkeypatch.delenv("SX_LLM_PRECISE", raising=False)
    monkeypatch.delenv("SX_LLM_PRECISE32", raising=False)
    for G in (1, 16, 17, 32):
        assert LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=G).precise
    monkeypatch.setenv("SX_LLM_PRECISE32", "0")
    assert LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=16).precise
    assert not LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=17).precise
    monkeypatch.delenv("SX_LLM_PRECISE32")
    monkeypatch.setenv("SX_LLM_PRECISE", "0")
    assert not LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=4).precise
    monkeypatch.delenv("SX_LLM_PRECISE")
    assert not LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=4, precise=False).precise
    # key splits of the T = 1 fp32 attention: ~1024 workgroups, none from 512 (sequence, head) pairs
    assert [LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=G).decode_nsplit_f32 for G in (1, 4, 12, 13, 32)] == [16, 7, 3, 1, 1]
    # footprint at 13B dims, 16 sequences, 1024-token cache: 26 GB of weights, as much again in decode tiles, cache by mode
    m = LlamaForCausalLM(dict(cfg), max_cache_len=1024, max_batch=16)
    fp = m.memory_footprint()                          # kv_v16 is decided at pack time (dtype): the constructor's figure is the all-fp32 cache
    per_layer = (3 * 5120 * 5120 + 5120 * 5120 + 2 * 13824 * 5120 + 5120 * 13824) * 2
    assert fp["weights"] == 40 * per_layer + (32330 + m.V_l) * 5120 * 2 and fp["decode_tiles"] == 40 * per_layer + m.V_l * 5120 * 2
    assert fp["kv_cache"] == 40 * 16 * 40 * 1024 * 128 * 8 and fp["total"] == fp["weights"] + fp["decode_tiles"] + fp["kv_cache"]
    m.kv_v16 = True
    assert m.memory_footprint()["kv_cache"] == 40 * 16 * 40 * 1024 * 128 * 6
    plain = LlamaForCausalLM(dict(cfg), max_cache_len=1024, max_batch=4, precise=False)
    assert plain.memory_footprint()["decode_tiles"] == 0 and plain.memory_footprint()["kv_cache"] == 40 * 4 * 40 * 1024 * 128 * 4
