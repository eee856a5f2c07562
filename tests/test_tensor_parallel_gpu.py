"""Tensor-parallel / CFG-parallel paths on ONE GPU with virtual ranks (threads; seedx_amd.parallel.ThreadComm): every rank
runs the real sharded HIP path, the collectives are exchanged through process memory. Checks: TP == single rank == oracle."""
import pytest
import torch

from oracle import restated, restated_unet as ru, weights

pytestmark = pytest.mark.gpu

TP_LLM = dict(hidden_size=512, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=4, vocab_size=500,
              rms_norm_eps=1e-5, max_position_embeddings=512)


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("tp", [2, 4])
def test_llama_tp_prefill_and_decode(dev, tp):
    from seedx_amd.llama import LlamaForCausalLM
    from seedx_amd.parallel import run_virtual_ranks
    dt = torch.float16
    cfg = TP_LLM
    sd = weights.llama_sd(cfg)
    g = torch.Generator().manual_seed(3)
    xe = torch.randn(1, 21, cfg["hidden_size"], generator=g) * 0.5
    lref, _, href = restated.llama_forward(sd, cfg, xe, table_dtype=dt)
    img_ids = torch.arange(400, 466, dtype=torch.int32, device=dev)

    def run(comm):
        torch.cuda.set_device(dev)
        llm = LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=1, comm=comm)
        llm.load_state_dict(dict(sd))
        llm.eval().to(dev, dt)
        out = llm(inputs_embeds=xe.to(dev), output_hidden_states=True)
        ids = torch.full((1, 8), -1, dtype=torch.int32, device=dev)
        hid = torch.zeros((1, 8, cfg["hidden_size"]), device=dev)
        llm._P["cur"].fill_(int(out["logits"][0, 0].argmax()))
        for _ in range(4):
            llm.decode_step(img_ids, ids, hid)
        torch.cuda.synchronize()
        return out["logits"][0, 0].float().cpu(), out["hidden_states"][0][0].float().cpu(), ids.cpu(), hid.cpu()

    from seedx_amd.parallel import Comm
    single = run(Comm())
    ranks = run_virtual_ranks(tp, run)
    e = relerr(single[0], lref[0, -1])
    assert e < 3e-3, e
    for r, (lg, hn, ids, hid) in enumerate(ranks):
        assert lg.shape[0] >= cfg["vocab_size"]
        e_l, e_h = relerr(lg[: cfg["vocab_size"]], lref[0, -1]), relerr(hn, href[0])
        print(f"tp={tp} rank {r}: logits rel-L2 vs oracle {e_l:.2e}, hidden {e_h:.2e}")
        assert e_l < 3e-3 and e_h < 3e-3
        assert torch.equal(lg, ranks[0][0]) and torch.equal(ids, ranks[0][2])      # ranks agree bit for bit
        assert torch.equal(ids, single[2]), (ids, single[2])                        # same greedy tokens as one rank
        assert relerr(hid[0, :4], single[3][0, :4]) < 2e-3


def test_denoise_cfg_parallel(dev):
    """t2i loop with the two guidance branches on two (virtual) ranks == one-rank loop == oracle loop."""
    from seedx_amd.detokenizer import EulerDiscreteScheduler, _DenoiseLoop
    from seedx_amd.parallel import run_virtual_ranks
    from seedx_amd.unet import UNet2DConditionModel
    dtype = torch.float16
    cfg = dict(ru.MINI_UNET, in_channels=4)
    sd = ru.unet_sd(cfg)
    g = torch.Generator().manual_seed(10)
    pe, ne = torch.randn(1, 16, 128, generator=g), torch.randn(1, 16, 128, generator=g)
    pp, npool = torch.randn(1, 128, generator=g), torch.randn(1, 128, generator=g)
    tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]])
    steps = 4
    _, _, init = ru.euler_tables(steps)
    lat0 = torch.randn(1, 4, 16, 16, generator=g) * init
    fn = lambda s, t, e, p, ti: ru.unet_forward(sd, cfg, s, t, e, p, ti)
    ref = ru.t2i_loop(fn, lat0, pe, ne, pp, npool, tid, steps)

    def run(comm):
        torch.cuda.set_device(dev)
        m = UNet2DConditionModel(**cfg)
        m.load_state_dict(dict(sd))
        m.to(dev, dtype)
        loop = _DenoiseLoop(m, use_graph=False, comm=comm)
        out = loop.run(0, lat0, torch.cat([ne, pe]), torch.cat([npool, pp]), tid.repeat(2, 1), EulerDiscreteScheduler(),
                       steps, 7.5)
        torch.cuda.synchronize()
        return out.float().cpu()

    outs = run_virtual_ranks(2, run)
    assert torch.equal(outs[0], outs[1])
    e = relerr(outs[0], ref)
    print(f"CFG-parallel denoise rel-L2 vs oracle {e:.2e}")
    assert e < 5e-3
