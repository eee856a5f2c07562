"""Import the reference's own modules on CPU (build container only: /root/reference does not travel).

torchvision / deepspeed / xformers are absent here, so stand-ins are registered in ``sys.modules`` before importing
``src.models.*`` (torchvision.transforms: the restated PIL-based transforms of oracle/restated_preproc.py). ``xformers.ops.memory_efficient_attention`` is restated with SDPA
following the call site modeling_llama_xformer.py:221-238 (layout [B, M, H, K]; causal iff attn_bias is a
LowerTriangularMask)."""
import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models"))


def _mod(name):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    return m


def install():
    import torch
    import torch.nn.functional as F
    import transformers  # noqa: F401  (must be imported before the shims)
    import transformers.activations  # noqa: F401  (pulls transformers.integrations.deepspeed before the fake exists)

    from . import restated_preproc
    restated_preproc.install_torchvision_shim()     # working Resize / CenterCrop / ToTensor / Normalize on PIL input
    if "deepspeed" not in sys.modules:
        ds = _mod("deepspeed")
        ds.zero = types.SimpleNamespace(GatheredParameters=None, Init=None)
        sys.modules["deepspeed"] = ds
    if "transformers.deepspeed" not in sys.modules:
        tds = _mod("transformers.deepspeed")
        tds.is_deepspeed_zero3_enabled = lambda: False
        sys.modules["transformers.deepspeed"] = tds
    if "xformers" not in sys.modules:
        xf = _mod("xformers")
        xops = _mod("xformers.ops")

        class LowerTriangularMask:
            pass

        def memory_efficient_attention(q, k, v, attn_bias=None):
            # [B, M, H, K] layout (modeling_llama_xformer.py:221-238)
            qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
            causal = isinstance(attn_bias, LowerTriangularMask)
            o = F.scaled_dot_product_attention(qt, kt, vt, is_causal=causal)
            return o.transpose(1, 2)

        xops.LowerTriangularMask = LowerTriangularMask
        xops.memory_efficient_attention = memory_efficient_attention
        xf.ops = xops
        sys.modules["xformers"] = xf
        sys.modules["xformers.ops"] = xops
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def reference_modules():
    """Returns the reference classes used as the ground truth."""
    install()
    from src.models.tokenizer.qwen_visual import VisionTransformerWithAttnPool, Resampler
    from src.models.mllm.modeling_llama_xformer import LlamaForCausalLM
    from src.models.detokenizer.resampler import ResamplerXLV2
    from src.models.mllm.generation import AutoImageTokenGenerationProcessor
    return dict(VisionTransformerWithAttnPool=VisionTransformerWithAttnPool, Resampler=Resampler,
                LlamaForCausalLM=LlamaForCausalLM, ResamplerXLV2=ResamplerXLV2,
                AutoImageTokenGenerationProcessor=AutoImageTokenGenerationProcessor)
