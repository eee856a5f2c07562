"""Drop-in plumbing for the reference's own scripts.

The reference's plugin boundary is hydra ``_target_`` factories in YAML + module method signatures: every
``src/inference/eval_*.py`` does ``hydra.utils.instantiate(OmegaConf.load(path), **overrides)`` and then calls
``.eval().to(...)`` / ``generate`` / ``init_pipe`` on the result, plus three ``from diffusers import ...`` classes.

Two ways to put this package underneath, neither touching the call sites:

  1. **YAML overlay** — ``configs/`` in this repo holds files with the reference's file names whose ``_target_`` strings
     point at ``seedx_amd.*``; copy (or ``--config-dir``) them over the reference's ``configs/``.
  2. **Zero-change mode** — ``seedx_amd.dropin.install()`` registers this package's modules under the reference's import
     paths (``src.models.tokenizer.qwen_visual`` …, ``src.processer.transforms``) and under ``diffusers`` for the three
     names the scripts import, so the reference's UNMODIFIED YAMLs and scripts resolve to the MI355X path:
     ``python -m seedx_amd.dropin src/inference/eval_seed_x_detokenizer.py``.

``instantiate`` / ``load_config`` are the small subset of hydra / OmegaConf the scripts rely on (recursive ``_target_``
resolution with keyword overrides); they are used when hydra is not installed and by the tests.
"""
import importlib
import importlib.machinery
import os
import runpy
import sys
import types

ALIASES = {
    "src.models.tokenizer.qwen_visual": "seedx_amd.visual_encoder",      # VisionTransformerWithAttnPool, Resampler
    "src.models.tokenizer.discrete_models": "seedx_amd.discrete_models",  # DiscreteModleIdentity
    "src.models.mllm.modeling_llama_xformer": "seedx_amd.llama",          # LlamaForCausalLM
    "src.models.mllm.seed_x": "seedx_amd.seed_x",                         # ContinuousLVLM
    "src.models.detokenizer.adapter_modules": "seedx_amd.detokenizer",    # SDXLAdapter, SDXLAdapterWithLatentImage
    "src.models.detokenizer.resampler": "seedx_amd.detokenizer",          # ResamplerXLV2
    "src.processer.transforms": "seedx_amd.image_ops",                    # get_transform
    "any_res": "seedx_amd.image_ops",                                     # process_anyres_image (scripts: `from any_res import`)
}


def _locate(path):
    """'pkg.mod.Class.method' → the object (hydra's _locate: longest importable module prefix, then getattr)."""
    parts = path.split(".")
    for i in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:i]))
        except ImportError:
            continue
        for p in parts[i:]:
            obj = getattr(obj, p)
        return obj
    raise ImportError(f"cannot locate {path!r}")


def instantiate(cfg, **overrides):
    """hydra.utils.instantiate for plain dict configs: nested dicts carrying ``_target_`` are instantiated first
    (hydra's default _recursive_=True), keyword overrides replace / add top-level entries."""
    cfg = dict(cfg)
    cfg.update(overrides)
    target = cfg.pop("_target_")
    kwargs = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in cfg.items()}
    return _locate(target)(**kwargs)


def load_config(path):
    """OmegaConf.load for the flat YAMLs of configs/ (no interpolation is used by the inference configs)."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


def _package(name, real_dir=None):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__path__ = [real_dir] if real_dir and os.path.isdir(real_dir) else []
    return m


_INSTALLED = []


def uninstall():
    """Removes every sys.modules entry install() added (tests; a process that wants the real `src` package back)."""
    while _INSTALLED:
        sys.modules.pop(_INSTALLED.pop(), None)


def install(reference_root=None):
    """Registers the alias modules. Parent packages (``src``, ``src.models`` …) are created as namespace stubs whose
    ``__path__`` points into ``reference_root`` (when given) so that other reference modules — data, train — still import."""
    import seedx_amd  # noqa: F401
    for alias, real in ALIASES.items():
        parts = alias.split(".")
        for i in range(1, len(parts)):
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                sys.modules[pkg] = _package(pkg, os.path.join(reference_root, *parts[:i]) if reference_root else None)
                _INSTALLED.append(pkg)
        mod = importlib.import_module(real)
        if alias not in sys.modules:
            _INSTALLED.append(alias)
        sys.modules[alias] = mod
        if len(parts) > 1:
            setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
    if "diffusers" not in sys.modules or getattr(sys.modules["diffusers"], "_seedx_dropin", False):
        from seedx_amd.detokenizer import EulerDiscreteScheduler
        from seedx_amd.unet import UNet2DConditionModel
        from seedx_amd.vae import AutoencoderKL
        d = types.ModuleType("diffusers")
        d.__spec__ = importlib.machinery.ModuleSpec("diffusers", None)
        d._seedx_dropin = True
        d.AutoencoderKL, d.UNet2DConditionModel, d.EulerDiscreteScheduler = AutoencoderKL, UNet2DConditionModel, EulerDiscreteScheduler

        class Transformer2DModel:   # eval_img2edit_seed_x_edit.py:8 imports the name and never uses it
            def __init__(self, *a, **kw):
                raise NotImplementedError("diffusers.Transformer2DModel is only imported, never used, by the SEED-X inference "
                                          "scripts; the UNet's transformer blocks live inside seedx_amd.unet.UNet2DConditionModel")
        d.Transformer2DModel = Transformer2DModel
        if "diffusers" not in sys.modules:
            _INSTALLED.append("diffusers")
        sys.modules["diffusers"] = d
    return sorted(ALIASES)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m seedx_amd.dropin <reference script.py> [args…]   (run from the reference's root)")
    install(reference_root=os.getcwd())
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
