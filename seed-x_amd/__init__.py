"""seedx-mi355x: MI355X-native (gfx950) SEED-X inference hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed plumbing); every dense op
runs in hand-written HIP kernels from ``csrc/`` behind the C-ABI declared in ``include/seedx_hip.h``.
"""
__version__ = "0.1.0"
