"""Print which hipBLASLt kernels torch.matmul picks for the large bf16 shapes (run under rocprofv3 --kernel-trace)."""
import torch
dev = torch.device("cuda:0")
for M, N, K in [(8192, 8192, 8192), (4096, 4096, 4096), (16384, 10240, 1280), (16384, 1280, 1280), (2048, 1280, 1280)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3):
        torch.matmul(a, w.t())
torch.cuda.synchronize()
