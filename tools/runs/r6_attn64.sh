#!/bin/bash
# round 6: head_dim-64 attention experiments. Variants (sx_attention_variant): 0 shipped; 16 + OPT = attn_kernel<.., 64, OPT>
# (18 = the round-5 kernel, 50 = + deferred reference max = shipped now); 2000 + OPT = attn64_kernel (one wave per SIMD; 2007 = full,
# + 16 no v_exp, + 32 no P.V MFMAs, + 64 no S MFMAs, + 128 no K/V DMA, + 256 no softmax VALU, + 512 s_memtime probe);
# 100000 + 1000 KiB + variant = attn_kernel variant with that much extra LDS (caps the workgroups per CU)
cd $GRAFT_REPO_ROOT
export ATTN_LAB_SHAPES=${ATTN_LAB_SHAPES:-0,1,5}
timeout 900 tools/lab/attn_lab 5 "$@" 2>&1 | tee gpurun_out/r6_attn64_lab.log | tail -80
