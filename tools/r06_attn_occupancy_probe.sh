#!/bin/bash
# How the attention-core forward (one wave per (title, head), attn_mfma_fwd_kernel<20, 30>) depends on the number of resident waves per CU:
# the occupancy a per-title fused news-encoder tail kernel would run its attention phase at (verdict r5 item 2).  Unused LDS per workgroup
# (variant build -DEBN_ATTN_EXP_PAD_LDS, EBN_ATTN_PAD_LDS bytes) lowers the resident workgroups per CU: 18.4 KB per 4-wave workgroup as shipped
# (8 workgroups = 32 waves per CU); + 8000 -> 6 (24 waves); + 21000 -> 4 (16); + 34000 -> 3 (12); + 60000 -> 2 (8); + 100000 -> 1 (4).
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/build_variant.sh attn_pad ebn_attention_mfma.hip -DEBN_ATTN_EXP_PAD_LDS > /dev/null 2>&1  # (against the current objects)
V=ebnerd-benchmark_amd/csrc/variants/attn_pad.so
for rep in 1 2; do for pad in 0 8000 21000 34000 60000 100000; do
  echo -n "pad=$pad "; EBNERD_HIP_LIB=$V EBN_ATTN_PAD_LDS=$pad python tools/tail_probe.py 800 30 a 2>&1 | grep "attn fwd"
done; done
