#!/bin/bash
# K = 300 projection: the three workgroups of a CU's first dispatch round started a third of a tile apart (variant build
# -DEBN_GEMM_EXP_STAGGER, EBN_GEMM_STAGGER_PCT = percent of that third; 0 = off).  tools/build_variant.sh gemm_stagger ebn_gemm.hip -DEBN_GEMM_EXP_STAGGER
cd "${GRAFT_REPO_ROOT:-.}"
V=ebnerd-benchmark_amd/csrc/variants/gemm_stagger.so
for rep in 1 2; do for pct in 0 50 100 150 200; do for shape in "24000 1200 300" "24000 1200 1024"; do
  echo -n "stagger_pct=$pct "; EBNERD_HIP_LIB=$V EBN_GEMM_STAGGER_PCT=$pct python tools/gemm_k_scan.py 0 0 ${shape% *} ${shape##* } 2>&1 | tail -1
done; done; done
for pct in 0 100 0 100; do for c in c1 c2; do
  echo -n "stagger_pct=$pct $c step: "; EBNERD_HIP_LIB=$V EBN_GEMM_STAGGER_PCT=$pct python bench.py --config $c --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
