#!/bin/bash
# K = 300 projection (c1 / c4 forward Q|K|V): what the C stores cost.  Three builds of the same kernel: product, the tile's MFMAs
# without its C stores (-DEBN_GEMM_EXP_NOSTORE), non-temporal C stores (-DEBN_GEMM_EXP_NT); tools/build_variant.sh builds the variants.
cd "${GRAFT_REPO_ROOT:-.}"
V=ebnerd-benchmark_amd/csrc/variants
for rep in 1 2; do
for lib in "" $V/gemm_nostore.so $V/gemm_nt.so; do
  for shape in "24000 1200 300" "24000 1200 1024" "52800 1200 300" "20480 1200 300" "19456 1216 304" "26880 1200 300"; do  # (the last three: 1520 / 1444 / 1995 tiles of 256 x 64 = 1.98 / 1.88 / 2.6 rounds of 768 resident workgroups)
    echo -n "lib=${lib:-product} "; env ${lib:+EBNERD_HIP_LIB=$lib} python tools/gemm_k_scan.py 0 0 ${shape% *} ${shape##* } 2>&1 | tail -1
  done
done; done
for lib in "" $V/gemm_nt.so; do for c in c1 c2; do
  echo -n "lib=${lib:-product} $c step: "; env ${lib:+EBNERD_HIP_LIB=$lib} python bench.py --config $c --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
