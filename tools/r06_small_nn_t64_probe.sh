#!/bin/bash
# user-encoder Q|K|V projection (640 x 1200 x 400 at c2, 640 x 768 x 256 at c3) on 64 x 64 tiles of 1024 threads (one dispatch round) instead of 32 x 32
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/build_variant.sh small_nn_t64 ebn_gemm.hip -DEBN_GEMM_EXP_SMALL_NN_T64 > /dev/null 2>&1
V=ebnerd-benchmark_amd/csrc/variants/small_nn_t64.so
for rep in 1 2; do for lib in "" $V; do for shape in "640 1200 400" "640 768 256" "1280 1200 400" "800 512 768"; do
  echo -n "lib=${lib:-product} "; env ${lib:+EBNERD_HIP_LIB=$lib} python tools/gemm_k_scan.py 0 0 ${shape% *} ${shape##* } 2>&1 | tail -1
done; done; done
for lib in "" $V "" $V; do for c in c2 c3; do
  echo -n "lib=${lib:-product} $c step: "; env ${lib:+EBNERD_HIP_LIB=$lib} python bench.py --config $c --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
