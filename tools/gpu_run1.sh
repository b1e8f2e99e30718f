set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r02a/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_c2.json 2> gpurun_out/r02a/bench_c2.err
timeout 300 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_c5.json 2> gpurun_out/r02a/bench_c5.err
timeout 300 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_c4.json 2> gpurun_out/r02a/bench_c4.err
timeout 300 python bench.py --gpus 2 --config c4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02a/bench_c4_2r.json 2> gpurun_out/r02a/bench_c4_2r.err
timeout 300 python bench.py --gpus 2 --config c5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02a/bench_c5_2r.json 2> gpurun_out/r02a/bench_c5_2r.err
timeout 300 python bench.py --config c3 --steps 20 --warmup 5 > gpurun_out/r02a/bench_c3.json 2> gpurun_out/r02a/bench_c3.err
timeout 300 python bench.py --config c1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_c1.json 2> gpurun_out/r02a/bench_c1.err
tail -3 gpurun_out/r02a/pytest.log
