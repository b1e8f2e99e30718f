mkdir -p gpurun_out/r06final
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r06final/smoke.log 2>&1; tail -1 gpurun_out/r06final/smoke.log
python -m pytest tests -q -m gpu > gpurun_out/r06final/tests.log 2>&1; tail -6 gpurun_out/r06final/tests.log
(time python bench.py) > gpurun_out/r06final/bench_default.log 2>&1; tail -4 gpurun_out/r06final/bench_default.log | cut -c1-300
