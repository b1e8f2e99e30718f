mkdir -p gpurun_out/r06e
python -m pytest tests/test_docvec_model.py -k "finale" -x -q -m gpu > gpurun_out/r06e/tests.log 2>&1; tail -3 gpurun_out/r06e/tests.log
bash tools/r06_k300_store_probe.sh > gpurun_out/r06e/k300_store_probe.txt 2>&1; cat gpurun_out/r06e/k300_store_probe.txt
bash tools/r06_gemm_stagger_probe.sh > gpurun_out/r06e/gemm_stagger_probe.txt 2>&1; cat gpurun_out/r06e/gemm_stagger_probe.txt
