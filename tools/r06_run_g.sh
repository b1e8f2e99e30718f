mkdir -p gpurun_out/r06g
bash tools/r06_attn_occupancy_probe.sh > gpurun_out/r06g/attn_occupancy_probe.txt 2>&1; cat gpurun_out/r06g/attn_occupancy_probe.txt
python -m pytest tests -q -m gpu > gpurun_out/r06g/tests.log 2>&1; tail -8 gpurun_out/r06g/tests.log
