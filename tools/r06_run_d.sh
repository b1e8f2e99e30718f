mkdir -p gpurun_out/r06d
python -m pytest tests/test_docvec_model.py tests/test_full_size_parity.py -k "docvec or c3 or finale" -x -q -m gpu > gpurun_out/r06d/tests.log 2>&1; tail -5 gpurun_out/r06d/tests.log
for i in 1 2; do
python bench.py --config c3 --no-cpu-baseline --no-probe --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['avg_launch_us'], d['roofline_step']['launches_per_step'])"
done
bash tools/r06_k300_store_probe.sh > gpurun_out/r06d/k300_store_probe.txt 2>&1; cat gpurun_out/r06d/k300_store_probe.txt
