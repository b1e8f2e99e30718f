mkdir -p gpurun_out/r06b
for cfg in c2 c1 c4; do for s in 0 1 0 1; do
EBN_SIDE_STREAM=$s python bench.py --config $cfg --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg side=$s', d['ms_per_step'], d['ms_per_step_repeats'])"
done; done > gpurun_out/r06b/side_ab.txt 2>&1
cat gpurun_out/r06b/side_ab.txt
python -m pytest tests/test_full_size_parity.py tests/test_docvec_model.py tests/test_nrms_model.py -x -q -m gpu > gpurun_out/r06b/tests.log 2>&1; tail -5 gpurun_out/r06b/tests.log
