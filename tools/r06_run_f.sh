mkdir -p gpurun_out/r06f
for cfg in c2 c1 c4 c5 c3; do for flag in "" "--no-adam-in-finish" "" "--no-adam-in-finish"; do
python bench.py --config $cfg $flag --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg [$flag]', d['ms_per_step'], d['roofline_step'].get('launches_per_step'))"
done; done > gpurun_out/r06f/adam_in_finish_ab.txt 2>&1
cat gpurun_out/r06f/adam_in_finish_ab.txt
bash tools/r06_attn_occupancy_probe.sh > gpurun_out/r06f/attn_occupancy_probe.txt 2>&1; cat gpurun_out/r06f/attn_occupancy_probe.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r06f/tests.log 2>&1; tail -5 gpurun_out/r06f/tests.log
