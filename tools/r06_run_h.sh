mkdir -p gpurun_out/r06h
python -m pytest tests/test_nrms_model.py -k "news_tail or adam_inside" tests/test_docvec_model.py -k "finale or fixed_point or news_tail or adam_inside" -x -q -m gpu > gpurun_out/r06h/tests.log 2>&1; tail -4 gpurun_out/r06h/tests.log
for cfg in c2 c5 c4; do for t in 0 1 0 1; do
EBN_NEWS_TAIL=$t python bench.py --config $cfg --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg news_tail=$t', d['ms_per_step'], d['roofline_step'].get('launches_per_step'))"
done; done > gpurun_out/r06h/news_tail_ab.txt 2>&1
cat gpurun_out/r06h/news_tail_ab.txt
cd /tmp && export TMPDIR=/tmp && EBN_NEWS_TAIL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06h/stats_tail -o c2 -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-fit-loop --no-split-leg --legs "" > /dev/null 2>&1; rm -f $GRAFT_REPO_ROOT/gpurun_out/r06h/stats_tail/*/*kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/r06h/stats_tail/*/*agent_info.csv
