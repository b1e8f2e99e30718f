mkdir -p gpurun_out/r06i
for st in 0 4 8 12 16 24; do
EBN_NEWS_TAIL=1 EBN_NEWS_TAIL_STAGGER=$st python bench.py --config c2 --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 news_tail stagger=$st', d['ms_per_step'])"
done > gpurun_out/r06i/news_tail_stagger.txt 2>&1
EBN_NEWS_TAIL=0 python bench.py --config c2 --no-cpu-baseline --no-fit-loop --no-split-leg --no-probe --no-roofline --legs "" --steps 50 --repeats 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 three kernels', d['ms_per_step'])" >> gpurun_out/r06i/news_tail_stagger.txt
cat gpurun_out/r06i/news_tail_stagger.txt
