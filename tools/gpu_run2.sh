set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r02c}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/$tag/pytest.log
for c in c2 c1 c3 c4 c5; do
  timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/bench_$c.json 2> gpurun_out/$tag/bench_$c.err
done
bash tools/profile_configs.sh $tag c1 c3 c2 > /dev/null 2>&1
tail -3 gpurun_out/$tag/pytest.log
