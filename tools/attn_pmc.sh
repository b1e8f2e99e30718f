#!/bin/bash
# L2 <-> memory request counters of the attention backward (tools/tail_probe.py, eager warm-up launches are the ones counted).
#   gpurun -- 'bash tools/attn_pmc.sh <tag> <n_titles> [variant.so]'   (EBN_ATTN_BWD_PER_WAVE is passed through)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-attn_pmc}; n=${2:-3200}
[ -n "$3" ] && export EBNERD_HIP_LIB=$PWD/ebnerd-benchmark_amd/csrc/variants/$3
out=gpurun_out/$tag
mkdir -p $out
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o b -- python tools/tail_probe.py $n 30 a > $out/p$i.log 2>&1
  rm -f $out/p$i/*kernel_trace.csv $out/p$i/*agent_info.csv
done
python - <<'P' "$out"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" in k and "bwd" in k:
            acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:44s} {sum(v) / len(v):16.1f}  (n={len(v)})")
P
