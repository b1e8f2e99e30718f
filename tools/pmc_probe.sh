#!/bin/bash
# SQ counters of the tail-kernel probe (kernel by kernel: the probe's graph replays are not counted, its eager warm-up
# launches are).  gpurun -- 'bash tools/pmc_probe.sh <tag> <which>'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-pmc}; which=${2:-a}
out=gpurun_out/$tag
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o b -- python tools/tail_probe.py 800 30 $which > $out/p$i.log 2>&1
  rm -f $out/p$i/*kernel_trace.csv $out/p$i/*agent_info.csv
done
python - <<'P' "$out"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" in k or "gemm" in k:
            acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")
P
