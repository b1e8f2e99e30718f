#!/bin/bash
# SQ / cache counters of the LDS-free GEMM on the c2 AttLayer2 shapes (the probe's eager warm-up launches are what is counted)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/r04d
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TCC|TA|TD)_[A-Z0-9_]+(_sum)?\b" | sort -u > $out/counters_available.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  for depth in 2; do
    EBN_GEMM_DIRECT_DEPTH=$depth timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p${i}_d$depth -o b -- python tools/direct_gemm_probe.py 24000 > $out/p${i}_d$depth.log 2>&1
    rm -f $out/p${i}_d$depth/*kernel_trace.csv $out/p${i}_d$depth/*agent_info.csv
  done
done
# the LDS-staged 32x32 big kernel on the same shapes for comparison (EBN_GEMM_DIRECT=0: tall16 / 128x128)
EBN_GEMM_DIRECT=0 timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1_lds -o b -- python tools/direct_gemm_probe.py 24000 > $out/p1_lds.log 2>&1
rm -f $out/p1_lds/*kernel_trace.csv $out/p1_lds/*agent_info.csv
python - <<'P' "$out"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    tag = f.split("/")[2]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" in k:
            acc[(tag.split("_", 1)[1], k.split("(")[0][-48:])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {sum(v) / len(v):16.1f}  (n={len(v)})")
P
tail -3 $out/p3_d2.log $out/p5_d2.log
