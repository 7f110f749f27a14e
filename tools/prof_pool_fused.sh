#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats + SQ / TCP counters of the pooled-input spatial kernel against pool2d + the plain spatial kernel.
TAG=${1:-r05p}
REPO=$(pwd); export TMPDIR=/tmp; mkdir -p "$REPO/gpurun_out"
cd /tmp && rm -rf /tmp/ppf_$TAG
ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ppf_$TAG -o x -- python "$REPO/tools/bench_pool_fused.py" > /dev/null 2>&1
cd "$REPO"; python tools/prof_summary.py "$(find /tmp/ppf_$TAG -name '*.db' | head -1)" gpurun_out/${TAG}_pool_fused_kernels.md
i=0; DBS=()
for CNT in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum"; do
  i=$((i+1)); cd /tmp && rm -rf /tmp/ppfc_$i
  ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex "sttm" -d /tmp/ppfc_$i -o x -- python "$REPO/tools/bench_pool_fused.py" > /dev/null 2>&1
  cd "$REPO"; D=$(find /tmp/ppfc_$i -name '*.db' | head -1); [ -n "$D" ] && cp "$D" gpurun_out/${TAG}_c$i.db && DBS+=(gpurun_out/${TAG}_c$i.db)
done
python tools/pmc_table.py gpurun_out/${TAG}_pool_fused_pmc.md "${DBS[@]}" > /dev/null
rm -f gpurun_out/${TAG}_c*.db
cat gpurun_out/${TAG}_pool_fused_pmc.md | cut -c1-400
