#!/bin/bash
# Run on the GPU box (via gpurun): the K1-shaped load floor probe (tools/micro/k1_probe) plain and under rocprofv3, then memory-system
# counters (separate --pmc passes, counters only) for the probe's kernels AND for the real spatial kernel of a short bench.py run.
# usage: tools/r05_k1_probe.sh <tag>
set -u
TAG=${1:-r05a}
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out"
P="$REPO/tools/micro/k1_probe"
[ -x "$P" ] || hipcc --offload-arch=gfx950 -O3 -o "$P" "$REPO/tools/micro/k1_probe.hip"
timeout 300 "$P" 24 > "$REPO/gpurun_out/${TAG}_k1_probe_events.txt" 2>&1
cd /tmp && rm -rf /tmp/kp_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kp_$TAG -o kp -- "$P" 24 > /dev/null 2> "$REPO/gpurun_out/${TAG}_k1_probe_prof.err"
cd "$REPO"
python tools/prof_summary.py "$(find /tmp/kp_$TAG -name '*.db' | head -1)" "gpurun_out/${TAG}_k1_probe_rocprofv3.md"
PASSES=("TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
        "TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
        "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY")
i=0
DBS_P=(); DBS_B=()
for CNT in "${PASSES[@]}"; do
    i=$((i + 1))
    cd /tmp && rm -rf /tmp/kpp_$i /tmp/kpb_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $CNT -d /tmp/kpp_$i -o x -- "$P" 6 > /dev/null 2> "$REPO/gpurun_out/${TAG}_pmc_probe_$i.err"
    timeout 600 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex sttm -d /tmp/kpb_$i -o x -- \
        python "$REPO/bench.py" --steps 2 --warmup 1 --videos-per-step 64 --profile-calls 8 --no-cpu-baseline --no-extensions --no-configs > /dev/null 2> "$REPO/gpurun_out/${TAG}_pmc_bench_$i.err"
    cd "$REPO"
    D=$(find /tmp/kpp_$i -name '*.db' | head -1); [ -n "$D" ] && cp "$D" gpurun_out/${TAG}_pp_$i.db && DBS_P+=(gpurun_out/${TAG}_pp_$i.db)
    D=$(find /tmp/kpb_$i -name '*.db' | head -1); [ -n "$D" ] && cp "$D" gpurun_out/${TAG}_pb_$i.db && DBS_B+=(gpurun_out/${TAG}_pb_$i.db)
done
python tools/pmc_table.py gpurun_out/${TAG}_k1_probe_pmc.md "${DBS_P[@]}" > /dev/null
python tools/pmc_table.py gpurun_out/${TAG}_k1_real_pmc.md "${DBS_B[@]}" > /dev/null
rm -f gpurun_out/${TAG}_pp_*.db gpurun_out/${TAG}_pb_*.db
cat gpurun_out/${TAG}_k1_probe_events.txt
