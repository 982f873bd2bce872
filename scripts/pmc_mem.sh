#!/usr/bin/env bash
# Memory-path counters of the iteration's kernels (TLB, L1 stalls, fabric request sizes, DRAM share).
# usage (on the GPU box through gpurun): scripts/pmc_mem.sh <tag> <config>
# (no TA_* set: a pass with TA_TA_BUSY_sum / TA_*_STALLED_BY_TC_CYCLES_sum hung rocprofv3 on this pool)
set -u
tag=${1:-mem}
config=${2:-c3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
short="python3 $root/bench.py --config $config --steps 20 --warmup 2 --no-cpu-baseline --profile-iters 2 --steady-steps 0 --no-collective-at-1"
i=0
for set in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pmcmem_${tag}_$i" -- $short > "$out/pmcmem_${tag}_$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pmcmem_${tag}_$i.log"; }
done
echo done
