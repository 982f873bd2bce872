#!/usr/bin/env bash
# Memory-path counters of the iteration's kernels (TLB, L1 stalls, fabric request sizes, DRAM share).
# usage (on the GPU box through gpurun): scripts/pmc_mem.sh <tag> <config> [extra bench.py arguments, e.g. "--groups 16"]
# Every pass runs under its own `timeout`: a counter set the hardware cannot collect in one pass makes rocprofv3 abort
# inside rocprofiler_create_counter_config at the FIRST dispatch of the process ("error code 38: Request exceeds the
# capabilities of the hardware to collect"), and its signal handler then waits for that dispatch for ever -- this is
# what the "hung" TA_* pass of round 2 and the GRBM_* pass of round 3 were (profiles/README.md); neither set is asked
# for any more.
set -u
tag=${1:-mem}
config=${2:-c3}
extra=${3:-}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
short="python3 $root/bench.py --config $config --steps 20 --warmup 2 --no-cpu-baseline --profile-iters 2 --steady-steps 0 --no-collective-at-1 $extra"
i=0
for set in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pmcmem_${tag}_$i" -- $short > "$out/pmcmem_${tag}_$i.log" 2>&1 || { echo "pass $i failed"; grep -m1 "error code" "$out/pmcmem_${tag}_$i.log"; }
  echo "pass $i done"
done
echo done
