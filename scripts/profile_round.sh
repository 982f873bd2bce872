#!/usr/bin/env bash
# rocprofv3 evidence for one round, run ON THE GPU BOX (through gpurun):
#   scripts/profile_round.sh r1
# kernel trace + stats of the default bench command, then separate --pmc passes (never
# combined with trace domains other than --kernel-trace).  Raw output -> gpurun_out/prof_<tag>_*;
# scripts/summarize_profile.py turns it into the small files committed under profiles/.
set -u
tag=${1:-r1}
config=${2:-c3}
steps=${3:-200}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="python3 $root/bench.py --config $config --steps $steps --warmup 20 --no-cpu-baseline"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_trace" -- $bench > "$out/prof_${tag}_trace.log" 2>&1 || exit 1
short="python3 $root/bench.py --config $config --steps 20 --warmup 2 --no-cpu-baseline --profile-iters 2 --steady-steps 0"
timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/prof_${tag}_fetch" -- $short > "$out/prof_${tag}_fetch.log" 2>&1 || exit 1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/prof_${tag}_write" -- $short > "$out/prof_${tag}_write.log" 2>&1 || exit 1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/prof_${tag}_l2" -- $short > "$out/prof_${tag}_l2.log" 2>&1 || exit 1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d "$out/prof_${tag}_sq" -- $short > "$out/prof_${tag}_sq.log" 2>&1 || exit 1
# LDS pipe and stall attribution
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$out/prof_${tag}_lds_a" -- $short > "$out/prof_${tag}_lds_a.log" 2>&1 || exit 1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out/prof_${tag}_lds_b" -- $short > "$out/prof_${tag}_lds_b.log" 2>&1 || exit 1
# matrix-core use (the K x L > 1024 pair stage); counter names may be missing on some stacks: not fatal
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d "$out/prof_${tag}_mfma" -- $short > "$out/prof_${tag}_mfma.log" 2>&1 || echo "mfma counters unavailable"
echo done
