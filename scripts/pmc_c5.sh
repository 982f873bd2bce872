#!/usr/bin/env bash
# Counter passes for the dense kernels at C5 (run on the GPU box through gpurun).
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
short="python3 $root/bench.py --config ${1:-c5} --steps 6 --warmup 2 --no-cpu-baseline --profile-iters 2 --steady-steps 0"
rm -rf "$out/pmc_a" "$out/pmc_b" "$out/pmc_c"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_a" -- $short > "$out/pmc_a.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$out/pmc_b" -- $short > "$out/pmc_b.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d "$out/pmc_c" -- $short > "$out/pmc_c.log" 2>&1 || exit 1
echo done
