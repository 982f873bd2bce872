#!/usr/bin/env bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf "$out/trace_shape"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_shape" -- python3 $root/scripts/shape_time.py "$@" > "$out/trace_shape.log" 2>&1
cat $out/trace_shape/*/*kernel_stats.csv | cut -c1-160 | head -12
