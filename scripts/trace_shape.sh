#!/usr/bin/env bash
# rocprofv3 kernel trace of scripts/shape_time.py at a given shape (run on the GPU box through gpurun):
#   scripts/trace_shape.sh N U I R K L [zipf|lognorm]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf "$out/trace_shape"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_shape" -- python3 $root/scripts/shape_time.py "$@" > "$out/trace_shape.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace_shape/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(f"{float(r['AverageNs']) / 1000:10.1f} us x{r['Calls']:>5}  {r['Percentage']:>6}%  {r['Name'][:100]}")
PY
