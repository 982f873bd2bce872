#!/usr/bin/env bash
# round 4: the two-launch iteration with split segments -- parity, then the MovieLens-100k shape's time
set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_launch or heavy_tailed or dense_data or skewed or xcd_local or slots_are" > gpurun_out/r4_split_tests.log 2>&1 || { tail -40 gpurun_out/r4_split_tests.log; exit 1; }
tail -3 gpurun_out/r4_split_tests.log
python scripts/movielens_shape.py > gpurun_out/r4_ml.log 2>&1
MMSBM_HIP_NO_FUSED_SPLIT=1 python scripts/movielens_shape.py > gpurun_out/r4_ml_nosplit.log 2>&1
cat gpurun_out/r4_ml.log gpurun_out/r4_ml_nosplit.log
