#!/usr/bin/env bash
# A/B of build variants of the two-launch path on ONE box (box-to-box spread is larger than the differences looked for):
#   scripts/ab_fused.sh "<flags A>" "<flags B>" ...     AB_CONFIGS="c2 c1" picks the problems (two and four launches),
#   AB_SHAPES="c3:8 3000000,300000,60000,5,20,20" times shapes[:slots] on the library's own path (scripts/iter_time.py)
set -u
i=0
for flags in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flags -o /tmp/libab$i.so mmsbm_amd/csrc/unity.hip &
  i=$((i+1))
  [ $((i % 6)) -eq 0 ] && wait
done
wait
for rep in 1 2 3; do
  i=0
  for flags in "$@"; do
    [ -f /tmp/libab$i.so ] || { echo "variant $i did not build"; exit 1; }
    echo "== variant $i [$flags] rep $rep"
    if [ -n "${AB_SHAPES:-}" ]; then MMSBM_HIP_LIBRARY=/tmp/libab$i.so python scripts/iter_time.py $AB_SHAPES 2>&1
    else MMSBM_HIP_LIBRARY=/tmp/libab$i.so python scripts/slots_time.py ${AB_CONFIGS:-c2} 2>&1 | grep "slots= 1 " | sed "s/us per restart-iteration//g"; fi
    i=$((i+1))
  done
done
