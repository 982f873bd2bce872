#!/usr/bin/env bash
# Everything profiles/ holds for one round, ON THE GPU BOX (through gpurun): kernel trace + counter passes of every
# bench config, then the memory-path passes for C3 and C5.   usage: scripts/profile_all.sh r3
set -u
round=${1:-r3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
scripts/profile_round.sh ${round}_c3 c3 200 && echo "c3 trace+pmc done"
scripts/profile_round.sh ${round}_c2 c2 200 && echo "c2 trace+pmc done"
scripts/profile_round.sh ${round}_c5 c5 50 && echo "c5 trace+pmc done"
scripts/pmc_mem.sh ${round}c3 c3
scripts/pmc_mem.sh ${round}c5 c5
echo all done
