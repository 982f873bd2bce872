"""Rows beyond 1,024 groups (seg_wide_kernel) against the widest group-of-lanes instantiation: per-stage times at 1M ratings."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mmsbm_amd import HipEM
from mmsbm_amd.synthetic import synthetic_triples
train = synthetic_triples(1_000_000, 100_000, 20_000, 5, 0)
for k, l in ((1024, 3), (1040, 3), (1500, 3), (2048, 3), (3, 1040), (3, 2048)):
    with HipEM(train, k, l, device=0) as em:
        em.init_params(np.random.SeedSequence(0).spawn(1)[0])
        em.iterate(2)
        it = min(em.time_iterations(5) for _ in range(2)) * 200
        st = [min(em.time_stage(s, 3) for _ in range(2)) for s in range(4)]
        print(f"K={k} L={l}: iteration {it:9.1f} us  seg {st[0]:9.1f}  T+S {st[1]:8.1f}  eta_p {st[2]:8.1f}  A {st[3]:8.1f}", flush=True)
