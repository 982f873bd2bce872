"""Eager launches against hipGraph replays (two iterations per graph) per EM iteration.  usage: graph_time.py [c1|c2|c3 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmsbm_amd import MMSBM
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
for tag in sys.argv[1:] or ["c1", "c2", "c3"]:
    n, u, i, r, k, l = CONFIGS[tag]
    train = synthetic_triples(n, u, i, r, 0)
    mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
    ctx = mm._ctx(0); ctx.init_params(mm.child_states[0])
    iters = 200 if tag == "c3" else 1000
    line = f"{tag}: {int(ctx.get_option('launches'))} launches per iteration;"
    for graph in (0, 1, 0, 1):
        ctx.set_graph_mode(graph)
        ctx.iterate(20)
        us = min(ctx.time_iterations(iters) for _ in range(3)) * 1000 / iters
        line += f"  {'graph' if graph else 'eager'} {us:7.2f} us"
    print(line, flush=True)
