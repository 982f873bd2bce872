#!/usr/bin/env python3
"""EM iterations/sec of the MI355X-native MMSBM core on BASELINE.json's headline config
(C3: 1M synthetic ratings, 100k users x 20k items, R=5, K=L=20, float64).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is ONE EM iteration (update_coefficients + the three normalisations,
src/mmsbm.py:244-250) over all N triples, device resident.  With N GPUs every rank runs its
own random restart (restart i on rank i; no data-path collective -> weak scaling) and the
job ends with one all-reduce that picks the maximum-likelihood restart.  Rank 0 prints ONE
JSON line.  `roofline` is for the dominant kernel (HIP events on the library's stream);
`cpu_baseline` is the numpy oracle (same dense dataflow as the reference) timed on this
host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_MEASURED_GBPS = 6290.0  # measured float4 copy ceiling, same guide


def cpu_baseline(cfg, sample_rows, iters):
    """The oracle (numpy restatement of the reference's numpy backend) on the first
    `sample_rows` triples of the same workload, 1 core; scaled to full-size it/s by rows."""
    from oracle import mmsbm_oracle as orc
    n, u, i, r, k, l = cfg
    train = orc.synthetic_triples(n, u, i, r, seed=0)
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    sub = train[:sample_rows]
    d_u, d_i = orc.degrees(sub, n_u, n_i)
    theta, eta, pr = orc.init_params(orc.child_seeds(0, 1)[0], n_u, n_i, n_r, k, l, d_u, d_i)
    theta, eta, pr = orc.em_step(sub, theta, eta, pr, d_u, d_i)  # warm-up (page faults)
    t0 = time.perf_counter()
    for _ in range(iters):
        theta, eta, pr = orc.em_step(sub, theta, eta, pr, d_u, d_i)
    dt = (time.perf_counter() - t0) / iters
    full = dt * (n / sample_rows)
    return {"value": 1.0 / full, "unit": "it/s", "cores": 1, "kind": "port",
            "sample": f"first {sample_rows} of {n} triples (same U,I,R,K,L), {iters} timed iterations "
                      f"after 1 warm-up, {dt:.3f} s/iteration on the sample, scaled by rows "
                      f"(dense N*K*L dataflow is linear in N)",
            "host_cpus": os.cpu_count()}


FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (MI355X_MICROARCH.md)


def fp64_valu(n, q, k, l, its_per_gpu):
    device = 8.0 * n * k + 6.0 * q * k * l
    dense = 6.0 * n * k * l
    return {"device_flops_per_iter": device, "device_tflops": device * its_per_gpu / 1e12,
            "frac_of_peak": device * its_per_gpu / 1e12 / FP64_VALU_PEAK_TFLOPS,
            "reference_dataflow_flops_per_iter": dense,
            "reference_dataflow_equiv_tflops": dense * its_per_gpu / 1e12,
            "peak_tflops": FP64_VALU_PEAK_TFLOPS}


def batched_rate(model, train, device, slots, iters):
    """`slots` restarts of the same training set advance with one set of
    launches.  Reported beside the headline, never as `value` (whose config is sampling=1)."""
    from mmsbm_amd import HipEM
    with HipEM(train, model.user_groups, model.item_groups, model.p + 1, model.m + 1,
               model._dims["n_ratings"], device=device, slots=slots) as em:
        d_u, d_i = em.degrees()
        seeds = np.random.SeedSequence(0).spawn(slots)
        for s in range(slots):
            em.select(s).set_params(*model.init_params(seeds[s], d_u, d_i))
        em.iterate(5)
        ms = min(em.time_iterations(iters) for _ in range(3))
    return {"slots": slots, "iterations": iters, "ms_per_step_all_slots": ms / iters,
            "value": slots * iters / (ms * 1e-3), "unit": "restart-iterations/s on one GPU (HIP events)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--profile-iters", type=int, default=20)
    ap.add_argument("--cpu-sample-rows", type=int, default=300_000)
    ap.add_argument("--cpu-iters", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batched-restarts", type=int, default=0,
                    help="after the timed region also report the rate with this many restarts per "
                         "GPU advancing as slots of one context (e.g. 8); never part of `value`.  Off by "
                         "default so that a kernel trace of the default command holds one-restart "
                         "launches only")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph instead of eager launches")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (use with --dist-backend gloo)")
    args = ap.parse_args()

    from mmsbm_amd import restarts  # imports torch first, then the library
    from mmsbm_amd import MMSBM
    from mmsbm_amd.synthetic import CONFIGS, algorithmic_bytes, synthetic_triples

    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local, device = restarts.init_from_env(args.dist_backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
    if device.type != "cuda":
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")

    cfg = CONFIGS[args.config]
    n, u, i, r, k, l = cfg
    train = synthetic_triples(n, u, i, r, seed=0)
    model = MMSBM(k, l, iterations=args.steps, sampling=world, seed=0, backend="hip")
    model._prepare_objects(train)
    restarts.check_single_hip_runtime()
    ctx = model._ctx(local)
    ctx.set_graph_mode(1 if args.graph else 0)
    d_u, d_i = ctx.degrees()
    ctx.set_params(*model.init_params(model.child_states[rank], d_u, d_i))  # restart `rank`

    ctx.iterate(args.warmup)

    def fence():
        restarts.barrier(device)
        ctx.synchronize()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    ctx.iterate(args.steps, sync=False)
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=restarts._collective_device(device))
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())

    # end of the job: likelihood of this rank's restart + ONE all-reduce to pick the best
    lik = ctx.likelihood()
    best, best_lik, liks = restarts.pick_max_likelihood({rank: lik}, world, device)

    out = None
    if rank == 0:
        its = world * args.steps / elapsed_max
        rd, wr = algorithmic_bytes(n, model.p + 1, model.m + 1, r, k, l)
        prof = ctx.profile_iterations(args.profile_iters)
        dom = max(prof, key=lambda nm: prof[nm][0] * prof[nm][1])
        dom_us, dom_launches, dom_rd, dom_wr = prof[dom]
        achieved = dom_rd / (dom_us * 1e-6) / 1e9
        ev_ms = ctx.time_iterations(args.steps)
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc_path):  # rocprofv3 --pmc passes of this same command (scripts/profile_round.sh)
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            ent = pmc.get(args.config, {}).get(dom)
            if ent:
                traffic, traffic_src = ent["hbm_bytes_per_launch"], ent["source"]
        out = {
            "metric": "EM iterations/sec (1M ratings, K=L=20)" if args.config == "c3"
                      else f"EM iterations/sec ({args.config})",
            "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config.upper()}: {n} synthetic ratings, {model.p + 1} users x "
                                   f"{model.m + 1} items, R={r}, K={k}, L={l}, one restart per GPU "
                                   f"(sampling={world}), uniform generator seed 0, model seed 0",
                       "launch": "hipGraph replay" if args.graph else "eager",
                       "pairs": ctx.n_pairs},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "frac_of_measured_copy_ceiling": achieved / HBM_MEASURED_GBPS,
                         "algorithmic_bytes_per_launch": dom_rd, "avg_launch_us": dom_us,
                         "traffic": traffic, "traffic_source": traffic_src},
            "iteration": {"algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr,
                          "achieved_gbps_per_gpu": rd * (its / world) / 1e9,
                          "frac_of_hbm_peak": rd * (its / world) / 1e9 / HBM_PEAK_GBPS,
                          "device_ms_per_step_events": ev_ms / args.steps,
                          # secondary bound: fp64 vector ALU.  Flops the DEVICE does
                          # (factorised form: 8NK in the two passes + 6QKL in the pair stage) and
                          # the reference dataflow's 6KL per triple, against the 78.6 TFLOP/s peak
                          "fp64_valu": fp64_valu(n, ctx.n_pairs, k, l, its / world)},
            "kernels_us": {nm: {"avg_us": v[0], "launches_per_iter": v[1],
                                "gbps": (v[2] / (v[0] * 1e-6) / 1e9) if v[0] > 0 else None}
                           for nm, v in prof.items()},
            "likelihoods": [float(x) for x in liks], "best_restart": best,
        }
        if args.batched_restarts > 1:
            out["batched_restarts"] = batched_rate(model, train, local, args.batched_restarts,
                                                   max(20, args.steps // 10))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, min(args.cpu_sample_rows, n), args.cpu_iters)
            out["gpu_over_cpu"] = its / out["cpu_baseline"]["value"]
    fence()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
