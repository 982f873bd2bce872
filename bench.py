#!/usr/bin/env python3
"""EM iterations/sec of the MI355X-native MMSBM core on BASELINE.json's headline config
(C3: 1M synthetic ratings, 100k users x 20k items, R=5, K=L=20, float64).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...          # starts its own N ranks (torch.distributed.run) as a child
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # what the driver does for N > 1

A step is ONE EM iteration (update_coefficients + the three normalisations,
src/mmsbm.py:244-250) over all N triples, device resident.  With N GPUs every rank runs its
own random restart (restart i on rank i; no data-path collective -> weak scaling) and the
job ends with one all-reduce (RCCL) that picks the maximum-likelihood restart.  Rank 0 prints
ONE JSON line.

`roofline` is for the dominant kernel: `achieved` = SURVEY 8(d)'s algorithmic bytes that launch
serves / its mean duration from HIP events on the library's stream; the `rocprofv3 --kernel-trace
--stats` duration and the PMC traffic of the same command come from profiles/ (tracked).
`cpu_baseline` is the numpy oracle (the reference's dense dataflow) timed on this host.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_MEASURED_GBPS = 6290.0  # measured float4 copy ceiling, same guide
INFINITY_CACHE_BYTES = 256 << 20
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (datasheet)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--profile-iters", type=int, default=20)
    ap.add_argument("--cpu-sample-rows", type=int, default=0,
                    help="rows of the workload the CPU baseline is timed on (0 = all of them where the "
                         "dense N x K x L oracle fits the time budget: C1-C3; C5: 60,000)")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batched-restarts", type=int, default=0,
                    help="after the timed region also report the rate with this many restarts per "
                         "GPU advancing as slots of one context (e.g. 8); never part of `value`.  Off by "
                         "default so that a kernel trace of the default command holds one-restart "
                         "launches only")
    ap.add_argument("--steady-steps", type=int, default=1000,
                    help="after the timed region, every rank also times this many iterations with HIP events "
                         "(`steady_state`: a 20-step region sits inside the ~1.7 ms clock ramp that follows an idle "
                         "period, DESIGN.md section 5); 0 = skip")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph instead of eager launches")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (use with --dist-backend gloo)")
    ap.add_argument("--no-collective-at-1", action="store_true",
                    help="with one GPU skip the one-rank process group (by default the end-of-job pick "
                         "goes through RCCL even at N=1, so that path is exercised on every run)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process
    (python -m torch.distributed.run), before anything in this process touches the GPU; the
    child's rank 0 prints the JSON line on the inherited stdout.  Never an exec."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def kernel_source_sha16():
    """Identity of the kernels a profile was taken with (profiles/pmc_summary.json records it): every
    source under mmsbm_amd/csrc, in name order -- the same digest the library is compiled with
    (mmsbm_hip_build_id)."""
    from mmsbm_amd.build import source_id
    return source_id()


def gather_ranks(mine, world, device=None):
    """[per-rank record] on every rank, in rank order: ONE all_gather of a fixed-size byte tensor (the record as
    JSON, padded) on the device the backend wants -- plain tensor collectives only, the path RCCL is built for; a
    list of one without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return [mine]
    width = 2048
    raw = json.dumps(mine).encode()
    if len(raw) > width:
        raise ValueError("per-rank record too long")
    buf = torch.zeros(width, dtype=torch.uint8)
    buf[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    buf = buf.to(device if device is not None else "cpu")
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return [json.loads(bytes(p.cpu().numpy().tobytes()).rstrip(b"\0").decode()) for p in parts]


def cpu_baseline(cfg, sample_rows, iters):
    """The oracle (numpy restatement of the reference's numpy backend: same dense N x K x L
    dataflow, one core) on the same workload -- all rows where that fits the budget (C3: ~3 s per
    iteration), else the first `sample_rows` scaled by rows.  `port_over_reference` is the ratio
    oracle time / real-reference time measured in the build container on the same arrays
    (scripts/calibrate_cpu_baseline.py -> oracle/calibration.json)."""
    import numpy as np
    from oracle import mmsbm_oracle as orc
    n, u, i, r, k, l = cfg
    train = orc.synthetic_triples(n, u, i, r, seed=0)
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    rows = n if sample_rows <= 0 else min(sample_rows, n)
    sub = train[:rows]
    d_u, d_i = orc.degrees(sub, n_u, n_i)
    theta, eta, pr = orc.init_params(orc.child_seeds(0, 1)[0], n_u, n_i, n_r, k, l, d_u, d_i)
    theta, eta, pr = orc.em_step(sub, theta, eta, pr, d_u, d_i)  # warm-up (page faults)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        theta, eta, pr = orc.em_step(sub, theta, eta, pr, d_u, d_i)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    full = dt * (n / rows)
    out = {"value": 1.0 / full, "unit": "it/s", "cores": 1, "kind": "port",
           "sample": (f"all {n} triples" if rows == n else
                      f"first {rows} of {n} triples (same U,I,R,K,L), scaled by rows (the dense "
                      f"N*K*L dataflow is linear in N)") +
                     f"; 1 warm-up + {iters} timed iterations, median {dt:.3f} s per iteration",
           "seconds_per_iteration": full, "host_cpus": os.cpu_count()}
    cal_path = os.path.join(ROOT, "oracle", "calibration.json")
    if os.path.exists(cal_path):
        with open(cal_path) as fh:
            cal = json.load(fh)
        out["port_over_reference"] = cal.get("port_over_reference")
        out["calibration"] = cal.get("note")
    return out


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
    import numpy as np
    from mmsbm_amd import HipEM
    with HipEM(train, model.user_groups, model.item_groups, model.p + 1, model.m + 1,
               model._dims["n_ratings"], device=device, slots=slots) as em:
        seeds = np.random.SeedSequence(0).spawn(slots)
        for s in range(slots):
            em.select(s).init_params(seeds[s])
        em.iterate(5)
        ms = min(em.time_iterations(iters) for _ in range(3))
    return {"slots": slots, "iterations": iters, "ms_per_step_all_slots": ms / iters,
            "us_per_restart_iteration": 1e3 * ms / iters / slots,
            "value": slots * iters / (ms * 1e-3), "unit": "restart-iterations/s on one GPU (HIP events)"}


def roofline_object(args, ctx, prof, n, k, l):
    """The dominant kernel against the HBM roofline, SURVEY 8(d)-based."""
    dom = max(prof, key=lambda nm: prof[nm][0] * prof[nm][1])
    dom_us, _, model_rd, _ = prof[dom]
    # 8(d): per triple three int32 indices + one theta row + one eta-side row.  Every triple-level
    # index load and row gather of the iteration happens in seg_pass (its two passes), so that
    # launch is charged all of N (12 + 8K + 8L); the K*L*R tile bytes belong to the pair stage.
    # Small problems run two launches per iteration (fused_small.hpp), one triple pass in each: each is charged
    # half of 8(d)'s bytes.
    if dom == "seg_pass_kernel":
        served, basis = n * (12 + 8 * k + 8 * l), ("SURVEY 8(d): N(12+8K+8L) -- all triple-level index loads and row "
                                                   "gathers of the iteration are in this launch")
    elif dom in ("pairs_fused_kernel", "tail_fused_kernel"):
        served, basis = n * (12 + 8 * k + 8 * l) // 2, ("SURVEY 8(d): N(12+8K+8L) / 2 -- the two launches of a small "
                                                        "problem's iteration hold one of the two triple passes each")
    else:
        served, basis = model_rd, "per-launch model (DESIGN.md section 4)"
    achieved = served / (dom_us * 1e-6) / 1e9
    resident = ctx.bytes_per_slot + 4 * (2 * n + 3 * ctx.n_pairs + ctx.n_users + ctx.n_items)
    out = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBPS,
           "frac_of_measured_copy_ceiling": achieved / HBM_MEASURED_GBPS,
           "algorithmic_bytes_per_launch": served,
           "algorithmic_bytes_basis": basis,
           "avg_launch_us": dom_us, "avg_launch_source": "HIP event pairs on the library's stream (bench.py, live)",
           "rocprof_avg_us": None, "rocprof_source": None,
           "algorithmic_bytes_model": model_rd,
           "achieved_model": model_rd / (dom_us * 1e-6) / 1e9,
           "resident_set_bytes": resident,
           "served_from": ("Infinity Cache / fabric: the iteration's resident set (%.0f MB) is below the 256 MiB "
                           "Infinity Cache, so `traffic` counts fabric requests that mostly hit it, not DRAM"
                           % (resident / 1e6)) if resident < INFINITY_CACHE_BYTES else
                          "HBM: the resident set (%.0f MB) exceeds the 256 MiB Infinity Cache" % (resident / 1e6),
           "traffic": None, "traffic_source": None}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path):  # rocprofv3 passes of this same command (scripts/profile_round.sh)
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        ent = pmc.get(args.config, {}).get(dom)
        meta = pmc.get("_meta", {}).get(args.config, {})
        if ent:
            if ent.get("avg_us") is not None:
                out["rocprof_avg_us"] = ent["avg_us"]
                out["rocprof_source"] = meta.get("stats_file")
            if meta.get("kernel_source_sha16") == kernel_source_sha16():
                out["traffic"], out["traffic_source"] = ent["hbm_bytes_per_launch"], ent["source"]
                out["l2_hit_rate"] = ent.get("l2_hit_rate")
            else:
                out["traffic_source"] = ("stale: profiles/pmc_summary.json was taken with other kernel sources "
                                         f"({meta.get('kernel_source_sha16')}); rerun scripts/profile_round.sh")
                out["rocprof_source"] = (out["rocprof_source"] or "") + " (older kernel sources)"
    return out


def main():
    args = parse_args()
    from mmsbm_amd.build import ensure_library   # (no HIP, no torch: safe before the ranks are started)
    ensure_library()                             # a fresh clone has no libmmsbm_hip.so yet
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # stdout carries exactly ONE line (rank 0's JSON): whatever libraries print while the job runs
    # (RCCL's version banner, for one) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np  # noqa: F401
    import torch  # (before the HIP library: one HIP runtime per process)
    import torch.distributed as dist
    from mmsbm_amd import restarts  # imports torch first, then the library
    from mmsbm_amd import MMSBM
    from mmsbm_amd.synthetic import CONFIGS, algorithmic_bytes, synthetic_triples

    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    # N > 1: the launcher's process group (RCCL).  N = 1: no group yet -- the one-rank group that lets the
    # end-of-job pick go through RCCL on every run is made AFTER the timed region (its barrier kernel and
    # proxy thread cost a 20-step run 2-3 us per step when they sit in front of it; a barrier among one
    # rank has nothing to wait for).
    rank, world, local, device = restarts.init_from_env(args.dist_backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    ctx.init_params(model.child_states[rank])  # restart `rank`, random start drawn on the device

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
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    # the binary that ran IS the one built from the sources in the tree (ensure_library rebuilds a stale one;
    # MMSBM_HIP_LIBRARY may point elsewhere): recorded, and `traffic` is dropped when it is not
    from mmsbm_amd import _lib
    build_id = _lib.build_id()
    # steady state: the same loop, long enough to leave the clock ramp behind (HIP events on the library's stream)
    steady_ms = ctx.time_iterations(args.steady_steps) / args.steady_steps if args.steady_steps > 0 else None

    # end of the job: likelihood of this rank's restart + ONE all-reduce to pick the best
    coll_error = None
    if world == 1 and not args.no_collective_at_1:
        try:
            restarts.init_from_env(args.dist_backend, force_init=True)
        except Exception as exc:  # a one-rank group is a nicety: report without it rather than not at all
            coll_error = f"{type(exc).__name__}: {exc}"
    lik = ctx.likelihood()
    best, best_lik, liks = restarts.pick_max_likelihood({rank: lik}, world, device)
    # who ran what: one record per rank (device identity from the library's own HIP runtime), gathered with
    # one all_gather_object -- so that a line from an 8-GPU node shows eight different PCI bus ids, each
    # rank's own time for the K steps and its restart's likelihood
    ident = _lib.device_identity(local)
    ranks = gather_ranks({"rank": rank, "local_rank": local, "device_index": local, "device_name": ident["name"],
                          "pci_bus_id": ident["pci_bus_id"], "compute_units": ident["compute_units"],
                          "hostname": socket.gethostname(), "pid": os.getpid(), "restart": rank,
                          "ms_per_step": 1000.0 * elapsed / args.steps,
                          "steady_ms_per_step": steady_ms, "likelihood": float(lik), "build_id": build_id}, world,
                         restarts._collective_device(device))

    out = None
    if rank == 0:
        its = world * args.steps / elapsed_max
        rd, wr = algorithmic_bytes(n, model.p + 1, model.m + 1, r, k, l)
        prof = ctx.profile_iterations(args.profile_iters)
        ev_ms = ctx.time_iterations(args.steps)
        coll = restarts.collective_info()
        if coll_error:
            coll["error"] = coll_error
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
                       "pairs": ctx.n_pairs,
                       "launches_per_iteration": int(ctx.get_option("launches")),
                       "pair_stage": {0.0: "vector ALUs (pair_block_kernel)", 1.0: "matrix cores (pair_mfma_kernel)",
                                      2.0: "matrix cores, blocked (mfma_rows_kernel + mfma_slab_kernel)"}[ctx.get_option("mfma")]},
            "collective": coll,
            "ranks": ranks,
            "distinct_devices": len({(r["hostname"], r["pci_bus_id"]) for r in ranks}),
            "steady_state": (None if steady_ms is None else
                             {"steps": args.steady_steps, "ms_per_step": max(r["steady_ms_per_step"] for r in ranks),
                              "value": world * 1000.0 / max(r["steady_ms_per_step"] for r in ranks), "unit": "it/s",
                              "frac_of_hbm_peak": rd * (1000.0 / max(r["steady_ms_per_step"] for r in ranks)) / 1e9 / HBM_PEAK_GBPS,
                              "source": "HIP events on the library's stream around `steps` iterations, after the "
                                        "timed region; max over ranks"}),
            "library": {"build_id": build_id, "source_id": kernel_source_sha16(),
                        "matches_sources": build_id == kernel_source_sha16()},
            "roofline": roofline_object(args, ctx, prof, n, k, l),
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
            rows = args.cpu_sample_rows
            if rows <= 0 and args.config == "c5":
                rows = 60_000  # the dense oracle needs 20 KB per row at K = L = 50
            out["cpu_baseline"] = cpu_baseline(cfg, rows, args.cpu_iters)
            out["gpu_over_cpu"] = its / out["cpu_baseline"]["value"]
    fence()
    if rank == 0:  # the line first: nothing that happens while the process group is torn down can lose it
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
