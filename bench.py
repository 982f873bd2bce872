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
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)   # internal: one process of the CPU baseline
    ap.add_argument("--batched-restarts", type=int, default=0,
                    help="after the timed region also report the rate with this many restarts per "
                         "GPU advancing as slots of one context (e.g. 8); never part of `value`.  Off by "
                         "default so that a kernel trace of the default command holds one-restart "
                         "launches only")
    ap.add_argument("--steady-steps", type=int, default=1000,
                    help="after the timed region, every rank also times this many iterations with HIP events "
                         "(`steady_state`: a 20-step region sits inside the ~1.7 ms clock ramp that follows an idle "
                         "period, DESIGN.md section 5); 0 = skip")
    ap.add_argument("--groups", type=int, default=0,
                    help="K = L override for the config's data (scripts/grid_report.py: the K sweep at C3's size); "
                         "one GPU only; 0 = the config's own")
    ap.add_argument("--mfma", type=int, default=-1, choices=[-1, 0, 1],
                    help="pair stage of big rating tiles: 1 matrix cores, 0 vector ALUs, -1 the library's choice "
                         "(scripts/grid_report.py: both forms of C5 from one run of the script)")
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
    if not args.no_cpu_baseline and CPU_BASELINE_ENV not in env:
        # the reference's side of an N-GPU number: `sampling` = N processes, one restart each (src/mmsbm.py:182-185),
        # timed HERE -- this parent never touches the GPU -- before the ranks exist, and handed to rank 0
        env[CPU_BASELINE_ENV] = json.dumps(cpu_baseline_processes(args, args.gpus))
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


CPU_BASELINE_ENV = "MMSBM_BENCH_CPU_BASELINE"


def cpu_rows(args, procs=1):
    """Rows of the workload one CPU process iterates over: all of them where the dense N x K x L oracle fits the
    time budget and the host's memory (C1-C3), else a bounded sample (C5: 60,000) -- scaled by rows afterwards."""
    from mmsbm_amd.synthetic import CONFIGS
    n, _, _, _, k, l = CONFIGS[args.config]
    rows = args.cpu_sample_rows if args.cpu_sample_rows > 0 else (60_000 if args.config == "c5" else n)
    rows = min(rows, n)
    try:   # `procs` processes at once, ~2.5 dense (rows, K, L) float64 tensors each: stay inside half of what is free
        import psutil
        fit = int(0.5 * psutil.virtual_memory().available / procs / (2.5 * 8 * k * l))
        rows = min(rows, max(1000, fit))
    except Exception:  # noqa: BLE001
        pass
    return rows


def cpu_worker(spec):
    """One process of the CPU baseline (`--cpu-worker` = JSON {config, rows, iters, restart, procs, dir}): the
    oracle's EM iteration for restart `restart` on the config's triples.  After the warm-up iteration the
    processes wait for each other (a file per process in `dir`), so that the timed iterations of all of them
    overlap like the reference's Pool(processes=sampling) workers do.  Imports numpy and the oracle only."""
    import numpy as np
    from oracle import mmsbm_oracle as orc
    from mmsbm_amd.synthetic import CONFIGS
    spec = json.loads(spec)
    n, u, i, r, k, l = CONFIGS[spec["config"]]
    train = orc.synthetic_triples(n, u, i, r, seed=0)
    n_u, n_i, n_r = (int(train[:, j].max()) + 1 for j in range(3))
    sub = train[:spec["rows"]]
    d_u, d_i = orc.degrees(sub, n_u, n_i)
    seed = orc.child_seeds(0, spec["procs"])[spec["restart"]]      # restart i of sampling = procs, model seed 0
    theta, eta, pr = orc.init_params(seed, n_u, n_i, n_r, k, l, d_u, d_i)
    theta, eta, pr = orc.em_step(sub, theta, eta, pr, d_u, d_i)    # warm-up (page faults)
    open(os.path.join(spec["dir"], f"ready.{spec['restart']}"), "w").close()
    deadline = time.time() + 600
    while len([f for f in os.listdir(spec["dir"]) if f.startswith("ready.")]) < spec["procs"]:
        if time.time() > deadline:
            raise SystemExit("cpu worker: the other processes never got ready")
        time.sleep(0.01)
    times = []
    t_begin = time.time()
    for _ in range(spec["iters"]):
        t0 = time.perf_counter()
        theta, eta, pr = orc.em_step(sub, theta, eta, pr, d_u, d_i)
        times.append(time.perf_counter() - t0)
    print(json.dumps({"restart": spec["restart"], "times": times, "begin": t_begin, "end": time.time(),
                      "checksum": float(np.sum(theta))}))


def cpu_baseline_processes(args, procs):
    """The CPU restatement on `procs` processes, one restart each -- what the reference's
    `Pool(processes=sampling)` (src/mmsbm.py:182-185) does with sampling = procs -- as child processes that
    never load torch or the HIP library.  value = procs iterations / the slowest process's median iteration."""
    import tempfile
    import numpy as np
    from mmsbm_amd.synthetic import CONFIGS
    n = CONFIGS[args.config][0]
    rows = cpu_rows(args, procs)
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    with tempfile.TemporaryDirectory(prefix="mmsbm_cpu_") as tmp:
        kids = []
        for j in range(procs):
            spec = {"config": args.config, "rows": rows, "iters": args.cpu_iters, "restart": j, "procs": procs, "dir": tmp}
            kids.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps(spec)],
                                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [kid.communicate() for kid in kids]
    recs = []
    for kid, (so, se) in zip(kids, outs):
        if kid.returncode != 0:
            raise RuntimeError("cpu baseline worker failed: " + se[-2000:])
        recs.append(json.loads(so.strip().splitlines()[-1]))
    per = [float(np.median(rec["times"])) for rec in recs]
    slowest = max(per) * (n / rows)
    overlap = min(rec["end"] for rec in recs) - max(rec["begin"] for rec in recs)
    out = {"value": procs / slowest, "unit": "it/s", "cores": procs, "kind": "port",
           "sample": (f"all {n} triples" if rows == n else f"first {rows} of {n} triples (same U,I,R,K,L), scaled by rows") +
                     f" in each of {procs} processes (one restart each, started together: the reference's "
                     f"Pool(processes=sampling), src/mmsbm.py:182-185); 1 warm-up + {args.cpu_iters} timed iterations, "
                     f"value = {procs} / slowest process's median iteration ({max(per):.3f} s; fastest {min(per):.3f} s)",
           "seconds_per_iteration": slowest, "per_process_seconds": [p_ * (n / rows) for p_ in per],
           "timed_regions_overlap_s": overlap, "host_cpus": os.cpu_count()}
    return add_calibration(out)


def add_calibration(out):
    cal_path = os.path.join(ROOT, "oracle", "calibration.json")
    if os.path.exists(cal_path):
        with open(cal_path) as fh:
            cal = json.load(fh)
        out["port_over_reference"] = cal.get("port_over_reference")
        out["calibration"] = cal.get("note")
    return out


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
    return add_calibration(out)


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


def _config_groups(config):
    from mmsbm_amd.synthetic import CONFIGS
    k, l = CONFIGS[config][4:6]
    return (k,) if k == l else ()


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
    # (the profiles are of the configs themselves: not of a K = L override, not of a forced pair-stage form)
    same_command = getattr(args, "mfma", -1) < 0 and getattr(args, "groups", 0) in (0,) + _config_groups(args.config)
    if os.path.exists(pmc_path) and same_command:  # rocprofv3 passes of this same command (scripts/profile_round.sh)
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
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    # (before anything can make this process's first HIP call: mmsbm_amd/restarts.py says why)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from mmsbm_amd.build import ensure_library   # (no HIP, no torch: safe before the ranks are started)
    ensure_library()                             # a fresh clone has no libmmsbm_hip.so yet
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # N > 1 under somebody else's launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`):
    # nobody has timed the CPU side yet.  Rank 0 does it NOW, before anything in this process touches the GPU, as
    # N child processes (one restart each); the other ranks wait for it in the rendezvous of the process group.
    cpu_pre = None
    if os.environ.get(CPU_BASELINE_ENV):
        cpu_pre = json.loads(os.environ[CPU_BASELINE_ENV])
    elif args.gpus > 1 and int(os.environ.get("RANK", "0")) == 0 and not args.no_cpu_baseline:
        cpu_pre = cpu_baseline_processes(args, args.gpus)
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
    elif args.gpus > 1 and "LOCAL_RANK" not in os.environ:
        raise SystemExit("bench.py --gpus N under a launcher that does not set LOCAL_RANK: every rank would use GPU 0 "
                         "(torch.distributed.run sets it; --share-gpu allows it for a rehearsal)")
    # N > 1: the launcher's process group (RCCL).  N = 1: no group yet -- the one-rank group that lets the
    # end-of-job pick go through RCCL on every run is made AFTER the timed region (its barrier kernel and
    # proxy thread cost a 20-step run 2-3 us per step when they sit in front of it; a barrier among one
    # rank has nothing to wait for).
    # (a generous rendezvous, for bench.py only: with N > 1 its rank 0 times the CPU baseline before it joins)
    import datetime
    rank, world, local, device = restarts.init_from_env(args.dist_backend, timeout=datetime.timedelta(minutes=30))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if device.type != "cuda":
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if world > 1:
        # N ranks must sit on N different GPUs BEFORE anything is timed: one all_gather of (hostname, PCI bus id).
        # Every rank sees the same records, so every rank leaves together (no rank waits in a collective for one that
        # has gone); nothing is printed on stdout.
        from mmsbm_amd import _lib as _l
        here = gather_ranks({"rank": rank, "hostname": socket.gethostname(),
                             "pci_bus_id": _l.device_identity(local)["pci_bus_id"]}, world,
                            restarts._collective_device(device))
        err = restarts.distinct_device_error(here, world, args.share_gpu)
        if err is None and restarts.collective_info()["world_size"] != args.gpus:
            err = f"process group of {restarts.collective_info()['world_size']} rank(s) for --gpus {args.gpus}"
        if err:
            if dist.is_initialized():
                dist.destroy_process_group()
            raise SystemExit(f"bench.py: {err}")

    cfg = CONFIGS[args.config]
    if args.groups > 0:
        if world > 1:
            raise SystemExit("--groups is a one-GPU option")
        cfg = cfg[:4] + (args.groups, args.groups)
    n, u, i, r, k, l = cfg
    train = synthetic_triples(n, u, i, r, seed=0)
    model = MMSBM(k, l, iterations=args.steps, sampling=world, seed=0, backend="hip")
    # `job`: the wall clock of what a restart-per-GPU fit consists of, stage by stage (max over ranks below)
    job_t = {}
    t_job = time.perf_counter()
    model._prepare_objects(train)
    restarts.check_single_hip_runtime()
    ctx = model._ctx(local)
    ctx.set_graph_mode(1 if args.graph else 0)
    if args.mfma >= 0:
        ctx.set_option("mfma", args.mfma)
    ctx.synchronize()
    job_t["context_s"] = time.perf_counter() - t_job
    t_job = time.perf_counter()
    ctx.init_params(model.child_states[rank])  # restart `rank`, random start drawn on the device
    ctx.synchronize()
    job_t["random_start_s"] = time.perf_counter() - t_job

    t_job = time.perf_counter()
    ctx.iterate(args.warmup)
    job_t["iterate_s"] = time.perf_counter() - t_job

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
    job_t["iterate_s"] += elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=restarts._collective_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    # the binary that ran IS the one built from the sources in the tree (ensure_library rebuilds a stale one;
    # MMSBM_HIP_LIBRARY may point elsewhere): recorded, and `traffic` is dropped when it is not
    from mmsbm_amd import _lib
    build_id = _lib.build_id()

    # end of the job: likelihood of this rank's restart + ONE all-reduce to pick the best + the winner's theta / eta /
    # pr from the rank that ran it to everyone (three tensor broadcasts) -- evaluated on the parameters after
    # warmup + steps iterations, BEFORE the steady-state loop below advances them further
    coll_error = None
    if world == 1 and not args.no_collective_at_1:
        try:    # (process-group start-up is not part of `job` at any N: for N > 1 it happened before the job began --
            # the fence's barrier is the group's first collective and builds the RCCL communicator, 0.85 s; here the
            # same is done by one throw-away all-reduce)
            restarts.init_from_env(args.dist_backend, force_init=True)
            restarts.all_likelihoods({0: 0.0}, 1, device)
        except Exception as exc:  # a one-rank group is a nicety: report without it rather than not at all
            coll_error = f"{type(exc).__name__}: {exc}"
    t_job = time.perf_counter()
    lik = ctx.likelihood()
    job_t["likelihood_s"] = time.perf_counter() - t_job
    t_job = time.perf_counter()
    best, best_lik, liks = restarts.pick_max_likelihood({rank: lik}, world, device)
    job_t["pick_allreduce_s"] = time.perf_counter() - t_job
    t_job = time.perf_counter()
    mine_res = None
    if best % world == rank:
        theta_w, eta_w, pr_w = ctx.get_params()
        mine_res = {"likelihood": lik, "theta": theta_w, "eta": eta_w, "pr": pr_w}
    winner = restarts.broadcast_result(mine_res, best % world, restarts.result_shapes(model, train), best_lik, device)
    job_t["winner_broadcast_s"] = time.perf_counter() - t_job
    winner_sum = float(winner["theta"].sum() + winner["eta"].sum() + winner["pr"].sum()) if winner is not None else None
    job_names = ["context_s", "random_start_s", "iterate_s", "likelihood_s", "pick_allreduce_s", "winner_broadcast_s"]
    jt = torch.tensor([job_t[nm] for nm in job_names] + [sum(job_t.values())], dtype=torch.float64,
                      device=restarts._collective_device(device))
    if dist.is_initialized():
        dist.all_reduce(jt, op=dist.ReduceOp.MAX)
    job = dict(zip(job_names + ["total_s"], (float(x) for x in jt.tolist())))
    job.update({"iterations": args.warmup + args.steps, "winner": best, "winner_checksum": winner_sum,
                "definition": "wall clock from building the context (upload + sorts) to every rank holding the "
                              "maximum-likelihood restart's theta / eta / pr: context, device-side random start, "
                              "warmup + steps iterations, likelihood, ONE all-reduce(MAX), three tensor broadcasts; "
                              "each stage the max over ranks (total_s: max over ranks of the rank's own sum); "
                              "process-group start-up excluded"})
    # steady state: the same loop, long enough to leave the clock ramp behind (HIP events on the library's stream)
    steady_ms = ctx.time_iterations(args.steady_steps) / args.steady_steps if args.steady_steps > 0 else None
    # SURVEY 8(d)'s own protocol beside it: >= 50 iterations per repeat (after the warm-up above), HIP events, the
    # MEDIAN of 5 repeats
    rep_iters = max(50, min(200, args.steady_steps // 5)) if args.steady_steps > 0 else 0
    rep_ms = [ctx.time_iterations(rep_iters) / rep_iters for _ in range(5)] if rep_iters else []
    # who ran what: one record per rank (device identity from the library's own HIP runtime), gathered with ONE
    # tensor all_gather of fixed-size records -- so that a line from an 8-GPU node shows eight different PCI bus
    # ids, each rank's own time for the K steps and its restart's likelihood
    ident = _lib.device_identity(local)
    ranks = gather_ranks({"rank": rank, "local_rank": local, "device_index": local, "device_name": ident["name"],
                          "pci_bus_id": ident["pci_bus_id"], "compute_units": ident["compute_units"],
                          "hostname": socket.gethostname(), "pid": os.getpid(), "restart": rank,
                          "ms_per_step": 1000.0 * elapsed / args.steps,
                          "steady_ms_per_step": steady_ms,
                          "median5_ms_per_step": (sorted(rep_ms)[2] if rep_ms else None),
                          "likelihood": float(lik), "build_id": build_id}, world,
                         restarts._collective_device(device))

    # the line must not be able to lie about the hardware it ran on: N ranks, N distinct GPUs, one group of N
    lie = restarts.distinct_device_error(ranks, world, args.share_gpu)
    if lie is None and world > 1 and restarts.collective_info()["world_size"] != args.gpus:
        lie = f"process group of {restarts.collective_info()['world_size']} rank(s) for --gpus {args.gpus}"
    if lie:
        os.close(json_fd)
        if dist.is_initialized():
            dist.destroy_process_group()
        raise SystemExit(f"bench.py: {lie}")

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
            "metric": "EM iterations/sec (1M ratings, K=L=20)" if (args.config == "c3" and args.groups in (0, 20))
                      else f"EM iterations/sec ({args.config}" + (f", K=L={args.groups})" if args.groups else ")"),
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
            "median_of_5": (None if not rep_ms else
                            {"iterations_per_repeat": rep_iters, "repeats": 5,
                             "ms_per_step": max(r["median5_ms_per_step"] for r in ranks),
                             "value": world * 1000.0 / max(r["median5_ms_per_step"] for r in ranks), "unit": "it/s",
                             "rank0_repeats_ms_per_step": rep_ms,
                             "source": "SURVEY 8(d): >= 50 iterations per repeat after warm-up, HIP events on the "
                                       "library's stream, median of 5 repeats; max over ranks"}),
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
            "iterations_at_likelihood": args.warmup + args.steps,
            "job": job,
        }
        if args.batched_restarts > 1:
            out["batched_restarts"] = batched_rate(model, train, local, args.batched_restarts,
                                                   max(20, args.steps // 10))
        if cpu_pre is not None:      # N > 1: timed on N processes before the ranks touched their GPUs (above)
            out["cpu_baseline"] = cpu_pre
            out["gpu_over_cpu"] = its / cpu_pre["value"]
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, cpu_rows(args), args.cpu_iters)
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
