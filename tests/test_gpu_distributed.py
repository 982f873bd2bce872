"""The multi-GPU path on the ONE GPU a test box has (``-m gpu``): the RCCL (``nccl``) branches of
mmsbm_amd/restarts.py executed through a one-rank process group, and ``bench.py --gpus N``
starting its own ranks.  (Restarts are independent, src/mmsbm.py:182-185: with N GPUs it is one
restart per rank and ONE all-reduce at the end -- nothing here depends on N being 1.)"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu

ONE_RANK_NCCL = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    import torch, torch.distributed as dist
    from conftest import load_golden
    from mmsbm_amd import restarts, MMSBM, _lib
    rank, world, local, device = restarts.init_from_env("nccl", force_init=True)
    assert (rank, world) == (0, 1) and device.type == "cuda"
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    assert restarts.collective_info() == {{"backend": "nccl", "world_size": 1}}
    g = load_golden("g2_c1_sampling3")
    train = g["train"]
    model = MMSBM(2, 2, iterations=10, sampling=3, seed=1)
    best, best_lik, liks = restarts.fit_distributed(model, train, device=device)   # restarts on the GPU
    restarts.check_single_hip_runtime()
    assert any("libmmsbm_hip" in ln for ln in open("/proc/self/maps")), "HIP library not loaded"
    # the reference's sampling=3 run (SURVEY B.6): likelihoods, winner, every restart's parameters
    assert np.allclose(liks, g["likelihoods"], rtol=1e-10, atol=0), (liks, g["likelihoods"])
    assert best == int(np.argmax(g["likelihoods"])) == model.best_by_likelihood
    assert len(model.results) == 3      # (one rank: all three restarts are its own)
    for key in ("theta", "eta", "pr"):  # the winner, as every rank of a bigger group would hold it after the broadcast
        assert np.array_equal(model.best_result[key], model.results[best][key])
    # the explicit gather and the winner's broadcast on CUDA tensors through RCCL (one rank: identity, but the nccl calls run)
    full = restarts.gather_results(dict(enumerate(model.results)), 3, restarts.result_shapes(model, train), liks, device)
    assert len(full) == 3
    for s in range(3):
        assert np.max(np.abs(model.results[s]["theta"] - g[f"theta_{{s}}"])) < 1e-9
    # the collectives themselves, on CUDA tensors through RCCL
    vec = restarts.all_likelihoods({{0: -3.0, 2: -1.0}}, 3, device)
    assert vec.tolist() == [-3.0, float("-inf"), -1.0]
    t = torch.arange(4, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.tolist() == [0.0, 1.0, 2.0, 3.0]
    restarts.barrier(device)
    # predict over the ranks' restarts (one rank here: the reduction is the identity), on the device
    from oracle import mmsbm_oracle as orc
    test = train[::2]
    matrix = restarts.predict_distributed(model, test, device=device)
    want = np.mean([orc.prod_dist(test, g[f"theta_{{s}}"], g[f"eta_{{s}}"], g[f"pr_{{s}}"]) for s in range(3)], axis=0)
    assert np.allclose(matrix, want, rtol=1e-9, atol=1e-300), np.max(np.abs(matrix - want))
    assert len(model.run_stats) == 3 and model.theta.shape == g["theta_0"].shape
    model._release()
    dist.destroy_process_group()
    print("nccl one rank ok")
""")


def _run(cmd, env=None, timeout=600):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    base.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    return subprocess.run(cmd, env=base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout, cwd=ROOT)


def test_nccl_branches_run_on_a_one_rank_process_group(tmp_path):
    """VERDICT r1 missing 1 / ADVICE: init_process_group("nccl", world_size=1) in a fresh process, then
    fit_distributed, all_likelihoods, all_gather_object and barrier(device) with the HIP library loaded."""
    script = tmp_path / "one_rank.py"
    script.write_text(ONE_RANK_NCCL.format(root=ROOT))
    res = _run([sys.executable, str(script)], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                  "MASTER_ADDR": "127.0.0.1"})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "nccl one rank ok" in res.stdout


def _json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-3000:]
    return json.loads(lines[0])


def test_bench_one_gpu_line_has_roofline_cpu_baseline_and_rccl_collective():
    """The default shape of the driver's command on a small workload (C2): ONE JSON line with the
    contract's keys; at N=1 the end-of-job pick still goes through a one-rank RCCL group."""
    res = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--config", "c2",
                "--cpu-iters", "2"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = _json_line(res.stdout)
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["unit"] == "it/s" and out["dtype"] == "f64"
    assert out["collective"]["backend"] == "nccl" and out["collective"]["world_size"] == 1, out["collective"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] and 0 < rf["frac"] < 1.5
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9)
    # N(12+8K+8L) per iteration; C2 runs the two-launch iteration, one triple pass in each launch
    assert out["config"]["launches_per_iteration"] == 2 and rf["kernel"] in ("pairs_fused_kernel", "tail_fused_kernel")
    assert rf["algorithmic_bytes_per_launch"] == 100_000 * (12 + 8 * 10 + 8 * 10) // 2
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "all 100000 triples" in cb["sample"]
    assert 0.5 < cb["port_over_reference"] < 2.0
    assert out["value"] == pytest.approx(20 / (out["ms_per_step"] * 20e-3), rel=1e-6)
    # the line says what it ran on and with which binary, and carries the long-run figure beside the short one
    assert len(out["ranks"]) == 1 and out["distinct_devices"] == 1
    r0 = out["ranks"][0]
    assert r0["rank"] == 0 and r0["device_index"] == 0 and "gfx950" in r0["device_name"] and ":" in r0["pci_bus_id"]
    assert r0["ms_per_step"] == pytest.approx(out["ms_per_step"], rel=1e-9) and r0["likelihood"] == out["likelihoods"][0]
    assert out["library"]["matches_sources"] and out["library"]["build_id"] == r0["build_id"]
    job = out["job"]
    assert job["winner"] == 0 and job["iterations"] == 25 and 0 < job["iterate_s"] <= job["total_s"]
    assert job["winner_broadcast_s"] >= 0 and job["winner_checksum"] > 0
    ss = out["steady_state"]
    assert ss["steps"] == 1000 and 0 < ss["ms_per_step"] < 2 * out["ms_per_step"]
    assert ss["value"] == pytest.approx(1000.0 / ss["ms_per_step"], rel=1e-9)


def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2 --share-gpu --dist-backend gloo` with no launcher around it (VERDICT r1
    item 1c): the parent starts torch.distributed.run as a child, rank 0's line comes back, two ranks
    (two restarts) were timed.  Both ranks share GPU 0 here -- a rehearsal of the launch and collective
    plumbing, not a scaling measurement."""
    res = _run([sys.executable, "bench.py", "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "20",
                "--warmup", "5", "--config", "c2"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = _json_line(res.stdout)
    assert out["n_gpus"] == 2 and out["collective"] == {"backend": "gloo", "world_size": 2}
    assert len(out["likelihoods"]) == 2 and len(set(out["likelihoods"])) == 2   # two different restarts
    assert out["best_restart"] == int(np.argmax(out["likelihoods"]))
    # the N > 1 line is complete -- the CPU restatement on as many processes as ranks (timed by the
    # GPU-free parent before the ranks exist), the roofline of the dominant kernel, the job's wall clock by stage
    cb = out["cpu_baseline"]
    assert cb["cores"] == 2 and cb["kind"] == "port" and cb["value"] > 0 and "in each of 2 processes" in cb["sample"]
    assert len(cb["per_process_seconds"]) == 2 and out["gpu_over_cpu"] == pytest.approx(out["value"] / cb["value"])
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1.5
    job = out["job"]
    assert set(job) >= {"context_s", "random_start_s", "iterate_s", "likelihood_s", "pick_allreduce_s",
                        "winner_broadcast_s", "total_s", "winner", "winner_checksum"}
    assert job["winner"] == out["best_restart"] and job["iterations"] == 25 == out["iterations_at_likelihood"]
    assert 0 < job["iterate_s"] <= job["total_s"] and all(job[k_] >= 0 for k_ in job if k_.endswith("_s"))
    assert out["value"] == pytest.approx(2 * 20 / (out["ms_per_step"] * 20e-3), rel=1e-6)
    # self-validation for the day a driver has N GPUs: one record per rank, gathered over the process group
    ranks = out["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["restart"] for r in ranks] == [0, 1]
    assert len({r["pid"] for r in ranks}) == 2                                    # two processes ...
    assert out["distinct_devices"] == 1 and {r["pci_bus_id"] for r in ranks} == {ranks[0]["pci_bus_id"]}   # ... sharing GPU 0 here
    for r in ranks:
        assert set(r) >= {"rank", "device_index", "device_name", "pci_bus_id", "ms_per_step", "steady_ms_per_step",
                          "likelihood", "build_id", "hostname"}
        assert "gfx950" in r["device_name"] and r["ms_per_step"] > 0 and r["build_id"] == out["library"]["build_id"]
    assert [r["likelihood"] for r in ranks] == out["likelihoods"]                 # len = sampling = world size
    assert out["ms_per_step"] == pytest.approx(max(r["ms_per_step"] for r in ranks), rel=1e-9)   # max over ranks
    assert out["steady_state"]["ms_per_step"] == max(r["steady_ms_per_step"] for r in ranks)
    assert out["steady_state"]["value"] == pytest.approx(2 * 1000.0 / out["steady_state"]["ms_per_step"], rel=1e-9)


def test_bench_under_the_drivers_launcher_times_the_cpu_side_before_touching_the_gpu():
    """What the driver runs for N > 1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- no
    parent of ours.  Rank 0 then times the CPU restatement on N processes itself, BEFORE anything in it touches the
    GPU (the other ranks wait in the rendezvous), and the line is as complete as the self-launched one."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--share-gpu", "--dist-backend",
                "gloo", "--steps", "20", "--warmup", "5", "--config", "c2", "--steady-steps", "100", "--cpu-iters", "1"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = _json_line(res.stdout)
    assert out["n_gpus"] == 2 and out["cpu_baseline"]["cores"] == 2 and "roofline" in out and "job" in out
    assert out["job"]["winner"] == out["best_restart"] == int(np.argmax(out["likelihoods"]))
