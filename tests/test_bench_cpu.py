"""bench.py's host logic without a GPU: the roofline object (SURVEY 8(d) basis, staleness of the
committed counter summary), the self-launch command and the argument defaults the driver relies on."""
import json
import os
import sys
import types

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (light: torch and the library are imported inside main())


class FakeCtx:
    bytes_per_slot = 107_000_000
    n_pairs, n_users, n_items = 99_993, 99_997, 20_000


PROF = {"seg_pass_kernel": (64.0, 1, 360_798_360, 32_000_000), "pair_block_kernel(T+S)": (18.0, 1, 37_400_000, 21_000_000),
        "eta_p_kernel": (9.0, 1, 27_600_000, 3_300_000), "pair_block_kernel(A)": (12.0, 1, 21_400_000, 16_000_000)}


def _args(config="c3"):
    return types.SimpleNamespace(config=config)


def test_roofline_object_follows_survey_8d_and_reads_the_committed_profile():
    n, k, l = 1_000_000, 20, 20
    rf = bench.roofline_object(_args(), FakeCtx(), PROF, n, k, l)
    assert rf["bound"] == "hbm" and rf["kernel"] == "seg_pass_kernel" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["algorithmic_bytes_per_launch"] == n * (12 + 8 * k + 8 * l) == 332_000_000
    assert rf["achieved"] == pytest.approx(332_000_000 / 64e-6 / 1e9)
    assert rf["frac"] == pytest.approx(rf["achieved"] / 8000.0)
    assert rf["algorithmic_bytes_model"] == 360_798_360 and rf["achieved_model"] > rf["achieved"]
    assert "Infinity Cache" in rf["served_from"] and rf["resident_set_bytes"] < 256 << 20
    # the committed counter summary: used when it was taken with the present kernel sources, otherwise
    # reported as stale (traffic null) -- never silently mixed with timings of other sources
    with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as fh:
        pmc = json.load(fh)
    if pmc["_meta"]["c3"]["kernel_source_sha16"] == bench.kernel_source_sha16():
        assert rf["traffic"] == pmc["c3"]["seg_pass_kernel"]["hbm_bytes_per_launch"] > rf["algorithmic_bytes_per_launch"]
        assert rf["rocprof_source"].endswith("_kernel_stats.csv")
    else:   # mid-development: rerun scripts/profile_round.sh + summarize_profile.py before the round ends
        assert rf["traffic"] is None and "stale" in rf["traffic_source"]
    assert rf["rocprof_avg_us"] == pmc["c3"]["seg_pass_kernel"]["avg_us"]
    # every bench config has its summary, with the LDS-pipe counters (C2 runs the two-launch iteration)
    for cfg, kern in (("c2", "pairs_fused_kernel"), ("c2", "tail_fused_kernel"), ("c5", "pair_block_kernel(T+S)")):
        ent = pmc[cfg][kern]
        assert ent["avg_us"] > 0 and "SQ_WAIT_INST_LDS" in ent and "SQ_LDS_BANK_CONFLICT" in ent


def test_roofline_traffic_is_null_when_the_kernel_sources_changed(monkeypatch):
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: "0" * 16)
    rf = bench.roofline_object(_args(), FakeCtx(), PROF, 1_000_000, 20, 20)
    assert rf["traffic"] is None and "stale" in rf["traffic_source"]
    assert rf["rocprof_avg_us"] is not None and "older kernel sources" in rf["rocprof_source"]


def test_roofline_says_hbm_when_the_resident_set_exceeds_the_infinity_cache():
    class Big(FakeCtx):
        bytes_per_slot = 2_700_000_000
    rf = bench.roofline_object(_args("c5"), Big(), PROF, 10_000_000, 50, 50)
    assert rf["served_from"].startswith("HBM")


def test_self_launch_starts_torch_distributed_run_as_a_child(monkeypatch):
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "cpu_baseline_processes", lambda args, procs: {"value": 1.5, "cores": procs, "kind": "port"})
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    rc = bench.self_launch(types.SimpleNamespace(gpus=4, no_cpu_baseline=False))
    assert rc == 7
    # the GPU-free parent times the CPU side on N processes BEFORE the ranks exist and hands it to rank 0
    assert json.loads(seen["env"][bench.CPU_BASELINE_ENV]) == {"value": 1.5, "cores": 4, "kind": "port"}
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_cpu_baseline_on_as_many_processes_as_ranks():
    """VERDICT r3 missing 2: the CPU side of an N-GPU line is the CPU restatement on N processes, one restart each
    (the reference's Pool(processes=sampling), src/mmsbm.py:182-185): child processes that load neither torch nor
    the HIP library, timed regions overlapping, value = N / the slowest process's median iteration."""
    args = types.SimpleNamespace(config="c1", cpu_sample_rows=0, cpu_iters=3)
    out = bench.cpu_baseline_processes(args, 2)
    assert out["cores"] == 2 and out["kind"] == "port" and out["unit"] == "it/s"
    assert len(out["per_process_seconds"]) == 2 and out["seconds_per_iteration"] == max(out["per_process_seconds"])
    assert out["value"] == pytest.approx(2 / out["seconds_per_iteration"])
    assert "all 100 triples in each of 2 processes" in out["sample"] and "port_over_reference" in out
    # a sample where the whole workload would not fit the budget: scaled by rows
    args = types.SimpleNamespace(config="c2", cpu_sample_rows=5000, cpu_iters=1)
    out = bench.cpu_baseline_processes(args, 2)
    assert "first 5000 of 100000 triples" in out["sample"] and out["cores"] == 2


def test_argument_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.config, a.no_cpu_baseline, a.batched_restarts) == (1, "c3", False, 0) and a.steps >= 100


GATHER_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = {{"rank": rank, "device_name": "AMD (gfx950:sramecc+:xnack-)", "pci_bus_id": "0000:%02x:00.0" % (5 + rank),
        "ms_per_step": 0.1 + rank, "steady_ms_per_step": None, "likelihood": -1.5 * rank}}
out = bench.gather_ranks(mine, world, torch.device("cpu"))
assert [r["rank"] for r in out] == [0, 1] and out[rank] == mine and out[1]["pci_bus_id"] == "0000:06:00.0", out
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_per_rank_records_travel_as_one_tensor_all_gather(tmp_path):
    """bench.gather_ranks: every rank's record (device identity, its own time, its likelihood) reaches every rank in
    rank order through ONE plain tensor all_gather -- two gloo ranks here, RCCL on the GPUs."""
    import socket
    import subprocess
    script = tmp_path / "gather_worker.py"
    script.write_text(GATHER_WORKER.format(root=ROOT))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out = p.communicate(timeout=240)[0]
        assert p.returncode == 0 and f"rank {rank} ok" in out, out[-2000:]


# ---- the first real N-GPU run must not be able to lie or to trip ("What's weak" 6) ----------------------

def test_distinct_device_error():
    from mmsbm_amd import restarts
    recs = [{"rank": r, "hostname": "node0", "pci_bus_id": "0000:%02x:00.0" % (5 + r)} for r in range(4)]
    assert restarts.distinct_device_error(recs, 4) is None
    same = [dict(r, pci_bus_id="0000:05:00.0") for r in recs]
    err = restarts.distinct_device_error(same, 4)
    assert err and "4 ranks on 1 distinct GPU(s)" in err and "LOCAL_RANK" in err and "[0, 1, 2, 3]" in err
    assert restarts.distinct_device_error(same, 4, share_gpu=True) is None          # the rehearsal
    two = [recs[0], recs[1], dict(recs[2], pci_bus_id=recs[1]["pci_bus_id"]), recs[3]]
    assert "4 ranks on 3 distinct GPU(s)" in restarts.distinct_device_error(two, 4)
    # the same bus id on two HOSTS is two GPUs
    hosts = [dict(r, hostname=f"node{r['rank']}", pci_bus_id="0000:05:00.0") for r in recs]
    assert restarts.distinct_device_error(hosts, 4) is None
    assert "3 rank record(s) for a world of 4" in restarts.distinct_device_error(recs[:3], 4)


def test_bench_refuses_ranks_without_local_rank():
    """Under a launcher that sets WORLD_SIZE but not LOCAL_RANK every rank would take GPU 0 and the line would still
    look plausible: bench.py stops before any rendezvous, non-zero, nothing on stdout."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK",)}
    env.update(WORLD_SIZE="2", RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1", MMSBM_HIP_LAUNCH_LOG="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "LOCAL_RANK" in r.stderr, (r.returncode, r.stderr[-800:])


def test_ranks_get_the_dmabuf_ipc_default_before_their_first_hip_call():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (mmsbm_amd/restarts.py says why) is set by importing the module every multi-GPU
    entry point imports first -- whatever launched the rank -- and a value the caller chose is kept."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import mmsbm_amd.restarts; "
            "print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stderr[-800:]
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "1", r.stderr[-800:]


DISTINCT_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from mmsbm_amd import restarts, _lib
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
same = {same!r}
_lib.device_identity = lambda d: {{"pci_bus_id": "0000:05:00.0" if same else "0000:%02x:00.0" % (5 + rank)}}
dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    recs = restarts.require_distinct_devices(0, torch.device("cpu"))
except SystemExit as e:
    assert same and "2 ranks on 1 distinct GPU(s)" in str(e), e
    assert not dist.is_initialized()          # every rank left the group together
    print("rank", rank, "refused")
else:
    assert not same and [r["rank"] for r in recs] == [0, 1]
    dist.destroy_process_group()
    print("rank", rank, "ok")
"""


@pytest.mark.parametrize("same", [False, True])
def test_two_gloo_ranks_on_one_gpu_are_refused_on_every_rank(tmp_path, same):
    import socket
    import subprocess
    script = tmp_path / "distinct_worker.py"
    script.write_text(DISTINCT_WORKER.format(root=ROOT, same=same))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out = p.communicate(timeout=240)[0]
        assert p.returncode == 0 and f"rank {rank} {'refused' if same else 'ok'}" in out, out[-2000:]
