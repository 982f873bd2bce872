"""Random restarts across GPUs: one process per GPU (``torch.distributed``; backend
``nccl`` is RCCL on ROCm, ``gloo`` on CPU for tests), restart ``i`` on rank ``i mod W``.

The reference runs restarts in a ``multiprocessing.Pool`` (src/mmsbm.py:182-185) and never
lets them communicate.  Restarts are independent here too: there is NO collective on the
data path.  After every rank has finished its restarts one all-reduce (MAX over a
``sampling``-long vector that each rank fills at its own restart indices, -inf elsewhere)
tells every rank all likelihoods, hence the maximum-likelihood run, whose theta / eta / pr are
then broadcast from the rank that ran it as three float64 tensors.  That is the whole end of the
job.  The reference's ``predict`` averages over all restarts (src/mmsbm.py:297-315):
``predict_distributed`` does that with one all-reduce of the (M, R) matrix, every restart staying
where it ran; ``fit_distributed(gather=True)`` hands every rank every restart if asked to (tensor
all_gathers).  Every collective here is a plain tensor collective -- nothing is pickled.

torch is imported here -- before the HIP library is first loaded -- so that the process
holds ONE HIP runtime (torch's wheel bundles its own libamdhip64).
"""
from __future__ import annotations

import os

# RCCL's intra-node transport (and torch's sharing of device tensors between processes) hands device memory from one
# process to another through hipIpcGetMemHandle / hipIpcOpenMemHandle.  On hosts whose driver only supports dmabuf IPC
# -- the MI355X pool this was built on -- the HSA runtime's legacy IPC mode fails there (`hipIpcGetMemHandle: invalid
# argument`); HSA_ENABLE_IPC_MODE_LEGACY=0 selects the dmabuf path.  Measured (scripts/ipc_mode_probe.py, one GPU, a
# device buffer opened by a spawned child): works with the variable at 0, fails with it unset.  The runtime reads it
# when it starts, i.e. at the process's FIRST HIP call, so the default goes in here -- at import, before torch or the
# library can have made one -- for every rank whatever launched it (torchrun, the driver, a scheduler).  A value the
# caller has set is left alone.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from . import _lib  # noqa: E402


def shard_restarts(sampling, rank, world):
    """Indices of the restarts rank ``rank`` of ``world`` runs (round robin)."""
    return list(range(rank, sampling, world))


def init_from_env(backend=None, force_init=False, timeout=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns
    (rank, world, local_rank, device).  Single process when WORLD_SIZE is unset or 1: no process
    group is made then, unless ``force_init`` asks for a one-rank group (so that the collective
    backend -- RCCL for ``nccl`` -- really runs even on one GPU).  ``timeout``: a ``datetime.timedelta`` for the
    group's rendezvous and collectives; None keeps torch's default, so that a dead rank or mismatched
    collectives surface after minutes, not half an hour (bench.py passes a long one: its rank 0 times the CPU
    baseline before it joins)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (see the top of this module: before the first HIP call)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local)
    backend = backend or ("nccl" if use_gpu else "gloo")
    if (world > 1 or force_init) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                os.environ["MASTER_PORT"] = "29500"
            else:  # a one-rank group needs no agreed port: take a free one
                import socket
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        extra = {} if timeout is None else {"timeout": timeout}
        dist.init_process_group(backend, rank=rank, world_size=world, **extra)
    # tensors for the collectives live where the backend wants them
    coll = torch.device("cuda", local) if (use_gpu and backend == "nccl") else torch.device("cpu")
    init_from_env.collective_device = coll
    return rank, world, local, (torch.device("cuda", local) if use_gpu else torch.device("cpu"))


def barrier(device=None):
    """dist.barrier() that tells RCCL which GPU this rank owns (avoids its device guess)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_backend() == "nccl" and device is not None and device.type == "cuda":
        dist.barrier(device_ids=[device.index])
    else:
        dist.barrier()


def distinct_device_error(records, world, share_gpu=False):
    """None when the ``world`` ranks sit on ``world`` different GPUs, else what is wrong, as a string.

    ``records``: one dict per rank with "hostname" and "pci_bus_id" (bench.gather_ranks).  A launcher that leaves
    LOCAL_RANK unset puts every rank on GPU 0: the job still runs, returns 0 and reports a plausible number -- for
    ONE GPU's worth of hardware.  ``share_gpu`` (a rehearsal on a one-GPU box) is the only legitimate case."""
    if share_gpu:
        return None
    if len(records) != world:
        return f"{len(records)} rank record(s) for a world of {world}"
    seen = {}
    for r in records:
        seen.setdefault((r.get("hostname"), r.get("pci_bus_id")), []).append(r.get("rank"))
    if len(seen) == world:
        return None
    shared = "; ".join(f"{host} {bus}: ranks {ranks}" for (host, bus), ranks in seen.items() if len(ranks) > 1)
    return (f"{world} ranks on {len(seen)} distinct GPU(s) ({shared}) -- is LOCAL_RANK set by the launcher? "
            "(--share-gpu allows it for a rehearsal)")


def require_distinct_devices(local, device=None, share_gpu=False):
    """Every rank of the process group on its own GPU, or SystemExit on EVERY rank: one all_gather of the ranks'
    (hostname, PCI bus id), padded to fixed-size byte tensors.  Call it before the first restart starts."""
    if not _grouped():
        return
    import socket
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = f"{socket.gethostname()}|{_lib.device_identity(local)['pci_bus_id']}".encode()[:255]
    coll = _collective_device(device)
    buf = torch.zeros(256, dtype=torch.uint8)
    buf[:len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    buf = buf.to(coll)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    recs = []
    for r, p in enumerate(parts):
        host, _, bus = bytes(p.cpu().numpy().tobytes()).rstrip(b"\0").decode().partition("|")
        recs.append({"rank": r, "hostname": host, "pci_bus_id": bus})
    err = distinct_device_error(recs, world, share_gpu)
    if err:
        dist.destroy_process_group()
        raise SystemExit(f"mmsbm_amd.restarts: {err}")
    return recs


def check_single_hip_runtime():
    libs = _lib.loaded_hip_runtimes()
    if len(libs) > 1:
        raise RuntimeError("two HIP runtimes are mapped into this process (" + ", ".join(libs) +
                           "): import torch (or mmsbm_amd.restarts) before the first HipEM is made")


def _collective_device(device):
    if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
        return torch.device("cpu")
    return device if device is not None else torch.device("cpu")


def all_likelihoods(local, sampling, device=None):
    """local: {restart index: likelihood}.  ONE all-reduce(MAX); returns the full vector
    (length ``sampling``) on every rank."""
    vec = torch.full((sampling,), float("-inf"), dtype=torch.float64,
                     device=_collective_device(device))
    for i, lik in local.items():
        vec[i] = float(lik)
    if dist.is_available() and dist.is_initialized():  # (a one-rank group still goes through RCCL)
        dist.all_reduce(vec, op=dist.ReduceOp.MAX)
    return vec.cpu().numpy()


def collective_info():
    """{"backend", "world_size"} of the process group the collectives run on (None: no group)."""
    if dist.is_available() and dist.is_initialized():
        return {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    return {"backend": None, "world_size": 1}


def pick_max_likelihood(local, sampling, device=None):
    """(index, likelihood, all likelihoods) of the maximum-likelihood restart; ties -> lowest
    index (np.argmax)."""
    liks = all_likelihoods(local, sampling, device)
    best = int(np.argmax(liks))
    return best, float(liks[best]), liks


def _grouped():
    return dist.is_available() and dist.is_initialized()


def result_shapes(model, train):
    """((U, K), (I, L), (K, L, R)) of a restart's theta / eta / pr -- every rank can tell them from the training
    triples and the model, so no collective has to carry them."""
    train = np.asarray(train)
    k, l = int(model.user_groups), int(model.item_groups)
    if train.ndim != 2 or len(train) == 0:
        return (0, k), (0, l), (k, l, 0)
    n_r = len(np.flatnonzero(np.bincount(train[:, 2])))
    return (int(train[:, 0].max()) + 1, k), (int(train[:, 1].max()) + 1, l), (k, l, n_r)


def _pack(res, shapes, device):
    """theta | eta | pr of a result dict as ONE flat float64 tensor where the backend wants it (zeros for None)."""
    sizes = [int(np.prod(sh)) for sh in shapes]
    if res is None:
        return torch.zeros(sum(sizes), dtype=torch.float64, device=device)
    flat = np.concatenate([np.ascontiguousarray(res[key], dtype=np.float64).reshape(-1)
                           for key in ("theta", "eta", "pr")])
    assert flat.size == sum(sizes), (flat.size, shapes)
    return torch.from_numpy(flat).to(device)


def _unpack(flat, shapes, likelihood):
    flat = flat.cpu().numpy()
    out, at = {"likelihood": likelihood}, 0
    for key, sh in zip(("theta", "eta", "pr"), shapes):
        n = int(np.prod(sh))
        out[key] = flat[at:at + n].reshape(sh).copy()
        at += n
    return out


def broadcast_result(res, src, shapes, likelihood, device=None):
    """One restart's theta / eta / pr from rank ``src`` to every rank: three float64 TENSOR broadcasts (RCCL over
    xGMI on the nccl backend; 19 MB at BASELINE's config 3, 440 MB at config 5) -- no pickles.  ``res`` is the
    result dict on ``src`` and ignored elsewhere; the likelihood is already known to everyone (all_likelihoods)."""
    if not _grouped():   # (a one-rank group still goes through the backend: that is how one GPU exercises RCCL)
        return res
    dev = _collective_device(device)
    me = dist.get_rank()
    # the owner says what it is about to send (2 + 2 + 3 integers): every rank allocates from THAT and every rank
    # sees a disagreement with what it expected at the same point -- before the big broadcasts, so that a runner
    # returning other shapes fails everywhere together instead of leaving the receivers blocked in a collective
    expected = [int(x) for sh in shapes for x in sh]
    if me == src:
        sent = [int(x) for key in ("theta", "eta", "pr") for x in np.shape(res[key])]
        if len(sent) != len(expected):
            sent = [-1] * len(expected)
    else:
        sent = [0] * len(expected)
    head = torch.tensor(sent, dtype=torch.int64, device=dev)
    dist.broadcast(head, src=src)
    sent = [int(x) for x in head.cpu().tolist()]
    if sent != expected:
        raise ValueError(f"broadcast_result: rank {src} holds theta / eta / pr of shapes {sent}, expected {expected}")
    out = {"likelihood": float(likelihood)}
    for key, sh in zip(("theta", "eta", "pr"), shapes):
        if me == src:
            t = torch.from_numpy(np.ascontiguousarray(res[key], dtype=np.float64)).to(dev)
        else:
            t = torch.empty(sh, dtype=torch.float64, device=dev)
        dist.broadcast(t, src=src)
        out[key] = res[key] if me == src else t.cpu().numpy()
    return out


def gather_results(local_results, sampling, shapes=None, liks=None, device=None):
    """local_results: {restart index: result dict}.  Every rank gets the list of all ``sampling`` results in
    restart order -- the explicit ``gather=True`` of fit_distributed (the reference's ``predict`` averages over
    all restarts, src/mmsbm.py:297-315; predict_distributed does that WITHOUT moving parameters).  Tensor
    ``all_gather``s, one per round of restarts (restart r + j * world in round j), each a flat theta | eta | pr
    vector; the likelihoods are the ones the all-reduce already gave every rank."""
    if not _grouped():
        return [local_results[i] for i in range(sampling)]
    world, me = dist.get_world_size(), dist.get_rank()
    dev = _collective_device(device)
    merged = dict(local_results)
    for j in range((sampling + world - 1) // world):
        mine = me + j * world
        buf = _pack(local_results.get(mine) if mine < sampling else None, shapes, dev)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        for r, part in enumerate(parts):
            i = r + j * world
            if i < sampling and i not in merged:
                merged[i] = _unpack(part, shapes, float(liks[i]))
    return [merged[i] for i in range(sampling)]


def fit_distributed(model, train, runner=None, gather=False, device=None, share_best=True):
    """Run ``model.sampling`` restarts sharded over the ranks of the current process group.

    model  : an ``mmsbm_amd.MMSBM`` (only ``sampling`` / ``child_states`` are used when a
             ``runner`` is given).
    runner : ``runner(i, child_seed) -> result dict``; default runs this rank's restarts on
             its GPU through ``model.fit_encoded`` (batched as slots of one context).

    End of the job (north_star / SURVEY 8(e)): ONE all-reduce(MAX) of the likelihood vector, then -- with
    ``share_best`` -- the maximum-likelihood restart's theta / eta / pr broadcast from the rank that ran it as
    three float64 tensors (``model.best_result`` on every rank).  Nothing else moves: ``model.results`` holds
    THIS rank's restarts (``model._restart_ids`` says which), and ``predict_distributed`` averages over all
    restarts without any rank ever holding another rank's parameters.  ``gather=True`` additionally hands
    every rank every restart (tensor all_gathers; 8 x 440 MB per rank at BASELINE's config 5 -- ask for it
    only if you need it).  Returns (best index, best likelihood, likelihood vector).
    """
    world = dist.get_world_size() if _grouped() else 1
    rank = dist.get_rank() if _grouped() else 0
    mine = shard_restarts(model.sampling, rank, world)
    if runner is None:  # this rank's restarts, batched as slots of one context on its GPU
        local_dev = device.index if (device is not None and device.type == "cuda") else 0
        check_single_hip_runtime()
        model.devices = [local_dev]
        model.fit_encoded(train, restarts=mine)
        local = dict(zip(mine, model.results))
    else:
        local = {i: runner(i, model.child_states[i]) for i in mine}
    best, best_lik, liks = pick_max_likelihood({i: r["likelihood"] for i, r in local.items()},
                                               model.sampling, device)
    model.best_by_likelihood = best
    model._all_liks = liks
    shapes = result_shapes(model, train)
    model.best_result = local.get(best)
    if share_best:
        model.best_result = broadcast_result(local.get(best), best % world, shapes, best_lik, device)
    if gather:
        model.results = gather_results(local, model.sampling, shapes, liks, device)
        model._restart_ids = list(range(model.sampling))
    else:
        model.results = [local[i] for i in mine]
        model._restart_ids = mine
    return best, best_lik, liks


def predict_distributed(model, data, device=None, share_best=True):
    """``MMSBM.predict`` (src/mmsbm.py:279-317) for restarts that live on different ranks
    (``fit_distributed``'s default; after ``gather=True`` every rank holds all restarts and each is scored by ONE
    of its holders, dealt round robin): every rank adds the rating distributions of ITS restarts on
    its GPU, ONE all-reduce(SUM) of the (M, R) matrix makes the mean over all restarts -- no rank ever holds
    another rank's parameters (at BASELINE's config 5 a restart is 440 MB; the matrix of 1M test rows is
    80 MB).  ``data``: what ``model.predict`` takes, or encoded (M,3) triples after ``fit_encoded``.
    Sets prediction_matrix / run_stats on every rank; theta / eta / pr / likelihood of the restart with the
    best test accuracy on the rank that ran it -- on all ranks with ``share_best`` (tensor broadcasts).
    Every collective is a plain tensor collective (who holds what: an all_gather of a 0/1 vector; the six raw
    sums of each restart: an all-reduce of a (sampling, 6) matrix).
    The mean differs from a one-process predict only in the association order of the sum over restarts."""
    test = model.data_handler.transform(data, model.logger) if model.data_handler is not None else np.asarray(data)
    held = list(model._restart_ids)
    multi = _grouped() and dist.get_world_size() > 1
    mine = held
    scorer = {i: 0 for i in held}
    dev = _collective_device(device)
    if multi:
        # A restart may be held by SEVERAL ranks (fit_distributed(gather=True) leaves every rank with all of
        # them): it is scored ONCE -- otherwise the all-reduce(SUM) below would count it once per holder -- by
        # holder number (restart mod holders), so that the work stays spread over the ranks.
        world, me = dist.get_world_size(), dist.get_rank()
        flags = torch.zeros(model.sampling, dtype=torch.int64, device=dev)
        if held:
            flags[torch.as_tensor(held, dtype=torch.int64, device=dev)] = 1
        everyone = [torch.empty_like(flags) for _ in range(world)]
        dist.all_gather(everyone, flags)
        table = torch.stack(everyone).cpu().numpy()            # (world, sampling)
        scorer, missing = {}, []
        for i in range(model.sampling):
            holders = np.flatnonzero(table[:, i])
            if holders.size == 0:
                missing.append(i)
            else:
                scorer[i] = int(holders[i % holders.size])
        if missing:
            raise RuntimeError(f"predict_distributed: no rank holds restart(s) {missing}")
        mine = [i for i in held if scorer[i] == me]
    raw_all = np.zeros((model.sampling, 6), dtype=np.float64)
    if mine:
        mean_local, _, _ = model._predict_runs(test, subset=[held.index(i) for i in mine])
        total = np.ascontiguousarray(mean_local * float(len(mine)), dtype=np.float64)
        for i, raw in zip(mine, model._raw_per_run):
            raw_all[i] = raw
    else:   # more ranks than restarts: this one only takes part in the collectives
        model.test = test
        total = np.zeros((len(test), len(model.ratings)), dtype=np.float64)
    if multi:
        t = torch.from_numpy(total).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = t.cpu().numpy()
        t6 = torch.from_numpy(raw_all).to(dev)                  # six numbers per restart, each filled by one rank
        dist.all_reduce(t6, op=dist.ReduceOp.SUM)
        raw_all = t6.cpu().numpy()
    from . import mmsbm as _host
    model.run_stats = [_host.HipEM.final_stats(raw_all[i]) for i in range(model.sampling)]
    model.prediction_matrix = total / float(model.sampling)
    model._scored = None                            # score(): from the matrix (host side)
    best = int(np.argmax([st["accuracy"] for st in model.run_stats]))    # first best restart, src/mmsbm.py:474-478
    res = model.results[held.index(best)] if best in held else None
    if share_best and multi:
        if getattr(model, "_all_liks", None) is not None:
            lik = float(model._all_liks[best])
        else:                                        # the owner tells everyone (one scalar)
            lt = torch.tensor([res["likelihood"] if dist.get_rank() == scorer[best] else 0.0], dtype=torch.float64, device=dev)
            dist.broadcast(lt, src=scorer[best])
            lik = float(lt.item())
        res = broadcast_result(res, scorer[best], result_shapes(model, model.train), lik, device)
    if res is not None:
        model._keep_best_run(best, res)
    return model.prediction_matrix
