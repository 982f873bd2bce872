"""Random restarts across GPUs: one process per GPU (``torch.distributed``; backend
``nccl`` is RCCL on ROCm, ``gloo`` on CPU for tests), restart ``i`` on rank ``i mod W``.

The reference runs restarts in a ``multiprocessing.Pool`` (src/mmsbm.py:182-185) and never
lets them communicate.  Restarts are independent here too: there is NO collective on the
data path.  After every rank has finished its restarts one all-reduce (MAX over a
``sampling``-long vector that each rank fills at its own restart indices, -inf elsewhere)
tells every rank all likelihoods, hence the maximum-likelihood run.  The parameters of all
restarts can additionally be gathered (the reference's ``predict`` averages over all of
them, src/mmsbm.py:297-315).

torch is imported here -- before the HIP library is first loaded -- so that the process
holds ONE HIP runtime (torch's wheel bundles its own libamdhip64).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def shard_restarts(sampling, rank, world):
    """Indices of the restarts rank ``rank`` of ``world`` runs (round robin)."""
    return list(range(rank, sampling, world))


def init_from_env(backend=None, force_init=False):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns
    (rank, world, local_rank, device).  Single process when WORLD_SIZE is unset or 1: no process
    group is made then, unless ``force_init`` asks for a one-rank group (so that the collective
    backend -- RCCL for ``nccl`` -- really runs even on one GPU)."""
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
        dist.init_process_group(backend, rank=rank, world_size=world)
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


def gather_results(local_results, sampling):
    """local_results: {restart index: result dict}.  Every rank gets the list of all
    ``sampling`` results in restart order."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local_results[i] for i in range(sampling)]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_results)
    merged = {}
    for part in parts:
        merged.update(part)
    return [merged[i] for i in range(sampling)]


def fit_distributed(model, train, runner=None, gather=True, device=None):
    """Run ``model.sampling`` restarts sharded over the ranks of the current process group.

    model  : an ``mmsbm_amd.MMSBM`` (only ``sampling`` / ``child_states`` are used when a
             ``runner`` is given).
    runner : ``runner(i, child_seed) -> result dict``; default runs this rank's restarts on
             its GPU through ``model.fit_encoded`` (batched as slots of one context).
    Returns (best index, best likelihood, likelihood vector); ``model.results`` holds all
    restarts (``gather``) or this rank's share.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
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
    model.results = gather_results(local, model.sampling) if gather else [local[i] for i in mine]
    model._restart_ids = list(range(model.sampling)) if gather else mine
    return best, best_lik, liks


def predict_distributed(model, data, device=None, share_best=True):
    """``MMSBM.predict`` (src/mmsbm.py:279-317) for restarts that live on different ranks
    (``fit_distributed(..., gather=False)``; after ``gather=True`` every rank holds all restarts and each is
    scored by the lowest rank holding it): every rank adds the rating distributions of ITS restarts on
    its GPU, ONE all-reduce(SUM) of the (M, R) matrix makes the mean over all restarts -- no rank ever holds
    another rank's parameters (at BASELINE's config 5 a restart is 440 MB; the matrix of 1M test rows is
    80 MB).  ``data``: what ``model.predict`` takes, or encoded (M,3) triples after ``fit_encoded``.
    Sets prediction_matrix / run_stats on every rank; theta / eta / pr / likelihood of the restart with the
    best test accuracy on the rank that ran it -- on all ranks with ``share_best`` (one broadcast).
    The mean differs from a one-process predict only in the association order of the sum over restarts."""
    test = model.data_handler.transform(data, model.logger) if model.data_handler is not None else np.asarray(data)
    held = list(model._restart_ids)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    mine = held
    if multi:
        # A restart may be held by SEVERAL ranks (fit_distributed(gather=True) leaves every rank with all of
        # them): it is scored once, by the lowest rank that holds it -- otherwise the all-reduce(SUM) below
        # would count it once per holder and the mean would come out world-size times too large.
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, held)
        me = dist.get_rank()
        first = {}
        for r, ids in enumerate(everyone):
            for i in ids:
                first.setdefault(i, r)
        missing = [i for i in range(model.sampling) if i not in first]
        if missing:
            raise RuntimeError(f"predict_distributed: no rank holds restart(s) {missing}")
        mine = [i for i in held if first[i] == me]
    if mine:
        mean_local, _, stats_local = model._predict_runs(test, subset=[held.index(i) for i in mine])
        total = np.ascontiguousarray(mean_local * float(len(mine)), dtype=np.float64)
    else:   # more ranks than restarts: this one only takes part in the collectives
        model.test, stats_local = test, []
        total = np.zeros((len(test), len(model.ratings)), dtype=np.float64)
    per_run = dict(zip(mine, stats_local))
    owner = {i: 0 for i in mine}
    if multi:
        dev = _collective_device(device)
        t = torch.from_numpy(total).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = t.cpu().numpy()
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, per_run)      # six numbers per restart
        per_run, owner = {}, {}
        for r, part in enumerate(parts):
            per_run.update(part)
            owner.update({i: r for i in part})
    model.run_stats = [per_run[i] for i in range(model.sampling)]
    model.prediction_matrix = total / float(model.sampling)
    model._scored = None                            # score(): from the matrix (host side)
    best = int(np.argmax([st["accuracy"] for st in model.run_stats]))    # first best restart, src/mmsbm.py:474-478
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    res = model.results[held.index(best)] if best in mine else None
    if share_best and multi:
        box = [res]
        dist.broadcast_object_list(box, src=owner[best])
        res = box[0]
    if res is not None:
        model._keep_best_run(best, res)
    return model.prediction_matrix

