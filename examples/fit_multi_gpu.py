#!/usr/bin/env python3
"""Random restarts across the GPUs of one node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/fit_multi_gpu.py --sampling 8 --iterations 400

Restart i runs on rank i mod W (no collective on the data path); one all-reduce over RCCL at the
end tells every rank all likelihoods, i.e. the maximum-likelihood restart, whose theta / eta / pr are
then broadcast from the rank that ran it as three tensors (model.best_result).  The reference's predict
averages over ALL restarts: every rank scores its own on its GPU and one more all-reduce (SUM of the
(M, R) matrix) makes the mean -- the restarts' parameters never leave the rank that computed them."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from mmsbm_amd import restarts  # noqa: E402  (imports torch before the HIP library)
from mmsbm_amd import MMSBM  # noqa: E402
from mmsbm_amd.synthetic import synthetic_triples  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sampling", type=int, default=8)
    ap.add_argument("--iterations", type=int, default=400)
    ap.add_argument("--ratings", type=int, default=1_000_000)
    ap.add_argument("--groups", type=int, default=20)
    ap.add_argument("--dist-backend", default=None, help="default nccl (= RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0 (with --dist-backend gloo)")
    args = ap.parse_args()
    if args.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1 and "LOCAL_RANK" not in os.environ:
        raise SystemExit("WORLD_SIZE > 1 but LOCAL_RANK is not set: every rank would use GPU 0 (start the ranks with "
                         "torch.distributed.run; --share-gpu allows it for a rehearsal)")
    rank, world, local, device = restarts.init_from_env(args.dist_backend)
    if world > 1 and device.type == "cuda" and not args.share_gpu:   # W ranks, W different GPUs -- or stop here
        restarts.require_distinct_devices(local, device)
    train = synthetic_triples(args.ratings, args.ratings // 10, args.ratings // 50, 5, seed=0)
    model = MMSBM(args.groups, args.groups, iterations=args.iterations, sampling=args.sampling, seed=0)
    best, best_lik, liks = restarts.fit_distributed(model, train, device=device)   # (gather=False is the default)
    test = synthetic_triples(max(args.ratings // 10, 1), args.ratings // 10, args.ratings // 50, 5, seed=1)
    matrix = restarts.predict_distributed(model, test, device=device)      # mean over all restarts, every rank
    stats = model.score(silent=True)["stats"]
    if rank == 0:
        print(f"{world} rank(s), {args.sampling} restarts: likelihoods {np.round(liks, 3).tolist()}")
        print(f"maximum-likelihood restart: {best} ({best_lik:.3f}); this rank ran {len(model.results)} of them and holds the "
              f"winner's theta {model.best_result['theta'].shape}")
        print(f"prediction over all restarts: {matrix.shape[0]} rows, accuracy {stats['accuracy']:.4f}, mae {stats['mae']:.4f}")
    if world > 1:
        import torch.distributed as dist
        restarts.barrier(device)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
