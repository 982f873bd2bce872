"""Host-side logic that needs no GPU: encoder vs the reference's DataHandler output (golden),
restart seeding / initial parameters vs the reference, restart sharding and the one-all-reduce
maximum-likelihood pick over a 2-rank gloo group."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, load_golden
from mmsbm_amd.encode import Encoder
from mmsbm_amd.mmsbm import MMSBM, normalize_with_self
from mmsbm_amd.synthetic import CONFIGS, algorithmic_bytes, synthetic_triples
from oracle import mmsbm_oracle as orc


def test_encoder_matches_reference_datahandler():
    g = load_golden("g1_c1_mock")
    enc = Encoder()
    train = enc.fit_transform((g["train_raw_users"], g["train_raw_items"], g["train_raw_ratings"]))
    assert np.array_equal(train, g["train"])
    assert enc.user_labels() == list(g["dict_users_keys"])
    assert enc.item_labels() == list(g["dict_items_keys"])
    assert enc.rating_labels() == list(g["dict_ratings_keys"]) == ["1", "2", "3", "4", "5"]
    test = enc.transform((g["test_raw_users"], g["test_raw_items"], g["test_raw_ratings"]))
    assert np.array_equal(test, g["test"])


def test_encoder_lexicographic_order_and_unseen_rows(caplog):
    import pandas as pd
    df = pd.DataFrame({"users": [10, 2, 2, 33], "items": ["b", "a", "b", "a"], "ratings": [5, 10, 5, 1]})
    enc = Encoder()
    out = enc.fit_transform(df)
    assert enc.user_labels() == ["10", "2", "33"]        # '10' < '2' < '33' as strings
    assert enc.rating_labels() == ["1", "10", "5"]
    assert out.tolist() == [[0, 1, 2], [1, 0, 1], [1, 1, 2], [2, 0, 0]]
    test = pd.DataFrame({"users": [2, 99, 10, 2], "items": ["a", "a", "zz", "b"], "ratings": [5, 5, 5, 7]})
    with caplog.at_level("WARNING", logger="MMSBM"):
        enc_test = enc.transform(test)
    assert enc_test.tolist() == [[1, 0, 2]]               # rows with unseen user / item / rating dropped
    assert "99" in caplog.text and "zz" in caplog.text and "7" in caplog.text
    with pytest.raises(AssertionError):
        Encoder().fit_transform(pd.DataFrame({"u": [1, None], "i": [1, 2], "r": [1, 2]}))


def test_restart_seeding_and_initial_parameters_match_reference():
    g = load_golden("g1_c1_mock")
    mm = MMSBM(2, 4, iterations=500, sampling=1, seed=1)       # constructor works without a GPU
    mm.p, mm.m = 4, 9
    mm._dims = {"n_ratings": 5}
    theta, eta, pr = mm.init_params(mm.child_states[0], g["d_u"], g["d_i"])
    assert np.array_equal(theta, g["c1_theta_0"])
    assert np.array_equal(eta, g["c1_eta_0"])
    assert np.array_equal(pr, g["c1_pr_0"])
    three = MMSBM(2, 4, sampling=3, seed=1)
    assert three.child_states[0].spawn_key == (0,) and three.child_states[2].spawn_key == (2,)
    z = np.zeros((2, 2, 3)); z[0, 0] = [1, 1, 2]
    assert normalize_with_self(z)[0, 0].tolist() == [0.25, 0.25, 0.5] and not normalize_with_self(z)[1].any()


def test_predict_before_fit_and_unknown_backend():
    mm = MMSBM(2, 2, seed=1)
    with pytest.raises(AssertionError):
        mm.predict(None)
    with pytest.raises(AssertionError):
        mm.score()
    mm.backend = "numpy"                                        # this package ships no such backend
    with pytest.raises(ImportError, match="Could not load any backend"):
        mm._prepare_objects(np.array([[0, 0, 0]]))
    mm2 = MMSBM(1, 1, seed=1)
    mm2._compute_stats = lambda x: {"accuracy": x}
    assert mm2.choose_best_run([0.1, 0.7, 0.3]) == 1            # reference tests/test_mmsbm.py:124-132


def test_synthetic_generator_and_byte_accounting():
    a = synthetic_triples(1000, 50, 20, 5, seed=0)
    assert np.array_equal(a, orc.synthetic_triples(1000, 50, 20, 5, seed=0))
    n, u, i, r, k, l = CONFIGS["c3"]
    rd, wr = algorithmic_bytes(n, u, i, r, k, l)
    assert rd == 332_016_000 and wr == 8 * (u * k + i * l + k * l * r)


def test_encoder_integer_columns_take_the_numeric_path_with_the_same_result():
    """Integer id columns (the usual case) are encoded without a string per row or per distinct value:
    presence table or hash, lexicographic order of the decimal strings computed numerically.  Same ids and
    labels as the general path (str(value), sorted), which the reference's DataHandler golden pins."""
    import pandas as pd
    from mmsbm_amd import encode

    def general(col):
        codes, uniq = pd.factorize(np.asarray(col), use_na_sentinel=True)
        labels, inv = np.unique(np.array([str(v) for v in uniq.tolist()], dtype=object).astype(str), return_inverse=True)
        return inv.astype(np.int32)[codes], labels

    rng = np.random.default_rng(3)
    cols = [rng.integers(0, 1000, 5000), rng.integers(0, 10 ** 18, 3000), np.array([0, 0, 10, 1, 100, 2, 20, 19, 9, 99, 1000000]),
            rng.integers(5, 50, 10).astype(np.int32), rng.integers(0, 2 ** 40, 4000).astype(np.uint64), np.array([7]),
            rng.integers(10 ** 6, 10 ** 6 + 30, 100), np.array([5, 3, 5, 12], dtype=np.int8),
            np.array([-1, 5, 3]),                                              # negative: general path
            np.array([2 ** 63 + 5, 3, 2 ** 63 + 5], dtype=np.uint64)]           # beyond int64: general path
    for col in cols:
        ids, labels = encode._factorize_as_str(col)
        want_ids, want_labels = general(col)
        assert np.array_equal(ids, want_ids) and labels.tolist() == want_labels.tolist(), col[:5]
        out = np.empty(len(col), dtype=np.int32)
        encode._factorize_as_str(col, out=out)
        assert np.array_equal(out, want_ids)
    vals = np.array([0, 1, 2, 9, 10, 11, 19, 20, 99, 100, 101, 1000])
    assert [str(v) for v in vals[encode._decimal_string_order(vals)]] == sorted(str(v) for v in vals)
    # a frame with integer ids and a test frame holding unseen ones
    df = pd.DataFrame({"u": [10, 2, 33, 2, 10], "i": [7, 7, 100, 8, 8], "r": [5, 1, 3, 1, 5]})
    enc = encode.Encoder()
    got = enc.fit_transform(df)
    assert enc.user_labels() == ["10", "2", "33"] and enc.item_labels() == ["100", "7", "8"] and enc.rating_labels() == ["1", "3", "5"]
    assert got.tolist() == [[0, 1, 2], [1, 1, 0], [2, 0, 1], [1, 2, 0], [0, 2, 2]]
    test = enc.transform(pd.DataFrame({"u": [2, 4, 33], "i": [8, 8, 7], "r": [3, 3, 2]}))
    assert test.tolist() == [[1, 2, 1]]                                       # user 4 and rating 2 were never seen
    # transform: table lookup (dense ids) / hash lookup (sparse ids) against the general string path
    import logging
    logging.disable(logging.WARNING)
    try:
        for trial in range(12):
            n, sparse = int(rng.integers(5, 300)), trial % 3 == 0
            scale = 10 ** 9 if sparse else 1
            tr = pd.DataFrame({"u": rng.integers(0, 50, n) * scale, "i": rng.integers(0, 30, n) * scale, "r": rng.integers(1, 6, n)})
            te = pd.DataFrame({"u": rng.integers(0, 60, 150) * scale, "i": rng.integers(0, 35, 150) * scale, "r": rng.integers(0, 7, 150)})
            a, b = encode.Encoder(), encode.Encoder()
            a.fit_transform(tr); b.fit_transform(tr)
            assert all(m is not None for m in a._int_maps) and ("table" in a._int_maps[0]) == (not sparse or n < 10)
            b._int_maps = [None] * 3                                               # force the string path
            assert np.array_equal(a.transform(te), b.transform(te)), trial
            assert np.array_equal(a.transform(te.astype(str)), b.transform(te))    # strings against integer training ids
    finally:
        logging.disable(logging.NOTSET)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch.distributed as dist
    from mmsbm_amd import restarts
    from mmsbm_amd.mmsbm import MMSBM
    from oracle import mmsbm_oracle as orc
    from conftest_path import load_golden

    rank, world, local, device = restarts.init_from_env("gloo")
    g = load_golden("g2_c1_sampling3")
    train = g["train"]
    model = MMSBM(2, 2, iterations=10, sampling=3, seed=1)

    def runner(i, seed):      # CPU stand-in for the GPU restart (the oracle), same seeds
        return orc.run_one_sampling(train, seed, 2, 2, 10)

    mine = restarts.shard_restarts(3, rank, world)
    # the default end of the job: ONE all-reduce + the winner's three tensors; nobody holds anybody else's restart
    best, best_lik, liks = restarts.fit_distributed(model, train, runner=runner, device=device)
    assert mine == ([0, 2] if rank == 0 else [1]), mine
    assert np.array_equal(liks, g["likelihoods"]), (liks, g["likelihoods"])
    assert best == int(np.argmax(g["likelihoods"])) and model.best_by_likelihood == best
    assert list(model._restart_ids) == mine and len(model.results) == len(mine)
    win = model.best_result
    assert win["likelihood"] == g["likelihoods"][best]
    for key in ("theta", "eta", "pr"):
        assert np.array_equal(win[key], g[f"{{key}}_{{best}}"]), key
    # without the broadcast only the rank that ran the winner has it
    restarts.fit_distributed(model, train, runner=runner, device=device, share_best=False)
    assert (model.best_result is not None) == (best in mine)
    # a runner that returns other shapes than the training set implies: EVERY rank raises, at the same point (the
    # owner's shapes travel first), nobody is left waiting in a broadcast
    def odd_runner(i, seed):
        res = dict(runner(i, seed))
        res["theta"] = res["theta"][:, :1]
        return res
    try:
        restarts.fit_distributed(model, train, runner=odd_runner, device=device)
        raise SystemExit("mismatched shapes went through")
    except ValueError as exc:
        assert "broadcast_result" in str(exc)
    # gather=True (explicit): every rank gets every restart, through tensor all_gathers
    best2, _, liks2 = restarts.fit_distributed(model, train, runner=runner, gather=True, device=device)
    assert best2 == best and np.array_equal(liks2, liks)
    assert len(model.results) == 3 and list(model._restart_ids) == [0, 1, 2]
    for s in range(3):
        for key in ("theta", "eta", "pr"):
            assert np.array_equal(model.results[s][key], g[f"{{key}}_{{s}}"]), (s, key)
        assert model.results[s]["likelihood"] == g["likelihoods"][s]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


PREDICT_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    import torch.distributed as dist
    import fake_device
    from mmsbm_amd import restarts, mmsbm as host
    from conftest import load_golden

    host.HipEM = fake_device.FakeHipEM                  # the oracle behind the device interface
    host.load_backend = lambda name: (None, None, None, "hip")
    restarts.check_single_hip_runtime = lambda: None
    rank, world, local, device = restarts.init_from_env("gloo")
    g = load_golden("g2_c1_sampling3")
    train, test = g["train"], g["train"][::3]
    model = host.MMSBM(2, 2, iterations=10, sampling=3, seed=1)
    gather = {gather}
    best, best_lik, liks = restarts.fit_distributed(model, train, gather=gather, device=device)
    if gather:    # every rank holds all three restarts: each must still be counted ONCE in the mean
        assert list(model._restart_ids) == [0, 1, 2] and len(model.results) == 3
    else:
        assert [r for r in model._restart_ids] == ([0, 2] if rank == 0 else [1])
        assert len(model.results) == len(model._restart_ids)      # nobody holds the other rank's parameters
        model.data_handler = type("Enc", (), dict(transform=staticmethod(lambda d, log: d)))()
        try:      # a mean over this rank's share would differ from rank to rank: predict() says so instead
            model.predict(test)
            raise SystemExit("predict() on a partial model went through")
        except RuntimeError as exc:
            assert "predict_distributed" in str(exc)
        model.data_handler = None
    matrix = restarts.predict_distributed(model, test, device=device)
    # the same three restarts in ONE process
    solo = host.MMSBM(2, 2, iterations=10, sampling=3, seed=1)
    solo.fit_encoded(train)
    want, _, per_run = solo._predict_runs(test)
    assert np.allclose(matrix, want, rtol=1e-14, atol=0), np.max(np.abs(matrix - want))
    assert np.allclose(matrix.sum(axis=1), 1.0, atol=1e-12)       # a mean of distributions, not a multiple of it
    assert model.run_stats == per_run
    b = int(np.argmax([st["accuracy"] for st in per_run]))
    assert np.array_equal(model.theta.values, solo.results[b]["theta"])    # broadcast from its owner
    assert model.likelihood == solo.results[b]["likelihood"]
    st = model.score(silent=True)["stats"]
    assert st["accuracy"] == solo._compute_stats(want)["accuracy"]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.parametrize("gather", [False, True])
def test_two_rank_gloo_predict_without_gathering_parameters(tmp_path, gather):
    """restarts.predict_distributed: each rank scores ITS restarts, one all-reduce(SUM) of the (M, R) matrix;
    equals the one-process predict of the same restarts; the best-accuracy restart's objects are broadcast.
    After fit_distributed(gather=True) every rank holds every restart: each is still counted once (the
    lowest rank holding it scores it), so the matrix is the mean and its rows sum to 1."""
    script = tmp_path / "predict_worker.py"
    script.write_text(PREDICT_WORKER.format(root=ROOT, gather=gather))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert f"rank {rank} ok" in out


@pytest.mark.parametrize("gather", [False, True])
def test_four_rank_gloo_more_ranks_than_restarts(tmp_path, gather):
    """sampling = 3 on four ranks: rank 3 runs nothing and only takes part in the collectives (the likelihood
    all-reduce, the (M, R) all-reduce of predict_distributed, the broadcast of the best restart); the result is the
    one-process result."""
    script = tmp_path / "predict_worker4.py"
    text = PREDICT_WORKER.format(root=ROOT, gather=gather)
    text = text.replace("assert [r for r in model._restart_ids] == ([0, 2] if rank == 0 else [1])",
                        "assert [r for r in model._restart_ids] == [[0], [1], [2], []][rank]")
    script.write_text(text)
    port = _free_port()
    procs = []
    for rank in range(4):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="4", LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert f"rank {rank} ok" in out


EIGHT_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    import torch.distributed as dist
    import fake_device
    from mmsbm_amd import restarts, mmsbm as host
    from conftest import load_golden
    from oracle import mmsbm_oracle as orc

    host.HipEM = fake_device.FakeHipEM                  # the oracle behind the device interface
    host.load_backend = lambda name: (None, None, None, "hip")
    restarts.check_single_hip_runtime = lambda: None
    rank, world, local, device = restarts.init_from_env("gloo")
    assert world == 8
    g = load_golden("g2_c1_sampling3")
    train, test = g["train"], g["train"][::3]
    model = host.MMSBM(2, 2, iterations=10, sampling=8, seed=1)
    best, best_lik, liks = restarts.fit_distributed(model, train, device=device)
    assert list(model._restart_ids) == [rank] and len(model.results) == 1     # one restart per rank, nothing gathered
    want = orc.fit(train, 2, 2, iterations=10, sampling=8, seed=1)             # the same eight restarts in one process
    assert np.array_equal(liks, np.array([w["likelihood"] for w in want]))
    assert np.array_equal(liks[:3], g["likelihoods"])                          # restart i does not depend on `sampling`
    assert best == int(np.argmax(liks)) and best_lik == liks[best]
    for key in ("theta", "eta", "pr"):                                         # the winner, on every rank
        assert np.array_equal(model.best_result[key], want[best][key]), key
    assert np.array_equal(model.results[0]["theta"], want[rank]["theta"])
    matrix = restarts.predict_distributed(model, test, device=device)
    ref = np.mean([orc.prod_dist(test, w["theta"], w["eta"], w["pr"]) for w in want], axis=0)
    assert np.allclose(matrix, ref, rtol=1e-13, atol=0), np.max(np.abs(matrix - ref))
    assert len(model.run_stats) == 8
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_eight_rank_gloo_one_restart_per_rank(tmp_path):
    """BASELINE's configs 3 and 4 are `sampling = 8` with ONE restart per GPU.  The same job shape on eight gloo ranks
    (the oracle behind the device interface): every rank runs its restart, one all-reduce gives everyone all eight
    likelihoods, the winner's parameters arrive by tensor broadcast, predict averages over all eight with one
    all-reduce -- equal to the eight restarts run in one process."""
    script = tmp_path / "eight_worker.py"
    script.write_text(EIGHT_WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="8", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert f"rank {rank} ok" in out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_restart_sharding_and_likelihood_pick(tmp_path):
    """world_size 2 on CPU (gloo): restart i on rank i mod 2, ONE all-reduce picks the winner,
    results gathered in restart order and identical to the reference's sampling=3 run."""
    (tmp_path / "conftest_path.py").write_text(
        f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\nfrom conftest import load_golden\n")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=str(tmp_path),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank} ok" in out


def test_single_process_pick_without_process_group():
    from mmsbm_amd import restarts
    best, lik, liks = restarts.pick_max_likelihood({0: -5.0, 1: -2.5, 2: -9.0}, 3)
    assert best == 1 and lik == -2.5 and liks.tolist() == [-5.0, -2.5, -9.0]
    assert restarts.shard_restarts(8, 3, 4) == [3, 7] and restarts.shard_restarts(2, 5, 8) == []


def test_encoder_threaded_columns_match_the_sequential_path():
    """From a million rows on the three columns are factorised on three threads."""
    import pandas as pd
    import mmsbm_amd.encode as enc_mod
    rng = np.random.default_rng(1)
    n = 1_000_050
    df = pd.DataFrame({"users": rng.integers(0, 5000, n), "items": rng.integers(0, 700, n).astype(str),
                       "ratings": rng.integers(1, 6, n)})
    enc = enc_mod.Encoder()
    got = enc.fit_transform(df)
    assert got.dtype == np.int32 and got.flags["F_CONTIGUOUS"]
    for j in range(3):
        ids, labels = enc_mod._factorize_as_str(df.iloc[:, j].to_numpy())
        assert np.array_equal(got[:, j], ids) and enc.labels[j].tolist() == labels.tolist()


def test_result_shapes_from_the_training_set_also_an_empty_one():
    """restarts.result_shapes: every rank derives the shapes of a restart's theta / eta / pr from the training triples
    (no collective carries them); an empty training set gives empty tables instead of an exception."""
    from mmsbm_amd import restarts
    model = MMSBM(3, 4, iterations=1, sampling=1, seed=0)
    train = np.array([[0, 0, 0], [2, 1, 2], [1, 4, 2]], dtype=np.int64)
    assert restarts.result_shapes(model, train) == ((3, 3), (5, 4), (3, 4, 2))    # (rating 1 never occurs: two rating values)
    assert restarts.result_shapes(model, np.zeros((0, 3), dtype=np.int64)) == ((0, 3), (0, 4), (3, 4, 0))
