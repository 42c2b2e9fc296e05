"""The sample-sharded host path (plaid_amd/sharded.py) under gloo with world_size 2 on CPU.

The collectives (flag MAX, {sum,count} SUM, max(rX) MAX, peer->root gather) are the product
code; the per-shard arithmetic is supplied by a stand-in phase engine built on the CPU oracle,
so this checks exactly what cannot be checked on a 1-GPU box: that shards + scalar
all-reduces reproduce the unsharded result."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import plaid_oracle as po
from plaid_amd import sharded, synth


class OraclePhaseEngine:
    """Same phase interface as sharded.HipPhaseEngine, on CPU tensors, via the oracle."""

    def __init__(self, g, Gp, Gi):
        self.g, self.m = g, len(Gp) - 1
        self.G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, self.m))
        self.k = np.diff(Gp).astype(np.float64)

    def new_flags(self):
        return torch.zeros(4, dtype=torch.int32)

    def spmm(self, X, stat="mean", alpha=1.0, beta=0.0, alpha_div=None, flags=None, ranks=False, normalize=False):
        Xn = X.numpy().T[:self.g]                               # g x n_local (a padded leading dimension is cut off)
        raw = np.asarray(self.G.T @ Xn)
        w = 1.0 / (1e-8 + self.k) if stat == "mean" else np.ones_like(self.k)
        a = alpha / float(alpha_div.item()) if alpha_div is not None else alpha
        S = a * (raw * w[:, None]) + beta * (self.k * w)[:, None]
        if flags is not None and S.size:
            flags[0] = max(int(flags[0]), int((S < 0).any()))
            flags[1] = max(int(flags[1]), int((S == 0).any()))
            flags[2] = max(int(flags[2]), int(np.isnan(S).any()))
        return torch.from_numpy(np.ascontiguousarray(S.T))

    def colranks(self, X, ties="average", signed=False, power=1.0):
        R = po.colranks(X.numpy().T, signed=signed, ties_method=ties) if X.shape[0] else np.zeros((self.g, 0))
        R = R ** power if power != 1.0 else R
        gmax = torch.tensor([R.max() if R.size else -np.inf], dtype=torch.float64)
        return torch.from_numpy(np.ascontiguousarray(R.T)), gmax

    def _epilogue(self, raw, stat, alpha, beta, alpha_div, flags):
        w = 1.0 / (1e-8 + self.k) if stat == "mean" else np.ones_like(self.k)
        a = alpha / float(alpha_div.item()) if alpha_div is not None else alpha
        with np.errstate(all="ignore"):
            S = a * (raw * w[:, None]) + beta * (self.k * w)[:, None]
        if flags is not None and S.size:
            flags[0] = max(int(flags[0]), int((S < 0).any()))
            flags[1] = max(int(flags[1]), int((S == 0).any()))
            flags[2] = max(int(flags[2]), int(np.isnan(S).any()))
        return torch.from_numpy(np.ascontiguousarray(S.T))

    def spmm_csc(self, X, stat="mean", alpha=1.0, beta=0.0, alpha_div=None, flags=None, values=None, rank_weights=False,
                 normalize=False):
        xx = (X.x if values is None else values).numpy()
        Xs = sp.csc_matrix((xx, X.i.numpy(), X.p.numpy()), shape=(self.g, X.n))
        return self._epilogue(np.asarray((self.G.T @ Xs).todense()), stat, alpha, beta, alpha_div, flags)

    def colranks_csc_dense(self, X, ties="average", signed=False, power=1.0, rows=None):
        lo, hi = (0, X.n) if rows is None else rows
        Xs = sp.csc_matrix((X.x.numpy(), X.i.numpy(), X.p.numpy()), shape=(self.g, X.n))[:, lo:hi]
        R = po.colranks(Xs.toarray(), signed=signed, ties_method=ties) if hi > lo else np.zeros((self.g, 0))
        R = R ** power if power != 1.0 else R
        ld = self.g + (self.g & 1)
        out = np.zeros((hi - lo, ld))
        out[:, :self.g] = R.T
        return torch.from_numpy(out)

    def sparse_colranks(self, X, ties="average", signed=False, power=1.0):
        Xs = sp.csc_matrix((X.x.numpy(), X.i.numpy(), X.p.numpy()), shape=(self.g, X.n))
        R = po.sparse_colranks(Xs, signed=signed, ties_method=ties).data if X.nnz else np.zeros(0)
        R = R ** power if power != 1.0 else R
        gmax = torch.tensor([max(R.max(), 0.0) if R.size else 0.0], dtype=torch.float64)
        return torch.from_numpy(np.ascontiguousarray(R, dtype=np.float64)), gmax

    def medians(self, S, flags):
        ignore_zero = bool(flags[1]) and not bool(flags[0])
        if S.shape[0] == 0:
            return torch.zeros(1, dtype=torch.float64), torch.zeros(2, dtype=torch.float64)
        _, med = po.normalize_medians(S.numpy().T, ignore_zero)
        ok = ~np.isnan(med)
        return torch.from_numpy(med), torch.tensor([med[ok].sum(), float(ok.sum())], dtype=torch.float64)

    def shift(self, S, med, red):
        if S.shape[0]:
            S.sub_(med[:, None]).add_(float(red[0] / red[1]))

    def shift_cast(self, S, med, red, out=None):
        r = ((S - med[:, None]) + float(red[0] / red[1])).to(torch.float32)
        if out is None:
            return r
        out.copy_(r)
        return out


    # plaid.test's row-wise reductions (HipPhaseEngine.row_group_sums / row_group_ssd / crossprod_sum)
    def row_group_sums(self, A, y):
        An, yn = A.numpy(), y.numpy()
        return torch.from_numpy(np.stack([An[yn == 0].sum(axis=0), An[yn == 1].sum(axis=0)]))

    def row_group_ssd(self, A, y, mean):
        An, yn, mu = A.numpy(), y.numpy(), mean.numpy()
        return torch.from_numpy(np.stack([((An[yn == k] - mu[k]) ** 2).sum(axis=0) for k in (0, 1)]))

    def crossprod_sum(self, F):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(self.G.T @ F.numpy().T[:self.g]).T))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, case, out_path):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g, m = 300, 41
        Gp, Gi = synth.geneset_csc(g, m, kmin=3, kmax=60)
        X = np.round(synth.dense_columns(g, 0, n), 1)
        if case == "zeros":
            X[:, :] = np.abs(X - 8.0)
            Gi = Gi.copy()
        lo, hi = sharded.shard_bounds(n, world, rank)
        Xl = torch.from_numpy(np.ascontiguousarray(X[:, lo:hi].T))
        eng = OraclePhaseEngine(g, Gp, Gi)
        res = {}
        res["plaid"] = sharded.sharded_plaid(eng, Xl)
        res["sing"] = sharded.sharded_sing(eng, Xl)
        res["ssgsea"] = sharded.sharded_ssgsea(eng, Xl, alpha=0.25)
        # the sparse workload of config 5: CSC shards, sparse_colranks, same scalars
        Xz = np.where(np.random.default_rng(5).random(X.shape) < 0.85, 0.0, X)
        if case == "zeros":
            Xz[:, n - 1] = 0.0                                     # a cell without stored values
        shard = sharded.CscShard.from_scipy(sp.csc_matrix(Xz), lo, hi)
        res["plaid_csc"] = sharded.sharded_plaid_csc(eng, shard)
        res["ssgsea_csc"] = sharded.sharded_ssgsea_csc(eng, shard, alpha=0.25)
        res["sing_csc"] = sharded.sharded_sing_csc(eng, shard, panel_bytes=2 * g * 8 * 2)   # panels of 4 cells
        full = {k: sharded.gather_scores(v, n, dst=0) for k, v in res.items()}
        # the gathers that complete at config 5's size: slabs of a few rows, to the host (one shared matrix, every rank
        # writes its rows) and to the device with an fp32 cast; and the refusals, raised on EVERY rank before any transfer
        host = sharded.gather_scores(res["plaid"], n, dst=0, to="host", chunk_rows=3)
        host32 = sharded.gather_scores(res["ssgsea"], n, dst=0, to="host", dtype=torch.float32, chunk_rows=2)
        dev32 = sharded.gather_scores(res["plaid"], n, dst=0, dtype=torch.float32, chunk_rows=2)
        dev_slabs = sharded.gather_scores(res["sing"], n, dst=0, chunk_rows=1)
        refused = 0
        for kw in ({"to": "device", "max_bytes": 8}, {"to": "host", "max_bytes": 8}):
            try:
                sharded.gather_scores(res["plaid"], n, dst=0, **kw)
            except sharded.GatherRefused as exc:
                refused += int(exc.code == 4 and exc.needed == n * 41 * 8 and exc.available == 8)
        assert refused == 2
        assert (host is None) == (rank != 0) and (dev32 is None) == (rank != 0)
        if rank == 0:
            assert isinstance(host, np.ndarray) and host.shape == (n, 41) and host32.dtype == np.float32
            np.savez(out_path, host=np.asarray(host).T, host32=np.asarray(host32).T, dev32=dev32.numpy().T,
                     dev_slabs=dev_slabs.numpy().T, **{k: v.numpy().T for k, v in full.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,case", [(11, "plain"), (8, "plain"), (1, "plain"), (7, "zeros")])
def test_sharded_equals_unsharded_gloo(tmp_path, n, case):
    world = 2
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, case, out), nprocs=world, join=True)
    got = np.load(out)
    g, m = 300, 41
    Gp, Gi = synth.geneset_csc(g, m, kmin=3, kmax=60)
    X = np.round(synth.dense_columns(g, 0, n), 1)
    if case == "zeros":
        X[:, :] = np.abs(X - 8.0)
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [str(k) for k in range(g)]
    np.testing.assert_allclose(got["plaid"], po.plaid(X, rn, G, rn), rtol=1e-10, atol=1e-12)
    assert np.array_equal(got["host"], got["plaid"]) and np.array_equal(got["dev_slabs"], got["sing"])
    assert np.array_equal(got["dev32"], got["plaid"].astype(np.float32))
    assert np.array_equal(got["host32"], got["ssgsea"].astype(np.float32))
    np.testing.assert_allclose(got["sing"], po.replaid_sing(X, rn, G, rn), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(got["ssgsea"], po.replaid_ssgsea(X, rn, G, rn, alpha=0.25), rtol=1e-10, atol=1e-12)
    Xz = np.where(np.random.default_rng(5).random(X.shape) < 0.85, 0.0, X)
    if case == "zeros":
        Xz[:, n - 1] = 0.0
    Xs = sp.csc_matrix(Xz)
    np.testing.assert_allclose(got["plaid_csc"], po.plaid(Xs, rn, G, rn), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(got["ssgsea_csc"], po.replaid_ssgsea(Xs, rn, G, rn, alpha=0.25), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(got["sing_csc"], po.replaid_sing(Xz, rn, G, rn), rtol=1e-10, atol=1e-12)   # zeros ranked


def test_csc_shard_bookkeeping():
    """CscShard re-bases the column pointers and knows nnz / the longest column without a device round trip per step"""
    X = sp.random(50, 9, density=0.3, format="csc", random_state=1)
    sh = sharded.CscShard.from_scipy(X, 3, 7)
    assert sh.n == 4 and int(sh.p[0]) == 0 and sh.nnz == X[:, 3:7].nnz
    assert sh.max_col_nnz == int(np.diff(X.indptr[3:8]).max())
    assert np.array_equal(sh.i.numpy(), X[:, 3:7].indices) and np.array_equal(sh.x.numpy(), X[:, 3:7].data)
    empty = sharded.CscShard.from_scipy(X, 9, 9)
    assert empty.n == 0 and empty.nnz == 0 and empty.max_col_nnz == 0


def test_single_process_path_needs_no_process_group():
    g, m, n = 120, 9, 5
    Gp, Gi = synth.geneset_csc(g, m, kmin=3, kmax=30)
    X = synth.dense_columns(g, 0, n)
    eng = OraclePhaseEngine(g, Gp, Gi)
    S = sharded.sharded_plaid(eng, torch.from_numpy(np.ascontiguousarray(X.T)))
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [str(k) for k in range(g)]
    np.testing.assert_allclose(S.numpy().T, po.plaid(X, rn, G, rn), rtol=1e-12)
    assert sharded.gather_scores(S, n) is S


def _plaid_test_worker(rank, world, port, n, out_path):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g, m = 300, 41
        Gp, Gi = synth.geneset_csc(g, m, kmin=3, kmax=60)
        X = np.round(synth.dense_columns(g, 0, n), 1)
        y = (np.arange(n) % 3 == 0).astype(np.int32)
        X[:40, y == 1] += 1.5                                           # a real group difference in the first genes
        lo, hi = sharded.shard_bounds(n, world, rank)
        eng = OraclePhaseEngine(g, Gp, Gi)
        Xl = torch.from_numpy(np.ascontiguousarray(X[:, lo:hi].T))
        yl = torch.from_numpy(y[lo:hi])
        res = {}
        for mp_ in ("fisher", "stouffer"):
            res[mp_] = sharded.sharded_plaid_test(eng, Xl, yl, Gp, ("one", "two", "lm"), mp_)
        res["one_lm"] = sharded.sharded_plaid_test(eng, Xl, yl, Gp, ("one", "lm"), "stouffer")
        res["two"] = sharded.sharded_plaid_test(eng, Xl, yl, Gp, ("two",))
        # scores handed in (a caller that has them already): this rank's rows of plaid(X, G)
        S_all = sharded.sharded_plaid(eng, Xl)
        res["given"] = sharded.sharded_plaid_test(eng, Xl, yl, Gp, ("lm",), gsetX_local=S_all)
        np.savez(out_path + f".{rank}.npz", **res)
    finally:
        dist.destroy_process_group()


def _gather3_worker(rank, world, port, n, m, out_path):
    os.environ["PLAIDHIP_GATHER_PARTS"] = "3"          # three files per rank: page cuts inside and between the blocks
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        lo, hi = sharded.shard_bounds(n, world, rank)
        S = torch.from_numpy(np.arange(lo, hi, dtype=np.float64)[:, None] * 1000.0 + np.arange(m, dtype=np.float64)[None, :])
        full = sharded.gather_scores(S, n, dst=1, to="host", chunk_rows=5)
        f32 = sharded.gather_scores(S, n, dst=1, to="host", dtype=torch.float32)
        assert (full is None) == (rank != 1)
        if rank == 1:
            np.savez(out_path, full=np.asarray(full), f32=np.asarray(f32))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,m", [(1000, 1037), (7, 513), (2, 4099)])
def test_host_gather_three_ranks_three_files_each(tmp_path, n, m):
    """gather_scores(to="host") with world size 3, a root that is not rank 0, and three /dev/shm files per rank: rows of
    8,296 / 4,104 / 32,792 bytes put every cut inside a page and some blocks inside one page; n = 2 leaves a rank empty"""
    world = 3
    out = str(tmp_path / "g3.npz")
    mp.spawn(_gather3_worker, args=(world, _free_port(), n, m, out), nprocs=world, join=True)
    got = np.load(out)
    exp = np.arange(n, dtype=np.float64)[:, None] * 1000.0 + np.arange(m, dtype=np.float64)[None, :]
    assert np.array_equal(got["full"], exp)
    assert np.array_equal(got["f32"], exp.astype(np.float32))
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("plaidhip_gather_")]


@pytest.mark.parametrize("n", [23, 9])
def test_sharded_plaid_test_equals_the_oracle_gloo(tmp_path, n):
    """plaid.test over two sample shards (R/plaid.R:392-474): group sums and sums of squared deviations all-reduced, the
    host half (plaidhip_plaid_test_finish, no device) run on every rank -- against the oracle on the whole matrix; n = 9
    leaves the second rank 4 columns and ONE sample of group 1 on it"""
    world = 2
    out = str(tmp_path / "pt")
    mp.spawn(_plaid_test_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    g, m = 300, 41
    Gp, Gi = synth.geneset_csc(g, m, kmin=3, kmax=60)
    X = np.round(synth.dense_columns(g, 0, n), 1)
    y = (np.arange(n) % 3 == 0).astype(np.int32)
    X[:40, y == 1] += 1.5
    G = sp.csc_matrix((np.ones(len(Gi)), Gi, Gp), shape=(g, m))
    rn = [str(k) for k in range(g)]
    cols = ["gsetFC", "p.one", "p.two", "p.lm", "p.meta", "q.meta"]
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    for key, tests, mp_ in (("fisher", ("one", "two", "lm"), "fisher"), ("stouffer", ("one", "two", "lm"), "stouffer"),
                            ("one_lm", ("one", "lm"), "stouffer"), ("two", ("two",), "fisher"), ("given", ("lm",), "fisher")):
        assert np.array_equal(r0[key], r1[key], equal_nan=True)        # every rank holds the same table
        exp = po.plaid_test(X, rn, y, G, rn, None, metap_method=mp_, tests=tests)
        for k, name in enumerate(cols):
            if name in exp:
                np.testing.assert_allclose(r0[key][:, k], exp[name], rtol=1e-8, atol=1e-300, err_msg=f"{key} {name}")
            else:
                assert np.isnan(r0[key][:, k]).all()


def _shift_in_gather_worker(rank, world, port, n, out_path):
    """ssGSEA on a dgCMatrix shard with the sweep of R/plaid.R:572 deferred to the gather (cast-with-shift per slab) against
    the same call that shifts in place and lets the gather only cast"""
    g, m = 300, 70
    Gp, Gi = synth.geneset_csc(g, m, kmin=3, kmax=60)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        eng = OraclePhaseEngine(g, Gp, Gi)
        Xs = sp.random(g, n, density=0.2, format="csc", random_state=3, data_rvs=lambda k: np.round(np.random.default_rng(4).gamma(2.0, 1.0, k), 1) + 0.1)
        lo, hi = sharded.shard_bounds(n, world, rank)
        shard = sharded.CscShard.from_scipy(Xs, lo, hi)
        S_plain = sharded.sharded_ssgsea_csc(eng, shard, alpha=0.25)
        full_a = sharded.gather_scores(S_plain, n, dst=0, to="device", dtype=torch.float32, chunk_rows=3)
        S_raw, med, red = sharded.sharded_ssgsea_csc(eng, shard, alpha=0.25, defer_shift=True)
        keep = S_raw.clone()
        full_b = sharded.gather_scores(S_raw, n, dst=0, to="device", dtype=torch.float32, chunk_rows=3, shift=(eng, med, red))
        assert torch.equal(S_raw, keep)                    # the gather leaves the un-shifted block alone
        full_c = sharded.gather_scores(S_raw.clone(), n, dst=0, to="device", shift=(eng, med, red))   # fp64: plain sweep, then gather
        if rank == 0:
            np.savez(out_path, a=full_a.numpy(), b=full_b.numpy(), c=full_c.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [11, 4])
def test_gather_that_shifts_while_it_casts_equals_shift_then_cast_gloo(tmp_path, n):
    """config 5's per-GPU flow without the shift pass: gather_scores(dtype=float32, shift=(engine, med, red)) on the
    un-shifted block == shift in place, then gather with a cast (two ranks, slabs of three rows, uneven blocks)"""
    out = str(tmp_path / "sg.npz")
    mp.spawn(_shift_in_gather_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out)
    assert got["a"].dtype == np.float32 and got["a"].shape[0] == n
    assert np.array_equal(got["a"], got["b"])
    assert np.array_equal(got["c"].astype(np.float32), got["a"])
