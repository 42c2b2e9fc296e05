"""Sample-sharded scoring across the GPUs of one node (one process per GPU, RCCL over xGMI).

Every output column depends on one input column (R/plaid.R:107, :634-642), so samples
shard embarrassingly: rank r owns a contiguous block of sample columns (contiguous bytes in
R's column-major layout) and a replica of the prepared membership G.  The only couplings
between shards are three SCALARS, each one small all-reduce:

    max(rX)            R/plaid.R:251   -> all_reduce(MAX)   (replaid.ssgsea)
    min(x) == 0        R/plaid.R:557   -> all_reduce(MAX) of the 0/1 flag words
    mean(medx)         R/plaid.R:572   -> all_reduce(SUM) of {sum, count}

and, when the caller wants one matrix, a final gather of the score shards to a root as
direct peer->root transfers (grouped send/recv: each of the root's xGMI links carries one
shard; a ring would be bound by a single link).

The arithmetic is delegated to a *phase engine* working on torch tensors: `HipPhaseEngine`
(the product: device pointers into the C ABI).  The collectives only need tensors, so the
same code runs under gloo with a stand-in engine in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous blocks of ceil(n/world) columns; trailing ranks may be short or empty."""
    per = -(-n // world) if world > 0 else n
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class CscShard:
    """This rank's sample columns of a dgCMatrix: the three CSC slots as tensors (`p` re-based to start at 0,
    int32; `i` int32; `x` float64) plus what the host knows about them without touching the device:
    nnz (p[-1]) and the longest column (sizes the rank launch)."""

    def __init__(self, p, i, x, g: int, nnz: int | None = None, max_col_nnz: int | None = None):
        self.p, self.i, self.x, self.g = p, i, x, int(g)
        self.n = int(p.shape[0]) - 1
        if nnz is None or max_col_nnz is None:                     # one small D2H at construction, none per step
            d = (p[1:] - p[:-1])
            nnz = int(p[-1].item()) if self.n > 0 else 0
            max_col_nnz = int(d.max().item()) if self.n > 0 else 0
        self.nnz, self.max_col_nnz = int(nnz), int(max_col_nnz)

    @staticmethod
    def from_scipy(X, lo: int, hi: int, device=None):
        """columns [lo, hi) of a scipy CSC matrix"""
        import torch
        p = np.asarray(X.indptr[lo:hi + 1], dtype=np.int64)
        i = np.asarray(X.indices[p[0]:p[-1]], dtype=np.int32)
        x = np.asarray(X.data[p[0]:p[-1]], dtype=np.float64)
        p = (p - p[0]).astype(np.int32)
        d = np.diff(p)
        t = [torch.from_numpy(np.ascontiguousarray(a)) for a in (p, i, x)]
        if device is not None:
            t = [a.to(device) for a in t]
        return CscShard(t[0], t[1], t[2], X.shape[0], int(p[-1]), int(d.max()) if len(d) else 0)


class HipPhaseEngine:
    """Phases of the hot path on one GPU, on torch CUDA tensors (float64, row-major
    (n_local, g) == column-major g x n_local)."""

    def __init__(self, ctx, geneset, device):
        import torch
        self.torch = torch
        self.ctx, self.gs, self.device = ctx, geneset, device
        self._fused = None      # (weak reference to the S of the last fused crossprod, its launch token): medians() resumes only for that tensor

    def _same_stream(self):
        # the library enqueues on ITS stream: tensors allocated, zeroed or all-reduced on another stream would race
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if self.ctx.stream is None or int(self.ctx.stream) != int(cur):
            raise RuntimeError("HipPhaseEngine: the plaidhip context must enqueue on torch's current stream "
                               f"(context stream {self.ctx.stream}, torch current stream {cur}): create the Context with "
                               "stream=<that stream>.cuda_stream and call inside `with torch.cuda.stream(<that stream>)`")

    def spmm_csc(self, X: CscShard, stat="mean", alpha=1.0, beta=0.0, alpha_div=None, flags=None, values=None,
                 rank_weights=False, normalize=False):
        """crossprod with a CSC shard; `values` replaces X.x (e.g. the ranks of the stored values).  `rank_weights`: the
        values lie in [0, *alpha_div] (rank^power and their global maximum): order-independent fixed-point sums.
        `normalize`: the caller will call medians(S, flags) on the result: the crossprod may classify its scores for them"""
        self._same_stream()
        t = self.torch
        S = t.empty((X.n, self.gs.m), dtype=t.float64, device=self.device)
        self._fused = None
        if X.n > 0:
            xx = X.x if values is None else values
            fl = flags.data_ptr() if flags is not None else None
            div = alpha_div.data_ptr() if alpha_div is not None else None
            if fl is not None and normalize:
                # a normalising caller: the crossprod also classifies its scores for the medians
                # (plaidhip_dev_spmm_csc_fused_f64; the plain kernels when the shapes do not call for it) -- medians() below
                # finishes them after the flag words have been all-reduced.  The token ties the pending candidates to THIS
                # launch and the weak reference to THIS tensor: an S that merely sits at the same address never matches.
                import weakref
                token = self.ctx.dev_spmm_csc_fused(self.gs, X.p.data_ptr(), X.i.data_ptr(), xx.data_ptr(), X.n, S.data_ptr(),
                                                    self.gs.m, stat, alpha, beta, fl, div, div if rank_weights else None, nnz=X.nnz)
                self._fused = (weakref.ref(S), token) if token else None
            elif rank_weights and alpha_div is not None:
                self.ctx.dev_spmm_csc_ranks(self.gs, X.p.data_ptr(), X.i.data_ptr(), xx.data_ptr(), X.n, S.data_ptr(),
                                            self.gs.m, div, stat, alpha, beta, fl, nnz=X.nnz)
            else:
                self.ctx.dev_spmm_csc(self.gs, X.p.data_ptr(), X.i.data_ptr(), xx.data_ptr(), X.n, S.data_ptr(), self.gs.m,
                                      stat, alpha, beta, fl, div, nnz=X.nnz)
        return S

    def sparse_colranks(self, X: CscShard, ties="average", signed=False, power=1.0):
        """sparse_colranks() of the shard: ranks of the stored values (same pattern) and max(rX) of the shard"""
        self._same_stream()
        t = self.torch
        Rx = t.empty_like(X.x)
        colmax = t.zeros(max(X.n, 1), dtype=t.float64, device=self.device)
        gmax = t.zeros(1, dtype=t.float64, device=self.device)       # implicit zeros: max(rX) >= 0
        if X.n > 0:
            self.ctx.dev_colranks_csc(X.p.data_ptr(), X.x.data_ptr(), X.n, X.max_col_nnz, Rx.data_ptr(), ties, signed,
                                      power, colmax.data_ptr())
            self.ctx.dev_max(colmax.data_ptr(), X.n, gmax.data_ptr())
        return Rx, gmax

    def colranks_csc_dense(self, X: CscShard, ties="average", signed=False, power=1.0, rows=None):
        """colranks(X sparse, keep.zero = FALSE) of the shard (R/plaid.R:602-609): the zeros are ranked, dense (cells, genes)
        result, built from the ranks of the stored values (any number of genes); columns with more than 20,352 stored
        values take the densify-and-rank entry.  `rows`: columns [lo, hi) of the shard (a panel)"""
        self._same_stream()
        t = self.torch
        lo, hi = (0, X.n) if rows is None else rows
        n = hi - lo
        ld = X.g + (X.g & 1)
        R = t.empty((n, ld), dtype=t.float64, device=self.device)
        if n == 0:
            return R
        p = X.p[lo:hi + 1]
        if X.max_col_nnz <= self.ctx.limit("sparse_rank_column"):
            Rx = t.empty(max(X.nnz, 1), dtype=t.float64, device=self.device)
            self.ctx.dev_colranks_csc_dense_nz(p.data_ptr(), X.i.data_ptr(), X.x.data_ptr(), X.g, n, X.max_col_nnz,
                                               Rx.data_ptr(), R.data_ptr(), ld, ties, signed, power)
        else:
            self.ctx.dev_colranks_csc_dense(p.data_ptr(), X.i.data_ptr(), X.x.data_ptr(), X.g, n, R.data_ptr(), ld, ties,
                                            signed, power)
        return R

    def new_flags(self):
        return self.torch.zeros(4, dtype=self.torch.int32, device=self.device)

    def spmm(self, X, stat="mean", alpha=1.0, beta=0.0, alpha_div=None, flags=None, ranks=False, normalize=False):
        """`ranks`: X is what colranks() wrote with power 1 (unsigned): the exact u16-staged crossprod takes it.
        `normalize`: the caller will call medians(S, flags) on the result: the crossprod may classify its scores for them"""
        self._same_stream()
        t = self.torch
        n = X.shape[0]
        S = t.empty((n, self.gs.m), dtype=t.float64, device=self.device)
        self._fused = None
        fl = flags.data_ptr() if flags is not None else None
        div = alpha_div.data_ptr() if alpha_div is not None else None
        if normalize and not ranks and fl is not None and n > 0:
            import weakref
            token = self.ctx.dev_spmm_dense_fused(self.gs, X.data_ptr(), X.shape[1], n, S.data_ptr(), self.gs.m, stat, alpha, beta, fl, div)
            self._fused = (weakref.ref(S), token) if token else None
            return S
        call = self.ctx.dev_spmm_ranks if ranks else self.ctx.dev_spmm_dense
        call(self.gs, X.data_ptr(), X.shape[1], n, S.data_ptr(), self.gs.m, stat, alpha, beta, fl, div)
        return S

    def colranks(self, X, ties="average", signed=False, power=1.0):
        self._same_stream()
        t = self.torch
        n, g = X.shape
        R = t.empty_like(X)
        colmax = t.empty(max(n, 1), dtype=t.float64, device=self.device)
        self.ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R.data_ptr(), g, ties, signed, power, colmax.data_ptr())
        gmax = t.full((1,), -np.inf, dtype=t.float64, device=self.device)
        if n > 0:
            self.ctx.dev_max(colmax.data_ptr(), n, gmax.data_ptr())
        return R, gmax

    def medians(self, S, flags):
        self._same_stream()
        t = self.torch
        n, m = S.shape
        med = t.empty(max(n, 1), dtype=t.float64, device=self.device)
        red = t.zeros(2, dtype=t.float64, device=self.device)
        if n > 0:
            # dev_col_medians -- unless this very tensor came out of a fused crossprod whose candidates are still pending on the
            # context: then only unresolved columns are swept
            fused, self._fused = self._fused, None
            token = fused[1] if (fused is not None and fused[0]() is S) else 0
            self.ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr(), token=token)
            self.ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
        return med, red

    def shift(self, S, med, red):
        self._same_stream()
        n, m = S.shape
        if n > 0:
            self.ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())

    def shift_cast(self, S, med, red, out=None):
        """(S - med[col] + mean(med)) as float32 into `out` (allocated when None); S stays as the crossprod wrote it.
        S: (rows, m) float64 rows of a score block, med: the medians of exactly those rows, red: {sum, count} of ALL medians"""
        self._same_stream()
        n, m = S.shape
        if out is None:
            out = self.torch.empty((n, m), dtype=self.torch.float32, device=S.device)
        if n > 0:
            assert S.is_contiguous() and out.is_contiguous() and med.is_contiguous()
            self.ctx.dev_shift_columns_cast_f32(S.data_ptr(), m, m, n, med.data_ptr(), out.data_ptr(), m, 0.0, red.data_ptr())
        return out


    def row_group_sums(self, A, y):
        """A (n_local, rows) row-major == rows x n_local column-major; y int32 (n_local,) of 0 / 1 -> (2, rows) sums"""
        self._same_stream()
        t = self.torch
        n, rows = A.shape
        out = t.zeros((2, rows), dtype=t.float64, device=self.device)
        if rows > 0:
            self.ctx.dev_row_group_sums(A.data_ptr(), rows, rows, n, y.data_ptr(), out.data_ptr())
        return out

    def row_group_ssd(self, A, y, mean):
        """sums of squared deviations of the rows of A from `mean` (2, rows), per group -> (2, rows)"""
        self._same_stream()
        t = self.torch
        n, rows = A.shape
        out = t.zeros((2, rows), dtype=t.float64, device=self.device)
        if rows > 0:
            self.ctx.dev_row_group_ssd(A.data_ptr(), rows, rows, n, y.data_ptr(), mean.data_ptr(), out.data_ptr())
        return out

    def crossprod_sum(self, F):
        """F (k, g) -> (k, m): t(G != 0) %*% F, the plain sums (R/plaid.R:478-479)"""
        return self.spmm(F, "sum", 1.0, 0.0, None, None)


def _world(group):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _collective(group):
    """whether the cross-shard reductions go through the process group: whenever one is initialised -- a group of ONE rank
    included, so that a single GPU runs the very RCCL calls (device tensors, int32 MAX / fp64 SUM) the 8-GPU job makes"""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def sharded_plaid(engine, X_local, stat="mean", normalize=True, alpha=1.0, beta=0.0, alpha_div=None,
                  group=None, x_is_ranks=False, defer_shift=False):
    """plaid() body (R/plaid.R:73-85) on this rank's sample shard; returns the local
    (n_local, m) score block.  Collective: every rank of `group` must call it.
    defer_shift (with normalize): the sweep of R/plaid.R:572 is left to the gather -- returns (S un-shifted, med, red) for
    gather_scores(..., dtype=float32, shift=(engine, med, red)), which applies it while it casts each slab."""
    import torch.distributed as dist
    world, _ = _world(group)
    flags = engine.new_flags()
    S = engine.spmm(X_local, stat, alpha, beta, alpha_div, flags, ranks=x_is_ranks, normalize=normalize)
    if normalize:
        if _collective(group):
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)       # min(x) == 0 over all samples
        med, red = engine.medians(S, flags)
        if _collective(group):
            dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)         # mean(medx) over all samples
        if defer_shift:
            return S, med, red
        engine.shift(S, med, red)
    return S


def sharded_plaid_csc(engine, X_local: "CscShard", stat="mean", normalize=True, alpha=1.0, beta=0.0, alpha_div=None,
                      values=None, group=None, rank_weights=False, defer_shift=False):
    """plaid() on a CSC shard (sparse branch of Matrix::crossprod, R/plaid.R:107); same collectives as sharded_plaid"""
    import torch.distributed as dist
    world, _ = _world(group)
    flags = engine.new_flags()
    S = engine.spmm_csc(X_local, stat, alpha, beta, alpha_div, flags, values, rank_weights=rank_weights, normalize=normalize)
    if normalize:
        if _collective(group):
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
        med, red = engine.medians(S, flags)
        if _collective(group):
            dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
        if defer_shift:
            return S, med, red
        engine.shift(S, med, red)
    return S


def sharded_ssgsea_csc(engine, X_local: "CscShard", alpha=0.0, group=None, defer_shift=False):
    """replaid.ssgsea on a dgCMatrix shard (BASELINE config 5's workload): sparse_colranks of the stored values
    (R/plaid.R:600-601, 631-650), rank^(1+alpha), global max(rX) = one all_reduce(MAX), then the crossprod with the
    `/max - 0.5` folded into its epilogue (the -0.5 reaches the implicit zeros too) and the median normalisation"""
    import torch.distributed as dist
    world, _ = _world(group)
    Rx, gmax = engine.sparse_colranks(X_local, "average", False, 1.0 + alpha)
    if _collective(group):
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    return sharded_plaid_csc(engine, X_local, "mean", True, 1.0, -0.5, gmax, Rx, group, rank_weights=True, defer_shift=defer_shift)


def sharded_sing(engine, X_local, group=None):
    """replaid.sing (R/plaid.R:213-219): no cross-shard coupling at all."""
    R, _ = engine.colranks(X_local, "min")
    return sharded_plaid(engine, R, "mean", False, 1.0 / X_local.shape[1], -0.5, None, group, x_is_ranks=True)


def sharded_sing_csc(engine, X_local: "CscShard", group=None, panel_bytes: int = 2 << 30):
    """replaid.sing on a dgCMatrix shard: colranks(X, ties.method = "min") ranks the zeros too (R/plaid.R:602-609), so the
    rank matrix is dense -- built panel by panel from the ranks of the stored values and multiplied at once; the shard
    itself never exists densely.  No cross-shard coupling."""
    import torch
    g, n = X_local.g, X_local.n
    panel = max(2, (panel_bytes // max(1, (g + (g & 1)) * 8)) & ~1)
    out = []
    for lo in range(0, max(n, 1), panel):
        hi = min(n, lo + panel)
        R = engine.colranks_csc_dense(X_local, "min", rows=(lo, hi))
        out.append(engine.spmm(R, "mean", 1.0 / g, -0.5, None, None, ranks=True))
    return out[0] if len(out) == 1 else torch.cat(out, dim=0)


def sharded_ssgsea(engine, X_local, alpha=0.0, group=None):
    """replaid.ssgsea (R/plaid.R:244-255): global max(rX) is one all_reduce(MAX)."""
    import torch.distributed as dist
    world, _ = _world(group)
    R, gmax = engine.colranks(X_local, "average", False, 1.0 + alpha)
    if _collective(group):
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    return sharded_plaid(engine, R, "mean", True, 1.0, -0.5, gmax, group, x_is_ranks=(alpha == 0.0))


def sharded_plaid_test(engine, X_local, y_local, Gp, tests=("one", "two", "lm"), metap_method="fisher", gsetX_local=None,
                       group=None):
    """plaid.test (R/plaid.R:392-474) on sample shards: every statistic it needs is a row-wise sum over the samples.

        fc = rowMeans(X[, y == 1]) - rowMeans(X[, y == 0])    :407-409  -> all_reduce(SUM) of the (2, g) group sums
        G^T fc, G^T fc^2                                       :478-479  -> computed on every rank (two columns)
        gsetX = plaid(X, G)                                    :424-427  -> sharded_plaid (its own three scalars)
        Welch per set over the score rows                      :429-431  -> all_reduce(SUM) of the (2, m) group sums, then of
                                                                            the (2, m) sums of squared deviations from the
                                                                            GLOBAL means (two passes, like the one-device call)

    X_local: (n_local, g) dense shard; y_local: int32 (n_local,) of 0 / 1; gsetX_local: this rank's (n_local, m) scores or
    None (computed).  Returns the sets x 6 table (gsetFC, p.one, p.two, p.lm, p.meta, q.meta; G's column order) as a numpy
    array on EVERY rank.  Collective: every rank of `group` must call it."""
    import torch
    import torch.distributed as dist
    from .engine import plaid_test_finish
    world, _ = _world(group)
    bits = sum({"one": 1, "two": 2, "lm": 4}[t_] for t_ in tests)
    mm = {"fisher": 0, "sumlog": 0, "stouffer": 1, "sumz": 1}
    if metap_method not in mm:
        raise ValueError(f"Invalid method: {metap_method}")                   # R/plaid.R:533
    # checked on the caller's dtype (0.5 or NaN must not be truncated into a legal label), and by all ranks together: the
    # rank holding the bad label must not be the only one that leaves before the collectives below
    y_ok = torch.tensor([1 if (y_local.numel() == 0 or bool(((y_local == 0) | (y_local == 1)).all())) else 0], dtype=torch.int32,
                        device=y_local.device)
    if _collective(group):
        dist.all_reduce(y_ok, op=dist.ReduceOp.MIN, group=group)
    if not bool(y_ok.item()):
        raise ValueError("elements of y must be 0 or 1")                      # R/plaid.R:394
    y_local = y_local.to(torch.int32)
    g = X_local.shape[1]
    cnt = torch.stack([(y_local == 0).sum(), (y_local == 1).sum()]).to(torch.float64)

    def allsum(t_):
        if _collective(group):
            dist.all_reduce(t_, op=dist.ReduceOp.SUM, group=group)
        return t_

    cnt = allsum(cnt)
    inv = torch.where(cnt > 0, 1.0 / cnt, torch.full_like(cnt, float("nan")))   # (an empty group: NaN means, as in R)
    mean = allsum(engine.row_group_sums(X_local, y_local)) * inv[:, None]
    fc = mean[1] - mean[0]
    F = torch.stack([fc, fc * fc]).contiguous()
    T = engine.crossprod_sum(F)                                               # (2, m), the same on every rank
    tot = F.sum(dim=1)
    SM = None
    if bits & 4:
        S = gsetX_local if gsetX_local is not None else sharded_plaid(engine, X_local, group=group)
        smean = allsum(engine.row_group_sums(S, y_local)) * inv[:, None]
        ssd = allsum(engine.row_group_ssd(S, y_local, smean.contiguous()))
        SM = torch.cat([smean, ssd]).cpu().numpy()
    n0, n1 = (int(v) for v in cnt.cpu().numpy())
    return plaid_test_finish(g, Gp, T.cpu().numpy(), float(tot[0]), float(tot[1]), SM, n0, n1, bits, mm[metap_method])


class GatherRefused(RuntimeError):
    """The requested gather cannot complete with the memory there is (code EUNSUPPORTED of the C ABI): nothing was
    allocated or sent.  `.needed` / `.available` are in bytes."""

    code = 4   # PLAIDHIP_EUNSUPPORTED

    def __init__(self, message: str, needed: int, available: int):
        super().__init__(f"plaidhip error 4: {message}")
        self.needed, self.available = int(needed), int(available)


class GatherError(RuntimeError):
    """A rank failed while the score matrix was being assembled on the host: raised on EVERY rank (the ranks exchange an ok
    flag before the root maps the files), the files are gone."""


def _free_device_bytes(t):
    if t.is_cuda:
        import torch
        return int(torch.cuda.mem_get_info(t.device)[0])
    return None


def _free_shm_bytes(directory="/dev/shm"):
    import os
    st = os.statvfs(directory)
    return int(st.f_bavail) * int(st.f_frsize)


def gather_scores(S_local, n_total: int, dst: int = 0, group=None, to: str = "device", dtype=None,
                  chunk_rows: int | None = None, max_bytes: int | None = None, shm_dir: str = "/dev/shm", shift=None):
    """Reassemble the (n_total, m) score matrix on `dst` from the per-rank blocks laid out by shard_bounds()
    (the reference fills ONE matrix chunk by chunk, R/plaid.R:110-119).  Collective over `group`.

    to="device"  direct peer->root transfers into one tensor on the root's device: each of the root's xGMI links carries
                 one shard (a ring would be bound by a single link).  The blocks travel in slabs of `chunk_rows` rows,
                 one grouped batch of send/recv per slab, so an optional cast (`dtype=torch.float32`: half the bytes on
                 the links and on the root) needs a slab of scratch, not a second copy of the shard.  The root refuses
                 (GatherRefused, EUNSUPPORTED) BEFORE allocating when n_total * m * itemsize exceeds its free device
                 memory: config 5 (1e6 cells x 50,000 sets) is 400 GB in fp64 -- more than one GPU's 288 GB.
    to="host"    every rank copies its block device->host into ITS rows of one host matrix shared between the processes
                 (a file under `shm_dir`, unlinked once everyone has mapped it): each GPU uses its own PCIe link, no GPU
                 ever holds more than its shard -- what completes at config 5.  Returns a numpy array on dst.
    shift=(engine, med, red) with dtype=torch.float32: S_local is the UN-shifted block a `defer_shift=True` call returned
                 and the sweep of R/plaid.R:572 is applied while each slab is cast (`engine.shift_cast`: one read of the
                 fp64 block and a half-size write, instead of shift_columns' read + write over the whole 50 GB shard
                 followed by the cast's read).  S_local is left un-shifted.  to="host" applies the shift in place first.
    `max_bytes` overrides the free-memory probe (tests).  Returns the full matrix on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = _world(group)
    if to not in ("device", "host"):
        raise ValueError("gather_scores: to must be 'device' or 'host'")
    out_dtype = dtype or S_local.dtype
    m = int(S_local.shape[1])
    itemsize = torch.empty((), dtype=out_dtype).element_size()
    need = int(n_total) * m * itemsize
    eng = med = red = None
    if shift is not None:
        eng, med, red = shift
        if to == "host" or out_dtype != torch.float32:
            # (the host lanes copy on side streams of their own, and an fp64 result has no cast to fuse with: plain sweep)
            eng.shift(S_local, med, red)
            eng = None

    def cast(block, r0, r1, out=None):
        """rows [r0, r1) of this rank's block in the result's dtype (into `out` when given), shifted if the gather shifts"""
        if eng is not None:
            return eng.shift_cast(block[r0:r1], med[r0:r1], red, out)
        if out is not None:
            out.copy_(block[r0:r1])
            return out
        slab_ = block[r0:r1]
        return slab_.contiguous() if out_dtype == slab_.dtype else slab_.to(out_dtype)

    rows = chunk_rows or max(1, (256 << 20) // max(1, m * 8))          # ~256 MB slabs
    if world == 1 and to == "device":
        if eng is None:
            return S_local if out_dtype == S_local.dtype else S_local.to(out_dtype)
        full1 = torch.empty((int(S_local.shape[0]), m), dtype=out_dtype, device=S_local.device)
        for r0 in range(0, int(S_local.shape[0]), rows):
            r1 = min(int(S_local.shape[0]), r0 + rows)
            cast(S_local, r0, r1, full1[r0:r1])
        return full1
    if to == "host":
        return _gather_to_host(S_local, n_total, dst, group, out_dtype, rows, max_bytes, shm_dir, need)
    # ---- device: the root decides, everyone learns the verdict (a refusal must not leave the peers in a send) ----
    verdict = torch.zeros(2, dtype=torch.int64)
    if rank == dst:
        lo, hi = shard_bounds(n_total, world, rank)
        have = max_bytes if max_bytes is not None else _free_device_bytes(S_local)
        if have is not None and need > have:
            verdict[0], verdict[1] = 1, have
    if world > 1:
        v = verdict.to(S_local.device) if dist.get_backend(group) == "nccl" else verdict
        dist.broadcast(v, src=dst, group=group)
        verdict = v.cpu()
    if int(verdict[0]):
        raise GatherRefused(f"gather_scores(to='device'): the {n_total} x {m} matrix needs {need / 1e9:.1f} GB on rank {dst}, "
                            f"{int(verdict[1]) / 1e9:.1f} GB are free; use to='host' or dtype=torch.float32", need, int(verdict[1]))
    full = None
    if rank == dst:
        full = torch.empty((n_total, m), dtype=out_dtype, device=S_local.device)
        lo, hi = shard_bounds(n_total, world, rank)
        for r0 in range(lo, hi, rows):
            r1 = min(hi, r0 + rows)
            cast(S_local, r0 - lo, r1 - lo, full[r0:r1])
    per = max(shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world))
    for s0 in range(0, per, rows):                       # slab s of every peer's block in one grouped batch
        ops, keep = [], []
        if rank == dst:
            for src in range(world):
                if src == dst:
                    continue
                lo, hi = shard_bounds(n_total, world, src)
                r0, r1 = lo + s0, min(hi, lo + s0 + rows)
                if r1 > r0:
                    ops.append(dist.P2POp(dist.irecv, full[r0:r1], src, group))
        else:
            r0, r1 = s0, min(S_local.shape[0], s0 + rows)
            if r1 > r0:
                slab = cast(S_local, r0, r1)
                keep.append(slab)
                ops.append(dist.P2POp(dist.isend, slab, dst, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
    return full


class _StitchedFiles:
    """One shared host matrix made of ONE /dev/shm FILE PER RANK.  A single file is a single inode, and every first touch of
    a page of it takes that inode's lock: however many ranks and threads write, a fresh shm file fills at 3.5-6 GB/s in
    total, and slower the more writers there are (tools/ubench/shm_fill.cpp: 0.5 GB/s with 4 x 32 threads); files of their
    own fill at 13-20 GB/s with four processes.  File k holds the bytes [cut[k], cut[k + 1]) of the matrix, the cuts being
    the ranks' first bytes rounded DOWN to a page: a rank's block lies in its own file except for its last partial page,
    which is the head of the next file.  The root maps the files back to back (MAP_FIXED into one reserved address range),
    which is one contiguous matrix again."""
    PAGE = 4096

    def __init__(self, base_path, n_total, m, itemsize, world, parts=1):
        """`parts` files per rank: a rank's block is cut into that many runs of rows, each a file (and a writer) of its own"""
        self.base_path, self.ranks, self.parts = base_path, world, parts
        self.world = world * parts                   # number of files
        self.row_bytes = m * itemsize
        self.total = n_total * self.row_bytes
        self.first_row = []                          # first row of file k (rank k // parts, part k % parts)
        for r in range(world):
            lo, hi = shard_bounds(n_total, world, r)
            per = -(-(hi - lo) // parts) if hi > lo else 0
            self.first_row += [min(hi, lo + j * per) for j in range(parts)]
        firsts = [r0 * self.row_bytes for r0 in self.first_row]
        self.cut = [0] + [(b // self.PAGE) * self.PAGE for b in firsts[1:]] + [-(-self.total // self.PAGE) * self.PAGE]
        self.maps = {}

    def rows_of(self, k, n_total):
        """the rows [r0, r1) file k's writer is responsible for"""
        r0 = self.first_row[k]
        rank = k // self.parts
        hi = shard_bounds(n_total, self.ranks, rank)[1]
        r1 = self.first_row[k + 1] if (k + 1) % self.parts != 0 else hi
        return r0, max(r0, r1)

    def path(self, k):
        return f"{self.base_path}.{k}"

    def size(self, k):
        return self.cut[k + 1] - self.cut[k]

    def create_rank(self, rank):
        for k in range(rank * self.parts, (rank + 1) * self.parts):
            self.create(k)

    def create(self, k):
        """file k, sized (its pages come with allocate(): posix_fallocate, 20 instead of 13 GB/s for the fill that follows)"""
        import os
        if self.size(k) <= 0:
            return
        fd = os.open(self.path(k), os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
        try:
            os.ftruncate(fd, self.size(k))
        finally:
            os.close(fd)

    def allocate(self, k):
        """(file k's writer, before it writes) the file's pages in one go; the writers of a rank do this side by side"""
        import os
        if self.size(k) <= 0:
            return
        fd = os.open(self.path(k), os.O_RDWR)
        try:
            os.posix_fallocate(fd, 0, self.size(k))
        except OSError:
            pass
        finally:
            os.close(fd)

    def _map(self, k):
        mm = self.maps.get(k)
        if mm is None:                               # (two writers may meet at a file's head page: the later mapping wins, both work)
            mm = np.memmap(self.path(k), mode="r+", dtype=np.uint8, shape=(self.size(k),))
            self.maps[k] = mm
        return mm

    def write(self, byte_off, src_u8):
        """src_u8 (a flat uint8 view) -> bytes [byte_off, byte_off + len) of the matrix, across file boundaries"""
        pos, end = byte_off, byte_off + src_u8.shape[0]
        k = max(0, min(self.world - 1, np.searchsorted(self.cut, pos, side="right") - 1))
        while pos < end:
            while self.size(k) <= 0 or self.cut[k + 1] <= pos:
                k += 1
            stop = min(end, self.cut[k + 1])
            self._map(k)[pos - self.cut[k]:stop - self.cut[k]] = src_u8[pos - byte_off:stop - byte_off]
            pos = stop

    def close_maps(self):
        self.maps.clear()

    def stitch(self, np_dtype, shape):
        """(root) every file mapped at its place in one reserved address range -> one ndarray; the files are unlinked (the
        mappings outlive the names), the range is unmapped when the array is collected"""
        import ctypes
        import mmap as _mmap
        import os
        import weakref
        libc = ctypes.CDLL(None, use_errno=True)
        libc.mmap.restype = ctypes.c_void_p
        libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
        libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        span = max(self.cut[-1], self.PAGE)
        MAP_FAILED = ctypes.c_void_p(-1).value
        base = libc.mmap(None, span, 0, _mmap.MAP_PRIVATE | _mmap.MAP_ANONYMOUS, -1, 0)
        if base in (None, MAP_FAILED):
            raise OSError(ctypes.get_errno(), "mmap (address range of the gathered matrix)")
        MAP_FIXED = 0x10
        try:
            for k in range(self.world):
                if self.size(k) <= 0:
                    continue
                fd = os.open(self.path(k), os.O_RDWR)
                try:
                    got = libc.mmap(base + self.cut[k], self.size(k), _mmap.PROT_READ | _mmap.PROT_WRITE,
                                    _mmap.MAP_SHARED | MAP_FIXED, fd, 0)
                finally:
                    os.close(fd)
                if got in (None, MAP_FAILED):
                    raise OSError(ctypes.get_errno(), f"mmap of {self.path(k)}")
        except BaseException:
            libc.munmap(base, span)
            raise
        finally:
            self.unlink()
        buf = (ctypes.c_char * max(self.total, 1)).from_address(base)
        arr = np.frombuffer(buf, dtype=np_dtype, count=shape[0] * shape[1]).reshape(shape)
        weakref.finalize(buf, libc.munmap, base, span)   # (arr -> buf: the range lives as long as any view of it)
        return arr

    def unlink(self):
        import os
        for k in range(self.world):
            try:
                os.unlink(self.path(k))
            except OSError:
                pass


def _gather_to_host(S_local, n_total, dst, group, out_dtype, rows, max_bytes, shm_dir, need):
    import os
    import time
    import torch
    import torch.distributed as dist
    world, rank = _world(group)
    m = int(S_local.shape[1])
    np_dtype = {torch.float64: np.float64, torch.float32: np.float32}[out_dtype]
    itemsize = np.dtype(np_dtype).itemsize
    # the root names the files (or refuses), every rank creates its own
    msg = [None]
    if rank == dst:
        have = max_bytes if max_bytes is not None else _free_shm_bytes(shm_dir)
        if need > have:
            msg[0] = ("refused", need, have)
        else:
            msg[0] = ("ok", os.path.join(shm_dir, f"plaidhip_gather_{os.getpid()}_{time.time_ns()}"))
    if world > 1:
        dist.broadcast_object_list(msg, src=dst, group=group)
    if msg[0][0] == "refused":
        raise GatherRefused(f"gather_scores(to='host'): the {n_total} x {m} matrix needs {msg[0][1] / 1e9:.1f} GB of shared host "
                            f"memory under {shm_dir}, {msg[0][2] / 1e9:.1f} GB are free", msg[0][1], msg[0][2])
    # eight files (and writers) per rank when the block is large: a writer moves ~2.4 GB/s into fresh shm pages, and one
    # file takes ~5.5 GB/s whoever writes it
    # (about sixteen writers on the node: 4 ranks x 8 writers filled 96 GB at 23 GB/s, 4 x 4 at 34 GB/s -- past that the
    # kernel's page allocation is what they wait for)
    parts = int(os.environ.get("PLAIDHIP_GATHER_PARTS",
                               max(2, min(8, 16 // max(world, 1))) if S_local.is_cuda and need // max(world, 1) >= (1 << 30) else 1))
    parts = max(1, parts)
    if world > 1:                                   # (every rank must cut alike)
        pl = [parts]
        dist.broadcast_object_list(pl, src=dst, group=group)
        parts = pl[0]
    files = _StitchedFiles(msg[0][1], int(n_total), m, itemsize, world, parts)
    lo, hi = shard_bounds(n_total, world, rank)
    nloc = hi - lo
    failure = None

    def all_ok(err):
        """collective: every rank learns whether any rank failed (and why) -- nobody is left in a barrier"""
        if world <= 1:
            return err
        seen = [None] * world
        dist.all_gather_object(seen, None if err is None else f"rank {rank}: {type(err).__name__}: {err}", group=group)
        bad = [s for s in seen if s is not None]
        return bad[0] if bad else None

    try:
        files.create_rank(rank)
    except Exception as exc:
        failure = exc
    bad = all_ok(failure)                            # (doubles as the barrier: a block's last partial page lives in the NEXT file)
    if bad is not None:
        files.unlink()
        raise GatherError(f"gather_scores(to='host'): creating the files failed ({bad})") from failure
    try:
        if nloc > 0:
            if S_local.is_cuda:
                # one lane per file of this rank: its own side stream and two pinned slabs -- the copy of slab k+1 runs on
                # the bus while slab k is moved into the lane's file (numpy releases the GIL while it copies, the event wait
                # does too); the lanes run side by side
                from concurrent.futures import ThreadPoolExecutor
                cur = torch.cuda.current_stream(S_local.device)
                dev = S_local.device

                def lane(k):
                    r0f, r1f = files.rows_of(k, int(n_total))
                    if r1f <= r0f:
                        return
                    torch.cuda.set_device(dev)
                    files.allocate(k)
                    side = torch.cuda.Stream(device=dev)
                    side.wait_stream(cur)
                    nr = min(max(1, rows // parts), r1f - r0f)
                    pin = [torch.empty((nr, m), dtype=out_dtype).pin_memory() for _ in range(2)]
                    evs = [None, None]
                    slabs = [(a0, min(r1f, a0 + nr)) for a0 in range(r0f, r1f, nr)]

                    def issue(i):
                        a0, a1 = slabs[i]
                        with torch.cuda.stream(side):
                            pin[i & 1][:a1 - a0].copy_(S_local[a0 - lo:a1 - lo], non_blocking=True)
                            evs[i & 1] = torch.cuda.Event()
                            evs[i & 1].record(side)
                    issue(0)
                    for i, (a0, a1) in enumerate(slabs):
                        evs[i & 1].synchronize()
                        if i + 1 < len(slabs):
                            issue(i + 1)
                        files.write(a0 * files.row_bytes, pin[i & 1][:a1 - a0].numpy().reshape(-1).view(np.uint8))
                mine = list(range(rank * parts, (rank + 1) * parts))
                if parts == 1:
                    lane(mine[0])
                else:
                    with ThreadPoolExecutor(parts) as pool:
                        for f_ in [pool.submit(lane, k) for k in mine]:
                            f_.result()
            else:
                src = S_local if out_dtype == S_local.dtype else S_local.to(out_dtype)
                for k in range(rank * parts, (rank + 1) * parts):
                    files.allocate(k)
                files.write(lo * files.row_bytes, np.ascontiguousarray(src.numpy()).reshape(-1).view(np.uint8))
        files.close_maps()
    except Exception as exc:
        failure = exc
    except BaseException:
        # KeyboardInterrupt / SystemExit on this rank: no collective any more (the others time out in all_ok), but the
        # tens of GB this rank may have put under /dev/shm must not outlive it
        files.unlink()
        raise
    bad = all_ok(failure)                            # (and the barrier before the root maps what the others wrote)
    if bad is not None:
        files.unlink()                               # (no names are left behind in /dev/shm; every rank raises alike)
        raise GatherError(f"gather_scores(to='host'): a rank failed while writing its block ({bad})") from failure
    if rank == dst:
        return files.stitch(np_dtype, (int(n_total), m))
    return None
