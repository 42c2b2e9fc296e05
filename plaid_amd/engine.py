"""Thin object layer over the C ABI: a device context, a prepared gene-set handle, and the
device-level / host-level calls.  Pointers are plain integers (`tensor.data_ptr()`,
`ndarray.ctypes.data`); no tensor types cross the boundary."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import STAT, TIES, check


def _np_ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _as_f64_fortran(a) -> np.ndarray:
    return np.asfortranarray(a, dtype=np.float64)


_INT32_MAX = 2**31 - 1


def _as_i32(a) -> np.ndarray:
    """dgCMatrix-style index arrays are 32-bit at the boundary (include/plaidhip.h).  scipy hands over int64
    for large matrices: refuse, instead of wrapping, anything a 32-bit slot cannot hold (more than 2^31-1
    stored values: split the matrix by columns first, as `chunked_crossprod` does, R/plaid.R:100-123)."""
    a = np.asarray(a)
    if a.dtype != np.int32 and a.size:
        lo, hi = int(a.min()), int(a.max())
        if lo < -_INT32_MAX - 1 or hi > _INT32_MAX:
            raise _lib.PlaidHipError(_lib.EUNSUPPORTED,
                                     f"index value {hi if hi > _INT32_MAX else lo} does not fit the 32-bit dgCMatrix slots of the "
                                     "C ABI (more than 2^31-1 stored values?): split the matrix by columns")
    return np.ascontiguousarray(a, dtype=np.int32)


class Geneset:
    """Device-resident prepared membership (plaidhip_geneset)."""

    def __init__(self, ctx: "Context", g: int, Gp, Gi):
        Gp = _as_i32(Gp)
        Gi = _as_i32(Gi)
        self.ctx = ctx
        self.g = int(g)
        self.m = int(len(Gp) - 1)
        self.sizes = np.diff(Gp).astype(np.int64)
        h = C.c_void_p()
        check(ctx.lib.plaidhip_geneset_create(ctx.handle, self.g, self.m, _np_ptr(Gp), _np_ptr(Gi), C.byref(h)))
        self.handle = h

    def info(self) -> dict:
        buf = (C.c_int64 * 8)()
        check(self.ctx.lib.plaidhip_geneset_info(self.handle, buf))
        return {"g": buf[0], "m": buf[1], "z": buf[2], "padded_slots": buf[3], "tiles": buf[4],
                "gene_slices": buf[5], "waves": buf[6], "padded_slots_pair": buf[7]}

    def close(self):
        if self.handle:
            self.ctx.lib.plaidhip_geneset_destroy(self.handle)
            self.handle = None

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass


class Context:
    """plaidhip_ctx: one device + one stream.  `stream` is a raw hipStream_t value (e.g.
    `torch.cuda.current_stream().cuda_stream`; 0 is the device's null stream, which is what torch's default
    stream is) or None for a private non-blocking stream."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.plaidhip_init(int(device), None, C.byref(h)))
        self.handle = h
        self.device = int(device)
        self.stream = None
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream: int):
        """enqueue on this hipStream_t from now on (0: the null stream)"""
        check(self.lib.plaidhip_set_stream(self.handle, C.c_void_p(int(stream)) if stream else None))
        self.stream = int(stream)

    def set_option(self, name: str, value):
        """kernel-selection knobs (plaidhip_set_option): see _lib.OPTIONS"""
        code, values = _lib.OPTIONS[name]
        check(self.lib.plaidhip_set_option(self.handle, code, values[value] if isinstance(value, str) else int(value)))

    def limit(self, name: str) -> int:
        """size limits a host routes by (plaidhip_limit): "sparse_rank_column", "lds_genes" """
        v = C.c_int64(0)
        check(self.lib.plaidhip_limit({"sparse_rank_column": 1, "lds_genes": 2}[name], C.byref(v)))
        return int(v.value)

    def close(self):
        if self.handle:
            self.lib.plaidhip_finalize(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.plaidhip_synchronize(self.handle))

    def set_precision(self, mode: str):
        """"f64" (default): fp64 throughout.  "mixed": the dense crossprod stages the sample columns as fp32
        (inputs rounded to 2^-24 relative, sums fp64) -- about 2x the SpMM rate, scores within ~1e-7."""
        check(self.lib.plaidhip_set_precision(self.handle, {"f64": 0, "mixed": 1}[mode]))

    def geneset(self, g: int, Gp, Gi) -> Geneset:
        return Geneset(self, g, Gp, Gi)

    # ---- device-level (raw device pointers) ------------------------------------------
    def dev_spmm_dense(self, gs: Geneset, X: int, ldx: int, n: int, S: int, lds: int, stat="mean",
                       alpha=1.0, beta=0.0, flags: int | None = None, alpha_div: int | None = None):
        check(self.lib.plaidhip_dev_spmm_dense_f64(self.handle, gs.handle, X, ldx, n, STAT[stat], alpha,
                                                   alpha_div, beta, S, lds, flags))

    def dev_spmm_dense_fused(self, gs: Geneset, X: int, ldx: int, n: int, S: int, lds: int, stat="mean",
                             alpha=1.0, beta=0.0, flags: int | None = None, alpha_div: int | None = None) -> int:
        """dev_spmm_dense that also classifies the scores for normalize_medians while it writes them (the fp64 pair
        kernel, more than 6,144 sets per column, >= 1e9 scores or fused_medians = on); dev_col_medians_resume then
        finishes the medians without a second pass over S.  Returns the launch's token (0: the plain route ran)"""
        check(self.lib.plaidhip_dev_spmm_dense_fused_f64(self.handle, gs.handle, X, ldx, n, STAT[stat], alpha,
                                                         alpha_div, beta, S, lds, flags))
        return self.dev_fused_medians_token()

    def dev_spmm_ranks(self, gs: Geneset, R: int, ldr: int, n: int, S: int, lds: int, stat="mean",
                       alpha=1.0, beta=0.0, flags: int | None = None, alpha_div: int | None = None):
        """the crossprod of a RANK matrix (what dev_colranks_dense wrote with power 1, unsigned): u16 staging, integer
        sums -- bit-identical to dev_spmm_dense on the same input"""
        check(self.lib.plaidhip_dev_spmm_ranks_f64(self.handle, gs.handle, R, ldr, n, STAT[stat], alpha,
                                                   alpha_div, beta, S, lds, flags))

    def dev_spmm_csc(self, gs: Geneset, Xp: int, Xi: int, Xx: int, n: int, S: int, lds: int,
                     stat="mean", alpha=1.0, beta=0.0, flags: int | None = None,
                     alpha_div: int | None = None, nnz: int = -1):
        """nnz: stored values of X when the caller knows it (one kernel is launched), -1: decided on the device"""
        check(self.lib.plaidhip_dev_spmm_csc_f64(self.handle, gs.handle, Xp, Xi, Xx, n, int(nnz), STAT[stat], alpha,
                                                 alpha_div, beta, S, lds, flags))

    def dev_spmm_csc_ranks(self, gs: Geneset, Xp: int, Xi: int, Rx: int, n: int, S: int, lds: int, rmax: int,
                           stat="mean", alpha=1.0, beta=0.0, flags: int | None = None, nnz: int = -1):
        """the sparse crossprod of rank weights (0 <= Rx <= *rmax, what dev_colranks_csc wrote; alpha is divided by
        *rmax): order-independent fixed-point sums in the scatter kernel"""
        check(self.lib.plaidhip_dev_spmm_csc_ranks_f64(self.handle, gs.handle, Xp, Xi, Rx, n, int(nnz), STAT[stat], alpha,
                                                       rmax, beta, S, lds, flags))

    def dev_spmm_csc_fused(self, gs: Geneset, Xp: int, Xi: int, Xx: int, n: int, S: int, lds: int, stat="mean", alpha=1.0,
                           beta=0.0, flags: int | None = None, alpha_div: int | None = None, rmax: int | None = None,
                           nnz: int = -1):
        """dev_spmm_csc (rmax None) / dev_spmm_csc_ranks (rmax set) that also classifies the scores for
        normalize_medians while it writes them; dev_col_medians_resume then finishes the medians without a second pass
        over S (plaidhip_dev_spmm_csc_fused_f64)"""
        check(self.lib.plaidhip_dev_spmm_csc_fused_f64(self.handle, gs.handle, Xp, Xi, Xx, n, int(nnz), STAT[stat], alpha,
                                                       alpha_div, beta, S, lds, flags, rmax))
        return self.dev_fused_medians_token()

    def dev_col_medians_resume(self, S: int, lds: int, m: int, n: int, ignore_zero, med: int, flags: int | None = None,
                               token: int | None = None):
        """dev_col_medians for the S the last dev_spmm_csc_fused on this context wrote.  `token` (what dev_spmm_csc_fused
        returned): the candidates are used only while they are the pending ones of that very launch; None = the caller
        resumes directly after the crossprod (S recognised by pointer and shape)"""
        iz = -1 if ignore_zero is None else int(bool(ignore_zero))
        if token is None:
            check(self.lib.plaidhip_dev_col_medians_resume(self.handle, S, lds, m, n, iz, flags, med))
        else:
            check(self.lib.plaidhip_dev_col_medians_resume_token(self.handle, int(token), S, lds, m, n, iz, flags, med))

    def dev_fused_medians_discard(self):
        check(self.lib.plaidhip_dev_fused_medians_discard(self.handle))

    def dev_fused_medians_token(self) -> int:
        """token of the pending fused launch (0: none -- the plain route ran, or it was consumed / superseded)"""
        return self.dev_fused_medians_info()[3]

    def dev_fused_medians_info(self):
        """(columns of the last fused crossprod or 0, device pointer of status[n], device pointer of the calibration,
        token of the pending launch or 0)"""
        buf = (C.c_int64 * 4)()
        check(self.lib.plaidhip_dev_fused_medians_info(self.handle, buf))
        return int(buf[0]), int(buf[1]), int(buf[2]), int(buf[3])

    def dev_crossprod_weighted(self, Wp: int, Wi: int, Wx: int, g: int, m: int, Y: int, ldy: int, n: int, S: int,
                               lds: int):
        """t(x) %*% y for a sparse x with arbitrary stored values (device dgCMatrix slots), y dense"""
        check(self.lib.plaidhip_dev_crossprod_weighted_f64(self.handle, Wp, Wi, Wx, int(g), int(m), Y, int(ldy), int(n),
                                                           S, int(lds)))

    def dev_crossprod_weighted_csc(self, Wp: int, Wi: int, Wx: int, g: int, m: int, Yp: int, Yi: int, Yx: int,
                                   n: int, S: int, lds: int):
        check(self.lib.plaidhip_dev_crossprod_weighted_csc_f64(self.handle, Wp, Wi, Wx, int(g), int(m), Yp, Yi, Yx,
                                                               int(n), S, int(lds)))

    def dev_colranks_dense(self, X: int, ldx: int, g: int, n: int, R: int, ldr: int, ties="average",
                           signed=False, power=1.0, colmax: int | None = None):
        check(self.lib.plaidhip_dev_colranks_dense_f64(self.handle, X, ldx, g, n, TIES[ties], int(signed),
                                                       power, R, ldr, colmax))

    def dev_colranks_csc(self, Xp: int, Xx: int, n: int, max_col_nnz: int, Rx: int, ties="average", signed=False,
                         power=1.0, colmax: int | None = None):
        """max_col_nnz: upper bound on the stored values of a column (sizes the launch; nrow(X) is always valid)"""
        check(self.lib.plaidhip_dev_colranks_csc_f64(self.handle, Xp, Xx, n, int(max_col_nnz), TIES[ties], int(signed),
                                                     power, Rx, colmax))

    def dev_colranks_csc_dense_nz(self, Xp: int, Xi: int, Xx: int, g: int, n: int, max_col_nnz: int, Rx_scratch: int, R: int,
                                  ldr: int, ties="average", signed=False, power=1.0, colmax: int | None = None):
        """dense ranks of CSC columns (zeros ranked) from the ranks of the stored values: any nrow(X); Rx_scratch: Xp[n]
        doubles; every column at most 20,352 stored values"""
        check(self.lib.plaidhip_dev_colranks_csc_dense_nz_f64(self.handle, Xp, Xi, Xx, int(g), int(n), int(max_col_nnz),
                                                              TIES[ties], int(signed), power, Rx_scratch, R, int(ldr), colmax))

    def dev_minflags(self, S: int, count: int, flags: int):
        check(self.lib.plaidhip_dev_minflags(self.handle, S, count, flags))

    def dev_col_medians(self, S: int, lds: int, m: int, n: int, ignore_zero, med: int,
                        flags: int | None = None):
        """ignore_zero: True / False, or None to resolve min(x)==0 on the device from `flags`."""
        iz = -1 if ignore_zero is None else int(bool(ignore_zero))
        check(self.lib.plaidhip_dev_col_medians(self.handle, S, lds, m, n, iz, flags, med))

    def dev_sum(self, v: int, count: int, out: int):
        check(self.lib.plaidhip_dev_sum(self.handle, v, count, out))

    def dev_max(self, v: int, count: int, out: int):
        check(self.lib.plaidhip_dev_max(self.handle, v, count, out))

    def dev_shift_columns(self, S: int, lds: int, m: int, n: int, med: int, add: float = 0.0,
                          red: int | None = None):
        """x - med[col] + add; with `red` (device {sum, count}) add = sum/count on the device."""
        check(self.lib.plaidhip_dev_shift_columns(self.handle, S, lds, m, n, med, float(add), red))

    def dev_shift_columns_cast_f32(self, S: int, lds: int, m: int, n: int, med: int, out: int, ldo: int, add: float = 0.0,
                                   red: int | None = None):
        """out (float32) = x - med[col] + add, S untouched: the shift fused with the cast a sharded gather makes"""
        check(self.lib.plaidhip_dev_shift_columns_cast_f32(self.handle, S, lds, m, n, med, float(add), red, out, ldo))

    def dev_row_group_sums(self, A: int, ld: int, rows: int, n: int, y: int, sums: int):
        """per-row sums over the columns with y == 0 / y == 1 -> sums[2][rows] (plaid.test, R/plaid.R:407-408, 431)"""
        check(self.lib.plaidhip_dev_row_group_sums(self.handle, A, ld, rows, n, y, sums))

    def dev_row_group_ssd(self, A: int, ld: int, rows: int, n: int, y: int, mean: int, ssd: int):
        """per-row sums of squared deviations from the given group means -> ssd[2][rows] (R/plaid.R:429)"""
        check(self.lib.plaidhip_dev_row_group_ssd(self.handle, A, ld, rows, n, y, mean, ssd))

    # ---- host-level (numpy in, numpy out; the library stages through HBM) -------------
    @staticmethod
    def _result(out, m, n):
        """the caller's own result buffer (any byte offset; Fortran order like an R matrix) or a fresh one"""
        if out is None:
            return np.empty((m, n), dtype=np.float64, order="F")
        if out.shape != (m, n) or out.dtype != np.float64 or not out.flags.f_contiguous or not out.flags.writeable:
            raise ValueError(f"out: a writeable Fortran-ordered float64 array of shape {(m, n)}")
        return out

    def plaid_dense(self, X, Gp, Gi, stat="mean", normalize=True, out=None) -> np.ndarray:
        X = _as_f64_fortran(X)
        g, n = X.shape
        Gp, Gi = _as_i32(Gp), _as_i32(Gi)
        m = len(Gp) - 1
        S = self._result(out, m, n)
        check(self.lib.plaidhip_plaid_dense(self.handle, _np_ptr(X), g, n, _np_ptr(Gp), _np_ptr(Gi), m,
                                            STAT[stat], int(bool(normalize)), _np_ptr(S)))
        return S

    def plaid_csc(self, Xp, Xi, Xx, g: int, Gp, Gi, stat="mean", normalize=True, out=None) -> np.ndarray:
        Xp, Xi = _as_i32(Xp), _as_i32(Xi)
        Xx = np.ascontiguousarray(Xx, dtype=np.float64)
        n = len(Xp) - 1
        Gp, Gi = _as_i32(Gp), _as_i32(Gi)
        m = len(Gp) - 1
        S = self._result(out, m, n)
        check(self.lib.plaidhip_plaid_csc(self.handle, _np_ptr(Xp), _np_ptr(Xi), _np_ptr(Xx), int(g), n,
                                          _np_ptr(Gp), _np_ptr(Gi), m, STAT[stat], int(bool(normalize)),
                                          _np_ptr(S)))
        return S

    def crossprod_weighted(self, Wp, Wi, Wx, g: int, Y=None, Yp=None, Yi=None, Yx=None) -> np.ndarray:
        """chunked_crossprod's t(x) %*% y for a sparse x with arbitrary stored values (R/plaid.R:100-123); y dense
        (`Y`, g x n) or its dgCMatrix slots"""
        Wp, Wi = _as_i32(Wp), _as_i32(Wi)
        Wx = np.ascontiguousarray(Wx, dtype=np.float64)
        m = len(Wp) - 1
        if Y is not None:
            Y = _as_f64_fortran(Y)
            n = Y.shape[1]
            S = np.empty((m, n), dtype=np.float64, order="F")
            check(self.lib.plaidhip_crossprod_weighted_dense(self.handle, _np_ptr(Wp), _np_ptr(Wi), _np_ptr(Wx), int(g), m,
                                                             _np_ptr(Y), n, _np_ptr(S)))
            return S
        Yp, Yi = _as_i32(Yp), _as_i32(Yi)
        Yx = np.ascontiguousarray(Yx, dtype=np.float64)
        n = len(Yp) - 1
        S = np.empty((m, n), dtype=np.float64, order="F")
        check(self.lib.plaidhip_crossprod_weighted_csc(self.handle, _np_ptr(Wp), _np_ptr(Wi), _np_ptr(Wx), int(g), m,
                                                       _np_ptr(Yp), _np_ptr(Yi), _np_ptr(Yx), n, _np_ptr(S)))
        return S

    def sing_csc(self, Xp, Xi, Xx, g: int, Gp, Gi) -> np.ndarray:
        """replaid.sing for a dgCMatrix X (zeros are ranked, R/plaid.R:215-217 with colranks' sparse branch :602-609)"""
        Xp, Xi = _as_i32(Xp), _as_i32(Xi)
        Xx = np.ascontiguousarray(Xx, dtype=np.float64)
        n = len(Xp) - 1
        Gp, Gi = _as_i32(Gp), _as_i32(Gi)
        m = len(Gp) - 1
        S = np.empty((m, n), dtype=np.float64, order="F")
        check(self.lib.plaidhip_sing_csc(self.handle, _np_ptr(Xp), _np_ptr(Xi), _np_ptr(Xx), int(g), n, _np_ptr(Gp),
                                         _np_ptr(Gi), m, _np_ptr(S)))
        return S

    def normalize_medians(self, S, ignore_zero=None):
        S = np.array(S, dtype=np.float64, order="F", copy=True)
        m, n = S.shape
        med = np.empty(n, dtype=np.float64)
        iz = -1 if ignore_zero is None else int(bool(ignore_zero))
        check(self.lib.plaidhip_normalize_medians(self.handle, _np_ptr(S), m, n, iz, _np_ptr(med)))
        return S, med

    def colranks_dense(self, X, ties="average", signed=False) -> np.ndarray:
        X = _as_f64_fortran(X)
        g, n = X.shape
        R = np.empty((g, n), dtype=np.float64, order="F")
        check(self.lib.plaidhip_colranks_dense(self.handle, _np_ptr(X), g, n, TIES[ties], int(bool(signed)),
                                               _np_ptr(R)))
        return R

    def colranks_csc(self, Xp, Xx, ties="average", signed=False) -> np.ndarray:
        Xp = _as_i32(Xp)
        Xx = np.ascontiguousarray(Xx, dtype=np.float64)
        R = np.empty(len(Xx), dtype=np.float64)
        check(self.lib.plaidhip_colranks_csc(self.handle, _np_ptr(Xp), _np_ptr(Xx), len(Xp) - 1, TIES[ties],
                                             int(bool(signed)), _np_ptr(R)))
        return R

    def colranks_csc_dense(self, Xp, Xi, Xx, g: int, ties="average", signed=False) -> np.ndarray:
        """colranks(sparse X, keep.zero=FALSE): zeros ranked, dense g x n result (R/plaid.R:602-609)"""
        Xp, Xi = _as_i32(Xp), _as_i32(Xi)
        Xx = np.ascontiguousarray(Xx, dtype=np.float64)
        n = len(Xp) - 1
        R = np.empty((int(g), n), dtype=np.float64, order="F")
        check(self.lib.plaidhip_colranks_csc_dense(self.handle, _np_ptr(Xp), _np_ptr(Xi), _np_ptr(Xx), int(g), n,
                                                   TIES[ties], int(bool(signed)), _np_ptr(R)))
        return R

    def sing_dense(self, X, Gp, Gi) -> np.ndarray:
        X = _as_f64_fortran(X)
        g, n = X.shape
        Gp, Gi = _as_i32(Gp), _as_i32(Gi)
        m = len(Gp) - 1
        S = np.empty((m, n), dtype=np.float64, order="F")
        check(self.lib.plaidhip_sing_dense(self.handle, _np_ptr(X), g, n, _np_ptr(Gp), _np_ptr(Gi), m, _np_ptr(S)))
        return S

    def ssgsea_dense(self, X, Gp, Gi, alpha=0.0) -> np.ndarray:
        X = _as_f64_fortran(X)
        g, n = X.shape
        Gp, Gi = _as_i32(Gp), _as_i32(Gi)
        m = len(Gp) - 1
        S = np.empty((m, n), dtype=np.float64, order="F")
        check(self.lib.plaidhip_ssgsea_dense(self.handle, _np_ptr(X), g, n, _np_ptr(Gp), _np_ptr(Gi), m,
                                             float(alpha), _np_ptr(S)))
        return S

    def ssgsea_csc(self, Xp, Xi, Xx, g: int, Gp, Gi, alpha=0.0) -> np.ndarray:
        Xp, Xi = _as_i32(Xp), _as_i32(Xi)
        Xx = np.ascontiguousarray(Xx, dtype=np.float64)
        n = len(Xp) - 1
        Gp, Gi = _as_i32(Gp), _as_i32(Gi)
        m = len(Gp) - 1
        S = np.empty((m, n), dtype=np.float64, order="F")
        check(self.lib.plaidhip_ssgsea_csc(self.handle, _np_ptr(Xp), _np_ptr(Xi), _np_ptr(Xx), int(g), n,
                                           _np_ptr(Gp), _np_ptr(Gi), m, float(alpha), _np_ptr(S)))
        return S


def _x_args(X):
    """(Xp, Xi, values, g, n, keepalive) for a dense ndarray or a scipy CSC matrix."""
    import scipy.sparse as sp
    if sp.issparse(X):
        X = sp.csc_matrix(X)
        Xp, Xi = _as_i32(X.indptr), _as_i32(X.indices)
        Xx = np.ascontiguousarray(X.data, dtype=np.float64)
        return _np_ptr(Xp), _np_ptr(Xi), _np_ptr(Xx), X.shape[0], X.shape[1], (Xp, Xi, Xx)
    Xd = _as_f64_fortran(X)
    return None, None, _np_ptr(Xd), Xd.shape[0], Xd.shape[1], (Xd,)


def _ucell(self, X, Gp, Gi, k_full, rmax=1500.0):
    xp, xi, xv, g, n, keep = _x_args(X)
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    kf = np.ascontiguousarray(k_full, dtype=np.float64)
    S = np.empty((m, n), dtype=np.float64, order="F")
    check(self.lib.plaidhip_ucell(self.handle, xp, xi, xv, g, n, _np_ptr(Gp), _np_ptr(Gi), m, _np_ptr(kf),
                                  float(rmax), _np_ptr(S)))
    return S


def _aucell(self, X, Gp, Gi, auc_max_rank):
    xp, xi, xv, g, n, keep = _x_args(X)
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    check(self.lib.plaidhip_aucell(self.handle, xp, xi, xv, g, n, _np_ptr(Gp), _np_ptr(Gi), m,
                                   float(auc_max_rank), _np_ptr(S)))
    return S


def _scse(self, X, Gp, Gi, remove_log2=None, score_mean=False):
    xp, xi, xv, g, n, keep = _x_args(X)
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    rl = -1 if remove_log2 is None else int(bool(remove_log2))
    removed = C.c_int(0)
    check(self.lib.plaidhip_scse(self.handle, xp, xi, xv, g, n, _np_ptr(Gp), _np_ptr(Gi), m, rl,
                                 int(bool(score_mean)), _np_ptr(S), C.byref(removed)))
    self.last_scse_removed_log2 = bool(removed.value)   # the automatic decision is taken on the device (R/plaid.R:160-161)
    return S


def _plaid_test(self, X, y, Gp, Gi, gsetX=None, tests=7, metap_method=0):
    """plaidhip_plaid_test: returns sets x 6 (gsetFC, p.one, p.two, p.lm, p.meta, q.meta), G's column order"""
    X = _as_f64_fortran(X)
    g, n = X.shape
    y = np.ascontiguousarray(y, dtype=np.int32)
    if y.shape != (n,):
        raise ValueError("y must have one entry per column of X")
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    sx = None
    if gsetX is not None:
        sx = _as_f64_fortran(gsetX)
        if sx.shape != (m, n):
            raise ValueError("gsetX must be sets x samples")
    out = np.empty((m, 6), dtype=np.float64, order="F")
    check(self.lib.plaidhip_plaid_test(self.handle, _np_ptr(X), g, n, _np_ptr(y), _np_ptr(Gp), _np_ptr(Gi), m,
                                       _np_ptr(sx) if sx is not None else None, int(tests), int(metap_method),
                                       _np_ptr(out)))
    return out


def plaid_test_finish(g, Gp, T, tot1, tot2, SM, n0, n1, tests=7, metap_method=0, lib=None):
    """plaidhip_plaid_test_finish (host only): the p-values, effect sizes, meta-p and FDR of plaid.test from the reduced
    statistics -- T (2, m) per-set sums of fc and fc^2, tot1 / tot2 their sums over all genes, SM (4, m) group means and
    sums of squared deviations of the score rows (None without "lm").  Returns sets x 6 like Context.plaid_test."""
    from ._lib import load
    lib = lib or load()
    Gp = _as_i32(Gp)
    m = len(Gp) - 1
    T = np.ascontiguousarray(T, dtype=np.float64)
    if T.shape != (2, m):
        raise ValueError("T must be (2, sets)")
    if SM is not None:
        SM = np.ascontiguousarray(SM, dtype=np.float64)
        if SM.shape != (4, m):
            raise ValueError("SM must be (4, sets)")
    out = np.empty((m, 6), dtype=np.float64, order="F")
    check(lib.plaidhip_plaid_test_finish(int(g), m, _np_ptr(Gp), _np_ptr(T), float(tot1), float(tot2),
                                         _np_ptr(SM) if SM is not None else None, int(n0), int(n1), int(tests),
                                         int(metap_method), _np_ptr(out)))
    return out


def _gsva(self, X, Gp, Gi, tau=0.0, rowtf="z"):
    if rowtf not in ("z", "ecdf"):
        raise ValueError("Error: unknown row transform" + str(rowtf))          # R/plaid.R:348
    X = _as_f64_fortran(X)
    g, n = X.shape
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    check(self.lib.plaidhip_gsva(self.handle, _np_ptr(X), g, n, _np_ptr(Gp), _np_ptr(Gi), m, float(tau),
                                 0 if rowtf == "z" else 1, _np_ptr(S)))
    return S


Context.gsva = _gsva
Context.plaid_test = _plaid_test
Context.ucell = _ucell
Context.aucell = _aucell
Context.scse = _scse

_default_ctx: Context | None = None


def default_context() -> Context:
    """Process-wide context on device LOCAL_RANK (or 0), private stream."""
    global _default_ctx
    if _default_ctx is None:
        import os
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx


# ---- several GPUs from one host process (plaidhip_*_multi: a host thread per device, no RCCL) ------------------
def shard_bounds(n: int, ndev: int, k: int):
    """columns [lo, hi) of shard k of n sample columns over ndev devices (needs no device)"""
    lo, hi = C.c_int64(0), C.c_int64(0)
    check(_lib.load().plaidhip_shard_bounds(int(n), int(ndev), int(k), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def _devices_arg(devices):
    if devices is None:
        raise ValueError("devices: a list of device ordinals or an int (the first ndev devices)")
    if isinstance(devices, int):
        return None, int(devices), None
    d = np.ascontiguousarray(devices, dtype=np.int32)
    return _np_ptr(d), len(d), d


def plaid_multi(X, Gp, Gi, stat="mean", normalize=True, devices=1) -> np.ndarray:
    """plaid() with the sample columns sharded over `devices` (an int: devices 0 .. n-1, or a list of ordinals)"""
    lib = _lib.load()
    xp, xi, xv, g, n, keep = _x_args(X)
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    dp, nd, dkeep = _devices_arg(devices)
    check(lib.plaidhip_plaid_multi(dp, nd, xp, xi, xv, g, n, _np_ptr(Gp), _np_ptr(Gi), m, STAT[stat], int(bool(normalize)),
                                   _np_ptr(S)))
    return S


def sing_multi(X, Gp, Gi, devices=1) -> np.ndarray:
    lib = _lib.load()
    X = _as_f64_fortran(X)
    g, n = X.shape
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    dp, nd, dkeep = _devices_arg(devices)
    check(lib.plaidhip_sing_multi(dp, nd, _np_ptr(X), g, n, _np_ptr(Gp), _np_ptr(Gi), m, _np_ptr(S)))
    return S


def ssgsea_multi(X, Gp, Gi, alpha=0.0, devices=1) -> np.ndarray:
    lib = _lib.load()
    xp, xi, xv, g, n, keep = _x_args(X)
    Gp, Gi = _as_i32(Gp), _as_i32(Gi)
    m = len(Gp) - 1
    S = np.empty((m, n), dtype=np.float64, order="F")
    dp, nd, dkeep = _devices_arg(devices)
    check(lib.plaidhip_ssgsea_multi(dp, nd, xp, xi, xv, g, n, _np_ptr(Gp), _np_ptr(Gi), m, float(alpha), _np_ptr(S)))
    return S


def multi_finalize():
    check(_lib.load().plaidhip_multi_finalize())
