"""ctypes binding of the C ABI in include/plaidhip.h (plaid_amd/csrc/libplaidhip.so).

There is no CPU fallback: if the shared library is missing or no gfx950 device is visible,
every compute entry point raises `PlaidHipError`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLAIDHIP_LIB") or os.path.join(_HERE, "csrc", "libplaidhip.so")   # override: A/B builds in tools/

OK, EINVAL, ENOMEM, EHIP, EUNSUPPORTED, ENODEVICE = range(6)
STAT = {"mean": 0, "sum": 1}
TIES = {"average": 0, "min": 1, "max": 2, "first": 3, "last": 4, "dense": 5, "random": 6}
FLAG_HAS_NEG, FLAG_HAS_ZERO, FLAG_HAS_NAN = 1, 2, 4
# enum plaidhip_option and its values (include/plaidhip.h)
OPTIONS = {
    "spmm_dense_kernel": (1, {"auto": 0, "single": 1, "pair": 2, "mfma": 3}),
    "spmm_sparse_kernel": (2, {"auto": 0, "scatter": 1, "gather": 2}),
    "nt_store": (3, {"auto": -1, "off": 0, "on": 1}),
    "ranks_f32": (4, {"off": 0, "on": 1, "f64": 0, "f32": 1, "u16": 2}),   # staging of rank inputs (default 2 = u16)
    "rank_kernel": (5, {"auto": 0, "network": 1, "bucket": 2, "bucket512": 3}),
    "scatter_fixed": (6, {"off": 0, "on": 1}),
    "scatter_order": (7, {"column": 0, "chunk": 1}),
    "fused_medians": (8, {"auto": 0, "on": 1, "off": 2}),
}


class PlaidHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"plaidhip error {code}: {message}")
        self.code = code


_vp, _i32, _i64, _f64, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_int

# name -> argtypes (restype is int unless noted); mirrors include/plaidhip.h one to one
SIGNATURES = {
    "plaidhip_version": [],
    "plaidhip_last_error_string": [],
    "plaidhip_device_count": [C.POINTER(_int)],
    "plaidhip_init": [_int, _vp, C.POINTER(_vp)],
    "plaidhip_finalize": [_vp],
    "plaidhip_synchronize": [_vp],
    "plaidhip_set_precision": [_vp, _int],
    "plaidhip_set_stream": [_vp, _vp],
    "plaidhip_set_option": [_vp, _int, _int],
    "plaidhip_malloc": [_vp, C.c_size_t, C.POINTER(_vp)],
    "plaidhip_free": [_vp, _vp],
    "plaidhip_memcpy_h2d": [_vp, _vp, _vp, C.c_size_t],
    "plaidhip_memcpy_d2h": [_vp, _vp, _vp, C.c_size_t],
    "plaidhip_geneset_create": [_vp, _i32, _i32, _vp, _vp, C.POINTER(_vp)],
    "plaidhip_geneset_destroy": [_vp],
    "plaidhip_geneset_info": [_vp, C.POINTER(_i64)],
    "plaidhip_dev_spmm_dense_f64": [_vp, _vp, _vp, _i64, _i32, _int, _f64, _vp, _f64, _vp, _i64, _vp],
    "plaidhip_dev_spmm_dense_fused_f64": [_vp, _vp, _vp, _i64, _i32, _int, _f64, _vp, _f64, _vp, _i64, _vp],
    "plaidhip_dev_spmm_ranks_f64": [_vp, _vp, _vp, _i64, _i32, _int, _f64, _vp, _f64, _vp, _i64, _vp],
    "plaidhip_dev_spmm_csc_f64": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _int, _f64, _vp, _f64, _vp, _i64, _vp],
    "plaidhip_dev_spmm_csc_ranks_f64": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _int, _f64, _vp, _f64, _vp, _i64, _vp],
    "plaidhip_dev_crossprod_weighted_f64": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _i32, _vp, _i64],
    "plaidhip_dev_crossprod_weighted_csc_f64": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i64],
    "plaidhip_dev_colranks_dense_f64": [_vp, _vp, _i64, _i32, _i32, _int, _int, _f64, _vp, _i64, _vp],
    "plaidhip_dev_colranks_csc_f64": [_vp, _vp, _vp, _i32, _i32, _int, _int, _f64, _vp, _vp],
    "plaidhip_dev_colranks_csc_dense_f64": [_vp, _vp, _vp, _vp, _i32, _i32, _int, _int, _f64, _vp, _i64, _vp],
    "plaidhip_dev_colranks_csc_dense_nz_f64": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _int, _int, _f64, _vp, _vp, _i64, _vp],
    "plaidhip_dev_minflags": [_vp, _vp, _i64, _vp],
    "plaidhip_dev_col_medians": [_vp, _vp, _i64, _i32, _i32, _int, _vp, _vp],
    "plaidhip_dev_spmm_csc_fused_f64": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _int, _f64, _vp, _f64, _vp, _i64, _vp, _vp],
    "plaidhip_dev_col_medians_resume": [_vp, _vp, _i64, _i32, _i32, _int, _vp, _vp],
    "plaidhip_dev_col_medians_resume_token": [_vp, _i64, _vp, _i64, _i32, _i32, _int, _vp, _vp],
    "plaidhip_dev_fused_medians_discard": [_vp],
    "plaidhip_dev_fused_medians_info": [_vp, C.POINTER(_i64)],
    "plaidhip_dev_sum": [_vp, _vp, _i64, _vp],
    "plaidhip_dev_shift_columns": [_vp, _vp, _i64, _i32, _i32, _vp, _f64, _vp],
    "plaidhip_dev_shift_columns_cast_f32": [_vp, _vp, _i64, _i32, _i32, _vp, _f64, _vp, _vp, _i64],
    "plaidhip_dev_max": [_vp, _vp, _i64, _vp],
    "plaidhip_plaid_dense": [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _int, _int, _vp],
    "plaidhip_plaid_csc": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _int, _int, _vp],
    "plaidhip_crossprod_weighted_dense": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp],
    "plaidhip_crossprod_weighted_csc": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp],
    "plaidhip_normalize_medians": [_vp, _vp, _i32, _i32, _int, _vp],
    "plaidhip_colranks_dense": [_vp, _vp, _i32, _i32, _int, _int, _vp],
    "plaidhip_colranks_csc": [_vp, _vp, _vp, _i32, _int, _int, _vp],
    "plaidhip_colranks_csc_dense": [_vp, _vp, _vp, _vp, _i32, _i32, _int, _int, _vp],
    "plaidhip_sing_csc": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp],
    "plaidhip_sing_dense": [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp],
    "plaidhip_ssgsea_dense": [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _f64, _vp],
    "plaidhip_ssgsea_csc": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _f64, _vp],
    "plaidhip_shard_bounds": [_i64, _int, _int, C.POINTER(_i64), C.POINTER(_i64)],
    "plaidhip_limit": [_int, C.POINTER(_i64)],
    "plaidhip_plaid_multi": [_vp, _int, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _int, _int, _vp],
    "plaidhip_sing_multi": [_vp, _int, _vp, _i32, _i32, _vp, _vp, _i32, _vp],
    "plaidhip_sing_csc_multi": [_vp, _int, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp],
    "plaidhip_ssgsea_multi": [_vp, _int, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _f64, _vp],
    "plaidhip_multi_finalize": [],
    "plaidhip_multi_set_precision": [_int],
    "plaidhip_ucell": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _f64, _vp],
    "plaidhip_aucell": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _f64, _vp],
    "plaidhip_scse": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _int, _int, _vp, C.POINTER(_int)],
    "plaidhip_gsva": [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _f64, _int, _vp],
    "plaidhip_plaid_test": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _int, _int, _vp],
    "plaidhip_dev_row_group_sums": [_vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "plaidhip_dev_row_group_ssd": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp],
    "plaidhip_plaid_test_finish": [_i32, _i32, _vp, _vp, _f64, _f64, _vp, _i64, _i64, _int, _int, _vp],
    # host-only GMT text path (gmt.cpp)
    "plaidhip_gmt_read": [C.c_char_p, _int, _i64, C.POINTER(_vp)],
    "plaidhip_gmt_parse": [C.c_char_p, _i64, _int, _int, _i64, C.POINTER(_vp)],
    "plaidhip_gmt_nsets": [_vp],
    "plaidhip_gmt_set_name": [_vp, _i64],
    "plaidhip_gmt_set_size": [_vp, _i64],
    "plaidhip_gmt_set_gene": [_vp, _i64, _i64],
    "plaidhip_gmt_text": [_vp, C.POINTER(_i64)],
    "plaidhip_gmt_destroy": [_vp],
    "plaidhip_gmt2mat": [_vp, _i64, _i64, C.POINTER(C.c_char_p), _i64, C.POINTER(_vp)],
    "plaidhip_gmtmat_dims": [_vp, C.POINTER(_i64)],
    "plaidhip_gmtmat_p": [_vp],
    "plaidhip_gmtmat_i": [_vp],
    "plaidhip_gmtmat_names": [_vp, _int, C.POINTER(_i64)],
    "plaidhip_gmtmat_destroy": [_vp],
}

# return types other than the int status code
RESTYPES = {
    "plaidhip_last_error_string": C.c_char_p,
    "plaidhip_gmt_nsets": C.c_int64,
    "plaidhip_gmt_set_name": C.c_char_p,
    "plaidhip_gmt_set_size": C.c_int64,
    "plaidhip_gmt_set_gene": C.c_char_p,
    "plaidhip_gmt_text": C.c_void_p,
    "plaidhip_gmtmat_p": C.c_void_p,
    "plaidhip_gmtmat_i": C.c_void_p,
    "plaidhip_gmtmat_names": C.c_void_p,
}

_lib = None


def load():
    """Load libplaidhip.so (once).  Raises PlaidHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PlaidHipError(
            ENODEVICE,
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C plaid_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    try:
        # share one HIP runtime with the host framework when it is present: torch ships its own
        # libamdhip64.so under the same SONAME, which the loader then reuses for this library.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the R-style host API
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, _int)
    _lib = lib
    return lib


def check(code: int):
    if code != OK:
        msg = load().plaidhip_last_error_string()
        raise PlaidHipError(code, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = _int(0)
    try:
        check(load().plaidhip_device_count(C.byref(n)))
    except PlaidHipError:
        return 0
    return n.value
