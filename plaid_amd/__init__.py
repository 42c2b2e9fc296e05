"""plaid_amd -- MI355X-native single-sample gene-set scoring behind the bigomics/plaid API.

Host side of the drop-in: the reference's R function names (`plaid`, `colranks`,
`normalize_medians`, `replaid.sing` ...) with dots as underscores, calling hand-written
gfx950 kernels through the C ABI of include/plaidhip.h.  See DESIGN.md / INTEGRATION.md.
"""
from ._lib import PlaidHipError, device_count
from .api import (aligned_pattern, chunked_crossprod, colranks, normalize_medians, plaid, plaid_test, replaid_gsva,
                  replaid_aucell, replaid_scse, replaid_sing, replaid_ssgsea, replaid_ucell,
                  sparse_colranks)
from .engine import (Context, Geneset, default_context, multi_finalize, plaid_multi, shard_bounds, sing_multi,
                     ssgsea_multi)
from .gmt import GmtList, gmt2mat, mat2gmt, read_gmt, write_gmt
from .matrix import NamedMatrix, as_named

__all__ = [
    "PlaidHipError", "device_count", "Context", "Geneset", "default_context", "NamedMatrix",
    "as_named", "GmtList", "read_gmt", "write_gmt", "gmt2mat", "mat2gmt", "plaid",
    "chunked_crossprod", "normalize_medians", "colranks", "sparse_colranks", "replaid_sing",
    "replaid_ssgsea", "replaid_ucell", "replaid_aucell", "replaid_scse", "aligned_pattern", "plaid_test", "replaid_gsva",
    "plaid_multi", "sing_multi", "ssgsea_multi", "shard_bounds", "multi_finalize",
]
__version__ = "0.2.0"
