"""A matrix with dimnames: the Python stand-in for an R `matrix` / `dgCMatrix` with
`rownames()` / `colnames()` that the reference's functions take and return."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


class NamedMatrix:
    """`values`: Fortran-ordered float64 ndarray (R `matrix`) or scipy CSC (R `dgCMatrix`)."""

    def __init__(self, values, rownames=None, colnames=None):
        if sp.issparse(values):
            values = sp.csc_matrix(values)
            values.sort_indices()
        else:
            values = np.asarray(values, dtype=np.float64)
            if values.ndim == 1:
                values = values.reshape(-1, 1)       # R/plaid.R:63  cbind(X)
            values = np.asfortranarray(values)
        self.values = values
        nr, nc = values.shape
        self.rownames = [str(r) for r in rownames] if rownames is not None else [f"row{k + 1}" for k in range(nr)]
        self.colnames = [str(c) for c in colnames] if colnames is not None else [f"col{k + 1}" for k in range(nc)]
        if len(self.rownames) != nr or len(self.colnames) != nc:
            raise ValueError("dimnames do not match matrix shape")

    @property
    def shape(self):
        return self.values.shape

    @property
    def is_sparse(self):
        return sp.issparse(self.values)

    def dense(self) -> np.ndarray:
        return np.asfortranarray(self.values.toarray()) if self.is_sparse else self.values

    def to_pandas(self):
        import pandas as pd
        return pd.DataFrame(self.dense(), index=self.rownames, columns=self.colnames)

    def __repr__(self):
        kind = "dgCMatrix" if self.is_sparse else "matrix"
        return f"<NamedMatrix {self.shape[0]} x {self.shape[1]} {kind}>"


def as_named(x) -> NamedMatrix:
    if isinstance(x, NamedMatrix):
        return x
    try:
        import pandas as pd
        if isinstance(x, pd.DataFrame):
            return NamedMatrix(x.to_numpy(dtype=np.float64), list(x.index), list(x.columns))
    except ImportError:  # pragma: no cover
        pass
    return NamedMatrix(x)
