"""GMT gene-set utilities with the reference's names and rules (R/gmt-utils.R).

Host glue: produces the 0/1 genes x sets CSC matrix the device consumes.  `read_gmt` and
`gmt2mat` run in the native library (plaid_amd/csrc/gmt.cpp behind the C ABI; SURVEY.md 8f-2:
the R versions take 50 s for a 50k-set collection); `gmt2mat_file` goes from the path to the
matrix without materialising the gene lists in Python."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from .matrix import NamedMatrix


def _check(rc):
    if rc != 0:
        lib = _lib.load()
        raise _lib.PlaidHipError(rc, (lib.plaidhip_last_error_string() or b"").decode())


def _bytes_at(ptr, n):
    return C.string_at(ptr, n) if ptr and n else b""


class _NativeGmt:
    """Owns a plaidhip_gmt handle."""

    def __init__(self, handle):
        self.lib = _lib.load()
        self.handle = handle

    @classmethod
    def from_file(cls, path, add_source=False, nrows=-1):
        lib = _lib.load()
        h = C.c_void_p()
        _check(lib.plaidhip_gmt_read(str(path).encode(), int(bool(add_source)), int(nrows), C.byref(h)))
        return cls(h)

    @classmethod
    def from_list(cls, gmt):
        """exchange format: one set per line, name TAB source TAB gene TAB gene ..."""
        lib = _lib.load()
        for nm, s in zip(gmt.names, gmt.sets):
            if any(("\t" in x) or ("\n" in x) for x in s) or "\t" in nm or "\n" in nm:
                raise ValueError("gene-set and gene names must not contain tabs or newlines")
        text = "\n".join(nm + "\t\t" + "\t".join(s) for nm, s in zip(gmt.names, gmt.sets)).encode()
        h = C.c_void_p()
        _check(lib.plaidhip_gmt_parse(text, len(text), 1, 0, -1, C.byref(h)))
        return cls(h)

    def to_list(self) -> "GmtList":
        nb = C.c_int64()
        ptr = self.lib.plaidhip_gmt_text(self.handle, C.byref(nb))
        text = _bytes_at(ptr, nb.value).decode()
        names, sets = [], []
        if self.lib.plaidhip_gmt_nsets(self.handle):
            for line in text.split("\n"):
                f = line.split("\t")
                names.append(f[0])
                sets.append(f[1:])
        return GmtList(names, sets)

    def to_matrix(self, max_genes=-1, ntop=-1, bg=None, sparse=True) -> NamedMatrix:
        nbg = 0 if bg is None else len(bg)
        arr = None
        if nbg:
            arr = (C.c_char_p * nbg)(*[str(x).encode() for x in bg])
        m = C.c_void_p()
        _check(self.lib.plaidhip_gmt2mat(self.handle, int(max_genes), int(ntop), arr, nbg, C.byref(m)))
        try:
            dims = (C.c_int64 * 3)()
            _check(self.lib.plaidhip_gmtmat_dims(m, dims))
            g, ns, z = dims[0], dims[1], dims[2]
            p = np.ctypeslib.as_array(C.cast(self.lib.plaidhip_gmtmat_p(m), C.POINTER(C.c_int32)), shape=(ns + 1,)).copy()
            i = (np.ctypeslib.as_array(C.cast(self.lib.plaidhip_gmtmat_i(m), C.POINTER(C.c_int32)), shape=(z,)).copy()
                 if z else np.zeros(0, dtype=np.int32))
            nb = C.c_int64()
            rn = _bytes_at(self.lib.plaidhip_gmtmat_names(m, 0, C.byref(nb)), nb.value).decode()
            rn = rn.split("\n") if g else []
            cn = _bytes_at(self.lib.plaidhip_gmtmat_names(m, 1, C.byref(nb)), nb.value).decode()
            cn = cn.split("\n") if ns else []
        finally:
            self.lib.plaidhip_gmtmat_destroy(m)
        D = sp.csc_matrix((np.ones(z, dtype=np.float64), i, p), shape=(g, ns))
        return NamedMatrix(D if sparse else D.toarray(), rn, cn)

    def close(self):
        if self.handle:
            self.lib.plaidhip_gmt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GmtList:
    """An R named list of character vectors; names may repeat, so not a dict."""

    def __init__(self, names, sets):
        self.names = list(names)
        self.sets = [list(s) for s in sets]
        if len(self.names) != len(self.sets):
            raise ValueError("names and sets differ in length")

    @classmethod
    def coerce(cls, gmt):
        if isinstance(gmt, cls):
            return gmt
        if isinstance(gmt, dict):
            return cls(list(gmt.keys()), list(gmt.values()))
        names, sets = gmt
        return cls(names, sets)

    def __len__(self):
        return len(self.names)

    def __getitem__(self, name):
        return self.sets[self.names.index(name)]


def _unique(seq):
    seen, out = set(), []
    for s in seq:
        if s not in seen:
            seen.add(s)
            out.append(s)
    return out


def _gmt_path(gmt_file, dir):  # noqa: A002
    path = gmt_file
    if dir is not None and not str(gmt_file).startswith("/"):   # R/gmt-utils.R:104-105
        path = str(dir).rstrip("/") + "/" + str(gmt_file)
    return path


def read_gmt(gmt_file, dir=None, add_source=False, nrows=-1) -> GmtList:  # noqa: A002
    """read.gmt(), R/gmt-utils.R:99-125: one set per line, '#' comments, tab separated,
    field 1 name, field 2 source, the rest genes; "" / "NA" / duplicates dropped (:117)."""
    nat = _NativeGmt.from_file(_gmt_path(gmt_file, dir), add_source, nrows)
    try:
        return nat.to_list()
    finally:
        nat.close()


def gmt2mat_file(gmt_file, dir=None, add_source=False, nrows=-1, max_genes=-1, ntop=-1, sparse=True,  # noqa: A002
                 bg=None) -> NamedMatrix:
    """gmt2mat(read.gmt(file)) without building the gene lists in the host language."""
    nat = _NativeGmt.from_file(_gmt_path(gmt_file, dir), add_source, nrows)
    try:
        return nat.to_matrix(max_genes, ntop, bg, sparse)
    finally:
        nat.close()


def write_gmt(gmt, file, source=None):
    """write.gmt(), R/gmt-utils.R:139-144."""
    gmt = GmtList.coerce(gmt)
    src = gmt.names if source is None else ([source] * len(gmt) if isinstance(source, str) else list(source))
    with open(file, "w", encoding="utf-8") as fh:
        for nm, s, genes in zip(gmt.names, src, gmt.sets):
            fh.write(nm + "\t" + str(s) + "\t" + "\t".join(genes) + "\n")


def gmt2mat(gmt, max_genes=-1, ntop=-1, sparse=True, bg=None, use_multicore=True) -> NamedMatrix:
    """gmt2mat(), R/gmt-utils.R:19-66 -> genes x sets 0/1 matrix.  Sets by decreasing size
    (:25), duplicated names dropped (:26), head(ntop) (:27), rows = background genes by
    decreasing frequency (:31), head(max.genes) (:35), final row order by decreasing row sum
    (:62).  `use_multicore` is accepted for signature parity (in the reference both branches
    are single-threaded, :47-60)."""
    gmt = GmtList.coerce(gmt)
    nat = _NativeGmt.from_list(gmt)
    try:
        return nat.to_matrix(max_genes, ntop, bg, sparse)
    finally:
        nat.close()


def mat2gmt(mat) -> GmtList:
    """mat2gmt(), R/gmt-utils.R:80-85: non-zero rows of every column."""
    M = sp.csc_matrix(mat.values)
    sets = []
    for j in range(M.shape[1]):
        sl = slice(M.indptr[j], M.indptr[j + 1])
        rows = M.indices[sl][M.data[sl] != 0]
        sets.append([mat.rownames[r] for r in rows])
    keep = [j for j, s in enumerate(sets) if s]                 # tapply drops empty groups
    return GmtList([mat.colnames[j] for j in keep], [sets[j] for j in keep])
