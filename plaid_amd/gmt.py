"""GMT gene-set utilities with the reference's names and rules (R/gmt-utils.R).

Host glue: produces the 0/1 genes x sets CSC matrix the device consumes.  (SURVEY.md 8f
ranks a fast native builder as a later row; this is the plain host implementation.)"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .matrix import NamedMatrix


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


def read_gmt(gmt_file, dir=None, add_source=False, nrows=-1) -> GmtList:  # noqa: A002
    """read.gmt(), R/gmt-utils.R:99-125: one set per line, '#' comments, tab separated,
    field 1 name, field 2 source, the rest genes; "" / "NA" / duplicates dropped (:117)."""
    path = gmt_file
    if dir is not None and not str(gmt_file).startswith("/"):
        path = str(dir).rstrip("/") + "/" + str(gmt_file)
    names, sets = [], []
    with open(path, "r", encoding="utf-8") as fh:
        for raw in fh:
            line = raw.rstrip("\r\n").split("#", 1)[0]
            if not line.strip():
                continue
            fields = line.split("\t")
            source = fields[1] if len(fields) > 1 else "NA"
            genes = " ".join(fields[2:]).replace("\t", " ").split(" ") if len(fields) >= 3 else []
            name = f"{fields[0]} ({source})" if add_source else fields[0]
            names.append(name)
            sets.append(_unique(x for x in genes if x not in ("", "NA")))
            if 0 < nrows <= len(names):
                break
    return GmtList(names, sets)


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
    order = sorted(range(len(gmt)), key=lambda k: -len(gmt.sets[k]))       # stable
    names = [gmt.names[k] for k in order]
    sets = [gmt.sets[k] for k in order]
    seen, keep = set(), []
    for k, nm in enumerate(names):
        if nm not in seen:
            seen.add(nm)
            keep.append(k)
    names = [names[k] for k in keep]
    sets = [sets[k] for k in keep]
    if ntop > 0:
        sets = [s[:ntop] for s in sets]
    if not names:
        names = []
    if bg is None:
        cnt: dict = {}
        for s in sets:
            for x in s:
                cnt[x] = cnt.get(x, 0) + 1
        bg = sorted(sorted(cnt), key=lambda x: -cnt[x])
    bg = list(bg)
    if max_genes < 0:
        max_genes = len(bg)
    gg = bg[:max_genes]
    pos = {x: k for k, x in enumerate(gg)}
    rows, cols = [], []
    for j, s in enumerate(sets):
        for x in _unique(s):
            r = pos.get(x)
            if r is not None:
                rows.append(r)
                cols.append(j)
    D = sp.csc_matrix((np.ones(len(rows)), (np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64))),
                      shape=(len(gg), len(names)), dtype=np.float64)
    D.sum_duplicates()
    D.data[:] = 1.0
    rs = np.asarray((D != 0).sum(axis=1)).ravel()
    ro = np.argsort(-rs, kind="stable")
    D = D[ro, :].tocsc()
    D.sort_indices()
    rn = [gg[k] for k in ro]
    if not sparse:
        return NamedMatrix(D.toarray(), rn, names)
    return NamedMatrix(D, rn, names)


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
