"""Minimal reader for R `save()` files (gzip'd RDX3 / XDR serialisation).

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Used by
`oracle/make_golden.py` to turn the reference's bundled fixture
`inst/extdata/pbmc3k-50cells.rda` (made by `dev/extdata.R:1-15`) into plain
numpy arrays under `tests/golden/`.  R is not installed in the build
container, so the format is decoded directly.  Only the SEXP types that occur
in a `dgCMatrix` + character vector are supported; anything else raises.

Format notes (R internals manual, "Serialization Formats"):
  header  "RDX3\n" "X\n"  int version, int writer_version, int min_version,
          [v3] int nelen + native-encoding bytes
  item    int flags: type = flags & 0xff, is_obj = bit 8, has_attr = bit 9,
          has_tag = bit 10, gp = bits 12..27
"""
from __future__ import annotations

import gzip
import struct

import numpy as np

NILVALUE_SXP = 254
REFSXP = 255
ALTREP_SXP = 238
SYMSXP, LISTSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, S4SXP = (
    1, 2, 9, 10, 13, 14, 16, 19, 25)
LANGSXP = 6
NA_INT = -2147483648


class RObject:
    """An R object with attributes (S4 instances keep slots in `attr`)."""

    def __init__(self, value=None, attr=None, kind=""):
        self.value = value
        self.attr = attr or {}
        self.kind = kind

    def __repr__(self):
        return f"RObject(kind={self.kind!r}, attr={list(self.attr)})"


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        self.o = 0
        self.refs: list = []

    def _int(self) -> int:
        v = struct.unpack_from(">i", self.b, self.o)[0]
        self.o += 4
        return v

    def _bytes(self, n: int) -> bytes:
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def _len(self) -> int:
        n = self._int()
        if n == -1:  # long vector: two ints (hi, lo)
            hi, lo = self._int(), self._int()
            n = (hi << 32) + lo
        return n

    def header(self):
        if self._bytes(5) != b"RDX3\n":
            raise ValueError("not an RDX3 file")
        if self._bytes(2) != b"X\n":
            raise ValueError("only XDR serialisation supported")
        version = self._int()
        self._int()
        self._int()
        if version == 3:
            self._bytes(self._int())
        elif version != 2:
            raise ValueError(f"unsupported serialisation version {version}")

    def _attrs(self) -> dict:
        out = {}
        pl = self.item()
        for tag, val in pl or []:
            out[tag] = val
        return out

    def item(self):
        flags = self._int()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)
        if t == NILVALUE_SXP:
            return None
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self._int()
            return self.refs[idx - 1]
        if t == SYMSXP:
            name = self.item()
            self.refs.append(name)
            return name
        if t in (LISTSXP, LANGSXP):
            # pairlist -> python list of (tag, value); iterative on cdr
            out = []
            while True:
                attr = self._attrs() if has_attr else None  # noqa: F841
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                flags = self._int()
                t2 = flags & 0xFF
                if t2 == NILVALUE_SXP:
                    return out
                if t2 not in (LISTSXP, LANGSXP):
                    raise ValueError("dotted pairlist not supported")
                has_attr = bool(flags & 0x200)
                has_tag = bool(flags & 0x400)
        if t == CHARSXP:
            n = self._int()
            if n == -1:
                return None
            return self._bytes(n).decode("utf-8", "replace")
        if t in (LGLSXP, INTSXP):
            n = self._len()
            v = np.frombuffer(self._bytes(4 * n), dtype=">i4").astype(np.int32)
            return self._wrap(v, has_attr, "int" if t == INTSXP else "lgl")
        if t == REALSXP:
            n = self._len()
            v = np.frombuffer(self._bytes(8 * n), dtype=">f8").astype(np.float64)
            return self._wrap(v, has_attr, "real")
        if t == STRSXP:
            n = self._len()
            v = [self.item() for _ in range(n)]
            return self._wrap(v, has_attr, "str")
        if t == VECSXP:
            n = self._len()
            v = [self.item() for _ in range(n)]
            return self._wrap(v, has_attr, "list")
        if t == S4SXP:
            return RObject(None, self._attrs() if has_attr else {}, "S4")
        if t == ALTREP_SXP:
            info = self.item()
            state = self.item()
            attr = self.item()
            cls = info[0][1]
            if cls == "compact_intseq":
                n, start, inc = (int(x) for x in _val(state))
                v = (start + inc * np.arange(n)).astype(np.int32)
            elif cls == "compact_realseq":
                n, start, inc = _val(state)
                v = start + inc * np.arange(int(n), dtype=np.float64)
            elif cls.startswith("wrap_"):
                v = _val(_val(state)[0])
            elif cls == "deferred_string":
                src = _val(state if not isinstance(state, list) else state[0][1])
                v = [_fmt(x) for x in src]
            else:
                raise ValueError(f"unsupported ALTREP class {cls}")
            if attr:
                return RObject(v, dict(attr), "altrep")
            return v
        raise ValueError(f"unsupported SEXP type {t} at offset {self.o}")

    def _wrap(self, v, has_attr, kind):
        if has_attr:
            return RObject(v, self._attrs(), kind)
        return v


def _val(x):
    return x.value if isinstance(x, RObject) else x


def _fmt(x):
    return str(int(x)) if float(x).is_integer() else repr(float(x))


def read_rda(path: str) -> dict:
    """Return {variable name: decoded object} for an R `save()` file."""
    with gzip.open(path, "rb") as fh:
        r = _Reader(fh.read())
    r.header()
    top = r.item()
    return {tag: val for tag, val in top}


def dgcmatrix_to_csc(obj: RObject):
    """Slots of a Matrix::dgCMatrix -> (p int32, i int32, x float64, dim, rownames, colnames)."""
    a = obj.attr
    p = np.asarray(_val(a["p"]), dtype=np.int32)
    i = np.asarray(_val(a["i"]), dtype=np.int32)
    x = np.asarray(_val(a["x"]), dtype=np.float64)
    dim = tuple(int(d) for d in _val(a["Dim"]))
    dn = _val(a["Dimnames"])
    rn = _val(dn[0]) if dn[0] is not None else None
    cn = _val(dn[1]) if dn[1] is not None else None
    return p, i, x, dim, rn, cn
