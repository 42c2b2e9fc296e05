"""CPU-side checks: the C-ABI library loads and exports every symbol include/plaidhip.h
declares (no compute without a GPU), GMT / alignment host glue, synthetic generators."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import plaid_amd
from plaid_amd import _lib, synth
from plaid_amd.sharded import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "plaidhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(plaidhip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/plaidhip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.plaidhip_version() == 100


def test_no_cpu_fallback():
    """Without a gfx950 device the product path must fail loudly, not fall back."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(plaid_amd.PlaidHipError) as ei:
        plaid_amd.Context(0)
    assert "no CPU path" in str(ei.value)
    X = plaid_amd.NamedMatrix(np.ones((3, 2)), ["a", "b", "c"], ["s1", "s2"])
    G = plaid_amd.NamedMatrix(sp.csc_matrix(np.ones((3, 1))), ["a", "b", "c"], ["set"])
    with pytest.raises(plaid_amd.PlaidHipError):
        plaid_amd.plaid(X, G)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "plaid_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("# oracle", ""), f"{f} mentions the oracle"


def test_read_gmt_and_gmt2mat_match_oracle(golden_dir):
    from oracle import plaid_oracle as po
    path = os.path.join(golden_dir, "hallmarks.gmt")
    gmt = plaid_amd.read_gmt(path)
    names, sets = po.read_gmt(path)
    assert gmt.names == names and gmt.sets == sets and len(gmt) == 50
    M = plaid_amd.gmt2mat(gmt)
    D, rn, cn = po.gmt2mat(names, sets)
    assert M.shape == (4386, 50) and M.rownames == rn and M.colnames == cn
    assert (M.values != D).nnz == 0
    sizes = np.diff(M.values.indptr)
    assert np.all(np.diff(sizes) <= 0)                       # sets by decreasing size (R/gmt-utils.R:25)
    rs = np.asarray(M.values.sum(axis=1)).ravel()
    assert np.all(np.diff(rs) <= 0)                          # genes by decreasing frequency (:62)


def test_read_gmt_rules(tmp_path):
    p = tmp_path / "t.gmt"
    p.write_text("# comment\nS1\tsrc\tA\tB\tA\tNA\t\tC\nS2\tsrc\nS1\tdup\tD\tE\tF\tG\n")
    gmt = plaid_amd.read_gmt(str(p))
    assert gmt.names == ["S1", "S2", "S1"]
    assert gmt.sets[0] == ["A", "B", "C"] and gmt.sets[1] == []       # "", NA, duplicates dropped (:117)
    M = plaid_amd.gmt2mat(gmt)
    # sorted by size first, THEN duplicated names dropped -> the larger S1 survives (:25-26)
    assert M.colnames == ["S1", "S2"] and sorted(M.rownames) == ["D", "E", "F", "G"]
    assert plaid_amd.read_gmt(str(p), add_source=True).names[0] == "S1 (src)"
    q = tmp_path / "o.gmt"
    plaid_amd.write_gmt(gmt, str(q))
    assert plaid_amd.read_gmt(str(q)).sets == gmt.sets
    back = plaid_amd.mat2gmt(M)
    assert back.names == ["S1"] and sorted(back["S1"]) == ["D", "E", "F", "G"]


def test_aligned_pattern_follows_intersect_semantics():
    X = plaid_amd.NamedMatrix(np.arange(10.0).reshape(5, 2), ["g3", "g1", "gX", "g2", "g1"], ["a", "b"])
    Gd = np.array([[1, 0, 1], [1, 1, 0], [0, 2, 0], [1, 0, 0.0]])
    G = plaid_amd.NamedMatrix(sp.csc_matrix(Gd), ["g1", "g2", "g9", "g3"], ["s1", "s2", "s3"])
    Gp, Gi = plaid_amd.aligned_pattern(X, G)
    # rows of X: g1 -> 1 (first occurrence), g2 -> 3, g3 -> 0 ; g9 absent from X is dropped
    assert Gp.tolist() == [0, 3, 4, 5]
    assert sorted(Gi[0:3].tolist()) == [0, 1, 3] and Gi[3] == 3 and Gi[4] == 1
    assert plaid_amd.aligned_pattern(X, plaid_amd.NamedMatrix(sp.csc_matrix(Gd), ["a", "b", "c", "d"])) is None


def test_synthetic_generators_are_reproducible_by_block():
    a = synth.dense_columns(100, 0, 600)
    b = synth.dense_columns(100, 250, 520)
    assert a.flags.f_contiguous and np.array_equal(a[:, 250:520], b)
    assert abs(a.mean() - 8.0) < 0.05 and abs(a.std() - 2.0) < 0.05
    t = synth.dense_columns(100, 3, 9, tied=True)
    assert np.array_equal(t, np.round(a[:, 3:9], 1))
    Gp, Gi = synth.geneset_csc(2000, 300)
    k = np.diff(Gp)
    assert Gp[0] == 0 and k.min() >= 15 and k.max() <= 500 and np.all(np.diff(k) <= 0)
    for j in (0, 150, 299):
        s = Gi[Gp[j]:Gp[j + 1]]
        assert np.all(np.diff(s) > 0) and s.max() < 2000
    Xp, Xi, Xx = synth.sparse_columns(5000, 0, 20)
    nnz = np.diff(Xp)
    assert 150 < nnz.mean() < 350 and Xx.min() > 0 and len(np.unique(Xx[Xp[0]:Xp[1]])) <= 50


def test_shard_bounds_cover_all_columns():
    for n, w in ((10, 3), (8, 8), (5, 8), (1000003, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[r][1] == spans[r + 1][0] for r in range(w - 1))
