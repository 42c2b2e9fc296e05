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
    assert lib.plaidhip_version() == 200


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


from oracle import plaid_oracle as po  # noqa: E402  (tests may use the oracle as the checker)


def _random_gmt_text(rng, nsets, pool):
    lines = ["# header comment", ""]
    for j in range(nsets):
        k = int(rng.integers(0, 40))
        genes = [pool[i] for i in rng.integers(0, len(pool), size=k)]          # repeats on purpose
        genes += ["NA"] * int(rng.integers(0, 2)) + [""] * int(rng.integers(0, 2))
        rng.shuffle(genes)
        name = f"SET{int(rng.integers(0, max(2, nsets * 3 // 4)))}"             # repeated names on purpose
        sep = "\t" if j % 3 else " "                                            # genes split on ' ' or tab (:116)
        lines.append(name + "\tsrc" + str(j % 5) + "\t" + sep.join(genes) + ("  # trailing comment" if j % 7 == 0 else ""))
    lines.append("LONELY")                                                      # a name only: empty set
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_native_gmt_matches_oracle_on_random_text(tmp_path, seed):
    """read.gmt / gmt2mat in the native library against the oracle's Python restatement: comments, blank
    lines, "" / NA / repeated genes, repeated set names, empty sets, ntop, max.genes and a given bg"""
    rng = np.random.default_rng(seed)
    pool = [f"G{i}" for i in range(300)] + ["g-a b".replace(" ", "_"), "ÄÖ", "zz"]
    p = tmp_path / "r.gmt"
    p.write_text(_random_gmt_text(rng, 120, pool), encoding="utf-8")
    for add_source, nrows in [(False, -1), (True, -1), (False, 17)]:
        gmt = plaid_amd.read_gmt(str(p), add_source=add_source, nrows=nrows)
        names, sets = po.read_gmt(str(p), add_source=add_source, nrows=nrows)
        assert gmt.names == names and gmt.sets == sets
    gmt = plaid_amd.read_gmt(str(p))
    names, sets = po.read_gmt(str(p))
    bg = [pool[i] for i in rng.permutation(len(pool))[:150]] + ["NOT_A_GENE"]
    for kw in [{}, {"ntop": 5}, {"max_genes": 40}, {"bg": bg}, {"bg": bg, "max_genes": 60, "ntop": 9}]:
        M = plaid_amd.gmt2mat(gmt, **kw)
        D, rn, cn = po.gmt2mat(names, sets, **kw)
        assert M.rownames == rn and M.colnames == cn
        assert (sp.csc_matrix(M.values) != D).nnz == 0
        F = plaid_amd.gmt.gmt2mat_file(str(p), **kw)                              # path -> matrix, no lists in between
        assert F.rownames == rn and F.colnames == cn and (sp.csc_matrix(F.values) != D).nnz == 0


def test_native_gmt2mat_is_fast_at_50k_sets(tmp_path):
    """the R gmt2mat takes 50.9 s for a 50k-set collection; the native path has to do the same
    text -> matrix job in seconds (checked loosely: 20 s on a loaded CI core), with the documented orderings"""
    import time
    from plaid_amd import synth
    g, m = 20000, 50000
    Gp, Gi = synth.geneset_csc(g, m, sort_by_size=False)
    with open(tmp_path / "big.gmt", "w") as fh:
        for j in range(m):
            fh.write(f"S{j}\tsyn\t" + "\t".join(f"G{x}" for x in Gi[Gp[j]:Gp[j + 1]]) + "\n")
    t0 = time.perf_counter()
    M = plaid_amd.gmt.gmt2mat_file(str(tmp_path / "big.gmt"))
    dt = time.perf_counter() - t0
    assert dt < 20.0, f"native gmt2mat took {dt:.1f} s"
    A = sp.csc_matrix(M.values)
    assert A.shape == (g, m) and A.nnz == int(Gp[-1])
    sizes = np.diff(A.indptr)
    assert np.all(np.diff(sizes) <= 0)                                            # sets by decreasing size (:25)
    rs = np.asarray(A.sum(axis=1)).ravel()
    assert np.all(np.diff(rs) <= 0)                                               # genes by decreasing frequency (:62)
    j = M.colnames.index("S123")
    assert sorted(M.rownames[r] for r in A.indices[A.indptr[j]:A.indptr[j + 1]]) == sorted(f"G{x}" for x in Gi[Gp[123]:Gp[124]])


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


def test_distribution_functions_of_the_plaid_test_tail_match_scipy():
    """stats.cpp: 2*pt(|t|, df, lower=FALSE), pchisq(x, 2k, lower=FALSE), qnorm, pnorm upper -- reached through
    a test hook of the library (host code; no device involved)"""
    import ctypes as C
    import scipy.stats as st
    lib = _lib.load()
    f = lib.plaidhip_debug_pvalue
    f.argtypes = [C.c_int, C.c_double, C.c_double]
    f.restype = C.c_double
    rng = np.random.default_rng(5)
    for t, df in zip(np.concatenate([rng.normal(0, 3, 200), [0.0, 1e-9, 40.0, 300.0, -7.5]]),
                     np.concatenate([rng.uniform(1.0, 300.0, 200), [1.0, 2.5, 4.0, 17.3, 1e6]])):
        np.testing.assert_allclose(f(0, abs(t), df), 2 * st.t.sf(abs(t), df), rtol=2e-10, atol=1e-300)
    for x in [0.0, 1e-3, 1.0, 9.2, 55.0, 460.0, 900.0]:
        for k in (1, 2, 3):
            np.testing.assert_allclose(f(1, x, k), st.chi2.sf(x, 2 * k), rtol=1e-12, atol=1e-300)
    for p_ in [1e-99, 1e-20, 1e-6, 0.01, 0.3, 0.5, 0.77, 0.999, 1 - 1e-12]:
        np.testing.assert_allclose(f(2, p_, 0), st.norm.ppf(p_), rtol=1e-13, atol=1e-15)
    for z in [-38.0, -5.0, -0.3, 0.0, 1.7, 8.0, 21.0]:
        np.testing.assert_allclose(f(3, z, 0), st.norm.sf(z), rtol=1e-13, atol=1e-300)
    assert np.isnan(f(0, 1.0, 0.0)) and f(0, np.inf, 3.0) == 0.0


def test_oracle_twosample_t_is_welch_t_of_set_vs_rest():
    """R/plaid.R:488-520: the statistic is the Welch t of the set's logFC against all other genes (up to the
    1e-8 guards); its degrees of freedom are the reference's own formula, kept as written (":510 NEED CHECKING")"""
    import scipy.stats as st
    rng = np.random.default_rng(1)
    g, m = 400, 12
    F = rng.normal(0, 1, g)
    G = sp.random(g, m, density=0.1, random_state=3, format="csc")
    f, t, p = po.matrix_twosample_ttest(F, G)
    Gd = (G.toarray() != 0)
    for j in range(m):
        a, b = F[Gd[:, j]], F[~Gd[:, j]]
        ref = st.ttest_ind(a, b, equal_var=False)
        np.testing.assert_allclose(t[j], ref.statistic, rtol=1e-6)
        np.testing.assert_allclose(f[j], a.mean() - b.mean(), rtol=1e-6)
    q = po.p_adjust_fdr(p)
    o = np.argsort(p)
    assert np.all(np.diff(q[o]) >= -1e-15) and np.all(q >= p - 1e-15) and np.all(q <= 1)


@pytest.mark.parametrize("g,m", [(500, 70), (10224, 300), (20000, 700), (30001, 200)])
def test_pair_plan_schedules_every_membership_once(g, m):
    """host schedule of the two-columns-per-pass SpMM (geneset.cpp): every membership appears exactly once, in the
    stream of the wavefront that owns its tile, for 1-3 gene slices; the only LDS conflicts are the deliberate
    second reads of over-full slots (a small fraction of the steps), none with PLAIDHIP_CONFLICT_COST >= 1"""
    import ctypes as C
    lib = _lib.load()
    fn = lib.plaidhip_debug_pair_plan_check
    fn.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    fn.restype = C.c_int
    Gp, Gi = synth.geneset_csc(g, m, kmin=1, kmax=min(g, 400), sort_by_size=False)
    out = (C.c_int64 * 8)()
    assert fn(g, m, Gp.ctypes.data, Gi.ctypes.data, 16, out) == 0
    slices, chunks, found, conflicts, wrong = out[0], out[1], out[2], out[3], out[4]
    assert slices == (g + 10223) // 10224
    assert found == int(Gp[-1]) and wrong == 0
    assert conflicts <= 0.25 * chunks * 8                       # a minority of the steps carries one
    if m >= 300:
        assert int(Gp[-1]) / (chunks * 512) > 0.5               # slot efficiency stays sane (full tiles)


def test_product_library_has_no_diagnostic_surface():
    """ablation / stamp kernel variants, the env-var tuning knobs and plaidhip_debug_set_ablation exist only in the
    tools/ build (-DPLAIDHIP_DIAG, `make diag`): the product library neither exports the setter nor reads the
    environment on any launch path"""
    import re
    from plaid_amd import _lib
    lib = _lib.load()
    assert not hasattr(lib, "plaidhip_debug_set_ablation")
    csrc = os.path.join(ROOT, "plaid_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".cpp", ".hip", ".h")):
            continue
        depth, diag = 0, []
        for line in open(os.path.join(csrc, name)):
            t = line.strip()
            if t.startswith("#if"):
                diag.append("PLAIDHIP_DIAG" in t and not t.startswith("#ifndef"))
            elif t.startswith("#else") and diag:
                diag[-1] = False
            elif t.startswith("#endif") and diag:
                diag.pop()
            elif re.search(r"\bgetenv\s*\(", t) and not t.startswith("//"):
                assert any(diag), f"{name}: getenv outside #ifdef PLAIDHIP_DIAG: {t}"


def test_int32_slots_refuse_values_that_do_not_fit():
    """scipy hands over int64 index arrays for big matrices: wrapping them into the 32-bit dgCMatrix slots of the C ABI
    would read garbage; the host layer refuses instead (the R-side answer is chunking, R/plaid.R:100-123)"""
    import numpy as np
    from plaid_amd import _lib
    from plaid_amd.engine import _as_i32
    assert _as_i32(np.array([0, 5, 2**31 - 1], dtype=np.int64)).dtype == np.int32
    with pytest.raises(_lib.PlaidHipError):
        _as_i32(np.array([0, 2**31], dtype=np.int64))


def test_c_abi_shard_bounds_match_the_python_side():
    """plaidhip_shard_bounds (the multi-device host entry's shard rule) needs no device; it is the same rule as
    sharded.shard_bounds (the RCCL path): contiguous blocks of ceil(n / ndev) columns, trailing shards short or empty"""
    for n in (0, 1, 7, 8, 9, 1000, 100000):
        for ndev in (1, 2, 3, 8):
            cover = []
            for k in range(ndev):
                lo, hi = plaid_amd.shard_bounds(n, ndev, k)
                assert (lo, hi) == shard_bounds(n, ndev, k)
                cover += list(range(lo, hi)) if n <= 1000 else []
            if n <= 1000:
                assert cover == list(range(n))
    with pytest.raises(_lib.PlaidHipError):
        plaid_amd.shard_bounds(5, 0, 0)


def _split_top_level(argstr):
    out, depth, cur, quote = [], 0, "", None
    for ch in argstr:
        if quote:
            cur += ch
            if ch == quote:
                quote = None
            continue
        if ch in "\"'":
            quote = ch
            cur += ch
        elif ch in "([{":
            depth += 1
            cur += ch
        elif ch in ")]}":
            depth -= 1
            cur += ch
        elif ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _call_sites(text, opener):
    """argument strings of every `opener(` ... matching `)` in text"""
    i = 0
    while True:
        i = text.find(opener + "(", i)
        if i < 0:
            return
        j = i + len(opener) + 1
        depth, k, quote = 1, j, None
        while depth:
            ch = text[k]
            if quote:
                quote = None if ch == quote else quote
            elif ch in "\"'":
                quote = ch
            elif ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            k += 1
        yield text[j:k - 1]
        i = k


def test_r_shim_is_consistent_without_r():
    """R is not installed here, so the .Call shim is never compiled: check statically that every .Call in
    r-pkg/R/plaid-hip.R names a routine registered in r-pkg/src/plaidhip_R.c with that many arguments, that every
    registered routine is defined with that many SEXP parameters, that every C-ABI function the shim calls is declared
    in include/plaidhip.h with the number of arguments the shim passes, and that NAMESPACE exports functions that exist"""
    rsrc = open(os.path.join(ROOT, "r-pkg", "R", "plaid-hip.R")).read()
    rsrc = "\n".join(line.split("##")[0] if line.lstrip().startswith("##") else line for line in rsrc.splitlines())
    csrc = open(os.path.join(ROOT, "r-pkg", "src", "plaidhip_R.c")).read()
    header = open(os.path.join(ROOT, "include", "plaidhip.h")).read()
    registered = {m.group(1): int(m.group(2))
                  for m in re.finditer(r'\{"(R_plaidhip_\w+)",\s*\(DL_FUNC\)&\1,\s*(\d+)\}', csrc)}
    assert len(registered) >= 18
    defined = {}
    for m in re.finditer(r"^SEXP (R_plaidhip_\w+)\(([^)]*)\)\s*\{", csrc, re.M | re.S):
        defined[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip().startswith("SEXP")])
    assert registered == defined, "registration table and definitions disagree"
    calls = 0
    for args in _call_sites(rsrc, ".Call"):
        parts = _split_top_level(args)
        name = parts[0].strip("\"")
        nargs = len([a for a in parts[1:] if not a.startswith("PACKAGE")])
        assert name in registered, f".Call of an unregistered routine {name}"
        assert registered[name] == nargs, f".Call({name}) passes {nargs} arguments, {registered[name]} registered"
        calls += 1
    assert calls >= 20
    # C-ABI calls of the shim against the header
    code = re.sub(r"/\*.*?\*/", "", csrc, flags=re.S)
    hdr = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    proto = {}
    for m in re.finditer(r"\b(?:int|const char\*|const int32_t\*|int64_t)\s+(plaidhip_\w+)\s*\(([^;]*?)\)\s*;", hdr, re.S):
        a = m.group(2).strip()
        proto[m.group(1)] = 0 if a in ("", "void") else len(_split_top_level(a))
    used = set()
    for m in re.finditer(r"\b(plaidhip_\w+)\s*\(", code):
        name = m.group(1)
        if name in ("plaidhip_ctx", "plaidhip_gmt", "plaidhip_gmtmat"):
            continue
        assert name in proto, f"the shim calls {name}, which include/plaidhip.h does not declare"
        args = next(_call_sites(code[m.start():], name))
        n = 0 if args.strip() == "" else len(_split_top_level(args))
        assert n == proto[name], f"{name}: the shim passes {n} arguments, the header declares {proto[name]}"
        used.add(name)
    assert {"plaidhip_plaid_dense", "plaidhip_colranks_csc_dense", "plaidhip_plaid_multi", "plaidhip_ssgsea_multi",
            "plaidhip_set_precision"} <= used
    ns = open(os.path.join(ROOT, "r-pkg", "NAMESPACE")).read()
    exported = re.findall(r"[\w.]+", ns.split("export(")[1].split(")")[0])
    for fn in exported:
        assert re.search(r"^" + re.escape(fn) + r"\s*<-\s*function", rsrc, re.M), f"NAMESPACE exports {fn}, not defined"


def test_host_code_is_clean_under_asan_and_ubsan():
    """sanitizers on the CPU build.  `make host-asan` compiles the library's host code -- the index
    planners of geneset.cpp (incl. a 50,000-set collection and every slice shape), the GMT parser / gmt2mat, the p-value
    tails -- with g++ -fsanitize=address,undefined against host stand-ins for the HIP runtime and runs it; any report
    aborts the binary."""
    import subprocess
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "plaid_amd", "csrc"), "host-asan"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "[host-asan] ok" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


def test_r_shim_goes_through_a_c_compiler():
    """r-pkg/src/plaidhip_R.c cannot be built here (no R): it is at least compiled (-fsyntax-only -Wall -Wextra) against
    test-only declarations of the R API functions it uses and against include/plaidhip.h -- argument counts and types of
    every plaidhip_* call, every registration entry, every PROTECT / allocMatrix use are checked by the compiler"""
    import subprocess
    out = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
                          "-I" + os.path.join(ROOT, "tests", "r_api_stub"), "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "r-pkg", "src", "plaidhip_R.c")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


def test_planners_take_the_reference_shaped_collection(tmp_path):
    """The reference benchmarks with playdata::GSETxGENE: 61,459 real gene sets (experiments/benchmark/benchmark-plaid.R:18-35),
    not 15-500-gene synthetic ones.  `make plan-probe` builds geneset.cpp for the host (HIP stand-ins) and runs
    plaidhip_geneset_create on a collection of that shape (synth.geneset_csc_real: sizes 3 ... 5,000 + an all-genes set,
    Zipf gene popularity, duplicated sets): every membership is scheduled exactly once in the pair plan and in the scatter
    plan's id lists, within stated time / memory / padding budgets."""
    import json
    import subprocess
    import numpy as np
    from plaid_amd import synth
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "plaid_amd", "csrc"), "plan-probe"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    probe = os.path.join(ROOT, "plaid_amd", "csrc", "diag", "plan_probe")
    for g, m, gen in ((12010, 61459, synth.geneset_csc_real), (25000, 3000, synth.geneset_csc_real),
                      (20000, 5000, synth.geneset_csc)):
        Gp, Gi = gen(g, m)
        path = str(tmp_path / "gs.bin")
        with open(path, "wb") as f:
            f.write(np.array([g, m], dtype=np.int32).tobytes())
            f.write(Gp.tobytes())
            f.write(Gi.tobytes())
        r = subprocess.run([probe, path, "check"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        d = json.loads(r.stdout)
        z = int(Gp[-1])
        assert d["rc"] == 0 and d["z"] == z
        assert d["pair_found"] == z and d["pair_wrong"] == 0            # pair plan: each membership once, right set and slice
        assert d["scatter_found"] == z and d["scatter_wrong"] == 0      # scatter plan: each membership once, right gene and chunk
        assert d["create_s"] < 30.0, d                                   # (2.5 s here; R's gmt2mat alone: 50.9 s for 50k sets)
        assert d["device_bytes"] < 400e6 + 64.0 * z, d                   # index lists + metadata + the partial-sum scratch
        real = gen is synth.geneset_csc_real
        if m >= 5000:   # (with few sets the all-genes set's tile -- as long as its longest lane -- dominates the padding)
            assert z / d["slots_one_column"] > (0.70 if real else 0.85), d   # slot efficiency (real / padded index slots)
            assert z / d["slots_pair"] > (0.60 if real else 0.78), d
        assert d["pair_conflicts"] < 0.01 * z                            # deliberate two-way conflicts of over-full slots only
        assert d["scatter_collisions"] < 0.08 * z                        # ids sharing an LDS bank inside a 16-lane group


def test_a_cpp_exception_becomes_an_error_code():
    """every `int plaidhip_*` entry point is a function-try-block (common.h: on_exception): a C++ exception inside the
    library -- here std::length_error from a string of 2^62 bytes, thrown before anything is read -- comes back as a code
    and a message instead of unwinding into the caller's C stack (an R session would be gone)"""
    import ctypes as C
    from plaid_amd import _lib
    lib = _lib.load()
    out = C.c_void_p()
    rc = lib.plaidhip_gmt_parse(C.c_char_p(b"a\tb\tc\n"), C.c_int64(1 << 62), 0, 0, C.c_int64(-1), C.byref(out))
    assert rc != 0 and not out.value
    msg = lib.plaidhip_last_error_string().decode()
    assert "exception" in msg or "memory" in msg, msg
    # and the sources: no int entry point without the handler
    import re
    csrc = os.path.join(ROOT, "plaid_amd", "csrc")
    for name in ("api.cpp", "multi.cpp", "geneset.cpp", "gmt.cpp"):
        text = open(os.path.join(csrc, name)).read()
        for m_ in re.finditer(r'^(?:extern "C" )?int (plaidhip_\w+)\([^;{]*\)\s*(try\s*)?\{', text, flags=re.M):
            body_one_line = text[m_.end():text.index("\n", m_.end())].strip().endswith("}")
            assert m_.group(2) or body_one_line, f"{name}: {m_.group(1)} has no function-try-block"
