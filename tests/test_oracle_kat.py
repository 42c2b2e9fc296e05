"""The oracle against the reference's only numeric pins: the vignette outputs printed in
doc/plaid-vignette.html (lines 798, 809, 857-869) for the bundled fixture pair
inst/extdata/hallmarks.gmt + inst/extdata/pbmc3k-50cells.rda (committed here as data under
tests/golden/).  p.lm is a Welch t-test on the MEDIAN-NORMALISED plaid() scores, so it pins
plaid() + normalize_medians() including the ignore-zero rule; p.one pins the G^T.v crossprod.
The HTML's gsetFC column comes from an older formula and is not used."""
import os

import numpy as np
import scipy.sparse as sp

from oracle import plaid_oracle as po

# doc/plaid-vignette.html:857-869  (head(res) sorted by p.meta; Stouffer, html:875)
KAT = {
    "HALLMARK_INTERFERON_GAMMA_RESPONSE": (0.003668116, 8.246828e-06, 3.868049e-07),
    "HALLMARK_ALLOGRAFT_REJECTION": (0.102407488, 1.071307e-05, 4.781538e-05),
    "HALLMARK_P53_PATHWAY": (0.038355508, 1.906952e-04, 8.369509e-05),
    "HALLMARK_INTERFERON_ALPHA_RESPONSE": (0.032562973, 9.261621e-03, 1.491854e-03),
    "HALLMARK_PEROXISOME": (0.016625538, 4.052692e-02, 3.080580e-03),
    "HALLMARK_G2M_CHECKPOINT": (0.012385507, 6.049535e-02, 3.638628e-03),
}
KAT_ORDER = list(KAT)
# doc/plaid-vignette.html:864-869: q.meta = p.adjust(p.meta, "fdr") over all 50 sets (R/plaid.R:463) -- pins the BH step,
# incl. the tie 3.032190e-02 of ranks 5 and 6 (the cumulative minimum from the largest p downwards)
QMETA = [1.934024e-05, 1.195384e-03, 1.394918e-03, 1.864818e-02, 3.032190e-02, 3.032190e-02]


def _load(pbmc, golden_dir):
    d, e = pbmc
    X = sp.csc_matrix((d["x"], d["i"], d["p"]), shape=tuple(d["dim"]))
    names, gsets = po.read_gmt(os.path.join(golden_dir, "hallmarks.gmt"))
    D, grn, gcn = po.gmt2mat(names, gsets)
    return d, e, X, D, grn, gcn


def test_dims_match_vignette(pbmc, golden_dir):
    d, e, X, D, grn, gcn = _load(pbmc, golden_dir)
    assert X.shape == (7728, 50) and X.nnz == 38744
    assert D.shape == (4386, 50)                                   # html:798
    S = po.plaid(X, list(d["rownames"]), D, grn)
    assert S.shape == (50, 50)                                     # html:809
    raw = po.plaid(X, list(d["rownames"]), D, grn, normalize=False)
    assert (raw == 0).sum() == 119                                 # exact zeros -> ignore.zero branch taken
    assert len(po.align(list(d["rownames"]), grn)[0]) == 2217


def test_vignette_pvalues(pbmc, golden_dir):
    d, e, X, D, grn, gcn = _load(pbmc, golden_dir)
    rn = list(d["rownames"])
    S = po.plaid(X, rn, D, grn)
    y = (d["celltype"] == "B").astype(int)
    res = po.plaid_test(X, rn, y, D, grn, S, metap_method="stouffer")
    idx = {nm: k for k, nm in enumerate(gcn)}
    for nm, (p_one, p_lm, p_meta) in KAT.items():
        k = idx[nm]
        # printed with 7 significant digits
        np.testing.assert_allclose(res["p.one"][k], p_one, rtol=2e-6)
        np.testing.assert_allclose(res["p.lm"][k], p_lm, rtol=2e-6)
        np.testing.assert_allclose(res["p.meta"][k], p_meta, rtol=2e-6)
    top6 = [gcn[k] for k in np.argsort(res["p.meta"])[:6]]
    assert top6 == KAT_ORDER
    for nm, q in zip(KAT_ORDER, QMETA):                              # html:864-869
        np.testing.assert_allclose(res["q.meta"][idx[nm]], q, rtol=2e-6)
    assert res["q.meta"][idx[KAT_ORDER[4]]] == res["q.meta"][idx[KAT_ORDER[5]]]   # the BH tie, exactly


def test_scse_on_the_fixture_removes_log2(pbmc, golden_dir):
    """doc/plaid-vignette.html:920-921: the vignette runs replaid.scse(X, matG, removeLog2=TRUE, ...) on this very matrix and
    the function reports that it removes the log2.  The automatic rule of R/plaid.R:160-161 (min(X) == 0 && max(X) < 20)
    must come to the same decision on it: the fixture is log-scale data with implicit zeros."""
    d, e, X, D, grn, gcn = _load(pbmc, golden_dir)
    rn = list(d["rownames"])
    assert X.nnz < X.shape[0] * X.shape[1] and X.data.min() > 0 and X.data.max() < 20
    auto = po.replaid_scse(X, rn, D, grn, remove_log2=None)
    np.testing.assert_array_equal(auto, po.replaid_scse(X, rn, D, grn, remove_log2=True))
    assert np.abs(auto - po.replaid_scse(X, rn, D, grn, remove_log2=False)).max() > 1e-3


def test_wrong_median_rule_breaks_the_pin(pbmc, golden_dir):
    """Sanity of the pin itself: medians that do NOT ignore zeros change p.lm beyond the
    printed precision, so the KAT really constrains the ignore-zero branch."""
    d, e, X, D, grn, gcn = _load(pbmc, golden_dir)
    rn = list(d["rownames"])
    raw = po.plaid(X, rn, D, grn, normalize=False)
    wrong, _ = po.normalize_medians(raw, ignore_zero=False)
    y = (d["celltype"] == "B").astype(int)
    p = po.welch_ttests(wrong, y)
    k = gcn.index("HALLMARK_INTERFERON_GAMMA_RESPONSE")
    assert abs(p[k] / 8.246828e-06 - 1) > 1e-4


def test_committed_expected_outputs_are_current(pbmc, golden_dir):
    """tests/golden/pbmc3k50_expected.npz is what oracle/make_golden.py writes."""
    d, e, X, D, grn, gcn = _load(pbmc, golden_dir)
    rn = list(d["rownames"])
    assert np.array_equal(D.indptr, e["G_p"]) and np.array_equal(D.indices, e["G_i"])
    np.testing.assert_allclose(po.plaid(X, rn, D, grn), e["plaid_norm"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(po.replaid_sing(X, rn, D, grn), e["sing"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(po.replaid_ssgsea(X, rn, D, grn, alpha=0.25), e["ssgsea_a025"], rtol=1e-12, atol=1e-14)
    assert np.array_equal(po.sparse_colranks(X).data, e["sparse_colranks_avg"])
