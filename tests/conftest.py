import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def pbmc():
    """The reference's bundled 50-cell fixture as plain arrays (+ oracle outputs)."""
    d = dict(np.load(os.path.join(GOLDEN, "pbmc3k50.npz"), allow_pickle=False))
    e = dict(np.load(os.path.join(GOLDEN, "pbmc3k50_expected.npz"), allow_pickle=False))
    return d, e


@pytest.fixture(scope="session")
def synth():
    return dict(np.load(os.path.join(GOLDEN, "synthetic_cases.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def hip_ctx():
    """One device context for the GPU tests; fails loudly if the HIP library is missing."""
    import plaid_amd
    ctx = plaid_amd.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture
def pinned_ctx(hip_ctx):
    """the session context with kernel-selection options pinned for one test (plaidhip_set_option), reset afterwards"""
    defaults = {"spmm_dense_kernel": "auto", "spmm_sparse_kernel": "auto", "nt_store": "auto", "ranks_f32": 2,
                "rank_kernel": "auto", "scatter_fixed": "on", "scatter_order": "chunk", "fused_medians": "auto"}

    def pin(**opts):
        for k, v in defaults.items():
            hip_ctx.set_option(k, opts.get(k, v))
        return hip_ctx

    yield pin
    for k, v in defaults.items():
        hip_ctx.set_option(k, v)
