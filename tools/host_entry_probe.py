import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, plaid_amd
from plaid_amd import synth, _lib
from plaid_amd.engine import _np_ptr
g, n, m = 20000, 10000, 5000
Gp, Gi = synth.geneset_csc(g, m)
X = np.asfortranarray(np.random.default_rng(0).normal(8, 2, size=(g, n)))
ctx = plaid_amd.Context(0)
ctx.plaid_dense(X[:, :256], Gp, Gi)
for label, S in (("fresh np.empty S", None), ("pre-touched S", np.zeros((m, n), order="F"))):
    ts = []
    for _ in range(3):
        if label.startswith("fresh"):
            S = np.empty((m, n), dtype=np.float64, order="F")
        t0 = time.perf_counter()
        _lib.check(ctx.lib.plaidhip_plaid_dense(ctx.handle, _np_ptr(X), g, n, _np_ptr(Gp), _np_ptr(Gi), m, 0, 1, _np_ptr(S)))
        ts.append(time.perf_counter() - t0)
    print(label, [round(t * 1e3, 1) for t in ts])
S = np.zeros((m, n), order="F")
for nn in (2500, 5000, 10000):
    t0 = time.perf_counter()
    _lib.check(ctx.lib.plaidhip_plaid_dense(ctx.handle, _np_ptr(X), g, nn, _np_ptr(Gp), _np_ptr(Gi), m, 0, 1, _np_ptr(S)))
    print("n", nn, round((time.perf_counter() - t0) * 1e3, 1), "ms")
