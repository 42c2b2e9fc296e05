"""Runs the column-median kernel the library selects (or the one PLAIDHIP_MEDIAN_KERNEL forces in the tools/ build) on
seeded matrices of several shapes and saves the medians, so that two kernels can be compared bit for bit:
    python3 tools/check_medians.py --out a.npz;  PLAIDHIP_MEDIAN_KERNEL=wave PLAIDHIP_LIB=... python3 tools/check_medians.py --out b.npz --against a.npz"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--against", default=None)
    a = ap.parse_args()
    import torch
    import plaid_amd
    dev = torch.device("cuda", 0)
    ctx = plaid_amd.Context(0)
    res = {}
    for m, n in ((5000, 3000), (4999, 257), (1, 5), (2, 9), (63, 100), (64, 100), (65, 100), (1024, 513), (1025, 300), (2048, 300),
                 (3000, 300), (4096, 300), (4097, 300), (5120, 300), (5121, 300), (6144, 300)):
        gen = torch.Generator(device=dev)
        gen.manual_seed(m * 131 + n)
        S = torch.randn((n, m), dtype=torch.float64, device=dev, generator=gen) * 3 + 1
        S[1::7] = torch.round(S[1::7])                      # many ties
        S[2::7][:, ::3] = 0.0                               # exact zeros
        if n > 4:
            S[3] = 0.0                                      # an all-zero column
            S[4] = float("nan")
        S[5::11][:, 1::5] = float("nan")
        S[6::13] = torch.round(S[6::13] * 1e-3) * 1e3       # nearly constant columns
        S[0, 0] = -0.0
        for iz in (None, True, False):
            med = torch.empty(n, dtype=torch.float64, device=dev)
            flags = torch.zeros(4, dtype=torch.int32, device=dev)
            if iz is None:
                flags[0] = 1
            torch.cuda.synchronize()
            ctx.dev_col_medians(S.data_ptr(), m, m, n, iz, med.data_ptr(), flags.data_ptr())
            ctx.synchronize()
            res[f"{m}_{n}_{iz}"] = med.cpu().numpy()
        # numpy as a third opinion on the plain case
        ref = np.nanmedian(S.cpu().numpy(), axis=1)
        got = res[f"{m}_{n}_False"]
        ok = np.array_equal(np.isnan(ref), np.isnan(got)) and np.allclose(ref[~np.isnan(ref)], got[~np.isnan(ref)], rtol=0, atol=0)
        print(f"m={m} n={n}: equals numpy.nanmedian bit for bit: {ok}")
    np.savez(a.out, **res)
    if a.against:
        other = np.load(a.against)
        bad = [k for k in res if not np.array_equal(res[k], other[k], equal_nan=True)]
        print("identical to", a.against, ":", not bad, bad[:5])


if __name__ == "__main__":
    main()
