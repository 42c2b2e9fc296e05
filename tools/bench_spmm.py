#!/usr/bin/python3
"""Micro-benchmark of single kernels (for rocprofv3 --pmc passes and A/B work):
   python tools/bench_spmm.py [--kernel spmm|medians|ranks] [--genes G --samples N --sets M] [--iters K]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="spmm")
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--sets", type=int, default=5000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ties", default="average")
    ap.add_argument("--precision", default="f64", choices=["f64", "mixed"])
    ap.add_argument("--real-sets", action="store_true", help="a collection with the shape of the reference's (synth.geneset_csc_real: sizes 3..5,000 + an all-genes set, hub genes)")
    ap.add_argument("--unsorted", action="store_true", help="gene sets in random order (not by decreasing size)")
    ap.add_argument("--ablate", type=int, default=0, help="diagnostic SpMM variant 1..7 (wrong results by design; needs PLAIDHIP_LIB=<the make diag library>)")
    ap.add_argument("--fused", action="store_true", help="c3 / c4: medians selected inside the crossprod launch (dev_spmm_csc_fused / dev_spmm_dense_fused + dev_col_medians_resume)")
    ap.add_argument("--stamps", action="store_true", help="in-kernel phase stamps of the scatter kernel (diag library)")
    ap.add_argument("--dense-kernel", default="auto", choices=["auto", "single", "pair", "mfma"])
    ap.add_argument("--sparse-kernel", default="auto", choices=["auto", "scatter", "gather"])
    ap.add_argument("--nt-store", default="auto", choices=["auto", "off", "on"], help="non-temporal stores of the scores")
    ap.add_argument("--scatter-fixed", default="on", choices=["on", "off"], help="c3: u64 fixed-point accumulators in the scatter kernel")
    ap.add_argument("--scatter-order", default="chunk", choices=["chunk", "column"], help="c3: item order of the scatter kernel")
    a = ap.parse_args()
    import numpy as np
    import torch
    import plaid_amd
    from plaid_amd import synth
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx = plaid_amd.Context(0, stream.cuda_stream)
    ctx.set_precision(a.precision)
    ctx.set_option("spmm_dense_kernel", a.dense_kernel)
    ctx.set_option("spmm_sparse_kernel", a.sparse_kernel)
    ctx.set_option("nt_store", a.nt_store)
    ctx.set_option("scatter_fixed", a.scatter_fixed)
    ctx.set_option("scatter_order", a.scatter_order)
    if a.fused:
        ctx.set_option("fused_medians", "on")   # (by default only from 1e9 scores on)
    dbg = None
    if a.ablate or a.stamps:
        import ctypes
        dbg = torch.zeros(4096 * 16 * 4, dtype=torch.int64, device=dev)
        ctx.lib.plaidhip_debug_set_ablation.argtypes = [ctypes.c_int, ctypes.c_void_p]
        ctx.lib.plaidhip_debug_set_ablation(a.ablate if a.ablate else 100, dbg.data_ptr())
    g, n, m = a.genes, a.samples, a.sets
    t0 = time.perf_counter()
    Gp, Gi = synth.geneset_csc_real(g, m) if a.real_sets else synth.geneset_csc(g, m, sort_by_size=not a.unsorted)
    t1 = time.perf_counter()
    gs = ctx.geneset(g, Gp, Gi)
    t2 = time.perf_counter()
    info = gs.info()
    print(f"geneset: gen {t1 - t0:.2f}s prepare {t2 - t1:.2f}s info {info} "
          f"slot efficiency {info['z'] / max(info['padded_slots'], 1):.3f}")
    if a.kernel in ("c3", "c4"):
        # BASELINE configs 3 / 4 at a reduced sample count: replaid.ssgsea on sparse (C3: ranks of the
        # non-zeros, CSC SpMM) or dense (C4: dense ranks ^1.25, SpMM) input, 50k-set collections
        def ev():
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            return e
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        med = torch.empty(n, dtype=torch.float64, device=dev)
        red = torch.zeros(4, dtype=torch.float64, device=dev)
        colmax = torch.empty(n, dtype=torch.float64, device=dev)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        if a.kernel == "c3":
            Xp, Xi, Xx = synth.sparse_columns(g, 0, n)
            dXp = torch.from_numpy(Xp.astype(np.int32)).to(dev)
            dXi = torch.from_numpy(Xi).to(dev)
            dXx = torch.from_numpy(Xx).to(dev)
            dRx = torch.empty_like(dXx)
            max_nnz = int(np.diff(Xp).max())
            print(f"sparse X: nnz {len(Xx)} ({len(Xx)/n:.0f} per cell, longest {max_nnz})")
        else:
            X = torch.randn((n, g), dtype=torch.float64, device=dev) * 2 + 8
            R = torch.empty_like(X)
        for it in range(a.iters):
            with torch.cuda.stream(stream):
                flags.zero_()
                e0 = ev()
                if a.kernel == "c3":
                    ctx.dev_colranks_csc(dXp.data_ptr(), dXx.data_ptr(), n, max_nnz, dRx.data_ptr(), "average", False, 1.25, colmax.data_ptr())
                else:
                    ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R.data_ptr(), g, "average", False, 1.25, colmax.data_ptr())
                ctx.dev_max(colmax.data_ptr(), n, red.data_ptr() + 16)
                e1 = ev()
                if a.kernel == "c3" and a.fused:
                    ctx.dev_spmm_csc_fused(gs, dXp.data_ptr(), dXi.data_ptr(), dRx.data_ptr(), n, S.data_ptr(), m, "mean", 1.0, -0.5,
                                           flags.data_ptr(), None, red.data_ptr() + 16, nnz=len(Xx))
                elif a.kernel == "c3":
                    ctx.dev_spmm_csc_ranks(gs, dXp.data_ptr(), dXi.data_ptr(), dRx.data_ptr(), n, S.data_ptr(), m,
                                           red.data_ptr() + 16, "mean", 1.0, -0.5, flags.data_ptr(), nnz=len(Xx))
                elif a.fused:
                    ctx.dev_spmm_dense_fused(gs, R.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, -0.5, flags.data_ptr(),
                                             red.data_ptr() + 16)
                else:
                    ctx.dev_spmm_dense(gs, R.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, -0.5, flags.data_ptr(),
                                       red.data_ptr() + 16)
                e2 = ev()
                if a.fused:
                    ctx.dev_col_medians_resume(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
                else:
                    ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
                e2b = ev()
                ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
                ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
                e3 = ev()
            torch.cuda.synchronize()
            if dbg is not None and a.kernel == "c3" and it == a.iters - 1:
                d = dbg.cpu().numpy()[: 256 * 16 * 8].reshape(256, 16, 8).astype(float)
                tot = d[:, :, :8].sum(axis=2).mean()
                names = ["between", "walk: prefetch wait", "barrier1", "epilogue", "barrier2", "walk: first ids in hand",
                         "walk: static pipeline", "walk: further segments"]
                print("  scatter stamps (cycles per launch, mean over workgroups x waves): " +
                      "  ".join(f"{nm} {d[:, :, k].mean():.0f} ({100 * d[:, :, k].mean() / tot:.0f}%)" for k, nm in enumerate(names)))
                print("  walk cycles by wave, WG 0:", (d[0, :, 1] + d[0, :, 5] + d[0, :, 6] + d[0, :, 7]).astype(int).tolist())
                for k, nm in enumerate(names):
                    print(f"  {nm} by wave, WG 5:", d[5, :, k].astype(int).tolist())
            print(f"{a.kernel} ({g}x{n}x{m}): rank {e0.elapsed_time(e1):.3f} ms  spmm {e1.elapsed_time(e2):.3f} ms  "
                  f"normalize {e2.elapsed_time(e3):.3f} ms (medians {e2.elapsed_time(e2b):.3f})  total {e0.elapsed_time(e3):.3f} ms -> "
                  f"{m*n/e0.elapsed_time(e3)/1e-3:.3e} scores/s")
        return
    if a.kernel == "sing":
        # crossprod of a rank matrix (replaid.sing) under the three exact stagings: u16 (default), fp32, fp64
        X = torch.round((torch.randn((n, g), dtype=torch.float64, device=dev) * 2 + 8) * 10) / 10
        R = torch.empty_like(X)
        S = torch.empty((n, m), dtype=torch.float64, device=dev)
        ref = None
        with torch.cuda.stream(stream):
            ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R.data_ptr(), g, "average", False, 1.0, None)
        for name in ("u16", "f32", "f64"):
            ctx.set_option("ranks_f32", name)
            ms = []
            for it in range(a.iters + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(stream):
                    e0.record(stream)
                    ctx.dev_spmm_ranks(gs, R.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0 / g, -0.5, None)
                    e1.record(stream)
                torch.cuda.synchronize()
                if it:
                    ms.append(e0.elapsed_time(e1))
            if ref is None:
                ref = S.clone()
            print(f"sing crossprod ({g}x{n}x{m}) staging {name}: min {min(ms):.3f} ms median {sorted(ms)[len(ms)//2]:.3f} ms "
                  f"-> {m*n/min(ms)/1e-3:.3e} scores/s; bit-identical to u16: {bool(torch.equal(S, ref))}")
            if dbg is not None and a.ablate == 4:
                waves = info["waves"]
                nwg = min((n + 3) // 4, 256)
                d = dbg.cpu().numpy()[: nwg * waves * 4].reshape(nwg, waves, 4).astype(float)
                tot = d[:, :, 3].mean()
                print(f"  stamps: stage {100*d[:,:,0].mean()/tot:.1f}% gather {100*d[:,:,1].mean()/tot:.1f}% "
                      f"end-barrier wait {100*d[:,:,2].mean()/tot:.1f}% total {tot:.0f} cycles")
        return
    if a.kernel == "host_plaid":
        # PCIe-inclusive rate of the host-pointer entry point (what R's .Call binds)
        Xh = np.asfortranarray(np.random.default_rng(0).normal(8, 2, size=(g, n)))
        ctx.plaid_dense(Xh[:, :64], Gp, Gi)
        t = []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            ctx.plaid_dense(Xh, Gp, Gi)
            t.append(time.perf_counter() - t0)
        print(f"host_plaid (H2D X + G prep + kernels + D2H S): min {min(t)*1e3:.1f} ms -> {m*n/min(t):.3e} scores/s "
              f"({(g*n*8 + m*n*8)/min(t)/1e9:.1f} GB/s over PCIe incl. everything)")
        return
    X = torch.randn((n, g), dtype=torch.float64, device=dev) * 2 + 8
    if a.kernel == "ranks":
        X = torch.round(X * 10) / 10
    S = torch.empty((n, m), dtype=torch.float64, device=dev)
    R = torch.empty_like(X) if a.kernel == "ranks" else None
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    med = torch.empty(n, dtype=torch.float64, device=dev)
    with torch.cuda.stream(stream):
        ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, flags.data_ptr())
    torch.cuda.synchronize()
    mdbg = None
    if a.kernel == "medians" and hasattr(ctx.lib, "plaidhip_debug_set_median_stamps"):
        import ctypes
        mdbg = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device=dev)
        ctx.lib.plaidhip_debug_set_median_stamps.argtypes = [ctypes.c_void_p]
        ctx.lib.plaidhip_debug_set_median_stamps(mdbg.data_ptr())
    if a.kernel == "step":
        # the plaid() step phase by phase (crossprod / medians + mean / shift), as bench.py's C2 block enqueues it
        red = torch.zeros(2, dtype=torch.float64, device=dev)
        rows = []
        for k in range(a.iters + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            with torch.cuda.stream(stream):
                flags.zero_()
                e[0].record(stream)
                ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, flags.data_ptr())
                e[1].record(stream)
                ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
                ctx.dev_sum(med.data_ptr(), n, red.data_ptr())
                e[2].record(stream)
                ctx.dev_shift_columns(S.data_ptr(), m, m, n, med.data_ptr(), 0.0, red.data_ptr())
                e[3].record(stream)
            torch.cuda.synchronize()
            if k:
                rows.append([e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]), e[0].elapsed_time(e[3])])
        r = np.array(rows)
        print(f"step ({g}x{n}x{m}) nt={a.nt_store}: min ms  spmm {r[:,0].min():.4f}  "
              f"medians+sum {r[:,1].min():.4f}  shift {r[:,2].min():.4f}  total {r[:,3].min():.4f}  (median total {np.median(r[:,3]):.4f})")
        return
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for k in range(a.iters):
        with torch.cuda.stream(stream):
            ev[k][0].record(stream)
            if a.kernel == "spmm":
                ctx.dev_spmm_dense(gs, X.data_ptr(), g, n, S.data_ptr(), m, "mean", 1.0, 0.0, flags.data_ptr())
            elif a.kernel == "medians":
                ctx.dev_col_medians(S.data_ptr(), m, m, n, None, med.data_ptr(), flags.data_ptr())
            elif a.kernel == "ranks":
                ctx.dev_colranks_dense(X.data_ptr(), g, g, n, R.data_ptr(), g, a.ties, False, 1.0, None)
            ev[k][1].record(stream)
    torch.cuda.synchronize()
    ms = [e[0].elapsed_time(e[1]) for e in ev]
    print(f"{a.kernel}: ms per launch min {min(ms):.4f} median {sorted(ms)[len(ms) // 2]:.4f} ({g}x{n}x{m})")
    if a.kernel == "medians":
        print("  flags words (3 = bracket misses over all launches):", flags.cpu().tolist())
        if mdbg is not None and m > 6144:
            d = mdbg.cpu().numpy().reshape(-1, 8).astype(float)
            d = d[d[:, :6].sum(axis=1) > 0]
            tot = d[:, :6].sum(axis=1).mean()
            names = ("sample+sort", "classify sweep", "scan", "generic/hist sweeps", "collect+sort", "upper-middle sweep")
            print(f"  stamps (mean over {len(d)} wavefronts, last launch; cycles per wavefront, {n / max(len(d), 1):.2f} columns each): "
                  + "  ".join(f"{nm} {d[:, q].mean():.0f} ({100 * d[:, q].mean() / tot:.0f}%)" for q, nm in enumerate(names)))
        elif mdbg is not None:
            d = mdbg.cpu().numpy().reshape(-1, 4).astype(float)
            d = d[d[:, 0] > 0]
            tot = d[:, :3].sum(axis=1).mean()
            print(f"  stamps (mean over {len(d)} wavefronts, last launch; 100 MHz ticks): load+keys+minmax {d[:,0].mean():.0f} "
                  f"({100*d[:,0].mean()/tot:.0f}%)  passes {d[:,1].mean():.0f} ({100*d[:,1].mean()/tot:.0f}%)  fetch+upper {d[:,2].mean():.0f} "
                  f"({100*d[:,2].mean()/tot:.0f}%)  passes per column (last launch) {d[:,3].sum() / n:.2f}")
    if a.ablate == 4 or a.ablate in (2, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16):
        waves = info["waves"]
        nwg = min(n, 256)
        d = dbg.cpu().numpy()[: nwg * waves * 4].reshape(nwg, waves, 4).astype(float)
        tot = d[:, :, 3].mean()
        print(f"  stamps (mean over {nwg} WGs x {waves} waves, cycles per launch): stage {d[:,:,0].mean():.0f} "
              f"({100*d[:,:,0].mean()/tot:.1f}%) gather {d[:,:,1].mean():.0f} ({100*d[:,:,1].mean()/tot:.1f}%) "
              f"end-barrier wait {d[:,:,2].mean():.0f} ({100*d[:,:,2].mean()/tot:.1f}%) total {tot:.0f}")
        print("  per-wave gather cycles, WG 0:", d[0, :, 1].astype(int).tolist())
        print("  per-wave wait   cycles, WG 0:", d[0, :, 2].astype(int).tolist())
    if a.kernel == "spmm":
        z = int(Gp[-1])
        b = g * n * 8 + 4 * z + 4 * (m + 1) + m * n * 8
        print(f"  algorithmic {b / 1e9:.3f} GB -> {b / min(ms) / 1e6:.1f} GB/s; "
              f"wave-gathers/column {info['padded_slots'] // 64} -> "
              f"{min(ms) * 1e-3 * 2.4e9 / (n / 256) / max(info['padded_slots'] / 64, 1):.2f} cycles per wave-gather per CU @2.4GHz")


if __name__ == "__main__":
    main()
