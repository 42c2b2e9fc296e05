#!/usr/bin/python3
"""profiles/traffic.json from the PMC summaries of one profile round (tools/profile_round.sh writes pmc_<tag>_summary.txt:
per kernel the mean FETCH_SIZE / WRITE_SIZE (KB) over its launches, collected in SEPARATE --pmc passes).  Correction as
MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE x 2 for the 16-byte-per-lane / streaming loads these kernels issue
(the counter tallies their 128-byte requests at 64 bytes), WRITE_SIZE as is.  Every entry is re-measured by the round, so no
entry can cite a kernel that was rewritten since.
    python tools/update_traffic.py gpurun_out/r04x profiles/r04x"""
import json
import os
import re
import sys

src_dir, prof_prefix = sys.argv[1], sys.argv[2]
CORR = ("FETCH_SIZE x2 (gfx950 tallies the 128-B requests of 16-B/lane streaming loads at 64 B, MI355X_MICROARCH.md HBM "
        "section); WRITE_SIZE as is; separate --pmc passes (tools/profile_round.sh)")


def summary(tag):
    path = os.path.join(src_dir, f"pmc_{tag}_summary.txt")
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+mean=([\d.e+-]+)", line)
            if m:
                out[cur][m.group(1)] = float(m.group(3))
    return out


def find(summ, needle):
    hits = [k for k in summ if needle in k]
    if not hits:
        raise SystemExit(f"no kernel matching {needle!r}")
    return summ[hits[0]]


# (traffic.json key, pmc tag, kernel-name needle, columns of the measured launch or None for per-launch, note)
ENTRIES = [
    ("spmm_colpair_f64/20000x10000x5000", "c2", "spmm_colpair_f64<false, 0, false, false, false>", None,
     "fabric-side bytes: 1.6 GB X + 0.4 GB S + the partial sums of the two gene slices (written and re-read ~30 us later; served "
     "by L2 / Infinity Cache, which these counters do not separate from HBM)"),
    ("col_medians_wave_kernel/10000x5000", "c2step", "col_medians_wave_kernel", None, "every column read once into registers"),
    ("shift_columns_kernel/10000x5000", "c2step", "shift_columns_kernel", None, "read + write of S"),
    ("spmm_scatter_csc_f64/20000xNx50000", "c3", "spmm_scatter_csc_f64<true, 1024, false>", 8192,
     "round-6 kernel (sets dealt to the chunks in interleaved blocks, second segments in the static pipeline, u64 fixed-point "
     "accumulators, chunk-major item order): fetched + written against 0.41 MB algorithmic per column"),
    ("spmm_scatter_csc_f64<med>/20000xNx50000", "c3fused", "spmm_scatter_csc_f64<true, 1024, true>", 8192,
     "the same kernel with the classifying epilogue (medians selected inside the launch): + the candidate slices and counts"),
    ("col_medians_stream_kernel/Nx50000", "c3", "col_medians_stream_kernel", 8192,
     "one sweep of a 400 KB column + the candidate list of the sample interval written and read back"),
    ("shift_columns_kernel/Nx50000", "c3", "shift_columns_kernel", 8192, "read + write of S: the algorithmic 16 m bytes per column"),
    ("colranks_bucket_kernel<256,8>/csc", "c3", "colranks_bucket_kernel<256, 8>", 8192, "1000 stored values per column: 8 KB read + 8 KB written + colmax"),
    ("colranks_bucket_kernel<1024,20>/20000xN", "c4", "colranks_bucket_kernel<1024, 20>", 4096,
     "160 KB read + 160 KB written per column = the algorithmic 16 g bytes (+ the spilled registers' scratch)"),
    ("spmm_colpair_f64/20000xNx50000", "c4", "spmm_colpair_f64<false, 0, false, false, true>", 4096,
     "0.8 MB of partial sums per column pair written and re-read between the two gene slices + tile streams missing L2"),
    ("spmm_colpair_f64<med>/20000xNx50000", "c4fused", "spmm_colpair_f64<false, 0, false, true, true>", (4096 + 256) // 2,
     "the pair kernel with the classifying tile ends (medians selected inside the launch): partial sums as above + the candidate "
     "slices and counts.  Two launches per call -- the calibration on 256 columns and the main one on 4,096 -- whose MEAN the "
     "summary holds: per column = mean / ((4096 + 256) / 2)"),
    ("spmm_colquad_u16/20000xNx50000", "sing", "spmm_colquad_u16", 4096, "rank crossprod, u16 staging (one gene slice, no partial sums)"),
]
out = {}
for key, tag, needle, cols, note in ENTRIES:
    try:
        c = find(summary(tag), needle)
    except (FileNotFoundError, SystemExit) as exc:
        print(f"skip {key}: {exc}", file=sys.stderr)
        continue
    total = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
    e = {"fetch_size_kb_raw": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "correction": CORR, "note": note,
         "source": f"{prof_prefix}_pmc_{tag}_summary.txt"}
    if cols is None:
        e["hbm_bytes_per_launch"] = int(total)
    else:
        e["hbm_bytes_per_column"] = round(total / cols, 1)
        e["measured_columns"] = cols
    for extra in ("TCC_HIT_sum", "TCC_MISS_sum"):
        if extra in c:
            e[extra.lower()] = c[extra]
    # the clock the chip held in the GRBM pass and the share of the launch the LDS was active (256 CUs, 8 XCDs: the counters
    # are sums over them) -- what says "on the LDS roof" next to lds_frac
    if "GRBM_GUI_ACTIVE" in c and "DURATION_NS_GRBM_PASS" in c and c["DURATION_NS_GRBM_PASS"] > 0:
        e["clock_ghz"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / c["DURATION_NS_GRBM_PASS"], 3)
        if "SQ_LDS_IDX_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            e["lds_active"] = round((c["SQ_LDS_IDX_ACTIVE"] / 256.0) / (c["GRBM_GUI_ACTIVE"] / 8.0), 3)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            e["lds_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
        if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
            e["wait_share"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    out[key] = e
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch", v.get("hbm_bytes_per_column")) for k, v in out.items()}, indent=1))
