// S = G^T X : sparse-binary gene-set membership x expression crossprod
// (replaces Matrix::crossprod at R/plaid.R:107 with the 1/|set| column scaling of
//  R/plaid.R:74-77 folded into the epilogue).  gfx950 / wave64 only.
//
// Kernel shape ("column-resident gather"):
//   * one workgroup owns one sample column at a time.  R's layout is column-major, so a
//     sample column is one contiguous, perfectly coalesced HBM read (g * 8 B).
//   * the column lives in LDS as 8-byte entries (g <= 20448 fits the CU's 160 KiB);
//     32 trailing zero entries absorb padded index slots, so the inner loop is branch-free.
//   * lanes = gene sets.  A wavefront walks one "tile" of 64 sets (pre-sorted by size, so
//     lanes finish together); each lane streams its set's u16 gene ids (8 per 16-byte
//     load, 1 KiB per wave, L2-resident -- the lists are shared by every column) and
//     gathers the gene's value from LDS (ds_read_b64), accumulating in fp64.
//   * epilogue per set: alpha * (sum * w) + beta * (k * w), plus min()==0 bookkeeping for
//     normalize_medians (R/plaid.R:556-557).
// Algorithmic HBM bytes per column: 8 g (X) + 8 m (S); the index lists (2 B per
// membership) are read from L2, once per column.
#include "common.h"

namespace plaidhip {

struct SpmmArgs {
  const double* X;
  int64_t ldx;
  const int32_t* Xp;
  const int32_t* Xi;
  const double* Xx;
  int32_t g, n, m, tiles;
  const uint4* tile_idx;
  const int32_t* tile_chunk_off;
  const int32_t* lane_set;
  const int32_t* set_size;
  int32_t stat;
  double alpha, beta;
  const double* alpha_div;  // device scalar: alpha /= *alpha_div (global max(rX)), may be null
  double* S;
  int64_t lds;
  uint32_t* flags;
};

__device__ __forceinline__ void publish_flags(uint32_t f, uint32_t* flags) {
  // flags[0..2] = has_neg / has_zero / has_nan as 0/1 words (element-wise MAX all-reducible).
  // wave-level OR, then plain idempotent stores of 1 (test first: the words saturate early).
  for (int off = 32; off >= 1; off >>= 1) f |= __shfl_xor(f, off, 64);
  if (flags != nullptr && (threadIdx.x & 63) == 0) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if ((f >> b) & 1u) {
        if (__hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
          __hip_atomic_store(&flags[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
}

template <bool CSC_X>
__global__ void __launch_bounds__(1024)
spmm_colgather_f64(SpmmArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* col = reinterpret_cast<double*>(smem_raw);

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  uint32_t f = 0;
  const double alpha = (a.alpha_div != nullptr) ? a.alpha / *a.alpha_div : a.alpha;

  for (int c = blockIdx.x; c < a.n; c += gridDim.x) {
    // ---- stage the sample column in LDS ------------------------------------------
    if constexpr (!CSC_X) {
      const double* xc = a.X + (int64_t)c * a.ldx;
      if ((((uintptr_t)xc) & 15) == 0) {
        const double2* xc2 = reinterpret_cast<const double2*>(xc);
        double2* col2 = reinterpret_cast<double2*>(col);
        const int g2 = a.g >> 1;
        for (int i = tid; i < g2; i += nthr) col2[i] = xc2[i];
        if ((a.g & 1) && tid == 0) col[a.g - 1] = xc[a.g - 1];
      } else {
        for (int i = tid; i < a.g; i += nthr) col[i] = xc[i];
      }
      if (tid < kPadSlots) col[a.g + tid] = 0.0;
    } else {
      for (int i = tid; i < a.g + kPadSlots; i += nthr) col[i] = 0.0;
      __syncthreads();
      const int p0 = a.Xp[c], p1 = a.Xp[c + 1];
      for (int p = p0 + tid; p < p1; p += nthr) col[a.Xi[p]] = a.Xx[p];
    }
    __syncthreads();

    // ---- gather: one tile of 64 sets per wave -------------------------------------
    for (int t = wave; t < a.tiles; t += nwaves) {
      const int c0 = a.tile_chunk_off[t], c1 = a.tile_chunk_off[t + 1];
      const uint4* ip = a.tile_idx + (int64_t)c0 * 64 + lane;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int ch = c0; ch < c1; ++ch, ip += 64) {
        const uint4 q = *ip;
        s0 += col[q.x & 0xffffu];
        s1 += col[q.x >> 16];
        s2 += col[q.y & 0xffffu];
        s3 += col[q.y >> 16];
        s0 += col[q.z & 0xffffu];
        s1 += col[q.z >> 16];
        s2 += col[q.w & 0xffffu];
        s3 += col[q.w >> 16];
      }
      const double sum = (s0 + s1) + (s2 + s3);
      const int j = a.lane_set[t * 64 + lane];
      if (j >= 0) {
        const double k = (double)a.set_size[j];
        const double w = (a.stat == PLAIDHIP_STAT_MEAN) ? 1.0 / (1e-8 + k) : 1.0;
        const double v = alpha * (sum * w) + a.beta * (k * w);
        a.S[(int64_t)c * a.lds + j] = v;
        f |= (v < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;
        f |= (v == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;
        f |= (v != v) ? PLAIDHIP_FLAG_HAS_NAN : 0u;
      }
    }
    __syncthreads();  // column is overwritten by the next iteration
  }
  publish_flags(f, a.flags);
}

// Fallback for g beyond the LDS-resident limit: one thread per (set, column), gene values
// gathered straight from global memory (the column is L2-resident).  Correctness path.
__global__ void __launch_bounds__(256)
spmm_global_f64(const double* X, int64_t ldx, int32_t n, int32_t m, const int32_t* Gp,
                const int32_t* Gi, int32_t stat, double alpha, const double* alpha_div, double beta,
                double* S, int64_t lds, uint32_t* flags) {
  uint32_t f = 0;
  if (alpha_div != nullptr) alpha /= *alpha_div;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (j < m) {
    const double* xc = X + (int64_t)c * ldx;
    const int p0 = Gp[j], p1 = Gp[j + 1];
    double s = 0.0;
    for (int p = p0; p < p1; ++p) s += xc[Gi[p]];
    const double k = (double)(p1 - p0);
    const double w = (stat == PLAIDHIP_STAT_MEAN) ? 1.0 / (1e-8 + k) : 1.0;
    const double v = alpha * (s * w) + beta * (k * w);
    S[(int64_t)c * lds + j] = v;
    f |= (v < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;
    f |= (v == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;
    f |= (v != v) ? PLAIDHIP_FLAG_HAS_NAN : 0u;
  }
  publish_flags(f, flags);
}

static int block_for_genes(int32_t g) { return g > 8192 ? 1024 : (g > 2048 ? 512 : 256); }

template <bool CSC_X>
static int launch_colgather(plaidhip_ctx* ctx, const plaidhip_geneset* gs, SpmmArgs& a) {
  const size_t smem = (size_t)(gs->g + kPadSlots) * sizeof(double);
  static bool attr_set[2] = {false, false};
  if (!attr_set[CSC_X]) {
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_colgather_f64<CSC_X>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    attr_set[CSC_X] = true;
  }
  const int block = block_for_genes(gs->g);
  const int grid = a.n;
  hipLaunchKernelGGL(spmm_colgather_f64<CSC_X>, dim3(grid), dim3(block), smem, ctx->stream, a);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

static void fill_args(SpmmArgs& a, const plaidhip_geneset* gs, int32_t n, int stat, double alpha,
                      const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags) {
  a.alpha_div = alpha_div;
  a.g = gs->g;
  a.n = n;
  a.m = gs->m;
  a.tiles = gs->tiles;
  a.tile_idx = reinterpret_cast<const uint4*>(gs->d_tile_idx);
  a.tile_chunk_off = gs->d_tile_chunk_off;
  a.lane_set = gs->d_lane_set;
  a.set_size = gs->d_set_size;
  a.stat = stat;
  a.alpha = alpha;
  a.beta = beta;
  a.S = S;
  a.lds = lds;
  a.flags = flags;
}

int launch_spmm_dense_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const double* X,
                          int64_t ldx, int32_t n, int stat, double alpha, const double* alpha_div,
                          double beta, double* S, int64_t lds, uint32_t* flags) {
  if (n == 0 || gs->m == 0) return PLAIDHIP_OK;
  if (gs->lds_ok) {
    SpmmArgs a{};
    a.X = X;
    a.ldx = ldx;
    fill_args(a, gs, n, stat, alpha, alpha_div, beta, S, lds, flags);
    return launch_colgather<false>(ctx, gs, a);
  }
  dim3 grid((gs->m + 255) / 256, n);
  hipLaunchKernelGGL(spmm_global_f64, grid, dim3(256), 0, ctx->stream, X, ldx, n, gs->m, gs->d_Gp,
                     gs->d_Gi, stat, alpha, alpha_div, beta, S, lds, flags);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_spmm_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp,
                        const int32_t* Xi, const double* Xx, int32_t n, int stat, double alpha,
                        const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags) {
  if (n == 0 || gs->m == 0) return PLAIDHIP_OK;
  if (!gs->lds_ok) {
    set_error("spmm_csc: g=%d exceeds the LDS-resident limit %d (sparse-X large-g path not built yet)",
              gs->g, kMaxLdsGenes);
    return PLAIDHIP_EUNSUPPORTED;
  }
  SpmmArgs a{};
  a.Xp = Xp;
  a.Xi = Xi;
  a.Xx = Xx;
  fill_args(a, gs, n, stat, alpha, alpha_div, beta, S, lds, flags);
  return launch_colgather<true>(ctx, gs, a);
}

}  // namespace plaidhip
