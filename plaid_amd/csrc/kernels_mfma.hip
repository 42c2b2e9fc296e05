// The dense "rank-weight GEMM" form of the crossprod on the matrix cores (BASELINE config 4, opt-in):
//     S = G^T W,   G dense 0/1 (exact in bf16),  W = the expression / rank-weight panel.
// The reference never runs a dense contraction (replaid.ssgsea multiplies the SPARSE membership too, R/plaid.R:253 ->
// :80 -> Matrix::crossprod at :107), and with |set| ~ 140 of 20,000 genes G is 0.7 % dense: this backend does ~145x
// the multiply-adds of the SpMM kernels (x3 again for the precision split below).  It exists as the alternate backend
// config 4 names, parity-checked against the SpMM route, so that its rate can be put beside the SpMM time
// (bench.py c4 block): it loses by the factor the arithmetic says.
//
//   * W (fp64) is split into three bf16 terms, W = hi + mid + lo (24 significant bits, ~6e-8 relative: inside the
//     1e-5 bar); three MFMA products accumulate into the same fp32 tile.  Sums of <= 500 products of magnitude <= 1
//     keep ~1e-7 relative in fp32.
//   * tile: 128 samples x 128 sets per 256-thread workgroup, K step 64; four wavefronts in 2 x 2, each 64 x 64 =
//     2 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16.  Samples are the MFMA rows and sets the columns, so that a
//     lane of the accumulator tile is a set: 32 lanes store 32 neighbouring rows of S (column-major sets x samples).
//   * operands staged through LDS with rows padded to 144 bytes (the four 16-lane groups of ds_read_b128 then hit
//     16 distinct 16-byte slots: conflict-free), next K step prefetched into registers during the MFMAs.
#include <mutex>

#include "common.h"

namespace plaidhip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMfmaTile = 128;   // samples and sets per workgroup tile
constexpr int kMfmaBK = 64;      // K (genes) per step
constexpr int kMfmaRowB = 144;   // bytes per LDS row: 64 bf16 + 16 bytes of padding

// X (fp64, column-major genes x samples: a sample's genes are contiguous) -> three bf16 matrices [rows_pad][gk]
__global__ void __launch_bounds__(256)
split3_bf16_kernel(const double* __restrict__ X, int64_t ldx, int32_t g, int32_t gk, int32_t ncols, int32_t rows_pad,
                   __bf16* __restrict__ W3) {
  const int64_t plane = (int64_t)rows_pad * gk;
  for (int c = blockIdx.y; c < rows_pad; c += gridDim.y) {
    const double* xc = X + (int64_t)c * ldx;
    __bf16* w = W3 + (int64_t)c * gk;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < gk; i += gridDim.x * blockDim.x) {
      double x = (c < ncols && i < g) ? xc[i] : 0.0;
      const __bf16 hi = (__bf16)(float)x;
      x -= (double)(float)hi;
      const __bf16 mid = (__bf16)(float)x;
      x -= (double)(float)mid;
      const __bf16 lo = (__bf16)(float)x;
      w[i] = hi;
      w[plane + i] = mid;
      w[2 * plane + i] = lo;
    }
  }
}

__global__ void __launch_bounds__(256)
densify_sets_kernel(const int32_t* __restrict__ Gp, const int32_t* __restrict__ Gi, int32_t m, int32_t gk,
                    __bf16* __restrict__ Gd) {
  for (int j = blockIdx.x; j < m; j += gridDim.x) {
    __bf16* row = Gd + (int64_t)j * gk;
    for (int p = Gp[j] + threadIdx.x; p < Gp[j + 1]; p += blockDim.x) row[Gi[p]] = (__bf16)1.0f;
  }
}

struct MfmaArgs {
  const __bf16* W3;     // three planes [rows_pad][gk]
  const __bf16* Gd;     // [mpad][gk]
  int32_t gk, rows_pad, ncols, m;
  const double* w;      // per set 1/(1e-8 + size)
  const double* k;      // per set size
  int32_t stat;
  double alpha, beta;
  const double* alpha_div;
  double* S;            // first column of the panel
  int64_t lds;
  uint32_t* flags;
};

__device__ __forceinline__ void publish_flags_mfma(uint32_t f, uint32_t* flags) {
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

__global__ void __launch_bounds__(256, 2)
crossprod_mfma_bf16x3_kernel(MfmaArgs a) {
  __shared__ __align__(16) unsigned char lds[4 * kMfmaTile * kMfmaRowB];   // W hi / mid / lo, G : 72 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * kMfmaTile;   // first set of the tile
  const int m0 = blockIdx.y * kMfmaTile;   // first sample of the tile
  const int64_t plane = (int64_t)a.rows_pad * a.gk;
  // staging: 4 tiles x 128 rows x 8 pieces of 16 bytes = 4096 pieces, 16 per thread; piece q of a tile: row q / 8
  const uint4* src[4];
  uint32_t dst[4];
  {
    const int row = tid >> 3, c8 = tid & 7;   // piece q = tid + 256 i -> row + 32 i
    for (int t = 0; t < 3; ++t)
      src[t] = reinterpret_cast<const uint4*>(a.W3 + t * plane + (int64_t)(m0 + row) * a.gk) + c8;
    src[3] = reinterpret_cast<const uint4*>(a.Gd + (int64_t)(n0 + row) * a.gk) + c8;
    for (int t = 0; t < 4; ++t) dst[t] = (uint32_t)(t * kMfmaTile * kMfmaRowB + row * kMfmaRowB + c8 * 16);
  }
  const int64_t rstep = (int64_t)32 * a.gk * 2 / 16;   // 32 rows further, in uint4 units
  uint4 pf[4][4];
#define PH_MFMA_PREFETCH(kstep)                                                       \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) pf[t][i] = src[t][(int64_t)i * rstep + (int64_t)(kstep) * (kMfmaBK * 2 / 16)];
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int nk = a.gk / kMfmaBK;
  PH_MFMA_PREFETCH(0)
  const uint32_t arow = (uint32_t)((wm * 64 + (lane & 31)) * kMfmaRowB + (lane >> 5) * 16);
  const uint32_t brow = (uint32_t)(3 * kMfmaTile * kMfmaRowB + (wn * 64 + (lane & 31)) * kMfmaRowB + (lane >> 5) * 16);
  for (int ks = 0; ks < nk; ++ks) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(lds + dst[t] + i * 32 * kMfmaRowB) = pf[t][i];
    __syncthreads();
    if (ks + 1 < nk) { PH_MFMA_PREFETCH(ks + 1) }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(lds + brow + j * 32 * kMfmaRowB + kk * 32);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 afr = *reinterpret_cast<const bf16x8*>(lds + t * kMfmaTile * kMfmaRowB + arow + i * 32 * kMfmaRowB + kk * 32);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#undef PH_MFMA_PREFETCH
  // epilogue: alpha * (sum * w) + beta * (k * w), the same as the SpMM kernels'
  const double alpha = (a.alpha_div != nullptr) ? a.alpha / *a.alpha_div : a.alpha;
  uint32_t f = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int set = n0 + wn * 64 + j * 32 + (lane & 31);
    const bool sok = set < a.m;
    const double kj = sok ? a.k[set] : 0.0;
    const double wj = sok ? (a.stat == PLAIDHIP_STAT_MEAN ? a.w[set] : 1.0) : 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int sample = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (sok && sample < a.ncols) {
          const double v = alpha * ((double)acc[i][j][r] * wj) + a.beta * (kj * wj);
          a.S[(int64_t)sample * a.lds + set] = v;
          f |= (v < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;
          f |= (v == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;
          f |= (v != v) ? PLAIDHIP_FLAG_HAS_NAN : 0u;
        }
      }
  }
  publish_flags_mfma(f, a.flags);
}

// dense 0/1 membership, sets x genes, bf16, built on first use from the pattern kept with the gene-set collection
static int ensure_dense_g(plaidhip_ctx* ctx, plaidhip_geneset* gs) {
  // the lazily built matrix is the one mutable part of a prepared collection: built into local pointers, published
  // (d_dense_g, dense_gk) only once every step has succeeded, under a lock
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (gs->d_dense_g != nullptr) return PLAIDHIP_OK;
  const int32_t gk = (gs->g + kMfmaBK - 1) / kMfmaBK * kMfmaBK;
  const int32_t mpad = (gs->m + kMfmaTile - 1) / kMfmaTile * kMfmaTile;
  const size_t bytes = (size_t)mpad * gk * 2;
  struct Tmp {   // freed on every exit path
    void* p = nullptr;
    ~Tmp() { if (p) hipFree(p); }
  } dG, dGp, dGi;
  PH_HIP(hipMalloc(&dG.p, bytes));
  PH_HIP(hipMemsetAsync(dG.p, 0, bytes, ctx->stream));
  PH_HIP(hipMalloc(&dGp.p, (size_t)(gs->m + 1) * 4));
  PH_HIP(hipMalloc(&dGi.p, std::max<size_t>(gs->h_Gi.size(), 1) * 4));
  PH_HIP(hipMemcpyAsync(dGp.p, gs->h_Gp.data(), (size_t)(gs->m + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!gs->h_Gi.empty())
    PH_HIP(hipMemcpyAsync(dGi.p, gs->h_Gi.data(), gs->h_Gi.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(densify_sets_kernel, dim3(gs->m < 65535 ? gs->m : 65535), dim3(256), 0, ctx->stream,
                     static_cast<const int32_t*>(dGp.p), static_cast<const int32_t*>(dGi.p), gs->m, gk,
                     reinterpret_cast<__bf16*>(dG.p));
  PH_HIP(hipGetLastError());
  PH_HIP(hipStreamSynchronize(ctx->stream));
  gs->dense_gk = gk;
  gs->d_dense_g = dG.p;
  dG.p = nullptr;
  return PLAIDHIP_OK;
}

int launch_spmm_mfma_f64(plaidhip_ctx* ctx, plaidhip_geneset* gs, const double* X, int64_t ldx, int32_t n, int stat,
                         double alpha, const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags) {
  if (n == 0 || gs->m == 0) return PLAIDHIP_OK;
  int rc = ensure_dense_g(ctx, gs);
  if (rc != PLAIDHIP_OK) return rc;
  const int32_t gk = gs->dense_gk;
  // sample panels of at most 8,192 columns: the split operand costs 6 bytes per value
  const int32_t panel = 8192;
  const int32_t prow = ((n < panel ? n : panel) + kMfmaTile - 1) / kMfmaTile * kMfmaTile;
  rc = ensure_workspace(ctx, (size_t)3 * prow * gk * 2);
  if (rc != PLAIDHIP_OK) return rc;
  __bf16* W3 = reinterpret_cast<__bf16*>(ctx->ws);
  for (int32_t c0 = 0; c0 < n; c0 += panel) {
    const int32_t nc = (n - c0) < panel ? (n - c0) : panel;
    const int32_t rows_pad = (nc + kMfmaTile - 1) / kMfmaTile * kMfmaTile;
    hipLaunchKernelGGL(split3_bf16_kernel, dim3((gk + 255) / 256, rows_pad < 4096 ? rows_pad : 4096), dim3(256), 0, ctx->stream,
                       X + (int64_t)c0 * ldx, ldx, gs->g, gk, nc, rows_pad, W3);
    MfmaArgs a{};
    a.W3 = W3;
    a.Gd = reinterpret_cast<const __bf16*>(gs->d_dense_g);
    a.gk = gk;
    a.rows_pad = rows_pad;
    a.ncols = nc;
    a.m = gs->m;
    a.w = gs->scatter.d_w;
    a.k = gs->scatter.d_k;
    a.stat = stat;
    a.alpha = alpha;
    a.beta = beta;
    a.alpha_div = alpha_div;
    a.S = S + (int64_t)c0 * lds;
    a.lds = lds;
    a.flags = flags;
    const int mt = (gs->m + kMfmaTile - 1) / kMfmaTile;
    hipLaunchKernelGGL(crossprod_mfma_bf16x3_kernel, dim3(mt, rows_pad / kMfmaTile), dim3(256), 0, ctx->stream, a);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

}  // namespace plaidhip
