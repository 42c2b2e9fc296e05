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
//   * tile (round 5): 256 samples x 256 sets per 512-thread workgroup, one per CU; eight wavefronts in 2 x 4, each
//     128 samples x 64 sets = 4 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator registers).  Samples are
//     the MFMA rows and sets the columns, so that a lane of the accumulator tile is a set: 32 lanes store 32
//     neighbouring rows of S (column-major sets x samples).  The three planes share the G fragments: per 16 genes a
//     wavefront reads 2 G + 12 W fragments for 24 MFMAs.
//   * K step 32 genes, TWO LDS stages (2 x 4 tiles x 256 rows x 80 bytes = the whole 160 KB): rows padded from 64 to
//     80 bytes, so that the four 16-lane groups of a ds_read_b128 hit 16 distinct 16-byte slots (80 = 5 x 16, 5 is
//     odd: conflict-free); step k + 1 is written to the other stage and step k + 2 requested from memory before the
//     MFMAs of step k issue: one barrier per step; the W fragments of plane t + 1 are read while the MFMAs of plane t
//     issue.  0.57-0.59 of the bf16 peak.  (Rounds 2-4: 128 x 128 tiles, 256 threads, one stage, two barriers per step:
//     0.10.)
#include <mutex>

#include "common.h"

namespace plaidhip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMfmaTile = 256;   // samples and sets per workgroup tile
constexpr int kMfmaBK = 32;      // K (genes) per step
constexpr int kMfmaRowB = 80;    // bytes per LDS row: 32 bf16 + 16 bytes of padding
constexpr int kMfmaGkPad = 64;   // the gene dimension of the bf16 operands is padded to a multiple of this
constexpr int kMfmaStageB = 4 * kMfmaTile * kMfmaRowB;   // W hi / mid / lo, G: 81,920 bytes per stage

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

__global__ void __launch_bounds__(512)
crossprod_mfma_bf16x3_kernel(MfmaArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];   // two stages of {W hi, W mid, W lo, G} tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;    // 2 x 4 wavefronts: 128 samples x 64 sets each
  const int n0 = blockIdx.x * kMfmaTile;   // first set of the tile
  const int m0 = blockIdx.y * kMfmaTile;   // first sample of the tile
  const int64_t plane = (int64_t)a.rows_pad * a.gk;
  // staging: 4 tiles x 256 rows x 4 pieces of 16 bytes = 4,096 pieces per step, 8 per thread: tile t = 0 .. 3, piece
  // q = tid + 512 i (i = 0, 1): row q / 4, 16-byte column q % 4.  (Scalars, not arrays: under this kernel's register
  // pressure hipcc leaves a private array in scratch memory -- the first build staged every step through it.)
  const int srow = tid >> 2, sc4 = tid & 3;   // i adds 128 rows
  const uint4* src0 = reinterpret_cast<const uint4*>(a.W3 + (int64_t)(m0 + srow) * a.gk) + sc4;
  const uint4* src1 = reinterpret_cast<const uint4*>(a.W3 + plane + (int64_t)(m0 + srow) * a.gk) + sc4;
  const uint4* src2 = reinterpret_cast<const uint4*>(a.W3 + 2 * plane + (int64_t)(m0 + srow) * a.gk) + sc4;
  const uint4* src3 = reinterpret_cast<const uint4*>(a.Gd + (int64_t)(n0 + srow) * a.gk) + sc4;
  const uint32_t dst0 = (uint32_t)(srow * kMfmaRowB + sc4 * 16);
  const int64_t rstep = (int64_t)128 * a.gk * 2 / 16;   // 128 rows further, in uint4 units
  uint4 pf00, pf01, pf10, pf11, pf20, pf21, pf30, pf31;
#define PH_MFMA_PREFETCH(kstep)                                        \
  {                                                                    \
    const int64_t ko_ = (int64_t)(kstep) * (kMfmaBK * 2 / 16);         \
    pf00 = src0[ko_]; pf01 = src0[rstep + ko_];                        \
    pf10 = src1[ko_]; pf11 = src1[rstep + ko_];                        \
    pf20 = src2[ko_]; pf21 = src2[rstep + ko_];                        \
    pf30 = src3[ko_]; pf31 = src3[rstep + ko_];                        \
  }
#define PH_MFMA_ST1(stage, t, i, v) \
  *reinterpret_cast<uint4*>(lds + (stage) * kMfmaStageB + (t) * kMfmaTile * kMfmaRowB + dst0 + (i) * 128 * kMfmaRowB) = v;
#define PH_MFMA_STAGE(stage)                                                                      \
  PH_MFMA_ST1(stage, 0, 0, pf00) PH_MFMA_ST1(stage, 0, 1, pf01) PH_MFMA_ST1(stage, 1, 0, pf10) PH_MFMA_ST1(stage, 1, 1, pf11) \
  PH_MFMA_ST1(stage, 2, 0, pf20) PH_MFMA_ST1(stage, 2, 1, pf21) PH_MFMA_ST1(stage, 3, 0, pf30) PH_MFMA_ST1(stage, 3, 1, pf31)
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int nk = a.gk / kMfmaBK;
  PH_MFMA_PREFETCH(0)
  PH_MFMA_STAGE(0)
  if (nk > 1) { PH_MFMA_PREFETCH(1) }
  __syncthreads();
  const uint32_t arow = (uint32_t)((wm * 128 + (lane & 31)) * kMfmaRowB + (lane >> 5) * 16);
  const uint32_t brow = (uint32_t)(3 * kMfmaTile * kMfmaRowB + (wn * 64 + (lane & 31)) * kMfmaRowB + (lane >> 5) * 16);
  // One K step = two halves of 16 genes, 24 MFMAs each.  The eight LDS writes that stage step ks + 1 are SPREAD over the first
  // half (one per three MFMAs) instead of leading the step: issued together they hold the LDS pipe for ~830 cycles while the
  // fragment reads of all eight wavefronts queue behind them (profiles/r05j: 0.52 of the peak in that form); the requests for
  // step ks + 2 follow between the halves, when the prefetch registers are free again.
// A half step: the W fragments of plane t + 1 are requested BEFORE the eight MFMAs of plane t issue (two register sets,
// scheduling barriers between the groups): left to itself hipcc keeps two fragment reads in flight and every MFMA pair
// waits for an LDS round trip.
#define PH_MFMA_AFR(dst, t_, kk)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                    \
    dst[i] = *reinterpret_cast<const bf16x8*>(st + (t_) * kMfmaTile * kMfmaRowB + arow + i * 32 * kMfmaRowB + (kk) * 32);
#define PH_MFMA_PLANE(afr_, WA, WB, WC, WD)                                                                        \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[0], bfr0, acc[0][0], 0, 0, 0);                          \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[0], bfr1, acc[0][1], 0, 0, 0); WA                       \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[1], bfr0, acc[1][0], 0, 0, 0);                          \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[1], bfr1, acc[1][1], 0, 0, 0); WB                       \
  acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[2], bfr0, acc[2][0], 0, 0, 0);                          \
  acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[2], bfr1, acc[2][1], 0, 0, 0); WC                       \
  acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[3], bfr0, acc[3][0], 0, 0, 0);                          \
  acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr_[3], bfr1, acc[3][1], 0, 0, 0); WD
#define PH_MFMA_HALF(kk, W0, W1, W2, W3, W4, W5, W6, W7)                                                        \
  {                                                                                                               \
    const bf16x8 bfr0 = *reinterpret_cast<const bf16x8*>(st + brow + (kk) * 32);                                  \
    const bf16x8 bfr1 = *reinterpret_cast<const bf16x8*>(st + brow + 32 * kMfmaRowB + (kk) * 32);                 \
    bf16x8 afa[4], afb[4];                                                                                        \
    PH_MFMA_AFR(afa, 0, kk)                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    PH_MFMA_AFR(afb, 1, kk)                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    PH_MFMA_PLANE(afa, W0, W1, W2, W3)                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    PH_MFMA_AFR(afa, 2, kk)                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    PH_MFMA_PLANE(afb, W4, W5, W6, W7)                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    PH_MFMA_PLANE(afa, , , , )                                                                                    \
  }
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1, nx = cur ^ 1;
    const unsigned char* st = lds + cur * kMfmaStageB;
    // step ks + 1 (in registers since the step before) goes to the other stage: its readers of step ks - 1 are behind the
    // barrier that ended that step.  Unconditional on purpose (no second copy of the MFMA sequence, no branches inside it): the
    // last step stages stale registers into a stage nobody reads again and re-requests its own slab.
    PH_MFMA_HALF(0, PH_MFMA_ST1(nx, 0, 0, pf00), PH_MFMA_ST1(nx, 0, 1, pf01), PH_MFMA_ST1(nx, 1, 0, pf10), PH_MFMA_ST1(nx, 1, 1, pf11),
                 PH_MFMA_ST1(nx, 2, 0, pf20), PH_MFMA_ST1(nx, 2, 1, pf21), PH_MFMA_ST1(nx, 3, 0, pf30), PH_MFMA_ST1(nx, 3, 1, pf31))
    __builtin_amdgcn_sched_barrier(0);   // (keeps the second half's fragment reads behind the first half: registers)
    PH_MFMA_PREFETCH(ks + 2 < nk ? ks + 2 : nk - 1)
    PH_MFMA_HALF(1, , , , , , , , )
    __syncthreads();
  }
#undef PH_MFMA_HALF
#undef PH_MFMA_PLANE
#undef PH_MFMA_AFR
#undef PH_MFMA_PREFETCH
#undef PH_MFMA_STAGE
#undef PH_MFMA_ST1
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
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int sample = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
  const int32_t gk = (gs->g + kMfmaGkPad - 1) / kMfmaGkPad * kMfmaGkPad;
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
    PH_FULL_LDS(ctx, (&crossprod_mfma_bf16x3_kernel));
    hipLaunchKernelGGL(crossprod_mfma_bf16x3_kernel, dim3(mt, rows_pad / kMfmaTile), dim3(512), 2 * kMfmaStageB, ctx->stream, a);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

}  // namespace plaidhip
