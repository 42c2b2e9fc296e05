// S = G^T X : sparse-binary gene-set membership x expression crossprod
// (replaces Matrix::crossprod at R/plaid.R:107 with the 1/|set| column scaling of
//  R/plaid.R:74-77 folded into the epilogue).  gfx950 / wave64 only.
//
// Kernel shape ("column-resident gather", persistent):
//   * a workgroup owns one sample column at a time and walks columns c, c+grid, ...  R's
//     layout is column-major, so a column is one contiguous, perfectly coalesced HBM read
//     (8 g bytes).  The NEXT column is prefetched into registers (16-byte loads) while the
//     current one is being consumed, then written to LDS behind a barrier.
//   * the column lives in LDS as 8-byte entries (g <= 20448 fits the CU's 160 KiB); 32
//     trailing zero entries absorb idle index slots, so the inner loop is branch-free.
//   * lanes = gene sets.  Each wavefront streams ITS OWN pre-built list of tiles (64 sets per
//     tile; geneset.cpp): 8 u16 gene ids per lane per 16-byte load (1 KiB per wave, coalesced,
//     L2-resident: the lists are shared by every column), two loads kept in flight.  Every
//     id becomes one ds_read_b64 gather; the visiting order was edge-coloured on the host so
//     the 32 lanes of each LDS lane group always hit 32 different bank pairs (conflict-free).
//   * fp64 accumulation (4 partial sums per lane), epilogue per set:
//       alpha * (sum * w) + beta * (k * w),  w = 1/(1e-8 + k) or 1,
//     plus the min(x)==0 bookkeeping normalize_medians needs (R/plaid.R:556-557).
//   * g > 20448 genes: the column is consumed in equal gene slices, one launch per slice;
//     partial sums rest in S between launches (acc_mode), the epilogue runs in the last one.
// Algorithmic HBM bytes per column: 8 g (X) + 8 m (S); the index lists (2 B per membership
// slot) come from L2 once per column.
#include "common.h"

#include <stdlib.h>
#include <string.h>

namespace plaidhip {

// diagnostic kernel variants: only in the tools/ build (make diag, -DPLAIDHIP_DIAG); the product library
// instantiates the plain kernels alone
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 1
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#else
static constexpr int g_ablate = 0;
#endif

// Plan arrays are read-only for the lifetime of a launch: wave-uniform reads of them go through the
// constant address space, i.e. the scalar cache (s_load), not through a vector load + readfirstlane,
// whose result would have to be awaited with a vmcnt that drains the index ring.
typedef __attribute__((address_space(4))) const int32_t* cptr_i32;
typedef __attribute__((address_space(4))) const plaidhip_pair_slice_dev* cptr_pair_slice;

struct SpmmArgs {
  const double* X;
  int64_t ldx;
  const int32_t* Xp;
  const int32_t* Xi;
  const double* Xx;
  int32_t g, n, m;   // g: genes of THIS slice
  int32_t g0;        // first gene of the slice (CSC rows are filtered / re-based by it)
  int32_t nt_store;  // 1: the lanes of a tile store neighbouring rows of S (streaming stores); 0: scattered rows, let L2 merge them
  int32_t acc_mode;  // 0 single slice; 1 first (store raw sum); 2 middle (S += sum); 3 last (S + sum, then epilogue)
  const uint4* tile_idx;
  const int32_t* wave_chunk_off;
  const int32_t* wave_tile_off;
  const int32_t* wtile_end;
  const int32_t* meta_j;     // [wave-stream tile k][lane] set id or -1
  const double* meta_w;      // [k][lane] 1/(1e-8 + size)  (R/plaid.R:75-76)
  const double* meta_k;      // [k][lane] size
  int32_t stat;
  double alpha, beta;
  const double* alpha_div;  // device scalar: alpha /= *alpha_div (global max(rX)), may be null
  double* S;
  int64_t lds;
  uint32_t* flags;
  // speculative launch + guarded fallback (see spec_guard): the u16 quad kernel is exact only for rank inputs, so it
  // publishes its flag words to spec[1..3] and stores spec_gen in spec[0] when it staged a value that is not a rank
  // (NaN / Inf / negative / not a multiple of 1/2 / 2x >= 2^16); the fp64 launch
  // enqueued behind it runs only then.  null: unconditional launch.
  uint32_t* spec;
  uint32_t spec_gen;
  unsigned long long* dbg;   // ABLATE==4 only: per (workgroup, wave) {stage, gather, tail-wait, total} cycles
};

// Fallback side of a speculative launch (the kernel enqueued right behind it on the same stream).  Returns true when this
// launch has nothing to do: the speculative kernel's input was what it assumed -- its private flag words are merged into
// the caller's (one thread) and the launch returns at once.  Otherwise the private words are dropped (the speculative
// scores are overwritten by this launch, which publishes its own flags).
__device__ __forceinline__ bool spec_guard(uint32_t* spec, uint32_t gen, uint32_t* flags) {
  if (spec == nullptr) return false;
  const bool bad = __hip_atomic_load(&spec[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const uint32_t w = __hip_atomic_load(&spec[1 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w != 0u) {
        if (!bad && flags != nullptr) __hip_atomic_store(&flags[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&spec[1 + b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  return !bad;
}

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

// u16 gene id (low / high half of a dword) -> LDS byte offset id*8.  Written so that the
// backend's SDWA peephole folds each into ONE v_lshlrev_b32_sdwa (sub-dword source select).
// Inline asm is avoided on purpose: next to asm statements hipcc stops counting vmcnt and
// drains the whole index ring (s_waitcnt vmcnt(0)) before every new load.
__device__ __forceinline__ uint32_t off_lo(uint32_t q) {
  // instcombine turns (q & 0xffff) << 3 into (q << 3) & 0x7fff8, which the peephole misses: spell it
  uint32_t r;
  asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
      : "=v"(r) : "v"(q));
  return r;
}
__device__ __forceinline__ uint32_t off_hi(uint32_t q) { return (q >> 16) << 3; }

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const double lds_cf64;
// The kernel has no static LDS, so the dynamic region starts at LDS address 0 and a byte
// offset into the column IS the LDS address (checked once at kernel entry).
template <int ABLATE = 0>
__device__ __forceinline__ double lds_at(uint32_t byte_off) {
  if constexpr (ABLATE == 1) return __longlong_as_double((long long)byte_off);
  return *reinterpret_cast<lds_cf64*>(static_cast<uintptr_t>(byte_off));
}

// the prefetched column lives in NAMED registers (an indexed array ends up in scratch)
#define PLAIDHIP_ITEMS(M) \
  M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19)

// ABLATE (diagnostic builds only, selected by tools/ through plaidhip_debug_set_ablation):
//   0 product kernel; 1 no LDS gathers (index stream + VALU only); 2 no index loads (synthetic
//   conflict-free ids: LDS + VALU only); 3 no column prefetch / staging.  Modes 1-3 give wrong
//   scores by construction and exist to price one pipe at a time.
// (the ablation / stamp arms exist in the tools/ build only: behind PLAIDHIP_DIAG at the macro, the product kernel reads plain)
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 2
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#else
#define PH_COL_PF true
#define PH_COL_STAMP false
#define PH_COL_LDS_AT(o_) lds_at<0>(o_)
#define PH_COL_SYNTH_IDS false
#endif
template <bool CSC_X, int BLOCK, int ABLATE = 0>
__global__ void __launch_bounds__(BLOCK)
spmm_colgather_f64(SpmmArgs a) {
  if (spec_guard(a.spec, a.spec_gen, a.flags)) return;
  // f64x2 registers per thread holding the prefetched next column
  constexpr int ITEMS2 = (BLOCK == 1024) ? 10 : (BLOCK == 512 ? 20 : 4);
  static_assert(BLOCK * ITEMS2 * 2 >= (BLOCK == 256 ? 2048 : kMaxLdsGenes), "prefetch span");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* col = reinterpret_cast<double*>(smem_raw);
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    if ((uint32_t)(uintptr_t)((lds_u8*)smem_raw) != 0u) __builtin_trap();
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t f = 0;
  const double alpha = (a.alpha_div != nullptr) ? a.alpha / *a.alpha_div : a.alpha;

  const int ch_begin = __builtin_amdgcn_readfirstlane(a.wave_chunk_off[wave]);
  const int ch_end = __builtin_amdgcn_readfirstlane(a.wave_chunk_off[wave + 1]);
  const int tk_begin = __builtin_amdgcn_readfirstlane(a.wave_tile_off[wave]);

  const int g2 = a.g >> 1;
  const uint32_t lane_off16 = (uint32_t)tid * 16u;
  const bool vec_ok = !CSC_X && g2 > 0 && ((a.ldx & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  f64x2 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12, p13, p14, p15, p16, p17, p18, p19;
  p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = p8 = p9 = f64x2{0.0, 0.0};
  p10 = p11 = p12 = p13 = p14 = p15 = p16 = p17 = p18 = p19 = f64x2{0.0, 0.0};

// loads are "uniform base (SGPRs) + 32-bit lane offset": no per-load address registers
#define PLAIDHIP_PF_ONE(k)                                                                      \
  if constexpr (k < ITEMS2) {                                                                    \
    if ((k + 1) * BLOCK <= g2) { /* wave-uniform: whole item in range */                         \
      p##k = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xb_ + (size_t)k * BLOCK * 16 + lane_off16)); \
    } else if (k * BLOCK < g2) { /* boundary item */                                             \
      if (tid + k * BLOCK < g2)                                                                  \
        p##k = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xb_ + (size_t)k * BLOCK * 16 + lane_off16)); \
    }                                                                                            \
  }
#define PLAIDHIP_PREFETCH(cc)                                                                        \
  do {                                                                                                \
    const char* xb_ = reinterpret_cast<const char*>(a.X) + (int64_t)(cc) * a.ldx * 8;                 \
    PLAIDHIP_ITEMS(PLAIDHIP_PF_ONE)                                                                   \
  } while (0)
#define PLAIDHIP_ST_ONE(k)                        \
  if constexpr (k < ITEMS2) if (k * BLOCK < g2) {  \
    const int i_ = tid + k * BLOCK;                \
    if (i_ < g2) col2[i_] = p##k;                  \
  }

  int c = blockIdx.x;
  if (PH_COL_PF && vec_ok && c < a.n && g2 > 0) PLAIDHIP_PREFETCH(c);

  unsigned long long t_stage = 0, t_gather = 0, t_wait = 0, t_all0 = 0;
  if constexpr (PH_COL_STAMP) t_all0 = __builtin_amdgcn_s_memtime();
  for (; c < a.n; c += gridDim.x) {
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if constexpr (PH_COL_STAMP) ts0 = __builtin_amdgcn_s_memtime();
    // ---- stage the sample column in LDS ------------------------------------------------
    if constexpr (!CSC_X) {
      const double* xc = a.X + (int64_t)c * a.ldx;
      if (!PH_COL_PF) {
      } else if (vec_ok) {
        f64x2* col2 = reinterpret_cast<f64x2*>(col);
        PLAIDHIP_ITEMS(PLAIDHIP_ST_ONE)
        if ((a.g & 1) && tid == 0) col[a.g - 1] = xc[a.g - 1];
      } else {
        for (int i = tid; i < a.g; i += BLOCK) col[i] = xc[i];
      }
      if (tid < kPadSlots) col[a.g + tid] = 0.0;
    } else {
      for (int i = tid; i < a.g + kPadSlots; i += BLOCK) col[i] = 0.0;
      __syncthreads();
      const int p0 = a.Xp[c], p1 = a.Xp[c + 1];
      for (int p = p0 + tid; p < p1; p += BLOCK) {
        const int r = a.Xi[p] - a.g0;
        if (r >= 0 && r < a.g) col[r] = a.Xx[p];
      }
    }
    __syncthreads();
    if constexpr (PH_COL_STAMP) ts1 = __builtin_amdgcn_s_memtime();
    const int cnext = c + gridDim.x;
    const bool want_pf = PH_COL_PF && vec_ok && cnext < a.n;

    // ---- gather: this wave's tile stream ---------------------------------------------------
    if (ch_begin < ch_end) {
      const char* ibase = reinterpret_cast<const char*>(a.tile_idx) + (int64_t)ch_begin * 1024;  // uniform
      const uint32_t ioff = (uint32_t)lane * 16u;
#define PLAIDHIP_LOADQ(rel) \
  (PH_COL_SYNTH_IDS ? make_uint4(lane | ((lane + 64u) << 16), (lane + 128u) | ((lane + 192u) << 16), \
                            (lane + 256u) | ((lane + 320u) << 16), (lane + 384u) | ((lane + 448u + (rel)) << 16)) \
               : *reinterpret_cast<const uint4*>(ibase + (int64_t)(rel) * 1024 + ioff))
#define PLAIDHIP_GATHER8(V, q)                                               \
  V##0 = PH_COL_LDS_AT(off_lo((q).x)); V##1 = PH_COL_LDS_AT(off_hi((q).x)); \
  V##2 = PH_COL_LDS_AT(off_lo((q).y)); V##3 = PH_COL_LDS_AT(off_hi((q).y)); \
  V##4 = PH_COL_LDS_AT(off_lo((q).z)); V##5 = PH_COL_LDS_AT(off_hi((q).z)); \
  V##6 = PH_COL_LDS_AT(off_lo((q).w)); V##7 = PH_COL_LDS_AT(off_hi((q).w));
#define PLAIDHIP_ADD8(V)                       \
  s0 += V##0; s1 += V##1; s2 += V##2; s3 += V##3; \
  s0 += V##4; s1 += V##5; s2 += V##6; s3 += V##7;
#define PLAIDHIP_TILE_END(chv)                                                                 \
  if ((chv) + 1 == next_end) { /* wave-uniform: tile finished -> epilogue */                   \
    double sum = (s0 + s1) + (s2 + s3);                                                        \
    if (mj >= 0) {                                                                             \
      double* sp_ = &a.S[(int64_t)c * a.lds + mj];                                             \
      if (a.acc_mode >= 2) sum += *sp_;               /* partial sums of earlier gene slices */ \
      if (a.acc_mode == 1 || a.acc_mode == 2) {                                                \
        *sp_ = sum;                                                                            \
      } else {                                                                                 \
        const double w = (a.stat == PLAIDHIP_STAT_MEAN) ? mw : 1.0;                            \
        const double v = alpha * (sum * w) + a.beta * (mk * w);                                \
        if (a.nt_store) __builtin_nontemporal_store(v, sp_); else *sp_ = v;                    \
        f |= (v < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;                                           \
        f |= (v == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;                                         \
        f |= (v != v) ? PLAIDHIP_FLAG_HAS_NAN : 0u;                                            \
      }                                                                                        \
    }                                                                                          \
    ++k;                                                                                       \
    next_end = __builtin_amdgcn_readfirstlane(a.wtile_end[k]);                                 \
    /* metadata of the next tile: issued now, consumed a whole tile later */                   \
    mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);  \
    mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);   \
    mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);   \
    s0 = s1 = s2 = s3 = 0.0;                                                                   \
  }
      // Software pipeline: 4 index chunks (4 KiB per wave) in flight from L2, and the LDS
      // gathers of chunk i+1 are issued BEFORE the adds of chunk i, so a wave never sits on
      // its own LDS latency.  The pipeline over-reads up to 5 chunks past the stream (spare
      // chunks exist behind the array; the gathered values are never added).
      uint4 qa = PLAIDHIP_LOADQ(0);
      uint4 qb = PLAIDHIP_LOADQ(1);
      uint4 qc = PLAIDHIP_LOADQ(2);
      uint4 qd = PLAIDHIP_LOADQ(3);
      int k = tk_begin;
      int next_end = __builtin_amdgcn_readfirstlane(a.wtile_end[k]);
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const uint32_t moff4 = (uint32_t)lane * 4u, moff8 = (uint32_t)lane * 8u;
      int mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);
      double mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);
      double mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);
      double va0, va1, va2, va3, va4, va5, va6, va7;
      double vb0, vb1, vb2, vb3, vb4, vb5, vb6, vb7;

      if constexpr (BLOCK == 1024) {
        // 16 waves per CU: thread-level parallelism hides the LDS latency; keep the loop lean
        // (128-VGPR budget).  4 index chunks in flight.
        int ch = ch_begin;
        ibase += 4 * 1024;
        for (; ch + 3 < ch_end; ch += 4, ibase += 4 * 1024) {
          PLAIDHIP_GATHER8(va, qa)
          qa = PLAIDHIP_LOADQ(0);
          PLAIDHIP_ADD8(va)
          PLAIDHIP_TILE_END(ch)
          PLAIDHIP_GATHER8(va, qb)
          qb = PLAIDHIP_LOADQ(1);
          PLAIDHIP_ADD8(va)
          PLAIDHIP_TILE_END(ch + 1)
          PLAIDHIP_GATHER8(va, qc)
          qc = PLAIDHIP_LOADQ(2);
          PLAIDHIP_ADD8(va)
          PLAIDHIP_TILE_END(ch + 2)
          PLAIDHIP_GATHER8(va, qd)
          qd = PLAIDHIP_LOADQ(3);
          PLAIDHIP_ADD8(va)
          PLAIDHIP_TILE_END(ch + 3)
        }
        if (ch < ch_end) { PLAIDHIP_GATHER8(va, qa) PLAIDHIP_ADD8(va) PLAIDHIP_TILE_END(ch) ++ch; }
        if (ch < ch_end) { PLAIDHIP_GATHER8(va, qb) PLAIDHIP_ADD8(va) PLAIDHIP_TILE_END(ch) ++ch; }
        if (ch < ch_end) { PLAIDHIP_GATHER8(va, qc) PLAIDHIP_ADD8(va) PLAIDHIP_TILE_END(ch) ++ch; }
      } else {
      PLAIDHIP_GATHER8(va, qa)
      qa = PLAIDHIP_LOADQ(4);
      int ch = ch_begin;
      ibase += 5 * 1024;   // LOADQ(rel) below: rel counted from chunk ch+5
      for (; ch + 3 < ch_end; ch += 4, ibase += 4 * 1024) {
        PLAIDHIP_GATHER8(vb, qb)
        qb = PLAIDHIP_LOADQ(0);
        PLAIDHIP_ADD8(va)
        PLAIDHIP_TILE_END(ch)
        PLAIDHIP_GATHER8(va, qc)
        qc = PLAIDHIP_LOADQ(1);
        PLAIDHIP_ADD8(vb)
        PLAIDHIP_TILE_END(ch + 1)
        PLAIDHIP_GATHER8(vb, qd)
        qd = PLAIDHIP_LOADQ(2);
        PLAIDHIP_ADD8(va)
        PLAIDHIP_TILE_END(ch + 2)
        PLAIDHIP_GATHER8(va, qa)
        qa = PLAIDHIP_LOADQ(3);
        PLAIDHIP_ADD8(vb)
        PLAIDHIP_TILE_END(ch + 3)
      }
      if (ch < ch_end) {
        PLAIDHIP_GATHER8(vb, qb)
        PLAIDHIP_ADD8(va)
        PLAIDHIP_TILE_END(ch)
        ++ch;
        if (ch < ch_end) {
          PLAIDHIP_GATHER8(va, qc)
          PLAIDHIP_ADD8(vb)
          PLAIDHIP_TILE_END(ch)
          ++ch;
          if (ch < ch_end) {
            PLAIDHIP_ADD8(va)
            PLAIDHIP_TILE_END(ch)
          }
        }
      }
      }  // BLOCK != 1024
      // next column: issued when this wave's stream is done (older waves finish first and
      // their loads fly while the younger ones still gather); no VMEM inside the hot loop
      // besides the index ring, which keeps hipcc's vmcnt counting exact.
      if (want_pf) PLAIDHIP_PREFETCH(cnext);
    } else if (want_pf) {
      PLAIDHIP_PREFETCH(cnext);
    }
    if constexpr (PH_COL_STAMP) ts2 = __builtin_amdgcn_s_memtime();
    __syncthreads();  // column is overwritten by the next iteration
    if constexpr (PH_COL_STAMP) {
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      t_stage += ts1 - ts0;
      t_gather += ts2 - ts1;
      t_wait += ts3 - ts2;
    }
  }
  if constexpr (PH_COL_STAMP) {
    if (lane == 0 && a.dbg != nullptr) {
      unsigned long long* d = a.dbg + ((size_t)blockIdx.x * (BLOCK / 64) + wave) * 4;
      d[0] = t_stage; d[1] = t_gather; d[2] = t_wait; d[3] = __builtin_amdgcn_s_memtime() - t_all0;
    }
  }
  publish_flags(f, a.flags);
#undef PH_COL_PF
#undef PH_COL_STAMP
#undef PH_COL_LDS_AT
#undef PH_COL_SYNTH_IDS
#undef PLAIDHIP_GATHER8
#undef PLAIDHIP_ADD8
#undef PLAIDHIP_TILE_END
#undef PLAIDHIP_LOADQ
#undef PLAIDHIP_PREFETCH
#undef PLAIDHIP_PF_ONE
#undef PLAIDHIP_ST_ONE
}


// ---------------------------------------------------------------------------------------------
// Pair kernel (dense X): TWO sample columns per pass.  An LDS entry is 16 bytes {A_i, B_i}, so one
// address op + one ds_read_b128 serve two scores and the index stream is read once per pair.  Only
// half the genes fit (<= 10,224 per slice): a workgroup walks the gene slices of its pair inside
// one launch; partial sums rest in a private [workgroup][tile][lane] scratch (L2-resident) between
// slices, the epilogue runs in the last slice.  The host schedule (geneset.cpp, plan_tile_b128)
// is conflict-free for the four 16-lane groups a ds_read_b128 is served in.
struct SpmmPairArgs {
  const double* X;
  int64_t ldx;
  int64_t sparse_cells;   // CSC X, auto mode: return at once if 8 * nnz(X) < g * n (the scatter kernel takes it); 0 = always run
  const int32_t* Xp;   // CSC X (dgCMatrix slots) instead of dense X
  const int32_t* Xi;
  const double* Xx;
  int32_t n, npairs, nslices, ktiles;
  int32_t nt_store;  // see SpmmArgs
  const plaidhip_pair_slice_dev* slices;
  const int32_t* wave_tile_off;
  const int32_t* meta_j;
  const double* meta_w;
  const double* meta_k;
  f64x2* partial;
  int32_t stat;
  double alpha, beta;
  const double* alpha_div;
  double* S;
  int64_t lds;
  uint32_t* flags;
  uint32_t* spec;            // guarded fallback of a speculative launch (SpmmArgs::spec), null: unconditional
  uint32_t spec_gen;
  unsigned long long* dbg;   // STAMP only
  // MED (dense X; medians selected on the fly, like spmm_scatter_csc_f64<.., MED> below): per-gene mean set weight u[] of
  // this launch's statistic and beta x kappa (the column's mean score is alpha * sum_i x[i, c] u[i] + beta * kappa: the
  // workgroup stages every x[i, c] anyway), the calibration {offset, half width, ignore-zero}, per (column, wavefront) a
  // slice of `med_capc` candidate scores and the counts {below, zero, NaN, candidates}; med_pred: the predicted means (out)
  const double* med_u;
  double med_beta_kappa;
  const double* med_cal;
  double* med_pred;
  unsigned long long* med_cand;
  uint32_t* med_cnt;
  int32_t med_capc;
};

__device__ __forceinline__ uint32_t off16_lo(uint32_t q) {
  uint32_t r;
  asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
      : "=v"(r) : "v"(q));
  return r;
}
__device__ __forceinline__ uint32_t off16_hi(uint32_t q) { return (q >> 16) << 4; }
typedef __attribute__((address_space(3))) const f64x2 lds_cf64x2;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const int32_t* gptr_i32;
typedef __attribute__((address_space(1))) const unsigned char* gptr_u8;
typedef __attribute__((address_space(1))) const u32x4* gptr_u32x4;
__device__ __forceinline__ f64x2 lds_pair_at(uint32_t byte_off) {
  return *reinterpret_cast<lds_cf64x2*>(static_cast<uintptr_t>(byte_off));
}

// MED (dense X only): the launch also classifies every score it writes for normalize_medians (R/plaid.R:561-572) -- counts
// below the bracket / exact zeros / NaN per (column, wavefront) and the scores inside the bracket to a candidate list --
// so that the medians of a 50,000-set result need no second pass over the scores (launch_spmm_dense_fused_f64).  The
// bracket sits around the column's MEAN score, which the workgroup computes from the X it stages: sum_i x[i, c] u[i].
// PNT: the partial sums of the slice before are read with non-temporal loads (a compile-time form: hipcc merges the two
// arms of a run-time choice into ONE plain load).
template <bool STAMP, int ABL = 0, bool CSC_X = false, bool MED = false, bool PNT = false>
__global__ void __launch_bounds__(1024)
spmm_colpair_f64(SpmmPairArgs a) {
  constexpr int BLOCK = 1024;
  static_assert(!(MED && CSC_X), "the classifying epilogue is built for dense X");
  if (spec_guard(a.spec, a.spec_gen, a.flags)) return;
  if constexpr (CSC_X) {
    if (a.sparse_cells != 0 && ((int64_t)a.Xp[a.n] - a.Xp[0]) * 8 < a.sparse_cells) return;
  }
  unsigned long long t_stage = 0, t_gather = 0, t_wait = 0, t_all0 = 0;
  if constexpr (STAMP) t_all0 = __builtin_amdgcn_s_memtime();
  extern __shared__ __align__(16) unsigned char smem_raw[];
  f64x2* ent = reinterpret_cast<f64x2*>(smem_raw);
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    if ((uint32_t)(uintptr_t)((lds_u8*)smem_raw) != 0u) __builtin_trap();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t f = 0;
  // (wave-uniform, but the division leaves it in vector registers: moved to scalar ones -- the gather loop has no vector
  // register to spare: 128 -> 126, C2 0.93 -> 0.89 ms)
  const double alpha_v = (a.alpha_div != nullptr) ? a.alpha / *a.alpha_div : a.alpha;
  const double alpha = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(alpha_v)),
                                        __builtin_amdgcn_readfirstlane(__double2loint(alpha_v)));
  const int tk_begin = ((cptr_i32)a.wave_tile_off)[wave];
  const int ns = a.nslices;
  const bool is_mean = a.stat == PLAIDHIP_STAT_MEAN;
  f64x2 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9;   // next slice: p0..p4 column A, p5..p9 column B
  p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = p8 = p9 = f64x2{0.0, 0.0};

#define PLAIDHIP_PF_ONE(k, reg, base)                                                              \
  if ((k + 1) * BLOCK <= g2_) {                                                                     \
    reg = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(base + (size_t)k * BLOCK * 16 + lane_off16)); \
  } else { /* dead during the gather loop: never carries an old value across it */                  \
    reg = f64x2{0.0, 0.0};                                                                          \
    if (k * BLOCK < g2_ && tid + k * BLOCK < g2_)                                                   \
      reg = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(base + (size_t)k * BLOCK * 16 + lane_off16)); \
  }
#define PLAIDHIP_PREFETCH(pp_, si_)                                                                 \
  do {                                                                                              \
    uint32_t lane_off16 = (uint32_t)tid * 16u;                                                      \
    asm volatile("" : "+v"(lane_off16)); /* keep address math out of the enclosing loops */         \
    const int g0_ = ((cptr_pair_slice)a.slices)[si_].g0;                                            \
    const int g2_ = ((cptr_pair_slice)a.slices)[si_].gs >> 1;                                                       \
    const int ca_ = 2 * (pp_);                                                                      \
    const int cb_ = (ca_ + 1 < a.n) ? ca_ + 1 : ca_;                                                \
    const char* xa_ = reinterpret_cast<const char*>(a.X + (int64_t)ca_ * a.ldx + g0_);              \
    const char* xb_ = reinterpret_cast<const char*>(a.X + (int64_t)cb_ * a.ldx + g0_);              \
    PLAIDHIP_PF_ONE(0, p0, xa_) PLAIDHIP_PF_ONE(1, p1, xa_) PLAIDHIP_PF_ONE(2, p2, xa_)             \
    PLAIDHIP_PF_ONE(3, p3, xa_) PLAIDHIP_PF_ONE(4, p4, xa_)                                         \
    PLAIDHIP_PF_ONE(0, p5, xb_) PLAIDHIP_PF_ONE(1, p6, xb_) PLAIDHIP_PF_ONE(2, p7, xb_)             \
    PLAIDHIP_PF_ONE(3, p8, xb_) PLAIDHIP_PF_ONE(4, p9, xb_)                                         \
  } while (0)
#define PLAIDHIP_ST_ONE(k, ra, rb)                      \
  if (k * BLOCK < g2) {                                  \
    const int i_ = tid_o + k * BLOCK;                    \
    if (i_ < g2) {                                       \
      ent[2 * i_] = f64x2{ra.x, rb.x};                   \
      ent[2 * i_ + 1] = f64x2{ra.y, rb.y};               \
    }                                                    \
  }

  int p = blockIdx.x;
  if (!CSC_X && p < a.npairs) PLAIDHIP_PREFETCH(p, 0);
  const char* part = reinterpret_cast<const char*>(a.partial + (size_t)blockIdx.x * (size_t)(a.ktiles + 1) * 64);  // uniform

  // MED: wave-uniform state of the pair being written (scalar registers)
  double med_dotA = 0.0, med_dotB = 0.0;                      // this thread's share of sum_i x[i, c] u[i]
  double med_loA = 0.0, med_hiA = 0.0, med_loB = 0.0, med_hiB = 0.0;
  uint32_t w_ltA = 0, w_zeroA = 0, w_candA = 0, w_ltB = 0, w_zeroB = 0, w_candB = 0;
  uint32_t w_nanA = 0, w_nanB = 0;                            // a NaN score of column A / B was written by this wavefront
  uint32_t med_fw = 0;                                        // flag bits from the lane masks (wave-uniform)
  unsigned long long* med_sliceA = nullptr;
  unsigned long long* med_sliceB = nullptr;
  bool med_iz = false;
  if constexpr (MED) med_iz = a.med_cal[2] != 0.0;

  for (; p < a.npairs; p += gridDim.x) {
    const int cA = 2 * p;
    const bool hasB = cA + 1 < a.n;
    const int cB = hasB ? cA + 1 : cA;
    if constexpr (MED) {
      med_dotA = med_dotB = 0.0;
      w_ltA = w_zeroA = w_candA = w_ltB = w_zeroB = w_candB = 0u;
      w_nanA = w_nanB = 0u;
      med_sliceA = a.med_cand + ((int64_t)cA * (BLOCK / 64) + wave) * a.med_capc;
      med_sliceB = a.med_cand + ((int64_t)cB * (BLOCK / 64) + wave) * a.med_capc;
    }
    for (int si = 0; si < ns; ++si) {
      const cptr_pair_slice sl = (cptr_pair_slice)(a.slices + si);
      const int gs_ = sl->gs;
      const int g2 = gs_ >> 1;
      unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
      if constexpr (STAMP) ts0 = __builtin_amdgcn_s_memtime();
      int tid_o = tid;
      asm volatile("" : "+v"(tid_o));
      if constexpr (MED) {
        // this slice's share of the column means: the prefetched X of the slice is in p0..p9 (zero past the slice's end), u
        // comes from L2 (160 KB per statistic); unconditional loads at clamped indices: one round trip for all five
        const char* ub_ = reinterpret_cast<const char*>(a.med_u + sl->g0);
        const int last2_ = g2 > 0 ? g2 - 1 : 0;
#define PLAIDHIP_DOT_ONE(k, ra, rb)                                                                    \
  if (k * BLOCK < g2) {                                                                                 \
    const int i_ = tid_o + k * BLOCK;                                                                   \
    const f64x2 u_ = *reinterpret_cast<const f64x2*>(ub_ + (size_t)(i_ < last2_ ? i_ : last2_) * 16);  \
    med_dotA += ra.x * u_.x + ra.y * u_.y;                                                              \
    med_dotB += rb.x * u_.x + rb.y * u_.y;                                                              \
  }
        PLAIDHIP_DOT_ONE(0, p0, p5) PLAIDHIP_DOT_ONE(1, p1, p6) PLAIDHIP_DOT_ONE(2, p2, p7)
        PLAIDHIP_DOT_ONE(3, p3, p8) PLAIDHIP_DOT_ONE(4, p4, p9)
#undef PLAIDHIP_DOT_ONE
        if ((gs_ & 1) && tid == 0) {
          const int64_t gl = (int64_t)sl->g0 + gs_ - 1;
          const double u_ = a.med_u[gl];
          med_dotA += a.X[(int64_t)cA * a.ldx + gl] * u_;
          med_dotB += a.X[(int64_t)cB * a.ldx + gl] * u_;
        }
        if (si == ns - 1) {
          // the column means, the same bits on every lane: wave sums -> 16 LDS entries (the slice of the pass before is
          // dead behind its end barrier, this one is not staged yet) -> one fixed-order sum
          double sa_ = med_dotA, sb_ = med_dotB;
          for (int off = 32; off >= 1; off >>= 1) { sa_ += __shfl_xor(sa_, off, 64); sb_ += __shfl_xor(sb_, off, 64); }
          if (lane == 0) ent[wave] = f64x2{sa_, sb_};
          __syncthreads();
          const f64x2 t_ = ent[lane & 15];
          sa_ = t_.x; sb_ = t_.y;
          for (int off = 8; off >= 1; off >>= 1) { sa_ += __shfl_xor(sa_, off, 64); sb_ += __shfl_xor(sb_, off, 64); }
          __syncthreads();   // (the staging below overwrites the entries)
          const double prA_ = alpha * sa_ + a.med_beta_kappa, prB_ = alpha * sb_ + a.med_beta_kappa;
          if (tid == 0) {
            a.med_pred[cA] = prA_;
            if (hasB) a.med_pred[cB] = prB_;
          }
          const double off_ = a.med_cal[0], hw_ = a.med_cal[1];
          auto uni = [](double x) {
            return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
          };
          med_loA = uni((prA_ + off_) - hw_);
          med_hiA = uni((prA_ + off_) + hw_);
          med_loB = uni((prB_ + off_) - hw_);
          med_hiB = uni((prB_ + off_) + hw_);
        }
      }
      if constexpr (!CSC_X) {
        // ---- stage the slice of both columns, interleaved ---------------------------------
        PLAIDHIP_ST_ONE(0, p0, p5) PLAIDHIP_ST_ONE(1, p1, p6) PLAIDHIP_ST_ONE(2, p2, p7)
        PLAIDHIP_ST_ONE(3, p3, p8) PLAIDHIP_ST_ONE(4, p4, p9)
        if ((gs_ & 1) && tid == 0) {
          const int64_t gl = (int64_t)sl->g0 + gs_ - 1;
          ent[gs_ - 1] = f64x2{a.X[(int64_t)cA * a.ldx + gl], a.X[(int64_t)cB * a.ldx + gl]};
        }
        if (tid < kPadSlotsPair) ent[gs_ + tid] = f64x2{0.0, 0.0};
      } else {
        // ---- sparse columns: zero the slice, then scatter the stored values of both columns ----
        for (int i = tid_o; i < gs_ + kPadSlotsPair; i += BLOCK) ent[i] = f64x2{0.0, 0.0};
        __syncthreads();
        double* entd = reinterpret_cast<double*>(ent);
        const int g0_ = sl->g0;
        {
          const int q0 = a.Xp[cA], q1 = a.Xp[cA + 1];
          for (int q = q0 + tid_o; q < q1; q += BLOCK) {
            const int r = a.Xi[q] - g0_;
            if (r >= 0 && r < gs_) entd[2 * r] = a.Xx[q];
          }
        }
        if (hasB) {
          const int q0 = a.Xp[cB], q1 = a.Xp[cB + 1];
          for (int q = q0 + tid_o; q < q1; q += BLOCK) {
            const int r = a.Xi[q] - g0_;
            if (r >= 0 && r < gs_) entd[2 * r + 1] = a.Xx[q];
          }
        }
      }
      __syncthreads();
      if constexpr (STAMP) ts1 = __builtin_amdgcn_s_memtime();
      int nsi = si + 1, np = p;
      if (nsi == ns) { nsi = 0; np = p + gridDim.x; }
      const bool want_pf = !CSC_X && np < a.npairs;
      const bool first = si == 0, last = si == ns - 1;
      const cptr_i32 wco = (cptr_i32)sl->wave_chunk_off;
      const int ch_begin = wco[wave];
      const int ch_end = wco[wave + 1];

      if (ch_begin < ch_end) {
        // pointers read from memory are generic to the compiler: say "global" or it emits flat loads
        const cptr_i32 wtile_end = (cptr_i32)sl->wtile_end;
        gptr_u8 ibase = (gptr_u8)sl->tile_idx + (int64_t)ch_begin * 1024;  // uniform
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 3
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
        uint32_t lane_o = (uint32_t)lane;
        asm volatile("" : "+v"(lane_o));
        const uint32_t ioff = lane_o * 16u;
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 4
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#else
#define PLAIDHIP_LOADQ(rel) (*(gptr_u32x4)(ibase + (int64_t)(rel) * 1024 + ioff))
#define PH_PAIR_PSTORE true
#define PH_PAIR_PLOAD true
#define PH_PAIR_PSLOT(k_) (k_)
#define PH_PAIR_PST_NT false
#define PH_PAIR_PLD_NT false
#define PH_PAIR_PST_SC1 false
#define PH_PAIR_PLD_SC1 false
#define PH_PAIR_SST_SC1 false
#define PH_PAIR_META true
#endif
#define PLAIDHIP_GATHER4A(q)                                                   \
  va0 = lds_pair_at(off16_lo((q).x)); va1 = lds_pair_at(off16_hi((q).x));       \
  va2 = lds_pair_at(off16_lo((q).y)); va3 = lds_pair_at(off16_hi((q).y));
#define PLAIDHIP_GATHER4B(q)                                                   \
  vb0 = lds_pair_at(off16_lo((q).z)); vb1 = lds_pair_at(off16_hi((q).z));       \
  vb2 = lds_pair_at(off16_lo((q).w)); vb3 = lds_pair_at(off16_hi((q).w));
#define PLAIDHIP_ADD4(V)                                                       \
  a0 += (V##0).x; b0 += (V##0).y; a1 += (V##1).x; b1 += (V##1).y;                       \
  a2 += (V##2).x; b2 += (V##2).y; a3 += (V##3).x; b3 += (V##3).y;
#define PLAIDHIP_EPI(sum, cc, out_)                                            \
  {                                                                            \
    const double w_ = is_mean ? mw : 1.0;                                      \
    const double v_ = alpha * ((sum) * w_) + a.beta * (mk * w_);               \
    if (PH_PAIR_SST_SC1) __hip_atomic_store(&a.S[(int64_t)(cc) * a.lds + mj], v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    else if (a.nt_store) __builtin_nontemporal_store(v_, &a.S[(int64_t)(cc) * a.lds + mj]);  \
    else a.S[(int64_t)(cc) * a.lds + mj] = v_;                                 \
    if constexpr (!MED) {   /* (MED: the flag words come out of the classification's lane masks, below) */ \
      f |= (v_ < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;                            \
      f |= (v_ == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;                          \
      f |= (v_ != v_) ? PLAIDHIP_FLAG_HAS_NAN : 0u;                            \
    }                                                                          \
    out_ = v_;                                                                 \
  }
/* MED: five compares per score, all as lane masks in scalar registers -- below the bracket, not above it, exactly zero,  \
   negative, NaN: scalar counters and the three flag words (no per-lane flag arithmetic in this form); the scores inside   \
   the bracket are appended to this wavefront's own slice of the column's candidate list (no shared counter, no atomic).   \
   Lanes without a set carry NaN and fail the first four compares; `set_` masks them out of the fifth. */                 \
#define PLAIDHIP_MED_CLASSIFY(v_, X, set_)                                                                        \
  {                                                                                                               \
    const uint64_t lt_ = __ballot((v_) < med_lo##X);                                                              \
    const uint64_t le_ = __ballot((v_) <= med_hi##X);                                                             \
    const uint64_t zero_ = __ballot((v_) == 0.0);                                                                 \
    const uint64_t neg_ = __ballot((v_) < 0.0);                                                                   \
    const uint64_t nan_ = __ballot((v_) != (v_)) & (set_);                                                        \
    med_fw |= (neg_ != 0ull ? PLAIDHIP_FLAG_HAS_NEG : 0u) | (zero_ != 0ull ? PLAIDHIP_FLAG_HAS_ZERO : 0u) |        \
              (nan_ != 0ull ? PLAIDHIP_FLAG_HAS_NAN : 0u);                                                        \
    w_nan##X |= nan_ != 0ull ? 1u : 0u;                                                                           \
    const uint64_t in_ = (le_ & ~lt_) & (med_iz ? ~zero_ : ~0ull);                                                \
    w_lt##X += (uint32_t)__popcll(lt_);                                                                           \
    w_zero##X += (uint32_t)__popcll(zero_);                                                                       \
    if (in_ != 0ull) {                                                                                            \
      const uint32_t slot_ = w_cand##X + __builtin_amdgcn_mbcnt_hi((uint32_t)(in_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)in_, 0u)); \
      if (__builtin_amdgcn_inverse_ballot_w64(in_) && slot_ < (uint32_t)a.med_capc)                               \
        med_slice##X[slot_] = (unsigned long long)__double_as_longlong(v_);                                      \
      w_cand##X += (uint32_t)__popcll(in_);                                                                       \
    }                                                                                                             \
  }
#define PLAIDHIP_TILE_END(chv)                                                                 \
  if ((chv) + 1 == next_end) { /* wave-uniform: tile finished */                               \
    const double sumA = ((a0 + a1) + (a2 + a3)) + old.x;                                       \
    const double sumB = ((b0 + b1) + (b2 + b3)) + old.y;                                       \
    if (!last) {                                                                               \
      if (PH_PAIR_PSTORE) {                                                                    \
        f64x2* dst_ = reinterpret_cast<f64x2*>(const_cast<char*>(part) + (int64_t)PH_PAIR_PSLOT(k) * 1024 + ioff);  \
        if (PH_PAIR_PST_NT) __builtin_nontemporal_store(f64x2{sumA, sumB}, dst_);              \
        else if (PH_PAIR_PST_SC1) {                                                            \
          __hip_atomic_store(reinterpret_cast<double*>(dst_), sumA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      \
          __hip_atomic_store(reinterpret_cast<double*>(dst_) + 1, sumB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
        } else *dst_ = f64x2{sumA, sumB};                                                      \
      }                                                                                        \
    } else {                                                                                   \
      double vA_ = __longlong_as_double(0x7ff8000000000000ll), vB_ = vA_;                      \
      if (mj >= 0) {                                                                           \
        PLAIDHIP_EPI(sumA, cA, vA_)                                                            \
        if (hasB) PLAIDHIP_EPI(sumB, cB, vB_)                                                  \
      }                                                                                        \
      if constexpr (MED) {                                                                     \
        const uint64_t set_ = __ballot(mj >= 0);                                               \
        PLAIDHIP_MED_CLASSIFY(vA_, A, set_)                                                    \
        PLAIDHIP_MED_CLASSIFY(vB_, B, hasB ? set_ : 0ull)                                      \
      }                                                                                        \
    }                                                                                          \
    ++k;                                                                                       \
    next_end = wtile_end[k];                                   \
    if (last && PH_PAIR_META) {                                                                    \
      mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);  \
      mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);   \
      mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);   \
    }                                                                                          \
    if (!first && PH_PAIR_PLOAD) {                                                             \
      const f64x2* src_ = reinterpret_cast<const f64x2*>(part + (int64_t)PH_PAIR_PSLOT(k) * 1024 + ioff);  \
      if (PH_PAIR_PLD_SC1) {                                                                   \
        old.x = __hip_atomic_load(reinterpret_cast<const double*>(src_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      \
        old.y = __hip_atomic_load(reinterpret_cast<const double*>(src_) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
      } else old = (PH_PAIR_PLD_NT || PNT) ? __builtin_nontemporal_load(src_) : *src_;     \
    }                                                                                          \
    a0 = a1 = a2 = a3 = b0 = b1 = b2 = b3 = 0.0;                                               \
  }
        // 8 index chunks (8 KiB per wave) in flight: the lists come from L2 (~1 us away under load)
        u32x4 qa = PLAIDHIP_LOADQ(0);
        u32x4 qb = PLAIDHIP_LOADQ(1);
        u32x4 qc = PLAIDHIP_LOADQ(2);
        u32x4 qd = PLAIDHIP_LOADQ(3);
        u32x4 qe = PLAIDHIP_LOADQ(4);
        u32x4 qf = PLAIDHIP_LOADQ(5);
        u32x4 qg = PLAIDHIP_LOADQ(6);
        u32x4 qh = PLAIDHIP_LOADQ(7);
        int k = tk_begin;
        int next_end = wtile_end[k];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        const uint32_t moff4 = lane_o * 4u, moff8 = lane_o * 8u;
        int mj = -1;
        double mw = 0.0, mk = 0.0;
        if (last) {
          mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);
          mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);
          mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);
        }
        f64x2 old = f64x2{0.0, 0.0};
        if (!first && PH_PAIR_PLOAD) {
          const f64x2* src_ = reinterpret_cast<const f64x2*>(part + (int64_t)PH_PAIR_PSLOT(k) * 1024 + ioff);
          old = (PH_PAIR_PLD_NT || PNT) ? __builtin_nontemporal_load(src_) : *src_;
        }
        f64x2 va0, va1, va2, va3, vb0, vb1, vb2, vb3;

        // Half-chunk software pipeline: the four gathers of the next half are in the LDS queue
        // while the eight adds of the current half issue, so the wave always has reads in flight.
        int ch = ch_begin;
        ibase += 8 * 1024;
#define PLAIDHIP_STEP(qcur, qnext, rel, chv)                                             \
  PLAIDHIP_GATHER4B(qcur)                                                                \
  qcur = PLAIDHIP_LOADQ(rel);                                                            \
  PLAIDHIP_ADD4(va)                                                                      \
  PLAIDHIP_GATHER4A(qnext) /* first half of the next chunk (spare chunks exist behind the stream) */ \
  PLAIDHIP_ADD4(vb)                                                                      \
  PLAIDHIP_TILE_END(chv)
#define PLAIDHIP_STEP_TAIL(qcur, qnext)                                                  \
  if (ch < ch_end) {                                                                     \
    PLAIDHIP_GATHER4B(qcur) PLAIDHIP_ADD4(va) PLAIDHIP_GATHER4A(qnext) PLAIDHIP_ADD4(vb) \
    PLAIDHIP_TILE_END(ch)                                                                \
    ++ch;                                                                                \
  }
        PLAIDHIP_GATHER4A(qa)
        for (; ch + 7 < ch_end; ch += 8, ibase += 8 * 1024) {
          PLAIDHIP_STEP(qa, qb, 0, ch)
          PLAIDHIP_STEP(qb, qc, 1, ch + 1)
          PLAIDHIP_STEP(qc, qd, 2, ch + 2)
          PLAIDHIP_STEP(qd, qe, 3, ch + 3)
          PLAIDHIP_STEP(qe, qf, 4, ch + 4)
          PLAIDHIP_STEP(qf, qg, 5, ch + 5)
          PLAIDHIP_STEP(qg, qh, 6, ch + 6)
          PLAIDHIP_STEP(qh, qa, 7, ch + 7)
        }
        PLAIDHIP_STEP_TAIL(qa, qb)
        PLAIDHIP_STEP_TAIL(qb, qc)
        PLAIDHIP_STEP_TAIL(qc, qd)
        PLAIDHIP_STEP_TAIL(qd, qe)
        PLAIDHIP_STEP_TAIL(qe, qf)
        PLAIDHIP_STEP_TAIL(qf, qg)
        PLAIDHIP_STEP_TAIL(qg, qh)
#undef PLAIDHIP_STEP
#undef PLAIDHIP_STEP_TAIL
#undef PLAIDHIP_GATHER4A
#undef PLAIDHIP_GATHER4B
#undef PLAIDHIP_ADD4
#undef PLAIDHIP_TILE_END
#undef PLAIDHIP_EPI
#undef PLAIDHIP_MED_CLASSIFY
#undef PLAIDHIP_LOADQ
#undef PH_PAIR_PSTORE
#undef PH_PAIR_PLOAD
#undef PH_PAIR_PSLOT
#undef PH_PAIR_PST_NT
#undef PH_PAIR_PLD_NT
#undef PH_PAIR_PST_SC1
#undef PH_PAIR_PLD_SC1
#undef PH_PAIR_SST_SC1
#undef PH_PAIR_META
      }
      if (want_pf) {
        PLAIDHIP_PREFETCH(np, nsi);
      } else {  // (tells the register allocator the old values are not needed across the gather loop)
        p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = p8 = p9 = f64x2{0.0, 0.0};
      }
      if constexpr (STAMP) ts2 = __builtin_amdgcn_s_memtime();
      __syncthreads();  // the slice is overwritten by the next iteration
      if constexpr (STAMP) {
        const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
        t_stage += ts1 - ts0;
        t_gather += ts2 - ts1;
        t_wait += ts3 - ts2;
      }
    }
    if constexpr (MED) {   // this wavefront's counts of the pair's two columns (a wavefront without tiles writes zeros)
      const uint32_t nanA_ = w_nanA, nanB_ = w_nanB;
      if (lane == 0) {
        uint4* o_ = reinterpret_cast<uint4*>(a.med_cnt);
        // {scores below the bracket that take part, exact zeros, NaN (lanes, not scores: only "any" matters), candidates}
        o_[(int64_t)cA * (BLOCK / 64) + wave] = make_uint4((med_iz && med_loA > 0.0) ? w_ltA - w_zeroA : w_ltA, w_zeroA, nanA_, w_candA);
        if (hasB) o_[(int64_t)cB * (BLOCK / 64) + wave] = make_uint4((med_iz && med_loB > 0.0) ? w_ltB - w_zeroB : w_ltB, w_zeroB, nanB_, w_candB);
      }
    }
  }
  if constexpr (STAMP) {
    if (lane == 0 && a.dbg != nullptr) {
      unsigned long long* d = a.dbg + ((size_t)blockIdx.x * (BLOCK / 64) + wave) * 4;
      d[0] = t_stage; d[1] = t_gather; d[2] = t_wait; d[3] = __builtin_amdgcn_s_memtime() - t_all0;
    }
  }
  if constexpr (MED) f |= med_fw;
  publish_flags(f, a.flags);
#undef PLAIDHIP_PREFETCH
#undef PLAIDHIP_PF_ONE
#undef PLAIDHIP_ST_ONE
}


// ---------------------------------------------------------------------------------------------
// Scatter kernel (sparse X, dgCMatrix): work proportional to the stored values.  A workgroup owns
// one sample column; the scores of a chunk of gene sets (<= 20,480) are fp64 accumulators in LDS.
// Every stored value x[i, c] is added to the accumulators of the sets that contain gene i
// (ds_add_f64; G is stored gene-major in 128-id segments, geneset.cpp), then the chunk is scaled
// and written out coalesced.  A wavefront loads 64 stored values at once (row, value, segment
// range per lane) and walks them eight at a time with two 256-byte id segments per value in
// flight.  At 5 % density this is ~20x fewer LDS operations than gathering every membership.
// Sums are accumulated in arrival order, so the last bits differ from run to run (fp64, ~1e-16).
// raw buffer loads: address = descriptor base + per-lane VGPR offset + SCALAR offset -- a wave-uniform part of the address
// costs no vector instruction (a global load wants the whole address in vector registers or a 64-bit scalar base per load)
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
__device__ int32_t raw_buffer_load_i32(i32x4 rsrc, int32_t voffset, int32_t soffset, int32_t aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ __forceinline__ i32x4 make_raw_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  i32x4 r;
  r.x = (int32_t)(uint32_t)a;
  r.y = (int32_t)((uint32_t)(a >> 32) & 0xffffu);   // stride 0
  r.z = (int32_t)bytes;
  r.w = 0x00020000;                                 // 32-bit raw data format (gfx90a / gfx94x / gfx950)
  return r;
}

// fmin() compiles to a canonicalising v_max_f64 x, x in front of every v_min_f64 (IEEE mode: quiet a signalling NaN first);
// the instruction alone already returns the other operand for any NaN -- two vector instructions per score saved
__device__ __forceinline__ double min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double min_abs_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct ScatterArgs {
  const int32_t* Xp;
  const int32_t* Xi;
  const double* Xx;
  int32_t n, m, g, ch, nch;
  int64_t dense_cells;      // auto mode: run only if 8 * nnz(X) < g * n (0 = always run)
  const int32_t* seg;
  const uint16_t* ids;
  uint32_t ids_bytes;
  int32_t dummy_seg;
  const double* w;
  const double* k;
  const f64x2* kw;           // {size x weight, weight} per set for this launch's statistic
  int32_t stat, nt_store;
  double alpha, beta;
  const double* alpha_div;
  double* S;
  int64_t lds;
  uint32_t* flags;
  int32_t chunk_major;       // item order: 0 (column, chunk, round) | 1 (chunk, column, round): every workgroup is on the
                             // same chunk of sets at the same time, so the id lists in use are one chunk's (a third of them)
  // Choice ON THE DEVICE between the fixed-point (FIXED) and the fp64 launch of the same call (scatter_fixed_ok): both are
  // enqueued, the one that does not apply returns at once.  sel = {0 or -1, max, smallest value > 0} of the stored values
  // (launch_nonneg_range, swept right before); sel_want 1: run only if fixed point applies, 2: only otherwise, 0: run (fp64).
  // bounded: the caller declares the values to lie in [0, xmax] (rank weights; xmax = max(rX) of the WHOLE matrix, on the
  // device or, when xmax_dev is null, on the host) -- the grid then follows that xmax, the same for every shard of a call;
  // otherwise it follows the swept maximum.  kbits = bits of the largest set size.
  const double* xmax_dev;
  double xmax_host;
  int32_t kbits, bounded;
  const double* sel;
  int32_t sel_want;
  // MED (medians selected on the fly, see the kernel): predicted column means, {offset, half width, ignore-zero} of the
  // calibration, per (column, chunk, wavefront) a slice of `med_capc` candidate scores and the counts {below, zero, NaN,
  // candidates} of what that wavefront wrote of the chunk
  const double* med_pred;
  const double* med_cal;
  unsigned long long* med_cand;
  uint32_t* med_cnt;
  int32_t med_capc;
  unsigned long long* dbg;   // tools/ build: per workgroup, wave 0: cycles in {walk, barrier, epilogue, barrier}
  int32_t abl;               // tools/ build: 1 no LDS atomics | 2 no id loads (synthetic conflict-free ids) | 3 no score stores | 4 = 1 + 2 | 5 no walk
                             // | 6 score stores with agent scope (sc1: the lines do not stay in L2) | 7 plain score stores
                             // | 8 no further segments | 9 no factor loads in the epilogue
};

#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 5
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#else
#define PH_SC_ABL(k) false
#endif
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 6
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#else
#define PH_SC_STAMP(k) do { } while (0)
#define PH_SC_STAMP_IDS(x) do { } while (0)
#endif

__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// FIXED: the accumulators are u64 fixed-point numbers (ds_add_u64: 6.0 cycles per wave-instruction against 8.05 for
// ds_add_f64, tools/ubench/lds_atomics.hip).  For stored values in [0, xmax] the scale 2^e is chosen so that the largest
// possible sum (the largest set size x xmax) stays below 2^63; every value is rounded ONCE to that grid, the integer sums
// are exact, so a score does not depend on the order in which the LDS atomics arrive: bit-reproducible from run to run,
// which the fp64 atomics are not.
// What the rounding costs, rigorously: a stored value v > 0 moves by at most 2^-(e+1) and contributes at least
// min_nz = the smallest stored value > 0 to any sum it is part of (all terms are >= 0), so EVERY score is within
// 2^-(e+1) / min_nz relative of the exact sum.  Fixed point is used only when that bound is <= 2^-40 (9.1e-13), i.e. when
// the dynamic range xmax / min_nz of the input is below ~2^(23 - kbits) (or the column-sum form of it, below); raw counts next to values near 1, one huge outlier,
// xmax >= 2^1000, a negative, NaN or infinite stored value, or (bounded callers) a value above the declared xmax all take
// the fp64 accumulators instead.  The predicate is evaluated on the device by both launches of a call (one of them
// returns at once); nothing about it is left to the caller.
__device__ __forceinline__ bool scatter_fixed_ok(const double* sel, int bounded, const double* xmax_dev, double xmax_host,
                                                 int kbits, int& e_out, double& xmax_out) {
  const double seen_ok = sel[0], seen_max = sel[1], min_nz = sel[2];
  const double xmax = bounded ? ((xmax_dev != nullptr) ? *xmax_dev : xmax_host) : seen_max;
  bool ok = seen_ok >= 0.0 && xmax >= 0.0 && xmax < 0x1p1000 && seen_max <= xmax;   // (false for a NaN xmax)
  int q = 0;
  if (xmax > 0.0 && xmax < 0x1p1000) (void)frexp(xmax, &q);     // xmax < 2^q
  // Bits of the largest possible sum: (largest set size) x xmax.  Where that bound leaves the grid too few bits -- ONE set
  // with thousands of genes next to rank weights up to 2^13 needs 28 bits, e = 35 -- the largest sum of a whole column's
  // stored values takes its place (sel[3], launch_colsum_max: all terms are >= 0, so no set's sum exceeds its column's;
  // the ~1,000 stored values of a cell sum to 2^22).  Only THEN: a call the first bound
  // serves keeps its grid, which depends on nothing but the declared xmax and the collection -- the same on every shard.
  int e = 63 - (q + kbits);
  int qm = 1;
  if (min_nz < INFINITY) (void)frexp(min_nz, &qm);               // min_nz >= 2^(qm - 1)
  bool fine = min_nz == INFINITY || (qm - 1 + e + 1 >= 40);      // 2^-(e+1) / min_nz <= 2^-40
  const double colsum = sel[3];
  if (!fine && colsum > 0.0 && colsum < 0x1p1000) {
    int qs = 0;
    (void)frexp(colsum, &qs);                                    // colsum < 2^qs
    if (qs < q + kbits) {   // sums < 2^(e + qs) = 2^63, the roundings of <= 2^31 values add < 2^30: no u64 wraps
      e = 63 - qs;
      fine = qm - 1 + e + 1 >= 40;
    }
  }
  ok = ok && fine;
  e_out = e;
  xmax_out = xmax;
  return ok;
}
// MED: the launch also CLASSIFIES every score it writes for normalize_medians (R/plaid.R:561-572), so that the column
// medians of a 50,000-set result need no second pass over 40 GB of scores.  The median of a column lies within a few 1e-4
// of (column mean + a collection-wide offset), and the column mean is known BEFORE the crossprod (sum_i x[i, c] u[i],
// colmean_predict_kernel); the offset and its spread are calibrated on the first columns (median_calibrate_kernel).  The
// chunk epilogue counts the scores below the bracket [lo, hi], the exact zeros and the NaN with ballots (scalar counters)
// and appends the 1-5 % of the scores inside the bracket to the (column, chunk) slice of a candidate list; a small kernel
// (median_select_kernel) then picks the two middle order statistics out of <= 4,096 candidates per column -- the same values
// the standalone kernels select, bit for bit -- and columns whose bracket missed, overflowed or met the other
// ignore.zero rule go to the standalone kernel (device-side status, no host round trip).
template <bool FIXED, int BLOCK, bool MED = false>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_num_vgpr(60)))   // v120.. are the prefetch registers (asm)
spmm_scatter_csc_f64(ScatterArgs a) {
  constexpr int NW = BLOCK / 64;   // wavefronts of the workgroup
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* acc = reinterpret_cast<double*>(smem_raw);
  if (a.dense_cells != 0 && ((int64_t)a.Xp[a.n] - a.Xp[0]) * 8 >= a.dense_cells) return;   // the gather kernel takes this input
  int fx_e = 0;
  double fx_xmax = 0.0;
  if (a.sel_want != 0) {   // the other accumulator format takes it
    const bool fixed_ok = scatter_fixed_ok(a.sel, a.bounded, a.xmax_dev, a.xmax_host, a.kbits, fx_e, fx_xmax);
    if ((a.sel_want == 1) != fixed_ok) return;
  }
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    if ((uint32_t)(uintptr_t)((lds_u8*)smem_raw) != 0u) __builtin_trap();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t f = 0;
  double vmin = INFINITY, vamin = INFINITY;   // smallest score / smallest magnitude this lane wrote (NaN skipped)
  uint32_t nnan = 0;                          // NaN scores this lane wrote
  const double alpha = (a.alpha_div != nullptr) ? a.alpha / *a.alpha_div : a.alpha;
  double fx_scale = 1.0, fx_inv = 1.0;
  if constexpr (FIXED) {   // (only ever launched with sel_want == 1: every stored value lies in [0, xmax] and is finite)
    fx_scale = ldexp(1.0, fx_e);
    fx_inv = ldexp(1.0, -fx_e);
  }
  for (int i = tid; i < a.ch + kScatterTrash; i += BLOCK) acc[i] = 0.0;   // (all-zero bits: 0 in either number format)
  __syncthreads();
  const uint32_t* __restrict__ idw = reinterpret_cast<const uint32_t*>(a.ids);   // two u16 ids per lane and load
  const int lane_i = lane;
  const int32_t loff_b = lane * 4;
  // buffer descriptor of the id lists: base, no stride, size in bytes (loads past it return 0), raw dword format
  const i32x4 ids_rsrc = make_raw_rsrc(a.ids, a.ids_bytes);
  // one dword = two u16 accumulator ids per lane -> two ds_add wave-instructions.  No compare and no branch, per lane or
  // per wavefront: padded slots -- a whole empty second instruction included -- hold ids of trash accumulators behind the
  // chunk, on 16 different banks per 16 lanes (geneset.cpp).  (Round 3 marked an empty second instruction with 0xffff and
  // skipped it: a v_readfirstlane, a compare and a branch per (value, chunk) to save an atomic the LDS has room for.)
  // The accumulators start at LDS address 0, so id << 3 IS the LDS address.  (Plain C on purpose: next to an inline-asm
  // statement hipcc stops counting vmcnt and drains every load in flight, which would serialise the double buffer.)
  typedef __attribute__((address_space(3))) double lds_f64;
  typedef __attribute__((address_space(3))) unsigned long long lds_u64;
  // (FIXED: `val` carries the bits of the u64 fixed-point value in a double, see the conversion where v is set)
#define PLAIDHIP_ADD_AT(addr_, val)                                                                              \
  if (!PH_SC_ABL(1)) {                                                                                           \
    if constexpr (FIXED)                                                                                         \
      __hip_atomic_fetch_add(reinterpret_cast<lds_u64*>(static_cast<uintptr_t>(addr_)),                          \
                             (unsigned long long)__double_as_longlong(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    else                                                                                                         \
      __hip_atomic_fetch_add(reinterpret_cast<lds_f64*>(static_cast<uintptr_t>(addr_)), (val), __ATOMIC_RELAXED, \
                             __HIP_MEMORY_SCOPE_WORKGROUP);                                                      \
  }
#define PLAIDHIP_SCATTER2(id2, val)                                                                              \
  {                                                                                                              \
    const double val_ = (val);                                                                                   \
    PLAIDHIP_ADD_AT(off_lo(id2), val_)                                                                           \
    PLAIDHIP_ADD_AT(off_hi(id2), val_)                                                                           \
  }
  // The id segments a wavefront has to apply are a flat work list: lane u holds stored value u of its 64 and the
  // segment range [s0, s1) of that value's gene in the current chunk; "pass" p takes segment s0 + p of every lane
  // that has one.  The list is walked UN segments at a time (wave-uniform bookkeeping in scalar registers), and
  // the loads of the next group are in flight while the LDS atomics of the current one issue: the kernel was
  // bound by the L2 latency of these loads, not by the atomics.
#ifndef PLAIDHIP_SCATTER_UN
#define PLAIDHIP_SCATTER_UN 8
#endif
  constexpr int UN = PLAIDHIP_SCATTER_UN;   // (A/B builds: -DPLAIDHIP_SCATTER_UN=16)
#define PLAIDHIP_FETCH_GROUP(SEGV, VALV, CNT)                                                \
  {                                                                                          \
    CNT = 0;                                                                                 \
    _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                          \
      while (wmask == 0ull && more) {                                                        \
        ++pass;                                                                              \
        wmask = __ballot(ns > pass);                                                         \
        more = wmask != 0ull;                                                                \
      }                                                                                      \
      if (wmask != 0ull) {                                                                   \
        const int src = __builtin_ctzll(wmask);                                              \
        wmask &= wmask - 1ull;                                                               \
        SEGV[u] = __builtin_amdgcn_readlane(s0, src) + pass;                                 \
        VALV[u] = readlane_f64(v, src);                                                      \
        CNT = u + 1;                                                                         \
      } else {                                                                               \
        SEGV[u] = a.dummy_seg;                                                               \
        VALV[u] = 0.0;                                                                       \
      }                                                                                      \
    }                                                                                        \
  }
#define PLAIDHIP_LOAD_GROUP(IDV, SEGV) \
  _Pragma("unroll") for (int u = 0; u < UN; ++u) IDV[u] = idw[(int64_t)SEGV[u] * 64 + lane];
#define PLAIDHIP_APPLY_GROUP(IDV, VALV, CNT) \
  _Pragma("unroll") for (int u = 0; u < UN; ++u) if (u < CNT) PLAIDHIP_SCATTER2(IDV[u], VALV[u])

  // A wavefront applies its 64 stored values one after the other.  Per value the lanes need the value (the data of
  // the atomic) and the first id segment of its gene in this chunk (the address of the id load).  Both are wave-uniform
  // and come out of the owning lane with v_readlane at a CONSTANT lane index (the walk is unrolled over the 64 values):
  // the segment number goes to scalar registers and becomes the scalar base of the id load, the value's two halves come
  // back through two v_mov.  Round 3 staged {segment, count, value} in LDS and read them back with wave-uniform addresses:
  // two broadcast reads per (value, chunk) on the one LDS pipe all sixteen wavefronts share with the atomics themselves
  // (12k of the 65k LDS-active cycles per column, PMC) and ~20 vector instructions per (value, chunk); round 2's first
  // version used v_readlane with a dynamic lane (ballot / ctz bookkeeping, 23 scalar + 12 vector instructions).  Segment 0
  // of every value goes through a static pipeline (48 id loads in flight); the few genes with more than 128 sets in the
  // chunk walk their further segments afterwards (flattened work list).
#define PLAIDHIP_S0_OF(u) ((uint32_t)__builtin_amdgcn_readlane(s0e, (u)))
#define PLAIDHIP_V_OF(u) readlane_f64(v, (u))
#define PLAIDHIP_ID_LOAD(u)                                                                             \
  (PH_SC_ABL(2) ? ((uint32_t)(2 * lane_i) | ((uint32_t)(2 * lane_i + 1) << 16))                          \
                : (uint32_t)raw_buffer_load_i32(ids_rsrc, loff_b, (int32_t)(PLAIDHIP_S0_OF(u) << 8), 0))
#define PLAIDHIP_WALK_SEGMENTS()                                                                       \
  {                                                                                                    \
    const int s0e = ns > 0 ? s0 : a.dummy_seg;                                                         \
    /* 32 .. 48 + HE id loads in flight per wavefront.  Segment 0 of the wavefront's 64 values goes through a static pipeline   \
       (groups A..D, owning lane = a CONSTANT readlane index).  Genes in more than 128 sets of the chunk have FURTHER       \
       segments (~7 per wavefront and item at config 3).  Round 5 walked them behind the pipeline with scalar bookkeeping   \
       per entry (ballot / find-first / dynamic readlane, ~20 dependent scalar instructions and three branches each) and a  \
       load round trip of their own: 21 % of the launch (tools/bench_spmm.py --ablate 108), most of it the scalar code.     \
       Now the second segments are COMPACTED with vector instructions -- the lanes that have one send {segment, value} to   \
       lane rank-among-them by ds_permute_b32 (a full permutation: the other lanes fill up behind) -- so that entry u sits   \
       in lane u like a first segment, and the first HE of them are group E of the same static pipeline, requested right    \
       behind A, B, C.  What is left for the scalar walk behind the pipeline: third and later segments (a gene in more than \
       256 sets of one chunk) and second segments beyond HE per wavefront. */                         \
    constexpr int HW = 16;                                                                             \
    constexpr int HE = PLAIDHIP_SCATTER_HE;                                                            \
    static_assert(HE % 4 == 0 && HE <= 16, "group E is applied four entries at a time");              \
    uint32_t idA[HW], idB[HW], idC[HW], idD[HW], idE[HE];                                              \
    const int nval = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__ballot(have)));            \
    _Pragma("unroll") for (int u = 0; u < HW; ++u) idA[u] = PLAIDHIP_ID_LOAD(u);                       \
    _Pragma("unroll") for (int u = 0; u < HW; ++u) idB[u] = PLAIDHIP_ID_LOAD(HW + u);                  \
    if (PLAIDHIP_SCATTER_DEPTH >= 3) { _Pragma("unroll") for (int u = 0; u < HW; ++u) idC[u] = PLAIDHIP_ID_LOAD(2 * HW + u); } \
    PLAIDHIP_ISSUE_PREFETCH()                                                                          \
    const uint64_t m1 = PH_SC_ABL(8) ? 0ull : __ballot(ns > 1);                                        \
    const int cnt1 = __builtin_popcountll(m1);                                                         \
    int segE = a.dummy_seg;                                                                            \
    double vE = 0.0;                                                                                   \
    {                                                                                                  \
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u)); \
      const int dest = (ns > 1 ? (int)below : cnt1 + lane - (int)below) << 2;                          \
      const int sg_ = __builtin_amdgcn_ds_permute(dest, s0 + 1);                                       \
      const int lo_ = __builtin_amdgcn_ds_permute(dest, __double2loint(v));                            \
      const int hi_ = __builtin_amdgcn_ds_permute(dest, __double2hiint(v));                            \
      if (lane < cnt1) {   /* (lanes behind the compacted entries: the all-padding segment, value 0) */ \
        segE = sg_;                                                                                    \
        vE = __hiloint2double(hi_, lo_);                                                               \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int u = 0; u < HE; ++u)                                                     \
      idE[u] = PH_SC_ABL(2) ? ((uint32_t)(2 * lane_i) | ((uint32_t)(2 * lane_i + 1) << 16))            \
                            : (uint32_t)raw_buffer_load_i32(ids_rsrc, loff_b, (int32_t)((uint32_t)__builtin_amdgcn_readlane(segE, u) << 8), 0); \
    PH_SC_STAMP_IDS(idA[0]);   /* tools/ build: the first ids in hand */                               \
    _Pragma("unroll") for (int u = 0; u < HW; ++u) PLAIDHIP_SCATTER2(idA[u], PLAIDHIP_V_OF(u))         \
    if (PLAIDHIP_SCATTER_DEPTH < 3) { _Pragma("unroll") for (int u = 0; u < HW; ++u) idC[u] = PLAIDHIP_ID_LOAD(2 * HW + u); } \
    else { _Pragma("unroll") for (int u = 0; u < HW; ++u) idD[u] = PLAIDHIP_ID_LOAD(3 * HW + u); }     \
    if (nval > HW) { _Pragma("unroll") for (int u = 0; u < HW; ++u) PLAIDHIP_SCATTER2(idB[u], PLAIDHIP_V_OF(HW + u)) }      \
    if (PLAIDHIP_SCATTER_DEPTH < 3) { _Pragma("unroll") for (int u = 0; u < HW; ++u) idD[u] = PLAIDHIP_ID_LOAD(3 * HW + u); } \
    if (nval > 2 * HW) { _Pragma("unroll") for (int u = 0; u < HW; ++u) PLAIDHIP_SCATTER2(idC[u], PLAIDHIP_V_OF(2 * HW + u)) } \
    if (nval > 3 * HW) { _Pragma("unroll") for (int u = 0; u < HW; ++u) PLAIDHIP_SCATTER2(idD[u], PLAIDHIP_V_OF(3 * HW + u)) } \
    PH_SC_STAMP(6);                                                                                    \
    _Pragma("unroll") for (int q = 0; q < HE / 4; ++q)                                                 \
      if (cnt1 > 4 * q) { _Pragma("unroll") for (int u = 4 * q; u < 4 * q + 4; ++u) PLAIDHIP_SCATTER2(idE[u], readlane_f64(vE, u)) } \
    /* the rest, scalar walk (flattened list, pass-major): second segments beyond HE, then segments 2, 3, ... */ \
    const uint64_t m2 = PH_SC_ABL(8) ? 0ull : __ballot(ns > 2);                                        \
    if (cnt1 > HE || m2 != 0ull) {                                                                     \
      uint64_t wmask = 0ull;                                                                           \
      if (cnt1 > HE) {                                                                                 \
        wmask = m1;                                                                                    \
        for (int u = 0; u < HE; ++u) wmask &= wmask - 1ull;                                            \
      }                                                                                                \
      int pass = 1;                                                                                    \
      bool more = true;                                                                                \
      int sgX[UN], cntX;                                                                               \
      double vX[UN];                                                                                   \
      uint32_t idX[UN];                                                                                \
      do {                                                                                             \
        PLAIDHIP_FETCH_GROUP(sgX, vX, cntX)                                                            \
        PLAIDHIP_LOAD_GROUP(idX, sgX)                                                                  \
        PLAIDHIP_APPLY_GROUP(idX, vX, cntX)                                                            \
      } while (cntX == UN);                                                                            \
    }                                                                                                  \
    PH_SC_STAMP(7);                                                                                    \
  }
// The per-set factors of the chunk's sets -- 18 sets per thread at most -- are requested when the wavefront has applied
// its stored values, BEFORE the barrier that ends the walk: they land while the slower wavefronts finish, and the
// epilogue behind the barrier is LDS reads, arithmetic and stores with no load latency in it (it used to be four to five
// dependent round trips to L2 per chunk, 15 % of the kernel).
  constexpr int kEpiSets = (kScatterChunk + BLOCK - 1) / BLOCK;
#define PLAIDHIP_EPI_PREFETCH()                                                                    \
  f64x2 kwv[kEpiSets];                                                                              \
  {                                                                                                 \
    int jb_p = jb;                                                                                  \
    asm volatile("" : "+v"(jb_p));                                                                  \
    _Pragma("unroll") for (int u = 0; u < kEpiSets; ++u) {                                          \
      const int j = jb_p + u * jstep;   /* the set of slot tid + u * BLOCK of this chunk (common.h: blocks dealt to the chunks) */ \
      kwv[u] = PH_SC_ABL(9) ? f64x2{1.0, 0.5} : a.kw[j < a.m ? j : a.m - 1];                        \
    }                                                                                               \
  }
#define PLAIDHIP_EPI_ONE(u)                                                                        \
  {                                                                                                 \
    const int i = tid_e + (u) * BLOCK;                                                              \
    const int j = jb_e + (u) * jstep;                                                               \
    double val = __longlong_as_double(0x7ff8000000000000ll);   /* (lanes without a set: fails every compare below) */ \
    if (j < a.m) {                                                                                  \
      double sum = acc[i];                                                                          \
      if constexpr (FIXED) {   /* u64 -> double, one rounding: hi * 2^32 + lo as a single fma */             \
        const unsigned long long b_ = (unsigned long long)__double_as_longlong(sum);                \
        sum = __fma_rn((double)(uint32_t)(b_ >> 32), 4294967296.0, (double)(uint32_t)b_) * fx_inv;  \
      }                                                                                             \
      acc[i] = 0.0;                                                                                 \
      val = alpha * (sum * kwv[u].y) + a.beta * kwv[u].x;                                           \
      if (PH_SC_ABL(6)) __hip_atomic_store(&a.S[(int64_t)c * a.lds + j], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* sc1 */ \
      else if (PH_SC_ABL(7)) a.S[(int64_t)c * a.lds + j] = val;                                     \
      else if (!PH_SC_ABL(3)) __builtin_nontemporal_store(val, &a.S[(int64_t)c * a.lds + j]);       \
      /* the three flags, cheaply (the epilogue is bound by its vector instructions): the smallest score and the */  \
      /* smallest magnitude by v_min_f64 (which skips NaN), NaN by one compare counted into a lane counter */        \
      vmin = min_f64(vmin, val);                                                                    \
      vamin = min_abs_f64(vamin, val);                                                              \
      nnan += (val != val) ? 1u : 0u;                                                               \
    }                                                                                               \
    if constexpr (MED) {                                                                            \
      /* (converged again) three compares per score, as ballots -- scalar counters: below the bracket, not above it, */ \
      /* exactly zero.  A NaN score fails all three like a lane without a set; whether the wavefront wrote any NaN */   \
      /* comes from its lane counters at the end of the item (such a column is left to the standalone kernel).     */   \
      const uint64_t lt_ = __ballot(val < med_lo);                                                  \
      const uint64_t le_ = __ballot(val <= med_hi);                                                 \
      const uint64_t zero_ = __ballot(val == 0.0);                                                  \
      const uint64_t in_ = (le_ & ~lt_) & (med_iz ? ~zero_ : ~0ull);                                \
      w_lt += (uint32_t)__popcll(lt_);                                                              \
      w_zero += (uint32_t)__popcll(zero_);                                                          \
      if (in_ != 0ull) {   /* (wave-uniform) append to this wavefront's own slice: no shared counter, no atomic */ \
        const uint32_t slot_ = w_cand + __builtin_amdgcn_mbcnt_hi((uint32_t)(in_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)in_, 0u)); \
        if (__builtin_amdgcn_inverse_ballot_w64(in_) && slot_ < (uint32_t)a.med_capc)                \
          med_slice[slot_] = (unsigned long long)__double_as_longlong(val);                         \
        w_cand += (uint32_t)__popcll(in_);                                                          \
      }                                                                                             \
    }                                                                                               \
  }
#define PLAIDHIP_CHUNK_EPILOGUE()                                                                  \
  {                                                                                                 \
    /* all factors are awaited before the first store: one whose use is skipped (no set in the slot) would stay "pending" in */ \
    /* hipcc's bookkeeping, and the next write to its register would wait with vmcnt(0) -- for the stores' acks */   \
    _Pragma("unroll") for (int u = 0; u < kEpiSets; ++u) asm volatile("" : "+v"(kwv[u]));          \
    int tid_e = tid, jb_e = jb;                                                                     \
    asm volatile("" : "+v"(tid_e), "+v"(jb_e));                                                     \
    _Pragma("unroll") for (int u = 0; u < kEpiSets; ++u) {                                          \
      PLAIDHIP_EPI_ONE(u)                                                                           \
    }                                                                                               \
    if constexpr (MED) {   /* the wavefront's counts of this (column, chunk) */                     \
      const uint32_t w_nan = (uint32_t)__popcll(__ballot(nnan != nnan0));   /* (lanes, not scores: only "any" matters) */ \
      if (lane == 0) {                                                                              \
        uint4* o_ = reinterpret_cast<uint4*>(a.med_cnt) + (((int64_t)c * a.nch + chunk) * NW + wave); \
        /* {scores below the bracket that take part, exact zeros, NaN, candidates}: with the zeros masked (ignore.zero) */ \
        /* a zero below the bracket does not count                                                                     */ \
        *o_ = make_uint4((med_iz && med_lo > 0.0) ? w_lt - w_zero : w_lt, w_zero, w_nan, w_cand);  \
      }                                                                                             \
    }                                                                                               \
  }

  // The work of a workgroup is a sequence of ITEMS (column, chunk of sets, round): round r of a column is its stored
  // values [1024 r, 1024 (r + 1)), one per thread; all rounds of a (column, chunk) add into the same accumulators, then
  // the chunk's epilogue runs.  What an item needs from memory -- the thread's (gene, value) and the segment range of
  // that gene in the item's chunk, the second depending on the first -- is requested TWO and ONE items ahead, while an
  // earlier item is being applied, across chunk and column boundaries alike; any number of stored values per column
  // runs the same pipeline.  (Round 2 of the build had a pipelined path for columns of <= 1,024 values and a plain
  // loop for longer ones: with ~1,000 +- 30 values per cell a fifth of the columns took the plain loop and a quarter
  // of the kernel's time.)
  // These prefetches are issued through inline asm into v120..v125 -- registers outside the compiler's range
  // (amdgpu_num_vgpr on the kernel) -- and fetched with plain moves behind a wait placed by hand at the end of the
  // walk, where every load has been consumed anyway.  As ordinary loads they cost stalls in every item: hipcc copied
  // loop-carried results right behind the load instruction (a full L2 round trip before the walk had started), and
  // at the top of the next item it waited with vmcnt(0) -- it cannot count the conditional S stores issued since --
  // i.e. for the acknowledgement of a whole chunk's stores.
#define PLAIDHIP_ASM_LOAD_SEG(ptr_)   /* {seg[gene], seg[gene + 1]} -> v[120:121] */                    \
  asm volatile("global_load_dwordx2 v[120:121], %0, off" : : "v"(ptr_) : "memory")
#define PLAIDHIP_ASM_LOAD_CELL(pi_, px_)   /* Xi[q] -> v122, Xx[q] -> v[124:125] */                     \
  asm volatile("global_load_dword v122, %0, off\n\tglobal_load_dwordx2 v[124:125], %1, off"            \
               : : "v"(pi_), "v"(px_) : "memory")
#define PLAIDHIP_ASM_TAKE(s0_, s1_, g_, xlo_, xhi_)                                                     \
  asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v120\n\tv_mov_b32 %1, v121\n\tv_mov_b32 %2, v122\n\t" \
               "v_mov_b32 %3, v124\n\tv_mov_b32 %4, v125"                                               \
               : "=v"(s0_), "=v"(s1_), "=v"(g_), "=v"(xlo_), "=v"(xhi_) : : "memory")
// item iterator (wave-uniform): next round of the column, else next chunk, else next column of this workgroup.
// A column of nnz stored values takes nr = ceil(nnz / 1024) rounds of per = ceil(nnz / nr) values (balanced rounds:
// 1,100 values are 2 x 550, not 1,024 + 76); inside a round the values are dealt to the wavefronts in groups of 16
// (group i of the round goes to wavefront i mod 16), so every wavefront applies the same number of groups +- 1.
#define PLAIDHIP_ITEM_ROUNDS(q0_, q1_) (((q1_) - (q0_) + BLOCK - 1) / BLOCK > 1 ? ((q1_) - (q0_) + BLOCK - 1) / BLOCK : 1)
#define PLAIDHIP_ITEM_NEXT(c_, ch_, r_, q0_, q1_)                                                       \
  if (c_ < a.n) {                                                                                       \
    if (r_ + 1 < PLAIDHIP_ITEM_ROUNDS(q0_, q1_)) {                                                      \
      ++r_;                                                                                             \
    } else if (a.chunk_major) { /* next column of this workgroup; behind the last one, its first column in the next chunk */ \
      r_ = 0;                                                                                           \
      c_ += gridDim.x;                                                                                  \
      if (c_ >= a.n) { ++ch_; c_ = (ch_ < a.nch) ? (int)blockIdx.x : a.n; }                             \
      if (c_ < a.n) { q0_ = ((cptr_i32)a.Xp)[c_]; q1_ = ((cptr_i32)a.Xp)[c_ + 1]; }                     \
    } else {                                                                                            \
      r_ = 0;                                                                                           \
      if (ch_ + 1 < a.nch) {                                                                            \
        ++ch_;                                                                                          \
      } else {                                                                                          \
        ch_ = 0;                                                                                        \
        c_ += gridDim.x;                                                                                \
        if (c_ < a.n) { q0_ = ((cptr_i32)a.Xp)[c_]; q1_ = ((cptr_i32)a.Xp)[c_ + 1]; }                   \
      }                                                                                                 \
    }                                                                                                   \
  }
// the thread's stored value in round r_ of column [q0_, q1_): index (clamped into the column) and whether it exists
#define PLAIDHIP_ITEM_MINE(r_, q0_, q1_, qi_, have_)                                                    \
  {                                                                                                     \
    const int nnz_ = (q1_) - (q0_);                                                                     \
    const int nr_ = PLAIDHIP_ITEM_ROUNDS(q0_, q1_);                                                     \
    const int per_ = nr_ == 1 ? nnz_ : (nnz_ + nr_ - 1) / nr_;                                          \
    const int lo_ = (r_) * per_;                                                                        \
    const int cnt_ = nnz_ - lo_ < per_ ? nnz_ - lo_ : per_;                                             \
    const int i_ = (((lane >> 4) * NW + wave) << 4) + (lane & 15);                                      \
    have_ = i_ < cnt_;                                                                                  \
    qi_ = (q0_) + (have_ ? lo_ + i_ : 0);                                                               \
  }
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 7
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
  // current item, the next one (n1: gene / value here, segment range on its way during the current walk) and the one
  // after (n2: gene / value on their way)
  asm volatile("" : : : "v125");   // (the one clobber that makes the kernel's register count cover v120..v125)
  int c = blockIdx.x, chunk = 0, rr = 0, q0 = 0, q1 = 0;
  if (c < a.n) { q0 = ((cptr_i32)a.Xp)[c]; q1 = ((cptr_i32)a.Xp)[c + 1]; }
  int c1 = c, chunk1 = chunk, r1 = rr, q01 = q0, q11 = q1;
  PLAIDHIP_ITEM_NEXT(c1, chunk1, r1, q01, q11)
  int c2 = c1, chunk2 = chunk1, r2 = r1, q02 = q01, q12 = q11;
  PLAIDHIP_ITEM_NEXT(c2, chunk2, r2, q02, q12)
  int gene = 0, s0 = 0, s1 = 0, gene1 = 0;
  double v = 0.0, v1 = 0.0;
  if (c < a.n && q1 > q0) {
    int qi; bool hv;
    PLAIDHIP_ITEM_MINE(rr, q0, q1, qi, hv)
    gene = a.Xi[qi];
    v = a.Xx[qi];
    s0 = a.seg[gene];
    s1 = a.seg[gene + 1];
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(gene), "+v"(v));   // (awaited here, once)
  }
  if (c1 < a.n && q11 > q01) {
    int qi; bool hv;
    PLAIDHIP_ITEM_MINE(r1, q01, q11, qi, hv)
    gene1 = a.Xi[qi];
    v1 = a.Xx[qi];
    asm volatile("" : "+v"(gene1), "+v"(v1));
  }
  while (c < a.n) {
    bool have; int qi_cur;
    PLAIDHIP_ITEM_MINE(rr, q0, q1, qi_cur, have)
    (void)qi_cur;
    const int ns = have ? s1 - s0 : 0;
    if (!have) v = 0.0;
    if constexpr (FIXED)   // one rounding to the fixed-point grid
      v = __longlong_as_double((long long)__double2ull_rn(v * fx_scale));
    const bool n1_loads = c1 < a.n, n2_loads = c2 < a.n && q12 > q02;
    // requests for the items behind this one: BEHIND the walk's first 48 id loads (a wavefront's loads return in order, and two
    // of these miss to HBM: in front of the id loads they kept every wavefront from its first ids for 9k cycles per item)
#define PLAIDHIP_ISSUE_PREFETCH()                                                        \
    {                                                                                    \
      if (n1_loads) PLAIDHIP_ASM_LOAD_SEG(a.seg + (int64_t)chunk1 * a.g + gene1);        \
      if (n2_loads) {                                                                    \
        int qi; bool hv;                                                                 \
        PLAIDHIP_ITEM_MINE(r2, q02, q12, qi, hv)                                         \
        PLAIDHIP_ASM_LOAD_CELL(a.Xi + qi, a.Xx + qi);                                    \
      }                                                                                  \
    }
    PH_SC_STAMP(0);
    if (__ballot(have) != 0ull && !PH_SC_ABL(5)) {   // (a wavefront without a value in this round has nothing to apply)
      PLAIDHIP_WALK_SEGMENTS()
    } else {
      PLAIDHIP_ISSUE_PREFETCH()
    }
#undef PLAIDHIP_ISSUE_PREFETCH
    int s0_1, s1_1, gene2, xlo, xhi;
    PLAIDHIP_ASM_TAKE(s0_1, s1_1, gene2, xlo, xhi);   // (the walk has consumed every load of its own: nothing else to wait for)
    PH_SC_STAMP(1);
    if (rr + 1 >= PLAIDHIP_ITEM_ROUNDS(q0, q1)) {   // last round of this (column, chunk): scale and write the chunk's scores
      const int jb = chunk * BLOCK + tid;   // the set of this thread's slot `tid`; slot tid + u * BLOCK holds set jb + u * jstep
      const int jstep = a.nch * BLOCK;
      double med_lo = 0.0, med_hi = 0.0;
      bool med_iz = false;
      unsigned long long* med_slice = nullptr;
      uint32_t w_lt = 0, w_zero = 0, w_cand = 0;
      const uint32_t nnan0 = nnan;
      if constexpr (MED) {
        const double ctr = a.med_pred[c] + a.med_cal[0];
        med_lo = ctr - a.med_cal[1];
        med_hi = ctr + a.med_cal[1];
        med_iz = a.med_cal[2] != 0.0;
        med_slice = a.med_cand + (((int64_t)c * a.nch + chunk) * NW + wave) * a.med_capc;
      }
      PLAIDHIP_EPI_PREFETCH()
      __syncthreads();
      PH_SC_STAMP(2);
      PLAIDHIP_CHUNK_EPILOGUE()
      PH_SC_STAMP(3);
      __syncthreads();
      PH_SC_STAMP(4);
    }
    // (otherwise the next round stages into the same per-wave slots: the LDS operations of a wavefront are in order
    // and only the wavefront itself reads its slots)
    // rotate the pipeline
    c = c1; chunk = chunk1; rr = r1; q0 = q01; q1 = q11;
    gene = gene1; v = v1; s0 = n1_loads ? s0_1 : 0; s1 = n1_loads ? s1_1 : 0;
    c1 = c2; chunk1 = chunk2; r1 = r2; q01 = q02; q11 = q12;
    gene1 = n2_loads ? gene2 : 0;
    v1 = n2_loads ? __hiloint2double(xhi, xlo) : 0.0;
    PLAIDHIP_ITEM_NEXT(c2, chunk2, r2, q02, q12)
  }
#undef PLAIDHIP_ITEM_NEXT
#undef PLAIDHIP_ITEM_MINE
#undef PLAIDHIP_ITEM_ROUNDS
#undef PLAIDHIP_ASM_TAKE
#undef PLAIDHIP_ASM_LOAD_SEG
#undef PLAIDHIP_ASM_LOAD_CELL
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 8
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
#undef PLAIDHIP_WALK_SEGMENTS
#undef PLAIDHIP_S0_OF
#undef PLAIDHIP_V_OF
#undef PLAIDHIP_ID_LOAD
#undef PLAIDHIP_CHUNK_EPILOGUE
#undef PLAIDHIP_EPI_ONE
#undef PLAIDHIP_EPI_PREFETCH
  f |= (vmin < 0.0 ? PLAIDHIP_FLAG_HAS_NEG : 0u) | (vamin == 0.0 ? PLAIDHIP_FLAG_HAS_ZERO : 0u) | (nnan ? PLAIDHIP_FLAG_HAS_NAN : 0u);
  publish_flags(f, a.flags);
#undef PLAIDHIP_ADD_AT
#undef PLAIDHIP_SCATTER2
#undef PLAIDHIP_FETCH_GROUP
#undef PLAIDHIP_LOAD_GROUP
#undef PLAIDHIP_APPLY_GROUP
}

// how a sparse X is multiplied: plaidhip_set_option(PLAIDHIP_OPT_SPMM_SPARSE_KERNEL): 0 auto (scatter below
// 12.5 % stored values, decided on the device from nnz(X)), 1 scatter, 2 gather
static int sparse_mode(const plaidhip_ctx* ctx) { return ctx->opt_sparse_kernel; }

int launch_spmm_scatter_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp,
                                const int32_t* Xi, const double* Xx, int32_t n, int stat, double alpha,
                                const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags,
                                bool auto_select, bool bounded, const double* xmax_dev, double xmax_host, int64_t nnz,
                                const plaidhip_scatter_med* med) {
  const plaidhip_scatter_plan& sp = gs->scatter;
  ScatterArgs a{};
  // One sweep over the stored values ({all finite and >= 0, max, smallest > 0}; its range comes from Xp on the device, `nnz`
  // only sizes its grid) decides ON THE DEVICE between the fixed-point and the fp64 accumulators (scatter_fixed_ok): both
  // launches are enqueued and the one that does not apply returns at once.  Callers that declare the values bounded (rank
  // weights) get the grid of THEIR xmax -- the same on every shard of a sharded call -- but the sweep still decides
  // whether fixed point is safe (a NaN rank weight, e.g., takes the fp64 kernel and propagates as in the reference).
  const bool try_fixed = ctx->opt_scatter_fixed != 0 && ctx->d_sel != nullptr;
  if (try_fixed) {
    int rc = launch_nonneg_range(ctx, Xx, Xp, n, nnz, ctx->d_sel);
    if (rc != PLAIDHIP_OK) return rc;
    // a collection with a set of more than 1,024 genes: the bound (set size) x xmax may leave the grid too few bits where
    // the largest column sum does not (scatter_fixed_ok) -- one more pass over the stored values (~0.2 ms per 1e8)
    if (sp.kbits > 10) {
      rc = launch_colsum_max(ctx, Xx, Xp, n, ctx->d_sel + 3);
      if (rc != PLAIDHIP_OK) return rc;
    }
  }
  a.chunk_major = ctx->opt_scatter_order;
  a.xmax_dev = xmax_dev;
  a.xmax_host = xmax_host;
  a.kbits = sp.kbits;
  a.bounded = bounded ? 1 : 0;
  a.Xp = Xp;
  a.Xi = Xi;
  a.Xx = Xx;
  a.n = n;
  a.m = gs->m;
  a.g = gs->g;
  a.ch = sp.ch;
  a.nch = sp.nch;
  a.dense_cells = auto_select ? (int64_t)gs->g * n : 0;
  a.seg = sp.d_seg;
  a.ids = sp.d_ids;
  a.ids_bytes = (uint32_t)std::min<uint64_t>(((uint64_t)sp.nseg + 1) * 256, 0xffffffffull);
  a.dummy_seg = (int32_t)sp.nseg;
  a.w = sp.d_w;
  a.k = sp.d_k;
  a.kw = reinterpret_cast<const f64x2*>(sp.d_kw) + (stat == PLAIDHIP_STAT_MEAN ? 0 : gs->m);
  a.stat = stat;
  a.alpha = alpha;
  a.beta = beta;
  a.alpha_div = alpha_div;
  a.S = S;
  a.lds = lds;
  a.flags = flags;
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 9
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
  const size_t smem = (size_t)(sp.ch + kScatterTrash) * sizeof(double);
  {
    // the kernel reserves v120..v125 by hand (amdgpu_num_vgpr + one clobber): if a toolchain ever sized its register
    // file differently it could not be launched with 1,024 threads -- say so here instead of failing at the launch
    static std::atomic<int> checked{0};
    if (checked.load(std::memory_order_acquire) == 0) {
      hipFuncAttributes fa{};
      hipFuncAttributes fb{};
      PH_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&spmm_scatter_csc_f64<false, kScatterBlock>)));
      PH_HIP(hipFuncGetAttributes(&fb, reinterpret_cast<const void*>(&spmm_scatter_csc_f64<true, kScatterBlock>)));
      if (fb.numRegs > fa.numRegs) fa.numRegs = fb.numRegs;
      if (fb.maxThreadsPerBlock < fa.maxThreadsPerBlock) fa.maxThreadsPerBlock = fb.maxThreadsPerBlock;
      if (fa.maxThreadsPerBlock < kScatterBlock || fa.numRegs > 128) {
        set_error("spmm_scatter_csc_f64 was built with %d registers (max %d threads per workgroup): it needs <= 128 (sixteen "
                  "wavefronts per CU); rebuild with the toolchain of the Makefile", fa.numRegs, fa.maxThreadsPerBlock);
        return PLAIDHIP_EUNSUPPORTED;
      }
      checked.store(1, std::memory_order_release);
    }
  }
  PH_FULL_LDS(ctx, (&spmm_scatter_csc_f64<false, kScatterBlock>));
  PH_FULL_LDS(ctx, (&spmm_scatter_csc_f64<true, kScatterBlock>));
  // TWO workgroups of 512 threads per CU, each with half the LDS: while one is in its barriers and chunk epilogue (45 % of
  // an item with one 1,024-thread workgroup per CU: tools/bench_spmm.py --kernel c3 --ablate 105) the other one's walk
  // keeps the LDS atomic unit busy
  int grid = ctx->num_cu * (1024 / kScatterBlock);
  if (grid > n) grid = n;
  if (med != nullptr) {   // medians selected on the fly (launch_spmm_csc_fused_f64)
    a.med_pred = med->pred;
    a.med_cal = med->cal;
    a.med_cand = med->cand;
    a.med_cnt = med->cnt;
    a.med_capc = med->capc;
    PH_FULL_LDS(ctx, (&spmm_scatter_csc_f64<false, kScatterBlock, true>));
    PH_FULL_LDS(ctx, (&spmm_scatter_csc_f64<true, kScatterBlock, true>));
    if (try_fixed) {
      a.sel = ctx->d_sel;
      a.sel_want = 1;
      hipLaunchKernelGGL((spmm_scatter_csc_f64<true, kScatterBlock, true>), dim3(grid), dim3(kScatterBlock), smem, ctx->stream, a);
      a.sel_want = 2;
      hipLaunchKernelGGL((spmm_scatter_csc_f64<false, kScatterBlock, true>), dim3(grid), dim3(kScatterBlock), smem, ctx->stream, a);
    } else {
      hipLaunchKernelGGL((spmm_scatter_csc_f64<false, kScatterBlock, true>), dim3(grid), dim3(kScatterBlock), smem, ctx->stream, a);
    }
    PH_HIP(hipGetLastError());
    return PLAIDHIP_OK;
  }
  if (try_fixed) {
    a.sel = ctx->d_sel;
    a.sel_want = 1;
    hipLaunchKernelGGL((spmm_scatter_csc_f64<true, kScatterBlock>), dim3(grid), dim3(kScatterBlock), smem, ctx->stream, a);
    a.sel_want = 2;
    hipLaunchKernelGGL((spmm_scatter_csc_f64<false, kScatterBlock>), dim3(grid), dim3(kScatterBlock), smem, ctx->stream, a);
  } else {
    hipLaunchKernelGGL((spmm_scatter_csc_f64<false, kScatterBlock>), dim3(grid), dim3(kScatterBlock), smem, ctx->stream, a);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}


// ---------------------------------------------------------------------------------------------
// Mixed-precision pair kernel (opt-in, plaidhip_set_precision(ctx, PLAIDHIP_PRECISION_MIXED)):
// the two sample columns of a pair are staged in LDS as FLOATS, 8 bytes {A_i, B_i} per gene, so the
// whole 20k-gene pair fits the LDS (no gene slices, no partial sums) and one ds_read_b64 serves two
// scores -- a quarter of the LDS bytes per score of the fp64 one-column kernel.  The eight values a
// lane gathers per chunk are summed with packed fp32 adds (four terms per partial sum) and the
// chunk's partials are added to fp64 accumulators, so the only precision given up is the rounding of
// the inputs to fp32 (2^-24 relative per value, ~6e-8 on the scores; the fp64 kernels stay the
// default and the parity reference).  Same host schedule as the one-column kernel (32-lane halves
// of ds_read_b64 against 32 bank pairs), same per-wave tile streams and epilogue.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x2 lds_cf32x2;
__device__ __forceinline__ f32x2 lds_f32x2_at(uint32_t byte_off) {
  return *reinterpret_cast<lds_cf32x2*>(static_cast<uintptr_t>(byte_off));
}

template <bool STAMP>
__global__ void __launch_bounds__(1024)
spmm_colpair_mixed(SpmmArgs a) {
  constexpr int BLOCK = 1024;
  unsigned long long t_stage = 0, t_gather = 0, t_wait = 0, t_all0 = 0;
  if constexpr (STAMP) t_all0 = __builtin_amdgcn_s_memtime();
  extern __shared__ __align__(16) unsigned char smem_raw[];
  f32x2* ent = reinterpret_cast<f32x2*>(smem_raw);
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    if ((uint32_t)(uintptr_t)((lds_u8*)smem_raw) != 0u) __builtin_trap();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t f = 0;
  const double alpha = (a.alpha_div != nullptr) ? a.alpha / *a.alpha_div : a.alpha;
  const bool is_mean = a.stat == PLAIDHIP_STAT_MEAN;
  const int ch_begin = __builtin_amdgcn_readfirstlane(a.wave_chunk_off[wave]);
  const int ch_end = __builtin_amdgcn_readfirstlane(a.wave_chunk_off[wave + 1]);
  const int tk_begin = __builtin_amdgcn_readfirstlane(a.wave_tile_off[wave]);
  const int g2 = a.g >> 1;
  const int npairs = (a.n + 1) >> 1;
  // next pair, as loaded: 10 x 16 bytes of column A and of column B per thread
  f64x2 pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7, pa8, pa9, pb0, pb1, pb2, pb3, pb4, pb5, pb6, pb7, pb8, pb9;
  pa0 = pa1 = pa2 = pa3 = pa4 = pa5 = pa6 = pa7 = pa8 = pa9 = f64x2{0.0, 0.0};
  pb0 = pb1 = pb2 = pb3 = pb4 = pb5 = pb6 = pb7 = pb8 = pb9 = f64x2{0.0, 0.0};
#define PLAIDHIP_ITEMS10(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)
#define PLAIDHIP_PF_ONE(k)                                                                          \
  if ((k + 1) * BLOCK <= g2) {                                                                      \
    pa##k = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xa_ + (size_t)k * BLOCK * 16 + lane_off16)); \
    pb##k = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xb_ + (size_t)k * BLOCK * 16 + lane_off16)); \
  } else {                                                                                          \
    pa##k = pb##k = f64x2{0.0, 0.0};                                                                \
    if (k * BLOCK < g2 && tid + k * BLOCK < g2) {                                                   \
      pa##k = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xa_ + (size_t)k * BLOCK * 16 + lane_off16)); \
      pb##k = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xb_ + (size_t)k * BLOCK * 16 + lane_off16)); \
    }                                                                                               \
  }
#define PLAIDHIP_PREFETCH(pp_)                                                                      \
  do {                                                                                              \
    uint32_t lane_off16 = (uint32_t)tid * 16u;                                                      \
    asm volatile("" : "+v"(lane_off16));                                                            \
    const int ca_ = 2 * (pp_);                                                                      \
    const int cb_ = (ca_ + 1 < a.n) ? ca_ + 1 : ca_;                                                \
    const char* xa_ = reinterpret_cast<const char*>(a.X + (int64_t)ca_ * a.ldx);                    \
    const char* xb_ = reinterpret_cast<const char*>(a.X + (int64_t)cb_ * a.ldx);                    \
    PLAIDHIP_ITEMS10(PLAIDHIP_PF_ONE)                                                               \
  } while (0)
#define PLAIDHIP_ST_ONE(k)                                                                          \
  if (k * BLOCK < g2) {                                                                             \
    const int i_ = tid_o + k * BLOCK;                                                               \
    if (i_ < g2)                                                                                    \
      ent4[i_] = f32x4{(float)pa##k.x, (float)pb##k.x, (float)pa##k.y, (float)pb##k.y};             \
  }

  int p = blockIdx.x;
  if (p < npairs) PLAIDHIP_PREFETCH(p);
  for (; p < npairs; p += gridDim.x) {
    const int cA = 2 * p;
    const bool hasB = cA + 1 < a.n;
    const int cB = hasB ? cA + 1 : cA;
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if constexpr (STAMP) ts0 = __builtin_amdgcn_s_memtime();
    {
      f32x4* ent4 = reinterpret_cast<f32x4*>(smem_raw);
      PLAIDHIP_ITEMS10(PLAIDHIP_ST_ONE)
      if ((a.g & 1) && tid == 0)
        ent[a.g - 1] = f32x2{(float)a.X[(int64_t)cA * a.ldx + a.g - 1], (float)a.X[(int64_t)cB * a.ldx + a.g - 1]};
      if (tid < kPadSlots) ent[a.g + tid] = f32x2{0.0f, 0.0f};
    }
    __syncthreads();
    if constexpr (STAMP) ts1 = __builtin_amdgcn_s_memtime();
    const int np = p + gridDim.x;
    const bool want_pf = np < npairs;

    if (ch_begin < ch_end) {
      const char* ibase = reinterpret_cast<const char*>(a.tile_idx) + (int64_t)ch_begin * 1024;  // uniform
      uint32_t lane_o = (uint32_t)lane;
      asm volatile("" : "+v"(lane_o));
      const uint32_t ioff = lane_o * 16u;
      const uint32_t moff4 = lane_o * 4u, moff8 = lane_o * 8u;
#define PLAIDHIP_LOADQ(rel) (*reinterpret_cast<const uint4*>(ibase + (int64_t)(rel) * 1024 + ioff))
#define PLAIDHIP_GATHER8(q)                                                  \
  v0 = lds_f32x2_at(off_lo((q).x)); v1 = lds_f32x2_at(off_hi((q).x));         \
  v2 = lds_f32x2_at(off_lo((q).y)); v3 = lds_f32x2_at(off_hi((q).y));         \
  v4 = lds_f32x2_at(off_lo((q).z)); v5 = lds_f32x2_at(off_hi((q).z));         \
  v6 = lds_f32x2_at(off_lo((q).w)); v7 = lds_f32x2_at(off_hi((q).w));
      // packed fp32 partial sums of four values each, folded into the fp64 accumulators per chunk
#define PLAIDHIP_ADD8                                                        \
  {                                                                          \
    const f32x2 s0_ = (v0 + v1) + (v2 + v3);                                 \
    const f32x2 s1_ = (v4 + v5) + (v6 + v7);                                 \
    dA0 += (double)s0_.x; dB0 += (double)s0_.y;                              \
    dA1 += (double)s1_.x; dB1 += (double)s1_.y;                              \
  }
#define PLAIDHIP_EPI(sum, cc)                                                  \
  {                                                                            \
    const double w_ = is_mean ? mw : 1.0;                                      \
    const double v_ = alpha * ((sum) * w_) + a.beta * (mk * w_);               \
    if (a.nt_store) __builtin_nontemporal_store(v_, &a.S[(int64_t)(cc) * a.lds + mj]);  \
    else a.S[(int64_t)(cc) * a.lds + mj] = v_;                                 \
    f |= (v_ < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;                              \
    f |= (v_ == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;                            \
    f |= (v_ != v_) ? PLAIDHIP_FLAG_HAS_NAN : 0u;                              \
  }
#define PLAIDHIP_TILE_END(chv)                                                                 \
  if ((chv) + 1 == next_end) { /* wave-uniform: tile finished -> epilogue */                   \
    if (mj >= 0) {                                                                             \
      PLAIDHIP_EPI(dA0 + dA1, cA)                                                              \
      if (hasB) PLAIDHIP_EPI(dB0 + dB1, cB)                                                    \
    }                                                                                          \
    ++k;                                                                                       \
    next_end = __builtin_amdgcn_readfirstlane(a.wtile_end[k]);                                 \
    mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);  \
    mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);   \
    mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);   \
    dA0 = dA1 = dB0 = dB1 = 0.0;                                                               \
  }
      uint4 qa = PLAIDHIP_LOADQ(0);
      uint4 qb = PLAIDHIP_LOADQ(1);
      uint4 qc = PLAIDHIP_LOADQ(2);
      uint4 qd = PLAIDHIP_LOADQ(3);
      int k = tk_begin;
      int next_end = __builtin_amdgcn_readfirstlane(a.wtile_end[k]);
      double dA0 = 0.0, dA1 = 0.0, dB0 = 0.0, dB1 = 0.0;
      int mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);
      double mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);
      double mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);
      f32x2 v0, v1, v2, v3, v4, v5, v6, v7;
      int ch = ch_begin;
      ibase += 4 * 1024;
      for (; ch + 3 < ch_end; ch += 4, ibase += 4 * 1024) {
        PLAIDHIP_GATHER8(qa)
        qa = PLAIDHIP_LOADQ(0);
        PLAIDHIP_ADD8
        PLAIDHIP_TILE_END(ch)
        PLAIDHIP_GATHER8(qb)
        qb = PLAIDHIP_LOADQ(1);
        PLAIDHIP_ADD8
        PLAIDHIP_TILE_END(ch + 1)
        PLAIDHIP_GATHER8(qc)
        qc = PLAIDHIP_LOADQ(2);
        PLAIDHIP_ADD8
        PLAIDHIP_TILE_END(ch + 2)
        PLAIDHIP_GATHER8(qd)
        qd = PLAIDHIP_LOADQ(3);
        PLAIDHIP_ADD8
        PLAIDHIP_TILE_END(ch + 3)
      }
      if (ch < ch_end) { PLAIDHIP_GATHER8(qa) PLAIDHIP_ADD8 PLAIDHIP_TILE_END(ch) ++ch; }
      if (ch < ch_end) { PLAIDHIP_GATHER8(qb) PLAIDHIP_ADD8 PLAIDHIP_TILE_END(ch) ++ch; }
      if (ch < ch_end) { PLAIDHIP_GATHER8(qc) PLAIDHIP_ADD8 PLAIDHIP_TILE_END(ch) ++ch; }
#undef PLAIDHIP_GATHER8
#undef PLAIDHIP_ADD8
#undef PLAIDHIP_TILE_END
#undef PLAIDHIP_EPI
#undef PLAIDHIP_LOADQ
    }
    if (want_pf) {
      PLAIDHIP_PREFETCH(np);
    } else {
      pa0 = pa1 = pa2 = pa3 = pa4 = pa5 = pa6 = pa7 = pa8 = pa9 = f64x2{0.0, 0.0};
      pb0 = pb1 = pb2 = pb3 = pb4 = pb5 = pb6 = pb7 = pb8 = pb9 = f64x2{0.0, 0.0};
    }
    if constexpr (STAMP) ts2 = __builtin_amdgcn_s_memtime();
    __syncthreads();  // the pair is overwritten by the next iteration
    if constexpr (STAMP) {
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      t_stage += ts1 - ts0;
      t_gather += ts2 - ts1;
      t_wait += ts3 - ts2;
    }
  }
  if constexpr (STAMP) {
    if (lane == 0 && a.dbg != nullptr) {
      unsigned long long* d = a.dbg + ((size_t)blockIdx.x * (BLOCK / 64) + wave) * 4;
      d[0] = t_stage; d[1] = t_gather; d[2] = t_wait; d[3] = __builtin_amdgcn_s_memtime() - t_all0;
    }
  }
  publish_flags(f, a.flags);
#undef PLAIDHIP_PREFETCH
#undef PLAIDHIP_PF_ONE
#undef PLAIDHIP_ST_ONE
#undef PLAIDHIP_ITEMS10
}

static int launch_colpair_mixed(plaidhip_ctx* ctx, const plaidhip_geneset* gs, SpmmArgs a) {
  const plaidhip_slice& sl = gs->slices[0];
  a.g = sl.gs;
  a.g0 = 0;
  a.acc_mode = 0;
  a.tile_idx = reinterpret_cast<const uint4*>(sl.d_tile_idx);
  a.wave_chunk_off = sl.d_wave_chunk_off;
  a.wave_tile_off = sl.d_wave_tile_off;
  a.wtile_end = sl.d_wtile_end;
  a.meta_j = sl.d_meta_j;
  a.meta_w = sl.d_meta_w;
  a.meta_k = sl.d_meta_k;
  const size_t smem = (size_t)(sl.gs + kPadSlots) * sizeof(double);
  PH_FULL_LDS(ctx, (&spmm_colpair_mixed<false>));
  int grid = ctx->num_cu;
  const int npairs = (a.n + 1) / 2;
  if (grid > npairs) grid = npairs;
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 10
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
  {
    hipLaunchKernelGGL(spmm_colpair_mixed<false>, dim3(grid), dim3(1024), smem, ctx->stream, a);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}



// ---------------------------------------------------------------------------------------------
// Quad kernel for RANK-valued X (replaid.sing R/plaid.R:215-217, replaid.ssgsea(alpha = 0) :245-253, replaid.ucell
// :277-279, replaid.aucell, replaid.gsva(tau = 0) on unsigned ranks): what `colranks` returns are half-integers in
// [0.5, nrow(X)], so 2 * rank is an integer <= 40,896 for every column the LDS-resident kernels take -- a u16.
// FOUR sample columns share one 8-byte LDS entry {2rA, 2rB, 2rC, 2rD}; one address op + one ds_read_b64 serve four
// scores (2 bytes of LDS per score against 8 in the fp64 kernels and 4 with fp32 staging), the whole 20k-gene column
// quad fits the LDS (no gene slices, no partial sums) and the sums are 32-bit integers: EXACT, independent of the
// summation order, and -- after the one conversion sum / 2 in the epilogue -- bit-identical to what the fp64 kernels
// produce (their sums of half-integers are exact too).  Same host schedule as the one-column kernel (32-lane halves
// of ds_read_b64 against 32 bank pairs), same per-wave tile streams, same epilogue.
// X arrives as the doubles `colranks` wrote: four column streams, converted while staging with one fp64 add per value (the
// low word of x + 2^51 is 2x).  (A rank kernel that writes the u16 matrix itself would quarter those bytes and let the whole
// next quad be prefetched in registers -- staging is 20 % of the kernel at 5,000 sets, 3 % at 50,000; not built.)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 lds_cu32x2;
__device__ __forceinline__ u32x2 lds_u32x2_at(uint32_t byte_off) {
  return *reinterpret_cast<lds_cu32x2*>(static_cast<uintptr_t>(byte_off));
}
// 2x of a half-integer 0 <= x < 2^31 as an integer: the ulp of x + 2^51 is 1/2, so its low mantissa word counts halves
__device__ __forceinline__ uint32_t add_hi16(uint32_t acc, uint32_t v) {   // acc + (v >> 16) in one instruction
  uint32_t r;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(acc), "v"(v));
  return r;
}
// Two tests say whether x was a rank: the high word of x + 2^51 is 0x43200000 exactly when 0 <= x < 2^31 (NaN, +-Inf,
// negatives and huge values differ), and (x + 2^51) - 2^51 == x exactly when x is a multiple of 1/2 (the sum rounds to the
// grid of halves; the difference is exact).  (The wave's sticky IEEE "inexact" status bit would give the second test for
// free, but gfx950 does not record it with the exception masked: tools/ubench/inexact_flag.hip.)
__device__ __forceinline__ uint32_t twice_as_u32(double x, uint32_t& not_a_rank) {
  const double y = x + 0x1p51;
  not_a_rank |= ((uint32_t)__double2hiint(y) ^ 0x43200000u) | ((y - 0x1p51 != x) ? 1u : 0u);
  return (uint32_t)__double2loint(y);
}

struct SpmmQuadArgs {
  SpmmArgs s;
};

template <bool STAMP>
__global__ void __launch_bounds__(1024)
spmm_colquad_u16(SpmmQuadArgs qa_) {
  const SpmmArgs& a = qa_.s;
  constexpr int BLOCK = 1024;
  unsigned long long t_stage = 0, t_gather = 0, t_wait = 0, t_all0 = 0;
  if constexpr (STAMP) t_all0 = __builtin_amdgcn_s_memtime();
  extern __shared__ __align__(16) unsigned char smem_raw[];
  u32x2* ent = reinterpret_cast<u32x2*>(smem_raw);
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    if ((uint32_t)(uintptr_t)((lds_u8*)smem_raw) != 0u) __builtin_trap();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t f = 0, chk = 0, chkh = 0;
  const double alpha = (a.alpha_div != nullptr) ? a.alpha / *a.alpha_div : a.alpha;
  const bool is_mean = a.stat == PLAIDHIP_STAT_MEAN;
  const int ch_begin = ((cptr_i32)a.wave_chunk_off)[wave];
  const int ch_end = ((cptr_i32)a.wave_chunk_off)[wave + 1];
  const int tk_begin = ((cptr_i32)a.wave_tile_off)[wave];
  const int nquads = (a.n + 3) >> 2;

  // ---- staging ------------------------------------------------------------------------------------------------
  // fp64 input: an "item" is 1,024 gene pairs: per thread one 16-byte load {x[2i], x[2i+1]} from each of the four
  // columns -> two LDS entries, one ds_write_b128.  Items 0..5 of the NEXT quad are requested when the wavefront has
  // finished its stream (their registers are free then); items 6..9 follow behind the barrier.
  const int g2 = a.g >> 1;   // gene pairs
  f64x2 pa0, pb0, pc0, pd0, pa1, pb1, pc1, pd1, pa2, pb2, pc2, pd2, pa3, pb3, pc3, pd3, pa4, pb4, pc4, pd4, pa5, pb5, pc5, pd5;
  pa0 = pb0 = pc0 = pd0 = pa1 = pb1 = pc1 = pd1 = pa2 = pb2 = pc2 = pd2 = f64x2{0.0, 0.0};
  pa3 = pb3 = pc3 = pd3 = pa4 = pb4 = pc4 = pd4 = pa5 = pb5 = pc5 = pd5 = f64x2{0.0, 0.0};

#define PLAIDHIP_QCOLS(qq_)                                                     \
  const int c0_ = 4 * (qq_);                                                     \
  const int c1_ = (c0_ + 1 < a.n) ? c0_ + 1 : c0_;                               \
  const int c2_ = (c0_ + 2 < a.n) ? c0_ + 2 : c0_;                               \
  const int c3_ = (c0_ + 3 < a.n) ? c0_ + 3 : c0_;
#define PLAIDHIP_LD_F64(k, ra, rb, rc, rd)                                                                          \
  {                                                                                                                  \
    const bool in_ = (k + 1) * BLOCK <= g2 || (k * BLOCK < g2 && tid_o + k * BLOCK < g2);                            \
    ra = rb = rc = rd = f64x2{0.0, 0.0};                                                                             \
    if (in_) {                                                                                                       \
      ra = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xa_ + (size_t)k * BLOCK * 16 + loff16));       \
      rb = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xb_ + (size_t)k * BLOCK * 16 + loff16));       \
      rc = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xc_ + (size_t)k * BLOCK * 16 + loff16));       \
      rd = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(xd_ + (size_t)k * BLOCK * 16 + loff16));       \
    }                                                                                                                \
  }
#define PLAIDHIP_ST_F64(k, ra, rb, rc, rd)                                                                           \
  if (k * BLOCK < g2) {                                                                                              \
    const int i_ = tid_o + k * BLOCK;                                                                                \
    if (i_ < g2) {                                                                                                   \
      const uint32_t a0_ = twice_as_u32(ra.x, chkh), b0_ = twice_as_u32(rb.x, chkh), c0v_ = twice_as_u32(rc.x, chkh), d0_ = twice_as_u32(rd.x, chkh); \
      const uint32_t a1_ = twice_as_u32(ra.y, chkh), b1_ = twice_as_u32(rb.y, chkh), c1v_ = twice_as_u32(rc.y, chkh), d1_ = twice_as_u32(rd.y, chkh); \
      chk |= (a0_ | b0_) | (c0v_ | d0_) | (a1_ | b1_) | (c1v_ | d1_);                                                \
      ent4[i_] = u32x4{a0_ | (b0_ << 16), c0v_ | (d0_ << 16), a1_ | (b1_ << 16), c1v_ | (d1_ << 16)};                \
    }                                                                                                                \
  }
#define PLAIDHIP_XPTRS_F64(qq_)                                                                      \
  PLAIDHIP_QCOLS(qq_)                                                                                 \
  const char* xa_ = reinterpret_cast<const char*>(a.X + (int64_t)c0_ * a.ldx);                        \
  const char* xb_ = reinterpret_cast<const char*>(a.X + (int64_t)c1_ * a.ldx);                        \
  const char* xc_ = reinterpret_cast<const char*>(a.X + (int64_t)c2_ * a.ldx);                        \
  const char* xd_ = reinterpret_cast<const char*>(a.X + (int64_t)c3_ * a.ldx);
  // what a wavefront requests of the next quad when its stream is done
#define PLAIDHIP_PREFETCH(qq_)                                                                        \
  do {                                                                                                \
    int tid_o = tid;                                                                                  \
    asm volatile("" : "+v"(tid_o));                                                                   \
    const uint32_t loff16 = (uint32_t)tid_o * 16u;                                                    \
    PLAIDHIP_XPTRS_F64(qq_)                                                                           \
    PLAIDHIP_LD_F64(0, pa0, pb0, pc0, pd0) PLAIDHIP_LD_F64(1, pa1, pb1, pc1, pd1) PLAIDHIP_LD_F64(2, pa2, pb2, pc2, pd2) \
    PLAIDHIP_LD_F64(3, pa3, pb3, pc3, pd3) PLAIDHIP_LD_F64(4, pa4, pb4, pc4, pd4) PLAIDHIP_LD_F64(5, pa5, pb5, pc5, pd5) \
  } while (0)

  int q = blockIdx.x;
  if (q < nquads) PLAIDHIP_PREFETCH(q);
  for (; q < nquads; q += gridDim.x) {
    const int cA = 4 * q;
    const int ncol = (a.n - cA) < 4 ? (a.n - cA) : 4;   // columns of this quad that exist
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if constexpr (STAMP) ts0 = __builtin_amdgcn_s_memtime();
    {
      int tid_o = tid;
      asm volatile("" : "+v"(tid_o));
      const uint32_t loff16 = (uint32_t)tid_o * 16u;
      u32x4* ent4 = reinterpret_cast<u32x4*>(smem_raw);
      {
        PLAIDHIP_XPTRS_F64(q)
        PLAIDHIP_ST_F64(0, pa0, pb0, pc0, pd0) PLAIDHIP_ST_F64(1, pa1, pb1, pc1, pd1) PLAIDHIP_ST_F64(2, pa2, pb2, pc2, pd2)
        // items 6..8 take the registers of 0..2 while 3..5 are converted; item 9 follows
        PLAIDHIP_LD_F64(6, pa0, pb0, pc0, pd0) PLAIDHIP_LD_F64(7, pa1, pb1, pc1, pd1) PLAIDHIP_LD_F64(8, pa2, pb2, pc2, pd2)
        PLAIDHIP_ST_F64(3, pa3, pb3, pc3, pd3) PLAIDHIP_ST_F64(4, pa4, pb4, pc4, pd4) PLAIDHIP_ST_F64(5, pa5, pb5, pc5, pd5)
        PLAIDHIP_LD_F64(9, pa3, pb3, pc3, pd3)
        PLAIDHIP_ST_F64(6, pa0, pb0, pc0, pd0) PLAIDHIP_ST_F64(7, pa1, pb1, pc1, pd1) PLAIDHIP_ST_F64(8, pa2, pb2, pc2, pd2)
        PLAIDHIP_ST_F64(9, pa3, pb3, pc3, pd3)
        if ((a.g & 1) && tid_o == 0) {
          const int64_t gl = a.g - 1;
          const uint32_t va = twice_as_u32(reinterpret_cast<const double*>(xa_)[gl], chkh), vb = twice_as_u32(reinterpret_cast<const double*>(xb_)[gl], chkh);
          const uint32_t vc = twice_as_u32(reinterpret_cast<const double*>(xc_)[gl], chkh), vd = twice_as_u32(reinterpret_cast<const double*>(xd_)[gl], chkh);
          chk |= (va | vb) | (vc | vd);
          ent[gl] = u32x2{va | (vb << 16), vc | (vd << 16)};
        }
      }
      if (tid_o < kPadSlots) ent[a.g + tid_o] = u32x2{0u, 0u};
    }
    __syncthreads();
    if constexpr (STAMP) ts1 = __builtin_amdgcn_s_memtime();
    const int nq = q + gridDim.x;
    const bool want_pf = nq < nquads;

    if (ch_begin < ch_end) {
      gptr_u8 ibase = (gptr_u8)a.tile_idx + (int64_t)ch_begin * 1024;  // uniform
      uint32_t lane_o = (uint32_t)lane;
      asm volatile("" : "+v"(lane_o));
      const uint32_t ioff = lane_o * 16u;
      const uint32_t moff4 = lane_o * 4u, moff8 = lane_o * 8u;
      const cptr_i32 wtile_end = (cptr_i32)a.wtile_end;
#define PLAIDHIP_LOADQ(rel) (*(gptr_u32x4)(ibase + (int64_t)(rel) * 1024 + ioff))
#define PLAIDHIP_GATHER4A(qv)                                                \
  va0 = lds_u32x2_at(off_lo((qv).x)); va1 = lds_u32x2_at(off_hi((qv).x));     \
  va2 = lds_u32x2_at(off_lo((qv).y)); va3 = lds_u32x2_at(off_hi((qv).y));
#define PLAIDHIP_GATHER4B(qv)                                                \
  vb0 = lds_u32x2_at(off_lo((qv).z)); vb1 = lds_u32x2_at(off_hi((qv).z));     \
  vb2 = lds_u32x2_at(off_lo((qv).w)); vb3 = lds_u32x2_at(off_hi((qv).w));
      // Four u16 fields per gathered entry {lo | hi << 16, lo' | hi' << 16}.  Per dword two running sums: H = sum of the
      // high fields (v_add_u32_sdwa: the field select is free) and T = sum of the WHOLE dwords modulo 2^32 (v_add3_u32:
      // one instruction per two gathers); the low fields' sum is T - (H << 16) modulo 2^32, exact because it is < 2^32
      // (<= 20,448 ranks of <= 40,896).  3 vector instructions per gather instead of 4 adds + 4 field extractions.
#define PLAIDHIP_ACC2(v, w)                                                  \
  tx = tx + (v).x + (w).x; ty = ty + (v).y + (w).y;                           \
  hx = add_hi16(hx, (v).x); hx = add_hi16(hx, (w).x);                         \
  hy = add_hi16(hy, (v).y); hy = add_hi16(hy, (w).y);
#define PLAIDHIP_ADD4(V) PLAIDHIP_ACC2(V##0, V##1) PLAIDHIP_ACC2(V##2, V##3)
#define PLAIDHIP_EPI(isum, cc)                                                 \
  {                                                                            \
    const double sum_ = 0.5 * (double)(isum);   /* exact */                    \
    const double w_ = is_mean ? mw : 1.0;                                      \
    const double v_ = alpha * (sum_ * w_) + a.beta * (mk * w_);                \
    if (a.nt_store) __builtin_nontemporal_store(v_, &a.S[(int64_t)(cc) * a.lds + mj]);  \
    else a.S[(int64_t)(cc) * a.lds + mj] = v_;                                 \
    f |= (v_ < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;                              \
    f |= (v_ == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;                            \
    f |= (v_ != v_) ? PLAIDHIP_FLAG_HAS_NAN : 0u;                              \
  }
#define PLAIDHIP_TILE_END(chv)                                                                 \
  if ((chv) + 1 == next_end) { /* wave-uniform: tile finished -> epilogue */                   \
    if (mj >= 0) {                                                                             \
      PLAIDHIP_EPI(tx - (hx << 16), cA)                                                        \
      if (ncol > 1) PLAIDHIP_EPI(hx, cA + 1)                                                   \
      if (ncol > 2) PLAIDHIP_EPI(ty - (hy << 16), cA + 2)                                      \
      if (ncol > 3) PLAIDHIP_EPI(hy, cA + 3)                                                   \
    }                                                                                          \
    ++k;                                                                                       \
    next_end = wtile_end[k];                                                                   \
    mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);  \
    mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);   \
    mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);   \
    tx = ty = hx = hy = 0u;                                                                    \
  }
      // 8 index chunks (8 KiB per wave) in flight: a chunk of this kernel is worth few cycles (3 vector instructions and
      // one 8-byte LDS read per FOUR scores), so the index stream from L2 needs the depth to stay ahead
      u32x4 qa = PLAIDHIP_LOADQ(0);
      u32x4 qb = PLAIDHIP_LOADQ(1);
      u32x4 qc = PLAIDHIP_LOADQ(2);
      u32x4 qd = PLAIDHIP_LOADQ(3);
      u32x4 qe = PLAIDHIP_LOADQ(4);
      u32x4 qf = PLAIDHIP_LOADQ(5);
      u32x4 qg = PLAIDHIP_LOADQ(6);
      u32x4 qh = PLAIDHIP_LOADQ(7);
      int k = tk_begin;
      int next_end = wtile_end[k];
      uint32_t tx = 0u, ty = 0u, hx = 0u, hy = 0u;
      int mj = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.meta_j) + (int64_t)k * 256 + moff4);
      double mw = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_w) + (int64_t)k * 512 + moff8);
      double mk = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.meta_k) + (int64_t)k * 512 + moff8);
      u32x2 va0, va1, va2, va3, vb0, vb1, vb2, vb3;
      // half-chunk software pipeline: the four gathers of the next half are in the LDS queue while the adds of the
      // current half issue (spare chunks exist behind the stream: over-read ids are gathered, never added)
      int ch = ch_begin;
      ibase += 8 * 1024;
#define PLAIDHIP_STEP(qcur, qnext, rel, chv)                                             \
  PLAIDHIP_GATHER4B(qcur)                                                                \
  qcur = PLAIDHIP_LOADQ(rel);                                                            \
  PLAIDHIP_ADD4(va)                                                                      \
  PLAIDHIP_GATHER4A(qnext)                                                               \
  PLAIDHIP_ADD4(vb)                                                                      \
  PLAIDHIP_TILE_END(chv)
#define PLAIDHIP_STEP_TAIL(qcur, qnext)                                                  \
  if (ch < ch_end) {                                                                     \
    PLAIDHIP_GATHER4B(qcur) PLAIDHIP_ADD4(va) PLAIDHIP_GATHER4A(qnext) PLAIDHIP_ADD4(vb) \
    PLAIDHIP_TILE_END(ch)                                                                \
    ++ch;                                                                                \
  }
      PLAIDHIP_GATHER4A(qa)
      for (; ch + 7 < ch_end; ch += 8, ibase += 8 * 1024) {
        PLAIDHIP_STEP(qa, qb, 0, ch)
        PLAIDHIP_STEP(qb, qc, 1, ch + 1)
        PLAIDHIP_STEP(qc, qd, 2, ch + 2)
        PLAIDHIP_STEP(qd, qe, 3, ch + 3)
        PLAIDHIP_STEP(qe, qf, 4, ch + 4)
        PLAIDHIP_STEP(qf, qg, 5, ch + 5)
        PLAIDHIP_STEP(qg, qh, 6, ch + 6)
        PLAIDHIP_STEP(qh, qa, 7, ch + 7)
      }
      PLAIDHIP_STEP_TAIL(qa, qb)
      PLAIDHIP_STEP_TAIL(qb, qc)
      PLAIDHIP_STEP_TAIL(qc, qd)
      PLAIDHIP_STEP_TAIL(qd, qe)
      PLAIDHIP_STEP_TAIL(qe, qf)
      PLAIDHIP_STEP_TAIL(qf, qg)
      PLAIDHIP_STEP_TAIL(qg, qh)
#undef PLAIDHIP_STEP
#undef PLAIDHIP_STEP_TAIL
#undef PLAIDHIP_GATHER4A
#undef PLAIDHIP_GATHER4B
#undef PLAIDHIP_ACC2
#undef PLAIDHIP_ADD4
#undef PLAIDHIP_TILE_END
#undef PLAIDHIP_EPI
#undef PLAIDHIP_LOADQ
    }
    if (want_pf) {
      PLAIDHIP_PREFETCH(nq);
    } else {
      pa0 = pb0 = pc0 = pd0 = pa1 = pb1 = pc1 = pd1 = pa2 = pb2 = pc2 = pd2 = f64x2{0.0, 0.0};
      pa3 = pb3 = pc3 = pd3 = pa4 = pb4 = pc4 = pd4 = pa5 = pb5 = pc5 = pd5 = f64x2{0.0, 0.0};
        }
    if constexpr (STAMP) ts2 = __builtin_amdgcn_s_memtime();
    __syncthreads();  // the quad is overwritten by the next iteration
    if constexpr (STAMP) {
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      t_stage += ts1 - ts0;
      t_gather += ts2 - ts1;
      t_wait += ts3 - ts2;
    }
  }
  if constexpr (STAMP) {
    if (lane == 0 && a.dbg != nullptr) {
      unsigned long long* d = a.dbg + ((size_t)blockIdx.x * (BLOCK / 64) + wave) * 4;
      d[0] = t_stage; d[1] = t_gather; d[2] = t_wait; d[3] = __builtin_amdgcn_s_memtime() - t_all0;
    }
  }
  // a value that is not a rank was staged (2x does not fit 16 bits, NaN, +-Inf, negative): these scores are wrong, and the
  // fp64 launch enqueued behind this one replaces them (spec_guard); the flag words of this launch go to spec[1..3]
  chk = (chk >> 16) | chkh;
  for (int off = 32; off >= 1; off >>= 1) chk |= __shfl_xor(chk, off, 64);
  if (lane == 0 && chk != 0u) __hip_atomic_store(&a.spec[0], a.spec_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  publish_flags(f, a.spec + 1);
#undef PLAIDHIP_PREFETCH
#undef PLAIDHIP_QCOLS
#undef PLAIDHIP_LD_F64
#undef PLAIDHIP_ST_F64
#undef PLAIDHIP_XPTRS_F64
}

// g <= kMaxLdsGenes and 2 g < 65,536 always hold for the one-slice plan (20,448 genes)
static int launch_colquad(plaidhip_ctx* ctx, const plaidhip_geneset* gs, SpmmArgs a) {
  const plaidhip_slice& sl = gs->slices[0];
  a.g = sl.gs;
  a.g0 = 0;
  a.acc_mode = 0;
  a.tile_idx = reinterpret_cast<const uint4*>(sl.d_tile_idx);
  a.wave_chunk_off = sl.d_wave_chunk_off;
  a.wave_tile_off = sl.d_wave_tile_off;
  a.wtile_end = sl.d_wtile_end;
  a.meta_j = sl.d_meta_j;
  a.meta_w = sl.d_meta_w;
  a.meta_k = sl.d_meta_k;
  SpmmQuadArgs qa{a};
  const size_t smem = (size_t)(sl.gs + kPadSlots) * sizeof(double);
  int grid = ctx->num_cu;
  const int nquads = (a.n + 3) / 4;
  if (grid > nquads) grid = nquads;
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 11
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
  {
    PH_FULL_LDS(ctx, (&spmm_colquad_u16<false>));
    hipLaunchKernelGGL((spmm_colquad_u16<false>), dim3(grid), dim3(1024), smem, ctx->stream, qa);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

static int nt_store_mode(const plaidhip_ctx* ctx, const plaidhip_geneset* gs) {
  if (ctx->opt_nt_store >= 0) return ctx->opt_nt_store;
  return gs->rows_in_order ? 1 : 0;
}

// which dense-X kernel (plaidhip_set_option(PLAIDHIP_OPT_SPMM_DENSE_KERNEL)): 0 one-column, 1 default (pair where
// it applies), 2 pair wherever possible
static int pair_kernel_mode(const plaidhip_ctx* ctx) {
  return ctx->opt_dense_kernel == 1 ? 0 : (ctx->opt_dense_kernel == 2 ? 2 : 1);
}

// what the MED form of the pair kernel needs (dense X; launch_spmm_dense_fused_f64)
struct plaidhip_pair_med {
  const double* u;
  double beta_kappa;
  const double* cal;
  double* pred;
  unsigned long long* cand;
  uint32_t* cnt;
  int32_t capc;
};

static int launch_colpair(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const double* X, int64_t ldx,
                          const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t n,
                          int stat, double alpha, const double* alpha_div, double beta, double* S, int64_t lds,
                          uint32_t* flags, bool auto_select = false, uint32_t* spec = nullptr, uint32_t spec_gen = 0,
                          const plaidhip_pair_med* med = nullptr) {
  const plaidhip_pair_plan& pl = gs->pair;
  int32_t gmax = 0;
  for (const plaidhip_pair_slice& sl : pl.slices) gmax = sl.gs > gmax ? sl.gs : gmax;
  const size_t smem = (size_t)(gmax + kPadSlotsPair) * 16;
  PH_FULL_LDS(ctx, (&spmm_colpair_f64<false>));
  SpmmPairArgs a{};
  a.X = X;
  a.ldx = ldx;
  a.Xp = Xp;
  a.Xi = Xi;
  a.Xx = Xx;
  a.sparse_cells = auto_select ? (int64_t)gs->g * n : 0;
  a.n = n;
  a.npairs = (n + 1) / 2;
  a.nslices = (int32_t)pl.slices.size();
  a.ktiles = pl.ktiles;
  a.nt_store = nt_store_mode(ctx, gs);
  // the partial sums of the slice before are read with non-temporal loads when the scratches of an XCD's 32 workgroups
  // (1 KiB per tile each) come to more than 8 MiB -- twice the XCD's 4 MiB L2, where they are read once and dead and would
  // only displace the index lists (50,000 sets, 12.5 MiB: -4 % kernel time; 5,000 sets, L2-resident scratch: +3 %, so not
  // there; the threshold sits between the two measured shapes)
  const bool part_nt = a.nslices > 1 && (int64_t)(pl.ktiles + 1) * 1024 * 32 > (8ll << 20);
  a.slices = pl.d_slices;
  a.wave_tile_off = pl.d_wave_tile_off;
  a.meta_j = pl.d_meta_j;
  a.meta_w = pl.d_meta_w;
  a.meta_k = pl.d_meta_k;
  a.partial = reinterpret_cast<f64x2*>(pl.d_partial);
  a.stat = stat;
  a.alpha = alpha;
  a.beta = beta;
  a.alpha_div = alpha_div;
  a.S = S;
  a.lds = lds;
  a.flags = flags;
  a.spec = spec;
  a.spec_gen = spec_gen;
  int per_cu = (int)(kLdsBytes / smem);
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  int grid = ctx->num_cu * per_cu;
  if (a.nslices > 1 && grid > pl.partial_wgs) grid = pl.partial_wgs;
  if (grid > a.npairs) grid = a.npairs;
  if (med != nullptr) {   // dense X, medians selected on the fly
    a.med_u = med->u;
    a.med_beta_kappa = med->beta_kappa;
    a.med_cal = med->cal;
    a.med_pred = med->pred;
    a.med_cand = med->cand;
    a.med_cnt = med->cnt;
    a.med_capc = med->capc;
    if (part_nt) {
      PH_FULL_LDS(ctx, (&spmm_colpair_f64<false, 0, false, true, true>));
      hipLaunchKernelGGL((spmm_colpair_f64<false, 0, false, true, true>), dim3(grid), dim3(1024), smem, ctx->stream, a);
    } else {
      PH_FULL_LDS(ctx, (&spmm_colpair_f64<false, 0, false, true>));
      hipLaunchKernelGGL((spmm_colpair_f64<false, 0, false, true>), dim3(grid), dim3(1024), smem, ctx->stream, a);
    }
  } else if (Xp != nullptr) {   // sparse X
    PH_FULL_LDS(ctx, (&spmm_colpair_f64<false, 0, true>));
    hipLaunchKernelGGL((spmm_colpair_f64<false, 0, true>), dim3(grid), dim3(1024), smem, ctx->stream, a);
  }
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 12
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
  else if (part_nt) {
    PH_FULL_LDS(ctx, (&spmm_colpair_f64<false, 0, false, false, true>));
    hipLaunchKernelGGL((spmm_colpair_f64<false, 0, false, false, true>), dim3(grid), dim3(1024), smem, ctx->stream, a);
  } else {
    hipLaunchKernelGGL(spmm_colpair_f64<false>, dim3(grid), dim3(1024), smem, ctx->stream, a);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 13
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif

template <bool CSC_X, int BLOCK>
static int launch_one(plaidhip_ctx* ctx, const plaidhip_slice& sl, SpmmArgs& a) {
  const size_t smem = (size_t)(sl.gs + kPadSlots) * sizeof(double);
  PH_FULL_LDS(ctx, (&spmm_colgather_f64<CSC_X, BLOCK>));
  // persistent: as many workgroups as fit on the chip at once (LDS- or wave-limited)
  int per_cu = (int)(kLdsBytes / smem);
  const int wave_cap = 2048 / BLOCK;
  if (per_cu > wave_cap) per_cu = wave_cap;
  if (per_cu < 1) per_cu = 1;
  int grid = ctx->num_cu * per_cu;
  if (grid > a.n) grid = a.n;
#ifdef PLAIDHIP_DIAG
#define PH_DIAG_SECTION 14
#include "kernels_spmm_diag.inc"
#undef PH_DIAG_SECTION
#endif
  hipLaunchKernelGGL((spmm_colgather_f64<CSC_X, BLOCK>), dim3(grid), dim3(BLOCK), smem, ctx->stream, a);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

// One launch per gene slice; with several slices the partial sums live in S between launches.
template <bool CSC_X>
static int launch_colgather(plaidhip_ctx* ctx, const plaidhip_geneset* gs, SpmmArgs a) {
  const int ns = (int)gs->slices.size();
  const double* Xbase = a.X;
  for (int si = 0; si < ns; ++si) {
    const plaidhip_slice& sl = gs->slices[si];
    a.g = sl.gs;
    a.g0 = sl.g0;
    a.acc_mode = ns == 1 ? 0 : (si == 0 ? 1 : (si == ns - 1 ? 3 : 2));
    if (!CSC_X) a.X = Xbase + sl.g0;
    a.tile_idx = reinterpret_cast<const uint4*>(sl.d_tile_idx);
    a.wave_chunk_off = sl.d_wave_chunk_off;
    a.wave_tile_off = sl.d_wave_tile_off;
    a.wtile_end = sl.d_wtile_end;
    a.meta_j = sl.d_meta_j;
    a.meta_w = sl.d_meta_w;
    a.meta_k = sl.d_meta_k;
    const int block = sl.waves * 64;
    int rc;
    if (block == 1024) rc = launch_one<CSC_X, 1024>(ctx, sl, a);
    else if (block == 512) rc = launch_one<CSC_X, 512>(ctx, sl, a);
    else rc = launch_one<CSC_X, 256>(ctx, sl, a);
    if (rc != PLAIDHIP_OK) return rc;
  }
  return PLAIDHIP_OK;
}

static void fill_args(const plaidhip_ctx* ctx, SpmmArgs& a, const plaidhip_geneset* gs, int32_t n, int stat, double alpha,
                      const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags) {
  a.nt_store = nt_store_mode(ctx, gs);
  a.alpha_div = alpha_div;
  a.n = n;
  a.m = gs->m;
  a.stat = stat;
  a.alpha = alpha;
  a.beta = beta;
  a.S = S;
  a.lds = lds;
  a.flags = flags;
}

int launch_spmm_dense_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const double* X,
                          int64_t ldx, int32_t n, int stat, double alpha, const double* alpha_div,
                          double beta, double* S, int64_t lds, uint32_t* flags, int x_kind) {
  ctx->fmed.valid = false;   // (any other crossprod on this context: what a fused launch left behind no longer describes S)
  if (n == 0 || gs->m == 0) return PLAIDHIP_OK;
  if (ctx->opt_dense_kernel == 3)   // opt-in: the dense contraction on the matrix cores (kernels_mfma.hip)
    return launch_spmm_mfma_f64(ctx, const_cast<plaidhip_geneset*>(gs), X, ldx, n, stat, alpha, alpha_div, beta, S, lds, flags);
  // Exact compact staging for rank inputs (PLAIDHIP_OPT_RANKS_F32: 0 keeps them on the fp64 kernels -- tests compare):
  //   u16: X holds what colranks returns (half-integers <= nrow): 2x as u16, four columns per LDS entry, integer sums;
  //   fp32: (half-)integers <= 20,448 of any sign are exact in fp32, and so are the four-term fp32 partial sums of the
  //   kernel (< 2^17 with one fractional bit).  The opt-in mixed precision takes the fp32 kernel for any X.
  // (the 16-byte loads {x[2i], x[2i+1]} of these kernels need no 16-byte alignment: global memory takes dword-aligned
  //  dwordx4 accesses, so an odd leading dimension -- every other column 8 bytes off -- runs the same kernels)
  const bool one_slice_16 = gs->slices.size() == 1 && gs->slices[0].waves == 16 && (g_ablate == 0 || g_ablate == 4);
  uint32_t* spec = nullptr;
  uint32_t spec_gen = 0;
  if (x_kind == PLAIDHIP_X_RANKS && ctx->opt_ranks_f32 >= 2 && one_slice_16 && ctx->d_spec != nullptr) {
    // speculative: exact and bit-identical to the fp64 kernels IF every value is a rank.  A value that is not (NaN ranks
    // of NaN inputs, matrixStats keeps NA; anything a device-API caller passes) is seen while staging, and the fp64
    // kernel enqueued right behind then recomputes the scores -- it returns at once otherwise (spec_guard).
    SpmmArgs a{};
    a.X = X;
    a.ldx = ldx;
    fill_args(ctx, a, gs, n, stat, alpha, alpha_div, beta, S, lds, flags);
    if (++ctx->spec_gen == 0u) ctx->spec_gen = 1u;
    a.spec = spec = ctx->d_spec;
    a.spec_gen = spec_gen = ctx->spec_gen;
    const int rc = launch_colquad(ctx, gs, a);
    if (rc != PLAIDHIP_OK) return rc;
    x_kind = PLAIDHIP_X_ANY;   // (the fallback is the plain fp64 route)
  }
  const bool x_exact_in_f32 = x_kind != PLAIDHIP_X_ANY && ctx->opt_ranks_f32 >= 1;
  if (spec == nullptr && (ctx->precision == PLAIDHIP_PRECISION_MIXED || x_exact_in_f32) && one_slice_16) {
    // fp32 operand staging; one gene slice and the 1024-thread schedule only
    SpmmArgs a{};
    a.X = X;
    a.ldx = ldx;
    fill_args(ctx, a, gs, n, stat, alpha, alpha_div, beta, S, lds, flags);
    return launch_colpair_mixed(ctx, gs, a);
  }
  {
    // two columns per pass
    const int mode = pair_kernel_mode(ctx);
    const bool diag = g_ablate == 0 || g_ablate == 2 || (g_ablate >= 4 && g_ablate <= 16);
    if (diag && mode != 0 && !gs->pair.slices.empty())
      return launch_colpair(ctx, gs, X, ldx, nullptr, nullptr, nullptr, n, stat, alpha, alpha_div, beta, S, lds, flags, false,
                            spec, spec_gen);
  }
  SpmmArgs a{};
  a.X = X;
  a.ldx = ldx;
  fill_args(ctx, a, gs, n, stat, alpha, alpha_div, beta, S, lds, flags);
  a.spec = spec;
  a.spec_gen = spec_gen;
  return launch_colgather<false>(ctx, gs, a);
}

int launch_spmm_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp,
                        const int32_t* Xi, const double* Xx, int32_t n, int64_t nnz, int stat, double alpha,
                        const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags,
                        bool bounded, const double* xmax_dev, double xmax_host) {
  ctx->fmed.valid = false;
  if (n == 0 || gs->m == 0) return PLAIDHIP_OK;
  if ((g_ablate == 0 || g_ablate >= 100) && pair_kernel_mode(ctx) != 0 && !gs->pair.slices.empty()) {
    // sparse-aware scatter or dense-work gather.  With nnz(X) from the caller the choice is made here (one
    // launch); without it (nnz < 0: only the device knows) both are enqueued and the one that does not apply
    // returns at once.
    int sm = sparse_mode(ctx);
    if (sm == 0 && nnz >= 0) sm = (nnz * 8 < (int64_t)gs->g * n) ? 1 : 2;
    if (sm != 2) {
      const int rc = launch_spmm_scatter_csc_f64(ctx, gs, Xp, Xi, Xx, n, stat, alpha, alpha_div, beta, S, lds, flags, sm == 0,
                                                 bounded, xmax_dev, xmax_host, nnz);
      if (rc != PLAIDHIP_OK || sm == 1) return rc;
    }
    return launch_colpair(ctx, gs, nullptr, 0, Xp, Xi, Xx, n, stat, alpha, alpha_div, beta, S, lds, flags, sm == 0);
  }
  SpmmArgs a{};
  a.Xp = Xp;
  a.Xi = Xi;
  a.Xx = Xx;
  fill_args(ctx, a, gs, n, stat, alpha, alpha_div, beta, S, lds, flags);
  return launch_colgather<true>(ctx, gs, a);
}

// Candidate slots per column of the fused-medians scratch.  The bracket catches 1-5 % of a column's m scores (half width 2.3 x
// the 90th percentile of the calibration columns' deviations) and median_select_kernel takes at most 4,096 of them, so the
// slices of a column hold 16 % of m between them, 1,024 at least and 8,192 at most (round 5 reserved 8,192 whatever m: 64 KB
// per column -- more than the column's own scores below 8,192 sets, 8 GB for a 125,000-cell shard).  A slice that
// overflows sends its column to the standalone kernel (the counts say so), so the size is a speed matter only.  The scratch
// is 8 bytes per slot + 16 bytes per (column, slice): <= 0.2 x the bytes of S.
static int32_t fused_cand_per_column(int32_t m) {
  const int64_t want = ((int64_t)m * 16 + 99) / 100;
  return (int32_t)std::min<int64_t>(8192, std::max<int64_t>(1024, want));
}

// ---- the sparse crossprod that also selects the column medians of its result (normalize_medians, R/plaid.R:561-572) ----
// Applies when the scatter kernel takes the input and the result has more sets per column than the register-resident median
// kernel takes (m > 6,144: there the standalone median kernel is a second pass over the whole score matrix -- 40 GB at
// config 3); everything else runs the plain crossprod and launch_col_medians_resume the standalone kernels.
//   1. calibration: crossprod + standalone medians of the first K columns (their scores are written again below);
//   2. the predicted mean of every column (one pass over the stored values of X);
//   3. {offset, half width, ignore-zero rule} of the bracket from 1. and 2.;
//   4. the crossprod of ALL columns with the classifying epilogue.
// Nothing is read back: the decisions are taken in steps 3 / 5 on the device.  (5 = launch_col_medians_resume, after the
// caller has all-reduced the flag words of a sharded run: selection among the candidates, standalone kernel for the rest.)
int launch_spmm_csc_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                              int32_t n, int64_t nnz, int stat, double alpha, const double* alpha_div, double beta, double* S,
                              int64_t lds, uint32_t* flags, bool bounded, const double* xmax_dev, double xmax_host,
                              int64_t nnz_choice) {
  ctx->fmed.valid = false;
  ctx->fmed.token = 0;
  ctx->fmed.n = 0;
  constexpr int K = 256;                         // calibration columns
  const plaidhip_scatter_plan& sp = gs->scatter;
  int sm = sparse_mode(ctx);
  if (nnz_choice < 0) nnz_choice = nnz;   // (a shard of a larger call: the density of the WHOLE matrix picks the kernel)
  if (sm == 0 && nnz_choice >= 0) sm = (nnz_choice * 8 < (int64_t)gs->g * n) ? 1 : 2;
  const int32_t nslice = sp.nch * (kScatterBlock / 64);
  // candidate slots per (column, chunk, wavefront): fused_cand_per_column(m) per column over its slices
  const int32_t kCapC = std::max<int32_t>(32, (fused_cand_per_column(gs->m) / std::max<int32_t>(nslice, 1)) & ~15);
  // worth it from ~1e9 scores on (measured: the classifying epilogue costs 0.4 ms per 1e9 scores and the calibration
  // ~0.45 ms per call, the standalone median kernel 1.4 ms per 1e9 scores -- but it has a floor of ~1 ms as soon as a few
  // hundred columns are left to it; at 6e8 scores, the reference's pbmc3k shape, the plain pair is faster: 6.5 against
  // 7.7 ms for plaid() on counts, also with the selection kernel of late round 4; rank weights break even at 2e8)
  const bool big_enough = ctx->opt_fused_medians == 1 || (int64_t)gs->m * n >= 1000000000ll;
  const bool eligible = ctx->opt_fused_medians != 2 && big_enough && sm == 1 && nnz >= 0 && gs->m > 6144 && n >= 4 * K &&
                        flags != nullptr && nslice <= 256 && g_ablate == 0 && pair_kernel_mode(ctx) != 0 && !gs->pair.slices.empty();
  if (!eligible)
    return launch_spmm_csc_f64(ctx, gs, Xp, Xi, Xx, n, nnz_choice, stat, alpha, alpha_div, beta, S, lds, flags, bounded, xmax_dev,
                               xmax_host);
  // scratch: [pred n f64][cal 4 f64][medK K f64][flagsK 4 u32 (+pad)][status n i32 (+pad)][cnt n nslice 4 u32][cand n nslice capc u64]
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_pred = 0, o_cal = up(o_pred + (size_t)n * 8), o_medK = up(o_cal + 32), o_flagsK = up(o_medK + (size_t)K * 8),
               o_status = up(o_flagsK + 16), o_cnt = up(o_status + (size_t)n * 4), o_cand = up(o_cnt + (size_t)n * nslice * 16),
               total = o_cand + (size_t)n * nslice * kCapC * 8;
  if (ctx->fmed_bytes < total) {
    PH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->fmed_buf) PH_HIP(hipFree(ctx->fmed_buf));
    ctx->fmed_buf = nullptr;
    ctx->fmed_bytes = 0;
    if (hipMalloc(&ctx->fmed_buf, total) != hipSuccess) {   // no room for the candidate lists: the plain route
      (void)hipGetLastError();
      return launch_spmm_csc_f64(ctx, gs, Xp, Xi, Xx, n, nnz_choice, stat, alpha, alpha_div, beta, S, lds, flags, bounded, xmax_dev,
                                 xmax_host);
    }
    ctx->fmed_bytes = total;
  }
  char* base = static_cast<char*>(ctx->fmed_buf);
  double* pred = reinterpret_cast<double*>(base + o_pred);
  double* cal = reinterpret_cast<double*>(base + o_cal);
  double* medK = reinterpret_cast<double*>(base + o_medK);
  uint32_t* flagsK = reinterpret_cast<uint32_t*>(base + o_flagsK);
  PH_HIP(hipMemsetAsync(flagsK, 0, 16, ctx->stream));
  int rc = launch_spmm_scatter_csc_f64(ctx, gs, Xp, Xi, Xx, K, stat, alpha, alpha_div, beta, S, lds, flagsK, false, bounded, xmax_dev,
                                       xmax_host, nnz / n * K + 1);
  if (rc != PLAIDHIP_OK) return rc;
  rc = launch_col_medians(ctx, S, lds, gs->m, K, -1, flagsK, medK);
  if (rc != PLAIDHIP_OK) return rc;
  const int si = stat == PLAIDHIP_STAT_MEAN ? 0 : 1;
  rc = launch_colmean_predict(ctx, Xp, Xi, Xx, n, sp.d_u + (size_t)si * gs->g, alpha, alpha_div, beta * sp.kappa[si], pred);
  if (rc != PLAIDHIP_OK) return rc;
  rc = launch_median_calibrate(ctx, medK, pred, K, flagsK, cal);
  if (rc != PLAIDHIP_OK) return rc;
  plaidhip_scatter_med med{pred, cal, reinterpret_cast<unsigned long long*>(base + o_cand), reinterpret_cast<uint32_t*>(base + o_cnt), kCapC};
  rc = launch_spmm_scatter_csc_f64(ctx, gs, Xp, Xi, Xx, n, stat, alpha, alpha_div, beta, S, lds, flags, false, bounded, xmax_dev,
                                   xmax_host, nnz, &med);
  if (rc != PLAIDHIP_OK) return rc;
  ctx->fmed.valid = true;
  ctx->fmed.token = ++ctx->fmed_gen;
  ctx->fmed.S = S;
  ctx->fmed.lds = lds;
  ctx->fmed.m = gs->m;
  ctx->fmed.n = n;
  ctx->fmed.nslice = nslice;
  ctx->fmed.capc = kCapC;
  ctx->fmed.pred = pred;
  ctx->fmed.cal = cal;
  ctx->fmed.cnt = med.cnt;
  ctx->fmed.cand = med.cand;
  ctx->fmed.status = reinterpret_cast<int32_t*>(base + o_status);
  return PLAIDHIP_OK;
}

// ---- the DENSE crossprod that also selects the column medians of its result (round 5) -------------------------------
// Same idea and same resume call as launch_spmm_csc_fused_f64, for the pair kernel: the workgroup of a column pair stages
// every x[i, c] anyway, so it computes the pair's mean scores itself (alpha * sum_i x[i, c] u[i] + beta * kappa: no extra
// pass over X) before the last gene slice, whose tile ends classify the scores they write.  Applies when the fp64 pair
// kernel takes the input (not the u16 / fp32 stagings of rank inputs, not the MFMA backend), the result has more sets per
// column than the register-resident median kernel takes (m > 6,144) and is large enough for the calibration to pay
// (>= 1e9 scores, PLAIDHIP_OPT_FUSED_MEDIANS overrides); everything else runs the plain crossprod.
//   1. crossprod (classifying form, empty bracket) + standalone medians of the first K columns -> their predicted means too;
//   2. {offset, half width, ignore-zero rule} of the bracket (median_calibrate_kernel);
//   3. the crossprod of ALL columns with the classifying tile ends.
int launch_spmm_dense_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const double* X, int64_t ldx, int32_t n, int stat,
                                double alpha, const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags,
                                int x_kind) {
  ctx->fmed.valid = false;
  ctx->fmed.token = 0;
  ctx->fmed.n = 0;
  constexpr int K = 256;
  constexpr int32_t nslice = 16;                 // wavefronts of the pair kernel's workgroup
  const int32_t kCapC = std::max<int32_t>(32, (fused_cand_per_column(gs->m) / nslice) & ~15);   // candidate slots per (column, wavefront)
  const bool one_slice_16 = gs->slices.size() == 1 && gs->slices[0].waves == 16;
  const bool compact = (x_kind != PLAIDHIP_X_ANY && ctx->opt_ranks_f32 >= 1 && one_slice_16) ||
                       (ctx->precision == PLAIDHIP_PRECISION_MIXED && one_slice_16);
  const bool big_enough = ctx->opt_fused_medians == 1 || (int64_t)gs->m * n >= 1000000000ll;
  const bool eligible = ctx->opt_fused_medians != 2 && big_enough && gs->m > 6144 && n >= 4 * K && flags != nullptr &&
                        ctx->opt_dense_kernel != 3 && !compact && g_ablate == 0 && pair_kernel_mode(ctx) != 0 &&
                        !gs->pair.slices.empty() && gs->scatter.d_u != nullptr && gs->m > 0 && ctx->d_sel != nullptr;
  if (!eligible) return launch_spmm_dense_f64(ctx, gs, X, ldx, n, stat, alpha, alpha_div, beta, S, lds, flags, x_kind);
  // scratch: [pred n f64][cal 4 f64][medK K f64][flagsK 4 u32 (+pad)][status n i32 (+pad)][cnt n nslice 4 u32][cand n nslice capc u64]
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_pred = 0, o_cal = up(o_pred + (size_t)n * 8), o_medK = up(o_cal + 32),
               o_flagsK = up(o_medK + (size_t)K * 8), o_status = up(o_flagsK + 16), o_cnt = up(o_status + (size_t)n * 4),
               o_cand = up(o_cnt + (size_t)n * nslice * 16), total = o_cand + (size_t)n * nslice * kCapC * 8;
  if (ctx->fmed_bytes < total) {
    PH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->fmed_buf) PH_HIP(hipFree(ctx->fmed_buf));
    ctx->fmed_buf = nullptr;
    ctx->fmed_bytes = 0;
    if (hipMalloc(&ctx->fmed_buf, total) != hipSuccess) {   // no room for the candidate lists: the plain route
      (void)hipGetLastError();
      return launch_spmm_dense_f64(ctx, gs, X, ldx, n, stat, alpha, alpha_div, beta, S, lds, flags, x_kind);
    }
    ctx->fmed_bytes = total;
  }
  char* base = static_cast<char*>(ctx->fmed_buf);
  double* pred = reinterpret_cast<double*>(base + o_pred);
  double* cal = reinterpret_cast<double*>(base + o_cal);
  const double* cal0 = reinterpret_cast<const double*>(reinterpret_cast<const char*>(ctx->d_sel) + 96);   // the empty bracket (plaidhip_create)
  double* medK = reinterpret_cast<double*>(base + o_medK);
  uint32_t* flagsK = reinterpret_cast<uint32_t*>(base + o_flagsK);
  const int si = stat == PLAIDHIP_STAT_MEAN ? 0 : 1;
  const plaidhip_scatter_plan& sp = gs->scatter;
  plaidhip_pair_med med{sp.d_u + (size_t)si * gs->g, beta * sp.kappa[si], cal0, pred,
                        reinterpret_cast<unsigned long long*>(base + o_cand), reinterpret_cast<uint32_t*>(base + o_cnt), kCapC};
  PH_HIP(hipMemsetAsync(flagsK, 0, 16, ctx->stream));
  int rc = launch_colpair(ctx, gs, X, ldx, nullptr, nullptr, nullptr, K, stat, alpha, alpha_div, beta, S, lds, flagsK, false, nullptr, 0, &med);
  if (rc != PLAIDHIP_OK) return rc;
  rc = launch_col_medians(ctx, S, lds, gs->m, K, -1, flagsK, medK);
  if (rc != PLAIDHIP_OK) return rc;
  rc = launch_median_calibrate(ctx, medK, pred, K, flagsK, cal);
  if (rc != PLAIDHIP_OK) return rc;
  med.cal = cal;
  rc = launch_colpair(ctx, gs, X, ldx, nullptr, nullptr, nullptr, n, stat, alpha, alpha_div, beta, S, lds, flags, false, nullptr, 0, &med);
  if (rc != PLAIDHIP_OK) return rc;
  ctx->fmed.valid = true;
  ctx->fmed.token = ++ctx->fmed_gen;
  ctx->fmed.S = S;
  ctx->fmed.lds = lds;
  ctx->fmed.m = gs->m;
  ctx->fmed.n = n;
  ctx->fmed.nslice = nslice;
  ctx->fmed.capc = kCapC;
  ctx->fmed.pred = pred;
  ctx->fmed.cal = cal;
  ctx->fmed.cnt = med.cnt;
  ctx->fmed.cand = med.cand;
  ctx->fmed.status = reinterpret_cast<int32_t*>(base + o_status);
  return PLAIDHIP_OK;
}

// `token`: < 0 = the caller vouches that S is what the last fused launch on this context wrote and nothing touched it since
// (the library's own pipelines, which call the two back to back); otherwise the value plaidhip_dev_fused_medians_info gave
// after the fused call -- a stale or zero token takes the standalone kernels (always correct, one more pass over S).
int launch_col_medians_resume(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n, int ignore_zero,
                              const uint32_t* flags, double* med, int64_t token) {
  const auto& f = ctx->fmed;
  const bool mine = f.valid && f.S == S && f.lds == lds && f.m == m && f.n == n && (token < 0 || (uint64_t)token == f.token);
  if (!mine) {
    ctx->fmed.valid = false;   // (whatever was pending describes another S, or one the caller no longer vouches for)
    return launch_col_medians(ctx, S, lds, m, n, ignore_zero, flags, med);
  }
  ctx->fmed.valid = false;   // (consumed: S is about to be shifted)
  const int rc = launch_median_select(ctx, f.cand, f.cnt, n, f.nslice, f.capc, m, f.cal, ignore_zero, flags, med, f.status);
  if (rc != PLAIDHIP_OK) return rc;
  // (a workgroup per open column -- 17 short sweeps each -- instead of the streaming kernel's wavefront per column was
  // measured for the few dozen columns usually left: 1.36 against 1.18 ms for the whole phase at C3; not kept)
  return launch_col_medians(ctx, S, lds, m, n, ignore_zero, flags, med, f.status);
}

}  // namespace plaidhip
