// Bucket ranker: per-sample column ranks WITHOUT sorting (gfx950, wave64).
//
// colranks() / sparse_colranks() need, for every element x_i of a column,
//     lb = #{x_j < x_i}   and   ub = #{x_j <= x_i}
// (rank = lb + 1 | ub | (lb + 1 + ub) / 2 for ties.method min | max | average: R/plaid.R:611-619,
// 631-650) -- counts, not an order.  One workgroup per column:
//
//   1. the column is read ONCE, coalesced, into registers (KPT keys per thread) as order-preserving
//      u64 keys (exact IEEE order, -0 == +0, NaN set aside);
//   2. a coarse histogram over K1 equal key-space intervals of [min, max] gives the column's
//      empirical CDF; every coarse interval is cut into as many equal FINE buckets as it holds keys,
//      so there are as many fine buckets as keys and a key's fine bucket is an interpolated rank
//      estimate that is monotone in the key.  A second histogram + prefix sum over the fine buckets
//      gives `start` = #{keys in earlier fine buckets} (all of them smaller);
//   3. fine buckets hold ~1 key for any locally smooth distribution; what is left is settled exactly
//      inside the bucket: the keys of multi-key buckets are scattered to their bucket's slots in LDS
//      (8 bytes per key: a 20k-gene column fills the CU's LDS, which is why the histograms live in
//      the same LDS BEFORE the keys and everything a key needs afterwards rests in registers) and
//      each owner counts the keys of its bucket below / not above its own.
//   4. ties: a bucket with more than kBigT keys is first compared with one representative key; if
//      all its keys are equal (single-cell data: ~50 distinct values per column, or a dense column
//      that is 95 % zeros) the bounds are `start` and `start + count` with no scan at all.
//   5. a bucket that is large AND mixed (clustered values two histogram levels cannot separate)
//      would make the in-bucket scan quadratic: the column is handed to the sorting-network kernel
//      instead (device-side list, no host round trip).  Exactness never depends on the data.
//
// The owner of element i ends up with its bounds in registers, so the ranks go back to HBM as one
// coalesced write of the column in input order; fused: signed ranks, rank^power, column maximum.
#pragma once
#ifndef PH_POW
#define PH_POW pow
#endif

#include "common.h"
#include "device_sort.h"

namespace plaidhip {

constexpr int kBigT = 8;              // buckets with more keys are tested for "all keys equal"
constexpr int kFallbackCount = 256;   // a mixed bucket beyond this many keys sends the column to the network kernel
constexpr int kRankMisc = 1024;       // bytes of reduction scratch in front of the histograms / keys

struct RankBucketArgs {
  const double* Xv;          // values: dense matrix or CSC @x
  int64_t ldx;
  int32_t g_dense;           // dense: column length
  const int32_t* Xp;         // CSC column pointers (nullptr: dense)
  int32_t n;
  int32_t ties, is_signed;
  double power;
  int32_t pow_q4;            // 4 * power when that is an integer in 1..16 (power by square roots), else 0
  double* R;
  int64_t ldr;
  double* colmax;
  const int32_t* Xi_dense;   // non-null: CSC input, DENSE result (zeros ranked)
  double* dense_scratch;     // g_dense doubles per workgroup
  int32_t* fb_count;         // device counter + list of columns left to the network kernel
  int32_t* fb_list;
  unsigned long long* dbg;   // tools/ build only: per workgroup, cycles of wave 0 per phase (8 words)
};

#ifdef PLAIDHIP_DIAG
#define PH_STAMP(k)                                                                   \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                        \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                \
    t_ph[k] += t_ - t_last;                                                            \
    t_last = t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#else
#define PH_STAMP(k) do { } while (0)
#endif

// v_min_f64 / v_max_f64 as single instructions (fmin / fmax put a canonicalising v_max_f64 x, x in front of each)
__device__ __forceinline__ double ph_min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double ph_max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// r^(1/4) for a normal r > 0 (ranks are >= 0.5): two raw reciprocal square roots (v_rsq_f64, ~23 bits) and two steps of a
// simplified Newton iteration z += z (r - z^4) / (4 r) with 1 / (4 r) taken from the first estimate (its 2^-22 error only
// scales the correction: 2^-22 -> 2^-44 -> 2^-66): 12 fp64 instructions against ~30 for sqrt(sqrt(r)) -- the rank kernel is
// bound by its vector instructions (5.2k per wavefront and column, profiles/r06p_pmc_c4_summary.txt), and ssGSEA's
// alpha = 0.25 spent a fifth of them on the two correctly rounded square roots.  Within 1 ulp of sqrt(sqrt(r)).
__device__ __forceinline__ double root4_pos(double r) {
  const double y = __builtin_amdgcn_rsq(r);          // ~ r^-1/2
  double z = __builtin_amdgcn_rsq(y);                // ~ r^1/4
  const double c = 0.25 * (y * y);                   // ~ 1 / (4 r)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double w = z * z;
    const double e = __fma_rn(-w, w, r);             // r - z^4
    z = __fma_rn(z * c, e, z);
  }
  return z;
}

// r^(q/4) for r > 0 by (correctly rounded) square roots and multiplications: a few ulp, ~6x cheaper than pow()
__device__ __forceinline__ double pow_quarters(double r, int q) {
  double res = 1.0, base = r;
  for (int e = q >> 2; e; e >>= 1) {
    if (e & 1) res *= base;
    base *= base;
  }
  if ((q & 3) == 1) return res * root4_pos(r);       // 1.25 (ssGSEA's default exponent), 2.25, ...
  if (q & 3) {
    const double s = sqrt(r);
    if (q & 2) res *= s;
    if (q & 1) res *= sqrt(s);
  }
  return res;
}

// Wave-level scans / reductions on DPP (row shifts inside the four rows of 16 lanes, then the two row
// broadcasts): no lane-address registers to keep alive, unlike ds_bpermute-based shuffles.
#define PH_DPP(old, src, ctrl, rmask) \
  ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (rmask), 0xf, false))
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143;

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += PH_DPP(0u, v, kDppShr1, 0xf);
  v += PH_DPP(0u, v, kDppShr2, 0xf);
  v += PH_DPP(0u, v, kDppShr4, 0xf);
  v += PH_DPP(0u, v, kDppShr8, 0xf);
  v += PH_DPP(0u, v, kDppBcast15, 0xa);
  v += PH_DPP(0u, v, kDppBcast31, 0xc);
  return v;
}

template <bool IS_MIN>
__device__ __forceinline__ uint64_t wave_minmax_u64(uint64_t v) {   // result valid in lane 63
  const uint32_t idl = IS_MIN ? 0xffffffffu : 0u;
#define PH_STEP64(ctrl, rmask)                                                            \
  {                                                                                        \
    const uint32_t lo_ = PH_DPP(idl, (uint32_t)v, ctrl, rmask);                            \
    const uint32_t hi_ = PH_DPP(idl, (uint32_t)(v >> 32), ctrl, rmask);                    \
    const uint64_t o_ = ((uint64_t)hi_ << 32) | lo_;                                       \
    v = IS_MIN ? (o_ < v ? o_ : v) : (o_ > v ? o_ : v);                                    \
  }
  PH_STEP64(kDppShr1, 0xf) PH_STEP64(kDppShr2, 0xf) PH_STEP64(kDppShr4, 0xf) PH_STEP64(kDppShr8, 0xf)
  PH_STEP64(kDppBcast15, 0xa) PH_STEP64(kDppBcast31, 0xc)
#undef PH_STEP64
  return v;
}

__device__ __forceinline__ double wave_max_f64_dpp(double x) {   // result valid in lane 63; -inf identity
  uint64_t v = (uint64_t)__double_as_longlong(x);
#define PH_STEPF(ctrl, rmask)                                                             \
  {                                                                                        \
    const uint32_t lo_ = PH_DPP(0u, (uint32_t)v, ctrl, rmask);                             \
    const uint32_t hi_ = PH_DPP(0xfff00000u, (uint32_t)(v >> 32), ctrl, rmask);            \
    const double o_ = __longlong_as_double((long long)(((uint64_t)hi_ << 32) | lo_));      \
    const double c_ = __longlong_as_double((long long)v);                                  \
    v = (uint64_t)__double_as_longlong(o_ > c_ ? o_ : c_);                                 \
  }
  PH_STEPF(kDppShr1, 0xf) PH_STEPF(kDppShr2, 0xf) PH_STEPF(kDppShr4, 0xf) PH_STEPF(kDppShr8, 0xf)
  PH_STEPF(kDppBcast15, 0xa) PH_STEPF(kDppBcast31, 0xc)
#undef PH_STEPF
  return __longlong_as_double((long long)v);
}

// exclusive prefix sum over the workgroup's threads; s_wave: BLOCK/64 words of LDS; ends with a barrier
template <int BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave, uint32_t lane, uint32_t wave) {
  const uint32_t inc = wave_incl_scan_u32(v);
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w) {
    const uint32_t t = s_wave[w];
    base += ((uint32_t)w < wave) ? t : 0u;
  }
  __syncthreads();
  return base + inc - v;
}

template <int BLOCK, int KPT>
struct RankBucketLayout {
  static constexpr int CAP = BLOCK * KPT;
  static constexpr int K1 = CAP >= 16384 ? 4096 : (CAP >= 8192 ? 2048 : (CAP >= 4096 ? 1024 : 512));
  static constexpr int NBIG = ((CAP / (kBigT + 1) + 4) & ~3);
  static constexpr int off_c1 = kRankMisc;                   // K1 counters, [K1] end marker, [K1 + 1] trash
  static constexpr int off_c2 = off_c1 + (K1 + 4) * 4;       // CAP counters, [CAP] end marker, [CAP + 1] trash
  static constexpr int off_rep = off_c2 + (CAP + 4) * 4;     // NBIG u64
  static constexpr int off_mixed = off_rep + NBIG * 8;       // NBIG words
  static constexpr int hist_bytes = off_mixed + NBIG * 4;
  static constexpr int off_keys = kRankMisc;                 // 8 bytes per key, overlays the histograms
  static_assert(K1 % BLOCK == 0 && KPT % 4 == 0, "scan shapes");
  static_assert(hist_bytes % 16 == 0 && off_c2 % 16 == 0 && off_rep % 16 == 0, "alignment");
};

// Per-key state word after the fine prefix sum.
//   settled  (bit 31 = 0): bits 14:0 start, bits 30:16 count            -> lb = start, ub = start + count
//   to scan  (bit 31 = 1): bits 14:0 start, bits 22:15 slot, bits 30:23 count - 1 (count <= 256)
__device__ __forceinline__ uint32_t st_settled(uint32_t start, uint32_t count) { return start | (count << 16); }
__device__ __forceinline__ uint32_t st_scan(uint32_t start, uint32_t count, uint32_t slot) {
  return 0x80000000u | start | (slot << 15) | ((count - 1u) << 23);
}

template <int BLOCK, int KPT>
__global__ void __launch_bounds__(BLOCK)
colranks_bucket_kernel(RankBucketArgs a) {
  using L = RankBucketLayout<BLOCK, KPT>;
  constexpr int K1 = L::K1, CAP = L::CAP;
  constexpr int LOG2K1 = K1 == 4096 ? 12 : (K1 == 2048 ? 11 : (K1 == 1024 ? 10 : 9));
  constexpr int NW = BLOCK / 64;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // misc scratch: [0,64) scan words, [64,192) minima, [192,320) maxima, [384,512) f64 maxima, [512] flag
  uint32_t* s_wave = reinterpret_cast<uint32_t*>(smem_raw);
  uint64_t* s_min = reinterpret_cast<uint64_t*>(smem_raw + 64);
  uint64_t* s_max = reinterpret_cast<uint64_t*>(smem_raw + 192);
  double* s_f64 = reinterpret_cast<double*>(smem_raw + 384);
  uint32_t* s_flag = reinterpret_cast<uint32_t*>(smem_raw + 512);
  uint32_t* c1 = reinterpret_cast<uint32_t*>(smem_raw + L::off_c1);
  uint32_t* c2 = reinterpret_cast<uint32_t*>(smem_raw + L::off_c2);
  uint64_t* rep = reinterpret_cast<uint64_t*>(smem_raw + L::off_rep);
  uint32_t* mixed = reinterpret_cast<uint32_t*>(smem_raw + L::off_mixed);
  uint64_t* lkeys = reinterpret_cast<uint64_t*>(smem_raw + L::off_keys);

#ifdef PLAIDHIP_DIAG
  unsigned long long t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
  // PERSISTENT use (grid < columns): the NEXT column's values are requested into the key registers as soon as the
  // current column's keys are dead (behind the in-bucket counts) and land while the current column's ranks are written:
  // a 20k-gene column fills the LDS, so one workgroup owns a CU and nothing else would overlap its load and store phases.
  bool have_next = false;          // key[] holds the raw values of column c (requested during the previous column)
  const double* xc_n = nullptr;
  double* rc_n = nullptr;
  uint32_t cnt_n = 0;
  uint64_t key[KPT];
  for (int c = blockIdx.x; c < a.n; c += gridDim.x) {
    // an opaque copy of the thread id per column: nothing derived from it is hoisted out of the column loop
    // (LICM would keep dozens of per-thread addresses alive across all phases and spill)
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const uint32_t lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const double* xc;
    double* rc;
    uint32_t cnt;
    // descriptor of column cc (dense columns and CSC @x; the densified-CSC form builds its column below)
#define PH_COLUMN_DESC(cc, xp_, rp_, cn_)                                                                            \
    if (a.Xp != nullptr) {                                                                                             \
      /* (wave-uniform and read-only: through the scalar cache -- as a vector load + readfirstlane the column bounds  \
         are one more dependent L2 round trip in front of every column's value loads) */                              \
      typedef __attribute__((address_space(4))) const int32_t* cptr_i32_;                                              \
      const int p0_ = ((cptr_i32_)a.Xp)[cc];                                                                           \
      cn_ = (uint32_t)(((cptr_i32_)a.Xp)[(cc) + 1] - p0_);                                                             \
      xp_ = a.Xv + p0_;                                                                                                \
      rp_ = a.R + p0_;                                                                                                 \
    } else {                                                                                                           \
      cn_ = (uint32_t)a.g_dense;                                                                                       \
      xp_ = a.Xv + (int64_t)(cc) * a.ldx;                                                                              \
      rp_ = a.R + (int64_t)(cc) * a.ldr;                                                                               \
    }
    // every value of the column requested at once (clamped index instead of a guarded load: no exec games).
    // (Two neighbouring elements per lane -- 16-byte loads and stores, half the vector-memory instructions -- were
    // measured: no faster, A/B on one box 2.06 vs 2.03 ms per 8,192 columns; the phases are not bound by their issue.)
#define PH_COLUMN_LOAD(xp_, cn_)                                                                                     \
    {                                                                                                                  \
      const uint32_t last_ = (cn_) - 1u;                                                                               \
      _Pragma("unroll") for (int j0 = 0; j0 < KPT; j0 += 4) {                                                        \
        if ((uint32_t)(j0) * BLOCK < (cn_)) {                                                                          \
          _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
            const uint32_t i_ = tid + (uint32_t)(j0 + u) * BLOCK;                                                      \
            key[j0 + u] = (uint64_t)__double_as_longlong(__builtin_nontemporal_load((xp_) + (i_ < (cn_) ? i_ : last_))); \
          }                                                                                                            \
        } else {                                                                                                       \
          _Pragma("unroll") for (int u = 0; u < 4; ++u) key[j0 + u] = ~0ull;                                         \
        }                                                                                                              \
      }                                                                                                                \
    }
    if (a.Xi_dense != nullptr) {
      // colranks(sparse X, keep.zero = FALSE): the reference ranks the densified column (R/plaid.R:603-609)
      double* dcol = a.dense_scratch + (int64_t)blockIdx.x * a.g_dense;
      for (int i = (int)tid; i < a.g_dense; i += BLOCK) dcol[i] = 0.0;
      __syncthreads();
      const int p0 = a.Xp[c], p1 = a.Xp[c + 1];
      for (int p = p0 + (int)tid; p < p1; p += BLOCK) dcol[a.Xi_dense[p]] = a.Xv[p];
      __syncthreads();
      cnt = (uint32_t)a.g_dense;
      xc = dcol;
      rc = a.R + (int64_t)c * a.ldr;
    } else if (have_next) {
      xc = xc_n; rc = rc_n; cnt = cnt_n;
    } else {
      PH_COLUMN_DESC(c, xc, rc, cnt)
    }
    // The KPT items of a thread are handled in groups of four; a group takes part when any of its 4 * BLOCK
    // elements exists (the same for every thread: no divergence, straight-line code inside a group).  Lanes
    // past the end and NaN are "invalid": they run the same instructions on trash slots of the histograms.
#define PH_GROUP(j0) ((uint32_t)(j0) * BLOCK < cnt)
#define PH_FOR_ITEMS(...)                                        \
  _Pragma("unroll") for (int j0 = 0; j0 < KPT; j0 += 4) {        \
    if (PH_GROUP(j0)) {                                          \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {            \
        const int j = j0 + u;                                    \
        __VA_ARGS__                                              \
      }                                                          \
    }                                                            \
  }

    // ---- 1. the column -> registers as ordered keys ------------------------------------------------
    uint64_t validmask = 0, nanmask = 0, negmask = 0;
    uint64_t kmin = ~0ull, kmax = 0ull;
    // (the extremes are taken on the DOUBLES -- v_min_f64 / v_max_f64, one instruction each per key; the key map is monotone --
    // and turned into keys once per thread: the u64 compare-and-select pairs per key were 8 vector instructions)
    double dmin = INFINITY, dmax = -INFINITY;
    {
      if (!have_next) PH_COLUMN_LOAD(xc, cnt)
      have_next = false;
      // zero the histograms (and the tie flags) while the loads are in flight
      {
        uint4* z = reinterpret_cast<uint4*>(smem_raw + L::off_c1);
        constexpr int NZ = (L::hist_bytes - L::off_c1) / 16;
        for (int i = (int)tid; i < NZ; i += BLOCK) z[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0) *s_flag = 0;
      }
      PH_FOR_ITEMS({
        const uint32_t i = tid + (uint32_t)j * BLOCK;
        const bool inr = i < cnt;
        double xv = __longlong_as_double((long long)key[j]);
        negmask |= (inr && xv < 0.0) ? (1ull << j) : 0u;
        if (a.is_signed) xv = fabs(xv);
        const bool isnan_ = xv != xv;
        nanmask |= (inr && isnan_) ? (1ull << j) : 0u;
        const bool ok = inr && !isnan_;
        validmask |= ok ? (1ull << j) : 0u;
        const uint64_t u = (uint64_t)__double_as_longlong(xv + 0.0);       // -0 -> +0
        const uint64_t k = ((long long)u < 0) ? ~u : (u | 0x8000000000000000ull);
        key[j] = ok ? k : ~0ull;
        const double xn = ok ? xv + 0.0 : dmin;                              // (a key that does not count repeats the minimum)
        dmin = ph_min_f64(dmin, xn);
        dmax = ph_max_f64(dmax, ok ? xv + 0.0 : dmax);
      })
    }
    if (validmask != 0) {   // (per thread: the extremes of its own valid keys, as keys)
      const uint64_t u1 = (uint64_t)__double_as_longlong(dmin), u2 = (uint64_t)__double_as_longlong(dmax);
      kmin = ((long long)u1 < 0) ? ~u1 : (u1 | 0x8000000000000000ull);
      kmax = ((long long)u2 < 0) ? ~u2 : (u2 | 0x8000000000000000ull);
    }
    kmin = wave_minmax_u64<true>(kmin);
    kmax = wave_minmax_u64<false>(kmax);
    if (lane == 63) { s_min[wave] = kmin; s_max[wave] = kmax; }
    __syncthreads();
    kmin = ~0ull;
    kmax = 0ull;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint64_t o1 = s_min[w], o2 = s_max[w];
      kmin = o1 < kmin ? o1 : kmin;
      kmax = o2 > kmax ? o2 : kmax;
    }
    PH_STAMP(0);   // load + convert + min / max
    const uint64_t lo = kmin;                                         // ~0 when the column has no real key at all
    const uint64_t range = kmax > kmin ? kmax - kmin : 0ull;
    const int rbits = range ? 64 - __clzll((long long)range) : 0;
    // A key's place in [lo, lo + range] as a 32-bit fraction d32 = (key - lo) scaled to 32 bits (the top 32 bits of a longer
    // offset, all bits of a shorter one): monotone in the key, which is all the buckets need -- the coarse interval is its top
    // LOG2K1 bits, the position inside the interval the 32 - LOG2K1 bits below (round 6: the 64-bit shifts, products and
    // differences per key of the former form were a third of the fine-bucket phase's vector instructions)
    const int d_shr = rbits > 32 ? rbits - 32 : 0;
    const int d_shl = rbits > 32 ? 0 : (rbits > 0 ? 32 - rbits : 0);
#define PH_D32(k_) ((uint32_t)(((k_) - lo) >> d_shr) << d_shl)

    // ---- 2a. coarse histogram over key space ---------------------------------------------------------
    PH_FOR_ITEMS({
      const uint32_t b = PH_D32(key[j]) >> (32 - LOG2K1);
      atomicAdd(&c1[((validmask >> j) & 1ull) ? b : (uint32_t)(K1 + 1)], 1u);
    })
    __syncthreads();
    {
      constexpr int PER = K1 / BLOCK;
      uint32_t s = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) s += c1[tid * PER + q];
      uint32_t ex = block_excl_scan<BLOCK>(s, s_wave, lane, wave);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const uint32_t v = c1[tid * PER + q];
        c1[tid * PER + q] = ex;
        ex += v;
      }
      if (tid == BLOCK - 1) c1[K1] = ex;    // end marker = number of valid keys
    }
    __syncthreads();

    PH_STAMP(1);   // coarse histogram + scan
    // ---- 2b. fine bucket = CDF(coarse) + interpolation inside the coarse interval -----------------------
    uint32_t st[KPT];    // first: fine id | slot << 16; then the state word above
#pragma unroll
    for (int j = 0; j < KPT; ++j) st[j] = 0;
    PH_FOR_ITEMS({
      const bool ok = (validmask >> j) & 1u;
      const uint32_t d32 = ok ? PH_D32(key[j]) : 0u;
      const uint32_t b = d32 >> (32 - LOG2K1);
      const uint32_t cb = c1[b], cn = c1[b + 1] - cb;
      const uint32_t fr = d32 << LOG2K1;                               // the position inside the coarse interval, 32-bit fraction
      uint32_t f = cb + __umulhi(fr, cn);
      f = ok ? f : (uint32_t)(CAP + 1);
      const uint32_t slot = atomicAdd(&c2[f], 1u);
      st[j] = f | (slot << 16);
    })
    __syncthreads();
    {
      // exclusive scan of the fine counts, packed with the number of "big" buckets in the high half
      uint4* c2v = reinterpret_cast<uint4*>(c2) + tid * (KPT / 4);
      uint32_t s = 0;
#pragma unroll
      for (int q = 0; q < KPT / 4; ++q) {
        const uint4 v = c2v[q];
        s += v.x + v.y + v.z + v.w;
        s += (v.x > (uint32_t)kBigT ? 0x10000u : 0u) + (v.y > (uint32_t)kBigT ? 0x10000u : 0u) +
             (v.z > (uint32_t)kBigT ? 0x10000u : 0u) + (v.w > (uint32_t)kBigT ? 0x10000u : 0u);
      }
      uint32_t ex = block_excl_scan<BLOCK>(s, s_wave, lane, wave);
#pragma unroll
      for (int q = 0; q < KPT / 4; ++q) {
        const uint4 v = c2v[q];
        uint4 o;
        o.x = ex; ex += v.x + (v.x > (uint32_t)kBigT ? 0x10000u : 0u);
        o.y = ex; ex += v.y + (v.y > (uint32_t)kBigT ? 0x10000u : 0u);
        o.z = ex; ex += v.z + (v.z > (uint32_t)kBigT ? 0x10000u : 0u);
        o.w = ex; ex += v.w + (v.w > (uint32_t)kBigT ? 0x10000u : 0u);
        c2v[q] = o;
      }
      if (tid == BLOCK - 1) c2[CAP] = ex;
    }
    __syncthreads();
    PH_STAMP(2);   // fine histogram + scan
    uint64_t bigmask = 0;
    PH_FOR_ITEMS({
      const bool ok = (validmask >> j) & 1u;
      const uint32_t f = st[j] & 0xffffu, slot = st[j] >> 16;
      const uint32_t p0 = c2[f], p1 = c2[f + 1];
      const uint32_t start = p0 & 0xffffu, count = (p1 - p0) & 0xffffu;
      const bool big = ok && count > (uint32_t)kBigT;
      if (big) {
        bigmask |= 1ull << j;              // keeps (fine id, slot) until the tie test below is through
        rep[p0 >> 16] = key[j];          // any key of the bucket (racing stores of whole 8-byte words)
      }
      const uint32_t w = count > 1u ? st_scan(start, count, slot & 0xffu) : st_settled(start, 1u);
      st[j] = big ? st[j] : (ok ? w : 0u);
    })
    __syncthreads();
    if (bigmask) {
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        if ((bigmask >> j) & 1u) {
          const uint32_t bi = c2[st[j] & 0xffffu] >> 16;
          if (rep[bi] != key[j]) mixed[bi] = 1u;
        }
    }
    __syncthreads();
    if (bigmask) {
      bool giveup = false;
#pragma unroll
      for (int j = 0; j < KPT; ++j)
        if ((bigmask >> j) & 1u) {
          const uint32_t f = st[j] & 0xffffu, slot = st[j] >> 16;
          const uint32_t p0 = c2[f], p1 = c2[f + 1];
          const uint32_t start = p0 & 0xffffu, count = (p1 - p0) & 0xffffu;
          if (mixed[p0 >> 16] == 0u) {
            st[j] = st_settled(start, count);                  // every key of the bucket is this key
          } else if (count <= (uint32_t)kFallbackCount) {
            st[j] = st_scan(start, count, slot);
          } else {
            st[j] = 0;
            giveup = true;
          }
        }
      if (giveup) *s_flag = 1u;
    }
    __syncthreads();                        // everyone is done with the histograms: the keys may overwrite them
    if (*s_flag != 0u) {
      if (tid == 0) a.fb_list[atomicAdd(a.fb_count, 1)] = c;
      __syncthreads();
      continue;          // (have_next is false here: the next column is loaded at its own start)
    }

    PH_STAMP(3);   // bucket state + tie test
    // ---- 3. keys of multi-key, not-all-equal buckets -> their bucket's slots in LDS; count inside ----------
    PH_FOR_ITEMS({
      if (st[j] >> 31) lkeys[(st[j] & 0x7fffu) + ((st[j] >> 15) & 0xffu)] = key[j];
    })
    if (tid == 0) lkeys[-1] = ~0ull;   // the probes' sentinel (the last 8 bytes of the reduction scratch in front of the keys)
    __syncthreads();
    // the in-bucket counts first (the keys die here), then the output pass
    uint64_t zeromask = 0;
    PH_FOR_ITEMS({
      // the first four slots of the bucket are probed with all four reads in flight (and the four items of a group
      // interleave: nothing here branches); longer buckets -- rare unless the column has close clusters -- finish
      // in a loop.  A settled key probes nothing (cs = 0).
      const bool scan = st[j] >> 31;
      const uint32_t start = st[j] & 0x7fffu;
      const uint32_t cs = scan ? ((st[j] >> 23) & 0xffu) + 1u : 0u;
      const uint64_t k = key[j];
      uint32_t less = 0, leq = 0;
      _Pragma("unroll") for (uint32_t s = 0; s < 4; ++s) {
        // (a slot the bucket does not have reads the sentinel in front of the keys -- the largest u64, above every valid
        // key -- so that neither count needs a mask)
        const uint64_t o = lkeys[(int32_t)(s < cs ? start + s : 0xffffffffu)];
        less += (o < k) ? 1u : 0u;
        leq += (o <= k) ? 1u : 0u;
      }
      if (cs > 4u) {
        for (uint32_t s = 4; s < cs; ++s) {
          const uint64_t o = lkeys[start + s];
          less += (o < k) ? 1u : 0u;
          leq += (o <= k) ? 1u : 0u;
        }
      }
      const uint32_t lb = start + less;
      const uint32_t ub = scan ? start + leq : start + ((st[j] >> 16) & 0x7fffu);
      st[j] = lb | (ub << 16);
      zeromask |= (k == 0x8000000000000000ull) ? (1ull << j) : 0u;
    })
    PH_STAMP(4);   // scatter + in-bucket counts
    {
      // the keys are dead: their registers take the next column's values, which land during the output pass below
      // (only for the shapes whose keys fill most of the LDS: smaller columns overlap through occupancy, and the extra
      // live state would cost the 256-thread kernels a wavefront per SIMD)
      constexpr bool kPrefetch = (size_t)CAP * 8 > (size_t)64 * 1024;
      const int cnx = c + (int)gridDim.x;
      if (kPrefetch && a.Xi_dense == nullptr && cnx < a.n) {
        PH_COLUMN_DESC(cnx, xc_n, rc_n, cnt_n)
        if (cnt_n > 0) {
          PH_COLUMN_LOAD(xc_n, cnt_n)
          have_next = true;
        }
      }
    }
    double vmax = (a.Xp != nullptr && a.Xi_dense == nullptr) ? 0.0 : -INFINITY;   // sparse ranks: the implicit zeros
    const bool generic_pow = a.pow_q4 == 0 && a.power != 1.0;   // uniform
    if (generic_pow) __syncthreads();                           // every in-bucket count is done: the LDS is free again
    double* lrank = reinterpret_cast<double*>(lkeys);
    PH_FOR_ITEMS({
      const uint32_t i = tid + (uint32_t)j * BLOCK;
      const uint32_t lb = st[j] & 0xffffu, ub = st[j] >> 16;
      double r = (a.ties == PLAIDHIP_TIES_MIN) ? (double)(lb + 1)
                                               : ((a.ties == PLAIDHIP_TIES_MAX) ? (double)ub : 0.5 * (double)(lb + 1 + ub));
      if (a.pow_q4 > 0) r = pow_quarters(r, a.pow_q4);
      const bool ok = (validmask >> j) & 1u;
      if (!generic_pow) vmax = (ok && r > vmax) ? r : vmax;     // max |value| of the column (before the sign goes on)
      if (a.is_signed) r = ((zeromask >> j) & 1u) ? 0.0 : (((negmask >> j) & 1u) ? -r : r);
      r = ((nanmask >> j) & 1u) ? __longlong_as_double(0x7ff8000000000000ll) : r;
      if (i < cnt) {
        if (generic_pow) lrank[i] = r;                          // (signed) plain rank; the power goes on below
        else __builtin_nontemporal_store(r, rc + i);
      }
    })
    if (generic_pow) {
      // rank^power for an arbitrary exponent: ONE copy of pow() in a rolled loop over this thread's own ranks
      // (staged in LDS; unrolled KPT times it would not fit the register file)
      for (uint32_t i = tid; i < cnt; i += BLOCK) {
        const double v = lrank[i];
        double r = v;
        if (v == v && v != 0.0) {
          const double pr = PH_POW(fabs(v), a.power);
          vmax = pr > vmax ? pr : vmax;
          r = v < 0.0 ? -pr : pr;
        }
        __builtin_nontemporal_store(r, rc + i);
      }
    }
    if (a.colmax != nullptr) {
      vmax = wave_max_f64_dpp(vmax);
      if (lane == 63) s_f64[wave] = vmax;
      __syncthreads();
      if (tid == 0) {
        double v = s_f64[0];
        for (int w = 1; w < NW; ++w) v = s_f64[w] > v ? s_f64[w] : v;
        a.colmax[c] = v;
      }
    }
    __syncthreads();
    PH_STAMP(5);   // ranks -> power -> store (+ column maximum)
#undef PH_D32
#undef PH_GROUP
#undef PH_FOR_ITEMS
#undef PH_COLUMN_DESC
#undef PH_COLUMN_LOAD
  }
#ifdef PLAIDHIP_DIAG
  if (a.dbg != nullptr && threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) a.dbg[(size_t)blockIdx.x * 8 + k] = t_ph[k];
#endif
}

}  // namespace plaidhip
