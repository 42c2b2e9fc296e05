// normalize_medians() (R/plaid.R:554-575) as three device phases, plus the small
// reductions a sample-sharded host needs between them.  gfx950 / wave64 only.
//
//   minflags    : min(x, na.rm=TRUE) == 0  <=>  HAS_ZERO && !HAS_NEG        (R/plaid.R:556-557)
//   col_medians : per-sample median over gene sets; exact zeros masked when ignore_zero
//                 (R/plaid.R:562-565), all-masked column -> 0 (R/plaid.R:566).  Even count:
//                 mean of the two middle order statistics (matrixStats::colMedians).
//   shift       : (x - med[col]) + add, add = mean(medx)                      (R/plaid.R:572)
#include <cmath>
#include <cstdlib>

#include "common.h"

#include <type_traits>
#include "device_sort.h"

namespace plaidhip {

__global__ void __launch_bounds__(256)
minflags_kernel(const double* __restrict__ S, int64_t count, uint32_t* flags) {
  uint32_t f = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    const double v = S[i];
    f |= (v < 0.0) ? PLAIDHIP_FLAG_HAS_NEG : 0u;
    f |= (v == 0.0) ? PLAIDHIP_FLAG_HAS_ZERO : 0u;
    f |= (v != v) ? PLAIDHIP_FLAG_HAS_NAN : 0u;
  }
  for (int off = 32; off >= 1; off >>= 1) f |= __shfl_xor(f, off, 64);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if ((f >> b) & 1u) {
        if (__hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
          __hip_atomic_store(&flags[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  }
}

// ignore.zero resolved on the device: explicit 0/1, or (-1) min(x)==0 from the flag words
__device__ __forceinline__ int resolve_ignore_zero(int ignore_zero, const uint32_t* flags) {
  if (ignore_zero >= 0) return ignore_zero;
  return (flags[1] != 0u && flags[0] == 0u) ? 1 : 0;
}

// order-preserving key of v, or the all-ones key when v is masked (NaN: na.rm = TRUE; exact zero
// when ignore_zero).  Branch-free on purpose: with control flow hipcc waits for every load
// before issuing the next one and the column sweeps serialise on L2 latency.
__device__ __forceinline__ uint64_t masked_key(double v, int ignore_zero) {
  const uint64_t u = (uint64_t)__double_as_longlong(v + 0.0);          // -0 -> +0
  const uint64_t key = u ^ ((u >> 63) ? ~0ull : 0x8000000000000000ull);
  const bool masked = (v != v) | ((ignore_zero != 0) & (v == 0.0));
  return masked ? ~0ull : key;
}

// m <= kMaxLdsGenes: the column is sorted in LDS.
__global__ void __launch_bounds__(1024)
col_medians_lds_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                       int ignore_zero_mode, const uint32_t* __restrict__ flags,
                       double* __restrict__ med, int32_t key_slots) {
  const int ignore_zero = resolve_ignore_zero(ignore_zero_mode, flags);
  extern __shared__ __align__(16) unsigned char smem_raw[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
  uint32_t* s_u32 = reinterpret_cast<uint32_t*>(smem_raw + (size_t)key_slots * 8);
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const double* sc = S + (int64_t)c * lds;
    if (tid == 0) s_u32[0] = 0;
    __syncthreads();
    uint32_t masked = 0;
    for (int i = tid; i < m; i += nthr) {
      const uint64_t k = masked_key(sc[i], ignore_zero);
      masked += (k == ~0ull);
      keys[i] = k;
    }
    if (masked) atomicAdd(&s_u32[0], masked);
    bitonic_sort_lds(keys, (uint32_t)m);
    if (tid == 0) {
      const uint32_t cnt = (uint32_t)m - s_u32[0];
      double r;
      if (cnt == 0) {
        r = ignore_zero ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
      } else if (cnt & 1) {
        r = key_to_f64(keys[cnt >> 1]);
      } else {
        r = 0.5 * (key_to_f64(keys[(cnt >> 1) - 1]) + key_to_f64(keys[cnt >> 1]));
      }
      med[c] = r;
    }
    __syncthreads();
  }
}

// m <= BLOCK*ITEMS: register-resident bitwise selection.  The column is read coalesced; every
// thread keeps ITEMS 32-bit key words in registers and the k-th order statistic is found by
// binary search on the key VALUE, one bit per pass: count(word < candidate) is ITEMS x
// (v_cmp + ballot popcount) per wave plus one tiny cross-wave sum.  64-bit keys are resolved
// in two 32-pass phases (high words, then the low words of the keys that share the selected
// high word), so the register cost is one dword per element.  For an even count the second
// middle value is either the same key (ties) or the smallest key above it (one more sweep).
template <int BLOCK, int ITEMS>
__global__ void __launch_bounds__(BLOCK)
col_medians_bits_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                        int ignore_zero_mode, const uint32_t* __restrict__ flags,
                        double* __restrict__ med) {
  constexpr int NW = BLOCK / 64;
  __shared__ uint32_t s_cnt[2][NW];
  __shared__ uint32_t s_mm[2][2][NW];
  __shared__ unsigned long long s_min[NW];
  const int ignore_zero = resolve_ignore_zero(ignore_zero_mode, flags);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int parity = 0;
  uint32_t w[ITEMS];

  // block-wide count of words below `cand` (one barrier; the two s_cnt buffers alternate)
  auto count_below = [&](uint32_t cand, bool inclusive) -> uint32_t {
    uint32_t wc = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
      wc += (uint32_t)__popcll(__ballot(inclusive ? (w[j] <= cand) : (w[j] < cand)));
    if (lane == 0) s_cnt[parity][wave] = wc;
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) tot += s_cnt[parity][k];
    parity ^= 1;
    return tot;
  };
  // block-wide {min, max} of per-thread values (same alternating-buffer discipline)
  auto block_minmax = [&](uint32_t mn, uint32_t mx, uint32_t& omn, uint32_t& omx) {
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t a = __shfl_xor(mn, off, 64), b2 = __shfl_xor(mx, off, 64);
      mn = a < mn ? a : mn;
      mx = b2 > mx ? b2 : mx;
    }
    if (lane == 0) { s_mm[parity][0][wave] = mn; s_mm[parity][1][wave] = mx; }
    __syncthreads();
    omn = 0xffffffffu; omx = 0u;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      omn = s_mm[parity][0][k] < omn ? s_mm[parity][0][k] : omn;
      omx = s_mm[parity][1][k] > omx ? s_mm[parity][1][k] : omx;
    }
    parity ^= 1;
  };
  // value of rank k (0-based) among the words, all of which lie in [mn, mx] (inactive words are
  // 0xffffffff): the bits above the highest bit in which mn and mx differ are common to every
  // candidate, so the binary search on the value starts below them -- and is skipped entirely when
  // mn == mx (one key, or all ties).
  auto select_word = [&](uint32_t k, uint32_t mn, uint32_t mx) -> uint32_t {
    if (mn == mx) return mn;
    const int top = 31 - __clz((int)(mn ^ mx));
    uint32_t v = (top == 31) ? 0u : (mn & ~((2u << top) - 1u));
    for (int bit = top; bit >= 0; --bit) {
      const uint32_t cand = v | (1u << bit);
      if (count_below(cand, false) <= k) v = cand;
    }
    return v;
  };

  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const double* sc = S + (int64_t)c * lds;
    // ---- the column is read ONCE: 64-bit keys stay in registers (ITEMS <= 32), the 32-bit
    //      working words w[] are re-derived from them per phase ---------------------------
    constexpr bool KEEP = ITEMS <= 32;
    uint32_t khi[KEEP ? ITEMS : 1], klo[KEEP ? ITEMS : 1];
    uint32_t tmn = 0xffffffffu, tmx = 0u;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int i = tid + j * BLOCK;
      const uint64_t key = (i < m) ? masked_key(sc[i < m ? i : m - 1], ignore_zero) : ~0ull;
      w[j] = (uint32_t)(key >> 32);
      if constexpr (KEEP) { khi[j] = w[j]; klo[j] = (uint32_t)key; }
      const bool valid = key != ~0ull;
      tmn = (valid && w[j] < tmn) ? w[j] : tmn;
      tmx = (valid && w[j] > tmx) ? w[j] : tmx;
      if (ITEMS > 32 && (j & 15) == 15) asm volatile("" ::: "memory");   // bound the live 64-bit temporaries
    }
    uint32_t hmn, hmx;
    block_minmax(tmn, tmx, hmn, hmx);
    const uint32_t cnt = count_below(0xffffffffu, false);   // valid keys never have an all-ones high word
    double r;
    if (cnt == 0) {
      r = ignore_zero ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
    } else {
      const uint32_t k_lo = (cnt - 1) >> 1, k_hi = cnt >> 1;
      const uint32_t H = select_word(k_lo, hmn, hmx);
      const uint32_t below_H = count_below(H, false);
      // ---- phase 2: low words of the keys whose high word is H --------------------
      tmn = 0xffffffffu; tmx = 0u;
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        uint32_t hw, lw;
        if constexpr (KEEP) { hw = khi[j]; lw = klo[j]; }
        else {
          const int i = tid + j * BLOCK;
          const uint64_t key = (i < m) ? masked_key(sc[i < m ? i : m - 1], ignore_zero) : ~0ull;
          hw = (uint32_t)(key >> 32); lw = (uint32_t)key;
          if ((j & 15) == 15) asm volatile("" ::: "memory");
        }
        const bool act = (hw == H) && !(hw == 0xffffffffu && lw == 0xffffffffu);
        w[j] = act ? lw : 0xffffffffu;
        tmn = (act && lw < tmn) ? lw : tmn;
        tmx = (act && lw > tmx) ? lw : tmx;
      }
      uint32_t lmn, lmx;
      block_minmax(tmn, tmx, lmn, lmx);
      const uint32_t L = select_word(k_lo - below_H, lmn, lmx);
      const uint64_t V = ((uint64_t)H << 32) | L;
      uint64_t V2 = V;
      if (k_hi != k_lo) {
        // keys <= V: below_H + (same high word, low word <= L).  A key with another high word
        // carries 0xffffffff here and is only (wrongly) counted when L is 0xffffffff itself;
        // then the sweep below counts exactly.
        uint32_t le_V = 0;
        if (L != 0xffffffffu) le_V = below_H + count_below(L, true);
        const bool need_sweep = (L == 0xffffffffu) || (le_V <= k_hi);
        if (need_sweep) {
          // ---- phase 3: smallest key above V (and the exact count of keys <= V) ----
          uint64_t mn = ~0ull;
          uint32_t le = 0;
#pragma unroll
          for (int j = 0; j < ITEMS; ++j) {
            uint64_t key;
            if constexpr (KEEP) key = ((uint64_t)khi[j] << 32) | klo[j];
            else {
              const int i = tid + j * BLOCK;
              key = (i < m) ? masked_key(sc[i < m ? i : m - 1], ignore_zero) : ~0ull;
              if ((j & 15) == 15) asm volatile("" ::: "memory");
            }
            le += (uint32_t)__popcll(__ballot(key <= V));
            mn = (key > V && key < mn) ? key : mn;
          }
          for (int off = 32; off >= 1; off >>= 1) {
            const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)mn, off, 64);
            mn = o < mn ? o : mn;
          }
          if (lane == 0) { s_min[wave] = mn; s_cnt[parity][wave] = le; }
          __syncthreads();
          uint64_t bm = ~0ull;
          uint32_t tot = 0;
#pragma unroll
          for (int k = 0; k < NW; ++k) { bm = s_min[k] < bm ? s_min[k] : bm; tot += s_cnt[parity][k]; }
          parity ^= 1;
          V2 = (tot > k_hi) ? V : bm;
          __syncthreads();   // s_min is reused by the next column
        }
      }
      r = (V2 == V) ? key_to_f64(V) : 0.5 * (key_to_f64(V) + key_to_f64(V2));
    }
    if (tid == 0) med[c] = r;
  }
}

// Any m: 8-bit MSD radix select over the column in global memory (L2-resident).
// Selects order statistic `kth` (0-based) among unmasked keys.
__device__ uint64_t radix_select_global(const double* sc, int32_t m, int ignore_zero, uint32_t kth,
                                        uint32_t* hist /*256*/, uint32_t* s_sel /*2*/) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  uint64_t prefix = 0, mask = 0;
  for (int pass = 7; pass >= 0; --pass) {
    for (int b = tid; b < 256; b += nthr) hist[b] = 0;
    __syncthreads();
    const int shift = pass * 8;
    for (int i = tid; i < m; i += nthr) {
      const uint64_t k = masked_key(sc[i], ignore_zero);
      if (k != ~0ull && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t cum = 0, b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] > kth) break;
        cum += hist[b];
      }
      s_sel[0] = b;
      s_sel[1] = kth - cum;
    }
    __syncthreads();
    prefix |= (uint64_t)s_sel[0] << shift;
    mask |= 0xffull << shift;
    kth = s_sel[1];
    __syncthreads();
  }
  return prefix;
}

__global__ void __launch_bounds__(1024)
col_medians_select_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                          int ignore_zero_mode, const uint32_t* __restrict__ flags,
                          double* __restrict__ med) {
  const int ignore_zero = resolve_ignore_zero(ignore_zero_mode, flags);
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_sel[2];
  __shared__ uint32_t s_cnt;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const double* sc = S + (int64_t)c * lds;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    uint32_t valid = 0;
    for (int i = tid; i < m; i += nthr) valid += (masked_key(sc[i], ignore_zero) != ~0ull);
    for (int off = 32; off >= 1; off >>= 1) valid += __shfl_xor(valid, off, 64);
    if ((tid & 63) == 0 && valid) atomicAdd(&s_cnt, valid);
    __syncthreads();
    const uint32_t cnt = s_cnt;
    double r;
    if (cnt == 0) {
      r = ignore_zero ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
    } else if (cnt & 1) {
      r = key_to_f64(radix_select_global(sc, m, ignore_zero, cnt >> 1, hist, s_sel));
    } else {
      const double lo = key_to_f64(radix_select_global(sc, m, ignore_zero, (cnt >> 1) - 1, hist, s_sel));
      const double hi = key_to_f64(radix_select_global(sc, m, ignore_zero, cnt >> 1, hist, s_sel));
      r = 0.5 * (lo + hi);
    }
    if (tid == 0) med[c] = r;
    __syncthreads();
  }
}

// Sample-bracket selection: the fast median path.  One workgroup per column, keys streamed
// from global memory (the column was just written by the SpMM: L2 / Infinity Cache resident).
//   1. count the valid keys c; take s = BLOCK*SP keys at a fixed stride as a sample, sort it
//      in LDS;
//   2. the sample quantiles 4 sigma either side of the middle rank give a bracket [lo, hi]
//      that contains the middle order statistics with probability > 0.9999;
//   3. one sweep counts keys < lo, == lo, == hi and collects the keys strictly inside
//      (about 4/sqrt(s) of the column) into LDS, which are sorted there;
//   4. the middle ranks are read off the segments [<lo][==lo][inside][==hi].  A miss (rank
//      outside the bracket, or more inside keys than fit) falls back to the exact radix
//      select -- the result is exact either way; ties are handled by the == counters.
// ITEMS > 0: the column's keys are loaded ONCE into ITEMS 64-bit registers per thread (all
// loads in flight together) and every sweep runs from registers (m <= BLOCK*ITEMS).
// ITEMS == 0: keys are streamed from L2 in batches of 8 independent loads per thread.
template <int BLOCK, int ITEMS>
__global__ void __launch_bounds__(BLOCK)
col_medians_sample_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                          int ignore_zero_mode, const uint32_t* __restrict__ flags,
                          double* __restrict__ med, int32_t sp, int32_t cap) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int ignore_zero = resolve_ignore_zero(ignore_zero_mode, flags);
  const int tid = threadIdx.x;
  const int ns = BLOCK * sp;                              // sample slots
  uint64_t* sample = reinterpret_cast<uint64_t*>(smem_raw);
  uint64_t* inside = sample + ns;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(inside + cap);   // [0] valid [1] <lo [2] ==lo [3] inside [4] ==hi
  uint32_t* hist = cnt + 8;                                // 256 + 2: radix-select fallback scratch
  constexpr int NK = ITEMS > 0 ? ITEMS : 1;
  uint64_t key[NK];

  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const double* sc = S + (int64_t)c * lds;
    if (tid < 8) cnt[tid] = 0;
    if constexpr (ITEMS > 0) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const int i = tid + j * BLOCK;
        const double v = sc[i < m ? i : m - 1];              // clamped: unconditional loads, all in flight
        key[j] = (i < m) ? masked_key(v, ignore_zero) : ~0ull;
      }
    }
    __syncthreads();
    // one sweep over the column's keys: FN(key) for every element this thread owns
#define PLAIDHIP_SWEEP(FN)                                                              \
    if constexpr (ITEMS > 0) {                                                          \
      _Pragma("unroll") for (int j = 0; j < ITEMS; ++j) { FN(key[j]) }                  \
    } else {                                                                            \
      for (int i0 = tid; i0 < m; i0 += 8 * BLOCK) {                                     \
        double v_[8];                                                                   \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                 \
          const int i = i0 + u * BLOCK;                                                 \
          v_[u] = sc[i < m ? i : m - 1];                                                \
        }                                                                               \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                 \
          const uint64_t k_ = (i0 + u * BLOCK < m) ? masked_key(v_[u], ignore_zero) : ~0ull; \
          FN(k_)                                                                        \
        }                                                                               \
      }                                                                                 \
    }
    // ---- 1. valid count + strided sample ----------------------------------------------
    uint32_t valid = 0;
#define PLAIDHIP_FN_VALID(k) valid += ((k) != ~0ull);
    PLAIDHIP_SWEEP(PLAIDHIP_FN_VALID)
#undef PLAIDHIP_FN_VALID
    for (int off = 32; off >= 1; off >>= 1) valid += __shfl_xor(valid, off, 64);
    if ((tid & 63) == 0 && valid) atomicAdd(&cnt[0], valid);
    for (int j = tid; j < ns; j += BLOCK) {
      const int64_t i = ((int64_t)j * m) / ns;
      sample[j] = (m >= ns || j < m) ? masked_key(sc[m >= ns ? i : j], ignore_zero) : ~0ull;
    }
    bitonic_sort_lds(sample, (uint32_t)ns);               // starts and ends with a barrier
    const uint32_t cv = cnt[0];
    double r;
    if (cv == 0) {
      r = ignore_zero ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
    } else {
      const uint32_t k_lo = (cv - 1) >> 1, k_hi = cv >> 1;
      // valid samples are the ones below the all-ones key (masked keys sorted last)
      uint32_t sv = lower_bound_lds(sample, (uint32_t)ns, ~0ull);
      uint64_t lo = 0, hi = ~0ull - 1;                     // defaults: bracket = every valid key
      if ((int64_t)cv > cap && sv >= 64) {
        const double q = (double)sv / (double)cv;
        const int32_t delta = (int32_t)(2.0 * sqrt((double)sv)) + 1;
        const int32_t p_lo = (int32_t)(q * k_lo) - delta, p_hi = (int32_t)(q * k_hi) + delta + 1;
        if (p_lo >= 0) lo = sample[p_lo];
        if (p_hi < (int32_t)sv) hi = sample[p_hi];
      }
      // ---- 3. sweep: segment counts + collect the inside keys ---------------------------
      uint32_t below = 0, eqlo = 0, eqhi = 0;
#define PLAIDHIP_FN_SEG(k)                                           \
      if ((k) != ~0ull) {                                            \
        if ((k) < lo) ++below;                                       \
        else if ((k) == lo) ++eqlo;                                  \
        else if ((k) < hi) {                                         \
          const uint32_t pos = atomicAdd(&cnt[3], 1u);               \
          if (pos < (uint32_t)cap) inside[pos] = (k);                \
        } else if ((k) == hi) ++eqhi;                                \
      }
      PLAIDHIP_SWEEP(PLAIDHIP_FN_SEG)
#undef PLAIDHIP_FN_SEG
#undef PLAIDHIP_SWEEP
      for (int off = 32; off >= 1; off >>= 1) {
        below += __shfl_xor(below, off, 64);
        eqlo += __shfl_xor(eqlo, off, 64);
        eqhi += __shfl_xor(eqhi, off, 64);
      }
      if ((tid & 63) == 0) {
        if (below) atomicAdd(&cnt[1], below);
        if (eqlo) atomicAdd(&cnt[2], eqlo);
        if (eqhi) atomicAdd(&cnt[4], eqhi);
      }
      __syncthreads();
      const uint32_t nb = cnt[3], c_below = cnt[1], c_eqlo = cnt[2], c_eqhi = cnt[4];
      const bool fits = nb <= (uint32_t)cap;
      if (fits) bitonic_sort_lds(inside, nb);               // uniform branch (nb is block-wide)
      // ---- 4. read the two middle ranks off the segments ---------------------------------
      auto resolve = [&](uint32_t k, uint64_t& out) -> bool {
        if (!fits || k < c_below) return false;
        uint32_t kk = k - c_below;
        if (kk < c_eqlo) { out = lo; return true; }
        kk -= c_eqlo;
        if (kk < nb) { out = inside[kk]; return true; }
        kk -= nb;
        if (kk < c_eqhi) { out = hi; return true; }
        return false;
      };
      uint64_t v1 = 0, v2 = 0;
      const bool ok1 = resolve(k_lo, v1), ok2 = resolve(k_hi, v2);
#ifdef PLAIDHIP_DIAG
      if ((!ok1 || !ok2) && tid == 0 && flags != nullptr)   // tools/ build only: flags[3] counts bracket misses
        atomicAdd(const_cast<uint32_t*>(&flags[3]), 1u);
#endif
      if (!ok1) v1 = radix_select_global(sc, m, ignore_zero, k_lo, hist, hist + 256);   // rare
      if (!ok2) v2 = (k_hi == k_lo) ? v1 : radix_select_global(sc, m, ignore_zero, k_hi, hist, hist + 256);
      r = (v1 == v2) ? key_to_f64(v1) : 0.5 * (key_to_f64(v1) + key_to_f64(v2));
    }
    if (tid == 0) med[c] = r;
    __syncthreads();
  }
}

// deterministic single-workgroup reductions (n samples: tiny)
__global__ void __launch_bounds__(1024)
sum_kernel(const double* __restrict__ v, int64_t count, double* out) {
  __shared__ double s_sum[1024];
  __shared__ double s_cnt[1024];
  const int tid = threadIdx.x;
  double s = 0.0, c = 0.0;
  for (int64_t i = tid; i < count; i += 1024) {
    const double x = v[i];
    if (x == x) { s += x; c += 1.0; }
  }
  s_sum[tid] = s;
  s_cnt[tid] = c;
  __syncthreads();
  for (int h = 512; h >= 1; h >>= 1) {
    if (tid < h) { s_sum[tid] += s_sum[tid + h]; s_cnt[tid] += s_cnt[tid + h]; }
    __syncthreads();
  }
  if (tid == 0) { out[0] = s_sum[0]; out[1] = s_cnt[0]; }
}

__global__ void __launch_bounds__(1024)
max_kernel(const double* __restrict__ v, int64_t count, double* out) {
  __shared__ double s_max[1024];
  const int tid = threadIdx.x;
  double s = -INFINITY;
  for (int64_t i = tid; i < count; i += 1024) {
    const double x = v[i];
    s = (x > s) ? x : s;
  }
  s_max[tid] = s;
  __syncthreads();
  for (int h = 512; h >= 1; h >>= 1) {
    if (tid < h) s_max[tid] = (s_max[tid + h] > s_max[tid]) ? s_max[tid + h] : s_max[tid];
    __syncthreads();
  }
  if (tid == 0) out[0] = s_max[0];
}

// (x - med[col]) + mean(med): one streaming read + write of S.  16-byte accesses, four of them in flight per thread
// before the first is used, non-temporal both ways (S does not fit any cache and is not read again by this kernel).
__global__ void __launch_bounds__(256)
shift_columns_kernel(double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                     const double* __restrict__ med, double add, const double* __restrict__ red) {
  typedef double f64x2_s __attribute__((ext_vector_type(2)));
  constexpr int UN = 4;   // (measured with 2 / 4 / 8: 4 is the best or within 3 % of it at m = 5,000 and 50,000, aligned or not)
  if (red != nullptr) add = red[0] / red[1];   // mean(medx, na.rm=TRUE) from {sum, count}
  // grid.y walks columns, grid.x * block walks the rows of a column
  for (int c = blockIdx.y; c < n; c += gridDim.y) {
    double* sc = S + (int64_t)c * lds;
    const double md = med[c];
    const int head = (int)((reinterpret_cast<uintptr_t>(sc) >> 3) & 1u);   // first element not 16-byte aligned
    const int npairs = (m - head) >> 1;
    if (blockIdx.x == 0) {
      if (threadIdx.x == 0 && head) sc[0] = (sc[0] - md) + add;
      if (threadIdx.x == 1 && ((m - head) & 1)) sc[m - 1] = (sc[m - 1] - md) + add;
    }
    f64x2_s* p = reinterpret_cast<f64x2_s*>(sc + head);
    for (int base = blockIdx.x * 256 * UN; base < npairs; base += gridDim.x * 256 * UN) {
      f64x2_s v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = base + u * 256 + (int)threadIdx.x;
        v[u] = __builtin_nontemporal_load(p + (i < npairs ? i : npairs - 1));
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = base + u * 256 + (int)threadIdx.x;
        if (i < npairs) {
          f64x2_s r;
          r.x = (v[u].x - md) + add;
          r.y = (v[u].y - md) + add;
          __builtin_nontemporal_store(r, p + i);
        }
      }
    }
  }
}

// (x - med[col]) + mean(med) written as FLOAT into another matrix: the last step of normalize_medians (R/plaid.R:572) fused
// with the cast a sample-sharded job makes anyway before its scores travel to the root -- config 5's 1e6 x 50,000 result is
// 400 GB in fp64, more than one GPU holds, so the gather carries fp32 (sharded.gather_scores(dtype = float32)).  S itself
// stays as the crossprod wrote it: one read of S and a half-size write replace the read + write of the shift and the read +
// half-size write of the cast.  16-byte loads (two scores), 8-byte stores, four loads in flight per thread; the rounding
// is that of a plain fp64 -> fp32 conversion of the shifted value -- bit-identical to shift_columns followed by a cast.
__global__ void __launch_bounds__(256)
shift_columns_cast_f32_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n, const double* __restrict__ med,
                              double add, const double* __restrict__ red, float* __restrict__ out, int64_t ldo) {
  typedef double f64x2_s __attribute__((ext_vector_type(2)));
  typedef float f32x2_s __attribute__((ext_vector_type(2)));
  constexpr int UN = 4;
  if (red != nullptr) add = red[0] / red[1];
  for (int c = blockIdx.y; c < n; c += gridDim.y) {
    const double* sc = S + (int64_t)c * lds;
    float* oc = out + (int64_t)c * ldo;
    const double md = med[c];
    // pairs are taken where BOTH the fp64 source (16 bytes) and the fp32 destination (8 bytes) are aligned; else by element
    const bool pairs_ok = ((reinterpret_cast<uintptr_t>(sc) & 15u) == 0u) && ((reinterpret_cast<uintptr_t>(oc) & 7u) == 0u);
    if (!pairs_ok) {
      for (int i = blockIdx.x * 256 + (int)threadIdx.x; i < m; i += gridDim.x * 256)
        oc[i] = (float)((sc[i] - md) + add);
      continue;
    }
    const int npairs = m >> 1;
    if (blockIdx.x == 0 && threadIdx.x == 0 && (m & 1)) oc[m - 1] = (float)((sc[m - 1] - md) + add);
    const f64x2_s* p = reinterpret_cast<const f64x2_s*>(sc);
    f32x2_s* q = reinterpret_cast<f32x2_s*>(oc);
    for (int base = blockIdx.x * 256 * UN; base < npairs; base += gridDim.x * 256 * UN) {
      f64x2_s v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = base + u * 256 + (int)threadIdx.x;
        v[u] = __builtin_nontemporal_load(p + (i < npairs ? i : npairs - 1));
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = base + u * 256 + (int)threadIdx.x;
        if (i < npairs) {
          f32x2_s r;
          r.x = (float)((v[u].x - md) + add);
          r.y = (float)((v[u].y - md) + add);
          __builtin_nontemporal_store(r, q + i);
        }
      }
    }
  }
}

// ---- element-wise / column helpers for the rank-transform callers -------------------------
// (replaid.ucell R/plaid.R:276-282, replaid.aucell :304-309, replaid.scse :155-190)
__global__ void __launch_bounds__(256)
map_kernel(double* __restrict__ v, int64_t count, int op, double p0, const double* __restrict__ scalar) {
  if (op >= 4) {
    // replaid.scse's automatic removeLog2 (R/plaid.R:160-161), decided on the device from {min, max} at `scalar`
    // (p0 != 0: a sparse X whose implicit zeros take part): the transform runs iff min == 0 and max < 20
    double mn = scalar[0], mx = scalar[1];
    if (p0 != 0.0) { mn = mn < 0.0 ? mn : 0.0; mx = mx > 0.0 ? mx : 0.0; }
    if (!(mn == 0.0 && mx < 20.0)) return;
    op -= 2;
    scalar = nullptr;
  }
  const double sc = scalar != nullptr ? *scalar : 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    double x = v[i];
    if (op == 0) x = fmin(sc - x, p0);                                 // pmin(max(rX) - rX, rmax + 1)
    else if (op == 1) x = 1.08 * fmax((x - (sc - p0)) / p0, 0.0);      // 1.08 * pmax((rX - (max - K)) / K, 0)
    else if (op == 2) x = (x > 0.0) ? exp2(x) : x;                     // X[X > 0] <- 2 ** X[X > 0]
    else x = exp2(x);                                                  // X@x <- 2 ** X@x
    v[i] = x;
  }
}

// out[c] = sum |X[, c]| over a dense column (len rows) or the stored values of a CSC column
__global__ void __launch_bounds__(256)
col_abs_sums_kernel(const double* __restrict__ X, int64_t ldx, int32_t len, const int32_t* __restrict__ Xp,
                    int32_t n, double* __restrict__ out) {
  __shared__ double s_part[4];
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const double* xc = Xp ? X + Xp[c] : X + (int64_t)c * ldx;
    const int cnt = Xp ? Xp[c + 1] - Xp[c] : len;
    double s = 0.0;
    for (int i = threadIdx.x; i < cnt; i += 256) s += fabs(xc[i]);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    __syncthreads();
  }
}

// S[j, c] = S[j, c] * mul / (col_div ? col_div[c] * div_scale + 1e-8 : 1) + (row_add ? row_add[j] : 0) + add
__global__ void __launch_bounds__(256)
affine_kernel(double* __restrict__ S, int64_t lds, int32_t m, int32_t n, double mul,
              const double* __restrict__ col_div, double div_scale, const double* __restrict__ row_add, double add) {
  for (int c = blockIdx.y; c < n; c += gridDim.y) {
    double* sc = S + (int64_t)c * lds;
    const double f = col_div ? mul / (col_div[c] * div_scale + 1e-8) : mul;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
      sc[i] = sc[i] * f + (row_add ? row_add[i] : 0.0) + add;
  }
}

// min / max over non-NaN values, two stages (deterministic): every workgroup of stage one reduces a slice of v to
// {min, max} (part[b], part[nb + b]); one workgroup folds the partials into out[0] = min, out[1] = max
__device__ __forceinline__ void block_minmax_1024(double mn, double mx, double* s_mn, double* s_mx, double& omn, double& omx) {
  const int tid = threadIdx.x;
  s_mn[tid] = mn; s_mx[tid] = mx;
  __syncthreads();
  for (int h = 512; h >= 1; h >>= 1) {
    if (tid < h) {
      s_mn[tid] = s_mn[tid + h] < s_mn[tid] ? s_mn[tid + h] : s_mn[tid];
      s_mx[tid] = s_mx[tid + h] > s_mx[tid] ? s_mx[tid + h] : s_mx[tid];
    }
    __syncthreads();
  }
  omn = s_mn[0]; omx = s_mx[0];
}

__global__ void __launch_bounds__(1024)
minmax_partial_kernel(const double* __restrict__ v, int64_t count, double* __restrict__ part) {
  __shared__ double s_mn[1024], s_mx[1024];
  double mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 1024) {
    const double x = v[i];
    if (x == x) { mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
  }
  double omn, omx;
  block_minmax_1024(mn, mx, s_mn, s_mx, omn, omx);
  if (threadIdx.x == 0) { part[blockIdx.x] = omn; part[gridDim.x + blockIdx.x] = omx; }
}

__global__ void __launch_bounds__(1024)
minmax_final_kernel(const double* __restrict__ part, int nb, double* out) {
  __shared__ double s_mn[1024], s_mx[1024];
  double mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += 1024) {
    const double a = part[i], b = part[nb + i];
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  double omn, omx;
  block_minmax_1024(mn, mx, s_mn, s_mx, omn, omx);
  if (threadIdx.x == 0) { out[0] = omn; out[1] = omx; }
}

// {0 if every stored value is finite and >= 0, else -1; max; smallest value > 0 (+inf if there is none)} over the stored
// values Xx[Xp[0] .. Xp[n]) of a dgCMatrix -- the range is read from Xp ON THE DEVICE, so a caller that does not know
// nnz(X) (or knows it only approximately: a shard of a larger matrix) cannot make the sweep read too much or too little.
// It decides whether the scatter crossprod may sum in fixed point (kernels_spmm.hip: scatter_fixed_ok).
__global__ void __launch_bounds__(1024)
nonneg_range_partial_kernel(const double* __restrict__ v, const int32_t* __restrict__ Xp, int32_t n,
                            double* __restrict__ part) {
  __shared__ double s_mn[1024], s_mx[1024];
  const int64_t begin = Xp[0], end = Xp[n];
  double ok = 0.0, mx = 0.0, mnz = INFINITY;
  for (int64_t i = begin + (int64_t)blockIdx.x * 1024 + threadIdx.x; i < end; i += (int64_t)gridDim.x * 1024) {
    const double x = v[i];
    ok = ((x >= 0.0) && (x < INFINITY)) ? ok : -1.0;      // false for NaN, negatives and +inf
    mx = x > mx ? x : mx;
    mnz = (x > 0.0 && x < mnz) ? x : mnz;
  }
  double omn, omx;
  block_minmax_1024(ok, mx, s_mn, s_mx, omn, omx);
  if (threadIdx.x == 0) { part[blockIdx.x] = omn; part[gridDim.x + blockIdx.x] = omx; }
  __syncthreads();
  block_minmax_1024(mnz, 0.0, s_mn, s_mx, omn, omx);
  if (threadIdx.x == 0) part[2 * gridDim.x + blockIdx.x] = omn;
}

__global__ void __launch_bounds__(1024)
nonneg_range_final_kernel(const double* __restrict__ part, int nb, double* out) {
  __shared__ double s_mn[1024], s_mx[1024];
  double ok = 0.0, mx = 0.0, mnz = INFINITY;
  for (int i = threadIdx.x; i < nb; i += 1024) {
    const double a = part[i], b = part[nb + i], c = part[2 * nb + i];
    ok = a < ok ? a : ok;
    mx = b > mx ? b : mx;
    mnz = c < mnz ? c : mnz;
  }
  double omn, omx;
  block_minmax_1024(ok, mx, s_mn, s_mx, omn, omx);
  if (threadIdx.x == 0) { out[0] = omn; out[1] = omx; }
  __syncthreads();
  block_minmax_1024(mnz, 0.0, s_mn, s_mx, omn, omx);
  if (threadIdx.x == 0) { out[2] = omn; out[3] = INFINITY; }   // out[3]: the largest column sum, when launch_colsum_max follows
}

// the largest sum of a column's stored values (all >= 0 where it matters): no score can exceed it, whatever the size of its set
__global__ void __launch_bounds__(256)
colsum_kernel(const double* __restrict__ v, const int32_t* __restrict__ Xp, int32_t n, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += nw) {
    const int64_t q0 = Xp[c], q1 = Xp[c + 1];
    double s = 0.0;
    for (int64_t i = q0 + lane; i < q1; i += 64) s += v[i];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[c] = s;
  }
}

// nnz_hint only sizes the grid (< 0: unknown); the swept range is Xx[Xp[0] .. Xp[n]) whatever it says
int launch_nonneg_range(plaidhip_ctx* ctx, const double* Xx, const int32_t* Xp, int32_t n, int64_t nnz_hint, double* out) {
  const int cap = ctx->num_cu * 2;
  int64_t nb64 = nnz_hint < 0 ? cap : (nnz_hint + 8 * 1024 - 1) / (8 * 1024);
  const int nb = nb64 < 1 ? 1 : (nb64 > cap ? cap : (int)nb64);
  int rc = ensure_workspace(ctx, (size_t)nb * 3 * sizeof(double));
  if (rc != PLAIDHIP_OK) return rc;
  double* part = reinterpret_cast<double*>(ctx->ws);
  hipLaunchKernelGGL(nonneg_range_partial_kernel, dim3(nb), dim3(1024), 0, ctx->stream, Xx, Xp, n, part);
  hipLaunchKernelGGL(nonneg_range_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, part, nb, out);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

// out[0] = max over the columns of the sum of their stored values (per-column sums in the workspace, then one reduction)
int launch_colsum_max(plaidhip_ctx* ctx, const double* Xx, const int32_t* Xp, int32_t n, double* out) {
  if (n <= 0) return PLAIDHIP_OK;
  int rc = ensure_workspace(ctx, (size_t)n * sizeof(double));
  if (rc != PLAIDHIP_OK) return rc;
  double* sums = reinterpret_cast<double*>(ctx->ws);
  const int need = (n + 3) / 4, cap = ctx->num_cu * 16;
  hipLaunchKernelGGL(colsum_kernel, dim3(need < cap ? need : cap), dim3(256), 0, ctx->stream, Xx, Xp, n, sums);
  return launch_max(ctx, sums, n, out);
}

int launch_map(plaidhip_ctx* ctx, double* v, int64_t count, int op, double p0, const double* scalar) {
  if (count == 0) return PLAIDHIP_OK;
  int64_t blocks = (count + 256 * 8 - 1) / (256 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(map_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, v, count, op, p0, scalar);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_col_abs_sums(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t len, const int32_t* Xp,
                        int32_t n, double* out) {
  if (n == 0) return PLAIDHIP_OK;
  hipLaunchKernelGGL(col_abs_sums_kernel, dim3(n < 8192 ? n : 8192), dim3(256), 0, ctx->stream, X, ldx, len, Xp, n, out);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_affine(plaidhip_ctx* ctx, double* S, int64_t lds, int32_t m, int32_t n, double mul,
                  const double* col_div, double div_scale, const double* row_add, double add) {
  ctx->fmed.valid = false;
  if (n == 0 || m == 0) return PLAIDHIP_OK;
  int bx = (m + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(affine_kernel, dim3(bx, n < 32768 ? n : 32768), dim3(256), 0, ctx->stream, S, lds, m, n, mul,
                     col_div, div_scale, row_add, add);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_minmax(plaidhip_ctx* ctx, const double* v, int64_t count, double* out) {
  int64_t nb64 = (count + 8 * 1024 - 1) / (8 * 1024);
  const int cap = ctx->num_cu * 2;
  const int nb = nb64 < 1 ? 1 : (nb64 > cap ? cap : (int)nb64);
  int rc = ensure_workspace(ctx, (size_t)nb * 2 * sizeof(double));
  if (rc != PLAIDHIP_OK) return rc;
  double* part = reinterpret_cast<double*>(ctx->ws);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(1024), 0, ctx->stream, v, count, part);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, part, nb, out);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}


// m <= BLOCK*ITEMS: register-resident RADIX selection, 8 bits per pass.  The column is read once
// (coalesced); every thread keeps ITEMS 64-bit keys.  Keys are binned on (key - lo) >> shift over
// the current range [lo, lo + range] (first the column's [min, max], then the bin that holds the
// wanted rank): 256 bins, LDS-atomic histogram, one wavefront scans the bins; the search stops as
// soon as the bin holds a single key or is one key wide.  Doubles of similar magnitude need two
// to three passes where the bitwise search above needs one pass per differing bit.  (A variant on
// 32-bit words -- high words first, then the low words of the keys sharing the selected high word --
// needs more passes and measured slower: 0.235 vs 0.20 ms on C2; the passes are latency-, not
// issue-bound.)
template <int BLOCK, int ITEMS>
__global__ void __launch_bounds__(BLOCK)
col_medians_radix_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                         int ignore_zero_mode, const uint32_t* __restrict__ flags,
                         double* __restrict__ med) {
  constexpr int NW = BLOCK / 64;
  __shared__ __align__(16) uint32_t s_hist[256];
  __shared__ unsigned long long s_mn[NW], s_mx[NW];
  __shared__ uint32_t s_cnt[NW];
  __shared__ uint32_t s_pick[3];        // digit, keys of the range below the bin, keys in the bin
  __shared__ unsigned long long s_key;
  const int ignore_zero = resolve_ignore_zero(ignore_zero_mode, flags);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += BLOCK) s_hist[i] = 0;
  __syncthreads();

  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const double* sc = S + (int64_t)c * lds;
    uint64_t key[ITEMS];
    uint64_t tmn = ~0ull, tmx = 0ull;
    uint32_t vc = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int i = tid + j * BLOCK;
      const uint64_t k = (i < m) ? masked_key(sc[i < m ? i : m - 1], ignore_zero) : ~0ull;
      key[j] = k;
      const bool valid = k != ~0ull;
      vc += (uint32_t)__popcll(__ballot(valid));
      tmn = (valid && k < tmn) ? k : tmn;
      tmx = (valid && k > tmx) ? k : tmx;
    }
    for (int off = 32; off >= 1; off >>= 1) {
      const uint64_t a = (uint64_t)__shfl_xor((unsigned long long)tmn, off, 64);
      const uint64_t b = (uint64_t)__shfl_xor((unsigned long long)tmx, off, 64);
      tmn = a < tmn ? a : tmn;
      tmx = b > tmx ? b : tmx;
    }
    if (lane == 0) { s_mn[wave] = tmn; s_mx[wave] = tmx; s_cnt[wave] = vc; }
    __syncthreads();
    uint64_t kmin = ~0ull, kmax = 0ull;
    uint32_t cnt = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      kmin = s_mn[w] < kmin ? s_mn[w] : kmin;
      kmax = s_mx[w] > kmax ? s_mx[w] : kmax;
      cnt += s_cnt[w];
    }
    cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
    double r;
    if (cnt == 0) {
      r = ignore_zero ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
    } else {
      const uint32_t k_lo = (cnt - 1) >> 1, k_hi = cnt >> 1;
      uint64_t lo = kmin, range = kmax - kmin;
      uint32_t k = k_lo;         // rank wanted inside [lo, lo + range]
      uint32_t count = cnt;      // keys inside [lo, lo + range]
      while (range != 0ull && count > 1u) {
        const int bits = 64 - __clzll((long long)range);
        const int shift = bits > 8 ? bits - 8 : 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const uint64_t d = key[j] - lo;
          if (key[j] >= lo && d <= range) atomicAdd(&s_hist[(uint32_t)(d >> shift)], 1u);   // masked keys lie above kmax
        }
        __syncthreads();
        if (wave == 0) {
          // 256-bin scan by one wavefront: lane l owns bins 4l .. 4l+3
          const uint4 h4 = *reinterpret_cast<const uint4*>(&s_hist[lane * 4]);
          *reinterpret_cast<uint4*>(&s_hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
          const uint32_t mine = h4.x + h4.y + h4.z + h4.w;
          uint32_t incl = mine;
          for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
          }
          uint32_t excl = incl - mine;
          if (mine != 0 && excl <= k && k < incl) {
            uint32_t d = 0, hh = h4.x;
            if (k >= excl + h4.x) { excl += h4.x; d = 1; hh = h4.y;
              if (k >= excl + h4.y) { excl += h4.y; d = 2; hh = h4.z;
                if (k >= excl + h4.z) { excl += h4.z; d = 3; hh = h4.w; } } }
            s_pick[0] = (uint32_t)lane * 4u + d;
            s_pick[1] = excl;
            s_pick[2] = hh;
          }
        }
        __syncthreads();
        const uint32_t dsel = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pick[0]);
        const uint32_t below = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pick[1]);
        count = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pick[2]);
        k -= below;
        lo += (uint64_t)dsel << shift;
        range = shift ? ((1ull << shift) - 1ull) : 0ull;
      }
      uint64_t V = lo;                       // range == 0: `count` copies of lo
      if (range != 0ull) {                   // a single key inside a wider bin: fetch it
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const uint64_t d = key[j] - lo;
          if (key[j] >= lo && d <= range) s_key = key[j];
        }
        __syncthreads();
        V = s_key;
      }
      const uint32_t c_le = (k_lo - k) + count;   // keys <= V
      uint64_t V2 = V;
      if (k_hi != k_lo && k_hi >= c_le) {
        // even count and the upper middle is the next distinct key: smallest key above V
        uint64_t mn = ~0ull;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) mn = (key[j] > V && key[j] < mn) ? key[j] : mn;
        for (int off = 32; off >= 1; off >>= 1) {
          const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)mn, off, 64);
          mn = o < mn ? o : mn;
        }
        __syncthreads();            // every wave is done reading s_mn of the min/max step
        if (lane == 0) s_mn[wave] = mn;
        __syncthreads();
        V2 = ~0ull;
#pragma unroll
        for (int w = 0; w < NW; ++w) V2 = s_mn[w] < V2 ? s_mn[w] : V2;
      }
      r = (V2 == V) ? key_to_f64(V) : 0.5 * (key_to_f64(V) + key_to_f64(V2));
    }
    if (tid == 0) med[c] = r;
    __syncthreads();   // s_mn / s_mx / s_cnt / s_key are rewritten by the next column
  }
}


// Any m: ONE WAVEFRONT per column, no workgroup barriers.  The column is swept three or four
// times (the first sweep from HBM, the others from L2): min/max of the keys; a 256-bin histogram
// over the current key range (repeated on the bin that holds the wanted rank while that bin has
// more than CAP keys); a collect sweep that compacts the keys of the bin into LDS, where the
// wavefront sorts them (bitonic) and reads the middle key(s) off.  Every wavefront works on its
// own column with its own 1 KiB histogram and CAP-key list, so a CU keeps 16+ columns in flight
// and nothing waits for another wavefront; the register-resident kernels above spend most of
// their time in workgroup barriers once m grows.
__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wavefront are executed in order; only the compiler must not reorder
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef double f64x2_t __attribute__((ext_vector_type(2)));

// Wave-level scan and reductions on the DPP network (row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then row_bcast 15 and
// 31 across them: the gfx9 sequence) instead of __shfl_up / __shfl_xor, which hipcc lowers to ds_bpermute_b32 -- a round
// trip through the LDS crossbar per step, six to twelve of them in a dependent chain per scan or 64-bit reduction, in
// kernels whose wavefronts have nothing else to issue meanwhile.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
#define PH_DPP_STEPS(STEP) STEP(0x111, 0xf, 0xf) STEP(0x112, 0xf, 0xf) STEP(0x114, 0xf, 0xe) STEP(0x118, 0xf, 0xc) \
                           STEP(0x142, 0xa, 0xf) STEP(0x143, 0xc, 0xf)
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
#define PH_STEP(C, R, B) v += dpp_u32<C, R, B>(0u, v);
  PH_DPP_STEPS(PH_STEP)
#undef PH_STEP
  return v;
}
// reductions: the result of all 64 lanes, wave-uniform (read from lane 63)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define PH_STEP(C, R, B) { const uint32_t t = dpp_u32<C, R, B>(0xffffffffu, v); v = t < v ? t : v; }
  PH_DPP_STEPS(PH_STEP)
#undef PH_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define PH_STEP(C, R, B) { const uint32_t t = dpp_u32<C, R, B>(0u, v); v = t > v ? t : v; }
  PH_DPP_STEPS(PH_STEP)
#undef PH_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#define PH_STEP(C, R, B)                                                                                  \
  {                                                                                                        \
    const uint64_t t = ((uint64_t)dpp_u32<C, R, B>(0xffffffffu, (uint32_t)(v >> 32)) << 32) |             \
                       dpp_u32<C, R, B>(0xffffffffu, (uint32_t)v);                                         \
    v = t < v ? t : v;                                                                                     \
  }
  PH_DPP_STEPS(PH_STEP)
#undef PH_STEP
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#define PH_STEP(C, R, B)                                                                                  \
  {                                                                                                        \
    const uint64_t t = ((uint64_t)dpp_u32<C, R, B>(0u, (uint32_t)(v >> 32)) << 32) | dpp_u32<C, R, B>(0u, (uint32_t)v); \
    v = t > v ? t : v;                                                                                     \
  }
  PH_DPP_STEPS(PH_STEP)
#undef PH_STEP
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
}
#undef PH_DPP_STEPS

// m <= 64 * ITEMS: the same radix selection with ONE WAVEFRONT per column and the keys in its registers (up to 96 per
// lane): no workgroup barrier anywhere -- a pass is ITEMS LDS atomics per lane into the wavefront's own 256-bin histogram,
// a scan of the bins by the same wavefront, and wave-uniform results come back through readlane instead of LDS.  Two
// wavefronts per SIMD (eight columns in flight per CU, 40 KB each at C2) keep the memory system busy while the others
// select; the workgroup-per-column kernel above holds four columns per CU and spends its time in the two barriers of a pass.
// The kernel is bound by its vector instructions (~35 per key), so the common steps work on the HIGH dword of the keys:
// the first range is [min high dword << 32, max high dword << 32 | ~0] -- wider than [min, max] but covering it -- and
// while a pass shifts by >= 32 bits (the first one or two do) bin and range test are 32-bit operations; the exact 64-bit
// form takes over below that.
template <int ITEMS, int WG_PER_CU>
__global__ void __launch_bounds__(256, WG_PER_CU)
col_medians_wave_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                        int ignore_zero_mode, const uint32_t* __restrict__ flags, double* __restrict__ med,
                        unsigned long long* __restrict__ dbg) {
#ifdef PLAIDHIP_DIAG
#define PH_MSTAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[k] += t_ - tl; tl = t_; }
  unsigned long long st[5] = {0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime(), npass = 0;
#else
#define PH_MSTAMP(k)
#endif
  // per wavefront 256 bins + 64 private trash bins (one per lane) that keys outside the current range count into: the
  // atomic is unconditional, so no per-key lane mask has to live in scalar registers across the unrolled loop
  __shared__ __align__(16) uint32_t s_hist[4][320];
  const int ignore_zero = resolve_ignore_zero(ignore_zero_mode, flags);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* hist = s_hist[wave];
  *reinterpret_cast<uint4*>(&hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
  hist[256 + lane] = 0u;
  wave_lds_sync();
  const uint32_t trash = 256u + (uint32_t)lane;
  const int nwaves = gridDim.x * 4;
  for (int c = blockIdx.x * 4 + wave; c < n; c += nwaves) {
    const double* sc = S + (int64_t)c * lds;
    uint64_t key[ITEMS];
    double raw[ITEMS];
    int lane_o = lane;                       // opaque per column: the offsets are recomputed, not kept in ITEMS registers
    asm volatile("" : "+v"(lane_o));
    // The first FULL = ITEMS - 16 rows of 64 lie inside every column this instantiation is launched for (m > 64 FULL):
    // plain loads, no mask.  Only the last 16 rows can reach past the column's end: clamped address, masked below.
    constexpr int FULL = ITEMS - 16;
    const double* __restrict__ scl = sc + lane_o;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      if (j < FULL) {
        raw[j] = __builtin_nontemporal_load(scl + j * 64);
      } else {
        const int i = lane_o + j * 64;
        raw[j] = __builtin_nontemporal_load(sc + (i < m ? i : m - 1));
      }
    }
    // (the bound is re-read through an opaque copy: the lane masks of the address clamps above must not be kept in
    //  scalar registers until the values arrive)
    int m_use = m;
    asm volatile("" : "+s"(m_use));
    // a valid key never has an all-ones high dword (that would be a NaN): masked <=> high dword == ~0
    uint32_t hmn = ~0u, hmx = 0u, cnt = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      // masked_key() in 32-bit steps, all-ones for NaN, ignored zeros and the rows behind the column's end
      const double v = raw[j];
      const double c0 = v + 0.0;                                   // -0 -> +0
      const uint32_t h0 = (uint32_t)__double2hiint(c0), l0 = (uint32_t)__double2loint(c0);
      const uint32_t sgn = (uint32_t)((int32_t)h0 >> 31);
      bool masked = (v != v) | ((ignore_zero != 0) & (v == 0.0));
      if (j >= FULL) masked = masked | (lane_o + j * 64 >= m_use);
      const uint32_t h = masked ? ~0u : (h0 ^ (sgn | 0x80000000u));
      uint32_t l = masked ? ~0u : (l0 ^ sgn);
      asm volatile("" : "+v"(l));   // computed HERE: hipcc otherwise sinks it to its first use and keeps sign, mask and raw dword per key until then
      key[j] = ((uint64_t)h << 32) | l;
      cnt += (uint32_t)__popcll(__ballot(h != ~0u));
      hmn = h < hmn ? h : hmn;
      const uint32_t hx = masked ? 0u : h;
      hmx = hx > hmx ? hx : hmx;
      if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (keeps the unrolled loops from running ahead: registers)
    }
    hmn = wave_min_u32(hmn);
    hmx = wave_max_u32(hmx);
    PH_MSTAMP(0)   // loads + keys + min/max
    double r;
    if (cnt == 0) {
      r = ignore_zero ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
    } else {
      const uint32_t k_lo = (cnt - 1) >> 1, k_hi = cnt >> 1;
      uint64_t lo = (uint64_t)hmn << 32, range = ((uint64_t)(hmx - hmn) << 32) | 0xffffffffull;
      if (hmx == hmn) {
        // every valid key shares its high dword (constant or nearly constant column): exact [min, max] of the low dwords
        uint32_t lmn = ~0u, lmx = 0u;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const bool valid = (uint32_t)(key[j] >> 32) != ~0u;
          const uint32_t l = (uint32_t)key[j];
          lmn = (valid && l < lmn) ? l : lmn;
          lmx = (valid && l > lmx) ? l : lmx;
        }
        lmn = wave_min_u32(lmn);
        lmx = wave_max_u32(lmx);
        lo |= (uint64_t)lmn;
        range = (uint64_t)(lmx - lmn);
      }
      uint32_t k = k_lo;         // rank wanted inside [lo, lo + range]
      uint32_t count = cnt;      // keys inside [lo, lo + range]
      while (range != 0ull && count > 1u) {
        const int bits = 64 - __clzll((long long)range);
        const int shift = bits > 8 ? bits - 8 : 0;
        if (shift >= 32) {
          // lo has a zero low dword and range an all-ones one (true of the first range and kept by every pass that
          // shifts by >= 32): bin and range test from the high dwords alone
          const uint32_t lo_h = (uint32_t)(lo >> 32), range_h = (uint32_t)(range >> 32);
          const int sh = shift - 32;
#pragma unroll
          for (int j = 0; j < ITEMS; ++j) {
            const uint32_t dh = (uint32_t)(key[j] >> 32) - lo_h;      // below lo wraps above every range; masked keys lie above
            const uint32_t bin = (dh <= range_h) ? (dh >> sh) : trash;
            atomicAdd(&hist[bin], 1u);
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < ITEMS; ++j) {
            const uint64_t d = key[j] - lo;
            const uint32_t bin = (d <= range) ? (uint32_t)(d >> shift) : trash;
            atomicAdd(&hist[bin], 1u);
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
          }
        }
        wave_lds_sync();
        // lane l owns bins 4l .. 4l+3
        const uint4 h4 = *reinterpret_cast<const uint4*>(&hist[lane * 4]);
        *reinterpret_cast<uint4*>(&hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t mine = h4.x + h4.y + h4.z + h4.w;
        const uint32_t incl = wave_scan_add_u32(mine);
        uint32_t excl = incl - mine;
        const bool here = mine != 0 && excl <= k && k < incl;
        uint32_t d = 0, hh = h4.x;
        if (here) {
          if (k >= excl + h4.x) { excl += h4.x; d = 1; hh = h4.y;
            if (k >= excl + h4.y) { excl += h4.y; d = 2; hh = h4.z;
              if (k >= excl + h4.z) { excl += h4.z; d = 3; hh = h4.w; } } }
        }
        const int src = __builtin_ctzll(__ballot(here));     // exactly one lane holds the wanted rank
        const uint32_t dsel = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)lane * 4u + d), src);
        const uint32_t below = (uint32_t)__builtin_amdgcn_readlane((int)excl, src);
        count = (uint32_t)__builtin_amdgcn_readlane((int)hh, src);
        wave_lds_sync();
        k -= below;
        lo += (uint64_t)dsel << shift;
        range = shift ? ((1ull << shift) - 1ull) : 0ull;
#ifdef PLAIDHIP_DIAG
        ++npass;
#endif
      }
      PH_MSTAMP(1)   // histogram passes
      uint64_t V = lo;                       // range == 0: `count` copies of lo
      if (range != 0ull) {                   // a single key inside a wider bin: fetch it
        uint64_t f = ~0ull;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const uint64_t d = key[j] - lo;
          f = (d <= range) ? key[j] : f;
          if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        V = wave_min_u64(f);
      }
      const uint32_t c_le = (k_lo - k) + count;   // keys <= V
      uint64_t V2 = V;
      if (k_hi != k_lo && k_hi >= c_le) {
        // even count and the upper middle is the next distinct key: smallest key above V
        uint64_t mn = ~0ull;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
          const uint64_t t = key[j] - V - 1ull;        // key <= V wraps to the top
          mn = t < mn ? t : mn;
          if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        V2 = wave_min_u64(mn) + V + 1ull;
      }
      r = (V2 == V) ? key_to_f64(V) : 0.5 * (key_to_f64(V) + key_to_f64(V2));
    }
    if (lane == 0) med[c] = r;
    PH_MSTAMP(2)   // single-key fetch + upper middle
  }
#ifdef PLAIDHIP_DIAG
  if (dbg != nullptr && lane == 0) {
    unsigned long long* d = dbg + (size_t)(blockIdx.x * 4 + wave) * 4;
    d[0] = st[0]; d[1] = st[1]; d[2] = st[2]; d[3] = npass;
  }
#endif
#undef PH_MSTAMP
}

// key of one value as two dwords (same order as masked_key), masked entries -> {~0, ~0}
struct Key32 { uint32_t hi, lo; };
__device__ __forceinline__ Key32 masked_key32(double v, int ignore_zero) {
  const double c = v + 0.0;                                    // -0 -> +0
  const uint32_t h = (uint32_t)__double2hiint(c), l = (uint32_t)__double2loint(c);
  const uint32_t sgn = (uint32_t)((int32_t)h >> 31);           // 0 / ~0
  Key32 k{h ^ (sgn | 0x80000000u), l ^ sgn};
  const bool masked = (v != v) | ((ignore_zero != 0) & (v == 0.0));
  if (masked) { k.hi = 0xffffffffu; k.lo = 0xffffffffu; }
  return k;
}

// One wavefront visits every key of a column: f(key) is called by ALL lanes together (masked or
// out-of-range entries carry the all-ones key), 16-byte loads, 8 KiB per wavefront in flight.
template <typename F>
__device__ __forceinline__ void sweep_column(const double* __restrict__ sc, int32_t m, int ignore_zero, int lane, F&& f) {
  constexpr int UN = 8;
  const int head = (int)((reinterpret_cast<uintptr_t>(sc) >> 3) & 1u);   // first element not 16-byte aligned
  const int npairs = (m - head) >> 1;
  const int tail = (m - head) & 1;
  {
    Key32 k{0xffffffffu, 0xffffffffu};
    if (lane == 0 && head) k = masked_key32(sc[0], ignore_zero);
    if (lane == 1 && tail) k = masked_key32(sc[m - 1], ignore_zero);
    f(k);
  }
  const f64x2_t* __restrict__ p = reinterpret_cast<const f64x2_t*>(sc + head);
  for (int base = 0; base < npairs; base += 64 * UN) {
    f64x2_t v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = base + u * 64 + lane;
      v[u] = p[i < npairs ? i : npairs - 1];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool ok = base + u * 64 + lane < npairs;
      Key32 a = masked_key32(v[u].x, ignore_zero), b = masked_key32(v[u].y, ignore_zero);
      if (!ok) { a.hi = a.lo = b.hi = b.lo = 0xffffffffu; }
      f(a);
      f(b);
    }
  }
}

// The same walk handing out the raw doubles: f(value, exists) is called by ALL lanes together.
template <typename F>
__device__ __forceinline__ void sweep_column_f64(const double* __restrict__ sc, int32_t m, int lane, F&& f) {
  constexpr int UN = 8;
  const int head = (int)((reinterpret_cast<uintptr_t>(sc) >> 3) & 1u);   // first element not 16-byte aligned
  const int npairs = (m - head) >> 1;
  const int tail = (m - head) & 1;
  {
    double v = 0.0;
    bool ok = false;
    if (lane == 0 && head) { v = sc[0]; ok = true; }
    if (lane == 1 && tail) { v = sc[m - 1]; ok = true; }
    f(v, ok);
  }
  const f64x2_t* __restrict__ p = reinterpret_cast<const f64x2_t*>(sc + head);
  for (int base = 0; base < npairs; base += 64 * UN) {
    f64x2_t v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = base + u * 64 + lane;
      v[u] = p[i < npairs ? i : npairs - 1];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool ok = base + u * 64 + lane < npairs;
      f(v[u].x, ok);
      f(v[u].y, ok);
    }
  }
}

// The same walk, software-pipelined: the 8 KiB of batch k + 1 are requested before batch k is processed (two register
// buffers), so a wavefront always has a batch in flight while it classifies the previous one.  The plain walk issues a
// batch, waits for all of it, processes it, and only then asks for the next: with the 16 wavefronts per CU this kernel's
// LDS lists allow, the column sweep ran at 3.5 TB/s with the vector units 60 % idle.  f must issue the same memory
// operations for every batch (no wave-uniform branch around a store): hipcc's wait counters then stay exact and the
// wait before batch k is "all but the 8 loads of batch k + 1", not "everything".  after_batch() runs behind every batch
// (wave-uniform work: flushing a staging list).
template <typename F, typename G>
__device__ __forceinline__ void sweep_column_f64_pipelined(const double* __restrict__ sc, int32_t m, int lane, F&& f, G&& after_batch) {
  constexpr int UN = 8;
  const int head = (int)((reinterpret_cast<uintptr_t>(sc) >> 3) & 1u);   // first element not 16-byte aligned
  const int npairs = (m - head) >> 1;
  const int tail = (m - head) & 1;
  {
    double v = 0.0;
    bool ok = false;
    if (lane == 0 && head) { v = sc[0]; ok = true; }
    if (lane == 1 && tail) { v = sc[m - 1]; ok = true; }
    f(v, ok);
    after_batch();
  }
  const f64x2_t* __restrict__ p = reinterpret_cast<const f64x2_t*>(sc + head);
  f64x2_t va[UN], vb[UN];
#define PH_SWEEP_LOAD(buf, b0)                                          \
  _Pragma("unroll") for (int u = 0; u < UN; ++u) {                       \
    const int i = (b0) + u * 64 + lane;                                  \
    buf[u] = __builtin_nontemporal_load(p + (i < npairs ? i : (npairs > 0 ? npairs - 1 : 0))); \
  }
#define PH_SWEEP_USE(buf, b0)                                            \
  _Pragma("unroll") for (int u = 0; u < UN; ++u) {                       \
    const bool ok = (b0) + u * 64 + lane < npairs;                       \
    f(buf[u].x, ok);                                                     \
    f(buf[u].y, ok);                                                     \
  }
  if (npairs > 0) {
    PH_SWEEP_LOAD(va, 0)
    for (int base = 0; base < npairs; base += 2 * 64 * UN) {
      PH_SWEEP_LOAD(vb, base + 64 * UN)
      PH_SWEEP_USE(va, base)
      after_batch();
      PH_SWEEP_LOAD(va, base + 2 * 64 * UN)
      PH_SWEEP_USE(vb, base + 64 * UN)
      after_batch();
    }
  }
#undef PH_SWEEP_LOAD
#undef PH_SWEEP_USE
}

// k-th smallest (0-based) of the wavefront's register-resident keys (ITEMS per lane; all-ones = no key) by radix selection
// over [kmin, kmax] with the wavefront's own histogram (256 bins + a trash bin per lane, all zero on entry and on return):
// the selection loop of col_medians_wave_kernel in its plain 64-bit form.  `count` = number of keys.
template <int ITEMS>
__device__ __forceinline__ uint64_t wave_radix_select(const uint64_t (&key)[ITEMS], uint32_t k, uint32_t count, uint64_t kmin,
                                                      uint64_t kmax, uint32_t* hist, int lane) {
  const uint32_t trash = 256u + (uint32_t)lane;
  uint64_t lo = kmin, range = kmax - kmin;
  while (range != 0ull && count > 1u) {
    const int bits = 64 - __clzll((long long)range);
    const int shift = bits > 8 ? bits - 8 : 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const uint64_t d = key[j] - lo;          // below lo wraps above every range in use; all-ones lies above kmax
      atomicAdd(&hist[(d <= range) ? (uint32_t)(d >> shift) : trash], 1u);
    }
    wave_lds_sync();
    const uint4 h4 = *reinterpret_cast<const uint4*>(&hist[lane * 4]);
    *reinterpret_cast<uint4*>(&hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t mine = h4.x + h4.y + h4.z + h4.w;
    const uint32_t incl = wave_scan_add_u32(mine);
    uint32_t excl = incl - mine;
    const bool here = mine != 0 && excl <= k && k < incl;
    uint32_t d = 0, hh = h4.x;
    if (here) {
      if (k >= excl + h4.x) { excl += h4.x; d = 1; hh = h4.y;
        if (k >= excl + h4.y) { excl += h4.y; d = 2; hh = h4.z;
          if (k >= excl + h4.z) { excl += h4.z; d = 3; hh = h4.w; } } }
    }
    const int src = __builtin_ctzll(__ballot(here));
    const uint32_t dsel = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)lane * 4u + d), src);
    const uint32_t below = (uint32_t)__builtin_amdgcn_readlane((int)excl, src);
    count = (uint32_t)__builtin_amdgcn_readlane((int)hh, src);
    wave_lds_sync();
    k -= below;
    lo += (uint64_t)dsel << shift;
    range = shift ? ((1ull << shift) - 1ull) : 0ull;
  }
  if (range == 0ull) return lo;              // `count` copies of lo
  uint64_t f = ~0ull;                        // a single key inside a wider bin: fetch it
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const uint64_t d = key[j] - lo;
    f = (d <= range) ? key[j] : f;
  }
  return wave_min_u64(f);
}

// The search interval is [lo, lo + 2^B - 1]; a key K lies inside iff K - lo does not borrow and
// (K - lo) >> B == 0.  Its bin is (K - lo) >> shift, shift = max(B - 8, 0).  Everything per key is
// 32-bit arithmetic (64-bit integer compares and shifts run at a quarter of that rate).
struct RangeTest {
  uint32_t lohi, lolo;
  int shift;        // bin = d >> shift
  uint32_t nbins;   // 1 << (B - shift) <= 256
  // returns the bin, or 0xffffffff when the key is outside the interval
  __device__ __forceinline__ uint32_t bin(const Key32& k) const {
    const uint32_t dlo = k.lo - lolo;
    const uint32_t borrow = k.lo < lolo ? 1u : 0u;
    const uint32_t dhi = k.hi - lohi - borrow;
    const bool under = (k.hi < lohi) | ((k.hi == lohi) & (borrow != 0u));
    uint32_t b;
    bool hi_ok = true;
    if (shift >= 32) {
      b = dhi >> (shift - 32);
    } else {
      b = shift ? __builtin_amdgcn_alignbit(dhi, dlo, (uint32_t)shift) : dlo;
      hi_ok = (dhi >> shift) == 0u;   // shift == 0: dhi must be 0
    }
    const bool valid = k.hi != 0xffffffffu;   // masked entries (no valid key has an all-ones high word)
    return (valid && !under && hi_ok && b < nbins) ? b : 0xffffffffu;
  }
};

template <int CAP, int kSampleChunks>   // kSampleChunks x 64 sample values, spread over the column
__global__ void __launch_bounds__(256, 4)   // four workgroups per CU is what the LDS lists allow: 128 registers
col_medians_stream_kernel(const double* __restrict__ S, int64_t lds, int32_t m, int32_t n,
                          int ignore_zero_mode, const uint32_t* __restrict__ flags,
                          double* __restrict__ med, unsigned long long* __restrict__ cand_all, int32_t ccap,
                          unsigned long long* __restrict__ dbg, const int32_t* __restrict__ status = nullptr) {
#ifdef PLAIDHIP_DIAG
#define PH_SSTAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[k] += t_ - tl; tl = t_; }
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#else
#define PH_SSTAMP(k)
#endif
  // (+64: a private trash bin / trash slot per lane, so that the classification sweep below is free of branches)
  __shared__ __align__(16) uint32_t s_hist[4][256 + 64];
  __shared__ unsigned long long s_list[4][CAP + 64];
  // (measured on 8,192 columns x 50k: 16 chunks no faster than 8; 3 sigma 20 % SLOWER -- a miss costs three sweeps)
  constexpr float kSampleSigmas = 4.0f;
  const int ignore_zero = resolve_ignore_zero(ignore_zero_mode, flags);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* hist = s_hist[wave];
  unsigned long long* list = s_list[wave];
  *reinterpret_cast<uint4*>(&hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
  wave_lds_sync();

  const int nwaves = gridDim.x * 4;
  // this wavefront's candidate list in global memory (see the sampled start below)
  unsigned long long* cand = cand_all != nullptr ? cand_all + (size_t)(blockIdx.x * 4 + wave) * (size_t)ccap : nullptr;
  for (int c = blockIdx.x * 4 + wave; c < n; c += nwaves) {
    if (status != nullptr && status[c] != 0) continue;   // (wave-uniform) its median came out of the crossprod launch
    const double* sc = S + (int64_t)c * lds;
    uint32_t cnt = 0, k_lo = 0, k_hi = 0, k = 0, count = 0;
    uint32_t ncand = 0;          // keys of the sample interval written to `cand` by the sampled start (0: none / overflow)
    uint64_t hi_cap = ~0ull;     // sampled start: its counts cover the keys <= qb only (the histogram interval is padded
                                 // to a power of two and may reach beyond qb) -- every later sweep applies the same cap
    uint64_t lo = 0;
    int B = 0;
    bool seeded = false;
    // ---- sampled start (large columns): a 512-entry sample (8 chunks spread over the column) gives a key
    //      interval around the middle rank, 4 sigma of a sample quantile either side; ONE sweep then counts the
    //      valid keys, the keys below the interval and a 256-bin histogram inside it, which replaces the min/max
    //      sweep and the first two histogram sweeps of the generic path.  If the interval misses the middle
    //      rank (probability ~1e-4 per column for exchangeable data) the generic path starts from scratch.
    if (m > 4 * CAP) {
      // the sample stays in registers (8 keys per lane) and the two bracket keys are SELECTED (two short radix
      // selections on the wavefront's histogram) instead of read off a sorted list: sorting 512 keys in LDS was 45
      // compare-exchange stages with four round trips each, 13 % of the kernel's time (in-kernel stamps)
      uint64_t skey[kSampleChunks];
      double sraw[kSampleChunks];
#pragma unroll
      for (int u = 0; u < kSampleChunks; ++u) {
        int64_t i = (int64_t)u * m / kSampleChunks + lane;
        sraw[u] = sc[i < m ? i : m - 1];
      }
      uint32_t ns = 0;
      uint64_t smn = ~0ull, smx = 0ull;
#pragma unroll
      for (int u = 0; u < kSampleChunks; ++u) {
        const uint64_t kk = masked_key(sraw[u], ignore_zero);
        skey[u] = kk;
        const bool valid = kk != ~0ull;
        ns += (uint32_t)__popcll(__ballot(valid));
        smn = kk < smn ? kk : smn;
        smx = (valid && kk > smx) ? kk : smx;
      }
      smn = wave_min_u64(smn);
      smx = wave_max_u64(smx);
      const uint32_t mid = ns > 0 ? (ns - 1u) >> 1 : 0u;
      const uint32_t w = (uint32_t)(kSampleSigmas * 0.5f * sqrtf((float)ns)) + 2u;   // kSampleSigmas sigma of a sample quantile's rank
      if (ns >= 256u && mid > w && mid + 1u + w < ns - 1u) {
        const uint64_t qa = wave_radix_select<kSampleChunks>(skey, mid - w, ns, smn, smx, hist, lane);
        const uint64_t qb = wave_radix_select<kSampleChunks>(skey, mid + 1u + w, ns, smn, smx, hist, lane);
        const int Bw = qb == qa ? 1 : 64 - __clzll((long long)(qb - qa));   // [qa, qa + 2^Bw - 1] covers [qa, qb]
        RangeTest rt;
        rt.lohi = (uint32_t)(qa >> 32);
        rt.lolo = (uint32_t)qa;
        rt.shift = Bw > 8 ? Bw - 8 : 0;
        rt.nbins = 1u << (Bw - rt.shift);
        // ONE sweep: classification on the doubles themselves (three compares per key; counters in scalar registers
        // through ballots), and only the keys inside [qa, qb] -- a sixth of the column -- are turned into keys, binned
        // and appended to this wavefront's candidate list in global memory; the selection below then reads the list
        // instead of sweeping the column a second time (1.36 instead of 2 passes over a column that fits no cache).
        const double qa_d = key_to_f64(qa), qb_d = key_to_f64(qb);
        uint32_t below = 0;
        PH_SSTAMP(0)   // sample: strided loads + sort
        // Branch-free per value: every lane counts into the histogram (lanes outside the interval into their private trash
        // bin) and writes its key into the wavefront's LDS list (outside the interval, or past the list's end: into its
        // trash slot), so the 16 values of a batch are one basic block the scheduler can interleave.  Behind each batch
        // the staged keys -- about a sixth of the batch -- go to the candidate list in global memory with full-wave
        // stores.  A batch that stages more than CAP keys (an interval far too wide: heavy ties) gives the list up; the
        // collect sweep then reads the column, as it does without a list.
        uint32_t nstage = 0;
        bool list_ok = cand != nullptr;
        const uint32_t trash_bin = 256u + (uint32_t)lane, trash_slot = (uint32_t)CAP + (uint32_t)lane;
        auto classify = [&](auto iz_c) {
          return [&](double v, bool ok) {
          // (ordered compares are false for a NaN: `lt` and `in` need no validity test of their own)
          // The wave-level masks are built from ballots of SINGLE compares combined with scalar ANDs: the ballot of a
          // combined predicate is materialised by hipcc as v_cndmask + v_cmp per ballot (6 of the 31 vector instructions
          // per value); the lane's own `in` below is the same combination as a predicate (scalar ANDs of the same masks).
          const bool nz = decltype(iz_c)::value ? (v != 0.0) : true;
          const bool in = ok && nz && !(v < qa_d) && (v <= qb_d);
          const unsigned long long ltm_ = __ballot(v < qa_d);
          unsigned long long live = __ballot(ok) & __ballot(v == v);
          if (decltype(iz_c)::value) live &= __ballot(v != 0.0);
          const unsigned long long bal = live & __ballot(v <= qb_d) & ~ltm_;
          cnt += (uint32_t)__popcll(live);
          below += (uint32_t)__popcll(live & ltm_);
          const double c0 = v + 0.0;                                   // -0 -> +0
          const uint32_t h = (uint32_t)__double2hiint(c0), l = (uint32_t)__double2loint(c0);
          const uint32_t sgn = (uint32_t)((int32_t)h >> 31);
          const uint64_t key = ((uint64_t)(h ^ (sgn | 0x80000000u)) << 32) | (uint64_t)(l ^ sgn);
          const uint32_t b = (uint32_t)((key - qa) >> rt.shift);
          atomicAdd(&hist[in ? b : trash_bin], 1u);
          const uint32_t pos = nstage + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          list[(in && pos < (uint32_t)CAP) ? pos : trash_slot] = key;
          nstage += (uint32_t)__popcll(bal);
        };
        };
        auto flush = [&]() {
          nstage = (uint32_t)__builtin_amdgcn_readfirstlane((int)nstage);
          if (nstage > (uint32_t)CAP) list_ok = false;
          if (list_ok && nstage != 0u) {
            wave_lds_sync();
            for (uint32_t i = (uint32_t)lane; i < nstage; i += 64u)
              if (ncand + i < (uint32_t)ccap) cand[ncand + i] = list[i];
            wave_lds_sync();
          }
          ncand += nstage;
          nstage = 0;
        };
        // (the ignore.zero test is compiled in or out: it is the same for every column of the call)
        if (ignore_zero != 0) sweep_column_f64_pipelined(sc, m, lane, classify(std::true_type{}), flush);
        else sweep_column_f64_pipelined(sc, m, lane, classify(std::false_type{}), flush);
        if (!list_ok) ncand = 0xffffffffu;
        PH_SSTAMP(1)   // classification sweep
        cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
        below = (uint32_t)__builtin_amdgcn_readfirstlane((int)below);
        ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);
        if (cand == nullptr || ncand > (uint32_t)ccap) ncand = 0;   // no list (or given up, or overflown): the collect sweep reads the column
        wave_lds_sync();
        const uint4 h4 = *reinterpret_cast<const uint4*>(&hist[lane * 4]);
        *reinterpret_cast<uint4*>(&hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_sync();
        const uint32_t mine = h4.x + h4.y + h4.z + h4.w;
        const uint32_t incl = wave_scan_add_u32(mine);
        const uint32_t inside = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (cnt > 0u) {
          k_lo = (cnt - 1u) >> 1;
          k_hi = cnt >> 1;
          if (below <= k_lo && k_lo - below < inside) {
            k = k_lo - below;
            uint32_t excl = incl - mine;
            const bool owner = mine != 0 && excl <= k && k < incl;
            uint32_t d = 0, hh = h4.x;
            if (k >= excl + h4.x) { excl += h4.x; d = 1; hh = h4.y;
              if (k >= excl + h4.y) { excl += h4.y; d = 2; hh = h4.z;
                if (k >= excl + h4.z) { excl += h4.z; d = 3; hh = h4.w; } } }
            const int src = (int)__builtin_ctzll(__ballot(owner));
            const uint32_t dsel = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)lane * 4u + d), src);
            k -= (uint32_t)__builtin_amdgcn_readlane((int)excl, src);
            count = (uint32_t)__builtin_amdgcn_readlane((int)hh, src);
            lo = qa + ((uint64_t)dsel << rt.shift);
            B = rt.shift;
            seeded = true;
            hi_cap = qb;
          }
        }
        if (!seeded) ncand = 0;
      } else {
        wave_lds_sync();
      }
    }
    PH_SSTAMP(2)   // scan of the seeded histogram
    if (!seeded) {
    // ---- generic start, sweep 0: range of the keys' high words and the number of unmasked entries ---------
    uint32_t hmin = 0xffffffffu, hmax = 0u;
    cnt = 0;
    sweep_column(sc, m, ignore_zero, lane, [&](const Key32& k_) {
      const bool valid = k_.hi != 0xffffffffu;      // no valid key has an all-ones high word
      cnt += valid ? 1u : 0u;
      hmin = k_.hi < hmin ? k_.hi : hmin;            // (a masked key never lowers the minimum)
      hmax = (valid && k_.hi > hmax) ? k_.hi : hmax;
    });
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t a_ = __shfl_xor(hmin, off, 64), b_ = __shfl_xor(hmax, off, 64);
      hmin = a_ < hmin ? a_ : hmin;
      hmax = b_ > hmax ? b_ : hmax;
      cnt += __shfl_xor(cnt, off, 64);
    }
    cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
    hmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)hmin);
    hmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)hmax);
    if (cnt != 0u) {
      k_lo = (cnt - 1) >> 1;
      k_hi = cnt >> 1;
      lo = (uint64_t)hmin << 32;
      B = 32 + (hmax == hmin ? 0 : 32 - __clz((int)(hmax - hmin)));   // interval [lo, lo + 2^B - 1] holds every key
      k = k_lo;       // rank wanted inside the interval
      count = cnt;    // keys inside the interval
    }
    }
    double r;
    if (cnt == 0) {
      r = ignore_zero ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
    } else {
      // ---- histogram sweeps until the interval fits the list ------------------------------
      while (B != 0 && count > (uint32_t)CAP) {
        RangeTest rt;
        rt.lohi = (uint32_t)(lo >> 32);
        rt.lolo = (uint32_t)lo;
        rt.shift = B > 8 ? B - 8 : 0;
        rt.nbins = 1u << (B - rt.shift);
        sweep_column(sc, m, ignore_zero, lane, [&](const Key32& key) {
          const uint32_t b = rt.bin(key);
          if (b != 0xffffffffu && ((((uint64_t)key.hi << 32) | key.lo) <= hi_cap)) atomicAdd(&hist[b], 1u);
        });
        wave_lds_sync();
        const uint4 h4 = *reinterpret_cast<const uint4*>(&hist[lane * 4]);
        *reinterpret_cast<uint4*>(&hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_sync();
        const uint32_t mine = h4.x + h4.y + h4.z + h4.w;
        const uint32_t incl = wave_scan_add_u32(mine);
        uint32_t excl = incl - mine;
        const bool owner = mine != 0 && excl <= k && k < incl;
        uint32_t d = 0, hh = h4.x;
        if (k >= excl + h4.x) { excl += h4.x; d = 1; hh = h4.y;
          if (k >= excl + h4.y) { excl += h4.y; d = 2; hh = h4.z;
            if (k >= excl + h4.z) { excl += h4.z; d = 3; hh = h4.w; } } }
        const int src = (int)__builtin_ctzll(__ballot(owner));   // exactly one lane owns the wanted rank
        const uint32_t dsel = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)lane * 4u + d), src);
        const uint32_t below = (uint32_t)__builtin_amdgcn_readlane((int)excl, src);
        count = (uint32_t)__builtin_amdgcn_readlane((int)hh, src);
        k -= below;
        lo += (uint64_t)dsel << rt.shift;
        B = rt.shift;
      }
      PH_SSTAMP(3)   // generic start / further histogram sweeps (none after a seeded start that fits the list)
      uint64_t V = lo, V2 = lo;
      bool need_above = false;
      if (B != 0) {
        // ---- collect sweep: keys of the interval -> LDS ------------------------------------
        RangeTest rt;
        rt.lohi = (uint32_t)(lo >> 32);
        rt.lolo = (uint32_t)lo;
        rt.shift = B > 8 ? B - 8 : 0;
        rt.nbins = 1u << (B - rt.shift);
        uint32_t base = 0;
        auto collect = [&](const Key32& key) {
          const bool in = rt.bin(key) != 0xffffffffu && ((((uint64_t)key.hi << 32) | key.lo) <= hi_cap);
          const unsigned long long bal = __ballot(in);
          if (in) {
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (pos < (uint32_t)CAP) list[pos] = ((unsigned long long)key.hi << 32) | key.lo;
          }
          base += (uint32_t)__popcll(bal);
        };
        if (ncand != 0u) {
          // the interval lies inside the sample interval: its keys are among the candidates (written by this very
          // wavefront a moment ago: same-wave stores and loads are ordered)
          // (16-byte loads, 8 KiB per batch, the next batch requested before the current one is used: read 256 keys at a
          //  time with a round trip each, this loop was most of the 18 % of the kernel spent behind the sweep)
          typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
          const u64x2_t* __restrict__ cp = reinterpret_cast<const u64x2_t*>(cand);
          const uint32_t npair = (ncand + 1u) >> 1;   // (an odd count reads one slot past the last key: inside the list, masked below)
          constexpr int CB = 8;
          u64x2_t ka[CB], kb[CB];
#define PH_CAND_LOAD(buf, b0)                                            \
  _Pragma("unroll") for (int u = 0; u < CB; ++u) {                        \
    const uint32_t i = (b0) + (uint32_t)u * 64u + (uint32_t)lane;         \
    buf[u] = cp[i < npair ? i : npair - 1u];                              \
  }
#define PH_CAND_USE(buf, b0)                                              \
  _Pragma("unroll") for (int u = 0; u < CB; ++u) {                        \
    const uint32_t i = 2u * ((b0) + (uint32_t)u * 64u + (uint32_t)lane);  \
    Key32 k0{(uint32_t)(buf[u].x >> 32), (uint32_t)buf[u].x}, k1{(uint32_t)(buf[u].y >> 32), (uint32_t)buf[u].y}; \
    if (i >= ncand) { k0.hi = 0xffffffffu; k0.lo = 0xffffffffu; }         \
    if (i + 1u >= ncand) { k1.hi = 0xffffffffu; k1.lo = 0xffffffffu; }    \
    collect(k0);                                                          \
    collect(k1);                                                          \
  }
          PH_CAND_LOAD(ka, 0u)
          for (uint32_t i0 = 0; i0 < npair; i0 += 2u * 64u * CB) {
            PH_CAND_LOAD(kb, i0 + 64u * CB)
            PH_CAND_USE(ka, i0)
            PH_CAND_LOAD(ka, i0 + 2u * 64u * CB)
            PH_CAND_USE(kb, i0 + 64u * CB)
          }
#undef PH_CAND_LOAD
#undef PH_CAND_USE
        } else {
          sweep_column(sc, m, ignore_zero, lane, collect);
        }
        // ---- sort the list (count <= CAP keys, padded with all-ones to a power of two) ------
        uint32_t N = 2;
        while (N < count) N <<= 1;
        for (uint32_t i = count + lane; i < N; i += 64) list[i] = ~0ull;
        wave_lds_sync();
        for (uint32_t kk = 2; kk <= N; kk <<= 1)
          for (uint32_t j = kk >> 1; j >= 1; j >>= 1) {
            for (uint32_t t = lane; t < (N >> 1); t += 64) {
              const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // lower index of the pair
              const uint32_t p = i | j;
              const bool up = (i & kk) == 0;
              const unsigned long long x = list[i], y = list[p];
              if ((x > y) == up) { list[i] = y; list[p] = x; }
            }
            wave_lds_sync();
          }
        V = list[k];
        V2 = V;
        if (k_hi != k_lo) {
          if (k + 1 < count) V2 = list[k + 1];
          else need_above = true;
        }
        wave_lds_sync();   // the list is refilled by the next column
      } else {
        // `count` copies of the key lo; the upper middle is another copy or the next larger key
        const uint32_t c_le = (k_lo - k) + count;
        need_above = k_hi != k_lo && k_hi >= c_le;
      }
      PH_SSTAMP(4)   // collect (candidates or column) + sort
      if (need_above) {   // rare: the upper middle key is the smallest key above V (one more sweep)
        uint64_t above = ~0ull;
        sweep_column(sc, m, ignore_zero, lane, [&](const Key32& key) {
          const uint64_t kk = ((uint64_t)key.hi << 32) | key.lo;
          above = (kk > V && key.hi != 0xffffffffu && kk < above) ? kk : above;
        });
        for (int off = 32; off >= 1; off >>= 1) {
          const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)above, off, 64);
          above = o < above ? o : above;
        }
        V2 = above;
      }
      r = (V2 == V) ? key_to_f64(V) : 0.5 * (key_to_f64(V) + key_to_f64(V2));
    }
    if (lane == 0) med[c] = r;
    PH_SSTAMP(5)   // upper-middle sweep (rare)
  }
#ifdef PLAIDHIP_DIAG
  if (dbg != nullptr && lane == 0) {
    unsigned long long* d = dbg + (size_t)(blockIdx.x * 4 + wave) * 8;
    for (int q = 0; q < 6; ++q) d[q] = st[q];
  }
#endif
#undef PH_SSTAMP
}

int launch_minflags(plaidhip_ctx* ctx, const double* S, int64_t count, uint32_t* flags) {
  if (count == 0) return PLAIDHIP_OK;
  int64_t blocks = (count + 256 * 8 - 1) / (256 * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(minflags_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, S, count, flags);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

template <int BLOCK, int ITEMS>
static void launch_bits(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n,
                        int ignore_zero, const uint32_t* flags, double* med) {
  // one workgroup per column; cap the grid and let workgroups walk columns
  const int cap = ctx->num_cu * (2048 / BLOCK) * 4;
  const int grid = n < cap ? n : cap;
  hipLaunchKernelGGL((col_medians_bits_kernel<BLOCK, ITEMS>), dim3(grid), dim3(BLOCK), 0, ctx->stream, S, lds,
                     m, n, ignore_zero, flags, med);
}

#ifdef PLAIDHIP_DIAG
static unsigned long long* g_med_dbg = nullptr;   // tools/ build: per-phase stamps of the wave-per-column kernel
void debug_set_median_stamps(void* dbg) { g_med_dbg = static_cast<unsigned long long*>(dbg); }
static unsigned long long* median_stamps() { return g_med_dbg; }
#else
static unsigned long long* median_stamps() { return nullptr; }
#endif

template <int ITEMS, int WG_PER_CU>
static void launch_wave(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n,
                        int ignore_zero, const uint32_t* flags, double* med) {
  static_assert(ITEMS >= 16 && ITEMS % 16 == 0, "classes of 1,024 values");
  const int cap = ctx->num_cu * WG_PER_CU * 4;            // WG_PER_CU workgroups of four wavefronts per CU, several rounds
  const int need = (n + 3) / 4;
  const int grid = need < cap ? need : cap;
  hipLaunchKernelGGL((col_medians_wave_kernel<ITEMS, WG_PER_CU>), dim3(grid), dim3(256), 0, ctx->stream, S, lds, m, n, ignore_zero,
                     flags, med, median_stamps());
}

template <int BLOCK, int ITEMS>
static void launch_radix(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n,
                         int ignore_zero, const uint32_t* flags, double* med) {
  const int cap = ctx->num_cu * (2048 / BLOCK) * 4;
  const int grid = n < cap ? n : cap;
  hipLaunchKernelGGL((col_medians_radix_kernel<BLOCK, ITEMS>), dim3(grid), dim3(BLOCK), 0, ctx->stream, S, lds,
                     m, n, ignore_zero, flags, med);
}

// ---- medians selected while the sparse crossprod writes the scores (spmm_scatter_csc_f64<.., MED>, kernels_spmm.hip) ----
// 1. the mean score of every column before the crossprod: alpha * sum_i x[i, c] u[i] + beta * kappa (geneset.cpp: u, kappa)
__global__ void __launch_bounds__(256)
colmean_predict_kernel(const int32_t* __restrict__ Xp, const int32_t* __restrict__ Xi, const double* __restrict__ Xx, int32_t n,
                       const double* __restrict__ u, double alpha, const double* __restrict__ alpha_div, double beta_kappa,
                       double* __restrict__ pred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double al = alpha_div != nullptr ? alpha / *alpha_div : alpha;
  for (int c = blockIdx.x * 4 + wave; c < n; c += gridDim.x * 4) {
    // (four independent chains: the loop is a load, a dependent gather and an add -- one round trip per 64 values when
    // rolled, 0.54 ms for the 1e8 stored values of config 3)
    const int q1 = Xp[c + 1];
    int q = Xp[c] + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (; q + 192 < q1; q += 256) {
      const int i0 = Xi[q], i1 = Xi[q + 64], i2 = Xi[q + 128], i3 = Xi[q + 192];
      const double x0 = Xx[q], x1 = Xx[q + 64], x2 = Xx[q + 128], x3 = Xx[q + 192];
      s0 += x0 * u[i0];
      s1 += x1 * u[i1];
      s2 += x2 * u[i2];
      s3 += x3 * u[i3];
    }
    for (; q < q1; q += 64) s0 += Xx[q] * u[Xi[q]];
    double s = (s0 + s1) + (s2 + s3);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) pred[c] = al * s + beta_kappa;
  }
}

// 2. calibration on the first K <= 256 columns (crossprod + standalone medians of those ran before), ROBUST against odd
//    columns among them (empty cells, outliers, NaN): offset = the MEDIAN of (column median - predicted mean), half width =
//    2.3 x the 90th percentile of the absolute deviations from it (normal deviates: 1.645 sigma -> a bracket of ~3.8 sigma
//    either side, 1-5 % of a column's scores); the ignore-zero rule of the sample is what the bracket is calibrated for.
//    Columns outside the bracket are simply left to the standalone kernel.
__global__ void __launch_bounds__(256)
median_calibrate_kernel(const double* __restrict__ medK, const double* __restrict__ pred, int32_t K,
                        const uint32_t* __restrict__ flagsK, double* __restrict__ cal) {
  __shared__ double s_v[256];
  __shared__ double s_pick;
  __shared__ int s_cnt;
  const int t = threadIdx.x;
  double d = INFINITY;
  if (t < K) {
    const double x = medK[t] - pred[t];
    if (x == x && fabs(x) < INFINITY) d = x;
  }
  if (t == 0) s_cnt = 0;
  __syncthreads();
  if (d < INFINITY) atomicAdd(&s_cnt, 1);
  auto select = [&](double mine, int k) {   // the k-th smallest (0-based) of the 256 values, ties by thread index
    s_v[t] = mine;
    __syncthreads();
    int r = 0;
    for (int j = 0; j < 256; ++j) r += (s_v[j] < mine || (s_v[j] == mine && j < t)) ? 1 : 0;
    if (r == k) s_pick = mine;
    __syncthreads();
    const double out = s_pick;
    __syncthreads();
    return out;
  };
  __syncthreads();
  const int n_ok = s_cnt;
  if (n_ok < 16) {   // too few usable columns: an empty bracket (every column goes to the standalone kernel)
    if (t == 0) { cal[0] = 0.0; cal[1] = -1.0; cal[2] = 0.0; }
    return;
  }
  const double off = select(d, (n_ok - 1) / 2);
  const double dev = d < INFINITY ? fabs(d - off) : INFINITY;
  const double q90 = select(dev, (int)(0.9 * (n_ok - 1)));
  if (t == 0) {
    cal[0] = off;
    cal[1] = 2.3 * q90 + 4.0 * fabs(off) * 0x1p-52;
    cal[2] = (flagsK[1] != 0u && flagsK[0] == 0u) ? 1.0 : 0.0;
  }
}

// 3. one wavefront per column: the counts of its (chunk, wavefront) slices say whether both middle order statistics lie
//    among the candidates; if so they are selected from them (<= 64 per lane, in registers: wave_radix_select) -- the same
//    two values the standalone kernels select, averaged the same way.  status[c] = 1: med[c] is final; 0: unresolved
//    (bracket missed, a slice overflowed, empty column, or the matrix as a whole follows the other ignore.zero rule than
//    the calibration sample did).
constexpr int kFmedItems = 64;   // candidates per lane: 4,096 per column
template <int ITEMS>
__device__ __forceinline__ void fmed_select_from(const unsigned long long* __restrict__ cand, int c, int32_t nslice, int32_t capc,
                                                 const uint32_t* s_off, uint32_t total, int64_t k1, int64_t k2, uint32_t below,
                                                 uint32_t* s_hist, int lane, double* __restrict__ med, int32_t* __restrict__ status) {
  // gather the candidates: flat index f -> slice by binary search in the offsets
  uint64_t key[ITEMS];
  uint64_t kmin = ~0ull, kmax = 0ull;
#pragma unroll
  for (int t = 0; t < ITEMS; ++t) {
    const uint32_t f = (uint32_t)t * 64u + (uint32_t)lane;
    uint64_t kk = ~0ull;
    if (f < total) {
      int lo = 0, hi = nslice;              // largest s with s_off[s] <= f
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= f) lo = mid; else hi = mid; }
      const double v = __longlong_as_double((long long)cand[((int64_t)c * nslice + lo) * capc + (f - s_off[lo])]);
      kk = masked_key(v, 0);
      kmin = kk < kmin ? kk : kmin;
      kmax = kk > kmax ? kk : kmax;
    }
    key[t] = kk;
  }
  kmin = wave_min_u64(kmin);
  kmax = wave_max_u64(kmax);
  const uint64_t a1 = wave_radix_select<ITEMS>(key, (uint32_t)(k1 - below), total, kmin, kmax, s_hist, lane);
  const uint64_t a2 = (k2 == k1) ? a1 : wave_radix_select<ITEMS>(key, (uint32_t)(k2 - below), total, kmin, kmax, s_hist, lane);
  if (lane == 0) {
    med[c] = (a1 == a2) ? key_to_f64(a1) : 0.5 * (key_to_f64(a1) + key_to_f64(a2));
    status[c] = 1;
  }
}

__global__ void __launch_bounds__(64)
median_select_kernel(const unsigned long long* __restrict__ cand, const uint4* __restrict__ cnt, int32_t n, int32_t nslice,
                     int32_t capc, int32_t m, const double* __restrict__ cal, int ignore_zero_mode,
                     const uint32_t* __restrict__ flags, double* __restrict__ med, int32_t* __restrict__ status) {
  __shared__ __align__(16) uint32_t s_hist[256 + 64];
  __shared__ uint32_t s_off[257];
  const int lane = threadIdx.x;
  const int iz_true = resolve_ignore_zero(ignore_zero_mode, flags);
  const bool mode_ok = (cal[2] != 0.0) == (iz_true != 0) && cal[1] >= 0.0;
  *reinterpret_cast<uint4*>(&s_hist[lane * 4]) = make_uint4(0u, 0u, 0u, 0u);
  s_hist[256 + lane] = 0u;
  wave_lds_sync();
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    // counts of the column's slices (nslice <= 256: chunks x wavefronts)
    uint32_t below = 0, zero = 0, nan = 0, total = 0;
    bool over = false;
    for (int s0 = 0; s0 < nslice; s0 += 64) {
      const int sidx = s0 + lane;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (sidx < nslice) v = cnt[(int64_t)c * nslice + sidx];
      over |= v.w > (uint32_t)capc;
      const uint32_t incl = wave_scan_add_u32(v.w);
      if (sidx < nslice) s_off[sidx] = total + incl - v.w;
      total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      uint32_t b = v.x, z = v.y, q = v.z;
      for (int off = 32; off >= 1; off >>= 1) { b += __shfl_xor(b, off, 64); z += __shfl_xor(z, off, 64); q += __shfl_xor(q, off, 64); }
      below += b; zero += z; nan += q;
    }
    if (lane == 0) s_off[nslice] = total;
    wave_lds_sync();
    const bool any_over = __ballot(over) != 0ull;
    // (the crossprod launch only notes WHETHER a wavefront wrote NaN scores -- they are skipped, na.rm -- not how many: such
    // a column is left to the standalone kernel)
    const int64_t nv = (int64_t)m - (iz_true ? zero : 0);
    const int64_t k1 = (nv - 1) >> 1, k2 = nv >> 1;
    const bool ok = mode_ok && !any_over && nan == 0 && nv > 0 && total <= (uint32_t)(kFmedItems * 64) && (int64_t)below <= k1 &&
                    k2 < (int64_t)below + total;
    if (!ok) {
      if (lane == 0) status[c] = 0;
      wave_lds_sync();
      continue;
    }
    // (ITEMS candidates per lane, by how many there are: the gather and the selection passes cost in proportion, and a
    // bracket of 1-5 % of 50,000 scores holds 500 ... 2,500 candidates, not the 4,096 the lists could hold)
    if (total <= 8u * 64u) fmed_select_from<8>(cand, c, nslice, capc, s_off, total, k1, k2, below, s_hist, lane, med, status);
    else if (total <= 16u * 64u) fmed_select_from<16>(cand, c, nslice, capc, s_off, total, k1, k2, below, s_hist, lane, med, status);
    else if (total <= 32u * 64u) fmed_select_from<32>(cand, c, nslice, capc, s_off, total, k1, k2, below, s_hist, lane, med, status);
    else fmed_select_from<kFmedItems>(cand, c, nslice, capc, s_off, total, k1, k2, below, s_hist, lane, med, status);
    wave_lds_sync();
  }
}

int launch_colmean_predict(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t n, const double* u,
                           double alpha, const double* alpha_div, double beta_kappa, double* pred) {
  if (n == 0) return PLAIDHIP_OK;
  const int cap = ctx->num_cu * 8;
  const int need = (n + 3) / 4;
  hipLaunchKernelGGL(colmean_predict_kernel, dim3(need < cap ? need : cap), dim3(256), 0, ctx->stream, Xp, Xi, Xx, n, u, alpha,
                     alpha_div, beta_kappa, pred);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_median_calibrate(plaidhip_ctx* ctx, const double* medK, const double* pred, int32_t K, const uint32_t* flagsK,
                            double* cal) {
  hipLaunchKernelGGL(median_calibrate_kernel, dim3(1), dim3(256), 0, ctx->stream, medK, pred, K, flagsK, cal);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_median_select(plaidhip_ctx* ctx, const unsigned long long* cand, const uint32_t* cnt, int32_t n, int32_t nslice,
                         int32_t capc, int32_t m, const double* cal, int ignore_zero, const uint32_t* flags, double* med,
                         int32_t* status) {
  if (n == 0) return PLAIDHIP_OK;
  const int cap = ctx->num_cu * 16;
  hipLaunchKernelGGL(median_select_kernel, dim3(n < cap ? n : cap), dim3(64), 0, ctx->stream, cand,
                     reinterpret_cast<const uint4*>(cnt), n, nslice, capc, m, cal, ignore_zero, flags, med, status);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_col_medians(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n,
                       int ignore_zero, const uint32_t* flags, double* med, const int32_t* status) {
  if (n == 0) return PLAIDHIP_OK;
  // default: register-resident radix selection up to 6,144 values per column (one wavefront per column), wave-per-column
  // streaming beyond
  // (switch-over measured).  PLAIDHIP_MEDIAN_KERNEL = stream | radix | bits | sample | sort |
  // select forces one of the kernels in the tools/ build (make diag; the older ones are kept as cross-checks)
#ifdef PLAIDHIP_DIAG
  static const char* force = getenv("PLAIDHIP_MEDIAN_KERNEL");
#else
  const char* const force = nullptr;
#endif
  const bool f2 = force && force[0] == 's';
  const bool want_stream = (f2 && force[1] == 't') || (!force && m > 6144);
  const bool want_radix = force && force[0] == 'r';
  const bool want_bits = force && force[0] == 'b';
  const bool want_sample = f2 && force[1] == 'a';
  const bool want_select = f2 && force[1] == 'e';   // "sort" (or anything else): the LDS bitonic sort when it fits
  // wave-per-column register-resident selection up to 6,144 values (measured against the workgroup-per-column radix
  // kernel on 10k columns, one box: 1,000 sets 0.023 vs 0.048 ms, 3,000: 0.058 vs 0.096, 5,000: 0.106 vs 0.179, 6,000:
  // 0.123 vs 0.202); the workgroup kernel stays selectable in the tools/ build as a cross-check
  const bool want_wave = (force && force[0] == 'w') || (!force && m <= 6144);
  if (want_wave && m <= 6144) {
    // (the kernel reads its first ITEMS - 16 rows of 64 values without a bound: the class follows from m, here and only here)
    const int cls = m <= 1024 ? 16 : ((m + 1023) / 1024) * 16;
    switch (cls) {
      case 16: launch_wave<16, 4>(ctx, S, lds, m, n, ignore_zero, flags, med); break;
      case 32: launch_wave<32, 4>(ctx, S, lds, m, n, ignore_zero, flags, med); break;
      case 48: launch_wave<48, 3>(ctx, S, lds, m, n, ignore_zero, flags, med); break;
      case 64: launch_wave<64, 3>(ctx, S, lds, m, n, ignore_zero, flags, med); break;
      case 80: launch_wave<80, 2>(ctx, S, lds, m, n, ignore_zero, flags, med); break;
      default: launch_wave<96, 2>(ctx, S, lds, m, n, ignore_zero, flags, med); break;
    }
  } else if (want_stream) {
#ifdef PLAIDHIP_DIAG
    static const char* wg_env = getenv("PLAIDHIP_STREAM_WGS");
    const int cap = ctx->num_cu * (wg_env ? atoi(wg_env) : 8);
#else
    const int cap = ctx->num_cu * 8;                      // 8 workgroups x 4 wavefronts per CU
#endif
    const int need = (n + 3) / 4;
    // candidate lists of the sampled start: a quarter of a column per wavefront in flight (the sample interval holds
    // about a sixth; a list that overflows is not used and the column is swept twice as before)
    const int grid = need < cap ? need : cap;
    const int32_t ccap = m > 4 * 1024 ? ((m / 4 + 63) & ~63) : 0;
    unsigned long long* cand = nullptr;
    if (ccap > 0) {
      const int rc = ensure_workspace(ctx, (size_t)grid * 4 * (size_t)ccap * 8);
      if (rc != PLAIDHIP_OK) return rc;
      cand = reinterpret_cast<unsigned long long*>(ctx->ws);
    }
    // sample size: 512 values up to 32,768 sets, 1,024 beyond (a narrower bracket: 12.5 % instead of 17.6 % of the column
    // become candidates; measured on 8,192 columns, one box: 50k sets 0.975 -> 0.89 ms, 20k equal, 8k 0.160 -> 0.177: the two
    // bracket selections cost 30 us per column at 512 values and 50 us at 1,024; 2,048 values lose everywhere)
#ifdef PLAIDHIP_DIAG
    static const char* sc_env = getenv("PLAIDHIP_SAMPLE_CHUNKS");
    const int sc = sc_env ? atoi(sc_env) : (m > 32768 ? 16 : 8);
#else
    const int sc = m > 32768 ? 16 : 8;
#endif
    if (sc == 16)
      hipLaunchKernelGGL((col_medians_stream_kernel<1024, 16>), dim3(grid), dim3(256), 0, ctx->stream, S,
                         lds, m, n, ignore_zero, flags, med, cand, ccap, median_stamps(), status);
    else
      hipLaunchKernelGGL((col_medians_stream_kernel<1024, 8>), dim3(grid), dim3(256), 0, ctx->stream, S,
                         lds, m, n, ignore_zero, flags, med, cand, ccap, median_stamps(), status);
  } else if (want_radix && m <= 16384) {
    if (m <= 2048) launch_radix<256, 8>(ctx, S, lds, m, n, ignore_zero, flags, med);
    else if (m <= 4096) launch_radix<256, 16>(ctx, S, lds, m, n, ignore_zero, flags, med);
    // (beyond 16 keys per thread the 256-thread kernel drops to 4 waves per SIMD: 512 threads x 10 / 12 keys measured
    //  5 % / 9 % faster at m = 5,000 / 6,000, and 20-60 % slower than 256 threads below 4,096)
    else if (m <= 5120) launch_radix<512, 10>(ctx, S, lds, m, n, ignore_zero, flags, med);
    else if (m <= 6144) launch_radix<512, 12>(ctx, S, lds, m, n, ignore_zero, flags, med);
    else launch_radix<512, 32>(ctx, S, lds, m, n, ignore_zero, flags, med);
  } else if (want_sample) {
    // workgroup-per-column sample-bracket selection (superseded by the streaming kernel's sampled start): BLOCK 512 up to 16k sets, 1024 beyond; the sample grows with m so
    // that the expected number of keys inside the bracket (4 m / sqrt(samples)) stays below cap/2
    const int block = m <= 16384 ? 512 : 1024;
    const int cap = block * 8;
    int sp = 1;
    while (sp < 16 && 4.0 * m / sqrt((double)block * sp) > cap / 2) sp *= 2;
    const size_t smem = ((size_t)block * sp + cap) * 8 + 8 * 4 + 258 * 4 + 16;
#define PLAIDHIP_LAUNCH_SAMPLE(B, I, PER_CU)                                                             \
  {                                                                                                       \
    PH_FULL_LDS(ctx, (&col_medians_sample_kernel<B, I>));                                                 \
    const int cap_grid = ctx->num_cu * PER_CU * 4;                                                        \
    hipLaunchKernelGGL((col_medians_sample_kernel<B, I>), dim3(n < cap_grid ? n : cap_grid), dim3(B), smem, \
                       ctx->stream, S, lds, m, n, ignore_zero, flags, med, sp, cap);                      \
  }
    if (m <= 4096) PLAIDHIP_LAUNCH_SAMPLE(512, 8, 4)
    else if (m <= 8192) PLAIDHIP_LAUNCH_SAMPLE(512, 16, 4)
    else if (m <= 16384) PLAIDHIP_LAUNCH_SAMPLE(512, 32, 2)
    else PLAIDHIP_LAUNCH_SAMPLE(1024, 0, 2)
#undef PLAIDHIP_LAUNCH_SAMPLE
  } else if (want_bits && m <= 65536) {
    if (m <= 2048) launch_bits<256, 8>(ctx, S, lds, m, n, ignore_zero, flags, med);
    else if (m <= 6144) launch_bits<256, 24>(ctx, S, lds, m, n, ignore_zero, flags, med);
    else if (m <= 16384) launch_bits<512, 32>(ctx, S, lds, m, n, ignore_zero, flags, med);
    else if (m <= 32768) launch_bits<1024, 32>(ctx, S, lds, m, n, ignore_zero, flags, med);
    else launch_bits<1024, 64>(ctx, S, lds, m, n, ignore_zero, flags, med);
  } else if (!want_select && m <= kMaxLdsGenes) {
    PH_FULL_LDS(ctx, &col_medians_lds_kernel);
    const int block = m > 8192 ? 1024 : (m > 2048 ? 512 : 256);
    const int32_t key_slots = (m + 1) & ~1;
    const size_t smem = (size_t)key_slots * 8 + 16;
    hipLaunchKernelGGL(col_medians_lds_kernel, dim3(n), dim3(block), smem, ctx->stream, S, lds, m, n,
                       ignore_zero, flags, med, key_slots);
  } else {
    hipLaunchKernelGGL(col_medians_select_kernel, dim3(n), dim3(1024), 0, ctx->stream, S, lds, m, n,
                       ignore_zero, flags, med);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_sum(plaidhip_ctx* ctx, const double* v, int64_t count, double* out) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, v, count, out);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_max(plaidhip_ctx* ctx, const double* v, int64_t count, double* out) {
  hipLaunchKernelGGL(max_kernel, dim3(1), dim3(1024), 0, ctx->stream, v, count, out);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_shift_columns(plaidhip_ctx* ctx, double* S, int64_t lds, int32_t m, int32_t n,
                         const double* med, double add, const double* red) {
  ctx->fmed.valid = false;   // (S changes: the candidates of a fused crossprod are history)
  if (n == 0 || m == 0) return PLAIDHIP_OK;
  // workgroups per column: one trip of 2,048 values each, up to 32 -- a 50,000-set column is one trip for every workgroup.
  // One grid row per column while the grid allows it (65,535 rows); beyond that 2,048 rows that each walk ~n / 2,048
  // columns (with 65,535 rows a third of them would walk two columns and the rest one).  Measured, late round 4
  // (tools/bench_shift.py big, PLAIDHIP_SHIFT_BX / _BY in the tools build; it was 16 workgroups and 32,768 rows):
  // same box, old -> new: 100,000 x 50,000: 14.44 -> 14.23 ms (13.1 on another box); 8,192 x 50,000: 1.183 -> 1.14 ms;
  // 8,192 x 49,999 (every other column misaligned): 1.325 -> 1.16 ms; 10,000 x 61,459: 1.94 -> 1.82 ms; 10,000 x 5,000:
  // unchanged (three workgroups per column either way).
  int bx = (m / 2 + 256 * 4 - 1) / (256 * 4);
  if (bx < 1) bx = 1;
  int bx_cap = 32;
  int by = n <= 65535 ? n : 2048;
#ifdef PLAIDHIP_DIAG
  if (const char* e = getenv("PLAIDHIP_SHIFT_BX")) bx_cap = atoi(e);
  if (const char* e = getenv("PLAIDHIP_SHIFT_BY")) by = n < atoi(e) ? n : atoi(e);   // (rows = min(n, value))
#endif
  if (bx > bx_cap) bx = bx_cap;
  hipLaunchKernelGGL(shift_columns_kernel, dim3(bx, by), dim3(256), 0, ctx->stream, S, lds, m, n, med, add, red);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_shift_columns_cast_f32(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n, const double* med,
                                  double add, const double* red, float* out, int64_t ldo) {
  if (n == 0 || m == 0) return PLAIDHIP_OK;
  int bx = (m / 2 + 256 * 4 - 1) / (256 * 4);
  if (bx < 1) bx = 1;
  if (bx > 32) bx = 32;
  const int by = n <= 65535 ? n : 2048;
  hipLaunchKernelGGL(shift_columns_cast_f32_kernel, dim3(bx, by), dim3(256), 0, ctx->stream, S, lds, m, n, med, add, red, out, ldo);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

}  // namespace plaidhip
