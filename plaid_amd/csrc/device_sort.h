// Device helpers shared by the rank and median kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plaidhip {

// IEEE-754 double -> order-preserving unsigned key.  -0.0 and +0.0 map to the same key
// (they tie in R's rank()); NaN maps to the all-ones key so it sorts last.
__device__ __forceinline__ uint64_t f64_to_key(double x) {
  if (x != x) return ~0ull;
  if (x == 0.0) x = 0.0;  // canonicalise -0
  uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ double key_to_f64(uint64_t k) {
  uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v) {
  if (v <= 1) return 1;
  return 1u << (32 - __clz(v - 1));
}

// In-place ascending sort of keys[0..n) in LDS by the whole workgroup.
// "Normalised" bitonic network (every comparator puts the minimum at the lower index), so
// the sequence can be padded VIRTUALLY to a power of two with +inf keys that never move:
// comparators whose upper index is >= n are skipped.  n need not be a power of two.
// Ends with a barrier.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* keys, uint32_t n) {
  const uint32_t tid = threadIdx.x;
  const uint32_t nthr = blockDim.x;
  const uint32_t N = next_pow2(n);
  __syncthreads();
  for (uint32_t lk = 1; (1u << lk) <= N; ++lk) {
    const uint32_t k = 1u << lk;
    {
      // flip: element t of a k-block pairs with k-1-t; only comparators whose upper index is < n
      const uint32_t lh = lk - 1, half = k >> 1;
      const uint32_t full = n >> lk, rem = n & (k - 1);
      const uint32_t extra = rem > half ? rem - half : 0;
      const uint32_t npairs = (full << lh) + extra;
      for (uint32_t q = tid; q < npairs; q += nthr) {
        uint32_t blk = q >> lh, t = q & (half - 1);
        if (blk == full) t += (k - rem);
        const uint32_t l = (blk << lk) + t;
        const uint32_t r = (blk << lk) + (k - 1 - t);
        const uint64_t a = keys[l], b = keys[r];
        if (a > b) { keys[l] = b; keys[r] = a; }
      }
      __syncthreads();
    }
    for (uint32_t lh = lk >= 2 ? lk - 2 : 0, go = lk >= 2; go; go = lh > 0, lh = lh ? lh - 1 : 0) {
      const uint32_t h = 1u << lh;
      const uint32_t full = n >> (lh + 1), rem = n & ((h << 1) - 1);
      const uint32_t extra = rem > h ? rem - h : 0;
      const uint32_t npairs = (full << lh) + extra;
      for (uint32_t q = tid; q < npairs; q += nthr) {
        const uint32_t blk = q >> lh, t = q & (h - 1);
        const uint32_t l = (blk << (lh + 1)) + t;
        const uint32_t r = l + h;
        const uint64_t a = keys[l], b = keys[r];
        if (a > b) { keys[l] = b; keys[r] = a; }
      }
      __syncthreads();
      if (lh == 0) break;
    }
  }
}

// Same network on DOUBLES: a compare-exchange is v_min_f64 + v_max_f64 (two VALU ops, no
// 64-bit integer compare / select chain).  Callers must have removed NaN (mapped to +inf and
// counted) and canonicalised -0 to +0.  Block/pair indices use shifts (k, h are powers of two)
// and the pair loop stops at the last comparator that touches a real element.
__device__ __forceinline__ void bitonic_sort_f64_lds(double* keys, uint32_t n) {
  const uint32_t tid = threadIdx.x;
  const uint32_t nthr = blockDim.x;
  const uint32_t N = next_pow2(n);
  __syncthreads();
  for (uint32_t lk = 1; (1u << lk) <= N; ++lk) {
    const uint32_t k = 1u << lk;
    {
      // flip: element t of a k-block pairs with k-1-t.  A pair is real iff its upper index < n.
      const uint32_t lh = lk - 1, half = k >> 1;
      // pairs of full blocks + pairs of the last partial block whose partner exists
      const uint32_t full = n >> lk, rem = n & (k - 1);
      const uint32_t extra = rem > half ? rem - half : 0;   // r = base + k-1-t < n  <=>  t >= k - rem
      const uint32_t npairs = (full << lh) + extra;
      for (uint32_t q = tid; q < npairs; q += nthr) {
        uint32_t blk = q >> lh, t = q & (half - 1);
        if (blk == full) t += (k - rem);                   // partial block: only t in [k-rem, half)
        const uint32_t l = (blk << lk) + t;
        const uint32_t r = (blk << lk) + (k - 1 - t);
        const double a = keys[l], b = keys[r];
        keys[l] = fmin(a, b);
        keys[r] = fmax(a, b);
      }
      __syncthreads();
    }
    for (uint32_t lh = lk >= 2 ? lk - 2 : 0, go = lk >= 2; go; go = lh > 0, lh = lh ? lh - 1 : 0) {
      const uint32_t h = 1u << lh;
      // disperse: l = blk*2h + t, r = l + h, real iff r < n
      const uint32_t full = n >> (lh + 1), rem = n & ((h << 1) - 1);
      const uint32_t extra = rem > h ? rem - h : 0;
      const uint32_t npairs = (full << lh) + extra;
      for (uint32_t q = tid; q < npairs; q += nthr) {
        const uint32_t blk = q >> lh, t = q & (h - 1);
        const uint32_t l = (blk << (lh + 1)) + t;
        const uint32_t r = l + h;
        const double a = keys[l], b = keys[r];
        keys[l] = fmin(a, b);
        keys[r] = fmax(a, b);
      }
      __syncthreads();
      if (lh == 0) break;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Register-blocked version of the same normalised bitonic network for LDS-resident columns.
// The workgroup has T = N/32 threads (N = network size, a power of two >= 2048); every pass a
// thread pulls 32 keys into registers, runs up to FIVE network substages on them
// (v_min_f64 / v_max_f64 compare-exchanges) and writes them back: 26 LDS round trips for
// N = 32768 instead of 120.  Which 32 keys a thread owns changes per pass:
//   pass A    : 32 contiguous keys            -> all of merge levels 1..5
//   pass F(k) : keys { b ^ (c << (lk-5)), (b ^ (c << (lk-5))) ^ (k-1) },  c = 0..15
//               -> flip(k) and the disperse substages of bits lk-2 .. lk-5
//   pass D(b) : keys { base | (s << b_lo) }, s = 0..31, b_lo = max(0, b-4)
//               -> the disperse substages of bits b .. b_lo
// LDS positions are XOR-swizzled, phys(i) = i ^ ((i >> 5) & 31): every access pattern above is
// bank-conflict free, and because phys() is GF(2)-linear an element address is
// phys(base) ^ constant, i.e. one v_xor per key (the constants live in SGPRs).  Keys with index
// >= n are virtual +inf: never stored, materialised on load.
__device__ __forceinline__ uint32_t swz(uint32_t i) { return i ^ ((i >> 5) & 31u); }

// compare-exchange = exactly v_min_f64 + v_max_f64.  (fmin()/fmax() make hipcc canonicalise
// both inputs first -- v_max_f64 x, x -- which doubles the VALU work; the keys are never NaN.)
#define PLAIDHIP_CE(a, b)                                                                         \
  {                                                                                               \
    double lo_, hi_;                                                                              \
    asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo_), "=&v"(hi_) : "v"(a), "v"(b)); \
    a = lo_;                                                                                      \
    b = hi_;                                                                                      \
  }

__device__ __forceinline__ double lds_key_load(const unsigned char* lds, uint32_t byte_addr, bool ok) {
  return ok ? *reinterpret_cast<const double*>(lds + byte_addr) : INFINITY;
}

// disperse substages on local bits jmax..0 of a 32-key register block
__device__ __forceinline__ void regs_disperse32(double (&v)[32], int jmax) {
  if (jmax >= 4) {
#pragma unroll
    for (int s = 0; s < 16; ++s) PLAIDHIP_CE(v[s], v[s + 16])
  }
  if (jmax >= 3) {
#pragma unroll
    for (int s = 0; s < 32; ++s) if ((s & 8) == 0) PLAIDHIP_CE(v[s], v[s + 8])
  }
  if (jmax >= 2) {
#pragma unroll
    for (int s = 0; s < 32; ++s) if ((s & 4) == 0) PLAIDHIP_CE(v[s], v[s + 4])
  }
  if (jmax >= 1) {
#pragma unroll
    for (int s = 0; s < 32; ++s) if ((s & 2) == 0) PLAIDHIP_CE(v[s], v[s + 2])
  }
#pragma unroll
  for (int s = 0; s < 32; s += 2) PLAIDHIP_CE(v[s], v[s + 1])
}

// pass A on a contiguous block: merge levels lk = 1..5 (flip + disperses each)
__device__ __forceinline__ void regs_sort32(double (&v)[32]) {
#pragma unroll
  for (int s = 0; s < 32; s += 2) PLAIDHIP_CE(v[s], v[s + 1])                       // lk=1
#pragma unroll
  for (int b = 0; b < 32; b += 4) { PLAIDHIP_CE(v[b], v[b + 3]) PLAIDHIP_CE(v[b + 1], v[b + 2]) }   // flip(4)
#pragma unroll
  for (int s = 0; s < 32; s += 2) PLAIDHIP_CE(v[s], v[s + 1])
#pragma unroll
  for (int b = 0; b < 32; b += 8) {                                                    // flip(8)
#pragma unroll
    for (int t = 0; t < 4; ++t) PLAIDHIP_CE(v[b + t], v[b + 7 - t])
  }
  regs_disperse32(v, 1);
#pragma unroll
  for (int b = 0; b < 32; b += 16) {                                                   // flip(16)
#pragma unroll
    for (int t = 0; t < 8; ++t) PLAIDHIP_CE(v[b + t], v[b + 15 - t])
  }
  regs_disperse32(v, 2);
#pragma unroll
  for (int t = 0; t < 16; ++t) PLAIDHIP_CE(v[t], v[31 - t])                            // flip(32)
  regs_disperse32(v, 3);
}

template <bool CHECK>
__device__ __forceinline__ void bitonic_pass_D(unsigned char* lds, uint32_t n, uint32_t tid, int b_top) {
  const int b_lo = b_top >= 4 ? b_top - 4 : 0;
  const uint32_t t_lo = tid & ((1u << b_lo) - 1u), t_hi = tid >> b_lo;
  const uint32_t base = (t_hi << (b_lo + 5)) | t_lo;
  const uint32_t P0 = swz(base) << 3;
  double v[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const uint32_t f = (uint32_t)s << b_lo;               // wave-uniform -> SALU
    v[s] = lds_key_load(lds, P0 ^ (swz(f) << 3), !CHECK || (base | f) < n);
  }
  regs_disperse32(v, b_top - b_lo);
  // recompute addresses / predicates for the store (opaque copies stop the compiler from keeping
  // 32 address registers and 32 lane masks alive across the compare-exchange network -> spills)
  uint32_t P1 = P0, base1 = base;
  asm volatile("" : "+v"(P1), "+v"(base1));
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const uint32_t f = (uint32_t)s << b_lo;
    if (!CHECK || (base1 | f) < n)
      *reinterpret_cast<double*>(lds + (P1 ^ (swz(f) << 3))) = v[s];
  }
}

template <bool CHECK>
__device__ __forceinline__ void bitonic_pass_F(unsigned char* lds, uint32_t n, uint32_t tid, int lk) {
  const int sh = lk - 5;
  const uint32_t M = (1u << lk) - 1u;
  const uint32_t t_lo = tid & ((1u << sh) - 1u), t_hi = tid >> sh;
  const uint32_t base = (t_hi << lk) | t_lo;
  const uint32_t P0 = swz(base) << 3, KM = swz(M) << 3;
  double v[32];   // v[c] = key(base | c<<sh), v[16+c] = its flip partner key((base | c<<sh) ^ M)
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const uint32_t f = (uint32_t)c << sh;
    const uint32_t K = swz(f) << 3;
    v[c] = lds_key_load(lds, P0 ^ K, !CHECK || (base | f) < n);
    v[16 + c] = lds_key_load(lds, P0 ^ K ^ KM, !CHECK || ((base | f) ^ M) < n);
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) PLAIDHIP_CE(v[c], v[16 + c])                           // flip(k)
  // disperse bits lk-2 .. lk-5 = local bits 3..0 of c.  In the partner half the index is the
  // complement, so the LOWER index is the one with the local bit SET.
#pragma unroll
  for (int j = 3; j >= 0; --j) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if ((c & (1 << j)) == 0) {
        PLAIDHIP_CE(v[c], v[c | (1 << j)])
        PLAIDHIP_CE(v[16 + (c | (1 << j))], v[16 + c])
      }
  }
  uint32_t P1 = P0, base1 = base;
  asm volatile("" : "+v"(P1), "+v"(base1));
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const uint32_t f = (uint32_t)c << sh;
    const uint32_t K = swz(f) << 3;
    if (!CHECK || (base1 | f) < n) *reinterpret_cast<double*>(lds + (P1 ^ K)) = v[c];
    if (!CHECK || ((base1 | f) ^ M) < n) *reinterpret_cast<double*>(lds + (P1 ^ K ^ KM)) = v[16 + c];
  }
}

// Sorts the n keys already placed at swizzled positions by pass A (see the rank kernel) -- i.e.
// runs merge levels lk = 6 .. L for the network size N = 2^L = 32 * blockDim.x.  On return the
// sorted keys sit at their NATURAL positions lds[0..n).
__device__ __forceinline__ void bitonic_finish_regs(unsigned char* lds, uint32_t n, int L) {
  for (int lk = 6; lk <= L; ++lk) {
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));     // opaque per level: no hoisting of per-thread address sets out of the loops
    __syncthreads();
    // a k-block that lies entirely below n needs no bounds checks (wave-uniform decision is not
    // possible in general: threads of one wave sit in one block only when 2^lk >= 2048 -> decide
    // per thread, both branches are the same code apart from the predicates)
    const uint32_t blk_beg = (tid >> (lk - 5)) << lk, blk_end = blk_beg + (1u << lk);
    if (blk_end <= n) bitonic_pass_F<false>(lds, n, tid, lk);
    else if (blk_beg < n) bitonic_pass_F<true>(lds, n, tid, lk);      // else: only virtual +inf keys
    for (int b_top = lk - 6; b_top >= 0; b_top -= 5) {
      __syncthreads();
      const int b_lo = b_top >= 4 ? b_top - 4 : 0;
      const uint32_t span_beg = (tid >> b_lo) << (b_lo + 5), span_end = span_beg + (32u << b_lo);
      if (span_end <= n) bitonic_pass_D<false>(lds, n, tid, b_top);
      else if (span_beg < n) bitonic_pass_D<true>(lds, n, tid, b_top);
    }
  }
  __syncthreads();
  // un-swizzle: the searches that follow want the keys at their natural positions.  Thread t moves keys
  // t, t + T, t + 2T, ...: both the swizzled reads and the natural writes of 32 neighbouring lanes fall into
  // 32 different banks.  (Storing the last merge pass straight to natural positions -- 32 consecutive keys
  // per thread -- is a 32-way bank conflict on every store and cost a third of the whole kernel.)
  // (eight keys per thread and round: the rounds touch disjoint aligned 32-key groups, one barrier each)
  {
    const uint32_t t = threadIdx.x, T = blockDim.x;
    for (int k0 = 0; k0 < 32; k0 += 8) {
      if ((uint32_t)k0 * T >= n) break;      // uniform: nothing left
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t i = t + (uint32_t)(k0 + k) * T;
        v[k] = (i < n) ? *reinterpret_cast<const double*>(lds + (swz(i) << 3)) : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t i = t + (uint32_t)(k0 + k) * T;
        if (i < n) *reinterpret_cast<double*>(lds + (i << 3)) = v[k];
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ uint32_t lower_bound_f64(const double* sorted, uint32_t n, double x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sorted[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t upper_bound_f64(const double* sorted, uint32_t n, double x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sorted[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// number of keys < key / <= key in sorted[0..n)
__device__ __forceinline__ uint32_t lower_bound_lds(const uint64_t* sorted, uint32_t n, uint64_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sorted[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t upper_bound_lds(const uint64_t* sorted, uint32_t n, uint64_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sorted[mid] <= key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ double wave_max_f64(double v) {
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = (o > v) ? o : v;
  }
  return v;
}

}  // namespace plaidhip
