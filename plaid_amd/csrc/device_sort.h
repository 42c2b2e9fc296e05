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
  for (uint32_t k = 2; k <= N; k <<= 1) {
    // flip: within each block of k, element t pairs with k-1-t
    {
      const uint32_t half = k >> 1;
      for (uint32_t q = tid; q < (N >> 1); q += nthr) {
        const uint32_t blk = q / half, t = q - blk * half;
        const uint32_t l = blk * k + t;
        const uint32_t r = blk * k + (k - 1 - t);
        if (r < n) {
          const uint64_t a = keys[l], b = keys[r];
          if (a > b) { keys[l] = b; keys[r] = a; }
        }
      }
      __syncthreads();
    }
    // disperse: distances k/4, k/8, ..., 1
    for (uint32_t h = k >> 2; h >= 1; h >>= 1) {
      for (uint32_t q = tid; q < (N >> 1); q += nthr) {
        const uint32_t blk = q / h, t = q - blk * h;
        const uint32_t l = blk * (h << 1) + t;
        const uint32_t r = l + h;
        if (r < n) {
          const uint64_t a = keys[l], b = keys[r];
          if (a > b) { keys[l] = b; keys[r] = a; }
        }
      }
      __syncthreads();
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
