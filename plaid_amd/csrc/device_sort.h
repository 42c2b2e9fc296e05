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
