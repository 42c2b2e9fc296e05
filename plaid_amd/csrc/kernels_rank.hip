// Per-sample column ranking: colranks() dense branch (R/plaid.R:611-619 ->
// matrixStats::colRanks) and sparse_colranks() (R/plaid.R:631-650 -> base::rank over the
// stored non-zeros of each CSC column).  gfx950 / wave64 only.
//
// Kernel shape: one workgroup per column.  The column's doubles (exact IEEE order, -0
// canonicalised to +0, NaN parked at +inf and counted) are sorted IN PLACE in LDS (20k
// doubles = 160 KB fill the CU's LDS exactly, so the sort carries no payload) by a bitonic
// network whose compare-exchange is v_min_f64 + v_max_f64; LDS-resident columns use the
// register-blocked form (32 keys per thread, up to five substages per LDS round trip).  Ranks are then recovered by binary search of each element's key in the sorted
// keys: lb = #{x_j < x_i}, ub = #{x_j <= x_i};
//   min = lb + 1,  max = ub,  average = (lb + 1 + ub) / 2      (bit-exact half-integers)
// which is the definition of rank(ties.method=) for NaN-free input.  NaN inputs return NaN
// (R: NA stays NA) and do not disturb the ranks of the others.
#include "common.h"
#include "device_sort.h"
#include "rank_bucket.h"

namespace plaidhip {

__device__ __forceinline__ double rank_from_bounds(uint32_t lb, uint32_t ub, int ties) {
  if (ties == PLAIDHIP_TIES_MIN) return (double)(lb + 1);
  if (ties == PLAIDHIP_TIES_MAX) return (double)ub;
  return 0.5 * (double)(lb + 1 + ub);
}

__device__ __forceinline__ double sign_of(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); }

// keys: LDS (or, for the large-column fallback, a global scratch slice) with room for cnt keys;
// scratch: 2 uint32 + nwaves doubles in LDS.
template <bool GLOBAL_KEYS>
__global__ void __launch_bounds__(1024)
colranks_f64_kernel(const double* __restrict__ Xv,  // values: dense matrix or CSC @x
                    int64_t ldx, int32_t g_dense,   // dense: column stride / length
                    const int32_t* __restrict__ Xp, // CSC: column pointers (nullptr for dense)
                    int32_t n, int ties, int is_signed, double power, double* __restrict__ R,
                    int64_t ldr, double* __restrict__ colmax, uint64_t* gkeys, int64_t gkeys_stride,
                    const int32_t* __restrict__ Xi_dense,  // non-null: CSC input, DENSE result (zeros ranked)
                    double* __restrict__ dense_scratch,    // g_dense doubles per workgroup
                    const int32_t* __restrict__ col_list,  // non-null: rank only these columns (left over by the
                    const int32_t* __restrict__ col_count) { // bucket kernel; both live in device memory)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;

  double* keys;
  uint32_t* s_u32;
  double* s_f64;
  if constexpr (GLOBAL_KEYS) {
    keys = reinterpret_cast<double*>(gkeys) + (int64_t)blockIdx.x * gkeys_stride;
    s_u32 = reinterpret_cast<uint32_t*>(smem_raw);
    s_f64 = reinterpret_cast<double*>(smem_raw + 16);
  } else {
    keys = reinterpret_cast<double*>(smem_raw);
    // scratch sits behind the keys; offset supplied through gkeys_stride (in keys)
    s_u32 = reinterpret_cast<uint32_t*>(smem_raw + gkeys_stride * 8);
    s_f64 = reinterpret_cast<double*>(smem_raw + gkeys_stride * 8 + 16);
  }

  const int ncols = (col_list != nullptr) ? *col_count : n;
  for (int ci = blockIdx.x; ci < ncols; ci += gridDim.x) {
    const int c = (col_list != nullptr) ? col_list[ci] : ci;
    const double* xc;
    double* rc;
    uint32_t cnt;
    if (Xi_dense != nullptr) {
      // colranks(sparse X, keep.zero=FALSE): the reference ranks the densified column
      // (sparseMatrixStats::colRanks, R/plaid.R:603-609).  Densify into this workgroup's
      // scratch column, then proceed exactly like the dense branch.
      double* dcol = dense_scratch + (int64_t)blockIdx.x * g_dense;
      for (int i = tid; i < g_dense; i += nthr) dcol[i] = 0.0;
      __syncthreads();
      const int p0 = Xp[c], p1 = Xp[c + 1];
      for (int p = p0 + tid; p < p1; p += nthr) dcol[Xi_dense[p]] = Xv[p];
      __syncthreads();
      cnt = (uint32_t)g_dense;
      xc = dcol;
      rc = R + (int64_t)c * ldr;
    } else if (Xp != nullptr) {
      const int p0 = Xp[c];
      cnt = (uint32_t)(Xp[c + 1] - p0);
      xc = Xv + p0;
      rc = R + p0;
    } else {
      cnt = (uint32_t)g_dense;
      xc = Xv + (int64_t)c * ldx;
      rc = R + (int64_t)c * ldr;
    }
    if (tid == 0) s_u32[0] = 0;
    __syncthreads();
    uint32_t my_nan = 0;
    for (uint32_t i = tid; i < cnt; i += nthr) {
      double x = xc[i];
      if (is_signed) x = fabs(x);
      const bool isnan_ = (x != x);
      my_nan += isnan_;
      keys[i] = isnan_ ? INFINITY : (x + 0.0);   // NaN sorts last (counted); -0 -> +0
    }
    if (my_nan) atomicAdd(&s_u32[0], my_nan);
    bitonic_sort_f64_lds(keys, cnt);  // starts and ends with a barrier
    const uint32_t nvalid = cnt - s_u32[0];

    double vmax = (Xp != nullptr && Xi_dense == nullptr) ? 0.0 : -INFINITY;   // sparse ranks: the implicit zeros
    for (uint32_t i = tid; i < cnt; i += nthr) {
      const double x0 = xc[i];
      const double x = is_signed ? fabs(x0) : x0;
      double r;
      if (x != x) {
        r = __longlong_as_double(0x7ff8000000000000ll);
      } else {
        const uint32_t lb = lower_bound_f64(keys, nvalid, x);
        const uint32_t ub = upper_bound_f64(keys, nvalid, x);
        r = rank_from_bounds(lb, ub, ties);
        if (power != 1.0) r = pow(r, power);
        vmax = (r > vmax) ? r : vmax;            // max |value| of the column (before the sign goes on)
        if (is_signed) r *= sign_of(x0);
      }
      rc[i] = r;
    }
    if (colmax != nullptr) {
      vmax = wave_max_f64(vmax);
      if (lane == 0) s_f64[wave] = vmax;
      __syncthreads();
      if (tid == 0) {
        double v = s_f64[0];
        for (int w = 1; w < nwaves; ++w) v = (s_f64[w] > v) ? s_f64[w] : v;
        colmax[c] = v;
      }
    }
    __syncthreads();
  }
}

// LDS-resident columns: register-blocked sort (device_sort.h), T = N/32 threads, N = 2^L.
__global__ void __launch_bounds__(1024)
colranks_regs_kernel(const double* __restrict__ Xv, int64_t ldx, int32_t g_dense,
                     const int32_t* __restrict__ Xp, int32_t n, int ties, int is_signed, double power,
                     double* __restrict__ R, int64_t ldr, double* __restrict__ colmax, int L,
                     int32_t key_bytes, const int32_t* __restrict__ Xi_dense,
                     double* __restrict__ dense_scratch, const int32_t* __restrict__ col_list,
                     const int32_t* __restrict__ col_count) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  unsigned char* keys = smem_raw;
  uint32_t* s_u32 = reinterpret_cast<uint32_t*>(smem_raw + key_bytes);
  double* s_f64 = reinterpret_cast<double*>(smem_raw + key_bytes + 16);

  const int ncols = (col_list != nullptr) ? *col_count : n;
  for (int ci = blockIdx.x; ci < ncols; ci += gridDim.x) {
    const int c = (col_list != nullptr) ? col_list[ci] : ci;
    const double* xc;
    double* rc;
    uint32_t cnt;
    if (Xi_dense != nullptr) {
      double* dcol = dense_scratch + (int64_t)blockIdx.x * g_dense;
      for (int i = tid; i < g_dense; i += nthr) dcol[i] = 0.0;
      __syncthreads();
      const int p0 = Xp[c], p1 = Xp[c + 1];
      for (int p = p0 + tid; p < p1; p += nthr) dcol[Xi_dense[p]] = Xv[p];
      __syncthreads();
      cnt = (uint32_t)g_dense;
      xc = dcol;
      rc = R + (int64_t)c * ldr;
    } else if (Xp != nullptr) {
      const int p0 = Xp[c];
      cnt = (uint32_t)(Xp[c + 1] - p0);
      xc = Xv + p0;
      rc = R + p0;
    } else {
      cnt = (uint32_t)g_dense;
      xc = Xv + (int64_t)c * ldx;
      rc = R + (int64_t)c * ldr;
    }
    if (tid == 0) s_u32[0] = 0;
    __syncthreads();
    // ---- coalesced load -> LDS (swizzled positions), canonicalise, count NaN ------------
    {
      uint32_t my_nan = 0;
      for (uint32_t i = tid; i < cnt; i += nthr) {
        double x = xc[i];
        if (is_signed) x = fabs(x);
        const bool isnan_ = (x != x);
        my_nan += isnan_;
        *reinterpret_cast<double*>(keys + (swz(i) << 3)) = isnan_ ? INFINITY : (x + 0.0);   // -0 -> +0
      }
      if (my_nan) atomicAdd(&s_u32[0], my_nan);
    }
    __syncthreads();
    // ---- pass A: 32 contiguous keys per thread: merge levels 1..5 in registers ----------
    {
      uint32_t tid_ = (uint32_t)tid;
      asm volatile("" : "+v"(tid_));   // per-column opaque copy: keeps LICM from hoisting (and spilling) 32 addresses
      const uint32_t base = tid_ * 32u;
      if (base < cnt) {
        double v[32];
        const uint32_t P0 = swz(base) << 3;
#pragma unroll
        for (int s = 0; s < 32; ++s)
          v[s] = lds_key_load(keys, P0 ^ ((uint32_t)s << 3), base + s < cnt);
        regs_sort32(v);
        uint32_t P1 = P0, base1 = base;
        asm volatile("" : "+v"(P1), "+v"(base1));
#pragma unroll
        for (int s = 0; s < 32; ++s)
          if (base1 + s < cnt) *reinterpret_cast<double*>(keys + (P1 ^ ((uint32_t)s << 3))) = v[s];
      }
    }
    bitonic_finish_regs(keys, cnt, L);   // starts and ends with a barrier
    const uint32_t nvalid = cnt - s_u32[0];

    // ---- ranks: lb = #{keys < x} by a branch-free binary search with a fixed trip count (two
    //      elements interleaved for ILP); ub = #{keys <= x} by galloping from lb (tie runs are
    //      short unless the data is tie-heavy, and then the gallop is still logarithmic)
    const double* sk = reinterpret_cast<const double*>(keys);
    int steps = 0;
    while ((1u << steps) <= nvalid) ++steps;              // ceil(log2(nvalid + 1))
    double vmax = (Xp != nullptr && Xi_dense == nullptr) ? 0.0 : -INFINITY;   // sparse ranks: the implicit zeros
    for (uint32_t i0 = tid; i0 < cnt; i0 += 2 * nthr) {
      const uint32_t i1 = i0 + nthr;
      const bool has1 = i1 < cnt;
      const double xa0 = xc[i0], xb0 = has1 ? xc[i1] : 0.0;
      const double xa = is_signed ? fabs(xa0) : xa0, xb = is_signed ? fabs(xb0) : xb0;
      uint32_t la = 0, lb_ = 0;
      for (int k = steps - 1; k >= 0; --k) {
        const uint32_t h = 1u << k;
        const uint32_t pa = la + h, pb = lb_ + h;
        const bool oka = pa <= nvalid, okb = pb <= nvalid;
        const double ka = oka ? sk[pa - 1] : INFINITY, kb = okb ? sk[pb - 1] : INFINITY;
        la = (oka && ka < xa) ? pa : la;
        lb_ = (okb && kb < xb) ? pb : lb_;
      }
      auto upper_from = [&](uint32_t lo, double x) -> uint32_t {
        // smallest u >= lo with sk[u] > x (or nvalid): gallop, then bisect the last interval
        uint32_t step = 1, u = lo;
        while (u + step <= nvalid && sk[u + step - 1] <= x) { u += step; step <<= 1; }
        for (step >>= 1; step >= 1; step >>= 1)
          if (u + step <= nvalid && sk[u + step - 1] <= x) u += step;
        return u;
      };
      {
        double r;
        if (xa != xa) {
          r = __longlong_as_double(0x7ff8000000000000ll);
        } else {
          r = rank_from_bounds(la, upper_from(la, xa), ties);
          if (power != 1.0) r = pow(r, power);
          vmax = (r > vmax) ? r : vmax;          // max |value| of the column
          if (is_signed) r *= sign_of(xa0);
        }
        rc[i0] = r;
      }
      if (has1) {
        double r;
        if (xb != xb) {
          r = __longlong_as_double(0x7ff8000000000000ll);
        } else {
          r = rank_from_bounds(lb_, upper_from(lb_, xb), ties);
          if (power != 1.0) r = pow(r, power);
          vmax = (r > vmax) ? r : vmax;
          if (is_signed) r *= sign_of(xb0);
        }
        rc[i1] = r;
      }
    }
    if (colmax != nullptr) {
      vmax = wave_max_f64(vmax);
      if (lane == 0) s_f64[wave] = vmax;
      __syncthreads();
      if (tid == 0) {
        double v = s_f64[0];
        for (int w = 1; w < nwaves; ++w) v = (s_f64[w] > v) ? s_f64[w] : v;
        colmax[c] = v;
      }
    }
    __syncthreads();
  }
}

__global__ void fill_f64_kernel(double* p, int32_t n, double v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

#ifdef PLAIDHIP_DIAG
static unsigned long long* g_rank_dbg = nullptr;   // tools/ build: per-phase stamps of the bucket kernel
void debug_set_rank_stamps(void* dbg) { g_rank_dbg = static_cast<unsigned long long*>(dbg); }
#endif

// the sorting-network kernels: all columns (col_list == nullptr) or the columns the bucket kernel left over
static int launch_network(plaidhip_ctx* ctx, const double* Xv, int64_t ldx, int32_t g_dense, const int32_t* Xp,
                          int32_t n, int32_t max_len, int ties, int is_signed, double power, double* R, int64_t ldr,
                          double* colmax, const int32_t* Xi_dense, double* dscratch, int grid_cap, size_t ws_off,
                          const int32_t* col_list, const int32_t* col_count) {
  const int block = max_len > 8192 ? 1024 : (max_len > 2048 ? 512 : 256);
  const size_t scratch = 16 + 16 * sizeof(double);
  if (max_len > 8192 && max_len <= kMaxLdsGenes) {
    // long LDS-resident columns: register-blocked sort (measured faster from N = 16384 up;
    // below that the plain LDS network with more workgroups per CU wins)
    PH_FULL_LDS(ctx, &colranks_regs_kernel);
    int L = 11;                                  // network size N = 2^L >= 2048, T = N/32 threads
    while ((1 << L) < max_len) ++L;
    const int threads = (1 << L) / 32;
    const int32_t key_bytes = ((max_len + 31) & ~31) * 8;     // swizzle permutes inside 32-key groups
    const size_t smem = (size_t)key_bytes + scratch;
    hipLaunchKernelGGL(colranks_regs_kernel, dim3(grid_cap), dim3(threads), smem, ctx->stream, Xv, ldx,
                       g_dense, Xp, n, ties, is_signed, power, R, ldr, colmax, L, key_bytes, Xi_dense,
                       dscratch, col_list, col_count);
  } else if (max_len <= kMaxLdsGenes) {
    PH_FULL_LDS(ctx, &colranks_f64_kernel<false>);
    const int64_t key_slots = ((int64_t)max_len + 1) & ~1ll;  // keep scratch 16-B aligned
    const size_t smem = (size_t)key_slots * 8 + scratch;
    hipLaunchKernelGGL(colranks_f64_kernel<false>, dim3(grid_cap), dim3(block), smem, ctx->stream, Xv, ldx,
                       g_dense, Xp, n, ties, is_signed, power, R, ldr, colmax, (uint64_t*)nullptr,
                       key_slots, Xi_dense, dscratch, col_list, col_count);
  } else {
    // large columns: keys live in a global scratch slice per workgroup (L2-resident)
    const int grid = n < 2 * ctx->num_cu ? n : 2 * ctx->num_cu;
    const int64_t stride = ((int64_t)max_len + 1) & ~1ll;
    hipLaunchKernelGGL(colranks_f64_kernel<true>, dim3(grid), dim3(1024), scratch, ctx->stream, Xv,
                       ldx, g_dense, Xp, n, ties, is_signed, power, R, ldr, colmax,
                       reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(ctx->ws) + ws_off), stride,
                       Xi_dense, dscratch, col_list, col_count);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

template <int BLOCK, int KPT>
static int launch_bucket(plaidhip_ctx* ctx, const RankBucketArgs& a, int32_t max_len, int grid) {
  using L = RankBucketLayout<BLOCK, KPT>;
  const size_t keys_bytes = (size_t)kRankMisc + (size_t)max_len * 8;
  const size_t smem = keys_bytes > (size_t)L::hist_bytes ? keys_bytes : (size_t)L::hist_bytes;
  PH_FULL_LDS(ctx, (&colranks_bucket_kernel<BLOCK, KPT>));
  // persistent: as many workgroups as the chip holds at once (LDS- or thread-limited), each walking its columns with the
  // next column's values requested while the current one's ranks are written (rank_bucket.h)
  int per_cu = (int)(kLdsBytes / (smem ? smem : 1));
  if (per_cu > 2048 / BLOCK) per_cu = 2048 / BLOCK;
  if (per_cu < 1) per_cu = 1;
  bool persistent = (size_t)BLOCK * KPT * 8 > (size_t)64 * 1024;
#ifdef PLAIDHIP_DIAG
  if (getenv("PLAIDHIP_RANK_NOPF")) persistent = false;   // one workgroup per column, nothing prefetched (A/B)
#endif
  if (persistent && grid > ctx->num_cu * per_cu) grid = ctx->num_cu * per_cu;
  hipLaunchKernelGGL((colranks_bucket_kernel<BLOCK, KPT>), dim3(grid), dim3(BLOCK), smem, ctx->stream, a);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

// longest column the bucket kernel takes: 8 bytes per key + its scratch must fit the CU's LDS
constexpr int kMaxBucketKeys = (kLdsBytes - kRankMisc) / 8;   // 20,352

static int launch_ranks(plaidhip_ctx* ctx, const double* Xv, int64_t ldx, int32_t g_dense,
                        const int32_t* Xp, int32_t n, int32_t max_len, int ties, int is_signed,
                        double power, double* R, int64_t ldr, double* colmax,
                        const int32_t* Xi_dense = nullptr) {
  if (n == 0) return PLAIDHIP_OK;
  if (max_len == 0) {
    // nothing to rank; a sparse column's maximum is its implicit zeros (max(rX) of R/plaid.R:251 is then 0)
    if (colmax != nullptr) {
      hipLaunchKernelGGL(fill_f64_kernel, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0,
                         ctx->stream, colmax, n, Xp != nullptr ? 0.0 : -INFINITY);
      PH_HIP(hipGetLastError());
    }
    return PLAIDHIP_OK;
  }
  // kernel choice: the bucket ranker for every column that fits the LDS (measured); the sorting
  // network beyond, as the bucket kernel's fallback for clustered columns, and when the context asks for it
  const bool can_bucket = max_len <= kMaxBucketKeys;
  const bool use_bucket = can_bucket && (ctx->opt_rank_kernel >= 2 || (ctx->opt_rank_kernel == 0 && max_len > 256));
  // workspace: [densify scratch (CSC input, dense result)] [fallback counter + list] [global key scratch]
  int grid_cap = n;
  size_t ws_off = 0;
  if (Xi_dense != nullptr) {
    grid_cap = n < 2 * ctx->num_cu ? n : 2 * ctx->num_cu;
    ws_off = (size_t)grid_cap * (size_t)g_dense * 8;
  }
  const size_t fb_off = ws_off;
  if (use_bucket) ws_off += ((size_t)n * 4 + 16 + 255) & ~(size_t)255;
  size_t ws_need = ws_off;
  if (max_len > kMaxLdsGenes) {
    const int grid = n < 2 * ctx->num_cu ? n : 2 * ctx->num_cu;
    ws_need += (size_t)grid * (size_t)(((int64_t)max_len + 1) & ~1ll) * 8;
  }
  if (ws_need > 0) {
    const int rc = ensure_workspace(ctx, ws_need);
    if (rc != PLAIDHIP_OK) return rc;
  }
  double* dscratch = Xi_dense != nullptr ? reinterpret_cast<double*>(ctx->ws) : nullptr;
  if (!use_bucket)
    return launch_network(ctx, Xv, ldx, g_dense, Xp, n, max_len, ties, is_signed, power, R, ldr, colmax, Xi_dense,
                          dscratch, grid_cap, ws_off, nullptr, nullptr);

  int32_t* fb_count = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ctx->ws) + fb_off);
  int32_t* fb_list = fb_count + 4;
  PH_HIP(hipMemsetAsync(fb_count, 0, 16, ctx->stream));
  RankBucketArgs a{};
  a.Xv = Xv;
  a.ldx = ldx;
  a.g_dense = g_dense;
  a.Xp = Xp;
  a.n = n;
  a.ties = ties;
  a.is_signed = is_signed;
  a.power = power;
  const double q4 = power * 4.0;
  a.pow_q4 = (power != 1.0 && q4 >= 1.0 && q4 <= 16.0 && q4 == (double)(int)q4) ? (int)q4 : 0;
  a.R = R;
  a.ldr = ldr;
  a.colmax = colmax;
  a.Xi_dense = Xi_dense;
  a.dense_scratch = dscratch;
  a.fb_count = fb_count;
  a.fb_list = fb_list;
#ifdef PLAIDHIP_DIAG
  a.dbg = g_rank_dbg;
#endif
  int rc;
  // a 20k-gene column fills the CU's LDS, so ONE workgroup runs per CU: with 1,024 threads x 20 keys it brings four
  // wavefronts per SIMD instead of the two of 512 threads x 40 keys, and the kernel -- 46 % of its wave cycles waiting, its
  // LDS active 21 % of the time (profiles/r03i_pmc_c4_summary.txt) -- has something to hide its latencies with.  The
  // 128-register cap of 1,024 threads costs 19 spilled registers; measured all the same (tools/bench_rank.py, 8,192 columns
  // x 20,000 genes): 1.55 against 1.94 ms tie-free, 1.81 / 2.24 ms with the fused power, 1.62 / 2.02 ms rounded values,
  // 2.27 / 2.67 ms with 95 % zeros.  PLAIDHIP_OPT_RANK_KERNEL = 3 keeps the 512 x 40 shape for comparison.
  if (max_len <= 2048) rc = launch_bucket<256, 8>(ctx, a, max_len, grid_cap);
  else if (max_len <= 4096) rc = launch_bucket<256, 16>(ctx, a, max_len, grid_cap);
  else if (max_len <= 8192) rc = launch_bucket<512, 16>(ctx, a, max_len, grid_cap);
  else if (max_len <= 12288) rc = launch_bucket<512, 24>(ctx, a, max_len, grid_cap);
  else if (ctx->opt_rank_kernel == 3) rc = launch_bucket<512, 40>(ctx, a, max_len, grid_cap);   // (A/B: two waves per SIMD)
  else rc = launch_bucket<1024, 20>(ctx, a, max_len, grid_cap);
  if (rc != PLAIDHIP_OK) return rc;
  // columns the bucket kernel gave up on (clustered values): a small persistent grid reads the device-side list
  const int fb_grid = grid_cap < ctx->num_cu ? grid_cap : ctx->num_cu;
  return launch_network(ctx, Xv, ldx, g_dense, Xp, n, max_len, ties, is_signed, power, R, ldr, colmax, Xi_dense,
                        dscratch, fb_grid, ws_off, fb_list, fb_count);
}

// ---- dense columns beyond the LDS: rank by value partition --------------------------------------------------------------
// The bucket ranker takes columns of at most kMaxBucketKeys (20,352) keys -- what fits the CU's LDS; longer columns fell to
// the sorting network on a global scratch (2.4-3.4e9 keys/s against 7.6e10).  A longer column is cut BY VALUE instead:
// K - 1 splitters (quantiles of a 1,024-key sample, duplicates dropped) define up to K open intervals and as many
// single-value classes {t_i}; the keys of an open interval go to a scratch segment together with their row numbers,
// every segment is ranked by the bucket ranker like a CSC column (launch_ranks), and the rank inside the segment plus
// the number of keys in the classes below is the rank in the column -- exactly, for every ties method.  A key equal to a
// splitter needs no ranking at all (lb = #keys below, ub = lb + #equal): a value that makes up a large part of the column
// (the zeros of a dense single-cell matrix) is almost surely a sample quantile and costs two counters.  An open interval
// that still holds more than kMaxBucketKeys keys sends the column to the network kernel (device-side list).
constexpr int kPartMax = 16;                  // open intervals per column at most
constexpr int kPartTarget = 12288;            // keys per open interval aimed at (slack for the sample's error)
struct PartMeta {
  uint64_t t[kPartMax];                       // distinct splitter keys, ascending
  int32_t less[kPartMax], eq[kPartMax];       // #keys < t_j, #keys == t_j
  int32_t open[kPartMax + 1];                 // sizes of the open intervals (-inf, t_0), (t_0, t_1), ..., (t_{p-1}, +inf)
  int32_t p, fallback;
  double smax;                                // largest value written for the single-value classes
};

__device__ __forceinline__ uint64_t part_key(double x, int is_signed) { return f64_to_key(is_signed ? fabs(x) : x); }

__global__ void __launch_bounds__(1024)
rank_part_count_kernel(const double* __restrict__ X, int64_t ldx, int32_t g_dense, const int32_t* __restrict__ Xp, int32_t n,
                       int is_signed, int K, PartMeta* __restrict__ meta, int32_t* __restrict__ col_open,
                       int32_t* __restrict__ fb_count, int32_t* __restrict__ fb_list, int32_t col0) {
  __shared__ uint64_t skeys[1024];
  __shared__ uint64_t s_t[kPartMax];
  __shared__ uint32_t s_less[kPartMax], s_eq[kPartMax], s_nan;
  __shared__ int s_p;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    // a dense column, or the stored values of a CSC column (sparse_colranks: any length, also 0)
    const double* xc = Xp != nullptr ? X + Xp[c] : X + (int64_t)c * ldx;
    const int32_t g = Xp != nullptr ? Xp[c + 1] - Xp[c] : g_dense;
    skeys[tid] = g > 0 ? part_key(xc[((int64_t)tid * g) >> 10], is_signed) : ~0ull;
    if (tid < kPartMax) { s_less[tid] = 0; s_eq[tid] = 0; }
    if (tid == 0) s_nan = 0;
    bitonic_sort_lds(skeys, 1024);            // starts and ends with a barrier; NaN keys (all ones) sort last
    if (tid == 0) {
      int ns = 0;
      for (int b = 512; b >= 1; b >>= 1) ns += (skeys[ns + b - 1] != ~0ull) ? b : 0;   // number of non-NaN sample keys
      if (ns < 1024 && skeys[ns] != ~0ull) ++ns;
      int p = 0;
      for (int i = 1; i < K && ns > 0; ++i) {
        const uint64_t q = skeys[((int64_t)i * ns) / K];
        if (p == 0 || q != s_t[p - 1]) s_t[p++] = q;
      }
      s_p = p;
    }
    __syncthreads();
    const int p = s_p;
    uint32_t less[kPartMax], eq[kPartMax], nnan = 0;
#pragma unroll
    for (int j = 0; j < kPartMax; ++j) { less[j] = 0; eq[j] = 0; }
    for (int i = tid; i < g; i += 1024) {
      const uint64_t k = part_key(xc[i], is_signed);
      nnan += (k == ~0ull) ? 1u : 0u;
#pragma unroll
      for (int j = 0; j < kPartMax; ++j)
        if (j < p) {
          const uint64_t t = s_t[j];
          less[j] += (k < t) ? 1u : 0u;
          eq[j] += (k == t) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < kPartMax; ++j)
      if (j < p) {
        const uint32_t a = wave_incl_scan_u32(less[j]), b = wave_incl_scan_u32(eq[j]);
        if (lane == 63) { atomicAdd(&s_less[j], a); atomicAdd(&s_eq[j], b); }
      }
    nnan = wave_incl_scan_u32(nnan);
    if (lane == 63) atomicAdd(&s_nan, nnan);
    __syncthreads();
    if (tid == 0) {
      PartMeta& mt = meta[c];
      const int32_t nvalid = g - (int32_t)s_nan;
      int32_t total = 0, worst = 0;
      for (int a = 0; a <= p; ++a) {
        const int32_t below = (a == 0) ? 0 : (int32_t)(s_less[a - 1] + s_eq[a - 1]);      // keys <= t_{a-1}
        const int32_t upto = (a == p) ? nvalid : (int32_t)s_less[a];                      // keys <  t_a
        const int32_t sz = upto - below;
        mt.open[a] = sz;
        total += sz;
        worst = sz > worst ? sz : worst;
      }
      for (int a = p + 1; a <= kPartMax; ++a) mt.open[a] = 0;
      for (int j = 0; j < kPartMax; ++j) { mt.t[j] = j < p ? s_t[j] : ~0ull; mt.less[j] = (int32_t)s_less[j]; mt.eq[j] = (int32_t)s_eq[j]; }
      mt.p = p;
      mt.smax = -INFINITY;
      const int fb = worst > kMaxBucketKeys ? 1 : 0;
      mt.fallback = fb;
      col_open[c] = fb ? 0 : total;
      if (fb) fb_list[atomicAdd(fb_count, 1)] = col0 + c;
    }
    __syncthreads();
  }
}

// column bases (exclusive scan of the open-interval totals) and the CSR pointer array of all segments: PS per column
__global__ void __launch_bounds__(1024)
rank_part_scan_kernel(const PartMeta* __restrict__ meta, const int32_t* __restrict__ col_open, int32_t n, int PS,
                      int32_t* __restrict__ seg_ptr) {
  __shared__ int32_t s_sum[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int c0 = tid * per, c1 = (c0 + per < n) ? c0 + per : n;
  int32_t sum = 0;
  for (int c = c0; c < c1; ++c) sum += col_open[c];
  s_sum[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int32_t v = (tid >= off) ? s_sum[tid - off] : 0;
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  int32_t run = s_sum[tid] - sum;
  for (int c = c0; c < c1; ++c) {
    const PartMeta& mt = meta[c];
    for (int a = 0; a < PS; ++a) {
      seg_ptr[(int64_t)c * PS + a] = run;
      if (!mt.fallback && a <= mt.p) run += mt.open[a];
    }
  }
  if (tid == 1023) seg_ptr[(int64_t)n * PS] = s_sum[1023];
}

__device__ __forceinline__ double part_power(double r, double power, int pow_q4) {
  return power == 1.0 ? r : (pow_q4 > 0 ? pow_quarters(r, pow_q4) : PH_POW(r, power));
}

// single-value classes and NaN go straight to R; the keys of the open intervals to their segments (value + row)
__global__ void __launch_bounds__(1024)
rank_part_scatter_kernel(const double* __restrict__ X, int64_t ldx, int32_t g_dense, const int32_t* __restrict__ Xp, int32_t n,
                         int ties, int is_signed, double power, int pow_q4, PartMeta* __restrict__ meta,
                         const int32_t* __restrict__ seg_ptr, int PS, double* __restrict__ Vs, int32_t* __restrict__ Is,
                         double* __restrict__ R, int64_t ldr) {
  __shared__ uint32_t s_cur[kPartMax + 1];
  __shared__ double s_max[16];
  __shared__ PartMeta mt;                      // this column's splitters and counts (the global copy is written below)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    if (meta[c].fallback) continue;            // (uniform: the whole column is ranked by the network kernel)
    if (tid < (int)(sizeof(PartMeta) / 4)) reinterpret_cast<uint32_t*>(&mt)[tid] = reinterpret_cast<const uint32_t*>(&meta[c])[tid];
    if (tid <= kPartMax) s_cur[tid] = 0;
    __syncthreads();
    const int p = mt.p;
    const double* xc = Xp != nullptr ? X + Xp[c] : X + (int64_t)c * ldx;
    double* rc = Xp != nullptr ? R + Xp[c] : R + (int64_t)c * ldr;
    const int32_t g = Xp != nullptr ? Xp[c + 1] - Xp[c] : g_dense;
    const int32_t* sp = seg_ptr + (int64_t)c * PS;
    double vmax = Xp != nullptr ? 0.0 : -INFINITY;   // (sparse ranks: the implicit zeros, as in the bucket kernel)
    for (int i0 = 0; i0 < g; i0 += 1024) {
      const int i = i0 + tid;
      const bool in = i < g;
      const double x0 = in ? xc[i] : 0.0;
      const uint64_t k = part_key(x0, is_signed);
      int a = 0;
      bool e = false;
#pragma unroll
      for (int j = 0; j < kPartMax; ++j)
        if (j < p) {
          const uint64_t t = mt.t[j];
          a += (t < k) ? 1 : 0;
          e = e || (t == k);
        }
      const bool isnan_ = k == ~0ull;
      if (in && isnan_) rc[i] = __longlong_as_double(0x7ff8000000000000ll);
      if (in && !isnan_ && e) {
        const uint32_t lb = (uint32_t)mt.less[a], ub = lb + (uint32_t)mt.eq[a];
        double r = part_power(rank_from_bounds(lb, ub, ties), power, pow_q4);
        vmax = r > vmax ? r : vmax;
        if (is_signed) r *= sign_of(x0);
        rc[i] = r;
      }
      const bool open = in && !isnan_ && !e;
      for (int cls = 0; cls <= p; ++cls) {
        const unsigned long long m = __ballot(open && a == cls);
        if (m != 0ull) {
          const int leader = __builtin_ctzll(m);
          uint32_t base = 0;
          if (lane == leader) base = atomicAdd(&s_cur[cls], (uint32_t)__popcll(m));
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
          if (open && a == cls) {
            const int64_t q = (int64_t)sp[cls] + base +
                              __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            Vs[q] = x0;
            Is[q] = i;
          }
        }
      }
    }
    vmax = wave_max_f64_dpp(vmax);
    if (lane == 63) s_max[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      double v = s_max[0];
      for (int w = 1; w < 16; ++w) v = s_max[w] > v ? s_max[w] : v;
      meta[c].smax = v;
    }
    __syncthreads();
  }
}

// rank inside the segment + keys in the classes below = rank in the column; back to the rows
__global__ void __launch_bounds__(1024)
rank_part_finish_kernel(int32_t n, const int32_t* __restrict__ Xp, int is_signed, double power, int pow_q4,
                        const PartMeta* __restrict__ meta, const int32_t* __restrict__ seg_ptr, int PS,
                        const double* __restrict__ Rseg, const int32_t* __restrict__ Is, double* __restrict__ R, int64_t ldr,
                        double* __restrict__ colmax) {
  __shared__ double s_max[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const PartMeta& mt = meta[c];
    if (mt.fallback) continue;
    double* rc = Xp != nullptr ? R + Xp[c] : R + (int64_t)c * ldr;
    const int32_t* sp = seg_ptr + (int64_t)c * PS;
    double vmax = -INFINITY;
    for (int a = 0; a <= mt.p; ++a) {
      const double off = (a == 0) ? 0.0 : (double)(mt.less[a - 1] + mt.eq[a - 1]);
      const int q1 = sp[a + 1];
      for (int q = sp[a] + tid; q < q1; q += 1024) {
        const double rs = Rseg[q];
        double r;
        if (is_signed) {
          const double mag = (rs != 0.0) ? part_power(fabs(rs) + off, power, pow_q4) : 0.0;
          vmax = mag > vmax ? mag : vmax;
          r = rs < 0.0 ? -mag : mag;
        } else {
          r = part_power(rs + off, power, pow_q4);
          vmax = r > vmax ? r : vmax;
        }
        rc[Is[q]] = r;
      }
    }
    if (colmax != nullptr) {
      vmax = wave_max_f64_dpp(vmax);
      if (lane == 63) s_max[wave] = vmax;
      __syncthreads();
      if (tid == 0) {
        double v = mt.smax;
        for (int w = 0; w < 16; ++w) v = s_max[w] > v ? s_max[w] : v;
        colmax[c] = v;
      }
      __syncthreads();
    }
  }
}

// X dense (Xp == nullptr: n columns of g rows, leading dimensions ldx / ldr) or the stored values of CSC columns (Xp: n + 1
// device pointers, g = the longest column; the ranks go to R + Xp[c])
static int launch_colranks_partitioned(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t g, const int32_t* Xp, int32_t n,
                                       int ties, int is_signed, double power, double* R, int64_t ldr, double* colmax) {
  const int K = (g + kPartTarget - 1) / kPartTarget;
  const int PS = K + 1;                       // segments per column in the pointer array (the last one stays empty)
  const double q4 = power * 4.0;
  const int pow_q4 = (power != 1.0 && q4 >= 1.0 && q4 <= 16.0 && q4 == (double)(int)q4) ? (int)q4 : 0;
  // scratch per column: segment values (8 g) + rows (4 g) + segment ranks (8 g) + metadata; panels of <= 2 GiB
  const size_t per_col = (size_t)g * 20 + sizeof(PartMeta) + (size_t)PS * 4 + 8;
  int64_t panel = (int64_t)(((size_t)2 << 30) / per_col);
  panel = panel < 1 ? 1 : (panel > n ? n : panel);
  const size_t a8 = ((size_t)panel * g * 8 + 255) & ~(size_t)255, a4 = ((size_t)panel * g * 4 + 255) & ~(size_t)255;
  const size_t am = ((size_t)panel * sizeof(PartMeta) + 255) & ~(size_t)255;
  const size_t asp = (((size_t)panel * PS + 1) * 4 + 255) & ~(size_t)255, aco = ((size_t)panel * 4 + 255) & ~(size_t)255;
  const size_t afb = ((size_t)n * 4 + 16 + 255) & ~(size_t)255;
  const size_t need = 2 * a8 + a4 + am + asp + aco + afb;
  if (ctx->rank_scratch_bytes < need) {
    PH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->rank_scratch) PH_HIP(hipFree(ctx->rank_scratch));
    ctx->rank_scratch = nullptr;
    ctx->rank_scratch_bytes = 0;
    PH_HIP(hipMalloc(&ctx->rank_scratch, need));
    ctx->rank_scratch_bytes = need;
  }
  char* base = static_cast<char*>(ctx->rank_scratch);
  double* Vs = reinterpret_cast<double*>(base);
  double* Rseg = reinterpret_cast<double*>(base + a8);
  int32_t* Is = reinterpret_cast<int32_t*>(base + 2 * a8);
  PartMeta* meta = reinterpret_cast<PartMeta*>(base + 2 * a8 + a4);
  int32_t* seg_ptr = reinterpret_cast<int32_t*>(base + 2 * a8 + a4 + am);
  int32_t* col_open = reinterpret_cast<int32_t*>(base + 2 * a8 + a4 + am + asp);
  int32_t* fb_count = reinterpret_cast<int32_t*>(base + 2 * a8 + a4 + am + asp + aco);
  int32_t* fb_list = fb_count + 4;
  PH_HIP(hipMemsetAsync(fb_count, 0, 16, ctx->stream));
  for (int64_t c0 = 0; c0 < n; c0 += panel) {
    const int32_t nc = (int32_t)((n - c0) < panel ? (n - c0) : panel);
    const double* Xc = Xp != nullptr ? X : X + c0 * ldx;        // (CSC: the pointers are absolute offsets into @x and R)
    double* Rc = Xp != nullptr ? R : R + c0 * ldr;
    const int32_t* Xpc = Xp != nullptr ? Xp + c0 : nullptr;
    const int grid = nc < 2 * ctx->num_cu ? nc : 2 * ctx->num_cu;
    hipLaunchKernelGGL(rank_part_count_kernel, dim3(grid), dim3(1024), 0, ctx->stream, Xc, ldx, g, Xpc, nc, is_signed, K, meta,
                       col_open, fb_count, fb_list, (int32_t)c0);
    hipLaunchKernelGGL(rank_part_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, meta, col_open, nc, PS, seg_ptr);
    hipLaunchKernelGGL(rank_part_scatter_kernel, dim3(grid), dim3(1024), 0, ctx->stream, Xc, ldx, g, Xpc, nc, ties, is_signed,
                       power, pow_q4, meta, seg_ptr, PS, Vs, Is, Rc, ldr);
    PH_HIP(hipGetLastError());
    // every segment is a CSC column of at most kMaxBucketKeys values: the bucket ranker (its own clustered-values fallback
    // included); signed ranks come back with their sign
    const int rc = launch_ranks(ctx, Vs, 0, 0, seg_ptr, nc * PS, kMaxBucketKeys, ties, is_signed, 1.0, Rseg, 0, nullptr);
    if (rc != PLAIDHIP_OK) return rc;
    hipLaunchKernelGGL(rank_part_finish_kernel, dim3(grid), dim3(1024), 0, ctx->stream, nc, Xpc, is_signed, power, pow_q4, meta,
                       seg_ptr, PS, Rseg, Is, Rc, ldr, colmax != nullptr ? colmax + c0 : nullptr);
    PH_HIP(hipGetLastError());
  }
  // columns with an over-full open interval (heavy clusters the sample did not isolate): the sorting network on a global
  // scratch, as before; a persistent grid reads the device-side list
  const int fb_grid = n < ctx->num_cu ? n : ctx->num_cu;
  const int64_t stride = ((int64_t)g + 1) & ~1ll;
  const int rcw = ensure_workspace(ctx, (size_t)(n < 2 * ctx->num_cu ? n : 2 * ctx->num_cu) * (size_t)stride * 8);
  if (rcw != PLAIDHIP_OK) return rcw;
  return launch_network(ctx, X, ldx, Xp != nullptr ? 0 : g, Xp, n, g, ties, is_signed, power, R, ldr, colmax, nullptr, nullptr,
                        fb_grid, 0, fb_list, fb_count);
}


// ---- ties.method = "first" / "last" / "dense" (matrixStats::colRanks and base::rank take them: R/plaid.R:593,614-617,
// 639-642 forward any ties.method) -------------------------------------------------------------------------------------
// Composed from the min-rank kernels, which stay the only ranking code: with lb_i = #{x_j < x_i} (min rank - 1, the same
// for every member of a tie group and different between groups) the pairs (lb_i, i) are all distinct, and
//     first_i = 1 + #{(lb_j, j) < (lb_i, i)}        -- the min rank of the tie-free column  y_i = lb_i * 2^26 + i
//     last_i  = the same with  y_i = lb_i * 2^26 + (cnt - 1 - i)     (rank(c(1,1,1), ties = "last") is 3 2 1)
//     dense_i = 1 + #{tie groups below x_i}: the first member of a group is its leader (first_i == lb_i + 1); the min rank
//               of the column that holds lb at the leaders and NaN elsewhere IS the dense rank at the leaders (a NaN gets
//               no rank and disturbs nobody's); the leaders leave it at slot lb of a scratch column, every member reads
//               its group's slot.
// y is exact in a double (lb, i < 2^26).  NaN inputs stay NaN through every pass.  Signed: the ranks of |x|, signed at the
// end.  These methods are off the hot path (plaid's own callers use "average" and "min" only): 2 - 3 passes of the fast
// kernels plus element-wise kernels, any column length the rank kernels take.
constexpr double kTieShift = 67108864.0;   // 2^26

struct TieCols {   // columns of a dense matrix (Xp == nullptr) or the stored values of CSC columns
  const int32_t* Xp;
  int32_t g, n;
  int64_t ldx, ldr, lds;   // leading dimensions of X, R and of the scratch columns
};

template <typename F>
__device__ __forceinline__ void tie_for_each(const TieCols& t, F f) {
  for (int c = blockIdx.y; c < t.n; c += gridDim.y) {
    int64_t xb, rb, sb;
    int32_t cnt;
    if (t.Xp != nullptr) { xb = rb = sb = t.Xp[c]; cnt = t.Xp[c + 1] - t.Xp[c]; }
    else { xb = (int64_t)c * t.ldx; rb = (int64_t)c * t.ldr; sb = (int64_t)c * t.lds; cnt = t.g; }
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) f(xb + i, rb + i, sb, i, cnt);
  }
}

// A = lb (NaN for a NaN input), B = the tie-free column y.  R holds sign * min rank of |x| (signed) or the min rank.
__global__ void __launch_bounds__(256)
tie_prep_kernel(TieCols t, const double* __restrict__ X, const double* __restrict__ R, int is_signed, int last,
                double* __restrict__ A, double* __restrict__ B) {
  tie_for_each(t, [&](int64_t xi, int64_t ri, int64_t sb, int32_t i, int32_t cnt) {
    const double x = X[xi], r = R[ri];
    // signed ranks are 0 at x == 0, whose |x| is the smallest value of the column: lb = 0
    const double lb = (r != r) ? r : ((is_signed && x == 0.0) ? 0.0 : fabs(r) - 1.0);
    A[sb + i] = lb;
    B[sb + i] = lb * kTieShift + (double)(last ? cnt - 1 - i : i);
  });
}

// first / last: R holds the min rank of y = the wanted rank of |x| (or x); signed: put the sign on
__global__ void __launch_bounds__(256)
tie_sign_kernel(TieCols t, const double* __restrict__ X, double* __restrict__ R) {
  tie_for_each(t, [&](int64_t xi, int64_t ri, int64_t, int32_t, int32_t) {
    const double x = X[xi], r = R[ri];
    R[ri] = (x == 0.0) ? 0.0 : ((x < 0.0) ? -r : r);   // (NaN: neither branch, r is NaN already)
  });
}

// dense, step 1: R holds first ranks, A the lower bounds: B = lb at the leaders, NaN elsewhere
__global__ void __launch_bounds__(256)
tie_leader_kernel(TieCols t, const double* __restrict__ R, const double* __restrict__ A, double* __restrict__ B) {
  tie_for_each(t, [&](int64_t, int64_t ri, int64_t sb, int32_t i, int32_t) {
    const double lb = A[sb + i];
    B[sb + i] = (R[ri] == lb + 1.0) ? lb : __longlong_as_double(0x7ff8000000000000ll);
  });
}
// step 2: R holds the dense rank at the leaders (NaN elsewhere): leave it at slot lb of the scratch column
__global__ void __launch_bounds__(256)
tie_scatter_kernel(TieCols t, const double* __restrict__ R, const double* __restrict__ A, double* __restrict__ B) {
  tie_for_each(t, [&](int64_t, int64_t ri, int64_t sb, int32_t i, int32_t) {
    const double d = R[ri];
    if (d == d) B[sb + (int64_t)A[sb + i]] = d;
  });
}
// step 3: every member reads its group's slot
__global__ void __launch_bounds__(256)
tie_gather_kernel(TieCols t, const double* __restrict__ X, const double* __restrict__ A, const double* __restrict__ B,
                  int is_signed, double* __restrict__ R) {
  tie_for_each(t, [&](int64_t xi, int64_t ri, int64_t sb, int32_t i, int32_t) {
    const double lb = A[sb + i], x = X[xi];
    double r = (lb != lb) ? lb : B[sb + (int64_t)lb];
    if (is_signed) r = (x == 0.0) ? 0.0 : ((x < 0.0) ? -r : r);
    R[ri] = r;
  });
}

// `total`: number of values (dense: unused; CSC: Xp[n], which the caller reads back -- these methods are not stream-ordered)
static int launch_colranks_composed(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t g, const int32_t* Xp, int32_t n,
                                    int32_t max_len, int64_t total, int ties, int is_signed, double* R, int64_t ldr) {
  if (n == 0 || max_len == 0) return PLAIDHIP_OK;
  const int64_t lds = ((int64_t)g + 1) & ~1ll;
  const size_t count = Xp != nullptr ? (size_t)total : (size_t)lds * n;
  // two scratch columns per column (grown on demand, kept with the context)
  if (ctx->tie_scratch_bytes < 2 * count * 8) {
    PH_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->tie_scratch) PH_HIP(hipFree(ctx->tie_scratch));
    ctx->tie_scratch = nullptr;
    ctx->tie_scratch_bytes = 0;
    PH_HIP(hipMalloc(&ctx->tie_scratch, 2 * count * 8));
    ctx->tie_scratch_bytes = 2 * count * 8;
  }
  double* A = static_cast<double*>(ctx->tie_scratch);
  double* B = A + count;
  auto ranks_min = [&](const double* V, int64_t ldv, int sgn, double* Out, int64_t ldo) -> int {
    if (Xp != nullptr) return launch_colranks_csc_f64(ctx, Xp, V, n, max_len, PLAIDHIP_TIES_MIN, sgn, 1.0, Out, nullptr);
    return launch_colranks_dense_f64(ctx, V, ldv, g, n, PLAIDHIP_TIES_MIN, sgn, 1.0, Out, ldo, nullptr);
  };
  TieCols t{Xp, g, n, ldx, ldr, lds};
  const dim3 grid((unsigned)std::min<int64_t>(((int64_t)max_len + 255) / 256, 64), (unsigned)std::min(n, 16384));
  int rc = ranks_min(X, ldx, is_signed, R, ldr);                                  // lb + 1 (with the sign when signed)
  if (rc != PLAIDHIP_OK) return rc;
  hipLaunchKernelGGL(tie_prep_kernel, grid, dim3(256), 0, ctx->stream, t, X, R, is_signed, ties == PLAIDHIP_TIES_LAST ? 1 : 0, A, B);
  rc = ranks_min(B, lds, 0, R, ldr);                                              // first (or last) ranks
  if (rc != PLAIDHIP_OK) return rc;
  if (ties != PLAIDHIP_TIES_DENSE) {
    if (is_signed) hipLaunchKernelGGL(tie_sign_kernel, grid, dim3(256), 0, ctx->stream, t, X, R);
  } else {
    hipLaunchKernelGGL(tie_leader_kernel, grid, dim3(256), 0, ctx->stream, t, R, A, B);
    rc = ranks_min(B, lds, 0, R, ldr);                                            // dense ranks at the leaders
    if (rc != PLAIDHIP_OK) return rc;
    hipLaunchKernelGGL(tie_scatter_kernel, grid, dim3(256), 0, ctx->stream, t, R, A, B);
    hipLaunchKernelGGL(tie_gather_kernel, grid, dim3(256), 0, ctx->stream, t, X, A, B, is_signed, R);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_colranks_dense_f64(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t g, int32_t n,
                              int ties, int is_signed, double power, double* R, int64_t ldr,
                              double* colmax) {
  if (ties >= PLAIDHIP_TIES_FIRST)   // "first" / "last" / "dense": composed from min-rank passes (power 1, no column maximum)
    return launch_colranks_composed(ctx, X, ldx, g, nullptr, n, g, 0, ties, is_signed, R, ldr);
  // columns beyond the bucket ranker's LDS: cut by value into segments it takes (above), unless the context pins a kernel
  if (g > kMaxBucketKeys && (g + kPartTarget - 1) / kPartTarget <= kPartMax && ctx->opt_rank_kernel != 1)
    return launch_colranks_partitioned(ctx, X, ldx, g, nullptr, n, ties, is_signed, power, R, ldr, colmax);
  return launch_ranks(ctx, X, ldx, g, nullptr, n, g, ties, is_signed, power, R, ldr, colmax);
}

int launch_colranks_csc_dense_f64(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                                  int32_t g, int32_t n, int ties, int is_signed, double power, double* R,
                                  int64_t ldr, double* colmax) {
  return launch_ranks(ctx, Xx, 0, g, Xp, n, g, ties, is_signed, power, R, ldr, colmax, Xi);
}

// ---- dense ranks of a sparse column WITHOUT densifying it --------------------------------------------------------------
// colranks(X sparse, keep.zero = FALSE) (R/plaid.R:602-609) ranks the zeros too and returns a dense matrix.  All implicit
// zeros tie, so the dense ranks follow from the ranks among the STORED values (nnz per column, the fast CSC path) and
// three counts: with z implicit zeros, e0 stored zeros and `neg` stored negatives,
//     stored v > 0 : rank_nz + z      stored v < 0 : rank_nz      stored v == 0 : rank_nz + z * {min 0, average 1/2, max 1}
//     implicit zero: lb = neg, ub = neg + e0 + z  ->  lb + 1 | (lb + 1 + ub) / 2 | ub
// (signed: rank(|x|) puts every stored non-zero above the z zeros, the zeros themselves give sign(0) * rank = 0).
// The kernel writes the zero rank over the whole column and then the stored entries' ranks over their rows: O(nnz) work
// and one dense write, for ANY number of rows -- densify-and-rank is a 20,000-key sort per column and leaves the fast
// rank kernel beyond 20,352 rows (a 10x Genomics matrix has 33,538 or 36,601).
__global__ void __launch_bounds__(256)
expand_sparse_ranks_kernel(const int32_t* __restrict__ Xp, const int32_t* __restrict__ Xi, const double* __restrict__ Xx,
                           const double* __restrict__ Rx, int32_t g, int32_t n, int ties, int is_signed, double power,
                           int pow_q4, double* __restrict__ R, int64_t ldr, double* __restrict__ colmax) {
  // the power as the bucket kernel applies it: by square roots when 4 * power is a small integer, pow() otherwise
  auto powr = [&](double r) { return pow_q4 > 0 ? pow_quarters(r, pow_q4) : PH_POW(r, power); };
  __shared__ uint32_t s_cnt[2];
  __shared__ double s_max[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  for (int c = blockIdx.x; c < n; c += gridDim.x) {
    const int q0 = Xp[c], q1 = Xp[c + 1];
    const double z = (double)(g - (q1 - q0));
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    uint32_t neg = 0, e0 = 0;
    for (int q = q0 + tid; q < q1; q += 256) {
      const double v = Xx[q];
      neg += (v < 0.0) ? 1u : 0u;
      e0 += (v == 0.0) ? 1u : 0u;
    }
    neg = wave_incl_scan_u32(neg);
    e0 = wave_incl_scan_u32(e0);
    if (lane == 63) { atomicAdd(&s_cnt[0], neg); atomicAdd(&s_cnt[1], e0); }
    __syncthreads();
    const double lb0 = (double)s_cnt[0], ub0 = lb0 + (double)s_cnt[1] + z;
    double r0 = (ties == PLAIDHIP_TIES_MIN) ? lb0 + 1.0 : ((ties == PLAIDHIP_TIES_MAX) ? ub0 : 0.5 * (lb0 + 1.0 + ub0));
    if (is_signed) r0 = 0.0;
    else if (power != 1.0) r0 = powr(r0);
    const double t0 = (ties == PLAIDHIP_TIES_MIN) ? 0.0 : ((ties == PLAIDHIP_TIES_MAX) ? 1.0 : 0.5);
    double* rc = R + (int64_t)c * ldr;
    for (int i = tid; i < g; i += 256) __builtin_nontemporal_store(r0, rc + i);
    __syncthreads();   // the column is filled (this workgroup's stores are performed) before single rows are overwritten
    double vmax = (z > 0.0 && !is_signed) ? r0 : ((z > 0.0) ? 0.0 : -INFINITY);
    for (int q = q0 + tid; q < q1; q += 256) {
      const double v = Xx[q], rn = Rx[q];
      double r;
      if (is_signed) {
        const double mag = fabs(rn) + z;                    // rank of |v| > 0 among all rows
        double pm = (power != 1.0) ? powr(mag) : mag;
        r = (v == 0.0) ? 0.0 : ((v < 0.0) ? -pm : pm);
        if (v != v) { r = nan; pm = -INFINITY; }
        if (v == 0.0) pm = 0.0;
        vmax = pm > vmax ? pm : vmax;
      } else {
        r = rn + z * ((v > 0.0) ? 1.0 : ((v == 0.0) ? t0 : 0.0));
        if (power != 1.0) r = powr(r);
        if (v != v) r = nan;
        vmax = (r > vmax) ? r : vmax;                       // (false for a NaN)
      }
      rc[Xi[q]] = r;
    }
    if (colmax != nullptr) {
      vmax = wave_max_f64_dpp(vmax);
      if (lane == 63) s_max[wave] = vmax;
      __syncthreads();
      if (tid == 0) {
        double v = s_max[0];
        for (int w = 1; w < 4; ++w) v = s_max[w] > v ? s_max[w] : v;
        colmax[c] = v;
      }
    }
    __syncthreads();
  }
}

// dense ranks from CSC through the ranks of the stored values (Rx_scratch: Xp[n] doubles); every column must have at most
// kMaxBucketKeys stored values (the caller states the longest, as for launch_colranks_csc_f64)
int launch_colranks_csc_dense_nz_f64(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t g,
                                     int32_t n, int32_t max_col_nnz, int ties, int is_signed, double power,
                                     double* Rx_scratch, double* R, int64_t ldr, double* colmax) {
  if (n == 0 || g == 0) return PLAIDHIP_OK;
  const int rc = launch_ranks(ctx, Xx, 0, 0, Xp, n, max_col_nnz, ties, is_signed, 1.0, Rx_scratch, 0, nullptr);
  if (rc != PLAIDHIP_OK) return rc;
  const int cap = ctx->num_cu * 8;
  const double q4 = power * 4.0;
  const int pow_q4 = (power != 1.0 && q4 >= 1.0 && q4 <= 16.0 && q4 == (double)(int)q4) ? (int)q4 : 0;
  hipLaunchKernelGGL(expand_sparse_ranks_kernel, dim3(n < cap ? n : cap), dim3(256), 0, ctx->stream, Xp, Xi, Xx, Rx_scratch,
                     g, n, ties, is_signed, power, pow_q4, R, ldr, colmax);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}
int max_sparse_rank_column() { return kMaxBucketKeys; }

// stream-ordered: the caller states the longest column (include/plaidhip.h), nothing is read back
int launch_colranks_csc_f64(plaidhip_ctx* ctx, const int32_t* Xp, const double* Xx, int32_t n, int32_t max_col_nnz,
                            int ties, int is_signed, double power, double* Rx, double* colmax) {
  if (ties >= PLAIDHIP_TIES_FIRST && n > 0) {   // "first" / "last": composed; needs nnz(X) on the host (one read-back)
    int32_t ends[1] = {0};
    int32_t first[1] = {0};
    PH_HIP(hipMemcpyAsync(ends, Xp + n, 4, hipMemcpyDeviceToHost, ctx->stream));
    PH_HIP(hipMemcpyAsync(first, Xp, 4, hipMemcpyDeviceToHost, ctx->stream));
    PH_HIP(hipStreamSynchronize(ctx->stream));
    if (first[0] != 0) { set_error("colranks_csc: ties.method first / last need Xp[0] == 0"); return PLAIDHIP_EINVAL; }
    return launch_colranks_composed(ctx, Xx, 0, max_col_nnz, Xp, n, max_col_nnz, ends[0], ties, is_signed, Rx, 0);
  }
  // columns with more stored values than the bucket ranker's LDS holds: cut by value (launch_colranks_partitioned)
  if (n > 0 && max_col_nnz > kMaxBucketKeys && (max_col_nnz + kPartTarget - 1) / kPartTarget <= kPartMax && ctx->opt_rank_kernel != 1)
    return launch_colranks_partitioned(ctx, Xx, 0, max_col_nnz, Xp, n, ties, is_signed, power, Rx, 0, colmax);
  return launch_ranks(ctx, Xx, 0, 0, Xp, n, max_col_nnz, ties, is_signed, power, Rx, 0, colmax);
}

}  // namespace plaidhip
