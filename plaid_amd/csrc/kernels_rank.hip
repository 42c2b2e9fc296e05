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
  const bool use_bucket = can_bucket && (ctx->opt_rank_kernel == 2 || (ctx->opt_rank_kernel == 0 && max_len > 256));
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
  // 512 threads x 40 keys for a 20k-gene column: 1,024 threads would cap the kernel at 128 registers, short of
  // the 120 a thread needs for its keys and their state alone
  if (max_len <= 2048) rc = launch_bucket<256, 8>(ctx, a, max_len, grid_cap);
  else if (max_len <= 4096) rc = launch_bucket<256, 16>(ctx, a, max_len, grid_cap);
  else if (max_len <= 8192) rc = launch_bucket<512, 16>(ctx, a, max_len, grid_cap);
  else if (max_len <= 12288) rc = launch_bucket<512, 24>(ctx, a, max_len, grid_cap);
  else rc = launch_bucket<512, 40>(ctx, a, max_len, grid_cap);
  if (rc != PLAIDHIP_OK) return rc;
  // columns the bucket kernel gave up on (clustered values): a small persistent grid reads the device-side list
  const int fb_grid = grid_cap < ctx->num_cu ? grid_cap : ctx->num_cu;
  return launch_network(ctx, Xv, ldx, g_dense, Xp, n, max_len, ties, is_signed, power, R, ldr, colmax, Xi_dense,
                        dscratch, fb_grid, ws_off, fb_list, fb_count);
}

int launch_colranks_dense_f64(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t g, int32_t n,
                              int ties, int is_signed, double power, double* R, int64_t ldr,
                              double* colmax) {
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
  return launch_ranks(ctx, Xx, 0, 0, Xp, n, max_col_nnz, ties, is_signed, power, Rx, 0, colmax);
}

}  // namespace plaidhip
