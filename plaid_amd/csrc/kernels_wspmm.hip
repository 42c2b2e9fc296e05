// t(x) %*% y for a sparse x with ARBITRARY stored values: chunked_crossprod() (R/plaid.R:100-123) is written for
// any two matrices; plaid() only ever hands it the (column-scaled) 0/1 membership matrix (:73-80), which is what the
// scheduled kernels of kernels_spmm.hip are built for.  This kernel is the general case behind the same entry: x stays
// in its dgCMatrix slots (@p, @i, @x: no host-side plan), one sample column of y is resident per workgroup, and a
// 16-lane group walks one column of x at a time:
//     S[j, c] = sum over the stored entries k of x[, j] of  x@x[k] * y[x@i[k], c]
//   * y dense: the column is copied into LDS (nrow <= 20,480) or gathered straight from global memory (L2) above that;
//   * y a dgCMatrix: the column's stored values are scattered into a zeroed dense column (LDS, or the workgroup's slice
//     of a global scratch) and taken out again after the walk, so the zeroing is paid once per workgroup.
// Stored zeros of x are multiplied like any other value (0 * NaN is NaN, as in Matrix::crossprod).  Per column the
// workgroup reads x once from L2: nnz(x) * 12 bytes -- an order of magnitude more fabric traffic per score than the
// scheduled membership kernels (1.5 bytes per membership), which is why those stay the plaid() path.
#include "common.h"

namespace plaidhip {
namespace {

constexpr int kWBlock = 1024;
constexpr int kWGroups = kWBlock / 16;

struct WeightedArgs {
  const int32_t* Wp;
  const int32_t* Wi;
  const double* Wx;
  int32_t g, m, n;
  const double* Y;      // dense y (ldy) ...
  int64_t ldy;
  const int32_t* Yp;    // ... or its CSC slots
  const int32_t* Yi;
  const double* Yx;
  double* S;
  int64_t lds;
  double* scratch;      // !IN_LDS && Y_CSC: gridDim.x dense columns of gpad doubles, zeroed by the kernel
  int64_t gpad;
};

template <bool IN_LDS, bool Y_CSC>
__global__ void __launch_bounds__(kWBlock)
crossprod_weighted_kernel(WeightedArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* lcol = reinterpret_cast<double*>(smem_raw);
  double* gcol = (!IN_LDS && Y_CSC) ? a.scratch + (int64_t)blockIdx.x * a.gpad : nullptr;
  const int tid = threadIdx.x;
  const int sub = tid & 15;
  const int grp = tid >> 4;

  if constexpr (Y_CSC) {
    // the dense column the stored values are scattered into: zero it once
    if constexpr (IN_LDS) {
      for (int i = tid; i < a.g; i += kWBlock) lcol[i] = 0.0;
    } else {
      for (int i = tid; i < a.g; i += kWBlock) gcol[i] = 0.0;
    }
    __syncthreads();
  }

  for (int c = blockIdx.x; c < a.n; c += gridDim.x) {
    const double* ycol = nullptr;   // !IN_LDS: the column in global memory
    int q0 = 0, q1 = 0;
    if constexpr (Y_CSC) {
      q0 = a.Yp[c];
      q1 = a.Yp[c + 1];
      for (int q = q0 + tid; q < q1; q += kWBlock) {
        const int i = a.Yi[q];
        const double v = a.Yx[q];
        if constexpr (IN_LDS) lcol[i] = v; else gcol[i] = v;
      }
      ycol = gcol;
    } else {
      ycol = a.Y + (int64_t)c * a.ldy;
      if constexpr (IN_LDS) {
        for (int i = tid; i < a.g; i += kWBlock) lcol[i] = __builtin_nontemporal_load(ycol + i);
      }
    }
    if constexpr (IN_LDS || Y_CSC) __syncthreads();

    double* scol = a.S + (int64_t)c * a.lds;
    for (int j = grp; j < a.m; j += kWGroups) {
      const int k0 = a.Wp[j], k1 = a.Wp[j + 1];
      double acc0 = 0.0, acc1 = 0.0;
      int k = k0 + sub;
      for (; k + 16 < k1; k += 32) {   // two independent chains
        const int i0 = a.Wi[k], i1 = a.Wi[k + 16];
        const double w0 = a.Wx[k], w1 = a.Wx[k + 16];
        const double y0 = IN_LDS ? lcol[i0] : ycol[i0];
        const double y1 = IN_LDS ? lcol[i1] : ycol[i1];
        acc0 = __builtin_fma(w0, y0, acc0);
        acc1 = __builtin_fma(w1, y1, acc1);
      }
      if (k < k1) {
        const int i0 = a.Wi[k];
        const double w0 = a.Wx[k];
        const double y0 = IN_LDS ? lcol[i0] : ycol[i0];
        acc0 = __builtin_fma(w0, y0, acc0);
      }
      double acc = acc0 + acc1;
      acc += __shfl_xor(acc, 8, 16);
      acc += __shfl_xor(acc, 4, 16);
      acc += __shfl_xor(acc, 2, 16);
      acc += __shfl_xor(acc, 1, 16);
      if (sub == 0) __builtin_nontemporal_store(acc, scol + j);
    }

    if constexpr (IN_LDS || Y_CSC) __syncthreads();
    if constexpr (Y_CSC) {
      // take the column's values out again: the dense column is all zeros for the next one
      for (int q = q0 + tid; q < q1; q += kWBlock) {
        const int i = a.Yi[q];
        if constexpr (IN_LDS) lcol[i] = 0.0; else gcol[i] = 0.0;
      }
      __syncthreads();
    }
  }
}

template <bool IN_LDS, bool Y_CSC>
int launch_one(plaidhip_ctx* ctx, const WeightedArgs& a, int grid) {
  const size_t smem = IN_LDS ? (size_t)a.g * sizeof(double) : 0;
  if (IN_LDS) PH_FULL_LDS(ctx, (&crossprod_weighted_kernel<IN_LDS, Y_CSC>));
  hipLaunchKernelGGL((crossprod_weighted_kernel<IN_LDS, Y_CSC>), dim3(grid), dim3(kWBlock), smem, ctx->stream, a);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

}  // namespace

int launch_crossprod_weighted_f64(plaidhip_ctx* ctx, const int32_t* Wp, const int32_t* Wi, const double* Wx, int32_t g,
                                  int32_t m, const double* Y, int64_t ldy, const int32_t* Yp, const int32_t* Yi,
                                  const double* Yx, int32_t n, double* S, int64_t lds) {
  ctx->fmed.valid = false;
  if (n == 0 || m == 0) return PLAIDHIP_OK;
  WeightedArgs a{};
  a.Wp = Wp; a.Wi = Wi; a.Wx = Wx;
  a.g = g; a.m = m; a.n = n;
  a.Y = Y; a.ldy = ldy;
  a.Yp = Yp; a.Yi = Yi; a.Yx = Yx;
  a.S = S; a.lds = lds;
  const bool csc = Yp != nullptr;
  const bool in_lds = g <= kMaxLdsKeys;
  // 1,024-thread workgroups: two per CU unless one column needs more than half the LDS
  int grid = (in_lds && (size_t)g * sizeof(double) > (size_t)kLdsBytes / 2) ? ctx->num_cu : 2 * ctx->num_cu;
  if (grid > n) grid = n;
  if (csc && !in_lds) {
    a.gpad = ((int64_t)g + 31) & ~(int64_t)31;
    const int rc = ensure_workspace(ctx, (size_t)grid * a.gpad * sizeof(double));
    if (rc != PLAIDHIP_OK) return rc;
    a.scratch = static_cast<double*>(ctx->ws);
  }
  if (in_lds) return csc ? launch_one<true, true>(ctx, a, grid) : launch_one<true, false>(ctx, a, grid);
  return csc ? launch_one<false, true>(ctx, a, grid) : launch_one<false, false>(ctx, a, grid);
}

}  // namespace plaidhip
