// Row-wise two-group moments for plaid.test() (R/plaid.R:392-474): the statistics are reduced
// on the device so that neither X (genes x samples) nor the score matrix (sets x samples) has
// to visit the host -- only O(rows) numbers do.
//
//   rowMeans(A[, y == 1]) , rowMeans(A[, y == 0])                       (R/plaid.R:407-408, 431)
//   per-row group variances for the Welch tests Rfast::ttests(t(gsetX), ina = y + 1)   (:429)
//
// A is column-major (rows contiguous), so a thread owns a row and a workgroup walks a block of
// columns: every load is coalesced.  Columns are cut into blocks of kColBlock; a block writes its
// partial sums to [block][stat][row] and a second kernel adds the blocks in a fixed order
// (deterministic, no atomics).  Variances are two-pass (sum of squared deviations from the group
// mean), which is at least as accurate as the sum-of-squares formula of the reference.
#include "common.h"

namespace plaidhip {

constexpr int kColBlock = 128;

// pass 1: sums per group.  part: [nblk][2][rows]
__global__ void __launch_bounds__(256)
row_group_sums_kernel(const double* __restrict__ A, int64_t ld, int32_t rows, int32_t n,
                      const int32_t* __restrict__ y, double* __restrict__ part) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * kColBlock;
  const int c1 = c0 + kColBlock < n ? c0 + kColBlock : n;
  double s0 = 0.0, s1 = 0.0;
  if (r < rows) {
    for (int c = c0; c < c1; ++c) {
      const int lab = y[c];                       // wave-uniform
      const double v = A[(int64_t)c * ld + r];
      s0 += lab == 0 ? v : 0.0;
      s1 += lab == 1 ? v : 0.0;
    }
    double* p = part + (int64_t)blockIdx.y * 2 * rows;
    p[r] = s0;
    p[rows + r] = s1;
  }
}

// pass 2: sums of squared deviations from the group means.  mean: [2][rows]; part: [nblk][2][rows]
__global__ void __launch_bounds__(256)
row_group_ssd_kernel(const double* __restrict__ A, int64_t ld, int32_t rows, int32_t n,
                     const int32_t* __restrict__ y, const double* __restrict__ mean,
                     double* __restrict__ part) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * kColBlock;
  const int c1 = c0 + kColBlock < n ? c0 + kColBlock : n;
  if (r < rows) {
    const double m0 = mean[r], m1 = mean[rows + r];
    double q0 = 0.0, q1 = 0.0;
    for (int c = c0; c < c1; ++c) {
      const int lab = y[c];
      const double v = A[(int64_t)c * ld + r];
      const double d0 = v - m0, d1 = v - m1;
      q0 += lab == 0 ? d0 * d0 : 0.0;
      q1 += lab == 1 ? d1 * d1 : 0.0;
    }
    double* p = part + (int64_t)blockIdx.y * 2 * rows;
    p[r] = q0;
    p[rows + r] = q1;
  }
}

// out[s][r] = scale[s] * sum over blocks of part[b][s][r]   (blocks added in order)
__global__ void __launch_bounds__(256)
reduce_blocks_kernel(const double* __restrict__ part, int32_t rows, int32_t nblk, double scale0,
                     double scale1, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // over 2 * rows
  if (i >= 2 * rows) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * 2 * rows + i];
  out[i] = s * (i < rows ? scale0 : scale1);
}

// fc = m1 - m0 and fc^2 as the two columns of a rows x 2 matrix with leading dimension ld2
__global__ void __launch_bounds__(256)
fold_change_kernel(const double* __restrict__ mean, int32_t rows, int64_t ld2, double* __restrict__ F) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const double fc = mean[rows + r] - mean[r];
  F[r] = fc;
  F[ld2 + r] = fc * fc;
}

// replaid.gsva row transform, R/plaid.R:343: z = (x - rowMeans(X)) / (1e-8 + rowSds(X)), in place.
// mom: [2][rows] group-0 means, ssd: [2][rows] group-0 sums of squared deviations (all samples in group 0)
__global__ void __launch_bounds__(256)
row_ztransform_kernel(double* __restrict__ A, int64_t ld, int32_t rows, int32_t n,
                      const double* __restrict__ mean, const double* __restrict__ ssd) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * kColBlock;
  const int c1 = c0 + kColBlock < n ? c0 + kColBlock : n;
  if (r >= rows) return;
  const double mu = mean[r];
  const double den = 1e-8 + sqrt(ssd[r] / (double)(n - 1));   // sd with n - 1 (NaN for a single sample, as in R)
  for (int c = c0; c < c1; ++c) {
    double* p = &A[(int64_t)c * ld + r];
    *p = (*p - mu) / den;   // a true division, as the reference does: ties between genes stay ties
  }
}

int launch_row_ztransform(plaidhip_ctx* ctx, double* A, int64_t ld, int32_t rows, int32_t n, const double* d_mean,
                          const double* d_ssd) {
  if (rows == 0 || n == 0) return PLAIDHIP_OK;
  const int nblk = (n + kColBlock - 1) / kColBlock;
  hipLaunchKernelGGL(row_ztransform_kernel, dim3((rows + 255) / 256, nblk), dim3(256), 0, ctx->stream, A, ld, rows, n,
                     d_mean, d_ssd);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

// B (cols x rows, ldb) = transpose of A (rows x cols, lda), both column-major; 32 x 32 tiles through LDS
// (+1 padding), coalesced on both sides.  Used by the ECDF row transform of replaid.gsva, which needs
// per-GENE ranks across samples (R/plaid.R:346): genes become columns, the column rank kernel does the rest.
__global__ void __launch_bounds__(256)
transpose_f64_kernel(const double* __restrict__ A, int64_t lda, int32_t rows, int32_t cols,
                     double* __restrict__ B, int64_t ldb) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + tx, c = c0 + j;
    if (r < rows && c < cols) tile[j][tx] = A[(int64_t)c * lda + r];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + tx, r = r0 + j;
    if (r < rows && c < cols) B[(int64_t)r * ldb + c] = tile[tx][j];
  }
}

int launch_transpose_f64(plaidhip_ctx* ctx, const double* A, int64_t lda, int32_t rows, int32_t cols, double* B,
                         int64_t ldb) {
  if (rows == 0 || cols == 0) return PLAIDHIP_OK;
  hipLaunchKernelGGL(transpose_f64_kernel, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, ctx->stream, A, lda,
                     rows, cols, B, ldb);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

// Group means (and optionally sums of squared deviations) of every row of A.
// d_mean: [2][rows] (group 0, group 1); d_ssd: [2][rows] or null.  n0 / n1: group sizes.
// ws: scratch of at least 2 * rows * ceil(n / kColBlock) doubles.
int launch_row_group_moments(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n,
                             const int32_t* d_y, int64_t n0, int64_t n1, double* d_mean, double* d_ssd,
                             double* ws) {
  if (rows == 0) return PLAIDHIP_OK;
  const int nblk = (n + kColBlock - 1) / kColBlock;
  const dim3 grid((rows + 255) / 256, nblk > 0 ? nblk : 1);
  const int red_blocks = (2 * rows + 255) / 256;
  if (n > 0) hipLaunchKernelGGL(row_group_sums_kernel, grid, dim3(256), 0, ctx->stream, A, ld, rows, n, d_y, ws);
  hipLaunchKernelGGL(reduce_blocks_kernel, dim3(red_blocks), dim3(256), 0, ctx->stream, ws, rows, n > 0 ? nblk : 0,
                     n0 > 0 ? 1.0 / (double)n0 : __builtin_nan(""), n1 > 0 ? 1.0 / (double)n1 : __builtin_nan(""), d_mean);
  if (d_ssd != nullptr) {
    if (n > 0) hipLaunchKernelGGL(row_group_ssd_kernel, grid, dim3(256), 0, ctx->stream, A, ld, rows, n, d_y, d_mean, ws);
    hipLaunchKernelGGL(reduce_blocks_kernel, dim3(red_blocks), dim3(256), 0, ctx->stream, ws, rows, n > 0 ? nblk : 0,
                       1.0, 1.0, d_ssd);
  }
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

// pass 2 alone: sums of squared deviations from GIVEN group means (a sample-sharded caller passes the means of all shards)
int launch_row_group_ssd(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n, const int32_t* d_y,
                         const double* d_mean, double* d_ssd, double* ws) {
  if (rows == 0) return PLAIDHIP_OK;
  const int nblk = (n + kColBlock - 1) / kColBlock;
  const dim3 grid((rows + 255) / 256, nblk > 0 ? nblk : 1);
  const int red_blocks = (2 * rows + 255) / 256;
  if (n > 0) hipLaunchKernelGGL(row_group_ssd_kernel, grid, dim3(256), 0, ctx->stream, A, ld, rows, n, d_y, d_mean, ws);
  hipLaunchKernelGGL(reduce_blocks_kernel, dim3(red_blocks), dim3(256), 0, ctx->stream, ws, rows, n > 0 ? nblk : 0, 1.0, 1.0,
                     d_ssd);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int launch_fold_change(plaidhip_ctx* ctx, const double* d_mean, int32_t rows, int64_t ld2, double* d_F) {
  if (rows == 0) return PLAIDHIP_OK;
  hipLaunchKernelGGL(fold_change_kernel, dim3((rows + 255) / 256), dim3(256), 0, ctx->stream, d_mean, rows, ld2, d_F);
  PH_HIP(hipGetLastError());
  return PLAIDHIP_OK;
}

int64_t row_group_ws_doubles(int32_t rows, int32_t n) {
  const int64_t nblk = (n + kColBlock - 1) / kColBlock;
  return 2 * (int64_t)rows * (nblk > 0 ? nblk : 1);
}

}  // namespace plaidhip
