// C-ABI entry points (include/plaidhip.h): context lifecycle, device-level wrappers and the
// host-buffer pipelines the R `.Call` shim binds.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <algorithm>
#include <limits>
#include <vector>

#include "common.h"

namespace plaidhip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* last_error_cstr() { return g_err; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return (e == hipErrorOutOfMemory) ? PLAIDHIP_ENOMEM : PLAIDHIP_EHIP;
}

int ensure_workspace(plaidhip_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return PLAIDHIP_OK;
  if (ctx->ws) {
    PH_HIP(hipStreamSynchronize(ctx->stream));
    PH_HIP(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  bytes = (bytes + 4095) & ~(size_t)4095;
  PH_HIP(hipMalloc(&ctx->ws, bytes));
  ctx->ws_bytes = bytes;
  return PLAIDHIP_OK;
}

int32_t host_max_col_nnz(const int32_t* Xp, int32_t n) {
  int32_t mx = 0;
  for (int32_t c = 0; c < n; ++c) mx = std::max(mx, Xp[c + 1] - Xp[c]);
  return mx;
}

int ctx_buffer(plaidhip_ctx* ctx, int k, size_t bytes, void** out) {
  if (bytes < 256) bytes = 256;
  if (ctx->hbuf_bytes[k] < bytes) {
    if (ctx->hbuf[k]) {
      PH_HIP(hipStreamSynchronize(ctx->stream));
      PH_HIP(hipFree(ctx->hbuf[k]));
      ctx->hbuf[k] = nullptr;
      ctx->hbuf_bytes[k] = 0;
    }
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    PH_HIP(hipMalloc(&ctx->hbuf[k], want));
    ctx->hbuf_bytes[k] = want;
  }
  *out = ctx->hbuf[k];
  return PLAIDHIP_OK;
}

int allow_full_lds(plaidhip_ctx* ctx, const void* kernel, std::atomic<uint32_t>* done_mask) {
  if (ctx->device >= 32) {   // no bit for it: set the attribute every time (idempotent)
    PH_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    return PLAIDHIP_OK;
  }
  const uint32_t bit = 1u << ctx->device;
  if (done_mask->load(std::memory_order_acquire) & bit) return PLAIDHIP_OK;
  PH_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  done_mask->fetch_or(bit, std::memory_order_release);
  return PLAIDHIP_OK;
}

}  // namespace plaidhip

using namespace plaidhip;

#define PH_CTX(ctx)                                   \
  PH_REQUIRE((ctx) != nullptr, "null plaidhip_ctx");  \
  PH_HIP(hipSetDevice((ctx)->device))

#define PH_TRY(expr)                      \
  do {                                    \
    int rc_ = (expr);                     \
    if (rc_ != PLAIDHIP_OK) return rc_;   \
  } while (0)

namespace plaidhip {
int DevBuf::alloc(size_t bytes) {
  hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
  if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
  return PLAIDHIP_OK;
}
}  // namespace plaidhip

namespace {

// The host-level entry points prepare the gene-set collection themselves.  An R session calls them again and
// again with the same matG, so the last few prepared collections stay in the context (keyed by sizes and a hash
// of the pattern) instead of being rebuilt: preparing C2's 5,000 sets costs ~20 ms of a 76 ms call.
struct GenesetHolder {
  plaidhip_geneset* gs = nullptr;   // owned by the context's cache
};

// two independent 64-bit hashes over the int32 words of the pattern (a 128-bit key; it only ever lives in this process).
// Two ids per step and four interleaved lanes per hash: one multiply chain is latency-bound at ~1.4 ns per id (7.5 ms of
// every call for the 5.2e6 ids of a 61,459-set collection), four chains side by side fill the multiplier.
void hash_words(const int32_t* w, size_t count, uint64_t& h1, uint64_t& h2) {
  uint64_t a[4], b[4];
  for (int l = 0; l < 4; ++l) {
    a[l] = h1 + (uint64_t)l * 0x9e3779b97f4a7c15ull;
    b[l] = h2 ^ ((uint64_t)(l + 1) * 0xd6e8feb86659fd93ull);
  }
  size_t i = 0;
  for (; i + 8 <= count; i += 8)
    for (int l = 0; l < 4; ++l) {
      uint64_t v;
      memcpy(&v, w + i + 2 * l, 8);
      a[l] = (a[l] ^ v) * 0x100000001b3ull;
      a[l] ^= a[l] >> 29;
      b[l] = (b[l] + v + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
      b[l] ^= b[l] >> 31;
    }
  for (; i < count; ++i) {   // the last words, with their position
    const uint64_t v = ((uint64_t)(uint32_t)w[i] << 3) | (uint64_t)(i & 7);
    a[0] = (a[0] ^ v) * 0x100000001b3ull;
    a[0] ^= a[0] >> 29;
    b[0] = (b[0] + v + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
    b[0] ^= b[0] >> 31;
  }
  h1 = count;
  h2 = ~(uint64_t)count;
  for (int l = 0; l < 4; ++l) {
    h1 = (h1 ^ a[l]) * 0x94d049bb133111ebull;
    h1 ^= h1 >> 32;
    h2 = (h2 + b[l]) * 0xff51afd7ed558ccdull;
    h2 ^= h2 >> 33;
  }
}

}  // namespace
namespace plaidhip {
int acquire_geneset(plaidhip_ctx* ctx, int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, plaidhip_geneset** out) {
  *out = nullptr;
  uint64_t h = 1469598103934665603ull, h2 = 0x243f6a8885a308d3ull;
  if (Gp != nullptr && m >= 0) {
    hash_words(Gp, (size_t)m + 1, h, h2);
    const int64_t z = m > 0 ? Gp[m] : 0;
    if (Gi != nullptr && z > 0) hash_words(Gi, (size_t)z, h, h2);
  }
  for (size_t k = 0; k < ctx->gs_cache.size(); ++k) {
    plaidhip_ctx::cached_geneset& e = ctx->gs_cache[k];
    if (e.hash == h && e.hash2 == h2 && e.g == g && e.m == m) {
      plaidhip_ctx::cached_geneset hit = e;
      ctx->gs_cache.erase(ctx->gs_cache.begin() + (long)k);
      ctx->gs_cache.push_back(hit);   // most recently used last
      *out = hit.gs;
      return PLAIDHIP_OK;
    }
  }
  plaidhip_geneset* gs = nullptr;
  PH_TRY(plaidhip_geneset_create(ctx, g, m, Gp, Gi, &gs));
  if (ctx->gs_cache.size() >= 4) {
    plaidhip_geneset_destroy(ctx->gs_cache.front().gs);
    ctx->gs_cache.erase(ctx->gs_cache.begin());
  }
  ctx->gs_cache.push_back(plaidhip_ctx::cached_geneset{h, h2, g, m, gs});
  *out = gs;
  return PLAIDHIP_OK;
}
}  // namespace plaidhip
namespace {

// (pageable memory: through the pinned staging ring once it is worth it, multi.cpp)
int h2d(plaidhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  return upload_host(ctx, dst, 1, src, 1, (int64_t)bytes);
}

// Dense host matrix (g x n, column-major) -> device with an EVEN leading dimension, so that both
// columns of a pair start 16-byte aligned (the two-columns-per-pass SpMM kernel needs that).
inline int64_t even_ld(int32_t g) { return (int64_t)g + (g & 1); }
int h2d_cols(plaidhip_ctx* ctx, void* dst, int64_t ldd, const double* src, int32_t g, int32_t n) {
  if ((int64_t)g * n == 0) return PLAIDHIP_OK;
  if (ldd == g) return h2d(ctx, dst, src, (size_t)g * n * 8);
  return upload_host(ctx, dst, (size_t)ldd * 8, src, (size_t)g * 8, n);
}

// normalize_medians on a device-resident S (R/plaid.R:554-575), fully enqueued: ignore.zero is
// resolved on the device from the flag words, mean(medx) from the {sum, count} pair.
int normalize_on_device(plaidhip_ctx* ctx, double* dS, int32_t m, int32_t n, int ignore_zero,
                        uint32_t* d_flags, bool have_flags, double* d_med, double* d_red) {
  if (ignore_zero == PLAIDHIP_IGNORE_ZERO_AUTO && !have_flags) {
    PH_HIP(hipMemsetAsync(d_flags, 0, 4 * sizeof(uint32_t), ctx->stream));
    PH_TRY(launch_minflags(ctx, dS, (int64_t)m * n, d_flags));
  }
  PH_TRY(launch_col_medians(ctx, dS, m, m, n, ignore_zero, d_flags, d_med));
  PH_TRY(launch_sum(ctx, d_med, n, d_red));
  PH_TRY(launch_shift_columns(ctx, dS, m, m, n, d_med, 0.0, d_red));
  return PLAIDHIP_OK;
}

}  // namespace
namespace plaidhip {
// dgCMatrix slots handed over by a host language: @p non-decreasing from 0, @i inside [0, g)
int check_host_csc(const int32_t* Xp, const int32_t* Xi, int32_t g, int32_t n) {
  PH_REQUIRE(Xp != nullptr, "null Xp");
  PH_REQUIRE(Xp[0] == 0, "Xp[0] = %d, expected 0", Xp[0]);
  for (int32_t c = 0; c < n; ++c)
    PH_REQUIRE(Xp[c + 1] >= Xp[c], "Xp decreases at column %d (more than 2^31-1 stored values? split the matrix by columns)", c);
  if (Xi != nullptr) {
    const int64_t zx = Xp[n];
    for (int64_t q = 0; q < zx; ++q)
      PH_REQUIRE(Xi[q] >= 0 && Xi[q] < g, "Xi[%lld] = %d outside [0, %d)", (long long)q, Xi[q], g);
  }
  return PLAIDHIP_OK;
}

int check_host_common(const void* G_p, int32_t g, int32_t n, int32_t m) {
  PH_REQUIRE(g > 0 && n >= 0 && m >= 0, "bad dims g=%d n=%d m=%d", g, n, m);
  PH_REQUIRE(G_p != nullptr, "null Gp");
  return PLAIDHIP_OK;
}
}  // namespace plaidhip
namespace {

}  // namespace

extern "C" {

int plaidhip_version(void) { return PLAIDHIP_VERSION; }

int plaidhip_set_precision(plaidhip_ctx* ctx, int mode) try {
  PH_CTX(ctx);
  PH_REQUIRE(mode == PLAIDHIP_PRECISION_F64 || mode == PLAIDHIP_PRECISION_MIXED, "set_precision: unknown mode %d", mode);
  ctx->precision = mode;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_set_stream(plaidhip_ctx* ctx, void* stream) try {
  PH_CTX(ctx);
  PH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream) PH_HIP(hipStreamDestroy(ctx->stream));
  ctx->stream = reinterpret_cast<hipStream_t>(stream);   // nullptr: the null stream
  ctx->own_stream = false;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_set_option(plaidhip_ctx* ctx, int option, int value) try {
  PH_CTX(ctx);
  switch (option) {
    case PLAIDHIP_OPT_SPMM_DENSE_KERNEL:
      PH_REQUIRE(value >= 0 && value <= 3, "set_option: dense kernel %d (0 auto, 1 one-column, 2 pair, 3 mfma)", value);
      ctx->opt_dense_kernel = value;
      break;
    case PLAIDHIP_OPT_SPMM_SPARSE_KERNEL:
      PH_REQUIRE(value >= 0 && value <= 2, "set_option: sparse kernel %d (0 auto, 1 scatter, 2 gather)", value);
      ctx->opt_sparse_kernel = value;
      break;
    case PLAIDHIP_OPT_NT_STORE:
      PH_REQUIRE(value >= -1 && value <= 1, "set_option: nt_store %d (-1 auto, 0, 1)", value);
      ctx->opt_nt_store = value;
      break;
    case PLAIDHIP_OPT_RANKS_F32:
      PH_REQUIRE(value >= 0 && value <= 2, "set_option: ranks_f32 %d (0 fp64, 1 fp32 staging, 2 u16 staging)", value);
      ctx->opt_ranks_f32 = value;
      break;
    case PLAIDHIP_OPT_SCATTER_FIXED:
      PH_REQUIRE(value == 0 || value == 1, "set_option: scatter_fixed %d (0, 1)", value);
      ctx->opt_scatter_fixed = value;
      break;
    case PLAIDHIP_OPT_SCATTER_ORDER:
      PH_REQUIRE(value == 0 || value == 1, "set_option: scatter_order %d (0 column-major, 1 chunk-major)", value);
      ctx->opt_scatter_order = value;
      break;
    case PLAIDHIP_OPT_FUSED_MEDIANS:
      PH_REQUIRE(value >= 0 && value <= 2, "set_option: fused_medians %d (0 by size, 1 whenever possible, 2 never)", value);
      ctx->opt_fused_medians = value;
      break;
    case PLAIDHIP_OPT_RANK_KERNEL:
      PH_REQUIRE(value >= 0 && value <= 3, "set_option: rank kernel %d (0 auto, 1 network, 2 bucket, 3 bucket with 512 x 40 for long columns)", value);
      ctx->opt_rank_kernel = value;
      break;
    default:
      PH_REQUIRE(false, "set_option: unknown option %d", option);
  }
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

#ifdef PLAIDHIP_DIAG
int plaidhip_debug_set_ablation(int mode, void* dbg) { debug_set_ablation(mode, dbg); return PLAIDHIP_OK; }
int plaidhip_debug_set_rank_stamps(void* dbg) { debug_set_rank_stamps(dbg); return PLAIDHIP_OK; }
int plaidhip_debug_set_median_stamps(void* dbg) { debug_set_median_stamps(dbg); return PLAIDHIP_OK; }
#endif

const char* plaidhip_last_error_string(void) { return g_err; }

int plaidhip_device_count(int* count) try {
  PH_REQUIRE(count != nullptr, "device_count: null out pointer");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__);
  }
  *count = n;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_init(int device, void* stream, plaidhip_ctx** out) try {
  PH_REQUIRE(out != nullptr, "init: null out pointer");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    set_error("no HIP device visible (this library has no CPU path)");
    return PLAIDHIP_ENODEVICE;
  }
  PH_REQUIRE(device >= 0 && device < n, "init: device %d out of range [0,%d)", device, n);
  PH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PH_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return PLAIDHIP_ENODEVICE;
  }
  auto* ctx = new (std::nothrow) plaidhip_ctx();
  if (!ctx) { set_error("out of host memory"); return PLAIDHIP_ENOMEM; }
  ctx->device = device;
  ctx->num_cu = prop.multiProcessorCount;
  if (stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    ctx->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    ctx->own_stream = true;
  }
  {
    void* sel = nullptr;   // two doubles the sparse crossprod decides its accumulator format from (allocated here: a launch
                           // inside a stream capture must not allocate)
    hipError_t e = hipMalloc(&sel, 128);
    if (e == hipSuccess) e = hipMemset(sel, 0, 128);
    // bytes 96..127: the EMPTY median bracket {offset 0, half width -1, ignore-zero 0} of a calibration launch (no score is a
    // candidate), written once here so that no launch needs a host-to-device copy
    static const double kEmptyBracket[4] = {0.0, -1.0, 0.0, 0.0};
    if (e == hipSuccess) e = hipMemcpy(static_cast<char*>(sel) + 96, kEmptyBracket, 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      if (ctx->own_stream) hipStreamDestroy(ctx->stream);
      delete ctx;
      return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    }
    ctx->d_sel = static_cast<double*>(sel);
    ctx->d_spec = reinterpret_cast<uint32_t*>(static_cast<char*>(sel) + 64);
  }
  *out = ctx;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_limit(int which, int64_t* value) try {
  PH_REQUIRE(value != nullptr, "limit: null value");
  switch (which) {
    case PLAIDHIP_LIMIT_SPARSE_RANK_COLUMN: *value = max_sparse_rank_column(); return PLAIDHIP_OK;
    case PLAIDHIP_LIMIT_LDS_GENES: *value = kMaxLdsGenes; return PLAIDHIP_OK;
    default: set_error("limit: unknown id %d", which); return PLAIDHIP_EINVAL;
  }
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_finalize(plaidhip_ctx* ctx) try {
  if (!ctx) return PLAIDHIP_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (plaidhip_ctx::cached_geneset& e : ctx->gs_cache) plaidhip_geneset_destroy(e.gs);
  ctx->gs_cache.clear();
  if (ctx->ws) hipFree(ctx->ws);
  if (ctx->rank_scratch) hipFree(ctx->rank_scratch);
  if (ctx->tie_scratch) hipFree(ctx->tie_scratch);
  if (ctx->fmed_buf) hipFree(ctx->fmed_buf);
  if (ctx->d_sel) hipFree(ctx->d_sel);
  for (int k = 0; k < plaidhip_ctx::kHostBufs; ++k)
    if (ctx->hbuf[k]) hipFree(ctx->hbuf[k]);
  for (int t = 0; t < plaidhip_ctx::kFeeders; ++t) {
    for (int b = 0; b < 2; ++b)
      if (ctx->pin[t][b]) hipHostFree(ctx->pin[t][b]);
    if (ctx->copy_stream[t]) hipStreamDestroy(ctx->copy_stream[t]);
  }
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_synchronize(plaidhip_ctx* ctx) try {
  PH_CTX(ctx);
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_malloc(plaidhip_ctx* ctx, size_t bytes, void** dptr) try {
  PH_CTX(ctx);
  PH_REQUIRE(dptr != nullptr, "malloc: null out pointer");
  PH_HIP(hipMalloc(dptr, bytes ? bytes : 16));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_free(plaidhip_ctx* ctx, void* dptr) try {
  PH_CTX(ctx);
  if (dptr) PH_HIP(hipFree(dptr));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_memcpy_h2d(plaidhip_ctx* ctx, void* dst, const void* src, size_t bytes) try {
  PH_CTX(ctx);
  if (bytes) PH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_memcpy_d2h(plaidhip_ctx* ctx, void* dst, const void* src, size_t bytes) try {
  PH_CTX(ctx);
  if (bytes) PH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

// ---- device-level ---------------------------------------------------------------------

int plaidhip_dev_spmm_dense_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* X,
                                int64_t ldx, int32_t n, int stat, double alpha, const void* alpha_div,
                                double beta, void* S, int64_t lds, void* flags) try {
  PH_CTX(ctx);
  PH_REQUIRE(gs != nullptr, "spmm: null geneset");
  PH_REQUIRE(n >= 0, "spmm: n=%d", n);
  PH_REQUIRE(n == 0 || (X != nullptr && S != nullptr), "spmm: null X/S");
  PH_REQUIRE(ldx >= gs->g && lds >= gs->m, "spmm: leading dims ldx=%lld (g=%d) lds=%lld (m=%d)",
             (long long)ldx, gs->g, (long long)lds, gs->m);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "spmm: bad stat %d", stat);
  return launch_spmm_dense_f64(ctx, gs, static_cast<const double*>(X), ldx, n, stat, alpha,
                               static_cast<const double*>(alpha_div), beta, static_cast<double*>(S), lds,
                               static_cast<uint32_t*>(flags));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_spmm_dense_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* X,
                                      int64_t ldx, int32_t n, int stat, double alpha, const void* alpha_div,
                                      double beta, void* S, int64_t lds, void* flags) try {
  PH_CTX(ctx);
  PH_REQUIRE(gs != nullptr, "spmm_dense_fused: null geneset");
  PH_REQUIRE(n >= 0, "spmm_dense_fused: n=%d", n);
  PH_REQUIRE(n == 0 || (X != nullptr && S != nullptr), "spmm_dense_fused: null X/S");
  PH_REQUIRE(ldx >= gs->g && lds >= gs->m, "spmm_dense_fused: leading dims ldx=%lld (g=%d) lds=%lld (m=%d)",
             (long long)ldx, gs->g, (long long)lds, gs->m);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "spmm_dense_fused: bad stat %d", stat);
  return launch_spmm_dense_fused_f64(ctx, gs, static_cast<const double*>(X), ldx, n, stat, alpha,
                                     static_cast<const double*>(alpha_div), beta, static_cast<double*>(S), lds,
                                     static_cast<uint32_t*>(flags));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_spmm_ranks_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* R,
                                int64_t ldr, int32_t n, int stat, double alpha, const void* alpha_div,
                                double beta, void* S, int64_t lds, void* flags) try {
  PH_CTX(ctx);
  PH_REQUIRE(gs != nullptr, "spmm_ranks: null geneset");
  PH_REQUIRE(n >= 0, "spmm_ranks: n=%d", n);
  PH_REQUIRE(n == 0 || (R != nullptr && S != nullptr), "spmm_ranks: null R/S");
  PH_REQUIRE(ldr >= gs->g && lds >= gs->m, "spmm_ranks: leading dims ldr=%lld (g=%d) lds=%lld (m=%d)",
             (long long)ldr, gs->g, (long long)lds, gs->m);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "spmm_ranks: bad stat %d", stat);
  return launch_spmm_dense_f64(ctx, gs, static_cast<const double*>(R), ldr, n, stat, alpha,
                               static_cast<const double*>(alpha_div), beta, static_cast<double*>(S), lds,
                               static_cast<uint32_t*>(flags), PLAIDHIP_X_RANKS);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_spmm_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* Xp,
                              const void* Xi, const void* Xx, int32_t n, int64_t nnz, int stat, double alpha,
                              const void* alpha_div, double beta, void* S, int64_t lds, void* flags) try {
  PH_CTX(ctx);
  PH_REQUIRE(gs != nullptr, "spmm_csc: null geneset");
  PH_REQUIRE(n >= 0, "spmm_csc: n=%d", n);
  PH_REQUIRE(n == 0 || (Xp != nullptr && S != nullptr), "spmm_csc: null Xp/S");
  PH_REQUIRE(lds >= gs->m, "spmm_csc: lds=%lld < m=%d", (long long)lds, gs->m);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "spmm_csc: bad stat %d", stat);
  return launch_spmm_csc_f64(ctx, gs, static_cast<const int32_t*>(Xp), static_cast<const int32_t*>(Xi),
                             static_cast<const double*>(Xx), n, nnz, stat, alpha,
                             static_cast<const double*>(alpha_div), beta, static_cast<double*>(S), lds,
                             static_cast<uint32_t*>(flags));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_crossprod_weighted_f64(plaidhip_ctx* ctx, const void* Wp, const void* Wi, const void* Wx, int32_t g,
                                        int32_t m, const void* Y, int64_t ldy, int32_t n, void* S, int64_t lds) try {
  PH_CTX(ctx);
  PH_REQUIRE(g > 0 && m >= 0 && n >= 0, "crossprod_weighted: bad dims g=%d m=%d n=%d", g, m, n);
  PH_REQUIRE(Wp != nullptr, "crossprod_weighted: null x@p");
  PH_REQUIRE((int64_t)m * n == 0 || (Y != nullptr && S != nullptr), "crossprod_weighted: null y/S");
  PH_REQUIRE(ldy >= g && lds >= m, "crossprod_weighted: leading dims ldy=%lld (g=%d) lds=%lld (m=%d)", (long long)ldy, g,
             (long long)lds, m);
  return launch_crossprod_weighted_f64(ctx, static_cast<const int32_t*>(Wp), static_cast<const int32_t*>(Wi),
                                       static_cast<const double*>(Wx), g, m, static_cast<const double*>(Y), ldy, nullptr,
                                       nullptr, nullptr, n, static_cast<double*>(S), lds);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_crossprod_weighted_csc_f64(plaidhip_ctx* ctx, const void* Wp, const void* Wi, const void* Wx, int32_t g,
                                            int32_t m, const void* Yp, const void* Yi, const void* Yx, int32_t n,
                                            void* S, int64_t lds) try {
  PH_CTX(ctx);
  PH_REQUIRE(g > 0 && m >= 0 && n >= 0, "crossprod_weighted_csc: bad dims g=%d m=%d n=%d", g, m, n);
  PH_REQUIRE(Wp != nullptr, "crossprod_weighted_csc: null x@p");
  PH_REQUIRE((int64_t)m * n == 0 || (Yp != nullptr && S != nullptr), "crossprod_weighted_csc: null y@p/S");
  PH_REQUIRE(lds >= m, "crossprod_weighted_csc: lds=%lld < m=%d", (long long)lds, m);
  return launch_crossprod_weighted_f64(ctx, static_cast<const int32_t*>(Wp), static_cast<const int32_t*>(Wi),
                                       static_cast<const double*>(Wx), g, m, nullptr, 0, static_cast<const int32_t*>(Yp),
                                       static_cast<const int32_t*>(Yi), static_cast<const double*>(Yx), n,
                                       static_cast<double*>(S), lds);
} catch (...) { return plaidhip::on_exception(); }

// which: 0 dense columns (matrixStats::colRanks: average / min / max / first / last / dense), 1 the stored values of a
// dgCMatrix (base::rank, R/plaid.R:639-642: no "dense"), 2 a dgCMatrix with its zeros ranked (sparseMatrixStats::colRanks,
// R/plaid.R:605-608: max / average / min only).  "random" is legal in R and refused here: its result is not a function of
// the input.  first / last / dense come without the fused power and column maximum (no caller of the reference combines them).
static int check_ties(int ties, int which = 0, double power = 1.0, const void* colmax = nullptr) {
  if (ties == PLAIDHIP_TIES_RANDOM) {
    set_error("colranks: ties.method = \"random\" is not supported (the ranks would not be reproducible)");
    return PLAIDHIP_EUNSUPPORTED;
  }
  PH_REQUIRE(ties >= PLAIDHIP_TIES_AVERAGE && ties <= PLAIDHIP_TIES_DENSE, "colranks: unknown ties.method code %d", ties);
  if (ties >= PLAIDHIP_TIES_FIRST) {
    PH_REQUIRE(which != 2, "colranks: 'arg' should be one of \"max\", \"average\", \"min\" for a sparse matrix whose zeros are "
               "ranked (sparseMatrixStats::colRanks, R/plaid.R:605-608)");
    PH_REQUIRE(!(which == 1 && ties == PLAIDHIP_TIES_DENSE), "sparse_colranks: 'arg' should be one of \"average\", \"first\", \"last\", "
               "\"random\", \"max\", \"min\" (base::rank, R/plaid.R:639-642)");
    PH_REQUIRE(power == 1.0 && colmax == nullptr, "colranks: ties.method first / last / dense come without power / colmax");
  }
  return PLAIDHIP_OK;
}

int plaidhip_dev_spmm_csc_ranks_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* Xp,
                                    const void* Xi, const void* Rx, int32_t n, int64_t nnz, int stat, double alpha,
                                    const void* rmax, double beta, void* S, int64_t lds, void* flags) try {
  PH_CTX(ctx);
  PH_REQUIRE(gs != nullptr, "spmm_csc_ranks: null geneset");
  PH_REQUIRE(n >= 0, "spmm_csc_ranks: n=%d", n);
  PH_REQUIRE(n == 0 || (Xp != nullptr && S != nullptr), "spmm_csc_ranks: null Xp/S");
  PH_REQUIRE(rmax != nullptr, "spmm_csc_ranks: rmax (device double: the maximum of Rx) is required");
  PH_REQUIRE(lds >= gs->m, "spmm_csc_ranks: lds=%lld < m=%d", (long long)lds, gs->m);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "spmm_csc_ranks: bad stat %d", stat);
  return launch_spmm_csc_f64(ctx, gs, static_cast<const int32_t*>(Xp), static_cast<const int32_t*>(Xi),
                             static_cast<const double*>(Rx), n, nnz, stat, alpha, static_cast<const double*>(rmax), beta,
                             static_cast<double*>(S), lds, static_cast<uint32_t*>(flags), /*bounded=*/true,
                             static_cast<const double*>(rmax), 0.0);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_spmm_csc_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const void* Xp, const void* Xi,
                                    const void* Xx, int32_t n, int64_t nnz, int stat, double alpha, const void* alpha_div,
                                    double beta, void* S, int64_t lds, void* flags, const void* rmax) try {
  PH_CTX(ctx);
  PH_REQUIRE(gs != nullptr, "spmm_csc_fused: null geneset");
  PH_REQUIRE(n >= 0, "spmm_csc_fused: n=%d", n);
  PH_REQUIRE(n == 0 || (Xp != nullptr && S != nullptr), "spmm_csc_fused: null Xp/S");
  PH_REQUIRE(lds >= gs->m, "spmm_csc_fused: lds=%lld < m=%d", (long long)lds, gs->m);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "spmm_csc_fused: bad stat %d", stat);
  PH_REQUIRE(rmax == nullptr || alpha_div == nullptr || alpha_div == rmax, "spmm_csc_fused: rmax and alpha_div differ");
  const double* div = static_cast<const double*>(rmax != nullptr ? rmax : alpha_div);
  return launch_spmm_csc_fused_f64(ctx, gs, static_cast<const int32_t*>(Xp), static_cast<const int32_t*>(Xi),
                                   static_cast<const double*>(Xx), n, nnz, stat, alpha, div, beta, static_cast<double*>(S), lds,
                                   static_cast<uint32_t*>(flags), rmax != nullptr, static_cast<const double*>(rmax), 0.0);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_col_medians_resume(plaidhip_ctx* ctx, const void* S, int64_t lds, int32_t m, int32_t n, int ignore_zero,
                                    const void* flags, void* med) try {
  PH_CTX(ctx);
  PH_REQUIRE(m >= 0 && n >= 0 && lds >= m, "col_medians_resume: bad dims m=%d n=%d lds=%lld", m, n, (long long)lds);
  PH_REQUIRE(n == 0 || (S != nullptr && med != nullptr), "col_medians_resume: null S/med");
  PH_REQUIRE(ignore_zero >= -1 && ignore_zero <= 1, "col_medians_resume: ignore_zero=%d", ignore_zero);
  PH_REQUIRE(ignore_zero >= 0 || flags != nullptr, "col_medians_resume: ignore_zero = auto needs the flag words");
  return launch_col_medians_resume(ctx, static_cast<const double*>(S), lds, m, n, ignore_zero,
                                   static_cast<const uint32_t*>(flags), static_cast<double*>(med));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_col_medians_resume_token(plaidhip_ctx* ctx, int64_t token, const void* S, int64_t lds, int32_t m, int32_t n,
                                          int ignore_zero, const void* flags, void* med) try {
  PH_CTX(ctx);
  PH_REQUIRE(token >= 0, "col_medians_resume_token: token=%lld (0 = none, > 0 = what fused_medians_info returned)", (long long)token);
  PH_REQUIRE(m >= 0 && n >= 0 && lds >= m, "col_medians_resume_token: bad dims m=%d n=%d lds=%lld", m, n, (long long)lds);
  PH_REQUIRE(n == 0 || (S != nullptr && med != nullptr), "col_medians_resume_token: null S/med");
  PH_REQUIRE(ignore_zero >= -1 && ignore_zero <= 1, "col_medians_resume_token: ignore_zero=%d", ignore_zero);
  PH_REQUIRE(ignore_zero >= 0 || flags != nullptr, "col_medians_resume_token: ignore_zero = auto needs the flag words");
  return launch_col_medians_resume(ctx, static_cast<const double*>(S), lds, m, n, ignore_zero,
                                   static_cast<const uint32_t*>(flags), static_cast<double*>(med), token);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_fused_medians_discard(plaidhip_ctx* ctx) try {
  PH_CTX(ctx);
  ctx->fmed.valid = false;
  ctx->fmed.token = 0;
  // the caller will not normalise: the candidate scratch (up to 0.2 x the bytes of S) goes back too.  (Stream-ordered work
  // may still read it: wait for the stream first.  The next fused crossprod allocates again.)
  if (ctx->fmed_buf != nullptr) {
    PH_HIP(hipStreamSynchronize(ctx->stream));
    PH_HIP(hipFree(ctx->fmed_buf));
    ctx->fmed_buf = nullptr;
    ctx->fmed_bytes = 0;
    ctx->fmed = decltype(ctx->fmed){};
  }
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_fused_medians_info(plaidhip_ctx* ctx, int64_t info[4]) try {
  PH_CTX(ctx);
  PH_REQUIRE(info != nullptr, "fused_medians_info: null info");
  info[0] = ctx->fmed.n;
  info[1] = (int64_t)reinterpret_cast<intptr_t>(ctx->fmed.status);
  info[2] = (int64_t)reinterpret_cast<intptr_t>(ctx->fmed.cal);
  info[3] = ctx->fmed.valid ? (int64_t)ctx->fmed.token : 0;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_colranks_dense_f64(plaidhip_ctx* ctx, const void* X, int64_t ldx, int32_t g,
                                    int32_t n, int ties, int is_signed, double power, void* R,
                                    int64_t ldr, void* colmax) try {
  PH_CTX(ctx);
  PH_TRY(check_ties(ties, 0, power, colmax));
  PH_REQUIRE(g >= 0 && n >= 0, "colranks: bad dims g=%d n=%d", g, n);
  PH_REQUIRE(g == 0 || n == 0 || (X && R), "colranks: null X/R");
  PH_REQUIRE(ldx >= g && ldr >= g, "colranks: leading dims below g");
  return launch_colranks_dense_f64(ctx, static_cast<const double*>(X), ldx, g, n, ties, is_signed,
                                   power, static_cast<double*>(R), ldr, static_cast<double*>(colmax));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_colranks_csc_f64(plaidhip_ctx* ctx, const void* Xp, const void* Xx, int32_t n,
                                  int32_t max_col_nnz, int ties, int is_signed, double power, void* Rx,
                                  void* colmax) try {
  PH_CTX(ctx);
  PH_TRY(check_ties(ties, 1, power, colmax));
  PH_REQUIRE(n >= 0 && max_col_nnz >= 0, "colranks_csc: n=%d max_col_nnz=%d", n, max_col_nnz);
  PH_REQUIRE(n == 0 || Xp != nullptr, "colranks_csc: null Xp");
  return launch_colranks_csc_f64(ctx, static_cast<const int32_t*>(Xp), static_cast<const double*>(Xx),
                                 n, max_col_nnz, ties, is_signed, power, static_cast<double*>(Rx),
                                 static_cast<double*>(colmax));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_colranks_csc_dense_f64(plaidhip_ctx* ctx, const void* Xp, const void* Xi, const void* Xx,
                                        int32_t g, int32_t n, int ties, int is_signed, double power,
                                        void* R, int64_t ldr, void* colmax) try {
  PH_CTX(ctx);
  PH_TRY(check_ties(ties, 2));
  PH_REQUIRE(g >= 0 && n >= 0 && ldr >= g, "colranks_csc_dense: bad dims g=%d n=%d ldr=%lld", g, n, (long long)ldr);
  PH_REQUIRE(n == 0 || g == 0 || (Xp && R), "colranks_csc_dense: null Xp/R");
  return launch_colranks_csc_dense_f64(ctx, static_cast<const int32_t*>(Xp), static_cast<const int32_t*>(Xi),
                                       static_cast<const double*>(Xx), g, n, ties, is_signed, power,
                                       static_cast<double*>(R), ldr, static_cast<double*>(colmax));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_colranks_csc_dense_nz_f64(plaidhip_ctx* ctx, const void* Xp, const void* Xi, const void* Xx, int32_t g,
                                           int32_t n, int32_t max_col_nnz, int ties, int is_signed, double power,
                                           void* Rx_scratch, void* R, int64_t ldr, void* colmax) try {
  PH_CTX(ctx);
  PH_TRY(check_ties(ties, 2));
  PH_REQUIRE(g >= 0 && n >= 0 && ldr >= g, "colranks_csc_dense_nz: bad dims g=%d n=%d ldr=%lld", g, n, (long long)ldr);
  PH_REQUIRE(n == 0 || g == 0 || (Xp && R), "colranks_csc_dense_nz: null Xp/R");
  PH_REQUIRE(max_col_nnz >= 0 && max_col_nnz <= g, "colranks_csc_dense_nz: max_col_nnz=%d outside [0, nrow(X)=%d]", max_col_nnz, g);
  if (max_col_nnz > max_sparse_rank_column()) {
    set_error("colranks_csc_dense_nz: a column with %d stored values (at most %d are ranked in one pass: use "
              "plaidhip_dev_colranks_csc_dense_f64)", max_col_nnz, max_sparse_rank_column());
    return PLAIDHIP_EUNSUPPORTED;
  }
  PH_REQUIRE(max_col_nnz == 0 || Rx_scratch != nullptr, "colranks_csc_dense_nz: null Rx_scratch");
  return launch_colranks_csc_dense_nz_f64(ctx, static_cast<const int32_t*>(Xp), static_cast<const int32_t*>(Xi),
                                          static_cast<const double*>(Xx), g, n, max_col_nnz, ties, is_signed, power,
                                          static_cast<double*>(Rx_scratch), static_cast<double*>(R), ldr,
                                          static_cast<double*>(colmax));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_minflags(plaidhip_ctx* ctx, const void* S, int64_t count, void* flags) try {
  PH_CTX(ctx);
  PH_REQUIRE(flags != nullptr && count >= 0, "minflags: bad arguments");
  return launch_minflags(ctx, static_cast<const double*>(S), count, static_cast<uint32_t*>(flags));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_col_medians(plaidhip_ctx* ctx, const void* S, int64_t lds, int32_t m, int32_t n,
                             int ignore_zero, const void* flags, void* med) try {
  PH_CTX(ctx);
  PH_REQUIRE(m >= 0 && n >= 0 && lds >= m, "col_medians: bad dims m=%d n=%d lds=%lld", m, n, (long long)lds);
  PH_REQUIRE(ignore_zero >= -1 && ignore_zero <= 1, "col_medians: bad ignore_zero %d", ignore_zero);
  PH_REQUIRE(ignore_zero >= 0 || flags != nullptr, "col_medians: ignore_zero=auto needs the flags words");
  PH_REQUIRE(n == 0 || med != nullptr, "col_medians: null med");
  return launch_col_medians(ctx, static_cast<const double*>(S), lds, m, n, ignore_zero,
                            static_cast<const uint32_t*>(flags), static_cast<double*>(med));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_sum(plaidhip_ctx* ctx, const void* v, int64_t count, void* out) try {
  PH_CTX(ctx);
  PH_REQUIRE(out != nullptr && count >= 0, "sum: bad arguments");
  return launch_sum(ctx, static_cast<const double*>(v), count, static_cast<double*>(out));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_max(plaidhip_ctx* ctx, const void* v, int64_t count, void* out) try {
  PH_CTX(ctx);
  PH_REQUIRE(out != nullptr && count >= 0, "max: bad arguments");
  return launch_max(ctx, static_cast<const double*>(v), count, static_cast<double*>(out));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_shift_columns(plaidhip_ctx* ctx, void* S, int64_t lds, int32_t m, int32_t n,
                               const void* med, double add, const void* red) try {
  PH_CTX(ctx);
  PH_REQUIRE(m >= 0 && n >= 0 && lds >= m, "shift_columns: bad dims");
  return launch_shift_columns(ctx, static_cast<double*>(S), lds, m, n, static_cast<const double*>(med), add,
                              static_cast<const double*>(red));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_shift_columns_cast_f32(plaidhip_ctx* ctx, const void* S, int64_t lds, int32_t m, int32_t n, const void* med,
                                        double add, const void* red, void* out, int64_t ldo) try {
  PH_CTX(ctx);
  PH_REQUIRE(m >= 0 && n >= 0 && lds >= m && ldo >= m, "shift_columns_cast_f32: bad dims");
  PH_REQUIRE((m == 0 || n == 0) || (S != nullptr && med != nullptr && out != nullptr), "shift_columns_cast_f32: null argument");
  return launch_shift_columns_cast_f32(ctx, static_cast<const double*>(S), lds, m, n, static_cast<const double*>(med), add,
                                       static_cast<const double*>(red), static_cast<float*>(out), ldo);
} catch (...) { return plaidhip::on_exception(); }

// ---- host-level pipelines ---------------------------------------------------------------


int plaidhip_plaid_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n,
                         const int32_t* Gp, const int32_t* Gi, int32_t m, int stat, int normalize,
                         double* S_out) try {
  PH_CTX(ctx);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "plaid_dense: bad stat %d", stat);
  PH_REQUIRE(n == 0 || (X && S_out), "plaid_dense: null X/S_out");
  // one shard on this context: pipelined upload, crossprod per column panel, normalize_medians, download (multi.cpp)
  return run_sharded(&ctx, 1, 0, nullptr, nullptr, X, g, n, Gp, Gi, m, stat, normalize, 0.0, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_plaid_csc(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                       int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                       int stat, int normalize, double* S_out) try {
  PH_CTX(ctx);
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "plaid_csc: bad stat %d", stat);
  PH_REQUIRE(Xp != nullptr && (n == 0 || S_out), "plaid_csc: null Xp/S_out");
  return run_sharded(&ctx, 1, 0, Xp, Xi, Xx, g, n, Gp, Gi, m, stat, normalize, 0.0, S_out);
} catch (...) { return plaidhip::on_exception(); }

// chunked_crossprod with a general sparse x: upload the slots, one launch, download (host pointers)
static int crossprod_weighted_host(plaidhip_ctx* ctx, const int32_t* Wp, const int32_t* Wi, const double* Wx, int32_t g,
                                   int32_t m, const double* Y, const int32_t* Yp, const int32_t* Yi, const double* Yx,
                                   int32_t n, double* S_out) {
  PH_REQUIRE(g > 0 && m >= 0 && n >= 0, "crossprod_weighted: bad dims g=%d m=%d n=%d", g, m, n);
  if ((int64_t)m * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(S_out != nullptr, "crossprod_weighted: null S_out");
  PH_TRY(check_host_csc(Wp, Wi, g, m));
  const int64_t zw = Wp[m];
  PH_REQUIRE(zw == 0 || (Wi != nullptr && Wx != nullptr), "crossprod_weighted: null x@i / x@x");
  int64_t zy = 0;
  if (Yp != nullptr) {
    PH_TRY(check_host_csc(Yp, Yi, g, n));
    zy = Yp[n];
    PH_REQUIRE(zy == 0 || (Yi != nullptr && Yx != nullptr), "crossprod_weighted: null y@i / y@x");
  } else {
    PH_REQUIRE(Y != nullptr, "crossprod_weighted: null y");
  }
  DevBuf dWp, dWi, dWx, dY, dYi, dYx, dS;
  PH_TRY(dWp.alloc((size_t)(m + 1) * 4));
  PH_TRY(dWi.alloc((size_t)(zw > 0 ? zw : 1) * 4));
  PH_TRY(dWx.alloc((size_t)(zw > 0 ? zw : 1) * 8));
  PH_TRY(dS.alloc((size_t)m * n * 8));
  PH_TRY(h2d(ctx, dWp.p, Wp, (size_t)(m + 1) * 4));
  PH_TRY(h2d(ctx, dWi.p, Wi, (size_t)zw * 4));
  PH_TRY(h2d(ctx, dWx.p, Wx, (size_t)zw * 8));
  if (Yp != nullptr) {
    PH_TRY(dY.alloc((size_t)(n + 1) * 4));
    PH_TRY(dYi.alloc((size_t)(zy > 0 ? zy : 1) * 4));
    PH_TRY(dYx.alloc((size_t)(zy > 0 ? zy : 1) * 8));
    PH_TRY(h2d(ctx, dY.p, Yp, (size_t)(n + 1) * 4));
    PH_TRY(h2d(ctx, dYi.p, Yi, (size_t)zy * 4));
    PH_TRY(h2d(ctx, dYx.p, Yx, (size_t)zy * 8));
    PH_TRY(launch_crossprod_weighted_f64(ctx, dWp.as<int32_t>(), dWi.as<int32_t>(), dWx.as<double>(), g, m, nullptr, 0,
                                         dY.as<int32_t>(), dYi.as<int32_t>(), dYx.as<double>(), n, dS.as<double>(), m));
  } else {
    PH_TRY(dY.alloc((size_t)g * n * 8));
    PH_TRY(h2d(ctx, dY.p, Y, (size_t)g * n * 8));
    PH_TRY(launch_crossprod_weighted_f64(ctx, dWp.as<int32_t>(), dWi.as<int32_t>(), dWx.as<double>(), g, m, dY.as<double>(),
                                         g, nullptr, nullptr, nullptr, n, dS.as<double>(), m));
  }
  PH_TRY(copy_home(ctx, S_out, dS.p, (size_t)m * n * 8));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
}

int plaidhip_crossprod_weighted_dense(plaidhip_ctx* ctx, const int32_t* Wp, const int32_t* Wi, const double* Wx,
                                      int32_t g, int32_t m, const double* Y, int32_t n, double* S_out) try {
  PH_CTX(ctx);
  return crossprod_weighted_host(ctx, Wp, Wi, Wx, g, m, Y, nullptr, nullptr, nullptr, n, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_crossprod_weighted_csc(plaidhip_ctx* ctx, const int32_t* Wp, const int32_t* Wi, const double* Wx,
                                    int32_t g, int32_t m, const int32_t* Yp, const int32_t* Yi, const double* Yx,
                                    int32_t n, double* S_out) try {
  PH_CTX(ctx);
  PH_REQUIRE(Yp != nullptr, "crossprod_weighted_csc: null y@p");
  return crossprod_weighted_host(ctx, Wp, Wi, Wx, g, m, nullptr, Yp, Yi, Yx, n, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_normalize_medians(plaidhip_ctx* ctx, double* S, int32_t m, int32_t n, int ignore_zero,
                               double* med_out) try {
  PH_CTX(ctx);
  PH_REQUIRE(m >= 0 && n >= 0, "normalize_medians: bad dims");
  PH_REQUIRE(ignore_zero >= -1 && ignore_zero <= 1, "normalize_medians: bad ignore_zero %d", ignore_zero);
  if ((int64_t)m * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(S != nullptr, "normalize_medians: null S");
  DevBuf dS, dsmall;
  PH_TRY(dS.alloc((size_t)m * n * 8));
  PH_TRY(dsmall.alloc(64 + (size_t)n * 8));
  uint32_t* d_flags = dsmall.as<uint32_t>();
  double* d_red = reinterpret_cast<double*>(dsmall.as<char>() + 16);
  double* d_med = reinterpret_cast<double*>(dsmall.as<char>() + 64);
  PH_TRY(h2d(ctx, dS.p, S, (size_t)m * n * 8));
  PH_TRY(normalize_on_device(ctx, dS.as<double>(), m, n, ignore_zero, d_flags, false, d_med, d_red));
  PH_TRY(copy_home(ctx, S, dS.p, (size_t)m * n * 8));
  if (med_out) PH_HIP(hipMemcpyAsync(med_out, d_med, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_colranks_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n, int ties,
                            int is_signed, double* R_out) try {
  PH_CTX(ctx);
  PH_TRY(check_ties(ties, 0));
  PH_REQUIRE(g >= 0 && n >= 0, "colranks_dense: bad dims");
  if ((int64_t)g * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(X && R_out, "colranks_dense: null X/R_out");
  DevBuf dX, dR;
  PH_TRY(dX.alloc((size_t)g * n * 8));
  PH_TRY(dR.alloc((size_t)g * n * 8));
  PH_TRY(h2d(ctx, dX.p, X, (size_t)g * n * 8));
  PH_TRY(launch_colranks_dense_f64(ctx, dX.as<double>(), g, g, n, ties, is_signed, 1.0, dR.as<double>(), g, nullptr));
  PH_TRY(copy_home(ctx, R_out, dR.p, (size_t)g * n * 8));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_colranks_csc(plaidhip_ctx* ctx, const int32_t* Xp, const double* Xx, int32_t n,
                          int ties, int is_signed, double* Rx_out) try {
  PH_CTX(ctx);
  PH_TRY(check_ties(ties, 1));
  PH_REQUIRE(n >= 0 && Xp != nullptr, "colranks_csc: bad arguments");
  PH_TRY(check_host_csc(Xp, nullptr, 0, n));
  const int64_t zx = Xp[n];
  if (zx == 0) return PLAIDHIP_OK;
  PH_REQUIRE(Xx && Rx_out, "colranks_csc: null Xx/Rx_out");
  DevBuf dXp, dXx, dR;
  PH_TRY(dXp.alloc((size_t)(n + 1) * 4));
  PH_TRY(dXx.alloc((size_t)zx * 8));
  PH_TRY(dR.alloc((size_t)zx * 8));
  PH_TRY(h2d(ctx, dXp.p, Xp, (size_t)(n + 1) * 4));
  PH_TRY(h2d(ctx, dXx.p, Xx, (size_t)zx * 8));
  PH_TRY(launch_colranks_csc_f64(ctx, dXp.as<int32_t>(), dXx.as<double>(), n, host_max_col_nnz(Xp, n), ties, is_signed,
                                 1.0, dR.as<double>(), nullptr));
  PH_TRY(copy_home(ctx, Rx_out, dR.p, (size_t)zx * 8));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_colranks_csc_dense(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                                int32_t g, int32_t n, int ties, int is_signed, double* R_out) try {
  PH_CTX(ctx);
  PH_TRY(check_ties(ties, 2));
  PH_REQUIRE(g >= 0 && n >= 0 && Xp != nullptr, "colranks_csc_dense: bad arguments");
  PH_TRY(check_host_csc(Xp, Xi, g, n));
  if ((int64_t)g * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(R_out != nullptr, "colranks_csc_dense: null R_out");
  const int64_t zx = Xp[n];
  DevBuf dXp, dXi, dXx, dR;
  PH_TRY(dXp.alloc((size_t)(n + 1) * 4));
  PH_TRY(dXi.alloc((size_t)zx * 4));
  PH_TRY(dXx.alloc((size_t)zx * 8));
  PH_TRY(dR.alloc((size_t)g * n * 8));
  PH_TRY(h2d(ctx, dXp.p, Xp, (size_t)(n + 1) * 4));
  PH_TRY(h2d(ctx, dXi.p, Xi, (size_t)zx * 4));
  PH_TRY(h2d(ctx, dXx.p, Xx, (size_t)zx * 8));
  // zeros tie: the dense ranks follow from the ranks of the stored values (kernels_rank.hip) unless a column stores
  // more values than the rank kernel takes in one pass -- then the column is densified and ranked as a dense one
  const int32_t max_nnz = host_max_col_nnz(Xp, n);
  if (max_nnz <= max_sparse_rank_column()) {
    DevBuf dRx;
    PH_TRY(dRx.alloc((size_t)(zx > 0 ? zx : 1) * 8));
    PH_TRY(launch_colranks_csc_dense_nz_f64(ctx, dXp.as<int32_t>(), dXi.as<int32_t>(), dXx.as<double>(), g, n, max_nnz, ties,
                                            is_signed, 1.0, dRx.as<double>(), dR.as<double>(), g, nullptr));
    PH_TRY(copy_home(ctx, R_out, dR.p, (size_t)g * n * 8));
    PH_HIP(hipStreamSynchronize(ctx->stream));   // (dRx is released behind this)
    return PLAIDHIP_OK;
  }
  PH_TRY(launch_colranks_csc_dense_f64(ctx, dXp.as<int32_t>(), dXi.as<int32_t>(), dXx.as<double>(), g, n, ties,
                                       is_signed, 1.0, dR.as<double>(), g, nullptr));
  PH_TRY(copy_home(ctx, R_out, dR.p, (size_t)g * n * 8));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_sing_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n,
                        const int32_t* Gp, const int32_t* Gi, int32_t m, double* S_out) try {
  PH_CTX(ctx);
  // rX = colranks(X, ties.method="min") / nrow(X) - 0.5 ; plaid(rX, normalize=FALSE)  (R/plaid.R:215-217)
  return run_sharded(&ctx, 1, 1, nullptr, nullptr, X, g, n, Gp, Gi, m, PLAIDHIP_STAT_MEAN, 0, 0.0, S_out);
} catch (...) { return plaidhip::on_exception(); }

// replaid.sing for a dgCMatrix X: colranks(X, ties.method = "min") ranks the zeros too (sparse branch without keep.zero,
// R/plaid.R:602-609), / nrow(X) - 0.5, plaid(normalize = FALSE) (:215-217).  The reference densifies X to rank it; here
// the CSC slots go to the device as they are, the dense min-ranks are built per panel of columns from the ranks of the
// stored values (zeros tie), and the rank crossprod runs on each panel (multi.cpp: shard_worker): neither the host nor
// the PCIe link sees a dense X.
int plaidhip_sing_csc(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t g, int32_t n,
                      const int32_t* Gp, const int32_t* Gi, int32_t m, double* S_out) try {
  PH_CTX(ctx);
  PH_REQUIRE(Xp != nullptr, "sing_csc: null Xp");
  return run_sharded(&ctx, 1, 1, Xp, Xi, Xx, g, n, Gp, Gi, m, PLAIDHIP_STAT_MEAN, 0, 0.0, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_ssgsea_dense(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n,
                          const int32_t* Gp, const int32_t* Gi, int32_t m, double alpha,
                          double* S_out) try {
  PH_CTX(ctx);
  // rX = colranks(X, ties="average")^(1+alpha) ; rX/max(rX) - 0.5 ; plaid(mean, normalize=TRUE)  (R/plaid.R:245-253)
  return run_sharded(&ctx, 1, 2, nullptr, nullptr, X, g, n, Gp, Gi, m, PLAIDHIP_STAT_MEAN, 1, alpha, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_ssgsea_csc(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                        int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                        double alpha, double* S_out) try {
  PH_CTX(ctx);
  PH_REQUIRE(Xp != nullptr, "ssgsea_csc: null Xp");
  // sparse branch: ranks of the non-zeros only, zeros stay 0 (R/plaid.R:600-601, 631-650); the "- 0.5" of
  // R/plaid.R:251 applies to the zeros too, which the (alpha, beta) epilogue covers
  return run_sharded(&ctx, 1, 2, Xp, Xi, Xx, g, n, Gp, Gi, m, PLAIDHIP_STAT_MEAN, 1, alpha, S_out);
} catch (...) { return plaidhip::on_exception(); }

}  // extern "C"

// ---- "next" rows: ucell / aucell / scse ---------------------------------------------------
namespace {

// uploads X (dense when Xp == nullptr, else CSC) and produces the dense average ranks on the device
struct RankedInput {
  DevBuf dX, dXp, dXi, dR, dRx, dsmall;
  double* R = nullptr;
  double* d_colmax = nullptr;
  double* d_gmax = nullptr;     // device scalar max(rX)
};

int dense_average_ranks(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                        int32_t g, int32_t n, RankedInput& ri) {
  PH_TRY(ri.dR.alloc((size_t)g * n * 8));
  PH_TRY(ri.dsmall.alloc(64 + (size_t)n * 8));
  ri.R = ri.dR.as<double>();
  ri.d_gmax = ri.dsmall.as<double>();
  ri.d_colmax = reinterpret_cast<double*>(ri.dsmall.as<char>() + 64);
  if (Xp == nullptr) {
    PH_TRY(ri.dX.alloc((size_t)g * n * 8));
    PH_TRY(h2d(ctx, ri.dX.p, X_or_x, (size_t)g * n * 8));
    PH_TRY(launch_colranks_dense_f64(ctx, ri.dX.as<double>(), g, g, n, PLAIDHIP_TIES_AVERAGE, 0, 1.0, ri.R, g,
                                     ri.d_colmax));
  } else {
    PH_TRY(check_host_csc(Xp, Xi, g, n));
    const int64_t zx = Xp[n];
    PH_TRY(ri.dXp.alloc((size_t)(n + 1) * 4));
    PH_TRY(ri.dXi.alloc((size_t)zx * 4));
    PH_TRY(ri.dX.alloc((size_t)zx * 8));
    PH_TRY(h2d(ctx, ri.dXp.p, Xp, (size_t)(n + 1) * 4));
    PH_TRY(h2d(ctx, ri.dXi.p, Xi, (size_t)zx * 4));
    PH_TRY(h2d(ctx, ri.dX.p, X_or_x, (size_t)zx * 8));
    // zeros tie: dense ranks from the ranks of the stored values (any nrow(X)) unless a column stores too many
    const int32_t max_nnz = host_max_col_nnz(Xp, n);
    if (max_nnz <= max_sparse_rank_column()) {
      PH_TRY(ri.dRx.alloc((size_t)(zx > 0 ? zx : 1) * 8));
      PH_TRY(launch_colranks_csc_dense_nz_f64(ctx, ri.dXp.as<int32_t>(), ri.dXi.as<int32_t>(), ri.dX.as<double>(), g, n, max_nnz,
                                              PLAIDHIP_TIES_AVERAGE, 0, 1.0, ri.dRx.as<double>(), ri.R, g, ri.d_colmax));
    } else {
      PH_TRY(launch_colranks_csc_dense_f64(ctx, ri.dXp.as<int32_t>(), ri.dXi.as<int32_t>(), ri.dX.as<double>(), g, n,
                                           PLAIDHIP_TIES_AVERAGE, 0, 1.0, ri.R, g, ri.d_colmax));
    }
  }
  PH_TRY(launch_max(ctx, ri.d_colmax, n, ri.d_gmax));
  return PLAIDHIP_OK;
}

int plaid_on_device(plaidhip_ctx* ctx, plaidhip_geneset* gs, const double* dX, int32_t g, int32_t n, int32_t m,
                    int stat, int normalize, double* dS, DevBuf& dsmall, int x_kind = PLAIDHIP_X_ANY) {
  PH_TRY(dsmall.alloc(64 + (size_t)n * 8));
  uint32_t* d_flags = dsmall.as<uint32_t>();
  double* d_red = reinterpret_cast<double*>(dsmall.as<char>() + 16);
  double* d_med = reinterpret_cast<double*>(dsmall.as<char>() + 64);
  PH_HIP(hipMemsetAsync(d_flags, 0, 16, ctx->stream));
  PH_TRY(launch_spmm_dense_f64(ctx, gs, dX, g, n, stat, 1.0, nullptr, 0.0, dS, m, d_flags, x_kind));
  if (normalize) PH_TRY(normalize_on_device(ctx, dS, m, n, PLAIDHIP_IGNORE_ZERO_AUTO, d_flags, true, d_med, d_red));
  return PLAIDHIP_OK;
}

}  // namespace

extern "C" {

int plaidhip_ucell(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                   int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                   const double* k_full, double rmax, double* S_out) try {
  PH_CTX(ctx);
  PH_TRY(check_host_common(Gp, g, n, m));
  PH_REQUIRE(rmax > 0, "ucell: rmax must be positive");
  if ((int64_t)m * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(X_or_x && S_out && k_full, "ucell: null X/S_out/k_full");
  GenesetHolder gh;
  PH_TRY(acquire_geneset(ctx, g, m, Gp, Gi, &gh.gs));
  RankedInput ri;
  PH_TRY(dense_average_ranks(ctx, Xp, Xi, X_or_x, g, n, ri));
  PH_TRY(launch_map(ctx, ri.R, (int64_t)g * n, 0, rmax + 1.0, ri.d_gmax));            // R/plaid.R:278
  DevBuf dS, dsmall, dadd;
  PH_TRY(dS.alloc((size_t)m * n * 8));
  // pmin(max(rX) - rX, rmax + 1) of average ranks: still half-integers in [0, nrow(X)] when rmax + 1 is one
  const double cap2 = 2.0 * (rmax + 1.0);
  const int xk = (cap2 == std::floor(cap2) && cap2 < 65536.0) ? PLAIDHIP_X_RANKS : PLAIDHIP_X_ANY;
  PH_TRY(plaid_on_device(ctx, gh.gs, ri.R, g, n, m, PLAIDHIP_STAT_MEAN, 1, dS.as<double>(), dsmall, xk));   // :279
  std::vector<double> add(m);
  for (int32_t j = 0; j < m; ++j) add[j] = 1.0 + (k_full[j] + 1.0) / (2.0 * rmax);   // :280
  PH_TRY(dadd.alloc((size_t)m * 8));
  PH_TRY(h2d(ctx, dadd.p, add.data(), (size_t)m * 8));
  PH_TRY(launch_affine(ctx, dS.as<double>(), m, m, n, -1.0 / rmax, nullptr, 1.0, dadd.as<double>(), 0.0));
  PH_TRY(copy_home(ctx, S_out, dS.p, (size_t)m * n * 8));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_aucell(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                    int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                    double auc_max_rank, double* S_out) try {
  PH_CTX(ctx);
  PH_TRY(check_host_common(Gp, g, n, m));
  PH_REQUIRE(auc_max_rank > 0, "aucell: aucMaxRank must be positive");
  if ((int64_t)m * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(X_or_x && S_out, "aucell: null X/S_out");
  GenesetHolder gh;
  PH_TRY(acquire_geneset(ctx, g, m, Gp, Gi, &gh.gs));
  RankedInput ri;
  PH_TRY(dense_average_ranks(ctx, Xp, Xi, X_or_x, g, n, ri));
  PH_TRY(launch_map(ctx, ri.R, (int64_t)g * n, 1, auc_max_rank, ri.d_gmax));         // R/plaid.R:306
  DevBuf dS, dsmall;
  PH_TRY(dS.alloc((size_t)m * n * 8));
  PH_TRY(plaid_on_device(ctx, gh.gs, ri.R, g, n, m, PLAIDHIP_STAT_MEAN, 1, dS.as<double>(), dsmall));   // :307
  PH_TRY(copy_home(ctx, S_out, dS.p, (size_t)m * n * 8));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_scse(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                  int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m,
                  int remove_log2, int score_mean, double* S_out, int* removed_log2) try {
  PH_CTX(ctx);
  if (removed_log2 != nullptr) *removed_log2 = remove_log2 > 0 ? 1 : 0;
  PH_TRY(check_host_common(Gp, g, n, m));
  if ((int64_t)m * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(X_or_x && S_out, "scse: null X/S_out");
  GenesetHolder gh;
  PH_TRY(acquire_geneset(ctx, g, m, Gp, Gi, &gh.gs));
  const bool sparse = Xp != nullptr;
  if (sparse) PH_TRY(check_host_csc(Xp, Xi, g, n));
  const int64_t nvals = sparse ? (int64_t)Xp[n] : (int64_t)g * n;
  DevBuf dX, dXp, dXi, dS, dsmall, dcol;
  PH_TRY(dX.alloc((size_t)nvals * 8));
  PH_TRY(h2d(ctx, dX.p, X_or_x, (size_t)nvals * 8));
  if (sparse) {
    PH_TRY(dXp.alloc((size_t)(n + 1) * 4));
    PH_TRY(dXi.alloc((size_t)nvals * 4));
    PH_TRY(h2d(ctx, dXp.p, Xp, (size_t)(n + 1) * 4));
    PH_TRY(h2d(ctx, dXi.p, Xi, (size_t)nvals * 4));
  }
  PH_TRY(dcol.alloc(64 + (size_t)n * 8));
  double* d_mm = dcol.as<double>();
  double* d_colsum = reinterpret_cast<double*>(dcol.as<char>() + 64);
  if (remove_log2 < 0) {                                        // R/plaid.R:160-161: decided and applied on the device
    PH_TRY(launch_minmax(ctx, dX.as<double>(), nvals, d_mm));
    const bool implicit_zeros = sparse && nvals < (int64_t)g * n;   // they take part in min / max
    PH_TRY(launch_map(ctx, dX.as<double>(), nvals, sparse ? 5 : 4, implicit_zeros ? 1.0 : 0.0, d_mm));   // :163-171
  } else if (remove_log2) {
    PH_TRY(launch_map(ctx, dX.as<double>(), nvals, sparse ? 3 : 2, 0.0, nullptr));                       // :163-171
  }
  PH_TRY(launch_col_abs_sums(ctx, dX.as<double>(), g, g, sparse ? dXp.as<int32_t>() : nullptr, n, d_colsum));
  PH_TRY(dS.alloc((size_t)m * n * 8));
  const int stat = score_mean ? PLAIDHIP_STAT_MEAN : PLAIDHIP_STAT_SUM;
  if (sparse) {
    PH_TRY(launch_spmm_csc_f64(ctx, gh.gs, dXp.as<int32_t>(), dXi.as<int32_t>(), dX.as<double>(), n, nvals, stat, 1.0,
                               nullptr, 0.0, dS.as<double>(), m, nullptr));
  } else {
    PH_TRY(launch_spmm_dense_f64(ctx, gh.gs, dX.as<double>(), g, n, stat, 1.0, nullptr, 0.0, dS.as<double>(), m,
                                 nullptr));
  }
  // mean: sX / (colMeans|X| + 1e-8) (:176-177); sum: sX / (colSums|X| + 1e-8) * 100 (:181-182)
  PH_TRY(launch_affine(ctx, dS.as<double>(), m, m, n, score_mean ? 1.0 : 100.0, d_colsum,
                       score_mean ? 1.0 / (double)g : 1.0, nullptr, 0.0));
  PH_TRY(copy_home(ctx, S_out, dS.p, (size_t)m * n * 8));
  double mm[2] = {0.0, 0.0};
  if (remove_log2 < 0 && removed_log2 != nullptr)
    PH_HIP(hipMemcpyAsync(mm, d_mm, 16, hipMemcpyDeviceToHost, ctx->stream));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  if (remove_log2 < 0 && removed_log2 != nullptr) {   // what map_kernel decided from the same two numbers (:160-161)
    double mn = mm[0], mx = mm[1];
    if (sparse && nvals < (int64_t)g * n) { mn = mn < 0.0 ? mn : 0.0; mx = mx > 0.0 ? mx : 0.0; }
    *removed_log2 = (mn == 0.0 && mx < 20.0) ? 1 : 0;
  }
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

// Row-wise two-group sums / sums of squared deviations on device pointers: the pieces of plaid.test that a sample-sharded
// caller all-reduces between (plaid_amd/sharded.py: sharded_plaid_test).  A: rows x n column-major with leading dimension
// ld; y: 0 / 1 per column; sums / ssd: [2][rows] (group 0, group 1).
int plaidhip_dev_row_group_sums(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n, const int32_t* y,
                                double* sums) try {
  PH_CTX(ctx);
  PH_REQUIRE(rows >= 0 && n >= 0 && ld >= rows, "row_group_sums: bad shape rows=%d n=%d ld=%lld", rows, n, (long long)ld);
  if (rows == 0) return PLAIDHIP_OK;
  PH_REQUIRE(sums && (n == 0 || (A && y)), "row_group_sums: null argument");
  PH_TRY(ensure_workspace(ctx, (size_t)row_group_ws_doubles(rows, n) * 8));
  return launch_row_group_moments(ctx, A, ld, rows, n, y, 1, 1, sums, nullptr, static_cast<double*>(ctx->ws));
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_dev_row_group_ssd(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n, const int32_t* y,
                               const double* mean, double* ssd) try {
  PH_CTX(ctx);
  PH_REQUIRE(rows >= 0 && n >= 0 && ld >= rows, "row_group_ssd: bad shape rows=%d n=%d ld=%lld", rows, n, (long long)ld);
  if (rows == 0) return PLAIDHIP_OK;
  PH_REQUIRE(ssd && mean && (n == 0 || (A && y)), "row_group_ssd: null argument");
  PH_TRY(ensure_workspace(ctx, (size_t)row_group_ws_doubles(rows, n) * 8));
  return launch_row_group_ssd(ctx, A, ld, rows, n, y, mean, ssd, static_cast<double*>(ctx->ws));
} catch (...) { return plaidhip::on_exception(); }

// The host half of plaid.test (R/plaid.R:410-474): p-values, effect sizes, meta-p and FDR from the reduced statistics.
// Shared by plaidhip_plaid_test and by sample-sharded callers, which all-reduce the statistics first.
int plaidhip_plaid_test_finish(int32_t g, int32_t m, const int32_t* Gp, const double* T, double tot1, double tot2,
                               const double* SM, int64_t n0, int64_t n1, int tests, int metap_method, double* out) try {
  PH_REQUIRE(g >= 0 && m >= 0 && (m == 0 || (Gp && out)), "plaid_test_finish: null Gp / out");
  PH_REQUIRE((tests & 7) != 0 && (tests & ~7) == 0, "plaid_test: tests is a bit mask of 1 (one), 2 (two), 4 (lm)");
  PH_REQUIRE(metap_method == 0 || metap_method == 1, "Invalid method: %d", metap_method);      // R/plaid.R:533
  PH_REQUIRE(m == 0 || !(tests & 3) || T, "plaid_test_finish: null T");
  PH_REQUIRE(m == 0 || !(tests & 4) || SM, "plaid_test_finish: null SM");
  if (m == 0) return PLAIDHIP_OK;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  double* o_fc = out;
  double* o_p1 = out + (size_t)m;
  double* o_p2 = out + 2 * (size_t)m;
  double* o_p3 = out + 3 * (size_t)m;
  double* o_pm = out + 4 * (size_t)m;
  double* o_q = out + 5 * (size_t)m;
  for (int32_t j = 0; j < m; ++j) {
    const double k = (double)(Gp[j + 1] - Gp[j]);
    double eff = 0.0, pv[3];
    int np = 0;
    o_p1[j] = o_p2[j] = o_p3[j] = nan;
    if (tests & 1) {
      double mean1;
      o_p1[j] = clamp_p(onesample_p(k, T[j], T[(size_t)m + j], &mean1));
      eff += mean1;
      pv[np++] = o_p1[j];
    }
    if (tests & 2) {
      double diff;
      o_p2[j] = clamp_p(twosample_p((double)g, k, T[j], T[(size_t)m + j], tot1, tot2, &diff));
      eff += diff;
      pv[np++] = o_p2[j];
    }
    if (tests & 4) {
      const double m0 = SM[j], m1 = SM[(size_t)m + j];
      o_p3[j] = clamp_p(welch_p(m0, m1, SM[2 * (size_t)m + j], SM[3 * (size_t)m + j], (double)n0, (double)n1));
      eff += m1 - m0;                                                                           // :431
      pv[np++] = o_p3[j];
    }
    o_fc[j] = eff / np;                                                                         // rowMeans(F), :453
    o_pm[j] = np > 1 ? combine_p(pv, np, metap_method) : pv[0];                                 // :455-460
  }
  p_adjust_fdr(o_pm, m, o_q);                                                                   // :463
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_plaid_test(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n, const int32_t* y,
                        const int32_t* Gp, const int32_t* Gi, int32_t m, const double* gsetX, int tests,
                        int metap_method, double* out) try {
  PH_CTX(ctx);
  PH_TRY(check_host_common(Gp, g, n, m));
  PH_REQUIRE(m == 0 || out, "plaid_test: null out");
  PH_REQUIRE(n == 0 || (X && y), "plaid_test: null X / y");
  PH_REQUIRE((tests & 7) != 0 && (tests & ~7) == 0, "plaid_test: tests is a bit mask of 1 (one), 2 (two), 4 (lm)");
  PH_REQUIRE(metap_method == 0 || metap_method == 1, "Invalid method: %d", metap_method);      // R/plaid.R:533
  int64_t n0 = 0, n1 = 0;
  for (int32_t c = 0; c < n; ++c) {
    PH_REQUIRE(y[c] == 0 || y[c] == 1, "elements of y must be 0 or 1");                        // R/plaid.R:394
    if (y[c]) ++n1; else ++n0;
  }
  if (m == 0) return PLAIDHIP_OK;
  GenesetHolder gh;
  PH_TRY(acquire_geneset(ctx, g, m, Gp, Gi, &gh.gs));
  const int64_t ldg = even_ld(g);
  DevBuf dX, dy, dmean, dF, dT, dws, dS, dsm, dsmall;
  PH_TRY(dX.alloc((size_t)ldg * n * 8));
  PH_TRY(dy.alloc((size_t)n * 4));
  PH_TRY(dmean.alloc((size_t)g * 2 * 8));
  PH_TRY(dF.alloc((size_t)ldg * 2 * 8));
  PH_TRY(dT.alloc((size_t)m * 2 * 8));
  const int64_t wsd = std::max(row_group_ws_doubles(g, n), row_group_ws_doubles(m, n));
  PH_TRY(dws.alloc((size_t)wsd * 8));
  PH_TRY(h2d_cols(ctx, dX.p, ldg, X, g, n));
  PH_TRY(h2d(ctx, dy.p, y, (size_t)n * 4));
  // fc = rowMeans(X[, y == 1]) - rowMeans(X[, y == 0])   (R/plaid.R:407-409); Gt fc and Gt fc^2 (:478-479)
  PH_TRY(launch_row_group_moments(ctx, dX.as<double>(), ldg, g, n, dy.as<int32_t>(), n0, n1, dmean.as<double>(), nullptr,
                                  dws.as<double>()));
  PH_HIP(hipMemsetAsync(dF.p, 0, (size_t)ldg * 2 * 8, ctx->stream));
  PH_TRY(launch_fold_change(ctx, dmean.as<double>(), g, ldg, dF.as<double>()));
  PH_TRY(launch_spmm_dense_f64(ctx, gh.gs, dF.as<double>(), ldg, 2, PLAIDHIP_STAT_SUM, 1.0, nullptr, 0.0, dT.as<double>(),
                               m, nullptr));
  std::vector<double> T((size_t)m * 2), F((size_t)ldg * 2), SM;
  PH_HIP(hipMemcpyAsync(T.data(), dT.p, (size_t)m * 2 * 8, hipMemcpyDeviceToHost, ctx->stream));
  PH_HIP(hipMemcpyAsync(F.data(), dF.p, (size_t)ldg * 2 * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (tests & 4) {
    // scores stay on the device: given (uploaded) or plaid(X, G) computed here (R/plaid.R:424-427)
    PH_TRY(dS.alloc((size_t)m * n * 8));
    if (gsetX != nullptr) {
      PH_TRY(h2d(ctx, dS.p, gsetX, (size_t)m * n * 8));
    } else {
      PH_TRY(plaid_on_device(ctx, gh.gs, dX.as<double>(), (int32_t)ldg, n, m, PLAIDHIP_STAT_MEAN, 1, dS.as<double>(), dsmall));
    }
    PH_TRY(dsm.alloc((size_t)m * 4 * 8));
    PH_TRY(launch_row_group_moments(ctx, dS.as<double>(), m, m, n, dy.as<int32_t>(), n0, n1, dsm.as<double>(),
                                    dsm.as<double>() + 2 * (size_t)m, dws.as<double>()));
    SM.resize((size_t)m * 4);
    PH_HIP(hipMemcpyAsync(SM.data(), dsm.p, (size_t)m * 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  PH_HIP(hipStreamSynchronize(ctx->stream));
  double tot1 = 0.0, tot2 = 0.0;
  for (int32_t i = 0; i < g; ++i) { tot1 += F[i]; tot2 += F[(size_t)ldg + i]; }
  return plaidhip_plaid_test_finish(g, m, Gp, T.data(), tot1, tot2, (tests & 4) ? SM.data() : nullptr, n0, n1, tests,
                                    metap_method, out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_gsva(plaidhip_ctx* ctx, const double* X, int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi,
                  int32_t m, double tau, int rowtf, double* S_out) try {
  PH_CTX(ctx);
  PH_TRY(check_host_common(Gp, g, n, m));
  if ((int64_t)m * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(X && S_out, "gsva: null X/S_out");
  PH_REQUIRE(rowtf == 0 || rowtf == 1, "Error: unknown row transform %d", rowtf);                   // R/plaid.R:348
  GenesetHolder gh;
  PH_TRY(acquire_geneset(ctx, g, m, Gp, Gi, &gh.gs));
  const int64_t ldg = even_ld(g);
  DevBuf dX, dR, dS, dy, dmom, dws, dsmall;
  PH_TRY(dX.alloc((size_t)ldg * n * 8));
  PH_TRY(dR.alloc((size_t)ldg * n * 8));
  PH_TRY(dS.alloc((size_t)m * n * 8));
  PH_TRY(dy.alloc((size_t)n * 4));
  PH_TRY(dmom.alloc((size_t)g * 4 * 8));
  PH_TRY(dws.alloc((size_t)row_group_ws_doubles(g, n) * 8));
  PH_TRY(dsmall.alloc(64 + (size_t)n * 16));
  uint32_t* d_flags = dsmall.as<uint32_t>();
  double* d_red = reinterpret_cast<double*>(dsmall.as<char>() + 16);
  double* d_med = reinterpret_cast<double*>(dsmall.as<char>() + 64);
  double* d_colmax = d_med + n;
  double* d_gmax = d_red + 2;
  PH_TRY(h2d_cols(ctx, dX.p, ldg, X, g, n));
  PH_HIP(hipMemsetAsync(dy.p, 0, (size_t)n * 4, ctx->stream));                         // one group: every sample
  if (rowtf == 0) {
    // zX = (X - rowMeans(X)) / (1e-8 + rowSds(X))                                     (R/plaid.R:341-343)
    PH_TRY(launch_row_group_moments(ctx, dX.as<double>(), ldg, g, n, dy.as<int32_t>(), n, 0, dmom.as<double>(),
                                    dmom.as<double>() + 2 * (size_t)g, dws.as<double>()));
    PH_TRY(launch_row_ztransform(ctx, dX.as<double>(), ldg, g, n, dmom.as<double>(), dmom.as<double>() + 2 * (size_t)g));
  } else {
    // zX = t(apply(X, 1, function(x) ecdf(x)(x)))  (:346): ecdf(x)(x_i) = #{x <= x_i} / n = rank(x, "max") / n per
    // gene.  Genes become columns (transpose), the column rank kernel ranks them, and the result goes back; the
    // factor 1/n is dropped because only the per-sample ORDER of zX is used afterwards (:352).
    PH_TRY(launch_transpose_f64(ctx, dX.as<double>(), ldg, g, n, dR.as<double>(), n));              // dR: n x g
    PH_TRY(launch_colranks_dense_f64(ctx, dR.as<double>(), n, n, g, PLAIDHIP_TIES_MAX, 0, 1.0, dX.as<double>(), n, nullptr));
    PH_TRY(launch_transpose_f64(ctx, dX.as<double>(), n, n, g, dR.as<double>(), ldg));              // dR: g x n
    PH_HIP(hipMemcpyAsync(dX.p, dR.p, (size_t)ldg * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  }
  // rX = colranks(zX, signed = TRUE, "average"); rX / max|rX|; sign * |rX|^(1 + tau)   (:352-358)
  //    = sign * rank^(1+tau) / max(rank^(1+tau)): the power is fused into the rank kernel, the division into the
  //    SpMM epilogue (alpha_div), by linearity of the mean statistic
  PH_TRY(launch_colranks_dense_f64(ctx, dX.as<double>(), ldg, g, n, PLAIDHIP_TIES_AVERAGE, 1, tau > 0.0 ? 1.0 + tau : 1.0,
                                   dR.as<double>(), ldg, d_colmax));
  PH_TRY(launch_max(ctx, d_colmax, n, d_gmax));
  PH_HIP(hipMemsetAsync(d_flags, 0, 16, ctx->stream));
  PH_TRY(launch_spmm_dense_f64(ctx, gh.gs, dR.as<double>(), ldg, n, PLAIDHIP_STAT_MEAN, 1.0, d_gmax, 0.0, dS.as<double>(),
                               m, d_flags, tau > 0.0 ? PLAIDHIP_X_ANY : PLAIDHIP_X_EXACT_F32));   // signed average ranks
  PH_TRY(normalize_on_device(ctx, dS.as<double>(), m, n, PLAIDHIP_IGNORE_ZERO_AUTO, d_flags, true, d_med, d_red));   // :360 plaid()
  PH_TRY(copy_home(ctx, S_out, dS.p, (size_t)m * n * 8));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

}  // extern "C"
