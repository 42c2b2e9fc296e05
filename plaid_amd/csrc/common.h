// Internal declarations shared by the C-ABI translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/plaidhip.h"

namespace plaidhip {

// LDS budget of one CU on gfx950: 160 KiB, one workgroup may own all of it.
constexpr int kLdsBytes = 160 * 1024;
// The LDS-resident column kernels keep one 8-byte entry per gene plus kPadSlots zero
// entries that padded index slots point at.
constexpr int kPadSlots = 32;
constexpr int kMaxLdsGenes = kLdsBytes / 8 - kPadSlots;  // 20448
// bitonic sort in LDS: 8-byte keys
constexpr int kMaxLdsKeys = kLdsBytes / 8;               // 20480
// pair kernel: 16-byte entries (two sample columns), 16 zero entries behind them
constexpr int kPadSlotsPair = 16;
constexpr int kMaxLdsGenesPair = kLdsBytes / 16 - kPadSlotsPair;   // 10224
constexpr int kMaxPairSlices = 8;
// scatter kernel (sparse X): fp64 accumulators of one chunk of gene sets in LDS
constexpr int kScatterTrash = 64;                      // accumulators behind a chunk that padded id slots add into
// threads per workgroup of the scatter kernel (512: two workgroups per CU, measured slower; the -D override is for the
// A/B builds of tools/: make variant NAME=b512 DEFS=-DPLAIDHIP_SCATTER_BLOCK=512)
#ifndef PLAIDHIP_SCATTER_BLOCK
#define PLAIDHIP_SCATTER_BLOCK 1024
#endif
constexpr int kScatterBlock = PLAIDHIP_SCATTER_BLOCK;
// further id segments (genes in more than 128 sets of a chunk) whose loads ride the scatter walk's static pipeline per
// wavefront and item; a wavefront with more of them fetches the rest group by group (A/B builds: -DPLAIDHIP_SCATTER_HE=8)
// groups of 16 first-segment loads a wavefront requests before it applies the first one: 2 (3, round 5's depth, puts 960
// requests into the CU's in-order memory pipe at the start of an item and the youngest wavefronts' first ids behind all of
// them: +2.5 % at 16,384 x 50,000, profiles/r06h_scatter_depth_ab.txt; A/B builds: -DPLAIDHIP_SCATTER_DEPTH=3)
#ifndef PLAIDHIP_SCATTER_DEPTH
#define PLAIDHIP_SCATTER_DEPTH 2
#endif
#ifndef PLAIDHIP_SCATTER_HE
#define PLAIDHIP_SCATTER_HE 12
#endif
// sets per chunk: 17 passes of the workgroup's 1,024 threads over the accumulators (136 KiB of the 160; the chunk epilogue
// keeps one 16-byte factor pair per pass in registers, and 20 of them left none for anything else: 50,000 sets are three
// chunks either way)
constexpr int kScatterChunk = 17 * kScatterBlock;

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
// Every `int plaidhip_*` entry point is a function-try-block that ends here: a C++ exception (std::bad_alloc from a host-side
// plan, std::system_error from a thread the system refuses) becomes an error code and a message -- it never unwinds into
// the caller's C stack (R's, ctypes').  Exceptions inside the library's own worker threads are not covered by this.
inline int on_exception() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    set_error("out of host memory");
    return PLAIDHIP_ENOMEM;
  } catch (const std::exception& e) {
    set_error("unexpected C++ exception: %s", e.what());
    return PLAIDHIP_EHIP;
  } catch (...) {
    set_error("unexpected C++ exception");
    return PLAIDHIP_EHIP;
  }
}

#define PH_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) return ::plaidhip::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

#define PH_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      ::plaidhip::set_error(__VA_ARGS__);     \
      return PLAIDHIP_EINVAL;                 \
    }                                         \
  } while (0)

}  // namespace plaidhip

struct plaidhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // growable scratch
  void* ws = nullptr;
  size_t ws_bytes = 0;
  int num_cu = 256;
  // prepared gene-set collections of the host-level entry points (api.cpp: acquire_geneset), most recent last
  struct cached_geneset { uint64_t hash, hash2; int32_t g, m; struct plaidhip_geneset* gs; };
  std::vector<cached_geneset> gs_cache;
  int precision = 0;   // PLAIDHIP_PRECISION_*: 0 fp64 throughout (default), 1 fp32 operand staging in the dense SpMM
  // plaidhip_set_option (include/plaidhip.h: enum plaidhip_option)
  int opt_dense_kernel = 0;    // 0 auto | 1 one-column | 2 pair wherever it applies | 3 dense bf16x3 GEMM on MFMA
  int opt_sparse_kernel = 0;   // 0 auto | 1 scatter | 2 gather
  int opt_nt_store = -1;       // -1 auto | 0 | 1
  int opt_ranks_f32 = 2;       // rank inputs: 0 fp64 kernels | 1 fp32 staging | 2 u16 staging, integer sums (all exact)
  int opt_rank_kernel = 0;     // 0 auto | 1 sorting network | 2 bucket ranker
  int opt_scatter_fixed = 1;   // scatter kernel: u64 fixed-point accumulators for inputs declared bounded (rank weights)
  int opt_scatter_order = 1;   // scatter kernel: 0 (column, chunk) | 1 (chunk, column) item order
  int opt_fused_medians = 0;   // medians selected inside the sparse crossprod: 0 by size (>= 1e9 scores) | 1 whenever possible | 2 never
  double* d_sel = nullptr;        // {0 or -1, max, smallest > 0, largest column sum or +inf} of the stored values of a sparse X (scatter kernel's choice of accumulators)
                                  // (a 128-byte block: d_spec at byte 64; bytes 96..127 hold the EMPTY median bracket {0, -1, 0, 0} a
                                  //  calibration launch of the dense fused crossprod classifies against, written once at plaidhip_init)
  uint32_t* d_spec = nullptr;     // speculative launches (u16 quad kernel): [0] generation that saw a non-rank, [1..3] its private flag words
  uint32_t spec_gen = 0;          // generation of the last speculative launch (host side)
  // medians selected inside the last sparse crossprod launch (launch_spmm_csc_fused_f64): what plaidhip_dev_col_medians_resume
  // needs to finish them.  The scratch holds, per sample column: the predicted mean, the status word, per (chunk, wavefront)
  // four counts and a slice of candidate scores.
  void* fmed_buf = nullptr;
  size_t fmed_bytes = 0;
  uint64_t fmed_gen = 0;          // generation counter of the fused launches: the token a caller hands back to resume
  struct fused_medians {
    bool valid = false;
    uint64_t token = 0;           // generation of the launch that left this state (0: none)
    const double* S = nullptr;
    int64_t lds = 0;
    int32_t m = 0, n = 0, nslice = 0, capc = 0;
    double* pred = nullptr;
    double* cal = nullptr;
    uint32_t* cnt = nullptr;
    unsigned long long* cand = nullptr;
    int32_t* status = nullptr;
  } fmed;
  void* tie_scratch = nullptr;    // ties.method first / last / dense: two scratch columns per column (kernels_rank.hip)
  size_t tie_scratch_bytes = 0;
  void* rank_scratch = nullptr;   // value-partitioned ranking of columns beyond the LDS (kernels_rank.hip), grown on demand
  size_t rank_scratch_bytes = 0;
  int debug_fail_crossprod = 0;   // test hook (plaidhip_debug_sharded_on_one_device): this context's shard fails in the crossprod phase
  // pinned staging of the pipelined host uploads (multi.cpp): kFeeders feeder threads x 2 buffers, their streams
  static constexpr int kFeeders = 4;
  void* pin[kFeeders][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  size_t pin_bytes = 0;
  hipStream_t copy_stream[kFeeders] = {nullptr, nullptr, nullptr, nullptr};
  // device buffers of the host-level pipelines, kept between calls (an R session scores matrix after matrix of the
  // same shape; hipMalloc + hipFree of gigabytes per call cost milliseconds and a device synchronisation)
  static constexpr int kHostBufs = 6;
  void* hbuf[kHostBufs] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t hbuf_bytes[kHostBufs] = {0, 0, 0, 0, 0, 0};
};

// Prepared membership (built by geneset.cpp, see the header comment there).
//   tile_idx : u16 gene ids, layout [chunk][lane 0..63][8]  (a chunk = 8 gather steps; one
//              16-byte load per lane per chunk, 1 KiB per wave, coalesced).  Idle slots hold
//              g + r (r < kPadSlots) and read a zero entry behind the column in LDS.
//   wave_chunk_off[w] .. [w+1] : the contiguous chunk stream of wavefront w of the workgroup
//   wave_tile_off[w]  .. [w+1] : its tiles, as entries of wtile_end / wtile_id
//   wtile_end[k]      : absolute chunk index one past tile k's last chunk
//   meta_j/meta_w/meta_k[k*64 + lane] : per lane of wave-stream tile k: set id (or -1),
//                       1/(1e-8 + size) and size -- everything the epilogue needs, one
//                       coalesced load each, fetched a tile ahead
// one gene slice (g0 .. g0+gs) of the prepared membership; g <= kMaxLdsGenes needs one slice
struct plaidhip_slice {
  int32_t g0 = 0, gs = 0;
  int32_t waves = 0;           // wavefronts per workgroup the plan was built for
  int64_t chunks = 0;          // 8-step chunks over all tiles of this slice
  uint16_t* d_tile_idx = nullptr;
  int32_t* d_wave_chunk_off = nullptr;
  int32_t* d_wave_tile_off = nullptr;
  int32_t* d_wtile_end = nullptr;
  int32_t* d_meta_j = nullptr;
  double* d_meta_w = nullptr;
  double* d_meta_k = nullptr;
};

// pair plan: gene slices of <= kMaxLdsGenesPair genes; tiles, tile->wave assignment and the
// per-lane metadata are shared by all slices (see geneset.cpp)
struct plaidhip_pair_slice {
  int32_t g0 = 0, gs = 0;
  uint16_t* d_tile_idx = nullptr;
  int32_t* d_wave_chunk_off = nullptr;
  int32_t* d_wtile_end = nullptr;
};
// what the kernel reads per slice (device array, scalar loads)
struct plaidhip_pair_slice_dev {
  const uint16_t* tile_idx;
  const int32_t* wave_chunk_off;
  const int32_t* wtile_end;
  int32_t g0, gs;
};
struct plaidhip_pair_plan {
  int32_t waves = 0;
  int32_t ktiles = 0;                          // wave-stream tiles (all waves)
  int64_t chunks = 0;
  std::vector<plaidhip_pair_slice> slices;
  plaidhip_pair_slice_dev* d_slices = nullptr;
  // partial sums between gene slices: [workgroup][wave-stream tile + 1][lane] x {A, B}
  double* d_partial = nullptr;
  int32_t partial_wgs = 0;
  int32_t* d_wave_tile_off = nullptr;
  int32_t* d_meta_j = nullptr;
  double* d_meta_w = nullptr;
  double* d_meta_k = nullptr;
};

// Scatter plan (sparse X): G transposed, gene-major.  The sets are dealt to `nch` chunks of `ch` accumulator slots (the
// LDS accumulators of one chunk) in BLOCKS of kScatterBlock consecutive sets, block b to chunk b % nch at slots
// (b / nch) * kScatterBlock ..: a collection straight from gmt2mat() has its sets in decreasing size (R/gmt-utils.R:25), and
// chunks of consecutive sets gave the first chunk 72 % of all memberships -- two to three id segments per gene there, one
// fifth-full segment in the last chunk, 4.35 (value, chunk) pairs per stored value at config 3 instead of the 3.3 the
// interleaved deal needs (every chunk sees the whole range of set sizes).  A chunk pass of the 1,024 threads still writes
// 1,024 consecutive scores.  The sets of gene i inside chunk c are stored as whole segments
// of 128 u16 slot ids (a dword = two ids per lane; padding: a trash accumulator behind the chunk): segments seg[c*g + i] ..
// seg[c*g + i + 1] - 1 of d_ids.  (seg has nch*g + 1 entries, chunk-major, so the ranges of
// consecutive (chunk, gene) pairs are contiguous.)
#if defined(__HIPCC__)
#define PH_HD __host__ __device__
#else
#define PH_HD
#endif
PH_HD inline int32_t scatter_chunk_of(int32_t j, int32_t nch) { return (j / plaidhip::kScatterBlock) % nch; }
PH_HD inline int32_t scatter_slot_of(int32_t j, int32_t nch) {
  return ((j / plaidhip::kScatterBlock) / nch) * plaidhip::kScatterBlock + j % plaidhip::kScatterBlock;
}
PH_HD inline int32_t scatter_set_of(int32_t chunk, int32_t slot, int32_t nch) {
  return ((slot / plaidhip::kScatterBlock) * nch + chunk) * plaidhip::kScatterBlock + slot % plaidhip::kScatterBlock;
}
struct plaidhip_scatter_plan {
  int32_t ch = 0, nch = 0;
  int32_t kbits = 0;           // bits of the largest set size + 1: headroom of the fixed-point sums
  int64_t nseg = 0;
  int32_t* d_seg = nullptr;
  uint16_t* d_ids = nullptr;
  double* d_w = nullptr;   // per set 1/(1e-8 + size)
  double* d_k = nullptr;   // per set size
  double* d_kw = nullptr;  // {size x weight, weight} per set, m entries for STAT_MEAN then m for STAT_SUM: one 16-byte load per set in the scatter kernel's epilogue
  double* d_u = nullptr;   // per gene (1 / m) sum of the weights of its sets: g entries for STAT_MEAN, then g for STAT_SUM
  double kappa[2] = {0.0, 0.0};   // (1 / m) sum_j size_j weight_j per statistic
#ifdef PLAIDHIP_KEEP_HOST_PLANS
  std::vector<int32_t> h_seg;    // host-only tools build (tools/plan_probe): the uploaded plan, for its checker
  std::vector<uint16_t> h_ids;
#endif
};

struct plaidhip_geneset {
  plaidhip_ctx* ctx = nullptr;
  int32_t g = 0, m = 0;
  int64_t z = 0;
  int32_t tiles = 0;
  int64_t chunks = 0;          // total over slices
  std::vector<plaidhip_slice> slices;   // the column is consumed slice by slice when g > kMaxLdsGenes
  plaidhip_pair_plan pair;              // dense-X kernel: two columns per pass
  plaidhip_scatter_plan scatter;        // sparse-X kernel: nonzeros are scattered into per-set LDS accumulators
  bool rows_in_order = false;           // the sets came sorted by decreasing size: a tile's lanes are neighbouring rows of S
  // MFMA backend (opt-in, kernels_mfma.hip): the pattern is kept on the host and the dense bf16 sets x genes matrix
  // is built on first use
  std::vector<int32_t> h_Gp, h_Gi;
  void* d_dense_g = nullptr;
  int32_t dense_gk = 0;
};

namespace plaidhip {

int ensure_workspace(plaidhip_ctx* ctx, size_t bytes);
// RAII for the device buffers of one host-level call
struct DevBuf {
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) hipFree(p); }
  int alloc(size_t bytes);
  template <typename T> T* as() { return static_cast<T*>(p); }
};
// slot `k` of the context's persistent host-pipeline buffers, grown to at least `bytes` (contents undefined)
int ctx_buffer(plaidhip_ctx* ctx, int k, size_t bytes, void** out);
// host-level helpers shared by api.cpp and multi.cpp
int acquire_geneset(plaidhip_ctx* ctx, int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, plaidhip_geneset** out);
int check_host_csc(const int32_t* Xp, const int32_t* Xi, int32_t g, int32_t n);
int check_host_common(const void* G_p, int32_t g, int32_t n, int32_t m);
const char* last_error_cstr();
// the sample-sharded host pipeline behind plaid / replaid.sing / replaid.ssgsea (multi.cpp): ndev contexts, one per
// device; X dense (Xp == nullptr) or CSC; method 0 plaid, 1 sing, 2 ssgsea
int run_sharded(plaidhip_ctx* const* ctxs, int ndev, int method, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, int stat, int normalize, double alpha,
                double* S_out);
// A result's way home into the caller's pageable buffer (multi.cpp).  R hands over FRESH memory (allocMatrix -> malloc ->
// mmap): every page faults on its first write, inside the device-to-host copy -- 4.9 GB of scores took 309 ms instead of
// 92 (tools/ubench/d2h_fresh.cpp).  prepare() asks for transparent huge pages on the range (madvise; a hint, ignored where
// the system has them off) and starts threads that touch it chunk by chunk; copy() issues the chunk copies on the context's
// stream as their pages appear and returns when the last one is enqueued (the pageable copies themselves are synchronous
// in the runtime).  Small results are one plain copy.  The destination's previous contents are destroyed from prepare() on.
class HomeBuffer {
 public:
  HomeBuffer() = default;
  HomeBuffer(const HomeBuffer&) = delete;
  HomeBuffer& operator=(const HomeBuffer&) = delete;
  ~HomeBuffer() { finish(); }
  void prepare(void* dst, size_t bytes);
  int copy(plaidhip_ctx* ctx, const void* src_dev);   // (prepare() first; copies `bytes` of it)
  void finish();                                      // joins the touch threads (idempotent)
 private:
  struct State;
  State* st_ = nullptr;
};
// prepare + copy + finish
int copy_home(plaidhip_ctx* ctx, void* dst, const void* src_dev, size_t bytes);
// the other direction (multi.cpp): `cols` rows of row_bytes bytes from pageable host memory to device rows ldd_bytes apart
// (a flat array: row_bytes = ldd_bytes = 1, cols = bytes), through the context's pinned staging ring; stream-ordered for
// consumers on the context's stream, returns when the host side is done with `src`
int upload_host(plaidhip_ctx* ctx, void* dst, size_t ldd_bytes, const void* src, size_t row_bytes, int64_t cols);
// opt a kernel into the full 160 KiB of dynamic LDS, once per (kernel, device)
int allow_full_lds(plaidhip_ctx* ctx, const void* kernel, std::atomic<uint32_t>* done_mask);
#define PH_FULL_LDS(ctx, kernel)                                                          \
  do {                                                                                    \
    static std::atomic<uint32_t> mask_{0};   /* per call site; bit = device ordinal; host threads of several devices meet here */ \
    int rc_l_ = ::plaidhip::allow_full_lds((ctx), reinterpret_cast<const void*>(kernel), &mask_); \
    if (rc_l_ != PLAIDHIP_OK) return rc_l_;                                               \
  } while (0)
int spmm_block_for_genes(int32_t g);   // workgroup size of the column-resident SpMM kernel

// kernels_stats.hip / stats.cpp  (plaid.test)
int launch_row_group_ssd(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n, const int32_t* d_y,
                         const double* d_mean, double* d_ssd, double* ws);
int launch_row_group_moments(plaidhip_ctx* ctx, const double* A, int64_t ld, int32_t rows, int32_t n,
                             const int32_t* d_y, int64_t n0, int64_t n1, double* d_mean, double* d_ssd,
                             double* ws);
int launch_row_ztransform(plaidhip_ctx* ctx, double* A, int64_t ld, int32_t rows, int32_t n, const double* d_mean,
                          const double* d_ssd);
int launch_transpose_f64(plaidhip_ctx* ctx, const double* A, int64_t lda, int32_t rows, int32_t cols, double* B,
                         int64_t ldb);
int launch_fold_change(plaidhip_ctx* ctx, const double* d_mean, int32_t rows, int64_t ld2, double* d_F);
int64_t row_group_ws_doubles(int32_t rows, int32_t n);
double onesample_p(double k, double s1, double s2, double* mean_out);
double twosample_p(double g, double k, double s1, double s2, double tot1, double tot2, double* diff_out);
double welch_p(double m0, double m1, double ssd0, double ssd1, double n0, double n1);
double clamp_p(double p);
double combine_p(const double* p, int np, int method);
void p_adjust_fdr(const double* p, int64_t m, double* q);

// kernels_spmm.hip
int launch_spmm_dense_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const double* X,
                          int64_t ldx, int32_t n, int stat, double alpha, const double* alpha_div,
                          double beta, double* S, int64_t lds, uint32_t* flags,
                          int x_kind = 0);   // PLAIDHIP_X_*: what the caller knows about the values of X
// values of X as the internal callers know them (a compact exact staging is chosen from this, never a rounding one)
enum { PLAIDHIP_X_ANY = 0,        // arbitrary doubles
       PLAIDHIP_X_EXACT_F32 = 1,  // (half-)integers of magnitude <= 20,448, any sign (signed ranks): exact in fp32
       PLAIDHIP_X_RANKS = 2 };    // what colranks returns: half-integers in [0, nrow(X)]: 2x is a u16
int launch_spmm_mfma_f64(plaidhip_ctx* ctx, plaidhip_geneset* gs, const double* X, int64_t ldx, int32_t n, int stat,
                         double alpha, const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags);
int launch_spmm_scatter_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp,
                                const int32_t* Xi, const double* Xx, int32_t n, int stat, double alpha,
                                const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags,
                                bool auto_select, bool bounded = false, const double* xmax_dev = nullptr, double xmax_host = 0.0,
                                int64_t nnz = -1, const struct plaidhip_scatter_med* med = nullptr);
// what the MED form of the scatter kernel needs (medians selected while the scores are written)
struct plaidhip_scatter_med {
  const double* pred;
  const double* cal;
  unsigned long long* cand;
  uint32_t* cnt;
  int32_t capc;
};
// the sparse crossprod + everything normalize_medians can know by then; launch_col_medians_resume finishes the medians
int launch_spmm_csc_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                              int32_t n, int64_t nnz, int stat, double alpha, const double* alpha_div, double beta, double* S,
                              int64_t lds, uint32_t* flags, bool bounded, const double* xmax_dev, double xmax_host,
                              int64_t nnz_choice = -1 /* what picks scatter / gather when it is not nnz itself */);
// the dense crossprod (fp64 pair kernel) + everything normalize_medians can know by then (round 5); same resume call
int launch_spmm_dense_fused_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const double* X, int64_t ldx, int32_t n, int stat,
                                double alpha, const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags,
                                int x_kind = 0);
int launch_col_medians_resume(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n, int ignore_zero,
                              const uint32_t* flags, double* med, int64_t token = -1);
// {all values finite and >= 0 ? 0 : -1, max} of a device vector -> out[2] (kernels_norm.hip)
int launch_nonneg_range(plaidhip_ctx* ctx, const double* Xx, const int32_t* Xp, int32_t n, int64_t nnz_hint, double* out);
int launch_colsum_max(plaidhip_ctx* ctx, const double* Xx, const int32_t* Xp, int32_t n, double* out);
int launch_spmm_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp,
                        const int32_t* Xi, const double* Xx, int32_t n, int64_t nnz /* -1: unknown */, int stat, double alpha,
                        const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags,
                        // bounded: every stored value lies in [0, xmax] (rank weights; xmax = max(rX) on the device, or on
                        // the host when xmax_dev is null): the scatter kernel may use exact fixed-point accumulators
                        bool bounded = false, const double* xmax_dev = nullptr, double xmax_host = 0.0);
// kernels_wspmm.hip: t(x) %*% y for a sparse x with arbitrary values (device CSC slots); y dense (Yp == nullptr) or CSC
int launch_crossprod_weighted_f64(plaidhip_ctx* ctx, const int32_t* Wp, const int32_t* Wi, const double* Wx, int32_t g,
                                  int32_t m, const double* Y, int64_t ldy, const int32_t* Yp, const int32_t* Yi,
                                  const double* Yx, int32_t n, double* S, int64_t lds);
#ifdef PLAIDHIP_DIAG
void debug_set_ablation(int mode, void* dbg);   // diagnostic kernel variants (tools/ build only, make diag)
void debug_set_rank_stamps(void* dbg);          // per-phase cycle stamps of the bucket rank kernel
void debug_set_median_stamps(void* dbg);        // per-phase cycle stamps of the wave-per-column median kernel
#endif
// kernels_rank.hip
int launch_colranks_dense_f64(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t g, int32_t n,
                              int ties, int is_signed, double power, double* R, int64_t ldr,
                              double* colmax);
int launch_colranks_csc_f64(plaidhip_ctx* ctx, const int32_t* Xp, const double* Xx, int32_t n, int32_t max_col_nnz,
                            int ties, int is_signed, double power, double* Rx, double* colmax);
int32_t host_max_col_nnz(const int32_t* Xp, int32_t n);   // longest column of a host-side CSC pointer array
int launch_colranks_csc_dense_f64(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx,
                                  int32_t g, int32_t n, int ties, int is_signed, double power, double* R,
                                  int64_t ldr, double* colmax);
// dense ranks of CSC columns from the ranks of their stored values (zeros tie): any number of rows, every column at most
// max_sparse_rank_column() stored values; Rx_scratch: Xp[n] doubles
int launch_colranks_csc_dense_nz_f64(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t g,
                                     int32_t n, int32_t max_col_nnz, int ties, int is_signed, double power,
                                     double* Rx_scratch, double* R, int64_t ldr, double* colmax);
int max_sparse_rank_column();
// kernels_norm.hip
int launch_minflags(plaidhip_ctx* ctx, const double* S, int64_t count, uint32_t* flags);
int launch_col_medians(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n,
                       int ignore_zero, const uint32_t* flags, double* med,
                       // non-null (streaming kernel, m > 6,144 only): columns with status[c] != 0 already have their median
                       const int32_t* status = nullptr);
// medians selected inside the sparse crossprod launch (kernels_norm.hip / kernels_spmm.hip: MED)
int launch_colmean_predict(plaidhip_ctx* ctx, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t n, const double* u,
                           double alpha, const double* alpha_div, double beta_kappa, double* pred);
int launch_median_calibrate(plaidhip_ctx* ctx, const double* medK, const double* pred, int32_t K, const uint32_t* flagsK,
                            double* cal);
int launch_median_select(plaidhip_ctx* ctx, const unsigned long long* cand, const uint32_t* cnt, int32_t n, int32_t nslice,
                         int32_t capc, int32_t m, const double* cal, int ignore_zero, const uint32_t* flags, double* med,
                         int32_t* status);
int launch_sum(plaidhip_ctx* ctx, const double* v, int64_t count, double* out);
int launch_max(plaidhip_ctx* ctx, const double* v, int64_t count, double* out);
int launch_shift_columns(plaidhip_ctx* ctx, double* S, int64_t lds, int32_t m, int32_t n,
                         const double* med, double add, const double* red);
int launch_shift_columns_cast_f32(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n, const double* med,
                                  double add, const double* red, float* out, int64_t ldo);

// element-wise / column helpers (replaid.ucell / aucell / scse)
int launch_map(plaidhip_ctx* ctx, double* v, int64_t count, int op, double p0, const double* scalar);
int launch_col_abs_sums(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t len, const int32_t* Xp,
                        int32_t n, double* out);
int launch_affine(plaidhip_ctx* ctx, double* S, int64_t lds, int32_t m, int32_t n, double mul,
                  const double* col_div, double div_scale, const double* row_add, double add);
int launch_minmax(plaidhip_ctx* ctx, const double* v, int64_t count, double* out);

}  // namespace plaidhip
