// Host-buffer pipelines for one or several GPUs of a node, driven from ONE host process.
//
// The reference is a single R process (no process-per-GPU runtime to lean on), so the multi-GPU form of the host entry
// points is a thread per device inside the library: the sample columns are cut into contiguous shards
// (plaidhip_shard_bounds: ceil(n / ndev) columns each, R's column-major layout makes a shard one byte range of X and of
// S), every device moves its shard over its own PCIe link, and the three scalars that couple the samples -- max(rX)
// (R/plaid.R:251), min(x) == 0 (R/plaid.R:556-557) and mean(medx) (R/plaid.R:572) -- are combined on the host between
// the phases.  No RCCL: nothing but those scalars crosses between devices.  (One process per GPU over RCCL is the
// other form, plaid_amd/sharded.py.)  The single-device entry points run the same code with one shard.
//
// Uploads are pipelined: R hands over pageable memory, which the HIP runtime copies at ~21 GB/s; staged through
// pinned buffers by a few feeder threads (memcpy at ~75 GB/s with four threads, tools/ubench/pcie.cpp) the DMA
// runs at the link rate (~57 GB/s) and the kernels of a column panel start as soon as the panel has landed.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "common.h"

using namespace plaidhip;

#define PH_TRY(expr)                      \
  do {                                    \
    int rc_ = (expr);                     \
    if (rc_ != PLAIDHIP_OK) return rc_;   \
  } while (0)

extern "C" int plaidhip_shard_bounds(int64_t n, int ndev, int k, int64_t* lo, int64_t* hi) try {
  PH_REQUIRE(n >= 0 && ndev > 0 && k >= 0 && k < ndev && lo && hi, "shard_bounds: bad arguments n=%lld ndev=%d k=%d",
             (long long)n, ndev, k);
  const int64_t per = (n + ndev - 1) / ndev;
  *lo = std::min(n, (int64_t)k * per);
  *hi = std::min(n, *lo + per);
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

namespace plaidhip {

// ---- HomeBuffer ---------------------------------------------------------------------------------------------------------
struct HomeBuffer::State {
  char* dst = nullptr;
  size_t bytes = 0, nchunk = 0;
  std::vector<std::atomic<int>> done;
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  explicit State(size_t n) : done(n) {}
};
namespace {
constexpr size_t kHomeChunk = (size_t)64 << 20;
constexpr size_t kHomeMin = (size_t)16 << 20;   // below this one plain copy (a few thousand page faults)
constexpr int kHomeThreads = 8;                 // touching 4.9 GB of huge pages: 29 ms with 8 threads, 68 with 4 (ubench)
}  // namespace

void HomeBuffer::prepare(void* dst, size_t bytes) {
  finish();
  const size_t nchunk = bytes >= kHomeMin ? (bytes + kHomeChunk - 1) / kHomeChunk : 0;
  st_ = new State(nchunk);
  st_->dst = static_cast<char*>(dst);
  st_->bytes = bytes;
  st_->nchunk = nchunk;
  if (nchunk == 0) return;
  for (auto& d : st_->done) d.store(0, std::memory_order_relaxed);
  {
    const uintptr_t b = ((uintptr_t)dst + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)dst + bytes) & ~(uintptr_t)4095;
    if (e > b) (void)madvise(reinterpret_cast<void*>(b), e - b, MADV_HUGEPAGE);
  }
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = (int)std::max<size_t>(1, std::min<size_t>({(size_t)kHomeThreads, nchunk, hw > 1 ? hw / 2 : 1}));
  State* st = st_;
  auto spawn = [&](auto&& body) {   // (a thread the system refuses is one helper less, not an exception through a C ABI)
    try {
      st_->th.emplace_back(body);
    } catch (...) {
    }
  };
  for (int t = 0; t < nt; ++t)
    spawn([st] {
      for (;;) {
        const size_t k = st->next.fetch_add(1);
        if (k >= st->nchunk) return;
        const size_t b = k * kHomeChunk, e = std::min(st->bytes, b + kHomeChunk);
        // one write per 4 KiB page, at the page's first byte inside the buffer
        for (size_t o = b; o < e; o = ((((uintptr_t)st->dst + o) | 4095) + 1) - (uintptr_t)st->dst)
          *reinterpret_cast<volatile char*>(st->dst + o) = 0;
        st->done[k].store(1, std::memory_order_release);
      }
    });
  if (st_->th.empty()) st_->nchunk = 0;   // no helper at all: one plain copy (copy() below)
}

int HomeBuffer::copy(plaidhip_ctx* ctx, const void* src_dev) {
  PH_REQUIRE(st_ != nullptr, "HomeBuffer::copy before prepare");
  if (st_->bytes == 0) return PLAIDHIP_OK;
  if (st_->nchunk == 0) {
    PH_HIP(hipMemcpyAsync(st_->dst, src_dev, st_->bytes, hipMemcpyDeviceToHost, ctx->stream));
    return PLAIDHIP_OK;
  }
  for (size_t k = 0; k < st_->nchunk; ++k) {
    while (st_->done[k].load(std::memory_order_acquire) == 0) std::this_thread::yield();
    const size_t b = k * kHomeChunk, len = std::min(kHomeChunk, st_->bytes - b);
    PH_HIP(hipMemcpyAsync(st_->dst + b, static_cast<const char*>(src_dev) + b, len, hipMemcpyDeviceToHost, ctx->stream));
  }
  return PLAIDHIP_OK;
}

void HomeBuffer::finish() {
  if (st_ == nullptr) return;
  for (auto& t : st_->th) t.join();
  delete st_;
  st_ = nullptr;
}

int copy_home(plaidhip_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
  HomeBuffer hb;
  hb.prepare(dst, bytes);
  const int rc = hb.copy(ctx, src_dev);
  hb.finish();
  return rc;
}

}  // namespace plaidhip

namespace {

constexpr size_t kPanelBytes = (size_t)48 << 20;   // pinned staging buffer: 2 per feeder thread

int ensure_pinned(plaidhip_ctx* ctx) {
  if (ctx->pin_bytes >= kPanelBytes) return PLAIDHIP_OK;
  // whatever a partial failure leaves behind stays recorded in the context (null-checked here, freed by plaidhip_finalize)
  for (int t = 0; t < plaidhip_ctx::kFeeders; ++t) {
    for (int b = 0; b < 2; ++b)
      if (ctx->pin[t][b] == nullptr) PH_HIP(hipHostMalloc(&ctx->pin[t][b], kPanelBytes, hipHostMallocDefault));
    if (ctx->copy_stream[t] == nullptr) PH_HIP(hipStreamCreateWithFlags(&ctx->copy_stream[t], hipStreamNonBlocking));
  }
  ctx->pin_bytes = kPanelBytes;
  return PLAIDHIP_OK;
}

// Host (pageable) -> device copy of `rows x cols` column-major doubles (or of a flat byte array: rows = bytes per
// "column") with destination leading dimension ldd, pipelined through the context's pinned buffers.  `on_panel(c0, c1)`
// (may be empty) is called on the calling thread, in column order, after the compute stream has been made to wait for
// the panel's DMA: it enqueues whatever consumes columns [c0, c1).
int upload_pipelined(plaidhip_ctx* ctx, char* dst, size_t ldd_bytes, const char* src, size_t row_bytes, int64_t cols,
                     const std::function<int(int64_t, int64_t)>& on_panel) {
  if (cols == 0 || row_bytes == 0) return PLAIDHIP_OK;
  if (row_bytes * (size_t)cols < ((size_t)8 << 20) || 2 * ldd_bytes > kPanelBytes) {
    // small input, or columns so long that a panel of two (the least the pair kernel takes) overflows a staging
    // buffer: one plain copy
    if (ldd_bytes == row_bytes) {
      PH_HIP(hipMemcpyAsync(dst, src, row_bytes * (size_t)cols, hipMemcpyHostToDevice, ctx->stream));
    } else {
      PH_HIP(hipMemcpy2DAsync(dst, ldd_bytes, src, row_bytes, row_bytes, (size_t)cols, hipMemcpyHostToDevice, ctx->stream));
    }
    return on_panel ? on_panel(0, cols) : PLAIDHIP_OK;
  }
  PH_TRY(ensure_pinned(ctx));
  constexpr int T = plaidhip_ctx::kFeeders;
  const int64_t pcols = std::max<int64_t>(1, (int64_t)(kPanelBytes / ldd_bytes));
  // panel boundaries: the first round of panels (one per feeder) is an eighth of the size, the second a half -- the
  // bus starts after a fraction of a millisecond of staging instead of after a whole 48 MB memcpy (measured: 3.3 ms
  // of fill in front of the first DMA with equal panels)
  std::vector<int64_t> pb{0};
  for (int64_t r = 0; pb.back() < cols; ++r) {
    int64_t w = r < T ? pcols / 8 : (r < 2 * T ? pcols / 2 : pcols);
    w = std::max<int64_t>(2, w & ~(int64_t)1);   // even: the pair kernel takes two columns per pass
    pb.push_back(std::min(cols, pb.back() + w));
  }
  const int64_t npan = (int64_t)pb.size() - 1;
  struct Events {   // destroyed on every exit path
    std::vector<hipEvent_t> v;
    ~Events() { for (hipEvent_t e : v) if (e) hipEventDestroy(e); }
    hipEvent_t& operator[](size_t i) { return v[i]; }
  } done;
  done.v.assign((size_t)npan, nullptr);
  for (auto& e : done.v) PH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  std::vector<std::atomic<int>> ready((size_t)npan);
  for (auto& r : ready) r.store(0, std::memory_order_relaxed);
  std::atomic<int> failed{0};
  const int device = ctx->device;
  auto feeder = [&](int t) {
    if (hipSetDevice(device) != hipSuccess) { failed.store(1); }
    hipEvent_t freeb[2] = {nullptr, nullptr};
    for (int64_t p = t, it = 0; p < npan; p += T, ++it) {
      const int b = (int)(it & 1);
      const int64_t c0 = pb[(size_t)p], c1 = pb[(size_t)p + 1];
      bool ok = failed.load() == 0;
      if (ok && freeb[b] != nullptr) ok = hipEventSynchronize(freeb[b]) == hipSuccess;   // the DMA that last read this buffer
      if (ok) {
        char* stage = static_cast<char*>(ctx->pin[t][b]);
        if (ldd_bytes == row_bytes) {
          memcpy(stage, src + (size_t)c0 * row_bytes, (size_t)(c1 - c0) * row_bytes);
        } else {
          for (int64_t c = c0; c < c1; ++c) memcpy(stage + (size_t)(c - c0) * ldd_bytes, src + (size_t)c * row_bytes, row_bytes);
        }
        ok = hipMemcpyAsync(dst + (size_t)c0 * ldd_bytes, stage, (size_t)(c1 - c0) * ldd_bytes, hipMemcpyHostToDevice,
                            ctx->copy_stream[t]) == hipSuccess;
        if (ok) ok = hipEventRecord(done[(size_t)p], ctx->copy_stream[t]) == hipSuccess;
        if (ok) {
          if (freeb[b] == nullptr) ok = hipEventCreateWithFlags(&freeb[b], hipEventDisableTiming) == hipSuccess;
          if (ok) ok = hipEventRecord(freeb[b], ctx->copy_stream[t]) == hipSuccess;
        }
      }
      if (!ok) failed.store(1);
      ready[(size_t)p].store(1, std::memory_order_release);
    }
    for (int b = 0; b < 2; ++b)
      if (freeb[b] != nullptr) { hipEventSynchronize(freeb[b]); hipEventDestroy(freeb[b]); }
  };
  std::vector<std::thread> th;
  th.reserve((size_t)T);
  for (int t = 0; t < T && t < npan; ++t) {
    try {
      th.emplace_back(feeder, t);
    } catch (...) {   // (a thread the system refuses: its panels are staged by this thread, before the loop below waits)
      feeder(t);
    }
  }
  int rc = PLAIDHIP_OK;
  try {
    for (int64_t p = 0; p < npan; ++p) {
      while (ready[(size_t)p].load(std::memory_order_acquire) == 0) std::this_thread::yield();
      if (failed.load() != 0 || rc != PLAIDHIP_OK) continue;
      if (hipStreamWaitEvent(ctx->stream, done[(size_t)p], 0) != hipSuccess) { failed.store(1); continue; }
      if (on_panel) rc = on_panel(pb[(size_t)p], pb[(size_t)p + 1]);
    }
  } catch (...) {   // (the feeders hold references into this frame: they are joined before anything unwinds)
    failed.store(1);
    for (auto& t : th) t.join();
    throw;
  }
  for (auto& t : th) t.join();
  if (failed.load() != 0 && rc == PLAIDHIP_OK) {
    set_error("pipelined host-to-device copy failed (%s)", hipGetErrorString(hipGetLastError()));
    rc = PLAIDHIP_EHIP;
  }
  return rc;
}

// all threads of a sharded call meet here between phases; a thread that failed keeps arriving (doing nothing in
// between), so nobody waits forever
class Rendezvous {
 public:
  explicit Rendezvous(int n) : n_(n) {}
  void arrive_and_wait() {
    std::unique_lock<std::mutex> lk(mu_);
    const int gen = gen_;
    if (++count_ == n_) {
      count_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return gen_ != gen; });
    }
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, count_ = 0, gen_ = 0;
};

struct Shared {
  explicit Shared(int n) : rv(n) {}
  Rendezvous rv;
  std::mutex mu;
  double gmax = 0.0;               // max(rX) over all shards
  bool gmax_set = false;
  uint32_t flags[4] = {0, 0, 0, 0};
  std::vector<double> med_all;     // medx of every sample column, global column order (each shard writes its block)
  std::atomic<int> abort{0};
};

struct Call {
  int method;   // 0 plaid, 1 sing, 2 ssgsea
  const int32_t* Xp;
  const int32_t* Xi;
  const double* X;   // dense values or CSC @x
  int32_t g, n;
  const int32_t* Gp;
  const int32_t* Gi;
  int32_t m;
  int stat, normalize;
  double alpha;
  double* S_out;
};

inline int64_t even_ld(int32_t g) { return (int64_t)g + (g & 1); }

// device buffers that live in the context between calls (ctx_buffer); same interface as DevBuf
struct CtxBuf {
  plaidhip_ctx* ctx;
  int slot;
  void* p = nullptr;
  int alloc(size_t bytes) { return ctx_buffer(ctx, slot, bytes, &p); }
  template <typename T> T* as() { return static_cast<T*>(p); }
};

// one device's part of a sharded call.  Returns a status; `sh` carries the cross-shard scalars.
#ifdef PLAIDHIP_DIAG
#define PH_TRACE(tag)                                                                                              \
  do {                                                                                                             \
    if (getenv("PLAIDHIP_TRACE"))                                                                                  \
      fprintf(stderr, "[trace] %-22s %8.2f ms\n", tag,                                                             \
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_trace0).count());     \
  } while (0)
#else
#define PH_TRACE(tag) do { } while (0)
#endif

// {sum, count} of the non-NaN entries of v in exactly the order of sum_kernel (kernels_norm.hip): 1,024 strided partial
// sums, then a halving tree -- IEEE additions in the same order give the same bits on the host
double mean_like_device_sum(const double* v, int64_t count) {
  std::vector<double> s(1024, 0.0), cnt(1024, 0.0);
  for (int t = 0; t < 1024; ++t)
    for (int64_t i = t; i < count; i += 1024) {
      const double x = v[i];
      if (x == x) { s[(size_t)t] += x; cnt[(size_t)t] += 1.0; }
    }
  for (int h = 512; h >= 1; h >>= 1)
    for (int t = 0; t < h; ++t) { s[(size_t)t] += s[(size_t)(t + h)]; cnt[(size_t)t] += cnt[(size_t)(t + h)]; }
  return s[0] / cnt[0];
}

int shard_worker(plaidhip_ctx* ctx, const Call& c, int ndev, int k, Shared& sh) {
  int rc = PLAIDHIP_OK;
#ifdef PLAIDHIP_DIAG
  const auto t_trace0 = std::chrono::steady_clock::now();
#endif
  // every `step` is skipped once this shard or any other has failed; the rendezvous points are always reached
  auto live = [&] { return rc == PLAIDHIP_OK && sh.abort.load() == 0; };
  auto step = [&](const std::function<int()>& fn) {
    if (!live()) return;
    try {
      rc = fn();
    } catch (...) {   // (a worker thread has no function-try-block above it; the rendezvous points must still be reached)
      rc = on_exception();
    }
    if (rc != PLAIDHIP_OK) sh.abort.store(1);
  };
  int64_t lo64 = 0, hi64 = 0;
  plaidhip_shard_bounds(c.n, ndev, k, &lo64, &hi64);
  const int32_t lo = (int32_t)lo64, nloc = (int32_t)(hi64 - lo64);
  const int32_t g = c.g, m = c.m;
  const bool sparse = c.Xp != nullptr;
  const bool ranks = c.method != 0;
  plaidhip_geneset* gs = nullptr;
  CtxBuf dX{ctx, 0}, dXp{ctx, 1}, dXi{ctx, 2}, dR{ctx, 3}, dS{ctx, 4}, dsmall{ctx, 5};
  // the caller's S is usually fresh, untouched memory (R: allocMatrix): its pages are made while the upload and the
  // kernels run, the copy home follows chunk by chunk (HomeBuffer, common.h)
  HomeBuffer home;
  const int64_t ldg = even_ld(g);
  uint32_t* d_flags = nullptr;
  double *d_red = nullptr, *d_med = nullptr, *d_colmax = nullptr, *d_gmax = nullptr;
  int64_t zx = 0, z0 = 0;
  int32_t max_nnz = 0;
  std::vector<int32_t> ploc;

  step([&]() -> int {
    PH_HIP(hipSetDevice(ctx->device));
    PH_TRY(acquire_geneset(ctx, g, m, c.Gp, c.Gi, &gs));
    PH_TRACE("geneset acquired");
    PH_TRY(dsmall.alloc(64 + (size_t)std::max(nloc, 1) * 16));
    d_flags = dsmall.as<uint32_t>();
    d_red = reinterpret_cast<double*>(dsmall.as<char>() + 16);
    d_gmax = d_red + 2;
    d_med = reinterpret_cast<double*>(dsmall.as<char>() + 64);
    d_colmax = d_med + std::max(nloc, 1);
    PH_HIP(hipMemsetAsync(dsmall.p, 0, 64, ctx->stream));
    PH_TRY(dS.alloc((size_t)m * std::max(nloc, 1) * 8));
    if (nloc == 0) return PLAIDHIP_OK;
    if (!sparse) {
      PH_TRY(dX.alloc((size_t)ldg * nloc * 8));
      if (ranks) PH_TRY(dR.alloc((size_t)ldg * nloc * 8));
      const double* Xh = c.X + (int64_t)lo * g;
      // the kernels of a column panel follow its DMA: the crossprod itself for plaid(), the ranks for the others
      // (their crossprod needs max(rX) of ALL columns first, R/plaid.R:251)
      auto on_panel = [&](int64_t c0, int64_t c1) -> int {
        const int32_t nc = (int32_t)(c1 - c0);
        const double* xp = dX.as<double>() + c0 * ldg;
        if (c.method == 0)
          return launch_spmm_dense_f64(ctx, gs, xp, ldg, nc, c.stat, 1.0, nullptr, 0.0, dS.as<double>() + c0 * m, m, d_flags);
        return launch_colranks_dense_f64(ctx, xp, ldg, g, nc, c.method == 1 ? PLAIDHIP_TIES_MIN : PLAIDHIP_TIES_AVERAGE, 0,
                                         c.method == 2 ? 1.0 + c.alpha : 1.0, dR.as<double>() + c0 * ldg, ldg,
                                         c.method == 2 ? d_colmax + c0 : nullptr);
      };
      PH_TRACE("buffers ready");
      PH_TRY(upload_pipelined(ctx, dX.as<char>(), (size_t)ldg * 8, reinterpret_cast<const char*>(Xh), (size_t)g * 8, nloc,
                              on_panel));
      PH_TRACE("upload enqueued");
      // (the result's pages are made from here on, not earlier: eight threads faulting pages in next to the four feeder
      // threads cost the upload a quarter of its rate)
      home.prepare(c.S_out + (int64_t)lo * m, (size_t)m * nloc * 8);
    } else {
      z0 = c.Xp[lo];
      zx = (int64_t)c.Xp[lo + nloc] - z0;
      ploc.resize((size_t)nloc + 1);
      for (int32_t j = 0; j <= nloc; ++j) {
        ploc[(size_t)j] = (int32_t)(c.Xp[lo + j] - z0);
        if (j > 0) max_nnz = std::max(max_nnz, ploc[(size_t)j] - ploc[(size_t)j - 1]);
      }
      PH_TRY(dXp.alloc((size_t)(nloc + 1) * 4));
      PH_TRY(dXi.alloc((size_t)zx * 4));
      PH_TRY(dX.alloc((size_t)zx * 8));
      // replaid.sing ranks the zeros too (colranks' sparse branch without keep.zero, R/plaid.R:602-609: a dense rank
      // matrix): built panel by panel from the ranks of the stored values, each panel multiplied at once
      int64_t panel = ((int64_t)2 << 30) / (ldg * 8);
      panel = std::min<int64_t>(std::max<int64_t>(panel & ~(int64_t)1, 2), nloc);
      if (c.method == 1) PH_TRY(dR.alloc((size_t)(panel * ldg + zx) * 8));   // a panel of dense ranks | ranks of the stored values
      else if (ranks) PH_TRY(dR.alloc((size_t)zx * 8));
      PH_HIP(hipMemcpyAsync(dXp.p, ploc.data(), (size_t)(nloc + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
      PH_TRY(upload_pipelined(ctx, dXi.as<char>(), 1, reinterpret_cast<const char*>(c.Xi + z0), 1, zx * 4, nullptr));
      PH_TRY(upload_pipelined(ctx, dX.as<char>(), 1, reinterpret_cast<const char*>(c.X + z0), 1, zx * 8, nullptr));
      home.prepare(c.S_out + (int64_t)lo * m, (size_t)m * nloc * 8);
      if (c.method == 1) {
        double* dRd = dR.as<double>();
        double* dRx = dRd + panel * ldg;
        const bool by_stored = max_nnz <= max_sparse_rank_column();
        for (int64_t c0 = 0; c0 < nloc; c0 += panel) {
          const int32_t nc = (int32_t)std::min<int64_t>(panel, nloc - c0);
          const int32_t* xp = dXp.as<int32_t>() + c0;          // (absolute offsets into the shard's @i / @x)
          if (by_stored)
            PH_TRY(launch_colranks_csc_dense_nz_f64(ctx, xp, dXi.as<int32_t>(), dX.as<double>(), g, nc, max_nnz,
                                                    PLAIDHIP_TIES_MIN, 0, 1.0, dRx, dRd, ldg, nullptr));
          else   // a column with more stored values than one pass ranks: densify and rank
            PH_TRY(launch_colranks_csc_dense_f64(ctx, xp, dXi.as<int32_t>(), dX.as<double>(), g, nc, PLAIDHIP_TIES_MIN, 0, 1.0,
                                                 dRd, ldg, nullptr));
          PH_TRY(launch_spmm_dense_f64(ctx, gs, dRd, ldg, nc, PLAIDHIP_STAT_MEAN, 1.0 / (double)g, nullptr, -0.5,   // R/plaid.R:216
                                       dS.as<double>() + c0 * m, m, d_flags, PLAIDHIP_X_RANKS));
        }
      } else if (ranks)   // sparse_colranks: the stored values among themselves (R/plaid.R:600-601, 631-650)
        PH_TRY(launch_colranks_csc_f64(ctx, dXp.as<int32_t>(), dX.as<double>(), nloc, max_nnz,
                                       c.method == 1 ? PLAIDHIP_TIES_MIN : PLAIDHIP_TIES_AVERAGE, 0,
                                       c.method == 2 ? 1.0 + c.alpha : 1.0, dR.as<double>(), c.method == 2 ? d_colmax : nullptr));
    }
    return PLAIDHIP_OK;
  });

  // ---- max(rX) over every shard (replaid.ssgsea, R/plaid.R:251) ----------------------------------------------------
  double gmax = 0.0;
  if (c.method == 2) {
    double mine = sparse ? 0.0 : -INFINITY;      // a dgCMatrix has implicit zeros
    step([&]() -> int {
      if (nloc == 0) return PLAIDHIP_OK;
      PH_TRY(launch_max(ctx, d_colmax, nloc, d_gmax));
      PH_HIP(hipMemcpyAsync(&mine, d_gmax, 8, hipMemcpyDeviceToHost, ctx->stream));
      PH_HIP(hipStreamSynchronize(ctx->stream));
      return PLAIDHIP_OK;
    });
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      if (nloc > 0 && rc == PLAIDHIP_OK) {
        sh.gmax = sh.gmax_set ? std::max(sh.gmax, mine) : mine;
        sh.gmax_set = true;
      }
    }
    sh.rv.arrive_and_wait();
    gmax = sh.gmax;
  }

  // ---- crossprod of the rank-based callers (plaid() did it per panel) ------------------------------------------------
  step([&]() -> int {
    if (ctx->debug_fail_crossprod) { set_error("injected failure in the crossprod phase (test hook)"); return PLAIDHIP_EHIP; }
    if (nloc == 0) return PLAIDHIP_OK;
    if (c.method == 0 && !sparse) return PLAIDHIP_OK;
    if (c.method == 1 && sparse) return PLAIDHIP_OK;   // done panel by panel above
    double a = 1.0, b = 0.0;
    int stat = c.stat;
    if (c.method == 1) { a = 1.0 / (double)g; b = -0.5; stat = PLAIDHIP_STAT_MEAN; }       // R/plaid.R:216
    if (c.method == 2) { a = 1.0 / gmax; b = -0.5; stat = PLAIDHIP_STAT_MEAN; }            // R/plaid.R:251
    const double* vals = ranks ? dR.as<double>() : dX.as<double>();
    if (sparse) {
      // scatter or gather is chosen from the density of the WHOLE matrix, not of the shard: every sharding of the same
      // call takes the same kernel (the gather kernels are then bit-identical across shardings; the scatter kernel adds
      // in arrival order and agrees to the last bits only, as it does from run to run -- and its fixed-point grid, where
      // the column-sum bound picks it, follows the SHARD's largest column sum: 2^-40 relative across shardings)
      const int64_t nnz_choice = (int64_t)((double)c.Xp[c.n] / (double)c.n * (double)nloc);
      // replaid.ssgsea: the values are rank weights in [0, max(rX)] (the scatter kernel may sum them in fixed point)
      // (normalised results: the crossprod also classifies its scores for the medians below, launch_col_medians_resume)
      const bool will_norm = c.method == 2 || (c.method == 0 && c.normalize);
      if (will_norm)
        return launch_spmm_csc_fused_f64(ctx, gs, dXp.as<int32_t>(), dXi.as<int32_t>(), vals, nloc, zx, stat, a, nullptr, b,
                                         dS.as<double>(), m, d_flags, /*bounded=*/c.method == 2, nullptr, gmax, nnz_choice);
      return launch_spmm_csc_f64(ctx, gs, dXp.as<int32_t>(), dXi.as<int32_t>(), vals, nloc, nnz_choice, stat, a, nullptr, b,
                                 dS.as<double>(), m, d_flags, /*bounded=*/c.method == 2, nullptr, gmax);
    }
    const int x_kind = (c.method == 1 || (c.method == 2 && c.alpha == 0.0)) ? PLAIDHIP_X_RANKS : PLAIDHIP_X_ANY;
    // (normalised results on the fp64 pair kernel: the crossprod also classifies its scores for the medians below)
    if (c.method == 2 || (c.method == 0 && c.normalize))
      return launch_spmm_dense_fused_f64(ctx, gs, vals, ldg, nloc, stat, a, nullptr, b, dS.as<double>(), m, d_flags, x_kind);
    return launch_spmm_dense_f64(ctx, gs, vals, ldg, nloc, stat, a, nullptr, b, dS.as<double>(), m, d_flags, x_kind);
  });

  // ---- normalize_medians (R/plaid.R:554-575): two more scalars --------------------------------------------------------
  const bool norm = c.method == 2 || (c.method == 0 && c.normalize);
  if (norm) {
    uint32_t fl[4] = {0, 0, 0, 0};
    step([&]() -> int {
      if (nloc == 0) return PLAIDHIP_OK;
      PH_HIP(hipMemcpyAsync(fl, d_flags, 16, hipMemcpyDeviceToHost, ctx->stream));
      PH_HIP(hipStreamSynchronize(ctx->stream));
      PH_TRACE("crossprod done (flags)");
      return PLAIDHIP_OK;
    });
    {
      std::lock_guard<std::mutex> lk(sh.mu);
      for (int q = 0; q < 4; ++q) sh.flags[q] |= fl[q];
    }
    sh.rv.arrive_and_wait();
    const int ignore_zero = (sh.flags[1] != 0 && sh.flags[0] == 0) ? 1 : 0;   // min(x) == 0, R/plaid.R:556-557
    step([&]() -> int {
      if (nloc == 0) return PLAIDHIP_OK;
      PH_TRY(launch_col_medians_resume(ctx, dS.as<double>(), m, m, nloc, ignore_zero, nullptr, d_med));
      PH_HIP(hipMemcpyAsync(sh.med_all.data() + lo, d_med, (size_t)nloc * 8, hipMemcpyDeviceToHost, ctx->stream));
      PH_HIP(hipStreamSynchronize(ctx->stream));
      return PLAIDHIP_OK;
    });
    sh.rv.arrive_and_wait();
    // mean(medx, na.rm = TRUE), R/plaid.R:572, over ALL columns in the summation order of the device's sum kernel
    // (launch_sum): the value does not depend on how the columns were sharded, so every sharding -- one device
    // included -- normalises with the same bits
    const double mean_med = live() ? mean_like_device_sum(sh.med_all.data(), c.n) : 0.0;
    step([&]() -> int {
      if (nloc == 0) return PLAIDHIP_OK;
      return launch_shift_columns(ctx, dS.as<double>(), m, m, nloc, d_med, mean_med, nullptr);
    });
  }

  // ---- the score shard goes home (pageable destination: the runtime's own staging runs at ~53 GB/s) ---------------------
  PH_TRACE("normalise enqueued");
  step([&]() -> int {
    if (nloc > 0) PH_TRY(home.copy(ctx, dS.p));
    PH_HIP(hipStreamSynchronize(ctx->stream));
    PH_TRACE("scores home");
    return PLAIDHIP_OK;
  });
  if (rc == PLAIDHIP_OK && sh.abort.load() != 0) {
    hipStreamSynchronize(ctx->stream);
    return PLAIDHIP_EHIP;   // another shard failed; its error text is reported
  }
  if (rc != PLAIDHIP_OK) hipStreamSynchronize(ctx->stream);
  return rc;
}

}  // namespace

namespace plaidhip {

// pageable host memory -> device through the pinned staging ring (the other host entry points' uploads: a plain
// hipMemcpy from pageable memory runs at ~21 GB/s, the ring at the link rate)
int upload_host(plaidhip_ctx* ctx, void* dst, size_t ldd_bytes, const void* src, size_t row_bytes, int64_t cols) {
  return upload_pipelined(ctx, static_cast<char*>(dst), ldd_bytes, static_cast<const char*>(src), row_bytes, cols, nullptr);
}

int run_sharded(plaidhip_ctx* const* ctxs, int ndev, int method, const int32_t* Xp, const int32_t* Xi, const double* X_or_x,
                int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, int stat, int normalize, double alpha,
                double* S_out) {
  PH_REQUIRE(ndev >= 1 && ctxs != nullptr, "sharded call: no device");
  for (int k = 0; k < ndev; ++k) PH_REQUIRE(ctxs[k] != nullptr, "sharded call: null context %d", k);
  PH_TRY(check_host_common(Gp, g, n, m));
  if ((int64_t)m * n == 0) return PLAIDHIP_OK;
  PH_REQUIRE(X_or_x != nullptr || (Xp != nullptr && Xp[n] == 0), "null X");
  PH_REQUIRE(S_out != nullptr, "null S_out");
  if (Xp != nullptr) PH_TRY(check_host_csc(Xp, Xi, g, n));
  Call c{method, Xp, Xi, X_or_x, g, n, Gp, Gi, m, stat, normalize, alpha, S_out};
  Shared sh(ndev);
  sh.med_all.assign((size_t)n, 0.0);
  if (ndev == 1) return shard_worker(ctxs[0], c, 1, 0, sh);
  std::vector<int> rcs((size_t)ndev, PLAIDHIP_OK);
  std::vector<std::string> errs((size_t)ndev);
  std::vector<std::thread> th;
  for (int k = 0; k < ndev; ++k)
    th.emplace_back([&, k] {
      rcs[(size_t)k] = shard_worker(ctxs[k], c, ndev, k, sh);
      if (rcs[(size_t)k] != PLAIDHIP_OK) errs[(size_t)k] = last_error_cstr();   // the worker's thread-local text
    });
  for (auto& t : th) t.join();
  // report the failure that started it (the others only say "another shard failed")
  int rc = PLAIDHIP_OK;
  for (int k = 0; k < ndev; ++k)
    if (rcs[(size_t)k] != PLAIDHIP_OK && !errs[(size_t)k].empty()) {
      rc = rcs[(size_t)k];
      set_error("device %d: %s", ctxs[k]->device, errs[(size_t)k].c_str());
      break;
    }
  if (rc == PLAIDHIP_OK)
    for (int k = 0; k < ndev; ++k)
      if (rcs[(size_t)k] != PLAIDHIP_OK) { rc = rcs[(size_t)k]; set_error("a device shard failed"); break; }
  return rc;
}

}  // namespace plaidhip

// ---- multi-device entry points (include/plaidhip.h) ---------------------------------------------------------------------
namespace {

std::mutex g_multi_mu;
std::vector<plaidhip_ctx*> g_multi_ctx;   // one lazily created context per device, owned by the library
int g_multi_precision = PLAIDHIP_PRECISION_F64;   // plaidhip_multi_set_precision: applies to these contexts

int multi_contexts(const int* devices, int ndev, std::vector<plaidhip_ctx*>& out) {
  PH_REQUIRE(ndev >= 1 && ndev <= 64, "multi: ndev = %d", ndev);
  int count = 0;
  PH_TRY(plaidhip_device_count(&count));
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if ((int)g_multi_ctx.size() < count) g_multi_ctx.resize((size_t)count, nullptr);
  out.clear();
  for (int k = 0; k < ndev; ++k) {
    const int d = devices ? devices[k] : k;
    PH_REQUIRE(d >= 0 && d < count, "multi: device %d out of range [0, %d)", d, count);
    for (int q = 0; q < k; ++q) PH_REQUIRE((devices ? devices[q] : q) != d, "multi: device %d listed twice", d);
    if (g_multi_ctx[(size_t)d] == nullptr) PH_TRY(plaidhip_init(d, nullptr, &g_multi_ctx[(size_t)d]));
    g_multi_ctx[(size_t)d]->precision = g_multi_precision;
    out.push_back(g_multi_ctx[(size_t)d]);
  }
  return PLAIDHIP_OK;
}

}  // namespace

extern "C" {

// Test hook (not part of include/plaidhip.h): the multi-device engine with `nshards` contexts on ONE device -- worker
// threads, rendezvous, cross-shard scalars and the failure path are what a 1-GPU box can exercise of plaidhip_*_multi.
// method 0 plaid, 1 sing, 2 ssgsea; fail_shard >= 0: that shard fails in its crossprod phase (the call must return an
// error, not hang).
int plaidhip_debug_sharded_on_one_device(int device, int nshards, int fail_shard, int method, const int32_t* Xp,
                                         const int32_t* Xi, const double* X_or_x, int32_t g, int32_t n, const int32_t* Gp,
                                         const int32_t* Gi, int32_t m, int stat, int normalize, double alpha, double* S_out) try {
  PH_REQUIRE(nshards >= 1 && nshards <= 64, "debug_sharded: nshards = %d", nshards);
  std::vector<plaidhip_ctx*> ctxs((size_t)nshards, nullptr);
  int rc = PLAIDHIP_OK;
  for (int k = 0; k < nshards && rc == PLAIDHIP_OK; ++k) {
    rc = plaidhip_init(device, nullptr, &ctxs[(size_t)k]);
    if (rc == PLAIDHIP_OK && k == fail_shard) ctxs[(size_t)k]->debug_fail_crossprod = 1;
  }
  if (rc == PLAIDHIP_OK)
    rc = run_sharded(ctxs.data(), nshards, method, Xp, Xi, X_or_x, g, n, Gp, Gi, m, stat, normalize, alpha, S_out);
  const std::string err = rc != PLAIDHIP_OK ? std::string(last_error_cstr()) : std::string();
  for (plaidhip_ctx* c : ctxs)
    if (c) plaidhip_finalize(c);
  if (rc != PLAIDHIP_OK) set_error("%s", err.c_str());
  return rc;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_multi_set_precision(int mode) try {
  PH_REQUIRE(mode == PLAIDHIP_PRECISION_F64 || mode == PLAIDHIP_PRECISION_MIXED, "multi_set_precision: bad mode %d", mode);
  std::lock_guard<std::mutex> lk(g_multi_mu);
  g_multi_precision = mode;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_multi_finalize(void) try {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  for (plaidhip_ctx*& c : g_multi_ctx)
    if (c) { plaidhip_finalize(c); c = nullptr; }
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_plaid_multi(const int* devices, int ndev, const int32_t* Xp, const int32_t* Xi, const double* X_or_x, int32_t g,
                         int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, int stat, int normalize, double* S_out) try {
  std::vector<plaidhip_ctx*> ctxs;
  PH_TRY(multi_contexts(devices, ndev, ctxs));
  PH_REQUIRE(stat == PLAIDHIP_STAT_MEAN || stat == PLAIDHIP_STAT_SUM, "plaid_multi: bad stat %d", stat);
  return run_sharded(ctxs.data(), ndev, 0, Xp, Xi, X_or_x, g, n, Gp, Gi, m, stat, normalize, 0.0, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_sing_multi(const int* devices, int ndev, const double* X, int32_t g, int32_t n, const int32_t* Gp,
                        const int32_t* Gi, int32_t m, double* S_out) try {
  std::vector<plaidhip_ctx*> ctxs;
  PH_TRY(multi_contexts(devices, ndev, ctxs));
  return run_sharded(ctxs.data(), ndev, 1, nullptr, nullptr, X, g, n, Gp, Gi, m, PLAIDHIP_STAT_MEAN, 0, 0.0, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_sing_csc_multi(const int* devices, int ndev, const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t g,
                            int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, double* S_out) try {
  PH_REQUIRE(Xp != nullptr, "sing_csc_multi: null Xp");
  std::vector<plaidhip_ctx*> ctxs;
  PH_TRY(multi_contexts(devices, ndev, ctxs));
  return run_sharded(ctxs.data(), ndev, 1, Xp, Xi, Xx, g, n, Gp, Gi, m, PLAIDHIP_STAT_MEAN, 0, 0.0, S_out);
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_ssgsea_multi(const int* devices, int ndev, const int32_t* Xp, const int32_t* Xi, const double* X_or_x, int32_t g,
                          int32_t n, const int32_t* Gp, const int32_t* Gi, int32_t m, double alpha, double* S_out) try {
  std::vector<plaidhip_ctx*> ctxs;
  PH_TRY(multi_contexts(devices, ndev, ctxs));
  return run_sharded(ctxs.data(), ndev, 2, Xp, Xi, X_or_x, g, n, Gp, Gi, m, PLAIDHIP_STAT_MEAN, 1, alpha, S_out);
} catch (...) { return plaidhip::on_exception(); }

}  // extern "C"
