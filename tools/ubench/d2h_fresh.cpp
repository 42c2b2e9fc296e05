// What the score matrix's way home costs when the caller's buffer is FRESH memory (R: allocMatrix -> malloc -> mmap: every
// page faults on its first write), tools only.  4.9 GB like the reference-shaped result (61,459 sets x 10,000 columns).
//   1. page-touch rate of T threads over a fresh mapping, plain and after madvise(MADV_HUGEPAGE)
//   2. hipMemcpy device -> fresh pageable / touched pageable
//   3. chunked: touch threads run ahead, the copy of chunk k is issued when its pages exist
//   4. pinned ring: DMA into pinned staging buffers, T threads memcpy them into the (fresh) destination
//   hipcc -O2 d2h_fresh.cpp -o d2h_fresh -lpthread && ./d2h_fresh
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static char* fresh(size_t n, bool huge) {
  char* p = (char*)mmap(nullptr, n + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { perror("mmap"); exit(1); }
  if (huge) madvise(p, n + (2u << 20), MADV_HUGEPAGE);
  return p;
}
static void drop(char* p, size_t n) { munmap(p, n + (2u << 20)); }

static void touch_chunks(char* p, size_t n, size_t chunk, int nt, std::atomic<int>* done) {
  std::atomic<size_t> next{0};
  const size_t nchunk = (n + chunk - 1) / chunk;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&] {
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= nchunk) return;
        const size_t b = k * chunk, e = std::min(n, b + chunk);
        for (size_t o = b; o < e; o += 4096) *reinterpret_cast<volatile char*>(p + o) = 0;
        if (done != nullptr) done[k].store(1, std::memory_order_release);
      }
    });
  for (auto& t : th) t.join();
}

int main() {
  const size_t N = (size_t)61459 * 10000 * 8;
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char line[128] = "?";
  if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
  printf("transparent_hugepage/enabled: %s", line);
  printf("hardware threads: %u\n", std::thread::hardware_concurrency());
  char* dev;
  if (hipMalloc((void**)&dev, N) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(dev, 1, N);
  hipStream_t st;
  hipStreamCreate(&st);
  hipDeviceSynchronize();

  for (int huge = 0; huge < 2; ++huge)
    for (int nt : {1, 2, 4, 8, 16, 32}) {
      char* p = fresh(N, huge);
      const double t0 = now();
      touch_chunks(p, N, (size_t)32 << 20, nt, nullptr);
      const double dt = now() - t0;
      printf("touch fresh %s  %2d threads: %7.1f ms  %6.1f GB/s\n", huge ? "MADV_HUGEPAGE" : "4K pages     ", nt, dt * 1e3, N / dt * 1e-9);
      drop(p, N);
    }
  for (int huge = 0; huge < 2; ++huge) {
    char* p = fresh(N, huge);
    double t0 = now();
    hipMemcpyAsync(p, dev, N, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    double dt = now() - t0;
    printf("hipMemcpy D2H into fresh %s: %7.1f ms  %6.1f GB/s\n", huge ? "MADV_HUGEPAGE" : "4K pages     ", dt * 1e3, N / dt * 1e-9);
    t0 = now();
    hipMemcpyAsync(p, dev, N, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    dt = now() - t0;
    printf("hipMemcpy D2H into the same, touched      : %7.1f ms  %6.1f GB/s\n", dt * 1e3, N / dt * 1e-9);
    drop(p, N);
  }
  // touch threads ahead, chunk copies behind
  for (int huge = 0; huge < 2; ++huge)
    for (int nt : {2, 4, 8, 16})
      for (size_t chunk_mb : {32, 128}) {
        const size_t chunk = chunk_mb << 20;
        const size_t nchunk = (N + chunk - 1) / chunk;
        char* p = fresh(N, huge);
        std::vector<std::atomic<int>> done(nchunk);
        for (auto& d : done) d.store(0);
        const double t0 = now();
        std::thread toucher([&] { touch_chunks(p, N, chunk, nt, done.data()); });
        for (size_t k = 0; k < nchunk; ++k) {
          while (done[k].load(std::memory_order_acquire) == 0) std::this_thread::yield();
          const size_t b = k * chunk, len = std::min(chunk, N - b);
          hipMemcpyAsync(p + b, dev + b, len, hipMemcpyDeviceToHost, st);
        }
        hipStreamSynchronize(st);
        const double dt = now() - t0;
        toucher.join();
        printf("touch ahead (%2d threads, %3zu MB chunks, %s) + chunk copies: %7.1f ms  %6.1f GB/s\n", nt, chunk_mb,
               huge ? "MADV_HUGEPAGE" : "4K pages     ", dt * 1e3, N / dt * 1e-9);
        drop(p, N);
      }
  // pinned ring: DMA into pinned buffers, worker threads copy them out into the fresh destination
  for (int huge = 0; huge < 2; ++huge)
    for (int nt : {4, 8, 16}) {
      const size_t chunk = (size_t)32 << 20;
      const int ring = 2 * nt;
      std::vector<char*> pin(ring);
      std::vector<hipEvent_t> ev(ring);
      for (int r = 0; r < ring; ++r) { hipHostMalloc((void**)&pin[r], chunk, hipHostMallocDefault); hipEventCreateWithFlags(&ev[r], hipEventDisableTiming); memset(pin[r], 0, chunk); }
      const size_t nchunk = (N + chunk - 1) / chunk;
      char* p = fresh(N, huge);
      // thread t owns chunks t, t + nt, ... and ring slots {t, t + nt}
      const double t0 = now();
      std::vector<std::thread> th;
      std::atomic<int> err{0};
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
          hipStream_t s;
          hipStreamCreate(&s);
          size_t k = t;
          int slot = 0;
          size_t pend_k[2] = {0, 0};
          bool pend[2] = {false, false};
          for (;; k += nt) {
            const int cur = slot & 1;
            if (pend[cur]) {
              hipEventSynchronize(ev[t + cur * nt]);
              const size_t b = pend_k[cur] * chunk, len = std::min(chunk, N - b);
              memcpy(p + b, pin[t + cur * nt], len);
              pend[cur] = false;
            }
            if (k < nchunk) {
              const size_t b = k * chunk, len = std::min(chunk, N - b);
              if (hipMemcpyAsync(pin[t + cur * nt], dev + b, len, hipMemcpyDeviceToHost, s) != hipSuccess) err.store(1);
              hipEventRecord(ev[t + cur * nt], s);
              pend[cur] = true;
              pend_k[cur] = k;
            } else if (!pend[0] && !pend[1]) {
              break;
            }
            ++slot;
          }
          hipStreamDestroy(s);
        });
      for (auto& t : th) t.join();
      const double dt = now() - t0;
      printf("pinned ring, %2d threads x 2 x 32 MB, into fresh %s: %7.1f ms  %6.1f GB/s%s\n", nt,
             huge ? "MADV_HUGEPAGE" : "4K pages     ", dt * 1e3, N / dt * 1e-9, err.load() ? "  (errors)" : "");
      // correctness of the last byte
      if (p[N - 1] != 1 || p[0] != 1 || p[N / 2] != 1) printf("  WRONG DATA\n");
      drop(p, N);
      for (int r = 0; r < ring; ++r) { hipHostFree(pin[r]); hipEventDestroy(ev[r]); }
    }
  return 0;
}
