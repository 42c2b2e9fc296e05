// Host <-> device transfer rates behind the pipelined host entry points (tools only):
// pageable vs pinned vs registered memory, and the multi-threaded memcpy that feeds the pinned staging buffers.
//   hipcc -O2 pcie.cpp -o pcie -lpthread && ./pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_memcpy(char* d, const char* s, size_t n, int nt) {
  std::vector<std::thread> th;
  const size_t per = (n / nt + 4095) & ~(size_t)4095;
  for (int t = 0; t < nt; ++t) {
    const size_t o = (size_t)t * per;
    if (o >= n) break;
    const size_t len = std::min(per, n - o);
    th.emplace_back([=] { memcpy(d + o, s + o, len); });
  }
  for (auto& t : th) t.join();
}

int main() {
  const size_t N = (size_t)1600 << 20;   // 1.6 GB like C2's X
  char* pageable = (char*)malloc(N);
  memset(pageable, 1, N);
  char* pinned;
  hipHostMalloc((void**)&pinned, N, hipHostMallocDefault);
  memset(pinned, 2, N);
  char* dev;
  hipMalloc((void**)&dev, N);
  hipStream_t st;
  hipStreamCreate(&st);
  auto time_copy = [&](const char* name, void* dst, const void* src, hipMemcpyKind k) {
    hipMemcpyAsync(dst, src, 64 << 20, k, st);
    hipStreamSynchronize(st);
    const double t0 = now();
    hipMemcpyAsync(dst, src, N, k, st);
    hipStreamSynchronize(st);
    const double t = now() - t0;
    printf("%-38s %7.1f ms  %6.1f GB/s\n", name, t * 1e3, N / t / 1e9);
  };
  time_copy("H2D pageable", dev, pageable, hipMemcpyHostToDevice);
  time_copy("H2D pinned", dev, pinned, hipMemcpyHostToDevice);
  time_copy("D2H pageable", pageable, dev, hipMemcpyDeviceToHost);
  time_copy("D2H pinned", pinned, dev, hipMemcpyDeviceToHost);
  {
    const double t0 = now();
    hipError_t e = hipHostRegister(pageable, N, hipHostRegisterDefault);
    const double t1 = now();
    printf("hipHostRegister(1.6 GB): %s %.1f ms\n", hipGetErrorString(e), (t1 - t0) * 1e3);
    if (e == hipSuccess) {
      time_copy("H2D registered", dev, pageable, hipMemcpyHostToDevice);
      time_copy("D2H registered", pageable, dev, hipMemcpyDeviceToHost);
      const double t2 = now();
      hipHostUnregister(pageable);
      printf("hipHostUnregister: %.1f ms\n", (now() - t2) * 1e3);
    }
  }
  for (int nt : {1, 2, 4, 8, 16, 32}) {
    par_memcpy(pinned, pageable, N, nt);
    const double t0 = now();
    par_memcpy(pinned, pageable, N, nt);
    const double t = now() - t0;
    printf("memcpy pageable -> pinned, %2d threads: %7.1f ms  %6.1f GB/s\n", nt, t * 1e3, N / t / 1e9);
  }
  return 0;
}
