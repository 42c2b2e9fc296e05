// Calibration of rocprofv3's FETCH_SIZE for the access widths this library uses (MI355X_MICROARCH.md, HBM section: the
// counter reads exactly half the bytes of 16-byte-per-lane streaming loads on gfx950 and is UNCALIBRATED for other widths:
// "calibrate on a known byte count in your own access pattern").  Each kernel below reads a buffer far larger than L2 +
// Infinity Cache exactly once, so the bytes that must cross the fabric are known:
//   seg256_dword   the scatter kernel's id loads: random 256-byte segments, one dword per lane (buffer/global_load_dword)
//   stream_dword   the same width, consecutive segments
//   stream_x4      16 bytes per lane, consecutive (the reference case of the guide: FETCH_SIZE = bytes / 2)
//   stream_x2      8 bytes per lane, consecutive (the median / rank kernels' loads)
// Run under rocprofv3 --pmc FETCH_SIZE (and TCC_MISS_sum TCC_HIT_sum in a second pass); the program prints the byte
// count of every kernel, tools/profile_round.sh puts counter and byte count side by side.
//     hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib && ./fetch_calib [GiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) seg256_dword(const uint32_t* __restrict__ buf, uint64_t nseg, uint64_t mult, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
  uint32_t acc = 0;
  for (uint64_t s = wave; s < nseg; s += nwaves) {
    const uint64_t seg = (s * mult) % nseg;            // a permutation of the segments (mult odd and coprime to nseg)
    acc += buf[seg * 64 + lane];
  }
  if (acc == 0x12345678u) out[0] = acc;                // (keeps the loads alive)
}
__global__ void __launch_bounds__(256) stream_dword(const uint32_t* __restrict__ buf, uint64_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc += buf[i];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) stream_x2(const uint2* __restrict__ buf, uint64_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { const uint2 v = buf[i]; acc += v.x + v.y; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) stream_x4(const uint4* __restrict__ buf, uint64_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { const uint4 v = buf[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  const uint64_t bytes = ((uint64_t)(gib * (1ull << 30)) / 4096) * 4096;
  void* buf; uint32_t* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int grid = pr.multiProcessorCount * 8;
  const uint64_t nseg = bytes / 256;
  uint64_t mult = 2654435761ull | 1ull;
  auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
  while (gcd(mult, nseg) != 1) mult += 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timed = [&](const char* name, auto launch) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s reads %llu bytes once (%.3f GB): %.3f ms, %.2f TB/s\n", name, (unsigned long long)bytes, bytes / 1e9, ms, bytes / ms / 1e9);
  };
  timed("seg256_dword", [&] { hipLaunchKernelGGL(seg256_dword, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, nseg, mult, out); });
  timed("stream_dword", [&] { hipLaunchKernelGGL(stream_dword, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out); });
  timed("stream_x2", [&] { hipLaunchKernelGGL(stream_x2, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out); });
  timed("stream_x4", [&] { hipLaunchKernelGGL(stream_x4, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out); });
  return 0;
}
