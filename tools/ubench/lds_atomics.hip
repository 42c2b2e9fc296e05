// LDS atomic-add microbenchmark (gfx950): LDS-array time per wave-instruction for the accumulator types the
// scatter SpMM could use (f64, u64, f32, u32), with addresses as the scatter kernel sees them: 64 distinct random
// accumulators per wave-instruction (a gene's sets), 16 waves per CU issuing back to back.
//   hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename T> __device__ __forceinline__ void lds_add(T* p, T v);
template <> __device__ __forceinline__ void lds_add<double>(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<float>(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<unsigned long long>(unsigned long long* p, unsigned long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<unsigned>(unsigned* p, unsigned v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// MODE 0: random distinct slots per instruction (host-permuted ids); 1: conflict-free (slot = lane + 64 k); 2: plain
// (non-atomic) read-modify-write for reference
// stride probe: lane l adds into slot (l * stride + rot) mod 16384 -- which strides are conflict-free tells the lane
// groups and the bank width of the LDS atomic unit
template <typename T>
__global__ void __launch_bounds__(1024) kstride(int stride, int group, int iters, unsigned long long* cyc) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* acc = reinterpret_cast<T*>(smem);
  for (int i = threadIdx.x; i < 20480; i += 1024) acc[i] = (T)0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lanes inside a group of `group` lanes are `stride` slots apart; groups sit in different rows of 32 slots
  const unsigned base = (unsigned)((lane % group) * stride + (lane / group) * 32 * 17);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  T v = (T)1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 8
    for (int q = 0; q < 256; ++q) lds_add<T>(&acc[(base + (unsigned)(q + wave) * 32u) % 16384u], v);
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename T, int MODE>
__global__ void __launch_bounds__(1024) k(const unsigned short* ids, int nid, int iters, unsigned long long* cyc, T* out) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* acc = reinterpret_cast<T*>(smem);
  const int nslots = 20480 * 8 / sizeof(T) > 20480 ? 20480 : 20480;
  for (int i = threadIdx.x; i < nslots; i += 1024) acc[i] = (T)0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* my = ids + (size_t)wave * nid * 64 + lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  T v = (T)1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 8
    for (int q = 0; q < nid; ++q) {
      const unsigned slot = MODE == 1 ? (unsigned)(lane + 64 * ((q + wave) & 255)) : my[(size_t)q * 64];
      if (MODE == 2) acc[slot] += v; else lds_add<T>(&acc[slot], v);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (out) out[blockIdx.x * 1024 + threadIdx.x] = acc[threadIdx.x];
}

// ids held in registers (32 per lane, applied round and round): no id load between the atomics, so the LDS unit is the
// only bound (the kernel above loads a u16 per lane and instruction and saturates the vector-memory issue at ~14 cycles)
template <typename T>
__global__ void __launch_bounds__(1024) kreg(const unsigned short* ids, int nid, int iters, unsigned long long* cyc) {
  extern __shared__ __align__(16) unsigned char smem[];
  T* acc = reinterpret_cast<T*>(smem);
  for (int i = threadIdx.x; i < 20480; i += 1024) acc[i] = (T)0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* my = ids + (size_t)wave * nid * 64 + lane;
  unsigned id[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) id[q] = my[(size_t)q * 64];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  T v = (T)1;
  for (int it = 0; it < iters * (nid / 32); ++it) {
#pragma unroll
    for (int q = 0; q < 32; ++q) lds_add<T>(&acc[id[q]], v);
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename T>
static void run_reg(const char* name, const unsigned short* d_ids, int nid, int iters) {
  unsigned long long* d_cyc;
  hipMalloc(&d_cyc, 256 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kreg<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  kreg<T><<<256, 1024, 160 * 1024>>>(d_ids, nid, 1, d_cyc);
  kreg<T><<<256, 1024, 160 * 1024>>>(d_ids, nid, iters, d_cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(256);
  hipMemcpy(c.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto x : c) cyc += (double)x; cyc /= 256;
  printf("%-56s %7.2f shader cycles per wave-instruction per CU (ids in registers)\n", name, cyc / (16.0 * nid * iters));
  hipFree(d_cyc);
}

template <typename T, int MODE>
static double run(const char* name, const unsigned short* d_ids, int nid, int iters) {
  unsigned long long* d_cyc;
  hipMalloc(&d_cyc, 256 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<T, MODE><<<256, 1024, 160 * 1024>>>(d_ids, nid, 1, d_cyc, nullptr);
  hipEventRecord(e0);
  k<T, MODE><<<256, 1024, 160 * 1024>>>(d_ids, nid, iters, d_cyc, nullptr);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(256);
  hipMemcpy(c.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto x : c) cyc += (double)x; cyc /= 256;
  const double winstr = 16.0 * nid * iters;                        // wave-instructions per CU
  printf("%-28s %8.3f ms  %7.2f shader cycles per wave-instruction per CU  (%.2f G lane-adds/s chip)\n", name, ms,
         cyc / winstr, 256.0 * winstr * 64 / (ms * 1e-3) / 1e9);
  hipFree(d_cyc);
  return ms;
}

int main() {
  const int nid = 256, iters = 50;
  std::vector<unsigned short> ids((size_t)16 * nid * 64);
  srand(1);
  for (size_t i = 0; i < ids.size(); i += 64) {   // 64 distinct random slots per instruction
    std::vector<int> seen;
    for (int l = 0; l < 64; ++l) {
      int s;
      bool dup;
      do { s = rand() % 20480; dup = false; for (int x : seen) dup |= (x == s); } while (dup);
      seen.push_back(s);
      ids[i + l] = (unsigned short)s;
    }
  }
  unsigned short* d_ids;
  hipMalloc(&d_ids, ids.size() * 2);
  hipMemcpy(d_ids, ids.data(), ids.size() * 2, hipMemcpyHostToDevice);
  {
    // banks distinct inside every 16-lane group (what the scatter plan achieves where the counts allow), rows random
    std::vector<unsigned short> ids2(ids.size());
    for (size_t i = 0; i < ids2.size(); i += 16) {
      int perm[16];
      for (int r = 0; r < 16; ++r) perm[r] = r;
      for (int r = 15; r > 0; --r) { const int j = rand() % (r + 1); const int t = perm[r]; perm[r] = perm[j]; perm[j] = t; }
      for (int l = 0; l < 16; ++l) ids2[i + l] = (unsigned short)(16 * (rand() % 1280) + perm[l]);
    }
    unsigned short* d_ids2;
    hipMalloc(&d_ids2, ids2.size() * 2);
    hipMemcpy(d_ids2, ids2.data(), ids2.size() * 2, hipMemcpyHostToDevice);
    run_reg<double>("ds_add_f64 banks distinct per 16 lanes, random rows", d_ids2, nid, iters);
    run_reg<unsigned long long>("ds_add_u64 banks distinct per 16 lanes, random rows", d_ids2, nid, iters);
    run_reg<double>("ds_add_f64 64 distinct random slots", d_ids, nid, iters);
    run_reg<unsigned long long>("ds_add_u64 64 distinct random slots", d_ids, nid, iters);
    // the same with fewer distinct 128-byte rows per 16-lane group: 1 (16 consecutive slots at a random base), 2, 4, 8
    for (int nrows : {1, 2, 4, 8}) {
      for (size_t i = 0; i < ids2.size(); i += 16) {
        int perm[16], rows[8];
        for (int r = 0; r < 16; ++r) perm[r] = r;
        for (int r = 15; r > 0; --r) { const int j = rand() % (r + 1); const int t = perm[r]; perm[r] = perm[j]; perm[j] = t; }
        for (int r = 0; r < nrows; ++r) rows[r] = rand() % 1280;
        for (int l = 0; l < 16; ++l) ids2[i + l] = (unsigned short)(16 * rows[l % nrows] + perm[l]);
      }
      hipMemcpy(d_ids2, ids2.data(), ids2.size() * 2, hipMemcpyHostToDevice);
      char nm[96];
      snprintf(nm, sizeof nm, "ds_add_f64 banks distinct, %d rows per 16 lanes", nrows);
      run_reg<double>(nm, d_ids2, nid, iters);
    }
    hipFree(d_ids2);
  }
  run<double, 0>("ds_add_f64 random", d_ids, nid, iters);
  run<double, 1>("ds_add_f64 conflict-free", d_ids, nid, iters);
  run<unsigned long long, 0>("ds_add_u64 random", d_ids, nid, iters);
  run<unsigned long long, 1>("ds_add_u64 conflict-free", d_ids, nid, iters);
  run<float, 0>("ds_add_f32 random", d_ids, nid, iters);
  run<float, 1>("ds_add_f32 conflict-free", d_ids, nid, iters);
  run<unsigned, 0>("ds_add_u32 random", d_ids, nid, iters);
  run<unsigned, 1>("ds_add_u32 conflict-free", d_ids, nid, iters);
  run<double, 2>("plain f64 rmw random (racy)", d_ids, nid, iters);
  {
    unsigned long long* d_cyc;
    hipMalloc(&d_cyc, 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&kstride<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int groups[] = {64, 32, 16, 8};
    const int strides[] = {1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33};
    for (int g : groups)
      for (int st : strides) {
        kstride<double><<<256, 1024, 160 * 1024>>>(st, g, 20, d_cyc);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(256);
        hipMemcpy(c.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
        double cyc = 0; for (auto x : c) cyc += (double)x; cyc /= 256;
        printf("ds_add_f64 lanes-per-group %2d stride %2d slots: %6.2f cycles per wave-instruction per CU\n", g, st, cyc / (16.0 * 256 * 20));
      }
  }
  return 0;
}
