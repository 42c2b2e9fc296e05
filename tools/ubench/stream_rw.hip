// Ceiling of an in-place read-modify-write sweep (what shift_columns does: x - a + b over the whole score matrix):
// flat grid-stride kernel, 16-byte accesses, UN of them in flight per thread, plain or non-temporal, several block counts.
//   hipcc --offload-arch=gfx950 -O3 stream_rw.hip -o stream_rw ; ./stream_rw [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double f64x2_t __attribute__((ext_vector_type(2)));

template <int UN, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) rw_kernel(f64x2_t* __restrict__ p, long npairs, double a, double b) {
  const long stride = (long)gridDim.x * 256 * UN;
  for (long base = (long)blockIdx.x * 256 * UN; base < npairs; base += stride) {
    f64x2_t v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long i = base + u * 256 + threadIdx.x;
      const f64x2_t* q = p + (i < npairs ? i : npairs - 1);
      v[u] = NTL ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long i = base + u * 256 + threadIdx.x;
      if (i < npairs) {
        f64x2_t r;
        r.x = (v[u].x - a) + b;
        r.y = (v[u].y - a) + b;
        if (NTS) __builtin_nontemporal_store(r, p + i); else p[i] = r;
      }
    }
  }
}

template <int UN, bool NTL, bool NTS>
static void run(f64x2_t* p, long npairs, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 6; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rw_kernel<UN, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, p, npairs, 0.25, 0.5);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  printf("  UN=%d ntload=%d ntstore=%d blocks=%6d: %.3f ms  %.2f TB/s (read + write)\n", UN, (int)NTL, (int)NTS, blocks, best,
         (double)npairs * 32 / best / 1e9);
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 3.2;
  const long npairs = (long)(gb * 1e9 / 16);
  f64x2_t* p;
  (void)hipMalloc(&p, npairs * 16);
  (void)hipMemset(p, 0, npairs * 16);
  printf("in-place x - a + b over %.2f GB\n", gb);
  for (int blocks : {2048, 8192, 32768, 131072}) {
    run<4, true, true>(p, npairs, blocks);
    run<4, false, false>(p, npairs, blocks);
    run<4, true, false>(p, npairs, blocks);
    run<4, false, true>(p, npairs, blocks);
    run<8, true, true>(p, npairs, blocks);
    run<2, true, true>(p, npairs, blocks);
  }
  return 0;
}
