// What does the access pattern of the wave-per-column kernels allow?  One wavefront streams one column (16-byte loads,
// UN KiB per batch, optionally with the next batch requested before the current one is used) and only adds the values up:
// no selection work at all.  Sweeps waves per CU and batch size.   hipcc --offload-arch=gfx950 -O3 column_stream.hip -o column_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double f64x2_t __attribute__((ext_vector_type(2)));

template <int UN, bool PIPE, bool NT>
__global__ void __launch_bounds__(256) stream_kernel(const double* __restrict__ S, long lds, int m, int n, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  for (int c = blockIdx.x * 4 + wave; c < n; c += nwaves) {
    const f64x2_t* __restrict__ p = reinterpret_cast<const f64x2_t*>(S + (long)c * lds);
    const int npairs = m >> 1;
    double acc = 0.0;
    f64x2_t va[UN], vb[UN];
#define LD(buf, b0) _Pragma("unroll") for (int u = 0; u < UN; ++u) { const int i = (b0) + u * 64 + lane; const f64x2_t* q = p + (i < npairs ? i : npairs - 1); buf[u] = NT ? __builtin_nontemporal_load(q) : *q; }
#define USE(buf, b0) _Pragma("unroll") for (int u = 0; u < UN; ++u) { const bool ok = (b0) + u * 64 + lane < npairs; acc += ok ? buf[u].x + buf[u].y : 0.0; }
    if (PIPE) {
      LD(va, 0)
      for (int base = 0; base < npairs; base += 2 * 64 * UN) {
        LD(vb, base + 64 * UN)
        USE(va, base)
        LD(va, base + 2 * 64 * UN)
        USE(vb, base + 64 * UN)
      }
    } else {
      for (int base = 0; base < npairs; base += 64 * UN) {
        LD(va, base)
        USE(va, base)
      }
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[c] = acc;
  }
}

template <int UN, bool PIPE, bool NT>
static void run(const double* S, long lds, int m, int n, double* out, int wg_per_cu, int cus) {
  const int grid = cus * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<UN, PIPE, NT>), dim3(grid), dim3(256), 0, 0, S, lds, m, n, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  printf("  UN=%d pipe=%d nt=%d  %2d waves/CU: %.3f ms  %.2f TB/s\n", UN, (int)PIPE, (int)NT, wg_per_cu * 4, best, (double)m * n * 8 / best / 1e9);
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 50000, n = argc > 2 ? atoi(argv[2]) : 8192;
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  double *S, *out;
  hipMalloc(&S, (size_t)m * n * 8); hipMalloc(&out, (size_t)n * 8);
  hipMemset(S, 0, (size_t)m * n * 8);
  printf("wave-per-column read-only sweep: %d x %d doubles (%.2f GB), %d CUs\n", m, n, (double)m * n * 8 / 1e9, cus);
  for (int w : {2, 4, 8}) {
    run<8, false, false>(S, m, m, n, out, w, cus);
    run<8, true, false>(S, m, m, n, out, w, cus);
    run<8, true, true>(S, m, m, n, out, w, cus);
    run<4, true, true>(S, m, m, n, out, w, cus);
    run<16, false, true>(S, m, m, n, out, w, cus);
  }
  return 0;
}
