// Does the sticky IEEE "inexact" bit of the wave's TRAPSTS register (EXCP[5]) record a rounded v_add_f64 on gfx950 with
// exception traps disabled?  If it does, the u16 staging of the rank crossprod (x + 2^51: exact iff x is a half-integer)
// can tell "some staged value was not a rank" for free: clear the bit, stage, read the bit.
//   hipcc --offload-arch=gfx950 -O3 inexact_flag.hip -o inexact_flag ; ./inexact_flag
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(const double* x, int n, unsigned* out, double* sink) {
  const int t = threadIdx.x;
  // clear EXCP (TRAPSTS bits 0..8)
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory");
  double acc = 0.0;
  for (int i = t; i < n; i += blockDim.x) {
    double y;
    asm volatile("v_add_f64 %0, %1, %2" : "=v"(y) : "v"(x[i]), "v"(0x1p51));
    acc = __longlong_as_double(__double_as_longlong(acc) ^ __double_as_longlong(y));
  }
  sink[blockIdx.x * blockDim.x + t] = acc;
  unsigned ts;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(ts) :: "memory");
  if ((t & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + t / 64] = ts;
}

int main() {
  const int n = 4096;
  double *x, *sink;
  unsigned* out;
  (void)hipMallocManaged(&x, n * 8);
  (void)hipMallocManaged(&sink, 256 * 8);
  (void)hipMallocManaged(&out, 64);
  const char* names[] = {"half-integers (exact)", "one value 8.1 in lane 5 (wave 0 only)", "all 8.1", "one NaN", "one -3.0", "one +Inf"};
  for (int c = 0; c < 6; ++c) {
    for (int i = 0; i < n; ++i) x[i] = 0.5 * (i % 40000);
    if (c == 1) x[5] = 8.1;
    if (c == 2) for (int i = 0; i < n; ++i) x[i] = 8.1;
    if (c == 3) x[7] = __builtin_nan("");
    if (c == 4) x[7] = -3.0;
    if (c == 5) x[7] = __builtin_inf();
    for (int i = 0; i < 4; ++i) out[i] = 0xdead;
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, x, n, out, sink);
    (void)hipDeviceSynchronize();
    printf("%-44s TRAPSTS.EXCP per wave: %03x %03x %03x %03x   (bit 5 = inexact, bit 0 = invalid)\n", names[c], out[0], out[1], out[2], out[3]);
  }
  return 0;
}
