// Instruction issue-cost microbenchmark (gfx950): shader cycles (s_memtime) per wave-instruction
// per SIMD for the ops of the SpMM inner loop, and for the fused gather loop itself.
// hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define SDWA_LO(r, q) asm volatile("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(q))
#define SDWA_HI(r, q) asm volatile("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(q))
#define SDWA4_LO(r, q) asm volatile("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(q))
#define SDWA4_HI(r, q) asm volatile("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(q))
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f64x2 ld2;
#define LD2(o) (*(ld2*)(uintptr_t)(o))

__device__ const uint4* g_idx;   // index chunks for MODE 13 / 14: [chunk][lane] uint4 of valid 16-byte-entry ids

template <int MODE>
__global__ void k(double* out, unsigned* outu, unsigned long long* cyc, int iters) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  const double x = out[threadIdx.x & 7];
  const unsigned lane = threadIdx.x & 63;
  unsigned q0 = lane | ((lane + 64) << 16), q1 = (lane + 128) | ((lane + 192) << 16), q2 = (lane + 256) | ((lane + 320) << 16), q3 = (lane + 384) | ((lane + 448) << 16);
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
  f64x2 pv0 = {0, 0}, pv1 = {0, 0}, pv2 = {0, 0}, pv3 = {0, 0};
  uint4 ring0 = make_uint4(0, 0, 0, 0), ring1 = ring0, ring2 = ring0, ring3 = ring0;
  if (MODE == 13) {
    const uint4* ib0 = g_idx + ((size_t)(threadIdx.x >> 6) * 128) * 64 + lane;
    ring0 = ib0[0]; ring1 = ib0[64]; ring2 = ib0[128]; ring3 = ib0[192];
  }
  if (MODE == 14) {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(1))) const void* gptr;
    const uint4* ib0 = g_idx + ((size_t)(threadIdx.x >> 6) * 128) * 64 + lane;
    lds_u8* ring = (lds_u8*)(uintptr_t)(65536u + (threadIdx.x >> 6) * 4096u);
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gptr)(ib0 + j * 64), (__attribute__((address_space(3))) void*)(ring + j * 1024), 16, 0, 0);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                   "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    } else if (MODE == 1) {
      SDWA_LO(r0, q0); SDWA_HI(r1, q0); SDWA_LO(r2, q1); SDWA_HI(r3, q1); SDWA_LO(r4, q2); SDWA_HI(r5, q2); SDWA_LO(r6, q3); SDWA_HI(r7, q3);
    } else if (MODE == 2) {
      asm volatile("v_lshlrev_b32 %0, 3, %8\n v_lshlrev_b32 %1, 3, %9\n v_lshlrev_b32 %2, 3, %10\n v_lshlrev_b32 %3, 3, %11\n"
                   "v_lshlrev_b32 %4, 4, %8\n v_lshlrev_b32 %5, 4, %9\n v_lshlrev_b32 %6, 4, %10\n v_lshlrev_b32 %7, 4, %11\n"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
                   : "v"(q0), "v"(q1), "v"(q2), "v"(q3));
    } else if (MODE == 3) {
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0));
    } else if (MODE == 4) {  // ds_read_b64 only (conflict-free, fixed addresses), 8 per iter
      double v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %10\n ds_read_b64 %3, %11\n"
                   "ds_read_b64 %4, %8 offset:4096\n ds_read_b64 %5, %9 offset:4096\n ds_read_b64 %6, %10 offset:4096\n ds_read_b64 %7, %11 offset:4096\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(lane * 8), "v"(lane * 8 + 512), "v"(lane * 8 + 1024), "v"(lane * 8 + 1536));
      a0 += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;  // keeps them live; adds cost too (see MODE 5 for the loop proper)
    } else if (MODE == 5) {  // the fused inner loop: 8 x (sdwa, ds_read_b64), then 8 adds
      unsigned o0, o1, o2, o3, o4, o5, o6, o7;
      SDWA_LO(o0, q0); SDWA_HI(o1, q0); SDWA_LO(o2, q1); SDWA_HI(o3, q1); SDWA_LO(o4, q2); SDWA_HI(o5, q2); SDWA_LO(o6, q3); SDWA_HI(o7, q3);
      typedef __attribute__((address_space(3))) const double ld;
      const double v0 = *(ld*)(uintptr_t)o0, v1 = *(ld*)(uintptr_t)o1, v2 = *(ld*)(uintptr_t)o2, v3 = *(ld*)(uintptr_t)o3;
      const double v4 = *(ld*)(uintptr_t)o4, v5 = *(ld*)(uintptr_t)o5, v6 = *(ld*)(uintptr_t)o6, v7 = *(ld*)(uintptr_t)o7;
      a0 += v0; a1 += v1; a2 += v2; a3 += v3; a0 += v4; a1 += v5; a2 += v6; a3 += v7;
      q0 += 0x00010001u * (i & 1);  // keep the addresses loop-variant
    } else if (MODE == 7) {  // v_mul_f32_sdwa by 8.0 on a u16 half: (denormal) bits * 8 == integer << 3
      asm volatile("v_mul_f32_sdwa %0, %12, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                   "v_mul_f32_sdwa %1, %12, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                   "v_mul_f32_sdwa %2, %12, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                   "v_mul_f32_sdwa %3, %12, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                   "v_mul_f32_sdwa %4, %12, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                   "v_mul_f32_sdwa %5, %12, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                   "v_mul_f32_sdwa %6, %12, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                   "v_mul_f32_sdwa %7, %12, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
                   : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(8.0f));
    } else if (MODE == 8) {  // v_min_f64 / v_max_f64 pairs (compare-exchange of the bitonic sort)
      asm volatile("v_min_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                   "v_min_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    } else if (MODE == 9) {  // 64-bit integer compare + 2x v_cndmask pairs (the u64-key alternative)
      unsigned long long k0 = __double_as_longlong(a0), k1 = __double_as_longlong(a1);
      asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n"
                   "v_cmp_lt_u64 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n"
                   "v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %6, %6, %7, vcc\n"
                   : "+v"(k0), "+v"(k1), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) :: "vcc");
      a0 = __longlong_as_double(k0); a1 = __longlong_as_double(k1);
    } else if (MODE == 10) {  // ds_read_b128 only (16-byte entries, conflict-free per 16-lane group), 8 per iter
      f64x2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %9\n ds_read_b128 %2, %10\n ds_read_b128 %3, %11\n"
                   "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %9 offset:4096\n ds_read_b128 %6, %10 offset:4096\n ds_read_b128 %7, %11 offset:4096\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(lane * 16), "v"(lane * 16 + 1024), "v"(lane * 16 + 2048), "v"(lane * 16 + 3072));
      asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    } else if (MODE == 11) {  // pair loop: 8 x (sdwa, ds_read_b128), then 16 adds
      unsigned o0, o1, o2, o3, o4, o5, o6, o7;
      SDWA4_LO(o0, q0); SDWA4_HI(o1, q0); SDWA4_LO(o2, q1); SDWA4_HI(o3, q1); SDWA4_LO(o4, q2); SDWA4_HI(o5, q2); SDWA4_LO(o6, q3); SDWA4_HI(o7, q3);
      const f64x2 v0 = LD2(o0), v1 = LD2(o1), v2 = LD2(o2), v3 = LD2(o3), v4 = LD2(o4), v5 = LD2(o5), v6 = LD2(o6), v7 = LD2(o7);
      a0 += v0.x; a4 += v0.y; a1 += v1.x; a5 += v1.y; a2 += v2.x; a6 += v2.y; a3 += v3.x; a7 += v3.y;
      a0 += v4.x; a4 += v4.y; a1 += v5.x; a5 += v5.y; a2 += v6.x; a6 += v6.y; a3 += v7.x; a7 += v7.y;
      q0 += 0x00010001u * (i & 1);
    } else if (MODE == 12) {  // pair loop, half-chunk pipelined: next 4 reads in flight during 8 adds
      unsigned o0, o1, o2, o3;
      SDWA4_LO(o0, q2); SDWA4_HI(o1, q2); SDWA4_LO(o2, q3); SDWA4_HI(o3, q3);
      const f64x2 w0 = LD2(o0), w1 = LD2(o1), w2 = LD2(o2), w3 = LD2(o3);
      a0 += pv0.x; a4 += pv0.y; a1 += pv1.x; a5 += pv1.y; a2 += pv2.x; a6 += pv2.y; a3 += pv3.x; a7 += pv3.y;
      q0 += 0x00010001u * (i & 1);
      SDWA4_LO(o0, q0); SDWA4_HI(o1, q0); SDWA4_LO(o2, q1); SDWA4_HI(o3, q1);
      pv0 = LD2(o0); pv1 = LD2(o1); pv2 = LD2(o2); pv3 = LD2(o3);
      a0 += w0.x; a4 += w0.y; a1 += w1.x; a5 += w1.y; a2 += w2.x; a6 += w2.y; a3 += w3.x; a7 += w3.y;
    } else if (MODE == 13) {  // pair loop + the index chunk through VGPRs (global_load_dwordx4, 4 in flight), 4 chunks per trip
      const uint4* ib = g_idx + ((size_t)(threadIdx.x >> 6) * 128) * 64 + lane;
#define PAIR_CHUNK(q)                                                                                     \
      {                                                                                                    \
        unsigned o0, o1, o2, o3, o4, o5, o6, o7;                                                           \
        SDWA4_LO(o0, (q).x); SDWA4_HI(o1, (q).x); SDWA4_LO(o2, (q).y); SDWA4_HI(o3, (q).y);                 \
        SDWA4_LO(o4, (q).z); SDWA4_HI(o5, (q).z); SDWA4_LO(o6, (q).w); SDWA4_HI(o7, (q).w);                 \
        const f64x2 v0 = LD2(o0), v1 = LD2(o1), v2 = LD2(o2), v3 = LD2(o3), v4 = LD2(o4), v5 = LD2(o5), v6 = LD2(o6), v7 = LD2(o7); \
        a0 += v0.x; a4 += v0.y; a1 += v1.x; a5 += v1.y; a2 += v2.x; a6 += v2.y; a3 += v3.x; a7 += v3.y;   \
        a0 += v4.x; a4 += v4.y; a1 += v5.x; a5 += v5.y; a2 += v6.x; a6 += v6.y; a3 += v7.x; a7 += v7.y;   \
      }
      const int c0 = (i + 4) & 127;
      PAIR_CHUNK(ring0) ring0 = ib[((c0 + 0) & 127) * 64];
      PAIR_CHUNK(ring1) ring1 = ib[((c0 + 1) & 127) * 64];
      PAIR_CHUNK(ring2) ring2 = ib[((c0 + 2) & 127) * 64];
      PAIR_CHUNK(ring3) ring3 = ib[((c0 + 3) & 127) * 64];
      i += 3;
    } else if (MODE == 14) {  // the index chunk through LDS-DMA (global_load_lds_dwordx4) and one ds_read_b128, 4 chunks per trip
      typedef __attribute__((address_space(3))) unsigned char lds_u8;
      typedef __attribute__((address_space(1))) const void* gptr;
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) const u32x4_t lds_u4;
      const uint4* ib = g_idx + ((size_t)(threadIdx.x >> 6) * 128) * 64 + lane;
      const unsigned rb = 65536u + (threadIdx.x >> 6) * 4096u;
      const int c0 = (i + 4) & 127;
#define GLDS_CHUNK(slot)                                                                                  \
      {                                                                                                    \
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   /* the oldest DMA has landed */                  \
        const u32x4_t q = *(lds_u4*)(uintptr_t)(rb + (slot) * 1024u + lane * 16u);                          \
        PAIR_CHUNK(q)                                                                                      \
        __builtin_amdgcn_global_load_lds((gptr)(ib + ((c0 + (slot)) & 127) * 64),                           \
                                         (__attribute__((address_space(3))) void*)(lds_u8*)(uintptr_t)(rb + (slot) * 1024u), 16, 0, 0); \
      }
      GLDS_CHUNK(0) GLDS_CHUNK(1) GLDS_CHUNK(2) GLDS_CHUNK(3)
      i += 3;
    } else if (MODE == 6) {  // ds_read only, no adds: LDS issue rate
      double v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %10\n ds_read_b64 %3, %11\n"
                   "ds_read_b64 %4, %8 offset:4096\n ds_read_b64 %5, %9 offset:4096\n ds_read_b64 %6, %10 offset:4096\n ds_read_b64 %7, %11 offset:4096\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(lane * 8), "v"(lane * 8 + 512), "v"(lane * 8 + 1024), "v"(lane * 8 + 1536));
      asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + pv0.x + pv1.y + pv2.x + pv3.y + ring0.x + ring1.y + ring2.z + ring3.w;
  outu[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
  double* d; unsigned* u; unsigned long long* c;
  (void)hipMalloc(&d, 256 * 1024 * 8 * 2); (void)hipMalloc(&u, 256 * 1024 * 4 * 2); (void)hipMalloc(&c, 256 * 16 * 8);
  (void)hipMemset(d, 0, 256 * 1024 * 8);
  const int iters = 20000;
  static uint4* d_idx = nullptr;
  if (d_idx == nullptr) {   // 16 wavefront streams x 128 chunks x 1 KiB (2 MiB, L2-resident like the kernel's lists), ids < 4096 (64 KiB of 16-byte entries), conflict-free per 16-lane group
    const size_t nq = (size_t)16 * 132 * 64;
    std::vector<uint4> h(nq);
    for (size_t c = 0; c < nq / 64; ++c)
      for (unsigned l = 0; l < 64; ++l) {
        unsigned id[8];
        for (int e = 0; e < 8; ++e) id[e] = ((unsigned)((c * 8 + e) * 37u) & 0xff0u) | (l & 15u);   // slot = lane mod 16
        h[c * 64 + l] = make_uint4(id[0] | (id[1] << 16), id[2] | (id[3] << 16), id[4] | (id[5] << 16), id[6] | (id[7] << 16));
      }
    (void)hipMalloc(&d_idx, nq * 16);
    (void)hipMemcpy(d_idx, h.data(), nq * 16, hipMemcpyHostToDevice);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_idx), &d_idx, sizeof(d_idx));
  }
  const size_t smem = (MODE == 14) ? 65536 + 16 * 4096 : 65536;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<MODE><<<256, threads, smem>>>(d, u, c, iters);   // warm the clocks
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<256, threads, smem>>>(d, u, c, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * threads / 64);
  (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double wave_cycles = (double)h[h.size() / 2];
  const double wps = threads / 64.0 / 4.0;
  printf("%-12s %4d thr (%.0f waves/SIMD): %.3f ms, median wave %.0f cyc -> %.2f cyc per wave-instr-group-of-8 per wave, "
         "%.2f cyc per instr per SIMD, clock %.2f GHz\n", name, threads, wps, ms, wave_cycles, wave_cycles / iters,
         wave_cycles / iters / 8.0 / wps, wave_cycles / (ms * 1e-3) / 1e9);
  (void)hipFree(d); (void)hipFree(u); (void)hipFree(c);
}

int main() {
  // value check of the denormal trick is done by the kernel tests (addresses must match)

  for (int t : {1024}) {
    run<10>("ds_read_b128", t); run<11>("pair loop", t); run<12>("pair loop pipelined", t);
    run<13>("pair + idx VGPR", t); run<14>("pair + idx LDS-DMA", t);
  }
  if (getenv("UBENCH_ONLY_PAIR")) return 0;
  for (int t : {256, 512, 1024}) {
    run<0>("v_add_f64", t); run<1>("lshl_sdwa", t); run<2>("v_lshlrev", t); run<3>("v_add_f32", t);
    run<8>("min/max_f64", t); run<9>("cmp_u64+cnd", t); run<7>("mul_f32_sdwa", t); run<6>("ds_read_b64", t); run<4>("ds_read+adds", t); run<5>("fused loop", t);
  }
  return 0;
}
