// Bare pattern of the sparse scatter crossprod (spmm_scatter_csc_f64, R/plaid.R:107 with a dgCMatrix y): 256-byte id
// segments out of a 17 MB list (L2 / Infinity-Cache resident) -> two ds_add_u64 wave-instructions per segment into the
// accumulators of one chunk of sets, ~1,100 segments per (column, chunk) ITEM, a chunk epilogue per item.  Two structures on
// the same synthetic work and with the same epilogue, so that what the structure alone is worth can be read off:
//   cur   every one of the 16 wavefronts loads its own segments (48 loads in flight) and applies them; workgroup barrier,
//         epilogue, barrier (the product kernel's structure)
//   ring  4 producer wavefronts fetch segments by LDS-DMA (buffer_load ... lds) into a ring of 4 x 28 slots in LDS and run
//         ahead across item boundaries; 12 consumer wavefronts read ids from the ring and issue the atomics; per-item
//         barriers in software, among the consumers only
// and two floors: the atomics alone (ids in registers) and the epilogue alone.
//   hipcc --offload-arch=gfx950 -O3 scatter_ring.hip -o scatter_ring && ./scatter_ring [columns per CU] [epilogue 0|1|2]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
__device__ int32_t raw_buffer_load_i32(i32x4 rsrc, int32_t voffset, int32_t soffset, int32_t aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ __forceinline__ i32x4 make_raw_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  i32x4 r;
  r.x = (int32_t)(uint32_t)a;
  r.y = (int32_t)((uint32_t)(a >> 32) & 0xffffu);
  r.z = (int32_t)bytes;
  r.w = 0x00020000;
  return r;
}
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void lds_add_u64(uint32_t addr, unsigned long long v) {
  __hip_atomic_fetch_add(reinterpret_cast<lds_u64*>(static_cast<uintptr_t>(addr)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t off_lo(uint32_t q) {
  uint32_t r;
  asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(q));
  return r;
}
__device__ __forceinline__ uint32_t off_hi(uint32_t q) { return (q >> 16) << 3; }

constexpr int kCh = 16667;        // sets per chunk (50,000 sets = 3 chunks)
constexpr int kTrash = 64;
constexpr int kListCap = 1280;    // entries per item list (padded)

struct Args {
  const uint16_t* ids;      // segments of 128 u16 ids
  uint32_t ids_bytes;
  const int32_t* work;      // [wg][item][kListCap] segment numbers
  const int32_t* cnt;       // [wg][item] entries of the item
  int items;
  int epi;                  // 0 none | 1 LDS read + zero | 2 + 8-byte streaming stores
  unsigned long long* S;    // [wg][item][kCh] (epi 2) -- or one row per wg reused when S_rows == 1
  int S_per_item;           // 1: every item has its own row (checked), 0: one row per workgroup (timing at full length)
  unsigned long long* cyc;  // per workgroup: cycles of the whole loop
  unsigned long long* chk;  // per workgroup: checksum of all accumulators over all items (epi < 2)
  // what the product moves besides the ids (emulated): extra & 1: the epilogue loads a 16-byte factor pair per set (extra & 4:
  // 8 bytes) from a per-chunk table; extra & 2: a segment number is looked up per entry in an 88 KB per-chunk table (gather)
  int extra;
  const double* kw;         // [3][kCh][2]
  const int32_t* segtab;    // identity table over the segment numbers
  const unsigned short* sz; // [3][kCh] set sizes (here: 1)
  int cols_per_cu;
};

// ------------------------------------------------------------------------------------------------------------------
// the chunk epilogue both structures share: every accumulator is read, zeroed, scaled and (epi 2) stored
template <int NT>
__device__ __forceinline__ void epilogue(const Args& a, int t, int item, unsigned long long* acc, uint32_t acc_base_words,
                                         unsigned long long& chk) {
  (void)acc_base_words;
  unsigned long long* row = a.S + ((size_t)blockIdx.x * (a.S_per_item ? a.items : 1) + (a.S_per_item ? item : 0)) * kCh;
  for (int i = t; i < kCh; i += NT) {
    const unsigned long long b = acc[i];
    acc[i] = 0ull;
    if (a.epi == 2) {
      // u64 -> double as the product does (one fma), a scale, and the bits out as a streaming store
      double f0 = 0.5, f1 = 0.0;
      if (a.extra & 1) { const double* kp = a.kw + ((size_t)(item / a.cols_per_cu) * kCh + i) * 2; f0 = kp[0]; f1 = kp[1]; }
      else if (a.extra & 4) { f0 = a.kw[((size_t)(item / a.cols_per_cu) * kCh + i) * 2]; }
      const double s = __fma_rn((double)(uint32_t)(b >> 32), 4294967296.0, (double)(uint32_t)b) * f0 + f1;
      __builtin_nontemporal_store((unsigned long long)__double_as_longlong(s), &row[i]);
    } else {
      chk += b * (unsigned long long)(i + 1);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// cur: the product's structure.  Entries of an item are dealt to the wavefronts in rounds of 1,024 (balanced), 64 per
// wavefront and round, applied through a static pipeline with 48 loads in flight.
template <bool STAMP>
__global__ void __launch_bounds__(1024) walk_cur(Args a, unsigned long long* stamps) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < kCh + kTrash; i += 1024) acc[i] = 0ull;
  __syncthreads();
  const i32x4 rsrc = make_raw_rsrc(a.ids, a.ids_bytes);
  const int32_t loff = lane * 4;
  const int32_t dummy = (int32_t)(a.ids_bytes / 256) - 1;
  unsigned long long chk = 0;
  unsigned long long ph[4] = {0, 0, 0, 0}, tl = 0;   // cycles: item start -> first ids in hand | -> walk done | -> behind barrier 1 | -> behind barrier 2
#define STAMP_AT(k) if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; __builtin_amdgcn_sched_barrier(0); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  tl = t0;
  for (int item = 0; item < a.items; ++item) {
    const int32_t* list = a.work + ((size_t)blockIdx.x * a.items + item) * kListCap;
    const int n = a.cnt[(size_t)blockIdx.x * a.items + item];
    const int nr = (n + 1023) / 1024 > 1 ? (n + 1023) / 1024 : 1;
    const int per = nr == 1 ? n : (n + nr - 1) / nr;
    for (int r = 0; r < nr; ++r) {
      const int lo = r * per;
      const int c = n - lo < per ? n - lo : per;
      const int i_ = (((lane >> 4) * 16 + wave) << 4) + (lane & 15);
      const bool have = i_ < c;
      int s0e = have ? list[lo + i_] : dummy;
      if ((a.extra & 2) && have) s0e = a.segtab[s0e];
      const unsigned long long v = have ? (unsigned long long)(lo + i_ + 1) : 0ull;
      const uint32_t vlo = (uint32_t)v, vhi = (uint32_t)(v >> 32);
      if (__ballot(have) == 0ull) continue;
      const int nval = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__ballot(have)));
      constexpr int HW = 16;
      uint32_t idA[HW], idB[HW], idC[HW], idD[HW];
#define ID_LOAD(u) ((uint32_t)raw_buffer_load_i32(rsrc, loff, (int32_t)((uint32_t)__builtin_amdgcn_readlane(s0e, (u)) << 8), 0))
#define V_OF(u) (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)vhi, (u)) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)vlo, (u)))
#define SCATTER2(id2, val) { const unsigned long long val_ = (val); lds_add_u64(off_lo(id2), val_); lds_add_u64(off_hi(id2), val_); }
#pragma unroll
      for (int u = 0; u < HW; ++u) idA[u] = ID_LOAD(u);
#pragma unroll
      for (int u = 0; u < HW; ++u) idB[u] = ID_LOAD(HW + u);
#pragma unroll
      for (int u = 0; u < HW; ++u) idC[u] = ID_LOAD(2 * HW + u);
      if constexpr (STAMP) { asm volatile("" : "+v"(idA[0])); STAMP_AT(0) }
#pragma unroll
      for (int u = 0; u < HW; ++u) SCATTER2(idA[u], V_OF(u))
#pragma unroll
      for (int u = 0; u < HW; ++u) idD[u] = ID_LOAD(3 * HW + u);
      if (nval > HW) {
#pragma unroll
        for (int u = 0; u < HW; ++u) SCATTER2(idB[u], V_OF(HW + u))
      }
      if (nval > 2 * HW) {
#pragma unroll
        for (int u = 0; u < HW; ++u) SCATTER2(idC[u], V_OF(2 * HW + u))
      }
      if (nval > 3 * HW) {
#pragma unroll
        for (int u = 0; u < HW; ++u) SCATTER2(idD[u], V_OF(3 * HW + u))
      }
#undef ID_LOAD
#undef V_OF
    }
    if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); STAMP_AT(1) }
    if (a.extra & 24) {
      // the product's factor traffic, requested BEFORE the barrier as the product does: extra & 8: a 16-byte pair per set
      // (267 KB per item), extra & 16: a u16 per set (33 KB per item)
      constexpr int NE = (kCh + 1023) / 1024;
      double f0[NE];
      const int chunk = item / a.cols_per_cu;
      if (a.extra & 8) {
        f64x2 kwv[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) { const int i = tid + u * 1024; kwv[u] = reinterpret_cast<const f64x2*>(a.kw)[(size_t)chunk * kCh + (i < kCh ? i : kCh - 1)]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NE; ++u) f0[u] = kwv[u].x + kwv[u].y;
      } else {
        unsigned short szv[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) { const int i = tid + u * 1024; szv[u] = a.sz[(size_t)chunk * kCh + (i < kCh ? i : kCh - 1)]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NE; ++u) f0[u] = (double)szv[u];
      }
      STAMP_AT(2)
      unsigned long long* row = a.S + ((size_t)blockIdx.x * a.items + item) * kCh;
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const int i = tid + u * 1024;
        if (i < kCh) {
          const unsigned long long b = acc[i];
          acc[i] = 0ull;
          const double sc = __fma_rn((double)(uint32_t)(b >> 32), 4294967296.0, (double)(uint32_t)b) * f0[u];
          __builtin_nontemporal_store((unsigned long long)__double_as_longlong(sc), &row[i]);
        }
      }
    } else {
      __syncthreads();
      STAMP_AT(2)
      if (a.epi != 0) epilogue<1024>(a, tid, item, acc, 0, chk);
    }
    __syncthreads();
    STAMP_AT(3)
  }
  if constexpr (STAMP) {
    if (lane == 0) for (int k = 0; k < 4; ++k) stamps[((size_t)blockIdx.x * 16 + wave) * 4 + k] = ph[k];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (a.epi == 0)
    for (int i = tid; i < kCh; i += 1024) chk += acc[i] * (unsigned long long)(i + 1);
  for (int off = 32; off >= 1; off >>= 1) chk += __shfl_xor(chk, off, 64);
  if (lane == 0) atomicAdd(&a.chk[blockIdx.x], chk);
  if (tid == 0) a.cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------------------------
// steal: the item's entries {segment, value} sit in a table in LDS (behind the accumulators) and the wavefronts TAKE groups
// of 16 of them as they go (one returning LDS atomic per group), two groups in flight per wavefront -- the wavefronts the
// memory pipe serves first simply take more groups, so nobody waits at the end-of-walk barrier for the ones served last.
// The loads stay unconditional (hipcc counts vmcnt exactly): a wavefront that finds the table empty issues one or two
// groups of loads of the all-padding segment (L1 hits) before it leaves the loop.
constexpr int kEntOff = (kCh + kTrash) * 8;                 // entry table: u32 segment per entry (the value is index + 1 here)
constexpr int kEntCap = 1536;
constexpr int kGrabOff = kEntOff + kEntCap * 4;
static_assert(kGrabOff + 16 <= 160 * 1024, "LDS");
template <bool STAMP>
__global__ void __launch_bounds__(1024) walk_steal(Args a, unsigned long long* stamps) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);
  uint32_t* ent = reinterpret_cast<uint32_t*>(smem + kEntOff);
  uint32_t* grab = reinterpret_cast<uint32_t*>(smem + kGrabOff);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < kCh + kTrash; i += 1024) acc[i] = 0ull;
  __syncthreads();
  const i32x4 rsrc = make_raw_rsrc(a.ids, a.ids_bytes);
  const int32_t loff = lane * 4;
  const int32_t dummy = (int32_t)(a.ids_bytes / 256) - 1;
  unsigned long long chk = 0;
  unsigned long long ph[4] = {0, 0, 0, 0}, tl = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  tl = t0;
  for (int item = 0; item < a.items; ++item) {
    const int32_t* list = a.work + ((size_t)blockIdx.x * a.items + item) * kListCap;
    const int n = a.cnt[(size_t)blockIdx.x * a.items + item];
    // publish the entries (the product's threads would publish their own {segment, value} here)
    for (int e = tid; e < n; e += 1024) ent[e] = (uint32_t)list[e];
    if (tid == 0) *grab = 0u;
    __syncthreads();
    STAMP_AT(0)
    const int ngroups = (n + 15) >> 4;
    uint32_t idA[16], idB[16];
    int segA = dummy, segB = dummy, eA = 0, eB = 0;
    bool vA, vB;
#define TAKE_GROUP(SEG, E0, VALID)                                                                                   \
    {                                                                                                                 \
      uint32_t g_ = 0;                                                                                                \
      if (lane == 0) g_ = __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(kGrabOff)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      g_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_);                                                         \
      VALID = (int)g_ < ngroups;                                                                                      \
      E0 = (int)g_ * 16;                                                                                              \
      const int e_ = E0 + (lane & 15);                                                                                \
      SEG = (VALID && e_ < n) ? (int)ent[e_ < kEntCap ? e_ : 0] : dummy;                                              \
    }
#define ISSUE_GROUP(ID, SEG)                                                                                         \
    _Pragma("unroll") for (int u = 0; u < 16; ++u)                                                                  \
      ID[u] = (uint32_t)raw_buffer_load_i32(rsrc, loff, (int32_t)((uint32_t)__builtin_amdgcn_readlane(SEG, u) << 8), 0);
#define APPLY_GROUP(ID, E0)                                                                                          \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                                                \
      const unsigned long long val_ = (E0 + u < n) ? (unsigned long long)(E0 + u + 1) : 0ull;                        \
      lds_add_u64(off_lo(ID[u]), val_);                                                                               \
      lds_add_u64(off_hi(ID[u]), val_);                                                                               \
    }
    TAKE_GROUP(segA, eA, vA)
    ISSUE_GROUP(idA, segA)
    TAKE_GROUP(segB, eB, vB)
    ISSUE_GROUP(idB, segB)
    while (vA) {
      APPLY_GROUP(idA, eA)
      TAKE_GROUP(segA, eA, vA)
      ISSUE_GROUP(idA, segA)
      if (!vB) break;
      APPLY_GROUP(idB, eB)
      TAKE_GROUP(segB, eB, vB)
      ISSUE_GROUP(idB, segB)
    }
    // (a wavefront leaves with its last one or two groups of padding loads in flight; they are awaited here so that
    // the register arrays are dead at the top of the next item)
    asm volatile("" : "+v"(idA[0]), "+v"(idB[0]));
#undef TAKE_GROUP
#undef ISSUE_GROUP
#undef APPLY_GROUP
    if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); STAMP_AT(1) }
    __syncthreads();
    STAMP_AT(2)
    if (a.epi != 0) epilogue<1024>(a, tid, item, acc, 0, chk);
    __syncthreads();
    STAMP_AT(3)
  }
  if constexpr (STAMP) {
    if (lane == 0) for (int k = 0; k < 4; ++k) stamps[((size_t)blockIdx.x * 16 + wave) * 4 + k] = ph[k];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (a.epi == 0)
    for (int i = tid; i < kCh; i += 1024) chk += acc[i] * (unsigned long long)(i + 1);
  for (int off = 32; off >= 1; off >>= 1) chk += __shfl_xor(chk, off, 64);
  if (lane == 0) atomicAdd(&a.chk[blockIdx.x], chk);
  if (tid == 0) a.cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------------------------
// quad: the same structure as `cur`, but one vector-memory instruction fetches FOUR segments: a buffer_load_dwordx4 whose
// 16-lane rows read 256 bytes each (16 bytes per lane), row q of step t taking the segment of lane 16 q + t (its number and
// its value reach the row by DPP row_newbcast -- no scalar registers, no readlane).  A lane then holds 8 ids of its row's
// segment and the row's 16 lanes apply them with 8 atomic wave-instructions, each row with its own value: the same 2 atomic
// instructions per segment, a quarter of the vector-memory instructions (the texture-addresser takes ~13-16 cycles per
// wave-instruction whether it moves 256 bytes or 1 KiB).  The id lists are laid out for it: position 8 r + j of a segment is
// what lane r of the row adds in instruction j, and the 16 positions {8 r + j} sit on 16 different banks.
typedef int32_t i32x4v __attribute__((ext_vector_type(4)));
__device__ i32x4v raw_buffer_load_i32x4(i32x4 rsrc, int32_t voffset, int32_t soffset, int32_t aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
#define BCAST16(x, t) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x150 + (t), 0xf, 0xf, true))
template <bool STAMP>
__global__ void __launch_bounds__(1024) walk_quad(Args a, const uint16_t* ids4, unsigned long long* stamps) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < kCh + kTrash; i += 1024) acc[i] = 0ull;
  __syncthreads();
  const i32x4 rsrc = make_raw_rsrc(ids4, a.ids_bytes);
  const uint32_t r16 = (uint32_t)(lane & 15) * 16u;
  const int32_t dummy = (int32_t)(a.ids_bytes / 256) - 1;
  unsigned long long chk = 0;
  unsigned long long ph[4] = {0, 0, 0, 0}, tl = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  tl = t0;
  for (int item = 0; item < a.items; ++item) {
    const int32_t* list = a.work + ((size_t)blockIdx.x * a.items + item) * kListCap;
    const int n = a.cnt[(size_t)blockIdx.x * a.items + item];
    const int nr = (n + 1023) / 1024 > 1 ? (n + 1023) / 1024 : 1;
    const int per = nr == 1 ? n : (n + nr - 1) / nr;
    for (int r = 0; r < nr; ++r) {
      const int lo = r * per;
      const int c = n - lo < per ? n - lo : per;
      const int i_ = (((lane >> 4) * 16 + wave) << 4) + (lane & 15);
      const bool have = i_ < c;
      int s0q = have ? list[lo + i_] : dummy;
      if ((a.extra & 2) && have) s0q = a.segtab[s0q];
      const uint32_t segoff = (uint32_t)s0q << 8;
      const unsigned long long v = have ? (unsigned long long)(lo + i_ + 1) : 0ull;
      const uint32_t vlo = (uint32_t)v, vhi = (uint32_t)(v >> 32);
      const uint64_t bal = __ballot(have);
      if (bal == 0ull) continue;
      const int nsteps = __builtin_amdgcn_readfirstlane(__builtin_popcount((uint32_t)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xffffull)));
      i32x4v q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, q10, q11, q12, q13, q14, q15;
#define QLOAD(t) q##t = raw_buffer_load_i32x4(rsrc, (int32_t)(BCAST16(segoff, t) + r16), 0, 0);
#define QADD2(w, val) { lds_add_u64(off_lo((uint32_t)(w)), val); lds_add_u64(off_hi((uint32_t)(w)), val); }
#define QAPPLY(t) { const unsigned long long val_ = ((unsigned long long)BCAST16(vhi, t) << 32) | BCAST16(vlo, t); \
      QADD2(q##t.x, val_) QADD2(q##t.y, val_) QADD2(q##t.z, val_) QADD2(q##t.w, val_) }
      QLOAD(0) QLOAD(1) QLOAD(2) QLOAD(3) QLOAD(4) QLOAD(5) QLOAD(6) QLOAD(7) QLOAD(8) QLOAD(9) QLOAD(10) QLOAD(11)
      if constexpr (STAMP) { asm volatile("" : "+v"(q0)); STAMP_AT(0) }
      QAPPLY(0) QAPPLY(1) QAPPLY(2) QAPPLY(3)
      QLOAD(12) QLOAD(13) QLOAD(14) QLOAD(15)
      if (nsteps > 4) { QAPPLY(4) QAPPLY(5) QAPPLY(6) QAPPLY(7) }
      if (nsteps > 8) { QAPPLY(8) QAPPLY(9) QAPPLY(10) QAPPLY(11) }
      if (nsteps > 12) { QAPPLY(12) QAPPLY(13) QAPPLY(14) QAPPLY(15) }
#undef QLOAD
#undef QAPPLY
#undef QADD2
    }
    if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); STAMP_AT(1) }
    __syncthreads();
    STAMP_AT(2)
    if (a.epi != 0) epilogue<1024>(a, tid, item, acc, 0, chk);
    __syncthreads();
    STAMP_AT(3)
  }
  if constexpr (STAMP) {
    if (lane == 0) for (int k = 0; k < 4; ++k) stamps[((size_t)blockIdx.x * 16 + wave) * 4 + k] = ph[k];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (a.epi == 0)
    for (int i = tid; i < kCh; i += 1024) chk += acc[i] * (unsigned long long)(i + 1);
  for (int off = 32; off >= 1; off >>= 1) chk += __shfl_xor(chk, off, 64);
  if (lane == 0) atomicAdd(&a.chk[blockIdx.x], chk);
  if (tid == 0) a.cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------------------------
// ring: producers and consumers.  LDS: [0, kRingBytes) the four teams' id rings, then the groups' metadata and the control
// words, then the accumulators (their base goes into the atomics' 16-bit offset field, so an id << 3 is still the address
// operand; the ring sits low because the LDS-DMA base register M0 holds 16 address bits).
constexpr int kTeams = 4, kConsPerTeam = 3, kGrp = 4;     // entries per group: the unit of hand-over
constexpr int kNG = 7;                                    // groups per team ring
constexpr int kRS = kNG * kGrp;                           // 28 slots of 256 bytes per team
constexpr int kRingBytes = kTeams * kRS * 256;            // 28,672
constexpr int kMetaStride = 40;                           // per group: 4 x u64 value, u64 flag
constexpr int kMetaOff = kRingBytes;
constexpr int kCtrlOff = kMetaOff + kTeams * kNG * kMetaStride;   // + 1,120
// control words (u32): full[4] | nfree[4][4] | bar
constexpr int kFullOff = kCtrlOff;
constexpr int kFreeOff = kCtrlOff + 16;
constexpr int kBarOff = kCtrlOff + 16 + 64;
constexpr int kAccOff = ((kCtrlOff + 16 + 64 + 4 + 7) / 8) * 8;
static_assert(kAccOff < 65536, "the accumulators' base must fit the DS offset field");
static_assert(kAccOff + (kCh + kTrash) * 8 <= 160 * 1024, "LDS");
constexpr unsigned long long kFlagEnd = 1ull;
constexpr uint32_t kSpinLimit = 4000000u;

__device__ __forceinline__ uint32_t lds_read_u32(uint32_t addr) {
  return __hip_atomic_load(reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(addr)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_write_u32(uint32_t addr, uint32_t v) {
  __hip_atomic_store(reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(addr)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// ds_add_u64 at (acc base + id * 8): the base rides in the instruction's offset field
__device__ __forceinline__ void acc_add(uint32_t idoff, unsigned long long v) {
  asm volatile("ds_add_u64 %0, %1 offset:%2" : : "v"(idoff), "v"(v), "i"(kAccOff) : "memory");
}

template <int DG>   // groups a producer keeps in flight
__global__ void __launch_bounds__(1024) walk_ring(Args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    if ((uint32_t)(uintptr_t)((lds_u8*)smem) != 0u) __builtin_trap();
  }
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem + kAccOff);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (kAccOff + (kCh + kTrash) * 8) / 8; i += 1024) reinterpret_cast<unsigned long long*>(smem)[i] = 0ull;
  __syncthreads();
  unsigned long long chk = 0;
  uint32_t spin = 0;   // (a protocol error traps instead of hanging the box)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < kTeams) {
    // ---------------- producer of team `wave`
    const int p = wave;
    const i32x4 rsrc = make_raw_rsrc(a.ids, a.ids_bytes);
    const uint32_t voff = lane * 4;
    const uint32_t ring_base = p * kRS * 256;
    const uint32_t meta_base = kMetaOff + p * kNG * kMetaStride;
    uint32_t G = 0;          // groups issued
    uint32_t nf[kConsPerTeam] = {0, 0, 0};   // groups each consumer of the team has finished, as last read
// room in the ring for group G: group G - kNG -- the (G - kNG) / 3-th group of consumer (G - kNG) % 3 -- must be finished
#define WAIT_ROOM()                                                                                          \
  if (G >= (uint32_t)kNG) {                                                                                  \
    const uint32_t gp_ = G - kNG, q_ = gp_ / 3u, r_ = gp_ - 3u * q_;                                         \
    uint32_t have_ = r_ == 0 ? nf[0] : (r_ == 1 ? nf[1] : nf[2]);                                            \
    while (have_ <= q_) {                                                                                    \
      have_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(kFreeOff + p * 16 + r_ * 4));      \
      if (have_ <= q_) { __builtin_amdgcn_s_sleep(1); if (++spin > kSpinLimit) __builtin_trap(); }                \
    }                                                                                                        \
    if (r_ == 0) nf[0] = have_; else if (r_ == 1) nf[1] = have_; else nf[2] = have_;                         \
    asm volatile("" : : : "memory");                                                                         \
  }
    for (int item = 0; item < a.items; ++item) {
      const int32_t* list = a.work + ((size_t)blockIdx.x * a.items + item) * kListCap;
      const int n = a.cnt[(size_t)blockIdx.x * a.items + item];
      const int per = (n + kTeams - 1) / kTeams;
      const int lo = p * per;
      const int mine = n - lo < per ? (n - lo > 0 ? n - lo : 0) : per;
      int seg_next = 0;
      if (mine > 0) {   // (the first batch of an item: its list behind the previous item's drain -- the consumers are in their epilogue)
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(seg_next) : "v"(list + lo + (lane < mine ? lane : 0)) : "memory");
      }
      for (int b0 = 0; b0 < mine; b0 += 64) {
        const int c = mine - b0 < 64 ? mine - b0 : 64;
        const bool have = lane < c;
        const int seg = have ? seg_next : 0;
        const bool more = b0 + 64 < mine;
        if (more) {   // the next batch's segment numbers: requested before this batch's DMAs, taken behind them
          const int nx = b0 + 64 + lane < mine ? b0 + 64 + lane : mine - 1;
          asm volatile("global_load_dword %0, %1, off" : "=v"(seg_next) : "v"(list + lo + nx) : "memory");
        }
        const unsigned long long v = have ? (unsigned long long)(lo + b0 + lane + 1) : 0ull;
        const int ngr = (c + kGrp - 1) / kGrp;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          if (g < ngr) {
            // room in the ring for group G: group G - kNG must have been finished
            WAIT_ROOM()
            const uint32_t gslot = G % kNG;
#pragma unroll
            for (int k = 0; k < kGrp; ++k) {
              const uint32_t s = (uint32_t)__builtin_amdgcn_readlane(seg, g * kGrp + k) << 8;
              const uint32_t dst = ring_base + (gslot * kGrp + k) * 256;
              asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dword %1, %2, %3 offen lds"
                           : : "s"(dst), "v"(voff), "s"(rsrc), "s"(s) : "memory");
            }
            // the group's values and a cleared flag
            if ((lane >> 2) == g) {
              *reinterpret_cast<lds_u64*>(static_cast<uintptr_t>(meta_base + gslot * kMetaStride + (lane & 3) * 8)) = v;
              if ((lane & 3) == 0)
                *reinterpret_cast<lds_u64*>(static_cast<uintptr_t>(meta_base + gslot * kMetaStride + 32)) = 0ull;
            }
            ++G;
            // all but the youngest 4 DG loads have landed: groups < G - DG are complete
            asm volatile("s_waitcnt vmcnt(%0)" : : "i"(kGrp * DG) : "memory");
            if (G > (uint32_t)DG && lane == 0) lds_write_u32(kFullOff + p * 4, G - DG);
          }
        }
        if (more) {   // (older than every DMA of this batch: landed once kGrp * DG younger loads exist, else wait for all)
          if (ngr >= DG) asm volatile("" : "+v"(seg_next) : : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" : "+v"(seg_next) : : "memory");
        }
      }
      // end of the item: every load landed, then one END group per consumer
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
      if (lane == 0) lds_write_u32(kFullOff + p * 4, G);
      for (int e = 0; e < kConsPerTeam; ++e) {
        WAIT_ROOM()
        const uint32_t gslot = G % kNG;
        if (lane == 0) *reinterpret_cast<lds_u64*>(static_cast<uintptr_t>(meta_base + gslot * kMetaStride + 32)) = kFlagEnd;
        asm volatile("" : : : "memory");
        ++G;
        if (lane == 0) lds_write_u32(kFullOff + p * 4, G);
      }
    }
  } else {
    // ---------------- consumer j of team p
    const int cw = wave - kTeams;            // 0..11
    const int p = cw / kConsPerTeam, j = cw % kConsPerTeam;
    const uint32_t ring_base = p * kRS * 256;
    const uint32_t meta_base = kMetaOff + p * kNG * kMetaStride;
    const int ct = cw * 64 + lane;           // epilogue thread index among the 768 consumer threads
    uint32_t G = j, full = 0, done = 0, nbar = 0;
    for (int item = 0; item < a.items; ++item) {
      for (;;) {
        while ((int32_t)full <= (int32_t)G) {
          full = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(kFullOff + p * 4));
          if ((int32_t)full <= (int32_t)G) { __builtin_amdgcn_s_sleep(1); if (++spin > kSpinLimit) __builtin_trap(); }
        }
        asm volatile("" : : : "memory");
        const uint32_t gslot = G % kNG;
        const unsigned long long mv = *reinterpret_cast<lds_u64*>(
            static_cast<uintptr_t>(meta_base + gslot * kMetaStride + (lane < 5 ? lane : 4) * 8));
        const uint32_t mlo = (uint32_t)mv, mhi = (uint32_t)(mv >> 32);
        const bool end = (uint32_t)__builtin_amdgcn_readlane((int)mlo, 4) == (uint32_t)kFlagEnd;
        if (!end) {
          uint32_t id[kGrp];
#pragma unroll
          for (int k = 0; k < kGrp; ++k)
            id[k] = *reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(ring_base + (gslot * kGrp + k) * 256 + lane * 4));
#pragma unroll
          for (int k = 0; k < kGrp; ++k) {
            const unsigned long long val = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mhi, k) << 32) |
                                           (uint32_t)__builtin_amdgcn_readlane((int)mlo, k);
            acc_add(off_lo(id[k]), val);
            acc_add(off_hi(id[k]), val);
          }
        }
        ++done;
        if (lane == 0) lds_write_u32(kFreeOff + p * 16 + j * 4, done);   // (behind this wavefront's reads of the slot: DS operations of a wavefront are in order)
        G += kConsPerTeam;
        if (end) break;
      }
      // barrier among the 12 consumers (the producers never wait at a barrier: they run ahead)
#define CONS_BARRIER()                                                                                     \
  {                                                                                                        \
    ++nbar;                                                                                                \
    if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(kBarOff)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    while ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)lds_read_u32(kBarOff)) - nbar * 12u) < 0) { __builtin_amdgcn_s_sleep(1); if (++spin > kSpinLimit) __builtin_trap(); } \
    asm volatile("" : : : "memory");                                                                       \
  }
      CONS_BARRIER()
      if (a.epi != 0) epilogue<768>(a, ct, item, acc, 0, chk);
      CONS_BARRIER()
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (a.epi == 0)
    for (int i = tid; i < kCh; i += 1024) chk += acc[i] * (unsigned long long)(i + 1);
  for (int off = 32; off >= 1; off >>= 1) chk += __shfl_xor(chk, off, 64);
  if (lane == 0) atomicAdd(&a.chk[blockIdx.x], chk);
  if (tid == 0) a.cyc[blockIdx.x] = t1 - t0;
}

// floors: (1) the atomics alone -- the ids of 32 segments in registers, applied round and round, no loads, no barriers;
// (2) the epilogue alone
__global__ void __launch_bounds__(1024) atomics_only(Args a, int pairs_per_wave) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < kCh + kTrash; i += 1024) acc[i] = 0ull;
  __syncthreads();
  const uint32_t* idw = reinterpret_cast<const uint32_t*>(a.ids);
  uint32_t id[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) id[q] = idw[((size_t)((blockIdx.x * 16 + (tid >> 6)) * 32 + q) % (size_t)(a.ids_bytes / 256 - 1)) * 64 + lane];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < pairs_per_wave / 32; ++it) {
#pragma unroll
    for (int q = 0; q < 32; ++q) { lds_add_u64(off_lo(id[q]), 1ull); lds_add_u64(off_hi(id[q]), 1ull); }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) a.cyc[blockIdx.x] = t1 - t0;
}
template <int NT>
__global__ void __launch_bounds__(1024) epilogue_only(Args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < kCh + kTrash; i += 1024) acc[i] = 1ull;
  __syncthreads();
  unsigned long long chk = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int item = 0; item < a.items; ++item) {
    if (tid < NT) epilogue<NT>(a, tid, item, acc, 0, chk);
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (chk == 12345ull) a.chk[blockIdx.x] = chk;
  if (tid == 0) a.cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------------------------
static double mean_cyc(unsigned long long* d_cyc, int nwg) {
  std::vector<unsigned long long> c(nwg);
  CK(hipMemcpy(c.data(), d_cyc, nwg * 8, hipMemcpyDeviceToHost));
  double s = 0;
  for (auto x : c) s += (double)x;
  return s / nwg;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int cols_per_cu = argc > 1 ? atoi(argv[1]) : 16;
  const int epi = argc > 2 ? atoi(argv[2]) : 2;
  const int nwg = 256, nchunk = 3;
  const int items = cols_per_cu * nchunk;
  const int segs_per_chunk = argc > 4 ? atoi(argv[4]) : 22200;   // 20,000 genes x ~1.11 segments: 17 MB of lists in all (argv[4]: fewer = L2-resident lists)
  const int nseg = segs_per_chunk * nchunk;
  std::mt19937_64 rng(20250614);
  // id segments: 128 ids, every 16 lanes of either instruction on 16 different 8-byte banks (what the plan reaches where
  // the counts allow); the last ~10 % of the lanes of the second instruction point at trash accumulators
  std::vector<uint16_t> ids((size_t)(nseg + 1) * 128, 0);
  const int trash_mode = argc > 6 ? atoi(argv[6]) : 0;   // 1: as the product's plan -- 16 trash accumulators, and 11 % of the segments hold 2 ids + padding (a gene's SECOND segment)
  for (int s = 0; s < nseg; ++s) {
    const int fill = (trash_mode == 1 && rng() % 100 < 11) ? 2 : 100 + (int)(rng() % 29);   // ids in use
    for (int half = 0; half < 2; ++half)
      for (int g16 = 0; g16 < 4; ++g16) {
        int perm[16];
        for (int r = 0; r < 16; ++r) perm[r] = r;
        for (int r = 15; r > 0; --r) { const int q = (int)(rng() % (r + 1)); std::swap(perm[r], perm[q]); }
        for (int l = 0; l < 16; ++l) {
          const int lane = g16 * 16 + l;
          const int pos = half * 64 + lane;
          uint16_t id;
          if (pos < fill) id = (uint16_t)(16 * (rng() % (kCh / 16)) + perm[l]);
          else if (trash_mode == 1) id = (uint16_t)(kCh + 5 + (l & 15));   // (kCh + 5 is a multiple of 16: 16 trash slots on 16 banks)
          else id = (uint16_t)(kCh + ((kCh + kTrash - kCh) > 16 ? (16 * (rng() % 3) + perm[l]) : perm[l]));
          ids[(size_t)s * 128 + lane * 2 + half] = id;   // dword of lane = {lo: instruction 0, hi: instruction 1}
        }
      }
  }
  for (int q = 0; q < 128; ++q) ids[(size_t)nseg * 128 + q] = (uint16_t)(kCh + ((q >> 1) & 15));   // the all-padding segment
  // (kCh is not a multiple of 16: trash ids start at kCh, banks differ per lane all the same)
  // the same ids per segment in the quad layout: (instruction half h, 16-lane group k, lane r) -> position 8 r + 2 k + h
  std::vector<uint16_t> ids4v(ids.size());
  for (int sg = 0; sg <= nseg; ++sg)
    for (int k = 0; k < 4; ++k)
      for (int r = 0; r < 16; ++r)
        for (int h = 0; h < 2; ++h)
          ids4v[(size_t)sg * 128 + 8 * r + 2 * k + h] = ids[(size_t)sg * 128 + (16 * k + r) * 2 + h];
  std::vector<int32_t> work((size_t)nwg * items * kListCap, 0), cnt((size_t)nwg * items);
  double total_entries = 0;
  for (int w = 0; w < nwg; ++w)
    for (int it = 0; it < items; ++it) {
      const int chunk = it / cols_per_cu;   // chunk-major: every workgroup is in the same chunk at the same time
      const int n = (argc > 3 && atoi(argv[3]) == 1 ? 1075 : 965) + (int)(rng() % 71);   // 1,000 +- 35 segments (argv[3] = 1: 1,110 +- 35, two rounds in `cur`)
      cnt[(size_t)w * items + it] = n;
      total_entries += n;
      for (int e = 0; e < n; ++e) work[((size_t)w * items + it) * kListCap + e] = chunk * segs_per_chunk + (int)(rng() % segs_per_chunk);
    }
  uint16_t* d_ids; int32_t *d_work, *d_cnt; unsigned long long *d_S, *d_S2, *d_cyc, *d_chk;
  CK(hipMalloc(&d_ids, ids.size() * 2));
  CK(hipMemcpy(d_ids, ids.data(), ids.size() * 2, hipMemcpyHostToDevice));
  uint16_t* d_ids4;
  CK(hipMalloc(&d_ids4, ids4v.size() * 2));
  CK(hipMemcpy(d_ids4, ids4v.data(), ids4v.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_work, work.size() * 4));
  CK(hipMemcpy(d_work, work.data(), work.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_cnt, cnt.size() * 4));
  CK(hipMemcpy(d_cnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice));
  const size_t Swords = (size_t)nwg * items * kCh;
  CK(hipMalloc(&d_S, Swords * 8));
  CK(hipMalloc(&d_S2, Swords * 8));
  CK(hipMalloc(&d_cyc, nwg * 8));
  CK(hipMalloc(&d_chk, nwg * 8));
  const int extra = argc > 5 ? atoi(argv[5]) : 0;
  std::vector<double> kwv((size_t)3 * kCh * 2);
  for (size_t i = 0; i < kwv.size(); i += 2) { kwv[i] = (extra & 8) ? 0.25 : 0.5; kwv[i + 1] = (extra & 8) ? 0.25 : 0.0; }
  std::vector<unsigned short> szv((size_t)3 * kCh, 1);
  unsigned short* d_sz;
  CK(hipMalloc(&d_sz, szv.size() * 2));
  CK(hipMemcpy(d_sz, szv.data(), szv.size() * 2, hipMemcpyHostToDevice));
  std::vector<int32_t> segtabv((size_t)nseg + 1);
  for (int i = 0; i <= nseg; ++i) segtabv[i] = i;
  double* d_kw; int32_t* d_segtab;
  CK(hipMalloc(&d_kw, kwv.size() * 8));
  CK(hipMemcpy(d_kw, kwv.data(), kwv.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_segtab, segtabv.size() * 4));
  CK(hipMemcpy(d_segtab, segtabv.data(), segtabv.size() * 4, hipMemcpyHostToDevice));
  Args a{};
  a.sz = d_sz;
  a.extra = extra; a.kw = d_kw; a.segtab = d_segtab; a.cols_per_cu = cols_per_cu;
  a.ids = d_ids; a.ids_bytes = (uint32_t)(((size_t)nseg + 1) * 256); a.work = d_work; a.cnt = d_cnt; a.items = items;
  a.epi = epi; a.S = d_S; a.S_per_item = 1; a.cyc = d_cyc; a.chk = d_chk;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_cur<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_cur<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_quad<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_quad<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_ring<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_ring<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_ring<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_ring<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&atomics_only), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&epilogue_only<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&epilogue_only<768>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const size_t smem_cur = (size_t)(kCh + kTrash) * 8, smem_ring = (size_t)kAccOff + (size_t)(kCh + kTrash) * 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double pairs_per_item = total_entries / ((double)nwg * items);
  printf("work: %d workgroups x %d items (%d columns x 3 chunks), %.0f segments per item, %d segments of 256 B (%.1f MB), epilogue mode %d, extra traffic %d\n",
         nwg, items, cols_per_cu, pairs_per_item, nseg, nseg * 256.0 / 1e6, epi, extra);
  auto report = [&](const char* name, float ms) {
    const double cyc = mean_cyc(d_cyc, nwg);
    std::vector<unsigned long long> ck(nwg);
    CK(hipMemcpy(ck.data(), d_chk, nwg * 8, hipMemcpyDeviceToHost));
    unsigned long long x = 0;
    for (auto v : ck) x ^= v * 0x9e3779b97f4a7c15ull + (x << 7);
    printf("%-34s %8.3f ms  %9.0f cycles per item  %6.2f cycles per segment per CU  (clock %.2f GHz)  -> C3 (100k cells) %.1f ms   chk %016llx\n",
           name, ms, cyc / items, cyc / items / pairs_per_item, cyc / (ms * 1e-3) / 1e9, ms * (100000.0 / (nwg * cols_per_cu)), x);
  };
  std::vector<unsigned long long> S_ref;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(d_chk, 0, nwg * 8));
    a.S = d_S;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(walk_cur<false>, dim3(nwg), dim3(1024), smem_cur, 0, a, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) report("cur (16 wavefronts load + add)", ms);
  }
  {   // the same with stamps: where a wavefront's item goes (mean over workgroups, per wavefront)
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, (size_t)nwg * 16 * 4 * 8));
    CK(hipMemset(d_chk, 0, nwg * 8));
    hipLaunchKernelGGL(walk_cur<true>, dim3(nwg), dim3(1024), smem_cur, 0, a, d_st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nwg * 16 * 4);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    printf("    cur, stamps per item (cycles): wavefront: first ids in hand | walk | barrier 1 | epilogue + barrier 2\n");
    for (int w = 0; w < 16; ++w) {
      double ph[4] = {0, 0, 0, 0};
      for (int b = 0; b < nwg; ++b) for (int k = 0; k < 4; ++k) ph[k] += (double)st[((size_t)b * 16 + w) * 4 + k];
      printf("      w%-2d %7.0f %7.0f %7.0f %7.0f\n", w, ph[0] / nwg / items, ph[1] / nwg / items, ph[2] / nwg / items, ph[3] / nwg / items);
    }
    CK(hipFree(d_st));
  }
  if (epi == 2) { S_ref.resize(Swords); CK(hipMemcpy(S_ref.data(), d_S, Swords * 8, hipMemcpyDeviceToHost)); }
  {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(d_chk, 0, nwg * 8));
      a.S = d_S2;
      CK(hipMemset(d_S2, 0xff, Swords * 8));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(walk_quad<false>, dim3(nwg), dim3(1024), smem_cur, 0, a, d_ids4, (unsigned long long*)nullptr);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) report("quad (4 segments per dwordx4)", ms);
    }
    if (epi == 2) {
      std::vector<unsigned long long> S2(Swords);
      CK(hipMemcpy(S2.data(), d_S2, Swords * 8, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < Swords; ++i) bad += S2[i] != S_ref[i];
      printf("    scores vs cur: %zu of %zu differ\n", bad, Swords);
    }
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, (size_t)nwg * 16 * 4 * 8));
    CK(hipMemset(d_chk, 0, nwg * 8));
    hipLaunchKernelGGL(walk_quad<true>, dim3(nwg), dim3(1024), smem_cur, 0, a, d_ids4, d_st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nwg * 16 * 4);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    printf("    quad, stamps per item (cycles): wavefront: first ids in hand | walk | barrier 1 | epilogue + barrier 2\n");
    for (int w = 0; w < 16; w += 3) {
      double ph[4] = {0, 0, 0, 0};
      for (int b = 0; b < nwg; ++b) for (int k = 0; k < 4; ++k) ph[k] += (double)st[((size_t)b * 16 + w) * 4 + k];
      printf("      w%-2d %7.0f %7.0f %7.0f %7.0f\n", w, ph[0] / nwg / items, ph[1] / nwg / items, ph[2] / nwg / items, ph[3] / nwg / items);
    }
    CK(hipFree(d_st));
  }
  {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_steal<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk_steal<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t smem_steal = (size_t)kGrabOff + 16;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(d_chk, 0, nwg * 8));
      a.S = d_S2;
      CK(hipMemset(d_S2, 0xff, Swords * 8));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(walk_steal<false>, dim3(nwg), dim3(1024), smem_steal, 0, a, (unsigned long long*)nullptr);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) report("steal (groups of 16 taken from LDS)", ms);
    }
    if (epi == 2) {
      std::vector<unsigned long long> S2(Swords);
      CK(hipMemcpy(S2.data(), d_S2, Swords * 8, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < Swords; ++i) bad += S2[i] != S_ref[i];
      printf("    scores vs cur: %zu of %zu differ\n", bad, Swords);
    }
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, (size_t)nwg * 16 * 4 * 8));
    CK(hipMemset(d_chk, 0, nwg * 8));
    hipLaunchKernelGGL(walk_steal<true>, dim3(nwg), dim3(1024), smem_steal, 0, a, d_st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nwg * 16 * 4);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    printf("    steal, stamps per item (cycles): wavefront: publish + barrier | walk | barrier 1 | epilogue + barrier 2\n");
    for (int w = 0; w < 16; w += 3) {
      double ph[4] = {0, 0, 0, 0};
      for (int b = 0; b < nwg; ++b) for (int k = 0; k < 4; ++k) ph[k] += (double)st[((size_t)b * 16 + w) * 4 + k];
      printf("      w%-2d %7.0f %7.0f %7.0f %7.0f\n", w, ph[0] / nwg / items, ph[1] / nwg / items, ph[2] / nwg / items, ph[3] / nwg / items);
    }
    CK(hipFree(d_st));
  }
  auto run_ring = [&](auto kern, const char* name) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(d_chk, 0, nwg * 8));
      a.S = d_S2;
      CK(hipMemset(d_S2, 0xff, Swords * 8));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(nwg), dim3(1024), smem_ring, 0, a);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) report(name, ms);
    }
    if (epi == 2) {
      std::vector<unsigned long long> S2(Swords);
      CK(hipMemcpy(S2.data(), d_S2, Swords * 8, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < Swords; ++i) bad += S2[i] != S_ref[i];
      printf("    scores vs cur: %zu of %zu differ\n", bad, Swords);
    }
  };
  run_ring(walk_ring<2>, "ring (4 prod + 12 cons), 2 in flight");
  run_ring(walk_ring<4>, "ring (4 prod + 12 cons), 4 in flight");
  {
    const int ppw = ((int)(pairs_per_item * items / 16) / 32) * 32;
    hipLaunchKernelGGL(atomics_only, dim3(nwg), dim3(1024), smem_cur, 0, a, ppw);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(atomics_only, dim3(nwg), dim3(1024), smem_cur, 0, a, ppw);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = mean_cyc(d_cyc, nwg);
    printf("%-34s %8.3f ms  %9.0f cycles per item  %6.2f cycles per segment per CU\n", "floor: atomics alone (ids in regs)", ms,
           cyc / items, cyc / (16.0 * ppw));
  }
  for (int nt = 0; nt < 2; ++nt) {
    a.S = d_S;
    CK(hipEventRecord(e0));
    if (nt == 0) hipLaunchKernelGGL(epilogue_only<1024>, dim3(nwg), dim3(1024), smem_cur, 0, a);
    else hipLaunchKernelGGL(epilogue_only<768>, dim3(nwg), dim3(1024), smem_cur, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = mean_cyc(d_cyc, nwg);
    printf("%-34s %8.3f ms  %9.0f cycles per item  (%.2f TB/s of score stores)\n",
           nt == 0 ? "floor: epilogue alone, 1024 thr" : "floor: epilogue alone, 768 thr", ms, cyc / items,
           epi == 2 ? (double)nwg * items * kCh * 8 / (ms * 1e-3) / 1e12 : 0.0);
  }
  return 0;
}
