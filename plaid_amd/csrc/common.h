// Internal declarations shared by the C-ABI translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/plaidhip.h"

namespace plaidhip {

// LDS budget of one CU on gfx950: 160 KiB, one workgroup may own all of it.
constexpr int kLdsBytes = 160 * 1024;
// The LDS-resident column kernels keep one 8-byte entry per gene plus kPadSlots zero
// entries that padded index slots point at.
constexpr int kPadSlots = 32;
constexpr int kMaxLdsGenes = kLdsBytes / 8 - kPadSlots;  // 20448
// bitonic sort in LDS: 8-byte keys
constexpr int kMaxLdsKeys = kLdsBytes / 8;               // 20480

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define PH_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) return ::plaidhip::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

#define PH_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      ::plaidhip::set_error(__VA_ARGS__);     \
      return PLAIDHIP_EINVAL;                 \
    }                                         \
  } while (0)

}  // namespace plaidhip

struct plaidhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // growable scratch
  void* ws = nullptr;
  size_t ws_bytes = 0;
  int num_cu = 256;
};

// Prepared membership.  Sets are processed in "tiles" of 64 (one per wavefront lane),
// taken in order of decreasing size so a tile's lanes have similar list lengths.
//   tile_idx : u16 gene ids, layout [tile chunk][lane 0..63][8]  (a chunk = 8 steps;
//              one 16-byte load per lane per chunk, 1 KiB per wave, coalesced).  Padded
//              slots hold g + (a pad slot id) and read a zero entry of the LDS column.
//   tile_chunk_off[t] : first chunk of tile t (tiles+1 entries)
//   lane_set[t*64+l]  : original set id handled by lane l of tile t, or -1
//   set_size[j]       : k_j
struct plaidhip_geneset {
  plaidhip_ctx* ctx = nullptr;
  int32_t g = 0, m = 0;
  int64_t z = 0;
  int32_t tiles = 0;
  int64_t chunks = 0;          // total 8-step chunks over all tiles
  // device
  uint16_t* d_tile_idx = nullptr;
  int32_t* d_tile_chunk_off = nullptr;
  int32_t* d_lane_set = nullptr;
  int32_t* d_set_size = nullptr;
  int32_t* d_Gp = nullptr;     // plain CSC copy (fallback kernel, large-g path)
  int32_t* d_Gi = nullptr;
  bool lds_ok = false;         // g <= kMaxLdsGenes
};

namespace plaidhip {

int ensure_workspace(plaidhip_ctx* ctx, size_t bytes);

// kernels_spmm.hip
int launch_spmm_dense_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const double* X,
                          int64_t ldx, int32_t n, int stat, double alpha, const double* alpha_div,
                          double beta, double* S, int64_t lds, uint32_t* flags);
int launch_spmm_csc_f64(plaidhip_ctx* ctx, const plaidhip_geneset* gs, const int32_t* Xp,
                        const int32_t* Xi, const double* Xx, int32_t n, int stat, double alpha,
                        const double* alpha_div, double beta, double* S, int64_t lds, uint32_t* flags);
// kernels_rank.hip
int launch_colranks_dense_f64(plaidhip_ctx* ctx, const double* X, int64_t ldx, int32_t g, int32_t n,
                              int ties, int is_signed, double power, double* R, int64_t ldr,
                              double* colmax);
int launch_colranks_csc_f64(plaidhip_ctx* ctx, const int32_t* Xp, const double* Xx, int32_t n,
                            int ties, int is_signed, double power, double* Rx, double* colmax);
// kernels_norm.hip
int launch_minflags(plaidhip_ctx* ctx, const double* S, int64_t count, uint32_t* flags);
int launch_col_medians(plaidhip_ctx* ctx, const double* S, int64_t lds, int32_t m, int32_t n,
                       int ignore_zero, const uint32_t* flags, double* med);
int launch_sum(plaidhip_ctx* ctx, const double* v, int64_t count, double* out);
int launch_max(plaidhip_ctx* ctx, const double* v, int64_t count, double* out);
int launch_shift_columns(plaidhip_ctx* ctx, double* S, int64_t lds, int32_t m, int32_t n,
                         const double* med, double add, const double* red);

}  // namespace plaidhip
