// Host-side preparation of the gene-set membership matrix G for the device kernels.
// Input is what gmt2mat() (R/gmt-utils.R:19-66) produces after plaid()'s alignment and
// binarisation (R/plaid.R:65-73): a 0/1 CSC pattern, genes x sets, rows indexed in X's
// row space.  colSums(G) (R/plaid.R:75) are the column lengths.
#include <algorithm>
#include <cstring>
#include <numeric>

#include "common.h"

using namespace plaidhip;

namespace {

struct HostTiles {
  std::vector<uint16_t> idx;        // [chunk][lane][8]
  std::vector<int32_t> chunk_off;   // tiles + 1
  std::vector<int32_t> lane_set;    // tiles * 64
};

// Tiles of 64 sets in order of decreasing size (stable), every lane's list padded to the
// tile's longest list rounded up to 8 steps.  Padded slots read the zero entries that
// follow the column in LDS (g .. g+kPadSlots-1).
void build_tiles(int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, HostTiles& t) {
  std::vector<int32_t> order(m);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    return (Gp[a + 1] - Gp[a]) > (Gp[b + 1] - Gp[b]);
  });
  const int32_t tiles = (m + 63) / 64;
  t.chunk_off.assign(tiles + 1, 0);
  t.lane_set.assign((size_t)tiles * 64, -1);
  for (int32_t ti = 0; ti < tiles; ++ti) {
    int32_t longest = 0;
    for (int l = 0; l < 64; ++l) {
      const int32_t s = ti * 64 + l;
      if (s >= m) break;
      const int32_t j = order[s];
      t.lane_set[(size_t)ti * 64 + l] = j;
      longest = std::max(longest, Gp[j + 1] - Gp[j]);
    }
    t.chunk_off[ti + 1] = t.chunk_off[ti] + (longest + 7) / 8;
  }
  const size_t chunks = (size_t)t.chunk_off[tiles];
  t.idx.assign(chunks * 64 * 8, 0);
  for (int32_t ti = 0; ti < tiles; ++ti) {
    const int32_t c0 = t.chunk_off[ti], c1 = t.chunk_off[ti + 1];
    for (int l = 0; l < 64; ++l) {
      const int32_t j = t.lane_set[(size_t)ti * 64 + l];
      const int32_t p0 = j >= 0 ? Gp[j] : 0;
      const int32_t len = j >= 0 ? Gp[j + 1] - Gp[j] : 0;
      const uint16_t pad = (uint16_t)(g + (l & (kPadSlots - 1)));
      for (int32_t c = c0; c < c1; ++c) {
        uint16_t* dst = &t.idx[((size_t)c * 64 + l) * 8];
        for (int e = 0; e < 8; ++e) {
          const int32_t step = (c - c0) * 8 + e;
          dst[e] = step < len ? (uint16_t)Gi[p0 + step] : pad;
        }
      }
    }
  }
}

template <typename T>
int upload(plaidhip_ctx* ctx, const std::vector<T>& h, T** d) {
  *d = nullptr;
  const size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
  if (hipMalloc(reinterpret_cast<void**>(d), bytes) != hipSuccess) {
    set_error("hipMalloc(%zu) failed", bytes);
    return PLAIDHIP_ENOMEM;
  }
  if (!h.empty())
    PH_HIP(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return PLAIDHIP_OK;
}

}  // namespace

extern "C" int plaidhip_geneset_create(plaidhip_ctx* ctx, int32_t g, int32_t m, const int32_t* Gp,
                                       const int32_t* Gi, plaidhip_geneset** out) {
  PH_REQUIRE(ctx && out, "geneset_create: null ctx/out");
  PH_REQUIRE(g > 0 && m >= 0, "geneset_create: bad dims g=%d m=%d", g, m);
  PH_REQUIRE(Gp != nullptr, "geneset_create: Gp is null");
  PH_REQUIRE(Gp[0] == 0, "geneset_create: Gp[0] must be 0");
  for (int32_t j = 0; j < m; ++j)
    PH_REQUIRE(Gp[j + 1] >= Gp[j], "geneset_create: Gp not non-decreasing at set %d", j);
  const int64_t z = m > 0 ? Gp[m] : 0;
  PH_REQUIRE(z == 0 || Gi != nullptr, "geneset_create: Gi is null");
  for (int64_t p = 0; p < z; ++p)
    PH_REQUIRE(Gi[p] >= 0 && Gi[p] < g, "geneset_create: row index %d out of range [0,%d)", Gi[p], g);
  PH_HIP(hipSetDevice(ctx->device));

  auto* gs = new (std::nothrow) plaidhip_geneset();
  if (!gs) { set_error("out of host memory"); return PLAIDHIP_ENOMEM; }
  gs->ctx = ctx;
  gs->g = g;
  gs->m = m;
  gs->z = z;
  gs->lds_ok = g <= kMaxLdsGenes;
  gs->tiles = (m + 63) / 64;

  int rc = PLAIDHIP_OK;
  std::vector<int32_t> sizes(m);
  for (int32_t j = 0; j < m; ++j) sizes[j] = Gp[j + 1] - Gp[j];
  std::vector<int32_t> hGp(Gp, Gp + m + 1), hGi(Gi, Gi + z);
  if ((rc = upload(ctx, sizes, &gs->d_set_size)) != PLAIDHIP_OK) goto fail;
  if ((rc = upload(ctx, hGp, &gs->d_Gp)) != PLAIDHIP_OK) goto fail;
  if ((rc = upload(ctx, hGi, &gs->d_Gi)) != PLAIDHIP_OK) goto fail;
  if (gs->lds_ok) {
    HostTiles t;
    build_tiles(g, m, Gp, Gi, t);
    gs->chunks = t.chunk_off.back();
    if ((rc = upload(ctx, t.idx, &gs->d_tile_idx)) != PLAIDHIP_OK) goto fail;
    if ((rc = upload(ctx, t.chunk_off, &gs->d_tile_chunk_off)) != PLAIDHIP_OK) goto fail;
    if ((rc = upload(ctx, t.lane_set, &gs->d_lane_set)) != PLAIDHIP_OK) goto fail;
    // host vectors die at scope exit: make sure the copies have landed
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = PLAIDHIP_EHIP; goto fail; }
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = PLAIDHIP_EHIP; goto fail; }
  *out = gs;
  return PLAIDHIP_OK;
fail:
  plaidhip_geneset_destroy(gs);
  return rc;
}

extern "C" int plaidhip_geneset_destroy(plaidhip_geneset* gs) {
  if (!gs) return PLAIDHIP_OK;
  hipFree(gs->d_tile_idx);
  hipFree(gs->d_tile_chunk_off);
  hipFree(gs->d_lane_set);
  hipFree(gs->d_set_size);
  hipFree(gs->d_Gp);
  hipFree(gs->d_Gi);
  delete gs;
  return PLAIDHIP_OK;
}

extern "C" int plaidhip_geneset_info(const plaidhip_geneset* gs, int64_t info[8]) {
  PH_REQUIRE(gs && info, "geneset_info: null argument");
  std::memset(info, 0, 8 * sizeof(int64_t));
  info[0] = gs->g;
  info[1] = gs->m;
  info[2] = gs->z;
  info[3] = gs->chunks * 64 * 8;
  info[4] = gs->tiles;
  info[5] = gs->lds_ok ? 1 : 0;
  return PLAIDHIP_OK;
}
