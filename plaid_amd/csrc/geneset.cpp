// Host-side preparation of the gene-set membership matrix G for the device kernels.
// Input is what gmt2mat() (R/gmt-utils.R:19-66) produces after plaid()'s alignment and
// binarisation (R/plaid.R:65-73): a 0/1 CSC pattern, genes x sets, rows indexed in X's
// row space.  colSums(G) (R/plaid.R:75) are the column lengths.
//
// What is built (once per G, reused for every sample column, chunk and GPU):
//   1. sets are sorted by decreasing size (stable) and cut into tiles of 64 -- one set per
//      wavefront lane, so the lanes of a tile have similar list lengths;
//   2. inside each half-tile (32 lanes = one LDS lane group of ds_read_b64) the order in which
//      every lane visits its genes is chosen by EDGE-COLOURING the bipartite multigraph
//      lanes x LDS bank-slots (slot = gene mod 32): at every step the 32 lanes read 32
//      different bank-slots, so the gather is LDS-bank-conflict free by construction.  By
//      Koenig's theorem max(longest list, busiest slot) steps suffice; idle (lane, step) pairs
//      read one of 32 zero entries behind the column, again on an unused slot;
//   3. tiles are dealt to the W wavefronts of the workgroup by longest-processing-time, and
//      every wavefront's tiles are laid out back to back as one contiguous stream of 16-byte
//      chunks (8 u16 gene ids per lane), so the kernel's index prefetch never restarts.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <numeric>
#include <thread>

#include "common.h"

using namespace plaidhip;

// measured on MI355X (tools/bench_spmm.py --ablate 4), see wave_weights()
// (the pair plan interpolates its shares by tile count, build_pair_plan)
static const double kAgeShare16Single[4] = {1.6, 1.2, 0.8, 0.4};    // one-column plan: tuned on its main user, the mixed-precision pair kernel

namespace {

struct Edge {
  uint8_t u, v;      // lane (0..31), slot vertex (0..31, or 0..63 with two read ports per slot)
  uint16_t gene;
  int32_t color;
};

// Proper edge colouring of a bipartite multigraph with D = max degree colours
// (alternating-path / Koenig construction).  32 + 32 vertices.
void color_bipartite(std::vector<Edge>& E, int D, int nV = 32) {
  if (E.empty()) return;
  std::vector<int32_t> atU((size_t)32 * D, -1), atV((size_t)nV * D, -1);
  std::vector<int32_t> path;
  // smallest colour that may still be free at a vertex: colours fill from the low end, so the scan for the first free one
  // starts here instead of at 0 (a tile with an all-genes set has D = 12,010 colours and 1.4e5 edges: the full scans
  // were 1.4 of the 1.9 s that preparing the reference-shaped collection took)
  int32_t loU[32], loV[64];
  for (int k = 0; k < 32; ++k) loU[k] = 0;
  for (int k = 0; k < 64; ++k) loV[k] = 0;
  auto first_free = [&](const std::vector<int32_t>& at, int32_t* lo, int x) {
    const int32_t* row = &at[(size_t)x * D];
    int c = lo[x];
    while (c < D && row[c] >= 0) ++c;
    lo[x] = c;
    return c < D ? c : -1;
  };
  for (int32_t e = 0; e < (int32_t)E.size(); ++e) {
    const int u = E[e].u, v = E[e].v;
    const int a = first_free(atU, loU, u);
    int c = -1;
    if (atV[(size_t)v * D + a] < 0) {
      c = a;
    } else {
      const int b = first_free(atV, loV, v);
      if (atU[(size_t)u * D + b] < 0) {
        c = b;
      } else {
        // a is free at u only, b is free at v only: swap a<->b on the alternating path from v
        path.clear();
        int x = v, col = a;
        bool at_v = true;
        for (;;) {
          const int32_t pe = at_v ? atV[(size_t)x * D + col] : atU[(size_t)x * D + col];
          if (pe < 0) break;
          path.push_back(pe);
          x = at_v ? E[pe].u : E[pe].v;
          at_v = !at_v;
          col = (col == a) ? b : a;
        }
        for (int32_t pe : path) {
          atU[(size_t)E[pe].u * D + E[pe].color] = -1;
          atV[(size_t)E[pe].v * D + E[pe].color] = -1;
          loU[E[pe].u] = std::min(loU[E[pe].u], E[pe].color);
          loV[E[pe].v] = std::min(loV[E[pe].v], E[pe].color);
        }
        for (int32_t pe : path) {
          const int nc = (E[pe].color == a) ? b : a;
          E[pe].color = nc;
          atU[(size_t)E[pe].u * D + nc] = pe;
          atV[(size_t)E[pe].v * D + nc] = pe;
        }
        c = a;  // now free at both ends
      }
    }
    E[e].color = c;
    atU[(size_t)u * D + c] = e;
    atV[(size_t)v * D + c] = e;
  }
}

struct TilePlan {
  int32_t steps = 0;                 // multiple of 8
  std::vector<uint16_t> idx;         // [step][lane 0..63]
};

// Relative cost of one extra LDS pass (a 2-way conflict inside one lane group) against a whole
// extra gather step; PLAIDHIP_CONFLICT_COST overrides it in the tools/ build (>= 1 disables conflicted steps).
double conflict_cost() {
#ifdef PLAIDHIP_DIAG
  static const double c = [] {
    const char* e = getenv("PLAIDHIP_CONFLICT_COST");
    return e ? atof(e) : 0.2;
  }();
  return c;
#else
  return 0.2;
#endif
}

// Number of steps T (a multiple of 8) for a tile whose longest lane has Lmax reads and whose slot
// degrees are deg[0..n).  Koenig needs max(longest lane, busiest slot) steps for a conflict-free
// schedule, and the busiest slot sits ~2 sigma above the mean.  A slot may instead serve TWO lanes
// in a few steps (its reads beyond the first T go to a second vertex of that slot): such a step
// costs one more LDS pass for that lane group but no instruction, no index bytes and no
// VGPR-return cycles.  T minimises  T + conflict_cost * (reads beyond T over all slots).
int choose_steps(int Lmax, const int* deg, int n) {
  int Dmax = 0;
  for (int k = 0; k < n; ++k) Dmax = std::max(Dmax, deg[k]);
  const double cx = conflict_cost();
  const int t_lo = std::max(8, (std::max(Lmax, (Dmax + 1) / 2) + 7) & ~7);
  const int t_hi = std::max(t_lo, (Dmax + 7) & ~7);
  if (cx >= 1.0) return t_hi;
  int T = t_hi;
  double best = 1e300;
  for (int t = t_lo; t <= t_hi; t += 8) {
    int64_t excess = 0;
    for (int k = 0; k < n; ++k) excess += std::max(0, deg[k] - t);
    const double cost = (double)t + cx * (double)excess;
    if (cost < best) { best = cost; T = t; }
  }
  return T;
}

// Schedule one tile (64 lanes; lane l handles set lane_set[l] or nothing): edge-colour each
// 32-lane half (the unit ds_read_b64 is served in) against the 32 bank pairs, slot = gene mod 32.
void plan_tile(int32_t g, const int32_t* Gp, const int32_t* Gi, const int32_t* lane_set, TilePlan& tp) {
  std::vector<Edge> half[2];
  int degV[64] = {0};
  int Lmax = 0;
  for (int h = 0; h < 2; ++h)
    for (int l = 0; l < 32; ++l) {
      const int32_t j = lane_set[h * 32 + l];
      if (j < 0) continue;
      for (int32_t p = Gp[j]; p < Gp[j + 1]; ++p) {
        const int32_t gene = Gi[p];
        Edge e{(uint8_t)l, (uint8_t)(gene & 31), (uint16_t)gene, -1};
        half[h].push_back(e);
        ++degV[h * 32 + (gene & 31)];
      }
      Lmax = std::max(Lmax, Gp[j + 1] - Gp[j]);
    }
  const int T = choose_steps(Lmax, degV, 64);
  for (int h = 0; h < 2; ++h) {
    int seen[32] = {0};
    for (Edge& e : half[h])
      if (seen[e.v]++ >= T) e.v = (uint8_t)(e.v + 32);   // second vertex of an over-full slot
    color_bipartite(half[h], T, 64);
  }
  tp.steps = T;                                            // >= 1 chunk: empty sets still get their 0 written
  tp.idx.assign((size_t)tp.steps * 64, 0);
  std::vector<uint8_t> used((size_t)tp.steps * 64, 0);     // (step, half*32 + slot) taken by a real read
  std::vector<uint8_t> filled((size_t)tp.steps * 64, 0);   // (step, lane) has a real read
  for (int h = 0; h < 2; ++h)
    for (const Edge& e : half[h]) {
      const size_t s = (size_t)e.color;
      tp.idx[s * 64 + h * 32 + e.u] = e.gene;
      filled[s * 64 + h * 32 + e.u] = 1;
      used[s * 64 + h * 32 + (e.v & 31)] = 1;
    }
  // idle (lane, step) pairs read a zero entry behind the column (index g + r, r < 32) whose
  // bank-slot ((g + r) mod 32) nobody else uses in this step: #free slots >= #idle lanes.
  for (int32_t s = 0; s < tp.steps; ++s)
    for (int h = 0; h < 2; ++h) {
      int slot = 0;
      for (int l = 0; l < 32; ++l) {
        if (filled[(size_t)s * 64 + h * 32 + l]) continue;
        while (used[(size_t)s * 64 + h * 32 + slot]) ++slot;
        const int r = ((slot - g) % 32 + 32) % 32;
        tp.idx[(size_t)s * 64 + h * 32 + l] = (uint16_t)(g + r);
        ++slot;
      }
    }
}

struct HostPlan {
  int32_t waves = 0;
  std::vector<uint16_t> idx;          // [chunk][lane][8]
  std::vector<int32_t> wave_chunk_off;  // waves + 1
  std::vector<int32_t> wave_tile_off;   // waves + 1  (into wtile_*)
  std::vector<int32_t> wtile_end;       // absolute chunk index one past the tile's last chunk
  std::vector<int32_t> wtile_id;        // tile ordinal -> lane_set[tile*64 ..]
  std::vector<int32_t> lane_set;        // tiles * 64
  std::vector<int32_t> meta_j;          // [wave-stream tile k][lane]
  std::vector<double> meta_w, meta_k;
  int64_t chunks = 0;
};


// Share of the work a wavefront gets, by its age rank on its SIMD (wave w of a workgroup runs on
// SIMD w % 4; w / 4 is its age there).  The SIMD arbiter serves the oldest ready wave first, so with
// equal shares the older waves finish early and the youngest one ends up alone on the SIMD, which a
// single wave cannot keep busy.  Shares proportional to the speed each age actually gets let all
// waves reach the end-of-column barrier together.  PLAIDHIP_WAVE_WEIGHTS="a,b,c,d" overrides (tools/ build).
void wave_weights(int waves, std::vector<double>& wt, const double* share16) {
  double age[4] = {1.0, 1.0, 1.0, 1.0};
  if (waves == 16) { age[0] = share16[0]; age[1] = share16[1]; age[2] = share16[2]; age[3] = share16[3]; }
#ifdef PLAIDHIP_DIAG
  if (const char* e = getenv("PLAIDHIP_WAVE_WEIGHTS")) {
    double a, b, c, d;
    if (sscanf(e, "%lf,%lf,%lf,%lf", &a, &b, &c, &d) == 4 && a > 0 && b > 0 && c > 0 && d > 0) {
      age[0] = a; age[1] = b; age[2] = c; age[3] = d;
    }
  }
#endif
  wt.resize(waves);
  for (int w = 0; w < waves; ++w) wt[w] = age[std::min(3, w / 4)];
}

// longest-processing-time assignment of tiles (cost[t]) to wavefronts with capacity weights
void assign_tiles(const std::vector<int64_t>& cost, int waves, std::vector<std::vector<int32_t>>& mine,
                  const double* share16) {
  const int32_t tiles = (int32_t)cost.size();
  std::vector<double> wt;
  wave_weights(waves, wt, share16);
  std::vector<int32_t> by_len(tiles);
  std::iota(by_len.begin(), by_len.end(), 0);
  std::stable_sort(by_len.begin(), by_len.end(), [&](int32_t a, int32_t b) { return cost[a] > cost[b]; });
  mine.assign(waves, {});
  std::vector<double> load(waves, 0.0);
  for (int32_t t : by_len) {
    int best = 0;
    double bv = 1e300;
    for (int w = 0; w < waves; ++w) {
      const double v = (load[w] + (double)cost[t]) / wt[w];
      if (v < bv) { bv = v; best = w; }
    }
    mine[best].push_back(t);
    load[best] += (double)cost[t];
  }
}

void build_plan(int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, const int32_t* total_size,
                int waves, HostPlan& hp) {
  std::vector<int32_t> order(m);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int32_t a, int32_t b) { return (Gp[a + 1] - Gp[a]) > (Gp[b + 1] - Gp[b]); });
  const int32_t tiles = (m + 63) / 64;
  hp.waves = waves;
  hp.lane_set.assign((size_t)tiles * 64, -1);
  for (int32_t s = 0; s < m; ++s) hp.lane_set[s] = order[s];

  std::vector<TilePlan> plans(tiles);
  {
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt, 16u));
    if (tiles < 8) nt = 1;
    std::vector<std::thread> pool;
    std::atomic<int> pool_failed{0};
    for (unsigned w = 0; w < nt; ++w)
      pool.emplace_back([&, w]() {
        try {
          for (int32_t t = (int32_t)w; t < tiles; t += (int32_t)nt)
            plan_tile(g, Gp, Gi, &hp.lane_set[(size_t)t * 64], plans[t]);
        } catch (...) {   // (out of memory inside a helper thread: reported by the caller's thread, below)
          pool_failed.store(1);
        }
      });
    for (auto& th : pool) th.join();
    if (pool_failed.load()) throw std::bad_alloc();
  }

  // longest-processing-time assignment of tiles to wavefronts
  std::vector<std::vector<int32_t>> mine;
  {
    std::vector<int64_t> cost(tiles);
    for (int32_t t = 0; t < tiles; ++t) cost[t] = plans[t].steps;
    assign_tiles(cost, waves, mine, kAgeShare16Single);
  }
  hp.wave_chunk_off.assign(waves + 1, 0);
  hp.wave_tile_off.assign(waves + 1, 0);
  int32_t chunk = 0;
  for (int w = 0; w < waves; ++w) {
    hp.wave_chunk_off[w] = chunk;
    hp.wave_tile_off[w] = (int32_t)hp.wtile_id.size();
    for (int32_t t : mine[w]) {
      chunk += plans[t].steps / 8;
      hp.wtile_end.push_back(chunk);
      hp.wtile_id.push_back(t);
    }
  }
  hp.wave_chunk_off[waves] = chunk;
  hp.wave_tile_off[waves] = (int32_t)hp.wtile_id.size();
  hp.chunks = chunk;
  // 8 spare chunks behind the stream: the kernel's 4-deep index prefetch may run past the end
  hp.idx.assign(((size_t)chunk + 12) * 64 * 8, (uint16_t)g);
  for (int w = 0; w < waves; ++w) {
    int32_t c = hp.wave_chunk_off[w];
    for (int32_t t : mine[w]) {
      const TilePlan& tp = plans[t];
      for (int32_t s = 0; s < tp.steps; ++s) {
        const int32_t ch = c + s / 8, e = s & 7;
        for (int l = 0; l < 64; ++l) hp.idx[(((size_t)ch * 64) + l) * 8 + e] = tp.idx[(size_t)s * 64 + l];
      }
      c += tp.steps / 8;
    }
  }
  // sentinel so that reading entry k one past a wave's last tile is harmless
  hp.wtile_end.push_back(-1);
  hp.wtile_id.push_back(0);
  const size_t nk = hp.wtile_id.size();
  hp.meta_j.assign(nk * 64, -1);
  hp.meta_w.assign(nk * 64, 0.0);
  hp.meta_k.assign(nk * 64, 0.0);
  for (size_t k = 0; k + 1 < nk; ++k)
    for (int l = 0; l < 64; ++l) {
      const int32_t j = hp.lane_set[(size_t)hp.wtile_id[k] * 64 + l];
      if (j < 0) continue;
      const double size = (double)total_size[j];   // the WHOLE set (all slices): colSums(G)
      hp.meta_j[k * 64 + l] = j;
      hp.meta_w[k * 64 + l] = 1.0 / (1e-8 + size);   // R/plaid.R:75-76
      hp.meta_k[k * 64 + l] = size;
    }
}

// ---- pair plan: two sample columns per 16-byte LDS entry (ds_read_b128) -------------------
// ds_read_b128 serves a wave in four groups of 16 lanes (MI355X LDS table); a group is conflict
// free when its 16 lanes read 16 different 16-byte slots, slot = gene mod 16.
static const uint8_t kB128Groups[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

// Schedule the genes of one tile that fall into [g0, g0+gs): edge-colour every 16-lane group
// against the 16 slots (see choose_steps for the number of steps).  Idle (lane, step) pairs read a
// zero entry gs + r (r < 16) on a free slot.
void plan_tile_b128(int32_t g0, int32_t gs, const int32_t* Gp, const int32_t* Gi, const int32_t* lane_set,
                    TilePlan& tp) {
  std::vector<Edge> grp[4];
  int degV[4][16] = {{0}};
  int Lmax = 0;
  for (int q = 0; q < 4; ++q) {
    for (int l = 0; l < 16; ++l) {
      const int32_t j = lane_set[kB128Groups[q][l]];
      if (j < 0) continue;
      int du = 0;
      for (int32_t p = Gp[j]; p < Gp[j + 1]; ++p) {
        const int32_t gene = Gi[p] - g0;
        if (gene < 0 || gene >= gs) continue;
        Edge e{(uint8_t)l, (uint8_t)(gene & 15), (uint16_t)gene, -1};
        grp[q].push_back(e);
        ++du;
        ++degV[q][gene & 15];
      }
      Lmax = std::max(Lmax, du);
    }
  }
  const int T = choose_steps(Lmax, &degV[0][0], 64);
  for (int q = 0; q < 4; ++q) {
    int seen[16] = {0};
    for (Edge& e : grp[q])
      if (seen[e.v]++ >= T) e.v = (uint8_t)(e.v + 16);   // second vertex of an over-full slot
    color_bipartite(grp[q], T, 32);
  }
  tp.steps = T;
  tp.idx.assign((size_t)tp.steps * 64, 0);
  std::vector<uint8_t> used((size_t)tp.steps * 64, 0), filled((size_t)tp.steps * 64, 0);
  for (int q = 0; q < 4; ++q)
    for (const Edge& e : grp[q]) {
      const size_t st = (size_t)e.color;
      tp.idx[st * 64 + kB128Groups[q][e.u]] = e.gene;
      filled[st * 64 + kB128Groups[q][e.u]] = 1;
      used[st * 64 + q * 16 + (e.v & 15)] = 1;
    }
  for (int32_t st = 0; st < tp.steps; ++st)
    for (int q = 0; q < 4; ++q) {
      int slot = 0;
      for (int l = 0; l < 16; ++l) {
        const int lane = kB128Groups[q][l];
        if (filled[(size_t)st * 64 + lane]) continue;
        while (used[(size_t)st * 64 + q * 16 + slot]) ++slot;   // #free slots >= #idle lanes
        const int r = ((slot - gs) % 16 + 16) % 16;
        tp.idx[(size_t)st * 64 + lane] = (uint16_t)(gs + r);
        ++slot;
      }
    }
}

struct PairSliceHost {
  int32_t g0 = 0, gs = 0;
  std::vector<uint16_t> idx;             // [chunk][lane][8]
  std::vector<int32_t> wave_chunk_off;   // waves + 1
  std::vector<int32_t> wtile_end;        // per wave-stream tile k (+ sentinel)
};
struct PairPlanHost {
  std::vector<PairSliceHost> slices;
  std::vector<int32_t> wave_tile_off;    // waves + 1, shared by all slices
  std::vector<int32_t> meta_j;           // [k][lane], shared
  std::vector<double> meta_w, meta_k;
  int64_t chunks = 0;
};

// Tiles (lane <-> set) and the tile -> wavefront assignment are the SAME in every gene slice, so a
// lane meets the same set again in the next slice and its partial sum can round-trip through S
// privately (no cross-wave hand-off).
void build_pair_plan(int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, int waves, PairPlanHost& pp) {
  const int32_t nsl = (g + kMaxLdsGenesPair - 1) / kMaxLdsGenesPair;
  int32_t width = (g + nsl - 1) / nsl;
  width = (width + 1) & ~1;
  std::vector<int32_t> starts;
  for (int32_t g0 = 0; g0 < g; g0 += width) starts.push_back(g0);
  const int S = (int)starts.size();
  // Tiles: 64 sets of similar total size.  (Clustering on the per-slice sizes, k-d fashion, saves
  // another 4 % of the steps but scatters the set ids of a tile, and the scattered 8-byte S stores
  // cost far more than that: measured 1.25 vs 1.05 ms on C2.)
  std::vector<int32_t> order(m);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int32_t a, int32_t b) { return (Gp[a + 1] - Gp[a]) > (Gp[b + 1] - Gp[b]); });
  const int32_t tiles = (m + 63) / 64;
  std::vector<int32_t> lane_set((size_t)tiles * 64, -1);
  for (int32_t s = 0; s < m; ++s) lane_set[s] = order[s];
  std::vector<std::vector<TilePlan>> plans(S, std::vector<TilePlan>(tiles));
  {
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt, 16u));
    if ((int64_t)tiles * S < 8) nt = 1;
    std::vector<std::thread> pool;
    std::atomic<int> pool_failed{0};
    for (unsigned w = 0; w < nt; ++w)
      pool.emplace_back([&, w]() {
        try {
          for (int64_t u = (int64_t)w; u < (int64_t)tiles * S; u += nt) {
            const int si = (int)(u / tiles);
            const int32_t t = (int32_t)(u % tiles);
            plan_tile_b128(starts[si], std::min(width, g - starts[si]), Gp, Gi, &lane_set[(size_t)t * 64], plans[si][t]);
          }
        } catch (...) {
          pool_failed.store(1);
        }
      });
    for (auto& th : pool) th.join();
    if (pool_failed.load()) throw std::bad_alloc();
  }
  // longest-processing-time on the steps summed over slices
  std::vector<int64_t> tot(tiles, 0);
  for (int si = 0; si < S; ++si)
    for (int32_t t = 0; t < tiles; ++t) tot[t] += plans[si][t].steps;
  std::vector<std::vector<int32_t>> mine;
  // The share an age class can take depends on how long its stream is: with ~5 tiles per wavefront (C2: 5,000 sets)
  // the youngest four lag most and get 0.46 of an equal share, with ~50 (50,000 sets) 0.57 (in-kernel stamps per
  // wavefront, tools/bench_spmm.py --ablate 4: with the C2 shares the youngest class of a 50k-set plan sat 40 % of
  // the gather phase at the end-of-slice barrier).  Interpolated in log(tiles) between the two measured plans.
  double share[4];
  {
    static const double lo[4] = {1.53, 1.20, 0.81, 0.46}, hi[4] = {1.40, 1.16, 0.87, 0.57};
    double t = (std::log((double)std::max(tiles, 1)) - std::log(80.0)) / (std::log(800.0) - std::log(80.0));
    t = std::min(1.0, std::max(0.0, t));
    for (int q = 0; q < 4; ++q) share[q] = lo[q] + t * (hi[q] - lo[q]);
  }
  assign_tiles(tot, waves, mine, share);
  pp.wave_tile_off.assign(waves + 1, 0);
  std::vector<int32_t> ktile;   // wave-stream order -> tile
  for (int w = 0; w < waves; ++w) {
    pp.wave_tile_off[w] = (int32_t)ktile.size();
    for (int32_t t : mine[w]) ktile.push_back(t);
  }
  pp.wave_tile_off[waves] = (int32_t)ktile.size();
  const size_t nk = ktile.size() + 1;   // + sentinel
  pp.meta_j.assign(nk * 64, -1);
  pp.meta_w.assign(nk * 64, 0.0);
  pp.meta_k.assign(nk * 64, 0.0);
  for (size_t k = 0; k + 1 < nk; ++k)
    for (int l = 0; l < 64; ++l) {
      const int32_t j = lane_set[(size_t)ktile[k] * 64 + l];
      if (j < 0) continue;
      const double size = (double)(Gp[j + 1] - Gp[j]);
      pp.meta_j[k * 64 + l] = j;
      pp.meta_w[k * 64 + l] = 1.0 / (1e-8 + size);   // R/plaid.R:75-76
      pp.meta_k[k * 64 + l] = size;
    }
  pp.slices.resize(S);
  pp.chunks = 0;
  for (int si = 0; si < S; ++si) {
    PairSliceHost& ps = pp.slices[si];
    ps.g0 = starts[si];
    ps.gs = std::min(width, g - starts[si]);
    ps.wave_chunk_off.assign(waves + 1, 0);
    int32_t chunk = 0;
    for (int w = 0; w < waves; ++w) {
      ps.wave_chunk_off[w] = chunk;
      for (int32_t t : mine[w]) {
        chunk += plans[si][t].steps / 8;
        ps.wtile_end.push_back(chunk);
      }
    }
    ps.wave_chunk_off[waves] = chunk;
    ps.wtile_end.push_back(-1);
    pp.chunks += chunk;
    ps.idx.assign(((size_t)chunk + 12) * 64 * 8, (uint16_t)ps.gs);
    int32_t c = 0;
    for (int w = 0; w < waves; ++w)
      for (int32_t t : mine[w]) {
        const TilePlan& tp = plans[si][t];
        for (int32_t st = 0; st < tp.steps; ++st) {
          const int32_t ch = c + st / 8, e = st & 7;
          for (int l = 0; l < 64; ++l) ps.idx[(((size_t)ch * 64) + l) * 8 + e] = tp.idx[(size_t)st * 64 + l];
        }
        c += tp.steps / 8;
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

namespace plaidhip {
int spmm_block_for_genes(int32_t g) {
#ifdef PLAIDHIP_DIAG
  static const char* e = getenv("PLAIDHIP_SPMM_BLOCK");   // tuning knob (tools/ build)
  if (e && g > 2048) return atoi(e) == 512 ? 512 : 1024;
#endif
  return g > 8192 ? 1024 : (g > 2048 ? 512 : 256);
}
}  // namespace plaidhip

extern "C" int plaidhip_geneset_create(plaidhip_ctx* ctx, int32_t g, int32_t m, const int32_t* Gp,
                                       const int32_t* Gi, plaidhip_geneset** out) try {
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
  struct Guard {   // (an exception on the way -- std::bad_alloc from a plan -- must not leak what exists so far)
    plaidhip_geneset* p;
    ~Guard() { if (p) plaidhip_geneset_destroy(p); }
  } guard{gs};
  gs->ctx = ctx;
  gs->g = g;
  gs->m = m;
  gs->z = z;
  gs->h_Gp.assign(Gp, Gp + m + 1);
  gs->h_Gi.assign(Gi, Gi + z);
  gs->tiles = (m + 63) / 64;

  int rc = PLAIDHIP_OK;
#if defined(PLAIDHIP_DIAG) && defined(PLAIDHIP_KEEP_HOST_PLANS)   // tools/plan_probe: PLAIDHIP_TSW=1 prints where the preparation spends its time
  const auto tsw0 = std::chrono::steady_clock::now();
  auto TSW = [&](const char* tag) {
    if (getenv("PLAIDHIP_TSW"))
      fprintf(stderr, "[tsw] %-24s %8.1f ms\n", tag, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tsw0).count());
  };
#else
  auto TSW = [](const char*) {};
#endif
  std::vector<int32_t> total(m);
  for (int32_t j = 0; j < m; ++j) total[j] = Gp[j + 1] - Gp[j];
  if (m > 0) {
    // gene slices: the LDS holds at most kMaxLdsGenes 8-byte entries; wider matrices are consumed
    // slice by slice (equal widths, even starts so 16-byte column loads stay aligned)
    const int32_t nsl = (g + kMaxLdsGenes - 1) / kMaxLdsGenes;
    int32_t width = (g + nsl - 1) / nsl;
    width = (width + 1) & ~1;
    for (int32_t g0 = 0; g0 < g; g0 += width) {
      const int32_t gsz = std::min(width, g - g0);
      // sub-pattern of this slice, rows re-based to the slice
      std::vector<int32_t> sp(m + 1, 0), si;
      si.reserve((size_t)z / nsl + 16);
      for (int32_t j = 0; j < m; ++j) {
        for (int32_t p = Gp[j]; p < Gp[j + 1]; ++p)
          if (Gi[p] >= g0 && Gi[p] < g0 + gsz) si.push_back(Gi[p] - g0);
        sp[j + 1] = (int32_t)si.size();
      }
      HostPlan hp;
      plaidhip_slice sl;
      sl.g0 = g0;
      sl.gs = gsz;
      sl.waves = spmm_block_for_genes(gsz) / 64;
      build_plan(gsz, m, sp.data(), si.data(), total.data(), sl.waves, hp);
      sl.chunks = hp.chunks;
      gs->chunks += hp.chunks;
      gs->slices.push_back(sl);
      plaidhip_slice& d = gs->slices.back();
      if ((rc = upload(ctx, hp.idx, &d.d_tile_idx)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, hp.wave_chunk_off, &d.d_wave_chunk_off)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, hp.wave_tile_off, &d.d_wave_tile_off)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, hp.wtile_end, &d.d_wtile_end)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, hp.meta_j, &d.d_meta_j)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, hp.meta_w, &d.d_meta_w)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, hp.meta_k, &d.d_meta_k)) != PLAIDHIP_OK) goto fail;
      // host vectors die at scope exit: make sure the copies have landed
      if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = PLAIDHIP_EHIP; goto fail; }
    }
  }
  TSW("one-column plans");
  gs->rows_in_order = true;   // tiles take the sets in decreasing-size order: identity iff already sorted
  for (int32_t j = 0; j + 1 < m; ++j)
    if (Gp[j + 2] - Gp[j + 1] > Gp[j + 1] - Gp[j]) { gs->rows_in_order = false; break; }
  if (m > 0) {
    // pair plan (two sample columns per LDS entry) for the dense-X kernel
    PairPlanHost pp;
    std::vector<plaidhip_pair_slice_dev> hd;
    gs->pair.waves = 16;
    build_pair_plan(g, m, Gp, Gi, gs->pair.waves, pp);
    TSW("pair plan built");
    gs->pair.chunks = pp.chunks;
    if ((rc = upload(ctx, pp.wave_tile_off, &gs->pair.d_wave_tile_off)) != PLAIDHIP_OK) goto fail;
    if ((rc = upload(ctx, pp.meta_j, &gs->pair.d_meta_j)) != PLAIDHIP_OK) goto fail;
    if ((rc = upload(ctx, pp.meta_w, &gs->pair.d_meta_w)) != PLAIDHIP_OK) goto fail;
    if ((rc = upload(ctx, pp.meta_k, &gs->pair.d_meta_k)) != PLAIDHIP_OK) goto fail;
    for (PairSliceHost& ps : pp.slices) {
      plaidhip_pair_slice d;
      d.g0 = ps.g0;
      d.gs = ps.gs;
      gs->pair.slices.push_back(d);
      plaidhip_pair_slice& dd = gs->pair.slices.back();
      if ((rc = upload(ctx, ps.idx, &dd.d_tile_idx)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, ps.wave_chunk_off, &dd.d_wave_chunk_off)) != PLAIDHIP_OK) goto fail;
      if ((rc = upload(ctx, ps.wtile_end, &dd.d_wtile_end)) != PLAIDHIP_OK) goto fail;
    }
    for (const plaidhip_pair_slice& d : gs->pair.slices)
      hd.push_back(plaidhip_pair_slice_dev{d.d_tile_idx, d.d_wave_chunk_off, d.d_wtile_end, d.g0, d.gs});
    if ((rc = upload(ctx, hd, &gs->pair.d_slices)) != PLAIDHIP_OK) goto fail;
    gs->pair.ktiles = pp.wave_tile_off[gs->pair.waves];
    if (gs->pair.slices.size() > 1) {
      gs->pair.partial_wgs = ctx->num_cu;
      const size_t bytes = (size_t)gs->pair.partial_wgs * (gs->pair.ktiles + 1) * 64 * 2 * sizeof(double);
      if (hipMalloc(reinterpret_cast<void**>(&gs->pair.d_partial), bytes) != hipSuccess) {
        set_error("hipMalloc(%zu) for the slice partial sums failed", bytes);
        rc = PLAIDHIP_ENOMEM;
        goto fail;
      }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = PLAIDHIP_EHIP; goto fail; }
  }
  TSW("pair plan uploaded");
  if (m > 0) {
    // scatter plan (sparse X): gene-major membership in 128-id segments (one dword = 2 ids per lane) per (chunk of sets, gene)
    plaidhip_scatter_plan& sp = gs->scatter;
    // as few chunks as the LDS allows: every chunk is one more pass over the column's stored values.  (Dealing
    // eight smaller chunks to the eight XCDs so that an XCD's L2 holds only its share of the id lists was measured:
    // 1.7x SLOWER -- the passes are bound by their chains of dependent loads, not by L2 misses.)
    sp.nch = (m + kScatterChunk - 1) / kScatterChunk;
    {   // blocks of kScatterBlock consecutive sets dealt round-robin to the chunks (common.h); ch is a multiple of 16: the trash slots keep their banks
      const int32_t nb = (m + kScatterBlock - 1) / kScatterBlock;
      sp.ch = ((nb + sp.nch - 1) / sp.nch) * kScatterBlock;     // <= kScatterChunk: nb <= 17 nch
    }
    {
      int32_t kmax = 1;
      for (int32_t j = 0; j < m; ++j) kmax = std::max(kmax, Gp[j + 1] - Gp[j]);
      sp.kbits = 1;
      while (((int64_t)1 << sp.kbits) <= (int64_t)kmax) ++sp.kbits;   // kmax < 2^kbits
    }
    std::vector<int32_t> cnt((size_t)sp.nch * g, 0);
    for (int32_t j = 0; j < m; ++j)
      for (int32_t p = Gp[j]; p < Gp[j + 1]; ++p) ++cnt[(size_t)scatter_chunk_of(j, sp.nch) * g + Gi[p]];
    std::vector<int32_t> seg((size_t)sp.nch * g + 1, 0);
    for (size_t i = 0; i < cnt.size(); ++i) seg[i + 1] = seg[i] + (cnt[i] + 127) / 128;
    sp.nseg = seg.back();
    // padded slots add into trash accumulators behind the chunk (no compare, no branch per lane; slot ch + lane % 16:
    // no bank collision among the padding lanes of a 16-lane group); + one all-padding segment (index nseg)
    std::vector<uint16_t> ids(((size_t)sp.nseg + 1) * 128);
    for (size_t q = 0; q < ids.size(); ++q) ids[q] = (uint16_t)(sp.ch + (int)(((q & 127) >> 1) & 15));
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int32_t j = 0; j < m; ++j)
      for (int32_t p = Gp[j]; p < Gp[j + 1]; ++p) {
        const size_t cell = (size_t)scatter_chunk_of(j, sp.nch) * g + Gi[p];
        ids[(size_t)seg[cell] * 128 + cnt[cell]++] = (uint16_t)scatter_slot_of(j, sp.nch);
      }
    TSW("scatter lists filled");
    // Bank-conflict-free order inside every (chunk, gene) list.  The kernel turns a segment into two ds_add_f64
    // wave-instructions (the low and the high u16 of each lane's dword).  Measured (tools/ubench/lds_atomics.hip): the
    // LDS executes a 64-bit atomic in four groups of 16 consecutive lanes, 2 cycles per group when the 16 accumulators
    // sit on 16 different 8-byte banks (slot mod 16), and N times that for an N-way collision -- 8.4 cycles per
    // instruction conflict-free, 32 with random ids.  The ids of a list are static, so they are dealt to the
    // (instruction, 16-lane group) cells such that a group holds each residue mod 16 at most once where the counts
    // allow it (a residue with more ids than the list has groups keeps the few collisions left), and the padding
    // lanes of a group get trash accumulators on the banks the group does not use.
    {
      // (cells are independent: a few threads, each with its own scratch, take blocks of cells -- this ordering was a third
      // of the time plaidhip_geneset_create takes on a 61,459-set collection)
      const size_t ncell = seg.size() - 1;
      unsigned nt = std::thread::hardware_concurrency();
      nt = std::max(1u, std::min(nt, 16u));
      if (ncell < 4096) nt = 1;
      auto order_cells = [&](size_t cell_lo, size_t cell_hi) {
      std::vector<uint16_t> byres[16];
      std::vector<int> gload, gres;    // ids per group, ids of the current residue per group
      std::vector<std::vector<uint16_t>> grp;
      for (size_t cell = cell_lo; cell < cell_hi; ++cell) {
        const int N = cnt[cell];
        const int nsg = seg[cell + 1] - seg[cell];
        uint16_t* base = ids.data() + (size_t)seg[cell] * 128;
        if (N == 0) continue;
        // instructions of 64 lanes are filled one after the other (their number, not their occupancy, is the cost):
        // a list of N ids takes ceil(N / 64) of the 2 * nsg instruction slots; an unused second instruction of the
        // last segment is marked by 0xffff in every high half and skipped by the kernel
        const int nins = (N + 63) / 64, ngr = 4 * nins;
        for (auto& v : byres) v.clear();
        for (int q = 0; q < N; ++q) byres[base[q] & 15].push_back(base[q]);
        int order[16];
        for (int r = 0; r < 16; ++r) order[r] = r;
        std::sort(order, order + 16, [&](int x, int y) { return byres[x].size() > byres[y].size(); });
        gload.assign(ngr, 0);
        grp.assign(ngr, std::vector<uint16_t>());
        for (int oi = 0; oi < 16; ++oi) {
          const std::vector<uint16_t>& cls = byres[order[oi]];
          if (cls.empty()) break;
          gres.assign(ngr, 0);
          for (uint16_t id : cls) {
            int best = -1;
            for (int gq = 0; gq < ngr; ++gq) {
              if (gload[gq] >= 16) continue;
              if (best < 0 || gres[gq] < gres[best] || (gres[gq] == gres[best] && gload[gq] < gload[best])) best = gq;
            }
            grp[best].push_back(id);
            ++gload[best];
            ++gres[best];
          }
        }
        // group q = (instruction q / 4, lanes 16 * (q & 3) ..): instruction i is the low (i even) or high (i odd)
        // half of segment i / 2; lane l holds list positions 2 l and 2 l + 1 of its segment.  An unused second
        // instruction of the last segment keeps the trash ids every slot starts with: the kernel applies both halves of
        // every segment without a test (a test per (value, chunk) costs more than the spare LDS atomic)
        if (nins & 1)   // (the sequential fill above left list entries in those high halves)
          for (int l = 0; l < 64; ++l) base[(size_t)(nsg - 1) * 128 + 2 * l + 1] = (uint16_t)(sp.ch + (l & 15));
        for (int gq = 0; gq < ngr; ++gq) {
          const int ins = gq >> 2, quarter = gq & 3, sgm = ins >> 1, half = ins & 1;
          bool used[16] = {false};
          for (uint16_t id : grp[gq]) used[id & 15] = true;
          int next_free = 0;
          for (int l = 0; l < 16; ++l) {
            uint16_t v;
            if (l < (int)grp[gq].size()) {
              v = grp[gq][l];
            } else {
              while (used[(sp.ch + next_free) & 15]) ++next_free;     // a trash accumulator on an idle bank
              v = (uint16_t)(sp.ch + next_free);
              used[(sp.ch + next_free) & 15] = true;
            }
            base[(size_t)sgm * 128 + 2 * (16 * quarter + l) + half] = v;
          }
        }
      }
      };
      if (nt == 1) {
        order_cells(0, ncell);
      } else {
        std::atomic<size_t> next{0};
        std::atomic<int> pool_failed{0};
        const size_t blk = 2048;
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nt; ++w)
          pool.emplace_back([&]() {
            try {
              for (;;) {
                const size_t lo = next.fetch_add(blk);
                if (lo >= ncell) return;
                order_cells(lo, std::min(ncell, lo + blk));
              }
            } catch (...) {
              pool_failed.store(1);
            }
          });
        for (auto& th : pool) th.join();
        if (pool_failed.load()) throw std::bad_alloc();
      }
    }
    TSW("scatter lists ordered");
    std::vector<double> w(m), k(m);
    for (int32_t j = 0; j < m; ++j) {
      k[j] = (double)(Gp[j + 1] - Gp[j]);
      w[j] = 1.0 / (1e-8 + k[j]);   // R/plaid.R:75-76
    }
    if ((rc = upload(ctx, seg, &sp.d_seg)) != PLAIDHIP_OK) goto fail;
    if ((rc = upload(ctx, ids, &sp.d_ids)) != PLAIDHIP_OK) goto fail;
#ifdef PLAIDHIP_KEEP_HOST_PLANS   // host-only tools (tools/plan_probe): the checker below reads the plan back
    sp.h_seg = seg;
    sp.h_ids = ids;
#endif
    if ((rc = upload(ctx, w, &sp.d_w)) != PLAIDHIP_OK) goto fail;
    if ((rc = upload(ctx, k, &sp.d_k)) != PLAIDHIP_OK) goto fail;
    {
      // per set {size x weight, weight} for each statistic (mean: weight = 1 / (1e-8 + size); sum: weight = 1): the
      // scatter epilogue's  alpha * (sum * weight) + beta * (size * weight)  then needs no select and one product less
      // (size * weight is the same IEEE product here as on the device)
      std::vector<double> kw((size_t)4 * m);
      for (int32_t j = 0; j < m; ++j) {
        kw[2 * (size_t)j] = k[j] * w[j];
        kw[2 * (size_t)j + 1] = w[j];
        kw[2 * ((size_t)m + j)] = k[j];
        kw[2 * ((size_t)m + j) + 1] = 1.0;
      }
      if ((rc = upload(ctx, kw, &sp.d_kw)) != PLAIDHIP_OK) goto fail;
      // the MEAN score of a sample column without the crossprod: mean_j S[j, c] = alpha * sum_i x[i, c] u[i] + beta * kappa
      // with u[i] = (1 / m) sum_{j containing i} weight_j and kappa = (1 / m) sum_j size_j weight_j (per statistic, like
      // kw above).  The scatter launch that selects the column medians on the fly brackets them around this mean
      // (kernels_norm.hip: colmean_predict_kernel).
      std::vector<double> uv((size_t)2 * g, 0.0);
      sp.kappa[0] = sp.kappa[1] = 0.0;
      for (int32_t j = 0; j < m; ++j) {
        for (int32_t p = Gp[j]; p < Gp[j + 1]; ++p) { uv[(size_t)Gi[p]] += w[j]; uv[(size_t)g + Gi[p]] += 1.0; }
        sp.kappa[0] += k[j] * w[j];
        sp.kappa[1] += k[j];
      }
      for (double& x : uv) x /= (double)m;
      sp.kappa[0] /= (double)m;
      sp.kappa[1] /= (double)m;
      if ((rc = upload(ctx, uv, &sp.d_u)) != PLAIDHIP_OK) goto fail;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = PLAIDHIP_EHIP; goto fail; }
  }
  TSW("done");
  guard.p = nullptr;
  *out = gs;
  return PLAIDHIP_OK;
fail:
  return rc;   // (the guard destroys gs)
} catch (...) { return plaidhip::on_exception(); }

extern "C" int plaidhip_geneset_destroy(plaidhip_geneset* gs) try {
  if (!gs) return PLAIDHIP_OK;
  for (plaidhip_slice& d : gs->slices) {
    hipFree(d.d_tile_idx);
    hipFree(d.d_wave_chunk_off);
    hipFree(d.d_wave_tile_off);
    hipFree(d.d_wtile_end);
    hipFree(d.d_meta_j);
    hipFree(d.d_meta_w);
    hipFree(d.d_meta_k);
  }
  hipFree(gs->pair.d_wave_tile_off);
  hipFree(gs->pair.d_meta_j);
  hipFree(gs->pair.d_meta_w);
  hipFree(gs->pair.d_meta_k);
  hipFree(gs->d_dense_g);
  hipFree(gs->scatter.d_seg);
  hipFree(gs->scatter.d_ids);
  hipFree(gs->scatter.d_w);
  hipFree(gs->scatter.d_k);
  hipFree(gs->scatter.d_kw);
  hipFree(gs->scatter.d_u);
  hipFree(gs->pair.d_slices);
  hipFree(gs->pair.d_partial);
  for (plaidhip_pair_slice& d : gs->pair.slices) {
    hipFree(d.d_tile_idx);
    hipFree(d.d_wave_chunk_off);
    hipFree(d.d_wtile_end);
  }
  delete gs;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

extern "C" int plaidhip_geneset_info(const plaidhip_geneset* gs, int64_t info[8]) try {
  PH_REQUIRE(gs && info, "geneset_info: null argument");
  std::memset(info, 0, 8 * sizeof(int64_t));
  info[0] = gs->g;
  info[1] = gs->m;
  info[2] = gs->z;
  info[3] = gs->chunks * 64 * 8;
  info[4] = gs->tiles;
  info[5] = (int64_t)gs->slices.size();   // gene slices (1 when the whole column fits the LDS)
  info[6] = gs->slices.empty() ? 0 : gs->slices[0].waves;
  info[7] = gs->pair.chunks * 64 * 8;     // padded index slots of the pair plan (dense-X kernel)
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

// Diagnostic / test hook (not part of include/plaidhip.h): builds the pair plan on the host only
// and checks it.  out[0]=slices out[1]=chunks (all slices) out[2]=memberships found in the plan
// out[3]=conflicts (two lanes of one ds_read_b128 lane group on the same 16-byte slot in one
// step) out[4]=memberships scheduled for the wrong set / twice / out of slice.
extern "C" int plaidhip_debug_pair_plan_check(int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi,
                                              int32_t waves, int64_t out[8]) try {
  PairPlanHost pp;
  build_pair_plan(g, m, Gp, Gi, waves, pp);
  int64_t found = 0, conflicts = 0, wrong = 0;
  std::vector<uint8_t> seen((size_t)Gp[m], 0);
  for (const PairSliceHost& ps : pp.slices) {
    for (int w = 0; w < waves; ++w) {
      int32_t k = pp.wave_tile_off[w];
      for (int32_t ch = ps.wave_chunk_off[w]; ch < ps.wave_chunk_off[w + 1]; ++ch) {
        for (int e = 0; e < 8; ++e) {
          for (int q = 0; q < 4; ++q) {
            uint32_t slots = 0;
            for (int l = 0; l < 16; ++l) {
              const int lane = kB128Groups[q][l];
              const int32_t id = ps.idx[(((size_t)ch * 64) + lane) * 8 + e];
              if (id >= ps.gs + kPadSlotsPair) { ++wrong; continue; }
              if (slots & (1u << (id & 15))) ++conflicts;
              slots |= 1u << (id & 15);
              if (id >= ps.gs) continue;
              const int32_t j = pp.meta_j[(size_t)k * 64 + lane];
              if (j < 0) { ++wrong; continue; }
              const int32_t* lo = std::lower_bound(Gi + Gp[j], Gi + Gp[j + 1], ps.g0 + id);
              if (lo == Gi + Gp[j + 1] || *lo != ps.g0 + id || seen[lo - Gi]) { ++wrong; continue; }
              seen[lo - Gi] = 1;
              ++found;
            }
          }
        }
        if (ch + 1 == ps.wtile_end[k]) ++k;
      }
      if (k != pp.wave_tile_off[w + 1]) ++wrong;
    }
  }
  out[0] = (int64_t)pp.slices.size();
  out[1] = pp.chunks;
  out[2] = found;
  out[3] = conflicts;
  out[4] = wrong;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

#ifdef PLAIDHIP_KEEP_HOST_PLANS
// Test hook of the host-only tools build: checks the scatter plan (gene-major id segments per chunk of sets) against the
// pattern.  out[0] = chunks, out[1] = segments, out[2] = memberships found (each exactly once, in the list of its gene and
// chunk), out[3] = wrong / repeated / out-of-chunk ids, out[4] = LDS atomic wave-instructions over all lists, two per
// segment (what a stored value of that gene costs), out[5] = ids that share their 8-byte bank (id mod 16) with an earlier id of the same
// 16-lane group of an instruction (each costs one more 2-cycle pass of that group).
extern "C" int plaidhip_debug_scatter_plan_check(const plaidhip_geneset* gs, int64_t out[8]) try {
  PH_REQUIRE(gs && out, "scatter_plan_check: null argument");
  const plaidhip_scatter_plan& sp = gs->scatter;
  const int32_t g = gs->g, m = gs->m;
  const int32_t* Gp = gs->h_Gp.data();
  const int32_t* Gi = gs->h_Gi.data();
  std::vector<uint8_t> seen((size_t)gs->z, 0);
  int64_t found = 0, wrong = 0, instr = 0, coll = 0;
  for (int32_t ch = 0; ch < sp.nch; ++ch)
    for (int32_t gene = 0; gene < g; ++gene) {
      const size_t cell = (size_t)ch * g + gene;
      for (int32_t sgm = sp.h_seg[cell]; sgm < sp.h_seg[cell + 1]; ++sgm) {
        const uint16_t* base = sp.h_ids.data() + (size_t)sgm * 128;
        for (int half = 0; half < 2; ++half) {
          ++instr;
          for (int quarter = 0; quarter < 4; ++quarter) {
            uint32_t banks = 0;
            for (int l = 0; l < 16; ++l) {
              const uint16_t id = base[2 * (16 * quarter + l) + half];
              if (banks & (1u << (id & 15))) ++coll;
              banks |= 1u << (id & 15);
              if (id >= sp.ch) { if (id >= sp.ch + kScatterTrash) ++wrong; continue; }   // a trash accumulator
              const int32_t j = scatter_set_of(ch, id, sp.nch);
              if (j >= m) { ++wrong; continue; }
              const int32_t* lo = std::lower_bound(Gi + Gp[j], Gi + Gp[j + 1], gene);
              if (lo == Gi + Gp[j + 1] || *lo != gene || seen[lo - Gi]) { ++wrong; continue; }
              seen[lo - Gi] = 1;
              ++found;
            }
          }
        }
      }
    }
  out[0] = sp.nch;
  out[1] = sp.nseg;
  out[2] = found;
  out[3] = wrong;
  out[4] = instr;
  out[5] = coll;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }
#endif
