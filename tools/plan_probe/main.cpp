// Host-only probe of the gene-set planners (geneset.cpp) on a collection read from a file: how long
// plaidhip_geneset_create takes, how many bytes it uploads, the slot efficiency (real / padded index slots) of the
// one-column and pair plans, and the pair plan's self-check (every membership scheduled exactly once, in its slice, for
// its set; bank conflicts beyond the deliberate two-way ones).  Built with g++ against the host stand-ins of the HIP
// runtime (tools/host_asan/hip_stubs.cpp) by `make -C plaid_amd/csrc plan-probe`; tests/test_host_logic.py runs it on a
// collection with the shape of the reference's benchmark (61,459 sets, Zipf gene popularity, an all-genes set).
// File: int32 g, int32 m, int32 Gp[m + 1], int32 Gi[Gp[m]] (little endian).  Prints one JSON object.
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../plaid_amd/csrc/common.h"

extern "C" int plaidhip_debug_pair_plan_check(int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, int32_t waves,
                                              int64_t out[8]);
extern "C" int plaidhip_debug_scatter_plan_check(const plaidhip_geneset* gs, int64_t out[8]);
extern "C" void plaidhip_stub_alloc_stats(int64_t* live, int64_t* peak, int64_t* total);

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: plan_probe <file> [check]\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  int32_t g = 0, m = 0;
  if (fread(&g, 4, 1, f) != 1 || fread(&m, 4, 1, f) != 1) return 2;
  std::vector<int32_t> Gp((size_t)m + 1);
  if (fread(Gp.data(), 4, Gp.size(), f) != Gp.size()) return 2;
  std::vector<int32_t> Gi((size_t)Gp[m]);
  if (!Gi.empty() && fread(Gi.data(), 4, Gi.size(), f) != Gi.size()) return 2;
  fclose(f);
  const bool check = argc > 2;
  plaidhip_ctx ctx;
  plaidhip_geneset* gs = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = plaidhip_geneset_create(&ctx, g, m, Gp.data(), Gi.data(), &gs);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc != PLAIDHIP_OK) { printf("{\"rc\": %d}\n", rc); return 1; }
  int64_t info[8], live = 0, peak = 0, total = 0, pc[8] = {0}, sc[8] = {0};
  plaidhip_geneset_info(gs, info);
  plaidhip_stub_alloc_stats(&live, &peak, &total);
  double sec_check = 0.0;
  if (check) {
    const auto t1 = std::chrono::steady_clock::now();
    if (plaidhip_debug_pair_plan_check(g, m, Gp.data(), Gi.data(), 16, pc) != PLAIDHIP_OK) return 1;
    if (plaidhip_debug_scatter_plan_check(gs, sc) != PLAIDHIP_OK) return 1;
    sec_check = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  }
  printf("{\"rc\": 0, \"g\": %d, \"m\": %d, \"z\": %" PRId64 ", \"create_s\": %.3f, \"device_bytes\": %" PRId64
         ", \"slots_one_column\": %" PRId64 ", \"slots_pair\": %" PRId64 ", \"gene_slices\": %" PRId64
         ", \"pair_slices\": %" PRId64 ", \"pair_found\": %" PRId64 ", \"pair_conflicts\": %" PRId64 ", \"pair_wrong\": %" PRId64
         ", \"scatter_chunks\": %" PRId64 ", \"scatter_segments\": %" PRId64 ", \"scatter_found\": %" PRId64
         ", \"scatter_wrong\": %" PRId64 ", \"scatter_instructions\": %" PRId64 ", \"scatter_collisions\": %" PRId64
         ", \"check_s\": %.3f}\n",
         g, m, info[2], sec, live, info[3], info[7], info[5], pc[0], pc[2], pc[3], pc[4], sc[0], sc[1], sc[2], sc[3], sc[4],
         sc[5], sec_check);
  plaidhip_geneset_destroy(gs);
  return 0;
}
