// Sanitizer run of the library's host code (ASan / UBSan on the CPU build): the index planners
// of geneset.cpp (tile schedules, edge colouring, pair slices, scatter segments -- 840 lines of index arithmetic),
// the GMT parser and gmt2mat of gmt.cpp, and the p-value tails of stats.cpp.  Built and run by
// `make -C plaid_amd/csrc host-asan`; any sanitizer report aborts with a non-zero status.
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../plaid_amd/csrc/common.h"

extern "C" int plaidhip_debug_pair_plan_check(int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, int32_t waves,
                                              int64_t out[8]);

#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } \
  } while (0)

static void random_sets(std::mt19937_64& rng, int32_t g, int32_t m, int kmin, int kmax, bool sorted, std::vector<int32_t>& Gp,
                        std::vector<int32_t>& Gi) {
  std::vector<int32_t> sizes(m);
  std::uniform_real_distribution<double> u(std::log((double)kmin), std::log((double)kmax));
  for (auto& k : sizes) k = std::min<int32_t>(g, std::max<int32_t>(0, (int32_t)std::lround(std::exp(u(rng))) - (rng() % 11 == 0 ? 1000000 : 0)));
  if (sorted) std::sort(sizes.rbegin(), sizes.rend());
  Gp.assign(1, 0);
  Gi.clear();
  std::vector<int32_t> perm(g);
  for (int32_t i = 0; i < g; ++i) perm[i] = i;
  for (int32_t j = 0; j < m; ++j) {
    const int32_t k = std::max<int32_t>(0, sizes[j]);
    for (int32_t t = 0; t < k; ++t) std::swap(perm[t], perm[t + rng() % (g - t)]);
    std::vector<int32_t> mem(perm.begin(), perm.begin() + k);
    std::sort(mem.begin(), mem.end());
    Gi.insert(Gi.end(), mem.begin(), mem.end());
    Gp.push_back((int32_t)Gi.size());
  }
}

static void plans() {
  std::mt19937_64 rng(7);
  struct Case { int32_t g, m; int kmin, kmax; bool sorted; };
  const Case cases[] = {{37, 3, 1, 30, true},      {1000, 70, 1, 200, false},  {10224, 130, 5, 300, true},
                        {10226, 129, 5, 300, false}, {20000, 700, 15, 500, true}, {20448, 65, 1, 2000, false},
                        {20449, 64, 15, 500, true}, {45000, 300, 15, 500, false}, {1, 5, 1, 1, true},
                        {20000, 5000, 15, 500, true}, {20000, 50000, 15, 500, true}};
  plaidhip_ctx ctx;
  for (const Case& c : cases) {
    std::vector<int32_t> Gp, Gi;
    random_sets(rng, c.g, c.m, c.kmin, c.kmax, c.sorted, Gp, Gi);
    int64_t out[8] = {0};
    REQUIRE(plaidhip_debug_pair_plan_check(c.g, c.m, Gp.data(), Gi.data(), 16, out) == PLAIDHIP_OK);
    REQUIRE(out[2] == (int64_t)Gi.size());   // every membership scheduled exactly once
    REQUIRE(out[4] == 0);                    // none for the wrong set, twice, or outside its slice
    plaidhip_geneset* gs = nullptr;
    REQUIRE(plaidhip_geneset_create(&ctx, c.g, c.m, Gp.data(), Gi.data(), &gs) == PLAIDHIP_OK);
    int64_t info[8];
    REQUIRE(plaidhip_geneset_info(gs, info) == PLAIDHIP_OK);
    REQUIRE(info[0] == c.g && info[1] == c.m && info[2] == (int64_t)Gi.size() && info[3] >= info[2]);
    REQUIRE(plaidhip_geneset_destroy(gs) == PLAIDHIP_OK);
    printf("  plan g=%d m=%d z=%zu: pair slices %" PRId64 ", padded slots %" PRId64 " (one-column) / %" PRId64 " (pair)\n", c.g, c.m,
           Gi.size(), out[0], info[3], info[7]);
  }
  // rejected inputs: a decreasing pointer array, an index out of range
  {
    plaidhip_geneset* gs = nullptr;
    const int32_t badp[] = {0, 3, 2}, gi[] = {0, 1, 2};
    REQUIRE(plaidhip_geneset_create(&ctx, 5, 2, badp, gi, &gs) == PLAIDHIP_EINVAL);
    const int32_t p2[] = {0, 2}, gi2[] = {0, 9};
    REQUIRE(plaidhip_geneset_create(&ctx, 5, 1, p2, gi2, &gs) == PLAIDHIP_EINVAL);
  }
}

static void gmt() {
  std::mt19937_64 rng(11);
  for (int rep = 0; rep < 40; ++rep) {
    // random GMT text: comments, blank lines, empty fields, "NA", repeats, spaces inside the gene field, CRLF, no
    // trailing newline, very long lines
    std::string text;
    const int nsets = (int)(rng() % 60);
    for (int j = 0; j < nsets; ++j) {
      if (rng() % 9 == 0) text += "# comment\tline\n";
      if (rng() % 13 == 0) text += "\n";
      text += "SET_" + std::to_string(rng() % 40) + "\t" + (rng() % 3 ? "src" : "") ;
      const int k = (int)(rng() % (rep == 7 ? 20000 : 50));
      for (int t = 0; t < k; ++t) {
        text += (rng() % 7 == 0) ? " " : "\t";
        const int r = (int)(rng() % 10);
        text += r == 0 ? "NA" : (r == 1 ? "" : "G" + std::to_string(rng() % 300));
      }
      text += (rng() % 5 == 0) ? "\r\n" : "\n";
    }
    if (!text.empty() && rng() % 2) text.pop_back();
    for (int raw = 0; raw < 2; ++raw) {
      plaidhip_gmt* gm = nullptr;
      REQUIRE(plaidhip_gmt_parse(text.data(), (int64_t)text.size(), raw, (int)(rng() % 2), rng() % 4 == 0 ? 5 : 0, &gm) == PLAIDHIP_OK);
      const int64_t ns = plaidhip_gmt_nsets(gm);
      int64_t total = 0;
      for (int64_t j = 0; j < ns; ++j) {
        REQUIRE(plaidhip_gmt_set_name(gm, j) != nullptr);
        const int64_t k = plaidhip_gmt_set_size(gm, j);
        for (int64_t t = 0; t < k; ++t) total += (int64_t)strlen(plaidhip_gmt_set_gene(gm, j, t));
      }
      int64_t nb = 0;
      REQUIRE(plaidhip_gmt_text(gm, &nb) != nullptr || ns == 0);
      plaidhip_gmtmat* mat = nullptr;
      const char* bg[] = {"G1", "G2", "G299", "nope"};
      REQUIRE(plaidhip_gmt2mat(gm, rep % 3 == 0 ? 25 : -1, rep % 4 == 0 ? 7 : -1, rep % 5 == 0 ? bg : nullptr, rep % 5 == 0 ? 4 : 0, &mat) == PLAIDHIP_OK);
      int64_t dims[3];
      REQUIRE(plaidhip_gmtmat_dims(mat, dims) == PLAIDHIP_OK);
      const int32_t* p = plaidhip_gmtmat_p(mat);
      const int32_t* ii = plaidhip_gmtmat_i(mat);
      for (int64_t j = 0; j < dims[1]; ++j)
        for (int32_t q = p[j]; q < p[j + 1]; ++q) REQUIRE(ii[q] >= 0 && ii[q] < dims[0] && (q == p[j] || ii[q - 1] < ii[q]));
      REQUIRE(dims[1] == 0 || p[dims[1]] == dims[2]);
      for (int axis = 0; axis < 2; ++axis) {
        int64_t bytes = 0;
        const char* names = plaidhip_gmtmat_names(mat, axis, &bytes);
        REQUIRE(names != nullptr || dims[axis] == 0);
      }
      REQUIRE(plaidhip_gmtmat_destroy(mat) == PLAIDHIP_OK);
      REQUIRE(plaidhip_gmt_destroy(gm) == PLAIDHIP_OK);
      (void)total;
    }
  }
  printf("  gmt: 80 random texts parsed, listed and turned into matrices\n");
}

static void tails() {
  using namespace plaidhip;
  std::mt19937_64 rng(3);
  std::normal_distribution<double> N(0.0, 1.0);
  std::vector<double> p;
  for (int rep = 0; rep < 2000; ++rep) {
    const double k = 1 + (double)(rng() % 400), s1 = N(rng) * k, s2 = s1 * s1 / k + std::fabs(N(rng)) * k;
    double mean = 0, diff = 0;
    const double a = onesample_p(k, s1, s2, &mean);
    const double b = twosample_p(5000.0, k, s1, s2, N(rng) * 5000, 7000.0 + std::fabs(N(rng)) * 100, &diff);
    const double c = welch_p(N(rng), N(rng), std::fabs(N(rng)), std::fabs(N(rng)), 2 + (double)(rng() % 30), 2 + (double)(rng() % 30));
    for (double v : {a, b, c}) REQUIRE(std::isnan(v) || (v >= 0.0 && v <= 1.0));
    const double q[3] = {clamp_p(a), clamp_p(b), clamp_p(c)};
    for (int method = 0; method < 2; ++method) {
      const double m = combine_p(q, 3, method);
      REQUIRE(std::isnan(m) || (m >= 0.0 && m <= 1.0));
    }
    p.push_back(rep % 50 == 0 ? NAN : q[rep % 3]);
  }
  // degenerate inputs: zero variance, one observation, empty
  double tmp;
  (void)onesample_p(1.0, 3.0, 9.0, &tmp);
  (void)onesample_p(0.0, 0.0, 0.0, &tmp);
  (void)welch_p(1.0, 1.0, 0.0, 0.0, 2.0, 2.0);
  std::vector<double> q(p.size());
  p_adjust_fdr(p.data(), (int64_t)p.size(), q.data());
  p_adjust_fdr(p.data(), 0, q.data());
  for (size_t i = 0; i < p.size(); ++i) REQUIRE(std::isnan(p[i]) ? std::isnan(q[i]) : (q[i] >= p[i] - 1e-15 && q[i] <= 1.0));
  printf("  stats: 2000 random moment sets through the t / chi^2 / normal tails, combine_p and BH\n");
}

int main() {
  printf("[host-asan] planners\n");
  plans();
  printf("[host-asan] GMT parser / gmt2mat\n");
  gmt();
  printf("[host-asan] p-value tails\n");
  tails();
  printf("[host-asan] ok\n");
  return 0;
}
