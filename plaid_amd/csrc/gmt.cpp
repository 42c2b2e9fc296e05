// Native GMT reader and gmt2mat (host only): the text side of the gene-set path.
//
// The reference builds the 0/1 genes x sets matrix in R (R/gmt-utils.R:19-66, 99-125): 50.9 s for a
// 50k-set collection (experiments/benchmark/benchmark-plaid.R:42), i.e. far longer than the GPU
// scoring it feeds.  This is the same rule set -- one pass over the text, hashed gene ids, counting
// sorts -- behind the C ABI of include/plaidhip.h (plaidhip_gmt_* / plaidhip_gmtmat_*).
//
// read.gmt (R/gmt-utils.R:99-125): one set per line; '#' starts a comment; fields are tab
// separated: name, source, genes...; the gene fields are re-joined with ' ' and split on ' ' or
// '\t' (:115-116), "", "NA" and repeats are dropped (:117, setdiff); add.source appends
// " (source)" to the name (:120-121); nrows limits the sets read.
// gmt2mat (R/gmt-utils.R:19-66): sets by decreasing size, stable (:25); repeated names dropped
// (:26); head(ntop) (:27); rows = bg or the genes by decreasing count over all sets, ties in name
// order (:30, sort(table(.))); head(max.genes) (:35); 0/1 entries (:47-60); rows re-ordered by
// decreasing number of sets, stable (:62).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "common.h"

using namespace plaidhip;

struct plaidhip_gmt {
  std::string buf;                                   // private copy of the text; tokens are NUL-terminated in place
  std::vector<std::string> names;                    // set names (may carry " (source)")
  std::vector<const char*> gene_name;                // interned gene names (into buf)
  std::unordered_map<std::string_view, int32_t> intern;
  std::vector<int64_t> set_ptr{0};                   // CSR over sets
  std::vector<int32_t> set_gene;                     // gene ids (file text: repeats inside a set already dropped)
  std::vector<int32_t> stamp;                        // per gene: last set that contained it
  std::string joined;                                // scratch for the bulk accessor
};

struct plaidhip_gmtmat {
  int64_t g = 0, m = 0, z = 0;
  std::vector<std::string> rownames, colnames;
  std::vector<int32_t> p, i;
  std::string joined[2];
};

namespace {

// genes of one set: text[b, e) split on '\t' (and on ' ' when reading a file); tokens are terminated
// in place, interned, and appended once per set.  File text drops "" and "NA" (R/gmt-utils.R:117).
void add_genes(plaidhip_gmt& g, char* text, size_t b, size_t e, bool file_rules) {
  const int32_t set_index = (int32_t)g.names.size() - 1;
  size_t tb = b;
  for (size_t k = b; k <= e; ++k) {
    if (k == e || text[k] == '\t' || (file_rules && text[k] == ' ')) {
      const size_t len = k - tb;
      if (len != 0 && !(file_rules && len == 2 && text[tb] == 'N' && text[tb + 1] == 'A')) {
        text[k] = '\0';   // k == e is the line end or the '#' (both ours to overwrite)
        auto it = g.intern.emplace(std::string_view(text + tb, len), (int32_t)g.gene_name.size());
        if (it.second) {
          g.gene_name.push_back(text + tb);
          g.stamp.push_back(-1);
        }
        const int32_t id = it.first->second;
        if (!file_rules) {
          g.set_gene.push_back(id);          // an in-memory list is taken as it is (repeats count in table(), :30)
        } else if (g.stamp[id] != set_index) {
          g.stamp[id] = set_index;           // setdiff() also drops repeats (:117)
          g.set_gene.push_back(id);
        }
      }
      tb = k + 1;
    }
  }
  g.set_ptr.push_back((int64_t)g.set_gene.size());
}

// raw = true: the in-memory exchange format (name TAB source TAB gene TAB gene ..., nothing
// filtered but empty tokens and repeats, no comments)
int parse_text(plaidhip_gmt& g, int add_source, int64_t nrows, bool raw) {
  char* text = g.buf.data();
  const size_t n = g.buf.size();
  g.intern.reserve(1 << 16);
  size_t b = 0;
  while (b < n) {
    size_t e = b;
    while (e < n && text[e] != '\n') ++e;
    const size_t next = e + 1;
    size_t le = e;                                     // line = [b, le)
    while (le > b && text[le - 1] == '\r') --le;
    if (!raw) {
      for (size_t k = b; k < le; ++k)
        if (text[k] == '#') { le = k; break; }           // comment.char = "#" (R/gmt-utils.R:106)
      bool blank = true;
      for (size_t k = b; k < le; ++k)
        if (text[k] != ' ' && text[k] != '\t' && text[k] != '\r') { blank = false; break; }
      if (blank) { b = next; continue; }
    } else if (le == b) {
      b = next;
      continue;
    }
    size_t t1 = b;
    while (t1 < le && text[t1] != '\t') ++t1;
    std::string nm(text + b, t1 - b);
    size_t t2 = le;
    if (t1 < le) {
      t2 = t1 + 1;
      while (t2 < le && text[t2] != '\t') ++t2;
    }
    if (add_source) {
      nm += " (";
      if (t1 < le) nm.append(text + t1 + 1, t2 - t1 - 1);
      else nm += "NA";
      nm += ")";
    }
    g.names.push_back(std::move(nm));
    if (t2 < le) add_genes(g, text, t2 + 1, le, !raw);
    else g.set_ptr.push_back((int64_t)g.set_gene.size());
    b = next;
    if (nrows > 0 && (int64_t)g.names.size() >= nrows) break;
  }
  return PLAIDHIP_OK;
}

void join_names(const std::vector<std::string>& v, std::string& out) {
  size_t tot = 0;
  for (const std::string& s : v) tot += s.size() + 1;
  out.clear();
  out.reserve(tot);
  for (size_t k = 0; k < v.size(); ++k) {
    if (k) out += '\n';
    out += v[k];
  }
}

}  // namespace

extern "C" {

int plaidhip_gmt_read(const char* path, int add_source, int64_t nrows, plaidhip_gmt** out) try {
  PH_REQUIRE(path && out, "gmt_read: null path/out");
  *out = nullptr;
  FILE* fh = fopen(path, "rb");
  if (!fh) {
    set_error("gmt_read: cannot open '%s'", path);
    return PLAIDHIP_EINVAL;
  }
  plaidhip_gmt* g = new plaidhip_gmt();
  char tmp[1 << 16];
  size_t got;
  while ((got = fread(tmp, 1, sizeof(tmp), fh)) > 0) g->buf.append(tmp, got);
  fclose(fh);
  g->buf.push_back('\n');   // room to terminate the last token
  parse_text(*g, add_source, nrows, false);
  *out = g;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_gmt_parse(const char* text, int64_t nbytes, int raw, int add_source, int64_t nrows, plaidhip_gmt** out) try {
  PH_REQUIRE(out && (text || nbytes == 0) && nbytes >= 0, "gmt_parse: bad arguments");
  plaidhip_gmt* g = new plaidhip_gmt();
  if (nbytes) g->buf.assign(text, (size_t)nbytes);
  g->buf.push_back('\n');
  parse_text(*g, add_source, nrows, raw != 0);
  *out = g;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int64_t plaidhip_gmt_nsets(const plaidhip_gmt* g) { return g ? (int64_t)g->names.size() : 0; }

const char* plaidhip_gmt_set_name(const plaidhip_gmt* g, int64_t j) {
  return (g && j >= 0 && j < (int64_t)g->names.size()) ? g->names[j].c_str() : nullptr;
}

int64_t plaidhip_gmt_set_size(const plaidhip_gmt* g, int64_t j) {
  return (g && j >= 0 && j < (int64_t)g->names.size()) ? g->set_ptr[j + 1] - g->set_ptr[j] : -1;
}

const char* plaidhip_gmt_set_gene(const plaidhip_gmt* g, int64_t j, int64_t k) {
  if (!g || j < 0 || j >= (int64_t)g->names.size() || k < 0 || k >= g->set_ptr[j + 1] - g->set_ptr[j]) return nullptr;
  return g->gene_name[g->set_gene[g->set_ptr[j] + k]];
}

// all sets as text: one line per set, name '\t' gene '\t' gene ... (for bulk transfer to a host language)
const char* plaidhip_gmt_text(plaidhip_gmt* g, int64_t* nbytes) {
  if (!g) return nullptr;
  std::string& o = g->joined;
  o.clear();
  o.reserve(g->buf.size());
  for (size_t j = 0; j < g->names.size(); ++j) {
    if (j) o += '\n';
    o += g->names[j];
    for (int64_t q = g->set_ptr[j]; q < g->set_ptr[j + 1]; ++q) {
      o += '\t';
      o += g->gene_name[g->set_gene[q]];
    }
  }
  if (nbytes) *nbytes = (int64_t)o.size();
  return o.c_str();
}

int plaidhip_gmt_destroy(plaidhip_gmt* g) try {
  delete g;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_gmt2mat(const plaidhip_gmt* gmt, int64_t max_genes, int64_t ntop, const char* const* bg, int64_t nbg,
                     plaidhip_gmtmat** out) try {
  PH_REQUIRE(gmt && out, "gmt2mat: null gmt/out");
  PH_REQUIRE(nbg <= 0 || bg, "gmt2mat: null bg");
  *out = nullptr;
  const int64_t ns = (int64_t)gmt->names.size();
  const int64_t ngene = (int64_t)gmt->gene_name.size();
  auto full_len = [&](int64_t k) { return gmt->set_ptr[k + 1] - gmt->set_ptr[k]; };
  // sets by decreasing size (stable), repeated names dropped, head(ntop)          R/gmt-utils.R:25-27
  std::vector<int64_t> order(ns);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return full_len(a) > full_len(b); });
  std::unordered_map<std::string_view, int> seen_names;
  seen_names.reserve((size_t)ns * 2);
  std::vector<int64_t> keep;
  for (int64_t k : order)
    if (seen_names.emplace(std::string_view(gmt->names[k]), 1).second) keep.push_back(k);
  const int64_t m = (int64_t)keep.size();
  auto set_len = [&](int64_t j) {
    const int64_t len = full_len(keep[j]);
    return (ntop > 0 && ntop < len) ? ntop : len;
  };
  // background: bg as given, or names(sort(table(unlist(gmt)), decreasing = TRUE))  (:30)
  std::vector<int32_t> pos(ngene, -1);   // gene id -> row of gg, -1 if absent
  std::vector<std::string> gg;
  if (nbg > 0) {
    int64_t lim = nbg;
    if (max_genes >= 0 && max_genes < lim) lim = max_genes;                       // :33-35
    gg.reserve(lim);
    for (int64_t k = 0; k < lim; ++k) gg.emplace_back(bg[k]);
    for (int64_t r = lim - 1; r >= 0; --r) {                                      // match(): first occurrence wins
      auto it = gmt->intern.find(std::string_view(gg[r]));
      if (it != gmt->intern.end()) pos[it->second] = (int32_t)r;
    }
  } else {
    std::vector<int64_t> gcount(ngene, 0);
    for (int64_t j = 0; j < m; ++j) {
      const int32_t* s = gmt->set_gene.data() + gmt->set_ptr[keep[j]];
      for (int64_t k = 0, len = set_len(j); k < len; ++k) ++gcount[s[k]];
    }
    std::vector<int32_t> by;
    by.reserve(ngene);
    for (int32_t k = 0; k < (int32_t)ngene; ++k)
      if (gcount[k] > 0) by.push_back(k);
    std::sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return strcmp(gmt->gene_name[a], gmt->gene_name[b]) < 0; });
    std::stable_sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return gcount[a] > gcount[b]; });
    if (max_genes >= 0 && max_genes < (int64_t)by.size()) by.resize(max_genes);   // :33-35
    gg.reserve(by.size());
    for (size_t r = 0; r < by.size(); ++r) {
      gg.emplace_back(gmt->gene_name[by[r]]);
      pos[by[r]] = (int32_t)r;
    }
  }
  const int64_t g = (int64_t)gg.size();
  // 0/1 entries                                                                   (:47-60)
  std::vector<int64_t> rowsum(g, 0);
  std::vector<int32_t> p(m + 1, 0);
  std::vector<int32_t> in_set(g, -1);   // row -> last set that had it (a repeated gene is one entry)
  int64_t z = 0;
  for (int64_t j = 0; j < m; ++j) {
    const int32_t* s = gmt->set_gene.data() + gmt->set_ptr[keep[j]];
    for (int64_t k = 0, len = set_len(j); k < len; ++k) {
      const int32_t r = pos[s[k]];
      if (r >= 0 && in_set[r] != (int32_t)j) { in_set[r] = (int32_t)j; ++rowsum[r]; ++z; }
    }
    if (z > INT32_MAX) {
      set_error("gmt2mat: more than 2^31-1 memberships do not fit the int32 slots of a dgCMatrix");
      return PLAIDHIP_EUNSUPPORTED;
    }
    p[j + 1] = (int32_t)z;
  }
  // rows by decreasing number of sets, stable                                     (:62)
  std::vector<int32_t> ro(g);
  std::iota(ro.begin(), ro.end(), 0);
  std::stable_sort(ro.begin(), ro.end(), [&](int32_t a, int32_t b) { return rowsum[a] > rowsum[b]; });
  std::vector<int32_t> newrow(g);
  for (int64_t k = 0; k < g; ++k) newrow[ro[k]] = (int32_t)k;
  plaidhip_gmtmat* M = new plaidhip_gmtmat();
  M->g = g;
  M->m = m;
  M->z = z;
  M->rownames.reserve(g);
  for (int64_t k = 0; k < g; ++k) M->rownames.push_back(gg[ro[k]]);
  M->colnames.reserve(m);
  for (int64_t j = 0; j < m; ++j) M->colnames.push_back(gmt->names[keep[j]]);
  M->p = std::move(p);
  M->i.resize(z);
  for (int64_t j = 0; j < m; ++j) {
    const int32_t* s = gmt->set_gene.data() + gmt->set_ptr[keep[j]];
    int32_t* dst = M->i.data() + M->p[j];
    int32_t c = 0;
    for (int64_t k = 0, len = set_len(j); k < len; ++k) {
      const int32_t r = pos[s[k]];
      if (r >= 0 && in_set[r] != (int32_t)(m + j)) { in_set[r] = (int32_t)(m + j); dst[c++] = newrow[r]; }
    }
    std::sort(dst, dst + c);
  }
  *out = M;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

int plaidhip_gmtmat_dims(const plaidhip_gmtmat* M, int64_t dims[3]) try {
  PH_REQUIRE(M && dims, "gmtmat_dims: null argument");
  dims[0] = M->g;
  dims[1] = M->m;
  dims[2] = M->z;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

const int32_t* plaidhip_gmtmat_p(const plaidhip_gmtmat* M) { return M ? M->p.data() : nullptr; }
const int32_t* plaidhip_gmtmat_i(const plaidhip_gmtmat* M) { return M ? M->i.data() : nullptr; }

// '\n'-joined dimnames: axis 0 = genes (rows), 1 = sets (columns)
const char* plaidhip_gmtmat_names(plaidhip_gmtmat* M, int axis, int64_t* nbytes) {
  if (!M || axis < 0 || axis > 1) return nullptr;
  join_names(axis == 0 ? M->rownames : M->colnames, M->joined[axis]);
  if (nbytes) *nbytes = (int64_t)M->joined[axis].size();
  return M->joined[axis].c_str();
}

int plaidhip_gmtmat_destroy(plaidhip_gmtmat* M) try {
  delete M;
  return PLAIDHIP_OK;
} catch (...) { return plaidhip::on_exception(); }

}  // extern "C"
