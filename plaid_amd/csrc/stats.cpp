// Host-side tail of plaid.test() (R/plaid.R:392-537): from per-set sufficient statistics (O(sets)
// numbers reduced on the device) to t statistics, p-values, the combined p and the FDR.
// Distribution functions are plain double-precision implementations of the textbook forms R uses:
//   2*pt(|t|, df, lower=FALSE)  = I_{df/(df+t^2)}(df/2, 1/2)     regularised incomplete beta
//   pchisq(x, 2k, lower=FALSE)  = exp(-x/2) * sum_{i<k} (x/2)^i / i!
//   qnorm / pnorm upper tails   = Wichura's AS241 (PPND16) / erfc
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

#include "common.h"

namespace plaidhip {

namespace {

// continued fraction of the incomplete beta function (modified Lentz)
double betacf(double a, double b, double x) {
  const double tiny = 1e-300, eps = 1e-16;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (std::fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 10000; ++m) {
    const double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d;
    if (std::fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c;
    if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d;
    if (std::fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c;
    if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < eps) break;
  }
  return h;
}

// regularised incomplete beta I_x(a, b); y = 1 - x is passed in separately so that it keeps its
// digits when x rounds to 1
double betai(double a, double b, double x, double y) {
  if (!(x > 0.0)) return 0.0;
  if (!(y > 0.0)) return 1.0;
  // log(1 / B(a, b)); for b = 1/2 and large a the three lgamma values are ~a ln a each and their
  // difference loses digits, so use the Stirling series of ln Gamma(a + 1/2) - ln Gamma(a)
  double lnorm;
  if (b == 0.5 && a >= 100.0) {
    const double ia = 1.0 / a, ia2 = ia * ia;
    lnorm = 0.5 * std::log(a) - ia * (0.125 - ia2 * (1.0 / 192.0 - ia2 * (1.0 / 640.0))) - 0.5 * std::log(M_PI);
  } else {
    lnorm = std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b);
  }
  const double lbt = lnorm + a * std::log(x) + b * std::log(y);
  if (x < (a + 1.0) / (a + b + 2.0)) return std::exp(lbt) * betacf(a, b, x) / a;
  return 1.0 - std::exp(lbt) * betacf(b, a, y) / b;
}

}  // namespace

double chisq_upper_even(double x, int k);
double qnorm_lower(double p);
double pnorm_upper(double z);

// 2 * pt(|t|, df, lower.tail = FALSE)
double t_two_sided_p(double t, double df) {
  if (std::isnan(t) || std::isnan(df) || !(df > 0.0)) return std::numeric_limits<double>::quiet_NaN();
  if (std::isinf(t)) return 0.0;
  const double t2 = t * t;
  // I_x(df/2, 1/2) at x = df / (df + t^2): small p <=> small x, which betai evaluates directly (no 1 - ...)
  return betai(0.5 * df, 0.5, df / (df + t2), t2 / (df + t2));
}

// pchisq(x, 2k, lower.tail = FALSE), k a positive integer
double chisq_upper_even(double x, int k) {
  if (std::isnan(x)) return x;
  if (!(x > 0.0)) return 1.0;
  const double h = 0.5 * x;
  double term = 1.0, sum = 1.0;
  for (int i = 1; i < k; ++i) {
    term *= h / i;
    sum += term;
  }
  return std::exp(-h) * sum;
}

// qnorm(p): Wichura (1988) AS241, PPND16
double qnorm_lower(double p) {
  if (std::isnan(p) || p < 0.0 || p > 1.0) return std::numeric_limits<double>::quiet_NaN();
  if (p == 0.0) return -std::numeric_limits<double>::infinity();
  if (p == 1.0) return std::numeric_limits<double>::infinity();
  const double q = p - 0.5;
  if (std::fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    const double num = (((((((2509.0809287301226727 * r + 33430.575583588128105) * r + 67265.770927008700853) * r +
                            45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r +
                          133.14166789178437745) * r + 3.387132872796366608);
    const double den = (((((((5226.495278852545925 * r + 28729.085735721942674) * r + 39307.89580009271061) * r +
                            21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r +
                          42.313330701600911252) * r + 1.0);
    return q * num / den;
  }
  double r = q < 0.0 ? p : 1.0 - p;
  r = std::sqrt(-std::log(r));
  double val;
  if (r <= 5.0) {
    r -= 1.6;
    const double num = (((((((7.7454501427834140764e-4 * r + 0.0227238449892691845833) * r + 0.24178072517745061177) * r +
                            1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r +
                          4.6303378461565452959) * r + 1.42343711074968357734);
    const double den = (((((((1.05075007164441684324e-9 * r + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r +
                            0.14810397642748007459) * r + 0.68976733498510000455) * r + 1.6763848301838038494) * r +
                          2.05319162663775882187) * r + 1.0);
    val = num / den;
  } else {
    r -= 5.0;
    const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r +
                            0.026532189526576123093) * r + 0.29656057182850489123) * r + 1.7848265399172913358) * r +
                          5.4637849111641143699) * r + 6.6579046435011037772);
    const double den = (((((((2.04426310338993978564e-15 * r + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r +
                            7.868691311456132591e-4) * r + 0.0148753612908506148525) * r + 0.13692988092273580531) * r +
                          0.59983224619603520104) * r + 1.0);
    val = num / den;
    // polish in the far tail (two Newton steps on the tail probability, which erfc gives to full
    // relative accuracy): the rational approximation alone is good to ~1e-8 here
    const double tail = q < 0.0 ? p : 1.0 - p;   // P(Z > val)
    for (int it = 0; it < 2; ++it) {
      const double pt = 0.5 * std::erfc(val / std::sqrt(2.0));
      const double dens = std::exp(-0.5 * val * val) / std::sqrt(2.0 * M_PI);
      if (!(dens > 0.0)) break;
      val += (pt - tail) / dens;
    }
  }
  return q < 0.0 ? -val : val;
}

double pnorm_upper(double z) { return 0.5 * std::erfc(z / std::sqrt(2.0)); }

// stats::p.adjust(p, method = "fdr") (Benjamini-Hochberg); NaN stay NaN and do not count
void p_adjust_fdr(const double* p, int64_t m, double* q) {
  std::vector<int64_t> idx;
  idx.reserve(m);
  for (int64_t i = 0; i < m; ++i) {
    if (std::isnan(p[i])) q[i] = p[i];
    else idx.push_back(i);
  }
  const int64_t n = (int64_t)idx.size();
  std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return p[a] > p[b]; });   // decreasing
  double run = std::numeric_limits<double>::infinity();
  for (int64_t k = 0; k < n; ++k) {
    const int64_t i = idx[k];
    const double v = p[i] * (double)n / (double)(n - k);   // rank (n - k) in increasing order
    run = std::min(run, v);
    q[i] = std::min(1.0, run);
  }
}

// One set's tests.  k sets size, s1 = sum fc, s2 = sum fc^2 over the set; totals over all genes.
// R/plaid.R:476-486
double onesample_p(double k, double s1, double s2, double* mean_out) {
  const double meanx = s1 / (1e-8 + k);
  const double sdx = std::sqrt((s2 - meanx * meanx * k) / (k - 1.0));
  const double t = meanx / (1e-8 + sdx) * std::sqrt(k);
  *mean_out = meanx;
  return t_two_sided_p(std::fabs(t), std::max(k - 1.0, 1.0));
}

// R/plaid.R:488-520
double twosample_p(double g, double k, double s1, double s2, double tot1, double tot2, double* diff_out) {
  const double sum1 = k, sum0 = g - k;
  const double ssq1 = s2, ssq0 = tot2 - s2;
  const double mean1 = s1 / (1e-8 + sum1), mean0 = (tot1 - s1) / (1e-8 + sum0);
  const double var0 = (ssq0 - mean0 * mean0 * sum0) / (sum0 - 1.0);
  const double var1 = (ssq1 - mean1 * mean1 * sum1) / (sum1 - 1.0);
  const double varsum = var0 / sum0 + var1 / sum1;
  const double dof = varsum * varsum / (var0 / sum0 * (sum0 - 1.0) + var1 / sum1 * (sum1 - 1.0));
  const double f = mean1 - mean0;
  const double t = f / std::sqrt(varsum);
  *diff_out = f;
  return t_two_sided_p(std::fabs(t), std::max(dof, 1.0));
}

// Welch test from group means / sums of squared deviations (Rfast::ttests(x, ina), R/plaid.R:429)
double welch_p(double m0, double m1, double ssd0, double ssd1, double n0, double n1) {
  const double v0 = ssd0 / (n0 - 1.0), v1 = ssd1 / (n1 - 1.0);
  const double a = v0 / n0, b = v1 / n1;
  const double t = (m0 - m1) / std::sqrt(a + b);
  const double dof = (a + b) * (a + b) / (a * a / (n0 - 1.0) + b * b / (n1 - 1.0));
  return t_two_sided_p(std::fabs(t), dof);
}

// P1[is.na(P1)] <- 1; pmin(pmax(P1, 1e-99), 1 - 1e-99)   (R/plaid.R:441-446)
double clamp_p(double p) {
  if (std::isnan(p)) p = 1.0;
  return std::min(std::max(p, 1e-99), 1.0 - 1e-99);
}

// matrix_combine_p, R/plaid.R:522-537.  method 0 = fisher / sumlog, 1 = stouffer / sumz
double combine_p(const double* p, int np, int method) {
  if (method == 0) {
    double chisq = 0.0;
    for (int i = 0; i < np; ++i) chisq += std::log(p[i]);
    return chisq_upper_even(-2.0 * chisq, np);
  }
  double zz = 0.0;
  for (int i = 0; i < np; ++i) zz += -qnorm_lower(p[i]);   // qnorm(p, lower.tail = FALSE)
  return pnorm_upper(zz / std::sqrt((double)np));
}

}  // namespace plaidhip

// Test hook (not part of include/plaidhip.h): the distribution functions above, so that the CPU-side
// tests can check them against scipy without a device.  kind 0: 2*pt(|x|, a, lower=FALSE);
// 1: pchisq(x, 2*a, lower=FALSE); 2: qnorm(x); 3: pnorm(x, lower=FALSE)
extern "C" double plaidhip_debug_pvalue(int kind, double x, double a) {
  switch (kind) {
    case 0: return plaidhip::t_two_sided_p(x, a);
    case 1: return plaidhip::chisq_upper_even(x, (int)a);
    case 2: return plaidhip::qnorm_lower(x);
    case 3: return plaidhip::pnorm_upper(x);
    default: return std::numeric_limits<double>::quiet_NaN();
  }
}
