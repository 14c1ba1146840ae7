// galois.hpp -- host-side GF(2^q) arithmetic and cyclic-code construction for
// the MI355X decoder library (product code; the CPU never decodes frames, it
// only builds the per-code tables that are uploaded once at cc_code_create).
// Templates over the symbol type: uint8_t for q <= 8, uint16_t for q = 9..15 (the reference's storage_type,
// src/math/galois.h:44-53: "uint16_t allows galois fields up to 2^15 to be used").
//
// Behavioural contract (what must equal the reference, file:line relative to
// the reference repository):
//   tables        src/math/galois.h:269-301   doubled exp/log, log[0] = 0
//   modular poly  src/math/galois.h:18-20     defaults for q <= 8; beyond that the caller names one
//                                             (modular_polynomial<>, :23-25 / default_modular_polynomial :57-67)
//   from_power    src/math/galois.h:182-184   exponent reduced mod 2^q (sic)
//   BCH generator src/codes/bch.h:28-78       lcm of minimal polynomials of alpha^1,3,..,2t-1
//   RS generator  src/codes/rs.h:18-39        prod (x - alpha^(mu+i*step)), i < 2t
//   h = (x^n+1)/g src/codes/cyclic.h:120-123,:272
//   dmin          src/codes/cyclic.h:186-204  consecutive_zeroes(g) + 1 (over-counts for RS: kept)
#pragma once
#include <algorithm>
#include <cstdint>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace ccamd {

// default primitive polynomials, src/math/galois.h:18-20 (0 = none: q > 8 needs an explicit one)
inline unsigned default_modular_polynomial(unsigned q) {
  constexpr unsigned k[9] = {0, 0x3, 0x7, 0xb, 0x13, 0x25, 0x43, 0x83, 0x11d};
  return q <= 8 ? k[q] : 0u;
}

template <typename T>
struct FieldT {
  unsigned q = 0, size = 0, n = 0, poly = 0;  // size = 2^q, n = size - 1 non-zero elements
  std::vector<T> exp, log;                    // 2*size entries each
  explicit FieldT(unsigned q_, unsigned poly_ = 0)
      : q(q_), size(1u << q_), n((1u << q_) - 1), poly(poly_ ? poly_ : default_modular_polynomial(q_)),
        exp(2 * (1u << q_), 0), log(2 * (1u << q_), 0) {
    if (q < 2 || q > 8 * sizeof(T) - (sizeof(T) > 1 ? 1 : 0) || q > 15) throw std::invalid_argument("GF(2^q): q out of range for the symbol type");
    if (!poly) throw std::invalid_argument("GF(2^q), q > 8: a modular polynomial must be given");
    if ((poly >> q) != 1u) throw std::invalid_argument("modular polynomial must have degree q");
    unsigned v = 1;
    std::vector<char> seen(size, 0);
    for (unsigned p = 0; p < n; ++p) {
      if (seen[v]) throw std::invalid_argument("modular polynomial is not primitive");
      seen[v] = 1;
      log[v] = log[v + size] = static_cast<T>(p);
      exp[p] = exp[p + n] = static_cast<T>(v);
      v <<= 1;
      if (v & size) v ^= poly;
    }
    exp[n] = 1;
    exp[2 * n] = 1;  // exp[2*size-1] stays 0, as in the reference
  }
  T mul(T a, T b) const { return (a && b) ? exp[log[a] + log[b]] : T(0); }
  T div(T a, T b) const { return a ? exp[log[a] + n - log[b]] : T(0); }
  T inv(T a) const { return div(1, a); }
  T alpha_pow(unsigned p) const { return exp[p % size]; }  // reference from_power
  T alpha_pow_true(unsigned p) const { return exp[p % n]; }  // mathematically reduced exponent
};
using Field = FieldT<uint8_t>;

template <typename T>
using PolyT = std::vector<T>;  // index = power of x
using Poly = PolyT<uint8_t>;

template <typename T>
int degree(const PolyT<T> &p) {
  for (int i = static_cast<int>(p.size()) - 1; i >= 0; --i)
    if (p[static_cast<size_t>(i)]) return i;
  return -1;
}

template <typename T>
PolyT<T> multiply(const FieldT<T> &f, const PolyT<T> &a, const PolyT<T> &b) {
  const int da = degree(a), db = degree(b);
  if (da < 0 || db < 0) return PolyT<T>{0};
  PolyT<T> r(static_cast<size_t>(da + db + 1), 0);
  for (int i = 0; i <= da; ++i)
    if (a[i])
      for (int j = 0; j <= db; ++j) r[i + j] ^= f.mul(a[i], b[j]);
  return r;
}

template <typename T>
void divide(const FieldT<T> &f, const PolyT<T> &num, const PolyT<T> &den, PolyT<T> &quot, PolyT<T> &rem) {
  const int dn = degree(num), dd = degree(den);
  if (dd < 0) throw std::invalid_argument("polynomial division by zero");
  rem = num;
  if (dn < dd) {
    quot = PolyT<T>{0};
    return;
  }
  quot.assign(static_cast<size_t>(dn - dd + 1), 0);
  const T lead_inv = f.inv(den[dd]);
  for (int pos = dn; pos >= dd; --pos) {
    const T c = f.mul(rem[pos], lead_inv);
    if (!c) continue;
    quot[pos - dd] = c;
    for (int j = 0; j <= dd; ++j) rem[pos - dd + j] ^= f.mul(den[j], c);
  }
}

template <typename T>
T evaluate(const FieldT<T> &f, const PolyT<T> &p, T x) {
  if (p.empty() || x == 0) return 0;  // polynomial.h:274-275
  T acc = p.back();
  for (size_t i = p.size() - 1; i-- > 0;) acc = f.mul(acc, x) ^ p[i];
  return acc;
}

template <typename T>
struct CodeTablesT {
  int family = 0;
  unsigned q = 0, t = 0, n = 0, k = 0, l = 0, dmin = 0, mu = 1, step = 1;
  PolyT<T> g, h;
  std::vector<T> roots;               // 2t syndrome evaluation points
  std::vector<unsigned> root_powers;  // their exponents (log)
  std::vector<unsigned> row0_support;  // columns j with H[0][j] != 0 (h reversed), ascending
  std::vector<T> row0;                 // length n: h reversed, zero padded
  bool binary_h = true;                // every h coefficient in {0,1}
};
using CodeTables = CodeTablesT<uint8_t>;

// throws std::invalid_argument for parameters the reference itself would reject
template <typename T>
CodeTablesT<T> build_code(const FieldT<T> &f, int family, unsigned t, unsigned mu, unsigned step) {
  if (t < 1 || 2 * t >= f.n) throw std::invalid_argument("correction capability out of range");
  CodeTablesT<T> c;
  c.family = family;
  c.q = f.q;
  c.t = t;
  c.n = f.n;
  c.mu = mu;
  c.step = step;

  PolyT<T> g{1};
  if (family == 0) {
    // Union of the cyclotomic cosets of 1, 3, .., 2t-1; each coset contributes
    // one (binary) minimal polynomial, so the product equals the reference's lcm.
    std::set<unsigned> exponents;
    for (unsigned p = 1; p < 2 * t; p += 2) {
      unsigned e = p % f.n;
      do {
        exponents.insert(e);
        e = (2 * e) % f.n;
      } while (e != p % f.n);
    }
    for (unsigned e : exponents) g = multiply(f, g, PolyT<T>{f.alpha_pow(e), 1});
    for (unsigned i = 1; i <= 2 * t; ++i) c.roots.push_back(f.alpha_pow(i));
  } else if (family == 1) {
    for (unsigned i = 0; i < 2 * t; ++i) {
      const T r = f.alpha_pow(mu + i * step);
      g = multiply(f, g, PolyT<T>{r, 1});
      c.roots.push_back(r);
    }
  } else {
    throw std::invalid_argument("unknown code family");
  }
  for (T r : c.roots) {
    if (r == 0) throw std::invalid_argument("syndrome root is zero (exponent wraps mod 2^q in the reference)");
    c.root_powers.push_back(f.log[r]);
  }
  g.resize(static_cast<size_t>(degree(g) + 1));
  c.g = g;
  c.k = static_cast<unsigned>(degree(g));
  if (c.k >= c.n) throw std::invalid_argument("generator polynomial leaves no information symbols");
  c.l = c.n - c.k;

  PolyT<T> xn1(c.n + 1, 0), rem;
  xn1[0] = 1;
  xn1[c.n] = 1;
  divide(f, xn1, g, c.h, rem);
  if (degree(rem) >= 0) throw std::logic_error("g does not divide x^n + 1");
  c.h.resize(static_cast<size_t>(degree(c.h) + 1));

  // dmin = consecutive_zeroes(g) + 1, reproducing cyclic.h:186-204 literally.
  std::vector<unsigned> powers;
  for (unsigned v = 1; v < f.size; ++v)
    if (evaluate(f, g, static_cast<T>(v)) == 0) powers.push_back(f.log[v]);
  std::sort(powers.begin(), powers.end());
  auto first = std::find(powers.begin(), powers.end(), 1u);
  auto last = std::adjacent_find(first, powers.end(), [](unsigned a, unsigned b) { return a + 1 != b; });
  c.dmin = static_cast<unsigned>(last - first) + 1 + 1;
  if (c.dmin > c.n) throw std::invalid_argument("dmin > n");

  c.row0.assign(c.n, 0);
  for (size_t j = 0; j < c.h.size(); ++j) {
    const T v = c.h[c.h.size() - 1 - j];
    c.row0[j] = v;
    if (v) c.row0_support.push_back(static_cast<unsigned>(j));
    if (v > 1) c.binary_h = false;
  }
  return c;
}

}  // namespace ccamd
