// galois.cpp -- see galois.hpp.
#include "galois.hpp"

#include <algorithm>
#include <set>
#include <stdexcept>

namespace ccamd {

namespace {
// default primitive polynomials, src/math/galois.h:18-20
constexpr unsigned kModularPolynomial[9] = {0, 0x3, 0x7, 0xb, 0x13, 0x25, 0x43, 0x83, 0x11d};
}  // namespace

Field::Field(unsigned q_) : q(q_), size(1u << q_), n((1u << q_) - 1), exp(2 * (1u << q_), 0), log(2 * (1u << q_), 0) {
  if (q < 2 || q > 8) throw std::invalid_argument("GF(2^q): q must be in 2..8");
  unsigned v = 1;
  for (unsigned p = 0; p < n; ++p) {
    log[v] = log[v + size] = static_cast<uint8_t>(p);
    exp[p] = exp[p + n] = static_cast<uint8_t>(v);
    v <<= 1;
    if (v & size) v ^= kModularPolynomial[q];
  }
  exp[n] = 1;
  exp[2 * n] = 1;  // exp[2*size-1] stays 0, as in the reference
}

int degree(const Poly &p) {
  for (int i = static_cast<int>(p.size()) - 1; i >= 0; --i)
    if (p[static_cast<size_t>(i)]) return i;
  return -1;
}

Poly multiply(const Field &f, const Poly &a, const Poly &b) {
  const int da = degree(a), db = degree(b);
  if (da < 0 || db < 0) return Poly{0};
  Poly r(static_cast<size_t>(da + db + 1), 0);
  for (int i = 0; i <= da; ++i)
    if (a[i])
      for (int j = 0; j <= db; ++j) r[i + j] ^= f.mul(a[i], b[j]);
  return r;
}

void divide(const Field &f, const Poly &num, const Poly &den, Poly &quot, Poly &rem) {
  const int dn = degree(num), dd = degree(den);
  if (dd < 0) throw std::invalid_argument("polynomial division by zero");
  rem = num;
  if (dn < dd) {
    quot = Poly{0};
    return;
  }
  quot.assign(static_cast<size_t>(dn - dd + 1), 0);
  const uint8_t lead_inv = f.inv(den[dd]);
  for (int pos = dn; pos >= dd; --pos) {
    const uint8_t c = f.mul(rem[pos], lead_inv);
    if (!c) continue;
    quot[pos - dd] = c;
    for (int j = 0; j <= dd; ++j) rem[pos - dd + j] ^= f.mul(den[j], c);
  }
}

uint8_t evaluate(const Field &f, const Poly &p, uint8_t x) {
  if (p.empty() || x == 0) return 0;  // polynomial.h:274-275
  uint8_t acc = p.back();
  for (size_t i = p.size() - 1; i-- > 0;) acc = f.mul(acc, x) ^ p[i];
  return acc;
}

CodeTables build_code(const Field &f, int family, unsigned t, unsigned mu, unsigned step) {
  if (t < 1 || 2 * t >= f.n) throw std::invalid_argument("correction capability out of range");
  CodeTables c;
  c.family = family;
  c.q = f.q;
  c.t = t;
  c.n = f.n;
  c.mu = mu;
  c.step = step;

  Poly g{1};
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
    for (unsigned e : exponents) g = multiply(f, g, Poly{f.alpha_pow(e), 1});
    for (unsigned i = 1; i <= 2 * t; ++i) c.roots.push_back(f.alpha_pow(i));
  } else if (family == 1) {
    for (unsigned i = 0; i < 2 * t; ++i) {
      const uint8_t r = f.alpha_pow(mu + i * step);
      g = multiply(f, g, Poly{r, 1});
      c.roots.push_back(r);
    }
  } else {
    throw std::invalid_argument("unknown code family");
  }
  for (uint8_t r : c.roots) {
    if (r == 0) throw std::invalid_argument("syndrome root is zero (exponent wraps mod 2^q in the reference)");
    c.root_powers.push_back(f.log[r]);
  }
  g.resize(static_cast<size_t>(degree(g) + 1));
  c.g = g;
  c.k = static_cast<unsigned>(degree(g));
  if (c.k >= c.n) throw std::invalid_argument("generator polynomial leaves no information symbols");
  c.l = c.n - c.k;

  Poly xn1(c.n + 1, 0), rem;
  xn1[0] = 1;
  xn1[c.n] = 1;
  divide(f, xn1, g, c.h, rem);
  if (degree(rem) >= 0) throw std::logic_error("g does not divide x^n + 1");
  c.h.resize(static_cast<size_t>(degree(c.h) + 1));

  // dmin = consecutive_zeroes(g) + 1, reproducing cyclic.h:186-204 literally.
  std::vector<unsigned> powers;
  for (unsigned v = 1; v < f.size; ++v)
    if (evaluate(f, g, static_cast<uint8_t>(v)) == 0) powers.push_back(f.log[v]);
  std::sort(powers.begin(), powers.end());
  auto first = std::find(powers.begin(), powers.end(), 1u);
  auto last = std::adjacent_find(first, powers.end(), [](unsigned a, unsigned b) { return a + 1 != b; });
  c.dmin = static_cast<unsigned>(last - first) + 1 + 1;
  if (c.dmin > c.n) throw std::invalid_argument("dmin > n");

  c.row0.assign(c.n, 0);
  for (size_t j = 0; j < c.h.size(); ++j) {
    const uint8_t v = c.h[c.h.size() - 1 - j];
    c.row0[j] = v;
    if (v) c.row0_support.push_back(static_cast<unsigned>(j));
    if (v > 1) c.binary_h = false;
  }
  return c;
}

}  // namespace ccamd
