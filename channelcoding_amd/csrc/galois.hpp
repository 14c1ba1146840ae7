// galois.hpp -- host-side GF(2^q) arithmetic and cyclic-code construction for
// the MI355X decoder library (product code; the CPU never decodes frames, it
// only builds the per-code tables that are uploaded once at cc_code_create).
//
// Behavioural contract (what must equal the reference, file:line relative to
// the reference repository):
//   tables        src/math/galois.h:269-301   doubled exp/log, log[0] = 0
//   from_power    src/math/galois.h:182-184   exponent reduced mod 2^q (sic)
//   BCH generator src/codes/bch.h:28-78       lcm of minimal polynomials of alpha^1,3,..,2t-1
//   RS generator  src/codes/rs.h:18-39        prod (x - alpha^(mu+i*step)), i < 2t
//   h = (x^n+1)/g src/codes/cyclic.h:120-123,:272
//   dmin          src/codes/cyclic.h:186-204  consecutive_zeroes(g) + 1 (over-counts for RS: kept)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ccamd {

struct Field {
  unsigned q = 0, size = 0, n = 0;  // size = 2^q, n = size - 1 non-zero elements
  std::vector<uint8_t> exp, log;    // 2*size entries each
  explicit Field(unsigned q_);
  uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
  uint8_t div(uint8_t a, uint8_t b) const { return a ? exp[log[a] + n - log[b]] : 0; }
  uint8_t inv(uint8_t a) const { return div(1, a); }
  uint8_t alpha_pow(unsigned p) const { return exp[p % size]; }  // reference from_power
  uint8_t alpha_pow_true(unsigned p) const { return exp[p % n]; }  // mathematically reduced exponent
};

using Poly = std::vector<uint8_t>;  // index = power of x

int degree(const Poly &p);
Poly multiply(const Field &f, const Poly &a, const Poly &b);
void divide(const Field &f, const Poly &num, const Poly &den, Poly &quot, Poly &rem);
uint8_t evaluate(const Field &f, const Poly &p, uint8_t x);

struct CodeTables {
  int family = 0;
  unsigned q = 0, t = 0, n = 0, k = 0, l = 0, dmin = 0, mu = 1, step = 1;
  Poly g, h;
  std::vector<uint8_t> roots;         // 2t syndrome evaluation points
  std::vector<unsigned> root_powers;  // their exponents (log)
  std::vector<unsigned> row0_support;  // columns j with H[0][j] != 0 (h reversed), ascending
  std::vector<uint8_t> row0;           // length n: h reversed, zero padded
  bool binary_h = true;                // every h coefficient in {0,1}
};

// throws std::invalid_argument for parameters the reference itself would reject
CodeTables build_code(const Field &f, int family, unsigned t, unsigned mu, unsigned step);

}  // namespace ccamd
