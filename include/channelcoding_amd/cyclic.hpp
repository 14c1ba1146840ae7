// channelcoding_amd/cyclic.hpp -- header-only C++14 facade that re-creates the template API of
// hannesweisbach/channelcoding for the decode path, on top of the C ABI (channelcoding_amd.h):
//
//   errors<>, dmin<>, decoding_failure                         src/codes/codes.h:7-36
//   cyclic::{peterson_gorenstein_zierler,berlekamp_massey,euklid}_tag   src/codes/hard_decision.h:15-24
//   min_sum_tag<It> ... normalized_2d_min_sum_tag<It,Alpha,Beta>         src/codes/soft_decision.h:20-73
//   cyclic::multiplication_tag / division_tag                  src/codes/cyclic.h:19-23
//   cyclic::primitive_bch<q, Capability, Sigma, N, Coding>     src/codes/bch.h:16-19
//   cyclic::rs<q, Capability, Sigma, N, Coding, mu, step>      src/codes/rs.h:6-10
//   members encode / decode / correct / H / to_string / rate / n / t    src/codes/cyclic.h:94-95,:111,:282-359
//   math::ef_element<2, q> (value type of RS symbols)          src/math/galois.h:89-267
//
// A program written against the reference compiles against this header by changing its includes
// (see INTEGRATION.md); every decode runs on the GPU.  New members: the *_batch forms and the stop
// rule (cc_stop_rule, default CC_STOP_PARITY) as a constructor argument.
#pragma once

#include <algorithm>
#include <cstdint>
#include <iterator>
#include <memory>
#include <ratio>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../channelcoding_amd.h"

// ---- codes.h ----
template <unsigned e> struct errors {
  static constexpr unsigned value = e;
};
template <unsigned d> struct dmin {
  static constexpr unsigned value = d;
};
template <typename T> struct correction_capability;
template <unsigned v> struct correction_capability<dmin<v>> {
  static constexpr unsigned value = (v - 1) / 2;
};
template <unsigned v> struct correction_capability<errors<v>> {
  static constexpr unsigned value = v;
};

class decoding_failure : public std::runtime_error {
public:
  using std::runtime_error::runtime_error;
};

struct algorithm_tag {};
struct hard_decision_tag : algorithm_tag {};
struct soft_decision_tag : algorithm_tag {};

// ---- soft_decision.h:20-73 (alpha / beta exactly as the reference's tags compute them) ----
template <unsigned Iterations = 50> struct min_sum_tag : soft_decision_tag {
  static constexpr unsigned iterations = Iterations;
  static constexpr int cc_alg = CC_ALG_MS;
  static constexpr double alpha = 1.0, beta = 0.0;
  static std::string to_string() { return "MS"; }
};
template <unsigned Iterations, typename T = std::ratio<1>> struct normalized_min_sum_tag : soft_decision_tag {
  static constexpr unsigned iterations = Iterations;
  static constexpr int cc_alg = CC_ALG_NMS;
  static constexpr double alpha = static_cast<double>(T::num) / T::den, beta = 0.0;
  static std::string to_string() { return "NMS"; }
};
template <unsigned Iterations = 50, typename T = std::ratio<0>> struct offset_min_sum_tag : soft_decision_tag {
  static constexpr unsigned iterations = Iterations;
  static constexpr int cc_alg = CC_ALG_OMS;
  static constexpr double alpha = 1.0, beta = static_cast<double>(T::num) / T::den;
  static std::string to_string() { return "OMS"; }
};
template <unsigned Iterations = 50> struct self_correcting_1_min_sum_tag : soft_decision_tag {
  static constexpr unsigned iterations = Iterations;
  static constexpr int cc_alg = CC_ALG_SCMS1;
  static constexpr double alpha = 1.0, beta = 0.0;
  static std::string to_string() { return "SCMS1"; }
};
template <unsigned Iterations = 50> struct self_correcting_2_min_sum_tag : soft_decision_tag {
  static constexpr unsigned iterations = Iterations;
  static constexpr int cc_alg = CC_ALG_SCMS2;
  static constexpr double alpha = 1.0, beta = 0.0;
  static std::string to_string() { return "SCMS2"; }
};
template <unsigned Iterations = 50, typename Alpha = std::ratio<1>, typename Beta = std::ratio<1, 10>>
struct normalized_2d_min_sum_tag : soft_decision_tag {
  static constexpr unsigned iterations = Iterations;
  static constexpr int cc_alg = CC_ALG_2DNMS;
  static constexpr double alpha = static_cast<double>(Alpha::num) / Alpha::den;
  static constexpr double beta = static_cast<double>(Beta::num) / Alpha::den;  // sic: soft_decision.h:71
  static std::string to_string() { return "2DNMS"; }
};

// ---- matrix.h (storage of H only; end() const is the repaired one) ----
template <typename T> class matrix {
  std::vector<std::vector<T>> data;
  size_t cols = 0;

public:
  matrix() = default;
  matrix(size_t rows, size_t cols_) : data(rows, std::vector<T>(cols_, T())), cols(cols_) {}
  std::vector<T> &at(size_t i) { return data.at(i); }
  const std::vector<T> &at(size_t i) const { return data.at(i); }
  size_t rows() const { return data.size(); }
  size_t columns() const { return cols; }
  void push_back(const std::vector<T> &row) {  // matrix.h push_back: rows must match the matrix width
    if (row.size() != cols) throw std::runtime_error("Row size does not match matrix size");
    data.push_back(row);
  }
  typename std::vector<std::vector<T>>::const_iterator begin() const { return data.begin(); }
  typename std::vector<std::vector<T>>::const_iterator end() const { return data.end(); }
};

namespace math {
// math::modular_polynomial<> (galois.h:23-25) and the default per field size (galois.h:57-67): defaults exist for
// q <= 8; a field GF(2^q) with q = 9 .. 15 has to be named by the caller, exactly as in the reference --
//     namespace math { namespace detail {
//     template <> struct default_modular_polynomial<10> { using type = ::math::modular_polynomial<0x409>; };
//     } }
// before the first use of cyclic::primitive_bch<10, ...> / cyclic::rs<10, ...> / math::ef_element<2, 10>.
template <uint16_t poly> struct modular_polynomial {
  static constexpr uint16_t value = poly;
};
namespace detail {
template <unsigned q> struct default_modular_polynomial {
  static_assert(q > 0, "GF(2^0) does not make sense. Choose q > 0.");
  static_assert(q < 9, "modular polynomials for GF(2^q), q > 8 have to be specified manually: specialise "
                       "math::detail::default_modular_polynomial<q>");
  static constexpr uint16_t table[9] = {0, 0x3, 0x7, 0xb, 0x13, 0x25, 0x43, 0x83, 0x11d};  // galois.h:18-20
  using type = ::math::modular_polynomial<table[q < 9 ? q : 0]>;
};
template <unsigned q> constexpr uint16_t default_modular_polynomial<q>::table[9];
}  // namespace detail

// GF(2^q) element as a value type (galois.h:89-267): enough for symbol I/O and printing.  storage_type is uint8_t
// for q <= 8 and uint16_t beyond (galois.h:44-53).
template <long prime, long power, typename Modular_Polynomial = typename detail::default_modular_polynomial<power>::type>
class ef_element;
template <long Power, typename Mp> class ef_element<2, Power, Mp> {
  static_assert(Power >= 1 && Power <= 15, "q <= 15");
  static constexpr unsigned size = 1u << Power;

public:
  using storage_type = typename std::conditional<(Power > 8), uint16_t, uint8_t>::type;
  static constexpr unsigned mod_polynomial = Mp::value;

private:
  struct tables {
    std::vector<storage_type> exp, log;
    tables() : exp(2 * size, 0), log(2 * size, 0) {
      unsigned v = 1;
      for (unsigned p = 0; p + 1 < size; ++p) {
        log[v] = log[v + size] = static_cast<storage_type>(p);
        exp[p] = exp[p + size - 1] = static_cast<storage_type>(v);
        v <<= 1;
        if (v & size) v ^= mod_polynomial;
      }
      exp[size - 1] = exp[2 * size - 2] = 1;
    }
  };
  static const tables &tab() {
    static const tables t;
    return t;
  }
  storage_type value = 0;

public:
  static constexpr size_t digits = Power;
  ef_element() = default;
  explicit ef_element(const storage_type &v) : value(v) {
    if (value & ~(size - 1)) throw std::runtime_error("Value is not an element of the field.");
  }
  static ef_element from_power(unsigned p) { return ef_element(tab().exp[p % size]); }  // galois.h:182-184
  unsigned power() const { return tab().log[value]; }
  ef_element operator+(const ef_element &r) const { return ef_element(static_cast<storage_type>(value ^ r.value)); }
  ef_element operator*(const ef_element &r) const {
    if (!value || !r.value) return ef_element(0);
    return ef_element(tab().exp[power() + r.power()]);
  }
  bool operator==(const ef_element &r) const { return value == r.value; }
  bool operator!=(const ef_element &r) const { return value != r.value; }
  explicit operator bool() const { return value != 0; }
  explicit operator storage_type() const { return value; }
  explicit operator unsigned() const { return value; }
  explicit operator int() const { return value; }
  friend std::ostream &operator<<(std::ostream &os, const ef_element &e) {
    if (e.value == 0) return os << 0;
    return os << "\xce\xb1^" << e.power();
  }
};
}  // namespace math

namespace cyclic {

struct coding_tag {};
struct multiplication_tag : coding_tag {
  static constexpr int cc_coding = CC_CODING_MULTIPLICATION;
};
struct division_tag : coding_tag {
  static constexpr int cc_coding = CC_CODING_DIVISION;
};

struct peterson_gorenstein_zierler_tag : hard_decision_tag {
  static constexpr int cc_alg = CC_ALG_PGZ;
  static constexpr unsigned iterations = 0;
  static constexpr double alpha = 1.0, beta = 0.0;
  static std::string to_string() { return "PGZ"; }
};
struct berlekamp_massey_tag : hard_decision_tag {
  static constexpr int cc_alg = CC_ALG_BM;
  static constexpr unsigned iterations = 0;
  static constexpr double alpha = 1.0, beta = 0.0;
  static std::string to_string() { return "BM"; }
};
struct euklid_tag : hard_decision_tag {
  static constexpr int cc_alg = CC_ALG_EUKLID;
  static constexpr unsigned iterations = 0;
  static constexpr double alpha = 1.0, beta = 0.0;
  static std::string to_string() { return "EUKLID"; }
};

namespace detail {
inline void check(int rc, const char *where) {
  if (rc == CC_OK) return;
  std::ostringstream os;
  os << where << ": " << cc_status_string(rc);
  const char *d = cc_last_error();
  if (d && *d) os << " (" << d << ")";
  throw std::runtime_error(os.str());
}
struct code_deleter {
  void operator()(cc_code *c) const { cc_code_destroy(c); }
};
template <typename T> struct to_byte {
  static uint8_t get(const T &v) { return static_cast<uint8_t>(v); }
};
template <long P, typename Mp> struct to_byte<math::ef_element<2, P, Mp>> {
  static uint8_t get(const math::ef_element<2, P, Mp> &v) { return static_cast<uint8_t>(static_cast<unsigned>(v)); }
};
template <typename S, typename T> struct to_symbol {
  static S get(const T &v) { return static_cast<S>(v); }
};
template <typename S, long P, typename Mp> struct to_symbol<S, math::ef_element<2, P, Mp>> {
  static S get(const math::ef_element<2, P, Mp> &v) { return static_cast<S>(static_cast<unsigned>(v)); }
};
// symbol type and single-call entry points by field size: bytes for q <= 8, 16 bits beyond (the _u16 entry points)
template <bool Wide> struct symbol_io;
template <> struct symbol_io<false> {
  using symbol = uint8_t;
  static int encode(const cc_code *c, const symbol *m, symbol *w) { return cc_encode_batch(c, m, w, 1); }
  static int extract(const cc_code *c, const symbol *w, symbol *m) { return cc_extract_batch(c, w, m, 1); }
  static int correct(const cc_code *c, const symbol *in, const uint16_t *er, const uint32_t *off, symbol *out, int32_t *nerr,
                     int32_t *st) {
    return cc_correct_hard_batch(c, in, er, off, out, nerr, st, 1);
  }
};
template <> struct symbol_io<true> {
  using symbol = uint16_t;
  static int encode(const cc_code *c, const symbol *m, symbol *w) { return cc_encode_batch_u16(c, m, w, 1); }
  static int extract(const cc_code *c, const symbol *w, symbol *m) { return cc_extract_batch_u16(c, w, m, 1); }
  static int correct(const cc_code *c, const symbol *in, const uint16_t *er, const uint32_t *off, symbol *out, int32_t *nerr,
                     int32_t *st) {
    return cc_correct_hard_batch_u16(c, in, er, off, out, nerr, st, 1);
  }
};
inline const char *failure_text(int st) {
  switch (st) {
    case CC_FRAME_NOT_CONVERGED: return "Decoding failure";
    case CC_FRAME_LOCATOR: return "\xce\xa3(x) does not have as many distinct zeroes as its degree";
    case CC_FRAME_RECHECK: return "Corrected word is not a codeword";
    case CC_FRAME_ERASURES: return "Number of erasures exceed error correction capability.";
    default: return "Decoding failure";
  }
}
}  // namespace detail

// Result of a batch decode: one row per frame.
struct batch_result {
  std::vector<uint8_t> words;   // B * n corrected symbols (hard-decided input for failed frames)
  std::vector<int32_t> status;  // CC_FRAME_*
  std::vector<int32_t> nerr;    // hard algorithms: corrected symbols or -1
  std::vector<uint16_t> iters;  // soft algorithms: index of the returning iteration
  std::vector<float> L;         // soft algorithms, when requested
  std::vector<uint8_t> msg;     // decode_batch: B * l message symbols
};

template <int Family, unsigned q, typename Capability, typename Algorithm, unsigned N, typename Coding, unsigned mu,
          unsigned step>
class code_base {
  static_assert(std::is_base_of<coding_tag, Coding>::value, "Coding must be division_tag or multiplication_tag");
  static_assert(std::is_base_of<algorithm_tag, Algorithm>::value, "Algorithm must be an algorithm tag");
  static_assert(N == (1u << q) - 1, "shortened codes are not supported by the device path");

public:
  using Element = math::ef_element<2, q>;  // (q > 8: through default_modular_polynomial<q>, see namespace math)
  static constexpr bool wide = q > 8;      // 16-bit symbols, the _u16 entry points
  using io = detail::symbol_io<wide>;
  using symbol = typename io::symbol;
  static constexpr unsigned n = N;
  static constexpr unsigned t = correction_capability<Capability>::value;
  static constexpr bool soft = std::is_base_of<soft_decision_tag, Algorithm>::value;

protected:
  std::shared_ptr<cc_code> handle;  // codes are copyable values, as in the reference
  unsigned k = 0, l = 0, dmin_ = 0;

public:
  double rate = 0;  // cyclic.h:111

  explicit code_base(cc_stop_rule stop = CC_STOP_PARITY, int device = CC_DEVICE_CURRENT) {
    cc_desc d;
    cc_desc_init(&d);
    d.family = Family;
    d.q = q;
    d.t = t;
    d.mu = mu;
    d.step = step;
    d.coding = Coding::cc_coding;
    d.algorithm = Algorithm::cc_alg;
    d.iterations = Algorithm::iterations;
    d.alpha = Algorithm::alpha;
    d.beta = Algorithm::beta;
    d.stop_rule = stop;
    d.device = device;
    if (wide) d.modular_polynomial = Element::mod_polynomial;
    cc_code *c = nullptr;
    detail::check(cc_code_create(&d, &c), "cc_code_create");
    handle.reset(c, detail::code_deleter());
    k = cc_k(c);
    l = cc_l(c);
    dmin_ = cc_dmin(c);
    rate = cc_rate(c);
  }

  const cc_code *c_handle() const { return handle.get(); }
  unsigned parity_symbols() const { return k; }
  unsigned information_symbols() const { return l; }

  std::string to_string() const {  // cyclic.h:282-287
    char buf[96];
    detail::check(cc_to_string(handle.get(), buf, sizeof buf), "cc_to_string");
    return buf;
  }

  template <typename T> matrix<T> H() const {  // cyclic.h:346-359
    std::vector<uint8_t> flat(static_cast<size_t>(k) * n);
    detail::check(cc_get_H(handle.get(), flat.data()), "cc_get_H");
    matrix<T> m(k, n);
    for (unsigned i = 0; i < k; ++i)
      for (unsigned j = 0; j < n; ++j) m.at(i).at(j) = T(flat[i * n + j]);
    return m;
  }

  template <typename T> matrix<T> H_alt() const {  // cyclic.h:361-385 (t*q rows: binary image of alpha^(col*(2 row+1)))
    std::vector<uint8_t> flat(static_cast<size_t>(t) * q * n);
    uint32_t rows = 0;
    detail::check(cc_get_H_alt(handle.get(), flat.data(), &rows), "cc_get_H_alt");
    matrix<T> m(rows, n);
    for (unsigned i = 0; i < rows; ++i)
      for (unsigned j = 0; j < n; ++j) m.at(i).at(j) = T(flat[i * n + j]);
    return m;
  }

  // ---- encode, cyclic.h:289-311 ----
  template <typename InputSequence, typename OutputIterator> void encode(const InputSequence &a, OutputIterator &&out) const {
    if (a.size() != l) {
      std::ostringstream os;
      os << "Source code word has wrong length (" << a.size() << "). Expected " << l;
      throw std::runtime_error(os.str());
    }
    std::vector<symbol> msg(l), cw(n);
    std::transform(a.begin(), a.end(), msg.begin(), [](const typename InputSequence::value_type &e) {
      return detail::to_symbol<symbol, typename InputSequence::value_type>::get(e);
    });
    const int rc = io::encode(handle.get(), msg.data(), cw.data());
    if (rc == CC_ERR_NOT_IN_FIELD) throw std::runtime_error("Value is not an element of the field.");
    detail::check(rc, "cc_encode_batch");
    for (symbol v : cw) *out++ = typename InputSequence::value_type(v);
  }

  // ---- correct, cyclic.h:331-344 ----
  template <typename Return_type = symbol, typename InputSequence>
  std::vector<Return_type> correct(const InputSequence &b, const std::vector<unsigned> &erasures = std::vector<unsigned>()) const {
    const std::vector<symbol> w = correct_symbols(b, erasures);
    std::vector<Return_type> r;
    r.reserve(n);
    for (symbol v : w) r.push_back(Return_type(v));
    return r;
  }

  // ---- decode, cyclic.h:313-327 ----
  template <typename InputSequence, typename Return_type = typename InputSequence::value_type>
  std::vector<Return_type> decode(const InputSequence &b, const std::vector<unsigned> &erasures = std::vector<unsigned>()) const {
    const std::vector<symbol> w = correct_symbols(b, erasures);
    std::vector<symbol> msg(l);
    detail::check(io::extract(handle.get(), w.data(), msg.data()), "cc_extract_batch");
    std::vector<Return_type> r;
    r.reserve(l);
    for (symbol v : msg) r.push_back(Return_type(v));
    return r;
  }

  // ---- batch forms (new): B frames of n symbols / soft values, frame-contiguous ----
  // (q <= 8 only: a q > 8 code batches through cc_correct_hard_batch_u16 / cc_encode_batch_u16 on handle.get())
  batch_result correct_batch(const uint8_t *symbols, size_t B) const {
    batch_result r;
    r.words.resize(B * n);
    r.status.resize(B);
    r.nerr.resize(B);
    detail::check(cc_correct_hard_batch(handle.get(), symbols, nullptr, nullptr, r.words.data(), r.nerr.data(), r.status.data(), B),
                  "cc_correct_hard_batch");
    return r;
  }
  batch_result correct_batch(const float *values, size_t B, bool want_L = false) const {
    batch_result r;
    r.words.resize(B * n);
    r.status.resize(B);
    if (soft) {
      r.iters.resize(B);
      if (want_L) r.L.resize(B * n);
      detail::check(cc_correct_soft_batch(handle.get(), values, nullptr, nullptr, r.words.data(), want_L ? r.L.data() : nullptr,
                                          r.iters.data(), r.status.data(), B),
                    "cc_correct_soft_batch");
    } else {
      r.nerr.resize(B);
      detail::check(cc_correct_hard_f32_batch(handle.get(), values, nullptr, nullptr, r.words.data(), r.nerr.data(),
                                              r.status.data(), B),
                    "cc_correct_hard_f32_batch");
    }
    return r;
  }

  // decode = correct + message extraction for B frames (cc_decode_hard_batch / cc_decode_soft_batch)
  batch_result decode_batch(const uint8_t *symbols, size_t B) const {
    batch_result r;
    r.words.resize(B * n);
    r.msg.resize(B * l);
    r.status.resize(B);
    r.nerr.resize(B);
    detail::check(cc_decode_hard_batch(handle.get(), symbols, nullptr, nullptr, r.msg.data(), r.words.data(), r.nerr.data(),
                                       r.status.data(), B),
                  "cc_decode_hard_batch");
    return r;
  }
  batch_result decode_batch(const float *values, size_t B) const {
    batch_result r;
    r.words.resize(B * n);
    r.msg.resize(B * l);
    r.status.resize(B);
    if (soft) r.iters.resize(B);
    detail::check(cc_decode_soft_batch(handle.get(), values, nullptr, nullptr, r.msg.data(), r.words.data(),
                                       soft ? r.iters.data() : nullptr, r.status.data(), B),
                  "cc_decode_soft_batch");
    return r;
  }

private:
  template <typename InputSequence>
  std::vector<symbol> correct_symbols(const InputSequence &b, const std::vector<unsigned> &erasures) const {
    using V = typename InputSequence::value_type;
    if (b.size() != n) {  // cyclic.h:213-218
      std::ostringstream os;
      os << "Channel code word has the wrong size (" << b.size() << "). Expected " << n;
      throw std::runtime_error(os.str());
    }
    std::vector<uint16_t> er(erasures.begin(), erasures.end());
    for (unsigned e : erasures)
      if (e >= n) throw std::out_of_range("erasure position");  // copy.at(erasure)
    const uint32_t off[2] = {0, static_cast<uint32_t>(er.size())};
    const uint16_t *erp = er.empty() ? nullptr : er.data();
    const uint32_t *offp = er.empty() ? nullptr : off;
    std::vector<symbol> out(n);
    int32_t status = 0, nerr = 0;
    int rc;
    if (std::is_signed<V>::value) {  // signed value_type: soft value, bit = (x < 0)  (cyclic.h:163-173,:220-222)
      rc = correct_signed(b, erp, offp, out, &nerr, &status, std::integral_constant<bool, wide>());
    } else {
      if (soft) throw std::runtime_error("min-sum needs a signed (soft) input sequence");
      std::vector<symbol> sym(n);
      std::transform(b.begin(), b.end(), sym.begin(), [](const V &v) { return detail::to_symbol<symbol, V>::get(v); });
      rc = io::correct(handle.get(), sym.data(), erp, offp, out.data(), &nerr, &status);
      if (rc == CC_ERR_NOT_IN_FIELD) throw std::runtime_error("Value is not an element of the field.");
    }
    detail::check(rc, "correct");
    if (status != CC_FRAME_OK) throw decoding_failure(detail::failure_text(status));
    return out;
  }
  template <typename InputSequence>
  int correct_signed(const InputSequence &b, const uint16_t *erp, const uint32_t *offp, std::vector<symbol> &out, int32_t *nerr,
                     int32_t *status, std::false_type) const {
    using V = typename InputSequence::value_type;
    std::vector<float> y(n);
    std::transform(b.begin(), b.end(), y.begin(), [](const V &v) { return as_float(v); });
    if (soft) return cc_correct_soft_batch(handle.get(), y.data(), erp, offp, out.data(), nullptr, nullptr, status, 1);
    return cc_correct_hard_f32_batch(handle.get(), y.data(), erp, offp, out.data(), nerr, status, 1);
  }
  template <typename InputSequence>
  int correct_signed(const InputSequence &b, const uint16_t *erp, const uint32_t *offp, std::vector<symbol> &out, int32_t *nerr,
                     int32_t *status, std::true_type) const {  // q > 8: the sign bits are the word (all in GF(2) of GF(2^q))
    using V = typename InputSequence::value_type;
    std::vector<symbol> sym(n);
    std::transform(b.begin(), b.end(), sym.begin(), [](const V &v) { return static_cast<symbol>(as_float(v) < 0.0f); });
    return io::correct(handle.get(), sym.data(), erp, offp, out.data(), nerr, status);
  }
  template <typename V> static typename std::enable_if<std::is_arithmetic<V>::value, float>::type as_float(const V &v) {
    return static_cast<float>(v);
  }
  template <typename V> static typename std::enable_if<!std::is_arithmetic<V>::value, float>::type as_float(const V &) {
    return 0.0f;
  }
};

template <unsigned q, typename Capability, typename Sigma = peterson_gorenstein_zierler_tag, unsigned N = (1u << q) - 1,
          typename Coding = division_tag>
class primitive_bch : public code_base<CC_FAMILY_BCH, q, Capability, Sigma, N, Coding, 1, 1> {
  using Base = code_base<CC_FAMILY_BCH, q, Capability, Sigma, N, Coding, 1, 1>;

public:
  using Base::Base;
  primitive_bch() : Base() {}
};

template <unsigned q, typename Capability, typename Sigma = peterson_gorenstein_zierler_tag, unsigned N = (1u << q) - 1,
          typename Coding = division_tag, unsigned mu = 1, unsigned step = 1>
class rs : public code_base<CC_FAMILY_RS, q, Capability, Sigma, N, Coding, mu, step> {
  using Base = code_base<CC_FAMILY_RS, q, Capability, Sigma, N, Coding, mu, step>;

public:
  using Base::Base;
  rs() : Base() {}
};

}  // namespace cyclic

// ---- soft_decision.h:220-295: the free functions min_sum<R, U>(H, y, tag) on any parity-check matrix ----
// Returns (hard decision b, a-posteriori L, index of the accepting iteration) and throws decoding_failure when no
// iteration satisfies the stop rule (soft_decision.h:199-201).  One launch per call: for throughput keep the
// handle (cc_minsum_create) and feed cc_correct_soft_batch(_dev) with many frames.
template <typename R, typename U = unsigned, typename Q, typename Tag>
typename std::enable_if<std::is_base_of<soft_decision_tag, Tag>::value,
                        std::tuple<std::vector<U>, std::vector<R>, unsigned>>::type
min_sum(const matrix<U> &H, const std::vector<Q> &y, Tag, cc_stop_rule stop = CC_STOP_PARITY) {
  const size_t rows = H.rows(), cols = H.columns();
  if (y.size() != cols) throw std::runtime_error("min_sum: y does not match the matrix width");
  std::vector<uint8_t> flat(rows * cols);
  for (size_t i = 0; i < rows; ++i)
    for (size_t j = 0; j < cols; ++j) flat[i * cols + j] = cyclic::detail::to_byte<U>::get(H.at(i).at(j));
  cc_desc d;
  cc_desc_init(&d);
  d.algorithm = Tag::cc_alg;
  d.iterations = Tag::iterations;
  d.alpha = Tag::alpha;
  d.beta = Tag::beta;
  d.stop_rule = stop;
  cc_code *c = nullptr;
  cyclic::detail::check(cc_minsum_create(&d, flat.data(), static_cast<uint32_t>(rows), static_cast<uint32_t>(cols), &c),
                        "cc_minsum_create");
  std::unique_ptr<cc_code, cyclic::detail::code_deleter> guard(c);
  std::vector<float> yf(y.begin(), y.end()), Lf(cols);
  std::vector<uint8_t> b(cols);
  uint16_t iter = 0;
  int32_t status = 0;
  cyclic::detail::check(cc_correct_soft_batch(c, yf.data(), nullptr, nullptr, b.data(), Lf.data(), &iter, &status, 1),
                        "cc_correct_soft_batch");
  if (status != CC_FRAME_OK) throw decoding_failure(cyclic::detail::failure_text(status));
  std::vector<U> bu;
  for (uint8_t v : b) bu.push_back(U(v));
  return std::make_tuple(std::move(bu), std::vector<R>(Lf.begin(), Lf.end()), static_cast<unsigned>(iter));
}
