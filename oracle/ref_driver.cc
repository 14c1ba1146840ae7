// oracle/ref_driver.cc -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C-ABI wrapper around the *real* reference (hannesweisbach/channelcoding),
// compiled from its headers where they lie under /root/reference/src (see
// oracle/Makefile).  No reference source is copied into this repository; the
// resulting shared objects live in oracle/_ref/ (git-ignored).
//
// Two builds of this one file:
//   libccref_o0.so  -DREF_FIX_END=0  reference exactly as shipped (SURVEY F1:
//                                    matrix::end() const returns begin(), so
//                                    every min-sum returns after iteration 0).
//   libccref_o1.so  -DREF_FIX_END=1  the one-token repair of
//                                    src/math/matrix.h:50 expressed as an
//                                    explicit member specialisation below.
//                                    With U=uint8_t this is oracle "O1"
//                                    (published behaviour), with
//                                    U=math::ef_element<2,1> it is "O2"
//                                    (intended GF(2) parity check).
//
// Portability shims (none of them changes semantics):
//   * -include <...> flags in the Makefile supply std headers the reference
//     relied on libc++ to pull in transitively.
//   * src/math/polynomial.h:47,74 return std::make_tuple where std::pair is
//     declared (accepted by libc++ only); the macro below maps it to
//     make_pair for that one header.
#include <algorithm>
#include <array>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <memory>
#include <numeric>
#include <ratio>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/types.h>
#include <tuple>
#include <utility>
#include <vector>

#define make_tuple make_pair
#include "math/polynomial.h"
#undef make_tuple

#include "math/galois.h"
#include "math/matrix.h"

#ifndef REF_FIX_END
#define REF_FIX_END 0
#endif

#if REF_FIX_END
// matrix.h:50 `const_iterator end() const noexcept { return data.begin(); }`
// repaired to data.end() for the two element types min_sum is instantiated
// with.  Must precede the first implicit instantiation.
template <>
auto matrix<uint8_t>::end() const noexcept -> const_iterator {
  return data.end();
}
template <>
auto matrix<math::ef_element<2, 1>>::end() const noexcept -> const_iterator {
  return data.end();
}
#endif

// q > 8: the reference carries default modular polynomials for q <= 8 only and asks for the rest to be "specified
// manually" (src/math/galois.h:57-67).  These explicit specialisations ARE that manual specification -- user code in
// the sense of the reference, no source of it is touched; they must precede the first use in cyclic.h.
namespace math {
namespace detail {
template <> struct default_modular_polynomial<9> {
  using type = ::math::modular_polynomial<0x211>;  // x^9 + x^4 + 1
};
template <> struct default_modular_polynomial<10> {
  using type = ::math::modular_polynomial<0x409>;  // x^10 + x^3 + 1
};
}  // namespace detail
}  // namespace math

#include "codes/bch.h"
#include "codes/rs.h"

#define API extern "C" __attribute__((visibility("default")))

namespace {

using gf2 = math::ef_element<2, 1>;

struct null_buf : std::streambuf {
  int overflow(int c) override { return c; }
};
null_buf g_null;
struct silence {
  silence() { std::cout.rdbuf(&g_null); }
} g_silence;

enum { ALG_PGZ = 0, ALG_BM = 1, ALG_EUKLID = 2 };
enum { ST_OK = 0, ST_DECODING_FAILURE = 1, ST_RUNTIME_ERROR = 2, ST_OTHER = 3, ST_BAD_ARG = -1 };

void set_what(char *what, int len, const char *msg) {
  if (what && len > 0) {
    std::strncpy(what, msg, static_cast<size_t>(len) - 1);
    what[len - 1] = 0;
  }
}

template <typename F> int guarded(char *what, int whatlen, F &&f) {
  try {
    f();
    set_what(what, whatlen, "");
    return ST_OK;
  } catch (const decoding_failure &e) {
    set_what(what, whatlen, e.what());
    return ST_DECODING_FAILURE;
  } catch (const std::runtime_error &e) {
    set_what(what, whatlen, e.what());
    return ST_RUNTIME_ERROR;
  } catch (const std::exception &e) {
    set_what(what, whatlen, e.what());
    return ST_OTHER;
  }
}

/* ---- type-erased view of one reference code (three locator algorithms) --- */
struct code_iface {
  virtual ~code_iface() = default;
  int family = 0, q = 0, cap_kind = 0, cap = 0;
  unsigned n = 0, k = 0, l = 0, t = 0, dmin = 0;
  double rate = 0;
  std::vector<uint8_t> g, h, roots;
  virtual std::string to_string(int alg) const = 0;
  virtual std::vector<uint8_t> H_u8() const = 0;
  virtual void encode(const uint8_t *msg, uint8_t *cw) const = 0;
  virtual void correct_u8(int alg, const uint8_t *in, const std::vector<unsigned> &er, uint8_t *out) const = 0;
  virtual void correct_f32(int alg, const float *in, const std::vector<unsigned> &er, uint8_t *out) const = 0;
  virtual void decode_u8(int alg, const uint8_t *in, const std::vector<unsigned> &er, uint8_t *out) const = 0;
  virtual void locator(int alg, const uint8_t *in, const std::vector<unsigned> &er, std::vector<uint8_t> &synd,
                       std::vector<uint8_t> &sigma) const = 0;
  virtual int minsum(int variant, unsigned iters, int utype, const float *y, uint8_t *b, float *L,
                     unsigned *iter) const = 0;
  virtual void encode_mult(const uint8_t *msg, uint8_t *cw) const = 0;
  virtual void decode_mult(const uint8_t *cw, uint8_t *msg) const = 0;
  virtual std::vector<uint8_t> H_alt_u8() const = 0;
  virtual int minsum_alt(int variant, unsigned iters, int utype, const float *y, uint8_t *b, float *L,
                         unsigned *iter) const = 0;
};

/* Protected members (g, h, roots, k, l, dmin) live in the cyclic::cyclic<...>
 * base; primitive_bch / rs hide `g` behind a private static g().  Reach them
 * through a pointer-to-member formed in a class derived from the deduced base. */
template <typename Base> struct Access : Base {
  using Poly = typename Base::Polynomial;
  using Elem = typename Base::Element;
  static const Poly &G(const Base &b) { return b.*(&Access::g); }
  static const Poly &Hp(const Base &b) { return b.*(&Access::h); }
  static const std::vector<Elem> &R(const Base &b) { return b.*(&Access::roots); }
  static unsigned K(const Base &b) { return b.*(&Access::k); }
  static unsigned Lm(const Base &b) { return b.*(&Access::l); }
  static unsigned D(const Base &b) { return b.*(&Access::dmin); }
};
template <unsigned q, typename C, typename A, unsigned N, typename Co, typename E>
const cyclic::cyclic<q, C, A, N, Co, E> &base_of(const cyclic::cyclic<q, C, A, N, Co, E> &c) {
  return c;
}
template <typename Code> struct Peek : Code {
  using Base = typename std::decay<decltype(base_of(std::declval<const Code &>()))>::type;
  const typename Code::Polynomial &G() const { return Access<Base>::G(*this); }
  const typename Code::Polynomial &Hp() const { return Access<Base>::Hp(*this); }
  const std::vector<typename Code::Element> &R() const { return Access<Base>::R(*this); }
  unsigned K() const { return Access<Base>::K(*this); }
  unsigned Lm() const { return Access<Base>::Lm(*this); }
  unsigned D() const { return Access<Base>::D(*this); }
};

/* min-sum variants available (template parameters are compile-time in the
 * reference: soft_decision.h:20-73). */
enum {
  V_MS = 0,
  V_NMS_8_10 = 1,
  V_OMS_1_100 = 2,
  V_SCMS1 = 3,
  V_SCMS2 = 4,
  V_2DNMS_DEFAULT = 5,
  V_2DNMS_34_910 = 6,
  V_NMS_3_4 = 7,
  V_OMS_15_100 = 8,
  V_COUNT
};

template <unsigned It, typename U>
std::tuple<std::vector<U>, std::vector<float>, unsigned> run_variant(int variant, const matrix<U> &H,
                                                                    const std::vector<float> &y) {
  switch (variant) {
  case V_MS:
    return min_sum<float, U>(H, y, min_sum_tag<It>{});
  case V_NMS_8_10:
    return min_sum<float, U>(H, y, normalized_min_sum_tag<It, std::ratio<8, 10>>{});
  case V_OMS_1_100:
    return min_sum<float, U>(H, y, offset_min_sum_tag<It, std::ratio<1, 100>>{});
  case V_SCMS1:
    return min_sum<float, U>(H, y, self_correcting_1_min_sum_tag<It>{});
  case V_SCMS2:
    return min_sum<float, U>(H, y, self_correcting_2_min_sum_tag<It>{});
  case V_2DNMS_DEFAULT:
    return min_sum<float, U>(H, y, normalized_2d_min_sum_tag<It>{});
  case V_2DNMS_34_910:
    return min_sum<float, U>(H, y, normalized_2d_min_sum_tag<It, std::ratio<3, 4>, std::ratio<9, 10>>{});
  case V_NMS_3_4:
    return min_sum<float, U>(H, y, normalized_min_sum_tag<It, std::ratio<3, 4>>{});
  case V_OMS_15_100:
    return min_sum<float, U>(H, y, offset_min_sum_tag<It, std::ratio<15, 100>>{});
  default:
    throw std::invalid_argument("unknown min-sum variant");
  }
}

template <typename U>
std::tuple<std::vector<U>, std::vector<float>, unsigned> run_iters(unsigned iters, int variant, const matrix<U> &H,
                                                                  const std::vector<float> &y) {
  switch (iters) {
  case 1:
    return run_variant<1, U>(variant, H, y);
  case 2:
    return run_variant<2, U>(variant, H, y);
  case 3:
    return run_variant<3, U>(variant, H, y);
  case 5:
    return run_variant<5, U>(variant, H, y);
  case 10:
    return run_variant<10, U>(variant, H, y);
  case 20:
    return run_variant<20, U>(variant, H, y);
  case 50:
    return run_variant<50, U>(variant, H, y);
  default:
    throw std::invalid_argument("iteration count not instantiated");
  }
}

template <typename PGZ, typename BM, typename EUK> struct code_impl : code_iface {
  Peek<PGZ> pgz;
  Peek<BM> bm;
  Peek<EUK> euk;
  using Element = typename PGZ::Element;
  using Polynomial = typename PGZ::Polynomial;

  code_impl(int family_, int q_, int cap_kind_, int cap_) {
    family = family_;
    q = q_;
    cap_kind = cap_kind_;
    cap = cap_;
    n = PGZ::n;
    t = PGZ::t;
    k = pgz.K();
    l = pgz.Lm();
    dmin = pgz.D();
    rate = pgz.rate;
    for (const auto &e : pgz.G())
      g.push_back(static_cast<uint8_t>(static_cast<unsigned>(e)));
    for (const auto &e : pgz.Hp())
      h.push_back(static_cast<uint8_t>(static_cast<unsigned>(e)));
    for (const auto &e : pgz.R())
      roots.push_back(static_cast<uint8_t>(static_cast<unsigned>(e)));
  }

  std::string to_string(int alg) const override {
    switch (alg) {
    case ALG_PGZ:
      return pgz.to_string();
    case ALG_BM:
      return bm.to_string();
    default:
      return euk.to_string();
    }
  }

  std::vector<uint8_t> H_u8() const override {
    auto Hm = pgz.template H<uint8_t>();
    std::vector<uint8_t> out;
    for (size_t r = 0; r < Hm.rows(); r++)
      for (size_t c = 0; c < Hm.columns(); c++)
        out.push_back(Hm.at(r).at(c));
    return out;
  }

  void encode(const uint8_t *msg, uint8_t *cw) const override {
    std::vector<uint8_t> a(msg, msg + l), out;
    pgz.encode(a, std::back_inserter(out));
    if (out.size() != n)
      throw std::logic_error("encode produced wrong length");
    std::copy(out.begin(), out.end(), cw);
  }

  /* the free functions of cyclic.h:29-33 and :42-46 with this code's generator: what a code instantiated with
   * Coding = multiplication_tag runs inside encode() (:303) and decode() (:318-319) */
  void encode_mult(const uint8_t *msg, uint8_t *cw) const override {
    Polynomial a;
    for (unsigned i = 0; i < l; i++)
      a.push_back(Element(msg[i]));
    auto enc = ::cyclic::encode(pgz.G(), a, ::cyclic::multiplication_tag());
    if (enc.size() > n)
      throw std::logic_error("encode produced wrong length");
    std::fill(cw, cw + n, 0); /* fill_n(out, n - enc.size(), 0), cyclic.h:310 */
    for (size_t i = 0; i < enc.size(); i++)
      cw[i] = static_cast<uint8_t>(static_cast<unsigned>(enc[i]));
  }
  void decode_mult(const uint8_t *cw, uint8_t *msg) const override {
    Polynomial b;
    for (unsigned i = 0; i < n; i++)
      b.push_back(Element(cw[i]));
    auto dec = ::cyclic::decode(pgz.G(), b, ::cyclic::multiplication_tag());
    std::fill(msg, msg + l, 0); /* fill_n(back_inserter(r), l - b_.size(), 0), cyclic.h:325 */
    for (size_t i = 0; i < dec.size() && i < l; i++)
      msg[i] = static_cast<uint8_t>(static_cast<unsigned>(dec[i]));
  }

  template <typename Seq> void correct_any(int alg, const Seq &b, const std::vector<unsigned> &er, uint8_t *out) const {
    std::vector<uint8_t> r;
    switch (alg) {
    case ALG_PGZ:
      r = pgz.template correct<uint8_t>(b, er);
      break;
    case ALG_BM:
      r = bm.template correct<uint8_t>(b, er);
      break;
    case ALG_EUKLID:
      r = euk.template correct<uint8_t>(b, er);
      break;
    default:
      throw std::invalid_argument("unknown algorithm");
    }
    if (r.size() != n)
      throw std::logic_error("correct produced wrong length");
    std::copy(r.begin(), r.end(), out);
  }

  void correct_u8(int alg, const uint8_t *in, const std::vector<unsigned> &er, uint8_t *out) const override {
    std::vector<uint8_t> b(in, in + n);
    correct_any(alg, b, er, out);
  }
  void correct_f32(int alg, const float *in, const std::vector<unsigned> &er, uint8_t *out) const override {
    std::vector<float> b(in, in + n);
    correct_any(alg, b, er, out);
  }

  void decode_u8(int alg, const uint8_t *in, const std::vector<unsigned> &er, uint8_t *out) const override {
    std::vector<uint8_t> b(in, in + n), r;
    switch (alg) {
    case ALG_PGZ:
      r = pgz.decode(b, er);
      break;
    case ALG_BM:
      r = bm.decode(b, er);
      break;
    case ALG_EUKLID:
      r = euk.decode(b, er);
      break;
    default:
      throw std::invalid_argument("unknown algorithm");
    }
    if (r.size() != l)
      throw std::logic_error("decode produced wrong length");
    std::copy(r.begin(), r.end(), out);
  }

  void locator(int alg, const uint8_t *in, const std::vector<unsigned> &er, std::vector<uint8_t> &synd,
               std::vector<uint8_t> &sigma) const override {
    Polynomial b_;
    for (unsigned i = 0; i < n; i++)
      b_.push_back(Element(in[i]));
    auto S = cyclic::calculate_syndromes(b_, pgz.R());
    synd.clear();
    for (const auto &s : S)
      synd.push_back(static_cast<uint8_t>(static_cast<unsigned>(s)));
    sigma.clear();
    if (std::none_of(S.begin(), S.end(), [](const Element &e) { return bool(e); }))
      return;
    Polynomial sig;
    switch (alg) {
    case ALG_PGZ:
      sig = cyclic::error_locator_polynomial<Polynomial>(S, er, cyclic::peterson_gorenstein_zierler_tag{});
      break;
    case ALG_BM:
      sig = cyclic::error_locator_polynomial<Polynomial>(S, er, cyclic::berlekamp_massey_tag{});
      break;
    default:
      sig = cyclic::error_locator_polynomial<Polynomial>(S, er, cyclic::euklid_tag{});
      break;
    }
    for (const auto &s : sig)
      sigma.push_back(static_cast<uint8_t>(static_cast<unsigned>(s)));
  }

  std::vector<uint8_t> H_alt_u8() const override {
    auto Hm = pgz.template H_alt<uint8_t>();
    std::vector<uint8_t> out;
    for (size_t r = 0; r < Hm.rows(); r++)
      for (size_t c = 0; c < Hm.columns(); c++)
        out.push_back(Hm.at(r).at(c));
    return out;
  }

  /* min_sum over the alternative parity-check matrix cyclic.h:361-385 */
  int minsum_alt(int variant, unsigned iters, int utype, const float *yin, uint8_t *b, float *L,
                 unsigned *iter) const override {
    std::vector<float> y(yin, yin + n);
    if (utype == 0) {
      auto res = run_iters<uint8_t>(iters, variant, pgz.template H_alt<uint8_t>(), y);
      for (unsigned i = 0; i < n; i++) {
        b[i] = std::get<0>(res)[i];
        L[i] = std::get<1>(res)[i];
      }
      *iter = std::get<2>(res);
    } else {
      /* H_alt<ef_element<2,1>>() does not compile in the reference (cyclic.h:379: no push_back(bool)) */
      throw std::invalid_argument("H_alt<gf2> is ill-formed in the reference");
    }
    return 0;
  }

  int minsum(int variant, unsigned iters, int utype, const float *yin, uint8_t *b, float *L,
             unsigned *iter) const override {
    std::vector<float> y(yin, yin + n);
    if (utype == 0) {
      /* exactly cyclic.h:264-265: min_sum<float, uint8_t>(H<uint8_t>(), copy, Algorithm{}) */
      auto res = run_iters<uint8_t>(iters, variant, pgz.template H<uint8_t>(), y);
      const auto &bb = std::get<0>(res);
      const auto &LL = std::get<1>(res);
      for (unsigned i = 0; i < n; i++) {
        b[i] = bb[i];
        L[i] = LL[i];
      }
      *iter = std::get<2>(res);
    } else {
      auto res = run_iters<gf2>(iters, variant, pgz.template H<gf2>(), y);
      const auto &bb = std::get<0>(res);
      const auto &LL = std::get<1>(res);
      for (unsigned i = 0; i < n; i++) {
        b[i] = bool(bb[i]) ? 1 : 0;
        L[i] = LL[i];
      }
      *iter = std::get<2>(res);
    }
    return 0;
  }
};

template <unsigned q, typename Cap> std::unique_ptr<code_iface> make_bch(int cap_kind, int cap) {
  using P = cyclic::primitive_bch<q, Cap, cyclic::peterson_gorenstein_zierler_tag>;
  using B = cyclic::primitive_bch<q, Cap, cyclic::berlekamp_massey_tag>;
  using E = cyclic::primitive_bch<q, Cap, cyclic::euklid_tag>;
  return std::unique_ptr<code_iface>(new code_impl<P, B, E>(0, q, cap_kind, cap));
}
template <unsigned q, typename Cap> std::unique_ptr<code_iface> make_rs(int cap_kind, int cap) {
  using P = cyclic::rs<q, Cap, cyclic::peterson_gorenstein_zierler_tag>;
  using B = cyclic::rs<q, Cap, cyclic::berlekamp_massey_tag>;
  using E = cyclic::rs<q, Cap, cyclic::euklid_tag>;
  return std::unique_ptr<code_iface>(new code_impl<P, B, E>(1, q, cap_kind, cap));
}

std::vector<std::unique_ptr<code_iface>> &codes() {
  static std::vector<std::unique_ptr<code_iface>> v = [] {
    std::vector<std::unique_ptr<code_iface>> c;
    /* cap_kind: 0 = errors<>, 1 = dmin<> */
    c.push_back(make_bch<4, errors<2>>(0, 2));  /*  0 BCH(15,7)   */
    c.push_back(make_bch<4, dmin<7>>(1, 7));    /*  1 BCH(15,5)   exercises 6.1 */
    c.push_back(make_bch<4, dmin<5>>(1, 5));    /*  2 BCH(15,7)   exercises 6.2 */
    c.push_back(make_bch<4, dmin<6>>(1, 6));    /*  3 BCH(15,7)   exercises 6.3 */
    c.push_back(make_bch<5, dmin<7>>(1, 7));    /*  4 BCH(31,16)  bitflips.c++  */
    c.push_back(make_bch<6, errors<3>>(0, 3));  /*  5 BCH(63,45)  */
    c.push_back(make_bch<8, errors<3>>(0, 3));  /*  6 BCH(255,231) */
    c.push_back(make_rs<3, errors<1>>(0, 1));   /*  7 RS(7,5)     exercises 6.4 */
    c.push_back(make_rs<3, errors<2>>(0, 2));   /*  8 RS(7,3)     exercises 6.6-6.9 */
    c.push_back(make_rs<4, errors<3>>(0, 3));   /*  9 RS(15,9)    exercises 6.5 */
    c.push_back(make_rs<8, errors<16>>(0, 16)); /* 10 RS(255,223) */
    c.push_back(make_bch<7, dmin<5>>(1, 5));    /* 11 BCH(127,113) benchmark.c++ family */
    c.push_back(make_bch<5, dmin<5>>(1, 5));    /* 12 BCH(31,21)  */
    c.push_back(make_bch<6, dmin<9>>(1, 9));    /* 13 BCH(63,39)  */
    c.push_back(make_rs<8, errors<40>>(0, 40)); /* 14 RS(255,175): more than 64 syndromes */
    c.push_back(make_bch<8, errors<40>>(0, 40)); /* 15 BCH(255,47) */
    return c;
  }();
  return v;
}

code_iface *get(int id) {
  auto &c = codes();
  if (id < 0 || id >= static_cast<int>(c.size()))
    return nullptr;
  return c[static_cast<size_t>(id)].get();
}

std::vector<unsigned> er_vec(const unsigned *er, int ne) {
  return (er && ne > 0) ? std::vector<unsigned>(er, er + ne) : std::vector<unsigned>();
}

/* Full class path cyclic::correct_(soft_decision_tag) (cyclic.h:254-267) for
 * a few fixed instantiations; pins erasure zeroing and result conversion. */
template <typename Code> void soft_class(const float *yin, const std::vector<unsigned> &er, uint8_t *out) {
  static const Code code;
  std::vector<float> y(yin, yin + Code::n);
  auto r = code.template correct<uint8_t>(y, er);
  std::copy(r.begin(), r.end(), out);
}

} // namespace

API int ref_fix_end(void) { return REF_FIX_END; }
API int ref_num_codes(void) { return static_cast<int>(codes().size()); }
API int ref_num_variants(void) { return V_COUNT; }

API int ref_code_info(int id, int *family, int *q, int *cap_kind, int *cap, unsigned *n, unsigned *k, unsigned *l,
                      unsigned *t, unsigned *dmin, double *rate) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  *family = c->family;
  *q = c->q;
  *cap_kind = c->cap_kind;
  *cap = c->cap;
  *n = c->n;
  *k = c->k;
  *l = c->l;
  *t = c->t;
  *dmin = c->dmin;
  *rate = c->rate;
  return ST_OK;
}

/* which: 0 = g, 1 = h, 2 = roots. Returns length written (<= cap) or <0. */
API int ref_get_poly(int id, int which, uint8_t *out, int cap) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  const std::vector<uint8_t> &v = which == 0 ? c->g : (which == 1 ? c->h : c->roots);
  int len = static_cast<int>(v.size());
  if (len > cap)
    return ST_BAD_ARG;
  std::copy(v.begin(), v.end(), out);
  return len;
}

API int ref_to_string(int id, int alg, char *out, int cap) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  set_what(out, cap, c->to_string(alg).c_str());
  return ST_OK;
}

API int ref_get_H(int id, uint8_t *out) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  auto H = c->H_u8();
  std::copy(H.begin(), H.end(), out);
  return ST_OK;
}

API int ref_encode(int id, const uint8_t *msg, uint8_t *cw, char *what, int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->encode(msg, cw); });
}

API int ref_encode_mult(int id, const uint8_t *msg, uint8_t *cw, char *what, int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->encode_mult(msg, cw); });
}

API int ref_decode_mult(int id, const uint8_t *cw, uint8_t *msg, char *what, int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->decode_mult(cw, msg); });
}

API int ref_correct_u8(int id, int alg, const uint8_t *in, const unsigned *er, int ne, uint8_t *out, char *what,
                       int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->correct_u8(alg, in, er_vec(er, ne), out); });
}

API int ref_correct_f32(int id, int alg, const float *in, const unsigned *er, int ne, uint8_t *out, char *what,
                        int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->correct_f32(alg, in, er_vec(er, ne), out); });
}

API int ref_decode_u8(int id, int alg, const uint8_t *in, const unsigned *er, int ne, uint8_t *out, char *what,
                      int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->decode_u8(alg, in, er_vec(er, ne), out); });
}

/* syndromes (2t bytes) and the locator polynomial as returned by
 * error_locator_polynomial (hard_decision.h). *nsigma = 0 when all syndromes
 * vanish. */
API int ref_locator(int id, int alg, const uint8_t *in, const unsigned *er, int ne, uint8_t *synd, uint8_t *sigma,
                    int *nsigma, int sigma_cap, char *what, int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  *nsigma = 0;
  return guarded(what, whatlen, [&] {
    std::vector<uint8_t> s, sg;
    c->locator(alg, in, er_vec(er, ne), s, sg);
    std::copy(s.begin(), s.end(), synd);
    if (static_cast<int>(sg.size()) > sigma_cap)
      throw std::length_error("sigma too long");
    std::copy(sg.begin(), sg.end(), sigma);
    *nsigma = static_cast<int>(sg.size());
  });
}

/* utype 0: U = uint8_t  (what cyclic::correct_ instantiates, cyclic.h:258-265)
 * utype 1: U = math::ef_element<2,1>  (true GF(2) parity check, "O2") */
API int ref_minsum(int id, int variant, unsigned iters, int utype, const float *y, uint8_t *b, float *L,
                   unsigned *iter, char *what, int whatlen) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->minsum(variant, iters, utype, y, b, L, iter); });
}

/* Batch form: per-frame status (0 ok / 1 decoding_failure / other), outputs
 * untouched on failure.  Returns wall seconds through *seconds. */
API int ref_minsum_batch(int id, int variant, unsigned iters, int utype, const float *y, size_t frames, uint8_t *b,
                         float *L, unsigned *iter, int *status, double *seconds) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  std::vector<uint8_t> bt(c->n);
  std::vector<float> Lt(c->n);
  auto t0 = std::chrono::steady_clock::now();
  for (size_t f = 0; f < frames; f++) {
    unsigned it = 0;
    int st = guarded(nullptr, 0, [&] { c->minsum(variant, iters, utype, y + f * c->n, bt.data(), Lt.data(), &it); });
    if (status)
      status[f] = st;
    if (st == ST_OK) {
      if (b)
        std::copy(bt.begin(), bt.end(), b + f * c->n);
      if (L)
        std::copy(Lt.begin(), Lt.end(), L + f * c->n);
      if (iter)
        iter[f] = it;
    } else if (iter) {
      iter[f] = iters;
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds)
    *seconds = std::chrono::duration<double>(t1 - t0).count();
  return ST_OK;
}

API int ref_get_H_alt(int id, uint8_t *out) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  auto H = c->H_alt_u8();
  std::copy(H.begin(), H.end(), out);
  return ST_OK;
}

API int ref_minsum_alt_batch(int id, int variant, unsigned iters, int utype, const float *y, size_t frames,
                             uint8_t *b, float *L, unsigned *iter, int *status) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  std::vector<uint8_t> bt(c->n);
  std::vector<float> Lt(c->n);
  for (size_t f = 0; f < frames; f++) {
    unsigned it = 0;
    int st = guarded(nullptr, 0, [&] { c->minsum_alt(variant, iters, utype, y + f * c->n, bt.data(), Lt.data(), &it); });
    status[f] = st;
    iter[f] = st == ST_OK ? it : iters;
    if (st == ST_OK) {
      std::copy(bt.begin(), bt.end(), b + f * c->n);
      std::copy(Lt.begin(), Lt.end(), L + f * c->n);
    }
  }
  return ST_OK;
}

API int ref_correct_u8_batch(int id, int alg, const uint8_t *in, size_t frames, uint8_t *out, int *status,
                             double *seconds) {
  auto c = get(id);
  if (!c)
    return ST_BAD_ARG;
  std::vector<unsigned> none;
  auto t0 = std::chrono::steady_clock::now();
  for (size_t f = 0; f < frames; f++) {
    int st = guarded(nullptr, 0, [&] { c->correct_u8(alg, in + f * c->n, none, out + f * c->n); });
    if (status)
      status[f] = st;
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds)
    *seconds = std::chrono::duration<double>(t1 - t0).count();
  return ST_OK;
}

/* Class-path soft decode: selector = small fixed table of instantiations. */
API int ref_soft_class(int selector, const float *y, const unsigned *er, int ne, uint8_t *out, char *what,
                       int whatlen) {
  using namespace cyclic;
  auto e = er_vec(er, ne);
  return guarded(what, whatlen, [&] {
    switch (selector) {
    case 0:
      soft_class<primitive_bch<4, errors<2>, min_sum_tag<10>>>(y, e, out);
      break;
    case 1:
      soft_class<primitive_bch<4, errors<2>, normalized_min_sum_tag<10, std::ratio<8, 10>>>>(y, e, out);
      break;
    case 2:
      soft_class<primitive_bch<4, errors<2>, offset_min_sum_tag<10, std::ratio<1, 100>>>>(y, e, out);
      break;
    case 3:
      soft_class<primitive_bch<4, errors<2>, self_correcting_1_min_sum_tag<10>>>(y, e, out);
      break;
    case 4:
      soft_class<primitive_bch<4, errors<2>, self_correcting_2_min_sum_tag<10>>>(y, e, out);
      break;
    case 5:
      soft_class<primitive_bch<4, errors<2>, normalized_2d_min_sum_tag<10>>>(y, e, out);
      break;
    case 6:
      soft_class<primitive_bch<5, dmin<7>, min_sum_tag<50>>>(y, e, out);
      break;
    case 7:
      soft_class<primitive_bch<6, errors<3>, min_sum_tag<10>>>(y, e, out);
      break;
    case 8:
      soft_class<primitive_bch<8, errors<3>, min_sum_tag<20>>>(y, e, out);
      break;
    default:
      throw std::invalid_argument("unknown selector");
    }
  });
}

/* alpha / beta as the tag types report them (soft_decision.h:36-73, Q11). */
API int ref_tag_constants(double *nms_8_10_alpha, double *oms_1_100_beta, double *d2_default_alpha,
                          double *d2_default_beta, double *d2_34_910_alpha, double *d2_34_910_beta) {
  *nms_8_10_alpha = normalized_min_sum_tag<10, std::ratio<8, 10>>::alpha;
  *oms_1_100_beta = offset_min_sum_tag<10, std::ratio<1, 100>>::beta;
  *d2_default_alpha = normalized_2d_min_sum_tag<10>::alpha;
  *d2_default_beta = normalized_2d_min_sum_tag<10>::beta;
  *d2_34_910_alpha = normalized_2d_min_sum_tag<10, std::ratio<3, 4>, std::ratio<9, 10>>::alpha;
  *d2_34_910_beta = normalized_2d_min_sum_tag<10, std::ratio<3, 4>, std::ratio<9, 10>>::beta;
  return ST_OK;
}


/* ---------------------------------------------------------------------------------------------------------------
 * q > 8: symbols are uint16_t (the reference's storage_type, galois.h:44-53).  Two codes, three / two algorithms:
 *   wide id 0: primitive_bch<9, errors<3>>   (511, 484, 7)     PGZ, BM, Euklid
 *   wide id 1: rs<10, errors<4>>             (1023, 1015, 10)  PGZ, BM, Euklid
 * ------------------------------------------------------------------------------------------------------------- */
namespace {
struct wide_iface {
  virtual ~wide_iface() = default;
  unsigned n = 0, k = 0, l = 0, t = 0, dmin = 0, q = 0;
  int family = 0;
  std::vector<uint16_t> g, h, roots;
  virtual std::string to_string(int alg) const = 0;
  virtual void encode(const uint16_t *msg, uint16_t *cw) const = 0;
  virtual void correct(int alg, const uint16_t *in, const std::vector<unsigned> &er, uint16_t *out) const = 0;
  virtual void decode(int alg, const uint16_t *in, const std::vector<unsigned> &er, uint16_t *out) const = 0;
};
template <typename PGZ, typename BM, typename EUK> struct wide_impl : wide_iface {
  Peek<PGZ> pgz;
  Peek<BM> bm;
  Peek<EUK> euk;
  wide_impl(int family_, unsigned q_) {
    family = family_;
    q = q_;
    n = PGZ::n;
    t = PGZ::t;
    k = pgz.K();
    l = pgz.Lm();
    dmin = pgz.D();
    for (const auto &e : pgz.G()) g.push_back(static_cast<uint16_t>(static_cast<unsigned>(e)));
    for (const auto &e : pgz.Hp()) h.push_back(static_cast<uint16_t>(static_cast<unsigned>(e)));
    for (const auto &e : pgz.R()) roots.push_back(static_cast<uint16_t>(static_cast<unsigned>(e)));
  }
  std::string to_string(int alg) const override {
    return alg == ALG_PGZ ? pgz.to_string() : alg == ALG_BM ? bm.to_string() : euk.to_string();
  }
  void encode(const uint16_t *msg, uint16_t *cw) const override {
    std::vector<uint16_t> a(msg, msg + l), out;
    pgz.encode(a, std::back_inserter(out));
    if (out.size() != n) throw std::logic_error("encode produced wrong length");
    std::copy(out.begin(), out.end(), cw);
  }
  void correct(int alg, const uint16_t *in, const std::vector<unsigned> &er, uint16_t *out) const override {
    std::vector<uint16_t> b(in, in + n), r;
    switch (alg) {
    case ALG_PGZ: r = pgz.template correct<uint16_t>(b, er); break;
    case ALG_BM: r = bm.template correct<uint16_t>(b, er); break;
    default: r = euk.template correct<uint16_t>(b, er); break;
    }
    if (r.size() != n) throw std::logic_error("correct produced wrong length");
    std::copy(r.begin(), r.end(), out);
  }
  void decode(int alg, const uint16_t *in, const std::vector<unsigned> &er, uint16_t *out) const override {
    std::vector<uint16_t> b(in, in + n), r;
    switch (alg) {
    case ALG_PGZ: r = pgz.template decode<std::vector<uint16_t>, uint16_t>(b, er); break;
    case ALG_BM: r = bm.template decode<std::vector<uint16_t>, uint16_t>(b, er); break;
    default: r = euk.template decode<std::vector<uint16_t>, uint16_t>(b, er); break;
    }
    if (r.size() != l) throw std::logic_error("decode produced wrong length");
    std::copy(r.begin(), r.end(), out);
  }
};
std::vector<std::unique_ptr<wide_iface>> &wide_codes() {
  static std::vector<std::unique_ptr<wide_iface>> v = [] {
    std::vector<std::unique_ptr<wide_iface>> c;
    c.emplace_back(new wide_impl<cyclic::primitive_bch<9, errors<3>, cyclic::peterson_gorenstein_zierler_tag>,
                                 cyclic::primitive_bch<9, errors<3>, cyclic::berlekamp_massey_tag>,
                                 cyclic::primitive_bch<9, errors<3>, cyclic::euklid_tag>>(0, 9));
    c.emplace_back(new wide_impl<cyclic::rs<10, errors<4>, cyclic::peterson_gorenstein_zierler_tag>,
                                 cyclic::rs<10, errors<4>, cyclic::berlekamp_massey_tag>,
                                 cyclic::rs<10, errors<4>, cyclic::euklid_tag>>(1, 10));
    return c;
  }();
  return v;
}
wide_iface *wget(int id) {
  auto &v = wide_codes();
  return (id >= 0 && static_cast<size_t>(id) < v.size()) ? v[static_cast<size_t>(id)].get() : nullptr;
}
}  // namespace

API int refw_num_codes(void) { return static_cast<int>(wide_codes().size()); }
API int refw_info(int id, int *family, unsigned *q, unsigned *n, unsigned *k, unsigned *l, unsigned *t, unsigned *dmin) {
  auto c = wget(id);
  if (!c) return ST_BAD_ARG;
  *family = c->family;
  *q = c->q;
  *n = c->n;
  *k = c->k;
  *l = c->l;
  *t = c->t;
  *dmin = c->dmin;
  return ST_OK;
}
API int refw_get_poly(int id, int which, uint16_t *out, int cap) {
  auto c = wget(id);
  if (!c) return -1;
  const std::vector<uint16_t> &v = which == 0 ? c->g : which == 1 ? c->h : c->roots;
  if (static_cast<int>(v.size()) > cap) return -1;
  std::copy(v.begin(), v.end(), out);
  return static_cast<int>(v.size());
}
API int refw_to_string(int id, int alg, char *out, int cap) {
  auto c = wget(id);
  if (!c) return ST_BAD_ARG;
  set_what(out, cap, c->to_string(alg).c_str());
  return ST_OK;
}
API int refw_encode(int id, const uint16_t *msg, uint16_t *cw, char *what, int whatlen) {
  auto c = wget(id);
  if (!c) return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->encode(msg, cw); });
}
API int refw_correct(int id, int alg, const uint16_t *in, const unsigned *er, int ne, uint16_t *out, char *what,
                     int whatlen) {
  auto c = wget(id);
  if (!c) return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->correct(alg, in, er_vec(er, ne), out); });
}
API int refw_decode(int id, int alg, const uint16_t *in, const unsigned *er, int ne, uint16_t *out, char *what,
                    int whatlen) {
  auto c = wget(id);
  if (!c) return ST_BAD_ARG;
  return guarded(what, whatlen, [&] { c->decode(alg, in, er_vec(er, ne), out); });
}
