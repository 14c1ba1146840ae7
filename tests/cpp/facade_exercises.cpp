// Drop-in check of include/channelcoding_amd/cyclic.hpp: the lecture tasks 6.1-6.10 that
// src/exercises.c++ of the reference runs (tasks :36-274), written against the facade with the
// reference's own spelling of types and members.  Expected outputs: SURVEY.md App. B.1 /
// tests/golden/exercises.json.  Exit code 0 = all expectations met.  Needs a GPU at run time.
#include <cstdio>
#include <iostream>
#include <vector>

#include "channelcoding_amd/cyclic.hpp"

// GF(2^10) and GF(2^9) as the reference names them (galois.h:57-67: no default beyond q = 8)
namespace math {
namespace detail {
template <> struct default_modular_polynomial<10> {
  using type = ::math::modular_polynomial<0x409>;
};
template <> struct default_modular_polynomial<9> {
  using type = ::math::modular_polynomial<0x211>;
};
template <> struct default_modular_polynomial<11> {
  using type = ::math::modular_polynomial<0x803>;  // x^11 + x + 1: reducible, so not primitive
};
}  // namespace detail
}  // namespace math

template <typename C1, typename C2> static void expect_equal(const char *what, const C1 &a, const C2 &b) {
  if (a.size() != b.size()) throw std::runtime_error(std::string(what) + ": size mismatch");
  for (size_t i = 0; i < a.size(); i++)
    if (static_cast<unsigned>(a[i]) != static_cast<unsigned>(b[i])) throw std::runtime_error(std::string(what) + ": Decoding Error.");
  std::printf("ok   %s\n", what);
}
template <typename F> static void expect_failure(const char *what, F &&f) {
  try {
    f();
  } catch (const decoding_failure &e) {
    std::printf("ok   %s (decoding_failure: %s)\n", what, e.what());
    return;
  }
  throw std::runtime_error(std::string(what) + ": expected decoding failure");
}

int main() try {
  {  // 6.1  primitive_bch<4, dmin<7>>, default algorithm (PGZ)
    cyclic::primitive_bch<4, dmin<7>> code;
    const std::vector<unsigned char> a({1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1});
    std::vector<unsigned> b1({1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 1, 1});
    std::vector<unsigned> b2({1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1});
    expect_equal("6.1 b1", a, code.correct(b1));
    expect_equal("6.1 b2", a, code.correct(b2));
    if (code.to_string() != "(15, 5, 7)-PGZ") throw std::runtime_error("to_string: " + code.to_string());
    if (code.n != 15 || code.t != 3) throw std::runtime_error("n / t");
  }
  {  // 6.2  dmin<5>: decoding failure expected
    cyclic::primitive_bch<4, dmin<5>> code;
    const std::vector<unsigned> b({1, 0, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 0, 0, 0});
    expect_failure("6.2", [&] { code.correct(b); });
  }
  {  // 6.3  dmin<6>
    cyclic::primitive_bch<4, dmin<6>> code;
    const std::vector<unsigned> a({1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 1, 1});
    const std::vector<unsigned> b1({1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 1});
    const std::vector<unsigned> b2({1, 1, 1, 1, 0, 1, 1, 1, 0, 1, 0, 0, 1, 0, 1});
    expect_equal("6.3 b1", a, code.correct<unsigned>(b1));
    expect_equal("6.3 b2", a, code.correct<unsigned>(b2));
  }
  {  // 6.6 / 6.7 / 6.8 / 6.9  rs<3, errors<2>> with Element-valued sequences and erasures
    using RS_pgz = cyclic::rs<3, errors<2>>;
    using RS_bm = cyclic::rs<3, errors<2>, cyclic::berlekamp_massey_tag>;
    using Element = RS_bm::Element;
    auto P = [](unsigned p) { return Element::from_power(p); };
    const std::vector<Element> a({P(6), P(2), P(2), P(5), P(4), P(6), P(5)});
    std::vector<Element> b66({P(6), P(2), P(2), P(5), Element(0), Element(0), P(5)});
    RS_pgz pgz;
    RS_bm bm;
    expect_equal("6.6", a, pgz.correct<Element>(b66));
    std::vector<Element> b67(a);
    const std::vector<unsigned> erasures = {5, 4, 3, 2};
    for (auto e : erasures) b67.at(e) = Element(0);
    expect_equal("6.7 (erasures)", a, bm.correct<Element>(b67, erasures));
    std::vector<Element> b68({P(2), P(0), P(4), P(0), P(5), P(0), P(2)});
    const std::vector<Element> a68({P(2), P(5), P(4), P(6), P(5), P(6), P(2)});
    expect_equal("6.8 (erasures)", a68, bm.correct<Element>(b68, {1, 3}));
    std::vector<Element> b69({P(3), P(4), P(0), P(3), P(4), P(3), P(3)});
    expect_equal("6.9 pgz == bm", pgz.correct<Element>(b69), bm.correct<Element>(b69));
    if (bm.to_string() != "(7, 3, 6)-BM") throw std::runtime_error("to_string: " + bm.to_string());
  }
  {  // 6.5  rs<4, errors<3>>: failure
    cyclic::rs<4, errors<3>> code;
    using Element = cyclic::rs<4, errors<3>>::Element;
    std::vector<Element> b(15, Element(0));
    for (int i = 0; i < 4; i++) b[i] = Element(1);
    expect_failure("6.5", [&] { code.correct<Element>(b); });
  }
  {  // 6.10  all three locator algorithms agree
    const std::vector<uint8_t> a({1, 0, 1, 0, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1});
    const std::vector<uint8_t> want({1, 0, 1, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1});
    expect_equal("6.10 PGZ", want, cyclic::primitive_bch<4, errors<2>, cyclic::peterson_gorenstein_zierler_tag>().correct(a));
    expect_equal("6.10 BM", want, cyclic::primitive_bch<4, errors<2>, cyclic::berlekamp_massey_tag>().correct(a));
    expect_equal("6.10 EUKLID", want, cyclic::primitive_bch<4, errors<2>, cyclic::euklid_tag>().correct(a));
  }
  {  // encode / decode round trip and the soft path (SURVEY App. B.2 known answer: all-zero word at iteration 0)
    cyclic::primitive_bch<4, errors<2>, cyclic::berlekamp_massey_tag> code;
    const std::vector<uint8_t> msg({1, 0, 1, 1, 0, 0, 1});
    std::vector<uint8_t> cw;
    code.encode(msg, std::back_inserter(cw));
    if (cw.size() != 15) throw std::runtime_error("encode length");
    cw[3] ^= 1;
    cw[11] ^= 1;
    expect_equal("encode -> 2 errors -> decode", msg, code.decode(cw));
    cyclic::primitive_bch<4, errors<2>, min_sum_tag<10>> soft;
    const std::vector<float> y({0.9f, 1.1f, -0.3f, 0.8f, 1.2f, 0.7f, 1.0f, -0.2f, 0.6f, 1.3f, 0.95f, 1.05f, 0.85f, 1.15f, 0.75f});
    expect_equal("min-sum B.2", std::vector<uint8_t>(15, 0), soft.correct<uint8_t>(y));
    if (soft.to_string() != "(15, 7, 5)-MS") throw std::runtime_error("to_string: " + soft.to_string());
    const auto H = soft.H<uint8_t>();
    if (H.rows() != 8 || H.columns() != 15 || !H.at(0).at(7) || H.at(0).at(2)) throw std::runtime_error("H");
    static_assert(normalized_2d_min_sum_tag<10>::alpha == 1.0 && normalized_2d_min_sum_tag<10>::beta == 1.0, "Q11");
    static_assert(normalized_2d_min_sum_tag<10, std::ratio<3, 4>, std::ratio<9, 10>>::beta == 2.25, "Q11");
    // batch form
    std::vector<float> batch;
    for (int f = 0; f < 1000; f++) batch.insert(batch.end(), y.begin(), y.end());
    const auto res = soft.correct_batch(batch.data(), 1000, true);
    for (int f = 0; f < 1000; f++)
      if (res.status[f] != CC_FRAME_OK || res.iters[f] != 0 || res.L[f * 15] != 0.7f) throw std::runtime_error("batch");
    std::printf("ok   batch of 1000 soft frames\n");
    const auto dec = soft.decode_batch(batch.data(), 1000);
    const auto one = soft.decode<std::vector<float>, uint8_t>(y);
    for (int f = 0; f < 1000; f++)
      for (unsigned j = 0; j < one.size(); j++)
        if (dec.msg[f * one.size() + j] != one[j]) throw std::runtime_error("decode_batch (soft)");
    std::vector<uint8_t> words, clean;
    code.encode(msg, std::back_inserter(clean));
    for (int f = 0; f < 50; f++) {
      std::vector<uint8_t> w(clean);
      w[f % 15] ^= 1;
      words.insert(words.end(), w.begin(), w.end());
    }
    const auto hd = code.decode_batch(words.data(), 50);
    for (int f = 0; f < 50; f++)
      for (unsigned j = 0; j < msg.size(); j++)
        if (hd.status[f] != CC_FRAME_OK || hd.msg[f * msg.size() + j] != msg[j]) throw std::runtime_error("decode_batch (hard)");
    std::printf("ok   decode_batch\n");
  }
  {  // H_alt (cyclic.h:361-385) and the free min_sum<R, U>(H, y, tag) of soft_decision.h:220-295
    cyclic::primitive_bch<4, errors<2>, min_sum_tag<10>> code;
    const auto Ha = code.H_alt<uint8_t>();
    if (Ha.rows() != 8 || Ha.columns() != 15) throw std::runtime_error("H_alt shape");
    // row 0..3 = bits of alpha^col: alpha^0 = 1 -> column 0 is (1,0,0,0); alpha^1 = 2 -> column 1 is (0,1,0,0)
    if (!Ha.at(0).at(0) || Ha.at(1).at(0) || Ha.at(0).at(1) || !Ha.at(1).at(1)) throw std::runtime_error("H_alt bits");
    const std::vector<float> y({0.9f, 1.1f, -0.3f, 0.8f, 1.2f, 0.7f, 1.0f, -0.2f, 0.6f, 1.3f, 0.95f, 1.05f, 0.85f, 1.15f, 0.75f});
    const auto viaH = min_sum<float, uint8_t>(code.H<uint8_t>(), y, min_sum_tag<10>());
    expect_equal("free min_sum on H == code.correct", code.correct<uint8_t>(y), std::get<0>(viaH));
    if (std::get<2>(viaH) != 0 || std::get<1>(viaH)[0] != 0.7f) throw std::runtime_error("free min_sum L / iteration");
    const auto viaAlt = min_sum<float, uint8_t>(Ha, y, normalized_min_sum_tag<10, std::ratio<8, 10>>());
    expect_equal("free min_sum on H_alt", std::vector<uint8_t>(15, 0), std::get<0>(viaAlt));
    matrix<uint8_t> one(0, 3);
    one.push_back({1, 1, 0});
    // under the published stop rule (integer dot product) the word 110 is never accepted
    expect_failure("free min_sum, never accepted", [&] {
      min_sum<float, uint8_t>(one, std::vector<float>({-5.0f, -5.0f, 1.0f}), min_sum_tag<3>(), CC_STOP_PUBLISHED);
    });
    const auto ok110 = min_sum<float, uint8_t>(one, std::vector<float>({-5.0f, -5.0f, 1.0f}), min_sum_tag<3>());
    expect_equal("free min_sum, GF(2) parity rule", std::vector<uint8_t>({1, 1, 0}), std::get<0>(ok110));
  }
  {  // q > 8: 16-bit symbols through the _u16 entry points, modular polynomial named by the caller
    using RS = cyclic::rs<10, errors<4>, cyclic::berlekamp_massey_tag>;
    using Element = RS::Element;
    static_assert(std::is_same<Element::storage_type, uint16_t>::value && RS::n == 1023 && RS::wide, "wide field");
    RS code;
    std::vector<Element> msg;
    for (unsigned j = 0; j < RS::n - 8; j++) msg.push_back(Element::from_power(7 * j + 3));
    std::vector<Element> cw;
    code.encode(msg, std::back_inserter(cw));
    if (cw.size() != 1023) throw std::runtime_error("wide encode length");
    std::vector<Element> rx(cw);
    rx[5] = rx[5] + Element::from_power(600);
    rx[400] = rx[400] + Element(1);
    rx[1000] = Element(0);
    rx[1022] = rx[1022] + Element::from_power(1);
    expect_equal("rs<10, errors<4>> correct", cw, code.correct<Element>(rx));
    expect_equal("rs<10, errors<4>> decode", msg, code.decode(rx));
    rx[7] = rx[7] + Element(1);
    rx[8] = rx[8] + Element(1);  // six errors: outside the capability (a miscorrection is possible, equality is not)
    try {
      if (code.correct<Element>(rx) == cw) throw std::runtime_error("rs<10>: six errors corrected?");
      std::printf("ok   rs<10, errors<4>> six errors (another word)\n");
    } catch (const decoding_failure &) {
      std::printf("ok   rs<10, errors<4>> six errors (decoding_failure)\n");
    }
    std::vector<Element> er(cw);
    const std::vector<unsigned> erasures = {0, 17, 512, 1021};
    for (auto e : erasures) er.at(e) = Element(0);
    er[300] = er[300] + Element::from_power(9);
    er[301] = er[301] + Element::from_power(10);
    expect_equal("rs<10> 4 erasures + 2 errors", cw, code.correct<Element>(er, erasures));

    cyclic::primitive_bch<9, errors<3>, cyclic::euklid_tag> bch;
    const unsigned l9 = 511 - 27;
    if (bch.n != 511 || bch.to_string() != "(511, 484, 7)-EUKLID") throw std::runtime_error("bch<9>: " + bch.to_string());
    std::vector<uint16_t> m9(l9), w9;
    for (unsigned j = 0; j < l9; j++) m9[j] = (j * 2654435761u >> 13) & 1;
    bch.encode(m9, std::back_inserter(w9));
    std::vector<uint16_t> r9(w9);
    r9[0] ^= 1;
    r9[255] ^= 1;
    r9[510] ^= 1;
    expect_equal("primitive_bch<9, errors<3>> correct", w9, bch.correct(r9));
    expect_equal("primitive_bch<9, errors<3>> decode", m9, bch.decode(r9));
    std::vector<float> soft9(w9.size());
    for (size_t j = 0; j < w9.size(); j++) soft9[j] = r9[j] ? -1.0f : 1.0f;  // signed input: bit = (x < 0)
    expect_equal("primitive_bch<9> signed input", w9, bch.correct<uint16_t>(soft9));
    r9[3] = 2;  // a non-binary field element in a BCH word is accepted (the field is GF(2^9)), 512 is not an element
    r9[4] = 512;
    try {
      bch.correct(r9);
      throw std::logic_error("512 accepted in GF(2^9)");
    } catch (const std::runtime_error &e) {
      if (std::string(e.what()) != "Value is not an element of the field.") throw;
      std::printf("ok   GF(2^9) range check\n");
    }
    try {
      cyclic::rs<11, errors<2>> bad;
      throw std::logic_error("reducible modular polynomial accepted");
    } catch (const std::runtime_error &e) {
      std::printf("ok   non-primitive modular polynomial refused (%s)\n", e.what());
    }
  }
  std::printf("ALL OK\n");
  return 0;
} catch (const std::exception &e) {
  std::fprintf(stderr, "FAILED: %s\n", e.what());
  return 1;
}
