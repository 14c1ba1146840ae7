// Drop-in check of include/channelcoding_amd/cyclic.hpp: the lecture tasks 6.1-6.10 that
// src/exercises.c++ of the reference runs (tasks :36-274), written against the facade with the
// reference's own spelling of types and members.  Expected outputs: SURVEY.md App. B.1 /
// tests/golden/exercises.json.  Exit code 0 = all expectations met.  Needs a GPU at run time.
#include <cstdio>
#include <iostream>
#include <vector>

#include "channelcoding_amd/cyclic.hpp"

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
  }
  std::printf("ALL OK\n");
  return 0;
} catch (const std::exception &e) {
  std::fprintf(stderr, "FAILED: %s\n", e.what());
  return 1;
}
