// The reference's `class decoder` (simulation.h:23-69) over GPU-backed codes: the calls awgn_simulation and
// bitflip_simulation make (simulation.c++:100-136, :181-199), a hard and a soft decoder behind the same type, the
// word-error counting of :128-135, and -- with --bench -- single-frame correct() calls per second through the
// whole stack (facade -> C ABI host entry point -> kernel), the figure INTEGRATION.md quotes next to the
// reference's CPU rate.  Exit code 0 = all expectations met.  Needs a GPU at run time.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "channelcoding_amd/simulation.hpp"

static unsigned word_errors(const decoder &d, double ebno_db, size_t frames, uint64_t seed) {
  std::mt19937_64 generator(seed);
  const double sigma = 1.0 / std::sqrt(2.0 * d.rate() * std::pow(10.0, ebno_db / 10.0));  // simulation.c++:83-85
  std::normal_distribution<float> distribution(1.0f, static_cast<float>(sigma));
  std::vector<float> b(d.n());
  unsigned errors = 0;
  for (size_t i = 0; i < frames; i++) {
    for (float &v : b) v = distribution(generator);
    try {
      auto result = d.correct(b);
      for (const auto &bit : result)
        if (bool(bit)) {
          errors++;
          break;
        }
    } catch (const decoding_failure &) {
      errors++;
    }
  }
  return errors;
}

template <typename Code> static void bench(const char *what, size_t calls) {
  decoder d{Code()};
  std::vector<float> b(d.n(), 1.0f);
  b[1] = -0.5f;
  d.correct(b);  // tables, streams and staging buffers exist from here on
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < calls; i++) d.correct(b);
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("bench %-28s %8.0f single-frame correct() calls/s (%.1f us per call)\n", what, calls / s, 1e6 * s / calls);
}

int main(int argc, char **argv) try {
  decoder hard{cyclic::primitive_bch<5, dmin<7>, cyclic::berlekamp_massey_tag>()};
  decoder soft{cyclic::primitive_bch<5, dmin<7>, min_sum_tag<50>>()};
  if (hard.to_string() != "(31, 16, 7)-BM" || soft.to_string() != "(31, 16, 7)-MS") throw std::runtime_error("to_string");
  if (hard.n() != 31 || std::fabs(hard.rate() - 16.0 / 31.0) > 1e-12) throw std::runtime_error("n / rate");
  decoder copy = hard;  // value semantics: copies share the code object
  std::vector<float> clean(31, 1.0f);
  for (const auto &bit : copy.correct(clean))
    if (bool(bit)) throw std::runtime_error("clean frame");
  std::vector<float> one_flip(clean);
  one_flip[7] = -1.0f;
  for (const auto &bit : hard.correct(one_flip))
    if (bool(bit)) throw std::runtime_error("single flip not corrected");
  std::vector<float> four_flips(clean);
  for (int p : {1, 5, 9, 20}) four_flips[p] = -1.0f;
  bool threw = false;
  try {
    hard.correct(four_flips);
  } catch (const decoding_failure &) {
    threw = true;  // (a miscorrection into another codeword would also be legal: weight 4 > t = 3)
  }
  std::printf("ok   hard decoder through class decoder (4 flips: %s)\n", threw ? "decoding_failure" : "other codeword");
  // (31, 16, 7)-BM, WER 0.2079 at 3.0 dB, 0.0225 at 5.0 dB
  const unsigned e3 = word_errors(hard, 3.0, 2000, 1), e5 = word_errors(hard, 5.0, 2000, 2);
  std::printf("ok   word errors of 2000 frames: %u at 3 dB, %u at 5 dB\n", e3, e5);
  if (e3 < 330 || e3 > 500 || e5 < 20 || e5 > 80) throw std::runtime_error("WER outside the reference's curve");
  const unsigned s5 = word_errors(soft, 5.0, 1000, 3);
  std::printf("ok   min-sum decoder through class decoder: %u word errors of 1000 at 5 dB\n", s5);
  // batched form: the same frames, one call
  {
    std::mt19937_64 generator(9);
    std::normal_distribution<float> distribution(1.0f, 0.45f);
    const size_t B = 4096;
    std::vector<float> y(B * 31);
    for (float &v : y) v = distribution(generator);
    std::vector<uint8_t> words;
    std::vector<int32_t> status;
    hard.correct_batch(y.data(), B, words, status);
    unsigned batch_errors = 0, loop_errors = 0;
    for (size_t f = 0; f < B; f++) {
      bool bad = status[f] != CC_FRAME_OK;
      for (size_t j = 0; j < 31 && !bad; j++) bad = words[f * 31 + j] != 0;
      batch_errors += bad;
      std::vector<float> b(y.begin() + f * 31, y.begin() + (f + 1) * 31);
      try {
        bool nz = false;
        for (const auto &bit : hard.correct(b)) nz = nz || bool(bit);
        loop_errors += nz;
      } catch (const decoding_failure &) {
        loop_errors++;
      }
    }
    if (batch_errors != loop_errors) throw std::runtime_error("batch and per-frame word errors differ");
    std::printf("ok   correct_batch == %zu single-frame calls (%u word errors)\n", B, batch_errors);
  }
  if (argc > 1 && std::strcmp(argv[1], "--bench") == 0) {
    bench<cyclic::primitive_bch<4, errors<2>, cyclic::berlekamp_massey_tag>>("BCH(15,7) BM", 20000);
    bench<cyclic::primitive_bch<6, errors<3>, cyclic::berlekamp_massey_tag>>("BCH(63,45) BM", 20000);
    bench<cyclic::primitive_bch<8, errors<3>, cyclic::berlekamp_massey_tag>>("BCH(255,231) BM", 20000);
    bench<cyclic::primitive_bch<6, errors<3>, min_sum_tag<10>>>("BCH(63,45) MS<10>", 20000);
    bench<cyclic::primitive_bch<8, errors<3>, min_sum_tag<20>>>("BCH(255,231) MS<20>", 20000);
  }
  std::printf("ALL OK\n");
  return 0;
} catch (const std::exception &e) {
  std::fprintf(stderr, "FAILED: %s\n", e.what());
  return 1;
}
