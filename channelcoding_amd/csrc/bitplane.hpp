// bitplane.hpp -- GF(2^8) arithmetic on bit planes (32 frames per register): compile-time XOR networks for the
// multiplication by a constant, and the byte <-> plane transposition.  Shared by bitslice.hip and algebraic_chunk.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>

namespace ccamd {
namespace bitplane {

constexpr uint32_t kPoly = 0x11d;
constexpr uint32_t times_alpha(uint32_t v) { return ((v << 1) & 0x100u) ? ((v << 1) ^ kPoly) : (v << 1); }
constexpr uint32_t times_alpha_pow(uint32_t v, int J) {
  for (int i = 0; i < J; ++i) v = times_alpha(v);
  return v;
}
// does plane C of the operand enter plane B of alpha^J * operand?
template <int J, int B, int C> struct Tap {
  static constexpr bool value = (times_alpha_pow(1u << C, J) >> B) & 1u;
};

template <int J, int B, int... C>
__device__ __forceinline__ uint32_t horner_plane(const uint32_t (&s)[8], uint32_t r, std::integer_sequence<int, C...>) {
  uint32_t acc = r;
  ((Tap<J, B, C>::value ? (void)(acc ^= s[C]) : (void)0), ...);
  return acc;
}
template <int J, int... B>
__device__ __forceinline__ void horner_planes(uint32_t (&s)[8], const uint32_t (&r)[8], std::integer_sequence<int, B...>) {
  const uint32_t o[8] = {horner_plane<J, B>(s, r[B], std::make_integer_sequence<int, 8>())...};
#pragma unroll
  for (int b = 0; b < 8; ++b) s[b] = o[b];
}
// s <- s * alpha^J + r on 32 frames at once
template <int J> __device__ __forceinline__ void horner(uint32_t (&s)[8], const uint32_t (&r)[8]) {
  horner_planes<J>(s, r, std::make_integer_sequence<int, 8>());
}

// 8 words x 32 bits: exchange bit s of the word index with bit s of the bit position, s = 0, 1, 2 (an involution)
__device__ __forceinline__ void butterfly(uint32_t (&w)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const uint32_t t = ((w[k] >> 1) ^ w[k + 1]) & 0x55555555u;
    w[k + 1] ^= t;
    w[k] ^= t << 1;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k & 2) continue;
    const uint32_t t = ((w[k] >> 2) ^ w[k + 2]) & 0x33333333u;
    w[k + 2] ^= t;
    w[k] ^= t << 2;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t t = ((w[k] >> 4) ^ w[k + 4]) & 0x0F0F0F0Fu;
    w[k + 4] ^= t;
    w[k] ^= t << 4;
  }
}


}  // namespace bitplane
}  // namespace ccamd
