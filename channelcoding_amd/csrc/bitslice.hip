// bitslice.hip -- syndromes of GF(2^8) codes on bit planes, 32 frames per register.
//
// The syndromes S_j = b(alpha^j), j = 1 .. 2t (cyclic.h:53-63) are Horner chains whose multiplier alpha^j is a
// constant of the code.  With the frames of a batch laid across the BITS of a register -- plane c of a symbol
// position holds bit c of that symbol for 32 frames -- a multiplication by a constant is a fixed XOR network on
// the eight planes (row b of the 8 x 8 binary matrix of "times alpha^j" names the planes that enter plane b),
// on average 25.8 XORs per Horner step for j = 1 .. 32 including the addition of the next symbol: 0.8
// instructions per symbol-multiply instead of a logarithm add, a wrap, a table gather and an XOR per symbol.
//
//   bitslice_planes_kernel     bytes [frame][n] -> planes [block of 64 groups][p][group][8] (group = 32 frames;
//                              2 KB per position and block, read by one wavefront at a time), and the copy of the
//                              received words into the output buffer (the corrector only patches the errors);
//                              wavefront = group, lane = four adjacent positions
//   bitslice_syndrome_kernel   lane = group, wavefront w of a workgroup = syndromes 8w+1 .. 8w+8 of the same 64
//                              groups; the result goes back to bytes as [block][j][group][32]
//
// The byte <-> plane transposition is three butterfly stages on eight registers whose word k carries the
// frames {k, 8+k, 16+k, 24+k} of the group, so no bit permutation is left over; the consumer
// (algebraic_chunk_kernel<PRE>) reads syndrome j of frame 8i + k of group g = 64 gb + gl at byte
// ((gb 2t + j) 64 + gl) 32 + 4k + i.
//
// Field: GF(2^8) with the default modular polynomial 0x11d (galois.h:18-20), consecutive roots alpha^1 .. alpha^2t
// (mu = step = 1), n = 255.  Everything else stays on the table kernels.
#include <cstdlib>
#include <utility>

#include "cc_internal.hpp"

namespace ccamd {
namespace {

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

// One wavefront per group of 32 frames.  Lane l owns the four positions q .. q+3, q = min(4 l, n - 4): one (unaligned)
// dword per frame and lane instead of four byte loads -- the byte form was bound by the rate of the memory
// instructions, 64 B each.  The last lane re-does position n - 4 .. 4 l - 1 of its neighbour (same values).
template <bool FLOAT_IN>
__global__ void __launch_bounds__(256)
bitslice_planes_kernel(const void *__restrict__ in_raw, uint8_t *__restrict__ out, uint4 *__restrict__ planes,
                       unsigned long long B, unsigned long long G, int n) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  const int q = 4 * lane < n - 4 ? 4 * lane : n - 4;
  for (unsigned long long g = wave; g < G; g += nwaves) {
    const unsigned long long f0 = g * 32;
    const int frames = static_cast<int>((B - f0) < 32ull ? (B - f0) : 32ull);
    uint32_t v[32];
    auto fetch = [&](int f) -> uint32_t {
      const unsigned long long at = (f0 + f) * static_cast<unsigned long long>(n) + q;
      if (FLOAT_IN) {  // hard decision of a signed sequence: cyclic.h:163-173, codes.h:43-52
        const float *x = static_cast<const float *>(in_raw) + at;
        return (x[0] < 0.0f ? 1u : 0u) | (x[1] < 0.0f ? 0x100u : 0u) | (x[2] < 0.0f ? 0x10000u : 0u) |
               (x[3] < 0.0f ? 0x1000000u : 0u);
      }
      uint32_t r;
      __builtin_memcpy(&r, static_cast<const uint8_t *>(in_raw) + at, 4);
      return r;
    };
    if (frames == 32) {
#pragma unroll
      for (int f = 0; f < 32; ++f) v[f] = fetch(f);
#pragma unroll
      for (int f = 0; f < 32; ++f) __builtin_memcpy(out + (f0 + f) * static_cast<unsigned long long>(n) + q, &v[f], 4);
    } else {
#pragma unroll
      for (int f = 0; f < 32; ++f) {
        v[f] = 0;
        if (f < frames) {
          v[f] = fetch(f);
          __builtin_memcpy(out + (f0 + f) * static_cast<unsigned long long>(n) + q, &v[f], 4);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // word k = byte i of the frames {k, 8+k, 16+k, 24+k}
        const uint32_t sel = 0x0c0c0000u | static_cast<uint32_t>((4 + i) << 8) | static_cast<uint32_t>(i);
        const uint32_t lo = __builtin_amdgcn_perm(v[8 + k], v[k], sel), hi = __builtin_amdgcn_perm(v[24 + k], v[16 + k], sel);
        w[k] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      butterfly(w);  // word b, bit f = bit b of the symbol of frame f
      uint4 *dst = planes + (((g >> 6) * n + (q + i)) * 64 + (g & 63)) * 2;
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

template <int J0>  // syndromes of the roots alpha^(J0+1) .. alpha^(J0+8)
__device__ __forceinline__ void syndromes8(const uint4 *__restrict__ planes, uint8_t *__restrict__ synd,
                                           unsigned long long G, unsigned long long g, int n, int t2) {
  uint32_t s[8][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int b = 0; b < 8; ++b) s[j][b] = 0;
  const uint4 *src = planes + ((g >> 6) * n * 64 + (g & 63)) * 2;
  const unsigned long long pitch = 128;
  auto all8 = [&](const uint32_t (&r)[8]) {
    horner<J0 + 1>(s[0], r);
    horner<J0 + 2>(s[1], r);
    horner<J0 + 3>(s[2], r);
    horner<J0 + 4>(s[3], r);
    horner<J0 + 5>(s[4], r);
    horner<J0 + 6>(s[5], r);
    horner<J0 + 7>(s[6], r);
    horner<J0 + 8>(s[7], r);
  };
  // two positions per trip: the planes written by the first step are the operands of the second, so the in-place
  // update needs no register copies
  int p = n - 1;
  uint4 a0 = src[static_cast<unsigned long long>(p) * pitch], b0 = src[static_cast<unsigned long long>(p) * pitch + 1];
  if ((n & 1) != 0) {
    const uint32_t r[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
    all8(r);
    --p;
    if (p >= 0) {
      a0 = src[static_cast<unsigned long long>(p) * pitch];
      b0 = src[static_cast<unsigned long long>(p) * pitch + 1];
    }
  }
  for (; p >= 1; p -= 2) {
    const uint4 a1 = src[static_cast<unsigned long long>(p - 1) * pitch], b1 = src[static_cast<unsigned long long>(p - 1) * pitch + 1];
    const uint32_t r0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
    const uint32_t r1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
    const int pn = p >= 2 ? p - 2 : 0;  // the last trip re-reads position 0 (no branch around the prefetch)
    a0 = src[static_cast<unsigned long long>(pn) * pitch];
    b0 = src[static_cast<unsigned long long>(pn) * pitch + 1];
    all8(r0);
    all8(r1);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (J0 + j >= t2) break;
    butterfly(s[j]);  // word k = bytes of the frames {k, 8+k, 16+k, 24+k}
    uint4 *dst = reinterpret_cast<uint4 *>(synd + (((g >> 6) * t2 + (J0 + j)) * 64 + (g & 63)) * 32);
    dst[0] = make_uint4(s[j][0], s[j][1], s[j][2], s[j][3]);
    dst[1] = make_uint4(s[j][4], s[j][5], s[j][6], s[j][7]);
  }
}

__global__ void __launch_bounds__(256)
bitslice_syndrome_kernel(const uint4 *__restrict__ planes, uint8_t *__restrict__ synd, unsigned long long G, int n, int t2) {
  const int wid = threadIdx.x >> 6;
  const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * 64 + (threadIdx.x & 63);
  if (g >= G) return;
  switch (wid) {
    case 0: syndromes8<0>(planes, synd, G, g, n, t2); break;
    case 1: syndromes8<8>(planes, synd, G, g, n, t2); break;
    case 2: syndromes8<16>(planes, synd, G, g, n, t2); break;
    default: syndromes8<24>(planes, synd, G, g, n, t2); break;
  }
}

}  // namespace

bool bitslice_supported(const cc_code *code) {
  static const bool disabled = [] {
    const char *e = std::getenv("CC_AMD_NO_BITSLICE");
    return e && e[0] == '1';
  }();
  if (disabled || !code->field) return false;
  const CodeTables &t = code->tab;
  if (t.q != 8 || t.n != 255 || code->field->poly != kPoly) return false;
  const size_t t2 = t.roots.size();
  if (t2 < 8 || t2 > 32 || t.root_powers.size() != t2) return false;
  for (size_t j = 0; j < t2; ++j)
    if (t.root_powers[j] != j + 1) return false;
  return true;
}

// planes: G64 * n * 32 bytes, synd: G64 * t2 * 32 bytes, G64 = ceil(B / 2048) * 64 groups of 32 frames
int launch_bitslice_syndromes(const cc_code *code, bool float_in, const void *d_in, uint8_t *d_out, void *d_planes,
                              uint8_t *d_synd, size_t B, hipStream_t stream) {
  const int n = static_cast<int>(code->tab.n), t2 = static_cast<int>(code->tab.roots.size());
  const unsigned long long G = (B + 31) / 32, Bq = B;
  const unsigned long long want = (G + 3) / 4, cap = static_cast<unsigned long long>(code->num_cus) * 32;
  const int grid = static_cast<int>(want < cap ? want : cap);
  if (float_in)
    hipLaunchKernelGGL((bitslice_planes_kernel<true>), dim3(grid), dim3(256), 0, stream, d_in, d_out,
                       static_cast<uint4 *>(d_planes), Bq, G, n);
  else
    hipLaunchKernelGGL((bitslice_planes_kernel<false>), dim3(grid), dim3(256), 0, stream, d_in, d_out,
                       static_cast<uint4 *>(d_planes), Bq, G, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "bitslice planes kernel launch");
  const int waves = (t2 + 7) / 8;
  hipLaunchKernelGGL(bitslice_syndrome_kernel, dim3(static_cast<unsigned>((G + 63) / 64)), dim3(64 * waves), 0, stream,
                     static_cast<const uint4 *>(d_planes), d_synd, G, n, t2);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "bitslice syndrome kernel launch");
  return CC_OK;
}

}  // namespace ccamd
