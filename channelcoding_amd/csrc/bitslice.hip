// bitslice.hip -- syndromes of GF(2^8) codes on bit planes, 32 frames per register.
//
// The syndromes S_j = b(alpha^j), j = 1 .. 2t (cyclic.h:53-63) are Horner chains whose multiplier alpha^j is a
// constant of the code.  With the frames of a batch laid across the BITS of a register -- plane c of a symbol
// position holds bit c of that symbol for 32 frames -- a multiplication by a constant is a fixed XOR network on
// the eight planes (row b of the 8 x 8 binary matrix of "times alpha^j" names the planes that enter plane b),
// on average 25.8 XORs per Horner step for j = 1 .. 32 including the addition of the next symbol: 0.8
// instructions per symbol-multiply instead of a logarithm add, a wrap, a table gather and an XOR per symbol.
//
//   bitslice_fused_syndrome_kernel   bytes [frame][n] -> planes in LDS -> syndromes, back to bytes as
//                                    [block of 64 groups][j][group][32] (group = 32 frames), and the copy of the received
//                                    words into the output buffer (the corrector only patches the errors); RAW: the
//                                    encoder's evaluations, left on planes
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

#include "bitplane.hpp"
#include "cc_internal.hpp"

namespace ccamd {
namespace {

using namespace bitplane;

// ---------------- bytes -> planes in LDS -> syndromes, one kernel ----------------
// Rounds 1-2 had two kernels that exchanged the planes through HBM (255 B written + 255 B read per frame), and the
// syndrome kernel could only fill 2 wavefronts per SIMD (a lane owned a whole group: 2^20 frames were 2048 wavefronts).  Now: a
// workgroup of eight wavefronts takes 8 groups (256 frames, 65 280 contiguous bytes); wavefront w transposes group w
// into LDS, then -- lane = (group, segment of 32 positions), wavefront w = syndromes 4w+1 .. 4w+4 -- every lane runs the
// Horner chains of ITS segment, part_s = sum_i b[32 s + i] alpha^(j i), and the eight segments of a group are folded as
// S_j = sum_s alpha^(32 j s) part_s by a three-level tree over adjacent lanes (multiplications by the constants
// alpha^(32 j), alpha^(64 j), alpha^(128 j): fixed XOR networks again, 13 % on top of the chains).
// LDS: word quadruple `half` of the 32 bytes of (position p, group g) at uint4 index ((p & 31) * 2 + half) * 64 +
// 8 (p >> 5) + (g ^ ((p & 31) >> 2)): the 64 lanes of a syndrome wavefront read 1 KB contiguously, the eight lanes of a
// transposing wavefront that share a segment write to eight different 16-byte columns.
constexpr int kFusedGroups = 8, kFusedThreads = 512;
constexpr size_t kFusedLdsBytes = 32 * 2 * 64 * sizeof(uint4);  // 64 KB: two workgroups per CU, four wavefronts per SIMD

template <int E> __device__ __forceinline__ void times_alpha_e(uint32_t (&x)[8]) {  // x <- x alpha^E
  const uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  horner<E>(x, zero);
}
// part += alpha^E * (part of the lane CTRL names: the next segment, two or four further)
template <int E, int CTRL> __device__ __forceinline__ void fold_segments(uint32_t (&part)[8]) {
  uint32_t t[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) t[b] = part[b];
  times_alpha_e<E>(t);
#pragma unroll
  for (int b = 0; b < 8; ++b)
    part[b] ^= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(t[b]), CTRL, 0xF, 0xF, false));
}
template <int J> __device__ __forceinline__ void fold_all(uint32_t (&part)[8]) {  // valid in the lanes of segment 0
  fold_segments<(32 * J) % 255, 0xB1>(part);   // quad_perm [1,0,3,2]: segment s ^ 1
  fold_segments<(64 * J) % 255, 0x4E>(part);   // quad_perm [2,3,0,1]: segment s ^ 2
  fold_segments<(128 * J) % 255, 0x104>(part);  // row_shl:4: segment s + 4
}

// syndromes J0+1 .. J0+4 of the workgroup's groups: lane = (group lane >> 3, segment lane & 7)
// RAW keeps the result on planes ([block][j][group][8]) for the encoder's interpolation
template <int J0, bool RAW>
__device__ __forceinline__ void fused_syndromes4(const uint4 *__restrict__ lds, uint8_t *__restrict__ synd,
                                                 unsigned long long group0, unsigned long long G, int t2) {
  const int lane = threadIdx.x & 63, g = lane >> 3, seg = lane & 7;
  uint32_t s[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < 8; ++b) s[j][b] = 0;
  auto all4 = [&](const uint32_t (&r)[8]) {
    horner<J0 + 1>(s[0], r);
    horner<J0 + 2>(s[1], r);
    horner<J0 + 3>(s[2], r);
    horner<J0 + 4>(s[3], r);
  };
  // two positions per trip (the in-place update needs no register copies); i >> 2 is the same for both
  for (int i = 31; i >= 1; i -= 2) {
    const int m = 8 * seg + (g ^ (i >> 2));
    const uint4 a0 = lds[(i * 2) * 64 + m], b0 = lds[(i * 2 + 1) * 64 + m];
    const uint4 a1 = lds[(i * 2 - 2) * 64 + m], b1 = lds[(i * 2 - 1) * 64 + m];
    const uint32_t r0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
    const uint32_t r1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
    all4(r0);
    all4(r1);
  }
  fold_all<J0 + 1>(s[0]);
  fold_all<J0 + 2>(s[1]);
  fold_all<J0 + 3>(s[2]);
  fold_all<J0 + 4>(s[3]);
  const unsigned long long gg = group0 + g;
  if (seg != 0 || gg >= G) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (J0 + j >= t2) break;
    if (!RAW) butterfly(s[j]);  // word k = bytes of the frames {k, 8+k, 16+k, 24+k}
    uint4 *dst = reinterpret_cast<uint4 *>(synd + (((gg >> 6) * t2 + (J0 + j)) * 64 + (gg & 63)) * 32);
    dst[0] = make_uint4(s[j][0], s[j][1], s[j][2], s[j][3]);
    dst[1] = make_uint4(s[j][4], s[j][5], s[j][6], s[j][7]);
  }
}

// decoding: n_in = n, off = 0.  Encoding (RAW): the n_in message symbols of a frame go to the positions off .. n - 1 of
// its codeword (cyclic.h:29-40) -- in `out` and on the planes, whose positions below off stay zero -- and the result is
// (a x^k)(alpha^j) on planes for bitslice_parity_kernel.
template <bool FLOAT_IN, bool RAW>
__global__ void __launch_bounds__(kFusedThreads, 4)
bitslice_fused_syndrome_kernel(const void *in_raw, uint8_t *out, uint8_t *__restrict__ synd, unsigned long long B,
                               unsigned long long G, int n, int t2, int n_in, int off) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fused_smem[];
  uint4 *lds = reinterpret_cast<uint4 *>(fused_smem);
#ifdef CC_AMD_EXPERIMENTS  // CC_EXP_FUSED bits (in n >> 16): 1 no copy of the words, 2 no syndromes (phase timing, E27), 4 + 16 k: stagger (E33)
  const int xs = n >> 16;
  n &= 0xFFFF;
#else
  constexpr int xs = 0;
#endif
  const bool copy = static_cast<const void *>(out) != in_raw && !(xs & 1);  // a call that decodes in place has its words there
#ifdef CC_AMD_EXPERIMENTS
  if ((xs & 4) && (blockIdx.x & 1) && blockIdx.x < 512)  // E33: every other workgroup of the first round starts late
    for (int i = 0; i < (xs >> 4); ++i) __builtin_amdgcn_s_sleep(127);
#endif
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long group0 = static_cast<unsigned long long>(blockIdx.x) * kFusedGroups;
  // positions n .. 255 do not exist, positions below off carry nothing yet (the parity symbols): zero planes
  for (int k = threadIdx.x; k < (256 - n + off) * 16; k += kFusedThreads) {
    const int z = k >> 4, p = z < off ? z : n + (z - off), i = p & 31;
    lds[(i * 2 + ((k >> 3) & 1)) * 64 + 8 * (p >> 5) + (k & 7)] = make_uint4(0, 0, 0, 0);
  }
  {  // ---- wavefront = group: lane l owns the four input symbols q .. q+3, q = min(4 l, n_in - 4): one (unaligned) dword per
     //      frame and lane; the last lane re-does symbols of its neighbour (same values) ----
    const unsigned long long g = group0 + wid, f0 = g * 32;
    const int frames = g >= G ? 0 : static_cast<int>((B - f0) < 32ull ? (B - f0) : 32ull);
    const int q = 4 * lane < n_in - 4 ? 4 * lane : n_in - 4;
    if (4 * lane < n_in + 3) {
      uint32_t v[32];
      auto fetch = [&](int f) -> uint32_t {
        const unsigned long long at = (f0 + f) * static_cast<unsigned long long>(n_in) + q;
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
        if (copy) {
#pragma unroll
          for (int f = 0; f < 32; ++f) __builtin_memcpy(out + (f0 + f) * static_cast<unsigned long long>(n) + off + q, &v[f], 4);
        }
      } else {
#pragma unroll
        for (int f = 0; f < 32; ++f) {
          v[f] = 0;
          if (f < frames) {
            v[f] = fetch(f);
            if (copy) __builtin_memcpy(out + (f0 + f) * static_cast<unsigned long long>(n) + off + q, &v[f], 4);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // word k = byte c of the frames {k, 8+k, 16+k, 24+k}
          const uint32_t sel = 0x0c0c0000u | static_cast<uint32_t>((4 + c) << 8) | static_cast<uint32_t>(c);
          const uint32_t lo = __builtin_amdgcn_perm(v[8 + k], v[k], sel), hi = __builtin_amdgcn_perm(v[24 + k], v[16 + k], sel);
          w[k] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        butterfly(w);  // word b, bit f = bit b of the symbol of frame f
        const int p = off + q + c, i = p & 31;
        uint4 *dst = lds + (i * 2) * 64 + 8 * (p >> 5) + (wid ^ (i >> 2));
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[64] = make_uint4(w[4], w[5], w[6], w[7]);
      }
    }
  }
  __syncthreads();
  if (4 * wid >= t2 || (xs & 2)) return;
  switch (wid) {
    case 0: fused_syndromes4<0, RAW>(lds, synd, group0, G, t2); break;
    case 1: fused_syndromes4<4, RAW>(lds, synd, group0, G, t2); break;
    case 2: fused_syndromes4<8, RAW>(lds, synd, group0, G, t2); break;
    case 3: fused_syndromes4<12, RAW>(lds, synd, group0, G, t2); break;
    case 4: fused_syndromes4<16, RAW>(lds, synd, group0, G, t2); break;
    case 5: fused_syndromes4<20, RAW>(lds, synd, group0, G, t2); break;
    case 6: fused_syndromes4<24, RAW>(lds, synd, group0, G, t2); break;
    default: fused_syndromes4<28, RAW>(lds, synd, group0, G, t2); break;
  }
}

// ---------------- systematic encoding by evaluation and interpolation ----------------
// c(x) = a(x) x^k + r(x), deg r < k = 2t, and c(alpha^j) = 0 for j = 1 .. 2t: with E_j = (a x^k)(alpha^j) from the
// Horner kernel above, r is the polynomial with r(alpha^j) = E_j, i.e. r_i = sum_j W[i][j] E_j with W the inverse of
// the Vandermonde matrix V[j][i] = alpha^(j i) -- k^2 multiplications by constants per frame instead of k (n - k),
// and constants are XOR networks on the planes.  Same codeword as the division (the remainder is unique).
struct Gf256 {
  uint8_t exp[512], log[256];
  constexpr Gf256() : exp{}, log{} {
    uint32_t v = 1;
    for (int i = 0; i < 255; ++i) {
      exp[i] = static_cast<uint8_t>(v);
      exp[i + 255] = static_cast<uint8_t>(v);
      log[v] = static_cast<uint8_t>(i);
      v = times_alpha(v);
    }
  }
  constexpr uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
  constexpr uint8_t inv(uint8_t a) const { return exp[255 - log[a]]; }
};
template <int K> struct InverseVandermonde {
  uint8_t w[K][K];
  constexpr InverseVandermonde() : w{} {
    Gf256 f;
    uint8_t v[K][K] = {};
    for (int j = 0; j < K; ++j)
      for (int i = 0; i < K; ++i) {
        v[j][i] = f.exp[((j + 1) * i) % 255];
        w[j][i] = i == j;
      }
    for (int c = 0; c < K; ++c) {  // Gauss-Jordan (V is invertible: distinct evaluation points)
      int piv = c;
      while (v[piv][c] == 0) ++piv;
      for (int i = 0; i < K; ++i) {
        const uint8_t t0 = v[c][i], t1 = w[c][i];
        v[c][i] = v[piv][i];
        v[piv][i] = t0;
        w[c][i] = w[piv][i];
        w[piv][i] = t1;
      }
      const uint8_t s = f.inv(v[c][c]);
      for (int i = 0; i < K; ++i) {
        v[c][i] = f.mul(v[c][i], s);
        w[c][i] = f.mul(w[c][i], s);
      }
      for (int r = 0; r < K; ++r) {
        if (r == c || v[r][c] == 0) continue;
        const uint8_t m = v[r][c];
        for (int i = 0; i < K; ++i) {
          v[r][i] ^= f.mul(m, v[c][i]);
          w[r][i] ^= f.mul(m, w[c][i]);
        }
      }
    }
  }
};
template <int K> inline constexpr InverseVandermonde<K> kInvV{};

constexpr uint32_t times_const(uint32_t c, uint32_t v) {  // c * v in GF(2^8)
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) {
    if ((v >> i) & 1u) r ^= c;
    c = times_alpha(c);
  }
  return r;
}
// acc += C * e on 32 frames (C a compile-time constant: the taps fold to a fixed XOR network)
template <int C> __device__ __forceinline__ void mac(uint32_t (&acc)[8], const uint32_t (&e)[8]) {
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if ((times_const(C, 1u << c) >> b) & 1u) acc[b] ^= e[c];
}
template <int K, int I0, int J, int... I>
__device__ __forceinline__ void interp_column(uint32_t (&acc)[8][8], const uint32_t (&e)[8], std::integer_sequence<int, I...>) {
  (mac<kInvV<K>.w[I0 + I][J]>(acc[I], e), ...);
}
template <int K, int I0, int J>
__device__ __forceinline__ void interp_one(uint32_t (&acc)[8][8], const uint4 *__restrict__ ev) {
  const uint4 a = ev[static_cast<unsigned long long>(J) * 128], b = ev[static_cast<unsigned long long>(J) * 128 + 1];
  const uint32_t e[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  interp_column<K, I0, J>(acc, e, std::make_integer_sequence<int, 8>());
  if (J % 4 == 3) {  // cut the XOR expressions here: reassociation across all K evaluations would keep every one of them live
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(acc[i][b]));
  }
}
template <int K, int I0, int... J>
__device__ __forceinline__ void interp_all(uint32_t (&acc)[8][8], const uint4 *__restrict__ ev, std::integer_sequence<int, J...>) {
  (interp_one<K, I0, J>(acc, ev), ...);
}
// parity symbols I0 .. I0+7 of the 32 frames of group g -> cw[frame][I0 .. I0+7]
template <int K, int I0>
__device__ __forceinline__ void parity8(const uint4 *__restrict__ evals, uint8_t *__restrict__ cw, unsigned long long g,
                                        unsigned long long B, int n) {
  uint32_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[i][b] = 0;
  interp_all<K, I0>(acc, evals + ((g >> 6) * K * 64 + (g & 63)) * 2, std::make_integer_sequence<int, K>());
#pragma unroll
  for (int i = 0; i < 8; ++i) butterfly(acc[i]);  // word k of symbol i = its bytes for the frames {k, 8+k, 16+k, 24+k}
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const unsigned long long frame = g * 32 + 8 * h + k;
      if (frame >= B) continue;
      const uint32_t sel = 0x0c0c0000u | static_cast<uint32_t>((4 + h) << 8) | static_cast<uint32_t>(h);
      uint32_t w[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // byte h of the words k of four adjacent symbols
        const uint32_t lo = __builtin_amdgcn_perm(acc[4 * half + 1][k], acc[4 * half][k], sel);
        const uint32_t hi = __builtin_amdgcn_perm(acc[4 * half + 3][k], acc[4 * half + 2][k], sel);
        w[half] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      __builtin_memcpy(cw + frame * static_cast<unsigned long long>(n) + I0, w, 8);
    }
}
template <int K>
__global__ void __launch_bounds__(256)
bitslice_parity_kernel(const uint4 *__restrict__ evals, uint8_t *__restrict__ cw, unsigned long long B, unsigned long long G,
                       int n) {
  const int wid = threadIdx.x >> 6;
  const unsigned long long g = static_cast<unsigned long long>(blockIdx.x) * 64 + (threadIdx.x & 63);
  if (g >= G) return;
  if constexpr (K == 32) {
    switch (wid) {
      case 0: parity8<K, 0>(evals, cw, g, B, n); break;
      case 1: parity8<K, 8>(evals, cw, g, B, n); break;
      case 2: parity8<K, 16>(evals, cw, g, B, n); break;
      default: parity8<K, 24>(evals, cw, g, B, n); break;
    }
  } else {
    if (wid == 0) parity8<K, 0>(evals, cw, g, B, n);
    else parity8<K, 8>(evals, cw, g, B, n);
  }
}

// ---------------- root search on planes ----------------
// lambda(alpha^-p) for p = 0 .. 254 as a Chien search: the terms T_m = lambda_m alpha^(-m p) advance by the constant
// alpha^(-m) = alpha^(255 - m) per position -- sixteen fixed XOR networks -- and their sum is tested for zero on all 32
// frames of a group at once: bit i of the mask word of (group, p) says "p is a root" for the frame with
// i = 8 (f & 3) + (f >> 2) (the order chunk_bm_kernel transposes in).  Coefficients 0 .. 16; frames whose locator is
// longer are searched by chunk_fix_kernel itself.  Wavefront = 64 groups x 32 positions (segment SEG starts at
// p = 32 SEG with T_m advanced by alpha^(-32 m SEG)), so a lane writes one 128-byte line of masks[group][256].
constexpr int kChienCoef = 17;      // locators of degree <= 16 = t of the largest code: calls without erasures
constexpr int kChienCoefLong = 25;  // calls with erasures: combined locators up to degree 24 (algebraic_chunk.hip)
// rows of the 8 x 8 binary matrices of "times alpha^(-32 m seg)", m = 1 .. 16, seg = 0 .. 7: the advance of T_m to the
// first position of a segment is the one multiplication whose constant differs from wavefront to wavefront, so it
// runs as a masked sum (plane c enters plane b under a wave-uniform mask) instead of a hard-wired network
struct ChienAdvance {
  uint8_t row[8][kChienCoefLong - 1][8];
  constexpr ChienAdvance() : row{} {
    Gf256 f;
    for (int seg = 0; seg < 8; ++seg)
      for (int m = 1; m < kChienCoefLong; ++m) {
        const int e = ((255 - m) * 32 * seg) % 255;
        for (int b = 0; b < 8; ++b) {
          uint32_t r = 0;
          for (int c = 0; c < 8; ++c) r |= ((static_cast<uint32_t>(f.exp[(e + c) % 255]) >> b) & 1u) << c;  // alpha^e x^c
          row[seg][m - 1][b] = static_cast<uint8_t>(r);
        }
      }
  }
};
__constant__ constexpr ChienAdvance kChienAdvance{};

// The seventeen terms of a lane are 136 registers -- two wavefronts per SIMD.  So the coefficients are shared out to
// TWO wavefronts of the same (64 groups, segment): half 0 carries T_0 .. T_8, half 1 carries T_9 .. T_16; each forms
// its partial sums for two positions, one of them (taking turns) hands its 16 words per lane to the other through
// LDS, and that one tests the sums and writes the mask words.  Four wavefronts per SIMD.
template <int M0, int NT, int... M>
__device__ __forceinline__ void chien_step(uint32_t (&T)[NT][8], std::integer_sequence<int, M...>) {
  const uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ((M0 + M == 0 ? (void)0 : horner<(255 - (M0 + M)) % 255>(T[M], zero)), ...);
}
// NCOEF = 17: H = 0 carries coefficients 0 .. 8, H = 1 coefficients 9 .. 16 (its last T unused); NCOEF = 25: 0 .. 12 / 13 .. 24
template <int H, int NCOEF>
__device__ __forceinline__ void chien_half(const uint4 *__restrict__ lamp, uint2 *__restrict__ masks, uint32_t *__restrict__ xch,
                                           unsigned long long blk, int seg, unsigned long long G) {
  constexpr int NT = (NCOEF + 1) / 2, M0 = H ? NT : 0, MC = H ? NCOEF / 2 : NT, PB = 2;  // PB positions per batch (four: spills)
  const int lane = threadIdx.x & 63;
  const unsigned long long g = blk * 64 + lane;
  const bool live = g < G;
  uint32_t T[NT][8];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
#pragma unroll
    for (int b = 0; b < 8; ++b) T[k][b] = 0;
    if (k < MC && live) {
      const uint4 *src = lamp + ((blk * NCOEF + (M0 + k)) * 64 + lane) * 2;
      const uint4 a = src[0], c = src[1];
      T[k][0] = a.x, T[k][1] = a.y, T[k][2] = a.z, T[k][3] = a.w;
      T[k][4] = c.x, T[k][5] = c.y, T[k][6] = c.z, T[k][7] = c.w;
    }
  }
  if (seg != 0) {  // T_m times alpha^(-32 m seg): a masked sum, the constant differs from wavefront to wavefront
#pragma unroll
    for (int k = 0; k < MC; ++k) {
      if (M0 + k == 0) continue;
      uint32_t o[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint32_t row = kChienAdvance.row[seg][M0 + k - 1][b];  // wave-uniform
        uint32_t acc = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc ^= T[k][c] & (0u - ((row >> c) & 1u));
        o[b] = acc;
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) T[k][b] = o[b];
    }
  }
  using Mine = std::make_integer_sequence<int, MC>;
  for (int batch = 0; batch < 32 / PB; ++batch) {
    uint32_t part[PB][8];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        uint32_t sum = T[0][b];
#pragma unroll
        for (int k = 1; k < MC; ++k) sum ^= T[k][b];
        part[u][b] = sum;
      }
      chien_step<M0, NT>(T, Mine());
      // keep the steps apart: flattened over several steps the XOR networks grow into sums over every earlier plane
#pragma unroll
      for (int k = 0; k < MC; ++k)
#pragma unroll
        for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(T[k][b]));
    }
    uint32_t *buf = xch + (batch & 1) * (PB * 8 * 64);
    const bool finisher = (batch & 1) == H;  // wave-uniform
    if (!finisher) {
#pragma unroll
      for (int u = 0; u < PB; ++u)
#pragma unroll
        for (int b = 0; b < 8; ++b) buf[(u * 8 + b) * 64 + lane] = part[u][b];
    }
    __syncthreads();  // (both segments of the workgroup take the same number of trips)
    if (finisher) {
      uint32_t out[PB];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        uint32_t nz = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) nz |= part[u][b] ^ buf[(u * 8 + b) * 64 + lane];
        out[u] = ~nz;
      }
      if (live) masks[g * 128 + 16 * seg + batch] = make_uint2(out[0], out[1]);
    }
  }
}
// workgroup = 64 groups x 2 segments x 2 coefficient halves; a lane writes the 128-byte line of masks[group][256] of its
// segment, 8 bytes at a time
template <int NCOEF>
__global__ void __launch_bounds__(256, NCOEF <= 17 ? 4 : 2)
bitslice_chien_kernel(const uint4 *__restrict__ lamp, uint2 *__restrict__ masks, unsigned long long G) {
  __shared__ uint32_t xch_all[2][2 * 16 * 64];  // [segment of the workgroup][buffer][word][lane]: 16 KB
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long blk = blockIdx.x >> 2;
  const int sg = wid >> 1, seg = static_cast<int>(blockIdx.x & 3) * 2 + sg;
  if (wid & 1) chien_half<1, NCOEF>(lamp, masks, xch_all[sg], blk, seg, G);
  else chien_half<0, NCOEF>(lamp, masks, xch_all[sg], blk, seg, G);
}

// masks[group][256] (word = position, bit = frame in plane order) -> rootsT[group][8][32] (word = frame in plane order,
// bit = position within the segment of 32): a 32 x 32 bit transpose per (group, segment), one lane each, so that the
// lane-per-frame corrector reads the eight words of ITS frame
__global__ void __launch_bounds__(256)
bitslice_roots_transpose_kernel(const uint4 *__restrict__ masks, uint4 *__restrict__ rootsT, unsigned long long tasks) {
  const unsigned long long t = static_cast<unsigned long long>(blockIdx.x) * 256 + threadIdx.x;  // group * 8 + segment
  if (t >= tasks) return;
  uint32_t w[32];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint4 v = masks[t * 8 + i];
    w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int sh = 16; sh >= 1; sh >>= 1) {  // exchange bit `sh` of the word index with bit `sh` of the bit position
    const uint32_t m = sh == 16 ? 0x0000FFFFu : sh == 8 ? 0x00FF00FFu : sh == 4 ? 0x0F0F0F0Fu : sh == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (k & sh) continue;
      const uint32_t x = ((w[k] >> sh) ^ w[k + sh]) & m;
      w[k + sh] ^= x;
      w[k] ^= x << sh;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) rootsT[t * 8 + i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
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
  // Round 2 kept codes with fewer than 8 syndromes on the table kernels (the fixed passes over the batch cost more than
  // the table arithmetic they replaced, r02 E10); with the fused syndrome kernel and the round-3 corrector the planes
  // win for every 2t: BCH(255,231) 1.15 -> 2.55 G frames/s, BCH(255,247) 1.89 -> 3.04 G (profiles/r03_experiments.md, E31)
#ifdef CC_AMD_EXPERIMENTS
  static const size_t min_t2 = [] {
    const char *e = std::getenv("CC_EXP_MIN_T2");
    return e ? static_cast<size_t>(std::atoi(e)) : size_t(2);
  }();
#else
  constexpr size_t min_t2 = 2;
#endif
  if (t2 < min_t2 || t2 > 32 || t.root_powers.size() != t2) return false;
  for (size_t j = 0; j < t2; ++j)
    if (t.root_powers[j] != j + 1) return false;
  return true;
}

// synd: G64 * t2 * 32 bytes, G64 = ceil(B / 2048) * 64 groups of 32 frames; `out` receives the copy of the words
int launch_bitslice_syndromes(const cc_code *code, bool float_in, const void *d_in, uint8_t *d_out, uint8_t *d_synd, size_t B,
                              hipStream_t stream) {
  int n = static_cast<int>(code->tab.n);
  const int t2 = static_cast<int>(code->tab.roots.size());
#ifdef CC_AMD_EXPERIMENTS
  if (const char *x = std::getenv("CC_EXP_FUSED")) n |= std::atoi(x) << 16;
#endif
  const unsigned long long G = (B + 31) / 32, Bq = B;
  const unsigned grid = static_cast<unsigned>((G + kFusedGroups - 1) / kFusedGroups);
  hipError_t e = hipSuccess;
  if (float_in) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bitslice_fused_syndrome_kernel<true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFusedLdsBytes));
    if (e == hipSuccess)
      hipLaunchKernelGGL((bitslice_fused_syndrome_kernel<true, false>), dim3(grid), dim3(kFusedThreads), kFusedLdsBytes, stream,
                         d_in, d_out, d_synd, Bq, G, n, t2, n & 0xFFFF, 0);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bitslice_fused_syndrome_kernel<false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFusedLdsBytes));
    if (e == hipSuccess)
      hipLaunchKernelGGL((bitslice_fused_syndrome_kernel<false, false>), dim3(grid), dim3(kFusedThreads), kFusedLdsBytes, stream,
                         d_in, d_out, d_synd, Bq, G, n, t2, n & 0xFFFF, 0);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "bitslice fused syndrome kernel launch");
  return CC_OK;
}

// lamp: G64 * 17 (25 with long locators) * 32 bytes (chunk_bm kernels), masks: G64 * 256 words
int launch_bitslice_chien(const void *d_lamp, void *d_masks, size_t B, bool long_locators, hipStream_t stream) {
  const unsigned long long G = (B + 31) / 32;
  const dim3 grid(static_cast<unsigned>(4 * ((G + 63) / 64)));
  if (long_locators)
    hipLaunchKernelGGL((bitslice_chien_kernel<kChienCoefLong>), grid, dim3(256), 0, stream, static_cast<const uint4 *>(d_lamp),
                       static_cast<uint2 *>(d_masks), G);
  else
    hipLaunchKernelGGL((bitslice_chien_kernel<kChienCoef>), grid, dim3(256), 0, stream, static_cast<const uint4 *>(d_lamp),
                       static_cast<uint2 *>(d_masks), G);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "bitslice chien kernel launch");
  return CC_OK;
}

int launch_bitslice_roots_transpose(const void *d_masks, void *d_rootsT, size_t B, hipStream_t stream) {
  const unsigned long long tasks = ((B + 31) / 32) * 8;
  hipLaunchKernelGGL(bitslice_roots_transpose_kernel, dim3(static_cast<unsigned>((tasks + 255) / 256)), dim3(256), 0, stream,
                     static_cast<const uint4 *>(d_masks), static_cast<uint4 *>(d_rootsT), tasks);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "bitslice roots transpose kernel launch");
  return CC_OK;
}

bool bitslice_encode_supported(const cc_code *code) {
  if (!bitslice_supported(code) || code->desc.coding != CC_CODING_DIVISION) return false;
  const size_t k = code->tab.roots.size();  // the generator is the product of the (x - root_j): k = 2t parity symbols
  return code->tab.family == CC_FAMILY_RS && code->tab.k == k && (k == 16 || k == 32);
}

// systematic encoder (cyclic.h:29-40, division_tag) on bit planes: message -> planes and codeword body, evaluations
// at the 2t roots, interpolation of the remainder
int launch_bitslice_encode(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, hipStream_t stream) {
  const int n = static_cast<int>(code->tab.n), k = static_cast<int>(code->tab.k), l = static_cast<int>(code->tab.l);
  const unsigned long long G = (B + 31) / 32, Bq = B;
  const size_t G64 = static_cast<size_t>((G + 63) / 64) * 64;
  const size_t eval_bytes = G64 * k * 32;
  uint8_t *d_eval = nullptr;  // stream-ordered and pool-cached
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&d_eval), eval_bytes, stream));
  // message -> codeword body and planes in LDS -> evaluations at the 2t roots (the fused kernel of the decoder, RAW)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bitslice_fused_syndrome_kernel<false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFusedLdsBytes));
  if (e == hipSuccess) {
    hipLaunchKernelGGL((bitslice_fused_syndrome_kernel<false, true>), dim3(static_cast<unsigned>((G + kFusedGroups - 1) / kFusedGroups)),
                       dim3(kFusedThreads), kFusedLdsBytes, stream, static_cast<const void *>(d_msg), d_cw, d_eval, Bq, G, n, k, l, k);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {  // interpolation of the remainder
    if (k == 32)
      hipLaunchKernelGGL((bitslice_parity_kernel<32>), dim3(static_cast<unsigned>((G + 63) / 64)), dim3(256), 0, stream,
                         reinterpret_cast<const uint4 *>(d_eval), d_cw, Bq, G, n);
    else
      hipLaunchKernelGGL((bitslice_parity_kernel<16>), dim3(static_cast<unsigned>((G + 63) / 64)), dim3(128), 0, stream,
                         reinterpret_cast<const uint4 *>(d_eval), d_cw, Bq, G, n);
    e = hipGetLastError();
  }
  (void)hipFreeAsync(d_eval, stream);
  if (e != hipSuccess) return hip_fail(e, "bitslice encode kernels launch");
  return CC_OK;
}

}  // namespace ccamd
