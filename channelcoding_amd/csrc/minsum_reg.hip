// minsum_reg.hip -- register-resident min-sum kernel (the fast path for the
// benchmark geometries).  One codeword per wavefront; lane l owns the columns
// j = l + 64c (c < C), all per-edge state lives in VGPRs:
//
//   R[i][c]   one float per (row, owned column) slot.  Between iterations it
//             holds r_ij (check->variable message), inside an iteration it is
//             overwritten in place by q_ij (variable->check) and back.
//             Slots that are not edges of H stay +0.0f forever.
//
// Per iteration and per batch of RB rows:
//   pre    (inline asm, EXEC = edge mask of the slot, from SGPRs):
//            q = (cs - r) + y            two roundings, soft_decision.h:135-136,:207-209
//            (m1, m2) <- insert |q|      lane-local sorted pair
//            neg  ^= ballot(q < 0)       sign parity via SALU popcount
//   reduce (DPP, integer min on the float bit patterns, 1 VALU op / stage):
//            M1 = min over the wave of m1
//            M2 = min over the wave of (lane holds M1 first ? m2 : m1)
//                 = second smallest counting multiplicity
//   post   (inline asm, EXEC = edge mask):
//            t = med3(|q|, M1, M2)  -> M1^M2^t is the exclusive minimum
//            r = sign * h(min) ; cs' += r   (ascending rows, soft_decision.h:88-95)
//
// Exactness argument for dropping the explicit zero count of horizontal__
// (soft_decision.h:109-118): a zero message only matters through sign = 0 for
// the OTHER edges of its row, and for those the exclusive minimum is 0, so
// r = +-h(0) = +-0 for MS / NMS / 2D-NMS / OMS with beta >= 0 -- numerically
// the reference's 0.  q is never -0.0f here (cs starts at +0.0f, y is
// canonicalised on load), so `q < 0` is exactly signum(q) == -1.
// Configurations outside that argument (SCMS1/2, negative offset, non-finite
// alpha) are routed to the generic kernel by the launcher.
#include <cstdio>
#include <cstdlib>

#include "cc_internal.hpp"
#include "wave_ops.hpp"

namespace ccamd {
namespace {

constexpr float kFltMax = 3.402823466e+38f;

// ---- inline-asm row bodies -------------------------------------------------
// A wave issues in order, so the bodies are arranged for ILP: the q = (cs - r) + y updates of all C
// slots of a row run UNMASKED back to back (independent), only the check-node partials run under
// EXEC = edge mask.  Consequence: R of a slot that is not an edge of H holds don't-care garbage (it
// is never read under a mask that includes it, and never accumulated into a column sum).
#define CC_SUB(n) "v_sub_f32 %[r" #n "], %[cs" #n "], %[r" #n "]\n\t"
#define CC_MUL(n) "v_mul_f32 %[r" #n "], %[beta], %[r" #n "]\n\t"
#define CC_ADD(n) "v_add_f32 %[r" #n "], %[r" #n "], %[y" #n "]\n\t"
// masked: insert |q| into the lane-local sorted pair (m1 <= m2); sg ^= q (bit 31 = parity of negatives)
#define CC_PART(n)                                     \
  "s_mov_b64 exec, %[k" #n "]\n\t"                     \
  "v_med3_f32 %[m2], %[m1], |%[r" #n "]|, %[m2]\n\t"   \
  "v_xor_b32 %[sg], %[sg], %[r" #n "]\n\t"             \
  "v_min_f32_e64 %[m1], %[m1], |%[r" #n "]|\n\t"
#define CC_EXEC_ALL "s_mov_b64 exec, -1"

// POST (MS): r = (med3(|q|, M1, M2) ^ Y) + (q & 0x80000000), Y = (M1 ^ M2) | (row sign parity << 31):
//   med3 ^ (M1 ^ M2) is the exclusive minimum, bit 31 adds the sign of q to the row parity.
#define CC_MAG_MS(n) "v_med3_f32 %[t" #n "], |%[r" #n "]|, %[M1], %[M2]\n\t"
// POST (NMS / OMS / 2D-NMS): magnitude = (|q| == M1) ? h(M2) : h(M1)
#define CC_MAG_H(n)                                   \
  "v_cmp_eq_f32_e64 vcc, |%[r" #n "]|, %[M1]\n\t"     \
  "v_cndmask_b32 %[t" #n "], %[H1], %[H2], vcc\n\t"
#define CC_SGN(n) "v_and_b32 %[r" #n "], 0x80000000, %[r" #n "]\n\t"
#define CC_XAD(n) "v_xad_u32 %[r" #n "], %[t" #n "], %[Y], %[r" #n "]\n\t"
#define CC_ACC(n)                                     \
  "s_mov_b64 exec, %[k" #n "]\n\t"                    \
  "v_add_f32 %[cn" #n "], %[cn" #n "], %[r" #n "]\n\t"

template <int C, bool BETA>
struct RowAsm;

template <bool BETA>
struct RowAsm<1, BETA> {
  static __device__ __forceinline__ void pre(float (&r)[1], float &m1, float &m2, uint32_t &sg, const float (&cs)[1],
                                             const float (&y)[1], const uint64_t (&k)[1], float beta) {
    if constexpr (BETA)
      asm(CC_SUB(0) CC_MUL(0) CC_ADD(0) CC_PART(0) CC_EXEC_ALL
                   : [r0] "+v"(r[0]), [m1] "+v"(m1), [m2] "+v"(m2), [sg] "+v"(sg)
                   : [cs0] "v"(cs[0]), [y0] "v"(y[0]), [k0] "s"(k[0]), [beta] "v"(beta));
    else
      asm(CC_SUB(0) CC_ADD(0) CC_PART(0) CC_EXEC_ALL
                   : [r0] "+v"(r[0]), [m1] "+v"(m1), [m2] "+v"(m2), [sg] "+v"(sg)
                   : [cs0] "v"(cs[0]), [y0] "v"(y[0]), [k0] "s"(k[0]));
  }
  static __device__ __forceinline__ void post_ms(float (&r)[1], float (&cn)[1], uint32_t M1, uint32_t M2, uint32_t Y,
                                                 const uint64_t (&k)[1]) {
    uint32_t t0;
    asm("s_nop 1\n\t" CC_MAG_MS(0) CC_SGN(0) CC_XAD(0) CC_ACC(0) CC_EXEC_ALL
                 : [r0] "+v"(r[0]), [cn0] "+v"(cn[0]), [t0] "=&v"(t0)
                 : [M1] "s"(M1), [M2] "v"(M2), [Y] "s"(Y), [k0] "s"(k[0]));
  }
  static __device__ __forceinline__ void post_h(float (&r)[1], float (&cn)[1], uint32_t M1, uint32_t H1, uint32_t H2,
                                                uint32_t Y, const uint64_t (&k)[1]) {
    uint32_t t0;
    asm("s_nop 1\n\t" CC_MAG_H(0) CC_SGN(0) CC_XAD(0) CC_ACC(0) CC_EXEC_ALL
                 : [r0] "+v"(r[0]), [cn0] "+v"(cn[0]), [t0] "=&v"(t0)
                 : [M1] "s"(M1), [H1] "v"(H1), [H2] "v"(H2), [Y] "s"(Y), [k0] "s"(k[0])
                 : "vcc");
  }
};

template <bool BETA>
struct RowAsm<2, BETA> {
  static __device__ __forceinline__ void pre(float (&r)[2], float &m1, float &m2, uint32_t &sg, const float (&cs)[2],
                                             const float (&y)[2], const uint64_t (&k)[2], float beta) {
    if constexpr (BETA)
      asm(CC_SUB(0) CC_SUB(1) CC_MUL(0) CC_MUL(1) CC_ADD(0) CC_ADD(1) CC_PART(0) CC_PART(1) CC_EXEC_ALL
          : [r0] "+v"(r[0]), [r1] "+v"(r[1]), [m1] "+v"(m1), [m2] "+v"(m2), [sg] "+v"(sg)
          : [cs0] "v"(cs[0]), [cs1] "v"(cs[1]), [y0] "v"(y[0]), [y1] "v"(y[1]), [k0] "s"(k[0]), [k1] "s"(k[1]),
            [beta] "v"(beta));
    else
      asm(CC_SUB(0) CC_SUB(1) CC_ADD(0) CC_ADD(1) CC_PART(0) CC_PART(1) CC_EXEC_ALL
          : [r0] "+v"(r[0]), [r1] "+v"(r[1]), [m1] "+v"(m1), [m2] "+v"(m2), [sg] "+v"(sg)
          : [cs0] "v"(cs[0]), [cs1] "v"(cs[1]), [y0] "v"(y[0]), [y1] "v"(y[1]), [k0] "s"(k[0]), [k1] "s"(k[1]));
  }
  static __device__ __forceinline__ void post_ms(float (&r)[2], float (&cn)[2], uint32_t M1, uint32_t M2, uint32_t Y,
                                                 const uint64_t (&k)[2]) {
    uint32_t t0, t1;
    asm("s_nop 1\n\t" CC_MAG_MS(0) CC_MAG_MS(1) CC_SGN(0) CC_SGN(1) CC_XAD(0) CC_XAD(1) CC_ACC(0) CC_ACC(1)
                 CC_EXEC_ALL
                 : [r0] "+v"(r[0]), [r1] "+v"(r[1]), [cn0] "+v"(cn[0]), [cn1] "+v"(cn[1]), [t0] "=&v"(t0), [t1] "=&v"(t1)
                 : [M1] "s"(M1), [M2] "v"(M2), [Y] "s"(Y), [k0] "s"(k[0]), [k1] "s"(k[1]));
  }
  static __device__ __forceinline__ void post_h(float (&r)[2], float (&cn)[2], uint32_t M1, uint32_t H1, uint32_t H2,
                                                uint32_t Y, const uint64_t (&k)[2]) {
    uint32_t t0, t1;
    asm("s_nop 1\n\t" CC_MAG_H(0) CC_MAG_H(1) CC_SGN(0) CC_SGN(1) CC_XAD(0) CC_XAD(1) CC_ACC(0) CC_ACC(1)
                 CC_EXEC_ALL
                 : [r0] "+v"(r[0]), [r1] "+v"(r[1]), [cn0] "+v"(cn[0]), [cn1] "+v"(cn[1]), [t0] "=&v"(t0), [t1] "=&v"(t1)
                 : [M1] "s"(M1), [H1] "v"(H1), [H2] "v"(H2), [Y] "s"(Y), [k0] "s"(k[0]), [k1] "s"(k[1])
                 : "vcc");
  }
};

template <bool BETA>
struct RowAsm<4, BETA> {
  static __device__ __forceinline__ void pre(float (&r)[4], float &m1, float &m2, uint32_t &sg, const float (&cs)[4],
                                             const float (&y)[4], const uint64_t (&k)[4], float beta) {
#define CC_PRE4_OUT \
  [r0] "+v"(r[0]), [r1] "+v"(r[1]), [r2] "+v"(r[2]), [r3] "+v"(r[3]), [m1] "+v"(m1), [m2] "+v"(m2), [sg] "+v"(sg)
#define CC_PRE4_IN                                                                                              \
  [cs0] "v"(cs[0]), [cs1] "v"(cs[1]), [cs2] "v"(cs[2]), [cs3] "v"(cs[3]), [y0] "v"(y[0]), [y1] "v"(y[1]),       \
      [y2] "v"(y[2]), [y3] "v"(y[3]), [k0] "s"(k[0]), [k1] "s"(k[1]), [k2] "s"(k[2]), [k3] "s"(k[3])
#define CC_PRE4_PARTS CC_PART(0) CC_PART(1) CC_PART(2) CC_PART(3) CC_EXEC_ALL
    if constexpr (BETA)
      asm(CC_SUB(0) CC_SUB(1) CC_SUB(2) CC_SUB(3) CC_MUL(0) CC_MUL(1) CC_MUL(2) CC_MUL(3) CC_ADD(0) CC_ADD(1)
                   CC_ADD(2) CC_ADD(3) CC_PRE4_PARTS
                   : CC_PRE4_OUT
                   : CC_PRE4_IN, [beta] "v"(beta));
    else
      asm(CC_SUB(0) CC_SUB(1) CC_SUB(2) CC_SUB(3) CC_ADD(0) CC_ADD(1) CC_ADD(2) CC_ADD(3) CC_PRE4_PARTS
                   : CC_PRE4_OUT
                   : CC_PRE4_IN);
#undef CC_PRE4_OUT
#undef CC_PRE4_IN
#undef CC_PRE4_PARTS
  }
#define CC_POST4_OUT                                                                                          \
  [r0] "+v"(r[0]), [r1] "+v"(r[1]), [r2] "+v"(r[2]), [r3] "+v"(r[3]), [cn0] "+v"(cn[0]), [cn1] "+v"(cn[1]), \
      [cn2] "+v"(cn[2]), [cn3] "+v"(cn[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
#define CC_POST4_TAIL                                                                                     \
  CC_SGN(0) CC_SGN(1) CC_SGN(2) CC_SGN(3) CC_XAD(0) CC_XAD(1) CC_XAD(2) CC_XAD(3) CC_ACC(0) CC_ACC(1) CC_ACC(2) \
      CC_ACC(3) CC_EXEC_ALL
  static __device__ __forceinline__ void post_ms(float (&r)[4], float (&cn)[4], uint32_t M1, uint32_t M2, uint32_t Y,
                                                 const uint64_t (&k)[4]) {
    uint32_t t0, t1, t2, t3;
    asm("s_nop 1\n\t" CC_MAG_MS(0) CC_MAG_MS(1) CC_MAG_MS(2) CC_MAG_MS(3) CC_POST4_TAIL
                 : CC_POST4_OUT
                 : [M1] "s"(M1), [M2] "v"(M2), [Y] "s"(Y), [k0] "s"(k[0]), [k1] "s"(k[1]), [k2] "s"(k[2]), [k3] "s"(k[3]));
  }
  static __device__ __forceinline__ void post_h(float (&r)[4], float (&cn)[4], uint32_t M1, uint32_t H1, uint32_t H2,
                                                uint32_t Y, const uint64_t (&k)[4]) {
    uint32_t t0, t1, t2, t3;
    asm("s_nop 1\n\t" CC_MAG_H(0) CC_MAG_H(1) CC_MAG_H(2) CC_MAG_H(3) CC_POST4_TAIL
                 : CC_POST4_OUT
                 : [M1] "s"(M1), [H1] "v"(H1), [H2] "v"(H2), [Y] "s"(Y), [k0] "s"(k[0]), [k1] "s"(k[1]), [k2] "s"(k[2]),
                   [k3] "s"(k[3])
                 : "vcc");
  }
#undef CC_POST4_OUT
#undef CC_POST4_TAIL
};

template <int VARIANT>
__device__ __forceinline__ float horizontal(float m, float alpha_f, double beta_d) {
  if constexpr (VARIANT == CC_ALG_NMS || VARIANT == CC_ALG_2DNMS) {  // soft_decision.h:211-213
    return __fmul_rn(alpha_f, m);
  } else if constexpr (VARIANT == CC_ALG_OMS) {  // :245-251, evaluated in double
    const double a = static_cast<double>(m) - beta_d;
    return static_cast<float>((a < 0.0) ? 0.0 : a);
  } else {
    return m;
  }
}

// K rows, C columns per lane, RB rows per batch (RB divides K).  256 threads = 4 independent waves.
template <int K, int C, int VARIANT, int RB>
__global__ void __launch_bounds__(256, (K * C <= 96 ? 4 : 2))  // 4 waves/SIMD: plain VALU needs >= 4 waves for full rate
minsum_reg_kernel(MinSumParams p, const uint64_t *__restrict__ emask, const float *__restrict__ llr,
                  const uint16_t *__restrict__ er, const uint32_t *__restrict__ er_off, uint8_t *__restrict__ hard,
                  float *__restrict__ Lout, uint16_t *__restrict__ iters_out, int32_t *__restrict__ status_out,
                  unsigned long long B) {
  static_assert(K % RB == 0 && K <= 32, "row batching");
  constexpr bool BETA = (VARIANT == CC_ALG_2DNMS);
  constexpr bool PLAIN = (VARIANT == CC_ALG_MS);
  using Asm = RowAsm<C, BETA>;
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  const int n = p.n;

  uint32_t cm[C];
  bool colv[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    cm[c] = p.colmask[c * 64 + lane];
    colv[c] = (lane + 64 * c) < n;
  }

  for (unsigned long long frame = wave; frame < B; frame += nwaves) {
    float y[C], cs[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      y[c] = colv[c] ? (llr[frame * n + lane + 64 * c] + 0.0f) : 0.0f;  // -0.0f -> +0.0f
      cs[c] = 0.0f;
    }
    if (er_off != nullptr) {  // cyclic.h:259-262
      for (uint32_t e = er_off[frame]; e < er_off[frame + 1]; ++e) {
        const int pos = er[e];
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (pos == lane + 64 * c) y[c] = 0.0f;
      }
    }
    float R[K][C];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int c = 0; c < C; ++c) R[i][c] = 0.0f;

    unsigned my_iter = p.iterations;
    for (unsigned it = 0; it < p.iterations; ++it) {
      float cn[C];
#pragma unroll
      for (int c = 0; c < C; ++c) cn[c] = 0.0f;

#pragma unroll
      for (int rb = 0; rb < K / RB; ++rb) {
        float m1[RB], m2[RB];
        uint32_t parity = 0;
        // ---- pre: variable-node update + lane-local check-node partials ----
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
          const int i = rb * RB + ii;
          uint64_t k[C];
#pragma unroll
          for (int c = 0; c < C; ++c) k[c] = emask[i * C + c];
          m1[ii] = kFltMax;
          m2[ii] = kFltMax;
          uint32_t sg = 0;
          Asm::pre(R[i], m1[ii], m2[ii], sg, cs, y, k, p.beta_f);
          // q is never -0.0f (header), so bit 31 of the XOR of the row's q is the parity of its negative messages
          parity |= (static_cast<uint32_t>(__builtin_popcountll(__ballot(static_cast<int32_t>(sg) < 0))) & 1u) << ii;
        }
        // ---- reduce: M1 = min, M2 = second smallest (with multiplicity) over the wave ----
        uint32_t M1[RB], M2[RB], red[RB];
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) red[ii] = f2u(m1[ii]);
        wave_umin_batch<RB>(red);
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) M1[ii] = __builtin_amdgcn_readlane(red[ii], 63);
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
          const unsigned long long holders = __ballot(f2u(m1[ii]) == M1[ii]);
          const bool first = __builtin_amdgcn_inverse_ballot_w64(holders & (0ull - holders));
          red[ii] = first ? f2u(m2[ii]) : f2u(m1[ii]);
        }
        wave_umin_batch<RB>(red);
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) M2[ii] = __builtin_amdgcn_readlane(red[ii], 63);
        // ---- post: check-node outputs + column sums (rows ascending) ----
#pragma unroll
        for (int ii = 0; ii < RB; ++ii) {
          const int i = rb * RB + ii;
          uint64_t k[C];
#pragma unroll
          for (int c = 0; c < C; ++c) k[c] = emask[i * C + c];
          const uint32_t sign31 = ((parity >> ii) & 1u) << 31;
          if constexpr (PLAIN) {
            Asm::post_ms(R[i], cn, M1[ii], M2[ii], (M1[ii] ^ M2[ii]) | sign31, k);
          } else {
            const uint32_t H1 = f2u(horizontal<VARIANT>(u2f(M1[ii]), p.alpha_f, p.beta_d));
            const uint32_t H2 = f2u(horizontal<VARIANT>(u2f(M2[ii]), p.alpha_f, p.beta_d));
            Asm::post_h(R[i], cn, M1[ii], H1, H2, sign31, k);
          }
        }
      }

      // ---- a-posteriori values, hard decision, stop test (soft_decision.h:178-186) ----
      uint32_t pv = 0, any = 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        cs[c] = cn[c];
        const bool bit = colv[c] && (cs[c] + y[c] < 0.0f);   // L = cs + y :180-182, b = L < 0 codes.h:51
        pv ^= bit ? cm[c] : 0u;                              // GF(2) syndrome bits of this lane's columns
        any |= bit ? cm[c] : 0u;                             // rows with a non-zero integer dot product
      }
      bool ok;
      if (p.stop_rule == CC_STOP_AS_SHIPPED)
        ok = true;
      else if (p.stop_rule == CC_STOP_PARITY)
        ok = __builtin_amdgcn_readlane(wave_xor(pv), 63) == 0;
      else
        ok = __builtin_amdgcn_readlane(wave_or(any), 63) == 0;
      if (ok) {
        my_iter = it;
        break;
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (colv[c]) {
        const float Lc = (p.iterations == 0) ? 0.0f : cs[c] + y[c];  // the last iteration's L
        hard[frame * n + lane + 64 * c] = (Lc < 0.0f) ? 1 : 0;
        if (Lout) Lout[frame * n + lane + 64 * c] = Lc;
      }
    }
    if (lane == 0) {
      if (iters_out) iters_out[frame] = static_cast<uint16_t>(my_iter);
      if (status_out) status_out[frame] = (my_iter < p.iterations) ? CC_FRAME_OK : CC_FRAME_NOT_CONVERGED;
    }
  }
}

template <int K, int C, int RB>
hipError_t launch_reg(const MinSumParams &p, const uint64_t *emask, int grid, hipStream_t st, const float *llr,
                      const uint16_t *er, const uint32_t *er_off, uint8_t *hard, float *L, uint16_t *iters,
                      int32_t *status, unsigned long long B) {
#define CC_LAUNCH(V)                                                                                              \
  hipLaunchKernelGGL((minsum_reg_kernel<K, C, V, RB>), dim3(grid), dim3(256), 0, st, p, emask, llr, er, er_off, hard, \
                     L, iters, status, B);                                                                        \
  break
  switch (p.variant) {
    case CC_ALG_MS: CC_LAUNCH(CC_ALG_MS);
    case CC_ALG_NMS: CC_LAUNCH(CC_ALG_NMS);
    case CC_ALG_OMS: CC_LAUNCH(CC_ALG_OMS);
    case CC_ALG_2DNMS: CC_LAUNCH(CC_ALG_2DNMS);
    default: return hipErrorInvalidValue;
  }
#undef CC_LAUNCH
  return hipGetLastError();
}

// (k, C, rows per batch) combinations instantiated below.  k*C <= 96 keeps the kernel at 4 waves/SIMD.
struct RegGeometry {
  int K, C, RB;
};
constexpr RegGeometry kRegGeometries[] = {
    {24, 4, 4},  // BCH(255,231) t=3   (headline)
    {16, 4, 4},  // BCH(255,239) t=2
    {8, 4, 4},   // BCH(255,247) t=1
    {32, 4, 4},  // BCH(255,223) t=4   (128 slots: 2-3 waves/SIMD)
    {14, 2, 7},  // BCH(127,113) t=2
    {21, 2, 7},  // BCH(127,106) t=3
    {7, 2, 7},   // BCH(127,120) t=1
    {18, 1, 6},  // BCH(63,45)   t=3
    {24, 1, 6},  // BCH(63,39)   t=4
    {12, 1, 6},  // BCH(63,51)   t=2
    {6, 1, 6},   // BCH(63,57)   t=1
};

int rows_per_batch(int K, int C) {
  static const int env = [] {
    const char *e = std::getenv("CC_AMD_RB");  // tuning knob for the headline geometry only
    return e ? std::atoi(e) : 0;
  }();
  if (K == 24 && C == 4 && (env == 2 || env == 3 || env == 6 || env == 8 || env == 12)) return env;
  for (const RegGeometry &g : kRegGeometries)
    if (g.K == K && g.C == C) return g.RB;
  return 0;
}

}  // namespace

// returns true when a register-resident instantiation exists for this code + algorithm
bool minsum_reg_supported(const cc_code *code) {
  const int alg = code->desc.algorithm;
  if (alg != CC_ALG_MS && alg != CC_ALG_NMS && alg != CC_ALG_OMS && alg != CC_ALG_2DNMS) return false;
  if (alg == CC_ALG_OMS && !(code->desc.beta >= 0.0)) return false;  // h(0) must be 0 (see header)
  const float a = static_cast<float>(code->desc.alpha);
  if ((alg == CC_ALG_NMS || alg == CC_ALG_2DNMS) && !(a == a && a - a == 0.0f)) return false;  // finite alpha
  if (code->geo.W != 64 || code->d_emask == nullptr || !code->custom_H.empty()) return false;
  return rows_per_batch(static_cast<int>(code->tab.k), code->geo.C) != 0;
}

const char *minsum_reg_name(const cc_code *code) {
  const int K = static_cast<int>(code->tab.k), C = code->geo.C;
  static thread_local char buf[64];
  std::snprintf(buf, sizeof buf, "minsum_reg_kernel<K=%d,C=%d,RB=%d>", K, C, rows_per_batch(K, C));
  return buf;
}

int launch_minsum_reg(const cc_code *code, const MinSumParams &p, const float *d_llr, const uint16_t *d_er,
                      const uint32_t *d_er_off, uint8_t *d_hard, float *d_L, uint16_t *d_iters, int32_t *d_status,
                      size_t B, hipStream_t stream) {
  const int K = static_cast<int>(code->tab.k), C = code->geo.C;
  const unsigned long long blocks_needed = (B + 3) / 4;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
  const int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
  hipError_t e = hipErrorInvalidValue;
  const int rb = rows_per_batch(K, C);
#define CC_GO(KK, CCC, RRB)                   \
  if (K == KK && C == CCC && rb == RRB)       \
  e = launch_reg<KK, CCC, RRB>(p, code->d_emask, grid, stream, d_llr, d_er, d_er_off, d_hard, d_L, d_iters, d_status, B)
  CC_GO(24, 4, 4);
  CC_GO(24, 4, 2);
  CC_GO(24, 4, 3);
  CC_GO(24, 4, 6);
  CC_GO(24, 4, 8);
  CC_GO(24, 4, 12);
  CC_GO(16, 4, 4);
  CC_GO(8, 4, 4);
  CC_GO(32, 4, 4);
  CC_GO(14, 2, 7);
  CC_GO(21, 2, 7);
  CC_GO(7, 2, 7);
  CC_GO(18, 1, 6);
  CC_GO(24, 1, 6);
  CC_GO(12, 1, 6);
  CC_GO(6, 1, 6);
#undef CC_GO
  if (e != hipSuccess) return hip_fail(e, "minsum_reg kernel launch");
  return CC_OK;
}

}  // namespace ccamd
