// minsum_diag_impl.hpp -- diagonal-parallel min-sum kernel: 16 (or 8) lanes per codeword, four (eight) codewords
// per wavefront, every (lane, slot, row) triple is a real edge of H.  The text below describes the BCH(255,231)
// instantiation; the template parameters generalise it (see minsum_diag.hip).
//
// Why (measured on MI355X, profiles/r01_ubench_instruction_rates.txt, r03_experiments.md): at the two waves per SIMD
// that 168 message registers allow, a wave issues one instruction per ~7 cycles whatever its class, so what counts is
// the NUMBER of instructions per frame-iteration.  A column-parallel layout (lane = column, round 1) spends 96
// slot-instructions per frame-iteration at 44 % lane utilisation (H is a band of density 112/255) plus 2 x 6 DPP
// stages per row for ONE frame.  Here:
//   * H is banded Toeplitz (cyclic.h:346-359): edge (row i, diagonal s) sits in column s + i.  The
//     w = 112 diagonals of BCH(255,231) are dealt 7 to each of 16 lanes, so a frame costs 7 x 24 dense
//     slot-instructions shared by the 4 frames of the wave (42 per frame-iteration), no EXEC masks,
//     no scalar mask loads;
//   * a row reduction is a 4-stage all-reduce inside a 16-lane DPP row, shared by 4 frames;
//   * per-edge messages r/q stay in VGPRs (168 per lane); the per-column state lives in LDS:
//       CY[col] = {cs, y}   column sums of the previous iteration and the channel value (one ds_read_b64)
//       CN[col] = cs'       column sums being accumulated, read-modify-written row by row in ascending
//                           row order -- the reference's summation order (soft_decision.h:88-95).
//     Within one row all 112 diagonals hit distinct columns, successive rows are ordered by program
//     order (LDS operations of a wave execute in order), so no atomics are needed.
//   * lane groups are persistent: a group that converged (or ran out of iterations) stores its frame and sets
//     the next one up while the other groups keep iterating.  The next frame's channel values are already in
//     LDS by then: global_load_lds copied them there (no VGPR on the way, so nothing the register allocator
//     could hand to a live value while a load is in flight) during the whole lifetime of the current frame.
//
// Exactness argument for dropping the explicit zero count of horizontal__ (soft_decision.h:109-118): a zero message
// only matters through sign = 0 for the OTHER edges of its row, and for those the exclusive minimum is 0, so
// r = +-h(0) = +-0 for MS / NMS / 2D-NMS / OMS with beta >= 0 -- numerically the reference's 0.  q is never -0.0f
// (cs starts at +0.0f, y is canonicalised on load), so `q < 0` is exactly signum(q) == -1.  The exclusive minimum
// (|q| == min1 ? min2 : min1) equals the reference's O(w^2) search, ties included.  Configurations outside that
// argument (negative offset, non-finite alpha) are routed to the generic kernel by the launcher; SCMS1 / SCMS2 create
// zeros on purpose and carry their own bookkeeping (BITS1 / KEEPQ below).
// (kernel template and per-geometry launcher; instantiated by minsum_diag.hip and minsum_diag_small.hip)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "cc_internal.hpp"
#include "wave_ops.hpp"


namespace ccamd {
namespace {



// compile-time loop: indices are literal constants from the start, so the register arrays below are
// scalarised by SROA before any unrolling heuristics get a say (a partially unrolled loop over R[K][D]
// sends the array to scratch memory)
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}
// Paired slots (see build_paired_table in minsum_diag.hip): slots 2p and 2p+1 hold diagonals s and s + G_p in every
// lane, so both edges are addressed from ONE base register with compile-time offsets and the compiler merges the
// two LDS accesses into ds_read2_b64 / ds_read2_b32 / ds_write2_b32.
template <int... Gs>
struct PairGaps {
  static constexpr int NP = sizeof...(Gs);
  static constexpr int gap(int p) {
    constexpr int a[sizeof...(Gs) + 1] = {Gs..., 0};
    return a[p];
  }
};
template <typename PG>
constexpr int slot_base(int d) { return d < 2 * PG::NP ? (d & ~1) : d; }
template <typename PG>
constexpr int slot_gap(int d) { return (d < 2 * PG::NP && (d & 1)) ? PG::gap(d >> 1) : 0; }

// (a ^ b) + c in one VALU op
__device__ __forceinline__ uint32_t xad(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int CTRL, uint32_t IDENT>
__device__ __forceinline__ uint32_t dpp16(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(IDENT), static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }

// one butterfly stage of the (min1, min2-with-multiplicity, sign parity) all-reduce inside a 16-lane row.
// Instruction order pinned: max_dpp(m1), min_dpp(m1), xor_dpp(sg), mov_dpp(m2), min3.  A DPP operand needs two
// wait states after the VALU write of its register; in this order every register is written at least three
// instructions before the next stage reads it through DPP, so no s_nop is issued (one per stage before).
template <int CTRL>
__device__ __forceinline__ void reduce_stage(uint32_t &m1, uint32_t &m2, uint32_t &sg) {
  // all four patterns used here (quad_perm, row_half_mirror, row_mirror) read a valid lane everywhere, so the
  // `old` operand is never selected: mov_dpp leaves it undefined, update_dpp(m2, m2) would cost a v_mov to tie it
  const uint32_t hi = umax32(m1, dpp16<CTRL, 0u>(m1));
  __builtin_amdgcn_sched_barrier(0);
  m1 = umin32(m1, dpp16<CTRL, 0xFFFFFFFFu>(m1));
  __builtin_amdgcn_sched_barrier(0);
  sg = sg ^ dpp16<CTRL, 0u>(sg);
  __builtin_amdgcn_sched_barrier(0);
  const uint32_t o2 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m2), CTRL, 0xF, 0xF, true));
  __builtin_amdgcn_sched_barrier(0);
  m2 = umin32(hi, umin32(m2, o2));
  __builtin_amdgcn_sched_barrier(0);
}
template <int CTRL>
__device__ __forceinline__ uint32_t xor_stage(uint32_t v) { return v ^ dpp16<CTRL, 0u>(v); }
template <int CTRL>
__device__ __forceinline__ uint32_t or_stage(uint32_t v) { return v | dpp16<CTRL, 0u>(v); }

// RB rows reduced together, stage-major; LPF = lanes per frame (8 or 16)
template <int RB, int LPF>
__device__ __forceinline__ void row_allreduce(uint32_t (&m1)[RB], uint32_t (&m2)[RB], uint32_t (&sg)[RB]) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < RB; ++r) reduce_stage<0xB1>(m1[r], m2[r], sg[r]);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int r = 0; r < RB; ++r) reduce_stage<0x4E>(m1[r], m2[r], sg[r]);   // quad_perm [2,3,0,1]
#pragma unroll
  for (int r = 0; r < RB; ++r) reduce_stage<0x141>(m1[r], m2[r], sg[r]);  // row_half_mirror
  if constexpr (LPF == 16) {
#pragma unroll
    for (int r = 0; r < RB; ++r) reduce_stage<0x140>(m1[r], m2[r], sg[r]);  // row_mirror
  }
}
template <int LPF>
__device__ __forceinline__ uint32_t group_xor(uint32_t v) {
  v = xor_stage<0xB1>(v);
  v = xor_stage<0x4E>(v);
  v = xor_stage<0x141>(v);
  return LPF == 16 ? xor_stage<0x140>(v) : v;
}
template <int LPF>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
  v = or_stage<0xB1>(v);
  v = or_stage<0x4E>(v);
  v = or_stage<0x141>(v);
  return LPF == 16 ? or_stage<0x140>(v) : v;
}

// Reduction of FOUR rows at once inside a 16-lane DPP row (see the TRED branch of the kernel).  In: per lane the sorted
// pair (a1 <= a2, bit patterns of non-negative floats) and the sign word of rows 0..3.  Out: in the four lanes of bank b
// (lanes 4b..4b+3) the row's minimum, second minimum with multiplicity and sign parity (bit 31) of ROW b.
// One asm statement: the assembler inserts no wait states inside inline asm, so the order below keeps every register
// at least two instructions between its VALU write and a DPP read of it (the inputs of rows 0 and 2 are the oldest:
// they go first).  A DPP instruction with a bank mask leaves its destination untouched in the other banks, so each
// kept value is written by two instructions, one per half of the banks.
__device__ __forceinline__ void transposed_reduce4(const uint32_t (&a1)[4], const uint32_t (&a2)[4], const uint32_t (&sg)[4],
                                                   uint32_t &z1, uint32_t &z2, uint32_t &zs) {
  uint32_t x1, x2, xs, y1, y2, ys, h, t, h2, t2;
  asm volatile(
      // (a register copy the compiler may have placed right in front of this statement must have landed)
      "s_nop 1\n\t"
      // stage 1, lane bit 3 (partner = lane ^ 8 = row_ror:8): banks 0,1 keep rows 0 / 1, banks 2,3 keep rows 2 / 3
      "v_max_u32_dpp %[h], %[a10], %[a10] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_max_u32_dpp %[h], %[a12], %[a12] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_min_u32_dpp %[x1], %[a10], %[a10] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_min_u32_dpp %[x1], %[a12], %[a12] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_min_u32_dpp %[t], %[a20], %[a20] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_min_u32_dpp %[t], %[a22], %[a22] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_xor_b32_dpp %[xs], %[s0], %[s0] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_xor_b32_dpp %[xs], %[s2], %[s2] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_min_u32 %[x2], %[t], %[h]\n\t"
      "v_max_u32_dpp %[h2], %[a11], %[a11] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_max_u32_dpp %[h2], %[a13], %[a13] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_min_u32_dpp %[y1], %[a11], %[a11] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_min_u32_dpp %[y1], %[a13], %[a13] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_min_u32_dpp %[t2], %[a21], %[a21] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_min_u32_dpp %[t2], %[a23], %[a23] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_xor_b32_dpp %[ys], %[s1], %[s1] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_xor_b32_dpp %[ys], %[s3], %[s3] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_min_u32 %[y2], %[t2], %[h2]\n\t"
      // stage 2, lane bit 2 (partner = lane ^ 4): banks 0,2 keep x (read lane + 4), banks 1,3 keep y (read lane - 4)
      "v_max_u32_dpp %[h], %[x1], %[x1] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_min_u32_dpp %[z1], %[x1], %[x1] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_min_u32_dpp %[t], %[x2], %[x2] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_xor_b32_dpp %[zs], %[xs], %[xs] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_max_u32_dpp %[h], %[y1], %[y1] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_min_u32_dpp %[z1], %[y1], %[y1] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_min_u32_dpp %[t], %[y2], %[y2] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_xor_b32_dpp %[zs], %[ys], %[ys] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_min_u32 %[z2], %[t], %[h]\n\t"
      // stages 3 and 4, inside the quads (butterfly: every lane of a bank ends with the bank's row)
      "v_max_u32_dpp %[h], %[z1], %[z1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_min_u32_dpp %[z1], %[z1], %[z1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_xor_b32_dpp %[zs], %[zs], %[zs] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %[t], %[z2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_min3_u32 %[z2], %[h], %[z2], %[t]\n\t"
      "v_max_u32_dpp %[h], %[z1], %[z1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_min_u32_dpp %[z1], %[z1], %[z1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_xor_b32_dpp %[zs], %[zs], %[zs] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %[t], %[z2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_min3_u32 %[z2], %[h], %[z2], %[t]"
      : [x1] "=&v"(x1), [x2] "=&v"(x2), [xs] "=&v"(xs), [y1] "=&v"(y1), [y2] "=&v"(y2), [ys] "=&v"(ys), [h] "=&v"(h),
        [t] "=&v"(t), [h2] "=&v"(h2), [t2] "=&v"(t2), [z1] "=&v"(z1), [z2] "=&v"(z2), [zs] "=&v"(zs)
      : [a10] "v"(a1[0]), [a11] "v"(a1[1]), [a12] "v"(a1[2]), [a13] "v"(a1[3]), [a20] "v"(a2[0]), [a21] "v"(a2[1]),
        [a22] "v"(a2[2]), [a23] "v"(a2[3]), [s0] "v"(sg[0]), [s1] "v"(sg[1]), [s2] "v"(sg[2]), [s3] "v"(sg[3]));
}

// |.|-modified minimum / maximum / median without the canonicalisation the compiler puts in front of fminf on a
// value that comes straight from memory (the hardware instructions quiet a signalling NaN themselves)
__device__ __forceinline__ float min_abs2(float a, float b) {
  float r;
  asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_abs2(float a, float b) {
  float r;
  asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float min_abs1(float a, float q) {
  float r;
  asm("v_min_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(q));
  return r;
}
__device__ __forceinline__ float med3_abs3(float a, float b, float c) {
  float r;
  asm("v_med3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min3_abs3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float med3_abs23(float a, float b, float c) {
  float r;
  asm("v_med3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min3_abs23(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float med3_abs1(float a1, float q, float a2) {
  float r;
  asm("v_med3_f32 %0, %1, |%2|, %3" : "=v"(r) : "v"(a1), "v"(q), "v"(a2));
  return r;
}

// (acc << 1) | (t == 0.0f) in two instructions: compare into VCC, add with carry
__device__ __forceinline__ uint32_t shift_in_is_zero(uint32_t acc, float t) {
  asm("v_cmp_eq_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(t) : "vcc");
  return acc;
}

__device__ __forceinline__ int signum(float v) { return (0.0f < v) - (v < 0.0f); }
// variable-node functor of the self-correcting variants: q = fn(e + y, q_old)
template <int VARIANT>
__device__ __forceinline__ float self_correct(float tmp, float q_old) {
  if constexpr (VARIANT == CC_ALG_SCMS1) {  // soft_decision.h:261-266
    const int so = signum(q_old);
    return (so == 0 || so == signum(tmp)) ? tmp : 0.0f;
  } else {  // SCMS2, :275-280
    return (tmp * q_old > 0.0f) ? tmp : 0.5f * (tmp + q_old);
  }
}

template <int VARIANT>
__device__ __forceinline__ float horizontal(float m, float alpha_f, double beta_d) {
  if constexpr (VARIANT == CC_ALG_NMS || VARIANT == CC_ALG_2DNMS) {
    return __fmul_rn(alpha_f, m);
  } else if constexpr (VARIANT == CC_ALG_OMS) {
    const double a = static_cast<double>(m) - beta_d;
    return static_cast<float>((a < 0.0) ? 0.0 : a);
  } else {
    return m;
  }
}

// K rows, D diagonals per lane (LPF * D = row weight), RB rows per reduction batch, LPF lanes per frame (64 / LPF
// frames per wavefront), CPL columns per lane (LPF * CPL >= n), OCC = waves per SIMD the register budget targets
// SINGLE: every frame runs exactly one iteration (stop rule O0 "as shipped", or Iterations == 1): all messages
// and column sums are zero when that iteration starts, so q = (0 - 0) + y = y (y is never -0.0f) -- no {cs, r}
// operands, no subtraction / addition, and no message registers at all.
// CHAIN: the PG pairs are gap-1 links -- slots 2p (the "tail", diagonal s) and 2p + 1 (the "head", diagonal s + 1)
// of a lane.  The head's edge of row i and the tail's edge of row i + 1 sit in the SAME column s + 1 + i, and no
// other diagonal touches that column in between, so the tail takes the {cs, y} pair the head read one row earlier
// and continues the head's running column sum in registers: per link and row one {cs, y} read, one cs' read and
// one cs' write less, in the reference's summation order (soft_decision.h:88-95) all the same.
template <int K, int D, int VARIANT, int RB, int LPF, int CPL, int OCC, bool PARTIAL, typename PG = PairGaps<>,
          bool SINGLE = false, bool CHAIN = false>
__global__ void __launch_bounds__(256, OCC)
minsum_diag_kernel(MinSumParams p, const uint16_t *__restrict__ diag_s, const uint32_t *__restrict__ colbits,
                   const float *__restrict__ llr, const uint16_t *__restrict__ er, const uint32_t *__restrict__ er_off,
                   uint8_t *__restrict__ hard, float *__restrict__ Lout, uint16_t *__restrict__ iters_out,
                   int32_t *__restrict__ status_out, unsigned long long B) {
  static_assert(K % RB == 0 && K <= 32, "row batching");
  // two-pass decoding (cc_internal.hpp: MinSumParams): strategy gate and batch size decided on the device
  if (p.gate == 1 && !two_pass_sampled(p)) return;
  if (p.dual && two_pass_in_effect(p.ctl, p.sample, p.list_cap)) {  // second pass: the compacted frames
    llr = p.llr2;
    hard = p.hard2;
    Lout = nullptr;
    iters_out = p.iters2;
    status_out = p.status2;
    B = p.ctl[1] < p.list_cap ? p.ctl[1] : p.list_cap;
  }
  static_assert(LPF == 8 || LPF == 16, "a frame occupies half or all of a 16-lane DPP row");
  static_assert(2 * PG::NP <= D && !(PARTIAL && PG::NP > 0), "paired slots");
  static_assert(!CHAIN || PG::NP == 0 || (PG::gap(0) == 1 && PG::gap(PG::NP - 1) == 1), "links are gap-1 pairs");
  constexpr int NLK = CHAIN ? PG::NP : 0;  // links; slot 2p = tail, 2p + 1 = head for p < NLK
  constexpr int FPW = 64 / LPF;          // frames per wavefront
  // columns of one frame's LDS region: + 16 pad (odd frames start 16 banks later); PARTIAL geometries (row weight
  // not a multiple of LPF) append 32 scratch columns that absorb the accesses of the lanes whose last slot is empty
  constexpr int RC = LPF * CPL + (PARTIAL ? 48 : (LPF == 8 && CPL == 8 ? 8 : 16));  // n = 63: 8 measured best (683 vs 646 M frames/s)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // CY: one float2 {cs, y} per column, frame f at f * RC.  CN: one float per column at an 8-byte
  // stride, the two frames of a 32-lane half interleaved on even / odd dwords (conflict-free, and the byte
  // offset of a slot differs from its CY offset by a per-lane constant: one address register per slot).
  constexpr int CY_BYTES = FPW * RC * 8, CN_BYTES = (FPW / 2) * RC * 8;
  // SINGLE: the one iteration accumulates the new column sums in the (otherwise unused, zero) cs field of the CY cells:
  // an edge costs one 8-byte read {cs', y} and one 4-byte write instead of two reads and a write, and without the CN
  // area three workgroups fit a CU
  constexpr int WAVE_BYTES = CY_BYTES + (SINGLE ? 0 : CN_BYTES);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int fl = lane / LPF, lam = lane & (LPF - 1);
  char *cy_base = smem + wid * WAVE_BYTES;
  char *cn_lane = cy_base + CY_BYTES + ((fl >> 1) - fl) * RC * 8 + (fl & 1) * 4;  // + aCY[d] -> CN slot
  uint32_t *cbits = reinterpret_cast<uint32_t *>(smem + 4 * WAVE_BYTES);  // [256] per workgroup
  // STG[c][lane]: channel values of the frame each group decodes NEXT, written by global_load_lds (no VGPR on the
  // way) while the current frame iterates; CPL x 256 bytes per wavefront
  char *stg = smem + 4 * WAVE_BYTES + 1024 + wid * (CPL * 256);
  const uint32_t stg_lds = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<size_t>(stg)));
  cbits[threadIdx.x] = colbits[threadIdx.x];
  __shared__ unsigned wg_next;  // ordinal of the workgroup's next unassigned frame (see take_frame below)
  if (threadIdx.x == 0) wg_next = 2 * 4 * (64 / LPF);  // every lane group starts with two frames of its own
  __syncthreads();

  const int n = p.n;
  // byte offsets of this lane's D diagonals (row 0) and of its 16 owned columns
  int aCY[D];
#pragma unroll
  for (int d = 0; d < D; ++d) aCY[d] = (fl * RC + diag_s[d * LPF + lam]) * 8;
  // PARTIAL: slot D - 1 of some lanes holds no diagonal (marker 0xFFFF).  Such a lane runs the same instructions
  // on the frame's scratch columns; its magnitude is replaced by numeric_limits<float>::max() (neutral for the
  // minima, NaN-proof: v_max returns the other operand), its sign bit is masked out of the row parity.
  float pen = 0.0f;
  uint32_t sgn_keep = 0xFFFFFFFFu;
  if constexpr (PARTIAL) {
    if (diag_s[(D - 1) * LPF + lam] == 0xFFFFu) {
      aCY[D - 1] = (fl * RC + LPF * CPL + 16) * 8;
      pen = 3.402823466e+38f;
      sgn_keep = 0u;
    }
  }
  const int col0 = fl * RC + lam;  // owned columns: lam + LPF c
  // SINGLE: the frames of a 32-lane group with an odd index keep their cells the other way round, {y, cs'}: the 4-byte
  // column-sum writes of a half-wave then go to even dwords (even frame) and odd dwords (odd frame) -- 32 banks instead
  // of 16, so that what is left are the two-way conflicts inside a frame, which a ds_write_b32 hides behind its
  // register transfer (MI355X_MICROARCH.md, LDS).  Round 2 wrote both frames to the even dwords: four-way, 62 % of the
  // kernel's LDS cycles were bank conflicts (profiles/r02_pmc_minsum_single_o0_after.txt).  The price is two selects
  // per edge on a kernel whose vector unit is half idle.
  const bool oddf = SINGLE && (fl & 1);
#ifndef CC_EXP_O0_B32
  auto cell_y = [&](float2 v) -> float { return SINGLE ? (oddf ? v.x : v.y) : v.y; };
  auto cell_cs = [&](float2 v) -> float { return SINGLE ? (oddf ? v.y : v.x) : v.x; };
#else  // (E35: the halves are read by their own addresses, already sorted)
  auto cell_y = [&](float2 v) -> float { return v.y; };
  auto cell_cs = [&](float2 v) -> float { return v.x; };
#endif
  auto make_cell = [&](float cs, float y) -> float2 { return (SINGLE && oddf) ? make_float2(y, cs) : make_float2(cs, y); };
  const int y_off = (SINGLE && oddf) ? 0 : 4;  // byte offset of y inside a cell
  int aWR[SINGLE ? D : 1];                      // SINGLE: byte address of the cs' half of a slot's row-0 cell
  if constexpr (SINGLE) {
#pragma unroll
    for (int d = 0; d < D; ++d) aWR[d] = aCY[d] + (oddf ? 4 : 0);
  }

  const unsigned long long ngroups = static_cast<unsigned long long>(gridDim.x) * 4 * FPW;
  // Frames are dealt to WORKGROUPS statically (workgroup b owns the frames b GPW + u + k ngroups, u < GPW = 16 or 32
  // lane groups, k = 0, 1, ...: the same set as round 2) and to the lane groups of a workgroup DYNAMICALLY: a group that
  // has finished takes the workgroup's next frame off a counter in LDS.  With a fixed share per group the kernel ended
  // when the unluckiest of 8192 groups had run its 128 frames -- at 4 dB (1 .. 20 iterations per frame) ~9 % after the
  // average one, and a wavefront executes its row block as long as ANY of its four groups has work.
  constexpr unsigned GPW = 4 * FPW;  // lane groups per workgroup
  auto ordinal_frame = [&](unsigned o) -> unsigned long long {
    return static_cast<unsigned long long>(blockIdx.x) * GPW + (o % GPW) + static_cast<unsigned long long>(o / GPW) * ngroups;
  };
  // the calling lanes are whole groups (those that have just finished): the leader draws, the group reads its draw
  auto take_frame = [&]() -> unsigned long long {
    unsigned o = 0;
    if (lam == 0) o = atomicAdd(&wg_next, 1u);
    o = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute((lane & ~(LPF - 1)) * 4, static_cast<int>(o)));
    return ordinal_frame(o);
  };
  unsigned long long frame = (static_cast<unsigned long long>(blockIdx.x) * 4 + wid) * FPW + fl;
  unsigned long long nextf = frame + ngroups;  // staged while `frame` is being decoded
  bool active = frame < B;
  unsigned it = 0;
  // SCMS2 (soft_decision.h:273-282) needs the previous variable->check message q of every edge AS A VALUE
  // (0.5 (t + q_old)).  Keeping q next to the check->variable message r doubles the message registers (one wavefront
  // per SIMD on the headline geometry: 33 M frames/s).  But r is a function of q and of three numbers of its ROW --
  // r = +-(|q| == m1 ? m2 : m1), sign = row parity x sign(q): the two instructions of row_back -- so the kernel keeps q
  // only, parks (m1, m2 | parity) of every row in LDS (8 bytes per row and frame) and works r out again where the next
  // iteration subtracts it: 2 instructions per edge more, half the registers, the occupancy of plain min-sum (E38).
  constexpr bool NEEDQ = (VARIANT == CC_ALG_SCMS1 || VARIANT == CC_ALG_SCMS2);
  // It pays where q AND r would not fit the register budget of plain min-sum (measured over the registry's codes, E38:
  // BCH(255,231) SCMS2 33 -> 46 M frames/s, BCH(127,113) 78 -> 94; with room for both arrays the two extra
  // instructions per edge cost 10 .. 30 %), and it is what lets the two geometries with >= 196 message registers run
  // SCMS2 at all.  SCMS1 needs only two BITS of q_old and keeps them in bit words next to r (BITS1 below) -- except on
  // the headline geometry, where those words pushed 48 registers to scratch: there it keeps q too (38.6 -> 45 M).
  constexpr bool QONLY = !SINGLE && ((VARIANT == CC_ALG_SCMS2 && 2 * K * D + 64 > 256) ||
                                     (VARIANT == CC_ALG_SCMS1 && K * D >= 160 && OCC >= 2));
  constexpr bool KEEPQ = (VARIANT == CC_ALG_SCMS2 && !SINGLE) || QONLY;
  float R[(SINGLE || QONLY) ? 1 : K][(SINGLE || QONLY) ? 1 : D];
  float Q[KEEPQ ? K : 1][KEEPQ ? D : 1];
  __shared__ uint2 row_state[QONLY ? 4 * FPW * K : 1];  // [lane group of the workgroup][row] = {m1, m2 | parity << 31}
  uint2 *const my_rows = row_state + (QONLY ? (wid * FPW + fl) * K : 0);
  uint2 rsq = make_uint2(0u, 0u);  // the state of the row in hand, fetched with its operands
  // SCMS1 (soft_decision.h:261-266) needs two bits of the previous variable->check message of an edge, not the
  // message: its sign S and whether it was zero Z -- q = (Z or S == sign(t)) ? t : 0 (for t = +-0 both arms are
  // zero, so signum(t) need not be looked at).  Bit field of row i: D bits at RPW-row words, edge d at bit
  // D - 1 - d of the field (the order v_alignbit / v_addc shift them in); updated in place row by row.
  constexpr bool BITS1 = VARIANT == CC_ALG_SCMS1 && !SINGLE && !QONLY;
  constexpr int RPW = 32 / D;                  // rows per 32-bit word
  constexpr int NW = (K + RPW - 1) / RPW;      // words per lane
  uint32_t SW[BITS1 ? NW : 1], ZW[BITS1 ? NW : 1];

  // Asynchronous copy of frame f's channel values into STG, no VGPR on the way.  EXEC is the caller's: the lanes of
  // one group, or of several groups that finish in the same iteration -- so nothing here may depend on the group.
  // X4: global_load_lds_dwordx4; lane t's 16 bytes land at M0 + instruction offset + 16 t and the instruction
  // offset moves the global address too (profiles/ubench/ldsdma_probe2.hip).  Instruction c copies floats
  // [4 LPF c, 4 LPF (c + 1)) of every active group's frame into block c of STG (1 KiB, lane-linear), M0 being set
  // so that the shared offset lands there; the last instruction is shifted back to end on float n - 1 (it copies a
  // few floats twice, it never reads past the frame).  NF + 1 instructions per frame instead of CPL.  Frames
  // shorter than 4 LPF floats (n = 15, 31) keep the dword form, STG[c][lane].
  constexpr bool X4 = (LPF * CPL - 1) >= 4 * LPF;
  constexpr int NF = (LPF * CPL - 1) / (4 * LPF);  // instructions that start on a multiple of 4 LPF floats
  auto stage = [&](unsigned long long f) {
    if (f < B) {
      const float *src = llr + f * n;
      if constexpr (X4) {
        const float *v = src + 4 * lam;
        const float *vt = v + (n - 4 * LPF);  // the shifted last copy
        // one asm statement: M0 is changed behind the compiler's back and must be back before it looks again.
        // Global bytes advance by 16 LPF per instruction (the offset field), the LDS side must advance by 1024.
        uint32_t m0_save;
        const uint32_t sb = __builtin_amdgcn_readfirstlane(stg_lds);  // (uniform already; pins it to an SGPR)
        static_assert(NF == 1 || NF == 3, "x4 staging is written out for one or three aligned copies");
        if constexpr (NF == 3) {
          asm volatile(
              "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
              "s_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%7\n\t"
              "s_add_u32 m0, %3, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%8\n\t"
              "s_add_u32 m0, %3, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
              : "=&s"(m0_save)
              : "v"(v), "v"(vt), "s"(sb), "n"(1024 - LPF * 16), "n"(2048 - 2 * LPF * 16), "n"(3072), "n"(LPF * 16),
                "n"(2 * LPF * 16)
              : "memory", "scc");
        } else {
          asm volatile(
              "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
              "s_add_u32 m0, %3, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
              : "=&s"(m0_save)
              : "v"(v), "v"(vt), "s"(sb)
              : "memory", "scc");
        }
      } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int j = lam + LPF * c;
          if (j < n) {
            uint32_t m0_save;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_save) : "v"(src + j), "s"(stg_lds + c * 256) : "memory");
          }
        }
      }
    }
  };
  // channel value of column lam + LPF c of the frame staged for this group.  X4 layout: float e of the frame sits
  // 4 (e mod 4 LPF) bytes into its group's 16 LPF-byte slice of block e / (4 LPF); the floats of the shifted last
  // block are counted from n - 4 LPF.
  const char *stg_full = stg + fl * LPF * 16 + lam * 4;
  const char *stg_tail = stg + NF * 1024 + fl * LPF * 16 + (lam - n + 4 * LPF) * 4;
  auto staged = [&](int c) -> float {
    if constexpr (X4) {
      if (c < 4 * NF) return *reinterpret_cast<const float *>(stg_full + (c / 4) * 1024 + (c % 4) * 4 * LPF);
      return *reinterpret_cast<const float *>(stg_tail + 4 * LPF * c);
    } else {
      return *reinterpret_cast<const float *>(stg + c * 256 + lane * 4);
    }
  };
  // set a frame up from the staged channel values: {cs = 0, y}, all messages 0 (soft_decision.h:168-170); cs' is
  // zero already (cleared once below, and by the stop test of every iteration since).  Only the last owned column
  // can lie beyond the frame (LPF (CPL - 1) < n <= LPF CPL for every geometry, checked by the launcher).
  auto setup = [&](unsigned long long f) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      float y = staged(c) + 0.0f;  // -0.0f -> +0.0f
      if (c == CPL - 1 && lam + LPF * c >= n) y = 0.0f;
      *reinterpret_cast<float2 *>(cy_base + (col0 + LPF * c) * 8) = make_cell(0.0f, y);
    }
    if (er_off != nullptr) {  // cyclic.h:259-262
      for (uint32_t e = er_off[f]; e < er_off[f + 1]; ++e) {
        const int pos = er[e];
        if ((pos & (LPF - 1)) == lam) *reinterpret_cast<float *>(cy_base + (fl * RC + pos) * 8 + y_off) = 0.0f;
      }
    }
    if constexpr (!SINGLE && !QONLY) static_for<K>([&](auto I) { static_for<D>([&](auto Dd) { R[I][Dd] = 0.0f; }); });
    if constexpr (KEEPQ) static_for<K>([&](auto I) { static_for<D>([&](auto Dd) { Q[I][Dd] = 0.0f; }); });
    if constexpr (QONLY)
      for (int r = lam; r < K; r += LPF) my_rows[r] = make_uint2(0u, 0u);  // r_old = 0 everywhere
    if constexpr (BITS1) static_for<NW>([&](auto Wd) { SW[Wd] = 0u; ZW[Wd] = 0xFFFFFFFFu; });  // q_old = 0 everywhere
  };

  // first frame of every group: staged like all the others, then the one after it
#pragma unroll
  for (int c = 0; c < CPL; ++c)
    if constexpr (!SINGLE) *reinterpret_cast<float *>(cn_lane + (col0 + LPF * c) * 8) = 0.0f;
  if (active) {
    stage(frame);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    setup(frame);
    stage(nextf);
  }

  while (__any(active)) {
    // ---------------- one min-sum iteration for the four resident frames ----------------
    // software pipeline over rows: the {cs, y} operands of row i+1 are fetched from LDS before row i is
    // reduced and scattered, so their latency hides behind the DPP chain and the read-modify-writes
    // (SCMS1 on a geometry that already fills the register file fetches a row's operands when the row starts:
    //  the bit words below take the 14 registers the prefetch would hold)
    // (SINGLE: the cs' half of a cell is written by the row before, so a row's operands are fetched when it starts)
    constexpr bool PREFETCH = !(BITS1 && K * D >= 160) && !SINGLE;
    // four rows per (transposed) reduction, see below: bit-exact and 180 instructions per iteration shorter, but
    // measured 1.4 % SLOWER than the butterfly (profiles/r03_experiments.md, E19: the reduction was what covered the LDS
    // latency of each row's operands; without it the schedule needs four counted waits per row) -- experiments only
#ifdef CC_EXP_TRED
    constexpr bool TRED = LPF == 16 && K % 4 == 0 && !SINGLE;
#else
    constexpr bool TRED = false;
#endif
    constexpr bool EARLY_PF = TRED && PREFETCH;
    float2 cyq[D];
    float2 carry_cy[NLK ? NLK : 1];  // the head's operands of the row before: the tail's operands of this row
    float carry_sum[NLK ? NLK : 1];  // the head's running column sum of the row before
    auto fetch = [&](auto IC) {
      constexpr int row = decltype(IC)::value;
      if constexpr (QONLY) rsq = my_rows[row];
      static_for<D>([&](auto DD) {
        constexpr int d = DD;
        if constexpr (d < 2 * NLK && (d & 1) == 0 && row >= 1) {
          cyq[d] = carry_cy[d / 2];
          return;
        }
        // (SINGLE uses y only, but reads the {cs, y} pair all the same: a 4-byte read of the 8-byte cells puts the
        //  two frames of a half-wave on the same 16 of 32 banks -- SQ_LDS_BANK_CONFLICT 48 % of the LDS cycles --
        //  while the 8-byte read spreads them over 64)
#ifdef CC_EXP_NO_CYR
        asm volatile("" : "=v"(cyq[d].x), "=v"(cyq[d].y));
#else
#ifdef CC_EXP_O0_B32
        if constexpr (SINGLE) {
          // E35 (measured 4 % SLOWER, experiments only): the two halves of the cell by their own per-lane addresses --
          // two ds_read_b32 move the same bytes as one ds_read_b64 and save the two selects per edge that sort the
          // halves out, 20 % of the kernel's VALU instructions; the LDS instructions they add cost more
          const int at = aCY[slot_base<PG>(d)] + 8 * (row + slot_gap<PG>(d));
          cyq[d].x = *reinterpret_cast<const float *>(cy_base + at + (4 - y_off));
          cyq[d].y = *reinterpret_cast<const float *>(cy_base + at + y_off);
          return;
        }
#endif
        cyq[d] = *reinterpret_cast<const float2 *>(cy_base + aCY[slot_base<PG>(d)] + 8 * (row + slot_gap<PG>(d)));
#endif
      });
    };
    float Tloc[D];  // SINGLE: q, then r, of the row in hand (nothing is kept across rows)
    // ---- a row in three parts, so that the reduction between them can serve one row (butterfly) or four (transposed) ----
    // front: operands, q = (cs - r) + y, the lane's two smallest magnitudes and its sign parity
    auto row_front = [&](auto IR, uint32_t &o1, uint32_t &o2, uint32_t &os) {
      constexpr int i = decltype(IR)::value;
      if constexpr (!PREFETCH) fetch(IR);
      // q, then r, of this row's edges: the message registers, or temporaries when nothing is kept
      // (accessors with literal indices: a reference to R[i] would keep the whole array out of registers)
      [[maybe_unused]] auto wget = [&](auto DD) -> float {
        if constexpr (SINGLE) return Tloc[DD];
        else if constexpr (QONLY) return Q[i][DD];
        else return R[i][DD];
      };
      auto wset = [&](auto DD, float v) {
        if constexpr (SINGLE) Tloc[DD] = v;
        else if constexpr (!QONLY) R[i][DD] = v;  // (QONLY: q already sits in Q[i][d])
      };
      // QONLY: last iteration's r of this row from its q and the row's parked (m1, m2, parity), exactly as row_back made it
      [[maybe_unused]] const float old_hi = u2f(rsq.y & 0x7FFFFFFFu);
      [[maybe_unused]] uint32_t old_Y = (rsq.x ^ rsq.y);  // (m1 ^ m2) | parity << 31: m1 has a clear sign bit
      if constexpr (QONLY) asm volatile("" : "+v"(old_Y));
      {
        // with one diagonal per lane the lane's second minimum is "none": numeric_limits<float>::max(), the
        // starting value of the reference's own search (soft_decision.h:110); with D >= 2 both are overwritten
        float a1 = 0.0f, a2 = (D == 1) ? 3.402823466e+38f : 0.0f;
        uint32_t s = 0;
        uint32_t qs[D];
        float mag[D], qv[D];
        // SCMS1: all t = e + y of the row first, their sign / zero bits shifted into two words, ONE three-input bit
        // operation for the keep decision of the whole row, then an arithmetic bit-field extract + AND per edge
        uint32_t keep = 0;
        constexpr int wi = BITS1 ? i / RPW : 0, sh = BITS1 ? (i % RPW) * D : 0;
        if constexpr (BITS1) {
          uint32_t tw = 0, tz = 0;
          static_for<D>([&](auto DD) {
            constexpr int d = DD;
            const float t = (cyq[d].x - R[i][d]) + cyq[d].y;  // soft_decision.h:135-136
            R[i][d] = t;
            tw = __builtin_amdgcn_alignbit(tw, f2u(t), 31);  // (tw << 1) | sign(t)
            tz = shift_in_is_zero(tz, t);
          });
          constexpr uint32_t field = ((D == 32 ? 0u : (1u << D)) - 1u) << sh;
          const uint32_t tws = tw << sh, tzs = tz << sh;
          keep = ZW[wi] | ~(SW[wi] ^ tws);                       // Z or signs agree (bits outside the field: unused)
          SW[wi] = (SW[wi] & ~field) | (tws & keep);             // sign of the new q (a dropped q is +0)
          ZW[wi] = (ZW[wi] & ~field) | ((~keep & field) | tzs);  // new q is zero: dropped, or t itself was
        }
        static_for<D>([&](auto DD) {
          constexpr int d = DD;
          float q;
          if constexpr (BITS1) {
            const int32_t m = __builtin_amdgcn_sbfe(static_cast<int32_t>(keep), sh + D - 1 - d, 1);  // 0 or ~0
            q = u2f(f2u(R[i][d]) & static_cast<uint32_t>(m));
          } else if constexpr (SINGLE) {
            q = cell_y(cyq[d]);
            if constexpr (NEEDQ) q = self_correct<VARIANT>(q, 0.0f);
          } else {
            float r_old;
            if constexpr (QONLY) r_old = u2f(f2u(__builtin_amdgcn_fmed3f(Q[i][d], -old_hi, old_hi)) ^ old_Y);
            else r_old = R[i][d];
            float e = cyq[d].x - r_old;                                         // soft_decision.h:135
            if constexpr (VARIANT == CC_ALG_2DNMS) e = __fmul_rn(p.beta_f, e);  // :215-218
            q = e + cyq[d].y;                                                   // :136,:207-209
            if constexpr (QONLY && VARIANT == CC_ALG_SCMS1) {
              // soft_decision.h:261-266 without compares: keep t iff q_old is zero or their sign bits agree.  Neither t
              // nor a kept q is ever -0.0f here (t = e + y with y canonicalised on load; a dropped q is +0.0f), so
              // "q_old == 0" is "its bits are 0" and signum(t) = 0 lands on the right arm by its sign bit alone.
              const uint32_t so = f2u(Q[i][d]), st = f2u(q);
              const int differ = static_cast<int>(so ^ st) >> 31;   // all ones: the sign bits differ
              const uint32_t zero_old = umin32(so, 1u) - 1u;        // all ones: q_old == 0
              q = u2f(static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(static_cast<int>(st), static_cast<int>(zero_old), differ,
                                                                        0xD0)));  // t & (zero_old | ~differ)
              Q[i][d] = q;
            } else if constexpr (KEEPQ) {
              q = self_correct<VARIANT>(q, Q[i][d]);
              Q[i][d] = q;
            }
          }
          wset(DD, q);
          qv[d] = q;
        });
        // TRED: nothing hides the operands' latency between two fronts (the reduction comes after the fourth), so the
        // next row's operands are requested as soon as this row's have been consumed -- the pair tracking and the sign
        // parity below (~12 instructions) run while they travel
        if constexpr (EARLY_PF) {
          static_for<NLK>([&](auto P) { carry_cy[P] = cyq[2 * P + 1]; });
          // (one row's requests per barrier: merged across rows they become ds_read2_b64, which costs the LDS twice
          //  the cycles of two ds_read_b64 -- profiles/r02_experiments.md E3 -- and keeps four rows of operands live)
          asm volatile("" ::: "memory");
          if constexpr (i + 1 < K) fetch(std::integral_constant<int, i + 1>{});
          asm volatile("" ::: "memory");
        }
        static_for<D>([&](auto DD) {
          constexpr int d = DD;
          const float q = qv[d];
          float a = __builtin_fabsf(q);
          uint32_t qb = f2u(q);
          if constexpr (PARTIAL && d == D - 1) {
            a = __builtin_fmaxf(a, pen);
            qb &= sgn_keep;
          }
          qs[d] = qb;
          // The lane's two smallest magnitudes.  Values enter three at a time, then two at a time: min3 / med3 give the
          // two smallest of three; two more values x, y against the sorted pair (a1 <= a2): a1' = min3(a1, x, y) and
          // a2' = min(a2, med3(a1, x, y)) -- the smallest of the four is the smallest of (a1, x, y), the runner-up the
          // smaller of that triple's median and a2.  8 instructions for 7 values (one value at a time: 12).
          // (|.| is a source modifier on all of them; a loaded / masked q goes through the asm forms, see min_abs2.)
          constexpr bool RAW = (SINGLE || BITS1) && !PARTIAL && D >= 2;
          mag[d] = RAW ? q : a;
          if constexpr (D == 2 && d == 1) {
            a2 = RAW ? max_abs2(mag[0], mag[1]) : __builtin_fmaxf(mag[0], mag[1]);
            a1 = RAW ? min_abs2(mag[0], mag[1]) : __builtin_fminf(mag[0], mag[1]);
          } else if constexpr (d == 0 && D == 1) {
            a1 = a;
          } else if constexpr (d == 2) {
            a2 = RAW ? med3_abs3(mag[0], mag[1], mag[2]) : __builtin_amdgcn_fmed3f(mag[0], mag[1], mag[2]);
            a1 = RAW ? min3_abs3(mag[0], mag[1], mag[2]) : __builtin_fminf(__builtin_fminf(mag[0], mag[1]), mag[2]);
          } else if constexpr (d >= 4 && (d & 1) == 0) {
            const float md = RAW ? med3_abs23(a1, mag[d - 1], mag[d]) : __builtin_amdgcn_fmed3f(a1, mag[d - 1], mag[d]);
            a1 = RAW ? min3_abs23(a1, mag[d - 1], mag[d]) : __builtin_fminf(__builtin_fminf(a1, mag[d - 1]), mag[d]);
            a2 = __builtin_fminf(a2, md);
          } else if constexpr (d == D - 1 && d >= 3 && (d & 1) == 1) {  // one value left over
            a2 = RAW ? med3_abs1(a1, mag[d], a2) : __builtin_amdgcn_fmed3f(a1, mag[d], a2);
            a1 = RAW ? min_abs1(a1, mag[d]) : __builtin_fminf(a1, mag[d]);
          }
        });
        // sign parity: three-input XORs (v_bitop3_b32), half the instructions of a chain of v_xor_b32
        s = qs[0];
        static_for<(D - 1) / 2>([&](auto T) {
          s = static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(static_cast<int>(s), static_cast<int>(qs[1 + 2 * T]),
                                                                static_cast<int>(qs[2 + 2 * T]), 0x96));  // a ^ b ^ c
        });
        if constexpr ((D - 1) % 2) s ^= qs[D - 1];
        o1 = f2u(a1);
        o2 = f2u(a2);
        os = s;
      }
      if constexpr (!EARLY_PF) {
        static_for<NLK>([&](auto P) { carry_cy[P] = cyq[2 * P + 1]; });
        if constexpr (PREFETCH && i + 1 < K) fetch(std::integral_constant<int, i + 1>{});  // the next row's operands
      }
    };
    // column sums accumulated so far, requested before the reduction so that their latency hides behind it
    auto row_cn = [&](auto IR, float (&cn)[D]) {
      constexpr int i = decltype(IR)::value;
      static_for<D>([&](auto DD) {
        constexpr int d = DD;
        if constexpr (d < 2 * NLK && (d & 1) == 0 && i >= 1)
          cn[d] = carry_sum[d / 2];  // the head added to this column in row i - 1 and kept the sum
        else if constexpr (SINGLE)
          cn[d] = cell_cs(cyq[d]);
        else
#ifdef CC_EXP_NO_CNR
          asm volatile("" : "=v"(cn[d]));
#else
          cn[d] = *reinterpret_cast<const float *>(cn_lane + aCY[slot_base<PG>(d)] + 8 * (i + slot_gap<PG>(d)));
#endif
      });
    };
    // back: r from the row's minima and parity, then the column sums (soft_decision.h:101-122, :86-98)
    auto row_back = [&](auto IR, uint32_t m1v, uint32_t m2v, uint32_t sg0, float (&cn)[D]) {
      constexpr int i = decltype(IR)::value;
      float rt[QONLY ? D : 1];  // QONLY: this row's r, used for the column sums and dropped
      bool made = false;        // (wget returns q until the row's r has been made)
      (void)made;
      [[maybe_unused]] auto wget = [&](auto DD) -> float {
        if constexpr (SINGLE) return Tloc[DD];
        else if constexpr (QONLY) return made ? rt[DD] : Q[i][DD];
        else return R[i][DD];
      };
      auto wset = [&](auto DD, float v) {
        if constexpr (SINGLE) Tloc[DD] = v;
        else if constexpr (QONLY) rt[DD] = v;
        else R[i][DD] = v;
      };
      // the parity leaves the last DPP stage in a register of its own: folded into the mask below, the compiler
      // undoes the DPP form of that stage (a bit operation with three inputs takes no DPP operand)
      asm volatile("" : "+v"(sg0));
      const uint32_t sign31 = sg0 & 0x80000000u;
      if constexpr (VARIANT == CC_ALG_MS || VARIANT == CC_ALG_SCMS1 || VARIANT == CC_ALG_SCMS2) {  // h(m) = m
        // exclusive minimum with the exclusive sign in TWO instructions per edge: u = med3(q, -m2, m2) is
        // sign(q) min(|q|, m2), i.e. +-m1 for the holder of the minimum and +-m2 for everybody else (|q| >= m2);
        // XOR with m1 ^ m2 swaps the two magnitudes, XOR with the row parity (bit 31) turns sign(q) into the
        // product of the OTHER signs.  A zero keeps its sign bit through med3, as it did through the old
        // (t ^ Y) + signbit(q) form; with m1 = m2 = 0 the result is +-0 either way.
        uint32_t Y = (m1v ^ m2v) | sign31;
#ifndef CC_EXP_BITOP3_POST
        // Y in a register of its own: otherwise the compiler folds the OR into every edge's XOR (v_bitop3_b32 with three
        // inputs, a half-rate VOP3 encoding) where a plain v_xor_b32 issues at the full rate
        asm volatile("" : "+v"(Y));
#endif
        const float hi = u2f(m2v);
        static_for<D>([&](auto DD) {
          wset(DD, u2f(f2u(__builtin_amdgcn_fmed3f(wget(DD), -hi, hi)) ^ Y));
        });
      } else {
        const uint32_t H1 = f2u(horizontal<VARIANT>(u2f(m1v), p.alpha_f, p.beta_d));
        const uint32_t H2 = f2u(horizontal<VARIANT>(u2f(m2v), p.alpha_f, p.beta_d));
        static_for<D>([&](auto DD) {
          const uint32_t mag = (__builtin_fabsf(wget(DD)) == u2f(m1v)) ? H2 : H1;
          wset(DD, u2f(xad(mag, sign31, f2u(wget(DD)) & 0x80000000u)));
        });
      }
      made = true;
      if constexpr (QONLY) {  // park the row's minima and parity for the next iteration's front (one lane per frame writes)
        if (lam == 0) my_rows[i] = make_uint2(m1v, m2v | sign31);
      }
      // column sums: all D additions behind ONE wait (the last-issued read first: once it has arrived the others
      // have too, LDS returns in order), then the D stores -- instead of a wait in front of every addition
      float sum[D];
      static_for<D>([&](auto DD) {
        constexpr int d = D - 1 - DD;
        sum[d] = cn[d] + wget(std::integral_constant<int, d>{});  // ascending rows
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<D>([&](auto DD) {
        constexpr int d = DD;
        if constexpr (d < 2 * NLK && (d & 1) == 1 && i + 1 < K)
          carry_sum[d / 2] = sum[d];  // the tail of this link continues it in row i + 1
        else if constexpr (SINGLE)
          *reinterpret_cast<float *>(cy_base + aWR[slot_base<PG>(d)] + 8 * (i + slot_gap<PG>(d))) = sum[d];
        else
#ifdef CC_EXP_NO_CNW
          asm volatile("" ::"v"(sum[d]));
#else
          *reinterpret_cast<float *>(cn_lane + aCY[slot_base<PG>(d)] + 8 * (i + slot_gap<PG>(d))) = sum[d];
#endif
      });
      // The next row reads column sums that OTHER lanes have just written (column s + i is diagonal s - 1 of
      // row i + 1).  The hardware keeps LDS operations of a wavefront in order; the compiler must too: per
      // lane the two addresses differ by a constant, so without this barrier it may hoist the next row's read
      // above this row's write (it did for D = 1, where nothing else sits between them).
      asm volatile("" ::: "memory");
    };
    if constexpr (PREFETCH) fetch(std::integral_constant<int, 0>{});
    if constexpr (TRED) {
      // Four rows per reduction (LPF = 16, K a multiple of four).  A butterfly all-reduce leaves every lane with every
      // row's result at (4 + 1) instructions x 4 stages per row; here the first two stages HALVE the data instead:
      // across lane bit 3 the lanes of banks 0, 1 keep rows 0, 1 of the group and hand rows 2, 3 to their partners
      // in banks 2, 3 (and the other way round), across lane bit 2 the same again, so that bank b ends up with row b
      // alone -- DPP bank masks select the writing lanes, so "keep mine, take yours" costs no select instruction
      // (7 + 2 instructions per row PAIR and stage) -- and only two butterfly stages inside the quads remain, on one
      // row per lane: 37 instructions per four rows instead of 80.  ds_swizzle hands a bank's result to the other
      // three (12 per group, LDS crossbar, no memory access).
      static_for<K / 4>([&](auto IG) {
        constexpr int g = IG;
        uint32_t a1[4], a2[4], sg[4];
        static_for<4>([&](auto J) { row_front(std::integral_constant<int, 4 * g + J>{}, a1[J], a2[J], sg[J]); });
        uint32_t z1, z2, zs;
        transposed_reduce4(a1, a2, sg, z1, z2, zs);
        uint32_t M1[4], M2[4], SG[4];
        static_for<4>([&](auto J) {
          constexpr int pat = 0x13 | (J << 7);  // bit mode: lane' = (lane & 0x13) | (J << 2): bank J of the own 16-lane row
          M1[J] = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(z1), pat));
          M2[J] = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(z2), pat));
          SG[J] = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(zs), pat));
        });
        static_for<4>([&](auto J) {
          float cn[D];
          row_cn(std::integral_constant<int, 4 * g + J>{}, cn);
          row_back(std::integral_constant<int, 4 * g + J>{}, M1[J], M2[J], SG[J], cn);
        });
      });
    } else {
      static_for<K>([&](auto IR) {
        uint32_t m1[1], m2[1], sg[1];
        row_front(IR, m1[0], m2[0], sg[0]);
        float cn[D];
        row_cn(IR, cn);
#ifndef CC_EXP_NO_REDUCE
        row_allreduce<1, LPF>(m1, m2, sg);
#endif
        row_back(IR, m1[0], m2[0], sg[0], cn);
      });
    }

    // ---------------- a-posteriori values, stop test (soft_decision.h:178-186) ----------------
    // acc: XOR of the check masks of the columns whose hard decision is 1 (the frame's syndrome, GF(2) rule O2) or
    // their OR (rule O1: only the all-zero word passes).  Only the last owned column can lie beyond the frame.
    uint32_t acc = 0;
    // All reads first, then the arithmetic, then the writes: written column by column (read, test, write) the compiler
    // must keep every column's reads behind the previous column's writes -- it cannot tell the addresses apart -- and
    // the scan pays one LDS round trip per column, 16 per iteration (round 2: 8 % of the kernel's time).  The check
    // masks come along unconditionally (one v_cndmask per column instead of a read under EXEC = hard decision).
    auto stop_scan = [&](auto ORC) {
      float cnv[CPL], yc[CPL];
      uint32_t cb[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        if constexpr (SINGLE) {
          const float2 cy = *reinterpret_cast<const float2 *>(cy_base + (col0 + LPF * c) * 8);  // {cs', y}
          cnv[c] = cy.x;
          yc[c] = cy.y;
        } else {
          cnv[c] = *reinterpret_cast<const float *>(cn_lane + (col0 + LPF * c) * 8);
          yc[c] = *reinterpret_cast<const float *>(cy_base + (col0 + LPF * c) * 8 + 4);
        }
        cb[c] = cbits[lam + LPF * c];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        // L = cs + y :180-182, b = L < 0 codes.h:51.  L is never -0.0f (cs' is a sum that started at +0.0f, y was
        // canonicalised on load), so b is L's sign bit: spread it over the word (one shift) and fold mask and
        // accumulation into one three-input bit operation -- no compare, no select, no VCC.  Columns beyond the frame
        // (only the last owned one can be) have an all-zero check mask.
        const int m = static_cast<int>(f2u(cnv[c] + yc[c])) >> 31;
        if constexpr (decltype(ORC)::value)
          acc = static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(static_cast<int>(acc), m, static_cast<int>(cb[c]), 0xF8));  // a | (b & c)
        else
          acc = static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(static_cast<int>(acc), m, static_cast<int>(cb[c]), 0x78));  // a ^ (b & c)
      }
      // next iteration: cs := cs', cs' := 0 (SINGLE: there is none, and cs' already sits in the cell)
      if constexpr (!SINGLE) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          *reinterpret_cast<float *>(cy_base + (col0 + LPF * c) * 8) = cnv[c];
          *reinterpret_cast<float *>(cn_lane + (col0 + LPF * c) * 8) = 0.0f;
        }
      }
    };
    bool ok;
    if (SINGLE && p.stop_rule == CC_STOP_AS_SHIPPED) {
      ok = true;  // no test at all, and nothing to move
    } else if (p.stop_rule == CC_STOP_PUBLISHED) {
      stop_scan(std::true_type{});
      ok = group_or<LPF>(acc) == 0;
    } else {
      stop_scan(std::false_type{});
      ok = (p.stop_rule == CC_STOP_AS_SHIPPED) || group_xor<LPF>(acc) == 0;
    }
#ifdef CC_EXP_NEVER_STOP
    ok = false;
#endif
    const bool finished = SINGLE || ok || (it + 1 >= p.iterations);
    if (finished && active) {
      const unsigned long long done = frame;
      const unsigned done_it = ok ? it : p.iterations;  // (SINGLE without convergence: Iterations == 1)
      frame = nextf;
      active = frame < B;
      uint8_t *hp = hard + done * n + lam;
      float *Lp = Lout ? Lout + done * n + lam : nullptr;
      // column by column: result of the decided frame out of LDS, the next frame's channel value in, result to
      // HBM -- nothing is held in registers across columns.  The wait comes before the first store: the next frame
      // was staged a whole frame ago and nothing else of this wave is in flight, so it is free.
      // first pass of two: a frame that has not stopped is handed on (nothing of it is written here)
      const bool handed = SINGLE && p.first_pass != 0 && !ok;
      auto emit = [&](int c, float L) {
        if (handed) return;
        if (c < CPL - 1 || lam + LPF * c < n) {
          hp[LPF * c] = (L < 0.0f) ? 1 : 0;
          if (Lp) Lp[LPF * c] = L;
        }
      };
      if (active) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const float2 cy = *reinterpret_cast<const float2 *>(cy_base + (col0 + LPF * c) * 8);  // {cs (new), y}
          float y = staged(c) + 0.0f;  // -0.0f -> +0.0f
          if (c == CPL - 1 && lam + LPF * c >= n) y = 0.0f;
          *reinterpret_cast<float2 *>(cy_base + (col0 + LPF * c) * 8) = make_cell(0.0f, y);
          emit(c, cy.x + cy.y);
        }
        if (er_off != nullptr) {  // cyclic.h:259-262
          for (uint32_t e = er_off[frame]; e < er_off[frame + 1]; ++e) {
            const int pos = er[e];
            if ((pos & (LPF - 1)) == lam) *reinterpret_cast<float *>(cy_base + (fl * RC + pos) * 8 + y_off) = 0.0f;
          }
        }
        if constexpr (!SINGLE && !QONLY) static_for<K>([&](auto I) { static_for<D>([&](auto Dd) { R[I][Dd] = 0.0f; }); });
        if constexpr (KEEPQ) static_for<K>([&](auto I) { static_for<D>([&](auto Dd) { Q[I][Dd] = 0.0f; }); });
        if constexpr (QONLY)
          for (int r = lam; r < K; r += LPF) my_rows[r] = make_uint2(0u, 0u);
        if constexpr (BITS1) static_for<NW>([&](auto Wd) { SW[Wd] = 0u; ZW[Wd] = 0xFFFFFFFFu; });
        it = 0;
      } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const float2 cy = *reinterpret_cast<const float2 *>(cy_base + (col0 + LPF * c) * 8);
          emit(c, cy.x + cy.y);
        }
      }
      if (lam == 0) {
        if (handed) {
          if (p.first_pass == 2) {
            atomicAdd(&p.ctl[2], 1u);
          } else {
            const uint32_t at = atomicAdd(&p.ctl[1], 1u);
            if (at < p.list_cap) p.list[at] = static_cast<uint32_t>(done);
            else p.ctl[3] = 1u;
          }
        } else {
          if (iters_out) iters_out[done] = static_cast<uint16_t>(done_it);
          if (status_out) status_out[done] = ok ? CC_FRAME_OK : CC_FRAME_NOT_CONVERGED;
        }
      }
      if (active) {  // after the reads of STG above have been consumed
        nextf = take_frame();
        stage(nextf);
      }
    } else {
      ++it;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS write-back may outlive the workgroup's allocation
}

}  // namespace

namespace {

// PARTS / PART: the variants of a geometry are spread over PARTS objects so that the big geometries build in parallel
// (diag_variant_part in cc_internal.hpp; minsum_diag.hip dispatches); this object instantiates the variants of PART
template <int K, int D, int LPF, int CPL, int OCC, bool SCMS = false, bool PARTIAL = false, typename PG = PairGaps<>,
          bool CHAIN = false, int PARTS = 1, int PART = 0>
int launch_diag_geometry(const cc_code *code, const MinSumParams &p, const float *d_llr, const uint16_t *d_er,
                         const uint32_t *d_er_off, uint8_t *d_hard, float *d_L, uint16_t *d_iters, int32_t *d_status,
                         size_t B, hipStream_t stream) {
  constexpr int RB = 1;  // rows per reduction batch (the row-pipelined body reduces one row at a time)
  // SCMS2 carries K D more registers per lane (q next to r) and its own occupancy target -- unless that would leave a
  // single wave per SIMD: then the kernel keeps q INSTEAD of r (QONLY in the kernel: r is worked out again from q and
  // the row's parked minima) and runs at the occupancy of plain min-sum, plus 8 bytes of LDS per row and frame
  constexpr bool Q_ONLY_2 = 2 * K * D + 64 > 256, Q_ONLY_1 = K * D >= 160 && OCC >= 2;
  constexpr int OCC_S = Q_ONLY_2 ? OCC : (2 * K * D + 64 <= 128) ? 4 : (2 * K * D + 64 <= 168) ? 3 : 2;
  constexpr int FPW = 64 / LPF;
  if (!(code->tab.n > static_cast<unsigned>(LPF * (CPL - 1)) && code->tab.n <= static_cast<unsigned>(LPF * CPL))) {
    set_last_error("minsum_diag: only the last owned column of a lane may lie beyond the frame");
    return CC_ERR_UNSUPPORTED;
  }
  const DiagGeometry g{0, 0, PARTIAL ? 1u : static_cast<unsigned>(LPF * D), D, LPF, CPL, SCMS};
  const size_t lds = minsum_diag_lds_bytes(g);
  const unsigned long long blocks_needed = (B + 4 * FPW - 1) / (4 * FPW);
  unsigned long long per_cu = (160 * 1024) / lds;  // resident workgroups: LDS, then the register budget (OCC waves per SIMD)
  const unsigned long long occ = p.variant == CC_ALG_SCMS2 ? static_cast<unsigned long long>(OCC_S) : OCC;
  if ((p.variant == CC_ALG_SCMS2 && Q_ONLY_2) || (p.variant == CC_ALG_SCMS1 && Q_ONLY_1))
    per_cu = (160 * 1024) / (lds + 4 * FPW * K * sizeof(uint2));  // + row_state (static)
  if (per_cu > occ) per_cu = occ;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * per_cu;
  const int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
  const unsigned long long Bq = B;
  hipError_t e = hipSuccess;
  const bool single = p.stop_rule == CC_STOP_AS_SHIPPED || p.iterations == 1 || p.first_pass != 0;
  // the message-free kernel has no CN area (see SINGLE in the kernel) and few registers: LDS alone caps its occupancy
  constexpr size_t rc_cols = static_cast<size_t>(LPF) * CPL + (PARTIAL ? 48 : (LPF == 8 && CPL == 8 ? 8 : 16));
  const size_t lds_s = lds - 4 * (FPW / 2) * rc_cols * 8;
  unsigned long long per_cu_s = (160 * 1024) / lds_s;
  if (per_cu_s > 8) per_cu_s = 8;
  const unsigned long long max_grid_s = static_cast<unsigned long long>(code->num_cus) * per_cu_s;
  const int grid_s = static_cast<int>(blocks_needed < max_grid_s ? blocks_needed : max_grid_s);
#define CC_LAUNCH_S(V, O, S)                                                                                       \
  {                                                                                                                \
    const size_t lds_k = (S) ? lds_s : lds;                                                                        \
    const int grid_k = (S) ? grid_s : grid;                                                                        \
    e = hipFuncSetAttribute(                                                                                       \
        reinterpret_cast<const void *>(&minsum_diag_kernel<K, D, V, RB, LPF, CPL, O, PARTIAL, PG, S, CHAIN>),      \
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_k));                                     \
    if (e == hipSuccess)                                                                                           \
      hipLaunchKernelGGL((minsum_diag_kernel<K, D, V, RB, LPF, CPL, O, PARTIAL, PG, S, CHAIN>), dim3(grid_k), dim3(256), lds_k, \
                         stream, p, code->d_diag, code->d_colbits, d_llr, d_er, d_er_off, d_hard, d_L, d_iters,    \
                         d_status, Bq);                                                                            \
  }
  // one iteration per frame by construction (stop rule O0, SURVEY F1, or Iterations == 1): the message-free kernel
#define CC_LAUNCH(V, O)                                                                                            \
  {                                                                                                                \
    if constexpr (diag_variant_part(V, PARTS) != PART) e = hipErrorInvalidValue;                                   \
    else if (single) CC_LAUNCH_S(V, OCC, true) else CC_LAUNCH_S(V, O, false)                                       \
  }
  switch (p.variant) {
    case CC_ALG_MS: CC_LAUNCH(CC_ALG_MS, OCC) break;
    case CC_ALG_NMS: CC_LAUNCH(CC_ALG_NMS, OCC) break;
    case CC_ALG_OMS: CC_LAUNCH(CC_ALG_OMS, OCC) break;
    case CC_ALG_2DNMS: CC_LAUNCH(CC_ALG_2DNMS, OCC) break;
    case CC_ALG_SCMS1:
      CC_LAUNCH(CC_ALG_SCMS1, OCC)
      break;
    case CC_ALG_SCMS2:
      if constexpr (SCMS) CC_LAUNCH(CC_ALG_SCMS2, OCC_S) else e = hipErrorInvalidValue;
      break;
    default: e = hipErrorInvalidValue;
  }
#undef CC_LAUNCH
#undef CC_LAUNCH_S
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "minsum_diag kernel launch");
  return CC_OK;
}

}  // namespace

}  // namespace ccamd
