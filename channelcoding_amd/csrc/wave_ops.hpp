// wave_ops.hpp -- wave64 cross-lane helpers for gfx950 built on DPP (no LDS traffic).
// A DPP-fused VALU op runs at half rate (~4.5 cycles per wave instruction per SIMD, measured:
// profiles/r01_ubench_instruction_rates.txt), a dependent DPP pair additionally needs two wait
// states, so the *_batch forms interleave N independent chains stage by stage.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ccamd {

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// one DPP stage of a wave-wide unsigned-min reduction; old = identity so the
// combiner may fold the move into v_min_u32_dpp also for partial row masks
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umin(uint32_t v) {
  return umin32(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(0xFFFFFFFFu), static_cast<int>(v),
                                                                      CTRL, ROW_MASK, 0xF, false)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_xor(uint32_t v) {
  return v ^ static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or(uint32_t v) {
  return v | static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xF, false));
}
// N independent wave-wide minima, stage-major so that consecutive DPP ops never depend on each other
// (a dependent DPP pair costs two wait states); results valid in lane 63
template <int N>
__device__ __forceinline__ void wave_umin_batch(uint32_t (&v)[N]) {
  // sched_barrier(0) pins the stage-major order; without it the scheduler re-serialises each chain
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_umin<0xB1, 0xF>(v[i]);  // quad_perm [1,0,3,2]
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_umin<0x4E, 0xF>(v[i]);  // quad_perm [2,3,0,1]
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_umin<0x141, 0xF>(v[i]);  // row_half_mirror
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_umin<0x140, 0xF>(v[i]);  // row_mirror
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_umin<0x142, 0xA>(v[i]);  // row_bcast:15
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_umin<0x143, 0xC>(v[i]);  // row_bcast:31
  __builtin_amdgcn_sched_barrier(0);
}
// result valid in lane 63
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) {
  v = dpp_umin<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_umin<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_umin<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_umin<0x140, 0xF>(v);  // row_mirror
  v = dpp_umin<0x142, 0xA>(v);  // row_bcast:15
  v = dpp_umin<0x143, 0xC>(v);  // row_bcast:31
  return v;
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
  v = dpp_xor<0xB1, 0xF>(v);
  v = dpp_xor<0x4E, 0xF>(v);
  v = dpp_xor<0x141, 0xF>(v);
  v = dpp_xor<0x140, 0xF>(v);
  v = dpp_xor<0x142, 0xA>(v);
  v = dpp_xor<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
  v = dpp_or<0xB1, 0xF>(v);
  v = dpp_or<0x4E, 0xF>(v);
  v = dpp_or<0x141, 0xF>(v);
  v = dpp_or<0x140, 0xF>(v);
  v = dpp_or<0x142, 0xA>(v);
  v = dpp_or<0x143, 0xC>(v);
  return v;
}


}  // namespace ccamd
