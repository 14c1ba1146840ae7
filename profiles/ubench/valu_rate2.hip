// Second microbenchmark: rates of the candidate ops for the reduction rewrite (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) ub(float *out, int iters, uint64_t m0, uint64_t m1) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f;
  uint32_t s0 = 0;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {  // v_permlane32_swap on 4 independent pairs
      asm volatile(REP8("v_permlane32_swap_b32 %0,%1\n v_permlane32_swap_b32 %2,%3\n v_permlane32_swap_b32 %4,%5\n v_permlane32_swap_b32 %6,%7\n v_permlane32_swap_b32 %0,%1\n v_permlane32_swap_b32 %2,%3\n v_permlane32_swap_b32 %4,%5\n v_permlane32_swap_b32 %6,%7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (OP == 1) {  // v_permlane16_swap
      asm volatile(REP8("v_permlane16_swap_b32 %0,%1\n v_permlane16_swap_b32 %2,%3\n v_permlane16_swap_b32 %4,%5\n v_permlane16_swap_b32 %6,%7\n v_permlane16_swap_b32 %0,%1\n v_permlane16_swap_b32 %2,%3\n v_permlane16_swap_b32 %4,%5\n v_permlane16_swap_b32 %6,%7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (OP == 2) {  // v_readlane_b32 to distinct sgprs (through one output)
      asm volatile(REP8("v_readlane_b32 %8,%0,63\n v_readlane_b32 %8,%1,63\n v_readlane_b32 %8,%2,63\n v_readlane_b32 %8,%3,63\n v_readlane_b32 %8,%4,63\n v_readlane_b32 %8,%5,63\n v_readlane_b32 %8,%6,63\n v_readlane_b32 %8,%7,63\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(s0));
    } else if constexpr (OP == 3) {  // v_cndmask with sgpr-pair mask
      asm volatile(REP8("v_cndmask_b32 %0,%0,%8,%9\n v_cndmask_b32 %1,%1,%8,%10\n v_cndmask_b32 %2,%2,%8,%9\n v_cndmask_b32 %3,%3,%8,%10\n v_cndmask_b32 %4,%4,%8,%9\n v_cndmask_b32 %5,%5,%8,%10\n v_cndmask_b32 %6,%6,%8,%9\n v_cndmask_b32 %7,%7,%8,%10\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m0), "s"(m1));
    } else if constexpr (OP == 4) {  // v_min_u32 plain (VOP2, full rate?)
      asm volatile(REP8("v_min_u32 %0,%0,%8\n v_min_u32 %1,%1,%8\n v_min_u32 %2,%2,%8\n v_min_u32 %3,%3,%8\n v_min_u32 %4,%4,%8\n v_min_u32 %5,%5,%8\n v_min_u32 %6,%6,%8\n v_min_u32 %7,%7,%8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 5) {  // v_min_u32_dpp row_shr (not quad_perm) 
      asm volatile(REP8("v_min_u32_dpp %0,%0,%0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1,%1,%1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %2,%2,%2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3,%3,%3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %4,%4,%4 row_mirror row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %5,%5,%5 row_mirror row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %6,%6,%6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_min_u32_dpp %7,%7,%7 row_bcast:31 row_mask:0xc bank_mask:0xf\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (OP == 6) {  // v_mov_b32_dpp (data move only)
      asm volatile(REP8("v_mov_b32_dpp %0,%1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2,%3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4,%5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6,%7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1,%0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3,%2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5,%4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7,%6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (OP == 7) {  // v_xad_u32 / v_and_or / 3-operand int
      asm volatile(REP8("v_xad_u32 %0,%0,%8,%1\n v_xad_u32 %1,%1,%8,%2\n v_xad_u32 %2,%2,%8,%3\n v_xad_u32 %3,%3,%8,%4\n v_xad_u32 %4,%4,%8,%5\n v_xad_u32 %5,%5,%8,%6\n v_xad_u32 %6,%6,%8,%7\n v_xad_u32 %7,%7,%8,%0\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 8) {  // v_min_f32 e64 with abs (VOP3 2-operand)
      asm volatile(REP8("v_min_f32_e64 %0,%0,|%8|\n v_min_f32_e64 %1,%1,|%8|\n v_min_f32_e64 %2,%2,|%8|\n v_min_f32_e64 %3,%3,|%8|\n v_min_f32_e64 %4,%4,|%8|\n v_min_f32_e64 %5,%5,|%8|\n v_min_f32_e64 %6,%6,|%8|\n v_min_f32_e64 %7,%7,|%8|\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 9) {  // v_cmp_lt_f32 -> vcc (VOPC e32)
      asm volatile(REP8("v_cmp_lt_f32 vcc,%0,%8\n v_cmp_lt_f32 vcc,%1,%8\n v_cmp_lt_f32 vcc,%2,%8\n v_cmp_lt_f32 vcc,%3,%8\n v_cmp_lt_f32 vcc,%4,%8\n v_cmp_lt_f32 vcc,%5,%8\n v_cmp_lt_f32 vcc,%6,%8\n v_cmp_lt_f32 vcc,%7,%8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    } else if constexpr (OP == 10) {  // ds_write_b32 conflict-free
      asm volatile(REP8("ds_write_b32 %0,%1\n ds_write_b32 %0,%1 offset:256\n ds_write_b32 %0,%1 offset:512\n ds_write_b32 %0,%1 offset:768\n ds_write_b32 %0,%1 offset:1024\n ds_write_b32 %0,%1 offset:1280\n ds_write_b32 %0,%1 offset:1536\n ds_write_b32 %0,%1 offset:1792\n") "s_waitcnt lgkmcnt(0)\n"
                   : : "v"((threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 2048), "v"(b) : "memory");
    } else if constexpr (OP == 11) {  // ds_read_b64 conflict-free
      double d0, d1, d2, d3;
      asm volatile(REP8("ds_read_b64 %0,%4\n ds_read_b64 %1,%4 offset:512\n ds_read_b64 %2,%4 offset:1024\n ds_read_b64 %3,%4 offset:1536\n ds_read_b64 %0,%4 offset:2048\n ds_read_b64 %1,%4 offset:2560\n ds_read_b64 %2,%4 offset:3072\n ds_read_b64 %3,%4 offset:3584\n") "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"((threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 4096) : "memory");
      a0 += (float)d0;
    }
  }
  __shared__ float dummy[4096 * 4 / 4 + 16];
  if (iters < 0) dummy[threadIdx.x] = a0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)m0 + dummy[0] + (float)s0;
}
template <int OP> void run(const char *name, int instr_per_iter) {
  float *out; hipMalloc(&out, 256 * 64 * 256 * sizeof(float));
  const int iters = 2000;
  printf("%-38s", name);
  for (int w : {1, 2, 4, 8}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    ub<OP><<<256 * w, 256>>>(out, 10, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);
    hipDeviceSynchronize();
    hipEventRecord(a); ub<OP><<<256 * w, 256>>>(out, iters, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull); hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  w=%d: %6.2f", w, ms * 1e-3 * 2.4e9 / ((double)iters * instr_per_iter * w));
  }
  printf("   [cycles @2.4GHz per wave-instr per SIMD]\n");
  hipFree(out);
}
int main() {
  run<0>("v_permlane32_swap", 64);
  run<1>("v_permlane16_swap", 64);
  run<2>("v_readlane_b32", 64);
  run<3>("v_cndmask_b32 sgpr mask (VOP3)", 64);
  run<4>("v_min_u32 (VOP2)", 64);
  run<5>("v_min_u32_dpp row_ror/mirror/bcast", 64);
  run<6>("v_mov_b32_dpp quad_perm", 64);
  run<7>("v_xad_u32 (3 operands)", 64);
  run<8>("v_min_f32_e64 |abs| (VOP3, 2 operands)", 64);
  run<9>("v_cmp_lt_f32 vcc", 64);
  run<10>("ds_write_b32 no-conflict", 64);
  run<11>("ds_read_b64 no-conflict", 64);
  return 0;
}
