// Microbenchmark: sustained wave64 instruction rate per SIMD on gfx950 for the
// instruction kinds the min-sum kernels use, at 1/2/3/4/8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) ub(float *out, int iters, uint64_t m0, uint64_t m1) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {  // independent v_add_f32
      asm volatile(REP8("v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %2,%2,%8\n v_add_f32 %3,%3,%8\n v_add_f32 %4,%4,%8\n v_add_f32 %5,%5,%8\n v_add_f32 %6,%6,%8\n v_add_f32 %7,%7,%8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 1) {  // v_med3_f32 with abs modifier
      asm volatile(REP8("v_med3_f32 %0,%0,|%8|,%1\n v_med3_f32 %1,%1,|%8|,%2\n v_med3_f32 %2,%2,|%8|,%3\n v_med3_f32 %3,%3,|%8|,%4\n v_med3_f32 %4,%4,|%8|,%5\n v_med3_f32 %5,%5,|%8|,%6\n v_med3_f32 %6,%6,|%8|,%7\n v_med3_f32 %7,%7,|%8|,%0\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 2) {  // v_min_u32_dpp quad_perm, independent
      asm volatile(REP8("v_min_u32_dpp %0,%0,%0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1,%1,%1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %2,%2,%2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3,%3,%3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %4,%4,%4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %5,%5,%5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %6,%6,%6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %7,%7,%7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 3) {  // dependent v_add_f32 chain
      asm volatile(REP8("v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 4) {  // exec switch + 1 VALU (masked-op pattern)
      asm volatile(REP8("s_mov_b64 exec,%9\n v_add_f32 %0,%0,%8\n s_mov_b64 exec,%10\n v_add_f32 %1,%1,%8\n s_mov_b64 exec,%9\n v_add_f32 %2,%2,%8\n s_mov_b64 exec,%10\n v_add_f32 %3,%3,%8\n") "s_mov_b64 exec,-1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m0), "s"(m1));
    } else if constexpr (OP == 6) {  // v_xor_b32 / v_and / v_xad mix (integer VALU)
      asm volatile(REP8("v_xor_b32 %0,%0,%8\n v_and_b32 %1,%1,%8\n v_xad_u32 %2,%2,%8,%3\n v_xor_b32 %3,%3,%8\n v_min_f32_e64 %4,%4,|%8|\n v_sub_f32 %5,%8,%5\n v_xor_b32 %6,%6,%8\n v_and_b32 %7,%7,%8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 7) {  // pure SALU
      asm volatile(REP8("s_xor_b64 %0,%0,%1\n s_xor_b64 %0,%0,%1\n s_xor_b64 %0,%0,%1\n s_xor_b64 %0,%0,%1\n s_xor_b64 %0,%0,%1\n s_xor_b64 %0,%0,%1\n s_xor_b64 %0,%0,%1\n s_xor_b64 %0,%0,%1\n")
                   : "+s"(m0) : "s"(m1) : "scc");
    } else if constexpr (OP == 8) {  // VALU + SALU alternating (independent)
      asm volatile(REP8("v_add_f32 %0,%0,%[b]\n s_xor_b64 %[m],%[m],%[n]\n v_add_f32 %1,%1,%[b]\n s_xor_b64 %[m],%[m],%[n]\n v_add_f32 %2,%2,%[b]\n s_xor_b64 %[m],%[m],%[n]\n v_add_f32 %3,%3,%[b]\n s_xor_b64 %[m],%[m],%[n]\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [m] "+s"(m0) : [b] "v"(b), [n] "s"(m1) : "scc");
    } else if constexpr (OP == 9) {  // ds_add_f32 (LDS float atomic), distinct addresses per lane
      asm volatile(REP8("ds_add_f32 %0,%1\n ds_add_f32 %0,%1 offset:256\n ds_add_f32 %0,%1 offset:512\n ds_add_f32 %0,%1 offset:768\n ds_add_f32 %0,%1 offset:1024\n ds_add_f32 %0,%1 offset:1280\n ds_add_f32 %0,%1 offset:1536\n ds_add_f32 %0,%1 offset:1792\n") "s_waitcnt lgkmcnt(0)\n"
                   : : "v"((threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 2048), "v"(b) : "memory");
    } else if constexpr (OP == 10) {  // ds_read_b32 conflict-free
      asm volatile(REP8("ds_read_b32 %0,%8\n ds_read_b32 %1,%8 offset:256\n ds_read_b32 %2,%8 offset:512\n ds_read_b32 %3,%8 offset:768\n ds_read_b32 %4,%8 offset:1024\n ds_read_b32 %5,%8 offset:1280\n ds_read_b32 %6,%8 offset:1536\n ds_read_b32 %7,%8 offset:1792\n") "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"((threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 2048) : "memory");
    }
  }
  __shared__ float dummy[2048 * 4 / 4 + 16];
  if (iters < 0) dummy[threadIdx.x] = a0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)m0 + dummy[0];
}
template <int OP> void run(const char *name, int instr_per_iter) {
  float *out; hipMalloc(&out, 256 * 64 * 256 * sizeof(float));
  const int iters = 2000;
  printf("%-38s", name);
  for (int w : {1, 2, 3, 4, 8}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    ub<OP><<<256 * w, 256>>>(out, 10, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);
    hipDeviceSynchronize();
    hipEventRecord(a); ub<OP><<<256 * w, 256>>>(out, iters, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull); hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * instr_per_iter * w);
    printf("  w=%d: %5.2f", w, cyc);
  }
  printf("   [cycles @2.4GHz per wave-instr per SIMD]\n");
  hipFree(out);
}
int main() {
  run<0>("v_add_f32 independent", 64);
  run<3>("v_add_f32 dependent chain", 64);
  run<1>("v_med3_f32 |abs| (ring dep)", 64);
  run<2>("v_min_u32_dpp quad_perm indep", 64);
  run<6>("int/f32 VALU mix", 64);
  run<4>("s_mov exec + v_add (pair=2 instr)", 64 + 1);
  run<7>("s_xor_b64 dependent", 64);
  run<8>("v_add + s_xor alternating", 64);
  run<9>("ds_add_f32 no-conflict", 64);
  run<10>("ds_read_b32 no-conflict", 64);
  return 0;
}
