// Third microbenchmark (gfx950): packed f32 VALU and two-address LDS operations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) ub(float *out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double b = 1.0001;
  float f0 = threadIdx.x, f1 = 1.5f;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {  // v_pk_add_f32, 8 independent chains
      asm volatile(REP8("v_pk_add_f32 %0,%0,%8\n v_pk_add_f32 %1,%1,%8\n v_pk_add_f32 %2,%2,%8\n v_pk_add_f32 %3,%3,%8\n v_pk_add_f32 %4,%4,%8\n v_pk_add_f32 %5,%5,%8\n v_pk_add_f32 %6,%6,%8\n v_pk_add_f32 %7,%7,%8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 1) {  // v_pk_mul_f32
      asm volatile(REP8("v_pk_mul_f32 %0,%0,%8\n v_pk_mul_f32 %1,%1,%8\n v_pk_mul_f32 %2,%2,%8\n v_pk_mul_f32 %3,%3,%8\n v_pk_mul_f32 %4,%4,%8\n v_pk_mul_f32 %5,%5,%8\n v_pk_mul_f32 %6,%6,%8\n v_pk_mul_f32 %7,%7,%8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (OP == 2) {  // ds_read2_b32 conflict-free
      asm volatile(REP8("ds_read2_b32 %0,%4 offset0:0 offset1:64\n ds_read2_b32 %1,%4 offset0:128 offset1:192\n ds_read2_b32 %2,%4 offset0:1 offset1:65\n ds_read2_b32 %3,%4 offset0:129 offset1:193\n ds_read2_b32 %0,%4 offset0:2 offset1:66\n ds_read2_b32 %1,%4 offset0:130 offset1:194\n ds_read2_b32 %2,%4 offset0:3 offset1:67\n ds_read2_b32 %3,%4 offset0:131 offset1:195\n") "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"((threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096) : "memory");
    } else if constexpr (OP == 3) {  // ds_write2_b32 conflict-free
      asm volatile(REP8("ds_write2_b32 %0,%1,%2 offset0:0 offset1:64\n ds_write2_b32 %0,%1,%2 offset0:128 offset1:192\n ds_write2_b32 %0,%1,%2 offset0:1 offset1:65\n ds_write2_b32 %0,%1,%2 offset0:129 offset1:193\n ds_write2_b32 %0,%1,%2 offset0:2 offset1:66\n ds_write2_b32 %0,%1,%2 offset0:130 offset1:194\n ds_write2_b32 %0,%1,%2 offset0:3 offset1:67\n ds_write2_b32 %0,%1,%2 offset0:131 offset1:195\n") "s_waitcnt lgkmcnt(0)\n"
                   : : "v"((threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096), "v"(f0), "v"(f1) : "memory");
    } else if constexpr (OP == 4) {  // ds_read_u8 gather, 64 pseudo-random bytes of a 510-byte table
      uint32_t r0, r1, r2, r3;
      asm volatile(REP8("ds_read_u8 %0,%4\n ds_read_u8 %1,%4 offset:1\n ds_read_u8 %2,%4 offset:2\n ds_read_u8 %3,%4 offset:3\n ds_read_u8 %0,%4 offset:4\n ds_read_u8 %1,%4 offset:5\n ds_read_u8 %2,%4 offset:6\n ds_read_u8 %3,%4 offset:7\n") "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(((threadIdx.x * 2654435761u) >> 7) % 500u) : "memory");
      f0 += r0;
    } else if constexpr (OP == 5) {  // ds_read_u8 gather confined to 252 bytes (one dword per bank)
      uint32_t r0, r1, r2, r3;
      asm volatile(REP8("ds_read_u8 %0,%4\n ds_read_u8 %1,%4 offset:1\n ds_read_u8 %2,%4 offset:2\n ds_read_u8 %3,%4 offset:3\n ds_read_u8 %0,%4 offset:0\n ds_read_u8 %1,%4 offset:1\n ds_read_u8 %2,%4 offset:2\n ds_read_u8 %3,%4 offset:3\n") "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(((threadIdx.x * 2654435761u) >> 7) % 252u) : "memory");
      f0 += r0;
    }
  }
  __shared__ float dummy[4096 * 4 / 4 + 1024];
  if (iters < 0) dummy[threadIdx.x] = f0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + f0 + dummy[0];
}
template <int OP> void run(const char *name, int instr_per_iter) {
  float *out; hipMalloc(&out, 256 * 64 * 256 * sizeof(float));
  const int iters = 2000;
  printf("%-46s", name);
  for (int w : {1, 2, 4, 8}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    ub<OP><<<256 * w, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a); ub<OP><<<256 * w, 256>>>(out, iters); hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  w=%d: %6.2f", w, ms * 1e-3 * 2.4e9 / ((double)iters * instr_per_iter * w));
  }
  printf("   [cycles @2.4GHz per wave-instr per SIMD]\n");
  hipFree(out);
}
int main() {
  run<0>("v_pk_add_f32 (2 flops per lane)", 64);
  run<1>("v_pk_mul_f32", 64);
  run<2>("ds_read2_b32 no-conflict", 64);
  run<3>("ds_write2_b32 no-conflict", 64);
  run<4>("ds_read_u8 random gather, 500-byte table", 64);
  run<5>("ds_read_u8 random gather, 252-byte table", 64);
  return 0;
}
