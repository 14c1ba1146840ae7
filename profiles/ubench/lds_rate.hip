// LDS microbenchmark for the round-3 layout questions (gfx950): does a ds_read_b64 / ds_write_b64 at an address that
// is 4 (mod 8) work and what does it cost; ds_read_b128, ds_write_b64, ds_write2_b32 against the two-instruction
// forms; 2-way conflicts.  Prints cycles per wave instruction per CU (all four SIMDs issuing, W waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
typedef float f4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ void __launch_bounds__(256) ub(float *out, int iters, int stride_b, int off_b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<float *>(smem)[i] = i;
  __syncthreads();
  uint32_t a = static_cast<uint32_t>(reinterpret_cast<size_t>(smem)) + wid * 4096 + lane * stride_b + off_b;
  float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  float2 v2 = make_float2(1.0f, 2.0f);
  f4 v4 = {1, 2, 3, 4};
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {
      float2 t;
      asm volatile(REP8("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory");
      r0 += t.x; r1 += t.y;
    } else if constexpr (OP == 1) {
      float t;
      asm volatile(REP8("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory");
      r0 += t;
    } else if constexpr (OP == 2) {
      f4 t;
      asm volatile(REP8("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory");
      r0 += t.x; r1 += t.w;
    } else if constexpr (OP == 3) {
      asm volatile(REP8("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v2.x) : "memory");
    } else if constexpr (OP == 4) {
      asm volatile(REP8("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v2) : "memory");
    } else if constexpr (OP == 5) {
      asm volatile(REP8("ds_write2_b32 %0, %1, %2 offset1:1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v2.x), "v"(v2.y) : "memory");
    } else if constexpr (OP == 6) {
      asm volatile(REP8("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v4) : "memory");
    } else if constexpr (OP == 7) {
      float2 t;
      asm volatile(REP8("ds_read2_b32 %0, %1 offset1:1\n") "s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory");
      r0 += t.x; r1 += t.y;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3;
}
__global__ void check_misaligned(float *out) {
  __shared__ __attribute__((aligned(16))) float s[256];
  for (int i = threadIdx.x; i < 256; i += 64) s[i] = i;
  __syncthreads();
  uint32_t a = static_cast<uint32_t>(reinterpret_cast<size_t>(s)) + threadIdx.x * 8 + 4;  // 4 (mod 8)
  float2 t;
  asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory");
  out[2 * threadIdx.x] = t.x; out[2 * threadIdx.x + 1] = t.y;
  float2 w = make_float2(1000.0f + threadIdx.x, 2000.0f + threadIdx.x);
  __syncthreads();
  if (threadIdx.x < 16) asm volatile("ds_write_b64 %0, %1\ns_waitcnt lgkmcnt(0)" :: "v"(a + 512), "v"(w) : "memory");
  __syncthreads();
  out[128 + threadIdx.x] = s[128 + threadIdx.x];
}
template <int OP> void run(const char *name, int stride_b, int off_b) {
  float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const int iters = 2000;
  printf("%-52s", name);
  for (int w : {1, 2, 4}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    ub<OP><<<256 * w, 256, 16384>>>(out, 10, stride_b, off_b);
    hipDeviceSynchronize();
    hipEventRecord(a); ub<OP><<<256 * w, 256, 16384>>>(out, iters, stride_b, off_b); hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: w workgroups x 4 waves x iters x 8 instructions
    printf("  w=%d: %6.2f", w, ms * 1e-3 * 2.4e9 / (double(w) * 4 * iters * 8));
  }
  printf("   [cycles @2.4GHz per wave-instr per CU]\n");
  hipFree(out);
}
int main() {
  float *o; hipMalloc(&o, 1024 * 4); float h[256];
  check_misaligned<<<1, 64>>>(o); hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) bad += (h[2 * l] != 2 * l + 1) || (h[2 * l + 1] != 2 * l + 2);
  printf("ds_read_b64 at 4 (mod 8): %s (lane 0 got %g %g, lane 5 got %g %g)\n", bad ? "WRONG" : "correct", h[0], h[1], h[10], h[11]);
  printf("ds_write_b64 at 4 (mod 8): s[129..132] = %g %g %g %g (expect 1000 2000 1001 2001)\n", h[128 + 1], h[128 + 2], h[128 + 3], h[128 + 4]);
  run<0>("ds_read_b64 stride 8 aligned", 8, 0);
  run<0>("ds_read_b64 stride 8 at 4 (mod 8)", 8, 4);
  run<0>("ds_read_b64 stride 16 (2-way)", 16, 0);
  run<1>("ds_read_b32 stride 4", 4, 0);
  run<1>("ds_read_b32 stride 8 (2-way)", 8, 0);
  run<2>("ds_read_b128 stride 16", 16, 0);
  run<7>("ds_read2_b32 stride 8", 8, 0);
  run<3>("ds_write_b32 stride 4", 4, 0);
  run<3>("ds_write_b32 stride 8 (2-way)", 8, 0);
  run<3>("ds_write_b32 stride 16 (4-way)", 16, 0);
  run<4>("ds_write_b64 stride 8 aligned", 8, 0);
  run<4>("ds_write_b64 stride 8 at 4 (mod 8)", 8, 4);
  run<5>("ds_write2_b32 stride 8", 8, 0);
  run<6>("ds_write_b128 stride 16", 16, 0);
  return 0;
}
