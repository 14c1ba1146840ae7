// Probe 2 (gfx950): global_load_lds_dwordx4 -- where do the 16 bytes of lane t land, does the instruction offset
// shift the LDS side too, does an LDS base that is only 4-byte aligned work, and what happens under a partial EXEC.
//   test 0: all 64 lanes, M0 = base, offset 0          -> expect LDS float index 4 t + k = source index
//   test 1: offset:256 on the instruction              -> global + 256 B; LDS + 256 B ?
//   test 2: M0 = base + 4 (4-byte aligned only)        -> lands at float index 1 + 4 t + k ?
//   test 3: EXEC = lanes 16..31 only                   -> only [64, 128) written, at 4 t + k ?
//   test 4: global address 4-byte aligned only (p + 1) -> works ?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *p, float *o, int test) {
  __shared__ __attribute__((aligned(16))) float buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = -1.0f;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)buf + (test == 2 ? 4u : 0u);
  const float *src = p + 4 * threadIdx.x + (test == 4 ? 1 : 0);
  unsigned save;
  if (test == 1) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:256\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(src), "s"(base) : "memory");
  } else if (test == 3) {
    if (threadIdx.x >= 16 && threadIdx.x < 32)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(save) : "v"(src), "s"(base) : "memory");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(src), "s"(base) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) o[i] = buf[i];
}
int main() {
  std::vector<float> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 1024 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int test = 0; test < 5; ++test) {
    k<<<1, 64>>>(d, o, test);
    std::vector<float> r(1024);
    if (hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("test %d: copy failed\n", test); return 1; }
    int first = -1, last = -1, count = 0;
    for (int i = 0; i < 1024; ++i) if (r[i] >= 0) { if (first < 0) first = i; last = i; ++count; }
    printf("test %d: %d floats written, LDS index range [%d, %d]; LDS[first]=%g LDS[first+1]=%g LDS[first+4]=%g LDS[last]=%g\n",
           test, count, first, last, first >= 0 ? r[first] : -1, first >= 0 ? r[first + 1] : -1, first >= 0 ? r[first + 4] : -1,
           last >= 0 ? r[last] : -1);
  }
  return 0;
}
