// Probe: global -> LDS direct load (no VGPR destination) on gfx950, M0 saved and restored inside the asm block.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *p, float *o) {
  __shared__ float junk[64];
  const unsigned base = (unsigned)(size_t)junk;
  unsigned save;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(save) : "v"(p + threadIdx.x), "s"(base) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  o[threadIdx.x] = junk[threadIdx.x];
}
int main() {
  float h[64], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = 100.0f + i;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof h);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  float r[64]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) bad += (r[i] != h[i]);
  printf("lds-dma probe: %d mismatches (r[0]=%g r[63]=%g)\n", bad, r[0], r[63]);
  return bad != 0;
}
