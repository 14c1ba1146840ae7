// occupancy_probe.hip -- how many workgroups of 512 threads with N KB of dynamic LDS does a gfx950 CU hold?
// hipcc --offload-arch=gfx950 -O2 occupancy_probe.hip -o occupancy_probe && ./occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512, 4) probe(float *out) {
  extern __shared__ float s[];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = s[(threadIdx.x + 1) & 511];
}
int main() {
  for (int kb : {16, 32, 48, 64, 65, 72, 80, 96, 128, 160}) {
    int n = -1;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, probe, 512, static_cast<size_t>(kb) * 1024);
    printf("%3d KB dynamic LDS, 512 threads: %d workgroups per CU (%s)\n", kb, n, hipGetErrorString(e));
  }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  regsPerBlock %d\n", p.sharedMemPerBlock,
         p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
  return 0;
}
