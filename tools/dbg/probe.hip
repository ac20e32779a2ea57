// Co-residency probe: a kernel that spins for a number of clock ticks with a chosen workgroup size, LDS footprint and
// register footprint.  hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/dbg/libprobe.so tools/dbg/probe.hip
#include <hip/hip_runtime.h>
template <int V>
__global__ void probe_kernel(long ticks, int *sink) {
  extern __shared__ int lds[];
  if (V == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
  if (V == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  if (V == 88) asm volatile("v_mov_b32 v87, 0" ::: "v87");
  if (V == 1) asm volatile("s_mov_b32 s100, 0" ::: "s100");
  long t0 = wall_clock64();
  int acc = 0;
  while (wall_clock64() - t0 < ticks) acc += 1;   // wall clock: 100 MHz
  if (acc == -1) { lds[threadIdx.x] = acc; sink[0] = lds[0]; }
}
extern "C" int probe_launch(int variant, int grid, int threads, int lds_bytes, long ticks, int *sink, void *stream) {
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, (hipStream_t)stream, ticks, sink);
  };
  switch (variant) {
    case 72: go(probe_kernel<72>); break;
    case 80: go(probe_kernel<80>); break;
    case 88: go(probe_kernel<88>); break;
    case 1: go(probe_kernel<1>); break;
    default: go(probe_kernel<0>); break;
  }
  return (int)hipGetLastError();
}
