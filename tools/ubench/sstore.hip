// Micro-benchmark: ways to get 64-bit wave ballots (exceedance words) to memory on gfx950.
//   0  v_writelane_b32 x2 per ballot into lane w of a register pair, one coalesced 8-byte-per-lane store per 32 ballots
//   1  s_store_dwordx2 per ballot (scalar store through the scalar data cache, s_dcache_wb at the end)
//   2  s_store_dwordx4 per two ballots
// Every wave produces kRows rows of 32 ballots from data it holds in registers; prints ns per ballot and CU and checks the
// scalar-store results against the vector path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kRows = 256;

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long *out, const float *x) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  float xr[32];
#pragma unroll
  for (int w = 0; w < 32; ++w) xr[w] = x[(w * 64 + lane) & 4095];
  unsigned long long *row = out + size_t(wave) * kRows * 32;
  for (int r = 0; r < kRows; ++r) {
    const float thr = float(r) * (1.0f / kRows);
    if constexpr (KIND == 0) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int w = 0; w < 32; ++w) {
        const unsigned long long m = __ballot(xr[w] > thr);
        asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                     : "+v"(lo), "+v"(hi) : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "i"(w));
      }
      if (lane < 32) row[r * 32 + lane] = ((unsigned long long)hi << 32) | lo;
    } else if constexpr (KIND == 1) {
      unsigned long long *p = row + r * 32;
#pragma unroll
      for (int w = 0; w < 32; ++w) {
        const unsigned long long m = __ballot(xr[w] > thr);
        asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(m), "s"(p), "i"(w * 8) : "memory");
      }
    } else {
      unsigned long long *p = row + r * 32;
#pragma unroll
      for (int w = 0; w < 32; w += 2) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 mm;
        mm.x = __ballot(xr[w] > thr);
        mm.y = __ballot(xr[w + 1] > thr);
        asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(mm), "s"(p), "i"(w * 8) : "memory");
      }
    }
  }
  if constexpr (KIND != 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}

int main() {
  const int grid = 256 * 8, waves = grid * 4;
  const size_t n = size_t(waves) * kRows * 32;
  unsigned long long *o[3];
  float *x;
  std::vector<float> hx(4096);
  for (int i = 0; i < 4096; ++i) hx[i] = float((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
  CK(hipMalloc(&x, 4096 * 4));
  CK(hipMemcpy(x, hx.data(), 4096 * 4, hipMemcpyHostToDevice));
  for (int kd = 0; kd < 3; ++kd) { CK(hipMalloc(&o[kd], n * 8)); CK(hipMemset(o[kd], 0xee, n * 8)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (int kd = 0; kd < 3; ++kd) {
      CK(hipEventRecord(e0));
      if (kd == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, o[0], x);
      if (kd == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, o[1], x);
      if (kd == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, o[2], x);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("kind %d: %.3f ms for %zu ballots: %.2f ns per ballot and CU, %.1f GB/s written\n", kd, ms, n,
                      ms * 1e6 / (double(n) / 256), n * 8 / ms * 1e-6);
    }
  std::vector<unsigned long long> h0(n), h1(n);
  CK(hipMemcpy(h0.data(), o[0], n * 8, hipMemcpyDeviceToHost));
  for (int kd = 1; kd < 3; ++kd) {
    CK(hipMemcpy(h1.data(), o[kd], n * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += h0[i] != h1[i];
    printf("kind %d vs vector path: %zu of %zu words differ\n", kd, bad, n);
  }
  return 0;
}
