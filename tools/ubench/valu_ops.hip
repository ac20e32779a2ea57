// Which gfx950 vector instructions issue at 2 cycles per wave64 instruction and SIMD (with >= 2 waves per SIMD) and
// which at 4?  One line per instruction: cycles per instruction and SIMD at 2 waves per SIMD, workgroup span measured
// with the in-kernel clock.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIter = 10000;
#define KBODY(NAME, ASM, ...)                                                                        \
  __global__ void NAME(float *out, unsigned long long *clk) {                                        \
    float v[16];                                                                                     \
    for (int i = 0; i < 16; ++i) v[i] = float(threadIdx.x * 16 + i) * 1.0001f;                       \
    unsigned long long t0 = __builtin_readcyclecounter();                                            \
    for (int it = 0; it < kIter; ++it) {                                                             \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]) __VA_ARGS__); \
    }                                                                                                \
    unsigned long long t1 = __builtin_readcyclecounter();                                            \
    float s = 0;                                                                                     \
    for (int i = 0; i < 16; ++i) s += v[i];                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                  \
    if ((threadIdx.x & 63) == 0) { clk[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; clk[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; } \
  }
KBODY(k_add_f32, "v_add_f32 %0, %0, %1")
KBODY(k_mul_f32, "v_mul_f32 %0, %0, %1")
KBODY(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KBODY(k_sub_u32, "v_sub_u32 %0, %0, %1")
KBODY(k_xor, "v_xor_b32 %0, %0, %1")
KBODY(k_or, "v_or_b32 %0, %0, %1")
KBODY(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KBODY(k_ashr, "v_ashrrev_i32 %0, 1, %0")
KBODY(k_mov, "v_mov_b32 %0, %1")
KBODY(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", : "vcc")
KBODY(k_cmp_f32, "v_cmp_gt_f32 vcc, %0, %1", : "vcc")
KBODY(k_cmp_i32, "v_cmp_gt_i32 vcc, %0, %1", : "vcc")
KBODY(k_cmp_u32, "v_cmp_gt_u32 vcc, %0, %1", : "vcc")
KBODY(k_cmp_sgpr, "v_cmp_gt_f32 s[20:21], %0, %1", : "s20", "s21")
KBODY(k_sub_co, "v_sub_co_u32 %0, vcc, %0, %1", : "vcc")
KBODY(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc", : "vcc")
KBODY(k_min_f32, "v_min_f32 %0, %0, %1")
KBODY(k_max3, "v_max3_f32 %0, %0, %1, %2")
KBODY(k_add3, "v_add3_u32 %0, %0, %1, %2")
KBODY(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KBODY(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KBODY(k_perm, "v_perm_b32 %0, %0, %1, %2")
KBODY(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KBODY(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KBODY(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KBODY(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KBODY(k_ffbl, "v_ffbl_b32 %0, %1")
KBODY(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KBODY(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KBODY(k_cvt, "v_cvt_f32_i32 %0, %1")
KBODY(k_sad, "v_sad_u32 %0, %0, %1, %2")
KBODY(k_sub_f32, "v_sub_f32 %0, %0, %1")
KBODY(k_max_i16, "v_max_i16 %0, %0, %1")
KBODY(k_min3_i32, "v_min3_i32 %0, %0, %1, %2")
KBODY(k_readlane, "v_readlane_b32 s20, %1, 3\n\tv_mov_b32 %0, s20", : "s20")
KBODY(k_dpp_add, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KBODY(k_dpp_add_qp, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KBODY(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")

typedef void (*kfn)(float *, unsigned long long *);
void run(const char *name, kfn f, int ipi = 16) {
  static float *out = nullptr; static unsigned long long *clk = nullptr;
  if (!out) { CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&clk, 256 * 32 * 8)); }
  double res[3];
  int k = 0;
  for (int threads : {256, 512, 1024}) {
    const int blocks = 256;
    f<<<blocks, threads>>>(out, clk);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 32);
    CK(hipMemcpy(h.data(), clk, blocks * 32 * 8, hipMemcpyDeviceToHost));
    const int nw = threads / 64;
    double mean = 0;
    for (int b = 0; b < blocks; ++b) {
      unsigned long long lo = ~0ull, hi = 0;
      for (int w = 0; w < nw; ++w) { lo = std::min(lo, h[(b * 16 + w) * 2]); hi = std::max(hi, h[(b * 16 + w) * 2 + 1]); }
      mean += double(hi - lo);
    }
    mean /= blocks;
    res[k++] = mean / (double(kIter) * ipi * (threads / 256));
  }
  printf("%-22s cyc/instr/SIMD at 1/2/4 waves per SIMD: %.2f %.2f %.2f\n", name, res[0], res[1], res[2]);
}
#define R(n) run(#n, n)
int main() {
  R(k_add_f32); R(k_mul_f32); R(k_fma_f32); R(k_sub_f32); R(k_sub_u32); R(k_xor); R(k_or); R(k_lshl); R(k_ashr); R(k_mov);
  R(k_cndmask); R(k_cmp_f32); R(k_cmp_i32); R(k_cmp_u32); R(k_cmp_sgpr); R(k_sub_co); R(k_addc); R(k_min_f32); R(k_max3);
  R(k_mul_lo); R(k_ffbl); R(k_bcnt); R(k_mbcnt); R(k_cvt); R(k_sad); R(k_max_i16); R(k_min3_i32);
  run("k_readlane+mov", k_readlane, 32); R(k_dpp_add); R(k_dpp_add_qp); R(k_sdwa);
  return 0;
}
