// Second-round VALU pricing: which operand forms / opcodes run at the full (2-cycle) FP32 rate on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// 32-bit form: %0 = acc (rw VGPR), %1 = b (VGPR), %2 = c (VGPR), %3 = sb (SGPR)
#define K32(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters, float sb) {                   \
    float b = 1.0f + threadIdx.x * 1e-9f, c = 1e-9f; float a[16];                                   \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;                         \
    for (int it = 0; it < iters; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i)                                               \
        asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(sb));                                  \
    }                                                                                               \
    float s = 0.f; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i];                       \
    if (s == 12345.678f) out[0] = s;                                                                \
  }
// 64-bit form: %0 = acc pair (rw), %1 = b2, %2 = c2 (VGPR pairs), %3 = s2 (SGPR pair)
#define K64(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters, f2 s2) {                      \
    float b = 1.0f + threadIdx.x * 1e-9f, c = 1e-9f; f2 b2 = {b, b}, c2 = {c, c}; f2 p[16];        \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) p[i] = f2{(float)threadIdx.x + i, 1.0f};        \
    for (int it = 0; it < iters; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i)                                               \
        asm volatile(ASM : "+v"(p[i]) : "v"(b2), "v"(c2), "s"(s2));                                \
    }                                                                                               \
    float s = 0.f; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;            \
    if (s == 12345.678f) out[0] = s;                                                                \
  }

K32(k_fma_vvv, "v_fma_f32 %0, %1, %2, %0")
K32(k_fma_svv, "v_fma_f32 %0, %3, %2, %0")
K32(k_fma_vsv, "v_fma_f32 %0, %2, %3, %0")
K32(k_fma_same, "v_fma_f32 %0, %1, %1, %0")
K32(k_fmac_vv, "v_fmac_f32 %0, %1, %2")
K32(k_fmac_sv, "v_fmac_f32 %0, %3, %2")
K32(k_fmac_lit, "v_fmac_f32 %0, 0x3f8ccccd, %2")
K32(k_fmac_inl, "v_fmac_f32 %0, 0.5, %2")
K32(k_mul_vv, "v_mul_f32 %0, %1, %0")
K32(k_add_vv, "v_add_f32 %0, %1, %0")
K32(k_add_lit, "v_add_f32 %0, 0xc2ff0000, %0")
K32(k_sub_vv, "v_sub_f32 %0, %0, %1")
K32(k_max_vv, "v_max_f32 %0, %1, %0")
K32(k_cvt_ub0, "v_cvt_f32_ubyte0 %0, %1")
K32(k_cvt_ub3, "v_cvt_f32_ubyte3 %0, %1")
K32(k_cvt_u32, "v_cvt_f32_u32 %0, %1")
K32(k_and, "v_and_b32 %0, %1, %0")
K32(k_lshl, "v_lshlrev_b32 %0, 1, %0")
K32(k_bfe, "v_bfe_u32 %0, %1, 8, 8")
K32(k_mov, "v_mov_b32 %0, %1")
K32(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
K32(k_add_dpp, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
K32(k_rcp, "v_rcp_f32 %0, %0")
K32(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
K32(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
K32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
K32(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %1")
K32(k_cvt_sdwa, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
K32(k_add_sdwa, "v_add_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
K64(k_pkfma_vvv, "v_pk_fma_f32 %0, %1, %2, %0")
K64(k_pkfma_svv, "v_pk_fma_f32 %0, %3, %2, %0")
K64(k_pkfma_sbc, "v_pk_fma_f32 %0, %3, %2, %0 op_sel_hi:[0,1,1]")
K64(k_pkfma_vbc, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]")
K64(k_pkmul, "v_pk_mul_f32 %0, %1, %0")
K64(k_pkadd, "v_pk_add_f32 %0, %1, %0")
K64(k_pkmov, "v_pk_mov_b32 %0, %1, %2")

template <typename KT, typename AT>
static void run(const char* name, KT kern, AT sarg, float* d_out, int wps) {
  int blocks = 256 * wps, iters = 100000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(d_out, 20000, sarg);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<blocks, 256>>>(d_out, iters, sarg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double winstr = (double)blocks * 4 * iters * 16;
  double r = winstr / (ms * 1e-3) / 1024.0;
  printf("{\"op\":\"%s\",\"wps\":%d,\"ms\":%.2f,\"Gwinstr_s_simd\":%.4f,\"ns_per_winstr\":%.3f}\n", name, wps, ms, r * 1e-9, 1e9 / r);
  fflush(stdout);
}
#define R32(k) run(#k, k, 1.0f, d_out, wps)
#define R64(k) run(#k, k, f2{1.0f, 1.0f}, d_out, wps)
int main() {
  float* d_out; CK(hipMalloc(&d_out, 1024));
  for (int wps : {4, 8}) {
    R32(k_fma_vvv); R32(k_fma_svv); R32(k_fma_vsv); R32(k_fma_same); R32(k_fmac_vv); R32(k_fmac_sv); R32(k_fmac_lit); R32(k_fmac_inl);
    R32(k_mul_vv); R32(k_add_vv); R32(k_add_lit); R32(k_sub_vv); R32(k_max_vv); R32(k_cvt_ub0); R32(k_cvt_ub3); R32(k_cvt_u32);
    R32(k_and); R32(k_lshl); R32(k_bfe); R32(k_mov); R32(k_mov_dpp); R32(k_add_dpp); R32(k_rcp); R32(k_cndmask); R32(k_bfi);
    R32(k_mad_u32_u24); R32(k_cvt_f32_f16); R32(k_cvt_sdwa); R32(k_add_sdwa);
    R64(k_pkfma_vvv); R64(k_pkfma_svv); R64(k_pkfma_sbc); R64(k_pkfma_vbc); R64(k_pkmul); R64(k_pkadd); R64(k_pkmov);
  }
  return 0;
}
