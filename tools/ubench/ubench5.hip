// What bounds the FIR loop: LDS read rate vs VALU, by ablation (R=4 window, 47 ds_read_b128 per iteration).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
template <int HI> __device__ __forceinline__ void pk_s(f2_t& acc, f2_t tp, f2_t x) {
  if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tp), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tp), "v"(x));
}
// MODE 0: reads only (xor-reduce)   1: reads + pk_fma with SGPR taps   2: reads + pk_fma with one VGPR tap   3: pk_fma only (no LDS)
// STRIDE: lane stride in bytes
template <int MODE, int RS>
__global__ void __launch_bounds__(64) loop(const float* __restrict__ h, float* out, int iters, int used) {
  constexpr int T = 64, D = 10, R = 4, RD = R * D, HP = T - D, NW = RD + HP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < used / 4; i += 64) reinterpret_cast<float*>(smem)[i] = (float)(i % 251) - 125.f;
  __syncthreads();
  f2_t hp[T / 2];
#pragma unroll
  for (int k = 0; k < T / 2; ++k) hp[k] = f2_t{h[2 * k], h[2 * k + 1]};
  unsigned woff = lane * RS;
  f2_t tot = {0.f, 0.f};
  f2_t vt = {h[lane & 7], h[lane & 7]};
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(woff) :: "memory");
    const unsigned char* win = smem + woff;
    f2_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f2_t{0.f, 0.f};
    f4_t xr = {0, 0, 0, 0};
#pragma unroll
    for (int j2 = 0; j2 < NW / 2; ++j2) {
      const int row = (2 * j2) / RD, col = (2 * j2) % RD;
      f4_t v;
      if constexpr (MODE == 3) v = f4_t{tot.x, tot.y, vt.x, vt.y}; else v = *reinterpret_cast<const f4_t*>(win + row * (RD * 8 + (RS - RD * 8)) + col * 8);
      if constexpr (MODE == 0) { xr += v; continue; }
      const f2_t x0 = {v.x, v.y}, x1 = {v.z, v.w};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int p0 = 2 * j2 - r * D;
        if (p0 >= 0 && p0 < T) { const int k = T - 1 - p0;
          if constexpr (MODE == 2) acc[r] = __builtin_elementwise_fma(vt, x0, acc[r]);
          else { if (k & 1) pk_s<1>(acc[r], hp[k / 2], x0); else pk_s<0>(acc[r], hp[k / 2], x0); } }
        const int p1 = p0 + 1;
        if (p1 >= 0 && p1 < T) { const int k = T - 1 - p1;
          if constexpr (MODE == 2) acc[r] = __builtin_elementwise_fma(vt, x1, acc[r]);
          else { if (k & 1) pk_s<1>(acc[r], hp[k / 2], x1); else pk_s<0>(acc[r], hp[k / 2], x1); } }
      }
    }
    if constexpr (MODE == 0) tot += f2_t{xr.x + xr.z, xr.y + xr.w};
#pragma unroll
    for (int r = 0; r < R; ++r) tot += acc[r];
  }
  if (tot.x == 1234.5f) out[0] = tot.y;
}
template <int MODE, int RS>
static void run(const char* name, const float* d_h, float* d_o, int wpc) {
  int used = 67 * RS;
  int lds = 160 * 1024 / wpc; lds -= lds % 256;
  if (lds < used) return;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop<MODE, RS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int blocks = 256 * wpc, iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  loop<MODE, RS><<<blocks, 64, lds>>>(d_h, d_o, 50, used); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  loop<MODE, RS><<<blocks, 64, lds>>>(d_h, d_o, iters, used);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double iters_per_cu = (double)wpc * iters;
  printf("{\"mode\":\"%s\",\"row_stride\":%d,\"waves_per_cu\":%d,\"ms\":%.3f,\"ns_per_iter_per_cu\":%.1f,\"ns_per_read_per_cu\":%.2f,\"ns_per_pkfma_per_simd\":%.2f}\n", name, RS, wpc, ms,
         ms * 1e6 / iters_per_cu, ms * 1e6 / iters_per_cu / 47.0, ms * 1e6 / (iters_per_cu / 4.0) / 256.0);
  fflush(stdout);
}
int main() {
  float hh[64]; for (int i = 0; i < 64; ++i) hh[i] = 0.01f * (i + 1);
  float *d_h, *d_o; CK(hipMalloc(&d_h, sizeof(hh))); CK(hipMalloc(&d_o, 64));
  CK(hipMemcpy(d_h, hh, sizeof(hh), hipMemcpyHostToDevice));
  for (int w : {4, 6}) {
    run<0, 336>("reads_only", d_h, d_o, w); run<1, 336>("reads+pk_sgpr", d_h, d_o, w); run<2, 336>("reads+pk_vgpr", d_h, d_o, w); run<3, 336>("pk_sgpr_only", d_h, d_o, w);
    run<0, 320>("reads_only", d_h, d_o, w); run<1, 320>("reads+pk_sgpr", d_h, d_o, w);
    run<0, 352>("reads_only", d_h, d_o, w); run<1, 352>("reads+pk_sgpr", d_h, d_o, w);
    run<0, 368>("reads_only", d_h, d_o, w); run<1, 368>("reads+pk_sgpr", d_h, d_o, w);
  }
  return 0;
}
