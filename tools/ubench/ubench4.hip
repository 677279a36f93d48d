// FIR inner loop in isolation: R outputs per lane from an LDS-resident f32x2 tile, SGPR taps, v_pk_fma_f32.
// Prices the loop against waves/CU (controlled through the dynamic LDS size) and R.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
template <int HI> __device__ __forceinline__ void pk_fma_bcast(f2_t& acc, f2_t tap_pair, f2_t x) {
  if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}
template <int T, int D, int R>
__global__ void __launch_bounds__(64) fir_loop(const float* __restrict__ h, float* out, int iters, int lds_bytes_used) {
  constexpr int RD = R * D, HP = T - D, NW = RD + HP;
  constexpr int ROWPAD = ((RD / 2) % 2 == 0) ? 2 : 0, RS = (RD + ROWPAD) * 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < lds_bytes_used / 4; i += 64) reinterpret_cast<float*>(smem)[i] = (float)(i % 251) - 125.f;
  __syncthreads();
  f2_t hp[T / 2];
#pragma unroll
  for (int k = 0; k < T / 2; ++k) hp[k] = f2_t{h[2 * k], h[2 * k + 1]};
  unsigned woff = lane * RS;
  f2_t tot = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(woff) :: "memory");   // opaque OFFSET per iteration (keeps the LDS address space)
    const unsigned char* win = smem + woff;
    f2_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f2_t{0.f, 0.f};
#pragma unroll
    for (int j2 = 0; j2 < NW / 2; ++j2) {
      const int row = (2 * j2) / RD, col = (2 * j2) % RD;
      const f4_t v = *reinterpret_cast<const f4_t*>(win + row * RS + col * 8);
      const f2_t x0 = {v.x, v.y}, x1 = {v.z, v.w};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int p0 = 2 * j2 - r * D;
        if (p0 >= 0 && p0 < T) { const int k = T - 1 - p0; if (k & 1) pk_fma_bcast<1>(acc[r], hp[k / 2], x0); else pk_fma_bcast<0>(acc[r], hp[k / 2], x0); }
        const int p1 = p0 + 1;
        if (p1 >= 0 && p1 < T) { const int k = T - 1 - p1; if (k & 1) pk_fma_bcast<1>(acc[r], hp[k / 2], x1); else pk_fma_bcast<0>(acc[r], hp[k / 2], x1); }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) tot += acc[r];
  }
  if (tot.x == 1234.5f) out[0] = tot.y;
}
template <int T, int D, int R>
static void run(const float* d_h, float* d_o, int waves_per_cu) {
  constexpr int RD = R * D, HP = T - D, NW = RD + HP, RS = (RD + (((RD / 2) % 2 == 0) ? 2 : 0)) * 8;
  int used = (64 + (NW + RD - 1) / RD) * RS;     // rows the windows touch
  int lds = 160 * 1024 / waves_per_cu; lds -= lds % 256;
  if (lds < used) { printf("{\"R\":%d,\"waves_per_cu\":%d,\"skip\":\"tile %d B does not fit\"}\n", R, waves_per_cu, used); return; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_loop<T, D, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int blocks = 256 * waves_per_cu, iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  fir_loop<T, D, R><<<blocks, 64, lds>>>(d_h, d_o, 50, used); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  fir_loop<T, D, R><<<blocks, 64, lds>>>(d_h, d_o, iters, used);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double pk = (double)blocks * iters * R * T;           // pk_fma wave-instructions
  double samples = (double)blocks * iters * 64.0 * RD;  // input samples "consumed"
  printf("{\"ubench\":\"fir_loop\",\"T\":%d,\"R\":%d,\"waves_per_cu\":%d,\"ms\":%.3f,\"ns_per_pkfma_per_simd\":%.3f,\"Tsamples_per_s\":%.3f,\"reads_per_iter\":%d}\n", T, R,
         waves_per_cu, ms, ms * 1e6 / (pk / 1024.0), samples / (ms * 1e-3) * 1e-12, NW / 2);
  fflush(stdout);
}
int main() {
  float hh[64]; for (int i = 0; i < 64; ++i) hh[i] = 0.01f * (i + 1);
  float *d_h, *d_o; CK(hipMalloc(&d_h, sizeof(hh))); CK(hipMalloc(&d_o, 64));
  CK(hipMemcpy(d_h, hh, sizeof(hh), hipMemcpyHostToDevice));
  for (int w : {4, 6, 8, 12, 16}) { run<64, 10, 2>(d_h, d_o, w); run<64, 10, 3>(d_h, d_o, w); run<64, 10, 4>(d_h, d_o, w); run<64, 10, 8>(d_h, d_o, w); }
  return 0;
}
