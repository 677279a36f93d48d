// "bytes-in-LDS" FIR loop: raw u8 I/Q in LDS, each lane converts its own window (v_cvt_f32_ubyteN + v_pk_add_f32)
// and runs R outputs with SGPR taps / v_pk_fma_f32.  Prices design (i) of DESIGN.md against waves/CU and R.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
template <int HI> __device__ __forceinline__ void pk_fma_bcast(f2_t& acc, f2_t tap_pair, f2_t x) {
  if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}
__device__ __forceinline__ f2_t cvt_pair(unsigned w, int hi) {   // bytes (I,Q) of the low / high half of w -> (I-127.5, Q-127.5)
  f2_t c;
  if (hi) { c.x = (float)((w >> 16) & 0xffu); c.y = (float)(w >> 24); }            // -> v_cvt_f32_ubyte2 / 3
  else { c.x = (float)(w & 0xffu); c.y = (float)((w >> 8) & 0xffu); }                // -> v_cvt_f32_ubyte0 / 1
  return c - f2_t{127.5f, 127.5f};
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int T, int D, int R>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) fir_bytes(const float* __restrict__ h, float* out, int iters, int used, int rs) {
  constexpr int RD = R * D, HP = T - D, NW = RD + HP;          // samples; 8 samples per b128
  constexpr int NRD = (NW + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < used / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0x80818283u + i * 0x01010101u;
  __syncthreads();
  f2_t hp[T / 2];
#pragma unroll
  for (int k = 0; k < T / 2; ++k) hp[k] = f2_t{h[2 * k], h[2 * k + 1]};
  unsigned woff = lane * rs;
  f2_t tot = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(woff) :: "memory");
    const unsigned char* win = smem + woff;
    f2_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f2_t{0.f, 0.f};
    static_for<0, NRD>([&](auto J8) {
      constexpr int j8 = decltype(J8)::value;
      const u4_t v = *reinterpret_cast<const u4_t*>(win + j8 * 16);
      static_for<0, 8>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int j = j8 * 8 + s;
        if constexpr (j < NW) {
          const unsigned w = (s / 2 == 0) ? v.x : (s / 2 == 1) ? v.y : (s / 2 == 2) ? v.z : v.w;
          const f2_t x = cvt_pair(w, s & 1);
          static_for<0, R>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            constexpr int p0 = j - r * D;
            if constexpr (p0 >= 0 && p0 < T) {
              constexpr int k = T - 1 - p0;
              if constexpr (k & 1) pk_fma_bcast<1>(acc[r], hp[k / 2], x); else pk_fma_bcast<0>(acc[r], hp[k / 2], x);
            }
          });
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int r = 0; r < R; ++r) tot += acc[r];
  }
  if (tot.x == 1234.5f) out[0] = tot.y;
}
template <int T, int D, int R>
static void run(const float* d_h, float* d_o, int wpc) {
  constexpr int RD = R * D, HP = T - D, NW = RD + HP;
  int rs = RD * 2; if ((rs / 16) % 2 == 0) rs += 16;             // odd number of 16-B slots per lane row
  int used = 64 * rs + ((NW * 2 + 15) & ~15) + 64;
  int lds = 160 * 1024 / wpc; lds -= lds % 256;
  if (lds < used) { printf("{\"R\":%d,\"waves_per_cu\":%d,\"skip\":\"tile %d B\"}\n", R, wpc, used); return; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_bytes<T, D, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int blocks = 256 * wpc, iters = 1000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  fir_bytes<T, D, R><<<blocks, 64, lds>>>(d_h, d_o, 20, used, rs); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  fir_bytes<T, D, R><<<blocks, 64, lds>>>(d_h, d_o, iters, used, rs);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double samples = (double)blocks * iters * 64.0 * RD;
  double pk = (double)blocks * iters * R * T;
  printf("{\"ubench\":\"fir_bytes\",\"T\":%d,\"R\":%d,\"waves_per_cu\":%d,\"ms\":%.3f,\"ns_per_pkfma_per_simd\":%.3f,\"Tsamples_per_s\":%.3f,\"tile_bytes\":%d}\n", T, R, wpc, ms,
         ms * 1e6 / (pk / 1024.0), samples / (ms * 1e-3) * 1e-12, used);
  fflush(stdout);
}
int main() {
  float hh[64]; for (int i = 0; i < 64; ++i) hh[i] = 0.01f * (i + 1);
  float *d_h, *d_o; CK(hipMalloc(&d_h, sizeof(hh))); CK(hipMalloc(&d_o, 64));
  CK(hipMemcpy(d_h, hh, sizeof(hh), hipMemcpyHostToDevice));
  for (int w : {4, 8}) { run<64, 10, 8>(d_h, d_o, w); run<64, 10, 12>(d_h, d_o, w); run<64, 10, 16>(d_h, d_o, w); }
  return 0;
}
