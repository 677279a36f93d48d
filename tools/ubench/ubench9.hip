// v_mfma_f32_4x4x1_16b_f32 on gfx950: operand layout, bit-equality with an fmaf chain, issue rate alone and beside VALU work.
// Question behind it: can the 64-tap decimating FIR ride the matrix pipe as rank-1 updates whose
// instruction order IS the oracle's oldest-first chain order, with zero taps as bit-neutral padding?
//   layout hypothesis: lane l = 4*blk + r.  A: lane supplies A_blk[row r].  B: lane supplies B_blk[col r].
//                      D: lane (blk, col r) receives D_blk[row i][col r] in register i (i = 0..3).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- exactness: each wave runs K rank-1 updates on one accumulator quad -----------------------------------------------
__global__ void __launch_bounds__(64) k_chain(const float* A, const float* B, float* out, int K) {
  const size_t w = blockIdx.x;
  const int lane = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) {
    const float a = A[(w * K + k) * 64 + lane], b = B[(w * K + k) * 64 + lane];
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
  }
  reinterpret_cast<f4*>(out)[w * 64 + lane] = acc;
}

// ---- rate: NACC independent accumulator quads, FILL VALU instructions of one kind after every MFMA -------------------
template <int KIND, int FILL>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, float sa) {
  f4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = sa + threadIdx.x, b = 1.0f + threadIdx.x * 1e-3f;
  float v[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pk[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk[i] = f2{(float)threadIdx.x, (float)i};
  unsigned raw = threadIdx.x * 0x01010101u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) {
        const int r = (u * FILL + f) & 7;
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(a), "v"(b));
        if constexpr (KIND == 1) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(v[r]) : "v"(raw));
        if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[r & 3]) : "v"(pk[(r + 1) & 3]));
        if constexpr (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pk[r & 3]) : "v"(pk[(r + 1) & 3]));
        if constexpr (KIND == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[r]) : "v"(b));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += pk[i].x + pk[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <typename KT>
static void run(const char* name, int fill, KT kern, float* d_out, int wps) {
  const int blocks = 256 * wps, iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(d_out, 2000, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<blocks, 256>>>(d_out, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = (double)wps * iters * 16;   // one wave per SIMD per `wps`
  printf("{\"op\":\"mfma4x4x1+%s\",\"fill_per_mfma\":%d,\"waves_per_simd\":%d,\"ms\":%.2f,\"ns_per_mfma_per_simd\":%.3f,\"TFLOPs\":%.1f}\n",
         name, fill, wps, ms, ms * 1e6 / mfma_per_simd, 1024.0 * mfma_per_simd * 512 / (ms * 1e-3) * 1e-12);
  fflush(stdout);
}

int main() {
  // ---- 1. layout + exactness ----
  const int W = 4096, K = 94;          // 4096 waves x 256 chains = 1 048 576 chains of 94 updates (the Toeplitz window T + 3D)
  std::vector<float> A((size_t)W * K * 64), B((size_t)W * K * 64), out((size_t)W * 256);
  srand(11);
  for (size_t w = 0; w < (size_t)W; ++w)
    for (int k = 0; k < K; ++k)
      for (int l = 0; l < 64; ++l) {
        const int row = l & 3;
        // taps: Toeplitz pattern of a 64-tap filter over 4 outputs 10 apart (zeros outside), random signed values of mixed scale
        const int tapidx = row * 10 + 63 - k;
        float a = 0.0f;
        if (tapidx >= 0 && tapidx < 64) {
          a = (float)((rand() % 20001) - 10000) / 65536.0f * (1.0f + (rand() % 1000) * 1e-6f);
          if ((rand() & 63) == 0) a = 0.0f;                 // exact zero taps occur in windowed sincs
          if ((rand() & 255) == 0) a *= 1e-30f;             // tiny taps: subnormal products
        }
        A[(w * K + k) * 64 + l] = a;
        B[(w * K + k) * 64 + l] = (float)(rand() & 255) - 127.5f;   // what K1 produces
      }
  float *dA, *dB, *dO;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dO, out.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  k_chain<<<W, 64>>>(dA, dB, dO, K);
  CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
  long bad = 0, negzero = 0;
  for (size_t w = 0; w < (size_t)W; ++w)
    for (int blk = 0; blk < 16; ++blk)
      for (int col = 0; col < 4; ++col)
        for (int i = 0; i < 4; ++i) {
          float acc = 0.0f;
          for (int k = 0; k < K; ++k) acc = fmaf(A[(w * K + k) * 64 + 4 * blk + i], B[(w * K + k) * 64 + 4 * blk + col], acc);
          const float got = out[(w * 64 + 4 * blk + col) * 4 + i];
          if (memcmp(&acc, &got, 4)) ++bad;
          if (acc == 0.0f && std::signbit(acc)) ++negzero;
        }
  printf("{\"check\":\"v_mfma_f32_4x4x1_16b_f32 x%d == fmaf chain, D[blk][i][col] in lane 4*blk+col reg i\",\"chains\":%ld,\"mismatches\":%ld,\"negative_zero_results\":%ld}\n",
         K, (long)W * 256, bad, negzero);
  // ---- 2. issue rate ----
  float* d_out; CK(hipMalloc(&d_out, 1024));
  for (int wps : {1, 2, 4}) {
    run("none", 0, k_rate<0, 0>, d_out, wps);
    run("v_fma_f32", 1, k_rate<0, 1>, d_out, wps);
    run("v_fma_f32", 2, k_rate<0, 2>, d_out, wps);
    run("v_fma_f32", 3, k_rate<0, 3>, d_out, wps);
    run("v_fma_f32", 4, k_rate<0, 4>, d_out, wps);
    run("v_cvt_f32_ubyte1", 1, k_rate<1, 1>, d_out, wps);
    run("v_cvt_f32_ubyte1", 2, k_rate<1, 2>, d_out, wps);
    run("v_cvt_f32_ubyte1", 3, k_rate<1, 3>, d_out, wps);
    run("v_pk_add_f32", 1, k_rate<2, 1>, d_out, wps);
    run("v_pk_add_f32", 2, k_rate<2, 2>, d_out, wps);
    run("v_pk_fma_f32", 1, k_rate<3, 1>, d_out, wps);
    run("v_pk_fma_f32", 2, k_rate<3, 2>, d_out, wps);
    run("v_add_f32", 2, k_rate<4, 2>, d_out, wps);
    run("v_add_f32", 3, k_rate<4, 3>, d_out, wps);
  }
  return 0;
}
