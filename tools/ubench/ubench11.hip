// Gate (b) of design Q: the i8 matrix pipe of gfx950 BESIDE the vector pipe.
//   1. operand layout of v_mfma_i32_16x16x64_i8: checked against a CPU product with the layout design Q assumes
//      (lane l: A row / B column l & 15, bytes K = 16 (l >> 4) .. +15; D[4 (l >> 4) + r][l & 15] in register r);
//   2. ns per MFMA per SIMD alone, 1..4 waves per SIMD, 3 independent accumulators (design Q's three tap digits);
//   3. the same with n filler instructions per MFMA placed between the MFMAs (the kinds design Q needs beside them):
//      does the filler's issue time hide in the MFMA's shadow (f32 MFMA: it does not, profiles/ubench_r02)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k_layout(const i4* a, const i4* b, i4* d) {
  const int l = threadIdx.x;
  i4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], acc, 0, 0, 0);
  d[l] = acc;
}

// FKIND: 0 v_xor_b32; 1 v_cvt_f32_i32; 2 v_fma_f32; 3 v_pk_fma_f32; 4 v_rcp_f32; 5 ds_read_b128; 6 v_lshl_add_u32; 7 v_pk_mul_f32;
//        8 v_cndmask_b32 (vcc)
template <int FKIND, int FILL>
__global__ void __launch_bounds__(256) k_mix(float* out, int iters, int seed) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * seed;
  __syncthreads();
  i4 a = {seed * 3 + (int)threadIdx.x, seed, seed * 7, seed + 5}, b = {seed + 11, (int)threadIdx.x, seed * 5, seed * 13};
  i4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = i4{0, 0, 0, 0};
  unsigned u[8];
  float v[8];
  f2 pk[4];
  i4 ld[2] = {i4{0, 0, 0, 0}, i4{0, 0, 0, 0}};
#pragma unroll
  for (int i = 0; i < 8; ++i) { u[i] = threadIdx.x * 0x01010101u + i; v[i] = 1.0f + threadIdx.x + i; }
#pragma unroll
  for (int i = 0; i < 4; ++i) pk[i] = f2{(float)threadIdx.x, (float)i};
  const float fa = 0.5f + threadIdx.x, fb = 0.25f;
  const unsigned ldsaddr = (threadIdx.x & 255) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 15; ++m) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[m % 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < FILL; ++f) {
        const int r = (m * FILL + f) & 7;
        if constexpr (FKIND == 0) asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(u[r]));
        if constexpr (FKIND == 1) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(v[r]) : "v"(u[(r + 1) & 7]));
        if constexpr (FKIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(fa), "v"(fb));
        if constexpr (FKIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pk[r & 3]) : "v"(pk[(r + 1) & 3]));
        if constexpr (FKIND == 4) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[r]) : "v"(fa));
        if constexpr (FKIND == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[r & 1]) : "v"(ldsaddr));
        if constexpr (FKIND == 6) asm volatile("v_lshl_add_u32 %0, %1, 8, %0" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
        if constexpr (FKIND == 7) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(pk[r & 3]) : "v"(pk[(r + 1) & 3]));
        if constexpr (FKIND == 8) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]));
      }
    }
    if constexpr (FKIND == 5) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (float)(acc[i].x + acc[i].y + acc[i].z + acc[i].w);
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += pk[i].x + pk[i].y;
  s += (float)(ld[0].x + ld[1].y);
  if (s == 12345.678f) out[0] = s;
}

static const char* fill_name[] = {"v_xor_b32", "v_cvt_f32_i32", "v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "ds_read_b128", "v_lshl_add_u32", "v_pk_mul_f32", "v_cndmask_b32"};

template <typename KT>
static void run(int fk, int fill, KT kern, float* d_out, int wps) {
  const int blocks = 256 * wps, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(d_out, 500, 3);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<blocks, 256>>>(d_out, iters, 3);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = (double)wps * iters * 15;
  printf("{\"mfma\":\"i32_16x16x64_i8\",\"filler\":\"%s\",\"fill_per_mfma\":%d,\"waves_per_simd\":%d,\"ns_per_mfma_per_simd\":%.3f,\"mfma_TOPs\":%.1f}\n",
         fill ? fill_name[fk] : "none", fill, wps, ms * 1e6 / mfma_per_simd, 1024.0 * mfma_per_simd * 32768.0 / (ms * 1e-3) * 1e-12);
  fflush(stdout);
}
#define RUN(F, N) run(F, N, k_mix<F, N>, d_out, wps)

int main() {
  // ---- 1. layout ----------------------------------------------------------------------------------------------------
  std::vector<int8_t> A(16 * 64), B(64 * 16);
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int8_t)(x >> 33); };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  std::vector<int8_t> ha(64 * 16), hb(64 * 16);
  for (int l = 0; l < 64; ++l)
    for (int p = 0; p < 16; ++p) {
      ha[l * 16 + p] = A[(l & 15) * 64 + 16 * (l >> 4) + p];          // A[row][k]
      hb[l * 16 + p] = B[(16 * (l >> 4) + p) * 16 + (l & 15)];        // B[k][col]
    }
  i4 *da, *db, *dd;
  CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 1024));
  CK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
  k_layout<<<1, 64>>>(da, db, dd);
  std::vector<int> hd(256);
  CK(hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * (l >> 4) + r, col = l & 15;
      int ref = 0;
      for (int k = 0; k < 64; ++k) ref += (int)A[row * 64 + k] * (int)B[k * 16 + col];
      bad += (ref != hd[l * 4 + r]);
    }
  printf("{\"probe\":\"layout v_mfma_i32_16x16x64_i8\",\"mismatches\":%d,\"of\":256}\n", bad);
  // ---- 2./3. throughput --------------------------------------------------------------------------------------------
  float* d_out; CK(hipMalloc(&d_out, 1024));
  for (int wps : {1, 2, 3, 4}) {
    RUN(0, 0);
    RUN(0, 1); RUN(0, 2); RUN(0, 4); RUN(0, 6); RUN(0, 8);
    RUN(2, 2); RUN(2, 4); RUN(2, 6); RUN(2, 8);
    RUN(1, 2); RUN(1, 4);
    RUN(3, 1); RUN(3, 2); RUN(3, 4);
    RUN(7, 2); RUN(7, 4);
    RUN(4, 1); RUN(4, 2);
    RUN(5, 1); RUN(5, 2);
    RUN(6, 2); RUN(6, 4);
    RUN(8, 2); RUN(8, 4);
  }
  return 0;
}
