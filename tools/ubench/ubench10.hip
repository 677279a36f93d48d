// Does the f32 matrix pipe run BESIDE the vector pipe on gfx950, and for which instruction shapes?
// ubench9 showed v_mfma_f32_4x4x1_16b_f32 + n VALU fillers costs ~ (8 + 4n) cycles: no overlap.  This probe varies
//   * where the accumulator lives (VGPR / AGPR / C = inline 0),
//   * the filler (VALU with 0..3 VGPR reads, SALU, LDS read),
//   * the MFMA shape (4x4x1 16 blocks: 2 passes; 16x16x4: 8 passes; 32x32x2: 16 passes),
// always with 4 independent accumulators and fillers on registers the MFMA never touches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SHAPE: 0 = 4x4x1 acc in VGPR, 1 = 4x4x1 acc in AGPR, 2 = 4x4x1 C = 0 (result overwritten), 3 = 16x16x4 VGPR, 4 = 32x32x2 VGPR,
//        5 = 16x16x4 AGPR
// FKIND: 0 v_fma_f32 v,v,v,v ; 1 v_mov_b32 v, s ; 2 v_add_f32 v, s, v ; 3 s_add_u32 ; 4 ds_read_b32 ; 5 v_cvt_f32_ubyte1 ; 6 v_pk_fma_f32
template <int SHAPE, int FKIND, int FILL>
__global__ void __launch_bounds__(256) k_mix(float* out, int iters, float sa, unsigned sb) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = sa + threadIdx.x, b = 1.0f + threadIdx.x * 1e-3f;
  float v[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pk[4];
  float fa = 0.5f + threadIdx.x, fb = 0.25f;       // filler operands: never MFMA operands
  unsigned raw = threadIdx.x * 0x01010101u, sacc = sb;
  const unsigned ldsaddr = (threadIdx.x & 255) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk[i] = f2{(float)threadIdx.x, (float)i};
  f4 acc[4];
  f16v big[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
  if constexpr (SHAPE == 1 || SHAPE == 5) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc[i].x));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (SHAPE == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
      if constexpr (SHAPE == 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a), "v"(b));
      if constexpr (SHAPE == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(acc[u & 3]) : "v"(a), "v"(b));
      if constexpr (SHAPE == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
      if constexpr (SHAPE == 5) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a), "v"(b));
      if constexpr (SHAPE == 4) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big[u & 1]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < FILL; ++f) {
        const int r = (u * FILL + f) & 7;
        if constexpr (FKIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(fa), "v"(fb));
        if constexpr (FKIND == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "s"(sa));
        if constexpr (FKIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[r]) : "s"(sa));
        if constexpr (FKIND == 3) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
        if constexpr (FKIND == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r]) : "v"(ldsaddr));
        if constexpr (FKIND == 5) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(v[r]) : "v"(raw));
        if constexpr (FKIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pk[r & 3]) : "v"(pk[(r + 1) & 3]));
      }
    }
    if constexpr (FKIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
  float s = (float)sacc;
  if constexpr (SHAPE == 1 || SHAPE == 5) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[i].x)); s += t; }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) s += big[0][j] + big[1][j];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += pk[i].x + pk[i].y;
  if (s == 12345.678f) out[0] = s;
}

static const char* shape_name[] = {"4x4x1_vgpr", "4x4x1_agpr", "4x4x1_c0", "16x16x4_vgpr", "32x32x2_vgpr", "16x16x4_agpr"};
static const double shape_flop[] = {512, 512, 512, 2048, 4096, 2048};
static const char* fill_name[] = {"v_fma_f32_vvv", "v_mov_b32_s", "v_add_f32_sv", "s_add_u32", "ds_read_b32", "v_cvt_f32_ubyte1", "v_pk_fma_f32"};

template <typename KT>
static void run(int shape, int fk, int fill, KT kern, float* d_out, int wps) {
  const int blocks = 256 * wps, iters = 8000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(d_out, 1000, 1.0f, 5u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<blocks, 256>>>(d_out, iters, 1.0f, 5u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = (double)wps * iters * 16;
  printf("{\"mfma\":\"%s\",\"filler\":\"%s\",\"fill_per_mfma\":%d,\"waves_per_simd\":%d,\"ns_per_mfma_per_simd\":%.3f,\"mfma_TFLOPs\":%.1f}\n",
         shape_name[shape], fill ? fill_name[fk] : "none", fill, wps, ms * 1e6 / mfma_per_simd,
         1024.0 * mfma_per_simd * shape_flop[shape] / (ms * 1e-3) * 1e-12);
  fflush(stdout);
}
#define RUN(S, F, N) run(S, F, N, k_mix<S, F, N>, d_out, wps)

int main() {
  float* d_out; CK(hipMalloc(&d_out, 1024));
  for (int wps : {2, 4}) {
    RUN(0, 0, 0); RUN(1, 0, 0); RUN(2, 0, 0); RUN(3, 0, 0); RUN(4, 0, 0); RUN(5, 0, 0);
    // 4x4x1, accumulator placement, two fillers per MFMA
    RUN(0, 0, 2); RUN(1, 0, 2); RUN(2, 0, 2);
    RUN(1, 0, 1); RUN(1, 0, 3); RUN(1, 5, 1); RUN(1, 5, 2); RUN(1, 6, 1); RUN(1, 6, 2);
    // 4x4x1 beside other kinds of instruction
    RUN(0, 1, 2); RUN(0, 2, 2); RUN(0, 3, 2); RUN(0, 4, 1); RUN(0, 4, 2); RUN(1, 1, 2); RUN(1, 3, 2);
    // 16x16x4 (8 passes): how many VALU hide in its shadow?
    RUN(3, 0, 2); RUN(3, 0, 4); RUN(3, 0, 6); RUN(3, 0, 8); RUN(3, 0, 12); RUN(3, 5, 4); RUN(3, 5, 6); RUN(3, 6, 4); RUN(3, 6, 6);
    RUN(5, 0, 4); RUN(5, 0, 8); RUN(5, 0, 12); RUN(5, 5, 6); RUN(5, 6, 6);
    // 32x32x2 (16 passes)
    RUN(4, 0, 8); RUN(4, 0, 16); RUN(4, 0, 24); RUN(4, 5, 8); RUN(4, 5, 12);
  }
  return 0;
}
