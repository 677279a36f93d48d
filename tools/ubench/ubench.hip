// Micro-benchmarks that price the VALU / memory primitives the IQ->FM kernel is built from
// on the actual MI355X (gfx950).  Not part of the product path; results are quoted in DESIGN.md.
//   build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { OP_FMA = 0, OP_PKFMA, OP_FMAMIX, OP_CVTUB, OP_PKADD32, OP_PKADD16, OP_PERM, OP_FMA_SGPR, OP_ADD32, OP_COUNT };
static const char* op_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_mix_f32", "v_cvt_f32_ubyte1", "v_pk_add_f32",
                                "v_pk_add_f16", "v_perm_b32", "v_fma_f32(sgpr tap)", "v_add_f32"};

template <int OP>
__global__ void __launch_bounds__(256) valu_kernel(float* out, int iters, float sb) {
  float b = 1.0f + threadIdx.x * 1e-9f, c = 1e-9f;
  f2 b2 = {b, b}, c2 = {c, c};
  float a[16]; f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; p[i] = f2{a[i], a[i]}; }
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_FMA_SGPR) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sb), "v"(c));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_ADD32) {
#define X(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(b2), "v"(c2));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_FMAMIX) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_CVTUB) {
#define X(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(b));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_PKADD32) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_PKADD16) {
#define X(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if constexpr (OP == OP_PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP16(X)
#undef X
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int OP>
static void run_valu(float* d_out, int waves_per_simd) {
  int ncu = 256;
  int blocks = ncu * waves_per_simd;  // 256 thr = 4 waves = 1 wave/SIMD per block
  int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  valu_kernel<OP><<<blocks, 256>>>(d_out, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  valu_kernel<OP><<<blocks, 256>>>(d_out, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double winstr = (double)blocks * 4 * iters * 16;         // wave-instructions
  double per_simd_per_s = winstr / (ms * 1e-3) / (ncu * 4); // wave-instr per second per SIMD
  printf("{\"ubench\":\"valu\",\"op\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.3f,\"Gwinstr_per_s_per_simd\":%.4f,\"ns_per_winstr\":%.4f,\"cyc_at_2.4GHz\":%.3f}\n",
         op_name[OP], waves_per_simd, ms, per_simd_per_s * 1e-9, 1e9 / per_simd_per_s, 2.4e9 / per_simd_per_s);
}

// ---------------- HBM / L3 read ceiling ----------------
__global__ void __launch_bounds__(256) read_kernel(const uint4* __restrict__ in, size_t n16, unsigned* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) { uint4 a = in[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x12345679u) out[0] = acc;
}

static void run_read(size_t bytes, int blocks) {
  uint4* d; unsigned* o;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4));
  CK(hipMemset(d, 0x5a, bytes));
  size_t n16 = bytes / 16;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) read_kernel<<<blocks, 256>>>(d, n16, o);
  CK(hipDeviceSynchronize());
  int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) read_kernel<<<blocks, 256>>>(d, n16, o);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"ubench\":\"read\",\"bytes\":%zu,\"blocks\":%d,\"ms_per_pass\":%.4f,\"GBps\":%.1f}\n", bytes, blocks, ms / reps,
         bytes / (ms / reps * 1e-3) * 1e-9);
  CK(hipFree(d)); CK(hipFree(o));
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\":\"%s\",\"arch\":\"%s\",\"cus\":%d,\"clock_khz\":%d,\"mem_clock_khz\":%d,\"l2\":%d}\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate, prop.memoryClockRate, prop.l2CacheSize);
  float* d_out; CK(hipMalloc(&d_out, 1024));
  for (int w : {1, 2, 4, 8}) {
    run_valu<OP_FMA>(d_out, w); run_valu<OP_PKFMA>(d_out, w); run_valu<OP_FMAMIX>(d_out, w); run_valu<OP_FMA_SGPR>(d_out, w);
    run_valu<OP_ADD32>(d_out, w); run_valu<OP_CVTUB>(d_out, w); run_valu<OP_PKADD32>(d_out, w); run_valu<OP_PKADD16>(d_out, w);
    run_valu<OP_PERM>(d_out, w);
  }
  for (int blocks : {1024, 2048, 4096}) {
    run_read((size_t)122880000, blocks);
    run_read((size_t)1228800000, blocks);
  }
  return 0;
}
