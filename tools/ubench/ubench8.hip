// Mixed-precision FMA pricing on gfx950: can a FIR tap multiply read its sample as an f16 half (exact for b - 127.5) and
// still accumulate in f32 at the full scalar-FMA rate?  Also checks that v_fma_mix_f32 == fmaf(tap, (float)half, acc)
// bit for bit and that a D16 typed buffer load returns (half)byte.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ h4 llvm_raw_buffer_load_format_v4f16(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f16");
#define RSRC_U8X4_USCALED 0x52FAC

#define K32(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters, float sb) {                   \
    float b = 1.0f + threadIdx.x * 1e-9f, c = __int_as_float(0x3c003800 + threadIdx.x); float a[16];  \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;                         \
    for (int it = 0; it < iters; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; ++i)                                               \
        asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(sb));                                  \
    }                                                                                               \
    float s = 0.f; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i];                       \
    if (s == 12345.678f) out[0] = s;                                                                \
  }
K32(k_fma_vvv, "v_fma_f32 %0, %1, %2, %0")
K32(k_mix_vlo, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]")
K32(k_mix_vhi, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]")
K32(k_mix_slo, "v_fma_mix_f32 %0, %3, %2, %0 op_sel_hi:[0,1,0]")
K32(k_mix_f32, "v_fma_mix_f32 %0, %1, %2, %0")
K32(k_pkaddf16, "v_pk_add_f16 %0, %1, %0")
K32(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %1")

__global__ void check_mix(const float* taps, const unsigned* halves, const float* acc, float* out_lo, float* out_hi, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = taps[i], a = acc[i], lo = a, hi = a; unsigned h = halves[i];
  asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(lo) : "v"(t), "v"(h));
  asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(hi) : "v"(t), "v"(h));
  out_lo[i] = lo; out_hi[i] = hi;
}
__global__ void check_d16(const unsigned char* in, _Float16* out, unsigned n) {
  unsigned long long a = (unsigned long long)in;
  i4 r = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)n, RSRC_U8X4_USCALED};
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 < n) {
    h4 v = llvm_raw_buffer_load_format_v4f16(r, i * 4, 0, 0);
    v = v - (h4){(_Float16)127.5f, (_Float16)127.5f, (_Float16)127.5f, (_Float16)127.5f};
    out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
  }
}

template <typename KT>
static void run(const char* name, KT kern, float* d_out, int wps) {
  int blocks = 256 * wps, iters = 100000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(d_out, 20000, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<blocks, 256>>>(d_out, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double winstr = (double)blocks * 4 * iters * 16;
  double r = winstr / (ms * 1e-3) / 1024.0;
  printf("{\"op\":\"%s\",\"wps\":%d,\"ms\":%.2f,\"Gwinstr_s_simd\":%.4f,\"ns_per_winstr\":%.3f}\n", name, wps, ms, r * 1e-9, 1e9 / r);
  fflush(stdout);
}
#define R32(k) run(#k, k, d_out, wps)

static float half_to_float(unsigned short h) { __half_raw r; r.x = h; return __half2float(__half(r)); }

int main() {
  float* d_out; CK(hipMalloc(&d_out, 1024));
  // ---- exactness of v_fma_mix_f32 against fmaf(tap, (float)half, acc) ----
  const int n = 1 << 16;
  std::vector<float> taps(n), acc(n), lo(n), hi(n); std::vector<unsigned> hv(n);
  srand(7);
  for (int i = 0; i < n; ++i) {
    taps[i] = (float)((rand() % 20001) - 10000) / 65536.0f * (1.0f + (rand() % 1000) * 1e-6f);
    acc[i] = (float)((rand() % 2000001) - 1000000) / 997.0f;
    unsigned short a = __half_as_ushort(__float2half((float)(rand() % 256) - 127.5f)), b = __half_as_ushort(__float2half((float)(rand() % 256) - 127.5f));
    hv[i] = (unsigned)a | ((unsigned)b << 16);
  }
  float *d_t, *d_a, *d_lo, *d_hi; unsigned* d_h;
  CK(hipMalloc(&d_t, 4 * n)); CK(hipMalloc(&d_a, 4 * n)); CK(hipMalloc(&d_lo, 4 * n)); CK(hipMalloc(&d_hi, 4 * n)); CK(hipMalloc(&d_h, 4 * n));
  CK(hipMemcpy(d_t, taps.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(d_a, acc.data(), 4 * n, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_h, hv.data(), 4 * n, hipMemcpyHostToDevice));
  check_mix<<<n / 256, 256>>>(d_t, d_h, d_a, d_lo, d_hi, n);
  CK(hipMemcpy(lo.data(), d_lo, 4 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hi.data(), d_hi, 4 * n, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    float xl = half_to_float((unsigned short)(hv[i] & 0xffff)), xh = half_to_float((unsigned short)(hv[i] >> 16));
    float wl = fmaf(taps[i], xl, acc[i]), wh = fmaf(taps[i], xh, acc[i]);
    if (memcmp(&wl, &lo[i], 4) || memcmp(&wh, &hi[i], 4)) ++bad;
  }
  printf("{\"check\":\"v_fma_mix_f32 == fmaf(tap,(float)half,acc)\",\"n\":%d,\"mismatches\":%d}\n", n, bad);
  // ---- D16 typed load: (half)byte - 127.5 exact? ----
  std::vector<unsigned char> bytes(1024); for (int i = 0; i < 1024; ++i) bytes[i] = (unsigned char)(i * 7 + (i >> 8));
  unsigned char* d_b; _Float16* d_o; CK(hipMalloc(&d_b, 1024)); CK(hipMalloc(&d_o, 2048));
  CK(hipMemcpy(d_b, bytes.data(), 1024, hipMemcpyHostToDevice));
  check_d16<<<1, 256>>>(d_b, d_o, 1024);
  std::vector<unsigned short> oh(1024); CK(hipMemcpy(oh.data(), d_o, 2048, hipMemcpyDeviceToHost));
  bad = 0;
  for (int i = 0; i < 1024; ++i) if (half_to_float(oh[i]) != (float)bytes[i] - 127.5f) ++bad;
  printf("{\"check\":\"buffer_load_format_d16_xyzw(8_8_8_8 USCALED) - 127.5h == byte - 127.5\",\"n\":1024,\"mismatches\":%d}\n", bad);
  for (int wps : {4, 8}) {
    R32(k_fma_vvv); R32(k_mix_vlo); R32(k_mix_vhi); R32(k_mix_slo); R32(k_mix_f32); R32(k_pkaddf16); R32(k_cvt_f32_f16);
  }
  return 0;
}
