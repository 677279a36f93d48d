// LDS read rate vs access width and per-lane stride (one wave-instruction = 64 lanes x W bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
template <int W> struct VT; template <> struct VT<4> { typedef float t; }; template <> struct VT<8> { typedef f2_t t; }; template <> struct VT<16> { typedef f4_t t; };
template <int W, int NR>
__global__ void __launch_bounds__(64) rd(float* out, int iters, int stride, int used) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < used / 4; i += 64) reinterpret_cast<float*>(smem)[i] = (float)(i % 251);
  __syncthreads();
  unsigned poff = lane * stride;
  typedef typename VT<W>::t V;
  V acc = {};
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(poff) :: "memory");   // opaque OFFSET (keeps the LDS address space: ds_read, not flat)
    const unsigned char* p = smem + poff;
    V r[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) r[j] = *reinterpret_cast<const V*>(p + j * W);
#pragma unroll
    for (int j = 0; j < NR; ++j) acc += r[j];
  }
  float s; if constexpr (W == 4) s = acc; else if constexpr (W == 8) s = acc.x + acc.y; else s = acc.x + acc.y + acc.z + acc.w;
  if (s == 1234.5f) out[0] = s;
}
template <int W, int NR>
static void run(float* d_o, int stride, int wpc) {
  int used = 63 * stride + NR * W + 64; used = (used + 255) & ~255;
  int lds = 160 * 1024 / wpc; lds -= lds % 256;
  if (lds < used) return;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rd<W, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int blocks = 256 * wpc, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rd<W, NR><<<blocks, 64, lds>>>(d_o, 50, stride, used); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rd<W, NR><<<blocks, 64, lds>>>(d_o, iters, stride, used);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double reads_per_cu = (double)wpc * iters * NR;
  double ns = ms * 1e6 / reads_per_cu;
  printf("{\"width\":%d,\"lane_stride\":%d,\"waves_per_cu\":%d,\"ns_per_read_per_cu\":%.2f,\"bytes_per_ns_per_cu\":%.1f}\n", W, stride, wpc, ns, 64.0 * W / ns);
  fflush(stdout);
}
int main() {
  float* d_o; CK(hipMalloc(&d_o, 64));
  for (int wpc : {4, 8}) {
    for (int s : {16, 48, 80, 240, 336, 1040}) run<16, 16>(d_o, s, wpc);
    for (int s : {8, 24, 40, 168, 248, 1032}) run<8, 16>(d_o, s, wpc);
    for (int s : {4, 12, 20, 84, 1028}) run<4, 16>(d_o, s, wpc);
  }
  return 0;
}
