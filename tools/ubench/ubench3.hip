// u8 -> f32 conversion paths priced as streaming kernels: typed buffer loads (conversion in the texture unit)
// vs plain loads + v_cvt_f32_ubyteN.  Also checks that the typed path returns exactly (float)byte on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ f4 llvm_raw_buffer_load_format_v4f32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
// word3: dst_sel xyzw = R,G,B,A ; num_format = USCALED(2) ; data_format = 8_8_8_8 (10)
#define RSRC_U8X4_USCALED 0x52FAC

__device__ __forceinline__ i4 make_rsrc(const void* p, unsigned bytes) {
  unsigned long long a = (unsigned long long)p;
  return i4{(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, RSRC_U8X4_USCALED};
}

__global__ void check_kernel(const unsigned char* in, float* out, unsigned n) {
  i4 r = make_rsrc(in, n);
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 < n) {
    f4 v = llvm_raw_buffer_load_format_v4f32(r, i * 4, 0, 0);
    out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
  }
}

// each block handles a contiguous chunk; 4 typed loads in flight per thread per iteration
template <int MODE>
__global__ void __launch_bounds__(256) stream_kernel(const unsigned char* __restrict__ in, size_t bytes_per_block, float* out) {
  const unsigned char* base = in + (size_t)blockIdx.x * bytes_per_block;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if constexpr (MODE == 0) {          // typed xyzw: 4 B per lane per load
    i4 r = make_rsrc(base, (unsigned)bytes_per_block);
    for (unsigned off = threadIdx.x * 4; off < bytes_per_block; off += 256 * 4 * 4) {
      f4 a = llvm_raw_buffer_load_format_v4f32(r, off, 0, 0);
      f4 b = llvm_raw_buffer_load_format_v4f32(r, off + 1024, 0, 0);
      f4 c = llvm_raw_buffer_load_format_v4f32(r, off + 2048, 0, 0);
      f4 d = llvm_raw_buffer_load_format_v4f32(r, off + 3072, 0, 0);
      a -= 127.5f; b -= 127.5f; c -= 127.5f; d -= 127.5f;
      s0 += a.x + b.x + c.x + d.x; s1 += a.y + b.y + c.y + d.y; s2 += a.z + b.z + c.z + d.z; s3 += a.w + b.w + c.w + d.w;
    }
  } else if constexpr (MODE == 1) {   // plain dword + cvt
    const unsigned* p = reinterpret_cast<const unsigned*>(base);
    for (unsigned i = threadIdx.x; i < bytes_per_block / 4; i += 256 * 4) {
      unsigned w[4] = {p[i], p[i + 256], p[i + 512], p[i + 768]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s0 += (float)(w[k] & 0xff) - 127.5f; s1 += (float)((w[k] >> 8) & 0xff) - 127.5f;
        s2 += (float)((w[k] >> 16) & 0xff) - 127.5f; s3 += (float)(w[k] >> 24) - 127.5f;
      }
    }
  } else {                            // dwordx4 + cvt
    const uint4* p = reinterpret_cast<const uint4*>(base);
    for (unsigned i = threadIdx.x; i < bytes_per_block / 16; i += 256) {
      uint4 q = p[i];
      unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s0 += (float)(w[k] & 0xff) - 127.5f; s1 += (float)((w[k] >> 8) & 0xff) - 127.5f;
        s2 += (float)((w[k] >> 16) & 0xff) - 127.5f; s3 += (float)(w[k] >> 24) - 127.5f;
      }
    }
  }
  float s = s0 + s1 + s2 + s3;
  if (s == 1234.5678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, const unsigned char* d, size_t bytes, int blocks, float* o) {
  size_t bpb = bytes / blocks;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) stream_kernel<MODE><<<blocks, 256>>>(d, bpb, o);
  CK(hipDeviceSynchronize());
  int reps = 50;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) stream_kernel<MODE><<<blocks, 256>>>(d, bpb, o);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"ubench\":\"u8_to_f32_stream\",\"path\":\"%s\",\"bytes\":%zu,\"blocks\":%d,\"us_per_pass\":%.2f,\"input_GBps\":%.1f}\n", name, bytes, blocks,
         ms / reps * 1e3, bytes / (ms / reps * 1e-3) * 1e-9);
}

int main() {
  size_t bytes = 122880000;
  unsigned char* d; float* o;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 1 << 20));
  std::vector<unsigned char> h(bytes);
  unsigned x = 12345;
  for (size_t i = 0; i < bytes; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned char)(x >> 24); }
  CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
  // correctness of the typed path on the first 64 KiB
  unsigned n = 65536;
  check_kernel<<<n / 4 / 256, 256>>>(d, o, n);
  std::vector<float> r(n);
  CK(hipMemcpy(r.data(), o, n * sizeof(float), hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (unsigned i = 0; i < n; ++i) bad += (r[i] != (float)h[i]);
  printf("{\"ubench\":\"typed_load_check\",\"bytes\":%u,\"mismatches\":%zu,\"first\":[%.1f,%.1f,%.1f,%.1f],\"expect\":[%d,%d,%d,%d]}\n", n, bad, r[0], r[1], r[2], r[3],
         h[0], h[1], h[2], h[3]);
  for (int blocks : {1920, 3840, 7680}) {
    run<0>("typed_xyzw", d, bytes, blocks, o);
    run<1>("dword+cvt", d, bytes, blocks, o);
    run<2>("dwordx4+cvt", d, bytes, blocks, o);
  }
  return 0;
}
