// tools/ldsbank/ldsbank.hip — what an LDS access pattern costs on gfx950, measured: one dispatch per pattern (64 per-lane byte addresses
// and an operation), every wave repeats the one instruction ITERS times; under `rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS` the ratio of
// the two counters per dispatch is the LDS-array cycles that instruction takes (conflict-free ds_read_b64: 2 ...).  Patterns come from a text
// file written by tools/ldsbank/patterns.py: one line per pattern, `op a0 a1 ... a63` (op 0 = ds_read_b64, 1 = ds_write_b64,
// 2 = ds_read_b128, 3 = ds_write_b128, 4 = ds_read_b32, 5 = ds_write_b32, 6 = ds_read2_b64 offset1:1, 7 = ds_write2_b32 offset1:1; addresses in bytes).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ldsbank/ldsbank tools/ldsbank/ldsbank.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int OP>
__global__ void __launch_bounds__(64) k_pat(const int* __restrict__ addr, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = (int)threadIdx.x;
  const unsigned a = (unsigned)addr[lane] + (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  float acc = 0.f;
  f2 v2 = {1.f, 2.f};
  f4 v4 = {1.f, 2.f, 3.f, 4.f};
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) { f2 r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += 0.f * r.x; }
      if (OP == 1) asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v2) : "memory");
      if (OP == 2) { f4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += 0.f * r.x; }
      if (OP == 3) asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v4) : "memory");
      if (OP == 4) { float r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += 0.f * r; }
      if (OP == 5) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v2.x) : "memory");
      if (OP == 6) { f4 r; asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(r) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += 0.f * r.x; }
      if (OP == 7) asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(a), "v"(v2.x), "v"(v2.y) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 123.f) sink[lane] = acc;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: ldsbank patterns.txt [iters] [workgroups]\n"); return 2; }
  const int iters = argc > 2 ? atoi(argv[2]) : 512, wgs = argc > 3 ? atoi(argv[3]) : 1024;
  FILE* f = fopen(argv[1], "r");
  if (!f) { perror(argv[1]); return 2; }
  std::vector<int> ops, addrs;
  int op;
  while (fscanf(f, "%d", &op) == 1) {
    ops.push_back(op);
    for (int l = 0; l < 64; ++l) { int a; if (fscanf(f, "%d", &a) != 1) { fprintf(stderr, "short line\n"); return 2; } addrs.push_back(a); }
  }
  fclose(f);
  int* d_addr; float* d_sink;
  (void)hipMalloc(&d_addr, addrs.size() * sizeof(int)); (void)hipMalloc(&d_sink, 256);
  (void)hipMemcpy(d_addr, addrs.data(), addrs.size() * sizeof(int), hipMemcpyHostToDevice);
  void (*ks[8])(const int*, int, float*) = {k_pat<0>, k_pat<1>, k_pat<2>, k_pat<3>, k_pat<4>, k_pat<5>, k_pat<6>, k_pat<7>};
  for (int k = 0; k < 8; ++k) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ks[k]), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (size_t p = 0; p < ops.size(); ++p) {
    hipLaunchKernelGGL(ks[ops[p]], dim3(wgs), dim3(64), 40960, 0, d_addr + 64 * p, iters, d_sink);   // 40 KB: 4 workgroups (one per SIMD) per CU
  }
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "failed\n"); return 1; }
  printf("%zu patterns dispatched\n", ops.size());
  return 0;
}
