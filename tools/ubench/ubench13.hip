// v_smfmac_i32_16x16x128_i8 on gfx950: operand layout and rate.  Design Q's A operand (the tap Toeplitz slice) is exactly 2:4 sparse —
// an I row holds taps at even K positions only, a Q row at odd ones — so the sparse instruction covers 128 K bytes per issue.
// Assumed layout (checked here against a CPU product):
//   A (sparse): lane l = row (l & 15), K group g = l >> 4 covers dense K 32 g .. 32 g + 31 as 8 groups of 4; the lane's 16 bytes are the
//               two kept values of each group in order; idx (one VGPR) holds, per group q, bits [4q+1:4q] = position of the first kept
//               value and bits [4q+3:4q+2] = position of the second (first < second);
//   B (dense):  lane l = column (l & 15), 32 bytes = K 32 g .. 32 g + 31;
//   C / D:      row 4 (l >> 4) + r, column l & 15 in register r (as every 16x16 MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));

__global__ void k_layout(const i4* a, const i8v* b, const int* idx, i4* d) {
  const int l = threadIdx.x;
  i4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a[l], b[l], acc, idx[l], 0, 0);
  d[l] = acc;
}

template <int SPARSE, int FILL>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, int seed) {
  i4 a = {seed * 3 + (int)threadIdx.x, seed, seed * 7, seed + 5};
  i8v b = {seed + 11, (int)threadIdx.x, seed * 5, seed * 13, seed, seed + 1, seed + 2, seed + 3};
  i4 b4 = {seed + 11, (int)threadIdx.x, seed * 5, seed * 13};
  const int idx = (threadIdx.x & 1) ? 0xDDDDDDDD : 0x88888888;
  i4 acc[3];
  unsigned u[8];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = i4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 0x01010101u + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 15; ++m) {
      if constexpr (SPARSE) asm volatile("v_smfmac_i32_16x16x128_i8 %0, %1, %2, %3" : "+v"(acc[m % 3]) : "v"(a), "v"(b), "v"(idx));
      else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[m % 3]) : "v"(a), "v"(b4));
#pragma unroll
      for (int f = 0; f < FILL; ++f) asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(u[(m * FILL + f) & 7]));
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (float)(acc[i].x + acc[i].y + acc[i].z + acc[i].w);
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)u[i];
  if (s == 12345.678f) out[0] = s;
}

template <typename KT>
static void run(const char* name, int fill, KT kern, float* d_out, int wps) {
  const int blocks = 256 * wps, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, 256>>>(d_out, 500, 3);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<blocks, 256>>>(d_out, iters, 3);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"instr\":\"%s\",\"xor_per_instr\":%d,\"waves_per_simd\":%d,\"ns_per_instr_per_simd\":%.3f}\n", name, fill, wps, ms * 1e6 / ((double)wps * iters * 15));
  fflush(stdout);
}

int main() {
  std::vector<int8_t> A(16 * 128, 0), B(128 * 16);
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int8_t)(x >> 33); };
  for (auto& v : B) v = rnd();
  // 2:4 sparse A with design Q's pattern: even rows keep positions {0, 2} of every group of 4, odd rows {1, 3}; plus a second test with
  // random position pairs per group
  for (int variant = 0; variant < 2; ++variant) {
    std::vector<int8_t> ha(64 * 16), hb(64 * 32);
    std::vector<int> hidx(64, 0);
    for (auto& v : A) v = 0;
    for (int l = 0; l < 64; ++l) {
      const int row = l & 15, g = l >> 4;
      unsigned idx = 0;
      for (int q = 0; q < 8; ++q) {
        int p0 = (row & 1) ? 1 : 0, p1 = (row & 1) ? 3 : 2;
        if (variant == 1) { p0 = (int)((x >> 20) % 3); x ^= x << 13; x ^= x >> 7; x ^= x << 17; p1 = p0 + 1 + (int)((x >> 20) % (3 - p0)); x ^= x << 13; x ^= x >> 7; x ^= x << 17; }
        const int8_t v0 = rnd(), v1 = rnd();
        A[row * 128 + 32 * g + 4 * q + p0] = v0;
        A[row * 128 + 32 * g + 4 * q + p1] = v1;
        ha[l * 16 + 2 * q] = v0; ha[l * 16 + 2 * q + 1] = v1;
        idx |= (unsigned)(p0 | (p1 << 2)) << (4 * q);
      }
      hidx[l] = (int)idx;
      for (int p = 0; p < 32; ++p) hb[l * 32 + p] = B[(32 * g + p) * 16 + (l & 15)];
    }
    i4 *da, *dd; i8v* db; int* di;
    CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dd, 1024)); CK(hipMalloc(&di, 256));
    CK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, hidx.data(), 256, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(da, db, di, dd);
    std::vector<int> hd(256);
    CK(hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        int ref = 0;
        for (int k = 0; k < 128; ++k) ref += (int)A[row * 128 + k] * (int)B[k * 16 + col];
        bad += (ref != hd[l * 4 + r]);
      }
    printf("{\"probe\":\"layout v_smfmac_i32_16x16x128_i8\",\"pattern\":\"%s\",\"mismatches\":%d,\"of\":256}\n", variant ? "random pairs" : "design Q (even / odd positions)", bad);
  }
  float* d_out; CK(hipMalloc(&d_out, 1024));
  for (int wps : {1, 2, 4}) {
    run("v_mfma_i32_16x16x64_i8", 0, k_rate<0, 0>, d_out, wps);
    run("v_smfmac_i32_16x16x128_i8", 0, k_rate<1, 0>, d_out, wps);
    run("v_mfma_i32_16x16x64_i8", 4, k_rate<0, 4>, d_out, wps);
    run("v_smfmac_i32_16x16x128_i8", 4, k_rate<1, 4>, d_out, wps);
    run("v_smfmac_i32_16x16x128_i8", 8, k_rate<1, 8>, d_out, wps);
  }
  return 0;
}
