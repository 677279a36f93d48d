// What does the HBM deliver to an LDS-DMA stream (buffer_load_dwordx4 ... lds, 1 KiB per instruction, NSLOT KiB in flight per wave,
// one-wave workgroups, no compute) as a function of WHERE the concurrently running waves read?
//   pattern 0 "private regions": wave w streams its own contiguous region [w R, (w + 1) R) — design Q's run per wave;
//   pattern 1 "sweep": the waves of a group of G read consecutive KiB of one region together: wave (g, i) reads chunks i, i + G, ...
//                      of the group's region (G = 1 is pattern 0);
//   pattern 2 "item sweep": consecutive waves take consecutive ITEMS of `item` KiB: wave w reads items w, w + W, w + 2 W, ...
// Cold input: the buffer (default 512 MiB) is larger than the Infinity Cache and every launch starts at a different offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ void dma(i4 rsrc, __attribute__((address_space(3))) void* lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

template <int NSLOT, int AUX>
__global__ void __launch_bounds__(64) k_stream(const uint8_t* base, unsigned total_kib, unsigned kib_per_wave, int pattern, unsigned G, unsigned item, unsigned* sink, unsigned long long* stamps) {
  const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned w = blockIdx.x, W = gridDim.x, lane = threadIdx.x;
  const unsigned long long ga = (unsigned long long)base;
  const i4 rsrc = {(int)(unsigned)ga, (int)(unsigned)(ga >> 32), (int)(total_kib << 10), 0x00020000};
  auto chunk_addr = [&](unsigned q) -> unsigned {            // KiB index of this wave's q-th chunk
    if (pattern == 0) return w * kib_per_wave + q;
    if (pattern == 1) { const unsigned grp = w / G, i = w % G; return grp * G * kib_per_wave + q * G + i; }
    const unsigned it = q / item, o = q % item; return (w + it * W) * item + o;
  };
  unsigned issued = 0;
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) { dma(rsrc, (__attribute__((address_space(3))) void*)(smem + 1024 * q), 16, (chunk_addr(issued) << 10) + 16 * lane, 0, 0, AUX); ++issued; }
  unsigned acc = 0;
  int slot = 0;
  for (unsigned q = 0; q < kib_per_wave; ++q) {
    __builtin_amdgcn_s_waitcnt(0x0f70 | (NSLOT - 1));       // the oldest chunk has landed
    acc += *(const unsigned*)(smem + 1024 * slot + 4 * lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned nq = issued < kib_per_wave ? chunk_addr(issued) : 0x3fffffu;   // past the end: out of range, no traffic
    dma(rsrc, (__attribute__((address_space(3))) void*)(smem + 1024 * slot), 16, (nq << 10) + 16 * lane, 0, 0, AUX);
    ++issued;
    slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);
  if (stamps && threadIdx.x == 0) { stamps[3 * blockIdx.x] = t_in; stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); stamps[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NSLOT, int AUX>
static void run(const uint8_t* d, size_t buf_kib, unsigned waves, unsigned kib_per_wave, int pattern, unsigned G, unsigned item, unsigned* sink) {
  const unsigned total = waves * kib_per_wave;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = getenv("UB_ITERS") ? atoi(getenv("UB_ITERS")) : 20;
  size_t off = 0;
  auto launch = [&] {
    if (off + total > buf_kib) off = 0;
    hipLaunchKernelGGL((k_stream<NSLOT, AUX>), dim3(waves), dim3(64), NSLOT * 1024, 0, d + (off << 10), total, kib_per_wave, pattern, G, item, sink, (unsigned long long*)nullptr);
    off += total;
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  if (getenv("UB_STAMPS")) {   // one more launch with per-wave entry / exit times (100 MHz), summarised for XCC 0
    unsigned long long* ds; CK(hipMalloc(&ds, (size_t)waves * 24)); CK(hipMemset(ds, 0, (size_t)waves * 24));
    if (off + total > buf_kib) off = 0;
    hipLaunchKernelGGL((k_stream<NSLOT, AUX>), dim3(waves), dim3(64), NSLOT * 1024, 0, d + (off << 10), total, kib_per_wave, pattern, G, item, sink, ds);
    CK(hipDeviceSynchronize());
    unsigned long long* hs = (unsigned long long*)malloc((size_t)waves * 24);
    CK(hipMemcpy(hs, ds, (size_t)waves * 24, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0, tin_max = 0, tout_min = ~0ull;
    for (unsigned w = 0; w < waves; ++w) if (hs[3 * w + 2] == 0) { if (hs[3 * w] < t0) t0 = hs[3 * w]; if (hs[3 * w] > tin_max) tin_max = hs[3 * w]; if (hs[3 * w + 1] > t1) t1 = hs[3 * w + 1]; if (hs[3 * w + 1] < tout_min) tout_min = hs[3 * w + 1]; }
    printf("{\"stamps_xcc0_us\":{\"last_entry\":%.2f,\"first_exit\":%.2f,\"last_exit\":%.2f}}\n", (tin_max - t0) * 0.01, (tout_min - t0) * 0.01, (t1 - t0) * 0.01);
    free(hs); CK(hipFree(ds));
  }
  printf("{\"nslot\":%d,\"aux\":%d,\"waves\":%u,\"kib_per_wave\":%u,\"pattern\":%d,\"G\":%u,\"item_kib\":%u,\"MB\":%.1f,\"us_per_launch\":%.2f,\"TBps\":%.3f}\n",
         NSLOT, AUX, waves, kib_per_wave, pattern, G, item, total / 1024.0 * 1.048576, us, (double)total * 1024 / (us * 1e-6) * 1e-12);
  fflush(stdout);
}

int main() {
  const size_t buf_kib = 640 * 1024;
  uint8_t* d; CK(hipMalloc(&d, buf_kib << 10)); CK(hipMemset(d, 1, buf_kib << 10));
  unsigned* sink; CK(hipMalloc(&sink, 64));
  // ~120 MiB per launch like configs[2]
  for (unsigned waves : {2048u, 3072u, 4096u}) {
    const unsigned kpw = 120u * 1024u / waves;
    run<5, 0>(d, buf_kib, waves, kpw, 0, 1, 1, sink);
    run<5, 0>(d, buf_kib, waves, kpw, 1, 8, 1, sink);
    run<5, 0>(d, buf_kib, waves, kpw, 1, 64, 1, sink);
    run<5, 0>(d, buf_kib, waves, kpw, 2, 1, 5, sink);
    run<5, 0>(d, buf_kib, waves, kpw, 2, 1, 10, sink);
    run<5, 2>(d, buf_kib, waves, kpw, 0, 1, 1, sink);
    run<5, 2>(d, buf_kib, waves, kpw, 2, 1, 10, sink);
    run<10, 0>(d, buf_kib, waves, kpw, 0, 1, 1, sink);
    run<10, 0>(d, buf_kib, waves, kpw, 2, 1, 10, sink);
    run<10, 2>(d, buf_kib, waves, kpw, 2, 1, 10, sink);
  }
  // few long-lived waves with a deep ring
  run<15, 0>(d, buf_kib, 1024, 120, 0, 1, 1, sink);
  run<15, 0>(d, buf_kib, 1024, 120, 2, 1, 15, sink);
  run<15, 0>(d, buf_kib, 2048, 60, 2, 1, 15, sink);
  return 0;
}
