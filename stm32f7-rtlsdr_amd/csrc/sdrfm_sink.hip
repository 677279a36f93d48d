/*
 * sdrfm_sink.hip — device-side audio sink for the batched path: demodulated audio (f32, 48 kHz) ->
 * FM de-emphasis -> int16 stereo-interleaved PCM in the layout the reference board's sink consumes,
 * BSP_AUDIO_OUT_Play(uint16_t* pBuffer, uint32_t Size) (Utilities/STM32746G-Discovery/stm32746g_discovery_audio.c:224;
 * L and R carry the same mono programme).  Same arithmetic, operation for operation, as the host routine
 * sdrfm_pcm_deemph_s16 (csrc/pcm_sink.c), so the two are interchangeable bit for bit:
 *
 *   y[n]   = fmaf(alpha, x[n] - y[n-1], y[n-1])        one rounded difference, one fused multiply-add
 *   pcm[n] = (int16) rint(clamp(y[n] * gain, -32768, 32767))      round-half-even, L = R
 *
 * Parallelisation: the recursion rounds at every step, so a scan (which re-associates) cannot reproduce it bit for bit;
 * the chain of one stream is therefore walked by ONE lane, 64 streams per wave, and what IS parallel — moving the data —
 * is kept coalesced by transposing 64 x 64 tiles through LDS:
 *
 *   HBM --row r: 64 lanes x 4 B, 256 B coalesced--> LDS tile[r][t] (row stride 65 words: conflict-free both ways)
 *   lane = stream: 64 dependent steps from LDS, packed (L | R << 16) back into the tile in place
 *   LDS --row r--> HBM 256 B coalesced stores of the interleaved int16 pairs
 *
 * The chain is latency-bound (sub -> fma per sample): ~4800 samples x ~10 cycles ~ 25 us per launch at any stream count
 * up to 64 per CU-resident wave; it runs on its own stream beside the next batch's demodulation.
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/sdrfm.h"

namespace {

struct SinkParams {
  const float* audio;
  size_t audio_stride;   // floats
  int16_t* pcm;
  size_t pcm_stride;     // int16 elements per stream (>= 2 * n)
  float* state;          // [n_streams] y[n-1]
  uint32_t n_streams, n;
  float alpha, gain;
};

__global__ void __launch_bounds__(64) k_pcm_sink(SinkParams p) {
  __shared__ unsigned tile[64 * 65];
  const uint32_t lane = threadIdx.x;
  const uint32_t s0 = blockIdx.x * 64;
  const uint32_t rows = (p.n_streams - s0 < 64u) ? p.n_streams - s0 : 64u;
  const uint32_t mine = s0 + lane;
  float y = (lane < rows) ? p.state[mine] : 0.0f;
  for (uint32_t t0 = 0; t0 < p.n; t0 += 64) {
    const uint32_t cols = (p.n - t0 < 64u) ? p.n - t0 : 64u;
    if (lane < cols)
      for (uint32_t r = 0; r < rows; ++r)
        tile[r * 65 + lane] = __float_as_uint(p.audio[(size_t)(s0 + r) * p.audio_stride + t0 + lane]);
    __syncthreads();
    if (lane < rows) {
      for (uint32_t i = 0; i < cols; ++i) {
        const float x = __uint_as_float(tile[lane * 65 + i]);
        y = __builtin_fmaf(p.alpha, x - y, y);
        float v = y * p.gain;
        if (v > 32767.0f) v = 32767.0f;
        if (v < -32768.0f) v = -32768.0f;
        const unsigned s = (unsigned)(int)__builtin_rintf(v) & 0xffffu;
        tile[lane * 65 + i] = s | (s << 16);
      }
    }
    __syncthreads();
    if (lane < cols)
      for (uint32_t r = 0; r < rows; ++r)
        reinterpret_cast<unsigned*>(p.pcm + (size_t)(s0 + r) * p.pcm_stride)[t0 + lane] = tile[r * 65 + lane];
    __syncthreads();
  }
  if (lane < rows) p.state[mine] = y;
}

}  // namespace

struct sdrfm_pcm_sink {
  uint32_t n_streams;
  float alpha, gain;
  int device;
  hipStream_t own_stream, stream;
  float* d_state;
  float* d_audio;      // staging for host-pointer calls
  int16_t* d_pcm;
  uint32_t cap;        // samples per stream the staging holds
};

#define STRY(expr, code)                                                                                       \
  do {                                                                                                         \
    hipError_t e__ = (expr);                                                                                   \
    if (e__ != hipSuccess) {                                                                                   \
      fprintf(stderr, "[sdrfm] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__);  \
      return (code);                                                                                           \
    }                                                                                                          \
  } while (0)

static void sink_free(sdrfm_pcm_sink* k) {
  if (!k) return;
  (void)hipSetDevice(k->device);
  if (k->d_state) (void)hipFree(k->d_state);
  if (k->d_audio) (void)hipFree(k->d_audio);
  if (k->d_pcm) (void)hipFree(k->d_pcm);
  if (k->own_stream) (void)hipStreamDestroy(k->own_stream);
  delete k;
}

extern "C" {

int sdrfm_pcm_sink_create(uint32_t n_streams, float alpha, float gain, int32_t device, sdrfm_pcm_sink_t** out) {
  if (!out) return SDRFM_EINVAL;
  *out = nullptr;
  if (!n_streams || !(alpha > 0.0f) || alpha > 1.0f || !(gain == gain)) return SDRFM_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return SDRFM_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SDRFM_NO_DEVICE;
  STRY(hipSetDevice(device), SDRFM_NO_DEVICE);
  sdrfm_pcm_sink* k = new (std::nothrow) sdrfm_pcm_sink();
  if (!k) return SDRFM_ENOMEM;
  memset(static_cast<void*>(k), 0, sizeof(*k));
  k->n_streams = n_streams; k->alpha = alpha; k->gain = gain; k->device = device;
  if (hipStreamCreateWithFlags(&k->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&k->d_state, sizeof(float) * n_streams) != hipSuccess) { sink_free(k); return SDRFM_ENOMEM; }
  k->stream = k->own_stream;
  const int rc = sdrfm_pcm_sink_reset(k);
  if (rc != SDRFM_OK) { sink_free(k); return rc; }
  *out = k;
  return SDRFM_OK;
}

void sdrfm_pcm_sink_destroy(sdrfm_pcm_sink_t* k) {
  if (!k) return;
  (void)hipSetDevice(k->device);
  (void)hipStreamSynchronize(k->stream);
  sink_free(k);
}

int sdrfm_pcm_sink_reset(sdrfm_pcm_sink_t* k) {
  if (!k) return SDRFM_EINVAL;
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  STRY(hipMemsetAsync(k->d_state, 0, sizeof(float) * k->n_streams, k->stream), SDRFM_FAIL);
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

int sdrfm_pcm_sink_set_stream(sdrfm_pcm_sink_t* k, void* hip_stream) {
  if (!k) return SDRFM_EINVAL;
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  k->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : k->own_stream;
  return SDRFM_OK;
}

int sdrfm_pcm_sink_synchronize(sdrfm_pcm_sink_t* k) {
  if (!k) return SDRFM_EINVAL;
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

int sdrfm_pcm_sink_process_batch(sdrfm_pcm_sink_t* k, const float* audio, size_t audio_stride, uint32_t n, int16_t* pcm,
                                 size_t pcm_stride, uint32_t flags) {
  if (!k) return SDRFM_EINVAL;
  if (flags & ~SDRFM_F_DEVICE_PTRS) return SDRFM_EINVAL;
  if (n == 0) return SDRFM_OK;
  if (!audio || !pcm) return SDRFM_EINVAL;
  if (k->n_streams > 1 && (audio_stride < n || pcm_stride < 2 * (size_t)n)) return SDRFM_ECAPACITY;
  if (pcm_stride & 1u) return SDRFM_EINVAL;                          // rows are written as (L,R) dwords
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  SinkParams p;
  p.state = k->d_state; p.n_streams = k->n_streams; p.n = n; p.alpha = k->alpha; p.gain = k->gain;
  const dim3 grid((k->n_streams + 63) / 64);
  if (flags & SDRFM_F_DEVICE_PTRS) {
    if ((uintptr_t)pcm % 4 != 0) return SDRFM_EINVAL;
    p.audio = audio; p.audio_stride = audio_stride; p.pcm = pcm; p.pcm_stride = pcm_stride;
    hipLaunchKernelGGL(k_pcm_sink, grid, dim3(64), 0, k->stream, p);
    STRY(hipGetLastError(), SDRFM_FAIL);
    return SDRFM_OK;
  }
  // host buffers: stage, run, copy back, synchronous
  if (n > k->cap) {
    if (k->d_audio) (void)hipFree(k->d_audio);
    if (k->d_pcm) (void)hipFree(k->d_pcm);
    k->d_audio = nullptr; k->d_pcm = nullptr; k->cap = 0;
    const uint32_t cap = (n + 1023u) & ~1023u;
    if (hipMalloc(&k->d_audio, sizeof(float) * (size_t)cap * k->n_streams) != hipSuccess ||
        hipMalloc(&k->d_pcm, sizeof(int16_t) * 2 * (size_t)cap * k->n_streams) != hipSuccess) return SDRFM_ENOMEM;
    k->cap = cap;
  }
  const size_t as = (k->n_streams > 1) ? audio_stride : n, ps = (k->n_streams > 1) ? pcm_stride : 2 * (size_t)n;
  STRY(hipMemcpy2DAsync(k->d_audio, sizeof(float) * k->cap, audio, sizeof(float) * as, sizeof(float) * n, k->n_streams,
                        hipMemcpyHostToDevice, k->stream), SDRFM_FAIL);
  p.audio = k->d_audio; p.audio_stride = k->cap; p.pcm = k->d_pcm; p.pcm_stride = 2 * (size_t)k->cap;
  hipLaunchKernelGGL(k_pcm_sink, grid, dim3(64), 0, k->stream, p);
  STRY(hipGetLastError(), SDRFM_FAIL);
  STRY(hipMemcpy2DAsync(pcm, sizeof(int16_t) * ps, k->d_pcm, sizeof(int16_t) * 2 * k->cap, sizeof(int16_t) * 2 * n, k->n_streams,
                        hipMemcpyDeviceToHost, k->stream), SDRFM_FAIL);
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

/* Host copy of the carried de-emphasis state y[n-1] of every stream (tests, and switching between this sink and the host one). */
int sdrfm_pcm_sink_get_state(sdrfm_pcm_sink_t* k, float* state_out) {
  if (!k || !state_out) return SDRFM_EINVAL;
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  STRY(hipMemcpy(state_out, k->d_state, sizeof(float) * k->n_streams, hipMemcpyDeviceToHost), SDRFM_FAIL);
  return SDRFM_OK;
}

}  // extern "C"
