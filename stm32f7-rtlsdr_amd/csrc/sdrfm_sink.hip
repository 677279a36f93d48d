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
 * The chain is latency-bound (one lane per stream, and every 64 samples a tile of 64 dependent row loads): measured 1.35 ms per 256 x 4800
 * launch (profiles/r06_sink.txt; rounds 2 - 5 quoted an estimate of 25 us that had never been timed) — sixty demodulator calls.  It is kept
 * as the EXACT form (SDRFM_PCM_F_EXACT): bit-identical to the host routine.
 *
 * The default since round 6 is a BLOCKED SCAN (k_pcm_sink_scan): the recursion is linear — y[n] = (1 - alpha) y[n-1] + alpha x[n] —,
 * so a stream's call is cut into segments of 256 chunks of 19 samples, one lane per chunk, one workgroup of 256 lanes per stream:
 *   1. lane t walks its chunk from state 0 (lane 0: from the carried state) -> e[t], the chunk's own contribution to its last sample;
 *   2. the carries s[t] = (1 - alpha)^c s[t-1] + e[t]: six shuffle steps within each wave (the powers squared on the way), the four waves' totals combined through
 *      four words of LDS (one barrier; the first version's Hillis-Steele scan over 256 lanes in LDS took sixteen: 7.4 -> 7.0 us per launch);
 *   3. lane t walks its chunk AGAIN, now from its true carry-in s[t-1], with exactly the exact form's operations, and packs the PCM.
 * What differs from the exact chain is therefore only the carry-in of a chunk (re-associated: ~1e-7 relative), and that difference decays with
 * (1 - alpha)^k inside the chunk: the PCM is within 1 LSB of the exact form's (equal but where y * gain sits within 1e-3 of a rounding boundary),
 * the carried state within 2.5e-7 (tests/test_pcm_sink_gpu.py).  The audio row goes through LDS once (coalesced loads, chunk stride odd: conflict-free),
 * the PCM row back the same way; the chunk sits in registers for both walks.  7.0 us per 256 x 4800 launch against the exact form's 1.35 ms
 * (profiles/r06_sink.txt).
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/sdrfm.h"
#include "sdrfm_sink_chain.h"

namespace {

struct SinkParams {
  const float* audio;
  size_t audio_stride;   // floats
  int16_t* pcm;
  size_t pcm_stride;     // int16 elements per stream (>= 2 * n)
  const unsigned long long* sg_in;   // [n_streams] {tag << 32 | bits of y[n-1]}: the slot of the tag before this call (sdrfm_sink_chain.h) ...
  unsigned long long* sg_out;        // ... and of the tag behind it
  uint32_t gen_next;     // the tag behind this call
  uint32_t n_streams, n;
  float alpha, gain;
};

__global__ void __launch_bounds__(64) k_pcm_sink(SinkParams p) {
  __shared__ unsigned tile[64 * 65];
  const uint32_t lane = threadIdx.x;
  const uint32_t s0 = blockIdx.x * 64;
  const uint32_t rows = (p.n_streams - s0 < 64u) ? p.n_streams - s0 : 64u;
  const uint32_t mine = s0 + lane;
  float y = (lane < rows) ? __uint_as_float((unsigned)p.sg_in[mine]) : 0.0f;
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
  if (lane < rows) p.sg_out[mine] = ((unsigned long long)p.gen_next << 32) | __float_as_uint(y);
}

// ---- the blocked scan (the default): one workgroup of 256 lanes per stream; segments of SINK_NT * SINK_C samples through LDS ------------------------------
// (A variant with ONE WAVE per stream, 75-sample chunks in registers and no LDS at all was built to slip in beside the demodulator's waves, which hold all but
// 0.9 KiB of a CU's LDS: 10.5 us alone against 7.4 us for this one, and no faster in the consumer loop — profiles/r06_sink.txt.  Not kept.)
constexpr uint32_t SINK_NT = 256, SINK_C = 19, SINK_SEG = SINK_NT * SINK_C;
static_assert(SINK_NT == 256, "four waves: the carries between them are combined by hand");   // 4864 samples per segment (BASELINE's 4800 per call: one segment), 19 KiB of LDS;
                                                                             // lanes SINK_C = 19 words apart (odd): conflict-free LDS accesses
// LIST: the workgroup's stream is list[blockIdx.x], and the call is PART of a sink call whose other streams the chain inside a demodulator launch serves
// (sdrfm_sink_chain.h): the state is taken by that protocol — the tagged word of slot sg_in, waited for on the device (bounded) — and published the same way.
template <bool LIST>
__global__ void __launch_bounds__(256) k_pcm_sink_scan(SinkParams p, float pc, const uint32_t* list, uint32_t* err) {
  __shared__ float x[SINK_SEG];                                 // the segment's samples, then (in place) the packed PCM words
  __shared__ float sc[4];                                       // the waves' totals; then the segment's last state
  unsigned* const xw = reinterpret_cast<unsigned*>(x);
  const uint32_t s = LIST ? list[blockIdx.x] : blockIdx.x, t = threadIdx.x;
  const float* const row = p.audio + (size_t)s * p.audio_stride;
  unsigned* const out = reinterpret_cast<unsigned*>(p.pcm + (size_t)s * p.pcm_stride);
  float y0;                                                     // the state before the segment (every lane holds it)
  if constexpr (LIST) {
    if (t == 0) {
      unsigned long long pv = 0ull, zero = 0ull;
      asm volatile("" : "+v"(zero));                             // (an opaque 0: a read-modify-write the compiler cannot turn back into a load, which may hit a stale line)
      int it = 0;
      for (;; ++it) {
        pv = __hip_atomic_fetch_add(const_cast<unsigned long long*>(p.sg_in) + s, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(pv >> 32) == p.gen_next - 1u || it == (1 << 19)) break;
        __builtin_amdgcn_s_sleep(16);
      }
      if (it == (1 << 19)) { pv = 0ull; __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      sc[0] = __uint_as_float((unsigned)pv);
    }
    __syncthreads();
    y0 = sc[0];
    __syncthreads();
  } else {
    y0 = __uint_as_float((unsigned)p.sg_in[s]);
  }
  for (uint32_t base = 0; base < p.n; base += SINK_SEG) {
    const uint32_t m = (p.n - base < SINK_SEG) ? p.n - base : SINK_SEG;   // samples of this segment
#pragma unroll
    for (uint32_t q = 0; q < SINK_C; ++q) {                     // coalesced: SINK_C independent loads per lane in flight
      const uint32_t i = t + SINK_NT * q;
      if (i < m) x[i] = row[base + i];
    }
    __syncthreads();
    // the lane's chunk [i0, i0 + cnt) in registers: both walks below then run at the chain's own latency (sub -> fma), no LDS round trip inside
    const uint32_t i0 = t * SINK_C < m ? t * SINK_C : m, cnt = (m - i0 < SINK_C) ? m - i0 : SINK_C;
    float xr[SINK_C];
#pragma unroll
    for (uint32_t q = 0; q < SINK_C; ++q) xr[q] = q < cnt ? x[i0 + q] : 0.0f;
    // 1. the chunk's own contribution to its last sample (lane 0 starts from the real state: its chain is the exact one already)
    float y = t == 0 ? y0 : 0.0f;
#pragma unroll
    for (uint32_t q = 0; q < SINK_C; ++q)
      if (q < cnt) y = __builtin_fmaf(p.alpha, xr[q] - y, y);
    // 2. s[t] = pc s[t-1] + e[t], pc = (1 - alpha)^SINK_C (only the last non-empty chunk may be short, and nothing follows it): within a wave by six shuffle steps
    // (the powers squared on the way), between the four waves through four words of LDS — one barrier where a Hillis-Steele scan over 256 lanes in LDS took sixteen
    const uint32_t wl = t & 63u, wv = t >> 6;
    float sv = y, pw = pc;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
      const float o = __shfl_up(sv, d, 64);
      const float sn = __builtin_fmaf(pw, o, sv);
      sv = wl >= d ? sn : sv;
      pw *= pw;
    }                                                           // (pw = pc^64 now: what a whole wave's chunks leave of a state)
    if (wl == 63u) sc[wv] = sv;
    float pl = 1.0f, pb = pc;                                   // pc^wl: what the chunks of this wave before the lane's leave of the wave's carry-in
#pragma unroll
    for (uint32_t bit = 0; bit < 6u; ++bit) {
      pl = ((wl >> bit) & 1u) ? pl * pb : pl;
      pb *= pb;
    }
    const float prev = __shfl_up(sv, 1u, 64);
    __syncthreads();
    float cw = 0.0f;                                            // the state at the end of the previous wave's chunks
    if (wv >= 1u) cw = sc[0];
    if (wv >= 2u) cw = __builtin_fmaf(pw, cw, sc[1]);
    if (wv >= 3u) cw = __builtin_fmaf(pw, cw, sc[2]);
    // 3. the exact form's chain from the true carry-in
    y = wl == 0u ? (wv == 0u ? y0 : cw) : __builtin_fmaf(pl, cw, prev);
    __syncthreads();                                            // (every carry-in is in a register before sc[0] takes the segment's last state below)
#pragma unroll
    for (uint32_t q = 0; q < SINK_C; ++q)
      if (q < cnt) {
        y = __builtin_fmaf(p.alpha, xr[q] - y, y);
        float v = y * p.gain;
        if (v > 32767.0f) v = 32767.0f;
        if (v < -32768.0f) v = -32768.0f;
        const unsigned w = (unsigned)(int)__builtin_rintf(v) & 0xffffu;
        xw[i0 + q] = w | (w << 16);
      }
    if (cnt > 0 && i0 + cnt == m) sc[0] = y;                     // the lane that holds the segment's last sample: the state behind it
    __syncthreads();
    y0 = sc[0];
#pragma unroll
    for (uint32_t q = 0; q < SINK_C; ++q) {
      const uint32_t i = t + SINK_NT * q;
      if (i < m) out[base + i] = xw[i];
    }
    __syncthreads();
  }
  if (t == 0) {
    const unsigned long long w = ((unsigned long long)p.gen_next << 32) | __float_as_uint(y0);
    if constexpr (LIST) __hip_atomic_store(p.sg_out + s, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else p.sg_out[s] = w;
  }
}

}  // namespace

struct sdrfm_pcm_sink {
  uint32_t n_streams;
  float alpha, gain;
  int device;
  hipStream_t own_stream, stream;
  unsigned long long* d_sg;   // [SDRFM_CHAIN_SG_SLOTS][n_streams] {tag << 32 | bits of y[n-1]}, tag t in slot t % 8 (sdrfm_sink_chain.h)
  float* d_dpow;       // [SDRFM_CHAIN_FIX] (1 - alpha)^(k + 1)
  uint32_t calls;      // calls issued so far (mod 2^32): the tag d_sg holds when every one of them is through
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
  if (k->d_sg) (void)hipFree(k->d_sg);
  if (k->d_dpow) (void)hipFree(k->d_dpow);
  if (k->d_audio) (void)hipFree(k->d_audio);
  if (k->d_pcm) (void)hipFree(k->d_pcm);
  if (k->own_stream) (void)hipStreamDestroy(k->own_stream);
  delete k;
}

// ---- the sink inside a demodulator launch (sdrfm_sink_chain.h) ---------------------------------------------------------------------------------------
int sdrfm_sink_chain_params(sdrfm_pcm_sink* k, int device, uint32_t n_streams, SdrfmSinkChain* out) {
  if (!k || !out || k->device != device || k->n_streams != n_streams) return 0;
  out->pcm = nullptr; out->pcm_stride = 0; out->runstate = nullptr;
  out->sg = k->d_sg; out->n_streams = n_streams; out->dpow = k->d_dpow; out->err = reinterpret_cast<uint32_t*>(k->d_dpow + SDRFM_CHAIN_FIX);
  out->call = k->calls;
  out->alpha = k->alpha; out->gain = k->gain;
  out->pc = (float)pow(1.0 - (double)k->alpha, (double)SDRFM_CHAIN_CH);
  for (uint32_t q = 0; q < SDRFM_CHAIN_CH; ++q) out->w[q] = (float)((double)k->alpha * pow(1.0 - (double)k->alpha, (double)(SDRFM_CHAIN_CH - 1u - q)));
  for (uint32_t q = 0; q < SDRFM_CHAIN_CH; ++q) out->dinv[q] = (float)pow(1.0 - (double)k->alpha, -(double)q);
  return k->alpha >= SDRFM_CHAIN_MIN_ALPHA ? 2 : 1;
}

void sdrfm_sink_chain_issued(sdrfm_pcm_sink* k) { ++k->calls; }

int sdrfm_sink_launch_on(sdrfm_pcm_sink* k, const float* audio, size_t audio_stride, uint32_t n, int16_t* pcm, size_t pcm_stride, hipStream_t stream) {
  if (!k) return SDRFM_EINVAL;
  if (n == 0) return SDRFM_OK;
  SinkParams p;
  p.sg_in = k->d_sg + (size_t)(k->calls % SDRFM_CHAIN_SG_SLOTS) * k->n_streams; p.sg_out = k->d_sg + (size_t)((k->calls + 1u) % SDRFM_CHAIN_SG_SLOTS) * k->n_streams;
  p.n_streams = k->n_streams; p.n = n; p.alpha = k->alpha; p.gain = k->gain;
  p.gen_next = k->calls + 1u;
  p.audio = audio; p.audio_stride = audio_stride; p.pcm = pcm; p.pcm_stride = pcm_stride;
  hipLaunchKernelGGL(k_pcm_sink_scan<false>, dim3(k->n_streams), dim3(SINK_NT), 0, stream, p, (float)pow(1.0 - (double)k->alpha, (double)SINK_C), nullptr, nullptr);
  STRY(hipGetLastError(), SDRFM_FAIL);
  ++k->calls;
  return SDRFM_OK;
}

int sdrfm_sink_launch_list_on(sdrfm_pcm_sink* k, const SdrfmSinkChain& c, const uint32_t* list_dev, uint32_t n_list, const float* audio, size_t audio_stride, uint32_t n,
                              int16_t* pcm, size_t pcm_stride, hipStream_t stream, hipEvent_t done) {
  if (!k || !list_dev) return SDRFM_EINVAL;
  if (n == 0 || n_list == 0) return SDRFM_OK;
  SinkParams p;
  p.sg_in = k->d_sg + (size_t)(c.call % SDRFM_CHAIN_SG_SLOTS) * k->n_streams; p.sg_out = k->d_sg + (size_t)((c.call + 1u) % SDRFM_CHAIN_SG_SLOTS) * k->n_streams;
  p.n_streams = k->n_streams; p.n = n; p.alpha = k->alpha; p.gain = k->gain;
  p.gen_next = c.call + 1u;
  p.audio = audio; p.audio_stride = audio_stride; p.pcm = pcm; p.pcm_stride = pcm_stride;
  hipExtLaunchKernelGGL(k_pcm_sink_scan<true>, dim3(n_list), dim3(SINK_NT), 0, stream, nullptr, done, 0, p, (float)pow(1.0 - (double)k->alpha, (double)SINK_C), list_dev, c.err);
  STRY(hipGetLastError(), SDRFM_FAIL);
  return SDRFM_OK;
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
      hipMalloc(&k->d_sg, sizeof(unsigned long long) * SDRFM_CHAIN_SG_SLOTS * n_streams) != hipSuccess ||
      hipMalloc(&k->d_dpow, sizeof(float) * (SDRFM_CHAIN_FIX + 1)) != hipSuccess) { sink_free(k); return SDRFM_ENOMEM; }   // (+ the error word)
  {
    float dp[SDRFM_CHAIN_FIX + 1];
    for (uint32_t i = 0; i < SDRFM_CHAIN_FIX; ++i) dp[i] = (float)pow(1.0 - (double)alpha, (double)(i + 1));
    dp[SDRFM_CHAIN_FIX] = 0.0f;
    if (hipMemcpy(k->d_dpow, dp, sizeof(dp), hipMemcpyHostToDevice) != hipSuccess) { sink_free(k); return SDRFM_FAIL; }
  }
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
  STRY(hipMemsetAsync(k->d_sg, 0, sizeof(unsigned long long) * SDRFM_CHAIN_SG_SLOTS * k->n_streams, k->stream), SDRFM_FAIL);
  STRY(hipMemsetAsync(k->d_dpow + SDRFM_CHAIN_FIX, 0, sizeof(uint32_t), k->stream), SDRFM_FAIL);   // (the chain's error word)
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  k->calls = 0;
  return SDRFM_OK;
}

int sdrfm_pcm_sink_set_stream(sdrfm_pcm_sink_t* k, void* hip_stream) {
  if (!k) return SDRFM_EINVAL;
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  k->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : k->own_stream;
  return SDRFM_OK;
}

// a run of a demodulator launch gave up waiting for its predecessor's word (sdrfm_sink_chain.h): the PCM since then is not to be trusted
static int sink_chain_error(sdrfm_pcm_sink* k) {
  uint32_t e = 0;
  if (hipMemcpy(&e, k->d_dpow + SDRFM_CHAIN_FIX, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return SDRFM_FAIL;
  if (e) fprintf(stderr, "[sdrfm] PCM sink: a run of a demodulator launch waited in vain for its predecessor's state (csrc/sdrfm_sink_chain.h)\n");
  return e ? SDRFM_FAIL : SDRFM_OK;
}

int sdrfm_pcm_sink_synchronize(sdrfm_pcm_sink_t* k) {
  if (!k) return SDRFM_EINVAL;
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  STRY(hipStreamSynchronize(k->stream), SDRFM_FAIL);
  return sink_chain_error(k);
}

int sdrfm_pcm_sink_process_batch(sdrfm_pcm_sink_t* k, const float* audio, size_t audio_stride, uint32_t n, int16_t* pcm,
                                 size_t pcm_stride, uint32_t flags) {
  if (!k) return SDRFM_EINVAL;
  if (flags & ~(SDRFM_F_DEVICE_PTRS | SDRFM_PCM_F_EXACT)) return SDRFM_EINVAL;
  if (n == 0) return SDRFM_OK;
  if (!audio || !pcm) return SDRFM_EINVAL;
  if (k->n_streams > 1 && (audio_stride < n || pcm_stride < 2 * (size_t)n)) return SDRFM_ECAPACITY;
  if (pcm_stride & 1u) return SDRFM_EINVAL;                          // rows are written as (L,R) dwords
  STRY(hipSetDevice(k->device), SDRFM_FAIL);
  SinkParams p;
  p.sg_in = k->d_sg + (size_t)(k->calls % SDRFM_CHAIN_SG_SLOTS) * k->n_streams; p.sg_out = k->d_sg + (size_t)((k->calls + 1u) % SDRFM_CHAIN_SG_SLOTS) * k->n_streams;
  p.n_streams = k->n_streams; p.n = n; p.alpha = k->alpha; p.gain = k->gain;
  p.gen_next = k->calls + 1u;
  const dim3 grid((k->n_streams + 63) / 64);
  const float pc = (float)pow(1.0 - (double)k->alpha, (double)SINK_C);   // the blocked scan's carry factor: (1 - alpha)^(samples per chunk)
  const bool exact = (flags & SDRFM_PCM_F_EXACT) != 0;
  auto launch = [&]() {
    if (exact) hipLaunchKernelGGL(k_pcm_sink, grid, dim3(64), 0, k->stream, p);
    else hipLaunchKernelGGL(k_pcm_sink_scan<false>, dim3(k->n_streams), dim3(SINK_NT), 0, k->stream, p, pc, nullptr, nullptr);
  };
  if (flags & SDRFM_F_DEVICE_PTRS) {
    if ((uintptr_t)pcm % 4 != 0) return SDRFM_EINVAL;
    p.audio = audio; p.audio_stride = audio_stride; p.pcm = pcm; p.pcm_stride = pcm_stride;
    launch();
    STRY(hipGetLastError(), SDRFM_FAIL);
    ++k->calls;
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
  launch();
  STRY(hipGetLastError(), SDRFM_FAIL);
  ++k->calls;
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
  unsigned long long* tmp = new (std::nothrow) unsigned long long[k->n_streams];
  if (!tmp) return SDRFM_ENOMEM;
  const hipError_t e = hipMemcpy(tmp, k->d_sg + (size_t)(k->calls % SDRFM_CHAIN_SG_SLOTS) * k->n_streams, sizeof(unsigned long long) * k->n_streams, hipMemcpyDeviceToHost);
  for (uint32_t i = 0; e == hipSuccess && i < k->n_streams; ++i) { const unsigned b = (unsigned)tmp[i]; memcpy(state_out + i, &b, sizeof(float)); }
  delete[] tmp;
  return e == hipSuccess ? sink_chain_error(k) : SDRFM_FAIL;
}

}  // extern "C"
