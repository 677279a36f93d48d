/*
 * sdrfm_wbfm.hip — multi-channel WBFM path (BASELINE configs[4]) behind the sdrfm_wbfm_* entry points of include/sdrfm.h:
 * 16-band critically-sampled polyphase channelizer -> per-band FM discriminator -> rational L/M audio resampler.
 *
 * Same boundary as the narrow-band path: it consumes the RTL2832 bulk-IN buffer (RTLSDR_CommItfTypedef,
 * Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Inc/usbh_rtlsdr.h:165-173) from the hook the reference leaves empty
 * (usbh_rtlsdr.c:1094-1097).  Arithmetic: DESIGN.md "WBFM spec" — fp32 fmaf chains oldest-first, a fixed radix-2 DIT
 * 16-point DFT graph, the K3 discriminator of sdrfm_math.h.  Correctness-first kernels (not yet tuned):
 *   k_wbfm_chan : one block = NT channelizer steps of one stream: stage x (f32) in LDS, 16 polyphase branches per step,
 *                 16-point DFT per step, discriminator per band -> d scratch in HBM [stream][band][t]
 *   k_wbfm_res  : one thread = one audio sample of one band; extra blocks hand the d history over
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/sdrfm.h"
#include "sdrfm_math.h"

namespace {

constexpr int NB = SDRFM_WBFM_BANDS;

struct WParams {
  const uint8_t* iq;
  size_t iq_stride;
  float* audio;
  size_t band_stride;           // floats between bands; stream stride = NB * band_stride
  const float2* hist_x_in;      // [ns][P-1]
  float2* hist_x_out;
  const float2* cprev_in;       // [ns][NB]
  float2* cprev_out;
  const float* hist_d_in;       // [ns][NB][HD]
  float* hist_d_out;
  float* dbuf;                  // [ns][NB][dcap]   new discriminator outputs of this call
  const float* p;               // P prototype taps
  const float* g;               // Tg resampler taps
  uint32_t P, Tg, L, M, HD, dcap;
  uint32_t N;                   // new IQ samples per stream
  uint32_t Tn;                  // channelizer steps this call
  uint32_t A;                   // audio samples per band this call
  uint32_t phase_x;
  unsigned long long n_d, n_a;  // d / audio samples produced before this call
  int res_q0;                   // floor(n_a*M/L) - n_d   (call-relative index of the newest d of audio sample n_a)
  uint32_t res_r0;              // (n_a*M) mod L
  uint32_t NT, tiles_per_stream, n_streams;
};

__device__ __forceinline__ float2 wload_x(const WParams& w, uint32_t stream, int s) {
  if (s < 0) return (s >= -(int)(w.P - 1)) ? w.hist_x_in[(size_t)stream * (w.P - 1) + (w.P - 1 + s)] : make_float2(0.f, 0.f);
  const uchar2 v = *reinterpret_cast<const uchar2*>(w.iq + (size_t)stream * w.iq_stride + 2 * (size_t)s);
  return make_float2((float)v.x - 127.5f, (float)v.y - 127.5f);
}

// exp(+j*2*pi*t/16), t = 0..7 (same float table as the oracle)
__constant__ float W16_RE[8] = {1.0f, 0x1.d906bcp-1f, 0x1.6a09e6p-1f, 0x1.87de2ap-2f, 0.0f, -0x1.87de2ap-2f, -0x1.6a09e6p-1f, -0x1.d906bcp-1f};
__constant__ float W16_IM[8] = {0.0f, 0x1.87de2ap-2f, 0x1.6a09e6p-1f, 0x1.d906bcp-1f, 1.0f, 0x1.d906bcp-1f, 0x1.6a09e6p-1f, 0x1.87de2ap-2f};

// radix-2 DIT, bit-reversed input, stages m = 2,4,8,16; butterfly exactly as DESIGN.md / the oracle state it
__device__ __forceinline__ void fft16_dit(float (&ar)[16], float (&ai)[16]) {
  constexpr int rev[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  float xr[16], xi[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { xr[k] = ar[rev[k]]; xi[k] = ai[rev[k]]; }
#pragma unroll
  for (int m = 2; m <= 16; m <<= 1) {
    const int half = m / 2, step = 16 / m;
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += m)
#pragma unroll
      for (int t = 0; t < half; ++t) {
        const float wr = W16_RE[t * step], wi = W16_IM[t * step];
        const float br = xr[g0 + t + half], bi = xi[g0 + t + half];
        const float tr = __builtin_fmaf(wr, br, -(wi * bi));
        const float ti = __builtin_fmaf(wr, bi, wi * br);
        const float cr = xr[g0 + t], ci = xi[g0 + t];
        xr[g0 + t] = cr + tr; xi[g0 + t] = ci + ti;
        xr[g0 + t + half] = cr - tr; xi[g0 + t + half] = ci - ti;
      }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) { ar[k] = xr[k]; ai[k] = xi[k]; }
}

// LDS: xs[NX] f32x2 | us[(NT+1)*NB] f32x2 (branch outputs, then band outputs in place) | ps[P]
__global__ void __launch_bounds__(256) k_wbfm_chan(WParams w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t P = w.P, Q = P / NB, NT = w.NT;
  const uint32_t NX = (NT + 1) * NB + P;
  float2* xs = reinterpret_cast<float2*>(smem);
  float2* us = xs + NX;
  float* ps = reinterpret_cast<float*>(us + (NT + 1) * NB);
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  for (uint32_t k = tid; k < P; k += nthr) ps[k] = w.p[k];

  const uint32_t n_tile_blocks = w.n_streams * w.tiles_per_stream;
  if (blockIdx.x >= n_tile_blocks) {   // state: new input history (and c_prev when no step happened)
    const uint32_t stream = blockIdx.x - n_tile_blocks;
    for (uint32_t k = tid; k + 1 < P; k += nthr)
      w.hist_x_out[(size_t)stream * (P - 1) + k] = wload_x(w, stream, (int)w.N - (int)(P - 1) + (int)k);
    if (w.Tn == 0)
      for (uint32_t b = tid; b < (uint32_t)NB; b += nthr) w.cprev_out[(size_t)stream * NB + b] = w.cprev_in[(size_t)stream * NB + b];
    return;
  }
  const uint32_t stream = blockIdx.x / w.tiles_per_stream;
  const uint32_t tile = blockIdx.x % w.tiles_per_stream;
  const int t0 = (int)(tile * NT);
  int t1 = t0 + (int)NT;                       // exclusive
  if (t1 > (int)w.Tn) t1 = (int)w.Tn;
  if (t0 >= t1) return;
  const int ta = t0 > 0 ? t0 - 1 : 0;          // first step computed here (t0-1 is recomputed as the discriminator's y[m-1])
  const int e_ta = (ta + 1) * NB - 1 - (int)w.phase_x;
  const int xlo = e_ta - (NB - 1) - (int)NB * (int)(Q - 1);   // oldest input of step ta
  const int xhi = t1 * NB - 1 - (int)w.phase_x;               // newest input of step t1-1
  for (int s = xlo + (int)tid; s <= xhi; s += (int)nthr) xs[s - xlo] = wload_x(w, stream, s);
  __syncthreads();
  // polyphase branches: u_r[t], oldest sample first
  const int nsteps = t1 - ta;
  for (int idx = (int)tid; idx < nsteps * NB; idx += (int)nthr) {
    const int t = ta + idx / NB, r = idx % NB;
    const int e = (t + 1) * NB - 1 - (int)w.phase_x;
    float ar = 0.0f, ai = 0.0f;
    for (int q = (int)Q - 1; q >= 0; --q) {
      const float2 x = xs[e - r - NB * q - xlo];
      const float c = ps[r + NB * q];
      ar = __builtin_fmaf(c, x.x, ar);
      ai = __builtin_fmaf(c, x.y, ai);
    }
    us[idx] = make_float2(ar, ai);
  }
  __syncthreads();
  // 16-point DFT per step (one thread per step), in place
  for (int st = (int)tid; st < nsteps; st += (int)nthr) {
    float ar[16], ai[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float2 u = us[st * NB + r]; ar[r] = u.x; ai[r] = u.y; }
    fft16_dit(ar, ai);
#pragma unroll
    for (int b = 0; b < 16; ++b) us[st * NB + b] = make_float2(ar[b], ai[b]);
  }
  __syncthreads();
  // discriminator per (step, band); c[-1] comes from the old state
  for (int idx = (int)tid; idx < (t1 - t0) * NB; idx += (int)nthr) {
    const int b = idx / (t1 - t0), t = t0 + idx % (t1 - t0);   // t fastest: coalesced stores into dbuf[stream][band][t]
    const float2 c = us[(t - ta) * NB + b];
    const float2 pv = (t == 0) ? w.cprev_in[(size_t)stream * NB + b] : us[(t - 1 - ta) * NB + b];
    w.dbuf[((size_t)stream * NB + b) * w.dcap + t] = sdrfm_discriminate(c.x, c.y, pv.x, pv.y);
    if (t == (int)w.Tn - 1) w.cprev_out[(size_t)stream * NB + b] = c;
  }
}

// one thread per (band, audio sample); blocks beyond the audio grid hand the d history over
__global__ void __launch_bounds__(256) k_wbfm_res(WParams w) {
  const uint32_t per_stream = (w.A * NB + 255) / 256;          // audio blocks per stream
  const uint32_t n_audio_blocks = w.n_streams * per_stream;
  const int HD = (int)w.HD;
  if (blockIdx.x >= n_audio_blocks) {
    const uint32_t stream = blockIdx.x - n_audio_blocks;
    for (int idx = (int)threadIdx.x; idx < NB * HD; idx += (int)blockDim.x) {
      const int b = idx / HD, k = idx % HD;
      const int li = (int)w.Tn - HD + k;                        // call-relative index of the k-th newest-window entry
      const size_t sb = (size_t)stream * NB + b;
      w.hist_d_out[sb * HD + k] = (li < 0) ? w.hist_d_in[sb * HD + (HD + li)] : w.dbuf[sb * w.dcap + li];
    }
    return;
  }
  const uint32_t stream = blockIdx.x / per_stream;
  const uint32_t idx = (blockIdx.x % per_stream) * 256 + threadIdx.x;
  if (idx >= w.A * NB) return;
  const uint32_t b = idx / w.A, jl = idx % w.A;
  // (n_a + jl)*M = n_a*M + jl*M: quotient and remainder by L from the host's 64-bit part and a 32-bit local part
  const uint32_t loc = w.res_r0 + jl * w.M;
  const int nj = w.res_q0 + (int)(loc / w.L);                // call-relative index of the newest d
  const uint32_t phi = loc % w.L;
  const int imax = (int)((w.Tg - 1 - phi) / w.L);
  const size_t sb = (size_t)stream * NB + b;
  float acc = 0.0f;
  for (int i = imax; i >= 0; --i) {
    const int li = nj - i;
    const float d = (li < 0) ? w.hist_d_in[sb * HD + (HD + li)] : w.dbuf[sb * w.dcap + li];
    acc = __builtin_fmaf(w.g[phi + w.L * (uint32_t)i], d, acc);
  }
  w.audio[(size_t)stream * NB * w.band_stride + (size_t)b * w.band_stride + jl] = acc;
}

}  // namespace

struct sdrfm_wbfm {
  sdrfm_wbfm_config cfg;
  int device;
  hipStream_t own_stream, stream;
  float *d_p, *d_g;
  float2* d_hist_x[2];
  float2* d_cprev[2];
  float* d_hist_d[2];
  float* d_dbuf;
  uint32_t HD, dcap;
  int cur;
  uint32_t phase_x;
  unsigned long long n_d, n_a;
  uint8_t* d_iq; size_t d_iq_stride;
  float* d_audio; size_t d_band_stride;
  uint32_t max_bytes, NT;
  size_t lds_bytes;
};

#define WTRY(expr, code)                                                                                     \
  do {                                                                                                       \
    hipError_t e__ = (expr);                                                                                 \
    if (e__ != hipSuccess) {                                                                                 \
      fprintf(stderr, "[sdrfm_wbfm] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return (code);                                                                                         \
    }                                                                                                        \
  } while (0)

static void wfree(sdrfm_wbfm* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  void* ptrs[] = {h->d_p, h->d_g, h->d_hist_x[0], h->d_hist_x[1], h->d_cprev[0], h->d_cprev[1], h->d_hist_d[0], h->d_hist_d[1],
                  h->d_dbuf, h->d_iq, h->d_audio};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  free(const_cast<float*>(h->cfg.proto_coeffs));
  free(const_cast<float*>(h->cfg.resamp_coeffs));
  delete h;
}

static void wcounts(const sdrfm_wbfm* h, uint32_t nbytes, uint32_t* Tn, uint32_t* A) {
  const uint64_t N = nbytes / 2;
  const uint64_t t = (h->phase_x + N) / NB;
  const unsigned long long nd = h->n_d + t;
  // audio outputs j exist while floor(j*M/L) < nd  <=>  j < ceil(nd*L/M)
  const unsigned long long jend = (nd * h->cfg.resamp_up + h->cfg.resamp_down - 1) / h->cfg.resamp_down;
  *Tn = (uint32_t)t;
  *A = (uint32_t)(jend > h->n_a ? jend - h->n_a : 0);
}

extern "C" {

int sdrfm_wbfm_create(const sdrfm_wbfm_config* cfg, sdrfm_wbfm_t** out) {
  if (!out) return SDRFM_EINVAL;
  *out = nullptr;
  if (!cfg || cfg->struct_size != sizeof(sdrfm_wbfm_config) || cfg->flags) return SDRFM_EINVAL;
  if (!cfg->n_streams || !cfg->proto_coeffs || !cfg->resamp_coeffs) return SDRFM_EINVAL;
  if (!cfg->proto_taps || cfg->proto_taps % NB || cfg->proto_taps > 512) return SDRFM_EINVAL;
  if (!cfg->resamp_taps || cfg->resamp_taps > 512 || !cfg->resamp_up || !cfg->resamp_down || cfg->resamp_up > 64 ||
      cfg->resamp_down > 256) return SDRFM_EINVAL;
  for (uint32_t k = 0; k < cfg->proto_taps; ++k) if (!std::isfinite(cfg->proto_coeffs[k])) return SDRFM_EINVAL;
  for (uint32_t k = 0; k < cfg->resamp_taps; ++k) if (!std::isfinite(cfg->resamp_coeffs[k])) return SDRFM_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return SDRFM_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SDRFM_NO_DEVICE;
  WTRY(hipSetDevice(cfg->device), SDRFM_NO_DEVICE);
  sdrfm_wbfm* h = new (std::nothrow) sdrfm_wbfm();
  if (!h) return SDRFM_ENOMEM;
  memset(static_cast<void*>(h), 0, sizeof(*h));
  h->cfg = *cfg;
  h->device = cfg->device;
  h->max_bytes = (cfg->max_bytes_per_call ? cfg->max_bytes_per_call : (1u << 20)) & ~1u;
  float* pc = (float*)malloc(4 * cfg->proto_taps);
  float* gc = (float*)malloc(4 * cfg->resamp_taps);
  h->cfg.proto_coeffs = pc; h->cfg.resamp_coeffs = gc;
  if (!pc || !gc) { wfree(h); return SDRFM_ENOMEM; }
  memcpy(pc, cfg->proto_coeffs, 4 * cfg->proto_taps);
  memcpy(gc, cfg->resamp_coeffs, 4 * cfg->resamp_taps);
  const size_t ns = cfg->n_streams, P = cfg->proto_taps;
  h->HD = (cfg->resamp_taps + cfg->resamp_up - 1) / cfg->resamp_up;
  h->dcap = h->max_bytes / 2 / NB + 2;
#define CR(expr) do { if ((expr) != hipSuccess) { wfree(h); return SDRFM_ENOMEM; } } while (0)
  CR(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  CR(hipMalloc(&h->d_p, 4 * P));
  CR(hipMalloc(&h->d_g, 4 * cfg->resamp_taps));
  for (int i = 0; i < 2; ++i) {
    CR(hipMalloc(&h->d_hist_x[i], sizeof(float2) * ns * (P - 1)));
    CR(hipMalloc(&h->d_cprev[i], sizeof(float2) * ns * NB));
    CR(hipMalloc(&h->d_hist_d[i], sizeof(float) * ns * NB * h->HD));
  }
  CR(hipMalloc(&h->d_dbuf, sizeof(float) * ns * NB * h->dcap));
  CR(hipMemcpy(h->d_p, pc, 4 * P, hipMemcpyHostToDevice));
  CR(hipMemcpy(h->d_g, gc, 4 * cfg->resamp_taps, hipMemcpyHostToDevice));
#undef CR
  h->NT = 64;
  h->lds_bytes = ((size_t)(h->NT + 1) * NB + P) * 8 + (size_t)(h->NT + 1) * NB * 8 + P * 4;
  if (h->lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_wbfm_chan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) {
    wfree(h);
    return SDRFM_NOT_SUPPORTED;
  }
  const int rc = sdrfm_wbfm_reset(h);
  if (rc != SDRFM_OK) { wfree(h); return rc; }
  *out = h;
  return SDRFM_OK;
}

void sdrfm_wbfm_destroy(sdrfm_wbfm_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  wfree(h);
}

int sdrfm_wbfm_reset(sdrfm_wbfm_t* h) {
  if (!h) return SDRFM_EINVAL;
  WTRY(hipSetDevice(h->device), SDRFM_FAIL);
  const size_t ns = h->cfg.n_streams, P = h->cfg.proto_taps;
  for (int i = 0; i < 2; ++i) {
    WTRY(hipMemsetAsync(h->d_hist_x[i], 0, sizeof(float2) * ns * (P - 1), h->stream), SDRFM_FAIL);
    WTRY(hipMemsetAsync(h->d_cprev[i], 0, sizeof(float2) * ns * NB, h->stream), SDRFM_FAIL);
    WTRY(hipMemsetAsync(h->d_hist_d[i], 0, sizeof(float) * ns * NB * h->HD, h->stream), SDRFM_FAIL);
  }
  WTRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  h->cur = 0; h->phase_x = 0; h->n_d = 0; h->n_a = 0;
  return SDRFM_OK;
}

int sdrfm_wbfm_audio_count(const sdrfm_wbfm_t* h, uint32_t nbytes, uint32_t* n_audio) {
  if (!h || !n_audio) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  uint32_t Tn;
  wcounts(h, nbytes, &Tn, n_audio);
  return SDRFM_OK;
}

int sdrfm_wbfm_set_stream(sdrfm_wbfm_t* h, void* hip_stream) {
  if (!h) return SDRFM_EINVAL;
  WTRY(hipSetDevice(h->device), SDRFM_FAIL);
  WTRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return SDRFM_OK;
}

int sdrfm_wbfm_synchronize(sdrfm_wbfm_t* h) {
  if (!h) return SDRFM_EINVAL;
  WTRY(hipSetDevice(h->device), SDRFM_FAIL);
  WTRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

static int wenqueue(sdrfm_wbfm* h, const uint8_t* d_iq, size_t iq_stride, uint32_t nbytes, float* d_audio, size_t band_stride) {
  const sdrfm_wbfm_config& c = h->cfg;
  uint32_t Tn, A;
  wcounts(h, nbytes, &Tn, &A);
  const uint32_t N = nbytes / 2;
  if (N == 0) return SDRFM_OK;
  WParams w;
  w.iq = d_iq; w.iq_stride = iq_stride; w.audio = d_audio; w.band_stride = band_stride;
  w.hist_x_in = h->d_hist_x[h->cur]; w.hist_x_out = h->d_hist_x[h->cur ^ 1];
  w.cprev_in = h->d_cprev[h->cur]; w.cprev_out = h->d_cprev[h->cur ^ 1];
  w.hist_d_in = h->d_hist_d[h->cur]; w.hist_d_out = h->d_hist_d[h->cur ^ 1];
  w.dbuf = h->d_dbuf; w.p = h->d_p; w.g = h->d_g;
  w.P = c.proto_taps; w.Tg = c.resamp_taps; w.L = c.resamp_up; w.M = c.resamp_down; w.HD = h->HD; w.dcap = h->dcap;
  w.N = N; w.Tn = Tn; w.A = A; w.phase_x = h->phase_x; w.n_d = h->n_d; w.n_a = h->n_a;
  w.res_q0 = (int)((long long)((h->n_a * c.resamp_down) / c.resamp_up) - (long long)h->n_d);
  w.res_r0 = (uint32_t)((h->n_a * c.resamp_down) % c.resamp_up);
  w.NT = h->NT; w.tiles_per_stream = (Tn + h->NT - 1) / h->NT; w.n_streams = c.n_streams;
  hipLaunchKernelGGL(k_wbfm_chan, dim3(c.n_streams * w.tiles_per_stream + c.n_streams), dim3(256), h->lds_bytes, h->stream, w);
  WTRY(hipGetLastError(), SDRFM_FAIL);
  const uint32_t per_stream = (A * NB + 255) / 256;
  hipLaunchKernelGGL(k_wbfm_res, dim3(c.n_streams * per_stream + c.n_streams), dim3(256), 0, h->stream, w);
  WTRY(hipGetLastError(), SDRFM_FAIL);
  h->cur ^= 1;
  h->phase_x = (uint32_t)((h->phase_x + (uint64_t)N) % NB);
  h->n_d += Tn; h->n_a += A;
  return SDRFM_OK;
}

int sdrfm_wbfm_process_batch(sdrfm_wbfm_t* h, const uint8_t* iq, size_t iq_stride, uint32_t nbytes, float* audio,
                             size_t band_stride, uint32_t* n_audio, uint32_t flags) {
  if (!h || !n_audio) return SDRFM_EINVAL;
  if (flags & ~SDRFM_F_DEVICE_PTRS) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  if (nbytes == 0) { *n_audio = 0; return SDRFM_OK; }
  if (!iq) return SDRFM_EINVAL;
  if (nbytes > h->max_bytes) return SDRFM_ECAPACITY;
  const uint32_t ns = h->cfg.n_streams;
  if (ns > 1 && iq_stride < nbytes) return SDRFM_ECAPACITY;
  uint32_t Tn, A;
  wcounts(h, nbytes, &Tn, &A);
  *n_audio = A;
  if (A && !audio) return SDRFM_EINVAL;
  if (band_stride < A) return SDRFM_ECAPACITY;
  WTRY(hipSetDevice(h->device), SDRFM_FAIL);
  if (flags & SDRFM_F_DEVICE_PTRS) return wenqueue(h, iq, iq_stride, nbytes, audio, band_stride);
  if (!h->d_iq) {
    h->d_iq_stride = ((size_t)h->max_bytes + 255) & ~(size_t)255;
    h->d_band_stride = (((size_t)h->dcap * h->cfg.resamp_up) / h->cfg.resamp_down + 64) & ~(size_t)63;
    WTRY(hipMalloc(&h->d_iq, ns * h->d_iq_stride), SDRFM_ENOMEM);
    WTRY(hipMalloc(&h->d_audio, sizeof(float) * ns * NB * h->d_band_stride), SDRFM_ENOMEM);
  }
  WTRY(hipMemcpy2DAsync(h->d_iq, h->d_iq_stride, iq, ns > 1 ? iq_stride : nbytes, nbytes, ns, hipMemcpyHostToDevice, h->stream), SDRFM_FAIL);
  const int rc = wenqueue(h, h->d_iq, h->d_iq_stride, nbytes, h->d_audio, h->d_band_stride);
  if (rc != SDRFM_OK) return rc;
  if (A)
    WTRY(hipMemcpy2DAsync(audio, band_stride * sizeof(float), h->d_audio, h->d_band_stride * sizeof(float), A * sizeof(float),
                          (size_t)ns * NB, hipMemcpyDeviceToHost, h->stream), SDRFM_FAIL);
  WTRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

}  // extern "C"
