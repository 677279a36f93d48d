/*
 * sdrfm.hip — MI355X (gfx950) implementation of the IQ -> FM-audio path behind the C-ABI of include/sdrfm.h.
 *
 * What it replaces in the reference: nothing that exists — it FILLS the empty consumer hook of the RTL2832 bulk-IN FSM
 * (RTLSDR_XFER_COMPLETE, Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Src/usbh_rtlsdr.c:1094-1097; commented
 * poll in src/main.c:76-79).  Buffer contract in: RTLSDR_CommItfTypedef {buff, buffSize} (usbh_rtlsdr.h:165-173).
 *
 * Host side: plain C-style code (handles, status codes, no exceptions, no torch types).
 * Device side: hand-written HIP kernels for gfx950 only; compiled with -ffp-contract=off so that every FMA is explicit
 * and the FIR chains round exactly like the oracle's.
 *
 * Data layout in HBM (per handle):
 *   iq       [n_streams][iq_stride]      u8   interleaved I,Q  (caller's device buffer, or the handle's staging copy)
 *   audio    [n_streams][audio_stride]   f32
 *   hist_x   2 x [n_streams][T-1]        f32x2  last T-1 DC-shifted inputs, oldest first   (ping-pong per call)
 *   y_prev   2 x [n_streams]             f32x2  last decimated sample
 *   hist_d   2 x [n_streams][Ta-1]       f32    last Ta-1 discriminator outputs
 *   taps     h[T], g[Ta]                 f32
 * Decimator phases are identical for all streams of a handle (every stream advances by the same nbytes) and live on the
 * host.
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

struct CallParams {
  const uint8_t* iq;
  size_t iq_stride;     // bytes
  float* audio;
  size_t audio_stride;  // floats
  const float2* hist_x_in;
  float2* hist_x_out;
  const float2* yprev_in;
  float2* yprev_out;
  const float* hist_d_in;
  float* hist_d_out;
  const float* h;  // T taps
  const float* g;  // Ta taps
  uint32_t T, D, Ta, Da;
  uint32_t N;   // new IQ samples per stream in this call
  uint32_t M;   // decimated outputs y[0..M) produced by this call
  uint32_t A;   // audio outputs a[0..A) produced by this call
  int32_t e0;   // chunk index of the newest input of y[0]:  D-1-phase_x
  int32_t f0;   // call-relative index of the newest d of a[0]: Da-1-phase_d
  uint32_t NA;  // audio outputs per tile
  uint32_t tiles_per_stream;
  uint32_t n_streams;
};

// ---- virtual input: chunk index s in [-(T-1), N) ---------------------------------------------------------------
__device__ __forceinline__ float2 load_x(const CallParams& p, uint32_t stream, int s) {
  if (s < 0) return p.hist_x_in[(size_t)stream * (p.T - 1) + (p.T - 1 + s)];
  const uint8_t* b = p.iq + (size_t)stream * p.iq_stride + 2 * (size_t)s;
  const uchar2 v = *reinterpret_cast<const uchar2*>(b);
  return make_float2((float)v.x - 127.5f, (float)v.y - 127.5f);
}

// =================================================================================================================
//  Generic kernel: any (T, D, Ta, Da).  One block = one tile of NA audio outputs of one stream, or (for the last
//  n_streams blocks) the state hand-over of one stream.  Everything it needs before the tile is recomputed from the
//  input (halo), so blocks are independent.
//  LDS: xs[NX] f32x2 | ys[NY] f32x2 | ds[ND] f32 | hs[T] | gs[Ta]
// =================================================================================================================
__global__ void __launch_bounds__(256) k_generic(CallParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t T = p.T, D = p.D, Ta = p.Ta, Da = p.Da;
  const uint32_t ND_MAX = (p.NA - 1) * Da + Ta;
  const uint32_t NY_MAX = ND_MAX + 1;
  const uint32_t NX_MAX = (NY_MAX - 1) * D + T;
  float2* xs = reinterpret_cast<float2*>(smem);
  float2* ys = xs + NX_MAX;
  float* ds = reinterpret_cast<float*>(ys + NY_MAX);
  float* hs = ds + ND_MAX;
  float* gs = hs + T;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;

  for (uint32_t k = tid; k < T; k += nthr) hs[k] = p.h[k];
  for (uint32_t k = tid; k < Ta; k += nthr) gs[k] = p.g[k];

  const uint32_t n_tile_blocks = p.n_streams * p.tiles_per_stream;
  if (blockIdx.x < n_tile_blocks) {
    // ------------------------------------------------------------------ audio tile
    const uint32_t stream = blockIdx.x / p.tiles_per_stream;
    const uint32_t tile = blockIdx.x % p.tiles_per_stream;
    const int j0 = (int)(tile * p.NA);
    int j1 = j0 + (int)p.NA;
    if (j1 > (int)p.A) j1 = (int)p.A;
    if (j0 >= j1) return;
    const int dlo = p.f0 + j0 * (int)Da - (int)(Ta - 1);  // oldest d needed (may be < 0: history)
    const int dhi = p.f0 + (j1 - 1) * (int)Da;            // newest d needed
    const int dc0 = dlo > 0 ? dlo : 0;                    // first d that must be computed here
    const int yc0 = dc0 > 0 ? dc0 - 1 : 0;                // first y computed here (y[-1] comes from state)
    const int xlo = p.e0 + yc0 * (int)D - (int)(T - 1);   // oldest input needed
    const int xhi = p.e0 + dhi * (int)D;                  // newest input needed
    // stage DC-shifted inputs
    for (int s = xlo + (int)tid; s <= xhi; s += (int)nthr) xs[s - xlo] = load_x(p, stream, s);
    __syncthreads();
    // K2: y[i], i in [yc0, dhi]
    for (int i = yc0 + (int)tid; i <= dhi; i += (int)nthr) {
      const float2* w = xs + (p.e0 + i * (int)D - (int)(T - 1) - xlo);
      float ar = 0.0f, ai = 0.0f;
      for (uint32_t j = 0; j < T; ++j) {
        const float c = hs[T - 1 - j];
        const float2 x = w[j];
        ar = __builtin_fmaf(c, x.x, ar);
        ai = __builtin_fmaf(c, x.y, ai);
      }
      ys[i - yc0] = make_float2(ar, ai);
    }
    __syncthreads();
    // K3: d[i], i in [dlo, dhi]
    for (int i = dlo + (int)tid; i <= dhi; i += (int)nthr) {
      float d;
      if (i < 0) {
        d = p.hist_d_in[(size_t)stream * (Ta - 1) + (Ta - 1 + i)];
      } else {
        const float2 y = ys[i - yc0];
        const float2 pr = (i == 0) ? p.yprev_in[stream] : ys[i - 1 - yc0];
        d = sdrfm_discriminate(y.x, y.y, pr.x, pr.y);
      }
      ds[i - dlo] = d;
    }
    __syncthreads();
    // K4: a[j]
    for (int j = j0 + (int)tid; j < j1; j += (int)nthr) {
      const float* w = ds + (size_t)(j - j0) * Da;  // oldest d of output j
      float acc = 0.0f;
      for (uint32_t k = 0; k < Ta; ++k) acc = __builtin_fmaf(gs[Ta - 1 - k], w[k], acc);
      p.audio[(size_t)stream * p.audio_stride + j] = acc;
    }
  } else {
    // ------------------------------------------------------------------ state hand-over of one stream
    const uint32_t stream = blockIdx.x - n_tile_blocks;
    const int N = (int)p.N, M = (int)p.M;
    // new input history = last T-1 samples of [old history | chunk]
    for (uint32_t k = tid; k + 1 < T; k += nthr)
      p.hist_x_out[(size_t)stream * (T - 1) + k] = load_x(p, stream, N - (int)(T - 1) + (int)k);
    // y[M-Ta .. M-1] (those that exist) -> ys[0..Ta)
    __syncthreads();
    const int ylo = M - (int)Ta;
    for (int q = (int)tid; q < (int)Ta; q += (int)nthr) {
      const int i = ylo + q;
      float2 y = make_float2(0.f, 0.f);
      if (i >= 0) {
        const int s0 = p.e0 + i * (int)D - (int)(T - 1);
        float ar = 0.0f, ai = 0.0f;
        for (uint32_t j = 0; j < T; ++j) {
          const float c = hs[T - 1 - j];
          const float2 x = load_x(p, stream, s0 + (int)j);
          ar = __builtin_fmaf(c, x.x, ar);
          ai = __builtin_fmaf(c, x.y, ai);
        }
        y = make_float2(ar, ai);
      } else if (i == -1) {
        y = p.yprev_in[stream];
      }
      ys[q] = y;
    }
    __syncthreads();
    if (tid == 0) p.yprev_out[stream] = (M > 0) ? ys[Ta - 1] : p.yprev_in[stream];
    // new d history = d[M-(Ta-1) .. M-1]
    for (int q = (int)tid; q + 1 < (int)Ta; q += (int)nthr) {
      const int i = M - (int)(Ta - 1) + q;
      float d;
      if (i < 0) {
        d = p.hist_d_in[(size_t)stream * (Ta - 1) + (Ta - 1 + i)];
      } else {
        const float2 y = ys[i - ylo];
        const float2 pr = ys[i - 1 - ylo];  // i-1 >= -1 == ylo + (Ta-1-M) ... always inside ys (i-1-ylo >= 0)
        d = sdrfm_discriminate(y.x, y.y, pr.x, pr.y);
      }
      p.hist_d_out[(size_t)stream * (Ta - 1) + q] = d;
    }
  }
}

}  // namespace

// =================================================================================================================
//  Host side (C-ABI)
// =================================================================================================================
struct sdrfm {
  sdrfm_config cfg;
  int device;
  hipStream_t own_stream;
  hipStream_t stream;  // the one in use (own_stream or caller's)
  float* d_h;
  float* d_g;
  float2* d_hist_x[2];
  float2* d_yprev[2];
  float* d_hist_d[2];
  int cur;  // index of the state set holding the current state
  uint32_t phase_x, phase_d;
  // staging for host-pointer calls
  uint8_t* d_iq;
  size_t d_iq_stride;
  float* d_audio;
  size_t d_audio_stride;
  uint32_t max_bytes;
  // generic kernel geometry
  uint32_t NA;
  size_t lds_bytes;
  char kernel_name[64];
};

#define HIP_TRY(expr, code)                                                                          \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess) {                                                                         \
      fprintf(stderr, "[sdrfm] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return (code);                                                                                 \
    }                                                                                                \
  } while (0)

static uint32_t max_audio_for(const sdrfm_config& c, uint32_t nbytes) {
  const uint64_t n = nbytes / 2;
  const uint64_t m = (n + c.fir_decim - 1) / c.fir_decim + 1;
  return (uint32_t)((m + c.audio_decim - 1) / c.audio_decim + 1);
}

static void free_handle(sdrfm* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_h) (void)hipFree(h->d_h);
  if (h->d_g) (void)hipFree(h->d_g);
  for (int i = 0; i < 2; ++i) {
    if (h->d_hist_x[i]) (void)hipFree(h->d_hist_x[i]);
    if (h->d_yprev[i]) (void)hipFree(h->d_yprev[i]);
    if (h->d_hist_d[i]) (void)hipFree(h->d_hist_d[i]);
  }
  if (h->d_iq) (void)hipFree(h->d_iq);
  if (h->d_audio) (void)hipFree(h->d_audio);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  free(const_cast<float*>(h->cfg.fir_coeffs));
  free(const_cast<float*>(h->cfg.audio_coeffs));
  delete h;
}

static int ensure_staging(sdrfm* h) {
  if (h->d_iq) return SDRFM_OK;
  const size_t ns = h->cfg.n_streams;
  h->d_iq_stride = ((size_t)h->max_bytes + 255) & ~(size_t)255;
  h->d_audio_stride = (max_audio_for(h->cfg, h->max_bytes) + 63) & ~(size_t)63;
  HIP_TRY(hipMalloc(&h->d_iq, ns * h->d_iq_stride), SDRFM_ENOMEM);
  HIP_TRY(hipMalloc(&h->d_audio, ns * h->d_audio_stride * sizeof(float)), SDRFM_ENOMEM);
  return SDRFM_OK;
}

extern "C" {

uint32_t sdrfm_abi_version(void) { return SDRFM_ABI_VERSION; }

const char* sdrfm_strerror(int status) {
  switch (status) {
    case SDRFM_OK: return "ok";
    case SDRFM_BUSY: return "busy";
    case SDRFM_FAIL: return "HIP runtime failure during processing";
    case SDRFM_NOT_SUPPORTED: return "not supported";
    case SDRFM_UNRECOVERED_ERROR: return "unrecovered error";
    case SDRFM_EINVAL: return "invalid argument";
    case SDRFM_EODD: return "byte count is not a whole number of I/Q pairs";
    case SDRFM_ECAPACITY: return "buffer capacity exceeded";
    case SDRFM_NO_DEVICE: return "no usable gfx950 HIP device (this library has no CPU fallback)";
    case SDRFM_ENOMEM: return "out of (device) memory";
    default: return "unknown status";
  }
}

int sdrfm_create(const sdrfm_config* cfg, sdrfm_t** out) {
  if (!out) return SDRFM_EINVAL;
  *out = nullptr;
  if (!cfg || cfg->struct_size != sizeof(sdrfm_config)) return SDRFM_EINVAL;
  if (!cfg->n_streams || !cfg->fir_coeffs || !cfg->audio_coeffs || cfg->flags) return SDRFM_EINVAL;
  if (!cfg->fir_taps || cfg->fir_taps > SDRFM_MAX_TAPS || !cfg->audio_taps || cfg->audio_taps > SDRFM_MAX_TAPS)
    return SDRFM_EINVAL;
  if (!cfg->fir_decim || cfg->fir_decim > SDRFM_MAX_DECIM || !cfg->audio_decim || cfg->audio_decim > SDRFM_MAX_DECIM)
    return SDRFM_EINVAL;
  for (uint32_t k = 0; k < cfg->fir_taps; ++k)
    if (!std::isfinite(cfg->fir_coeffs[k])) return SDRFM_EINVAL;
  for (uint32_t k = 0; k < cfg->audio_taps; ++k)
    if (!std::isfinite(cfg->audio_coeffs[k])) return SDRFM_EINVAL;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SDRFM_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return SDRFM_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return SDRFM_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fprintf(stderr, "[sdrfm] device %d is %s; this library carries gfx950 code only\n", cfg->device, prop.gcnArchName);
    return SDRFM_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(cfg->device), SDRFM_NO_DEVICE);

  sdrfm* h = new (std::nothrow) sdrfm();
  if (!h) return SDRFM_ENOMEM;
  memset(static_cast<void*>(h), 0, sizeof(*h));
  h->cfg = *cfg;
  h->device = cfg->device;
  h->max_bytes = cfg->max_bytes_per_call ? cfg->max_bytes_per_call : (1u << 20);
  h->max_bytes &= ~1u;
  float* hc = (float*)malloc(sizeof(float) * cfg->fir_taps);
  float* gc = (float*)malloc(sizeof(float) * cfg->audio_taps);
  h->cfg.fir_coeffs = hc;
  h->cfg.audio_coeffs = gc;
  if (!hc || !gc) { free_handle(h); return SDRFM_ENOMEM; }
  memcpy(hc, cfg->fir_coeffs, sizeof(float) * cfg->fir_taps);
  memcpy(gc, cfg->audio_coeffs, sizeof(float) * cfg->audio_taps);

  const size_t ns = cfg->n_streams, T = cfg->fir_taps, Ta = cfg->audio_taps;
  const size_t hx = (T > 1 ? T - 1 : 1), hd = (Ta > 1 ? Ta - 1 : 1);
#define CR(expr) do { if ((expr) != hipSuccess) { free_handle(h); return SDRFM_ENOMEM; } } while (0)
  CR(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  CR(hipMalloc(&h->d_h, sizeof(float) * T));
  CR(hipMalloc(&h->d_g, sizeof(float) * Ta));
  for (int i = 0; i < 2; ++i) {
    CR(hipMalloc(&h->d_hist_x[i], sizeof(float2) * ns * hx));
    CR(hipMalloc(&h->d_yprev[i], sizeof(float2) * ns));
    CR(hipMalloc(&h->d_hist_d[i], sizeof(float) * ns * hd));
  }
  CR(hipMemcpy(h->d_h, hc, sizeof(float) * T, hipMemcpyHostToDevice));
  CR(hipMemcpy(h->d_g, gc, sizeof(float) * Ta, hipMemcpyHostToDevice));
#undef CR

  // generic-kernel tile: as many audio outputs per block as fit ~48 KiB of LDS, capped at 64
  uint32_t NA = 64;
  for (;;) {
    const size_t ND = (size_t)(NA - 1) * cfg->audio_decim + Ta, NY = ND + 1, NX = (NY - 1) * cfg->fir_decim + T;
    h->lds_bytes = NX * 8 + NY * 8 + ND * 4 + T * 4 + Ta * 4;
    if (h->lds_bytes <= 48 * 1024 || NA == 1) break;
    NA /= 2;
  }
  if (h->lds_bytes > 160 * 1024) { free_handle(h); return SDRFM_NOT_SUPPORTED; }
  h->NA = NA;
  if (h->lds_bytes > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_generic), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h->lds_bytes) != hipSuccess) { free_handle(h); return SDRFM_NOT_SUPPORTED; }
  }
  snprintf(h->kernel_name, sizeof(h->kernel_name), "generic T%u D%u Ta%u Da%u NA%u", cfg->fir_taps, cfg->fir_decim,
           cfg->audio_taps, cfg->audio_decim, NA);
  const int rc = sdrfm_reset(h);
  if (rc != SDRFM_OK) { free_handle(h); return rc; }
  *out = h;
  return SDRFM_OK;
}

void sdrfm_destroy(sdrfm_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  free_handle(h);
}

int sdrfm_reset(sdrfm_t* h) {
  if (!h) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  const size_t ns = h->cfg.n_streams, T = h->cfg.fir_taps, Ta = h->cfg.audio_taps;
  const size_t hx = (T > 1 ? T - 1 : 1), hd = (Ta > 1 ? Ta - 1 : 1);
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipMemsetAsync(h->d_hist_x[i], 0, sizeof(float2) * ns * hx, h->stream), SDRFM_FAIL);
    HIP_TRY(hipMemsetAsync(h->d_yprev[i], 0, sizeof(float2) * ns, h->stream), SDRFM_FAIL);
    HIP_TRY(hipMemsetAsync(h->d_hist_d[i], 0, sizeof(float) * ns * hd, h->stream), SDRFM_FAIL);
  }
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  h->cur = 0;
  h->phase_x = h->phase_d = 0;
  return SDRFM_OK;
}

int sdrfm_audio_count(const sdrfm_t* h, uint32_t nbytes, uint32_t* n_audio) {
  if (!h || !n_audio) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  const uint64_t N = nbytes / 2;
  const uint64_t M = (h->phase_x + N) / h->cfg.fir_decim;
  *n_audio = (uint32_t)((h->phase_d + M) / h->cfg.audio_decim);
  return SDRFM_OK;
}

int sdrfm_set_stream(sdrfm_t* h, void* hip_stream) {
  if (!h) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return SDRFM_OK;
}

int sdrfm_synchronize(sdrfm_t* h) {
  if (!h) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

const char* sdrfm_kernel_name(const sdrfm_t* h) { return h ? h->kernel_name : ""; }

// Enqueue one call on device-resident buffers and advance the host-side phases / state set.
static int enqueue(sdrfm* h, const uint8_t* d_iq, size_t iq_stride, uint32_t nbytes, float* d_audio, size_t audio_stride,
                   uint32_t* n_audio) {
  const sdrfm_config& c = h->cfg;
  const uint32_t N = nbytes / 2;
  const uint32_t M = (uint32_t)(((uint64_t)h->phase_x + N) / c.fir_decim);
  const uint32_t A = (uint32_t)(((uint64_t)h->phase_d + M) / c.audio_decim);
  if (n_audio) *n_audio = A;
  if (N == 0) return SDRFM_OK;
  if (A > audio_stride && c.n_streams > 1) return SDRFM_ECAPACITY;

  CallParams p;
  p.iq = d_iq; p.iq_stride = iq_stride; p.audio = d_audio; p.audio_stride = audio_stride;
  p.hist_x_in = h->d_hist_x[h->cur]; p.hist_x_out = h->d_hist_x[h->cur ^ 1];
  p.yprev_in = h->d_yprev[h->cur]; p.yprev_out = h->d_yprev[h->cur ^ 1];
  p.hist_d_in = h->d_hist_d[h->cur]; p.hist_d_out = h->d_hist_d[h->cur ^ 1];
  p.h = h->d_h; p.g = h->d_g;
  p.T = c.fir_taps; p.D = c.fir_decim; p.Ta = c.audio_taps; p.Da = c.audio_decim;
  p.N = N; p.M = M; p.A = A;
  p.e0 = (int32_t)(c.fir_decim - 1 - h->phase_x);
  p.f0 = (int32_t)(c.audio_decim - 1 - h->phase_d);
  p.NA = h->NA;
  p.tiles_per_stream = (A + h->NA - 1) / h->NA;
  p.n_streams = c.n_streams;
  const uint32_t grid = c.n_streams * p.tiles_per_stream + c.n_streams;
  hipLaunchKernelGGL(k_generic, dim3(grid), dim3(256), h->lds_bytes, h->stream, p);
  HIP_TRY(hipGetLastError(), SDRFM_FAIL);

  h->cur ^= 1;
  h->phase_x = (uint32_t)(((uint64_t)h->phase_x + N) % c.fir_decim);
  h->phase_d = (uint32_t)(((uint64_t)h->phase_d + M) % c.audio_decim);
  return SDRFM_OK;
}

int sdrfm_process_batch(sdrfm_t* h, const uint8_t* iq, size_t iq_stride, uint32_t nbytes, float* audio,
                        size_t audio_stride, uint32_t* n_audio, uint32_t flags) {
  if (!h || !n_audio) return SDRFM_EINVAL;
  if (flags & ~SDRFM_F_DEVICE_PTRS) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  if (nbytes == 0) { *n_audio = 0; return SDRFM_OK; }
  if (!iq) return SDRFM_EINVAL;
  const uint32_t ns = h->cfg.n_streams;
  if (ns > 1 && iq_stride < nbytes) return SDRFM_ECAPACITY;
  uint32_t A = 0;
  (void)sdrfm_audio_count(h, nbytes, &A);
  if (A && !audio) return SDRFM_EINVAL;
  if (ns > 1 && audio_stride < A) return SDRFM_ECAPACITY;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);

  if (flags & SDRFM_F_DEVICE_PTRS) return enqueue(h, iq, iq_stride, nbytes, audio, audio_stride, n_audio);

  // host buffers: stage -> kernels -> copy back, synchronous (the caller may re-arm `iq` as soon as we return,
  // like the reference FSM does with CommItf.buff)
  if (nbytes > h->max_bytes) return SDRFM_ECAPACITY;
  int rc = ensure_staging(h);
  if (rc != SDRFM_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(h->d_iq, h->d_iq_stride, iq, ns > 1 ? iq_stride : nbytes, nbytes, ns, hipMemcpyHostToDevice,
                           h->stream), SDRFM_FAIL);
  rc = enqueue(h, h->d_iq, h->d_iq_stride, nbytes, h->d_audio, h->d_audio_stride, n_audio);
  if (rc != SDRFM_OK) return rc;
  if (A)
    HIP_TRY(hipMemcpy2DAsync(audio, (ns > 1 ? audio_stride : A) * sizeof(float), h->d_audio,
                             h->d_audio_stride * sizeof(float), A * sizeof(float), ns, hipMemcpyDeviceToHost, h->stream),
            SDRFM_FAIL);
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

int sdrfm_process(sdrfm_t* h, const uint8_t* iq, uint32_t nbytes, float* audio, uint32_t audio_cap, uint32_t* n_audio) {
  if (!h || !n_audio) return SDRFM_EINVAL;
  if (h->cfg.n_streams != 1) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  uint32_t A = 0;
  (void)sdrfm_audio_count(h, nbytes, &A);
  if (A > audio_cap) return SDRFM_ECAPACITY;
  return sdrfm_process_batch(h, iq, nbytes, nbytes, audio, audio_cap, n_audio, 0);
}

/* Host evaluation of the device's atan2 / discriminator arithmetic (same header, same rounding) so that its accuracy
 * can be unit-tested without a GPU.  Not used by any compute path. */
float sdrfm_host_atan2f(float y, float x) { return (x == 0.0f && y == 0.0f) ? 0.0f : sdrfm_atan2f(y, x); }
float sdrfm_host_discriminate(float yr, float yi, float pr, float pi) { return sdrfm_discriminate(yr, yi, pr, pi); }

}  // extern "C"
