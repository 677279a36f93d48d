// sdrfm_spectrum.hip — spectrum view of the IQ buffer (reference README.md:29), gfx950 only.
//
// Spec (build-defined, oracle/sdrfm_spectrum_oracle.c): frames of N samples, x = ((I-127.5)*w[n], (Q-127.5)*w[n]), one fixed
// radix-2 DIT graph (bit-reversed input; T = W*B with T.re = fmaf(W.re,B.re,-(W.im*B.im)), T.im = fmaf(W.re,B.im,W.im*B.re);
// lo = A+T, hi = A-T), P = fmaf(re,re,im*im), S += P in frame order, out = fftshift(S) * (1/F).  Every operation is the
// oracle's, in an order the graph allows, so the result is bit-identical.
//
// Kernel: one workgroup per stream walks the stream's frames in rounds of NWF = 8 (N <= 1024) frames, ONE WAVE PER FRAME:
// a frame lives in that wave's own LDS region as float2[N] and its FFT needs no workgroup barrier (a wave's LDS
// operations execute in order).  Samples arrive through typed buffer loads (u8 pair -> 2 floats in the texture unit), one
// round ahead of the FFT; the DIT stages are applied two at a time (a lane takes the 4 points that two consecutive stages
// couple, so each pass reads and writes a point once).  After a round the frames' powers are added to the running sums
// (registers, N/threads bins per thread) in frame order — the spec's sum over frames is sequential — which costs two
// workgroup barriers per round instead of ~7 per frame.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/sdrfm.h"

typedef float sf2_t __attribute__((ext_vector_type(2)));
typedef int si4_t __attribute__((ext_vector_type(4)));
__device__ sf2_t spec_typed_load_xy(si4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
// buffer resource word3: dst_sel = (R, G, 0, 1), num_format = USCALED (2), data_format = 8_8 (3)
#define SDRFM_SPEC_RSRC_U8X2 (4 | (5 << 3) | (0 << 6) | (1 << 9) | (2 << 12) | (3 << 15))

namespace {

struct SParams {
  const uint8_t* iq;
  size_t iq_stride;
  uint32_t iq_span;       // bytes from iq to the end of the last stream's frames (descriptor range)
  float* power;
  size_t power_stride;
  const float2* tw;       // N/2 twiddles
  const float* win;       // N window values
  uint32_t F;             // frames per stream
  float inv_frames;       // 1.0f / F
};

__device__ __forceinline__ void butterfly(float2& a, float2& b, float2 w) {   // (A, B) -> (A + W B, A - W B)
  const float tr = __builtin_fmaf(w.x, b.x, -(w.y * b.y));
  const float ti = __builtin_fmaf(w.x, b.y, w.y * b.x);
  const float2 A = a;
  a = make_float2(A.x + tr, A.y + ti);
  b = make_float2(A.x - tr, A.y - ti);
}

__device__ __forceinline__ void wave_sync() {                  // LDS written by this wave is visible to all of its lanes
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// LDS index padding: a DIT pass touches points at power-of-two strides (and the bit-reversed scatter at stride N/64), which
// without padding put all 64 lanes into 4 of the 16 float2 bank slots; one pad slot per 16 and per 256 elements makes every
// access pattern of every pass conflict-free (checked exhaustively for N = 256..4096, tools/fft_lds_padding.py)
__host__ __device__ constexpr int spad(int i) { return i + (i >> 4) + (i >> 8); }

// Stages S .. S+K-1 of the radix-2 DIT graph in ONE pass over LDS: the 2^K points {base + c h}, h = 2^(S-1), are closed under
// these K stages, so a lane loads them, applies the K layers of butterflies in registers and stores them — one LDS round
// trip per K stages instead of one per stage.  Every butterfly is the spec's (same operands, same twiddle
// tw[position_in_stage * N/m]); only the order of independent butterflies differs from the oracle's loops.
// LOGB >= LOGN: the wave's block holds 2^(LOGB-LOGN) independent frames side by side; stages <= LOGN never couple points of
// different frames, so the same pass transforms all of them at once (index arithmetic over the block, twiddles of N).
// PWO != nullptr (last pass only, kernels with a separate power region): the pass stores |point|^2 at PWO[point index] instead of
// writing the points back — the power phase's LDS round trip (16 writes + 16 reads of 8 bytes per lane and frame) disappears.
template <int LOGB, int LOGN, int S, int K>
__device__ __forceinline__ void fft_pass(float2* X, const float2* TW, const float2 (&W1)[8], float* PWO, int lane) {
  constexpr int H = 1 << (S - 1), G = 1 << K, NG = (1 << LOGB) >> K;
  constexpr int UNR = LOGB >= 12 ? 1 : 4;                      // (4096 points: 16 points x 4 groups unrolled would not fit in VGPRs)
#pragma unroll UNR
  for (int g0 = 0; g0 < NG; g0 += 64) {
    const int g = g0 + lane;
    if (NG >= 64 || g < NG) {
      const int pos = g & (H - 1), base = ((g >> (S - 1)) << (S - 1 + K)) + pos;   // (pos: position within the stage-S block)
      float2 v[G];
#pragma unroll
      for (int c = 0; c < G; ++c) v[c] = X[spad(base + c * H)];
#pragma unroll
      for (int t = 0; t < K; ++t) {                            // stage S + t: half = H 2^t, partner c ^ 2^t
#pragma unroll
        for (int c = 0; c < G; ++c) {
          if ((c >> t) & 1) continue;
          const int pos_t = pos + (c & ((1 << t) - 1)) * H;    // position of the pair within its stage-(S+t) block
          if constexpr (S == 1 && K == 4) butterfly(v[c], v[c + (1 << t)], W1[(c & ((1 << t) - 1)) << (3 - t)]);   // first pass: the 8 twiddles W16^k, wave-uniform registers
          else butterfly(v[c], v[c + (1 << t)], TW[spad(pos_t << (LOGN - S - t))]);
        }
      }
      if (S + K - 1 == LOGN && PWO) {
#pragma unroll
        for (int c = 0; c < G; ++c) PWO[base + c * H] = __builtin_fmaf(v[c].x, v[c].x, v[c].y * v[c].y);
      } else {
#pragma unroll
        for (int c = 0; c < G; ++c) X[spad(base + c * H)] = v[c];
      }
    }
  }
  wave_sync();
}
template <int LOGB, int LOGN, int S>
__device__ __forceinline__ void fft_passes(float2* X, const float2* TW, const float2 (&W1)[8], float* PWO, int lane) {
  if constexpr (S <= LOGN) {
    // up to 4 stages per pass, but no more than leaves a group (2^K points) for each of the 64 lanes
    constexpr int KMAX = (LOGB - 6) >= 4 ? 4 : ((LOGB - 6) >= 2 ? (LOGB - 6) : 2);
    constexpr int K = (LOGN - S + 1) >= KMAX ? KMAX : (LOGN - S + 1);
    fft_pass<LOGB, LOGN, S, K>(X, TW, W1, PWO, lane);
    fft_passes<LOGB, LOGN, S + K>(X, TW, W1, PWO, lane);
  }
}

// First pass (stages 1..4) straight from registers: lane l holds the 16 points at bit-reversed-order positions 16 l .. 16 l + 15 of the
// wave's 1024-point block (loaded from global memory in exactly that pattern: for a fixed register the 64 lanes read a permutation of
// 64 consecutive samples, so the loads coalesce as before).  Saves the scatter store and the first pass's reads (16 + 16 LDS accesses
// of 8 bytes per lane and frame).
__device__ __forceinline__ void fft_first_pass_from_regs(float2 (&v)[16], float2* X, const float2 (&W1)[8], int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if ((c >> t) & 1) continue;
      butterfly(v[c], v[c + (1 << t)], W1[(c & ((1 << t) - 1)) << (3 - t)]);
    }
#pragma unroll
  for (int c = 0; c < 16; ++c) X[spad(16 * lane + c)] = v[c];
  wave_sync();
}

// waves per workgroup, one block of frames each: 8, or 4 for 4096 points (LDS: 160 KiB per CU); the short-frame kernels fit
// 128 VGPRs without the prefetch registers and run 16 waves, which hide the load latency instead
constexpr int spec_nwf(int logn) { return logn <= 8 ? 16 : (logn <= 11 ? 8 : 4); }
// points per wave and round: a frame, or for N < 1024 as many whole frames as make 1024 points (all 64 lanes stay busy)
constexpr int spec_logb(int logn) { return logn < 10 ? 10 : logn; }
// kernels whose LDS has room for two separate power regions (NWF blocks of floats each, written alternately) fuse the power into the
// last FFT pass and need one workgroup barrier per round instead of two
constexpr bool spec_fusep(int logn) { return logn == 9 || logn == 10; }

template <int LOGN>
__global__ void __launch_bounds__(64 * spec_nwf(LOGN)) k_spectrum(SParams p) {
  constexpr int N = 1 << LOGN, NWF = spec_nwf(LOGN), NT = 64 * NWF;
  constexpr int LOGB = spec_logb(LOGN), B = 1 << LOGB, FPW = B / N, FPR = FPW * NWF;   // block, frames per wave / per round
  constexpr int PPL = B / 64;                                  // points per lane of a wave's block
  constexpr int PPT = (N + NT - 1) / NT;                       // bins per thread of the running sum
  extern __shared__ __attribute__((aligned(16))) unsigned char spec_smem[];
  constexpr int NPT = spad(N / 2 - 1) + 1, NPX = spad(B - 1) + 1;   // padded sizes of the twiddle table and of a block
  float2* TW = reinterpret_cast<float2*>(spec_smem);           // N/2 twiddles at TW[spad(t)]
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float2* X = reinterpret_cast<float2*>(spec_smem) + NPT + wv * NPX;   // this wave's block: point i at X[spad(i)]
  constexpr bool FUSEP = spec_fusep(LOGN);
  constexpr int PWS = FUSEP ? B : 2 * NPX;                     // floats between the power blocks of consecutive waves
  float* const PWsep = reinterpret_cast<float*>(reinterpret_cast<float2*>(spec_smem) + NPT + NWF * NPX);   // (FUSEP) separate region
  const float* PW = FUSEP ? PWsep : reinterpret_cast<const float*>(reinterpret_cast<float2*>(spec_smem) + NPT);   // wave w's powers at PW[PWS w + i]
  const uint32_t stream = blockIdx.x;
  for (int i = tid; i < N / 2; i += NT) TW[spad(i)] = p.tw[i];
  float2 W1[8];                                                // the first pass's twiddles W16^k = tw[k N/16] (wave-uniform)
#pragma unroll
  for (int k = 0; k < 8; ++k) W1[k] = N >= 16 ? p.tw[k * (N / 16)] : make_float2(1.f, 0.f);
  // up to 1024 points the lane keeps its window values and the next round's samples in registers; longer frames would
  // need > 256 VGPRs for that, so they read the window through the cache and load their samples when they need them
  constexpr bool REGS = (LOGB <= 10) && (NWF <= 8);
  static_assert(!REGS || PPL == 16, "the register path holds one 16-point first-pass group per lane");
  // (REGS) register q of a lane holds the point at bit-reversed-order position u = 16 lane + q of the block = sample regs_sample(q)
  auto regs_sample = [&](int q) -> int {
    const uint32_t u = (uint32_t)(16 * lane + q);
    return (int)((u & ~(uint32_t)(N - 1)) | (__brev(u & (uint32_t)(N - 1)) >> (32 - LOGN)));
  };
  constexpr int PR = REGS ? PPL : 1;
  float wv_win[PR], S[PPT];
  if constexpr (REGS) {
#pragma unroll
    for (int q = 0; q < PPL; ++q) wv_win[q] = p.win[regs_sample(q) & (N - 1)];
  }
#pragma unroll
  for (int q = 0; q < PPT; ++q) S[q] = 0.0f;
  const unsigned long long ga = (unsigned long long)p.iq;
  const si4_t rsrc = {(int)(unsigned)ga, (int)(unsigned)(ga >> 32), (int)p.iq_span, SDRFM_SPEC_RSRC_U8X2};
  const uint32_t sbase = stream * (uint32_t)p.iq_stride;
  sf2_t cur[PR];
  auto fetch = [&](uint32_t f) {                               // the block starting at frame f -> cur (out-of-range reads return 0)
    if constexpr (REGS) {
#pragma unroll
      for (int q = 0; q < PPL; ++q)
        cur[q] = spec_typed_load_xy(rsrc, (int)(sbase + 2u * (f * (uint32_t)N + (uint32_t)regs_sample(q))), 0, 0);
    }
  };
  if ((uint32_t)(wv * FPW) < p.F) fetch((uint32_t)(wv * FPW));
  __syncthreads();                                             // TW visible
  uint32_t par = 0;                                            // (FUSEP) which of the two power regions this round writes
  for (uint32_t f0 = 0; f0 < p.F; f0 += FPR, par ^= 1u) {      // a round: frames f0 .. f0+FPR-1, FPW consecutive ones per wave
    const uint32_t f = f0 + (uint32_t)(wv * FPW);
    if (f < p.F) {                                             // (wave-uniform; frames past F in the block are computed, not summed)
      float2 v1[16];
      if constexpr (REGS) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v1[q] = make_float2((cur[q].x - 127.5f) * wv_win[q], (cur[q].y - 127.5f) * wv_win[q]);
      } else {
#pragma unroll 8
        for (int q = 0; q < PPL; ++q) {
          const int n = lane + 64 * q;
          const sf2_t c = spec_typed_load_xy(rsrc, (int)(sbase + 2u * (f * (uint32_t)N + (uint32_t)n)), 0, 0);
          const float wn = p.win[n & (N - 1)];
          X[spad((int)((uint32_t)(n & ~(N - 1)) | (__brev((uint32_t)(n & (N - 1))) >> (32 - LOGN))))] = make_float2((c.x - 127.5f) * wn, (c.y - 127.5f) * wn);
        }
      }
      if (f + FPR < p.F) fetch(f + FPR);                       // next round's bytes: in flight during this block's FFTs
      // DIT stages in passes of up to 4 stages, each pass entirely in registers (see fft_pass)
      if constexpr (REGS) {
        fft_first_pass_from_regs(v1, X, W1, lane);
        fft_passes<LOGB, LOGN, 5>(X, TW, W1, FUSEP ? PWsep + par * (NWF * B) + wv * B : nullptr, lane);
      } else {
        wave_sync();
        fft_passes<LOGB, LOGN, 1>(X, TW, W1, FUSEP ? PWsep + wv * B : nullptr, lane);
      }
      if constexpr (!FUSEP) {
        // powers of this block, written over the start of its own region (PW[2 NPX wv + i]) in blocks of 16 points per lane:
        // block b overwrites float slots [1024 b, 1024 b + 1024), i.e. float2 slots below 512 (b + 1) — points already consumed
        // (a point's padded slot is never below its index)
        constexpr int PB = PPL < 16 ? PPL : 16;
        for (int q0 = 0; q0 < PPL; q0 += PB) {
          float pw[PB];
#pragma unroll
          for (int q = 0; q < PB; ++q) {
            const float2 v = X[spad(lane + 64 * (q0 + q))];
            pw[q] = __builtin_fmaf(v.x, v.x, v.y * v.y);
          }
          wave_sync();
#pragma unroll
          for (int q = 0; q < PB; ++q) reinterpret_cast<float*>(X)[lane + 64 * (q0 + q)] = pw[q];
          wave_sync();
        }
      }
    }
    __syncthreads();
    // the spec's sum over frames is sequential: add this round's frames in frame order (frame o of the round = frame
    // o mod FPW of wave o / FPW).  The adds are a serial chain by the spec; the LDS reads are not: eight frames' rows are fetched at once and
    // then added in order (one LDS latency per eight frames and bin instead of one per frame).  A thread whose bin index is past N
    // (N < threads) adds bin k mod N into a sum it never stores.
    const uint32_t nfr = (p.F - f0) < (uint32_t)FPR ? (p.F - f0) : (uint32_t)FPR;
    const float* const PWr = PW + (FUSEP ? par * (NWF * B) : 0u) + (uint32_t)(tid & (N - 1));
    auto pw_at = [&](uint32_t o, int q) -> float { return PWr[PWS * (o / FPW) + (o % FPW) * N + (uint32_t)((NT * q) & (N - 1))]; };
    uint32_t o = 0;
    for (; o + 8 <= nfr; o += 8) {
      float t[PPT][8];
#pragma unroll
      for (int q = 0; q < PPT; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) t[q][j] = pw_at(o + (uint32_t)j, q);
#pragma unroll
      for (int q = 0; q < PPT; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) S[q] = S[q] + t[q][j];
    }
    for (; o < nfr; ++o)
#pragma unroll
      for (int q = 0; q < PPT; ++q) S[q] = S[q] + pw_at(o, q);
    if constexpr (!FUSEP) __syncthreads();                     // the blocks are rewritten by the next round (FUSEP: the next round writes
                                                               // the OTHER power region; the barrier above orders this sum before the round after)
  }
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    const int k = tid + NT * q;
    if (k < N) p.power[(size_t)stream * p.power_stride + (size_t)((k + N / 2) & (N - 1))] = S[q] * p.inv_frames;
  }
}

typedef void (*spec_kernel_t)(SParams);
spec_kernel_t pick_kernel(uint32_t logn) {
  switch (logn) {
    case 6: return k_spectrum<6>;
    case 7: return k_spectrum<7>;
    case 8: return k_spectrum<8>;
    case 9: return k_spectrum<9>;
    case 10: return k_spectrum<10>;
    case 11: return k_spectrum<11>;
    case 12: return k_spectrum<12>;
    default: return nullptr;
  }
}

}  // namespace

struct sdrfm_spectrum {
  sdrfm_spectrum_config cfg;
  int device;
  uint32_t logn, max_bytes;
  hipStream_t own_stream, stream;
  float2* d_tw;
  float* d_win;
  uint8_t* d_iq; size_t d_iq_stride;
  float* d_power;
  spec_kernel_t kernel;
  size_t lds_bytes;
};

#define STRY(expr, code)                                                                                     \
  do {                                                                                                       \
    hipError_t e__ = (expr);                                                                                 \
    if (e__ != hipSuccess) {                                                                                 \
      fprintf(stderr, "[sdrfm_spectrum] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return (code);                                                                                         \
    }                                                                                                        \
  } while (0)

static void sfree(sdrfm_spectrum* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  void* ptrs[] = {h->d_tw, h->d_win, h->d_iq, h->d_power};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

extern "C" {

int sdrfm_spectrum_create(const sdrfm_spectrum_config* cfg, sdrfm_spectrum_t** out) {
  if (!out) return SDRFM_EINVAL;
  *out = nullptr;
  if (!cfg || cfg->struct_size != sizeof(sdrfm_spectrum_config) || cfg->flags || !cfg->n_streams) return SDRFM_EINVAL;
  uint32_t logn = 0;
  while ((1u << logn) < cfg->nfft) ++logn;
  if ((1u << logn) != cfg->nfft || logn < 6 || logn > 12) return SDRFM_EINVAL;
  if (cfg->window) for (uint32_t n = 0; n < cfg->nfft; ++n) if (!std::isfinite(cfg->window[n])) return SDRFM_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return SDRFM_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SDRFM_NO_DEVICE;
  sdrfm_spectrum* h = new sdrfm_spectrum();
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg; h->cfg.window = nullptr;
  h->device = cfg->device; h->logn = logn;
  h->max_bytes = cfg->max_bytes_per_call ? cfg->max_bytes_per_call : (1u << 20);
  h->kernel = pick_kernel(logn);
  h->lds_bytes = ((size_t)(spad((int)cfg->nfft / 2 - 1) + 1) + (size_t)spec_nwf((int)logn) * (size_t)(spad((1 << spec_logb((int)logn)) - 1) + 1)) * sizeof(float2) +
                 (spec_fusep((int)logn) ? 2 * (size_t)spec_nwf((int)logn) * ((size_t)1 << spec_logb((int)logn)) * sizeof(float) : 0);   // two power regions
#define CR(expr) do { if ((expr) != hipSuccess) { sfree(h); return SDRFM_ENOMEM; } } while (0)
  CR(hipSetDevice(h->device));
  if (h->lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(h->kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) {
    sfree(h);
    return SDRFM_NOT_SUPPORTED;
  }
  CR(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  const uint32_t N = cfg->nfft;
  float2* tw = (float2*)malloc(sizeof(float2) * (N / 2));
  float* win = (float*)malloc(sizeof(float) * N);
  const double two_pi = 6.283185307179586476925286766559;
  for (uint32_t t = 0; t < N / 2; ++t) tw[t] = make_float2((float)cos(two_pi * (double)t / (double)N), (float)(-sin(two_pi * (double)t / (double)N)));
  for (uint32_t n = 0; n < N; ++n) win[n] = cfg->window ? cfg->window[n] : (float)(0.5 - 0.5 * cos(two_pi * (double)n / (double)N));
  hipError_t e = hipMalloc(&h->d_tw, sizeof(float2) * (N / 2));
  if (e == hipSuccess) e = hipMalloc(&h->d_win, sizeof(float) * N);
  if (e == hipSuccess) e = hipMemcpy(h->d_tw, tw, sizeof(float2) * (N / 2), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_win, win, sizeof(float) * N, hipMemcpyHostToDevice);
  free(tw); free(win);
  CR(e);
#undef CR
  *out = h;
  return SDRFM_OK;
}

void sdrfm_spectrum_destroy(sdrfm_spectrum_t* h) { sfree(h); }

int sdrfm_spectrum_set_stream(sdrfm_spectrum_t* h, void* hip_stream) {
  if (!h) return SDRFM_EINVAL;
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return SDRFM_OK;
}

int sdrfm_spectrum_synchronize(sdrfm_spectrum_t* h) {
  if (!h) return SDRFM_EINVAL;
  STRY(hipSetDevice(h->device), SDRFM_FAIL);
  STRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

static int senqueue(sdrfm_spectrum* h, const uint8_t* d_iq, size_t iq_stride, uint32_t nbytes, float* d_power, size_t power_stride,
                    uint32_t F) {
  const uint32_t ns = h->cfg.n_streams, N = h->cfg.nfft;
  if (F == 0) {                                               // nothing to view: all zeros
    STRY(hipMemset2DAsync(d_power, power_stride * sizeof(float), 0, N * sizeof(float), ns, h->stream), SDRFM_FAIL);
    return SDRFM_OK;
  }
  const uint64_t span = (uint64_t)(ns - 1) * iq_stride + 2ull * F * N;
  if (span >= (1ull << 32)) return SDRFM_ECAPACITY;           // one buffer descriptor spans the batch
  (void)nbytes;
  SParams p;
  p.iq = d_iq; p.iq_stride = iq_stride; p.iq_span = (uint32_t)span; p.power = d_power; p.power_stride = power_stride;
  p.tw = h->d_tw; p.win = h->d_win; p.F = F; p.inv_frames = 1.0f / (float)F;
  hipLaunchKernelGGL(h->kernel, dim3(ns), dim3(64 * spec_nwf((int)h->logn)), h->lds_bytes, h->stream, p);
  STRY(hipGetLastError(), SDRFM_FAIL);
  return SDRFM_OK;
}

int sdrfm_spectrum_process_batch(sdrfm_spectrum_t* h, const uint8_t* iq, size_t iq_stride, uint32_t nbytes, float* power,
                                 size_t power_stride, uint32_t* n_frames, uint32_t flags) {
  if (!h || !n_frames || !power) return SDRFM_EINVAL;
  if (flags & ~SDRFM_F_DEVICE_PTRS) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  if (nbytes && !iq) return SDRFM_EINVAL;
  if (nbytes > h->max_bytes) return SDRFM_ECAPACITY;
  const uint32_t ns = h->cfg.n_streams, N = h->cfg.nfft;
  if (ns > 1 && iq_stride < nbytes) return SDRFM_ECAPACITY;
  if (power_stride < N) return SDRFM_ECAPACITY;
  const uint32_t F = (nbytes / 2) / N;
  *n_frames = F;
  STRY(hipSetDevice(h->device), SDRFM_FAIL);
  if (flags & SDRFM_F_DEVICE_PTRS) return senqueue(h, iq, iq_stride, nbytes, power, power_stride, F);
  if (!h->d_iq) {
    h->d_iq_stride = ((size_t)h->max_bytes + 255) & ~(size_t)255;
    STRY(hipMalloc(&h->d_iq, ns * h->d_iq_stride), SDRFM_ENOMEM);
    STRY(hipMalloc(&h->d_power, sizeof(float) * ns * N), SDRFM_ENOMEM);
  }
  if (nbytes)
    STRY(hipMemcpy2DAsync(h->d_iq, h->d_iq_stride, iq, ns > 1 ? iq_stride : nbytes, nbytes, ns, hipMemcpyHostToDevice, h->stream), SDRFM_FAIL);
  const int rc = senqueue(h, h->d_iq, h->d_iq_stride, nbytes, h->d_power, N, F);
  if (rc != SDRFM_OK) return rc;
  STRY(hipMemcpy2DAsync(power, power_stride * sizeof(float), h->d_power, N * sizeof(float), N * sizeof(float), ns, hipMemcpyDeviceToHost, h->stream), SDRFM_FAIL);
  STRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

}  // extern "C"
