/*
 * sdrfm_wbfm.hip — multi-channel WBFM path (BASELINE configs[4]) behind the sdrfm_wbfm_* entry points of include/sdrfm.h:
 * 16-band critically-sampled polyphase channelizer -> per-band FM discriminator -> rational L/M audio resampler.
 *
 * Same boundary as the narrow-band path: it consumes the RTL2832 bulk-IN buffer (RTLSDR_CommItfTypedef,
 * Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Inc/usbh_rtlsdr.h:165-173) from the hook the reference leaves empty
 * (usbh_rtlsdr.c:1094-1097).  Arithmetic: DESIGN.md "WBFM spec" — fp32 fmaf chains oldest-first, a fixed radix-2 DIT
 * 16-point DFT graph, the K3 discriminator of sdrfm_math.h.  Three bit-identical implementations:
 *   k_wbfm_steps : the product path for the BASELINE shape (P = 128, <= 10 resampler taps per phase, 4 L <= M): one lane per
 *                  channelizer step, DFT in registers, everything in one kernel (see the comment above the kernel)
 *   k_wbfm_fused : one lane per polyphase branch, DFT across 16 lanes by DPP (round 1; fallback for L = 1 or 4 L > M)
 *   k_wbfm_chan + k_wbfm_res : any prototype length / ratio.  k_wbfm_chan: one block = NT channelizer steps of one stream, x (f32)
 *                  staged in LDS, 16 polyphase branches per step, 16-point DFT per step, discriminator per band -> d scratch in
 *                  HBM [stream][band][t];  k_wbfm_res: one thread = one audio sample of one band; extra blocks hand the d history over
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

#include <type_traits>
typedef int wi4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr int NB = SDRFM_WBFM_BANDS;

template <int I, int N, class F>
__device__ __forceinline__ void static_for_w(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_w<I + 1, N>(f);
  }
}

struct WParams {
  const uint8_t* iq;
  size_t iq_stride;
  uint32_t iq_span;                                              // bytes from iq to the end of the last stream (fused kernel)
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
  const float* pperm;           // step kernel: prototype taps in DFT input order, pperm[16 q + k] = p[bitrev4(k) + 16 q]
  uint32_t inv_L32;             // step kernel: floor(2^32 / L) + 1
  uint32_t swap_pct;            // step kernel: share of a run after which the priority passes from the slot-0 to the slot-1 wave
#ifdef SDRFM_DEV
  unsigned long long* dbg;      // development build: 256 words per wave of phase time stamps (nullptr = off)
  uint32_t dbg_light;           // ... only the entry / exit times (no per-phase waits: the kernel runs at full speed)
  uint32_t ablate;              // timing experiments, WRONG results (SDRFM_WBFM_ABLATE): 1 = the FIR reads 3 of its 8 step groups from LDS and
                                // re-uses registers for the rest — the LDS traffic "several steps per lane" would leave (2.75 group reads per step)
#endif
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

// K3 of the WBFM path (all three kernels, scalar / pair / eight pairs: the same operations, bit-identical to each other): the spec's
// conjugate product (one fused, two rounded products) and the shorter atan2 of design Q — range reduction as in sdrfm_atan2f, a
// 6-coefficient minimax polynomial in s = v^2 (|error| <= 3.9e-7 rad; sdrfm_atan2f: 8 coefficients, 6.5e-8), (0, 0) -> +-0 through a
// clamp of the larger magnitude instead of a select.  A step runs 16 of these: they were 45 % of the step kernel's vector instructions.
// Audio within 7e-7 of the oracle (tolerance 1e-5).
#define WBFM_ATAN_C5 0x1.e34882p-8f
#define WBFM_ATAN_C4 -0x1.22fc74p-5f
#define WBFM_ATAN_C3 0x1.509024p-4f
#define WBFM_ATAN_C2 -0x1.12688cp-3f
#define WBFM_ATAN_C1 0x1.96c562p-3f
#define WBFM_ATAN_C0 -0x1.554086p-2f
__device__ __forceinline__ float wbfm_discriminate(float yr, float yi, float pr, float pi) {
  const float re = __builtin_fmaf(yr, pr, yi * pi);
  const float im = yi * pr - yr * pi;
  const float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
  const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 0x1p-120f), mn = __builtin_fminf(ax, ay);
  const float v = mn * __builtin_amdgcn_rcpf(mx);
  const float s2 = v * v;
  float q = WBFM_ATAN_C5;
  q = __builtin_fmaf(q, s2, WBFM_ATAN_C4);
  q = __builtin_fmaf(q, s2, WBFM_ATAN_C3);
  q = __builtin_fmaf(q, s2, WBFM_ATAN_C2);
  q = __builtin_fmaf(q, s2, WBFM_ATAN_C1);
  q = __builtin_fmaf(q, s2, WBFM_ATAN_C0);
  float a = __builtin_fmaf(v, s2 * q, v);
  if (ay > ax) a = 0x1.921fb6p+0f - a;
  if (re < 0.0f) a = 0x1.921fb6p+1f - a;
  return __builtin_copysignf(a, im);
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
    w.dbuf[((size_t)stream * NB + b) * w.dcap + t] = wbfm_discriminate(c.x, c.y, pv.x, pv.y);
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

// =================================================================================================================
//  Fused WBFM kernel (P = 16*Q prototype taps, resampler history HD = ceil(Tg/L) <= HDMAX), no LDS tile, no barriers.
//
//  A group of 16 lanes owns a RUN of consecutive channelizer steps of one stream; lane k of the group computes polyphase
//  branch r = bitrev4(k).  Per step every lane
//    * loads ITS one new input sample (typed buffer load u8x2 -> 2 x f32; the 16 lanes of a group read 32 consecutive
//      bytes), subtracts 127.5 and pushes it into its private Q-deep sliding window (registers, static renaming) — every
//      input sample is converted exactly once, by exactly one lane;
//    * runs the Q-tap chain of its branch (v_pk_fma_f32 on (I,Q), oldest sample first);
//    * takes part in the 16-point DFT ACROSS the 16 lanes: the branches sit in bit-reversed lane order, stage m pairs
//      lane k with lane k^(m/2) (one cross-lane exchange), both lanes of a pair evaluate T = w*B with the same
//      instructions, the low lane keeps A+T, the high lane A-T — exactly the radix-2 DIT graph of the spec, so lane b ends
//      up with c_b[t] bit-identical to the oracle;
//    * (steps are processed in pairs) discriminates its band for two steps at once (packed K3), shifts its private
//      d window and, when an audio sample is due (wave-uniform test), evaluates the L/M resampler chain from registers.
//  Runs are warmed up by recomputing Q-1 + HD + 1 steps before their first output (from the old state when the run starts
//  at the call start); the last run of a stream hands the streaming state over.
// =================================================================================================================
typedef float wf2_t __attribute__((ext_vector_type(2)));
__device__ wf2_t wbfm_typed_load_xy(wi4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
// buffer resource word3: dst_sel = (R, G, 0, 1), num_format = USCALED (2), data_format = 8_8 (3)
#define SDRFM_RSRC_U8X2_USCALED (4 | (5 << 3) | (0 << 6) | (1 << 9) | (2 << 12) | (3 << 15))

namespace {

template <int HI>
__device__ __forceinline__ void wpk_fma_v(wf2_t& acc, wf2_t tap_pair, wf2_t x) {   // acc += tap * x, tap = half of a VGPR pair
  if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(tap_pair), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(tap_pair), "v"(x));
}

// packed K3 for two consecutive steps of one band (same roundings as wbfm_discriminate)
__device__ __forceinline__ wf2_t watan2_pair(wf2_t y, wf2_t x) {
  const wf2_t ax = __builtin_elementwise_abs(x), ay = __builtin_elementwise_abs(y);
  const wf2_t mx = __builtin_elementwise_max(__builtin_elementwise_max(ax, ay), wf2_t{0x1p-120f, 0x1p-120f}), mn = __builtin_elementwise_min(ax, ay);
  const wf2_t t = mn * wf2_t{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
  const wf2_t s = t * t;
  wf2_t q = wf2_t{WBFM_ATAN_C5, WBFM_ATAN_C5};
  q = __builtin_elementwise_fma(q, s, wf2_t{WBFM_ATAN_C4, WBFM_ATAN_C4});
  q = __builtin_elementwise_fma(q, s, wf2_t{WBFM_ATAN_C3, WBFM_ATAN_C3});
  q = __builtin_elementwise_fma(q, s, wf2_t{WBFM_ATAN_C2, WBFM_ATAN_C2});
  q = __builtin_elementwise_fma(q, s, wf2_t{WBFM_ATAN_C1, WBFM_ATAN_C1});
  q = __builtin_elementwise_fma(q, s, wf2_t{WBFM_ATAN_C0, WBFM_ATAN_C0});
  const wf2_t a = __builtin_elementwise_fma(t, s * q, t);
  float a0 = a.x, a1 = a.y;
  if (ay.x > ax.x) a0 = 0x1.921fb6p+0f - a0;
  if (ay.y > ax.y) a1 = 0x1.921fb6p+0f - a1;
  if (x.x < 0.0f) a0 = 0x1.921fb6p+1f - a0;
  if (x.y < 0.0f) a1 = 0x1.921fb6p+1f - a1;
  return wf2_t{__builtin_copysignf(a0, y.x), __builtin_copysignf(a1, y.y)};
}
__device__ __forceinline__ wf2_t wdisc_pair(wf2_t c0, wf2_t cm1, wf2_t c1) {   // (d[t], d[t+1]) from c[t-1], c[t], c[t+1]
  const wf2_t yr = {c0.x, c1.x}, yi = {c0.y, c1.y}, pr = {cm1.x, c0.x}, pi = {cm1.y, c0.y};
  const wf2_t re = __builtin_elementwise_fma(yr, pr, yi * pi);
  const wf2_t im = yi * pr - yr * pi;
  return watan2_pair(im, re);
}

// One stage of the 16-point DFT across the 16 lanes of a group (a DPP row), butterfly partner = lane ^ HALF.
// The spec's butterfly is (A, B) -> (A + W B, A - W B) with A in the low lane and B in the high lane.  Every lane first
// forms z = high ? W v : v (a select, not a multiplication by (1, 0): that could turn a -0 into +0, and the sign of a zero
// decides between +pi and -pi in K3), so z is W B in the high lane and A in the low one; then
//     out = z(partner) + (high ? -z : z)          (low: A + W B;  high: A - W B, since x - y == x + (-y) exactly)
// is one DPP add per component; the sign flip is an xor with a per-lane mask.  No LDS traffic.
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_partner(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, BANK, BANK == 0xF));   // full mask: bound_ctrl lets the compiler fold the move into the add
}
template <int HALF>
__device__ __forceinline__ wf2_t dft_partner(wf2_t z) {
  if constexpr (HALF == 1) return wf2_t{dpp_partner<0xB1, 0xF>(0.f, z.x), dpp_partner<0xB1, 0xF>(0.f, z.y)};        // quad_perm [1,0,3,2]
  else if constexpr (HALF == 2) return wf2_t{dpp_partner<0x4E, 0xF>(0.f, z.x), dpp_partner<0x4E, 0xF>(0.f, z.y)};   // quad_perm [2,3,0,1]
  else if constexpr (HALF == 8) return wf2_t{dpp_partner<0x128, 0xF>(0.f, z.x), dpp_partner<0x128, 0xF>(0.f, z.y)}; // row_ror:8
  else return wf2_t{dpp_partner<0x141, 0xF>(0.f, z.x), dpp_partner<0x141, 0xF>(0.f, z.y)};   // row_half_mirror: physical lane ^ 7 = logical index ^ 4

}
template <int HALF, bool UNIT>
__device__ __forceinline__ wf2_t dft_stage(wf2_t v, float ar, float ai, int sign) {
  wf2_t z = v;
  if constexpr (!UNIT) {                                       // (stage m = 2 has W = 1 in every lane)
    const float tr = __builtin_fmaf(ar, v.x, -(ai * v.y)), ti = __builtin_fmaf(ar, v.y, ai * v.x);
    z.x = sign ? tr : v.x;
    z.y = sign ? ti : v.y;
  }
  const wf2_t p = dft_partner<HALF>(z);
  const float zx = z.x, zy = z.y;                              // (scalars first: bit_cast of a vector-element lvalue reads element 0)
  const float sx = __int_as_float(__float_as_int(zx) ^ sign), sy = __int_as_float(__float_as_int(zy) ^ sign);
  return wf2_t{p.x + sx, p.y + sy};                        // scalar adds: v_add_f32_dpp (a packed add cannot take DPP)
}

template <int Q, int HDMAX>
__global__ void __launch_bounds__(256) k_wbfm_fused(WParams w) {
  __shared__ float gsh[512];                                  // resampler taps (broadcast reads)
  __shared__ float osh[4][16][64];                             // per wave: 16 audio samples per lane, [slot][lane], lane-private columns
  for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) gsh[i] = i < w.Tg ? w.g[i] : 0.0f;   // zero-padded: taps past Tg add +0
  __syncthreads();
  // logical index k of this lane within its group: DPP offers lane ^ 1, ^ 2, ^ 8 and the half-row mirror (lane ^ 7) but not
  // lane ^ 4, so lanes are numbered such that index ^ 4 is the mirror: k = lane with bits 0..2 flipped when bit 2 is set
  const int lane = (int)(threadIdx.x & 63), k = (lane & 11) ^ ((lane & 4) ? 7 : 0);
  const int r = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);   // bit-reversed: lane k computes branch r
  // 16-lane group = one run of one stream.  The 4 groups of a wave take the SAME run of 4 neighbouring streams: they share
  // ta, tb and the audio-output pattern (all streams of a handle are in phase), so a wave never diverges.
  // The wave index is made explicitly wave-uniform so that ta, tb, the step counter and all resampler bookkeeping live in
  // SGPRs and every branch on them is a scalar branch.
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const uint32_t quads = (w.n_streams + 3) / 4;
  const uint32_t run = wv / quads, stream_raw = (wv % quads) * 4 + (uint32_t)(lane >> 4);
  if (run >= w.tiles_per_stream) return;                       // (whole wave; the only barrier is behind us)
  const bool active = stream_raw < w.n_streams;                // a partial quad computes on its first stream and stores nothing
  const uint32_t stream = active ? stream_raw : (wv % quads) * 4;
  // run = steps [ta, tb) of the call; tiles_per_stream runs of w.NT steps each (the last one takes the remainder)
  const int ta = (int)(run * w.NT);
  const int tb = (run + 1 == w.tiles_per_stream) ? (int)w.Tn : ta + (int)w.NT;
  const int HD = (int)w.HD;
  // taps of this lane's branch: p[r + 16 q], q = 0..Q-1, as Q/2 pairs (q even in .x)
  wf2_t tp[Q / 2];
#pragma unroll
  for (int q = 0; q < Q / 2; ++q) tp[q] = wf2_t{w.p[r + 16 * (2 * q)], w.p[r + 16 * (2 * q + 1)]};
  // DFT constants of this lane: stage m = 2,4,8,16 -> partner k ^ (m/2), twiddle W16[(k mod m/2) * 16/m]
  // high lanes (bit m/2 of k set) apply the twiddle and the minus sign
  const int sg1 = (k & 1) ? (int)0x80000000 : 0, sg2 = (k & 2) ? (int)0x80000000 : 0, sg4 = (k & 4) ? (int)0x80000000 : 0,
            sg8 = (k & 8) ? (int)0x80000000 : 0;
  const float w2r = W16_RE[(k & 1) * 4], w2i = W16_IM[(k & 1) * 4];
  const float w4r = W16_RE[(k & 3) * 2], w4i = W16_IM[(k & 3) * 2];
  const float w8r = W16_RE[k & 7], w8i = W16_IM[k & 7];

  // one wave-uniform descriptor over the whole batch (a per-stream descriptor would differ between the 4 groups of a wave
  // and cost a waterfall loop per load); the host only selects this kernel when the batch spans < 4 GiB
  const unsigned long long ga = (unsigned long long)w.iq;
  const wi4_t rsrc = {(int)(unsigned)ga, (int)(unsigned)(ga >> 32), (int)w.iq_span, SDRFM_RSRC_U8X2_USCALED};
  const uint32_t sbase = stream * (uint32_t)w.iq_stride + 2u * (uint32_t)(15 - (int)w.phase_x - r + 16);   // byte offset of step 0's sample, +32
  auto sample_fast = [&](int s) -> wf2_t {                     // steps whose sample lies in this call's bytes (n >= 0);
    const wf2_t c = wbfm_typed_load_xy(rsrc, (int)(sbase + 32u * (uint32_t)s - 32u), 0, 0);   // past the batch end -> 0, unused
    return c - wf2_t{127.5f, 127.5f};
  };
  auto sample = [&](int s) -> wf2_t {                          // this lane's input of step s: x[16 s + 15 - r] (call-relative)
    const int n = (s + 1) * 16 - 1 - (int)w.phase_x - r;
    if (n < 0) { const float2 h = wload_x(w, stream, n); return wf2_t{h.x, h.y}; }
    return sample_fast(s);
  };

  // ---- warm-up ---------------------------------------------------------------------------------------------------
  wf2_t win[Q];                                                // win[0] = oldest
  wf2_t cprev;
  float dring[HDMAX + 1];                                      // after a pair (s, s+1): dring[HDMAX] = d[s+1], dring[HDMAX-1] = d[s], ...
#pragma unroll
  for (int i = 0; i <= HDMAX; ++i) dring[i] = 0.0f;
  int t0;                                                      // first step whose c is computed
  if (ta == 0) {
    t0 = 0;
    const float2 cp = w.cprev_in[(size_t)stream * NB + k];
    cprev = wf2_t{cp.x, cp.y};
    for (int i = 0; i < HD; ++i) {
      const float v = w.hist_d_in[((size_t)stream * NB + k) * HD + i];
#pragma unroll
      for (int z = 0; z <= HDMAX; ++z) if (z == i + (HDMAX + 1 - HD)) dring[z] = v;   // right-aligned: d[-1] at dring[HDMAX]
    }
  } else {
    t0 = ta - (int)((w.HD + 2) & ~1u);                         // c from t0, d from t0+1: >= HD valid d's before step ta; an even
    cprev = wf2_t{0.f, 0.f};                                   // lead keeps every run's pairs on even steps
  }
#pragma unroll
  for (int q = 0; q < Q - 1; ++q) win[q + 1] = sample(t0 - (Q - 1) + q);          // win[1..Q-1] = steps t0-Q+1 .. t0-1

  // resampler bookkeeping (uniform per group; identical in all 16 lanes): next output and the step of its newest d
  const unsigned long long gd0 = w.n_d;                        // global d index of call-relative step 0
  unsigned long long jn0 = w.n_a;                              // next audio index (global)
  {
    // first output whose newest d lies at or after step ta: j = ceil((gd0 + ta) * L / M) but not before n_a
    const unsigned long long need = ((gd0 + (unsigned long long)ta) * w.L + w.M - 1) / w.M;
    if (need > jn0) jn0 = need;
  }
  int nj_rel = (int)((long long)((jn0 * w.M) / w.L) - (long long)gd0);             // call-relative step of its newest d
  uint32_t phi = (uint32_t)((jn0 * w.M) % w.L);
  uint32_t jrel = (uint32_t)(jn0 - w.n_a);                     // its index in this call's output
  const uint32_t mq = w.M / w.L, mr = w.M % w.L;
  float* const aout = w.audio + ((size_t)stream * NB + k) * w.band_stride;

  wf2_t pf[Q];                                                 // samples of the next Q steps, loaded one block ahead
  auto one_step = [&](int s, auto QI) -> wf2_t {               // returns c_b[s] of this lane's band; QI = static window phase
    constexpr int qi = decltype(QI)::value;                    // the new sample goes to slot qi; oldest is slot (qi+1) % Q
    (void)s;
    win[qi] = pf[qi];
    wf2_t acc = {0.f, 0.f};
#pragma unroll
    for (int q = Q - 1; q >= 0; --q) {                         // oldest first: q = Q-1 is the oldest sample, tap p[r+16q]
      const wf2_t x = win[(qi + Q - q) % Q];
      if (q & 1) wpk_fma_v<1>(acc, tp[q / 2], x); else wpk_fma_v<0>(acc, tp[q / 2], x);
    }
    wf2_t v = dft_stage<1, true>(acc, 1.0f, 0.0f, sg1);       // stage m = 2
    v = dft_stage<2, false>(v, w2r, w2i, sg2);                // stage m = 4: twiddle W16[(k mod 2) * 4]
    v = dft_stage<4, false>(v, w4r, w4i, sg4);                // stage m = 8: twiddle W16[(k mod 4) * 2]
    v = dft_stage<8, false>(v, w8r, w8i, sg8);                // stage m = 16: twiddle W16[k mod 8]
    return v;
  };

  typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
  float* const ocol = &osh[(threadIdx.x >> 6) & 3][0][lane];
  uint32_t nbuf = 0;                                           // samples waiting in the column BEFORE the current slot (uniform)
  auto flush = [&](uint32_t before, uint32_t last_slot, uint32_t j_last) {   // slots last_slot-before .. last_slot -> aout[.. j_last]
    if (!active) return;
    float* const row = aout + (j_last - last_slot);            // address of slot 0
    if (before == 15u && last_slot == 15u) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f4u_t*>(row + 4 * q) = f4u_t{ocol[(4 * q) * 64], ocol[(4 * q + 1) * 64], ocol[(4 * q + 2) * 64], ocol[(4 * q + 3) * 64]};
    } else {
      for (uint32_t i = last_slot - before; i <= last_slot; ++i) row[i] = ocol[i * 64];
    }
  };
  auto emit = [&](auto OFF) {                                  // one audio sample; OFF = 1: its newest d is the pair's first step
    constexpr int off = decltype(OFF)::value;
    float gt[HDMAX];                                           // taps g[phi + L i]; indices past Tg read the zero padding, and
#pragma unroll                                                 // 0 * d + a == a, so the chain equals the spec's i <= imax chain
    for (int i = 0; i < HDMAX; ++i) gt[i] = gsh[phi + w.L * (uint32_t)i];
    float a = 0.0f;
#pragma unroll
    for (int i = HDMAX - 1; i >= 0; --i) a = __builtin_fmaf(gt[i], dring[HDMAX - off - i], a);
    // Audio leaves through a lane-private LDS column and is written 16 samples (64 bytes of one band row) at a time: as
    // single 4-byte stores the 64 rows of each of ~3000 waves kept ~25 MB of half-filled lines open, more than the L2s
    // hold, and rocprof showed 7.6x the algorithmic write traffic.
    const uint32_t slot = jrel & 15u;
    ocol[slot * 64] = a;
    if (slot == 15u) { flush(nbuf, 15u, jrel); nbuf = 0; } else ++nbuf;
  };
  auto push_pair_and_emit = [&](wf2_t d, int s) {              // d = (d[s], d[s+1])
#pragma unroll
    for (int z = 0; z + 2 <= HDMAX; ++z) dring[z] = dring[z + 2];
    dring[HDMAX - 1] = d.x;
    dring[HDMAX] = d.y;
    while (nj_rel <= s + 1 && nj_rel < tb) {         // audio samples whose newest d is one of these two steps
      if (nj_rel == s) emit(std::integral_constant<int, 1>{}); else emit(std::integral_constant<int, 0>{});
      ++jrel;                                                  // next output: position advances by M = mq*L + mr
      phi += mr;
      nj_rel += (int)mq;
      if (phi >= w.L) { phi -= w.L; ++nj_rel; }
    }
  };

  // ---- main loop: pairs of steps, window phase unrolled by Q ---------------------------------------------------------
  int s = t0;
  wf2_t cprev0 = cprev;                                        // c of the last pair's first step (odd tail)
  // window slot of step s is (s - t0) % Q when slot bookkeeping starts at 0 for step t0: win[1..Q-1] hold the Q-1 older
  // samples in age order, so step t0 writes slot 0 and the oldest is slot 1.
  wf2_t pn[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) pn[q] = sample(t0 + q);
  while (s < tb) {
#pragma unroll
    for (int q = 0; q < Q; ++q) pf[q] = pn[q];
#pragma unroll
    for (int q = 0; q < Q; ++q) pn[q] = sample_fast(s + Q + q);   // in flight during this block (s + Q >= 8: n >= 0)
    static_for_w<0, Q / 2>([&](auto PI) {
      constexpr int pi2 = decltype(PI)::value;
      if (s < tb) {
        const wf2_t c0 = one_step(s, std::integral_constant<int, (2 * pi2) % Q>{});
        const wf2_t c1 = one_step(s + 1, std::integral_constant<int, (2 * pi2 + 1) % Q>{});   // may run one step past tb: unused
        const wf2_t d2 = wdisc_pair(c0, cprev, c1);
        push_pair_and_emit(d2, s);
        cprev0 = c0; cprev = c1;
        s += 2;
      }
    });
  }
  if (nbuf) flush(nbuf - 1, (jrel - 1) & 15u, jrel - 1);       // the run's last, partial group of audio samples
  // ---- state hand-over by the last run of the stream -------------------------------------------------------------------
  if (run + 1 == w.tiles_per_stream && active) {
    if (s > tb) {                                              // the last pair ran one step past the end: drop that step
      cprev = cprev0;
#pragma unroll
      for (int z = HDMAX; z >= 1; --z) dring[z] = dring[z - 1];
    }
    w.cprev_out[(size_t)stream * NB + k] = make_float2(cprev.x, cprev.y);
#pragma unroll
    for (int z = 0; z <= HDMAX; ++z)
      if (z >= HDMAX + 1 - HD) w.hist_d_out[((size_t)stream * NB + k) * HD + (z - (HDMAX + 1 - HD))] = dring[z];
    for (uint32_t i = (uint32_t)k; i + 1 < w.P; i += 16)
      w.hist_x_out[(size_t)stream * (w.P - 1) + i] = wload_x(w, stream, (int)w.N - (int)(w.P - 1) + (int)i);
  }
}

}  // namespace


// =================================================================================================================
//  WBFM step kernel: ONE LANE = ONE CHANNELIZER STEP (P = 128, HD <= HDMAX).
//
//  A wave owns a run of consecutive steps of one stream and walks it in blocks of 64 steps, lane l = step s0 + l.  Per block
//    1. polyphase FIR: for q = Q-1 .. 0 (oldest first, the spec's order) the lane reads the group of step s - q from an LDS tile
//       (8 x 16 bytes, lane stride 144 bytes: conflict-free; three groups in flight) and feeds all 16 branches, two per
//       v_pk_fma_f32: accumulator pair k' holds the branches at DFT positions (2k', 2k'+1), the tap pair is an SGPR pair
//       (s_buffer_load_dwordx16 of taps stored pre-permuted, half of the 128 taps at a time).  The tile is a ring of 80 step
//       groups, each two 16-float planes (re | im) in DFT INPUT ORDER (plane position k = branch bitrev4(k));
//    2. right after the FIR the NEXT block's 16 samples per lane (32 bytes, fetched two blocks ahead) are converted and stored
//       into the ring slots this block's FIR was the last to read: every input byte is fetched and converted once (the kernel
//       with one lane per branch re-computed ~20 warm-up steps per run);
//    3. the 16-point DFT runs in the lane's registers on (position 2k', 2k'+1) pairs: stage m = 2 is an add/sub inside each
//       pair (its twiddle is (1, 0) and its inputs are FIR outputs, which are never -0: multiplying by (1, 0) is then the
//       identity, bit for bit), stages 4, 8, 16 are the spec's butterflies on whole pairs with packed twiddle constants —
//       no cross-lane traffic, no selects;  band pairs (2i, 2i+1) come out as (re, re) / (im, im) register pairs, exactly
//       the packing the two-at-a-time discriminator wants;
//    4. c[s-1] comes from the lane below by DPP (wave_shr:1; lane 0 takes what lane 63 left in LDS a block earlier), K3 runs
//       packed on band pairs, stage by stage over the 8 pairs;
//    5. the 16 d's of the step go to per-band rows in LDS (a ring of 96 columns, the last HDMAX mirrored below column 0); the
//       audio samples of a block (about 64 L / M per band) are evaluated ONE BLOCK LATER by lane (band, 4 consecutive samples)
//       from those rows and a [phase][tap] table: their LDS reads are issued before the DFT of the next block and consumed
//       after it.
//  A run that does not start the call warms up HD + 1 steps (the FIR has no memory beyond its window); the last run of a
//  stream hands the streaming state over.  The two waves of a SIMD swap priority roles once per run (see the kernel).
// =================================================================================================================
// scalar (SGPR) load of 16 consecutive floats through a buffer descriptor: wave-uniform taps without spending VGPRs or LDS cycles
typedef float wf16_t __attribute__((ext_vector_type(16)));
__device__ wf16_t wbfm_s_buffer_load16(wi4_t rsrc, int offset, int aux) __asm("llvm.amdgcn.s.buffer.load.v16f32");

namespace {

typedef float wf4_t __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) wraw16 { unsigned int w[4]; };

// K3 for all 16 bands of a step, two bands per packed operation, written stage by stage over the 8 band pairs so that the eight
// dependent chains are interleaved in program order (same roundings as wbfm_discriminate / watan2_pair)
__device__ __forceinline__ void wdisc_bands8(const wf2_t (&yr)[8], const wf2_t (&yi)[8], const wf2_t (&pr)[8], const wf2_t (&pi)[8], wf2_t (&d)[8]) {
  wf2_t re[8], im[8], t[8], s[8], q[8], ax[8], ay[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { re[k] = __builtin_elementwise_fma(yr[k], pr[k], yi[k] * pi[k]); im[k] = yi[k] * pr[k] - yr[k] * pi[k]; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {                                 // atan2(y = im, x = re)
    ax[k] = __builtin_elementwise_abs(re[k]); ay[k] = __builtin_elementwise_abs(im[k]);
    const wf2_t mx = __builtin_elementwise_max(__builtin_elementwise_max(ax[k], ay[k]), wf2_t{0x1p-120f, 0x1p-120f}), mn = __builtin_elementwise_min(ax[k], ay[k]);
    t[k] = mn * wf2_t{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
    s[k] = t[k] * t[k];
    q[k] = wf2_t{WBFM_ATAN_C5, WBFM_ATAN_C5};
  }
  constexpr float cf[5] = {WBFM_ATAN_C4, WBFM_ATAN_C3, WBFM_ATAN_C2, WBFM_ATAN_C1, WBFM_ATAN_C0};
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = __builtin_elementwise_fma(q[k], s[k], wf2_t{cf[c], cf[c]});
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const wf2_t a = __builtin_elementwise_fma(t[k], s[k] * q[k], t[k]);
    float a0 = a.x, a1 = a.y;
    if (ay[k].x > ax[k].x) a0 = 0x1.921fb6p+0f - a0;
    if (ay[k].y > ax[k].y) a1 = 0x1.921fb6p+0f - a1;
    if (re[k].x < 0.0f) a0 = 0x1.921fb6p+1f - a0;
    if (re[k].y < 0.0f) a1 = 0x1.921fb6p+1f - a1;
    d[k] = wf2_t{__builtin_copysignf(a0, im[k].x), __builtin_copysignf(a1, im[k].y)};
  }
}

// value of the lane below (wave_shr:1); lane 0 keeps `keep`
__device__ __forceinline__ float wshr1(float keep, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}

// keeps both the IR passes (memory clobber) and the machine scheduler (sched_barrier) from moving code across: without it
// every LDS read of a block is hoisted to the top and the kernel spills
__device__ __forceinline__ void wfence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// the spec's butterfly on two element pairs at once: (A, B) -> (A + W B, A - W B), W = (wr, wi) per half
__device__ __forceinline__ void wbfly(wf2_t& ar, wf2_t& ai, wf2_t& br, wf2_t& bi, wf2_t wr, wf2_t wi) {
  const wf2_t tr = __builtin_elementwise_fma(wr, br, -(wi * bi));
  const wf2_t ti = __builtin_elementwise_fma(wr, bi, wi * br);
  br = ar - tr; bi = ai - ti;
  ar = ar + tr; ai = ai + ti;
}

// LDS geometry of the step kernel (one definition for the kernel and for the launch)
template <int Q, int HDMAX>
struct WStepsLds {
  static constexpr int GS = 144;                                // bytes per step group in the tile: 16 re + 16 im floats + 16 pad
  static constexpr int RING = 80;                               // tile slots: a ring over the step index (>= 64 + Q - 1; a multiple of 16
                                                                // keeps a lane group's 16 slots on 16 different bank quads across the wrap)
  static constexpr int DRING = 96;                              // d columns per band: a ring over the step index (>= 64 + HDMAX)
  static constexpr int DROW = HDMAX + DRING + 2;                // words per band row: [mirror of the last HDMAX columns | DRING columns | pad]
  static constexpr int GT = (HDMAX + 3) & ~3;                   // words per phase row of the tap table
  static_assert(RING >= 64 + Q - 1 && DRING >= 64 + HDMAX, "rings");
  static constexpr size_t bytes(uint32_t L) { return (size_t)RING * GS + GS + (size_t)16 * DROW * 4 + (size_t)L * GT * 4; }   // tile, c slot, d rows, tap table
};

template <int Q, int HDMAX>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_wbfm_steps(WParams w) {
  using G_ = WStepsLds<Q, HDMAX>;
  constexpr int GS = G_::GS, RING = G_::RING, DRING = G_::DRING, DROW = G_::DROW, GT = G_::GT;
  constexpr int rev[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const xt = smem;                               // RING step groups
  float* const cslot = reinterpret_cast<float*>(xt + RING * GS);   // c of the step before the block: 16 re | 16 im (| pad)
  float* const drow = cslot + GS / 4;
  float* const gT = drow + 16 * DROW;                           // [L][GT]: gT[phi][i] = g[phi + L i], zero beyond Tg / HDMAX
  const int lane = (int)threadIdx.x;
#ifdef SDRFM_DEV   // development build: per-wave time stamps (shader cycles) at the phase boundaries of every block.  A stamp waits for
  // everything outstanding first, so a phase's figure includes its own memory latency (and the kernel runs slower: diagnostic only)
  unsigned long long* const tsp = w.dbg ? w.dbg + 256 * (size_t)blockIdx.x : nullptr;
  int tsi = 0;
#define WSTAMP() do { if (tsp && !w.dbg_light) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lane == 0 && tsi < 250) tsp[tsi] = __builtin_readcyclecounter(); ++tsi; asm volatile("" ::: "memory"); } } while (0)
  if (tsp && lane == 0) { tsp[254] = __builtin_amdgcn_s_memrealtime(); tsp[253] = __builtin_amdgcn_s_getreg((31 << 11) | 4); }
#else
#define WSTAMP() do { } while (0)
#endif
  WSTAMP();                                                     // 0: entry
  const uint32_t run = blockIdx.x / w.n_streams, stream = blockIdx.x % w.n_streams;
  const int HD = (int)w.HD, L = (int)w.L;
  const int ta = (int)(run * w.NT);
  const bool last_run = run + 1 == w.tiles_per_stream;
  const int tb = last_run ? (int)w.Tn : ta + (int)w.NT;
  const int t0 = run == 0 ? 0 : ta - (HD + 1);                  // first step computed: c from t0, d from t0 + 1

  for (int i = lane; i < L * GT; i += 64) {
    const int phi = i / GT, k = i % GT;
    gT[i] = (k < HDMAX && (uint32_t)(phi + L * k) < w.Tg) ? w.g[phi + L * k] : 0.0f;
  }
  for (int i = lane; i < 16 * HDMAX; i += 64) drow[(i / HDMAX) * DROW + i % HDMAX] = 0.0f;   // columns -HDMAX .. -1: the d's before the run (zero taps
                                                                                            // may meet them: no NaN bit patterns)
  if (lane < 32) cslot[lane] = 0.0f;                            // c[t0 - 1]: only d[t0] depends on it, which no audio sample reads
  __syncthreads();
  if (run == 0) {                                               // the call's carried state: d history and c[-1]
    for (int i = lane; i < 16 * HD; i += 64) {
      const int b = i / HD, k = i % HD;
      drow[b * DROW + HDMAX - HD + k] = w.hist_d_in[((size_t)stream * NB + b) * HD + k];
    }
    if (lane < NB) {
      const float2 cp = w.cprev_in[(size_t)stream * NB + lane];
      cslot[lane] = cp.x;
      cslot[16 + lane] = cp.y;
    }
  }

  const uint8_t* const row = w.iq + (size_t)stream * w.iq_stride;
  const unsigned long long pa = (unsigned long long)w.pperm;
  const wi4_t prsrc = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)(4u * w.P), 0x00020000};
  // ---- input of one step group: fast (all 16 samples inside this call's bytes), zero (beyond them: such steps do not exist),
  //      or slow (reaches before the call start: carried history, sample by sample)
  auto group_kind = [&](int step) -> int {
    const int n_lo = 16 * step - (int)w.phase_x;
    if (n_lo >= 0 && n_lo + 16 <= (int)w.N) return 0;
    return n_lo < 0 ? 2 : 1;
  };
  auto fetch = [&](int step, wraw16& a, wraw16& b) {
    const wraw16* gp = reinterpret_cast<const wraw16*>(row + 2 * (size_t)(16 * step - (int)w.phase_x));
    a = gp[0]; b = gp[1];
  };
  auto store_group = [&](int off, const wraw16& a, const wraw16& b) {   // convert + store in DFT input order (off = slot * GS)
    float re[16], im[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int j = 15 - rev[k];                                // branch r = rev[k] reads sample 15 - r of the group
      const unsigned int d = (j >> 1) < 4 ? a.w[j >> 1] : b.w[(j >> 1) - 4];
      re[k] = (float)((d >> (16 * (j & 1))) & 0xffu) - 127.5f;
      im[k] = (float)((d >> (16 * (j & 1) + 8)) & 0xffu) - 127.5f;
    }
    wf4_t* dst = reinterpret_cast<wf4_t*>(xt + off);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      dst[m] = wf4_t{re[4 * m], re[4 * m + 1], re[4 * m + 2], re[4 * m + 3]};
      dst[4 + m] = wf4_t{im[4 * m], im[4 * m + 1], im[4 * m + 2], im[4 * m + 3]};
    }
  };
  auto store_group_slow = [&](int off, int step, int kind) {
    float* dst = reinterpret_cast<float*>(xt + off);
    for (int k = 0; k < 16; ++k) {
      float2 x = make_float2(0.f, 0.f);
      if (kind == 2) {
        const int r = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);
        const int n = 16 * step - (int)w.phase_x + 15 - r;
        if (n < (int)w.N) x = wload_x(w, stream, n);
      }
      dst[k] = x.x; dst[16 + k] = x.y;
    }
  };

  wraw16 ra, rb;                                                // raw bytes of this lane's step of the CURRENT block
  int kind = group_kind(t0 + lane);
  if (kind == 0) fetch(t0 + lane, ra, rb);
  // the first block's groups and the Q - 1 groups before it -> tile; prefetch the second block
  if (kind == 0) store_group((Q - 1 + lane) * GS, ra, rb);
  else store_group_slow((Q - 1 + lane) * GS, t0 + lane, kind);
  kind = group_kind(t0 + 64 + lane);
  if (kind == 0 && t0 + 64 < tb) fetch(t0 + 64 + lane, ra, rb);
  if (lane < Q - 1) {
    const int st = t0 - (Q - 1) + lane, kd = group_kind(st);
    if (kd == 0) { wraw16 ha, hb; fetch(st, ha, hb); store_group(lane * GS, ha, hb); }
    else store_group_slow(lane * GS, st, kd);
  }
  // resampler bookkeeping: call-relative audio index jl has its newest d at call-relative step res_q0 + floor((res_r0 + jl M) / L)
  auto first_jl = [&](int x) -> int {                           // first jl whose newest d lies at or after step x
    const int num = (x - w.res_q0) * L - (int)w.res_r0;         // (the host admits this kernel only while these fit 32 bits)
    if (num <= 0) return 0;
    const uint32_t jl = ((uint32_t)num + w.M - 1) / w.M;
    return jl > w.A ? (int)w.A : (int)jl;
  };
  int jl_lo = first_jl(ta);

  wf2_t cr[8], ci[8];
  int s0_last = t0;
  // The audio samples of a block are evaluated one block later: their LDS reads (taps by phase, d windows) are issued before the
  // next block's DFT and consumed after it, so their latency hides behind arithmetic instead of standing between a write and a
  // read of the same rows.  Lane (band b = lane / 4, quarter o = lane % 4) evaluates up to 4 consecutive samples.
  int dcb = 0;                                                  // d ring column of lane 0's step in the current block
  int p_s0 = 0, p_dcb = 0, p_lo = 0, p_hi = 0;                  // previous block: first step, its column, audio range [p_lo, p_hi)
  float rgt[4][GT], rdv[4][HDMAX];
  bool rok[4];
  float* const aout = w.audio + ((size_t)stream * NB + (lane >> 2)) * w.band_stride;
  auto resample_issue = [&]() {
    const float* const myrow = drow + (lane >> 2) * DROW + HDMAX;      // myrow[c] = column c, c in [-HDMAX, DRING)
    // (the lane's four samples are consecutive: one division for the first — multiplications by 2^32 / L run at a quarter of the rate —,
    // then loc grows by M: the quotient by M div L, the phase by M mod L, with a carry)
    const uint32_t Mq = w.M / (uint32_t)L, Mr = w.M - Mq * (uint32_t)L;      // (wave-uniform: scalar code)
    const uint32_t loc0 = w.res_r0 + (uint32_t)(p_lo + 4 * (lane & 3)) * w.M;
    uint32_t qq = __umulhi(loc0, w.inv_L32), phi = loc0 - qq * (uint32_t)L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int jl = p_lo + 4 * (lane & 3) + i;
      if (i > 0) {
        phi += Mr; qq += Mq;
        if (phi >= (uint32_t)L) { phi -= (uint32_t)L; qq += 1u; }
      }
      const int nrel = w.res_q0 + (int)qq;                        // call-relative step of the sample's newest d
      rok[i] = jl < p_hi;
      int c = p_dcb + (nrel - p_s0);
      if (c >= DRING) c -= DRING;
      const float* dp = myrow + (rok[i] ? c : 0);                 // (inactive lanes read inside the row)
      const wf4_t* tp = reinterpret_cast<const wf4_t*>(gT + (rok[i] ? phi : 0u) * GT);
#pragma unroll
      for (int m = 0; m < GT / 4; ++m) { const wf4_t t4 = tp[m]; rgt[i][4 * m] = t4.x; rgt[i][4 * m + 1] = t4.y; rgt[i][4 * m + 2] = t4.z; rgt[i][4 * m + 3] = t4.w; }
#pragma unroll
      for (int k = 0; k < HDMAX; ++k) rdv[i][k] = dp[-k];
    }
  };
  auto resample_finish = [&]() {
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = HDMAX - 1; k >= 0; --k)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = __builtin_fmaf(rgt[i][k], rdv[i][k], a[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (rok[i]) aout[p_lo + 4 * (lane & 3) + i] = a[i];
  };
  int ring0 = Q - 1;                                            // ring slot of lane 0's step (the Q - 1 older groups sit below it)
  // The two waves of a SIMD are arbitrated oldest-first: at equal priority the wave in hardware slot 0 takes every issue slot it can
  // use and the other gets what is left; when the first is done the second finishes alone (measured: ends at 81 and 108 us).
  // Strict turns (priority passed at every phase boundary) end the pair together but lower its joint throughput (105 us again):
  // one wave running unimpeded plus one filling its gaps is the more efficient mode.  So the roles are swapped once, in the middle
  // of the run: the slot-0 wave has the priority for the first 40 % of its steps, the slot-1 wave for the rest (both then end
  // within 2 us of each other: 105 -> 100 us).
  const uint32_t myslot = __builtin_amdgcn_s_getreg((31 << 11) | 4) & 1u;   // HW_ID.wave_id & 1
  const int swap_at = t0 + (int)(((long long)(tb - t0) * (int)w.swap_pct) / 100);
  WSTAMP();                                                     // 1: prologue done; then 7 stamps per block
  for (int s0 = t0; s0 < tb; s0 += 64) {
    s0_last = s0;
    const int s = s0 + lane;
    int tap0 = 0;                                               // (opaque zero: keeps the tap loads inside the block loop)
    asm volatile("" : "+s"(tap0));
    wf16_t th[4];                                               // taps of q = Q-1 .. Q-4 (SGPRs); the other half follows when these are spent
#pragma unroll
    for (int j = 0; j < 4; ++j) th[j] = wbfm_s_buffer_load16(prsrc, tap0 + 64 * (Q - 1 - j), 0);
    int off0 = (ring0 + lane) * GS;                             // byte offset of this lane's group in the ring (stored by the previous
    if (off0 >= RING * GS) off0 -= RING * GS;                   // iteration / the prologue)
    wfence();

    WSTAMP();                                                   // block top (tap loads issued)
    if ((s0 < swap_at) == (myslot == 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    // ---- 2. polyphase FIR, oldest tap first
    wf2_t ar[8], ai[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { ar[k] = wf2_t{0.f, 0.f}; ai[k] = wf2_t{0.f, 0.f}; }
    // software-pipelined: DEPTH groups are in flight while one is accumulated (the LDS serves eight waves, a read waits in its
    // queue for hundreds of cycles; LDS returns in order, so the waits are exact); the fence after each q keeps the scheduler
    // from issuing all 64 reads at once (256 VGPRs)
    auto group_of = [&](int q) -> const wf4_t* {                // the group of step s - q
      int o = off0 - q * GS;
      if (o < 0) o += RING * GS;
      return reinterpret_cast<const wf4_t*>(xt + o);
    };
    constexpr int DEPTH = 3;                                    // groups in flight ahead of the one being accumulated
    wf4_t xr[DEPTH + 1][4], xi[DEPTH + 1][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const wf4_t* gp = group_of(Q - 1 - d);
#ifdef SDRFM_DEV
      if ((w.ablate & 1u) && d > 0) {                           // (timing experiment: groups Q-2, Q-3 re-use group Q-1's registers)
#pragma unroll
        for (int m = 0; m < 4; ++m) { xr[d][m] = xr[0][m]; xi[d][m] = xi[0][m]; }
        continue;
      }
#endif
#pragma unroll
      for (int m = 0; m < 4; ++m) { xr[d][m] = gp[m]; xi[d][m] = gp[4 + m]; }
    }
    wf16_t tl[4];
#pragma unroll
    for (int q = Q - 1; q >= 0; --q) {
      if (q - DEPTH >= 0) {
        const wf4_t* gp = group_of(q - DEPTH);
#ifdef SDRFM_DEV
        if ((w.ablate & 1u) && (q - DEPTH) % 3 != 1) {          // (timing experiment: only groups 4 and 1 are read here)
#pragma unroll
          for (int m = 0; m < 4; ++m) { xr[DEPTH][m] = xr[DEPTH - 1][m]; xi[DEPTH][m] = xi[DEPTH - 1][m]; }
        } else
#endif
        {
#pragma unroll
          for (int m = 0; m < 4; ++m) { xr[DEPTH][m] = gp[m]; xi[DEPTH][m] = gp[4 + m]; }
        }
      }
      const wf16_t tq = q >= 4 ? th[Q - 1 - q] : tl[3 - q];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const wf2_t t0p = wf2_t{tq[4 * m], tq[4 * m + 1]}, t1p = wf2_t{tq[4 * m + 2], tq[4 * m + 3]};
        ar[2 * m] = __builtin_elementwise_fma(t0p, wf2_t{xr[0][m].x, xr[0][m].y}, ar[2 * m]);
        ar[2 * m + 1] = __builtin_elementwise_fma(t1p, wf2_t{xr[0][m].z, xr[0][m].w}, ar[2 * m + 1]);
        ai[2 * m] = __builtin_elementwise_fma(t0p, wf2_t{xi[0][m].x, xi[0][m].y}, ai[2 * m]);
        ai[2 * m + 1] = __builtin_elementwise_fma(t1p, wf2_t{xi[0][m].z, xi[0][m].w}, ai[2 * m + 1]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { asm volatile("" : "+v"(ar[k])); asm volatile("" : "+v"(ai[k])); }   // pins this q's FMAs before the fence
      wfence();
      if (q == 4) {                                             // first half of the taps spent: fetch the second half into the same SGPRs
        int tap1 = 0;
        asm volatile("" : "+s"(tap1));
#pragma unroll
        for (int j = 0; j < 4; ++j) tl[j] = wbfm_s_buffer_load16(prsrc, tap1 + 64 * (3 - j), 0);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int m = 0; m < 4; ++m) { xr[d][m] = xr[d + 1][m]; xi[d][m] = xi[d + 1][m]; }
    }
    WSTAMP();                                                   // FIR
    wfence();
    // ---- 1'. the NEXT block's samples -> tile (their slots hold steps that this block's FIR was the last to read), then the
    //          prefetch of the block after it.  Placed here, the wait for the prefetched bytes does not also wait for the audio
    //          stores of the resampler below (stores count in vmcnt too): those are a whole FIR old when the next wait comes.
    if (s0 + 64 < tb) {
      int offn = off0 + 64 * GS;
      if (offn >= RING * GS) offn -= RING * GS;
      if (kind == 0) store_group(offn, ra, rb);
      else store_group_slow(offn, s + 64, kind);
      wfence();
      kind = group_kind(s + 128);
      if (kind == 0 && s0 + 128 < tb) fetch(s + 128, ra, rb);
    }
    WSTAMP();                                                   // next tile stored, prefetch issued (and, stamped, waited for)
    wfence();
    resample_issue();                                           // previous block's audio (none before the first: empty range): reads in flight during the DFT
    // ---- 3. 16-point DFT on position pairs (2k', 2k'+1)
#pragma unroll
    for (int k = 0; k < 8; ++k) {                               // stage m = 2: partner inside the pair, W = (1, 0)
      ar[k] = wf2_t{ar[k].x + ar[k].y, ar[k].x - ar[k].y};
      ai[k] = wf2_t{ai[k].x + ai[k].y, ai[k].x - ai[k].y};
    }
#pragma unroll
    for (int g = 0; g < 8; g += 2)                              // stage m = 4: positions (g0, g0+1) with (g0+2, g0+3), W = (W0, W4)
      wbfly(ar[g], ai[g], ar[g + 1], ai[g + 1], wf2_t{1.0f, 0.0f}, wf2_t{0.0f, 1.0f});
#pragma unroll
    for (int g = 0; g < 8; g += 4) {                            // stage m = 8: W = (W0, W2), (W4, W6)
      wbfly(ar[g], ai[g], ar[g + 2], ai[g + 2], wf2_t{1.0f, 0x1.6a09e6p-1f}, wf2_t{0.0f, 0x1.6a09e6p-1f});
      wbfly(ar[g + 1], ai[g + 1], ar[g + 3], ai[g + 3], wf2_t{0.0f, -0x1.6a09e6p-1f}, wf2_t{1.0f, 0x1.6a09e6p-1f});
    }
    // stage m = 16: W = (W0, W1), (W2, W3), (W4, W5), (W6, W7)
    wbfly(ar[0], ai[0], ar[4], ai[4], wf2_t{1.0f, 0x1.d906bcp-1f}, wf2_t{0.0f, 0x1.87de2ap-2f});
    wbfly(ar[1], ai[1], ar[5], ai[5], wf2_t{0x1.6a09e6p-1f, 0x1.87de2ap-2f}, wf2_t{0x1.6a09e6p-1f, 0x1.d906bcp-1f});
    wbfly(ar[2], ai[2], ar[6], ai[6], wf2_t{0.0f, -0x1.87de2ap-2f}, wf2_t{1.0f, 0x1.d906bcp-1f});
    wbfly(ar[3], ai[3], ar[7], ai[7], wf2_t{-0x1.6a09e6p-1f, -0x1.d906bcp-1f}, wf2_t{0x1.6a09e6p-1f, 0x1.87de2ap-2f});
#pragma unroll
    for (int k = 0; k < 8; ++k) { cr[k] = ar[k]; ci[k] = ai[k]; }   // band pair k = (2k, 2k+1)

#pragma unroll
    for (int k = 0; k < 8; ++k) { asm volatile("" : "+v"(cr[k])); asm volatile("" : "+v"(ci[k])); }   // pins the DFT before the fence
    WSTAMP();                                                   // previous block's resampler reads issued + DFT
    wfence();
    resample_finish();
    wfence();
    WSTAMP();                                                   // previous block's audio samples
    // ---- 4. c[s-1] from the left lane: one DPP move per component (wave_shr:1); lane 0 keeps the value handed over in LDS by
    //         lane 63 of the previous block (or the call's carried state / zeros)
    float dn[16];
    {
      const wf4_t* hand = reinterpret_cast<const wf4_t*>(cslot);
      wf2_t pr[8], pi[8];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const wf4_t a = hand[m], b = hand[4 + m];
        pr[2 * m] = wf2_t{wshr1(a.x, cr[2 * m].x), wshr1(a.y, cr[2 * m].y)};
        pr[2 * m + 1] = wf2_t{wshr1(a.z, cr[2 * m + 1].x), wshr1(a.w, cr[2 * m + 1].y)};
        pi[2 * m] = wf2_t{wshr1(b.x, ci[2 * m].x), wshr1(b.y, ci[2 * m].y)};
        pi[2 * m + 1] = wf2_t{wshr1(b.z, ci[2 * m + 1].x), wshr1(b.w, ci[2 * m + 1].y)};
      }
      wf2_t d8[8];
      wdisc_bands8(cr, ci, pr, pi, d8);
#pragma unroll
      for (int k = 0; k < 8; ++k) { dn[2 * k] = d8[k].x; dn[2 * k + 1] = d8[k].y; }
    }
    wfence();
    if (lane == 63) {                                           // for the next block: this block's last c
      wf4_t* z = reinterpret_cast<wf4_t*>(cslot);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        z[m] = wf4_t{cr[2 * m].x, cr[2 * m].y, cr[2 * m + 1].x, cr[2 * m + 1].y};
        z[4 + m] = wf4_t{ci[2 * m].x, ci[2 * m].y, ci[2 * m + 1].x, ci[2 * m + 1].y};
      }
    }
#ifdef SDRFM_DEV
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(dn[k]));   // (keeps the discriminator before its stamp)
#endif
    WSTAMP();                                                   // discriminator
    // ---- 5. d rows (ring columns; the last HDMAX columns are mirrored below column 0, so a window never wraps)
    {
      int c = dcb + lane;
      if (c >= DRING) c -= DRING;
      float* const col = drow + HDMAX + c;
#pragma unroll
      for (int b = 0; b < 16; ++b) col[b * DROW] = dn[b];
      if (c >= DRING - HDMAX) {
#pragma unroll
        for (int b = 0; b < 16; ++b) col[b * DROW - DRING] = dn[b];
      }
    }
    const int hi = s0 + 64 < tb ? s0 + 64 : tb;
    p_s0 = s0; p_dcb = dcb; p_lo = jl_lo; p_hi = first_jl(hi);
    jl_lo = p_hi;
    dcb += 64;
    if (dcb >= DRING) dcb -= DRING;
    ring0 += 64;
    if (ring0 >= RING) ring0 -= RING;
    __syncthreads();
    WSTAMP();                                                   // d rows, bookkeeping
  }
  resample_issue();                                             // the last block's audio
  wfence();
  resample_finish();
  __syncthreads();
#ifdef SDRFM_DEV
  if (tsp && lane == 0) tsp[255] = __builtin_amdgcn_s_memrealtime();
#endif
#undef WSTAMP

  // ---- state hand-over by the last run of the stream ----------------------------------------------------------------
  if (last_run) {
    const int sl = (int)w.Tn - 1 - s0_last;                      // lane of the call's last step in the last block
    if (lane == sl) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w.cprev_out[(size_t)stream * NB + 2 * k] = make_float2(cr[k].x, ci[k].x);
        w.cprev_out[(size_t)stream * NB + 2 * k + 1] = make_float2(cr[k].y, ci[k].y);
      }
    }
    for (int i = lane; i < 16 * HD; i += 64) {
      const int b = i / HD, k = i % HD;
      int c = p_dcb + ((int)w.Tn - s0_last) - HD + k;           // column of step Tn - HD + k (p_dcb: the last block's column base)
      if (c >= DRING) c -= DRING;
      w.hist_d_out[((size_t)stream * NB + b) * HD + k] = drow[b * DROW + HDMAX + c];
    }
    for (uint32_t i = (uint32_t)lane; i + 1 < w.P; i += 64)
      w.hist_x_out[(size_t)stream * (w.P - 1) + i] = wload_x(w, stream, (int)w.N - (int)(w.P - 1) + (int)i);
  }
}

}  // namespace

struct sdrfm_wbfm {
  sdrfm_wbfm_config cfg;
  int device;
  hipStream_t own_stream, stream;
  float *d_p, *d_g, *d_pperm;
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
  bool fused_ok;          // P = 128, HD <= 10: the one-lane-per-branch fused kernel applies
  bool steps_ok;          // ... and L >= 2: the one-lane-per-step kernel applies (preferred)
  uint32_t n_cu;          // compute units (fused kernel: run-length choice)
  uint32_t force_nt;      // SDRFM_WBFM_CFG_RUN_STEPS: fixed run length (tests)
  char kernel_name[48];
#ifdef SDRFM_DEV
  unsigned long long* d_dbg;   // development build: time stamps of the step kernel (SDRFM_WBFM_PROFILE=1)
  uint32_t dbg_waves;
#endif
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
  void* ptrs[] = {h->d_p, h->d_g, h->d_pperm, h->d_hist_x[0], h->d_hist_x[1], h->d_cprev[0], h->d_cprev[1], h->d_hist_d[0], h->d_hist_d[1],
                  h->d_dbuf, h->d_iq, h->d_audio};
  for (void* q : ptrs) if (q) (void)hipFree(q);
#ifdef SDRFM_DEV
  if (h->d_dbg) (void)hipFree(h->d_dbg);
#endif
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
  if (!cfg || cfg->struct_size != sizeof(sdrfm_wbfm_config) || (cfg->flags & 0xfcu)) return SDRFM_EINVAL;
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
  CR(hipMalloc(&h->d_pperm, 4 * P));
  for (int i = 0; i < 2; ++i) {
    CR(hipMalloc(&h->d_hist_x[i], sizeof(float2) * ns * (P - 1)));
    CR(hipMalloc(&h->d_cprev[i], sizeof(float2) * ns * NB));
    CR(hipMalloc(&h->d_hist_d[i], sizeof(float) * ns * NB * h->HD));
  }
  CR(hipMalloc(&h->d_dbuf, sizeof(float) * ns * NB * h->dcap));
  CR(hipMemcpy(h->d_p, pc, 4 * P, hipMemcpyHostToDevice));
  CR(hipMemcpy(h->d_g, gc, 4 * cfg->resamp_taps, hipMemcpyHostToDevice));
  {
    // taps in DFT input order for the step kernel: pperm[16 q + k] = p[bitrev4(k) + 16 q]
    float* pp = (float*)malloc(4 * P);
    if (!pp) { wfree(h); return SDRFM_ENOMEM; }
    for (size_t q = 0; q < P / NB; ++q)
      for (int k = 0; k < NB; ++k) {
        const int r = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);
        pp[NB * q + k] = pc[r + NB * q];
      }
    const hipError_t e = hipMemcpy(h->d_pperm, pp, 4 * P, hipMemcpyHostToDevice);
    free(pp);
    CR(e);
  }
#undef CR
  h->NT = 64;
  h->lds_bytes = ((size_t)(h->NT + 1) * NB + P) * 8 + (size_t)(h->NT + 1) * NB * 8 + P * 4;
  if (h->lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_wbfm_chan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) {
    wfree(h);
    return SDRFM_NOT_SUPPORTED;
  }
  h->fused_ok = (cfg->proto_taps == 128 && h->HD <= 10 && cfg->resamp_up * 10u <= 512u && !(cfg->flags & SDRFM_WBFM_CFG_FORCE_GENERIC));
  h->steps_ok = h->fused_ok && cfg->resamp_up >= 2 && !(cfg->flags & SDRFM_WBFM_CFG_BRANCH_LANES);
  h->n_cu = (uint32_t)prop.multiProcessorCount;
  h->force_nt = (cfg->flags >> SDRFM_WBFM_CFG_RUN_STEPS_SHIFT) & ~1u;   // test hook: fixed run length of the fused kernel (0 = chosen per call)
  if (h->force_nt && (h->force_nt < 64u || h->force_nt > 8192u)) { wfree(h); return SDRFM_EINVAL; }   // the range the kernels' own choice stays in
  snprintf(h->kernel_name, sizeof(h->kernel_name), "%s", h->steps_ok ? "wbfm-fused (k_wbfm_steps<8,10>)" : h->fused_ok ? "wbfm-fused (k_wbfm_fused<8,10>)" : "wbfm-generic (k_wbfm_chan + k_wbfm_res)");
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

const char* sdrfm_wbfm_kernel_name(const sdrfm_wbfm_t* h) { return h ? h->kernel_name : ""; }

#ifdef SDRFM_DEV
/* Development library only (not in include/sdrfm.h): the step kernel's time stamps of the last stamped launch, 256 words per wave:
 * [0] entry, [1] prologue done, then 7 per block (top, FIR, next tile, resampler issue + DFT, resampler finish, discriminator,
 * d rows); [253] HW_ID, [254] / [255] s_memrealtime at entry / exit.  Returns the number of waves through *n_waves. */
int sdrfm_wbfm_dev_read_debug(sdrfm_wbfm_t* h, unsigned long long* out, uint32_t max_waves, uint32_t* n_waves) {
  if (!h || !out || !n_waves || !h->d_dbg) return SDRFM_EINVAL;
  WTRY(hipSetDevice(h->device), SDRFM_FAIL);
  WTRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  const uint32_t n = h->dbg_waves < max_waves ? h->dbg_waves : max_waves;
  WTRY(hipMemcpy(out, h->d_dbg, sizeof(unsigned long long) * 256 * (size_t)n, hipMemcpyDeviceToHost), SDRFM_FAIL);
  *n_waves = n;
  return SDRFM_OK;
}
#endif

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
  w.dbuf = h->d_dbuf; w.p = h->d_p; w.g = h->d_g; w.pperm = h->d_pperm;
#ifdef SDRFM_DEV
  w.dbg = nullptr; w.dbg_light = 0;
  w.ablate = getenv("SDRFM_WBFM_ABLATE") ? (uint32_t)atoi(getenv("SDRFM_WBFM_ABLATE")) : 0u;
#endif
  w.swap_pct = 40u;   // measured on configs[4]: swap at 0 / 30 / 35 / 40 / 45 / 50 / 60 / 100 % -> 106 / 102 / 101 / 100 / 100.6 / 101 / 104 / 104.5 us
  w.inv_L32 = c.resamp_up >= 2 ? (uint32_t)((1ull << 32) / c.resamp_up) + 1u : 0u;
  w.P = c.proto_taps; w.Tg = c.resamp_taps; w.L = c.resamp_up; w.M = c.resamp_down; w.HD = h->HD; w.dcap = h->dcap;
  w.N = N; w.Tn = Tn; w.A = A; w.phase_x = h->phase_x; w.n_d = h->n_d; w.n_a = h->n_a;
  w.res_q0 = (int)((long long)((h->n_a * c.resamp_down) / c.resamp_up) - (long long)h->n_d);
  w.res_r0 = (uint32_t)((h->n_a * c.resamp_down) % c.resamp_up);
  const uint64_t span = (uint64_t)(c.n_streams - 1) * iq_stride + nbytes;
  w.iq_span = (uint32_t)span;
  if (h->steps_ok && Tn >= 64 && 4u * c.resamp_up <= c.resamp_down && ((uint64_t)A + 16) * c.resamp_down + c.resamp_up < (1ull << 26) && ((uint64_t)Tn + 64) * c.resamp_up < (1ull << 31)) {
    // step kernel: one wave per run of NT steps of one stream; a run that does not start the call re-computes HD + 1 steps.
    // One round of two waves per SIMD (8 waves per CU) when the streams allow it, runs of at least 128 steps.
    uint64_t runs = (8ull * h->n_cu + c.n_streams - 1) / c.n_streams;
    if (runs > Tn / 128) runs = Tn / 128;
    if (runs < 1) runs = 1;
    w.NT = (uint32_t)((Tn + runs - 1) / runs);
    if (h->force_nt) w.NT = h->force_nt;
    w.tiles_per_stream = (Tn + w.NT - 1) / w.NT;
    w.n_streams = c.n_streams;
    const size_t lds = WStepsLds<8, 10>::bytes(c.resamp_up);
#ifdef SDRFM_DEV
    if (getenv("SDRFM_WBFM_PROFILE") && c.n_streams * w.tiles_per_stream <= 16384u) {   // development build: stamp this launch
      if (!h->d_dbg) WTRY(hipMalloc(&h->d_dbg, sizeof(unsigned long long) * 256 * 16384), SDRFM_ENOMEM);
      w.dbg = h->d_dbg;
      w.dbg_light = atoi(getenv("SDRFM_WBFM_PROFILE")) == 2;
      h->dbg_waves = c.n_streams * w.tiles_per_stream;
    }
#endif
    hipLaunchKernelGGL((k_wbfm_steps<8, 10>), dim3(c.n_streams * w.tiles_per_stream), dim3(64), lds, h->stream, w);
    WTRY(hipGetLastError(), SDRFM_FAIL);
    snprintf(h->kernel_name, sizeof(h->kernel_name), "wbfm-fused (k_wbfm_steps<8,10>)");   // the kernel that served THIS call
    h->cur ^= 1;
    h->phase_x = (uint32_t)((h->phase_x + (uint64_t)N) % NB);
    h->n_d += Tn; h->n_a += A;
    return SDRFM_OK;
  }
  if (h->fused_ok && Tn >= 64 && span < (1ull << 32)) {
    // fused kernel: every stream is cut into runs of NT steps (even: steps are processed in pairs); one 16-lane group per
    // run, the 4 groups of a wave = the same run of 4 neighbouring streams, 4 waves per block.  Every run re-computes ~20 warm-up
    // steps (12.7 % of the work and of the input traffic at NT = 158), so runs should be long; the kernel is VALU-bound and two
    // blocks per CU (two waves per SIMD) already keep the pipe as busy as four do (128 streams x 20 000 steps, same box, three
    // measurements each: NT = 158, 4 blocks per CU: 113-127 us; NT = 210, 3 per CU: 118-132 us; NT = 312-320, 2 per CU: 116-120 us).
    // NT is chosen so that the grid is R full rounds of 2 blocks per CU, with the R that minimises rounds x (NT + 20); a handful
    // of surplus blocks costs nothing, they start in the skew of the first finishers.
    const uint64_t quads = (c.n_streams + 3) / 4;
    uint64_t best = ~0ull;
    w.NT = 64;
    for (uint32_t R = 1; R <= 64; ++R) {
      uint64_t runs = 8ull * h->n_cu * R / quads;               // runs per stream that make R rounds of 2 blocks per CU
      if (runs < 1) runs = 1;
      uint64_t nt = ((Tn + runs - 1) / runs + 1) & ~1ull;
      if (nt < 64) nt = 64;
      if (nt > 8192) nt = 8192;
      const uint64_t blocks = (quads * ((Tn + nt - 1) / nt) + 3) / 4;
      const uint64_t rounds = (blocks + 2ull * h->n_cu - 1) / (2ull * h->n_cu);
      const uint64_t cost = rounds * (nt + 20);
      if (cost < best) { best = cost; w.NT = (uint32_t)nt; }
      if (nt == 64) break;
    }
    if (h->force_nt) w.NT = h->force_nt;
    w.tiles_per_stream = (Tn + w.NT - 1) / w.NT;
    w.n_streams = c.n_streams;
    const uint32_t waves = (uint32_t)quads * w.tiles_per_stream;
    hipLaunchKernelGGL((k_wbfm_fused<8, 10>), dim3((waves + 3) / 4), dim3(256), 0, h->stream, w);
    WTRY(hipGetLastError(), SDRFM_FAIL);
    snprintf(h->kernel_name, sizeof(h->kernel_name), "wbfm-fused (k_wbfm_fused<8,10>)");
    h->cur ^= 1;
    h->phase_x = (uint32_t)((h->phase_x + (uint64_t)N) % NB);
    h->n_d += Tn; h->n_a += A;
    return SDRFM_OK;
  }
  w.NT = h->NT; w.tiles_per_stream = (Tn + h->NT - 1) / h->NT; w.n_streams = c.n_streams;
  hipLaunchKernelGGL(k_wbfm_chan, dim3(c.n_streams * w.tiles_per_stream + c.n_streams), dim3(256), h->lds_bytes, h->stream, w);
  WTRY(hipGetLastError(), SDRFM_FAIL);
  const uint32_t per_stream = (A * NB + 255) / 256;
  hipLaunchKernelGGL(k_wbfm_res, dim3(c.n_streams * per_stream + c.n_streams), dim3(256), 0, h->stream, w);
  WTRY(hipGetLastError(), SDRFM_FAIL);
  snprintf(h->kernel_name, sizeof(h->kernel_name), "wbfm-generic (k_wbfm_chan + k_wbfm_res)");
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
