/*
 * sdrfm_b.h — design B ("bytes in LDS") as a device function, with the call parameters and the input / state helpers it shares with the other
 * bit-exact kernels of sdrfm.hip.  Internal to the library.  Two translation units include it: sdrfm.hip (the kernel k_fastb and everything else
 * that uses the helpers) and sdrfm_q.hip (the one-launch kernel of a mixed batch: design B workgroups for the noise-only streams beside design Q's
 * workgroups for the others).  Both are compiled with -ffp-contract=off; every operation below is an explicit IEEE one, so the bits do not depend
 * on which unit the body was compiled in.
 */
#ifndef SDRFM_B_H
#define SDRFM_B_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "sdrfm_math.h"

typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef int i4_t __attribute__((ext_vector_type(4)));
__device__ f4_t llvm_amdgcn_raw_buffer_load_format_v4f32(i4_t rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ int llvm_amdgcn_raw_buffer_load_i32(i4_t rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.i32");
__device__ i4_t llvm_amdgcn_raw_buffer_load_v4i32(i4_t rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4i32");
// LDS-DMA: global -> LDS without VGPRs; lane t's `size` bytes land at lds + t * size (buffer_load_dwordx4 ... lds)
__device__ void llvm_amdgcn_raw_buffer_load_lds(i4_t rsrc, __attribute__((address_space(3))) void* lds, int size, int voffset,
                                                int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

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
  const uint8_t* hist_b_in;   // raw I/Q bytes of the last T-1 inputs (2*(T-1) per stream), same samples as hist_x
  uint8_t* hist_b_out;
  const float* h;  // T taps
  const float* g;  // Ta taps
  uint32_t T, D, Ta, Da;
  uint32_t N;   // new IQ samples per stream in this call
  uint32_t M;   // decimated outputs y[0..M) produced by this call
  uint32_t A;   // audio outputs a[0..A) produced by this call
  int32_t e0;   // chunk index of the newest input of y[0]:  D-1-phase_x
  int32_t f0;   // call-relative index of the newest d of a[0]: Da-1-phase_d
  uint32_t NA;  // audio outputs per tile (generic kernel) / per segment (fast kernel)
  uint32_t tiles_per_stream;  // tiles (generic) / segments (fast) per stream
  uint32_t n_streams;
  uint32_t phase_x;           // inputs already consumed towards y[0] (0..D-1)
  uint32_t AB;                // fast kernel: sub-tiles of d buffered per audio flush
  uint32_t warm_ahead;        // fast kernel: L2 warm-up distance in sub-tiles (0 = off)
  uint32_t prio_balance;      // design B: wave priority falls with progress (keeps the two waves of a SIMD in step)
  uint32_t end_prio;          // design S: priorities of the last body / head pass / audio stage: 2 bits each, slot-0 wave in bits 0-5, slot-1 wave in bits 6-11
  uint32_t dbg_tag;           // profiling build: 1 on the one launch whose wave start/end skew is recorded
  uint32_t fold_state;        // design B: the last segment's wave hands the state over (no state blocks in the grid)
  unsigned long long* dbg;    // phase-cycle accumulators (profiling build of the fast kernel only), else nullptr
  const uint32_t* slist;      // nullptr: the launch serves streams 0 .. n_streams-1; else the i-th stream of the launch is stream slist[i] of the
                              // handle (round 5: streams routed per stream between design Q and the bit-exact kernels; n_streams = the list's length)
  const uint8_t* iq_prev;     // design B beside an overlapped design-Q call: the previous call's buffer (rows of N_prev samples, iq_prev_stride bytes
  size_t iq_prev_stride;      //   apart).  The samples before the call are read from its end and NO carried state is read: the stream's first segment
  uint32_t N_prev;            //   recomputes the d's before its first audio output like every other segment does.  nullptr: the carried state, as ever
};

namespace {

// the handle's stream number of the launch's i-th stream
__device__ __forceinline__ uint32_t launch_stream(const CallParams& p, uint32_t i) { return p.slist ? p.slist[i] : i; }

// ---- virtual input: chunk index s in [-(T-1), N) ---------------------------------------------------------------
__device__ __forceinline__ float2 load_x(const CallParams& p, uint32_t stream, int s) {
  if (s < 0 && !p.iq_prev) return p.hist_x_in[(size_t)stream * (p.T - 1) + (p.T - 1 + s)];
  const uint8_t* b = (s < 0) ? p.iq_prev + (size_t)stream * p.iq_prev_stride + 2 * (size_t)((int)p.N_prev + s)
                             : p.iq + (size_t)stream * p.iq_stride + 2 * (size_t)s;
  const uchar2 v = *reinterpret_cast<const uchar2*>(b);
  return make_float2((float)v.x - 127.5f, (float)v.y - 127.5f);
}


// raw I/Q byte pair of chunk index s in [-(T-1), N) (low byte = I); with p.iq_prev: s in [-N_prev, N)
__device__ __forceinline__ unsigned load_raw(const CallParams& p, uint32_t stream, int s) {
  const uint8_t* b = (s >= 0) ? p.iq + (size_t)stream * p.iq_stride + 2 * (size_t)s
                   : p.iq_prev ? p.iq_prev + (size_t)stream * p.iq_prev_stride + 2 * (size_t)((int)p.N_prev + s)
                               : p.hist_b_in + ((size_t)stream * (p.T - 1) + (p.T - 1 + s)) * 2;
  return *reinterpret_cast<const unsigned short*>(b);
}

// ---- state hand-over of one stream: new FIR history, y[M-1], new discriminator history -------------------------
// Runs in its own block(s) of the same launch as the audio tiles; reads only the call's inputs and the OLD state set,
// writes only the NEW state set, so it is independent of every other block.
// scratch: xs (capacity xs_cap samples) and ys (>= Ta+1 entries) in LDS; hs = FIR taps in LDS.
__device__ __forceinline__ void state_handover(const CallParams& p, uint32_t stream, float2* xs, uint32_t xs_cap, float2* ys,
                                               const float* hs) {
  const uint32_t T = p.T, D = p.D, Ta = p.Ta;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  const int N = (int)p.N, M = (int)p.M;
  // new input history = last T-1 samples of [old history | chunk]
  for (uint32_t k = tid; k + 1 < T; k += nthr) {
    p.hist_x_out[(size_t)stream * (T - 1) + k] = load_x(p, stream, N - (int)(T - 1) + (int)k);
    reinterpret_cast<unsigned short*>(p.hist_b_out)[(size_t)stream * (T - 1) + k] =
        (unsigned short)load_raw(p, stream, N - (int)(T - 1) + (int)k);
  }
  // y[M-Ta .. M-1] (those that exist) -> ys[0..Ta)
  const int ylo = M - (int)Ta;
  const int yc0 = ylo > 0 ? ylo : 0;
  const int xlo = p.e0 + yc0 * (int)D - (int)(T - 1);
  const int xhi = p.e0 + (M - 1) * (int)D;
  const bool staged = (M > 0) && ((uint32_t)(xhi - xlo + 1) <= xs_cap);
  if (staged)
    for (int s = xlo + (int)tid; s <= xhi; s += (int)nthr) xs[s - xlo] = load_x(p, stream, s);
  __syncthreads();
  for (int q = (int)tid; q < (int)Ta; q += (int)nthr) {
    const int i = ylo + q;
    float2 y = make_float2(0.f, 0.f);
    if (i >= 0) {
      const int s0 = p.e0 + i * (int)D - (int)(T - 1);
      float ar = 0.0f, ai = 0.0f;
      for (uint32_t j = 0; j < T; ++j) {
        const float c = hs[T - 1 - j];
        const float2 x = staged ? xs[s0 + (int)j - xlo] : load_x(p, stream, s0 + (int)j);
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
      const float2 pr = ys[i - 1 - ylo];  // i-1-ylo >= 0 always; index -1 was filled from the old state above
      d = sdrfm_discriminate(y.x, y.y, pr.x, pr.y);
    }
    p.hist_d_out[(size_t)stream * (Ta - 1) + q] = d;
  }
}

// buffer resource word3: dst_sel = (R,G,B,A), num_format = USCALED (2), data_format = 8_8_8_8 (10)
#define SDRFM_RSRC_U8X4_USCALED 0x52FAC

// acc += tap * x, tap = low (HI=0) or high (HI=1) half of a wave-uniform SGPR pair, broadcast to both lanes of the pack
template <int HI>
__device__ __forceinline__ void pk_fma_bcast(f2_t& acc, f2_t tap_pair, f2_t x) {
  if constexpr (HI == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(tap_pair), "v"(x));
}

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
// bytes of the x tile: positions [0, HP + NST) in rows of R*D samples, last row trimmed, rounded up to 16 B
constexpr int fast_xbytes(int T, int D, int R) {
  const int RD = R * D, HP = T - D, NST = 64 * RD, RS = (RD + (((RD / 2) % 2 == 0) ? 2 : 0)) * 8;
  const int last = HP + NST - 1;
  return (((last / RD) * RS + (last % RD + 1) * 8) + 15) & ~15;
}

// ---- packed (2-wide) discriminator: two consecutive outputs per instruction stream --------------------------------
__device__ __forceinline__ f2_t pk_splat(float c) { return f2_t{c, c}; }
__device__ __forceinline__ f2_t atan2_pair(f2_t y, f2_t x) {
  const f2_t ax = __builtin_elementwise_abs(x), ay = __builtin_elementwise_abs(y);
  const f2_t mx = __builtin_elementwise_max(ax, ay), mn = __builtin_elementwise_min(ax, ay);
  const f2_t t = mn * f2_t{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
  const f2_t s = t * t;
  f2_t q = pk_splat(0x1.57b128p-9f);
  q = __builtin_elementwise_fma(q, s, pk_splat(-0x1.efda1p-7f));
  q = __builtin_elementwise_fma(q, s, pk_splat(0x1.50dd96p-5f));
  q = __builtin_elementwise_fma(q, s, pk_splat(-0x1.2dbcfap-4f));
  q = __builtin_elementwise_fma(q, s, pk_splat(0x1.b11b74p-4f));
  q = __builtin_elementwise_fma(q, s, pk_splat(-0x1.228754p-3f));
  q = __builtin_elementwise_fma(q, s, pk_splat(0x1.99673ep-3f));
  q = __builtin_elementwise_fma(q, s, pk_splat(-0x1.55546cp-2f));
  f2_t a = __builtin_elementwise_fma(t, s * q, t);
  float a0 = a.x, a1 = a.y;
  if (ay.x > ax.x) a0 = 0x1.921fb6p+0f - a0;
  if (ay.y > ax.y) a1 = 0x1.921fb6p+0f - a1;
  if (x.x < 0.0f) a0 = 0x1.921fb6p+1f - a0;
  if (x.y < 0.0f) a1 = 0x1.921fb6p+1f - a1;
  return f2_t{__builtin_copysignf(a0, y.x), __builtin_copysignf(a1, y.y)};
}
// d for outputs (y0 | prev p0) and (y1 | prev y0): same roundings as sdrfm_discriminate, two at a time
__device__ __forceinline__ f2_t discriminate_pair(f2_t y0, f2_t p0, f2_t y1) {
  const f2_t yr = {y0.x, y1.x}, yi = {y0.y, y1.y}, pr = {p0.x, y0.x}, pi = {p0.y, y0.y};
  const f2_t re = __builtin_elementwise_fma(yr, pr, yi * pi);
  const f2_t im = yi * pr - yr * pi;
  const f2_t a = atan2_pair(im, re);
  return f2_t{(re.x == 0.0f && im.x == 0.0f) ? 0.0f : a.x, (re.y == 0.0f && im.y == 0.0f) ? 0.0f : a.y};
}

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
// =================================================================================================================
//  Fast kernel, design B ("bytes in LDS"): the wave's tile holds the RAW u8 I/Q bytes (2 B per sample instead of 8),
//  every lane converts its own window on the fly and computes R = 8..12 consecutive outputs from it.
//
//    HBM --buffer_load_dwordx4 (16 B/lane, prefetched one sub-tile ahead in VGPRs)--> ds_write_b128 --> LDS raw tile
//    LDS --ds_read_b128 (8 samples)--> v_cvt_f32_ubyte0..3 + v_pk_add_f32(-127.5) --> (I,Q) f32 pairs
//        --> v_pk_fma_f32 with SGPR taps, up to T/D outputs per sample --> y --> discriminator --> d ring --> audio
//
//  Why (measured, tools/ubench): design A moves 8 B per sample through ds_write_b128, whose VGPR->LDS path costs ~13
//  cycles per wave-instruction per CU; at two waves per SIMD that path, the LDS reads and the VALU work do not overlap
//  and the kernel runs at ~3x the time of any one of them.  Here the store path carries 4x fewer bytes, LDS reads are
//  ~8x fewer, and the only saturated resource is the VALU: conversion is repeated for the (T-D)-sample overlap between
//  neighbouring lanes, (R*D+T-D)/(R*D) = 1.45x at R = 12, which costs less than the LDS round trip of the floats.
//
//  Requirements (else design A or the generic kernel): as design A, plus R*D % 8 == 0 and all T-1 history samples real
//  (the zero-history start of a stream cannot be expressed in bytes: the first call after a reset runs elsewhere).
// =================================================================================================================
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int fastb_hp(int T, int D) { return ((T - D) + 7) & ~7; }                      // halo samples (16-B granular)
constexpr int fastb_rs(int D, int R) { return R * D * 2 + ((((R * D * 2) / 16) % 2 == 0) ? 16 : 0); }  // row stride, B
// sub-tiles of d's buffered per audio flush: the small tile (R = 4: 256 d's = 51 audio outputs per sub-tile, which leaves the flush's three chains per lane
// two thirds idle) flushes every third sub-tile
#ifndef SDRFM_B_AB_SMALL
#define SDRFM_B_AB_SMALL 3
#endif
constexpr int fastb_ab(int R) { return R <= 4 ? SDRFM_B_AB_SMALL : 1; }
constexpr int fastb_xbytes(int T, int D, int R) {
  const int RD = R * D, last = fastb_hp(T, D) + 64 * RD - 1;
  return (((last / RD) * fastb_rs(D, R) + (last % RD + 1) * 2) + 15) & ~15;
}

// (I - 127.5, Q - 127.5) of the low / high half of a dword holding two I/Q byte pairs
template <int HIHALF>
__device__ __forceinline__ f2_t cvt_iq(unsigned w) {
  f2_t c;
  if constexpr (HIHALF) { c.x = (float)((w >> 16) & 0xffu); c.y = (float)(w >> 24); }   // v_cvt_f32_ubyte2 / 3
  else { c.x = (float)(w & 0xffu); c.y = (float)((w >> 8) & 0xffu); }                  // v_cvt_f32_ubyte0 / 1
  return c - f2_t{127.5f, 127.5f};
}

// same with the tap pair held in a (wave-uniform) VGPR pair: frees SGPRs when the kernel is SGPR-bound
template <int HI>
__device__ __forceinline__ void pk_fma_bcast_v(f2_t& acc, f2_t tap_pair, f2_t x) {
  if constexpr (HI == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(tap_pair), "v"(x));
}

// The body of one workgroup (one wave): `bid` is its index among the launch's design-B workgroups (the kernel's bid, or — in the
// one-launch kernel of a mixed batch, sdrfm_q.hip — its index among the workgroups that run this body).
template <int T, int D, int R, int TA, int DA, int MODE>
__device__ __forceinline__ void fastb_body(const CallParams& p, const uint32_t bid) {
  constexpr int RD = R * D, NYT = 64 * R, NST = 64 * RD, HALO = T - D, HP = fastb_hp(T, D), OFF = HP - HALO;
  constexpr int NW = RD + HALO;                      // samples one lane needs
  constexpr int NRD = (OFF + NW + 7) / 8;            // ds_read_b128 (8 samples) per lane
  constexpr int RS = fastb_rs(D, R);                 // lane row stride in bytes (odd number of 16-B slots)
  constexpr bool LINEAR = (RS == RD * 2);
  constexpr int XBYTES = fastb_xbytes(T, D, R);
  constexpr int NLOAD = RD / 8;                      // 16-B loads per lane per sub-tile
  static_assert(T % 2 == 0 && D % 2 == 0 && T >= D && RD % 8 == 0 && R % 2 == 0, "design B geometry");
  constexpr bool PROF = (MODE == 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xb = smem;                                   // raw tile: position u <-> sub-tile sample s' = u - HP
  constexpr uint32_t Ta = TA, Da = DA;                         // audio stage geometry is compile-time in design B
  constexpr uint32_t DOFF = (Ta - 1 + 3u) & ~3u;
  constexpr int DCAP = fastb_ab(R) * NYT;                     // one audio flush per sub-tile (every third one at R = 4)
  float* dbuf = reinterpret_cast<float*>(smem + XBYTES);
  float* gs = dbuf + DOFF + DCAP;
  float* hs = gs + Ta;
  const int lane = (int)threadIdx.x;
  unsigned t_entry = 0;
  unsigned long long rt_entry = 0;
  if constexpr (PROF) { t_entry = (unsigned)__builtin_readcyclecounter(); rt_entry = __builtin_amdgcn_s_memrealtime(); }

  const uint32_t n_seg_blocks = p.n_streams * p.tiles_per_stream;
  if (bid >= n_seg_blocks) {   // only launched when the state is not folded into the last segment (see below)
    for (uint32_t k = lane; k < (uint32_t)T; k += 64) hs[k] = p.h[k];
    __syncthreads();
    state_handover(p, launch_stream(p, bid - n_seg_blocks), reinterpret_cast<float2*>(xb), XBYTES / 8, reinterpret_cast<float2*>(dbuf), hs);
    return;
  }
  const uint32_t stream = launch_stream(p, bid / p.tiles_per_stream);
  const uint32_t seg = bid % p.tiles_per_stream;
  const int j0 = (int)(seg * p.NA);
  int j1 = j0 + (int)p.NA;
  if (j1 > (int)p.A) j1 = (int)p.A;
  if (j0 >= j1) return;
  for (uint32_t k = lane; k < Ta; k += 64) gs[k] = p.g[Ta - 1 - k];

  int ibase = p.f0 + j0 * (int)Da - (int)(Ta - 1) - 1;
  // (with the previous call's buffer at hand the stream's first segment is a segment like any other: its first d's — one of them never used,
  // its y[m-1] being unknown — come from the samples before the call, which are read where the previous call left them)
  const bool use_hist = ibase <= 0 && !p.iq_prev;
  if (use_hist) ibase = 0;
  int i_end = p.f0 + (j1 - 1) * (int)Da;                      // newest d needed by this segment's audio
  // The wave of the LAST segment also hands the streaming state over (y[M-1], the last Ta-1 d's, the last T-1 inputs):
  // it already holds all of it at the end of its last sub-tile, so no separate state blocks (and no tail) are needed.
  const bool hand_over = p.fold_state && (j1 == (int)p.A);
  if (hand_over && (int)p.M - 1 > i_end) i_end = (int)p.M - 1;
  const int nst = (i_end - ibase) / NYT + 1;
  int cs = (int)D * ibase - (int)p.phase_x;                   // chunk index of sub-tile sample s' = 0 (even)

  // taps: wave-uniform pairs; the first NVT pairs live in VGPRs, the rest in SGPRs (64 taps alone would take 64 of the
  // ~100 usable SGPRs and push kernel arguments into spills)
  constexpr int NVT = (T >= 64) ? 12 : 0;   // (all 32 pairs in VGPRs — round 5, after an FMA with an SGPR operand was found to issue at half rate in design Q: 130 VGPRs at R = 4, no faster: profiles/r05_mixed_batches.txt)
  f2_t hp[T / 2];
#pragma unroll
  for (int k = 0; k < T / 2; ++k) {
    hp[k] = f2_t{p.h[2 * k], p.h[2 * k + 1]};
    if (k < NVT) asm volatile("" : "+v"(hp[k]));
  }

  auto pos_addr = [&](int u) -> unsigned char* {              // LDS address of tile position u
    if constexpr (LINEAR) return xb + 2 * u;
    else return xb + (u / RD) * RS + (u % RD) * 2;
  };
  const unsigned char* win = xb + lane * RS;                  // window of this lane: positions [RD*lane, RD*lane + 8*NRD)

  const unsigned long long gaddr = (unsigned long long)(p.iq + (size_t)stream * p.iq_stride);
  const i4_t rsrc = {(int)(unsigned)gaddr, (int)(unsigned)(gaddr >> 32), (int)(2u * p.N), 0x00020000};

  i4_t pre[NLOAD];
#pragma unroll
  for (int q = 0; q < NLOAD; ++q) pre[q] = llvm_amdgcn_raw_buffer_load_v4i32(rsrc, 2 * cs + (64 * q + lane) * 16, 0, 0);

  // prologue: halo of the first sub-tile (raw bytes, from the old raw history where the chunk has not started), d history
  for (int u = lane; u < HP; u += 64) {
    const int c = cs - HP + u;
    *reinterpret_cast<unsigned short*>(pos_addr(u)) = (c >= -(int)(T - 1) || p.iq_prev) ? (unsigned short)load_raw(p, stream, c) : (unsigned short)0;
  }
  for (uint32_t k = lane; k < Ta - 1; k += 64)
    dbuf[DOFF - (Ta - 1) + k] = use_hist ? p.hist_d_in[(size_t)stream * (Ta - 1) + k] : 0.0f;
  f2_t carry = {0.f, 0.f};
  if (use_hist) { const float2 yp = p.yprev_in[stream]; carry = f2_t{yp.x, yp.y}; }
  int dpos = 0, ibA = ibase, ibA_last = ibase;
  unsigned* tph = reinterpret_cast<unsigned*>(hs);
  unsigned tlast = 0;
  if constexpr (PROF) {
    if (lane < 8) tph[lane] = 0;
    tlast = (unsigned)__builtin_readcyclecounter();
  }
#define SDRFM_TICK(i)                                                         \
  if constexpr (PROF) {                                                       \
    const unsigned tn = (unsigned)__builtin_readcyclecounter();               \
    if (lane == 0) tph[i] += tn - tlast;                                      \
    tlast = tn;                                                               \
  }

  if constexpr (PROF) {
    __syncthreads();
    const unsigned tn = (unsigned)__builtin_readcyclecounter();
    if (lane == 0) tph[7] = tn - t_entry;                     // prologue cycles of this segment (kernel entry -> loop)
    tlast = tn;
  }
  for (int st = 0; st < nst; ++st) {
    // The two waves of a SIMD are arbitrated oldest-first, so one runs ahead and the other finishes alone at the
    // single-wave issue rate (measured: first wave done at 28 us, last at 47 us).  Priority falls with progress, so
    // whichever wave is behind wins the issue slot and the pair finishes together.
    if (p.prio_balance) {
      const int left = nst - 1 - st;
      if (left >= 3) __builtin_amdgcn_s_setprio(3);
      else if (left == 2) __builtin_amdgcn_s_setprio(2);
      else if (left == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    // ---- stage the prefetched raw bytes, then prefetch the next sub-tile --------------------------------------
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
      const int u = HP + 8 * (64 * q + lane);
      *reinterpret_cast<i4_t*>(pos_addr(u)) = pre[q];
    }
    cs += NST;
    if (st + 1 < nst) {
#pragma unroll
      for (int q = 0; q < NLOAD; ++q) pre[q] = llvm_amdgcn_raw_buffer_load_v4i32(rsrc, 2 * cs + (64 * q + lane) * 16, 0, 0);
    }
    __syncthreads();
    if (st == 0 && cs - NST < 0) {
      // The chunk starts inside this sub-tile: 16-B loads that begin before byte 0 come back as zeros in full, so every
      // sample of those loads is rewritten: from the old raw history (index < 0) or from the chunk itself.
      const int nfix = (-(cs - NST) + 7) & ~7;
      for (int s = lane; s < nfix; s += 64) {
        const int c = cs - NST + s;
        *reinterpret_cast<unsigned short*>(pos_addr(HP + s)) = (c < (int)p.N) ? (unsigned short)load_raw(p, stream, c) : (unsigned short)0;
      }
      __syncthreads();
    }
    SDRFM_TICK(0)
    // ---- K1 + K2: convert the lane's window 8 samples at a time, feed every output the sample belongs to -----
    f2_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f2_t{0.f, 0.f};
    u4_t cur = *reinterpret_cast<const u4_t*>(win), nxt = cur;
    static_for<0, NRD>([&](auto J8) {
      constexpr int j8 = decltype(J8)::value;
      constexpr int jn = 8 * (j8 + 1);                       // first window position of the next 8-sample chunk
      if constexpr (j8 + 1 < NRD) nxt = *reinterpret_cast<const u4_t*>(win + (jn / RD) * RS + (jn % RD) * 2);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 8>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr int j = j8 * 8 + s;                         // window position; output r uses positions OFF+r*D .. +T-1
        if constexpr (j >= OFF && j < OFF + NW) {
          const unsigned w = (s / 2 == 0) ? cur.x : (s / 2 == 1) ? cur.y : (s / 2 == 2) ? cur.z : cur.w;
          f2_t x;
          if constexpr (MODE == 2 && j < OFF + HALO) { x = f2_t{__uint_as_float(w), __uint_as_float(w)}; }   // ablation: halo not converted
          else if constexpr (MODE == 3 || MODE == 5) { x = f2_t{__uint_as_float(w), __uint_as_float(w)}; }    // ablation: nothing converted
          else x = cvt_iq<(s & 1)>(w);
          static_for<0, R>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            constexpr int p0 = j - OFF - r * D;               // 0 = oldest sample of output r
            if constexpr (MODE == 5) { if constexpr (p0 == 0) acc[r] = x; }   // ablation: no FIR (one use per output)
            else if constexpr (p0 >= 0 && p0 < T) {
              constexpr int k = T - 1 - p0;
              if constexpr (k / 2 < NVT) {
                if constexpr (k & 1) pk_fma_bcast_v<1>(acc[r], hp[k / 2], x); else pk_fma_bcast_v<0>(acc[r], hp[k / 2], x);
              } else {
                if constexpr (k & 1) pk_fma_bcast<1>(acc[r], hp[k / 2], x); else pk_fma_bcast<0>(acc[r], hp[k / 2], x);
              }
            }
          });
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    });
    if constexpr (PROF) asm volatile("" :: "v"(acc[0]), "v"(acc[R - 1]));
    SDRFM_TICK(1)
    // ---- K3 -------------------------------------------------------------------------------------------------------
    f2_t prev;
    prev.x = __shfl_up(acc[R - 1].x, 1);
    prev.y = __shfl_up(acc[R - 1].y, 1);
    if (lane == 0) prev = carry;
    carry.x = __shfl(acc[R - 1].x, 63);
    carry.y = __shfl(acc[R - 1].y, 63);
    float dv[R];
#pragma unroll
    for (int r = 0; r + 1 < R; r += 2) {
      f2_t d2;
      if constexpr (MODE == 4 || MODE == 5) d2 = f2_t{acc[r].x + prev.x, acc[r + 1].y};     // ablation: no discriminator
      else d2 = discriminate_pair(acc[r], r == 0 ? prev : acc[r - 1], acc[r + 1]);
      dv[r] = d2.x;
      dv[r + 1] = d2.y;
    }
#pragma unroll
    for (int r = 0; r < R; r += 4) {
      if (r + 4 <= R) *reinterpret_cast<f4_t*>(dbuf + DOFF + dpos + R * lane + r) = f4_t{dv[r], dv[r + 1], dv[r + 2], dv[r + 3]};
      else { dbuf[DOFF + dpos + R * lane + r] = dv[r]; dbuf[DOFF + dpos + R * lane + r + 1] = dv[r + 1]; }
    }
    dpos += NYT;
    const bool last = (st + 1 == nst);
    SDRFM_TICK(2)
    if (last && hand_over) {                                  // park y[M-1] in LDS; the hand-over itself runs after the loop
      const int o = (int)p.M - 1 - (ibase + st * NYT);        // y[M-1] is output o of this sub-tile (0 <= o < NYT)
      if (lane == o / R) {
        f2_t y = acc[0];
#pragma unroll
        for (int r = 1; r < R; ++r) if (o % R == r) y = acc[r];
        *reinterpret_cast<f2_t*>(hs + 16) = y;
      }
      ibA_last = ibA;
    }
    if (dpos == DCAP || last) {
      __syncthreads();
      int jl = (ibA - p.f0 + (int)Da - 1);
      jl = jl > 0 ? jl / (int)Da : 0;
      if (jl < j0) jl = j0;
      int jh = (ibA + dpos - 1 - p.f0);
      jh = jh >= 0 ? jh / (int)Da + 1 : 0;
      if (jh > j1) jh = j1;
      // three outputs per lane at a time (independent chains share the tap reads; a single chain is LDS-latency bound)
      for (int j = jl + lane; j < jh; j += 192) {
        const int jb = j + 64, jc = j + 128;
        const float* w0 = dbuf + DOFF + (p.f0 + j * (int)Da - ibA) - (int)(Ta - 1);
        const float* w1 = (jb < jh) ? w0 + 64 * (int)Da : w0;
        const float* w2 = (jc < jh) ? w0 + 128 * (int)Da : w0;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll 4
        for (uint32_t k = 0; k < Ta; ++k) {
          const float gk = gs[k];
          a0 = __builtin_fmaf(gk, w0[k], a0);
          a1 = __builtin_fmaf(gk, w1[k], a1);
          a2 = __builtin_fmaf(gk, w2[k], a2);
        }
        float* o = p.audio + (size_t)stream * p.audio_stride;
        __builtin_nontemporal_store(a0, o + j);               // streamed out once: do not leave dirty lines in L2
        if (jb < jh) __builtin_nontemporal_store(a1, o + jb);
        if (jc < jh) __builtin_nontemporal_store(a2, o + jc);
      }
      if (!last) {
        float keep[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t k = lane + 64 * i;
          keep[i] = (k < Ta - 1) ? dbuf[DOFF + dpos - (Ta - 1) + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t k = lane + 64 * i;
          if (k < Ta - 1) dbuf[DOFF - (Ta - 1) + k] = keep[i];
        }
      }
      ibA += dpos;
      dpos = 0;
    }
    SDRFM_TICK(3)
    // ---- carry the raw halo: positions [NST, NST+HP) -> [0, HP) ---------------------------------------------------
    if (!last) {
      if (lane < HP / 8) *reinterpret_cast<i4_t*>(pos_addr(8 * lane)) = *reinterpret_cast<const i4_t*>(pos_addr(NST + 8 * lane));
      __syncthreads();
    }
    SDRFM_TICK(4)
  }
  if (hand_over) {
    // state hand-over by the wave that computed the end of the stream's chunk (outside the loop: the taps are dead here).
    // Ring position of d[i] is DOFF + (i - ibA_last); d[M-(Ta-1) .. M-1] all lie inside the ring because M >= Ta.
    __syncthreads();
    if (lane == 0) { const f2_t y = *reinterpret_cast<const f2_t*>(hs + 16); p.yprev_out[stream] = make_float2(y.x, y.y); }
    for (uint32_t k = lane; k + 1 < Ta; k += 64)
      p.hist_d_out[(size_t)stream * (Ta - 1) + k] = dbuf[(int)DOFF + ((int)p.M - (int)(Ta - 1) + (int)k - ibA_last)];
    for (uint32_t k = lane; k + 1 < (uint32_t)T; k += 64) {
      const int c = (int)p.N - (int)(T - 1) + (int)k;
      p.hist_x_out[(size_t)stream * (T - 1) + k] = load_x(p, stream, c);
      reinterpret_cast<unsigned short*>(p.hist_b_out)[(size_t)stream * (T - 1) + k] = (unsigned short)load_raw(p, stream, c);
    }
  }
  if constexpr (PROF) {
    if (lane == 0 && p.dbg) {
      for (int i = 0; i < 5; ++i) atomicAdd(p.dbg + 8 * (bid & 63) + i, (unsigned long long)tph[i]);
      atomicAdd(p.dbg + 8 * (bid & 63) + 5, (unsigned long long)nst);
      atomicAdd(p.dbg + 8 * (bid & 63) + 6, 1ull);
      atomicAdd(p.dbg + 8 * (bid & 63) + 7, (unsigned long long)tph[7]);
      // launch-wide skew: earliest/latest wave start and end (100 MHz real-time ticks), in the last 4 debug slots
      const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
      // (real-time counters are per XCD and not synchronised: compare waves of XCC 0 only; HW_REG_XCC_ID = hwreg 20)
      if (p.dbg_tag) {   // per XCC: slots 448 + 4*xcc + {0: min start, 1: max start, 2: min end, 3: max end}
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
        unsigned long long* q = p.dbg + 520 + 4 * xcc;
        atomicMin(q, rt_entry); atomicMax(q + 1, rt_entry); atomicMin(q + 2, rt_end); atomicMax(q + 3, rt_end);
      }
      // whole-wave duration in 100 MHz real-time ticks and in shader cycles -> effective shader clock
      atomicAdd(p.dbg + 8 * (bid & 63) + 4, ((__builtin_amdgcn_s_memrealtime() - rt_entry) << 32));
      atomicAdd(p.dbg + 8 * (bid & 63) + 6, ((unsigned long long)((unsigned)__builtin_readcyclecounter() - t_entry)) << 20);
    }
  }
#undef SDRFM_TICK
}

template <int T, int D, int R, int TA, int DA, int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 3))) k_fastb(CallParams p) {
  fastb_body<T, D, R, TA, DA, MODE>(p, blockIdx.x);
}

}  // namespace

#endif
