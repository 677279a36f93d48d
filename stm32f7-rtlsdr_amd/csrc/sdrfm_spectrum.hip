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
// Up to 1024 points a second kernel serves the call, k_spectrum_chain (below): no barrier in the frame loop at all — the running sum is handed
// from wave to wave —, raw dword loads, the first pass fed from registers; k_spectrum serves those lengths only when iq or iq_stride is odd.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/sdrfm.h"

typedef float sf2_t __attribute__((ext_vector_type(2)));
typedef int si4_t __attribute__((ext_vector_type(4)));
typedef float sf4_t __attribute__((ext_vector_type(4)));
__device__ sf2_t spec_typed_load_xy(si4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
__device__ int spec_raw_load_dword(si4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
// buffer resource word3: dst_sel = (R, G, 0, 1), num_format = USCALED (2), data_format = 8_8 (3)
#define SDRFM_SPEC_RSRC_U8X2 (4 | (5 << 3) | (0 << 6) | (1 << 9) | (2 << 12) | (3 << 15))

namespace {

#define SPEC_WAIT_TICKS 200000000ull   /* k_spectrum_chain: s_memrealtime ticks (100 MHz) a wave waits for its predecessor's sum before it gives up: 2 s */
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
  unsigned int* err;      // host-mapped word: k_spectrum_chain sets bit 0 when a wave gave up waiting for its predecessor's sum (the host answers SDRFM_FAIL)
#ifdef SDRFM_DEV
  unsigned int* dbg;      // development library: per-wave cycle sums per phase of k_spectrum_chain (or null)
#endif
};
#ifdef SDRFM_DEV
// phase stamps (development library only): s_memtime at the phase boundaries of k_spectrum_chain, summed per wave.  A stamp waits for the
// wave's outstanding LDS operations, so a phase's figure includes the drain of what it issued.
struct Stamps { unsigned int acc[8]; unsigned int last; };
__device__ __forceinline__ void stamp(Stamps& st, int i) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned int now = (unsigned int)__builtin_amdgcn_s_memtime();
  st.acc[i] += now - st.last;
  st.last = now;
}
#define SPEC_STAMP(i) stamp(*stp, i)
#define SPEC_STAMP_ARG , Stamps* stp
#define SPEC_STAMP_PASS , stp
#else
#define SPEC_STAMP(i)
#define SPEC_STAMP_ARG
#define SPEC_STAMP_PASS
#endif

__device__ __forceinline__ void butterfly(float2& a, float2& b, float2 w) {   // (A, B) -> (A + W B, A - W B)
  const float tr = __builtin_fmaf(w.x, b.x, -(w.y * b.y));
  const float ti = __builtin_fmaf(w.x, b.y, w.y * b.x);
  const float2 A = a;
  a = make_float2(A.x + tr, A.y + ti);
  b = make_float2(A.x - tr, A.y - ti);
}

// byte address within the workgroup's LDS of a pointer into it (what a ds_* instruction written in inline asm takes)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
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
// PO != nullptr (last pass only): the powers stay in registers.  TR != nullptr: the pass's twiddles come from registers (filled once per
// kernel in the order TR[(G - 1) group + 2^t - 1 + j]) instead of the LDS table.  TC != nullptr (one group per lane, H < 64): they come from a
// table of this pass alone, TC[(2^t - 1 + j) H] for the lane's pos (TC points at its column): consecutive lanes read consecutive entries
// whatever N is (the N/2-entry table is read at strides of N / 2^(S+t); measured for N = 512: 117.5 -> 107 us with the compact table).
template <int LOGB, int LOGN, int S, int K>
__device__ __forceinline__ void fft_pass(float2* X, const float2* TW, const float2 (&W1)[8], float* PWO, int lane SPEC_STAMP_ARG, float* PO = nullptr, const float2* TR = nullptr, const float2* TC = nullptr) {
  constexpr int H = 1 << (S - 1), G = 1 << K, NG = (1 << LOGB) >> K;
  // (4096 points: 16 points x 4 groups unrolled would not fit in VGPRs; PO / TR, LOGB = 10 only: the group loop must unroll completely)
  constexpr int UNR = LOGB >= 12 ? 1 : (LOGB == 10 ? 8 : 4);
#pragma unroll UNR
  for (int g0 = 0; g0 < NG; g0 += 64) {
    const int g = g0 + lane;
    if (NG >= 64 || g < NG) {
      // Padded addresses without per-access arithmetic.  pos = position within the stage-S block, base = first point of the group.  With
      // H >= 64 both are (a constant of g0) + lane; with H < 64 there is one group per lane.  spad() is additive over the pieces used below
      // (a piece that is a multiple of 16 / 256 does not carry into the >> 4 / >> 8 terms of the other: see the asserts), so a point is at
      // sb + spad(c H) and a twiddle at tb[t] + spad(constant): one address register per pass and per stage, immediate offsets for the rest.
      static_assert(H >= 16 || H * G <= 16, "c H must not carry into bit 4 of the padded index");
      static_assert(H >= 256 || H * G <= 256, "c H must not carry into bit 8 of the padded index");
      static_assert((S == 1 && K == 4) || LOGN - K >= 4, "twiddle strides below 16 would carry (the first pass takes its twiddles from registers)");
      const int pos0 = H >= 64 ? lane : (g & (H - 1)), posc = H >= 64 ? (g0 & (H - 1)) : 0;
      const int pos = pos0 + posc;
      const int basec = H >= 64 ? ((g0 >> (S - 1)) << (S - 1 + K)) + posc : 0;
      const int base = H >= 64 ? basec + lane : ((g >> (S - 1)) << (S - 1 + K)) + pos;
      const int sb = H >= 64 ? spad(lane) + spad(basec) : spad(base);
      float2 v[G];
#pragma unroll
      for (int c = 0; c < G; ++c) v[c] = X[sb + spad(c * H)];
      if constexpr (S == 5) { SPEC_STAMP(2); }
#pragma unroll
      for (int t = 0; t < K; ++t) {                            // stage S + t: half = H 2^t, partner c ^ 2^t
        const int tb = spad(pos0 << (LOGN - S - t));
#pragma unroll
        for (int c = 0; c < G; ++c) {
          if ((c >> t) & 1) continue;
          const int jc = posc + (c & ((1 << t) - 1)) * H;      // pos_t = pos0 + jc: position of the pair within its stage-(S+t) block
          if constexpr (S == 1 && K == 4) butterfly(v[c], v[c + (1 << t)], W1[(c & ((1 << t) - 1)) << (3 - t)]);   // first pass: the 8 twiddles W16^k, wave-uniform registers
          else if (TR) butterfly(v[c], v[c + (1 << t)], TR[(g0 >> 6) * (G - 1) + (1 << t) - 1 + (c & ((1 << t) - 1))]);   // twiddles kept in registers
          else if (TC) butterfly(v[c], v[c + (1 << t)], TC[((1 << t) - 1 + (c & ((1 << t) - 1))) * H]);   // compact table of this pass: TC[(2^t - 1 + j) H + pos]
          else butterfly(v[c], v[c + (1 << t)], TW[tb + spad(jc << (LOGN - S - t))]);
        }
      }
      if (S + K - 1 == LOGN && PO) {                           // the powers stay in registers: PO[G (g0 / 64) + c] = |point base + c H|^2
#pragma unroll
        for (int c = 0; c < G; ++c) PO[(g0 >> 6) * G + c] = __builtin_fmaf(v[c].x, v[c].x, v[c].y * v[c].y);
      } else if (S + K - 1 == LOGN && PWO) {
#pragma unroll
        for (int c = 0; c < G; ++c) PWO[base + c * H] = __builtin_fmaf(v[c].x, v[c].x, v[c].y * v[c].y);
      } else {
#pragma unroll
        for (int c = 0; c < G; ++c) X[sb + spad(c * H)] = v[c];
      }
    }
    if constexpr (LOGB == 11) __builtin_amdgcn_sched_barrier(0);   // (2048 points: one group's loads and butterflies at a time: 160 -> 143 - 154 us)
  }
  wave_sync();
  SPEC_STAMP(S == 5 ? 3 : 5);
}
template <int LOGB, int LOGN, int S>
__device__ __forceinline__ void fft_passes(float2* X, const float2* TW, const float2 (&W1)[8], float* PWO, int lane SPEC_STAMP_ARG, float* PO = nullptr) {
  if constexpr (S <= LOGN) {
    // up to 4 stages per pass, but no more than leaves a group (2^K points) for each of the 64 lanes
    constexpr int KMAX = (LOGB - 6) >= 4 ? 4 : ((LOGB - 6) >= 2 ? (LOGB - 6) : 2);
    constexpr int K = (LOGN - S + 1) >= KMAX ? KMAX : (LOGN - S + 1);
    fft_pass<LOGB, LOGN, S, K>(X, TW, W1, PWO, lane SPEC_STAMP_PASS, PO);
    fft_passes<LOGB, LOGN, S + K>(X, TW, W1, PWO, lane SPEC_STAMP_PASS, PO);
  }
}

// First pass (stages 1..4) straight from registers: a lane holds the 16 points at bit-reversed-order positions 16 g .. 16 g + 15 of the
// wave's 1024-point block, g = spec_group_of_lane (loaded from global memory in exactly that pattern).  Saves the scatter store and the
// first pass's reads (16 + 16 LDS accesses of 8 bytes per lane and frame).
// Which 16-point group a lane takes: group g of a frame needs the samples brev4(c) N/16 + brev(g), c = 0..15, so lane l takes the group whose
// bit-reversed index is l (within its frame, for blocks of several frames): register c of the 64 lanes is then 64 CONSECUTIVE samples in lane
// order — one fully coalesced 128-byte request.  (With group = lane the same 64 samples arrive in bit-reversed lane order, and the vector L1
// works through such a request a few lanes at a time: the loads of a round were still outstanding a whole round later.)
__host__ __device__ constexpr int spec_brev4(int q) { return ((q & 1) << 3) | ((q & 2) << 1) | ((q & 4) >> 1) | ((q & 8) >> 3); }
template <int LOGN>
__device__ __forceinline__ int spec_group_of_lane(int lane) {
  constexpr int LG = LOGN >= 10 ? 6 : LOGN - 4;                // log2(groups per frame), at most the 64 of a wave
  return (lane & ~((1 << LG) - 1)) | (int)(__brev((uint32_t)(lane & ((1 << LG) - 1))) >> (32 - LG));
}
__device__ __forceinline__ void fft_first_pass_from_regs(float2 (&v)[16], float2* X, const float2 (&W1)[8], int grp16) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if ((c >> t) & 1) continue;
      butterfly(v[c], v[c + (1 << t)], W1[(c & ((1 << t) - 1)) << (3 - t)]);
    }
#pragma unroll
  for (int c = 0; c < 16; ++c) X[spad(16 * grp16 + c)] = v[c];
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
  constexpr bool REGS = LOGN == 9 || LOGN == 10;
  static_assert(!REGS || PPL == 16, "the register path holds one 16-point first-pass group per lane");
  // (REGS) register q of a lane holds the point at bit-reversed-order position u = 16 group + q of the block = sample regs_sample(q)
  auto regs_sample = [&](int q) -> int {
    const uint32_t u = (uint32_t)(16 * spec_group_of_lane<LOGN>(lane) + q);
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
#ifdef SDRFM_DEV
  Stamps stv; Stamps* stp = &stv;                              // (the stamps of the shared passes go nowhere in this kernel)
  for (int i = 0; i < 8; ++i) stv.acc[i] = 0;
  stv.last = 0;
#endif
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
        fft_first_pass_from_regs(v1, X, W1, spec_group_of_lane<LOGN>(lane));
        fft_passes<LOGB, LOGN, 5>(X, TW, W1, FUSEP ? PWsep + par * (NWF * B) + wv * B : nullptr, lane SPEC_STAMP_PASS);
      } else {
        wave_sync();
        fft_passes<LOGB, LOGN, 1>(X, TW, W1, FUSEP ? PWsep + wv * B : nullptr, lane SPEC_STAMP_PASS);
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
    const float* const PWr = PW + (FUSEP ? par * (NWF * B) : 0u);
    auto pw_at = [&](uint32_t o, int q) -> float { return PWr[PWS * (o / FPW) + (o % FPW) * N + (uint32_t)((tid + NT * q) & (N - 1))]; };
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
    if constexpr (!FUSEP) __syncthreads();        // the blocks are rewritten by the next round (FUSEP: the next round writes
                                                               // the OTHER power region; the barrier above orders this sum before the round after)
  }
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    const int k = tid + NT * q;
    if (k < N) p.power[(size_t)stream * p.power_stride + (size_t)((k + N / 2) & (N - 1))] = S[q] * p.inv_frames;
  }
}

// The kernel of 64 .. 1024 points: THE RUNNING SUM TRAVELS, NO WORKGROUP BARRIER IN THE FRAME LOOP.
// k_spectrum's waves meet at a barrier every round to add their power rows, so they move in lock step — all of them load, all of them
// transform, all of them wait for LDS at the same time (measured: ~3300 of 7300 cycles per round spent issuing the sample loads, 1350 at the
// barrier) — and every round ends with a tail in which the SIMDs drain.  Here a wave takes RUNS of R consecutive blocks of 1024 points
// (a frame, or two 512-point frames side by side): run r = w, w + NWF, ...  The last pass leaves a block's powers in registers (lane l: the
// bins l + 64 i + 256 c of its frame(s)); after its run the wave fetches the running sum after run r - 1 from one of two LDS slots (written
// by the wave before it, a per-lane tag written last), adds its frames to it IN FRAME ORDER — S = S + P, the spec's sequential sum; the
// same lane holds the same bins in every wave — and hands it on; the wave with the last run scales and stores it.  The hand-over is one
// serial chain through the workgroup (~1100-1400 cycles per hop under load: two LDS round trips and the adds), so the waves spread over
// the phases of a frame by construction: no barrier tail, no all-to-all of power rows.  R = 1 is bound by that chain (234 hops: 110 us),
// R = 2 is not (89 us); R = 3 gains nothing more and costs 16 registers.  Two slots suffice: the sum after run r + 2 is written by a wave
// that has read slot (r + 1) & 1, which was written after run r + 1's wave had read slot r & 1 completely.
// Samples arrive as raw dwords (lane pairs share one: `buffer_load_dword`, 16 per block and lane at immediate offsets from one address)
// and are split with v_cvt_f32_ubyte: a typed 2-byte load costs ~3x the issue time.  Needs iq and iq_stride even (else k_spectrum).
// The window values of a lane's 16 samples and the last pass's twiddles are the same for every block: LDS (4 x ds_read_b128) / registers.
// 64 .. 256 points (SMALL, 4 .. 16 frames per block): a frame's bins end up spread over a quarter of the lanes, so a run's powers go through the
// wave's own block once (it is free after the last pass) and come back as the N/64 consecutive bins a lane sums — before the wave waits for the sum.
template <int LOGN, int NWF, int R>
__global__ void __launch_bounds__(64 * NWF) k_spectrum_chain(SParams p) {
  static_assert(LOGN >= 6 && LOGN <= 10, "blocks of 1024 points");
  // SMALL (64 .. 256 points, 4 .. 16 frames per block): the last pass leaves a frame's bins spread over a quarter of the lanes, so the block's
  // powers go through the wave's own (by then free) LDS block once — rows [frame][bin] — and come back as the N/64 consecutive bins a lane sums
  constexpr bool SMALL = LOGN <= 8;
  constexpr int N = 1 << LOGN, NT = 64 * NWF, LOGB = 10, B = 1 << LOGB, FPW = B / N, SV = SMALL ? 4 : 16 / FPW;   // SV: floats per lane of a hand-over slot
  constexpr int NBL = SMALL ? N / 64 : SV;                     // bins a lane sums
  static_assert(!SMALL || R * B * 4 <= (spad(B - 1) + 1) * 8, "(SMALL) the run's power rows must fit the wave's own block");
  extern __shared__ __attribute__((aligned(16))) unsigned char spec_smem[];
  constexpr int NPT = spad(N / 2 - 1) + 1, NPX = spad(B - 1) + 1;
  float2* TW = reinterpret_cast<float2*>(spec_smem);
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float2* X = reinterpret_cast<float2*>(spec_smem) + NPT + wv * NPX;
  // (16-byte aligned — NPT is odd for 512 points: a ds_read_b128 / ds_write_b128 at an address that is only 8-byte aligned is legal and costs
  //  several times its cycles: 107 -> 82 us for the 512-point launch when the slots, the window values and the tags moved up by 8 bytes)
  float4* const SL = reinterpret_cast<float4*>(reinterpret_cast<float2*>(spec_smem) + ((NPT + NWF * NPX + 1) & ~1));   // slot s: SL[(s SV/4 + j) 64 + lane]
  uint32_t* const TAG = reinterpret_cast<uint32_t*>(SL + 2 * (SV / 4) * 64);   // TAG[64 s + lane] = 1 + the block whose sum slot s holds (one word per lane)
  const uint32_t stream = blockIdx.x;
  for (int i = tid; i < N / 2; i += NT) TW[spad(i)] = p.tw[i];
  if (tid < 128) TAG[tid] = 0u;
  float2 W1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) W1[k] = p.tw[k * (N / 16)];
  const int grp = spec_group_of_lane<LOGN>(lane);
  auto regs_sample = [&](int q) -> int {
    const uint32_t u = (uint32_t)(16 * grp + q);
    return (int)((u & ~(uint32_t)(N - 1)) | (__brev(u & (uint32_t)(N - 1)) >> (32 - LOGN)));
  };
  // the 16 window values of a lane's samples: the same in every wave (and block), kept in LDS as WL[j 64 + lane] (4 x ds_read_b128 per block)
  float4* const WL = reinterpret_cast<float4*>(TAG + 128);
  float2* const T2 = reinterpret_cast<float2*>(WL + 4 * 64);   // (512 points) the second pass's twiddles: T2[(2^t - 1 + j) 16 + pos] = tw[(pos + 16 j) N / 2^(5+t)]
  if constexpr (LOGN == 9) {                                   // (1024 points: no gain, and the extra address register spills)
    for (int i = tid; i < 15 * 16; i += NT) {
      const int e = i >> 4, pos = i & 15, t = 31 - __builtin_clz((unsigned)(e + 1)), j = e + 1 - (1 << t);
      T2[i] = p.tw[(pos + 16 * j) << (LOGN - 5 - t)];
    }
  }
  if (wv == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      WL[j * 64 + lane] = make_float4(p.win[regs_sample(4 * j) & (N - 1)], p.win[regs_sample(4 * j + 1) & (N - 1)],
                                      p.win[regs_sample(4 * j + 2) & (N - 1)], p.win[regs_sample(4 * j + 3) & (N - 1)]);
  }
  const unsigned long long ga = (unsigned long long)p.iq, gaw = ga & ~3ull;
  const uint32_t adj = (uint32_t)(ga & 3ull);                  // 0 or 2
  // (the range covers whole dwords: with iq = 2 mod 4 the first and the last sample of the batch share theirs with two bytes outside it)
  const si4_t rsrcw = {(int)(unsigned)gaw, (int)(unsigned)(gaw >> 32), (int)((p.iq_span + adj + 3u) & ~3u), 0x00020000};
  const uint32_t sbase = stream * (uint32_t)p.iq_stride;
  const uint32_t shw = ((adj + sbase + 2u * (uint32_t)regs_sample(0)) & 2u) << 3;   // which half of its dword a lane's samples are
  int curw[16];
  auto fetch = [&](uint32_t b) {                               // block b -> curw (past the batch's span: zeros, never used)
    const uint32_t v0 = (adj + sbase + 2u * (b * (uint32_t)B + (uint32_t)regs_sample(0))) & ~3u;
#pragma unroll
    for (int q = 0; q < 16; ++q) curw[q] = spec_raw_load_dword(rsrcw, (int)(v0 + 2u * (uint32_t)(spec_brev4(q) * (N / 16))), 0, 0);
  };
  const uint32_t NB = (p.F + (uint32_t)FPW - 1u) / (uint32_t)FPW;   // blocks (the last one may hold a frame past F: computed, not summed)
  const uint32_t NR = (NB + (uint32_t)R - 1u) / (uint32_t)R;        // runs of R consecutive blocks: one hand-over of the running sum per run
  fetch((uint32_t)(wv * R));
  __syncthreads();                                             // TW, TAG and WL visible; the only workgroup barrier
#ifdef SDRFM_DEV
  Stamps stv; Stamps* stp = &stv;
  for (int i = 0; i < 8; ++i) stv.acc[i] = 0;
  stv.last = (unsigned int)__builtin_amdgcn_s_memtime();
#endif
  constexpr int K3 = SMALL ? 1 : LOGN - 8, G3 = 1 << K3, NI3 = (B >> K3) / 64;   // (512 / 1024 points) the last pass: stages 9 .. LOGN, NI3 groups per lane
  float2 TR3[NI3 * (G3 - 1)];                                  // its twiddles: the same for every block, kept in registers
  if constexpr (!SMALL) {
#pragma unroll
    for (int i = 0; i < NI3; ++i)
#pragma unroll
      for (int t = 0; t < K3; ++t)
#pragma unroll
        for (int j = 0; j < (1 << t); ++j) {
          const int posc = (64 * i) & 255, jc = posc + j * 256, sh = LOGN - 9 - t;
          TR3[i * (G3 - 1) + (1 << t) - 1 + j] = TW[spad(lane << sh) + spad(jc << sh)];
        }
  }
  for (uint32_t r = (uint32_t)wv; r < NR; r += (uint32_t)NWF) {
    float PO[R][16];
#pragma unroll
    for (int jr = 0; jr < R; ++jr) {
      const uint32_t b = r * (uint32_t)R + (uint32_t)jr;
      if (b < NB) {                                            // (wave-uniform)
        float2 v1[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 wn4 = WL[j * 64 + lane];
          const float wn[4] = {wn4.x, wn4.y, wn4.z, wn4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t w = (uint32_t)curw[4 * j + q] >> shw;
            v1[4 * j + q] = make_float2(((float)(w & 0xffu) - 127.5f) * wn[q], ((float)((w >> 8) & 0xffu) - 127.5f) * wn[q]);
          }
        }
        fetch(jr + 1 < R ? b + 1u : (r + (uint32_t)NWF) * (uint32_t)R);   // this wave's next block: in flight during this one's passes
        SPEC_STAMP(4);
        fft_first_pass_from_regs(v1, X, W1, grp);
        SPEC_STAMP(1);
        if constexpr (SMALL) {
          fft_pass<LOGB, LOGN, 5, LOGN - 4>(X, TW, W1, nullptr, lane SPEC_STAMP_PASS, PO[jr]);   // stages 5 .. LOGN: the last pass
        } else {
          fft_pass<LOGB, LOGN, 5, 4>(X, TW, W1, nullptr, lane SPEC_STAMP_PASS, nullptr, nullptr, LOGN == 9 ? T2 + (lane & 15) : nullptr);
          fft_pass<LOGB, LOGN, 9, K3>(X, TW, W1, nullptr, lane SPEC_STAMP_PASS, PO[jr], TR3);
        }
      }
    }
    // the running sum after run r - 1: ONE batch of LDS reads, the slot's tag first — DS operations of a wave execute in order and the
    // writer stores the tag last, so a batch that sees the tag sees the sum (a batch that does not is repeated).  While the predecessor
    // itself is still waiting (tag < r - 1) only the tag is watched, one cheap read per 64 cycles.
    float S[SV];
    float PV[SMALL ? R * FPW : 1][NBL];                        // (SMALL) the run's powers of this lane's bins, frame by frame
    if constexpr (SMALL) {
      // PO[jr][G i + c] = frame (lane >> 4) + 4 i of block jr, bin (lane & 15) + 16 c  ->  row-major rows in the wave's block (free now)  ->  PV
      constexpr int K = LOGN - 4, G = 1 << K, NI = (B >> K) / 64;
      float* const PR = reinterpret_cast<float*>(X);
#pragma unroll
      for (int jr = 0; jr < R; ++jr)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int c = 0; c < G; ++c) PR[jr * B + (((lane >> 4) + 4 * i) << LOGN) + (lane & 15) + 16 * c] = PO[jr][G * i + c];
      wave_sync();
#pragma unroll
      for (int jr = 0; jr < R; ++jr)
#pragma unroll
        for (int f = 0; f < FPW; ++f)
#pragma unroll
          for (int j = 0; j < NBL; ++j) PV[jr * FPW + f][j] = PR[jr * B + (f << LOGN) + NBL * lane + j];
      wave_sync();                                             // (the next run's first pass writes the block)
    }
    __builtin_amdgcn_s_setprio(3);                             // the hand-over is the one serial chain through the workgroup
    if (r > 0u) {
      const uint32_t sl = (r - 1u) & 1u;
      const uint32_t a_tag = lds_addr(TAG + sl * 64 + lane), a_dat = lds_addr(SL + sl * (SV / 4) * 64 + lane);
      const uint32_t a_tagp = lds_addr(TAG + (sl ^ 1u) * 64 + lane);
      uint32_t tag;
      sf4_t t[SV / 4];
      auto batch = [&]() {
        if constexpr (SV == 4)
          asm volatile("ds_read_b32 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(tag), "=&v"(t[0]) : "v"(a_tag), "v"(a_dat) : "memory");
        else if constexpr (SV == 16)
          asm volatile("ds_read_b32 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:1024\n\tds_read_b128 %3, %6 offset:2048\n\t"
                       "ds_read_b128 %4, %6 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(tag), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[SV / 4 - 2]), "=&v"(t[SV / 4 - 1]) : "v"(a_tag), "v"(a_dat) : "memory");
        else
          asm volatile("ds_read_b32 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(tag), "=&v"(t[0]), "=&v"(t[1]) : "v"(a_tag), "v"(a_dat) : "memory");
      };
      batch();
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)tag) != r) {
        // A wave never waits longer than one turn of the chain, ~100 us whatever F is.  The wait is bounded in REAL TIME (s_memrealtime, 100 MHz:
        // two seconds — a predecessor parked by a debugger or by wave pre-emption does not run the budget down the way a poll count did), and a wave
        // whose budget runs out neither hangs the device nor traps (a trap takes the whole HIP context and every handle of the process with it): it
        // sets the handle's error word, goes on with whatever the slot holds and publishes its own tag as usual, so its successors are not stuck
        // behind it; the host answers SDRFM_FAIL at the next synchronisation (the powers of that call are not valid).
        // (The protocol rests on two properties of the LDS: the DS operations of ONE wave execute in order — the writer stores its 64 lanes' sums,
        // then their tags, in one instruction sequence — and a DS instruction is performed for all 64 lanes before the next one of any wave touches
        // the same words; that is why lane 0's tag stands for all 64.)
        const unsigned long long t_wait0 = __builtin_amdgcn_s_memrealtime();
        bool gave_up = false;
        if (r > 1u) {                                          // far from the head of the chain: wait until the predecessor has its input
          uint32_t tp;
          for (;;) {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(tp) : "v"(a_tagp) : "memory");
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)tp) >= r - 1u) break;
            if (__builtin_amdgcn_s_memrealtime() - t_wait0 > SPEC_WAIT_TICKS) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        if (!gave_up) {
          uint32_t polls = 0;
          do {                                                 // next in line: whole batches, one round trip after the tag lands
            batch();
            if ((++polls & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t_wait0 > SPEC_WAIT_TICKS) { gave_up = true; break; }
          } while ((uint32_t)__builtin_amdgcn_readfirstlane((int)tag) != r);
        }
        if (gave_up && lane == 0 && p.err) __hip_atomic_fetch_or(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#pragma unroll
      for (int j = 0; j < SV / 4; ++j) { S[4 * j] = t[j].x; S[4 * j + 1] = t[j].y; S[4 * j + 2] = t[j].z; S[4 * j + 3] = t[j].w; }
      SPEC_STAMP(6);
    } else {
#pragma unroll
      for (int j = 0; j < SV; ++j) S[j] = 0.0f;
    }
    // N = 1024: PO[.][4 i + c] = bin lane + 64 i + 256 c.  N = 512: PO[.][2 i + c], i < 4: first frame's bin lane + 64 i + 256 c; i >= 4: second frame's
    if constexpr (SMALL) {
#pragma unroll
      for (int jr = 0; jr < R; ++jr)
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
          if ((r * (uint32_t)R + (uint32_t)jr) * (uint32_t)FPW + (uint32_t)f < p.F) {   // (wave-uniform; frames past F: computed, not summed)
#pragma unroll
            for (int j = 0; j < NBL; ++j) S[j] = S[j] + PV[jr * FPW + f][j];
          }
        }
    } else
#pragma unroll
    for (int jr = 0; jr < R; ++jr) {
      const uint32_t b = r * (uint32_t)R + (uint32_t)jr;
      if (b < NB) {
#pragma unroll
        for (int j = 0; j < SV; ++j) S[j] = S[j] + PO[jr][j];
        if constexpr (FPW == 2) {
          if (b * 2u + 1u < p.F) {
#pragma unroll
            for (int j = 0; j < SV; ++j) S[j] = S[j] + PO[jr][SV + j];
          }
        }
      }
    }
    if (r + 1u < NR) {
      const uint32_t sl = r & 1u;
      const uint32_t a_tag = lds_addr(TAG + sl * 64 + lane), a_dat = lds_addr(SL + sl * (SV / 4) * 64 + lane);
      sf4_t t[SV / 4];
#pragma unroll
      for (int j = 0; j < SV / 4; ++j) { t[j].x = S[4 * j]; t[j].y = S[4 * j + 1]; t[j].z = S[4 * j + 2]; t[j].w = S[4 * j + 3]; }
      const uint32_t tag = r + 1u;
      if constexpr (SV == 4)
        asm volatile("ds_write_b128 %1, %2\n\tds_write_b32 %0, %3" :: "v"(a_tag), "v"(a_dat), "v"(t[0]), "v"(tag) : "memory");
      else if constexpr (SV == 16)
        asm volatile("ds_write_b128 %1, %2\n\tds_write_b128 %1, %3 offset:1024\n\tds_write_b128 %1, %4 offset:2048\n\tds_write_b128 %1, %5 offset:3072\n\t"
                     "ds_write_b32 %0, %6" :: "v"(a_tag), "v"(a_dat), "v"(t[0]), "v"(t[1]), "v"(t[SV / 4 - 2]), "v"(t[SV / 4 - 1]), "v"(tag) : "memory");
      else
        asm volatile("ds_write_b128 %1, %2\n\tds_write_b128 %1, %3 offset:1024\n\tds_write_b32 %0, %4"
                     :: "v"(a_tag), "v"(a_dat), "v"(t[0]), "v"(t[1]), "v"(tag) : "memory");
      __builtin_amdgcn_s_setprio(0);
      SPEC_STAMP(7);
    } else {
      constexpr int GP = FPW == 1 ? 4 : 2;                     // points per group of the last pass
#pragma unroll
      for (int j = 0; j < NBL; ++j) {
        const int k = SMALL ? NBL * lane + j : lane + 64 * (j / GP) + 256 * (j % GP);
        p.power[(size_t)stream * p.power_stride + (size_t)((k + N / 2) & (N - 1))] = S[j] * p.inv_frames;
      }
    }
  }
#ifdef SDRFM_DEV
  if (p.dbg && lane == 0) for (int i = 0; i < 8; ++i) p.dbg[(stream * NWF + wv) * 8 + i] = stv.acc[i];
#endif
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
  spec_kernel_t kernel;     // k_spectrum<log2 N>: every N; for N = 512 / 1024 only when iq or iq_stride is odd
  size_t lds_bytes;
  spec_kernel_t chain;      // k_spectrum_chain (N = 512 / 1024, iq and iq_stride even) or null
  size_t chain_lds_bytes;
  int chain_nwf;            // its waves per workgroup
  char name[2][48];         // kernel names as the profiler prints them: [0] k_spectrum<..>, [1] k_spectrum_chain<..>
  int last;                 // which of the two the last call launched
  unsigned int* err_host;   // host-mapped error word (SParams.err) and its device address
  unsigned int* err_dev;
#ifdef SDRFM_DEV
  unsigned int* d_dbg;      // phase stamps (SDRFM_SPEC_STAMPS=1)
  int stamps;
#endif
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
#ifdef SDRFM_DEV
  if (h->d_dbg) {
    (void)hipDeviceSynchronize();
    const uint32_t nw = h->cfg.n_streams * (uint32_t)h->chain_nwf;
    unsigned int* hb = (unsigned int*)malloc((size_t)nw * 8 * sizeof(unsigned int));
    (void)hipMemcpy(hb, h->d_dbg, (size_t)nw * 8 * sizeof(unsigned int), hipMemcpyDeviceToHost);
    double sum[8] = {0};
    for (uint32_t w = 0; w < nw; ++w) for (int i = 0; i < 8; ++i) sum[i] += hb[w * 8 + i];
    // [4] samples -> points, next block's loads issued  [1] first pass  [2] second pass: points read  [3] its butterflies and write-back
    // [5] last pass  [6] waiting for / fetching the running sum  [7] adds and hand-over   (shader cycles, summed over the wave's blocks)
    fprintf(stderr, "[spec stamps] k_spectrum_chain, %d waves per workgroup: mean shader cycles per wave in the last launch:", h->chain_nwf);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " [%d] %.0f", i, sum[i] / nw);
    fprintf(stderr, "\n");
    free(hb);
    (void)hipFree(h->d_dbg);
  }
#endif
  void* ptrs[] = {h->d_tw, h->d_win, h->d_iq, h->d_power};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  if (h->err_host) (void)hipHostFree(h->err_host);
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
  const size_t npt = (size_t)(spad((int)cfg->nfft / 2 - 1) + 1), npx = (size_t)(spad((1 << spec_logb((int)logn)) - 1) + 1);
  h->lds_bytes = (npt + (size_t)spec_nwf((int)logn) * npx) * sizeof(float2) +
                 (spec_fusep((int)logn) ? 2 * (size_t)spec_nwf((int)logn) * ((size_t)1 << spec_logb((int)logn)) * sizeof(float) : 0);   // two power regions
  snprintf(h->name[0], sizeof(h->name[0]), "k_spectrum<%u>", logn);
  if (logn <= 10) {                                            // 12 waves per workgroup, runs of 2 blocks (see k_spectrum_chain)
    h->chain_nwf = 12;
    h->chain = logn == 6 ? k_spectrum_chain<6, 12, 2> : logn == 7 ? k_spectrum_chain<7, 12, 2> : logn == 8 ? k_spectrum_chain<8, 12, 2>
             : logn == 9 ? k_spectrum_chain<9, 12, 2> : k_spectrum_chain<10, 12, 2>;
#ifdef SDRFM_DEV
    // development library: SDRFM_SPEC_VARIANT = waves per workgroup x 10 + blocks per run (0: k_spectrum), SDRFM_SPEC_STAMPS=1: phase stamps
    const char* var_ = getenv("SDRFM_SPEC_VARIANT");
    const int variant = var_ ? atoi(var_) : 122, nwf = variant / 10, rr = variant % 10;
    h->stamps = getenv("SDRFM_SPEC_STAMPS") != nullptr;
    if (variant == 0) h->chain = nullptr;
#define CK(NW, RR) if (nwf == NW && rr == RR && logn >= 9) { h->chain = logn == 9 ? k_spectrum_chain<9, NW, RR> : k_spectrum_chain<10, NW, RR>; h->chain_nwf = NW; }
    if (logn <= 8 && nwf == 16 && rr == 2) { h->chain = logn == 6 ? k_spectrum_chain<6, 16, 2> : logn == 7 ? k_spectrum_chain<7, 16, 2> : k_spectrum_chain<8, 16, 2>; h->chain_nwf = 16; }
    if (logn <= 8 && nwf == 16 && rr == 1) { h->chain = logn == 6 ? k_spectrum_chain<6, 16, 1> : logn == 7 ? k_spectrum_chain<7, 16, 1> : k_spectrum_chain<8, 16, 1>; h->chain_nwf = 16; }
    CK(8, 1) CK(8, 2) CK(8, 3) CK(12, 1) CK(12, 3) CK(16, 1)
#undef CK
#endif
    int rr_ = 2;
#ifdef SDRFM_DEV
    if (h->chain && var_ && variant) rr_ = rr;
#endif
    snprintf(h->name[1], sizeof(h->name[1]), "k_spectrum_chain<%u, %d, %d>", logn, h->chain_nwf, rr_);
    h->chain_lds_bytes = (((npt + (size_t)h->chain_nwf * npx + 1) & ~(size_t)1)) * sizeof(float2) + 2 * 4096 + 512 + 4096 + 1920;   // twiddles, blocks, two sum slots, their tags, the lanes' window values, the second pass's twiddles
  }
#define CR(expr) do { if ((expr) != hipSuccess) { sfree(h); return SDRFM_ENOMEM; } } while (0)
  CR(hipSetDevice(h->device));
  if ((h->lds_bytes > 64 * 1024 &&
       hipFuncSetAttribute(reinterpret_cast<const void*>(h->kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) ||
      (h->chain && h->chain_lds_bytes > 64 * 1024 &&
       hipFuncSetAttribute(reinterpret_cast<const void*>(h->chain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->chain_lds_bytes) != hipSuccess)) {
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
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->err_host), 64, hipHostMallocMapped);
  if (e == hipSuccess) { *h->err_host = 0u; e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->err_dev), h->err_host, 0); }
  free(tw); free(win);
  CR(e);
#undef CR
  h->last = h->chain ? 1 : 0;
  *out = h;
  return SDRFM_OK;
}

void sdrfm_spectrum_destroy(sdrfm_spectrum_t* h) { sfree(h); }

const char* sdrfm_spectrum_kernel_name(const sdrfm_spectrum_t* h) { return h ? h->name[h->last] : ""; }

int sdrfm_spectrum_set_stream(sdrfm_spectrum_t* h, void* hip_stream) {
  if (!h) return SDRFM_EINVAL;
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return SDRFM_OK;
}

// the chain kernel's error word, read (and cleared) behind a synchronisation: a wave gave up waiting for its predecessor — the call's powers are not valid
static int spec_take_error(sdrfm_spectrum* h) {
  if (!h->err_host || !*reinterpret_cast<volatile unsigned int*>(h->err_host)) return SDRFM_OK;
  *reinterpret_cast<volatile unsigned int*>(h->err_host) = 0u;
  return SDRFM_FAIL;
}

int sdrfm_spectrum_synchronize(sdrfm_spectrum_t* h) {
  if (!h) return SDRFM_EINVAL;
  STRY(hipSetDevice(h->device), SDRFM_FAIL);
  STRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return spec_take_error(h);
}

static int senqueue(sdrfm_spectrum* h, const uint8_t* d_iq, size_t iq_stride, uint32_t nbytes, float* d_power, size_t power_stride,
                    uint32_t F) {
  const uint32_t ns = h->cfg.n_streams, N = h->cfg.nfft;
  if (F == 0) {                                               // nothing to view: all zeros
    STRY(hipMemset2DAsync(d_power, power_stride * sizeof(float), 0, N * sizeof(float), ns, h->stream), SDRFM_FAIL);
    return SDRFM_OK;
  }
  const uint64_t span = (uint64_t)(ns - 1) * iq_stride + 2ull * F * N;
  if (span >= (1ull << 32) - 8) return SDRFM_ECAPACITY;       // one buffer descriptor spans the batch (rounded out to whole dwords)
  (void)nbytes;
  SParams p;
  p.iq = d_iq; p.iq_stride = iq_stride; p.iq_span = (uint32_t)span; p.power = d_power; p.power_stride = power_stride;
  p.tw = h->d_tw; p.win = h->d_win; p.F = F; p.inv_frames = 1.0f / (float)F; p.err = h->err_dev;
  // the raw-dword kernel needs every sample pair inside one aligned dword: iq and iq_stride even
  const bool use_chain = h->chain && !(((uintptr_t)d_iq | (uintptr_t)iq_stride) & 1u);
#ifdef SDRFM_DEV
  if (h->stamps && use_chain && !h->d_dbg && hipMalloc(&h->d_dbg, (size_t)ns * 16 * 8 * sizeof(unsigned int)) != hipSuccess) h->d_dbg = nullptr;
  p.dbg = use_chain ? h->d_dbg : nullptr;
#endif
  h->last = use_chain ? 1 : 0;
  if (use_chain) hipLaunchKernelGGL(h->chain, dim3(ns), dim3(64 * h->chain_nwf), h->chain_lds_bytes, h->stream, p);
  else hipLaunchKernelGGL(h->kernel, dim3(ns), dim3(64 * spec_nwf((int)h->logn)), h->lds_bytes, h->stream, p);
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
  return spec_take_error(h);
}

}  // extern "C"
