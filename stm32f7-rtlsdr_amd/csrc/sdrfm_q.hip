/*
 * sdrfm_q.hip — design Q: stage K2 (the decimating channel FIR) on the i8 matrix pipe of gfx950, fused with K1, K3, K4.
 *
 * What it fills in the reference: the same empty consumer hook as the rest of the library (RTLSDR_XFER_COMPLETE,
 * Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Src/usbh_rtlsdr.c:1094-1097); byte format usbh_rtlsdr.h:165-173; FIR
 * convention CMSIS/core/arm_math.h:3291-3331.  Arithmetic: qtaps.c / DESIGN.md §2, §4.Q — NOT bit-identical to the fp32 fmaf
 * chain of the oracle (designs S / B / generic are): K2 is evaluated exactly in integers from taps rounded to 24-bit fixed
 * point, which lands within 1e-4 (absolute) of the chain's y: within 1e-6 of the oracle's audio wherever the phase of
 * y[m] conj(y[m-1]) is well conditioned, and where it is not (a deep fade; a d next to +-pi) the CONDITIONING GUARD lists the
 * output pair and the repair path recomputes it with the definition's own chain from the raw bytes (repair_flagged below).
 * With the default (statistical) radius the 1e-5 tolerance is a MEASURED statement (tests/test_q_guard_gpu.py, tools/fuzz_q.py at the plain
 * criterion: no violation); SDRFM_CFG_GUARD_WORST_CASE derives the radius from the proven worst case of |dy|; the guarantee is
 * SDRFM_CFG_BIT_EXACT (include/sdrfm.h has the three statements side by side).
 * Instances: (D, Da) = (10, 5) — 2.4 MS/s, BASELINE —, (8, 8) — 2.048 MS/s — and (16, 5) — 3.2 MS/s; the step sizes below
 * are the (10, 5) ones (QGeo has the others: steps of 2 / 4 whole KiB chunks, a swizzled ring).
 *
 * Why: designs S / B are VALU-issue-bound at 14.7 vector instructions per sample and lane, 6.4 of them the FIR's FMAs and 3.4
 * the u8 -> f32 conversion (profiles/r02_*).  The i8 matrix pipe takes the raw bytes as they are (one v_xor per 4 bytes) and
 * runs beside the vector pipe (profiles/ubench_r03): the vector pipe is left with the digit recombination (1.2 instructions per
 * sample and lane), the discriminator and the audio FIR.
 *
 * One WAVE per workgroup (no barriers).  A wave owns a RUN of consecutive STEPS of one stream; a step is 16 BLOCKS (the MFMA's
 * 16 columns) of 8 outputs = 128 outputs = 1280 samples = 2560 bytes:
 *
 *   HBM --buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB of consecutive bytes per instruction, whole 128-byte lines)-->
 *       LDS ring of NSLOT KiB, filled NSLOT KiB ahead of the step being computed
 *   LDS --ds_read_b128: lane (column n, K-group g) reads bytes [160 (n-1) + 64 c + 16 g, +16) of the step for chunk c = the
 *       B operand as the MFMA wants it (conflict-free: lane stride 160 B)--> v_xor 0x80808080
 *   3 K-chunks of 128 bytes x 3 digits v_smfmac_i32_16x16x128_i8 (A = tap tables, exactly 2:4 sparse: I rows hold taps at even
 *   bytes, Q rows at odd ones; wave-constant, 36 VGPRs + one index word) --> S0, S1, S2 exact in i32
 *   --> y = fma(f32(S0 + (S1 << 8)), q, fma(f32(S2), 65536 q, 0.5 sum h)): lane (n, g) holds (I, Q) of outputs 8 n + 2 g, +1
 *   --> y[m-1] of the lane's first output from lane - 16 (ds_bpermute) --> K3 twice (scalar code: packed f32 instructions stall
 *       the matrix pipe, profiles/ubench_r03) --> d's into an LDS buffer
 *   every DA = 5 steps (640 d's = 128 audio outputs): K4, two consecutive outputs per lane from 20 aligned 8-byte reads --> parked in LDS
 *   after the run's last step: the parked outputs --> HBM, 8 bytes per lane (no store inside the loop: it would count in vmcnt)
 *
 * A run that does not start its stream first recomputes the step before it ("warm-up": only its last four blocks are fetched
 * and matter) for the 31 d's and the y[m-1] its first audio outputs need; the stream's first run takes them from the carried
 * state — or, for a call made with SDRFM_F_OVERLAP, warms up the same way from the last bytes of the previous call's buffer, so
 * that the call depends on nothing the previous call computes (bit-identical either way: y is exact).  The wave that holds the
 * end of the stream's chunk hands the state over.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "sdrfm_math.h"
#include "sdrfm_q.h"
#include "sdrfm_b.h"

typedef int qi4_t __attribute__((ext_vector_type(4)));
typedef int qi8_t __attribute__((ext_vector_type(8)));
typedef float qf2_t __attribute__((ext_vector_type(2)));   // (LDS reads use ext vectors: a HIP float2 struct load makes the compiler drain vmcnt)
__device__ void q_raw_buffer_load_lds(qi4_t rsrc, __attribute__((address_space(3))) void* lds, int size, int voffset, int soffset,
                                      int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

namespace {

constexpr int QTA = (int)SDRFM_Q_TA;            // audio taps (every instance)
constexpr int DB0 = 128;                        // d buffer: word DB0 + sigma = first d of the current audio stage
constexpr int ABS = 4;                          // audio stages (128 outputs each) parked in LDS before they are stored (2, which lets 16 waves fit a CU: no faster, r06_q_experiments.txt item 5)
constexpr int ABW = 128 * ABS;                  // words, after the d buffer
constexpr int FLW = 64 + 2;                     // words, after the parked audio: up to 128 two-byte entries of lanes waiting for the repair path, two counters
#ifndef SDRFM_Q_GTAB
#define SDRFM_Q_GTAB 1   // 1: the audio taps are read from a table in LDS (one address for the whole wave: broadcasts) into VGPRs at every audio stage — a v_fma_f32
#endif                   // with a tap from an SGPR issues at half rate (tools/ubench, profiles/ubench_r01), and 32 taps kept in SGPRs crowd the loop's scalars out; 0: rounds 3 - 4 (taps in SGPRs)
constexpr int GTW = SDRFM_Q_GTAB ? 2 + (int)SDRFM_Q_TA : 0;    // words, after the list (16-byte aligned: FLW + 2 is a multiple of 4 words)
typedef float qf4_t __attribute__((ext_vector_type(4)));
constexpr int QTP = (int)SDRFM_Q_TP;            // the repair path's chain length (taps padded with zeros)
// geometry of an instance: FIR decimation D (even), audio decimation DA
template <int D, int DA>
struct QGeo {
  static constexpr int NCH = D / 2;             // 64-byte pieces of a window (two blocks of 16 D bytes): one ds_read_b128 per lane each
  static constexpr int NSC = (D + 3) / 4;       // K-chunks of 128 window bytes: one v_smfmac_i32_16x16x128_i8 per digit each
  static constexpr int BLKB = 16 * D;           // bytes per block of 8 outputs
  static constexpr int STEPB = 16 * BLKB;       // bytes per step
  static constexpr int PRE = BLKB;              // pre-halo: the block before the ring's first byte
  static constexpr int DBW = DB0 + 128 * DA + 16;   // words of the d buffer: history, one audio stage (128 outputs = DA steps), slack
  static constexpr int RWIN = QTP + 2 * D;      // samples under the three outputs y[m0 - 1], y[m0], y[m0 + 1] a repaired lane recomputes
  // Steps that are whole KiB chunks (D = 8, 16): the ring is two steps, refilled a step at a time, and — because a block is then 128 or 256
  // bytes, so that the 16 window reads of a lane group would all fall on the same LDS banks — kept SWIZZLED: ring byte L sits at
  // L ^ (((L / BLKB) & 7) << 4).  The LDS-DMA writes lane l's 16 bytes at chunk + 16 l whatever they are, so the swizzle is applied to the
  // GLOBAL offset a lane fetches (the pieces of a chunk are permuted within the chunk: same lines, same traffic).  D = 10: steps of 2.5
  // chunks, blocks of 160 bytes: conflict-free as they are (the layout of rounds 3 - 4, untouched).
  static constexpr bool ALIGNED = (STEPB % 1024) == 0;
  static constexpr int CS = STEPB / 1024;       // (ALIGNED) chunks per step
  static_assert(QTP % 4 == 0 && (2 * D) % 4 == 0, "the repair path loads whole groups of four samples, none of which straddles the call's first sample");
  static_assert(QTP + 2 * D <= 96, "the tap table of the repair path sits below the d history");
  static_assert(D % 2 == 0 && D <= 16 && (ALIGNED || D == 10), "geometries the ring logic is written for");
};
#ifndef SDRFM_Q_HO_RING
#define SDRFM_Q_HO_RING 1   // 1 (round 6): the input-only part of the state hand-over copied from the ring when the run's last step begins (D = 10), not fetched and stored behind it
#endif
#ifndef SDRFM_Q_K3
#define SDRFM_Q_K3 1    // 1 (round 6): K3 as a difference of angles, the guard's norms taken from the arctangent (q_angle); 0: the conjugate-product form of rounds 3 - 5
#endif
#ifndef SDRFM_Q_XORSKIP
#define SDRFM_Q_XORSKIP 1   // 1 (round 6): D = 10, T <= 64: window bytes 0 .. 51 meet no tap (qtaps.c: tap k = 89 + 10 o - w < 64), so the first dword of a lane's first piece
#endif                      // (bytes 16 g .. 16 g + 3 <= 51) needs no byte - 128: one v_xor less per step
#ifndef SDRFM_Q_PEEL
#define SDRFM_Q_PEEL 1      // 1 (round 6): the run's last step peeled out of the step loop (what only the last step does is compiled out of the others)
#endif
#ifndef SDRFM_Q_MICRO
#define SDRFM_Q_MICRO 1     // 1 (round 6): the carried angle enters lane 0 through v_writelane (one instruction instead of a move and a select); the d write's address
#endif                      // is one shift-add on a pointer kept in a VGPR (the compiler re-added the buffer's offset every step)
#ifndef SDRFM_Q_AUX
#define SDRFM_Q_AUX 2   // cache policy of the ring's fetches: 2 = nt (streamed once; measured 0.4-1 us per launch better than the default policy)
#endif
#ifndef SDRFM_Q_WQB
#define SDRFM_Q_WQB 4   // blocks of a warm-up = the unit runs are cut in (>= 4: QTA - 1 d's and the y before them)
#endif
constexpr int WQB = SDRFM_Q_WQB;
static_assert(WQB >= 4 && WQB <= 16 && 8 * WQB >= (int)SDRFM_Q_TA, "a warm-up holds the d history of a run's first audio output");

// K3 for design Q: the spec's conjugate product (sdrfm_math.h: one fused, two rounded products) and this kernel's own atan2 — the
// same range reduction as sdrfm_atan2f with a shorter minimax polynomial (6 coefficients in s = v^2, |error| <= 3.9e-7 rad
// evaluated in fp32; sdrfm_atan2f: 8 coefficients, 6.5e-8), written for the fewest vector instructions: the vector pipe is what
// bounds this kernel, and the audio stays within 1e-6 of the oracle (tolerance 1e-5).  (0, 0) -> 0 through the clamp of the
// larger magnitude (no select); scalar code on purpose (packed f32 instructions stall the matrix pipe: profiles/ubench_r03).
__device__ __forceinline__ float q_discriminate(float yr, float yi, float pr, float pi) {
  const float re = __builtin_fmaf(yr, pr, yi * pi);
  const float im = yi * pr - yr * pi;
  const float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
  const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 0x1p-120f), mn = __builtin_fminf(ax, ay);   // v_max3_f32, v_min_f32
  const float v = mn * __builtin_amdgcn_rcpf(mx);
  const float s2 = v * v;
  float q = 0x1.e34882p-8f;
  q = __builtin_fmaf(q, s2, -0x1.22fc74p-5f);
  q = __builtin_fmaf(q, s2, 0x1.509024p-4f);
  q = __builtin_fmaf(q, s2, -0x1.12688cp-3f);
  q = __builtin_fmaf(q, s2, 0x1.96c562p-3f);
  q = __builtin_fmaf(q, s2, -0x1.554086p-2f);
  float a = __builtin_fmaf(v, s2 * q, v);
  if (ay > ax) a = 0x1.921fb6p+0f - a;
  if (re < 0.0f) a = 0x1.921fb6p+1f - a;
  return __builtin_copysignf(a, im);
}

// K3 as a DIFFERENCE OF ANGLES (round 6, SDRFM_Q_K3 = 1): theta[m] = atan2(yi, yr) once per output, d[m] = theta[m] - theta[m-1] wrapped into
// [-pi, pi].  Against the conjugate-product form above this drops the five instructions of the product per output and halves the
// neighbour exchange (one angle instead of a complex number); the arctangent's larger magnitude max(|yr|, |yi|) IS the norm the
// conditioning guard tests, so the guard's three v_max and its v_min3 go too (sdrfm_q.hip: "the conditioning guard").  The price is
// one more rounding of the arctangent in every d (both angles carry the polynomial's 3.9e-7) and the wrap: |d - definition's d| <= 1.1e-6
// worst case where the phase is well conditioned (the guard's business where it is not), measured 7e-7 (tools/q_emulate.py k3="diff").
__device__ __forceinline__ float q_angle(float yr, float yi, float& mx_out) {
  const float ax = __builtin_fabsf(yr), ay = __builtin_fabsf(yi);
  const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 0x1p-120f), mn = __builtin_fminf(ax, ay);   // v_max3_f32, v_min_f32
  mx_out = mx;
  const float v = mn * __builtin_amdgcn_rcpf(mx);
  const float s2 = v * v;
  float q = 0x1.e34882p-8f;
  q = __builtin_fmaf(q, s2, -0x1.22fc74p-5f);
  q = __builtin_fmaf(q, s2, 0x1.509024p-4f);
  q = __builtin_fmaf(q, s2, -0x1.12688cp-3f);
  q = __builtin_fmaf(q, s2, 0x1.96c562p-3f);
  q = __builtin_fmaf(q, s2, -0x1.554086p-2f);
  float a = __builtin_fmaf(v, s2 * q, v);
  if (ay > ax) a = 0x1.921fb6p+0f - a;
  if (yr < 0.0f) a = 0x1.921fb6p+1f - a;
  return __builtin_copysignf(a, yi);                            // (the two reflections as sign transfers — v_bfi around pi / 4 and pi / 2, no compare + select — measured
}                                                               // 0.8 % slower and cost two more roundings: profiles/r06_q_experiments.txt item 4)
// x in (-2 pi, 2 pi) -> x - 2 pi rint(x / 2 pi): three full-rate instructions (the rounding through the 1.5 * 2^23 constant; v_rndne_f32 is a quarter-rate one)
__device__ __forceinline__ float q_wrap(float x) {
  float t = __builtin_fmaf(x, 0x1.45f306p-3f, 0x1.8p+23f);
  asm volatile("" : "+v"(t));                                   // (keeps the two roundings apart)
  const float k = t - 0x1.8p+23f;
  return __builtin_fmaf(k, -0x1.921fb6p+2f, x);
}

// v -> (int16) rint(clamp(v, -32768, 32767)) in both halves of a word, as the host routine has it (csrc/pcm_sink.c): one v_med3 (v is never a NaN for finite
// audio: the routine's two compares give the same value), the rounding to nearest-even through 1.5 * 2^23 (whose sum's low 16 bits are the integer's two's
// complement), one v_perm for the two copies
__device__ __forceinline__ unsigned q_pcm_word(float v) {
  const float c = __builtin_amdgcn_fmed3f(v, -32768.0f, 32767.0f);
  float t = c + 0x1.8p+23f;
  asm volatile("" : "+v"(t));
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, t), __builtin_bit_cast(unsigned, t), 0x05040504u);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 16, "vmcnt immediate");
  __builtin_amdgcn_s_waitcnt(0x0f70 | N);
}

#ifdef SDRFM_Q_PHASES   // development harness: shader cycles per phase of a step, summed per wave (words 8..15 of the wave's 16 debug words)
#define Q_PHASE(i) do { const unsigned long long tn_ = __builtin_readcyclecounter(); t_ph[i] += tn_ - t_last; t_last = tn_; } while (0)
#else
#define Q_PHASE(i) do { } while (0)
#endif

// The body of one workgroup (one wave): `bid` is its index among the launch's design-Q workgroups (the kernel's bid, or — in k_mix
// below — its index among the workgroups that run this body).
// PCM (k_mfir_pcm: the kernel argument is a QPcmArgs): every run also sinks its own audio outputs — de-emphasis, int16 pairs — where they are parked
// (sdrfm_sink_chain.h); 256 bytes of LDS behind the workgroup's q_lds bytes hold the 64 outputs its predecessor's state still reaches.
template <int D, int DA, int NSLOT>
constexpr uint32_t q_lds() { return (uint32_t)(QGeo<D, DA>::PRE + 1024 * NSLOT + 4 * (QGeo<D, DA>::DBW + ABW + FLW + GTW)); }
struct QPcmArgs { SdrfmQParams p; SdrfmSinkChain t; };
static_assert(ABS * 128 <= 64 * (int)SDRFM_CHAIN_CH, "a flush of parked audio is one scan of the wave");
template <int C0, int NSLOT, int D, int DA, bool PCM = false>
__device__ __forceinline__ void mfir_body(const SdrfmQParams& p, const uint32_t bid) {
  using G = QGeo<D, DA>;
  [[maybe_unused]] constexpr int NCH = G::NCH;
  constexpr int QD = D, QDA = DA, NSC = G::NSC, BLKB = G::BLKB, STEPB = G::STEPB, PRE = G::PRE, DBW = G::DBW, RWIN = G::RWIN;
  constexpr bool ALIGNED = G::ALIGNED;
  constexpr int CS = G::CS;
  constexpr int RINGB = NSLOT * 1024;
  static_assert(RINGB % (2 * STEPB) == 0 && (ALIGNED ? NSLOT == 2 * CS : (NSLOT >= 5 && NSLOT - 4 < 16)), "ring: whole pairs of steps");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const db = reinterpret_cast<float*>(smem + PRE + RINGB);
  float* const ab = db + DBW;                                   // parked audio outputs
  unsigned short* const fl16 = reinterpret_cast<unsigned short*>(ab + ABW);   // lanes waiting for the repair path: (step slot + 1) << 6 | lane
  const int lane = (int)threadIdx.x, n = lane & 15, g = lane >> 4;
  const uint32_t si = bid / p.runs, run = bid - si * p.runs;
  const uint32_t stream = p.slist ? p.slist[si] : si;             // (round 5: a launch may serve a list of the handle's streams)
  // Runs are cut in QUADS of four blocks (32 outputs = the warm-up a run needs: QTA - 1 d's and the y before them; 4 BLKB bytes = whole
  // 128-byte lines).  Every wave but the stream's first one (when that takes the carried state) walks one warm-up quad before what it
  // owns; a wave's STEP GRID starts at its warm-up quad (not at a multiple of 16 blocks of the stream), so the warm-up costs a quarter
  // of a step and the Gq = Qt + (warm-up quads) grid quads of a stream are dealt out evenly: the waves' walks differ by one quad at most
  // (configs[2]: 63 or 64 quads = 16 steps for every wave; rounds 3 - 4 cut at whole steps and recomputed a whole step: 16 or 17).
  const uint32_t Bt = (p.M + 7u) >> 3, Qt = (Bt + WQB - 1u) / WQB, vs = p.iq_prev ? 1u : 0u;
  const uint32_t Gq = Qt + p.runs - 1u + vs;
  const uint32_t e0 = (uint32_t)(((uint64_t)run * Gq) / p.runs), e1 = (uint32_t)(((uint64_t)(run + 1) * Gq) / p.runs);
  const int q0 = run == 0 ? 0 : (int)e0 - (int)(run - 1u + vs), q1 = (int)e1 - (int)(run + vs);
  if (q0 >= q1) return;
  if (lane == 0) *reinterpret_cast<uint2*>(fl16 + 128) = make_uint2(0u, 0u);   // repair statistics of this wave (before any LDS-DMA is in flight: no wait)
#ifdef SDRFM_Q_STAMPS   // development harness (tools/qbench): per-wave time stamps, 8 words per wave
  unsigned long long* const tsp = p.dbg ? p.dbg + 16 * (size_t)bid : nullptr;
  unsigned long long t_wait = 0, t_first = 0;
  const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c_entry = __builtin_readcyclecounter();
#endif
#ifdef SDRFM_Q_PHASES
  unsigned long long t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
#endif
  // A run warms up unless it starts the stream's chunk AND takes the carried state; with iq_prev (the previous call's buffer:
  // SDRFM_F_OVERLAP) the stream's first run warms up too, from that buffer's last bytes, and the call depends on nothing the previous
  // call computes.
  const bool from_prev = run == 0 && p.iq_prev != nullptr;
  const bool warm = run > 0 || from_prev;
  const int b0 = WQB * q0;                                      // owned blocks [b0, b1)
  int b1 = WQB * q1;
  if (b1 > (int)Bt) b1 = (int)Bt;
  const bool last_run = b1 == (int)Bt;
  const int bs = warm ? b0 - WQB : b0;                          // the wave's grid starts at this block (-4 for the stream's first run under SDRFM_F_OVERLAP)
  const int nsteps = (b1 - bs + 15) >> 4;
  const int nlast = b1 - bs - 16 * (nsteps - 1);                // blocks of the last step that exist (the lanes beyond hold whatever the ring held)
  const int o0 = 8 * bs;                                        // first output of the grid = output index of d-buffer word DB0 + sigma
  const int jg0 = (o0 + 64 * QDA) / QDA - 64;                   // floor(o0 / QDA): first audio output whose newest d lies in the grid
  const int jlo = warm ? (o0 + 8 * WQB) / QDA : 0;                  // first audio output whose newest d lies in an OWNED block (the ones before it are the previous run's)
  int j1 = (8 * b1) / QDA;
  if (j1 > (int)p.A_out) j1 = (int)p.A_out;
  const int phi = QDA * jg0 + QDA - 1 - o0;                     // newest d of output jg0, relative to the grid's first output (0..QDA-1)
  const int sigma = (phi + 1) & 1;                              // shifts the d buffer so that every lane's window starts on an even word

  // ---- carried state (the stream's first run): the loads are issued here and land in LDS after the ring's prologue has left ------
  // y[m-1] of the step's first output (lane 0).  A warm-up has none: the grid's first d is never read (a run's first audio output reaches
  // QTA - 1 = 31 d's back, the warm-up quad holds 32), and a large value keeps the guard from listing lane 0 for it.
  float cr = 1.0e4f, ci = 0.0f;
  const int HT = (int)p.T - 1;
  unsigned short hb0 = 0, hb1 = 0;
  float hd0 = 0.0f;
  if (!warm) {
    const float2 yp = p.yprev_in[stream];
    cr = yp.x; ci = yp.y;
    const unsigned short* hbp = reinterpret_cast<const unsigned short*>(p.hist_b_in) + (size_t)stream * HT;
    if (lane < HT) hb0 = hbp[lane];                             // the last T-1 <= 89 samples before the call
    if (lane + 64 < HT) hb1 = hbp[lane + 64];
    if (lane < QTA - 1) hd0 = p.hist_d_in[(size_t)stream * (QTA - 1) + lane];
  }

  // ---- the ring --------------------------------------------------------------------------------------------------------------
  const unsigned long long gaddr = (unsigned long long)(p.iq + (size_t)stream * p.iq_stride);
  uint32_t hi = (uint32_t)BLKB * (uint32_t)b1;                  // bytes of the row this run may touch: [.., hi)
  if (hi > 2u * p.N) hi = 2u * p.N;
  const qi4_t rsrc = {(int)(unsigned)gaddr, (int)(unsigned)(gaddr >> 32), (int)hi, 0x00020000};
  // (ALIGNED) the lane's 16 bytes of a chunk, swizzled: the piece that belongs at chunk + 16 lane of the ring is logical piece
  // lane ^ key, key = the block's index mod 8 (a chunk holds 1024 / BLKB blocks; D = 16: two variants, by the parity of the chunk)
  const int lsw0 = ALIGNED ? ((16 * lane) ^ (((((16 * lane) / BLKB)) & 7) << 4)) : 16 * lane;
  const int lsw1 = ALIGNED ? ((16 * lane) ^ (((((1024 + 16 * lane) / BLKB)) & 7) << 4)) : 16 * lane;   // odd chunks (differs for BLKB = 256 only)
  const int gb = BLKB * bs;                                     // row offset of the grid's first byte: a multiple of 4 BLKB (whole 128-byte lines); -4 BLKB for from_prev
  int vpos = gb + (ALIGNED ? 0 : 16 * lane);                    // row offset of the next chunk (ALIGNED: without the lane's part)
  auto slot_ptr = [&](int slot) { return (__attribute__((address_space(3))) void*)(smem + PRE + 1024 * slot); };
  auto lsw = [&](int chunk) { return (BLKB == 256 && (chunk & 1)) ? lsw1 : lsw0; };
  // prologue: the block before the grid (the pre-halo: the first block's window reaches into it; BLKB / 16 lanes), then NSLOT chunks.
  // For the stream's first run under SDRFM_F_OVERLAP the pre-halo and the warm-up quad are the last five blocks of the PREVIOUS
  // call's row: lanes are switched between the two rows by EXEC, not by an out-of-range offset, so that no lane's piece is written
  // twice (a chunk fetched in two parts counts twice in vmcnt: more than the first step's wait assumes, never less).  The first step's
  // chunks (what the first step waits for) go out first, then the requests for the L2-resident tables, then the rest of the ring: the
  // opening burst of all waves' first steps is what every wave's start waits behind.
  constexpr int HL = BLKB / 16, HK = ALIGNED ? 7 : 0;           // lanes of the pre-halo; its swizzle key (block index -1)
  const unsigned long long pa = from_prev ? (unsigned long long)(p.iq_prev + (size_t)stream * p.iq_prev_stride) + 2ull * p.N_prev - (WQB + 1) * BLKB : 0ull;
  const qi4_t rprev = {(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (WQB + 1) * BLKB, 0x00020000};   // the previous row's last WQB + 1 blocks: row offset x (< 0) sits at x + (WQB + 1) BLKB
  if (warm && lane < HL) {
    const int hv = 16 * (lane ^ HK);
    if (from_prev) q_raw_buffer_load_lds(rprev, (__attribute__((address_space(3))) void*)smem, 16, hv, 0, 0, SDRFM_Q_AUX);
    else q_raw_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, gb - BLKB + hv, 0, 0, SDRFM_Q_AUX);
  }
  if constexpr (ALIGNED) {
#pragma unroll
    for (int q = 0; q < CS; ++q) {                              // the first step
      const int off = gb + 1024 * q + lsw(q);
      if (from_prev && off < 0) q_raw_buffer_load_lds(rprev, slot_ptr(q), 16, off + (WQB + 1) * BLKB, 0, 0, SDRFM_Q_AUX);
      else q_raw_buffer_load_lds(rsrc, slot_ptr(q), 16, off, 0, 0, SDRFM_Q_AUX);
    }
  } else {
    // (the instruction's immediate offset moves BOTH the memory address and the LDS address: a group of up to four chunks shares
    // one voffset register and one LDS base)
    auto first_step_chunk = [&](int c, __attribute__((address_space(3))) void* lds, int imm) {
      const int off = vpos + 1024 * c;                          // row offset of the lane's piece of grid chunk c
      if (from_prev && off < 0) q_raw_buffer_load_lds(rprev, lds, 16, off - imm + (WQB + 1) * BLKB, 0, imm, SDRFM_Q_AUX);
      else q_raw_buffer_load_lds(rsrc, lds, 16, off - imm, 0, imm, SDRFM_Q_AUX);
    };
    first_step_chunk(0, slot_ptr(0), 0);
    first_step_chunk(1, slot_ptr(1), 0);
    first_step_chunk(2, slot_ptr(0), 2048);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  // ---- wave constants: tap tables (A operands), audio taps: 9 KiB per wave out of L2 / L1, behind the first step's HBM requests in
  // the CU's texture path (they come back in order, right after those bytes)
  qi4_t At[NSC - C0][SDRFM_Q_DIGITS];
#pragma unroll
  for (int c = C0; c < NSC; ++c)
#pragma unroll
    for (int t = 0; t < SDRFM_Q_DIGITS; ++t)
#if defined(SDRFM_Q_ABLATE) && (SDRFM_Q_ABLATE & 4)
      At[c - C0][t] = qi4_t{lane + c, lane * t, c - t, lane};
#else
      At[c - C0][t] = *reinterpret_cast<const qi4_t*>(p.A + ((size_t)((c * SDRFM_Q_DIGITS + t) * 64 + lane)) * 16);
#endif
  const int sidx = (lane & 1) ? (int)0xDDDDDDDD : (int)0x88888888;   // 2:4 index word: a Q row keeps positions 1, 3 of every four, an I row 0, 2
#if SDRFM_Q_GTAB
  // gt[k] multiplies the k-th oldest d of a window: a table in LDS, read into VGPRs at every audio stage
  float* const gt = reinterpret_cast<float*>(fl16) + FLW + 2;
  static_assert(QTA <= 64 && (FLW + 2) % 4 == 0, "one lane per audio tap; ds_read_b128 of the table");
  const float gt0 = lane < QTA ? p.g[QTA - 1 - lane] : 0.0f;
#else
  float gr[QTA];                                                // gr[k] multiplies the k-th oldest d of a window
#pragma unroll
  for (int k = 0; k < QTA; ++k)                                 // wave-uniform: kept in SGPRs (the tap tables take 60 VGPRs; four waves per SIMD need the rest)
    gr[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.g[QTA - 1 - k])));
#endif

  // the repair path's taps (rare; see repair_flagged) as a table in LDS: hz[u] = h[QTP + 2 QD - 1 - u] for u >= 2 QD, zero below — sample i
  // of a repaired lane's window meets output m0 - 1 + o through hz[i - QD o + 2 QD].  It sits in the d buffer's first words, which nothing
  // reads or writes (the d history of a stage sits in the QTA words right below DB0 + sigma).
  static_assert(QTP + 2 * QD <= DB0 - (int)SDRFM_Q_TA && DB0 == 128, "the tap table of the repair path sits below the d history");
  const float hz0 = lane >= 2 * QD ? p.hpad[QTP + 2 * QD - 1 - lane] : 0.0f;
  const float hz1 = lane < 2 * QD ? p.hpad[2 * QD - 1 - lane] : 0.0f;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  if constexpr (ALIGNED) {                                      // the second step of the ring: exactly CS instructions (what a step's wait leaves in flight)
#pragma unroll
    for (int q = 0; q < CS; ++q) q_raw_buffer_load_lds(rsrc, slot_ptr(CS + q), 16, vpos + STEPB + 1024 * q + lsw(CS + q), 0, 0, SDRFM_Q_AUX);
  } else {
#pragma unroll
    for (int q = 3; q < NSLOT; ++q) q_raw_buffer_load_lds(rsrc, slot_ptr(q & ~3), 16, vpos + 1024 * (q & ~3), 0, 1024 * (q & 3), SDRFM_Q_AUX);
  }
  vpos += 1024 * NSLOT;
  if (!warm) {
    // -> end of the pre-halo (the block before the ring: block index -1, swizzle key 7 where the ring is swizzled)
    constexpr int HSW = ALIGNED ? 0x70 : 0;
#ifdef SDRFM_Q_LDSXOR
    hb0 ^= 0x8080; hb1 ^= 0x8080;                              // (experiment: the ring holds byte - 128 already)
#endif
    if (lane < HT) *reinterpret_cast<unsigned short*>(smem + ((PRE - 2 * HT + 2 * lane) ^ HSW)) = hb0;
    if (lane + 64 < HT) *reinterpret_cast<unsigned short*>(smem + ((PRE - 2 * HT + 2 * (lane + 64)) ^ HSW)) = hb1;
    if (lane < QTA - 1) db[DB0 + sigma - (QTA - 1) + lane] = hd0;
  }
#if SDRFM_Q_K3
  // the angle of y[m-1] of the step's first output and whether its norm is inside the guard's radius (a warm-up: neither matters, see above)
  float cth = 0.0f;
  unsigned long long cfl = 0ull;
  if (!warm) {
    float mxp;
    const float tp = q_angle(cr, ci, mxp);
    cth = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tp)));
    cfl = __builtin_amdgcn_ballot_w64(mxp < p.guard_r) & 1ull;
  }
#endif
  db[lane] = hz0;
  if (lane < 2 * QD) db[64 + lane] = hz1;
#if SDRFM_Q_GTAB
  if (lane < QTA) gt[lane] = gt0;
#endif
  __builtin_amdgcn_wave_barrier();                              // (one wave per workgroup: lanes exchange data through LDS in program order; this only pins that order for the compiler)
  int slot = 0;                                                 // ring slot of the next chunk
  int ringoff = 0;                                              // ring byte offset of the current step
  int osm = 0;                                                  // steps in the d buffer since the last audio stage
  int mbase = o0;                                               // output index of d-buffer word DB0 + sigma
  const int baddr = BLKB * n + 16 * g;                          // window of block n starts at PRE + ringoff - BLKB + BLKB n
  // (ALIGNED) the same pieces through the swizzle: piece c of the window = ring bytes BLKB (n - 1) + 64 c + 16 g, .. + 16 (relative to the
  // step): block n - 1 + (64 c + 16 g) / BLKB, key = that block's index mod 8 (a step is 16 blocks: the key does not depend on the step)
  int rdoff[ALIGNED ? 2 * NSC : 1];
  if constexpr (ALIGNED) {
#pragma unroll
    for (int c = 0; c < 2 * NSC; ++c) {
      const int o = 64 * c + 16 * g, blk = n - 1 + o / BLKB;
      rdoff[c] = BLKB * blk + ((o % BLKB) ^ ((blk & 7) << 4)) + PRE;
    }
  } else {
    rdoff[0] = 0;
  }
  const int srcaddr = 4 * (g > 0 ? lane - 16 : ((lane + 47) & 63));   // lane holding y[m-1] of this lane's first output
  const int dlane = 8 * n + 2 * g;
  const int ylast = (int)p.M - 1 - o0;                          // the call's last output, relative to the grid
  const int ylast_lane = ((ylast & 127) >> 3) + 16 * ((ylast & 7) >> 1);   // (its step is the last run's last one: blocks [bs, Bt) end with the block that holds output M - 1)

  // (A version software-pipelined by one stage — the window of step kk + 1 read, and the ring refilled, while the discriminators of
  // step kk run — measured 33.4 us against 30.5 us for this straight order on the same box: the earlier wait for the next step's bytes
  // costs more than the hidden LDS round trip saves.  Not kept.)
  constexpr int NB = 2 * NSC - 2 * C0;                          // 64-byte pieces a lane reads per step: 2 NSC, the last one (when D / 2 is odd)
  qi4_t B[NB];                                                  // lies beyond the window — the tables hold no tap there, it only completes
                                                                // the last issue's B operand
  auto read_step = [&](qi4_t (&dst)[NB], [[maybe_unused]] bool first) {   // wait for the step at `ringoff`, read its window, park the halo at a wrap
#ifdef SDRFM_Q_STAMPS
    const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
    // this step's bytes have landed: everything but the youngest chunk of the 2.5-chunk steps' ring / but the next step's CS chunks
#ifdef SDRFM_Q_FIRSTWAIT   // experiment (round 5): the first step waits for its own 2.5 chunks only, not for the ring's fourth chunk as well
    if (!ALIGNED && first) wait_vmcnt<(ALIGNED ? CS : NSLOT - 3)>();
    else
#endif
    wait_vmcnt<(ALIGNED ? CS : NSLOT - 4)>();
    asm volatile("" ::: "memory");
#ifdef SDRFM_Q_LDSXOR   // experiment (round 3: slower at the full clock; round 4: re-measured in the sustained regime): byte - 128 once per byte, in the LDS
#pragma unroll
    for (int i = 0; i < STEPB / 512; ++i)
      (void)__hip_atomic_fetch_xor(reinterpret_cast<unsigned long long*>(__builtin_assume_aligned(smem + PRE + ringoff + 8 * (lane + 64 * i), 8)), 0x8080808080808080ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
#ifdef SDRFM_Q_STAMPS
    t_wait += __builtin_readcyclecounter() - tw0;
    if (t_first == 0) t_first = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (ALIGNED) {
      // The block before the step is ALWAYS read from the pre-halo: a step's half of the ring is refilled as soon as its window is in
      // registers, so its last block — the next step's "block before" — is parked there first, every step (a copy of the block as it
      // lies: its index, 15 or 31, and -1 share the swizzle key 7).
      const int ro0 = n == 0 ? 0 : ringoff;
#pragma unroll
      for (int c = 2 * C0; c < 2 * NSC; ++c) dst[c - 2 * C0] = *reinterpret_cast<const qi4_t*>(smem + rdoff[c] + (64 * c < BLKB ? ro0 : ringoff));
      if (lane < BLKB / 16) {
        const qi4_t hcp = *reinterpret_cast<const qi4_t*>(smem + PRE + ringoff + STEPB - BLKB + 16 * lane);
        *reinterpret_cast<qi4_t*>(smem + 16 * lane) = hcp;
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      const unsigned char* wb = smem + baddr + ringoff;
#pragma unroll
      for (int c = 2 * C0; c < 2 * NSC; ++c) dst[c - 2 * C0] = *reinterpret_cast<const qi4_t*>(wb + 64 * c);
      if (ringoff + STEPB == RINGB) {                           // the next step starts the ring over: its pre-halo = the ring's last block
        if (lane < BLKB / 16) {
          const qi4_t hcp = *reinterpret_cast<const qi4_t*>(smem + PRE + RINGB - BLKB + 16 * lane);
          *reinterpret_cast<qi4_t*>(smem + 16 * lane) = hcp;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if constexpr (RINGB == 2 * STEPB) ringoff ^= STEPB;          // (a ring of two steps: one s_xor)
    else ringoff = (ringoff + STEPB == RINGB) ? 0 : ringoff + STEPB;
  };
  auto refill_step = [&](int k) {                               // step k's window is in registers: its slots are free, refill them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (ALIGNED) {                                    // step k + 2 into the half of the ring step k occupied
#pragma unroll
      for (int q = 0; q < CS; ++q) q_raw_buffer_load_lds(rsrc, slot_ptr(slot + q), 16, vpos + 1024 * q + lsw(q), 0, 0, SDRFM_Q_AUX);
      slot = slot ? 0 : CS;
      vpos += STEPB;
      return;
    }
    if constexpr (NSLOT == 5 && SDRFM_Q_PEEL) {
      // The ring of five chunks is two steps of 2.5: the slots a step frees repeat with period two — an odd step refills slots 2, 3, an even one slots 4, 0, 1 (step 0:
      // 0, 1 only, the prologue having filled the ring) — so the slot is a compile-time constant on either side of ONE parity test (round 6: the running slot counter
      // and its wrap-arounds were ten scalar instructions per step).
      if (k & 1) {
        q_raw_buffer_load_lds(rsrc, slot_ptr(2), 16, vpos, 0, 0, SDRFM_Q_AUX);
        q_raw_buffer_load_lds(rsrc, slot_ptr(2), 16, vpos, 0, 1024, SDRFM_Q_AUX);
        vpos += 2048;
      } else {
        if (k > 0) {
          q_raw_buffer_load_lds(rsrc, slot_ptr(4), 16, vpos, 0, 0, SDRFM_Q_AUX);
          vpos += 1024;
        }
        q_raw_buffer_load_lds(rsrc, slot_ptr(0), 16, vpos, 0, 0, SDRFM_Q_AUX);
        q_raw_buffer_load_lds(rsrc, slot_ptr(0), 16, vpos, 0, 1024, SDRFM_Q_AUX);
        vpos += 2048;
      }
      return;
    }
    if (k > 0 && !(k & 1)) {
      q_raw_buffer_load_lds(rsrc, slot_ptr(slot), 16, vpos, 0, 0, SDRFM_Q_AUX);
      vpos += 1024;
      slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
    }
    q_raw_buffer_load_lds(rsrc, slot_ptr(slot), 16, vpos, 0, 0, SDRFM_Q_AUX);      // a pair never straddles the ring's end (pairs start at
    q_raw_buffer_load_lds(rsrc, slot_ptr(slot), 16, vpos, 0, 1024, SDRFM_Q_AUX);   // chunk numbers = 0 or 2 mod 5); offset:1024 moves both addresses
    slot = (slot + 2 >= NSLOT) ? slot + 2 - NSLOT : slot + 2;
    vpos += 2048;
  };
  // Kernel arguments that only rare branches and the epilogue need are read again from the argument segment there (scalar loads through
  // an opaque pointer), not kept in scalar registers across the step loop: those are all taken, and every one more costs a reload per step.
  typedef const __attribute__((address_space(4))) SdrfmQParams* KargPtr;
  auto kargs = [&]() -> KargPtr {
    KargPtr pp = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pp));
    return pp;
  };
  // The audio is not stored stage by stage: a store counts in vmcnt like the ring's fetches, and the next step's wait for "all but the
  // youngest fetches" then also waits for the stores' acknowledgement AND for the fetches issued before them — a full round trip through
  // the memory system every five steps (measured: 4.5 of 27.5 us per call on configs[2]).  The outputs are parked in LDS and stored
  // after the run's last step (every ABS stages in a long run), whole 8-byte pairs when the row allows.
  int npend = 0, jfl = jg0;                                     // parked stages; audio output index of the first parked word
  // (PCM) the run's de-emphasis state so far, whether nothing has been flushed yet, the first outputs' values; the sink's parameters from the argument segment
  [[maybe_unused]] float yrun = 0.0f;
  [[maybe_unused]] bool pfirst = true;
  [[maybe_unused]] float* const pstash = reinterpret_cast<float*>(smem + q_lds<D, DA, NSLOT>());
  typedef const __attribute__((address_space(4))) SdrfmSinkChain* KChainPtr;
  [[maybe_unused]] auto kchain = [&]() -> KChainPtr {
    const __attribute__((address_space(4))) unsigned char* pp = (const __attribute__((address_space(4))) unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pp));
    return (KChainPtr)(pp + offsetof(QPcmArgs, t));
  };
  // (PCM) the first poll of the predecessor's word, issued inside the run's last flush — as soon as the scan has the run's end state — and consumed behind it
  [[maybe_unused]] unsigned long long pv_early = 0ull;
  [[maybe_unused]] bool published = false;
  [[maybe_unused]] float dp_early = 0.0f;                        // (1 - alpha)^(lane + 1), fetched with them
  auto flush_audio = [&](const bool fin = false) {
    __builtin_amdgcn_wave_barrier();
    // Row elements [max(jfl, jlo), min(jfl + 128 npend, j1)) come from ab[e - jfl] (the outputs before jlo — a warm-up's — are the previous
    // run's; those from j1 on the next one's), stored as 8-byte pairs aligned in MEMORY whatever the parity of jfl (4-byte stores cost
    // read-modify-writes: round 3), single words at the two ends only.
#if defined(SDRFM_Q_ABLATE) && (SDRFM_Q_ABLATE & 256)   // every wave stores to the same 2 KiB (same instructions, no write traffic to speak of)
    float* row = p.audio - jfl;
    const int ln_ = lane;
#else
    uint32_t st_ = stream;                                      // (derived here, from opaque copies: a pointer kept across the step loop costs registers there)
    int ln_ = lane;
    asm volatile("" : "+s"(st_), "+v"(ln_));
    float* row = kargs()->audio + (size_t)st_ * kargs()->audio_stride;
#endif
    const int lo = jfl > jlo ? jfl : jlo;
#if defined(SDRFM_Q_ABLATE) && (SDRFM_Q_ABLATE & 128)
    const int hi_ = (ab[lane] == 1234.5f) ? lo + 2 : lo;
#else
    const int hi_ = (j1 < jfl + 128 * npend) ? j1 : jfl + 128 * npend;
#endif
    const int par = (int)((reinterpret_cast<uintptr_t>(row) >> 2) & 1u);   // row + e is 8-byte aligned where e + par is even
    bool want_audio = true;
    if constexpr (PCM) want_audio = kargs()->audio != nullptr;   // (sdrfm_process_batch_pcm without an audio buffer: the PCM is all the caller takes)
    if (want_audio) for (int e = lo - ((lo + par) & 1) + 2 * ln_; e < hi_; e += 128) {
      const bool v0 = e >= lo, v1 = e + 1 < hi_;
      const float a0 = v0 ? ab[e - jfl] : 0.0f, a1 = v1 ? ab[e + 1 - jfl] : 0.0f;
      // (PCM: the audio beside the PCM words is stored with the non-temporal hint — with two sets of stores at the end of a wave, 25.1 -> 24.9 us per call; for the
      // one set of the kernel without the chain the hint costs 0.35 us, and on the PCM words 0.25: profiles/r06_sink.txt)
      if constexpr (PCM) {
        if (v0 && v1) __builtin_nontemporal_store(qf2_t{a0, a1}, reinterpret_cast<qf2_t*>(row + e));
        else if (v0) __builtin_nontemporal_store(a0, row + e);
        else if (v1) __builtin_nontemporal_store(a1, row + e + 1);
      } else {
        if (v0 && v1) *reinterpret_cast<qf2_t*>(row + e) = qf2_t{a0, a1};
        else if (v0) row[e] = a0;
        else if (v1) row[e + 1] = a1;
      }
    }
    if constexpr (PCM) {
      // ---- the sink's chain over the same outputs, where they lie (sdrfm_sink_chain.h): a blocked scan of the wave from the run's state so far (0 at its
      // first flush); the packed words take the outputs' place in LDS and are stored as they lie.  The first SDRFM_CHAIN_FIX outputs of the run are kept as
      // values: the predecessor's state still reaches them (finished after the run's last step).
      const int cntf = hi_ - lo;
      if (cntf > 0) {
        constexpr int PCH = (int)SDRFM_CHAIN_CH, FIX = (int)SDRFM_CHAIN_FIX;
        const KChainPtr tp = kchain();
        const float alpha = tp->alpha, gain = tp->gain;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the stores' operands have left the LDS)
        __builtin_amdgcn_wave_barrier();
        float* const xa = ab + (lo - jfl);
        unsigned* const xw = reinterpret_cast<unsigned*>(xa);
        const int i0 = PCH * ln_;
        float xr[PCH];
#pragma unroll
        for (int q = 0; q < PCH; ++q) xr[q] = (i0 + q < cntf) ? xa[i0 + q] : 0.0f;
        // 1. the chunk's own contribution to its last sample: sum of alpha d^(7-q) x[q] (the weights from the host, in scalar registers; lane 0 adds what is left
        // of the run's state so far).  Samples past the flush's end are zeros: they add nothing.
        float sc = 0.0f;
#pragma unroll
        for (int q = 0; q < PCH; ++q) sc = __builtin_fmaf(tp->w[q], xr[q], sc);
        float pw = tp->pc;
        if (ln_ == 0) sc = __builtin_fmaf(pw, yrun, sc);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {                      // 2. the carries between chunks
          const float o = __shfl_up(sc, (unsigned)d, 64);
          const float sn = __builtin_fmaf(pw, o, sc);
          sc = ln_ >= d ? sn : sc;
          pw *= pw;
        }
        if (fin) {
          // The run's end state is known HERE, before the second walk: the scan's value of the lane that holds the last output is that state decayed over the
          // zeros behind it in the lane's chunk — undone by the host's (1 - alpha)^-k.  Published at once, and the predecessor's word asked for at once: both
          // round trips run under the second walk and the stores instead of behind them.  (Re-associated like every carry: 1e-7 relative.)
          const int Ll = (cntf - 1) / PCH, rl = (cntf - 1) % PCH;
          const float ypub = __shfl(sc, Ll, 64) * tp->dinv[PCH - 1 - rl];
          uint32_t bid2 = bid;
          asm volatile("" : "+s"(bid2));
          if (ln_ == 0) {
            const uint32_t call = tp->call, nst = tp->n_streams;
            const unsigned long long w = ((unsigned long long)(call + 1u) << 32) | __builtin_bit_cast(unsigned, ypub);
            __hip_atomic_store(last_run ? tp->sg + (size_t)((call + 1u) % SDRFM_CHAIN_SG_SLOTS) * nst + st_ : tp->runstate + bid2, w, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long zero = 0ull;
            asm volatile("" : "+v"(zero));
            pv_early = __hip_atomic_fetch_add(run == 0 ? tp->sg + (size_t)(call % SDRFM_CHAIN_SG_SLOTS) * nst + st_ : tp->runstate + (bid2 - 1u), zero,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          dp_early = ln_ < (int)SDRFM_CHAIN_FIX ? tp->dpow[ln_] : 0.0f;
          published = true;
        }
        float y = __shfl_up(sc, 1u, 64);                        // 3. the exact form's chain from the true carry-in
        if (ln_ == 0) y = yrun;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < PCH; ++q) {
          const float yn = __builtin_fmaf(alpha, xr[q] - y, y);
          y = (i0 + q < cntf) ? yn : y;
          if (i0 + q < cntf) {
            xw[i0 + q] = q_pcm_word(yn * gain);
            if (pfirst && i0 + q < FIX) pstash[i0 + q] = yn;
          }
        }
        yrun = __shfl(y, (cntf - 1) / PCH, 64);
        __builtin_amdgcn_wave_barrier();
        // the words as they lie, 8-byte pairs aligned in memory (as the audio above)
        unsigned* const out = reinterpret_cast<unsigned*>(tp->pcm + (size_t)st_ * tp->pcm_stride);
        const int parp = (int)((reinterpret_cast<uintptr_t>(out) >> 2) & 1u), ps = lo + (pfirst ? FIX : 0);
        for (int e = ps - ((ps + parp) & 1) + 2 * ln_; e < hi_; e += 128) {
          const bool v0 = e >= ps, v1 = e + 1 < hi_;
          const unsigned w0 = v0 ? xw[e - lo] : 0u, w1 = v1 ? xw[e + 1 - lo] : 0u;
          if (v0 && v1) *reinterpret_cast<uint2*>(out + e) = make_uint2(w0, w1);
          else if (v0) out[e] = w0;
          else if (v1) out[e + 1] = w1;
        }
        pfirst = false;
      }
    }
    jfl += 128 * npend;
    npend = 0;
  };
  // ---- the repair path (rare: see the guard in the loop).  Every listed lane (one per lane of the wave, 64 at a time) recomputes the three
  // outputs under its two d's — y[m0 - 1], y[m0], y[m0 + 1], m0 = the lane's first output of that step — with the definition's chain:
  // acc = fmaf(h[k], x, acc), oldest sample first, x = byte - 127.5, from the raw bytes in global memory (the ring's slots have been
  // refilled by now): RWIN = 84 consecutive samples, 21 aligned 8-byte loads.  Samples before the call come from the previous call's
  // buffer (SDRFM_F_OVERLAP) or from the 64 raw samples the previous design-Q call left in hist_q.  The taps are wave-uniform LDS reads
  // (h[k] = 0 for k >= T: fmaf(0, x, acc) = acc bit for bit, acc is never -0).  Then the definition's own
  // discriminator (sdrfm_math.h) — the d's are the bit-exact kernels' d's, a fixed function of the bytes like everything else here.
  int nflag = 0;
  auto repair_flagged = [&]() {
    // Everything this path needs is derived here, from opaque copies of the lane and stream numbers, so that none of it is hoisted into
    // the step loop (whose scalar registers are all taken); and the wave constants are loaded again at the end (L2) instead of being kept
    // across it: their 36 VGPRs and 32 SGPRs are what the chains run in.
    int ln = lane;
    uint32_t st = stream;
    asm volatile("" : "+v"(ln), "+s"(st));
    const KargPtr pp = kargs();
#ifdef SDRFM_Q_PHASES   // development harness: cycles of a repair call by part, two 32-bit sums per word (t_ph[0]: set-up | data wait, t_ph[7]: chains | reload)
    unsigned long long r_t0 = __builtin_readcyclecounter(), r_t1;
#define R_PHASE(word, shift, waitfirst) do { if (waitfirst) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); r_t1 = __builtin_readcyclecounter(); \
                                             t_ph[word] += ((r_t1 - r_t0) & 0xffffffffull) << shift; r_t0 = r_t1; } while (0)
#else
#define R_PHASE(word, shift, waitfirst) do { } while (0)
#endif
    const unsigned char* const row = pp->iq + (size_t)st * pp->iq_stride;
    const unsigned char* const pre = from_prev ? pp->iq_prev + (size_t)st * pp->iq_prev_stride + 2 * (size_t)pp->N_prev
                                               : pp->hist_q_in + (size_t)(2 * QTP) * ((size_t)st + 1);
    unsigned* const cnt = reinterpret_cast<unsigned*>(fl16 + 128);
    __builtin_amdgcn_wave_barrier();                            // (the list was written by other lanes)
    unsigned nrep = 0;
    for (int e0 = 0; e0 < nflag; e0 += 64) {
      const int e = e0 + ln;
      int l = 0, slot = 0, m0 = (int)pp->M;
      if (e < nflag) {
        const unsigned ent = fl16[e];
        l = (int)(ent & 63u);
        slot = (int)(ent >> 6) - 1;
        m0 = mbase + 128 * slot + 8 * (l & 15) + 2 * (l >> 4);
      }
      const bool act = m0 < (int)pp->M;                           // (not the lanes beyond the call's last output in a partly filled last step)
      nrep += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
      if (act) {
        const int n0 = QD * m0 - QTP;                           // first sample under y[m0 - 1]; a multiple of 4
        uint2 w[RWIN / 4];
#pragma unroll
        for (int c = 0; c < RWIN / 4; ++c) {
          const int n = n0 + 4 * c;
          w[c] = *reinterpret_cast<const uint2*>((n >= 0 ? row : pre) + 2 * (ptrdiff_t)n);
        }
        R_PHASE(0, 0, 0);
        R_PHASE(0, 32, 1);
        // (I, Q) pairs: v_pk_fma_f32 / v_pk_add_f32 — per lane the same IEEE operations as the scalar chain, half the instructions.  Sample i
        // meets output m0 - 1 + o through tap hz[i - QD o + 2 QD] (wave-uniform LDS reads: broadcasts), where that index is >= 2 QD and below
        // QTP + 2 QD; indices below 2 QD read zeros, which leave the accumulator as it is (fmaf(0, x, acc) = acc bit for bit; acc is never -0).
        qf2_t a[3] = {qf2_t{0.0f, 0.0f}, qf2_t{0.0f, 0.0f}, qf2_t{0.0f, 0.0f}};
        const float* const hz = db;
#pragma unroll
        for (int i = 0; i < RWIN; ++i) {
          const unsigned wd = (i & 2) ? w[i >> 2].y : w[i >> 2].x;
          const qf2_t b = {(float)((wd >> ((i & 1) ? 16 : 0)) & 0xffu), (float)((wd >> ((i & 1) ? 24 : 8)) & 0xffu)};   // v_cvt_f32_ubyteN
          const qf2_t x = b - qf2_t{127.5f, 127.5f};
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            const int u = i - QD * o + 2 * QD;
            if (u < 2 * QD || u >= QTP + 2 * QD) continue;      // (no tap of this output meets the sample: nothing issued)
            const float t = hz[u];
            a[o] = __builtin_elementwise_fma(qf2_t{t, t}, x, a[o]);
          }
          if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from converting the whole window up front: registers)
        }
        if (m0 == 0 && !warm && pp->yprev_exact) {                // the call's first output after a reset or a bit-exact kernel: y[-1] is the carried one
          const float2 yp = pp->yprev_in[st];
          a[0] = qf2_t{yp.x, yp.y};
        }
        float* dst = db + DB0 + sigma + 128 * slot + 8 * (l & 15) + 2 * (l >> 4);
        dst[0] = sdrfm_discriminate(a[1].x, a[1].y, a[0].x, a[0].y);
        dst[1] = sdrfm_discriminate(a[2].x, a[2].y, a[1].x, a[1].y);
      }
    }
    R_PHASE(7, 0, 1);
    if (ln == 0) {                                              // statistics: per wave in LDS (an atomic per pass from every wave queues up at one L2 address)
      cnt[0] += nrep;
      cnt[1] += 1u;
    }
    nflag = 0;
    R_PHASE(7, 32, 1);
  };
  // Issue priority by age, from the middle of the run on.  The arbiter serves the oldest wave of a SIMD first and the memory pipeline's
  // queues do the same, so the first / second / third wave of a SIMD end 20.0 / 21.8 / 23.6 us after the launch (configs[2], §4.Q of
  // DESIGN.md) and the kernel waits for the last one.  From its middle step on a wave runs at priority = its age rank (HW_ID.wave_id:
  // 0 = oldest), which closes the gap: 26.5 -> 25.7 us per call.  (For the whole run it over-corrects: the oldest waves end last.)
  // Not for overlapped calls — two kernels share the SIMDs there and the ranks mean something else: measured +0.3 us.  Priorities by
  // remaining steps (a wave that is behind outranks one that is ahead), alone or on top of the rank: no better (25.9 - 26.4 us).
#ifdef SDRFM_Q_GUARD_L1
  float guard_r = 2.0f * p.guard_r, guard_a = p.guard_a;
#else
  float guard_r = p.guard_r, guard_a = p.guard_a;             // (in VGPRs: the loop's scalar registers are all taken)
#endif
  asm volatile("" : "+v"(guard_r), "+v"(guard_a));
#ifdef SDRFM_Q_SCALE_VGPR   // experiment (round 5): the recombination's factors in VGPRs (an FMA with an SGPR operand issues at half rate) — measured 0.2 us
  float q0v = p.q0, q2v = p.q2, cstv = p.cst;                  // per call SLOWER than leaving them in SGPRs (the kernel sits at its 128 VGPRs): profiles/r05_q_experiments.txt
  asm volatile("" : "+v"(q0v), "+v"(q2v), "+v"(cstv));
#else
  const float q0v = p.q0, q2v = p.q2, cstv = p.cst;
#endif
#if SDRFM_Q_MICRO && SDRFM_Q_K3
  int doff = (PRE + RINGB) + 4 * (DB0 + sigma + dlane);          // LDS byte offset of the lane's two d's of the stage's first step (opaque: an offset, not a
  asm volatile("" : "+v"(doff));                                // pointer — an opaque POINTER loses its address space and the write becomes a flat store)
#endif
  const int wrank = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 4) & 15u);
  const int prio_at = p.prio_by_age ? nsteps / 2 : -1;
  // One step.  LAST (a compile-time flag): the run's last step, peeled out of the loop (round 6) — everything that only the last step does (the call's y[-1] and d
  // history handed over, the partly filled step's lanes kept off the guard's list, the repair pass and the audio stage that must not wait for a full stage) then costs
  // the other steps nothing: ~20 scalar instructions and four exec-mask round trips per step, a tenth of a wave's own instruction stream (profiles/r06_q_experiments.txt
  // item 6).  The last step also refills nothing: no step is left to read the ring.
#if SDRFM_Q_PEEL
  auto step = [&](auto LAST_, const int kk) {
    constexpr bool LAST = decltype(LAST_)::value;
#else
  for (int kk = 0; kk < nsteps; ++kk) {
    const bool LAST = kk == nsteps - 1;
#endif
    if (kk == prio_at) {
      if (wrank == 0) __builtin_amdgcn_s_setprio(0);
      else if (wrank == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(2);
    }
#ifdef SDRFM_Q_PHASES
    t_last = __builtin_readcyclecounter();
#endif
    read_step(B, kk == 0);
#if SDRFM_Q_HO_RING
    // (round 6) The part of the state hand-over that is a copy of INPUT bytes — the call's last T - 1 samples in two formats, the last 64 raw ones for the next
    // call's repair path — is taken from the RING when the run's last step begins (the row's last bytes lie in that step's window, which has just landed; no
    // fetch is issued after this point) and stored at once: behind the last step it was a fetch from memory and three stores at the very end of the stream's
    // last wave, and a wave cannot end before its stores are acknowledged (profiles/r06_q_experiments.txt item 8).  Linear rings only (D = 10): ring byte of grid
    // byte G is PRE + G mod RINGB; the swizzled rings keep the copy behind the loop.
    if constexpr (!ALIGNED) {
      if (LAST && last_run) {
        const KargPtr pp = kargs();
        const int HTe = (int)pp->T - 1, Ge = 2 * (int)pp->N - gb;     // grid byte offset of the row's end
        for (int k = lane; k < HTe; k += 64) {
          const unsigned raw = *reinterpret_cast<const unsigned short*>(smem + PRE + ((Ge - 2 * HTe + 2 * k) % RINGB));
          reinterpret_cast<unsigned short*>(pp->hist_b_out)[(size_t)stream * HTe + k] = (unsigned short)raw;
          pp->hist_x_out[(size_t)stream * HTe + k] = make_float2((float)(raw & 0xffu) - 127.5f, (float)(raw >> 8) - 127.5f);
        }
        if (lane < 2 * QTP / 16)
          *reinterpret_cast<qi4_t*>(pp->hist_q_out + (size_t)(2 * QTP) * stream + 16 * lane) =
              *reinterpret_cast<const qi4_t*>(smem + PRE + ((Ge - 2 * QTP + 16 * lane) % RINGB));
      }
    }
#endif
    Q_PHASE(1);                                                 // wait for the step's bytes, window reads issued
    if (!(SDRFM_Q_PEEL && LAST)) refill_step(kk);
#ifdef SDRFM_Q_EARLYSTORE   // experiment (round 5): the stages parked so far are stored BEFORE the run's last step computes (their acknowledgements then come
    if (LAST && npend > 0) flush_audio();   // back during it: a wave ends only when its stores are acknowledged), the last stage alone after it
#endif
    Q_PHASE(2);                                                 // LDS round trip of the window, refill issue
    // ---- K2 on the matrix pipe ------------------------------------------------------------------------------------------------
    qi4_t acc[SDRFM_Q_DIGITS];
#pragma unroll
    for (int t = 0; t < SDRFM_Q_DIGITS; ++t) acc[t] = qi4_t{0, 0, 0, 0};   // (zeroed by the matrix pipe itself — a dense issue with a zero A operand — measured slower: r06_q_experiments.txt item 3)
    [[maybe_unused]] constexpr int XM = (int)0x80808080;
#ifndef SDRFM_Q_LDSXOR
#pragma unroll
    for (int c = 0; c < NB - 1 + (NCH & 1 ? 0 : 1); ++c)                                      // byte - 128 as i8 (not the piece beyond the window)
      B[c] = B[c] ^ ((SDRFM_Q_XORSKIP && QD == 10 && C0 == 0 && c == 0) ? qi4_t{0, XM, XM, XM} : qi4_t{XM, XM, XM, XM});
#endif
#pragma unroll
    for (int c = C0; c < NSC; ++c) {
      // 128 window bytes per issue: the lane's two 16-byte pieces 64 bytes apart
      const qi4_t lo = B[2 * c - 2 * C0], hi = B[2 * c + 1 - 2 * C0];
      const qi8_t b = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int t = 0; t < SDRFM_Q_DIGITS; ++t) {
#if defined(SDRFM_Q_ABLATE) && (SDRFM_Q_ABLATE & 2)
        acc[t] += At[c - C0][t] & lo;
#else
        acc[t] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(At[c - C0][t], b, acc[t], sidx, 0, 0);
#endif
      }
    }
    // ---- digits -> y: rows 4 g + {0, 1, 2, 3} = (I, Q) of output 2 g, (I, Q) of output 2 g + 1 of block n -------------------------
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (The digit sums as exact floats — accumulators started at the bit pattern of 1.5 * 2^23, a full-rate subtraction instead of the conversions and the
      // shift-add, three fused multiply-adds — measured no faster with the factors in SGPRs and slower with them in VGPRs: r06_q_experiments.txt item 2.)
      const int s01 = acc[0][r] + acc[1][r] * 256;              // exact: |S0| <= 2^20, |S1 << 8| <= 2^28
      y[r] = __builtin_fmaf((float)s01, q0v, __builtin_fmaf((float)acc[2][r], q2v, cstv));
    }
#ifdef SDRFM_Q_PHASES
    asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
#endif
    Q_PHASE(3);                                                 // xor, MFMAs, recombination
    // ---- K3: y[m-1] of the lane's first output sits in lane - 16 (or is the previous step's last output) -------------------------
#if SDRFM_Q_K3
    // (round 6) angles, not products: the lane's second angle first — it is what the neighbour needs —, the exchange in flight under the first one's chain
    float mx0, mx1;
    const float th1 = q_angle(y[2], y[3], mx1);
    float thp = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(srcaddr, __builtin_bit_cast(int, th1)));
    const float th0 = q_angle(y[0], y[1], mx0);
#if SDRFM_Q_MICRO
    asm("v_writelane_b32 %0, %1, 0" : "+v"(thp) : "s"(cth));    // lane 0: the carried angle
#else
    if (lane == 0) thp = cth;
#endif
#ifdef SDRFM_Q_PHASES
    asm volatile("" : "+v"(thp));
#endif
    Q_PHASE(4);                                                 // neighbour exchange
    cth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, th1), 63));
    const float d1 = q_wrap(th1 - th0);
    const float d0 = q_wrap(th0 - thp);
#else
    float pr = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(srcaddr, __builtin_bit_cast(int, y[2])));
    float pi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(srcaddr, __builtin_bit_cast(int, y[3])));
    if (lane == 0) { pr = cr; pi = ci; }
#ifdef SDRFM_Q_PHASES
    asm volatile("" : "+v"(pr), "+v"(pi));
#endif
    Q_PHASE(4);                                                 // neighbour exchange
    cr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y[2]), 63));
    ci = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y[3]), 63));
#if defined(SDRFM_Q_ABLATE) && (SDRFM_Q_ABLATE & 1)   // timing experiments of the development harness only (wrong results)
    const float d0 = y[0] + pr + y[1] * pi, d1 = y[2] + y[3] * y[1] + y[0];
#else
    const float d1 = q_discriminate(y[2], y[3], y[0], y[1]);   // (the pair that needs no neighbour first: its chain runs while the exchange is in flight)
    const float d0 = q_discriminate(y[0], y[1], pr, pi);
#endif
#endif
    {
#if SDRFM_Q_MICRO && SDRFM_Q_K3
      float* dst = reinterpret_cast<float*>(smem + (doff + 512 * osm));
#else
      float* dst = db + DB0 + sigma + 128 * osm + dlane;
#endif
      dst[0] = d0;
      dst[1] = d1;
    }
    if (LAST && last_run && lane == ylast_lane) kargs()->yprev_out[stream] = make_float2(y[2], y[3]);   // (the call's last output lies in the last run's last step)
#if !(defined(SDRFM_Q_ABLATE) && (SDRFM_Q_ABLATE & 512))
    // ---- the conditioning guard: design Q's y is within E ~ 1e-4 (absolute) of the definition's fmaf chain, so its d is within
    // E / |y| + E / |p| of the definition's — fine while both magnitudes are large, not at a deep fade, and near d = +-pi the two may land
    // on different sides of the branch cut.  A lane one of whose three y's is small (max norm below guard_r) or one of whose d's is
    // within reach of the cut (above guard_a) is put on a list; its two d's are recomputed by the definition's own chain before anything
    // reads them (repair_flagged).  An FM carrier never gets here; noise-only input does, a few lanes per step.
    {
#if SDRFM_Q_K3
      // (round 6) the norms are the arctangents' own larger magnitudes: two compares; the neighbour's y[m-1] is small exactly when the lane it came
      // from found its second norm small — a shift of that lane mask in scalar registers (lane l > 15 took it from lane l - 16, lane n in 1 .. 15 from
      // lane 47 + n, lane 0 from the previous step's lane 63) instead of a third norm and a v_min3
      float t1;
      unsigned long long fm, m0, m1;
      asm("v_cmp_lt_f32_e64 %[m0], %[x0], %[gr]\n\t"
          "v_cmp_lt_f32_e64 %[m1], %[x1], %[gr]\n\t"
          "v_max_f32_e64 %[t1], |%[d0]|, |%[d1]|\n\t"
          "v_cmp_gt_f32_e64 %[fm], %[t1], %[ga]"
          : [t1] "=&v"(t1), [fm] "=&s"(fm), [m0] "=&s"(m0), [m1] "=&s"(m1)
          : [x0] "v"(mx0), [x1] "v"(mx1), [d0] "v"(d0), [d1] "v"(d1), [gr] "v"(guard_r), [ga] "v"(guard_a));
      fm |= m0 | m1 | cfl;                                        // (a carrier: all four masks are empty — the shifted copies of m1 are formed only when something is set)
      if (fm) fm |= (m1 << 16) | ((m1 >> 47) & 0xFFFEull);
      cfl = m1 >> 63;
#else
      float t0, t1, t2;
      unsigned long long fm, fm2;
#ifdef SDRFM_Q_GUARD_L1   // experiment (round 5): |re| + |im| (a full-rate add) instead of max(|re|, |im|) against twice the radius: flags a superset
      asm("v_add_f32_e64 %[t0], |%[y0]|, |%[y1]|\n\t"
          "v_add_f32_e64 %[t1], |%[y2]|, |%[y3]|\n\t"
          "v_add_f32_e64 %[t2], |%[pr]|, |%[pi]|\n\t"
          "v_min3_f32 %[t0], %[t0], %[t1], %[t2]\n\t"
#else
      asm("v_max_f32_e64 %[t0], |%[y0]|, |%[y1]|\n\t"
          "v_max_f32_e64 %[t1], |%[y2]|, |%[y3]|\n\t"
          "v_max_f32_e64 %[t2], |%[pr]|, |%[pi]|\n\t"
          "v_min3_f32 %[t0], %[t0], %[t1], %[t2]\n\t"
#endif
          "v_max_f32_e64 %[t1], |%[d0]|, |%[d1]|\n\t"
          "v_cmp_lt_f32_e64 %[fm], %[t0], %[gr]\n\t"
          "v_cmp_gt_f32_e64 %[fm2], %[t1], %[ga]\n\t"
          "s_or_b64 %[fm], %[fm], %[fm2]"
          : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [fm] "=&s"(fm), [fm2] "=&s"(fm2)
          : [y0] "v"(y[0]), [y1] "v"(y[1]), [y2] "v"(y[2]), [y3] "v"(y[3]), [pr] "v"(pr), [pi] "v"(pi), [d0] "v"(d0), [d1] "v"(d1),
            [gr] "v"(guard_r), [ga] "v"(guard_a)
          : "scc");
#endif
      if (fm) {                                                                                // wave-uniform, rare
        if (LAST && nlast < 16) fm &= 0x0001000100010001ull * ((1ull << nlast) - 1ull);   // the run's last step: only its first nlast blocks exist
        if (fm) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
          if ((fm >> lane) & 1ull) fl16[nflag + rank] = (unsigned short)(((osm + 1) << 6) | lane);
          nflag += __builtin_popcountll(fm);
        }
      }
    }
#endif
    ++osm;
    // everything the list holds is repaired before the d's are read: at every audio stage, at the run's last step (the d history below), and
    // whenever another step's worth of lanes might not fit
    if (nflag > 0 && (osm == QDA || LAST || nflag > 64)) {
      __builtin_amdgcn_s_setprio(3);                            // a wave in the repair path is behind its SIMD's others: first in line until it is through
      repair_flagged();
      if (prio_at >= 0 && kk >= prio_at) {
        if (wrank == 0) __builtin_amdgcn_s_setprio(0);
        else if (wrank == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(2);
      } else {
        __builtin_amdgcn_s_setprio(0);
      }
    }
    __builtin_amdgcn_wave_barrier();                            // (the d's are read by other lanes from here on)
    // the stream's last 31 d's, taken before the audio stage below may move the buffer on (the call's last step need not be full)
    if (LAST && last_run && lane < QTA - 1) {
      const KargPtr pp = kargs();
      pp->hist_d_out[(size_t)stream * (QTA - 1) + lane] = db[DB0 + sigma + ((int)pp->M - (QTA - 1) + lane - mbase)];
    }
#ifdef SDRFM_Q_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    Q_PHASE(5);                                                 // discriminators, d write

    // ---- K4: 128 audio outputs per five owned steps, two consecutive outputs per lane -------------------------------------------
    if (osm == QDA || (LAST && osm > 0)) {
      const float* w = db + DB0 + sigma + phi + 2 * QDA * lane - (QTA - 1);   // oldest d of the stage's output 2 lane: an even word
      // The whole 40-word window is read up front (one LDS round trip), then four independent chains: each output's 32 taps as two
      // halves of 16, oldest d first within a half, summed at the end.  (Reading the window eight words at a time in two chains took 340
      // - 470 cycles per step of the wave's time against 280 for this; the kernel's time did not move: the steps wait for their bytes.)
      float dw[QTA + QDA + 3];
#pragma unroll
      for (int i = 0; i < (QTA + QDA + 3) / 2; ++i) {
        const qf2_t v = *reinterpret_cast<const qf2_t*>(w + 2 * i);
        dw[2 * i] = v.x;
        dw[2 * i + 1] = v.y;
      }
      float a0 = 0.0f, a0b = 0.0f, a1 = 0.0f, a1b = 0.0f;
#if SDRFM_Q_GTAB
#pragma unroll
      for (int q = 0; q < QTA / 8; ++q) {                       // (the same four chains in the same order: bit-identical to the taps-in-SGPRs form)
        const qf4_t tl = *reinterpret_cast<const qf4_t*>(gt + 4 * q), th = *reinterpret_cast<const qf4_t*>(gt + QTA / 2 + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = 4 * q + i;
          a0 = __builtin_fmaf(tl[i], dw[k], a0);
          a0b = __builtin_fmaf(th[i], dw[QTA / 2 + k], a0b);
          a1 = __builtin_fmaf(tl[i], dw[QDA + k], a1);
          a1b = __builtin_fmaf(th[i], dw[QDA + QTA / 2 + k], a1b);
        }
      }
#else
#pragma unroll
      for (int k = 0; k < QTA / 2; ++k) {
        a0 = __builtin_fmaf(gr[k], dw[k], a0);
        a0b = __builtin_fmaf(gr[QTA / 2 + k], dw[QTA / 2 + k], a0b);
        a1 = __builtin_fmaf(gr[k], dw[QDA + k], a1);
        a1b = __builtin_fmaf(gr[QTA / 2 + k], dw[QDA + QTA / 2 + k], a1b);
      }
#endif
      a0 += a0b;
      a1 += a1b;
      *reinterpret_cast<qf2_t*>(ab + 128 * npend + 2 * lane) = qf2_t{a0, a1};
      ++npend;
      if (osm == QDA) {                                         // the stage's last 32 d's become the next stage's history
        if (lane < QTA) {
          const float hv = db[DB0 + sigma + 128 * QDA - QTA + lane];
          db[DB0 + sigma - QTA + lane] = hv;
        }
        __builtin_amdgcn_wave_barrier();
        osm = 0;
        mbase += 128 * QDA;
      }
      if (npend == ABS) flush_audio();                          // (long runs only: configs[2]'s runs hold 3.4 stages)
      Q_PHASE(6);                                               // audio stage
    }
#if SDRFM_Q_PEEL
  };
  for (int kk = 0; kk < nsteps - 1; ++kk) step(std::false_type{}, kk);
  step(std::true_type{}, nsteps - 1);
#else
  }
#endif
#ifdef SDRFM_Q_STAMPS
  const unsigned long long t_loop = __builtin_amdgcn_s_memrealtime();
#endif
  wait_vmcnt<0>();                                              // nothing may still be in flight towards this wave's LDS when it ends
  flush_audio(true);
  if constexpr (PCM) {
    // ---- the run's end state published, its predecessor's taken, the run's first outputs finished (sdrfm_sink_chain.h) ---------------------------------
    const KChainPtr tp = kchain();
    uint32_t st_ = stream, bid_ = bid;
    int ln_ = lane;
    asm volatile("" : "+s"(st_), "+s"(bid_), "+v"(ln_));
    const uint32_t call = tp->call, nst = tp->n_streams;
    unsigned long long* const rs = tp->runstate;
    unsigned long long* const sg = tp->sg;
    const float dp = published ? dp_early : (ln_ < (int)SDRFM_CHAIN_FIX ? tp->dpow[ln_] : 0.0f);
    unsigned long long pv = pv_early;
    if (ln_ == 0) {
      if (!published) {                                          // (a last flush with nothing in it: the state as the earlier flushes left it)
        const unsigned long long w = ((unsigned long long)(call + 1u) << 32) | __builtin_bit_cast(unsigned, yrun);
        __hip_atomic_store(last_run ? sg + (size_t)((call + 1u) % SDRFM_CHAIN_SG_SLOTS) * nst + st_ : rs + bid_, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned long long* const src = run == 0 ? sg + (size_t)(call % SDRFM_CHAIN_SG_SLOTS) * nst + st_ : rs + (bid_ - 1u);
      const uint32_t want = run == 0 ? call : call + 1u;
      unsigned long long zero = 0ull;
      asm volatile("" : "+v"(zero));                             // (opaque: "add 0" is a read-modify-write the compiler would turn back into a load, and a load may hit a stale line)
      // (the predecessor ends when this run does: a few polls at most.  The wait is BOUNDED — 2^19 polls, about a second —: a protocol error must not hang the
      // machine; it is reported through the sink instead — sdrfm_pcm_sink_synchronize / _get_state answer SDRFM_FAIL — and the run goes on from state 0)
      int it = 0;
      for (;; ++it) {
        if (it > 0 || !published) pv = __hip_atomic_fetch_add(src, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the first answer is the early poll's)
        if ((uint32_t)(pv >> 32) == want || it == (1 << 19)) break;
        __builtin_amdgcn_s_sleep(16);
      }
      if (it == (1 << 19)) { pv = 0ull; __hip_atomic_store(tp->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    const float carry = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)(unsigned)pv));
    const int nfix = (j1 - jlo < (int)SDRFM_CHAIN_FIX) ? j1 - jlo : (int)SDRFM_CHAIN_FIX;
    if (ln_ < nfix) {
      const float yv = __builtin_fmaf(dp, carry, pstash[ln_]);
      reinterpret_cast<unsigned*>(tp->pcm + (size_t)st_ * tp->pcm_stride)[jlo + ln_] = q_pcm_word(yv * tp->gain);
    }
  }
  const KargPtr pe = kargs();
  if (pe->n_repaired && lane == 0) {
    const uint2 cnt = *reinterpret_cast<const uint2*>(fl16 + 128);
    if (cnt.y) {
      atomicAdd(pe->n_repaired, cnt.x);
      atomicAdd(pe->n_repaired + 1, cnt.y);
      if (pe->stream_pass) atomicAdd(pe->stream_pass + stream, cnt.y);   // per stream (device memory, one address per stream: only waves that did repair)
    }
  }

  // ---- state hand-over by the wave that holds the end of the stream's chunk --------------------------------------------------
  if (last_run && !(SDRFM_Q_HO_RING && !ALIGNED)) {               // (linear rings: copied from the ring when the last step began)
    const unsigned char* row = pe->iq + (size_t)stream * pe->iq_stride;
    const int HTe = (int)pe->T - 1;
    const size_t Ne = pe->N;
    for (int k = lane; k < HTe; k += 64) {
      const unsigned raw = *reinterpret_cast<const unsigned short*>(row + 2 * (Ne - HTe + k));
      reinterpret_cast<unsigned short*>(pe->hist_b_out)[(size_t)stream * HTe + k] = (unsigned short)raw;
      pe->hist_x_out[(size_t)stream * HTe + k] = make_float2((float)(raw & 0xffu) - 127.5f, (float)(raw >> 8) - 127.5f);
    }
    if (lane < 2 * QTP / 16)                                    // the last 64 raw samples, for the next call's repair path
      *reinterpret_cast<qi4_t*>(pe->hist_q_out + (size_t)(2 * QTP) * stream + 16 * lane) =
          *reinterpret_cast<const qi4_t*>(row + 2 * Ne - 2 * QTP + 16 * lane);
  }
#ifdef SDRFM_Q_STAMPS
  if (tsp && lane == 0) {
    tsp[0] = t_entry; tsp[1] = t_first; tsp[2] = t_loop; tsp[3] = __builtin_amdgcn_s_memrealtime(); tsp[4] = t_wait;
#ifdef SDRFM_Q_PHASES
    for (int i = 0; i < 8; ++i) tsp[8 + i] = t_ph[i];
#endif
    tsp[5] = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8);
    tsp[6] = (unsigned long long)nsteps; tsp[7] = __builtin_readcyclecounter() - c_entry;   // word 5: XCC_ID | HW_ID << 8
  }
#endif
}


template <int C0, int NSLOT, int D, int DA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(D == 16 ? 3 : 4))) k_mfir(SdrfmQParams p) {
  mfir_body<C0, NSLOT, D, DA>(p, blockIdx.x);
}

// The same launch with the PCM sink's chain in it (round 6, sdrfm_sink_chain.h): every run sinks its own audio outputs.  A kernel of its own, so that the
// instruction stream of k_mfir does not change by a byte.
template <int C0, int NSLOT, int D, int DA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(D == 16 ? 3 : 4))) k_mfir_pcm(QPcmArgs a) {
  mfir_body<C0, NSLOT, D, DA, true>(a.p, blockIdx.x);
}

// =================================================================================================================
//  One launch for a mixed batch (round 5, DESIGN.md 4.Q "Routing"): the first `nb` one-wave workgroups run design B (sdrfm_b.h: the bit-exact
//  kernel with the smallest tile, R = 4) over the noise-only streams of b.slist, the others design Q over the streams of q.slist.  Two launches
//  on one queue run one after the other (the barrier bit), two queues cost a dependence between queues (~10 - 18 us each: profiles/
//  r05_mixed_batches.txt); one grid holds both kinds of wave on every CU from the first microsecond.  The long waves (design B: about twice
//  design Q's work per stream) have the low workgroup numbers, so they are dispatched first.  Each body is the one its own kernel runs: the
//  bits a stream gets do not depend on what shares the launch.
// =================================================================================================================
template <int C0, int NSLOT, int D, int DA, int T, int R>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) k_mix(SdrfmQParams q, CallParams b, uint32_t nb) {
  if (blockIdx.x < nb) fastb_body<T, D, R, (int)SDRFM_Q_TA, DA, 0>(b, blockIdx.x);
  else mfir_body<C0, NSLOT, D, DA>(q, blockIdx.x - nb);
}

// ... with the PCM sink's chain in the design-Q workgroups (the clean streams; the noisy streams' audio is sunk by the sink's own list kernel behind this launch:
// sdrfm.hip).  The argument's first member is k_mfir_pcm's argument: the body reads both parts through the same offsets.
struct MixPcmArgs { QPcmArgs qa; CallParams b; uint32_t nb; };
template <int C0, int NSLOT, int D, int DA, int T, int R>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) k_mix_pcm(MixPcmArgs a) {
  if (blockIdx.x < a.nb) fastb_body<T, D, R, (int)SDRFM_Q_TA, DA, 0>(a.b, blockIdx.x);
  else mfir_body<C0, NSLOT, D, DA, true>(a.qa.p, blockIdx.x - a.nb);
}

// y[-1] as the definition has it, from the 64 raw samples before the next call (hist_q) — for a bit-exact kernel that takes over from
// design Q (whose own carried y[-1] is only within 1e-4 of it).  One lane per stream; the chain of sdrfm_math.h / DESIGN.md "Frozen spec".
__global__ void __launch_bounds__(64) k_q_fix_yprev(const uint8_t* hist_q, const float* hpad, float2* yprev, uint32_t n_streams, const uint32_t* list) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_streams) return;
  const uint32_t s = list ? list[i] : i;
  const uint8_t* b = hist_q + (size_t)(2 * QTP) * s;
  float ar = 0.0f, ai = 0.0f;
  for (int i = 0; i < QTP; ++i) {                               // oldest sample first: sample -QTP + i meets tap QTP - 1 - i
    const float t = hpad[QTP - 1 - i];
    ar = __builtin_fmaf(t, (float)b[2 * i] - 127.5f, ar);
    ai = __builtin_fmaf(t, (float)b[2 * i + 1] - 127.5f, ai);
  }
  yprev[s] = make_float2(ar, ai);
}

// What the memory system delivers to design Q's access pattern with nothing else going on: every wave streams its own contiguous
// region through the same LDS-DMA ring (1 KiB per instruction, NSLOT KiB in flight, nt), reads one word per chunk, computes nothing.
// The measured ceiling bench.py prints beside the 8 TB/s specification ("the fraction against both").
template <int NSLOT>
__global__ void __launch_bounds__(64) k_q_read_stream(const uint8_t* base, unsigned total_kib, unsigned kib_per_wave, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned w = blockIdx.x, lane = threadIdx.x;
  const unsigned long long ga = (unsigned long long)base + ((unsigned long long)w * kib_per_wave << 10);
  const unsigned left = total_kib > w * kib_per_wave ? total_kib - w * kib_per_wave : 0u;
  const unsigned mine = left < kib_per_wave ? left : kib_per_wave;
  const qi4_t rsrc = {(int)(unsigned)ga, (int)(unsigned)(ga >> 32), (int)(mine << 10), 0x00020000};   // past the region: out of range, no traffic
  auto slot_ptr = [&](int slot) { return (__attribute__((address_space(3))) void*)(smem + 1024 * slot); };
  unsigned issued = 0;
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) { q_raw_buffer_load_lds(rsrc, slot_ptr(q), 16, (int)((issued << 10) + 16 * lane), 0, 0, SDRFM_Q_AUX); ++issued; }
  unsigned acc = 0;
  int slot = 0;
  for (unsigned q = 0; q < kib_per_wave; ++q) {
    wait_vmcnt<NSLOT - 1>();                                    // the oldest chunk has landed
    acc += *reinterpret_cast<const unsigned*>(smem + 1024 * slot + 4 * lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    q_raw_buffer_load_lds(rsrc, slot_ptr(slot), 16, (int)((issued << 10) + 16 * lane), 0, 0, SDRFM_Q_AUX);
    ++issued;
    slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
  }
  wait_vmcnt<0>();
  if (acc == 0x12345678u) sink[0] = acc;
}

typedef void (*QKernel)(SdrfmQParams);
typedef void (*QPcmKernel)(QPcmArgs);
struct QVariant { uint32_t c0, nslot, d, da, lds; QKernel k; const char* name; QPcmKernel kp; };
// (the kernel with the PCM chain: for the ring sizes the library launches — sdrfm_q_default_nslot —, not for the experiments' 10 and 15)
#define QV(C0_, NS_) { C0_, NS_, 10, 5, q_lds<10, 5, NS_>(), k_mfir<C0_, NS_, 10, 5>, "k_mfir<" #C0_ "," #NS_ ">", NS_ == 5 ? k_mfir_pcm<C0_, 5, 10, 5> : nullptr }
#define QVD(C0_, NS_, D_, DA_) { C0_, NS_, D_, DA_, q_lds<D_, DA_, NS_>(), k_mfir<C0_, NS_, D_, DA_>, "k_mfir<" #C0_ "," #NS_ "," #D_ "," #DA_ ">", k_mfir_pcm<C0_, NS_, D_, DA_> }
// D = 10 / DA = 5: the 2.4 MS/s front end of BASELINE (ring of 5 KiB; 10 and 15 for experiments).  D = 8 / DA = 8: 2.048 MS/s -> 256 kS/s ->
// 32 kHz (ring of two 2-KiB steps).  D = 16 / DA = 5: 3.2 MS/s -> 200 kS/s -> 40 kHz (ring of two 4-KiB steps).
const QVariant kQVariants[] = {QV(0, 5), QV(0, 10), QV(0, 15), QV(1, 5), QV(1, 10), QV(1, 15), QVD(0, 4, 8, 8), QVD(1, 4, 8, 8), QVD(0, 8, 16, 5), QVD(1, 8, 16, 5)};

const QVariant* q_find(uint32_t c0, uint32_t nslot, uint32_t d, uint32_t da) {
  if (c0 > 1) c0 = 1;
  for (const QVariant& v : kQVariants)
    if (v.c0 == c0 && v.nslot == nslot && v.d == d && v.da == da) return &v;
  return nullptr;
}

// ---- the one-launch kernel of a mixed batch: every shape both designs serve (where design B has no instance the host launches two kernels) ----------
typedef void (*MixKernel)(SdrfmQParams, CallParams, uint32_t);
typedef void (*MixPcmKernel)(MixPcmArgs);
struct MixVariant { uint32_t c0, nslot, d, da, T, R, lds; MixKernel k; MixPcmKernel kp; uint32_t lds_pcm; };
template <int T, int D, int R, int TA>
constexpr uint32_t b_lds() { return (uint32_t)fastb_xbytes(T, D, R) + 4u * (uint32_t)(((TA - 1 + 3) & ~3) + fastb_ab(R) * 64 * R + T + TA); }   // as sdrfm.hip sizes design B's workgroup
template <int C0, int NSLOT, int D, int DA, int T, int R>
constexpr uint32_t mix_lds() { return q_lds<D, DA, NSLOT>() > b_lds<T, D, R, (int)SDRFM_Q_TA>() ? q_lds<D, DA, NSLOT>() : b_lds<T, D, R, (int)SDRFM_Q_TA>(); }
template <int C0, int NSLOT, int D, int DA, int T, int R>
constexpr uint32_t mix_lds_pcm() {   // (design Q's workgroups keep SDRFM_CHAIN_FIX words behind their own bytes)
  return q_lds<D, DA, NSLOT>() + 4u * SDRFM_CHAIN_FIX > b_lds<T, D, R, (int)SDRFM_Q_TA>() ? q_lds<D, DA, NSLOT>() + 4u * SDRFM_CHAIN_FIX : b_lds<T, D, R, (int)SDRFM_Q_TA>();
}
#define MV(C0_, NS_, D_, DA_, T_, R_) { C0_, NS_, D_, DA_, T_, R_, mix_lds<C0_, NS_, D_, DA_, T_, R_>(), k_mix<C0_, NS_, D_, DA_, T_, R_>, \
                                        k_mix_pcm<C0_, NS_, D_, DA_, T_, R_>, mix_lds_pcm<C0_, NS_, D_, DA_, T_, R_>() }
// (design B's tile: R = 4 everywhere; R = 8 — 12.9 KB of LDS — measured no faster at the BASELINE shape)
const MixVariant kMixVariants[] = {MV(0, 5, 10, 5, 64, 4), MV(1, 5, 10, 5, 64, 4), MV(0, 5, 10, 5, 32, 4), MV(1, 5, 10, 5, 32, 4), MV(0, 5, 10, 5, 16, 4), MV(1, 5, 10, 5, 16, 4),
                                   MV(0, 4, 8, 8, 64, 4), MV(1, 4, 8, 8, 64, 4), MV(0, 4, 8, 8, 16, 4), MV(1, 4, 8, 8, 16, 4), MV(0, 8, 16, 5, 64, 4), MV(1, 8, 16, 5, 64, 4)};

const MixVariant* mix_find(uint32_t c0, uint32_t nslot, uint32_t d, uint32_t da, uint32_t T, uint32_t R) {
  if (c0 > 1) c0 = 1;
  for (const MixVariant& v : kMixVariants)
    if (v.c0 == c0 && v.nslot == nslot && v.d == d && v.da == da && v.T == T && v.R == R) return &v;
  return nullptr;
}

}  // namespace

uint32_t sdrfm_q_mix_lds(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, uint32_t T, uint32_t R) {
  const MixVariant* v = mix_find(first_chunk, nslot, d, da, T, R);
  return v ? v->lds : 0u;
}

int sdrfm_q_mix_blocks_per_cu(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, uint32_t T, uint32_t R) {
  const MixVariant* v = mix_find(first_chunk, nslot, d, da, T, R);
  int nb = 0;
  if (!v || hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(v->k), 64, v->lds) != hipSuccess) return 0;
  return nb;
}

hipError_t sdrfm_q_launch_mix(const SdrfmQParams& q, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, const CallParams& b, uint32_t b_blocks,
                              uint32_t b_R, hipStream_t stream, hipEvent_t done) {
  const MixVariant* v = mix_find(first_chunk, nslot, d, da, b.T, b_R);
  if (!v || b.Ta != SDRFM_Q_TA || b.Da != da || b.D != d) return hipErrorInvalidValue;
  hipExtLaunchKernelGGL(v->k, dim3(b_blocks + q.n_streams * q.runs), dim3(64), v->lds, stream, nullptr, done, 0, q, b, b_blocks);
  return hipGetLastError();
}

hipError_t sdrfm_q_launch_mix_pcm(const SdrfmQParams& q, const SdrfmSinkChain& t, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, const CallParams& b,
                                  uint32_t b_blocks, uint32_t b_R, hipStream_t stream, hipEvent_t done) {
  const MixVariant* v = mix_find(first_chunk, nslot, d, da, b.T, b_R);
  if (!v || b.Ta != SDRFM_Q_TA || b.Da != da || b.D != d) return hipErrorInvalidValue;
  const MixPcmArgs a = {{q, t}, b, b_blocks};
  hipExtLaunchKernelGGL(v->kp, dim3(b_blocks + q.n_streams * q.runs), dim3(64), v->lds_pcm, stream, nullptr, done, 0, a);
  return hipGetLastError();
}

uint32_t sdrfm_q_default_nslot(uint32_t d) { return d == 10 ? 5u : (d == 8 ? 4u : (d == 16 ? 8u : 0u)); }

bool sdrfm_q_geometry_ok(uint32_t d, uint32_t da) { return q_find(0, sdrfm_q_default_nslot(d), d, da) != nullptr; }

uint32_t sdrfm_q_lds_bytes(uint32_t nslot, uint32_t d, uint32_t da) {
  const QVariant* v = q_find(0, nslot, d, da);
  return v ? v->lds : 0u;
}

const char* sdrfm_q_kernel_symbol(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da) {
  const QVariant* v = q_find(first_chunk, nslot, d, da);
  return v ? v->name : "";
}

int sdrfm_q_blocks_per_cu(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da) {
  const QVariant* v = q_find(first_chunk, nslot, d, da);
  int nb = 0;
  if (!v || hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(v->k), 64, v->lds) != hipSuccess) return 0;
  return nb;
}

hipError_t sdrfm_q_fix_yprev(const uint8_t* hist_q, const float* hpad, float2* yprev, uint32_t n_streams, const uint32_t* list, hipStream_t stream) {
  hipLaunchKernelGGL(k_q_fix_yprev, dim3((n_streams + 63) / 64), dim3(64), 0, stream, hist_q, hpad, yprev, n_streams, list);
  return hipGetLastError();
}

hipError_t sdrfm_q_launch(const SdrfmQParams& p, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, hipStream_t stream, hipEvent_t done) {
  const QVariant* v = q_find(first_chunk, nslot, d, da);
  if (!v) return hipErrorInvalidValue;
  if (done) hipExtLaunchKernelGGL(v->k, dim3(p.n_streams * p.runs), dim3(64), v->lds, stream, nullptr, done, 0, p);
  else hipLaunchKernelGGL(v->k, dim3(p.n_streams * p.runs), dim3(64), v->lds, stream, p);
  return hipGetLastError();
}

bool sdrfm_q_has_pcm_chain(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da) {
  const QVariant* v = q_find(first_chunk, nslot, d, da);
  return v && v->kp;
}

hipError_t sdrfm_q_launch_pcm(const SdrfmQParams& p, const SdrfmSinkChain& t, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, hipStream_t stream,
                              hipEvent_t done) {
  const QVariant* v = q_find(first_chunk, nslot, d, da);
  if (!v || !v->kp) return hipErrorInvalidValue;
  const QPcmArgs a = {p, t};
  const uint32_t lds = v->lds + 4u * SDRFM_CHAIN_FIX;             // (+ the run's first outputs, kept until its predecessor's state is there)
  if (done) hipExtLaunchKernelGGL(v->kp, dim3(p.n_streams * p.runs), dim3(64), lds, stream, nullptr, done, 0, a);
  else hipLaunchKernelGGL(v->kp, dim3(p.n_streams * p.runs), dim3(64), lds, stream, a);
  return hipGetLastError();
}

// Test / measurement hook (include/sdrfm_dev.h): read-only LDS-DMA stream over nbufs device buffers of bytes_each bytes, `passes` passes
// taken in turn over the buffers (so that a pass reads cold HBM when the buffers together exceed the Infinity Cache), timed with HIP
// events on a stream of its own.  *gbytes_per_s = bytes read / elapsed.  Computes nothing on behalf of a caller.
extern "C" int sdrfm_debug_read_ceiling(int device, const void* const* bufs, uint32_t nbufs, size_t bytes_each, uint32_t passes, double* gbytes_per_s) {
  if (!bufs || !nbufs || !gbytes_per_s || bytes_each < (1u << 20) || bytes_each >= (1ull << 41) || !passes) return 16;   // SDRFM_EINVAL
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return 19;   // SDRFM_NO_DEVICE
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 19;
  const unsigned waves = 12u * (unsigned)prop.multiProcessorCount;          // design Q's residency
  const unsigned total_kib = (unsigned)(bytes_each >> 10), kpw = (total_kib + waves - 1) / waves;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  unsigned* sink = nullptr;
  int rc = 2;                                                               // SDRFM_FAIL
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
      hipMalloc(&sink, 64) == hipSuccess) {
    auto pass = [&](uint32_t k) {
      hipLaunchKernelGGL(k_q_read_stream<5>, dim3(waves), dim3(64), 5 * 1024, st, static_cast<const uint8_t*>(bufs[k % nbufs]), total_kib, kpw, sink);
    };
    for (uint32_t k = 0; k < (passes < 5 ? passes : 5u); ++k) pass(k);
    float ms = 0.0f;
    if (hipStreamSynchronize(st) == hipSuccess && hipEventRecord(e0, st) == hipSuccess) {
      for (uint32_t k = 0; k < passes; ++k) pass(k);
      if (hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess &&
          hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.0f) {
        *gbytes_per_s = (double)(total_kib) * 1024.0 * passes / (ms * 1e-3) * 1e-9;
        rc = 0;
      }
    }
  }
  if (sink) (void)hipFree(sink);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}
