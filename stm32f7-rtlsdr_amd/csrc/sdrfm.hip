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
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include "../../include/sdrfm.h"
#include "../../include/sdrfm_dev.h"
#include "sdrfm_math.h"
#include "sdrfm_q.h"

#include "sdrfm_b.h"

namespace {

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
    const uint32_t stream = launch_stream(p, blockIdx.x / p.tiles_per_stream);
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
    __syncthreads();  // taps are in LDS
    state_handover(p, launch_stream(p, blockIdx.x - n_tile_blocks), xs, NX_MAX, ys, hs);
  }
}

// =================================================================================================================
//  Fast kernel: compile-time (T, D, R).  One WAVE per block, one contiguous segment of one stream per wave, processed
//  as a sequence of sub-tiles of 64*R decimated outputs with everything carried in LDS/registers between sub-tiles:
//
//    HBM --typed buffer_load_format_xyzw (u8x4 -> 4 x f32 in the texture unit, 256 B per wave-instruction)--> VGPRs
//        (prefetched one sub-tile ahead, in flight during the previous sub-tile's FIR)
//    VGPR -0x1.fep+6 (= -127.5f)--> LDS x-tile, f32x2 per sample, rows of R*D samples (+16 B pad: conflict-free b128)
//    LDS --ds_read_b128--> sliding window of lane t (its R consecutive outputs share (R-1)*D+T samples)
//        v_pk_fma_f32 acc(I,Q) += tap * x(I,Q), tap broadcast from an SGPR pair: 64 taps live in 64 SGPRs
//    y --DPP/bpermute neighbour--> conj product --> atan2 --> LDS d ring --> audio FIR --> HBM
//
//  Why this shape (measured on MI355X, tools/ubench): v_pk_fma_f32 with an SGPR tap runs at the full fp32 rate while
//  a scalar v_fma_f32 with an SGPR operand runs at half rate; v_cvt_f32_ubyteN is a half-rate op but the typed buffer
//  load converts for free at the full streaming rate (6.6 TB/s of u8 in); a single wave issues VALU at <= ~60 % of peak,
//  so two resident waves per SIMD are needed, which bounds the LDS tile to ~23 KiB per wave.
//
//  Requirements (else the generic kernel runs): T even, T >= D, D even, Ta-1 <= 64*R, decimator phase even,
//  iq 4-byte aligned with 4-byte-multiple stride.
// =================================================================================================================
#ifdef SDRFM_DEV   // design A and every MODE != 0 instantiation exist only in libsdrfm_dev.so
// MODE 0 = product kernel.  Other modes exist for timing experiments only and are reachable only through the
// SDRFM_PHASE_PROFILE / SDRFM_ABLATE environment variables: 1 = per-phase cycle counters; 2 = no LDS staging writes;
// 3 = no FIR arithmetic; 4 = no discriminator/audio; 5 = no LDS staging writes and no FIR LDS reads; 6 = staging only;
// 7 = staging only and without the typed loads.
template <int T, int D, int R, int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_fast(CallParams p) {
  constexpr bool PROF = (MODE == 1);
  constexpr int RD = R * D, NYT = 64 * R, NST = 64 * RD, HP = T - D, NW = RD + HP;
  constexpr int ROWPAD = ((RD / 2) % 2 == 0) ? 2 : 0, RS = (RD + ROWPAD) * 8;
  constexpr int XBYTES = fast_xbytes(T, D, R);
  constexpr int NLOAD = RD / 2;
  constexpr int QSTEP = RD / cgcd(128, RD);          // loads q and q+QSTEP land a whole number of rows apart
  constexpr int QROWS = 128 * QSTEP / RD;
  static_assert(T % 2 == 0 && D % 2 == 0 && T >= D && NLOAD % QSTEP == 0 && NW % 2 == 0 && HP % 2 == 0, "geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xb = smem;                                   // x tile: position u <-> sub-tile sample s' = u - HP
  const uint32_t Ta = p.Ta, Da = p.Da;
  const uint32_t DOFF = (Ta - 1 + 3u) & ~3u;                  // d ring: [DOFF-(Ta-1), DOFF) history | [DOFF, DOFF+AB*NYT) new
  const int DCAP = (int)p.AB * NYT;                           // new d's buffered before the audio stage runs
  float* dbuf = reinterpret_cast<float*>(smem + XBYTES);
  float* gs = dbuf + DOFF + DCAP;                             // audio taps, reversed
  float* hs = gs + Ta;                                        // FIR taps (state blocks only)
  const int lane = (int)threadIdx.x;

  const uint32_t n_seg_blocks = p.n_streams * p.tiles_per_stream;
  if (blockIdx.x >= n_seg_blocks) {
    for (uint32_t k = lane; k < (uint32_t)T; k += 64) hs[k] = p.h[k];
    __syncthreads();
    float2* ys = reinterpret_cast<float2*>(dbuf);             // Ta+1 float2 fit: checked on the host
    state_handover(p, launch_stream(p, blockIdx.x - n_seg_blocks), reinterpret_cast<float2*>(xb), XBYTES / 8, ys, hs);
    return;
  }
  const uint32_t sblk = blockIdx.x;
  const uint32_t stream = launch_stream(p, sblk / p.tiles_per_stream);
  const uint32_t seg = sblk % p.tiles_per_stream;
  const int j0 = (int)(seg * p.NA);
  int j1 = j0 + (int)p.NA;
  if (j1 > (int)p.A) j1 = (int)p.A;
  if (j0 >= j1) return;
  for (uint32_t k = lane; k < Ta; k += 64) gs[k] = p.g[Ta - 1 - k];   // gs[k] multiplies the k-th oldest d

  // ---- segment geometry -------------------------------------------------------------------------------------
  int ibase = p.f0 + j0 * (int)Da - (int)(Ta - 1) - 1;        // first output computed here (its d is not used)
  const bool use_hist = ibase <= 0;                           // segment starts at the call start: take old state
  if (use_hist) ibase = 0;
  const int i_end = p.f0 + (j1 - 1) * (int)Da;                // newest d needed
  const int nst = (i_end - ibase) / NYT + 1;                  // sub-tiles
  int cs = (int)D * ibase - (int)p.phase_x;                   // chunk index of sub-tile sample s' = 0 (even)

  // ---- taps: T/2 wave-uniform pairs -> SGPRs ------------------------------------------------------------------
  f2_t hp[T / 2];
#pragma unroll
  for (int k = 0; k < T / 2; ++k) hp[k] = f2_t{p.h[2 * k], p.h[2 * k + 1]};

  // ---- per-lane LDS addresses (computed once) -----------------------------------------------------------------
  const unsigned char* win = xb + lane * RS;                  // lane window = positions [RD*lane, RD*lane + NW)
  unsigned char* wr[QSTEP];                                   // staging destinations of loads q = 0..QSTEP-1
#pragma unroll
  for (int q = 0; q < QSTEP; ++q) {
    const int u = HP + 128 * q + 2 * lane;
    wr[q] = xb + (u / RD) * RS + (u % RD) * 8;
  }
  const int hu = NST + 2 * lane;                              // halo carry: lane < HP/2 copies positions hu,hu+1 -> 2*lane
  const unsigned char* hsrc = xb + (hu / RD) * RS + (hu % RD) * 8;
  unsigned char* hdst = xb + ((2 * lane) / RD) * RS + ((2 * lane) % RD) * 8;

  // ---- typed-load descriptor of this stream's chunk -----------------------------------------------------------
  const unsigned long long gaddr = (unsigned long long)(p.iq + (size_t)stream * p.iq_stride);
  const i4_t rsrc = {(int)(unsigned)gaddr, (int)(unsigned)(gaddr >> 32), (int)(2u * p.N), SDRFM_RSRC_U8X4_USCALED};

  f4_t pre[NLOAD];
#pragma unroll
  for (int q = 0; q < NLOAD; ++q) pre[q] = llvm_amdgcn_raw_buffer_load_format_v4f32(rsrc, 2 * cs + (64 * q + lane) * 4, 0, 0);

  // ---- prologue: halo of the first sub-tile, d history ------------------------------------------------------
  for (int u = lane; u < HP; u += 64) {
    const int c = cs - HP + u;
    const float2 x = (c >= -(int)(T - 1)) ? load_x(p, stream, c) : make_float2(0.f, 0.f);
    *reinterpret_cast<float2*>(xb + (u / RD) * RS + (u % RD) * 8) = x;
  }
  for (uint32_t k = lane; k < Ta - 1; k += 64)
    dbuf[DOFF - (Ta - 1) + k] = use_hist ? p.hist_d_in[(size_t)stream * (Ta - 1) + k] : 0.0f;
  f2_t carry = {0.f, 0.f};                                    // y[ibase-1]
  if (use_hist) { const float2 yp = p.yprev_in[stream]; carry = f2_t{yp.x, yp.y}; }
  int warm = 0, warm_sink = 0;
  int dpos = 0;                                               // new d's in the ring since the last audio flush
  int ibA = ibase;                                            // output index of ring position DOFF
  // PROF build only: shader cycles per phase (stage, fir, disc, audio, carry) accumulated in LDS (keeps SGPRs for the taps)
  unsigned* tph = reinterpret_cast<unsigned*>(hs);            // hs is unused by segment blocks
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

  for (int st = 0; st < nst; ++st) {
    // ---- stage: DC shift + write the prefetched sub-tile, then prefetch the next one -------------------------
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
      const f4_t v = pre[q] - 127.5f;
      if constexpr (MODE == 2 || MODE == 5) asm volatile("" ::"v"(v));
      else *reinterpret_cast<f4_t*>(wr[q % QSTEP] + (q / QSTEP) * QROWS * RS) = v;
    }
    cs += NST;
    if (st + 1 < nst && MODE != 7) {
#pragma unroll
      for (int q = 0; q < NLOAD; ++q) pre[q] = llvm_amdgcn_raw_buffer_load_format_v4f32(rsrc, 2 * cs + (64 * q + lane) * 4, 0, 0);
    }
    if (p.warm_ahead) {
      // L2 warm-up: one dword from each 128-B line of the sub-tile(s) after next (8 KiB per instruction, 1 VGPR).
      // The typed loads above then hit L2 instead of paying the full HBM round trip with only NLOAD KiB in flight.
      warm_sink ^= warm;
      warm = llvm_amdgcn_raw_buffer_load_i32(rsrc, 2 * (cs + (int)p.warm_ahead * NST) + lane * 128, 0, 0);
    }
    __syncthreads();
    if (st == 0 && cs - NST < 0) {                            // samples before the chunk start come from the old history
      const int nneg = -(cs - NST);
      for (int s = lane; s < nneg; s += 64) {
        const int u = HP + s;
        *reinterpret_cast<float2*>(xb + (u / RD) * RS + (u % RD) * 8) = load_x(p, stream, cs - NST + s);
      }
      __syncthreads();
    }
    SDRFM_TICK(0)
    // ---- K2: R outputs per lane, oldest sample first ---------------------------------------------------------
    // Software-pipelined by hand: the window is read in batches of NB ds_read_b128, batch b+1 is issued before the
    // FMAs of batch b (the compiler on its own keeps only ~2 reads in flight, which exposes the LDS latency).
    f2_t acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = f2_t{0.f, 0.f};
    constexpr int NRD = (MODE == 6 || MODE == 7) ? 0 : NW / 2;  // ds_read_b128 per lane (2 samples each)
    constexpr int NB = (NRD % 7 == 0) ? 7 : ((NRD % 6 == 0) ? 6 : ((NRD % 8 == 0) ? 8 : ((NRD % 5 == 0) ? 5 : 1)));
    constexpr int NBATCH = NRD / NB;
    f4_t wb[2][NB];
#pragma unroll
    for (int i = 0; i < (NRD ? NB : 0); ++i) {
      const int row = (2 * i) / RD, col = (2 * i) % RD;
      wb[0][i] = *reinterpret_cast<const f4_t*>(win + row * RS + col * 8);
    }
#pragma unroll
    for (int b = 0; b < NBATCH; ++b) {
      if (b + 1 < NBATCH) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int j2 = (b + 1) * NB + i;
          const int row = (2 * j2) / RD, col = (2 * j2) % RD;
          wb[(b + 1) & 1][i] = *reinterpret_cast<const f4_t*>(win + row * RS + col * 8);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int j2 = b * NB + i;
        const f4_t v = wb[b & 1][i];
        const f2_t x0 = {v.x, v.y}, x1 = {v.z, v.w};
        if constexpr (MODE == 3) { acc[j2 % R] += x0 + x1; continue; }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int p0 = 2 * j2 - r * D;                      // position of x0 in output r's window (0 = oldest)
          if (p0 >= 0 && p0 < T) {
            const int k = T - 1 - p0;
            if (k & 1) pk_fma_bcast<1>(acc[r], hp[k / 2], x0); else pk_fma_bcast<0>(acc[r], hp[k / 2], x0);
          }
          const int p1 = p0 + 1;
          if (p1 >= 0 && p1 < T) {
            const int k = T - 1 - p1;
            if (k & 1) pk_fma_bcast<1>(acc[r], hp[k / 2], x1); else pk_fma_bcast<0>(acc[r], hp[k / 2], x1);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PROF) asm volatile("" :: "v"(acc[0]), "v"(acc[R - 1]));
    SDRFM_TICK(1)
    // ---- K3: discriminator; y[m-1] of the lane's first output comes from the neighbour lane -------------------
    f2_t prev;
    prev.x = __shfl_up(acc[R - 1].x, 1);
    prev.y = __shfl_up(acc[R - 1].y, 1);
    if (lane == 0) prev = carry;
    carry.x = __shfl(acc[R - 1].x, 63);
    carry.y = __shfl(acc[R - 1].y, 63);
    float dv[R];
    if constexpr (MODE == 4 || MODE == 6 || MODE == 7) {
#pragma unroll
      for (int r = 0; r < R; ++r) dv[r] = acc[r].x + acc[r].y + prev.x;
    } else {
#pragma unroll
    for (int r = 0; r + 1 < R; r += 2) {
      const f2_t d2 = discriminate_pair(acc[r], r == 0 ? prev : acc[r - 1], acc[r + 1]);
      dv[r] = d2.x;
      dv[r + 1] = d2.y;
    }
    if constexpr (R % 2 == 1) {
      const f2_t pv = (R == 1) ? prev : acc[R - 2];
      dv[R - 1] = sdrfm_discriminate(acc[R - 1].x, acc[R - 1].y, pv.x, pv.y);
    }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) dbuf[DOFF + dpos + R * lane + r] = dv[r];
    dpos += NYT;
    const bool last = (st + 1 == nst);
    SDRFM_TICK(2)
    if ((dpos == DCAP || last) && ((MODE != 4 && MODE != 6 && MODE != 7) || last)) {
      __syncthreads();
      // ---- K4: audio outputs whose newest d lies in ring positions [DOFF, DOFF + dpos) -------------------------
      int jl = (ibA - p.f0 + (int)Da - 1);
      jl = jl > 0 ? jl / (int)Da : 0;                         // ceil((ibA - f0)/Da), clamped at 0
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
        o[j] = a0;
        if (jb < jh) o[jb] = a1;
        if (jc < jh) o[jc] = a2;
      }
      if (!last) {                                            // d history for the next batch
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
    // ---- carry the x halo for the next sub-tile -----------------------------------------------------------------
    if (!last) {
      if (lane < HP / 2) *reinterpret_cast<f4_t*>(hdst) = *reinterpret_cast<const f4_t*>(hsrc);
      __syncthreads();
    }
    SDRFM_TICK(4)
  }
  if (warm_sink == 0x7fffffff && p.dbg) p.dbg[7] = (unsigned long long)warm;   // keeps the warm-up loads alive
  if constexpr (PROF) {
    if (lane == 0 && p.dbg) {
      for (int i = 0; i < 5; ++i) atomicAdd(p.dbg + 8 * (blockIdx.x & 63) + i, (unsigned long long)tph[i]);
      atomicAdd(p.dbg + 8 * (blockIdx.x & 63) + 5, (unsigned long long)nst);
      atomicAdd(p.dbg + 8 * (blockIdx.x & 63) + 6, 1ull);
    }
  }
#undef SDRFM_TICK
}

#endif  // SDRFM_DEV


// =================================================================================================================
//  Fast kernel, design S ("streaming lanes"): every LANE owns a contiguous segment of L = NB*S*D samples of one stream and
//  walks it once, sample by sample; the wave's 64 segments are consecutive pieces of the same stream.
//
//    HBM --buffer_load_dwordx4 ... lds (LDS-DMA, no VGPRs): 10 lanes x 16 B = one lane-segment's next 80 samples-->
//        LDS ring, 2 stages x 64 lanes x 160 B (the transposition: coalesced 160-B runs in, one 16-B read per lane out)
//    LDS --ds_read_b128 (8 samples of the lane's own segment)--> v_cvt_f32_ubyte0..3 + v_pk_add_f32(-127.5): EVERY SAMPLE
//        IS CONVERTED ONCE (design B converts the (T-D)-sample overlap of neighbouring lanes 1.45 times)
//        --> v_pk_fma_f32 into the S = 8 rotating accumulators ("slots") of the outputs the sample belongs to:
//            slot s holds output m = s (mod S); at phase p = n mod S*D its tap is h[(D*s + D-1 - p) mod S*D] (or none);
//            the tap is wave-uniform (all lanes are in phase), broadcast from an SGPR/VGPR pair; the first tap of a chain
//            writes acc = fma(h, x, 0), so no accumulator is ever cleared; the chain order is the oracle's (oldest first)
//    y (own lane, no shuffles) --> packed conj-product + atan2, two outputs at a time --> 48 d's per lane in VGPRs
//    end of the walk: d's --> LDS (the ring is free by then), 32-tap audio FIR with lanes = audio outputs --> HBM
//
//  What a lane needs from before its segment (FIR history, y[m-1]) comes from one warm-up stage that runs only the chains
//  which complete inside the segment; what the audio FIR needs from before the wave's span (Ta-1 d's) comes from lane 0,
//  which re-walks the segment before the span and emits nothing (1/64 of the work) — or from the carried state when the
//  span starts the call.  No inter-wave communication, no barriers (one wave per workgroup), no halo carry.
//
//  Requirements (else design B): decimator phases 0, N a multiple of L and of D*Da, 16-byte aligned rows, T-1 real history
//  samples (as design B: the first call after a reset is patched by the generic kernel).
// =================================================================================================================

// acc = fma(tap, x, +0): the first (oldest) tap of a chain — no accumulator is ever cleared
template <int HI>
__device__ __forceinline__ void pk_first_bcast_v(f2_t& acc, f2_t tap_pair, f2_t x) {
  if constexpr (HI == 0)
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(acc) : "v"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(acc) : "v"(tap_pair), "v"(x));
}

#ifndef SDRFM_STREAM_AUX
#define SDRFM_STREAM_AUX 0   // cache policy bits of the ring's line fetches (2 = nt)
#endif
// line fetched by the refill at the last (lo = false) / first (lo = true) chunk u with u % 4 == 0 below `uend` resp. at or after
// the start of the last body of NCH = 10 chunks (helper of a static_assert)
constexpr int stream_refill_line_range(int uend, bool first_of_last_body) {
  int u = first_of_last_body ? ((uend - 10 + 3) / 4) * 4 : ((uend - 1) / 4) * 4;
  const int cls = (u >> 2) & 1;
  return ((u + 4 * cls) >> 3) + 1;
}

// The lane's view of the LDS ring: two 128-byte line slots per lane (slot stride 8 KiB, lane region `base`), read 16 bytes at a
// time.  t16 = 16 x the index of the next piece, counted from the start of the lane's first line; bit 7 of t16 is the slot, bits
// 4..6 the piece, XOR-swizzled per region so that the 16 lanes a ds_read_b128 serves together hit 16 different bank groups.
struct StreamRing {
  unsigned base, rot16, t16;
  __device__ __forceinline__ u4_t read(const unsigned char* smem) {
    const unsigned a = base + (((t16 ^ rot16) & 0x70u) | ((t16 & 0x80u) << 6));
    t16 += 16;
    return *reinterpret_cast<const u4_t*>(smem + a);
  }
};

// slots whose chain wraps from one body into the next (they are 0 .. NH-1): the outputs a lane cannot finish without the samples
// before its segment — its "head" outputs
template <int HI>
__device__ __forceinline__ void pk_first_bcast(f2_t& acc, f2_t tap_pair, f2_t x) {   // as pk_first_bcast_v, tap pair in SGPRs
  if constexpr (HI == 0)
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(acc) : "s"(tap_pair), "v"(x));
  else
    asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(acc) : "s"(tap_pair), "v"(x));
}

constexpr int stream_heads(int T, int D, int S) {
  const int P = S * D;
  int nh = 0;
  for (int sl = 0; sl < S; ++sl) {
    const int cs = ((D * sl + D - 1 - (T - 1)) % P + P) % P;
    if (cs + T - 1 >= P) ++nh;
  }
  return nh;
}

// One body = S*D consecutive samples of every lane's segment, 8 samples (one 16-byte piece, `read()`) per chunk.  `ub + c` is the
// index u of the piece chunk c uses, counted from the segment's first piece; the ring hooks hang on it:
//   before a piece u with u % 4 == 0 is read, `wait()`: a class of lanes may be about to start a line (even-segment lanes at
//     u % 8 == 0, odd-segment lanes at u % 8 == 4), which has to have landed;
//   when chunk u with u % 4 == 0 begins, `refill(u)`: one class of lanes has used up a line, the line after next is fetched into it.
// A slot's chain that starts in one body and completes in the next is split into its "head" (the phases in the first body) and
// its "tail"; the first NH slots wrap, so a segment's first NH outputs need the NH heads run on the samples BEFORE the segment:
//   FIRST : the segment's first body — tails are skipped (their heads have not run): outputs NH.. only; the first CB pieces are
//           kept (raw) for the HEADB pass
//   MID   : all chains, S outputs -> S discriminator values dn[]
//   LAST  : as MID, but heads are not started (they belong to the next lane's segment)
//   HEADA : after the walk, on the LEFT neighbour's last pieces (still in its ring slots): the heads of this lane's first outputs
//   HEADB : then on the lane's own first pieces (kept by FIRST): their tails -> outputs 0 .. NH-1 and d[0 .. NH]
constexpr int SBODY_FIRST = 0, SBODY_MID = 1, SBODY_LAST = 2, SBODY_HEADA = 3, SBODY_HEADB = 4;
constexpr int stream_kept_pieces(int T, int D, int S) { return (D * (stream_heads(T, D, S) - 1) + D - 1) / 8 + 1; }   // pieces HEADB reads
constexpr int stream_sgpr_pairs(int T) { return T >= 64 ? 8 : 0; }   // tap pairs (the last ones) held in SGPRs instead of VGPRs
template <int T, int D, int S, int MODE, class RD, class FW, class FR>
__device__ __forceinline__ void stream_body(RD&& read, int ub, FW&& wait, FR&& refill, const f2_t (&hp)[T / 2], f2_t (&acc)[S], f2_t& prev,
                                            float (&dn)[S], f2_t& ysave, u4_t (&kept)[stream_kept_pieces(T, D, S)]) {
  constexpr int P = S * D, NCH = P / 8, NH = stream_heads(T, D, S);
  constexpr int CS0 = ((D - 1 - (T - 1)) % P + P) % P;          // phase at which slot 0 starts its (wrapping) chain: the first head phase
  constexpr int CR = (MODE == SBODY_HEADA) ? CS0 / 8 : 0;      // first chunk that is read / computed
  constexpr int CE = (MODE == SBODY_HEADB) ? (D * (NH - 1) + D - 1) / 8 + 1 : NCH;   // one past the last chunk (HEADB: the last tail phase)
  static_assert(NH >= 1 && NH < S - 1, "design S geometry");
  if (((ub + CR) & 3) == 0) wait();
  u4_t cur = read(std::integral_constant<int, CR>{}), nxt = cur;
  f2_t x = cvt_iq<0>(cur.x);                                   // converted one sample ahead of its use: the FMAs (inline asm) never
  static_for<CR, CE>([&](auto CC) {                            // directly follow the instruction that produced their operand
    constexpr int c = decltype(CC)::value;
    const int u = ub + c;
    if ((u & 3) == 0 && u >= 4) refill(u);
    if constexpr (MODE == SBODY_FIRST && c < stream_kept_pieces(T, D, S)) kept[c] = cur;
    if constexpr (c + 1 < CE) {
      if (((u + 1) & 3) == 0) wait();
      nxt = read(std::integral_constant<int, c + 1>{});
    }
    __builtin_amdgcn_sched_barrier(0);                         // one 8-sample chunk is one scheduling region (bounds live ranges)
    static_for<0, 8>([&](auto S8) {
      constexpr int s8 = decltype(S8)::value;
      constexpr int ph = 8 * c + s8;
      f2_t xn = x;
      if constexpr (s8 < 7) {
        constexpr int t8 = s8 + 1;
        const unsigned w = (t8 / 2 == 0) ? cur.x : (t8 / 2 == 1) ? cur.y : (t8 / 2 == 2) ? cur.z : cur.w;
        xn = cvt_iq<(t8 & 1)>(w);
      } else if constexpr (c + 1 < CE) {
        xn = cvt_iq<0>(nxt.x);
      }
      static_for<0, S>([&](auto SS) {
        constexpr int sl = decltype(SS)::value;
        constexpr int e = ((D * sl + D - 1 - ph) % P + P) % P;  // tap index of slot sl at this phase (>= T: idle)
        constexpr int cs = ((D * sl + D - 1 - (T - 1)) % P + P) % P;   // phase at which slot sl starts a chain
        constexpr bool wraps = (cs + T - 1 >= P);              // that chain completes in the NEXT body
        constexpr bool head = wraps && ph >= cs;               // this phase starts a chain completing in the next body
        constexpr bool tail = wraps && ph <= D * sl + D - 1;   // this phase completes a chain started in the previous body
        constexpr bool run = (e < T) && ((MODE == SBODY_MID) || (MODE == SBODY_LAST && !head) || (MODE == SBODY_FIRST && !tail) ||
                                         (MODE == SBODY_HEADA && head) || (MODE == SBODY_HEADB && tail));
        if constexpr (run) {
          // the tap is wave-uniform: one half of a register pair (VGPR; the last few pairs SGPR), broadcast to both halves of the
          // pack by op_sel
          constexpr bool sg = e / 2 >= T / 2 - stream_sgpr_pairs(T);
          if constexpr (e == T - 1) {
            if constexpr (sg) { if constexpr (e & 1) pk_first_bcast<1>(acc[sl], hp[e / 2], x); else pk_first_bcast<0>(acc[sl], hp[e / 2], x); }
            else { if constexpr (e & 1) pk_first_bcast_v<1>(acc[sl], hp[e / 2], x); else pk_first_bcast_v<0>(acc[sl], hp[e / 2], x); }
          } else {
            if constexpr (sg) { if constexpr (e & 1) pk_fma_bcast<1>(acc[sl], hp[e / 2], x); else pk_fma_bcast<0>(acc[sl], hp[e / 2], x); }
            else { if constexpr (e & 1) pk_fma_bcast_v<1>(acc[sl], hp[e / 2], x); else pk_fma_bcast_v<0>(acc[sl], hp[e / 2], x); }
          }
        }
      });
      // outputs complete at phases D-1, 2D-1, ...: slot ph / D; every second one closes a pair for the packed discriminator.
      // Outputs 0 .. NH-1 of a segment (and with them d[0 .. NH]) are not available in its FIRST body: HEADB supplies them.
      if constexpr (MODE != SBODY_HEADA && ph % D == D - 1 && ((ph / D) & 1) == 1) {
        constexpr int s1 = ph / D, s0 = s1 - 1;
        if constexpr (MODE == SBODY_FIRST && s1 < NH) {
        } else if constexpr (MODE == SBODY_FIRST && s0 < NH) {     // s0 = NH-1, s1 = NH (NH odd): y[NH] waits for y[NH-1]
          ysave = acc[s1];
          prev = acc[s1];
        } else if constexpr (MODE == SBODY_FIRST && s0 == NH) {    // (NH even): d[NH] waits for y[NH-1], d[NH+1] is complete
          const f2_t d2 = discriminate_pair(acc[s0], acc[s0], acc[s1]);
          dn[s1] = d2.y;
          ysave = acc[s0];
          prev = acc[s1];
        } else if constexpr (MODE == SBODY_HEADB && s1 >= NH) {    // (the pair that closes at output NH lies beyond HEADB's last
        } else {                                                   // chunk: it is evaluated after the loop, below)
          const f2_t d2 = discriminate_pair(acc[s0], prev, acc[s1]);
          dn[s0] = d2.x;
          dn[s1] = d2.y;
          prev = acc[s1];
        }
      }
      x = xn;
    });
    cur = nxt;
  });
  if constexpr (MODE == SBODY_HEADB && (NH % 2) == 0) {        // d[NH] = K3(y[NH] kept by FIRST | y[NH-1])
    const f2_t d2 = discriminate_pair(ysave, prev, ysave);
    dn[NH] = d2.x;
  }
  if constexpr (MODE == SBODY_HEADB && (NH % 2) == 1) {        // NH odd (T = 32: NH = 3): the loop ends before the phase that would close
    const f2_t d2 = discriminate_pair(acc[NH - 1], prev, ysave);   // the pair (y[NH-1], y[NH]): prev is y[NH-2] (y[-1] when NH = 1), ysave is
    dn[NH - 1] = d2.x;                                          // y[NH], kept by FIRST
    dn[NH] = d2.y;
  }
}

template <int T, int D, int S, int NB, int TA, int DA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_stream(CallParams p) {
  constexpr int P = S * D, L = NB * P, OPL = NB * S;           // samples per body / per lane segment, outputs per lane
  constexpr int NCH = P / 8;                                    // 16-byte pieces (8 samples) a body reads
  constexpr int SLOT = 64 * 128;                                // bytes per line slot of the ring (64 lanes x one 128-byte line)
  constexpr int NLINES = (NB * NCH + 4 + 7) / 8;                // lines a lane walks through (odd-segment lanes start mid-line)
  constexpr int NH = stream_heads(T, D, S);                     // head outputs per segment (deferred to the end of the walk)
  constexpr int CS0 = ((D - 1 - (T - 1)) % P + P) % P, NHA = NCH - CS0 / 8;   // pieces of the left neighbour the heads read
  static_assert(P % 8 == 0 && S % 2 == 0 && P >= T && T % 2 == 0 && (OPL % 4) == 0 && TA - 1 <= OPL, "design S geometry");
  static_assert((L * 2) % 128 == 64 && NHA <= 8, "design S: segment starts alternate between line starts and line middles; the heads fit the neighbour's last line(s)");
  static_assert((NB * NCH) % 8 == 4, "design S: a class-0 lane ends, and a class-1 lane starts, in the middle of a line (the half-line fetches rely on it)");
  // the ring schedule relies on: every refill due in the FIRST / MID bodies fetches a line that exists (vmcnt(4) then always
  // leaves exactly the youngest refill outstanding), and none is due in the LAST body (which waits for everything)
  static_assert(stream_refill_line_range((NB - 1) * NCH, false) < NLINES && stream_refill_line_range(NB * NCH, true) >= NLINES,
                "design S ring schedule");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = (int)threadIdx.x;
  const uint32_t stream = launch_stream(p, blockIdx.x / p.tiles_per_stream);
  const int w = (int)(blockIdx.x % p.tiles_per_stream);         // wave index inside the stream
  const int segs = (int)(p.N / L);                              // lane segments of this stream in this call
  const int g0 = 63 * w - 1;                                    // segment of lane 0 (-1: before the call)
  int nuse = segs - 63 * w;                                     // useful lanes 1..nuse
  if (nuse > 63) nuse = 63;

  f2_t hp[T / 2];                                               // taps: wave-uniform pairs, op_sel picks the half at each use; most live in
#pragma unroll                                                  // VGPRs (64 taps do not fit the SGPR file next to the kernel's scalars), the
  for (int k = 0; k < T / 2; ++k) hp[k] = f2_t{p.h[2 * k], p.h[2 * k + 1]};   // last few in SGPRs (VGPRs are the scarcer file here)
#pragma unroll
  for (int k = 0; k < T / 2; ++k) {
    if (k < T / 2 - stream_sgpr_pairs(T)) asm volatile("" : "+v"(hp[k]));
    else asm volatile("" : "+s"(hp[k]));
  }

  // ---- the ring: HBM --LDS-DMA--> LDS, whole 128-byte lines --------------------------------------------------------------
  // A lane's byte stream starts at its segment: at a line start for even segments (class 0), in the middle of a line for odd
  // ones (class 1: L*2 = 7.5 lines), so class 1 begins at piece 4 of its first line.  Lanes are grouped by class in LDS (region
  // rho = 32 class + lane / 2): one DMA instruction fills the current line slot of 8 regions of ONE class (lane t of the
  // instruction: region 8 i + t / 8, piece t % 8, fetched from column piece ^ swizzle), so a class is refilled the moment its
  // lanes cross a line boundary.  Lanes whose segment lies outside the call fetch (and never use) the row's first line; no
  // instruction is ever skipped, so vmcnt counts DMA instructions exactly.  The line an even segment shares with its odd right
  // neighbour is requested by both, half each; the L2 still fills whole 128-byte lines (TCC_EA0_RDREQ_128B did not move when the
  // halves were introduced), so the row crosses the fabric 16/15 times: 133.5 MB read for 122.9 MB of input.
  const unsigned long long gaddr = (unsigned long long)(p.iq + (size_t)stream * p.iq_stride);
  const i4_t rsrc = {(int)(unsigned)gaddr, (int)(unsigned)(gaddr >> 32), (int)(2u * p.N), 0x00020000};
  const int g0odd = g0 & 1;
  auto swz = [](int rho) { return ((rho >> 1) + 4 * (rho >> 5)) & 7; };
  int voffs[2][4];                                              // byte offset (in the row) of this lane's piece of line 0, per (class, i)
#pragma unroll
  for (int cls = 0; cls < 2; ++cls)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = 16 * i + 2 * (lane >> 3) + (cls ^ g0odd), g = g0 + l, rho = 32 * cls + 8 * i + (lane >> 3);
      const int col = (lane & 7) ^ swz(rho);
      voffs[cls][i] = ((g >= 0 && g < segs) ? g * (L * 2) - 64 * cls : 0) + 16 * col;
    }
  // half: 0 = whole lines; 1 / 2 = only the lower / upper 64 bytes of every line (the other lanes of the instruction are switched
  // off: the instruction still counts in vmcnt).  A class-1 lane starts at piece 4 of its line 0 and a class-0 lane ends with piece
  // 3 of its last line; the other half of those lines is the neighbour segment's, who fetches it itself.  The line offset goes in
  // the scalar offset operand (no vector add per instruction).
  auto fill = [&](int cls, int n, int half = 0) {               // line n of every lane of one class -> slot n & 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vo = cls ? voffs[1][i] : voffs[0][i];
      auto* dst = (__attribute__((address_space(3))) void*)(smem + (n & 1) * SLOT + (32 * cls + 8 * i) * 128);
      if (half == 0) llvm_amdgcn_raw_buffer_load_lds(rsrc, dst, 16, vo, 128 * n, 0, SDRFM_STREAM_AUX);
      else if ((half == 2) == ((vo & 64) != 0)) llvm_amdgcn_raw_buffer_load_lds(rsrc, dst, 16, vo, 128 * n, 0, SDRFM_STREAM_AUX);
    }
  };
  auto refill = [&](int u) {                                    // chunk u begins: class (u / 4) % 2 has just finished a line
    const int cls = (u >> 2) & 1, n = ((u + 4 * cls) >> 3) + 1;
    if (n < NLINES) {
      if (cls) fill(1, n);
      else if (n == NLINES - 1) fill(0, n, 1);
      else fill(0, n);
    }
  };
  auto wait4 = [] { __builtin_amdgcn_s_waitcnt(0x0f74); };      // vmcnt(4): everything but the youngest refill (4 instructions) has landed
  auto wait0 = [] { __builtin_amdgcn_s_waitcnt(0x0f70); };
  auto no_wait = [] {};
  auto no_refill = [](int) {};

#ifdef SDRFM_DEV   // development build: per-wave time stamps (shader cycles) at the phase boundaries, 32 words per wave
  unsigned long long* const tsp = (p.dbg && p.dbg_tag) ? p.dbg + 32 * (size_t)blockIdx.x : nullptr;
  int tsi = 0;
#define SDRFM_STAMP() do { if (tsp && lane == 0) tsp[tsi] = __builtin_readcyclecounter(); ++tsi; } while (0)
  if (tsp && lane == 0) { tsp[30] = __builtin_amdgcn_s_memrealtime(); tsp[29] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32); }
#else
#define SDRFM_STAMP() do { } while (0)
#endif
  SDRFM_STAMP();                                                // 0: entry
  fill(0, 0); fill(1, 0, 2); fill(1, 1);                        // class 1 starts mid-line: it needs its second line after 32 samples
  f2_t acc[S];
#pragma unroll
  for (int k = 0; k < S; ++k) acc[k] = f2_t{0.f, 0.f};
  f2_t prev = {0.f, 0.f}, ysave = {0.f, 0.f};
  float dn[S], dreg[OPL];
  u4_t kept[stream_kept_pieces(T, D, S)];
  const int mycls = (g0 + lane) & 1, myrho = 32 * mycls + (lane >> 1);
  StreamRing ring = {(unsigned)(myrho * 128), (unsigned)(16 * swz(myrho)), (unsigned)(64 * mycls)};
  auto rd = [&](auto) { return ring.read(smem); };
  __builtin_amdgcn_s_waitcnt(0x0f74);                           // vmcnt(4): the first line of both classes has landed
  fill(0, 1);                                                   // class 0's second line is not needed for 64 samples: it queues behind the
  SDRFM_STAMP();                                                // opening burst instead of lengthening it.  1: first lines in LDS
  stream_body<T, D, S, SBODY_FIRST>(rd, 0, wait4, refill, hp, acc, prev, dn, ysave, kept);
#pragma unroll
  for (int k = 0; k < S; ++k) dreg[k] = dn[k];                  // (d[0 .. NH] are placeholders until the head pass)
  SDRFM_STAMP();                                                // 2: first body done
  for (int b = 1; b < NB - 1; ++b) {
    // The two waves of a SIMD are arbitrated oldest-first: one runs ahead at the single-wave rate and the other finishes alone.
    // Priority falls with progress, so whichever is behind wins the issue slot and the pair finishes together.
    if (p.prio_balance) {
      const int left = NB - 1 - b;
      if (left >= 3) __builtin_amdgcn_s_setprio(3);
      else if (left == 2) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(1);
    }
    SDRFM_STAMP();                                              // 3, 5, 7, ...
    stream_body<T, D, S, SBODY_MID>(rd, NCH * b, wait4, refill, hp, acc, prev, dn, ysave, kept);
#pragma unroll
    for (int bb = 1; bb < NB - 1; ++bb)
      if (b == bb) {
        asm volatile("" ::: "memory");                          // keep this a scalar branch: S moves, not NB*S selects
#pragma unroll
        for (int k = 0; k < S; ++k) dreg[bb * S + k] = dn[k];
      }
    SDRFM_STAMP();                                              // 4, 6, 8, ...: body done
  }
  // From here on both waves of a SIMD would run at priority 0, where the arbiter prefers the wave in hardware slot 0: it ran ahead
  // at the single-wave rate and the slot-1 wave finished the kernel alone (3.8 us later, measured).  The slot-1 wave therefore
  // keeps priority 1 through the last body and the head pass: the pair then ends within 0.4 us of each other.
  const uint32_t endp = (p.end_prio >> ((__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1) ? 6 : 0)) & 63u;   // HW_ID.wave_id & 1
  auto set_end_prio = [&](uint32_t v) {
    if (!p.prio_balance) return;
    v &= 3u;
    if (v == 0) __builtin_amdgcn_s_setprio(0); else if (v == 1) __builtin_amdgcn_s_setprio(1);
    else if (v == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
  };
  set_end_prio(endp);
  // the last body: no line is left to fetch (the refills it would trigger lie beyond the lane's last line), so every wait is for
  // all outstanding requests
  stream_body<T, D, S, SBODY_LAST>(rd, NCH * (NB - 1), wait0, no_refill, hp, acc, prev, dn, ysave, kept);
#pragma unroll
  for (int k = 0; k < S; ++k) dreg[(NB - 1) * S + k] = dn[k];
  const f2_t ylast = prev;                                      // y of the segment's last output
  SDRFM_STAMP();                                                // last body done

  set_end_prio(endp >> 2);
  // ---- the heads: outputs 0 .. NH-1 of every segment, from the left neighbour's last NHA pieces (in its ring slots: nothing has
  // been fetched into them since) and the lane's own first pieces (kept raw by the first body) ---------------------------------
  {
    const int ll = lane > 0 ? lane - 1 : 0, lcls = (g0 + ll) & 1, lrho = 32 * lcls + (ll >> 1);
    if (w == 0) {
      // the first segment of the call (lane 1) has the carried raw history for a left neighbour: the last T-1 samples of "segment
      // -1" go where lane 0's ring region would hold them (lane 0's own walk ran on the row's first line: unused)
      __syncthreads();
      const int rho0 = 32 * ((g0 + 0) & 1), rot0 = 16 * swz(rho0);
      for (int k = lane; k < T - 1; k += 64) {
        const int sl = L - (T - 1) + k, t16 = 16 * ((sl >> 3) + 4 * ((g0 + 0) & 1));
        *reinterpret_cast<unsigned short*>(smem + rho0 * 128 + (((t16 ^ rot0) & 0x70) | ((t16 & 0x80) << 6)) + 2 * (sl & 7)) =
            reinterpret_cast<const unsigned short*>(p.hist_b_in)[(size_t)stream * (T - 1) + k];
      }
      __syncthreads();
    }
    StreamRing lring = {(unsigned)(lrho * 128), (unsigned)(16 * swz(lrho)), (unsigned)(16 * (NB * NCH - NHA + 4 * lcls))};
    auto lrd = [&](auto) { return lring.read(smem); };
    auto krd = [&](auto C) { return kept[decltype(C)::value]; };
    prev.x = __shfl_up(ylast.x, 1);                             // y[-1] of the segment = the left neighbour's last output
    prev.y = __shfl_up(ylast.y, 1);
    if (w == 0 && lane == 1) { const float2 yp = p.yprev_in[stream]; prev = f2_t{yp.x, yp.y}; }
    stream_body<T, D, S, SBODY_HEADA>(lrd, 0, no_wait, no_refill, hp, acc, prev, dn, ysave, kept);
    stream_body<T, D, S, SBODY_HEADB>(krd, 0, no_wait, no_refill, hp, acc, prev, dn, ysave, kept);
#pragma unroll
    for (int k = 0; k <= NH; ++k) dreg[k] = dn[k];
  }
  SDRFM_STAMP();                                                // heads done

  set_end_prio(endp >> 4);
  // ---- audio stage: d's of the whole span -> LDS (lane 0's segment first), lanes = audio outputs ---------------------------
  // All eight waves of a CU reach this stage together and it is LDS-bound, so the layout is chosen for the LDS: the d array is
  // linear in the d index with a 4-word gap after every second segment (position(i) = i + 4 (i / (2 OPL))), which spreads the
  // 16-byte stores of 8 neighbouring lanes (lane stride OPL words = 16 mod 32 banks otherwise: 4-way conflicts) over all banks;
  // the windows are read as aligned 8-byte pairs at an even lane stride of NOUT DA words (conflict-free at 256 B per clock; the
  // 4-byte reads they replace ran at 128 B per clock, 2-way conflicting).
  float* dl = reinterpret_cast<float*>(smem);
  constexpr int DLW = 64 * OPL + 4 * 32;                        // words of the gapped d array
  auto dpos = [](int i) { return i + 4 * (int)((unsigned)i / (unsigned)(2 * OPL)); };
  float* gs = dl + DLW;                                         // (TA words kept free: the window reads of the last lanes run into them)
  float gv[TA];                                                 // audio taps, reversed: wave-uniform scalar loads, issued before the d's
#pragma unroll                                                  // are written so that their latency hides behind the LDS stores
  for (int k = 0; k < TA; ++k) gv[k] = p.g[TA - 1 - k];
  __syncthreads();
  {
    float* myd = dl + lane * OPL + 4 * (lane >> 1);
#pragma unroll
    for (int k = 0; k < OPL; k += 4) *reinterpret_cast<f4_t*>(myd + k) = f4_t{dreg[k], dreg[k + 1], dreg[k + 2], dreg[k + 3]};
  }
  SDRFM_STAMP();                                                // d's written
  if (w == 0 && lane < TA - 1) dl[OPL - (TA - 1) + lane] = p.hist_d_in[(size_t)stream * (TA - 1) + lane];   // (before the first gap)
  __syncthreads();
  SDRFM_STAMP();                                                // history in LDS
  const int G0 = 63 * w * OPL, G1 = G0 + nuse * OPL;           // d's [G0, G1) belong to this wave; local index = d - G0 + OPL
  const int jl = G0 / DA, jh = G1 / DA;                         // audio outputs whose newest d lies in [G0, G1)
  // Every lane computes NOUT consecutive outputs from ND consecutive d's: all LDS reads are issued up front (nothing in the
  // chains waits on memory), the taps sit in registers, and the results go back through LDS so that the stores are coalesced.
  // The lanes start one output early where that makes the window's first d an even local index (the extra output belongs to
  // the previous wave and is not stored), so that every pair read is 8-byte aligned and never straddles a gap.
  constexpr int NOUT = ((63 * OPL + DA) / DA + 63) / 64, ND = (NOUT - 1) * DA + TA, NR = (ND + 1) / 2;
  static_assert((DA % 2) == 1 && ((NOUT * DA) % 2) == 0 && (OPL % 2) == 0 && OPL >= TA + DA - 1 && 2 * NR <= 2 * OPL,
                "audio stage of design S: even window starts, at most one gap inside a window");
  float* outl = gs + TA;                                        // 64 * NOUT results
  static_assert((DLW + TA + 64 * NOUT) * 4 <= 2 * SLOT && (DA * 64 * NOUT + TA + OPL + 4 * 34 + 2) * 4 <= 2 * SLOT, "audio stage scratch exceeds the ring");
  const int jlo = jl - ((DA * jl + DA - 1 - (TA - 1) - G0 + OPL) & 1);
  {
    const int j0 = jlo + lane * NOUT;
    // local index of the oldest d of output j0 (even).  The lanes around the last output read past the d array, into the taps
    // and the (not yet written) result area behind it: still inside the ring, and only into outputs that are never stored
    const int base = DA * j0 + DA - 1 - (TA - 1) - G0 + OPL;
    const int gq = (int)((unsigned)base / (unsigned)(2 * OPL)), kb = 2 * OPL * (gq + 1) - base;   // window elements >= kb lie behind the next gap
    const float* w0 = dl + base + 4 * gq;
    const float* w1 = w0 + 4;
    f2_t dw2[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) dw2[i] = *reinterpret_cast<const f2_t*>((2 * i >= kb ? w1 : w0) + 2 * i);
    float a[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) a[i] = 0.0f;
    SDRFM_STAMP();                                              // window reads issued
#pragma unroll
    for (int k = 0; k < TA; ++k)                               // NOUT independent chains side by side, each in the oracle's order
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {                         // (plain fmaf: the compiler packs pairs of chains and spends 120 v_mov on it)
        const int e = DA * i + k;
        if (e & 1) asm("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(gv[k]), "v"(dw2[e >> 1].y));
        else asm("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(gv[k]), "v"(dw2[e >> 1].x));
      }
#pragma unroll
    for (int i = 0; i < NOUT; i += 2) {
      if (i + 1 < NOUT) *reinterpret_cast<f2_t*>(outl + lane * NOUT + i) = f2_t{a[i], a[i + 1]};
      else outl[lane * NOUT + i] = a[i];
    }
  }
  __syncthreads();
  SDRFM_STAMP();                                                // results in LDS
  float* out = p.audio + (size_t)stream * p.audio_stride;
  float res[NOUT];
#pragma unroll
  for (int m = 0; m < NOUT; ++m) res[m] = outl[64 * m + lane];
#pragma unroll
  for (int m = 0; m < NOUT; ++m) asm volatile("" : "+v"(res[m]));   // all reads first, then the stores
#pragma unroll
  for (int m = 0; m < NOUT; ++m) {
    const int j = jlo + 64 * m + lane;
    if (j >= jl && j < jh) __builtin_nontemporal_store(res[m], out + j);
  }
  SDRFM_STAMP();                                                // audio stage done
#ifdef SDRFM_DEV
  if (tsp && lane == 0) tsp[31] = __builtin_amdgcn_s_memrealtime();
#endif
#undef SDRFM_STAMP
  // ---- state hand-over by the wave that holds the end of the stream's chunk ------------------------------------------------
  if (p.fold_state && 63 * w + nuse == segs) {
    if (lane == nuse) p.yprev_out[stream] = make_float2(ylast.x, ylast.y);
    for (int k = lane; k < TA - 1; k += 64) p.hist_d_out[(size_t)stream * (TA - 1) + k] = dl[dpos((nuse + 1) * OPL - (TA - 1) + k)];
    for (int k = lane; k < T - 1; k += 64) {
      const int c = (int)p.N - (T - 1) + k;
      p.hist_x_out[(size_t)stream * (T - 1) + k] = load_x(p, stream, c);
      reinterpret_cast<unsigned short*>(p.hist_b_out)[(size_t)stream * (T - 1) + k] = (unsigned short)load_raw(p, stream, c);
    }
  }
}

// test hook: evaluates K3 on the device both ways the kernels do (scalar form and packed-pair form)
__global__ void k_debug_discriminate(const float* yr, const float* yi, const float* pr, const float* pi, float* out_scalar,
                                     float* out_pair, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_scalar[i] = sdrfm_discriminate(yr[i], yi[i], pr[i], pi[i]);
  // pair form: element .x is (y | p); element .y is fed the same operands through the second slot
  const f2_t d2 = discriminate_pair(f2_t{pr[i], pi[i]}, f2_t{0.f, 0.f}, f2_t{yr[i], yi[i]});
  out_pair[i] = d2.y;   // second slot computes disc(y1 = (yr,yi) | y0 = (pr,pi))
}

// per-stream routing: a closed window's repair passes per stream -> host-mapped memory, the device words cleared for the window after the next
__global__ void __launch_bounds__(256) k_route_collect(uint32_t* dev, uint32_t* host, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { host[i] = dev[i]; dev[i] = 0u; }
}

struct FastVariant {
  char kind;                         // 'a' = float tile (design A), 'b' = raw-byte tile (design B), 's' = streaming lanes (design S)
  uint32_t T, D, R;                  // R: outputs per lane and sub-tile (A, B) / accumulator slots (S)
  uint32_t Ta, Da;                   // design B only: compile-time audio geometry (0 = any)
  void (*kernel[8])(CallParams);   // [0] product; [1..7] timing experiments (profile / ablations)
  uint32_t xbytes;                   // LDS bytes of the sample tile (A, B) / of the whole ring (S)
  uint32_t seg;                      // design S: samples per lane segment (0 otherwise)
};
#ifdef SDRFM_DEV
#define SDRFM_FAST(T_, D_, R_) { 'a', T_, D_, R_, 0, 0, {k_fast<T_, D_, R_, 0>, k_fast<T_, D_, R_, 1>, k_fast<T_, D_, R_, 2>, k_fast<T_, D_, R_, 3>, k_fast<T_, D_, R_, 4>, k_fast<T_, D_, R_, 5>, k_fast<T_, D_, R_, 6>, k_fast<T_, D_, R_, 7>}, (uint32_t)fast_xbytes(T_, D_, R_), 0 }
#define SDRFM_FAST_LITE(T_, D_, R_) { 'a', T_, D_, R_, 0, 0, {k_fast<T_, D_, R_, 0>, k_fast<T_, D_, R_, 1>, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)fast_xbytes(T_, D_, R_), 0 }
#define SDRFM_FASTB2(T_, D_, R_, TA_, DA_) { 'b', T_, D_, R_, TA_, DA_, {k_fastb<T_, D_, R_, TA_, DA_, 0>, k_fastb<T_, D_, R_, TA_, DA_, 1>, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)fastb_xbytes(T_, D_, R_), 0 }
// headline shape only, timing ablations with wrong results (SDRFM_ABLATE=2..5): [2] halo samples not converted, [3] no sample
// converted, [4] no discriminator, [5] no conversion, no FIR, no discriminator (staging, LDS window reads and audio stage remain)
#define SDRFM_FASTB2_ABL(T_, D_, R_, TA_, DA_) { 'b', T_, D_, R_, TA_, DA_, {k_fastb<T_, D_, R_, TA_, DA_, 0>, k_fastb<T_, D_, R_, TA_, DA_, 1>, k_fastb<T_, D_, R_, TA_, DA_, 2>, k_fastb<T_, D_, R_, TA_, DA_, 3>, k_fastb<T_, D_, R_, TA_, DA_, 4>, k_fastb<T_, D_, R_, TA_, DA_, 5>, nullptr, nullptr}, (uint32_t)fastb_xbytes(T_, D_, R_), 0 }
#else   // product library: the result-correct kernel of every shape and nothing else
#define SDRFM_FASTB2(T_, D_, R_, TA_, DA_) SDRFM_FASTB2_LITE(T_, D_, R_, TA_, DA_)
#define SDRFM_FASTB2_ABL(T_, D_, R_, TA_, DA_) SDRFM_FASTB2_LITE(T_, D_, R_, TA_, DA_)
#endif
#define SDRFM_FASTB2_LITE(T_, D_, R_, TA_, DA_) { 'b', T_, D_, R_, TA_, DA_, {k_fastb<T_, D_, R_, TA_, DA_, 0>, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, (uint32_t)fastb_xbytes(T_, D_, R_), 0 }
#define SDRFM_FASTB(T_, D_, R_) SDRFM_FASTB2(T_, D_, R_, 32, 5)
#define SDRFM_STREAM(T_, D_, S_, NB_, TA_, DA_) { 's', T_, D_, S_, TA_, DA_, {k_stream<T_, D_, S_, NB_, TA_, DA_>, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, 2u * 64u * 128u, (uint32_t)(NB_) * (S_) * (D_) }
const FastVariant kFastVariants[] = {
    // design S (streaming lanes): the BASELINE configs[2]/[3] shape; serves calls that are whole numbers of lane segments
    // (a 16-tap instance is correct too but no faster than design B on cold inputs: 35.2 vs 34.1 us; it is not instantiated)
    SDRFM_STREAM(64, 10, 8, 6, 32, 5), SDRFM_STREAM(32, 10, 8, 6, 32, 5),
    // 2.4 MS/s -> 240 kS/s -> 48 kHz: the rate the firmware programs (usbh_rtlsdr.c:898) and the BASELINE configs
    SDRFM_FASTB2_ABL(64, 10, 12, 32, 5), SDRFM_FASTB2_LITE(64, 10, 8, 32, 5), SDRFM_FASTB(16, 10, 12), SDRFM_FASTB2_LITE(32, 10, 12, 32, 5),
    // R = 4: the noisy streams' workgroups beside design Q's (per-stream routing; inside design Q's launch: k_mix): 6.8 KB of LDS per wave — a slot one of design Q's waves (10.9 KB) leaves
    // takes one, which the 19 KB of the R = 12 instance cannot count on while design Q's waves keep coming
    SDRFM_FASTB2_LITE(64, 10, 4, 32, 5), SDRFM_FASTB2_LITE(32, 10, 4, 32, 5), SDRFM_FASTB2_LITE(16, 10, 4, 32, 5),
    // the other rates RTLSDR_set_sample_rate accepts and a dongle is commonly run at:
    // 2.048 MS/s -> 256 kS/s -> 32 kHz, 1.024 MS/s -> 256 kS/s -> 32 kHz, 3.2 MS/s -> 200 kS/s -> 40 kHz
    SDRFM_FASTB2_LITE(64, 8, 12, 32, 8), SDRFM_FASTB2_LITE(16, 8, 12, 32, 8), SDRFM_FASTB2_LITE(64, 4, 12, 32, 8), SDRFM_FASTB2_LITE(64, 16, 8, 32, 5),
    SDRFM_FASTB2_LITE(64, 8, 4, 32, 8), SDRFM_FASTB2_LITE(16, 8, 4, 32, 8), SDRFM_FASTB2_LITE(64, 16, 4, 32, 5),
#ifdef SDRFM_DEV
    // design A (float tile): kept as the measured alternative; ablation modes only on the documented shape
    SDRFM_FAST(64, 10, 3), SDRFM_FAST_LITE(16, 10, 2), SDRFM_FAST_LITE(32, 10, 2),
#endif
};

}  // namespace

// =================================================================================================================
//  Host side (C-ABI)
// =================================================================================================================
#define SDRFM_Q_ADAPT_WINDOW 16u     /* design-Q calls per window of per-stream repair statistics */
#define SDRFM_Q_ADAPT_SETS 4u        /* counter sets taken in turn: a window's statistics take effect when its set comes round again — at the first design-Q call of the
                                        window SETS windows later, a FIXED call: the host waits there if the device has not delivered them (a host > (SETS - 1) windows ahead) */
#define SDRFM_Q_ADAPT_BACKOFF 1024u  /* eligible calls served by the bit-exact kernels after a window of noise-like input */
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
  uint8_t* d_hist_b[2];   // raw-byte twin of hist_x (design B reads its halo from it)
  uint64_t n_seen;        // IQ samples consumed since reset (history is all-real once >= T-1)
  int cur;  // index of the state set holding the current state
  uint32_t phase_x, phase_d;
  // staging for host-pointer calls
  uint8_t* d_iq;
  size_t d_iq_stride;
  float* d_audio;
  size_t d_audio_stride;
  // small single-stream calls (URB-sized hand-offs): host-mapped pinned buffers the kernel reads / writes directly
  uint8_t* zc_iq; float* zc_audio; uint8_t* zc_iq_dev; float* zc_audio_dev; uint32_t zc_audio_cap; bool zc_off;
  uint32_t max_bytes;
  // generic kernel geometry
  uint32_t NA;
  size_t lds_bytes;
  // fast kernel (when one is instantiated for this T/D)
  const FastVariant* fast;
  const FastVariant* fast_s;  // design S variant of this geometry, if one is instantiated (serves the calls it is eligible for)
  const FastVariant* fast_mix; size_t fast_mix_lds;   // design B with the smallest tile (R = 4): the noisy streams' workgroups beside design Q's
  bool mix_split_off; double mix_rho;                            // (development library: SDRFM_MIX_SPLIT_OFF keeps the share-only split of the wave slots; SDRFM_MIX_RHO: the weight)
  uint32_t mix_lds, mix_waves_per_cu, mix_R; double mix_cost;                 // ... or INSIDE design Q's launch (sdrfm_q.hip: k_mix) where an instance exists: LDS bytes of a workgroup (0 = none), workgroups a CU holds
  uint32_t n_cu;              // compute units of the device
  char fast_s_name[64];
  size_t fast_lds;
  uint32_t waves_target;   // resident waves the fast kernel aims for (CUs x waves that fit by LDS)
  uint32_t min_subtiles;   // minimum sub-tiles per segment (bounds the per-segment halo recompute)
  uint32_t AB;             // sub-tiles of d buffered per audio flush
  unsigned long long* d_dbg;  // phase profile accumulators (only with SDRFM_PHASE_PROFILE=1)
  uint32_t warm_ahead;        // L2 warm-up distance (sub-tiles)
  uint32_t dbg_launches;      // launches since the debug counters were last reset
  int fast_mode;              // 0 product; 1..7 timing experiments (libsdrfm_dev.so only)
  uint32_t prio_balance, fold_state_ok;   // design B knobs, fixed at create
  uint32_t end_prio;                      // design S: see CallParams
  uint32_t stream_profile;                // development build: d_dbg holds per-wave time stamps of design S
  char kernel_name[112];
  char generic_name[64];
  char fast_name[64];
  // design Q (matrix-pipe FIR, sdrfm_q.hip): operand tables on the device, scale / offset, first K-chunk that holds taps
  int8_t* d_qA;
  float q_scale, q_cst;
  uint32_t q_c0, q_nslot, q_waves_per_cu;
  char fast_q_name[64];
  // design Q's conditioning guard: thresholds, the zero-padded taps of the repair path's chain, the last 64 raw samples of
  // every stream (kept like the other state sets), statistics; and what the current state set holds: y[-1] as the definition has it
  // (reset, or a bit-exact kernel served the previous call) or design Q's own value; hist_q written by design Q or not
  float q_guard_r, q_guard_a;
  float* d_hpad;
  uint8_t* d_hist_q[2];
  unsigned int* d_qstat;
  bool yprev_exact, hist_q_valid;
  // Which kernel serves a stream is also a matter of what the stream holds: noise-only input sends design Q to its repair path at
  // almost every audio stage (3 x the time of a carrier's call; the bit-exact kernels: 1.4 x), and a kernel lasts as long as its slowest
  // wave — so the choice is made PER STREAM (round 5; rounds 3 - 4: per handle).  Design Q's waves add their repair passes into a word per
  // stream (device memory); a window of SDRFM_Q_ADAPT_WINDOW design-Q calls is read back on a side stream behind the completion events of
  // the window's last kernels (hipExtLaunchKernelGGL stop events: no marker packets in the compute queues).  A stream more than a quarter of whose audio stages needed a repair
  // pass is served by the bit-exact kernels for SDRFM_Q_ADAPT_BACKOFF calls (design-B workgroups over the list of such streams INSIDE design
  // Q's launch over the others — sdrfm_q.hip: k_mix — or, where design B has no instance, a launch ahead of it), then tried on design Q again.
  // DETERMINISTIC since round 6: a window's statistics take effect at a FIXED call — the first design-Q call of the window
  // SDRFM_Q_ADAPT_SETS windows later, when the window's counter set comes round again — whenever the read-back arrived; should it not have arrived by then
  // (a host more than SETS - 1 windows of calls ahead of its device) that call waits for it.  Which kernel serves a stream at which call is therefore a
  // function of the bytes and the call sequence alone: two runs of one capture give the same bits (tests/test_route_gpu.py).  Every choice is within the
  // tolerance, a stream's audio is bit-identical to what its kernel gives alone, and SDRFM_CFG_BIT_EXACT pins the kernels.
  uint32_t* rt_pass_dev[SDRFM_Q_ADAPT_SETS]; uint32_t* rt_pass_host[SDRFM_Q_ADAPT_SETS]; uint32_t* rt_pass_host_dev[SDRFM_Q_ADAPT_SETS];   // repair passes per stream: the set the open window adds into / pinned read-back
  hipStream_t rt_mon;                                           // side stream of the read-backs
  hipEvent_t rt_rb_done[SDRFM_Q_ADAPT_SETS]; bool rt_rb_pending[SDRFM_Q_ADAPT_SETS]; uint64_t rt_rb_stages[SDRFM_Q_ADAPT_SETS];   // read-back of set i: its event; audio stages per stream its window covered
  hipEvent_t rt_win_evt[SDRFM_Q_ADAPT_SETS][3];                 // completion events of a window's last kernels: [set][internal stream 0, 1, the handle's stream]
  bool rt_win_need[3], rt_win_used[3];                          // open window: stream holds kernels no event covers / holds kernels at all
  uint32_t rt_win_calls, rt_set; uint64_t rt_win_stages;
  uint8_t* rt_noisy; uint64_t* rt_retry_at; uint32_t rt_n_noisy; uint64_t rt_calls, rt_next_retry;   // host: per stream, served by the bit-exact kernels until call rt_retry_at
  uint32_t* rt_list_dev[2]; uint32_t* rt_list_host[2]; int rt_list_cur; bool rt_dirty;   // stream lists: the clean streams first, then the noisy ones
  hipEvent_t rt_applied[2]; bool rt_applied_pending[2];         // list version v is in place (recorded on the handle's stream behind its copy: its host buffer may be rewritten after it)
  hipEvent_t rt_bx_evt[2]; uint32_t rt_bx_slot; hipStream_t rt_bx_last;   // the bit-exact sub-launches: completion events (stop events), and the stream the latest went to
                                                                // (each takes the state the one before left: on another stream it waits for that one's event)
  bool rt_off;                                                  // (development: design Q whatever the streams hold)
  // SDRFM_F_OVERLAP: two internal streams taken in turn, so that consecutive calls run concurrently on the device (a call's ramp-up
  // under the previous call's tail).  ovl_in orders a call behind what the handle's stream holds when it is made; join_overlap()
  // records ovl_done[k] behind the calls put on internal stream k and makes the handle's stream wait for it.
  hipStream_t ovl_stream[2];
  hipEvent_t ovl_in, ovl_done[2];
  // sdrfm_process_batch_pcm: the sink whose chain the call's design-Q launch ends with (sdrfm_sink_chain.h); pcm_fused: that launch took it
  sdrfm_pcm_sink* pcm_sink; int16_t* pcm_out; size_t pcm_out_stride; bool pcm_fused;
  bool pcm_no_audio; float* d_pcm_audio; size_t pcm_audio_stride; uint32_t pcm_audio_cap, pcm_audio_flip;   // a call without an audio buffer: the library's own, for the calls that need one
  const int16_t* prev_ovl_pcm; size_t prev_ovl_pcm_stride;       // the PCM rows the previous overlapped call may still be writing
  int16_t* d_pcm_stage;                                          // sdrfm_process_batch_pcm with host buffers: the PCM rows before they are copied back
  unsigned long long* d_runstate; uint32_t runstate_cap;        // the runs' hand-off words (sdrfm_sink_chain.h), allocated at the first such call
  bool ovl_pending[2], ovl_bound[2], ovl_join_style;   // (bound: the latest kernel of stream k carries ovl_done[k] as its stop event; join_style: the caller joins after every call)
  uint32_t ovl_next;
  // the previous call's device buffer (valid after a SDRFM_F_DEVICE_PTRS call): what an overlapped call warms its streams up from
  const uint8_t* prev_iq; size_t prev_stride; uint32_t prev_nbytes;
  // the audio buffer the most recent overlapped call writes (it may still be running when the next call is made)
  const float* prev_ovl_audio; size_t prev_ovl_audio_stride; uint32_t prev_ovl_audio_n;
};

static int enqueue(sdrfm* h, const uint8_t* d_iq, size_t iq_stride, uint32_t nbytes, float* d_audio, size_t audio_stride,
                   uint32_t* n_audio, uint32_t call_flags = 0);
static int join_overlap(sdrfm* h);
static int route_create(sdrfm* h);
static int route_reset(sdrfm* h);
static int route_apply(sdrfm* h);

#define HIP_TRY(expr, code)                                                                          \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess) {                                                                         \
      fprintf(stderr, "[sdrfm] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return (code);                                                                                 \
    }                                                                                                \
  } while (0)

// ---- per-stream routing between design Q and the bit-exact kernels (the handle's comment has the scheme) ------------------------------
static int route_create(sdrfm* h) {
  const size_t ns = h->cfg.n_streams;
  h->rt_noisy = static_cast<uint8_t*>(calloc(ns, 1));
  h->rt_retry_at = static_cast<uint64_t*>(calloc(ns, sizeof(uint64_t)));
  if (!h->rt_noisy || !h->rt_retry_at) return SDRFM_ENOMEM;
  for (uint32_t i = 0; i < SDRFM_Q_ADAPT_SETS; ++i) {
    HIP_TRY(hipMalloc(&h->rt_pass_dev[i], ns * sizeof(uint32_t)), SDRFM_ENOMEM);
    HIP_TRY(hipMemset(h->rt_pass_dev[i], 0, ns * sizeof(uint32_t)), SDRFM_FAIL);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->rt_pass_host[i]), ns * sizeof(uint32_t), hipHostMallocMapped), SDRFM_ENOMEM);
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->rt_pass_host_dev[i]), h->rt_pass_host[i], 0), SDRFM_FAIL);
    HIP_TRY(hipEventCreateWithFlags(&h->rt_rb_done[i], hipEventDisableTiming), SDRFM_ENOMEM);
    for (int k = 0; k < 3; ++k) HIP_TRY(hipEventCreateWithFlags(&h->rt_win_evt[i][k], hipEventDisableTiming), SDRFM_ENOMEM);
  }
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipMalloc(&h->rt_list_dev[i], ns * sizeof(uint32_t)), SDRFM_ENOMEM);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->rt_list_host[i]), ns * sizeof(uint32_t), hipHostMallocDefault), SDRFM_ENOMEM);
    HIP_TRY(hipEventCreateWithFlags(&h->rt_applied[i], hipEventDisableTiming), SDRFM_ENOMEM);
    HIP_TRY(hipEventCreateWithFlags(&h->rt_bx_evt[i], hipEventDisableTiming), SDRFM_ENOMEM);
  }
  // (No third stream for the noisy streams' launches: tried two ways, lost both times — profiles/r05_mixed_batches.txt.  An ordinary stream shares the handle's
  // stream's hardware queue (streams of one priority share a small pool): its kernels sat between that stream's event markers and every overlapped call waited
  // for the previous call's launch; a CU-mask stream (a queue of its own) ran the overlapped calls at 53 us against 36 - 41 us with the launch ahead of design Q's
  // on the call's own stream.  hipExtAnyOrderLaunch — the launch behind design Q's without a barrier bit — is ignored on gfx950.  Where both designs have an
  // instance the two kinds of workgroup share ONE launch: sdrfm_q.hip, k_mix.)
  HIP_TRY(hipStreamCreateWithFlags(&h->rt_mon, hipStreamNonBlocking), SDRFM_ENOMEM);
  h->rt_next_retry = ~0ull;
  return SDRFM_OK;
}

// Everything the device may still hold for this handle has been synchronised by the caller (sdrfm_reset): every stream back on design Q.
static int route_reset(sdrfm* h) {
  if (!h->rt_noisy) return SDRFM_OK;
  const size_t ns = h->cfg.n_streams;
  if (h->rt_mon) HIP_TRY(hipStreamSynchronize(h->rt_mon), SDRFM_FAIL);
  for (int k = 0; k < 2; ++k)
    if (h->ovl_stream[k]) HIP_TRY(hipStreamSynchronize(h->ovl_stream[k]), SDRFM_FAIL);
  for (uint32_t i = 0; i < SDRFM_Q_ADAPT_SETS; ++i) {
    HIP_TRY(hipMemset(h->rt_pass_dev[i], 0, ns * sizeof(uint32_t)), SDRFM_FAIL);
    h->rt_rb_pending[i] = false;
  }
  memset(h->rt_noisy, 0, ns);
  h->rt_n_noisy = 0; h->rt_next_retry = ~0ull; h->rt_dirty = false; h->rt_applied_pending[0] = h->rt_applied_pending[1] = false; h->rt_bx_last = nullptr; h->rt_calls = 0;
  h->rt_win_calls = 0; h->rt_win_stages = 0; h->rt_set = 0;
  for (int k = 0; k < 3; ++k) { h->rt_win_need[k] = false; h->rt_win_used[k] = false; }
  return SDRFM_OK;
}

// The read-back of counter set i is taken in — at a FIXED call: enqueue() calls this at the first design-Q call of the window that will count into set i
// again, SDRFM_Q_ADAPT_SETS windows after the one the read-back covers.  A stream more than a quarter of whose audio stages needed a repair pass leaves
// design Q for a while.  The read-back was enqueued (SETS - 1) windows of calls ago; a host that far ahead of its device waits here (the one place a
// SDRFM_F_DEVICE_PTRS call can wait: include/sdrfm.h) — the price of a kernel assignment that is a function of the bytes and the call sequence alone.
static int route_take(sdrfm* h, uint32_t i) {
  if (!h->rt_rb_pending[i]) return SDRFM_OK;
  HIP_TRY(hipEventSynchronize(h->rt_rb_done[i]), SDRFM_FAIL);
  h->rt_rb_pending[i] = false;
  const uint32_t* pass = h->rt_pass_host[i];
  for (uint32_t s = 0; s < h->cfg.n_streams; ++s)
    if (!h->rt_noisy[s] && (uint64_t)pass[s] * 4u > h->rt_rb_stages[i]) {
      h->rt_noisy[s] = 1;
      h->rt_retry_at[s] = h->rt_calls + SDRFM_Q_ADAPT_BACKOFF;
      if (h->rt_retry_at[s] < h->rt_next_retry) h->rt_next_retry = h->rt_retry_at[s];
      h->rt_dirty = true;
    }
  return SDRFM_OK;
}

// streams whose time on the bit-exact kernels is over (counted in calls): design Q is tried again
static void route_retry(sdrfm* h) {
  if (h->rt_calls < h->rt_next_retry) return;
  h->rt_next_retry = ~0ull;
  for (uint32_t s = 0; s < h->cfg.n_streams; ++s) {
    if (!h->rt_noisy[s]) continue;
    if (h->rt_retry_at[s] <= h->rt_calls) { h->rt_noisy[s] = 0; h->rt_dirty = true; }
    else if (h->rt_retry_at[s] < h->rt_next_retry) h->rt_next_retry = h->rt_retry_at[s];
  }
}

static uint32_t max_audio_for(const sdrfm_config& c, uint32_t nbytes) {
  const uint64_t n = nbytes / 2;
  const uint64_t m = (n + c.fir_decim - 1) / c.fir_decim + 1;
  return (uint32_t)((m + c.audio_decim - 1) / c.audio_decim + 1);
}

// Everything design Q and its per-stream routing own (free_handle; and sdrfm_create when any of it could not be had: the handle then runs the bit-exact
// kernels, with nothing half-allocated left behind — ADVICE r05).  Every member is null or valid; all are null afterwards.
static void q_free(sdrfm* h) {
  if (h->d_runstate) (void)hipFree(h->d_runstate);
  if (h->d_pcm_audio) (void)hipFree(h->d_pcm_audio);
  if (h->d_pcm_stage) (void)hipFree(h->d_pcm_stage);
  h->d_runstate = nullptr; h->d_pcm_audio = nullptr; h->d_pcm_stage = nullptr;
  if (h->d_qA) (void)hipFree(h->d_qA);
  if (h->d_hpad) (void)hipFree(h->d_hpad);
  h->d_qA = nullptr; h->d_hpad = nullptr;
  for (uint32_t i = 0; i < SDRFM_Q_ADAPT_SETS; ++i) {
    if (h->rt_pass_dev[i]) (void)hipFree(h->rt_pass_dev[i]);
    if (h->rt_pass_host[i]) (void)hipHostFree(h->rt_pass_host[i]);
    if (h->rt_rb_done[i]) (void)hipEventDestroy(h->rt_rb_done[i]);
    h->rt_pass_dev[i] = nullptr; h->rt_pass_host[i] = nullptr; h->rt_pass_host_dev[i] = nullptr; h->rt_rb_done[i] = nullptr;
    for (int k = 0; k < 3; ++k) {
      if (h->rt_win_evt[i][k]) (void)hipEventDestroy(h->rt_win_evt[i][k]);
      h->rt_win_evt[i][k] = nullptr;
    }
  }
  for (int i = 0; i < 2; ++i) {
    if (h->rt_list_dev[i]) (void)hipFree(h->rt_list_dev[i]);
    if (h->rt_list_host[i]) (void)hipHostFree(h->rt_list_host[i]);
    if (h->rt_applied[i]) (void)hipEventDestroy(h->rt_applied[i]);
    if (h->rt_bx_evt[i]) (void)hipEventDestroy(h->rt_bx_evt[i]);
    h->rt_list_dev[i] = nullptr; h->rt_list_host[i] = nullptr; h->rt_applied[i] = nullptr; h->rt_bx_evt[i] = nullptr;
  }
  if (h->rt_mon) (void)hipStreamDestroy(h->rt_mon);
  h->rt_mon = nullptr;
  free(h->rt_noisy);
  free(h->rt_retry_at);
  h->rt_noisy = nullptr; h->rt_retry_at = nullptr;
  if (h->d_qstat) (void)hipFree(h->d_qstat);
  h->d_qstat = nullptr;
  for (int i = 0; i < 2; ++i) {
    if (h->d_hist_q[i]) (void)hipFree(h->d_hist_q[i]);
    h->d_hist_q[i] = nullptr;
  }
}

static void free_handle(sdrfm* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (int k = 0; k < 2; ++k)
    if (h->ovl_stream[k]) (void)hipStreamSynchronize(h->ovl_stream[k]);     // overlapped calls still use the buffers freed below
  if (h->rt_mon) (void)hipStreamSynchronize(h->rt_mon);
  if (h->d_h) (void)hipFree(h->d_h);
  if (h->d_g) (void)hipFree(h->d_g);
  for (int i = 0; i < 2; ++i) {
    if (h->d_hist_x[i]) (void)hipFree(h->d_hist_x[i]);
    if (h->d_yprev[i]) (void)hipFree(h->d_yprev[i]);
    if (h->d_hist_d[i]) (void)hipFree(h->d_hist_d[i]);
    if (h->d_hist_b[i]) (void)hipFree(h->d_hist_b[i]);
  }
  if (h->d_iq) (void)hipFree(h->d_iq);
  if (h->zc_iq) (void)hipHostFree(h->zc_iq);
  if (h->zc_audio) (void)hipHostFree(h->zc_audio);
  if (h->d_audio) (void)hipFree(h->d_audio);
  if (h->d_dbg) (void)hipFree(h->d_dbg);
  q_free(h);
  for (int k = 0; k < 2; ++k) {
    if (h->ovl_stream[k]) (void)hipStreamDestroy(h->ovl_stream[k]);
    if (h->ovl_done[k]) (void)hipEventDestroy(h->ovl_done[k]);
  }
  if (h->ovl_in) (void)hipEventDestroy(h->ovl_in);
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

// URB-sized synchronous calls spend most of their time in two tiny DMA transfers.  Up to SDRFM_ZC_MAX bytes the bytes are
// instead copied by the CPU into a pinned, device-mapped buffer that the kernel reads over PCIe, and the kernel writes its
// audio straight into mapped host memory: one launch and one stream wait per call.
#define SDRFM_ZC_MAX 65536u
static int ensure_zero_copy(sdrfm* h) {
  if (h->zc_iq) return SDRFM_OK;
  h->zc_audio_cap = (uint32_t)((max_audio_for(h->cfg, SDRFM_ZC_MAX) + 63) & ~(size_t)63);
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->zc_iq), SDRFM_ZC_MAX + 256, hipHostMallocMapped), SDRFM_ENOMEM);
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->zc_audio), sizeof(float) * h->zc_audio_cap, hipHostMallocMapped), SDRFM_ENOMEM);
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zc_iq_dev), h->zc_iq, 0), SDRFM_FAIL);
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->zc_audio_dev), h->zc_audio, 0), SDRFM_FAIL);
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
  if (!cfg->n_streams || !cfg->fir_coeffs || !cfg->audio_coeffs || (cfg->flags & ~(SDRFM_CFG_FORCE_GENERIC | SDRFM_CFG_NO_ZEROCOPY | SDRFM_CFG_BIT_EXACT | SDRFM_CFG_GUARD_WORST_CASE))) return SDRFM_EINVAL;
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
  h->zc_off = (cfg->flags & SDRFM_CFG_NO_ZEROCOPY) != 0;
  h->prio_balance = 1; h->fold_state_ok = 1; h->end_prio = (1u | (1u << 2)) << 6;
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
    CR(hipMalloc(&h->d_hist_b[i], 2 * ns * hx));
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
  snprintf(h->generic_name, sizeof(h->generic_name), "generic T%u D%u Ta%u Da%u NA%u", cfg->fir_taps, cfg->fir_decim,
           cfg->audio_taps, cfg->audio_decim, NA);
  snprintf(h->kernel_name, sizeof(h->kernel_name), "%s", h->generic_name);
  if (!(cfg->flags & SDRFM_CFG_FORCE_GENERIC)) {
    char want_kind = 'b';
    uint32_t want_r = 12, ab_env = 0;
#ifdef SDRFM_DEV   // every environment knob is a development aid: read once, here, and only in libsdrfm_dev.so
    if (const char* e = getenv("SDRFM_FAST_KIND")) want_kind = e[0];
    want_r = (want_kind == 'b') ? 12 : 3;
    if (const char* e = getenv("SDRFM_FAST_R")) want_r = (uint32_t)atoi(e);
    if (const char* e = getenv("SDRFM_AUDIO_BATCH")) ab_env = (uint32_t)atoi(e);
    if (getenv("SDRFM_NO_PRIO")) h->prio_balance = 0;
    if (const char* e = getenv("SDRFM_END_PRIO")) h->end_prio = (uint32_t)strtoul(e, nullptr, 0);
    if (getenv("SDRFM_NO_FOLD")) h->fold_state_ok = 0;
#endif
    for (const FastVariant& v : kFastVariants) {
      if (v.kind != 's' || v.T != cfg->fir_taps || v.D != cfg->fir_decim || v.Ta != cfg->audio_taps || v.Da != cfg->audio_decim) continue;
#ifdef SDRFM_DEV
      if (getenv("SDRFM_NO_STREAM")) continue;
#endif
#ifdef SDRFM_DEV
      if (getenv("SDRFM_STREAM_PROFILE") && !h->d_dbg) {        // 16 words per wave, up to 16384 waves
        if (hipMalloc(&h->d_dbg, 32 * 16384 * sizeof(unsigned long long)) != hipSuccess) h->d_dbg = nullptr;
        else { (void)hipMemset(h->d_dbg, 0, 32 * 16384 * sizeof(unsigned long long)); h->stream_profile = 1; }
      }
#endif
      h->fast_s = &v;
      h->n_cu = (uint32_t)prop.multiProcessorCount;
      snprintf(h->fast_s_name, sizeof(h->fast_s_name), "fast-s T%u D%u S%u L%u Ta%u Da%u", v.T, v.D, v.R, v.seg, v.Ta, v.Da);
    }
    // design Q: K2 on the i8 matrix pipe (sdrfm_q.hip).  Not bit-identical to the fmaf-chain kernels (within 1e-6 of the
    // oracle where the phase is well conditioned, repaired to the definition's own d where it is not: the guard below), so a handle
    // created with SDRFM_CFG_BIT_EXACT never selects it.
    // PERFORMANCE heuristic, not a correctness condition: it is offered LOW-PASS channel filters only, sum|h| <= 2 |sum h| (a windowed sinc
    // has 1.2 - 1.5) — with heavy cancellation (no pass band around DC) |y| is small against the chain's partial sums for every input,
    // the guard sends most outputs to the repair path and the bit-exact kernels are the faster way to the same numbers.
    double q_abs = 0.0, q_sum = 0.0;
    for (uint32_t k = 0; k < cfg->fir_taps; ++k) { q_abs += std::fabs((double)hc[k]); q_sum += (double)hc[k]; }
    // The conditioning guard's thresholds (qtaps.c: sdrfm_q_guard).  A guard that would send a carrier at an eighth of full scale to the
    // repair path makes design Q pointless for these taps: the bit-exact kernels serve them.
    float q_R = 0.0f, q_A = 4.0f;
    // (SDRFM_CFG_GUARD_WORST_CASE: the radius from the proven worst-case bound — 6.9 x at 64 taps; a carrier at a third of full scale must still clear it)
    const bool q_wc = (cfg->flags & SDRFM_CFG_GUARD_WORST_CASE) != 0;
    const bool q_guard_ok = sdrfm_q_guard2(hc, cfg->fir_taps, gc, cfg->audio_taps, q_wc ? 1 : 0, &q_R, &q_A) == 0 &&
                            (double)q_R <= (q_wc ? 0.33 : 0.125) * 127.5 * std::fabs(q_sum) && q_A > 3.0f;
    // instances: (D, Da) = (10, 5) — the 2.4 MS/s front end of BASELINE —, (8, 8) and (16, 5): the 2.048 and 3.2 MS/s rates
    // RTLSDR_set_sample_rate accepts (usbh_rtlsdr.c:676-678); 32 audio taps each
    const size_t q_tab_bytes = (size_t)SDRFM_Q_SPARSE_CHUNKS(cfg->fir_decim) * SDRFM_Q_DIGITS * 64 * 16;
    if (!(cfg->flags & SDRFM_CFG_BIT_EXACT) && sdrfm_q_geometry_ok(cfg->fir_decim, cfg->audio_decim) && cfg->audio_taps == SDRFM_Q_TA &&
        cfg->fir_taps <= SDRFM_Q_TP && cfg->fir_taps <= 9 * cfg->fir_decim && q_abs <= 2.0 * std::fabs(q_sum) && q_guard_ok) {
      int8_t* tab = (int8_t*)malloc(q_tab_bytes);
      float qs = 0.f, qc = 0.f, hpad[SDRFM_Q_TP];
      uint32_t c0 = 0;
      for (uint32_t k = 0; k < SDRFM_Q_TP; ++k) hpad[k] = k < cfg->fir_taps ? hc[k] : 0.0f;
      const bool q_built = tab && sdrfm_q_build(hc, cfg->fir_taps, cfg->fir_decim, tab, &qs, &qc, &c0) == 0;   // (false: taps the tables cannot hold)
      if (q_built &&
          hipMalloc(&h->d_qA, q_tab_bytes) == hipSuccess &&
          hipMemcpy(h->d_qA, tab, q_tab_bytes, hipMemcpyHostToDevice) == hipSuccess &&
          hipMalloc(&h->d_hpad, sizeof(hpad)) == hipSuccess && hipMemcpy(h->d_hpad, hpad, sizeof(hpad), hipMemcpyHostToDevice) == hipSuccess &&
          hipMalloc(&h->d_hist_q[0], 2 * SDRFM_Q_TP * ns) == hipSuccess && hipMalloc(&h->d_hist_q[1], 2 * SDRFM_Q_TP * ns) == hipSuccess &&
          hipMalloc(&h->d_qstat, 2 * sizeof(unsigned int)) == hipSuccess && hipMemset(h->d_qstat, 0, 2 * sizeof(unsigned int)) == hipSuccess &&
          route_create(h) == SDRFM_OK) {
        h->q_scale = qs; h->q_cst = qc; h->q_c0 = c0 > 1 ? 1 : c0;
        h->q_guard_r = q_R; h->q_guard_a = q_A;
        h->q_nslot = sdrfm_q_default_nslot(cfg->fir_decim); h->q_waves_per_cu = 12;
        { const uint32_t lb = sdrfm_q_lds_bytes(h->q_nslot, cfg->fir_decim, cfg->audio_decim); if (lb && 163840u / lb < h->q_waves_per_cu) h->q_waves_per_cu = 163840u / lb; }   // (D = 16: 11 one-wave workgroups fit a CU's LDS)
#ifdef SDRFM_DEV
        if (const char* e = getenv("SDRFM_Q_NSLOT")) h->q_nslot = (uint32_t)atoi(e);
        if (const char* e = getenv("SDRFM_Q_WAVES_PER_CU")) h->q_waves_per_cu = (uint32_t)atoi(e);
        if (const char* e = getenv("SDRFM_Q_GUARD_R")) h->q_guard_r = (float)atof(e);            // 0 and 4: the guard never fires (timing / soak experiments)
        if (const char* e = getenv("SDRFM_Q_GUARD_A")) h->q_guard_a = (float)atof(e);
        if (getenv("SDRFM_NO_Q")) { (void)hipFree(h->d_qA); h->d_qA = nullptr; }
        if (getenv("SDRFM_Q_NO_ADAPT")) h->rt_off = true;                                         // design Q whatever the streams hold (timing experiments)
#endif
        h->n_cu = (uint32_t)prop.multiProcessorCount;
        snprintf(h->fast_q_name, sizeof(h->fast_q_name), "fast-q T%u D%u Ta%u Da%u %s", cfg->fir_taps, cfg->fir_decim, cfg->audio_taps,
                 cfg->audio_decim, sdrfm_q_kernel_symbol(h->q_c0, h->q_nslot, cfg->fir_decim, cfg->audio_decim));
      } else {
        // (the handle serves every call with the bit-exact kernels; said once, so that the slower path is not silent)
        if (q_built) fprintf(stderr, "[sdrfm] the matrix-pipe kernel's tables or routing state could not be allocated: this handle runs the bit-exact kernels only\n");
        (void)hipGetLastError();
        q_free(h);
      }
      free(tab);
    }
    for (int pass = 0; pass < 3 && !h->fast; ++pass)
    for (const FastVariant& v : kFastVariants) {
      if (v.kind == 's') continue;
      if (v.T != cfg->fir_taps || v.D != cfg->fir_decim) continue;
      if (v.Ta && (v.Ta != cfg->audio_taps || v.Da != cfg->audio_decim)) continue;
      if (pass == 0 && (v.R != want_r || v.kind != want_kind)) continue;
      if (pass == 1 && v.kind != want_kind) continue;
      const uint32_t NYT = 64 * v.R, DOFF = (cfg->audio_taps - 1 + 3u) & ~3u;
      // audio flush every AB sub-tiles: AB = Da makes every flush exactly 64*R outputs (all lanes busy)
      uint32_t AB = v.kind == 'b' ? (uint32_t)fastb_ab((int)v.R) : (ab_env ? ab_env : cfg->audio_decim);   // (design B: a compile-time property of the tile)
      if (AB > 8) AB = 8;
      while (AB > 1 && (size_t)v.xbytes + (size_t)(DOFF + AB * NYT + v.T + cfg->audio_taps) * 4 > 40 * 1024) --AB;
      if (cfg->audio_taps - 1 > AB * NYT || DOFF + AB * NYT < 2 * (cfg->audio_taps + 1)) continue;
      const size_t lds = (size_t)v.xbytes + (size_t)(DOFF + AB * NYT + v.T + cfg->audio_taps) * 4;
      if (lds > 160 * 1024) continue;
      if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(v.kernel[0]),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        continue;
      h->AB = AB;
      h->warm_ahead = 0;
#ifdef SDRFM_DEV
      if (const char* e = getenv("SDRFM_WARM_AHEAD")) h->warm_ahead = (uint32_t)atoi(e);
      if (const char* e = getenv("SDRFM_ABLATE")) { const int m = atoi(e); if (m >= 2 && m <= 7 && v.kernel[m]) h->fast_mode = m; }
      if (getenv("SDRFM_PHASE_PROFILE") && !h->d_dbg && v.kernel[1] && !h->stream_profile) {
        if (hipMalloc(&h->d_dbg, 560 * sizeof(unsigned long long)) != hipSuccess) h->d_dbg = nullptr;
        else { (void)hipMemset(h->d_dbg, 0, 560 * sizeof(unsigned long long)); for (int x = 0; x < 8; ++x) { (void)hipMemset(h->d_dbg + 520 + 4 * x, 0xff, 8); (void)hipMemset(h->d_dbg + 522 + 4 * x, 0xff, 8); } h->fast_mode = 1; }
      }
#endif
      h->fast = &v;
      h->fast_lds = lds;
      uint32_t per_cu = (uint32_t)((160 * 1024) / lds);
      if (per_cu > 16) per_cu = 16;
#ifdef SDRFM_DEV
      if (const char* e = getenv("SDRFM_WAVES_PER_CU")) per_cu = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : per_cu;
#endif
      h->waves_target = (uint32_t)prop.multiProcessorCount * per_cu;
      h->min_subtiles = 4;
#ifdef SDRFM_DEV
      if (const char* e = getenv("SDRFM_MIN_SUBTILES")) h->min_subtiles = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : 4;
#endif
      snprintf(h->fast_name, sizeof(h->fast_name), "fast-%c T%u D%u R%u Ta%u Da%u AB%u", v.kind, v.T, v.D, v.R,
               cfg->audio_taps, cfg->audio_decim, AB);
      snprintf(h->kernel_name, sizeof(h->kernel_name), "%s", h->fast_name);
      break;
    }
  }
  if (h->fast && h->fast->kind == 'b' && h->d_qA)
    for (const FastVariant& v : kFastVariants) {
      if (v.kind != 'b' || v.R != 4 || v.T != cfg->fir_taps || v.D != cfg->fir_decim || (v.Ta && (v.Ta != cfg->audio_taps || v.Da != cfg->audio_decim))) continue;
      const uint32_t NYT = 64 * v.R, DOFF = (cfg->audio_taps - 1 + 3u) & ~3u;
      if (cfg->audio_taps - 1 > NYT || DOFF + NYT < 2 * (cfg->audio_taps + 1)) continue;
      h->fast_mix = &v;
      h->fast_mix_lds = (size_t)v.xbytes + (size_t)(DOFF + (uint32_t)fastb_ab((int)v.R) * NYT + v.T + cfg->audio_taps) * 4;
      // a stream costs the design-B workgroups about mix_cost times what it costs design Q's: the shares of the wave slots (measured: 2.0 / 2.7 / 3.2 ->
      // 41.2 / 38.9 / 40.3 us serial, 31.5 / 30.8 / 33.0 us overlapped with a quarter of the streams noisy: profiles/r05_mixed_batches.txt)
      h->mix_R = v.R; h->mix_cost = 2.7; h->mix_rho = 12.7;
#ifdef SDRFM_DEV
      if (const char* e = getenv("SDRFM_MIX_COST")) h->mix_cost = atof(e);
      if (getenv("SDRFM_MIX_SPLIT_OFF")) h->mix_split_off = true;
      if (const char* e = getenv("SDRFM_MIX_RHO")) h->mix_rho = atof(e);
#endif
      h->mix_lds = (cfg->audio_taps == SDRFM_Q_TA) ? sdrfm_q_mix_lds(h->q_c0, h->q_nslot, cfg->fir_decim, cfg->audio_decim, v.T, h->mix_R) : 0u;
      if (h->mix_lds) {
        const int nb = sdrfm_q_mix_blocks_per_cu(h->q_c0, h->q_nslot, cfg->fir_decim, cfg->audio_decim, v.T, h->mix_R);
        h->mix_waves_per_cu = nb > 0 ? (uint32_t)nb : 163840u / h->mix_lds;
        if (h->mix_waves_per_cu > h->q_waves_per_cu) h->mix_waves_per_cu = h->q_waves_per_cu;   // (15 fit; design Q's own 12 are faster: same file)
#ifdef SDRFM_DEV
        if (getenv("SDRFM_MIX_OFF")) h->mix_lds = 0;
        if (const char* e = getenv("SDRFM_MIX_WAVES_PER_CU")) h->mix_waves_per_cu = (uint32_t)atoi(e);
#endif
      }
    }
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
  { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
  h->prev_iq = nullptr;
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipMemsetAsync(h->d_hist_x[i], 0, sizeof(float2) * ns * hx, h->stream), SDRFM_FAIL);
    HIP_TRY(hipMemsetAsync(h->d_yprev[i], 0, sizeof(float2) * ns, h->stream), SDRFM_FAIL);
    HIP_TRY(hipMemsetAsync(h->d_hist_d[i], 0, sizeof(float) * ns * hd, h->stream), SDRFM_FAIL);
    HIP_TRY(hipMemsetAsync(h->d_hist_b[i], 0, 2 * ns * hx, h->stream), SDRFM_FAIL);
    if (h->d_hist_q[i]) HIP_TRY(hipMemsetAsync(h->d_hist_q[i], 0, 2 * SDRFM_Q_TP * ns, h->stream), SDRFM_FAIL);
  }
  h->yprev_exact = true;                                          // y[-1] = 0, as the definition has it
  h->hist_q_valid = false;
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  { const int rrc = route_reset(h); if (rrc != SDRFM_OK) return rrc; }   // (behind the synchronisation: no kernel is left that could add to the statistics)
  h->cur = 0;
  h->phase_x = h->phase_d = 0;
  h->n_seen = 0;
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

// Everything issued with SDRFM_F_OVERLAP so far is ordered before whatever the handle's stream is given next.
static int join_overlap(sdrfm* h) {
  for (int k = 0; k < 2; ++k)
    if (h->ovl_pending[k]) {
      if (!h->ovl_bound[k]) HIP_TRY(hipEventRecord(h->ovl_done[k], h->ovl_stream[k]), SDRFM_FAIL);   // (else: the stream's latest kernel carries the event — enqueue())
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ovl_done[k], 0), SDRFM_FAIL);
      h->ovl_pending[k] = false;
    }
  return SDRFM_OK;
}

// A new assignment of streams to kernels: the list [clean streams | noisy streams] goes to the other version's buffers, and the state
// every stream carries is made the same for both kinds of kernel first (y[-1] of the streams design Q served last becomes the
// definition's, the raw-sample history design Q's repair path reads is refreshed for all) — behind everything issued so far, ahead of
// everything issued from here on, by events alone (no host wait).
static int route_apply(sdrfm* h) {
  const sdrfm_config& c = h->cfg;
  const uint32_t ns = c.n_streams;
  { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
  const int v = h->rt_list_cur ^ 1;
  if (h->rt_applied_pending[v]) {                                // (this version's host buffer feeds a copy enqueued two assignments ago: over, but for a host that far ahead)
    HIP_TRY(hipEventSynchronize(h->rt_applied[v]), SDRFM_FAIL);
    h->rt_applied_pending[v] = false;
  }
  uint32_t* L = h->rt_list_host[v];
  uint32_t nc = 0, nn = 0;
  for (uint32_t s = 0; s < ns; ++s) if (!h->rt_noisy[s]) L[nc++] = s;
  for (uint32_t s = 0; s < ns; ++s) if (h->rt_noisy[s]) L[nc + nn++] = s;
  if (!h->yprev_exact && h->hist_q_valid)                        // (design Q served ns - rt_n_noisy streams at the previous call: the clean part of the list in use)
    HIP_TRY(sdrfm_q_fix_yprev(h->d_hist_q[h->cur], h->d_hpad, h->d_yprev[h->cur], ns - h->rt_n_noisy, h->rt_n_noisy ? h->rt_list_dev[h->rt_list_cur] : nullptr,
                              h->stream), SDRFM_FAIL);
  if (c.fir_taps > 1)
    HIP_TRY(hipMemcpy2DAsync(h->d_hist_q[h->cur] + 2 * (SDRFM_Q_TP - (c.fir_taps - 1)), 2 * SDRFM_Q_TP, h->d_hist_b[h->cur], 2 * (size_t)(c.fir_taps - 1),
                             2 * (size_t)(c.fir_taps - 1), ns, hipMemcpyDeviceToDevice, h->stream), SDRFM_FAIL);
  h->yprev_exact = true; h->hist_q_valid = true;
  HIP_TRY(hipMemcpyAsync(h->rt_list_dev[v], L, (size_t)ns * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream), SDRFM_FAIL);
  HIP_TRY(hipEventRecord(h->rt_applied[v], h->stream), SDRFM_FAIL);
  h->rt_applied_pending[v] = true;
  for (int k = 0; k < 2; ++k)
    if (h->ovl_stream[k]) HIP_TRY(hipStreamWaitEvent(h->ovl_stream[k], h->rt_applied[v], 0), SDRFM_FAIL);
  h->rt_list_cur = v; h->rt_n_noisy = nn; h->rt_dirty = false;
  return SDRFM_OK;
}

int sdrfm_flush(sdrfm_t* h) {
  if (!h) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  return join_overlap(h);
}

// All overlapped calls but the most recent one: what a consumer of call k - 1's audio on the handle's stream needs while call k runs.
int sdrfm_flush_previous(sdrfm_t* h) {
  if (!h) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  const uint32_t k = h->ovl_next;                                // the stream the NEXT call takes = the one the call before the last took
  h->ovl_join_style = true;                                     // (a join after every call: from now on the kernels carry their stream's completion event)
  if (h->ovl_pending[k]) {
    if (!h->ovl_bound[k]) HIP_TRY(hipEventRecord(h->ovl_done[k], h->ovl_stream[k]), SDRFM_FAIL);
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ovl_done[k], 0), SDRFM_FAIL);
    h->ovl_pending[k] = false;
  }
  return SDRFM_OK;
}

// A stream of the caller's behind all overlapped calls but the most recent one; the handle's own stream is not touched (it keeps its pending joins).
int sdrfm_wait_previous(sdrfm_t* h, void* hip_stream) {
  if (!h || !hip_stream) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  const uint32_t k = h->ovl_next;                                // the stream the NEXT call takes = the one the call before the last took
  h->ovl_join_style = true;                                     // (a join after every call: from now on the kernels carry their stream's completion event)
  if (h->ovl_pending[k]) {
    if (!h->ovl_bound[k]) { HIP_TRY(hipEventRecord(h->ovl_done[k], h->ovl_stream[k]), SDRFM_FAIL); h->ovl_bound[k] = true; }   // (recorded once: it covers the stream's latest call)
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), h->ovl_done[k], 0), SDRFM_FAIL);
  }
  return SDRFM_OK;
}

int sdrfm_set_stream(sdrfm_t* h, void* hip_stream) {
  if (!h) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  return SDRFM_OK;
}

int sdrfm_synchronize(sdrfm_t* h) {
  if (!h) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  return SDRFM_OK;
}

const char* sdrfm_kernel_name(const sdrfm_t* h) { return h ? h->kernel_name : ""; }

// Do rows [a + i sa, a + i sa + la) and [b + j sb, b + j sb + lb), i, j < n, share a byte?  Exact for equal strides (two views of one
// buffer at different offsets do not); otherwise the two whole ranges are compared.
static bool rows_overlap(const uint8_t* a, size_t sa, size_t la, const uint8_t* b, size_t sb, size_t lb, uint32_t n) {
  if (!a || !b || n == 0 || la == 0 || lb == 0) return false;
  const uintptr_t ua = (uintptr_t)a, ub = (uintptr_t)b;
  if (n == 1 || sa != sb || sa == 0) {
    const uintptr_t ea = ua + (uintptr_t)(n - 1) * sa + la, eb = ub + (uintptr_t)(n - 1) * sb + lb;
    return ua < eb && ub < ea;
  }
  // row i of a and row j of b start d - (i - j) s apart (d = b - a): they share a byte iff that distance lies in (-lb, la)
  const long long s = (long long)sa, d = (long long)(ub - ua);
  long long k = d / s;                                          // i - j candidates: floor(d / s) and its neighbours
  for (long long kk = k - 1; kk <= k + 1; ++kk) {
    if (kk <= -(long long)n || kk >= (long long)n) continue;
    const long long r = d - kk * s;
    if (r > -(long long)lb && r < (long long)la) return true;
  }
  return false;
}

// Enqueue one call on device-resident buffers and advance the host-side phases / state set.
static int enqueue(sdrfm* h, const uint8_t* d_iq, size_t iq_stride, uint32_t nbytes, float* d_audio, size_t audio_stride,
                   uint32_t* n_audio, uint32_t call_flags) {
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
  p.hist_b_in = h->d_hist_b[h->cur]; p.hist_b_out = h->d_hist_b[h->cur ^ 1];
  p.h = h->d_h; p.g = h->d_g;
  p.T = c.fir_taps; p.D = c.fir_decim; p.Ta = c.audio_taps; p.Da = c.audio_decim;
  p.N = N; p.M = M; p.A = A;
  p.e0 = (int32_t)(c.fir_decim - 1 - h->phase_x);
  p.f0 = (int32_t)(c.audio_decim - 1 - h->phase_d);
  p.n_streams = c.n_streams;
  p.phase_x = h->phase_x;
  p.AB = h->AB;
  p.dbg = h->d_dbg;
  p.prio_balance = h->prio_balance;
  p.end_prio = h->end_prio;
  p.dbg_tag = (h->d_dbg && ++h->dbg_launches == 16) ? 1u : 0u;
  p.warm_ahead = h->warm_ahead;
  p.iq_prev = nullptr; p.iq_prev_stride = 0; p.N_prev = 0;
  // Design B cannot express the zero history at the start of a stream in bytes: until T-1 real samples have been seen its
  // first y_aff outputs are wrong and the generic kernel recomputes the audio that depends on them (below).  The STATE it
  // hands over (last Ta-1 discriminator outputs, y[M-1]) must not contain any of those outputs either, so a first call that
  // short runs on the generic kernel entirely.
  const uint32_t y_aff = (c.fir_taps + c.fir_decim - 1) / c.fir_decim + 1;            // outputs touching never-seen samples
  const bool short_first = h->fast && h->fast->kind == 'b' && h->n_seen + 1 < c.fir_taps && M < y_aff + c.audio_taps;
  const bool fast_ok = h->fast && A > 0 && (h->phase_x % 2 == 0) && ((uintptr_t)d_iq % 4 == 0) && (iq_stride % 4 == 0) &&
                       N < (1u << 30) && !short_first;
  // Design Q serves whole numbers of audio periods at decimator phase 0 on 16-byte aligned rows, when the call holds enough steps
  // (128 outputs each) to put at least two waves on every CU; the first call after a reset must be long enough that the state it
  // hands over holds no output computed from the (inexpressible in bytes) zero history.  It also serves one dongle's second of IQ
  // (BASELINE configs[1]: 1875 steps cut into two-step runs, 5.7 us against 9.6 - 13 us for design B).
  const uint32_t q_steps = (M + SDRFM_Q_STEP_OUT - 1) / SDRFM_Q_STEP_OUT;
  const bool q_fit = h->d_qA && A > 0 && h->phase_x == 0 && h->phase_d == 0 && (N % (c.fir_decim * c.audio_decim * 8u)) == 0 &&
                    ((uintptr_t)d_iq % 16 == 0) && (iq_stride % 16 == 0) && N < (1u << 30) && M >= c.audio_taps &&
                    (h->n_seen + 1 >= c.fir_taps || M >= y_aff + c.audio_taps) &&
                    (uint64_t)c.n_streams * q_steps >= 2ull * h->n_cu;
  // ---- which streams design Q serves at this call (the handle's comment: per-stream routing) -----------------------------------------------
  const uint32_t ns_all = c.n_streams;
  if (q_fit && h->rt_noisy && !h->rt_off) {
    // the first design-Q call of a window: the statistics of the window that last counted into this window's set take effect HERE, at a call the call
    // sequence fixes (route_take waits for a read-back that has not arrived: a host more than SETS - 1 windows ahead of its device)
    if (h->rt_win_calls == 0) { const int trc = route_take(h, h->rt_set); if (trc != SDRFM_OK) return trc; }
    route_retry(h);
    if (h->rt_dirty) {
      const int arc = route_apply(h);
      if (arc != SDRFM_OK) return arc;
    }
  }
  const uint32_t n_noisy = (q_fit && h->rt_noisy) ? h->rt_n_noisy : 0u, n_clean = ns_all - n_noisy;
  const bool q_ok = q_fit && n_clean > 0 && 2 * n_noisy < ns_all; // design Q serves n_clean streams (all of them when no stream is noisy); with half of the streams
                                                                 // noisy the bit-exact kernels take the whole batch (design S fills the machine then)
  const bool bx_all = !q_ok;                                     // the bit-exact kernels serve every stream, on the handle's stream (as every call design Q cannot take)
  const bool mixed = q_ok && n_noisy > 0;                        // ... or the noisy ones beside design Q: in its launch (k_mix) or ahead of it
  ++h->rt_calls;
  // SDRFM_F_OVERLAP: the call goes to one of two internal streams and warms every stream up from the previous call's buffer instead
  // of reading the carried state, so that it depends on nothing the previous call computes (the state sets are still written, for
  // whatever call comes next without the flag).  Any other call first orders the handle's stream behind the overlapped ones.
  // What the flag asks of the caller is checked where that is cheap: a call whose rows overlap the previous call's rows (one buffer used
  // for every call) or whose audio buffer is the one the previous overlapped call may still be writing runs as if the flag were absent.
  // sdrfm_process_batch_pcm: will design Q's launch hold the sink's chain (sdrfm_sink_chain.h)?  Every stream's whole audio row from this launch (no routed stream,
  // no outputs the generic kernel recomputes behind it at the start of a stream), enough quads for runs longer than their predecessor's reach, a sink whose time
  // constant lets runs be sunk independently.  Any other call is followed by the sink's own kernel.
  SdrfmSinkChain chain;
  const uint32_t q_quads = ((M + 7u) / 8u + 3u) / 4u;
  // (a batch with routed streams: where both designs go out in ONE launch — fuse, below — its design-Q workgroups hold the chain for the clean streams and the sink's
  // list kernel follows on the same queue for the routed ones)
  const bool fuse_early = mixed && h->mix_lds && fast_ok && h->fast_mix && M >= c.audio_taps && h->fold_state_ok;
  bool with_chain = q_ok && h->pcm_sink && (!mixed || fuse_early) && !(h->n_seen + 1 < c.fir_taps) && q_quads >= 13u &&
                   sdrfm_q_has_pcm_chain(h->q_c0, h->q_nslot, c.fir_decim, c.audio_decim) && sdrfm_sink_chain_params(h->pcm_sink, h->device, ns_all, &chain) == 2;
  if (with_chain && !h->d_runstate) {                              // per-run hand-off words: SDRFM_CHAIN_SETS sets of one word per workgroup of the largest grid
    h->runstate_cap = h->q_waves_per_cu * h->n_cu;
    if (hipMalloc(&h->d_runstate, sizeof(unsigned long long) * SDRFM_CHAIN_SETS * h->runstate_cap) != hipSuccess ||
        // (zeroed BEFORE any launch can see it: hipMemset alone may return before the device has done it, and the internal streams do not wait for the null stream —
        // a memset landing in the middle of the first launch wiped the runs' published words and their successors waited for ever)
        hipMemsetAsync(h->d_runstate, 0, sizeof(unsigned long long) * SDRFM_CHAIN_SETS * h->runstate_cap, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
      (void)hipGetLastError();
      if (h->d_runstate) (void)hipFree(h->d_runstate);
      h->d_runstate = nullptr; with_chain = false;
    }
  }
  const bool ovl = q_ok && (call_flags & SDRFM_F_OVERLAP) && h->prev_iq && h->n_seen + 1 >= c.fir_taps &&
                   h->prev_nbytes >= 2u * c.fir_decim * SDRFM_Q_STEP_OUT && (h->prev_nbytes % 16 == 0) && ((uintptr_t)h->prev_iq % 16 == 0) &&
                   (h->prev_stride % 16 == 0) &&
                   !rows_overlap(h->prev_iq, h->prev_stride, h->prev_nbytes, d_iq, iq_stride, nbytes, c.n_streams) &&
                   // (a call that writes no audio — no buffer given, the chain inside the launch — has no audio rows to collide)
                   !(!(with_chain && h->pcm_no_audio) && h->prev_ovl_audio &&
                     rows_overlap(reinterpret_cast<const uint8_t*>(h->prev_ovl_audio), h->prev_ovl_audio_stride * sizeof(float),
                                  h->prev_ovl_audio_n * sizeof(float), reinterpret_cast<const uint8_t*>(d_audio),
                                  audio_stride * sizeof(float), (size_t)A * sizeof(float), c.n_streams)) &&
                   !(with_chain && h->prev_ovl_pcm && rows_overlap(reinterpret_cast<const uint8_t*>(h->prev_ovl_pcm), h->prev_ovl_pcm_stride * sizeof(int16_t),
                                                                   h->prev_ovl_audio_n * 2 * sizeof(int16_t), reinterpret_cast<const uint8_t*>(h->pcm_out),
                                                                   h->pcm_out_stride * sizeof(int16_t), (size_t)A * 2 * sizeof(int16_t), c.n_streams));
  if (!ovl) { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
  if (bx_all) {
    // A bit-exact kernel takes over from design Q: the y[-1] it is handed must be the definition's (design Q's own is within 1e-4 of it,
    // which a small |y| would turn into a wrong d[0]): recomputed from the 64 raw samples design Q left.
    // Only the streams design Q served at the previous call (the clean part of the list in use, as in route_apply): a routed stream's y[-1] IS
    // the definition's — its bit-exact kernel wrote it — and design Q left no raw samples for it (a call design Q cannot take behind
    // a mixed call overwrote it from stale rows; tests/test_route_gpu.py::test_a_call_design_q_cannot_take_behind_mixed_calls).
    if (!h->yprev_exact && h->hist_q_valid) {
      const uint32_t nn = h->rt_noisy ? h->rt_n_noisy : 0u;
      HIP_TRY(sdrfm_q_fix_yprev(h->d_hist_q[h->cur], h->d_hpad, h->d_yprev[h->cur], c.n_streams - nn, nn ? h->rt_list_dev[h->rt_list_cur] : nullptr, h->stream),
              SDRFM_FAIL);
    }
    h->yprev_exact = true; h->hist_q_valid = false;
    h->prev_ovl_audio = nullptr;
  }
  p.slist = nullptr;
  const uint32_t* list_dev = (h->rt_noisy && n_noisy) ? h->rt_list_dev[h->rt_list_cur] : nullptr;   // [clean streams][noisy streams]
  char q_name[sizeof(h->kernel_name)] = "";
  bool halo_bytes = q_ok;                                        // a kernel that reads its halo as bytes served (some of) the call
  bool behind_in = false;                                        // overlapped call: the handle's stream holds work the launches have to be ordered behind
  if (ovl || mixed) {
    if (!h->ovl_done[1]) {                                      // (created piece by piece: a failure half-way leaves nothing half-used)
      if (!h->ovl_in) HIP_TRY(hipEventCreateWithFlags(&h->ovl_in, hipEventDisableTiming), SDRFM_ENOMEM);
      // Two streams only overlap when they sit on different hardware queues, and the runtime hands streams of one priority a small
      // shared pool of queues (two streams created back to back were seen on the same one: the calls then ran one after the other).
      // Queues are pooled per priority, so the two internal streams take the two priorities ordinary streams do not use.
      int pr_least = 0, pr_greatest = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest), SDRFM_FAIL);
      for (int i = 0; i < 2; ++i) {
        if (!h->ovl_stream[i]) HIP_TRY(hipStreamCreateWithPriority(&h->ovl_stream[i], hipStreamNonBlocking, i == 0 ? pr_greatest : pr_least), SDRFM_ENOMEM);
        if (!h->ovl_done[i]) HIP_TRY(hipEventCreateWithFlags(&h->ovl_done[i], hipEventDisableTiming), SDRFM_ENOMEM);
      }
    }
    // behind whatever the handle's stream still holds (it may produce iq); an idle stream — the steady state of a caller whose
    // buffers are filled elsewhere — costs one query instead of two packets
    if (ovl && hipStreamQuery(h->stream) != hipSuccess) {
      (void)hipGetLastError();
      HIP_TRY(hipEventRecord(h->ovl_in, h->stream), SDRFM_FAIL);
      behind_in = true;
    }
  }
  // the stream this call's launches go to: one of the two internal streams in turn (overlapped calls), or the handle's stream
  hipStream_t qs = h->stream;
  uint32_t k = 2;                                                 // its slot: internal stream 0 / 1, or 2 = the handle's stream
  if (ovl) { k = h->ovl_next; h->ovl_next ^= 1u; qs = h->ovl_stream[k]; if (behind_in) HIP_TRY(hipStreamWaitEvent(qs, h->ovl_in, 0), SDRFM_FAIL); }
  const char* bx_name = nullptr;
  // Mixed calls share the machine between the two launches: a stream costs the bit-exact kernels about twice what it costs design Q, so design Q's
  // grid is cut for its share of the CUs' wave slots (fewer runs per stream) and design B's for the LDS that leaves (its launch goes out first: its
  // waves are the longer ones).
  // Where the shape has a one-launch kernel (k_mix) both kinds of workgroup go out in ONE grid that fills the machine once: the wave slots are dealt out by
  // the same shares, a design-B segment is about as long as a design-Q run, and neither launch waits for the other.
  const bool fuse = fuse_early;
  uint32_t q_total = h->q_waves_per_cu * h->n_cu, bx_waves = h->waves_target;   // design Q's workgroups, the bit-exact kernels' waves
  if (mixed) {
    const double cost = fuse ? h->mix_cost : 2.0;
    const double share = (cost * n_noisy) / ((double)n_clean + cost * n_noisy);
    if (fuse) {
      const uint32_t total = h->mix_waves_per_cu * h->n_cu;
      bx_waves = (uint32_t)((double)total * share + 0.5);
      if (bx_waves < n_noisy) bx_waves = n_noisy;
      // The whole grid is resident at once, so the launch lasts as long as its LONGEST wave, and both kinds of wave come in whole units: a design-B segment walks
      // ceil(NA Da / 256) sub-tiles (a partly filled one costs a whole one), a design-Q run ceil((quads + runs) / runs) quads.  Around the share above, the number of
      // segments per routed stream is therefore chosen for the smaller of the two maxima — a sub-tile of design B weighs 12.7 quads of design Q in a wave's time at
      // this shape (64 taps, / 10: calibrated at 25 % routed streams, where 16 segments of six sub-tiles beside runs of 76 quads balance) —, ties for the fuller
      // sub-tiles: 28.4 -> 27.1 us per call with a quarter of the streams routed (profiles/r06_mixed_split.txt).  Other shapes keep the share as it is.
      if (c.fir_taps == 64 && c.fir_decim == 10 && h->mix_R == 4 && A >= 64 && !h->mix_split_off) {
        const uint32_t nyt = 64u * h->mix_R, qt = ((M + 7u) / 8u + 3u) / 4u, s0 = bx_waves / n_noisy;
        double best_cost = 1e30, best_eff = 0.0;
        uint32_t best_s = 0;
        for (uint32_t sg = s0 > 6u ? s0 - 6u : 1u; sg <= s0 + 4u; ++sg) {
          const uint32_t na = (A + sg - 1u) / sg, tiles = (A + na - 1u) / na;
          if ((uint64_t)tiles * n_noisy + n_clean > total) break;
          const uint32_t sub = (na * c.audio_decim + nyt - 1u) / nyt;
          uint32_t rr = (total - tiles * n_noisy) / n_clean;
          if (rr > q_steps / 2u) rr = q_steps / 2u;
          if (rr < 1u) continue;
          const uint32_t quads = (qt + rr + rr - 1u) / rr;
          const double cost = (double)sub * h->mix_rho > (double)quads ? (double)sub * h->mix_rho : (double)quads;
          const double eff = (double)(na * c.audio_decim) / (double)(sub * nyt);
          if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && eff > best_eff)) { best_cost = cost; best_eff = eff; best_s = tiles; }
        }
        if (best_s) bx_waves = best_s * n_noisy;
      }
      q_total = total > bx_waves + n_clean ? total - bx_waves : n_clean;
    } else {
      uint32_t q_slots = (uint32_t)((double)h->q_waves_per_cu * (1.0 - share) + 0.5);
      if (q_slots + 1 > h->q_waves_per_cu) q_slots = h->q_waves_per_cu - 1;
      if (q_slots < 2) q_slots = 2;
      const size_t q_lds = sdrfm_q_lds_bytes(h->q_nslot, c.fir_decim, c.audio_decim);
      const size_t left = 163840u > q_slots * q_lds ? 163840u - q_slots * q_lds : 0u;
      const size_t b_lds = h->fast_mix ? h->fast_mix_lds : h->fast_lds;
      uint32_t per_cu = b_lds ? (uint32_t)(left / b_lds) : 1u;
      if (per_cu < 1) per_cu = 1;
      bx_waves = h->n_cu * per_cu;
      q_total = q_slots * h->n_cu;
    }
  }
  CallParams pb;                                                  // fuse: design B's part of the one launch
  uint32_t pb_blocks = 0, pb_R = 0;
  // (the bit-exact launch as a closure)
  auto launch_bx = [&]() -> int {
    // ---- the bit-exact kernels: every stream (on the handle's stream), or the noisy streams of a mixed call: inside design Q's launch where k_mix has an
    // instance (fuse: only the parameters are prepared here), else a launch of their own ahead of design Q's on the call's stream
    const uint32_t nsub = bx_all ? ns_all : n_noisy;
    hipStream_t bs = h->stream;
    hipEvent_t bdone = nullptr;                                   // mixed: the launch's own completion event (a stop event: no marker packet)
    if (mixed && !fuse) {
      // The noisy streams' launch goes ahead of design Q's on the call's own stream: no third stream (streams of one priority share a small pool of
      // hardware queues — one was seen on the queue of the handle's stream, and every dependence across queues costs ~10 us of a queue's time), no
      // fork and join.  Overlapped calls run it beside the OTHER internal stream's call; it takes the state the previous such launch left, which sits
      // on that other stream then: behind its completion event (launched a whole call earlier: a wait that is over when it is reached).
      bs = qs;
      bdone = h->rt_bx_evt[h->rt_bx_slot]; h->rt_bx_slot ^= 1u;
      if (h->rt_bx_last && h->rt_bx_last != bs) HIP_TRY(hipStreamWaitEvent(bs, h->rt_bx_evt[h->rt_bx_slot], 0), SDRFM_FAIL);   // (the slot just left: the previous launch's event)
      h->rt_bx_last = bs;
    }
    if (mixed) p.slist = list_dev + n_clean;
    p.n_streams = nsub;
    const bool stream_ok = fast_ok && h->fast_s && h->phase_x == 0 && h->phase_d == 0 && (N % h->fast_s->seg) == 0 &&
                           (M % c.audio_decim) == 0 && M >= c.audio_taps && ((uintptr_t)d_iq % 16 == 0) && (iq_stride % 16 == 0) &&
                           h->fold_state_ok &&
                           // a lane-segment wave is long (its 64 lanes walk 480 samples each, ~25 us alone on a SIMD): design S pays when
                           // the launch fills the machine (>= one wave per SIMD); a single dongle's call is served faster by design B,
                           // which cuts its segments as short as the call needs
                           (uint64_t)nsub * ((N / h->fast_s->seg + 62) / 63) >= 4ull * h->n_cu &&
                           !mixed;   // (beside design Q's launch: design B's small tile — a wave of design S needs 16 KB of LDS and lasts 25 us)
    const char* bname = h->generic_name;
    if (stream_ok) {
      const uint32_t segs = N / h->fast_s->seg;
      p.tiles_per_stream = (segs + 62) / 63;                      // waves per stream: 63 useful lane segments each
      p.fold_state = 1;
      hipExtLaunchKernelGGL(h->fast_s->kernel[0], dim3(nsub * p.tiles_per_stream), dim3(64), h->fast_s->xbytes, bs, nullptr, bdone, 0, p);
      bname = h->fast_s_name; halo_bytes = true;
    } else if (fast_ok) {
      // split every stream into segments so that ~waves_target waves are resident; each segment >= min_subtiles sub-tiles
      const FastVariant* fv = (mixed && h->fast_mix) ? h->fast_mix : h->fast;
      const size_t flds = (mixed && h->fast_mix) ? h->fast_mix_lds : h->fast_lds;
      const uint32_t NYT = 64 * (fuse ? h->mix_R : fv->R);
      const uint32_t sub_total = (M + NYT - 1) / NYT;
      uint32_t segs = bx_waves / nsub;
      // a segment pays a fixed prologue, so it normally covers >= min_subtiles sub-tiles; when that would leave most of the
      // GPU without a wave (few streams: the reference's one dongle), shorter segments win: one stream x 1 s runs in 9.6 us
      // with single-sub-tile segments against 18.8 us with four
      uint32_t ms = h->min_subtiles;
      while (ms > 1 && (uint64_t)nsub * (sub_total / ms) < bx_waves / 2) ms >>= 1;
      const uint32_t seg_cap = sub_total / ms;
      if (segs > seg_cap) segs = seg_cap;
      if (segs < 1) segs = 1;
      p.NA = (A + segs - 1) / segs;
      p.tiles_per_stream = (A + p.NA - 1) / p.NA;
      uint32_t grid = nsub * p.tiles_per_stream + nsub;
      // design B: state hand-over folded into the last segment's wave (needs M >= Ta so that the d ring alone holds the
      // new history, and the last sub-tile must contain y[M-1], which the kernel arranges)
      p.fold_state = (fv->kind == 'b' && M >= c.audio_taps && h->fold_state_ok) ? 1u : 0u;
      if (p.fold_state) grid -= nsub;
      if (fuse) {
        // inside design Q's launch (below); an overlapped call reads what lies before the call from the previous call's buffer, as design Q does:
        // no state of the previous call — which may still be running on the other internal stream — is read
        if (ovl) { p.iq_prev = h->prev_iq; p.iq_prev_stride = h->prev_stride; p.N_prev = h->prev_nbytes / 2; }
        pb = p; pb_blocks = grid; pb_R = h->mix_R;
        p.iq_prev = nullptr; p.iq_prev_stride = 0; p.N_prev = 0;
      } else {
        hipExtLaunchKernelGGL(fv->kernel[fv == h->fast ? h->fast_mode : 0], dim3(grid), dim3(64), flds, bs, nullptr, bdone, 0, p);
      }
      bname = h->fast_name; halo_bytes = halo_bytes || fv->kind == 'b';
    } else {
      p.NA = h->NA;
      p.tiles_per_stream = (A + h->NA - 1) / h->NA;
      const uint32_t grid = nsub * p.tiles_per_stream + nsub;
      hipExtLaunchKernelGGL(k_generic, dim3(grid), dim3(256), h->lds_bytes, bs, nullptr, bdone, 0, p);
    }
    bx_name = bname;
    p.slist = nullptr; p.n_streams = ns_all;
    return SDRFM_OK;
  };
  if (bx_all || mixed) { const int brc = launch_bx(); if (brc != SDRFM_OK) return brc; }
  if (q_ok) {
    SdrfmQParams q;
    q.iq_prev = nullptr; q.iq_prev_stride = 0; q.N_prev = 0;
    if (ovl) { q.iq_prev = h->prev_iq; q.iq_prev_stride = h->prev_stride; q.N_prev = h->prev_nbytes / 2; }
    q.iq = d_iq; q.iq_stride = iq_stride; q.audio = d_audio; q.audio_stride = audio_stride;
    q.yprev_in = p.yprev_in; q.yprev_out = p.yprev_out; q.hist_d_in = p.hist_d_in; q.hist_d_out = p.hist_d_out;
    q.hist_b_in = p.hist_b_in; q.hist_b_out = p.hist_b_out; q.hist_x_out = p.hist_x_out;
    q.A = h->d_qA; q.g = h->d_g; q.q0 = h->q_scale; q.q2 = 65536.0f * h->q_scale; q.cst = h->q_cst;
    q.T = c.fir_taps; q.N = N; q.M = M; q.A_out = A; q.steps_total = q_steps; q.n_streams = n_clean; q.dbg = nullptr; q.prio_by_age = ovl ? 0u : 1u;
    q.slist = list_dev;                                           // (its first n_clean entries)
    // the guard's repair path reads the last 64 raw samples before the call: left by the previous design-Q call, or taken now from the
    // T - 1 the other kernels keep (the 64th, which only y[-1] needs, is then not there — nor needed: their y[-1] is the definition's)
    if (!h->hist_q_valid && !ovl && c.fir_taps > 1)
      HIP_TRY(hipMemcpy2DAsync(h->d_hist_q[h->cur] + 2 * (SDRFM_Q_TP - (c.fir_taps - 1)), 2 * SDRFM_Q_TP, p.hist_b_in, 2 * (size_t)(c.fir_taps - 1),
                               2 * (size_t)(c.fir_taps - 1), c.n_streams, hipMemcpyDeviceToDevice, h->stream), SDRFM_FAIL);
    q.hpad = h->d_hpad; q.hist_q_in = h->d_hist_q[h->cur]; q.hist_q_out = h->d_hist_q[h->cur ^ 1];
    q.guard_r = h->q_guard_r; q.guard_a = h->q_guard_a; q.yprev_exact = h->yprev_exact ? 1u : 0u; q.n_repaired = h->d_qstat;
    // runs (waves) per stream: fill the machine once; every run at least four owned steps (a run warms up over a quarter of a step), two
    // when the call is too small to fill the machine otherwise
    uint32_t runs = q_total / n_clean;
    const uint32_t min_steps = ((uint64_t)n_clean * (q_steps / 4) >= (uint64_t)q_total / 2) ? 4u : 2u;
    if (runs > q_steps / min_steps) runs = q_steps / min_steps;
    // (the sink's chain inside the launch: every run must own more outputs than its predecessor's state reaches — 13 quads: 12 owned = 76 audio outputs >=
    // SDRFM_CHAIN_FIX —: a small call is cut into fewer runs for it)
    if (with_chain && runs > q_quads / 13u) runs = q_quads / 13u;
    if (runs < 1) runs = 1;
    q.runs = runs;
    // the window of per-stream repair statistics this call adds to (none while the set's previous read-back is still under way), and
    // whether its kernel's completion carries one of the window's events (the window's last two calls: both internal streams are covered)
    hipEvent_t done = nullptr;
    q.stream_pass = nullptr;
    if (h->rt_noisy && !h->rt_off) {                              // (every window counts: its set was taken in at the window's first call)
      q.stream_pass = h->rt_pass_dev[h->rt_set];
      h->rt_win_used[k] = true;
      if (h->rt_win_calls + 2 >= SDRFM_Q_ADAPT_WINDOW) { done = h->rt_win_evt[h->rt_set][k]; h->rt_win_need[k] = false; }
      else h->rt_win_need[k] = true;
    }
    // For a caller that joins after every call (sdrfm_flush_previous: the consumer loop of INTEGRATION.md) an overlapped call's kernel carries its internal
    // stream's completion event itself (a stop event: the dispatch's own completion signal), so that the join orders the handle's stream behind it WITHOUT a marker
    // packet in that queue: 31 -> 29.4 us per call of that loop.  The kernel is the last thing the call puts on that stream, and a stream completes in order: the
    // event also stands for the window's kernels on it.  A caller that joins once in a while (sdrfm_flush at the end of a burst: bench.py) keeps kernels without a
    // stop event and pays one marker per join: under the kernel tracer a stop event on every kernel shortens the time two kernels are resident (0.96 -> 0.75 of a
    // 100-call burst, +3 % per call; un-profiled within the noise): profiles/r05_q_experiments.txt item 15.
    const bool carry = ovl && h->ovl_join_style;
    if (carry) { done = h->ovl_done[k]; if (h->rt_win_used[k]) h->rt_win_need[k] = false; }
    if (with_chain && (uint64_t)n_clean * runs > h->runstate_cap) with_chain = false;
    if (fuse && pb_blocks && with_chain) {
      // one launch of both designs with the chain in design Q's workgroups, then the sink's list kernel for the routed streams on the same queue (their audio is the
      // launch's design-B workgroups'); the call's completion event rides on that last kernel
      chain.pcm = h->pcm_out; chain.pcm_stride = h->pcm_out_stride;
      if (h->pcm_no_audio) q.audio = nullptr;                      // (the clean streams store no audio; the routed ones' goes to the library's own rows)
      chain.runstate = h->d_runstate + (size_t)(chain.call % SDRFM_CHAIN_SETS) * h->runstate_cap;
      HIP_TRY(sdrfm_q_launch_mix_pcm(q, chain, h->q_c0, h->q_nslot, c.fir_decim, c.audio_decim, pb, pb_blocks, pb_R, qs, nullptr), SDRFM_FAIL);
      { const int lrc = sdrfm_sink_launch_list_on(h->pcm_sink, chain, list_dev + n_clean, n_noisy, d_audio, audio_stride, A, h->pcm_out, h->pcm_out_stride, qs, done);
        if (lrc != SDRFM_OK) return lrc; }
      sdrfm_sink_chain_issued(h->pcm_sink);
      h->pcm_fused = true;
    }
    else if (fuse && pb_blocks) HIP_TRY(sdrfm_q_launch_mix(q, h->q_c0, h->q_nslot, c.fir_decim, c.audio_decim, pb, pb_blocks, pb_R, qs, done), SDRFM_FAIL);
    else if (with_chain) {
      chain.pcm = h->pcm_out; chain.pcm_stride = h->pcm_out_stride;
      if (h->pcm_no_audio) q.audio = nullptr;
      chain.runstate = h->d_runstate + (size_t)(chain.call % SDRFM_CHAIN_SETS) * h->runstate_cap;
      HIP_TRY(sdrfm_q_launch_pcm(q, chain, h->q_c0, h->q_nslot, c.fir_decim, c.audio_decim, qs, done), SDRFM_FAIL);
      sdrfm_sink_chain_issued(h->pcm_sink);
      h->pcm_fused = true;
    }
    else HIP_TRY(sdrfm_q_launch(q, h->q_c0, h->q_nslot, c.fir_decim, c.audio_decim, qs, done), SDRFM_FAIL);
    if (ovl) { h->ovl_pending[k] = true; h->ovl_bound[k] = carry; }
    h->prev_ovl_audio = (ovl && !(with_chain && h->pcm_no_audio)) ? d_audio : nullptr; h->prev_ovl_audio_stride = audio_stride; h->prev_ovl_audio_n = A;
    h->prev_ovl_pcm = (ovl && with_chain) ? h->pcm_out : nullptr; h->prev_ovl_pcm_stride = h->pcm_out_stride;
    h->yprev_exact = false; h->hist_q_valid = true;
    if (h->rt_noisy && !h->rt_off) {
      h->rt_win_stages += (uint64_t)runs * ((q_steps / runs + c.audio_decim - 1) / c.audio_decim);   // audio stages of ONE stream's waves in this call
      if (++h->rt_win_calls >= SDRFM_Q_ADAPT_WINDOW) {
        // the window closes: read its statistics back behind its last kernels, on the side stream (nothing here waits, no packet enters a compute queue
        // unless a stream holds kernels of the window that none of the attached events covers — call patterns other than "all on one stream" or "two in turn")
        hipStream_t wst[3] = {h->ovl_stream[0], h->ovl_stream[1], h->stream};
        const uint32_t i = h->rt_set;
        for (int kk = 0; kk < 3; ++kk) {
          if (!h->rt_win_used[kk]) continue;
          if (h->rt_win_need[kk]) HIP_TRY(hipEventRecord(h->rt_win_evt[i][kk], wst[kk]), SDRFM_FAIL);
          // (an internal stream's latest kernel carries ovl_done[kk] — see above —: at or behind the window's last kernel on that stream)
          HIP_TRY(hipStreamWaitEvent(h->rt_mon, (kk < 2 && h->ovl_bound[kk]) ? h->ovl_done[kk] : h->rt_win_evt[i][kk], 0), SDRFM_FAIL);
          h->rt_win_used[kk] = false; h->rt_win_need[kk] = false;
        }
        // (one small kernel instead of a copy and a memset: 12 us of host time instead of 40 — a window that closes while the host is only a call ahead
        // of the device, as in a 20-call burst from rest, otherwise leaves the queues dry for the difference)
        hipLaunchKernelGGL(k_route_collect, dim3((ns_all + 255) / 256), dim3(256), 0, h->rt_mon, h->rt_pass_dev[i], h->rt_pass_host_dev[i], ns_all);
        HIP_TRY(hipEventRecord(h->rt_rb_done[i], h->rt_mon), SDRFM_FAIL);
        h->rt_rb_pending[i] = true; h->rt_rb_stages[i] = h->rt_win_stages;
        h->rt_set = (h->rt_set + 1u) % SDRFM_Q_ADAPT_SETS; h->rt_win_calls = 0; h->rt_win_stages = 0;
      }
    }
    snprintf(q_name, sizeof(q_name), "%s%s%s", h->fast_q_name, ovl ? " overlapped" : "", h->pcm_fused ? " + pcm" : "");
  }
  if (mixed) {
    const char* sp = strchr(bx_name, ' ');                         // ("fast-b", "fast-s", "generic": the name's first word)
    snprintf(h->kernel_name, sizeof(h->kernel_name), "%s + %.*s (%u streams)%s", q_name, (int)(sp ? sp - bx_name : (long)strlen(bx_name)), bx_name, n_noisy,
             pb_blocks ? " in one launch" : "");
  } else {
    snprintf(h->kernel_name, sizeof(h->kernel_name), "%s", bx_all ? bx_name : q_name);
  }
  if (halo_bytes && h->n_seen + 1 < c.fir_taps) {
    // Designs Q, B and S read their halo as bytes, which cannot express the zero history at the start of a stream: the few audio
    // outputs that depend on inputs before the first real sample are recomputed by the generic kernel (tile 0..k only), for every stream
    // (a call this early is never overlapped, so the handle's stream is behind every launch above).
    const uint32_t a_aff = (y_aff + c.audio_taps + c.audio_decim - 1) / c.audio_decim;  // audio outputs touching them
    CallParams q = p;
    q.NA = h->NA;
    q.tiles_per_stream = (a_aff + h->NA - 1) / h->NA;
    const uint32_t full = (A + h->NA - 1) / h->NA;
    if (q.tiles_per_stream > full) q.tiles_per_stream = full;
    hipLaunchKernelGGL(k_generic, dim3(c.n_streams * q.tiles_per_stream), dim3(256), h->lds_bytes, h->stream, q);
  }
  HIP_TRY(hipGetLastError(), SDRFM_FAIL);

  h->cur ^= 1;
  h->n_seen += N;
  if (call_flags & SDRFM_F_DEVICE_PTRS) { h->prev_iq = d_iq; h->prev_stride = iq_stride; h->prev_nbytes = nbytes; }
  else h->prev_iq = nullptr;
  h->phase_x = (uint32_t)(((uint64_t)h->phase_x + N) % c.fir_decim);
  h->phase_d = (uint32_t)(((uint64_t)h->phase_d + M) % c.audio_decim);
  return SDRFM_OK;
}

int sdrfm_process_batch(sdrfm_t* h, const uint8_t* iq, size_t iq_stride, uint32_t nbytes, float* audio,
                        size_t audio_stride, uint32_t* n_audio, uint32_t flags) {
  if (!h || !n_audio) return SDRFM_EINVAL;
  if (flags & ~(SDRFM_F_DEVICE_PTRS | SDRFM_F_OVERLAP)) return SDRFM_EINVAL;
  if ((flags & SDRFM_F_OVERLAP) && !(flags & SDRFM_F_DEVICE_PTRS)) return SDRFM_EINVAL;
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

  if (flags & SDRFM_F_DEVICE_PTRS) return enqueue(h, iq, iq_stride, nbytes, audio, audio_stride, n_audio, flags);

  // host buffers: stage -> kernels -> copy back, synchronous (the caller may re-arm `iq` as soon as we return,
  // like the reference FSM does with CommItf.buff)
  if (nbytes > h->max_bytes) return SDRFM_ECAPACITY;
  if (ns == 1 && nbytes <= SDRFM_ZC_MAX && !h->zc_off) {
    int zrc = ensure_zero_copy(h);
    if (zrc != SDRFM_OK) return zrc;
    memcpy(h->zc_iq, iq, nbytes);
    zrc = enqueue(h, h->zc_iq_dev, SDRFM_ZC_MAX + 256, nbytes, h->zc_audio_dev, h->zc_audio_cap, n_audio);
    if (zrc != SDRFM_OK) return zrc;
    HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
    if (A) memcpy(audio, h->zc_audio, sizeof(float) * A);
    return SDRFM_OK;
  }
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

/* One call of the demodulator AND of the PCM sink (include/sdrfm.h).  Where design Q serves the whole call its launch ends with the sink's chain
 * (sdrfm_sink_chain.h: no second launch, no queue to wait on; a batch with routed streams: the sink's list kernel for those on the call's own queue); any other
 * call — bit-exact kernels alone, the first call of a stream — is followed
 * by the sink's stand-alone kernel on the handle's stream behind the call. */
int sdrfm_process_batch_pcm(sdrfm_t* h, sdrfm_pcm_sink_t* sink, const uint8_t* iq, size_t iq_stride, uint32_t nbytes, float* audio, size_t audio_stride,
                            int16_t* pcm, size_t pcm_stride, uint32_t* n_audio, uint32_t flags) {
  if (!h || !sink || !n_audio) return SDRFM_EINVAL;
  if (flags & ~(SDRFM_F_DEVICE_PTRS | SDRFM_F_OVERLAP)) return SDRFM_EINVAL;
  if ((flags & SDRFM_F_OVERLAP) && !(flags & SDRFM_F_DEVICE_PTRS)) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  if (nbytes == 0) { *n_audio = 0; return SDRFM_OK; }
  if (!(flags & SDRFM_F_DEVICE_PTRS)) {
    // Host buffers: stage -> the call below on the staging buffers -> copy back; synchronous, as sdrfm_process_batch with host buffers (the caller may re-arm `iq`
    // as soon as this returns, like the reference FSM does with CommItf.buff, and hand `pcm` to BSP_AUDIO_OUT_Play).
    if (!iq) return SDRFM_EINVAL;
    const uint32_t ns = h->cfg.n_streams;
    if (nbytes > h->max_bytes) return SDRFM_ECAPACITY;
    if (ns > 1 && iq_stride < nbytes) return SDRFM_ECAPACITY;
    uint32_t A = 0;
    (void)sdrfm_audio_count(h, nbytes, &A);
    if (A && (!pcm || (pcm_stride & 1u))) return SDRFM_EINVAL;
    if (ns > 1 && (pcm_stride < 2 * (size_t)A || (audio && audio_stride < A))) return SDRFM_ECAPACITY;
    HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
    int rc = ensure_staging(h);
    if (rc != SDRFM_OK) return rc;
    const size_t pstride = 2 * h->d_audio_stride;
    if (!h->d_pcm_stage) HIP_TRY(hipMalloc(&h->d_pcm_stage, sizeof(int16_t) * pstride * ns), SDRFM_ENOMEM);
    HIP_TRY(hipMemcpy2DAsync(h->d_iq, h->d_iq_stride, iq, ns > 1 ? iq_stride : nbytes, nbytes, ns, hipMemcpyHostToDevice, h->stream), SDRFM_FAIL);
    rc = sdrfm_process_batch_pcm(h, sink, h->d_iq, h->d_iq_stride, nbytes, audio ? h->d_audio : nullptr, h->d_audio_stride, h->d_pcm_stage, pstride, n_audio,
                                 SDRFM_F_DEVICE_PTRS);
    if (rc != SDRFM_OK) return rc;
    { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
    if (*n_audio) {
      HIP_TRY(hipMemcpy2DAsync(pcm, (ns > 1 ? pcm_stride : 2 * (size_t)*n_audio) * sizeof(int16_t), h->d_pcm_stage, pstride * sizeof(int16_t),
                               2 * (size_t)*n_audio * sizeof(int16_t), ns, hipMemcpyDeviceToHost, h->stream), SDRFM_FAIL);
      if (audio)
        HIP_TRY(hipMemcpy2DAsync(audio, (ns > 1 ? audio_stride : *n_audio) * sizeof(float), h->d_audio, h->d_audio_stride * sizeof(float),
                                 *n_audio * sizeof(float), ns, hipMemcpyDeviceToHost, h->stream), SDRFM_FAIL);
    }
    HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
    return SDRFM_OK;
  }
  uint32_t A = 0;
  (void)sdrfm_audio_count(h, nbytes, &A);
  if (A && (!pcm || ((uintptr_t)pcm & 3u) || (pcm_stride & 1u))) return SDRFM_EINVAL;
  if (h->cfg.n_streams > 1 && pcm_stride < 2 * (size_t)A) return SDRFM_ECAPACITY;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  const bool no_audio = audio == nullptr;
  if (no_audio && A) {                                           // the library's own rows, for the calls that need some (every call but those whose launch holds the chain)
    if (A > h->pcm_audio_cap) {
      { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
      HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
      if (h->d_pcm_audio) (void)hipFree(h->d_pcm_audio);
      h->d_pcm_audio = nullptr; h->pcm_audio_cap = 0;
      const uint32_t cap = (A + 63u) & ~63u;
      if (hipMalloc(&h->d_pcm_audio, 2 * sizeof(float) * (size_t)cap * h->cfg.n_streams) != hipSuccess) { (void)hipGetLastError(); return SDRFM_ENOMEM; }
      h->pcm_audio_cap = cap; h->pcm_audio_stride = cap;
    }
    // two sets of rows in turn: the routed streams of an overlapped call write theirs while the previous call's are still being read
    h->pcm_audio_flip ^= 1u;
    audio = h->d_pcm_audio + (size_t)h->pcm_audio_flip * h->pcm_audio_cap * h->cfg.n_streams; audio_stride = h->pcm_audio_stride;
  }
  SdrfmSinkChain probe;
  if (sdrfm_sink_chain_params(sink, h->device, h->cfg.n_streams, &probe) == 0) return SDRFM_EINVAL;   // (a sink of this device and this many streams)
  h->pcm_sink = sink; h->pcm_out = pcm; h->pcm_out_stride = pcm_stride; h->pcm_fused = false; h->pcm_no_audio = no_audio;
  const int rc = sdrfm_process_batch(h, iq, iq_stride, nbytes, audio, audio_stride, n_audio, flags);
  const bool fused = h->pcm_fused;
  h->pcm_sink = nullptr; h->pcm_fused = false; h->pcm_no_audio = false;
  if (rc != SDRFM_OK || fused || *n_audio == 0) return rc;
  { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
  return sdrfm_sink_launch_on(sink, audio, audio_stride, *n_audio, pcm, pcm_stride, h->stream);
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

/* ------------------------------------------------------------------------------------------------------------------
 * Streaming front-end adapter: a ring of pinned host buffers, the multi-buffer scheme the reference
 * declares but never uses (DEFAULT_BUF_NUMBER 15 x DEFAULT_BUF_LENGTH 16*32*512, Class/RTLSDR/Inc/usbh_rtlsdr.h:277-278).
 * submit() copies the just-filled USB buffer into the next free slot and enqueues H2D -> kernel -> D2H without waiting;
 * collect() hands back finished audio in order.  Both are non-blocking and answer SDRFM_BUSY like the reference's FSM
 * steps do (USBH_BUSY, usbh_def.h:303-311).  H2D of slot k+1 overlaps the kernel of slot k (separate copy streams).
 * ------------------------------------------------------------------------------------------------------------------ */
struct sdrfm_ring {
  sdrfm* h;
  uint32_t n, slot_bytes, audio_cap;
  uint8_t** host_iq;      // pinned
  float** host_audio;     // pinned
  uint8_t** dev_iq;
  float** dev_audio;
  uint32_t* n_audio;      // per slot
  hipEvent_t *ev_h2d, *ev_kernel, *ev_done;
  hipStream_t s_h2d, s_d2h;
  uint32_t head, tail, count;   // head: next slot to submit; tail: oldest slot in flight
};

static void ring_free(sdrfm_ring* r) {
  if (!r) return;
  (void)hipSetDevice(r->h->device);
  if (r->s_h2d) (void)hipStreamSynchronize(r->s_h2d);
  (void)hipStreamSynchronize(r->h->stream);
  if (r->s_d2h) (void)hipStreamSynchronize(r->s_d2h);
  for (uint32_t i = 0; i < r->n; ++i) {
    if (r->host_iq && r->host_iq[i]) (void)hipHostFree(r->host_iq[i]);
    if (r->host_audio && r->host_audio[i]) (void)hipHostFree(r->host_audio[i]);
    if (r->dev_iq && r->dev_iq[i]) (void)hipFree(r->dev_iq[i]);
    if (r->dev_audio && r->dev_audio[i]) (void)hipFree(r->dev_audio[i]);
    if (r->ev_h2d && r->ev_h2d[i]) (void)hipEventDestroy(r->ev_h2d[i]);
    if (r->ev_kernel && r->ev_kernel[i]) (void)hipEventDestroy(r->ev_kernel[i]);
    if (r->ev_done && r->ev_done[i]) (void)hipEventDestroy(r->ev_done[i]);
  }
  if (r->s_h2d) (void)hipStreamDestroy(r->s_h2d);
  if (r->s_d2h) (void)hipStreamDestroy(r->s_d2h);
  free(r->host_iq); free(r->host_audio); free(r->dev_iq); free(r->dev_audio); free(r->n_audio);
  free(r->ev_h2d); free(r->ev_kernel); free(r->ev_done);
  delete r;
}

int sdrfm_ring_create(sdrfm_t* h, uint32_t n_buffers, uint32_t buffer_bytes, sdrfm_ring_t** out) {
  if (!out) return SDRFM_EINVAL;
  *out = nullptr;
  if (!h || h->cfg.n_streams != 1 || n_buffers < 2 || n_buffers > 64 || !buffer_bytes || (buffer_bytes & 1u)) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  sdrfm_ring* r = new (std::nothrow) sdrfm_ring();
  if (!r) return SDRFM_ENOMEM;
  memset(static_cast<void*>(r), 0, sizeof(*r));
  r->h = h; r->n = n_buffers; r->slot_bytes = buffer_bytes;
  r->audio_cap = max_audio_for(h->cfg, buffer_bytes);
  r->host_iq = (uint8_t**)calloc(n_buffers, sizeof(void*)); r->host_audio = (float**)calloc(n_buffers, sizeof(void*));
  r->dev_iq = (uint8_t**)calloc(n_buffers, sizeof(void*)); r->dev_audio = (float**)calloc(n_buffers, sizeof(void*));
  r->n_audio = (uint32_t*)calloc(n_buffers, sizeof(uint32_t));
  r->ev_h2d = (hipEvent_t*)calloc(n_buffers, sizeof(hipEvent_t)); r->ev_kernel = (hipEvent_t*)calloc(n_buffers, sizeof(hipEvent_t));
  r->ev_done = (hipEvent_t*)calloc(n_buffers, sizeof(hipEvent_t));
  bool ok = r->host_iq && r->host_audio && r->dev_iq && r->dev_audio && r->n_audio && r->ev_h2d && r->ev_kernel && r->ev_done;
  ok = ok && hipStreamCreateWithFlags(&r->s_h2d, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&r->s_d2h, hipStreamNonBlocking) == hipSuccess;
  for (uint32_t i = 0; ok && i < n_buffers; ++i) {
    ok = ok && hipHostMalloc((void**)&r->host_iq[i], buffer_bytes, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&r->host_audio[i], sizeof(float) * r->audio_cap, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&r->dev_iq[i], ((size_t)buffer_bytes + 255) & ~(size_t)255) == hipSuccess;
    ok = ok && hipMalloc((void**)&r->dev_audio[i], sizeof(float) * r->audio_cap) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&r->ev_h2d[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&r->ev_kernel[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&r->ev_done[i], hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) { ring_free(r); return SDRFM_ENOMEM; }
  *out = r;
  return SDRFM_OK;
}

void sdrfm_ring_destroy(sdrfm_ring_t* r) { ring_free(r); }

int sdrfm_ring_submit(sdrfm_ring_t* r, const uint8_t* iq, uint32_t nbytes) {
  if (!r || (nbytes && !iq)) return SDRFM_EINVAL;
  if (nbytes & 1u) return SDRFM_EODD;
  if (nbytes > r->slot_bytes) return SDRFM_ECAPACITY;
  if (nbytes == 0) return SDRFM_OK;
  if (r->count == r->n) return SDRFM_BUSY;                    // every slot in flight: collect first
  sdrfm* h = r->h;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  const uint32_t s = r->head;
  memcpy(r->host_iq[s], iq, nbytes);                          // the caller's (single, reused) buffer is free again after this
  HIP_TRY(hipMemcpyAsync(r->dev_iq[s], r->host_iq[s], nbytes, hipMemcpyHostToDevice, r->s_h2d), SDRFM_FAIL);
  HIP_TRY(hipEventRecord(r->ev_h2d[s], r->s_h2d), SDRFM_FAIL);
  HIP_TRY(hipStreamWaitEvent(h->stream, r->ev_h2d[s], 0), SDRFM_FAIL);
  const int rc = enqueue(h, r->dev_iq[s], nbytes, nbytes, r->dev_audio[s], r->audio_cap, &r->n_audio[s]);
  if (rc != SDRFM_OK) return rc;
  HIP_TRY(hipEventRecord(r->ev_kernel[s], h->stream), SDRFM_FAIL);
  HIP_TRY(hipStreamWaitEvent(r->s_d2h, r->ev_kernel[s], 0), SDRFM_FAIL);
  if (r->n_audio[s])
    HIP_TRY(hipMemcpyAsync(r->host_audio[s], r->dev_audio[s], sizeof(float) * r->n_audio[s], hipMemcpyDeviceToHost, r->s_d2h), SDRFM_FAIL);
  HIP_TRY(hipEventRecord(r->ev_done[s], r->s_d2h), SDRFM_FAIL);
  r->head = (s + 1) % r->n;
  ++r->count;
  return SDRFM_OK;
}

int sdrfm_ring_collect(sdrfm_ring_t* r, float* audio, uint32_t audio_cap, uint32_t* n_audio, int wait) {
  if (!r || !n_audio) return SDRFM_EINVAL;
  *n_audio = 0;
  if (r->count == 0) return SDRFM_BUSY;                       // nothing in flight
  HIP_TRY(hipSetDevice(r->h->device), SDRFM_FAIL);
  const uint32_t s = r->tail;
  if (wait) {
    HIP_TRY(hipEventSynchronize(r->ev_done[s]), SDRFM_FAIL);
  } else {
    const hipError_t q = hipEventQuery(r->ev_done[s]);
    if (q == hipErrorNotReady) return SDRFM_BUSY;
    if (q != hipSuccess) return SDRFM_FAIL;
  }
  if (r->n_audio[s] > audio_cap || (r->n_audio[s] && !audio)) return SDRFM_ECAPACITY;
  memcpy(audio, r->host_audio[s], sizeof(float) * r->n_audio[s]);
  *n_audio = r->n_audio[s];
  r->tail = (s + 1) % r->n;
  --r->count;
  return SDRFM_OK;
}

#ifdef SDRFM_DEV   // counters of the instrumented kernels: development library only
/* Profiling aid (SDRFM_PHASE_PROFILE=1 at create): cumulative shader cycles per phase of the fast kernel, summed over
 * waves: out[0..4] = stage, FIR, discriminator, audio, carry; out[5] = sub-tiles; out[6] = waves. Resets the counters. */
int sdrfm_debug_phase_cycles(sdrfm_t* h, unsigned long long* out8) {
  if (!h || !out8) return SDRFM_EINVAL;
  if (!h->d_dbg) return SDRFM_NOT_SUPPORTED;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  unsigned long long tmp[512];
  HIP_TRY(hipMemcpy(tmp, h->d_dbg, sizeof(tmp), hipMemcpyDeviceToHost), SDRFM_FAIL);
  HIP_TRY(hipMemset(h->d_dbg, 0, sizeof(tmp)), SDRFM_FAIL);
  (void)hipMemset(h->d_dbg + 512, 0, 48 * sizeof(unsigned long long));
  for (int x = 0; x < 8; ++x) { (void)hipMemset(h->d_dbg + 520 + 4 * x, 0xff, 8); (void)hipMemset(h->d_dbg + 522 + 4 * x, 0xff, 8); }
  h->dbg_launches = 0;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  for (int g = 0; g < 64; ++g)
    for (int i = 0; i < 8; ++i) out8[i] += tmp[8 * g + i];
  return SDRFM_OK;
}

#endif

/* Test hook: K3 evaluated ON THE DEVICE for n operand sets (host arrays in, host arrays out): out_scalar uses the scalar
 * routine of the generic kernel / state hand-over, out_pair the packed two-at-a-time routine of the fast kernels. */
int sdrfm_debug_discriminate(int device, const float* yr, const float* yi, const float* pr, const float* pi,
                             float* out_scalar, float* out_pair, uint32_t n) {
  if (!yr || !yi || !pr || !pi || !out_scalar || !out_pair) return SDRFM_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SDRFM_NO_DEVICE;
  HIP_TRY(hipSetDevice(device), SDRFM_NO_DEVICE);
  float* d = nullptr;
  const size_t bytes = sizeof(float) * n;
  HIP_TRY(hipMalloc(&d, 6 * bytes + 64), SDRFM_ENOMEM);
  const float* in[4] = {yr, yi, pr, pi};
  for (int k = 0; k < 4; ++k)
    if (hipMemcpy(d + (size_t)k * n, in[k], bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return SDRFM_FAIL; }
  hipLaunchKernelGGL(k_debug_discriminate, dim3((n + 255) / 256), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n,
                     d + 4 * (size_t)n, d + 5 * (size_t)n, (int)n);
  int rc = SDRFM_OK;
  if (hipMemcpy(out_scalar, d + 4 * (size_t)n, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = SDRFM_FAIL;
  if (hipMemcpy(out_pair, d + 5 * (size_t)n, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = SDRFM_FAIL;
  (void)hipFree(d);
  return rc;
}

/* Test hook: design Q's conditioning guard on this handle — its two thresholds and how many lanes (pairs of discriminator outputs) the
 * repair path has recomputed, in how many passes, since create.  SDRFM_NOT_SUPPORTED when the handle has no matrix-pipe kernel. */
// Test hook (include/sdrfm_dev.h): which streams the bit-exact kernels serve.  mask != nullptr sets it (mask[s] != 0: stream s goes to the bit-exact kernels from
// the next call on and stays there; 0: design Q, until the statistics say otherwise); *n_noisy / noisy_out[s] report the assignment in force after the call.
int sdrfm_debug_route(sdrfm_t* h, const uint8_t* mask, uint32_t* n_noisy, uint8_t* noisy_out) {
  if (!h) return SDRFM_EINVAL;
  if (!h->d_qA || !h->rt_noisy) return SDRFM_NOT_SUPPORTED;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  const uint32_t ns = h->cfg.n_streams;
  if (mask) {
    h->rt_next_retry = ~0ull;
    for (uint32_t s = 0; s < ns; ++s) {
      const uint8_t want = mask[s] ? 1 : 0;
      if (h->rt_noisy[s] != want) h->rt_dirty = true;
      h->rt_noisy[s] = want;
      h->rt_retry_at[s] = ~0ull;
    }
    if (h->rt_dirty) {
      const int arc = route_apply(h);
      if (arc != SDRFM_OK) return arc;
    }
  }                                                              // (no mask: the assignment as the last call left it — statistics are taken in by calls only, at fixed ones)
  if (n_noisy) *n_noisy = h->rt_n_noisy;
  if (noisy_out) memcpy(noisy_out, h->rt_noisy, ns);
  return SDRFM_OK;
}

int sdrfm_debug_q_guard(sdrfm_t* h, float* guard_r, float* guard_a, unsigned long long* lanes, unsigned long long* passes) {
  if (!h) return SDRFM_EINVAL;
  if (!h->d_qA) return SDRFM_NOT_SUPPORTED;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  { const int jrc = join_overlap(h); if (jrc != SDRFM_OK) return jrc; }
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  unsigned int st[2] = {0, 0};
  HIP_TRY(hipMemcpy(st, h->d_qstat, sizeof(st), hipMemcpyDeviceToHost), SDRFM_FAIL);
  if (guard_r) *guard_r = h->q_guard_r;
  if (guard_a) *guard_a = h->q_guard_a;
  if (lanes) *lanes = st[0];
  if (passes) *passes = st[1];
  return SDRFM_OK;
}

#ifdef SDRFM_DEV
/* Profiling aid: raw dump of the 560 debug words (slots 520+4x.. = per-XCC min/max wave start, min/max wave end in 100 MHz ticks). */
int sdrfm_debug_raw(sdrfm_t* h, unsigned long long* out512) {
  if (!h || !out512) return SDRFM_EINVAL;
  if (!h->d_dbg) return SDRFM_NOT_SUPPORTED;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  HIP_TRY(hipMemcpy(out512, h->d_dbg, 560 * sizeof(unsigned long long), hipMemcpyDeviceToHost), SDRFM_FAIL);
  return SDRFM_OK;
}

#endif

#ifdef SDRFM_DEV
/* Development library only (include/sdrfm_dev.h): copy of the first n debug words (design S: 16 time stamps per wave of the
 * launch tagged by the 16th call after create / after the last read). */
int sdrfm_dev_read_debug(sdrfm_t* h, unsigned long long* out, uint32_t n) {
  if (!h || !out || !h->d_dbg || !h->stream_profile || n > 32u * 16384u) return SDRFM_EINVAL;
  HIP_TRY(hipSetDevice(h->device), SDRFM_FAIL);
  HIP_TRY(hipStreamSynchronize(h->stream), SDRFM_FAIL);
  HIP_TRY(hipMemcpy(out, h->d_dbg, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost), SDRFM_FAIL);
  h->dbg_launches = 0;
  return SDRFM_OK;
}
#endif

/* Host evaluation of the device's atan2 / discriminator arithmetic (same header, same rounding) so that its accuracy
 * can be unit-tested without a GPU.  Not used by any compute path. */
float sdrfm_host_atan2f(float y, float x) { return (x == 0.0f && y == 0.0f) ? 0.0f : sdrfm_atan2f(y, x); }
float sdrfm_host_discriminate(float yr, float yi, float pr, float pi) { return sdrfm_discriminate(yr, yi, pr, pi); }

}  // extern "C"
