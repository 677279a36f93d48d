/*
 * sdrfm_q.h — design Q ("matrix-pipe FIR"): launch interface between the C-ABI host code (sdrfm.hip) and the kernel's own
 * translation unit (sdrfm_q.hip).  Internal to the library; the drop-in boundary is include/sdrfm.h.
 */
#ifndef SDRFM_Q_H
#define SDRFM_Q_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdrfm_q_host.h"
#include "sdrfm_sink_chain.h"

struct SdrfmQParams {
  const uint8_t* iq;            // [n_streams][iq_stride] interleaved u8 I/Q (device), rows 16-byte aligned
  size_t iq_stride;             // bytes
  const uint8_t* iq_prev;       // the previous call's buffer (rows of N_prev samples, iq_prev_stride bytes apart, 16-byte aligned, N_prev >= 1280
  size_t iq_prev_stride;        //   and 2 N_prev % 16 == 0) or nullptr: with it the stream's first run needs no carried state
  uint32_t N_prev;
  float* audio;                 // [n_streams][audio_stride]
  size_t audio_stride;          // floats
  const float2* yprev_in;       // streaming state, as the other kernels keep it (sdrfm.hip: CallParams)
  float2* yprev_out;
  const float* hist_d_in;       // [n_streams][31]
  float* hist_d_out;
  const uint8_t* hist_b_in;     // [n_streams][T-1] raw I/Q byte pairs
  uint8_t* hist_b_out;
  float2* hist_x_out;           // [n_streams][T-1] the same samples DC-shifted (the generic kernel's history format)
  const int8_t* A;              // operand tables [3][3][64][16]: sparse K-chunk, digit, lane, kept slot (qtaps.c)
  const float* g;               // audio taps g[0..32)
  float q0, q2, cst;            // y = q0 * (S0 + 256 S1) + q2 * S2 + cst
  uint32_t T;                   // channel taps (for the history hand-over only; the arithmetic is in the tables)
  uint32_t N, M, A_out;         // per stream and call: IQ samples, decimated outputs, audio outputs
  uint32_t steps_total;         // ceil(M / 128)
  uint32_t runs;                // runs (waves) per stream
  uint32_t n_streams;
  uint32_t prio_by_age;         // 1: from the middle of its run a wave's issue priority is its age rank in the SIMD (calls one after the other)
  // ---- the conditioning guard and its repair path (DESIGN.md 4.Q "guard"): where the phase of y[m] conj(y[m-1]) is ill-conditioned the d's are
  // recomputed with the definition's own fmaf chain from the raw bytes, so that they equal the bit-exact kernels' d's exactly
  const float* hpad;            // [SDRFM_Q_TP] channel taps h[k], zero for k >= T
  const uint8_t* hist_q_in;     // [n_streams][2 SDRFM_Q_TP] raw I/Q bytes of the last SDRFM_Q_TP samples before the call (16-byte aligned rows)
  uint8_t* hist_q_out;
  float guard_r;                // a lane is repaired when max(|re|, |im|) of one of its three y's is below this ...
  float guard_a;                // ... or one of its two |d|'s is above this (the branch cut); guard_r = 0 and guard_a = 4 switch the guard off
  uint32_t yprev_exact;         // yprev_in holds the definition's y[-1] (reset, or the previous call ran on a bit-exact kernel): hist_q_in[0] is not valid then
  unsigned int* n_repaired;     // statistics (device, two words: repaired lanes, repair passes) or nullptr
  unsigned int* stream_pass;    // [n_streams of the handle] or nullptr: repair passes per stream, added up by the waves that ran any (the host routes a stream
                                // whose windows are mostly repair work to the bit-exact kernels: sdrfm.hip)
  const uint32_t* slist;        // nullptr: the launch serves streams 0 .. n_streams-1; else its i-th stream is stream slist[i] of the handle (n_streams = the list's length)
  unsigned long long* dbg;      // development build: per-wave time stamps (else nullptr)
};

// geometry the host needs
#define SDRFM_Q_D 10u            /* FIR decimation of the BASELINE front end (2.4 MS/s); instances exist for D = 8 (2.048 MS/s) and 16 (3.2 MS/s) too */
#define SDRFM_Q_TA 32u           /* audio taps (every instance) */
#define SDRFM_Q_DA 5u            /* audio decimation of the BASELINE front end */
#define SDRFM_Q_STEP_OUT 128u    /* decimated outputs per wave step (16 columns x 8 outputs) */
#define SDRFM_Q_TP 64u           /* the repair path's chain length: channel taps padded with zeros to this many (design Q serves T <= 64) */

// is there an instance for FIR decimation d and audio decimation da (with SDRFM_Q_TA audio taps)?  its ring size in KiB
bool sdrfm_q_geometry_ok(uint32_t d, uint32_t da);
uint32_t sdrfm_q_default_nslot(uint32_t d);
// LDS bytes of one wave for a ring of `nslot` KiB
uint32_t sdrfm_q_lds_bytes(uint32_t nslot, uint32_t d, uint32_t da);
// one-wave workgroups of this variant the runtime says a CU can hold at once (0 = unknown)
int sdrfm_q_blocks_per_cu(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da);
// Enqueue one call: grid = n_streams * runs one-wave workgroups.  first_chunk = 0 or 1 (from sdrfm_q_build).
// Returns hipSuccess or the launch error.
// done != nullptr: the event is signalled by the kernel's own completion (hipExtLaunchKernelGGL's stop event: no marker packet in the queue).
hipError_t sdrfm_q_launch(const SdrfmQParams& p, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, hipStream_t stream, hipEvent_t done = nullptr);
// the same launch with the PCM sink's chain as its tail (sdrfm_sink_chain.h): p must serve every stream of the handle that owns the sink's counters
bool sdrfm_q_has_pcm_chain(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da);
hipError_t sdrfm_q_launch_pcm(const SdrfmQParams& p, const SdrfmSinkChain& t, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, hipStream_t stream,
                              hipEvent_t done = nullptr);
const char* sdrfm_q_kernel_symbol(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da);
// ---- one launch for a mixed batch (sdrfm_q.hip: k_mix): b_blocks design-B workgroups (tile R = b_R; b as k_fastb takes it, fold_state = 1) over the
// streams of b.slist, then q.n_streams * q.runs design-Q workgroups.  An instance exists for every shape both designs have one for (LDS bytes of one workgroup; 0 = none).
struct CallParams;   // sdrfm_b.h
uint32_t sdrfm_q_mix_lds(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, uint32_t T, uint32_t b_R);
int sdrfm_q_mix_blocks_per_cu(uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, uint32_t T, uint32_t b_R);
hipError_t sdrfm_q_launch_mix(const SdrfmQParams& q, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, const CallParams& b, uint32_t b_blocks,
                              uint32_t b_R, hipStream_t stream, hipEvent_t done);
// the same launch with the PCM sink's chain in its design-Q workgroups (sdrfm_sink_chain.h): the clean streams' PCM; q.slist's streams only
hipError_t sdrfm_q_launch_mix_pcm(const SdrfmQParams& q, const SdrfmSinkChain& t, uint32_t first_chunk, uint32_t nslot, uint32_t d, uint32_t da, const CallParams& b,
                                  uint32_t b_blocks, uint32_t b_R, hipStream_t stream, hipEvent_t done);
// yprev[s] = the definition's y[-1] of stream s from the SDRFM_Q_TP raw samples in hist_q (a bit-exact kernel takes over from design Q)
// (of the streams list[0 .. n_streams), or of streams 0 .. n_streams-1 when list is nullptr)
hipError_t sdrfm_q_fix_yprev(const uint8_t* hist_q, const float* hpad, float2* yprev, uint32_t n_streams, const uint32_t* list, hipStream_t stream);

#endif
