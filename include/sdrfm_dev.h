/*
 * sdrfm_dev.h — test hooks and development aids of libsdrfm.  NOT part of the drop-in boundary (that is include/sdrfm.h): nothing a
 * front end binds lives here.  Only tests/ and tools/ include or bind these.
 *
 *   exported by the product library libsdrfm.so (they compute nothing on behalf of a caller; tests use them to check the
 *   arithmetic the kernels are built from):
 *     sdrfm_host_atan2f, sdrfm_host_discriminate   HOST evaluation of the device's K3 arithmetic (same source, same rounding)
 *     sdrfm_debug_discriminate                     K3 evaluated ON THE DEVICE for n operand sets, both code forms
 *     sdrfm_q_build                                design Q: the channel taps as i8 matrix-pipe operand tables (csrc/qtaps.c)
 *     sdrfm_q_guard                                design Q: the conditioning guard's thresholds for a tap set (csrc/qtaps.c; no GPU)
 *     sdrfm_debug_q_guard                          design Q: a handle's guard thresholds and how often its repair path ran
 *     sdrfm_debug_read_ceiling                     what a read-only stream with design Q's access pattern gets out of the memory system
 *     sdrfm_debug_route                            design Q: which streams of a handle the bit-exact kernels serve (read, or set for a test)
 *   exported by the development library libsdrfm_dev.so only (built with -DSDRFM_DEV):
 *     sdrfm_debug_phase_cycles, sdrfm_debug_raw    instrumented kernels' counters (SDRFM_PHASE_PROFILE=1 at create)
 *     sdrfm_dev_read_debug                         per-wave time stamps of design S (SDRFM_STREAM_PROFILE=1)
 */
#ifndef SDRFM_DEV_H
#define SDRFM_DEV_H

#include "sdrfm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HOST evaluation of the exact arithmetic the device uses for stage K3 (same source, same rounding), so that its accuracy against
 * libm can be checked without a GPU.  No compute path calls these. */
float sdrfm_host_atan2f(float y, float x);
float sdrfm_host_discriminate(float yr, float yi, float pr, float pi);

/* Stage K3 evaluated ON THE DEVICE for n operand sets (host arrays): out_scalar = the scalar routine of the generic kernel / state
 * hand-over, out_pair = the packed two-at-a-time routine of the specialised kernels. */
int sdrfm_debug_discriminate(int device, const float* yr, const float* yi, const float* pr, const float* pi,
                             float* out_scalar, float* out_pair, uint32_t n);

/* Design Q (csrc/qtaps.c): h[0..T) -> A[D/2][3][64][16] (K-chunk, digit, lane, byte) i8 operand tables, the fp32 scale q with
 * y = q (S0 + 256 S1 + 65536 S2) + cst, cst = 0.5 sum(h), and the first K-chunk holding a non-zero tap.  0 on success. */
int sdrfm_q_build(const float* h, uint32_t T, uint32_t D, int8_t* A, float* q, float* cst, uint32_t* first_chunk);

/* Design Q (csrc/qtaps.c): the conditioning guard's thresholds for channel taps h[0..T) and audio taps g[0..Ta).  0 on success. */
int sdrfm_q_guard(const float* h, uint32_t T, const float* g, uint32_t Ta, float* guard_r, float* guard_a);
int sdrfm_q_guard2(const float* h, uint32_t T, const float* g, uint32_t Ta, int worst_case, float* guard_r, float* guard_a);   /* worst_case = 1: SDRFM_CFG_GUARD_WORST_CASE's radius */

/* Design Q's conditioning guard on this handle (csrc/sdrfm_q.hip): a lane is repaired — its two discriminator outputs recomputed with the
 * definition's own fmaf chain — when one of its y's has max(|re|, |im|) < *guard_r or one of its |d|'s exceeds *guard_a; *lanes / *passes =
 * lanes repaired / repair passes run since create (32-bit counters on the device).  Synchronises the handle.  SDRFM_NOT_SUPPORTED when the
 * handle has no matrix-pipe kernel (SDRFM_CFG_BIT_EXACT, other geometries). */
int sdrfm_debug_q_guard(sdrfm_t* h, float* guard_r, float* guard_a, unsigned long long* lanes, unsigned long long* passes);

/* Per-stream routing (csrc/sdrfm.hip: the handle's comment): a stream whose windows of design-Q calls are mostly repair work is served by the bit-exact
 * kernels for a while — design-B workgroups inside design Q's launch over the other streams, or a launch ahead of it.  mask != NULL ([n_streams] bytes) SETS the assignment for a
 * test (non-zero: the bit-exact kernels, for good; zero: design Q until the statistics say otherwise); mask == NULL takes in whatever statistics have arrived.
 * *n_noisy (may be NULL) = streams the bit-exact kernels serve from the next call on; noisy_out (may be NULL, [n_streams] bytes) = which.
 * SDRFM_NOT_SUPPORTED when the handle has no matrix-pipe kernel. */
int sdrfm_debug_route(sdrfm_t* h, const uint8_t* mask, uint32_t* n_noisy, uint8_t* noisy_out);

/* The measured read ceiling bench.py prints beside the 8 TB/s specification: a read-only LDS-DMA stream with design Q's access
 * pattern (one-wave workgroups, 12 per CU, 5 KiB in flight each, non-temporal) over nbufs device buffers of bytes_each bytes, `passes` passes
 * taken in turn over the buffers, timed with HIP events on a stream of its own (synchronous).  *gbytes_per_s = bytes read / elapsed. */
int sdrfm_debug_read_ceiling(int device, const void* const* bufs, uint32_t nbufs, size_t bytes_each, uint32_t passes, double* gbytes_per_s);

/* Development library only: cumulative shader cycles per phase of the instrumented design-B kernel summed over waves (out[0..4] =
 * stage, FIR, discriminator, audio, carry; out[5] = sub-tiles; out[6] = waves), reset on read; raw dump of its 560 debug words;
 * per-wave time stamps of design S. */
int sdrfm_debug_phase_cycles(sdrfm_t* h, unsigned long long* out8);
int sdrfm_debug_raw(sdrfm_t* h, unsigned long long* out560);
int sdrfm_dev_read_debug(sdrfm_t* h, unsigned long long* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* SDRFM_DEV_H */
