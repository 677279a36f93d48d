/*
 * sdrfm_sink_chain.h — the PCM sink's chain INSIDE the demodulator's own launch (round 6).  Internal to the library: the
 * interface between sdrfm_sink.hip (which owns the sink's state), sdrfm_q.hip (whose kernels run the chain) and sdrfm.hip (the C-ABI call
 * sdrfm_process_batch_pcm).
 *
 * Why inside the launch: a consumer on a third queue behind the overlapped calls' two costs the loop 9 us per call whatever the sink's own duration
 * (profiles/r06_sink.txt: while a queue holds a wait on their kernels the two demodulator queues stop running side by side), and a consumer on the call's own queue
 * serialises that queue.  Inside the launch there is no queue to wait on.
 *
 * How.  The de-emphasis y[n] = (1 - alpha) y[n-1] + alpha x[n] forgets: with d = 1 - alpha = 0.7575 (75 us at 48 kHz), d^64 = 1.9e-8 — below half an fp32 ulp.
 * Every RUN of design Q (one wave: ~400 consecutive audio outputs of one stream, parked in the wave's LDS before they are stored) therefore sinks its OWN
 * outputs where they lie, as a blocked scan started from state 0:
 *   1. lane t walks its chunk of 8 samples from state 0 (lane 0: from the run's state so far) -> the chunk's own contribution to its last sample;
 *   2. carries s[t] = d^8 s[t-1] + e[t] over the 64 lanes (six shuffle steps, the powers squared on the way);
 *   3. lane t walks its chunk again from its true carry-in with exactly the host routine's operations (sdrfm_pcm_deemph_s16, csrc/pcm_sink.c), packs, stores.
 * What a run cannot know is the state its predecessor — the stream's previous run, or the previous CALL's last run — ends with; that state only reaches the run's
 * first 64 outputs (beyond them it is below rounding), and linearly: y[k] = y_local[k] + d^(k+1) * carry.  So a run keeps y_local[0..64) aside, PUBLISHES its own
 * end state (which, 64 outputs in, no longer depends on the carry either) as one 64-bit word {call tag, state}, reads its predecessor's word (a read-modify-write
 * atomic: performed at the device's coherence point whatever the L2s hold), and finishes its first 64 outputs with one fused multiply-add each.  (The end state
 * is taken from the scan of the run's last flush, before the second walk, and the predecessor's word is asked for in the same breath: both round trips run under
 * the walk and the stores — 0.25 us per call less than publishing behind them.)  No wave waits for
 * more than its neighbour's last step; nothing is counted, nothing is read back from memory, no cache is written back or invalidated.
 * (First version, kept in the history: the stream's LAST wave re-read the whole row and ran the chain as a tail — 30 us per call against the demodulator's 22:
 * four dependent round trips through a saturated memory system at the very end of the launch, which the next launch on that queue waits for.)
 * Against the exact chain: the carry-in of a chunk is re-associated (1e-7 relative) and d^64 * carry is dropped: PCM within 1 LSB (different only where y * gain
 * sits on a rounding boundary), carried state within 1e-6 relative — the stand-alone blocked scan's tolerance (tests/test_pcm_sink_gpu.py).
 *
 * Order between calls: the word {calls applied to stream s, state} is published by the stream's last run of call c with tag c + 1 (in slot tag % 8 of sg: a
 * word is never overwritten while a call that may still read it is incomplete); run 0 of call c + 1 waits (an s_sleep loop of one lane) for that tag.  Call c was launched before call c + 1 and none of its waves waits on anything but its own lower-numbered
 * neighbour, so it always gets there.  The stand-alone sink kernels keep sg up to date too (stream order), so the two styles can follow one another.
 */
#ifndef SDRFM_SINK_CHAIN_H
#define SDRFM_SINK_CHAIN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

struct sdrfm_pcm_sink;

#define SDRFM_CHAIN_FIX 64u        /* outputs of a run its predecessor's state still reaches */
#define SDRFM_CHAIN_CH 8u          /* samples per lane of a run's scan: 512 = the most design Q parks before it stores */
#define SDRFM_CHAIN_SETS 4u        /* sets of per-run words: calls c and c + 1 may be in flight together, c + 2 is ordered behind c */
#define SDRFM_CHAIN_SG_SLOTS 8u    /* slots of the per-stream word: tag t lives in slot t % 8.  Call t reads tag t; while call t is incomplete only calls t, t + 1 and
                                     t + 3 can publish (t + 2 waits for t on its queue, t + 5 for t + 3, which needs t + 2's state): tags t + 1, t + 2, t + 4 — never t + 8 */
#define SDRFM_CHAIN_MIN_ALPHA 0.231f   /* (1 - alpha)^64 <= 5e-8: below it a run's end state would still depend on its predecessor's */

struct SdrfmSinkChain {
  int16_t* pcm;                   // [n_streams][pcm_stride] interleaved (L, R) int16, rows 4-byte aligned
  size_t pcm_stride;              // int16 elements, even
  unsigned long long* sg;         // [SDRFM_CHAIN_SG_SLOTS][n_streams] {tag << 32 | bits of y[n-1]}: the stream's state behind `tag` calls, in slot tag % 8
  uint32_t n_streams;             // of the handle (the slots' row length)
  unsigned long long* runstate;   // [grid] this call's set of per-run words {call + 1 << 32 | bits of the run's end state} (sdrfm.hip owns it)
  const float* dpow;              // [SDRFM_CHAIN_FIX] (1 - alpha)^(k + 1)
  uint32_t* err;                  // one word: set when a run gave up waiting for its predecessor's word (a protocol error: the sink reports it)
  uint32_t call;                  // this call's number (mod 2^32)
  float alpha, gain, pc;          // pc = (1 - alpha)^SDRFM_CHAIN_CH
  float w[SDRFM_CHAIN_CH];         // w[q] = alpha (1 - alpha)^(SDRFM_CHAIN_CH - 1 - q): what sample q of a chunk adds to the chunk's last output
  float dinv[SDRFM_CHAIN_CH];      // dinv[k] = (1 - alpha)^-k: undoes the decay over the k zeros behind a flush's last output in its lane's chunk
};

// ---- host side (sdrfm_sink.hip) ---------------------------------------------------------------------------------------------------------------------------
// The parameters for the sink's NEXT call (pcm / pcm_stride / runstate left for the caller to fill).  0: the sink does not fit the handle (other device, other
// stream count); 1: it fits, but its time constant is too long for runs to be sunk independently (alpha < SDRFM_CHAIN_MIN_ALPHA): stand-alone kernel only; 2: fits.
int sdrfm_sink_chain_params(sdrfm_pcm_sink* k, int device, uint32_t n_streams, SdrfmSinkChain* out);
// The launch that carried `out` is in the queue: the sink's call counter moves on.
void sdrfm_sink_chain_issued(sdrfm_pcm_sink* k);
// The stand-alone blocked scan on `stream` (device buffers), as one call of the sink: what a call that no kernel with the chain served is followed by.
// The stand-alone blocked scan over the streams list[0 .. n_list) only, as PART of the sink's call `c.call` (the chain inside a launch serves the other streams): on
// `stream`, behind the launch that wrote those streams' audio; it takes each stream's state by the chain's own protocol (the tagged word, a bounded wait on the device)
// and publishes the next.  Does not move the sink's call counter (sdrfm_sink_chain_issued does, once per call).
// done != nullptr: the event is signalled by this kernel's own completion (a stop event).
int sdrfm_sink_launch_list_on(sdrfm_pcm_sink* k, const SdrfmSinkChain& c, const uint32_t* list_dev, uint32_t n_list, const float* audio, size_t audio_stride, uint32_t n,
                              int16_t* pcm, size_t pcm_stride, hipStream_t stream, hipEvent_t done);
int sdrfm_sink_launch_on(sdrfm_pcm_sink* k, const float* audio, size_t audio_stride, uint32_t n, int16_t* pcm, size_t pcm_stride, hipStream_t stream);

#endif
