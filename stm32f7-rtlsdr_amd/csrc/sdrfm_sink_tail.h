/*
 * sdrfm_sink_tail.h — the PCM sink's chain as the TAIL of the demodulator's own launch (round 6).  Internal to the
 * library: the interface between sdrfm_sink.hip (which owns the sink's state), sdrfm_q.hip (whose kernels run the tail) and sdrfm.hip (the C-ABI call
 * sdrfm_process_batch_pcm).
 *
 * Why a tail and not a kernel of its own: a consumer on a third queue behind the overlapped calls' two costs the loop 9 us per call whatever the sink's own
 * duration (profiles/r06_sink.txt: while a queue holds a wait on their kernels the two demodulator queues stop running side by side), and a consumer on the
 * call's own queue serialises that queue.  Inside the launch there is no queue to wait on: every wave of a stream counts itself done behind its audio stores
 * (release, device scope); the wave that counts last walks the stream's whole row (it is in L2 / the Infinity Cache: the launch has just written it) through the
 * de-emphasis chain and stores the interleaved int16 pairs.  One wave per stream, segments of 64 x 39 samples through the wave's own LDS (free by then; memory
 * is touched 1 KiB per instruction: a first version with 76-sample chunks read and written in place, lanes 304 bytes apart, spent 35 us per call in its stores):
 *   1. lane t walks its chunk from state 0 (lane 0: from the carried state) -> the chunk's own contribution to its last sample;
 *   2. carries s[t] = (1 - alpha)^39 s[t-1] + e[t] over the 64 lanes (six shuffle steps, the powers squared on the way);
 *   3. lane t walks its chunk AGAIN from its true carry-in with exactly the host routine's operations (sdrfm_pcm_deemph_s16, csrc/pcm_sink.c), packs, stores.
 * The same construction as the stand-alone blocked scan (sdrfm_sink.hip: k_pcm_sink_scan) with another chunk length: within 1 LSB of the exact chain, the carried
 * state within 2.5e-7 relative (tests/test_pcm_sink_gpu.py).
 *
 * Order between calls.  The chain of call c continues the state call c - 1 left, and two overlapped calls run concurrently.  gen[s] counts the calls whose chain
 * has been applied to stream s; the tail wave of call c waits (s_sleep loop of one lane) until gen[s] == c, then publishes state and gen[s] = c + 1 (release).
 * Call c - 1 was launched before call c and none of its waves waits on anything, so it always gets there; the wait is over before it starts unless call c
 * overtakes a call launched a whole call earlier.  Stand-alone sink launches keep gen up to date too (stream order), so the two styles can follow one another.
 */
#ifndef SDRFM_SINK_TAIL_H
#define SDRFM_SINK_TAIL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

struct sdrfm_pcm_sink;

#define SDRFM_TAIL_C 39u          /* samples per lane and segment of the tail's wave (odd): a segment = 64 x 39 = 2496 samples through 9984 bytes of the wave's LDS; BASELINE's 4800 per call: two */
#define SDRFM_TAIL_LDS (64u * SDRFM_TAIL_C * 4u)
#define SDRFM_TAIL_SETS 4u        /* sets of "waves done" counters: calls c and c + 1 may be in flight together, c + 2 is ordered behind c */

struct SdrfmSinkTail {
  int16_t* pcm;                   // [n_streams][pcm_stride] interleaved (L, R) int16, rows 4-byte aligned
  size_t pcm_stride;              // int16 elements, even
  float* state;                   // [n_streams] y[n-1] of the de-emphasis
  uint32_t* gen;                  // [n_streams] calls applied to the stream so far (mod 2^32)
  uint32_t* cnt;                  // [SDRFM_TAIL_SETS][n_streams] 64-bit words, 8-byte aligned: waves of this call that are done and where they ran (zero between calls)
  uint32_t call;                  // this call's number (mod 2^32)
  uint32_t n_streams;             // of the handle (the counters' row length)
  float alpha, gain, pc;          // pc = (1 - alpha)^SDRFM_TAIL_C
};

// ---- host side (sdrfm_sink.hip) ---------------------------------------------------------------------------------------------------------------------------
// The tail's parameters for the sink's NEXT call (pcm / pcm_stride left for the caller to fill); false when the sink does not fit (other device, other stream count).
bool sdrfm_sink_tail_params(sdrfm_pcm_sink* k, int device, uint32_t n_streams, SdrfmSinkTail* out);
// The launch that carried `out` is in the queue: the sink's call counter moves on.
void sdrfm_sink_tail_issued(sdrfm_pcm_sink* k);
// The stand-alone blocked scan on `stream` (device buffers), as one call of the sink: what a call that no kernel with a tail served is followed by.
int sdrfm_sink_launch_on(sdrfm_pcm_sink* k, const float* audio, size_t audio_stride, uint32_t n, int16_t* pcm, size_t pcm_stride, hipStream_t stream);

#ifdef __HIPCC__
// ---- device side: the tail of ONE wave (64 lanes) -----------------------------------------------------------------------------------------------------------
// Called by every one-wave workgroup of the launch when its own work for `stream` is done and stored (THROUGH the L2: device-scope stores); `parts` = workgroups
// of the launch that serve the stream (< 65536); lds = the workgroup's own LDS, free by now, at least SDRFM_TAIL_LDS bytes.
//
// Coherence between the XCDs' L2s.  The audio row is written by `parts` waves and read by one.  The launch maps a stream's workgroups to ONE XCD where it can
// (the caller's business: workgroups go round the XCDs in turn), and every wave adds its XCC_ID — and its square — to the stream's counter: when the sums say
// that all of them ran on the reader's own XCD, the row is coherent in that L2 as it is (write-through stores update it) and nothing is invalidated; if not — a
// grid that does not divide by the number of XCDs, a dispatcher that deals differently — the reader invalidates its L2's foreign lines first (buffer_inv sc1:
// correct always, measured 13 us per call when every stream's reader does it).  State and gen are read by read-modify-write atomics, which are performed at
// the device's coherence point whatever the caches hold (a device-scope LOAD of gen was seen to spin on a stale line for microseconds).
#ifndef SDRFM_TAIL_EXP
#define SDRFM_TAIL_EXP 0
#endif
__device__ __forceinline__ void sdrfm_sink_tail(const SdrfmSinkTail& t, const float* audio, size_t audio_stride, uint32_t n, uint32_t stream, uint32_t parts,
                                                float* lds) {
  const uint32_t lane = threadIdx.x;
  unsigned long long* const c = reinterpret_cast<unsigned long long*>(t.cnt) + (size_t)(t.call % SDRFM_TAIL_SETS) * t.n_streams + stream;
  const unsigned long long xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;       // HW_REG_XCC_ID
  const unsigned long long mine = 1ull | (xcc << 16) | ((xcc * xcc) << 36);             // waves done [0, 16) | sum of XCC_ID [16, 36) | sum of its square [36, 60)
  // This wave's audio is visible to the device before it counts itself done: stored through the L2, so "acknowledged" is "visible" (a release fence here is a
  // buffer_wbl2 per wave: measured 340 us per call with 3072 waves).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(c, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long tot = (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(old >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)old)) + mine;
  if ((uint32_t)(tot & 0xffffu) != parts) return;
  // ---- the stream's last wave -------------------------------------------------------------------------------------------------------------------------
  // (first in line on its SIMD from here on: the launch ends when the last tail does, and the tail's two dependent walks take four to six times as long when
  // they share the issue slots evenly with the next call's waves; what the others lose is the tail's own 2 % of the work)
  if (!(SDRFM_TAIL_EXP & 256)) __builtin_amdgcn_s_setprio(3);
  const bool one_xcd = ((tot >> 16) & 0xfffffull) == xcc * parts && (tot >> 36) == xcc * xcc * parts;
  unsigned y0b = 0;
  if (lane == 0) {
    __hip_atomic_store(c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (the set's next user is ordered behind this launch)
    unsigned zero = 0u;
    asm volatile("" : "+v"(zero));                               // (opaque: "add 0" is a read-modify-write the compiler would turn back into a load)
    while (__hip_atomic_fetch_add(t.gen + stream, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t.call) __builtin_amdgcn_s_sleep(8);
    y0b = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(t.state + stream), zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (((SDRFM_TAIL_EXP & 128) || !one_xcd) && !(SDRFM_TAIL_EXP & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (SDRFM_TAIL_EXP & 4) { if (lane == 0) __hip_atomic_store(t.gen + stream, t.call + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  float y0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)y0b));
  const float* const row = audio + (size_t)stream * audio_stride;
  unsigned* const out = reinterpret_cast<unsigned*>(t.pcm + (size_t)stream * t.pcm_stride);
  unsigned* const ldw = reinterpret_cast<unsigned*>(lds);
  constexpr uint32_t C = SDRFM_TAIL_C, SEG = 64u * C;
  static_assert(C % 2 == 1 && SEG % 4 == 0, "chunks an odd number of words apart: conflict-free LDS accesses; segments of whole 16-byte groups");
  const bool vin = (reinterpret_cast<uintptr_t>(row) & 15u) == 0, vout = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;   // 16-byte rows: four samples per instruction
  for (uint32_t base = 0; base < n; base += SEG) {
    const uint32_t m = (n - base < SEG) ? n - base : SEG;       // samples of this segment
    // the segment into LDS as it lies: every instruction moves 1 KiB (256 B) of consecutive memory.  (A group of four is loaded whole when it holds a sample of
    // the segment: the words past the end lie in the same 16 bytes as that sample — readable — and are never used; stores are exact.)
    if (vin) {
#pragma unroll
      for (uint32_t i = 0; i < (SEG / 4 + 63) / 64; ++i) {
        const uint32_t e = 4u * (lane + 64u * i);
        if (SDRFM_TAIL_EXP & 32) { if (e < m) *reinterpret_cast<float4*>(lds + e) = make_float4(0.1f, 0.2f, 0.3f, 0.4f); } else
        if (e < m) *reinterpret_cast<float4*>(lds + e) = *reinterpret_cast<const float4*>(row + base + e);
      }
    } else {
#pragma unroll
      for (uint32_t i = 0; i < C; ++i) {
        const uint32_t e = lane + 64u * i;
        if (e < m) lds[e] = row[base + e];
      }
    }
    __builtin_amdgcn_wave_barrier();                            // (one wave: program order; this pins it for the compiler)
    const uint32_t i0 = lane * C < m ? lane * C : m, cnt = (m - i0 < C) ? m - i0 : C;
    float xr[C];
#pragma unroll
    for (uint32_t q = 0; q < C; ++q) xr[q] = q < cnt ? lds[i0 + q] : 0.0f;
    // 1. the chunk's own contribution to its last sample (lane 0 starts from the real state: its chain is the exact one already)
    float y = lane == 0 ? y0 : 0.0f;
#pragma unroll
    for (uint32_t q = 0; q < C; ++q) {
      const float yn = __builtin_fmaf(t.alpha, xr[q] - y, y);
      y = q < cnt ? yn : y;
    }
    // 2. s[t] = pc s[t-1] + e[t] (only the last non-empty chunk may be short, and nothing follows it)
    float sc = y, pw = t.pc;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
      const float o = __shfl_up(sc, d, 64);
      const float sn = __builtin_fmaf(pw, o, sc);
      sc = lane >= d ? sn : sc;
      pw *= pw;
    }
    // 3. the exact form's chain from the true carry-in; the packed words back into the segment's place
    y = __shfl_up(sc, 1u, 64);
    if (lane == 0) y = y0;
#pragma unroll
    for (uint32_t q = 0; q < C; ++q) {
      const float yn = __builtin_fmaf(t.alpha, xr[q] - y, y);
      y = q < cnt ? yn : y;
      float v = yn * t.gain;
      v = __builtin_fminf(__builtin_fmaxf(v, -32768.0f), 32767.0f);         // (v is never a NaN for finite audio; the host routine's two compares give the same value)
      const unsigned sw = (unsigned)(int)__builtin_rintf(v) & 0xffffu;
      if (q < cnt) ldw[i0 + q] = sw | (sw << 16);
    }
    y0 = __shfl(y, (int)((m - 1u) / C), 64);                     // the lane that holds the segment's last sample: the state behind it
    __builtin_amdgcn_wave_barrier();
    if (vout) {
#pragma unroll
      for (uint32_t i = 0; i < (SEG / 4 + 63) / 64; ++i) {
        const uint32_t e = 4u * (lane + 64u * i);
        if (SDRFM_TAIL_EXP & 64) { if (e + 4u <= m && ldw[e] == 0x12345u) out[base + e] = 1; } else
        if (e + 4u <= m) *reinterpret_cast<uint4*>(out + base + e) = *reinterpret_cast<const uint4*>(ldw + e);
        else if (e < m)
          for (uint32_t j = 0; j < m - e; ++j) out[base + e + j] = ldw[e + j];
      }
    } else {
#pragma unroll
      for (uint32_t i = 0; i < C; ++i) {
        const uint32_t e = lane + 64u * i;
        if (e < m) out[base + e] = ldw[e];
      }
    }
    __builtin_amdgcn_wave_barrier();                            // (the next segment overwrites the LDS words just read)
  }
  if (lane == 0) {                                              // (state through the L2 too, acknowledged before gen moves on)
    __hip_atomic_store(reinterpret_cast<unsigned*>(t.state + stream), __builtin_bit_cast(unsigned, y0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(t.gen + stream, t.call + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#endif

#endif
