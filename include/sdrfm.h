/*
 * sdrfm.h — C-ABI of the MI355X-native IQ -> FM-audio path ("sdrfm").
 *
 * This is the drop-in boundary for the consumer hook that the reference firmware leaves empty:
 *   - the bulk-IN FSM fills `CommItf.buff` / `CommItf.buffSize` with interleaved uint8 I/Q
 *     (Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Inc/usbh_rtlsdr.h:165-173) and then does
 *     nothing in RTLSDR_XFER_COMPLETE (.../RTLSDR/Src/usbh_rtlsdr.c:1094-1097); the application-side
 *     poll of `xferState` is commented out (src/main.c:76-79).
 *   - sdrfm_process() is what either hook calls with (buff, USBH_LL_GetLastXferSize()) — see INTEGRATION.md.
 *
 * Conventions follow the reference's class-driver layer:
 *   - return value 0 == OK, same numeric values as USBH_StatusTypeDef for the first five codes
 *     (Middlewares/ST/STM32_USB_Host_Library/Core/Inc/usbh_def.h:303-311);
 *   - errors are returned, never thrown/aborted; a handle is not thread-safe (single superloop model,
 *     src/main.c:72-80); independent handles may be used concurrently;
 *   - the callee is finished with `iq` when the call returns (the reference re-arms the same buffer
 *     immediately, usbh_rtlsdr.c:1068-1077), except in SDRFM_F_DEVICE_PTRS mode (see below).
 *
 * Arithmetic (build-defined; the reference holds no DSP code — see DESIGN.md "Frozen spec"):
 *   x[n]  = ((float)I_n - 127.5f) + j((float)Q_n - 127.5f)
 *   y[m]  = sum_{k} h[k] * x[(m+1)*D - 1 - k]        x[n<0] = 0, fp32 FMA chain, oldest sample first
 *   d[m]  = atan2f(Im(y[m]*conj(y[m-1])), Re(...))   y[-1] = 0, d = 0 when both parts are exactly 0
 *   a[j]  = sum_{k} g[k] * d[(j+1)*Da - 1 - k]       d[m<0] = 0, fp32 FMA chain, oldest sample first
 * All state (FIR history, y[m-1], d history, decimator phases) persists across calls; chunk lengths need
 * not be multiples of D.
 * Which kernel evaluates this is chosen per call (sdrfm_kernel_name() tells): kernels that are bit-identical to the definition, or — for
 * machine-filling calls of whole audio periods with low-pass taps, unless the handle has SDRFM_CFG_BIT_EXACT — the matrix-pipe kernel
 * ("fast-q").  That one evaluates y exactly in integers from taps rounded to 24-bit fixed point, which is within ~1e-4 (absolute) of the
 * definition's fp32 chain, i.e. within 1e-6 of the definition's AUDIO wherever the phase of y[m] conj(y[m-1]) is well conditioned; where
 * it is not — |y| small (a deep fade: noise-only input gets there, a carrier does not) or d within reach of +-pi (the branch cut) — its
 * conditioning guard recomputes the affected d's with the definition's own chain from the raw bytes, so that they are the bit-identical
 * kernels' d's.  How far "within reach" goes rests on a bound E on |y_fast-q - y_definition|.  By default it is STATISTICAL:
 * E = 1.25 sqrt(T) 127.5 sum|h| 2^-24 (csrc/qtaps.c; 28 standard deviations of what uniform random bytes produce, 9 of a full-scale
 * carrier's).  So "fast-q" within the 1e-5 tolerance (|a - b| <= 1e-5 max(|b|, 1), for audio taps of about unit absolute sum) is by
 * default a MEASURED statement, not a proven one: no violation, worst 9.8e-7, in the device soaks over every input class — uniform random
 * bytes, constant and counter bytes, carriers, and the classes built to probe the bound (strong out-of-band carriers at one to three guard
 * radii, 2 - 8 LSB carriers, periodic byte patterns: tools/q_classes.py, profiles/r06_fuzz_q.txt).  What can be had beyond that, and its price:
 *   SDRFM_CFG_GUARD_WORST_CASE  the radius from the PROVEN worst case of E (every rounding the same way: 6.9 x at 64 taps).  Proves every
 *                               unrepaired d within 5e-6 / max|g| + 1.1e-6 of the definition's; the AUDIO bound that follows for arbitrary bytes is
 *                               sum|g| times that (4.6e-5 for the BASELINE audio taps: all 32 d's of a window at the radius, errors aligned), 1e-5
 *                               when at most one pair of a window is marginal.  Costs a carrier nothing measurable (profiles/r06_guard_worst_case.txt).
 *   SDRFM_CFG_BIT_EXACT         the guarantee: the bit-identical kernels only — 0.42 of the HBM roofline against 0.60 / 0.72 (bench.py: bit_exact_kernel).
 *                               A worst-case radius that PROVED 1e-5 for arbitrary bytes would be 2 sum|g| E_wc / 8e-6 = 180 of 127.5 full scale:
 *                               every output repaired, i.e. this flag at a higher price.
 * A stream whose windows of calls are mostly repair work (noise only) is moved to the bit-identical kernels for a while, per stream
 * (DESIGN.md 4.Q "routing").  Which kernel serves a stream at which call is a function of the bytes and the sequence of calls alone (round 6): a
 * window of 16 calls takes effect at the first call of the window four windows later, never "when noticed" — two runs of one capture give
 * the same bits.  Every choice is within the tolerance.
 *
 * There is NO CPU fallback in this library: every entry point that computes runs hand-written HIP kernels
 * on a gfx950 device and fails with SDRFM_NO_DEVICE when none is usable.
 */
#ifndef SDRFM_H
#define SDRFM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRFM_ABI_VERSION 1u

/* status codes: first five numerically equal USBH_OK..USBH_UNRECOVERED_ERROR (usbh_def.h:303-311) */
enum {
  SDRFM_OK = 0,
  SDRFM_BUSY = 1,
  SDRFM_FAIL = 2,
  SDRFM_NOT_SUPPORTED = 3,
  SDRFM_UNRECOVERED_ERROR = 4,
  /* extensions */
  SDRFM_EINVAL = 16,      /* NULL pointer, zero sizes, bad struct_size, non-finite taps            */
  SDRFM_EODD = 17,        /* nbytes not a multiple of 2 (half an I/Q pair)                          */
  SDRFM_ECAPACITY = 18,   /* audio buffer / stride too small, or nbytes > max_bytes_per_call        */
  SDRFM_NO_DEVICE = 19,   /* no usable gfx950 HIP device, or HIP runtime error at create            */
  SDRFM_ENOMEM = 20
};

#define SDRFM_MAX_TAPS 256u       /* limit for fir_taps and audio_taps */
#define SDRFM_MAX_DECIM 64u

/* flags for sdrfm_config.flags */
#define SDRFM_CFG_FORCE_GENERIC 1u  /* never select a (T,D)-specialised kernel: run the generic kernel (tests) */
#define SDRFM_CFG_BIT_EXACT     4u  /* only kernels whose audio is bit-identical to the fp32 fmaf-chain definition above (the generic kernel's):
                                       never the matrix-pipe kernel ("fast-q"), whose audio is within the tolerance of that definition for any
                                       input (see above) but not bit-identical to it.  Without the flag "fast-q" serves low-pass channel
                                       filters of up to 64 taps (sum|h| <= 2 |sum h|: a performance rule, not a correctness condition) at
                                       D = 10, 32 audio taps / 5 (and at D = 8 / 8, D = 16 / 5) — but for the streams that turn out to be
                                       noise only, which the bit-identical kernels serve faster: those are routed to them per stream;
                                       every other configuration runs the bit-identical kernels anyway */
#define SDRFM_CFG_NO_ZEROCOPY   2u  /* URB-sized host calls use the staged H2D/D2H path instead of mapped host memory (tests) */
#define SDRFM_CFG_GUARD_WORST_CASE 8u  /* "fast-q": the conditioning guard's radius from the PROVEN worst case of |y_fast-q - y_definition| — every rounding of
                                       the definition's chain and of the recombination falling the same way at the largest partial sum, every tap's
                                       quantisation error against a full-scale byte: (T + 4) 127.5 sum|h| 2^-24 + 64 T q (csrc/qtaps.c) — instead of the
                                       statistical bound; 6.9 times the radius at 64 taps (|y| < 32 of 127.5 is recomputed by the definition's chain).
                                       With it every unrepaired discriminator output is PROVABLY within 5e-6 / max|g| + 1.1e-6 of the definition's;
                                       a carrier above a quarter of full scale never meets the guard and costs the same (profiles/r06_guard_worst_case.txt),
                                       weaker streams are repaired more often and then routed to the bit-identical kernels.  See the note above. */

/* flags for sdrfm_process_batch */
#define SDRFM_F_DEVICE_PTRS 1u    /* iq and audio are device pointers on cfg.device; call is enqueued on the
                                     handle's stream and returns without synchronising; it does not wait for the device either while the
                                     caller is less than 48 calls ahead of it ("fast-q" handles: the per-stream statistics of a window of 16
                                     calls are read back on a side stream and take effect at a fixed call three windows after the window
                                     closed — a caller further ahead than that waits there for the read-back, which is what makes the kernel
                                     assignment a function of the bytes and the calls and not of timing (so a caller whose stream is held back by
                                     work it has not issued yet must not run that far ahead of it);
                                     tests/test_route_gpu.py test_calls_return_without_waiting_for_the_device).  The handle's own stream is
                                     created non-blocking: it does NOT order itself against the null stream or any
                                     other stream, so work that produces iq or touches audio elsewhere must be
                                     synchronised by the caller, or the caller's stream given via sdrfm_set_stream */

#define SDRFM_F_OVERLAP 2u        /* (with SDRFM_F_DEVICE_PTRS) the call may run on the device CONCURRENTLY with the previous
                                     call made with this flag: a call's start-up then hides under the previous call's tail (12 % more
                                     calls per second on BASELINE configs[2]).  The call is ordered behind what the handle's stream
                                     holds when it is made (so iq may be produced there), but the handle's stream does not wait for
                                     it: sdrfm_flush / sdrfm_synchronize / any call without the flag order the stream behind all
                                     overlapped calls.  The caller promises, until the call has completed: (a) the PREVIOUS call's
                                     iq buffer stays intact — the call warms its streams up from that buffer's last bytes instead of
                                     waiting for the previous call's state —, and (b) audio is not a buffer the previous overlapped
                                     call writes (two audio buffers taken in turn).  Same audio, bit for bit, as without the flag.
                                     Calls that the matrix-pipe kernel does not serve (see SDRFM_CFG_BIT_EXACT), and the first call
                                     after create / reset / a host-pointer call, run as if the flag were absent */

typedef struct sdrfm_config {
  uint32_t struct_size;           /* = sizeof(sdrfm_config) */
  uint32_t n_streams;             /* independent IQ streams processed per call (>= 1) */
  uint32_t fir_taps;              /* T  : channel low-pass taps, 1..SDRFM_MAX_TAPS */
  uint32_t fir_decim;             /* D  : 1..SDRFM_MAX_DECIM (2.4 MS/s / 10 = 240 kS/s, the rate the firmware
                                          programs: usbh_rtlsdr.c:898) */
  const float* fir_coeffs;        /* h[0..T), copied at create */
  uint32_t audio_taps;            /* Ta : audio low-pass taps, 1..SDRFM_MAX_TAPS */
  uint32_t audio_decim;           /* Da : 1..SDRFM_MAX_DECIM (240 kS/s / 5 = 48 kHz) */
  const float* audio_coeffs;      /* g[0..Ta), copied at create */
  uint32_t max_bytes_per_call;    /* per-stream upper bound for nbytes (sizes device staging); 0 = 1 MiB */
  int32_t  device;                /* HIP device ordinal */
  uint32_t flags;                 /* SDRFM_CFG_* */
} sdrfm_config;

typedef struct sdrfm sdrfm_t;

/* Create / destroy. The handle owns its device buffers, streaming state and output staging — it plays the role
 * of the class handle malloc'd in InterfaceInit and freed in InterfaceDeInit (usbh_rtlsdr.c:182-183,635-638). */
int  sdrfm_create(const sdrfm_config* cfg, sdrfm_t** out);
void sdrfm_destroy(sdrfm_t* h);

/* Zero all streaming state (as after create). */
int  sdrfm_reset(sdrfm_t* h);

/* Number of audio samples per stream that the NEXT process call with `nbytes` will produce (depends on the
 * decimator phases carried in the handle). */
int  sdrfm_audio_count(const sdrfm_t* h, uint32_t nbytes, uint32_t* n_audio);

/* Single-stream hand-off (handle must have n_streams == 1): host buffer in, host audio out, synchronous.
 *   iq      : interleaved uint8 I0 Q0 I1 Q1 ... exactly as the RTL2832 bulk pipe delivers it
 *             (RTLSDR_CommItfTypedef.buff, usbh_rtlsdr.h:165-173)
 *   nbytes  : valid bytes (USBH_LL_GetLastXferSize, usbh_conf.c:350-353); must be even; 0 is a no-op
 *   audio   : receives *n_audio floats (radians per 240 kS/s sample, low-passed, at 48 kHz)            */
int  sdrfm_process(sdrfm_t* h, const uint8_t* iq, uint32_t nbytes,
                   float* audio, uint32_t audio_cap, uint32_t* n_audio);

/* Batched hand-off: n_streams buffers of nbytes_per_stream bytes, stream s at iq + s*iq_stride (bytes);
 * audio for stream s at audio + s*audio_stride (floats). All streams advance by the same amount, so
 * *n_audio (per stream) is one number. With SDRFM_F_DEVICE_PTRS the buffers are device memory and the work is
 * only enqueued (use sdrfm_synchronize or the stream). */
int  sdrfm_process_batch(sdrfm_t* h, const uint8_t* iq, size_t iq_stride, uint32_t nbytes_per_stream,
                         float* audio, size_t audio_stride, uint32_t* n_audio, uint32_t flags);

/* Run on a caller-owned HIP stream (hipStream_t passed as void*; NULL = the handle's own stream). */
int  sdrfm_set_stream(sdrfm_t* h, void* hip_stream);
int  sdrfm_synchronize(sdrfm_t* h);
/* Order the handle's stream behind every call made with SDRFM_F_OVERLAP so far (enqueues two event waits; does not block the host). */
int  sdrfm_flush(sdrfm_t* h);
/* The same for every overlapped call but the most recent one: a consumer of call k-1's audio that runs on the handle's stream is put behind
 * call k-1 only, so call k keeps running beside it (and call k+1, ordered behind the consumer, may reuse call k-1's audio buffer). */
int  sdrfm_flush_previous(sdrfm_t* h);
/* The same ordering given to ANOTHER stream of the caller's (hipStream_t): `hip_stream` is put behind every overlapped call but the most recent one, and the
 * handle's own stream is left alone — so a consumer of call k-1's audio on a stream of its own (the device PCM sink, a copy engine) runs beside call k, and call
 * k+1 is ordered behind nothing but its own inputs: the consumer loop of INTEGRATION.md at the demodulator's own rate (23 us per BASELINE configs[2] call with the
 * device PCM sink as the consumer; a consumer on the handle's stream holds every second call back: 43 us).  The caller orders its reuse of an audio buffer against
 * that consumer itself (three audio buffers and an event).  Non-blocking. */
int  sdrfm_wait_previous(sdrfm_t* h, void* hip_stream);

/* Introspection used by bench/tests: the kernel variant that served the LAST call (before any call: the one the
 * configuration selects), e.g. "fast T64 D10 R4 Ta32 Da5" or "generic T7 D3 Ta5 Da4 NA64". */
const char* sdrfm_kernel_name(const sdrfm_t* h);
uint32_t    sdrfm_abi_version(void);
const char* sdrfm_strerror(int status);

/* ------------------------------------------------------------------------------------------------------------------
 * Streaming front-end adapter: a ring of n_buffers pinned host buffers of buffer_bytes each — the multi-buffer scheme the
 * reference declares but never uses (DEFAULT_BUF_NUMBER 15 x DEFAULT_BUF_LENGTH 16*32*512, usbh_rtlsdr.h:277-278).
 * Single-stream handles only.  submit() copies the just-filled USB buffer into the next free slot and enqueues
 * H2D -> kernel -> D2H without waiting (the caller's buffer may be re-armed at once); collect() returns finished audio in
 * submission order.  Both are non-blocking and answer SDRFM_BUSY (ring full / nothing ready), like the reference's FSM
 * steps answer USBH_BUSY; collect(wait != 0) blocks for the oldest slot.  Do not mix with sdrfm_process on the same handle
 * while slots are in flight.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct sdrfm_ring sdrfm_ring_t;
int  sdrfm_ring_create(sdrfm_t* h, uint32_t n_buffers, uint32_t buffer_bytes, sdrfm_ring_t** out);
void sdrfm_ring_destroy(sdrfm_ring_t* r);
int  sdrfm_ring_submit(sdrfm_ring_t* r, const uint8_t* iq, uint32_t nbytes);
int  sdrfm_ring_collect(sdrfm_ring_t* r, float* audio, uint32_t audio_cap, uint32_t* n_audio, int wait);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-channel WBFM (BASELINE configs[4]): one wide capture (3.2 MS/s) -> 16-band critically-sampled polyphase
 * channelizer (prototype low-pass p[0..P), P a multiple of 16; each band at fs/16 = 200 kS/s) -> per-band FM
 * discriminator -> rational L/M resampler (6/25 -> 48 kHz) with prototype g[0..Tg) given at the L-times-upsampled rate.
 * Same hand-off contract as sdrfm_process_batch (buffer format, status codes, threading); arithmetic: DESIGN.md.
 * audio layout: audio[(stream * 16 + band) * band_stride + j].
 * ------------------------------------------------------------------------------------------------------------------ */
#define SDRFM_WBFM_BANDS 16
/* flags for sdrfm_wbfm_config.flags (all are test hooks; results do not depend on them) */
#define SDRFM_WBFM_CFG_FORCE_GENERIC 1u        /* never run a fused kernel */
#define SDRFM_WBFM_CFG_BRANCH_LANES 2u         /* fused kernel with one lane per polyphase branch instead of one lane per step */
#define SDRFM_WBFM_CFG_RUN_STEPS_SHIFT 8       /* flags >> 8 = fixed run length (steps, 64..8192) of the fused kernel, 0 = chosen per call */

typedef struct sdrfm_wbfm_config {
  uint32_t struct_size;           /* = sizeof(sdrfm_wbfm_config) */
  uint32_t n_streams;
  uint32_t proto_taps;            /* P: multiple of 16, <= 512 */
  const float* proto_coeffs;
  uint32_t resamp_taps;           /* Tg <= 512 */
  uint32_t resamp_up;             /* L <= 64 */
  uint32_t resamp_down;           /* M <= 256 */
  const float* resamp_coeffs;
  uint32_t max_bytes_per_call;    /* per stream; 0 = 1 MiB */
  int32_t  device;
  uint32_t flags;                 /* SDRFM_WBFM_CFG_* */
} sdrfm_wbfm_config;

typedef struct sdrfm_wbfm sdrfm_wbfm_t;

int  sdrfm_wbfm_create(const sdrfm_wbfm_config* cfg, sdrfm_wbfm_t** out);
void sdrfm_wbfm_destroy(sdrfm_wbfm_t* h);
int  sdrfm_wbfm_reset(sdrfm_wbfm_t* h);
int  sdrfm_wbfm_audio_count(const sdrfm_wbfm_t* h, uint32_t nbytes, uint32_t* n_audio_per_band);
int  sdrfm_wbfm_process_batch(sdrfm_wbfm_t* h, const uint8_t* iq, size_t iq_stride, uint32_t nbytes_per_stream,
                              float* audio, size_t band_stride, uint32_t* n_audio_per_band, uint32_t flags);
int  sdrfm_wbfm_set_stream(sdrfm_wbfm_t* h, void* hip_stream);
int  sdrfm_wbfm_synchronize(sdrfm_wbfm_t* h);
/* the kernel that served the LAST call (before any call: the one the configuration selects): starts with "wbfm-fused" (128-tap
   prototype, <= 10 resampler taps per phase; the kernel follows in brackets) or "wbfm-generic" (any shape, two kernels; also
   serves calls too short for a fused kernel) */
const char* sdrfm_wbfm_kernel_name(const sdrfm_wbfm_t* h);

/* ------------------------------------------------------------------------------------------------------------------
 * Spectrum view of the IQ buffer — the reference's own next task ("Perform some FFT on the samples to check what we are
 * receiving", README.md:29) on the same buffer contract (RTLSDR_CommItfTypedef.buff, usbh_rtlsdr.h:165-173): per stream
 * the windowed nfft-point power spectrum averaged over the consecutive, non-overlapping frames of the buffer, DC in the
 * middle: power[stream * power_stride + i], i = 0..nfft-1, bin i = frequency (i - nfft/2) * fs/nfft.  Stateless: each
 * call views the buffer it is given; a tail shorter than nfft is ignored (*n_frames = 0 -> all zeros).
 * Arithmetic (one fixed radix-2 DIT graph, fp32): DESIGN.md §4.6.  window = NULL selects a periodic Hann window.
 * SDRFM_FAIL from a host-buffer call or from sdrfm_spectrum_synchronize also reports a device-side time-out: the 512 / 1024-point kernel hands
 * a stream's running sum from wave to wave, and a wave that waited two seconds for its predecessor (a wait of ~100 us when nothing is wrong) sets
 * an error word instead of hanging or aborting the context; the powers of that call are not valid, the handle stays usable.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct sdrfm_spectrum_config {
  uint32_t struct_size;           /* = sizeof(sdrfm_spectrum_config) */
  uint32_t n_streams;
  uint32_t nfft;                  /* power of two, 64 .. 4096 */
  const float* window;            /* nfft floats or NULL */
  uint32_t max_bytes_per_call;    /* per stream; 0 = 1 MiB */
  int32_t  device;
  uint32_t flags;                 /* must be 0 */
} sdrfm_spectrum_config;

typedef struct sdrfm_spectrum sdrfm_spectrum_t;

int  sdrfm_spectrum_create(const sdrfm_spectrum_config* cfg, sdrfm_spectrum_t** out);
void sdrfm_spectrum_destroy(sdrfm_spectrum_t* h);
int  sdrfm_spectrum_process_batch(sdrfm_spectrum_t* h, const uint8_t* iq, size_t iq_stride, uint32_t nbytes_per_stream,
                                  float* power, size_t power_stride, uint32_t* n_frames, uint32_t flags);
int  sdrfm_spectrum_set_stream(sdrfm_spectrum_t* h, void* hip_stream);
int  sdrfm_spectrum_synchronize(sdrfm_spectrum_t* h);
/* The kernel the last call launched, as a profiler prints it (before the first call: the one a device buffer with even iq and iq_stride
 * gets).  Up to 1024 points a kernel that reads the samples as halves of aligned dwords serves the call; one whose iq or iq_stride is odd is
 * served by the typed-load kernel of the longer lengths.  Same results either way. */
const char* sdrfm_spectrum_kernel_name(const sdrfm_spectrum_t* h);

/* ------------------------------------------------------------------------------------------------------------------
 * Host-only helpers (no GPU): the two pure computations the reference performs when it programs the RTL2832 for this
 * stream, so that a non-MCU front end configures a dongle identically.
 *   sdrfm_rtl_pack_fir  : RTLSDR_set_fir, state RTLSDR_FIR_CALC (Class/RTLSDR/Src/usbh_rtlsdr.c:552-575): RTLSDR_FIR[16]
 *                         (8 x int8 then 8 x int12) -> 20 bytes for demod page 1 regs 0x1c..0x2f.
 *   sdrfm_rtl_resampler : RTLSDR_set_sample_rate state 0 (usbh_rtlsdr.c:676-691); xtal_hz = 28 800 000 for the stock dongle.
 *   sdrfm_e4k_pll_params: E4K_compute_pll_params (Class/RTLSDR/Src/tuner_e4k.c:689-737, band table :301-312): the E4000
 *                         synthesiser word for a wanted LO — band divider R and its SYNTH7 code, integer part Z, 16-bit
 *                         fraction X (Y = 65536) and the LO actually obtained, floor(fosc*(Z + X/Y) / R) in integers.
 * All return SDRFM_EINVAL where the firmware would only log (or, for the PLL, return 0).
 * ------------------------------------------------------------------------------------------------------------------ */
int sdrfm_rtl_pack_fir(const int* fir16, uint8_t* out20);
int sdrfm_rtl_resampler(uint32_t samp_rate, uint32_t xtal_hz, uint32_t* rsamp_ratio, uint32_t* real_rsamp_ratio,
                        double* real_rate);
/* Multi-GPU fan-out / fan-in of stream batches (streams are independent: one handle per GPU, no data-path collective): the contiguous
 * block [*first, *first + *count) of the n_streams streams that rank `rank` of `world` owns; the first n_streams % world ranks own one
 * stream more.  examples/multi_gpu_main.c (one C process, N devices, RCCL send/recv) and the Python fan-out use this one definition. */
int sdrfm_shard_range(uint32_t n_streams, uint32_t world, uint32_t rank, uint32_t* first, uint32_t* count);

typedef struct sdrfm_e4k_pll {   /* mirrors struct e4k_pll_params (Class/RTLSDR/Inc/tuner_e4k.h:227-236) */
  uint32_t fosc, intended_flo, flo;
  uint16_t x;
  uint8_t z, r, r_idx, threephase;
} sdrfm_e4k_pll;
int sdrfm_e4k_pll_params(uint32_t fosc_hz, uint32_t intended_flo_hz, sdrfm_e4k_pll* out);

/* Audio sink format of the reference board (host-side, plain C): de-emphasis y += alpha*(x - y) carried in *state, then
 * int16 stereo-interleaved PCM (L = R) as BSP_AUDIO_OUT_Play(uint16_t*, Size) takes it
 * (Utilities/STM32746G-Discovery/stm32746g_discovery_audio.c:224).  pcm_stereo receives 2*n samples. */
int   sdrfm_pcm_deemph_s16(const float* audio, uint32_t n, float alpha, float gain, float* state, int16_t* pcm_stereo);
float sdrfm_pcm_alpha(float fs_hz, float tau_s);   /* 1 - exp(-1/(fs*tau)); tau = 75e-6 (US) / 50e-6 (EU) */

/* The same sink ON THE DEVICE for the batched path: audio[stream * audio_stride + i] (f32, as sdrfm_process_batch leaves it) ->
 * pcm[stream * pcm_stride + 2*i + {0,1}] (int16, L = R), de-emphasis state carried per stream in the handle.  pcm_stride is in int16
 * elements, even, >= 2*n.  With SDRFM_F_DEVICE_PTRS both buffers are device memory (pcm 4-byte aligned) and the call only
 * enqueues on the sink's stream; give it the demodulator's stream (or synchronise) so that it runs after the audio exists.
 * Two forms of the same recursion (csrc/sdrfm_sink.hip):
 *   default            a blocked scan — 256 lanes per stream walk chunks of the call, the carries between chunks by a scan, every chunk then
 *                      re-walked from its true carry-in with the exact form's own operations: PCM within 1 LSB of the exact form's (different only
 *                      where y * gain sits on a rounding boundary), carried state within 1e-6 relative; microseconds per 256 x 4800 call
 *                      (profiles/r06_sink.txt);
 *   SDRFM_PCM_F_EXACT  one lane per stream, the operations of sdrfm_pcm_deemph_s16 in its order: BIT-IDENTICAL to it (and to an exact-rational
 *                      restatement: tests/test_pcm_sink_gpu.py); latency-bound: 1.35 ms per 256 x 4800 call. */
#define SDRFM_PCM_F_EXACT 4u      /* flag for sdrfm_pcm_sink_process_batch (beside SDRFM_F_DEVICE_PTRS) */
typedef struct sdrfm_pcm_sink sdrfm_pcm_sink_t;
int  sdrfm_pcm_sink_create(uint32_t n_streams, float alpha, float gain, int32_t device, sdrfm_pcm_sink_t** out);
void sdrfm_pcm_sink_destroy(sdrfm_pcm_sink_t* k);
int  sdrfm_pcm_sink_reset(sdrfm_pcm_sink_t* k);
int  sdrfm_pcm_sink_process_batch(sdrfm_pcm_sink_t* k, const float* audio, size_t audio_stride, uint32_t n,
                                  int16_t* pcm, size_t pcm_stride, uint32_t flags);
int  sdrfm_pcm_sink_set_stream(sdrfm_pcm_sink_t* k, void* hip_stream);
int  sdrfm_pcm_sink_synchronize(sdrfm_pcm_sink_t* k);
int  sdrfm_pcm_sink_get_state(sdrfm_pcm_sink_t* k, float* state_out /* n_streams floats */);

/* ONE call of the demodulator and of the sink (round 6): sdrfm_process_batch(h, iq, ..., audio, ..., flags) and then the sink's default form over that
 * audio into pcm.  With SDRFM_F_DEVICE_PTRS the buffers are device memory and the call only enqueues (SDRFM_F_OVERLAP as for sdrfm_process_batch, with pcm
 * rotated like audio); without it they are host memory and the call stages, runs and copies back, synchronous like sdrfm_process_batch with host buffers — the
 * reference superloop's two steps on one filled CommItf.buff: demodulate it, hand int16 stereo to BSP_AUDIO_OUT_Play.
 * Where the matrix-pipe kernel serves the whole call, the sink's chain runs INSIDE its launch (csrc/sdrfm_sink_chain.h): the de-emphasis forgets — (1 - alpha)^64
 * is below rounding —, so every wave sinks the ~400 outputs it has just computed where they lie, publishes its end state in one word, and finishes its first 64
 * outputs with its neighbour's.  No second launch, no stream to order, nothing between two overlapped calls: the consumer loop of INTEGRATION.md section 3 runs
 * within 10 % of the demodulator's own rate (profiles/r06_sink.txt).  A batch with streams routed to the bit-exact kernels keeps the chain for the others (both
 * designs in one launch) and is followed, on the call's own queue, by the sink's kernel over the routed streams only (+4.5 us per call).  Any other call (the
 * bit-exact kernels alone, the first call of a stream, a sink whose alpha is below 0.231) is followed by the sink's own kernel on the handle's stream, behind a join
 * of the overlapped calls.
 * Same results either way up to the blocked scan's tolerance: PCM within 1 LSB of sdrfm_pcm_deemph_s16's, state within 1e-6 relative.
 * audio may be NULL: the PCM is then all the call leaves (a launch that holds the chain does not store the float audio at all; other calls use rows of the
 * library's own); otherwise the audio is written as by sdrfm_process_batch.
 * The sink must belong to h's device and have h's n_streams; between calls made this way and sdrfm_pcm_sink_process_batch calls on the same sink, synchronise
 * both (the chain of call c waits ON THE DEVICE for the sink's call c - 1, which must already be in a queue); sdrfm_pcm_sink_get_state, _reset and _destroy
 * after sdrfm_synchronize(h) (the launches hold pointers into the sink).  The wait on the device is bounded (2^19 polls: about a second): should it ever run out — a protocol error —
 * the run goes on from state 0 and sdrfm_pcm_sink_synchronize / _get_state answer SDRFM_FAIL from then on (until _reset).
 * kernel name: "... + pcm". */
int  sdrfm_process_batch_pcm(sdrfm_t* h, sdrfm_pcm_sink_t* sink, const uint8_t* iq, size_t iq_stride, uint32_t nbytes, float* audio, size_t audio_stride,
                             int16_t* pcm, size_t pcm_stride, uint32_t* n_audio, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif /* SDRFM_H */
