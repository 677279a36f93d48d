/*
 * consumer_loop_main.c — plain C99: what the consumer loop of INTEGRATION.md section 3 costs per call, timed from a C host (the reference's host is one C
 * superloop, src/main.c:72-80; a Python host spends 30 us per iteration on its own).  BASELINE configs[2] shape by default: 256 streams x 0.1 s of 2.4 MS/s IQ
 * per call, 64-tap FIR / 10 + FM discriminator + 32 taps / 5, five capture buffers in turn (614 MB: cold HBM reads); the consumer is the device PCM sink of the
 * reference board's format (BSP_AUDIO_OUT_Play, Utilities/STM32746G-Discovery/stm32746g_discovery_audio.c:224).  Forms:
 *   calls     overlapped calls (SDRFM_F_OVERLAP), one sdrfm_flush per region — no consumer: the demodulator's own rate
 *   simple    two audio buffers, sdrfm_flush_previous + the sink on the HANDLE'S stream after every call
 *   fast      NA audio buffers, the sink on a stream OF ITS OWN behind sdrfm_wait_previous, the reuse of an audio buffer guarded by hipEventQuery on the host
 *             (the host then stays at most NA calls ahead of the device: NA = 3 keeps the queues nearly empty, NA = 6 keeps them fed)
 *   fused     sdrfm_process_batch_pcm: overlapped calls whose launch ends with the sink's chain (csrc/sdrfm_sink_chain.h) — no second launch, no stream to
 *             order; NA audio and PCM buffers in turn, one sdrfm_flush per region
 *   fused_pcm_only   the same without an audio buffer: the launch stores the PCM only
 *
 *   consumer_loop_main [n_streams=256] [regions=10] [calls_per_region=300] [NA=6]
 * Prints one JSON line: us per call of every region and the median of the later half, per form.  Measurement only: nothing is checked against the oracle here
 * (tests/test_pcm_sink_gpu.py holds the same arrangement to the host sink).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdrfm.h"

#define NBUF 5
#define HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
#define SDRC(call) do { int s_ = (call); if (s_ != SDRFM_OK) { fprintf(stderr, "%s: %s\n", #call, sdrfm_strerror(s_)); return 1; } } while (0)

static void lowpass(float* h, int T, double fc) {   /* Hamming-windowed sinc, unit DC gain */
  double s = 0.0;
  for (int k = 0; k < T; ++k) {
    const double x = k - (T - 1) / 2.0, w = 0.54 - 0.46 * cos(2 * M_PI * k / (T - 1));
    const double v = (x == 0.0 ? 2 * fc : sin(2 * M_PI * fc * x) / (M_PI * x)) * w;
    h[k] = (float)v; s += v;
  }
  for (int k = 0; k < T; ++k) h[k] = (float)(h[k] / s);
}

static int cmp_d(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

int main(int argc, char** argv) {
  const uint32_t ns = argc > 1 ? (uint32_t)atoi(argv[1]) : 256u;
  const int regions = argc > 2 ? atoi(argv[2]) : 10, per = argc > 3 ? atoi(argv[3]) : 300;
  int NA = argc > 4 ? atoi(argv[4]) : 6;                       /* audio buffers of the fast form */
  if (NA < 3) NA = 3;
  if (NA > 8) NA = 8;
  const uint32_t nsamp = 240000u, nbytes = 2u * nsamp;
  float h[64], g[32];
  lowpass(h, 64, 100e3 / 2.4e6); lowpass(g, 32, 15e3 / 240e3);

  /* the capture: 8 distinct FM carriers (three audio tones, 75 kHz deviation, +-20 kHz offsets), NBUF consecutive pieces each, tiled over the streams */
  const size_t total = (size_t)NBUF * nsamp;
  unsigned char* rows = (unsigned char*)malloc(8 * 2 * total);
  if (!rows) return 2;
  unsigned long long rs = 88172645463325252ull;
  for (int r = 0; r < 8; ++r) {
    double ph = 0.3 * r;
    const double fcar = (r * 5 - 20) * 1000.0;
    for (size_t n = 0; n < total; ++n) {
      const double t = (double)n / 2.4e6, a = 0.5 * sin(2 * M_PI * 1000 * t) + 0.3 * sin(2 * M_PI * 3100 * t) + 0.2 * sin(2 * M_PI * 7300 * t);
      ph += 2 * M_PI * (fcar + 75e3 * a) / 2.4e6;
      rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
      const double ni = ((double)(rs >> 40) / 16777216.0 - 0.5) * 8, nq = ((double)((rs >> 16) & 0xffffff) / 16777216.0 - 0.5) * 8;
      double vi = 127.5 + 100 * cos(ph) + ni, vq = 127.5 + 100 * sin(ph) + nq;
      vi = vi < 0 ? 0 : (vi > 255 ? 255 : vi); vq = vq < 0 ? 0 : (vq > 255 ? 255 : vq);
      rows[((size_t)r * total + n) * 2] = (unsigned char)lrint(vi); rows[((size_t)r * total + n) * 2 + 1] = (unsigned char)lrint(vq);
    }
  }

  HIPC(hipSetDevice(0));
  sdrfm_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg; cfg.n_streams = ns; cfg.fir_taps = 64; cfg.fir_decim = 10; cfg.fir_coeffs = h;
  cfg.audio_taps = 32; cfg.audio_decim = 5; cfg.audio_coeffs = g; cfg.max_bytes_per_call = nbytes; cfg.device = 0;
  sdrfm_t* fm = NULL;
  SDRC(sdrfm_create(&cfg, &fm));
  uint32_t na = 0;
  SDRC(sdrfm_audio_count(fm, nbytes, &na));
  const size_t astride = (na + 63u) & ~(size_t)63u;
  sdrfm_pcm_sink_t* sink = NULL;
  SDRC(sdrfm_pcm_sink_create(ns, sdrfm_pcm_alpha(48000.0f, 75e-6f), 16688.0f, 0, &sink));
  hipStream_t st, sst;
  HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  {                                                            /* the sink's stream: SINK_PRIO = low | high selects a priority class (default: the ordinary one) */
    int least = 0, greatest = 0;
    HIPC(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const char* pr = getenv("SINK_PRIO");
    if (pr && !strcmp(pr, "low")) HIPC(hipStreamCreateWithPriority(&sst, hipStreamNonBlocking, least));
    else if (pr && !strcmp(pr, "high")) HIPC(hipStreamCreateWithPriority(&sst, hipStreamNonBlocking, greatest));
    else HIPC(hipStreamCreateWithFlags(&sst, hipStreamNonBlocking));
  }
  SDRC(sdrfm_set_stream(fm, st));
  unsigned char* d_iq[NBUF];
  float* d_audio[8];
  short* d_pcm[8];
  for (int b = 0; b < NBUF; ++b) {
    HIPC(hipMalloc((void**)&d_iq[b], (size_t)ns * nbytes));
    for (uint32_t s = 0; s < ns; ++s)
      HIPC(hipMemcpy(d_iq[b] + (size_t)s * nbytes, rows + ((size_t)(s % 8) * total + (size_t)b * nsamp) * 2, nbytes, hipMemcpyHostToDevice));
  }
  for (int i = 0; i < NA; ++i) HIPC(hipMalloc((void**)&d_audio[i], (size_t)ns * astride * sizeof(float)));
  for (int i = 0; i < NA; ++i) HIPC(hipMalloc((void**)&d_pcm[i], (size_t)ns * 2 * astride * sizeof(short)));
  hipEvent_t e0, e1, consumed[8];
  HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
  for (int i = 0; i < NA; ++i) HIPC(hipEventCreateWithFlags(&consumed[i], hipEventDisableTiming));
  const uint32_t F = SDRFM_F_DEVICE_PTRS | SDRFM_F_OVERLAP;
  uint32_t n = 0;
  long call = 0;                                              /* calls made so far (the capture buffers and the audio buffers go round across regions) */

  printf("{\"n_streams\":%u,\"calls_per_region\":%d,\"fast_form_audio_buffers\":%d", ns, per, NA);
  const char* names[5] = {"calls", "simple", "fast", "fused", "fused_pcm_only"};
  char fused_kernel[128] = "";
  for (int form = 0; form < 5; ++form) {
    SDRC(sdrfm_pcm_sink_set_stream(sink, form == 2 ? sst : st));
    int have[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (getenv("LOOP_PROGRESS")) { fprintf(stderr, "[form %s]\n", names[form]); fflush(stderr); }
    double* us = (double*)malloc(sizeof(double) * (size_t)regions);
    for (int r = 0; r < regions; ++r) {
      if (getenv("LOOP_PROGRESS")) { fprintf(stderr, " r%d", r); fflush(stderr); }
      HIPC(hipEventRecord(e0, st));
      for (int k = 0; k < per; ++k, ++call) {
        const int nab = form >= 2 ? NA : 2, ab = (int)(call % nab), pab = (int)((call + nab - 1) % nab);
        if (form >= 3) {                                       /* (form 4: no audio buffer — the PCM is all the call leaves) */
          SDRC(sdrfm_process_batch_pcm(fm, sink, d_iq[call % NBUF], nbytes, nbytes, form == 3 ? d_audio[ab] : NULL, astride, d_pcm[ab], 2 * astride, &n, F));
          continue;
        }
        if (form == 2 && have[ab]) while (hipEventQuery(consumed[ab]) != hipSuccess) { }    /* audio[ab] is free once its consumer (of call - NA) is done */
        SDRC(sdrfm_process_batch(fm, d_iq[call % NBUF], nbytes, nbytes, d_audio[ab], astride, &n, F));
        if (form == 0 || k == 0) continue;
        if (form == 1) {
          SDRC(sdrfm_flush_previous(fm));
          SDRC(sdrfm_pcm_sink_process_batch(sink, d_audio[pab], astride, n, d_pcm[k & 1], 2 * astride, SDRFM_F_DEVICE_PTRS));
        } else {
          SDRC(sdrfm_wait_previous(fm, sst));
          SDRC(sdrfm_pcm_sink_process_batch(sink, d_audio[pab], astride, n, d_pcm[k & 1], 2 * astride, SDRFM_F_DEVICE_PTRS));
          HIPC(hipEventRecord(consumed[pab], sst));
          have[pab] = 1;
        }
      }
      SDRC(sdrfm_flush(fm));
      if (form == 2) { HIPC(hipEventRecord(consumed[0], sst)); HIPC(hipStreamWaitEvent(st, consumed[0], 0)); have[0] = 1; }
      HIPC(hipEventRecord(e1, st));
      HIPC(hipEventSynchronize(e1));
      float ms = 0.0f;
      HIPC(hipEventElapsedTime(&ms, e0, e1));
      us[r] = (double)ms * 1e3 / per;
    }
    printf(",\"%s\":{\"us_per_call_regions\":[", names[form]);
    for (int r = 0; r < regions; ++r) printf("%s%.2f", r ? "," : "", us[r]);
    double* tail = us + regions / 2;
    const int nt = regions - regions / 2;
    qsort(tail, (size_t)nt, sizeof(double), cmp_d);
    printf("],\"steady_us_per_call\":%.2f}", nt ? tail[nt / 2] : 0.0);
    free(us);
    if (form == 3) snprintf(fused_kernel, sizeof fused_kernel, "%s", sdrfm_kernel_name(fm));
    SDRC(sdrfm_synchronize(fm));
    HIPC(hipStreamSynchronize(sst));
  }
  printf(",\"kernel\":\"%s\"}\n", fused_kernel);

  sdrfm_pcm_sink_destroy(sink);
  sdrfm_destroy(fm);
  for (int b = 0; b < NBUF; ++b) HIPC(hipFree(d_iq[b]));
  for (int i = 0; i < NA; ++i) HIPC(hipFree(d_audio[i]));
  for (int i = 0; i < NA; ++i) HIPC(hipFree(d_pcm[i]));
  free(rows);
  return 0;
}
