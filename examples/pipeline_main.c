/*
 * pipeline_main.c — plain C99 host of the batched, device-resident path as a PIPELINE: a ring of capture buffers on the device, the
 * demodulator called with SDRFM_F_OVERLAP (consecutive calls run concurrently on the GPU), and the PCM sink of the reference board
 * (BSP_AUDIO_OUT_Play's int16 stereo, Utilities/STM32746G-Discovery/stm32746g_discovery_audio.c:224) consuming call k - 1's audio on
 * the same HIP stream while call k runs — the loop INTEGRATION.md section 3 shows.  What fills the ring here is a file; in the reference's
 * setting it is the bulk-IN FSM (one filled CommItf.buff per RTLSDR_XFER_COMPLETE, usbh_rtlsdr.c:1058-1101) of many dongles.
 *
 *   pipeline_main <iq.u8> <h.f32> <g.f32> <n_streams> <nbytes_per_stream_per_call> <n_calls> <pcm_out.s16> [--serial | --one-call]
 *   (no timing here: with the capture coming from pageable host memory the H2D copies set the pace; bench.py measures the calls)
 *
 * iq.u8 holds n_calls consecutive batches of [n_streams][nbytes] bytes (batch-major).  pcm_out: n_calls blocks of [n_streams][2 n_audio]
 * int16.  --serial makes the same calls without the flag (the output must be the same, bit for bit).  --one-call is the loop with no consumer at all:
 * sdrfm_process_batch_pcm — the sink runs inside the demodulator's own launch, no audio buffer, nothing to order (PCM within 1 LSB of the other modes').
 * Prints one JSON line.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdrfm.h"

#define RING 3   /* capture buffers on the device: call k reads ring[k % RING], ring[(k - 1) % RING] stays intact until call k is done */

#define HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
#define SDRC(call) do { int s_ = (call); if (s_ != SDRFM_OK) { fprintf(stderr, "%s: %s\n", #call, sdrfm_strerror(s_)); return 1; } } while (0)

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(sz > 0 ? (size_t)sz : 1);
  if (fread(p, 1, (size_t)sz, f) != (size_t)sz) { perror("fread"); exit(2); }
  fclose(f);
  *n = (size_t)sz;
  return p;
}

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s iq.u8 h.f32 g.f32 n_streams nbytes_per_stream_per_call n_calls pcm_out.s16 [--serial]\n", argv[0]); return 2; }
  size_t niq, nh, ng;
  unsigned char* iq = (unsigned char*)slurp(argv[1], &niq);
  float* h = (float*)slurp(argv[2], &nh);
  float* g = (float*)slurp(argv[3], &ng);
  const uint32_t ns = (uint32_t)atoi(argv[4]), nbytes = (uint32_t)atoi(argv[5]);
  const int n_calls = atoi(argv[6]);
  const int serial = argc > 8 && !strcmp(argv[8], "--serial");
  const int one_call = argc > 8 && !strcmp(argv[8], "--one-call");
  const size_t batch = (size_t)ns * nbytes;
  if (niq < batch * (size_t)n_calls) { fprintf(stderr, "iq file too short\n"); return 2; }

  sdrfm_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg; cfg.n_streams = ns;
  cfg.fir_taps = (uint32_t)(nh / sizeof(float)); cfg.fir_decim = 10; cfg.fir_coeffs = h;
  cfg.audio_taps = (uint32_t)(ng / sizeof(float)); cfg.audio_decim = 5; cfg.audio_coeffs = g;
  cfg.max_bytes_per_call = nbytes; cfg.device = 0;
  sdrfm_t* fm = NULL;
  SDRC(sdrfm_create(&cfg, &fm));
  uint32_t n_audio = 0;
  SDRC(sdrfm_audio_count(fm, nbytes, &n_audio));
  const size_t astride = (n_audio + 63u) & ~(size_t)63u;
  sdrfm_pcm_sink_t* sink = NULL;
  SDRC(sdrfm_pcm_sink_create(ns, sdrfm_pcm_alpha(48000.0f, 75e-6f), 16688.0f /* ~ 32767 / (2 pi 75 kHz / 240 kHz): full deviation = full scale */, 0, &sink));

  hipStream_t st;
  HIPC(hipSetDevice(0));
  HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  SDRC(sdrfm_set_stream(fm, st));
  SDRC(sdrfm_pcm_sink_set_stream(sink, st));
  unsigned char* d_ring[RING];
  float* d_audio[2];
  short* d_pcm;
  for (int i = 0; i < RING; ++i) HIPC(hipMalloc((void**)&d_ring[i], batch));
  for (int i = 0; i < 2; ++i) HIPC(hipMalloc((void**)&d_audio[i], (size_t)ns * astride * sizeof(float)));
  HIPC(hipMalloc((void**)&d_pcm, (size_t)n_calls * ns * 2 * astride * sizeof(short)));
  const uint32_t flags = SDRFM_F_DEVICE_PTRS | (serial ? 0u : SDRFM_F_OVERLAP);

  uint32_t na = 0;
  for (int k = 0; k < n_calls; ++k) {
    /* "the capture": batch k into ring[k % RING] on the stream.  The buffer was last read by call k - RING, which the consumer of call
       k - RING (already enqueued on this stream) waited for. */
    HIPC(hipMemcpyAsync(d_ring[k % RING], iq + (size_t)k * batch, batch, hipMemcpyHostToDevice, st));
    if (one_call) {                                            /* demodulator and sink in one call: the PCM is all it leaves */
      SDRC(sdrfm_process_batch_pcm(fm, sink, d_ring[k % RING], nbytes, nbytes, NULL, 0, d_pcm + (size_t)k * ns * 2 * astride, 2 * astride, &na, flags));
      continue;
    }
    SDRC(sdrfm_process_batch(fm, d_ring[k % RING], nbytes, nbytes, d_audio[k & 1], astride, &na, flags));
    if (k > 0) {                                               /* the consumer of call k - 1, behind call k - 1 only */
      SDRC(serial ? SDRFM_OK : sdrfm_flush_previous(fm));
      SDRC(sdrfm_pcm_sink_process_batch(sink, d_audio[(k - 1) & 1], astride, na, d_pcm + (size_t)(k - 1) * ns * 2 * astride, 2 * astride, SDRFM_F_DEVICE_PTRS));
    }
  }
  SDRC(sdrfm_flush(fm));
  if (!one_call)
    SDRC(sdrfm_pcm_sink_process_batch(sink, d_audio[(n_calls - 1) & 1], astride, na, d_pcm + (size_t)(n_calls - 1) * ns * 2 * astride, 2 * astride, SDRFM_F_DEVICE_PTRS));
  HIPC(hipStreamSynchronize(st));
  SDRC(sdrfm_pcm_sink_synchronize(sink));                      /* (also reports a chain error of the one-call mode) */

  short* pcm = (short*)malloc((size_t)n_calls * ns * 2 * astride * sizeof(short));
  HIPC(hipMemcpy(pcm, d_pcm, (size_t)n_calls * ns * 2 * astride * sizeof(short), hipMemcpyDeviceToHost));
  FILE* fo = fopen(argv[7], "wb");
  if (!fo) { perror(argv[7]); return 2; }
  for (int k = 0; k < n_calls; ++k)
    for (uint32_t s = 0; s < ns; ++s) fwrite(pcm + ((size_t)k * ns + s) * 2 * astride, sizeof(short), 2 * (size_t)na, fo);
  fclose(fo);
  printf("{\"n_streams\":%u,\"bytes_per_stream_per_call\":%u,\"n_calls\":%d,\"n_audio\":%u,\"mode\":\"%s\",\"kernel\":\"%s\"}\n", ns, nbytes, n_calls,
         na, serial ? "serial" : (one_call ? "one-call" : "overlapped"), sdrfm_kernel_name(fm));

  sdrfm_pcm_sink_destroy(sink);
  sdrfm_destroy(fm);
  for (int i = 0; i < RING; ++i) HIPC(hipFree(d_ring[i]));
  for (int i = 0; i < 2; ++i) HIPC(hipFree(d_audio[i]));
  HIPC(hipFree(d_pcm));
  HIPC(hipStreamDestroy(st));
  free(pcm); free(iq); free(h); free(g);
  return 0;
}
