/*
 * views_main.c — plain C99 use of the two "views" of the same IQ buffer next to the FM path: the spectrum view
 * (sdrfm_spectrum_*, the firmware's README.md:29 next task) and the 16-channel WBFM path (sdrfm_wbfm_*).
 * Reads a capture file of interleaved u8 I/Q, hands it over in buffSize-byte pieces like the reference's FSM does
 * (one reused buffer, Class/RTLSDR/Src/usbh_rtlsdr.c:1058-1101), and writes raw float32 results.
 *
 *   views_main <iq.u8> <spectrum_out.f32> <wbfm_out.f32> <proto.f32> <resamp.f32> <nfft> <buff_bytes>
 *
 * spectrum_out: nfft floats (power, DC in the middle) of the WHOLE file viewed in one call;
 * wbfm_out:     for every hand-off, 16 rows of n floats (band-major), concatenated in hand-off order.
 * Compiles with -std=c99 -Wall -Wextra -pedantic: include/sdrfm.h is a C header.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdrfm.h"

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

#define CHECK(call) do { int st_ = (call); if (st_ != SDRFM_OK) { fprintf(stderr, "%s: %s\n", #call, sdrfm_strerror(st_)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc != 8) { fprintf(stderr, "usage: %s iq.u8 spectrum.f32 wbfm.f32 proto.f32 resamp.f32 nfft buff_bytes\n", argv[0]); return 2; }
  size_t niq, np_, ng;
  unsigned char* iq = (unsigned char*)slurp(argv[1], &niq);
  float* proto = (float*)slurp(argv[4], &np_);
  float* resamp = (float*)slurp(argv[5], &ng);
  const uint32_t nfft = (uint32_t)atoi(argv[6]), buff_bytes = (uint32_t)atoi(argv[7]);
  niq &= ~(size_t)1;

  /* ---- spectrum view of the whole capture ---- */
  sdrfm_spectrum_config sc;
  memset(&sc, 0, sizeof sc);
  sc.struct_size = sizeof sc; sc.n_streams = 1; sc.nfft = nfft; sc.window = NULL; sc.max_bytes_per_call = (uint32_t)niq; sc.device = 0;
  sdrfm_spectrum_t* sp = NULL;
  CHECK(sdrfm_spectrum_create(&sc, &sp));
  float* power = (float*)malloc(sizeof(float) * nfft);
  uint32_t frames = 0;
  CHECK(sdrfm_spectrum_process_batch(sp, iq, 0, (uint32_t)niq, power, nfft, &frames, 0));
  FILE* fo = fopen(argv[2], "wb");
  fwrite(power, sizeof(float), nfft, fo);
  fclose(fo);
  sdrfm_spectrum_destroy(sp);

  /* ---- WBFM: the buffer handed over piece by piece, one reused buffer ---- */
  sdrfm_wbfm_config wc;
  memset(&wc, 0, sizeof wc);
  wc.struct_size = sizeof wc; wc.n_streams = 1;
  wc.proto_taps = (uint32_t)(np_ / sizeof(float)); wc.proto_coeffs = proto;
  wc.resamp_taps = (uint32_t)(ng / sizeof(float)); wc.resamp_up = 6; wc.resamp_down = 25; wc.resamp_coeffs = resamp;
  wc.max_bytes_per_call = buff_bytes; wc.device = 0;
  sdrfm_wbfm_t* wb = NULL;
  CHECK(sdrfm_wbfm_create(&wc, &wb));
  unsigned char* buff = (unsigned char*)malloc(buff_bytes);          /* CommItf.buff: reused for every transfer */
  uint32_t cap = 0;
  CHECK(sdrfm_wbfm_audio_count(wb, buff_bytes, &cap));
  cap += 2;
  float* audio = (float*)malloc(sizeof(float) * SDRFM_WBFM_BANDS * cap);
  fo = fopen(argv[3], "wb");
  unsigned long total = 0;
  for (size_t pos = 0; pos < niq; pos += buff_bytes) {
    const uint32_t len = (uint32_t)((niq - pos < buff_bytes) ? niq - pos : buff_bytes);
    memcpy(buff, iq + pos, len);
    uint32_t n = 0;
    CHECK(sdrfm_wbfm_process_batch(wb, buff, len, len, audio, cap, &n, 0));
    for (int b = 0; b < SDRFM_WBFM_BANDS; ++b) fwrite(audio + (size_t)b * cap, sizeof(float), n, fo);
    total += n;
    memset(buff, 0xEE, buff_bytes);                                   /* the library is done with buff when it returns */
  }
  fclose(fo);
  sdrfm_wbfm_destroy(wb);
  printf("spectrum: %u frames of %u points; wbfm: %lu audio samples per band (%s)\n", frames, nfft, total, "16 bands");
  free(iq); free(proto); free(resamp); free(power); free(buff); free(audio);
  return 0;
}
