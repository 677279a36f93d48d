/*
 * replay_main.c — plain-C host program: replays a captured RTL-SDR byte file through the reference's buffer hand-off
 * and demodulates it with libsdrfm.so (hand-written HIP kernels behind the C-ABI of include/sdrfm.h).
 *
 * It stands in for the firmware's superloop (src/main.c:72-80) with the consumer hook that the reference leaves
 * commented out (src/main.c:76-79) filled in.  The FSM below mirrors USBH_RTLSDR_Process
 * (Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Src/usbh_rtlsdr.c:1058-1101): START submits one URB into the single
 * reused buffer, WAIT polls for completion, COMPLETE is where the consumer runs before the FSM re-arms.
 *
 *   build: gcc -O2 -std=c99 -Iinclude examples/replay_main.c -Lstm32f7-rtlsdr_amd/csrc -lsdrfm -Wl,-rpath,'$ORIGIN/../stm32f7-rtlsdr_amd/csrc' -o examples/replay_main
 *   usage: replay_main <iq.u8> <audio.f32> <taps_h.f32> <taps_g.f32> [buffSize=512]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdrfm.h"

typedef enum { RTLSDR_XFER_START = 0, RTLSDR_XFER_WAIT, RTLSDR_XFER_COMPLETE } xfer_state; /* usbh_rtlsdr.h:156-162 */

typedef struct {
  uint8_t* buff;          /* CommItf.buff     (usbh_rtlsdr.h:165-173): ONE buffer, reused for every URB */
  uint32_t buffSize;      /* CommItf.buffSize: multiple of 512, <= 65535 (uint16_t URB length, usbh_ioreq.c:218-233) */
  xfer_state xferState;
  uint32_t lastXferSize;  /* USBH_LL_GetLastXferSize */
  FILE* src;
} fake_rtlsdr;

static int fake_process(fake_rtlsdr* d) { /* one call of the class BgndProcess */
  switch (d->xferState) {
    case RTLSDR_XFER_START: d->xferState = RTLSDR_XFER_WAIT; break;
    case RTLSDR_XFER_WAIT:
      d->lastXferSize = (uint32_t)fread(d->buff, 1, d->buffSize, d->src);
      d->lastXferSize &= ~1u;
      d->xferState = RTLSDR_XFER_COMPLETE;
      break;
    case RTLSDR_XFER_COMPLETE: d->xferState = RTLSDR_XFER_START; break;
  }
  return 0;
}

static float* read_f32(const char* path, uint32_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  float* v = (float*)malloc((size_t)sz);
  *n = (uint32_t)(fread(v, 4, (size_t)sz / 4, f));
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s iq.u8 audio.f32 h.f32 g.f32 [buffSize]\n", argv[0]); return 2; }
  fake_rtlsdr dev;
  memset(&dev, 0, sizeof(dev));
  dev.buffSize = argc > 5 ? (uint32_t)atoi(argv[5]) : 512u;
  if (dev.buffSize == 0 || dev.buffSize % 512 || dev.buffSize > 65535) { fprintf(stderr, "bad buffSize\n"); return 2; }
  dev.buff = (uint8_t*)malloc(dev.buffSize);
  dev.src = fopen(argv[1], "rb");
  FILE* out = fopen(argv[2], "wb");
  uint32_t T = 0, Ta = 0;
  float* h = read_f32(argv[3], &T);
  float* g = read_f32(argv[4], &Ta);
  if (!dev.src || !out || !h || !g) { fprintf(stderr, "cannot open inputs\n"); return 2; }

  sdrfm_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = sizeof(cfg);
  cfg.n_streams = 1;
  cfg.fir_taps = T; cfg.fir_decim = 10; cfg.fir_coeffs = h;       /* 2.4 MS/s -> 240 kS/s (usbh_rtlsdr.c:898) */
  cfg.audio_taps = Ta; cfg.audio_decim = 5; cfg.audio_coeffs = g;  /* -> 48 kHz */
  cfg.max_bytes_per_call = 65536;
  sdrfm_t* fm = NULL;
  int st = sdrfm_create(&cfg, &fm);
  if (st != SDRFM_OK) { fprintf(stderr, "sdrfm_create: %s\n", sdrfm_strerror(st)); return 1; }

  float audio[1400];
  unsigned long total = 0;
  for (;;) {                                   /* while (1) { USBH_Process(&hUSBHost); ... }  (src/main.c:72-80) */
    fake_process(&dev);
    if (dev.xferState == RTLSDR_XFER_COMPLETE) {              /* the hook of src/main.c:76-79 */
      if (dev.lastXferSize == 0) break;
      uint32_t n = 0;
      st = sdrfm_process(fm, dev.buff, dev.lastXferSize, audio, 1400, &n);
      if (st != SDRFM_OK) { fprintf(stderr, "sdrfm_process: %s\n", sdrfm_strerror(st)); return 1; }
      fwrite(audio, sizeof(float), n, out);
      total += n;
      memset(dev.buff, 0xEE, dev.buffSize);    /* the FSM re-arms the same buffer: nothing may still reference it */
    }
  }
  fprintf(stderr, "kernel: %s, %lu audio samples\n", sdrfm_kernel_name(fm), total);
  sdrfm_destroy(fm);
  fclose(out); fclose(dev.src); free(dev.buff); free(h); free(g);
  return 0;
}
