/*
 * sdrfm_oracle.c — scalar C99 restatement of the frozen IQ -> FM-audio spec.  TEST INFRASTRUCTURE ONLY
 * (see sdrfm_oracle.h: "PARITY UNPINNED" — the reference has no DSP code to restate).
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -mfma -fPIC -shared   (see Makefile)
 *   -ffp-contract=off : the compiler may not fuse or re-associate anything; every fused multiply-add below is an
 *                       explicit fmaf(), every other operation rounds once.
 *   -mfma             : fmaf() is the hardware instruction (one rounding) instead of the libm software routine; the
 *                       result is identical, only faster.
 *
 * Stage by stage (n = IQ sample index since reset, m = index at the 1/D rate, j = index at the audio rate):
 *   K1  x[n]  = ((float)I - 127.5f, (float)Q - 127.5f)          exact in fp32.  Input format = RTL2832 bulk bytes as
 *                                                                stored in RTLSDR_CommItfTypedef.buff (usbh_rtlsdr.h:165-173)
 *   K2  y[m]  = sum_k h[k] x[(m+1)D-1-k],  x[n<0] = 0           "one output per D inputs, zero initial state" — the
 *                                                                arm_fir_decimate_f32 convention whose header the reference
 *                                                                vendors (CMSIS/core/arm_math.h:3291-3331).  Accumulation:
 *                                                                acc = fmaf(h[k], x, acc), OLDEST sample first (k = T-1 .. 0).
 *   K3  p     = y[m] * conj(y[m-1]),  y[-1] = 0
 *       re    = fmaf(yr, pr, yi*pi);  im = yi*pr - yr*pi   (two rounded products, one subtraction)
 *       d[m]  = (re == 0 && im == 0) ? 0 : atan2f(im, re)
 *   K4  a[j]  = sum_k g[k] d[(j+1)Da-1-k],  d[m<0] = 0          same convention and accumulation order as K2.
 */
#include "sdrfm_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct sdrfm_oracle {
  uint32_t T, D, Ta, Da;
  float* h;          /* T  */
  float* g;          /* Ta */
  /* streaming state (fp32 path) */
  float* hx_re;      /* last T-1 inputs, oldest first */
  float* hx_im;
  float  yp_re, yp_im;
  float* hd;         /* last Ta-1 discriminator outputs, oldest first */
  uint32_t phase_x;  /* inputs since the last y output, 0..D-1  */
  uint32_t phase_d;  /* y outputs since the last audio output, 0..Da-1 */
  /* streaming state (fp64 shadow path, independent) */
  double* hx64_re; double* hx64_im; double yp64_re, yp64_im; double* hd64; uint32_t phase_x64, phase_d64;
  /* intermediates of the last fp32 call */
  float* last_y; float* last_d; size_t last_ny, last_cap;
};

sdrfm_oracle* sdrfm_oracle_create(uint32_t T, uint32_t D, const float* h, uint32_t Ta, uint32_t Da, const float* g) {
  if (!T || !D || !Ta || !Da || !h || !g) return NULL;
  sdrfm_oracle* o = (sdrfm_oracle*)calloc(1, sizeof(*o));
  if (!o) return NULL;
  o->T = T; o->D = D; o->Ta = Ta; o->Da = Da;
  o->h = (float*)malloc(sizeof(float) * T);
  o->g = (float*)malloc(sizeof(float) * Ta);
  o->hx_re = (float*)calloc(T, sizeof(float));
  o->hx_im = (float*)calloc(T, sizeof(float));
  o->hd = (float*)calloc(Ta, sizeof(float));
  o->hx64_re = (double*)calloc(T, sizeof(double));
  o->hx64_im = (double*)calloc(T, sizeof(double));
  o->hd64 = (double*)calloc(Ta, sizeof(double));
  if (!o->h || !o->g || !o->hx_re || !o->hx_im || !o->hd || !o->hx64_re || !o->hx64_im || !o->hd64) {
    sdrfm_oracle_destroy(o);
    return NULL;
  }
  memcpy(o->h, h, sizeof(float) * T);
  memcpy(o->g, g, sizeof(float) * Ta);
  return o;
}

void sdrfm_oracle_destroy(sdrfm_oracle* o) {
  if (!o) return;
  free(o->h); free(o->g); free(o->hx_re); free(o->hx_im); free(o->hd);
  free(o->hx64_re); free(o->hx64_im); free(o->hd64); free(o->last_y); free(o->last_d);
  free(o);
}

void sdrfm_oracle_reset(sdrfm_oracle* o) {
  memset(o->hx_re, 0, sizeof(float) * o->T); memset(o->hx_im, 0, sizeof(float) * o->T);
  memset(o->hd, 0, sizeof(float) * o->Ta);
  memset(o->hx64_re, 0, sizeof(double) * o->T); memset(o->hx64_im, 0, sizeof(double) * o->T);
  memset(o->hd64, 0, sizeof(double) * o->Ta);
  o->yp_re = o->yp_im = 0.0f; o->yp64_re = o->yp64_im = 0.0;
  o->phase_x = o->phase_d = o->phase_x64 = o->phase_d64 = 0;
  o->last_ny = 0;
}

long sdrfm_oracle_process(sdrfm_oracle* o, const uint8_t* iq, size_t nbytes, float* audio, size_t audio_cap) {
  if (!o || (nbytes & 1u) || (nbytes && !iq)) return -1;
  const uint32_t T = o->T, D = o->D, Ta = o->Ta, Da = o->Da;
  const size_t N = nbytes / 2;
  const size_t M = (o->phase_x + N) / D;            /* y outputs this call     */
  const size_t A = (o->phase_d + M) / Da;           /* audio outputs this call */
  if (A > audio_cap || (A && !audio)) return -1;

  /* work = [history (T-1) | converted chunk], so index (T-1)+n is chunk sample n */
  const size_t H = T - 1, Hd = Ta - 1;
  float* wr = (float*)malloc(sizeof(float) * (H + N + 1));
  float* wi = (float*)malloc(sizeof(float) * (H + N + 1));
  float* wd = (float*)malloc(sizeof(float) * (Hd + M + 1));
  if (o->last_cap < M + 1) {
    free(o->last_y); free(o->last_d);
    o->last_cap = M + 1;
    o->last_y = (float*)malloc(sizeof(float) * 2 * o->last_cap);
    o->last_d = (float*)malloc(sizeof(float) * o->last_cap);
  }
  if (!wr || !wi || !wd || !o->last_y || !o->last_d) { free(wr); free(wi); free(wd); return -1; }
  memcpy(wr, o->hx_re, sizeof(float) * H); memcpy(wi, o->hx_im, sizeof(float) * H);
  memcpy(wd, o->hd, sizeof(float) * Hd);

  /* K1: DC shift */
  for (size_t n = 0; n < N; ++n) {
    wr[H + n] = (float)iq[2 * n] - 127.5f;
    wi[H + n] = (float)iq[2 * n + 1] - 127.5f;
  }
  /* K2 + K3 */
  float pr = o->yp_re, pi = o->yp_im;
  for (size_t m = 0; m < M; ++m) {
    /* newest sample of output m is chunk index e = (m+1)*D - 1 - phase_x; window = work[e .. e+T-1] oldest first */
    const size_t e = (m + 1) * (size_t)D - 1 - o->phase_x;
    const float* xr = wr + e; const float* xi = wi + e;   /* == chunk index e-(T-1) shifted by H */
    float ar = 0.0f, ai = 0.0f;
    for (uint32_t j = 0; j < T; ++j) {
      const float c = o->h[T - 1 - j];
      ar = fmaf(c, xr[j], ar);
      ai = fmaf(c, xi[j], ai);
    }
    const float re = fmaf(ar, pr, ai * pi);
    const float im = ai * pr - ar * pi;             /* two rounded products: exactly 0 for y[m] == y[m-1] */
    const float d = (re == 0.0f && im == 0.0f) ? 0.0f : atan2f(im, re);
    wd[Hd + m] = d;
    o->last_y[2 * m] = ar; o->last_y[2 * m + 1] = ai; o->last_d[m] = d;
    pr = ar; pi = ai;
  }
  o->last_ny = M;
  /* K4 */
  for (size_t j = 0; j < A; ++j) {
    const size_t e = (j + 1) * (size_t)Da - 1 - o->phase_d;  /* newest d of audio output j, call-relative */
    const float* dd = wd + e;
    float acc = 0.0f;
    for (uint32_t k = 0; k < Ta; ++k) acc = fmaf(o->g[Ta - 1 - k], dd[k], acc);
    audio[j] = acc;
  }
  /* carry state */
  if (H) { memmove(o->hx_re, wr + N, sizeof(float) * H); memmove(o->hx_im, wi + N, sizeof(float) * H); }
  if (Hd) memmove(o->hd, wd + M, sizeof(float) * Hd);
  o->yp_re = pr; o->yp_im = pi;
  o->phase_x = (uint32_t)((o->phase_x + N) % D);
  o->phase_d = (uint32_t)((o->phase_d + M) % Da);
  free(wr); free(wi); free(wd);
  return (long)A;
}

long sdrfm_oracle_process_f64(sdrfm_oracle* o, const uint8_t* iq, size_t nbytes, double* audio, size_t audio_cap) {
  if (!o || (nbytes & 1u) || (nbytes && !iq)) return -1;
  const uint32_t T = o->T, D = o->D, Ta = o->Ta, Da = o->Da;
  const size_t N = nbytes / 2, M = (o->phase_x64 + N) / D, A = (o->phase_d64 + M) / Da;
  if (A > audio_cap || (A && !audio)) return -1;
  const size_t H = T - 1, Hd = Ta - 1;
  double* wr = (double*)malloc(sizeof(double) * (H + N + 1));
  double* wi = (double*)malloc(sizeof(double) * (H + N + 1));
  double* wd = (double*)malloc(sizeof(double) * (Hd + M + 1));
  if (!wr || !wi || !wd) { free(wr); free(wi); free(wd); return -1; }
  memcpy(wr, o->hx64_re, sizeof(double) * H); memcpy(wi, o->hx64_im, sizeof(double) * H);
  memcpy(wd, o->hd64, sizeof(double) * Hd);
  for (size_t n = 0; n < N; ++n) { wr[H + n] = (double)iq[2 * n] - 127.5; wi[H + n] = (double)iq[2 * n + 1] - 127.5; }
  double pr = o->yp64_re, pi = o->yp64_im;
  for (size_t m = 0; m < M; ++m) {
    const size_t e = (m + 1) * (size_t)D - 1 - o->phase_x64;
    double ar = 0.0, ai = 0.0;
    for (uint32_t j = 0; j < T; ++j) { const double c = (double)o->h[T - 1 - j]; ar += c * wr[e + j]; ai += c * wi[e + j]; }
    const double re = ar * pr + ai * pi, im = ai * pr - ar * pi;
    wd[Hd + m] = (re == 0.0 && im == 0.0) ? 0.0 : atan2(im, re);
    pr = ar; pi = ai;
  }
  for (size_t j = 0; j < A; ++j) {
    const size_t e = (j + 1) * (size_t)Da - 1 - o->phase_d64;
    double acc = 0.0;
    for (uint32_t k = 0; k < Ta; ++k) acc += (double)o->g[Ta - 1 - k] * wd[e + k];
    audio[j] = acc;
  }
  if (H) { memmove(o->hx64_re, wr + N, sizeof(double) * H); memmove(o->hx64_im, wi + N, sizeof(double) * H); }
  if (Hd) memmove(o->hd64, wd + M, sizeof(double) * Hd);
  o->yp64_re = pr; o->yp64_im = pi;
  o->phase_x64 = (uint32_t)((o->phase_x64 + N) % D);
  o->phase_d64 = (uint32_t)((o->phase_d64 + M) % Da);
  free(wr); free(wi); free(wd);
  return (long)A;
}

size_t sdrfm_oracle_last_stage(const sdrfm_oracle* o, float* y_out, float* d_out, size_t cap_y) {
  if (!o) return 0;
  size_t n = o->last_ny < cap_y ? o->last_ny : cap_y;
  if (y_out) memcpy(y_out, o->last_y, sizeof(float) * 2 * n);
  if (d_out) memcpy(d_out, o->last_d, sizeof(float) * n);
  return o->last_ny;
}

long sdrfm_oracle_process_many(sdrfm_oracle** os, uint32_t n, const uint8_t* iq, size_t iq_stride, size_t nbytes,
                               float* audio, size_t audio_stride) {
  long a = 0;
  for (uint32_t s = 0; s < n; ++s) {
    a = sdrfm_oracle_process(os[s], iq + (size_t)s * iq_stride, nbytes, audio + (size_t)s * audio_stride, audio_stride);
    if (a < 0) return a;
  }
  return a;
}
