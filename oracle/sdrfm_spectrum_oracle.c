/*
 * sdrfm_spectrum_oracle.c — scalar C99 restatement of the spectrum view of an IQ buffer (the reference's
 * own next task, README.md:29 "Perform some FFT on the samples to check what we are receiving").
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference contains no FFT code (arm_math.h is prototypes only and is not
 * included by any translation unit); this file DEFINES the arithmetic, it does not follow reference code.  Build-defined spec:
 *
 *   frames   f = 0 .. F-1, F = floor(nsamples / N), consecutive and non-overlapping; a tail shorter than N is ignored
 *   x_f[n]   = ((I - 127.5) * w[n], (Q - 127.5) * w[n])                  n = 0..N-1, the byte pair 2*(f*N + n)
 *   X_f[k]   = sum_n x_f[n] * exp(-j*2*pi*k*n/N)                          by ONE fixed radix-2 DIT graph: input in
 *              bit-reversed order, stages m = 2,4,..,N; butterfly j of a stage (half = m/2, pos = j mod half):
 *              A = X[i], B = X[i+half], W = tw[pos * N/m],  T.re = fmaf(W.re,B.re,-(W.im*B.im)), T.im = fmaf(W.re,B.im,W.im*B.re),
 *              X[i] = A + T, X[i+half] = A - T;   tw[t] = ((float)cos(2 pi t/N), (float)(-sin(2 pi t/N))), t < N/2, from double
 *   P_f[k]   = fmaf(X.re, X.re, X.im * X.im)
 *   S[k]     = (..((0 + P_0[k]) + P_1[k]) + ..) + P_{F-1}[k]              fp32, frame order
 *   out[i]   = S[(i + N/2) mod N] * (1.0f / (float)F)                     DC in the middle (fftshift); all 0 when F = 0
 *   default window: periodic Hann, w[n] = (float)(0.5 - 0.5*cos(2 pi n/N)) from double
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct sdrfm_spectrum_oracle {
  uint32_t N, logn;
  float *tw_re, *tw_im, *win, *xr, *xi, *acc;
} sdrfm_spectrum_oracle;

static const double TWO_PI = 6.283185307179586476925286766559;

void sdrfm_spectrum_oracle_destroy(sdrfm_spectrum_oracle* o) {
  if (!o) return;
  free(o->tw_re); free(o->tw_im); free(o->win); free(o->xr); free(o->xi); free(o->acc);
  free(o);
}

sdrfm_spectrum_oracle* sdrfm_spectrum_oracle_create(uint32_t nfft, const float* window) {
  uint32_t logn = 0;
  while ((1u << logn) < nfft) ++logn;
  if (nfft < 2 || (1u << logn) != nfft || nfft > (1u << 20)) return NULL;
  sdrfm_spectrum_oracle* o = (sdrfm_spectrum_oracle*)calloc(1, sizeof(*o));
  o->N = nfft; o->logn = logn;
  o->tw_re = (float*)malloc(sizeof(float) * (nfft / 2)); o->tw_im = (float*)malloc(sizeof(float) * (nfft / 2));
  o->win = (float*)malloc(sizeof(float) * nfft);
  o->xr = (float*)malloc(sizeof(float) * nfft); o->xi = (float*)malloc(sizeof(float) * nfft); o->acc = (float*)malloc(sizeof(float) * nfft);
  for (uint32_t t = 0; t < nfft / 2; ++t) {
    o->tw_re[t] = (float)cos(TWO_PI * (double)t / (double)nfft);
    o->tw_im[t] = (float)(-sin(TWO_PI * (double)t / (double)nfft));
  }
  for (uint32_t n = 0; n < nfft; ++n)
    o->win[n] = window ? window[n] : (float)(0.5 - 0.5 * cos(TWO_PI * (double)n / (double)nfft));
  return o;
}

static uint32_t bitrev(uint32_t v, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t b = 0; b < bits; ++b) r |= ((v >> b) & 1u) << (bits - 1 - b);
  return r;
}

/* power[N] <- averaged spectrum of iq[0..nbytes); returns the number of frames, or -1 on bad arguments */
long sdrfm_spectrum_oracle_process(sdrfm_spectrum_oracle* o, const uint8_t* iq, uint64_t nbytes, float* power) {
  if (!o || !power || (nbytes & 1u) || (nbytes && !iq)) return -1;
  const uint32_t N = o->N;
  const uint64_t F = (nbytes / 2) / N;
  for (uint32_t k = 0; k < N; ++k) o->acc[k] = 0.0f;
  for (uint64_t f = 0; f < F; ++f) {
    const uint8_t* b = iq + 2 * f * N;
    for (uint32_t n = 0; n < N; ++n) {
      const uint32_t u = bitrev(n, o->logn);
      o->xr[u] = ((float)b[2 * n] - 127.5f) * o->win[n];
      o->xi[u] = ((float)b[2 * n + 1] - 127.5f) * o->win[n];
    }
    for (uint32_t s = 1; s <= o->logn; ++s) {
      const uint32_t half = 1u << (s - 1);
      for (uint32_t j = 0; j < N / 2; ++j) {
        const uint32_t pos = j & (half - 1), i = ((j >> (s - 1)) << s) + pos;
        const float wr = o->tw_re[pos << (o->logn - s)], wi = o->tw_im[pos << (o->logn - s)];
        const float ar = o->xr[i], ai = o->xi[i], br = o->xr[i + half], bi = o->xi[i + half];
        const float tr = fmaf(wr, br, -(wi * bi));
        const float ti = fmaf(wr, bi, wi * br);
        o->xr[i] = ar + tr; o->xi[i] = ai + ti;
        o->xr[i + half] = ar - tr; o->xi[i + half] = ai - ti;
      }
    }
    for (uint32_t k = 0; k < N; ++k) o->acc[k] = o->acc[k] + fmaf(o->xr[k], o->xr[k], o->xi[k] * o->xi[k]);
  }
  const float inv = F ? 1.0f / (float)F : 0.0f;
  for (uint32_t i = 0; i < N; ++i) power[i] = o->acc[(i + N / 2) & (N - 1)] * inv;
  return (long)F;
}
