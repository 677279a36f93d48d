/*
 * sdrfm_wbfm_oracle.c — scalar C99 restatement of the multi-channel WBFM path (BASELINE configs[4]):
 * 16-band critically-sampled polyphase channelizer + per-band FM discriminator + rational L/M audio resampler.
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference has no DSP code (see sdrfm_oracle.h); build-defined spec:
 *
 *   x[n]    = (I - 127.5, Q - 127.5)
 *   u_r[t]  = sum_q p[r + NB*q] * x[t*NB + (NB-1) - r - NB*q]      r = 0..NB-1, q = 0..P/NB-1, x[n<0] = 0
 *             fp32 fmaf chain, oldest sample first (q descending)
 *   c_b[t]  = sum_r u_r[t] * exp(+j*2*pi*b*r/NB)                    NB = 16, by the radix-2 DIT graph below (bit-reversed
 *             input, stages m = 2,4,8,16; butterfly T = w*B with T.re = fmaf(w.re,B.re,-(w.im*B.im)),
 *             T.im = fmaf(w.re,B.im, w.im*B.re); lo = A + T; hi = A - T; twiddles = the float table W16)
 *   d_b[t]  = K3(c_b[t], c_b[t-1])                                  same discriminator as the narrow-band path, c_b[-1] = 0
 *   a_b[j]  = sum_i g[phi + L*i] * d_b[n_j - i],  n_j = floor(j*M/L), phi = (j*M) mod L, d[<0] = 0
 *             fmaf chain, oldest first (i descending); output j exists once d_b[n_j] exists
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NB 16
/* exp(+j*2*pi*t/16), t = 0..7, rounded to float */
static const float W16_RE[8] = {1.0f, 0x1.d906bcp-1f, 0x1.6a09e6p-1f, 0x1.87de2ap-2f, 0.0f, -0x1.87de2ap-2f, -0x1.6a09e6p-1f, -0x1.d906bcp-1f};
static const float W16_IM[8] = {0.0f, 0x1.87de2ap-2f, 0x1.6a09e6p-1f, 0x1.d906bcp-1f, 1.0f, 0x1.d906bcp-1f, 0x1.6a09e6p-1f, 0x1.87de2ap-2f};

typedef struct sdrfm_wbfm_oracle {
  uint32_t P, Tg, L, M;
  float *p, *g;
  float *hx_re, *hx_im;           /* last P-1 inputs */
  float cp_re[NB], cp_im[NB];     /* c_b[t-1] */
  float* hd;                      /* [NB][HD] last HD = ceil(Tg/L) discriminator outputs per band */
  uint32_t HD;
  uint32_t phase_x;               /* inputs since the last channelizer output, 0..NB-1 */
  uint64_t n_d, n_a;              /* discriminator outputs / audio outputs produced so far (per band) */
} sdrfm_wbfm_oracle;

sdrfm_wbfm_oracle* sdrfm_wbfm_oracle_create(uint32_t P, const float* p, uint32_t Tg, uint32_t L, uint32_t M, const float* g) {
  if (!P || P % NB || !p || !Tg || !L || !M || !g) return NULL;
  sdrfm_wbfm_oracle* o = (sdrfm_wbfm_oracle*)calloc(1, sizeof(*o));
  o->P = P; o->Tg = Tg; o->L = L; o->M = M; o->HD = (Tg + L - 1) / L;
  o->p = (float*)malloc(4 * P); o->g = (float*)malloc(4 * Tg);
  memcpy(o->p, p, 4 * P); memcpy(o->g, g, 4 * Tg);
  o->hx_re = (float*)calloc(P, 4); o->hx_im = (float*)calloc(P, 4);
  o->hd = (float*)calloc((size_t)NB * o->HD, 4);
  return o;
}
void sdrfm_wbfm_oracle_destroy(sdrfm_wbfm_oracle* o) {
  if (!o) return;
  free(o->p); free(o->g); free(o->hx_re); free(o->hx_im); free(o->hd); free(o);
}

static void fft16_dit(float* re, float* im) { /* in place, natural-order output, inverse-sign twiddles */
  static const int rev[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  float ar[16], ai[16];
  for (int k = 0; k < 16; ++k) { ar[k] = re[rev[k]]; ai[k] = im[rev[k]]; }
  for (int m = 2; m <= 16; m <<= 1) {
    const int half = m / 2, step = 16 / m;
    for (int g0 = 0; g0 < 16; g0 += m)
      for (int t = 0; t < half; ++t) {
        const float wr = W16_RE[t * step], wi = W16_IM[t * step];
        const float br = ar[g0 + t + half], bi = ai[g0 + t + half];
        const float tr = fmaf(wr, br, -(wi * bi));
        const float ti = fmaf(wr, bi, wi * br);
        const float xr = ar[g0 + t], xi = ai[g0 + t];
        ar[g0 + t] = xr + tr; ai[g0 + t] = xi + ti;
        ar[g0 + t + half] = xr - tr; ai[g0 + t + half] = xi - ti;
      }
  }
  for (int k = 0; k < 16; ++k) { re[k] = ar[k]; im[k] = ai[k]; }
}

static float disc(float yr, float yi, float pr, float pi) {
  const float re = fmaf(yr, pr, yi * pi);
  const float im = yi * pr - yr * pi;
  return (re == 0.0f && im == 0.0f) ? 0.0f : atan2f(im, re);
}

/* audio: [NB][audio_stride]; returns audio samples per band produced by this call, -1 on error */
long sdrfm_wbfm_oracle_process(sdrfm_wbfm_oracle* o, const uint8_t* iq, size_t nbytes, float* audio, size_t audio_stride) {
  if (!o || (nbytes & 1u) || (nbytes && !iq)) return -1;
  const uint32_t P = o->P, Q = P / NB, Tg = o->Tg, L = o->L, M = o->M, HD = o->HD;
  const size_t N = nbytes / 2, H = P - 1;
  const size_t Tn = (o->phase_x + N) / NB;                  /* channelizer outputs this call */
  float* wr = (float*)malloc(4 * (H + N + 1));
  float* wi = (float*)malloc(4 * (H + N + 1));
  float* wd = (float*)malloc(4 * (size_t)NB * (HD + Tn + 1));   /* per band [HD history | Tn new] */
  memcpy(wr, o->hx_re, 4 * H); memcpy(wi, o->hx_im, 4 * H);
  for (int b = 0; b < NB; ++b) memcpy(wd + (size_t)b * (HD + Tn), o->hd + (size_t)b * HD, 4 * HD);
  for (size_t n = 0; n < N; ++n) { wr[H + n] = (float)iq[2 * n] - 127.5f; wi[H + n] = (float)iq[2 * n + 1] - 127.5f; }
  for (size_t t = 0; t < Tn; ++t) {
    const size_t e = (t + 1) * (size_t)NB - 1 - o->phase_x;   /* chunk index of the newest input of step t */
    float ur[NB], ui[NB];
    for (int r = 0; r < NB; ++r) {
      float ar = 0.0f, ai = 0.0f;
      for (uint32_t q = Q; q-- > 0;) {                        /* oldest first */
        const size_t idx = H + e - (size_t)r - (size_t)NB * q; /* work index of x[t*NB + NB-1 - r - NB*q] */
        ar = fmaf(o->p[r + NB * q], wr[idx], ar);
        ai = fmaf(o->p[r + NB * q], wi[idx], ai);
      }
      ur[r] = ar; ui[r] = ai;
    }
    fft16_dit(ur, ui);
    for (int b = 0; b < NB; ++b) {
      wd[(size_t)b * (HD + Tn) + HD + t] = disc(ur[b], ui[b], o->cp_re[b], o->cp_im[b]);
      o->cp_re[b] = ur[b]; o->cp_im[b] = ui[b];
    }
  }
  /* resampler */
  const uint64_t nd_new = o->n_d + Tn;
  uint64_t j = o->n_a;
  long produced = 0;
  while ((j * M) / L < nd_new) {
    const uint64_t nj = (j * M) / L;
    const uint32_t phi = (uint32_t)((j * M) % L);
    const int imax = (int)((Tg - 1 - phi) / L);
    if ((size_t)produced >= audio_stride) { free(wr); free(wi); free(wd); return -1; }
    for (int b = 0; b < NB; ++b) {
      const float* d = wd + (size_t)b * (HD + Tn) + HD;       /* d[0] = first NEW output = global index o->n_d */
      float acc = 0.0f;
      for (int i = imax; i >= 0; --i) {
        const long li = (long)(nj - o->n_d) - i;              /* call-relative index, >= -HD by construction */
        acc = fmaf(o->g[phi + L * (uint32_t)i], d[li], acc);
      }
      audio[(size_t)b * audio_stride + produced] = acc;
    }
    ++produced; ++j;
  }
  /* carry */
  if (H) { memmove(o->hx_re, wr + N, 4 * H); memmove(o->hx_im, wi + N, 4 * H); }
  for (int b = 0; b < NB; ++b) memmove(o->hd + (size_t)b * HD, wd + (size_t)b * (HD + Tn) + Tn, 4 * HD);
  o->phase_x = (uint32_t)((o->phase_x + N) % NB);
  o->n_d = nd_new; o->n_a = j;
  free(wr); free(wi); free(wd);
  return produced;
}
