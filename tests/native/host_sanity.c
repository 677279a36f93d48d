/* host_sanity.c — the plain-C host pieces (RTL2832 / E4000 arithmetic, PCM sink) and the three oracles exercised under
 * AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool).  Built and run by
 * tests/test_host_sanitizers.py; exits 0 and prints "ok" when every check holds and the sanitizers stay silent. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sdrfm.h"
#include "../../oracle/sdrfm_oracle.h"

typedef struct sdrfm_wbfm_oracle sdrfm_wbfm_oracle;
sdrfm_wbfm_oracle* sdrfm_wbfm_oracle_create(uint32_t P, const float* p, uint32_t Tg, uint32_t L, uint32_t M, const float* g);
void sdrfm_wbfm_oracle_destroy(sdrfm_wbfm_oracle* o);
long sdrfm_wbfm_oracle_process(sdrfm_wbfm_oracle* o, const uint8_t* iq, size_t nbytes, float* audio, size_t band_cap);
typedef struct sdrfm_spectrum_oracle sdrfm_spectrum_oracle;
sdrfm_spectrum_oracle* sdrfm_spectrum_oracle_create(uint32_t nfft, const float* window);
void sdrfm_spectrum_oracle_destroy(sdrfm_spectrum_oracle* o);
long sdrfm_spectrum_oracle_process(sdrfm_spectrum_oracle* o, const uint8_t* iq, uint64_t nbytes, float* power);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint8_t next_byte(void) { rng ^= rng >> 12; rng ^= rng << 25; rng ^= rng >> 27; return (uint8_t)((rng * 0x2545F4914F6CDD1Dull) >> 56); }

int main(void) {
  /* --- rtlctl.c ------------------------------------------------------------------------------------------- */
  static const int fir[16] = {-54, -36, -41, -40, -32, -14, 14, 53, 101, 156, 215, 273, 327, 372, 404, 421};
  uint8_t img[20];
  CHECK(sdrfm_rtl_pack_fir(fir, img) == SDRFM_OK && img[0] == 0xca && img[19] == 0xa5);
  CHECK(sdrfm_rtl_pack_fir(NULL, img) == SDRFM_EINVAL);
  uint32_t r = 0, rr = 0; double rate = 0;
  CHECK(sdrfm_rtl_resampler(2400000, 28800000, &r, &rr, &rate) == SDRFM_OK && r == 0x03000000u);
  CHECK(sdrfm_rtl_resampler(100, 28800000, &r, &rr, &rate) == SDRFM_EINVAL);
  sdrfm_e4k_pll pll;
  CHECK(sdrfm_e4k_pll_params(28800000, 99700000, &pll) == SDRFM_OK && pll.flo == 99699993u && pll.z == 110 && pll.x == 50972);
  CHECK(sdrfm_e4k_pll_params(28800000, 2200000000u, &pll) == SDRFM_OK && pll.r == 2);
  CHECK(sdrfm_e4k_pll_params(1, 99700000, &pll) == SDRFM_EINVAL);
  /* --- pcm_sink.c: saturation at both rails, state carried, exact-size buffers ------------------------------- */
  enum { NA = 257 };
  float* a = (float*)malloc(sizeof(float) * NA);
  int16_t* pcm = (int16_t*)malloc(sizeof(int16_t) * 2 * NA);
  for (int i = 0; i < NA; ++i) a[i] = (i & 1) ? 1e9f : -1e9f;
  float st = 0.0f;
  CHECK(sdrfm_pcm_deemph_s16(a, NA, 1.0f, 1.0f, &st, pcm) == SDRFM_OK);
  CHECK(pcm[0] == -32768 && pcm[2] == 32767 && pcm[2 * NA - 1] == pcm[2 * NA - 2]);
  CHECK(sdrfm_pcm_deemph_s16(a, 0, 0.5f, 1.0f, &st, NULL) == SDRFM_OK);
  CHECK(sdrfm_pcm_deemph_s16(a, NA, 0.0f, 1.0f, &st, pcm) == SDRFM_EINVAL);
  free(a); free(pcm);
  /* --- the narrow-band oracle: ragged chunks == one shot, exact-size heap buffers ----------------------------- */
  enum { T = 64, D = 10, TA = 32, DA = 5, NS = 20011 };
  float h[T], g[TA];
  for (int k = 0; k < T; ++k) h[k] = 0.01f * (float)((k * 7) % 13 - 6);
  for (int k = 0; k < TA; ++k) g[k] = 0.02f * (float)((k * 5) % 11 - 5);
  uint8_t* iq = (uint8_t*)malloc(2 * NS);
  for (int i = 0; i < 2 * NS; ++i) iq[i] = next_byte();
  sdrfm_oracle* o1 = sdrfm_oracle_create(T, D, h, TA, DA, g);
  sdrfm_oracle* o2 = sdrfm_oracle_create(T, D, h, TA, DA, g);
  CHECK(o1 && o2);
  const size_t cap = NS / D / DA + 2;
  float* one = (float*)malloc(sizeof(float) * cap);
  float* two = (float*)malloc(sizeof(float) * cap);
  long n1 = sdrfm_oracle_process(o1, iq, 2 * NS, one, cap);
  long n2 = 0;
  for (size_t pos = 0; pos < 2 * (size_t)NS;) {
    size_t c = 2 * (size_t)(next_byte() % 97);                  /* 0 .. 192 bytes, zero-length calls included */
    if (pos + c > 2 * (size_t)NS) c = 2 * (size_t)NS - pos;
    const long k = sdrfm_oracle_process(o2, iq + pos, c, two + n2, cap - (size_t)n2);
    CHECK(k >= 0);
    n2 += k;
    pos += c ? c : 0;
    if (c == 0) {                                               /* make progress after an empty call */
      const long k2 = sdrfm_oracle_process(o2, iq + pos, 2, two + n2, cap - (size_t)n2);
      CHECK(k2 >= 0);
      n2 += k2; pos += 2;
    }
  }
  CHECK(n1 == n2 && n1 == NS / D / DA && memcmp(one, two, sizeof(float) * (size_t)n1) == 0);
  CHECK(sdrfm_oracle_process(o1, iq, 3, one, cap) < 0);          /* odd byte count is refused */
  sdrfm_oracle_destroy(o1); sdrfm_oracle_destroy(o2);
  free(one); free(two);
  /* --- WBFM oracle ----------------------------------------------------------------------------------------- */
  enum { P = 128, TG = 60 };
  float pp[P], gg[TG];
  for (int k = 0; k < P; ++k) pp[k] = 0.004f * (float)((k * 3) % 17 - 8);
  for (int k = 0; k < TG; ++k) gg[k] = 0.03f * (float)((k * 7) % 19 - 9);
  sdrfm_wbfm_oracle* w = sdrfm_wbfm_oracle_create(P, pp, TG, 6, 25, gg);
  CHECK(w);
  const size_t bcap = (NS / 16 + 2) * 6 / 25 + 2;
  float* wa = (float*)malloc(sizeof(float) * 16 * bcap);
  long nw = sdrfm_wbfm_oracle_process(w, iq, 2 * NS, wa, bcap);
  CHECK(nw > 0 && (size_t)nw <= bcap);
  for (long i = 0; i < nw; ++i) CHECK(isfinite(wa[i]) && fabsf(wa[i]) < 40.0f);
  sdrfm_wbfm_oracle_destroy(w); free(wa);
  /* --- spectrum oracle -------------------------------------------------------------------------------------- */
  sdrfm_spectrum_oracle* s = sdrfm_spectrum_oracle_create(256, NULL);
  CHECK(s && sdrfm_spectrum_oracle_create(100, NULL) == NULL);
  float* pw = (float*)malloc(sizeof(float) * 256);
  CHECK(sdrfm_spectrum_oracle_process(s, iq, 2 * NS, pw) == NS / 256);
  for (int i = 0; i < 256; ++i) CHECK(isfinite(pw[i]) && pw[i] >= 0.0f);
  CHECK(sdrfm_spectrum_oracle_process(s, iq, 100, pw) == 0 && pw[128] == 0.0f);
  sdrfm_spectrum_oracle_destroy(s); free(pw);
  free(iq);
  puts("ok");
  return 0;
}
