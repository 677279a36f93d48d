/*
 * siggen.c — deterministic synthetic RTL2832-style IQ byte streams for tests and bench.
 * Not part of the product path and not part of the oracle: it only manufactures INPUT bytes.
 *
 *   mode 0 "fm"      : wide-band FM test signal
 *                      phi[n] = phi[n-1] + 2*pi*(f_c + 75 kHz * a[n]) / Fs,
 *                      a[n]   = .5 sin(2pi 1000 t) + .3 sin(2pi 3100 t) + .2 sin(2pi 7300 t),
 *                      f_c uniform in +-20 kHz per stream,
 *                      I = clip(round(127.5 + 100 cos(phi) + N(0, sigma=2))), Q likewise with sin
 *   mode 1 "random"  : uniform random bytes (worst case for conditioning: |y| can be arbitrarily small)
 *   mode 2 "const"   : all bytes 128 (zero-signal edge: x = +0.5 everywhere)
 *   mode 3 "counter" : bytes n mod 256 — what the firmware really receives, because it leaves the RTL2832 in
 *                      test mode (usbh_rtlsdr.c:901, 662-664)
 * PRNG: xorshift64* seeded with 0x9E3779B97F4A7C15 ^ (stream_id + 1) * 0xD1B54A32D192ED03.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static inline uint64_t xs64(uint64_t* s) {
  uint64_t x = *s;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  *s = x;
  return x * 0x2545F4914F6CDD1DULL;
}
static inline double u01(uint64_t* s) { return ((double)(xs64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

static inline uint8_t clip8(double v) {
  double r = nearbyint(v);
  if (r < 0.0) r = 0.0;
  if (r > 255.0) r = 255.0;
  return (uint8_t)r;
}

/* Fill `out` with n_samples IQ pairs (2*n_samples bytes) of stream `stream_id`, starting at sample offset 0. */
void siggen_fill(uint8_t* out, size_t n_samples, uint64_t stream_id, int mode, double fs) {
  uint64_t st = 0x9E3779B97F4A7C15ULL ^ ((stream_id + 1) * 0xD1B54A32D192ED03ULL);
  if (!st) st = 1;
  for (int i = 0; i < 8; ++i) xs64(&st);
  if (mode == 1) {
    size_t nb = 2 * n_samples, i = 0;
    for (; i + 8 <= nb; i += 8) { uint64_t r = xs64(&st); for (int b = 0; b < 8; ++b) out[i + b] = (uint8_t)(r >> (8 * b)); }
    if (i < nb) { uint64_t r = xs64(&st); for (int b = 0; i < nb; ++i, ++b) out[i] = (uint8_t)(r >> (8 * b)); }
    return;
  }
  if (mode == 2) { for (size_t i = 0; i < 2 * n_samples; ++i) out[i] = 128; return; }
  if (mode == 3) { for (size_t i = 0; i < 2 * n_samples; ++i) out[i] = (uint8_t)(i & 0xFF); return; }
  const double two_pi = 6.283185307179586476925286766559;
  const double fc = (u01(&st) * 2.0 - 1.0) * 20000.0;
  double phi = u01(&st) * two_pi;
  const double w1 = two_pi * 1000.0 / fs, w2 = two_pi * 3100.0 / fs, w3 = two_pi * 7300.0 / fs;
  for (size_t n = 0; n < n_samples; ++n) {
    const double a = 0.5 * sin(w1 * (double)n) + 0.3 * sin(w2 * (double)n) + 0.2 * sin(w3 * (double)n);
    phi += two_pi * (fc + 75000.0 * a) / fs;
    if (phi > two_pi) phi -= two_pi;
    if (phi < 0.0) phi += two_pi;
    /* Box-Muller: one pair of N(0,1) per IQ sample */
    const double r = sqrt(-2.0 * log(u01(&st))), th = two_pi * u01(&st);
    out[2 * n]     = clip8(127.5 + 100.0 * cos(phi) + 2.0 * r * cos(th));
    out[2 * n + 1] = clip8(127.5 + 100.0 * sin(phi) + 2.0 * r * sin(th));
  }
}

/* n_streams streams back to back with byte stride `stride`; ids first_id .. first_id+n_streams-1 */
void siggen_fill_many(uint8_t* out, size_t stride, size_t n_samples, uint64_t first_id, uint32_t n_streams, int mode, double fs) {
  for (uint32_t s = 0; s < n_streams; ++s) siggen_fill(out + (size_t)s * stride, n_samples, first_id + s, mode, fs);
}

/* Hamming-windowed sinc low-pass, unity DC gain: the build's own generator for the 32/64/128-tap sets. */
void siggen_lowpass(float* taps, uint32_t n, double cutoff_over_fs) {
  const double pi = 3.14159265358979323846;
  double sum = 0.0, tmp[1024];
  if (n > 1024) n = 1024;
  for (uint32_t i = 0; i < n; ++i) {
    const double t = (double)i - 0.5 * (double)(n - 1);
    const double s = (t == 0.0) ? 2.0 * cutoff_over_fs : sin(2.0 * pi * cutoff_over_fs * t) / (pi * t);
    const double w = (n > 1) ? 0.54 - 0.46 * cos(2.0 * pi * (double)i / (double)(n - 1)) : 1.0;
    tmp[i] = s * w; sum += tmp[i];
  }
  for (uint32_t i = 0; i < n; ++i) taps[i] = (float)(tmp[i] / sum);
}
