/*
 * qtaps.c — design Q, host side: the channel FIR's taps as i8 matrix-pipe operands (plain C, no GPU).
 *
 * Design Q evaluates stage K2 (DESIGN.md "Frozen spec"; FIR convention: CMSIS/core/arm_math.h:3291-3331 of the reference, byte
 * format: Class/RTLSDR/Inc/usbh_rtlsdr.h:165-173) on the i8 matrix pipe of gfx950 (v_mfma_i32_16x16x64_i8):
 *
 *   y[m] = q * (S0 + 2^8 S1 + 2^16 S2) + 0.5 * sum(h),     S_t = sum_k digit_t(H[k]) * (byte - 128),   H[k] = round(h[k] / q)
 *
 * The input bytes are the B operand as they are (byte XOR 0x80 = byte - 128 as i8; the spec's x = byte - 127.5 is that plus 0.5,
 * which the constant restores); interleaved I/Q stays the K dimension.  The A operand is a Toeplitz slice of the taps: one MFMA
 * tile computes a BLOCK of 8 consecutive outputs (16 rows: row 2 o + comp, comp 0 = I, 1 = Q) for 16 blocks (columns) at once,
 * from each block's WINDOW of 32 D bytes = the block's own 16 D bytes and the 16 D bytes before them, K-chunked 64 bytes at a
 * time.  Window sample w (0 .. 16 D - 1, byte pair 2 w, 2 w + 1) meets output o through tap k = 8 D + D o + D - 1 - w.
 *
 * The A operand is exactly 2:4 sparse — an I row holds taps at even K positions only, a Q row at odd ones — so the kernel issues the
 * SPARSE instruction v_smfmac_i32_16x16x128_i8: 128 window bytes per issue at the cost of a dense 64 (tools/ubench/ubench13.hip),
 * ceil(D / 4) issues per digit instead of D / 2.  Operand layout (found by one-hot probing on the device, tools/ubench/ubench14.hip):
 *   A (sparse): lane l = row (l & 15), K group gA = l >> 4; its 16 bytes are slots s = 8 h + 2 q + j (h = 0, 1; q = 0..3; j = 0, 1) =
 *               the j-th kept value of the group of four dense positions K = 32 gA + 16 h + 4 q .. + 3;
 *   idx:        one VGPR; bits [4 (4 h + q) + 1 : 4 (4 h + q)] = position of the first kept value in its group, the next two bits the
 *               second: 0x88888888 for an I row (positions 0, 2), 0xDDDDDDDD for a Q row (1, 3);
 *   B (dense):  lane l = column (l & 15), gB = l >> 4; bytes 0..15 are K = 16 gB .. + 15, bytes 16..31 are K = 64 + 16 gB .. + 15 — the
 *               same two 16-byte pieces a lane holds for two consecutive dense v_mfma_i32_16x16x64_i8 issues.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "sdrfm_q_host.h"

int sdrfm_q_build(const float* h, uint32_t T, uint32_t D, int8_t* A, float* q_out, float* cst_out, uint32_t* first_chunk) {
  if (!h || !A || !q_out || !cst_out || !first_chunk) return -1;
  if (D < 2 || (D & 1u) || D > SDRFM_Q_MAX_D || T < 1 || T > 9 * D) return -1;
  double hmax = 0.0, hsum = 0.0;
  for (uint32_t k = 0; k < T; ++k) {
    if (!isfinite(h[k])) return -1;
    if (fabs((double)h[k]) > hmax) hmax = fabs((double)h[k]);
    hsum += (double)h[k];
  }
  if (hmax == 0.0) return -1;
  const double q = hmax / (double)SDRFM_Q_HMAX;
  const float qf = (float)q;
  if (!(qf > 0.0f) || !isfinite(65536.0f * qf)) return -1;        /* taps so small / large that the scale leaves fp32 */
  int8_t dig[SDRFM_Q_DIGITS][9 * SDRFM_Q_MAX_D];
  for (uint32_t k = 0; k < T; ++k) {
    /* quantise against the fp32 scale the device multiplies with, so that q_f32 * H is the best fixed-point image of h[k] */
    long long H = llround((double)h[k] / (double)qf);
    if (H > SDRFM_Q_HMAX) H = SDRFM_Q_HMAX;
    if (H < -SDRFM_Q_HMAX) H = -SDRFM_Q_HMAX;
    for (int t = 0; t < SDRFM_Q_DIGITS; ++t) {                     /* balanced base-256 digits, each in [-128, 127] */
      long long d = ((H + 128) % 256 + 256) % 256 - 128;
      dig[t][k] = (int8_t)d;
      H = (H - d) / 256;
    }
    if (H != 0) return -1;
  }
  const uint32_t nsc = (D + 3) / 4;                                /* sparse K-chunks of 128 window bytes */
  memset(A, 0, (size_t)nsc * SDRFM_Q_DIGITS * 64 * 16);
  uint32_t c0 = nsc;
  for (uint32_t c = 0; c < nsc; ++c)
    for (uint32_t l = 0; l < 64; ++l)
      for (uint32_t sl = 0; sl < 16; ++sl) {
        const uint32_t row = l & 15u, o = row >> 1, comp = row & 1u;
        const uint32_t h2 = sl >> 3, q4 = (sl >> 1) & 3u, j = sl & 1u;
        const uint32_t kb = 128 * c + 32 * (l >> 4) + 16 * h2 + 4 * q4 + comp + 2 * j;   /* window byte of this kept slot */
        const int k = (int)(8 * D + D * o + D - 1) - (int)(kb >> 1);
        if (kb >= 32 * D || k < 0 || k >= (int)T) continue;
        for (int t = 0; t < SDRFM_Q_DIGITS; ++t) {
          A[(((size_t)c * SDRFM_Q_DIGITS + t) * 64 + l) * 16 + sl] = dig[t][k];
          if (dig[t][k] != 0 && c < c0) c0 = c;
        }
      }
  *q_out = qf;
  *cst_out = (float)(0.5 * hsum);
  *first_chunk = c0;
  return 0;
}

/* The conditioning guard's thresholds (kernel: sdrfm_q.hip; derivation: DESIGN.md 4.Q "guard").
 * Design Q's y and the definition's fmaf chain differ by the chain's own rounding — T roundings at partial sums of up to P = 127.5 sum|h| —
 * plus the 24-bit taps and the three roundings of the recombination.  Two bounds E on |dy|:
 *   statistical (the default):  E = 1.25 sqrt(T) P 2^-24 (9.5e-5 for the BASELINE 64 taps: 28 standard deviations of what uniform random bytes
 *                               produce, 9 of a full-scale carrier's);
 *   worst case (SDRFM_CFG_GUARD_WORST_CASE, round 6):  E = (T + 4) P 2^-24 + 64 T q — every one of the chain's T roundings at the largest partial
 *                               sum and falling the same way ((T) P 2^-24 (1 + T 2^-24)), the recombination's conversion and two fused
 *                               multiply-adds (3 x 1.01 P 2^-24), every tap's quantisation error q / 2 against a byte of magnitude 128, the rounding
 *                               of the constant 0.5 sum h (< P 2^-24).  A PROVEN bound on |y_fast-q - y_definition| for any bytes; 6.9 times the
 *                               statistical one at T = 64.
 * A discriminator output then moves by at most asin(E / |y|) + asin(E / |p|) and the audio by max|g| times that per tap.  Outputs with a y below
 * R = 2 max|g| E / 5e-6 are recomputed by the definition's own chain (ONE ill-conditioned pair may use half of the 1e-5 tolerance; everything
 * else together is measured below 1e-6), and so are d's within 2 * 5e-6 / max|g| of +-pi, where an unrepaired pair could still land on the
 * other side of the branch cut.  What the worst-case radius PROVES: every unrepaired d is within 5e-6 / max|g| + 1.1e-6 of the definition's
 * (1.1e-6: two arctangents, the wrap), hence every audio sample within sum|g| (5e-6 / max|g| + 1.1e-6) of the oracle's — 4.6e-5 for the
 * BASELINE audio taps, reached only if all 32 d's of a window sat at the radius with aligned errors; within 1e-5 whenever at most one
 * pair of a window is that close to the radius.  The fully proven 1e-5 is SDRFM_CFG_BIT_EXACT (include/sdrfm.h).
 * Returns 0, or -1 for taps the guard cannot serve (all-zero audio taps are served: nothing to guard). */
int sdrfm_q_guard2(const float* h, uint32_t T, const float* g, uint32_t Ta, int worst_case, float* guard_r, float* guard_a) {
  if (!h || !g || !guard_r || !guard_a || T < 1 || Ta < 1) return -1;
  double habs = 0.0, hmax = 0.0, gmax = 0.0;
  for (uint32_t k = 0; k < T; ++k) { habs += fabs((double)h[k]); if (fabs((double)h[k]) > hmax) hmax = fabs((double)h[k]); }
  for (uint32_t k = 0; k < Ta; ++k) if (fabs((double)g[k]) > gmax) gmax = fabs((double)g[k]);
  if (!isfinite(habs) || !isfinite(gmax)) return -1;
  if (gmax == 0.0) { *guard_r = 0.0f; *guard_a = 4.0f; return 0; }
  const double P = 127.5 * habs, u = ldexp(1.0, -24);
  const double E = worst_case ? ((double)T + 4.0) * P * u + 64.0 * (double)T * (hmax / (double)SDRFM_Q_HMAX)
                              : 1.25 * sqrt((double)T) * P * u;
  *guard_r = (float)(2.0 * gmax * E / 5e-6);
  *guard_a = (float)(3.14159265358979323846 - 2.0 * 5e-6 / gmax);
  return 0;
}

int sdrfm_q_guard(const float* h, uint32_t T, const float* g, uint32_t Ta, float* guard_r, float* guard_a) {
  return sdrfm_q_guard2(h, T, g, Ta, 0, guard_r, guard_a);
}
