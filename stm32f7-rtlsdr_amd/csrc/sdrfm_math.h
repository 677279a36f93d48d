/*
 * sdrfm_math.h — the scalar arithmetic of stages K1/K3 written so that host and device evaluate it identically
 * (only fmaf, +, -, *, /, comparisons; the TU is compiled with -ffp-contract=off).
 *
 * K2/K4 (the FIR chains) are plain fmaf chains, "oldest sample first" — see DESIGN.md "Frozen spec".
 *
 * sdrfm_atan2f is this build's own atan2f (the oracle uses libm's): a degree-7 minimax polynomial in s = v^2 for
 * atan(v) = v + v*s*Q(s) on v = min/max in [0,1] (coefficients fitted by tools/fit_atan.py, core error 1.1 ulp), then
 * octant fix-up.  Total error <= ~3 ulp of the result, i.e. < 8e-7 rad — inside the 1e-5 parity tolerance.  On the device
 * the quotient is min * v_rcp_f32(max) (one more ulp); the host build keeps the IEEE divide.
 */
#ifndef SDRFM_MATH_H
#define SDRFM_MATH_H

#if defined(__HIPCC__)
#define SDRFM_HD __host__ __device__ __forceinline__
#else
#define SDRFM_HD static inline
#endif

SDRFM_HD float sdrfm_atan_unit(float v) { /* v in [0,1] */
  const float s = v * v;
  float q = 0x1.57b128p-9f;
  q = __builtin_fmaf(q, s, -0x1.efda1p-7f);
  q = __builtin_fmaf(q, s, 0x1.50dd96p-5f);
  q = __builtin_fmaf(q, s, -0x1.2dbcfap-4f);
  q = __builtin_fmaf(q, s, 0x1.b11b74p-4f);
  q = __builtin_fmaf(q, s, -0x1.228754p-3f);
  q = __builtin_fmaf(q, s, 0x1.99673ep-3f);
  q = __builtin_fmaf(q, s, -0x1.55546cp-2f);
  return __builtin_fmaf(v, s * q, v);
}

/* atan2f(y, x) for finite inputs; caller handles x == 0 && y == 0. Honours the sign of a zero y like libm. */
SDRFM_HD float sdrfm_atan2f(float y, float x) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
#if defined(__HIP_DEVICE_COMPILE__)
  float a = sdrfm_atan_unit(mn * __builtin_amdgcn_rcpf(mx)); /* v_rcp_f32 (1 ulp) instead of the ~12-instruction IEEE divide */
#else
  float a = sdrfm_atan_unit(mn / mx);
#endif
  if (ay > ax) a = 0x1.921fb6p+0f - a;     /* pi/2 - a */
  if (x < 0.0f) a = 0x1.921fb6p+1f - a;    /* pi - a   */
  return __builtin_copysignf(a, y);
}

/* K3: FM discriminator of consecutive decimated samples y (re,im) and previous p (re,im). */
SDRFM_HD float sdrfm_discriminate(float yr, float yi, float pr, float pi) {
  const float re = __builtin_fmaf(yr, pr, yi * pi);
  const float im = yi * pr - yr * pi; /* two rounded products (TU is built with -ffp-contract=off) */
  const float a = sdrfm_atan2f(im, re);   /* NaN when both are 0 (0/0); selected away, no branch */
  return (re == 0.0f && im == 0.0f) ? 0.0f : a;
}

#endif
