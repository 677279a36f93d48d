/*
 * sdrfm_q_host.h — design Q, host-side constants and the operand-table builder (plain C; see qtaps.c).
 * Internal to the library and its CPU tests (tests/test_q_tables.py calls sdrfm_q_build through libsdrfm.so); not part of
 * the drop-in boundary in include/sdrfm.h.
 */
#ifndef SDRFM_Q_HOST_H
#define SDRFM_Q_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRFM_Q_DIGITS 3                    /* balanced base-256 digits per tap (gate (a): tools/q_emulate.py) */
#define SDRFM_Q_HMAX (127 * 65793)          /* largest |H| three digits in [-128, 127] can hold: 127 (65536 + 256 + 1) */
#define SDRFM_Q_MAX_D 16                    /* FIR decimation: even, <= 16 (D / 2 K-chunks of 64 bytes per window) */

#define SDRFM_Q_SPARSE_CHUNKS(D) (((D) + 3u) / 4u)   /* K-chunks of 128 window bytes (v_smfmac_i32_16x16x128_i8) per digit */

/* h[0..T) -> A[ceil(D/4)][3][64][16] (sparse chunk, digit, lane, kept slot), the fp32 scale q (y = q * (S0 + 256 S1 + 65536 S2) + cst),
 * cst = 0.5 * sum(h), and the first chunk that holds a non-zero tap (chunks before it need no MFMA).
 * Returns 0, or -1 when the geometry / taps cannot be served (T > 9 D, D odd, non-finite or all-zero taps). */
int sdrfm_q_build(const float* h, uint32_t T, uint32_t D, int8_t* A, float* q, float* cst, uint32_t* first_chunk);

/* The conditioning guard's thresholds for channel taps h[0..T) and audio taps g[0..Ta) (see qtaps.c): a lane is repaired when one of its y's
 * has max(|re|, |im|) < *guard_r or one of its |d|'s exceeds *guard_a.  Returns 0 or -1. */
int sdrfm_q_guard(const float* h, uint32_t T, const float* g, uint32_t Ta, float* guard_r, float* guard_a);
/* ... with the bound on |y_fast-q - y_definition| the radius rests on chosen: worst_case = 0 the statistical one (sdrfm_q_guard), 1 the PROVEN
 * worst case of the chain's and the recombination's roundings and the taps' quantisation (SDRFM_CFG_GUARD_WORST_CASE; qtaps.c has the terms). */
int sdrfm_q_guard2(const float* h, uint32_t T, const float* g, uint32_t Ta, int worst_case, float* guard_r, float* guard_a);

#ifdef __cplusplus
}
#endif
#endif
