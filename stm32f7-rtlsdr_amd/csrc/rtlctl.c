/*
 * rtlctl.c — host-side restatement of the two pure-integer/double computations the reference performs when it
 * configures the RTL2832 for the stream this library consumes.  Plain C, no GPU, no USB: a non-MCU
 * front end can use it to program a dongle consistently with the firmware.  Known-answer tests: tests/test_rtlctl.py
 * (values obtained from the reference's own code).
 *
 *   sdrfm_rtl_pack_fir      : RTLSDR_set_fir's RTLSDR_FIR_CALC state — 8 x int8 + 8 x int12 -> 20 register bytes for demod
 *                             page 1, registers 0x1c..0x2f (Class/RTLSDR/Src/usbh_rtlsdr.c:552-575)
 *   sdrfm_rtl_resampler     : RTLSDR_set_sample_rate state 0 — rsamp_ratio, the sign-extended "real" ratio and the exact
 *                             rate (usbh_rtlsdr.c:676-691).  Unlike the firmware, which only logs, invalid input is refused.
 *   sdrfm_e4k_pll_params    : E4K_compute_pll_params (Class/RTLSDR/Src/tuner_e4k.c:689-737) — the E4000 LO synthesiser word
 */
#include <stdint.h>

#include "../../include/sdrfm.h"

int sdrfm_rtl_pack_fir(const int* fir16, uint8_t* out20) {
  if (!fir16 || !out20) return SDRFM_EINVAL;
  for (int i = 0; i < 8; ++i) {                 /* outer taps: int8 */
    if (fir16[i] < -128 || fir16[i] > 127) return SDRFM_EINVAL;
    out20[i] = (uint8_t)fir16[i];
  }
  for (int i = 0; i < 8; i += 2) {              /* inner taps: two int12 in three bytes */
    const int a = fir16[8 + i], b = fir16[8 + i + 1];
    if (a < -2048 || a > 2047 || b < -2048 || b > 2047) return SDRFM_EINVAL;
    uint8_t* o = out20 + 8 + i * 3 / 2;
    o[0] = (uint8_t)(a >> 4);
    o[1] = (uint8_t)((a << 4) | ((b >> 8) & 0x0f));
    o[2] = (uint8_t)b;
  }
  return SDRFM_OK;
}

int sdrfm_rtl_resampler(uint32_t samp_rate, uint32_t xtal_hz, uint32_t* rsamp_ratio, uint32_t* real_rsamp_ratio, double* real_rate) {
  if (!rsamp_ratio || !real_rsamp_ratio || !real_rate || !xtal_hz) return SDRFM_EINVAL;
  /* the resampler's validity window (usbh_rtlsdr.c:677-678) */
  if (samp_rate <= 225000u || samp_rate > 3200000u || (samp_rate > 300000u && samp_rate <= 900000u)) return SDRFM_EINVAL;
  const double num = (double)xtal_hz * 4194304.0;          /* xtal * 2^22 */
  uint32_t ratio = (uint32_t)(num / (double)samp_rate);
  ratio &= 0x0ffffffcu;
  const uint32_t real = ratio | ((ratio & 0x08000000u) << 1);
  *rsamp_ratio = ratio;
  *real_rsamp_ratio = real;
  *real_rate = num / (double)real;
  return SDRFM_OK;
}

/* E4000 LO bands: below `below_hz` the VCO runs at R x LO; `synth7` is the register code of that divider (bit 3 = 3-phase
 * mixing).  Values are the chip's (tuner_e4k.c:301-312); above the last band the firmware keeps R = 2, code 0. */
static const struct { uint32_t below_hz; uint8_t synth7, r; } e4k_bands[] = {
  {72400000u, 0x0f, 48}, {81200000u, 0x0e, 40}, {108300000u, 0x0d, 32}, {162500000u, 0x0c, 24}, {216600000u, 0x0b, 16},
  {325000000u, 0x0a, 12}, {350000000u, 0x09, 8}, {432000000u, 0x03, 8}, {667000000u, 0x02, 6}, {1200000000u, 0x01, 4},
};

int sdrfm_e4k_pll_params(uint32_t fosc_hz, uint32_t intended_flo_hz, sdrfm_e4k_pll* out) {
  if (!out || fosc_hz < 16000000u || fosc_hz > 30000000u) return SDRFM_EINVAL;   /* is_fosc_valid, tuner_e4k.c:325-333 */
  uint8_t r = 2, code = 0;
  for (unsigned i = 0; i < sizeof(e4k_bands) / sizeof(e4k_bands[0]); ++i)
    if (intended_flo_hz < e4k_bands[i].below_hz) { r = e4k_bands[i].r; code = e4k_bands[i].synth7; break; }
  const uint64_t fvco = (uint64_t)intended_flo_hz * r;        /* wanted VCO frequency */
  const uint64_t z = fvco / fosc_hz;                           /* integer multiplier */
  const uint64_t x = ((fvco - z * fosc_hz) << 16) / fosc_hz;   /* fraction in 1/65536 */
  /* what the chip then produces: fosc*Z + floor(fosc*X / 65536), divided by R (compute_fvco / compute_flo, :340-360) —
   * Z and X enter truncated to their register widths, as in the firmware's call */
  const uint64_t got = (uint64_t)fosc_hz * (uint8_t)z + (((uint64_t)fosc_hz * (uint16_t)x) >> 16);
  out->fosc = fosc_hz;
  out->intended_flo = intended_flo_hz;
  out->flo = (uint32_t)(got / r);
  out->x = (uint16_t)x;
  out->z = (uint8_t)z;
  out->r = r;
  out->r_idx = code;
  out->threephase = (code & 0x08) ? 1 : 0;
  return SDRFM_OK;
}

/* Multi-GPU fan-out: the contiguous block of streams rank `rank` of `world` owns — the first n_streams % world
 * ranks get one stream more.  The one definition both hosts use: examples/multi_gpu_main.c directly, the Python fan-out
 * (stm32f7-rtlsdr_amd/fanout.py: shard_range) through ctypes. */
int sdrfm_shard_range(uint32_t n_streams, uint32_t world, uint32_t rank, uint32_t* first, uint32_t* count) {
  if (!first || !count || world == 0 || rank >= world) return SDRFM_EINVAL;
  const uint32_t q = n_streams / world, r = n_streams % world;
  *first = rank * q + (rank < r ? rank : r);
  *count = q + (rank < r ? 1u : 0u);
  return SDRFM_OK;
}
