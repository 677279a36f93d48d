/*
 * pcm_sink.c — the step right after the path on the reference board: demodulated audio (radians per
 * 240 kS/s sample, at 48 kHz) -> FM de-emphasis -> int16 stereo-interleaved PCM in the layout BSP_AUDIO_OUT_Play consumes
 * (uint16_t* pBuffer, Size in bytes; Utilities/STM32746G-Discovery/stm32746g_discovery_audio.c:224; L and R carry the same
 * mono programme).  Host-side plain C on purpose: the de-emphasis is a one-pole recursion (serial per stream) over
 * 4.9 MB per batch, i.e. microseconds of CPU time next to the D2H copy; it is not part of the GPU hot path.
 *
 *   y[n]   = y[n-1] + alpha * (x[n] - y[n-1])          alpha = 1 - exp(-1 / (fs * tau)),  tau = 75e-6 (US) or 50e-6 (EU)
 *   pcm[n] = sat16(lrintf(y[n] * gain))                 gain: radians -> full scale, e.g. 32767 / (2*pi*75e3/240e3)
 */
#include <math.h>
#include <stdint.h>

#include "../../include/sdrfm.h"

int sdrfm_pcm_deemph_s16(const float* audio, uint32_t n, float alpha, float gain, float* state, int16_t* pcm_stereo) {
  if ((n && (!audio || !pcm_stereo)) || !state || !(alpha > 0.0f) || alpha > 1.0f) return SDRFM_EINVAL;
  float y = *state;
  for (uint32_t i = 0; i < n; ++i) {
    y = fmaf(alpha, audio[i] - y, y);
    float v = y * gain;
    if (v > 32767.0f) v = 32767.0f;
    if (v < -32768.0f) v = -32768.0f;
    const int16_t s = (int16_t)lrintf(v);
    pcm_stereo[2 * i] = s;
    pcm_stereo[2 * i + 1] = s;
  }
  *state = y;
  return SDRFM_OK;
}

float sdrfm_pcm_alpha(float fs_hz, float tau_s) { return 1.0f - expf(-1.0f / (fs_hz * tau_s)); }
