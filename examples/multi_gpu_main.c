/*
 * multi_gpu_main.c — the multi-GPU host in plain C (north-star: "host code stays in C", "RCCL over xGMI only for the fan-out /
 * fan-in of stream batches").  ONE process, N devices: one sdrfm_t per device (sdrfm_config.device), streams sharded
 * in contiguous blocks (sdrfm_shard_range), no collective on the data path.  The reference's host is a single C superloop too
 * (src/main.c:40-81); its hand-off point is RTLSDR_XFER_COMPLETE (Class/RTLSDR/Src/usbh_rtlsdr.c:1094-1097).
 *
 *   C1 fan-out : the root device holds the whole batch [n_streams][nbytes]; every other device receives its block with grouped
 *                ncclSend (root) / ncclRecv (peer) — or hipMemcpyPeerAsync with --peer-copy;
 *   hot path   : sdrfm_process_batch(SDRFM_F_DEVICE_PTRS) on every device's own stream, all devices concurrently;
 *   C2 fan-in  : the audio blocks come back to the root the same way.
 *
 *   multi_gpu_main <iq.u8> <h.f32> <g.f32> <n_streams> <nbytes_per_stream> <n_gpus (0 = all)> <audio_out.f32> [--peer-copy] [--reps K] [--overlap]
 *
 * iq.u8 holds n_streams rows of nbytes_per_stream bytes.  audio_out: n_streams rows of n_audio floats.  Prints one JSON line.
 * With --reps K the timed loop repeats fan-out -> hot path -> fan-in K times on the same batch (end-to-end rate, SURVEY 7-5 b).
 * With --overlap every device keeps TWO input and TWO audio buffers and makes its calls with SDRFM_F_OVERLAP (include/sdrfm.h): repetition
 * k's block arrives in buffer k & 1 while repetition k - 1 is still being demodulated, the calls of consecutive repetitions run concurrently
 * on the device, and the fan-in of repetition k - 1 goes out behind sdrfm_flush_previous while call k runs; sdrfm_flush before the last one.
 */
#define _POSIX_C_SOURCE 200112L   /* clock_gettime, setenv */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "sdrfm.h"

#define HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLC(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); return 1; } } while (0)
#define SDRC(call) do { int st_ = (call); if (st_ != SDRFM_OK) { fprintf(stderr, "%s: %s\n", #call, sdrfm_strerror(st_)); return 1; } } while (0)
#define MAX_GPUS 16

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(sz > 0 ? (size_t)sz : 1);
  if (fread(p, 1, (size_t)sz, f) != (size_t)sz) { perror("fread"); exit(2); }
  fclose(f);
  *n = (size_t)sz;
  return p;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s iq.u8 h.f32 g.f32 n_streams nbytes_per_stream n_gpus audio_out.f32 [--peer-copy] [--reps K] [--overlap]\n", argv[0]); return 2; }
  size_t niq, nh, ng;
  unsigned char* iq = (unsigned char*)slurp(argv[1], &niq);
  float* h = (float*)slurp(argv[2], &nh);
  float* g = (float*)slurp(argv[3], &ng);
  const uint32_t n_streams = (uint32_t)atoi(argv[4]), nbytes = (uint32_t)atoi(argv[5]);
  int world = atoi(argv[6]), peer_copy = 0, reps = 1, overlap = 0;
  for (int i = 8; i < argc; ++i) {
    if (!strcmp(argv[i], "--peer-copy")) peer_copy = 1;
    else if (!strcmp(argv[i], "--overlap")) overlap = 1;
    else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
  }
  if ((size_t)n_streams * nbytes != niq || (nbytes & 1u) || !n_streams) { fprintf(stderr, "iq.u8 must hold n_streams x nbytes_per_stream bytes (even)\n"); return 2; }
  /* RCCL's streams take hardware queues out of the runtime's pool of 4; the library's overlapped calls need two of their own beside the caller's stream
     (INTEGRATION.md): ask for 8 before the runtime starts, unless the environment already says otherwise */
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  int ndev = 0;
  HIPC(hipGetDeviceCount(&ndev));
  if (world <= 0 || world > ndev) world = ndev;
  if (world > MAX_GPUS) world = MAX_GPUS;
  if (world < 1) { fprintf(stderr, "no HIP device\n"); return 1; }

  /* ---- one handle, one stream, one shard per device ----------------------------------------------------------------------- */
  sdrfm_t* dm[MAX_GPUS];
  hipStream_t st[MAX_GPUS];
  unsigned char* d_iq[MAX_GPUS];
  float* d_audio[MAX_GPUS];
  unsigned char* d_iq2[MAX_GPUS];       /* --overlap: the second input / audio buffer of every device */
  float* d_audio2[MAX_GPUS];
  uint32_t first[MAX_GPUS], count[MAX_GPUS], n_audio = 0;
  ncclComm_t comm[MAX_GPUS];
  int devs[MAX_GPUS];
  for (int d = 0; d < world; ++d) devs[d] = d;
  if (!peer_copy) NCCLC(ncclCommInitAll(comm, world, devs));
  sdrfm_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.fir_taps = (uint32_t)(nh / sizeof(float)); cfg.fir_decim = 10; cfg.fir_coeffs = h;
  cfg.audio_taps = (uint32_t)(ng / sizeof(float)); cfg.audio_decim = 5; cfg.audio_coeffs = g;
  cfg.max_bytes_per_call = nbytes;
  unsigned char* d_iq_root = NULL;       /* the whole batch on the root device */
  float* d_audio_root = NULL;
  size_t audio_stride = 0;
  for (int d = 0; d < world; ++d) {
    SDRC(sdrfm_shard_range(n_streams, (uint32_t)world, (uint32_t)d, &first[d], &count[d]));
    HIPC(hipSetDevice(d));
    HIPC(hipStreamCreateWithFlags(&st[d], hipStreamNonBlocking));
    dm[d] = NULL; d_iq[d] = NULL; d_audio[d] = NULL; d_iq2[d] = NULL; d_audio2[d] = NULL;
    if (!count[d]) continue;
    cfg.n_streams = count[d]; cfg.device = d;
    SDRC(sdrfm_create(&cfg, &dm[d]));
    SDRC(sdrfm_set_stream(dm[d], (void*)st[d]));
    SDRC(sdrfm_audio_count(dm[d], nbytes, &n_audio));
    audio_stride = ((size_t)n_audio + 63) & ~(size_t)63;
    if (d == 0) {
      HIPC(hipMalloc((void**)&d_iq_root, (size_t)n_streams * nbytes));
      HIPC(hipMalloc((void**)&d_audio_root, (size_t)n_streams * audio_stride * sizeof(float)));
      HIPC(hipMemcpy(d_iq_root, iq, (size_t)n_streams * nbytes, hipMemcpyHostToDevice));
      d_iq[0] = d_iq_root + (size_t)first[0] * nbytes;           /* the root's own block is used in place */
      d_audio[0] = d_audio_root + (size_t)first[0] * audio_stride;
    } else {
      HIPC(hipMalloc((void**)&d_iq[d], (size_t)count[d] * nbytes));
      HIPC(hipMalloc((void**)&d_audio[d], (size_t)count[d] * audio_stride * sizeof(float)));
    }
    if (overlap) {
      HIPC(hipMalloc((void**)&d_iq2[d], (size_t)count[d] * nbytes));
      HIPC(hipMalloc((void**)&d_audio2[d], (size_t)count[d] * audio_stride * sizeof(float)));
      if (d == 0) HIPC(hipMemcpy(d_iq2[0], d_iq[0], (size_t)count[0] * nbytes, hipMemcpyDeviceToDevice));   /* the root's own block, second copy */
    }
  }
  for (int d = 0; d < world; ++d) { HIPC(hipSetDevice(d)); HIPC(hipDeviceSynchronize()); }

  const double t0 = now_s();
  for (int rep = 0; rep < reps + (overlap ? 1 : 0); ++rep) {
    const int cur = overlap ? (rep & 1) : 0, prv = cur ^ 1;
    const int call = rep < reps, fan_in_rep = overlap ? rep - 1 : rep;          /* --overlap: the fan-in runs one repetition behind the calls */
    unsigned char* const* iq_of = (overlap && cur) ? d_iq2 : d_iq;
    float* const* au_of = (overlap && cur) ? d_audio2 : d_audio;
    float* const* au_in = overlap ? (prv ? d_audio2 : d_audio) : d_audio;
    /* C1: fan-out of the IQ blocks from the root */
    if (world > 1 && call) {
      if (peer_copy) {
        for (int d = 1; d < world; ++d)
          if (count[d]) HIPC(hipMemcpyPeerAsync(iq_of[d], d, d_iq_root + (size_t)first[d] * nbytes, 0, (size_t)count[d] * nbytes, st[d]));
      } else {
        /* the root's sends run on its stream after whatever produced the batch; a peer's receive on the peer's stream */
        NCCLC(ncclGroupStart());
        for (int d = 1; d < world; ++d) {
          if (!count[d]) continue;
          NCCLC(ncclSend(d_iq_root + (size_t)first[d] * nbytes, (size_t)count[d] * nbytes, ncclUint8, d, comm[0], st[0]));
          NCCLC(ncclRecv(iq_of[d], (size_t)count[d] * nbytes, ncclUint8, 0, comm[d], st[d]));
        }
        NCCLC(ncclGroupEnd());
      }
    }
    /* the hot path on every device, enqueued without waiting: the devices run concurrently */
    for (int d = 0; d < world && call; ++d) {
      if (!count[d]) continue;
      uint32_t na = 0;
      SDRC(sdrfm_process_batch(dm[d], iq_of[d], nbytes, nbytes, au_of[d], audio_stride, &na, SDRFM_F_DEVICE_PTRS | (overlap ? SDRFM_F_OVERLAP : 0u)));
      if (na != n_audio) { fprintf(stderr, "device %d: %u audio samples, expected %u\n", d, na, n_audio); return 1; }
    }
    if (fan_in_rep < 0) continue;
    /* --overlap: the device's stream is put behind the call whose audio goes out now (all but the most recent one; all of them at the end) */
    for (int d = 0; d < world && overlap; ++d)
      if (count[d]) SDRC(call ? sdrfm_flush_previous(dm[d]) : sdrfm_flush(dm[d]));
    /* C2: fan-in of the audio blocks */
    if (world > 1) {
      if (peer_copy) {
        for (int d = 1; d < world; ++d)
          if (count[d]) HIPC(hipMemcpyPeerAsync(d_audio_root + (size_t)first[d] * audio_stride, 0, au_in[d], d, (size_t)count[d] * audio_stride * sizeof(float), st[d]));
      } else {
        NCCLC(ncclGroupStart());
        for (int d = 1; d < world; ++d) {
          if (!count[d]) continue;
          NCCLC(ncclSend(au_in[d], (size_t)count[d] * audio_stride, ncclFloat32, 0, comm[d], st[d]));
          NCCLC(ncclRecv(d_audio_root + (size_t)first[d] * audio_stride, (size_t)count[d] * audio_stride, ncclFloat32, d, comm[0], st[0]));
        }
        NCCLC(ncclGroupEnd());
      }
    }
    if (overlap && count[0] && au_in[0] != d_audio_root + (size_t)first[0] * audio_stride)    /* the root's own block sits in its second buffer */
      HIPC(hipMemcpyAsync(d_audio_root + (size_t)first[0] * audio_stride, au_in[0], (size_t)count[0] * audio_stride * sizeof(float), hipMemcpyDeviceToDevice, st[0]));
    if (!overlap || !call)
      for (int d = 0; d < world; ++d) { HIPC(hipSetDevice(d)); HIPC(hipStreamSynchronize(st[d])); }
  }
  const double dt = now_s() - t0;

  /* ---- results --------------------------------------------------------------------------------------------------------------- */
  float* audio = (float*)malloc((size_t)n_streams * n_audio * sizeof(float) + 4);
  HIPC(hipSetDevice(0));
  HIPC(hipMemcpy2D(audio, (size_t)n_audio * sizeof(float), d_audio_root, audio_stride * sizeof(float), (size_t)n_audio * sizeof(float), n_streams, hipMemcpyDeviceToHost));
  FILE* fo = fopen(argv[7], "wb");
  if (!fo) { perror(argv[7]); return 2; }
  fwrite(audio, sizeof(float), (size_t)n_streams * n_audio, fo);
  fclose(fo);
  printf("{\"n_gpus\":%d,\"n_streams\":%u,\"bytes_per_stream\":%u,\"n_audio\":%u,\"transport\":\"%s\",\"reps\":%d,\"seconds\":%.6f,"
         "\"end_to_end_MSamples_per_s\":%.1f,\"calls\":\"%s\",\"kernel\":\"%s\",\"shards\":[", world, n_streams, nbytes, n_audio,
         peer_copy ? "hipMemcpyPeerAsync" : "RCCL ncclSend/ncclRecv", reps, dt, (double)reps * n_streams * (nbytes / 2) / dt / 1e6,
         overlap ? "SDRFM_F_OVERLAP, two buffers per device, fan-in one call behind" : "one after the other", dm[0] ? sdrfm_kernel_name(dm[0]) : "");
  for (int d = 0; d < world; ++d) printf("%s[%u,%u]", d ? "," : "", first[d], count[d]);
  printf("]}\n");
  for (int d = 0; d < world; ++d) {
    HIPC(hipSetDevice(d));
    if (dm[d]) sdrfm_destroy(dm[d]);
    if (d > 0 && d_iq[d]) HIPC(hipFree(d_iq[d]));
    if (d > 0 && d_audio[d]) HIPC(hipFree(d_audio[d]));
    if (d_iq2[d]) HIPC(hipFree(d_iq2[d]));
    if (d_audio2[d]) HIPC(hipFree(d_audio2[d]));
    if (!peer_copy) ncclCommDestroy(comm[d]);
    HIPC(hipStreamDestroy(st[d]));
  }
  HIPC(hipSetDevice(0));
  HIPC(hipFree(d_iq_root)); HIPC(hipFree(d_audio_root));
  free(audio); free(iq); free(h); free(g);
  return 0;
}
