/* A client of include/ksa.h written in plain C99: no Python, no torch, no HIP headers -- what a non-Python host of the
 * reference's hot path would look like.  Built and run by tests/test_gpu_round3.py::test_plain_c_client on the GPU box:
 *   gcc -std=c99 -O2 -I include tests/c_client/ksa_client.c -L prgs-sdr-kspecanal_amd -lksa -lm ...
 * It mirrors python/kspecanal.py's zeroSpan frame loop (K:460-484) and one scan pass (K:621-668) through the host-pointer
 * entry points, on a synthetic on-bin tone whose answers are known in closed form:
 * a tone A*exp(j*2*pi*k*n/N) under a rectangular window reads 2A at bin k (K:391), i.e. 10*log10(2A) - gain dB. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ksa.h"

#define CHECK(call)                                                             \
  do {                                                                          \
    if ((call) != 0) {                                                          \
      fprintf(stderr, "%s failed: %s\n", #call, ksa_last_error());              \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

int main(void) {
  enum { N = 1024, FULL = 8192, XRES = 64 };
  const double q = 0.5, amp = 0.25, gain = 19.1;
  const int kbin = 100;
  const double pi = 3.14159265358979323846;
  static float iq[2 * FULL], win[N], mag[N], cur[N], mx[N], mn[N], av[N], hm[KSA_HM_ROWS * XRES];
  static int32_t starts[64];
  int nwin = 0, i;
  /* window starts exactly as K:368 / K:386-390 */
  for (i = 0; i < (int)(FULL / (N * q)); ++i) {
    const int s = (int)(i * N * q);
    if (s + N > FULL) break;
    starts[nwin++] = s;
  }
  for (i = 0; i < N; ++i) win[i] = 1.0f;
  for (i = 0; i < FULL; ++i) {
    iq[2 * i] = (float)(amp * cos(2 * pi * kbin * (double)i / N));
    iq[2 * i + 1] = (float)(amp * sin(2 * pi * kbin * (double)i / N));
  }
  ksa_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = KSA_ABI_VERSION;
  cfg.device = 0;
  cfg.fft_size = N;
  cfg.full_size = FULL;
  cfg.num_windows = nwin;
  cfg.window_starts = starts;
  cfg.window = win;
  cfg.mag_scale = 2.0 * 1.0 / N; /* winAdj = N / sum(win) = 1 for the rectangular window, K:373 + K:391 */
  cfg.cumu_mode = KSA_CUMU_AVG;
  cfg.gain = (float)gain;
  cfg.min_amp = (float)((1.0 / 256) * 0.00001);
  cfg.hm_width = XRES;
  cfg.max_frames = 8;
  cfg.u8_offset = 127.5f;
  cfg.u8_scale = 127.5f;
  cfg.scan_total_entries = 2 * N; /* a scan over two sampling-rate bands: 3 tuned bands at scanRangeNonOverlap 0.5 */
  cfg.scan_hop = N / 2;
  cfg.scan_hm_width = XRES;
  if (ksa_abi_version() != KSA_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  ksa_engine* e = NULL;
  CHECK(ksa_create(&cfg, &e));

  /* sdr_curscan drop-in (K:351-397): linear, fftshifted */
  CHECK(ksa_curscan_c64(e, iq, mag));
  int peak = 0;
  for (i = 1; i < N; ++i) if (mag[i] > mag[peak]) peak = i;
  const int want_bin = (kbin + N / 2) % N;
  if (peak != want_bin || fabs(mag[peak] - 2 * amp) > 1e-5 * 2 * amp) {
    fprintf(stderr, "curscan: peak %d (want %d) level %g (want %g)\n", peak, want_bin, mag[peak], 2 * amp);
    return 1;
  }
  /* the frame loop body (K:464-484), three frames */
  for (i = 0; i < 3; ++i) CHECK(ksa_frame_c64(e, iq));
  int32_t hm_index = -1;
  int64_t seen = -1;
  CHECK(ksa_read_state(e, cur, mx, mn, av, hm, &hm_index, &seen));
  const double want_db = 10 * log10(2 * amp) - gain;
  if (seen != 3 || hm_index != 3 || fabs(cur[want_bin] - want_db) > 1e-3 || fabs(mx[want_bin] - want_db) > 1e-3 ||
      fabs(av[want_bin] - want_db) > 1e-3 || fabs(hm[2 * XRES + want_bin / (N / XRES)] - want_db) > 1e-3) {
    fprintf(stderr, "zeroSpan state: frames %lld hm_index %d cur %g want %g\n", (long long)seen, hm_index, cur[want_bin], want_db);
    return 1;
  }
  /* one scan pass from host memory (K:621-668, K:696-697): three tuned bands, the middle one failed to tune */
  static float blocks[3 * 2 * FULL], scur[2 * N], savg[2 * N], shm[KSA_HM_ROWS * XRES];
  const uint8_t ok[3] = {1, 0, 1};
  for (i = 0; i < 3; ++i) memcpy(blocks + (size_t)i * 2 * FULL, iq, sizeof iq);
  CHECK(ksa_scan_pass_c64(e, blocks, 3, ok));
  int64_t passes = 0;
  CHECK(ksa_scan_read_state(e, scur, NULL, NULL, savg, shm, &hm_index, &passes));
  /* band 2 covers elements [N, 2N); its upper half is covered by no other band, and the peak (shifted bin 612 >= N/2) sits
   * there: element N + want_bin holds the tone's level.  Element want_bin itself lies where band 0's upper half was averaged
   * with the dummy band 1 (ones -> 10*log10(1) - gain, K:637-641, K:649): (level + (-gain)) / 2. */
  const double want_mix = (want_db + (0.0 - gain)) / 2;
  if (passes != 1 || hm_index != 1 || fabs(scur[N + want_bin] - want_db) > 1e-3 || fabs(scur[want_bin] - want_mix) > 1e-3 ||
      fabs(savg[N + want_bin] - want_db) > 1e-3) {
    fprintf(stderr, "scan state: passes %lld cur %g / %g want %g / %g\n", (long long)passes, scur[N + want_bin], scur[want_bin], want_db, want_mix);
    return 1;
  }
  /* device-side Levels decimation (K:205-221) */
  static float lv[4 * XRES];
  CHECK(ksa_read_levels(e, 0, 1 /* MAX */, XRES, lv));
  if (fabs(lv[want_bin / (N / XRES)] - want_db) > 1e-3) { fprintf(stderr, "levels\n"); return 1; }
  ksa_destroy(e);
  printf("c client ok: peak bin %d, %.4f dB\n", peak, cur[want_bin]);
  return 0;
}
