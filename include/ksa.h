/* libksa -- C ABI of the MI355X (gfx950) overlapped windowed-FFT spectrum / waterfall engine.
 *
 * Drop-in boundary for the numpy.fft hot path of hanishkvc/prgs-sdr-kspecanal.  The reference has
 * no FFI of its own: its de-facto seam is the module-level callable `sdr_curscan(d)` of
 * python/kspecanal.py (defined :351, called :464 / :523 / :636, rebound :531 / :543) plus the frame
 * and scan accumulate blocks that consume its result (:464-484, :636-668, :696-697).  Every entry
 * point below cites the reference lines it replaces ("K:" = python/kspecanal.py).
 *
 * Conventions: plain C types only; 0 = success, non-zero = error with text in ksa_last_error()
 * (thread local).  "host" pointers are ordinary CPU memory, "dev" pointers are HIP device memory of
 * the engine's device.  One engine = one GPU; no concurrent calls on one engine (the reference is
 * single threaded: K:505, K:1118-1123).  All device work is enqueued on the engine's stream
 * (ksa_set_stream) and host-pointer entry points synchronise that stream before returning.
 * Every entry point selects its engine's device for its own duration and hands the caller's current
 * HIP device back on return.  The library reads no environment variable.
 * Device output buffers (spectra, waterfall rows) must be 16-byte aligned; IQ buffers sample aligned.
 */
#ifndef KSA_H
#define KSA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSA_ABI_VERSION 4 /* 2: ksa_set_adj takes its target; allreduce_state, host-pointer scan pass, sharded scan entries
                             3: ksa_scan_spectra_dev, ksa_read_hm_rows, ksa_read_view; entry points restore the caller's current device
                             4: ksa_prof_clock
                             A binding takes the number from ksa_abi_version() of the library it loaded, never from a literal. */
#define KSA_HM_ROWS 128 /* waterfall history depth: maxHM K:448, fftHMMax K:611 */

/* d['curScanCumuMode'] K:31-34, K:58, consumed by data_cumu K:124-147 */
enum { KSA_CUMU_RAW = 0, KSA_CUMU_AVG = 1, KSA_CUMU_MAX = 2, KSA_CUMU_MIN = 3 };
/* IQ sample formats: complex64 (what sdr.read_samples hands over, narrowed from K:335's complex128)
 * and the dongle's native interleaved uint8 I,Q (pyrtlsdr packed_bytes_to_iq; K:301, K:339, K:346) */
enum { KSA_FMT_C64 = 0, KSA_FMT_U8 = 1 };
/* what the spectrum kernel writes per frame */
enum {
  KSA_OUT_LINEAR = 0,  /* sdr_curscan's return value: linear magnitude, fftshifted (K:391-396) */
  KSA_OUT_DB = 1,      /* zeroSpan: LogNoGain, -inf kept (K:469, K:106-112) */
  KSA_OUT_DB_CLIP = 2  /* scan: Clip2MinAmp then LogNoGain (K:640-641) */
};

typedef struct ksa_engine ksa_engine;

typedef struct ksa_config {
  int32_t abi_version;          /* KSA_ABI_VERSION */
  int32_t device;               /* HIP device ordinal */
  int32_t fft_size;             /* d['fftSize']: power of two, 16 .. 1048576 */
  int32_t full_size;            /* d['fullSize'] K:926-929: samples per captured block */
  int32_t num_windows;          /* windows actually transformed (K:385-390) */
  const int32_t* window_starts; /* host[num_windows]: iStart = int(i*fftSize*nonOverlap) K:386 */
  const float* window;          /* host[fft_size]: d['theWin'] K:932-936 */
  double mag_scale;             /* 2*winAdj/fftSize with winAdj = N/sum(win): K:373, K:391 */
  int32_t cumu_mode;            /* KSA_CUMU_*: within-block fold K:392-395 */
  float gain;                   /* d['gain'] subtracted by LogNoGain K:109 */
  float min_amp;                /* d['minAmp4Clip'] K:53, K:101 */
  int32_t hm_width;             /* d['PltHeatMapWidth'] K:449-455 (zeroSpan) ; must divide fft_size */
  int32_t max_frames;           /* largest batch handed to ksa_frames_dev / ksa_curscan_dev / steps per scan pass */
  float u8_offset, u8_scale;    /* uint8 unpack (b - offset) / scale; 127.5 / 127.5 by default */
  /* scan mode (0 = zeroSpan only): _scan_range K:568-698 */
  int32_t scan_total_entries;   /* totalEntries = numGroups*fftSize K:599-600 */
  int32_t scan_hop;             /* fftSize*scanRangeNonOverlap (validated integral, K:591-593) */
  int32_t scan_hm_width;        /* d['xRes'] : width of the per-pass waterfall row K:614, K:697 */
} ksa_config;

/* ---- lifetime ------------------------------------------------------------------------------- */
int ksa_abi_version(void);
const char* ksa_last_error(void);
/* Allocates tables, state and scratch on cfg->device.  Replaces the per-run setup of K:926-936. */
int ksa_create(const ksa_config* cfg, ksa_engine** out);
void ksa_destroy(ksa_engine* e);
/* hip_stream: a hipStream_t (NULL = the device's default stream).  When the stream changes, work already
 * enqueued on the old one is ordered in front of whatever is enqueued on the new one (event wait): a stream handed in
 * here must therefore stay alive until the engine's next ksa_set_stream (or ksa_destroy). */
int ksa_set_stream(ksa_engine* e, void* hip_stream);
/* Waits for everything the engine has enqueued on its stream, on the ENGINE's device, whatever device is current in the
 * caller. */
int ksa_synchronize(ksa_engine* e);

/* ---- sdr_curscan drop-ins (K:351-397) ------------------------------------------------------- */
/* One captured block in, linear fftshifted magnitudes out (host memory both sides). */
int ksa_curscan_c64(ksa_engine* e, const float* iq_host /* re,im interleaved [2*full_size] */,
                    float* mag_host /* [fft_size] */);
int ksa_curscan_u8(ksa_engine* e, const uint8_t* iq_host /* I,Q interleaved [2*full_size] */,
                   float* mag_host /* [fft_size] */);
/* Batched, device resident: nframes blocks spaced frame_stride samples apart -> out_dev[nframes][N]
 * in out_mode units.  nframes <= max_frames.  No state is touched. */
int ksa_curscan_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride,
                    int32_t nframes, int32_t out_mode, float* out_dev);

/* ---- zeroSpan frame accumulate (K:464-484), fused with curscan ------------------------------- */
/* Process `nframes` blocks in order: Cur/Max/Min/Avg + the 128-row waterfall ring.  The batch is
 * frames [first_index, first_index+nframes) of a logical run of total_frames frames (single GPU:
 * first_index 0, total_frames = nframes); AVG uses the closed form of the reference's (a+x)/2
 * recursion so that shards on several GPUs can be summed (ksa_partial_dev / ksa_commit).
 * cur_db_dev (optional, [nframes][N]) receives every frame's dB spectrum; hm_rows_dev (optional,
 * [nframes][hm_width]) every frame's waterfall row.  With commit != 0 the state is updated at once. */
int ksa_frames_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride,
                   int32_t nframes, int64_t first_index, int64_t total_frames,
                   float* cur_db_dev, float* hm_rows_dev, int32_t commit);
/* One block from host memory: the body of the reference's frame loop. */
int ksa_frame_c64(ksa_engine* e, const float* iq_host);
int ksa_frame_u8(ksa_engine* e, const uint8_t* iq_host);
/* zeroSpanPlay (K:547-564 feeding K:469-484): accumulate an already computed linear spectrum. */
int ksa_frame_spectrum(ksa_engine* e, const float* mag_host /* [fft_size], fftshifted */);
/* Partial block of the last ksa_frames_dev(commit=0): float[4][N] = {max, cur-or--inf, -min, sum}
 * on the device -- all-reduce rows 0-2 with MAX (the minimum travels negated) and row 3 with SUM across
 * the ranks that share a run, then ksa_commit on every rank. */
int ksa_partial_dev(ksa_engine* e, float** partial_dev);
int ksa_commit(ksa_engine* e, int64_t total_frames);
/* One-collective form of the same merge.  The partial block and the waterfall ring are one contiguous device
 * block, float[4*N + KSA_HM_ROWS*hm_width] (`nfloats`): all-gather it across the ranks (rank order) and hand
 * the gathered float[world][nfloats] to ksa_merge_gathered_dev, which reduces the partial rows (sum in rank
 * order: the same bits on every rank), takes every ring row from the rank that holds the newest frame mapped
 * to it (global frame f of the run lives in ring row (hm_index0 + f) % 128; rank r holds frames
 * [r*frames_per_rank, (r+1)*frames_per_rank)), and commits world*frames_per_rank frames.  Each rank must have
 * set its ring position to (hm_index0 + rank*frames_per_rank) % 128 (ksa_set_hm_index) before its
 * ksa_frames_dev(commit=0); afterwards the ring position is (hm_index0 + world*frames_per_rank) % 128. */
int ksa_exchange_dev(ksa_engine* e, float** xchg_dev, int64_t* nfloats);
int ksa_merge_gathered_dev(ksa_engine* e, const float* gathered_dev, int32_t world, int32_t frames_per_rank,
                           int32_t hm_index0);
/* SURVEY 8(b) `allreduce_state(handles[], n)`: the same merge for n engines of ONE process (one per GPU of the
 * node, or several on one GPU), no torch / RCCL needed -- the reference is a single process (K:1139-1155).  Engine r
 * must hold the uncommitted batch of frames [r*frames_per_rank, (r+1)*frames_per_rank) of a run of n*frames_per_rank
 * frames (ksa_set_hm_index((hm_index0 + r*frames_per_rank) % 128), then ksa_frames_dev(first_index = r*frames_per_rank,
 * total_frames = n*frames_per_rank, commit = 0)).  Every engine receives every exchange block (4N + 128W floats:
 * 320 KiB at fftSize 4096 -- latency, not bandwidth) by device-to-device / peer copies on its own stream, ordered
 * behind the producers by events, and runs the merge kernel: afterwards all n engines hold the same Cur/Max/Min/Avg
 * and waterfall ring, bit for bit (K:470-476, K:480-484 over the whole run).  Asynchronous like ksa_frames_dev. */
int ksa_allreduce_state(ksa_engine* const* handles, int32_t n, int32_t frames_per_rank, int32_t hm_index0);
/* GUI toggles bDataMax/bDataMin/bDataAvg (K:71-73, K:471-476) */
int ksa_set_flags(ksa_engine* e, int32_t b_max, int32_t b_min, int32_t b_avg);
/* d['Fft.Adj'] subtracted before the waterfall row and in ksa_read_levels / ksa_read_highs (K:400-411, K:478-480,
 * K:669, K:697).  scan = 0: the zeroSpan baseline, n == fft_size; scan != 0: the scan baseline, n ==
 * scan_total_entries (a one-band scan has both lengths equal: the target is never guessed from n).
 * adj_host == NULL clears that target. */
int ksa_set_adj(ksa_engine* e, int32_t scan, const float* adj_host, int32_t n);
int ksa_reset_state(ksa_engine* e);
/* Any pointer may be NULL.  cur..avg: [fft_size]; hm: [KSA_HM_ROWS][hm_width]. */
int ksa_read_state(ksa_engine* e, float* cur, float* max, float* min, float* avg, float* hm,
                   int32_t* hm_index, int64_t* frames_seen);
/* Device addresses of the persistent state (float[4][N]: cur,max,min,avg) and the ring. */
int ksa_state_dev(ksa_engine* e, float** state_dev, float** hm_ring_dev);
int ksa_set_hm_index(ksa_engine* e, int32_t hm_index);

/* ---- scan stitch + accumulate (K:621-668, K:696-697) ----------------------------------------- */
/* One full pass: nsteps tuned bands, block s at iq_dev + s*frame_stride.  step_ok (host, optional):
 * 0 marks a band whose tune failed -> dummy ones (K:637-639). */
int ksa_scan_pass_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride,
                      int32_t nsteps, const uint8_t* step_ok);
/* The same pass from caller-owned HOST memory -- the body of the reference's step loop K:621-668 plus the row of
 * K:696-697 with no device buffer on the caller's side: iq_host = [nsteps][full_size] samples (complex64 re,im or
 * uint8 I,Q), staged through an engine-owned device buffer (nsteps <= max_frames).  Synchronises before returning. */
int ksa_scan_pass_c64(ksa_engine* e, const float* iq_host, int32_t nsteps, const uint8_t* step_ok);
int ksa_scan_pass_u8(ksa_engine* e, const uint8_t* iq_host, int32_t nsteps, const uint8_t* step_ok);
/* Same, from per-step dB spectra already on the device ([nsteps][N], KSA_OUT_DB_CLIP units). */
int ksa_scan_stitch_dev(ksa_engine* e, const float* step_db_dev, int32_t nsteps);
/* A batch of `npasses` captured passes resident in HBM (block of pass p, step s at iq_dev + (p*nsteps + s)*
 * frame_stride): the spectrum stage runs once over all npasses*nsteps blocks (<= max_frames), then every element
 * of the stitched range walks the passes in order -- the same state as npasses ksa_scan_pass_dev calls, and the
 * last 128 passes' waterfall rows.  step_ok: host[npasses*nsteps] or NULL. */
int ksa_scan_passes_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride,
                        int32_t nsteps, int32_t npasses, const uint8_t* step_ok);
int ksa_scan_stitch_passes_dev(ksa_engine* e, const float* step_db_dev /* [npasses][nsteps][N] */,
                               int32_t nsteps, int32_t npasses);
/* Spectrum stage of a scan alone (K:636-641): nframes capture blocks spaced frame_stride samples apart -> out_dev
 * [nframes][N] after Clip2MinAmp + LogNoGain(infTo = 0); step_ok (host [nframes] or NULL): 0 marks a block whose tune
 * failed -> the dummy band ones(fftSize) through the same two steps (K:637-639), written by the library on the engine's
 * stream.  What a band-sharded driver calls per share of bands before ksa_scan_stitch_range_dev. */
int ksa_scan_spectra_dev(ksa_engine* e, const void* iq_dev, int32_t fmt, int64_t frame_stride, int32_t nframes,
                         const uint8_t* step_ok, float* out_dev);
/* Band-sharded scan (SURVEY 8e "freq-band"): this engine owns the tuned bands [step_lo, step_hi) of every pass and
 * the elements [elem_lo, elem_hi) of the stitched range (normally [step_lo*hop, step_hi*hop), the last rank up to
 * totalEntries).  own_db_dev = [npasses][step_hi-step_lo][N] dB spectra of its bands (own_band_major != 0:
 * [step_hi-step_lo][npasses][N], what one strided ksa_curscan_dev launch per band produces -- it lets a driver transform
 * the bands its neighbours wait for first and send them while the others are still being transformed); halo_db_dev =
 * [npasses][nhalo][N] spectra of the nhalo bands in front of step_lo that still cover owned elements (the overlap
 * K:645-650 averages: nhalo = ceil(N/hop) - 1 bands, 1 at the usual hop of N/2), received from the left neighbour.
 * Runs exactly the updates of K:643-668 on the owned elements only, and writes this engine's PARTIAL waterfall rows
 * (cell maxima over the owned elements, -inf where it owns none of a cell; K:696-697) for the batch's last
 * min(npasses,128) passes to rows_dev [rows][scan_hm_width] (engine scratch, see ksa_scan_rows_dev). */
int ksa_scan_stitch_range_dev(ksa_engine* e, const float* own_db_dev, int32_t own_band_major, const float* halo_db_dev,
                              int32_t nhalo, int32_t step_lo, int32_t step_hi, int32_t nsteps, int32_t npasses,
                              int32_t elem_lo, int32_t elem_hi);
/* The partial rows of the last ksa_scan_stitch_range_dev: float[rows][scan_hm_width] on the device. */
int ksa_scan_rows_dev(ksa_engine* e, float** rows_dev, int32_t* rows);
/* gathered_dev = every rank's partial rows, [world][rows][scan_hm_width] in rank order: cell = NaN-propagating max
 * over the ranks, stored into the ring rows of the batch's last `rows` passes; advances the ring by npasses. */
int ksa_scan_merge_rows_dev(ksa_engine* e, const float* gathered_dev, int32_t world, int32_t rows, int32_t npasses);
/* Single-process form of the whole exchange for n engines (one per GPU, or several on one): engine r has run the
 * spectrum stage of its bands into own_db[r] ([npasses][hi_r-lo_r][N], device memory of engine r's GPU), bands split
 * as [nsteps*r/n, nsteps*(r+1)/n).  Copies the halos between neighbours (peer copies), runs the range stitch on
 * every engine, gathers and merges the partial waterfall rows.  Afterwards every engine holds its slice of
 * Cur/Max/Min/Avg and the complete ring; ksa_scan_gather_state assembles the curves on the host. */
int ksa_scan_allstitch(ksa_engine* const* handles, int32_t n, float* const* own_db_dev, int32_t nsteps, int32_t npasses);
/* Host curves [scan_total_entries] assembled from the n engines' owned slices (any pointer may be NULL). */
int ksa_scan_gather_state(ksa_engine* const* handles, int32_t n, int32_t nsteps, float* cur, float* max, float* min,
                          float* avg);
int ksa_scan_read_state(ksa_engine* e, float* cur, float* max, float* min, float* avg, float* hm,
                        int32_t* hm_index, int64_t* passes);
int ksa_scan_state_dev(ksa_engine* e, float** state_dev, float** hm_ring_dev);
int ksa_scan_reset(ksa_engine* e);
/* d['bScanRangeBaseDataIsRaw'] (K:568, K:651-662): Max/Min/Avg fold every tuned band's own spectrum over
 * [iStart:iEnd] instead of the stitched Fft.Cur over [iStart:iDone]. */
int ksa_scan_set_base_is_raw(ksa_engine* e, int32_t on);

/* ---- plot-side decimation (data_plotcompress / _data_plotcompress, K:168-221) ------------------------------- */
/* The four curves (cur, max, min, avg; minus Fft.Adj if set, K:400-411) reduced on the device to `cells`
 * groups with pltCompress AVG (0) / MAX (1) / MIN (2): out_host[4][cells].  scan != 0 reads the scan state. */
int ksa_read_levels(ksa_engine* e, int32_t scan, int32_t mode, int32_t cells, float* out_host);

/* Peak markers of plot_highs (K:243-272) chosen on the device from the same decimated curve: walk `curve`
 * (0 cur, 1 max, 2 min, 3 avg; reduced to `cells` groups with `mode` as above) from its highest level down and
 * mark a cell unless an already marked one is closer than min_sep_cells (= pltHighsDelta4Marking * (x[-1]-x[0]) /
 * cell width, K:249-250, K:261-262), until `count` (<= 64) are marked (K:268-269).  As in the reference NaN sorts
 * above +inf and the lowest point is never visited (K:258).  Equal levels are taken higher index first; the reference's
 * order among equal levels is undefined (numpy's unstable argsort, K:251) and its spacing test runs on float64
 * frequencies, so tie order and a spacing of exactly min_sep_cells are parity-unpinned.
 * idx_host / lvl_host: [count]; *found = cells marked. */
int ksa_read_highs(ksa_engine* e, int32_t scan, int32_t mode, int32_t cells, int32_t curve, double min_sep_cells,
                   int32_t count, int32_t* idx_host, float* lvl_host, int32_t* found);

/* Rows [row0, row0 + nrows) (mod 128) of the waterfall ring (scan != 0: the scan's ring) -> out_host[nrows][width]:
 * the one new row of a frame / pass instead of the whole 128-row buffer (K:480-481, K:697, K:729). */
int ksa_read_hm_rows(ksa_engine* e, int32_t scan, int32_t row0, int32_t nrows, float* out_host);

/* The per-frame plot hand-off in ONE call and one synchronisation (K:477-504 / K:669-697): the decimated curves
 * (as ksa_read_levels -> levels_host[4][cells]), the peak markers of `curve` (as ksa_read_highs; count = 0 skips them)
 * and the newest hm_rows rows of the waterfall ring, oldest first (-> hm_rows_host[hm_rows][width]; 0 skips them);
 * *hm_index = ring position after the newest row.  Only cells-sized arrays cross PCIe: SURVEY 8 row f2. */
int ksa_read_view(ksa_engine* e, int32_t scan, int32_t mode, int32_t cells, float* levels_host, int32_t curve,
                  double min_sep_cells, int32_t count, int32_t* idx_host, float* lvl_host, int32_t* found,
                  int32_t hm_rows, float* hm_rows_host, int32_t* hm_index);

/* ---- pinned host memory for capture blocks (optional: any host pointer works, pinned ones copy faster) ------------ */
int ksa_host_alloc(void** out, int64_t bytes);
int ksa_host_free(void* p);

/* ---- measurement ------------------------------------------------------------------------------ */
/* HIP-event timing of the spectrum kernel on the engine's stream: enable, run, then read the sum
 * of kernel durations and the launch count since the last enable. */
int ksa_prof_enable(ksa_engine* e, int32_t on);
int ksa_prof_read(ksa_engine* e, double* spectrum_ms, int64_t* launches);
/* Shader clock the chip held under the profiled spectrum stages since the last ksa_prof_enable(1): a stamp kernel on
 * the engine's stream directly before and directly after each profiled stage (outside the event pair of ksa_prof_read)
 * reads s_memtime (shader cycles) and s_memrealtime (100 MHz) on most CUs; *shader_ghz is the median of d(cycles) /
 * d(time) over the CUs stamped at both ends and the last <= 16 stages, *ghz_min / *ghz_max (optional) the extremes,
 * *samples the (CU, stage) pairs behind them.  The spectrum kernels themselves carry no stamp; an unprofiled run launches
 * nothing extra. */
int ksa_prof_clock(ksa_engine* e, double* shader_ghz, double* ghz_min, double* ghz_max, int64_t* samples);
/* Resources of the spectrum kernel chosen for this engine (for DESIGN.md / bench.py). */
int ksa_kernel_info(ksa_engine* e, int32_t* threads, int32_t* lds_bytes, int32_t* vgprs,
                    int32_t* grid, int32_t* path /* 0 = single-workgroup LDS FFT, 16 points per thread; 3 = the same with
                                                    32 points per thread (N = 8192, 16384); 2 = radix-16 / 32 / 64 first
                                                    stage + single-workgroup FFT of N/16 (N = 32768 .. 262144), N/32
                                                    (524288) or N/64 (1048576) points; 4 = N = 1024 .. 4096: large
                                                    batches run two frames per workgroup in packed fp32 (lds / vgprs /
                                                    grid then describe that kernel), small ones the path-0 kernel;
                                                    5 = N = 64: complex64 input runs the 8 x 8 plan with adjacent-sample
                                                    loads (lds / vgprs / grid describe it), uint8 input the path-0 kernel */);

#ifdef __cplusplus
}
#endif
#endif /* KSA_H */
