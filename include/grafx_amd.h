/* grafx_amd — C ABI of the MI355X-native hot path of sh-lee97/grafx.
 *
 * The reference is pure Python/PyTorch: its "FFI" for this path is the set of
 * native calls its processors make through torch.  Each entry
 * point below replaces one of those call sites; the file:line cited is the
 * reference interface it stands in for (paths relative to
 * /root/reference/src/grafx/processors unless noted).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless marked "host";
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     callers pass every workspace (sizes from the *_bytes queries).  ONE
 *     exception: the first call on a device of any entry point built on the FFT
 *     tile (fftconv, fir_spectrum, fir_grad, iir_fsm, odd_alias) allocates that
 *     device's 80 KB twiddle table with hipMalloc, fills it on `stream` and waits
 *     for it (hipStreamSynchronize) under a process-wide mutex; the table lives
 *     for the process.  The first fftconv / fir_grad call on a device also loads
 *     the code object of the hand-scheduled kernels (GFX_SCHED_PIPE; gfx950
 *     assembly embedded in the library, hipModuleLoadData: device memory for its
 *     ~0.5 MB of code).  After those first calls nothing allocates or
 *     synchronises (make them outside a stream capture).
 *     The device is the CURRENT device (hipGetDevice): make the tensors' device
 *     current before calling (grafx_amd/ops.py does);
 *   - return 0 on success, a negative GFX_E* code on error; never throws;
 *   - signals are fp32; a "row" is one (batch x node) item, channels inside.
 *
 * Row addressing (gfx_rowmap_t): element (r, c, n) of a signal lives at
 *     base + (r / inner) * stride_outer + (r % inner) * stride_inner
 *          + c * stride_ch + n                                   [floats]
 * so kernels read and write slices of render_grafx's (B, V, C, L) signal
 * buffer in place (inner = nodes of this type, stride_outer = V*C*L,
 * stride_inner = C*L) as well as plain contiguous (R, C, L) tensors
 * (inner = R, stride_inner = C*L).
 */
#ifndef GRAFX_AMD_H
#define GRAFX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFX_OK 0
#define GFX_EINVAL -1   /* bad argument (shape, null pointer, unsupported size) */
#define GFX_ENOSPC -2   /* workspace too small */
#define GFX_ELAUNCH -3  /* HIP launch failed */

typedef struct {
    int64_t inner;
    int64_t stride_outer;
    int64_t stride_inner;
    int64_t stride_ch;
} gfx_rowmap_t;

/* library / device sanity: returns the ABI version (>0); fills *n_cu if non-null (host ptr). */
int gfx_abi_version(void);
int gfx_device_info(int* n_cu, size_t* lds_bytes);

/* ---- FIR convolution core --------------------------------------------------------------
 * replaces convolve(): core/convolution.py:119-134 (torch.fft.rfft x2, irfft), and
 * FIRConvolution._native_forward: core/convolution.py:82-83.
 *
 * y[r, c, n] = sum_k h[r, c_f, k] * x[r, c_x, n + off - k],  n in [0, Lout)
 * (x is zero outside [0, L); channel dims broadcast 1<->C like torch).
 * off = 0, Lout = L            -> mode "causal"
 * off = N/2, Lout = L          -> mode "zerophase"
 * off = 0, Lout = L + N - 1    -> the full linear convolution
 * Overlap-save on 16384-sample LDS FFT tiles; filters longer than 8193 taps are
 * split into 8192-tap partitions (frequency-domain delay line).
 *
 * Step 1: gfx_fir_spectrum_f32 turns the taps h (RCf rows of N, contiguous) into the
 * tile spectra `Hs` (private layout, gfx_fir_spectrum_bytes bytes).  `gain` (nullable) scales
 * row-channel rc by gain[rc / gain_div].
 * Step 2: gfx_fftconv_f32 streams x through the tiles.
 */
int64_t gfx_fftconv_nparts(int64_t N);
size_t gfx_fir_spectrum_bytes(int64_t RCf, int64_t N);
size_t gfx_fftconv_workspace_bytes(int64_t R, int64_t C_in, int64_t L, int64_t Lout, int64_t off, int64_t N);
int gfx_fir_spectrum_f32(const float* h, const float* gain, int64_t gain_div, void* Hs,
                         int64_t RCf, int64_t N, void* stream);
int gfx_fftconv_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, float* y, gfx_rowmap_t ymap,
                    int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N,
                    void* ws, size_t ws_bytes, void* stream);
/* Same, and additionally copies the input rows to `xcopy` (rows addressed by `cmap`, same R x C_in x L) from
 * the registers that already hold them.  render_grafx keeps every node's signal in one buffer
 * (render/graph.py:104-106: `signal_buffer[:, :num_sources] = input_signals`); when the first stage is a
 * convolution this folds that copy into the stage's kernel.  Only for the causal single-partition case
 * (off == 0, Lout >= L -- the full-length convolution the odd-length aliasing starts from included --, C_in == max(C_in, C_f),
 * N <= 8193); anything else returns GFX_EINVAL. */
int gfx_fftconv_tee_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, float* y, gfx_rowmap_t ymap,
                        float* xcopy, gfx_rowmap_t cmap,
                        int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N,
                        void* ws, size_t ws_bytes, void* stream);

/* The general form of the two above.  `h_rows` <= R: row r convolves with filter (r % h_rows) -- rows are
 * batch-major (r = b * nodes + node), so h_rows = nodes shares one filter per node across the batch, which is
 * what render_grafx's 4-D path means by un-batched parameters (render/graph.py:68-75 expands them B times;
 * here the spectra are built once per node and every batch row reads them).  xcopy may be null (no tee).
 * `part_len`: 0 for the default filter partitioning, or the value of gfx_fftconv_part_len(N, Lout) -- longer
 * partitions (fewer of them, fewer signal windows) for "long filter, short output" problems such as the filter
 * gradient of a training step (N = signal length taps, Lout = filter taps); the spectra must then come from
 * gfx_fir_spectrum_ex_f32 with the same part_len, sizes from the *_ex queries. */
int64_t gfx_fftconv_part_len(int64_t N, int64_t Lout);
size_t gfx_fir_spectrum_bytes_ex(int64_t RCf, int64_t N, int64_t part_len);
size_t gfx_fftconv_workspace_bytes_ex(int64_t R, int64_t C_in, int64_t L, int64_t Lout, int64_t off, int64_t N,
                                      int64_t part_len);
int gfx_fir_spectrum_ex_f32(const float* h, const float* gain, int64_t gain_div, void* Hs,
                            int64_t RCf, int64_t N, int64_t part_len, void* stream);
/* Spectra of the time-reversed signal rows: filter (r, c) has the L taps x[r, c, L-1-k], read in place through
 * `xmap` -- the "filter" of the filter-gradient correlation (autograd of convolve(): grad_h[k] = sum_n g[n] x[n+off-k])
 * without materialising x.flip(-1).  Buffer size: gfx_fir_spectrum_bytes_ex(R * C, L, part_len). */
int gfx_fir_spectrum_rev_f32(const float* x, gfx_rowmap_t xmap, int64_t R, int64_t C, int64_t L, int64_t part_len,
                             void* Hs, void* stream);
/* Filter gradient of a short-filter convolve() (autograd of core/convolution.py:119-134), N <= 8193 taps:
 *   gh[r, c, k] = sum_n g[r, c_g, n] x[r, c_x, n + off - k],  k in [0, N),  x zero outside [0, L), g outside [0, Lg)
 * (channels broadcast 1 <-> 2; gh is (R, max(C_x, C_g), N) contiguous).  One pass over x and g, no workspace. */
int gfx_fir_grad_f32(const float* x, gfx_rowmap_t xmap, const float* g, gfx_rowmap_t gmap, float* gh,
                     int64_t R, int64_t C_x, int64_t C_g, int64_t L, int64_t Lg, int64_t N, int64_t off, void* stream);
int gfx_fftconv_ex_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len,
                       float* y, gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap,
                       int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N,
                       void* ws, size_t ws_bytes, void* stream);

/* gfx_fftconv_ex_f32 with an explicit kernel schedule for filters of N <= 8193 taps (same results to rounding):
 *   GFX_SCHED_TILE  one 16384-sample tile per 256-thread workgroup (fftconv1_kernel, compiler-scheduled).
 *   GFX_SCHED_PIPE  the hand-scheduled persistent form of the same tile (generated gfx950 assembly, explicit register
 *                   allocation: the next tile's window, the filter spectrum and the output stores are interleaved
 *                   with the arithmetic of the running tile).  GFX_EINVAL where it does not apply.
 *   GFX_SCHED_AUTO  what gfx_fftconv_f32 / _tee_f32 / _ex_f32 use: the faster of the two for the shape at hand.
 * For longer filters (partitioned convolution: xspec_kernel + a product kernel) GFX_SCHED_TILE selects one output tile per
 * 256-thread workgroup (macinv_kernel) and GFX_SCHED_AUTO two consecutive ones per 512-thread workgroup (macinv_pair_kernel:
 * each window spectrum and filter partition fetched once per pair; equal to a few units in the last place of the largest
 * output); GFX_SCHED_PIPE: GFX_EINVAL.
 * (Round-2 experiments -- ping-pong, half-size exchanges, 512-thread tile -- live in tools/experiments/r2_schedules.) */
#define GFX_SCHED_AUTO 0
#define GFX_SCHED_TILE 1
#define GFX_SCHED_PIPE 2
int gfx_fftconv_sched_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len,
                          float* y, gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap,
                          int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N,
                          void* ws, size_t ws_bytes, int schedule, void* stream);

/* gfx_fftconv_sched_f32(GFX_SCHED_AUTO) that may also leave the bits of max |y| of every output row-channel in `rowmax`
 * (R * max(C_in, C_f) words, zeroed by the caller, row-major (row, channel)): the compiler-built tile kernels (one
 * partition, and the partitioned convolution's product kernels) take them as a by-product of their stores (one atomic
 * maximum per wave and tile) and set *rowmax_written = 1; the hand-scheduled kernel and the one-output-tile form leave the
 * words alone and *rowmax_written = 0.  For the full-length convolution
 * that feeds the odd-length aliasing (core/convolution.py:119-134), whose two-rows-per-transform form scales the second
 * row of a pair by these maxima (gfx_odd_alias_pair_max_f32): the separate pass over z is not needed then. */
int gfx_fftconv_rowmax_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len,
                           float* y, gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap,
                           int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N,
                           void* ws, size_t ws_bytes, uint32_t* rowmax, int* rowmax_written, void* stream);

/* Diagnostic: the name -- as rocprofv3's kernel trace prints it -- of the (dominant) kernel that the calling thread's last
 * successful gfx_fftconv_* call launched: "gfx_fftconv_pipe_t1_o8", "fftconv1_kernel<true>", "winmac_kernel",
 * "xspec_kernel+macinv_pair_kernel"; "" before the first call.  The string is static.  (bench.py labels its live per-launch
 * timings with it, so that the line's `roofline.kernel` is a name a profile of the same command contains.) */
const char* gfx_fftconv_last_kernel(void);

/* Short filters (N <= gfx_fir_direct_max_taps() = 512) as batched Toeplitz GEMMs on the fp32 matrix cores, taps given
 * directly (no spectra): the direct-form counterpart of gfx_fftconv_ex_f32 with the same meaning of every argument
 *   y[r, c, n] = sum_k h[r % h_rows, c_f, k] x[r, c_x, n + off - k],  n in [0, Lout),  x zero outside [0, L)
 * h is (h_rows * C_f, N) contiguous.  Replaces convolve() / FIRConvolution for short FIRs (core/convolution.py:85-134,
 * filter.py:34-39); exact fp32 (v_mfma_f32_16x16x4_f32).  MFMA-bound above ~64 taps, HBM-bound below. */
int64_t gfx_fir_direct_max_taps(void);
int gfx_fir_direct_f32(const float* x, gfx_rowmap_t xmap, const float* h, int64_t h_rows, float* y, gfx_rowmap_t ymap,
                       int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N,
                       void* stream);

/* ---- the reference's odd-length aliasing ----------------------------------------------------
 * convolve() (core/convolution.py:119-134) inverts a P-point spectrum (P = Lx + Lh - 1) with irfft's default length
 * 2 (P // 2): for odd P the result is  y = irfft_{P-1}(rfft_P(z))  of the linear convolution z -- what every reference
 * default length produces.  gfx_odd_alias_f32 computes y[:, lo : lo + len] from z (rows x P, contiguous) with two
 * chirp-z transforms on the LDS FFT tile (no FFT library, fp32); 3 <= P <= 11,184,811 odd (NFFT up to 64 x 32 x 8192), rows
 * <= 16383 per call, else the size queries return 0.  `plan` (per P: chirps and their spectra) comes from gfx_odd_alias_plan_f32, which needs a
 * workspace of gfx_odd_alias_workspace_bytes(1, P); the transform needs gfx_odd_alias_workspace_bytes(rows, P).
 * gfx_odd_alias_adjoint_f32 is the transposed map (the gradient the reference gets from differentiating its
 * rfft / irfft pair): gz (rows x P, contiguous) from gy (rows x len, row stride ldg), the gradient with respect to
 * y[:, lo : lo + len]; same plan, same workspace size. */
size_t gfx_odd_alias_plan_bytes(int64_t P);
size_t gfx_odd_alias_workspace_bytes(int64_t rows, int64_t P);
int gfx_odd_alias_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                      const void* plan, void* ws, size_t ws_bytes, void* stream);
/* The same with the output rows written straight into a signal addressed through a row map: row q = row0 + r of the call is
 * row q / C, channel q % C of `y` (a strided (B, n, C, len) view of render_grafx's signal buffer, render/core.py:80-98), so
 * that a stage on the aliasing path needs no copy of its result into the buffer.  `z` holds `rows` rows of this call. */
int gfx_odd_alias_rows_f32(const float* z, float* y, gfx_rowmap_t ymap, int64_t C, int64_t row0, int64_t lo, int64_t len,
                           int64_t rows, int64_t P, const void* plan, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_adjoint_f32(const float* gy, int64_t ldg, int64_t lo, int64_t len, float* gz, int64_t rows, int64_t P,
                              const void* plan, void* ws, size_t ws_bytes, void* stream);
/* The same maps with the transforms carried in double precision (fp32 data in and out, own plan and workspace, both
 * twice the size): for the energy envelope of the dynamics processors (core/envelope.py:34-49), whose aliased result
 * feeds log() and a gain curve -- the ~1e-6-of-peak noise floor of an fp32 transform pair is amplified beyond the
 * parity bound on quiet passages.  Same geometry limits and error codes. */
size_t gfx_odd_alias_precise_plan_bytes(int64_t P);
size_t gfx_odd_alias_precise_workspace_bytes(int64_t rows, int64_t P);
int gfx_odd_alias_precise_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_precise_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                              const void* plan, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_precise_adjoint_f32(const float* gy, int64_t ldg, int64_t lo, int64_t len, float* gz, int64_t rows,
                                      int64_t P, const void* plan, void* ws, size_t ws_bytes, void* stream);
/* The forward maps with TWO real rows per complex transform (csrc/czt_pair.hip): the pair z1 + i z2 goes through the
 * two chirp-z transforms as one complex row with the spectrum kept on both sides (P bins) and comes out as
 * A z1 + i A z2 -- 2P - 1 points per pair instead of (3P - 1) / 2 per row, a third fewer bytes through each of the same
 * passes.  Rows 2r and 2r + 1 of a call form a pair (an odd row count leaves the last row alone); the second row of a pair
 * goes through scaled by the exact power of two that brings it to the first row's binade (from max |z| of the rows: a pass
 * of the call itself, one word per row behind the workspace, or -- the _max forms -- words the producer of z left:
 * gfx_fftconv_rowmax_f32), so every row keeps an error relative to its own peak.  3 <= P <= 8 388 607 odd (2P - 1 <= 2^24 points), else the size queries return
 * 0 and the calls GFX_EINVAL (use the one-row forms above); own plan (gfx_odd_alias_pair_plan_f32, workspace of
 * gfx_odd_alias_pair_workspace_bytes(1, P)) and workspace (gfx_odd_alias_pair_workspace_bytes(rows, P)); for
 * gfx_odd_alias_pair_rows_f32 row0 must be even.  The `precise` forms carry the transforms in double (plan and
 * workspace twice the size).  Same results as the one-row forms up to rounding (tests/test_gpu_odd_alias_pair.py). */
size_t gfx_odd_alias_pair_plan_bytes(int64_t P);
size_t gfx_odd_alias_pair_workspace_bytes(int64_t rows, int64_t P);
int gfx_odd_alias_pair_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_pair_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                           const void* plan, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_pair_rows_f32(const float* z, float* y, gfx_rowmap_t ymap, int64_t C, int64_t row0, int64_t lo, int64_t len,
                                int64_t rows, int64_t P, const void* plan, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_pair_max_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                               const void* plan, void* ws, size_t ws_bytes, const uint32_t* rowmax, void* stream);
int gfx_odd_alias_pair_rows_max_f32(const float* z, float* y, gfx_rowmap_t ymap, int64_t C, int64_t row0, int64_t lo,
                                    int64_t len, int64_t rows, int64_t P, const void* plan, void* ws, size_t ws_bytes,
                                    const uint32_t* rowmax, void* stream);
size_t gfx_odd_alias_pair_precise_plan_bytes(int64_t P);
/* gfx_odd_alias_pair_precise_f32 with the rows' maxima given (`rowmax`, or NULL: taken by a pass of the call) and the envelope
 * smoother's relu (core/envelope.py:48) fused into the last pass (`relu` != 0): no separate clamp over the rows. */
int gfx_odd_alias_pair_precise_max_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                                       const void* plan, void* ws, size_t ws_bytes, const uint32_t* rowmax, int relu,
                                       void* stream);
size_t gfx_odd_alias_pair_precise_workspace_bytes(int64_t rows, int64_t P);
int gfx_odd_alias_pair_precise_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream);
int gfx_odd_alias_pair_precise_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                                   const void* plan, void* ws, size_t ws_bytes, void* stream);

/* ---- small inverse real DFT (parameter-side front-ends) ---------------------------------------
 * y = irfft(X, n) for any n <= 8192 as a direct sum (twiddles tabulated in LDS), K = n/2 + 1 bins per row, X complex
 * (rows, K, 2) or real (rows, K) when is_real; the result is rotated by `roll` and multiplied by `window` (n, nullable):
 *   y[row, (m + roll) mod n] = window[(m + roll) mod n] * irfft(X[row], n)[m]
 * Replaces the FFT-library calls of the zero-phase FIR design (core/fir.py:20-27: irfft, roll, window) and of the
 * surrogate delay (core/delay.py:73-76). */
int gfx_irdft_f32(const float* X, int is_real, float* y, int64_t rows, int64_t K, int64_t n, int64_t roll,
                  const float* window, void* stream);
/* The forward twin: X[row, k] = sum_m x[row, m] e^{-2 pi i k m / n}, k = 0 .. n/2 (torch.fft.rfft), X (rows, K, 2), any
 * n <= 8192, direct sum.  It is the adjoint of gfx_irdft_f32 up to the bin weights, i.e. the gradient of every
 * frequency-sampled front-end (core/iir.py:150 irfft(n=fsm_fir_len), reverb.py:176-184 istft frames): the training path's
 * parameter-side transforms stay off the FFT library. */
int gfx_rdft_f32(const float* x, float* X, int64_t rows, int64_t K, int64_t n, void* stream);

/* ---- frequency-sampled IIR -------------------------------------------------------------
 * replaces IIRFilter._process_fsm / iir_fsm / delay: core/iir.py:147-150, 263-276
 * (complex64 response of the biquad cascade on the N-point grid, then torch.fft.irfft(n=N)).
 * Bs, As: (RC, K, 3) contiguous; h: (RC, N) taps out.  N <= 4096 (Bluestein on the LDS tile).
 * `plan` is a per-N constant (gfx_iir_fsm_plan_bytes bytes) filled once by gfx_iir_fsm_plan_f32.
 */
/* 1 when gfx_iir_fsm_fir_f32 has a native kernel for fsm_fir_len = N: 1 <= N <= 4096 (Bluestein on one LDS tile; needs the
 * plan of gfx_iir_fsm_plan_f32) and N = 8192, 16384 (the tile's own inverse real transform; `plan` may be NULL). */
int gfx_iir_fsm_native(int64_t N);
size_t gfx_iir_fsm_plan_bytes(int64_t N);
int gfx_iir_fsm_plan_f32(void* plan, int64_t N, void* stream);
int gfx_iir_fsm_fir_f32(const float* Bs, const float* As, const void* plan, float* h,
                        int64_t RC, int64_t K, int64_t N, void* stream);
/* The same taps from coefficients given in DOUBLE precision: the cascade response is evaluated in double at the exact
 * sample points exp(-2 pi i d k / N) and rounded once -- for cascades whose float32 coefficients lose the filter (the
 * third-octave GraphicEqualizer's 9 Hz wide bands: 1 +- beta with beta = 6.5e-4; reference eq.py:339-436, core/geq.py),
 * where the reference's own float32 result is 1e-4 .. 4e-4 from a float64 evaluation of its formulas. */
int gfx_iir_fsm_fir_f64c_f32(const double* Bs, const double* As, const void* plan, float* h, int64_t RC, int64_t K,
                             int64_t N, void* stream);
/* Gradient of gfx_iir_fsm_fir_f32's taps with respect to the coefficients (what autograd derives from core/iir.py:147-152):
 * G = rfft(dL/dh, n = N) as (RC, N / 2 + 1) interleaved complex64 (gfx_rdft_f32), `delays` = the (3, N / 2 + 1) complex64
 * table e^(-j phi[d, k]) of the forward pass (float32 phases); gB / gA: (RC, K, 3), either may be NULL.  Double precision
 * inside (the sums over the bins cancel to 1e-5 of their terms); one launch instead of ~40 complex128 torch kernels. */
int gfx_iir_fsm_bwd_f32(const float* Bs, const float* As, const float* G, const float* delays, float* gB, float* gA,
                        int64_t RC, int64_t K, int64_t N, void* stream);

/* coefficient front-ends (elementwise over n = rows*channels items of K biquads)
 * gfx_peq_coeffs_f32    replaces ParametricEqualizer.forward's activations + RBJ formulas:
 *                       eq.py:291-314, filter.py:593-604, 645-656, 687-705, 736-754
 * gfx_biquad_coeffs_f32 replaces BiquadFilter.forward's stability activations: filter.py:144-153
 *                       (A0 nullable = "normalized" off)
 */
int gfx_peq_coeffs_f32(const float* w0, const float* q_inv, const float* log_gain, float* Bs, float* As,
                       int64_t n, int64_t K, int use_shelving, void* stream);
int gfx_biquad_coeffs_f32(const float* Bin, const float* A1_pre, const float* A2_pre, const float* A0,
                          float* Bs, float* As, int64_t n, void* stream);
/* autograd of gfx_peq_coeffs_f32: (dL/dBs, dL/dAs) (n, K, 3) -> dL/d(w0, q_inv, log_gain) (n, K), the chain rule
 * through the RBJ formulas and the activations in one elementwise pass (what torch autograd does with ~120 kernels). */
int gfx_peq_coeffs_bwd_f32(const float* w0, const float* q_inv, const float* log_gain, const float* gBs,
                           const float* gAs, float* gw0, float* gq_inv, float* glog_gain,
                           int64_t n, int64_t K, int use_shelving, void* stream);

/* ---- dynamics ---------------------------------------------------------------------------
 * gfx_dynamics_fused_f32 replaces Compressor.forward / NoiseGate.forward (dynamics.py:361-409,
 *   598-641) for energy_smoother in {None (smoother=0), "iir" (smoother=1)} and no gain
 *   smoother: energy -> truncated one-pole (exact recursive form, iir_len taps) -> log ->
 *   knee (0 hard, 1 quadratic, 2 exponential; gate=1 selects NoiseGate's curves) -> exp -> y.
 *   Per-row parameters are (R) arrays (the reference's (R,1) tensors).
 * The remaining entry points are the same stages as separate launches, for configurations
 * the fused kernel does not cover (ballistics, gain smoothing, odd-P compatibility):
 *   gfx_energy_f32       e = mean_c x^2                                   dynamics.py:390
 *   gfx_onepole_f32      TruncatedOnePoleIIRFilter on (R,L) rows -> (R,Lout), optional relu
 *                        core/envelope.py:34-60 (Lout = L + iir_len - 1 gives the full convolution)
 *   gfx_onepole_fir_f32  its taps h[n] = (1-a) exp(n log a)              core/envelope.py:51-60
 *   gfx_ballistics_*     Ballistics / torchcomp.compressor_core          core/envelope.py:84-101
 *                        (z_alpha: (R,2); the recursion is torchcomp's published one, the wheel is absent: see DESIGN.md)
 *   gfx_dyn_gain_f32     env -> gain: log(env+1e-5) -> knee [-> exp unless log_out]
 *   gfx_apply_gain_f32   y = (exp_gain ? exp(g) : g)[:,None,:] * x        dynamics.py:405
 *   gfx_stereo_gain_f32  StereoGain.forward                               stereo.py:38-41
 */
int gfx_dynamics_fused_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap,
                           const float* log_threshold, const float* log_ratio, const float* log_knee,
                           const float* z_alpha, int64_t R, int64_t C, int64_t L, int smoother,
                           int64_t iir_len, int knee, int gate, void* stream);
/* Same with parameters shared across the batch: row r reads parameter row (r % param_rows). */
int gfx_dynamics_fused_ex_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap,
                              const float* log_threshold, const float* log_ratio, const float* log_knee,
                              const float* z_alpha, int64_t param_rows, int64_t R, int64_t C, int64_t L,
                              int smoother, int64_t iir_len, int knee, int gate, void* stream);
/* The same with a workspace of gfx_dynamics_ws_bytes(param_rows) bytes (scratch for this call; may be NULL, which is
 * gfx_dynamics_fused_ex_f32) and an optional `u1` (R, L): the un-truncated smoother scan (1 - a) * U, kept for
 * gfx_dynamics_bwd_u1_f32.  With a workspace the smoothed configuration runs as dependency-free one-shot tiles: every
 * 1024-sample tile of every row is its own workgroup, which re-reads the H = ceil(log 1e-12 / log a) most recent
 * samples before the tile to rebuild the smoother state (exact to 1e-12 of the peak energy: the smoother is a FIR with
 * taps (1 - a) a^k) -- the access shape of a plain copy, where one workgroup per row is a set of long scattered
 * streams.  Rows whose history does not fit (H > 256, or a live a^N truncation term) are picked out on the device from
 * a per-row pole table and produced by the row kernel in the same call; no host synchronisation. */
size_t gfx_dynamics_ws_bytes(int64_t param_rows);
/* The workspace that also keeps rows with a LONG smoother memory on the tile grid (round 5): with
 * gfx_dynamics_ws_bytes_ex(param_rows, R, L) bytes -- the pole table, 64 bytes of counters and 8 bytes per row and
 * 512-sample tile -- a row whose history is longer than a tile may re-read (256 < H <= 64 * 512 samples, truncation term
 * dead) gets the state entering each tile from the AGGREGATES of the tiles before it: every tile publishes the state it
 * would leave from a zero entry state as soon as its own samples are scanned, then reads the ceil(H / 512) aggregates
 * before it (an in-launch hand-off through 8-byte granules, zeroed by a memset on the stream in front of the launch;
 * workgroups take their logical index from a ticket counter, so no tile ever waits for a workgroup that has not
 * started).  Only rows with a LIVE truncation term (a^iir_len > 1e-12) are left to the row kernel.  A workspace of
 * gfx_dynamics_ws_bytes(param_rows) bytes keeps the round-4 behaviour (those rows on the row kernel too). */
size_t gfx_dynamics_ws_bytes_ex(int64_t param_rows, int64_t R, int64_t L);
int gfx_dynamics_fused_ws_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap,
                              const float* log_threshold, const float* log_ratio, const float* log_knee,
                              const float* z_alpha, int64_t param_rows, int64_t R, int64_t C, int64_t L,
                              int smoother, int64_t iir_len, int knee, int gate, float* u1,
                              void* ws, size_t ws_bytes, void* stream);
/* Diagnostic, the twin of gfx_fftconv_last_kernel: the name -- as rocprofv3's kernel trace prints it -- of the kernel that
 * carries the rows of the calling thread's last successful gfx_dynamics_fused_* call: "dyn_oneshot_mix_kernel" (tiles with
 * the routing sums), "dyn_oneshot_kernel" (tiles; rows the pole table rejects ride on dyn_fused_kernel in the same call),
 * "dyn_oneshot_kernel+dyn_oneshot_mix_kernel" (the same with the look-back workspace: rows with a long smoother memory
 * are produced by the row-group walk, an instance of the routing-sum kernel without sums -- which of the two ran is
 * decided on the device) or "dyn_fused_kernel" (one workgroup per row: no workspace, or no smoother); "" before the first call. */
const char* gfx_dynamics_last_kernel(void);
/* The same with the routing sum that follows fused in (render/core.py:36-112, a "mix" stage that sums this call's rows):
 * rows come in graphs of `inner` consecutive rows (r = g * inner + j, R % inner == 0); destination d of graph g is
 * written to mix + g * mix_sb + d * mix_sv + c * mix_sc, every row's output to y as before.  Per destination the rows
 * are added in increasing j, from 0.0f -- the order of the gather-sum kernels, so the sums are identical to that separate
 * pass.  A destination holds one of n_acc <= 4 accumulators between its first and its last source; sched (device memory,
 * `inner` entries) says per row j: bits 0..3 = accumulators the row is added to; byte 1 + a = destination + 1 to store
 * accumulator a to (and clear it) after this row, 0 = none (the caller colours the live ranges:
 * grafx_amd.ops.mix_schedule).  `extras` (nullable; n_pre + n_post pairs of int64) names sources of the sum that are not
 * rows of this call but finished rows of the same buffer: (row offset from `mix` in units of mix_sv, code as in sched);
 * the first n_pre are added before the call's rows, the others after them.  Needs the one-pole smoother, the workspace and
 * 16-byte aligned rows with L % 4 == 0; GFX_EINVAL otherwise (callers run the two stages separately then). */
int gfx_dynamics_fused_mix_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                               const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                               int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                               float* u1, void* ws, size_t ws_bytes, const int64_t* sched, int64_t inner, int64_t n_acc,
                               float* mix, int64_t mix_sb, int64_t mix_sv, int64_t mix_sc, const int64_t* extras,
                               int64_t n_pre, int64_t n_post, void* stream);
/* The same with flags.  GFX_MIX_SKIP_ROWS: the call's own rows are NOT stored by the tile kernels -- for a render whose
 * caller wants the output node only (grafx_amd.render.render_grafx(keep_signal_buffer=False)) and a stage whose rows
 * nothing but the fused routing sums reads: the stage then moves 8 instead of 16 bytes per stereo sample plus the sums.
 * (Rows the pole table leaves to the row kernel are still written: the summing tiles read them back from y.) */
#define GFX_MIX_SKIP_ROWS 1
int gfx_dynamics_fused_mix_flags_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                                     const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                                     int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                                     float* u1, void* ws, size_t ws_bytes, const int64_t* sched, int64_t inner, int64_t n_acc,
                                     float* mix, int64_t mix_sb, int64_t mix_sv, int64_t mix_sc, const int64_t* extras,
                                     int64_t n_pre, int64_t n_post, int flags, void* stream);
int gfx_energy_f32(const float* x, gfx_rowmap_t xmap, float* e, int64_t R, int64_t C, int64_t L, void* stream);
int gfx_onepole_f32(const float* u, const float* z_alpha, float* out, int64_t R, int64_t L, int64_t Lout,
                    int64_t iir_len, int relu, void* stream);
/* gfx_onepole_f32 on the energy mean_c x^2 of a signal read in place through its row map (dynamics.py:390 + core/envelope.py:
 * 34-60 in one pass: no energy buffer), optionally leaving the bits of max |out| of every row in `rowmax` (R words; NULL: not
 * wanted) -- the by-product the odd-length aliasing of a full-length result (Lout = L + iir_len - 1) scales its pairs by. */
int gfx_onepole_energy_f32(const float* x, gfx_rowmap_t xmap, int64_t C, const float* z_alpha, float* out, int64_t R,
                           int64_t L, int64_t Lout, int64_t iir_len, int relu, uint32_t* rowmax, void* stream);
int gfx_onepole_fir_f32(const float* z_alpha, float* h, int64_t R, int64_t iir_len, void* stream);
/* Ballistics.forward (core/envelope.py:84-101): at, rt = sigmoid(z_alpha[:, 0]), sigmoid(z_alpha[:, 1]);  y[-1] = 1;
 * c = at if u[n] < y[n-1] else rt;  y[n] = (1 - c) y[n-1] + c u[n], the two products and the sum rounded separately
 * (torchcomp's CPU loop).  u, y: (R, L).  This entry walks every row whole, one lane per row. */
int gfx_ballistics_f32(const float* u, const float* z_alpha, float* y, int64_t R, int64_t L, void* stream);
/* The same values, bit for bit, produced from chunks of the rows: with a workspace of gfx_ballistics_ws_bytes(R) bytes
 * (scratch for this call: one flag per row) a row is cut into up to 64 chunks that start from a warmed-up guess and are
 * accepted only if every chunk is entered with exactly the state its left neighbour ends with; rows that fail that
 * check, or whose slower coefficient needs a longer warm-up than a chunk, are walked whole by a second launch in the same
 * call (no host synchronisation).  ws == NULL: gfx_ballistics_f32.  `is_coef` != 0: z_alpha holds at, rt themselves (no
 * sigmoid) -- how the parity tests hand the oracle's coefficients over. */
size_t gfx_ballistics_ws_bytes(int64_t R);
int gfx_ballistics_ws_f32(const float* u, const float* z_alpha, int is_coef, float* y, int64_t R, int64_t L, void* ws,
                          size_t ws_bytes, void* stream);
/* The same recursion over the energy of a signal, env = ballistics(mean_c x^2) (dynamics.py:390 followed by
 * core/envelope.py:84-101 -- Compressor / NoiseGate with energy_smoother="ballistics"): x (R, C, L) addressed through xmap
 * is read once, the energy never reaches memory.  env: (R, L). */
int gfx_ballistics_energy_f32(const float* x, gfx_rowmap_t xmap, int64_t C, const float* z_alpha, int is_coef, float* env,
                              int64_t R, int64_t L, void* ws, size_t ws_bytes, void* stream);
/* Compressor / NoiseGate with energy_smoother="ballistics" and no gain smoother as ONE pass over the signal
 * (dynamics.py:390-405 with core/envelope.py:84-101 inside): energy -> attack / release recursion (the float32 sequential
 * recursion exactly, as above) -> log -> knee -> exp -> y = gain * x.  x, y: (R, C, L) through their row maps (in-place
 * slices of the render buffer); per-row parameters as gfx_dynamics_fused_ex_f32 (row r reads row r % param_rows),
 * z_alpha: (param_rows, 2); ws: gfx_ballistics_ws_bytes(R) bytes, or NULL for the whole-row walk only. */
int gfx_dynamics_ballistics_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                                const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                                int64_t R, int64_t C, int64_t L, int knee, int gate, void* ws, size_t ws_bytes, void* stream);
/* Adjoint of the recursion above given the forward input x, output y and g = dL/dy:
 * gx = dL/dx (R, L), gz = dL/dz_alpha (R, 2).  The attack/release choice is treated as locally constant. */
int gfx_ballistics_bwd_f32(const float* x, const float* y, const float* g, const float* z_alpha, float* gx, float* gz,
                           int64_t R, int64_t L, void* stream);
/* The same with the rows cut into chunks that different workgroups walk (the adjoint is linear and a contraction: each
 * chunk starts 2048 samples later in time with a zero carry, exact to (1 - c)^2048 <= 6e-10 for coefficients >= 0.0103; a
 * 64-row group with a slower row is walked whole).  ws: gfx_ballistics_bwd_ws_bytes(R, L) bytes (per-chunk partial sums of
 * the two coefficient gradients, added in a fixed order). */
size_t gfx_ballistics_bwd_ws_bytes(int64_t R, int64_t L);
int gfx_ballistics_bwd_ws_f32(const float* x, const float* y, const float* g, const float* z_alpha, float* gx, float* gz,
                              int64_t R, int64_t L, void* ws, size_t ws_bytes, void* stream);
int gfx_dyn_gain_f32(const float* env, float* gain, const float* log_threshold, const float* log_ratio,
                     const float* log_knee, int64_t R, int64_t L, int knee, int gate, int log_out, void* stream);
/* Backward of the gain computer, for the training path (forward: gfx_dynamics_fused_f32).
 * gfx_dyn_gain_bwd_f32, one pass over x, the output gradient gy and the (smoothed) energy env (R, L):
 *   gain = exp(g(log(env + 1e-5)))                       -> gain (R, L)
 *   dg   = gain * sum_c gy[c] x[c];   denv = dg * dg/dG / (env + 1e-5)   -> denv (R, L)
 *   gparams[r, 0..2] += sum_n dg * (dg/dlog_threshold, dg/dlog_ratio, dg/dlog_knee)   (caller zero-fills)
 * with the partials of dynamics.py:444-489 / 676-721 (region masks piecewise constant, as in torch autograd).
 * gfx_dyn_dx_f32: gx = gain * gy + (2/C) * de * x, de = dL/d(mean_c x^2) (denv pushed back through the smoother). */
int gfx_dyn_gain_bwd_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap, const float* env,
                         const float* log_threshold, const float* log_ratio, const float* log_knee,
                         int64_t R, int64_t C, int64_t L, int knee, int gate,
                         float* gain, float* denv, float* gparams, void* stream);
int gfx_dyn_dx_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap, const float* gain,
                   const float* de, float* gx, int64_t R, int64_t C, int64_t L, void* stream);
/* The whole backward of Compressor / NoiseGate with the one-pole energy smoother and no gain smoother, in two
 * passes over the row (forward in time: recompute energy -> smoother -> gain, emit gain, the relu-masked d/d(smoothed
 * energy), the un-truncated scan u1 and the per-row parameter gradients; backward in time: the smoother's adjoint scan
 * and gx = gain * gy + (2/C) * de * x).  gx rows addressed by gxmap (e.g. a slice of the render's gradient buffer);
 * gparams (R,3); denv, u1 (R,L) each (workspace; the second pass recomputes the gain from u1 instead of reading a
 * stored copy).  dalpha (R), optional: dL/d(pole) of the smoother before
 * the sigmoid/clamp chain rule, accumulated by the second pass from u1, denv and its own adjoint scan (the D = dU/da
 * scan of core/envelope.py's truncated filter is moved onto the adjoint side, so no third scan is needed). */
int gfx_dynamics_bwd_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                         const float* log_threshold, const float* log_ratio, const float* log_knee,
                         const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                         float* gx, gfx_rowmap_t gxmap, float* gparams, float* denv, float* u1, float* dalpha,
                         void* stream);
/* The same backward when the forward pass has kept the scan: gfx_dynamics_fused_u1_f32 is gfx_dynamics_fused_ex_f32 that
 * also stores u1 (R, L) = (1-a) x the un-truncated one-pole scan of the energy (whole rows, one extra 4-byte store per
 * sample), and gfx_dynamics_bwd_u1_f32 is the second pass of gfx_dynamics_bwd_f32 alone, reading that u1.  Since round 2
 * gfx_dynamics_bwd_f32 itself runs as "scan x into u1" + that pass (the gain computer's derivatives are recomputed where
 * they are needed instead of being written out and read back); its `denv` workspace is no longer used and may be NULL. */
int gfx_dynamics_fused_u1_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap,
                              const float* log_threshold, const float* log_ratio, const float* log_knee,
                              const float* z_alpha, int64_t param_rows, int64_t R, int64_t C, int64_t L, int smoother,
                              int64_t iir_len, int knee, int gate, float* u1, void* stream);
int gfx_dynamics_bwd_u1_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                            const float* log_threshold, const float* log_ratio, const float* log_knee,
                            const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                            float* gx, gfx_rowmap_t gxmap, float* gparams, const float* u1, float* dalpha, void* stream);
/* The same with a workspace of gfx_dynamics_bwd_ws_bytes(R, L) bytes (scratch for this call; NULL = the call above): rows
 * with a short smoother memory -- chosen per row on the device, exactly as in gfx_dynamics_fused_ws_f32 -- run as
 * dependency-free one-shot tiles walking backward in time, the others on the row kernel.  The tiles' shares of the
 * per-row sums (gparams, dalpha) go through the workspace and are added in a fixed order: no float atomics, the gradients
 * are the same bits from run to run. */
size_t gfx_dynamics_bwd_ws_bytes(int64_t R, int64_t L);
int gfx_dynamics_bwd_u1_ws_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                               const float* log_threshold, const float* log_ratio, const float* log_knee,
                               const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                               float* gx, gfx_rowmap_t gxmap, float* gparams, const float* u1, float* dalpha,
                               void* ws, size_t ws_bytes, void* stream);
/* The same WITHOUT a scan kept by the forward pass (round 6): `u1_scratch` is R x L floats of scratch.  The one-shot tiles
 * rebuild the scan from x inside the tile (in the backward walk it is a suffix scan; the state entering a tile from its
 * far end is the dot product of the H samples beyond it), so for rows with a short smoother memory the scratch is never
 * touched and the forward pass of a training step has nothing to store; the rows of the row kernel get their scan written
 * to the scratch first.  Reference: autograd through dynamics.py:390-405 and core/envelope.py:34-60. */
int gfx_dynamics_bwd_rescan_ws_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                                   const float* log_threshold, const float* log_ratio, const float* log_knee,
                                   const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                                   float* gx, gfx_rowmap_t gxmap, float* gparams, float* u1_scratch, float* dalpha,
                                   void* ws, size_t ws_bytes, void* stream);
/* Pole gradient of TruncatedOnePoleIIRFilter (core/envelope.py:34-60) from the un-truncated scan U of its input and
 * the scan S of U:  da[r] = sum_n g[r,n] (c0 U[n] + c2 U[n-N]) + g[r,n+1] (c1 S[n] + c3 S[n-N]),  coef = (R, 4). */
int gfx_onepole_dz_f32(const float* g, const float* U, const float* D, const float* coef, float* da, int64_t R,
                       int64_t L, int64_t N, void* stream);
int gfx_apply_gain_f32(const float* x, gfx_rowmap_t xmap, const float* g, float* y, gfx_rowmap_t ymap,
                       int64_t R, int64_t C, int64_t L, int exp_gain, void* stream);
/* Gain computer and gain stage in one pass over an envelope a smoother left in memory (dynamics.py:394-405):
 * y[r,c,n] = exp(g(log(env[r,n] + 1e-5))) * x[r,c,n], g the knee of gfx_dyn_gain_f32; row r reads parameter row r % param_rows. */
int gfx_dyn_gain_apply_f32(const float* x, gfx_rowmap_t xmap, const float* env, float* y, gfx_rowmap_t ymap,
                           const float* log_threshold, const float* log_ratio, const float* log_knee, int64_t param_rows,
                           int64_t R, int64_t C, int64_t L, int knee, int gate, void* stream);
int gfx_stereo_gain_f32(const float* x, gfx_rowmap_t xmap, const float* log_gain, float* y, gfx_rowmap_t ymap,
                        int64_t R, int64_t C_in, int64_t L, void* stream);
/* StereoGain with the routing sum behind it fused in (the gain / pan stage in front of a bus): y as above, and the mix
 * destinations as gfx_dynamics_fused_mix_f32 produces them -- same sched / n_acc / extras words, same summation order,
 * identical sums.  log_gain is (R, 2); rows in graphs of `inner` (the row maps' `inner`); needs 16-byte aligned rows and
 * L % 4 == 0, GFX_EINVAL otherwise (callers run gfx_stereo_gain_f32 and the gather-sum then). */
int gfx_stereo_gain_mix_f32(const float* x, gfx_rowmap_t xmap, const float* log_gain, float* y, gfx_rowmap_t ymap,
                            int64_t R, int64_t C_in, int64_t L, const int64_t* sched, int64_t inner, int64_t n_acc,
                            float* mix, int64_t mix_sb, int64_t mix_sv, int64_t mix_sc, const int64_t* extras,
                            int64_t n_pre, int64_t n_post, void* stream);

/* Forward STFT of real rows: what torch.stft(x, n_fft, hop, window, center=True, pad_mode="reflect", return_complex=True)
 * returns, out = (rows, n_fft / 2 + 1, 1 + T / hop) complex64 (interleaved re, im).  Replaces the torch.stft call of
 * STFTMaskedNoiseReverb.sample_noise (reverb.py:116-128, fixed_noise=False).  n_fft even, <= 2048; rows <= 65535. */
int gfx_stft_f32(const float* x, const float* window, float* out, int64_t rows, int64_t T, int64_t n_fft, int64_t hop,
                 void* stream);
/* ---- STFT-masked noise reverb: impulse response ------------------------------------------
 * replaces STFTMaskedNoiseReverb.compute_stft_mask + compute_ir (reverb.py:161-200: mask, torch.istft),
 * ms_to_lr (core/midside.py:4-8) and the energy of normalize_impulse (core/utils.py:14-18).
 * noise_stft: (2, n_fft/2+1, num_frames) complex64 as interleaved floats (the module's buffer);
 * init/delta: (R, 2, n_fft/2+1); gain_env (nullable): (R, 2, num_frames); window: (n_fft).
 * Outputs ir (R, 2, ir_len), un-normalised, and row_gain (R) = 1/sqrt(mean_c sum_t ir^2 + 1e-12),
 * to be passed as `gain` (gain_div = 2) to gfx_fir_spectrum_f32.
 * `basis` is a per-(n_fft, window) constant from gfx_istft_basis_f32 (gfx_istft_basis_bytes: the windowed (kpad, n_fft)
 * matrix, the un-windowed half basis (n_fft/2+1, 2, roundup(n_fft/2+1, 16)) that the n_fft <= 384 matrix kernel uses, and
 * the n_fft complex factors e^(2 pi i k / n_fft), e^(2 pi i p / (n_fft/2)) of the FFT form, GFX_ISTFT_FFT below).
 */
size_t gfx_istft_basis_bytes(int64_t n_fft);
int gfx_istft_basis_f32(const float* window, float* basis, int64_t n_fft, void* stream);
size_t gfx_stft_reverb_workspace_bytes(int64_t R, int64_t n_fft, int64_t num_frames);
int gfx_stft_reverb_ir_f32(const float* noise_stft, const float* init_log_magnitude,
                           const float* delta_log_magnitude, const float* gain_env_log_magnitude,
                           const float* window, const float* basis, float* ir, float* row_gain,
                           int64_t R, int64_t ir_len, int64_t n_fft, int64_t hop, int64_t num_frames,
                           int ms_to_lr, void* ws, size_t ws_bytes, void* stream);
/* The same with one noise spectrum per row (`noise_rows` = R, noise_stft (R, 2, n_fft/2+1, num_frames) complex64) or one
 * shared by all rows (`noise_rows` = 1): STFTMaskedNoiseReverb(fixed_noise=False) draws fresh noise for every row
 * (reverb.py:63, 80-82, 165). */
int gfx_stft_reverb_ir_ex_f32(const float* noise_stft, int64_t noise_rows, const float* init_log_magnitude,
                              const float* delta_log_magnitude, const float* gain_env_log_magnitude, const float* window,
                              const float* basis, float* ir, float* row_gain, int64_t R, int64_t ir_len, int64_t n_fft,
                              int64_t hop, int64_t num_frames, int ms_to_lr, void* ws, size_t ws_bytes, void* stream);
/* The same with an explicit choice of how the frames are transformed (same results to rounding):
 *   GFX_ISTFT_GEMM  the inverse real DFT of a frame as a matrix product on the fp32 matrix cores (any even n_fft), the frames
 *                   written to the workspace and overlap-added by a second kernel.
 *   GFX_ISTFT_FFT   n_fft = 384 with hop = 192 (the reference's defaults, reverb.py:48-49) only, else GFX_EINVAL: a 192-point
 *                   complex FFT per frame (eight lanes a frame, 8 x 24 points), windowed and overlap-added in LDS; one
 *                   workgroup finishes 31 blocks of 192 samples of both channels of a row -- ~70x less arithmetic, and the
 *                   frames never reach memory.
 *   GFX_ISTFT_AUTO  what gfx_stft_reverb_ir_f32 / _ex_f32 use: FFT where it applies. */
#define GFX_ISTFT_AUTO 0
#define GFX_ISTFT_GEMM 1
#define GFX_ISTFT_FFT 2
size_t gfx_stft_reverb_workspace_bytes_sched(int64_t R, int64_t ir_len, int64_t n_fft, int64_t hop, int64_t num_frames,
                                             int schedule);   /* what _sched_f32 needs: the FFT form keeps no frames */
int gfx_stft_reverb_ir_sched_f32(const float* noise_stft, int64_t noise_rows, const float* init_log_magnitude,
                                 const float* delta_log_magnitude, const float* gain_env_log_magnitude,
                                 const float* window, const float* basis, float* ir, float* row_gain, int64_t R,
                                 int64_t ir_len, int64_t n_fft, int64_t hop, int64_t num_frames, int ms_to_lr, void* ws,
                                 size_t ws_bytes, int schedule, void* stream);

/* ---- routing ----------------------------------------------------------------------------
 * replaces read_single_tensor("index") + aggregate_tensor("sum"/"scatter") + inplace_write_tensor
 * (reference src/grafx/render/core.py:36-50, 101-112, 80-98) for the (B, V, C, L) signal buffer:
 *   out[b, j, c, n] = sum_{e in [seg_ptr[j], seg_ptr[j+1])} buf[b, src_idx[e], c, n]
 * buf/out are addressed by (batch, node, channel) strides in floats; src_idx (E) and
 * seg_ptr (J+1) are int64 device arrays (edges sorted by destination).
 */
int gfx_gather_sum_f32(const float* buf, int64_t buf_sb, int64_t buf_sv, int64_t buf_sc,
                       const int64_t* src_idx, const int64_t* seg_ptr,
                       float* out, int64_t out_sb, int64_t out_sv, int64_t out_sc,
                       int64_t B, int64_t J, int64_t C, int64_t L, void* stream);
/* Same result for fan-out routing (one source feeding several of the J <= 32 destinations; more than 8 is the shape of
 * a routing sum's adjoint: a few bus gradients onto every strip): every
 * distinct source row unique_src[u] is read once and added to the destinations whose bit is set in
 * dest_mask[u].  Needs 16-byte aligned rows (returns GFX_EINVAL otherwise: use gfx_gather_sum_f32). */
int gfx_gather_sum_fanout_f32(const float* buf, int64_t buf_sb, int64_t buf_sv, int64_t buf_sc,
                              const int64_t* unique_src, const int64_t* dest_mask, int64_t U,
                              float* out, int64_t out_sb, int64_t out_sv, int64_t out_sc,
                              int64_t B, int64_t J, int64_t C, int64_t L, void* stream);

/* ---- exact recursive biquad cascade -------------------------------------------------------
 * replaces IIRFilter._process_lfilter / _process_ssm: core/iir.py:154-261 (torchaudio.functional.lfilter per
 * section / the state-space form on torchlpc).  K second-order sections in series per row-channel, zero initial
 * state, coefficients normalised by a0:  y = b0 w[n] + b1 w[n-1] + b2 w[n-2],  w[n] = x[n] - a1 w[n-1] - a2 w[n-2].
 * A parallel scan over time (matrix powers of the 2x2 transition matrix), no FFT.
 * Bs, As: (R, C_f, K, 3) contiguous; channels broadcast 1<->C like gfx_fftconv_f32; K <= 32.
 * ssm_quirk != 0 reproduces upstream's "ssm" backend for K > 1, which feeds the ORIGINAL input to the
 * recursive part of every section (core/iir.py:226-246); for K == 1 both settings are the same filter. */
int gfx_biquad_cascade_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* Bs,
                           const float* As, int64_t R, int64_t C_in, int64_t C_f, int64_t K, int64_t L, int ssm_quirk,
                           void* stream);

/* ---- noise-shaping reverb impulse response ------------------------------------------------
 * replaces the envelope synthesis of FilteredNoiseShapingReverb.forward (reverb.py:343-366):
 *   ir[r,c,t] = sum_k noise[c,k,t] * log_gain[r,c,k] * (exp(t*d) - sigmoid(z_fade_in_gain)*exp(t*f))
 *   d = sigmoid(log_decay)*(max_decay-min_decay)+min_decay,  f = sigmoid(log_fade_in)*(d-min_decay)+min_decay
 * (the fade-in term only when both fade pointers are non-null; the gain is the raw parameter, as upstream).
 * noise: (C, K, noise_stride) band-split noise, the first ir_len samples of every band row are used (pass an
 * offset pointer for the "pseudo-random" start).  Parameters (R, C, K) contiguous; ir (R, C, ir_len).  K <= 64. */
int gfx_noise_shaping_ir_f32(const float* noise, int64_t noise_stride, const float* log_decay, const float* log_gain,
                             const float* log_fade_in, const float* z_fade_in_gain, float* ir, int64_t R, int64_t C,
                             int64_t K, int64_t ir_len, float min_decay, float max_decay, void* stream);

/* ---- memoryless waveshapers --------------------------------------------------------------
 * replaces the forward() of TanhDistortion (nonlinear.py:46-79), PiecewiseTanhDistortion (120-175),
 * PowerDistortion (210-233) and ChebyshevDistortion (270-307): one streaming pass.
 *   u = pre * (x - dc),  pre = exp(log_pre_gain[r]) (1 if null),  dc = dc[r*C + c] (0 if null)
 *   GFX_WS_TANH       y = tanh(u + b) - tanh(b),            b = p0[r] (0 if null)
 *   GFX_WS_PIECEWISE  p0 = log_hardness (R,2), p1 = z_threshold (R,2)  (upstream's split order is kept)
 *   GFX_WS_POWER      y = sum_k tanh(p0[r,k]) f(u^k),       k < K <= 32, f = tanh if use_tanh
 *   GFX_WS_CHEBYSHEV  y = sum_k tanh(p0[r,k]) f(T_k(u))
 * then y *= post, post = 1/pre if inverse_post_gain else exp(log_post_gain[r]) (1 if null).
 * gfx_row_mean_f32 computes the per-row-channel means the remove_dc option subtracts. */
#define GFX_WS_TANH 0
#define GFX_WS_PIECEWISE 1
#define GFX_WS_POWER 2
#define GFX_WS_CHEBYSHEV 3
int gfx_row_mean_f32(const float* x, gfx_rowmap_t xmap, float* mean, int64_t R, int64_t C, int64_t L, void* stream);
int gfx_waveshaper_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, int64_t R, int64_t C, int64_t L,
                       int mode, int use_tanh, int inverse_post_gain, const float* log_pre_gain,
                       const float* log_post_gain, const float* p0, const float* p1, int64_t K, const float* dc,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAFX_AMD_H */
