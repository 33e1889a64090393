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
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises, nothing allocates (callers pass workspaces whose
 *     size the *_bytes queries return);
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

#ifdef __cplusplus
}
#endif
#endif /* GRAFX_AMD_H */
