/* mtn.h -- C ABI of the MI355X-native MultiTaskNet hot path (libmtn_hip.so).
 *
 * The reference (yingkunwu/hand-gesture-recognition) has no FFI layer: its boundary is the
 * Python nn.Module `MultiTaskNet` (model/multitasknet.py:8-29) whose forward issues PyTorch
 * ATen ops.  Each entry point below replaces the ATen op sequence of the cited reference lines;
 * the Python host side (hand-gesture-recognition_amd/) binds them with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero otherwise (mtn_last_error() has the text);
 *     no C++ exception crosses the boundary, nothing calls exit().
 *   - enqueue-only on the caller's HIP stream (`stream` is a hipStream_t passed as void*);
 *     no allocation, no synchronisation, no retained pointers -> safe under hipGraph capture.
 *   - the caller (PyTorch) owns every buffer incl. workspaces (`*_workspace_bytes` sizes them).
 *   - dtype: MTN_DTYPE_F32 = fp32 storage + exact-f32 MFMA (validation mode),
 *            MTN_DTYPE_BF16 = bf16 storage + bf16 MFMA with fp32 accumulation (throughput mode).
 *     Statistics, gradients of parameters, optimizer state and losses are always fp32.
 *   - activations are NHWC ("channels last") / token-major [rows][channels]; `ld*` is the
 *     element stride between consecutive rows (pixels), so an op can read or write a channel
 *     slice of a wider buffer (this is how torch.cat / chunk of model/gelan.py:138,142 are free).
 *   - pointers and row strides must keep 16-byte alignment.
 */
#ifndef MTN_H_
#define MTN_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTN_DTYPE_F32 0
#define MTN_DTYPE_BF16 1
#define MTN_ACT_NONE 0
#define MTN_ACT_SILU 1
#define MTN_ACT_GELU 2

const char* mtn_last_error(void);
int mtn_abi_version(void);

/* ---- GELAN backbone: Conv2d(bias=False) -> BatchNorm2d(train) -> SiLU  (model/gelan.py:37-56) */

/* number of row tiles whose partial BN statistics mtn_conv_fwd writes: stats is [tiles][2][Cout] */
int mtn_conv_stats_tiles(int M, int Cout);

/* y_raw = conv(x) (k in {1,3}, pad k/2, stride in {1,2}); w packed [Cout][k][k][Cin];
 * stats (optional) receives per-tile per-channel sum / sum-of-squares of y_raw. */
int mtn_conv_fwd(const void* x, int ldx, const void* w, void* y, int ldy, float* stats,
                 int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                 void* stream);

/* dx = conv_transpose(dy) [+ res]; wt packed [Cin][k][k][Cout]; (H, W) is the INPUT size. */
int mtn_conv_dgrad(const void* dy, int lddy, const void* wt, void* dx, int lddx,
                   const void* res, int ldres, int dtype, int B, int H, int W, int Cin, int Cout,
                   int ksize, int stride, void* stream);

/* ---- nn.Linear / 1x1 projections  (model/transformer.py:34-37,56-58; multitasknet.py:13) ------
 * out = act(x @ w^T + bias) + res ; w is [N][K]; aux_out (optional) gets the pre-activation. */
int mtn_linear_fwd(const void* x, int ldx, const void* w, const float* bias, int act,
                   void* aux_out, int ld_aux, const void* res, int ldres, void* out, int ldo,
                   int dtype, int M, int N, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTN_H_ */
