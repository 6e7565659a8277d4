/* maggie_hip.h -- C ABI of libmaggie_hip.so: the MI355X (gfx950) kernels behind MaGGIe's matting hot path.
 *
 * The reference (hmchuong/MaGGIe) has no FFI layer of its own: its hot path is Python `nn.Module`s calling
 * third-party native code (cuDNN through torch, spconv, OpenCV). This header is the boundary a maintainer binds
 * instead of those calls; `maggie_amd/hip.py` is that binding (ctypes) and `maggie_amd/network/` mirrors the
 * reference's module API on top of it. Every entry point cites the reference call it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), 16-byte aligned, unless the name says host;
 *   - activations are NHWC ("rows x channels": row = (n*H + h)*W + w) in `dtype` = MG_F32 or MG_BF16;
 *     sparse feature matrices are rows x channels with row = active-site id (sorted (batch,y,x) order);
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); kernels are asynchronous;
 *   - return value 0 = launched, <0 = argument error, >0 = hipError_t.
 */
#ifndef MAGGIE_HIP_H
#define MAGGIE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MG_F32 0
#define MG_BF16 1

#define MG_ACT_NONE 0
#define MG_ACT_RELU 1
#define MG_ACT_LRELU 2

#define MG_MODE_CONV 0   /* src(m,tap) = (n, ho*stride - pad + ky*dil, wo*stride - pad + kx*dil)                 */
#define MG_MODE_TCONV 1  /* src(m,tap) = (n, (ho + pad - ky*dil)/stride, ...) when divisible: dgrad / ConvTranspose */
#define MG_MODE_GATHER 2 /* src(m,tap) = nbr[m*taps + tap]  (-1 = no neighbour): sparse convolutions               */

int mg_abi_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution (MFMA):  Y[m, yoff+co] = epi( sum_{tap,ci} X[src(m,tap), ci] * W[co, tap, ci] )
 *   epi(v): if pre_act v = act(v);  v = v*scale[co] + shift[co];  v += res;  if !pre_act v = act(v);  v += res2
 *   stats (optional, fp32 [2*Cout], pre-zeroed): per-channel sum and sum of squares of the stored Y (BatchNorm).
 * Replaces: nn.Conv2d / nn.ConvTranspose2d (+ folded BatchNorm, activation, residual add) in
 *   maggie/network/encoder/resnet.py:23-39,177-200; maggie/network/decoder/resnet.py:28-45;
 *   maggie/network/module/aspp.py:34-56; maggie/network/module/instance_matte_decoder.py:81-88,290;
 *   maggie/network/module/conv_gru.py:24-29; and spconv SubMConv2d / SparseInverseConv2d
 *   (maggie/network/decoder/resnet_inst_matt_spconv.py:69-130) in MG_MODE_GATHER.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mg_conv_params {
    const void* x;       /* [rows_in, ldx] activations                                            */
    const void* w;       /* [Cout, R*S, Cin] weights, same dtype, Cin padded like x               */
    void* y;             /* [M, ldy] output                                                       */
    const int32_t* nbr;  /* MG_MODE_GATHER: [M, R*S] source rows, -1 = none                       */
    const float* scale;  /* [Cout] or NULL                                                        */
    const float* shift;  /* [Cout] or NULL (bias / folded BN shift)                               */
    const void* res;     /* residual, [M or M/4, ldr], same dtype, or NULL                        */
    const void* res2;    /* post-activation residual, [M, ldr2] or NULL                           */
    float* stats;        /* [2*Cout] fp32 or NULL                                                 */
    int32_t dtype, mode;
    int32_t N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad, dil;
    int32_t M;           /* output rows (N*Hout*Wout, or active sites)                            */
    int32_t ldx, ldy, yoff, ldr, ldr2;
    int32_t act, pre_act, res_mode; /* res_mode: 1 = same rows, 2 = residual at half resolution (nearest x2) */
    float slope;
} mg_conv_params;

int mg_conv_fprop(const mg_conv_params* p, void* stream);

/* Weight gradient:  dW[co, tap, ci] (+)= sum_m dY[m, co] * X[src(m,tap), ci]   (fp32 accumulate, atomics across
 * row splits; dW must be pre-zeroed fp32 [Cout, R*S, Cin]). Same geometry struct; `y` = dY (read), `res` unused,
 * `w` unused, `stats` = dW. Replaces cuDNN wgrad / spconv's indice_conv_backward filter gradient. */
int mg_conv_wgrad(const mg_conv_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGGIE_HIP_H */
