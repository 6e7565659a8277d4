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
 *   - activations are NHWC ("rows x channels": row = (n*H + h)*W + w) in `dtype` = MG_F32, MG_BF16 or MG_F16 (every `dtype` argument below);
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
#define MG_F16 3 /* IEEE half storage, fp32 accumulate: the reference's `--precision 16` (torch.cuda.amp fp16 autocast + GradScaler, engine/train.py:208,227-229) */

#define MG_ACT_NONE 0
#define MG_ACT_RELU 1
#define MG_ACT_LRELU 2

#define MG_STAT_REPLICAS 32 /* BatchNorm statistic accumulators are [MG_STAT_REPLICAS][2*C]: atomics of different blocks spread over replicas */

#define MG_MODE_CONV 0   /* src(m,tap) = (n, ho*stride - pad + ky*dil, wo*stride - pad + kx*dil)                 */
#define MG_MODE_TCONV 1  /* src(m,tap) = (n, (ho + pad - ky*dil)/stride, ...) when divisible: dgrad / ConvTranspose */
#define MG_MODE_GATHER 2 /* src(m,tap) = nbr[m*taps + tap]  (-1 = no neighbour): sparse convolutions               */

int mg_abi_version(void);
/* Declare [base, base + bytes) zero-filled with every part handed to at most one accumulator (NULL, 0: no such range). Entry points that clear an
 * accumulator with a fill launch of their own skip it for buffers inside the range -- the host's per-graph zero arena (one memset node per replay). */
int mg_set_zeroed_range(void* base, long bytes);
/* The invariant is checked: an entry point asked to clear words of the range that an earlier call already claimed fails with hipErrorAlreadyMapped
 * (208) instead of silently keeping stale sums; mg_zeroed_range_conflicts() returns (and resets) how often that happened. */
int mg_zero_claim(void* p, long bytes);
int mg_zeroed_range_conflicts(void);

/* Bit-reproducible steps. The reference trains with torch.backends.cudnn.deterministic = True, benchmark = False (tools/main.py:135-136): two
 * runs of one step give the same bits, so the active-pixel index map -- a threshold of the coarse alpha (maggie/utils/utils.py:31) -- is the same
 * run to run. mg_set_deterministic(1) (the library's default) gives this library the same property: every cross-workgroup fp32 sum (BatchNorm
 * statistics and backward sums, bias / LayerNorm gradients, the token side of the attention backward, loss sums, SpectralNorm dot products, the
 * gradient norm) is formed as "one partial per workgroup, added in index order" instead of atomicAdd. mg_set_deterministic(0) restores the atomic
 * forms. The ordered sums stage their partials in a library-owned scratch that mg_det_init(bytes) allocates on the CURRENT device (call it once,
 * outside any stream capture; entry points return -7 when it is missing or too small). The scratch is shared by all launches: issue the library's
 * kernels on ONE stream at a time (stream order also holds inside a captured graph).
 * mg_stat_rows(): rows of the [rows][2C] statistics scratch the column-statistics kernels fill (MG_STAT_REPLICAS, or MG_DET_STAT_ROWS = 1024 in
 * deterministic mode: one row per row block, added in row order by mg_bn_finalize). */
#define MG_DET_STAT_ROWS 1024
int mg_set_deterministic(int on);
int mg_get_deterministic(void);
int mg_det_init(long bytes);
/* Round 5: ONE side stream per device may run this library's BatchNorm kernels concurrently with the main stream (the encoder's shortcut branches next to
 * the backbone, maggie/network/encoder/resnet.py:167-175): register it here (NULL: none) -- it gets a slot scratch of its own (`bytes`, allocated once,
 * never moved). Outside a stream capture. */
int mg_det_side_stream(void* stream, long bytes);
int mg_stat_rows(void);
/* test hook: dst[g][c] += sum_b slots[g][b][c] in the library's fixed order ([groups][nblk][nv] fp32) */
int mg_det_reduce_test(const float* slots, int nblk, int groups, int nv, float* dst, void* stream);

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
    float* stats;        /* [MG_STAT_REPLICAS][2*Cout] fp32 (pre-zeroed) or NULL                  */
    int32_t dtype, mode;
    int32_t N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad, dil;
    int32_t M;           /* output rows (N*Hout*Wout, or active sites)                            */
    int32_t ldx, ldy, yoff, ldr, ldr2;
    int32_t act, pre_act, res_mode; /* res_mode: 1 = same rows, 2 = residual at half resolution (nearest x2) */
    float slope;
    int32_t dw_dtype;    /* mg_conv_wgrad_ws only: dtype dW is written in (MG_F32 = 0 default; MG_BF16 / MG_F16 = the activations' 16-bit type, needs a workspace) */
    int32_t stat_mode;   /* mg_conv_fprop `stats`: 0 = [MG_STAT_REPLICAS][2*Cout] sum and sum of squares (default);
                            1 = ONE row [2*Cout], column sums only (small layers: feeds the exact two-pass variance) */
    const int32_t* m_dev; /* optional DEVICE row count (sparse head): the kernels run over min(*m_dev, M) rows, M is then the
                            capacity the buffers were sized for. The launch is a fixed, persistent grid (tiles are walked
                            grid-stride), so no host code ever needs the count: the detail stage stays free of device->host
                            reads and its launches can be captured into a hipGraph. NULL: M is the row count (dense layers). */
    /* mg_conv_fprop, optional (bnb_x != NULL): this launch computes the gradient dz arriving at the OUTPUT z = act(BN(x) [+ res]) of a
     * training BatchNorm layer (maggie/network/encoder/resnet.py:28-39 backward). The epilogue then writes g = dz * act'(z) to `y` and
     * accumulates that layer's backward reductions into `stats` (stat_mode 0 layout): stats[c] += sum g, stats[Cout + c] += sum g * xhat,
     * xhat = (x - mean) * invstd -- the separate mg_bn_bwd_reduce pass disappears. bnb_y / bnb_x: [M, bnb_ld] rows of z (NULL when the
     * layer has no activation) and x in the launch dtype; bnb_act: the layer's activation (its LeakyReLU slope is `slope`). */
    const void* bnb_y;
    const void* bnb_x;
    const float* bnb_mean;
    const float* bnb_invstd;
    int32_t bnb_act, bnb_ld;
    int32_t stat_rep;    /* rows of `stats` in stat_mode 0 (0 = MG_STAT_REPLICAS): output tile t adds to row t % stat_rep. With stat_rep >= the number of
                            output tiles (mg_conv_stat_rows) every word receives ONE addition and mg_bn_finalize adds the rows in index order: the
                            statistics are then bit-reproducible run to run (the reference runs with cudnn.deterministic = True, tools/main.py:135-136) */
    int32_t reserved0;
    /* Operand transform (round 5; conv + BatchNorm + activation as one pass in TRAINING, maggie/network/encoder/resnet.py:26-37,
     * maggie/network/decoder/resnet.py:20-45): xf_scale != NULL makes every IN-IMAGE element of `x` enter the contraction as
     *     x_eff[m, ci] = act(x[m, ci] * xf_scale[ci] + xf_shift[ci]),   act = xf_act with LeakyReLU slope xf_slope,
     * rounded to the storage type -- the value mg_affine_act would have written -- while padding (out-of-image taps) stays 0. `x` is then the RAW
     * output of the producing convolution and (xf_scale, xf_shift) the folded batch statistics of the BatchNorm layer between the two
     * convolutions (mg_bn_finalize): the normalised activation is never stored. Honoured by mg_conv_fprop (MG_MODE_CONV) and by
     * mg_conv_wgrad* (the `x` operand) for the kernel forms mg_conv_xform_ok reports; other forms return -9. */
    const float* xf_scale;
    const float* xf_shift;
    int32_t xf_act;
    float xf_slope;
    /* bnb_* launches whose BatchNorm layer never stored its activation output (operand-path BatchNorm: bnb_y == NULL although bnb_act != none): the
     * sign of the activation's argument is re-formed as bnb_x * bnb_scale + bnb_shift (the layer's folded scale | shift, as in xf_*). */
    const float* bnb_scale;
    const float* bnb_shift;
} mg_conv_params;

/* 1 when the kernel form this geometry dispatches to applies the operand transform (xf_*) in flight: which = 0 mg_conv_fprop[_ws], 1 mg_conv_wgrad*. */
int mg_conv_xform_ok(const mg_conv_params* p, int which);

int mg_conv_fprop(const mg_conv_params* p, void* stream);
/* Round 6: the 3x3 / stride-1 / pad-1 layers with Cin % 32 == 0 in the 16-bit types (forward and data gradient, with the operand transform, the
 * residual / statistics epilogue and the BatchNorm-link sums) are tried first on the producer / consumer halo-tile kernel of csrc/conv_halo3.hip
 * (reference: maggie/network/encoder/resnet.py:23-39,167-175, decoder/resnet.py:20-45). These two are A/B and experiment switches, not part of the
 * path's contract: mg_set_halo3(0) restores the round-2..5 kernel forms (returns the previous setting; environment MG_HALO3),
 * mg_set_halo3_cfg(TH, BN, NS) forces one tile form (tile rows 4 | 8, channels 32 | 64, ring depth 1 | 3 | 4, or 100 + depth = the persistent ring form that walks several tiles per workgroup -- measured, not selected by the
 * dispatch: DESIGN.md 12.1; 200 | 201 = never | always the one-slab persistent form for Cin == 32, Cout <= 32 layers, which the dispatch takes from 4 096 tiles up;
 * 0, 0, 0 = the measured dispatch). */
int mg_set_halo3(int on);
int mg_set_halo3_cfg(int th, int bn, int ns);
/* Rows of a `stats` buffer (stat_mode 0) that give every output tile of any forward kernel form its own row for an [M = N * Hout * Wout] output
 * (deterministic mode; MG_STAT_REPLICAS in atomic mode): what mg_conv_params.stat_rep must be at least, else the launch fails with -8. */
int mg_conv_stat_rows(int M, int N, int Hout, int Wout);
/* *out = the sticky error word of the cooperative single-launch kernels on the current device (mg_bn_train_bwd's one-launch form: a workgroup gave up
 * waiting for its peers -- the launch's result is then invalid); 0 = none. A host read (hipMemcpy): for tests and end-of-run checks. */
int mg_coop_error(int* out);
/* Switch mg_bn_train_bwd's one-launch form (reduce + ordered sum + apply behind a flag hand-shake; measured slower than the three launches, off by
 * default, MG_BN_COOP=1) on or off; returns the previous setting. */
int mg_set_bn_coop(int on);
/* Switch the last-arriver form of mg_bn_bwd_reduce / mg_bn_train_bwd (deterministic mode: the last row block of a channel group to finish adds the
 * partial rows in row order in its own tail, with the arithmetic of the separate ordered-sum launch -> same bits, one launch less per layer; measured
 * +30 us per layer, off by default, MG_BN_BWD_TAIL=1) on or off; returns the previous setting. */
int mg_set_bn_bwd_tail(int on);
/* Same, allowed to split the K dimension over several blocks per tile for deep layers with few output rows (M <= 8192,
 * K >= ~1152, Cout >= 64): mg_conv_fprop_workspace(p) = floats of scratch that plan needs (0: no split, identical to
 * mg_conv_fprop); partial tiles go to the workspace, a second kernel sums them and applies the epilogue (deterministic). */
long mg_conv_fprop_workspace(const mg_conv_params* p);
int mg_conv_fprop_ws(const mg_conv_params* p, float* workspace, long workspace_floats, void* stream);

/* Weight gradient:  dW[co, tap, ci] (+)= sum_m dY[m, co] * X[src(m,tap), ci]   (fp32 accumulate, atomics across
 * row splits; dW must be pre-zeroed fp32 [Cout, R*S, Cin]). Same geometry struct; `y` = dY (read), `res` unused,
 * `w` unused, `stats` = dW. Replaces cuDNN wgrad / spconv's indice_conv_backward filter gradient. */
int mg_conv_wgrad(const mg_conv_params* p, void* stream);
/* Deterministic, atomic-free variant: with `workspace` >= mg_conv_wgrad_workspace(p) floats every row split writes its own
 * partial and a second kernel reduces them; dW is then OVERWRITTEN (no pre-zeroing needed). */
long mg_conv_wgrad_workspace(const mg_conv_params* p);
int mg_conv_wgrad_ws(const mg_conv_params* p, float* workspace, long workspace_floats, void* stream);
/* Parked slab reduction: mg_conv_wgrad_park runs the weight-gradient GEMM of mg_conv_wgrad_ws WITHOUT its trailing "row-split slabs -> dW"
 * kernel and returns that reduction's descriptor; the caller keeps `workspace` alive and untouched and later hands all the descriptors of a
 * backward pass to mg_wgrad_reduce_batched: one launch (per 64 layers) instead of one per layer (~80 per training step), same arithmetic per
 * element, so dW has the same bits either way. splits == 0 in a returned descriptor: nothing was parked, dW is already complete. */
typedef struct mg_wgrad_parked {
    const float* ws;     /* the slabs: [splits][n] fp32 */
    void* dw;            /* destination, n elements of dw_dtype */
    long n;
    int32_t splits, form, dw_dtype, blocks;
} mg_wgrad_parked;
int mg_conv_wgrad_park(const mg_conv_params* p, float* workspace, long workspace_floats, mg_wgrad_parked* out, void* stream);
int mg_wgrad_reduce_batched(const mg_wgrad_parked* items, int count, void* stream);


/* ---------------------------------------------------------------------------------------------------------------
 * Row x channel normalisation / activation kernels (HBM-bound, 16-byte vector accesses).
 * Replace nn.BatchNorm2d / nn.BatchNorm1d (+ ReLU / LeakyReLU(0.2) / residual add) in
 *   maggie/network/encoder/resnet.py:23-39,167-175; maggie/network/decoder/resnet.py:28-45;
 *   maggie/network/module/aspp.py:34-56; maggie/network/decoder/resnet_inst_matt_spconv.py:69-130 (BatchNorm1d over
 *   the active rows), and nn.AvgPool2d(2,2) / nn.UpsamplingNearest2d(2) (encoder/resnet.py:113, decoder:143).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mg_rowwise_params {
    const void* x;        /* [M, ldx] pre-normalisation input (conv output)                       */
    void* y;              /* [M, ldy] (+yoff) output of the forward / saved output in backward     */
    const void* dy;       /* [M, lddy] upstream gradient (backward)                                */
    void* dx;             /* [M, lddx] gradient w.r.t. x (backward) or NULL                        */
    const void* res;      /* residual added before the activation, or NULL                         */
    const void* res2;     /* residual added after the activation, or NULL                          */
    void* dres;           /* [M, lddres] gradient of `res` (= dy * act'), or NULL                   */
    const float* scale;   /* [C] gamma*invstd (forward) / (backward)                               */
    const float* shift;   /* [C] beta - mean*gamma*invstd                                          */
    const float* mean;    /* [C] (backward)                                                        */
    const float* invstd;  /* [C] (backward)                                                        */
    float* sums;          /* [2C] backward reductions: sum g, sum g*xhat                            */
    const float* count_ptr; /* optional device scalar row count (SyncBN); else `count`               */
    int32_t dtype, M, C, H, W;
    int32_t ldx, ldy, yoff, ldr, ldr2, lddy, lddx, lddres;
    int32_t act, res_mode, mask_x_pos;
    float slope, count;
    const int32_t* m_dev; /* optional DEVICE row count (see mg_conv_params.m_dev): rows = min(*m_dev, M); BatchNorm then uses it as
                             the sample count (exact two-pass variance), NULL: M rows */
    int32_t count_mult;   /* mg_bn_train_fwd: every row stands for count_mult samples in the unbiased running-variance factor n / (n - 1)
                             (0 = 1). The decoder's skip branch (UpsamplingNearest2d(2) -> 1x1 conv -> BatchNorm, maggie/network/decoder/resnet.py:
                             143-147 of the reference) is evaluated BEFORE the up-sampling here: mean and biased variance are unchanged by the 2x2
                             replication, the reference's sample count is 4x the rows (count_mult = 4) */
    int32_t mask_from_x;  /* backward of an operand-path BatchNorm layer (round 5, mg_conv_params.xf_*): the activation output was never stored -- `y` is
                             ignored and the sign of act's argument is re-formed as x * scale + shift (needs `shift`) */
} mg_rowwise_params;

/* stats[rep][c] += sum_m x[m,c], stats[rep][C+c] += sum_m x[m,c]^2 (fp32 [MG_STAT_REPLICAS][2C], pre-zeroed) */
int mg_colstats(const void* x, int dtype, int M, int C, int ld, float* stats, void* stream);
/* batch statistics (`nrep` replicas of [2C], summed here) -> scale/shift/mean/invstd, running-stat update (momentum, unbiased var) */
/* exact two-pass variant for small M: stats[0:C] += sum, stats[C:2C] += sum (x - mean)^2 (stats [2C] must arrive zeroed) */
int mg_colstats_centered(const void* x, int dtype, int M, int C, int ld, float* stats, int have_sum, void* stream);
/* `centered` != 0: stats[C:2C] holds the centred second moment (mg_colstats_centered) instead of sum x^2 */
/* out[n] = sum over the rows of stats[nrep][n] in the library's fixed order (local statistics of a SyncBatchNorm layer before the exchange) */
int mg_stat_rows_sum(const float* stats, int nrep, int n, float* out, void* stream);
int mg_bn_finalize(const float* stats, int nrep, const float* count_ptr, float count, int C, int centered, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                   float* mean_out, float* invstd_out, void* stream);
/* Training BatchNorm(+residual+activation) behind one entry point each way (same kernels as the calls above; local statistics only):
 *   mg_bn_train_fwd: p: x, y, res*, act, slope, H, W, dtype, M, C, ld*; stats_ws: statistics scratch, [2C] floats with `exact`, else
 *                    [MG_STAT_REPLICAS][2C] (ws_zeroed != 0: the caller hands it over zeroed, e.g. a slice of a per-step zero arena;
 *                    else it is zeroed here); outs: 4C floats receiving scale | shift | mean | invstd (kept for the backward);
 *                    stats_in (or NULL): statistics already accumulated by the producing conv's epilogue -- stats_in_rows replicas
 *                    of [2C], or with `exact` one row of column sums; `exact`: two-pass variance (few rows per channel).
 *                    Updates running_mean / running_var (or NULL).
 *   mg_bn_train_bwd: p: dy, y, x, scale, mean, invstd, dx, dres, sums [2C] (zeroed here unless sums_zeroed; = dbeta | dgamma after). */
int mg_bn_train_fwd(const mg_rowwise_params* p, float* stats_ws, int ws_zeroed, float* outs, const float* stats_in, int stats_in_rows,
                    int exact, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                    void* stream);
int mg_bn_train_bwd(const mg_rowwise_params* p, int sums_zeroed, void* stream);
/* eval mode: fold running statistics into scale/shift */
int mg_bn_fold(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
               float eps, float* scale, float* shift, void* stream);
/* y = act(x*scale + shift + res) + res2 */
int mg_affine_act(const mg_rowwise_params* p, void* stream);
/* BatchNorm backward: reduce (sums) then apply (dx, dres) */
int mg_bn_bwd_reduce(const mg_rowwise_params* p, void* stream);
int mg_bn_bwd_apply(const mg_rowwise_params* p, void* stream);
/* apply pass for a layer whose reductions rode on the consumer conv's data-gradient epilogue (mg_conv_params.bnb_*): p->dy holds
 * g = dy * act'(y) already, sums_rep = nrep replicas of [2C] (summed here); sums_out [2C] receives dbeta | dgamma. Needs a power-of-two
 * number of 16-byte channel chunks per row (-3 otherwise). */
int mg_bn_bwd_apply_linked(const mg_rowwise_params* p, const float* sums_rep, int nrep, float* sums_out, void* stream);
/* op 0: 2x2 average pool, 1: 2x2 sum pool (out Ho x Wo from 2Ho x 2Wo); 2: 0.25*nearest-upsample, 3: nearest-upsample
 * (out Ho x Wo from Ho/2 x Wo/2) */
int mg_pool2x2(const void* in, void* out, int dtype, int op, int N, int Ho, int Wo, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Region ops on bit-packed planes (uint64 words, bit b of word j <-> pixel x = 64*j + b; rows padded to whole words).
 * Replace compute_unknown (maggie/utils/utils.py:27-55: threshold + CPU cv2.dilate with MORPH_ELLIPSE + H2D copy),
 * the rule-book side of spconv (`dummy_downscale`, maggie/network/decoder/resnet_inst_matt_spconv.py:61-66,217-218)
 * and torch.nonzero (:206). Integer work: results are bit-exact w.r.t. oracle/region.py.
 * ------------------------------------------------------------------------------------------------------------- */
#define MG_U8 2
/* mode 0: bit = (lo < a < hi); mode 1: bit = (a > 0). `a` is [P,H,W] planes of dtype MG_F32 or MG_U8. */
int mg_bits_pack(const void* a, int dtype, void* bits, int P, int H, int W, int mode, float lo, float hi, void* stream);
int mg_bits_unpack_u8(const void* bits, uint8_t* out, int P, int H, int W, void* stream);
int mg_bits_unpack_f32(const void* bits, float* out, int P, int H, int W, void* stream);   /* the same as 0.f / 1.f planes (loss weights) */
/* progressive refinement blend of `fuse` (resnet_inst_matt_spconv.py:272-290): out = bit ? a : b over fp32 planes [P][H][W] (= a*w + b*(1-w)
 * for the 0/1 weight plane the bits encode), and its backward da = bit ? dy : 0, db = bit ? 0 : dy (either may be NULL) */
int mg_bits_select(const void* bits, const float* a, const float* b, float* out, int P, int H, int W, void* stream);
int mg_bits_select_bwd(const void* bits, const float* dy, float* da, float* db, int P, int H, int W, void* stream);
/* out = dilate(in, ellipse(width)) [& andmask]; `widths` = per-plane device int32[P] (train mode) or NULL + width_all */
int mg_bits_dilate(const void* in, void* out, const void* andmask, int P, int H, int W, const int32_t* widths, int width_all,
                   void* stream);
/* SparseConv2d(k3,s2,p1) output-site rule: coarse is [P, ceil(Hf/2), ceil(Wf/2)] */
int mg_bits_downsample(const void* fine, void* coarse, int P, int Hf, int Wf, void* stream);
/* per-row exclusive scan: rowoff[P*H + 1] (rowoff[P*H] = number of active sites), wordoff[P*H*Ww] = rank of each word */
int mg_bits_rank(const void* bits, int P, int H, int W, int32_t* counts_tmp, int32_t* rowoff, int32_t* wordoff, void* stream);
/* Bounded sparse-head capacity: keep at most `cap` active sites of a level in rank order (bits cleared in place, *count clamped, *overflow set to 1
 * when sites were dropped; wordoff stays valid for the kept sites). The host raises on the sticky overflow flag at its next flag read. */
int mg_bits_truncate(void* bits, const int32_t* wordoff, int P, int H, int W, int cap, int32_t* count, int32_t* overflow, void* stream);
/* coords[R,3] = (plane, y, x) of the active sites in sorted order (== torch.nonzero) */
int mg_bits_coords(const void* bits, const int32_t* wordoff, int P, int H, int W, int32_t* coords, void* stream);
/* gather tables [R, k*k] into the (src_bits, src_wordoff) level. kind 0: submanifold k x k; kind 1: inverse conv
 * (rows fine, source coarse, tap valid iff fine = 2*coarse - 1 + tap); kind 2: strided (rows coarse, source fine) */
int mg_gather_table(const int32_t* coords, int R, int ksize, int kind, const void* src_bits, const int32_t* src_wordoff, int Hs,
                    int Ws, int32_t* nbr, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Dense <-> sparse gathers, alpha-plane kernels, input packing
 * (maggie/network/decoder/resnet_inst_matt_spconv.py:161-194,221-232,247-251,264-268,303-304,360-362;
 *  maggie/network/encoder/resnet.py:211-229).
 * ------------------------------------------------------------------------------------------------------------- */
/* out[r, yoff+c] = dense[plane/n_i, y, x, c] * (mul ? mul[plane/n_i, plane%n_i, c] : 1) */
int mg_gather_rows(const void* dense, int dtype, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C, const float* mul,
                   int mul_ninst, void* out, int ldo, int yoff, void* stream);
int mg_gather_rows_bwd(const void* dout, int dtype, int ldo, int yoff, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C,
                       const float* mul, int mul_ninst, const void* dense, float* ddense, float* dmul, void* stream);
/* atomic-free d(dense): each dense pixel sums the rows of the instance planes active there (bits/wordoff of that level) */
/* dmul[frame, inst, c] = sum over the rows r of plane (frame, inst) of dout[r, yoff + c] * dense[frame, y_r, x_r, c] (fp32 [N, mul_ninst, C],
 * OVERWRITTEN): the token-multiplier gradient of mg_gather_rows without atomics -- per-plane row ranges (coords sorted by plane), fixed-order sums. */
int mg_gather_rows_dmul_det(const void* dout, int dtype, int ldo, int yoff, const int32_t* coords, int R, int n_i, int N, int Hd, int Wd, int C,
                            int mul_ninst, const void* dense, float* dmul, const int32_t* r_dev, void* stream);
int mg_gather_rows_bwd_dense(const void* dout, int dtype, int ldo, int yoff, const void* bits, const int32_t* wordoff, int n_i, int N,
                             int Hd, int Wd, int C, const float* mul, int mul_ninst, void* ddense, void* stream);
/* plane[P,H,W] = fill everywhere, vals[r, col] at the active sites (SparseConvTensor.dense() - 99 trick) */
int mg_scatter_plane(const void* vals, int dtype, int ldv, int col, const int32_t* coords, int R, int P, int H, int W, float fill,
                     float* plane, void* stream);
int mg_gather_plane(const float* plane, const int32_t* coords, int R, int H, int W, void* vals, int dtype, int ldv, int col,
                    void* stream);
/* x[N,H,W,8] = [RGB, mask-ID embedding (3), 0, 0] from NCHW fp32 image, (N,n_m,Hm,Wm) masks (nearest upsampled) */
int mg_mask_embed(const float* image, const float* masks, const float* table, int N, int H, int W, int n_m, int Hm, int Wm,
                  int n_embed, void* out, int dtype, void* stream);
int mg_mask_embed_bwd(const void* dx, int dtype, const float* masks, int N, int H, int W, int n_m, int Hm, int Wm, int n_embed,
                      float* dtable, void* stream);
/* out[N,C,h*s,w*s] (fp32 planes) = (tanh(bilinear_up(in)) + 1)/2 ; `in` addressed with explicit strides */
int mg_upsample_tanh(const void* in, int dtype, long sn, long sc, long sy, long sx, int N, int C, int h, int w, int scale,
                     int apply_tanh, float* out, void* stream);
int mg_upsample_tanh_bwd(const float* dout, const float* out, long sn, long sc, long sy, long sx, int N, int C, int h, int w,
                         int scale, int apply_tanh, float* din, void* stream);
/* The same with a 0 / 1 scale per output plane (`pscale` [N*C] or NULL: `x_os8 * valid_masks`, resnet_inst_matt_spconv.py:331, without a second
 * pass over the planes) and `any_nonzero` (or NULL; a pre-zeroed int32 set to 1 when any output element is non-zero: the `x_os8.sum() == 0` test
 * of :314 without a reduction over the planes). The backward masks the gradient of the planes scaled by 0. */
int mg_upsample_tanh_ex(const void* in, int dtype, long sn, long sc, long sy, long sx, int N, int C, int h, int w, int scale,
                        int apply_tanh, float* out, const float* pscale, int32_t* any_nonzero, void* stream);
int mg_upsample_tanh_bwd_ex(const float* dout, const float* out, long sn, long sc, long sy, long sx, int N, int C, int h, int w,
                            int scale, int apply_tanh, float* din, const float* pscale, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * SpectralNorm weight preparation (maggie/network/module/spectral_norm.py:22-35,73-80): one power iteration on every
 * forward (u, v updated in place), sigma = u^T W v, and W/sigma written in the conv kernels' (Cout, taps, Cin_pad) layout.
 * W: fp32 [A][B][taps] (Conv2d: A=Cout,B=Cin; ConvTranspose2d (transposed=1): A=Cin,B=Cout). work: fp32 [B*taps + A + 4].
 * ------------------------------------------------------------------------------------------------------------- */
int mg_spectral_norm(const float* W, float* u, float* v, int A, int B, int taps, int transposed, int pad_in, void* out,
                     int out_dtype, float* work, void* stream);
/* dW = G/sigma - (<G,W>/sigma^2) u v^T ; G fp32 in the (Cout, taps, pad_in) layout; u, v, work as left by the forward */
int mg_spectral_norm_bwd(const float* G, const float* W, const float* u, const float* v, int A, int B, int taps, int transposed,
                         int pad_in, float* work, float* dW, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Instance-token <-> feature cross attention (fp32, one head, D = 128, T = 10 tokens) -- replaces the
 * nn.MultiheadAttention calls of CrossAttentionLayer (maggie/network/module/mask_attention.py:63-133) inside
 * InstanceMatteDecoder.forward (maggie/network/module/instance_matte_decoder.py:170-236). The token side (Q/K/V/out
 * projections of 10 tokens, ID position table) is folded into small matrices by the caller; these kernels make one pass
 * over the (B, L, D) feature rows per direction. ids[b,l] in [0, NID) selects the ID position row of a feature pixel.
 *   tok_fwd : S[t,l] = (qk[t].F[l] + btab[t,ids[l]])*scale; p = softmax_l S (B,T,L); ctx[t] = sum_l p F[l] (B,T,D)
 *   tok_bwd : given dctx (B,T,D) and dp (B,T,L or NULL): dqk, dbtab (B,T,NID), dfeat (B,L,D); gbuf (B,T,L), rowdot (B,T) scratch
 *   feat_fwd: S[l,t] = (F[l].kq[t] + b2[ids[l],t])*scale, -inf where pad_mask[b,t]; p = softmax_t (B,L,T);
 *             out[l] = sum_t p vp[t] + obias (B,L,D)
 *   feat_bwd: given dout: dfeat, dkq, dvp (B,T,D), db2 (B,NID,T), dobias (D or NULL)
 * Accumulated outputs are zeroed by the entry points. Returns -3 for unsupported D / T.
 * ------------------------------------------------------------------------------------------------------------- */
int mg_attn_tok_fwd(const float* qk, const float* btab, const float* feat, const int32_t* ids, int B, int T, int L, int D, int NID,
                    float scale, float* p, float* ctx, void* stream);
int mg_attn_tok_bwd(const float* p, const float* feat, const float* qk, const int32_t* ids, const float* dctx, const float* dp, int B, int T,
                    int L, int D, int NID, float scale, float* gbuf, float* rowdot, float* dqk, float* dbtab, float* dfeat, void* stream);
int mg_attn_feat_fwd(const float* feat, const float* kq, const float* b2, const float* vp, const float* obias, const uint8_t* pad_mask,
                     const int32_t* ids, int B, int T, int L, int D, int NID, float scale, float* out, float* p, void* stream);
int mg_attn_feat_bwd(const float* dout, const float* p, const float* feat, const float* kq, const float* vp, const int32_t* ids, int B, int T,
                     int L, int D, int NID, float scale, float* dfeat, float* dkq, float* dvp, float* db2, float* dobias, void* stream);
/* Round 5: b2_tn != 0 -- the score-bias table b2 and its gradient db2 are laid out (B, T, NID), the layout the token-side linear that forms the table
 * writes (no transposed copy in front of / behind the kernels). */
int mg_attn_feat_fwd_ex(const float* feat, const float* kq, const float* b2, const float* vp, const float* obias, const uint8_t* pad_mask,
                        const int32_t* ids, int B, int T, int L, int Dm, int NID, float scale, float* out, float* p, int b2_tn, void* stream);
int mg_attn_feat_bwd_ex(const float* dout, const float* p, const float* feat, const float* kq, const float* vp, const int32_t* ids, int B, int T,
                        int L, int Dm, int NID, float scale, float* dfeat, float* dkq, float* dvp, float* db2, float* dobias, int b2_tn, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused matting losses on fp32 planes [P,H,W] (maggie/network/arch/maggie.py:237-266,290-346; maggie/network/loss.py:67-191):
 * weighted L1, Sobel-gradient L1, 3-level Laplacian-pyramid L1 -- forward sums and exact backward. `flags[p]` = the
 * plane has any non-zero weight (planes with all-zero weights contribute nothing and are skipped).
 * ------------------------------------------------------------------------------------------------------------- */
int mg_plane_flags(const float* w, int P, int HW, int32_t* flags, void* stream);
/* as_float != 0: flags is float[P], 0.0 / 1.0 (a per-plane scale: `valid_masks`, resnet_inst_matt_spconv.py:320) */
int mg_plane_flags_ex(const float* w, int P, int HW, void* flags, int as_float, void* stream);
/* loss weight of the OS8 prediction (arch/maggie.py:271-281): out = [plane p of gt has a positive pixel] + (reweight && (gt or a8 in
 * [1/255, 254/255])); fp32 planes [P][HW]; flags_scratch: P int32 */
int mg_os8_weight(const float* gt, const float* a8, int P, long HW, int reweight, int32_t* flags_scratch, float* out, void* stream);
/* the same with a8 standing for a8 * pvalid[plane] (pvalid: int32 [P] 0 / 1 or NULL) */
int mg_os8_weight_ex(const float* gt, const float* a8, int P, long HW, int reweight, int32_t* flags_scratch, float* out, const int32_t* pvalid,
                     void* stream);
/* sums = 32 replicas x 16 floats (512 zeroed floats; a workgroup adds to replica `its index mod 32`: same-address atomics were the kernels' bound),
 * each replica [l1, grad, w, lap0, w0, lap1, w1, lap2, w2, ...] (accumulated by mg_loss_point_fwd / mg_pyr_lap_fwd; mg_pyr_lap_fwd takes
 * `sums + 3 + 2 * level`); the replicas are summed here -> out3 = (rec, lap, grad)
 * with the reference's normalisations (arch/maggie.py:237-262: eps 1e-8 on the L1 term; loss.py:137-191: eps 1e-6, 3-fold LapLoss);
 * mg_loss_coef: upstream gradient g3 of those three -> the five per-term coefficients the backward kernels take. */
int mg_loss_finish(const float* sums, float* out3, void* stream);
int mg_loss_coef(const float* g3, const float* sums, float* coef5, void* stream);
/* d = p - t; sums[0] += w|d|, sums[1] += |sobel(p*w) - sobel(t*w)|, sums[2] += w. `pvalid` (int32 [P] or NULL): planes with pvalid == 0 are
 * evaluated with p = 0 (`pred * valid_masks`, arch/maggie.py:112-118, folded into the loss kernels; their gradient is zero) */
int mg_loss_point_fwd(const float* p, const float* t, const float* w, const int32_t* flags, int P, int H, int W, float* d,
                      float* sums, const int32_t* pvalid, void* stream);
/* out[P,h/2,w/2] = (gauss5 (reflect) * x)[::2, ::2] */
int mg_pyr_down(const float* x, const int32_t* flags, int P, int h, int w, float* out, void* stream);
/* L = x - 4*gauss5*zero_stuff(down); sums[0] += |L|*wl, sums[1] += wl, G = wl*sign(L); wl = w0[y<<lvl, x<<lvl] */
int mg_pyr_lap_fwd(const float* x, const float* down, const float* w0, int lvl, int H0, int W0, const int32_t* flags, int P, int h,
                   int w, float* G, float* sums, void* stream);
/* r[P,h/2,w/2] = add - U^T(coef*q) ;  dd[P,h,w] = coef*q + D^T(r)   (adjoints of the pyramid's up / down operators) */
int mg_pyr_upT(const float* q, const float* coef, const float* add, const int32_t* flags, int P, int h, int w, float* r, void* stream);
int mg_pyr_downT(const float* r, const float* q, const float* coef, const int32_t* flags, int P, int h, int w, float* dd, void* stream);
/* dp = coef_rec*w*sign(p-t) + dd + coef_grad*w*SobelAdjoint(...)  (A, B: [P,H,W] scratch) */
int mg_loss_point_bwd(const float* p, const float* t, const float* w, const int32_t* flags, int P, int H, int W, const float* coef_rec,
                      const float* coef_grad, const float* dd, float* A, float* B, float* dp, const int32_t* pvalid, void* stream);

/* The whole fused-loss pipeline of S (<= 3) output scales per call (alpha_os1 / os4 / os8 of arch/maggie.py:283-300 share shape and target): scale s
 * reads p[s] and w[s] ([P][H][W] fp32 each) against the shared target t; pvalid as in mg_loss_point_fwd. Scratch buffers hold S * P planes,
 * scale-major: flags [S*P], d / G0 / dd0 / A / B / dp at (H, W), down0 / G1 / r0 / dd1 at (H/2, W/2), down1 / G2 / r1 / dd2 at (H/4, W/4), down2 / r2 at
 * (H/8, W/8); sums [S][512] accumulators, out [S][3] = (rec, lap, grad) per scale, g [S][3] their upstream gradients, coef [S][5]. H, W multiples of 8. */
int mg_matting_losses_fwd(const float* const* p, const float* t, const float* const* w, const int32_t* pvalid, int S, int P, int H, int W, int32_t* flags,
                          float* d, float* down0, float* down1, float* down2, float* G0, float* G1, float* G2, float* sums, float* out, void* stream);
int mg_matting_losses_bwd(const float* g, const float* sums, const float* const* p, const float* t, const float* const* w, const int32_t* pvalid,
                          const int32_t* flags, int S, int P, int H, int W, const float* G0, const float* G1, const float* G2, float* coef, float* r2,
                          float* dd2, float* r1, float* dd1, float* r0, float* dd0, float* A, float* B, float* dp, void* stream);

/* Batched SpectralNorm: all wrapped convolutions of a model in five launches. `descs` lives in device memory. */
typedef struct mg_sn_desc {
    const float* W;      /* fp32 parameter weight_bar, [A][B][taps]                                   */
    float* u;            /* weight_u [A]  (updated in place)                                          */
    float* v;            /* weight_v [B*taps] (updated in place)                                      */
    int64_t out_off;     /* element offset of this conv's (Cout, taps, pad_in) block in the output    */
    int64_t work_off;    /* float offset of this conv's [v (B*taps) | u (A) | scratch (4)] block      */
    int64_t dw_off;      /* float offset of this conv's gradient block (backward)                     */
    int32_t A, B, taps, transposed, pad_in;
    int32_t plain;       /* != 0: an ordinary (not spectrally normalised) conv weight: converted / transposed with sigma = 1, u and v unused;
                            its gradient comes back unchanged in the parameter's layout                      */
    int32_t k3_first, k3_count; /* this conv's run of (16 x 32) parameter tiles in items_k3: the backward adds their <G, W> partials in item order */
} mg_sn_desc;
int mg_spectral_norm_batched(const mg_sn_desc* descs, int n_conv, const int32_t* items_k1, int n1, const int32_t* items_k2, int n2,
                             const int32_t* items_k3, int n3, float* work_base, long work_floats, void* out_base, void* out_t_base,
                             int out_dtype, void* stream);   /* out_t_base (or NULL): same offsets, (Cin_pad, taps, Cout) copies for dgrad */
/* items_k1: (conv, block of 32 columns of W^T u, -, -); items_k2: (conv, group of 4 rows of W v, -, -); items_k3: (conv, a tile, b tile, -).
 * Every sum of the pipeline (W^T u, the two norms, <G, W>) is formed in a fixed order: no atomics, bit-reproducible u / v / sigma.
 * dot_part: n3 floats of scratch (per-tile partials of <G, W>). */
int mg_spectral_norm_batched_bwd(const mg_sn_desc* descs, int n_conv, const int32_t* items_k3, int n3, const void* const* Gptrs,
                                 int g_dtype, float* work_base, float* dW_base, float* dot_part, void* stream);
/* Round 5: the same with (a) per-conv destinations -- dWptrs: device array of n_conv float* (NULL entries / NULL array: dW_base + dw_off), so that
 * the gradients land in the caller's own storage (the optimizer's flat buffer) without a gather copy; (b) bit 0 of a ConvTranspose weight's
 * Gptrs entry set = that gradient is laid out like the twin, (Cin_pad, taps, Cout) -- what the role-swapped weight-gradient GEMM of a
 * transposed convolution writes -- and is read as such (no permute + copy in front). mg_spectral_norm_batched now emits the twin of ConvTranspose
 * weights as well (out_t_base). */
int mg_spectral_norm_batched_bwd_to(const mg_sn_desc* descs, int n_conv, const int32_t* items_k3, int n3, const void* const* Gptrs,
                                    int g_dtype, float* work_base, float* dW_base, float* const* dWptrs, float* dot_part, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Temporal (video) elementwise kernels. Rows x channels (NHWC) in `dtype`; rz = first gate conv's output (M, 2C) = [r | z] before
 * the sigmoid, cpre = second gate conv's output (M, C) before the tanh.
 *   mg_gru_gate_fwd : xrh (M,2C) = [x | sigmoid(r) * h]                      (maggie/network/module/conv_gru.py:24-25)
 *   mg_gru_out_fwd  : hn = (1 - sigmoid(z)) * h + sigmoid(z) * tanh(cpre)    (conv_gru.py:25-26)
 *   mg_gru_out_bwd  : from dhn: drz[:, C:], dc_pre, dh_part = dhn * (1 - z)
 *   mg_gru_gate_bwd : from dxrh: dx = dxrh[:, :C], drz[:, :C], dh_part = dxrh[:, C:] * r
 *   mg_temporal_fuse: eval-time alpha-level aggregation over frames (t-1, t, t+1) of maggie/network/arch/maggie_temp.py:34-77,
 *                     in place on fp32 planes (n_frames >= 3, P, H*W; t+1 = the LAST frame, :48, frames 1 and 2 are rewritten): thresholds the difference maps at 0.5, propagates t-1 -> t and t+1 -> t,
 *                     keeps the model's own prediction where the two disagree, then propagates t -> t+1.
 * ------------------------------------------------------------------------------------------------------------- */
int mg_gru_gate_fwd(const void* rz, const void* x, const void* h, int dtype, int M, int C, void* xrh, void* stream);
int mg_gru_gate_bwd(const void* dxrh, const void* rz, const void* h, int dtype, int M, int C, void* dx, void* drz, void* dh_part, void* stream);
int mg_gru_out_fwd(const void* rz, const void* cpre, const void* h, int dtype, int M, int C, void* hn, void* stream);
int mg_gru_out_bwd(const void* dhn, const void* rz, const void* cpre, const void* h, int dtype, int M, int C, void* drz, void* dc_pre,
                   void* dh_part, void* stream);
int mg_temporal_fuse(float* alphas, const float* prev, const float* df, const float* db, long plane_elems, int n_frames, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Sparse refinement head, parameter side (maggie/network/decoder/resnet_inst_matt_spconv.py:69-130: the spconv layers' weights
 * (Cout, k, k, Cin) and biases, and inst_spec_layer's two nn.Linear). One launch converts ALL of them for a step, one launch
 * brings all their gradients back:
 *   backward == 0: src fp32 parameter (cout, taps, cin) -> dst (cout_pad, taps, cin_pad) zero padded, in `dtype`;
 *                  dst_t (or NULL) (cin_pad, taps, cout_pad), taps reversed when flip_t (input-gradient operand of a
 *                  submanifold conv). A bias is an entry with cout = taps = 1.
 *   backward != 0: src (cout_pad, taps, cin_pad) in `dtype` (NULL = no gradient) -> dst fp32 (cout, taps, cin).
 * `entries` is a HOST array (n <= MG_WB_MAX_ENTRIES); it travels in the kernel arguments.
 *   mg_bias_act_bwd: backward of a "+bias, ReLU" epilogue over rows x C: g = dy * (y > 0) (y, g NULL: no activation) and
 *                  db[c] = sum_m g[m,c] (db NULL: no bias), one pass.
 * ------------------------------------------------------------------------------------------------------------- */
#define MG_WB_MAX_ENTRIES 64
typedef struct mg_wb_entry {
    const void* src;
    void* dst;
    void* dst_t;
    int32_t cout, taps, cin, cout_pad, cin_pad, flip_t, dtype, reserved;
} mg_wb_entry;
int mg_weight_bank(const mg_wb_entry* entries, int n, int backward, void* stream);
int mg_bias_act_bwd(const void* dy, const void* y, void* g, int dtype, int M, int C, float* db, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Optimizer step of the measured train step (maggie/engine/train.py:274,281: clip_grad_norm_(0.01), AdamW of engine/optim.py:110-118)
 * over ONE flat fp32 buffer holding every trainable parameter (p, g, m, v: n floats each, n % 4 == 0, 16-byte aligned).
 * sumsq_scratch != NULL: the squared gradient norm is reduced into it first (1 device double) and, with max_norm > 0, gradients
 * are scaled by min(1, max_norm / (norm + 1e-6)) inside the update (no separate scaling pass); norm_out (or NULL) receives the norm.
 * bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t) for step t (computed by the host mirror).
 * phases: bit 0 = reduce the norm of g[0:n] into sumsq_scratch, bit 1 = update [0:n] (3 = both). Parameters that got no gradient
 * this step are skipped like torch does: the host reduces the norm over the whole buffer once (phases 1), then updates each
 * contiguous run of parameters that have one (phases 2, same scratch).
 * ------------------------------------------------------------------------------------------------------------- */
int mg_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float bias_correction1, float bias_correction2_sqrt, double* sumsq_scratch, float max_norm, float* norm_out, int phases,
                  void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Input side: the tensor work of the reference's DataLoader after decoding, from uint8 buffers.
 *   mg_preprocess_image : ToTensor + Normalize (maggie/dataloader/transforms.py:720-778): in uint8 [frames][HW][3] ->
 *                         out fp32 [frames][3][HW] = (x / 255 - mean[c]) / std[c]  (IEEE divisions: bit-exact). HW % 4 == 0.
 *   mg_preprocess_planes: alpha / mask planes (transforms.py:744 `alphas < 5 -> 0`; maggie/dataloader/him.py:157-173): in uint8
 *                         [frames][n_in][H][W] -> out fp32 [frames][n_slots][Ho][Wo] = v / 255 of plane src_of_slot[frame*n_slots
 *                         + slot] (device int32; < 0 = empty slot -> zeros; NULL = identity), values below `thresh` -> 0,
 *                         sampled like F.interpolate(mode="nearest") when (Ho, Wo) != (H, W).
 * ------------------------------------------------------------------------------------------------------------- */
int mg_preprocess_image(const uint8_t* in, float* out, const float* mean3, const float* std3, long frames, long HW, void* stream);
int mg_preprocess_planes(const uint8_t* in, float* out, const int32_t* src_of_slot, int frames, int n_in, int n_slots, int H, int W,
                         int Ho, int Wo, int thresh, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Validation metrics on the device (maggie/utils/metric.py). fp32 planes, fp64 results. `trimap` may be NULL;
 * mask_mode: 0 = all ones, 1 = (trimap > 0) (Metric.update :47), 2 = (trimap == 1) (dtSSD.update :427).
 *   mg_metric_plane_sums: out[P][3] = per plane { sum |pred-gt| m, sum (pred-gt)^2 m, sum m }      (SAD :68-78, MSE :80-90, MAD :92-97)
 *   mg_metric_grad      : out[P] = per plane sum (|G(gt_n)| - |G(pred_n)|)^2 m, G = 9x9 Gaussian-derivative pair (filter_x81,
 *                         filter_y = its transpose, zero padding), x_n = (x - min) / (max - min + 1e-6) over the WHOLE tensor
 *                         (:388-405); scratch4: 4 int32 of device scratch
 *   mg_metric_dtssd     : pred/gt/trimap [B][T][N][HW]; out[N] = sum_{b,t<T-1,px} ((p[t+1]-p[t]) - (g[t+1]-g[t]))^2 m[t]  (:436-441)
 * ------------------------------------------------------------------------------------------------------------- */
int mg_metric_plane_sums(const float* pred, const float* gt, const float* trimap, int mask_mode, int P, long HW, double* out, void* stream);
int mg_metric_grad(const float* pred, const float* gt, const float* trimap, int mask_mode, int P, int H, int W, const float* filter_x81,
                   int32_t* scratch4, double* out, void* stream);
int mg_metric_dtssd(const float* pred, const float* gt, const float* trimap, int mask_mode, int B, int T, int N, long HW, double* out,
                    void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Inference post-path: `reverse_transform_tensor` (maggie/utils/postprocessing.py:36-64: crop the
 * bottom/right padding to (crop_h, crop_w), bilinear resize with align_corners=True to (Hout, Wout)) fused with the alpha
 * snapping of maggie/engine/test.py:139-142,229-231 (<= 1/255 -> 0, >= 254/255 -> 1 when `snap`). fp32 planes [P,Hin,Win].
 * ------------------------------------------------------------------------------------------------------------- */
int mg_postprocess_alpha(const float* in, int P, int Hin, int Win, int crop_h, int crop_w, int Hout, int Wout, int snap, float* out,
                         void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Sparse refinement head with a DEVICE row count (round 2).
 * The reference's detail stage (maggie/network/decoder/resnet_inst_matt_spconv.py:196-270) sizes every spconv feature matrix from
 * `torch.nonzero` (:206) -- a device->host read per forward, after which ~430 small launches are paced by the host. Here the site
 * counts of the index pyramid stay in device words (mg_bits_rank: rowoff[P*H]); feature matrices are (capacity x C) buffers and
 * every kernel below runs over min(*rows_dev, capacity) rows from a fixed grid (mg_conv_params.m_dev / mg_rowwise_params.m_dev for
 * the implicit-GEMM and BatchNorm entry points). `*_dev` = the entry point of the same name with the extra `rows_dev` argument
 * (NULL: identical to the plain one).
 * ------------------------------------------------------------------------------------------------------------- */
int mg_gather_rows_dev(const void* dense, int dtype, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C, const float* mul,
                       int mul_ninst, void* out, int ldo, int yoff, const int32_t* rows_dev, void* stream);
int mg_gather_rows_bwd_dev(const void* dout, int dtype, int ldo, int yoff, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C,
                           const float* mul, int mul_ninst, const void* dense, float* ddense, float* dmul, const int32_t* rows_dev, void* stream);
int mg_scatter_plane_dev(const void* vals, int dtype, int ldv, int col, const int32_t* coords, int R, int P, int H, int W, float fill,
                         float* plane, const int32_t* rows_dev, void* stream);
/* zero_rest != 0: the other ldv - 1 columns of every live row are zeroed (gradient of a 1-channel output padded to a 16-byte row) */
int mg_gather_plane_dev(const float* plane, const int32_t* coords, int R, int H, int W, void* vals, int dtype, int ldv, int col,
                        const int32_t* rows_dev, int zero_rest, void* stream);
int mg_gather_table_dev(const int32_t* coords, int R, int ksize, int kind, const void* src_bits, const int32_t* src_wordoff, int Hs,
                        int Ws, int32_t* nbr, const int32_t* rows_dev, void* stream);
int mg_colstats_dev(const void* x, int dtype, int M, int C, int ld, float* stats, const int32_t* rows_dev, void* stream);
int mg_colstats_centered_dev(const void* x, int dtype, int M, int C, int ld, float* stats, int have_sum, const int32_t* rows_dev, void* stream);
int mg_bias_act_bwd_dev(const void* dy, const void* y, void* g, int dtype, int M, int C, float* db, const int32_t* rows_dev, void* stream);
/* out = a * sigmoid(g): instance-specific guidance, `detail * guidance` of resnet_inst_matt_spconv.py:188-193 (a may be a channel slice
 * of a wider buffer: row pitch lda); backward da = dout * s, dg = dout * a * s * (1 - s) */
int mg_rows_sigmoid_mul_fwd(const void* a, int lda, const void* g, void* out, int dtype, int M, int C, const int32_t* rows_dev, void* stream);
int mg_rows_sigmoid_mul_bwd(const void* dout, const void* a, int lda, const void* g, void* da, void* dg, int dtype, int M, int C,
                            const int32_t* rows_dev, void* stream);
/* out = a + b over the live rows (gradient of a feature matrix with two consumers) */
int mg_rows_add(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int dtype, int M, int C, const int32_t* rows_dev, void* stream);
/* nn.Dropout(p) of FFNLayer (maggie/network/module/mask_attention.py:170-182) with a counter-based mask: keep = hash(state, salt, element)
 * >= p; `state` = device int64[2] (seed, step). The backward calls the same entry point on the gradient (same state and salt). */
int mg_rows_dropout(const void* x, void* y, int dtype, int M, int C, float p, const int64_t* state, int salt, const int32_t* rows_dev, void* stream);
/* y = LayerNorm(x + r) * gamma + beta per row (post-norm residual of FFNLayer, mask_attention.py:176-181); rstat [M][2] = (mean, rstd).
 * Backward: dz (= dx = dr), dgamma / dbeta [C] (zeroed here, fp32). C / (8 bf16 | 4 fp32) must be a power of two <= 64. */
int mg_rows_add_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, float eps, void* y, float* rstat, int dtype,
                              int M, int C, const int32_t* rows_dev, void* stream);
int mg_rows_add_layernorm_bwd(const void* dy, const void* x, const void* r, const float* gamma, const float* rstat, void* dz, float* dgamma,
                              float* dbeta, int dtype, int M, int C, const int32_t* rows_dev, void* stream);
/* resnet_inst_matt_spconv.py:347-348 ("dummy code to prevent all zeros"): if *count == 0, set bits [y0:y1, x0:x1] of every plane */
int mg_bits_patch_if_empty(void* bits, const int32_t* count, int P, int H, int W, int y0, int y1, int x0, int x1, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Temporal-consistency tail of the video model (maggie_amd/csrc/temporal2.hip), fp32 planes.
 *   mg_bifuse_fwd/_bwd : bidirectional alpha fusion of maggie/network/decoder/resnet_inst_matt_spconv_temp.py:35-79 over the T frames
 *                        of a clip. preds / fused (B,T,NI,HW); diffs (2(T-1), B, HW) = the difference logits in the reference's call
 *                        order (forward pairs, then backward pairs); fdiff / bdiff (B,T,HW) = the reference's forward_diffs /
 *                        backward_diffs (zero-padded), fsig / bsig their sigmoids. Backward: dpreds, ddiffs (summed over instances).
 *   mg_dtssd_fwd/_bwd  : loss_dtSSD (maggie/network/loss.py:7-16): sums[0] / sums[1]; p, g, m hold T frames of E elements per batch
 *                        item with batch strides *bs (frame slices of a wider tensor need no copy); sig != 0: p = sigmoid(logits)
 *                        (loss_temporal_sparsity, :183-203); m NULL = ones. Backward: dp = gout / sums[1] * d(numerator)/dp.
 *   mg_bce_logits_*    : F.binary_cross_entropy_with_logits(reduction='mean') of the difference logits (:189-190), same strides.
 *   mg_temporal_crop   : eval-time bounding-box crop (:115-142 with utils/utils.py:61-83): in place on alpha [P,H,W] and on the
 *                        detail bit planes [P,H,Ww]; scratch [P,H,W] floats, box int32[P][4].
 * ------------------------------------------------------------------------------------------------------------- */
int mg_bifuse_fwd(const float* preds, const float* diffs, int B, int T, int NI, long HW, float* fused, float* fdiff, float* bdiff, float* fsig,
                  float* bsig, void* stream);
int mg_bifuse_bwd(const float* dfused, const float* preds, const float* diffs, int B, int T, int NI, long HW, float* dpreds, float* ddiffs, void* stream);
int mg_dtssd_fwd(const float* p, long pbs, const float* g, long gbs, const float* m, long mbs, int B, int T, long E, int sig, float* sums, void* stream);
int mg_dtssd_bwd(const float* p, long pbs, const float* g, long gbs, const float* m, long mbs, int B, int T, long E, int sig, const float* sums,
                 const float* gout, float* dp, long dbs, void* stream);
int mg_bce_logits_fwd(const float* x, long xbs, const float* y, long ybs, int B, long TE, float* sum, void* stream);
int mg_bce_logits_bwd(const float* x, long xbs, const float* y, long ybs, int B, long TE, const float* gout, float* dx, long dbs, void* stream);
int mg_temporal_crop(float* alpha, void* bits, int P, int H, int W, float sigma, float thr, int pad, float* scratch, int32_t* box, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Loss bookkeeping (maggie_amd/csrc/losses.hip)
 *   mg_atten_loss_fwd/_bwd : compute_atten_loss of maggie/network/module/instance_matte_decoder.py (guidance_mask [rows, L], attention
 *                            matrix [rows, L], rows = batch * instance slots): out[0] = scale * sum_rows((sum_l gm != 0) - sum_l gm * att);
 *                            terms [rows] scratch. Backward: datt = -scale * gout[0] * gm.
 *   mg_scalar_lincomb/_bwd : out[0] = sum_i coef[i] * ptrs[i][0] over n <= 16 device scalars (ptrs / coef are HOST arrays, passed by value
 *                            into the launch): the loss sums of arch/maggie.py:283-300; backward gin[i] = coef[i] * gout[0].
 * ------------------------------------------------------------------------------------------------------------- */
int mg_atten_loss_fwd(const float* gm, const float* att, int rows, long L, float scale, float* terms, float* out, void* stream);
int mg_atten_loss_bwd(const float* gm, const float* gout, float scale, long n, float* datt, void* stream);
int mg_scalar_lincomb(const float* const* ptrs, const float* coef, int n, float* out, void* stream);
int mg_scalar_lincomb_bwd(const float* coef, int n, const float* gout, float* gin, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Token side of the instance matte decoder (maggie_amd/csrc/token_side.hip): the (batch x 10 tokens) x 128 operations of
 * maggie/network/module/mask_attention.py:9-206 -- projections, FFN / MLP layers, post-norm residual LayerNorms (SelfAttentionLayer
 * :9-60, CrossAttentionLayer :63-133 token side, FFNLayer :170-182, MLP :185-206) -- fp32, single-workgroup kernels.
 *   mg_token_linear_fwd: y[R,N] = LN( res + act( (x + xadd)[R,K] W[N,K]^T + bias ) ); xadd / bias / res / gamma,beta (LayerNorm) may be
 *                        NULL, relu 0/1. With a LayerNorm: z[R,N] (its input) and rstat[R][2] = (mean, rstd) are kept for the backward.
 *   mg_token_linear_bwd: from dy: dx[R,K] (also the gradient of xadd), dW[N,K], db[N], dres[R,N], dgamma / dbeta[N] (NULL when absent);
 *                        yout = the forward output (ReLU mask; only read when relu); dz_scratch [R,N] floats.
 *   mg_token_sa_fwd/bwd: out[b] = softmax(q[b] k[b]^T * scale, keys with pad[b][j] != 0 masked) v[b], prob [B,T,T] kept; T <= 16.
 * ------------------------------------------------------------------------------------------------------------- */
int mg_token_linear_fwd(const float* x, const float* xadd, const float* W, const float* bias, const float* res, int relu, const float* gamma,
                        const float* beta, float eps, float* y, float* z, float* rstat, int R, int K, int N, void* stream);
int mg_token_linear_bwd(const float* dy, const float* x, const float* xadd, const float* W, const float* yout, int relu, const float* gamma,
                        const float* z, const float* rstat, float* dx, float* dW, float* db, float* dres, float* dgamma, float* dbeta, float* dz_scratch,
                        int R, int K, int N, void* stream);
/* mg_token_linear_fwd / _bwd with `wt` != 0: W is given as [K][N] and y = (x + xadd) W (dW is returned as [K][N] too) -- the `x @ wk` products of
 * mask_attention.py (queries folded through the key projection) without a transposed copy of the weight each step. N % 4 == 0. */
int mg_token_linear_fwd_ex(const float* x, const float* xadd, const float* W, const float* bias, const float* res, int relu, const float* gamma,
                           const float* beta, float eps, float* y, float* z, float* rstat, int R, int K, int N, int wt, void* stream);
int mg_token_linear_bwd_ex(const float* dy, const float* x, const float* xadd, const float* W, const float* yout, int relu, const float* gamma,
                           const float* z, const float* rstat, float* dx, float* dW, float* db, float* dres, float* dgamma, float* dbeta,
                           float* dz, int R, int K, int N, int wt, void* stream);
/* Mask pre-processing of the instance matte decoder (instance_matte_decoder.py:131-153; utils.py:16-21): mask [B][NF][n_in][h*s][w*s] fp32 guidance
 * masks (s = integer factor over the OS8 map), gt (or NULL) [B][NF][n_gt][h*gs][w*gs] fp32 alphas -> feat_ids int32 [B][NF*h*w] = max_i (i+1) * [avg-pooled
 * mask_i > 0]; valid uint8 [B][n_i] padded to 4 bytes (zeroed here): 1 where slot i has a mask pixel; guidance fp32 [B][n_i][NF*h*w] = [max-pooled alpha_i > 0]
 * (only with gt). */
int mg_imd_prep(const float* mask, int n_in, int s, const float* gt, int n_gt, int gs, int B, int NF, int h, int w, int n_i, int32_t* feat_ids,
                float* guidance, unsigned char* valid, void* stream);
/* Up to 4 INDEPENDENT token linear layers per launch (the q / k / v projections of the token self-attention, the folded key / table products of a
 * cross attention: module/mask_attention.py:9-206): per layer the semantics of mg_token_linear_fwd_ex / _bwd_ex. Unused pointers are NULL. */
typedef struct mg_tok_lin {
    const float *x, *xadd, *W, *bias, *res, *gamma, *beta;   /* forward inputs ([R,K], [R,K], [N,K] or [K,N] with wt, [N], [R,N], [N], [N]) */
    float *y, *z, *rstat;                                    /* forward outputs (z: pre-LayerNorm values, rstat [R,2]: with gamma)          */
    const float *dy, *yout;                                  /* backward inputs ([R,N]; yout = y of a ReLU layer)                          */
    float *dx, *dW, *db, *dres, *dgamma, *dbeta, *dz;        /* backward outputs (dz: [R,N] scratch; = dy for a plain layer)               */
    int32_t R, K, N, relu, wt;
    float eps;
    int32_t dx_pair;                                         /* backward (round 5): 1 + index of another layer of the call that reads the SAME x (neither
                                                                has an xadd; that layer passes dx = NULL): its dz . W is added into this layer's dx --
                                                                one input gradient for the shared tensor instead of two and an add; 0: none */
} mg_tok_lin;
int mg_token_linear_multi_fwd(const mg_tok_lin* ops, int n, void* stream);
int mg_token_linear_multi_bwd(const mg_tok_lin* ops, int n, void* stream);
/* einsum('bqc,blc->blq') of the instance matte decoder (instance_matte_decoder.py:296-299): logits[b][l][q] = sum_c feat[b][l][c] tok[b][q][c] for
 * q < Q, zero up to the row pitch QP (= 16). feat / out / dlog / dfeat in `dtype` ([B][L][C] / [B][L][QP]); tok / dtok fp32 [B][Q][C] (dtok is
 * overwritten). C = 32 or 64, Q <= 16. The tokens are rounded to `dtype` first (the 1x1 convolution this replaces did the same). */
int mg_token_einsum_fwd(const void* feat, int dtype, const float* tok, int B, int L, int C, int Q, int QP, void* out, void* stream);
int mg_token_einsum_bwd(const void* dlog, const void* feat, int dtype, const float* tok, int B, int L, int C, int Q, int QP, void* dfeat,
                        float* dtok, void* stream);
/* out[i] = ((srcs[0][i] + srcs[1][i]) + srcs[2][i]) + ... : the gradient of a tensor with k <= 16 consumers in one launch and a fixed order (replaces
 * the k - 1 pairwise adds of the autograd engine; maggie/network/module/mask_attention.py:63-133 uses every token tensor several times). fp32. */
int mg_sum_k(const float* const* srcs, int k, long n, float* out, void* stream);
/* The same for tensors of `dtype` (MG_F32 / MG_BF16 / MG_F16; 16-byte aligned): 16-bit terms are added in fp32 in the given order and rounded once. */
int mg_sum_k_t(const void* const* srcs, int k, long n, void* out, int dtype, void* stream);
/* AdaptiveAvgPool2d(1) over NHWC rows (maggie/network/module/aspp.py:24-27,50-52). mode 0: x (N, HW, C) -> out (N, C), the mean of each sample's rows
 * (fp32 sums in a fixed order); mode 1: x = dy (N, C) -> out = dx (N, HW, C) = dy / HW; mode 2: like 0 without the division (the backward of broadcasting
 * the pooled row back over the map). ld: element stride between the rows of x in modes 0 / 2 (0 = C; a channel slice of a wider map is read in place). */
int mg_spatial_mean(const void* x, void* out, int dtype, int N, int HW, int C, int mode, int ld, void* stream);
/* dsts[j][0 .. bytes[j]) = srcs[j][...] for j < k <= 16 as ONE launch (replaces torch._foreach_copy_ of contiguous same-type tensors = one hipMemcpyAsync
 * each: the outputs a replayed graph hands to the caller, engine/train.py:229-241 keeps them across steps; gradient hand-over between graphs). */
int mg_copy_k(const void* const* srcs, void* const* dsts, const long* bytes, int k, void* stream);
/* Any number of copies in one launch, the job list in DEVICE memory: table = int64[4 * jobs], (src, dst, bytes, first 4 KiB block) per job with the first
 * blocks ascending from 0; nblk = total number of 4 KiB blocks. (A captured graph fills the table once, after the capture.) */
int mg_copy_table(const long* table, int jobs, long nblk, void* stream);
/* out[4] (int32, device) = [a NaN among the n fp32 tokens, nonzero[0] == 0, ovf[0] != 0, err[0]] -- NULL inputs give 0. The four words the host reads
 * between the trunk and the detail stage (maggie/network/module/mask_attention.py:95-98 raises on NaN tokens; resnet_inst_matt_spconv.py:314 switches
 * the guidance to the ground truth when the coarse alpha is zero), in one launch. */
int mg_step_flags(const float* tokens, long n, const int* nonzero, const int* ovf, const int* err, int* out, void* stream);
int mg_token_sa_fwd(const float* q, const float* k, const float* v, const unsigned char* pad, float scale, int B, int T, int D, float* out, float* prob,
                    void* stream);
int mg_token_sa_bwd(const float* dout, const float* q, const float* k, const float* v, const float* prob, float scale, int B, int T, int D, float* dq,
                    float* dk, float* dv, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Mailbox all-reduce (csrc/mailbox.hip): in-place fp32 sum over the ranks of a pack of <= MG_MAILBOX_PACK floats as ONE kernel launch on `stream`
 * (capturable into hipGraphs), through peer-mapped mailboxes instead of RCCL. Replaces: the statistics all-reduce of nn.SyncBatchNorm
 * (engine/train.py:159-161) when the step is replayed from graphs. Setup: every rank calls mg_mailbox_create (-> its mailbox + a 64-byte IPC handle),
 * exchanges the handles out of band, opens the peers' with mg_mailbox_open and fills mg_mailbox.peer[] (its own pointer at peer[rank]). `seq_dev`
 * (zeroed uint32) and `err_dev` (zeroed int32; set to 1 when a peer did not arrive within `spin_ticks` of the 100 MHz wall clock) are this rank's own
 * device words. Every rank must issue the same sequence of calls.
 * ------------------------------------------------------------------------------------------------------------- */
#define MG_MAILBOX_MAX_RANKS 8
#define MG_MAILBOX_SLOTS 64
#define MG_MAILBOX_PACK 1088
typedef struct mg_mailbox {
    float* peer[MG_MAILBOX_MAX_RANKS]; /* peer[r]: rank r's mailbox as mapped into THIS process */
    int32_t world, rank;
} mg_mailbox;
long mg_mailbox_bytes(int world);
int mg_mailbox_create(int world, void** ptr, void* handle64);
int mg_mailbox_open(const void* handle64, void** ptr);
int mg_mailbox_close(void* ptr);
int mg_mailbox_free(void* ptr);
int mg_mailbox_allreduce(const mg_mailbox* mb, float* data, int n, uint32_t* seq_dev, int32_t* err_dev, long spin_ticks, void* stream);
/* out-of-place form (src is left untouched: SyncBN backward keeps the LOCAL sums for dgamma / dbeta) */
int mg_mailbox_allreduce_to(const mg_mailbox* mb, const float* src, float* dst, int n, uint32_t* seq_dev, int32_t* err_dev, long spin_ticks, void* stream);
/* SyncBN forward statistics in one launch: this rank's [nrep][2C] sums (sum x | sum x^2) and row count -> exchanged -> outs = scale | shift | mean |
 * invstd [4C], *count_out = global count, running statistics updated (the arithmetic of mg_bn_finalize on the pooled moments). 2C + 1 <= MG_MAILBOX_PACK. */
int mg_mailbox_bn_finalize(const mg_mailbox* mb, const float* stats, int nrep, float count, int C, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* outs, float* count_out,
                           uint32_t* seq_dev, int32_t* err_dev, long spin_ticks, void* stream);
/* the same with this rank's row count read from the device (count_dev: int32 [1], NULL = use `count`): BatchNorm1d over the sparse head's live rows */
int mg_mailbox_bn_finalize_dev(const mg_mailbox* mb, const float* stats, int nrep, float count, const int32_t* count_dev, int C, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* outs, float* count_out,
                               uint32_t* seq_dev, int32_t* err_dev, long spin_ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGGIE_HIP_H */
