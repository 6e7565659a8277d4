// Temporal-consistency tail of the video model (gfx950, fp32 planes, HBM-bound elementwise / reduction kernels):
//  * bidirectional alpha fusion of maggie/network/decoder/resnet_inst_matt_spconv_temp.py:35-79 -- the forward and backward recursions
//    pred = prev * (1 - sigmoid(d)) + cur * sigmoid(d) over the frames of a clip, their average, and the sigmoid outputs -- as ONE kernel
//    each way (the reference / round 1: ~12 torch elementwise launches per frame pair);
//  * temporal derivative loss loss_dtSSD of maggie/network/loss.py:7-16 (also on sigmoid(difference logits)) and the BCE-with-logits of
//    loss_temporal_sparsity (:183-203) as fused reductions with exact backward kernels;
//  * the eval-time bounding-box crop (:115-142 + utils/utils.py:61-83): 7x7 smoothing with the reference's kernel (g[j]^2 on every row),
//    crop by the pad, bilinear resize back, threshold 0.1, per-plane bounding box padded by 30 px, applied to the coarse alpha and to
//    the detail bit planes -- no host loop, no torch.nonzero.
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;
constexpr int MAXT = 16;                                 // frames per clip handled in registers
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }
inline int grid_for(long total) { long b = (total + NT - 1) / NT; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

// preds (B,T,NI,HW); diffs (2(T-1), B, HW): slot i-1 = forward pair (i-1 -> i), slot 2T-3-k = backward logit stored at frame k.
// fused (B,T,NI,HW); fdiff / bdiff (B,T,HW) logits (frame 0 resp. T-1 zero); fsig / bsig = their sigmoids
__global__ __launch_bounds__(NT) void bifuse_fwd_kernel(const float* __restrict__ preds, const float* __restrict__ diffs, int B, int T, int NI, long HW,
                                                        float* __restrict__ fused, float* __restrict__ fdiff, float* __restrict__ bdiff,
                                                        float* __restrict__ fsig, float* __restrict__ bsig) {
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int b = (int)(i / HW); const long px = i - (long)b * HW;
        float sf[MAXT], sb[MAXT];
        for (int t = 0; t < T; ++t) {
            const float df = t >= 1 ? diffs[((long)(t - 1) * B + b) * HW + px] : 0.f;
            const float db = t <= T - 2 ? diffs[((long)(2 * T - 3 - t) * B + b) * HW + px] : 0.f;
            sf[t] = sigm(df); sb[t] = sigm(db);
            const long o = ((long)b * T + t) * HW + px;
            fdiff[o] = df; bdiff[o] = db; fsig[o] = sf[t]; bsig[o] = sb[t];
        }
        for (int n = 0; n < NI; ++n) {
            const float* p = preds + ((long)b * T * NI + n) * HW + px;
            float* f = fused + ((long)b * T * NI + n) * HW + px;
            const long st = (long)NI * HW;
            float fp[MAXT], bp;
            fp[0] = p[0];
            for (int t = 1; t < T; ++t) fp[t] = fp[t - 1] * (1.f - sf[t]) + p[t * st] * sf[t];
            bp = p[(T - 1) * st];
            f[(T - 1) * st] = bp;
            for (int t = T - 2; t >= 0; --t) {
                bp = bp * (1.f - sb[t]) + p[t * st] * sb[t];
                f[t * st] = t == 0 ? fp[0] : 0.5f * (fp[t] + bp);
            }
        }
    }
}

// backward of the above w.r.t. preds and the difference logits (summed over the instances)
__global__ __launch_bounds__(NT) void bifuse_bwd_kernel(const float* __restrict__ dfused, const float* __restrict__ preds, const float* __restrict__ diffs,
                                                        int B, int T, int NI, long HW, float* __restrict__ dpreds, float* __restrict__ ddiffs) {
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int b = (int)(i / HW); const long px = i - (long)b * HW;
        float sf[MAXT], sb[MAXT], gsf[MAXT], gsb[MAXT];
        for (int t = 0; t < T; ++t) {
            sf[t] = t >= 1 ? sigm(diffs[((long)(t - 1) * B + b) * HW + px]) : 0.f;
            sb[t] = t <= T - 2 ? sigm(diffs[((long)(2 * T - 3 - t) * B + b) * HW + px]) : 0.f;
            gsf[t] = 0.f; gsb[t] = 0.f;
        }
        const long st = (long)NI * HW;
        for (int n = 0; n < NI; ++n) {
            const long base = ((long)b * T * NI + n) * HW + px;
            const float* p = preds + base; const float* g = dfused + base; float* dp = dpreds + base;
            float pv[MAXT], fp[MAXT], bpv[MAXT], dpl[MAXT];
            for (int t = 0; t < T; ++t) { pv[t] = p[t * st]; dpl[t] = 0.f; }
            fp[0] = pv[0];
            for (int t = 1; t < T; ++t) fp[t] = fp[t - 1] * (1.f - sf[t]) + pv[t] * sf[t];
            bpv[T - 1] = pv[T - 1];
            for (int t = T - 2; t >= 0; --t) bpv[t] = bpv[t + 1] * (1.f - sb[t]) + pv[t] * sb[t];
            // forward chain: fused[0] = fp[0], fused[t] = (fp[t] + bp[t]) / 2 for 0 < t < T-1
            float carry = 0.f;
            for (int t = T - 2; t >= 1; --t) {
                const float gt = 0.5f * g[t * st] + carry;
                dpl[t] += gt * sf[t];
                gsf[t] += gt * (pv[t] - fp[t - 1]);
                carry = gt * (1.f - sf[t]);
            }
            dpl[0] += g[0] + carry;
            // backward chain: fused[T-1] = bp[T-1], fused[t] = (fp[t] + bp[t]) / 2; bp[0] is not used
            const float gcur = g[(T - 1) * st];                  // gradient reaching bp[T-1] directly
            // top-down accumulation for the backward chain: G[t] = dL/dbp[t]; G[0] = 0, G[t] = 0.5 g[t] + G[t-1] (1 - sb[t-1])
            float G = 0.f;
            for (int t = 1; t <= T - 1; ++t) {
                // contribution of bp[t-1] = bp[t] (1 - sb[t-1]) + p[t-1] sb[t-1], whose gradient is G (of bp[t-1])
                const float gprev = G;                           // dL/dbp[t-1]
                dpl[t - 1] += gprev * sb[t - 1];
                gsb[t - 1] += gprev * (pv[t - 1] - bpv[t]);
                G = (t <= T - 2 ? 0.5f * g[t * st] : gcur) + gprev * (1.f - sb[t - 1]);
            }
            dpl[T - 1] += G;
            for (int t = 0; t < T; ++t) dp[t * st] = dpl[t];
        }
        for (int t = 1; t < T; ++t) ddiffs[((long)(t - 1) * B + b) * HW + px] = gsf[t] * sf[t] * (1.f - sf[t]);
        for (int t = 0; t <= T - 2; ++t) ddiffs[((long)(2 * T - 3 - t) * B + b) * HW + px] = gsb[t] * sb[t] * (1.f - sb[t]);
    }
}

// ---- loss_dtSSD: sum_{b, t >= 1, e} ((p[t] - p[t-1]) - (g[t] - g[t-1]))^2 m[t] / sum (m[t] + 1e-6) ------------------------------------
// p / g / m: T frames of E elements, `xbs` elements between batches; `sig`: p = sigmoid(logits); m == NULL: ones
__global__ __launch_bounds__(NT) void dtssd_fwd_kernel(const float* __restrict__ p, long pbs, const float* __restrict__ g, long gbs, const float* __restrict__ m,
                                                       long mbs, int B, int T, long E, int sig, float* __restrict__ sums, float* __restrict__ slots) {
    __shared__ float red[2][NT / 64];
    float a0 = 0.f, a1 = 0.f;
    const long total = (long)B * E;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int b = (int)(i / E); const long e = i - (long)b * E;
        float pp = p[b * pbs + e]; if (sig) pp = sigm(pp);
        float gp = g[b * gbs + e];
        for (int t = 1; t < T; ++t) {
            float pc = p[b * pbs + t * E + e]; if (sig) pc = sigm(pc);
            const float gc = g[b * gbs + t * E + e];
            const float mv = m ? m[b * mbs + t * E + e] : 1.f;
            const float d = (pc - pp) - (gc - gp);
            a0 += d * d * mv; a1 += mv + 1e-6f;
            pp = pc; gp = gc;
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a0; red[1][threadIdx.x >> 6] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s0 = 0.f, s1 = 0.f;
        for (int w = 0; w < NT / 64; ++w) { s0 += red[0][w]; s1 += red[1][w]; }
        if (slots) { slots[2 * blockIdx.x] = s0; slots[2 * blockIdx.x + 1] = s1; }     // deterministic mode: one row per workgroup (csrc/det.hip)
        else { atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); }
    }
}

// dp (same layout as p, batch stride dbs) = gout / sums[1] * d/dp; frames without a neighbour get their single term
__global__ __launch_bounds__(NT) void dtssd_bwd_kernel(const float* __restrict__ p, long pbs, const float* __restrict__ g, long gbs, const float* __restrict__ m,
                                                       long mbs, int B, int T, long E, int sig, const float* __restrict__ sums,
                                                       const float* __restrict__ gout, float* __restrict__ dp, long dbs) {
    const float c = 2.f * gout[0] / sums[1];
    const long total = (long)B * E;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int b = (int)(i / E); const long e = i - (long)b * E;
        float raw_prev = p[b * pbs + e];
        float pp = sig ? sigm(raw_prev) : raw_prev;
        float gp = g[b * gbs + e];
        float acc_prev = 0.f;                                     // gradient accumulated so far on frame t-1 (in p space)
        for (int t = 1; t < T; ++t) {
            const float raw = p[b * pbs + t * E + e];
            const float pc = sig ? sigm(raw) : raw;
            const float gc = g[b * gbs + t * E + e];
            const float mv = m ? m[b * mbs + t * E + e] : 1.f;
            const float d = c * ((pc - pp) - (gc - gp)) * mv;
            const float gprev = acc_prev - d;
            dp[b * dbs + (t - 1) * E + e] = sig ? gprev * pp * (1.f - pp) : gprev;
            acc_prev = d;
            pp = pc; gp = gc;
        }
        dp[b * dbs + (T - 1) * E + e] = sig ? acc_prev * pp * (1.f - pp) : acc_prev;
    }
}

// ---- mean BCE with logits over (B, T, E) with batch strides -----------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void bce_fwd_kernel(const float* __restrict__ x, long xbs, const float* __restrict__ y, long ybs, int B, long TE,
                                                     float* __restrict__ sum, float* __restrict__ slots) {
    __shared__ float red[NT / 64];
    float a = 0.f;
    const long total = (long)B * TE;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int b = (int)(i / TE); const long e = i - (long)b * TE;
        const float xv = x[b * xbs + e], yv = y[b * ybs + e];
        a += fmaxf(xv, 0.f) - xv * yv + log1pf(__expf(-fabsf(xv)));
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.f; for (int w = 0; w < NT / 64; ++w) s += red[w]; if (slots) slots[blockIdx.x] = s; else atomicAdd(sum, s); }
}

__global__ __launch_bounds__(NT) void bce_bwd_kernel(const float* __restrict__ x, long xbs, const float* __restrict__ y, long ybs, int B, long TE,
                                                     const float* __restrict__ gout, float* __restrict__ dx, long dbs) {
    const long total = (long)B * TE;
    const float c = gout[0] / (float)total;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int b = (int)(i / TE); const long e = i - (long)b * TE;
        dx[b * dbs + e] = c * (sigm(x[b * xbs + e]) - y[b * ybs + e]);
    }
}

// ---- eval-time bounding-box crop ---------------------------------------------------------------------------------------------------------
// pass A: h[y][x] = sum_j g2[j] a[y][x + j - 3]  (zero outside the image; g2[j] = (g[j] / sum g)^2, g = exp(-(j-3)^2 / (2 sigma^2)))
__global__ __launch_bounds__(NT) void crop_hblur_kernel(const float* __restrict__ a, int P, int H, int W, float g0, float g1, float g2, float g3,
                                                        float* __restrict__ h) {
    const float gw[7] = {g0, g1, g2, g3, g2, g1, g0};
    const long total = (long)P * H * W;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int x = (int)(i % W); const long r = i / W;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) { const int xx = x + j - 3; if (xx >= 0 && xx < W) s += gw[j] * a[r * W + xx]; }
        h[i] = s;
    }
}

// pass B: smoothed = 7-row box sum of h (zero outside), cropped by 3 on every side, resized back to (H, W) (bilinear, align_corners = False),
// thresholded at 0.1: every pixel above it widens its plane's box [ymin, ymax, xmin, xmax] (atomics on int32[4] per plane)
__global__ __launch_bounds__(NT) void crop_bbox_kernel(const float* __restrict__ h, int P, int H, int W, float thr, int* __restrict__ box) {
    // grid (chunks of a plane, plane): every thread keeps its own box, waves and the workgroup reduce it, ONE atomic set per workgroup
    // (one atomic set per pixel above the threshold serialised ~6 M same-address atomics: 52 ms per call on an all-foreground clip)
    __shared__ int sbox[4][NT / 64];
    const int p = blockIdx.y;
    const int Hc = H - 6, Wc = W - 6;
    const float sy = (float)Hc / (float)H, sx = (float)Wc / (float)W;
    int ymin = 1 << 30, ymax = -1, xmin = 1 << 30, xmax = -1;
    const float* hp = h + (long)p * H * W;
    for (int i = blockIdx.x * NT + threadIdx.x; i < H * W; i += gridDim.x * NT) {
        const int y = i / W, x = i - y * W;
        float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
        if (fy < 0.f) fy = 0.f;
        if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hc - 1 ? 1 : 0), x1 = x0 + (x0 < Wc - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        float v[2][2];
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) {
                const int yy = (a_ ? y1 : y0) + 3, xx = (b_ ? x1 : x0) + 3;      // cropped (yy-3, xx-3) = full-size smoothed (yy, xx)
                float s = 0.f;
#pragma unroll
                for (int k = -3; k <= 3; ++k) { const int yk = yy + k; if (yk >= 0 && yk < H) s += hp[(long)yk * W + xx]; }
                v[a_][b_] = s;
            }
        const float val = (1.f - ly) * ((1.f - lx) * v[0][0] + lx * v[0][1]) + ly * ((1.f - lx) * v[1][0] + lx * v[1][1]);
        if (val > thr) { ymin = min(ymin, y); ymax = max(ymax, y); xmin = min(xmin, x); xmax = max(xmax, x); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ymin = min(ymin, __shfl_xor(ymin, o, 64)); ymax = max(ymax, __shfl_xor(ymax, o, 64));
        xmin = min(xmin, __shfl_xor(xmin, o, 64)); xmax = max(xmax, __shfl_xor(xmax, o, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sbox[0][wave] = ymin; sbox[1][wave] = ymax; sbox[2][wave] = xmin; sbox[3][wave] = xmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NT / 64; ++w) {
            ymin = min(ymin, sbox[0][w]); ymax = max(ymax, sbox[1][w]); xmin = min(xmin, sbox[2][w]); xmax = max(xmax, sbox[3][w]);
        }
        if (ymax >= 0) {
            atomicMin(&box[p * 4 + 0], ymin); atomicMax(&box[p * 4 + 1], ymax);
            atomicMin(&box[p * 4 + 2], xmin); atomicMax(&box[p * 4 + 3], xmax);
        }
    }
}

// pass C: inside = [ymin - pad, ymax + pad) x [xmin - pad, xmax + pad) clamped (an empty plane keeps everything, `continue` in the
// reference); alpha *= inside, bits &= inside
__global__ __launch_bounds__(NT) void crop_apply_kernel(float* __restrict__ alpha, unsigned long long* __restrict__ bits, const int* __restrict__ box, int P,
                                                        int H, int W, int Ww, int pad) {
    const long total = (long)P * H * Ww;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int wj = (int)(i % Ww); const long r = i / Ww; const int y = (int)(r % H); const int p = (int)(r / H);
        const int ymin = box[p * 4], ymax = box[p * 4 + 1], xmin = box[p * 4 + 2], xmax = box[p * 4 + 3];
        if (ymax < ymin) continue;                               // nothing above the threshold in this plane
        const int ya = max(ymin - pad, 0), yb = min(ymax + pad, H), xa = max(xmin - pad, 0), xb = min(xmax + pad, W);
        unsigned long long keep = 0ull;
        if (y >= ya && y < yb)
            for (int b = 0; b < 64; ++b) { const int x = wj * 64 + b; if (x >= xa && x < xb) keep |= 1ull << b; }
        if (bits) bits[i] &= keep;
        if (alpha)
            for (int b = 0; b < 64; ++b) { const int x = wj * 64 + b; if (x < W && !((keep >> b) & 1ull)) alpha[((long)p * H + y) * W + x] = 0.f; }
    }
}

__global__ void crop_box_init_kernel(int* __restrict__ box, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) { box[i * 4] = 1 << 30; box[i * 4 + 1] = -1; box[i * 4 + 2] = 1 << 30; box[i * 4 + 3] = -1; }
}

}  // namespace

extern "C" int mg_bifuse_fwd(const float* preds, const float* diffs, int B, int T, int NI, long HW, float* fused, float* fdiff, float* bdiff, float* fsig,
                             float* bsig, void* stream) {
    if (T < 2 || T > MAXT) return -2;
    if ((long)B * HW <= 0) return 0;
    hipLaunchKernelGGL(bifuse_fwd_kernel, dim3(grid_for((long)B * HW)), dim3(NT), 0, (hipStream_t)stream, preds, diffs, B, T, NI, HW, fused, fdiff, bdiff, fsig, bsig);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bifuse_bwd(const float* dfused, const float* preds, const float* diffs, int B, int T, int NI, long HW, float* dpreds, float* ddiffs,
                             void* stream) {
    if (T < 2 || T > MAXT) return -2;
    if ((long)B * HW <= 0) return 0;
    hipLaunchKernelGGL(bifuse_bwd_kernel, dim3(grid_for((long)B * HW)), dim3(NT), 0, (hipStream_t)stream, dfused, preds, diffs, B, T, NI, HW, dpreds, ddiffs);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_dtssd_fwd(const float* p, long pbs, const float* g, long gbs, const float* m, long mbs, int B, int T, long E, int sig, float* sums,
                            void* stream) {
    hipStream_t st = (hipStream_t)stream;
    { hipError_t e = mg_zero_words(sums, 2, st); if (e != hipSuccess) return (int)e; }
    if (T < 2 || (long)B * E <= 0) return 0;
    long blocks = ((long)B * E + NT - 1) / NT; if (blocks > 1024) blocks = 1024;
    float* slots = nullptr;
    if (mg_det_on && blocks > 1) { slots = mg_det_scratch(2 * blocks); if (!slots) return MG_DET_NO_SCRATCH; }
    hipLaunchKernelGGL(dtssd_fwd_kernel, dim3((unsigned)blocks), dim3(NT), 0, st, p, pbs, g, gbs, m, mbs, B, T, E, sig, sums, slots);
    MG_CHECK_LAUNCH();
    if (slots) return mg_det_reduce1(slots, (int)blocks, sums, 2, st);
    return 0;
}

extern "C" int mg_dtssd_bwd(const float* p, long pbs, const float* g, long gbs, const float* m, long mbs, int B, int T, long E, int sig, const float* sums,
                            const float* gout, float* dp, long dbs, void* stream) {
    if (T < 2 || (long)B * E <= 0) return 0;
    hipLaunchKernelGGL(dtssd_bwd_kernel, dim3(grid_for((long)B * E)), dim3(NT), 0, (hipStream_t)stream, p, pbs, g, gbs, m, mbs, B, T, E, sig, sums, gout, dp, dbs);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bce_logits_fwd(const float* x, long xbs, const float* y, long ybs, int B, long TE, float* sum, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    { hipError_t e = mg_zero_words(sum, 1, st); if (e != hipSuccess) return (int)e; }
    if ((long)B * TE <= 0) return 0;
    long blocks = ((long)B * TE + NT - 1) / NT; if (blocks > 1024) blocks = 1024;
    float* slots = nullptr;
    if (mg_det_on && blocks > 1) { slots = mg_det_scratch(blocks); if (!slots) return MG_DET_NO_SCRATCH; }
    hipLaunchKernelGGL(bce_fwd_kernel, dim3((unsigned)blocks), dim3(NT), 0, st, x, xbs, y, ybs, B, TE, sum, slots);
    MG_CHECK_LAUNCH();
    if (slots) return mg_det_reduce1(slots, (int)blocks, sum, 1, st);
    return 0;
}

extern "C" int mg_bce_logits_bwd(const float* x, long xbs, const float* y, long ybs, int B, long TE, const float* gout, float* dx, long dbs, void* stream) {
    if ((long)B * TE <= 0) return 0;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for((long)B * TE)), dim3(NT), 0, (hipStream_t)stream, x, xbs, y, ybs, B, TE, gout, dx, dbs);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_temporal_crop(float* alpha, void* bits, int P, int H, int W, float sigma, float thr, int pad, float* scratch, int32_t* box, void* stream) {
    if (P <= 0 || H < 8 || W < 8) return -2;
    hipStream_t st = (hipStream_t)stream;
    float g[4], s = 0.f;
    for (int j = 0; j < 4; ++j) g[j] = expf(-(float)((j - 3) * (j - 3)) / (2.f * sigma * sigma));
    s = 2.f * (g[0] + g[1] + g[2]) + g[3];
    for (int j = 0; j < 4; ++j) { g[j] /= s; g[j] *= g[j]; }
    const long total = (long)P * H * W;
    hipLaunchKernelGGL(crop_box_init_kernel, dim3((P + 63) / 64), dim3(64), 0, st, box, P);
    hipLaunchKernelGGL(crop_hblur_kernel, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)alpha, P, H, W, g[0], g[1], g[2], g[3], scratch);
    { long cb = ((long)H * W + NT - 1) / NT; if (cb > 128) cb = 128; hipLaunchKernelGGL(crop_bbox_kernel, dim3((unsigned)cb, P), dim3(NT), 0, st, (const float*)scratch, P, H, W, thr, box); }
    const int Ww = (W + 63) / 64;
    hipLaunchKernelGGL(crop_apply_kernel, dim3(grid_for((long)P * H * Ww)), dim3(NT), 0, st, alpha, (unsigned long long*)bits, (const int*)box, P, H, W, Ww, pad);
    MG_CHECK_LAUNCH();
    return 0;
}
