// Fused matting-loss kernels on fp32 alpha planes [P, H, W] (P = frames x instance slots):
//   weighted L1, Sobel-gradient L1 and the 3-level Laplacian-pyramid L1 -- forward sums and analytic backward.
// Reference: maggie/network/arch/maggie.py:237-266,290-346 (regression_loss / compute_loss) and
// maggie/network/loss.py:67-118 (GradientLoss), :120-191 (LapLoss). There the pyramids are ~50 cuDNN/elementwise
// launches per scale over all 10 instance slots; here
//   * the Laplacian pyramid is linear, so lap(pred) - lap(target) = lap(pred - target): one pyramid of d = pred - target;
//   * blur+decimate and zero-stuff+blur are evaluated directly at the surviving samples (reflect padding folded into the index);
//   * planes whose weight map is identically zero (padded instance slots: 8 of 10 in the headline workload) contribute
//     exactly 0 to every sum and are skipped through a per-plane flag;
//   * the backward pass applies the exact adjoints (U^T, D^T with their reflect-border terms) instead of autograd graphs.
// All kernels are HBM/L2-bound stencils; each block reduces its partial sums before one atomicAdd per sum.
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

#define MG_STENCIL_FN __device__ __forceinline__
#include "loss_stencils.h"

constexpr int NT = 256;
__device__ __constant__ float c_g1[5] = {1.f / 16.f, 4.f / 16.f, 6.f / 16.f, 4.f / 16.f, 1.f / 16.f};

__device__ __forceinline__ int refl(int k, int n) { return k < 0 ? -k : (k >= n ? 2 * (n - 1) - k : k); }
__device__ __forceinline__ int clampi(int k, int n) { return k < 0 ? 0 : (k >= n ? n - 1 : k); }

// The loss accumulators are REPLICATED: [LOSS_REPLICAS][LOSS_STRIDE] floats, a workgroup adds to replica (its index mod 32). With one copy every
// workgroup of a launch ended in 2-3 same-address atomics, and those -- not the stencils -- set the kernel time (point_fwd 36 -> 52 -> 86 us
// at 48 / 128 / 256 workgroups per plane); loss_finish / loss_coef sum the replicas.
constexpr int LOSS_REPLICAS = 32, LOSS_STRIDE = 16;
// The three output scales of the model (alpha_os1 / os4 / os8: same (P, H, W) shape, same target, their own weight planes) run through ONE set of
// launches (round 3): plane index pl = scale * Pper + q; the predictions and weights of the scales are separate tensors (P3), the pyramid
// scratch is one (S * Pper)-plane buffer, the accumulators of scale s start at sums + s * LOSS_SUMS. Pper == total planes: a single scale.
struct P3 { const float* a[3]; };
constexpr int LOSS_SUMS = LOSS_REPLICAS * LOSS_STRIDE;

// `slot` (deterministic mode, csrc/det.hip): the workgroup's sum is STORED there; mg_det_reduce adds the workgroups in index order
__device__ __forceinline__ void block_add(float v, float* dst, float* sh, float* slot = nullptr) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < NT / 64; ++i) s += sh[i];
        if (slot) *slot = s;
        else if (s != 0.f) atomicAdd(dst, s);
    }
    __syncthreads();
}

// any(wp[i] > 0) over the workgroup's share of a plane, eight loads in flight per thread (the plain grid-stride loop is one load -> wait per trip)
__device__ __forceinline__ int plane_any_pos(const float* __restrict__ wp, int HW) {
    const int stride = gridDim.x * NT;
    int i = blockIdx.x * NT + threadIdx.x;
    int any = 0;
    for (; i + 7 * stride < HW; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = wp[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) any |= (v[u] > 0.f);
    }
    for (; i < HW; i += stride) any |= (wp[i] > 0.f);
    return any;
}

// flags[p] = any(w[p] > 0)
__global__ __launch_bounds__(NT) void plane_flags_kernel(const float* __restrict__ w, int HW, int* __restrict__ flags, int one) {
    const int p = blockIdx.y;
    const float* wp = w + (long)p * HW;
    const int any = plane_any_pos(wp, HW);
    // every writer stores the same value: a plain store (no read-modify-write) is enough, and one per workgroup instead of one atomic per wave
    if (__syncthreads_or(any) && threadIdx.x == 0) flags[p] = one;
}

__global__ __launch_bounds__(NT) void plane_flags3_kernel(const P3 w, int Pper, int HW, int* __restrict__ flags) {
    const int p = blockIdx.y, sc = p / Pper;
    const float* wp = w.a[sc] + (long)(p - sc * Pper) * HW;
    const int any = plane_any_pos(wp, HW);
    if (__syncthreads_or(any) && threadIdx.x == 0) flags[p] = 1;
}

// loss weight of the OS8 prediction (maggie/network/arch/maggie.py:271-281): w = [plane has ground truth] + [pixel is in the unknown band
// (1/255 <= v <= 254/255) of the ground truth or of the prediction]; ~12 elementwise passes over the planes in the torch formulation
__global__ __launch_bounds__(NT) void os8_weight_kernel(const float* __restrict__ gt, const float* __restrict__ a8, const int* __restrict__ flags,
                                                        long HW, int reweight, float* __restrict__ out, const int* __restrict__ pvalid) {
    const int p = blockIdx.y;
    const float valid = flags[p] ? 1.f : 0.f;
    const float pv = (pvalid && !pvalid[p]) ? 0.f : 1.f;          // a8 stands for `a8 * valid_masks` (arch/maggie.py:112-118)
    const long base = (long)p * HW;
    const float lo = 1.0f / 255.0f, hi = 254.0f / 255.0f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        float w = valid;
        if (reweight) {
            const float g = gt[base + i], a = a8[base + i] * pv;
            if ((g <= hi && g >= lo) || (a <= hi && a >= lo)) w += 1.f;
        }
        out[base + i] = w;
    }
}

__device__ __forceinline__ float sobel_mag(const float* __restrict__ a, const float* __restrict__ w, int y, int x, int H, int W, float& gx,
                                           float& gy, float as = 1.f) {
    float v[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int yy = clampi(y + i - 1, H), xx = clampi(x + j - 1, W);
            v[i][j] = a[yy * W + xx] * as * w[yy * W + xx];
        }
    gx = ((v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0])) * 0.125f;
    gy = ((v[2][0] - v[0][0]) + 2.f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2])) * 0.125f;
    return sqrtf(gx * gx + gy * gy + 1e-6f);
}

// d = p - t ; sums[0] += w|d| ; sums[1] += |sobel(p w) - sobel(t w)| ; sums[2] += w
__global__ __launch_bounds__(NT) void point_fwd_kernel(const P3 p, const float* __restrict__ t, const P3 w,
                                                       const int* __restrict__ flags, int H, int W, float* __restrict__ d,
                                                       float* __restrict__ sums, const int* __restrict__ pvalid, int Pper, float* __restrict__ slots) {
    __shared__ float sh[NT / 64];
    const int pl = blockIdx.y;
    float* slot = slots ? slots + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 : nullptr;     // row [l1, grad, w] of this workgroup
    if (!flags[pl]) { if (slot && threadIdx.x < 3) slot[threadIdx.x] = 0.f; return; }
    const int sc = pl / Pper, pq = pl - sc * Pper;
    // pvalid (0 / 1 per plane): `pred * valid_masks` of arch/maggie.py:112-118 -- the prediction of a plane without ground-truth transition
    // region counts as zero -- applied on the fly instead of by three multiplies over the (N, 10, H, W) planes (and three in backward)
    const float pv = (pvalid && !pvalid[pq]) ? 0.f : 1.f;
    const long off = (long)pl * H * W, offq = (long)pq * H * W;
    const float *pp = p.a[sc] + offq, *tp = t + offq, *wp = w.a[sc] + offq;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = blockIdx.x * NT + threadIdx.x; i < H * W; i += gridDim.x * NT) {
        int y = i / W, x = i - y * W;
        float dv = pp[i] * pv - tp[i], wv = wp[i];
        d[off + i] = dv;
        s0 += wv * fabsf(dv);
        s2 += wv;
        float gx, gy;
        float mp = sobel_mag(pp, wp, y, x, H, W, gx, gy, pv);
        float mt = sobel_mag(tp, wp, y, x, H, W, gx, gy);
        s1 += fabsf(mp - mt);
    }
    float* srep = sums + sc * LOSS_SUMS + ((blockIdx.y * gridDim.x + blockIdx.x) & (LOSS_REPLICAS - 1)) * LOSS_STRIDE;   // spread the same-address atomics
    block_add(s0, &srep[0], sh, slot);
    block_add(s1, &srep[1], sh, slot ? slot + 1 : nullptr);
    block_add(s2, &srep[2], sh, slot ? slot + 2 : nullptr);
}

// out[P, h/2, w/2] = (gauss5 * x)(2y, 2x), reflect padding
__global__ __launch_bounds__(NT) void pyr_down_kernel(const float* __restrict__ x, const int* __restrict__ flags, int h, int w,
                                                      float* __restrict__ out) {
    const int pl = blockIdx.y;
    if (!flags[pl]) return;
    const int hd = h >> 1, wd = w >> 1;
    const float* xp = x + (long)pl * h * w;
    for (int o = blockIdx.x * NT + threadIdx.x; o < hd * wd; o += gridDim.x * NT) {
        int y = o / wd, xx = o - y * wd;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int yy = refl(2 * y + i - 2, h);
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) r += c_g1[j] * xp[yy * w + refl(2 * xx + j - 2, w)];
            acc += c_g1[i] * r;
        }
        out[(long)pl * hd * wd + o] = acc;
    }
}

// L = x - 4 * gauss5 * zero_stuff(down);  sums[0] += |L| wl ; sums[1] += wl ; G = wl * sign(L);  wl = w0[(y << lvl), (x << lvl)]
__global__ __launch_bounds__(NT) void pyr_lap_fwd_kernel(const float* __restrict__ x, const float* __restrict__ down, const P3 w0,
                                                         int lvl, int H0, int W0, const int* __restrict__ flags, int h, int w,
                                                         float* __restrict__ G, float* __restrict__ sums, int Pper, float* __restrict__ slots) {
    __shared__ float sh[NT / 64];
    const int pl = blockIdx.y;
    float* slot = slots ? slots + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 : nullptr;     // row [lap, w] of this workgroup
    if (!flags[pl]) { if (slot && threadIdx.x < 2) slot[threadIdx.x] = 0.f; return; }
    const int sc = pl / Pper;
    const int hd = h >> 1, wd = w >> 1;
    const float* xp = x + (long)pl * h * w;
    const float* dp = down + (long)pl * hd * wd;
    const float* wp = w0.a[sc] + (long)(pl - sc * Pper) * H0 * W0;
    float s0 = 0.f, s1 = 0.f;
    // four pixels per trip (independent chains: the plane is walked by few workgroups -- their count is bounded by the two same-address
    // atomics each ends with -- so a thread's ~20 pixels were ~20 exposed memory round trips)
    const int stride = gridDim.x * NT;
    for (int o4 = blockIdx.x * NT + threadIdx.x; o4 < h * w; o4 += 4 * stride)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int o = o4 + u * stride;
        if (o >= h * w) break;
        int y = o / w, xx = o - y * w;
        float up = 0.f;
        if (y >= 2 && y < h - 2 && xx >= 2 && xx < w - 2) {
            // interior: no reflection; only the taps that hit an even (stuffed) position contribute -- i = y mod 2 (+2, +4), same for j.
            // Same terms in the same order as the general walk below (which spends most of its time skipping the other 16-21 taps).
            const float g1[5] = {1.f / 16.f, 4.f / 16.f, 6.f / 16.f, 4.f / 16.f, 1.f / 16.f};
            const int i0 = y & 1, j0 = xx & 1;
            const float* row = dp + ((y + i0 - 2) >> 1) * wd + ((xx + j0 - 2) >> 1);
            if (j0 == 0) {
                for (int i = i0; i < 5; i += 2, row += wd) up += g1[i] * (g1[0] * row[0] + g1[2] * row[1] + g1[4] * row[2]);
            } else {
                for (int i = i0; i < 5; i += 2, row += wd) up += g1[i] * (g1[1] * row[0] + g1[3] * row[1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int yy = refl(y + i - 2, h);
                if (yy & 1) continue;
                float r = 0.f;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    int xs = refl(xx + j - 2, w);
                    if (!(xs & 1)) r += c_g1[j] * dp[(yy >> 1) * wd + (xs >> 1)];
                }
                up += c_g1[i] * r;
            }
        }
        float L = xp[o] - 4.f * up;
        float wl = wp[(long)(y << lvl) * W0 + (xx << lvl)];
        s0 += fabsf(L) * wl;
        s1 += wl;
        G[(long)pl * h * w + o] = L > 0.f ? wl : (L < 0.f ? -wl : 0.f);
    }
    float* srep = sums + sc * LOSS_SUMS + ((blockIdx.y * gridDim.x + blockIdx.x) & (LOSS_REPLICAS - 1)) * LOSS_STRIDE;
    block_add(s0, &srep[0], sh, slot);
    block_add(s1, &srep[1], sh, slot ? slot + 1 : nullptr);
}

// r[P, h/2, w/2] = add - coef * U^T(q),  U = 4 * gauss5 * zero_stuff (reflect); q: [P, h, w]
__global__ __launch_bounds__(NT) void pyr_upT_kernel(const float* __restrict__ q, const float* __restrict__ coef, const float* __restrict__ add,
                                                     const int* __restrict__ flags, int h, int w, float* __restrict__ r, int Pper) {
    const int pl = blockIdx.y;
    if (!flags[pl]) return;
    const int hd = h >> 1, wd = w >> 1;
    const float* qp = q + (long)pl * h * w;
    const float c = coef[(pl / Pper) * 5];               // coefficient of this plane's scale (coef5 rows, see loss_coef_kernel)
    for (int o = blockIdx.x * NT + threadIdx.x; o < hd * wd; o += gridDim.x * NT) {
        int a = o / wd, b = o - a * wd;
        float acc = 0.f;
        // stuffed index 2a and its reflect pre-images: -2a (a == 1), 2(h-1) - 2a (== h when a == h/2 - 1)
#pragma unroll
        for (int sa = 0; sa < 3; ++sa) {
            int ma = sa == 0 ? 2 * a : (sa == 1 ? -2 * a : 2 * (h - 1) - 2 * a);
            if (sa == 1 && a != 1) continue;
            if (sa == 2 && ma != h) continue;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int y = ma - i + 2;
                if (y < 0 || y >= h) continue;
                float rowacc = 0.f;
#pragma unroll
                for (int sb = 0; sb < 3; ++sb) {
                    int mb = sb == 0 ? 2 * b : (sb == 1 ? -2 * b : 2 * (w - 1) - 2 * b);
                    if (sb == 1 && b != 1) continue;
                    if (sb == 2 && mb != w) continue;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        int x = mb - j + 2;
                        if (x < 0 || x >= w) continue;
                        rowacc += c_g1[j] * qp[y * w + x];
                    }
                }
                acc += c_g1[i] * rowacc;
            }
        }
        float base = add ? add[(long)pl * hd * wd + o] : 0.f;
        r[(long)pl * hd * wd + o] = base - 4.f * c * acc;
    }
}

// dd[P, h, w] = coef * q + D^T(r),  D = decimate2(gauss5 * . ) (reflect); r: [P, h/2, w/2]
__global__ __launch_bounds__(NT) void pyr_downT_kernel(const float* __restrict__ r, const float* __restrict__ q, const float* __restrict__ coef,
                                                       const int* __restrict__ flags, int h, int w, float* __restrict__ dd, int Pper) {
    const int pl = blockIdx.y;
    if (!flags[pl]) return;
    const int hd = h >> 1, wd = w >> 1;
    const float* rp = r + (long)pl * hd * wd;
    const float c = coef[(pl / Pper) * 5];
    for (int o = blockIdx.x * NT + threadIdx.x; o < h * w; o += gridDim.x * NT) {
        int Y = o / w, X = o - Y * w;
        float acc = 0.f;
#pragma unroll
        for (int sa = 0; sa < 3; ++sa) {
            int my = sa == 0 ? Y : (sa == 1 ? -Y : 2 * (h - 1) - Y);
            if (sa == 1 && !(Y == 1 || Y == 2)) continue;
            if (sa == 2 && !(Y == h - 2 || Y == h - 3)) continue;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int ty = my - i + 2;
                if (ty < 0 || (ty & 1)) continue;
                int y = ty >> 1;
                if (y >= hd) continue;
                float rowacc = 0.f;
#pragma unroll
                for (int sb = 0; sb < 3; ++sb) {
                    int mx = sb == 0 ? X : (sb == 1 ? -X : 2 * (w - 1) - X);
                    if (sb == 1 && !(X == 1 || X == 2)) continue;
                    if (sb == 2 && !(X == w - 2 || X == w - 3)) continue;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        int tx = mx - j + 2;
                        if (tx < 0 || (tx & 1)) continue;
                        int x = tx >> 1;
                        if (x >= wd) continue;
                        rowacc += c_g1[j] * rp[y * wd + x];
                    }
                }
                acc += c_g1[i] * rowacc;
            }
        }
        dd[(long)pl * h * w + o] = c * q[(long)pl * h * w + o] + acc;
    }
}

// ---- round 5, second pass: the same four stencils with their loads in front of the arithmetic ---------------------------------------------------
// The walks above compile to one `global_load` + `s_waitcnt vmcnt(0)` per tap (hipcc -S: pyr_lap_fwd 124 loads / 112 full waits, pyr_downT 144 / 144,
// pyr_upT 57 / 57, point_bwd 22 / 36): every tap sits behind a run-time condition, so a pixel is 6-25 dependent memory round trips and the kernels ran
// at 12-18 % of the HBM rate. Away from the border the tap set is a function of the pixel's parity alone: a 2 x 2 output quad (2a + dy, 2b + dx) of
// pyr_lap_fwd / pyr_downT reads ONE 3 x 3 window of the half-resolution plane, an output of pyr_upT one 5 x 5 window, a pixel of point_bwd two
// 3 x 3 windows -- loaded unconditionally as a batch, then combined with the terms of the walks above in the same order (csrc/loss_stencils.h: the
// per-pixel values keep their bits; the per-workgroup sums of pyr_lap_fwd add the same values in a different order). Border cells take the general
// walks, and ring_map() enumerates them behind the inner cells so that whole waves take one path. MG_LOSS_BATCHED=0 selects the first forms.
__global__ __launch_bounds__(NT) void pyr_lap_fwd_quad_kernel(const float* __restrict__ x, const float* __restrict__ down, const P3 w0,
                                                              int lvl, int H0, int W0, const int* __restrict__ flags, int h, int w,
                                                              float* __restrict__ G, float* __restrict__ sums, int Pper, float* __restrict__ slots) {
    __shared__ float sh[NT / 64];
    const int pl = blockIdx.y;
    float* slot = slots ? slots + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 : nullptr;
    if (!flags[pl]) { if (slot && threadIdx.x < 2) slot[threadIdx.x] = 0.f; return; }
    const int sc = pl / Pper;
    const int hd = h >> 1, wd = w >> 1;
    const float* __restrict__ xp = x + (long)pl * h * w;
    const float* __restrict__ dp = down + (long)pl * hd * wd;
    const float* __restrict__ wp = w0.a[sc] + (long)(pl - sc * Pper) * H0 * W0;
    float* __restrict__ Gp = G + (long)pl * h * w;
    float s0 = 0.f, s1 = 0.f;
    // index space: the inner quads (one thread each, four pixels), then the border ring PIXEL by pixel (four threads per border quad: a border pixel
    // is a chain of dependent steps, and four of them in one thread were the longest path of the launch)
    const int hi = hd - 2, wi = wd - 2;
    const int n_int = (hi > 0 && wi > 0) ? hi * wi : 0;
    const int total = n_int + 4 * (hd * wd - n_int);
    for (int q = blockIdx.x * NT + threadIdx.x; q < total; q += gridDim.x * NT) {
        if (q >= n_int) {
            const int bt = q - n_int;
            int a, b;
            ring_map(n_int + (bt >> 2), hd, wd, 1, 1, 1, 1, a, b);
            const int y = 2 * a + ((bt >> 1) & 1), xx = 2 * b + (bt & 1);
            const float xv1 = xp[(long)y * w + xx];
            const float wl1 = wp[(long)(y << lvl) * W0 + (xx << lvl)];
            const float L = xv1 - 4.f * lap_up_general(dp, y, xx, h, w, wd);
            s0 += fabsf(L) * wl1;
            s1 += wl1;
            Gp[(long)y * w + xx] = L > 0.f ? wl1 : (L < 0.f ? -wl1 : 0.f);
            continue;
        }
        int a, b;
        ring_map(q, hd, wd, 1, 1, 1, 1, a, b);
        const int y0 = 2 * a, x0 = 2 * b;
        const float* __restrict__ xr0 = xp + (long)y0 * w + x0;
        const float xv[4] = {xr0[0], xr0[1], xr0[w], xr0[w + 1]};
        const long wr0 = (long)(y0 << lvl) * W0, wr1 = (long)((y0 + 1) << lvl) * W0;
        const int wc0 = x0 << lvl, wc1 = (x0 + 1) << lvl;
        const float wl[4] = {wp[wr0 + wc0], wp[wr0 + wc1], wp[wr1 + wc0], wp[wr1 + wc1]};
        float up[4];
        lap_up_inner(dp, a, b, wd, up);
        float gq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float L = xv[u] - 4.f * up[u];
            s0 += fabsf(L) * wl[u];
            s1 += wl[u];
            gq[u] = L > 0.f ? wl[u] : (L < 0.f ? -wl[u] : 0.f);
        }
        float* __restrict__ gr0 = Gp + (long)y0 * w + x0;
        gr0[0] = gq[0]; gr0[1] = gq[1]; gr0[w] = gq[2]; gr0[w + 1] = gq[3];
    }
    float* srep = sums + sc * LOSS_SUMS + ((blockIdx.y * gridDim.x + blockIdx.x) & (LOSS_REPLICAS - 1)) * LOSS_STRIDE;
    block_add(s0, &srep[0], sh, slot);
    block_add(s1, &srep[1], sh, slot ? slot + 1 : nullptr);
}

__global__ __launch_bounds__(NT) void pyr_upT_batched_kernel(const float* __restrict__ q, const float* __restrict__ coef, const float* __restrict__ add,
                                                             const int* __restrict__ flags, int h, int w, float* __restrict__ r, int Pper) {
    const int pl = blockIdx.y;
    if (!flags[pl]) return;
    const int hd = h >> 1, wd = w >> 1;
    const float* __restrict__ qp = q + (long)pl * h * w;
    const float c = coef[(pl / Pper) * 5];
    for (int o = blockIdx.x * NT + threadIdx.x; o < hd * wd; o += gridDim.x * NT) {
        int a, b;
        const bool inner = ring_map(o, hd, wd, 2, 1, 2, 1, a, b);
        const long oo = (long)pl * hd * wd + a * wd + b;
        const float base = add ? add[oo] : 0.f;
        const float acc = inner ? upT_inner(qp, a, b, w) : upT_general(qp, a, b, h, w);
        r[oo] = base - 4.f * c * acc;
    }
}

__global__ __launch_bounds__(NT) void pyr_downT_quad_kernel(const float* __restrict__ r, const float* __restrict__ q, const float* __restrict__ coef,
                                                            const int* __restrict__ flags, int h, int w, float* __restrict__ dd, int Pper) {
    const int pl = blockIdx.y;
    if (!flags[pl]) return;
    const int hd = h >> 1, wd = w >> 1;
    const float* __restrict__ rp = r + (long)pl * hd * wd;
    const float* __restrict__ qp = q + (long)pl * h * w;
    float* __restrict__ op = dd + (long)pl * h * w;
    const float c = coef[(pl / Pper) * 5];
    const int hi = hd - 4, wi = wd - 4;                          // inner quads first, then the border ring pixel by pixel (see pyr_lap_fwd_quad_kernel)
    const int n_int = (hi > 0 && wi > 0) ? hi * wi : 0;
    const int total = n_int + 4 * (hd * wd - n_int);
    for (int t = blockIdx.x * NT + threadIdx.x; t < total; t += gridDim.x * NT) {
        if (t >= n_int) {
            const int bt = t - n_int;
            int a, b;
            ring_map(n_int + (bt >> 2), hd, wd, 2, 2, 2, 2, a, b);
            const int Y = 2 * a + ((bt >> 1) & 1), X = 2 * b + (bt & 1);
            const float qv1 = qp[(long)Y * w + X];
            op[(long)Y * w + X] = c * qv1 + downT_general(rp, Y, X, h, w, hd, wd);
            continue;
        }
        int a, b;
        ring_map(t, hd, wd, 2, 2, 2, 2, a, b);
        const int y0 = 2 * a, x0 = 2 * b;
        const float* __restrict__ qr0 = qp + (long)y0 * w + x0;
        const float qv[4] = {qr0[0], qr0[1], qr0[w], qr0[w + 1]};
        float acc[4];
        downT_inner(rp, a, b, wd, acc);
        float* __restrict__ o0 = op + (long)y0 * w + x0;
        o0[0] = c * qv[0] + acc[0]; o0[1] = c * qv[1] + acc[1]; o0[w] = c * qv[2] + acc[2]; o0[w + 1] = c * qv[3] + acc[3];
    }
}

// Sobel backward pass 1: A = s * gx / mag, B = s * gy / mag with s = sign(mag_p - mag_t)   (coef applied in pass 2)
__global__ __launch_bounds__(NT) void sobel_bwd1_kernel(const P3 p, const float* __restrict__ t, const P3 w,
                                                        const int* __restrict__ flags, int H, int W, float* __restrict__ A, float* __restrict__ B,
                                                        const int* __restrict__ pvalid, int Pper) {
    const int pl = blockIdx.y;
    if (!flags[pl]) return;
    const int sc = pl / Pper, pq = pl - sc * Pper;
    if (pvalid && !pvalid[pq]) return;                  // masked prediction: point_bwd writes a zero gradient without reading A / B
    const long off = (long)pl * H * W, offq = (long)pq * H * W;
    const float *pp = p.a[sc] + offq, *tp = t + offq, *wp = w.a[sc] + offq;
    for (int i = blockIdx.x * NT + threadIdx.x; i < H * W; i += gridDim.x * NT) {
        int y = i / W, x = i - y * W;
        float gx, gy, tx, ty;
        float mp = sobel_mag(pp, wp, y, x, H, W, gx, gy);
        float mt = sobel_mag(tp, wp, y, x, H, W, tx, ty);
        float s = mp > mt ? 1.f : (mp < mt ? -1.f : 0.f);
        A[off + i] = s * gx / mp;
        B[off + i] = s * gy / mp;
    }
}

// dpred = coef_rec * w * sign(p - t) + dd_lap + coef_grad * w * SobelAdjoint(A, B)   (replicate padding adjoint)
__global__ __launch_bounds__(NT) void point_bwd_kernel(const P3 p, const float* __restrict__ t, const P3 w,
                                                       const int* __restrict__ flags, int H, int W, const float* __restrict__ coef_rec,
                                                       const float* __restrict__ coef_grad, const float* __restrict__ dd,
                                                       const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ dp,
                                                       const int* __restrict__ pvalid, int Pper) {
    const int pl = blockIdx.y;
    const int sc = pl / Pper, pq = pl - sc * Pper;
    const long off = (long)pl * H * W, offq = (long)pq * H * W;
    if (!flags[pl] || (pvalid && !pvalid[pq])) {        // a plane without weight, or a masked prediction (d(p * 0)/dp = 0): zero gradient, written here
        for (int idx = blockIdx.x * NT + threadIdx.x; idx < H * W; idx += gridDim.x * NT) dp[off + idx] = 0.f;
        return;
    }
    const float cr = coef_rec[sc * 5], cg = coef_grad[sc * 5];
    const float* __restrict__ pp = p.a[sc] + offq;
    const float* __restrict__ tp = t + offq;
    const float* __restrict__ wp = w.a[sc] + offq;
    const float kx[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < H * W; idx += gridDim.x * NT) {
        int Y = idx / W, X = idx - Y * W;
        float dv = pp[idx] - tp[idx];
        float wv = wp[idx];
        float g = cr * wv * (dv > 0.f ? 1.f : (dv < 0.f ? -1.f : 0.f));
        if (dd) g += dd[off + idx];
        float acc = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            int y = Y + dy;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                int x = X + dx;
                if (x < 0 || x >= W) continue;
                float a = A[off + y * W + x], b = B[off + y * W + x];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (clampi(y + i - 1, H) != Y) continue;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (clampi(x + j - 1, W) != X) continue;
                        acc += kx[i][j] * a + kx[j][i] * b;
                    }
                }
            }
        }
        dp[off + idx] = g + cg * wv * acc * 0.125f;
    }
}

// point_bwd with the two 3 x 3 windows of A and B loaded as one batch for the pixels whose neighbourhood does not touch the border
// (sobel_adj_inner); border pixels keep the walk (sobel_adj_general), enumerated last (ring_map)
__global__ __launch_bounds__(NT) void point_bwd_batched_kernel(const P3 p, const float* __restrict__ t, const P3 w,
                                                               const int* __restrict__ flags, int H, int W, const float* __restrict__ coef_rec,
                                                               const float* __restrict__ coef_grad, const float* __restrict__ dd,
                                                               const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ dp,
                                                               const int* __restrict__ pvalid, int Pper) {
    const int pl = blockIdx.y;
    const int sc = pl / Pper, pq = pl - sc * Pper;
    const long off = (long)pl * H * W, offq = (long)pq * H * W;
    if (!flags[pl] || (pvalid && !pvalid[pq])) {
        for (int idx = blockIdx.x * NT + threadIdx.x; idx < H * W; idx += gridDim.x * NT) dp[off + idx] = 0.f;
        return;
    }
    const float cr = coef_rec[sc * 5], cg = coef_grad[sc * 5];
    const float* __restrict__ pp = p.a[sc] + offq;
    const float* __restrict__ tp = t + offq;
    const float* __restrict__ wp = w.a[sc] + offq;
    const float* __restrict__ Ap = A + off;
    const float* __restrict__ Bp = B + off;
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < H * W; idx += gridDim.x * NT) {
        int Y, X;
        const bool inner = ring_map(idx, H, W, 1, 1, 1, 1, Y, X);
        const int px = Y * W + X;
        const float dv = pp[px] - tp[px];
        const float wv = wp[px];
        const float ddv = dd ? dd[off + px] : 0.f;
        const float acc = inner ? sobel_adj_inner(Ap, Bp, Y, X, W) : sobel_adj_general(Ap, Bp, Y, X, H, W);
        float g = cr * wv * (dv > 0.f ? 1.f : (dv < 0.f ? -1.f : 0.f));
        if (dd) g += ddv;
        dp[off + px] = g + cg * wv * acc * 0.125f;
    }
}

inline dim3 grid2(long per_plane, int P, long max_blocks = 1024) {
    long b = (per_plane + NT - 1) / NT;
    if (b > max_blocks) b = max_blocks;
    if (b < 1) b = 1;
    return dim3((unsigned)b, (unsigned)P);
}
// kernels that end with one atomicAdd per sum per block: every block hits the same 2-3 addresses, so keep the block count low
// (8 planes x 1024 blocks x 3 same-address atomics cost more than the stencil itself)
static const long REDUCING_BLOCKS_PER_PLANE = [] { const char* e = getenv("MG_LOSS_BLOCKS"); return e ? atol(e) : 256l; }();

// MG_LOSS_BATCHED (default 1): the batched-load forms of the pyramid / Sobel-adjoint stencils (even plane sizes; odd ones keep the first forms)
static const int LOSS_BATCHED = [] { const char* e = getenv("MG_LOSS_BATCHED"); return e ? atoi(e) : 1; }();

static void launch_pyr_lap_fwd(dim3 g, hipStream_t st, const float* x, const float* down, P3 ww, int lvl, int H0, int W0, const int* flags, int h, int w,
                               float* G, float* sums, int Pper, float* slots) {
    if (LOSS_BATCHED && !(h & 1) && !(w & 1))
        hipLaunchKernelGGL(pyr_lap_fwd_quad_kernel, g, dim3(NT), 0, st, x, down, ww, lvl, H0, W0, flags, h, w, G, sums, Pper, slots);
    else
        hipLaunchKernelGGL(pyr_lap_fwd_kernel, g, dim3(NT), 0, st, x, down, ww, lvl, H0, W0, flags, h, w, G, sums, Pper, slots);
}
static void launch_pyr_upT(int SP, hipStream_t st, const float* q, const float* coef, const float* add, const int* flags, int h, int w, float* r, int Pper) {
    const dim3 g = grid2((long)(h / 2) * (w / 2), SP);
    if (LOSS_BATCHED && !(h & 1) && !(w & 1))
        hipLaunchKernelGGL(pyr_upT_batched_kernel, g, dim3(NT), 0, st, q, coef, add, flags, h, w, r, Pper);
    else
        hipLaunchKernelGGL(pyr_upT_kernel, g, dim3(NT), 0, st, q, coef, add, flags, h, w, r, Pper);
}
static void launch_pyr_downT(int SP, hipStream_t st, const float* r, const float* q, const float* coef, const int* flags, int h, int w, float* dd, int Pper) {
    if (LOSS_BATCHED && !(h & 1) && !(w & 1))
        hipLaunchKernelGGL(pyr_downT_quad_kernel, grid2((long)(h / 2) * (w / 2), SP), dim3(NT), 0, st, r, q, coef, flags, h, w, dd, Pper);
    else
        hipLaunchKernelGGL(pyr_downT_kernel, grid2((long)h * w, SP), dim3(NT), 0, st, r, q, coef, flags, h, w, dd, Pper);
}
static void launch_point_bwd(int SP, hipStream_t st, P3 pp, const float* t, P3 ww, const int* flags, int H, int W, const float* coef_rec,
                             const float* coef_grad, const float* dd, const float* A, const float* B, float* dp, const int* pvalid, int Pper) {
    if (LOSS_BATCHED)
        hipLaunchKernelGGL(point_bwd_batched_kernel, grid2((long)H * W, SP), dim3(NT), 0, st, pp, t, ww, flags, H, W, coef_rec, coef_grad, dd, A, B, dp, pvalid, Pper);
    else
        hipLaunchKernelGGL(point_bwd_kernel, grid2((long)H * W, SP), dim3(NT), 0, st, pp, t, ww, flags, H, W, coef_rec, coef_grad, dd, A, B, dp, pvalid, Pper);
}

// sums = [l1, grad, w, lap0, w0, lap1, w1, lap2, w2] -> (rec, lap, grad) exactly as arch/maggie.py:237-262 / loss.py:67-191
// normalise them (eps 1e-8 for the L1 term, 1e-6 for the others; LapLoss is the 3-fold channel sum)
__device__ __forceinline__ void loss_sum_replicas(const float* __restrict__ rep, float* s) {      // 64 threads; result in s[0..16) (shared)
    if (threadIdx.x < LOSS_STRIDE) {
        float a = 0.f;
        for (int r = 0; r < LOSS_REPLICAS; ++r) a += rep[r * LOSS_STRIDE + threadIdx.x];
        s[threadIdx.x] = a;
    }
    __syncthreads();
}
__global__ void loss_finish_kernel(const float* __restrict__ rep, float* __restrict__ out) {      // one workgroup per scale
    __shared__ float sums[LOSS_STRIDE];
    rep += blockIdx.x * LOSS_SUMS; out += blockIdx.x * 3;
    loss_sum_replicas(rep, sums);
    if (threadIdx.x == 0) {
        out[0] = sums[0] / (sums[2] + 1e-8f);
        out[1] = 3.0f * (sums[3] / (sums[4] + 1e-6f) + sums[5] / (sums[6] + 1e-6f) + sums[7] / (sums[8] + 1e-6f));
        out[2] = sums[1] / (sums[2] + 1e-6f);
    }
}
// upstream gradient (d rec, d lap, d grad) -> the five per-term coefficients of the backward kernels
__global__ void loss_coef_kernel(const float* __restrict__ g, const float* __restrict__ rep, float* __restrict__ coef) {   // one workgroup per scale
    __shared__ float sums[LOSS_STRIDE];
    rep += blockIdx.x * LOSS_SUMS; g += blockIdx.x * 3; coef += blockIdx.x * 5;
    loss_sum_replicas(rep, sums);
    if (threadIdx.x == 0) {
        coef[0] = g[0] / (sums[2] + 1e-8f);
        coef[1] = g[2] / (sums[2] + 1e-6f);
        coef[2] = 3.0f * g[1] / (sums[4] + 1e-6f);
        coef[3] = 3.0f * g[1] / (sums[6] + 1e-6f);
        coef[4] = 3.0f * g[1] / (sums[8] + 1e-6f);
    }
}

}  // namespace

// deterministic mode: the slot rows of a reducing launch ([S][P * blocks per plane][ncol], scale-major like the planes) are added in order into
// replica 0 of each scale's accumulators, columns [col, col + ncol)
static int loss_slots(dim3 g, int ncol, float** slots) {
    *slots = nullptr;
    if (!mg_det_on) return 0;
    *slots = mg_det_scratch((long)g.x * g.y * ncol);
    return *slots ? 0 : MG_DET_NO_SCRATCH;
}
static int loss_slot_reduce(const float* slots, dim3 g, int S, int ncol, float* sums, int col, hipStream_t st) {
    if (!slots) return 0;
    mg_det_seg sg{sums + col, ncol, (long)LOSS_SUMS};
    return mg_det_reduce(slots, (int)(g.x * (g.y / S)), S, ncol, 0, &sg, 1, st);
}

extern "C" int mg_loss_finish(const float* sums, float* out3, void* stream) {
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, out3);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_loss_coef(const float* g3, const float* sums, float* coef5, void* stream) {
    hipLaunchKernelGGL(loss_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g3, sums, coef5);
    MG_CHECK_LAUNCH();
    return 0;
}

// as_float != 0: the flags are written as fp32 0.0 / 1.0 (a per-plane scale that a kernel multiplies with, `valid_masks` of
// resnet_inst_matt_spconv.py:320) instead of int32 0 / 1
extern "C" int mg_plane_flags_ex(const float* w, int P, int HW, void* flags, int as_float, void* stream) {
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = mg_zero_words(flags, (long)P, st);
    if (e != hipSuccess) return (int)e;
    dim3 g = grid2(HW, P);
    if (g.x > 64) g.x = 64;
    hipLaunchKernelGGL(plane_flags_kernel, g, dim3(NT), 0, st, w, HW, (int*)flags, as_float ? 0x3f800000 : 1);
    MG_CHECK_LAUNCH();
    return 0;
}
extern "C" int mg_plane_flags(const float* w, int P, int HW, int32_t* flags, void* stream) { return mg_plane_flags_ex(w, P, HW, flags, 0, stream); }

extern "C" int mg_os8_weight_ex(const float* gt, const float* a8, int P, long HW, int reweight, int32_t* flags_scratch, float* out, const int32_t* pvalid,
                                void* stream);
extern "C" int mg_os8_weight(const float* gt, const float* a8, int P, long HW, int reweight, int32_t* flags_scratch, float* out, void* stream) {
    return mg_os8_weight_ex(gt, a8, P, HW, reweight, flags_scratch, out, nullptr, stream);
}
extern "C" int mg_os8_weight_ex(const float* gt, const float* a8, int P, long HW, int reweight, int32_t* flags_scratch, float* out, const int32_t* pvalid,
                                void* stream) {
    if (P <= 0 || HW <= 0) return 0;
    int rc = mg_plane_flags(gt, P, (int)HW, flags_scratch, stream);
    if (rc) return rc;
    dim3 g = grid2(HW, P);
    hipLaunchKernelGGL(os8_weight_kernel, g, dim3(NT), 0, (hipStream_t)stream, gt, a8, (const int*)flags_scratch, HW, reweight, out, pvalid);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_loss_point_fwd(const float* p, const float* t, const float* w, const int32_t* flags, int P, int H, int W, float* d,
                                 float* sums, const int32_t* pvalid, void* stream) {
    if (P <= 0) return 0;
    const dim3 g = grid2((long)H * W, P, REDUCING_BLOCKS_PER_PLANE);
    float* slots; int rc = loss_slots(g, 3, &slots); if (rc) return rc;
    hipLaunchKernelGGL(point_fwd_kernel, g, dim3(NT), 0, (hipStream_t)stream, P3{{p, nullptr, nullptr}}, t, P3{{w, nullptr, nullptr}}, flags, H, W, d, sums, pvalid, P, slots);
    MG_CHECK_LAUNCH();
    return loss_slot_reduce(slots, g, 1, 3, sums, 0, (hipStream_t)stream);
}

extern "C" int mg_pyr_down(const float* x, const int32_t* flags, int P, int h, int w, float* out, void* stream) {
    if (P <= 0) return 0;
    if ((h & 1) || (w & 1) || h < 4 || w < 4) return -2;
    hipLaunchKernelGGL(pyr_down_kernel, grid2((long)(h / 2) * (w / 2), P), dim3(NT), 0, (hipStream_t)stream, x, flags, h, w, out);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_pyr_lap_fwd(const float* x, const float* down, const float* w0, int lvl, int H0, int W0, const int32_t* flags, int P, int h,
                              int w, float* G, float* sums, void* stream) {
    if (P <= 0) return 0;
    const dim3 g = grid2((long)h * w, P, REDUCING_BLOCKS_PER_PLANE);
    float* slots; int rc = loss_slots(g, 2, &slots); if (rc) return rc;
    launch_pyr_lap_fwd(g, (hipStream_t)stream, x, down, P3{{w0, nullptr, nullptr}}, lvl, H0, W0, flags, h, w, G, sums, P, slots);
    MG_CHECK_LAUNCH();
    return loss_slot_reduce(slots, g, 1, 2, sums, 0, (hipStream_t)stream);
}

extern "C" int mg_pyr_upT(const float* q, const float* coef, const float* add, const int32_t* flags, int P, int h, int w, float* r, void* stream) {
    if (P <= 0) return 0;
    launch_pyr_upT(P, (hipStream_t)stream, q, coef, add, flags, h, w, r, P);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_pyr_downT(const float* r, const float* q, const float* coef, const int32_t* flags, int P, int h, int w, float* dd, void* stream) {
    if (P <= 0) return 0;
    launch_pyr_downT(P, (hipStream_t)stream, r, q, coef, flags, h, w, dd, P);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_loss_point_bwd(const float* p, const float* t, const float* w, const int32_t* flags, int P, int H, int W, const float* coef_rec,
                                 const float* coef_grad, const float* dd, float* A, float* B, float* dp, const int32_t* pvalid, void* stream) {
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const P3 pp{{p, nullptr, nullptr}}, ww{{w, nullptr, nullptr}};
    hipLaunchKernelGGL(sobel_bwd1_kernel, grid2((long)H * W, P), dim3(NT), 0, st, pp, t, ww, flags, H, W, A, B, pvalid, P);
    launch_point_bwd(P, st, pp, t, ww, flags, H, W, coef_rec, coef_grad, dd, A, B, dp, pvalid, P);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---- the whole loss pipeline of S (<= 3) output scales in one call each way (round 3). Scale s reads prediction p[s] and weight w[s] ([P][H][W]
// each), all share the target t; every scratch buffer holds S * P planes (scale-major). Forward: 9 launches whatever S is (was 9 per scale).
extern "C" int mg_matting_losses_fwd(const float* const* p, const float* t, const float* const* w, const int32_t* pvalid, int S, int P, int H, int W,
                                     int32_t* flags, float* d, float* down0, float* down1, float* down2, float* G0, float* G1, float* G2, float* sums,
                                     float* out, void* stream) {
    if (S < 1 || S > 3 || P <= 0) return -2;
    if ((H & 7) || (W & 7) || H < 16 || W < 16) return -2;
    hipStream_t st = (hipStream_t)stream;
    P3 pp{{nullptr, nullptr, nullptr}}, ww{{nullptr, nullptr, nullptr}};
    for (int i = 0; i < S; ++i) { pp.a[i] = p[i]; ww.a[i] = w[i]; }
    const int SP = S * P;
    hipError_t e = mg_zero_words(flags, (long)SP, st); if (e != hipSuccess) return (int)e;
    e = mg_zero_words(sums, (long)S * LOSS_SUMS, st); if (e != hipSuccess) return (int)e;
    { dim3 g = grid2((long)H * W, SP); if (g.x > 64) g.x = 64; hipLaunchKernelGGL(plane_flags3_kernel, g, dim3(NT), 0, st, ww, P, H * W, flags); }
    {
        const dim3 g = grid2((long)H * W, SP, REDUCING_BLOCKS_PER_PLANE);
        float* slots; int rc = loss_slots(g, 3, &slots); if (rc) return rc;
        hipLaunchKernelGGL(point_fwd_kernel, g, dim3(NT), 0, st, pp, t, ww, flags, H, W, d, sums, pvalid, P, slots);
        rc = loss_slot_reduce(slots, g, S, 3, sums, 0, st); if (rc) return rc;
    }
    const float* x = d; float* downs[3] = {down0, down1, down2}; float* Gs[3] = {G0, G1, G2};
    int h = H, wd = W;
    for (int lvl = 0; lvl < 3; ++lvl) {
        hipLaunchKernelGGL(pyr_down_kernel, grid2((long)(h / 2) * (wd / 2), SP), dim3(NT), 0, st, x, flags, h, wd, downs[lvl]);
        const dim3 g = grid2((long)h * wd, SP, REDUCING_BLOCKS_PER_PLANE);
        float* slots; int rc = loss_slots(g, 2, &slots); if (rc) return rc;
        launch_pyr_lap_fwd(g, st, x, (const float*)downs[lvl], ww, lvl, H, W, flags, h, wd, Gs[lvl], sums + 3 + 2 * lvl, P, slots);
        rc = loss_slot_reduce(slots, g, S, 2, sums, 3 + 2 * lvl, st); if (rc) return rc;
        x = downs[lvl]; h >>= 1; wd >>= 1;
    }
    hipLaunchKernelGGL(loss_finish_kernel, dim3(S), dim3(64), 0, st, (const float*)sums, out);
    MG_CHECK_LAUNCH();
    return 0;
}

/* backward of the above: g [S][3] upstream gradients of (rec, lap, grad) -> dp [S*P][H][W]. Scratch: coef [S][5], r2 (H/8), dd2 (H/4), r1 (H/4), dd1
 * (H/2), r0 (H/2), dd0 (H), A, B (H): S*P planes each. */
extern "C" int mg_matting_losses_bwd(const float* g, const float* sums, const float* const* p, const float* t, const float* const* w,
                                     const int32_t* pvalid, const int32_t* flags, int S, int P, int H, int W, const float* G0, const float* G1,
                                     const float* G2, float* coef, float* r2, float* dd2, float* r1, float* dd1, float* r0, float* dd0, float* A, float* B,
                                     float* dp, void* stream) {
    if (S < 1 || S > 3 || P <= 0) return -2;
    hipStream_t st = (hipStream_t)stream;
    P3 pp{{nullptr, nullptr, nullptr}}, ww{{nullptr, nullptr, nullptr}};
    for (int i = 0; i < S; ++i) { pp.a[i] = p[i]; ww.a[i] = w[i]; }
    const int SP = S * P;
    hipLaunchKernelGGL(loss_coef_kernel, dim3(S), dim3(64), 0, st, g, sums, coef);
    launch_pyr_upT(SP, st, G2, (const float*)(coef + 4), (const float*)nullptr, flags, H / 4, W / 4, r2, P);
    launch_pyr_downT(SP, st, (const float*)r2, G2, (const float*)(coef + 4), flags, H / 4, W / 4, dd2, P);
    launch_pyr_upT(SP, st, G1, (const float*)(coef + 3), (const float*)dd2, flags, H / 2, W / 2, r1, P);
    launch_pyr_downT(SP, st, (const float*)r1, G1, (const float*)(coef + 3), flags, H / 2, W / 2, dd1, P);
    launch_pyr_upT(SP, st, G0, (const float*)(coef + 2), (const float*)dd1, flags, H, W, r0, P);
    launch_pyr_downT(SP, st, (const float*)r0, G0, (const float*)(coef + 2), flags, H, W, dd0, P);
    hipLaunchKernelGGL(sobel_bwd1_kernel, grid2((long)H * W, SP), dim3(NT), 0, st, pp, t, ww, flags, H, W, A, B, pvalid, P);
    launch_point_bwd(SP, st, pp, t, ww, flags, H, W, (const float*)coef, (const float*)(coef + 1), (const float*)dd0, (const float*)A, (const float*)B, dp, pvalid, P);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention guidance loss of the instance matte decoder (maggie/network/module/instance_matte_decoder.py: compute_atten_loss):
//   loss = scale * sum_rows [ (sum_l gm[row, l] != 0) - sum_l gm[row, l] * att[row, l] ],  d att = -scale * gout * gm.
// One workgroup per row writes its term, a second tiny kernel sums the rows: two launches, no atomics, deterministic (the torch expression was 8 launches forward + 5 backward per call, three calls per step).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void atten_loss_rows_kernel(const float* __restrict__ gm, const float* __restrict__ att, long L, float* __restrict__ terms) {
    __shared__ float sh[2 * (NT / 64)];
    const long base = (long)blockIdx.x * L;
    float sv = 0.f, sg = 0.f;
    for (long i = threadIdx.x; i < L; i += NT) { const float g = gm[base + i]; sv += g * att[base + i]; sg += g; }
    sv = wave_sum(sv); sg = wave_sum(sg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave] = sv; sh[NT / 64 + wave] = sg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < NT / 64; ++w) { a += sh[w]; b += sh[NT / 64 + w]; }
        terms[blockIdx.x] = (b != 0.f ? 1.f : 0.f) - a;
    }
}
__global__ void atten_loss_sum_kernel(const float* __restrict__ terms, int rows, float scale, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 64) s += terms[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s * scale;
}
__global__ __launch_bounds__(NT) void atten_loss_bwd_kernel(const float* __restrict__ gm, const float* __restrict__ gout, float scale, long n, float* __restrict__ datt) {
    const float c = -scale * gout[0];
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) datt[i] = c * gm[i];
}

extern "C" int mg_atten_loss_fwd(const float* gm, const float* att, int rows, long L, float scale, float* terms, float* out, void* stream) {
    if (rows <= 0 || L <= 0) return -3;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(atten_loss_rows_kernel, dim3(rows), dim3(NT), 0, st, gm, att, L, terms);
    hipLaunchKernelGGL(atten_loss_sum_kernel, dim3(1), dim3(64), 0, st, (const float*)terms, rows, scale, out);
    MG_CHECK_LAUNCH();
    return 0;
}
extern "C" int mg_atten_loss_bwd(const float* gm, const float* gout, float scale, long n, float* datt, void* stream) {
    if (n <= 0) return 0;
    long b = (n + NT - 1) / NT; if (b > 1024) b = 1024;
    hipLaunchKernelGGL(atten_loss_bwd_kernel, dim3((unsigned)b), dim3(NT), 0, (hipStream_t)stream, gm, gout, scale, n, datt);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weighted sum of up to 16 device scalars (the loss bookkeeping of arch/maggie.py:283-300: loss_rec = 2 r1 + r4 + r8, ..., total = sum_k w_k L_k)
// and its backward gin[i] = c[i] * gout: one launch each instead of a dozen 1-element torch kernels forward and as many backward.
// ---------------------------------------------------------------------------------------------------------------------
struct ScalarTerms { const float* p[16]; float c[16]; int n; };
__global__ void scalar_lincomb_kernel(const ScalarTerms t, float* __restrict__ out) {
    if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < t.n; ++i) s += t.c[i] * t.p[i][0]; out[0] = s; }
}
__global__ void scalar_lincomb_bwd_kernel(const ScalarTerms t, const float* __restrict__ gout, float* __restrict__ gin) {
    if ((int)threadIdx.x < t.n) gin[threadIdx.x] = t.c[threadIdx.x] * gout[0];
}
extern "C" int mg_scalar_lincomb(const float* const* ptrs, const float* coef, int n, float* out, void* stream) {
    if (n <= 0 || n > 16) return -3;
    ScalarTerms t; t.n = n;
    for (int i = 0; i < 16; ++i) { t.p[i] = i < n ? ptrs[i] : nullptr; t.c[i] = i < n ? coef[i] : 0.f; }
    hipLaunchKernelGGL(scalar_lincomb_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, out);
    MG_CHECK_LAUNCH();
    return 0;
}
extern "C" int mg_scalar_lincomb_bwd(const float* coef, int n, const float* gout, float* gin, void* stream) {
    if (n <= 0 || n > 16) return -3;
    ScalarTerms t; t.n = n;
    for (int i = 0; i < 16; ++i) { t.p[i] = nullptr; t.c[i] = i < n ? coef[i] : 0.f; }
    hipLaunchKernelGGL(scalar_lincomb_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, gout, gin);
    MG_CHECK_LAUNCH();
    return 0;
}
