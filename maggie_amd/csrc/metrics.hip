// Validation metrics as device reductions: maggie/utils/metric.py SAD / MSE / MAD (:68-97), Grad (:352-417),
// dtSSD (:422-448). The reference copies every prediction to the host and reduces in numpy (and ships it back to the GPU for the
// Gaussian-gradient convolutions); here the planes never leave HBM and only a handful of doubles come back.
// All HBM-bound: one pass over pred / gt / trimap per metric family; fp32 per-thread partials, fp64 block reduction and atomics.
#include <limits.h>
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float mask_of(const float* __restrict__ t, long i, int mode) {
    if (mode == 0 || t == nullptr) return 1.f;
    const float v = t[i];
    return mode == 1 ? (v > 0.f ? 1.f : 0.f) : (v == 1.f ? 1.f : 0.f);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int NV>
__device__ __forceinline__ void block_atomic_add(const float* part, double* dst) {
    __shared__ double sh[NT / 64][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum_d((double)part[k]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += sh[w][threadIdx.x];
        atomicAdd(dst + threadIdx.x, s);
    }
}

// out[p] = { sum |d| m, sum d^2 m, sum m }
__global__ __launch_bounds__(NT) void plane_sums_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ tri,
                                                        int mode, long HW, double* __restrict__ out) {
    const long base = (long)blockIdx.y * HW;
    float part[3] = {0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        const float d = pred[base + i] - gt[base + i], m = mask_of(tri, base + i, mode);
        part[0] += fabsf(d) * m;
        part[1] += d * d * m;
        part[2] += m;
    }
    block_atomic_add<3>(part, out + (long)blockIdx.y * 3);
}

// ---- Grad ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void minmax_init_kernel(int* mm) { if (threadIdx.x < 4) mm[threadIdx.x] = (threadIdx.x & 1) ? INT_MIN : INT_MAX; }

// mm[0..1] = (min, max) of a, mm[2..3] of b (order-preserving int encoding)
__global__ __launch_bounds__(NT) void minmax_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, int* __restrict__ mm) {
    float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float x = a[i], y = b[i];
        lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
        lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_down(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off, 64));
        }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm + 0, f2ord(lo[0])); atomicMax(mm + 1, f2ord(hi[0]));
        atomicMin(mm + 2, f2ord(lo[1])); atomicMax(mm + 3, f2ord(hi[1]));
    }
}

struct Filt { float f[81]; };
constexpr int GT = 16, GH = 4, GS = GT + 2 * GH;             // 16x16 outputs, halo 4 (9x9 taps)

// out[p] += sum_tile ( |grad(gt_normed)| - |grad(pred_normed)| )^2 * mask
__global__ __launch_bounds__(NT) void grad_metric_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ tri,
                                                         int mode, int H, int W, Filt fx, const int* __restrict__ mm, double* __restrict__ out) {
    __shared__ float sp[GS][GS + 1], sg[GS][GS + 1];
    const int p = blockIdx.z, ty0 = blockIdx.y * GT, tx0 = blockIdx.x * GT;
    const long base = (long)p * H * W;
    const float pmin = ord2f(mm[0]), pden = ord2f(mm[1]) - pmin + 1e-6f;
    const float gmin = ord2f(mm[2]), gden = ord2f(mm[3]) - gmin + 1e-6f;
    for (int i = threadIdx.x; i < GS * GS; i += NT) {
        const int ly = i / GS, lx = i - ly * GS, y = ty0 + ly - GH, x = tx0 + lx - GH;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;      // conv2d zero padding is applied to the NORMALISED image
        sp[ly][lx] = in ? __fdiv_rn(pred[base + (long)y * W + x] - pmin, pden) : 0.f;
        sg[ly][lx] = in ? __fdiv_rn(gt[base + (long)y * W + x] - gmin, gden) : 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x / GT, lx = threadIdx.x - ly * GT, y = ty0 + ly, x = tx0 + lx;
    float part[1] = {0.f};
    if (y < H && x < W) {
        float px = 0.f, py = 0.f, gx = 0.f, gy = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float a = sp[ly + i][lx + j], b = sg[ly + i][lx + j];
                px += a * fx.f[i * 9 + j]; py += a * fx.f[j * 9 + i];      // filter_y = filter_x transposed
                gx += b * fx.f[i * 9 + j]; gy += b * fx.f[j * 9 + i];
            }
        const float d = sqrtf(gx * gx + gy * gy) - sqrtf(px * px + py * py);
        part[0] = d * d * mask_of(tri, base + (long)y * W + x, mode);
    }
    block_atomic_add<1>(part, out + p);
}

// ---- dtSSD: out[n] += sum ((p[t+1]-p[t]) - (g[t+1]-g[t]))^2 * m[t]   over b, t < T-1, pixels ----------------------------------
__global__ __launch_bounds__(NT) void dtssd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ tri,
                                                   int mode, int T, int N, long HW, double* __restrict__ out) {
    const int n = blockIdx.y % N, bt = blockIdx.y / N, t = bt % (T - 1), b = bt / (T - 1);
    const long cur = (((long)b * T + t) * N + n) * HW, nxt = cur + (long)N * HW;
    float part[1] = {0.f};
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        const float d = (pred[nxt + i] - pred[cur + i]) - (gt[nxt + i] - gt[cur + i]);
        part[0] += d * d * mask_of(tri, cur + i, mode);
    }
    block_atomic_add<1>(part, out + n);
}

static inline unsigned blocks_for(long n, long per_thread, unsigned cap) {
    long b = (n + NT * per_thread - 1) / (NT * per_thread);
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int mg_metric_plane_sums(const float* pred, const float* gt, const float* trimap, int mask_mode, int P, long HW, double* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (P <= 0) return 0;
    if (P > 65535) return -3;
    hipError_t e = mg_zero_words(out, (long)P * 6, st);
    if (e != hipSuccess) return (int)e;
    if (HW <= 0) return 0;
    hipLaunchKernelGGL(plane_sums_kernel, dim3(blocks_for(HW, 16, 256), P), dim3(NT), 0, st, pred, gt, trimap, mask_mode, HW, out);
    return (int)hipGetLastError();
}

extern "C" int mg_metric_grad(const float* pred, const float* gt, const float* trimap, int mask_mode, int P, int H, int W, const float* filter_x81,
                              int32_t* scratch4, double* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (P <= 0) return 0;
    if (P > 65535) return -3;
    hipError_t e = mg_zero_words(out, (long)P * 2, st);
    if (e != hipSuccess) return (int)e;
    if (H <= 0 || W <= 0) return 0;
    Filt fx;
    for (int i = 0; i < 81; ++i) fx.f[i] = filter_x81[i];
    const long n = (long)P * H * W;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(64), 0, st, scratch4);
    hipLaunchKernelGGL(minmax_kernel, dim3(blocks_for(n, 16, 1024)), dim3(NT), 0, st, pred, gt, n, scratch4);
    hipLaunchKernelGGL(grad_metric_kernel, dim3((W + GT - 1) / GT, (H + GT - 1) / GT, P), dim3(NT), 0, st, pred, gt, trimap, mask_mode, H, W, fx,
                       scratch4, out);
    return (int)hipGetLastError();
}

extern "C" int mg_metric_dtssd(const float* pred, const float* gt, const float* trimap, int mask_mode, int B, int T, int N, long HW, double* out,
                               void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0) return 0;
    hipError_t e = mg_zero_words(out, (long)N * 2, st);
    if (e != hipSuccess) return (int)e;
    if (B <= 0 || T < 2 || HW <= 0) return 0;
    if ((long)B * (T - 1) * N > 65535) return -3;
    hipLaunchKernelGGL(dtssd_kernel, dim3(blocks_for(HW, 16, 128), B * (T - 1) * N), dim3(NT), 0, st, pred, gt, trimap, mask_mode, T, N, HW, out);
    return (int)hipGetLastError();
}
