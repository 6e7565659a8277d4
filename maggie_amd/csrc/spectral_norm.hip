// SpectralNorm weight preparation on the device (reference: maggie/network/module/spectral_norm.py:22-35,73-80 --
// `torch.mv` x2 + norms + dot + divide per wrapped conv on EVERY forward; ~160 tiny GEMVs per step there).
//
//   t = W^T u ;  v = t / (|t| + eps) ;  s = W v ;  u' = s / (|s| + eps) ;  sigma = u' . (W v) = |s|^2 / (|s| + eps)
//   out = W / sigma   written straight into the conv kernels' layout (Cout, taps, Cin_pad) and compute dtype
//
// W is the fp32 parameter viewed as [A][B*taps] (nn.Conv2d: A = Cout, B = Cin; nn.ConvTranspose2d: A = Cin, B = Cout);
// u has A entries, v has B*taps entries (both updated in place, like the reference's `.data` rebinding).
// Backward:  dW = G / sigma - (<G, W> / sigma^2) * u v^T   with G = dL/d(out) given in the kernels' layout (fp32).
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;
constexpr float SN_EPS = 1e-12f;

// scratch layout (fp32): [0] = |t|^2, [1] = |s'|^2 (s' = W t, unnormalised), [2] = <G, W>, [3] = sigma
__global__ __launch_bounds__(NT) void sn_wt_u_kernel(const float* __restrict__ W, const float* __restrict__ u, int A, int Wd,
                                                     float* __restrict__ t, int rows_per_block) {
    const int j = blockIdx.x * NT + threadIdx.x;
    const int i0 = blockIdx.y * rows_per_block, i1 = min(A, i0 + rows_per_block);
    if (j >= Wd) return;
    float acc = 0.f;
    for (int i = i0; i < i1; ++i) acc += W[(long)i * Wd + j] * u[i];
    atomicAdd(&t[j], acc);
}

// one wave per row: s'[i] = sum_j W[i,j] t[j];  block (0,0) also reduces |t|^2
__global__ __launch_bounds__(NT) void sn_w_t_kernel(const float* __restrict__ W, const float* __restrict__ t, int A, int Wd,
                                                    float* __restrict__ s, float* __restrict__ scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i < A) {
        float acc = 0.f;
        for (int j = lane; j < Wd; j += 64) acc += W[(long)i * Wd + j] * t[j];
        acc = wave_sum(acc);
        if (lane == 0) { s[i] = acc; atomicAdd(&scratch[1], acc * acc); }
    }
    if (blockIdx.x == 0) {
        float a = 0.f;
        for (int j = threadIdx.x; j < Wd; j += NT) a += t[j] * t[j];
        a = wave_sum(a);
        if (lane == 0) atomicAdd(&scratch[0], a);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void sn_finish_kernel(const float* __restrict__ W, const float* __restrict__ t, const float* __restrict__ s,
                                                       float* __restrict__ scratch, int A, int B, int taps, int transposed, int pad_in,
                                                       float* __restrict__ u, float* __restrict__ v, T* __restrict__ out) {
    const float nt = sqrtf(scratch[0]);
    const float sv = 1.f / (nt + SN_EPS);                         // v = t * sv
    const float ns = sqrtf(scratch[1]) * sv;                      // |s| with s = W v
    const float su = sv / (ns + SN_EPS);                          // u' = s' * su
    const float sigma = ns * ns / (ns + SN_EPS);
    const float inv_sigma = 1.f / sigma;
    const int Wd = B * taps;
    const long gid = (long)blockIdx.x * NT + threadIdx.x;
    if (gid < A) u[gid] = s[gid] * su;
    if (gid < Wd) v[gid] = t[gid] * sv;
    if (gid == 0) scratch[3] = sigma;
    // output element (co, tap, ci) with ci < pad_in
    const int Cout = transposed ? B : A, Cin = transposed ? A : B;
    const long total = (long)Cout * taps * pad_in;
    for (long o = gid; o < total; o += (long)gridDim.x * NT) {
        int ci = (int)(o % pad_in); long r = o / pad_in; int tap = (int)(r % taps); int co = (int)(r / taps);
        float val = 0.f;
        if (ci < Cin) {
            int a = transposed ? ci : co, b = transposed ? co : ci;
            val = W[((long)a * B + b) * taps + tap] * inv_sigma;
        }
        ElemTraits<T>::st(out + o, val);
    }
}

// <G, W> with G in (Cout, taps, pad_in) fp32 layout
__global__ __launch_bounds__(NT) void sn_bwd_dot_kernel(const float* __restrict__ G, const float* __restrict__ W, int A, int B, int taps,
                                                        int transposed, int pad_in, float* __restrict__ scratch) {
    const long total = (long)A * B * taps;
    float acc = 0.f;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int tap = (int)(e % taps); long r = e / taps; int b = (int)(r % B); int a = (int)(r / B);
        int co = transposed ? b : a, ci = transposed ? a : b;
        acc += G[((long)co * taps + tap) * pad_in + ci] * W[e];
    }
    __shared__ float sh[NT / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&scratch[2], sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ __launch_bounds__(NT) void sn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ u, const float* __restrict__ v,
                                                          const float* __restrict__ scratch, int A, int B, int taps, int transposed,
                                                          int pad_in, float* __restrict__ dW) {
    const float sigma = scratch[3];
    const float inv_sigma = 1.f / sigma;
    const float coef = scratch[2] * inv_sigma * inv_sigma;
    const long total = (long)A * B * taps;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int tap = (int)(e % taps); long r = e / taps; int b = (int)(r % B); int a = (int)(r / B);
        int co = transposed ? b : a, ci = transposed ? a : b;
        float g = G[((long)co * taps + tap) * pad_in + ci];
        dW[e] = g * inv_sigma - coef * u[a] * v[b * taps + tap];
    }
}

inline int grid_for(long total) { long b = (total + NT - 1) / NT; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

}  // namespace

// work: fp32 scratch of at least (A + B*taps + 4) floats: [t (B*taps) | s (A) | scratch (4)]; scratch[3] returns sigma.
extern "C" int mg_spectral_norm(const float* W, float* u, float* v, int A, int B, int taps, int transposed, int pad_in, void* out,
                                int out_dtype, float* work, void* stream) {
    if (A <= 0 || B <= 0 || taps <= 0) return -2;
    const int Wd = B * taps;
    hipStream_t st = (hipStream_t)stream;
    float* t = work;
    float* s = work + Wd;
    float* scratch = work + Wd + A;
    hipError_t e = hipMemsetAsync(work, 0, (size_t)(Wd + A + 4) * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    int rpb = 32;
    dim3 g1((Wd + NT - 1) / NT, (A + rpb - 1) / rpb);
    hipLaunchKernelGGL(sn_wt_u_kernel, g1, dim3(NT), 0, st, W, u, A, Wd, t, rpb);
    hipLaunchKernelGGL(sn_w_t_kernel, dim3((A + 3) / 4), dim3(NT), 0, st, W, t, A, Wd, s, scratch);
    const int Cout = transposed ? B : A;
    long total = (long)Cout * taps * pad_in;
    long need = total > Wd ? total : Wd;
    if (need < A) need = A;
    int blocks = grid_for(need);
    if ((long)blocks * NT < (Wd > A ? Wd : A)) blocks = (int)(((Wd > A ? Wd : A) + NT - 1) / NT);
    if (out_dtype == MG_BF16) hipLaunchKernelGGL(sn_finish_kernel<bf16raw>, dim3(blocks), dim3(NT), 0, st, W, t, s, scratch, A, B, taps, transposed, pad_in, u, v, (bf16raw*)out);
    else hipLaunchKernelGGL(sn_finish_kernel<float>, dim3(blocks), dim3(NT), 0, st, W, t, s, scratch, A, B, taps, transposed, pad_in, u, v, (float*)out);
    MG_CHECK_LAUNCH();
    return 0;
}

// G: fp32 (Cout, taps, pad_in); u, v: the vectors AFTER the forward's power iteration; work: the forward's scratch block
extern "C" int mg_spectral_norm_bwd(const float* G, const float* W, const float* u, const float* v, int A, int B, int taps, int transposed,
                                    int pad_in, float* work, float* dW, void* stream) {
    const int Wd = B * taps;
    hipStream_t st = (hipStream_t)stream;
    float* scratch = work + Wd + A;
    hipError_t e = hipMemsetAsync(scratch + 2, 0, sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    long total = (long)A * B * taps;
    int dot_blocks = grid_for(total / 8); if (dot_blocks > 256) dot_blocks = 256;
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(dot_blocks), dim3(NT), 0, st, G, W, A, B, taps, transposed, pad_in, scratch);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(grid_for(total)), dim3(NT), 0, st, G, u, v, scratch, A, B, taps, transposed, pad_in, dW);
    MG_CHECK_LAUNCH();
    return 0;
}
