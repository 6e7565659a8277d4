// BatchNorm (training statistics, folding, apply+activation+residual, backward) and related row x channel
// elementwise kernels for NHWC / sparse-row feature matrices on gfx950. All are HBM-bound: 16-byte vector accesses,
// per-thread fixed channel group so per-channel parameters live in registers, column reductions finished with one
// atomicAdd per channel per block.
// Replaces nn.BatchNorm2d / nn.BatchNorm1d (+ReLU / LeakyReLU / residual adds) of
//   maggie/network/encoder/resnet.py:23-39,167-175; maggie/network/decoder/resnet.py:28-45;
//   maggie/network/module/aspp.py:34-56; maggie/network/decoder/resnet_inst_matt_spconv.py:69-130.
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------------
// column statistics: stats[c] += sum_m x[m,c]; stats[C+c] += sum_m x[m,c]^2
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void colstats_kernel(const T* __restrict__ x, int M, int C, int ld, float* __restrict__ stats,
                                                      int rows_per_block, int nrep, int only_sum) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;                       // chunks per row (C % CE == 0)
    const int t = threadIdx.x;
    const int tpr = cpr < NT ? cpr : NT;          // threads used per row sweep
    const int rstep = NT / tpr;
    const int cc0 = t % tpr, rr = t / tpr;
    extern __shared__ float sred[];               // [2*C]
    for (int i = t; i < 2 * C; i += NT) sred[i] = 0.f;
    __syncthreads();
    const int mbeg = blockIdx.x * rows_per_block, mend = min(M, mbeg + rows_per_block);
    if (rr < rstep) {
        for (int cc = cc0; cc < cpr; cc += tpr) {
            float s1[CE], s2[CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
            for (int m = mbeg + rr; m < mend; m += rstep) {
                uint4 q = *(const uint4*)(x + (long)m * ld + cc * CE);
                float f[CE];
                TR::unpack(q, f);
#pragma unroll
                for (int e = 0; e < CE; ++e) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) { atomicAdd(&sred[cc * CE + e], s1[e]); atomicAdd(&sred[C + cc * CE + e], s2[e]); }
        }
    }
    __syncthreads();
    float* st = stats + (size_t)(blockIdx.x & (nrep - 1)) * 2 * C;
    const int lim = only_sum ? C : 2 * C;
    for (int i = t; i < lim; i += NT) atomicAdd(&st[i], sred[i]);
}

// second pass of the exact two-pass variance: stats[C+c] += sum_m (x[m,c] - stats[c]/M)^2   (stats[0:C] = column sums)
template <typename T>
__global__ __launch_bounds__(NT) void colstats_centered_kernel(const T* __restrict__ x, int M, int C, int ld, float* __restrict__ stats,
                                                               int rows_per_block) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const int t = threadIdx.x;
    const int tpr = cpr < NT ? cpr : NT;
    const int rstep = NT / tpr;
    const int cc0 = t % tpr, rr = t / tpr;
    extern __shared__ float sred[];               // [C]
    for (int i = t; i < C; i += NT) sred[i] = 0.f;
    __syncthreads();
    const float inv_m = 1.f / (float)M;
    const int mbeg = blockIdx.x * rows_per_block, mend = min(M, mbeg + rows_per_block);
    if (rr < rstep) {
        for (int cc = cc0; cc < cpr; cc += tpr) {
            float s2[CE], mu[CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) { s2[e] = 0.f; mu[e] = stats[cc * CE + e] * inv_m; }
            for (int m = mbeg + rr; m < mend; m += rstep) {
                float f[CE];
                TR::unpack(*(const uint4*)(x + (long)m * ld + cc * CE), f);
#pragma unroll
                for (int e = 0; e < CE; ++e) { float d = f[e] - mu[e]; s2[e] += d * d; }
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) atomicAdd(&sred[cc * CE + e], s2[e]);
        }
    }
    __syncthreads();
    for (int i = t; i < C; i += NT) atomicAdd(&stats[C + i], sred[i]);
}

// ---------------------------------------------------------------------------------------------------
// finalize: batch statistics -> (scale, shift, mean, invstd) + running-stat update (momentum, unbiased var)
// ---------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int nrep, const float* __restrict__ count_ptr, float count, int C, int centered,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float n = count_ptr ? *count_ptr : count;
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < nrep; ++r) { s1 += stats[(size_t)r * 2 * C + c]; s2 += stats[(size_t)r * 2 * C + C + c]; }
    float mean = s1 / n;
    float var = centered ? s2 / n : s2 / n - mean * mean;
    var = var > 0.f ? var : 0.f;
    float invstd = rsqrtf(var + eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - mean * g * invstd;
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    if (running_mean) {
        float unbiased = n > 1.f ? var * n / (n - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

__global__ void bn_fold_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                               float* __restrict__ scale, float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = rsqrtf(running_var[c] + eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - running_mean[c] * g * invstd;
}

// ---------------------------------------------------------------------------------------------------
// y = act(x*scale + shift + res) + res2      (res optionally at half resolution, nearest x2)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void affine_act_kernel(const mg_rowwise_params p) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = p.C / CE;
    const long total = (long)p.M * cpr;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ r1 = (const T*)p.res;
    const T* __restrict__ r2 = (const T*)p.res2;
    T* __restrict__ y = (T*)p.y;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int m = (int)(i / cpr), cc = (int)(i - (long)m * cpr);
        int c0 = cc * CE;
        float f[CE], a[CE], b[CE];
        TR::unpack(*(const uint4*)(x + (long)m * p.ldx + c0), f);
#pragma unroll
        for (int e = 0; e < CE; ++e) { a[e] = 0.f; b[e] = 0.f; }
        if (r1) {
            long rrow = m;
            if (p.res_mode == 2) {
                int hw = p.H * p.W; int n = m / hw; int rem = m - n * hw; int ho = rem / p.W; int wo = rem - ho * p.W;
                rrow = ((long)n * (p.H >> 1) + (ho >> 1)) * (p.W >> 1) + (wo >> 1);
            }
            TR::unpack(*(const uint4*)(r1 + rrow * p.ldr + c0), a);
        }
        if (r2) TR::unpack(*(const uint4*)(r2 + (long)m * p.ldr2 + c0), b);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            float v = f[e];
            float sc = p.scale ? p.scale[c0 + e] : 1.f, sh = p.shift ? p.shift[c0 + e] : 0.f;
            v = v * sc + sh + a[e];
            v = apply_act(v, p.act, p.slope) + b[e];
            f[e] = v;
        }
        *(uint4*)(y + (long)m * p.ldy + p.yoff + c0) = TR::pack(f);
    }
}

// ---------------------------------------------------------------------------------------------------
// BN backward, pass 1: sums[c] += sum_m g, sums[C+c] += sum_m g*xhat,  g = dy * act'(y) (y = post-activation output)
// BN backward, pass 2: dx = scale_c * (g - sum_g/n - xhat * sum_gx/n)   [ * (x > 0) for the ReLU-before-BN shortcut ]
//                      dres = g (optional, gradient of the pre-activation residual)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_g(const mg_rowwise_params& p, int m, int c0, float* g) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    TR::unpack(*(const uint4*)((const T*)p.dy + (long)m * p.lddy + c0), g);
    if (p.act != MG_ACT_NONE) {
        float yv[CE];
        TR::unpack(*(const uint4*)((const T*)p.y + (long)m * p.ldy + p.yoff + c0), yv);
        if (p.res2) {                      // y = act(.) + res2  -> recover the activation output sign
            float b[CE];
            TR::unpack(*(const uint4*)((const T*)p.res2 + (long)m * p.ldr2 + c0), b);
#pragma unroll
            for (int e = 0; e < CE; ++e) yv[e] -= b[e];
        }
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            if (!(yv[e] > 0.f)) g[e] = (p.act == MG_ACT_RELU) ? 0.f : g[e] * p.slope;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const mg_rowwise_params p, int rows_per_block) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int C = p.C, cpr = C / CE, t = threadIdx.x;
    const int tpr = cpr < NT ? cpr : NT, rstep = NT / tpr, cc0 = t % tpr, rr = t / tpr;
    extern __shared__ float sred[];
    for (int i = t; i < 2 * C; i += NT) sred[i] = 0.f;
    __syncthreads();
    const int mbeg = blockIdx.x * rows_per_block, mend = min(p.M, mbeg + rows_per_block);
    if (rr < rstep) {
        for (int cc = cc0; cc < cpr; cc += tpr) {
            const int c0 = cc * CE;
            float s1[CE], s2[CE], mu[CE], is[CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) { s1[e] = 0.f; s2[e] = 0.f; mu[e] = p.mean[c0 + e]; is[e] = p.invstd[c0 + e]; }
            for (int m = mbeg + rr; m < mend; m += rstep) {
                float g[CE], xv[CE];
                load_g<T>(p, m, c0, g);
                TR::unpack(*(const uint4*)((const T*)p.x + (long)m * p.ldx + c0), xv);
#pragma unroll
                for (int e = 0; e < CE; ++e) { s1[e] += g[e]; s2[e] += g[e] * (xv[e] - mu[e]) * is[e]; }
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) { atomicAdd(&sred[c0 + e], s1[e]); atomicAdd(&sred[C + c0 + e], s2[e]); }
        }
    }
    __syncthreads();
    for (int i = t; i < 2 * C; i += NT) atomicAdd(&p.sums[i], sred[i]);
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const mg_rowwise_params p) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = p.C / CE;
    const long total = (long)p.M * cpr;
    const float inv_n = 1.f / (p.count_ptr ? *p.count_ptr : p.count);
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int m = (int)(i / cpr), cc = (int)(i - (long)m * cpr);
        int c0 = cc * CE;
        float g[CE], xv[CE], o[CE];
        load_g<T>(p, m, c0, g);
        if (p.dres) *(uint4*)((T*)p.dres + (long)m * p.lddres + c0) = TR::pack(g);
        if (p.dx) {
            TR::unpack(*(const uint4*)((const T*)p.x + (long)m * p.ldx + c0), xv);
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                int c = c0 + e;
                float xh = (xv[e] - p.mean[c]) * p.invstd[c];
                float v = p.scale[c] * (g[e] - p.sums[c] * inv_n - xh * p.sums[p.C + c] * inv_n);
                if (p.mask_x_pos && !(xv[e] > 0.f)) v = 0.f;
                o[e] = v;
            }
            *(uint4*)((T*)p.dx + (long)m * p.lddx + c0) = TR::pack(o);
        }
    }
}

// dgamma[c] = sums[C+c], dbeta[c] = sums[c] are read directly by the host side (fp32 tensors).

// ---------------------------------------------------------------------------------------------------
// 2x2 average pool (AvgPool2d(2,2), encoder/resnet.py:113), its backward, 2x2 sum (nearest-upsample backward)
// ---------------------------------------------------------------------------------------------------
template <typename T, int OP>   // OP 0: out[n,ho,wo] = mean of 2x2 of in; 1: out = sum of 2x2 of in; 2: out[n,h,w] = 0.25*in[n,h/2,w/2]; 3: out = in[n,h/2,w/2]
__global__ __launch_bounds__(NT) void pool2x2_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int Ho, int Wo, int C,
                                                     int Hi, int Wi) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)N * Ho * Wo * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int cc = (int)(i % cpr); long r = i / cpr;
        int wo = (int)(r % Wo); r /= Wo; int ho = (int)(r % Ho); int n = (int)(r / Ho);
        float o[CE];
        if (OP <= 1) {
#pragma unroll
            for (int e = 0; e < CE; ++e) o[e] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float f[CE];
                    TR::unpack(*(const uint4*)(in + (((long)n * Hi + 2 * ho + dy) * Wi + 2 * wo + dx) * C + cc * CE), f);
#pragma unroll
                    for (int e = 0; e < CE; ++e) o[e] += f[e];
                }
            if (OP == 0) {
#pragma unroll
                for (int e = 0; e < CE; ++e) o[e] *= 0.25f;
            }
        } else {
            TR::unpack(*(const uint4*)(in + (((long)n * Hi + (ho >> 1)) * Wi + (wo >> 1)) * C + cc * CE), o);
            if (OP == 2) {
#pragma unroll
                for (int e = 0; e < CE; ++e) o[e] *= 0.25f;
            }
        }
        *(uint4*)(out + (((long)n * Ho + ho) * Wo + wo) * C + cc * CE) = TR::pack(o);
    }
}

inline int grid_for(long total) {
    long b = (total + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int mg_colstats(const void* x, int dtype, int M, int C, int ld, float* stats, void* stream) {
    if (M <= 0) return 0;
    const int ce = dtype == MG_BF16 ? 8 : 4;
    if (C % ce || ld % ce) return -3;
    int blocks = (M + 255) / 256; if (blocks > 1024) blocks = 1024;
    int rpb = (M + blocks - 1) / blocks;
    blocks = (M + rpb - 1) / rpb;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(colstats_kernel<bf16raw>, dim3(blocks), dim3(NT), 2 * C * 4, st, (const bf16raw*)x, M, C, ld, stats, rpb, MG_STAT_REPLICAS, 0);
    else hipLaunchKernelGGL(colstats_kernel<float>, dim3(blocks), dim3(NT), 2 * C * 4, st, (const float*)x, M, C, ld, stats, rpb, MG_STAT_REPLICAS, 0);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_colstats_centered(const void* x, int dtype, int M, int C, int ld, float* stats, void* stream) {
    if (M <= 0) return 0;
    const int ce = dtype == MG_BF16 ? 8 : 4;
    if (C % ce || ld % ce) return -3;
    int blocks = (M + 255) / 256; if (blocks > 1024) blocks = 1024;
    int rpb = (M + blocks - 1) / blocks;
    blocks = (M + rpb - 1) / rpb;
    hipStream_t st = (hipStream_t)stream;
    // `stats` must arrive zeroed (the caller hands out slices of a per-step zero arena): pass 1 adds the column sums only,
    // pass 2 the centred second moments
    if (dtype == MG_BF16) {
        hipLaunchKernelGGL(colstats_kernel<bf16raw>, dim3(blocks), dim3(NT), 2 * C * 4, st, (const bf16raw*)x, M, C, ld, stats, rpb, 1, 1);
        hipLaunchKernelGGL(colstats_centered_kernel<bf16raw>, dim3(blocks), dim3(NT), C * 4, st, (const bf16raw*)x, M, C, ld, stats, rpb);
    } else {
        hipLaunchKernelGGL(colstats_kernel<float>, dim3(blocks), dim3(NT), 2 * C * 4, st, (const float*)x, M, C, ld, stats, rpb, 1, 1);
        hipLaunchKernelGGL(colstats_centered_kernel<float>, dim3(blocks), dim3(NT), C * 4, st, (const float*)x, M, C, ld, stats, rpb);
    }
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bn_finalize(const float* stats, int nrep, const float* count_ptr, float count, int C, int centered, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                              float* mean_out, float* invstd_out, void* stream) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, nrep, count_ptr, count, C, centered, gamma,
                       beta, running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bn_fold(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, float* scale, float* shift, void* stream) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, C, gamma, beta, running_mean,
                       running_var, eps, scale, shift);
    MG_CHECK_LAUNCH();
    return 0;
}

static int rowwise_check(const mg_rowwise_params* p) {
    if (!p) return -1;
    const int ce = p->dtype == MG_BF16 ? 8 : 4;
    if (p->C % ce) return -3;
    return 0;
}

extern "C" int mg_affine_act(const mg_rowwise_params* p, void* stream) {
    int rc = rowwise_check(p); if (rc) return rc;
    if (p->M <= 0) return 0;
    const int ce = p->dtype == MG_BF16 ? 8 : 4;
    long total = (long)p->M * (p->C / ce);
    if (p->dtype == MG_BF16) hipLaunchKernelGGL(affine_act_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(affine_act_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, *p);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bn_bwd_reduce(const mg_rowwise_params* p, void* stream) {
    int rc = rowwise_check(p); if (rc) return rc;
    if (p->M <= 0) return 0;
    // every block ends with 2C global atomics on the same 2C addresses: keep the block count near 2 per CU
    int blocks = (p->M + 255) / 256; if (blocks > 512) blocks = 512;
    int rpb = (p->M + blocks - 1) / blocks;
    blocks = (p->M + rpb - 1) / rpb;
    if (p->dtype == MG_BF16) hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16raw>, dim3(blocks), dim3(NT), 2 * p->C * 4, (hipStream_t)stream, *p, rpb);
    else hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(blocks), dim3(NT), 2 * p->C * 4, (hipStream_t)stream, *p, rpb);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bn_bwd_apply(const mg_rowwise_params* p, void* stream) {
    int rc = rowwise_check(p); if (rc) return rc;
    if (p->M <= 0) return 0;
    const int ce = p->dtype == MG_BF16 ? 8 : 4;
    long total = (long)p->M * (p->C / ce);
    if (p->dtype == MG_BF16) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, *p);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_pool2x2(const void* in, void* out, int dtype, int op, int N, int Ho, int Wo, int C, void* stream) {
    const int ce = dtype == MG_BF16 ? 8 : 4;
    if (C % ce) return -3;
    long total = (long)N * Ho * Wo * (C / ce);
    if (total <= 0) return 0;
    int Hi = (op <= 1) ? Ho * 2 : Ho / 2, Wi = (op <= 1) ? Wo * 2 : Wo / 2;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(grid_for(total)), b(NT);
#define POOL_CASE(T, OP) hipLaunchKernelGGL((pool2x2_kernel<T, OP>), g, b, 0, st, (const T*)in, (T*)out, N, Ho, Wo, C, Hi, Wi)
    if (dtype == MG_BF16) {
        switch (op) { case 0: POOL_CASE(bf16raw, 0); break; case 1: POOL_CASE(bf16raw, 1); break; case 2: POOL_CASE(bf16raw, 2); break; case 3: POOL_CASE(bf16raw, 3); break; default: return -2; }
    } else {
        switch (op) { case 0: POOL_CASE(float, 0); break; case 1: POOL_CASE(float, 1); break; case 2: POOL_CASE(float, 2); break; case 3: POOL_CASE(float, 3); break; default: return -2; }
    }
#undef POOL_CASE
    MG_CHECK_LAUNCH();
    return 0;
}
