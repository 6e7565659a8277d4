// Temporal (video) elementwise kernels.
//  * ConvGRU gate math of maggie/network/module/conv_gru.py:22-27 fused around the two 3x3 gate convolutions (which run on the
//    implicit-GEMM kernel): one kernel turns the first conv's output into the second conv's input ([x, sigmoid(r) * h]), one turns
//    the second conv's output into the new hidden state ((1 - z) h + z tanh(c)); both with exact backward kernels. In the
//    reference that is sigmoid, split, mul, cat, tanh, three blend ops per frame -- and twice as many kernels backward.
//  * The eval-time alpha-level aggregation over frames 0, 1, 2 of maggie/network/arch/maggie_temp.py:34-77 as ONE in-place kernel
//    (threshold the difference maps, propagate t-1 -> t and t+1 -> t, keep the model's own prediction where they disagree,
//    propagate t -> t+1).
// All tensors are rows x channels (NHWC) in the compute dtype T; pure HBM streaming, 16 bytes per lane.
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

inline int grid_for(long total) {
    long b = (total + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// rz: (M, 2C) pre-activation [r | z]; x, h: (M, C).  xrh: (M, 2C) = [x | sigmoid(r) * h]
template <typename T>
__global__ __launch_bounds__(NT) void gru_gate_fwd_kernel(const T* __restrict__ rz, const T* __restrict__ x, const T* __restrict__ h, int M, int C,
                                                          T* __restrict__ xrh) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)M * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i / cpr; const int c0 = (int)(i - m * cpr) * CE;
        float r[CE], hv[CE];
        TR::unpack(*(const uint4*)(rz + m * 2 * C + c0), r);
        TR::unpack(*(const uint4*)(h + m * C + c0), hv);
#pragma unroll
        for (int e = 0; e < CE; ++e) r[e] = sigm(r[e]) * hv[e];
        *(uint4*)(xrh + m * 2 * C + c0) = *(const uint4*)(x + m * C + c0);
        *(uint4*)(xrh + m * 2 * C + C + c0) = TR::pack(r);
    }
}

// given d(xrh) (M, 2C): dx (+)= d[:, :C];  drz[:, :C] = d[:, C:] * h * r (1 - r);  dh_part = d[:, C:] * r   (written, not accumulated)
template <typename T>
__global__ __launch_bounds__(NT) void gru_gate_bwd_kernel(const T* __restrict__ dxrh, const T* __restrict__ rz, const T* __restrict__ h, int M, int C,
                                                          T* __restrict__ dx, T* __restrict__ dr_pre, int ld_dr, T* __restrict__ dh_part) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)M * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i / cpr; const int c0 = (int)(i - m * cpr) * CE;
        float g[CE], r[CE], hv[CE], o1[CE], o2[CE];
        TR::unpack(*(const uint4*)(dxrh + m * 2 * C + C + c0), g);
        TR::unpack(*(const uint4*)(rz + m * 2 * C + c0), r);
        TR::unpack(*(const uint4*)(h + m * C + c0), hv);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const float s = sigm(r[e]);
            o1[e] = g[e] * hv[e] * s * (1.f - s);
            o2[e] = g[e] * s;
        }
        *(uint4*)(dx + m * C + c0) = *(const uint4*)(dxrh + m * 2 * C + c0);
        *(uint4*)(dr_pre + m * ld_dr + c0) = TR::pack(o1);
        *(uint4*)(dh_part + m * C + c0) = TR::pack(o2);
    }
}

// hn = (1 - z) h + z tanh(c),  z = sigmoid(rz[:, C:])
template <typename T>
__global__ __launch_bounds__(NT) void gru_out_fwd_kernel(const T* __restrict__ rz, const T* __restrict__ cpre, const T* __restrict__ h, int M, int C,
                                                         T* __restrict__ hn) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)M * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i / cpr; const int c0 = (int)(i - m * cpr) * CE;
        float z[CE], c[CE], hv[CE];
        TR::unpack(*(const uint4*)(rz + m * 2 * C + C + c0), z);
        TR::unpack(*(const uint4*)(cpre + m * C + c0), c);
        TR::unpack(*(const uint4*)(h + m * C + c0), hv);
#pragma unroll
        for (int e = 0; e < CE; ++e) { const float zs = sigm(z[e]); z[e] = (1.f - zs) * hv[e] + zs * tanhf(c[e]); }
        *(uint4*)(hn + m * C + c0) = TR::pack(z);
    }
}

// given dhn: dz_pre = dhn (tanh c - h) z (1 - z) -> drz[:, C:];  dc_pre = dhn z (1 - tanh^2 c);  dh_part2 = dhn (1 - z)
template <typename T>
__global__ __launch_bounds__(NT) void gru_out_bwd_kernel(const T* __restrict__ dhn, const T* __restrict__ rz, const T* __restrict__ cpre,
                                                         const T* __restrict__ h, int M, int C, T* __restrict__ dz_pre, int ld_dz,
                                                         T* __restrict__ dc_pre, T* __restrict__ dh_part) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)M * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i / cpr; const int c0 = (int)(i - m * cpr) * CE;
        float g[CE], z[CE], c[CE], hv[CE], o1[CE], o2[CE], o3[CE];
        TR::unpack(*(const uint4*)(dhn + m * C + c0), g);
        TR::unpack(*(const uint4*)(rz + m * 2 * C + C + c0), z);
        TR::unpack(*(const uint4*)(cpre + m * C + c0), c);
        TR::unpack(*(const uint4*)(h + m * C + c0), hv);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const float zs = sigm(z[e]), tc = tanhf(c[e]);
            o1[e] = g[e] * (tc - hv[e]) * zs * (1.f - zs);
            o2[e] = g[e] * zs * (1.f - tc * tc);
            o3[e] = g[e] * (1.f - zs);
        }
        *(uint4*)(dz_pre + m * ld_dz + c0) = TR::pack(o1);
        *(uint4*)(dc_pre + m * C + c0) = TR::pack(o2);
        *(uint4*)(dh_part + m * C + c0) = TR::pack(o3);
    }
}

// alphas: (3, P, HW) fp32 frames t-1, t, t+1 (frame stride fs); prev: (P, HW) or NULL (= frame 0); df / db: forward / backward
// difference maps (3, P, HW) with the same strides. Writes frames 1 and 2 in place.
__global__ __launch_bounds__(NT) void temporal_fuse_kernel(float* __restrict__ alphas, const float* __restrict__ prev, const float* __restrict__ df,
                                                           const float* __restrict__ db, long fs, long n, long last) {
    // `last` = offset of the clip's final frame (alphas[:, -1] of maggie_temp.py:48): frame 2 for the 3-frame evaluation window
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float a0 = alphas[i], a1 = alphas[fs + i], a2 = alphas[last + i];
        const float pv = prev ? prev[i] : a0;
        const float f1 = df[fs + i] > 0.5f ? 1.f : 0.f, f2 = df[2 * fs + i] > 0.5f ? 1.f : 0.f, b1 = db[fs + i] > 0.5f ? 1.f : 0.f;
        float fwd = pv * (1.f - f1) + a1 * f1;                       // t-1 -> t
        const float bwd = a2 * (1.f - b1) + a1 * b1;                 // t+1 -> t
        if (fabsf(fwd - bwd) > 0.f) fwd = a1;                        // the two disagree: keep the model's own prediction
        alphas[fs + i] = fwd;
        alphas[2 * fs + i] = fwd * (1.f - f2) + a2 * f2;             // t -> t+1
    }
}

}  // namespace

static inline int gru_check(int dtype, int M, int C, long* total) {
    const int ce = MG_IS16(dtype) ? 8 : 4;
    if (!MG_IS16(dtype) && dtype != MG_F32) return -6;
    if (C % ce) return -3;
    *total = (long)M * (C / ce);
    return 0;
}

extern "C" int mg_gru_gate_fwd(const void* rz, const void* x, const void* h, int dtype, int M, int C, void* xrh, void* stream) {
    long total; int rc = gru_check(dtype, M, C, &total); if (rc) return rc;
    if (M <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gru_gate_fwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)rz, (const bf16raw*)x, (const bf16raw*)h, M, C, (bf16raw*)xrh);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gru_gate_fwd_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)rz, (const f16raw*)x, (const f16raw*)h, M, C, (f16raw*)xrh);
    else hipLaunchKernelGGL(gru_gate_fwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)rz, (const float*)x, (const float*)h, M, C, (float*)xrh);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_gru_gate_bwd(const void* dxrh, const void* rz, const void* h, int dtype, int M, int C, void* dx, void* drz, void* dh_part,
                               void* stream) {
    long total; int rc = gru_check(dtype, M, C, &total); if (rc) return rc;
    if (M <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gru_gate_bwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)dxrh, (const bf16raw*)rz, (const bf16raw*)h, M, C, (bf16raw*)dx, (bf16raw*)drz, 2 * C, (bf16raw*)dh_part);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gru_gate_bwd_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)dxrh, (const f16raw*)rz, (const f16raw*)h, M, C, (f16raw*)dx, (f16raw*)drz, 2 * C, (f16raw*)dh_part);
    else hipLaunchKernelGGL(gru_gate_bwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)dxrh, (const float*)rz, (const float*)h, M, C, (float*)dx, (float*)drz, 2 * C, (float*)dh_part);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_gru_out_fwd(const void* rz, const void* cpre, const void* h, int dtype, int M, int C, void* hn, void* stream) {
    long total; int rc = gru_check(dtype, M, C, &total); if (rc) return rc;
    if (M <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gru_out_fwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)rz, (const bf16raw*)cpre, (const bf16raw*)h, M, C, (bf16raw*)hn);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gru_out_fwd_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)rz, (const f16raw*)cpre, (const f16raw*)h, M, C, (f16raw*)hn);
    else hipLaunchKernelGGL(gru_out_fwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)rz, (const float*)cpre, (const float*)h, M, C, (float*)hn);
    MG_CHECK_LAUNCH();
    return 0;
}

// drz: (M, 2C); this call fills its z half [:, C:] (mg_gru_gate_bwd fills the r half)
extern "C" int mg_gru_out_bwd(const void* dhn, const void* rz, const void* cpre, const void* h, int dtype, int M, int C, void* drz, void* dc_pre,
                              void* dh_part, void* stream) {
    long total; int rc = gru_check(dtype, M, C, &total); if (rc) return rc;
    if (M <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gru_out_bwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)dhn, (const bf16raw*)rz, (const bf16raw*)cpre, (const bf16raw*)h, M, C, (bf16raw*)drz + C, 2 * C, (bf16raw*)dc_pre, (bf16raw*)dh_part);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gru_out_bwd_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)dhn, (const f16raw*)rz, (const f16raw*)cpre, (const f16raw*)h, M, C, (f16raw*)drz + C, 2 * C, (f16raw*)dc_pre, (f16raw*)dh_part);
    else hipLaunchKernelGGL(gru_out_bwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)dhn, (const float*)rz, (const float*)cpre, (const float*)h, M, C, (float*)drz + C, 2 * C, (float*)dc_pre, (float*)dh_part);
    MG_CHECK_LAUNCH();
    return 0;
}

// alphas (3, P, H*W) fp32 contiguous frames (t-1, t, t+1), updated in place (frames 1 and 2); prev (P, H*W) or NULL; df / db (3, P, H*W)
extern "C" int mg_temporal_fuse(float* alphas, const float* prev, const float* df, const float* db, long plane_elems, int n_frames, void* stream) {
    if (plane_elems <= 0) return 0;
    if (n_frames < 3) return -3;
    hipLaunchKernelGGL(temporal_fuse_kernel, dim3(grid_for(plane_elems)), dim3(NT), 0, (hipStream_t)stream, alphas, prev, df, db, plane_elems, plane_elems,
                       (long)(n_frames - 1) * plane_elems);
    MG_CHECK_LAUNCH();
    return 0;
}
