// Row kernels of the sparse refinement head with a DEVICE row count (gfx950, HBM-bound, 16-byte vector accesses).
//
// The feature matrices of the head are (capacity x C) buffers whose first *m_dev rows are live (sorted active sites). Everything
// between the gather convolutions of maggie/network/decoder/resnet_inst_matt_spconv.py:161-270 that the reference runs as torch
// ops on the spconv feature matrix -- `detail * sigmoid(guidance)` (:188-193), the FFN tail of `inst_spec_layer` (dropout, residual,
// LayerNorm: maggie/network/module/mask_attention.py:170-182), gradient accumulation of a tensor with two consumers -- is done
// here over min(*m_dev, cap) rows, grid-stride from a fixed grid, so that no host code depends on the count and the whole detail
// stage is a static launch sequence (hipGraph-capturable).
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;
inline int grid_for(long total) { long b = (total + NT - 1) / NT; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

// ---- out = a * sigmoid(g)  /  da = dout * s, dg = dout * a * s * (1 - s) --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void sigmul_fwd_kernel(const T* __restrict__ a, int lda, const T* __restrict__ g, T* __restrict__ out, int M, int C,
                                                        const int32_t* __restrict__ m_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)dev_rows(m_dev, M) * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int m = (int)(i / cpr), c0 = (int)(i - (long)m * cpr) * CE;
        float av[CE], gv[CE];
        TR::unpack(*(const uint4*)(a + (long)m * lda + c0), av);
        TR::unpack(*(const uint4*)(g + (long)m * C + c0), gv);
#pragma unroll
        for (int e = 0; e < CE; ++e) av[e] *= 1.f / (1.f + __expf(-gv[e]));
        *(uint4*)(out + (long)m * C + c0) = TR::pack(av);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void sigmul_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ a, int lda, const T* __restrict__ g,
                                                        T* __restrict__ da, T* __restrict__ dg, int M, int C, const int32_t* __restrict__ m_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)dev_rows(m_dev, M) * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int m = (int)(i / cpr), c0 = (int)(i - (long)m * cpr) * CE;
        float dv[CE], av[CE], gv[CE], o1[CE], o2[CE];
        TR::unpack(*(const uint4*)(dout + (long)m * C + c0), dv);
        TR::unpack(*(const uint4*)(a + (long)m * lda + c0), av);
        TR::unpack(*(const uint4*)(g + (long)m * C + c0), gv);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const float s = 1.f / (1.f + __expf(-gv[e]));
            o1[e] = dv[e] * s;
            o2[e] = dv[e] * av[e] * s * (1.f - s);
        }
        *(uint4*)(da + (long)m * C + c0) = TR::pack(o1);
        *(uint4*)(dg + (long)m * C + c0) = TR::pack(o2);
    }
}

// ---- out = a + b (gradient of a tensor with two consumers) ------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void rows_add_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, T* __restrict__ out, int ldo,
                                                      int M, int C, const int32_t* __restrict__ m_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)dev_rows(m_dev, M) * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int m = (int)(i / cpr), c0 = (int)(i - (long)m * cpr) * CE;
        float av[CE], bv[CE];
        TR::unpack(*(const uint4*)(a + (long)m * lda + c0), av);
        TR::unpack(*(const uint4*)(b + (long)m * ldb + c0), bv);
#pragma unroll
        for (int e = 0; e < CE; ++e) av[e] += bv[e];
        *(uint4*)(out + (long)m * ldo + c0) = TR::pack(av);
    }
}

// ---- dropout: y = x * keep / (1 - p), keep = [hash(seed, step, salt, element) >= p * 2^32] --------------------------------------
// Counter-based: the same (state, salt) reproduces the mask, so the backward pass recomputes it instead of storing it. `state` is a
// device int64[2] = (seed, step) snapshot taken by the forward (capturable: no host value is baked into a graph).
__device__ __forceinline__ uint32_t mix_hash(uint64_t seed, uint64_t step, uint32_t salt, uint64_t idx) {
    uint64_t z = seed ^ (step * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)salt << 56) ^ (idx * 0xD1B54A32D192ED03ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

template <typename T>
__global__ __launch_bounds__(NT) void rows_dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int M, int C, float p, const int64_t* __restrict__ state,
                                                          uint32_t salt, const int32_t* __restrict__ m_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const long total = (long)dev_rows(m_dev, M) * cpr;
    const uint64_t seed = (uint64_t)state[0], step = (uint64_t)state[1];
    const uint32_t thr = p >= 1.f ? 0xFFFFFFFFu : (uint32_t)((double)p * 4294967296.0);
    const float sc = p < 1.f ? 1.f / (1.f - p) : 0.f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int m = (int)(i / cpr), c0 = (int)(i - (long)m * cpr) * CE;
        float v[CE];
        TR::unpack(*(const uint4*)(x + (long)m * C + c0), v);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const uint32_t h = mix_hash(seed, step, salt, (uint64_t)m * C + c0 + e);
            v[e] = (p > 0.f && h < thr) ? 0.f : v[e] * sc;
        }
        *(uint4*)(y + (long)m * C + c0) = TR::pack(v);
    }
}

// ---- y = LayerNorm(x + r) * gamma + beta over the C channels of each row (post-norm residual of FFNLayer) -------------------------
// A row is held by LPR = C / CE consecutive lanes (a power of two <= 64); row statistics by butterflies over those lanes.
// rstat[m] = (mean, rstd) is kept for the backward.
template <typename T>
__global__ __launch_bounds__(NT) void add_layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ r, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, T* __restrict__ y, float* __restrict__ rstat,
                                                               int M, int C, const int32_t* __restrict__ m_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int lpr = C / CE;
    const int rows_per_block = NT / lpr;
    const int lane_c = threadIdx.x % lpr, lrow = threadIdx.x / lpr;
    const int Mv = dev_rows(m_dev, M);
    const int c0 = lane_c * CE;
    float gm[CE], bt[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e]; }
    const float inv_c = 1.f / (float)C;
    for (int m0 = blockIdx.x * rows_per_block; m0 < Mv; m0 += gridDim.x * rows_per_block) {      // block-uniform trip count
        const int m = m0 + lrow;
        const bool act = m < Mv;
        float v[CE], rv[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) { v[e] = 0.f; rv[e] = 0.f; }
        if (act) {
            TR::unpack(*(const uint4*)(x + (long)m * C + c0), v);
            TR::unpack(*(const uint4*)(r + (long)m * C + c0), rv);
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < CE; ++e) { v[e] += rv[e]; s += v[e]; }
        for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < CE; ++e) { const float d = v[e] - mean; q += d * d; }
        for (int o = 1; o < lpr; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q * inv_c + eps);
        if (act) {
#pragma unroll
            for (int e = 0; e < CE; ++e) v[e] = (v[e] - mean) * rstd * gm[e] + bt[e];
            *(uint4*)(y + (long)m * C + c0) = TR::pack(v);
            if (lane_c == 0) { rstat[2 * (long)m] = mean; rstat[2 * (long)m + 1] = rstd; }
        }
    }
}

// dz = rstd * (dy*gamma - mean_c(dy*gamma) - xhat * mean_c(dy*gamma*xhat)), z = x + r (dx = dr = dz); dgamma += dy * xhat, dbeta += dy
template <typename T>
__global__ __launch_bounds__(NT) void add_layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ r,
                                                               const float* __restrict__ gamma, const float* __restrict__ rstat, T* __restrict__ dz,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C,
                                                               const int32_t* __restrict__ m_dev, float* __restrict__ slots) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    __shared__ float sred[NT * 2 * 8];
    const int lpr = C / CE;
    const int rows_per_block = NT / lpr;
    const int lane_c = threadIdx.x % lpr, lrow = threadIdx.x / lpr;
    const int Mv = dev_rows(m_dev, M);
    const int c0 = lane_c * CE;
    float gm[CE], ag[CE], ab[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { gm[e] = gamma[c0 + e]; ag[e] = 0.f; ab[e] = 0.f; }
    const float inv_c = 1.f / (float)C;
    for (int m0 = blockIdx.x * rows_per_block; m0 < Mv; m0 += gridDim.x * rows_per_block) {
        const int m = m0 + lrow;
        const bool act = m < Mv;
        float d[CE], v[CE], rv[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) { d[e] = 0.f; v[e] = 0.f; rv[e] = 0.f; }
        float mean = 0.f, rstd = 0.f;
        if (act) {
            TR::unpack(*(const uint4*)(dy + (long)m * C + c0), d);
            TR::unpack(*(const uint4*)(x + (long)m * C + c0), v);
            TR::unpack(*(const uint4*)(r + (long)m * C + c0), rv);
            mean = rstat[2 * (long)m]; rstd = rstat[2 * (long)m + 1];
        }
        float s1 = 0.f, s2 = 0.f, xh[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            xh[e] = (v[e] + rv[e] - mean) * rstd;
            const float dg = d[e] * gm[e];
            s1 += dg; s2 += dg * xh[e];
            ag[e] += d[e] * xh[e]; ab[e] += d[e];
        }
        for (int o = 1; o < lpr; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (act) {
            float o_[CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) o_[e] = rstd * (d[e] * gm[e] - s1 * inv_c - xh[e] * s2 * inv_c);
            *(uint4*)(dz + (long)m * C + c0) = TR::pack(o_);
        }
    }
    // column sums over the block's row lanes, then one atomic per channel per block
#pragma unroll
    for (int e = 0; e < CE; ++e) { sred[(lrow * lpr + lane_c) * 2 * CE + e] = ag[e]; sred[(lrow * lpr + lane_c) * 2 * CE + CE + e] = ab[e]; }
    __syncthreads();
    for (int j = threadIdx.x; j < lpr * 2 * CE; j += NT) {
        float a = 0.f;
        for (int rr = 0; rr < rows_per_block; ++rr) a += sred[rr * lpr * 2 * CE + j];
        const int lc = j / (2 * CE), k = j - lc * 2 * CE;
        if (slots) slots[(size_t)blockIdx.x * 2 * C + (k < CE ? lc * CE + k : C + lc * CE + k - CE)] = a;    // deterministic mode: one row [dgamma | dbeta] per workgroup
        else if (k < CE) atomicAdd(&dgamma[lc * CE + k], a); else atomicAdd(&dbeta[lc * CE + k - CE], a);
    }
}

// ---- tiny device-side bookkeeping ----------------------------------------------------------------------------------------------
// the reference's "dummy code to prevent all zeros" (resnet_inst_matt_spconv.py:347-348): if the detail region is empty in training,
// force the square [y0:y1, x0:x1] of every plane -- decided on the device from the site count
__global__ void patch_if_empty_kernel(unsigned long long* __restrict__ bits, const int32_t* __restrict__ count, int P, int H, int Ww, int y0, int y1,
                                      int x0, int x1) {
    if (*count > 0) return;
    const long total = (long)P * (y1 - y0) * Ww;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int wj = (int)(i % Ww); long r = i / Ww; const int y = y0 + (int)(r % (y1 - y0)); const int p = (int)(r / (y1 - y0));
        unsigned long long m = 0ull;
        for (int b = 0; b < 64; ++b) { const int x = wj * 64 + b; if (x >= x0 && x < x1) m |= 1ull << b; }
        if (m) bits[((long)p * H + y) * Ww + wj] |= m;
    }
}

}  // namespace

static int rows_check(int dtype, int C) {
    if (!MG_IS16(dtype) && dtype != MG_F32) return -6;
    const int ce = MG_IS16(dtype) ? 8 : 4;
    if (C % ce) return -3;
    return 0;
}

extern "C" int mg_rows_sigmoid_mul_fwd(const void* a, int lda, const void* g, void* out, int dtype, int M, int C, const int32_t* m_dev, void* stream) {
    int rc = rows_check(dtype, C); if (rc) return rc;
    if (M <= 0) return 0;
    const long total = (long)M * (C / (MG_IS16(dtype) ? 8 : 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(sigmul_fwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)a, lda, (const bf16raw*)g, (bf16raw*)out, M, C, m_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL(sigmul_fwd_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)a, lda, (const f16raw*)g, (f16raw*)out, M, C, m_dev);
    else hipLaunchKernelGGL(sigmul_fwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)a, lda, (const float*)g, (float*)out, M, C, m_dev);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_rows_sigmoid_mul_bwd(const void* dout, const void* a, int lda, const void* g, void* da, void* dg, int dtype, int M, int C,
                                       const int32_t* m_dev, void* stream) {
    int rc = rows_check(dtype, C); if (rc) return rc;
    if (M <= 0) return 0;
    const long total = (long)M * (C / (MG_IS16(dtype) ? 8 : 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(sigmul_bwd_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)dout, (const bf16raw*)a, lda, (const bf16raw*)g, (bf16raw*)da, (bf16raw*)dg, M, C, m_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL(sigmul_bwd_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)dout, (const f16raw*)a, lda, (const f16raw*)g, (f16raw*)da, (f16raw*)dg, M, C, m_dev);
    else hipLaunchKernelGGL(sigmul_bwd_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)dout, (const float*)a, lda, (const float*)g, (float*)da, (float*)dg, M, C, m_dev);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_rows_add(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int dtype, int M, int C, const int32_t* m_dev, void* stream) {
    int rc = rows_check(dtype, C); if (rc) return rc;
    if (M <= 0) return 0;
    const long total = (long)M * (C / (MG_IS16(dtype) ? 8 : 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(rows_add_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)a, lda, (const bf16raw*)b, ldb, (bf16raw*)out, ldo, M, C, m_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL(rows_add_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)a, lda, (const f16raw*)b, ldb, (f16raw*)out, ldo, M, C, m_dev);
    else hipLaunchKernelGGL(rows_add_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)a, lda, (const float*)b, ldb, (float*)out, ldo, M, C, m_dev);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_rows_dropout(const void* x, void* y, int dtype, int M, int C, float p, const int64_t* state, int salt, const int32_t* m_dev, void* stream) {
    int rc = rows_check(dtype, C); if (rc) return rc;
    if (M <= 0) return 0;
    if (!state || p < 0.f || p > 1.f) return -2;
    const long total = (long)M * (C / (MG_IS16(dtype) ? 8 : 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(rows_dropout_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)x, (bf16raw*)y, M, C, p, state, (uint32_t)salt, m_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL(rows_dropout_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)x, (f16raw*)y, M, C, p, state, (uint32_t)salt, m_dev);
    else hipLaunchKernelGGL(rows_dropout_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)x, (float*)y, M, C, p, state, (uint32_t)salt, m_dev);
    MG_CHECK_LAUNCH();
    return 0;
}

static int ln_check(int dtype, int C) {
    int rc = rows_check(dtype, C); if (rc) return rc;
    const int lpr = C / (MG_IS16(dtype) ? 8 : 4);
    if (lpr < 1 || lpr > 64 || (lpr & (lpr - 1))) return -3;          // a row = a power-of-two group of lanes
    return 0;
}

extern "C" int mg_rows_add_layernorm_fwd(const void* x, const void* r, const float* gamma, const float* beta, float eps, void* y, float* rstat, int dtype,
                                         int M, int C, const int32_t* m_dev, void* stream) {
    int rc = ln_check(dtype, C); if (rc) return rc;
    if (M <= 0) return 0;
    const int lpr = C / (MG_IS16(dtype) ? 8 : 4), rpb = NT / lpr;
    long blocks = ((long)M + rpb - 1) / rpb; if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(add_layernorm_fwd_kernel<bf16raw>, dim3((unsigned)blocks), dim3(NT), 0, st, (const bf16raw*)x, (const bf16raw*)r, gamma, beta, eps, (bf16raw*)y, rstat, M, C, m_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL(add_layernorm_fwd_kernel<f16raw>, dim3((unsigned)blocks), dim3(NT), 0, st, (const f16raw*)x, (const f16raw*)r, gamma, beta, eps, (f16raw*)y, rstat, M, C, m_dev);
    else hipLaunchKernelGGL(add_layernorm_fwd_kernel<float>, dim3((unsigned)blocks), dim3(NT), 0, st, (const float*)x, (const float*)r, gamma, beta, eps, (float*)y, rstat, M, C, m_dev);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_rows_add_layernorm_bwd(const void* dy, const void* x, const void* r, const float* gamma, const float* rstat, void* dz, float* dgamma,
                                         float* dbeta, int dtype, int M, int C, const int32_t* m_dev, void* stream) {
    int rc = ln_check(dtype, C); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    { hipError_t e = mg_zero_words(dgamma, C, st); if (e != hipSuccess) return (int)e; e = mg_zero_words(dbeta, C, st); if (e != hipSuccess) return (int)e; }
    if (M <= 0) return 0;
    const int lpr = C / (MG_IS16(dtype) ? 8 : 4), rpb = NT / lpr;
    long blocks = ((long)M + rpb - 1) / rpb; if (blocks > 512) blocks = 512;
    float* slots = nullptr;
    if (mg_det_on) { slots = mg_det_scratch(blocks * 2 * C); if (!slots) return MG_DET_NO_SCRATCH; }
    if (dtype == MG_BF16) hipLaunchKernelGGL(add_layernorm_bwd_kernel<bf16raw>, dim3((unsigned)blocks), dim3(NT), 0, st, (const bf16raw*)dy, (const bf16raw*)x, (const bf16raw*)r, gamma, rstat, (bf16raw*)dz, dgamma, dbeta, M, C, m_dev, slots);
    else if (dtype == MG_F16) hipLaunchKernelGGL(add_layernorm_bwd_kernel<f16raw>, dim3((unsigned)blocks), dim3(NT), 0, st, (const f16raw*)dy, (const f16raw*)x, (const f16raw*)r, gamma, rstat, (f16raw*)dz, dgamma, dbeta, M, C, m_dev, slots);
    else hipLaunchKernelGGL(add_layernorm_bwd_kernel<float>, dim3((unsigned)blocks), dim3(NT), 0, st, (const float*)dy, (const float*)x, (const float*)r, gamma, rstat, (float*)dz, dgamma, dbeta, M, C, m_dev, slots);
    MG_CHECK_LAUNCH();
    if (slots) {
        mg_det_seg sg[2] = {{dgamma, C, 0}, {dbeta, C, 0}};
        return mg_det_reduce(slots, (int)blocks, 1, 2 * C, 0, sg, 2, st);
    }
    return 0;
}

extern "C" int mg_bits_patch_if_empty(void* bits, const int32_t* count, int P, int H, int W, int y0, int y1, int x0, int x1, void* stream) {
    if (P <= 0 || y1 <= y0 || x1 <= x0 || y1 > H || x1 > W || y0 < 0 || x0 < 0) return -2;
    const int Ww = (W + 63) / 64;
    const long total = (long)P * (y1 - y0) * Ww;
    long blocks = (total + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(patch_if_empty_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)bits, count, P, H, Ww, y0, y1, x0, x1);
    MG_CHECK_LAUNCH();
    return 0;
}
