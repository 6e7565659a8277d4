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
// Fixed-order block sum of one value per thread (wave butterfly, then the four waves in order): the building block of every norm / dot
// product below -- the result depends on the thread -> element mapping only, never on timing.
__device__ __forceinline__ float block_sum_ordered(float v, float* sh4) {
    v = wave_sum(v);
    __syncthreads();                                              // sh4 may still be read from a previous call
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

// t[j] = sum_i W[i, j] u[i]: a workgroup owns 32 columns over ALL rows -- thread (column c = t % 32, row lane r = t / 32) adds rows r, r + 8, ...
// in order, the 8 row lanes meet in LDS in order. (The earlier form split the rows over workgroups and added their partials atomically.)
__device__ __forceinline__ void wt_u_columns(const float* __restrict__ W, const float* __restrict__ u, int A, int Wd, int j0, float* __restrict__ t,
                                             float* sh /* [8][33] */) {
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int j = j0 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < Wd) {
        int i = r;
        for (; i + 24 < A; i += 32) {                             // four independent loads in flight, added in row order per accumulator
            a0 += W[(long)i * Wd + j] * u[i];
            a1 += W[(long)(i + 8) * Wd + j] * u[i + 8];
            a2 += W[(long)(i + 16) * Wd + j] * u[i + 16];
            a3 += W[(long)(i + 24) * Wd + j] * u[i + 24];
        }
        for (; i < A; i += 8) a0 += W[(long)i * Wd + j] * u[i];
    }
    sh[r * 33 + c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (r == 0 && j < Wd) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += sh[k * 33 + c];
        t[j] = v;
    }
}

__global__ __launch_bounds__(NT) void sn_wt_u_kernel(const float* __restrict__ W, const float* __restrict__ u, int A, int Wd, float* __restrict__ t) {
    __shared__ float sh[8 * 33];
    wt_u_columns(W, u, A, Wd, blockIdx.x * 32, t, sh);
}

// scratch[0] = |t|^2, scratch[1] = |s'|^2 in a fixed order (one workgroup)
__device__ __forceinline__ void sn_norms(const float* __restrict__ t, int Wd, const float* __restrict__ s, int A, float* __restrict__ scratch, float* sh4) {
    float a = 0.f;
    for (int j = threadIdx.x; j < Wd; j += NT) a += t[j] * t[j];
    a = block_sum_ordered(a, sh4);
    float b = 0.f;
    for (int i = threadIdx.x; i < A; i += NT) b += s[i] * s[i];
    b = block_sum_ordered(b, sh4);
    if (threadIdx.x == 0) { scratch[0] = a; scratch[1] = b; }
}
__global__ __launch_bounds__(NT) void sn_norms_kernel(const float* __restrict__ t, int Wd, const float* __restrict__ s, int A, float* __restrict__ scratch) {
    __shared__ float sh4[4];
    sn_norms(t, Wd, s, A, scratch, sh4);
}

// one wave per row: s'[i] = sum_j W[i,j] t[j]   (the two norms follow in sn_norms_kernel: fixed order, no atomics)
__global__ __launch_bounds__(NT) void sn_w_t_kernel(const float* __restrict__ W, const float* __restrict__ t, int A, int Wd,
                                                    float* __restrict__ s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i < A) {
        float acc = 0.f;
        for (int j = lane; j < Wd; j += 64) acc += W[(long)i * Wd + j] * t[j];
        acc = wave_sum(acc);
        if (lane == 0) s[i] = acc;
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

// <G, W> with G in (Cout, taps, pad_in) fp32 layout: one partial per workgroup into `part`, summed in order by sn_bwd_dot_finish_kernel
__global__ __launch_bounds__(NT) void sn_bwd_dot_kernel(const float* __restrict__ G, const float* __restrict__ W, int A, int B, int taps,
                                                        int transposed, int pad_in, float* __restrict__ part) {
    const long total = (long)A * B * taps;
    float acc = 0.f;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        int tap = (int)(e % taps); long r = e / taps; int b = (int)(r % B); int a = (int)(r / B);
        int co = transposed ? b : a, ci = transposed ? a : b;
        acc += G[((long)co * taps + tap) * pad_in + ci] * W[e];
    }
    __shared__ float sh[NT / 64];
    acc = block_sum_ordered(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(NT) void sn_bwd_dot_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ scratch) {
    __shared__ float sh[NT / 64];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) a += part[i];
    a = block_sum_ordered(a, sh);
    if (threadIdx.x == 0) scratch[2] = a;
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
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3((Wd + 31) / 32), dim3(NT), 0, st, W, u, A, Wd, t);
    hipLaunchKernelGGL(sn_w_t_kernel, dim3((A + 3) / 4), dim3(NT), 0, st, W, t, A, Wd, s);
    hipLaunchKernelGGL(sn_norms_kernel, dim3(1), dim3(NT), 0, st, (const float*)t, Wd, (const float*)s, A, scratch);
    const int Cout = transposed ? B : A;
    long total = (long)Cout * taps * pad_in;
    long need = total > Wd ? total : Wd;
    if (need < A) need = A;
    int blocks = grid_for(need);
    if ((long)blocks * NT < (Wd > A ? Wd : A)) blocks = (int)(((Wd > A ? Wd : A) + NT - 1) / NT);
    if (out_dtype == MG_BF16) hipLaunchKernelGGL(sn_finish_kernel<bf16raw>, dim3(blocks), dim3(NT), 0, st, W, t, s, scratch, A, B, taps, transposed, pad_in, u, v, (bf16raw*)out);
    else if (out_dtype == MG_F16) hipLaunchKernelGGL(sn_finish_kernel<f16raw>, dim3(blocks), dim3(NT), 0, st, W, t, s, scratch, A, B, taps, transposed, pad_in, u, v, (f16raw*)out);
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
    long total = (long)A * B * taps;
    int dot_blocks = grid_for(total / 8); if (dot_blocks > 256) dot_blocks = 256;
    // the per-workgroup partials of <G, W> land in dW (overwritten by the apply kernel afterwards; total >= 256 floats whenever 256 blocks run)
    if (dot_blocks > total) dot_blocks = (int)total;
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(dot_blocks), dim3(NT), 0, st, G, W, A, B, taps, transposed, pad_in, dW);
    hipLaunchKernelGGL(sn_bwd_dot_finish_kernel, dim3(1), dim3(NT), 0, st, (const float*)dW, dot_blocks, scratch);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(grid_for(total)), dim3(NT), 0, st, G, u, v, scratch, A, B, taps, transposed, pad_in, dW);
    MG_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================================================
// Batched variant: ALL spectrally-normalised convolutions of a model in a handful of launches (the image model wraps 54
// convs; per-conv launches are ~11 tiny kernels/memsets each). A descriptor table (device memory, built once) holds the
// parameter pointers and geometry; per-forward buffers (normalised weights, work) are passed as bases + per-conv offsets.
// =====================================================================================================================
namespace {

__global__ __launch_bounds__(NT) void snb_wt_u_kernel(const mg_sn_desc* __restrict__ descs, const int4* __restrict__ items, float* __restrict__ work_base) {
    __shared__ float sh[8 * 33];
    const int4 it = items[blockIdx.x];                       // (conv, block of 32 columns, -, -)
    const mg_sn_desc d = descs[it.x];
    wt_u_columns(d.W, d.u, d.A, d.B * d.taps, it.y * 32, work_base + d.work_off, sh);
}

__global__ __launch_bounds__(NT) void snb_w_t_kernel(const mg_sn_desc* __restrict__ descs, const int4* __restrict__ items, float* __restrict__ work_base) {
    const int4 it = items[blockIdx.x];                       // (conv, row group of 4, -, -)
    const mg_sn_desc d = descs[it.x];
    const int Wd = d.B * d.taps;
    const float* t = work_base + d.work_off;
    float* s = work_base + d.work_off + Wd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = it.y * 4 + wave;
    if (i < d.A) {
        float acc = 0.f;
        for (int j = lane; j < Wd; j += 64) acc += d.W[(long)i * Wd + j] * t[j];
        acc = wave_sum(acc);
        if (lane == 0) s[i] = acc;
    }
}

// |t|^2 and |W t|^2 of every conv in a fixed order: one workgroup per conv (was: atomics from every row's wave)
__global__ __launch_bounds__(NT) void snb_norms_kernel(const mg_sn_desc* __restrict__ descs, int n, float* __restrict__ work_base) {
    __shared__ float sh4[4];
    const int c = blockIdx.x;
    if (c >= n) return;
    const mg_sn_desc d = descs[c];
    if (d.plain) return;
    const int Wd = d.B * d.taps;
    float* t = work_base + d.work_off;
    sn_norms(t, Wd, t + Wd, d.A, t + Wd + d.A, sh4);
}

// ---- tiled passes over a parameter W[A][B][taps] (OIHW / IOHW, fp32) ------------------------------------------------------------
// The conv kernels want (Cout, taps, Cin_pad) [and (Cin_pad, taps, Cout) for dgrad]: a transposition of the two innermost
// index groups. A work item is a (SN_TA x SN_TB) tile of (a, b): the parameter side is always touched in its own order
// (SN_TB*taps contiguous floats per row), the KRSC side in ITS order (contiguous channel runs), and the exchange goes through an
// LDS tile -- every global access is a coalesced run instead of a 2-byte gather at stride taps (the element-per-thread version ran
// at ~1/9 of HBM speed).
constexpr int SN_TA = 16, SN_TB = 32, SN_MAXTAPS = 16;
constexpr int SN_PITCH = SN_TB * SN_MAXTAPS + 1;

struct SnTile { int a0, b0, na, nb; };
__device__ __forceinline__ SnTile sn_tile(const mg_sn_desc& d, const int4& it) {
    SnTile t;
    t.a0 = it.y * SN_TA; t.b0 = it.z * SN_TB;
    t.na = min(SN_TA, d.A - t.a0); t.nb = min(SN_TB, d.B - t.b0);
    return t;
}
// sT[al][bl * taps + tap] = W[a0 + al][b0 + bl][tap] * mul
__device__ __forceinline__ void sn_load_param_tile(const mg_sn_desc& d, const SnTile& t, float mul, float* sT) {
    const int run = t.nb * d.taps;
    if ((run & 3) == 0 && (((long)d.B * d.taps) & 3) == 0 && ((t.b0 * d.taps) & 3) == 0 && (((size_t)d.W) & 15) == 0) {
        const int r4 = run >> 2;                                  // 16-byte loads: a parameter row of the tile is one contiguous run
        for (int i = threadIdx.x; i < t.na * r4; i += NT) {
            const int al = i / r4, q = i - al * r4;
            const float4 v = *(const float4*)(d.W + ((long)(t.a0 + al) * d.B + t.b0) * d.taps + 4 * q);
            float* dst = sT + al * SN_PITCH + 4 * q;
            dst[0] = v.x * mul; dst[1] = v.y * mul; dst[2] = v.z * mul; dst[3] = v.w * mul;
        }
        return;
    }
    for (int i = threadIdx.x; i < t.na * run; i += NT) {
        const int al = i / run, r = i - al * run;
        sT[al * SN_PITCH + r] = d.W[((long)(t.a0 + al) * d.B + t.b0) * d.taps + r] * mul;
    }
}

// dst[row(i) * ld + c0 + (chunk of CE channels)] = pack(CE values gathered from the LDS tile by `val(row, channel)`), rows x nch channels, nch % CE == 0
// and every row start 16-byte aligned: one 16-byte store per CE channels instead of one 2-byte store per element
template <typename T, typename RowOff, typename Val>
__device__ __forceinline__ void sn_store_rows(T* __restrict__ dst, int rows, int nch, RowOff rowoff, Val val) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = nch / CE;
    for (int i = threadIdx.x; i < rows * cpr; i += NT) {
        const int cc = i % cpr, r = i / cpr;
        float v[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] = val(r, cc * CE + e);
        *(uint4*)(dst + rowoff(r) + cc * CE) = TR::pack(v);
    }
}
template <typename T> __device__ __forceinline__ bool sn_vec_ok(const void* base, long ld, int nch, int c0) {
    constexpr int CE = ElemTraits<T>::CE;
    return (nch % CE) == 0 && (ld % CE) == 0 && (c0 % CE) == 0 && (((size_t)base) & 15) == 0;
}

// emits W / sigma in the conv layouts; the vectors are finalised by snb_vectors_kernel (no block reads t/s after they changed)
template <typename T>
__global__ __launch_bounds__(NT) void snb_finish_kernel(const mg_sn_desc* __restrict__ descs, const int4* __restrict__ items, float* __restrict__ work_base,
                                                        T* __restrict__ out_base, T* __restrict__ out_t_base) {
    __shared__ float sT[SN_TA * SN_PITCH];
    const int4 it = items[blockIdx.x];                       // (conv, a tile, b tile, -)
    const mg_sn_desc d = descs[it.x];
    const int Wd = d.B * d.taps;
    const float* scratch = work_base + d.work_off + Wd + d.A;
    float sigma = 1.f;                                       // d.plain: an ordinary conv weight riding along for the layout conversion only
    if (!d.plain) {
        const float nt = sqrtf(scratch[0]);
        const float sv = 1.f / (nt + SN_EPS);
        const float ns = sqrtf(scratch[1]) * sv;
        sigma = ns * ns / (ns + SN_EPS);
    }
    const SnTile t = sn_tile(d, it);
    sn_load_param_tile(d, t, 1.f / sigma, sT);
    __syncthreads();
    const int taps = d.taps;
    const int Cout = d.transposed ? d.B : d.A, Cin = d.transposed ? d.A : d.B;
    T* out = out_base + d.out_off;
    if (!d.transposed) {
        // out[co = a][tap][ci = b]: runs of nb channels; the last b tile also zero-fills the channel padding [Cin, pad_in)
        const int nb_w = (t.b0 + t.nb == d.B) ? (d.pad_in - t.b0) : t.nb;
        if (sn_vec_ok<T>(out, d.pad_in, nb_w, t.b0)) {
            sn_store_rows<T>(out, t.na * taps, nb_w, [&](int r) { return ((long)(t.a0 + r / taps) * taps + r % taps) * d.pad_in + t.b0; },
                             [&](int r, int bl) { return bl < t.nb ? sT[(r / taps) * SN_PITCH + bl * taps + r % taps] : 0.f; });
        } else {
            for (int i = threadIdx.x; i < t.na * taps * nb_w; i += NT) {
                const int bl = i % nb_w, r = i / nb_w, tap = r % taps, al = r / taps;
                const float v = bl < t.nb ? sT[al * SN_PITCH + bl * taps + tap] : 0.f;
                ElemTraits<T>::st(out + ((long)(t.a0 + al) * taps + tap) * d.pad_in + t.b0 + bl, v);
            }
        }
        if (out_t_base) {                                    // out_t[ci = b][tap][co = a]: runs of na output channels
            T* ot = out_t_base + d.out_off;
            if (sn_vec_ok<T>(ot, Cout, t.na, t.a0)) {
                sn_store_rows<T>(ot, t.nb * taps, t.na, [&](int r) { return ((long)(t.b0 + r / taps) * taps + r % taps) * Cout + t.a0; },
                                 [&](int r, int al) { return sT[al * SN_PITCH + (r / taps) * taps + r % taps]; });
            } else {
                for (int i = threadIdx.x; i < t.nb * taps * t.na; i += NT) {
                    const int al = i % t.na, r = i / t.na, tap = r % taps, bl = r / taps;
                    ElemTraits<T>::st(ot + ((long)(t.b0 + bl) * taps + tap) * Cout + t.a0 + al, sT[al * SN_PITCH + bl * taps + tap]);
                }
            }
            if (t.b0 + t.nb == d.B) {                        // padded input channels of the twin: zero rows
                const int extra = d.pad_in - d.B;
                for (int i = threadIdx.x; i < extra * taps * t.na; i += NT) {
                    const int al = i % t.na, r = i / t.na;
                    ElemTraits<T>::st(ot + ((long)d.B * taps + r) * Cout + t.a0 + al, 0.f);
                }
            }
        }
    } else {
        // ConvTranspose parameter [Cin = A][Cout = B][taps] -> out[co = b][tap][ci = a]: runs of na input channels
        const int na_w = (t.a0 + t.na == d.A) ? (d.pad_in - t.a0) : t.na;
        if (sn_vec_ok<T>(out, d.pad_in, na_w, t.a0)) {
            sn_store_rows<T>(out, t.nb * taps, na_w, [&](int r) { return ((long)(t.b0 + r / taps) * taps + r % taps) * d.pad_in + t.a0; },
                             [&](int r, int al) { return al < t.na ? sT[al * SN_PITCH + (r / taps) * taps + r % taps] : 0.f; });
        } else {
            for (int i = threadIdx.x; i < t.nb * taps * na_w; i += NT) {
                const int al = i % na_w, r = i / na_w, tap = r % taps, bl = r / taps;
                const float v = al < t.na ? sT[al * SN_PITCH + bl * taps + tap] : 0.f;
                ElemTraits<T>::st(out + ((long)(t.b0 + bl) * taps + tap) * d.pad_in + t.a0 + al, v);
            }
        }
        if (out_t_base) {
            // the twin the data-gradient convolution of a ConvTranspose reads: out_t[ci = a][tap][co = b], the parameter's own index order with the
            // taps moved in front of the output channels -- runs of nb output channels (round 5: was a strided permute + copy per step and layer)
            T* ot = out_t_base + d.out_off;
            if (sn_vec_ok<T>(ot, Cout, t.nb, t.b0)) {
                sn_store_rows<T>(ot, t.na * taps, t.nb, [&](int r) { return ((long)(t.a0 + r / taps) * taps + r % taps) * Cout + t.b0; },
                                 [&](int r, int bl) { return sT[(r / taps) * SN_PITCH + bl * taps + r % taps]; });
            } else {
                for (int i = threadIdx.x; i < t.na * taps * t.nb; i += NT) {
                    const int bl = i % t.nb, r = i / t.nb, tap = r % taps, al = r / taps;
                    ElemTraits<T>::st(ot + ((long)(t.a0 + al) * taps + tap) * Cout + t.b0 + bl, sT[al * SN_PITCH + bl * taps + tap]);
                }
            }
            if (t.a0 + t.na == d.A) {                        // padded input channels of the twin: zero rows
                const int extra = d.pad_in - d.A;
                for (int i = threadIdx.x; i < extra * taps * t.nb; i += NT) {
                    const int bl = i % t.nb, r = i / t.nb;
                    ElemTraits<T>::st(ot + ((long)d.A * taps + r) * Cout + t.b0 + bl, 0.f);
                }
            }
        }
    }
    (void)Cin;
}

__global__ __launch_bounds__(NT) void snb_vectors_kernel(const mg_sn_desc* __restrict__ descs, int n, float* __restrict__ work_base) {
    const int c = blockIdx.x;
    if (c >= n) return;
    const mg_sn_desc d = descs[c];
    if (d.plain) return;
    const int Wd = d.B * d.taps;
    float* t = work_base + d.work_off;
    float* s = t + Wd;
    float* scratch = s + d.A;
    const float nt = sqrtf(scratch[0]);
    const float sv = 1.f / (nt + SN_EPS);
    const float ns = sqrtf(scratch[1]) * sv;
    const float su = sv / (ns + SN_EPS);
    for (int j = threadIdx.x; j < Wd; j += NT) { float v = t[j] * sv; t[j] = v; d.v[j] = v; }
    for (int i = threadIdx.x; i < d.A; i += NT) { float u = s[i] * su; s[i] = u; d.u[i] = u; }
    if (threadIdx.x == 0) scratch[3] = ns * ns / (ns + SN_EPS);
}

// KRSC-side gradient G[co][tap][ci_pad] of a tile -> sT[al][bl * taps + tap] (read in G's own order: channel runs)
// `twin` (ConvTranspose weights only): G is laid out like the twin, G[ci = a][tap][co = b] -- what the role-swapped weight-gradient GEMM of a
// transposed convolution writes; read in runs of output channels (was: a strided permute + copy into the (Cout, taps, Cin) layout per step)
template <typename T>
__device__ __forceinline__ void sn_load_grad_tile(const mg_sn_desc& d, const SnTile& t, const T* __restrict__ G, float* sT, bool twin = false) {
    const int taps = d.taps;
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    if (twin) {
        if (sn_vec_ok<T>(G, d.B, t.nb, t.b0)) {
            const int cpr = t.nb / CE;
            for (int i = threadIdx.x; i < t.na * taps * cpr; i += NT) {
                const int cc = i % cpr, r = i / cpr, tap = r % taps, al = r / taps;
                float v[CE];
                TR::unpack(*(const uint4*)(G + ((long)(t.a0 + al) * taps + tap) * d.B + t.b0 + cc * CE), v);
#pragma unroll
                for (int e = 0; e < CE; ++e) sT[al * SN_PITCH + (cc * CE + e) * taps + tap] = v[e];
            }
            return;
        }
        for (int i = threadIdx.x; i < t.na * taps * t.nb; i += NT) {
            const int bl = i % t.nb, r = i / t.nb, tap = r % taps, al = r / taps;
            sT[al * SN_PITCH + bl * taps + tap] = ElemTraits<T>::ld(G + ((long)(t.a0 + al) * taps + tap) * d.B + t.b0 + bl);
        }
        return;
    }
    if (!d.transposed) {
        if (sn_vec_ok<T>(G, d.pad_in, t.nb, t.b0)) {             // 16-byte loads along the channel runs
            const int cpr = t.nb / CE;
            for (int i = threadIdx.x; i < t.na * taps * cpr; i += NT) {
                const int cc = i % cpr, r = i / cpr, tap = r % taps, al = r / taps;
                float v[CE];
                TR::unpack(*(const uint4*)(G + ((long)(t.a0 + al) * taps + tap) * d.pad_in + t.b0 + cc * CE), v);
#pragma unroll
                for (int e = 0; e < CE; ++e) sT[al * SN_PITCH + (cc * CE + e) * taps + tap] = v[e];
            }
            return;
        }
        for (int i = threadIdx.x; i < t.na * taps * t.nb; i += NT) {
            const int bl = i % t.nb, r = i / t.nb, tap = r % taps, al = r / taps;
            sT[al * SN_PITCH + bl * taps + tap] = ElemTraits<T>::ld(G + ((long)(t.a0 + al) * taps + tap) * d.pad_in + t.b0 + bl);
        }
    } else {
        if (sn_vec_ok<T>(G, d.pad_in, t.na, t.a0)) {
            const int cpr = t.na / CE;
            for (int i = threadIdx.x; i < t.nb * taps * cpr; i += NT) {
                const int cc = i % cpr, r = i / cpr, tap = r % taps, bl = r / taps;
                float v[CE];
                TR::unpack(*(const uint4*)(G + ((long)(t.b0 + bl) * taps + tap) * d.pad_in + t.a0 + cc * CE), v);
#pragma unroll
                for (int e = 0; e < CE; ++e) sT[(cc * CE + e) * SN_PITCH + bl * taps + tap] = v[e];
            }
            return;
        }
        for (int i = threadIdx.x; i < t.nb * taps * t.na; i += NT) {
            const int al = i % t.na, r = i / t.na, tap = r % taps, bl = r / taps;
            sT[al * SN_PITCH + bl * taps + tap] = ElemTraits<T>::ld(G + ((long)(t.b0 + bl) * taps + tap) * d.pad_in + t.a0 + al);
        }
    }
}

// dot_part[item] = <G, W> over the tile; the apply kernel adds a conv's tiles in item order (was: one atomicAdd per tile onto scratch[2])
template <typename T>
__global__ __launch_bounds__(NT) void snb_bwd_dot_kernel(const mg_sn_desc* __restrict__ descs, const int4* __restrict__ items,
                                                         const void* const* __restrict__ Gptrs, float* __restrict__ dot_part) {
    __shared__ float sT[SN_TA * SN_PITCH];
    __shared__ float sh[NT / 64];
    const int4 it = items[blockIdx.x];                       // (conv, a tile, b tile, -)
    const mg_sn_desc d = descs[it.x];
    const size_t graw = (size_t)Gptrs[it.x];                 // bit 0: twin layout (ConvTranspose weights, see sn_load_grad_tile)
    const T* G = (const T*)(graw & ~(size_t)1);
    if (!G || d.plain) { if (threadIdx.x == 0) dot_part[blockIdx.x] = 0.f; return; }
    const SnTile t = sn_tile(d, it);
    sn_load_grad_tile<T>(d, t, G, sT, (graw & 1) && d.transposed);
    __syncthreads();
    float acc = 0.f;
    const int run = t.nb * d.taps;
    for (int i = threadIdx.x; i < t.na * run; i += NT) {
        const int al = i / run, r = i - al * run;
        acc += sT[al * SN_PITCH + r] * d.W[((long)(t.a0 + al) * d.B + t.b0) * d.taps + r];
    }
    acc = block_sum_ordered(acc, sh);
    if (threadIdx.x == 0) dot_part[blockIdx.x] = acc;
}

// dW[a][b][tap] = G / sigma - <G,W> / sigma^2 * u[a] v[b, tap]   (parameter layout, fp32)
template <typename T>
__global__ __launch_bounds__(NT) void snb_bwd_apply_kernel(const mg_sn_desc* __restrict__ descs, const int4* __restrict__ items,
                                                           const void* const* __restrict__ Gptrs, const float* __restrict__ work_base,
                                                           float* __restrict__ dW_base, float* const* __restrict__ dWptrs,
                                                           const float* __restrict__ dot_part) {
    __shared__ float sT[SN_TA * SN_PITCH];
    __shared__ float sh4[4];
    const int4 it = items[blockIdx.x];
    const mg_sn_desc d = descs[it.x];
    const size_t graw = (size_t)Gptrs[it.x];
    const T* G = (const T*)(graw & ~(size_t)1);
    const int Wd = d.B * d.taps;
    const float* v = work_base + d.work_off;
    const float* u = v + Wd;
    const float* scratch = u + d.A;
    float* dW = dW_base + d.dw_off;
    if (dWptrs && dWptrs[it.x]) dW = dWptrs[it.x];           // the caller's own destination (the optimizer's flat gradient buffer)
    const SnTile t = sn_tile(d, it);
    const int run = t.nb * d.taps;
    if (!G) {
        for (int i = threadIdx.x; i < t.na * run; i += NT) {
            const int al = i / run, r = i - al * run;
            dW[((long)(t.a0 + al) * d.B + t.b0) * d.taps + r] = 0.f;
        }
        return;
    }
    sn_load_grad_tile<T>(d, t, G, sT, (graw & 1) && d.transposed);
    __syncthreads();
    if (d.plain) {                                           // plain conv: the gradient itself, back in the parameter's layout
        for (int i = threadIdx.x; i < t.na * run; i += NT) {
            const int al = i / run, r = i - al * run;
            dW[((long)(t.a0 + al) * d.B + t.b0) * d.taps + r] = sT[al * SN_PITCH + r];
        }
        return;
    }
    // <G, W> of the conv = its tiles' partials, items [k3_first, k3_first + k3_count), added in item order by every workgroup of the conv
    float dot = 0.f;
    for (int i = threadIdx.x; i < d.k3_count; i += NT) dot += dot_part[d.k3_first + i];
    dot = block_sum_ordered(dot, sh4);
    const float inv_sigma = 1.f / scratch[3];
    const float coef = dot * inv_sigma * inv_sigma;
    for (int i = threadIdx.x; i < t.na * run; i += NT) {
        const int al = i / run, r = i - al * run;
        dW[((long)(t.a0 + al) * d.B + t.b0) * d.taps + r] = sT[al * SN_PITCH + r] * inv_sigma - coef * u[t.a0 + al] * v[t.b0 * d.taps + r];
    }
}

}  // namespace

// items_*: int32[n][4] work lists built by the host (see maggie_amd/functional.py: SpectralNormPlan)
extern "C" int mg_spectral_norm_batched(const mg_sn_desc* descs, int n_conv, const int32_t* items_k1, int n1, const int32_t* items_k2, int n2,
                                        const int32_t* items_k3, int n3, float* work_base, long work_floats, void* out_base, void* out_t_base,
                                        int out_dtype, void* stream) {
    if (n_conv <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    (void)work_floats;                                       // every word of the work block that is read has been written by the launches below
    if (n1 > 0) hipLaunchKernelGGL(snb_wt_u_kernel, dim3(n1), dim3(NT), 0, st, descs, (const int4*)items_k1, work_base);
    if (n2 > 0) hipLaunchKernelGGL(snb_w_t_kernel, dim3(n2), dim3(NT), 0, st, descs, (const int4*)items_k2, work_base);
    hipLaunchKernelGGL(snb_norms_kernel, dim3(n_conv), dim3(NT), 0, st, descs, n_conv, work_base);
    if (out_dtype == MG_BF16) hipLaunchKernelGGL(snb_finish_kernel<bf16raw>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, work_base, (bf16raw*)out_base, (bf16raw*)out_t_base);
    else if (out_dtype == MG_F16) hipLaunchKernelGGL(snb_finish_kernel<f16raw>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, work_base, (f16raw*)out_base, (f16raw*)out_t_base);
    else hipLaunchKernelGGL(snb_finish_kernel<float>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, work_base, (float*)out_base, (float*)out_t_base);
    hipLaunchKernelGGL(snb_vectors_kernel, dim3(n_conv), dim3(NT), 0, st, descs, n_conv, work_base);
    MG_CHECK_LAUNCH();
    return 0;
}

// Gptrs: device array of n_conv pointers to the weight gradients in the (Cout, taps, pad_in) layout (NULL = no gradient; bit 0 set on a
// ConvTranspose weight's entry: the gradient is in the twin's (Cin_pad, taps, Cout) layout); work_base: the forward's work buffer;
// dW_base: fp32 output, conv c at descs[c].dw_off laid out like the parameter -- unless dWptrs (device array of n_conv float*, or NULL) names
// another destination for it.
extern "C" int mg_spectral_norm_batched_bwd_to(const mg_sn_desc* descs, int n_conv, const int32_t* items_k3, int n3, const void* const* Gptrs,
                                               int g_dtype, float* work_base, float* dW_base, float* const* dWptrs, float* dot_part, void* stream) {
    if (n_conv <= 0) return 0;
    if (!dot_part) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (g_dtype == MG_BF16) {
        hipLaunchKernelGGL(snb_bwd_dot_kernel<bf16raw>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, Gptrs, dot_part);
        hipLaunchKernelGGL(snb_bwd_apply_kernel<bf16raw>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, Gptrs, work_base, dW_base, dWptrs, dot_part);
    } else if (g_dtype == MG_F16) {
        hipLaunchKernelGGL(snb_bwd_dot_kernel<f16raw>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, Gptrs, dot_part);
        hipLaunchKernelGGL(snb_bwd_apply_kernel<f16raw>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, Gptrs, work_base, dW_base, dWptrs, dot_part);
    } else {
        hipLaunchKernelGGL(snb_bwd_dot_kernel<float>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, Gptrs, dot_part);
        hipLaunchKernelGGL(snb_bwd_apply_kernel<float>, dim3(n3), dim3(NT), 0, st, descs, (const int4*)items_k3, Gptrs, work_base, dW_base, dWptrs, dot_part);
    }
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_spectral_norm_batched_bwd(const mg_sn_desc* descs, int n_conv, const int32_t* items_k3, int n3, const void* const* Gptrs,
                                            int g_dtype, float* work_base, float* dW_base, float* dot_part, void* stream) {
    return mg_spectral_norm_batched_bwd_to(descs, n_conv, items_k3, n3, Gptrs, g_dtype, work_base, dW_base, nullptr, dot_part, stream);
}
