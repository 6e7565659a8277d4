// Wavefront-level gather / scatter kernels between dense NHWC feature maps and sparse (active-site) feature rows,
// plus the input packing and the alpha-plane kernels of the refinement head. All HBM-bound; one 16-byte chunk per lane,
// consecutive lanes on consecutive channels of the same site so every gathered row is read/written as whole lines.
// Replaces the fancy-indexing + spconv.SparseConvTensor glue of
//   maggie/network/decoder/resnet_inst_matt_spconv.py:161-194 (combine_dense_sparse_feat / instance_spec_guidance),
//   :221-232 (OS8 gather + instance guidance), :247-251,264-268 (.dense() and the -99 fill),
//   :303-304,360-362 (bilinear upsample + (tanh+1)/2), maggie/network/encoder/resnet.py:211-229 (mask-ID embedding).
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;
inline int grid_for(long total) { long b = (total + NT - 1) / NT; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

// out[r, yoff + c] = dense[frame(r), y, x, c] * (mul ? mul[frame, inst, c] : 1)
template <typename T>
__global__ __launch_bounds__(NT) void gather_rows_kernel(const T* __restrict__ dense, const int* __restrict__ coords, int R, int n_i, int Hd,
                                                         int Wd, int C, const float* __restrict__ mul, int mul_ninst, T* __restrict__ out,
                                                         int ldo, int yoff, const int32_t* __restrict__ r_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    R = dev_rows(r_dev, R);
    const long total = (long)R * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int r = (int)(i / cpr), cc = (int)(i - (long)r * cpr);
        int p = coords[(long)r * 3], y = coords[(long)r * 3 + 1], x = coords[(long)r * 3 + 2];
        int frame = p / n_i;
        float f[CE];
        TR::unpack(*(const uint4*)(dense + (((long)frame * Hd + y) * Wd + x) * C + cc * CE), f);
        if (mul) {
            const float* mv = mul + ((long)frame * mul_ninst + (p - frame * n_i)) * C + cc * CE;
#pragma unroll
            for (int e = 0; e < CE; ++e) f[e] *= mv[e];
        }
        *(uint4*)(out + (long)r * ldo + yoff + cc * CE) = TR::pack(f);
    }
}

// backward of the gather: ddense[frame,y,x,c] += dout[r,c] * mul ; dmul[frame,inst,c] += dout[r,c]*dense[...]
// Rows are sorted by plane, so each thread (fixed channel chunk, ascending rows) keeps a running dmul sum for its current
// plane and flushes it only when the plane changes. A flush is a WAVE operation: the lanes that hold the same channel chunk
// are (almost always) in the same plane, so their sums are combined by a butterfly and one lane issues the atomics --
// otherwise ~threads x planes x CE atomics pile onto a few hundred addresses (1 ms at 1.1 M rows).
template <int CE>
__device__ __forceinline__ void wave_flush_dmul(float* run, long run_row, int cpr, float* __restrict__ dmul) {
    const int lane = threadIdx.x & 63;
    const bool pow2 = (cpr & (cpr - 1)) == 0 && cpr <= 32;
    const long first = __shfl(run_row, pow2 ? (lane & (cpr - 1)) : lane, 64);
    if (pow2 && __all(run_row == first)) {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            float v = run[e];
            for (int o = cpr; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < cpr && run_row >= 0 && v != 0.f) atomicAdd(&dmul[run_row + e], v);
            run[e] = 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            if (run_row >= 0 && run[e] != 0.f) atomicAdd(&dmul[run_row + e], run[e]);
            run[e] = 0.f;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void gather_rows_bwd_kernel(const T* __restrict__ dout, int ldo, int yoff, const int* __restrict__ coords,
                                                             int R, int n_i, int Hd, int Wd, int C, const float* __restrict__ mul,
                                                             int mul_ninst, const T* __restrict__ dense, float* __restrict__ ddense,
                                                             float* __restrict__ dmul, const int32_t* __restrict__ r_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    R = dev_rows(r_dev, R);
    const int tid = blockIdx.x * NT + threadIdx.x;
    const int nthreads = gridDim.x * NT;
    const int cc = tid % cpr;
    const int rstep = nthreads / cpr;
    const bool mine = tid < rstep * cpr;
    float run[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) run[e] = 0.f;
    long run_row = -1;
    for (int r = tid / cpr; ; r += rstep) {                       // wave-uniform trip count: flushes are wave operations
        const bool act = mine && r < R;
        if (!__any(act)) break;
        int p = 0, y = 0, x = 0;
        if (act) { p = coords[(long)r * 3]; y = coords[(long)r * 3 + 1]; x = coords[(long)r * 3 + 2]; }
        const int frame = p / n_i;
        const long drow = (((long)frame * Hd + y) * Wd + x) * C + cc * CE;
        const long mrow = ((long)frame * mul_ninst + (p - frame * n_i)) * C + cc * CE;
        float g[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) g[e] = 0.f;
        if (act) TR::unpack(*(const uint4*)(dout + (long)r * ldo + yoff + cc * CE), g);
        if (mul) {
            if (dmul) {
                const bool change = act && mrow != run_row;
                if (__any(change && run_row >= 0)) wave_flush_dmul<CE>(run, run_row, cpr, dmul);
                if (change) run_row = mrow;
                if (act) {
                    float d[CE];
                    TR::unpack(*(const uint4*)(dense + drow), d);
#pragma unroll
                    for (int e = 0; e < CE; ++e) run[e] += g[e] * d[e];
                }
            }
            if (act) {
#pragma unroll
                for (int e = 0; e < CE; ++e) g[e] *= mul[mrow + e];
            }
        }
        if (ddense && act) {
#pragma unroll
            for (int e = 0; e < CE; ++e) atomicAdd(&ddense[drow + e], g[e]);
        }
    }
    if (dmul) wave_flush_dmul<CE>(run, run_row, cpr, dmul);
}

// Deterministic form of the dmul part (csrc/det.hip): rows are sorted by plane, so the rows of plane p are ONE range of `coords` (found by
// binary search). Workgroup (chunk j, token slot q) adds the rows of its chunk of that range in a fixed order and stores one partial row [C];
// mg_det_reduce adds the chunks in order into dmul[q]. No atomics, no dependence on which thread meets which plane.
template <typename T, int NCH>
__global__ __launch_bounds__(NT) void gather_rows_dmul_det_kernel(const T* __restrict__ dout, int ldo, int yoff, const int* __restrict__ coords, int R,
                                                                  int n_i, int Hd, int Wd, int C, int mul_ninst, const T* __restrict__ dense,
                                                                  float* __restrict__ slots, const int32_t* __restrict__ r_dev) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    __shared__ float sred[NT * CE];
    const int cpr = C / CE, nrl = NT / cpr;
    const int q = blockIdx.y, j = blockIdx.x;
    const int frame = q / mul_ninst, inst = q - frame * mul_ninst;
    float* slot = slots + ((size_t)q * NCH + j) * C;
    R = dev_rows(r_dev, R);
    long beg = 0, end = 0;
    if (inst < n_i) {
        const int p = frame * n_i + inst;
        int lo = 0, hi = R;                                       // first row with plane >= p
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (coords[(long)mid * 3] < p) lo = mid + 1; else hi = mid; }
        const int a = lo;
        hi = R;                                                   // first row with plane > p
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (coords[(long)mid * 3] <= p) lo = mid + 1; else hi = mid; }
        const long n = lo - a;
        beg = a + n * j / NCH; end = a + n * (j + 1) / NCH;
    }
    const int lane_c = threadIdx.x % cpr, lrow = threadIdx.x / cpr;
    float run[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) run[e] = 0.f;
    if (lrow < nrl) {
        for (long r = beg + lrow; r < end; r += nrl) {
            const int y = coords[r * 3 + 1], x = coords[r * 3 + 2];
            float g[CE], d[CE];
            TR::unpack(*(const uint4*)(dout + r * ldo + yoff + lane_c * CE), g);
            TR::unpack(*(const uint4*)(dense + (((long)frame * Hd + y) * Wd + x) * C + lane_c * CE), d);
#pragma unroll
            for (int e = 0; e < CE; ++e) run[e] += g[e] * d[e];
        }
#pragma unroll
        for (int e = 0; e < CE; ++e) sred[lrow * C + lane_c * CE + e] = run[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
        float a = 0.f;
        for (int rr = 0; rr < nrl; ++rr) a += sred[rr * C + c];
        slot[c] = a;
    }
}

// atomic-free input gradient of the gather: every dense pixel sums the rows of the (at most n_i) instance planes that are
// active at its position; the row id comes from the level's bit plane + per-word rank (same lookup as the gather tables)
template <typename T>
__global__ __launch_bounds__(NT) void gather_rows_bwd_dense_kernel(const T* __restrict__ dout, int ldo, int yoff,
                                                                   const unsigned long long* __restrict__ bits, const int* __restrict__ wordoff,
                                                                   int n_i, int N, int Hd, int Wd, int C, const float* __restrict__ mul,
                                                                   int mul_ninst, T* __restrict__ ddense) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const int cpr = C / CE;
    const int Ww = (Wd + 63) >> 6;
    const long total = (long)N * Hd * Wd * cpr;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int cc = (int)(i % cpr); long r = i / cpr;
        int x = (int)(r % Wd); r /= Wd; int y = (int)(r % Hd); int frame = (int)(r / Hd);
        float acc[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] = 0.f;
        constexpr int MAXI = 10;                                  // the model's instance slots (resnet_inst_matt_spconv.py: max_inst = 10)
        if (n_i <= MAXI) {
            // the planes' words, then the ranks of the active ones, then their rows: three batches of loads instead of one dependent chain
            // (word -> rank -> row) per plane, plane after plane; the rows are added in plane order as before
            const int b = x & 63;
            unsigned long long m[MAXI];
            int wo[MAXI];
            uint4 q[MAXI];
#pragma unroll
            for (int inst = 0; inst < MAXI; ++inst)
                m[inst] = bits[((long)(frame * n_i + min(inst, n_i - 1)) * Hd + y) * Ww + (x >> 6)];
#pragma unroll
            for (int inst = 0; inst < MAXI; ++inst) {
                wo[inst] = 0;
                if (inst < n_i && ((m[inst] >> b) & 1ull)) wo[inst] = wordoff[((long)(frame * n_i + inst) * Hd + y) * Ww + (x >> 6)];
            }
#pragma unroll
            for (int inst = 0; inst < MAXI; ++inst) {
                q[inst] = make_uint4(0, 0, 0, 0);
                if (inst < n_i && ((m[inst] >> b) & 1ull)) {
                    const int row = wo[inst] + __popcll(m[inst] & ((1ull << b) - 1ull));
                    q[inst] = *(const uint4*)(dout + (long)row * ldo + yoff + cc * CE);
                }
            }
#pragma unroll
            for (int inst = 0; inst < MAXI; ++inst) {
                if (inst < n_i && ((m[inst] >> b) & 1ull)) {
                    float g[CE];
                    TR::unpack(q[inst], g);
                    if (mul) {
                        const float* mv = mul + ((long)frame * mul_ninst + inst) * C + cc * CE;
#pragma unroll
                        for (int e = 0; e < CE; ++e) g[e] *= mv[e];
                    }
#pragma unroll
                    for (int e = 0; e < CE; ++e) acc[e] += g[e];
                }
            }
        } else
        for (int inst = 0; inst < n_i; ++inst) {
            int p = frame * n_i + inst;
            long wi = ((long)p * Hd + y) * Ww + (x >> 6);
            unsigned long long m = bits[wi];
            int b = x & 63;
            if (!((m >> b) & 1ull)) continue;
            int row = wordoff[wi] + __popcll(m & ((1ull << b) - 1ull));
            float g[CE];
            TR::unpack(*(const uint4*)(dout + (long)row * ldo + yoff + cc * CE), g);
            if (mul) {
                const float* mv = mul + ((long)frame * mul_ninst + inst) * C + cc * CE;
#pragma unroll
                for (int e = 0; e < CE; ++e) g[e] *= mv[e];
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) acc[e] += g[e];
        }
        *(uint4*)(ddense + (((long)frame * Hd + y) * Wd + x) * C + cc * CE) = TR::pack(acc);
    }
}

// plane[p, y, x] = vals[r, col]  (plane pre-filled by the caller's fill kernel)
template <typename T>
__global__ __launch_bounds__(NT) void scatter_plane_kernel(const T* __restrict__ vals, int ldv, int col, const int* __restrict__ coords,
                                                           int R, int H, int W, float* __restrict__ plane, const int32_t* __restrict__ r_dev) {
    R = dev_rows(r_dev, R);
    for (int r = blockIdx.x * NT + threadIdx.x; r < R; r += gridDim.x * NT) {
        int p = coords[(long)r * 3], y = coords[(long)r * 3 + 1], x = coords[(long)r * 3 + 2];
        plane[((long)p * H + y) * W + x] = ElemTraits<T>::ld(vals + (long)r * ldv + col);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void gather_plane_kernel(const float* __restrict__ plane, const int* __restrict__ coords, int R, int H, int W,
                                                          T* __restrict__ vals, int ldv, int col, const int32_t* __restrict__ r_dev,
                                                          int zero_rest) {
    R = dev_rows(r_dev, R);
    for (int r = blockIdx.x * NT + threadIdx.x; r < R; r += gridDim.x * NT) {
        int p = coords[(long)r * 3], y = coords[(long)r * 3 + 1], x = coords[(long)r * 3 + 2];
        if (zero_rest) {                                    // the other (padding) columns of the row carry no gradient
            for (int c = 0; c < ldv; ++c) if (c != col) ElemTraits<T>::st(vals + (long)r * ldv + c, 0.f);
        }
        ElemTraits<T>::st(vals + (long)r * ldv + col, plane[((long)p * H + y) * W + x]);
    }
}

__global__ __launch_bounds__(NT) void fill_kernel(float* __restrict__ p, long n, float v) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) p[i] = v;
}

// ---- input packing: RGB + mask-ID embedding -> NHWC with 8 channels (6 real + 2 zero) -----------------------------------
// emb[c] = sum_i [mask_i>0] E[i+1, c] / (count + 1e-6)   with ids = (mask_i * (i+1)).long()   (encoder/resnet.py:214-225)
template <typename T>
__global__ __launch_bounds__(NT) void mask_embed_kernel(const float* __restrict__ image, const float* __restrict__ masks,
                                                        const float* __restrict__ table, int N, int H, int W, int n_m, int Hm, int Wm,
                                                        int n_embed, T* __restrict__ out) {
    const long total = (long)N * H * W;
    const int sy = H / Hm, sx = W / Wm;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int x = (int)(i % W); long r = i / W; int y = (int)(r % H); int n = (int)(r / H);
        float f[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) f[c] = image[(((long)n * 3 + c) * H + y) * W + x];
        float e[3] = {0.f, 0.f, 0.f};
        float cnt = 0.f;
        const float* mp = masks + ((long)n * n_m * Hm + (y / sy)) * Wm + (x / sx);
        for (int k = 0; k < n_m; ++k) {
            float mv = mp[(long)k * Hm * Wm];
            long id = (long)(mv * (float)(k + 1));                 // (masks * mask_ids).long()
            if (id > 0) {
                cnt += 1.f;
                for (int c = 0; c < n_embed && c < 3; ++c) e[c] += table[id * n_embed + c];
            }
        }
        float inv = 1.f / (cnt + 1e-6f);
        f[3] = e[0] * inv; f[4] = e[1] * inv; f[5] = e[2] * inv; f[6] = 0.f; f[7] = 0.f;
        T* dst = out + i * 8;
        if constexpr (sizeof(T) == 2) *(uint4*)dst = ElemTraits<T>::pack(f);
        else { *(uint4*)dst = ElemTraits<float>::pack(f); *(uint4*)((float*)dst + 4) = ElemTraits<float>::pack(f + 4); }
    }
}

// dtable[id, c] += sum_pixels [id active] * g[pixel, 3 + c] / (cnt + 1e-6)
template <typename T>
__global__ __launch_bounds__(NT) void mask_embed_bwd_kernel(const T* __restrict__ dx, const float* __restrict__ masks, int N, int H, int W,
                                                            int n_m, int Hm, int Wm, int n_embed, float* __restrict__ dtable) {
    __shared__ float acc[64 * 3];
    for (int i = threadIdx.x; i < 64 * 3; i += NT) acc[i] = 0.f;
    __syncthreads();
    const long total = (long)N * H * W;
    const int sy = H / Hm, sx = W / Wm;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        int x = (int)(i % W); long r = i / W; int y = (int)(r % H); int n = (int)(r / H);
        const float* mp = masks + ((long)n * n_m * Hm + (y / sy)) * Wm + (x / sx);
        float cnt = 0.f;
        for (int k = 0; k < n_m; ++k) { long id = (long)(mp[(long)k * Hm * Wm] * (float)(k + 1)); if (id > 0) cnt += 1.f; }
        if (cnt == 0.f) continue;
        float inv = 1.f / (cnt + 1e-6f);
        float g[3];
        for (int c = 0; c < 3; ++c) g[c] = ElemTraits<T>::ld(dx + i * 8 + 3 + c) * inv;
        for (int k = 0; k < n_m; ++k) {
            long id = (long)(mp[(long)k * Hm * Wm] * (float)(k + 1));
            if (id > 0 && id < 64) for (int c = 0; c < n_embed && c < 3; ++c) atomicAdd(&acc[id * 3 + c], g[c]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (n_m + 1) * 3; i += NT) {
        int id = i / 3, c = i - id * 3;
        if (c < n_embed && acc[i] != 0.f) atomicAdd(&dtable[id * n_embed + c], acc[i]);
    }
}

// Deterministic form (csrc/det.hip): ids are at most n_m <= 15, so every thread keeps its own [id][3] sums in registers (static indices:
// the id test is a compare per candidate), a workgroup adds its threads in a fixed order and stores ONE row [16][3] of the slot buffer.
template <typename T>
__global__ __launch_bounds__(NT) void mask_embed_bwd_det_kernel(const T* __restrict__ dx, const float* __restrict__ masks, int N, int H, int W,
                                                                int n_m, int Hm, int Wm, int n_embed, float* __restrict__ slots) {
    constexpr int NID = 16;
    __shared__ float sh[NT / 64][NID * 3];
    float acc[NID][3];
#pragma unroll
    for (int j = 0; j < NID; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; }
    const long total = (long)N * H * W;
    const int sy = H / Hm, sx = W / Wm;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; n_m > 0 && i < total; i += (long)gridDim.x * NT) {
        int x = (int)(i % W); long r = i / W; int y = (int)(r % H); int n = (int)(r / H);
        const float* mp = masks + ((long)n * n_m * Hm + (y / sy)) * Wm + (x / sx);
        // the pixel's mask values and gradient as ONE batch of loads (planes past n_m read plane n_m - 1 and are not used): the run-time-length
        // loop `for k < n_m: load, convert, compare` was a memory round trip per plane, twice per pixel (76 us for 1 M pixels)
        float mv[NID - 1], gx[3];
#pragma unroll
        for (int k = 0; k < NID - 1; ++k) mv[k] = mp[(long)min(k, n_m - 1) * Hm * Wm];
#pragma unroll
        for (int c = 0; c < 3; ++c) gx[c] = c < n_embed ? ElemTraits<T>::ld(dx + i * 8 + 3 + c) : 0.f;
        float cnt = 0.f;
#pragma unroll
        for (int k = 0; k < NID - 1; ++k) { if (k < n_m) { long id = (long)(mv[k] * (float)(k + 1)); if (id > 0) cnt += 1.f; } }
        if (cnt == 0.f) continue;
        float inv = 1.f / (cnt + 1e-6f);
        float g[3];
        for (int c = 0; c < 3; ++c) g[c] = c < n_embed ? gx[c] * inv : 0.f;
#pragma unroll
        for (int k = 0; k < NID - 1; ++k) {
            // (a predicate, not `if (k >= n_m) break`: with the early exit the loop was not unrolled, acc[][] was indexed at run time and lived in
            // scratch memory -- 208 bytes per thread, a scratch read-modify-write per hit)
            const int id = k < n_m ? (int)(long)(mv[k] * (float)(k + 1)) : 0;
            if (id == k + 1) { acc[k + 1][0] += g[0]; acc[k + 1][1] += g[1]; acc[k + 1][2] += g[2]; }      // a 0 / 1 mask: id is 0 or k + 1 (static index)
            else if (id > 0) {                                                                             // fractional mask value: any id below k + 1
#pragma unroll
                for (int j = 1; j < NID; ++j)
                    if (id == j) { acc[j][0] += g[0]; acc[j][1] += g[1]; acc[j][2] += g[2]; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NID; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = wave_sum(acc[j][c]);
            if (lane == 0) sh[wave][j * 3 + c] = v;
        }
    __syncthreads();
    const int nv = (n_m + 1) * 3;                                 // slot row = the table [n_m + 1][3]
    for (int i = threadIdx.x; i < nv; i += NT) {
        float v = 0.f;
        for (int wv = 0; wv < NT / 64; ++wv) v += sh[wv][i];
        slots[(size_t)blockIdx.x * nv + i] = v;
    }
}

// ---- bilinear upsample (align_corners=False) + (tanh+1)/2 into fp32 NCHW planes ------------------------------------------
// in element (n, c, y, x) at in[n*sn + c*sc + y*sy + x*sx]
__device__ __forceinline__ void bil_src(int d, int scale, int in_size, int& i0, int& i1, float& lam) {
    float s = ((float)d + 0.5f) / (float)scale - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    lam = s - (float)i0;
}

// grid (blocks over one plane, N * C planes): 32-bit index math inside a plane (the flat 64-bit walk spent ~4 long divisions per pixel) and
// (tanh(v) + 1) / 2 == sigmoid(2 v) = 1 / (1 + exp(-2 v)) through the fast exponential -- the kernel was VALU-bound (46 us for 42 MB).
template <typename T>
__global__ __launch_bounds__(NT) void upsample_tanh_kernel(const T* __restrict__ in, long sn, long sc, long sy, long sx, int N, int C, int h,
                                                           int w, int scale, int apply_tanh, float* __restrict__ out,
                                                           const float* __restrict__ pscale, int* __restrict__ any_nonzero) {
    const int H = h * scale, W = w * scale;
    const int pl = blockIdx.y;
    const int n = pl / C, c = pl - n * C;
    const T* base = in + n * sn + c * sc;
    float* op = out + (long)pl * H * W;
    // `pscale` (0 / 1 per plane): `x_os8 * valid_masks` of resnet_inst_matt_spconv.py:331 without a second pass over the planes; `any_nonzero`:
    // the "coarse alpha is identically zero" test of :314 (`x_os8.sum() == 0`, values are >= 0) without a 42 MB reduction
    const float ps = pscale ? pscale[pl] : 1.f;
    int nz = 0;
    if (pscale && ps == 0.f) {
        for (int i = blockIdx.x * NT + threadIdx.x; i < H * W; i += gridDim.x * NT) op[i] = 0.f;
        return;
    }
    if (scale != 1) {
        // four output pixels per trip, their sixteen source values requested before the first is used (the one-pixel loop below was load x 4 ->
        // wait -> exp -> store, sixteen dependent round trips per thread at 512 x 512); a pixel's arithmetic is unchanged
        const int stride = gridDim.x * NT;
        for (int i0 = blockIdx.x * NT + threadIdx.x; i0 < H * W; i0 += 4 * stride) {
            float v00[4], v01[4], v10[4], v11[4], ly[4], lx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * stride, H * W - 1);             // past the plane: the last pixel again, not stored
                const int Y = i / W, X = i - Y * W;
                int y0, y1, x0, x1;
                bil_src(Y, scale, h, y0, y1, ly[u]);
                bil_src(X, scale, w, x0, x1, lx[u]);
                v00[u] = ElemTraits<T>::ld(base + y0 * sy + x0 * sx); v01[u] = ElemTraits<T>::ld(base + y0 * sy + x1 * sx);
                v10[u] = ElemTraits<T>::ld(base + y1 * sy + x0 * sx); v11[u] = ElemTraits<T>::ld(base + y1 * sy + x1 * sx);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * stride;
                const float v = (1.f - ly[u]) * ((1.f - lx[u]) * v00[u] + lx[u] * v01[u]) + ly[u] * ((1.f - lx[u]) * v10[u] + lx[u] * v11[u]);
                const float o = (apply_tanh ? 1.f / (1.f + __expf(-2.f * v)) : v) * ps;
                if (i < H * W) { op[i] = o; nz |= (o != 0.f); }
            }
        }
        if (any_nonzero && __syncthreads_or(nz) && threadIdx.x == 0) *any_nonzero = 1;
        return;
    }
    for (int i = blockIdx.x * NT + threadIdx.x; i < H * W; i += gridDim.x * NT) {
        const int Y = i / W, X = i - Y * W;
        float v;
        if (scale == 1) {
            v = ElemTraits<T>::ld(base + Y * sy + X * sx);
        } else {
            int y0, y1, x0, x1; float ly, lx;
            bil_src(Y, scale, h, y0, y1, ly);
            bil_src(X, scale, w, x0, x1, lx);
            float v00 = ElemTraits<T>::ld(base + y0 * sy + x0 * sx), v01 = ElemTraits<T>::ld(base + y0 * sy + x1 * sx);
            float v10 = ElemTraits<T>::ld(base + y1 * sy + x0 * sx), v11 = ElemTraits<T>::ld(base + y1 * sy + x1 * sx);
            v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        }
        const float o = (apply_tanh ? 1.f / (1.f + __expf(-2.f * v)) : v) * ps;
        op[i] = o;
        nz |= (o != 0.f);
    }
    if (any_nonzero && __syncthreads_or(nz) && threadIdx.x == 0) *any_nonzero = 1;      // every writer stores the same value
}

// backward: din(n,c,y,x) (fp32, same strides, pre-zeroed) += bilinear^T( dout * (1 - t^2)/2 ), t = 2*out - 1
__global__ __launch_bounds__(NT) void upsample_tanh_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, long sn, long sc,
                                                               long sy, long sx, int N, int C, int h, int w, int scale, int apply_tanh,
                                                               float* __restrict__ din, const float* __restrict__ pscale) {
    const int H = h * scale, W = w * scale;
    const long total = (long)N * C * H * W;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        float g = dout[i];
        if (apply_tanh) { float t = 2.f * out[i] - 1.f; g *= 0.5f * (1.f - t * t); }
        int X = (int)(i % W); long r = i / W; int Y = (int)(r % H); r /= H; int c = (int)(r % C); int n = (int)(r / C);
        if (pscale && pscale[n * C + c] == 0.f) g = 0.f;                      // a 0 / 1 plane scale: masked planes pass no gradient
        float* base = din + n * sn + c * sc;
        if (scale == 1) { base[Y * sy + X * sx] = g; continue; }              // one writer per element: plain store, no pre-zeroed buffer needed
        if (g == 0.f) continue;
        int y0, y1, x0, x1; float ly, lx;
        bil_src(Y, scale, h, y0, y1, ly);
        bil_src(X, scale, w, x0, x1, lx);
        atomicAdd(base + y0 * sy + x0 * sx, g * (1.f - ly) * (1.f - lx));
        atomicAdd(base + y0 * sy + x1 * sx, g * (1.f - ly) * lx);
        atomicAdd(base + y1 * sy + x0 * sx, g * ly * (1.f - lx));
        atomicAdd(base + y1 * sy + x1 * sx, g * ly * lx);
    }
}

// Atomic-free transpose of the x4 / x8 bilinear upsampling: a block owns an 8x8 tile of SOURCE pixels, stages the tanh-scaled
// output gradient of the (8+2)*SCALE square that can reach them in LDS, and every source pixel gathers its weights. (The
// scatter version issues 4 atomics per output pixel, ~64-256 of them onto each source address.)
template <int SCALE>
__global__ __launch_bounds__(NT) void upsample_tanh_bwd_tile_kernel(const float* __restrict__ dout, const float* __restrict__ out, long sn, long sc,
                                                                    long sy, long sx, int C, int h, int w, int apply_tanh,
                                                                    float* __restrict__ din, const float* __restrict__ pscale) {
    constexpr int TS = 8, R = (TS + 2) * SCALE;                   // LDS tile edge in output pixels
    __shared__ float sg[R * R];
    const int H = h * SCALE, W = w * SCALE;
    const int plane = blockIdx.z, n = plane / C, c = plane - n * C;
    const bool masked = pscale && pscale[plane] == 0.f;           // 0 / 1 plane scale of the forward: a masked plane passes no gradient
    const int y_src0 = blockIdx.y * TS, x_src0 = blockIdx.x * TS;
    const int Y0 = (y_src0 - 1) * SCALE, X0 = (x_src0 - 1) * SCALE;
    const long pbase = (long)plane * H * W;
    for (int i = threadIdx.x; i < R * R; i += NT) {
        const int ry = i / R, rx = i - ry * R, Y = Y0 + ry, X = X0 + rx;
        float g = 0.f;
        if (!masked && Y >= 0 && Y < H && X >= 0 && X < W) {
            g = dout[pbase + (long)Y * W + X];
            if (apply_tanh) { const float t = 2.f * out[pbase + (long)Y * W + X] - 1.f; g *= 0.5f * (1.f - t * t); }
        }
        sg[i] = g;
    }
    __syncthreads();
    // 4 threads per source pixel: each takes a quarter of the window rows
    const int sp = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int ys = y_src0 + (sp >> 3), xs = x_src0 + (sp & 7);
    float acc = 0.f;
    constexpr int WIN = 2 * SCALE + 2;
    const int wy0 = (ys - 1) * SCALE + SCALE / 2 - 1, wx0 = (xs - 1) * SCALE + SCALE / 2 - 1;
    for (int iy = part; iy < WIN; iy += 4) {
        const int Y = wy0 + iy;
        if (Y < 0 || Y >= H) continue;
        int a0, a1; float la;
        bil_src(Y, SCALE, h, a0, a1, la);
        float wyv = 0.f;
        if (a0 == ys) wyv += 1.f - la;
        if (a1 == ys) wyv += la;
        if (wyv == 0.f) continue;
        float rowacc = 0.f;
        for (int ix = 0; ix < WIN; ++ix) {
            const int X = wx0 + ix;
            if (X < 0 || X >= W) continue;
            int b0, b1; float lb;
            bil_src(X, SCALE, w, b0, b1, lb);
            float wxv = 0.f;
            if (b0 == xs) wxv += 1.f - lb;
            if (b1 == xs) wxv += lb;
            if (wxv != 0.f) rowacc += wxv * sg[(Y - Y0) * R + (X - X0)];
        }
        acc += wyv * rowacc;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0 && ys < h && xs < w) din[n * sn + c * sc + ys * sy + xs * sx] = acc;   // one writer per element: no pre-zeroed buffer needed for the planes it covers
}

}  // namespace

extern "C" int mg_gather_rows_dev(const void* dense, int dtype, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C, const float* mul,
                                  int mul_ninst, void* out, int ldo, int yoff, const int32_t* r_dev, void* stream);
extern "C" int mg_gather_rows(const void* dense, int dtype, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C, const float* mul,
                              int mul_ninst, void* out, int ldo, int yoff, void* stream) {
    return mg_gather_rows_dev(dense, dtype, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, out, ldo, yoff, nullptr, stream);
}
extern "C" int mg_gather_rows_dev(const void* dense, int dtype, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C, const float* mul,
                                  int mul_ninst, void* out, int ldo, int yoff, const int32_t* r_dev, void* stream) {
    if (R <= 0) return 0;
    const int ce = MG_IS16(dtype) ? 8 : 4;
    if (C % ce || ldo % ce || yoff % ce) return -3;
    long total = (long)R * (C / ce);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gather_rows_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)dense, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, (bf16raw*)out, ldo, yoff, r_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gather_rows_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)dense, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, (f16raw*)out, ldo, yoff, r_dev);
    else hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)dense, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, (float*)out, ldo, yoff, r_dev);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_gather_rows_bwd_dev(const void* dout, int dtype, int ldo, int yoff, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C,
                                      const float* mul, int mul_ninst, const void* dense, float* ddense, float* dmul, const int32_t* r_dev, void* stream);
extern "C" int mg_gather_rows_bwd(const void* dout, int dtype, int ldo, int yoff, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C,
                                  const float* mul, int mul_ninst, const void* dense, float* ddense, float* dmul, void* stream) {
    return mg_gather_rows_bwd_dev(dout, dtype, ldo, yoff, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, dense, ddense, dmul, nullptr, stream);
}
extern "C" int mg_gather_rows_bwd_dev(const void* dout, int dtype, int ldo, int yoff, const int32_t* coords, int R, int n_i, int Hd, int Wd, int C,
                                      const float* mul, int mul_ninst, const void* dense, float* ddense, float* dmul, const int32_t* r_dev, void* stream) {
    if (R <= 0) return 0;
    const int ce = MG_IS16(dtype) ? 8 : 4;
    if (C % ce || ldo % ce || yoff % ce) return -3;
    long total = (long)R * (C / ce);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gather_rows_bwd_kernel<bf16raw>, dim3(grid_for(total) > 512 ? 512 : grid_for(total)), dim3(NT), 0, st, (const bf16raw*)dout, ldo, yoff, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, (const bf16raw*)dense, ddense, dmul, r_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gather_rows_bwd_kernel<f16raw>, dim3(grid_for(total) > 512 ? 512 : grid_for(total)), dim3(NT), 0, st, (const f16raw*)dout, ldo, yoff, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, (const f16raw*)dense, ddense, dmul, r_dev);
    else hipLaunchKernelGGL(gather_rows_bwd_kernel<float>, dim3(grid_for(total) > 512 ? 512 : grid_for(total)), dim3(NT), 0, st, (const float*)dout, ldo, yoff, coords, R, n_i, Hd, Wd, C, mul, mul_ninst, (const float*)dense, ddense, dmul, r_dev);
    MG_CHECK_LAUNCH();
    return 0;
}

/* dmul[frame, inst, c] = sum over the rows r of plane (frame, inst) of dout[r, yoff + c] * dense[frame, y_r, x_r, c]   (fp32 [N, mul_ninst, C], OVERWRITTEN):
 * the token-multiplier gradient of mg_gather_rows in bit-reproducible form (rows sorted by plane, as the site lists are). */
extern "C" int mg_gather_rows_dmul_det(const void* dout, int dtype, int ldo, int yoff, const int32_t* coords, int R, int n_i, int N, int Hd, int Wd, int C,
                                       int mul_ninst, const void* dense, float* dmul, const int32_t* r_dev, void* stream) {
    const int ce = MG_IS16(dtype) ? 8 : 4;
    if (C % ce || ldo % ce || yoff % ce || NT % (C / ce) || n_i > mul_ninst) return -3;
    if (N <= 0 || mul_ninst <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    constexpr int NCH = 16;
    const int Q = N * mul_ninst;
    hipError_t e = mg_zero_words(dmul, (long)Q * C, st);
    if (e != hipSuccess) return (int)e;
    if (R <= 0) return 0;
    float* slots = mg_det_scratch((long)Q * NCH * C);
    if (!slots) return MG_DET_NO_SCRATCH;
    if (dtype == MG_BF16) hipLaunchKernelGGL((gather_rows_dmul_det_kernel<bf16raw, NCH>), dim3(NCH, Q), dim3(NT), 0, st, (const bf16raw*)dout, ldo, yoff, coords, R, n_i, Hd, Wd, C, mul_ninst, (const bf16raw*)dense, slots, r_dev);
    else if (dtype == MG_F16) hipLaunchKernelGGL((gather_rows_dmul_det_kernel<f16raw, NCH>), dim3(NCH, Q), dim3(NT), 0, st, (const f16raw*)dout, ldo, yoff, coords, R, n_i, Hd, Wd, C, mul_ninst, (const f16raw*)dense, slots, r_dev);
    else hipLaunchKernelGGL((gather_rows_dmul_det_kernel<float, NCH>), dim3(NCH, Q), dim3(NT), 0, st, (const float*)dout, ldo, yoff, coords, R, n_i, Hd, Wd, C, mul_ninst, (const float*)dense, slots, r_dev);
    MG_CHECK_LAUNCH();
    mg_det_seg sg{dmul, C, (long)C};
    return mg_det_reduce(slots, NCH, Q, C, 0, &sg, 1, st);
}

extern "C" int mg_gather_rows_bwd_dense(const void* dout, int dtype, int ldo, int yoff, const void* bits, const int32_t* wordoff, int n_i, int N,
                                        int Hd, int Wd, int C, const float* mul, int mul_ninst, void* ddense, void* stream) {
    const int ce = MG_IS16(dtype) ? 8 : 4;
    if (C % ce || ldo % ce || yoff % ce) return -3;
    long total = (long)N * Hd * Wd * (C / ce);
    if (total <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gather_rows_bwd_dense_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const bf16raw*)dout, ldo, yoff, (const unsigned long long*)bits, wordoff, n_i, N, Hd, Wd, C, mul, mul_ninst, (bf16raw*)ddense);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gather_rows_bwd_dense_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, (const f16raw*)dout, ldo, yoff, (const unsigned long long*)bits, wordoff, n_i, N, Hd, Wd, C, mul, mul_ninst, (f16raw*)ddense);
    else hipLaunchKernelGGL(gather_rows_bwd_dense_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, (const float*)dout, ldo, yoff, (const unsigned long long*)bits, wordoff, n_i, N, Hd, Wd, C, mul, mul_ninst, (float*)ddense);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_scatter_plane_dev(const void* vals, int dtype, int ldv, int col, const int32_t* coords, int R, int P, int H, int W, float fill,
                                    float* plane, const int32_t* r_dev, void* stream);
extern "C" int mg_scatter_plane(const void* vals, int dtype, int ldv, int col, const int32_t* coords, int R, int P, int H, int W, float fill,
                                float* plane, void* stream) {
    return mg_scatter_plane_dev(vals, dtype, ldv, col, coords, R, P, H, W, fill, plane, nullptr, stream);
}
extern "C" int mg_scatter_plane_dev(const void* vals, int dtype, int ldv, int col, const int32_t* coords, int R, int P, int H, int W, float fill,
                                    float* plane, const int32_t* r_dev, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    long n = (long)P * H * W;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(NT), 0, st, plane, n, fill);
    if (R > 0) {
        if (dtype == MG_BF16) hipLaunchKernelGGL(scatter_plane_kernel<bf16raw>, dim3(grid_for(R)), dim3(NT), 0, st, (const bf16raw*)vals, ldv, col, coords, R, H, W, plane, r_dev);
        else if (dtype == MG_F16) hipLaunchKernelGGL(scatter_plane_kernel<f16raw>, dim3(grid_for(R)), dim3(NT), 0, st, (const f16raw*)vals, ldv, col, coords, R, H, W, plane, r_dev);
        else hipLaunchKernelGGL(scatter_plane_kernel<float>, dim3(grid_for(R)), dim3(NT), 0, st, (const float*)vals, ldv, col, coords, R, H, W, plane, r_dev);
    }
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_gather_plane_dev(const float* plane, const int32_t* coords, int R, int H, int W, void* vals, int dtype, int ldv, int col,
                                   const int32_t* r_dev, int zero_rest, void* stream);
extern "C" int mg_gather_plane(const float* plane, const int32_t* coords, int R, int H, int W, void* vals, int dtype, int ldv, int col,
                               void* stream) {
    return mg_gather_plane_dev(plane, coords, R, H, W, vals, dtype, ldv, col, nullptr, 0, stream);
}
extern "C" int mg_gather_plane_dev(const float* plane, const int32_t* coords, int R, int H, int W, void* vals, int dtype, int ldv, int col,
                                   const int32_t* r_dev, int zero_rest, void* stream) {
    if (R <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(gather_plane_kernel<bf16raw>, dim3(grid_for(R)), dim3(NT), 0, st, plane, coords, R, H, W, (bf16raw*)vals, ldv, col, r_dev, zero_rest);
    else if (dtype == MG_F16) hipLaunchKernelGGL(gather_plane_kernel<f16raw>, dim3(grid_for(R)), dim3(NT), 0, st, plane, coords, R, H, W, (f16raw*)vals, ldv, col, r_dev, zero_rest);
    else hipLaunchKernelGGL(gather_plane_kernel<float>, dim3(grid_for(R)), dim3(NT), 0, st, plane, coords, R, H, W, (float*)vals, ldv, col, r_dev, zero_rest);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_mask_embed(const float* image, const float* masks, const float* table, int N, int H, int W, int n_m, int Hm, int Wm,
                             int n_embed, void* out, int dtype, void* stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    if (H % Hm || W % Wm || n_embed > 3) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(mask_embed_kernel<bf16raw>, dim3(grid_for(total)), dim3(NT), 0, st, image, masks, table, N, H, W, n_m, Hm, Wm, n_embed, (bf16raw*)out);
    else if (dtype == MG_F16) hipLaunchKernelGGL(mask_embed_kernel<f16raw>, dim3(grid_for(total)), dim3(NT), 0, st, image, masks, table, N, H, W, n_m, Hm, Wm, n_embed, (f16raw*)out);
    else hipLaunchKernelGGL(mask_embed_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, st, image, masks, table, N, H, W, n_m, Hm, Wm, n_embed, (float*)out);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_mask_embed_bwd(const void* dx, int dtype, const float* masks, int N, int H, int W, int n_m, int Hm, int Wm, int n_embed,
                                 float* dtable, void* stream) {
    long total = (long)N * H * W;
    if (total <= 0) return 0;
    if (n_m + 1 > 64) return -2;
    hipStream_t st = (hipStream_t)stream;
    int g = grid_for(total); if (g > 1024) g = 1024;
    if (mg_det_on && n_embed == 3 && n_m < 16) {                  // one partial table per workgroup, added in workgroup order (csrc/det.hip)
        const int nv = (n_m + 1) * 3;
        if (g > 512) g = 512;
        float* slots = mg_det_scratch((long)g * nv);
        if (!slots) return MG_DET_NO_SCRATCH;
        if (dtype == MG_BF16) hipLaunchKernelGGL(mask_embed_bwd_det_kernel<bf16raw>, dim3(g), dim3(NT), 0, st, (const bf16raw*)dx, masks, N, H, W, n_m, Hm, Wm, n_embed, slots);
        else if (dtype == MG_F16) hipLaunchKernelGGL(mask_embed_bwd_det_kernel<f16raw>, dim3(g), dim3(NT), 0, st, (const f16raw*)dx, masks, N, H, W, n_m, Hm, Wm, n_embed, slots);
        else hipLaunchKernelGGL(mask_embed_bwd_det_kernel<float>, dim3(g), dim3(NT), 0, st, (const float*)dx, masks, N, H, W, n_m, Hm, Wm, n_embed, slots);
        MG_CHECK_LAUNCH();
        return mg_det_reduce1(slots, g, dtable, nv, st);
    }
    if (dtype == MG_BF16) hipLaunchKernelGGL(mask_embed_bwd_kernel<bf16raw>, dim3(g), dim3(NT), 0, st, (const bf16raw*)dx, masks, N, H, W, n_m, Hm, Wm, n_embed, dtable);
    else if (dtype == MG_F16) hipLaunchKernelGGL(mask_embed_bwd_kernel<f16raw>, dim3(g), dim3(NT), 0, st, (const f16raw*)dx, masks, N, H, W, n_m, Hm, Wm, n_embed, dtable);
    else hipLaunchKernelGGL(mask_embed_bwd_kernel<float>, dim3(g), dim3(NT), 0, st, (const float*)dx, masks, N, H, W, n_m, Hm, Wm, n_embed, dtable);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_upsample_tanh_ex(const void* in, int dtype, long sn, long sc, long sy, long sx, int N, int C, int h, int w, int scale,
                                   int apply_tanh, float* out, const float* pscale, int32_t* any_nonzero, void* stream);
extern "C" int mg_upsample_tanh(const void* in, int dtype, long sn, long sc, long sy, long sx, int N, int C, int h, int w, int scale,
                                int apply_tanh, float* out, void* stream) {
    return mg_upsample_tanh_ex(in, dtype, sn, sc, sy, sx, N, C, h, w, scale, apply_tanh, out, nullptr, nullptr, stream);
}
extern "C" int mg_upsample_tanh_ex(const void* in, int dtype, long sn, long sc, long sy, long sx, int N, int C, int h, int w, int scale,
                                   int apply_tanh, float* out, const float* pscale, int32_t* any_nonzero, void* stream) {
    long total = (long)N * C * h * w * scale * scale;
    if (total <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)h * w * scale * scale;
    long bx = (hw + NT - 1) / NT; if (bx > 64) bx = 64;
    dim3 grid((unsigned)bx, (unsigned)(N * C));
    if (dtype == MG_BF16) hipLaunchKernelGGL(upsample_tanh_kernel<bf16raw>, grid, dim3(NT), 0, st, (const bf16raw*)in, sn, sc, sy, sx, N, C, h, w, scale, apply_tanh, out, pscale, any_nonzero);
    else if (dtype == MG_F16) hipLaunchKernelGGL(upsample_tanh_kernel<f16raw>, grid, dim3(NT), 0, st, (const f16raw*)in, sn, sc, sy, sx, N, C, h, w, scale, apply_tanh, out, pscale, any_nonzero);
    else hipLaunchKernelGGL(upsample_tanh_kernel<float>, grid, dim3(NT), 0, st, (const float*)in, sn, sc, sy, sx, N, C, h, w, scale, apply_tanh, out, pscale, any_nonzero);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_upsample_tanh_bwd_ex(const float* dout, const float* out, long sn, long sc, long sy, long sx, int N, int C, int h, int w,
                                       int scale, int apply_tanh, float* din, const float* pscale, void* stream);
extern "C" int mg_upsample_tanh_bwd(const float* dout, const float* out, long sn, long sc, long sy, long sx, int N, int C, int h, int w,
                                    int scale, int apply_tanh, float* din, void* stream) {
    return mg_upsample_tanh_bwd_ex(dout, out, sn, sc, sy, sx, N, C, h, w, scale, apply_tanh, din, nullptr, stream);
}
extern "C" int mg_upsample_tanh_bwd_ex(const float* dout, const float* out, long sn, long sc, long sy, long sx, int N, int C, int h, int w,
                                       int scale, int apply_tanh, float* din, const float* pscale, void* stream) {
    long total = (long)N * C * h * w * scale * scale;
    if (total <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // (ragged source sizes: the last tiles run past the plane, their extra source pixels are not written -- the gather form has no atomics,
    // so the gradient is bit-reproducible at every size)
    if (scale == 4)
        hipLaunchKernelGGL(upsample_tanh_bwd_tile_kernel<4>, dim3((w + 7) / 8, (h + 7) / 8, N * C), dim3(NT), 0, st, dout, out, sn, sc, sy, sx, C, h, w, apply_tanh, din, pscale);
    else if (scale == 8)
        hipLaunchKernelGGL(upsample_tanh_bwd_tile_kernel<8>, dim3((w + 7) / 8, (h + 7) / 8, N * C), dim3(NT), 0, st, dout, out, sn, sc, sy, sx, C, h, w, apply_tanh, din, pscale);
    else
        hipLaunchKernelGGL(upsample_tanh_bwd_kernel, dim3(grid_for(total)), dim3(NT), 0, st, dout, out, sn, sc, sy, sx, N, C, h, w, scale, apply_tanh, din, pscale);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Inference post-path: maggie/utils/postprocessing.py:36-64 `reverse_transform_tensor` (crop the bottom/right
// padding, bilinear resize with align_corners=True back to the original size) fused with the alpha snapping of
// maggie/engine/test.py:139-142,229-231 (alpha <= 1/255 -> 0, >= 254/255 -> 1), on fp32 planes [P, Hin, Win] -> [P, Hout, Wout].
// The reference does the resize on the GPU, copies to the host and snaps in numpy; this keeps the result on the device.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(NT) void postprocess_alpha_kernel(const float* __restrict__ in, int Hin, int Win, int Hc, int Wc, int Hout, int Wout,
                                                              float sy, float sx, int snap, float* __restrict__ out) {
    const int p = blockIdx.y;
    const long n = (long)Hout * Wout;
    const float* src = in + (long)p * Hin * Win;
    float* dst = out + (long)p * n;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int y = (int)(i / Wout), x = (int)(i - (long)y * Wout);
        // F.interpolate(mode='bilinear', align_corners=True): source = dst * (in - 1) / (out - 1) on the CROPPED (Hc x Wc) image
        const float fy = y * sy, fx = x * sx;
        int y0 = (int)fy, x0 = (int)fx;
        if (y0 > Hc - 1) y0 = Hc - 1;
        if (x0 > Wc - 1) x0 = Wc - 1;
        const int y1 = y0 + 1 < Hc ? y0 + 1 : Hc - 1, x1 = x0 + 1 < Wc ? x0 + 1 : Wc - 1;
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float v00 = src[(long)y0 * Win + x0], v01 = src[(long)y0 * Win + x1];
        const float v10 = src[(long)y1 * Win + x0], v11 = src[(long)y1 * Win + x1];
        float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        if (snap) {
            if (v <= 1.0f / 255.0f) v = 0.f;
            if (v >= 254.0f / 255.0f) v = 1.f;
        }
        dst[i] = v;
    }
}
}  // namespace

extern "C" int mg_postprocess_alpha(const float* in, int P, int Hin, int Win, int crop_h, int crop_w, int Hout, int Wout, int snap, float* out,
                                    void* stream) {
    if (P <= 0 || Hout <= 0 || Wout <= 0) return 0;
    if (crop_h <= 0 || crop_w <= 0 || crop_h > Hin || crop_w > Win) return -2;
    const float sy = Hout > 1 ? (float)(crop_h - 1) / (float)(Hout - 1) : 0.f;
    const float sx = Wout > 1 ? (float)(crop_w - 1) / (float)(Wout - 1) : 0.f;
    long blocks = ((long)Hout * Wout + NT - 1) / NT;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(postprocess_alpha_kernel, dim3((unsigned)blocks, P), dim3(NT), 0, (hipStream_t)stream, in, Hin, Win, crop_h, crop_w, Hout,
                       Wout, sy, sx, snap, out);
    MG_CHECK_LAUNCH();
    return 0;
}
