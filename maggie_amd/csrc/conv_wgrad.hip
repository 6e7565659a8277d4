// Weight-gradient half of the implicit-GEMM convolution family for gfx950 (its own translation unit: the fprop / dgrad templates of
// conv_igemm.hip and these compile in parallel). Entry points: mg_conv_wgrad_workspace, mg_conv_wgrad_ws, mg_conv_wgrad.
#include "common.h"
#include "conv_xcd.h"
#include "../../include/maggie_hip.h"
#include <stdlib.h>
#include <type_traits>

#define MG_STAMP(i)

// =====================================================================================================================
// Weight gradient:  dW[co, tap, ci] += sum_m dY[m, co] * X[src(m, tap), ci]
// One block = one (co tile, tap, ci tile, row range). The reduction dimension (rows) is the NHWC-strided one, so the
// MFMA operands are "K-strided" in memory: tiles are staged row-major in LDS and read transposed -- with
// ds_read_b64_tr_b16 for bf16 (gfx950 transpose read; K permuted consistently between A and B) and with plain
// ds_read_b32 for the f32 16x16x4 MFMA whose operand layout is already one (row, k) scalar per lane.
// The 4 waves of a block split each row step between them (intra-block split-K), are reduced through LDS float
// atomics, and the block tile is added to the fp32 dW with global atomics (row ranges of different blocks overlap).
// =====================================================================================================================
namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

// XF (round 5): the x operand is a RAW conv output; act(x * xf_scale + xf_shift) is applied to the chunks that exist (in-image / live neighbour)
// between the staging registers and LDS. A thread's x chunks share one channel group (256 % CPX == 0): 16 constants in registers.
template <typename T, int TCO, int TCI, int MODE, bool XF = false>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const mg_conv_params p, int rows_per_block, float* __restrict__ ws) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    constexpr bool BF = sizeof(T) == 2;
    constexpr int KSTEP = BF ? 128 : 64;
    constexpr int WROWS = KSTEP / 4;
    constexpr int PAD = BF ? 8 : 16;
    constexpr int PCO = TCO + PAD, PCI = TCI + PAD;
    constexpr int CPY = TCO / CE, CPX = TCI / CE;                // 16-byte chunks per tile row
    constexpr int ITY = (KSTEP * CPY) / 256, ITX = (KSTEP * CPX) / 256;
    constexpr int FM = TCO / 16, FN = TCI / 16;
    static_assert((KSTEP * CPY) % 256 == 0 && (KSTEP * CPX) % 256 == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sY = (T*)smem;                                            // [KSTEP][PCO]
    T* sX = (T*)(smem + KSTEP * PCO * sizeof(T));                // [KSTEP][PCI]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int taps = p.R * p.S;
    const int nci = (p.Cin + TCI - 1) / TCI;
    // work order: (tap, ci tile) fastest, then the co tile, then the row split -- every block of one row split reads the same
    // rows of x and dY, so they run back to back on one XCD
    const int nco = (p.Cout + TCO - 1) / TCO;
    int nsplit = (p.M + rows_per_block - 1) / rows_per_block;        // from the capacity when the row count is a device word
    const int M = dev_rows(p.m_dev, p.M);
    if (p.m_dev) {
        // same number of row splits (one workspace slab each), the rows actually present divided evenly between them
        rows_per_block = (((M + nsplit - 1) / nsplit + KSTEP - 1) / KSTEP) * KSTEP;
        if (rows_per_block < KSTEP) rows_per_block = KSTEP;
    }
    int work;
    if (!xcd_order(nsplit * taps * nci * nco, work)) return;
    const int tc = work % (taps * nci);
    const int rest = work / (taps * nci);
    const int split = rest / nco;
    const int tap = tc / nci;
    const int ci0 = (tc - tap * nci) * TCI;
    const int co0 = (rest - split * nco) * TCO;
    const int mbeg = split * rows_per_block;
    const int mend = min(M, mbeg + rows_per_block);
    if (mbeg >= mend && !p.m_dev) return;                            // device row count: an empty split still writes its (zero) slab
    const int ky = tap / p.S, kx = tap - ky * p.S;
    const FastDiv div_hw(MODE == MG_MODE_GATHER ? 1 : p.Hout * p.Wout), div_w(MODE == MG_MODE_GATHER ? 1 : p.Wout);
    const int sshift = p.stride == 1 ? 0 : (p.stride == 2 ? 1 : (p.stride == 4 ? 2 : -1));
    const T* __restrict__ yb = (const T*)p.y;
    const T* __restrict__ xb = (const T*)p.x;
    const bool yvec = (p.ldy % CE == 0) && (p.yoff % CE == 0);

    uint4 ry[ITY], rx[ITX];
    [[maybe_unused]] float xsc[CE < 8 ? 8 : CE], xsh[CE < 8 ? 8 : CE];
    [[maybe_unused]] unsigned xlive = 0;
    [[maybe_unused]] const float xsl = xf_slope_of(p.xf_act, p.xf_slope);
    if constexpr (XF) {
        static_assert(BF && 256 % CPX == 0, "operand transform: 16-bit storage, one channel group per thread");
        const int c = t % CPX;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = ci0 + c * CE + e;
            xsc[e] = ci < p.Cin ? p.xf_scale[ci] : 0.f; xsh[e] = ci < p.Cin ? p.xf_shift[ci] : 0.f;
        }
    }
    auto load_step = [&](int mb) {
#pragma unroll
        for (int i = 0; i < ITY; ++i) {
            int idx = t + i * 256;
            int row = idx / CPY, c = idx - row * CPY;
            int m = mb + row, co = co0 + c * CE;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (m < mend && co < p.Cout) {
                const T* src = yb + (long)m * p.ldy + p.yoff + co;
                if (yvec && co + CE <= p.Cout) q = *(const uint4*)src;
                else {
                    float f[CE];
#pragma unroll
                    for (int e = 0; e < CE; ++e) f[e] = (co + e < p.Cout) ? TR::ld(src + e) : 0.f;
                    q = TR::pack(f);
                }
            }
            ry[i] = q;
        }
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            int idx = t + i * 256;
            int row = idx / CPX, c = idx - row * CPX;
            int m = mb + row, ci = ci0 + c * CE;
            long src = -1;
            if (m < mend && ci < p.Cin) {
                if (MODE == MG_MODE_GATHER) {
                    src = p.nbr[(long)m * taps + tap];
                } else {
                    int n, rem, ho, wo;
                    div_hw.divmod(m, n, rem);
                    div_w.divmod(rem, ho, wo);
                    if (MODE == MG_MODE_CONV) {
                        int hi = ho * p.stride - p.pad + ky * p.dil, wi = wo * p.stride - p.pad + kx * p.dil;
                        if (hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win) src = ((long)n * p.Hin + hi) * p.Win + wi;
                    } else {
                        int th = ho + p.pad - ky * p.dil, tw = wo + p.pad - kx * p.dil;
                        if (th >= 0 && tw >= 0) {
                            int hi, wi;
                            if (sshift >= 0) { hi = th >> sshift; wi = tw >> sshift; }
                            else { hi = th / p.stride; wi = tw / p.stride; }
                            if (hi * p.stride == th && wi * p.stride == tw && hi < p.Hin && wi < p.Win)
                                src = ((long)n * p.Hin + hi) * p.Win + wi;
                        }
                    }
                }
            }
            rx[i] = (src >= 0) ? *(const uint4*)(xb + src * p.ldx + ci) : make_uint4(0, 0, 0, 0);
            if constexpr (XF) xlive = src >= 0 ? (xlive | (1u << i)) : (xlive & ~(1u << i));
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < ITY; ++i) {
            int idx = t + i * 256; int row = idx / CPY, c = idx - row * CPY;
            *(uint4*)(sY + row * PCO + c * CE) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            int idx = t + i * 256; int row = idx / CPX, c = idx - row * CPX;
            if constexpr (XF) { if (xlive & (1u << i)) rx[i] = xf_apply8<T>(rx[i], xsc, xsh, xsl); }
            *(uint4*)(sX + row * PCI + c * CE) = rx[i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wr = wave * WROWS;
    load_step(mbeg);
    for (int mb = mbeg; mb < mend; mb += KSTEP) {
        store_step();
        __syncthreads();
        if (mb + KSTEP < mend) load_step(mb + KSTEP);
        if constexpr (BF) {
            s16x4 a[FM][2], b[FN][2];
            const int r0 = wr + g * 4 + (li >> 2), cq = (li & 3) * 4;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                a[i][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + r0 * PCO + i * 16 + cq));
                a[i][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + (r0 + 16) * PCO + i * 16 + cq));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                b[j][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + r0 * PCI + j * 16 + cq));
                b[j][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + (r0 + 16) * PCI + j * 16 + cq));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    union { s16x4 h[2]; uint4 v; } ua, ub;
                    ua.h[0] = a[i][0]; ua.h[1] = a[i][1]; ub.h[0] = b[j][0]; ub.h[1] = b[j][1];
                    acc[i][j] = mfma16<T>(ua.v, ub.v, acc[i][j]);
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < WROWS / 4; ++kk) {
                float a[FM], b[FN];
                const int r = wr + kk * 4 + g;
#pragma unroll
                for (int i = 0; i < FM; ++i) a[i] = ((const float*)sY)[r * PCO + i * 16 + li];
#pragma unroll
                for (int j = 0; j < FN; ++j) b[j] = ((const float*)sX)[r * PCI + j * 16 + li];
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // combine the 4 waves' partial tiles: each wave parks its accumulators in its own LDS slab (no LDS atomics: those
    // serialise badly), then all threads sum the four slabs for the elements they write out
    float* sR = (float*)smem;                                     // [4][TCO][TCI + 1]
    constexpr int LDRR = TCI + 1;
    constexpr int SLAB = TCO * LDRR;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sR[wave * SLAB + (i * 16 + g * 4 + e) * LDRR + j * 16 + li] = acc[i][j][e];
    __syncthreads();
    float* __restrict__ dw = p.stats;
    if (ws) {
        // deterministic two-stage reduction: this row-split's partial tile goes to its own slab of the workspace
        float* __restrict__ slab = ws + (long)split * p.Cout * taps * p.Cin;
        for (int i = t; i < TCO * TCI; i += 256) {
            int co = i / TCI, ci = i - co * TCI;
            if (co0 + co < p.Cout && ci0 + ci < p.Cin)
                slab[((long)(co0 + co) * taps + tap) * p.Cin + ci0 + ci] =
                    sR[co * LDRR + ci] + sR[SLAB + co * LDRR + ci] + sR[2 * SLAB + co * LDRR + ci] + sR[3 * SLAB + co * LDRR + ci];
        }
        return;
    }
    for (int i = t; i < TCO * TCI; i += 256) {
        int co = i / TCI, ci = i - co * TCI;
        if (co0 + co < p.Cout && ci0 + ci < p.Cin)
            atomicAdd(&dw[((long)(co0 + co) * taps + tap) * p.Cin + ci0 + ci],
                      sR[co * LDRR + ci] + sR[SLAB + co * LDRR + ci] + sR[2 * SLAB + co * LDRR + ci] + sR[3 * SLAB + co * LDRR + ci]);
    }
}

// ---- split slabs -> dW. Three forms, each a fixed function of (splits, n): the sum of an element never depends on the launch geometry, so
// the per-layer kernels below and the batched kernel (all the parked layers of a backward pass in one launch) give the same bits. ----
template <typename TO>
__device__ __forceinline__ void reduce_elems(const float* __restrict__ ws, int splits, long n, TO* __restrict__ dw, long lb, long nblk) {
    for (long i = lb * 256 + threadIdx.x; i < n; i += nblk * 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = 0;
        for (; s + 4 <= splits; s += 4) {
            a0 += ws[(long)s * n + i]; a1 += ws[(long)(s + 1) * n + i]; a2 += ws[(long)(s + 2) * n + i]; a3 += ws[(long)(s + 3) * n + i];
        }
        for (; s < splits; ++s) a0 += ws[(long)s * n + i];
        ElemTraits<TO>::st(dw + i, (a0 + a1) + (a2 + a3));
    }
}
// many splits x few elements: one wave per element, lanes stride over the splits
template <typename TO>
__device__ __forceinline__ void reduce_waves(const float* __restrict__ ws, int splits, long n, TO* __restrict__ dw, long lb, long nblk) {
    const int lane = threadIdx.x & 63;
    for (long i = lb * 4 + (threadIdx.x >> 6); i < n; i += nblk * 4) {
        float a = 0.f;
        for (int s = lane; s < splits; s += 64) a += ws[(long)s * n + i];
        a = wave_sum(a);
        if (lane == 0) ElemTraits<TO>::st(dw + i, a);
    }
}
// many splits: 32 consecutive elements x 8 split groups per block -- 128-byte coalesced rows, splits/8 loads per thread, LDS finish
template <typename TO>
__device__ __forceinline__ void reduce_tile(const float* __restrict__ ws, int splits, long n, TO* __restrict__ dw, long lb, float (*part)[33]) {
    const int e = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const long i = lb * 32 + e;
    float a0 = 0.f, a1 = 0.f;
    if (i < n) {
        int s = sg;
        for (; s + 8 < splits; s += 16) { a0 += ws[(long)s * n + i]; a1 += ws[(long)(s + 8) * n + i]; }
        if (s < splits) a0 += ws[(long)s * n + i];
    }
    part[sg][e] = a0 + a1;
    __syncthreads();
    if (sg == 0 && i < n)
        ElemTraits<TO>::st(dw + i, ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) + ((part[4][e] + part[5][e]) + (part[6][e] + part[7][e])));
}

template <typename TO>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, long n, TO* __restrict__ dw) {
    reduce_elems<TO>(ws, splits, n, dw, blockIdx.x, gridDim.x);
}
template <typename TO>
__global__ __launch_bounds__(256) void wgrad_reduce_wave_kernel(const float* __restrict__ ws, int splits, long n, TO* __restrict__ dw) {
    reduce_waves<TO>(ws, splits, n, dw, blockIdx.x, gridDim.x);
}
template <typename TO>
__global__ __launch_bounds__(256) void wgrad_reduce_tile_kernel(const float* __restrict__ ws, int splits, long n, TO* __restrict__ dw) {
    __shared__ float part[8][33];
    reduce_tile<TO>(ws, splits, n, dw, blockIdx.x, part);
}

// ---- parked reductions (mg_conv_wgrad_park / mg_wgrad_reduce_batched) ----------------------------------------------------------
// A backward pass launches ~80 of the reduce kernels above, each a few microseconds of work behind ~6 us of launch: 0.55 ms per step.
// A caller that keeps every layer's slabs until the point where all the weight gradients meet anyway (the batched SpectralNorm backward,
// the weight bank's backward) parks the reduction instead -- the GEMM writes its slabs, the descriptor comes back -- and runs ALL of them
// as one launch there. The table travels by value in the kernel arguments (a captured graph node owns its arguments).
#define RED_FORM_wgrad_reduce_kernel 0
#define RED_FORM_wgrad_reduce_wave_kernel 1
#define RED_FORM_wgrad_reduce_tile_kernel 2
struct ParkState { bool on; const float* ws; long n; int splits, form; long blocks; };
static thread_local ParkState g_park = {false, nullptr, 0, 0, 0, 0};

#define MG_RED_MAX 64
struct RedEntry { const float* ws; void* dw; long n; int splits; int16_t form, dtype; uint32_t blk0, nblk; };
struct RedTable { RedEntry e[MG_RED_MAX]; int count; };

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const RedTable tb) {
    __shared__ float part[8][33];
    int lo = 0, hi = tb.count - 1;                              // last entry whose first block is <= this block (uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blockIdx.x >= tb.e[mid].blk0) lo = mid; else hi = mid - 1;
    }
    const float* __restrict__ ws = tb.e[lo].ws;
    void* dw = tb.e[lo].dw;
    const long n = tb.e[lo].n;
    const int splits = tb.e[lo].splits, form = tb.e[lo].form, dtype = tb.e[lo].dtype;
    const long lb = (long)blockIdx.x - tb.e[lo].blk0, nblk = tb.e[lo].nblk;
#define MG_RED_FORMS(TO)                                                          \
    do {                                                                          \
        if (form == 0) reduce_elems<TO>(ws, splits, n, (TO*)dw, lb, nblk);        \
        else if (form == 1) reduce_waves<TO>(ws, splits, n, (TO*)dw, lb, nblk);   \
        else reduce_tile<TO>(ws, splits, n, (TO*)dw, lb, part);                   \
    } while (0)
    if (dtype == MG_BF16) MG_RED_FORMS(bf16raw);
    else if (dtype == MG_F16) MG_RED_FORMS(f16raw);
    else MG_RED_FORMS(float);
#undef MG_RED_FORMS
}

// split slabs -> dW in the weight-gradient dtype (fp32, bf16 or fp16); parked: only the descriptor is recorded
#define MG_REDUCE_LAUNCH(KERN, B, WS, SPLITS)                                                                                                          \
    do {                                                                                                                                               \
        if (g_park.on) { g_park.ws = (WS); g_park.n = n; g_park.splits = (int)(SPLITS); g_park.form = RED_FORM_##KERN; g_park.blocks = (long)(B); }    \
        else if (p.dw_dtype == MG_BF16) hipLaunchKernelGGL(KERN<bf16raw>, dim3((unsigned)(B)), dim3(256), 0, st, WS, (int)(SPLITS), n, (bf16raw*)p.stats);  \
        else if (p.dw_dtype == MG_F16) hipLaunchKernelGGL(KERN<f16raw>, dim3((unsigned)(B)), dim3(256), 0, st, WS, (int)(SPLITS), n, (f16raw*)p.stats); \
        else hipLaunchKernelGGL(KERN<float>, dim3((unsigned)(B)), dim3(256), 0, st, WS, (int)(SPLITS), n, p.stats);                                    \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// Halo-tile weight gradient for the 3x3 / stride 1 / pad 1 layers (bf16): one block = one (co tile, ci tile) of ALL nine taps over a
// range of 8x16-pixel spatial tiles. The dY tile (128 px) and the x halo tile (10x18 px) are staged once per spatial tile and serve
// nine MFMA sweeps -- the per-tap kernel above re-reads dY nine times and x nine times from L2 (32 FLOP per byte staged; that, not
// MFMA rate, bounded it at ~100 TFLOP/s). The reduction dimension is the pixel index: both operands are read transposed from
// row-major LDS tiles with ds_read_b64_tr_b16, the x rows shifted by the tap's (ky, kx) inside the halo tile. Waves own the four
// quadrants of the block tile (no cross-wave reduction); accumulators leave the registers as fp32 partials, one slab per spatial split.
// ---------------------------------------------------------------------------------------------------------------------
// XF (round 5): the x operand is the RAW output of the producing convolution; act(x * xf_scale + xf_shift) -- the BatchNorm + activation between the
// two convolutions -- is applied to the in-image chunks on their way from the staging registers into LDS (padding stays 0). A thread's chunks all
// sit in one 8-channel group (256 % CPX == 0), so its 16 constants live in registers for the whole walk.
template <int FM, int FN, typename T = bf16raw, bool XF = false>
__global__ __launch_bounds__(256) void igemm_wgrad_halo_kernel(const mg_conv_params p, int tiles_per_block, float* __restrict__ ws) {
    constexpr int TCO = 32 * FM, TCI = 32 * FN;
    constexpr int TH = 8, TW = 16, HW_ = TW + 2, HH = TH + 2;
    constexpr int PY = TCO + 16, PX = TCI + 16;                  // row pitches (elements): 96 / 160 bytes, conflict-free for the 4-row transpose reads
    constexpr int CPY = TCO / 8, CPX = TCI / 8;
    constexpr int NY = TH * TW * CPY, NX = HH * HW_ * CPX;
    constexpr int ITY = (NY + 255) / 256, ITX = (NX + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16raw* sY = (bf16raw*)smem;                                // [TH*TW][PY]
    bf16raw* sX = sY + TH * TW * PY;                             // [HH*HW_][PX]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int wco = (wave >> 1) * FM * 16, wci = (wave & 1) * FN * 16;
    const int nci = p.Cin / TCI, nco = p.Cout / TCO;
    const int tiles_x = (p.Wout + TW - 1) / TW, tiles_y = (p.Hout + TH - 1) / TH;
    const int S = p.N * tiles_y * tiles_x;
    const int nsplit = (S + tiles_per_block - 1) / tiles_per_block;
    int work;
    if (!xcd_order(nsplit * nco * nci, work)) return;
    const int cc = work % (nco * nci), split = work / (nco * nci);
    const int co0 = (cc / nci) * TCO, ci0 = (cc % nci) * TCI;
    const int s_beg = split * tiles_per_block, s_end = min(S, s_beg + tiles_per_block);
    const bf16raw* __restrict__ yb = (const bf16raw*)p.y;
    const bf16raw* __restrict__ xb = (const bf16raw*)p.x;

    // what this thread stages per spatial tile: ITY chunks of dY, ITX chunks of the x halo. The position of a chunk inside the tile is the
    // same for every tile, so its (row, column, element offset) are computed once; per tile only the origin and two bound checks remain.
    int y_ty[ITY], y_tx[ITY], y_off[ITY], x_hy[ITX], x_hx[ITX], x_off[ITX];
#pragma unroll
    for (int i = 0; i < ITY; ++i) {
        const int idx = t + i * 256;
        const int px = idx / CPY, c = idx - px * CPY;
        y_ty[i] = idx < NY ? px / TW : (1 << 20); y_tx[i] = px % TW;
        y_off[i] = idx < NY ? (y_ty[i] * p.Wout + y_tx[i]) * p.ldy + c * 8 : 0;
    }
#pragma unroll
    for (int i = 0; i < ITX; ++i) {
        const int idx = t + i * 256;
        const int px = idx / CPX, c = idx - px * CPX;
        x_hy[i] = idx < NX ? px / HW_ - 1 : (1 << 20); x_hx[i] = px % HW_ - 1;
        x_off[i] = idx < NX ? (x_hy[i] * p.Win + x_hx[i]) * p.ldx + c * 8 : 0;
    }
    uint4 ry[ITY], rx[ITX];
    [[maybe_unused]] float xsc[8], xsh[8];
    [[maybe_unused]] unsigned xlive = 0;                       // bit i: chunk i of the x halo is an in-image pixel (transformed), else padding (0)
    [[maybe_unused]] const float xsl = xf_slope_of(p.xf_act, p.xf_slope);
    if constexpr (XF) {
        static_assert(256 % CPX == 0, "a thread's x chunks must share one channel group");
        const int c = t % CPX;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xsc[e] = p.xf_scale[ci0 + c * 8 + e]; xsh[e] = p.xf_shift[ci0 + c * 8 + e]; }
    }
    auto load_tile = [&](int s) {
        const int n = s / (tiles_y * tiles_x);
        const int r = s - n * tiles_y * tiles_x;
        const int y0 = (r / tiles_x) * TH, x0 = (r % tiles_x) * TW;
        const bf16raw* ybase = yb + ((long)(n * p.Hout + y0) * p.Wout + x0) * p.ldy + p.yoff + co0;
        const bf16raw* xbase = xb + ((long)(n * p.Hin + y0) * p.Win + x0) * p.ldx + ci0;
#pragma unroll
        for (int i = 0; i < ITY; ++i) {
            uint4 q = make_uint4(0, 0, 0, 0);
            if (y0 + y_ty[i] < p.Hout && x0 + y_tx[i] < p.Wout) q = *(const uint4*)(ybase + y_off[i]);
            ry[i] = q;
        }
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            uint4 q = make_uint4(0, 0, 0, 0);
            const bool in_img = (unsigned)(y0 + x_hy[i]) < (unsigned)p.Hin && (unsigned)(x0 + x_hx[i]) < (unsigned)p.Win;
            if (in_img) q = *(const uint4*)(xbase + x_off[i]);
            if constexpr (XF) xlive = in_img ? (xlive | (1u << i)) : (xlive & ~(1u << i));
            rx[i] = q;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < ITY; ++i) {
            const int idx = t + i * 256;
            const int px = idx / CPY, c = idx - px * CPY;
            if (idx < NY) *(uint4*)(sY + px * PY + c * 8) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            const int idx = t + i * 256;
            const int px = idx / CPX, c = idx - px * CPX;
            if constexpr (XF) { if (xlive & (1u << i)) rx[i] = xf_apply8<T>(rx[i], xsc, xsh, xsl); }
            if (idx < NX) *(uint4*)(sX + px * PX + c * 8) = rx[i];
        }
    };

    f32x4 acc[9][FM][FN];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[tp][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int q = g * 4 + (li >> 2), cq = (li & 3) * 4;          // this lane's pixel column inside a 16-px tile row, and its 4-channel group
    MG_STAMP(0);
    if (s_beg < s_end) load_tile(s_beg);
    MG_STAMP(1);
    for (int s = s_beg; s < s_end; ++s) {
        if (s - s_beg < 6) MG_STAMP(2 + 2 * (s - s_beg));
        store_tile();
        __syncthreads();
        if (s - s_beg < 6) MG_STAMP(3 + 2 * (s - s_beg));
        if (s + 1 < s_end) load_tile(s + 1);
#pragma unroll
        for (int kc = 0; kc < TH / 2; ++kc) {                    // 32 pixels (two tile rows) per MFMA K step
            s16x4 a[FM][2];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                a[i][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + ((2 * kc) * TW + q) * PY + wco + i * 16 + cq));
                a[i][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + ((2 * kc + 1) * TW + q) * PY + wco + i * 16 + cq));
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    s16x4 b[FN][2];
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        b[j][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + ((2 * kc + ky) * HW_ + q + kx) * PX + wci + j * 16 + cq));
                        b[j][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + ((2 * kc + 1 + ky) * HW_ + q + kx) * PX + wci + j * 16 + cq));
                    }
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            union { s16x4 h[2]; uint4 v; } ua, ub;
                            ua.h[0] = a[i][0]; ua.h[1] = a[i][1]; ub.h[0] = b[j][0]; ub.h[1] = b[j][1];
                            acc[ky * 3 + kx][i][j] = mfma16<T>(ua.v, ub.v, acc[ky * 3 + kx][i][j]);
                        }
                }
        }
        __syncthreads();
    }
    MG_STAMP(14);
    float* __restrict__ slab = ws + (long)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    slab[((long)(co0 + wco + i * 16 + g * 4 + e) * 9 + tp) * p.Cin + ci0 + wci + j * 16 + li] = acc[tp][i][j][e];
    MG_STAMP(15);
}

// ---------------------------------------------------------------------------------------------------------------------
// All-taps weight gradient of the sparse head's 3x3 gather convolutions (MG_MODE_GATHER, bf16): one block = one (co tile, 32-channel ci
// tile) of all NINE taps over a range of active rows. Per 64-row stage the dY rows are staged ONCE and the nine neighbour rows of x are
// gathered through the neighbour table into nine LDS tiles; the per-tap kernel above re-reads dY nine times (it is L2-bandwidth bound:
// 370 MB per C64 launch). Same MFMA structure as the halo kernel (transposed LDS reads, waves own quadrants, fp32 slabs per row split); the
// row count is a device word: the fixed grid divides the live rows evenly, empty splits write zero slabs.
// ---------------------------------------------------------------------------------------------------------------------
template <int FM, typename T = bf16raw, bool XF = false>
__global__ __launch_bounds__(256) void igemm_wgrad_gather9_kernel(const mg_conv_params p, int nsplit, float* __restrict__ ws) {
    constexpr int FN = 1, TCO = 32 * FM, TCI = 32, RC = 64;
    constexpr int PY = TCO + 16, PX = TCI + 16;
    constexpr int CPY = TCO / 8, CPX = TCI / 8;
    constexpr int NY = RC * CPY, NX = 9 * RC * CPX;
    constexpr int ITY = (NY + 255) / 256, ITX = NX / 256;        // 9 * 64 * 4 / 256 = 9
    static_assert(NX % 256 == 0, "gather staging");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16raw* sY = (bf16raw*)smem;                                // [RC][PY]
    bf16raw* sX = sY + RC * PY;                                  // [9][RC][PX]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int wco = (wave >> 1) * FM * 16, wci = (wave & 1) * FN * 16;
    const int nci = p.Cin / TCI, nco = p.Cout / TCO;
    int work;
    if (!xcd_order(nsplit * nco * nci, work)) return;
    const int cc = work % (nco * nci), split = work / (nco * nci);
    const int co0 = (cc / nci) * TCO, ci0 = (cc % nci) * TCI;
    const int M = dev_rows(p.m_dev, p.M);
    const int rps = (((M + nsplit - 1) / nsplit + RC - 1) / RC) * RC;
    const int mbeg = split * rps, mend = min(M, mbeg + rps);
    const bf16raw* __restrict__ yb = (const bf16raw*)p.y;
    const bf16raw* __restrict__ xb = (const bf16raw*)p.x;

    uint4 ry[ITY], rx[ITX];
    [[maybe_unused]] float xsc[8], xsh[8];
    [[maybe_unused]] unsigned xlive = 0;                          // bit i: gathered chunk i exists (a live neighbour) -> transformed; missing neighbours stay 0
    [[maybe_unused]] const float xsl = xf_slope_of(p.xf_act, p.xf_slope);
    if constexpr (XF) {
        static_assert(256 % CPX == 0 && ITX <= 32, "operand transform: one channel group per thread");
        const int c = t % CPX;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xsc[e] = p.xf_scale[ci0 + c * 8 + e]; xsh[e] = p.xf_shift[ci0 + c * 8 + e]; }
    }
    auto load_stage = [&](int mb) {
#pragma unroll
        for (int i = 0; i < ITY; ++i) {
            const int idx = t + i * 256;
            const int row = idx / CPY, c = idx - row * CPY;
            const int m = mb + row;
            ry[i] = (idx < NY && m < mend) ? *(const uint4*)(yb + (long)m * p.ldy + p.yoff + co0 + c * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            const int idx = t + i * 256;                         // (tap, row, chunk): chunk fastest, then row, then tap
            const int c = idx % CPX, row = (idx / CPX) % RC, tap = idx / (CPX * RC);
            const int m = mb + row;
            int src = -1;
            if (m < mend) src = p.nbr[(long)m * 9 + tap];
            rx[i] = src >= 0 ? *(const uint4*)(xb + (long)src * p.ldx + ci0 + c * 8) : make_uint4(0, 0, 0, 0);
            if constexpr (XF) xlive = src >= 0 ? (xlive | (1u << i)) : (xlive & ~(1u << i));
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < ITY; ++i) {
            const int idx = t + i * 256;
            const int row = idx / CPY, c = idx - row * CPY;
            if (idx < NY) *(uint4*)(sY + row * PY + c * 8) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < ITX; ++i) {
            const int idx = t + i * 256;
            const int c = idx % CPX, row = (idx / CPX) % RC, tap = idx / (CPX * RC);
            if constexpr (XF) { if (xlive & (1u << i)) rx[i] = xf_apply8<T>(rx[i], xsc, xsh, xsl); }
            *(uint4*)(sX + (tap * RC + row) * PX + c * 8) = rx[i];
        }
    };

    f32x4 acc[9][FM][FN];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[tp][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int q = g * 4 + (li >> 2), cq = (li & 3) * 4;
    if (mbeg < mend) load_stage(mbeg);
    for (int mb = mbeg; mb < mend; mb += RC) {
        store_stage();
        __syncthreads();
        if (mb + RC < mend) load_stage(mb + RC);
#pragma unroll
        for (int kc = 0; kc < RC / 32; ++kc) {
            s16x4 a[FM][2];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                a[i][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + (kc * 32 + q) * PY + wco + i * 16 + cq));
                a[i][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + (kc * 32 + 16 + q) * PY + wco + i * 16 + cq));
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                s16x4 b[FN][2];
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    b[j][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + (tp * RC + kc * 32 + q) * PX + wci + j * 16 + cq));
                    b[j][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + (tp * RC + kc * 32 + 16 + q) * PX + wci + j * 16 + cq));
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        union { s16x4 h[2]; uint4 v; } ua, ub;
                        ua.h[0] = a[i][0]; ua.h[1] = a[i][1]; ub.h[0] = b[j][0]; ub.h[1] = b[j][1];
                        acc[tp][i][j] = mfma16<T>(ua.v, ub.v, acc[tp][i][j]);
                    }
            }
        }
        __syncthreads();
    }
    float* __restrict__ slab = ws + (long)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    slab[((long)(co0 + wco + i * 16 + g * 4 + e) * 9 + tp) * p.Cin + ci0 + wci + j * 16 + li] = acc[tp][i][j][e];
}

static inline bool wgrad_gather9_eligible(const mg_conv_params& p) {
    static const int on = [] { const char* e = getenv("MG_WGRAD_GATHER9"); return e ? atoi(e) : 1; }();
    // Cin >= 64 only: at Cin 32 (the OS1 level: ~20 stages of 64 rows per block, 18 MFMAs per stage) the kernel is bound by the gather latency
    // of its short stages and the per-tap kernel's 128-row steps win (measured 52 -> 64 us); at Cin 64: 55 -> 27, 54 -> 41, 32 -> 23 us
    return on && MG_IS16(p.dtype) && p.mode == MG_MODE_GATHER && p.nbr && p.R * p.S == 9 && p.Cin % 64 == 0 && p.Cout % 32 == 0 &&
           p.ldx % 8 == 0 && p.ldy % 8 == 0 && p.yoff % 8 == 0 && (long)p.Cout * 9 * p.Cin <= (16l << 20) && p.M >= 256;
}
static long plan_wgrad_gather9(const mg_conv_params& p) {
    // ~512 workgroups (two fit a CU), bounded by ~20 MB of fp32 partial slabs (written once, read once by the reduce)
    static const long target = [] { const char* e = getenv("MG_WGRAD_GATHER_BLOCKS"); return e ? atol(e) : 512l; }();
    static const long ws_cap = [] { const char* e = getenv("MG_WGRAD_GATHER_WS_MB"); return (e ? atol(e) : 20l) << 18; }();   // floats
    const int tco = p.Cout % 64 == 0 ? 64 : 32;
    const long cc = (long)(p.Cout / tco) * (p.Cin / 32);
    const long n = (long)p.Cout * 9 * p.Cin;
    long splits = (target + cc - 1) / cc;
    const long by_rows = (p.M + 127) / 128;
    if (splits > by_rows) splits = by_rows;
    if (splits > ws_cap / n) splits = ws_cap / n;
    return splits < 1 ? 1 : splits;
}
static int launch_wgrad_gather9(const mg_conv_params& p, float* ws, long ws_floats, hipStream_t st) {
    const long splits = plan_wgrad_gather9(p);
    const long n = (long)p.Cout * 9 * p.Cin;
    const bool out_bf16 = MG_IS16(p.dw_dtype);
    if (!ws || ws_floats < splits * n) return -4;
    const int tco = p.Cout % 64 == 0 ? 64 : 32;
    const long cc = (long)(p.Cout / tco) * (p.Cin / 32);
    dim3 grid(xcd_grid(splits * cc));
    const size_t lds = (size_t)(64 * (tco + 16) + 9 * 64 * (32 + 16)) * sizeof(bf16raw);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_gather9_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_gather9_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_gather9_kernel<1, f16raw>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)igemm_wgrad_gather9_kernel<2, f16raw>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    if (p.xf_scale) {                                        // x = a raw conv output, BatchNorm1d + activation applied on the way into LDS
        if (tco != 32) return MG_XF_UNSUPPORTED;
        static bool xattr = false;
        if (!xattr) {
            (void)hipFuncSetAttribute((const void*)igemm_wgrad_gather9_kernel<1, bf16raw, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void*)igemm_wgrad_gather9_kernel<1, f16raw, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            xattr = true;
        }
        if (p.dtype == MG_F16) hipLaunchKernelGGL((igemm_wgrad_gather9_kernel<1, f16raw, true>), grid, dim3(256), lds, st, p, (int)splits, ws);
        else hipLaunchKernelGGL((igemm_wgrad_gather9_kernel<1, bf16raw, true>), grid, dim3(256), lds, st, p, (int)splits, ws);
    } else if (p.dtype == MG_F16) {
        if (tco == 64) hipLaunchKernelGGL((igemm_wgrad_gather9_kernel<2, f16raw>), grid, dim3(256), lds, st, p, (int)splits, ws);
        else hipLaunchKernelGGL((igemm_wgrad_gather9_kernel<1, f16raw>), grid, dim3(256), lds, st, p, (int)splits, ws);
    } else if (tco == 64) hipLaunchKernelGGL((igemm_wgrad_gather9_kernel<2>), grid, dim3(256), lds, st, p, (int)splits, ws);
    else hipLaunchKernelGGL((igemm_wgrad_gather9_kernel<1>), grid, dim3(256), lds, st, p, (int)splits, ws);
    if (splits >= 8) {
        const long b = (n + 31) / 32;
        MG_REDUCE_LAUNCH(wgrad_reduce_tile_kernel, b, ws, splits);
    } else {
        long b = (n + 255) / 256; if (b > 2048) b = 2048;
        MG_REDUCE_LAUNCH(wgrad_reduce_kernel, b, ws, splits);
    }
    MG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride 1 convs that read the 8-CHANNEL network input (Cin == 8: 16 bytes per pixel), bf16. The layer is
// HBM-bound (1 M pixels: 64 MB of dY + 16 MB of x against 0.15 GFLOP), the per-tap kernel re-read both nine times (110 us). Here a block stages
// a dY tile (8x16 px x 32 co) and the x halo tile (10x18 px x 16 B) once; the GEMM's N dimension is (tap, ci): sixteen columns = the 8 channels
// of two horizontally adjacent taps = 32 contiguous bytes of the halo row, so the transposed LDS read of the halo image yields the operand
// directly. Wave w owns output-channel tile (w & 1) and tap pair (w >> 1) of all three filter rows (kx = 2 of pair 1 has a junk second half).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T = bf16raw>
__global__ __launch_bounds__(256) void igemm_wgrad_c8_kernel(const mg_conv_params p, int tiles_per_block, float* __restrict__ ws) {
    constexpr int TCO = 32, TH = 8, TW = 16, HW_ = TW + 2, HH = TH + 2;
    constexpr int PY = TCO + 16;
    constexpr int NY = TH * TW * (TCO / 8), NX = HH * HW_;
    constexpr int ITY = NY / 256;                                // 2
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16raw* sY = (bf16raw*)smem;                                // [TH*TW][PY]
    bf16raw* sX = sY + TH * TW * PY;                             // [HH*HW_ + 4][8]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int mt = wave & 1, pair = wave >> 1;
    const int nco = p.Cout / TCO;
    const int tiles_x = (p.Wout + TW - 1) / TW, tiles_y = (p.Hout + TH - 1) / TH;
    const int S = p.N * tiles_y * tiles_x;
    const int nsplit = (S + tiles_per_block - 1) / tiles_per_block;
    int work;
    if (!xcd_order(nsplit * nco, work)) return;
    const int co0 = (work % nco) * TCO, split = work / nco;
    const int s_beg = split * tiles_per_block, s_end = min(S, s_beg + tiles_per_block);
    const bf16raw* __restrict__ yb = (const bf16raw*)p.y;
    const bf16raw* __restrict__ xb = (const bf16raw*)p.x;

    int y_ty[ITY], y_tx[ITY], y_off[ITY];
#pragma unroll
    for (int i = 0; i < ITY; ++i) {
        const int idx = t + i * 256;
        const int px = idx >> 2, c = idx & 3;
        y_ty[i] = px / TW; y_tx[i] = px % TW;
        y_off[i] = (y_ty[i] * p.Wout + y_tx[i]) * p.ldy + c * 8;
    }
    const int x_hy = t < NX ? t / HW_ - 1 : (1 << 20), x_hx = t % HW_ - 1;
    const int x_off = t < NX ? (x_hy * p.Win + x_hx) * p.ldx : 0;
    if (t < 4) *(uint4*)(sX + (NX + t) * 8) = make_uint4(0, 0, 0, 0);       // the pad pixels behind the halo image (read by the junk half of pair 1)
    uint4 ry[ITY], rx;
    auto load_tile = [&](int s) {
        const int n = s / (tiles_y * tiles_x);
        const int r = s - n * tiles_y * tiles_x;
        const int y0 = (r / tiles_x) * TH, x0 = (r % tiles_x) * TW;
        const bf16raw* ybase = yb + ((long)(n * p.Hout + y0) * p.Wout + x0) * p.ldy + p.yoff + co0;
        const bf16raw* xbase = xb + ((long)(n * p.Hin + y0) * p.Win + x0) * p.ldx;
#pragma unroll
        for (int i = 0; i < ITY; ++i)
            ry[i] = (y0 + y_ty[i] < p.Hout && x0 + y_tx[i] < p.Wout) ? *(const uint4*)(ybase + y_off[i]) : make_uint4(0, 0, 0, 0);
        rx = ((unsigned)(y0 + x_hy) < (unsigned)p.Hin && (unsigned)(x0 + x_hx) < (unsigned)p.Win) ? *(const uint4*)(xbase + x_off) : make_uint4(0, 0, 0, 0);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < ITY; ++i) {
            const int idx = t + i * 256;
            *(uint4*)(sY + (idx >> 2) * PY + (idx & 3) * 8) = ry[i];
        }
        if (t < NX) *(uint4*)(sX + t * 8) = rx;
    };

    f32x4 acc[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) acc[ky] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int q = g * 4 + (li >> 2), cq = (li & 3) * 4;
    if (s_beg < s_end) load_tile(s_beg);
    for (int s = s_beg; s < s_end; ++s) {
        store_tile();
        __syncthreads();
        if (s + 1 < s_end) load_tile(s + 1);
#pragma unroll
        for (int kc = 0; kc < TH / 2; ++kc) {
            union { s16x4 h[2]; uint4 v; } ua, ub;
            ua.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + ((2 * kc) * TW + q) * PY + mt * 16 + cq));
            ua.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sY + ((2 * kc + 1) * TW + q) * PY + mt * 16 + cq));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                ub.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + ((2 * kc + ky) * HW_ + q + 2 * pair) * 8 + cq));
                ub.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sX + ((2 * kc + 1 + ky) * HW_ + q + 2 * pair) * 8 + cq));
                acc[ky] = mfma16<T>(ua.v, ub.v, acc[ky]);
            }
        }
        __syncthreads();
    }
    float* __restrict__ slab = ws + (long)split * p.Cout * 72;
    const int kx = 2 * pair + (li >> 3), ci = li & 7;
    if (kx < 3) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[((long)(co0 + mt * 16 + g * 4 + e) * 9 + ky * 3 + kx) * 8 + ci] = acc[ky][e];
    }
}

static inline bool wgrad_c8_eligible(const mg_conv_params& p) {
    static const int on = [] { const char* e = getenv("MG_WGRAD_C8"); return e ? atoi(e) : 1; }();
    return on && MG_IS16(p.dtype) && p.mode == MG_MODE_CONV && !p.m_dev && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && p.dil == 1 &&
           p.Hout == p.Hin && p.Wout == p.Win && p.Cin == 8 && p.Cout % 32 == 0 && p.ldx % 8 == 0 && p.ldy % 8 == 0 && p.yoff % 8 == 0 &&
           p.Wout >= 16 && p.Hout >= 8;
}
static long plan_wgrad_c8(const mg_conv_params& p, int* tpb_out) {
    const long S = (long)p.N * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
    const long nco = p.Cout / 32;
    long splits = (512 + nco - 1) / nco;
    if (splits > S) splits = S;
    if (splits < 1) splits = 1;
    const int tpb = (int)((S + splits - 1) / splits);
    splits = (S + tpb - 1) / tpb;
    if (tpb_out) *tpb_out = tpb;
    return splits;
}
static int launch_wgrad_c8(const mg_conv_params& p, float* ws, long ws_floats, hipStream_t st) {
    int tpb = 1;
    const long splits = plan_wgrad_c8(p, &tpb);
    const long n = (long)p.Cout * 72;
    const bool out_bf16 = MG_IS16(p.dw_dtype);
    if (!ws || ws_floats < splits * n) return -4;
    dim3 grid(xcd_grid(splits * (p.Cout / 32)));
    const size_t lds = (size_t)(8 * 16 * (32 + 16) + (10 * 18 + 4) * 8) * sizeof(bf16raw);
    if (p.dtype == MG_F16) hipLaunchKernelGGL(igemm_wgrad_c8_kernel<f16raw>, grid, dim3(256), lds, st, p, tpb, ws);
    else hipLaunchKernelGGL(igemm_wgrad_c8_kernel<bf16raw>, grid, dim3(256), lds, st, p, tpb, ws);
    const long b = (n + 31) / 32;
    MG_REDUCE_LAUNCH(wgrad_reduce_tile_kernel, b, ws, splits);
    MG_CHECK_LAUNCH();
    return 0;
}

static inline bool wgrad_halo_eligible(const mg_conv_params& p) {
    static const int on = [] { const char* e = getenv("MG_WGRAD_HALO"); return e ? atoi(e) : 1; }();
    return on && MG_IS16(p.dtype) && p.mode == MG_MODE_CONV && !p.m_dev && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && p.dil == 1 &&
           p.Hout == p.Hin && p.Wout == p.Win && p.Cin % 32 == 0 && p.Cout % 32 == 0 && p.ldx % 8 == 0 && p.ldy % 8 == 0 && p.yoff % 8 == 0 &&
           p.Wout >= 16 && p.Hout >= 8 && (long)p.Cout * 9 * p.Cin <= (16l << 20);
}

struct WgradHaloPlan { long splits; int tpb; };

static WgradHaloPlan plan_wgrad_halo(const mg_conv_params& p) {
    const long cc = (long)(p.Cin / 32) * (p.Cout / 32);
    const long S = (long)p.N * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
    // ~one workgroup per CU: with the slab reductions parked (mg_conv_wgrad_park) the slabs come back from HBM, not from the Infinity Cache, and
    // every spatial split is 36 KB per (co, ci) tile written and read once more -- 256 against 512: 11.61 against 11.70 ms per step (3 x 200-step
    // pairs on one lease; 128: 11.83)
    static const long target = [] { const char* e = getenv("MG_WGRAD_HALO_BLOCKS"); return e ? atol(e) : 256l; }();
    const long n = (long)p.Cout * 9 * p.Cin;
    long splits = (target + cc - 1) / cc;
    if (splits > S) splits = S;
    if (splits > (16l << 20) / n) splits = (16l << 20) / n;
    if (splits < 1) splits = 1;
    const int tpb = (int)((S + splits - 1) / splits);
    splits = (S + tpb - 1) / tpb;
    return {splits, tpb};
}

static int launch_wgrad_halo(const mg_conv_params& p, float* ws, long ws_floats, hipStream_t st) {
    const WgradHaloPlan pl = plan_wgrad_halo(p);
    const long n = (long)p.Cout * 9 * p.Cin;
    const bool out_bf16 = MG_IS16(p.dw_dtype);
    float* use_ws = nullptr;
    if ((pl.splits > 1 || out_bf16) && ws && ws_floats >= pl.splits * n) use_ws = ws;
    if (out_bf16 && !use_ws) return -4;
    if (pl.splits > 1 && !use_ws) return -4;                       // the halo form has no atomic fallback: callers size the workspace first
    if (pl.splits == 1 && !out_bf16) use_ws = p.stats;
    const long cc = (long)(p.Cin / 32) * (p.Cout / 32);
    dim3 grid(xcd_grid(pl.splits * cc));
    const size_t lds = (size_t)(8 * 16 * (32 + 16) + 10 * 18 * (32 + 16)) * sizeof(bf16raw);
    if (p.xf_scale) {
        if (p.dtype == MG_F16) hipLaunchKernelGGL((igemm_wgrad_halo_kernel<1, 1, f16raw, true>), grid, dim3(256), lds, st, p, pl.tpb, use_ws);
        else hipLaunchKernelGGL((igemm_wgrad_halo_kernel<1, 1, bf16raw, true>), grid, dim3(256), lds, st, p, pl.tpb, use_ws);
    } else if (p.dtype == MG_F16) hipLaunchKernelGGL((igemm_wgrad_halo_kernel<1, 1, f16raw>), grid, dim3(256), lds, st, p, pl.tpb, use_ws);
    else hipLaunchKernelGGL((igemm_wgrad_halo_kernel<1, 1>), grid, dim3(256), lds, st, p, pl.tpb, use_ws);
    if (use_ws != p.stats) {
        if (pl.splits >= 8) {
            const long b = (n + 31) / 32;
            MG_REDUCE_LAUNCH(wgrad_reduce_tile_kernel, b, use_ws, pl.splits);
        } else {
            long b = (n + 255) / 256; if (b > 2048) b = 2048;
            MG_REDUCE_LAUNCH(wgrad_reduce_kernel, b, use_ws, pl.splits);
        }
    }
    MG_CHECK_LAUNCH();
    return 0;
}

struct WgradPlan { long splits; int rpb; };

template <typename T, int TCO, int TCI>
WgradPlan plan_wgrad(const mg_conv_params& p) {
    constexpr int KSTEP = sizeof(T) == 2 ? 128 : 64;
    const int taps = p.R * p.S;
    const int nci = (p.Cin + TCI - 1) / TCI, nco = (p.Cout + TCO - 1) / TCO;
    const long tiles = (long)taps * nci * nco;
    // design point: ~8 row steps per block (amortises the tile epilogue), bounded by ~2048 blocks, 512 splits and a 64 MB
    // partial-tile workspace; never fewer blocks than ~1 per CU when the rows allow it
    static const long target = [] { const char* e = getenv("MG_WGRAD_BLOCKS"); return e ? atol(e) : 256l; }();
    const long n = (long)p.Cout * taps * p.Cin;
    long splits = p.M / (8 * KSTEP);
    // once the rows are split anyway (a reduce pass exists), ~3 blocks per CU hide more latency: +8 % on the C128 / C256 layers;
    // a layer that fits one split stays unsplit (no workspace round trip)
    static const long target_split = [] { const char* e = getenv("MG_WGRAD_BLOCKS_SPLIT"); return e ? atol(e) : 768l; }();
    long lo = ((splits > 1 ? target_split : target) + tiles - 1) / tiles;
    long by_rows = (p.M + 2 * KSTEP - 1) / (2 * KSTEP);           // at least 2 steps per block
    if (lo > by_rows) lo = by_rows;
    if (splits < lo) splits = lo;
    if (splits > 2048 / tiles) splits = 2048 / tiles;
    if (splits > 512) splits = 512;
    if (splits > (16l << 20) / n) splits = (16l << 20) / n;
    if (p.m_dev) {
        // sparse head: M is the CAPACITY (every site of the frame); the live rows are typically 10-20 % of it and are divided evenly over the
        // splits in-kernel, so a capacity-sized split count only buys zero slabs (33 MB written + 33 MB re-read per C64 launch; step time 15.20 / 15.10 / 15.10 / 15.18 ms at 512 / 128 / 64 / 32)
        static const long dev_splits = [] { const char* e = getenv("MG_WGRAD_DEV_SPLITS"); return e ? atol(e) : 128l; }();
        const long cap = (dev_splits * 9 + tiles - 1) / tiles;            // ~dev_splits row ranges for a 3x3 layer's 9 tap tiles
        if (splits > cap) splits = cap;
    }
    if (splits < 1) splits = 1;
    int rpb = (int)((p.M + splits - 1) / splits);
    rpb = ((rpb + KSTEP - 1) / KSTEP) * KSTEP;
    splits = (p.M + rpb - 1) / rpb;
    return {splits, rpb};
}

template <typename T, int TCO, int TCI>
int launch_wgrad(const mg_conv_params& p, float* ws, long ws_floats, hipStream_t st) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int KSTEP = BF ? 128 : 64;
    constexpr int PAD = BF ? 8 : 16;
    const int taps = p.R * p.S;
    const int nci = (p.Cin + TCI - 1) / TCI, nco = (p.Cout + TCO - 1) / TCO;
    WgradPlan pl = plan_wgrad<T, TCO, TCI>(p);
    const long n = (long)p.Cout * taps * p.Cin;
    const bool out_bf16 = MG_IS16(p.dw_dtype);                   // dW in bf16: always partials -> (converting) reduce
    float* use_ws = nullptr;
    if ((pl.splits > 1 || out_bf16) && ws && ws_floats >= pl.splits * n) use_ws = ws;
    if (out_bf16 && !use_ws) return -4;
    mg_conv_params q = p;
    if (pl.splits == 1 && !out_bf16) use_ws = p.stats;             // single split: the "slab" is dW itself (no atomics, no reduce)
    dim3 grid(xcd_grid((long)pl.splits * taps * nci * nco));
    size_t stage = (size_t)KSTEP * (TCO + PAD + TCI + PAD) * sizeof(T);
    size_t red = (size_t)4 * TCO * (TCI + 1) * 4;
    size_t lds = stage > red ? stage : red;
    if (p.xf_scale) {
        if constexpr (BF && TCO == 32 && TCI == 32) {
            if (p.mode == MG_MODE_CONV) hipLaunchKernelGGL((igemm_wgrad_kernel<T, TCO, TCI, MG_MODE_CONV, true>), grid, dim3(256), lds, st, q, pl.rpb, use_ws);
            else if (p.mode == MG_MODE_GATHER) hipLaunchKernelGGL((igemm_wgrad_kernel<T, TCO, TCI, MG_MODE_GATHER, true>), grid, dim3(256), lds, st, q, pl.rpb, use_ws);
            else return MG_XF_UNSUPPORTED;
        } else return MG_XF_UNSUPPORTED;
    } else
    switch (p.mode) {
        case MG_MODE_CONV: hipLaunchKernelGGL((igemm_wgrad_kernel<T, TCO, TCI, MG_MODE_CONV>), grid, dim3(256), lds, st, q, pl.rpb, use_ws); break;
        case MG_MODE_TCONV: hipLaunchKernelGGL((igemm_wgrad_kernel<T, TCO, TCI, MG_MODE_TCONV>), grid, dim3(256), lds, st, q, pl.rpb, use_ws); break;
        case MG_MODE_GATHER: hipLaunchKernelGGL((igemm_wgrad_kernel<T, TCO, TCI, MG_MODE_GATHER>), grid, dim3(256), lds, st, q, pl.rpb, use_ws); break;
        default: return -2;
    }
    if (use_ws && (pl.splits > 1 || out_bf16)) {
        if (pl.splits > 32 && n <= (1l << 16)) {
            long b = (n + 3) / 4; if (b > 8192) b = 8192;
            MG_REDUCE_LAUNCH(wgrad_reduce_wave_kernel, b, use_ws, pl.splits);
        } else {
            long b = (n + 255) / 256; if (b > 2048) b = 2048;
            MG_REDUCE_LAUNCH(wgrad_reduce_kernel, b, use_ws, pl.splits);
        }
    }
    MG_CHECK_LAUNCH();
    return 0;
}

// operand transform (mg_conv_params.xf_*): the halo form applies it between its staging registers and LDS
// ... and the sparse head's row matrices (device row count): gather 3x3 through the all-taps kernel (Cin 64) or the per-tap kernel (Cin 32), 1x1
// through the per-tap kernel (Cin 32); Cout <= 32 (32-wide co tiles)
static inline bool wgrad_xf_rows_ok(const mg_conv_params& p) {
    if (!MG_IS16(p.dtype) || !p.m_dev || p.Cout > 32 || p.ldx % 8) return false;
    if (p.mode == MG_MODE_GATHER && p.R * p.S == 9) return p.Cin == 32 || (p.Cin == 64 && wgrad_gather9_eligible(p));
    return p.mode == MG_MODE_CONV && p.R * p.S == 1 && p.stride == 1 && p.pad == 0 && p.Cin == 32;
}
static inline bool wgrad_xf_ok(const mg_conv_params& p) {
    return (MG_IS16(p.dtype) && wgrad_halo_eligible(p) && !wgrad_c8_eligible(p) && !wgrad_gather9_eligible(p)) || wgrad_xf_rows_ok(p);
}

template <typename T>
int dispatch_wgrad(const mg_conv_params& p, float* ws, long ws_floats, long* need, hipStream_t st) {
    const bool small_co = p.Cout <= 32, small_ci = p.Cin <= 32;
    const long n = (long)p.Cout * p.R * p.S * p.Cin;
    if (p.xf_scale && !wgrad_xf_ok(p)) { if (need) { *need = 0; return 0; } return MG_XF_UNSUPPORTED; }
    if (sizeof(T) == 2 && wgrad_c8_eligible(p)) {
        const long splits = plan_wgrad_c8(p, nullptr);
        if (need) { *need = splits * n; return 0; }
        if (ws && ws_floats >= splits * n) return launch_wgrad_c8(p, ws, ws_floats, st);
    }
    if (sizeof(T) == 2 && wgrad_gather9_eligible(p)) {
        const long splits = plan_wgrad_gather9(p);
        if (need) { *need = splits * n; return 0; }
        if (ws && ws_floats >= splits * n) return launch_wgrad_gather9(p, ws, ws_floats, st);
    }
    if (sizeof(T) == 2 && wgrad_halo_eligible(p)) {
        if (need) { const WgradHaloPlan pl = plan_wgrad_halo(p); *need = (pl.splits > 1 || MG_IS16(p.dw_dtype)) ? pl.splits * n : 0; return 0; }
        const WgradHaloPlan pl = plan_wgrad_halo(p);
        if (pl.splits == 1 || (ws && ws_floats >= pl.splits * n)) return launch_wgrad_halo(p, ws, ws_floats, st);
    }
    if (p.xf_scale && !wgrad_xf_rows_ok(p)) return MG_XF_UNSUPPORTED;      // (no workspace for the halo form: the per-tap kernels transform only the row-matrix forms)
#define MG_WG(TCO, TCI)                                                                             \
    do {                                                                                            \
        if (need) { WgradPlan pl = plan_wgrad<T, TCO, TCI>(p); *need = (pl.splits > 1 || MG_IS16(p.dw_dtype)) ? pl.splits * n : 0; return 0; } \
        return launch_wgrad<T, TCO, TCI>(p, ws, ws_floats, st);                                     \
    } while (0)
    if (small_co && small_ci) MG_WG(32, 32);
    if (small_co) MG_WG(32, 64);
    if (small_ci) MG_WG(64, 32);
    MG_WG(64, 64);
#undef MG_WG
}

int wgrad_check(const mg_conv_params* pp) {
    if (!pp) return -1;
    const mg_conv_params& p = *pp;
    const int ce = MG_IS16(p.dtype) ? 8 : 4;
    if (p.Cin % ce != 0 || p.ldx % ce != 0) return -3;
    if (p.mode == MG_MODE_GATHER && !p.nbr) return -5;
    if (!MG_IS16(p.dtype) && p.dtype != MG_F32) return -6;
    if (MG_IS16(p.dw_dtype) && p.dw_dtype != p.dtype) return -6;           // 16-bit dW comes in the activations' own 16-bit type
    return 0;
}

}  // namespace

bool mg_wgrad_xform_ok(const mg_conv_params& p) { return wgrad_xf_ok(p); }

// floats of workspace that make mg_conv_wgrad_ws deterministic and atomic-free for this geometry (0 = none needed)
extern "C" long mg_conv_wgrad_workspace(const mg_conv_params* pp) {
    if (wgrad_check(pp) || pp->M <= 0) return 0;
    long need = 0;
    if (MG_IS16(pp->dtype)) dispatch_wgrad<bf16raw>(*pp, nullptr, 0, &need, nullptr);
    else dispatch_wgrad<float>(*pp, nullptr, 0, &need, nullptr);
    return need;
}

// dW (fp32, p->stats) is fully OVERWRITTEN when a sufficient workspace is given (two-stage reduction over row splits);
// without workspace it is accumulated with atomics and must be pre-zeroed.
extern "C" int mg_conv_wgrad_ws(const mg_conv_params* pp, float* workspace, long workspace_floats, void* stream) {
    int rc = wgrad_check(pp);
    if (rc) return rc;
    if (!pp->stats) return -4;
    if (pp->M <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (pp->dtype == MG_BF16) return dispatch_wgrad<bf16raw>(*pp, workspace, workspace_floats, nullptr, st);
    if (pp->dtype == MG_F16) return dispatch_wgrad<f16raw>(*pp, workspace, workspace_floats, nullptr, st);
    return dispatch_wgrad<float>(*pp, workspace, workspace_floats, nullptr, st);
}

extern "C" int mg_conv_wgrad(const mg_conv_params* pp, void* stream) { return mg_conv_wgrad_ws(pp, nullptr, 0, stream); }

/* Same as mg_conv_wgrad_ws, but the slab reduction is PARKED: the GEMM writes its row-split slabs into `workspace` (which the caller keeps
 * untouched until the reduction has run) and `out` receives the descriptor of the reduction that was not launched (out->splits == 0: this
 * geometry needed none, dW is complete). mg_wgrad_reduce_batched runs any number of parked reductions as one launch (per 64 of them), with
 * the arithmetic of the per-layer kernels: parked or not, dW has the same bits. */
extern "C" int mg_conv_wgrad_park(const mg_conv_params* pp, float* workspace, long workspace_floats, mg_wgrad_parked* out, void* stream) {
    if (!out) return -1;
    out->ws = nullptr; out->dw = nullptr; out->n = 0; out->splits = 0; out->form = 0; out->dw_dtype = 0; out->blocks = 0;
    g_park = ParkState{true, nullptr, 0, 0, 0, 0};
    const int rc = mg_conv_wgrad_ws(pp, workspace, workspace_floats, stream);
    const ParkState got = g_park;
    g_park.on = false;
    if (rc) return rc;
    if (got.splits > 0) {
        out->ws = got.ws; out->dw = pp->stats; out->n = got.n; out->splits = got.splits; out->form = got.form; out->dw_dtype = pp->dw_dtype;
        out->blocks = got.blocks;
    }
    return 0;
}

extern "C" int mg_wgrad_reduce_batched(const mg_wgrad_parked* items, int count, void* stream) {
    if (count <= 0) return 0;
    if (!items) return -1;
    hipStream_t st = (hipStream_t)stream;
    RedTable tb;
    int k = 0;
    uint32_t blocks = 0;
    auto flush = [&]() -> int {
        if (k == 0) return 0;
        tb.count = k;
        hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blocks), dim3(256), 0, st, tb);
        k = 0; blocks = 0;
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    };
    for (int i = 0; i < count; ++i) {
        const mg_wgrad_parked& it = items[i];
        if (it.splits <= 0 || it.n <= 0) continue;
        if (!it.ws || !it.dw || it.blocks <= 0 || it.form < 0 || it.form > 2) return -2;
        if (k == MG_RED_MAX || (uint64_t)blocks + (uint64_t)it.blocks > 0x3fffffffull) { int rc = flush(); if (rc) return rc; }
        tb.e[k] = RedEntry{it.ws, it.dw, it.n, it.splits, (int16_t)it.form, (int16_t)it.dw_dtype, blocks, (uint32_t)it.blocks};
        blocks += (uint32_t)it.blocks;
        ++k;
    }
    return flush();
}

