// build-variants: MG_CONV_T=0,1,3
// Implicit-GEMM convolution family for gfx950 (CDNA4): one MFMA kernel template serves
//   * dense conv fprop            (MODE_CONV : NHWC activations, KRSC weights)
//   * dense dgrad / ConvTranspose (MODE_TCONV: "transposed" row provider, any stride)
//   * sparse gather conv          (MODE_GATHER: SubM / inverse / strided sparse conv via a neighbour table)
// replacing cuDNN conv calls (reference: maggie/network/encoder/resnet.py:15-18,61-66,169-172,
// maggie/network/decoder/resnet.py:20-25, maggie/network/module/aspp.py:14-30) and spconv's
// gather-GEMM-scatter (maggie/network/decoder/resnet_inst_matt_spconv.py:61-130).
//
//   Y[m, co] = epilogue( sum_{tap, ci} X[src(m, tap), ci] * W[co, tap, ci] )
//
// Tiling: 256 threads = 4 waves, block tile 128 (rows) x BN (out channels), K walked in 64-byte slabs
// (32 bf16 / 16 f32 per row) staged global -> VGPR -> LDS (double buffered, one barrier per slab; rows padded to
// 80 B so the ds_read_b128 fragment reads are bank-conflict free). bf16 uses v_mfma_f32_16x16x32_bf16, fp32 uses
// the exact v_mfma_f32_16x16x4_f32 (4 per slab, K permuted consistently between A and B).
// Epilogue goes through an fp32 LDS tile so global stores / residual loads are 16-byte vectors and the per-channel
// BatchNorm statistics (sum, sum of squares) are reduced per block before one atomicAdd per channel.
#include "common.h"
#include "conv_xcd.h"
#include "../../include/maggie_hip.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

// wave arrangement per (BM, BN) block tile; every wave tile is a multiple of 16 x 16
template <int BM, int BN> struct TileCfg { static constexpr int WAVES_M = (BN >= 64 ? 2 : 4), WAVES_N = (BN >= 64 ? 2 : 1); };

// LDS row pitch of a K stage of KS 64-byte slabs: +16 B so that the 16 rows of a ds_read_b128 fragment read fall on 16
// different 16-byte slots (pitch/16 is odd for KS = 1, 2, 4)
template <int KS> constexpr int rowb() { return 64 * KS + 16; }
template <int BM, int BN> constexpr int ep_passes() { return (BM * (BN + 4) * 4 > 40960) ? 2 : 1; }
template <int BM, int BN, int KS> constexpr int stage_bytes() { return (BM + BN) * rowb<KS>(); }
template <int BM, int BN> constexpr int ctile_bytes() { return (BM / ep_passes<BM, BN>()) * (BN + 4) * 4 + 8 * BN * 4; }
template <int BM, int BN, int KS> constexpr int lds_bytes() {
    return stage_bytes<BM, BN, KS>() > ctile_bytes<BM, BN>() ? stage_bytes<BM, BN, KS>() : ctile_bytes<BM, BN>();
}

struct RowCoord { int n, ho, wo; bool ok; };


#ifdef MG_HALO_TIMING
__device__ long long mg_dbg[32 * 24];
#define MG_STAMP(i) do { if ((threadIdx.x & 63) == 0 && (work & 63) == 0 && work / 64 < 32 && work >= 0 && (threadIdx.x >> 6) == 0) mg_dbg[(work / 64) * 24 + (i)] = clock64(); } while (0)
extern "C" int mg_debug_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mg_dbg), sizeof(mg_dbg)); }
#else
#define MG_STAMP(i)
#endif

// Phase decomposition of the stride-2 transposed walk (ConvTranspose k4 s2 p1, and the data gradient of a stride-2 3x3 / 1x1 conv): an
// output pixel (ho, wo) only meets the filter taps with ky = (ho + pad) mod 2, kx = (wo + pad) mod 2 -- a quarter of them. The rows of the
// GEMM are therefore ordered phase-major (4 sub-lattices (ho & 1, wo & 1), each [N][Hout/2][Wout/2], padded to whole row tiles) and every
// tile walks only its phase's taps: executed FLOPs == algorithmic FLOPs instead of 4x (k4) / 4x (k3) / 4x (k1) of them.
__host__ __device__ __forceinline__ bool tconv_phased(const mg_conv_params& p) {
    const int eps = MG_IS16(p.dtype) ? 32 : 16;
    return p.mode == MG_MODE_TCONV && p.stride == 2 && p.dil == 1 && !p.m_dev && p.Cin % eps == 0 && !(p.Hout & 1) && !(p.Wout & 1) &&
           p.R >= 1 && p.S >= 1;
}
__host__ __device__ __forceinline__ long row_tiles(const mg_conv_params& p, int M, int bm) {
    if (tconv_phased(p)) return 4l * ((M / 4 + bm - 1) / bm);
    return (M + bm - 1) / bm;
}
// phase-major GEMM row -> output pixel row. `pm` = row inside the phase.
struct PhaseMap {
    int H2, W2, py, px, Mp;
    __device__ __forceinline__ PhaseMap(const mg_conv_params& p, int phase) : H2(p.Hout >> 1), W2(p.Wout >> 1), py(phase >> 1), px(phase & 1), Mp(p.N * (p.Hout >> 1) * (p.Wout >> 1)) {}
    __device__ __forceinline__ void decode(int pm, int& n, int& ho, int& wo) const {
        n = pm / (H2 * W2);
        const int rem = pm - n * H2 * W2;
        const int i = rem / W2;
        ho = 2 * i + py; wo = 2 * (rem - i * W2) + px;
    }
};

// =====================================================================================================================
// Tile epilogue shared by the fprop kernels: accumulators -> fp32 LDS tile -> y = act?(acc) * scale + shift (+ res) -> act? (+ res2),
// rounded to T, 16-byte global stores; optional BatchNorm statistics (column sums / sums of squares of the ROUNDED values).
// `rowmap(rt)` maps tile row rt to the output row index (or -1: outside the tensor).
// Shape of the code (a workgroup usually owns a whole CU, one wave per SIMD, so every instruction is exposed): the residual-free /
// full-vector case is a separate instantiation without per-element branches; the activation is branch-free
// (act(x) = max(x, x * slope) covers none (slope 1), ReLU (0) and LeakyReLU); all LDS reads of a pass are issued before the first
// use; fp32 -> bf16 is the packed hardware conversion. (Before: ~1600 static instructions with ~170 branches, 8-12 k cycles per
// 128 x 64 tile -- as long as the whole K loop of a C128 3x3 layer, tools/halo_timeline.py.)
// =====================================================================================================================
template <typename T, int BM, int BN, int FM, int FN, bool RES, bool BNB, typename RowMap>
__device__ __forceinline__ void tile_epilogue_impl(const mg_conv_params& p, f32x4 (&acc)[FM][FN], int wm, int wn, int WM, int WN, int n0, int stat_slot,
                                                   char* smem, const RowMap& rowmap) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    constexpr int EP = ep_passes<BM, BN>();
    constexpr int PR = BM / EP;
    constexpr int LDC = BN + 4;
    constexpr int CPR = BN / CE;
    constexpr int RPP = 256 / CPR;
    constexpr int NIT = (PR + RPP - 1) / RPP;
    static_assert((CPR & (CPR - 1)) == 0 && CPR <= 32, "channel chunks per tile row must be a power of two");
    float* sC = (float*)smem;                         // [PR][LDC]
    float* sStat = (float*)(smem + PR * LDC * 4);     // [4 waves][2][BN]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int cc = t % CPR, rr = t / CPR;
    const int cbase = n0 + cc * CE;
    const bool col_ok = cbase < p.Cout;
    const bool full_vec = cbase + CE <= p.Cout;
    float sc[CE], sh[CE], s1[CE], s2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { sc[e] = 1.f; sh[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
    // BNB (round 3): this launch is the data gradient arriving at the OUTPUT of a training BatchNorm(+activation) layer. The epilogue
    // then writes g = dz * act'(z) and accumulates that layer's backward sums (sum g, sum g * xhat) -- the whole bn_bwd_reduce pass
    // (59 launches, three tensor reads each per step) rides on tiles that are in registers anyway.
    [[maybe_unused]] float bmu[CE], bis[CE], bsc[CE], bsh[CE];
    [[maybe_unused]] const bool bnb_lazy = BNB && p.bnb_scale != nullptr;       // the layer's activation output was never stored: mask from x * scale + shift
    if constexpr (BNB) {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int c = cbase + e;
            bmu[e] = c < p.Cout ? p.bnb_mean[c] : 0.f;
            bis[e] = c < p.Cout ? p.bnb_invstd[c] : 0.f;
            bsc[e] = (bnb_lazy && c < p.Cout) ? p.bnb_scale[c] : 0.f;
            bsh[e] = (bnb_lazy && c < p.Cout) ? p.bnb_shift[c] : 1.f;
        }
    }
    if (full_vec) {                                    // vector loads, in flight under the LDS tile write below
        if (p.scale) {
#pragma unroll
            for (int e = 0; e < CE; e += 4) *(float4*)&sc[e] = *(const float4*)(p.scale + cbase + e);
        }
        if (p.shift) {
#pragma unroll
            for (int e = 0; e < CE; e += 4) *(float4*)&sh[e] = *(const float4*)(p.shift + cbase + e);
        }
    } else {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const int c = cbase + e;
            if (p.scale && c < p.Cout) sc[e] = p.scale[c];
            if (p.shift && c < p.Cout) sh[e] = p.shift[c];
        }
    }
    // act(x) = max(x, x * slope): none -> 1, ReLU -> 0, LeakyReLU -> slope; applied before or after the affine part
    const float sl = p.act == MG_ACT_NONE ? 1.f : (p.act == MG_ACT_RELU ? 0.f : p.slope);
    const float sl_pre = p.pre_act ? sl : 1.f, sl_post = p.pre_act ? 1.f : sl;
    T* __restrict__ yb = (T*)p.y;
    const T* __restrict__ r1b = (const T*)p.res;
    const T* __restrict__ r2b = (const T*)p.res2;
    const bool stats = p.stats != nullptr;
    [[maybe_unused]] const T* __restrict__ bxb = (const T*)p.bnb_x;
    [[maybe_unused]] const T* __restrict__ byb = (const T*)p.bnb_y;
    [[maybe_unused]] const float bsl = p.bnb_act == MG_ACT_NONE ? 1.f : (p.bnb_act == MG_ACT_RELU ? 0.f : p.slope);
#pragma unroll
    for (int ep = 0; ep < EP; ++ep) {
        long mrow[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int r = rr + k * RPP;
            mrow[k] = (r < PR && col_ok) ? rowmap(ep * PR + r) : -1l;
        }
        [[maybe_unused]] uint4 qbx[NIT], qby[NIT];
        if constexpr (BNB) {                                   // the BatchNorm layer's input and output rows of this tile: requested before the
#pragma unroll                                                 // accumulators go through LDS, so the HBM round trip overlaps the tile dump
            for (int k = 0; k < NIT; ++k) {
                qbx[k] = make_uint4(0, 0, 0, 0); qby[k] = make_uint4(0, 0, 0, 0);
                if (mrow[k] >= 0 && full_vec) {
                    qbx[k] = *(const uint4*)(bxb + mrow[k] * p.bnb_ld + cbase);
                    if (byb) qby[k] = *(const uint4*)(byb + mrow[k] * p.bnb_ld + cbase);
                }
            }
        }
        if (ep > 0) __syncthreads();
        if ((wm * WM) / PR == ep) {
            const int rb0 = wm * WM - ep * PR;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sC[(rb0 + i * 16 + lg * 4 + e) * LDC + wn * WN + j * 16 + lr] = acc[i][j][e];
        }
        __syncthreads();
        float v[NIT][CE];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int r = rr + k * RPP;
#pragma unroll
            for (int e = 0; e < CE; e += 4) *(float4*)&v[k][e] = *(const float4*)&sC[(r < PR ? r : 0) * LDC + cc * CE + e];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const long m = mrow[k];
            if (m < 0) continue;
            float rv[CE], rv2[CE];
            if constexpr (RES) {
#pragma unroll
                for (int e = 0; e < CE; ++e) { rv[e] = 0.f; rv2[e] = 0.f; }
                if (r1b) {
                    long rrow = m;
                    if (p.res_mode == 2) {                    // residual lives at half resolution (nearest x2 upsample)
                        const int hw = p.Hout * p.Wout;
                        const int n = (int)(m / hw); const int rem = (int)(m - (long)n * hw); const int ho = rem / p.Wout; const int wo = rem - ho * p.Wout;
                        rrow = ((long)n * (p.Hout >> 1) + (ho >> 1)) * (p.Wout >> 1) + (wo >> 1);
                    }
                    if (full_vec) { const uint4 q = *(const uint4*)(r1b + rrow * p.ldr + cbase); TR::unpack(q, rv); }
                    else {
#pragma unroll
                        for (int e = 0; e < CE; ++e) if (cbase + e < p.Cout) rv[e] = TR::ld(r1b + rrow * p.ldr + cbase + e);
                    }
                }
                if (r2b) {
                    if (full_vec) { const uint4 q = *(const uint4*)(r2b + m * p.ldr2 + cbase); TR::unpack(q, rv2); }
                    else {
#pragma unroll
                        for (int e = 0; e < CE; ++e) if (cbase + e < p.Cout) rv2[e] = TR::ld(r2b + m * p.ldr2 + cbase + e);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                float x = v[k][e];
                x = fmaxf(x, x * sl_pre);
                x = x * sc[e] + sh[e];
                if constexpr (RES) x += rv[e];
                x = fmaxf(x, x * sl_post);
                if constexpr (RES) x += rv2[e];
                v[k][e] = x;
            }
            [[maybe_unused]] float bxv[CE];
            if constexpr (BNB) {
                float byv[CE];
                if (full_vec) { TR::unpack(qbx[k], bxv); TR::unpack(qby[k], byv); }
                else {
#pragma unroll
                    for (int e = 0; e < CE; ++e) {
                        const bool ok = cbase + e < p.Cout;
                        bxv[e] = ok ? TR::ld(bxb + m * p.bnb_ld + cbase + e) : 0.f;
                        byv[e] = (ok && byb) ? TR::ld(byb + m * p.bnb_ld + cbase + e) : 1.f;
                    }
                }
                if (byb) {
#pragma unroll
                    for (int e = 0; e < CE; ++e) v[k][e] = byv[e] > 0.f ? v[k][e] : v[k][e] * bsl;
                } else if (bnb_lazy) {
#pragma unroll
                    for (int e = 0; e < CE; ++e) v[k][e] = (bxv[e] * bsc[e] + bsh[e]) > 0.f ? v[k][e] : v[k][e] * bsl;
                }
            }
            const uint4 packed = TR::pack(v[k]);               // rounded once; the statistics are those of the rounded values
            if (stats) {
                TR::unpack(packed, v[k]);
                if constexpr (BNB) {
#pragma unroll
                    for (int e = 0; e < CE; ++e) { s1[e] += v[k][e]; s2[e] += v[k][e] * (bxv[e] - bmu[e]) * bis[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < CE; ++e) { s1[e] += v[k][e]; s2[e] += v[k][e] * v[k][e]; }
                }
            }
            T* dst = yb + m * p.ldy + p.yoff + cbase;
            if (full_vec) *(uint4*)dst = packed;
            else {
#pragma unroll
                for (int e = 0; e < CE; ++e) if (cbase + e < p.Cout) TR::st(dst + e, v[k][e]);
            }
        }
    }
    if (stats) {
        // lanes t, t+CPR, t+2*CPR, ... of a wave hold the same channel chunk: butterfly over them (CPR is a power of two <= 32),
        // one LDS row per wave, then one global atomic per channel per block into 1 of 32 replicas. (LDS float atomics here
        // serialised 64-way on the narrow tiles and doubled the kernel time of the 512x512 layers.)
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < CE; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        }
        if (lane < CPR) {
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                sStat[wave * 2 * BN + lane * CE + e] = s1[e];
                sStat[wave * 2 * BN + BN + lane * CE + e] = s2[e];
            }
        }
        __syncthreads();
        if (t < 2 * BN) {
            const int c = t < BN ? t : t - BN;
            if (n0 + c < p.Cout) {
                const float val = (sStat[t] + sStat[2 * BN + t]) + (sStat[4 * BN + t] + sStat[6 * BN + t]);
                if (p.stat_mode == 1) {                                       // one row, sums only
                    if (t < BN) atomicAdd(&p.stats[n0 + c], val);
                } else {
                    // row of the statistics buffer: tile index modulo the rows the caller allocated. 32 rows (default) only spread the same-address
                    // atomics; with at least as many rows as output tiles (deterministic mode) every word receives exactly ONE addition and the
                    // finalize kernel adds the rows in index order
                    float* st = p.stats + (size_t)((unsigned)stat_slot % (unsigned)(p.stat_rep > 0 ? p.stat_rep : MG_STAT_REPLICAS)) * 2 * p.Cout;
                    atomicAdd(&st[(t < BN ? 0 : p.Cout) + n0 + c], val);
                }
            }
        }
    }
}

// BNB is a compile-time property of the KERNEL: as a run-time branch the BatchNorm-backward path put its registers into every fprop kernel
// (occupancy 3 -> 2, 4 -> 3, 7 -> 4 across the family, -Rpass-analysis=kernel-resource-usage; the step lost 0.4 ms).
template <typename T, int BM, int BN, int FM, int FN, bool BNB = false, typename RowMap>
__device__ __forceinline__ void tile_epilogue(const mg_conv_params& p, f32x4 (&acc)[FM][FN], int wm, int wn, int WM, int WN, int n0, int stat_slot,
                                              char* smem, const RowMap& rowmap) {
    if constexpr (BNB) tile_epilogue_impl<T, BM, BN, FM, FN, true, true>(p, acc, wm, wn, WM, WN, n0, stat_slot, smem, rowmap);
    else if (p.res || p.res2) tile_epilogue_impl<T, BM, BN, FM, FN, true, false>(p, acc, wm, wn, WM, WN, n0, stat_slot, smem, rowmap);
    else tile_epilogue_impl<T, BM, BN, FM, FN, false, false>(p, acc, wm, wn, WM, WN, n0, stat_slot, smem, rowmap);
}

// K is walked in STAGES of KS slabs (KS*64 bytes per row). The loads of stage s+1 are issued right after the barrier that
// publishes stage s and stay in flight under its KS*FM*FN MFMAs; one LDS buffer, two barriers per stage. KS = 4 is used for
// K-heavy layers (few, fat memory round trips: these GEMMs are small, so exposed load latency -- not bandwidth or MFMA rate --
// is what bounds them), KS = 1 for the thin-K high-resolution layers.
// SPLIT (deep layers with few rows: M <= ~4096, K >= 1152): the K stages are divided over `splits` blocks per tile, each writing its
// raw fp32 partial tile to its own slab of `ws` ([splits][M][Cout]); splitk_finish_kernel sums the slabs and applies the epilogue.
// These layers could only fill the chip with 64x32 tiles (21 FLOP per byte staged through L2/LDS: ~100-220 TFLOP/s whatever the
// shape); with the K split, 128x128 tiles (64 FLOP/B) reach the same block count.
// XF (round 5, the sparse head's BatchNorm1d layers on the operand path): x is the RAW output of the producing convolution; the in-image /
// live-neighbour chunks are transformed -- act(x * xf_scale + xf_shift), rounded to T -- between the staging registers and LDS (missing
// neighbours and rows beyond the live count stay 0). Channel-aligned layers with Cin <= 64 only: a thread's chunk column is fixed, so its
// constants are one or two sets of 16 registers (slab parity).
template <typename T, int BM, int BN, int KS, int MODE, bool SPLIT = false, bool BNB = false, bool XF = false>
__device__ __forceinline__ void igemm_fprop_tile(const mg_conv_params& p, const int M, int work, float* __restrict__ ws, int splits, char* smem) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE, EPS = TR::EPS;
    constexpr int WAVES_M = TileCfg<BM, BN>::WAVES_M, WAVES_N = TileCfg<BM, BN>::WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, FM = WM / 16, FN = WN / 16;
    constexpr int A_ROWS = BM / 64;                 // A rows staged per thread (per slab)
    constexpr int B_ITERS = (BN * 4 + 255) / 256;   // B chunks per thread (per slab)
    constexpr int ROWB = rowb<KS>();

    char* sA = smem;                       // [BM][ROWB]
    char* sB = smem + BM * ROWB;           // [BN][ROWB]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int ntn = (p.Cout + BN - 1) / BN;
    int sp = 0;
    if constexpr (SPLIT) { sp = work % splits; work /= splits; }
    const int mt = work / ntn;                   // channel tiles of one row tile are consecutive: they share the A rows
    int m0 = mt * BM;
    const int n0 = (work - mt * ntn) * BN;
    const int taps = p.R * p.S;
    const int Ktot = taps * p.Cin;
    int nslab = (Ktot + EPS - 1) / EPS;
    // phase-major rows + phase tap walk (tconv_phased): this tile belongs to ONE of the four output sub-lattices
    const bool ph = !SPLIT && MODE == MG_MODE_TCONV && tconv_phased(p);
    const int tpp = ph ? (M / 4 + BM - 1) / BM : 1;                  // row tiles per phase
    const PhaseMap pmap(p, ph ? mt / tpp : 0);
    int ky0 = 0, kx0 = 0, kstep = 1;
    if (ph) {
        m0 = (mt - (mt / tpp) * tpp) * BM;                           // row offset inside the phase
        ky0 = (pmap.py + p.pad) & 1; kx0 = (pmap.px + p.pad) & 1; kstep = 2;
        const int nky = (p.R - ky0 + 1) >> 1, nkx = (p.S - kx0 + 1) >> 1;
        nslab = nky * nkx * (p.Cin / EPS);
    }
    const int Mrows = ph ? pmap.Mp : M;                              // rows of this tile's row space
    const int nstage_all = (nslab + KS - 1) / KS;
    int s_beg = 0, nstage = nstage_all;          // this block walks stages [s_beg, nstage)
    if constexpr (SPLIT) {
        const int per = (nstage_all + splits - 1) / splits;
        s_beg = sp * per;
        nstage = min(nstage_all, s_beg + per);
    }
    const char* __restrict__ xb = (const char*)p.x;
    const char* __restrict__ wb = (const char*)p.w;

    // ---- per-thread A rows (A_ROWS rows, fixed 16-byte chunk column inside every slab) ----
    const int a_c = t & 3;
    RowCoord rc[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        int m = m0 + (t >> 2) + i * 64;
        rc[i].ok = m < Mrows;
        if (ph) {
            pmap.decode(rc[i].ok ? m : 0, rc[i].n, rc[i].ho, rc[i].wo);
        } else if (MODE != MG_MODE_GATHER) {
            int hw = p.Hout * p.Wout;
            int mm = rc[i].ok ? m : 0;
            rc[i].n = mm / hw;
            int rem = mm - rc[i].n * hw;
            rc[i].ho = rem / p.Wout;
            rc[i].wo = rem - rc[i].ho * p.Wout;
        } else {
            rc[i].n = m; rc[i].ho = 0; rc[i].wo = 0;
        }
    }
    int a_k = s_beg * KS * EPS + a_c * CE; // running k index of this thread's chunk in slab 0 of the current stage
    int a_tap = a_k / p.Cin;
    int a_ci = a_k - a_tap * p.Cin;

    // register prefetch depth: PF stages of loads in flight per thread
    constexpr int PF = 1;           // measured (PMC): waves are parked ~40% / issuing ~40% per stage; deeper prefetch only costs registers
    uint4 ra[PF][KS][A_ROWS], rb[PF][KS][B_ITERS];
    [[maybe_unused]] float xsc[2][8], xsh[2][8];
    [[maybe_unused]] unsigned xlive[PF] = {}, xsub[PF] = {};         // per staged chunk: bit (j * A_ROWS + i) = live, bit j = which constant set
    [[maybe_unused]] const float xsl = xf_slope_of(p.xf_act, p.xf_slope);
    if constexpr (XF) {
        static_assert(sizeof(T) == 2 && KS * A_ROWS <= 32, "operand transform: 16-bit storage");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c0 = (q * EPS < p.Cin ? q * EPS : 0) + (t & 3) * CE;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xsc[q][e] = p.xf_scale[c0 + e]; xsh[q][e] = p.xf_shift[c0 + e]; }
        }
    }
    const int invS = (65536 + p.S - 1) / p.S;                       // tap / S == (tap * invS) >> 16 for tap < 256
    const int sshift = p.stride == 1 ? 0 : (p.stride == 2 ? 1 : (p.stride == 4 ? 2 : -1));

    auto load_stage = [&](int s, int u) {
        int tap = a_tap, ci = a_ci;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            // A operand
            int ky = (tap * invS) >> 16, kx = tap - ky * p.S;
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                long src = -1;
                if (rc[i].ok && tap < taps) {
                    if (MODE == MG_MODE_CONV) {
                        int hi = rc[i].ho * p.stride - p.pad + ky * p.dil;
                        int wi = rc[i].wo * p.stride - p.pad + kx * p.dil;
                        if (hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win) src = ((long)rc[i].n * p.Hin + hi) * p.Win + wi;
                    } else if (MODE == MG_MODE_TCONV) {
                        int th = rc[i].ho + p.pad - ky * p.dil;
                        int tw = rc[i].wo + p.pad - kx * p.dil;
                        if (th >= 0 && tw >= 0) {
                            int hi, wi;
                            if (sshift >= 0) { hi = th >> sshift; wi = tw >> sshift; }       // stride 1 / 2 / 4: no integer division
                            else { hi = th / p.stride; wi = tw / p.stride; }
                            if (hi * p.stride == th && wi * p.stride == tw && hi < p.Hin && wi < p.Win)
                                src = ((long)rc[i].n * p.Hin + hi) * p.Win + wi;
                        }
                    } else {
                        src = p.nbr[(long)rc[i].n * taps + tap];
                    }
                }
                if (src >= 0) ra[u][j][i] = *(const uint4*)(xb + (src * p.ldx + ci) * (long)sizeof(T));
                else ra[u][j][i] = make_uint4(0, 0, 0, 0);
            }
            // B operand (weights, K-contiguous)
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) {
                int idx = t + i * 256;
                int co = idx >> 2, c = idx & 3;
                int k0 = (s * KS + j) * EPS + c * CE;
                if (idx < BN * 4 && (n0 + co) < p.Cout && k0 < Ktot)
                    rb[u][j][i] = *(const uint4*)(wb + ((long)(n0 + co) * Ktot + k0) * (long)sizeof(T));
                else rb[u][j][i] = make_uint4(0, 0, 0, 0);
            }
            ci += EPS;
            while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
        }
        a_tap = tap; a_ci = ci;
    };
    // ---- aligned fast path (Cin % EPS == 0: every 64-byte K slab lies inside ONE filter tap) ----------------------
    // The tap walk (tap, ky, kx, channel offset) is then uniform over the block and lives in SGPRs; per (row, slab) the
    // vector work is two adds, two unsigned range checks, one address mad. The generic path above costs ~25 VALU + ~15
    // SALU per load and made the stage loop issue-bound (16 MFMAs against ~340 ALU instructions per stage).
    // second aligned case: Cin == one 16-byte chunk (the 8-channel network input): chunk c of slab q IS filter tap 4q + c
    const bool c8 = (CE == 8 && p.Cin == 8);
    const bool al = (p.Cin % EPS == 0 || c8) && (MODE != MG_MODE_TCONV || sshift >= 0);
    int hb[A_ROWS], wbs[A_ROWS], rbase[A_ROWS];
    const char* bptr[B_ITERS];
    bool bok[B_ITERS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        if (MODE == MG_MODE_CONV) { hb[i] = rc[i].ho * p.stride - p.pad; wbs[i] = rc[i].wo * p.stride - p.pad; }
        else { hb[i] = rc[i].ho + p.pad; wbs[i] = rc[i].wo + p.pad; }
        rbase[i] = (MODE == MG_MODE_GATHER) ? rc[i].n * taps : rc[i].n * p.Hin;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        int idx = t + i * 256;
        int co = idx >> 2, c = idx & 3;
        bok[i] = idx < BN * 4 && (n0 + co) < p.Cout;
        bptr[i] = wb + ((long)(n0 + (bok[i] ? co : 0)) * Ktot + c * CE) * (long)sizeof(T);
    }
    const int spt = p.Cin / EPS;             // slabs per tap (aligned path)
    int q_slab = s_beg * KS, q_sub = 0, q_ky = ky0, q_kx = kx0, q_tap = 0;
    if (SPLIT && al && !c8) { q_tap = q_slab / spt; q_sub = q_slab - q_tap * spt; q_ky = q_tap / p.S; q_kx = q_tap - q_ky * p.S; }
    auto load_stage_al = [&](int u) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            bool live = q_slab < nslab;
            int dh = q_ky * p.dil, dw = q_kx * p.dil, tap_g = q_tap;
            long coff = ((long)q_sub * EPS + a_c * CE) * (long)sizeof(T);
            if (c8) {
                tap_g = q_slab * 4 + a_c;
                live = tap_g < taps;
                const int ky = (tap_g * invS) >> 16, kx = tap_g - ky * p.S;
                dh = ky * p.dil; dw = kx * p.dil; coff = 0;
            }
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                long src = -1;
                if (MODE == MG_MODE_CONV) {
                    const int hi = hb[i] + dh, wi = wbs[i] + dw;
                    if ((unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win) src = (long)(rbase[i] + hi) * p.Win + wi;
                } else if (MODE == MG_MODE_TCONV) {
                    const int th = hb[i] - dh, tw = wbs[i] - dw;
                    const int hi = th >> sshift, wi = tw >> sshift;
                    if (th >= 0 && tw >= 0 && ((th | tw) & (p.stride - 1)) == 0 && hi < p.Hin && wi < p.Win)
                        src = (long)(rbase[i] + hi) * p.Win + wi;
                } else {
                    if (rc[i].ok && live) src = p.nbr[(long)rbase[i] + tap_g];
                }
                const bool have = src >= 0 && rc[i].ok && live;
                if (have) ra[u][j][i] = *(const uint4*)(xb + src * ((long)p.ldx * (long)sizeof(T)) + coff);
                else ra[u][j][i] = make_uint4(0, 0, 0, 0);
                if constexpr (XF) xlive[u] = have ? (xlive[u] | (1u << (j * A_ROWS + i))) : (xlive[u] & ~(1u << (j * A_ROWS + i)));
            }
            if constexpr (XF) xsub[u] = (q_sub & 1) ? (xsub[u] | (1u << j)) : (xsub[u] & ~(1u << j));
            const long boff = ph ? ((long)(q_ky * p.S + q_kx) * p.Cin + (long)q_sub * EPS) * (long)sizeof(T) : (long)q_slab * EPS * (long)sizeof(T);
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) {
                const bool blive = ph ? live : q_slab * EPS + ((t + i * 256) & 3) * CE < Ktot;
                if (bok[i] && blive) rb[u][j][i] = *(const uint4*)(bptr[i] + boff);
                else rb[u][j][i] = make_uint4(0, 0, 0, 0);
            }
            ++q_slab;
            if (++q_sub == spt) { q_sub = 0; ++q_tap; q_kx += kstep; if (q_kx >= p.S) { q_kx = kx0; q_ky += kstep; } }
        }
    };

    auto store_stage = [&](int u) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                if constexpr (XF) {
                    if (xlive[u] & (1u << (j * A_ROWS + i)))
                        ra[u][j][i] = (xsub[u] & (1u << j)) ? xf_apply8<T>(ra[u][j][i], xsc[1], xsh[1], xsl) : xf_apply8<T>(ra[u][j][i], xsc[0], xsh[0], xsl);
                }
                *(uint4*)(sA + ((t >> 2) + i * 64) * ROWB + j * 64 + a_c * 16) = ra[u][j][i];
            }
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) {
                int idx = t + i * 256;
                if (idx < BN * 4) *(uint4*)(sB + (idx >> 2) * ROWB + j * 64 + (idx & 3) * 16) = rb[u][j][i];
            }
        }
    };

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const char* a_base = sA + (wm * WM + lr) * ROWB + lg * 16;
    const char* b_base = sB + (wn * WN + lr) * ROWB + lg * 16;
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (s_beg + u < nstage) { if (al) load_stage_al(u); else load_stage(s_beg + u, u); }
    for (int s0 = s_beg; s0 < nstage; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {                 // fully unrolled: `u` is a compile-time register-bank index
            const int s = s0 + u;
            if (s < nstage) {
                store_stage(u);
                __syncthreads();
                if (s + PF < nstage) { if (al) load_stage_al(u); else load_stage(s + PF, u); }
        #pragma unroll
                for (int j = 0; j < KS; ++j) {
                    uint4 fa[FM], fb[FN];
        #pragma unroll
                    for (int i = 0; i < FM; ++i) fa[i] = *(const uint4*)(a_base + i * 16 * ROWB + j * 64);
        #pragma unroll
                    for (int i = 0; i < FN; ++i) fb[i] = *(const uint4*)(b_base + i * 16 * ROWB + j * 64);
        #pragma unroll
                    for (int i = 0; i < FM; ++i)
        #pragma unroll
                        for (int jj = 0; jj < FN; ++jj) {
                            if constexpr (sizeof(T) == 2) {
                                acc[i][jj] = mfma16<T>(fa[i], fb[jj], acc[i][jj]);
                            } else {
                                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[i].x), __uint_as_float(fb[jj].x), acc[i][jj], 0, 0, 0);
                                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[i].y), __uint_as_float(fb[jj].y), acc[i][jj], 0, 0, 0);
                                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[i].z), __uint_as_float(fb[jj].z), acc[i][jj], 0, 0, 0);
                                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[i].w), __uint_as_float(fb[jj].w), acc[i][jj], 0, 0, 0);
                            }
                        }
                }
                __syncthreads();
            }
        }
    }

    if constexpr (SPLIT) {
        // raw partial tile -> this split's slab (16 lanes = 16 consecutive channels = one 64-byte segment per row)
        float* slab = ws + (long)sp * M * p.Cout;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + wm * WM + i * 16 + lg * 4 + e, c = n0 + wn * WN + j * 16 + lr;
                    if (m < M && c < p.Cout) slab[(long)m * p.Cout + c] = acc[i][j][e];
                }
        return;
    }
    // ---------------- epilogue (tile_epilogue): accumulators -> fp32 LDS tile -> vectorised global stores ----------------
    auto rowmap = [&](int rt) -> long {
        int m = m0 + rt;
        if (m >= Mrows) return -1l;
        if (ph) { int n_, ho_, wo_; pmap.decode(m, n_, ho_, wo_); m = (n_ * p.Hout + ho_) * p.Wout + wo_; }
        return (long)m;
    };
    tile_epilogue<T, BM, BN, FM, FN, BNB>(p, acc, wm, wn, WM, WN, n0, mt, smem, rowmap);
}


// one tile per workgroup (dense layers: the row count is a host value)
template <typename T, int BM, int BN, int KS, int MODE, bool SPLIT = false, bool BNB = false>
__global__ __launch_bounds__(256) void igemm_fprop_kernel(const mg_conv_params p, float* __restrict__ ws, int splits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntn = (p.Cout + BN - 1) / BN;
    int work;
    if (!xcd_order((int)(SPLIT ? (long)((p.M + BM - 1) / BM) * ntn * splits : row_tiles(p, p.M, BM) * ntn), work)) return;
    igemm_fprop_tile<T, BM, BN, KS, MODE, SPLIT, BNB>(p, p.M, work, ws, splits, smem);
}

// Persistent form for the sparse head (p.m_dev): the row count lives in a device word, the grid is fixed (a multiple of 8 workgroups,
// a few per CU) and every workgroup walks tiles grid-stride in the same XCD-contiguous order. No host code depends on the count.
template <typename T, int BM, int BN, int KS, int MODE, bool XF = false>
__global__ __launch_bounds__(256) void igemm_fprop_persistent_kernel(const mg_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = dev_rows(p.m_dev, p.M);
    const int ntn = (p.Cout + BN - 1) / BN;
    const int L = ((M + BM - 1) / BM) * ntn;
    const int chunk = (L + NXCD - 1) / NXCD;
    for (int vb = blockIdx.x; vb / NXCD < chunk; vb += gridDim.x) {          // gridDim.x % 8 == 0: vb stays on this workgroup's XCD
        const int work = (vb % NXCD) * chunk + vb / NXCD;
        if (work < L) igemm_fprop_tile<T, BM, BN, KS, MODE, false, false, XF>(p, M, work, nullptr, 1, smem);
        __syncthreads();                                                      // the next tile's staging reuses the epilogue's LDS
    }
}

// =====================================================================================================================
// Direct-to-LDS ("async copy") form of the tile loop for the channel-aligned bf16 layers (Cin % 32 == 0: every 64-byte K slab
// lies inside one filter tap; Cout > 32). The register-staged loop above is latency-bound: one stage of loads in flight per
// block, two barriers per stage, global -> VGPR -> LDS (PMC: waves parked ~42 % of the time). Here `global_load_lds_dwordx4`
// writes the tile straight into LDS, so the staging registers disappear and a ring of NS stage buffers keeps NS - 1 stages of
// loads in flight per block behind ONE barrier per stage:
//     wait vmcnt(loads of the younger stages) -> s_barrier -> issue stage s + NS - 1 -> MFMAs of stage s
// LDS image: per slab [rows][64 B] unpadded (an LDS-DMA instruction writes wave-uniform base + lane * 16 B, so 64 lanes = 16
// rows x 4 chunks land contiguously); bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle applied on
// the SOURCE side (which lane fetches which 16-byte chunk is free, the global address is per lane): slot = chunk ^ (row & 8 ? 3 : 0)
// -- the 16 lanes of each ds_read_b128 service group then hit 16 different 16-byte slots of the 256-byte bank row.
// Out-of-image taps / rows beyond M read a 16-byte zero page instead (an LDS-DMA lane cannot be masked to zero).
// =====================================================================================================================
__device__ uint4 mg_zero_page[4];        // zero-initialised device memory

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

template <int BM, int BN, int KS, int NS> constexpr int async_stage_bytes() { return KS * (BM + BN) * 64; }
template <int BM, int BN, int KS, int NS> constexpr int async_lds_bytes() {
    return NS * async_stage_bytes<BM, BN, KS, NS>() > ctile_bytes<BM, BN>() ? NS * async_stage_bytes<BM, BN, KS, NS>() : ctile_bytes<BM, BN>();
}

template <int BM, int BN, int KS, int NS, int MODE, bool BNB = false, typename T = bf16raw>
__device__ __forceinline__ void igemm_fprop_async_tile(const mg_conv_params& p, const int M, int work, char* smem) {
    using TR = ElemTraits<T>;
    constexpr int CE = 8, EPS = 32;
    constexpr int WAVES_M = TileCfg<BM, BN>::WAVES_M, WAVES_N = TileCfg<BM, BN>::WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, FM = WM / 16, FN = WN / 16;
    constexpr int AG = BM / 64, BG = BN / 64;               // 16-row groups each wave fetches per slab (A, B)
    constexpr int SLAB_A = BM * 64, SLAB_B = BN * 64;       // bytes per slab
    constexpr int STAGE = KS * (SLAB_A + SLAB_B);
    constexpr int L = KS * (AG + BG);                       // LDS-DMA instructions per wave per stage
    static_assert(BN >= 64 && NS >= 2 && L * (NS - 1) <= 60, "tile / ring not supported");

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int mt = work / ntn;
    const int m0 = mt * BM, n0 = (work - mt * ntn) * BN;
    const int taps = p.R * p.S;
    const int Ktot = taps * p.Cin;
    const int nslab = Ktot / EPS;
    const int nstage = (nslab + KS - 1) / KS;
    const char* __restrict__ xb = (const char*)p.x;
    const char* __restrict__ wb = (const char*)p.w;
    const char* zpage = (const char*)mg_zero_page;
    const int sshift = p.stride == 1 ? 0 : (p.stride == 2 ? 1 : 2);

    // this lane's slot in every 16-row x 4-chunk DMA instruction, and the chunk it fetches into it (swizzle on the source side)
    const int lrow = lane >> 2;                              // row inside the 16-row group
    const int lch = (lane & 3) ^ (((lane >> 5) & 1) * 3);    // chunk of the 64-byte slab this lane fetches
    int hb[AG], wbs[AG], rbase[AG];
    bool rok[AG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
        const int m = m0 + (wave + 4 * i) * 16 + lrow;
        rok[i] = m < M;
        const int mm = rok[i] ? m : 0;
        if (MODE != MG_MODE_GATHER) {
            const int hw = p.Hout * p.Wout;
            const int n = mm / hw, rem = mm - n * hw, ho = rem / p.Wout, wo = rem - ho * p.Wout;
            if (MODE == MG_MODE_CONV) { hb[i] = ho * p.stride - p.pad; wbs[i] = wo * p.stride - p.pad; }
            else { hb[i] = ho + p.pad; wbs[i] = wo + p.pad; }
            rbase[i] = n * p.Hin;
        } else {
            hb[i] = 0; wbs[i] = 0; rbase[i] = mm * taps;
        }
    }
    const char* bptr[BG];
    bool bok[BG];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int co = n0 + (wave + 4 * i) * 16 + lrow;
        bok[i] = co < p.Cout;
        bptr[i] = wb + ((long)(bok[i] ? co : 0) * Ktot + lch * CE) * 2l;
    }
    const int spt = p.Cin / EPS;
    int q_slab = 0, q_sub = 0, q_ky = 0, q_kx = 0, q_tap = 0;
    const long xpitch = (long)p.ldx * 2l;

    auto issue_stage = [&](int buf) {
        char* sbase = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const bool live = q_slab < nslab;
            const int dh = q_ky * p.dil, dw = q_kx * p.dil;
            const long coff = ((long)q_sub * EPS + lch * CE) * 2l;
#pragma unroll
            for (int i = 0; i < AG; ++i) {
                long src = -1;
                if (MODE == MG_MODE_CONV) {
                    const int hi = hb[i] + dh, wi = wbs[i] + dw;
                    if ((unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win) src = (long)(rbase[i] + hi) * p.Win + wi;
                } else if (MODE == MG_MODE_TCONV) {
                    const int th = hb[i] - dh, tw = wbs[i] - dw;
                    const int hi = th >> sshift, wi = tw >> sshift;
                    if (th >= 0 && tw >= 0 && ((th | tw) & (p.stride - 1)) == 0 && hi < p.Hin && wi < p.Win)
                        src = (long)(rbase[i] + hi) * p.Win + wi;
                } else {
                    if (rok[i] && live) src = p.nbr[(long)rbase[i] + q_tap];
                }
                const char* g = (src >= 0 && rok[i] && live) ? xb + src * xpitch + coff : zpage;
                char* dst = sbase + j * SLAB_A + (wave + 4 * i) * 1024;             // wave-uniform; the hardware adds lane * 16
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
            }
            const long boff = (long)q_slab * EPS * 2l;
#pragma unroll
            for (int i = 0; i < BG; ++i) {
                const char* g = (bok[i] && live) ? bptr[i] + boff : zpage;
                char* dst = sbase + KS * SLAB_A + j * SLAB_B + (wave + 4 * i) * 1024;
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
            }
            ++q_slab;
            if (++q_sub == spt) { q_sub = 0; ++q_tap; if (++q_kx == p.S) { q_kx = 0; ++q_ky; } }
        }
    };

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;
    const int lgs = lg ^ (((lr >> 3) & 1) * 3);              // slot of k-chunk lg in this fragment row (row & 8 == lr & 8)
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_off = (wm * WM + lr) * 64 + lgs * 16;
    const int b_off = KS * SLAB_A + (wn * WN + lr) * 64 + lgs * 16;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the ring

#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nstage) issue_stage(u);
    for (int s = 0; s < nstage; ++s) {
        // this wave's loads of stage s have landed once at most the younger stages' loads are outstanding
        const int younger = min(NS - 2, nstage - 1 - s);
        if (NS == 2 || younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L) : "memory");
        __builtin_amdgcn_s_barrier();                        // everyone's stage s is in LDS; everyone is done reading stage s - 1
        asm volatile("" ::: "memory");
        if (s + NS - 1 < nstage) issue_stage((s + NS - 1) % NS);
        // Fragment reads as inline asm: hipcc's waitcnt pass treats every LDS-DMA in flight as a possible writer of whatever a
        // ds_read it can see reads, and puts `s_waitcnt vmcnt(0)` in front of it -- which would drain the ring every stage.
        // The ordering these reads need is exactly the counted wait + barrier above.
        const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE);
        u32x4 fa[KS][FM], fb[KS][FN];
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[j][i]) : "v"(sb + (unsigned)a_off), "n"(j * SLAB_A + i * 1024) : "memory");
#pragma unroll
            for (int i = 0; i < FN; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j][i]) : "v"(sb + (unsigned)b_off), "n"(j * SLAB_B + i * 1024) : "memory");
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            // slab j's FM + FN reads are complete once at most the later slabs' reads are outstanding
            if (j == KS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (j == KS - 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FM + FN) : "memory");
            else if (j == KS - 3) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((2 * (FM + FN)) > 15 ? 15 : 2 * (FM + FN)) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((3 * (FM + FN)) > 15 ? 15 : 3 * (FM + FN)) : "memory");
            __builtin_amdgcn_sched_barrier(0);               // the MFMAs below must not be hoisted above the wait
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jj = 0; jj < FN; ++jj)
                    acc[i][jj] = mfma16<T>(fa[j][i], fb[j][jj], acc[i][jj]);
        }
    }
    __syncthreads();                                         // all fragment reads done before the epilogue reuses the buffers

    // ---------------- epilogue (tile_epilogue, shared with igemm_fprop_tile) ----------------
    auto rowmap = [&](int rt) -> long { const int m = m0 + rt; return m < M ? (long)m : -1l; };
    tile_epilogue<T, BM, BN, FM, FN, BNB>(p, acc, wm, wn, WM, WN, n0, mt, smem, rowmap);
}

template <int BM, int BN, int KS, int NS, int MODE, bool BNB = false, typename T = bf16raw>
__global__ __launch_bounds__(256) void igemm_fprop_async_kernel(const mg_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntn = (p.Cout + BN - 1) / BN;
    int work;
    if (!xcd_order(((p.M + BM - 1) / BM) * ntn, work)) return;
    igemm_fprop_async_tile<BM, BN, KS, NS, MODE, BNB, T>(p, p.M, work, smem);
}

template <int BM, int BN, int KS, int NS, int MODE, typename T = bf16raw>
__global__ __launch_bounds__(256) void igemm_fprop_async_persistent_kernel(const mg_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = dev_rows(p.m_dev, p.M);
    const int ntn = (p.Cout + BN - 1) / BN;
    const int L = ((M + BM - 1) / BM) * ntn;
    const int chunk = (L + NXCD - 1) / NXCD;
    for (int vb = blockIdx.x; vb / NXCD < chunk; vb += gridDim.x) {
        const int work = (vb % NXCD) * chunk + vb / NXCD;
        if (work < L) igemm_fprop_async_tile<BM, BN, KS, NS, MODE, false, T>(p, M, work, smem);
        __syncthreads();
    }
}

// =====================================================================================================================
// Spatial halo-tile form for the 3x3 / stride 1 / pad 1 layers with Cin % 32 == 0 (bf16): the encoder's BasicBlocks and the
// decoder convs, forward and data gradient -- the bulk of the trunk's FLOPs (north_star: "LDS-staged 3x3 tiles").
// The im2col loops above re-fetch every input pixel once per filter tap: a 64 x 64 tile moves 32 FLOP per byte it pulls from
// L2, and the kernels sit at ~240 TFLOP/s = ~7.5 TB/s of L2 -> LDS traffic whatever the shape (measured; larger tiles leave CUs
// idle on these small layers). Here a workgroup owns TH x 16 output pixels x 64 output channels; per 32-channel slab it
// stages the (TH+2) x 18 input halo ONCE (direct-to-LDS) plus the nine 64 x 32 weight slabs, and forms the nine taps from LDS
// at constant offsets: 2.5-3x fewer bytes per FLOP. The data gradient of such a layer is the same convolution with the taps
// mirrored (MG_MODE_TCONV).
// LDS image of the halo: [TH+2][24 pixels][64 B] (pitch 24 px so that the bank swizzle slot = chunk ^ 2*((pixel>>2)&1) depends on
// x + kx only), weights per tap [64 rows][64 B] swizzled as in the im2col ring. NS = 3 stages in flight, one barrier per stage.
// =====================================================================================================================
template <int... Ks, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Ks...>, F&& f) { (f(std::integral_constant<int, Ks>{}), ...); }

#ifndef MG_HALO_ROW_SLIDE
#define MG_HALO_ROW_SLIDE 1              // 0: the per-tap walk (9 * (FM + FN) LDS reads per stage) -- kept for A/B builds of the halo kernels (-DMG_HALO_ROW_SLIDE=0)
#endif

template <int TH, int BN, int NS> struct HaloCfg {
    static constexpr int TW = 16, BM = TH * TW, PW = 24, HH = TH + 2;
    static constexpr int A_INSTR = (HH * PW + 15) / 16, A_PER_WAVE = (A_INSTR + 3) / 4, A_BYTES = A_PER_WAVE * 4 * 1024;
    // every wave issues B_PER_WAVE weight instructions per stage (the counted vmcnt waits rely on it): the B region is sized for ALL of them.
    // With BN = 32 the 18 real instructions round up to 20; sized as 9 * BN * 64 the two spare ones (zero-page loads) landed 2 KiB past the
    // stage -- in the NEXT ring buffer's halo rows while it was being read (Cin 96 with the two-slab ring: 20 % errors).
    static constexpr int B_GRP = BN / 16, B_INSTR = 9 * B_GRP, B_PER_WAVE = (B_INSTR + 3) / 4, B_BYTES = B_PER_WAVE * 4 * 1024;
    static constexpr int STAGE = A_BYTES + B_BYTES, L = A_PER_WAVE + B_PER_WAVE;
    static constexpr int LDS = NS * STAGE > ctile_bytes<BM, BN>() ? NS * STAGE : ctile_bytes<BM, BN>();
};


// XF (round 5, mg_conv_params.xf_*): `x` is the RAW output of the producing convolution and the BatchNorm + activation between the two layers is
// applied to the staged halo image IN LDS, once per pixel (not once per tap): after the counted wait that says "this lane's LDS-DMA loads of
// stage s have landed" every lane rewrites the 16-byte chunks its own loads deposited -- act(x * scale + shift), rounded to T -- and skips the
// chunks it pointed at the zero page (padding stays 0); the stage's barrier, which already separates the deposit from the fragment reads, then
// publishes the transformed image. scale | shift sit in an LDS table behind the ring ([2 * Cin] floats), filled from registers loaded before the
// first LDS-DMA instruction (in-order return: the first wait covers them) -- one extra barrier per tile.
template <int TH, int BN, int NS, int MODE, bool BNB = false, typename T = bf16raw, bool XF = false>
__device__ __forceinline__ void igemm_fprop_halo_tile(const mg_conv_params& p, int work, char* smem) {
    using TR = ElemTraits<T>;
    using HC = HaloCfg<TH, BN, NS>;
    constexpr int CE = 8, EPS = 32, TW = HC::TW, BM = HC::BM, PW = HC::PW, HH = HC::HH;
    constexpr int WAVES_N = BN >= 32 ? 2 : 1, WAVES_M = 4 / WAVES_N;   // BN = 16 (an 8-channel output, padded to one MFMA column tile): the four waves split the rows
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, FM = WM / 16, FN = WN / 16;
    constexpr int STAGE = HC::STAGE, L = HC::L, A_BYTES = HC::A_BYTES;
    static_assert((BN == 64 || BN == 32 || BN == 16) && (TH == 8 || TH == 4) && 2 * L <= 60, "halo tile configuration");

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int H = p.Hout, W = p.Wout;                        // stride 1, pad 1: input and output share the geometry
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int mt = work / ntn;
    const int n0 = (work - mt * ntn) * BN;
    const int img = mt / (tiles_y * tiles_x);
    const int trem = mt - img * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int Ktot = 9 * p.Cin;
    const int nstage = p.Cin / EPS;                          // one stage = one 32-channel slab, all nine taps
    const char* __restrict__ xb = (const char*)p.x;
    const char* __restrict__ wb = (const char*)p.w;
    const char* zpage = (const char*)mg_zero_page;
    const long xpitch = (long)p.ldx * 2l;

    // ---- what this lane fetches: A = halo pixels (A_PER_WAVE instructions per wave and stage), B = weight rows ----------------------
    const char* asrc[HC::A_PER_WAVE];
#pragma unroll
    for (int i = 0; i < HC::A_PER_WAVE; ++i) {
        const int a = wave + 4 * i;
        const int q = a * 16 + (lane >> 2);                  // pixel slot of the linear [HH][PW] halo image
        const int hy = q / PW, hx = q - hy * PW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = a < HC::A_INSTR && hy < HH && hx < TW + 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int ach = (lane & 3) ^ (((lane >> 4) & 1) * 2);  // chunk landing in this lane's slot: slot = chunk ^ 2*((q>>2)&1)
        asrc[i] = ok ? xb + ((long)(img * H + iy) * W + ix) * xpitch + ach * 16 : nullptr;
    }
    const char* bsrc[HC::B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < HC::B_PER_WAVE; ++i) {
        const int bi = wave + 4 * i;
        const int tap = bi / HC::B_GRP, grp = bi - tap * HC::B_GRP;
        const int co = n0 + grp * 16 + (lane >> 2);
        const int bch = (lane & 3) ^ (((lane >> 5) & 1) * 3);
        const bool ok = bi < HC::B_INSTR && co < p.Cout;
        bsrc[i] = ok ? wb + ((long)co * Ktot + (long)tap * p.Cin) * 2l + bch * 16 : nullptr;
    }
    auto issue_stage = [&](int s, int buf) {
        char* sbase = smem + buf * STAGE;
        const long coff = (long)s * EPS * 2l;
#pragma unroll
        for (int i = 0; i < HC::A_PER_WAVE; ++i) {
            const char* g = asrc[i] ? asrc[i] + coff : zpage;
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(sbase + (wave + 4 * i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < HC::B_PER_WAVE; ++i) {
            const char* g = bsrc[i] ? bsrc[i] + coff : zpage;
            const int bi = wave + 4 * i;
            // instruction bi = (tap, 16-row group): its 1 KiB lands at tap * BN * 64 + group * 1024 = bi * 1024 (B_GRP groups per tap)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(sbase + A_BYTES + bi * 1024), 16, 0, 0);
        }
    };

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // A fragment row i of this wave = tile row ty = wm * FM + i, pixel tx = lr; tap (aky, akx) reads halo pixel (ty + aky, lr + akx)
    unsigned a_lane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = lr + kx;
        a_lane[kx] = (unsigned)((wm * FM * PW + px) * 64 + ((lg ^ (((px >> 2) & 1) * 2)) << 4));
    }
    const unsigned b_lane = (unsigned)(A_BYTES + (wn * WN + lr) * 64 + ((lg ^ (((lr >> 3) & 1) * 3)) << 4));
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    [[maybe_unused]] u32x4 xf_reg = (u32x4){0u, 0u, 0u, 0u};
    [[maybe_unused]] const unsigned xf_tab = lds_base + (unsigned)HC::LDS;            // [Cin] scale | [Cin] shift, fp32
    [[maybe_unused]] const float xf_sl = xf_slope_of(p.xf_act, p.xf_slope);
    [[maybe_unused]] const int xf_ach = (lane & 3) ^ (((lane >> 4) & 1) * 2);         // the 8-channel group of a slab this lane's halo chunks hold
    // Single-stage forms (NS == 1: Cin 32 / 64, the large high-resolution layers): a lane's halo chunks always hold the same 8-channel group, and
    // there are at most two slabs -- its 2 x 16 constants sit in registers (loaded before the first LDS-DMA instruction; in-order return: the
    // stage's vmcnt(0) covers them), no LDS table, no extra barrier. (The table form cost the 512 x 512 C32 layer +9 us: nine-tap MFMA work of
    // a one-slab tile is as short as the table detour.)
    [[maybe_unused]] float xr_sc[2][8], xr_sh[2][8];
    if constexpr (XF && NS == 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c0 = (q < nstage ? q : 0) * EPS + xf_ach * 8;
            *(float4*)&xr_sc[q][0] = *(const float4*)(p.xf_scale + c0); *(float4*)&xr_sc[q][4] = *(const float4*)(p.xf_scale + c0 + 4);
            *(float4*)&xr_sh[q][0] = *(const float4*)(p.xf_shift + c0); *(float4*)&xr_sh[q][4] = *(const float4*)(p.xf_shift + c0 + 4);
        }
    } else if constexpr (XF) {
        if (t * 4 < 2 * p.Cin) xf_reg = *(const u32x4*)(t * 4 < p.Cin ? p.xf_scale + t * 4 : p.xf_shift + (t * 4 - p.Cin));
    }
    MG_STAMP(0);
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nstage) issue_stage(u, u);
    MG_STAMP(1);
    [[maybe_unused]] auto xf_stage = [&](int s) {
        if constexpr (NS == 1) {
            const unsigned sb = lds_base;
            u32x4 q[HC::A_PER_WAVE];
#pragma unroll
            for (int i = 0; i < HC::A_PER_WAVE; ++i)
                asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < HC::A_PER_WAVE; ++i) {
                if (asrc[i]) {
                    const uint4 r = s == 0 ? xf_apply8<T>(__builtin_bit_cast(uint4, q[i]), xr_sc[0], xr_sh[0], xf_sl)
                                           : xf_apply8<T>(__builtin_bit_cast(uint4, q[i]), xr_sc[1], xr_sh[1], xf_sl);
                    asm volatile("ds_write_b128 %0, %1" ::"v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)), "v"(__builtin_bit_cast(u32x4, r)) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return;
        }
        if (s == 0) {                                        // the table: every wave's share must be in LDS before anyone transforms
            if (t * 4 < 2 * p.Cin) asm volatile("ds_write_b128 %0, %1" ::"v"(xf_tab + (unsigned)t * 16u), "v"(xf_reg) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE);
        const unsigned tsc = xf_tab + (unsigned)((s * EPS + xf_ach * 8) * 4), tsh = tsc + (unsigned)(p.Cin * 4);
        f32x4 c0, c1, h0, h1;
        asm volatile("ds_read_b128 %0, %1" : "=v"(c0) : "v"(tsc) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(c1) : "v"(tsc) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(h0) : "v"(tsh) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(h1) : "v"(tsh) : "memory");
        u32x4 q[HC::A_PER_WAVE];
#pragma unroll
        for (int i = 0; i < HC::A_PER_WAVE; ++i)
            asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float sc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]}, sh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
        for (int i = 0; i < HC::A_PER_WAVE; ++i) {
            if (asrc[i]) {                                   // an in-image pixel: transformed; zero-page chunks (padding, spare slots) stay 0
                const uint4 r = xf_apply8<T>(__builtin_bit_cast(uint4, q[i]), sc, sh, xf_sl);
                asm volatile("ds_write_b128 %0, %1" ::"v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)), "v"(__builtin_bit_cast(u32x4, r)) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int s = 0; s < nstage; ++s) {
        if (s < 6) MG_STAMP(2 + 2 * s);
        if constexpr (NS == 1) {
            // one or two 32-channel slabs (the C32 / C64 high-resolution layers): no ring -- a single 48 KiB (BN = 64) stage lets three
            // workgroups share a CU and overlap each other's load / MFMA / epilogue phases instead
            if (s > 0) __builtin_amdgcn_s_barrier();
            issue_stage(s, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (XF) xf_stage(s);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            const int younger = min(NS - 2, nstage - 1 - s);
            if (NS == 2 || younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
            if constexpr (XF) xf_stage(s);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + NS - 1 < nstage) issue_stage(s + NS - 1, (s + NS - 1) % NS);
        }
        if (s < 6) MG_STAMP(3 + 2 * s);
        const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE);
        // Row-sliding tap walk (round 5). The tap loop below it (kept for A/B builds) reads FM A fragments + FN B fragments per tap: 9 * (FM + FN)
        // ds_read_b128 per stage and wave -- with 32-wide tiles (FN = 1) 45 KiB for 36 MFMAs, and the LDS read port (128 B / clk / CU, shared by
        // the two co-resident workgroups) is then busy ~2.5x as long as the matrix pipe: the K loop is LDS-read-bound. But output row i under tap
        // row ky reads the SAME halo row r = i + ky as output row i + 1 under ky - 1: a wave's FM output rows touch FM + 2 halo rows x 3 column
        // shifts = 3 * (FM + 2) distinct A fragments, not 9 * FM. So: all nine weight fragments of the stage are loaded once into registers
        // (9 * FN reads), the halo rows are walked top to bottom, each row's three fragments are read ONCE and feed every (output row, tap row)
        // pair that meets them -- 9 * FN + 3 * (FM + 2) reads per stage (27 against 45 at FM = 4, FN = 1; 36 against 54 at FN = 2). The next halo
        // row is in flight under the current row's MFMAs (counted lgkmcnt, as before).
#if MG_HALO_ROW_SLIDE
        // The walk is column-major: for column shift c = 0, 1, 2 the halo rows r = 0 .. FM + 1 stream through a three-deep fragment ring (two rows
        // in flight under the current row's MFMAs) and meet the three weight fragments B(ky, c) of that column; B(0, c + 1) and B(1, c + 1) are
        // loaded over their dead predecessors during the last two rows of column c, B(2, c) during row 0 -- one continuous stream of
        // 3 * (FM + 2) + 9 * FN reads with compile-time lgkmcnt counts, 12 * FN + 12 fragment registers.
        u32x4 rfb[3][FN], rfa[3];
        constexpr int NR = FM + 2, NSTEP = 3 * NR;
        const unsigned baddr = sb + b_lane;
        auto tap_of = [](int ky, int c) { return MODE == MG_MODE_TCONV ? (2 - ky) * 3 + (2 - c) : ky * 3 + c; };
        auto read_b = [&](auto ky_c, auto c_c) {
            constexpr int KY = decltype(ky_c)::value, C_ = decltype(c_c)::value;
            constexpr int TAP = MODE == MG_MODE_TCONV ? (2 - KY) * 3 + (2 - C_) : KY * 3 + C_;
            u32x4(&rb)[FN] = rfb[KY];
            const unsigned ba = baddr;
            if constexpr (FN >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[0]) : "v"(ba), "n"(TAP * BN * 64 + 0 * 1024) : "memory");
            if constexpr (FN >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[1]) : "v"(ba), "n"(TAP * BN * 64 + 1 * 1024) : "memory");
        };
        auto read_a = [&](auto k_c) {                             // halo row r under column shift c, stream position k = c * NR + r
            constexpr int K_ = decltype(k_c)::value, C_ = K_ / NR, R_ = K_ % NR;
            u32x4& ra = rfa[K_ % 3];
            const unsigned aa = sb + a_lane[C_];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra) : "v"(aa), "n"(R_ * PW * 64) : "memory");
        };
        // weight fragments (FN reads each) issued at stream position k: B(0, c + 1) at r == FM, B(1, c + 1) at r == FM + 1, B(2, c) at r == 0 (c > 0)
        auto nb_at = [](int k) { const int c = k / NR, r = k % NR; return k < 0 ? 0 : ((r == FM && c < 2) ? 1 : 0) + ((r == FM + 1 && c < 2) ? 1 : 0) + ((r == 0 && c > 0) ? 1 : 0); };
        (void)tap_of;
        read_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        read_b(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        read_b(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
        read_a(std::integral_constant<int, 0>{});
        read_a(std::integral_constant<int, 1>{});
        auto step = [&](auto k_c) {
            constexpr int K_ = decltype(k_c)::value, C_ = K_ / NR, R_ = K_ % NR;
            if constexpr (R_ == FM && C_ < 2) read_b(std::integral_constant<int, 0>{}, std::integral_constant<int, C_ + 1>{});
            if constexpr (R_ == FM + 1 && C_ < 2) read_b(std::integral_constant<int, 1>{}, std::integral_constant<int, C_ + 1>{});
            if constexpr (R_ == 0 && C_ > 0) read_b(std::integral_constant<int, 2>{}, std::integral_constant<int, C_>{});
            if constexpr (K_ + 2 < NSTEP) read_a(std::integral_constant<int, K_ + 2>{});
            // everything up to A(K_) has landed once at most the reads issued after it are outstanding: the weight loads of positions K_ - 1 and
            // K_ and the rows K_ + 1, K_ + 2 (in-order return)
            constexpr int after = (nb_at(K_ - 1) + nb_at(K_)) * FN + (K_ + 1 < NSTEP ? 1 : 0) + (K_ + 2 < NSTEP ? 1 : 0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(after) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            u32x4& ra = rfa[K_ % 3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int i = R_ - ky;                           // the output row that meets halo row R_ under tap row ky
                if (i >= 0 && i < FM) {
#pragma unroll
                    for (int jj = 0; jj < FN; ++jj) acc[i][jj] = mfma16<T>(ra, rfb[ky][jj], acc[i][jj]);
                }
            }
        };
        static_for(std::make_integer_sequence<int, NSTEP>{}, step);
#else
        u32x4 fa[2][FM], fb[2][FN];
        // taps are software-pipelined: the reads of tap t + 1 are issued before the MFMAs of tap t (fragment row i of the wave sits
        // i halo rows further down: a literal offset)
        auto read_tap = [&](auto tap_c, auto buf_c) {
            constexpr int TAP = decltype(tap_c)::value, BUF = decltype(buf_c)::value;
            constexpr int ky_ = TAP / 3, kx_ = TAP % 3;
            constexpr int aky_ = MODE == MG_MODE_TCONV ? 2 - ky_ : ky_, akx_ = MODE == MG_MODE_TCONV ? 2 - kx_ : kx_;
            const unsigned aaddr = sb + a_lane[akx_];
            const unsigned baddr = sb + b_lane;
            if constexpr (FM >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[BUF][0]) : "v"(aaddr), "n"((aky_ + 0) * PW * 64) : "memory");
            if constexpr (FM >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[BUF][1]) : "v"(aaddr), "n"((aky_ + 1) * PW * 64) : "memory");
            if constexpr (FM >= 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[BUF][2]) : "v"(aaddr), "n"((aky_ + 2) * PW * 64) : "memory");
            if constexpr (FM >= 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[BUF][3]) : "v"(aaddr), "n"((aky_ + 3) * PW * 64) : "memory");
            if constexpr (FN >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[BUF][0]) : "v"(baddr), "n"(TAP * BN * 64 + 0 * 1024) : "memory");
            if constexpr (FN >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[BUF][1]) : "v"(baddr), "n"(TAP * BN * 64 + 1 * 1024) : "memory");
        };
        auto mma_tap = [&](auto buf_c) {
            constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jj = 0; jj < FN; ++jj)
                    acc[i][jj] = mfma16<T>(fa[BUF][i], fb[BUF][jj], acc[i][jj]);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
#define MG_TAP_STEP(T_, CUR, NXT)                                                                       \
        read_tap(std::integral_constant<int, (T_) + 1>{}, NXT{});                                         \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FM + FN) : "memory");                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        mma_tap(CUR{});
        read_tap(I0{}, I0{});
        MG_TAP_STEP(0, I0, I1) MG_TAP_STEP(1, I1, I0) MG_TAP_STEP(2, I0, I1) MG_TAP_STEP(3, I1, I0)
        MG_TAP_STEP(4, I0, I1) MG_TAP_STEP(5, I1, I0) MG_TAP_STEP(6, I0, I1) MG_TAP_STEP(7, I1, I0)
#undef MG_TAP_STEP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_tap(I0{});
#endif
    }
    MG_STAMP(14);
    __syncthreads();

    // ---------------- epilogue (tile_epilogue): row r of the tile = pixel (y0 + r / 16, x0 + r % 16) ----------------
    auto rowmap = [&](int rt) -> long {
        const int y = y0 + rt / TW, x = x0 + (rt % TW);
        return (y < H && x < W) ? ((long)img * H + y) * W + x : -1l;
    };
    tile_epilogue<T, BM, BN, FM, FN, BNB>(p, acc, wm, wn, WM, WN, n0, mt, smem, rowmap);
    MG_STAMP(15);
}

template <int TH, int BN, int NS, int MODE, bool BNB = false, typename T = bf16raw, bool XF = false>
__global__ __launch_bounds__(256) void igemm_fprop_halo_kernel(const mg_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles = p.N * ((p.Hout + TH - 1) / TH) * ((p.Wout + 15) / 16) * ((p.Cout + BN - 1) / BN);
    int work;
    if (!xcd_order(tiles, work)) return;
    igemm_fprop_halo_tile<TH, BN, NS, MODE, BNB, T, XF>(p, work, smem);
}

// Deterministic mode: the statistics buffer must have one row per output tile of the kernel form that runs (mg_conv_params.stat_rep) -- with
// fewer rows two tiles would add to one word and their order would show. A caller that sized the buffer too small gets an error, not noise.
#define MG_STAT_ROWS_SHORT (-8)
static inline int stat_rows_check(const mg_conv_params& p, long mtiles) {
    return (mg_det_on && p.stats && p.stat_mode == 0 && (long)(p.stat_rep > 0 ? p.stat_rep : MG_STAT_REPLICAS) < mtiles) ? MG_STAT_ROWS_SHORT : 0;
}

static inline bool halo_eligible(const mg_conv_params& p) {
    static const int enabled = [] { const char* e = getenv("MG_FPROP_HALO"); return e ? atoi(e) : 1; }();
    if (!enabled || !MG_IS16(p.dtype) || p.m_dev || p.mode == MG_MODE_GATHER) return false;
    // Cin >= 96: three-stage ring over the 32-channel slabs; Cin 32 / 64: single-stage form (halo_small)
    static const int small = [] { const char* e = getenv("MG_FPROP_HALO_SMALL"); return e ? atoi(e) : 1; }();
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.dil != 1 || p.pad != 1 || p.Cin % 32 != 0) return false;
    if (p.Cin < 96 ? (!small || p.Cout < 8 || p.Cout % 8 != 0) : p.Cout < 64) return false;
    if (p.Hin != p.Hout || p.Win != p.Wout || p.Wout < 16 || p.Hout < 4) return false;
    return true;
}

template <typename T, int TH, int BN = 64, int NS = 3>
static int launch_fprop_halo(const mg_conv_params& p, hipStream_t st) {
    constexpr size_t lds = HaloCfg<TH, BN, NS>::LDS;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_CONV, false, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_TCONV, false, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_TCONV, true, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long tiles = (long)p.N * ((p.Hout + TH - 1) / TH) * ((p.Wout + 15) / 16) * ((p.Cout + BN - 1) / BN);
    if (int rcs = stat_rows_check(p, (long)p.N * ((p.Hout + TH - 1) / TH) * ((p.Wout + 15) / 16))) return rcs;
    dim3 grid(xcd_grid(tiles));
    if (p.xf_scale) {                                        // BatchNorm + activation of the producing layer applied to the staged halo (forward only)
        if constexpr (NS <= 2 && BN >= 32) {
            constexpr size_t lds_xf = lds + 4096;            // + the [2 * Cin] fp32 table (Cin <= 512)
            if (p.mode != MG_MODE_CONV || p.bnb_x || p.Cin > 512 || lds_xf > 160 * 1024) return MG_XF_UNSUPPORTED;
            static bool xf_attr = false;
            if (!xf_attr) {
                hipFuncSetAttribute((const void*)igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_CONV, false, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xf);
                xf_attr = true;
            }
            hipLaunchKernelGGL((igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_CONV, false, T, true>), grid, dim3(256), lds_xf, st, p);
            MG_CHECK_LAUNCH();
            return 0;
        } else return MG_XF_UNSUPPORTED;
    }
    if (p.bnb_x) {                                           // the data gradient of a 3x3 / stride 1 conv behind a BatchNorm layer
        if (p.mode != MG_MODE_TCONV) return -2;
        hipLaunchKernelGGL((igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_TCONV, true, T>), grid, dim3(256), lds, st, p);
    } else if (p.mode == MG_MODE_CONV) hipLaunchKernelGGL((igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_CONV, false, T>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((igemm_fprop_halo_kernel<TH, BN, NS, MG_MODE_TCONV, false, T>), grid, dim3(256), lds, st, p);
    MG_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int dispatch_fprop_halo(const mg_conv_params& p, hipStream_t st) {
    // 8 x 16 pixel tiles when they still give about one workgroup per CU (a workgroup holds a whole CU's LDS), else 4 x 16
    static const long want = [] { const char* e = getenv("MG_HALO_BLOCKS"); return e ? atol(e) : 200l; }();
    if (p.Cin < 96) {
        if (p.Cout <= 16) return p.Hout >= 8 ? launch_fprop_halo<T, 8, 16, 1>(p, st) : launch_fprop_halo<T, 4, 16, 1>(p, st);   // the 8-channel network input's data gradient
        if (p.Cout <= 32) return p.Hout >= 8 ? launch_fprop_halo<T, 8, 32, 1>(p, st) : launch_fprop_halo<T, 4, 32, 1>(p, st);
        return p.Hout >= 8 ? launch_fprop_halo<T, 8, 64, 1>(p, st) : launch_fprop_halo<T, 4, 64, 1>(p, st);
    }
    const long t8 = (long)p.N * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16) * ((p.Cout + 63) / 64);
    // 32-channel-wide tiles with a two-slab ring: 66 KiB of LDS -> TWO workgroups per CU, whose load / MFMA / epilogue phases overlap (the
    // 64-wide three-slab form owns the whole CU). Costs 30 % more halo traffic; measured C512->256 32x32 18.0 -> 15.5 us, C256->128 64x64
    // 15.2 -> 14.1 us, C128 / C256 unchanged, step 14.91 -> 14.74 ms. MG_HALO_NARROW=0 selects the wide form.
    static const int narrow = [] { const char* e = getenv("MG_HALO_NARROW"); return e ? atoi(e) : 1; }();
    if (p.xf_scale && p.Hout >= 8) return launch_fprop_halo<T, 8, 32, 2>(p, st);      // the operand transform lives in the one- / two-slab forms
    if (narrow && p.Hout >= 8) return launch_fprop_halo<T, 8, 32, 2>(p, st);      // (4 x 16 tiles for the layers with < 300 workgroups: no gain, measured)
    if (t8 >= want && p.Hout >= 8) return launch_fprop_halo<T, 8>(p, st);
    return launch_fprop_halo<T, 4>(p, st);
}

static inline bool async_eligible(const mg_conv_params& p) {
    // 1 (default): the 1x1 layers and the sparse gather convs (measured: 1x1 C128->64 15.6 -> 10.5 us; neutral or slightly slower than the
    // register-staged loop on the dense 3x3 shapes that the halo kernel does not take); 2: every aligned layer; 0: off
    static const int enabled = [] { const char* e = getenv("MG_FPROP_ASYNC"); return e ? atoi(e) : 1; }();
    if (!enabled || !MG_IS16(p.dtype) || p.Cin % 32 != 0 || p.Cout <= 32) return false;
    if (enabled == 1 && !(p.R * p.S == 1 || p.mode == MG_MODE_GATHER)) return false;
    if (p.mode == MG_MODE_TCONV && !(p.stride == 1 || p.stride == 2 || p.stride == 4)) return false;
    if ((long)p.R * p.S * p.Cin * 2l >= (1l << 31)) return false;
    return true;
}

template <typename T, int BM, int BN, int KS, int NS>
int launch_fprop_async(const mg_conv_params& p, hipStream_t st) {
    constexpr size_t lds = async_lds_bytes<BM, BN, KS, NS>();
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_CONV, false, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_TCONV, false, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_GATHER, false, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_async_persistent_kernel<BM, BN, KS, NS, MG_MODE_CONV, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_async_persistent_kernel<BM, BN, KS, NS, MG_MODE_GATHER, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_TCONV, true, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
    if (int rcs = stat_rows_check(p, (p.M + BM - 1) / BM)) return rcs;
    if (p.bnb_x) {                                           // data gradient of a 1x1 conv behind a BatchNorm layer
        if (p.m_dev || p.mode != MG_MODE_TCONV) return -2;
        hipLaunchKernelGGL((igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_TCONV, true, T>), dim3(xcd_grid(tiles)), dim3(256), lds, st, p);
        MG_CHECK_LAUNCH();
        return 0;
    }
    if (p.m_dev) {
        const long g = tiles < 2048 ? tiles : 2048;
        dim3 pg(xcd_grid(g < 1 ? 1 : g));
        if (p.mode == MG_MODE_CONV) hipLaunchKernelGGL((igemm_fprop_async_persistent_kernel<BM, BN, KS, NS, MG_MODE_CONV, T>), pg, dim3(256), lds, st, p);
        else if (p.mode == MG_MODE_GATHER) hipLaunchKernelGGL((igemm_fprop_async_persistent_kernel<BM, BN, KS, NS, MG_MODE_GATHER, T>), pg, dim3(256), lds, st, p);
        else return -2;
        MG_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid(xcd_grid(tiles));
    switch (p.mode) {
        case MG_MODE_CONV: hipLaunchKernelGGL((igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_CONV, false, T>), grid, dim3(256), lds, st, p); break;
        case MG_MODE_TCONV: hipLaunchKernelGGL((igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_TCONV, false, T>), grid, dim3(256), lds, st, p); break;
        case MG_MODE_GATHER: hipLaunchKernelGGL((igemm_fprop_async_kernel<BM, BN, KS, NS, MG_MODE_GATHER, false, T>), grid, dim3(256), lds, st, p); break;
        default: return -2;
    }
    MG_CHECK_LAUNCH();
    return 0;
}

// tile choice as in dispatch_fprop_ks (largest tile that still fills the chip); ring depth / stage width by K and tile
template <typename T>
static int dispatch_fprop_async(const mg_conv_params& p, hipStream_t st) {
    static const long want = [] { const char* e = getenv("MG_FPROP_BLOCKS"); return e ? atol(e) : 768l; }();
    auto blocks = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); };
    static const int ns_small = [] { const char* e = getenv("MG_ASYNC_NS"); return e ? atoi(e) : 4; }();
    if (p.Cout > 64) {
        if (blocks(128, 128) >= want) return launch_fprop_async<T, 128, 128, 2, 3>(p, st);
        if (blocks(128, 64) >= want) return ns_small >= 4 ? launch_fprop_async<T, 128, 64, 2, 4>(p, st) : launch_fprop_async<T, 128, 64, 2, 3>(p, st);
        return ns_small >= 4 ? launch_fprop_async<T, 64, 64, 2, 4>(p, st) : launch_fprop_async<T, 64, 64, 2, 3>(p, st);
    }
    if (blocks(128, 64) >= want) return ns_small >= 4 ? launch_fprop_async<T, 128, 64, 2, 4>(p, st) : launch_fprop_async<T, 128, 64, 2, 3>(p, st);
    return ns_small >= 4 ? launch_fprop_async<T, 64, 64, 2, 4>(p, st) : launch_fprop_async<T, 64, 64, 2, 3>(p, st);
}

// operand transform in the register-staged persistent form (the sparse head's row matrices: device row count, gather 3x3 or 1x1, Cin 32 / 64,
// Cout <= 32 -- wider outputs run the direct-to-LDS ring, which cannot transform in flight)
static inline bool fprop_xf_rows_ok(const mg_conv_params& p) {
    return MG_IS16(p.dtype) && p.m_dev && (p.mode == MG_MODE_GATHER || (p.mode == MG_MODE_CONV && p.R * p.S == 1 && p.stride == 1 && p.pad == 0)) &&
           (p.Cin == 32 || p.Cin == 64) && p.Cout <= 32 && p.ldx % 8 == 0;
}

template <typename T, int BM, int BN, int KS>
int launch_fprop(const mg_conv_params& p, hipStream_t st) {
    dim3 grid(xcd_grid(row_tiles(p, p.M, BM) * ((p.Cout + BN - 1) / BN)));
    if (int rcs = stat_rows_check(p, row_tiles(p, p.M, BM))) return rcs;
    constexpr size_t lds = lds_bytes<BM, BN, KS>();
    static bool attr_set = false;
    if (lds > 65536 && !attr_set) {
        hipFuncSetAttribute((const void*)igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_TCONV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_CONV, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_TCONV, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.bnb_x) {                                           // a data-gradient launch that also accumulates a BatchNorm layer's backward sums
        if (p.m_dev) return -2;
        if (p.mode == MG_MODE_CONV) hipLaunchKernelGGL((igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_CONV, false, true>), grid, dim3(256), lds, st, p, (float*)nullptr, 1);
        else if (p.mode == MG_MODE_TCONV) hipLaunchKernelGGL((igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_TCONV, false, true>), grid, dim3(256), lds, st, p, (float*)nullptr, 1);
        else return -2;
        MG_CHECK_LAUNCH();
        return 0;
    }
    if (p.m_dev) {
        // fixed persistent grid: enough workgroups to fill the chip (8 per CU at most), never more than the capacity needs
        long cap_tiles = (long)((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
        long g = cap_tiles < 2048 ? cap_tiles : 2048;
        dim3 pg(xcd_grid(g < 1 ? 1 : g));
        static bool attr_set_p = false;
        if (lds > 65536 && !attr_set_p) {
            hipFuncSetAttribute((const void*)igemm_fprop_persistent_kernel<T, BM, BN, KS, MG_MODE_CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)igemm_fprop_persistent_kernel<T, BM, BN, KS, MG_MODE_GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set_p = true;
        }
        if (p.xf_scale) {                                    // sparse head: BatchNorm1d + activation of the producing layer applied on the operand's way into LDS
            if constexpr (sizeof(T) == 2 && BM == 128 && BN <= 32 && KS <= 2) {
                if (!fprop_xf_rows_ok(p)) return MG_XF_UNSUPPORTED;
                if (p.mode == MG_MODE_CONV) hipLaunchKernelGGL((igemm_fprop_persistent_kernel<T, BM, BN, KS, MG_MODE_CONV, true>), pg, dim3(256), lds, st, p);
                else hipLaunchKernelGGL((igemm_fprop_persistent_kernel<T, BM, BN, KS, MG_MODE_GATHER, true>), pg, dim3(256), lds, st, p);
                MG_CHECK_LAUNCH();
                return 0;
            } else return MG_XF_UNSUPPORTED;
        }
        if (p.mode == MG_MODE_CONV) hipLaunchKernelGGL((igemm_fprop_persistent_kernel<T, BM, BN, KS, MG_MODE_CONV>), pg, dim3(256), lds, st, p);
        else if (p.mode == MG_MODE_GATHER) hipLaunchKernelGGL((igemm_fprop_persistent_kernel<T, BM, BN, KS, MG_MODE_GATHER>), pg, dim3(256), lds, st, p);
        else return -2;
        MG_CHECK_LAUNCH();
        return 0;
    }
    if (p.xf_scale) return MG_XF_UNSUPPORTED;
    switch (p.mode) {
        case MG_MODE_CONV: hipLaunchKernelGGL((igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_CONV>), grid, dim3(256), lds, st, p, (float*)nullptr, 1); break;
        case MG_MODE_TCONV: hipLaunchKernelGGL((igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_TCONV>), grid, dim3(256), lds, st, p, (float*)nullptr, 1); break;
        case MG_MODE_GATHER: hipLaunchKernelGGL((igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_GATHER>), grid, dim3(256), lds, st, p, (float*)nullptr, 1); break;
        default: return -2;
    }
    MG_CHECK_LAUNCH();
    return 0;
}

template <typename T, int KS>
int dispatch_fprop_ks(const mg_conv_params& p, hipStream_t st) {
    // Largest tile that still yields >= `want` blocks: these GEMMs are small, so exposed load latency is hidden by having
    // several co-resident blocks per CU (256 CUs), not by a deeper per-block pipeline.
    static const long want = [] { const char* e = getenv("MG_FPROP_BLOCKS"); return e ? atol(e) : 768l; }();
    auto blocks = [&](int bm, int bn) { return row_tiles(p, p.M, bm) * ((p.Cout + bn - 1) / bn); };
    static const long want_small = [] { const char* e = getenv("MG_FPROP_BLOCKS_SMALL"); return e ? atol(e) : 300l; }();   // below this many 64x64 blocks, 64x32 tiles (C512 16x16: +11 %, C256 32x32: +4 %)
    if (p.Cout > 64) {
        if (blocks(128, 128) >= want) return launch_fprop<T, 128, 128, KS>(p, st);
        if (blocks(128, 64) >= want) return launch_fprop<T, 128, 64, KS>(p, st);
        if (blocks(64, 64) >= want_small) return launch_fprop<T, 64, 64, KS>(p, st);
        return launch_fprop<T, 64, 32, KS>(p, st);
    }
    if (p.Cout > 32) {
        if (blocks(128, 64) >= want) return launch_fprop<T, 128, 64, KS>(p, st);
        if (blocks(64, 64) >= want_small) return launch_fprop<T, 64, 64, KS>(p, st);
        return launch_fprop<T, 64, 32, KS>(p, st);
    }
    if (p.Cout > 16) return launch_fprop<T, 128, 32, KS>(p, st);
    return launch_fprop<T, 128, 16, KS>(p, st);
}

// ---- split-K: sum the partial slabs, then the SAME epilogue arithmetic as igemm_fprop_kernel (pre-activation, scale/shift,
// residual, activation, post residual, rounding to T) and the BatchNorm statistics of the rounded values ----------------------
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const mg_conv_params p, const float* __restrict__ ws, int splits, int rows_per_block) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    __shared__ float sred[256 * 2 * CE];
    const int cpr = (p.Cout + CE - 1) / CE;                // 16-byte chunks per row
    const int tx = cpr < 32 ? cpr : 32;                    // chunk columns per block (cpr is a multiple of tx or handled by the guard)
    const int ty = 256 / tx;
    const int ix = threadIdx.x % tx, iy = threadIdx.x / tx;
    const int cc = blockIdx.y * tx + ix;
    const int cbase = cc * CE;
    const bool active = iy < ty && cbase < p.Cout;
    float sc[CE], sh[CE], s1[CE], s2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        const int c = cbase + e;
        sc[e] = (p.scale && c < p.Cout) ? p.scale[c] : 1.f;
        sh[e] = (p.shift && c < p.Cout) ? p.shift[c] : 0.f;
        s1[e] = 0.f; s2[e] = 0.f;
    }
    const bool full_vec = (cbase + CE <= p.Cout);
    T* __restrict__ yb = (T*)p.y;
    const T* __restrict__ r1b = (const T*)p.res;
    const T* __restrict__ r2b = (const T*)p.res2;
    const long slab = (long)p.M * p.Cout;
    const int mbeg = blockIdx.x * rows_per_block, mend = min(p.M, mbeg + rows_per_block);
    if (active) {
        for (int m = mbeg + iy; m < mend; m += ty) {
            float v[CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) v[e] = 0.f;
            const float* src = ws + (long)m * p.Cout + cbase;
            for (int s = 0; s < splits; ++s) {
                if (full_vec && (p.Cout % 4) == 0) {
#pragma unroll
                    for (int q = 0; q < CE / 4; ++q) {
                        const float4 a = *(const float4*)(src + s * slab + q * 4);
                        v[q * 4 + 0] += a.x; v[q * 4 + 1] += a.y; v[q * 4 + 2] += a.z; v[q * 4 + 3] += a.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < CE; ++e) if (cbase + e < p.Cout) v[e] += src[s * slab + e];
                }
            }
            long rrow = m;
            if (p.res_mode == 2) {
                const int hw = p.Hout * p.Wout;
                const int n = m / hw, rem = m - n * hw, ho = rem / p.Wout, wo = rem - ho * p.Wout;
                rrow = ((long)n * (p.Hout >> 1) + (ho >> 1)) * (p.Wout >> 1) + (wo >> 1);
            }
            float rv[CE], rv2[CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) { rv[e] = 0.f; rv2[e] = 0.f; }
            if (r1b) {
#pragma unroll
                for (int e = 0; e < CE; ++e) if (cbase + e < p.Cout) rv[e] = TR::ld(r1b + rrow * p.ldr + cbase + e);
            }
            if (r2b) {
#pragma unroll
                for (int e = 0; e < CE; ++e) if (cbase + e < p.Cout) rv2[e] = TR::ld(r2b + (long)m * p.ldr2 + cbase + e);
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                float x = v[e];
                if (p.pre_act) x = apply_act(x, p.act, p.slope);
                x = x * sc[e] + sh[e];
                x += rv[e];
                if (!p.pre_act) x = apply_act(x, p.act, p.slope);
                x += rv2[e];
                v[e] = x;
            }
            float bxv[CE];
            if (p.bnb_x) {                                     // BatchNorm-backward sums ride on the data-gradient tile (see tile_epilogue_impl)
                const T* bxb = (const T*)p.bnb_x; const T* byb = (const T*)p.bnb_y;
                const float bsl = p.bnb_act == MG_ACT_NONE ? 1.f : (p.bnb_act == MG_ACT_RELU ? 0.f : p.slope);
#pragma unroll
                for (int e = 0; e < CE; ++e) {
                    const bool ok = cbase + e < p.Cout;
                    const float bx_ = ok ? TR::ld(bxb + (long)m * p.bnb_ld + cbase + e) : 0.f;
                    bxv[e] = ok ? (bx_ - p.bnb_mean[cbase + e]) * p.bnb_invstd[cbase + e] : 0.f;
                    if (byb && ok && !(TR::ld(byb + (long)m * p.bnb_ld + cbase + e) > 0.f)) v[e] *= bsl;
                    else if (!byb && p.bnb_scale && ok && !(bx_ * p.bnb_scale[cbase + e] + p.bnb_shift[cbase + e] > 0.f)) v[e] *= bsl;
                }
            }
            const uint4 packed = TR::pack(v);                  // rounded once; the statistics are those of the rounded values
            if (p.stats) {
                TR::unpack(packed, v);
                if (p.bnb_x) {
#pragma unroll
                    for (int e = 0; e < CE; ++e) { s1[e] += v[e]; s2[e] += v[e] * bxv[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < CE; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
                }
            }
            T* dst = yb + (long)m * p.ldy + p.yoff + cbase;
            if (full_vec) *(uint4*)dst = packed;
            else {
#pragma unroll
                for (int e = 0; e < CE; ++e) if (cbase + e < p.Cout) TR::st(dst + e, v[e]);
            }
        }
    }
    if (p.stats) {                                          // uniform per launch: every thread reaches the barrier
        const int width = tx * 2 * CE;
        if (active) {
#pragma unroll
            for (int e = 0; e < CE; ++e) { sred[iy * width + ix * 2 * CE + e] = s1[e]; sred[iy * width + ix * 2 * CE + CE + e] = s2[e]; }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < width; j += 256) {
            float a = 0.f;
            for (int r = 0; r < ty; ++r) a += sred[r * width + j];
            const int cx = j / (2 * CE), k = j - cx * 2 * CE, sq = k / CE, e = k - sq * CE;
            const int c = (blockIdx.y * tx + cx) * CE + e;
            if (c < p.Cout) {
                if (p.stat_mode == 1) { if (!sq) atomicAdd(&p.stats[c], a); }
                else atomicAdd(&p.stats[(size_t)(blockIdx.x % (unsigned)(p.stat_rep > 0 ? p.stat_rep : MG_STAT_REPLICAS)) * 2 * p.Cout + (sq ? p.Cout : 0) + c], a);
            }
        }
    }
}

struct SplitPlan { int bn, splits; };
// Which layers: dense convs whose 64x64 tiling yields too few blocks (they run 64x32 tiles today) and whose K is deep enough
// (>= 18 stages of 4 slabs). MG_FPROP_SPLITK=0 switches the path off.
template <typename T>
static SplitPlan plan_splitk(const mg_conv_params& p) {
    SplitPlan sp{0, 1};
    static const int enabled = [] { const char* e = getenv("MG_FPROP_SPLITK"); return e ? atoi(e) : 1; }();
    static const long want_small = [] { const char* e = getenv("MG_FPROP_BLOCKS_SMALL"); return e ? atol(e) : 300l; }();
    if (!enabled || p.mode == MG_MODE_GATHER || p.Cout < 64 || p.M > 8192 || p.m_dev) return sp;
    if (tconv_phased(p)) return sp;              // a quarter of the taps per phase: the K walk is short again
    if (halo_eligible(p)) return sp;             // the halo-tile kernel beats split-K on the deep 3x3 layers (C256 32x32: 30 -> 14 us)
    constexpr int EPS = ElemTraits<T>::EPS;
    const int nslab = (p.R * p.S * p.Cin + EPS - 1) / EPS;
    const int nstage = (nslab + 3) / 4;
    auto blocks = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); };
    static const int min_stages = [] { const char* e = getenv("MG_SPLITK_MIN_K_STAGES"); return e ? atoi(e) : 18; }();   // K >= 2304 (bf16). Stand-alone launches: 48 -> 30 us
    // at 16x16x512 (K 4608), 53 -> 34 us at 32x32 512->256, even at K = 2304, a loss at K = 1152; inside the captured graphs (no host
    // cost for the extra launch) the forward + backward graphs take 13.60 / 13.43 / 13.25 / 13.40 ms for off / 24 / 18 / 9 stages
    if (nstage < min_stages || blocks(64, 64) >= want_small) return sp;
    sp.bn = p.Cout >= 128 ? 128 : 64;
    const long tiles = blocks(128, sp.bn);
    static const long target = [] { const char* e = getenv("MG_SPLITK_BLOCKS"); return e ? atol(e) : 512l; }();
    static const long smax = [] { const char* e = getenv("MG_SPLITK_MAX"); return e ? atol(e) : 8l; }();
    static const long smin_stages = [] { const char* e = getenv("MG_SPLITK_MIN_STAGES"); return e ? atol(e) : 3l; }();
    long s = (target + tiles - 1) / tiles;
    if (s > nstage / smin_stages) s = nstage / smin_stages;
    if (s > smax) s = smax;
    if (s < 2) { sp.bn = 0; return sp; }
    sp.splits = (int)s;
    return sp;
}

template <typename T, int BN>
static int launch_fprop_split(const mg_conv_params& p, float* ws, int splits, hipStream_t st) {
    constexpr int BM = 128, KS = 4;
    dim3 grid(xcd_grid((long)((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN) * splits));
    constexpr size_t lds = lds_bytes<BM, BN, KS>();
    static bool attr_set = false;
    if (lds > 65536 && !attr_set) {
        hipFuncSetAttribute((const void*)igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_CONV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_TCONV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.mode == MG_MODE_CONV) hipLaunchKernelGGL((igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_CONV, true>), grid, dim3(256), lds, st, p, ws, splits);
    else hipLaunchKernelGGL((igemm_fprop_kernel<T, BM, BN, KS, MG_MODE_TCONV, true>), grid, dim3(256), lds, st, p, ws, splits);
    MG_CHECK_LAUNCH();
    constexpr int CE = ElemTraits<T>::CE;
    const int cpr = (p.Cout + CE - 1) / CE, tx = cpr < 32 ? cpr : 32, ty = 256 / tx;
    const int groups = (cpr + tx - 1) / tx;
    int rb = (p.M + ty * 2 - 1) / (ty * 2);
    if (rb < 1) rb = 1;
    if (rb > 512) rb = 512;
    const int rpb = (p.M + rb - 1) / rb;
    if (int rcs = stat_rows_check(p, (p.M + rpb - 1) / rpb)) return rcs;
    hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3((p.M + rpb - 1) / rpb, groups), dim3(256), 0, st, p, (const float*)ws, splits, rpb);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 forward conv over the 8-CHANNEL network input (Cin == 8, Cout <= 32; bf16): HBM-bound (16 B in, 64 B out per pixel). One
// block = an 8x16-pixel tile (16x16 measured slower: 36 vs 30 us at 512x512): the halo image (16 B per pixel) is staged once; K = 9 taps x 8 channels is walked as three 32-wide MFMA
// steps whose A fragment of lane (pixel, k-group) IS one halo pixel (the 8 channels of tap 4*step + k-group: a 16-byte LDS read at a constant
// offset; taps 9..11 read a zero pixel); the weights (72 x Cout) live in registers. The im2col kernel re-read the input nine times through L2 (40 us isolated, 63 us in the step with the statistics epilogue; now 30 / 51).
// ---------------------------------------------------------------------------------------------------------------------
template <int TH, typename T = bf16raw>
__global__ __launch_bounds__(256) void igemm_fprop_c8_kernel(const mg_conv_params p) {
    constexpr int TW = 16, HP = 18, BM = TH * TW, BN = 32, FM = TH / 4, FN = 2, WM = BM / 4, WN = 32;
    constexpr int NH = (TH + 2) * HP;                             // halo pixels (+ 1 zero pixel)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* sH = (uint4*)(smem + ctile_bytes<BM, BN>());           // behind the epilogue's fp32 tile: no overlay hazards
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = p.Hout, W = p.Wout;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int work;
    if (!xcd_order(p.N * tiles_y * tiles_x, work)) return;
    const int img = work / (tiles_y * tiles_x);
    const int trem = work - img * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const T* __restrict__ xb = (const T*)p.x;
    const T* __restrict__ wb = (const T*)p.w;

    u32x4 fb[3][FN];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int tap = s * 4 + lg, co = j * 16 + lr;
            fb[s][j] = (tap < 9 && co < p.Cout) ? *(const u32x4*)(wb + ((long)co * 9 + tap) * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int idx = t + r * 256;
        if (idx < NH) {
            const int hy = idx / HP, hx = idx - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            sH[idx] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? *(const uint4*)(xb + ((long)(img * H + iy) * W + ix) * p.ldx)
                                                                                 : make_uint4(0, 0, 0, 0);
        } else if (idx == NH) sH[NH] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = s * 4 + lg;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int ty = wm * FM + i;
            const int px = tap < 9 ? (ty + ky) * HP + lr + kx : NH;
            const u32x4 fa = *(const u32x4*)&sH[px];
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = mfma16<T>(fa, fb[s][j], acc[i][j]);
        }
    }
    auto rowmap = [&](int rt) -> long {
        const int y = y0 + rt / TW, x = x0 + (rt % TW);
        return (y < H && x < W) ? ((long)img * H + y) * W + x : -1l;
    };
    tile_epilogue<T, BM, BN, FM, FN>(p, acc, wm, 0, WM, WN, 0, work, smem, rowmap);
}

static inline bool fprop_c8_eligible(const mg_conv_params& p) {
    static const int on = [] { const char* e = getenv("MG_FPROP_C8"); return e ? atoi(e) : 1; }();
    return on && MG_IS16(p.dtype) && p.mode == MG_MODE_CONV && !p.m_dev && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && p.dil == 1 &&
           p.Hout == p.Hin && p.Wout == p.Win && p.Cin == 8 && p.Cout <= 32 && p.ldx % 8 == 0;
}
template <typename T>
static int launch_fprop_c8(const mg_conv_params& p, hipStream_t st) {
    static const int th = [] { const char* e = getenv("MG_FPROP_C8_TH"); return e ? atoi(e) : 8; }();
    if (int rcs = stat_rows_check(p, (long)p.N * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16))) return rcs;
    if (th == 16) {
        const long tiles = (long)p.N * ((p.Hout + 15) / 16) * ((p.Wout + 15) / 16);
        const size_t lds = (size_t)ctile_bytes<256, 32>() + (18 * 18 + 1) * 16;
        hipLaunchKernelGGL((igemm_fprop_c8_kernel<16, T>), dim3(xcd_grid(tiles)), dim3(256), lds, st, p);
    } else {
        const long tiles = (long)p.N * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
        const size_t lds = (size_t)ctile_bytes<128, 32>() + (10 * 18 + 1) * 16;
        hipLaunchKernelGGL((igemm_fprop_c8_kernel<8, T>), dim3(xcd_grid(tiles)), dim3(256), lds, st, p);
    }
    MG_CHECK_LAUNCH();
    return 0;
}

// operand transform (mg_conv_params.xf_*): lives in the one- / two-slab halo forms of the forward 3x3 / stride-1 convolution
static inline bool fprop_xf_ok(const mg_conv_params& p) {
    return MG_IS16(p.dtype) && p.mode == MG_MODE_CONV && !p.bnb_x && !fprop_c8_eligible(p) && halo_eligible(p) && p.Cin <= 512 &&
           (p.Cin < 96 ? p.Cout > 16 : p.Hout >= 8);
}

template <typename T>
int dispatch_fprop(const mg_conv_params& p, hipStream_t st) {
    if (p.xf_scale) {
        if constexpr (sizeof(T) == 2) {
            if (fprop_xf_ok(p)) return dispatch_fprop_halo<T>(p, st);
            if (fprop_xf_rows_ok(p)) {                       // (the stage-width choice of the ordinary path below)
                const int nsl = p.R * p.S * p.Cin / 32;
                return nsl >= 8 ? dispatch_fprop_ks<T, 2>(p, st) : dispatch_fprop_ks<T, 1>(p, st);
            }
        }
        return MG_XF_UNSUPPORTED;
    }
    if constexpr (sizeof(T) == 2) {
        if (!p.bnb_x && fprop_c8_eligible(p)) return launch_fprop_c8<T>(p, st);
        if (halo_eligible(p)) return dispatch_fprop_halo<T>(p, st);
    }
    const int eps = sizeof(T) == 2 ? 32 : 16;
    if (tconv_phased(p)) {                       // stage width by the longest phase walk (ceil(R/2) * ceil(S/2) taps)
        const int nsl = ((p.R + 1) / 2) * ((p.S + 1) / 2) * (p.Cin / eps);
        if (nsl >= 8 && nsl <= 18) return dispatch_fprop_ks<T, 2>(p, st);
        if (nsl >= 8) return dispatch_fprop_ks<T, 4>(p, st);
        return dispatch_fprop_ks<T, 1>(p, st);
    }
    if constexpr (sizeof(T) == 2) {
        if (async_eligible(p)) return dispatch_fprop_async<T>(p, st);
    }
    const int nslab = (p.R * p.S * p.Cin + eps - 1) / eps;
    // stage width: 4 slabs (256 B of K per row) for the K-heavy layers, 2 slabs for K <= 576 (C32 / C64 3x3 layers: half the LDS
    // and staging registers -> more co-resident blocks; measured +31 % on the 512x512 C32 layers, +8 % on C64), 1 for tiny K
    static const int ks2_max = [] { const char* e = getenv("MG_FPROP_KS2_MAX"); return e ? atoi(e) : 18; }();
    if (nslab >= 8 && nslab <= ks2_max) return dispatch_fprop_ks<T, 2>(p, st);
    if (nslab >= 8) return dispatch_fprop_ks<T, 4>(p, st);
    return dispatch_fprop_ks<T, 1>(p, st);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Entry points. The kernel templates above are instantiated once per storage type (fp32, bf16, fp16); to keep the build short the three
// families compile as three objects of THIS file in parallel (__graft_entry__.build() reads the "build-variants" line at the top and passes
// -DMG_CONV_T=<code>): an object defines the internal per-type functions of its type, the MG_F32 object also the public entry points.
// Without -DMG_CONV_T (a plain `hipcc -c conv_igemm.hip`) everything lands in one object.
// ---------------------------------------------------------------------------------------------------------------------
#define MG_CONV_TYPE_FUNCS(NAME, T)                                                                                                            \
    extern "C" __attribute__((visibility("hidden"))) int mg_conv_fprop_##NAME(const mg_conv_params* pp, void* stream) {                        \
        return dispatch_fprop<T>(*pp, (hipStream_t)stream);                                                                                    \
    }                                                                                                                                          \
    extern "C" __attribute__((visibility("hidden"))) int mg_conv_fprop_split_##NAME(const mg_conv_params* pp, float* workspace, void* stream) { \
        const SplitPlan sp = plan_splitk<T>(*pp);                                                                                              \
        return sp.bn == 128 ? launch_fprop_split<T, 128>(*pp, workspace, sp.splits, (hipStream_t)stream)                                       \
                            : launch_fprop_split<T, 64>(*pp, workspace, sp.splits, (hipStream_t)stream);                                       \
    }
#if !defined(MG_CONV_T) || MG_CONV_T == MG_F32
MG_CONV_TYPE_FUNCS(f32, float)
#endif
#if !defined(MG_CONV_T) || MG_CONV_T == MG_BF16
MG_CONV_TYPE_FUNCS(bf16, bf16raw)
#endif
#if !defined(MG_CONV_T) || MG_CONV_T == MG_F16
MG_CONV_TYPE_FUNCS(f16, f16raw)
#endif

#if !defined(MG_CONV_T) || MG_CONV_T == MG_F32
extern "C" int mg_conv_fprop_f32(const mg_conv_params*, void*);
extern "C" int mg_conv_fprop_bf16(const mg_conv_params*, void*);
extern "C" int mg_conv_fprop_f16(const mg_conv_params*, void*);
extern "C" int mg_conv_fprop_split_f32(const mg_conv_params*, float*, void*);
extern "C" int mg_conv_fprop_split_bf16(const mg_conv_params*, float*, void*);
extern "C" int mg_conv_fprop_split_f16(const mg_conv_params*, float*, void*);

extern "C" long mg_conv_fprop_workspace(const mg_conv_params* pp) {
    if (!pp || pp->M <= 0) return 0;
    const SplitPlan sp = MG_IS16(pp->dtype) ? plan_splitk<bf16raw>(*pp) : plan_splitk<float>(*pp);
    return sp.bn ? (long)sp.splits * pp->M * pp->Cout : 0;
}

bool mg_wgrad_xform_ok(const mg_conv_params& p);          // conv_wgrad.hip

extern "C" int mg_conv_xform_ok(const mg_conv_params* pp, int which) {
    if (!pp || pp->M <= 0) return 0;
    return which == 0 ? (int)fprop_xf_ok(*pp) : (which == 1 ? (int)mg_wgrad_xform_ok(*pp) : 0);
}

/* Rows a BatchNorm statistics buffer needs so that every output tile of ANY forward kernel form of this file owns a row (deterministic mode:
 * one addition per word, mg_conv_params.stat_rep): the single source of this bound, next to the tile shapes it depends on (the
 * binding used to re-derive it). Forms: spatial halo tiles of >= 4 x 16 pixels (halo / c8 kernels); row tiles of >= 64 rows, four padded phases
 * for a stride-2 transposed walk (im2col and direct-to-LDS forms: row_tiles(p, M, 64) <= M / 64 + 8); the split-K finish kernel's row blocks
 * (M <= 8192: at most 512 blocks of >= 16 rows). Atomic mode: the 32 replicas. */
extern "C" int mg_conv_stat_rows(int M, int N, int Hout, int Wout) {
    if (!mg_det_on) return MG_STAT_REPLICAS;
    long rows = (long)N * ((Hout + 3) / 4) * ((Wout + 15) / 16);
    const long by_rows = (M + 63) / 64 + 8;
    if (by_rows > rows) rows = by_rows;
    if (rows < MG_STAT_REPLICAS + 1) rows = MG_STAT_REPLICAS + 1;
    if (M <= 8192) {
        long fin = (M + 15) / 16 + 1;
        if (fin > 512) fin = 512;
        if (fin > rows) rows = fin;
    }
    return (int)rows;
}

static int conv_fprop_check(const mg_conv_params& p) {
    const int ce = MG_IS16(p.dtype) ? 8 : 4;
    if (p.Cin % ce != 0 || p.ldx % ce != 0) return -3;                 // K chunks must not straddle taps / be 16-B aligned
    if (p.Cout >= ce && (p.ldy % ce != 0 || p.yoff % ce != 0)) return -4;
    if (p.mode == MG_MODE_GATHER && !p.nbr) return -5;
    return 0;
}

extern "C" int mg_conv_fprop(const mg_conv_params* pp, void* stream);

extern "C" int mg_conv_halo3(const mg_conv_params* pp, void* stream);        // conv_halo3.hip: 1 = not a layer of that form

extern "C" int mg_conv_fprop_ws(const mg_conv_params* pp, float* workspace, long workspace_floats, void* stream) {
    if (!pp) return -1;
    const mg_conv_params& p = *pp;
    if (p.M <= 0) return 0;
    // the 3x3 / stride-1 layers of the 16 x 16 and 32 x 32 maps: the round-6 halo form (4 x 16 pixel tiles, four-slab ring) beats split-K + finish
    // (C512 16 x 16: 13 us against 21 + 9.5), needs no workspace and keeps one launch
    if (conv_fprop_check(p) == 0) { int rc = mg_conv_halo3(pp, stream); if (rc != 1) return rc; }
    const long need = mg_conv_fprop_workspace(pp);
    if (!need || !workspace || workspace_floats < need) return mg_conv_fprop(pp, stream);
    int rc = conv_fprop_check(p); if (rc) return rc;
    if (p.Cout % (MG_IS16(p.dtype) ? 8 : 4)) return mg_conv_fprop(pp, stream);
    if (p.dtype == MG_BF16) return mg_conv_fprop_split_bf16(pp, workspace, stream);
    if (p.dtype == MG_F16) return mg_conv_fprop_split_f16(pp, workspace, stream);
    if (p.dtype == MG_F32) return mg_conv_fprop_split_f32(pp, workspace, stream);
    return -6;
}

extern "C" int mg_conv_fprop(const mg_conv_params* pp, void* stream) {
    if (!pp) return -1;
    const mg_conv_params& p = *pp;
    if (p.M <= 0) return 0;
    { int rc = conv_fprop_check(p); if (rc) return rc; }
    { int rc = mg_conv_halo3(pp, stream); if (rc != 1) return rc; }
    if (p.dtype == MG_BF16) return mg_conv_fprop_bf16(pp, stream);
    if (p.dtype == MG_F16) return mg_conv_fprop_f16(pp, stream);
    if (p.dtype == MG_F32) return mg_conv_fprop_f32(pp, stream);
    return -6;
}
#endif
