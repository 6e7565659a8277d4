// Spatial halo-tile convolution, round-6 form ("halo3"): the 3x3 / stride 1 / pad 1 layers with Cin % 32 == 0 in the 16-bit storage types --
// the encoder's BasicBlocks, the shortcut branches and the decoder convs, forward (MG_MODE_CONV) and data gradient (MG_MODE_TCONV: the same
// convolution with mirrored taps over the transposed weights): maggie/network/encoder/resnet.py:23-39,167-175, decoder/resnet.py:20-45.
//
// What the s_memtime timeline of the round-2 halo kernel showed (tools/halo_timeline.py, C128 64x64, batch 4: 13.3 k cycles per tile of
// which 2.3 k are MFMA issue): (1) every wave issued its LDS-DMA instructions of the next stage as ONE burst behind the stage barrier --
// 9 x 1 KiB per wave, 72 KiB per CU against a 64 B / clk vector-memory path: ~1100 cycles in which the wave sits in the issue queue and
// its MFMAs wait; (2) with a two-slab ring the next stage had one compute phase (~950 cycles) to land and did not (+400 cycles of
// vmcnt wait per stage); (3) the epilogue took the accumulators through an fp32 LDS tile, two barriers and a second pass (2.8 k cycles);
// (4) two 128 x 32 tiles per CU staged the same halo twice. This form:
//   * operands swapped in the MFMA (A := weight fragment, B := pixel fragment), so a lane's accumulator registers are 4 x FN CONSECUTIVE
//     output channels of ONE pixel (the weight rows are permuted on their way into LDS: row r of fragment j = channel r * FN + j). The
//     epilogue -- scale / shift, residuals, activation, rounding, 16-byte stores, BatchNorm statistics by DPP row sums -- runs from the
//     accumulator registers: no LDS tile, no barrier before the stores;
//   * the LDS-DMA instructions of stage s + NS - 1 are spread between the MFMAs of stage s (one per walk step), never a burst;
//   * ring depth NS up to 4 (160 KiB of LDS: one workgroup per CU owns it), tile 8 x 16 pixels x 64 channels where that still gives
//     >= one workgroup per CU (halo staged once for both channel halves);
//   * row-sliding tap walk of round 5 kept (9 FN + 3 (FM + 2) fragment reads per stage).
// LDS image of a stage: halo [TH + 2][24 px][64 B] (slot = chunk ^ 2 * ((px >> 2) & 1)) | weights [9 taps][BN rows][64 B]
// (slot = chunk ^ 3 * ((row >> 3) & 1)), both conflict-free for ds_read_b128 (as in conv_igemm.hip).
#include "common.h"
#include "conv_xcd.h"
#include "../../include/maggie_hip.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include <stdio.h>

namespace {

__device__ uint4 mg_h3_zero_page[4];            // zero-initialised device memory: what a lane outside the image / beyond Cout fetches

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

template <int... Ks, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Ks...>, F&& f) { (f(std::integral_constant<int, Ks>{}), ...); }

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (fixed order: a function of the lane layout only)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);              // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);              // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);             // row_half_mirror
    v += dpp_move<0x140>(v);             // row_mirror
    return v;
}

#ifdef MG_H3_TIMING
__device__ long long mg_h3_dbg[32 * 32];
#define H3_STAMP(i) do { if (threadIdx.x == 0 && (work & 31) == 0 && work / 32 < 32) mg_h3_dbg[(work / 32) * 32 + (i)] = (i) >= 30 ? (long long)wall_clock64() : (long long)clock64(); } while (0)
#else
#define H3_STAMP(i)
#endif

template <int TH, int BN, int NS> struct H3Cfg {
    static constexpr int TW = 16, BM = TH * TW, PW = 24, HH = TH + 2;
    static constexpr int A_INSTR = (HH * PW + 15) / 16, A_PER_WAVE = (A_INSTR + 3) / 4, A_BYTES = A_PER_WAVE * 4 * 1024;
    static constexpr int B_GRP = BN / 16, B_INSTR = 9 * B_GRP, B_PER_WAVE = (B_INSTR + 3) / 4, B_BYTES = B_PER_WAVE * 4 * 1024;
    static constexpr int STAGE = A_BYTES + B_BYTES, L = A_PER_WAVE + B_PER_WAVE;
    static constexpr int WAVES_N = BN >= 64 ? 2 : 1, WAVES_M = 4 / WAVES_N;
    static constexpr int FM = TH / WAVES_M, FN = 2, WN = 16 * FN;
    static constexpr int STAT_BYTES = WAVES_M * 2 * BN * 4;
    static constexpr int LDS = NS * STAGE > STAT_BYTES ? NS * STAGE : STAT_BYTES;
    static_assert(WAVES_N * WN == BN && FM >= 1 && TH % WAVES_M == 0, "halo3 tile configuration");
    static_assert((NS - 1) * L <= 63 || NS == 1, "vmcnt is a 6-bit counter");
};

#ifndef MG_H3_EXP
#define MG_H3_EXP 0      /* timing experiments only (results wrong): 1 = producers stop after the first ring fill, 2 = consumers only pass the barriers */
#endif
// MG_H3_EARLY = 1: a consumer wave arrives at the next stage's barrier (and issues that stage's first fragment reads) as soon as its LAST LDS read of
// the current stage has landed, ~20 MFMAs before the end of the stage. Correct (two-rank runs 6 of 6 once the lgkmcnt counts of h3_walk were right: the
// NaNs first blamed on this hand-over were the counts) and worth 1.4 % summed over the trunk's shapes, nothing in the step (798.0 / 796.6 against
// 796.3 / 799.2 inst-frames/s on one lease): off, the hand-over sits behind the stage's last MFMA. The ISA test covers both.
#ifndef MG_H3_EARLY
#define MG_H3_EARLY 0
#endif
#ifndef MG_H3_AD
#define MG_H3_AD 6
#endif

#ifdef MG_H3_DRAIN
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }     // (experiment: no counted waits)
#else
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif

__device__ __forceinline__ void h3_load_affine(const mg_conv_params& p, int c0, float (&sc)[8], float (&sh)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (c0 < p.Cout) {                                       // (Cout % 8 == 0: all or nothing)
        if (p.scale) { *(float4*)&sc[0] = *(const float4*)(p.scale + c0); *(float4*)&sc[4] = *(const float4*)(p.scale + c0 + 4); }
        if (p.shift) { *(float4*)&sh[0] = *(const float4*)(p.shift + c0); *(float4*)&sh[4] = *(const float4*)(p.shift + c0 + 4); }
    }
}

// One stage (a 32-channel slab, all nine taps) of a wave's FM x FN accumulator block from the LDS image at `sb`.
// Row-sliding tap walk: for column shift c = 0, 1, 2 the halo rows r = 0 .. FM + 1 of this wave stream through an AD-deep fragment ring and
// meet the three weight fragments W(ky, c); W(0, c + 1) and W(1, c + 1) are loaded over their dead predecessors during the last two rows of
// column c, W(2, c) during row 0. One continuous LDS stream with compile-time lgkmcnt counts (inline asm: hipcc otherwise drains vmcnt(0)
// before every LDS read it can see next to LDS-DMA). A := weight fragment, B := pixel fragment.
// Across stages: the fragment registers exist twice (PAR): `tail()` -- the caller's barrier that hands the buffer back and publishes stage s + 1, and
// stage s + 1's first nine fragment reads into the other register set -- runs behind the stage's last MFMA, or, with MG_H3_EARLY, as soon as the last
// LDS read of stage s has been issued and has landed (walk position KT = NSTEP - AD + 1, ~20 MFMAs earlier).
template <int FN, int AD> struct H3Frags { u32x4 b[2][3][FN]; u32x4 a[2][AD]; };

template <int FM, int FN, int BN, int PW, int MODE, int AD, int PAR>
__device__ __forceinline__ void h3_first_reads(const unsigned sb, const unsigned (&a_lane)[3], const unsigned b_lane, H3Frags<FN, AD>& fr) {
    constexpr int NR = FM + 2, NSTEP = 3 * NR;
    const unsigned ba = sb + b_lane;
    static_for(std::make_integer_sequence<int, 3>{}, [&](auto ky_c) {
        constexpr int KY = decltype(ky_c)::value;
        constexpr int TAP = MODE == MG_MODE_TCONV ? (2 - KY) * 3 + 2 : KY * 3;
        u32x4(&rb)[FN] = fr.b[PAR][KY];
        const unsigned ba_ = ba;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[0]) : "v"(ba_), "n"(TAP * BN * 64 + 0 * 1024) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[1]) : "v"(ba_), "n"(TAP * BN * 64 + 1 * 1024) : "memory");
    });
    const unsigned aa = sb + a_lane[0];
    static_for(std::make_integer_sequence<int, (AD - 1 < NSTEP ? AD - 1 : NSTEP)>{}, [&](auto d_c) {
        constexpr int R_ = decltype(d_c)::value;               // AD - 1 <= NR: all in column 0
        u32x4& ra = fr.a[PAR][R_ % AD];
        const unsigned aa_ = aa;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra) : "v"(aa_), "n"(R_ * PW * 64) : "memory");
    });
}

template <typename T, int FM, int FN, int BN, int PW, int MODE, int AD, int PAR, typename Tail>
__device__ __forceinline__ void h3_walk(const unsigned sb, const unsigned (&a_lane)[3], const unsigned b_lane, f32x4 (&acc)[FM][FN],
                                        H3Frags<FN, AD>& fr, Tail&& tail) {
    static_assert(FN == 2, "two weight fragments per wave");
    constexpr int NR = FM + 2, NSTEP = 3 * NR;
    static_assert(AD - 1 <= NR && AD >= 3, "the first reads of a stage stay inside column 0");
#if !MG_H3_EARLY
    constexpr int KT = NSTEP;                                // the hand-over behind the last MFMA of the stage (default, see MG_H3_EARLY above)
#else
    constexpr int KT = NSTEP - AD + 1;                       // first walk position behind the stage's last LDS read
#endif
    const unsigned baddr = sb + b_lane;
    auto read_b = [&](auto ky_c, auto c_c) {
        constexpr int KY = decltype(ky_c)::value, C_ = decltype(c_c)::value;
        constexpr int TAP = MODE == MG_MODE_TCONV ? (2 - KY) * 3 + (2 - C_) : KY * 3 + C_;
        u32x4(&rb)[FN] = fr.b[PAR][KY];
        const unsigned ba = baddr;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[0]) : "v"(ba), "n"(TAP * BN * 64 + 0 * 1024) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[1]) : "v"(ba), "n"(TAP * BN * 64 + 1 * 1024) : "memory");
    };
    auto read_a = [&](auto k_c) {                             // halo row r under column shift c, stream position k = c * NR + r
        constexpr int K_ = decltype(k_c)::value, C_ = K_ / NR, R_ = K_ % NR;
        const unsigned aa = sb + a_lane[C_];
        u32x4& ra = fr.a[PAR][K_ % AD];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra) : "v"(aa), "n"(R_ * PW * 64) : "memory");
    };
    // weight fragments (FN reads each) issued at stream position k: W(0, c + 1) at r == FM, W(1, c + 1) at r == FM + 1, W(2, c) at r == 0 (c > 0)
    auto nb_at = [](int k) { const int c = k / NR, r = k % NR; return k < 0 ? 0 : ((r == FM && c < 2) ? 1 : 0) + ((r == FM + 1 && c < 2) ? 1 : 0) + ((r == 0 && c > 0) ? 1 : 0); };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto step = [&](auto k_c) {
        constexpr int K_ = decltype(k_c)::value, C_ = K_ / NR, R_ = K_ % NR;
        if constexpr (R_ == FM && C_ < 2) read_b(I0{}, std::integral_constant<int, C_ + 1>{});
        if constexpr (R_ == FM + 1 && C_ < 2) read_b(I1{}, std::integral_constant<int, C_ + 1>{});
        if constexpr (R_ == 0 && C_ > 0) read_b(I2{}, std::integral_constant<int, C_>{});
        if constexpr (K_ + AD - 1 < NSTEP) read_a(std::integral_constant<int, K_ + AD - 1>{});
        if constexpr (K_ < KT) {
            // In-order return: an operand has landed once at most the reads issued BEHIND it are outstanding.
            //  * A(K_) was issued at position K_ - AD + 1: behind it the rows K_ + 1 .. K_ + AD - 1 and the weight loads of positions K_ - AD + 2 .. K_
            //    (counted here: those of K_ - 1 and K_ only -- fewer than there are, the safe side);
            //  * weight fragments are first used two positions behind their issue: when position K_ - 2 issued any, only what came behind THEM may be
            //    outstanding -- the row reads of positions K_ - 2, K_ - 1, K_ (three, not AD - 1: the rows K_ + 1 .. K_ + AD - 3 were issued BEFORE those
            //    weights) and the weight loads of K_ - 1 and K_. Round 6 shipped the first count for both with AD = 6: two reads too many, the first use of
            //    W(ky, c + 1) could meet W(ky, c) still in the registers -- never in a single process (an LDS read lands in ~100 cycles, the two positions
            //    are ~250), with a second process on the GPU in 3 runs out of 4 (tests/test_isa_load_chains_cpu.py now checks the counts in the ISA).
            constexpr int rows_after = (K_ + AD - 1 < NSTEP ? AD - 1 : NSTEP - 1 - K_);
            constexpr int rows_behind_w = (K_ - 2 + AD - 1 < NSTEP ? 1 : 0) + (K_ - 1 + AD - 1 < NSTEP ? 1 : 0) + (K_ + AD - 1 < NSTEP ? 1 : 0);
            constexpr int w_after = (nb_at(K_ - 1) + nb_at(K_)) * FN;
            constexpr int after = w_after + ((nb_at(K_ - 2) > 0 && rows_behind_w < rows_after) ? rows_behind_w : rows_after);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(after) : "memory");
        } else if constexpr (K_ == KT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of the stage has landed: the buffer is not needed any more
            tail();
        }                                                    // K_ > KT: operands in registers since KT
        __builtin_amdgcn_sched_barrier(0);
        u32x4& ra = fr.a[PAR][K_ % AD];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int i = R_ - ky;                           // the output row that meets halo row R_ under tap row ky
            if (i >= 0 && i < FM) {
#pragma unroll
                for (int jj = 0; jj < FN; ++jj) acc[i][jj] = mfma16<T>(fr.b[PAR][ky][jj], ra, acc[i][jj]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#if MG_H3_EXP == 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tail();
#else
    static_for(std::make_integer_sequence<int, NSTEP>{}, step);
#if !MG_H3_EARLY
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tail();
#endif
#endif
}

// Epilogue from the accumulator registers: acc[i][j][e] = pixel (y0 + wm * FM + i, x0 + lr), channel c0 + e * FN + j. Consumer waves only; the
// statistics tail has two workgroup barriers (the producers of the split form mirror them).
// Row operands of the epilogue that do not depend on the accumulators -- residual rows, the BatchNorm-link rows (bnb_x / bnb_y) -- are requested BEFORE the
// K loop (one 16-byte load per lane and pixel row each): the consumer waves wait for stage 0 to land anyway, and in the epilogue these loads were an exposed
// memory round trip per launch (linked data-gradient kernels: 15.2 against 12.8 us, tools/trace_bench.sh).
template <int FM, bool RES, bool BNB> struct H3Rows {
    int mrow[FM];                                            // output row (M < 2^31), -1: outside the tensor
    uint4 q1[RES ? FM : 1], q2[RES ? FM : 1], qx[BNB ? FM : 1], qy[BNB ? FM : 1];
};

template <typename T, int TH, int BN, int FM, int FN, int WAVES_N, bool RES, bool BNB>
__device__ __forceinline__ void h3_prefetch_rows(const mg_conv_params& p, H3Rows<FM, RES, BNB>& pr, int wave, int lane, int img, int y0, int x0, int n0) {
    constexpr int WN = 16 * FN;
    const int H = p.Hout, W = p.Wout;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = n0 + wn * WN + lg * 4 * FN;
    const bool col_ok = c0 < p.Cout;
    const int x = x0 + lr;
    [[maybe_unused]] const T* __restrict__ r1b = (const T*)p.res;
    [[maybe_unused]] const T* __restrict__ r2b = (const T*)p.res2;
    [[maybe_unused]] const T* __restrict__ bxb = (const T*)p.bnb_x;
    [[maybe_unused]] const T* __restrict__ byb = (const T*)p.bnb_y;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int y = y0 + wm * FM + i;
        pr.mrow[i] = (col_ok && y < H && x < W) ? (img * H + y) * W + x : -1;
        if constexpr (RES) {
            pr.q1[i] = make_uint4(0, 0, 0, 0); pr.q2[i] = make_uint4(0, 0, 0, 0);
            if (pr.mrow[i] >= 0) {
                if (r1b) {
                    int rrow = pr.mrow[i];
                    if (p.res_mode == 2) rrow = (img * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);   // residual at half resolution (nearest x2)
                    pr.q1[i] = *(const uint4*)(r1b + (long)rrow * p.ldr + c0);
                }
                if (r2b) pr.q2[i] = *(const uint4*)(r2b + (long)pr.mrow[i] * p.ldr2 + c0);
            }
        }
        if constexpr (BNB) {
            pr.qx[i] = make_uint4(0, 0, 0, 0); pr.qy[i] = make_uint4(0, 0, 0, 0);
            if (pr.mrow[i] >= 0) {
                pr.qx[i] = *(const uint4*)(bxb + (long)pr.mrow[i] * p.bnb_ld + c0);
                if (byb) pr.qy[i] = *(const uint4*)(byb + (long)pr.mrow[i] * p.bnb_ld + c0);
            }
        }
    }
}

// BNB (mg_conv_params.bnb_*, MAGGIE_BN_LINK): this launch is the data gradient arriving at the OUTPUT z = act(BN(x)) of a training BatchNorm layer whose only
// consumer is this convolution. The epilogue writes g = dz * act'(z) (mask from the stored z, or re-formed as x * scale + shift for an operand-path layer)
// and the layer's two backward sums (sum g | sum g * xhat, xhat = (x - mean) * invstd) go out through the statistics rows -- bn_bwd_reduce (10 us, three
// tensor reads) and its ordered-sum launch disappear for that layer. x / z rows arrive like a residual: one 16-byte load per lane and pixel.
template <typename T, int TH, int BN, int FM, int FN, int WAVES_N, bool RES, bool BNB = false>
__device__ __forceinline__ void h3_epilogue(const mg_conv_params& p, f32x4 (&acc)[FM][FN], const float (&sc)[8], const float (&sh)[8], char* smem,
                                            const H3Rows<FM, RES, BNB>& pr, int t, int wave, int lane, int n0, int mt, [[maybe_unused]] int work) {
    using TR = ElemTraits<T>;
    constexpr int WN = 16 * FN, WAVES_M = 4 / WAVES_N;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;
    const int c0 = n0 + wn * WN + lg * 4 * FN;
    const bool col_ok = c0 < p.Cout;
    const float sl = p.act == MG_ACT_NONE ? 1.f : (p.act == MG_ACT_RELU ? 0.f : p.slope);
    T* __restrict__ yb = (T*)p.y;
    const bool stats = p.stats != nullptr;
    const int (&mrow)[FM] = pr.mrow;
    H3_STAMP(16);
    // All FM x 8 values go through the epilogue one STEP at a time (source order = issue order): with one wave per SIMD a dependent VALU
    // chain costs ~8 cycles per instruction (measured: 660 cycles per pixel row when the five steps of a value sat back to back), independent
    // neighbours issue every 4. act(v) = max(v, v * slope): none -> 1, ReLU -> 0, LeakyReLU -> slope; before or after the affine part.
    float v[FM][8];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < FN; ++j) v[i][e * FN + j] = acc[i][j][e];
    if (p.pre_act && sl != 1.f) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[i][q] = fmaxf(v[i][q], v[i][q] * sl);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[i][q] = v[i][q] * sc[q] + sh[q];
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float rv[8];
            TR::unpack(pr.q1[i], rv);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[i][q] += rv[q];
        }
    }
    if (!p.pre_act && sl != 1.f) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[i][q] = fmaxf(v[i][q], v[i][q] * sl);
    }
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float rv2[8];
            TR::unpack(pr.q2[i], rv2);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[i][q] += rv2[q];
        }
    }
    [[maybe_unused]] float bxv[FM][8];
    if constexpr (BNB) {
        const bool has_y = p.bnb_y != nullptr;
        const float bsl = p.bnb_act == MG_ACT_NONE ? 1.f : (p.bnb_act == MG_ACT_RELU ? 0.f : p.slope);
        const bool lazy = p.bnb_scale != nullptr;
        float bsc[8], bsh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bsc[e] = 0.f; bsh[e] = 1.f; }
        if (lazy && col_ok) {
            *(float4*)&bsc[0] = *(const float4*)(p.bnb_scale + c0); *(float4*)&bsc[4] = *(const float4*)(p.bnb_scale + c0 + 4);
            *(float4*)&bsh[0] = *(const float4*)(p.bnb_shift + c0); *(float4*)&bsh[4] = *(const float4*)(p.bnb_shift + c0 + 4);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float byv[8];
            TR::unpack(pr.qx[i], bxv[i]);
            TR::unpack(pr.qy[i], byv);
            if (has_y) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[i][q] = byv[q] > 0.f ? v[i][q] : v[i][q] * bsl;
            } else if (lazy) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[i][q] = (bxv[i][q] * bsc[q] + bsh[q]) > 0.f ? v[i][q] : v[i][q] * bsl;
            }
        }
    }
    uint4 packed[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) packed[i] = TR::pack(v[i]);      // rounded once; the statistics are those of the rounded values
#pragma unroll
    for (int i = 0; i < FM; ++i)
        if (mrow[i] >= 0) *(uint4*)(yb + (long)mrow[i] * p.ldy + p.yoff + c0) = packed[i];
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if constexpr (BNB) {
        if (stats) {
            float bmu[8], bis[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { bmu[e] = 0.f; bis[e] = 0.f; }
            if (col_ok) {
                *(float4*)&bmu[0] = *(const float4*)(p.bnb_mean + c0); *(float4*)&bmu[4] = *(const float4*)(p.bnb_mean + c0 + 4);
                *(float4*)&bis[0] = *(const float4*)(p.bnb_invstd + c0); *(float4*)&bis[4] = *(const float4*)(p.bnb_invstd + c0 + 4);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                TR::unpack(packed[i], v[i]);
                const bool keep = mrow[i] >= 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float u = keep ? v[i][q] : 0.f; s1[q] += u; s2[q] += u * (bxv[i][q] - bmu[q]) * bis[q]; }
            }
        }
    } else if (stats) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            TR::unpack(packed[i], v[i]);
            const bool keep = mrow[i] >= 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float u = keep ? v[i][q] : 0.f; s1[q] += u; s2[q] += u * u; }
        }
    }
    H3_STAMP(15);
    if (stats) {
        // per channel: this lane's FM pixels -> the 16 pixel lanes of its DPP row -> the WAVES_M waves of the channel block (LDS, fixed order)
        // -> ONE addition into the tile's row of the statistics buffer (deterministic mode: a row per spatial tile)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = row16_sum(s1[e]); s2[e] = row16_sum(s2[e]); }
        float* sStat = (float*)smem;                         // [WAVES_M][2][BN]
        __syncthreads();                                     // every wave is out of the ring
        if (lr == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sStat[wm * 2 * BN + wn * WN + lg * 8 + e] = s1[e];
                sStat[wm * 2 * BN + BN + wn * WN + lg * 8 + e] = s2[e];
            }
        }
        __syncthreads();
        if (t < 2 * BN) {
            const int c = t < BN ? t : t - BN;
            if (n0 + c < p.Cout) {
                float val = sStat[t];
#pragma unroll
                for (int w_ = 1; w_ < WAVES_M; ++w_) val += sStat[w_ * 2 * BN + t];
                if (p.stat_mode == 1) {                                       // one row, sums only
                    if (t < BN) atomicAdd(&p.stats[n0 + c], val);
                } else {
                    float* st = p.stats + (size_t)((unsigned)mt % (unsigned)(p.stat_rep > 0 ? p.stat_rep : MG_STAT_REPLICAS)) * 2 * p.Cout;
                    atomicAdd(&st[(t < BN ? 0 : p.Cout) + n0 + c], val);
                }
            }
        }
    }
    H3_STAMP(31);
}


// Roles (NS > 1): waves 0-3 are CONSUMERS (fragment reads + MFMA + epilogue), waves 4-7 are PRODUCERS (LDS-DMA only). Measured on the
// single-role form (tools/h3_timeline.py): an LDS-DMA instruction holds the issuing wave ~90 cycles while four waves feed the one vector-memory
// path (36 pieces of a 128 x 32 stage: ~840 cycles per wave), and a wave that waits in the vector-memory issue queue issues no MFMA -- with one
// wave per SIMD a stage cost compute (1220) + issue (840) whatever the ring depth or the placement of the pieces. A producer wave shares its
// SIMD with one consumer wave and blocks alone. One s_barrier per stage joins the roles: behind barrier s every producer has seen its pieces
// of stage s land (counted vmcnt) and every consumer is done reading stage s - 1, whose buffer the producers then refill with stage s + NS - 1.
// NS == 1 (Cin 32 / 64: one or two slabs) stays single-role, 256 threads: up to four workgroups per CU overlap each other instead.
// XF (mg_conv_params.xf_*, forward only): `x` is the RAW output of the producing convolution; the BatchNorm + activation between the two layers is
// applied to the staged halo image IN LDS, once per pixel: behind the counted wait that says "this lane's LDS-DMA pieces of stage s have landed"
// every lane rewrites the 16-byte chunks its own loads deposited -- act(x * scale + shift), rounded to T, the bits mg_affine_act would have
// stored -- and skips the chunks it pointed at the zero page (padding stays 0); the stage's barrier publishes the result. In the split form this
// is PRODUCER work (they idle between issue bursts). Constants: stage 0's in registers (loaded before the first piece), the rest from an LDS
// table behind the ring ([2 * Cin] floats, written before barrier 0), in the single-role forms too.
template <typename T, int TH, int BN, int NS, int MODE, bool RES, bool XF = false, bool BNB = false>
__global__ __launch_bounds__(NS > 1 ? 512 : 256) void conv_halo3_kernel(const mg_conv_params p) {
    using TR = ElemTraits<T>;
    using HC = H3Cfg<TH, BN, NS>;
    constexpr bool SPLIT = NS > 1;
    constexpr int EPS = 32, TW = HC::TW, PW = HC::PW, HH = HC::HH;
    constexpr int WAVES_N = HC::WAVES_N, FM = HC::FM, FN = HC::FN, WN = HC::WN;
    constexpr int STAGE = HC::STAGE, L = HC::L, A_BYTES = HC::A_BYTES, APW = HC::A_PER_WAVE, BPW = HC::B_PER_WAVE;
    constexpr int AD = (MG_H3_AD - 1 <= FM + 2) ? MG_H3_AD : FM + 3;      // fragment-ring depth of the halo rows (first reads stay in column 0)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int H = p.Hout, W = p.Wout;                        // stride 1, pad 1: input and output share the geometry
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntn = (p.Cout + BN - 1) / BN;
    int work;
    if (!xcd_order(p.N * tiles_y * tiles_x * ntn, work)) return;
    H3_STAMP(30);
    H3_STAMP(0);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = (t >> 6) & 3;                           // index inside the role
    const bool producer = SPLIT && (t >> 8) != 0;
    const int mt = work / ntn;
    const int n0 = (work - mt * ntn) * BN;
    const int img = mt / (tiles_y * tiles_x);
    const int trem = mt - img * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int Ktot = 9 * p.Cin;
    const int nstage = p.Cin / EPS;                          // one stage = one 32-channel slab, all nine taps
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    if (!SPLIT || producer) {
        // ---- what this lane fetches: A = halo pixels (A_PER_WAVE instructions per wave and stage), B = weight rows ---------------
        const char* __restrict__ xb = (const char*)p.x;
        const char* __restrict__ wb = (const char*)p.w;
        const char* zpage = (const char*)mg_h3_zero_page;
        const long xpitch = (long)p.ldx * 2l;
        const char* asrc[APW];
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int a = wave + 4 * i;
            const int q = a * 16 + (lane >> 2);                  // pixel slot of the linear [HH][PW] halo image
            const int hy = q / PW, hx = q - hy * PW;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = a < HC::A_INSTR && hy < HH && hx < TW + 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int ach = (lane & 3) ^ (((lane >> 4) & 1) * 2);  // chunk landing in this lane's slot: slot = chunk ^ 2*((q>>2)&1)
            asrc[i] = ok ? xb + ((long)(img * H + iy) * W + ix) * xpitch + ach * 16 : nullptr;
        }
        const char* bsrc[BPW];
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int bi = wave + 4 * i;
            const int tap = bi / HC::B_GRP, grp = bi - tap * HC::B_GRP;
            const int rho = grp * 16 + (lane >> 2);              // LDS row of the tap's [BN][64 B] tile
            // row permutation: fragment j of channel block wn holds, in its row r, output channel r * FN + j of the block -- a lane's
            // accumulator registers (rows 4 lg .. 4 lg + 3 of FN fragments) are then 4 * FN consecutive channels
            const int blk = rho / WN, within = rho - blk * WN;
            const int co = n0 + blk * WN + (within & 15) * FN + (within >> 4);
            const int bch = (lane & 3) ^ (((lane >> 5) & 1) * 3);
            const bool ok = bi < HC::B_INSTR && co < p.Cout;
            bsrc[i] = ok ? wb + ((long)co * Ktot + (long)tap * p.Cin) * 2l + bch * 16 : nullptr;
        }
        // ---- operand transform state (XF) ----
        [[maybe_unused]] const int xf_ach = (lane & 3) ^ (((lane >> 4) & 1) * 2);     // the 8-channel group of a slab this lane's halo chunks hold
        [[maybe_unused]] const float xf_sl = xf_slope_of(p.xf_act, p.xf_slope);
        [[maybe_unused]] const unsigned xf_tab = lds_base + (unsigned)(NS * STAGE);   // [Cin] scale | [Cin] shift, fp32 (split form)
        [[maybe_unused]] float xr_sc[1][8], xr_sh[1][8];
        [[maybe_unused]] u32x4 xf_reg = (u32x4){0u, 0u, 0u, 0u};
        if constexpr (XF) {
#pragma unroll
            for (int q = 0; q < 1; ++q) {                        // stage 0's constants in registers; later stages read the LDS table
                const int c0 = q * EPS + xf_ach * 8;
                *(float4*)&xr_sc[q][0] = *(const float4*)(p.xf_scale + c0); *(float4*)&xr_sc[q][4] = *(const float4*)(p.xf_scale + c0 + 4);
                *(float4*)&xr_sh[q][0] = *(const float4*)(p.xf_shift + c0); *(float4*)&xr_sh[q][4] = *(const float4*)(p.xf_shift + c0 + 4);
            }
            {
                const int tt = SPLIT ? t - 256 : t;              // (producer) thread index: 4 floats of the table each
                if (tt * 4 < 2 * p.Cin) xf_reg = *(const u32x4*)(tt * 4 < p.Cin ? p.xf_scale + tt * 4 : p.xf_shift + (tt * 4 - p.Cin));
            }
        }
        // this lane's chunks of the stage in ring buffer `buf`, in place; `reg` >= 0: constants from the registers of slab `reg`, else from the table
        [[maybe_unused]] auto xf_stage = [&](int s, int buf, int reg) {
            const unsigned sb = lds_base + (unsigned)(buf * STAGE);
            float scv[8], shv[8];
            if (reg >= 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { scv[e] = xr_sc[0][e]; shv[e] = xr_sh[0][e]; }
            } else {
                const unsigned tsc = xf_tab + (unsigned)((s * EPS + xf_ach * 8) * 4), tsh = tsc + (unsigned)(p.Cin * 4);
                f32x4 c0, c1, h0, h1;
                asm volatile("ds_read_b128 %0, %1" : "=v"(c0) : "v"(tsc) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(c1) : "v"(tsc) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(h0) : "v"(tsh) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(h1) : "v"(tsh) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int e = 0; e < 4; ++e) { scv[e] = c0[e]; scv[4 + e] = c1[e]; shv[e] = h0[e]; shv[4 + e] = h1[e]; }
            }
            u32x4 q[APW];
#pragma unroll
            for (int i = 0; i < APW; ++i)
                asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // all chunks through the transform one STEP at a time (independent chains side by side: a dependent VALU chain costs ~8 cycles per
            // instruction with the one or two waves a SIMD has here); an in-image pixel is transformed, zero-page chunks (padding, spare slots) keep
            // their zeros (a transformed zero would be act(shift))
            float f[APW][8];
#pragma unroll
            for (int i = 0; i < APW; ++i) ElemTraits<T>::unpack(__builtin_bit_cast(uint4, q[i]), f[i]);
#pragma unroll
            for (int i = 0; i < APW; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) f[i][e] = f[i][e] * scv[e] + shv[e];
#pragma unroll
            for (int i = 0; i < APW; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) f[i][e] = fmaxf(f[i][e], f[i][e] * xf_sl);
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const uint4 r = ElemTraits<T>::pack(f[i]);
                const u32x4 w_ = asrc[i] ? __builtin_bit_cast(u32x4, r) : q[i];
                asm volatile("ds_write_b128 %0, %1" ::"v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)), "v"(w_) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        // all LDS-DMA instructions (1 KiB each: 64 lanes x 16 B) of stage s into ring buffer `buf`: halo pieces, then weight pieces
        auto issue_stage = [&](int s, int buf) {
            char* sbase = smem + buf * STAGE;
            const long coff = (long)s * EPS * 2l;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const char* g = asrc[i] ? asrc[i] + coff : zpage;
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(sbase + (wave + 4 * i) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const char* g = bsrc[i] ? bsrc[i] + coff : zpage;
                // instruction bi = (tap, 16-row group): its 1 KiB lands at tap * BN * 64 + group * 1024 = bi * 1024 (B_GRP groups per tap)
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(sbase + A_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
            }
        };
        if constexpr (SPLIT) {
            // Stage 0 alone goes out before the first barrier (the consumers' first MFMA waits for nothing else: issuing the whole ring first
            // put 27 pieces per wave -- ~3 k cycles of issue -- in front of it); the rest of the ring follows behind barrier 0.
            issue_stage(0, 0);
            wait_vm<0>();
            if constexpr (XF) {
                const int tt = t - 256;
                if (tt * 4 < 2 * p.Cin) asm volatile("ds_write_b128 %0, %1" ::"v"(xf_tab + (unsigned)tt * 16u), "v"(xf_reg) : "memory");
                xf_stage(0, 0, 0);                           // (ends in lgkmcnt(0): the table words are written, too)
            }
            __builtin_amdgcn_s_barrier();                    // barrier 0 (publishes the table to the other producer waves as well)
#pragma unroll
            for (int u = 1; u < NS; ++u)
                if (u < nstage) issue_stage(u, u);
            int fbuf = 0;                                    // ring buffer of stage s + NS - 1 = buffer of stage s - 1
            for (int s = 1; s < nstage; ++s) {
                const int infl = min(NS - 1, nstage - s) - 1;    // stages behind s already issued: s + 1 .. min(s + NS - 2, nstage - 1)
                if (infl <= 0) wait_vm<0>();
                else if (infl == 1) wait_vm<L>();
                else wait_vm<(NS > 3 ? 2 * L : 0)>();
                if constexpr (XF) xf_stage(s, fbuf + 1 == NS ? 0 : fbuf + 1, -1);   // stage s sits in the buffer behind stage s - 1's
                __builtin_amdgcn_s_barrier();                // barrier s: stage s is in LDS for everybody; stage s - 1 has been consumed
#if MG_H3_EXP != 1
                if (s + NS - 1 < nstage) issue_stage(s + NS - 1, fbuf);
#endif
                fbuf = fbuf + 1 == NS ? 0 : fbuf + 1;
            }
            if (p.stats) { __syncthreads(); __syncthreads(); }   // the two barriers of the consumers' statistics tail
            __syncthreads();                                     // the producers stay until the consumers are done (see the end of the kernel)
            return;
        } else {
            issue_stage(0, 0);
            // (single-role form continues below as its own consumer; later stages are issued between the barriers of the loop)
            (void)issue_stage;
        }
        if constexpr (!SPLIT) {
            // single-role K loop needs issue_stage in scope: run the whole consumer body here
            H3_STAMP(1);
            const int wm = wave / WAVES_N, wn = wave % WAVES_N;
            const int lr = lane & 15, lg = lane >> 4;
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
            float sc[8], sh[8];
            const int c0 = n0 + wn * WN + lg * 4 * FN;
            h3_load_affine(p, c0, sc, sh);
            H3Rows<FM, RES, BNB> pr;
            h3_prefetch_rows<T, TH, BN, FM, FN, WAVES_N, RES, BNB>(p, pr, wave, lane, img, y0, x0, n0);
            H3Frags<FN, AD> fr;
            for (int s = 0; s < nstage; ++s) {
                if (s < 6) H3_STAMP(2 + 2 * s);
                if (s > 0) {
                    __builtin_amdgcn_s_barrier();
                    issue_stage(s, 0);
                }
                wait_vm<0>();
                if constexpr (XF) {
                    if (s == 0 && t * 4 < 2 * p.Cin) asm volatile("ds_write_b128 %0, %1" ::"v"(xf_tab + (unsigned)t * 16u), "v"(xf_reg) : "memory");
                    xf_stage(s, 0, s == 0 ? 0 : -1);             // (ends in lgkmcnt(0); the barrier below publishes the table with stage 0)
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (s < 6) H3_STAMP(3 + 2 * s);
                h3_first_reads<FM, FN, BN, PW, MODE, AD, 0>(lds_base, a_lane, b_lane, fr);
                h3_walk<T, FM, FN, BN, PW, MODE, AD, 0>(lds_base, a_lane, b_lane, acc, fr, [] {});
            }
            H3_STAMP(14);
            h3_epilogue<T, TH, BN, FM, FN, WAVES_N, RES, BNB>(p, acc, sc, sh, smem, pr, t, wave, lane, n0, mt, work);
            return;
        }
    }
    if constexpr (SPLIT) {
        // ---------------- consumer waves ----------------
        H3_STAMP(1);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int lr = lane & 15, lg = lane >> 4;
        // pixel fragment i of this wave = tile row ty = wm * FM + i, pixel tx = lr; tap (ky, kx) reads halo pixel (ty + ky, lr + kx)
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
        // epilogue operands that do not depend on the accumulators: requested now, in flight under the whole K loop
        float sc[8], sh[8];
        const int c0 = n0 + wn * WN + lg * 4 * FN;               // this lane's 8 consecutive output channels
        h3_load_affine(p, c0, sc, sh);
        H3Rows<FM, RES, BNB> pr;
        h3_prefetch_rows<T, TH, BN, FM, FN, WAVES_N, RES, BNB>(p, pr, wave, lane, img, y0, x0, n0);
        H3Frags<FN, AD> fr;
        H3_STAMP(2);
        __builtin_amdgcn_s_barrier();                        // barrier 0: stage 0 is in LDS
        asm volatile("" ::: "memory");
        H3_STAMP(3);
        h3_first_reads<FM, FN, BN, PW, MODE, AD, 0>(lds_base, a_lane, b_lane, fr);
        int buf = 0;
        // two stages per trip: the fragment register sets alternate (PAR 0 | 1); barrier s + 1 and stage s + 1's first reads sit inside stage s's walk
        for (int s = 0; s < nstage; s += 2) {
            {
                const unsigned sb = lds_base + (unsigned)(buf * STAGE);
                buf = buf + 1 == NS ? 0 : buf + 1;
                const unsigned sbn = lds_base + (unsigned)(buf * STAGE);
                const bool nxt = s + 1 < nstage;
                h3_walk<T, FM, FN, BN, PW, MODE, AD, 0>(sb, a_lane, b_lane, acc, fr, [&] {
                    if (nxt) {
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        h3_first_reads<FM, FN, BN, PW, MODE, AD, 1>(sbn, a_lane, b_lane, fr);
                    }
                });
                if (s + 1 < 3) H3_STAMP(2 + 2 * (s + 1));
            }
            if (s + 1 < nstage) {
                const unsigned sb = lds_base + (unsigned)(buf * STAGE);
                buf = buf + 1 == NS ? 0 : buf + 1;
                const unsigned sbn = lds_base + (unsigned)(buf * STAGE);
                const bool nxt = s + 2 < nstage;
                h3_walk<T, FM, FN, BN, PW, MODE, AD, 1>(sb, a_lane, b_lane, acc, fr, [&] {
                    if (nxt) {
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        h3_first_reads<FM, FN, BN, PW, MODE, AD, 0>(sbn, a_lane, b_lane, fr);
                    }
                });
                if (s + 2 < 3) H3_STAMP(2 + 2 * (s + 2));
            }
        }
        H3_STAMP(14);
        h3_epilogue<T, TH, BN, FM, FN, WAVES_N, RES, BNB>(p, acc, sc, sh, smem, pr, t, wave, lane, n0, mt, work);
        // (no wave of the workgroup ends before the others: the producers wait at the same barrier)
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent producer / consumer form ("persistent workgroups that issue tile n + 1's first stages under tile
// n's epilogue"). One 512-thread workgroup per CU walks a LIST of tiles (virtual block ids blockIdx.x + k * gridDim.x through the XCD order);
// the ring of stages runs ACROSS tile boundaries: the producers are up to NS - 1 stages ahead whatever tile those stages belong to, so from the
// second tile on a tile's first stage has landed before the consumers have finished the previous tile's epilogue (the cold start -- 3-5 k of
// a 13 k-cycle launch -- is paid once per workgroup), and (with MG_H3_EARLY) the consumers' early barrier puts the next tile's first fragment reads under the last
// MFMAs of this one. The statistics tail of tile k (two workgroup barriers) comes BEHIND the barrier of tile k + 1's first stage in both roles.
// For layers with >= ~2 tiles per CU (batch 12, the video shapes); plain forward / data gradient (no operand transform, no BatchNorm link).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool h3_work_of(int v, int L, int& work) {
    const int chunk = (L + NXCD - 1) / NXCD;
    work = (v % NXCD) * chunk + v / NXCD;
    return v / NXCD < chunk && work < L;
}

template <typename T, int TH, int BN, int NS, int MODE, bool RES>
__global__ __launch_bounds__(512) void conv_halo3_persist_kernel(const mg_conv_params p) {
    using HC = H3Cfg<TH, BN, NS>;
    static_assert(NS >= 3, "the persistent form runs a ring");
    constexpr int EPS = 32, TW = HC::TW, PW = HC::PW, HH = HC::HH;
    constexpr int WAVES_N = HC::WAVES_N, FM = HC::FM, FN = HC::FN, WN = HC::WN;
    constexpr int STAGE = HC::STAGE, L = HC::L, A_BYTES = HC::A_BYTES, APW = HC::A_PER_WAVE, BPW = HC::B_PER_WAVE;
    constexpr int AD = (MG_H3_AD - 1 <= FM + 2) ? MG_H3_AD : FM + 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stat_lds = smem + NS * STAGE;                      // the ring stays live across tiles: the statistics tail has its own words

    const int H = p.Hout, W = p.Wout;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int total = p.N * tiles_y * tiles_x * ntn;
    const int G = (int)gridDim.x;
    int nt = 0;                                              // tiles of this workgroup
    { int w_; while (h3_work_of((int)blockIdx.x + nt * G, total, w_)) ++nt; }
    if (nt == 0) return;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = (t >> 6) & 3;
    const bool producer = (t >> 8) != 0;
    const int Ktot = 9 * p.Cin;
    const int nstage = p.Cin / EPS;
    const int Gt = nt * nstage;                              // stages of this workgroup, all tiles
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    [[maybe_unused]] int work = 0;

    if (producer) {
        const char* __restrict__ xb = (const char*)p.x;
        const char* __restrict__ wb = (const char*)p.w;
        const char* zpage = (const char*)mg_h3_zero_page;
        const long xpitch = (long)p.ldx * 2l;
        const char* asrc[APW];
        const char* bsrc[BPW];
        auto setup_tile = [&](int k) {                       // addresses of tile k of this workgroup's list
            int w_;
            h3_work_of((int)blockIdx.x + k * G, total, w_);
            const int mt = w_ / ntn, n0 = (w_ - mt * ntn) * BN;
            const int img = mt / (tiles_y * tiles_x), trem = mt - img * tiles_y * tiles_x;
            const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int a = wave + 4 * i;
                const int q = a * 16 + (lane >> 2);
                const int hy = q / PW, hx = q - hy * PW;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                const bool ok = a < HC::A_INSTR && hy < HH && hx < TW + 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int ach = (lane & 3) ^ (((lane >> 4) & 1) * 2);
                asrc[i] = ok ? xb + ((long)(img * H + iy) * W + ix) * xpitch + ach * 16 : nullptr;
            }
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const int bi = wave + 4 * i;
                const int tap = bi / HC::B_GRP, grp = bi - tap * HC::B_GRP;
                const int rho = grp * 16 + (lane >> 2);
                const int blk = rho / WN, within = rho - blk * WN;
                const int co = n0 + blk * WN + (within & 15) * FN + (within >> 4);
                const int bch = (lane & 3) ^ (((lane >> 5) & 1) * 3);
                const bool ok = bi < HC::B_INSTR && co < p.Cout;
                bsrc[i] = ok ? wb + ((long)co * Ktot + (long)tap * p.Cin) * 2l + bch * 16 : nullptr;
            }
        };
        int gi = 0, ki = 0, si = 0, ibuf = 0;                // next stage to issue: global index, its tile, its slab, its ring buffer
        setup_tile(0);
        auto issue_next = [&]() {
            char* sbase = smem + ibuf * STAGE;
            const long coff = (long)si * EPS * 2l;
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const char* g = asrc[i] ? asrc[i] + coff : zpage;
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(sbase + (wave + 4 * i) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const char* g = bsrc[i] ? bsrc[i] + coff : zpage;
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(sbase + A_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
            }
            ++gi;
            ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
            if (++si == nstage) { si = 0; if (++ki < nt) setup_tile(ki); }
        };
        for (int u = 0; u < NS - 1 && gi < Gt; ++u) issue_next();
        int stile = 0;                                       // stage index inside its tile of global stage g
        for (int g = 0; g < Gt; ++g) {
            const int infl = gi - 1 - g;                     // stage groups issued behind stage g
            if (infl <= 0) wait_vm<0>();
            else if (infl == 1) wait_vm<L>();
            else wait_vm<(NS > 3 ? 2 * L : 0)>();
            __builtin_amdgcn_s_barrier();                    // barrier g: stage g is in LDS for everybody; stage g - 1 has been consumed
            if (g > 0 && stile == 0 && p.stats) { __syncthreads(); __syncthreads(); }     // the previous tile's statistics tail
            if (gi < Gt) issue_next();
            if (++stile == nstage) stile = 0;
        }
        if (p.stats) { __syncthreads(); __syncthreads(); }   // the last tile's
        __syncthreads();                                     // stay until the consumers are done
        return;
    }

    // ---------------- consumer waves ----------------
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;
    unsigned a_lane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = lr + kx;
        a_lane[kx] = (unsigned)((wm * FM * PW + px) * 64 + ((lg ^ (((px >> 2) & 1) * 2)) << 4));
    }
    const unsigned b_lane = (unsigned)(A_BYTES + (wn * WN + lr) * 64 + ((lg ^ (((lr >> 3) & 1) * 3)) << 4));
    H3Frags<FN, AD> fr;
    __builtin_amdgcn_s_barrier();                            // barrier 0
    asm volatile("" ::: "memory");
    h3_first_reads<FM, FN, BN, PW, MODE, AD, 0>(lds_base, a_lane, b_lane, fr);
    int g = 0, buf = 0;
    for (int k = 0; k < nt; ++k) {
        h3_work_of((int)blockIdx.x + k * G, total, work);
        const int mt = work / ntn, n0 = (work - mt * ntn) * BN;
        const int img = mt / (tiles_y * tiles_x), trem = mt - img * tiles_y * tiles_x;
        const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float sc[8], sh[8];
        h3_load_affine(p, n0 + wn * WN + lg * 4 * FN, sc, sh);
        H3Rows<FM, RES, false> pr;
        h3_prefetch_rows<T, TH, BN, FM, FN, WAVES_N, RES, false>(p, pr, wave, lane, img, y0, x0, n0);
        // two stages per trip (nstage is even for this form: the fragment register sets alternate at compile-time positions)
        for (int s = 0; s < nstage; s += 2, g += 2) {
            {
                const unsigned sb = lds_base + (unsigned)(buf * STAGE);
                buf = buf + 1 == NS ? 0 : buf + 1;
                const unsigned sbn = lds_base + (unsigned)(buf * STAGE);
                h3_walk<T, FM, FN, BN, PW, MODE, AD, 0>(sb, a_lane, b_lane, acc, fr, [&] {
                    __builtin_amdgcn_s_barrier();            // (a second stage of the pair always exists)
                    asm volatile("" ::: "memory");
                    h3_first_reads<FM, FN, BN, PW, MODE, AD, 1>(sbn, a_lane, b_lane, fr);
                });
            }
            {
                const unsigned sb = lds_base + (unsigned)(buf * STAGE);
                buf = buf + 1 == NS ? 0 : buf + 1;
                const unsigned sbn = lds_base + (unsigned)(buf * STAGE);
                const bool nxt = g + 2 < Gt;
                h3_walk<T, FM, FN, BN, PW, MODE, AD, 1>(sb, a_lane, b_lane, acc, fr, [&] {
                    if (nxt) {
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        h3_first_reads<FM, FN, BN, PW, MODE, AD, 0>(sbn, a_lane, b_lane, fr);
                    }
                });
            }
        }
        // (the consumers are behind the barrier of the next tile's first stage here: its first fragments are in flight under this epilogue)
        h3_epilogue<T, TH, BN, FM, FN, WAVES_N, RES, false>(p, acc, sc, sh, stat_lds, pr, t, wave, lane, n0, mt, work);
    }
    __syncthreads();                                         // (the producers wait here)
}


// ---------------------------------------------------------------------------------------------------------------------
// One-slab persistent form (Cin == 32, Cout <= 32: the OS1 / OS2 layers of the decoder and of the fine shortcut branches -- 512 x 512 and 256 x 256
// maps at the headline geometry, 2 048 - 8 192 tiles per layer). In the per-tile forms above such a layer pays, PER 8 x 16-pixel tile: a workgroup start,
// 18 KiB of weights through the LDS-DMA path next to 15 KiB of pixels (that path delivers ~50 B per clock and CU: tools/h3_timeline.py), and the
// latency chain address set-up -> first piece landed -> first MFMA with nothing of its own to hide it. Here a 256-thread workgroup (three per CU: 52 KiB)
// stages the layer's weights ONCE and walks a list of tiles with the halo image double-buffered: tile k + 1's pieces are issued before tile k's tap walk,
// and its operand transform (XF: the producing layer's BatchNorm + activation, constants of the one slab in registers) runs between the walk and the
// epilogue of tile k. One workgroup barrier per tile (+ the two of the statistics tail).
// LDS: [A0 16 K][B 20 K][A1 16 K]; the weight image uses 18 of its 20 KiB, the statistics words live in the rest. The tap walk addresses the weights
// relative to the halo image of ITS buffer: buffer 1 passes b_lane - (A1 - A0).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int MODE, bool RES, bool XF>
__global__ __launch_bounds__(256) void conv_halo3_slab_kernel(const mg_conv_params p) {
    constexpr int TH = 8, BN = 32;
    using HC = H3Cfg<TH, BN, 1>;
    constexpr int TW = HC::TW, PW = HC::PW, HH = HC::HH;
    constexpr int WAVES_N = HC::WAVES_N, FM = HC::FM, FN = HC::FN, WN = HC::WN;
    constexpr int A_BYTES = HC::A_BYTES, B_BYTES = HC::B_BYTES, APW = HC::A_PER_WAVE, BPW = HC::B_PER_WAVE;
    constexpr int AD = (MG_H3_AD - 1 <= FM + 2) ? MG_H3_AD : FM + 3;
    constexpr unsigned A1_OFF = (unsigned)(A_BYTES + B_BYTES);
    static_assert(HC::B_INSTR * 1024 + HC::STAT_BYTES <= B_BYTES, "the statistics words sit behind the weight image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stat_lds = smem + A_BYTES + HC::B_INSTR * 1024;

    const int H = p.Hout, W = p.Wout;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int total = p.N * tiles_y * tiles_x;               // (one channel tile: Cout <= BN)
    const int G = (int)gridDim.x;
    int nt = 0;
    { int w_; while (h3_work_of((int)blockIdx.x + nt * G, total, w_)) ++nt; }
    if (nt == 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* __restrict__ xb = (const char*)p.x;
    const char* __restrict__ wb = (const char*)p.w;
    const char* zpage = (const char*)mg_h3_zero_page;
    const long xpitch = (long)p.ldx * 2l;

    // ---- the weights, once -------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int bi = wave + 4 * i;
        if (bi < HC::B_INSTR) {                              // (wave-uniform)
            const int tap = bi / HC::B_GRP, grp = bi - tap * HC::B_GRP;
            const int rho = grp * 16 + (lane >> 2);
            const int blk = rho / WN, within = rho - blk * WN;
            const int co = blk * WN + (within & 15) * FN + (within >> 4);
            const int bch = (lane & 3) ^ (((lane >> 5) & 1) * 3);
            const char* g = co < p.Cout ? wb + ((long)co * (9 * 32) + (long)tap * 32) * 2l + bch * 16 : zpage;
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(smem + A_BYTES + bi * 1024), 16, 0, 0);
        }
    }
    // ---- halo pieces of a tile: the slot geometry of this lane is the same for every tile ------------------------------------
    int hy_[APW], hx_[APW];
    bool slot_ok[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int a = wave + 4 * i;
        const int q = a * 16 + (lane >> 2);
        hy_[i] = q / PW; hx_[i] = q - hy_[i] * PW;
        slot_ok[i] = a < HC::A_INSTR && hy_[i] < HH && hx_[i] < TW + 2;
    }
    const int ach = (lane & 3) ^ (((lane >> 4) & 1) * 2);
    auto tile_of = [&](int k, int& mt, int& img, int& y0, int& x0) {
        int w_;
        h3_work_of((int)blockIdx.x + k * G, total, w_);
        mt = w_;
        img = mt / (tiles_y * tiles_x);
        const int trem = mt - img * tiles_y * tiles_x;
        y0 = (trem / tiles_x) * TH; x0 = (trem % tiles_x) * TW;
    };
    // issues tile (img, y0, x0) into the halo buffer at `abase`; returns which of this lane's chunks are in-image pixels (the operand transform skips padding)
    auto issue_tile = [&](int img, int y0, int x0, char* abase) -> unsigned {
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int iy = y0 - 1 + hy_[i], ix = x0 - 1 + hx_[i];
            const bool ok = slot_ok[i] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const char* g = ok ? xb + ((long)(img * H + iy) * W + ix) * xpitch + ach * 16 : zpage;
            m |= ok ? 1u << i : 0u;
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)(abase + (wave + 4 * i) * 1024), 16, 0, 0);
        }
        return m;
    };
    [[maybe_unused]] float xsc[8], xsh[8];
    [[maybe_unused]] const float xf_sl = xf_slope_of(p.xf_act, p.xf_slope);
    if constexpr (XF) {
        *(float4*)&xsc[0] = *(const float4*)(p.xf_scale + ach * 8); *(float4*)&xsc[4] = *(const float4*)(p.xf_scale + ach * 8 + 4);
        *(float4*)&xsh[0] = *(const float4*)(p.xf_shift + ach * 8); *(float4*)&xsh[4] = *(const float4*)(p.xf_shift + ach * 8 + 4);
    }
    [[maybe_unused]] auto xf_buf = [&](unsigned sb, unsigned m) {   // this lane's landed chunks of the halo image at `sb`, in place (see xf_stage above)
        u32x4 q[APW];
#pragma unroll
        for (int i = 0; i < APW; ++i)
            asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float f[APW][8];
#pragma unroll
        for (int i = 0; i < APW; ++i) ElemTraits<T>::unpack(__builtin_bit_cast(uint4, q[i]), f[i]);
#pragma unroll
        for (int i = 0; i < APW; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) f[i][e] = f[i][e] * xsc[e] + xsh[e];
#pragma unroll
        for (int i = 0; i < APW; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) f[i][e] = fmaxf(f[i][e], f[i][e] * xf_sl);
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const uint4 r = ElemTraits<T>::pack(f[i]);
            const u32x4 w_ = ((m >> i) & 1u) ? __builtin_bit_cast(u32x4, r) : q[i];
            asm volatile("ds_write_b128 %0, %1" ::"v"(sb + (unsigned)((wave + 4 * i) * 1024 + lane * 16)), "v"(w_) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    int mt, img, y0, x0;
    tile_of(0, mt, img, y0, x0);
    [[maybe_unused]] const unsigned amask0 = issue_tile(img, y0, x0, smem);
    int mtn = 0, imgn = 0, y0n = 0, x0n = 0;                 // tile k + 1
    [[maybe_unused]] unsigned amask_n = 0;
    if (nt > 1) {
        tile_of(1, mtn, imgn, y0n, x0n);
        amask_n = issue_tile(imgn, y0n, x0n, smem + A1_OFF);
    }
    // ---- consumer geometry ---------------------------------------------------------------------------------------------------
    const int wm = wave / WAVES_N;
    const int lr = lane & 15, lg = lane >> 4;
    unsigned a_lane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = lr + kx;
        a_lane[kx] = (unsigned)((wm * FM * PW + px) * 64 + ((lg ^ (((px >> 2) & 1) * 2)) << 4));
    }
    const unsigned b_lane0 = (unsigned)(A_BYTES + lr * 64 + ((lg ^ (((lr >> 3) & 1) * 3)) << 4));
    float sc[8], sh[8];
    h3_load_affine(p, lg * 4 * FN, sc, sh);
    wait_vm<0>();
    if constexpr (XF) {
        xf_buf(lds_base, amask0);
        if (nt > 1) xf_buf(lds_base + A1_OFF, amask_n);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // Tile k: walk from buffer k & 1 | everything outstanding has landed (tile k + 1's pieces were issued a whole epilogue + walk ago; only loads and
    // stores of the SAME age class are ever waited for together: vmcnt(0), no counted wait over mixed loads and stores) | transform tile k + 1 in its
    // buffer | ONE barrier: tile k + 1 is published, buffer k & 1 is free | tile k + 2's pieces into it | epilogue of tile k.
    for (int k = 0; k < nt; ++k) {
        const int cur = k & 1;
        const unsigned sb = lds_base + (cur ? A1_OFF : 0u);
        const unsigned b_lane = cur ? b_lane0 - A1_OFF : b_lane0;
        H3Rows<FM, RES, false> pr;
        h3_prefetch_rows<T, TH, BN, FM, FN, WAVES_N, RES, false>(p, pr, wave, lane, img, y0, x0, 0);
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        H3Frags<FN, AD> fr;
        h3_first_reads<FM, FN, BN, PW, MODE, AD, 0>(sb, a_lane, b_lane, fr);
        h3_walk<T, FM, FN, BN, PW, MODE, AD, 0>(sb, a_lane, b_lane, acc, fr, [] {});
        wait_vm<0>();
        if constexpr (XF) {
            if (k >= 1 && k + 1 < nt) xf_buf(lds_base + (cur ? 0u : A1_OFF), amask_n);      // (tile 1 went through it in the prologue)
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int mt2 = 0, img2 = 0, y02 = 0, x02 = 0;
        unsigned amask_2 = 0;
        if (k + 2 < nt) {
            tile_of(k + 2, mt2, img2, y02, x02);
            amask_2 = issue_tile(img2, y02, x02, smem + (cur ? A1_OFF : 0u));
        }
        h3_epilogue<T, TH, BN, FM, FN, WAVES_N, RES, false>(p, acc, sc, sh, stat_lds, pr, t, wave, lane, 0, mt, mt);
        mt = mtn; img = imgn; y0 = y0n; x0 = x0n;
        mtn = mt2; imgn = img2; y0n = y02; x0n = x02; amask_n = amask_2;
    }
}

template <typename T>
int launch_h3_slab(const mg_conv_params& p, hipStream_t st) {
    using HC = H3Cfg<8, 32, 1>;
    constexpr size_t lds = (size_t)2 * HC::A_BYTES + HC::B_BYTES;
    static_assert(3 * lds <= 160 * 1024, "three workgroups per CU");
    const bool res = p.res || p.res2;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const long tiles = (long)p.N * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
    if (mg_det_on && p.stats && p.stat_mode == 0 && (long)(p.stat_rep > 0 ? p.stat_rep : MG_STAT_REPLICAS) < tiles) return -8;
    const long maxwg = 3l * ncu, per = (tiles + maxwg - 1) / maxwg;
    dim3 grid(xcd_grid((tiles + per - 1) / per));
#define H3_SLAB(MODE_, RES_, XF_) hipLaunchKernelGGL((conv_halo3_slab_kernel<T, MODE_, RES_, XF_>), grid, dim3(256), lds, st, p)
    if (p.xf_scale) {
        if (p.mode != MG_MODE_CONV) return MG_XF_UNSUPPORTED;
        if (res) H3_SLAB(MG_MODE_CONV, true, true); else H3_SLAB(MG_MODE_CONV, false, true);
    } else if (p.mode == MG_MODE_CONV) {
        if (res) H3_SLAB(MG_MODE_CONV, true, false); else H3_SLAB(MG_MODE_CONV, false, false);
    } else {
        if (res) H3_SLAB(MG_MODE_TCONV, true, false); else H3_SLAB(MG_MODE_TCONV, false, false);
    }
#undef H3_SLAB
    MG_CHECK_LAUNCH();
    return 0;
}

int g_h3_enabled = -1;
int g_h3_force[3] = {0, 0, 0};          // MG_H3_CFG=TH,BN,NS: one tile form for every eligible layer (experiments)

template <typename T, int TH, int BN, int NS>
int launch_h3(const mg_conv_params& p, hipStream_t st) {
    using HC = H3Cfg<TH, BN, NS>;
    constexpr size_t lds = HC::LDS;
    const bool res = p.res || p.res2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<T, TH, BN, NS, MG_MODE_TCONV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<T, TH, BN, NS, MG_MODE_TCONV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long mtiles = (long)p.N * ((p.Hout + TH - 1) / TH) * ((p.Wout + 15) / 16);
    const long tiles = mtiles * ((p.Cout + BN - 1) / BN);
    if (mg_det_on && p.stats && p.stat_mode == 0 && (long)(p.stat_rep > 0 ? p.stat_rep : MG_STAT_REPLICAS) < mtiles) return -8;
    dim3 grid(xcd_grid(tiles));
    if (p.xf_scale) {                                        // BatchNorm + activation of the producing layer applied to the staged halo (forward only)
        constexpr size_t lds_xf = lds + 4096;                    // + the [2 * Cin] fp32 table (Cin <= 512)
        if (p.mode != MG_MODE_CONV || p.Cin > 512 || lds_xf > 160 * 1024) return MG_XF_UNSUPPORTED;
        static bool xf_attr = false;
        if (!xf_attr) {
            (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xf);
            (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xf);
            xf_attr = true;
        }
        if (res) hipLaunchKernelGGL((conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, true, true>), grid, dim3(NS > 1 ? 512 : 256), lds_xf, st, p);
        else hipLaunchKernelGGL((conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, false, true>), grid, dim3(NS > 1 ? 512 : 256), lds_xf, st, p);
        MG_CHECK_LAUNCH();
        return 0;
    }
    if (p.bnb_x) {                                           // the data gradient of a 3x3 / stride 1 conv behind a training BatchNorm layer (BnLink)
        if (p.mode != MG_MODE_TCONV || !p.stats || p.stat_mode != 0) return -2;
        static bool bnb_attr = false;
        if (!bnb_attr) {
            (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<T, TH, BN, NS, MG_MODE_TCONV, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            bnb_attr = true;
        }
        hipLaunchKernelGGL((conv_halo3_kernel<T, TH, BN, NS, MG_MODE_TCONV, true, false, true>), grid, dim3(NS > 1 ? 512 : 256), lds, st, p);
        MG_CHECK_LAUNCH();
        return 0;
    }
    if (p.mode == MG_MODE_CONV) {
        if (res) hipLaunchKernelGGL((conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, true>), grid, dim3(NS > 1 ? 512 : 256), lds, st, p);
        else hipLaunchKernelGGL((conv_halo3_kernel<T, TH, BN, NS, MG_MODE_CONV, false>), grid, dim3(NS > 1 ? 512 : 256), lds, st, p);
    } else {
        if (res) hipLaunchKernelGGL((conv_halo3_kernel<T, TH, BN, NS, MG_MODE_TCONV, true>), grid, dim3(NS > 1 ? 512 : 256), lds, st, p);
        else hipLaunchKernelGGL((conv_halo3_kernel<T, TH, BN, NS, MG_MODE_TCONV, false>), grid, dim3(NS > 1 ? 512 : 256), lds, st, p);
    }
    MG_CHECK_LAUNCH();
    return 0;
}

template <typename T, int TH, int BN, int NS>
int launch_h3_persist(const mg_conv_params& p, hipStream_t st) {
    using HC = H3Cfg<TH, BN, NS>;
    constexpr size_t lds = (size_t)NS * HC::STAGE + HC::STAT_BYTES;
    static_assert(lds <= 160 * 1024, "LDS of the persistent form");
    const bool res = p.res || p.res2;
    static bool attr_set = false;
    static int ncu = 256;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_CONV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_CONV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_TCONV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_TCONV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        attr_set = true;
    }
    const long mtiles = (long)p.N * ((p.Hout + TH - 1) / TH) * ((p.Wout + 15) / 16);
    const long tiles = mtiles * ((p.Cout + BN - 1) / BN);
    if (mg_det_on && p.stats && p.stat_mode == 0 && (long)(p.stat_rep > 0 ? p.stat_rep : MG_STAT_REPLICAS) < mtiles) return -8;
    long g = tiles < ncu ? tiles : ncu;
    dim3 grid(xcd_grid(g));
    if (p.mode == MG_MODE_CONV) {
        if (res) hipLaunchKernelGGL((conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_CONV, true>), grid, dim3(512), lds, st, p);
        else hipLaunchKernelGGL((conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_CONV, false>), grid, dim3(512), lds, st, p);
    } else {
        if (res) hipLaunchKernelGGL((conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_TCONV, true>), grid, dim3(512), lds, st, p);
        else hipLaunchKernelGGL((conv_halo3_persist_kernel<T, TH, BN, NS, MG_MODE_TCONV, false>), grid, dim3(512), lds, st, p);
    }
    MG_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int dispatch_h3(const mg_conv_params& p, hipStream_t st) {
    const int nstage = p.Cin / 32;
    const long sp8 = (long)p.N * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
    int th = 8, bn = 64, ns = 3;
    if (g_h3_force[0]) { th = g_h3_force[0]; bn = g_h3_force[1]; ns = g_h3_force[2]; }
    else {
        // Measured forms (tools/h3_check.py, batch 4 / 12 of the trunk's layers):
        //   * one or two slabs (Cin 32 / 64), or >= ~1.3 tiles per CU: single-role workgroups, no ring -- three or four of them share a CU and
        //     overlap each other's load / MFMA / epilogue phases;
        //   * about one tile per CU: the producer / consumer form with a ring (64 channels wide where that still gives ~a tile per CU, the halo
        //     is then staged once for both channel halves; 4 x 16 pixel tiles for the 16 x 16 maps).
        const long t64 = sp8 * ((p.Cout + 63) / 64), t32 = sp8 * ((p.Cout + 31) / 32);
        bn = p.Cout > 32 && (nstage <= 2 || t64 >= 200) ? 64 : 32;
        th = 8;
        if (nstage <= 2 || (bn == 64 && t64 >= 320)) ns = 1;
        else if (bn == 64) ns = 3;
        else { ns = 4; if (t32 < 200 || p.Hout < 8) th = 4; }
        if (p.Hout < 8) th = 4;
        if (th == 4 && !(bn == 32 && ns == 4)) { bn = 32; ns = 4; }
        static const int xf_single = [] { const char* e = getenv("MG_H3_XF_SINGLE"); return e ? atoi(e) : 1; }();   // A/B: 0 = operand transform of Cin > 64 layers in the ring form only
        if (!xf_single && p.xf_scale && ns == 1 && p.Cin > 64) ns = bn == 64 ? 3 : 4;
    }
    // one slab, one channel tile, >= ~5 tiles per resident workgroup: the persistent form with the weights staged once (ns 201 forces it, 200 forbids it).
    // Measured (tools/h3_check.py time, us, per-tile single-role form | this form): batch 4 C32 512 x 512 forward 47.1 | 41.6, data gradient 41.9 | 33.8,
    // C32 -> 8 data gradient 34.3 (round-2 im2col form) | 29.7; batch 12 512 x 512 131.5 | 115.7, 256 x 256 30.7 | 26.5; batch 4 256 x 256 (2 048 tiles,
    // 2.7 per workgroup) 13.0 | 14.3 -- hence the threshold. PMC (profiles/r06_pmc_slab.txt, C32 512 x 512 forward, per launch): vector-ALU instructions
    // 13.6 M -> 6.4 M (414 -> 194 per wave and tile), L2 read requests 1.59 M -> 0.82 M, wave cycles 57.9 M -> 35.3 M; the matrix pipe's share is 11 us of the 42.
    static const int slab_min = [] { const char* e = getenv("MG_H3_SLAB_MIN"); return e ? atoi(e) : 4096; }();
    if (ns != 200 && nstage == 1 && p.Cout <= 32 && !p.bnb_x && (ns == 201 || (!g_h3_force[0] && sp8 >= slab_min))) return launch_h3_slab<T>(p, st);
    if (ns >= 200 || p.Cout < 16) return 1;
    if (ns >= 100) {                                         // persistent ring forms (forced: mg_set_halo3_cfg(TH, BN, 100 + NS); chosen: see above)
        if (p.xf_scale || p.bnb_x || (nstage & 1)) return 1;
        if (th == 8 && bn == 64 && ns == 103) return launch_h3_persist<T, 8, 64, 3>(p, st);
        if (th == 8 && bn == 32 && ns == 104) return launch_h3_persist<T, 8, 32, 4>(p, st);
        return 1;
    }
#define H3_CASE(TH_, BN_, NS_) if (th == TH_ && bn == BN_ && ns == NS_) return launch_h3<T, TH_, BN_, NS_>(p, st);
    H3_CASE(8, 64, 3) H3_CASE(8, 64, 1) H3_CASE(8, 32, 4) H3_CASE(8, 32, 1) H3_CASE(4, 32, 4)
#ifdef MG_H3_EXTRA_FORMS
    H3_CASE(8, 64, 2) H3_CASE(8, 32, 2) H3_CASE(4, 64, 3)
#endif
#undef H3_CASE
    return 1;
}

void h3_init() {
    static bool done = false;
    if (done) return;
    done = true;
    const char* e = getenv("MG_HALO3");
    if (g_h3_enabled < 0) g_h3_enabled = e ? atoi(e) : 1;
    if (const char* f = getenv("MG_H3_CFG")) sscanf(f, "%d,%d,%d", &g_h3_force[0], &g_h3_force[1], &g_h3_force[2]);
}

bool h3_eligible(const mg_conv_params& p) {
    h3_init();
    if (!g_h3_enabled || !MG_IS16(p.dtype) || p.m_dev || (p.mode != MG_MODE_CONV && p.mode != MG_MODE_TCONV)) return false;
    if (p.R != 3 || p.S != 3 || p.stride != 1 || p.dil != 1 || p.pad != 1 || p.Cin % 32 != 0 || p.Cout % 8 != 0 || (p.Cout < 16 && p.Cin != 32)) return false;     // (Cout 8: the one-slab persistent form only)
    if (p.Hin != p.Hout || p.Win != p.Wout || p.Wout < 16 || p.Hout < 4) return false;
    if (p.bnb_x && (p.mode != MG_MODE_TCONV || p.bnb_ld % 8)) return false;
    if (p.xf_scale && (p.mode != MG_MODE_CONV || p.Cin > 512)) return false;
    if (p.ldx % 8 || p.ldy % 8 || p.yoff % 8 || (p.res && p.ldr % 8) || (p.res2 && p.ldr2 % 8)) return false;
    return true;
}

}  // namespace

// 0: launched; 1: not a layer of this form (the caller goes on to the older kernel forms); < 0: error
extern "C" __attribute__((visibility("hidden"))) int mg_conv_halo3(const mg_conv_params* pp, void* stream) {
    const mg_conv_params& p = *pp;
    if (!h3_eligible(p)) return 1;
    if (p.dtype == MG_BF16) return dispatch_h3<bf16raw>(p, (hipStream_t)stream);
#ifndef MG_H3_BF16_ONLY
    if (p.dtype == MG_F16) return dispatch_h3<f16raw>(p, (hipStream_t)stream);
#endif
    return 1;
}
extern "C" int mg_set_halo3(int on) { h3_init(); const int old = g_h3_enabled; g_h3_enabled = on; return old; }
extern "C" int mg_set_halo3_cfg(int th, int bn, int ns) { g_h3_force[0] = th; g_h3_force[1] = bn; g_h3_force[2] = ns; return 0; }
#ifdef MG_H3_TIMING
extern "C" int mg_h3_debug_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mg_h3_dbg), sizeof(mg_h3_dbg)); }
#endif
