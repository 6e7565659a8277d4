// Instance-token <-> feature cross attention of the instance matte decoder (fp32, one head, d = 128, T <= 16 tokens).
// Reference: maggie/network/module/mask_attention.py:63-133 (CrossAttentionLayer = nn.MultiheadAttention(q + pos, k + pos, v))
// as used by maggie/network/module/instance_matte_decoder.py:170-236. One side of every attention is only `max_inst` (10)
// tokens wide, so the token side is folded into small matrices on the host side (maggie_amd/network/module/mask_attention.py)
// and the kernels here make ONE pass over the (L x 128) feature rows per direction:
//
//   tokens <- features:  S[t,l] = (Qk[t] . F[l] + Btab[t, id[l]]) * scale,  P = softmax_l(S),  Ctx[t] = sum_l P[t,l] F[l]
//   features <- tokens:  S[l,t] = (F[l] . Kq[t] + B2[id[l], t]) * scale (masked),  P = softmax_t(S),  O[l] = sum_t P[l,t] Vp[t] + bias
//
// and their exact backward passes. A feature row is handled by 16 consecutive lanes (8 channels each, two float4 loads), the
// 10 token vectors live in LDS, dot products finish with a 4-step butterfly inside the 16-lane group. Token-side gradient
// accumulators stay in registers across the rows of a group and are reduced wave -> LDS -> one atomicAdd per value per block.
// All kernels are HBM/L2-bound: F is read once (forward) / twice (backward) per direction.
#include "common.h"
#include <stdlib.h>
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;
constexpr int D = 128;          // attention width
constexpr int GL = 16;          // lanes per feature row
constexpr int CPL = D / GL;     // channels per lane (8)
constexpr int NG = NT / GL;     // row groups per block (16)
constexpr int RPG = 4;          // rows per group per block
constexpr int RPB = NG * RPG;   // rows per block (64)
constexpr int TOK_CTX_CH = 64;  // feature rows per workgroup of tok_ctx_kernel (two halves of 32 rows)

// Sum over the 16 lanes of a row group, result in all of them -- on the VALU's data-parallel-primitive path (DPP), no LDS crossbar: quad_perm
// [1,0,3,2] and [2,3,0,1] make the quads uniform, row_half_mirror (lane i <-> 7 - i) then pairs quad 0 with quad 1 (2 with 3) and row_mirror
// (i <-> 15 - i) the two halves; with uniform quads / halves any pairing across the boundary gives the sum. (__shfl_xor is ds_bpermute_b32: ~40 of
// them per feature row made the row passes shuffle-bound -- feat_fwd 34-46 us for 16 MB of traffic.)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_move<0xB1>(v);              // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);              // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);             // row_half_mirror
    v += dpp_move<0x140>(v);             // row_mirror
    return v;
}
__device__ __forceinline__ void ld8(const float* __restrict__ p, float* f) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void st8(float* __restrict__ p, const float* f) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
// Sum over the 16 lanes of a row group of T <= 16 per-lane partials s[t], lane li receiving the total of token li (0 for li >= T): a reduce-scatter
// butterfly -- at every step a lane hands the half of the values its partner keeps and adds what it receives to the half it keeps (8 + 4 + 2 + 1 = 15
// shuffles). group_sum per token is 4 * T = 40 shuffles for the ten tokens, and the kernels built on it ran 34-37 us against 22 us for the pass that
// needs none (tok_bwd2): the reduction, not the 8 MB of feature rows, was their bound.
template <int T>
__device__ __forceinline__ float group_reduce_scatter(const float* s, int li) {
    float a[8], b[4], c[2];
    const bool h8 = li & 8, h4 = li & 4, h2 = li & 2, h1 = li & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float lo = j < T ? s[j] : 0.f, hi = j + 8 < T ? s[j + 8] : 0.f;
        const float got = __shfl_xor(h8 ? lo : hi, 8, 64);
        a[j] = (h8 ? hi : lo) + got;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float got = __shfl_xor(h4 ? a[j] : a[j + 4], 4, 64);
        b[j] = (h4 ? a[j + 4] : a[j]) + got;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float got = __shfl_xor(h2 ? b[j] : b[j + 2], 2, 64);
        c[j] = (h2 ? b[j + 2] : b[j]) + got;
    }
    const float got = __shfl_xor(h1 ? c[0] : c[1], 1, 64);
    return (h1 ? c[1] : c[0]) + got;
}
template <int T>
__device__ __forceinline__ float pick(const float* v, int i) {       // v[i] for a lane-dependent i without dynamic register indexing
    float r = v[0];
#pragma unroll
    for (int t = 1; t < T; ++t) r = (i == t) ? v[t] : r;
    return r;
}
__device__ __forceinline__ void stage_tokens(float* dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += NT) dst[i] = src[i];
}
// acc[T][CPL] summed over the 4 groups of a wave, then over the 4 waves through LDS, then atomically into dst[T][D]
// `slot` (deterministic mode, csrc/det.hip): the workgroup's partial is STORED there (T * D floats) instead of being added to dst atomically
template <int T>
__device__ __forceinline__ void reduce_token_acc(float (&acc)[T][CPL], float* sred, float* __restrict__ dst, float* __restrict__ slot = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & (GL - 1);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            float v = acc[t][e];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            acc[t][e] = v;
        }
    __syncthreads();
    if (lane < GL) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int e = 0; e < CPL; ++e) sred[(wave * T + t) * D + li * CPL + e] = acc[t][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * D; i += NT) {
        const float v = (sred[i] + sred[T * D + i]) + (sred[2 * T * D + i] + sred[3 * T * D + i]);
        if (slot) slot[i] = v;
        else if (v != 0.f) atomicAdd(dst + i, v);
    }
    __syncthreads();
}

// Per-id sums of a workgroup's rows in ROW order (was: LDS float atomics, whose order depends on wave timing): the lanes li < T of every row
// publish their value and the row's id (rows that do not exist: id -1); value (t, id) then walks the RPB rows once.
// out[i], i = (TMAJOR ? t * NID + id : id * T + t).
template <int T, bool TMAJOR>
__device__ __forceinline__ void id_sums_ordered(const float* __restrict__ sds, const int* __restrict__ sid, int NID, float* __restrict__ out) {
    for (int i = threadIdx.x; i < T * NID; i += NT) {
        const int t = TMAJOR ? i / NID : i % T, id = TMAJOR ? i % NID : i / T;
        float a = 0.f;
        for (int lr = 0; lr < RPB; ++lr) a += (sid[lr] == id) ? sds[lr * T + t] : 0.f;
        out[i] = a;
    }
}

// ------------------------------------------------------------------------------------------------ tokens <- features
// RG: feature rows per 16-lane group and workgroup. The forward row passes have no cross-workgroup sums, so their grid is free: fewer rows per
// workgroup = more workgroups per CU = more of the per-row latency chain (row load -> dots -> id -> table -> store) in flight.
template <int T, int RG>
__global__ __launch_bounds__(NT) void tok_scores_kernel(const float* __restrict__ qk, const float* __restrict__ btab, const float* __restrict__ feat,
                                                        const int32_t* __restrict__ ids, int L, int NID, float scale, float* __restrict__ s_out) {
    __shared__ float sq[T * D];
    const int b = blockIdx.y;
    stage_tokens(sq, qk + (long)b * T * D, T * D);
    __syncthreads();
    const int g = threadIdx.x / GL, li = threadIdx.x % GL;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int l = blockIdx.x * (NG * RG) + r * NG + g;
        if (l >= L) continue;
        float f[CPL], s[T];
        ld8(feat + ((long)b * L + l) * D + li * CPL, f);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < CPL; ++e) a += f[e] * sq[t * D + li * CPL + e];
            s[t] = a;
        }
        const float mine = group_reduce_scatter<T>(s, li);       // lane li: the dot product of token li
        if (li < T) {
            const int id = ids[(long)b * L + l];
            s_out[((long)b * T + li) * L + l] = (mine + btab[((long)b * T + li) * NID + id]) * scale;
        }
    }
}

// in-place softmax over the last dimension of a (rows x n) matrix, one block per row
__global__ __launch_bounds__(NT) void softmax_rows_kernel(float* __restrict__ x, int n) {
    __shared__ float sh[NT / 64];
    float* row = x + (long)blockIdx.x * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += NT) m = fmaxf(m, row[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) sh[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float z = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) z += __expf(row[i] - m);
    z = wave_sum(z);
    if (lane == 0) sh[wave] = z;
    __syncthreads();
    z = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    const float inv = 1.f / z;
    for (int i = threadIdx.x; i < n; i += NT) row[i] = __expf(row[i] - m) * inv;
}

// ctx[t, c] += sum_{l in chunk} P[t,l] F[l,c]
template <int T>
__global__ __launch_bounds__(NT) void tok_ctx_kernel(const float* __restrict__ p, const float* __restrict__ feat, int L, float* __restrict__ ctx,
                                                     float* __restrict__ slots) {
    constexpr int CH = TOK_CTX_CH;
    __shared__ float sp[T * CH];
    __shared__ float sr[T * D];
    const int b = blockIdx.y, l0 = blockIdx.x * CH;
    for (int i = threadIdx.x; i < T * CH; i += NT) {
        const int t = i / CH, r = i - t * CH;
        sp[i] = (l0 + r < L) ? p[((long)b * T + t) * L + l0 + r] : 0.f;
    }
    __syncthreads();
    const int c = threadIdx.x & (D - 1), half = threadIdx.x / D;
    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.f;
    const int rbeg = half * (CH / 2), rend = min(rbeg + CH / 2, L - l0);
    int r = rbeg;
    // sixteen rows' loads in flight per batch (the one-row-per-iteration loop waited for every load before issuing the next: 64 memory round trips per
    // thread, 20 us per launch for 8 MB); the additions keep the row order
    for (; r + 16 <= rend; r += 16) {
        float fv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) fv[u] = feat[((long)b * L + l0 + r + u) * D + c];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] += sp[t * CH + r + u] * fv[u];
        }
    }
    for (; r < rend; ++r) {
        const float f = feat[((long)b * L + l0 + r) * D + c];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] += sp[t * CH + r] * f;
    }
    if (half == 1) {
#pragma unroll
        for (int t = 0; t < T; ++t) sr[t * D + c] = acc[t];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (slots) slots[(((size_t)b * gridDim.x + blockIdx.x) * T + t) * D + c] = acc[t] + sr[t * D + c];     // deterministic mode: one row [T][D] per workgroup
            else atomicAdd(&ctx[((long)b * T + t) * D + c], acc[t] + sr[t * D + c]);
        }
    }
}

// backward pass 1: G[t,l] = dP[t,l] + dCtx[t] . F[l];  rowdot[t] += sum_l P[t,l] G[t,l]
template <int T>
__global__ __launch_bounds__(NT) void tok_bwd1_kernel(const float* __restrict__ p, const float* __restrict__ feat, const float* __restrict__ dctx,
                                                      const float* __restrict__ dp, int L, float* __restrict__ gbuf, float* __restrict__ rowdot,
                                                      float* __restrict__ slots) {
    __shared__ float sd[T * D];
    __shared__ float sacc[(NT / 64) * 16];
    const int b = blockIdx.y;
    stage_tokens(sd, dctx + (long)b * T * D, T * D);
    __syncthreads();
    const int g = threadIdx.x / GL, li = threadIdx.x % GL;
    float rd = 0.f;
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        const int l = blockIdx.x * RPB + r * NG + g;
        if (l >= L) continue;
        float f[CPL], s[T];
        ld8(feat + ((long)b * L + l) * D + li * CPL, f);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < CPL; ++e) a += f[e] * sd[t * D + li * CPL + e];
            s[t] = a;
        }
        const float mine = group_reduce_scatter<T>(s, li);
        if (li < T) {
            const long o = ((long)b * T + li) * L + l;
            const float gv = mine + (dp ? dp[o] : 0.f);
            gbuf[o] = gv;
            rd += p[o] * gv;
        }
    }
    rd += __shfl_xor(rd, 16, 64); rd += __shfl_xor(rd, 32, 64);          // the 4 groups of a wave share the lane -> token map
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 16) sacc[wave * 16 + lane] = rd;
    __syncthreads();
    if (threadIdx.x < T) {
        const float v = (sacc[threadIdx.x] + sacc[16 + threadIdx.x]) + (sacc[32 + threadIdx.x] + sacc[48 + threadIdx.x]);
        if (slots) slots[((size_t)b * gridDim.x + blockIdx.x) * T + threadIdx.x] = v;
        else atomicAdd(&rowdot[b * T + threadIdx.x], v);
    }
}

// backward pass 2: dS = P (G - rowdot) scale;  dF[l] = sum_t dS Qk[t] + P dCtx[t];  dQk[t] += sum_l dS F[l];  dBtab[t, id] += dS
template <int T>
__global__ __launch_bounds__(NT) void tok_bwd2_kernel(const float* __restrict__ p, const float* __restrict__ feat, const float* __restrict__ qk,
                                                      const float* __restrict__ dctx, const int32_t* __restrict__ ids, const float* __restrict__ gbuf,
                                                      const float* __restrict__ rowdot, int L, int NID, float scale, float* __restrict__ dfeat,
                                                      float* __restrict__ dqk, float* __restrict__ dbtab, float* __restrict__ slots) {
    extern __shared__ float smem[];
    float* sq = smem;                    // [T][D]
    float* sd = smem + T * D;            // [T][D]
    float* sred = smem + 2 * T * D;      // [4][T][D]
    float* sb = sred + 4 * T * D;        // [T][NID]
    float* sds = sb + T * NID;           // [RPB][T]  dS of the workgroup's rows
    int* sid = (int*)(sds + RPB * T);    // [RPB]     their ids (-1: no such row)
    const int b = blockIdx.y;
    stage_tokens(sq, qk + (long)b * T * D, T * D);
    stage_tokens(sd, dctx + (long)b * T * D, T * D);
    for (int i = threadIdx.x; i < T * NID; i += NT) sb[i] = 0.f;
    __syncthreads();
    const int g = threadIdx.x / GL, li = threadIdx.x % GL, gbase = (threadIdx.x & 63) - li;
    const float rdot = li < T ? rowdot[b * T + li] : 0.f;
    float acc[T][CPL];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < CPL; ++e) acc[t][e] = 0.f;
    int idr[RPG];                                                // the rows' ids, loaded up front (inside the loop each one was a late round trip of its own,
#pragma unroll                                                   // issued behind -- and so waiting for -- the previous row's stores)
    for (int r = 0; r < RPG; ++r) {
        const int l = blockIdx.x * RPB + r * NG + g;
        idr[r] = l < L ? ids[(long)b * L + l] : -1;
    }
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        const int l = blockIdx.x * RPB + r * NG + g;
        const bool ok = l < L;                                   // uniform inside a 16-lane group
        float f[CPL], out[CPL], pv = 0.f, ds = 0.f;
#pragma unroll
        for (int e = 0; e < CPL; ++e) { f[e] = 0.f; out[e] = 0.f; }
        if (ok) {
            ld8(feat + ((long)b * L + l) * D + li * CPL, f);
            if (li < T) {
                const long o = ((long)b * T + li) * L + l;
                pv = p[o];
                ds = pv * (gbuf[o] - rdot) * scale;
            }
        }
        if (li < T) sds[(r * NG + g) * T + li] = ds;
        if (li == 0) sid[r * NG + g] = idr[r];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float dst = __shfl(ds, gbase + t, 64), pt = __shfl(pv, gbase + t, 64);
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                out[e] += dst * sq[t * D + li * CPL + e] + pt * sd[t * D + li * CPL + e];
                acc[t][e] += dst * f[e];
            }
        }
        if (ok) st8(dfeat + ((long)b * L + l) * D + li * CPL, out);
    }
    float* slot = slots ? slots + ((size_t)b * gridDim.x + blockIdx.x) * (T * D + T * NID) : nullptr;     // row [dQk (T x D) | dBtab (T x NID)]
    reduce_token_acc<T>(acc, sred, dqk + (long)b * T * D, slot);       // (its barriers also publish sds / sid)
    id_sums_ordered<T, true>(sds, sid, NID, sb);
    for (int i = threadIdx.x; i < T * NID; i += NT) {                 // each thread reads back what it wrote itself
        if (slot) slot[T * D + i] = sb[i];
        else if (sb[i] != 0.f) atomicAdd(&dbtab[(long)b * T * NID + i], sb[i]);
    }
}

// ------------------------------------------------------------------------------------------------ features <- tokens
template <int T, int RG>
__global__ __launch_bounds__(NT) void feat_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ kq, const float* __restrict__ b2,
                                                      const float* __restrict__ vp, const float* __restrict__ obias, const uint8_t* __restrict__ pad,
                                                      const int32_t* __restrict__ ids, int L, int NID, float scale, float* __restrict__ out,
                                                      float* __restrict__ p_out, int b2_tn) {
    __shared__ float sk[T * D];
    __shared__ float sv[T * D];
    const int b = blockIdx.y;
    stage_tokens(sk, kq + (long)b * T * D, T * D);
    stage_tokens(sv, vp + (long)b * T * D, T * D);
    __syncthreads();
    const int g = threadIdx.x / GL, li = threadIdx.x % GL;
    float ob[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) ob[e] = obias ? obias[li * CPL + e] : 0.f;
    unsigned padmask = 0u;                                     // the sample's key-padding flags, one bit per token (loaded once)
    if (pad) {
#pragma unroll
        for (int t = 0; t < T; ++t) padmask |= (pad[b * T + t] ? 1u : 0u) << t;
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int l = blockIdx.x * (NG * RG) + r * NG + g;
        if (l >= L) continue;
        float f[CPL], s[T];
        ld8(feat + ((long)b * L + l) * D + li * CPL, f);
        const int id = ids[(long)b * L + l];
        // the row's T table values as T independent loads in front of the token loop (inside it the compiler serialised
        // "load mask byte -> wait -> branch -> load table value -> wait" per token: ~20 memory round trips per row, 28-43 us per launch)
        float tb[T];
#pragma unroll
        for (int t = 0; t < T; ++t) tb[t] = b2[(long)b * NID * T + (b2_tn ? t * NID + id : id * T + t)];      // b2_tn: the table as (B, T, NID)
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < CPL; ++e) a += f[e] * sk[t * D + li * CPL + e];
            a = (group_sum(a) + tb[t]) * scale;
            a = ((padmask >> t) & 1u) ? -INFINITY : a;
            s[t] = a;
            m = fmaxf(m, a);
        }
        float z = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) { s[t] = __expf(s[t] - m); z += s[t]; }
        const float inv = 1.f / z;
        float o[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) o[e] = ob[e];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            s[t] *= inv;
#pragma unroll
            for (int e = 0; e < CPL; ++e) o[e] += s[t] * sv[t * D + li * CPL + e];
        }
        st8(out + ((long)b * L + l) * D + li * CPL, o);
        if (li < T) p_out[((long)b * L + l) * T + li] = pick<T>(s, li);
    }
}

template <int T>
__global__ __launch_bounds__(NT) void feat_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ p, const float* __restrict__ feat,
                                                      const float* __restrict__ kq, const float* __restrict__ vp, const int32_t* __restrict__ ids,
                                                      int L, int NID, float scale, float* __restrict__ dfeat, float* __restrict__ dkq,
                                                      float* __restrict__ dvp, float* __restrict__ db2, float* __restrict__ dob, float* __restrict__ slots,
                                                      int b2_tn) {
    extern __shared__ float smem[];
    float* sk = smem;                    // [T][D]
    float* sv = smem + T * D;            // [T][D]
    float* sred = smem + 2 * T * D;      // [4][T][D]
    float* sb = sred + 4 * T * D;        // [NID][T]
    float* sds = sb + T * NID;           // [RPB][T]  dS of the workgroup's rows
    int* sid = (int*)(sds + RPB * T);    // [RPB]     their ids (-1: no such row)
    const int b = blockIdx.y;
    stage_tokens(sk, kq + (long)b * T * D, T * D);
    stage_tokens(sv, vp + (long)b * T * D, T * D);
    for (int i = threadIdx.x; i < T * NID; i += NT) sb[i] = 0.f;
    __syncthreads();
    const int g = threadIdx.x / GL, li = threadIdx.x % GL;
    float akq[T][CPL], avp[T][CPL], aob[CPL];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < CPL; ++e) { akq[t][e] = 0.f; avp[t][e] = 0.f; }
#pragma unroll
    for (int e = 0; e < CPL; ++e) aob[e] = 0.f;
    int idr[RPG];                                                // (see tok_bwd2_kernel)
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        const int l = blockIdx.x * RPB + r * NG + g;
        idr[r] = l < L ? ids[(long)b * L + l] : -1;
    }
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        const int l = blockIdx.x * RPB + r * NG + g;
        const bool ok = l < L;
        float f[CPL], go[CPL], pr[T], ds[T], df[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) { f[e] = 0.f; go[e] = 0.f; df[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < T; ++t) pr[t] = 0.f;
        if (ok) {
            ld8(feat + ((long)b * L + l) * D + li * CPL, f);
            ld8(dout + ((long)b * L + l) * D + li * CPL, go);
#pragma unroll
            for (int t = 0; t < T; ++t) pr[t] = p[((long)b * L + l) * T + t];
        }
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < CPL; ++e) a += go[e] * sv[t * D + li * CPL + e];
            ds[t] = group_sum(a);                                  // dP[t]
            dot += pr[t] * ds[t];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            ds[t] = pr[t] * (ds[t] - dot) * scale;
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                df[e] += ds[t] * sk[t * D + li * CPL + e];
                akq[t][e] += ds[t] * f[e];
                avp[t][e] += pr[t] * go[e];
            }
        }
#pragma unroll
        for (int e = 0; e < CPL; ++e) aob[e] += go[e];
        if (ok) st8(dfeat + ((long)b * L + l) * D + li * CPL, df);
        if (li < T) sds[(r * NG + g) * T + li] = ok ? pick<T>(ds, li) : 0.f;
        if (li == 0) sid[r * NG + g] = idr[r];
    }
    // deterministic mode: the workgroup's row [dKq (T x D) | dVp (T x D) | dB2 (NID x T) | dObias (D)]
    float* slot = slots ? slots + ((size_t)b * gridDim.x + blockIdx.x) * (2 * T * D + T * NID + D) : nullptr;
    reduce_token_acc<T>(akq, sred, dkq + (long)b * T * D, slot);
    reduce_token_acc<T>(avp, sred, dvp + (long)b * T * D, slot ? slot + T * D : nullptr);
    // bias gradient: 16-lane channel slices over the 16 groups of the block
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int e = 0; e < CPL; ++e) { float v = aob[e]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); aob[e] = v; }
        if (lane < GL) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) sred[wave * D + li * CPL + e] = aob[e];
        }
        __syncthreads();
        if (threadIdx.x < D) {
            const float v = (sred[threadIdx.x] + sred[D + threadIdx.x]) + (sred[2 * D + threadIdx.x] + sred[3 * D + threadIdx.x]);
            if (slot) slot[2 * T * D + T * NID + threadIdx.x] = v;
            else if (dob && v != 0.f) atomicAdd(&dob[threadIdx.x], v);
        }
    }
    id_sums_ordered<T, false>(sds, sid, NID, sb);
    for (int i = threadIdx.x; i < T * NID; i += NT) {
        const int o = b2_tn ? (i % T) * NID + i / T : i;          // sb is [NID][T]; b2_tn: the gradient leaves as (T, NID), like the table came
        if (slot) slot[2 * T * D + o] = sb[i];
        else if (sb[i] != 0.f) atomicAdd(&db2[(long)b * NID * T + o], sb[i]);
    }
}

inline dim3 row_grid(int L, int B) { return dim3((L + RPB - 1) / RPB, B); }
inline int fwd_rows_per_group() {                               // MG_ATTN_FWD_RG = 1 | 2 | 4 (A/B switch)
    static const int v = [] { const char* e = getenv("MG_ATTN_FWD_RG"); const int x = e ? atoi(e) : 1; return (x == 1 || x == 2) ? x : 4; }();
    return v;
}
inline int attn_check(int B, int T, int L, int Dm, int NID) {
    if (Dm != D || T != 10 || NID < 1 || NID > 64) return -3;      // built for maggie_{image,video}.yaml: attention_dim 128, max_inst 10
    if (B <= 0 || L <= 0) return -1;
    return 0;
}
constexpr int TT = 10;

}  // namespace

// zero n buffers, merging runs that are adjacent in memory (in the given order) into one fill launch
static int zero_adjacent(float* const* bufs, const long* words, int n, hipStream_t st) {
    int i = 0;
    while (i < n) {
        if (!bufs[i] || words[i] <= 0) { ++i; continue; }
        float* base = bufs[i];
        long total = words[i];
        int j = i + 1;
        while (j < n && bufs[j] && words[j] > 0 && bufs[j] == base + total) { total += words[j]; ++j; }
        hipError_t e = mg_zero_words(base, total, st);
        if (e != hipSuccess) return (int)e;
        i = j;
    }
    return 0;
}

extern "C" int mg_attn_tok_fwd(const float* qk, const float* btab, const float* feat, const int32_t* ids, int B, int T, int L, int Dm, int NID,
                               float scale, float* p, float* ctx, void* stream) {
    int rc = attn_check(B, T, L, Dm, NID); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    {
        const int rg = fwd_rows_per_group();
        const dim3 grid((L + NG * rg - 1) / (NG * rg), B);
        if (rg == 1) hipLaunchKernelGGL((tok_scores_kernel<TT, 1>), grid, dim3(NT), 0, st, qk, btab, feat, ids, L, NID, scale, p);
        else if (rg == 2) hipLaunchKernelGGL((tok_scores_kernel<TT, 2>), grid, dim3(NT), 0, st, qk, btab, feat, ids, L, NID, scale, p);
        else hipLaunchKernelGGL((tok_scores_kernel<TT, 4>), grid, dim3(NT), 0, st, qk, btab, feat, ids, L, NID, scale, p);
    }
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(B * T), dim3(NT), 0, st, p, L);
    hipError_t e = mg_zero_words(ctx, (long)B * T * Dm, st); if (e != hipSuccess) return (int)e;
    const int nblk = (L + TOK_CTX_CH - 1) / TOK_CTX_CH;
    float* slots = nullptr;
    if (mg_det_on && nblk > 1) { slots = mg_det_scratch((long)B * nblk * T * Dm); if (!slots) return MG_DET_NO_SCRATCH; }
    hipLaunchKernelGGL(tok_ctx_kernel<TT>, dim3(nblk, B), dim3(NT), 0, st, p, feat, L, ctx, slots);
    MG_CHECK_LAUNCH();
    if (slots) {
        mg_det_seg sg{ctx, T * Dm, (long)T * Dm};
        return mg_det_reduce(slots, nblk, B, T * Dm, 0, &sg, 1, st);
    }
    return 0;
}

extern "C" int mg_attn_tok_bwd(const float* p, const float* feat, const float* qk, const int32_t* ids, const float* dctx, const float* dp, int B, int T,
                               int L, int Dm, int NID, float scale, float* gbuf, float* rowdot, float* dqk, float* dbtab, float* dfeat,
                               void* stream) {
    int rc = attn_check(B, T, L, Dm, NID); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    {   // accumulators that sit back to back in memory (the Python binding carves them from one allocation) are zeroed by one fill launch
        float* bufs[3] = {rowdot, dqk, dbtab};
        const long words[3] = {(long)B * T, (long)B * T * Dm, (long)B * T * NID};
        int rcz = zero_adjacent(bufs, words, 3, st); if (rcz) return rcz;
    }
    // deterministic mode (csrc/det.hip): both kernels store one partial row per workgroup, added in workgroup order right behind them
    const int nblk = (int)row_grid(L, B).x;
    const int w2 = T * Dm + T * NID;
    float* slots = nullptr;
    if (mg_det_on && nblk > 1) { slots = mg_det_scratch((long)B * nblk * w2); if (!slots) return MG_DET_NO_SCRATCH; }
    hipLaunchKernelGGL(tok_bwd1_kernel<TT>, row_grid(L, B), dim3(NT), 0, st, p, feat, dctx, dp, L, gbuf, rowdot, slots);
    if (slots) {
        mg_det_seg sg{rowdot, T, (long)T};
        int rcd = mg_det_reduce(slots, nblk, B, T, 0, &sg, 1, st); if (rcd) return rcd;
    }
    const size_t lds = (size_t)(6 * TT * D + TT * NID + RPB * TT + RPB) * sizeof(float);
    hipLaunchKernelGGL(tok_bwd2_kernel<TT>, row_grid(L, B), dim3(NT), lds, st, p, feat, qk, dctx, ids, gbuf, rowdot, L, NID, scale, dfeat, dqk, dbtab, slots);
    MG_CHECK_LAUNCH();
    if (slots) {
        mg_det_seg sg[2] = {{dqk, T * Dm, (long)T * Dm}, {dbtab, T * NID, (long)T * NID}};
        return mg_det_reduce(slots, nblk, B, w2, 0, sg, 2, st);
    }
    return 0;
}

// b2_tn != 0: the score-bias table (and its gradient) is laid out (B, T, NID) -- the way the token-side linear writes it -- instead of (B, NID, T)
extern "C" int mg_attn_feat_fwd_ex(const float* feat, const float* kq, const float* b2, const float* vp, const float* obias, const uint8_t* pad_mask,
                                   const int32_t* ids, int B, int T, int L, int Dm, int NID, float scale, float* out, float* p, int b2_tn, void* stream) {
    int rc = attn_check(B, T, L, Dm, NID); if (rc) return rc;
    const int rg = fwd_rows_per_group();
    const dim3 grid((L + NG * rg - 1) / (NG * rg), B);
    if (rg == 1) hipLaunchKernelGGL((feat_fwd_kernel<TT, 1>), grid, dim3(NT), 0, (hipStream_t)stream, feat, kq, b2, vp, obias, pad_mask, ids, L, NID, scale, out, p, b2_tn);
    else if (rg == 2) hipLaunchKernelGGL((feat_fwd_kernel<TT, 2>), grid, dim3(NT), 0, (hipStream_t)stream, feat, kq, b2, vp, obias, pad_mask, ids, L, NID, scale, out, p, b2_tn);
    else hipLaunchKernelGGL((feat_fwd_kernel<TT, 4>), grid, dim3(NT), 0, (hipStream_t)stream, feat, kq, b2, vp, obias, pad_mask, ids, L, NID, scale, out, p, b2_tn);
    MG_CHECK_LAUNCH();
    return 0;
}
extern "C" int mg_attn_feat_fwd(const float* feat, const float* kq, const float* b2, const float* vp, const float* obias, const uint8_t* pad_mask,
                                const int32_t* ids, int B, int T, int L, int Dm, int NID, float scale, float* out, float* p, void* stream) {
    return mg_attn_feat_fwd_ex(feat, kq, b2, vp, obias, pad_mask, ids, B, T, L, Dm, NID, scale, out, p, 0, stream);
}

extern "C" int mg_attn_feat_bwd_ex(const float* dout, const float* p, const float* feat, const float* kq, const float* vp, const int32_t* ids, int B, int T,
                                   int L, int Dm, int NID, float scale, float* dfeat, float* dkq, float* dvp, float* db2, float* dobias, int b2_tn,
                                   void* stream);
extern "C" int mg_attn_feat_bwd(const float* dout, const float* p, const float* feat, const float* kq, const float* vp, const int32_t* ids, int B, int T,
                                int L, int Dm, int NID, float scale, float* dfeat, float* dkq, float* dvp, float* db2, float* dobias, void* stream) {
    return mg_attn_feat_bwd_ex(dout, p, feat, kq, vp, ids, B, T, L, Dm, NID, scale, dfeat, dkq, dvp, db2, dobias, 0, stream);
}
extern "C" int mg_attn_feat_bwd_ex(const float* dout, const float* p, const float* feat, const float* kq, const float* vp, const int32_t* ids, int B, int T,
                                   int L, int Dm, int NID, float scale, float* dfeat, float* dkq, float* dvp, float* db2, float* dobias, int b2_tn,
                                   void* stream) {
    int rc = attn_check(B, T, L, Dm, NID); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    {
        float* bufs[4] = {dkq, dvp, db2, dobias};
        const long words[4] = {(long)B * T * Dm, (long)B * T * Dm, (long)B * NID * T, dobias ? (long)Dm : 0l};
        int rcz = zero_adjacent(bufs, words, 4, st); if (rcz) return rcz;
    }
    const size_t lds = (size_t)(6 * TT * D + TT * NID + RPB * TT + RPB) * sizeof(float);
    const int nblk = (int)row_grid(L, B).x;
    const int w = 2 * T * Dm + T * NID + Dm;
    float* slots = nullptr;
    if (mg_det_on && (nblk > 1 || B > 1)) { slots = mg_det_scratch((long)B * nblk * w); if (!slots) return MG_DET_NO_SCRATCH; }
    hipLaunchKernelGGL(feat_bwd_kernel<TT>, row_grid(L, B), dim3(NT), lds, st, dout, p, feat, kq, vp, ids, L, NID, scale, dfeat, dkq, dvp, db2, dobias, slots,
                       b2_tn);
    MG_CHECK_LAUNCH();
    if (slots) {
        mg_det_seg sg[3] = {{dkq, T * Dm, (long)T * Dm}, {dvp, T * Dm, (long)T * Dm}, {db2, T * NID, (long)T * NID}};
        int rcd = mg_det_reduce(slots, nblk, B, w, 0, sg, 3, st); if (rcd) return rcd;
        if (dobias) {                                             // the output bias is shared by the samples: one sum over all B * nblk rows
            mg_det_seg sb{dobias, Dm, 0};
            return mg_det_reduce(slots, B * nblk, 1, w, 2 * T * Dm + T * NID, &sb, 1, st);
        }
    }
    return 0;
}
