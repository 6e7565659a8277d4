// Token side of the instance matte decoder (maggie/network/module/mask_attention.py:9-206, instance_matte_decoder.py:219-299): every
// operation on the (batch x 10 instance tokens) x 128 matrices -- query / key / value projections, the out-projections, FFN and MLP
// layers, post-norm residual LayerNorms, the 10 x 10 token self-attention. The reference runs them as ~150 cuBLAS + ~250 elementwise
// launches per training step on tensors of 40 rows; here each layer is one fused launch forward and two backward (rows in LDS, W in L2):
//   mg_token_linear_fwd : y = LN( res + act( (x + xadd) W^T + b ) )   [xadd, b, res, ReLU, LayerNorm each optional]
//   mg_token_linear_bwd : dx (= dxadd), dW, db, dres, dgamma, dbeta of the above (a row-parallel and a column-parallel launch)
//   mg_token_sa_fwd/bwd : softmax(q k^T / sqrt(d), key padding mask) v for the T tokens of each batch element
// fp32 throughout (the reference keeps these tiny operands in fp32 too). Any row count; K a multiple of 4.
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_row_sum(float v, float* red) {      // sum over the 256 threads of the block
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Row-parallel forward: a workgroup owns RB = 4 rows and all N columns. W (<= 64 KB) is staged into LDS once per workgroup with
// coalesced loads (transposed, pitch N + 1: conflict-free both ways) -- reading it per thread straight from L2 made the kernel a chain
// of exposed round trips (9 us per layer); the row block of x sits in LDS too (broadcast reads). The LayerNorm of a row is finished
// inside the same workgroup.
constexpr int RB = 4;

__device__ __forceinline__ void token_linear_fwd_body(int bx, const float* __restrict__ x, const float* __restrict__ xadd, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ res, int relu,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      float* __restrict__ y, float* __restrict__ z_out, float* __restrict__ rstat, int R, int K,
                                                      int N, int wt) {
    extern __shared__ float sm[];
    float* sx = sm;                      // [RB][K]  x + xadd
    float* sy = sm + RB * K;             // [RB][N]  pre-LayerNorm values
    float* sw = sy + RB * N;             // [K][N + 1]  W transposed
    const int t = threadIdx.x;
    const int r0 = bx * RB;
    const int nr = min(RB, R - r0);
    const int P = N + 1;
    {   // W -> LDS transposed; 8 x 16-byte loads in flight per thread (a load-store-load chain costs a round trip per element)
        const int total4 = N * K / 4;
        for (int base = t; base < total4; base += NT * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i4 = base + u * NT; if (i4 < total4) v[u] = ((const float4*)W)[i4]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i4 = base + u * NT;
                if (i4 < total4) {
                    const int i = i4 * 4;
                    if (wt) {                            // W given as [K][N] (y = x W: the `wk.t()` products of mask_attention.py without a transposed copy)
                        const int k = i / N, n = i - k * N;
                        sw[k * P + n] = v[u].x; sw[k * P + n + 1] = v[u].y; sw[k * P + n + 2] = v[u].z; sw[k * P + n + 3] = v[u].w;
                    } else {
                        const int n = i / K, k = i - n * K;
                        sw[k * P + n] = v[u].x; sw[(k + 1) * P + n] = v[u].y; sw[(k + 2) * P + n] = v[u].z; sw[(k + 3) * P + n] = v[u].w;
                    }
                }
            }
        }
    }
    for (int i = t; i < RB * K; i += NT) sx[i] = i < nr * K ? x[(size_t)r0 * K + i] + (xadd ? xadd[(size_t)r0 * K + i] : 0.f) : 0.f;
    __syncthreads();
    // thread -> (column n, row pair): with N >= 128 two row pairs run side by side, narrower layers use fewer threads
    const int cols = N < 128 ? N : 128;
    const int c = t % cols, h = t / cols;            // h-th group of rows
    const int groups = NT / cols;                    // row groups in flight
    for (int n = c; n < N; n += cols) {
        for (int rl0 = h; rl0 < RB; rl0 += groups) {
            float acc = 0.f;
            const float* xr = sx + rl0 * K;
#pragma unroll 8
            for (int k = 0; k < K; ++k) acc += sw[k * P + n] * xr[k];
            if (rl0 >= nr) continue;
            float v = acc + (bias ? bias[n] : 0.f);
            if (relu) v = v > 0.f ? v : 0.f;
            if (res) v += res[(size_t)(r0 + rl0) * N + n];
            if (gamma) sy[rl0 * N + n] = v;
            else y[(size_t)(r0 + rl0) * N + n] = v;
        }
    }
    if (!gamma) return;
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;          // LayerNorm: wave w normalises row w of the block
    for (int rl = wave; rl < nr; rl += 4) {
        float s = 0.f;
        for (int n = lane; n < N; n += 64) s += sy[rl * N + n];
        const float mean = wave_sum(s) / (float)N;
        float q = 0.f;
        for (int n = lane; n < N; n += 64) { const float d = sy[rl * N + n] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / (float)N + eps);
        for (int n = lane; n < N; n += 64) {
            const float v = sy[rl * N + n];
            if (z_out) z_out[(size_t)(r0 + rl) * N + n] = v;
            y[(size_t)(r0 + rl) * N + n] = (v - mean) * rstd * gamma[n] + beta[n];
        }
        if (lane == 0 && rstat) { rstat[2 * (r0 + rl)] = mean; rstat[2 * (r0 + rl) + 1] = rstd; }
    }
}

__global__ __launch_bounds__(NT) void token_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ xadd, const float* __restrict__ W,
                                                              const float* __restrict__ bias, const float* __restrict__ res, int relu,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              float* __restrict__ y, float* __restrict__ z_out, float* __restrict__ rstat, int R, int K,
                                                              int N, int wt) {
    token_linear_fwd_body(blockIdx.x, x, xadd, W, bias, res, relu, gamma, beta, eps, y, z_out, rstat, R, K, N, wt);
}

// Several INDEPENDENT layers in one launch (blockIdx.y = layer): the q / k / v projections of the token self-attention, the folded key /
// table products of a cross attention, ... -- each was a launch of 10 workgroups at the latency floor (9 us forward, 8 + 5 us backward).
struct TokLin {
    const float *x, *xadd, *W, *bias, *res, *gamma, *beta;
    float *y, *z, *rstat;
    // backward
    const float *dy, *yout;
    float *dx, *dW, *db, *dres, *dgamma, *dbeta, *dz;
    int R, K, N, relu, wt;
    float eps;
    int dx_pair;                         // backward: 1 + index of another layer of the set that reads the SAME x (no xadd on either): its dz . W is added into
                                         // this layer's dx (that layer has dx == NULL) -- one input gradient instead of two and an autograd add; 0: none
};
constexpr int TOK_MULTI = 6;
struct TokLinSet { TokLin op[TOK_MULTI]; int n; };

__global__ __launch_bounds__(NT) void token_linear_multi_fwd_kernel(const TokLinSet set) {
    const TokLin& p = set.op[blockIdx.y];
    if ((int)blockIdx.x * RB >= p.R) return;
    token_linear_fwd_body(blockIdx.x, p.x, p.xadd, p.W, p.bias, p.res, p.relu, p.gamma, p.beta, p.eps, p.y, p.z, p.rstat, p.R, p.K, p.N, p.wt);
}

// Backward, kernel A (row-parallel, RB rows per workgroup): dz = gradient at the pre-LayerNorm point (= dres), ReLU mask, and
// dx[r][k] = sum_n dz[r][n] W[n][k] with W staged row-major in LDS (thread per input column k: conflict-free).
__device__ __forceinline__ void token_linear_dz_body(int bx, const float* __restrict__ dy, const float* __restrict__ yout, int relu,
                                                     const float* __restrict__ gamma, const float* __restrict__ z, const float* __restrict__ rstat,
                                                     float* __restrict__ dz_out, float* __restrict__ dres, int R, int N) {
    // one wave per row: LayerNorm backward (two row reductions), residual gradient, ReLU mask -> dz
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = bx * 4 + wave;
    if (r >= R) return;
    const size_t ro = (size_t)r * N;
    float s1 = 0.f, s2 = 0.f, mean = 0.f, rstd = 1.f;
    if (gamma) {
        mean = rstat[2 * r]; rstd = rstat[2 * r + 1];
        for (int n = lane; n < N; n += 64) {
            const float gg = dy[ro + n] * gamma[n], xh = (z[ro + n] - mean) * rstd;
            s1 += gg; s2 += gg * xh;
        }
        s1 = wave_sum(s1) / (float)N; s2 = wave_sum(s2) / (float)N;
    }
    for (int n = lane; n < N; n += 64) {
        float g = dy[ro + n];
        if (gamma) { const float xh = (z[ro + n] - mean) * rstd; g = rstd * (g * gamma[n] - s1 - xh * s2); }
        if (dres) dres[ro + n] = g;                                   // the residual enters after the activation
        if (relu && !(yout[ro + n] > 0.f)) g = 0.f;
        dz_out[ro + n] = g;
    }
}

__global__ __launch_bounds__(NT) void token_linear_dz_kernel(const float* __restrict__ dy, const float* __restrict__ yout, int relu,
                                                            const float* __restrict__ gamma, const float* __restrict__ z, const float* __restrict__ rstat,
                                                            float* __restrict__ dz_out, float* __restrict__ dres, int R, int N) {
    token_linear_dz_body(blockIdx.x, dy, yout, relu, gamma, z, rstat, dz_out, dres, R, N);
}
__global__ __launch_bounds__(NT) void token_linear_multi_dz_kernel(const TokLinSet set) {
    const TokLin& p = set.op[blockIdx.y];
    if (p.dz == p.dy && !p.dres) return;                  // a plain layer: dz IS dy (see mg_token_linear_bwd)
    if ((int)blockIdx.x * 4 >= p.R) return;
    token_linear_dz_body(blockIdx.x, p.dy, p.yout, p.relu, p.gamma, p.z, p.rstat, p.dz, p.dres, p.R, p.N);
}

// dx[r][k] = sum_n dz[r][n] W[n][k]: RB rows per workgroup, W staged once in LDS (the row-parallel half of the main backward launch)
template <bool ADD = false>
__device__ __forceinline__ void token_linear_bwd_rows(float* sm, int rblock, const float* __restrict__ dz, const float* __restrict__ W,
                                                      float* __restrict__ dx, int R, int K, int N, int wt) {
    float* sdz = sm;                     // [RB][N]
    float* sw = sm + RB * N;             // [N][K]
    const int t = threadIdx.x;
    const int r0 = rblock * RB;
    const int nr = min(RB, R - r0);
    const int total4 = N * K / 4;
    for (int base = t; base < total4; base += NT * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i4 = base + u * NT; if (i4 < total4) v[u] = ((const float4*)W)[i4]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i4 = base + u * NT;
            if (i4 >= total4) continue;
            if (wt) {                                    // W is [K][N]: the LDS copy stays [N][K]
                const int i = i4 * 4, k = i / N, n = i - k * N;
                sw[n * K + k] = v[u].x; sw[(n + 1) * K + k] = v[u].y; sw[(n + 2) * K + k] = v[u].z; sw[(n + 3) * K + k] = v[u].w;
            } else ((float4*)sw)[i4] = v[u];
        }
    }
    for (int i = t; i < RB * N; i += NT) { const int rl = i / N; sdz[i] = rl < nr ? dz[(size_t)r0 * N + i] : 0.f; }
    __syncthreads();
    const int cols = K < 128 ? K : 128;
    const int c = t % cols, h = t / cols, groups = NT / cols;
    for (int k = c; k < K; k += cols) {
        for (int rl = h; rl < nr; rl += groups) {
            float acc = 0.f;
            const float* g = sdz + rl * N;
#pragma unroll 8
            for (int n = 0; n < N; ++n) acc += g[n] * sw[n * K + k];
            // ADD: the second source of a pair -- the same thread wrote this element in the first pass (same (k, rl) walk)
            if (ADD) dx[(size_t)(r0 + rl) * K + k] += acc; else dx[(size_t)(r0 + rl) * K + k] = acc;
        }
    }
}

// Backward, kernel B (column-parallel, CB = 4 output columns per workgroup): dW[n][k] = sum_r dz[r][n] x'[r][k], db[n] = sum_r dz[r][n],
// dgamma[n] = sum_r dy[r][n] xhat[r][n], dbeta[n] = sum_r dy[r][n]; x' = x + xadd staged in LDS in row chunks of 32.
constexpr int CB = 4;

__device__ __forceinline__ void token_linear_bwd_cols(float* sm, int cblock, const float* __restrict__ dz, const float* __restrict__ dy,
                                                      const float* __restrict__ x, const float* __restrict__ xadd, const float* __restrict__ gamma,
                                                      const float* __restrict__ z, const float* __restrict__ rstat, float* __restrict__ dW,
                                                      float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta, int R, int K, int N,
                                                      int wt) {
    constexpr int RC = 32;               // rows staged per pass
    float* sg = sm;                      // [RC][CB] dz of this block's columns
    float* sx = sm + RC * CB;            // [RC][K]
    const int t = threadIdx.x;
    const int n0 = cblock * CB;
    const int nc = min(CB, N - n0);
    const int cols = K < 128 ? K : 128;
    const int c = t % cols, h = t / cols, groups = NT / cols;          // thread -> (k, column group)
    // this thread's (k, column) pairs: k = c + ki * cols (ki < 2: K <= 256), column = h + cj * groups (cj < 2: groups >= 2)
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float sb = 0.f, ag = 0.f, ab = 0.f;
    for (int rb = 0; rb < R; rb += RC) {
        const int nrow = min(RC, R - rb);
        __syncthreads();
        for (int i = t; i < nrow * CB; i += NT) { const int r = i / CB, cc = i - r * CB; sg[i] = cc < nc ? dz[(size_t)(rb + r) * N + n0 + cc] : 0.f; }
        {   // K % 4 == 0 (tl_check): 16-byte loads, all of a thread's loads in flight before the first LDS store
            const int n4 = nrow * K / 4;
            const float4* x4 = (const float4*)(x + (size_t)rb * K);
            const float4* a4 = xadd ? (const float4*)(xadd + (size_t)rb * K) : nullptr;
            for (int base = t; base < n4; base += NT * 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = base + u * NT;
                    if (i < n4) {
                        v[u] = x4[i];
                        if (a4) { const float4 w = a4[i]; v[u].x += w.x; v[u].y += w.y; v[u].z += w.z; v[u].w += w.w; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = base + u * NT; if (i < n4) ((float4*)sx)[i] = v[u]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
            const int k = c + ki * cols;
            if (k >= K) continue;
#pragma unroll
            for (int cj = 0; cj < 2; ++cj) {
                const int ci = h + cj * groups;
                if (ci >= CB) continue;
                float a = 0.f;
#pragma unroll 8
                for (int r = 0; r < nrow; ++r) a += sg[r * CB + ci] * sx[r * K + k];
                acc[ki][cj] += a;
            }
        }
        if (t < 64) {
            // column sums (db, dgamma, dbeta): the first wave, lane = (row group, column) -- 16 row groups, so the dependent global loads of
            // the LayerNorm terms form chains of 2 instead of 32 (four threads walking all rows serially cost ~10 us per LayerNorm layer)
            const int cc = t & (CB - 1);
            for (int r = t / CB; r < nrow; r += 64 / CB) {
                if (cc < nc) {
                    sb += sg[r * CB + cc];
                    if (gamma) {
                        const float d = dy[(size_t)(rb + r) * N + n0 + cc];
                        ag += d * (z[(size_t)(rb + r) * N + n0 + cc] - rstat[2 * (rb + r)]) * rstat[2 * (rb + r) + 1];
                        ab += d;
                    }
                }
            }
        }
    }
    if (t < 64) {
#pragma unroll
        for (int o = CB; o < 64; o <<= 1) { sb += __shfl_xor(sb, o, 64); ag += __shfl_xor(ag, o, 64); ab += __shfl_xor(ab, o, 64); }
    }
#pragma unroll
    for (int ki = 0; ki < 2; ++ki)
#pragma unroll
        for (int cj = 0; cj < 2; ++cj) {
            const int k = c + ki * cols, ci = h + cj * groups;
            if (k < K && ci < nc) dW[wt ? (size_t)k * N + n0 + ci : (size_t)(n0 + ci) * K + k] = acc[ki][cj];
        }
    if (t < nc) {
        if (db) db[n0 + t] = sb;
        if (gamma) { dgamma[n0 + t] = ag; dbeta[n0 + t] = ab; }
    }
}

// The main backward launch: workgroups [0, nrb) produce dx (row-parallel), the rest dW / db / dgamma / dbeta (column-parallel) -- both halves
// only read dz (token_linear_dz_kernel), so they share the chip instead of running back to back (two ~12 us latency-bound launches before).
__global__ __launch_bounds__(NT) void token_linear_bwd_main_kernel(int nrb, const float* __restrict__ dz, const float* __restrict__ dy,
                                                                   const float* __restrict__ x, const float* __restrict__ xadd, const float* __restrict__ W,
                                                                   const float* __restrict__ gamma, const float* __restrict__ z,
                                                                   const float* __restrict__ rstat, float* __restrict__ dx, float* __restrict__ dW,
                                                                   float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta, int R, int K, int N,
                                                                   int wt) {
    extern __shared__ float sm[];
    if ((int)blockIdx.x < nrb) token_linear_bwd_rows(sm, blockIdx.x, dz, W, dx, R, K, N, wt);
    else token_linear_bwd_cols(sm, blockIdx.x - nrb, dz, dy, x, xadd, gamma, z, rstat, dW, db, dgamma, dbeta, R, K, N, wt);
}

__global__ __launch_bounds__(NT) void token_linear_multi_bwd_main_kernel(const TokLinSet set) {
    extern __shared__ float sm[];
    const TokLin& p = set.op[blockIdx.y];
    const int nrb = p.dx ? (p.R + RB - 1) / RB : 0;
    const int ncb = (p.N + CB - 1) / CB;
    if ((int)blockIdx.x >= nrb + ncb) return;
    if ((int)blockIdx.x < nrb) {
        token_linear_bwd_rows(sm, blockIdx.x, p.dz, p.W, p.dx, p.R, p.K, p.N, p.wt);
        if (p.dx_pair > 0) {                                  // a second layer reads the same x: its contribution is added here, in a fixed order
            const TokLin& q = set.op[p.dx_pair - 1];
            __syncthreads();                                  // the LDS tiles of the first pass are free
            token_linear_bwd_rows<true>(sm, blockIdx.x, q.dz, q.W, p.dx, p.R, p.K, q.N, q.wt);
        }
    } else token_linear_bwd_cols(sm, blockIdx.x - nrb, p.dz, p.dy, p.x, p.xadd, p.gamma, p.z, p.rstat, p.dW, p.db, p.dgamma, p.dbeta, p.R, p.K, p.N, p.wt);
}

// ---- token self-attention core: one workgroup per batch element, T <= 16 tokens, D <= 256 -------------------------------------------------
__global__ __launch_bounds__(NT) void token_sa_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                          const unsigned char* __restrict__ pad, float scale, int T, int D, float* __restrict__ out,
                                                          float* __restrict__ prob) {
    __shared__ float sp[16 * 16];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* qb = q + (size_t)b * T * D; const float* kb = k + (size_t)b * T * D; const float* vb = v + (size_t)b * T * D;
    if (t < T * T) {
        const int i = t / T, j = t - i * T;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += qb[i * D + d] * kb[j * D + d];
        s *= scale;
        if (pad && pad[b * T + j]) s = -INFINITY;
        sp[i * 16 + j] = s;
    }
    __syncthreads();
    if (t < T) {
        float m = -INFINITY;
        for (int j = 0; j < T; ++j) m = fmaxf(m, sp[t * 16 + j]);
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { const float e = __expf(sp[t * 16 + j] - m); sp[t * 16 + j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < T; ++j) { sp[t * 16 + j] *= inv; prob[((size_t)b * T + t) * T + j] = sp[t * 16 + j]; }
    }
    __syncthreads();
    for (int i = t; i < T * D; i += NT) {
        const int r = i / D, d = i - r * D;
        float a = 0.f;
        for (int j = 0; j < T; ++j) a += sp[r * 16 + j] * vb[j * D + d];
        out[(size_t)b * T * D + i] = a;
    }
}

__global__ __launch_bounds__(NT) void token_sa_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ prob, float scale, int T, int D,
                                                          float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv) {
    __shared__ float sp[16 * 16], sds[16 * 16];
    const int b = blockIdx.x, t = threadIdx.x;
    const size_t o = (size_t)b * T * D;
    if (t < T * T) {
        const int i = t / T, j = t - i * T;
        sp[i * 16 + j] = prob[((size_t)b * T + i) * T + j];
        float a = 0.f;                                               // dP[i][j] = dout[i] . v[j]
        for (int d = 0; d < D; ++d) a += dout[o + i * D + d] * v[o + j * D + d];
        sds[i * 16 + j] = a;
    }
    __syncthreads();
    if (t < T) {                                                     // softmax backward per row, then the 1/sqrt(d) factor
        float dot = 0.f;
        for (int j = 0; j < T; ++j) dot += sp[t * 16 + j] * sds[t * 16 + j];
        for (int j = 0; j < T; ++j) sds[t * 16 + j] = sp[t * 16 + j] * (sds[t * 16 + j] - dot) * scale;
    }
    __syncthreads();
    for (int i = t; i < T * D; i += NT) {
        const int r = i / D, d = i - r * D;
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int j = 0; j < T; ++j) {
            aq += sds[r * 16 + j] * k[o + j * D + d];                 // dq[r] = sum_j dS[r][j] k[j]
            ak += sds[j * 16 + r] * q[o + j * D + d];                 // dk[r] = sum_i dS[i][r] q[i]
            av += sp[j * 16 + r] * dout[o + j * D + d];               // dv[r] = sum_i P[i][r] dout[i]
        }
        dq[o + i] = aq; dk[o + i] = ak; dv[o + i] = av;
    }
}


// The same two kernels with their operands in LDS (round 5; T <= 16, D <= 128): the forms above read q / k / v / dout straight from global memory
// inside run-time-length loops -- `for d < D: s += q[i][d] * k[j][d]` is two loads and a full wait per d, 128 dependent round trips for one score,
// and the output loops another T per element (18 / 21 us per launch for 40 x 128 operands). Here the (T x D) matrices of the batch element are
// staged once, up to 8 predicated 4-byte loads per matrix and thread in flight, rows at pitch D + 1 (the (i, j) lanes of a wave read D-strided rows: at
// pitch 128 all on one bank); every sum keeps its order (d ascending, j ascending), so the results have the bits of the first forms.
constexpr int SA_MAXT = 16, SA_MAXD = 128, SA_P = SA_MAXD + 1;
template <int N>
__device__ __forceinline__ void sa_stage(float* const (&dst)[N], const float* const (&src)[N], int T, int D) {     // all N matrices' loads, then the stores
    float v[N][8];
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = threadIdx.x + u * NT; v[m][u] = i < T * D ? src[m][i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = threadIdx.x + u * NT;
        if (i < T * D) {
            const int r = i / D, c = i - r * D;
#pragma unroll
            for (int m = 0; m < N; ++m) dst[m][r * SA_P + c] = v[m][u];
        }
    }
}

__global__ __launch_bounds__(NT) void token_sa_fwd_lds_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                              const unsigned char* __restrict__ pad, float scale, int T, int D, float* __restrict__ out,
                                                              float* __restrict__ prob) {
    __shared__ float sp[16 * 16];
    __shared__ float sq[SA_MAXT * SA_P], sk[SA_MAXT * SA_P], sv[SA_MAXT * SA_P];
    const int b = blockIdx.x, t = threadIdx.x;
    {
        float* const dst[3] = {sq, sk, sv};
        const float* const src[3] = {q + (size_t)b * T * D, k + (size_t)b * T * D, v + (size_t)b * T * D};
        sa_stage<3>(dst, src, T, D);
    }
    const bool padded = (t < T * T && pad) ? pad[b * T + (t % T)] != 0 : false;
    __syncthreads();
    if (t < T * T) {
        const int i = t / T, j = t - i * T;
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) s += sq[i * SA_P + d] * sk[j * SA_P + d];
        s *= scale;
        if (padded) s = -INFINITY;
        sp[i * 16 + j] = s;
    }
    __syncthreads();
    if (t < T) {
        float m = -INFINITY;
        for (int j = 0; j < T; ++j) m = fmaxf(m, sp[t * 16 + j]);
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { const float e = __expf(sp[t * 16 + j] - m); sp[t * 16 + j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < T; ++j) { sp[t * 16 + j] *= inv; prob[((size_t)b * T + t) * T + j] = sp[t * 16 + j]; }
    }
    __syncthreads();
    for (int i = t; i < T * D; i += NT) {
        const int r = i / D, d = i - r * D;
        float a = 0.f;
        for (int j = 0; j < T; ++j) a += sp[r * 16 + j] * sv[j * SA_P + d];
        out[(size_t)b * T * D + i] = a;
    }
}

__global__ __launch_bounds__(NT) void token_sa_bwd_lds_kernel(const float* __restrict__ dout, const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ prob, float scale, int T, int D,
                                                              float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv) {
    __shared__ float sp[16 * 16], sds[16 * 16];
    __shared__ float sq[SA_MAXT * SA_P], sk[SA_MAXT * SA_P], sv[SA_MAXT * SA_P], sg[SA_MAXT * SA_P];
    const int b = blockIdx.x, t = threadIdx.x;
    const size_t o = (size_t)b * T * D;
    {
        float* const dst[4] = {sg, sv, sq, sk};
        const float* const src[4] = {dout + o, v + o, q + o, k + o};
        sa_stage<4>(dst, src, T, D);
    }
    const float pr = t < T * T ? prob[((size_t)b * T + t / T) * T + (t % T)] : 0.f;
    __syncthreads();
    if (t < T * T) {
        const int i = t / T, j = t - i * T;
        sp[i * 16 + j] = pr;
        float a = 0.f;                                               // dP[i][j] = dout[i] . v[j]
#pragma unroll 8
        for (int d = 0; d < D; ++d) a += sg[i * SA_P + d] * sv[j * SA_P + d];
        sds[i * 16 + j] = a;
    }
    __syncthreads();
    if (t < T) {                                                     // softmax backward per row, then the 1/sqrt(d) factor
        float dot = 0.f;
        for (int j = 0; j < T; ++j) dot += sp[t * 16 + j] * sds[t * 16 + j];
        for (int j = 0; j < T; ++j) sds[t * 16 + j] = sp[t * 16 + j] * (sds[t * 16 + j] - dot) * scale;
    }
    __syncthreads();
    for (int i = t; i < T * D; i += NT) {
        const int r = i / D, d = i - r * D;
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int j = 0; j < T; ++j) {
            aq += sds[r * 16 + j] * sk[j * SA_P + d];                 // dq[r] = sum_j dS[r][j] k[j]
            ak += sds[j * 16 + r] * sq[j * SA_P + d];                 // dk[r] = sum_i dS[i][r] q[i]
            av += sp[j * 16 + r] * sg[j * SA_P + d];                  // dv[r] = sum_i P[i][r] dout[i]
        }
        dq[o + i] = aq; dk[o + i] = ak; dv[o + i] = av;
    }
}


// ---- mask pre-processing of the instance matte decoder (instance_matte_decoder.py:131-153 + utils.py:16-21 resizeAnyShape(use_avg_pool_binary)):
// per OS8 cell: the guidance masks average-pooled and thresholded (> 0), the instance-ID position = max_i (i + 1) * m8_i, which instance slots
// have any mask pixel (token validity), and -- training -- the ground-truth guidance (max-pool of the alphas > 0). ~15 small torch launches
// (pool, compare, cast, multiply, amax, cat, permute copies ...) per step in one.
__global__ __launch_bounds__(NT) void imd_prep_kernel(const float* __restrict__ mask, int n_in, int s, const float* __restrict__ gt, int n_gt, int gs,
                                                      int B, int NF, int h, int w, int n_i, int32_t* __restrict__ feat_ids,
                                                      float* __restrict__ guidance, unsigned char* __restrict__ valid) {
    const long cells = (long)B * NF * h * w;
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= cells) return;
    const int x = (int)(i % w); long r = i / w; const int y = (int)(r % h); r /= h; const int f = (int)(r % NF); const int b = (int)(r / NF);
    const int Hm = h * s, Wm = w * s;
    int id = 0;
    for (int k = 0; k < n_in; ++k) {
        const float* mp = mask + (((long)(b * NF + f) * n_in + k) * Hm + (long)y * s) * Wm + (long)x * s;
        float acc = 0.f;
        for (int dy = 0; dy < s; ++dy)
            for (int dx = 0; dx < s; ++dx) acc += mp[(long)dy * Wm + dx];
        if (acc / (float)(s * s) > 0.f) { id = k + 1; valid[b * n_i + k] = 1; }        // (m8 * ids).amax(2): the largest set instance index
    }
    const long L = (long)NF * h * w, l = ((long)f * h + y) * w + x;
    feat_ids[b * L + l] = id;
    if (guidance) {
        const int H = h * gs, W = w * gs;
        for (int k = 0; k < n_i; ++k) {
            float m = 0.f;
            if (k < n_gt) {
                const float* gp = gt + (((long)(b * NF + f) * n_gt + k) * H + (long)y * gs) * W + (long)x * gs;
                m = gp[0];
                if ((gs & 3) == 0) {                             // 16-byte reads (x * gs * 4 bytes is 16-byte aligned): 8 x 2 loads per cell instead of 64
                    for (int dy = 0; dy < gs; ++dy)
                        for (int dx = 0; dx < gs; dx += 4) {
                            const float4 v = *(const float4*)(gp + (long)dy * W + dx);
                            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                        }
                } else {
                    for (int dy = 0; dy < gs; ++dy)
                        for (int dx = 0; dx < gs; ++dx) m = fmaxf(m, gp[(long)dy * W + dx]);
                }
            }
            guidance[((long)b * n_i + k) * L + l] = m > 0.f ? 1.f : 0.f;
        }
    }
}

// The same in a (cells, 1 + n_i) grid (round 5): blockIdx.y == 0 forms the instance-ID position and the token validity, blockIdx.y == 1 + k the
// ground-truth guidance of slot k. The single-thread-per-cell form walked n_in + n_gt windows one after the other with run-time loop bounds -- one
// memory round trip per load, ~170 of them per thread, on 64 workgroups (57 us at 64 x 64 cells). Here a thread owns one (cell, plane) pair and its
// window is a compile-time GS x GS block loaded as one batch (GS = 8: sixteen 16-byte loads); the ID position batches the n_in <= 16 pooled values of a
// cell the same way (S == 1: the guidance masks usually arrive at the cell resolution). Same values, same comparisons: bit-identical outputs.
template <int S, int GS>
__global__ __launch_bounds__(NT) void imd_prep_planes_kernel(const float* __restrict__ mask, int n_in, const float* __restrict__ gt, int n_gt,
                                                             int B, int NF, int h, int w, int n_i, int32_t* __restrict__ feat_ids,
                                                             float* __restrict__ guidance, unsigned char* __restrict__ valid) {
    const long cells = (long)B * NF * h * w;
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= cells) return;
    const int x = (int)(i % w); long r = i / w; const int y = (int)(r % h); r /= h; const int f = (int)(r % NF); const int b = (int)(r / NF);
    const long L = (long)NF * h * w, l = ((long)f * h + y) * w + x;
    if (blockIdx.y == 0) {
        constexpr int MAXP = 16;
        const int Hm = h * S, Wm = w * S;
        float acc[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const float* mp = mask + (((long)(b * NF + f) * n_in + min(k, n_in - 1)) * Hm + (long)y * S) * Wm + (long)x * S;
            float a = 0.f;
#pragma unroll
            for (int dy = 0; dy < S; ++dy)
#pragma unroll
                for (int dx = 0; dx < S; ++dx) a += mp[(long)dy * Wm + dx];
            acc[k] = a;
        }
        int id = 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k)
            if (k < n_in && acc[k] / (float)(S * S) > 0.f) { id = k + 1; valid[b * n_i + k] = 1; }
        feat_ids[b * L + l] = id;
        return;
    }
    const int k = (int)blockIdx.y - 1;
    float m = 0.f;
    if (k < n_gt) {
        const int H = h * GS, W = w * GS;
        const float* gp = gt + (((long)(b * NF + f) * n_gt + k) * H + (long)y * GS) * W + (long)x * GS;
        float4 v[GS][GS / 4];
#pragma unroll
        for (int dy = 0; dy < GS; ++dy)
#pragma unroll
            for (int dx = 0; dx < GS / 4; ++dx) v[dy][dx] = *(const float4*)(gp + (long)dy * W + dx * 4);
        m = v[0][0].x;
#pragma unroll
        for (int dy = 0; dy < GS; ++dy)
#pragma unroll
            for (int dx = 0; dx < GS / 4; ++dx) m = fmaxf(fmaxf(m, fmaxf(v[dy][dx].x, v[dy][dx].y)), fmaxf(v[dy][dx].z, v[dy][dx].w));
    }
    guidance[((long)b * n_i + k) * L + l] = m > 0.f ? 1.f : 0.f;
}

// ---- einsum('bqc,blc->blq') of the instance matte decoder (instance_matte_decoder.py:296-299: logits of every OS8 pixel against the Q = 10
// instance tokens of its batch element; C = output_dim = 32). The reference runs it as one einsum; round 2 ran it as one 1x1 convolution PER
// batch element (weights = that element's tokens: 4 fprop + 4 dgrad + 4 wgrad + 4 reduce launches, 4 weight conversions and a stack per
// step). One launch each way here: a thread owns a pixel row (C values in registers), the tokens of its batch element sit in LDS.
template <typename T, int C>
__global__ __launch_bounds__(NT) void token_einsum_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ tok, int L, int Q, int QP,
                                                              T* __restrict__ out) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    __shared__ float st[16 * C];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < Q * C; i += NT) st[i] = TR::rnd(tok[(size_t)b * Q * C + i]);      // tokens in the compute dtype, like the conv they replace
    __syncthreads();
    const int l = blockIdx.x * NT + threadIdx.x;
    if (l >= L) return;
    const T* fp = feat + ((size_t)b * L + l) * C;
    float f[C];
#pragma unroll
    for (int k = 0; k < C / CE; ++k) TR::unpack(*(const uint4*)(fp + k * CE), f + k * CE);
    T* op = out + ((size_t)b * L + l) * QP;
    float o[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float a = 0.f;
        if (q < Q) {
#pragma unroll
            for (int c = 0; c < C; ++c) a += f[c] * st[q * C + c];
        }
        o[q] = a;
    }
    for (int k = 0; k < QP / CE; ++k) *(uint4*)(op + k * CE) = TR::pack(o + k * CE);
}

// backward: dfeat[l][c] = sum_q dlog[l][q] tok[q][c];  dtok[q][c] += sum_l dlog[l][q] feat[l][c]. The token gradient is a (Q x C) outer-product
// sum over the block's 256 rows: the rows' g (16 values) and f (C values) go through LDS tiles of 64 rows, thread (q, c) pairs accumulate
// over the tile rows (a first version reduced every (q, c) product with a wave butterfly: 320 x 6 shuffles per thread, 164 us per launch).
template <typename T, int C>
__global__ __launch_bounds__(NT) void token_einsum_bwd_kernel(const T* __restrict__ dlog, const T* __restrict__ feat, const float* __restrict__ tok, int L, int Q,
                                                              int QP, T* __restrict__ dfeat, float* __restrict__ dtok, float* __restrict__ slots) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    constexpr int TR_ROWS = 64;
    __shared__ float st[16 * C];
    __shared__ float sg[TR_ROWS][17];
    __shared__ float sf[TR_ROWS][C + 1];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 16 * C; i += NT) st[i] = i < Q * C ? TR::rnd(tok[(size_t)b * Q * C + i]) : 0.f;
    __syncthreads();
    const int l = blockIdx.x * NT + threadIdx.x;
    const bool live = l < L;
    float g[16], f[C];
#pragma unroll
    for (int q = 0; q < 16; ++q) g[q] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) f[c] = 0.f;
    if (live) {
        const T* gp = dlog + ((size_t)b * L + l) * QP;
        for (int k = 0; k < QP / CE; ++k) TR::unpack(*(const uint4*)(gp + k * CE), g + k * CE);
        const T* fp = feat + ((size_t)b * L + l) * C;
#pragma unroll
        for (int k = 0; k < C / CE; ++k) TR::unpack(*(const uint4*)(fp + k * CE), f + k * CE);
        float d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) a += g[q] * st[q * C + c];      // rows q >= Q of st are zero
            d[c] = a;
        }
        T* dp = dfeat + ((size_t)b * L + l) * C;
#pragma unroll
        for (int k = 0; k < C / CE; ++k) *(uint4*)(dp + k * CE) = TR::pack(d + k * CE);
    }
    // dtok: thread t -> column c = t % C, token group qg = t / C (NT / C groups), tokens qg, qg + NT / C, ...
    constexpr int QG = NT / C;                                   // 8 (C = 32) or 4 (C = 64)
    constexpr int QPT = (16 + QG - 1) / QG;                      // tokens per thread: 2 or 4
    const int c = threadIdx.x % C, qg = threadIdx.x / C;
    float acc[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) acc[j] = 0.f;
    for (int tile = 0; tile < NT / TR_ROWS; ++tile) {
        __syncthreads();
        const int r = threadIdx.x - tile * TR_ROWS;
        if (r >= 0 && r < TR_ROWS) {                             // the 64 threads whose rows form this tile publish them
#pragma unroll
            for (int q = 0; q < 16; ++q) sg[r][q] = g[q];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) sf[r][cc] = f[cc];
        }
        __syncthreads();
        for (int rr = 0; rr < TR_ROWS; ++rr) {
            const float fv = sf[rr][c];
#pragma unroll
            for (int j = 0; j < QPT; ++j) { const int q = qg + j * QG; if (q < 16) acc[j] += sg[rr][q] * fv; }
        }
    }
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const int q = qg + j * QG;
        if (q < Q) {
            if (slots) slots[(((size_t)b * gridDim.x + blockIdx.x) * Q + q) * C + c] = acc[j];      // deterministic mode: one row [Q][C] per workgroup (csrc/det.hip)
            else atomicAdd(&dtok[((size_t)b * Q + q) * C + c], acc[j]);
        }
    }
}

}  // namespace

static int tl_check(int R, int K, int N) {
    if (R <= 0 || K <= 0 || K > 256 || N <= 0 || N > 256 || (K & 3)) return -3;
    return 0;
}

extern "C" int mg_token_linear_fwd_ex(const float* x, const float* xadd, const float* W, const float* bias, const float* res, int relu, const float* gamma,
                                      const float* beta, float eps, float* y, float* z, float* rstat, int R, int K, int N, int wt, void* stream);
extern "C" int mg_token_linear_fwd(const float* x, const float* xadd, const float* W, const float* bias, const float* res, int relu, const float* gamma,
                                   const float* beta, float eps, float* y, float* z, float* rstat, int R, int K, int N, void* stream) {
    return mg_token_linear_fwd_ex(x, xadd, W, bias, res, relu, gamma, beta, eps, y, z, rstat, R, K, N, 0, stream);
}
extern "C" int mg_token_linear_fwd_ex(const float* x, const float* xadd, const float* W, const float* bias, const float* res, int relu, const float* gamma,
                                      const float* beta, float eps, float* y, float* z, float* rstat, int R, int K, int N, int wt, void* stream) {
    int rc = tl_check(R, K, N); if (rc) return rc;
    if (wt && (N & 3)) return -3;
    if (gamma && (!beta || !z || !rstat)) return -2;
    const size_t lds = ((size_t)RB * (K + N) + (size_t)K * (N + 1)) * sizeof(float);
    if (lds > 150 * 1024) return -3;
    static bool attr_f = false;
    if (!attr_f) { (void)hipFuncSetAttribute((const void*)token_linear_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_f = true; }
    hipLaunchKernelGGL(token_linear_fwd_kernel, dim3((R + RB - 1) / RB), dim3(NT), lds, (hipStream_t)stream, x, xadd, W, bias, res, relu, gamma, beta, eps, y, z,
                       rstat, R, K, N, wt);
    MG_CHECK_LAUNCH();
    return 0;
}

// `dz` [R,N] scratch = gradient w.r.t. the linear output (after LayerNorm backward and the ReLU mask); dres may alias nothing (NULL when absent)
extern "C" int mg_token_linear_bwd_ex(const float* dy, const float* x, const float* xadd, const float* W, const float* yout, int relu, const float* gamma,
                                      const float* z, const float* rstat, float* dx, float* dW, float* db, float* dres, float* dgamma, float* dbeta,
                                      float* dz, int R, int K, int N, int wt, void* stream);
extern "C" int mg_token_linear_bwd(const float* dy, const float* x, const float* xadd, const float* W, const float* yout, int relu, const float* gamma,
                                   const float* z, const float* rstat, float* dx, float* dW, float* db, float* dres, float* dgamma, float* dbeta,
                                   float* dz, int R, int K, int N, void* stream) {
    return mg_token_linear_bwd_ex(dy, x, xadd, W, yout, relu, gamma, z, rstat, dx, dW, db, dres, dgamma, dbeta, dz, R, K, N, 0, stream);
}
extern "C" int mg_token_linear_bwd_ex(const float* dy, const float* x, const float* xadd, const float* W, const float* yout, int relu, const float* gamma,
                                      const float* z, const float* rstat, float* dx, float* dW, float* db, float* dres, float* dgamma, float* dbeta,
                                      float* dz, int R, int K, int N, int wt, void* stream) {
    int rc = tl_check(R, K, N); if (rc) return rc;
    if (wt && (N & 3)) return -3;
    if (gamma && (!z || !rstat || !dgamma || !dbeta)) return -2;
    if ((relu && !yout) || !dz) return -2;
    const size_t lds_r = ((size_t)RB * N + (size_t)N * K) * sizeof(float);
    const size_t lds_c = (size_t)32 * (CB + K) * sizeof(float);
    if (lds_r > 150 * 1024) return -3;
    static bool attr_b = false;
    if (!attr_b) { (void)hipFuncSetAttribute((const void*)token_linear_bwd_main_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_b = true; }
    hipStream_t st = (hipStream_t)stream;
    // a plain linear layer (no LayerNorm, no ReLU) has dz == dy: the caller passes dz = dy (and takes dy as the residual gradient) and the dz
    // kernel is skipped
    if (gamma || relu || dz != dy || dres)
        hipLaunchKernelGGL(token_linear_dz_kernel, dim3((R + 3) / 4), dim3(NT), 0, st, dy, yout, relu, gamma, z, rstat, dz, dres, R, N);
    const int nrb = dx ? (R + RB - 1) / RB : 0;
    hipLaunchKernelGGL(token_linear_bwd_main_kernel, dim3(nrb + (N + CB - 1) / CB), dim3(NT), dx ? (lds_r > lds_c ? lds_r : lds_c) : lds_c, st, nrb,
                       (const float*)dz, dy, x, xadd, W, gamma, z, rstat, dx, dW, db, dgamma, dbeta, R, K, N, wt);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_token_sa_fwd(const float* q, const float* k, const float* v, const unsigned char* pad, float scale, int B, int T, int D, float* out,
                               float* prob, void* stream) {
    if (B <= 0) return 0;
    if (T <= 0 || T > 16 || D <= 0) return -3;
    static const int lds_on = [] { const char* e = getenv("MG_TOKEN_SA_LDS"); return e ? atoi(e) : 1; }();
    if (lds_on && T <= SA_MAXT && D <= SA_MAXD && T * D <= 8 * NT)
        hipLaunchKernelGGL(token_sa_fwd_lds_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, q, k, v, pad, scale, T, D, out, prob);
    else
        hipLaunchKernelGGL(token_sa_fwd_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, q, k, v, pad, scale, T, D, out, prob);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_token_sa_bwd(const float* dout, const float* q, const float* k, const float* v, const float* prob, float scale, int B, int T, int D,
                               float* dq, float* dk, float* dv, void* stream) {
    if (B <= 0) return 0;
    if (T <= 0 || T > 16 || D <= 0) return -3;
    static const int lds_on = [] { const char* e = getenv("MG_TOKEN_SA_LDS"); return e ? atoi(e) : 1; }();
    if (lds_on && T <= SA_MAXT && D <= SA_MAXD && T * D <= 8 * NT)
        hipLaunchKernelGGL(token_sa_bwd_lds_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, dout, q, k, v, prob, scale, T, D, dq, dk, dv);
    else
        hipLaunchKernelGGL(token_sa_bwd_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, dout, q, k, v, prob, scale, T, D, dq, dk, dv);
    MG_CHECK_LAUNCH();
    return 0;
}

/* logits[b][l][q] = sum_c feat[b][l][c] * tok[b][q][c] for q < Q, zero for Q <= q < QP (QP = 16: the row pitch of the NHWC logits the up-sampling
 * kernel reads). feat / out / dlog / dfeat in `dtype`, tok / dtok fp32 ([B][Q][C]; dtok is zeroed here). C = 32, Q <= 16. */
extern "C" int mg_token_einsum_fwd(const void* feat, int dtype, const float* tok, int B, int L, int C, int Q, int QP, void* out, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if ((C != 32 && C != 64) || Q < 1 || Q > 16 || QP != 16) return -3;
    if (!MG_IS16(dtype) && dtype != MG_F32) return -6;
    dim3 grid((L + NT - 1) / NT, B);
    hipStream_t st = (hipStream_t)stream;
#define EINSUM_FWD(T, CC) hipLaunchKernelGGL((token_einsum_fwd_kernel<T, CC>), grid, dim3(NT), 0, st, (const T*)feat, tok, L, Q, QP, (T*)out)
    if (dtype == MG_BF16) { if (C == 32) EINSUM_FWD(bf16raw, 32); else EINSUM_FWD(bf16raw, 64); }
    else if (dtype == MG_F16) { if (C == 32) EINSUM_FWD(f16raw, 32); else EINSUM_FWD(f16raw, 64); }
    else { if (C == 32) EINSUM_FWD(float, 32); else EINSUM_FWD(float, 64); }
#undef EINSUM_FWD
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_token_einsum_bwd(const void* dlog, const void* feat, int dtype, const float* tok, int B, int L, int C, int Q, int QP, void* dfeat,
                                   float* dtok, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if ((C != 32 && C != 64) || Q < 1 || Q > 16 || QP != 16) return -3;
    if (!MG_IS16(dtype) && dtype != MG_F32) return -6;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = mg_zero_words(dtok, (long)B * Q * C, st);
    if (e != hipSuccess) return (int)e;
    dim3 grid((L + NT - 1) / NT, B);
    float* slots = nullptr;
    if (mg_det_on && grid.x > 1) { slots = mg_det_scratch((long)B * grid.x * Q * C); if (!slots) return MG_DET_NO_SCRATCH; }
#define EINSUM_BWD(T, CC) hipLaunchKernelGGL((token_einsum_bwd_kernel<T, CC>), grid, dim3(NT), 0, st, (const T*)dlog, (const T*)feat, tok, L, Q, QP, (T*)dfeat, dtok, slots)
    if (dtype == MG_BF16) { if (C == 32) EINSUM_BWD(bf16raw, 32); else EINSUM_BWD(bf16raw, 64); }
    else if (dtype == MG_F16) { if (C == 32) EINSUM_BWD(f16raw, 32); else EINSUM_BWD(f16raw, 64); }
    else { if (C == 32) EINSUM_BWD(float, 32); else EINSUM_BWD(float, 64); }
#undef EINSUM_BWD
    MG_CHECK_LAUNCH();
    if (slots) {
        mg_det_seg sg{dtok, Q * C, (long)Q * C};
        return mg_det_reduce(slots, (int)grid.x, B, Q * C, 0, &sg, 1, st);
    }
    return 0;
}

/* Mask pre-processing of the instance matte decoder: mask [B][NF][n_in][h*s][w*s] fp32 (guidance masks at s times the OS8 resolution), gt (or NULL)
 * [B][NF][n_gt][h*gs][w*gs] fp32 alphas -> feat_ids int32 [B][NF*h*w] (max_i (i+1) * [avg-pooled mask_i > 0]), valid uint8 [B][n_i] (zeroed here; 1 where
 * instance slot i has any mask pixel), guidance fp32 [B][n_i][NF*h*w] ([max-pooled alpha_i > 0], slots >= n_gt zero; only with gt). */
extern "C" int mg_imd_prep(const float* mask, int n_in, int s, const float* gt, int n_gt, int gs, int B, int NF, int h, int w, int n_i, int32_t* feat_ids,
                           float* guidance, unsigned char* valid, void* stream) {
    if (B <= 0 || NF <= 0 || h <= 0 || w <= 0) return 0;
    if (n_in < 0 || n_in > n_i || s < 1 || (gt && (gs < 1 || n_gt < 0))) return -2;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = mg_zero_words(valid, ((long)B * n_i + 3) / 4, st);          // `valid` must be padded to a multiple of 4 bytes by the caller
    if (e != hipSuccess) return (int)e;
    const long cells = (long)B * NF * h * w;
    static const int planes_on = [] { const char* e = getenv("MG_IMD_PREP_PLANES"); return e ? atoi(e) : 1; }();
    if (planes_on && gt && guidance && gs == 8 && n_in >= 1 && n_in <= 16 && (s == 1 || s == 2)) {
        const dim3 grid((unsigned)((cells + NT - 1) / NT), (unsigned)(1 + n_i));
        if (s == 1) hipLaunchKernelGGL((imd_prep_planes_kernel<1, 8>), grid, dim3(NT), 0, st, mask, n_in, gt, n_gt, B, NF, h, w, n_i, feat_ids, guidance, valid);
        else hipLaunchKernelGGL((imd_prep_planes_kernel<2, 8>), grid, dim3(NT), 0, st, mask, n_in, gt, n_gt, B, NF, h, w, n_i, feat_ids, guidance, valid);
        MG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(imd_prep_kernel, dim3((unsigned)((cells + NT - 1) / NT)), dim3(NT), 0, st, mask, n_in, s, gt, gt ? n_gt : 0, gs, B, NF, h, w, n_i,
                       feat_ids, gt ? guidance : nullptr, valid);
    MG_CHECK_LAUNCH();
    return 0;
}

/* Up to 6 INDEPENDENT token linear layers in one launch each way (same semantics per layer as mg_token_linear_fwd_ex / _bwd_ex). `ops`: array of
 * mg_tok_lin (include/maggie_hip.h); forward reads x, xadd, W, bias, res, relu, gamma, beta, eps, wt and writes y (z, rstat with a LayerNorm);
 * backward reads dy (+ the forward's x, xadd, W, yout, gamma, z, rstat) and writes dx, dW, db, dres, dgamma, dbeta through the scratch dz. */
static int tok_set(const mg_tok_lin* ops, int n, TokLinSet* set, int bwd) {
    if (!ops || n < 1 || n > TOK_MULTI) return -2;
    set->n = n;
    for (int i = 0; i < n; ++i) {
        const mg_tok_lin& o = ops[i];
        int rc = tl_check(o.R, o.K, o.N); if (rc) return rc;
        if (o.wt && (o.N & 3)) return -3;
        if (o.gamma && ((!bwd && !o.beta) || !o.z || !o.rstat)) return -2;
        if (bwd && ((o.relu && !o.yout) || !o.dz || !o.dW || (o.gamma && (!o.dgamma || !o.dbeta)))) return -2;
        TokLin& t = set->op[i];
        t.x = o.x; t.xadd = o.xadd; t.W = o.W; t.bias = o.bias; t.res = o.res; t.gamma = o.gamma; t.beta = o.beta;
        t.y = o.y; t.z = o.z; t.rstat = o.rstat; t.dy = o.dy; t.yout = o.yout; t.dx = o.dx; t.dW = o.dW; t.db = o.db; t.dres = o.dres;
        t.dgamma = o.dgamma; t.dbeta = o.dbeta; t.dz = o.dz; t.R = o.R; t.K = o.K; t.N = o.N; t.relu = o.relu; t.wt = o.wt; t.eps = o.eps;
        t.dx_pair = 0;
        if (bwd && o.dx_pair) {
            const int j = o.dx_pair - 1;
            if (j < 0 || j >= n || j == i || !o.dx || ops[j].dx || ops[j].dx_pair || ops[j].R != o.R || ops[j].K != o.K || o.xadd || ops[j].xadd) return -2;
            t.dx_pair = o.dx_pair;
        }
    }
    return 0;
}

extern "C" int mg_token_linear_multi_fwd(const mg_tok_lin* ops, int n, void* stream) {
    TokLinSet set;
    int rc = tok_set(ops, n, &set, 0); if (rc) return rc;
    size_t lds = 0; int gx = 0;
    for (int i = 0; i < n; ++i) {
        const size_t l = ((size_t)RB * (ops[i].K + ops[i].N) + (size_t)ops[i].K * (ops[i].N + 1)) * sizeof(float);
        if (l > lds) lds = l;
        const int g = (ops[i].R + RB - 1) / RB; if (g > gx) gx = g;
    }
    if (lds > 150 * 1024) return -3;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)token_linear_multi_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    hipLaunchKernelGGL(token_linear_multi_fwd_kernel, dim3(gx, n), dim3(NT), lds, (hipStream_t)stream, set);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_token_linear_multi_bwd(const mg_tok_lin* ops, int n, void* stream) {
    TokLinSet set;
    int rc = tok_set(ops, n, &set, 1); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    size_t lds = 0; int gx = 0, gz = 0; bool need_dz = false;
    for (int i = 0; i < n; ++i) {
        const mg_tok_lin& o = ops[i];
        const int nmax = o.dx_pair ? (ops[o.dx_pair - 1].N > o.N ? ops[o.dx_pair - 1].N : o.N) : o.N;      // (validated by tok_set)
        const size_t lr = ((size_t)RB * nmax + (size_t)nmax * o.K) * sizeof(float), lc = (size_t)32 * (CB + o.K) * sizeof(float);
        const size_t l = o.dx ? (lr > lc ? lr : lc) : lc;
        if (l > lds) lds = l;
        const int g = (o.dx ? (o.R + RB - 1) / RB : 0) + (o.N + CB - 1) / CB; if (g > gx) gx = g;
        const int z = (o.R + 3) / 4; if (z > gz) gz = z;
        if (o.gamma || o.relu || o.dz != o.dy || o.dres) need_dz = true;
    }
    if (lds > 150 * 1024) return -3;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)token_linear_multi_bwd_main_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    if (need_dz) hipLaunchKernelGGL(token_linear_multi_dz_kernel, dim3(gz, n), dim3(NT), 0, st, set);
    hipLaunchKernelGGL(token_linear_multi_bwd_main_kernel, dim3(gx, n), dim3(NT), lds, st, set);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---- gradient of a tensor with several consumers as ONE launch: out[i] = ((src_0[i] + src_1[i]) + src_2[i]) + ... in the given order (autograd
// adds the k gradients pairwise, k - 1 launches of a few microseconds each; the token side of the instance matte decoder has ~45 of them per step:
// maggie_amd/functional.py FanOut). fp32, n elements, k <= 16.
namespace {
struct SumSrcs { const float* p[16]; int k; };
__global__ __launch_bounds__(256) void sum_k_kernel(const SumSrcs s, long n, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = j < s.k ? s.p[j][i] : 0.f;
        float a = v[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) if (j < s.k) a += v[j];
        out[i] = a;
    }
}
}  // namespace
extern "C" int mg_sum_k(const float* const* srcs, int k, long n, float* out, void* stream) {
    if (!srcs || !out || k < 1 || k > 16 || n < 0) return -2;
    if (n == 0) return 0;
    SumSrcs s; s.k = k;
    for (int j = 0; j < 16; ++j) { s.p[j] = j < k ? srcs[j] : nullptr; if (j < k && !srcs[j]) return -2; }
    long b = (n + 255) / 256; if (b > 2048) b = 2048;
    hipLaunchKernelGGL(sum_k_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, s, n, out);
    MG_CHECK_LAUNCH();
    return 0;
}


// The same for 16-bit tensors (bf16 / fp16 activation gradients with several consumers: the ASPP input feeds five branches): the k terms are added in
// fp32 in the given order and rounded ONCE (autograd's k - 1 pairwise adds round k - 1 times). n % 8 == 0 is not required.
namespace {
struct SumSrcsRaw { const void* p[16]; int k; };
template <typename T>
__global__ __launch_bounds__(256) void sum_k16_kernel(const SumSrcsRaw s, long n, T* __restrict__ out) {
    using TR = ElemTraits<T>;
    constexpr int CE = TR::CE;
    const long nv = n / CE;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float a[CE];
        TR::unpack(((const uint4*)s.p[0])[i], a);
#pragma unroll
        for (int j = 1; j < 16; ++j) {
            if (j < s.k) {
                float v[CE];
                TR::unpack(((const uint4*)s.p[j])[i], v);
#pragma unroll
                for (int e = 0; e < CE; ++e) a[e] += v[e];
            }
        }
        ((uint4*)out)[i] = TR::pack(a);
    }
    if (blockIdx.x == 0) {
        for (long i = nv * CE + threadIdx.x; i < n; i += 256) {
            float a = TR::ld((const T*)s.p[0] + i);
            for (int j = 1; j < s.k; ++j) a += TR::ld((const T*)s.p[j] + i);
            TR::st(out + i, a);
        }
    }
}
}  // namespace
extern "C" int mg_sum_k_t(const void* const* srcs, int k, long n, void* out, int dtype, void* stream) {
    if (dtype == MG_F32) return mg_sum_k((const float* const*)srcs, k, n, (float*)out, stream);
    if (!MG_IS16(dtype)) return -3;
    if (!srcs || !out || k < 1 || k > 16 || n < 0) return -2;
    if (n == 0) return 0;
    SumSrcsRaw s; s.k = k;
    for (int j = 0; j < 16; ++j) {
        s.p[j] = j < k ? srcs[j] : nullptr;
        if (j < k && (!srcs[j] || (((size_t)srcs[j]) & 15))) return -2;
    }
    if (((size_t)out) & 15) return -2;
    long b = (n / 8 + 255) / 256; if (b > 2048) b = 2048; if (b < 1) b = 1;
    if (dtype == MG_BF16) hipLaunchKernelGGL(sum_k16_kernel<bf16raw>, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, s, n, (bf16raw*)out);
    else hipLaunchKernelGGL(sum_k16_kernel<f16raw>, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, s, n, (f16raw*)out);
    MG_CHECK_LAUNCH();
    return 0;
}
