// Region ops of the MaGGIe hot path on bit-packed planes (1 bit per pixel, 64 pixels per uint64 word):
//   * compute_unknown  = threshold (1/255 < a < 254/255) + OpenCV-ellipse binary dilation
//                        (reference: maggie/utils/utils.py:27-55, a device->HOST->device cv2.dilate round trip there)
//   * the active-site pyramid OS1 -> OS2 -> OS4 -> OS8 that spconv's SparseConv2d(k3,s2,p1) rule books define
//                        (reference: maggie/network/decoder/resnet_inst_matt_spconv.py:61-66,217-218 `dummy_downscale`)
//   * sorted (batch,y,x) site lists (== torch.nonzero order, :206-214) and the gather tables of the sparse convs.
// Everything stays on the device: integer/bit work, bit-exact by construction; the only host sync the caller needs
// is reading the four site counts to size the feature matrices.
#include "common.h"
#include "../../include/maggie_hip.h"
#include <math.h>
#include <mutex>

namespace {

typedef unsigned long long u64;
constexpr int MAXK = 32;

// ---- OpenCV getStructuringElement(MORPH_ELLIPSE,(k,k)) row spans, relative to the anchor k/2 -----------------------
struct SeTable { int8_t lo[MAXK][MAXK]; int8_t hi[MAXK][MAXK]; };   // [k][row]; lo > hi => empty row
__constant__ SeTable c_se;
std::once_flag g_se_once;
int g_se_rc = 0;

void build_se_table(SeTable& t) {
    for (int k = 0; k < MAXK; ++k)
        for (int i = 0; i < MAXK; ++i) { t.lo[k][i] = 1; t.hi[k][i] = 0; }
    for (int k = 1; k < MAXK; ++k) {
        if (k == 1) { t.lo[1][0] = 0; t.hi[1][0] = 0; continue; }
        int r = k / 2, c = k / 2;
        double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
        for (int i = 0; i < k; ++i) {
            int dy = i - r;
            if (abs(dy) <= r) {
                int dx = (int)nearbyint(c * sqrt((r * r - dy * dy) * inv_r2));   // cvRound: round half to even
                int j1 = c - dx < 0 ? 0 : c - dx;
                int j2 = c + dx + 1 > k ? k : c + dx + 1;
                if (j2 > j1) { t.lo[k][i] = (int8_t)(j1 - c); t.hi[k][i] = (int8_t)(j2 - 1 - c); }
            }
        }
    }
}

int ensure_se_table() {
    std::call_once(g_se_once, [] {
        SeTable t;
        build_se_table(t);
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_se), &t, sizeof(t));
        g_se_rc = (int)e;
    });
    return g_se_rc;
}

// ---- pack ------------------------------------------------------------------------------------------------------------
// mode 0: lo < a < hi ; mode 1: a > 0.  One wave packs one 64-pixel word with a ballot (coalesced 256-byte reads).
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const T* __restrict__ a, u64* __restrict__ bits, long nwords, int W, int Ww,
                                                   int mode, float lo, float hi) {
    const int lane = threadIdx.x & 63;
    long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; w < nwords; w += stride) {
        long row = w / Ww; int wj = (int)(w - row * Ww);
        int x = wj * 64 + lane;
        bool on = false;
        if (x < W) {
            float v = (float)a[row * W + x];
            on = mode == 0 ? (v > lo && v < hi) : (v > 0.f);
        }
        u64 m = __ballot(on);
        if (lane == 0) bits[w] = m;
    }
}

__global__ __launch_bounds__(256) void unpack_u8_kernel(const u64* __restrict__ bits, uint8_t* __restrict__ out, long nwords, int W, int Ww) {
    const int lane = threadIdx.x & 63;
    long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; w < nwords; w += stride) {
        long row = w / Ww; int wj = (int)(w - row * Ww);
        int x = wj * 64 + lane;
        if (x < W) out[row * W + x] = (uint8_t)((bits[w] >> lane) & 1ull);
    }
}

// the same straight to fp32 planes (loss weights: the uint8 planes were unpacked, then cast to fp32 by a second pass over 42 MB)
__global__ __launch_bounds__(256) void unpack_f32_kernel(const u64* __restrict__ bits, float* __restrict__ out, long nwords, int W, int Ww) {
    const int lane = threadIdx.x & 63;
    long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; w < nwords; w += stride) {
        long row = w / Ww; int wj = (int)(w - row * Ww);
        int x = wj * 64 + lane;
        if (x < W) out[row * W + x] = (float)((bits[w] >> lane) & 1ull);
    }
}

// ---- progressive refinement blend (maggie/network/decoder/resnet_inst_matt_spconv.py:272-290): alpha = a * w + b * (1 - w) with
// w = the unknown-region bit plane, i.e. a per-pixel select. Reads the bit plane itself: no uint8/float weight tensor, one pass.
__global__ __launch_bounds__(256) void select_kernel(const u64* __restrict__ bits, const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, long nwords, int W, int Ww) {
    const int lane = threadIdx.x & 63;
    long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; w < nwords; w += stride) {
        const long row = w / Ww; const int wj = (int)(w - row * Ww);
        const int x = wj * 64 + lane;
        if (x < W) {
            const long i = row * W + x;
            out[i] = ((bits[w] >> lane) & 1ull) ? a[i] : b[i];
        }
    }
}
__global__ __launch_bounds__(256) void select_bwd_kernel(const u64* __restrict__ bits, const float* __restrict__ dy, float* __restrict__ da,
                                                         float* __restrict__ db, long nwords, int W, int Ww) {
    const int lane = threadIdx.x & 63;
    long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (; w < nwords; w += stride) {
        const long row = w / Ww; const int wj = (int)(w - row * Ww);
        const int x = wj * 64 + lane;
        if (x < W) {
            const long i = row * W + x;
            const bool on = (bits[w] >> lane) & 1ull;
            const float g = dy[i];
            if (da) da[i] = on ? g : 0.f;
            if (db) db[i] = on ? 0.f : g;
        }
    }
}

// ---- dilation ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 span_or(u64 prev, u64 cur, u64 next, int lo, int hi) {
    // out bit j = OR_{t=lo..hi} src(x0 + j + t);  U bit i <-> pixel x0 - 32 + i
    unsigned __int128 U = ((unsigned __int128)cur << 32) | (unsigned __int128)(prev >> 32) | ((unsigned __int128)next << 96);
    const int L = hi - lo + 1;
    unsigned __int128 f = U;
    int done = 1;
    while (done * 2 <= L) { f |= f >> done; done *= 2; }
    if (done < L) f |= f >> (L - done);
    return (u64)(f >> (32 + lo));
}

__global__ __launch_bounds__(256) void dilate_kernel(const u64* __restrict__ in, u64* __restrict__ out, const u64* __restrict__ andmask,
                                                     int P, int H, int W, int Ww, const int* __restrict__ widths, int width_all) {
    long total = (long)P * H * Ww;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int wj = (int)(i % Ww); long r = i / Ww; int y = (int)(r % H); int p = (int)(r / H);
        int k = widths ? widths[p] : width_all;
        int a = k / 2;
        u64 acc = 0;
        const u64* plane = in + (long)p * H * Ww;
        // eight element rows per trip, their words and spans loaded as one batch from clamped addresses and masked afterwards: the row-by-row walk
        // (`if (lo > hi) continue; if (ys outside) continue; three loads`) was one memory round trip per element row, up to 29 per word
        const int wjm = wj > 0 ? wj - 1 : 0, wjp = wj + 1 < Ww ? wj + 1 : Ww - 1;
        for (int r0 = 0; r0 < k; r0 += 8) {
            u64 cur[8], prev[8], next[8];
            int lo[8], hi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = min(r0 + u, MAXK - 1);
                const int ys = min(max(y + row - a, 0), H - 1);
                const u64* rp = plane + (long)ys * Ww;
                cur[u] = rp[wj]; prev[u] = rp[wjm]; next[u] = rp[wjp];
                lo[u] = c_se.lo[k][row]; hi[u] = c_se.hi[k][row];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = r0 + u, ys = y + row - a;
                if (row < k && lo[u] <= hi[u] && ys >= 0 && ys < H)
                    acc |= span_or(wj > 0 ? prev[u] : 0ull, cur[u], wj + 1 < Ww ? next[u] : 0ull, lo[u], hi[u]);
            }
        }
        int rem = W - wj * 64;
        if (rem < 64) acc &= (rem <= 0) ? 0ull : ((1ull << rem) - 1ull);
        if (andmask) acc &= andmask[i];
        out[i] = acc;
    }
}

// ---- SparseConv2d(k3,s2,p1) active-set rule: coarse(y,x) = OR fine(2y-1+ky, 2x-1+kx) -----------------------------------
__device__ __forceinline__ u64 compress_even(u64 x) {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return x;
}

__global__ __launch_bounds__(256) void downsample_kernel(const u64* __restrict__ fine, u64* __restrict__ coarse, int P, int Hf, int Wwf,
                                                         int Hc, int Wc, int Wwc) {
    long total = (long)P * Hc * Wwc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int wj = (int)(i % Wwc); long r = i / Wwc; int y = (int)(r % Hc); int p = (int)(r / Hc);
        u64 acc = 0;
        for (int ky = 0; ky < 3; ++ky) {
            int yf = 2 * y - 1 + ky;
            if (yf < 0 || yf >= Hf) continue;
            const u64* rp = fine + ((long)p * Hf + yf) * Wwf;
            int a = 2 * wj, b = 2 * wj + 1;
            u64 A = a < Wwf ? rp[a] : 0ull, B = b < Wwf ? rp[b] : 0ull;
            u64 pm = (a > 0) ? (rp[a - 1] >> 63) : 0ull;
            u64 HA = A | (A << 1) | pm | (A >> 1) | (B << 63);
            u64 HB = B | (B << 1) | (A >> 63) | (B >> 1);
            acc |= compress_even(HA) | (compress_even(HB) << 32);
        }
        int rem = Wc - wj * 64;
        if (rem < 64) acc &= (rem <= 0) ? 0ull : ((1ull << rem) - 1ull);
        coarse[i] = acc;
    }
}

// ---- ranking: per-row popcounts, exclusive scan, absolute per-word offsets ------------------------------------------------
__global__ __launch_bounds__(256) void rowcount_kernel(const u64* __restrict__ bits, int nrows, int Ww, int* __restrict__ counts) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) {
        int c = 0;
        for (int j = 0; j < Ww; ++j) c += __popcll(bits[(long)r * Ww + j]);
        counts[r] = c;
    }
}

// single block, 1024 threads: rowoff[0..n] exclusive scan of counts[0..n-1]; rowoff[n] = total
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ counts, int n, int* __restrict__ rowoff) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(n, b + per);
    constexpr int MAXPER = 32;
    if (per <= MAXPER) {
        // up to 32 counts per thread (the image levels: <= 20): read ONCE, as one batch, kept in registers for the second walk; the block scan is a
        // shuffle scan per wave + one over the 16 wave totals (3 barriers). The general form below reads its counts twice, one load per round trip,
        // and goes through 20 barriers of 1024 threads (13 us per level, five levels per step).
        int c[MAXPER];
#pragma unroll
        for (int k = 0; k < MAXPER; ++k) c[k] = (k < per && b + k < n) ? counts[b + k] : 0;
        int s = 0;
#pragma unroll
        for (int k = 0; k < MAXPER; ++k) s += c[k];
        const int lane = t & 63, wave = t >> 6;
        int incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            const int x = lane < 16 ? part[lane] : 0;
            int sc = x;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) { const int v = __shfl_up(sc, off, 64); if (lane >= off) sc += v; }
            if (lane < 16) part[32 + lane] = sc - x;              // exclusive prefix of the wave totals
        }
        __syncthreads();
        int run = part[32 + wave] + incl - s;
#pragma unroll
        for (int k = 0; k < MAXPER; ++k) {
            if (k < per && b + k < n) { rowoff[b + k] = run; run += c[k]; }
        }
        if (t == 1023) rowoff[n] = run;                           // the last thread's exclusive prefix + its own counts = the total
        return;
    }
    int s = 0;
    for (int i = b; i < e; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; ++i) { rowoff[i] = run; run += counts[i]; }
    if (t == 1023) rowoff[n] = part[1023];
}

__global__ __launch_bounds__(256) void wordoff_kernel(const u64* __restrict__ bits, const int* __restrict__ rowoff, int nrows, int Ww,
                                                      int* __restrict__ wordoff) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) {
        int run = rowoff[r];
        for (int j = 0; j < Ww; ++j) { wordoff[(long)r * Ww + j] = run; run += __popcll(bits[(long)r * Ww + j]); }
    }
}

// coords[r] = (plane, y, x) for every set bit in sorted order; one thread per word
__global__ __launch_bounds__(256) void emit_coords_kernel(const u64* __restrict__ bits, const int* __restrict__ wordoff, long nwords, int H,
                                                          int Ww, int* __restrict__ coords) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (long)gridDim.x * 256) {
        u64 m = bits[i];
        if (!m) continue;
        int wj = (int)(i % Ww); long r = i / Ww; int y = (int)(r % H); int p = (int)(r / H);
        int row = wordoff[i];
        while (m) {
            int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            coords[(long)row * 3 + 0] = p; coords[(long)row * 3 + 1] = y; coords[(long)row * 3 + 2] = wj * 64 + b;
            ++row;
        }
    }
}

__device__ __forceinline__ int rank_of(const u64* __restrict__ bits, const int* __restrict__ wordoff, int p, int y, int x, int H, int W, int Ww) {
    if (y < 0 || y >= H || x < 0 || x >= W) return -1;
    long wi = ((long)p * H + y) * Ww + (x >> 6);
    u64 m = bits[wi];
    int b = x & 63;
    if (!((m >> b) & 1ull)) return -1;
    return wordoff[wi] + __popcll(m & ((1ull << b) - 1ull));
}

// KIND 0: submanifold k x k (same level);  KIND 1: inverse-conv gather (rows = fine sites, source = coarse level,
// tap (ky,kx) valid iff (y+1-ky) even ...);  KIND 2: strided gather (rows = coarse sites, source = fine level, i = 2o-1+k)
template <int KIND>
__global__ __launch_bounds__(256) void table_kernel(const int* __restrict__ coords, int R, int ksize, const u64* __restrict__ sbits,
                                                    const int* __restrict__ swordoff, int Hs, int Ws, int Wws, int* __restrict__ nbr,
                                                    const int32_t* __restrict__ r_dev) {
    R = dev_rows(r_dev, R);
    const int taps = ksize * ksize;
    const int total = R * taps;
    const int c = ksize / 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / taps, tap = i - r * taps;
        const int ky = tap / ksize, kx = tap - ky * ksize;
        const int p = coords[r * 3], y = coords[r * 3 + 1], x = coords[r * 3 + 2];
        int sy, sx;
        bool ok = true;
        if (KIND == 0) { sy = y + ky - c; sx = x + kx - c; }
        else if (KIND == 1) {
            const int ty = y + 1 - ky, tx = x + 1 - kx;
            ok = (ty >= 0) && (tx >= 0) && ((ty & 1) == 0) && ((tx & 1) == 0);
            sy = ty >> 1; sx = tx >> 1;
        } else { sy = 2 * y - 1 + ky; sx = 2 * x - 1 + kx; }
        int v = -1;
        if (ok && sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) {
            const int wi = (p * Hs + sy) * Wws + (sx >> 6);
            const u64 m = sbits[wi];
            const int b = sx & 63;
            if ((m >> b) & 1ull) v = swordoff[wi] + __popcll(m & ((1ull << b) - 1ull));
        }
        nbr[i] = v;
    }
}

// Bounded capacity of the sparse head (round 3): when a level has more active sites than the row buffers were sized for, the sites of rank
// >= cap are REMOVED from the bit planes (so every later kernel sees a consistent, smaller active set: no rank ever points past a buffer),
// the level's count word is clamped and a sticky device flag is raised -- the host raises on it at its next (already existing) flag read.
__global__ __launch_bounds__(256) void truncate_kernel(u64* __restrict__ bits, const int* __restrict__ wordoff, long nwords, int cap,
                                                       int* __restrict__ count, int* __restrict__ overflow) {
    if (*count <= cap) return;                            // (uniform: every thread reads the same word before anyone clamps it -- see the launcher)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (long)gridDim.x * 256) {
        const u64 m = bits[i];
        if (!m) continue;
        const int off = wordoff[i];
        if (off >= cap) { bits[i] = 0ull; continue; }
        int keep = cap - off;                             // >= 1
        if (__popcll(m) <= keep) continue;
        u64 r = m, out = 0ull;
        while (keep-- > 0) { const u64 low = r & (~r + 1ull); out |= low; r ^= low; }
        bits[i] = out;
    }
}
__global__ void truncate_finish_kernel(int cap, int* __restrict__ count, int* __restrict__ overflow) {
    if (*count > cap) { *count = cap; *overflow = 1; }
}

inline int grid_for(long total, int per_block) {
    long b = (total + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int mg_bits_pack(const void* a, int dtype, void* bits, int P, int H, int W, int mode, float lo, float hi, void* stream) {
    int Ww = (W + 63) / 64;
    long nwords = (long)P * H * Ww;
    if (nwords <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_F32) hipLaunchKernelGGL(pack_kernel<float>, dim3(grid_for(nwords, 4)), dim3(256), 0, st, (const float*)a, (u64*)bits, nwords, W, Ww, mode, lo, hi);
    else if (dtype == 2) hipLaunchKernelGGL(pack_kernel<uint8_t>, dim3(grid_for(nwords, 4)), dim3(256), 0, st, (const uint8_t*)a, (u64*)bits, nwords, W, Ww, mode, lo, hi);
    else return -6;
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_unpack_u8(const void* bits, uint8_t* out, int P, int H, int W, void* stream) {
    int Ww = (W + 63) / 64;
    long nwords = (long)P * H * Ww;
    if (nwords <= 0) return 0;
    hipLaunchKernelGGL(unpack_u8_kernel, dim3(grid_for(nwords, 4)), dim3(256), 0, (hipStream_t)stream, (const u64*)bits, out, nwords, W, Ww);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_unpack_f32(const void* bits, float* out, int P, int H, int W, void* stream) {
    const int Ww = (W + 63) / 64;
    long nwords = (long)P * H * Ww;
    if (nwords <= 0) return 0;
    hipLaunchKernelGGL(unpack_f32_kernel, dim3(grid_for(nwords, 4)), dim3(256), 0, (hipStream_t)stream, (const u64*)bits, out, nwords, W, Ww);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_select(const void* bits, const float* a, const float* b, float* out, int P, int H, int W, void* stream) {
    const int Ww = (W + 63) / 64;
    const long nwords = (long)P * H * Ww;
    if (nwords <= 0) return 0;
    hipLaunchKernelGGL(select_kernel, dim3(grid_for(nwords, 4)), dim3(256), 0, (hipStream_t)stream, (const u64*)bits, a, b, out, nwords, W, Ww);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_select_bwd(const void* bits, const float* dy, float* da, float* db, int P, int H, int W, void* stream) {
    const int Ww = (W + 63) / 64;
    const long nwords = (long)P * H * Ww;
    if (nwords <= 0) return 0;
    hipLaunchKernelGGL(select_bwd_kernel, dim3(grid_for(nwords, 4)), dim3(256), 0, (hipStream_t)stream, (const u64*)bits, dy, da, db, nwords, W, Ww);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_dilate(const void* in, void* out, const void* andmask, int P, int H, int W, const int32_t* widths, int width_all,
                              void* stream) {
    int rc = ensure_se_table();
    if (rc) return rc;
    if (!widths && (width_all < 1 || width_all >= MAXK)) return -2;
    int Ww = (W + 63) / 64;
    long total = (long)P * H * Ww;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(dilate_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const u64*)in, (u64*)out,
                       (const u64*)andmask, P, H, W, Ww, widths, width_all);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_downsample(const void* fine, void* coarse, int P, int Hf, int Wf, void* stream) {
    int Hc = (Hf - 1) / 2 + 1, Wc = (Wf - 1) / 2 + 1;
    int Wwf = (Wf + 63) / 64, Wwc = (Wc + 63) / 64;
    long total = (long)P * Hc * Wwc;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(downsample_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const u64*)fine, (u64*)coarse, P,
                       Hf, Wwf, Hc, Wc, Wwc);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_rank(const void* bits, int P, int H, int W, int32_t* counts_tmp, int32_t* rowoff, int32_t* wordoff, void* stream) {
    int Ww = (W + 63) / 64;
    int nrows = P * H;
    if (nrows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rowcount_kernel, dim3(grid_for(nrows, 256)), dim3(256), 0, st, (const u64*)bits, nrows, Ww, counts_tmp);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, counts_tmp, nrows, rowoff);
    hipLaunchKernelGGL(wordoff_kernel, dim3(grid_for(nrows, 256)), dim3(256), 0, st, (const u64*)bits, rowoff, nrows, Ww, wordoff);
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_bits_coords(const void* bits, const int32_t* wordoff, int P, int H, int W, int32_t* coords, void* stream) {
    int Ww = (W + 63) / 64;
    long nwords = (long)P * H * Ww;
    if (nwords <= 0) return 0;
    hipLaunchKernelGGL(emit_coords_kernel, dim3(grid_for(nwords, 256)), dim3(256), 0, (hipStream_t)stream, (const u64*)bits, wordoff, nwords,
                       H, Ww, coords);
    MG_CHECK_LAUNCH();
    return 0;
}

/* Keep at most `cap` active sites of a level (in rank order): clears the bits of the others in place, clamps *count (the level's site count,
 * = rowoff[P*H] of mg_bits_rank) and sets *overflow = 1 when anything was dropped. wordoff stays valid for the kept sites. */
extern "C" int mg_bits_truncate(void* bits, const int32_t* wordoff, int P, int H, int W, int cap, int32_t* count, int32_t* overflow, void* stream) {
    const int Ww = (W + 63) / 64;
    const long nwords = (long)P * H * Ww;
    if (nwords <= 0 || cap < 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(truncate_kernel, dim3(grid_for(nwords, 256)), dim3(256), 0, st, (u64*)bits, wordoff, nwords, cap, count, overflow);
    hipLaunchKernelGGL(truncate_finish_kernel, dim3(1), dim3(1), 0, st, cap, count, overflow);      // after every block has read the unclamped count
    MG_CHECK_LAUNCH();
    return 0;
}

extern "C" int mg_gather_table_dev(const int32_t* coords, int R, int ksize, int kind, const void* src_bits, const int32_t* src_wordoff, int Hs,
                                   int Ws, int32_t* nbr, const int32_t* r_dev, void* stream);
extern "C" int mg_gather_table(const int32_t* coords, int R, int ksize, int kind, const void* src_bits, const int32_t* src_wordoff, int Hs,
                               int Ws, int32_t* nbr, void* stream) {
    return mg_gather_table_dev(coords, R, ksize, kind, src_bits, src_wordoff, Hs, Ws, nbr, nullptr, stream);
}
extern "C" int mg_gather_table_dev(const int32_t* coords, int R, int ksize, int kind, const void* src_bits, const int32_t* src_wordoff, int Hs,
                                   int Ws, int32_t* nbr, const int32_t* r_dev, void* stream) {
    if (R <= 0) return 0;
    if (kind < 0 || kind > 2) return -2;
    long total = (long)R * ksize * ksize;
    if (total >= (1l << 31)) return -7;
    dim3 g(grid_for(total, 256)), b(256);
    hipStream_t st = (hipStream_t)stream;
    const int Wws = (Ws + 63) / 64;
    if (kind == 0) hipLaunchKernelGGL(table_kernel<0>, g, b, 0, st, coords, R, ksize, (const u64*)src_bits, src_wordoff, Hs, Ws, Wws, nbr, r_dev);
    else if (kind == 1) hipLaunchKernelGGL(table_kernel<1>, g, b, 0, st, coords, R, ksize, (const u64*)src_bits, src_wordoff, Hs, Ws, Wws, nbr, r_dev);
    else hipLaunchKernelGGL(table_kernel<2>, g, b, 0, st, coords, R, ksize, (const u64*)src_bits, src_wordoff, Hs, Ws, Wws, nbr, r_dev);
    MG_CHECK_LAUNCH();
    return 0;
}
