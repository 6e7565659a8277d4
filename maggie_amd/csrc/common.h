// Shared device helpers for the MaGGIe gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MG_F32 0
#define MG_BF16 1
#define MG_F16 3

#define MG_ACT_NONE 0
#define MG_ACT_RELU 1
#define MG_ACT_LRELU 2

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16raw;
// IEEE half storage (the reference's `--precision 16`: fp16 autocast + GradScaler, engine/train.py:208,227-229). A distinct 2-byte type so that
// every kernel templated on its 16-bit storage type gets an fp16 instantiation next to the bf16 one; same vector widths, same MFMA shape
// (v_mfma_f32_16x16x32_f16 runs at the bf16 rate on gfx950), fp32 accumulation everywhere.
struct f16raw { uint16_t v; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 mg_f16x2;

__device__ __forceinline__ float bf2f(bf16raw v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 has it in hardware (v_cvt_pk_bf16_f32, two values per instruction). The integer sequence
// this replaces (NaN test, +0x7fff + lsb, shift: ~8 VALU per value) made the conv epilogues VALU-bound: 13.4 k of the 26 k cycles of a
// C128 64x64 halo tile were the 32 outputs per thread being rounded twice (tools/halo_timeline.py).
typedef __attribute__((ext_vector_type(2))) float mg_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 mg_bf16x2;
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
    const mg_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mg_bf16x2));
}
__device__ __forceinline__ bf16raw f2bf(float f) { return (bf16raw)(f2bf_pk(f, 0.f) & 0xffffu); }

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int CE = 4;      // elements per 16-byte chunk
    static constexpr int EPS = 16;    // elements per 64-byte K slab
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }
    __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
        f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y); f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct ElemTraits<bf16raw> {
    static constexpr int CE = 8;
    static constexpr int EPS = 32;
    __device__ static __forceinline__ float ld(const bf16raw* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16raw* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
    __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
        f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
        f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
        f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xffff0000u);
        f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xffff0000u);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(f2bf_pk(f[0], f[1]), f2bf_pk(f[2], f[3]), f2bf_pk(f[4], f[5]), f2bf_pk(f[6], f[7]));
    }
};

__device__ __forceinline__ uint32_t f2h_pk(float lo, float hi) {
    const mg_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mg_f16x2));      // v_cvt_pk_f16_f32: round to nearest even, overflow -> inf
}
__device__ __forceinline__ void h2f_pk(uint32_t w, float& lo, float& hi) {
    const mg_f32x2 v = __builtin_convertvector(__builtin_bit_cast(mg_f16x2, w), mg_f32x2);
    lo = v[0]; hi = v[1];
}
template <> struct ElemTraits<f16raw> {
    static constexpr int CE = 8;
    static constexpr int EPS = 32;
    __device__ static __forceinline__ float ld(const f16raw* p) { return (float)__builtin_bit_cast(_Float16, p->v); }
    __device__ static __forceinline__ void st(f16raw* p, float v) { p->v = __builtin_bit_cast(uint16_t, (_Float16)v); }
    __device__ static __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
    __device__ static __forceinline__ void unpack(const uint4& q, float* f) {
        h2f_pk(q.x, f[0], f[1]); h2f_pk(q.y, f[2], f[3]); h2f_pk(q.z, f[4], f[5]); h2f_pk(q.w, f[6], f[7]);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(f2h_pk(f[0], f[1]), f2h_pk(f[2], f[3]), f2h_pk(f[4], f[5]), f2h_pk(f[6], f[7]));
    }
};

// one 16x16x32 MFMA step on 16-byte fragments (any 16-byte register type) of the 16-bit storage type T, fp32 accumulate
template <typename T> struct IsF16 { static constexpr bool value = false; };
template <> struct IsF16<f16raw> { static constexpr bool value = true; };
template <typename T, typename V>
__device__ __forceinline__ f32x4 mfma16(const V& a, const V& b, const f32x4& c) {
    static_assert(sizeof(V) == 16, "MFMA operand fragments are 16 bytes");
    if constexpr (IsF16<T>::value) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else if constexpr (sizeof(T) == 2) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return c;                                            // fp32 storage: the callers use the 16x16x4 f32 MFMA instead
}

// 16-bit activation / weight storage (bf16 or IEEE half): same vector widths and tile shapes
#define MG_IS16(code) ((code) == MG_BF16 || (code) == MG_F16)

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == MG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MG_ACT_LRELU) return v > 0.f ? v : v * slope;
    return v;
}

// Operand transform of the convolution family (mg_conv_params.xf_*): one 16-byte chunk (8 elements of the 16-bit storage type T) of the raw
// producer output -> act(x * scale + shift), rounded to T -- the bits mg_affine_act would have stored. `sl` = 1 (no activation), 0 (ReLU) or the
// LeakyReLU slope: act(v) = max(v, v * sl), branch-free like the conv epilogue.
#define MG_XF_UNSUPPORTED (-9)     /* xf_scale given but the kernel form this geometry dispatches to cannot transform its operand in flight */
template <typename T>
__device__ __forceinline__ uint4 xf_apply8(const uint4& q, const float* sc, const float* sh, float sl) {
    float f[8];
    ElemTraits<T>::unpack(q, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float v = f[e] * sc[e] + sh[e]; f[e] = fmaxf(v, v * sl); }
    return ElemTraits<T>::pack(f);
}
__host__ __device__ __forceinline__ float xf_slope_of(int act, float slope) { return act == MG_ACT_NONE ? 1.f : (act == MG_ACT_RELU ? 0.f : slope); }

// Row count of a sparse-head launch: the device word when given (clamped to the capacity the buffers were sized for), else the host value.
__device__ __forceinline__ int dev_rows(const int32_t* __restrict__ m_dev, int cap) {
    if (!m_dev) return cap;
    const int v = *m_dev;
    return v < 0 ? 0 : (v < cap ? v : cap);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Division by a loop-invariant divisor without v_rcp sequences in hot loops: q = umulhi(n, floor((2^32-1)/d)) is at most
// 2 below floor(n/d) for n < 2^31; two compare-and-fix steps make it exact.
struct FastDiv {
    uint32_t d, m;
    __device__ __forceinline__ explicit FastDiv(int div) : d((uint32_t)div), m(0xFFFFFFFFu / (uint32_t)div) {}
    __device__ __forceinline__ void divmod(int n, int& q, int& r) const {
        uint32_t qq = __umulhi((uint32_t)n, m);
        uint32_t rr = (uint32_t)n - qq * d;
        if (rr >= d) { ++qq; rr -= d; }
        if (rr >= d) { ++qq; rr -= d; }
        q = (int)qq; r = (int)rr;
    }
};

// Zero `n` 32-bit words with a kernel instead of hipMemsetAsync: memset nodes captured into a hipGraph were observed not to
// be ordered before the kernels that follow them on replay (ROCm 7.2), which corrupts accumulators; a kernel node is.
static __global__ void mg_zero_words_kernel(uint32_t* __restrict__ p, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < n; i += step) p[i] = 0u;
}
// Accumulators carved from a buffer the caller has ALREADY zeroed (the per-graph zero arena: one memset node at the head of every replay,
// maggie_amd/functional.py ZeroArena) need no fill launch of their own: the host registers the arena's address range
// (mg_set_zeroed_range) and promises to hand every slice out once. ~60 fill launches of ~4.7 us per step disappear from the captured graphs.
extern "C" char* mg_zeroed_lo;
extern "C" char* mg_zeroed_hi;
extern "C" int mg_zero_claim(void* p, long bytes);        // csrc/abi.hip: 1 = these words were already handed to an accumulator in the current range
static inline hipError_t mg_zero_words(void* p, long n_words, hipStream_t st) {
    if (n_words <= 0) return hipSuccess;
    if ((char*)p >= mg_zeroed_lo && (char*)p + 4 * n_words <= mg_zeroed_hi)
        return mg_zero_claim(p, 4 * n_words) ? hipErrorAlreadyMapped : hipSuccess;
    long blocks = (n_words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mg_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, n_words);
    return hipGetLastError();
}

// ---- deterministic cross-workgroup sums (csrc/det.hip) -------------------------------------------------------------------------
// mg_det_on (MAGGIE_DETERMINISTIC, default 1): kernels that end in a cross-workgroup fp32 sum store one partial per workgroup into a
// slot buffer (mg_det_scratch) instead of atomicAdd, and mg_det_reduce adds the slots in index order behind them.
#define MG_DET_MAX_SEGS 4
#ifndef MG_DET_STAT_ROWS
#define MG_DET_STAT_ROWS 1024      /* rows of a BatchNorm statistics scratch in deterministic mode (one per row block, <= 1024 of them) */
#endif
struct mg_det_seg { float* dst; int nv; long group_stride; };
extern int mg_det_on;
float* mg_det_scratch(long floats);                    // the library's slot scratch (nullptr: not initialised / too small)
float* mg_det_scratch_on(long floats, hipStream_t st);    // the same for a kernel that may run on the registered side stream (its own scratch there: det.hip)
// slots: [groups][nblk][rowstride]; the segments cover the columns [col0, col0 + sum nv) of a row
int mg_det_reduce(const float* slots, int nblk, int groups, int rowstride, int col0, const mg_det_seg* segs, int nseg, hipStream_t st);
// The arithmetic of det_reduce_kernel for ONE chunk of one column (rows b0 .. b0 + n of a slot matrix, `stride` floats apart): four running sums over
// the row index mod 4, sixteen / eight / four loads in flight, combined (a0 + a1) + (a2 + a3). Shared with the kernels that add their partial rows in
// their own tail (norm_act.hip: bn_bwd_reduce_kernel's last-arriver form): same chunking (MG_DET_CHUNKS chunks of ceil(nblk / MG_DET_CHUNKS) rows, chunk
// sums added in chunk order) -> same bits as the separate launch.
constexpr int MG_DET_CHUNKS = 16;
__device__ __forceinline__ float det_chunk_sum(const float* __restrict__ p, int n, size_t stride) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 15 < n; b += 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = p[(size_t)u * stride];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { a0 += x[u]; a1 += x[u + 1]; a2 += x[u + 2]; a3 += x[u + 3]; }
        p += 16 * stride;
    }
    for (; b + 7 < n; b += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = p[(size_t)u * stride];
#pragma unroll
        for (int u = 0; u < 8; u += 4) { a0 += x[u]; a1 += x[u + 1]; a2 += x[u + 2]; a3 += x[u + 3]; }
        p += 8 * stride;
    }
    for (; b + 3 < n; b += 4) {
        const float x0 = p[0], x1 = p[stride], x2 = p[2 * stride], x3 = p[3 * stride];
        a0 += x0; a1 += x1; a2 += x2; a3 += x3;
        p += 4 * stride;
    }
    for (; b < n; ++b) { a0 += p[0]; p += stride; }
    return (a0 + a1) + (a2 + a3);
}
// Ticket words of the last-arriver tails (zeroed once per device, self-resetting): one set for the registered side stream, one for everything else.
constexpr int MG_TAIL_WORDS = 64;
unsigned* mg_det_tail_words(hipStream_t st);
static inline int mg_det_reduce1(const float* slots, int nblk, float* dst, int nv, hipStream_t st) {
    mg_det_seg s{dst, nv, 0};
    return mg_det_reduce(slots, nblk, 1, nv, 0, &s, 1, st);
}
#define MG_COOP_WORDS 1024                             /* mg_coop_sync(): generation | error | 62 spare | <= 512 arrival flags ... */
unsigned* mg_coop_sync();                              // per-device hand-shake words of the cooperative single-launch kernels (zeroed once, persistent)
#define MG_DET_NO_SCRATCH (-7)     /* deterministic mode without (enough) slot scratch: mg_det_init was not called or the request is too large */

#define MG_CHECK_LAUNCH()                              \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)
