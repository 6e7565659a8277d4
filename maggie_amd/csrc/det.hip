// Deterministic cross-workgroup sums (MAGGIE_DETERMINISTIC, on by default).
//
// The reference trains with torch.backends.cudnn.deterministic = True (tools/main.py:135-136): two runs of one step give the same
// bits. Every kernel of this library that used to finish a cross-workgroup sum with fp32 atomicAdd (BatchNorm backward sums, bias /
// LayerNorm gradients, the token side of the attention backward, loss sums, the SpectralNorm dot products, the gradient norm) has a
// second form: each workgroup STORES its partial into its own slot of a scratch buffer, and mg_det_reduce -- one small launch behind
// the producer -- adds the slots in index order. The order of the additions is then a function of the launch geometry only.
//
// Slots layout: [groups][nblk][rowstride] fp32. A launch covers the columns [col0, col0 + sum of the segment widths) of a row: column c of
// segment s of group g is added into segs[s].dst[g * segs[s].group_stride + c] (+=: the destinations keep the semantics they had under
// atomicAdd).
//
// The scratch is ONE library-owned device buffer: launches on a stream execute in order, and a (producer, reduce) pair is always
// issued back to back on the same stream -- also inside a captured graph, where stream order becomes a dependency edge. Kernels of
// this library that run CONCURRENTLY on several streams (MAGGIE_BRANCHES, MAGGIE_SIDE_WGRAD: both off by default) must not use it.
#include "common.h"
#include "../../include/maggie_hip.h"

int mg_det_on = 1;
// One scratch PER DEVICE, allocated once and never freed or moved: captured graphs hold its address (the first version kept a
// single process-wide buffer and re-allocated it on growth or on a device change, under the feet of graphs captured before).
constexpr int MG_MAX_DEVICES = 16;
static char* g_buf[MG_MAX_DEVICES] = {};
static long g_bytes[MG_MAX_DEVICES] = {};
static unsigned* g_coop[MG_MAX_DEVICES] = {};     // mg_coop_sync(), below
static unsigned* g_tail[MG_MAX_DEVICES] = {};     // mg_det_tail_words(), below

extern "C" int mg_set_deterministic(int on) { mg_det_on = on ? 1 : 0; return 0; }
extern "C" int mg_get_deterministic(void) { return mg_det_on; }
extern "C" int mg_stat_rows(void) { return mg_det_on ? MG_DET_STAT_ROWS : MG_STAT_REPLICAS; }

/* Allocate the slot scratch of the CURRENT device (idempotent). Must be called outside a stream capture -- the Python binding does so on its
 * first call into the library on each device. A later call asking for MORE than the device's scratch holds fails with MG_DET_NO_SCRATCH (-7): the
 * buffer's address is part of every graph captured so far, so it never grows -- size it up front (MAGGIE_DET_SCRATCH_MB). */
extern "C" int mg_det_init(long bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MG_MAX_DEVICES) return -2;
    if (g_buf[dev]) return g_bytes[dev] >= bytes ? 0 : MG_DET_NO_SCRATCH;
    void* p = nullptr;
    e = hipMalloc(&p, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
    g_buf[dev] = (char*)p; g_bytes[dev] = bytes;
    void* c = nullptr;
    e = hipMalloc(&c, MG_COOP_WORDS * 4);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(c, 0, MG_COOP_WORDS * 4);
    if (e != hipSuccess) return (int)e;
    g_coop[dev] = (unsigned*)c;
    void* t = nullptr;
    e = hipMalloc(&t, 2 * MG_TAIL_WORDS * 4);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(t, 0, 2 * MG_TAIL_WORDS * 4);
    if (e != hipSuccess) return (int)e;
    g_tail[dev] = (unsigned*)t;
    return 0;
}

/* Cross-workgroup hand-shake words of the single-launch BatchNorm backward (csrc/norm_act.hip: bn_bwd_coop_kernel): [0] generation, [1] sticky
 * error (a peer workgroup that never arrived), [64 ...] one arrival flag per workgroup. Per device, allocated and ZEROED once (mg_det_init), never
 * moved: the generation scheme needs the words to survive from launch to launch, also across replays of a captured graph. */
unsigned* mg_coop_sync() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MG_MAX_DEVICES) return nullptr;
    return g_coop[dev];
}
extern "C" int mg_coop_error(int* out) {
    unsigned* w = mg_coop_sync();
    if (!w || !out) return -2;
    unsigned v = 0;
    hipError_t e = hipMemcpy(&v, w + 1, 4, hipMemcpyDeviceToHost);
    *out = (int)v;
    return e == hipSuccess ? 0 : (int)e;
}

float* mg_det_scratch(long floats) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MG_MAX_DEVICES) return nullptr;
    if (!g_buf[dev] || floats * 4 > g_bytes[dev]) return nullptr;
    return (float*)g_buf[dev];
}

// ---- a second scratch for ONE registered side stream per device (round 5) -------------------------------------------------------------------------
// The encoder's shortcut branches (maggie/network/encoder/resnet.py:167-175,194-198: conv -> ReLU -> BN, twice; needed only by the detail stage) run
// on a side stream next to the backbone -> ASPP -> decoder -> instance-token chain, forward and backward. Their BatchNorm kernels stage partial rows
// like everybody else: with a scratch of their own the two streams never meet in one buffer. Kernels that may run on the side stream ask with their
// stream (mg_det_scratch_on); everything else keeps mg_det_scratch.
static char* g_side_buf[MG_MAX_DEVICES] = {};
static long g_side_bytes[MG_MAX_DEVICES] = {};
static hipStream_t g_side_stream[MG_MAX_DEVICES] = {};

/* Register `stream` as the side stream of the CURRENT device (NULL: none) and allocate its scratch (`bytes`, once: never freed or moved -- captured
 * graphs hold the address). Call outside a stream capture. */
extern "C" int mg_det_side_stream(void* stream, long bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MG_MAX_DEVICES) return -2;
    if (stream && !g_side_buf[dev]) {
        void* p = nullptr;
        e = hipMalloc(&p, (size_t)bytes);
        if (e != hipSuccess) return (int)e;
        g_side_buf[dev] = (char*)p; g_side_bytes[dev] = bytes;
    }
    g_side_stream[dev] = (hipStream_t)stream;
    return 0;
}

unsigned* mg_det_tail_words(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MG_MAX_DEVICES || !g_tail[dev]) return nullptr;
    return g_tail[dev] + ((st && st == g_side_stream[dev]) ? MG_TAIL_WORDS : 0);
}

float* mg_det_scratch_on(long floats, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MG_MAX_DEVICES) return nullptr;
    if (st && st == g_side_stream[dev]) {
        if (!g_side_buf[dev] || floats * 4 > g_side_bytes[dev]) return nullptr;
        return (float*)g_side_buf[dev];
    }
    return mg_det_scratch(floats);
}

namespace {

constexpr int NT = 256;
constexpr int VPB = 16;                 // values per workgroup
constexpr int CH = NT / VPB;            // slot chunks per value (16)
static_assert(CH == MG_DET_CHUNKS, "det_chunk_sum's users chunk the rows the way this kernel does");

struct Segs { mg_det_seg s[MG_DET_MAX_SEGS]; int n; };

// One workgroup: VPB consecutive columns of one group; thread (v = t % 16, k = t / 16) adds the slots of chunk k in index order (four
// independent running sums over slot index mod 4, combined in a fixed order: the loads of a chunk are independent, the arithmetic is a
// function of (nblk) alone), the 16 chunk sums meet in LDS and are added in chunk order.
__global__ __launch_bounds__(NT) void det_reduce_kernel(const float* __restrict__ slots, int nblk, int rowstride, int col0, int ncols, const Segs segs) {
    __shared__ float sh[CH][VPB + 1];
    const int v = threadIdx.x & (VPB - 1), k = threadIdx.x / VPB;
    const int col = blockIdx.x * VPB + v;                      // relative to col0
    const int g = blockIdx.y;
    const int cs = (nblk + CH - 1) / CH;
    const int b0 = k * cs, b1 = min(nblk, b0 + cs);
    float a = 0.f;
    if (col < ncols && b0 < b1) a = det_chunk_sum(slots + ((size_t)g * nblk + b0) * rowstride + col0 + col, b1 - b0, (size_t)rowstride);
    sh[k][v] = a;
    __syncthreads();
    if (k == 0 && col < ncols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) t += sh[i][v];
        int c = col;
#pragma unroll
        for (int s = 0; s < MG_DET_MAX_SEGS; ++s) {
            if (s < segs.n) {
                if (c >= 0 && c < segs.s[s].nv) {
                    float* d = segs.s[s].dst + (size_t)g * segs.s[s].group_stride + c;
                    *d += t;
                }
                c -= segs.s[s].nv;
            }
        }
    }
}

}  // namespace

int mg_det_reduce(const float* slots, int nblk, int groups, int rowstride, int col0, const mg_det_seg* segs, int nseg, hipStream_t st) {
    if (nseg < 1 || nseg > MG_DET_MAX_SEGS || nblk < 1 || groups < 1) return -2;
    Segs sg;
    sg.n = nseg;
    int ncols = 0;
    for (int i = 0; i < nseg; ++i) { sg.s[i] = segs[i]; ncols += segs[i].nv; }
    for (int i = nseg; i < MG_DET_MAX_SEGS; ++i) sg.s[i] = mg_det_seg{nullptr, 0, 0};
    if (ncols <= 0) return 0;
    if (col0 < 0 || col0 + ncols > rowstride) return -2;
    hipLaunchKernelGGL(det_reduce_kernel, dim3((ncols + VPB - 1) / VPB, groups), dim3(NT), 0, st, slots, nblk, rowstride, col0, ncols, sg);
    MG_CHECK_LAUNCH();
    return 0;
}

/* Test hook: dst[g][c] += sum over the nblk slots, through the kernel above (tests/test_gpu_kernels.py compares it with a host sum in
 * the same order and with itself across runs). */
extern "C" int mg_det_reduce_test(const float* slots, int nblk, int groups, int nv, float* dst, void* stream) {
    mg_det_seg s{dst, nv, (long)nv};
    return mg_det_reduce(slots, nblk, groups, nv, 0, &s, 1, (hipStream_t)stream);
}
