// Small-message all-reduce over peer-mapped mailboxes: the SyncBatchNorm statistics exchange as ONE ordinary kernel per exchange (no RCCL, no
// host involvement, capturable into hipGraphs like any other kernel node). Replaces the per-layer collectives of nn.SyncBatchNorm
// (engine/train.py:159-161, sync_bn: true in configs/maggie_{image,video}.yaml) for packs of <= MG_MAILBOX_PACK floats ([sum x, sum x^2, n]).
//
// Every rank owns a mailbox [SLOTS][world][CELL] (fine-grained device memory, exported with hipIpcGetMemHandle and opened by every peer).
// Exchange number `seq` (a per-rank device counter, so a captured kernel advances it on every replay) uses slot seq % SLOTS:
//   1. the rank writes its pack into cell [slot][rank] of EVERY rank's mailbox, fences (system scope), then writes `seq` into the cell's flag word;
//   2. it waits until the flags of all `world` cells of slot `seq` in ITS OWN mailbox read `seq` (bounded spin: a peer that never arrives raises
//      the error word instead of hanging the GPU);
//   3. it sums the `world` packs in rank order -- the same order on every rank: bit-identical results everywhere.
// Slot reuse needs no acknowledgement: an exchange cannot finish before every rank has STARTED it, so when a rank comes back to a slot
// (SLOTS exchanges later) every peer has long finished reading it.
#include "common.h"
#include "../../include/maggie_hip.h"
#include <string.h>

namespace {

constexpr int SLOTS = MG_MAILBOX_SLOTS, PACK = MG_MAILBOX_PACK, CELL = MG_MAILBOX_PACK + 16;      // floats per cell: pack + flag word (+ padding to 64 B)

// deposit `n` values produced by `val(i)` into every rank's mailbox, wait for all ranks, hand the rank-ordered sums to `sink(i, sum)`
template <typename Val, typename Sink>
__device__ __forceinline__ void mailbox_exchange(const mg_mailbox& mb, int n, uint32_t* __restrict__ seq_dev, int32_t* __restrict__ err_dev, long spin_ticks,
                                                 Val val, Sink sink) {
    __shared__ uint32_t seq_s;
    const int t = threadIdx.x;
    if (t == 0) { seq_s = *seq_dev + 1u; *seq_dev = seq_s; }
    __syncthreads();
    const uint32_t seq = seq_s;
    const int slot = (int)(seq % (uint32_t)SLOTS);
    const long mine = ((long)slot * mb.world + mb.rank) * CELL;                // my cell in every mailbox
    // 1. deposit
    const int nt = (int)blockDim.x;
    for (int i = t; i < n; i += nt) {
        const float v = val(i);
        for (int p = 0; p < mb.world; ++p) __hip_atomic_store(mb.peer[p] + mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (t < mb.world) __hip_atomic_store((uint32_t*)(mb.peer[t] + mine + PACK), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. wait for everybody's deposit in MY mailbox
    float* own = mb.peer[mb.rank] + (long)slot * mb.world * CELL;
    if (t < mb.world) {
        const uint32_t* flag = (const uint32_t*)(own + (long)t * CELL + PACK);
        const long t0 = (long)wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if ((long)wall_clock64() - t0 > spin_ticks) { atomicCAS(err_dev, 0, 1 + t); break; }     // error word = 1 + the (first) rank that did not arrive
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __threadfence_system();
    __syncthreads();
    // 3. sum in rank order
    for (int i = t; i < n; i += nt) {
        float a = 0.f;
        for (int r = 0; r < mb.world; ++r) a += __hip_atomic_load(own + (long)r * CELL + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sink(i, a);
    }
}

__global__ __launch_bounds__(256) void mailbox_allreduce_kernel(const mg_mailbox mb, const float* __restrict__ src, float* __restrict__ dst, int n,
                                                                uint32_t* __restrict__ seq_dev, int32_t* __restrict__ err_dev, long spin_ticks) {
    mailbox_exchange(mb, n, seq_dev, err_dev, spin_ticks, [&](int i) { return src[i]; }, [&](int i, float a) { dst[i] = a; });
}

// SyncBatchNorm forward statistics in ONE launch: replica sums of this rank's [nrep][2C] statistics (sum x | sum x^2) + its row count ->
// exchange -> scale | shift | mean | invstd (outs[4C]), the global count (count_out) and the running statistics -- the arithmetic of
// bn_finalize_kernel (norm_act.hip) on the pooled moments. C <= (PACK - 1) / 2.
__global__ __launch_bounds__(1024) void mailbox_bn_finalize_kernel(const mg_mailbox mb, const float* __restrict__ stats, int nrep, float count, int C,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                                                   float* running_var, float momentum, float eps, float* __restrict__ outs,
                                                                   float* __restrict__ count_out, uint32_t* __restrict__ seq_dev, int32_t* __restrict__ err_dev,
                                                                   long spin_ticks, const int32_t* __restrict__ count_dev) {
    __shared__ float pooled[PACK];
    // replica sums first, all loads in flight at once (1024 threads, four independent accumulators): inside the deposit loop every element's 32
    // loads would wait behind the previous element's system-scope stores (19 us per launch, measured)
    for (int i = threadIdx.x; i < 2 * C; i += (int)blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = 0;
        for (; r + 4 <= nrep; r += 4) {
            a0 += stats[(size_t)r * 2 * C + i]; a1 += stats[(size_t)(r + 1) * 2 * C + i];
            a2 += stats[(size_t)(r + 2) * 2 * C + i]; a3 += stats[(size_t)(r + 3) * 2 * C + i];
        }
        for (; r < nrep; ++r) a0 += stats[(size_t)r * 2 * C + i];
        pooled[i] = (a0 + a1) + (a2 + a3);
    }
    if (threadIdx.x == 0) pooled[2 * C] = count_dev ? (float)count_dev[0] : count;      // (the sparse head's live-row count lives on the device)
    __syncthreads();
    mailbox_exchange(mb, 2 * C + 1, seq_dev, err_dev, spin_ticks, [&](int i) { return pooled[i]; }, [&](int i, float a) { pooled[i] = a; });
    __syncthreads();
    const float n = pooled[2 * C];
    if (threadIdx.x == 0) *count_out = n;
    for (int c = threadIdx.x; c < C; c += (int)blockDim.x) {
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        if (n <= 0.f) {                                          // no row on any rank: identity statistics, running statistics untouched
            outs[c] = g; outs[C + c] = b; outs[2 * C + c] = 0.f; outs[3 * C + c] = 1.f;
            continue;
        }
        const float mean = pooled[c] / n;
        float var = pooled[C + c] / n - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float invstd = rsqrtf(var + eps);
        outs[c] = g * invstd;
        outs[C + c] = b - mean * g * invstd;
        outs[2 * C + c] = mean;
        outs[3 * C + c] = invstd;
        if (running_mean) {
            const float unbiased = n > 1.f ? var * n / (n - 1.f) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

}  // namespace

extern "C" long mg_mailbox_bytes(int world) { return (long)SLOTS * world * CELL * (long)sizeof(float); }

extern "C" int mg_mailbox_create(int world, void** ptr, void* handle64) {
    if (!ptr || !handle64 || world < 1 || world > MG_MAILBOX_MAX_RANKS) return -2;
    const size_t bytes = (size_t)mg_mailbox_bytes(world);
    // peers poll it: it must not be cached incoherently. No coarse-grained fallback: across GPUs plain hipMalloc memory breaks the polling
    // protocol silently -- the caller falls back to the RCCL exchange instead (maggie_amd/parallel.py)
    hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    e = hipMemset(*ptr, 0, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 opaque bytes");
    return (int)hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, *ptr);
}

extern "C" int mg_mailbox_open(const void* handle64, void** ptr) {
    if (!handle64 || !ptr) return -2;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

extern "C" int mg_mailbox_close(void* ptr) { return ptr ? (int)hipIpcCloseMemHandle(ptr) : 0; }
extern "C" int mg_mailbox_free(void* ptr) { return ptr ? (int)hipFree(ptr) : 0; }

extern "C" int mg_mailbox_allreduce(const mg_mailbox* mb, float* data, int n, uint32_t* seq_dev, int32_t* err_dev, long spin_ticks, void* stream) {
    return mg_mailbox_allreduce_to(mb, data, data, n, seq_dev, err_dev, spin_ticks, stream);
}

extern "C" int mg_mailbox_allreduce_to(const mg_mailbox* mb, const float* src, float* dst, int n, uint32_t* seq_dev, int32_t* err_dev, long spin_ticks,
                                       void* stream) {
    if (!mb || !src || !dst || !seq_dev || !err_dev) return -1;
    if (n < 1 || n > PACK || mb->world < 1 || mb->world > MG_MAILBOX_MAX_RANKS || mb->rank < 0 || mb->rank >= mb->world) return -2;
    hipLaunchKernelGGL(mailbox_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *mb, src, dst, n, seq_dev, err_dev, spin_ticks);
    MG_CHECK_LAUNCH();
    return 0;
}

// count_dev (or NULL): this rank's row count as a device int32 (BatchNorm1d over the sparse head's live rows) -- `count` is then ignored
extern "C" int mg_mailbox_bn_finalize_dev(const mg_mailbox* mb, const float* stats, int nrep, float count, const int32_t* count_dev, int C,
                                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                          float* outs, float* count_out, uint32_t* seq_dev, int32_t* err_dev, long spin_ticks, void* stream) {
    if (!mb || !stats || !outs || !count_out || !seq_dev || !err_dev) return -1;
    if (C < 1 || 2 * C + 1 > PACK || nrep < 1 || mb->world < 1 || mb->world > MG_MAILBOX_MAX_RANKS || mb->rank < 0 || mb->rank >= mb->world) return -2;
    hipLaunchKernelGGL(mailbox_bn_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *mb, stats, nrep, count, C, gamma, beta, running_mean,
                       running_var, momentum, eps, outs, count_out, seq_dev, err_dev, spin_ticks, count_dev);
    MG_CHECK_LAUNCH();
    return 0;
}
extern "C" int mg_mailbox_bn_finalize(const mg_mailbox* mb, const float* stats, int nrep, float count, int C, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, float* outs, float* count_out,
                                      uint32_t* seq_dev, int32_t* err_dev, long spin_ticks, void* stream) {
    return mg_mailbox_bn_finalize_dev(mb, stats, nrep, count, nullptr, C, gamma, beta, running_mean, running_var, momentum, eps, outs, count_out, seq_dev,
                                      err_dev, spin_ticks, stream);
}
