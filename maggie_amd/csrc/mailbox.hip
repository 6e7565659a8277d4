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

__global__ __launch_bounds__(256) void mailbox_allreduce_kernel(const mg_mailbox mb, float* __restrict__ data, int n, uint32_t* __restrict__ seq_dev,
                                                                int32_t* __restrict__ err_dev, long spin_ticks) {
    __shared__ uint32_t seq_s;
    const int t = threadIdx.x;
    if (t == 0) { seq_s = *seq_dev + 1u; *seq_dev = seq_s; }
    __syncthreads();
    const uint32_t seq = seq_s;
    const int slot = (int)(seq % (uint32_t)SLOTS);
    const long mine = ((long)slot * mb.world + mb.rank) * CELL;                // my cell in every mailbox
    // 1. deposit
    for (int p = 0; p < mb.world; ++p) {
        float* dst = mb.peer[p] + mine;
        for (int i = t; i < n; i += 256) __hip_atomic_store(dst + i, data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
            if ((long)wall_clock64() - t0 > spin_ticks) { atomicExch(err_dev, 1); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __threadfence_system();
    __syncthreads();
    // 3. sum in rank order
    for (int i = t; i < n; i += 256) {
        float a = 0.f;
        for (int r = 0; r < mb.world; ++r) a += __hip_atomic_load(own + (long)r * CELL + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        data[i] = a;
    }
}

}  // namespace

extern "C" long mg_mailbox_bytes(int world) { return (long)SLOTS * world * CELL * (long)sizeof(float); }

extern "C" int mg_mailbox_create(int world, void** ptr, void* handle64) {
    if (!ptr || !handle64 || world < 1 || world > MG_MAILBOX_MAX_RANKS) return -2;
    const size_t bytes = (size_t)mg_mailbox_bytes(world);
    hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained);      // peers poll it: must not be cached incoherently
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(ptr, bytes); }
    if (e != hipSuccess) return (int)e;
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
    if (!mb || !data || !seq_dev || !err_dev) return -1;
    if (n < 1 || n > PACK || mb->world < 1 || mb->world > MG_MAILBOX_MAX_RANKS || mb->rank < 0 || mb->rank >= mb->world) return -2;
    hipLaunchKernelGGL(mailbox_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *mb, data, n, seq_dev, err_dev, spin_ticks);
    MG_CHECK_LAUNCH();
    return 0;
}
