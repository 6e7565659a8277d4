#include "common.h"
#include "../../include/maggie_hip.h"
extern "C" int mg_abi_version(void) { return 1; }

#include <utility>
#include <vector>

char* mg_zeroed_lo = nullptr;
char* mg_zeroed_hi = nullptr;
static std::vector<std::pair<char*, char*>> g_claims;      // the slices of the range that an accumulator has already been "cleared" in
static int g_conflicts = 0;
/* [base, base + bytes) is zero and every part of it is handed to at most one accumulator before it is zeroed again (NULL / 0: none).
 * mg_zero_words() inside the library then skips its fill launch for buffers inside the range. Single-threaded use (the capturing thread). */
extern "C" int mg_set_zeroed_range(void* base, long bytes) {
    mg_zeroed_lo = (char*)base;
    mg_zeroed_hi = base ? (char*)base + bytes : nullptr;
    g_claims.clear();
    return 0;
}
/* The invariant behind the skipped fills, checked: a second "clear" of words that an earlier call already claimed inside the current range means
 * that an accumulator is being re-zeroed after use (or two callees share a slice) -- under capture the fill would be skipped and stale sums would
 * survive. The skip is then refused (the caller's entry point fails with hipErrorAlreadyMapped) instead of trusted. */
extern "C" int mg_zero_claim(void* p, long bytes) {
    char* lo = (char*)p;
    char* hi = lo + bytes;
    for (const auto& c : g_claims)
        if (lo < c.second && c.first < hi) { ++g_conflicts; return 1; }
    g_claims.emplace_back(lo, hi);
    return 0;
}
extern "C" int mg_zeroed_range_conflicts(void) { const int n = g_conflicts; g_conflicts = 0; return n; }


// ---- up to 16 device-to-device copies as ONE launch (round 5). The step hands tensors across graph boundaries -- what the caller receives from a
// replayed graph (four 42 MB alpha planes + the index map), the detail graph's input gradients into the trunk graph's gradient slots -- and
// torch._foreach_copy_ of contiguous same-type tensors turns into one hipMemcpyAsync per tensor: 5 + 7 copy kernels of 5-25 us per step, each with
// its own launch floor and ramp. Here the jobs share one grid: a workgroup walks the concatenated 16-byte chunk space (sizes / addresses that are
// not 16-byte multiples take a byte tail), 2048 workgroups stream all buffers at once.
namespace {
struct CopyJobs { const char* src[16]; char* dst[16]; long bytes[16]; long first[17]; int k; };      // first[j]: first 4 KiB block of job j
__global__ __launch_bounds__(256) void copy_k_kernel(const CopyJobs c) {
    const long nblk = c.first[c.k];
    for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
        int j = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) if (q < c.k && b >= c.first[q]) j = q;
        const long off = (b - c.first[j]) * 4096;
        const long n = c.bytes[j] - off < 4096 ? c.bytes[j] - off : 4096;
        const char* s = c.src[j] + off;
        char* d = c.dst[j] + off;
        if (n == 4096 && (((size_t)s | (size_t)d) & 15) == 0) {
            ((uint4*)d)[threadIdx.x] = ((const uint4*)s)[threadIdx.x];
        } else {
            for (long i = threadIdx.x; i < n; i += 256) d[i] = s[i];
        }
    }
}
}  // namespace
extern "C" int mg_copy_k(const void* const* srcs, void* const* dsts, const long* bytes, int k, void* stream) {
    if (k < 0 || k > 16 || (k > 0 && (!srcs || !dsts || !bytes))) return -2;
    CopyJobs c;
    long blocks = 0;
    int m = 0;
    for (int j = 0; j < k; ++j) {
        if (bytes[j] < 0 || (bytes[j] > 0 && (!srcs[j] || !dsts[j]))) return -2;
        if (bytes[j] == 0 || srcs[j] == dsts[j]) continue;
        c.src[m] = (const char*)srcs[j]; c.dst[m] = (char*)dsts[j]; c.bytes[m] = bytes[j]; c.first[m] = blocks;
        blocks += (bytes[j] + 4095) / 4096;
        ++m;
    }
    if (m == 0) return 0;
    for (int j = m; j < 16; ++j) { c.src[j] = nullptr; c.dst[j] = nullptr; c.bytes[j] = 0; c.first[j] = blocks; }
    c.first[m] = blocks; c.first[16] = blocks; c.k = m;
    const long grid = blocks < 4096 ? blocks : 4096;
    hipLaunchKernelGGL(copy_k_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, c);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---- the step's device -> host flags in ONE launch ------------------------------------------------------------------------------------------
// Between the trunk and the detail stage the host needs four words (maggie_amd/network/arch/maggie.py:_forward_impl): NaN among the instance
// tokens (the reference raises, mask_attention.py:95-98), coarse alpha identically zero, the sparse head's sticky overflow word, the SyncBatchNorm
// mailbox's error word. Formed by isnan + any + two compares + a stack they were five small launches in front of the copy the host waits for.
namespace {
__global__ __launch_bounds__(256) void step_flags_kernel(const float* __restrict__ tokens, long n, const int* __restrict__ nonzero, const int* __restrict__ ovf,
                                                         const int* __restrict__ err, int* __restrict__ out) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    for (long i = threadIdx.x; i < n; i += 256) { const float v = tokens[i]; mine |= (v != v); }
    if (mine) bad = 1;                                       // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = bad;
        out[1] = nonzero ? (nonzero[0] == 0) : 0;
        out[2] = ovf ? (ovf[0] != 0) : 0;
        out[3] = err ? err[0] : 0;
    }
}
}  // namespace
extern "C" int mg_step_flags(const float* tokens, long n, const int* nonzero, const int* ovf, const int* err, int* out, void* stream) {
    if (!out || n < 0 || (n > 0 && !tokens)) return -2;
    hipLaunchKernelGGL(step_flags_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tokens, n, nonzero, ovf, err, out);
    MG_CHECK_LAUNCH();
    return 0;
}

// ---- any number of device-to-device copies as ONE launch, the job list in device memory ------------------------------------------------------------
// table: int64[4 * jobs] = (src, dst, bytes, first 4 KiB block) per job, first blocks ascending; nblk = total blocks. A captured backward graph ends by
// moving the ~190 small parameter gradients nobody wrote in place (BatchNorm weights, biases, token-side matrices) into the optimizer's flat buffer:
// torch._foreach_copy_ takes three multi-tensor launches (31 us) for them.
namespace {
__global__ __launch_bounds__(256) void copy_table_kernel(const long* __restrict__ table, int jobs, long nblk) {
    for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
        int lo = 0, hi = jobs - 1;                               // last job whose first block is <= b
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (b >= table[4 * mid + 3]) lo = mid; else hi = mid - 1;
        }
        const long bytes = table[4 * lo + 2];
        const long off = (b - table[4 * lo + 3]) * 4096;
        const long n = bytes - off < 4096 ? bytes - off : 4096;
        const char* s = (const char*)table[4 * lo] + off;
        char* d = (char*)table[4 * lo + 1] + off;
        if (n == 4096 && (((size_t)s | (size_t)d) & 15) == 0) {
            ((uint4*)d)[threadIdx.x] = ((const uint4*)s)[threadIdx.x];
        } else if (((n | (long)(size_t)s | (long)(size_t)d) & 3) == 0) {
            for (long i = threadIdx.x; i < n / 4; i += 256) ((uint32_t*)d)[i] = ((const uint32_t*)s)[i];
        } else {
            for (long i = threadIdx.x; i < n; i += 256) d[i] = s[i];
        }
    }
}
}  // namespace
extern "C" int mg_copy_table(const long* table, int jobs, long nblk, void* stream) {
    if (jobs <= 0 || nblk <= 0) return 0;
    if (!table) return -2;
    const long grid = nblk < 2048 ? nblk : 2048;
    hipLaunchKernelGGL(copy_table_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, table, jobs, nblk);
    MG_CHECK_LAUNCH();
    return 0;
}
