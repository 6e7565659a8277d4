// Weight bank of the sparse refinement head: every per-step layout/dtype conversion of its ~30 small parameters in ONE launch
// per direction (the per-parameter cast / pad / permute / flip chains were ~90 launches of a few microseconds each, on the
// host-paced part of the step).
//   forward : fp32 parameter (Cout, taps, Cin) [spconv layout; nn.Linear = taps 1; a bias = (1, 1, C)]
//             -> (Cout_pad, taps, Cin_pad) in the compute dtype, zero padded            (operand of the forward gather-GEMM)
//             -> (Cin_pad, taps', Cout_pad), taps' reversed for submanifold convs        (operand of the input-gradient GEMM)
//   backward: the weight-gradient GEMM's (Cout_pad, taps, Cin_pad) result -> fp32 gradient in the parameter's own layout
// Entries travel by value in the kernel arguments (no descriptor table in HBM, no host->device copy per step).
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

struct WbArgs { mg_wb_entry e[MG_WB_MAX_ENTRIES]; };

__device__ __forceinline__ float wb_ld(const void* p, long i, int dt) {
    return dt == MG_BF16 ? bf2f(((const bf16raw*)p)[i]) : dt == MG_F16 ? ElemTraits<f16raw>::ld((const f16raw*)p + i) : ((const float*)p)[i];
}
__device__ __forceinline__ void wb_st(void* p, long i, int dt, float v) {
    if (dt == MG_BF16) ((bf16raw*)p)[i] = f2bf(v); else if (dt == MG_F16) ElemTraits<f16raw>::st((f16raw*)p + i, v); else ((float*)p)[i] = v;
}

__global__ __launch_bounds__(256) void wb_fwd_kernel(const WbArgs a) {
    const mg_wb_entry& e = a.e[blockIdx.y];
    const int total = e.cout_pad * e.taps * e.cin_pad;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % e.cin_pad, r = i / e.cin_pad, tap = r % e.taps, co = r / e.taps;
        const float v = (co < e.cout && ci < e.cin) ? ((const float*)e.src)[((long)co * e.taps + tap) * e.cin + ci] : 0.f;
        wb_st(e.dst, i, e.dtype, v);
        if (e.dst_t) wb_st(e.dst_t, ((long)ci * e.taps + (e.flip_t ? e.taps - 1 - tap : tap)) * e.cout_pad + co, e.dtype, v);
    }
}

__global__ __launch_bounds__(256) void wb_bwd_kernel(const WbArgs a) {
    const mg_wb_entry& e = a.e[blockIdx.y];
    const int total = e.cout * e.taps * e.cin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % e.cin, r = i / e.cin;                 // r = co * taps + tap
        ((float*)e.dst)[i] = e.src ? wb_ld(e.src, (long)r * e.cin_pad + ci, e.dtype) : 0.f;
    }
}

}  // namespace

extern "C" int mg_weight_bank(const mg_wb_entry* entries, int n, int backward, void* stream) {
    if (n <= 0) return 0;
    if (n > MG_WB_MAX_ENTRIES) return (int)hipErrorInvalidValue;
    WbArgs a;
    int biggest = 1;
    for (int i = 0; i < n; ++i) {
        a.e[i] = entries[i];
        const int t = entries[i].cout_pad * entries[i].taps * entries[i].cin_pad;
        biggest = t > biggest ? t : biggest;
    }
    int bx = (biggest + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : (bx > 32 ? 32 : bx);
    if (backward) hipLaunchKernelGGL(wb_bwd_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(wb_fwd_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
