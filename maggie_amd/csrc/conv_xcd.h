// XCD-aware work order shared by the convolution translation units (conv_igemm.hip: fprop / dgrad family; conv_wgrad.hip: weight gradients).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// XCD-aware work order. The dispatcher places workgroup b on XCD b % 8 (8 XCDs, private 4 MiB L2 each), so consecutive
// block ids never share an L2. Launch 8 * ceil(L/8) blocks and give XCD x the CONTIGUOUS work items [x*chunk, (x+1)*chunk):
// neighbouring tiles (which share input halos, or the same rows under different filter taps) then hit the same L2 close in
// time. Returns false for the <= 7 padding blocks. (A speed choice only: any placement is correct.)
constexpr int NXCD = 8;
__device__ __forceinline__ bool xcd_order(int L, int& v) {
    const int chunk = (L + NXCD - 1) / NXCD;
    v = ((int)blockIdx.x % NXCD) * chunk + (int)blockIdx.x / NXCD;
    return v < L;
}
static inline unsigned xcd_grid(long L) { return (unsigned)(((L + NXCD - 1) / NXCD) * NXCD); }

}  // namespace
