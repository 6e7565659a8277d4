// Input side of the hot path: what the reference's DataLoader does on CPU tensors after decoding --
//   ToTensor + Normalize   maggie/dataloader/transforms.py:720-778  (HWC uint8 -> CHW fp32, /255, (x - mean) / std; alphas < 5 -> 0)
//   item assembly          maggie/dataloader/him.py:157-173          (alpha, mask / 255; scatter into the max_inst slots; nearest
//                                                                      downscale of the masks to H/8 x W/8)
// -- as two HBM-bound kernels fed from uint8 buffers (4x fewer bytes over PCIe than the fp32 tensors the reference ships).
// Arithmetic is the reference's, operation for operation (IEEE fp32 divisions, not reciprocal multiplies): bit-exact.
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

struct Norm3 { float mean[3], std[3]; };

// 4 pixels per thread: 12 contiguous bytes in, one float4 per channel plane out
__global__ __launch_bounds__(256) void preprocess_image_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, Norm3 nm, long HW, long quads) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const long f = blockIdx.y;
    if (q >= quads) return;
    const long p = q * 4;
    const uint8_t* src = in + (f * HW + p) * 3;
    float* dst = out + f * 3 * HW + p;
    if (p + 4 <= HW) {
        const uint32_t* s32 = (const uint32_t*)src;          // (f*HW + p)*3 is a multiple of 4 when HW % 4 == 0 (checked by the host)
        const uint32_t w0 = s32[0], w1 = s32[1], w2 = s32[2];
        uint8_t b[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) { b[i] = (w0 >> (8 * i)) & 255; b[4 + i] = (w1 >> (8 * i)) & 255; b[8 + i] = (w2 >> (8 * i)) & 255; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float4 v;
            v.x = __fdiv_rn(__fdiv_rn((float)b[c], 255.0f) - nm.mean[c], nm.std[c]);
            v.y = __fdiv_rn(__fdiv_rn((float)b[3 + c], 255.0f) - nm.mean[c], nm.std[c]);
            v.z = __fdiv_rn(__fdiv_rn((float)b[6 + c], 255.0f) - nm.mean[c], nm.std[c]);
            v.w = __fdiv_rn(__fdiv_rn((float)b[9 + c], 255.0f) - nm.mean[c], nm.std[c]);
            *(float4*)(dst + c * HW) = v;
        }
    } else {
        for (long i = p; i < HW; ++i)
            for (int c = 0; c < 3; ++c)
                out[f * 3 * HW + c * HW + i] = __fdiv_rn(__fdiv_rn((float)in[(f * HW + i) * 3 + c], 255.0f) - nm.mean[c], nm.std[c]);
    }
}

// out[f][slot][y][x] = in[f][src][sy][sx] / 255 (0 when below `thresh` or when the slot is empty); sy/sx: torch 'nearest'
__global__ __launch_bounds__(256) void preprocess_planes_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                                const int32_t* __restrict__ src_of_slot, int n_in, int n_slots, int H, int W,
                                                                int Ho, int Wo, int thresh) {
    const int fs = blockIdx.y;                                // frame * n_slots + slot
    const int f = fs / n_slots;
    const int src = src_of_slot ? src_of_slot[fs] : (fs - f * n_slots);
    const long n = (long)Ho * Wo;
    float* o = out + (long)fs * n;
    const bool empty = src < 0 || src >= n_in;
    const uint8_t* plane = in + ((long)f * n_in + (empty ? 0 : src)) * H * W;
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = 0.f;
        if (!empty) {
            const int y = (int)(i / Wo), x = (int)(i - (long)y * Wo);
            const int sy = (Ho == H) ? y : min((int)floorf(y * sh), H - 1);
            const int sx = (Wo == W) ? x : min((int)floorf(x * sw), W - 1);
            const int u = plane[(long)sy * W + sx];
            v = u < thresh ? 0.f : __fdiv_rn((float)u, 255.0f);
        }
        o[i] = v;
    }
}

}  // namespace

extern "C" int mg_preprocess_image(const uint8_t* in, float* out, const float* mean3, const float* std3, long frames, long HW, void* stream) {
    if (frames <= 0 || HW <= 0) return 0;
    if (HW % 4 || frames > 65535) return -3;
    Norm3 nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.std[c] = std3[c]; }
    const long quads = HW / 4;
    hipLaunchKernelGGL(preprocess_image_kernel, dim3((unsigned)((quads + 255) / 256), (unsigned)frames), dim3(256), 0, (hipStream_t)stream, in, out, nm, HW, quads);
    return (int)hipGetLastError();
}

extern "C" int mg_preprocess_planes(const uint8_t* in, float* out, const int32_t* src_of_slot, int frames, int n_in, int n_slots, int H, int W,
                                    int Ho, int Wo, int thresh, void* stream) {
    if (frames <= 0 || n_slots <= 0 || Ho <= 0 || Wo <= 0) return 0;
    if ((long)frames * n_slots > 65535 || n_in <= 0) return -3;
    const long n = (long)Ho * Wo;
    long bx = (n + 256 * 8 - 1) / (256 * 8);
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    hipLaunchKernelGGL(preprocess_planes_kernel, dim3((unsigned)bx, (unsigned)(frames * n_slots)), dim3(256), 0, (hipStream_t)stream, in, out, src_of_slot,
                       n_in, n_slots, H, W, Ho, Wo, thresh);
    return (int)hipGetLastError();
}
