// AdamW over one flat fp32 parameter buffer (torch.optim.AdamW semantics: decoupled weight decay, bias correction, eps added
// after the square root), with the gradient-norm clip of maggie/engine/train.py:274 folded in as a device-side scale:
//   g' = g * min(1, max_norm / (||g|| + 1e-6));  p *= 1 - lr*wd;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// One HBM pass: 4 streams in, 3 out (28 B per parameter; 30 M parameters = 0.84 GB ~ 0.1 ms at 8 TB/s). torch's fused multi-tensor
// AdamW needs 9 launches x 69 us for the same state plus 3 norm and 3 scale launches (0.84 ms measured).
#include "common.h"
#include "../../include/maggie_hip.h"

namespace {

// sumsq[0] += sum g^2 (fp32 partials per thread, fp64 atomics). Deterministic mode (`part` != NULL): one fp64 partial per workgroup, added
// in workgroup order by sumsq_finish_kernel -- the clip coefficient, and with it every parameter, is then bit-reproducible.
__global__ __launch_bounds__(256) void sumsq_kernel(const float4* __restrict__ g, long n4, double* __restrict__ out, double* __restrict__ part) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = g[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    double d = (double)acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (part) part[blockIdx.x] = v; else atomicAdd(out, v);
    }
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out += sh[0];
}

__global__ __launch_bounds__(256) void adamw_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v,
                                                    long n4, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const double* __restrict__ sumsq, float max_norm, float* __restrict__ norm_out) {
    float coef = 1.f;
    if (sumsq) {
        const float total = (float)sqrt(*sumsq);
        if (max_norm > 0.f) coef = fminf(1.f, max_norm / (total + 1e-6f));
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
    }
    const float decay = 1.f - lr * wd, step = lr / bc1, omb1 = 1.f - b1, omb2 = 1.f - b2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
#define MG_ADAM1(c)                                                        \
        {                                                                  \
            const float gg = G.c * coef;                                   \
            M.c = M.c + omb1 * (gg - M.c);                                 \
            V.c = V.c * b2 + omb2 * gg * gg;                               \
            P.c = P.c * decay - step * (M.c / (sqrtf(V.c) / bc2_sqrt + eps)); \
        }
        MG_ADAM1(x) MG_ADAM1(y) MG_ADAM1(z) MG_ADAM1(w)
#undef MG_ADAM1
        p[i] = P; m[i] = M; v[i] = V;
    }
}

}  // namespace

extern "C" int mg_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float bias_correction1, float bias_correction2_sqrt, double* sumsq_scratch, float max_norm,
                             float* norm_out, int phases, void* stream) {
    if (n <= 0) return 0;
    if (n % 4) return -3;
    hipStream_t st = (hipStream_t)stream;
    const long n4 = n / 4;
    long blocks = (n4 + 256 * 4 - 1) / (256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    if (sumsq_scratch && (phases & 1)) {
        hipError_t e = mg_zero_words(sumsq_scratch, 2, st);
        if (e != hipSuccess) return (int)e;
        const unsigned nb = (unsigned)(blocks > 1024 ? 1024 : blocks);
        double* part = nullptr;
        if (mg_det_on && nb > 1) { part = (double*)mg_det_scratch(2l * nb); if (!part) return MG_DET_NO_SCRATCH; }
        hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, (const float4*)g, n4, sumsq_scratch, part);
        if (part) hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)part, (int)nb, sumsq_scratch);
    }
    if (phases & 2)
        hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, n4, lr, beta1,
                       beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, sumsq_scratch, max_norm, norm_out);
    return (int)hipGetLastError();
}
