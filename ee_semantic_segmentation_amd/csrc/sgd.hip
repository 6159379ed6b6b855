// Multi-tensor SGD(momentum, weight_decay) step - one launch for every parameter.
// torch.optim.SGD semantics (deepv3_funcs.py:99; train_funcs.py:27), dampening 0,
// no nesterov: g = grad*grad_scale + wd*p; buf = first ? g : mu*buf + g; p -= lr*buf.
#include "eeseg_common.h"

namespace {
__global__ __launch_bounds__(256) void sgd_kernel(void* const* __restrict__ ptrs, const int64_t* __restrict__ sizes,
                                                  const float* __restrict__ lrs, float momentum, float wd,
                                                  float grad_scale, int first_step) {
    const int t = blockIdx.y;
    float* p = (float*)ptrs[3 * t + 0];
    const float* g = (const float*)ptrs[3 * t + 1];
    float* b = (float*)ptrs[3 * t + 2];
    if (g == nullptr) return;                     // parameter without gradient this step
    const long long n = sizes[t];
    const float lr = lrs[t];
    const long long n4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)b) & 15) == 0 ? n / 4 : 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 bv;
        if (!first_step) bv = reinterpret_cast<f32x4*>(b)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] * grad_scale + wd * pv[e];
            bv[e] = first_step ? gg : momentum * bv[e] + gg;
            pv[e] -= lr * bv[e];
        }
        reinterpret_cast<f32x4*>(b)[i] = bv;
        reinterpret_cast<f32x4*>(p)[i] = pv;
    }
    for (long long i = n4 * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float gg = g[i] * grad_scale + wd * p[i];
        const float bb = first_step ? gg : momentum * b[i] + gg;
        b[i] = bb;
        p[i] -= lr * bb;
    }
}
}  // namespace

extern "C" int eeseg_sgd_step(void* const* param_grad_buf, const int64_t* sizes, const float* lrs, int n,
                              float momentum, float weight_decay, float grad_scale, int first_step, void* stream) {
    EESEG_CHECK(param_grad_buf && sizes && lrs && n > 0, EESEG_ERR_ARG, "sgd_step: bad argument");
    EESEG_CHECK(n <= 65535, EESEG_ERR_ARG, "sgd_step: too many tensors (%d)", n);
    hipLaunchKernelGGL(sgd_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, param_grad_buf, sizes, lrs, momentum,
                       weight_decay, grad_scale, first_step);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
