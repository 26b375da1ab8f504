// AdamW over ONE flat fp32 arena: parameters, gradients and both moments of all LoRA factors laid out back to back
// (unsloth_amd/optim.py FlatAdamW; gradients = dp.LoRAGradArena). One launch per optimizer step.
//
// Where it sits on the reference's path: the optimizer step that closes every training step of the benchmark
// (unsloth/trainer.py:445-623 builds torch / bitsandbytes optimizers through HF's Trainer; the tokens/s metric times
// forward + backward + this). torch's fused AdamW walks the 448 LoRA tensors in multi_tensor_apply chunks: 52 launches,
// 2.5 ms per step for 168 MB of parameters (profiles/r02z_bench_kernel_stats.csv) -- 8 streams of 168 MB are 0.22 ms of
// HBM time. The arithmetic is torch.optim.AdamW's (decoupled weight decay, bias-corrected moments; fp32 throughout):
//     p  -= lr * wd * p
//     m   = b1 * m + (1 - b1) * g
//     v   = b2 * v + (1 - b2) * g * g
//     p  -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)            bc1 = 1 - b1^t, bc2 = 1 - b2^t
// `grad_scale` multiplies g first (gradient clipping: the caller passes clip / norm, 1 = none); `zero_grad` != 0 writes
// zeros back into g in the same pass (the arena is ADDED into by uamd_lora_tn, so it must start every step at zero:
// this replaces a separate 168 MB fill).
// HBM-bound: 16 B read + 12 (16 with zero_grad) B written per parameter.
#include "common.h"

namespace {

struct AdamArgs {
    float* p; float* g; float* m; float* v;
    int64_t n;
    float lr_wd, b1, b2, omb1, omb2, eps, step_size, bc2_sqrt, grad_scale;
    int zero_grad;
};

__device__ __forceinline__ void adamw_one(float& p, float& g, float& m, float& v, const AdamArgs& a) {
    const float gr = g * a.grad_scale;
    // same association as torch's fused kernel (fused_adam_utils.cuh adam_math): ... - step_size * m / denom
    p = p - a.lr_wd * p;
    m = a.b1 * m + a.omb1 * gr;
    v = a.b2 * v + a.omb2 * gr * gr;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * m / denom;
}

__global__ void __launch_bounds__(256) adamw_flat_kernel(AdamArgs a) {
    const int64_t n4 = a.n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<const float4*>(a.p)[i];
        float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<const float4*>(a.m)[i];
        float4 v = reinterpret_cast<const float4*>(a.v)[i];
        adamw_one(p.x, g.x, m.x, v.x, a);
        adamw_one(p.y, g.y, m.y, v.y, a);
        adamw_one(p.z, g.z, m.z, v.z, a);
        adamw_one(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.zero_grad) reinterpret_cast<float4*>(a.g)[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    // tail (n % 4 elements): first threads of block 0
    const int64_t t = (n4 << 2) + threadIdx.x;
    if (blockIdx.x == 0 && t < a.n) {
        adamw_one(a.p[t], a.g[t], a.m[t], a.v[t], a);
        if (a.zero_grad) a.g[t] = 0.f;
    }
}

// Mixed-precision form for FULL fine-tuning (unsloth_amd/full_finetune.py): the fp32 master copy + moments of one
// rank's SHARD of a flat bucket, the gradient shard in the model's 16-bit dtype (what the reduce-scatter delivered), and
// the updated parameters written back in the 16-bit dtype into the bucket slice the all-gather then broadcasts.
// 14 B read + 14 B written per parameter.
template <typename T>
__global__ void __launch_bounds__(256) adamw_shard_kernel(AdamArgs a, const T* __restrict__ g16, T* __restrict__ p16) {
    const int64_t n4 = a.n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<const float4*>(a.p)[i];
        float4 m = reinterpret_cast<const float4*>(a.m)[i];
        float4 v = reinterpret_cast<const float4*>(a.v)[i];
        union { uint2 raw; T e[4]; } gi, po;
        gi.raw = reinterpret_cast<const uint2*>(g16)[i];
        float g0 = to_f32(gi.e[0]), g1 = to_f32(gi.e[1]), g2 = to_f32(gi.e[2]), g3 = to_f32(gi.e[3]);
        adamw_one(p.x, g0, m.x, v.x, a);
        adamw_one(p.y, g1, m.y, v.y, a);
        adamw_one(p.z, g2, m.z, v.z, a);
        adamw_one(p.w, g3, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        po.e[0] = from_f32<T>(p.x); po.e[1] = from_f32<T>(p.y); po.e[2] = from_f32<T>(p.z); po.e[3] = from_f32<T>(p.w);
        reinterpret_cast<uint2*>(p16)[i] = po.raw;
    }
    const int64_t t = (n4 << 2) + threadIdx.x;
    if (blockIdx.x == 0 && t < a.n) {
        float g = to_f32(g16[t]);
        adamw_one(a.p[t], g, a.m[t], a.v[t], a);
        p16[t] = from_f32<T>(a.p[t]);
    }
}

}  // namespace

extern "C" int uamd_adamw_shard(float* p32, const void* g16, void* p16, float* m, float* v, int64_t n, double lr,
                                double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                                double bias_correction2_sqrt, double grad_scale, int dtype, void* stream) {
    if (!p32 || !g16 || !p16 || !m || !v || n < 0) return UAMD_ERR_ARG;
    if (!(bias_correction1 > 0.0) || !(bias_correction2_sqrt > 0.0)) return UAMD_ERR_ARG;
    if (!aligned16(p32) || !aligned16(m) || !aligned16(v) || (reinterpret_cast<uintptr_t>(g16) & 7) ||
        (reinterpret_cast<uintptr_t>(p16) & 7))
        return UAMD_ERR_ALIGN;
    if (n == 0) return UAMD_OK;
    AdamArgs a;
    a.p = p32; a.g = nullptr; a.m = m; a.v = v; a.n = n;
    a.lr_wd = (float)(lr * weight_decay); a.b1 = (float)beta1; a.b2 = (float)beta2;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.step_size = (float)(lr / bias_correction1); a.bc2_sqrt = (float)bias_correction2_sqrt;
    a.grad_scale = (float)grad_scale; a.zero_grad = 0;
    const int64_t n4 = (n + 3) >> 2;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UAMD_BF16)
        hipLaunchKernelGGL((adamw_shard_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, a, (const bf16_t*)g16, (bf16_t*)p16);
    else if (dtype == UAMD_F16)
        hipLaunchKernelGGL((adamw_shard_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, st, a, (const f16_t*)g16, (f16_t*)p16);
    else
        return UAMD_ERR_DTYPE;
    return uamd_launch_status();
}

extern "C" int uamd_adamw_flat(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                               double eps, double weight_decay, double bias_correction1, double bias_correction2_sqrt,
                               double grad_scale, int zero_grad, void* stream) {
    if (!p || !g || !m || !v || n < 0) return UAMD_ERR_ARG;
    if (!(bias_correction1 > 0.0) || !(bias_correction2_sqrt > 0.0)) return UAMD_ERR_ARG;
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return UAMD_ERR_ALIGN;
    if (n == 0) return UAMD_OK;
    AdamArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n;
    // hyper-parameters arrive as doubles (Python floats) and every derived constant is formed in double, then rounded
    // once -- like torch, whose kernels receive 1 - beta as a double-computed scalar (1.0f - 0.999f is off by 1.3e-5)
    a.lr_wd = (float)(lr * weight_decay); a.b1 = (float)beta1; a.b2 = (float)beta2;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.step_size = (float)(lr / bias_correction1); a.bc2_sqrt = (float)bias_correction2_sqrt;
    a.grad_scale = (float)grad_scale; a.zero_grad = zero_grad;
    const int64_t n4 = (n + 3) >> 2;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;            // 16 blocks per CU, grid-stride beyond
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return uamd_launch_status();
}
