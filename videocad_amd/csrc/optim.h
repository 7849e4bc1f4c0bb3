// optim.h — clip_grad_norm_(1.0) + Adam over ONE flat fp32 parameter buffer
// (reference trainer.py:493-494; torch.optim.Adam defaults, lr from main.py:80).
//
// HBM-bound: the norm pass reads 4 B/param, the Adam pass reads p,g,m,v and writes p,m,v (28 B/param)
// plus the optional bf16 weight shadow (2 B/param).  The clip coefficient stays on the device
// (norm_out[1]), so there is no host sync between backward and the update.
#pragma once
#include "vc_rt.h"

VC_KERNEL __launch_bounds__(256) void sumsq_stage1_kernel(const float* g, long n, float* partial) {
    VC_SHARED float red[256];
    float s = 0.f;
    // one 16-byte load per thread and trip (the per-element bound check of the first version made these four 4-byte loads at a 16-byte lane stride: 2.8 TB/s);
    // same elements per thread in the same order, so the sum is bit-identical
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const vc_u32x4 q = *reinterpret_cast<const vc_u32x4*>(g + i);
            const float v4[4] = {vc_bits_f32(q.x), vc_bits_f32(q.y), vc_bits_f32(q.z), vc_bits_f32(q.w)};
#pragma unroll
            for (int j = 0; j < 4; ++j) { float v = v4[j]; s += v * v; }
        } else {
            for (int j = 0; j < 4; ++j) if (i + j < n) { float v = g[i + j]; s += v * v; }
        }
    }
    red[threadIdx.x] = s;
    vc_sync();
    for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; vc_sync(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// norm_out[0] = total L2 norm, norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
VC_KERNEL __launch_bounds__(256) void sumsq_stage2_kernel(const float* partial, int nblk, float max_norm, float gscale, float* norm_out) {
    VC_SHARED double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    vc_sync();
    for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; vc_sync(); }
    if (threadIdx.x == 0) {
        float nrm = (float)sqrt(red[0]) * gscale;      // norm of the averaged gradient (gscale = 1/world)
        float coef = max_norm / (nrm + 1e-6f);
        norm_out[0] = nrm; norm_out[1] = coef < 1.0f ? coef : 1.0f;
        norm_out[2] = (nrm == nrm && nrm <= 3.0e38f) ? 1.0f : 0.0f;      // finite?  (an overflowed fp16 backward: the update is skipped, adam_kernel)
    }
}

struct AdamParams {
    float* p; const float* g; float* m; float* v; long n;
    float lr, beta1, beta2, eps, bc1, bc2;      // bc = 1 - beta^t
    const float* clip;                          // device scalar (norm_out + 1) or null
    float gscale;                               // extra gradient scale (1/world for DDP sum -> mean)
    const float* finite;                        // device scalar (norm_out + 2): 0 = the gradient norm is inf / NaN -> leave p, m, v alone
    vc_bf16* shadow;                            // optional bf16 copy of p (same flat offsets)
    uint32_t* shadow_pk;                        // optional pre-split (hi | lo bf16) copy of p for the bf16x3 GEMMs (gemm.h vc_pk)
    float* gw; float gw_mul;                    // optional (fp16 engines, deferred unscale): g[i] is written back as g[i] * gw_mul — the buffer holds true gradients again after the step
};
VC_KERNEL __launch_bounds__(256) void adam_kernel(AdamParams a) {
    if (a.finite && a.finite[0] == 0.0f) {       // skipped update: only the write-back (the values are those of an overflowed backward either way)
        if (a.gw) for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) a.gw[i] = a.g[i] * a.gw_mul;
        return;
    }
    const float c = (a.clip ? a.clip[0] : 1.0f) * a.gscale;
    const float step = a.lr / a.bc1, rs2 = 1.0f / sqrtf(a.bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
        const float g0 = a.g[i], g = g0 * c;
        if (a.gw) a.gw[i] = g0 * a.gw_mul;
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
        const float p = a.p[i] - step * (m / (sqrtf(v) * rs2 + a.eps));
        a.m[i] = m; a.v[i] = v; a.p[i] = p;
        if (a.shadow) a.shadow[i] = vc_f32_to_bf16(p);
        if (a.shadow_pk) a.shadow_pk[i] = vc_pk_pack(p);
    }
}
