// shapegan_amd/csrc/elementwise.hip — HBM-bound helpers: activations (K5), optimizers + clamp (K11),
// latent-table gather / scatter-add (K10), reductions and losses (K9), lerp/fade blends (K8).
//
// Reference sites: nn.LeakyReLU/ReLU/Tanh/sigmoid throughout model/*.py; optim.RMSprop (train_wgan.py:45-46,
// train_hybrid_progressive_gan.py:81-82, train_hybrid_wgan.py:56), optim.Adam (train_autoencoder.py:35,
// train_sdf_autodecoder.py:44-45, train_hybrid_wgan.py:53) with torch defaults; Discriminator.clip_weights
// (model/gan.py:67-69); latent_codes[model_indices] (train_sdf_autodecoder.py:78-82).
// All kernels are grid-stride, float4 where alignment allows, one launch over a flat buffer.
#include "common.h"
#include "../../include/shapegan_hip.h"

namespace sg {

static int ew_grid(long n, int per_thread) {
    long b = (n + 256L * per_thread - 1) / (256L * per_thread);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ float act_grad_from_output(float y, float dy, int act, float slope) {
    switch (act) {
        case SG_ACT_LEAKY: return y > 0.f ? dy : dy * slope;
        case SG_ACT_RELU: return y > 0.f ? dy : 0.f;
        case SG_ACT_TANH: return dy * (1.f - y * y);
        case SG_ACT_SIGMOID: return dy * y * (1.f - y);
        default: return dy;
    }
}

// The same with the row sums of dx on the way out: row = one (sample, channel) image of S voxels, one workgroup per row.
// The bias gradient of a convolution is the column sum of these row sums — the separate pass over dx is saved.
__global__ void __launch_bounds__(256) act_bwd_rowsum_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                             float* __restrict__ dx, float* __restrict__ rowsum, long S,
                                                             int act, float slope) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * S;
    float s = 0.f;
    if ((S & 3) == 0) {
        const float4* y4 = reinterpret_cast<const float4*>(y + base);
        const float4* g4 = reinterpret_cast<const float4*>(dy + base);
        float4* d4 = reinterpret_cast<float4*>(dx + base);
        for (long e = threadIdx.x; e < (S >> 2); e += 256) {
            const float4 a = y4[e], g = g4[e];
            float4 v;
            v.x = act_grad_from_output(a.x, g.x, act, slope);
            v.y = act_grad_from_output(a.y, g.y, act, slope);
            v.z = act_grad_from_output(a.z, g.z, act, slope);
            v.w = act_grad_from_output(a.w, g.w, act, slope);
            d4[e] = v;
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (long e = threadIdx.x; e < S; e += 256) {
            const float v = act_grad_from_output(y[base + e], dy[base + e], act, slope);
            dx[base + e] = v;
            s += v;
        }
    }
    s = sg_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rowsum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act,
                                                      float slope) {
    const long n4 = n >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(x)[e];
        v.x = sg_apply_act(v.x, act, slope);
        v.y = sg_apply_act(v.y, act, slope);
        v.z = sg_apply_act(v.z, act, slope);
        v.w = sg_apply_act(v.w, act, slope);
        reinterpret_cast<float4*>(y)[e] = v;
    }
    for (long e = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        y[e] = sg_apply_act(x[e], act, slope);
}

// dx = dy * act'(.) with the derivative expressed through the OUTPUT y of the activation
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long n, int act, float slope) {
    const long n4 = n >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(y)[e];
        const float4 g = reinterpret_cast<const float4*>(dy)[e];
        float4 v;
        v.x = act_grad_from_output(a.x, g.x, act, slope);
        v.y = act_grad_from_output(a.y, g.y, act, slope);
        v.z = act_grad_from_output(a.z, g.z, act, slope);
        v.w = act_grad_from_output(a.w, g.w, act, slope);
        reinterpret_cast<float4*>(dx)[e] = v;
    }
    for (long e = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        dx[e] = act_grad_from_output(y[e], dy[e], act, slope);
}

// torch.optim.RMSprop defaults (momentum 0, centered False, weight_decay 0):
//   sq = alpha*sq + (1-alpha)*g*g ;  p -= lr * g / (sqrt(sq) + eps) ; optional clamp to [-clip, clip] (clip > 0)
__global__ void __launch_bounds__(256) rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ sq, long n, float lr, float alpha, float eps,
                                                      float gscale, float clip) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float gg = g[e] * gscale;
        const float s = alpha * sq[e] + (1.f - alpha) * gg * gg;
        sq[e] = s;
        float v = p[e] - lr * (gg / (sqrtf(s) + eps));
        if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
        p[e] = v;
    }
}

// torch.optim.Adam defaults (amsgrad False, weight_decay 0):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale, const int* __restrict__ skip) {
    // `skip` (optional): a device word an earlier kernel of the stream sets when this step's batch must not be applied
    // (sg_sdf_batch_sort's bad-index word): the update is then a no-op — parameters and both moments keep their values
    if (skip && *skip) return;
    // (every rounding spelled out, the same in the three Adam kernels: left to the compiler, `b2 v + (1 - b2) g g` was contracted
    //  into different fma forms in different kernels — one ulp apart, which a trajectory test over several steps sees)
    const float step = lr / bc1, c1 = 1.f - b1, c2 = 1.f - b2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float gg = __fmul_rn(g[e], gscale);
        const float mm = __fmaf_rn(b1, m[e], __fmul_rn(c1, gg));
        const float vv = __fmaf_rn(b2, v[e], __fmul_rn(__fmul_rn(c2, gg), gg));
        m[e] = mm;
        v[e] = vv;
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vv), bc2_sqrt), eps);
        p[e] = __fsub_rn(p[e], __fmul_rn(step, __fdiv_rn(mm, denom)));
    }
}

// Graph-capturable Adam: the step counter lives on the device.  ONE launch (round 6; it was a one-thread "prepare" launch + the
// update).  Thread 0 of a workgroup reads the counter and shares it; every thread derives the two bias corrections of step
// t = counter + 1 itself, in double as torch does on the host.  The counter is advanced by the workgroup that is the LAST TO HAVE
// READ it: thread 0 takes an arrival ticket as soon
// as its own read has returned — the increment is made to depend on the loaded value — and the one that draws the last ticket
// stores t, publishes the corrections in corr[0..1] (readable by the host) and clears the tickets, all under the other
// workgroups' element traffic.  Tickets on two levels: atomics on ONE word are served one after the other (~10 ns each) — the 900
// workgroups of a 460 k-parameter update on a single ticket word were 9 of that launch's 14.7 us (the host-counter kernel: 5.6).
// Workgroup b arrives at word b % 16 (128 bytes apart); the last arrival of a word arrives at the master word, the last of those
// advances the counter.
struct AdamDevSet {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
    float lr, b1, b2, eps, gscale;
    long long* step_dev;
    float* corr;
    unsigned nblocks;
};
__device__ __forceinline__ void adam_dev_body(const AdamDevSet& a, const unsigned bx, const unsigned nb) {
    // ONE read of the counter per workgroup, shared through LDS: the ticket below says "this workgroup has read the counter" — with a
    // read per wave, a wave that reads late could see the value the last-ticket workgroup has already stored (found by the bit-equality
    // test of the fused launch: a few elements a step ahead in their bias correction)
    __shared__ long long t_shared;
    if (threadIdx.x == 0) t_shared = *a.step_dev + 1;
    __syncthreads();
    const long long t = t_shared;
    // beta^t for the integer t by repeated squaring in double (a few ulp of double, i.e. the same float after rounding except at
    // ties; the library pow() is a routine of several microseconds)
    double p1 = 1.0, p2 = 1.0, q1 = (double)a.b1, q2 = (double)a.b2;
    for (long long e = t; e > 0; e >>= 1) {
        if (e & 1) {
            p1 *= q1;
            p2 *= q2;
        }
        q1 *= q1;
        q2 *= q2;
    }
    const float bc1 = (float)(1.0 - p1), bc2_sqrt = (float)sqrt(1.0 - p2);
    if (threadIdx.x == 0) {
        unsigned one = 1u;
        asm volatile("" : "+v"(one) : "v"((unsigned)t));      // the tickets are taken after the counter's value has arrived
        unsigned* master = reinterpret_cast<unsigned*>(a.corr + 2);
        const unsigned w = bx & 15u, nw = nb < 16u ? nb : 16u;
        unsigned* mine = reinterpret_cast<unsigned*>(a.corr + 32 + 32 * w);
        if (atomicAdd(mine, one) == (nb - w + 15u) / 16u - 1u) {
            *mine = 0u;
            if (atomicAdd(master, one) == nw - 1u) {
                *a.step_dev = t;
                a.corr[0] = bc1;
                a.corr[1] = bc2_sqrt;
                *master = 0u;
            }
        }
    }
    const float step = a.lr / bc1, b1 = a.b1, b2 = a.b2;
    float* __restrict__ p = a.p;
    float* __restrict__ m = a.m;
    float* __restrict__ v = a.v;
    const float* __restrict__ g = a.g;
    // (roundings spelled out as in adam_kernel)
    const float c1 = 1.f - b1, c2 = 1.f - b2;
    for (long e = (long)bx * 256 + threadIdx.x; e < a.n; e += (long)nb * 256) {
        const float gg = __fmul_rn(g[e], a.gscale);
        const float mm = __fmaf_rn(b1, m[e], __fmul_rn(c1, gg));
        const float vv = __fmaf_rn(b2, v[e], __fmul_rn(__fmul_rn(c2, gg), gg));
        m[e] = mm;
        v[e] = vv;
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vv), bc2_sqrt), a.eps);
        p[e] = __fsub_rn(p[e], __fmul_rn(step, __fdiv_rn(mm, denom)));
    }
}
__global__ void __launch_bounds__(256) adam_dev_kernel(AdamDevSet a, const int* __restrict__ skip) {
    if (skip && *skip) return;           // a skipped step neither moves nor ages anything
    adam_dev_body(a, blockIdx.x, gridDim.x);
}
// up to four flat buffers (the optimizers of one training step: train_sdf_autodecoder.py:44-45 has two) in ONE launch: grid y = buffer
struct AdamDevMulti {
    AdamDevSet s[4];
};
__global__ void __launch_bounds__(256) adam_dev_multi_kernel(AdamDevMulti a, const int* __restrict__ skip) {
    if (skip && *skip) return;
    const AdamDevSet& s = a.s[blockIdx.y];
    if (blockIdx.x >= s.nblocks) return;
    adam_dev_body(s, blockIdx.x, s.nblocks);
}

__global__ void __launch_bounds__(256) clamp_kernel(float* __restrict__ p, long n, float lo, float hi) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        p[e] = fminf(fmaxf(p[e], lo), hi);
}

// the same for up to 16 tensors in one launch (Discriminator.clip_weights, model/gan.py:67-69: eight parameter tensors per update):
// blockIdx.y = tensor
struct ClampMulti {
    float* p[16];
    long n[16];
};
__global__ void __launch_bounds__(256) clamp_multi_kernel(ClampMulti t, float lo, float hi) {
    float* __restrict__ p = t.p[blockIdx.y];
    const long n = t.n[blockIdx.y];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) p[e] = fminf(fmaxf(p[e], lo), hi);
}

// out = clamp(x, -c, c) [/ divisor]: VoxelDataset.__getitem__'s clamp_ and /= (datasets.py:19-22), NaN-propagating like
// torch.clamp, true IEEE division like torch's `/= 0.1` (a reciprocal multiply differs in the last bit)
__global__ void __launch_bounds__(256) voxel_prepare_kernel(const float* __restrict__ x, float* __restrict__ out, long n,
                                                            float c, float divisor) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        float v = x[e];
        v = v < -c ? -c : (v > c ? c : v);
        out[e] = divisor > 0.f ? v / divisor : v;
    }
}

// out = a*x + b*y   (y may be null)
__global__ void __launch_bounds__(256) axpby_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    float* __restrict__ out, long n, float a, float b) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        out[e] = y ? fmaf(a, x[e], b * y[e]) : a * x[e];
}

// out[i][:] = table[idx[i]][:]
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx,
                                                          float* __restrict__ out, long n, int L) {
    const long total = n * L;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e / L;
        const int k = (int)(e - i * L);
        out[e] = table[idx[i] * L + k];
    }
}
// table_grad[idx[i]][:] += rows[i][:]   (index_add; float atomics, device scope)
__global__ void __launch_bounds__(256) scatter_add_rows_kernel(const float* __restrict__ rows, long rows_ld,
                                                               const int64_t* __restrict__ idx,
                                                               float* __restrict__ table_grad, long n, int L) {
    const long total = n * L;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e / L;
        const int k = (int)(e - i * L);
        atomicAdd(&table_grad[idx[i] * L + k], rows[i * rows_ld + k]);
    }
}

// two-stage deterministic sum: partial[block] then final.  Every element is accumulated in double; four b128 loads are in flight
// per lane (the scalar-load form streamed 2.3 TB/s where a read-only pass of this box does 4.0, profiles/r03_stream_calibration.json)
__global__ void __launch_bounds__(256) sum_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, long n) {
    double s0 = 0, s1 = 0;
    const long stride = (long)gridDim.x * 256;
    long done = 0;
    if ((((uintptr_t)x) & 15) == 0) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        const long n4 = n >> 2;
        long e = (long)blockIdx.x * 256 + threadIdx.x;
        for (; e + 3 * stride < n4; e += 4 * stride) {
            const f32x4 v0 = __builtin_nontemporal_load(x4 + e), v1 = __builtin_nontemporal_load(x4 + e + stride);
            const f32x4 v2 = __builtin_nontemporal_load(x4 + e + 2 * stride), v3 = __builtin_nontemporal_load(x4 + e + 3 * stride);
            s0 += ((double)v0[0] + (double)v0[1]) + ((double)v0[2] + (double)v0[3]);
            s1 += ((double)v1[0] + (double)v1[1]) + ((double)v1[2] + (double)v1[3]);
            s0 += ((double)v2[0] + (double)v2[1]) + ((double)v2[2] + (double)v2[3]);
            s1 += ((double)v3[0] + (double)v3[1]) + ((double)v3[2] + (double)v3[3]);
        }
        for (; e < n4; e += stride) {
            const f32x4 v = x4[e];
            s0 += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
        }
        done = n4 << 2;
    }
    for (long e = done + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) s1 += (double)x[e];
    double s = sg_wave_sum_d(s0 + s1);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(64) sum_final_kernel(const double* __restrict__ partial, float* __restrict__ out, int nb,
                                                       float scale) {
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
    s = sg_wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)(s * (double)scale);
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_act_fwd(const float* x, float* y, long n, int act, float slope, hipStream_t stream) {
    SG_CHECK_ARG(x && y && n > 0);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n, 4)), dim3(256), 0, stream, x, y, n, act, slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_act_bwd(const float* y, const float* dy, float* dx, long n, int act, float slope, hipStream_t stream) {
    SG_CHECK_ARG(y && dy && dx && n > 0);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n, 4)), dim3(256), 0, stream, y, dy, dx, n, act, slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_act_bwd_rowsum(const float* y, const float* dy, float* dx, float* rowsum, long rows, long S, int act, float slope,
                      hipStream_t stream) {
    SG_CHECK_ARG(y && dy && dx && rowsum && rows > 0 && rows < (1L << 31) && S > 0);
    hipLaunchKernelGGL(act_bwd_rowsum_kernel, dim3((unsigned)rows), dim3(256), 0, stream, y, dy, dx, rowsum, S, act, slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_rmsprop_step(float* p, const float* g, float* square_avg, long n, float lr, float alpha, float eps,
                    float grad_scale, float clip, hipStream_t stream) {
    SG_CHECK_ARG(p && g && square_avg && n > 0);
    hipLaunchKernelGGL(rmsprop_kernel, dim3(ew_grid(n, 2)), dim3(256), 0, stream, p, g, square_avg, n, lr, alpha, eps,
                       grad_scale, clip);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                 float eps, long step, float grad_scale, hipStream_t stream) {
    return sg_adam_step_guarded(p, g, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale, nullptr, stream);
}
int sg_adam_step_guarded(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                         float eps, long step, float grad_scale, const int* skip_if_nonzero, hipStream_t stream) {
    SG_CHECK_ARG(p && g && exp_avg && exp_avg_sq && n > 0 && step > 0);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n, 2)), dim3(256), 0, stream, p, g, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale, skip_if_nonzero);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_adam_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                     float eps, long long* step_dev, float* corr_dev, float grad_scale, hipStream_t stream) {
    return sg_adam_step_dev_guarded(p, g, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step_dev, corr_dev, grad_scale, nullptr,
                                    stream);
}
int sg_adam_step_dev_guarded(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                             float beta2, float eps, long long* step_dev, float* corr_dev, float grad_scale,
                             const int* skip_if_nonzero, hipStream_t stream) {
    SG_CHECK_ARG(p && g && exp_avg && exp_avg_sq && n > 0 && step_dev && corr_dev);
    AdamDevSet a{p, g, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, step_dev, corr_dev, (unsigned)ew_grid(n, 2)};
    hipLaunchKernelGGL(adam_dev_kernel, dim3(a.nblocks), dim3(256), 0, stream, a, skip_if_nonzero);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
// The same for up to four flat buffers in one launch (every argument an array of nsets entries; one guard word for all).
int sg_adam_step_dev_multi(int nsets, float* const* p, const float* const* g, float* const* exp_avg, float* const* exp_avg_sq,
                           const long* n, const float* lr, const float* beta1, const float* beta2, const float* eps,
                           long long* const* step_dev, float* const* corr_dev, const float* grad_scale, const int* skip_if_nonzero,
                           hipStream_t stream) {
    SG_CHECK_ARG(nsets > 0 && nsets <= 4 && p && g && exp_avg && exp_avg_sq && n && lr && beta1 && beta2 && eps && step_dev && corr_dev &&
                 grad_scale);
    AdamDevMulti a;
    unsigned most = 0;
    for (int i = 0; i < 4; ++i) {
        const int k = i < nsets ? i : 0;
        SG_CHECK_ARG(p[k] && g[k] && exp_avg[k] && exp_avg_sq[k] && n[k] > 0 && step_dev[k] && corr_dev[k]);
        a.s[i] = AdamDevSet{p[k], g[k], exp_avg[k], exp_avg_sq[k], n[k], lr[k], beta1[k], beta2[k], eps[k], grad_scale[k], step_dev[k],
                            corr_dev[k], (unsigned)ew_grid(n[k], 2)};
        if (i < nsets && a.s[i].nblocks > most) most = a.s[i].nblocks;
    }
    hipLaunchKernelGGL(adam_dev_multi_kernel, dim3(most, nsets), dim3(256), 0, stream, a, skip_if_nonzero);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_clamp(float* p, long n, float lo, float hi, hipStream_t stream) {
    SG_CHECK_ARG(p && n > 0);
    hipLaunchKernelGGL(clamp_kernel, dim3(ew_grid(n, 2)), dim3(256), 0, stream, p, n, lo, hi);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_clamp_multi(float* const* tensors, const long* counts, int ntensors, float lo, float hi, hipStream_t stream) {
    SG_CHECK_ARG(tensors && counts && ntensors > 0);
    for (int i0 = 0; i0 < ntensors; i0 += 16) {
        ClampMulti t;
        const int m = ntensors - i0 < 16 ? ntensors - i0 : 16;
        long most = 0;
        for (int i = 0; i < 16; ++i) {
            const int k = i < m ? i0 + i : i0;
            SG_CHECK_ARG(tensors[k] != nullptr && counts[k] > 0);
            t.p[i] = tensors[k];
            t.n[i] = counts[k];
            if (i < m && counts[k] > most) most = counts[k];
        }
        hipLaunchKernelGGL(clamp_multi_kernel, dim3(ew_grid(most, 2), m), dim3(256), 0, stream, t, lo, hi);
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_voxel_prepare(const float* x, float* out, long n, float clamp, float divisor, hipStream_t stream) {
    SG_CHECK_ARG(x && out && n > 0 && clamp >= 0.f);
    hipLaunchKernelGGL(voxel_prepare_kernel, dim3(ew_grid(n, 2)), dim3(256), 0, stream, x, out, n, clamp, divisor);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_axpby(const float* x, const float* y, float* out, long n, float a, float b, hipStream_t stream) {
    SG_CHECK_ARG(x && out && n > 0);
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n, 2)), dim3(256), 0, stream, x, y, out, n, a, b);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_gather_rows(const float* table, const int64_t* idx, float* out, long n, int L, hipStream_t stream) {
    SG_CHECK_ARG(table && idx && out && n > 0 && L > 0);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(n * L, 2)), dim3(256), 0, stream, table, idx, out, n, L);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_scatter_add_rows(const float* rows, long rows_ld, const int64_t* idx, float* table_grad, long n, int L,
                        hipStream_t stream) {
    SG_CHECK_ARG(rows && idx && table_grad && n > 0 && L > 0 && rows_ld >= L);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(ew_grid(n * L, 2)), dim3(256), 0, stream, rows, rows_ld, idx,
                       table_grad, n, L);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
size_t sg_reduce_workspace_bytes(void) { return 1024 * sizeof(double); }
int sg_reduce_sum(const float* x, float* out, long n, float scale, void* workspace, size_t workspace_bytes,
                  hipStream_t stream) {
    SG_CHECK_ARG(x && out && n > 0);
    if (!workspace || workspace_bytes < sg_reduce_workspace_bytes()) SG_FAIL(SG_ERR_WORKSPACE, "sg_reduce_sum: workspace too small");
    int nb = ew_grid(n, 8);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(256), 0, stream, x, (double*)workspace, n);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, out, nb, scale);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

static thread_local char g_err[512];
const char* sg_last_error(void) { return g_err; }
int sg_abi_version(void) { return SG_ABI_VERSION; }

}  // extern "C"

char* sg_err_buf() { return g_err; }
