// shapegan_amd/csrc/batchnorm.hip — BatchNorm3d / BatchNorm1d (K4) with the following LeakyReLU folded in.
//
// Replaces ATen batch_norm / batch_norm_backward behind nn.BatchNorm3d (model/gan.py:10,14,18;
// model/autoencoder.py:17,21,25,29,56,60,64) and nn.BatchNorm1d (model/autoencoder.py:38,46), with torch's
// conventions: biased variance for normalisation, unbiased for running_var, momentum 0.1, eps inside the sqrt.
//
// x is [N, C, S] (S = D*H*W, 1 for BatchNorm1d).  Statistics are HBM-bound reductions: each (channel, split)
// workgroup streams its slice with float4 loads, reduces per wave with shuffles (DPP), then across the 4 waves
// through LDS; the partials are combined in double at the head of the second pass (finalize + apply in one launch), whose
// (channel, 0) workgroup also updates the running statistics.
// Sums are shifted by the channel's first element so E[x^2]-E[x]^2 does not cancel when |mean| >> std.
#include "common.h"
#include "../../include/shapegan_hip.h"

namespace sg {

// iterate the elements of channel c: for n in [0,N): x[(n*C + c)*S + s], s in [0,S)
// flat index e in [0, N*S): n = e / S, s = e % S
__device__ __forceinline__ long chan_addr(long e, int c, int C, long S) {
    const long n = e / S, s = e - n * S;
    return (n * C + c) * S + s;
}

// the same walk without a 64-bit division per element: a float4 group at flat index e (e % 4 == 0, S % 4 == 0 so a group never
// straddles two samples) and steps of `step` elements
struct ChanWalk {
    long n, s, S;
    int c, C;
    __device__ ChanWalk(long e, int c_, int C_, long S_) : S(S_), c(c_), C(C_) {
        n = e / S_;
        s = e - n * S_;
    }
    __device__ __forceinline__ long addr() const { return (n * C + c) * S + s; }
    __device__ __forceinline__ void advance(long step) {
        s += step;
        if (s >= S) {
            if (step <= S) {
                s -= S;
                ++n;
            } else {
                const long k = s / S;
                n += k;
                s -= k * S;
            }
        }
    }
};

// partial[c][split] = {sum(x-K), sum((x-K)^2)}
// (grid z = group: the tensor is [groups][N][C][S] and every group of N samples has statistics of its own — several generator
// evaluations batched into one pass, shapegan_amd.train_steps.WGANTrainer.step; partial[group][c][split])
__global__ void __launch_bounds__(256) bn_stats_kernel(const float* __restrict__ x, double* __restrict__ partial, int N,
                                                       int C, long S, int nsplit) {
    const int c = blockIdx.x, sp = blockIdx.y;
    x += (long)blockIdx.z * N * C * S;
    partial += (long)blockIdx.z * C * nsplit * 2;
    const long total = (long)N * S;
    const long chunk = ((total + nsplit - 1) / nsplit + 3) & ~3L;
    const long beg = sp * chunk, end = min(total, beg + chunk);
    const float K = x[(long)c * S];
    float s1 = 0.f, s2 = 0.f;
    if ((S & 3) == 0 && (beg & 3) == 0) {
        for (long e = beg + (long)threadIdx.x * 4; e < end; e += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(x + chan_addr(e, c, C, S));
            const float a = v.x - K, b = v.y - K, cc = v.z - K, d = v.w - K;
            s1 += (a + b) + (cc + d);
            s2 += (a * a + b * b) + (cc * cc + d * d);
        }
    } else {
        for (long e = beg + threadIdx.x; e < end; e += 256) {
            const float a = x[chan_addr(e, c, C, S)] - K;
            s1 += a;
            s2 += a * a;
        }
    }
    __shared__ double red[2][4];
    double d1 = sg_wave_sum_d((double)s1), d2 = sg_wave_sum_d((double)s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = d1;
        red[1][threadIdx.x >> 6] = d2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((long)c * nsplit + sp) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[((long)c * nsplit + sp) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ void __launch_bounds__(256) bn_eval_stats_kernel(const float* __restrict__ running_mean,
                                                            const float* __restrict__ running_var,
                                                            float* __restrict__ mean, float* __restrict__ invstd, int C,
                                                            float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    invstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

// y = act(gamma * (x - mean) * invstd + beta)
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int C, long S, long total, int act, float slope) {
    if ((S & 3) == 0) {
        for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += (long)gridDim.x * 1024) {
            const int c = (int)((e / S) % C);
            const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
            float4 v = *reinterpret_cast<const float4*>(x + e);
            v.x = sg_apply_act(fmaf(v.x, sc, sh), act, slope);
            v.y = sg_apply_act(fmaf(v.y, sc, sh), act, slope);
            v.z = sg_apply_act(fmaf(v.z, sc, sh), act, slope);
            v.w = sg_apply_act(fmaf(v.w, sc, sh), act, slope);
            *reinterpret_cast<float4*>(y + e) = v;
        }
    } else {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int c = (int)((e / S) % C);
            const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
            y[e] = sg_apply_act(fmaf(x[e], sc, sh), act, slope);
        }
    }
}

// Training forward, second pass: FINALIZE + APPLY in one launch.  Grid (C, nsplit) like the statistics pass: every workgroup of
// channel c first combines the channel's `nparts` double partials (threads < nparts load one pair each, wave butterflies in
// double: a fixed order, so all workgroups of a channel — and every run — get bit-identical statistics), then streams its slice
// of the channel.  Workgroup (c, 0) publishes mean / invstd and updates the running statistics; (0, 0) bumps the batch counter.
// (One launch less per BatchNorm and no per-element `e / S % C`.)
// the statistics of (group, channel) from its partials: one wave, fixed order (bit-identical in every workgroup and run)
__device__ __forceinline__ void bn_group_stats(const float* __restrict__ x, const double* __restrict__ partial, int g, int c, int N,
                                               int C, long S, int nparts, float eps, float& mu, float& is, double& var_out) {
    const int lane = threadIdx.x & 63;
    const double* p = partial + ((long)g * C + c) * nparts * 2;
    double s1 = lane < nparts ? p[lane * 2] : 0.0;
    double s2 = lane < nparts ? p[lane * 2 + 1] : 0.0;
    s1 = sg_wave_sum_d(s1);
    s2 = sg_wave_sum_d(s2);
    const double n = (double)N * (double)S;
    const double K = (double)x[((long)g * N * C + c) * S];
    const double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0) var = 0;
    mu = (float)(K + m);
    is = (float)(1.0 / sqrt(var + (double)eps));
    var_out = var;
}
// running statistics after `groups` batches, applied one after the other exactly as `groups` separate calls would (same fp32
// operations in the same order); one wave of workgroup (c, 0, 0)
__device__ __forceinline__ void bn_update_running(const float* __restrict__ x, const double* __restrict__ partial, int c, int N, int C,
                                                  long S, int nparts, int groups, float eps, float momentum, float* running_mean,
                                                  float* running_var, long long* num_batches_tracked) {
    const double n = (double)N * (double)S;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
    for (int g = 0; g < groups; ++g) {
        float mu, is;
        double var;
        bn_group_stats(x, partial, g, c, N, C, S, nparts, eps, mu, is, var);
        const double unbiased = n > 1 ? var * n / (n - 1) : var;
        rm = (1.f - momentum) * rm + momentum * mu;
        rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
    if ((threadIdx.x & 63) == 0) {
        if (running_mean) {
            running_mean[c] = rm;
            running_var[c] = rv;
        }
        if (c == 0 && num_batches_tracked) *num_batches_tracked += groups;
    }
}

__global__ void __launch_bounds__(256) bn_finalize_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                const double* __restrict__ partial,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ mean, float* __restrict__ invstd,
                                                                float* running_mean, float* running_var,
                                                                long long* num_batches_tracked, int N, int C, long S, int nparts,
                                                                int nsplit, float eps, float momentum, int act, float slope) {
    const int c = blockIdx.x, sp = blockIdx.y, g = blockIdx.z, groups = gridDim.z;
    __shared__ float stat[2];
    if (threadIdx.x < 64) {      // nparts <= 64: one wave
        float mu, is;
        double var;
        bn_group_stats(x, partial, g, c, N, C, S, nparts, eps, mu, is, var);
        if (threadIdx.x == 0) {
            stat[0] = mu;
            stat[1] = is;
            if (sp == 0) {
                mean[(long)g * C + c] = mu;
                invstd[(long)g * C + c] = is;
            }
        }
        if (sp == 0 && g == 0)
            bn_update_running(x, partial, c, N, C, S, nparts, groups, eps, momentum, running_mean, running_var, num_batches_tracked);
    }
    __syncthreads();
    x += (long)g * N * C * S;
    y += (long)g * N * C * S;
    const float sc = gamma[c] * stat[1], sh = beta[c] - stat[0] * sc;
    const long total = (long)N * S;
    const long chunk = ((total + nsplit - 1) / nsplit + 3) & ~3L;
    const long beg = sp * chunk, end = min(total, beg + chunk);
    if ((S & 3) == 0) {
        for (long e = beg + (long)threadIdx.x * 4; e < end; e += 1024) {
            const long ad = chan_addr(e, c, C, S);
            float4 v = *reinterpret_cast<const float4*>(x + ad);
            v.x = sg_apply_act(fmaf(v.x, sc, sh), act, slope);
            v.y = sg_apply_act(fmaf(v.y, sc, sh), act, slope);
            v.z = sg_apply_act(fmaf(v.z, sc, sh), act, slope);
            v.w = sg_apply_act(fmaf(v.w, sc, sh), act, slope);
            *reinterpret_cast<float4*>(y + ad) = v;
        }
    } else {
        for (long e = beg + threadIdx.x; e < end; e += 256) {
            const long ad = chan_addr(e, c, C, S);
            y[ad] = sg_apply_act(fmaf(x[ad], sc, sh), act, slope);
        }
    }
}

// Statistics WITHOUT the apply pass: finalize of bn_stats_kernel's partials into mean / invstd, the running statistics, and the
// affine map  scale[c] = gamma[c] * invstd[c],  shift[c] = beta[c] - mean[c] * scale[c]  that a CONSUMER kernel applies on its loads
// (sg_convT3d_k4s2p1_to1_pre): the normalised tensor is never written.  One wave per channel, same summation order as
// bn_finalize_apply_kernel (bit-identical statistics).
__global__ void __launch_bounds__(256) bn_finalize_affine_kernel(const float* __restrict__ x, const double* __restrict__ partial,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ mean, float* __restrict__ invstd,
                                                                 float* running_mean, float* running_var,
                                                                 long long* num_batches_tracked, float* __restrict__ scale,
                                                                 float* __restrict__ shift, int N, int C, long S, int nparts,
                                                                 float eps, float momentum) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, g = blockIdx.y, groups = gridDim.y;
    if (c >= C) return;
    float mu, is;
    double var;
    bn_group_stats(x, partial, g, c, N, C, S, nparts, eps, mu, is, var);
    if (lane == 0) {
        mean[(long)g * C + c] = mu;
        invstd[(long)g * C + c] = is;
        const float sc = gamma[c] * is;
        scale[(long)g * C + c] = sc;
        shift[(long)g * C + c] = beta[c] - mu * sc;
    }
    if (g == 0) bn_update_running(x, partial, c, N, C, S, nparts, groups, eps, momentum, running_mean, running_var, num_batches_tracked);
}

__device__ __forceinline__ float bn_act_grad(float xhat, float g, float b, float dy, int act, float slope) {
    if (act == SG_ACT_LEAKY) return (fmaf(g, xhat, b) > 0.f) ? dy : dy * slope;
    if (act == SG_ACT_RELU) return (fmaf(g, xhat, b) > 0.f) ? dy : 0.f;
    if (act == SG_ACT_TANH) {   // derivative through the recomputed output, as the forward applied it (sg_apply_act)
        const float y = tanhf(fmaf(g, xhat, b));
        return dy * (1.f - y * y);
    }
    if (act == SG_ACT_SIGMOID) {
        const float y = 1.f / (1.f + expf(-fmaf(g, xhat, b)));
        return dy * y * (1.f - y);
    }
    return dy;
}

// partial[c][split] = {sum(g), sum(g * xhat)},  g = dy * act'(bn(x))
__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, double* __restrict__ partial,
                                                           int N, int C, long S, int nsplit, int act, float slope) {
    const int c = blockIdx.x, sp = blockIdx.y;
    const long total = (long)N * S;
    const long chunk = ((total + nsplit - 1) / nsplit + 3) & ~3L;
    const long beg = sp * chunk, end = min(total, beg + chunk);
    const float mu = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
    float s1 = 0.f, s2 = 0.f;
    if ((S & 3) == 0 && (beg & 3) == 0) {
        // b128 loads of both operands, two groups in flight per lane (the scalar form with a 64-bit division per element read
        // 134 MB in 54 us at 64 x 64 x 16^3)
        ChanWalk w(beg + (long)threadIdx.x * 4, c, C, S);
        long e = beg + (long)threadIdx.x * 4;
        for (; e + 1024 < end; e += 2048) {
            const long a0 = w.addr();
            w.advance(1024);
            const long a1 = w.addr();
            w.advance(1024);
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + a0), d0 = *reinterpret_cast<const f32x4*>(dy + a0);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + a1), d1 = *reinterpret_cast<const f32x4*>(dy + a1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh0 = (x0[j] - mu) * is, xh1 = (x1[j] - mu) * is;
                const float g0 = bn_act_grad(xh0, g, b, d0[j], act, slope), g1 = bn_act_grad(xh1, g, b, d1[j], act, slope);
                s1 += g0 + g1;
                s2 += g0 * xh0 + g1 * xh1;
            }
        }
        if (e < end) {
            const long a0 = w.addr();
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + a0), d0 = *reinterpret_cast<const f32x4*>(dy + a0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh0 = (x0[j] - mu) * is;
                const float g0 = bn_act_grad(xh0, g, b, d0[j], act, slope);
                s1 += g0;
                s2 += g0 * xh0;
            }
        }
    } else {
        for (long e = beg + threadIdx.x; e < end; e += 256) {
            const long ad = chan_addr(e, c, C, S);
            const float xh = (x[ad] - mu) * is;
            const float gg = bn_act_grad(xh, g, b, dy[ad], act, slope);
            s1 += gg;
            s2 += gg * xh;
        }
    }
    __shared__ double red[2][4];
    double d1 = sg_wave_sum_d((double)s1), d2 = sg_wave_sum_d((double)s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = d1;
        red[1][threadIdx.x >> 6] = d2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((long)c * nsplit + sp) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[((long)c * nsplit + sp) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// Backward, second pass: FINALIZE + APPLY (same scheme as bn_finalize_apply_kernel): every workgroup of channel c combines the
// channel's partial sums in a fixed order, workgroup (c, 0) writes dgamma / dbeta.
__global__ void __launch_bounds__(256) bn_bwd_finalize_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    float* __restrict__ dx, const double* __restrict__ partial,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                                    int C, long S, int nparts, int nsplit, int train, int act,
                                                                    float slope) {
    const int c = blockIdx.x, sp = blockIdx.y;
    __shared__ float sums[2];
    if (threadIdx.x < 64) {
        double s1 = threadIdx.x < nparts ? partial[((long)c * nparts + threadIdx.x) * 2] : 0.0;
        double s2 = threadIdx.x < nparts ? partial[((long)c * nparts + threadIdx.x) * 2 + 1] : 0.0;
        s1 = sg_wave_sum_d(s1);
        s2 = sg_wave_sum_d(s2);
        if (threadIdx.x == 0) {
            sums[0] = (float)s1;
            sums[1] = (float)s2;
            if (sp == 0) {
                dbeta[c] = (float)s1;
                dgamma[c] = (float)s2;
            }
        }
    }
    __syncthreads();
    const float inv_n = 1.f / ((float)N * (float)S);
    const float is = invstd[c], g = gamma[c], b = beta[c], mu = mean[c];
    const float db = sums[0] * inv_n, dg = sums[1] * inv_n;
    const long total = (long)N * S;
    const long chunk = ((total + nsplit - 1) / nsplit + 3) & ~3L;
    const long beg = sp * chunk, end = min(total, beg + chunk);
    if ((S & 3) == 0 && (beg & 3) == 0) {
        ChanWalk w(beg + (long)threadIdx.x * 4, c, C, S);
        for (long e = beg + (long)threadIdx.x * 4; e < end; e += 1024) {
            const long ad = w.addr();
            w.advance(1024);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ad), dv = *reinterpret_cast<const f32x4*>(dy + ad);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[j] - mu) * is;
                const float gg = bn_act_grad(xh, g, b, dv[j], act, slope);
                float v = gg;
                if (train) v = gg - db - xh * dg;
                o[j] = g * is * v;
            }
            *reinterpret_cast<f32x4*>(dx + ad) = o;
        }
        return;
    }
    for (long e = beg + threadIdx.x; e < end; e += 256) {
        const long ad = chan_addr(e, c, C, S);
        const float xh = (x[ad] - mu) * is;
        const float gg = bn_act_grad(xh, g, b, dy[ad], act, slope);
        float v = gg;
        if (train) v = gg - db - xh * dg;
        dx[ad] = g * is * v;
    }
}

static int bn_nsplit(int N, int C, long S) {
    const long per = (long)N * S;
    long want = 1024 / (C > 0 ? C : 1);  // ~4 workgroups per CU overall
    if (want < 1) want = 1;
    long maxs = per / 4096;              // each split at least 4096 elements
    if (maxs < 1) maxs = 1;
    long s = want < maxs ? want : maxs;
    if (s > 64) s = 64;
    return (int)s;
}
// slices per channel of the apply passes: enough workgroups to fill the chip (~8 per CU), each at least 4096 elements
static int bn_apply_split(int N, int C, long S) {
    const long per = (long)N * S;
    long want = (2048 + C - 1) / (C > 0 ? C : 1);
    long maxs = per / 4096;
    if (maxs < 1) maxs = 1;
    long s = want < maxs ? want : maxs;
    if (s > 1024) s = 1024;
    if (s < 1) s = 1;
    return (int)s;
}
static int ew_blocks(long total, int per_thread) {
    long b = (total + 256L * per_thread - 1) / (256L * per_thread);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_bn_workspace_bytes(int C) { return (size_t)C * 64 * 2 * sizeof(double); }

int sg_bn_train_fwd_grouped(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                            float* running_mean, float* running_var, long long* num_batches_tracked, int groups, int N, int C,
                            long S, float eps, float momentum, int act, float slope, void* workspace, size_t workspace_bytes,
                            hipStream_t stream) {
    SG_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && groups > 0 && groups <= 64 && N > 0 && C > 0 && S > 0);
    if (!workspace || workspace_bytes < (size_t)groups * sg_bn_workspace_bytes(C))
        SG_FAIL(SG_ERR_WORKSPACE, "sg_bn_train_fwd: workspace too small");
    const int ns = bn_nsplit(N, C, S);
    double* part = (double*)workspace;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns, groups), dim3(256), 0, stream, x, part, N, C, S, ns);
    // finalize + apply in one launch; the apply pass may use more slices per channel than the statistics pass
    const int na = bn_apply_split(N, C, S);
    hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(C, na, groups), dim3(256), 0, stream, x, y, (const double*)part, gamma, beta,
                       save_mean, save_invstd, running_mean, running_var, num_batches_tracked, N, C, S, ns, na, eps, momentum, act,
                       slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                    float* running_mean, float* running_var, long long* num_batches_tracked, int N, int C, long S,
                    float eps, float momentum, int act, float slope, void* workspace, size_t workspace_bytes,
                    hipStream_t stream) {
    return sg_bn_train_fwd_grouped(x, gamma, beta, y, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, 1, N,
                                   C, S, eps, momentum, act, slope, workspace, workspace_bytes, stream);
}

int sg_bn_train_stats_grouped(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                              float* running_mean, float* running_var, long long* num_batches_tracked, float* scale, float* shift,
                              int groups, int N, int C, long S, float eps, float momentum, void* workspace, size_t workspace_bytes,
                              hipStream_t stream) {
    SG_CHECK_ARG(x && gamma && beta && save_mean && save_invstd && scale && shift && groups > 0 && groups <= 64 && N > 0 && C > 0 &&
                 S > 0);
    if (!workspace || workspace_bytes < (size_t)groups * sg_bn_workspace_bytes(C))
        SG_FAIL(SG_ERR_WORKSPACE, "sg_bn_train_stats: workspace too small");
    const int ns = bn_nsplit(N, C, S);
    double* part = (double*)workspace;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, ns, groups), dim3(256), 0, stream, x, part, N, C, S, ns);
    hipLaunchKernelGGL(bn_finalize_affine_kernel, dim3((C + 3) / 4, groups), dim3(256), 0, stream, x, (const double*)part, gamma,
                       beta, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, scale, shift, N, C, S, ns, eps,
                       momentum);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_bn_train_stats(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                      float* running_mean, float* running_var, long long* num_batches_tracked, float* scale, float* shift, int N,
                      int C, long S, float eps, float momentum, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return sg_bn_train_stats_grouped(x, gamma, beta, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, scale,
                                     shift, 1, N, C, S, eps, momentum, workspace, workspace_bytes, stream);
}

int sg_bn_eval_fwd(const float* x, const float* gamma, const float* beta, float* y, const float* running_mean,
                   const float* running_var, float* save_mean, float* save_invstd, int N, int C, long S, float eps,
                   int act, float slope, hipStream_t stream) {
    SG_CHECK_ARG(x && gamma && beta && y && running_mean && running_var && save_mean && save_invstd && N > 0 && C > 0);
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(sg_cdiv(C, 256)), dim3(256), 0, stream, running_mean, running_var,
                       save_mean, save_invstd, C, eps);
    const long total = (long)N * C * S;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(total, 4)), dim3(256), 0, stream, x, y, gamma, beta,
                       (const float*)save_mean, (const float*)save_invstd, C, S, total, act, slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_bn_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
              const float* save_invstd, float* dx, float* dgamma, float* dbeta, int N, int C, long S, int train, int act,
              float slope, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(dy && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && N > 0 && C > 0 && S > 0);
    if (!workspace || workspace_bytes < sg_bn_workspace_bytes(C)) SG_FAIL(SG_ERR_WORKSPACE, "sg_bn_bwd: workspace too small");
    const int ns = bn_nsplit(N, C, S);
    double* part = (double*)workspace;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(C, ns), dim3(256), 0, stream, dy, x, gamma, beta, save_mean, save_invstd,
                       part, N, C, S, ns, act, slope);
    const int na = bn_apply_split(N, C, S);
    hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel, dim3(C, na), dim3(256), 0, stream, dy, x, dx, (const double*)part, gamma,
                       beta, save_mean, save_invstd, dgamma, dbeta, N, C, S, ns, na, train, act, slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"
