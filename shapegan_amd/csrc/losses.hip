// shapegan_amd/csrc/losses.hip — the loss compositions and blends of the training scripts as single-pass HBM-bound kernels
// (§8 row a13).
//
// Reference sites:
//   get_reconstruction_loss   train_autoencoder.py:57-62   mean |d|, d = out - target, d *= 32 where target < 0
//   kld_loss                  train_autoencoder.py:54-55   -0.5 * sum(1 + lv - mu^2 - exp(lv)) / numel
//   DeepSDF loss              train_sdf_autodecoder.py:88  mean |out - sdf|  +  sigma * mean(z_batch^2)
//   gradient penalty          train_hybrid_progressive_gan.py:103-111, train_point_gan.py:61-70
//                             lerp alpha*real + (1-alpha)*fake;  ((||g_b||_2 - 1)^2).mean() * lambda
//   fade-in blend             model/progressive_gan.py:48-50  fade*x + (1-fade)*from_SDF(x_in[:, ::2, ::2, ::2])
//   scatter_max               model/point_sdf_net.py:42-43 (torch_scatter, ragged `batch` vector)
// Reductions are two-stage and deterministic (per-workgroup double partials, one finishing wave); every backward is an
// elementwise pass that reads the upstream scalar gradient from device memory (no host round trip).
#include "common.h"
#include "../../include/shapegan_hip.h"

// the blends below promise the reference's rounding sequence (separate products and sums): no fma contraction in this file
#pragma clang fp contract(off)

namespace sg {

constexpr int kRedBlocks = 512;

static int red_grid(long n) {
    long b = (n + 2047) / 2048;
    if (b > kRedBlocks) b = kRedBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ void block_store_partial(double s, double* __restrict__ partial) {
    __shared__ double red[4];
    s = sg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(64) loss_final_kernel(const double* __restrict__ partial, float* __restrict__ out, int nb,
                                                        double scale) {
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
    s = sg_wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

// ---- difference of two batch means (WGAN losses) ---------------------------------------------------------------------
// loss = wa mean(x[0, na)) + wb mean(x[na, n)) in ONE single-workgroup launch (critic outputs: n = 2 x batch values), and its
// backward dx[i] = g * (i < na ? wa / na : wb / (n - na)).  Replaces the two means, the subtraction and the slice / expand /
// zero-fill / add nodes autograd would build for `mean(out[:na]) - mean(out[na:])`: 14 launches -> 2.
__global__ void __launch_bounds__(256) mean_split_fwd_kernel(const float* __restrict__ x, long n, long na, float wa, float wb,
                                                             float* __restrict__ loss, float* __restrict__ dx_unit, float ca,
                                                             float cb) {
    __shared__ double ra[4], rb[4];
    double sa = 0, sb = 0;
    for (long e = threadIdx.x; e < n; e += 256) {
        const double v = (double)x[e];
        if (e < na)
            sa += v;
        else
            sb += v;
        if (dx_unit) dx_unit[e] = e < na ? ca : cb;     // the backward for an upstream gradient of exactly 1
    }
    sa = sg_wave_sum_d(sa);
    sb = sg_wave_sum_d(sb);
    if ((threadIdx.x & 63) == 0) {
        ra[threadIdx.x >> 6] = sa;
        rb[threadIdx.x >> 6] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ta = (ra[0] + ra[1]) + (ra[2] + ra[3]), tb = (rb[0] + rb[1]) + (rb[2] + rb[3]);
        double v = 0;
        if (na > 0) v += (double)wa * ta / (double)na;
        if (n > na) v += (double)wb * tb / (double)(n - na);
        loss[0] = (float)v;
    }
}
__global__ void __launch_bounds__(256) mean_split_bwd_kernel(const float* __restrict__ gloss, float* __restrict__ dx, long n,
                                                             long na, float ca, float cb) {
    const float g = gloss[0];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) dx[e] = g * (e < na ? ca : cb);
}

// ---- classic-GAN losses on the [B] vector of discriminator outputs (train_gan.py:30,65,78,84) and the VAE reparameterisation
// (model/autoencoder.py:77-82): single-workgroup launches, double accumulation in a fixed order ------------------------------
// binary_cross_entropy(p, full_like(p, t)) = mean_i -( t max(log p_i, -100) + (1 - t) max(log(1 - p_i), -100) )  (torch clamps
// the logarithms at -100); MODE 1: -mean(log p) without the clamp (train_gan.py:65)
template <int MODE>
__global__ void __launch_bounds__(256) bce_fwd_kernel(const float* __restrict__ p, long n, float t, float* __restrict__ loss) {
    __shared__ double red[4];
    double s = 0;
    for (long e = threadIdx.x; e < n; e += 256) {
        const float v = p[e];
        if (MODE == 1)
            s -= (double)logf(v);
        else
            s -= (double)(t * fmaxf(logf(v), -100.f) + (1.f - t) * fmaxf(logf(1.f - v), -100.f));
    }
    s = sg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
}
// torch's backward: g (p - t) / max((1 - p) p, 1e-12) / n;  MODE 1: -g / (n p)
template <int MODE>
__global__ void __launch_bounds__(256) bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ gloss,
                                                      float* __restrict__ dp, long n, float t, float inv_n) {
    const float g = gloss[0] * inv_n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float v = p[e];
        dp[e] = MODE == 1 ? -g / v : g * (v - t) / fmaxf((1.f - v) * v, 1e-12f);
    }
}
// z = mean + exp(0.5 logvar) eps;   d logvar = gz eps 0.5 exp(0.5 logvar)   (d mean = gz, d eps is never needed)
__global__ void __launch_bounds__(256) reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                          const float* __restrict__ eps, float* __restrict__ z, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float sd = expf(lv[e] * 0.5f);
        z[e] = __fadd_rn(mu[e], __fmul_rn(sd, eps[e]));      // two roundings, as the reference's `mean + standard_deviation * eps`
    }
}
__global__ void __launch_bounds__(256) reparam_bwd_kernel(const float* __restrict__ lv, const float* __restrict__ eps,
                                                          const float* __restrict__ gz, float* __restrict__ dlv, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        dlv[e] = gz[e] * eps[e] * (0.5f * expf(lv[e] * 0.5f));
}

// ---- weighted L1 ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wl1_fwd_kernel(const float* __restrict__ o, const float* __restrict__ t, long n,
                                                      float negw, double* __restrict__ partial) {
    double s = 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float tt = t[e];
        float d = o[e] - tt;
        if (tt < 0.f) d *= negw;
        s += (double)fabsf(d);
    }
    block_store_partial(s, partial);
}
__global__ void __launch_bounds__(256) wl1_bwd_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                      const float* __restrict__ gloss, float* __restrict__ d_o, long n,
                                                      float negw, float inv_n) {
    const float g = gloss[0] * inv_n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float tt = t[e];
        const float w = tt < 0.f ? negw : 1.f;
        const float d = (o[e] - tt) * w;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        d_o[e] = g * sgn * w;
    }
}

// ---- KL divergence of N(mu, exp(lv)) from N(0, 1) -----------------------------------------------------------------
__global__ void __launch_bounds__(256) kld_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, long n,
                                                      double* __restrict__ partial) {
    double s = 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float m = mu[e], l = lv[e];
        s += (double)(1.f + l - m * m - expf(l));
    }
    block_store_partial(s, partial);
}
__global__ void __launch_bounds__(256) kld_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                      const float* __restrict__ gloss, float* __restrict__ dmu,
                                                      float* __restrict__ dlv, long n, float inv_n) {
    const float g = gloss[0] * inv_n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        dmu[e] = g * mu[e];
        dlv[e] = -0.5f * g * (1.f - expf(lv[e]));
    }
}

// ---- sign-mismatch count (voxel_difference, train_autoencoder.py:50-52) ----------------------------------------------
// count of elements with (a * b) < 0 where a * b is the ROUNDED fp32 product, exactly as `(input * target) < 0` evaluates it:
// a product that underflows to -0 does not count, a NaN operand does not count.  Integer partials, integer total: bit-exact and
// independent of the launch geometry.
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long other = ((unsigned long long)__shfl_xor(hi, off) << 32) | __shfl_xor(lo, off);
        const unsigned long long sum = (((unsigned long long)hi << 32) | lo) + other;
        lo = (unsigned)sum;
        hi = (unsigned)(sum >> 32);
    }
    return ((unsigned long long)hi << 32) | lo;
}
__global__ void __launch_bounds__(256) sign_mismatch_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                            unsigned long long* __restrict__ partial) {
    __shared__ unsigned long long red[4];
    unsigned long long c = 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float prod = a[e] * b[e];
        c += prod < 0.f ? 1ull : 0ull;
    }
    c = wave_sum_u64(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(64) sign_mismatch_final_kernel(const unsigned long long* __restrict__ partial,
                                                                 long long* __restrict__ count, int nb) {
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
    s = wave_sum_u64(s);
    if (threadIdx.x == 0) count[0] = (long long)s;
}

// ---- (row-weighted) mean of squares ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sq_fwd_kernel(const float* __restrict__ x, const float* __restrict__ roww, long n,
                                                     int L, double* __restrict__ partial) {
    double s = 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float v = x[e];
        s += (double)(roww ? roww[e / L] * v * v : v * v);
    }
    block_store_partial(s, partial);
}
__global__ void __launch_bounds__(256) sq_bwd_kernel(const float* __restrict__ x, const float* __restrict__ roww,
                                                     const float* __restrict__ gloss, float* __restrict__ dx, long n, int L,
                                                     float scale2) {
    const float g = gloss[0] * scale2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
        dx[e] = (roww ? roww[e / L] : 1.f) * g * x[e];
}

// ---- the whole DeepSDF loss (train_sdf_autodecoder.py:88) in one pass ---------------------------------------------------
// loss = mean|out - sdf| + sum_r w_r |z_r|^2 / denom: workgroups [0, nb1) take the data term, the rest the regulariser; the
// finishing wave rounds each term to fp32 and adds them in fp32, and the backward is one launch over both operands — the same
// arithmetic, in the same order, as sg_loss_weighted_l1 (neg_weight 1) + sg_loss_meansq + the fp32 add (7 launches -> 3).
__global__ void __launch_bounds__(256) deepsdf_fwd_kernel(const float* __restrict__ o, const float* __restrict__ t, long n,
                                                          int nb1, const float* __restrict__ x, const float* __restrict__ roww,
                                                          long m, int L, double* __restrict__ partial) {
    __shared__ double red[4];
    double s = 0;
    const int b = blockIdx.x;
    if (b < nb1) {
        for (long e = (long)b * 256 + threadIdx.x; e < n; e += (long)nb1 * 256) s += (double)fabsf(o[e] - t[e]);
    } else {
        const long nb2 = gridDim.x - nb1;
        for (long e = (long)(b - nb1) * 256 + threadIdx.x; e < m; e += nb2 * 256) {
            const float v = x[e];
            s += (double)(roww ? roww[e / L] * v * v : v * v);
        }
    }
    s = sg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[b < nb1 ? b : kRedBlocks + (b - nb1)] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(64) deepsdf_final_kernel(const double* __restrict__ partial, float* __restrict__ out, int nb1,
                                                           int nb2, double scale1, double scale2) {
    double s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < nb1; i += 64) s1 += partial[i];
    for (int i = threadIdx.x; i < nb2; i += 64) s2 += partial[kRedBlocks + i];
    s1 = sg_wave_sum_d(s1);
    s2 = sg_wave_sum_d(s2);
    if (threadIdx.x == 0) out[0] = (float)(s1 * scale1) + (float)(s2 * scale2);
}
__global__ void __launch_bounds__(256) deepsdf_bwd_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                          const float* __restrict__ gloss, float* __restrict__ d_o, long n,
                                                          float inv_n, int nb1, const float* __restrict__ x,
                                                          const float* __restrict__ roww, float* __restrict__ dx, long m, int L,
                                                          float scale2) {
    const int b = blockIdx.x;
    if (b < nb1) {
        const float g = gloss[0] * inv_n;
        for (long e = (long)b * 256 + threadIdx.x; e < n; e += (long)nb1 * 256) {
            const float d = o[e] - t[e];
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            d_o[e] = g * sgn;
        }
    } else {
        const long nb2 = gridDim.x - nb1;
        const float g = gloss[0] * scale2;
        for (long e = (long)(b - nb1) * 256 + threadIdx.x; e < m; e += nb2 * 256) dx[e] = (roww ? roww[e / L] : 1.f) * g * x[e];
    }
}

// The same loss as ONE launch (round 6): the pass of deepsdf_fwd_kernel also writes the gradient for an upstream gradient of
// exactly 1 — the loss is the root of the trainer's backward, as with sg_loss_mean_split_fwd — and the workgroup that arrives
// last runs the finishing wave of deepsdf_final_kernel (the same sums in the same order: bit-identical loss; the gradients are
// those of deepsdf_bwd_kernel with gloss = 1, bit for bit).  One device-scope release per workgroup, by one thread.
__global__ void __launch_bounds__(256) deepsdf_fused_kernel(const float* __restrict__ o, const float* __restrict__ t, long n,
                                                            int nb1, const float* __restrict__ x, const float* __restrict__ roww,
                                                            long m, int L, double* __restrict__ partial, float* __restrict__ d_o,
                                                            float* __restrict__ dx, float inv_n, float scale2g, double scale1,
                                                            double scale2, float* __restrict__ out, unsigned* __restrict__ ticket) {
    __shared__ double red[4];
    __shared__ int last;
    double s = 0;
    const int b = blockIdx.x;
    const int nb2 = gridDim.x - nb1;
    if (b < nb1) {
        for (long e = (long)b * 256 + threadIdx.x; e < n; e += (long)nb1 * 256) {
            const float d = o[e] - t[e];
            s += (double)fabsf(d);
            if (d_o) d_o[e] = inv_n * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
    } else {
        for (long e = (long)(b - nb1) * 256 + threadIdx.x; e < m; e += (long)nb2 * 256) {
            const float v = x[e];
            const float w = roww ? roww[e / L] : 1.f;
            s += (double)(roww ? w * v * v : v * v);
            if (dx) dx[e] = w * scale2g * v;
        }
    }
    s = sg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[b < nb1 ? b : kRedBlocks + (b - nb1)] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        double s1 = 0, s2 = 0;
        for (int i = threadIdx.x; i < nb1; i += 64) s1 += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = threadIdx.x; i < nb2; i += 64)
            s2 += __hip_atomic_load(&partial[kRedBlocks + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s1 = sg_wave_sum_d(s1);
        s2 = sg_wave_sum_d(s2);
        if (threadIdx.x == 0) {
            out[0] = (float)(s1 * scale1) + (float)(s2 * scale2);
            ticket[0] = 0u;
        }
    }
}

// ---- gradient penalty ----------------------------------------------------------------------------------------------
// one workgroup per sample row: norm_b = ||g_b||_2
__global__ void __launch_bounds__(256) row_norm_kernel(const float* __restrict__ g, float* __restrict__ norms, long M) {
    const float* row = g + (long)blockIdx.x * M;
    double s = 0;
    if ((M & 3) == 0) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        for (long e = threadIdx.x; e < (M >> 2); e += 256) {
            const float4 v = r4[e];
            s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
        }
    } else {
        for (long e = threadIdx.x; e < M; e += 256) s += (double)(row[e] * row[e]);
    }
    __shared__ double red[4];
    s = sg_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) norms[blockIdx.x] = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void __launch_bounds__(64) gp_final_kernel(const float* __restrict__ norms, float* __restrict__ out, int B,
                                                      float weight) {
    double s = 0;
    for (int i = threadIdx.x; i < B; i += 64) {
        const double d = (double)norms[i] - 1.0;
        s += d * d;
    }
    s = sg_wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)(s / B * (double)weight);
}
// d loss / d g[b, :] = gloss * weight * 2 (norm_b - 1) / B * g[b, :] / norm_b   (0 where norm_b == 0, as torch.norm's backward)
__global__ void __launch_bounds__(256) gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ norms,
                                                     const float* __restrict__ gloss, float* __restrict__ dg, long M,
                                                     float coef) {
    const float nb = norms[blockIdx.y];
    const float c = nb > 0.f ? gloss[0] * coef * (nb - 1.f) / nb : 0.f;
    const long base = (long)blockIdx.y * M;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < M; e += (long)gridDim.x * 256) dg[base + e] = c * g[base + e];
}

// out[b, :] = alpha[b] * a[b, :] + (1 - alpha[b]) * b[b, :]   (two rounded products and one rounded sum, as the reference)
__global__ void __launch_bounds__(256) lerp_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ alpha, float* __restrict__ out, long M) {
    const float al = alpha[blockIdx.y], be = 1.f - al;
    const long base = (long)blockIdx.y * M;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < M; e += (long)gridDim.x * 256) {
        const float t1 = al * a[base + e], t2 = be * b[base + e];   // (contraction is off in this file: no fma)
        out[base + e] = t1 + t2;
    }
}

// ---- fade-in blend -------------------------------------------------------------------------------------------------
// out[b, c, s] = fade * x[b, c, s] + (c == 0 ? (1 - fade) * half[b, s] : 0);  x may be null (embedding only)
__global__ void __launch_bounds__(256) fade_blend_kernel(const float* __restrict__ x, const float* __restrict__ half,
                                                         float* __restrict__ out, int C, long S, float fade, float hscale) {
    const long b = blockIdx.z;
    const int c = blockIdx.y;
    const long base = (b * C + c) * S;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < S; e += (long)gridDim.x * 256) {
        float v = x ? fade * x[base + e] : 0.f;
        if (c == 0) {
            const float h = hscale * half[b * S + e];
            v = v + h;
        }
        out[base + e] = v;
    }
}
// out[b, s] = a * g[b, 0, s]
__global__ void __launch_bounds__(256) chan0_kernel(const float* __restrict__ g, float* __restrict__ out, int C, long S,
                                                    float a) {
    const long b = blockIdx.y;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < S; e += (long)gridDim.x * 256)
        out[b * S + e] = a * g[b * C * S + e];
}
// nearest-neighbour x[:, ::2, ::2, ::2] of [B, R, R, R] grids and its adjoint (zeros at the odd positions)
__global__ void __launch_bounds__(256) subsample2_kernel(const float* __restrict__ x, float* __restrict__ out, int R,
                                                         long total) {
    const int h = R >> 1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k = (int)(e % h);
        long r = e / h;
        const int j = (int)(r % h);
        r /= h;
        const int i = (int)(r % h);
        const long b = r / h;
        out[e] = x[((b * R + 2 * i) * R + 2 * j) * R + 2 * k];
    }
}
__global__ void __launch_bounds__(256) subsample2_adj_kernel(const float* __restrict__ g, float* __restrict__ out, int R,
                                                             long total) {
    const int h = R >> 1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k = (int)(e % R);
        long r = e / R;
        const int j = (int)(r % R);
        r /= R;
        const int i = (int)(r % R);
        const long b = r / R;
        out[e] = ((i | j | k) & 1) ? 0.f : g[((b * h + (i >> 1)) * h + (j >> 1)) * h + (k >> 1)];
    }
}

// ---- second-order term of tanh / sigmoid backward ------------------------------------------------------------------
// ActBwd computes dx = dy * f'(.) through the OUTPUT y: d dx / d y = dy * (-2y) (tanh), dy * (1 - 2y) (sigmoid)
__global__ void __launch_bounds__(256) act_bwd_dy_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                         const float* __restrict__ ggx, float* __restrict__ out, long n,
                                                         int act) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float yy = y[e];
        const float d = act == SG_ACT_TANH ? -2.f * yy : (act == SG_ACT_SIGMOID ? 1.f - 2.f * yy : 0.f);
        out[e] = ggx[e] * dy[e] * d;
    }
}

// ---- scatter_max over a ragged `batch` vector ---------------------------------------------------------------------
// order-preserving float <-> uint map, so that atomicMax on the image is the float maximum
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__global__ void __launch_bounds__(256) fill_u32_kernel(unsigned* __restrict__ p, long n, unsigned v) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) p[e] = v;
}
__global__ void __launch_bounds__(256) scatter_max_pass1(const float* __restrict__ x, const int64_t* __restrict__ batch,
                                                         unsigned* __restrict__ ord, long N, int C, long B) {
    const long total = N * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e / C;
        const int c = (int)(e - i * C);
        const long b = batch[i];
        if (b >= 0 && b < B) atomicMax(&ord[b * C + c], f2ord(x[e]));
    }
}
// first row that attains the maximum (deterministic: atomicMin over the row index)
__global__ void __launch_bounds__(256) scatter_max_pass2(const float* __restrict__ x, const int64_t* __restrict__ batch,
                                                         const unsigned* __restrict__ ord, int* __restrict__ arg, long N,
                                                         int C, long B) {
    const long total = N * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e / C;
        const int c = (int)(e - i * C);
        const long b = batch[i];
        if (b >= 0 && b < B && f2ord(x[e]) == ord[b * C + c]) atomicMin(&arg[b * C + c], (int)i);
    }
}
__global__ void __launch_bounds__(256) scatter_max_finish(const unsigned* __restrict__ ord, int* __restrict__ arg,
                                                          float* __restrict__ out, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const bool empty = arg[e] == 0x7fffffff;
        out[e] = empty ? 0.f : ord2f(ord[e]);      // torch_scatter fills segments without a member with 0
        if (empty) arg[e] = -1;
    }
}
__global__ void __launch_bounds__(256) arg_scatter_kernel(const float* __restrict__ dy, const int* __restrict__ arg,
                                                          float* __restrict__ dx, long BC, int C) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < BC; e += (long)gridDim.x * 256) {
        const int i = arg[e];
        if (i >= 0) dx[(long)i * C + (e % C)] = dy[e];     // (i, c) pairs are unique per (b, c): no conflicts
    }
}
__global__ void __launch_bounds__(256) arg_gather_kernel(const float* __restrict__ x, const int* __restrict__ arg,
                                                         float* __restrict__ out, long BC, int C) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < BC; e += (long)gridDim.x * 256) {
        const int i = arg[e];
        out[e] = i >= 0 ? x[(long)i * C + (e % C)] : 0.f;
    }
}

static int ew_blocks(long n) {
    long b = (n + 1023) / 1024;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_loss_workspace_bytes(void) { return 2 * kRedBlocks * sizeof(double); }   // (two partial arrays: sg_loss_deepsdf)

#define SG_CHECK_WS()                                                             \
    if (!workspace || workspace_bytes < sg_loss_workspace_bytes())                \
    SG_FAIL(SG_ERR_WORKSPACE, "%s: workspace too small (need %zu bytes)", __func__, sg_loss_workspace_bytes())

int sg_loss_weighted_l1_fwd(const float* out, const float* target, long n, float neg_weight, float* loss, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(out && target && loss && n > 0);
    SG_CHECK_WS();
    const int nb = red_grid(n);
    hipLaunchKernelGGL(wl1_fwd_kernel, dim3(nb), dim3(256), 0, stream, out, target, n, neg_weight, (double*)workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, loss, nb, 1.0 / (double)n);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_weighted_l1_bwd(const float* out, const float* target, const float* gloss, float* dout, long n, float neg_weight,
                            hipStream_t stream) {
    SG_CHECK_ARG(out && target && gloss && dout && n > 0);
    hipLaunchKernelGGL(wl1_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, out, target, gloss, dout, n, neg_weight,
                       (float)(1.0 / (double)n));
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_mean_split_fwd(const float* x, long n, long n_first, float w_first, float w_rest, float* loss, float* dx_unit,
                           hipStream_t stream) {
    SG_CHECK_ARG(x && loss && n > 0 && n_first >= 0 && n_first <= n);
    const float ca = n_first > 0 ? (float)((double)w_first / (double)n_first) : 0.f;
    const float cb = n > n_first ? (float)((double)w_rest / (double)(n - n_first)) : 0.f;
    hipLaunchKernelGGL(mean_split_fwd_kernel, dim3(1), dim3(256), 0, stream, x, n, n_first, w_first, w_rest, loss, dx_unit, ca, cb);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_mean_split_bwd(const float* gloss, float* dx, long n, long n_first, float w_first, float w_rest, hipStream_t stream) {
    SG_CHECK_ARG(gloss && dx && n > 0 && n_first >= 0 && n_first <= n);
    const float ca = n_first > 0 ? (float)((double)w_first / (double)n_first) : 0.f;
    const float cb = n > n_first ? (float)((double)w_rest / (double)(n - n_first)) : 0.f;
    hipLaunchKernelGGL(mean_split_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, gloss, dx, n, n_first, ca, cb);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_bce_fwd(const float* p, long n, float target, float* loss, hipStream_t stream) {
    SG_CHECK_ARG(p && loss && n > 0);
    hipLaunchKernelGGL(bce_fwd_kernel<0>, dim3(1), dim3(256), 0, stream, p, n, target, loss);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_bce_bwd(const float* p, const float* gloss, float* dp, long n, float target, hipStream_t stream) {
    SG_CHECK_ARG(p && gloss && dp && n > 0);
    hipLaunchKernelGGL(bce_bwd_kernel<0>, dim3(ew_blocks(n)), dim3(256), 0, stream, p, gloss, dp, n, target, (float)(1.0 / (double)n));
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_neg_mean_log_fwd(const float* p, long n, float* loss, hipStream_t stream) {
    SG_CHECK_ARG(p && loss && n > 0);
    hipLaunchKernelGGL(bce_fwd_kernel<1>, dim3(1), dim3(256), 0, stream, p, n, 1.f, loss);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_neg_mean_log_bwd(const float* p, const float* gloss, float* dp, long n, hipStream_t stream) {
    SG_CHECK_ARG(p && gloss && dp && n > 0);
    hipLaunchKernelGGL(bce_bwd_kernel<1>, dim3(ew_blocks(n)), dim3(256), 0, stream, p, gloss, dp, n, 1.f, (float)(1.0 / (double)n));
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_vae_reparam_fwd(const float* mean, const float* log_variance, const float* eps, float* z, long n, hipStream_t stream) {
    SG_CHECK_ARG(mean && log_variance && eps && z && n > 0);
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, mean, log_variance, eps, z, n);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_vae_reparam_bwd(const float* log_variance, const float* eps, const float* gz, float* dlog_variance, long n,
                       hipStream_t stream) {
    SG_CHECK_ARG(log_variance && eps && gz && dlog_variance && n > 0);
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, log_variance, eps, gz, dlog_variance, n);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_kld_fwd(const float* mean, const float* log_variance, long n, float* loss, void* workspace, size_t workspace_bytes,
                    hipStream_t stream) {
    SG_CHECK_ARG(mean && log_variance && loss && n > 0);
    SG_CHECK_WS();
    const int nb = red_grid(n);
    hipLaunchKernelGGL(kld_fwd_kernel, dim3(nb), dim3(256), 0, stream, mean, log_variance, n, (double*)workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, loss, nb, -0.5 / (double)n);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_kld_bwd(const float* mean, const float* log_variance, const float* gloss, float* dmean, float* dlog_variance,
                    long n, hipStream_t stream) {
    SG_CHECK_ARG(mean && log_variance && gloss && dmean && dlog_variance && n > 0);
    hipLaunchKernelGGL(kld_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, mean, log_variance, gloss, dmean,
                       dlog_variance, n, (float)(1.0 / (double)n));
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_count_sign_mismatch(const float* a, const float* b, long n, long long* count, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
    SG_CHECK_ARG(a && b && count && n > 0);
    SG_CHECK_WS();
    const int nb = red_grid(n);
    hipLaunchKernelGGL(sign_mismatch_kernel, dim3(nb), dim3(256), 0, stream, a, b, n, (unsigned long long*)workspace);
    hipLaunchKernelGGL(sign_mismatch_final_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long*)workspace, count, nb);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_meansq_fwd(const float* x, const float* row_weight, long rows, int L, double denom, float* loss, void* workspace,
                       size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(x && loss && rows > 0 && L > 0 && denom > 0);
    SG_CHECK_WS();
    const long n = rows * L;
    const int nb = red_grid(n);
    hipLaunchKernelGGL(sq_fwd_kernel, dim3(nb), dim3(256), 0, stream, x, row_weight, n, L, (double*)workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, loss, nb, 1.0 / denom);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_meansq_bwd(const float* x, const float* row_weight, const float* gloss, float* dx, long rows, int L, double denom,
                       hipStream_t stream) {
    SG_CHECK_ARG(x && gloss && dx && rows > 0 && L > 0 && denom > 0);
    const long n = rows * L;
    hipLaunchKernelGGL(sq_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, x, row_weight, gloss, dx, n, L,
                       (float)(2.0 / denom));
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_deepsdf_fwd(const float* out, const float* target, long n, const float* z, const float* row_weight, long rows, int L,
                        double denom, float* loss, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(out && target && z && loss && n > 0 && rows > 0 && L > 0 && denom > 0);
    SG_CHECK_WS();
    const long m = rows * L;
    const int nb1 = red_grid(n), nb2 = red_grid(m);
    hipLaunchKernelGGL(deepsdf_fwd_kernel, dim3(nb1 + nb2), dim3(256), 0, stream, out, target, n, nb1, z, row_weight, m, L,
                       (double*)workspace);
    hipLaunchKernelGGL(deepsdf_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)workspace, loss, nb1, nb2,
                       1.0 / (double)n, 1.0 / denom);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_deepsdf_fused(const float* out, const float* target, long n, const float* z, const float* row_weight, long rows, int L,
                          double denom, float* loss, float* dout_unit, float* dz_unit, void* workspace, size_t workspace_bytes,
                          unsigned* ticket, hipStream_t stream) {
    SG_CHECK_ARG(out && target && z && loss && ticket && n > 0 && rows > 0 && L > 0 && denom > 0);
    SG_CHECK_WS();
    const long m = rows * L;
    const int nb1 = red_grid(n), nb2 = red_grid(m);
    hipLaunchKernelGGL(deepsdf_fused_kernel, dim3(nb1 + nb2), dim3(256), 0, stream, out, target, n, nb1, z, row_weight, m, L,
                       (double*)workspace, dout_unit, dz_unit, (float)(1.0 / (double)n), (float)(2.0 / denom), 1.0 / (double)n,
                       1.0 / denom, loss, ticket);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_loss_deepsdf_bwd(const float* out, const float* target, long n, const float* z, const float* row_weight, long rows, int L,
                        double denom, const float* gloss, float* dout, float* dz, hipStream_t stream) {
    SG_CHECK_ARG(out && target && z && gloss && dout && dz && n > 0 && rows > 0 && L > 0 && denom > 0);
    const long m = rows * L;
    const int nb1 = ew_blocks(n), nb2 = ew_blocks(m);
    hipLaunchKernelGGL(deepsdf_bwd_kernel, dim3(nb1 + nb2), dim3(256), 0, stream, out, target, gloss, dout, n,
                       (float)(1.0 / (double)n), nb1, z, row_weight, dz, m, L, (float)(2.0 / denom));
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_gradient_penalty_fwd(const float* grad, long B, long M, float weight, float* norms, float* loss, hipStream_t stream) {
    SG_CHECK_ARG(grad && norms && loss && B > 0 && B < 65536 && M > 0);
    hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)B), dim3(256), 0, stream, grad, norms, M);
    hipLaunchKernelGGL(gp_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)norms, loss, (int)B, weight);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_gradient_penalty_bwd(const float* grad, const float* norms, const float* gloss, float* dgrad, long B, long M,
                            float weight, hipStream_t stream) {
    SG_CHECK_ARG(grad && norms && gloss && dgrad && B > 0 && B < 65536 && M > 0);
    long bx = (M + 2047) / 2048;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(gp_bwd_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, stream, grad, norms, gloss, dgrad, M,
                       (float)(2.0 * (double)weight / (double)B));
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_lerp_rows(const float* a, const float* b, const float* alpha, float* out, long B, long M, hipStream_t stream) {
    SG_CHECK_ARG(a && b && alpha && out && B > 0 && B < 65536 && M > 0);
    long bx = (M + 2047) / 2048;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(lerp_rows_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, stream, a, b, alpha, out, M);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_fade_blend(const float* x, const float* half, float* out, long B, int C, long S, float fade, float half_scale,
                  hipStream_t stream) {
    SG_CHECK_ARG(half && out && B > 0 && B < 65536 && C > 0 && C < 65536 && S > 0);
    long bx = (S + 1023) / 1024;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(fade_blend_kernel, dim3((unsigned)bx, (unsigned)C, (unsigned)B), dim3(256), 0, stream, x, half, out, C,
                       S, fade, half_scale);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_channel0(const float* g, float* out, long B, int C, long S, float scale, hipStream_t stream) {
    SG_CHECK_ARG(g && out && B > 0 && B < 65536 && C > 0 && S > 0);
    long bx = (S + 1023) / 1024;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(chan0_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, stream, g, out, C, S, scale);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_subsample2(const float* x, float* out, long B, int R, hipStream_t stream) {
    SG_CHECK_ARG(x && out && B > 0 && R >= 2 && (R & 1) == 0);
    const long total = B * (R / 2) * (R / 2) * (R / 2);
    hipLaunchKernelGGL(subsample2_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, x, out, R, total);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_subsample2_adjoint(const float* g, float* out, long B, int R, hipStream_t stream) {
    SG_CHECK_ARG(g && out && B > 0 && R >= 2 && (R & 1) == 0);
    const long total = B * (long)R * R * R;
    hipLaunchKernelGGL(subsample2_adj_kernel, dim3(ew_blocks(total)), dim3(256), 0, stream, g, out, R, total);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_act_bwd_dy(const float* y, const float* dy, const float* ggx, float* out, long n, int act, hipStream_t stream) {
    SG_CHECK_ARG(y && dy && ggx && out && n > 0 && (act == SG_ACT_TANH || act == SG_ACT_SIGMOID));
    hipLaunchKernelGGL(act_bwd_dy_kernel, dim3(ew_blocks(n)), dim3(256), 0, stream, y, dy, ggx, out, n, act);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
size_t sg_scatter_max_workspace_bytes(long B, int C) { return (size_t)B * C * sizeof(unsigned); }
int sg_scatter_max_fwd(const float* x, const int64_t* batch, float* out, int* arg, long N, long B, int C, void* workspace,
                       size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(x && batch && out && arg && N > 0 && N < (1L << 31) && B > 0 && C > 0);
    if (!workspace || workspace_bytes < sg_scatter_max_workspace_bytes(B, C))
        SG_FAIL(SG_ERR_WORKSPACE, "sg_scatter_max_fwd: workspace too small");
    unsigned* ord = (unsigned*)workspace;
    hipLaunchKernelGGL(fill_u32_kernel, dim3(ew_blocks(B * C)), dim3(256), 0, stream, ord, B * C, 0u);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(ew_blocks(B * C)), dim3(256), 0, stream, (unsigned*)arg, B * C, 0x7fffffffu);
    hipLaunchKernelGGL(scatter_max_pass1, dim3(ew_blocks(N * C)), dim3(256), 0, stream, x, batch, ord, N, C, B);
    hipLaunchKernelGGL(scatter_max_pass2, dim3(ew_blocks(N * C)), dim3(256), 0, stream, x, batch, (const unsigned*)ord, arg,
                       N, C, B);
    hipLaunchKernelGGL(scatter_max_finish, dim3(ew_blocks(B * C)), dim3(256), 0, stream, (const unsigned*)ord, arg, out,
                       B * C);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_scatter_max_scatter(const float* dy, const int* arg, float* dx, long N, long B, int C, hipStream_t stream) {
    SG_CHECK_ARG(dy && arg && dx && N > 0 && B > 0 && C > 0);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(ew_blocks(N * C)), dim3(256), 0, stream, (unsigned*)dx, N * C, 0u);
    hipLaunchKernelGGL(arg_scatter_kernel, dim3(ew_blocks(B * C)), dim3(256), 0, stream, dy, arg, dx, B * C, C);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_scatter_max_gather(const float* x, const int* arg, float* out, long N, long B, int C, hipStream_t stream) {
    SG_CHECK_ARG(x && arg && out && N > 0 && B > 0 && C > 0);
    hipLaunchKernelGGL(arg_gather_kernel, dim3(ew_blocks(B * C)), dim3(256), 0, stream, x, arg, out, B * C, C);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"
