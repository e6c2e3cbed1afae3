// shapegan_amd/csrc/pointnet.hip — kernels of the PointNet-discriminator GAN family:
// LayerNorm(+per-shape bias, +ReLU) forward / backward for the SDFGenerator MLP (model/point_sdf_net.py:49-119),
// max over the points of a shape with its two adjoints (PointNet pooling, model/point_sdf_net.py:39-42), and a
// two-pass column sum for tall matrices (bias gradients of per-point Linear layers).
#include "common.h"

namespace sg {

constexpr int kLNMaxPerLane = 8;  // channels <= 512

// One wave per row.  z = x[r] + rowbias[r / rows_per_shape];  y = act(gamma * (z - mean) * rstd + beta).
// Two-pass statistics in registers (mean first, then the centred sum of squares: no cancellation).
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ rowbias, long rows_per_shape,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, long ldy, float* __restrict__ mean,
                                                            float* __restrict__ rstd, long R, int C, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int npl = (C + 63) >> 6;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (long)gridDim.x * 4) {
        const float* xr = x + r * ldx;
        const float* zb = rowbias ? rowbias + (r / rows_per_shape) * C : nullptr;
        float v[kLNMaxPerLane];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < kLNMaxPerLane; ++q) {
            const int c = q * 64 + lane;
            v[q] = (q < npl && c < C) ? xr[c] + (zb ? zb[c] : 0.f) : 0.f;
            s += v[q];
        }
        const float mu = sg_wave_sum(s) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < kLNMaxPerLane; ++q) {
            const int c = q * 64 + lane;
            const float d = (q < npl && c < C) ? v[q] - mu : 0.f;
            sq += d * d;
        }
        const float rs = rsqrtf(sg_wave_sum(sq) / (float)C + eps);
        float* yr = y + r * ldy;
#pragma unroll
        for (int q = 0; q < kLNMaxPerLane; ++q) {
            const int c = q * 64 + lane;
            if (q < npl && c < C) yr[c] = sg_apply_act((v[q] - mu) * rs * gamma[c] + beta[c], act, 0.f);
        }
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

// g = dy * act'(y);  gh = g * gamma;  dz = rstd * (gh - mean_c(gh) - xhat * mean_c(gh * xhat));  per-workgroup partial
// sums of g * xhat (dgamma) and g (dbeta) go to part[2][gridDim.x][C]; dz is also the gradient of the per-shape bias rows.
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ rowbias, long rows_per_shape,
                                                            const float* __restrict__ gamma, const float* __restrict__ y,
                                                            long ldy, const float* __restrict__ dy, long lddy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ dz, long lddz, float* __restrict__ part,
                                                            long R, int C, int act) {
    __shared__ float red[2][4][kLNMaxPerLane * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npl = (C + 63) >> 6;
    float ag[kLNMaxPerLane], ab[kLNMaxPerLane];
#pragma unroll
    for (int q = 0; q < kLNMaxPerLane; ++q) ag[q] = ab[q] = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
        const float* xr = x + r * ldx;
        const float* zb = rowbias ? rowbias + (r / rows_per_shape) * C : nullptr;
        const float mu = mean[r], rs = rstd[r];
        float xh[kLNMaxPerLane], gh[kLNMaxPerLane];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < kLNMaxPerLane; ++q) {
            const int c = q * 64 + lane;
            xh[q] = gh[q] = 0.f;
            if (q < npl && c < C) {
                float g = dy[r * lddy + c];
                if (act == SG_ACT_RELU) g = y[r * ldy + c] > 0.f ? g : 0.f;
                xh[q] = (xr[c] + (zb ? zb[c] : 0.f) - mu) * rs;
                gh[q] = g * gamma[c];
                ag[q] += g * xh[q];
                ab[q] += g;
                s1 += gh[q];
                s2 += gh[q] * xh[q];
            }
        }
        const float m1 = sg_wave_sum(s1) / (float)C, m2 = sg_wave_sum(s2) / (float)C;
#pragma unroll
        for (int q = 0; q < kLNMaxPerLane; ++q) {
            const int c = q * 64 + lane;
            if (q < npl && c < C) dz[r * lddz + c] = rs * (gh[q] - m1 - xh[q] * m2);
        }
    }
#pragma unroll
    for (int q = 0; q < kLNMaxPerLane; ++q) {
        red[0][wave][q * 64 + lane] = ag[q];
        red[1][wave][q * 64 + lane] = ab[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        part[(long)blockIdx.x * C + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        part[((long)gridDim.x + blockIdx.x) * C + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// out[b][j] = sum_i part[b][i][j] over nrows rows (second pass of the two-pass column sums; blockIdx.y = batch)
__global__ void __launch_bounds__(256) colsum_small_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int nrows, int cols) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    const float* p = part + (long)blockIdx.y * nrows * cols;
    float s = 0.f;
    for (int i = 0; i < nrows; ++i) s += p[(long)i * cols + j];
    out[(long)blockIdx.y * cols + j] = s;
}

// first pass: workgroup g of batch b sums rows g, g + G, ... (4 row lanes x 64-column strips, coalesced) into part[b][g][cols]
__global__ void __launch_bounds__(256) colsum_tall_kernel(const float* __restrict__ x, float* __restrict__ part, long rows,
                                                          int cols, long ld, long batch_stride) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (long)blockIdx.y * batch_stride;
    float* pb = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * cols;
    for (int c0 = 0; c0 < cols; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < cols)
            for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) s += xb[r * ld + c];
        red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && c < cols) pb[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        __syncthreads();
    }
}

// x [B][P][C] -> out[b][c] = max_p x[b][p][c], idx[b][c] = first p attaining it.  One workgroup per (shape, 64 channels):
// 4 waves take interleaved points, coalesced over channels; ties / NaN: first occurrence, NaN wins (torch.max semantics).
__global__ void __launch_bounds__(256) segmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         int* __restrict__ idx, long P, int C) {
    __shared__ float bv[4][64];
    __shared__ int bi[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long b = blockIdx.y;
    float best = -INFINITY;
    int bp = 0x7fffffff;
    bool nan = false;
    if (c < C) {
        const float* xb = x + b * P * C + c;
        for (long p = wave; p < P; p += 4) {
            const float v = xb[p * C];
            if (nan) continue;
            if (v != v) {
                best = v;
                bp = (int)p;
                nan = true;
            } else if (v > best || bp == 0x7fffffff) {
                best = v;
                bp = (int)p;
            }
        }
    }
    bv[wave][lane] = best;
    bi[wave][lane] = bp;
    __syncthreads();
    if (wave == 0 && c < C) {
        float v = bv[0][lane];
        int p = bi[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float u = bv[w][lane];
            const int q = bi[w][lane];
            if (q == 0x7fffffff) continue;
            const bool vn = v != v, un = u != u;
            const bool take = p == 0x7fffffff || (un && (!vn || q < p)) || (!vn && !un && (u > v || (u == v && q < p)));
            if (take) {
                v = u;
                p = q;
            }
        }
        out[b * C + c] = v;
        idx[b * C + c] = p;
    }
}

// dx[b][p][c] = (p == idx[b][c]) ? dy[b][c] : 0   (adjoint of the max; every element written)
__global__ void __launch_bounds__(256) segmax_scatter_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                             float* __restrict__ dx, long B, long P, int C) {
    const long total = B * P * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long bp = e / C;
        const long b = bp / P, p = bp - b * P;
        dx[e] = idx[b * C + c] == (int)p ? dy[b * C + c] : 0.f;
    }
}

// out[b][c] = x[b][idx[b][c]][c]   (adjoint of the scatter: the double-backward of the max)
__global__ void __launch_bounds__(256) segmax_gather_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                            float* __restrict__ out, long B, long P, int C) {
    const long total = B * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long b = e / C;
        out[e] = x[(b * P + idx[e]) * C + c];
    }
}

static int ln_blocks(long R) {
    long b = (R + 3) / 4;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_layernorm_fwd(const float* x, long ldx, const float* rowbias, long rows_per_shape, const float* gamma,
                     const float* beta, float* y, long ldy, float* mean, float* rstd, long R, int C, float eps, int act,
                     hipStream_t stream) {
    SG_CHECK_ARG(x && gamma && beta && y && mean && rstd && R > 0 && C > 0 && C <= 64 * kLNMaxPerLane);
    SG_CHECK_ARG(ldx >= C && ldy >= C && (!rowbias || rows_per_shape > 0) && (act == SG_ACT_NONE || act == SG_ACT_RELU));
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ln_blocks(R)), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape,
                       gamma, beta, y, ldy, mean, rstd, R, C, eps, act);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

size_t sg_layernorm_bwd_workspace_bytes(long R, int C) { return (size_t)2 * ln_blocks(R) * C * sizeof(float); }

int sg_layernorm_bwd(const float* x, long ldx, const float* rowbias, long rows_per_shape, const float* gamma, const float* y,
                     long ldy, const float* dy, long lddy, const float* mean, const float* rstd, float* dz, long lddz,
                     float* dgamma, float* dbeta, long R, int C, int act, void* workspace, size_t workspace_bytes,
                     hipStream_t stream) {
    SG_CHECK_ARG(x && gamma && dy && mean && rstd && dz && dgamma && dbeta && R > 0 && C > 0 && C <= 64 * kLNMaxPerLane);
    SG_CHECK_ARG((act == SG_ACT_NONE || (act == SG_ACT_RELU && y)) && (!rowbias || rows_per_shape > 0));
    if (!workspace || workspace_bytes < sg_layernorm_bwd_workspace_bytes(R, C))
        SG_FAIL(SG_ERR_WORKSPACE, "sg_layernorm_bwd: workspace too small");
    const int nb = ln_blocks(R);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape, gamma, y, ldy,
                       dy, lddy, mean, rstd, dz, lddz, part, R, C, act);
    hipLaunchKernelGGL(colsum_small_kernel, dim3(sg_cdiv(C, 256), 1), dim3(256), 0, stream, part, dgamma, nb, C);
    hipLaunchKernelGGL(colsum_small_kernel, dim3(sg_cdiv(C, 256), 1), dim3(256), 0, stream, part + (long)nb * C, dbeta, nb, C);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

static int colsum_blocks(long rows) {
    long b = (rows + 63) / 64;
    return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}
size_t sg_colsum_tall_workspace_bytes(long batch, long rows, int cols) {
    return (size_t)batch * colsum_blocks(rows) * cols * sizeof(float);
}

int sg_colsum_tall(const float* x, float* out, long batch, long batch_stride, long rows, int cols, long ld, void* workspace,
                   size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(x && out && batch > 0 && batch <= 65535 && rows > 0 && cols > 0 && ld >= cols);
    if (!workspace || workspace_bytes < sg_colsum_tall_workspace_bytes(batch, rows, cols))
        SG_FAIL(SG_ERR_WORKSPACE, "sg_colsum_tall: workspace too small");
    const int nb = colsum_blocks(rows);
    hipLaunchKernelGGL(colsum_tall_kernel, dim3(nb, (unsigned)batch), dim3(256), 0, stream, x, (float*)workspace, rows, cols,
                       ld, batch_stride);
    hipLaunchKernelGGL(colsum_small_kernel, dim3(sg_cdiv(cols, 256), (unsigned)batch), dim3(256), 0, stream,
                       (const float*)workspace, out, nb, cols);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_segmax_fwd(const float* x, float* out, int* idx, long B, long P, int C, hipStream_t stream) {
    SG_CHECK_ARG(x && out && idx && B > 0 && B <= 65535 && P > 0 && P < 0x7fffffff && C > 0);
    hipLaunchKernelGGL(segmax_fwd_kernel, dim3(sg_cdiv(C, 64), (unsigned)B), dim3(256), 0, stream, x, out, idx, P, C);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_segmax_scatter(const float* dy, const int* idx, float* dx, long B, long P, int C, hipStream_t stream) {
    SG_CHECK_ARG(dy && idx && dx && B > 0 && P > 0 && C > 0);
    long blocks = (B * P * C + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(segmax_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dy, idx, dx, B, P, C);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_segmax_gather(const float* x, const int* idx, float* out, long B, long P, int C, hipStream_t stream) {
    SG_CHECK_ARG(x && idx && out && B > 0 && P > 0 && C > 0);
    long blocks = (B * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(segmax_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, idx, out, B, P, C);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"
