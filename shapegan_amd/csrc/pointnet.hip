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

// ---- the same two kernels for rows that are 16-byte addressable (C, ldx, ldy ... multiples of 4, aligned bases): a lane owns
// 4 ADJACENT channels per 256-channel group (b128 loads / stores: one instruction per row and tensor where the kernels above
// issue four), statistics and formulas unchanged.  NQ = groups of 256 channels (C <= 512).
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <int NQ>
__global__ void __launch_bounds__(256) layernorm_fwd4_kernel(const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ rowbias, long rows_per_shape,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, long ldy, float* __restrict__ mean,
                                                             float* __restrict__ rstd, long R, int C, float eps, int act) {
    const int lane = threadIdx.x & 63;
    float4 gm[NQ], bt[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = (q * 64 + lane) * 4;
        gm[q] = c < C ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[q] = c < C ? ld4(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (long)gridDim.x * 4) {
        const float* xr = x + r * ldx;
        const float* zb = rowbias ? rowbias + (r / rows_per_shape) * C : nullptr;
        float4 v[NQ];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (q * 64 + lane) * 4;
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                v[q] = ld4(xr + c);
                if (zb) {
                    const float4 z = ld4(zb + c);
                    v[q].x += z.x, v[q].y += z.y, v[q].z += z.z, v[q].w += z.w;
                }
            }
            s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        }
        const float mu = sg_wave_sum(s) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (q * 64 + lane) * 4;
            if (c < C) {
                const float a = v[q].x - mu, b = v[q].y - mu, cc = v[q].z - mu, d = v[q].w - mu;
                sq += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(sg_wave_sum(sq) / (float)C + eps);
        float* yr = y + r * ldy;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (q * 64 + lane) * 4;
            if (c < C) {
                float4 o;
                o.x = sg_apply_act((v[q].x - mu) * rs * gm[q].x + bt[q].x, act, 0.f);
                o.y = sg_apply_act((v[q].y - mu) * rs * gm[q].y + bt[q].y, act, 0.f);
                o.z = sg_apply_act((v[q].z - mu) * rs * gm[q].z + bt[q].z, act, 0.f);
                o.w = sg_apply_act((v[q].w - mu) * rs * gm[q].w + bt[q].w, act, 0.f);
                *reinterpret_cast<float4*>(yr + c) = o;
            }
        }
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

template <int NQ>
__global__ void __launch_bounds__(256) layernorm_bwd4_kernel(const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ rowbias, long rows_per_shape,
                                                             const float* __restrict__ gamma, const float* __restrict__ y,
                                                             long ldy, const float* __restrict__ dy, long lddy,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ dz, long lddz, float* __restrict__ part,
                                                             long R, int C, int act) {
    __shared__ float red[2][4][kLNMaxPerLane * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gm[NQ], ag[NQ], ab[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = (q * 64 + lane) * 4;
        gm[q] = c < C ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[q] = ab[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
        const float* xr = x + r * ldx;
        const float* zb = rowbias ? rowbias + (r / rows_per_shape) * C : nullptr;
        const float mu = mean[r], rs = rstd[r];
        float4 xh[NQ], gh[NQ];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (q * 64 + lane) * 4;
            xh[q] = gh[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                float4 g = ld4(dy + r * lddy + c);
                if (act == SG_ACT_RELU) {
                    const float4 yy = ld4(y + r * ldy + c);
                    g.x = yy.x > 0.f ? g.x : 0.f, g.y = yy.y > 0.f ? g.y : 0.f, g.z = yy.z > 0.f ? g.z : 0.f, g.w = yy.w > 0.f ? g.w : 0.f;
                }
                float4 xv = ld4(xr + c);
                if (zb) {
                    const float4 z = ld4(zb + c);
                    xv.x += z.x, xv.y += z.y, xv.z += z.z, xv.w += z.w;
                }
                xh[q] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                gh[q] = make_float4(g.x * gm[q].x, g.y * gm[q].y, g.z * gm[q].z, g.w * gm[q].w);
                ag[q].x += g.x * xh[q].x, ag[q].y += g.y * xh[q].y, ag[q].z += g.z * xh[q].z, ag[q].w += g.w * xh[q].w;
                ab[q].x += g.x, ab[q].y += g.y, ab[q].z += g.z, ab[q].w += g.w;
                s1 += (gh[q].x + gh[q].y) + (gh[q].z + gh[q].w);
                s2 += (gh[q].x * xh[q].x + gh[q].y * xh[q].y) + (gh[q].z * xh[q].z + gh[q].w * xh[q].w);
            }
        }
        const float m1 = sg_wave_sum(s1) / (float)C, m2 = sg_wave_sum(s2) / (float)C;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (q * 64 + lane) * 4;
            if (c < C) {
                float4 o;
                o.x = rs * (gh[q].x - m1 - xh[q].x * m2);
                o.y = rs * (gh[q].y - m1 - xh[q].y * m2);
                o.z = rs * (gh[q].z - m1 - xh[q].z * m2);
                o.w = rs * (gh[q].w - m1 - xh[q].w * m2);
                *reinterpret_cast<float4*>(dz + r * lddz + c) = o;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = (q * 64 + lane) * 4;
        if (c < C) {
            *reinterpret_cast<float4*>(&red[0][wave][c]) = ag[q];
            *reinterpret_cast<float4*>(&red[1][wave][c]) = ab[q];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        part[(long)blockIdx.x * C + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        part[((long)gridDim.x + blockIdx.x) * C + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// out[b][j] = sum_i part[b][i][j] over nrows rows (second pass of the two-pass column sums; blockIdx.y = batch).  One workgroup
// per 64 columns: 16 waves take 16 contiguous row slices (four independent accumulators each, so four loads are in flight per
// lane), LDS, one wave adds the 16 slice sums in a fixed order.  (Round 2's version walked all rows in ONE thread per column from
// one or two workgroups: 1024 dependent adds behind 1024 load latencies = 164 us for a 2 MB table, 17 % of a point-GAN step.)
// (out1: where batch 1 goes instead of out + cols — the two sums of a LayerNorm backward live in different tensors)
__global__ void __launch_bounds__(1024) colsum_small_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            float* __restrict__ out1, int nrows, int cols) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const float* p = part + (long)blockIdx.y * nrows * cols + j;
    const int per = (nrows + 15) / 16;
    const int i0 = wave * per, i1 = min(nrows, i0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < cols) {
        int i = i0;
        for (; i + 3 < i1; i += 4) {
            a0 += p[(long)i * cols];
            a1 += p[(long)(i + 1) * cols];
            a2 += p[(long)(i + 2) * cols];
            a3 += p[(long)(i + 3) * cols];
        }
        for (; i < i1; ++i) a0 += p[(long)i * cols];
    }
    red[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && j < cols) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        if (out1 && blockIdx.y == 1)
            out1[j] = t;
        else
            out[(long)blockIdx.y * cols + j] = t;
    }
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
// the same for cols % 4 == 0, ld % 4 == 0 and 16-byte aligned x: a lane owns 4 adjacent columns (b128 loads, a wave covers 256
// columns of a row), two rows per step are in flight per lane; the strips of 256 columns are walked one after the other
__global__ void __launch_bounds__(256) colsum_tall4_kernel(const float* __restrict__ x, float* __restrict__ part, long rows,
                                                           int cols, long ld, long batch_stride) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (long)blockIdx.y * batch_stride;
    float* pb = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * cols;
    const long step = (long)gridDim.x * 4;
    for (int c0 = 0; c0 < cols; c0 += 256) {
        const int c = c0 + lane * 4;
        float4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (c < cols) {
            long r = (long)blockIdx.x * 4 + wave;
            for (; r + step < rows; r += 2 * step) {
                const float4 u = *reinterpret_cast<const float4*>(xb + r * ld + c);
                const float4 v = *reinterpret_cast<const float4*>(xb + (r + step) * ld + c);
                a.x += u.x, a.y += u.y, a.z += u.z, a.w += u.w;
                b.x += v.x, b.y += v.y, b.z += v.z, b.w += v.w;
            }
            if (r < rows) {
                const float4 u = *reinterpret_cast<const float4*>(xb + r * ld + c);
                a.x += u.x, a.y += u.y, a.z += u.z, a.w += u.w;
            }
        }
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        red[wave][lane] = a;
        __syncthreads();
        if (wave == 0 && c < cols) {
            float4 t;
            t.x = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
            t.y = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
            t.z = (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z);
            t.w = (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w);
            *reinterpret_cast<float4*>(pb + c) = t;
        }
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

// The same reduction cut into point chunks for the b128 case (C % 4 == 0): workgroup = (256 channels, chunk, shape), a lane owns
// 4 adjacent channels, the 4 waves take interleaved points of the chunk two at a time (two 16-byte loads in flight per lane),
// and a second small kernel merges the chunks in ascending order (an earlier chunk wins ties, the first NaN wins): the same
// result as the kernel above.  That one walks a shape's points in 4 waves per 64 channels — 48 workgroups and 8192 dependent
// 4-byte loads per lane at 6 x 32768 points: 1.5 - 3.7 ms per call, 16 % of a point-GAN step, for a 400 MB read.
constexpr int kSegNone = 0x7fffffff;
__device__ __forceinline__ void seg_upd(float& best, int& bp, float v, int p) {
    if (best == best && (v != v || v > best || bp == kSegNone)) {
        best = v;
        bp = p;
    }
}
// does (u, q) beat (v, p)?  q / p are point indices (kSegNone: empty)
__device__ __forceinline__ bool seg_beats(float u, int q, float v, int p) {
    if (q == kSegNone) return false;
    const bool vn = v != v, un = u != u;
    return p == kSegNone || (un && (!vn || q < p)) || (!vn && !un && (u > v || (u == v && q < p)));
}
__global__ void __launch_bounds__(256) segmax_part_kernel(const float* __restrict__ x, float* __restrict__ pv,
                                                          int* __restrict__ pi, long P, int C, int nchunk) {
    __shared__ float4 bv[4][64];
    __shared__ int4 bi[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int chunk = blockIdx.y;
    const long b = blockIdx.z;
    const long per = (P + nchunk - 1) / nchunk;
    const long p0 = chunk * per, p1 = min(P, p0 + per);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bp[4] = {kSegNone, kSegNone, kSegNone, kSegNone};
    if (c < C) {
        const float* xb = x + b * P * C + c;
        long p = p0 + wave;
        for (; p + 4 < p1; p += 8) {
            const float4 u = *reinterpret_cast<const float4*>(xb + p * C);
            const float4 v = *reinterpret_cast<const float4*>(xb + (p + 4) * C);
            seg_upd(best[0], bp[0], u.x, (int)p), seg_upd(best[1], bp[1], u.y, (int)p);
            seg_upd(best[2], bp[2], u.z, (int)p), seg_upd(best[3], bp[3], u.w, (int)p);
            seg_upd(best[0], bp[0], v.x, (int)p + 4), seg_upd(best[1], bp[1], v.y, (int)p + 4);
            seg_upd(best[2], bp[2], v.z, (int)p + 4), seg_upd(best[3], bp[3], v.w, (int)p + 4);
        }
        if (p < p1) {
            const float4 u = *reinterpret_cast<const float4*>(xb + p * C);
            seg_upd(best[0], bp[0], u.x, (int)p), seg_upd(best[1], bp[1], u.y, (int)p);
            seg_upd(best[2], bp[2], u.z, (int)p), seg_upd(best[3], bp[3], u.w, (int)p);
        }
    }
    bv[wave][lane] = make_float4(best[0], best[1], best[2], best[3]);
    bi[wave][lane] = make_int4(bp[0], bp[1], bp[2], bp[3]);
    __syncthreads();
    if (wave == 0 && c < C) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 u = bv[w][lane];
            const int4 q = bi[w][lane];
            if (seg_beats(u.x, q.x, best[0], bp[0])) best[0] = u.x, bp[0] = q.x;
            if (seg_beats(u.y, q.y, best[1], bp[1])) best[1] = u.y, bp[1] = q.y;
            if (seg_beats(u.z, q.z, best[2], bp[2])) best[2] = u.z, bp[2] = q.z;
            if (seg_beats(u.w, q.w, best[3], bp[3])) best[3] = u.w, bp[3] = q.w;
        }
        const long o = (b * nchunk + chunk) * C + c;
        *reinterpret_cast<float4*>(pv + o) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<int4*>(pi + o) = make_int4(bp[0], bp[1], bp[2], bp[3]);
    }
}
// (block = 64 channels x 4 chunk phases, four partials of a phase in flight: a cloud of 16 384 points arrives as 512 partials per
//  channel — one thread walking them one dependent load at a time took 260 us for 36 x 512 channels.  (value, point) pairs are
//  totally ordered by seg_beats, so the order in which partials are combined does not change the result.)
__global__ void __launch_bounds__(256) segmax_merge_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                           float* __restrict__ out, int* __restrict__ idx, int nchunk, int C) {
    __shared__ float sv[4][64];
    __shared__ int sp[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long b = blockIdx.y;
    float v = -INFINITY;
    int p = kSegNone;
    if (c < C) {
        const float* v0 = pv + b * nchunk * C + c;
        const int* p0 = pi + b * nchunk * C + c;
        int k = ph;
        for (; k + 12 < nchunk; k += 16) {
            float u[4];
            int q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u[j] = v0[(long)(k + 4 * j) * C];
                q[j] = p0[(long)(k + 4 * j) * C];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (seg_beats(u[j], q[j], v, p)) v = u[j], p = q[j];
        }
        for (; k < nchunk; k += 4) {
            const float u = v0[(long)k * C];
            const int q = p0[(long)k * C];
            if (seg_beats(u, q, v, p)) v = u, p = q;
        }
    }
    sv[ph][cl] = v;
    sp[ph][cl] = p;
    __syncthreads();
    if (ph == 0 && c < C) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (seg_beats(sv[w][cl], sp[w][cl], v, p)) v = sv[w][cl], p = sp[w][cl];
        out[b * C + c] = v;
        idx[b * C + c] = p;
    }
}

// ---- the selected points' last layer: only the DIAGONAL of nn1's 512-wide output is used (row c of a cloud's gathered batch is
// the point that holds the maximum of channel c, model/point_sdf_net.py PointNet.forward_selected), so the 256 -> 512 Linear of
// the recorded pass is a row-wise dot product with the row's own weight row instead of a [B*512, 512] GEMM of which 511/512 is
// thrown away — and so are its backward and double backward.  Three bilinear kernels, closed under differentiation:
//   rowdot:   out[r]    = bias[c] + sum_k h[r][k] w[c][k]            r = b C + c
//   rowscale: out[r][k] = g[r] w[c][k]
//   rowouter: out[c][k] = sum_b g[b C + c] h[b C + c][k]
__global__ void __launch_bounds__(256) rowdot_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, long R, int C, int K) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63, c = (int)(r % C);
    const float *hr = h + r * K, *wr = w + (long)c * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(hr[k], wr[k], s);
    s = sg_wave_sum(s);
    if (lane == 0) out[r] = s + (bias ? bias[c] : 0.f);
}
__global__ void __launch_bounds__(256) rowscale_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out,
                                                       long R, int C, int K) {
    const long total = R * K;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / K;
        const int k = (int)(e - r * K);
        out[e] = g[r] * w[(long)(r % C) * K + k];
    }
}
__global__ void __launch_bounds__(256) rowouter_kernel(const float* __restrict__ g, const float* __restrict__ h, float* __restrict__ out,
                                                       long B, int C, int K) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)C * K) return;
    const int c = (int)(e / K), k = (int)(e - (long)c * K);
    float s = 0.f;
    for (long b = 0; b < B; ++b) s = fmaf(g[b * C + c], h[(b * C + c) * K + k], s);
    out[e] = s;
}

// ---- adjoint of gathering the selected points of every cloud (PointNet.gather_points): dx[rows[b C + c]][k] += g[b C + c][k],
// where several channels of a cloud may have selected the SAME point.  ATen's index_add does this with atomic float adds (the
// order of duplicates, and with it the last bit, changes from run to run); here a cloud's C rows are one workgroup: thread c is
// the LEADER of its row if no c' < c names the same row, and a leader adds the duplicates in increasing c' and writes the sum —
// deterministic, no atomics.  dx is zeroed by the caller (rows that no channel selected stay zero).
__global__ void __launch_bounds__(1024) scatter_rows_grouped_kernel(const float* __restrict__ g, const int64_t* __restrict__ rows,
                                                                    float* __restrict__ dx, int C, int K) {
    // Duplicates are the rule, not the exception (with random weights a handful of points hold the maxima of hundreds of channels):
    // a leader walking its duplicates one after the other took 150 us.  Here every channel finds its NEXT duplicate (smallest
    // e > c with the same row) in one sweep, and the chains are summed by pointer jumping — acc[c] += acc[next[c]], next[c] =
    // next[next[c]], ten synchronous steps for 1024 channels — a fixed tree of additions whatever the timing: deterministic.
    __shared__ int srow[1024];       // row numbers relative to the group's first (a cloud's rows span less than 2^31)
    __shared__ short nxt[2][1024];
    __shared__ long long base;
    extern __shared__ float sacc[];  // [2][C * K]
    const long b = blockIdx.x;
    const int c = threadIdx.x, CK = C * K;
    const long long mine64 = c < C ? rows[b * C + c] : 0;
    if (c == 0) base = mine64;
    __syncthreads();
    const int mine = c < C ? (int)(mine64 - base) : 0x7fffffff - c;       // (padding entries: distinct from everything real)
    srow[c] = mine;
    for (int e = c; e < CK; e += 1024) sacc[e] = g[b * CK + e];
    __syncthreads();
    bool before = c >= C;
    int next = -1;
    if (c < C) {
        // (16-byte LDS reads; the row table is padded to 1024 entries with values that match nothing)
        const int c4 = c & ~3;
        for (int e = 0; e < c4; e += 4) {
            const int4 v = *reinterpret_cast<const int4*>(&srow[e]);
            before |= (v.x == mine) | (v.y == mine) | (v.z == mine) | (v.w == mine);
        }
        for (int e = c4; e < c; ++e) before |= srow[e] == mine;
        for (int e = ((C + 3) & ~3) - 4; e > c4; e -= 4) {
            const int4 v = *reinterpret_cast<const int4*>(&srow[e]);
            next = v.w == mine ? e + 3 : next;
            next = v.z == mine ? e + 2 : next;
            next = v.y == mine ? e + 1 : next;
            next = v.x == mine ? e : next;
        }
        for (int e = c4 + 3; e > c; --e) next = srow[e] == mine ? e : next;
    }
    nxt[0][c] = (short)next;
    __syncthreads();
    int cur = 0;
    for (int span = 1; span < C; span <<= 1) {       // chains are at most C long: ceil(log2 C) jumps
        const int n = nxt[cur][c];
        if (n >= 0) {
            for (int k = 0; k < K; ++k) sacc[(cur ^ 1) * CK + c * K + k] = sacc[cur * CK + c * K + k] + sacc[cur * CK + n * K + k];
            nxt[cur ^ 1][c] = nxt[cur][n];
        } else {
            if (c < C)
                for (int k = 0; k < K; ++k) sacc[(cur ^ 1) * CK + c * K + k] = sacc[cur * CK + c * K + k];
            nxt[cur ^ 1][c] = -1;       // (also the padding threads: an unwritten entry would be read as a pointer in the next step)
        }
        cur ^= 1;
        __syncthreads();
    }
    if (!before)
        for (int k = 0; k < K; ++k) dx[mine64 * K + k] = sacc[cur * CK + c * K + k];
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

// the b128 forms: every row start of every tensor involved is 16-byte addressable
static bool ln_vec4(int C, long l0, long l1, long l2, long l3) {
    return C % 4 == 0 && l0 % 4 == 0 && l1 % 4 == 0 && l2 % 4 == 0 && l3 % 4 == 0;
}
static bool ln_aligned(const void* a, const void* b, const void* c, const void* d, const void* e) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 15) == 0;   // (a null pointer is aligned)
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
    if (ln_vec4(C, ldx, ldy, ldx, ldx) && ln_aligned(x, y, rowbias, gamma, beta)) {
        if (C <= 256)
            hipLaunchKernelGGL((layernorm_fwd4_kernel<1>), dim3(ln_blocks(R)), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape,
                               gamma, beta, y, ldy, mean, rstd, R, C, eps, act);
        else
            hipLaunchKernelGGL((layernorm_fwd4_kernel<2>), dim3(ln_blocks(R)), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape,
                               gamma, beta, y, ldy, mean, rstd, R, C, eps, act);
    } else {
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ln_blocks(R)), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape,
                           gamma, beta, y, ldy, mean, rstd, R, C, eps, act);
    }
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
    if (ln_vec4(C, ldx, act == SG_ACT_RELU ? ldy : 4, lddy, lddz) && ln_aligned(x, act == SG_ACT_RELU ? y : x, rowbias, gamma, dy) &&
        ((uintptr_t)dz & 15) == 0) {
        if (C <= 256)
            hipLaunchKernelGGL((layernorm_bwd4_kernel<1>), dim3(nb), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape, gamma, y,
                               ldy, dy, lddy, mean, rstd, dz, lddz, part, R, C, act);
        else
            hipLaunchKernelGGL((layernorm_bwd4_kernel<2>), dim3(nb), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape, gamma, y,
                               ldy, dy, lddy, mean, rstd, dz, lddz, part, R, C, act);
    } else {
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, stream, x, ldx, rowbias, rows_per_shape, gamma, y, ldy,
                           dy, lddy, mean, rstd, dz, lddz, part, R, C, act);
    }
    // part is [2][nb][C]: both sums in one launch (blockIdx.y = which)
    hipLaunchKernelGGL(colsum_small_kernel, dim3(sg_cdiv(C, 64), 2), dim3(1024), 0, stream, part, dgamma, dbeta, nb, C);
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
    if (cols % 4 == 0 && ld % 4 == 0 && batch_stride % 4 == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(colsum_tall4_kernel, dim3(nb, (unsigned)batch), dim3(256), 0, stream, x, (float*)workspace, rows, cols,
                           ld, batch_stride);
    else
        hipLaunchKernelGGL(colsum_tall_kernel, dim3(nb, (unsigned)batch), dim3(256), 0, stream, x, (float*)workspace, rows, cols,
                           ld, batch_stride);
    hipLaunchKernelGGL(colsum_small_kernel, dim3(sg_cdiv(cols, 64), (unsigned)batch), dim3(1024), 0, stream,
                       (const float*)workspace, out, (float*)nullptr, nb, cols);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// point chunks of the two-stage form: enough of them for >= 1024 workgroups, at least 64 points each, at most 64
static int segmax_chunks(long B, long P, int C) {
    if (C % 4 != 0) return 1;
    const long base = B * sg_cdiv(C, 256);
    long n = (1024 + base - 1) / base;
    if (n > P / 64) n = P / 64;
    if (n > 64) n = 64;
    return n < 1 ? 1 : (int)n;
}
size_t sg_segmax_workspace_bytes(long B, long P, int C) {
    const int n = segmax_chunks(B, P, C);
    return n > 1 ? (size_t)B * n * C * 8 : 0;
}

int sg_segmax_fwd(const float* x, float* out, int* idx, long B, long P, int C, void* workspace, size_t workspace_bytes,
                  hipStream_t stream) {
    SG_CHECK_ARG(x && out && idx && B > 0 && B <= 65535 && P > 0 && P < 0x7fffffff && C > 0);
    const int nchunk = segmax_chunks(B, P, C);
    if (C % 4 != 0 || ((uintptr_t)x & 15) != 0 || ((uintptr_t)out & 15) != 0 || ((uintptr_t)idx & 15) != 0) {
        hipLaunchKernelGGL(segmax_fwd_kernel, dim3(sg_cdiv(C, 64), (unsigned)B), dim3(256), 0, stream, x, out, idx, P, C);
    } else if (nchunk == 1) {
        hipLaunchKernelGGL(segmax_part_kernel, dim3(sg_cdiv(C, 256), 1, (unsigned)B), dim3(256), 0, stream, x, out, idx, P, C, 1);
    } else {
        if (!workspace || workspace_bytes < sg_segmax_workspace_bytes(B, P, C) || ((uintptr_t)workspace & 15) != 0)
            SG_FAIL(SG_ERR_WORKSPACE, "sg_segmax_fwd: workspace too small or not 16-byte aligned");
        float* pv = (float*)workspace;
        int* pi = (int*)(pv + (size_t)B * nchunk * C);
        hipLaunchKernelGGL(segmax_part_kernel, dim3(sg_cdiv(C, 256), nchunk, (unsigned)B), dim3(256), 0, stream, x, pv, pi, P, C,
                           nchunk);
        hipLaunchKernelGGL(segmax_merge_kernel, dim3(sg_cdiv(C, 64), (unsigned)B), dim3(256), 0, stream, pv, pi, out, idx, nchunk,
                           C);
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// the second level of sg_segmax_fwd on its own: out / idx [B][C] from nchunk partial (value, point) pairs per cloud and channel,
// pv / pi [(b * nchunk + k) * C + c] (internal: the tiles of sg_pointnet_select)
int sg_segmax_merge_partials(const float* pv, const int* pi, float* out, int* idx, long B, int nchunk, int C, hipStream_t stream) {
    SG_CHECK_ARG(pv && pi && out && idx && B > 0 && nchunk > 0 && C > 0);
    hipLaunchKernelGGL(segmax_merge_kernel, dim3(sg_cdiv(C, 64), (unsigned)B), dim3(256), 0, stream, pv, pi, out, idx, nchunk, C);
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

// Deterministic adjoint of a grouped row gather (see scatter_rows_grouped_kernel): g [B*C][K], rows [B*C] (int64 row numbers into
// dx [N][K]; the rows of group b may repeat, rows of different groups are distinct), dx zeroed by the caller.  C <= 1024, C K <= 8192.
int sg_scatter_rows_grouped(const float* g, const int64_t* rows, float* dx, long B, int C, int K, hipStream_t stream) {
    SG_CHECK_ARG(g && rows && dx && B > 0 && C > 0 && C <= 1024 && K > 0 && (long)C * K <= 8192);
    hipLaunchKernelGGL(scatter_rows_grouped_kernel, dim3((unsigned)B), dim3(1024), (size_t)2 * C * K * sizeof(float), stream, g, rows, dx, C, K);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// The diagonal last layer of the selected-points pass (see rowdot_kernel): h [B*C][K], w [C][K], bias [C] or NULL, g / out [B*C].
int sg_rowdot(const float* h, const float* w, const float* bias, float* out, long B, int C, int K, hipStream_t stream) {
    SG_CHECK_ARG(h && w && out && B > 0 && C > 0 && K > 0);
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((B * C + 3) / 4)), dim3(256), 0, stream, h, w, bias, out, B * C, C, K);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_rowscale(const float* g, const float* w, float* out, long B, int C, int K, hipStream_t stream) {
    SG_CHECK_ARG(g && w && out && B > 0 && C > 0 && K > 0);
    long blocks = (B * C * K + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, w, out, B * C, C, K);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_rowouter(const float* g, const float* h, float* out, long B, int C, int K, hipStream_t stream) {
    SG_CHECK_ARG(g && h && out && B > 0 && C > 0 && K > 0);
    hipLaunchKernelGGL(rowouter_kernel, dim3((unsigned)(((long)C * K + 255) / 256)), dim3(256), 0, stream, g, h, out, B, C, K);
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
