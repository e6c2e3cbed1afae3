// shapegan_amd/csrc/head.hip — the critic's last layer as two streaming kernels (gfx950 only).
//
// gan.Discriminator ends in  LeakyReLU -> Conv3d(256 -> 1, kernel 4, stride 1) on a 4^3 grid  (model/gan.py:54-55): per sample
// one dot product over K = 256 * 64 values.  As a GEMM with one output column it ran on the generic MFMA skeleton with a
// split-K finalize; its backward was two more GEMMs, a column sum, the activation backward of the layer below and that
// layer's bias-gradient row / column sums: 9 launches of 5 - 15 us around 16 MB of traffic per critic update (0.29 ms of the
// 17.4 ms WGAN step, profiles/r03_wgan_step_timeline.txt).  Here:
//
//   forward   y[n]      = bias + sum_k act(z[n, k]) * w[k]                       one workgroup per sample
//   backward  gz[n, k]  = gy[n] * w[k] * act'(z[n, k])                           one workgroup per channel c (k = c * S + s):
//             gw[k]     = sum_n gy[n] * act(z[n, k])                             every z element is read once, every gz element
//             gbz[c]    = sum_{n, s} gz[n, c * S + s]                            written once; all sums in a fixed order
//             gb        = sum_n gy[n]
//
// z is the PRE-activation of the layer below (its LeakyReLU is applied on load), so that layer's activation backward and bias
// gradient come out of the same pass.  HBM-bound: 4 B read per element forward, 4 B read + 4 B written backward.
#include "common.h"
#include "../../include/shapegan_hip.h"

namespace sg {

__device__ __forceinline__ float head_act(float v, int act, float slope) {
    return act == SG_ACT_LEAKY ? (v > 0.f ? v : v * slope) : (act == SG_ACT_RELU ? (v > 0.f ? v : 0.f) : v);
}
__device__ __forceinline__ float head_dact(float v, int act, float slope) {   // derivative, read off the pre-activation
    return act == SG_ACT_LEAKY ? (v > 0.f ? 1.f : slope) : (act == SG_ACT_RELU ? (v > 0.f ? 1.f : 0.f) : 1.f);
}

// y[n] = bias + sum_k act(z[n][k]) w[k];  K % 4 == 0.  256 threads, up to 8 b128 loads of z in flight per lane.
__global__ void __launch_bounds__(256) head_dot_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, long K, int act,
                                                           float slope) {
    const long n = blockIdx.x;
    const f32x4* z4 = reinterpret_cast<const f32x4*>(z + n * K);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    const long k4 = K >> 2;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    long e = threadIdx.x;
    for (; e + 7 * 256 < k4; e += 8 * 256) {
        f32x4 zv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            zv[u] = __builtin_nontemporal_load(z4 + e + 256 * u);
            wv[u] = w4[e + 256 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = fmaf(head_act(zv[u][j], act, slope), wv[u][j], s[j]);
    }
    for (; e < k4; e += 256) {
        const f32x4 zv = z4[e], wv = w4[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = fmaf(head_act(zv[j], act, slope), wv[j], s[j]);
    }
    float t = sg_wave_sum((s[0] + s[1]) + (s[2] + s[3]));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) y[n] = ((red[0] + red[1]) + (red[2] + red[3])) + (bias ? bias[0] : 0.f);
}

// One workgroup per channel c: 512 threads = 16 float4 columns (S = 64 positions of the channel) x 32 sample groups.
// thread (col, grp) walks the samples n = grp, grp + 32, ...  All of a thread's loads are independent of each other; four are
// requested at a time.  (256 workgroups = one per CU at the critic's 256 channels: eight waves per CU instead of four keep twice
// the loads in flight — the kernel streams 25 MB in ~11 us, bound by latency, not bandwidth.)
constexpr int kHeadS = 64;
constexpr int kHeadG = 32;     // sample groups
__global__ void __launch_bounds__(512) head_dot_bwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                           const float* __restrict__ gy, float* __restrict__ gz,
                                                           float* __restrict__ gw, float* __restrict__ gb, float* __restrict__ gbz,
                                                           float4* __restrict__ ap, int N, long K, int act, float slope) {
    const int c = blockIdx.x;
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long k0 = (long)c * kHeadS + 4 * col;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k0);
    f32x4 aw = {0.f, 0.f, 0.f, 0.f};   // this thread's part of gw[k0 .. k0+3]
    float az = 0.f;                    // this thread's part of gbz[c]
    // ap (optional): gz in the A-fragment order of conv_wgrad_halo4_kernel (pack_wgrad_dy4_kernel's layout with nslice = N): the
    // 8-float chunk (position group gq = col / 2) of channel c is split into its even columns (fragment lane r = c % 32) and its odd
    // columns (lane r + 32); the two threads that hold the chunk's halves exchange them (adjacent lanes) and write one float4 each
    auto pack = [&](const f32x4& o, int nn) __attribute__((always_inline)) {
        f32x4 p;
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = __shfl_xor(o[j], 1, 64);
        const f32x4 v = (col & 1) ? f32x4{p[1], p[3], o[1], o[3]} : f32x4{o[0], o[2], p[0], p[2]};
        const long e = ((((long)(c >> 5) * N + nn) * 8 + (col >> 1)) * 64) + (col & 1) * 32 + (c & 31);
        ap[e] = make_float4(v[0], v[1], v[2], v[3]);
    };
    int n = grp;
    for (; n + 3 * kHeadG < N; n += 4 * kHeadG) {
        f32x4 zv[4];
        float g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            zv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + (long)(n + kHeadG * u) * K + k0));
            g[u] = gy[n + kHeadG * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                aw[j] = fmaf(g[u], head_act(zv[u][j], act, slope), aw[j]);
                o[j] = g[u] * wv[j] * head_dact(zv[u][j], act, slope);
            }
            az += (o[0] + o[1]) + (o[2] + o[3]);
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(gz + (long)(n + kHeadG * u) * K + k0));
            if (ap) pack(o, n + kHeadG * u);
        }
    }
    for (; n < N; n += kHeadG) {
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (long)n * K + k0);
        const float g = gy[n];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            aw[j] = fmaf(g, head_act(zv[j], act, slope), aw[j]);
            o[j] = g * wv[j] * head_dact(zv[j], act, slope);
        }
        az += (o[0] + o[1]) + (o[2] + o[3]);
        *reinterpret_cast<f32x4*>(gz + (long)n * K + k0) = o;
        if (ap) pack(o, n);
    }
    // fixed-order sums over the sample groups (gw) and over everything (gbz)
    __shared__ f32x4 redw[kHeadG][16];
    __shared__ float redz[16 * kHeadG];
    redw[grp][col] = aw;
    redz[threadIdx.x] = az;
    __syncthreads();
    if (grp == 0) {
        f32x4 t = redw[0][col];
#pragma unroll
        for (int g2 = 1; g2 < kHeadG; ++g2) t += redw[g2][col];
        if (gw) *reinterpret_cast<f32x4*>(gw + k0) = t;
    }
    if (threadIdx.x < 64) {
        float t = ((redz[threadIdx.x] + redz[threadIdx.x + 64]) + (redz[threadIdx.x + 128] + redz[threadIdx.x + 192])) +
                  ((redz[threadIdx.x + 256] + redz[threadIdx.x + 320]) + (redz[threadIdx.x + 384] + redz[threadIdx.x + 448]));
        t = sg_wave_sum(t);
        if (threadIdx.x == 0 && gbz) gbz[c] = t;
    }
    if (c == 0 && gb && threadIdx.x >= 64 && threadIdx.x < 128) {     // gb = sum_n gy[n] (one wave of workgroup 0)
        float t = 0.f;
        for (int i = threadIdx.x - 64; i < N; i += 64) t += gy[i];
        t = sg_wave_sum(t);
        if (threadIdx.x == 64) gb[0] = t;
    }
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_head_dot_fwd(const float* z, const float* w, const float* bias, float* y, int N, long K, int act, float slope,
                    hipStream_t stream) {
    SG_CHECK_ARG(z && w && y && N > 0 && K > 0 && K % 4 == 0);
    SG_CHECK_ARG(act == SG_ACT_NONE || act == SG_ACT_LEAKY || act == SG_ACT_RELU);
    SG_CHECK_ARG(((uintptr_t)z & 15) == 0 && ((uintptr_t)w & 15) == 0);
    hipLaunchKernelGGL(head_dot_fwd_kernel, dim3((unsigned)N), dim3(256), 0, stream, z, w, bias, y, K, act, slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_head_dot_bwd(const float* z, const float* w, const float* gy, float* gz, float* gw, float* gb, float* gbz, void* gz_image,
                    int N, int C, int S, int act, float slope, hipStream_t stream) {
    SG_CHECK_ARG(z && w && gy && gz && N > 0 && C > 0);      // gw / gb / gbz: optional outputs
    SG_CHECK_ARG(S == kHeadS);     // a 4^3 grid per channel (model/gan.py:55); other shapes take the GEMM path
    SG_CHECK_ARG(act == SG_ACT_NONE || act == SG_ACT_LEAKY || act == SG_ACT_RELU);
    SG_CHECK_ARG(((uintptr_t)z & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)gz & 15) == 0 && ((uintptr_t)gw & 15) == 0);
    SG_CHECK_ARG(!gz_image || C % 128 == 0);     // the image has no padded row tiles
    hipLaunchKernelGGL(head_dot_bwd_kernel, dim3((unsigned)C), dim3(512), 0, stream, z, w, gy, gz, gw, gb, gbz, (float4*)gz_image, N,
                       (long)C * S, act, slope);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"
