// shapegan_amd/csrc/gemm.hip — strided f32-MFMA GEMM (K3 "1^3<->4^3 convs as GEMMs", K6 nn.Linear).
//
// Replaces ATen addmm/mm behind nn.Linear (model/autoencoder.py:34,41-42,45; model/progressive_gan.py:28,30),
// the k4/s1 convolutions on 1^3 / 4^3 grids that are plain GEMMs (model/gan.py:9,55; model/autoencoder.py:28,51)
// and the SDFNet weight-gradient GEMMs (model/sdf_net.py:26-53 under autograd).
//
//   C(i,j) = act( sum_k A(i,k) * B(k,j) + bias_i[i] + bias_j[j >> bias_j_shift] )
//
// A, B, C are addressed through element strides; each operand must have one unit stride (either
// along its row or along k) so that staging loads coalesce.  j is the lane axis of the MFMA C/D
// fragment: pick j as the contiguous axis of C.
#include "mfma_tile.h"
#include "../../include/shapegan_hip.h"

namespace sg {

struct GemmEpi {
    float* c;
    long sci, scj;
    const float* bias_i;
    const float* bias_j;
    int bias_j_shift;
    int act;
    float slope;
    struct Col {
        long off;
        float bj;
    };
    __device__ Col col(int j) const { return Col{(long)j * scj, bias_j ? bias_j[j >> bias_j_shift] : 0.f}; }
    __device__ void store(const Col& cc, int i, int j, float v) const {
        v += cc.bj;
        if (bias_i) v += bias_i[i];
        c[cc.off + (long)i * sci] = sg_apply_act(v, act, slope);
    }
};

// sum over rows: out[j] = sum_i x[i*ld + j]   (bias gradients of Linear / 1^3 convs)
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                     int cols, long ld) {
    // one workgroup per strip of 64 columns: 4 waves take interleaved rows (a single thread walking every row of its
    // column is a chain of `rows` dependent loads: 23 us for a 128 x 256 bias-gradient matrix), LDS-reduced at the end
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (j < cols) {
        int i = wave;
        for (; i + 4 < rows; i += 8) {
            s0 += x[(long)i * ld + j];
            s1 += x[(long)(i + 4) * ld + j];
        }
        if (i < rows) s0 += x[(long)i * ld + j];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && j < cols) out[j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// out[r] = sum_{e<len} x[r*ld + e]: one 256-thread workgroup per row (float4 stream, wave shuffles, LDS across waves).
// Bias gradients of the SDFNet layers are rows of 20 000 - 4 000 000 points: one wave per row was 30 % of the
// auto-decoder step.
__global__ void __launch_bounds__(256) rowsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows,
                                                     long len, long ld) {
    const long r = blockIdx.x;
    const float* p = x + r * ld;
    float s = 0.f;
    if ((((uintptr_t)p) & 15) == 0 && (len & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const long n4 = len >> 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (long e = threadIdx.x; e < n4; e += 256) {
            const float4 v = p4[e];
            s0 += v.x;
            s1 += v.y;
            s2 += v.z;
            s3 += v.w;
        }
        s = (s0 + s1) + (s2 + s3);
    } else {
        for (long e = threadIdx.x; e < len; e += 256) s += p[e];
    }
    s = sg_wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[r] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[r*S + s] = sum_{e in [off[s], off[s+1])} x[r*ld + e]: one wave per (row, segment) pair
__global__ void __launch_bounds__(256) segsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows,
                                                     long ld, const int64_t* __restrict__ off, long S) {
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= rows * S) return;
    const long r = pair / S, sg = pair - r * S;
    const long beg = off[sg], end = off[sg + 1];
    const int lane = threadIdx.x & 63;
    const float* p = x + r * ld;
    float acc = 0.f;
    for (long e = beg + lane; e < end; e += 64) acc += p[e];
    acc = sg_wave_sum(acc);
    if (lane == 0) out[pair] = acc;
}

}  // namespace sg

using namespace sg;

extern "C" {

// up to 128 split-K partials: a 256x256 weight-gradient GEMM over 4M points still reaches 512 workgroups of 128x128 tiles
size_t sg_gemm_workspace_bytes(int M, int N) { return (size_t)128 * M * N * sizeof(float); }

int sg_gemm(const float* A, long sai, long sak, const float* B, long sbk, long sbj, float* C, long sci, long scj,
            const float* bias_i, const float* bias_j, int bias_j_shift, int M, int N, int K, int act, float slope,
            void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
    SG_CHECK_ARG(sai == 1 || sak == 1);
    SG_CHECK_ARG(sbk == 1 || sbj == 1);
    GemmEpi epi{C, sci, scj, bias_i, bias_j, bias_j_shift, act, slope};
    float* ws = (float*)workspace;
    // prefer the k-contiguous form when a dimension is degenerate (both strides legal)
    const bool a_kfast = (sak == 1);
    const bool b_kfast = (sbk == 1);
    if (a_kfast && b_kfast) {
        launch_tile_gemm(MatRowMajor{A, sai, 0}, MatRowMajor{B, sbj, 0}, epi, M, N, K, ws, workspace_bytes, stream);
    } else if (a_kfast && !b_kfast) {
        launch_tile_gemm(MatRowMajor{A, sai, 0}, MatColMajor{B, sbk, 0}, epi, M, N, K, ws, workspace_bytes, stream);
    } else if (!a_kfast && b_kfast) {
        launch_tile_gemm(MatColMajor{A, sak, 0}, MatRowMajor{B, sbj, 0}, epi, M, N, K, ws, workspace_bytes, stream);
    } else {
        launch_tile_gemm(MatColMajor{A, sak, 0}, MatColMajor{B, sbk, 0}, epi, M, N, K, ws, workspace_bytes, stream);
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_colsum(const float* x, float* out, int rows, int cols, long ld, hipStream_t stream) {
    SG_CHECK_ARG(x && out && rows > 0 && cols > 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(sg_cdiv(cols, 64)), dim3(256), 0, stream, x, out, rows, cols, ld);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_rowsum(const float* x, float* out, long rows, long len, long ld, hipStream_t stream) {
    SG_CHECK_ARG(x && out && rows > 0 && len > 0);
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, out, rows, len, ld);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_segsum(const float* x, float* out, long rows, long ld, const int64_t* seg_off, long nseg, hipStream_t stream) {
    SG_CHECK_ARG(x && out && seg_off && rows > 0 && nseg > 0);
    hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)((rows * nseg + 3) / 4)), dim3(256), 0, stream, x, out, rows, ld,
                       seg_off, nseg);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"
