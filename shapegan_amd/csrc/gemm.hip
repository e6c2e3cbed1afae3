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
#include <stdlib.h>

#include "mfma_tile.h"
#ifndef SG_GEMM128
#define SG_GEMM128 1   // sg_gemm's large products: 1 = gemm128 (persistent where eligible), 2 = one workgroup per tile, 0 = the skeleton (A/B builds)
#endif
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
// auto-decoder step.  Four b128 loads are requested before the first add (one load per loop trip left a lane with a single
// request in flight: 51 us per call at the progressive discriminator's 32 768-element rows).
__device__ __forceinline__ float rowsum_lane_part(const float* __restrict__ p, long len, int t, int nt) {
    float s = 0.f;
    if ((((uintptr_t)p) & 15) == 0 && (len & 3) == 0) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
        const long n4 = len >> 2;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        long e = t;
        for (; e + 3 * nt < n4; e += 4 * nt) {
            const f32x4 v0 = __builtin_nontemporal_load(p4 + e), v1 = __builtin_nontemporal_load(p4 + e + nt);
            const f32x4 v2 = __builtin_nontemporal_load(p4 + e + 2 * nt), v3 = __builtin_nontemporal_load(p4 + e + 3 * nt);
            a0 += v0;
            a1 += v1;
            a2 += v2;
            a3 += v3;
        }
        for (; e < n4; e += nt) a0 += p4[e];
        const f32x4 a = (a0 + a1) + (a2 + a3);
        s = (a[0] + a[1]) + (a[2] + a[3]);
    } else {
        for (long e = t; e < len; e += nt) s += p[e];
    }
    return s;
}
__global__ void __launch_bounds__(256) rowsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows,
                                                     long len, long ld) {
    const long r = blockIdx.x;
    float s = rowsum_lane_part(x + r * ld, len, threadIdx.x, 256);
    s = sg_wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[r] = (red[0] + red[1]) + (red[2] + red[3]);
}
// short rows (conv bias gradients on 4^3 .. 16^3 grids: 64 .. 4096 elements): one WAVE per row, four rows per workgroup
__global__ void __launch_bounds__(256) rowsum_wave_kernel(const float* __restrict__ x, float* __restrict__ out, long rows,
                                                          long len, long ld) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float s = sg_wave_sum(rowsum_lane_part(x + r * ld, len, threadIdx.x & 63, 64));
    if ((threadIdx.x & 63) == 0) out[r] = s;
}

// The same with the rows split into up to 8 equal groups that go to different destinations (the seven bias gradients of
// an SDFNet backward land in seven separate slices of the optimizer's flat gradient buffer).
struct RowsumDst {
    float* out[8];
    long stride[8];
};
__global__ void __launch_bounds__(256) rowsum_multi_kernel(const float* __restrict__ x, RowsumDst dst, long rows_per_dst,
                                                           long len, long ld) {
    const long r = blockIdx.x;
    float s = rowsum_lane_part(x + r * ld, len, threadIdx.x, 256);
    s = sg_wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dst.out[r / rows_per_dst][(r % rows_per_dst) * dst.stride[r / rows_per_dst]] = (red[0] + red[1]) + (red[2] + red[3]);
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




// ================================================================================================================
// C[M,N] = A[M,K] * B[N,K]^T with K contiguous in both operands and K >> M, N: the SDFNet weight gradients
// dW_l = dZ_l * H_{l-1}^T over the points of a batch (model/sdf_net.py:26-52 backward; 256 x 256 x 20 000 ... 4 M).
// LDS-staged like the conv halo kernels and free of vector address arithmetic in the loop:
//   * a workgroup owns a 128 x 128 tile of C and a K range (split-K, deterministic partials + finalize);
//   * a stage is 32 k: each thread moves four 16-byte pieces of A and four of B (8 lanes cover one 128-byte row segment)
//     with buffer loads — row in the scalar offset, rows beyond M / N get an out-of-range offset and read as 0 — and
//     writes them to LDS as [k / 4][row][4]: the piece IS the unit, because the MFMA k index is only a summation index,
//     lane (row, kh) can take k = 8 g + 4 kh + j for step j of group g in BOTH operands.  A fragment read is then one
//     ds_read_b128 per 4 MFMA steps, contiguous over the 32 rows of a half-wave;
//   * loads run two stages ahead (registers), LDS stores one stage ahead, one barrier per stage; the chunk stride is
//     padded by 16 bytes so that the 8 pieces of a row segment land in different banks.
constexpr int kNtKC = 32;                 // k per stage
constexpr int kNtChunk = 128 * 4 + 4;     // floats per (k/4) chunk of one operand tile
constexpr int kNtOp = 8 * kNtChunk;       // floats per operand tile per buffer

constexpr int kNtMaxBatch = 8;
struct GemmNtArgs {
    const float* A;
    const float* B;
    float* out;      // partials [batch][nsplit][M][N] (ldc = N) or C itself when nsplit == 1 and batch == 1
    long lda, ldb, ldc;
    int M, N;
    long K, kchunk;  // kchunk: multiple of kNtKC
    int nsplit;      // groups = batch * nsplit
    int ngroups;
    long a_off[kNtMaxBatch], b_off[kNtMaxBatch];   // element offsets of the batch members' operands
    // LNRELU: B holds LayerNorm-normalised rows xhat and the product is taken with relu(gamma[row] * xhat + beta[row]) — the
    // weight gradients of the LayerNorm MLP (model/point_sdf_net.py:104-116) read the activation images the fused backward needs
    // (xhat) and rebuild the layer outputs on the way into LDS, two VALU operations per element
    const float* gamma;
    const float* beta;
    long g_off[kNtMaxBatch];                       // element offset of the member's gamma / beta vectors
};

template <bool LNRELU>
__global__ void __launch_bounds__(256) gemm_nt_bigk_kernel(GemmNtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][A | B][8][kNtChunk]
    lds_float* const sl = (lds_float*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, kh = lane >> 5;
    // XCD-aware decode of a 1-D grid.  The tiles of one (batch member, K split) group read the same operand slices — every
    // 128-row slice of A serves the tiles of a tile row, every slice of B those of a tile column — and a 3-D grid spread them over
    // different XCDs (workgroup b runs on XCD b % 8): each XCD's L2 then fetched its own copy, 6.44 GB per launch against 3.22 GB
    // of operands at 262 144 points (profiles/r03_configs_hbm_traffic.json), HBM co-limiting a kernel the matrix pipe should
    // bound.  Here the tiles of a group are consecutive workgroups of ONE XCD, dispatched together: one fetch, the rest L2 hits.
    const int tiles_n = (a.N + 127) >> 7, tiles = tiles_n * ((a.M + 127) >> 7);
    const int xcd = blockIdx.x & 7, qx = blockIdx.x >> 3;
    const int grp = (qx / tiles) * 8 + xcd, tile = qx % tiles;          // group = batch member * nsplit + K split
    if (grp >= a.ngroups) return;
    const int i0 = (tile / tiles_n) * 128, j0 = (tile % tiles_n) * 128;
    const int bz = grp / a.nsplit, sz = grp - bz * a.nsplit;   // batch member, K split
    const long kbeg = (long)sz * a.kchunk, kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const int nstage = (int)((kend - kbeg + kNtKC - 1) / kNtKC);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // copy: thread -> (row inside a 32-row pass, 16-byte piece kq of the 128-byte segment)
    const int rowl = tid >> 3, kq = tid & 7;
    // The 16-byte pieces of the last stage overshoot K (their tail is zeroed in registers, mask_tail): inside the matrix that lands
    // in the next row, but behind the LAST row of an operand it is behind the operand — and behind the allocation when the operand
    // ends it (dZ7 ends `dz`).  An unmapped page there hangs the wave (seen once the caching allocator placed `dz` at the end of a
    // segment), so the resources end exactly at the operand's last element (extents below 2 GiB; larger ones keep the window).
    const __amdgpu_buffer_rsrc_t ares = make_rsrc_bytes(a.A + a.a_off[bz] + (long)i0 * a.lda + kbeg,
                                                        ((long)(a.M - 1 - i0) * a.lda + (a.K - kbeg)) * 4);
    const __amdgpu_buffer_rsrc_t bres = make_rsrc_bytes(a.B + a.b_off[bz] + (long)j0 * a.ldb + kbeg,
                                                        ((long)(a.N - 1 - j0) * a.ldb + (a.K - kbeg)) * 4);
    unsigned avoff[4], bvoff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 32 + rowl;   // the 32-row pass offset goes into the scalar offset (4 M-point rows: > 2^31 bytes)
        avoff[p] = i0 + row < a.M ? (unsigned)(((long)rowl * a.lda + 4 * kq) * 4) : kBufOutside;
        bvoff[p] = j0 + row < a.N ? (unsigned)(((long)rowl * a.ldb + 4 * kq) * 4) : kBufOutside;
        pin_vgpr(avoff[p]);
        pin_vgpr(bvoff[p]);
    }
    const unsigned pass_a = (unsigned)(32 * a.lda * 4), pass_b = (unsigned)(32 * a.ldb * 4);
    lds_float* sdst = sl + kq * kNtChunk + rowl * 4;
    pin_vgpr(sdst);
    const lds_float* afrag = sl + kh * kNtChunk + (wm * 64 + r) * 4;
    const lds_float* bfrag = sl + kNtOp + kh * kNtChunk + (wn * 64 + r) * 4;
    pin_vgpr(afrag);
    pin_vgpr(bfrag);

    float gr[4], br[4];   // LNRELU: gamma / beta of this thread's four B rows
    if constexpr (LNRELU) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = j0 + p * 32 + rowl;
            gr[p] = row < a.N ? a.gamma[a.g_off[bz] + row] : 0.f;
            br[p] = row < a.N ? a.beta[a.g_off[bz] + row] : 0.f;
        }
    }
    f32x4 ra[4], rb[4];   // pieces in flight (two stages ahead)
    auto issue = [&](int s) __attribute__((always_inline)) {   // loads of stage s (clamped: a stage past the end re-reads the last)
        const int sc = s < nstage ? s : nstage - 1;
        const unsigned so = (unsigned)sc * (kNtKC * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = buf_load4v(ares, avoff[p], so + (unsigned)p * pass_a);
            rb[p] = buf_load4v(bres, bvoff[p], so + (unsigned)p * pass_b);
        }
    };
    auto lnrelu = [&]() __attribute__((always_inline)) {   // the B pieces in registers: xhat -> relu(gamma xhat + beta)
        if constexpr (LNRELU) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) rb[p][j] = fmaxf(fmaf(gr[p], rb[p][j], br[p]), 0.f);
        }
    };
    auto mask_tail = [&](int s) __attribute__((always_inline)) {   // zero the k >= kend part of the last stage
        const long k = kbeg + (long)s * kNtKC + 4 * kq;
        if (k + 4 > kend) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j >= kend) ra[p][j] = rb[p][j] = 0.f;
            }
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *(lds_f32x4*)(sdst + buf * 2 * kNtOp + p * 128) = ra[p];
            *(lds_f32x4*)(sdst + buf * 2 * kNtOp + kNtOp + p * 128) = rb[p];
        }
    };

    if (nstage > 0) {
        issue(0);
        lnrelu();
        if (nstage == 1) mask_tail(0);
        commit(0);
        issue(1);
        __syncthreads();
        auto stage = [&](auto tag, int s) __attribute__((always_inline)) {
            constexpr int CUR = decltype(tag)::value, NXT = CUR ^ 1;
            lnrelu();
            if (s + 1 == nstage - 1) mask_tail(s + 1);   // the pieces in registers belong to stage s + 1
            commit(NXT);
            issue(s + 2);
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *(const lds_f32x4*)(afrag + CUR * 2 * kNtOp + t * 128);
                fb[t] = *(const lds_f32x4*)(bfrag + CUR * 2 * kNtOp + t * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 ca[2], cb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ca[t] = fa[t];
                    cb[t] = fb[t];
                }
                if (g + 1 < 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        fa[t] = *(const lds_f32x4*)(afrag + CUR * 2 * kNtOp + (g + 1) * 2 * kNtChunk + t * 128);
                        fb[t] = *(const lds_f32x4*)(bfrag + CUR * 2 * kNtOp + (g + 1) * 2 * kNtChunk + t * 128);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a0 = ca[0][j], a1 = ca[1][j], b0 = cb[0][j], b1 = cb[1][j];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        for (int s = 0; s + 1 < nstage; s += 2) {
            stage(IntTag<0>(), s);
            stage(IntTag<1>(), s + 1);
        }
        if (nstage & 1) stage(IntTag<0>(), nstage - 1);
    }

    float* out = a.out + (long)grp * a.M * a.ldc;   // partial image (batch member, split); C itself if there is one
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int j = j0 + wn * 64 + tj * 32 + r;
            if (j >= a.N) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = i0 + wm * 64 + ti * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (i < a.M) out[(long)i * a.ldc + j] = acc[ti][tj][q];
            }
        }
}

struct CopyEpi {   // finalize target: C[i][j], row stride ldc
    float* c;
    long ldc;
    struct Col {
        int j;
    };
    __device__ Col col(int j) const { return Col{j}; }
    __device__ void store(const Col& cc, int i, int j, float v) const { c[(long)i * ldc + cc.j] = v; }
};

// C_b[i][j] = sum over the nsplit partials of batch member b; the members' outputs are given as element offsets from `c`
struct GemmNtOut {
    long c_off[kNtMaxBatch];
    long ldc[kNtMaxBatch];
};
__global__ void __launch_bounds__(256) gemm_nt_finalize_kernel(const float* __restrict__ ws, float* __restrict__ c, GemmNtOut o,
                                                               int M, int N, int nsplit) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const long total = (long)M * N, e = (long)blockIdx.x * 64 + lane;
    const float* p = ws + (long)b * nsplit * total;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int s = wave;
        for (; s + 4 < nsplit; s += 8) {
            s0 += p[(long)s * total + e];
            s1 += p[(long)(s + 4) * total + e];
        }
        if (s < nsplit) s0 += p[(long)s * total + e];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && e < total) {
        const int i = (int)(e / N), j = (int)(e - (long)i * N);
        c[o.c_off[b] + (long)i * o.ldc[b] + j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    }
}

static void gemm_nt_plan(int M, int N, long K, int& nsplit, long& kchunk, int batch = 1) {
    const long tiles = (long)sg_cdiv(M, 128) * sg_cdiv(N, 128) * batch;
    long s = 512 / tiles > 0 ? 512 / tiles : 1;   // at most 512 workgroups: one round at 2 per CU, no straggler round
    const long maxs = K / (8 * kNtKC) > 0 ? K / (8 * kNtKC) : 1;   // >= 8 stages per workgroup
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    kchunk = ((K + s - 1) / s + kNtKC - 1) / kNtKC * kNtKC;
    nsplit = (int)((K + kchunk - 1) / kchunk);
}

// ================================================================================================================
// gemm128_kernel (round 5): the data movement of gemm_nt_bigk_kernel for sg_gemm's large dense products — the Linear layers of the
// PointNet GAN family run at 196 608 x 256 x 256 (model/point_sdf_net.py:50-119: forward x W^T, input gradient g W, weight gradient
// g^T x), where the generic 16-k skeleton of mfma_tile.h (4-byte staging, two barriers and a fragment-read phase per k-tile) sat at
// 0.63 - 0.68 of the matrix peak and took 69 % of the family's kernel time (profiles/r05_point_gan_kernel_stats.csv).
//   C(i, j) = epilogue( sum_k A(i, k) B(k, j) )       128 x 128 tile per workgroup, 32 k per stage, 16-byte loads, one barrier per stage
// Each operand is either k-major (element (row, k) at p[row * ld + k]: the 16-byte piece is 4 consecutive k of one row and goes to LDS
// as it is, [k / 4][row][4]) or row-major (element at p[k * ld + row]: a thread loads the pieces of 4 consecutive k for the same 4
// rows — a wave instruction covers 512 contiguous bytes of one k — transposes the 4 x 4 block in registers and writes the same LDS
// layout).  Fragment reads, MFMA loop and the XCD-aware tile order are those of gemm_nt_bigk_kernel.  Requirements of the fast path
// (sg_gemm checks them and keeps the skeleton otherwise): 16-byte aligned bases and leading dimensions; for a row-major operand its
// row count is a multiple of 4.
struct Gemm128Args {
    const float* A;
    const float* B;
    long lda, ldb;
    int M, N;
    long K, kchunk;
    int nsplit;
};

template <bool AK, bool BK, class EPI>
__global__ void __launch_bounds__(256) gemm128_kernel(Gemm128Args a, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][A | B][8][kNtChunk]
    lds_float* const sl = (lds_float*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, kh = lane >> 5;
    const int tiles_n = (a.N + 127) >> 7, tiles = tiles_n * ((a.M + 127) >> 7);
    // K splits are spread over the XCDs, the tiles of one split are consecutive workgroups of one XCD (they share operand slices)
    const int xcd = blockIdx.x & 7, qx = blockIdx.x >> 3;
    const int sz = a.nsplit > 1 ? (qx / tiles) * 8 + xcd : 0;
    const int tile = a.nsplit > 1 ? qx % tiles : (int)blockIdx.x;
    if (sz >= a.nsplit || tile >= tiles) return;
    // (without a K split consecutive workgroups walk a tile COLUMN: they share the 128 x K slice of B, the small operand of a Linear)
    const int i0 = a.nsplit > 1 ? (tile / tiles_n) * 128 : (tile % ((a.M + 127) >> 7)) * 128;
    const int j0 = a.nsplit > 1 ? (tile % tiles_n) * 128 : (tile / ((a.M + 127) >> 7)) * 128;
    const long kbeg = (long)sz * a.kchunk, kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const int nstage = (int)((kend - kbeg + kNtKC - 1) / kNtKC);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // ---- copy roles ----
    // k-major operand: thread -> (row inside a 32-row pass, piece kq of the stage's 128-byte row segment), 4 passes
    // row-major operand: thread -> (k chunk c of the stage, row piece rq): loads k = 4 c .. 4 c + 3 for rows 4 rq .. 4 rq + 3
    const int rowl = tid >> 3, kq = tid & 7, cc = tid >> 5, rq = tid & 31;
    const __amdgpu_buffer_rsrc_t ares =
        AK ? make_rsrc_bytes(a.A + (long)i0 * a.lda + kbeg, ((long)(a.M - 1 - i0) * a.lda + (a.K - kbeg)) * 4)
           : make_rsrc_bytes(a.A + kbeg * a.lda + i0, ((a.K - 1 - kbeg) * a.lda + (a.M - i0)) * 4);
    const __amdgpu_buffer_rsrc_t bres =
        BK ? make_rsrc_bytes(a.B + (long)j0 * a.ldb + kbeg, ((long)(a.N - 1 - j0) * a.ldb + (a.K - kbeg)) * 4)
           : make_rsrc_bytes(a.B + kbeg * a.ldb + j0, ((a.K - 1 - kbeg) * a.ldb + (a.N - j0)) * 4);
    unsigned avoff[4], bvoff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (AK) avoff[p] = i0 + p * 32 + rowl < a.M ? (unsigned)(((long)rowl * a.lda + 4 * kq) * 4) : kBufOutside;
        else avoff[p] = i0 + 4 * rq < a.M ? (unsigned)(((long)(4 * cc + p) * a.lda + 4 * rq) * 4) : kBufOutside;
        if (BK) bvoff[p] = j0 + p * 32 + rowl < a.N ? (unsigned)(((long)rowl * a.ldb + 4 * kq) * 4) : kBufOutside;
        else bvoff[p] = j0 + 4 * rq < a.N ? (unsigned)(((long)(4 * cc + p) * a.ldb + 4 * rq) * 4) : kBufOutside;
    }
    const unsigned pass_a = (unsigned)(32 * a.lda * 4), pass_b = (unsigned)(32 * a.ldb * 4);
    const unsigned stage_a = AK ? (unsigned)(kNtKC * 4) : (unsigned)(kNtKC * a.lda * 4);
    const unsigned stage_b = BK ? (unsigned)(kNtKC * 4) : (unsigned)(kNtKC * a.ldb * 4);
    lds_float* const adst = AK ? sl + kq * kNtChunk + rowl * 4 : sl + cc * kNtChunk + rq * 16;
    lds_float* const bdst = (BK ? sl + kq * kNtChunk + rowl * 4 : sl + cc * kNtChunk + rq * 16) + kNtOp;
    const lds_float* afrag = sl + kh * kNtChunk + (wm * 64 + r) * 4;
    const lds_float* bfrag = sl + kNtOp + kh * kNtChunk + (wn * 64 + r) * 4;

    f32x4 ra[4], rb[4];   // pieces in flight (two stages ahead)
    auto issue = [&](int s) __attribute__((always_inline)) {   // (a stage past the end re-reads the last one: never committed)
        const int sc = s < nstage ? s : nstage - 1;
        // row-major pieces of k >= kend are whole loads beyond the K range: out of range through the scalar offset
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (AK) ra[p] = buf_load4v(ares, avoff[p], (unsigned)sc * stage_a + (unsigned)p * pass_a);
            else ra[p] = buf_load4v(ares, kbeg + (long)sc * kNtKC + 4 * cc + p < kend ? avoff[p] : kBufOutside, (unsigned)sc * stage_a);
            if (BK) rb[p] = buf_load4v(bres, bvoff[p], (unsigned)sc * stage_b + (unsigned)p * pass_b);
            else rb[p] = buf_load4v(bres, kbeg + (long)sc * kNtKC + 4 * cc + p < kend ? bvoff[p] : kBufOutside, (unsigned)sc * stage_b);
        }
    };
    auto mask_tail = [&](int s) __attribute__((always_inline)) {   // k-major pieces: zero the k >= kend part of the last stage
        const long k = kbeg + (long)s * kNtKC + 4 * kq;
        if (k + 4 > kend) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j >= kend) {
                        if (AK) ra[p][j] = 0.f;
                        if (BK) rb[p][j] = 0.f;
                    }
            }
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        if (AK) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(lds_f32x4*)(adst + buf * 2 * kNtOp + p * 128) = ra[p];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = {ra[0][e], ra[1][e], ra[2][e], ra[3][e]};     // row 4 rq + e, k = 4 cc .. 4 cc + 3
                *(lds_f32x4*)(adst + buf * 2 * kNtOp + e * 4) = t;
            }
        }
        if (BK) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(lds_f32x4*)(bdst + buf * 2 * kNtOp + p * 128) = rb[p];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = {rb[0][e], rb[1][e], rb[2][e], rb[3][e]};
                *(lds_f32x4*)(bdst + buf * 2 * kNtOp + e * 4) = t;
            }
        }
    };

    if (nstage > 0) {
        issue(0);
        if (nstage == 1) mask_tail(0);
        commit(0);
        issue(1);
        __syncthreads();
        auto stage = [&](auto tag, int s) __attribute__((always_inline)) {
            constexpr int CUR = decltype(tag)::value, NXT = CUR ^ 1;
            if (s + 1 == nstage - 1) mask_tail(s + 1);   // the pieces in registers belong to stage s + 1
            commit(NXT);
            issue(s + 2);
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *(const lds_f32x4*)(afrag + CUR * 2 * kNtOp + t * 128);
                fb[t] = *(const lds_f32x4*)(bfrag + CUR * 2 * kNtOp + t * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 ca[2], cb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ca[t] = fa[t];
                    cb[t] = fb[t];
                }
                if (g + 1 < 4) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        fa[t] = *(const lds_f32x4*)(afrag + CUR * 2 * kNtOp + (g + 1) * 2 * kNtChunk + t * 128);
                        fb[t] = *(const lds_f32x4*)(bfrag + CUR * 2 * kNtOp + (g + 1) * 2 * kNtChunk + t * 128);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a0 = ca[0][j], a1 = ca[1][j], b0 = cb[0][j], b1 = cb[1][j];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        for (int s = 0; s + 1 < nstage; s += 2) {
            stage(IntTag<0>(), s);
            stage(IntTag<1>(), s + 1);
        }
        if (nstage & 1) stage(IntTag<0>(), nstage - 1);
    }
    // (a split's partial image is addressed through blockIdx.z by EpiWorkspace: the launcher puts the split there too — see below)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int j = j0 + wn * 64 + tj * 32 + r;
        if (j >= a.N) continue;
        const typename EPI::Col c = epi.col(j, sz);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = i0 + wm * 64 + ti * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (i < a.M) epi.store(c, i, j, acc[ti][tj][q]);
            }
    }
}

// gemm128p_kernel: the same tile loop, PERSISTENT over the tiles of a product without a K split.  At K = 256 a tile is only eight
// stages long, and a workgroup per tile paid two exposed memory round trips (first stage, second stage) and a store burst per
// 13.6 us of MFMAs: 0.54 of the matrix peak at 196 608 x 256 x 256 where gemm_nt_bigk_kernel (thousands of stages per tile) holds 0.85.
// Here a workgroup walks tiles b, b + G, b + 2 G, ... (down a tile column: the small operand B stays in L2) as ONE stream of stages:
// the loads of the next tile's first two stages are in flight while the current tile's last two stages compute, and its 64 result
// stores per lane drain under the next tile's MFMAs.
template <bool AK, bool BK>
__global__ void __launch_bounds__(256, 2) gemm128p_kernel(Gemm128Args a, GemmEpi epi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][A | B][8][kNtChunk]
    lds_float* const sl = (lds_float*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, kh = lane >> 5;
    const int tiles_m = (a.M + 127) >> 7, tiles = tiles_m * ((a.N + 127) >> 7);
    const int nstage = (int)((a.K + kNtKC - 1) / kNtKC);
    const int G = gridDim.x;
    const int n_my = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    if (n_my == 0 || nstage == 0) return;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int rowl = tid >> 3, kq = tid & 7, cc = tid >> 5, rq = tid & 31;
    const unsigned pass_a = (unsigned)(32 * a.lda * 4), pass_b = (unsigned)(32 * a.ldb * 4);
    const unsigned stage_a = AK ? (unsigned)(kNtKC * 4) : (unsigned)(kNtKC * a.lda * 4);
    const unsigned stage_b = BK ? (unsigned)(kNtKC * 4) : (unsigned)(kNtKC * a.ldb * 4);
    // lane offsets inside a tile's window (row validity is added per tile at issue time)
    unsigned alane[4], blane[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        alane[p] = AK ? (unsigned)(((long)rowl * a.lda + 4 * kq) * 4) : (unsigned)(((long)(4 * cc + p) * a.lda + 4 * rq) * 4);
        blane[p] = BK ? (unsigned)(((long)rowl * a.ldb + 4 * kq) * 4) : (unsigned)(((long)(4 * cc + p) * a.ldb + 4 * rq) * 4);
    }
    lds_float* const adst = AK ? sl + kq * kNtChunk + rowl * 4 : sl + cc * kNtChunk + rq * 16;
    lds_float* const bdst = (BK ? sl + kq * kNtChunk + rowl * 4 : sl + cc * kNtChunk + rq * 16) + kNtOp;
    const lds_float* afrag = sl + kh * kNtChunk + (wm * 64 + r) * 4;
    const lds_float* bfrag = sl + kNtOp + kh * kNtChunk + (wn * 64 + r) * 4;

    // tile `it` of this workgroup: global tile t = blockIdx.x + it * G, walking down a tile column (i fastest)
    auto origin = [&](int it, int& i0, int& j0) __attribute__((always_inline)) {
        const int t = (int)blockIdx.x + it * G;
        i0 = (t % tiles_m) * 128;
        j0 = (t / tiles_m) * 128;
    };
    f32x4 ra[4], rb[4];   // pieces in flight
    int is_it = 0, is_s = 0;                 // the (tile, stage) the NEXT issue fetches
    // operand windows and row validity of the tile being fetched: rebuilt when the fetch cursor enters a tile, not per stage (the
    // 64-bit scalar arithmetic of two resources and a tile decode per issue was 190 SALU instructions per stage and wave)
    __amdgpu_buffer_rsrc_t ares, bres;
    unsigned aoff[4], boff[4];
    bool is_real = true;
    auto enter_tile = [&]() __attribute__((always_inline)) {
        is_real = is_it < n_my;
        int i0, j0;
        origin(is_real ? is_it : n_my - 1, i0, j0);
        ares = AK ? make_rsrc_bytes(a.A + (long)i0 * a.lda, ((long)(a.M - 1 - i0) * a.lda + a.K) * 4)
                  : make_rsrc_bytes(a.A + i0, ((a.K - 1) * a.lda + (a.M - i0)) * 4);
        bres = BK ? make_rsrc_bytes(a.B + (long)j0 * a.ldb, ((long)(a.N - 1 - j0) * a.ldb + a.K) * 4)
                  : make_rsrc_bytes(a.B + j0, ((a.K - 1) * a.ldb + (a.N - j0)) * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool aok = is_real && (AK ? i0 + p * 32 + rowl < a.M : i0 + 4 * rq < a.M);
            const bool bok = is_real && (BK ? j0 + p * 32 + rowl < a.N : j0 + 4 * rq < a.N);
            aoff[p] = aok ? alane[p] : kBufOutside;
            boff[p] = bok ? blane[p] : kBufOutside;
        }
    };
    enter_tile();
    auto issue = [&]() __attribute__((always_inline)) {
        const long k0 = (long)is_s * kNtKC;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // (a row-major piece is ONE k: beyond K it is a whole load out of range)
            const unsigned ao = AK || k0 + 4 * cc + p < a.K ? aoff[p] : kBufOutside;
            const unsigned bo = BK || k0 + 4 * cc + p < a.K ? boff[p] : kBufOutside;
            ra[p] = buf_load4v(ares, ao, (unsigned)is_s * stage_a + (AK ? (unsigned)p * pass_a : 0u));
            rb[p] = buf_load4v(bres, bo, (unsigned)is_s * stage_b + (BK ? (unsigned)p * pass_b : 0u));
        }
        if (++is_s == nstage) {
            is_s = 0;
            ++is_it;
            enter_tile();
        }
    };
    auto mask_tail = [&](int s) __attribute__((always_inline)) {   // k-major pieces of a tile's last stage: zero k >= K
        const long k = (long)s * kNtKC + 4 * kq;
        if (k + 4 > a.K) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j >= a.K) {
                        if (AK) ra[p][j] = 0.f;
                        if (BK) rb[p][j] = 0.f;
                    }
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        if (AK) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(lds_f32x4*)(adst + buf * 2 * kNtOp + p * 128) = ra[p];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = {ra[0][e], ra[1][e], ra[2][e], ra[3][e]};
                *(lds_f32x4*)(adst + buf * 2 * kNtOp + e * 4) = t;
            }
        }
        if (BK) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *(lds_f32x4*)(bdst + buf * 2 * kNtOp + p * 128) = rb[p];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 t = {rb[0][e], rb[1][e], rb[2][e], rb[3][e]};
                *(lds_f32x4*)(bdst + buf * 2 * kNtOp + e * 4) = t;
            }
        }
    };
    const bool ktail = (a.K % kNtKC) != 0 && (AK || BK);
    int reg_s = 0;                           // stage (inside its tile) of the pieces currently in registers
    issue();                                 // stage 0 of the first tile
    if (ktail && nstage == 1) mask_tail(0);
    commit(0);
    issue();                                 // the second stage of the stream
    reg_s = nstage == 1 ? 0 : 1;
    __syncthreads();
    auto stage = [&](auto tag) __attribute__((always_inline)) {
        constexpr int CUR = decltype(tag)::value, NXT = CUR ^ 1;
        if (ktail && reg_s == nstage - 1) mask_tail(reg_s);
        commit(NXT);                         // (behind the last stage of the stream: clamped pieces into a buffer nobody reads)
        issue();
        reg_s = reg_s + 1 == nstage ? 0 : reg_s + 1;
        f32x4 fa[2], fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[t] = *(const lds_f32x4*)(afrag + CUR * 2 * kNtOp + t * 128);
            fb[t] = *(const lds_f32x4*)(bfrag + CUR * 2 * kNtOp + t * 128);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 ca[2], cb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ca[t] = fa[t];
                cb[t] = fb[t];
            }
            if (g + 1 < 4) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[t] = *(const lds_f32x4*)(afrag + CUR * 2 * kNtOp + (g + 1) * 2 * kNtChunk + t * 128);
                    fb[t] = *(const lds_f32x4*)(bfrag + CUR * 2 * kNtOp + (g + 1) * 2 * kNtChunk + t * 128);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = ca[0][j], a1 = ca[1][j], b0 = cb[0][j], b1 = cb[1][j];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    // (an even stage count per tile: the buffer parity is the same at every tile's first stage — sg_gemm's dispatch guarantees it)
    for (int it = 0; it < n_my; ++it) {
        for (int s2 = 0; s2 < nstage; s2 += 2) {
            stage(IntTag<0>());
            stage(IntTag<1>());
        }
        // the tile is complete: its results leave while the next tile's first stages (already requested) arrive
        {
            int i0, j0;
            origin(it, i0, j0);
            // raw buffer stores: the resource starts at C(i0, j0) and ENDS with row M - 1, so rows beyond M are out of range by
            // themselves; a lane keeps one offset per column tile (4 kh rows down, its column), everything else is scalar
            const __amdgpu_buffer_rsrc_t cres = make_rsrc_bytes(epi.c + (long)i0 * epi.sci + j0, ((long)(a.M - 1 - i0) * epi.sci + (a.N - j0)) * 4);
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int jl = wn * 64 + tj * 32 + r;
                const bool jok = j0 + jl < a.N;
                const float bj = epi.bias_j && jok ? epi.bias_j[(j0 + jl) >> epi.bias_j_shift] : 0.f;
                const unsigned voff = jok ? (unsigned)(((long)(4 * kh) * epi.sci + jl) * 4) : kBufOutside;
                const unsigned row4 = (unsigned)epi.sci * 4u;          // bytes per row of C (the window is below 2 GiB)
                const unsigned sbase = (unsigned)(wm * 64) * row4;
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const unsigned soff = sbase + (unsigned)(ti * 32 + (q & 3) + 8 * (q >> 2)) * row4;
                        float v = acc[ti][tj][q] + bj;
                        if (epi.act == SG_ACT_LEAKY) v = v > 0.f ? v : v * epi.slope;       // (uniform: the common cases stay branch-free)
                        else if (epi.act == SG_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (epi.act != SG_ACT_NONE) v = sg_apply_act(v, epi.act, epi.slope);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), cres, (int)voff, (int)soff, 0);
                        acc[ti][tj][q] = 0.f;
                    }
            }
        }
    }
}

// the two epilogues of gemm128_kernel: the product itself through sg_gemm's GemmEpi, or a split's partial image
struct Gemm128Direct {
    GemmEpi e;
    typedef GemmEpi::Col Col;
    __device__ Col col(int j, int) const { return e.col(j); }
    __device__ void store(const Col& c, int i, int j, float v) const { e.store(c, i, j, v); }
};
struct Gemm128Partial {
    float* ws;
    int M, N;
    struct Col {
        long off;
    };
    __device__ Col col(int j, int split) const { return Col{(long)split * M * N + j}; }
    __device__ void store(const Col& c, int i, int j, float v) const { ws[c.off + (long)i * N] = v; }
};

// Returns 1 if the product was launched here, 0 if the shape / layout is left to the skeleton.
static int gemm128_try(const float* A, long sai, long sak, const float* B, long sbk, long sbj, const GemmEpi& epi, int M, int N,
                       int K, float* ws, size_t ws_bytes, hipStream_t stream) {
    constexpr bool off = SG_GEMM128 == 0;
    const bool ak = sak == 1 && sai != 1, bk = sbk == 1 && sbj != 1;      // (a degenerate dimension keeps the skeleton)
    if (off || (!ak && sai != 1) || (!bk && sbj != 1) || (!ak && bk)) return 0;
    const long lda = ak ? sai : sak, ldb = bk ? sbj : sbk;
    if (M < 128 || N < 128 || K < 64 || (long)M * N < (1L << 16)) return 0;
    if ((lda & 3) || (ldb & 3) || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15)) return 0;
    if ((!ak && (M & 3)) || (!bk && (N & 3))) return 0;
    // 32-bit byte offsets inside a workgroup's window: 128 rows (k-major) or one K chunk of rows (row-major) of an operand
    const long tiles = (long)sg_cdiv(M, 128) * sg_cdiv(N, 128);
    long s = tiles >= 256 ? 1 : (512 + tiles - 1) / tiles;
    const long maxs = K / (8 * kNtKC) > 0 ? K / (8 * kNtKC) : 1;
    if (s > maxs) s = maxs;
    if (s > 128) s = 128;
    if (s > 1 && (!ws || ws_bytes < (size_t)s * M * N * sizeof(float))) s = ws ? (long)(ws_bytes / ((size_t)M * N * sizeof(float))) : 1;
    if (s < 1) s = 1;
    if (tiles * s < 128) return 0;                                          // too small to fill the chip either way
    Gemm128Args g;
    g.A = A;
    g.B = B;
    g.lda = lda;
    g.ldb = ldb;
    g.M = M;
    g.N = N;
    g.K = K;
    g.kchunk = ((K + s - 1) / s + kNtKC - 1) / kNtKC * kNtKC;
    g.nsplit = (int)((K + g.kchunk - 1) / g.kchunk);
    const long win_a = ak ? 128 * lda + g.kchunk : g.kchunk * lda + 128, win_b = bk ? 128 * ldb + g.kchunk : g.kchunk * ldb + 128;
    if (win_a * 4 >= (long)kBufRange || win_b * 4 >= (long)kBufRange) return 0;
    const size_t lds = (size_t)2 * 2 * kNtOp * sizeof(float);
    // a product without a K split and with more tiles than resident workgroups runs persistently (-DSG_GEMM128=2 builds: one workgroup per tile)
    constexpr bool persist_off = SG_GEMM128 == 2;
    // (its epilogue writes rows of C through one 32-bit-offset window per tile that ends with row M - 1: unit column stride, rows
    // that do not overlap (sci >= N: the window's end is what clips the rows beyond M), no per-row bias, 128 rows of C below 2 GiB;
    // its operand windows span the whole K range)
    const bool persistent = !persist_off && tiles > 512 && ((K + kNtKC - 1) / kNtKC) % 2 == 0 && epi.scj == 1 && epi.sci >= N && !epi.bias_i && 128 * epi.sci * 4 < (long)kBufRange &&
                            (ak ? 128 * lda + K : (long)K * lda + 128) * 4 < (long)kBufRange &&
                            (bk ? 128 * ldb + K : (long)K * ldb + 128) * 4 < (long)kBufRange;
    const unsigned wgs = g.nsplit > 1 ? (unsigned)((g.nsplit + 7) / 8 * 8 * tiles) : (unsigned)tiles;
#define SG_G128(AK_, BK_)                                                                                                         \
    do {                                                                                                                         \
        static SgPerDeviceOnce once_d, once_p;                                                                                  \
        if (g.nsplit == 1 && persistent) {                                                                                       \
            static SgPerDeviceOnce once_pp;                                                                                      \
            if (once_pp.begin()) {                                                                                               \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128p_kernel<AK_, BK_>),                              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
                once_pp.end();                                                                                                   \
            }                                                                                                                    \
            hipLaunchKernelGGL((gemm128p_kernel<AK_, BK_>), dim3(tiles < 512 ? (unsigned)tiles : 512u), dim3(256), lds, stream, g, epi); \
        } else if (g.nsplit == 1) {                                                                                              \
            if (once_d.begin()) {                                                                                                \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128_kernel<AK_, BK_, Gemm128Direct>),                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
                once_d.end();                                                                                                    \
            }                                                                                                                    \
            hipLaunchKernelGGL((gemm128_kernel<AK_, BK_, Gemm128Direct>), dim3(wgs), dim3(256), lds, stream, g, Gemm128Direct{epi}); \
        } else {                                                                                                                 \
            if (once_p.begin()) {                                                                                                \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128_kernel<AK_, BK_, Gemm128Partial>),               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
                once_p.end();                                                                                                    \
            }                                                                                                                    \
            hipLaunchKernelGGL((gemm128_kernel<AK_, BK_, Gemm128Partial>), dim3(wgs), dim3(256), lds, stream, g,                 \
                               Gemm128Partial{ws, M, N});                                                                        \
        }                                                                                                                        \
    } while (0)
    if (ak && bk) SG_G128(true, true);
    else if (ak && !bk) SG_G128(true, false);
    else SG_G128(false, false);
#undef SG_G128
    if (g.nsplit > 1) {
        if (g.nsplit >= 32 && (long)M * N <= (1L << 18)) {
            hipLaunchKernelGGL((splitk_finalize_deep_kernel<GemmEpi>), dim3((unsigned)(((long)M * N + 63) / 64)), dim3(256), 0, stream,
                               (const float*)ws, epi, M, N, g.nsplit);
        } else {
            const long fb = (long)M * ((N + 1023) >> 10);
            hipLaunchKernelGGL((splitk_finalize_kernel<GemmEpi>), dim3((unsigned)fb), dim3(256), 0, stream, (const float*)ws, epi, M, N,
                               g.nsplit);
        }
    }
    return 1;
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
    if (gemm128_try(A, sai, sak, B, sbk, sbj, epi, M, N, K, ws, workspace_bytes, stream) == 1) {
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
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
    if (len <= 4096 && rows >= 1024)      // many short rows: one wave per row
        hipLaunchKernelGGL(rowsum_wave_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, out, rows, len, ld);
    else
        hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, out, rows, len, ld);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_rowsum_multi(const float* x, float* const* outs, const long* out_strides, int ndst, long rows_per_dst, long len, long ld,
                    hipStream_t stream) {
    SG_CHECK_ARG(x && outs && ndst > 0 && ndst <= 8 && rows_per_dst > 0 && len > 0);
    RowsumDst d;
    for (int i = 0; i < 8; ++i) {
        d.out[i] = outs[i < ndst ? i : 0];
        d.stride[i] = out_strides ? out_strides[i < ndst ? i : 0] : 1;
    }
    for (int i = 0; i < ndst; ++i) SG_CHECK_ARG(outs[i] != nullptr);
    hipLaunchKernelGGL(rowsum_multi_kernel, dim3((unsigned)(rows_per_dst * ndst)), dim3(256), 0, stream, x, d, rows_per_dst,
                       len, ld);
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

static int gemm_nt_launch(const float* A, const long* a_off, long lda, const float* B, const long* b_off, long ldb, float* C,
                          const long* c_off, const long* ldc, int batch, int M, int N, long K, void* workspace,
                          size_t workspace_bytes, hipStream_t stream, const char* who, const float* gamma = nullptr,
                          const float* beta = nullptr, const long* g_off = nullptr) {
    int nsplit;
    long kchunk;
    gemm_nt_plan(M, N, K, nsplit, kchunk, batch);
    const bool direct = nsplit == 1 && batch == 1;
    if (!direct && (!workspace || workspace_bytes < (size_t)batch * nsplit * M * N * sizeof(float)))
        SG_FAIL(SG_ERR_WORKSPACE, "%s: workspace too small", who);
    GemmNtArgs a;
    a.A = A;
    a.B = B;
    a.lda = lda;
    a.ldb = ldb;
    a.M = M;
    a.N = N;
    a.K = K;
    a.kchunk = kchunk;
    a.nsplit = nsplit;
    for (int b = 0; b < kNtMaxBatch; ++b) {
        a.a_off[b] = b < batch ? a_off[b] : 0;
        a.b_off[b] = b < batch ? b_off[b] : 0;
        a.g_off[b] = (b < batch && g_off) ? g_off[b] : 0;
    }
    a.gamma = gamma;
    a.beta = beta;
    a.out = direct ? C + c_off[0] : (float*)workspace;
    a.ldc = direct ? ldc[0] : N;
    const size_t lds = (size_t)2 * 2 * kNtOp * sizeof(float);
    static SgPerDeviceOnce attr_once;   // > 48 KB of dynamic LDS needs the attribute once per DEVICE
    if (attr_once.begin()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bigk_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bigk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_once.end();
    }
    a.ngroups = batch * nsplit;
    const unsigned wgs = (unsigned)((a.ngroups + 7) / 8 * 8 * sg_cdiv(N, 128) * sg_cdiv(M, 128));
    if (gamma)
        hipLaunchKernelGGL(gemm_nt_bigk_kernel<true>, dim3(wgs), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL(gemm_nt_bigk_kernel<false>, dim3(wgs), dim3(256), lds, stream, a);
    if (!direct) {
        GemmNtOut o;
        for (int b = 0; b < kNtMaxBatch; ++b) {
            o.c_off[b] = b < batch ? c_off[b] : 0;
            o.ldc[b] = b < batch ? ldc[b] : 0;
        }
        hipLaunchKernelGGL(gemm_nt_finalize_kernel, dim3((unsigned)(((long)M * N + 63) / 64), batch), dim3(256), 0, stream,
                           (const float*)workspace, C, o, M, N, nsplit);
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}

size_t sg_gemm_nt_workspace_bytes(int M, int N, long K) {
    int nsplit;
    long kchunk;
    gemm_nt_plan(M, N, K, nsplit, kchunk);
    return nsplit > 1 ? (size_t)nsplit * M * N * sizeof(float) : 0;
}

int sg_gemm_nt(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, long K, void* workspace,
               size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K && ldc >= N);
    SG_CHECK_ARG(31L * lda * 4 + 64 < (long)kBufRange && 31L * ldb * 4 + 64 < (long)kBufRange);   // lane offset
    SG_CHECK_ARG(96L * lda * 4 + K * 4 < (1L << 32) && 96L * ldb * 4 + K * 4 < (1L << 32));       // scalar offset
    const long zero = 0;
    return gemm_nt_launch(A, &zero, lda, B, &zero, ldb, C, &zero, &ldc, 1, M, N, K, workspace, workspace_bytes, stream,
                          "sg_gemm_nt");
}

size_t sg_gemm_nt_batched_workspace_bytes(int batch, int M, int N, long K) {
    int nsplit;
    long kchunk;
    gemm_nt_plan(M, N, K, nsplit, kchunk, batch);
    return (size_t)batch * nsplit * M * N * sizeof(float);
}

int sg_gemm_nt_batched(const float* A, const long* a_off, long lda, const float* B, const long* b_off, long ldb, float* C,
                       const long* c_off, const long* ldc, int batch, int M, int N, long K, void* workspace,
                       size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(A && B && C && a_off && b_off && c_off && ldc && batch > 0 && batch <= kNtMaxBatch);
    SG_CHECK_ARG(M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K);
    SG_CHECK_ARG(31L * lda * 4 + 64 < (long)kBufRange && 31L * ldb * 4 + 64 < (long)kBufRange);
    SG_CHECK_ARG(96L * lda * 4 + K * 4 < (1L << 32) && 96L * ldb * 4 + K * 4 < (1L << 32));
    for (int b = 0; b < batch; ++b) SG_CHECK_ARG(ldc[b] >= N);
    return gemm_nt_launch(A, a_off, lda, B, b_off, ldb, C, c_off, ldc, batch, M, N, K, workspace, workspace_bytes, stream,
                          "sg_gemm_nt_batched");
}


// The same with B = LayerNorm-normalised rows: C_b = A_b * relu(gamma_b (.) B_b + beta_b)^T, gamma_b / beta_b = gamma / beta + g_off[b]
// ([N] each, one value per row of B_b).  The weight gradients of the fused LayerNorm MLP (sg_sdfgen_bwd).
int sg_gemm_nt_batched_lnrelu(const float* A, const long* a_off, long lda, const float* B, const long* b_off, long ldb,
                              const float* gamma, const float* beta, const long* g_off, float* C, const long* c_off, const long* ldc,
                              int batch, int M, int N, long K, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(A && B && C && a_off && b_off && c_off && ldc && gamma && beta && g_off && batch > 0 && batch <= kNtMaxBatch);
    SG_CHECK_ARG(M > 0 && N > 0 && K > 0 && lda >= K && ldb >= K);
    SG_CHECK_ARG(31L * lda * 4 + 64 < (long)kBufRange && 31L * ldb * 4 + 64 < (long)kBufRange);
    SG_CHECK_ARG(96L * lda * 4 + K * 4 < (1L << 32) && 96L * ldb * 4 + K * 4 < (1L << 32));
    for (int b = 0; b < batch; ++b) SG_CHECK_ARG(ldc[b] >= N);
    return gemm_nt_launch(A, a_off, lda, B, b_off, ldb, C, c_off, ldc, batch, M, N, K, workspace, workspace_bytes, stream,
                          "sg_gemm_nt_batched_lnrelu", gamma, beta, g_off);
}

}  // extern "C"
