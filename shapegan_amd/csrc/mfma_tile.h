// shapegan_amd/csrc/mfma_tile.h — the one LDS-tiled f32-MFMA GEMM skeleton every dense op is built on.
//
//   D[i][j] = sum_k A(i,k) * B(k,j)           i in [0,M)  j in [0,N)  k in [0,K)
//
// A and B are *loader functors* (plain strided matrices, or the implicit-GEMM gathers of the
// k4/s2/p1 3-D convolution — conv3d.hip), D goes through an *epilogue functor* that owns the
// output addressing (NCDHW scatter, bias, activation).  MFMA rows (i) are the operand that is
// NOT contiguous in the output, MFMA columns (j) are the contiguous one: in the C/D fragment
// of v_mfma_f32_32x32x2_f32 consecutive lanes hold consecutive j, so stores coalesce.
//
// gfx950 specifics (MI355X_MICROARCH.md / cdna_hip_programming.md §3):
//   * v_mfma_f32_32x32x2_f32: exact f32 (an fmaf chain), 64 cycles per SIMD, 157 TF chip peak.
//     A frag: lane l holds A[i=l&31][k=l>>5]; B frag: lane l holds B[k=l>>5][j=l&31];
//     D frag: j = l&31, i = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//   * LDS tiles are stored [k][row] with an odd leading dimension (BM+1): the fragment reads
//     (32 consecutive rows per half-wave) and both staging write patterns (lanes along rows,
//     or lanes along k) are bank-conflict free for ds_read_b32/ds_write_b32.
//   * 256 threads = 4 waves as 2x2; each wave owns TM x TN 32x32 tiles, so the block tile is
//     (64*TM) x (64*TN) x 16.  Global loads for tile t+1 are issued before the MFMAs of tile t
//     (register staging), two barriers per k-tile; >=2 blocks per CU hide the barrier.
#pragma once
#include "common.h"

namespace sg {

constexpr int kBK = 16;

// ---- loader concept -------------------------------------------------------------------------
//   static constexpr bool K_FAST;     staging lane order: lanes along k (true) or along rows
//   void  fix(int a);                 pins the slow index (row if !K_FAST, k if K_FAST)
//   float get(int b) const;           element at (pinned, b); only called in-bounds
//
// ---- epilogue concept -----------------------------------------------------------------------
//   struct Col;  Col col(int j) const;                    decode a column once
//   void store(const Col&, int i, int j, float v) const;  only called in-bounds

// Plain matrix, element (row,k) at p[row*ld + k]  (k contiguous)
struct MatRowMajor {
    static constexpr bool K_FAST = true;
    const float* p;
    long ld;
    int kk;
    __device__ void fix(int k) { kk = k; }
    __device__ float get(int row) const { return p[(long)row * ld + kk]; }
};
// Plain matrix, element (row,k) at p[k*ld + row]  (row contiguous)
struct MatColMajor {
    static constexpr bool K_FAST = false;
    const float* p;
    long ld;
    int rr;
    __device__ void fix(int row) { rr = row; }
    __device__ float get(int k) const { return p[(long)k * ld + rr]; }
};

// Split-K partial sums: ws[split][i][j]
struct EpiWorkspace {
    float* ws;
    int M, N;
    struct Col {
        long off;
    };
    __device__ Col col(int j) const { return Col{(long)blockIdx.z * M * N + j}; }
    __device__ void store(const Col& c, int i, int j, float v) const { ws[c.off + (long)i * N] = v; }
};

template <int TM, int TN, class LA, class LB, class EPI>
__global__ void __launch_bounds__(256) tile_gemm_kernel(LA la, LB lb, EPI epi, int M, int N, int K, int kchunk) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = kBK;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
    __shared__ float As[BK][BM + 1];
    __shared__ float Bs[BK][BN + 1];

    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * BN;
    const int i0 = blockIdx.y * BM;
    // kchunk > 0: blockIdx.z is a split-K slice;  kchunk <= 0: blockIdx.z is a batch index owned by the functors
    const int kbeg = kchunk > 0 ? blockIdx.z * kchunk : 0;
    const int kend = kchunk > 0 ? min(K, kbeg + kchunk) : K;

    // staging coordinates
    const int a_row = LA::K_FAST ? (tid >> 4) : (tid % BM);  // + it*16 if K_FAST
    const int a_k = LA::K_FAST ? (tid & 15) : (tid / BM);    // + it*(256/BM) if !K_FAST
    const int b_row = LB::K_FAST ? (tid >> 4) : (tid % BN);
    const int b_k = LB::K_FAST ? (tid & 15) : (tid / BN);

    if constexpr (!LA::K_FAST) la.fix(i0 + a_row);
    if constexpr (!LB::K_FAST) lb.fix(j0 + b_row);
    const bool a_row_ok = (i0 + a_row) < M;
    const bool b_row_ok = (j0 + b_row) < N;

    float ra[EA], rb[EB];

    auto gload = [&](int k0) {
        if constexpr (LA::K_FAST) {
            const int k = k0 + a_k;
            la.fix(k);
            const bool kok = k < kend;
#pragma unroll
            for (int it = 0; it < EA; ++it) {
                const int row = i0 + a_row + it * 16;
                ra[it] = (kok && row < M) ? la.get(row) : 0.f;
            }
        } else {
#pragma unroll
            for (int it = 0; it < EA; ++it) {
                const int k = k0 + a_k + it * (256 / BM);
                ra[it] = (a_row_ok && k < kend) ? la.get(k) : 0.f;
            }
        }
        if constexpr (LB::K_FAST) {
            const int k = k0 + b_k;
            lb.fix(k);
            const bool kok = k < kend;
#pragma unroll
            for (int it = 0; it < EB; ++it) {
                const int row = j0 + b_row + it * 16;
                rb[it] = (kok && row < N) ? lb.get(row) : 0.f;
            }
        } else {
#pragma unroll
            for (int it = 0; it < EB; ++it) {
                const int k = k0 + b_k + it * (256 / BN);
                rb[it] = (b_row_ok && k < kend) ? lb.get(k) : 0.f;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int it = 0; it < EA; ++it) {
            if constexpr (LA::K_FAST)
                As[a_k][a_row + it * 16] = ra[it];
            else
                As[a_k + it * (256 / BM)][a_row] = ra[it];
        }
#pragma unroll
        for (int it = 0; it < EB; ++it) {
            if constexpr (LB::K_FAST)
                Bs[b_k][b_row + it * 16] = rb[it];
            else
                Bs[b_k + it * (256 / BN)][b_row] = rb[it];
        }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, kh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = As[2 * s + kh][wm * 32 * TM + t * 32 + r];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = Bs[2 * s + kh][wn * 32 * TN + t * 32 + r];
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int j = j0 + wn * 32 * TN + tb * 32 + r;
        if (j >= N) continue;
        const typename EPI::Col c = epi.col(j);
#pragma unroll
        for (int ta = 0; ta < TM; ++ta) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = i0 + wm * 32 * TM + ta * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (i < M) epi.store(c, i, j, acc[ta][tb][q]);
            }
        }
    }
}

// sums the split-K partials and hands each element to the real epilogue
template <class EPI>
__global__ void __launch_bounds__(256) splitk_finalize_kernel(const float* ws, EPI epi, int M, int N, int splits) {
    const long total = (long)M * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int i = (int)(e / N), j = (int)(e % N);
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += ws[(long)s * total + e];
        epi.store(epi.col(j), i, j, v);
    }
}

// Picks the split-K factor: enough blocks to cover 256 CUs about twice, each split >= 8 k-tiles.
static inline int choose_splitk(int M, int N, int K, int tm, int tn, long ws_floats_avail) {
    const long tiles = (long)sg_cdiv(M, 64 * tm) * sg_cdiv(N, 64 * tn);
    if (tiles >= 256 || K < 2 * 8 * kBK) return 1;
    long s = (512 + tiles - 1) / tiles;
    const long smax_k = K / (8 * kBK);
    if (s > smax_k) s = smax_k;
    const long per = (long)M * N;
    if (per * s > ws_floats_avail) s = ws_floats_avail / per;
    if (s < 1) s = 1;
    return (int)s;
}

template <int TM, int TN, class LA, class LB, class EPI>
static int launch_tile_gemm_t(LA la, LB lb, EPI epi, int M, int N, int K, int splitk, float* ws, hipStream_t st) {
    dim3 grid(sg_cdiv(N, 64 * TN), sg_cdiv(M, 64 * TM), splitk);
    if (splitk == 1) {
        hipLaunchKernelGGL((tile_gemm_kernel<TM, TN, LA, LB, EPI>), grid, dim3(256), 0, st, la, lb, epi, M, N, K, K);
    } else {
        int kchunk = sg_cdiv(sg_cdiv(K, splitk), kBK) * kBK;
        EpiWorkspace wepi{ws, M, N};
        hipLaunchKernelGGL((tile_gemm_kernel<TM, TN, LA, LB, EpiWorkspace>), grid, dim3(256), 0, st, la, lb, wepi, M, N,
                           K, kchunk);
        long total = (long)M * N;
        int fb = (int)((total + 255) / 256);
        if (fb > 2048) fb = 2048;
        hipLaunchKernelGGL((splitk_finalize_kernel<EPI>), dim3(fb), dim3(256), 0, st, (const float*)ws, epi, M, N,
                           splitk);
    }
    return 0;
}

// ws may be null (then no split-K). ws_bytes is the size of the caller-owned workspace.
template <class LA, class LB, class EPI>
static int launch_tile_gemm(LA la, LB lb, EPI epi, int M, int N, int K, float* ws, size_t ws_bytes, hipStream_t st) {
    const int tm = M > 64 ? 2 : 1, tn = N > 64 ? 2 : 1;
    const int splitk = ws ? choose_splitk(M, N, K, tm, tn, (long)(ws_bytes / sizeof(float))) : 1;
    if (tm == 2 && tn == 2) return launch_tile_gemm_t<2, 2>(la, lb, epi, M, N, K, splitk, ws, st);
    if (tm == 2 && tn == 1) return launch_tile_gemm_t<2, 1>(la, lb, epi, M, N, K, splitk, ws, st);
    if (tm == 1 && tn == 2) return launch_tile_gemm_t<1, 2>(la, lb, epi, M, N, K, splitk, ws, st);
    return launch_tile_gemm_t<1, 1>(la, lb, epi, M, N, K, splitk, ws, st);
}

}  // namespace sg
