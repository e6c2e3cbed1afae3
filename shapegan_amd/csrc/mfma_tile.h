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
//     or lanes along k) are (nearly) bank-conflict free for ds_read_b32/ds_write_b32.
//   * 256 threads = 4 waves as 2x2; each wave owns TM x TN 32x32 tiles, so the block tile is
//     (64*TM) x (64*TN) x 16.  Global loads for tile t+1 are issued before the MFMAs of tile t
//     (register staging), two barriers per k-tile; the launcher sizes tiles / split-K so that
//     every CU holds >= 2 workgroups and one wave's staging VALU overlaps another's MFMAs.
//   * A loader owns its staging lane map and precomputes everything that does not change from
//     k-tile to k-tile (per-thread offsets, validity bits), so the per-element cost in the hot
//     loop is an add, a predicate and the load.  k-tiles are 16-aligned (kbeg is a multiple of 16).
#pragma once
#include "common.h"

namespace sg {

constexpr int kBK = 16;

// ---- loader concept (BR = rows of this operand's block tile, E = BR*16/256 elements per thread) ----------
//   template <int BR> void init(int tid, int row0, int nrows);          once per workgroup
//   template <int BR> void load(int k0, int kend, float (&r)[E]);       global -> registers, tile [k0, k0+16):
//        UNCONDITIONAL loads from clamped addresses (no branches), validity kept as a bit mask in the loader
//   template <int BR, int LO, int HI> void store(float (*S)[BR + 1], const float (&r)[E]) const;   elements [LO,HI)
//        registers -> LDS S[k][row]:
//        applies the validity mask here, so nothing consumes a load result before the MFMAs of the current tile
// Two staging lane maps are used:
//   lanes-along-k   : kk = tid & 15,          local row = (tid >> 4) + 16*it
//   lanes-along-rows: local row = tid % BR,   kk = tid / BR + it * (256 / BR)
//
// ---- epilogue concept -----------------------------------------------------------------------
//   struct Col;  Col col(int j) const;                    decode a column once
//   void store(const Col&, int i, int j, float v) const;  only called in-bounds

template <int BR>
struct StageKFast {  // lanes along k
    static constexpr int E = BR * kBK / 256;
    template <int LO, int HI>
    static __device__ __forceinline__ void store(float (*S)[BR + 1], const float (&r)[E], int tid, unsigned ok) {
        const int kk = tid & 15, lrow = tid >> 4;
#pragma unroll
        for (int it = LO; it < HI; ++it) S[kk][lrow + it * 16] = ((ok >> it) & 1u) ? r[it] : 0.f;
    }
};
template <int BR>
struct StageRowFast {  // lanes along rows
    static constexpr int E = BR * kBK / 256;
    static constexpr int STEP = 256 / BR;
    template <int LO, int HI>
    static __device__ __forceinline__ void store(float (*S)[BR + 1], const float (&r)[E], int tid, unsigned ok) {
        const int lrow = tid % BR, kq = tid / BR;
#pragma unroll
        for (int it = LO; it < HI; ++it) S[kq + it * STEP][lrow] = ((ok >> it) & 1u) ? r[it] : 0.f;
    }
};

// Plain matrix, element (row,k) at p[row*ld + k]  (k contiguous): lanes along k
struct MatRowMajor {
    const float* p;
    long ld;
    const float* q;  // p + (row0 + tid/16)*ld + tid%16
    int rbase, nrows_, kk, tid_;
    unsigned pend;
    template <int BR>
    __device__ void init(int tid, int row0, int nrows) {
        tid_ = tid;
        kk = tid & 15;
        rbase = row0 + (tid >> 4);
        nrows_ = nrows;
        q = p + (long)rbase * ld + kk;
    }
    template <int BR>
    __device__ void load(int k0, int kend, float (&r)[BR * kBK / 256]) {
        const bool kok = (k0 + kk) < kend;
        pend = 0;
#pragma unroll
        for (int it = 0; it < BR * kBK / 256; ++it) {
            const bool ok = kok && (rbase + it * 16) < nrows_;
            r[it] = *(ok ? q + ((long)it * 16 * ld + k0) : p);
            pend |= (ok ? 1u : 0u) << it;
        }
    }
    template <int BR, int LO, int HI>
    __device__ void store(float (*S)[BR + 1], const float (&r)[BR * kBK / 256]) const {
        StageKFast<BR>::template store<LO, HI>(S, r, tid_, pend);
    }
};
// Plain matrix, element (row,k) at p[k*ld + row]  (row contiguous): lanes along rows
struct MatColMajor {
    const float* p;
    long ld;
    const float* q;  // p + (tid/BR)*ld + row
    int kq, tid_;
    bool rok;
    unsigned pend;
    template <int BR>
    __device__ void init(int tid, int row0, int nrows) {
        tid_ = tid;
        const int row = row0 + tid % BR;
        kq = tid / BR;
        rok = row < nrows;
        q = p + (long)kq * ld + row;
    }
    template <int BR>
    __device__ void load(int k0, int kend, float (&r)[BR * kBK / 256]) {
        constexpr int STEP = 256 / BR;
        pend = 0;
#pragma unroll
        for (int it = 0; it < BR * kBK / 256; ++it) {
            const bool ok = rok && (k0 + kq + it * STEP) < kend;
            r[it] = *(ok ? q + (long)(k0 + it * STEP) * ld : p);
            pend |= (ok ? 1u : 0u) << it;
        }
    }
    template <int BR, int LO, int HI>
    __device__ void store(float (*S)[BR + 1], const float (&r)[BR * kBK / 256]) const {
        StageRowFast<BR>::template store<LO, HI>(S, r, tid_, pend);
    }
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

    la.template init<BM>(tid, i0, M);
    lb.template init<BN>(tid, j0, N);

    float ra[EA], rb[EB];

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

    // Pipeline (per k-tile t):  [MFMA burst(t) with the LDS stores of tile t+1 woven into its second half]
    //   -> barrier A (tile t+1 visible) -> issue global loads of tile t+2 -> read all fragments of tile t+1
    //   -> lgkmcnt(0) + raw barrier B (LDS free again) -> burst(t+1) ...
    // The loads of tile t+1 were issued a whole burst earlier, so nothing waits on HBM/L2 in front of an MFMA, and
    // the matrix pipe only idles for the two barriers and the fragment reads between bursts.
    float af[BK / 2][TM], bf[BK / 2][TN];
    auto read_frags = [&]() {
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
#pragma unroll
            for (int t = 0; t < TM; ++t) af[s][t] = As[2 * s + kh][wm * 32 * TM + t * 32 + r];
#pragma unroll
            for (int t = 0; t < TN; ++t) bf[s][t] = Bs[2 * s + kh][wn * 32 * TN + t * 32 + r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only (vmcnt/expcnt fields left at max)
        __builtin_amdgcn_s_barrier();
    };
    if (kbeg < kend) {
        la.template load<BM>(kbeg, kend, ra);
        lb.template load<BN>(kbeg, kend, rb);
        la.template store<BM, 0, EA>(As, ra);
        lb.template store<BN, 0, EB>(Bs, rb);
        __syncthreads();
        if (kbeg + BK < kend) {
            la.template load<BM>(kbeg + BK, kend, ra);
            lb.template load<BN>(kbeg + BK, kend, rb);
        }
        read_frags();
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][ta], bf[s][tb], acc[ta][tb], 0, 0, 0);
            if (more) {  // second half of the burst: a quarter of tile t+1's LDS stores behind each k-step
                if (s == 4) {
                    la.template store<BM, 0, EA / 4>(As, ra);
                    lb.template store<BN, 0, EB / 4>(Bs, rb);
                } else if (s == 5) {
                    la.template store<BM, EA / 4, EA / 2>(As, ra);
                    lb.template store<BN, EB / 4, EB / 2>(Bs, rb);
                } else if (s == 6) {
                    la.template store<BM, EA / 2, 3 * EA / 4>(As, ra);
                    lb.template store<BN, EB / 2, 3 * EB / 4>(Bs, rb);
                } else if (s == 7) {
                    la.template store<BM, 3 * EA / 4, EA>(As, ra);
                    lb.template store<BN, 3 * EB / 4, EB>(Bs, rb);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            __syncthreads();  // A: tile t+1 visible
            if (k0 + 2 * BK < kend) {
                la.template load<BM>(k0 + 2 * BK, kend, ra);
                lb.template load<BN>(k0 + 2 * BK, kend, rb);
            }
            read_frags();     // fragments of tile t+1, then barrier B
        }
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

// sums the split-K partials and hands each element to the real epilogue.  One workgroup = 1024 consecutive columns of
// one row (no per-element division); the partials of 4 columns are read as one dwordx4 when the row length allows it.
template <class EPI>
__global__ void __launch_bounds__(256) splitk_finalize_kernel(const float* ws, EPI epi, int M, int N, int splits) {
    const int ncb = (N + 1023) >> 10;
    const int i = blockIdx.x / ncb, j0 = (blockIdx.x - i * ncb) * 1024 + threadIdx.x * 4;
    if (j0 >= N) return;
    const long total = (long)M * N;
    const float* p = ws + (long)i * N + j0;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((N & 3) == 0) {
        int s = 0;
        for (; s + 4 <= splits; s += 4) {
            const float4 q0 = *(const float4*)(p + (long)s * total), q1 = *(const float4*)(p + (long)(s + 1) * total);
            const float4 q2 = *(const float4*)(p + (long)(s + 2) * total), q3 = *(const float4*)(p + (long)(s + 3) * total);
            v[0] += (q0.x + q1.x) + (q2.x + q3.x);
            v[1] += (q0.y + q1.y) + (q2.y + q3.y);
            v[2] += (q0.z + q1.z) + (q2.z + q3.z);
            v[3] += (q0.w + q1.w) + (q2.w + q3.w);
        }
        for (; s < splits; ++s) {
            const float4 q = *(const float4*)(p + (long)s * total);
            v[0] += q.x, v[1] += q.y, v[2] += q.z, v[3] += q.w;
        }
    } else {
        for (int s = 0; s < splits; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < N) v[u] += p[(long)s * total + u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (j0 + u < N) epi.store(epi.col(j0 + u), i, j0 + u, v[u]);
}

// many partials of a small matrix (Cin = 1 weight gradients: 512 partials of 64 x 64): one workgroup per 64 consecutive
// elements, the 4 waves take interleaved partials and meet in LDS — 128 serial loads per thread instead of 512
template <class EPI>
__global__ void __launch_bounds__(256) splitk_finalize_deep_kernel(const float* ws, EPI epi, int M, int N, int splits) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long total = (long)M * N, e = (long)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int s = wave;
        for (; s + 4 < splits; s += 8) {
            s0 += ws[(long)s * total + e];
            s1 += ws[(long)(s + 4) * total + e];
        }
        if (s < splits) s0 += ws[(long)s * total + e];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && e < total) {
        const int i = (int)(e / N), j = (int)(e - (long)i * N);
        epi.store(epi.col(j), i, j, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
    }
}

constexpr long kTargetBlocks = 512;  // >= 2 workgroups on each of the 256 CUs

struct TilePlan {
    int tm, tn, splitk;
};
// Largest tile that reaches kTargetBlocks workgroups, using split-K (when a workspace is there) to multiply the
// workgroup count of big tiles: big tiles stage fewer bytes per MFMA, split-K costs one extra pass over M*N*s floats.
static inline TilePlan plan_tiles(int M, int N, int K, long ws_floats_avail, int nbatch = 1) {
    const int cand[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
    const long smax_k = K / (8 * kBK) > 0 ? K / (8 * kBK) : 1;  // every split keeps >= 8 k-tiles
    const long smax_ws = ws_floats_avail > 0 ? ws_floats_avail / ((long)M * N) : 1;
    TilePlan best{1, 1, 1};
    long best_blocks = -1;
    for (int c = 0; c < 4; ++c) {
        const int tm = cand[c][0], tn = cand[c][1];
        if ((tm == 2 && M <= 64) || (tn == 2 && N <= 64)) continue;
        const long tiles = (long)sg_cdiv(M, 64 * tm) * sg_cdiv(N, 64 * tn) * nbatch;
        long s = (kTargetBlocks + tiles - 1) / tiles;
        if (s > smax_k) s = smax_k;
        if (s > smax_ws) s = smax_ws;
        if (s < 1) s = 1;
        if (tiles * s >= kTargetBlocks) return TilePlan{tm, tn, (int)s};
        if (tiles * s > best_blocks) {
            best_blocks = tiles * s;
            best = TilePlan{tm, tn, (int)s};
        }
    }
    return best;
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
        if (splitk >= 32 && (long)M * N <= (1L << 18)) {
            hipLaunchKernelGGL((splitk_finalize_deep_kernel<EPI>), dim3((unsigned)(((long)M * N + 63) / 64)), dim3(256), 0, st,
                               (const float*)ws, epi, M, N, splitk);
        } else {
            const long fb = (long)M * ((N + 1023) >> 10);
            hipLaunchKernelGGL((splitk_finalize_kernel<EPI>), dim3((unsigned)fb), dim3(256), 0, st, (const float*)ws, epi,
                               M, N, splitk);
        }
    }
    return 0;
}

// ws may be null (then no split-K). ws_bytes is the size of the caller-owned workspace.
template <class LA, class LB, class EPI>
static int launch_tile_gemm(LA la, LB lb, EPI epi, int M, int N, int K, float* ws, size_t ws_bytes, hipStream_t st) {
    const TilePlan p = plan_tiles(M, N, K, ws ? (long)(ws_bytes / sizeof(float)) : 0);
    if (p.tm == 2 && p.tn == 2) return launch_tile_gemm_t<2, 2>(la, lb, epi, M, N, K, p.splitk, ws, st);
    if (p.tm == 2 && p.tn == 1) return launch_tile_gemm_t<2, 1>(la, lb, epi, M, N, K, p.splitk, ws, st);
    if (p.tm == 1 && p.tn == 2) return launch_tile_gemm_t<1, 2>(la, lb, epi, M, N, K, p.splitk, ws, st);
    return launch_tile_gemm_t<1, 1>(la, lb, epi, M, N, K, p.splitk, ws, st);
}

}  // namespace sg
