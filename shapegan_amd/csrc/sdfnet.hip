// shapegan_amd/csrc/sdfnet.hip — the fused DeepSDF MLP (K7), the MFMA path of the north star.
//
// Replaces SDFNet.forward (model/sdf_net.py:56-61) and its autograd backward:
//   input = cat(points[N,3], latent[N,L]);  x = layers1(input);  x = cat(x, input);  x = layers2(x)
//   layers1 = Linear(3+L,256) ReLU, 3 x [Linear(256,256) ReLU]              (model/sdf_net.py:26-38)
//   layers2 = Linear(259+L,256) ReLU, 2 x [Linear(256,256) ReLU], Linear(256,1) Tanh   (:40-52)
//
// One workgroup (512 threads = 8 waves, 2 per SIMD) walks a tile of P points through all eight layers
// without touching HBM in between: the activation tile H[256][P] lives in LDS (feature-major, points
// contiguous, so the MFMA B-fragment read of 32 consecutive points per half-wave is conflict-free), each
// wave owns 32 output features x P points (P/32 accumulators of v_mfma_f32_32x32x2_f32, exact f32), and the
// weights stream from L2 in an MFMA-A-fragment-packed image (one coalesced global_load_dwordx4 per lane per
// 4 k-steps, 1 KiB per wave-instruction).  The 1.85 MB of weights are L2-resident; per point the kernel
// moves 12 B in (xyz) + 4 B out, the 921 088 FLOP/point (L=128) are all on the matrix pipe.
//
// Two input modes:
//   per-point latent (P=64):  X[3+L][P] is staged in LDS next to H; layers 1 and 5 run K over X as well
//                             (reference semantics for arbitrary latent_codes[N,L], train_sdf_autodecoder.py:80-87)
//   per-shape latent (P=64):  every point of a shape shares z_s (hybrid GANs sample a fixed grid per shape,
//                             train_hybrid_wgan.py:67-72, train_hybrid_progressive_gan.py:90-96) - the latent
//                             columns of layers 1 and 5 fold into per-shape bias vectors zb1/zb5 (a [S,L]x[L,256]
//                             GEMM), the [N,L] tiling (2.15 GB at 64^3, B=16) is never materialised.
//
// Training: `acts` receives H1..H7 (feature-major [7][256][ldn]); the fused backward-data kernel walks the
// chain in reverse (dZ_l = dH_l * (H_l > 0), dH_{l-1} = W_l^T dZ_l on the matrix pipe with transposed packs),
// writes dZ1..dZ7 for the weight-gradient GEMMs (gemm.hip, split-K over points) and the input gradient.
#include "common.h"
#include "../../include/shapegan_hip.h"

// tuning switches (scripts/ab_build.sh builds variants; the defaults are the product)
#ifndef SG_BWD_RING
#define SG_BWD_RING 4
#endif
#define SG_BWD_TILE 64
// cache policy of the activation / dZ image stores (aux of the raw buffer store: 0 default, 2 = nt: streaming, evict-first in L2 —
// the images are written once and read by a later kernel, the weight packs every wave streams should keep the L2)
#ifndef SG_IMG_AUX
#define SG_IMG_AUX 2
#endif   // (the tile layout of the partial sums is part of the ABI: include/shapegan_hip.h)

namespace sg {

constexpr int kH = 256;  // SDF_NET_BREADTH, model/sdf_net.py:21

struct PackDesc {
    const float* src;
    long rs, cs;  // element (r,c) at src[r*rs + c*cs]
    int R, C;     // valid extent
    int ntiles;   // padded rows / 32
    int nsq;      // padded cols / 8
    long dst_off;
};
struct PackVectors {     // the seven bias vectors, b8 and w8 (blockIdx.y == descs.n of the pack launch)
    const float* b[8];
    const float* w8;
    long dst_b, dst_w8;  // offsets into the packed buffer
    const float* g[7];   // LayerNorm weight / bias of the seven hidden layers (SDFGenerator, model/point_sdf_net.py:71); g[0] NULL: none
    const float* be[7];
    long dst_g, dst_be;
};
struct PackDescs {
    PackDesc d[16];
    int n;
    PackVectors v;
};

// dst[((t*nsq + sq)*64 + lane)*4 + j] = M(t*32 + (lane&31), (sq*4 + j)*2 + (lane>>5))
// (row blockIdx.y == descs.n: the bias vectors and w8 — one launch instead of two, round 6)
__device__ __forceinline__ void pack_mfma_a_body(const PackDescs& descs, float* __restrict__ dst, const unsigned bx, const unsigned by,
                                                 const unsigned gx) {
    if ((int)by == descs.n) {
        if (bx != 0) return;
        const int t = threadIdx.x;
#pragma unroll
        for (int l = 0; l < 7; ++l) dst[descs.v.dst_b + l * kH + t] = descs.v.b[l][t];
        dst[descs.v.dst_b + 7 * kH + t] = (t == 0) ? descs.v.b[7][0] : 0.f;
        dst[descs.v.dst_w8 + t] = descs.v.w8[t];
        if (descs.v.g[0]) {
#pragma unroll
            for (int l = 0; l < 7; ++l) {
                dst[descs.v.dst_g + l * kH + t] = descs.v.g[l][t];
                dst[descs.v.dst_be + l * kH + t] = descs.v.be[l][t];
            }
        }
        return;
    }
    const PackDesc d = descs.d[by];
    const long total = (long)d.ntiles * d.nsq * 256;
    for (long e = (long)bx * 256 + threadIdx.x; e < total; e += (long)gx * 256) {
        const int j = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        const long q = e >> 8;
        const int sq = (int)(q % d.nsq);
        const int t = (int)(q / d.nsq);
        const int r = t * 32 + (lane & 31);
        const int c = (sq * 4 + j) * 2 + (lane >> 5);
        dst[d.dst_off + e] = (r < d.R && c < d.C) ? d.src[(long)r * d.rs + (long)c * d.cs] : 0.f;
    }
}
__global__ void __launch_bounds__(256) pack_mfma_a_kernel(PackDescs descs, float* __restrict__ dst) {
    pack_mfma_a_body(descs, dst, blockIdx.x, blockIdx.y, gridDim.x);
}

struct SdfPackLayout {
    int KU, KUp, KUr;
    long F1, F2, F3, F4, F5x, F5i, F6, F7;  // forward packs: A(i=out, k=in)
    long T1, T2, T3, T4, T5x, T5i, T6, T7;  // transposed packs: A(i=in, k=out)
    long W8, B;                              // w8[256], b[8][256]
    long G, Be;                              // LayerNorm weight / bias [7][256] each (the SDFGenerator form)
    long total;
};
static SdfPackLayout make_layout(int KU) {
    SdfPackLayout L;
    L.KU = KU;
    L.KUp = (KU + 7) / 8 * 8;
    L.KUr = (KU + 31) / 32 * 32;
    long o = 0;
    auto take = [&](long n) {
        long r = o;
        o += n;
        return r;
    };
    L.F1 = take((long)kH * L.KUp);
    L.F2 = take(kH * kH);
    L.F3 = take(kH * kH);
    L.F4 = take(kH * kH);
    L.F5x = take(kH * kH);
    L.F5i = take((long)kH * L.KUp);
    L.F6 = take(kH * kH);
    L.F7 = take(kH * kH);
    L.T1 = take((long)L.KUr * kH);
    L.T2 = take(kH * kH);
    L.T3 = take(kH * kH);
    L.T4 = take(kH * kH);
    L.T5x = take(kH * kH);
    L.T5i = take((long)L.KUr * kH);
    L.T6 = take(kH * kH);
    L.T7 = take(kH * kH);
    L.W8 = take(kH);
    L.B = take(8 * kH);
    L.G = take(7 * kH);
    L.Be = take(7 * kH);
    L.total = o;
    return L;
}

// acc[t] += A_tile(32 x K) * B(K x [t*32, t*32+32)) ; wp = this wave's packed A rows (wave-uniform), Bs = LDS [K][ld].
// Software-pipelined like the conv halo kernels: A fragments come through a 4-deep ring of buffer loads (scalar base,
// fixed lane offset, k-group in the scalar offset), the B fragments of k-group sq+1 are read from LDS (immediate offsets
// from one running address) while the 4*NT MFMAs of group sq run; sched_barriers keep the loads where they are issued.
// One VALU instruction (the LDS address step) per 4*NT MFMAs.
template <int NT, int RING = 4, int TS = 32>   // TS: floats between the column tiles of a row (32: one [K][ld] tile; else one tile per chain)
__device__ __forceinline__ void mlp_gemm(f32x16 (&acc)[NT], const float4* __restrict__ wp, int nsq,
                                         const float* __restrict__ Bs, int ld, int lane) {
    const int r = lane & 31, kh = lane >> 5;
    // the packed rows are per wave: make the base a scalar so that the loads need no vector address arithmetic
    const unsigned long long wq = (unsigned long long)wp;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)wq), hi = __builtin_amdgcn_readfirstlane((unsigned)(wq >> 32));
    const __amdgpu_buffer_rsrc_t wres = make_rsrc((const void*)(((unsigned long long)hi << 32) | lo));
    const unsigned wvoff = lane * 16;
    const lds_float* bp = (const lds_float*)Bs + kh * ld + r;
    const int last = nsq - 1;
    float4 ar[RING];
#pragma unroll
    for (int u = 0; u < RING; ++u) ar[u] = buf_load4(wres, wvoff, (unsigned)(u < last ? u : last) * 1024u);
    float b[4][NT], bn[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[j][t] = bp[j * 2 * ld + t * TS];
    auto group = [&](float4 a, int sq) __attribute__((always_inline)) {
        // B of the next group (the last group re-reads its own: no branch)
        const lds_float* nb = bp + (sq < last ? 8 * ld : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[j][t] = nb[j * 2 * ld + t * TS];
        bp = nb;
        __builtin_amdgcn_sched_barrier(0);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[j][t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) b[j][t] = bn[j][t];
    };
    int sq = 0;
    for (; sq + RING <= nsq; sq += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const float4 a = ar[u];
            const int nx = sq + u + RING;
            ar[u] = buf_load4(wres, wvoff, (unsigned)(nx < last ? nx : last) * 1024u);
            group(a, sq + u);
        }
    }
    // remainder (group count not a multiple of the ring): the ring already holds these groups
#pragma unroll
    for (int u = 0; u < RING - 1; ++u)
        if (sq + u < nsq) group(ar[u], sq + u);
}

// The same GEMM with the weight ring as caller-visible state, for the 256-wide layers that follow one another:
//   * wring_start() issues the first RING weight loads of a layer.  The caller does that BEFORE the stores of the previous
//     layer's epilogue: vmcnt retires in order, so a ring started after the 32 - 64 activation stores of an epilogue makes the
//     first MFMA of the next layer wait for every one of those stores to be acknowledged by memory;
//   * hook() runs once, right after the LAST weight load of the layer has been issued (RING groups before the end): loads the
//     caller wants to have arrived by the end of the GEMM (the backward's ReLU-mask operand) go there — issued earlier they
//     would sit in front of the remaining weight loads in the in-order return queue and stall the MFMAs for a full memory
//     latency, issued later their latency is exposed in the epilogue.
template <int RING>
struct WRing {
    __amdgpu_buffer_rsrc_t res;
    float4 ar[RING];
};
template <int RING>
__device__ __forceinline__ void wring_start(WRing<RING>& w, const float4* __restrict__ wp, int lane) {
    const unsigned long long wq = (unsigned long long)wp;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)wq), hi = __builtin_amdgcn_readfirstlane((unsigned)(wq >> 32));
    w.res = make_rsrc((const void*)(((unsigned long long)hi << 32) | lo));
#pragma unroll
    for (int u = 0; u < RING; ++u) w.ar[u] = buf_load4(w.res, lane * 16u, (unsigned)u * 1024u);
}
template <int NT, int RING, int NSQ, class Hook>
__device__ __forceinline__ void mlp_gemm_ring(f32x16 (&acc)[NT], WRing<RING>& w, const float* __restrict__ Bs, int ld, int lane,
                                              Hook hook) {
    static_assert(NSQ >= 2 * RING, "ring deeper than the GEMM");
    const int r = lane & 31, kh = lane >> 5;
    const unsigned wvoff = lane * 16;
    const lds_float* bp = (const lds_float*)Bs + kh * ld + r;
    float b[4][NT], bn[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[j][t] = bp[j * 2 * ld + t * 32];
#pragma unroll
    for (int sq = 0; sq < NSQ; ++sq) {
        const float4 a = w.ar[sq % RING];
        if (sq + RING < NSQ) w.ar[sq % RING] = buf_load4(w.res, wvoff, (unsigned)(sq + RING) * 1024u);
        if (sq + RING == NSQ) hook();
        const lds_float* nb = bp + (sq + 1 < NSQ ? 8 * ld : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[j][t] = nb[j * 2 * ld + t * 32];
        bp = nb;
        __builtin_amdgcn_sched_barrier(0);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[j][t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) b[j][t] = bn[j][t];
    }
}

__device__ __forceinline__ int frag_row(int q, int kh) { return (q & 3) + 8 * (q >> 2) + 4 * kh; }

// The activation buffer of a training call: [7][256][ldn] fp32 images H1..H7 followed by the sign masks, unsigned short
// [7][16][ldn]: bit q of masks[l][2 w + kh][p] = (H_{l+1}[32 w + frag_row(q, kh)][p] > 0)  (sg_sdfnet_acts_floats in the header).
__host__ __device__ __forceinline__ const unsigned short* sdf_mask_base(const float* acts, long ldn) {
    return reinterpret_cast<const unsigned short*>(acts + 7L * kH * ldn);
}

struct SdfFwdArgs {
    const float* points;    // [*,3]
    long points_period;     // >0: point index = p % period (shared voxel grid); 0: p
    const float* latent;    // per-point mode: [N,L] rows, or table rows if latent_idx
    const int64_t* latent_idx;  // optional [N] row index into latent
    int L;
    const float* packed;
    SdfPackLayout lay;
    const float* zb1;  // per-shape mode: [S][256] (bias of layer 1 incl. latent part)
    const float* zb5;
    long pps;          // points per shape (per-shape mode, uniform segments)
    const int* sid;    // per-shape mode with ragged segments: shape index of every point (NULL: p / pps)
    float* out;        // [N]
    float* acts;       // optional [7][256][ldn]
    long ldn;
    long N;
    long nbig;         // workgroups [0, nbig): full tiles; the rest: kSmallTile points each
    float eps;         // NORM: LayerNorm epsilon
};

// The LayerNorm form (NORM; SDFGenerator, model/point_sdf_net.py:49-119 with hidden_channels 256, num_layers 8): the same eight
// layers with x = relu(LayerNorm(lin(x) [+ z_lin(z)])) (:104-116) instead of relu(lin(x)), no tanh at the end, and
// cat([x, pos]) (:100) where SDFNet has cat(x, input).  A point's 256 features are spread over the eight waves (32 rows each), so
// a layer's statistics are combined through LDS: every wave reduces its 32 rows of a point to (mean, sum of squared deviations)
// in registers (16 in-lane terms + one cross-half exchange), parks the pair in `red` in front of the barrier the write-back
// has anyway, and combines the eight pairs behind it (Chan's formula for equal counts) — no extra barrier, no E[x^2] - E[x]^2
// cancellation.  In training the images hold xhat = (x - mean) * rstd (the LayerNorm backward needs it where the ReLU is off
// too; the weight-gradient GEMM applies relu(gamma xhat + beta) when it loads them), the sign masks are those of the ReLU
// input, and rstd [7][ldn] follows the masks.
__host__ __device__ __forceinline__ const float* sdf_rstd_base(const float* acts, long ldn) { return acts + 7L * kH * ldn + 56L * ldn; }

template <int P, bool SHAPE_BIAS, bool TRAIN, bool NORM = false>   // TRAIN: `acts` is given (H images + sign masks are written)
__device__ __forceinline__ void sdfnet_fwd_tile(const SdfFwdArgs& a, const long p0) {
    constexpr int NT = P / 32;
    constexpr int LDX = P + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                // [256][P]
    float* Xs = Hs + kH * P;         // [KUp][LDX]
    float* red = Xs + a.lay.KUp * LDX;  // [16][P]
    float* Bl = red + 16 * P;           // [7][256]: the bias vectors (see init_acc_lds)
    float* GBs = Bl + 7 * kH;           // NORM: [8 waves][gamma 32 | beta 32] of the layer in flight

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;
    const int KU = a.lay.KU, KUp = a.lay.KUp;
    // The seven bias vectors go to LDS once per tile.  A layer used to start with four 16-byte global loads of its bias into the
    // accumulators and an s_waitcnt for them in front of its first MFMA — and vmcnt counts in issue order, so that wait also drained
    // everything older: the weight ring of the layer (started early on purpose) and, in training, the 32 activation-image stores
    // of the previous write-back (ISA listing, round 4).  From LDS the accumulators are initialised under lgkmcnt only.
    for (int e = tid; e < 7 * kH; e += 512) Bl[e] = (a.packed + a.lay.B)[e];

    // ---- stage X = [xyz | latent] feature-major ----
    for (int e = tid; e < 3 * P; e += 512) {
        const int p = e / 3, c = e - p * 3;
        const long gp = p0 + p;
        float v = 0.f;
        if (gp < a.N) {
            const long pi = a.points_period > 0 ? gp % a.points_period : gp;
            v = a.points[pi * 3 + c];
        }
        Xs[c * LDX + p] = v;
    }
    if constexpr (!SHAPE_BIAS) {
        const int L = a.L;
        for (int e = tid; e < P * L; e += 512) {
            const int p = e / L, k = e - p * L;
            const long gp = p0 + p;
            float v = 0.f;
            if (gp < a.N) {
                const long row = a.latent_idx ? (long)a.latent_idx[gp] : gp;
                v = a.latent[row * L + k];
            }
            Xs[(3 + k) * LDX + p] = v;
        }
    }
    for (int e = tid; e < (KUp - KU) * P; e += 512) {
        const int k = KU + e / P, p = e % P;
        Xs[k * LDX + p] = 0.f;
    }
    __syncthreads();

    const float* bias = a.packed + a.lay.B;
    const long shape = (SHAPE_BIAS && !a.sid) ? (p0 / a.pps) : 0;
    const float4* pk = reinterpret_cast<const float4*>(a.packed);

    f32x16 acc[NT];
    auto init_acc = [&](const float* b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float bv = b[wave * 32 + frag_row(q, kh)];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][q] = bv;
        }
    };
    auto init_acc_lds = [&](int layer) {      // the same from the LDS copy of bias vector `layer`
        const lds_float* b = (const lds_float*)Bl + layer * kH + wave * 32;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float bv = b[frag_row(q, kh)];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][q] = bv;
        }
    };
    // ragged per-shape mode: every point (= fragment column) looks its folded bias row up by its own shape index
    int psid[NT];
    if (SHAPE_BIAS && a.sid) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long gp = p0 + t * 32 + r;
            psid[t] = a.sid[gp < a.N ? gp : a.N - 1];
        }
    }
    auto init_acc_sid = [&](const float* zb) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* b = zb + (long)psid[t] * kH + wave * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = b[frag_row(q, kh)];
        }
    };
    // training: H_l also goes to `acts` — buffer stores from a scalar base (this wave's row block at the tile's first point),
    // lane offset = (4 kh) rows + point, fragment row in the scalar offset; lanes beyond N carry an out-of-range offset (dropped
    // by the hardware), so the epilogue has neither 64-bit address arithmetic nor exec-mask branches
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 32;
    unsigned astore[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        astore[t] = p0 + t * 32 + r < a.N ? (unsigned)((4L * kh * a.ldn + t * 32 + r) * 4) : kBufOutside;
    // ... and the SIGN MASK of H_l, 1 bit per element: a lane holds 16 rows (q) of one point per column tile, so it packs them
    // into one 16-bit word at masks[layer][16-row group = 2 wave + kh][point] (sg_sdfnet_mask_* in the header): 1/32 of the H
    // traffic.  The backward reads ReLU'(.) from these words instead of re-reading the fp32 images.
    unsigned mstore[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        mstore[t] = p0 + t * 32 + r < a.N ? (unsigned)(((long)kh * a.ldn + t * 32 + r) * 2) : kBufOutside;
    // (`save` stays a run-time condition even in the TRAIN instantiation: as a compile-time constant the stores lose their place
    // in the schedule, the write-back's live ranges grow and the kernel no longer fits the 128 VGPRs of two workgroups per CU)
    // NORM: this lane's LayerNorm weight (lanes 0..31) / bias (32..63) element of a layer, row wrow + r: requested in front of the
    // layer's GEMM, parked in LDS behind it (in front of the next weight ring: its wait covers only loads that were consumed)
    float gbv = 0.f;
    auto gb_load = [&](int layer) __attribute__((always_inline)) {
        if constexpr (NORM) gbv = (a.packed + (kh ? a.lay.Be : a.lay.G))[layer * kH + wave * 32 + r];
    };
    auto gb_commit = [&]() __attribute__((always_inline)) {
        if constexpr (NORM) GBs[wave * 64 + lane] = gbv;
    };
    auto writeback = [&](int layer) {  // H <- relu(acc) (NORM: relu(LayerNorm(acc))); optionally save
        float mean[NT], rstd[NT];
        if constexpr (NORM) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) s += acc[t][q];
                s += __shfl_xor(s, 32, 64);
                const float mw = s * (1.f / 32.f);
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float e = acc[t][q] - mw;
                    d = fmaf(e, e, d);
                }
                d += __shfl_xor(d, 32, 64);
                if (kh == 0) {
                    red[(2 * wave) * P + t * 32 + r] = mw;
                    red[(2 * wave + 1) * P + t * 32 + r] = d;
                }
            }
        }
        __syncthreads();
        const bool save = (NORM ? TRAIN : true) && a.acts != nullptr;
        if constexpr (NORM) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const lds_float* rp = (const lds_float*)red + t * 32 + r;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) sum += rp[(2 * w) * P];
                const float m = sum * 0.125f;
                float M2 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const float e = rp[(2 * w) * P] - m;
                    M2 += rp[(2 * w + 1) * P];
                    M2 = fmaf(32.f * e, e, M2);
                }
                mean[t] = m;
                rstd[t] = 1.f / sqrtf(M2 * (1.f / 256.f) + a.eps);
                if (TRAIN && save && wave == 0 && kh == 0 && p0 + t * 32 + r < a.N)
                    const_cast<float*>(sdf_rstd_base(a.acts, a.ldn))[(long)layer * a.ldn + p0 + t * 32 + r] = rstd[t];
            }
        }
        const __amdgpu_buffer_rsrc_t ares = make_rsrc(a.acts + ((long)layer * kH + wrow) * a.ldn + p0);
        unsigned mk[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) mk[t] = 0u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = wave * 32 + frag_row(q, kh);
            float gq = 1.f, bq = 0.f;
            if constexpr (NORM) {
                gq = GBs[wave * 64 + frag_row(q, kh)];
                bq = GBs[wave * 64 + 32 + frag_row(q, kh)];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float v, img;
                if constexpr (NORM) {
                    img = (acc[t][q] - mean[t]) * rstd[t];
                    acc[t][q] = fmaf(gq, img, bq);      // (the sign mask below is that of the ReLU input)
                    v = fmaxf(acc[t][q], 0.f);
                } else {
                    v = img = fmaxf(acc[t][q], 0.f);
                }
                Hs[row * P + t * 32 + r] = v;
                if (save)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, img), ares, (int)astore[t],
                                                          (int)(((q & 3) + 8 * (q >> 2)) * a.ldn * 4), SG_IMG_AUX);
            }
        }
        if constexpr (TRAIN) {
            if (save) {
                // sign word: mk = 2 mk + (acc > 0), rows 15 .. 0, so that bit q ends up belonging to row q — compare into vcc and
                // add-with-carry, two VALU instructions per element (the C form `mk |= v > 0 ? 1 << q : 0` took three and kept
                // 16 more values live).  The accumulators were all read by the loop above: no MFMA result hazard is left.
#pragma unroll
                for (int q = 15; q >= 0; --q)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mk[t]) : "v"(acc[t][q]) : "vcc");
                const __amdgpu_buffer_rsrc_t mres =
                    make_rsrc(sdf_mask_base(a.acts, a.ldn) + ((long)layer * 16 + (wrow >> 4)) * a.ldn + p0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)mk[t], mres, (int)mstore[t], 0, 0);
            }
        }
        __syncthreads();
    };
    auto wtile = [&](long off, int nsq) { return pk + (off >> 2) + (long)wave * nsq * 64; };

    // layer 1: K over X
    if (SHAPE_BIAS && a.sid)
        init_acc_sid(a.zb1);
    else
        SHAPE_BIAS ? init_acc(a.zb1 + shape * kH) : init_acc_lds(0);
    gb_load(0);
    mlp_gemm<NT>(acc, wtile(a.lay.F1, KUp / 8), KUp / 8, Xs, LDX, lane);
    gb_commit();
    // the weight ring of the next 256-wide layer is started before each write-back (its stores would otherwise sit in front of
    // the first weight loads in the in-order return queue, see WRing)
    WRing<4> wr;
    auto noop = []() {};
    auto next_ring = [&](long off) __attribute__((always_inline)) {
        wring_start(wr, wtile(off, kH / 8), lane);
        __builtin_amdgcn_sched_barrier(0);
    };
    next_ring(a.lay.F2);
    writeback(0);
    // layers 2..4
    const long Fnext[3] = {a.lay.F3, a.lay.F4, a.lay.F5x};
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        init_acc_lds(l + 1);
        gb_load(l + 1);
        mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
        gb_commit();
        next_ring(Fnext[l]);
        writeback(l + 1);
    }
    // layer 5: K over H (256) then X (skip connection, model/sdf_net.py:59)
    if (SHAPE_BIAS && a.sid)
        init_acc_sid(a.zb5);
    else
        SHAPE_BIAS ? init_acc(a.zb5 + shape * kH) : init_acc_lds(4);
    gb_load(4);
    mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
    mlp_gemm<NT>(acc, wtile(a.lay.F5i, KUp / 8), KUp / 8, Xs, LDX, lane);
    gb_commit();
    next_ring(a.lay.F6);
    writeback(4);
    // layers 6, 7
    init_acc_lds(5);
    gb_load(5);
    mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
    gb_commit();
    next_ring(a.lay.F7);
    writeback(5);
    init_acc_lds(6);
    gb_load(6);
    mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
    gb_commit();
    writeback(6);
    // layer 8: 256 -> 1, tanh.  The dot product is cut into sixteen 16-row groups summed in a fixed order, whatever the tile
    // size (a thread takes P / 32 groups of its point), so that a point's output does not depend on the tile it falls into.
    {
        constexpr int PARTS = 512 / P;     // threads per point
        constexpr int SUB = 16 / PARTS;    // row groups per thread
        const int p = tid % P, part = tid / P;
        const float* w8 = a.packed + a.lay.W8;
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const int g = part * SUB + u;
            float s = 0.f;
#pragma unroll
            for (int k = 16 * g; k < 16 * g + 16; ++k) s = fmaf(w8[k], Hs[k * P + p], s);
            red[g * P + p] = s;
        }
        __syncthreads();
        if (tid < P) {
            float v = bias[7 * kH];
#pragma unroll
            for (int q = 0; q < 16; ++q) v += red[q * P + tid];
            const long gp = p0 + tid;
            if (gp < a.N) a.out[gp] = NORM ? v : tanhf(v);   // (SDFGenerator ends in a plain Linear, point_sdf_net.py:106-111)
        }
    }
}

// Tile plan (see tile_plan below): workgroups [0, nbig) take P points each, the rest kSmallTile points each.  The small tiles
// are the remainder of the last round of workgroups: a 200 000-point launch is 6.1 rounds of 64-point tiles on 2 x 256 workgroup slots, and
// the 27 tiles of the seventh round would keep the whole chip waiting for a full tile time; cut into 32-point tiles they
// finish in about a third of it.  A point's arithmetic does not depend on the tile it is in (same k order in every layer, the
// last layer's dot product in fixed 16-row groups), so the plan never changes a result of the forward.
constexpr int kSmallTile = 32;
constexpr long kCUs = 256;

// (512 threads, 4 waves per SIMD = two workgroups per CU: the register budget is 128 VGPRs, stated explicitly — the kernel sat
// just below it by luck before, and one more live value silently halves the occupancy)
template <int P, bool SHAPE_BIAS, bool TRAIN, bool NORM = false>
__global__ void __launch_bounds__(512, 4) sdfnet_fwd_kernel(SdfFwdArgs a) {
    const long b = blockIdx.x;
    if (b < a.nbig)
        sdfnet_fwd_tile<P, SHAPE_BIAS, TRAIN, NORM>(a, b * P);
    else
        sdfnet_fwd_tile<kSmallTile, SHAPE_BIAS, TRAIN, NORM>(a, a.nbig * P + (b - a.nbig) * kSmallTile);
}

// ---------------------------------------------------------------------------------------------------------------------------
// PointNet.nn1 over a whole cloud, reduced to WHICH point holds each channel's maximum (model/point_sdf_net.py:14-23,40: Linear
// 4 -> 64 -> 128 -> 256 -> 512 with ReLU between, `x.max(dim=-2)`): the plain pass of the sparse-adjoint critic
// (shapegan_amd/model/point_sdf_net.py PointNet.selected_points).  One workgroup walks a tile of 32 points through the four layers
// on the fused-MLP machinery above — activations in LDS (H1 / H3 share a buffer, H2 has its own), weights streamed from their
// MFMA-packed L2-resident image — and never writes the 512-wide layer: a wave reduces its 32 x 32 fragment of it over the tile's
// points (five DPP max steps per row, the point found by a ballot on equality with the maximum: lowest point wins a tie) and the
// tile stores one (value, point) pair per channel.  Per point: 16 B in, 128 B / 32 of partials out instead of the 2 KB row of
// the last layer plus the 1.9 KB of the three hidden ones.  sg_pointnet_select merges the tiles of a cloud (segmax_merge_kernel).
// NaN: ignored by the maximum (a cloud whose channel is NaN everywhere selects its first point).
struct PnSelArgs {
    const float* x;        // [N][4]
    const float* packed;   // W1 | W2 | W3 | W4 images
    const float* b[4];
    float* pv;             // [tiles][512]
    int* pi;
    long N, pps;           // pps: points per cloud (a multiple of 32)
};
constexpr long kPnW1 = 0, kPnW2 = kPnW1 + 64 * 8, kPnW3 = kPnW2 + 128 * 64, kPnW4 = kPnW3 + 256 * 128, kPnTotal = kPnW4 + 512 * 256;

__global__ void __launch_bounds__(512, 6) pointnet_select_kernel(PnSelArgs a) {
    constexpr int P = 32, LDX = P + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const A = smem;                 // [256][P]: H1 (64 rows), later H3
    float* const Bm = A + 256 * P;         // [128][P]: H2
    float* const Xs = Bm + 128 * P;        // [8][LDX]
    float* const Bl = Xs + 8 * LDX;        // biases: 64 | 128 | 256 | 512
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;
    const long p0 = (long)blockIdx.x * P;
    if (tid < 4 * P) {
        const int pp = tid >> 2, c = tid & 3;
        Xs[c * LDX + pp] = p0 + pp < a.N ? a.x[(p0 + pp) * 4 + c] : 0.f;
    } else if (tid < 8 * P) {
        const int e = tid - 4 * P;
        Xs[(4 + (e >> 5)) * LDX + (e & 31)] = 0.f;
    }
    for (int e = tid; e < 960; e += 512) Bl[e] = e < 64 ? a.b[0][e] : e < 192 ? a.b[1][e - 64] : e < 448 ? a.b[2][e - 192] : a.b[3][e - 448];
    __syncthreads();
    const float4* pk = reinterpret_cast<const float4*>(a.packed);
    f32x16 acc[1];
    auto init = [&](int boff, int row0) __attribute__((always_inline)) {
        const lds_float* b = (const lds_float*)Bl + boff + row0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] = b[frag_row(q, kh)];
    };
    auto relu_to = [&](float* H, int row0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) H[(row0 + frag_row(q, kh)) * P + r] = fmaxf(acc[0][q], 0.f);
    };
    if (wave < 2) {       // 4 -> 64
        init(0, wave * 32);
        mlp_gemm<1>(acc, pk + (kPnW1 >> 2) + (long)wave * 1 * 64, 1, Xs, LDX, lane);
        relu_to(A, wave * 32);
    }
    __syncthreads();
    if (wave < 4) {       // 64 -> 128
        init(64, wave * 32);
        mlp_gemm<1>(acc, pk + (kPnW2 >> 2) + (long)wave * 8 * 64, 8, A, P, lane);
        relu_to(Bm, wave * 32);
    }
    __syncthreads();
    init(192, wave * 32);   // 128 -> 256
    mlp_gemm<1>(acc, pk + (kPnW3 >> 2) + (long)wave * 16 * 64, 16, Bm, P, lane);
    relu_to(A, wave * 32);
    __syncthreads();
    auto dpp_max = [&](float v, auto ctrl, auto rowmask) __attribute__((always_inline)) {
        return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xff800000u, __builtin_bit_cast(int, v), decltype(ctrl)::value,
                                                                             decltype(rowmask)::value, 0xf, false)));
    };
    const int plocal = (int)(p0 % a.pps);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {    // 256 -> 512, no activation behind it (point_sdf_net.py:22)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] = 0.f;
        const int t = pass * 8 + wave;
        mlp_gemm<1>(acc, pk + (kPnW4 >> 2) + (long)t * 32 * 64, 32, A, P, lane);
        // a wave's 32 rows end up one per lane: (maximum, point) of row L in lane L, written with v_writelane from the scalars the
        // reduction produces — two coalesced 128-byte stores per pass and wave (the first form parked every pair in LDS from inside
        // sixteen one-lane branches, a barrier and a second store behind them: 13 % of the kernel)
        float resm = 0.f;
        int resi = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float v = acc[0][q];        // (clouds are multiples of 32 points: every lane of every tile is a point)
            float m = dpp_max(v, IntTag<0xB1>(), IntTag<0xf>());
            m = dpp_max(m, IntTag<0x4E>(), IntTag<0xf>());
            m = dpp_max(m, IntTag<0x141>(), IntTag<0xf>());
            m = dpp_max(m, IntTag<0x140>(), IntTag<0xf>());
            m = dpp_max(m, IntTag<0x142>(), IntTag<0xa>());
            const int m0 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 31);
            const int m1 = __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 63);
            const unsigned lo = (unsigned)__ballot(v == __builtin_bit_cast(float, m0));          // (compares against the scalars: no
            const unsigned hi = (unsigned)(__ballot(v == __builtin_bit_cast(float, m1)) >> 32);   //  per-lane select of the half's maximum)
            const int i0 = lo ? __builtin_ctz(lo) : 0, i1 = hi ? __builtin_ctz(hi) : 0;
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(resm) : "s"(m0), "n"(frag_row(q, 0)));
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(resm) : "s"(m1), "n"(frag_row(q, 1)));
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(resi) : "s"(i0), "n"(frag_row(q, 0)));
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(resi) : "s"(i1), "n"(frag_row(q, 1)));
        }
        if (lane < 32) {
            // (the bias is the same for every point of a channel: added behind the maximum — rounding is monotonic, the value
            //  equals max(x + b) and the selected point is unchanged)
            const int c = t * 32 + lane;
            a.pv[(long)blockIdx.x * 512 + c] = resm + Bl[448 + c];
            a.pi[(long)blockIdx.x * 512 + c] = plocal + resi;
        }
    }
}

struct SdfBwdArgs {
    const float* dout;   // [N]
    const float* out;    // [N] forward output (tanh)
    const float* acts;   // [7][256][ldn]
    float* dz;           // [7][256][ldn]  dZ1..dZ7
    float* dz8;          // [N]
    float* bsum;         // optional [nblk][kPartRow], tile-major: blocks 0..6 of 256 = row sums of dZ1..dZ7 (bias-gradient partials),
                         // element 14*256 = the tile's sum of dz8; with `points` also block 7: sum_p dz8[p] H7[row][p] (w8 gradient),
                         // blocks 8+c / 11+c: sum_p dZ1 / dZ5 [row][p] * xyz_c[p] (the three point columns of dW1 / dW5)
    const float* points; // optional [*,3] (with bsum): the xyz of the points, for the extended partial sums
    long points_period;
    float* dx;           // optional: input gradient, row-major [N][dx_ld] (first KU columns written)
    long dx_ld;
    const float* packed;
    SdfPackLayout lay;
    long ldn;
    long N;
    long nbig;           // workgroups [0, nbig): P-point tiles; the rest: kSmallTile points each (sg_sdfnet_bwd_tile_start)
    long nblk;           // all workgroups = rows of bsum
};

// ---------------------------------------------------------------------------------------------------------------------------
// The backward-data chain of one tile as TWO SKEWED 32-POINT CHAINS per workgroup (round 6).
//
// dZ_l = dH_l * (H_l > 0), dH_{l-1} = W_l^T dZ_l on the matrix pipe (transposed packs); each wave owns rows [32 wave, 32 wave + 32) of
// every dH_l in MFMA fragment layout.  The points of a tile are independent all the way down the chain, so its two 32-point
// column tiles (chain A = points 0..31, chain B = points 32..63) need not move in lock-step.  Rounds 1 - 5 ran a layer as GEMM ->
// barrier -> epilogue (ReLU' select, LDS write-back, dZ image stores, DPP row sums) -> barrier with no MFMA during the epilogue
// (MFMA busy 0.77; git history has that form).  Here chain B lags half a layer behind chain A:
//
//     phase 1 of image i:   GEMM_A(i) reads G_A(i+1)   ||   epilogue_B(i+1) writes G_B(i+1)      barrier
//     phase 2 of image i:   GEMM_B(i) reads G_B(i+1)   ||   epilogue_A(i)   writes G_A(i)        barrier
//
// so every phase of every wave is a 128-MFMA GEMM with one sixteenth of the other chain's epilogue between the MFMAs of every
// second k-group.  Still two barriers per layer; a chain's tile is never written while its GEMM reads it.  The weight ring runs
// on from one GEMM into the next (chain A and chain B of an image read the same pack).  dz / dz8 / dx and the sign words keep
// their layouts; a tile is still 64 or 32 points (sg_sdfnet_bwd_tile_start).
//
// Row sums (bias-gradient partials, the point columns of dW1 / dW5): 80 - 320 DPP operations and 16 - 64 one-lane stores per wave
// and layer in the old form.  Now a thread owns 16 points of one row of a chain's LDS tile (row pitch 36 floats: the b128 reads
// of a wave's 32 rows x 2 halves hit every bank once) while that tile is the B operand of the next GEMM — in eight pieces
// between that GEMM's MFMAs —, adds chain B's share half a layer later and stores one value per row and tile.  The partial
// sums are TILE-MAJOR, [tiles][kPartRow]: a tile's 14 x 256 sums are 14 KB of contiguous stores (the old row-major
// [14*256][tiles] matrix took 3 584 scattered 4-byte writes per tile — as many memory transactions as the dZ images).
constexpr int kLdg = 36;
constexpr int kPartRow = SG_SDFNET_PARTIAL_ROW;   // 14 * 256 row sums + the tile's sum of dz8 (+ padding)
static_assert(kPartRow >= 14 * kH + 1, "partial row");
// the LayerNorm form appends 14 blocks: the LayerNorm weight gradients of images 0..6 (sum_p dY xhat), then the bias gradients (sum_p dY)
constexpr int kPartRowNorm = SG_SDFGEN_PARTIAL_ROW;
static_assert(kPartRowNorm == kPartRow + 14 * kH, "partial row of the LayerNorm form");

template <int RING, int NSQ, bool HAS_NEXT, class Slice>
__device__ __forceinline__ void chain_gemm(f32x16& acc, WRing<RING>& w, const __amdgpu_buffer_rsrc_t next,
                                           const float* __restrict__ Bs, int lane, Slice slice) {
    static_assert(NSQ % RING == 0 && NSQ >= 2 * RING, "ring and GEMM length");
    const int r = lane & 31, kh = lane >> 5;
    const unsigned wvoff = lane * 16;
    const lds_float* bp = (const lds_float*)Bs + kh * kLdg + r;
    float b[4], bn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = bp[j * 2 * kLdg];
#pragma unroll
    for (int sq = 0; sq < NSQ; ++sq) {
        const float4 a4 = w.ar[sq % RING];
        if (sq + RING < NSQ)
            w.ar[sq % RING] = buf_load4(w.res, wvoff, (unsigned)(sq + RING) * 1024u);
        else if (HAS_NEXT)
            w.ar[sq % RING] = buf_load4(next, wvoff, (unsigned)(sq + RING - NSQ) * 1024u);
        const lds_float* nb = bp + (sq + 1 < NSQ ? 8 * kLdg : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = nb[j * 2 * kLdg];
        bp = nb;
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[1], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        slice(sq);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bn[j];
    }
    if (HAS_NEXT) w.res = next;
}

#ifdef SG_ABL_NOBAR
#define SG_PHASE_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define SG_PHASE_BARRIER() __syncthreads()
#endif
// NORM (single-chain tiles only): the LayerNorm form, see sdfnet_fwd_tile.  With Y = gamma xhat + beta, H = relu(Y):
//   dY = dH * (Y > 0),  dYh = dY * gamma,  dZ = rstd * (dYh - mean_f(dYh) - xhat * mean_f(dYh * xhat))   (means over the 256 features)
// The two per-point means are combined across the eight waves through LDS like the forward's statistics: each wave reduces its
// 32 rows in registers right behind its GEMM and parks the pair in front of the phase barrier that is there anyway; the second
// half of the epilogue (behind the barrier) turns the accumulators into dZ.  The LayerNorm parameter gradients sum_p dY xhat and
// sum_p dY are sums over the POINTS of a fragment row, i.e. over the 32 lanes of a half-wave: five DPP steps per row and
// quantity, the lane that ends up with the total stores it into the tile's partial row (blocks 14 + image / 21 + image).
template <bool DUAL, bool NORM = false>
__device__ __forceinline__ void sdfnet_bwd_tile2(const SdfBwdArgs& a, const long p0) {
    static_assert(!(DUAL && NORM), "the LayerNorm form runs single-chain tiles");
    constexpr int prow = NORM ? kPartRowNorm : kPartRow;
    constexpr int NH = DUAL ? 2 : 1;
    constexpr int P = NH * 32;
    constexpr int RING = SG_BWD_RING;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const G0 = smem;                  // chain A: dZ of points p0 .. p0+31, [256][kLdg]
    float* const G1 = smem + kH * kLdg;      // chain B: points p0+32 .. p0+63
    float* const dz8s = smem + 2 * kH * kLdg;   // [64]
    float* const xs = dz8s + 64;                 // [3][64] xyz of the tile (only with a.points)
    float* const Ss = xs + 3 * 64;               // NORM: [8 waves][2][32] per-point partial means
    float* const GBs = Ss + 8 * 64;              // NORM: [8 waves][gamma 32 | beta 32] of the image in flight

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;
    const float4* pk = reinterpret_cast<const float4*>(a.packed);
    const int KUr = a.lay.KUr;
    const int nxt = KUr / 32;

    if (tid < P) {
        const long gp = p0 + tid;
        float v = 0.f;
        if (gp < a.N) {
            if constexpr (NORM) {
                v = a.dout[gp];      // (no tanh behind the last Linear)
            } else {
                const float o = a.out[gp];
                v = a.dout[gp] * (1.f - o * o);
            }
            a.dz8[gp] = v;
        }
        dz8s[tid] = v;
    }
    const bool ext = a.bsum && a.points;
    if (ext && tid >= 64 && tid < 64 + 3 * P) {
        const int e = tid - 64, c = e / P, pp = e - c * P;
        const long gp = p0 + pp;
        float v = 0.f;
        if (gp < a.N) v = a.points[(a.points_period > 0 ? gp % a.points_period : gp) * 3 + c];
        xs[c * 64 + pp] = v;
    }
    __syncthreads();
    if (a.bsum && tid < 64) {       // the tile's share of the layers2.6 bias gradient: sum of dz8 (element 14 * 256 of its partial row)
        const float t8 = sg_wave_sum(tid < P ? dz8s[tid] : 0.f);
        if (tid == 0) a.bsum[(long)blockIdx.x * prow + 14 * kH] = t8;
    }

    f32x16 acc[NH];
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 32;
    bool pok[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) pok[h] = p0 + h * 32 + r < a.N;
    const unsigned khoff = (unsigned)(4L * kh * a.ldn * 4);
    // (addresses of lanes beyond N: loads go to the tile's first point, stores out of range — see the form above)
    unsigned hload[NH], zstore[NH], mload[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        hload[h] = pok[h] ? khoff + (unsigned)(h * 32 + r) * 4u : khoff;
        zstore[h] = pok[h] ? khoff + (unsigned)(h * 32 + r) * 4u : kBufOutside;
        mload[h] = (unsigned)((long)kh * a.ldn * 2) + (pok[h] ? (unsigned)(h * 32 + r) * 2u : 0u);
    }
    auto layer_rsrc = [&](const float* image, int layer) __attribute__((always_inline)) {
        return make_rsrc(image + ((long)layer * kH + wrow) * a.ldn + p0);
    };
    auto load_mask = [&](int layer, int h) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t mres =
            make_rsrc(sdf_mask_base(a.acts, a.ldn) + ((long)layer * 16 + (wrow >> 4)) * a.ldn + p0);
        const unsigned m = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(mres, (int)mload[h], 0, 0);
#ifdef SG_ABL_NOLOAD
        return 0xffffu;
#else
        return pok[h] ? m : 0u;
#endif
    };
    // this lane's element q of a chain's tile: row wave*32 + frag_row(q, kh), point r
    lds_float* const gw[2] = {(lds_float*)G0 + (wave * 32 + 4 * kh) * kLdg + r, (lds_float*)G1 + (wave * 32 + 4 * kh) * kLdg + r};
    auto rowoff = [](int q) { return (q & 3) + 8 * (q >> 2); };

    // ---- the w8 gradient partial and dZ7 = (w8 (x) dz8) * (H7 > 0), both chains at once (once per tile) ----
    auto dpp_add = [&](float v, auto ctrl, auto rowmask) __attribute__((always_inline)) {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value,
                                                                         decltype(rowmask)::value, 0xf, false));
    };
    auto half_sum = [&](float v) __attribute__((always_inline)) {
        v = dpp_add(v, IntTag<0xB1>(), IntTag<0xf>());
        v = dpp_add(v, IntTag<0x4E>(), IntTag<0xf>());
        v = dpp_add(v, IntTag<0x141>(), IntTag<0xf>());
        v = dpp_add(v, IntTag<0x140>(), IntTag<0xf>());
        return dpp_add(v, IntTag<0x142>(), IntTag<0xa>());
    };
    WRing<RING> wr;
    auto pack_rsrc = [&](long toff) __attribute__((always_inline)) {
        const unsigned long long wq = (unsigned long long)(pk + (toff >> 2) + (long)wave * (kH / 8) * 64);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)wq), hi = __builtin_amdgcn_readfirstlane((unsigned)(wq >> 32));
        return make_rsrc((const void*)(((unsigned long long)hi << 32) | lo));
    };
    // ---- NORM: operands of an image's epilogue (requested in front of its GEMM) and the two halves of the epilogue ----
    float xh[16], gbv = 0.f, rsi = 0.f;
    unsigned mk[NH];
    auto norm_loads = [&](int i) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xres = layer_rsrc(a.acts, i);
#pragma unroll
        for (int q = 0; q < 16; ++q) xh[q] = buf_load(xres, hload[0], (unsigned)(rowoff(q) * a.ldn * 4));
        gbv = (a.packed + (kh ? a.lay.Be : a.lay.G))[i * kH + wave * 32 + r];
        rsi = pok[0] ? sdf_rstd_base(a.acts, a.ldn)[(long)i * a.ldn + p0 + r] : 0.f;
    };
    const bool nsums = a.bsum != nullptr;
    auto norm_part1 = [&](int i) __attribute__((always_inline)) {
        GBs[wave * 64 + lane] = gbv;     // (read back by this wave only: LDS operations of a wave execute in order)
        const unsigned bsoff = r == 31 ? (unsigned)(4 * kh * 4) : kBufOutside;
        const __amdgpu_buffer_rsrc_t gres = make_rsrc(a.bsum + (long)blockIdx.x * prow + kPartRow + i * kH + wrow);
        const __amdgpu_buffer_rsrc_t bres = make_rsrc(a.bsum + (long)blockIdx.x * prow + kPartRow + (7 + i) * kH + wrow);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float gq = GBs[wave * 64 + frag_row(q, kh)];
            const float dy = ((mk[0] >> q) & 1u) ? acc[0][q] : 0.f;
            if (nsums) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, half_sum(dy * xh[q])), gres, (int)bsoff,
                                                      (int)(rowoff(q) * 4), 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, half_sum(dy)), bres, (int)bsoff,
                                                      (int)(rowoff(q) * 4), 0);
            }
            const float dyh = dy * gq;
            acc[0][q] = dyh;
            s1 += dyh;
            s2 = fmaf(dyh, xh[q], s2);
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (kh == 0) {
            Ss[(2 * wave) * 32 + r] = s1;
            Ss[(2 * wave + 1) * 32 + r] = s2;
        }
    };
    auto norm_part2 = [&](const __amdgpu_buffer_rsrc_t zres) __attribute__((always_inline)) {
        const lds_float* sp = (const lds_float*)Ss + r;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            m1 += sp[(2 * w) * 32];
            m2 += sp[(2 * w + 1) * 32];
        }
        m1 *= (1.f / kH);
        m2 *= (1.f / kH);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float g = rsi * ((acc[0][q] - m1) - xh[q] * m2);
            gw[0][rowoff(q) * kLdg] = g;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), zres, (int)zstore[0], (int)(rowoff(q) * a.ldn * 4), SG_IMG_AUX);
        }
    };
    if constexpr (NORM) {
        // image 6: dH7 = w8 (x) dz8, the w8 gradient partial sum_p H7 dz8 with H7 = relu(gamma xhat + beta)
        norm_loads(6);
        mk[0] = load_mask(6, 0);
        wring_start(wr, pk + (a.lay.T7 >> 2) + (long)wave * (kH / 8) * 64, lane);
        __builtin_amdgcn_sched_barrier(0);
        const float* w8 = a.packed + a.lay.W8;
        const unsigned bsoff = r == 31 ? (unsigned)(4 * kh * 4) : kBufOutside;
        const __amdgpu_buffer_rsrc_t w8res = make_rsrc(a.bsum + (long)blockIdx.x * prow + (ext ? 7 : 0) * kH + wrow);
        GBs[wave * 64 + lane] = gbv;
        const float d8 = dz8s[r];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = frag_row(q, kh);
            const float gq = GBs[wave * 64 + row], bq = GBs[wave * 64 + 32 + row];
            const float h = pok[0] ? fmaxf(fmaf(gq, xh[q], bq), 0.f) : 0.f;
            if (ext)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, half_sum(h * d8)), w8res, (int)bsoff,
                                                      (int)(rowoff(q) * 4), 0);
            acc[0][q] = pok[0] ? w8[wave * 32 + row] * d8 : 0.f;
        }
        norm_part1(6);
        __syncthreads();
        norm_part2(layer_rsrc(a.dz, 6));
    } else {
        float hf[16][NH];
        const __amdgpu_buffer_rsrc_t hres = layer_rsrc(a.acts, 6);
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int h = 0; h < NH; ++h) hf[q][h] = buf_load(hres, hload[h], (unsigned)(rowoff(q) * a.ldn * 4));
        // (the first weight ring behind the H7 loads: its wait then covers them)
        wring_start(wr, pk + (a.lay.T7 >> 2) + (long)wave * (kH / 8) * 64, lane);
        __builtin_amdgcn_sched_barrier(0);
        const float* w8 = a.packed + a.lay.W8;
        const unsigned bsoff = r == 31 ? (unsigned)(4 * kh * 4) : kBufOutside;
        const __amdgpu_buffer_rsrc_t w8res = make_rsrc(a.bsum + (long)blockIdx.x * prow + (ext ? 7 : 0) * kH + wrow);
        const __amdgpu_buffer_rsrc_t zres = layer_rsrc(a.dz, 6);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float wv = w8[wave * 32 + frag_row(q, kh)];
            float s8 = 0.f;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float d8 = dz8s[h * 32 + r];
                const bool on = pok[h] && hf[q][h] > 0.f;
                const float g = on ? wv * d8 : 0.f;
                s8 = fmaf(pok[h] ? hf[q][h] : 0.f, d8, s8);
                gw[h][rowoff(q) * kLdg] = g;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), zres, (int)zstore[h],
                                                      (int)(rowoff(q) * a.ldn * 4), SG_IMG_AUX);
            }
            if (ext)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, half_sum(s8)), w8res, (int)bsoff,
                                                      (int)(rowoff(q) * 4), 0);
        }
    }
    __syncthreads();

    // ---- row sums of a chain's tile from LDS: thread (row = tid / 2, 16-point half = tid % 2) ----
    const int srow = tid >> 1, ssub = tid & 1;
    float rs[4];
    auto rowsum_pass = [&](const float* G, int h, bool xc) __attribute__((always_inline)) {
        const lds_f32x4* gp = (const lds_f32x4*)((const lds_float*)G + srow * kLdg + ssub * 16);
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gp[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[0] += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        if (xc) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const lds_f32x4* xp = (const lds_f32x4*)((const lds_float*)xs + c * 64 + h * 32 + ssub * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 x = xp[i];
                    rs[1 + c] = fmaf(v[i].x, x.x, fmaf(v[i].y, x.y, fmaf(v[i].z, x.z, fmaf(v[i].w, x.w, rs[1 + c]))));
                }
            }
        }
    };
    // the same pass cut into eight pieces that ride between the MFMAs of a GEMM (k-groups 1, 3, ... 15: a quarter of the row is
    // requested in one piece and added in the next), so that nothing but the first B fragments stands between a barrier and the
    // first MFMA of a phase (the pass in front of the GEMM cost 64 us of 1304 at 200 000 points)
    f32x4 rq[4];
    auto rowsum_slice = [&](const float* G, int h, bool xc, int sq) __attribute__((always_inline)) {
        if ((sq & 1) == 0 || sq >= 16) return;
        const int i = sq >> 2;
        if ((sq & 3) == 1) {
            rq[0] = ((const lds_f32x4*)((const lds_float*)G + srow * kLdg + ssub * 16))[i];
            if (xc) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rq[1 + c] = ((const lds_f32x4*)((const lds_float*)xs + c * 64 + h * 32 + ssub * 16))[i];
            }
        } else {
            rs[0] += (rq[0].x + rq[0].y) + (rq[0].z + rq[0].w);
            if (xc) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    rs[1 + c] = fmaf(rq[0].x, rq[1 + c].x, fmaf(rq[0].y, rq[1 + c].y, fmaf(rq[0].z, rq[1 + c].z, fmaf(rq[0].w, rq[1 + c].w, rs[1 + c]))));
            }
        }
    };
    auto rowsum_clear = [&]() __attribute__((always_inline)) { rs[0] = rs[1] = rs[2] = rs[3] = 0.f; };
    // both 16-point halves of a row -> partial block `blk` (and the three point-column blocks xblk..xblk+2 behind it)
    auto rowsum_store = [&](int blk, int xblk) __attribute__((always_inline)) {
        const int n = xblk >= 0 ? 4 : 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < n) {
                const float t = dpp_add(rs[i], IntTag<0xB1>(), IntTag<0xf>());
                if (ssub == 0) a.bsum[(long)blockIdx.x * prow + (i == 0 ? blk : xblk + i - 1) * kH + srow] = t;
            }
        }
    };

    // ---- epilogue of chain h for image `layer`, element q: dZ = acc * ReLU'(H) -> LDS tile, dz image ----
    auto epi_q = [&](int h, int q, const __amdgpu_buffer_rsrc_t zres) __attribute__((always_inline)) {
        const float g = ((mk[h] >> q) & 1u) ? acc[h][q] : 0.f;
        gw[h][rowoff(q) * kLdg] = g;
#ifndef SG_ABL_NOSTORE
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), zres, (int)zstore[h], (int)(rowoff(q) * a.ldn * 4), SG_IMG_AUX);
#endif
    };
    auto zero = [&](int h) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[h][q] = 0.f;
    };
#ifdef SG_ABL_NOSUM
    const bool sums = false;
#else
    const bool sums = a.bsum != nullptr;
#endif

    // image i: GEMMs with pack `toff`; `tnext`: the pack of image i - 1 (HAS_NEXT); xprev: the point-column block of image i + 1
    auto step = [&](auto itag, long toff, long tnext, auto has_next, auto xprev_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(itag)::value;
        constexpr bool HAS_NEXT = decltype(has_next)::value != 0;
        constexpr int XPREV = decltype(xprev_tag)::value;
        const bool xc = XPREV >= 0 && ext;
        // ---- phase 1: GEMM_A(i) || epilogue_B(i + 1) ----
        if (sums) rowsum_clear();
        const unsigned mka = load_mask(i, 0);
        if constexpr (NORM) norm_loads(i);
        zero(0);
        if (DUAL) {
            const __amdgpu_buffer_rsrc_t zprev = layer_rsrc(a.dz, i + 1);
            chain_gemm<RING, kH / 8, true>(acc[0], wr, pack_rsrc(toff), G0, lane, [&](int sq) __attribute__((always_inline)) {
                if (i < 5 && (sq & 1) == 0) epi_q(NH - 1, sq >> 1, zprev);
                if (sums) rowsum_slice(G0, 0, xc, sq);
            });
        } else {
            chain_gemm<RING, kH / 8, HAS_NEXT>(acc[0], wr, pack_rsrc(HAS_NEXT ? tnext : toff), G0, lane,
                                              [&](int sq) __attribute__((always_inline)) {
                                                  if (sums) rowsum_slice(G0, 0, xc, sq);
                                              });
        }
        mk[0] = mka;
        if constexpr (NORM) norm_part1(i);
        SG_PHASE_BARRIER();
        const __amdgpu_buffer_rsrc_t zres = layer_rsrc(a.dz, i);
        if (DUAL) {
            // ---- phase 2: GEMM_B(i) || epilogue_A(i) ----
            const unsigned mkb = load_mask(i, NH - 1);
            zero(NH - 1);
            chain_gemm<RING, kH / 8, HAS_NEXT>(acc[NH - 1], wr, pack_rsrc(HAS_NEXT ? tnext : toff), G1, lane,
                                              [&](int sq) __attribute__((always_inline)) {
                                                  if ((sq & 1) == 0) epi_q(0, sq >> 1, zres);
                                                  if (sums) {
                                                      rowsum_slice(G1, 1, xc, sq);
                                                      if (sq == 17) rowsum_store(i + 1, xc ? XPREV : -1);
                                                  }
                                              });
            mk[NH - 1] = mkb;
        } else {
            if (sums) rowsum_store(i + 1, xc ? XPREV : -1);
            if constexpr (NORM) {
                norm_part2(zres);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) epi_q(0, q, zres);
            }
        }
        __syncthreads();
    };
    step(IntTag<5>(), a.lay.T7, a.lay.T6, IntTag<1>(), IntTag<-1>());
    step(IntTag<4>(), a.lay.T6, a.lay.T5x, IntTag<1>(), IntTag<-1>());
    step(IntTag<3>(), a.lay.T5x, a.lay.T4, IntTag<1>(), IntTag<11>());     // (sums of image 4 = dZ5: + point columns of dW5)
    step(IntTag<2>(), a.lay.T4, a.lay.T3, IntTag<1>(), IntTag<-1>());
    step(IntTag<1>(), a.lay.T3, a.lay.T2, IntTag<1>(), IntTag<-1>());
    step(IntTag<0>(), a.lay.T2, -1, IntTag<0>(), IntTag<-1>());
    // ---- tail: epilogue_B(0), then the sums of image 0 = dZ1 (+ point columns of dW1) ----
    if (DUAL) {
        const __amdgpu_buffer_rsrc_t zres = layer_rsrc(a.dz, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) epi_q(NH - 1, q, zres);
        __syncthreads();
    }
    if (sums) {
        rowsum_clear();
        rowsum_pass(G0, 0, ext);
        if (DUAL) rowsum_pass(G1, 1, ext);
        rowsum_store(0, ext ? 8 : -1);
    }

    // ---- input gradient dX = W1^T dZ1 + W5[:,256:]^T dZ5 (rows = input features, 32-row tiles round-robin over waves) ----
    // The chains' tiles hold dZ1 now; the dZ5 tile is read back from the dz image this workgroup wrote (L2-hot).
    if (a.dx) {
        constexpr int HE = kH * P / 512;
        constexpr int RSTEP = 512 / P;
        const int mrow = tid / P, mp = tid % P;
        // (scalar row base + one 32-bit lane offset per thread: points beyond N carry an out-of-range offset and read 0)
        const unsigned rvoff = p0 + mp < a.N ? (unsigned)(((long)mrow * a.ldn + mp) * 4) : kBufOutside;
        float* const gdst = (mp < 32 ? G0 : G1) + mrow * kLdg + (mp & 31);
        auto reload = [&](int layer) {
            __syncthreads();
#pragma unroll 8
            for (int i = 0; i < HE; ++i) {
                const __amdgpu_buffer_rsrc_t zr = make_rsrc(a.dz + ((long)layer * kH + i * RSTEP) * a.ldn + p0);
                gdst[i * RSTEP * kLdg] = buf_load(zr, rvoff, 0);
            }
            __syncthreads();
        };
        for (int pass = 0; pass * 8 < nxt; ++pass) {
            const int xt = pass * 8 + wave;
            const bool mine = xt < nxt;
            if (pass > 0) reload(0);
#pragma unroll
            for (int h = 0; h < NH; ++h) zero(h);
            if (mine) mlp_gemm<NH, 4, kH * kLdg>(acc, pk + (a.lay.T1 >> 2) + (long)xt * (kH / 8) * 64, kH / 8, G0, kLdg, lane);
            reload(4);
            if (mine) {
                mlp_gemm<NH, 4, kH * kLdg>(acc, pk + (a.lay.T5i >> 2) + (long)xt * (kH / 8) * 64, kH / 8, G0, kLdg, lane);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const long gp = p0 + h * 32 + r;
                    if (gp < a.N) {
                        float* d = a.dx + gp * a.dx_ld;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int row = xt * 32 + frag_row(q, kh);
                            if (row < a.lay.KU) d[row] = acc[h][q];
                        }
                    }
                }
            }
        }
    }
}

template <int P>
__global__ void __launch_bounds__(512, 4) sdfnet_bwd_kernel(SdfBwdArgs a) {
    static_assert(P == 64, "two 32-point chains per workgroup");
    const long b = blockIdx.x;
    if (b < a.nbig)
        sdfnet_bwd_tile2<true>(a, b * P);
    else
        sdfnet_bwd_tile2<false>(a, a.nbig * P + (b - a.nbig) * kSmallTile);
}

__global__ void __launch_bounds__(512, 4) sdfgen_bwd_kernel(SdfBwdArgs a) {
    sdfnet_bwd_tile2<false, true>(a, (long)blockIdx.x * kSmallTile);
}

// ---- everything that is derived from the tile partials of one backward, in ONE launch (round 6; it replaces sdfnet_segsum +
// two rowsum_multi + the two-stage sum of dz8 = five launches of the auto-decoder step) ------------------------------------------
//   column sums over the tiles:  group g = 0..6 -> bias gradient of dZ_{g+1}; 7 -> w8 gradient; 8..10 / 11..13 -> point columns
//                                of dW1 / dW5 (written with the matrices' row strides); 14 -> b8 gradient (one value)
//   segment sums:                t[which][row][s] = sum over the points of segment s of dZ_layer[row][.] (which 0: dZ1, 1: dZ5),
//                                the per-shape sums behind the latent-table gradient and the latent columns of dW1 / dW5: the
//                                interior tiles of a shape's run come from the partials, only the (at most two) tiles its ends
//                                cut are read from the images.
// Column sums are two-level and deterministic: block (g, split) adds its range of tiles (thread = (tile phase, column): coalesced
// 1 KB rows), parks 256 values in the workspace and takes a ticket; the block that draws the last ticket of its group adds the
// splits IN SPLIT ORDER and writes the destination — the arrival order decides who does the addition, never its order.  The
// tickets are left at zero for the next launch.
struct SdfFinishArgs {
    const float* part;     // [nblk][part_row]
    int part_row;          // kPartRow, or kPartRowNorm (the LayerNorm form: 14 more groups, 15..21 LayerNorm weight, 22..28 bias gradients)
    int nws;               // groups per split in the workspace: 15 or 29
    const float* dz;       // [7][256][ldn]
    long ldn, nblk, nbig;
    int ngroups;           // 15 (29: LayerNorm form), or 7 + 1 (bias groups and the dz8 group) without the extended blocks
    int extended;
    int nsplit;
    long tiles_per_split;
    float* dst[29];
    long dst_stride[29];
    double* ws;            // [nsplit][nws * 256]
    unsigned* tickets;     // [16] ([32]: LayerNorm form), zero
    const int64_t* seg_off;
    long S;
    float* t1;
    float* t5;
    long ncol_blocks;      // ngroups * nsplit
};
__global__ void __launch_bounds__(1024) sdfnet_finish_kernel(SdfFinishArgs a) {
    __shared__ double red[4][kH];     // (sums over tiles in double: a sum that cancels to rounding level must not pick up the
    __shared__ float edge[kH];        //  noise of a thousand fp32 partial additions on top of the tiles' own)
    __shared__ int last;
    const int tid = threadIdx.x, phase = tid >> 8, col = tid & 255;
    if ((long)blockIdx.x < a.ncol_blocks) {
        const int gi = blockIdx.x / a.nsplit, sp = blockIdx.x - gi * a.nsplit;
        const int g = (a.extended || gi < 7) ? gi : 14;     // (without the extended blocks: groups 0..6 and 14)
        const long t0 = sp * a.tiles_per_split;
        const long t1 = t0 + a.tiles_per_split < a.nblk ? t0 + a.tiles_per_split : a.nblk;
        const bool on = g != 14 || col == 0;
        const float* src = a.part + (g < 14 ? g * kH + col : g == 14 ? 14 * kH : kPartRow + (g - 15) * kH + col);
        const long kPR = a.part_row;
        double acc = 0;
        if (on) {
            long t = t0 + phase;
            for (; t + 28 < t1; t += 32) {       // eight loads in flight per thread
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(t + 4 * u) * kPR];
                acc += (((double)v[0] + v[1]) + ((double)v[2] + v[3])) + (((double)v[4] + v[5]) + ((double)v[6] + v[7]));
            }
            for (; t < t1; t += 4) acc += src[t * kPR];
        }
        red[phase][col] = acc;
        __syncthreads();
        if (phase == 0) a.ws[((long)sp * a.nws + g) * kH + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);   // (a double)
        __syncthreads();      // (every wave's stores have been acknowledged by this XCD's L2: s_waitcnt vmcnt(0) in front of the barrier)
        if (tid == 0) {
            // ONE release per block: a device-scope fence writes this XCD's L2 back (the eight L2s are not coherent with each
            // other) — executed by all 16 waves of all 960 blocks of a 200 000-point launch it was 300 us of a 330 us kernel
            __threadfence();
            last = atomicAdd(&a.tickets[g], 1u) == (unsigned)(a.nsplit - 1);
        }
        __syncthreads();
        if (last) {
            // (the loads below are device-scope atomic loads: they are served coherently, no acquire fence needed)
            // the splits in split order, whoever adds them: thread (phase, col) takes splits phase, phase + 4, ... (all its loads
            // in flight together), the four phase sums are combined in a fixed order
            double s = 0;
            if (on) {
                int q = phase;
                for (; q + 12 < a.nsplit; q += 16) {
                    const double v0 = __hip_atomic_load(&a.ws[((long)q * a.nws + g) * kH + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double v1 = __hip_atomic_load(&a.ws[((long)(q + 4) * a.nws + g) * kH + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double v2 = __hip_atomic_load(&a.ws[((long)(q + 8) * a.nws + g) * kH + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double v3 = __hip_atomic_load(&a.ws[((long)(q + 12) * a.nws + g) * kH + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s += (v0 + v1) + (v2 + v3);
                }
                for (; q < a.nsplit; q += 4)
                    s += __hip_atomic_load(&a.ws[((long)q * a.nws + g) * kH + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();      // (red[] of the first level has been read by phase 0 above: behind the barriers in between)
            red[phase][col] = s;
            __syncthreads();
            if (phase == 0 && on) a.dst[g][(long)col * a.dst_stride[g]] = (float)((red[0][col] + red[1][col]) + (red[2][col] + red[3][col]));
            if (tid == 0) a.tickets[g] = 0u;
        }
        return;
    }
    // ---- segment sums: block = (segment, which) ----
    const long sb = (long)blockIdx.x - a.ncol_blocks;
    const long sg = sb >> 1;
    const int which = (int)(sb & 1);
    const int layer = which ? 4 : 0;
    const long beg = a.seg_off[sg], end = a.seg_off[sg + 1];
    const long edge0 = a.nbig * 64;   // first point of the small tiles
    // first tile that starts at or after beg / last tile boundary at or before end
    const long ta = beg <= edge0 ? (beg + 63) / 64 : a.nbig + (beg - edge0 + kSmallTile - 1) / kSmallTile;
    const long tb = end <= edge0 ? end / 64 : a.nbig + (end - edge0) / kSmallTile;
    auto start = [&](long t) { return t <= a.nbig ? t * 64 : edge0 + (t - a.nbig) * kSmallTile; };
    const bool interior = ta < tb;
    const long head = interior ? start(ta) : end, tail = interior ? start(tb) : end;
    // interior tiles from the partials: thread = (tile phase, row)
    double acc = 0;
    if (interior) {
        const float* src = a.part + layer * kH + col;
        const long kPR = a.part_row;
        long t = ta + phase;
        for (; t + 12 < tb; t += 16) {
            const float v0 = src[t * kPR], v1 = src[(t + 4) * kPR], v2 = src[(t + 8) * kPR], v3 = src[(t + 12) * kPR];
            acc += ((double)v0 + v1) + ((double)v2 + v3);
        }
        for (; t < tb; t += 4) acc += src[t * kPR];
    }
    red[phase][col] = acc;
    // the cut tiles from the image: wave w takes rows 16 w .. 16 w + 15, lanes walk the points ([beg, head) and [tail, end) are
    // shorter than a tile each; a run without an interior tile is walked whole)
    const int lane = tid & 63, wave = tid >> 6;
    // ([beg, head) is shorter than two tiles even for a run without an interior tile, [tail, end) shorter than one: three
    // loads per lane and row, all 48 of a wave's 16 rows requested before the first sum)
    float ev[16];
    const long x0 = beg + lane, x1 = beg + 64 + lane, x2 = tail + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float* p = a.dz + ((long)layer * kH + wave * 16 + j) * a.ldn;
        const float v0 = x0 < head ? p[x0] : 0.f, v1 = x1 < head ? p[x1] : 0.f, v2 = x2 < end ? p[x2] : 0.f;
        ev[j] = (v0 + v1) + v2;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float e = sg_wave_sum(ev[j]);
        if (lane == 0) edge[wave * 16 + j] = e;
    }
    __syncthreads();
    if (phase == 0) (which ? a.t5 : a.t1)[(long)col * a.S + sg] = (float)(((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + (double)edge[col]);
}

// ---- the per-shape latent fold itself, both matrices in one launch (round 6; it was two sg_gemm calls = 2 - 4 launches) ----
//   zb1[s][o] = b1[o] + sum_k z[s][k] W1[o][3 + k]        zb5[s][o] = b5[o] + sum_k z[s][k] W5[o][259 + k]
// A [S, L] x [L, 256] product of a few MFLOP: workgroup = 16 shapes x 16 outputs of one matrix, K in chunks of 64 through LDS,
// accumulated in DOUBLE and rounded once — the value does not depend on a tile or split-K plan, and it is the correctly rounded
// sum up to the last bit.
__device__ __forceinline__ void shape_fold_body(const float* __restrict__ z, int S, int L, const float* __restrict__ W1,
                                                const float* __restrict__ W5, const float* __restrict__ b1,
                                                const float* __restrict__ b5, float* __restrict__ zb1, float* __restrict__ zb5,
                                                const int fx, const int fy, const int which) {
    __shared__ float zs[16][65];
    __shared__ float ws[16][65];
    const float* W = which ? W5 + kH + 3 : W1 + 3;
    const long ldw = which ? kH + 3 + L : 3 + L;
    const int s0 = fx * 16, o0 = fy * 16;
    const int tid = threadIdx.x, sl = tid >> 4, ol = tid & 15;
    double acc = 0;
    for (int k0 = 0; k0 < L; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 6) + 4 * i, k = tid & 63;
            zs[r][k] = (s0 + r < S && k0 + k < L) ? z[(long)(s0 + r) * L + k0 + k] : 0.f;
            ws[r][k] = k0 + k < L ? W[(long)(o0 + r) * ldw + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 16
        for (int k = 0; k < 64; ++k) acc = fma((double)zs[sl][k], (double)ws[ol][k], acc);
    }
    if (s0 + sl < S) ((which ? zb5 : zb1) + (long)(s0 + sl) * kH)[o0 + ol] = (float)(acc + (double)(which ? b5 : b1)[o0 + ol]);
}
__global__ void __launch_bounds__(256) shape_fold_kernel(const float* __restrict__ z, int S, int L, const float* __restrict__ W1,
                                                         const float* __restrict__ W5, const float* __restrict__ b1,
                                                         const float* __restrict__ b5, float* __restrict__ zb1,
                                                         float* __restrict__ zb5) {
    shape_fold_body(z, S, L, W1, W5, b1, b5, zb1, zb5, blockIdx.x, blockIdx.y, blockIdx.z);
}
// the weight pack (kin_used = 3) and the latent fold of the same parameters in ONE launch — both read only the parameter tensors and
// the latents, and every step of the shape-sorted auto-decoder needs both behind its optimizer step (train_sdf_autodecoder.py:80-91):
// workgroups [0, 64 (n + 1)) pack, the rest fold
__global__ void __launch_bounds__(256) pack_fold_kernel(PackDescs descs, float* __restrict__ dst, const float* __restrict__ z, int S, int L,
                                                        const float* __restrict__ W1, const float* __restrict__ W5,
                                                        const float* __restrict__ b1, const float* __restrict__ b5,
                                                        float* __restrict__ zb1, float* __restrict__ zb5) {
    const unsigned npack = 64u * (unsigned)(descs.n + 1);
    if (blockIdx.x < npack) {
        pack_mfma_a_body(descs, dst, blockIdx.x & 63u, blockIdx.x >> 6, 64u);
    } else {
        const unsigned f = blockIdx.x - npack, sx = (unsigned)((S + 15) / 16);
        shape_fold_body(z, S, L, W1, W5, b1, b5, zb1, zb5, (int)(f % sx), (int)((f / sx) & 15u), (int)(f / (sx * 16u)));
    }
}

// ---- backward of the per-shape latent fold (the latent columns of layers1.0 / layers2.0 enter the forward as per-shape bias rows) ----
// Backward of the fold from the per-shape sums t1 / t5 [256][S] of dZ1 / dZ5:
//   blocks [0, 256):     row o of the latent columns  dW1[o][3 + k] = sum_s t1[o][s] z[s][k],  dW5[o][259 + k] = sum_s t5[o][s] z[s][k]
//   blocks [256, 256+S): latent gradient row          gz[s][k] = sum_o t1[o][s] W1[o][3 + k] + t5[o][s] W5[o][259 + k]
// thread = k (strided by 256 for L > 256); the broadcast operand (a t row / column) sits in LDS.
__global__ void __launch_bounds__(1024) shape_bias_bwd_kernel(const float* __restrict__ t1, const float* __restrict__ t5, int S,
                                                              const float* __restrict__ z, int L, const float* __restrict__ W1,
                                                              const float* __restrict__ W5, float* __restrict__ dW1,
                                                              float* __restrict__ dW5, float* __restrict__ gz,
                                                              const float* __restrict__ reg_weight, float reg_scale) {
    extern __shared__ float sh[];   // rows: 2 x S, columns: 2 x 256
    __shared__ double red[2][4][kH];
    const int tid = threadIdx.x, part = tid >> 8, kk = tid & 255;
    const int ld1 = 3 + L, ld5 = kH + 3 + L;
    // Both halves walk a short reduction (S shapes / 256 rows) with two global loads per term.  Written as a plain loop the loads
    // were issued one at a time behind the dependent double-precision FMA chain — 256 exposed L2 latencies, 163 us for 17 MFLOP
    // (18 % of the 20 000-point auto-decoder step).  Round 3: terms fetched in batches of 16 (all loads of a batch in flight
    // together): 26 us.  Round 6: the reduction range is cut into four contiguous quarters (thread = (quarter, k)), whose double
    // sums are added in quarter order — four times fewer dependent batches per thread: the 256-row half went from 16 to 4.
    constexpr int kB = 16;
    if (blockIdx.x < kH) {
        if (!dW1) return;
        const int o = blockIdx.x;
        for (int s = tid; s < S; s += 1024) {
            sh[s] = t1[(long)o * S + s];
            sh[S + s] = t5[(long)o * S + s];
        }
        __syncthreads();
        const int chunk = (S + 3) / 4, sa = part * chunk, sb = sa + chunk < S ? sa + chunk : S;
        for (int k0 = 0; k0 < L; k0 += 256) {
            const int k = k0 + kk;
            double a = 0, b = 0;    // few terms, ill-conditioned sums (per-shape sums of either sign): accumulate in double
            if (k < L) {
                for (int s0 = sa; s0 < sb; s0 += kB) {
                    float zv[kB];
#pragma unroll
                    for (int i = 0; i < kB; ++i) zv[i] = s0 + i < sb ? z[(long)(s0 + i) * L + k] : 0.f;
#pragma unroll
                    for (int i = 0; i < kB; ++i) {
                        if (s0 + i < sb) {
                            a = fma((double)sh[s0 + i], (double)zv[i], a);
                            b = fma((double)sh[S + s0 + i], (double)zv[i], b);
                        }
                    }
                }
            }
            red[0][part][kk] = a;
            red[1][part][kk] = b;
            __syncthreads();
            if (part == 0 && k < L) {
                dW1[(long)o * ld1 + 3 + k] = (float)(((red[0][0][kk] + red[0][1][kk]) + red[0][2][kk]) + red[0][3][kk]);
                dW5[(long)o * ld5 + kH + 3 + k] = (float)(((red[1][0][kk] + red[1][1][kk]) + red[1][2][kk]) + red[1][3][kk]);
            }
            __syncthreads();
        }
    } else {
        if (!gz) return;
        const int s = blockIdx.x - kH;
        if (tid < kH) {
            sh[tid] = t1[(long)tid * S + s];
            sh[kH + tid] = t5[(long)tid * S + s];
        }
        __syncthreads();
        const int oa = part * (kH / 4);
        for (int k0 = 0; k0 < L; k0 += 256) {
            const int k = k0 + kk;
            double a = 0, b = 0;
            if (k < L) {
                for (int o0 = oa; o0 < oa + kH / 4; o0 += kB) {
                    float w1v[kB], w5v[kB];
#pragma unroll
                    for (int i = 0; i < kB; ++i) {
                        w1v[i] = W1[(long)(o0 + i) * ld1 + 3 + k];
                        w5v[i] = W5[(long)(o0 + i) * ld5 + kH + 3 + k];
                    }
#pragma unroll
                    for (int i = 0; i < kB; ++i) {
                        a = fma((double)sh[o0 + i], (double)w1v[i], a);
                        b = fma((double)sh[kH + o0 + i], (double)w5v[i], b);
                    }
                }
            }
            red[0][part][kk] = a + b;
            __syncthreads();
            if (part == 0 && k < L) {
                float v = (float)(((red[0][0][kk] + red[0][1][kk]) + red[0][2][kk]) + red[0][3][kk]);
                // + the gradient of a quadratic latent regulariser sum_s w_s |z_s|^2 * reg_scale / 2 (train_sdf_autodecoder.py:88):
                // the expression of deepsdf_bwd_kernel, added in fp32 like the autograd accumulation it replaces (bit-identical)
                if (reg_scale != 0.f) v += ((reg_weight ? reg_weight[s] : 1.f) * reg_scale) * z[(long)s * L + k];
                gz[(long)s * L + k] = v;
            }
            __syncthreads();
        }
    }
}

// `nbig` tiles of P points followed by `nsmall` tiles of kSmallTile points: when the last round of `slots` concurrently resident
// workgroups would be at most three quarters full, its points are cut into small tiles.  A pure function of N (the slot counts
// are those of the full 256-CU device, not queried): the partial-sum layout of the backward, and with it the summation order of
// everything derived from it, is the same on every device and partition mode.
struct TilePlan {
    long nbig, nsmall;
};
// `per_cu` workgroups share a CU (slots = per_cu x 256).  The last, partly filled round of `rem` tiles:
//   * per_cu == 2 and slots/2 < rem <= 3 slots/4: rem big tiles would leave some CUs with two of them (128 points) next to CUs
//     with one; slots/2 big tiles (one per CU — the dispatcher places workgroups breadth-first) + the rest of the points as
//     small tiles gives every CU at most 64 + 32 points.  This is the reference's own 20 000-point batch (313 tiles);
//   * otherwise, when it is not the only round and at most 3/4 full: all of it as small tiles.
static TilePlan tile_plan(long N, int P, long slots) {
    const long tiles = (N + P - 1) / P;
    const long full = tiles / slots * slots, rem = tiles - full;
    if (slots == 2 * kCUs && 2 * rem > slots && 4 * rem <= 3 * slots) {
        const long nbig = full + slots / 2;
        return TilePlan{nbig, (N - nbig * P + kSmallTile - 1) / kSmallTile};
    }
    if (full == 0 || rem == 0 || 4 * rem > 3 * slots) return TilePlan{tiles, 0};
    return TilePlan{full, (N - full * P + kSmallTile - 1) / kSmallTile};
}
constexpr int kFinishMaxSplit = 32;
constexpr long kFwdSlots = kCUs;       // one workgroup per CU (LDS)
constexpr long kBwdSlots = 2 * kCUs;   // two per CU

static size_t fwd_lds_bytes(int P, int KUp, bool norm = false) {
    return ((size_t)kH * P + (size_t)KUp * (P + 1) + 16 * P + 7 * kH + (norm ? 8 * 64 : 0)) * sizeof(float);
}
static size_t bwd_lds_bytes(int P, int KUr, bool dx) { return ((size_t)2 * kH * kLdg + 4 * 64) * sizeof(float); }   // two chains [256][kLdg] + dz8 + xyz
static size_t bwd_lds_bytes_norm() { return ((size_t)2 * kH * kLdg + 4 * 64 + 16 * 64) * sizeof(float); }   // + per-point means + gamma / beta

template <class K>
static int set_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes);
        if (e != hipSuccess) return -1;
    }
    return 0;
}

}  // namespace sg

using namespace sg;

extern "C" int sg_segmax_merge_partials(const float* pv, const int* pi, float* out, int* idx, long B, int nchunk, int C, hipStream_t stream);

extern "C" {

size_t sg_sdfnet_packed_floats(int kin_used) { return (size_t)make_layout(kin_used).total; }

// floats of the activation buffer of a training call with leading dimension ldn: the seven fp32 images [7][256][ldn] plus the
// sign masks (unsigned short [7][16][ldn] = 56 ldn floats)
size_t sg_sdfnet_acts_floats(long ldn) { return (size_t)7 * kH * ldn + (size_t)56 * ldn; }

// params: host array of 16 device pointers in state_dict order
//   layers1.{0,2,4,6}.{weight,bias}, layers2.{0,2,4,6}.{weight,bias}   (model/sdf_net.py:26-53)
struct FoldJob {       // sdf_pack with the latent fold in the same launch (NULL z: pack only)
    const float* z;
    long nshapes;
    float *zb1, *zb5;
};
static int sdf_pack(const float* const* params, const float* const* norm_params, int latent, int kin_used, float* packed,
                    hipStream_t stream, const FoldJob* fold = nullptr) {
    const SdfPackLayout L = make_layout(kin_used);
    const int KIN = 3 + latent;
    const float *W1 = params[0], *W2 = params[2], *W3 = params[4], *W4 = params[6];
    const float *W5 = params[8], *W6 = params[10], *W7 = params[12], *W8 = params[14];
    PackDescs D;
    int n = 0;
    auto add = [&](const float* src, long rs, long cs, int R, int C, int Rpad, int Cpad, long off) {
        D.d[n++] = PackDesc{src, rs, cs, R, C, Rpad / 32, Cpad / 8, off};
    };
    const int ld5 = kH + KIN;
    add(W1, KIN, 1, kH, L.KU, kH, L.KUp, L.F1);
    add(W2, kH, 1, kH, kH, kH, kH, L.F2);
    add(W3, kH, 1, kH, kH, kH, kH, L.F3);
    add(W4, kH, 1, kH, kH, kH, kH, L.F4);
    add(W5, ld5, 1, kH, kH, kH, kH, L.F5x);
    add(W5 + kH, ld5, 1, kH, L.KU, kH, L.KUp, L.F5i);
    add(W6, kH, 1, kH, kH, kH, kH, L.F6);
    add(W7, kH, 1, kH, kH, kH, kH, L.F7);
    add(W1, 1, KIN, L.KU, kH, L.KUr, kH, L.T1);
    add(W2, 1, kH, kH, kH, kH, kH, L.T2);
    add(W3, 1, kH, kH, kH, kH, kH, L.T3);
    add(W4, 1, kH, kH, kH, kH, kH, L.T4);
    add(W5, 1, ld5, kH, kH, kH, kH, L.T5x);
    add(W5 + kH, 1, ld5, L.KU, kH, L.KUr, kH, L.T5i);
    add(W6, 1, kH, kH, kH, kH, kH, L.T6);
    add(W7, 1, kH, kH, kH, kH, kH, L.T7);
    D.n = n;
    for (int l = 0; l < 8; ++l) D.v.b[l] = params[2 * l + 1];
    D.v.w8 = W8;
    D.v.dst_b = L.B;
    D.v.dst_w8 = L.W8;
    for (int l = 0; l < 7; ++l) {
        D.v.g[l] = norm_params ? norm_params[2 * l] : nullptr;
        D.v.be[l] = norm_params ? norm_params[2 * l + 1] : nullptr;
    }
    D.v.dst_g = L.G;
    D.v.dst_be = L.Be;
    if (fold) {
        const unsigned blocks = 64u * (unsigned)(n + 1) + (unsigned)((fold->nshapes + 15) / 16) * 16u * 2u;
        hipLaunchKernelGGL(pack_fold_kernel, dim3(blocks), dim3(256), 0, stream, D, packed, fold->z, (int)fold->nshapes, latent, W1, W5,
                           params[1], params[9], fold->zb1, fold->zb5);
    } else {
        hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(64, n + 1), dim3(256), 0, stream, D, packed);
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_sdfnet_pack(const float* const* params, int latent, int kin_used, float* packed, hipStream_t stream) {
    SG_CHECK_ARG(params && packed && latent >= 0 && (kin_used == 3 || kin_used == 3 + latent));
    return sdf_pack(params, nullptr, latent, kin_used, packed, stream);
}

// sg_sdfnet_pack(params, latent, 3, packed) and sg_sdfnet_shape_bias(z, nshapes, latent, W1, b1, W5, b5, zb1, zb5) in ONE launch
// (per-shape mode: both are needed behind every optimizer step of the shape-sorted auto-decoder, train_sdf_autodecoder.py:80-91).
int sg_sdfnet_pack_shape_bias(const float* const* params, int latent, float* packed, const float* z, long nshapes, float* zb1,
                              float* zb5, hipStream_t stream) {
    SG_CHECK_ARG(params && packed && latent > 0 && z && zb1 && zb5 && nshapes > 0);
    const FoldJob job{z, nshapes, zb1, zb5};
    return sdf_pack(params, nullptr, latent, 3, packed, stream, &job);
}

// The SDFGenerator form (model/point_sdf_net.py:49-119, hidden_channels 256, num_layers 8).  params: the 16 tensors
// lins.{0..7}.{weight,bias} ([256,3], [256,256] x 3, [256,259], [256,256] x 2, [1,256]: the shapes of an SDFNet without latent
// columns); norm_params: norms.{0..6}.{weight,bias}.  Same packed size as sg_sdfnet_packed_floats(3).
int sg_sdfgen_pack(const float* const* params, const float* const* norm_params, float* packed, hipStream_t stream) {
    SG_CHECK_ARG(params && norm_params && packed);
    for (int i = 0; i < 16; ++i) SG_CHECK_ARG(params[i] != nullptr);
    for (int i = 0; i < 14; ++i) SG_CHECK_ARG(norm_params[i] != nullptr);
    return sdf_pack(params, norm_params, 0, 3, packed, stream);
}

// float offsets of the LayerNorm weight (which = 0) / bias (1) vectors [7][256] inside a sg_sdfgen_pack image (operands of
// sg_gemm_nt_batched_lnrelu)
long sg_sdfgen_packed_norm_offset(int which) {
    const SdfPackLayout L = make_layout(3);
    return which ? L.Be : L.G;
}

// floats of the activation buffer of a training call of the LayerNorm form: the xhat images, the sign masks and rstd [7][ldn]
size_t sg_sdfgen_acts_floats(long ldn) { return (size_t)7 * kH * ldn + (size_t)56 * ldn + (size_t)7 * ldn; }

// Forward of the LayerNorm form: out[p] = SDFGenerator(pos[p], z[shape of p]) with the latent entering as the per-shape rows
// zb1 = z_lin1(z) + lins.0.bias, zb5 = z_lin2(z) + lins.4.bias ([S,256] each; point_sdf_net.py:104-111).  Shapes are runs of
// points_per_shape points (a multiple of 128, or >= N) or named per point by shape_index[N].  acts (optional, training):
// sg_sdfgen_acts_floats(ldn) floats.
int sg_sdfgen_fwd(const float* points, const float* packed, const float* zb1, const float* zb5, long points_per_shape,
                  const int* shape_index, float eps, float* out, float* acts, long ldn, long N, hipStream_t stream) {
    SG_CHECK_ARG(points && packed && zb1 && zb5 && out && N > 0 && eps > 0.f);
    SG_CHECK_ARG(shape_index || (points_per_shape > 0 && (points_per_shape % 128 == 0 || points_per_shape >= N)));
    SdfFwdArgs a;
    a.points = points;
    a.points_period = 0;
    a.latent = nullptr;
    a.latent_idx = nullptr;
    a.L = 0;
    a.packed = packed;
    a.lay = make_layout(3);
    a.zb1 = zb1;
    a.zb5 = zb5;
    a.pps = points_per_shape;
    a.sid = shape_index;
    a.out = out;
    a.acts = acts;
    a.ldn = ldn;
    a.N = N;
    a.eps = eps;
    if (acts) SG_CHECK_ARG(ldn >= N);
    if (acts && ldn > (1L << 24)) SG_FAIL(SG_ERR_ARG, "sg_sdfgen_fwd: at most 16 777 216 points per training call (ldn = %ld)", ldn);
    const size_t lds = fwd_lds_bytes(64, a.lay.KUp, true);
    const TilePlan tp = tile_plan(N, 64, 2 * kFwdSlots);
    a.nbig = tp.nbig;
    const dim3 grid((unsigned)(tp.nbig + tp.nsmall));
    if (acts) {
        if (set_lds(sdfnet_fwd_kernel<64, true, true, true>, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfgen_fwd: cannot reserve %zu B LDS", lds);
        hipLaunchKernelGGL((sdfnet_fwd_kernel<64, true, true, true>), grid, dim3(512), lds, stream, a);
    } else {
        if (set_lds(sdfnet_fwd_kernel<64, true, false, true>, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfgen_fwd: cannot reserve %zu B LDS", lds);
        hipLaunchKernelGGL((sdfnet_fwd_kernel<64, true, false, true>), grid, dim3(512), lds, stream, a);
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// Forward.  per-point latent mode: zb1 == zb5 == NULL, latent = [N,L] rows (or table + latent_idx), packed built
// with kin_used = 3+L.  per-shape mode: zb1/zb5 = [S,256] folded biases, packed built with kin_used = 3,
// points_per_shape a multiple of 128 (or >= N for a single shape) — or shape_index[N] (int32) for ragged segments
// (points sorted by shape are not required for correctness, only for cache locality of the bias rows).
int sg_sdfnet_fwd(const float* points, long points_period, const float* latent, const int64_t* latent_idx, int latent_size,
                  const float* packed, int kin_used, const float* zb1, const float* zb5, long points_per_shape,
                  const int* shape_index, float* out, float* acts, long ldn, long N, hipStream_t stream) {
    SG_CHECK_ARG(points && packed && out && N > 0);
    SdfFwdArgs a;
    a.points = points;
    a.points_period = points_period;
    a.latent = latent;
    a.latent_idx = latent_idx;
    a.L = latent_size;
    a.packed = packed;
    a.lay = make_layout(kin_used);
    a.zb1 = zb1;
    a.zb5 = zb5;
    a.pps = points_per_shape;
    a.sid = shape_index;
    a.out = out;
    a.acts = acts;
    a.ldn = ldn;
    a.N = N;
    a.eps = 0.f;
    if (acts) SG_CHECK_ARG(ldn >= N);
    if (acts && ldn > (1L << 24)) SG_FAIL(SG_ERR_ARG, "sg_sdfnet_fwd: at most 16 777 216 points per training call (ldn = %ld)", ldn);
    const bool shape_bias = zb1 != nullptr;
    if (shape_bias) {
        SG_CHECK_ARG(zb5 && kin_used == 3);
        SG_CHECK_ARG(shape_index || (points_per_shape > 0 && (points_per_shape % 128 == 0 || points_per_shape >= N)));   // (tiles must not straddle shapes)
        // 64-point tiles: 68 KB of LDS and < 128 VGPRs put two workgroups on a CU, so that one's barriers / write-back overlap
        // the other's MFMA phases (measured against 128-point tiles, one workgroup per CU: 8 x 32^3 inference 1.687 -> 1.654 ms,
        // 16 x 64^3 26.0 -> 25.0 ms, the training forward at 200 000 points 1.31 -> 1.24 ms)
        const size_t lds = fwd_lds_bytes(64, a.lay.KUp);
        const TilePlan tp = tile_plan(N, 64, 2 * kFwdSlots);
        a.nbig = tp.nbig;
        const dim3 grid((unsigned)(tp.nbig + tp.nsmall));
#ifdef SG_SDF_NO_MASK
        if (false) {
#else
        if (acts) {
#endif
            if (set_lds(sdfnet_fwd_kernel<64, true, true>, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfnet_fwd: cannot reserve %zu B LDS", lds);
            hipLaunchKernelGGL((sdfnet_fwd_kernel<64, true, true>), grid, dim3(512), lds, stream, a);
        } else {
            if (set_lds(sdfnet_fwd_kernel<64, true, false>, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfnet_fwd: cannot reserve %zu B LDS", lds);
            hipLaunchKernelGGL((sdfnet_fwd_kernel<64, true, false>), grid, dim3(512), lds, stream, a);
        }
    } else {
        SG_CHECK_ARG(latent && kin_used == 3 + latent_size);
        const size_t lds = fwd_lds_bytes(64, a.lay.KUp);
        if (lds > 160 * 1024) SG_FAIL(SG_ERR_ARG, "sg_sdfnet_fwd: latent size %d needs %zu B LDS (> 160 KiB)", latent_size, lds);
        const TilePlan tp = tile_plan(N, 64, kFwdSlots);
        a.nbig = tp.nbig;
        const dim3 grid((unsigned)(tp.nbig + tp.nsmall));
#ifdef SG_SDF_NO_MASK
        if (false) {
#else
        if (acts) {
#endif
            if (set_lds(sdfnet_fwd_kernel<64, false, true>, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfnet_fwd: cannot reserve %zu B LDS", lds);
            hipLaunchKernelGGL((sdfnet_fwd_kernel<64, false, true>), grid, dim3(512), lds, stream, a);
        } else {
            if (set_lds(sdfnet_fwd_kernel<64, false, false>, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfnet_fwd: cannot reserve %zu B LDS", lds);
            hipLaunchKernelGGL((sdfnet_fwd_kernel<64, false, false>), grid, dim3(512), lds, stream, a);
        }
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// Per-shape mode: the latent fold zb1 = b1 + z W1[:, 3:]^T, zb5 = b5 + z W5[:, 259:]^T ([nshapes][256] each) from the latents
// z [nshapes][latent] and the full layers1.0 / layers2.0 matrices, one launch.
int sg_sdfnet_shape_bias(const float* z, long nshapes, int latent, const float* W1, const float* b1, const float* W5, const float* b5,
                         float* zb1, float* zb5, hipStream_t stream) {
    SG_CHECK_ARG(z && W1 && b1 && W5 && b5 && zb1 && zb5 && nshapes > 0 && latent > 0);
    hipLaunchKernelGGL(shape_fold_kernel, dim3((unsigned)((nshapes + 15) / 16), 16, 2), dim3(256), 0, stream, z, (int)nshapes, latent,
                       W1, W5, b1, b5, zb1, zb5);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// Per-shape mode: the backward of the latent fold (zb1 = b1 + z W1[:, 3:]^T, zb5 = b5 + z W5[:, 259:]^T) from the per-shape sums
// t1 / t5 [256][S] of dZ1 / dZ5 (sg_sdfnet_segsum or sg_rowsum): the latent columns of dW1 / dW5 (written in place, row strides of
// the full matrices; pass NULL to skip) and the latent gradient gz [S][L] (NULL to skip).
int sg_sdfnet_shape_bias_bwd(const float* t1, const float* t5, long nshapes, const float* z, int latent, const float* W1,
                             const float* W5, float* dW1, float* dW5, float* gz, const float* reg_weight, float reg_scale,
                             hipStream_t stream) {
    SG_CHECK_ARG(t1 && t5 && z && W1 && W5 && nshapes > 0 && latent > 0 && (dW1 == nullptr) == (dW5 == nullptr));
    SG_CHECK_ARG(nshapes <= 6144);   // a t row of every shape in 48 KB of LDS
    const size_t lds = (size_t)2 * (nshapes > kH ? nshapes : kH) * sizeof(float);
    hipLaunchKernelGGL(shape_bias_bwd_kernel, dim3((unsigned)(kH + nshapes)), dim3(1024), lds, stream, t1, t5, (int)nshapes, z,
                       latent, W1, W5, dW1, dW5, gz, reg_weight, reg_scale);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// Backward-data.  Writes dz8[N], dz[7][256][ldn], (if bias_partials != NULL) the per-workgroup row sums of dZ1..dZ7
// as [7*256][sg_sdfnet_bwd_blocks(N)] (sum each row for the bias gradients) and (if dx != NULL) the input gradient rows dx[N][dx_ld]
// (kin_used columns: d/dpoints (3) then d/dlatent (L) in per-point mode, d/dpoints only in per-shape mode).
long sg_sdfnet_bwd_blocks(long N) {
    const TilePlan tp = tile_plan(N, SG_BWD_TILE, kBwdSlots);
    return tp.nbig + tp.nsmall;
}
long sg_sdfnet_bwd_tile_start(long N, long t) {
    const TilePlan tp = tile_plan(N, SG_BWD_TILE, kBwdSlots);
    const long p = t <= tp.nbig ? t * 64 : tp.nbig * 64 + (t - tp.nbig) * kSmallTile;
    return p < N ? p : N;
}

int sg_sdfnet_bwd(const float* dout, const float* out, const float* acts, float* dz, float* dz8, float* bias_partials,
                  const float* points, long points_period, float* dx, long dx_ld, const float* packed, int kin_used, long ldn,
                  long N, hipStream_t stream) {
    SG_CHECK_ARG(dout && out && acts && dz && dz8 && packed && N > 0 && ldn >= N);
    SdfBwdArgs a;
    a.points = points;
    a.points_period = points_period;
    a.dout = dout;
    a.out = out;
    a.acts = acts;
    a.dz = dz;
    a.dz8 = dz8;
    a.bsum = bias_partials;
    a.dx = dx;
    a.dx_ld = dx_ld;
    a.packed = packed;
    a.lay = make_layout(kin_used);
    a.ldn = ldn;
    a.N = N;
    if (dx) SG_CHECK_ARG(dx_ld >= kin_used);
    if (ldn > (1L << 24)) SG_FAIL(SG_ERR_ARG, "sg_sdfnet_bwd: at most 16 777 216 points per call (ldn = %ld)", ldn);
    // 64-point tiles for both input modes: 64.3 KB LDS / <= 128 VGPRs put two workgroups on a CU, whose phases
    // interleave (measured: 2.5 ms vs 3.1 ms for 128-point tiles on 200 000 points)
    {
        const size_t lds = bwd_lds_bytes(SG_BWD_TILE, a.lay.KUr, dx != nullptr);
        if (set_lds(sdfnet_bwd_kernel<SG_BWD_TILE>, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfnet_bwd: cannot reserve %zu B LDS", lds);
        const TilePlan tp = tile_plan(N, SG_BWD_TILE, kBwdSlots);
        a.nbig = tp.nbig;
        a.nblk = tp.nbig + tp.nsmall;
        hipLaunchKernelGGL((sdfnet_bwd_kernel<SG_BWD_TILE>), dim3((unsigned)(tp.nbig + tp.nsmall)), dim3(512), lds, stream, a);
    }
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// The sums derived from one sg_sdfnet_bwd call's partials, one launch (see sdfnet_finish_kernel): bias gradients of the seven
// hidden layers (bias_grads[0..6], 256 floats each), and with `extended` (the call was given `points`) the layers2.6 weight
// gradient w8_grad[256] and the three point columns of dW1 / dW5 (w1_cols / w5_cols: element (row, c) at [row * ld + c]);
// b8_grad[1] = sum of dz8; optionally (seg_off != NULL) the per-segment sums t1 / t5 [256][nseg] of dZ1 / dZ5.
size_t sg_sdfnet_bwd_finish_workspace_bytes(long N) { return (size_t)kFinishMaxSplit * 15 * kH * sizeof(double); }
size_t sg_sdfgen_bwd_finish_workspace_bytes(long N) { return (size_t)kFinishMaxSplit * 29 * kH * sizeof(double); }

static int sdf_finish(const char* who, bool norm, const float* dz, const float* partials, long ldn, long N, int extended,
                      float* const* bias_grads, float* w8_grad, float* b8_grad, float* w1_cols, long w1_ld, float* w5_cols,
                      long w5_ld, float* const* norm_grads, const int64_t* seg_off, long nseg, float* t1, float* t5, void* workspace,
                      size_t workspace_bytes, unsigned* tickets, hipStream_t stream) {
    const TilePlan tp = norm ? TilePlan{0, (N + kSmallTile - 1) / kSmallTile} : tile_plan(N, SG_BWD_TILE, kBwdSlots);
    SdfFinishArgs a;
    a.part = partials;
    a.part_row = norm ? kPartRowNorm : kPartRow;
    a.nws = norm ? 29 : 15;
    a.dz = dz;
    a.ldn = ldn;
    a.nblk = tp.nbig + tp.nsmall;
    a.nbig = tp.nbig;
    a.extended = extended ? 1 : 0;
    a.ngroups = bias_grads ? (norm ? 29 : extended ? 15 : 8) : 0;
    // a split adds about 64 tiles (16 per thread, eight loads in flight), at most kFinishMaxSplit splits: every block ends in
    // a device-scope release (an L2 write-back), so few, longer blocks
    long nsplit = (a.nblk + 63) / 64;
    if (nsplit > kFinishMaxSplit) nsplit = kFinishMaxSplit;
    if (nsplit < 1) nsplit = 1;
    a.nsplit = (int)nsplit;
    a.tiles_per_split = (a.nblk + nsplit - 1) / nsplit;
    if (workspace_bytes < (size_t)nsplit * a.nws * kH * sizeof(double)) SG_FAIL(SG_ERR_WORKSPACE, "%s: workspace too small", who);
    for (int g = 0; g < 29; ++g) {
        a.dst[g] = nullptr;
        a.dst_stride[g] = 1;
    }
    if (bias_grads) {
        for (int g = 0; g < 7; ++g) {
            if (!bias_grads[g]) SG_FAIL(SG_ERR_ARG, "%s: bias_grads[%d] is NULL", who, g);
            a.dst[g] = bias_grads[g];
        }
        a.dst[14] = b8_grad;
        if (extended) {
            a.dst[7] = w8_grad;
            for (int c = 0; c < 3; ++c) {
                a.dst[8 + c] = w1_cols + c;
                a.dst_stride[8 + c] = w1_ld;
                a.dst[11 + c] = w5_cols + c;
                a.dst_stride[11 + c] = w5_ld;
            }
        }
        if (norm)
            for (int g = 0; g < 14; ++g) {
                if (!norm_grads[g]) SG_FAIL(SG_ERR_ARG, "%s: norm_grads[%d] is NULL", who, g);
                a.dst[15 + g] = norm_grads[g];
            }
    }
    a.ws = static_cast<double*>(workspace);
    a.tickets = tickets;
    a.seg_off = seg_off;
    a.S = nseg;
    a.t1 = t1;
    a.t5 = t5;
    a.ncol_blocks = (long)a.ngroups * a.nsplit;
    const long blocks = a.ncol_blocks + 2 * nseg;
    if (blocks == 0) return SG_OK;
    hipLaunchKernelGGL(sdfnet_finish_kernel, dim3((unsigned)blocks), dim3(1024), 0, stream, a);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_sdfnet_bwd_finish(const float* dz, const float* partials, long ldn, long N, int extended, float* const* bias_grads,
                         float* w8_grad, float* b8_grad, float* w1_cols, long w1_ld, float* w5_cols, long w5_ld,
                         const int64_t* seg_off, long nseg, float* t1, float* t5, void* workspace, size_t workspace_bytes,
                         unsigned* tickets, hipStream_t stream) {
    SG_CHECK_ARG(dz && partials && N > 0 && ldn >= N && workspace && tickets && nseg >= 0);
    SG_CHECK_ARG(!bias_grads || b8_grad);
    SG_CHECK_ARG(!extended || !bias_grads || (w8_grad && w1_cols && w5_cols));
    SG_CHECK_ARG(nseg == 0 || (seg_off && t1 && t5));
    return sdf_finish("sg_sdfnet_bwd_finish", false, dz, partials, ldn, N, extended, bias_grads, w8_grad, b8_grad, w1_cols, w1_ld,
                      w5_cols, w5_ld, nullptr, seg_off, nseg, t1, t5, workspace, workspace_bytes, tickets, stream);
}

// ---- the LayerNorm form (SDFGenerator, model/point_sdf_net.py:49-119) --------------------------------------------------------
// Backward-data: dz8 = dout (no tanh), dZ_l through LayerNorm + ReLU; 32-point tiles, partials [sg_sdfgen_bwd_blocks(N)]
// [SG_SDFGEN_PARTIAL_ROW] (the SDFNet row with `points`, then the 7 + 7 LayerNorm weight / bias gradient blocks).
long sg_sdfgen_bwd_blocks(long N) { return (N + kSmallTile - 1) / kSmallTile; }

int sg_sdfgen_bwd(const float* dout, const float* acts, float* dz, float* dz8, float* partials, const float* points,
                  const float* packed, long ldn, long N, hipStream_t stream) {
    SG_CHECK_ARG(dout && acts && dz && dz8 && partials && points && packed && N > 0 && ldn >= N);
    if (ldn > (1L << 24)) SG_FAIL(SG_ERR_ARG, "sg_sdfgen_bwd: at most 16 777 216 points per call (ldn = %ld)", ldn);
    SdfBwdArgs a;
    a.points = points;
    a.points_period = 0;
    a.dout = dout;
    a.out = nullptr;
    a.acts = acts;
    a.dz = dz;
    a.dz8 = dz8;
    a.bsum = partials;
    a.dx = nullptr;
    a.dx_ld = 0;
    a.packed = packed;
    a.lay = make_layout(3);
    a.ldn = ldn;
    a.N = N;
    a.nbig = 0;
    a.nblk = sg_sdfgen_bwd_blocks(N);
    const size_t lds = bwd_lds_bytes_norm();
    if (set_lds(sdfgen_bwd_kernel, lds)) SG_FAIL(SG_ERR_HIP, "sg_sdfgen_bwd: cannot reserve %zu B LDS", lds);
    hipLaunchKernelGGL(sdfgen_bwd_kernel, dim3((unsigned)a.nblk), dim3(512), lds, stream, a);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// The sums derived from one sg_sdfgen_bwd call's partials (one launch): bias_grads[0..6] = row sums of dZ1..dZ7, w8 / b8
// gradients, the point columns of dW1 / dW5 (element (row, c) at [row * ld + c]), norm_grads[0..6] = LayerNorm weight gradients
// of norms.0..6, norm_grads[7..13] = their bias gradients; optionally the per-segment sums t1 / t5 [256][nseg] of dZ1 / dZ5 (the
// gradients of the per-shape rows zb1 / zb5, transposed).  tickets: 32 words, zero.
int sg_sdfgen_bwd_finish(const float* dz, const float* partials, long ldn, long N, float* const* bias_grads, float* w8_grad,
                         float* b8_grad, float* w1_cols, long w1_ld, float* w5_cols, long w5_ld, float* const* norm_grads,
                         const int64_t* seg_off, long nseg, float* t1, float* t5, void* workspace, size_t workspace_bytes,
                         unsigned* tickets, hipStream_t stream) {
    SG_CHECK_ARG(dz && partials && N > 0 && ldn >= N && workspace && tickets && nseg >= 0);
    SG_CHECK_ARG(bias_grads && norm_grads && w8_grad && b8_grad && w1_cols && w5_cols);
    SG_CHECK_ARG(nseg == 0 || (seg_off && t1 && t5));
    return sdf_finish("sg_sdfgen_bwd_finish", true, dz, partials, ldn, N, 1, bias_grads, w8_grad, b8_grad, w1_cols, w1_ld, w5_cols,
                      w5_ld, norm_grads, seg_off, nseg, t1, t5, workspace, workspace_bytes, tickets, stream);
}


// ---- PointNet.nn1 + max as one selection pass (see pointnet_select_kernel) -----------------------------------------------------
size_t sg_pointnet_packed_floats(void) { return (size_t)kPnTotal; }

// params: nn1.{0,2,4,6}.weight ([64,4], [128,64], [256,128], [512,256])
int sg_pointnet_pack(const float* const* weights, float* packed, hipStream_t stream) {
    SG_CHECK_ARG(weights && packed && weights[0] && weights[1] && weights[2] && weights[3]);
    PackDescs D;
    D.d[0] = PackDesc{weights[0], 4, 1, 64, 4, 2, 1, kPnW1};
    D.d[1] = PackDesc{weights[1], 64, 1, 128, 64, 4, 8, kPnW2};
    D.d[2] = PackDesc{weights[2], 128, 1, 256, 128, 8, 16, kPnW3};
    D.d[3] = PackDesc{weights[3], 256, 1, 512, 256, 16, 32, kPnW4};
    D.n = 4;
    D.v.g[0] = nullptr;
    hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(64, 4), dim3(256), 0, stream, D, packed);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

size_t sg_pointnet_select_workspace_bytes(long B, long P) { return (size_t)B * ((P + 31) / 32) * 512 * 8; }

// x [B][P][4] (P a multiple of 32) -> out [B][512] = max over the cloud of nn1(x), idx [B][512] = the point that holds it
// (int32, lowest point on a tie); biases: nn1.{0,2,4,6}.bias.
int sg_pointnet_select(const float* x, const float* packed, const float* const* biases, long B, long P, float* out, int* idx,
                       void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(x && packed && biases && out && idx && B > 0 && P > 0 && P % 32 == 0 && B * P < (1L << 31));
    for (int i = 0; i < 4; ++i) SG_CHECK_ARG(biases[i] != nullptr);
    if (!workspace || workspace_bytes < sg_pointnet_select_workspace_bytes(B, P) || ((uintptr_t)workspace & 15) != 0)
        SG_FAIL(SG_ERR_WORKSPACE, "sg_pointnet_select: workspace too small or not 16-byte aligned");
    const long tiles = B * (P / 32);
    PnSelArgs a;
    a.x = x;
    a.packed = packed;
    for (int i = 0; i < 4; ++i) a.b[i] = biases[i];
    a.pv = static_cast<float*>(workspace);
    a.pi = reinterpret_cast<int*>(a.pv + tiles * 512);
    a.N = B * P;
    a.pps = P;
    const size_t lds = ((size_t)256 * 32 + 128 * 32 + 8 * 33 + 960) * sizeof(float);
    if (set_lds(pointnet_select_kernel, lds)) SG_FAIL(SG_ERR_HIP, "sg_pointnet_select: cannot reserve %zu B LDS", lds);
    hipLaunchKernelGGL(pointnet_select_kernel, dim3((unsigned)tiles), dim3(512), lds, stream, a);
    SG_CHECK_LAUNCH();
    return sg_segmax_merge_partials(a.pv, a.pi, out, idx, B, (int)(P / 32), 512, stream);
}

}  // extern "C"
