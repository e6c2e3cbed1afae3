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
#define SG_BWD_TILE 64   // (the tile layout of the partial sums is part of the ABI: include/shapegan_hip.h)

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
struct PackDescs {
    PackDesc d[16];
    int n;
};

// dst[((t*nsq + sq)*64 + lane)*4 + j] = M(t*32 + (lane&31), (sq*4 + j)*2 + (lane>>5))
__global__ void __launch_bounds__(256) pack_mfma_a_kernel(PackDescs descs, float* __restrict__ dst) {
    const PackDesc d = descs.d[blockIdx.y];
    const long total = (long)d.ntiles * d.nsq * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
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

__global__ void __launch_bounds__(256) copy_vectors_kernel(const float* b1, const float* b2, const float* b3,
                                                           const float* b4, const float* b5, const float* b6,
                                                           const float* b7, const float* b8, const float* w8,
                                                           float* __restrict__ dst_b, float* __restrict__ dst_w8) {
    const int t = threadIdx.x;
    const float* bs[7] = {b1, b2, b3, b4, b5, b6, b7};
#pragma unroll
    for (int l = 0; l < 7; ++l) dst_b[l * kH + t] = bs[l][t];
    dst_b[7 * kH + t] = (t == 0) ? b8[0] : 0.f;
    dst_w8[t] = w8[t];
}

struct SdfPackLayout {
    int KU, KUp, KUr;
    long F1, F2, F3, F4, F5x, F5i, F6, F7;  // forward packs: A(i=out, k=in)
    long T1, T2, T3, T4, T5x, T5i, T6, T7;  // transposed packs: A(i=in, k=out)
    long W8, B;                              // w8[256], b[8][256]
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
    L.total = o;
    return L;
}

// acc[t] += A_tile(32 x K) * B(K x [t*32, t*32+32)) ; wp = this wave's packed A rows (wave-uniform), Bs = LDS [K][ld].
// Software-pipelined like the conv halo kernels: A fragments come through a 4-deep ring of buffer loads (scalar base,
// fixed lane offset, k-group in the scalar offset), the B fragments of k-group sq+1 are read from LDS (immediate offsets
// from one running address) while the 4*NT MFMAs of group sq run; sched_barriers keep the loads where they are issued.
// One VALU instruction (the LDS address step) per 4*NT MFMAs.
template <int NT, int RING = 4>
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
        for (int t = 0; t < NT; ++t) b[j][t] = bp[j * 2 * ld + t * 32];
    auto group = [&](float4 a, int sq) __attribute__((always_inline)) {
        // B of the next group (the last group re-reads its own: no branch)
        const lds_float* nb = bp + (sq < last ? 8 * ld : 0);
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
};

template <int P, bool SHAPE_BIAS, bool TRAIN>   // TRAIN: `acts` is given (H images + sign masks are written)
__device__ __forceinline__ void sdfnet_fwd_tile(const SdfFwdArgs& a, const long p0) {
    constexpr int NT = P / 32;
    constexpr int LDX = P + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                // [256][P]
    float* Xs = Hs + kH * P;         // [KUp][LDX]
    float* red = Xs + a.lay.KUp * LDX;  // [16][P]
    float* Bl = red + 16 * P;           // [7][256]: the bias vectors (see init_acc_lds)

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
    auto writeback = [&](int layer) {  // H <- relu(acc); optionally save
        __syncthreads();
        const __amdgpu_buffer_rsrc_t ares = make_rsrc(a.acts + ((long)layer * kH + wrow) * a.ldn + p0);
        const bool save = a.acts != nullptr;
        unsigned mk[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) mk[t] = 0u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = wave * 32 + frag_row(q, kh);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float v = fmaxf(acc[t][q], 0.f);
                Hs[row * P + t * 32 + r] = v;
                if (save)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ares, (int)astore[t],
                                                          (int)(((q & 3) + 8 * (q >> 2)) * a.ldn * 4), 0);
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
    mlp_gemm<NT>(acc, wtile(a.lay.F1, KUp / 8), KUp / 8, Xs, LDX, lane);
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
        mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
        next_ring(Fnext[l]);
        writeback(l + 1);
    }
    // layer 5: K over H (256) then X (skip connection, model/sdf_net.py:59)
    if (SHAPE_BIAS && a.sid)
        init_acc_sid(a.zb5);
    else
        SHAPE_BIAS ? init_acc(a.zb5 + shape * kH) : init_acc_lds(4);
    mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
    mlp_gemm<NT>(acc, wtile(a.lay.F5i, KUp / 8), KUp / 8, Xs, LDX, lane);
    next_ring(a.lay.F6);
    writeback(4);
    // layers 6, 7
    init_acc_lds(5);
    mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
    next_ring(a.lay.F7);
    writeback(5);
    init_acc_lds(6);
    mlp_gemm_ring<NT, 4, kH / 8>(acc, wr, Hs, P, lane, noop);
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
            if (gp < a.N) a.out[gp] = tanhf(v);
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
template <int P, bool SHAPE_BIAS, bool TRAIN>
__global__ void __launch_bounds__(512, 4) sdfnet_fwd_kernel(SdfFwdArgs a) {
    const long b = blockIdx.x;
    if (b < a.nbig)
        sdfnet_fwd_tile<P, SHAPE_BIAS, TRAIN>(a, b * P);
    else
        sdfnet_fwd_tile<kSmallTile, SHAPE_BIAS, TRAIN>(a, a.nbig * P + (b - a.nbig) * kSmallTile);
}

struct SdfBwdArgs {
    const float* dout;   // [N]
    const float* out;    // [N] forward output (tanh)
    const float* acts;   // [7][256][ldn]
    float* dz;           // [7][256][ldn]  dZ1..dZ7
    float* dz8;          // [N]
    float* bsum;         // optional [7*256][nblk]: per-workgroup row sums of dZ1..dZ7 (bias-gradient partials); with `points`
                         // [14*256][nblk]: + rows 7*256.. : sum_p dz8[p] H7[row][p] (w8 gradient), rows (8+c)*256.. and
                         // (11+c)*256.. : sum_p dZ1 / dZ5 [row][p] * xyz_c[p] (the three point columns of dW1 / dW5)
    const float* points; // optional [*,3] (with bsum): the xyz of the points, for the extended partial sums
    long points_period;
    float* dx;           // optional: input gradient, row-major [N][dx_ld] (first KU columns written)
    long dx_ld;
    const float* packed;
    SdfPackLayout lay;
    long ldn;
    long N;
    long nbig;           // workgroups [0, nbig): P-point tiles; the rest: kSmallTile points each (sg_sdfnet_bwd_tile_start)
    long nblk;           // all workgroups = columns of bsum
};

// Backward-data chain for one tile of P points.  G[256][P] holds dH_l; the X-gradient tile DX[KUr][P+1]
// accumulates W5i^T dZ5 + W1^T dZ1 (only when a.dx != nullptr).
// P = 64: 64.3 KB of LDS and <= 128 VGPRs, so two workgroups (16 waves) share a CU and one's mask / write-back phases
// overlap the other's MFMA phases; P = 128 fills the LDS with one workgroup.
template <int P>
__device__ __forceinline__ void sdfnet_bwd_tile(const SdfBwdArgs& a, const long p0) {
    constexpr int NT = P / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Gs = smem;            // [256][P]
    float* dz8s = Gs + kH * P;   // [P]
    float* xs = dz8s + P;        // [3][P] xyz of the tile (only with a.points)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;
    const float4* pk = reinterpret_cast<const float4*>(a.packed);
    const int KUr = a.lay.KUr;
    const int nxt = KUr / 32;  // X row tiles

    if (tid < P) {
        const long gp = p0 + tid;
        float v = 0.f;
        if (gp < a.N) {
            const float o = a.out[gp];
            v = a.dout[gp] * (1.f - o * o);
            a.dz8[gp] = v;
        }
        dz8s[tid] = v;
    }
    const bool ext = a.bsum && a.points;
    if (ext && tid >= 64 && tid < 64 + 3 * P) {   // (P = 64: threads 64..255)
        const int e = tid - 64, c = e / P, pp = e - c * P;
        const long gp = p0 + pp;
        float v = 0.f;
        if (gp < a.N) v = a.points[(a.points_period > 0 ? gp % a.points_period : gp) * 3 + c];
        xs[c * P + pp] = v;
    }
    __syncthreads();

    f32x16 acc[NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    };
    // Each wave owns rows [32 wave, 32 wave + 32) of every dH_l in MFMA fragment layout (row = frag_row(q, kh), point =
    // t*32 + r).  The ReLU mask, the dZ_l write-back and the bias-gradient partial sums happen right there in the GEMM
    // epilogue: H_l is prefetched in the same layout while the GEMM runs (one lane offset + scalar offsets: buffer loads),
    // so there is no separate pass over the LDS tile and its HBM latency is off the critical path.
    // Addressing: the resource base of a layer image is the wave's own row block at the tile's first point (a scalar), the lane
    // adds (4 kh) rows + its point, the fragment row goes into the scalar offset: lane offset + scalar offset < 124 ldn + 256
    // bytes, inside the 2 GiB window for up to 16 M points per call (checked by the host).
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 32;
    bool pok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pok[t] = p0 + t * 32 + r < a.N;
    const unsigned khoff = (unsigned)(4L * kh * a.ldn * 4);
    // Loads of H_l for points beyond N (the ragged last tile) are redirected to the tile's first point: their values are
    // masked out below, but the ADDRESS must stay inside the tensor — an image that ends at the end of a mapped segment put the
    // stray reads on an unmapped page, and the faulting wave then retried forever (seen as a hang that depended on where the
    // caching allocator happened to place `acts`).  Stores of such lanes get an out-of-range offset: the hardware drops them,
    // so the epilogue has no exec-mask branches.
    unsigned hload[NT], zstore[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        hload[t] = pok[t] ? khoff + (unsigned)(t * 32 + r) * 4u : khoff;
        zstore[t] = pok[t] ? khoff + (unsigned)(t * 32 + r) * 4u : kBufOutside;
    }
    auto layer_rsrc = [&](const float* image, int layer) __attribute__((always_inline)) {
        return make_rsrc(image + ((long)layer * kH + wrow) * a.ldn + p0);
    };
    // ReLU'(H_l) for the layers below the last one comes from the forward's 16-bit sign words (one 2-byte load per column tile
    // instead of 16 dword loads): bit q of mk[t] <-> row frag_row(q, kh), point t*32 + r
    unsigned mk[NT];
    // this wave's mask row of the layer about to be processed (walks down one layer = 16 rows per step: a running scalar, so
    // that the six bases are not all computed — and kept in SGPRs — up front)
    const unsigned short* mkrow = sdf_mask_base(a.acts, a.ldn) + ((long)5 * 16 + (wrow >> 4)) * a.ldn + p0;
    auto load_mask = [&](int layer) __attribute__((always_inline)) {
        // the resource starts at this wave's (layer, group 2 wave + kh) row and the tile's first point: the lane offset is just
        // its point (lanes beyond N read the tile's first point; their bits are masked by pok)
        const __amdgpu_buffer_rsrc_t mres = make_rsrc(mkrow);
        mkrow -= 16 * a.ldn;
        asm volatile("" : "+s"(mkrow));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#ifdef SG_ABL_NOLOAD
            mk[t] = 0xffffu;
#else
            mk[t] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(mres, (int)(((hload[t] - khoff) >> 1) + (khoff >> 3)), 0, 0);
#endif
    };
    float hf[16][NT];
    auto load_h = [&](int layer) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t hres = layer_rsrc(a.acts, layer);
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#ifdef SG_ABL_NOLOAD
                hf[q][t] = 1.f;
#else
                hf[q][t] = buf_load(hres, hload[t], (unsigned)(((q & 3) + 8 * (q >> 2)) * a.ldn * 4));
#endif
    };
    // Row sums over the 32 points a half-wave holds, on the VALU's DPP path (quad swaps, half-row mirror, row mirror, then
    // lane 15 of the even rows broadcast into the odd rows): no LDS traffic and no dependent ds_bpermute round trips, so the 16
    // rows of an epilogue pipeline freely.  Lanes 16..31 / 48..63 end up with the totals of lanes 0..31 / 32..63; lanes 31 and 63
    // store (every other lane carries an out-of-range offset).
    auto dpp_add = [&](float v, auto ctrl, auto rowmask) __attribute__((always_inline)) {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value,
                                                                         decltype(rowmask)::value, 0xf, false));
    };
    auto half_sum = [&](float v) __attribute__((always_inline)) {
        v = dpp_add(v, IntTag<0xB1>(), IntTag<0xf>());    // quad_perm [1,0,3,2]
        v = dpp_add(v, IntTag<0x4E>(), IntTag<0xf>());    // quad_perm [2,3,0,1]
        v = dpp_add(v, IntTag<0x141>(), IntTag<0xf>());   // row_half_mirror
        v = dpp_add(v, IntTag<0x140>(), IntTag<0xf>());   // row_mirror
        return dpp_add(v, IntTag<0x142>(), IntTag<0xa>());   // row_bcast15 into rows 1 and 3
    };
    const unsigned bsoff = r == 31 ? (unsigned)((4L * kh * a.nblk + blockIdx.x) * 4) : kBufOutside;
    // partial-sum block `blk` (0..6: dZ1..dZ7, 7: w8, 8..10 / 11..13: point columns): [256][nblk], this wave's rows from wrow
    auto partial_rsrc = [&](int blk) __attribute__((always_inline)) {
        return make_rsrc(a.bsum + ((long)blk * kH + wrow) * a.nblk);
    };
    auto row_partial = [&](float v, __amdgpu_buffer_rsrc_t res, int q) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, half_sum(v)), res, (int)bsoff,
                                              (int)(((q & 3) + 8 * (q >> 2)) * a.nblk * 4), 0);
    };
    // dZ_l = acc * (H_l > 0): to the LDS tile (B operand of the next GEMM), to the dz image, row sums to bsum
    // XC: -1, or the block of extended partial rows (8: dW1 point columns, 11: dW5 point columns) this layer feeds
    auto mask_store = [&](int layer, auto xtag, auto bits_tag) __attribute__((always_inline)) {
        constexpr int XC = decltype(xtag)::value;
        constexpr bool BITS = decltype(bits_tag)::value != 0;    // ReLU' from the sign words (mk) instead of the fp32 image (hf)
        const __amdgpu_buffer_rsrc_t zres = layer_rsrc(a.dz, layer);
        const __amdgpu_buffer_rsrc_t bres = partial_rsrc(a.bsum ? layer : 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = wave * 32 + frag_row(q, kh);
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float g;
                if (BITS) {
                    // bit q of the sign word; lanes beyond N had their word cleared
                    // (toolchain note, hipcc ROCm 7.2: forming the result as `bit_cast<int>(acc[t][q]) & -(bit q)` — two VALU
                    // instructions, no compare — compiled to ANDs that all read element 0 of the accumulator vector; caught
                    // by the parity tests, kept as a select)
                    g = ((mk[t] >> q) & 1u) ? acc[t][q] : 0.f;
                } else {
                    g = (pok[t] && hf[q][t] > 0.f) ? acc[t][q] : 0.f;
                }
                Gs[row * P + t * 32 + r] = g;
#ifndef SG_ABL_NOSTORE
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), zres, (int)zstore[t],
                                                      (int)(((q & 3) + 8 * (q >> 2)) * a.ldn * 4), 0);
#endif
                rs += g;
            }
            if (a.bsum) row_partial(rs, bres, q);
        }
        if (XC >= 0 && ext) {
            // point columns: a second, short pass over the rows this wave has just written to LDS (same wave, LDS operations
            // are in order: no barrier), so that the products do not lengthen the register lifetimes of the loop above
            float xv[3][NT];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) xv[c][t] = xs[c * P + t * 32 + r];
            const __amdgpu_buffer_rsrc_t xres[3] = {partial_rsrc(XC), partial_rsrc(XC + 1), partial_rsrc(XC + 2)};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = wave * 32 + frag_row(q, kh);
                float gq[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) gq[t] = Gs[row * P + t * 32 + r];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) v = fmaf(gq[t], xv[c][t], v);
                    row_partial(v, xres[c], q);
                }
            }
        }
    };

    // dZ7 = (w8 (x) dz8) * (H7 > 0): the outer product is formed directly in fragment layout
    load_h(6);
    {
        const float* w8 = a.packed + a.lay.W8;
        const __amdgpu_buffer_rsrc_t w8res = partial_rsrc(ext ? 7 : 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float wv = w8[wave * 32 + frag_row(q, kh)];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][q] = wv * dz8s[t * 32 + r];
            if (ext) {   // w8 gradient partial: sum_p dz8[p] * H7[row][p]  (H7 = relu output, already in registers)
                float s8 = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) s8 = fmaf(pok[t] ? hf[q][t] : 0.f, dz8s[t * 32 + r], s8);
                row_partial(s8, w8res, q);
            }
        }
    }
    // the weight ring of the next GEMM is started before the stores of each epilogue (see WRing)
    WRing<SG_BWD_RING> wr;
    auto wtile_t = [&](long toff) { return pk + (toff >> 2) + (long)wave * (kH / 8) * 64; };
    wring_start(wr, wtile_t(a.lay.T7), lane);
    __builtin_amdgcn_sched_barrier(0);
    mask_store(6, IntTag<-1>(), IntTag<0>());      // (H7 itself is in registers: the w8 gradient above needs its values)
    __syncthreads();
    // dZ_layer+1 (LDS) -> dZ_layer; tnext: transposed pack of the following step (-1: none)
    // (whether a next step exists is a COMPILE-TIME tag: as a run-time test of the pack offset it was a branch around the ring start,
    // and at the join behind it the compiler — s_waitcnt vmcnt counts in issue order, a join takes the path with the fewest
    // younger loads — waited vmcnt(0) for the mask word: i.e. for the ring loads it had just issued, a full weight-fetch latency
    // in front of every epilogue, which is exactly what starting the ring early was meant to hide)
    auto back_step = [&](int layer, long tnext, auto xtag, auto has_next) __attribute__((always_inline)) {
        zero_acc();
#ifdef SG_SDF_NO_MASK   // A/B build (scripts/ab_build.sh): ReLU' from the fp32 images, as before round 3
        mlp_gemm_ring<NT, SG_BWD_RING, kH / 8>(acc, wr, Gs, P, lane, [&]() __attribute__((always_inline)) { load_h(layer); });
        __syncthreads();
        if (decltype(has_next)::value) wring_start(wr, wtile_t(tnext), lane);
        __builtin_amdgcn_sched_barrier(0);
        mask_store(layer, xtag, IntTag<0>());
#else
        mlp_gemm_ring<NT, SG_BWD_RING, kH / 8>(acc, wr, Gs, P, lane, [&]() __attribute__((always_inline)) { load_mask(layer); });
        __syncthreads();   // every wave is done reading the tile
        if (decltype(has_next)::value) wring_start(wr, wtile_t(tnext), lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) mk[t] = pok[t] ? mk[t] : 0u;
        mask_store(layer, xtag, IntTag<1>());
#endif
        __syncthreads();
    };
    back_step(5, a.lay.T6, IntTag<-1>(), IntTag<1>());    // dH6 -> dZ6
    back_step(4, a.lay.T5x, IntTag<11>(), IntTag<1>());   // dZ5 (+ point columns of dW5)
    back_step(3, a.lay.T4, IntTag<-1>(), IntTag<1>());    // dZ4
    back_step(2, a.lay.T3, IntTag<-1>(), IntTag<1>());    // dZ3
    back_step(1, a.lay.T2, IntTag<-1>(), IntTag<1>());    // dZ2
    back_step(0, -1, IntTag<8>(), IntTag<0>());           // dZ1 (+ point columns of dW1)
    // (element indexing of the dX part below)
    constexpr int HE = kH * P / 512;      // elements per thread: rows (tid / P) + i * (512 / P), point tid % P
    constexpr int RSTEP = 512 / P;
    const int mrow = tid / P, mp = tid % P;
    const long mgp = p0 + mp;
    const bool mok = mgp < a.N;
    const long moff = mok ? (long)mrow * a.ldn + mgp : 0;   // element 0 of this thread inside a layer's [256][ldn] image
    const long mstride = (long)RSTEP * a.ldn;
    // ---- input gradient dX = W1^T dZ1 + W5[:,256:]^T dZ5 (rows = input features, 32-row tiles round-robin over waves) ----
    // G holds dZ1 now; the dZ5 tile is read back from the dz image this workgroup wrote (L2-hot).  Accumulating both
    // products in registers and writing dx once needs neither LDS nor a read-modify-write.
    if (a.dx) {
        auto reload = [&](int layer) {  // G <- dZ_{layer+1} tile
            const float* z = a.dz + (long)layer * kH * a.ldn + moff;
            __syncthreads();
#pragma unroll 8
            for (int i = 0; i < HE; ++i) {
                const float v = z[mok ? i * mstride : 0];
                Gs[tid + i * 512] = mok ? v : 0.f;
            }
            __syncthreads();
        };
        for (int pass = 0; pass * 8 < nxt; ++pass) {
            const int xt = pass * 8 + wave;
            const bool mine = xt < nxt;
            if (pass > 0) reload(0);
            zero_acc();
            if (mine) mlp_gemm<NT>(acc, pk + (a.lay.T1 >> 2) + (long)xt * (kH / 8) * 64, kH / 8, Gs, P, lane);
            reload(4);
            if (mine) {
                mlp_gemm<NT>(acc, pk + (a.lay.T5i >> 2) + (long)xt * (kH / 8) * 64, kH / 8, Gs, P, lane);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const long gp = p0 + t * 32 + r;
                    if (gp < a.N) {
                        float* d = a.dx + gp * a.dx_ld;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int row = xt * 32 + frag_row(q, kh);
                            if (row < a.lay.KU) d[row] = acc[t][q];
                        }
                    }
                }
            }
        }
    }
}

template <int P>
__global__ void __launch_bounds__(512, (P == 64 ? 4 : 2)) sdfnet_bwd_kernel(SdfBwdArgs a) {
    const long b = blockIdx.x;
    if (b < a.nbig)
        sdfnet_bwd_tile<P>(a, b * P);
    else
        sdfnet_bwd_tile<kSmallTile>(a, a.nbig * P + (b - a.nbig) * kSmallTile);
}

// t[which][row][s] = sum over the points of segment s of dZ_layer[row][.] (which 0: dZ1, 1: dZ5): the per-shape sums behind
// the latent-table gradient and the latent columns of dW1 / dW5.  The fused backward has already reduced every row over every
// tile (bias partials), so a segment costs its interior tiles' partials plus the points of the (at most two) tiles its ends cut:
// 3 loads per lane instead of a pass over the two [256][N] images.  One wave per (row, segment).
__global__ void __launch_bounds__(256) sdfnet_segsum_kernel(const float* __restrict__ dz, const float* __restrict__ bsum,
                                                            long ldn, long nbig, long nblk, const int64_t* __restrict__ off,
                                                            long S, float* __restrict__ t1, float* __restrict__ t5) {
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (long)kH * S) return;
    const int layer = blockIdx.y ? 4 : 0;
    const long row = pair / S, sg = pair - row * S;
    const long beg = off[sg], end = off[sg + 1];
    const int lane = threadIdx.x & 63;
    const long edge = nbig * 64;   // first point of the small tiles
    // first tile that starts at or after beg / last tile boundary at or before end
    const long ta = beg <= edge ? (beg + 63) / 64 : nbig + (beg - edge + kSmallTile - 1) / kSmallTile;
    const long tb = end <= edge ? end / 64 : nbig + (end - edge) / kSmallTile;
    auto start = [&](long t) { return t <= nbig ? t * 64 : edge + (t - nbig) * kSmallTile; };
    const float* p = dz + ((long)layer * kH + row) * ldn;
    float acc = 0.f;
    if (ta >= tb) {
        for (long e = beg + lane; e < end; e += 64) acc += p[e];
    } else {
        const long head = start(ta), tail = start(tb);
        for (long e = beg + lane; e < head; e += 64) acc += p[e];
        const float* q = bsum + ((long)layer * kH + row) * nblk;
        for (long t = ta + lane; t < tb; t += 64) acc += q[t];
        for (long e = tail + lane; e < end; e += 64) acc += p[e];
    }
    acc = sg_wave_sum(acc);
    if (lane == 0) (blockIdx.y ? t5 : t1)[pair] = acc;
}

// ---- backward of the per-shape latent fold (the latent columns of layers1.0 / layers2.0 enter the forward as per-shape bias rows) ----
// Backward of the fold from the per-shape sums t1 / t5 [256][S] of dZ1 / dZ5:
//   blocks [0, 256):     row o of the latent columns  dW1[o][3 + k] = sum_s t1[o][s] z[s][k],  dW5[o][259 + k] = sum_s t5[o][s] z[s][k]
//   blocks [256, 256+S): latent gradient row          gz[s][k] = sum_o t1[o][s] W1[o][3 + k] + t5[o][s] W5[o][259 + k]
// thread = k (strided by 256 for L > 256); the broadcast operand (a t row / column) sits in LDS.
__global__ void __launch_bounds__(256) shape_bias_bwd_kernel(const float* __restrict__ t1, const float* __restrict__ t5, int S,
                                                             const float* __restrict__ z, int L, const float* __restrict__ W1,
                                                             const float* __restrict__ W5, float* __restrict__ dW1,
                                                             float* __restrict__ dW5, float* __restrict__ gz) {
    extern __shared__ float sh[];   // rows: 2 x S, columns: 2 x 256
    const int tid = threadIdx.x;
    const int ld1 = 3 + L, ld5 = kH + 3 + L;
    // Both halves walk a short reduction (S shapes / 256 rows) with two global loads per term.  Written as a plain loop the loads
    // were issued one at a time behind the dependent double-precision FMA chain — 256 exposed L2 latencies, 163 us for 17 MFLOP
    // (18 % of the 20 000-point auto-decoder step).  The terms are now fetched in batches of 16 into registers (all loads of a
    // batch in flight together) and then accumulated in the same order as before.
    constexpr int kB = 16;
    if (blockIdx.x < kH) {
        if (!dW1) return;
        const int o = blockIdx.x;
        for (int s = tid; s < S; s += 256) {
            sh[s] = t1[(long)o * S + s];
            sh[S + s] = t5[(long)o * S + s];
        }
        __syncthreads();
        for (int k = tid; k < L; k += 256) {
            double a = 0, b = 0;    // few terms, ill-conditioned sums (per-shape sums of either sign): accumulate in double
            for (int s0 = 0; s0 < S; s0 += kB) {
                float zv[kB];
#pragma unroll
                for (int i = 0; i < kB; ++i) zv[i] = s0 + i < S ? z[(long)(s0 + i) * L + k] : 0.f;
#pragma unroll
                for (int i = 0; i < kB; ++i) {
                    if (s0 + i < S) {
                        a = fma((double)sh[s0 + i], (double)zv[i], a);
                        b = fma((double)sh[S + s0 + i], (double)zv[i], b);
                    }
                }
            }
            dW1[(long)o * ld1 + 3 + k] = (float)a;
            dW5[(long)o * ld5 + kH + 3 + k] = (float)b;
        }
    } else {
        if (!gz) return;
        const int s = blockIdx.x - kH;
        sh[tid] = t1[(long)tid * S + s];
        sh[kH + tid] = t5[(long)tid * S + s];
        __syncthreads();
        for (int k = tid; k < L; k += 256) {
            double a = 0, b = 0;
            for (int o0 = 0; o0 < kH; o0 += kB) {
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
            gz[(long)s * L + k] = (float)(a + b);
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
constexpr long kFwdSlots = kCUs;       // one workgroup per CU (LDS)
constexpr long kBwdSlots = 2 * kCUs;   // two per CU

static size_t fwd_lds_bytes(int P, int KUp) { return ((size_t)kH * P + (size_t)KUp * (P + 1) + 16 * P + 7 * kH) * sizeof(float); }
static size_t bwd_lds_bytes(int P, int KUr, bool dx) { return ((size_t)kH * P + 4 * P) * sizeof(float); }

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

extern "C" {

size_t sg_sdfnet_packed_floats(int kin_used) { return (size_t)make_layout(kin_used).total; }

// floats of the activation buffer of a training call with leading dimension ldn: the seven fp32 images [7][256][ldn] plus the
// sign masks (unsigned short [7][16][ldn] = 56 ldn floats)
size_t sg_sdfnet_acts_floats(long ldn) { return (size_t)7 * kH * ldn + (size_t)56 * ldn; }

// params: host array of 16 device pointers in state_dict order
//   layers1.{0,2,4,6}.{weight,bias}, layers2.{0,2,4,6}.{weight,bias}   (model/sdf_net.py:26-53)
int sg_sdfnet_pack(const float* const* params, int latent, int kin_used, float* packed, hipStream_t stream) {
    SG_CHECK_ARG(params && packed && latent >= 0 && (kin_used == 3 || kin_used == 3 + latent));
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
    hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(64, n), dim3(256), 0, stream, D, packed);
    hipLaunchKernelGGL(copy_vectors_kernel, dim3(1), dim3(256), 0, stream, params[1], params[3], params[5], params[7],
                       params[9], params[11], params[13], params[15], W8, packed + L.B, packed + L.W8);
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

// Per-shape mode: the backward of the latent fold (zb1 = b1 + z W1[:, 3:]^T, zb5 = b5 + z W5[:, 259:]^T) from the per-shape sums
// t1 / t5 [256][S] of dZ1 / dZ5 (sg_sdfnet_segsum or sg_rowsum): the latent columns of dW1 / dW5 (written in place, row strides of
// the full matrices; pass NULL to skip) and the latent gradient gz [S][L] (NULL to skip).
int sg_sdfnet_shape_bias_bwd(const float* t1, const float* t5, long nshapes, const float* z, int latent, const float* W1,
                             const float* W5, float* dW1, float* dW5, float* gz, hipStream_t stream) {
    SG_CHECK_ARG(t1 && t5 && z && W1 && W5 && nshapes > 0 && latent > 0 && (dW1 == nullptr) == (dW5 == nullptr));
    SG_CHECK_ARG(nshapes <= 6144);   // a t row of every shape in 48 KB of LDS
    const size_t lds = (size_t)2 * (nshapes > kH ? nshapes : kH) * sizeof(float);
    hipLaunchKernelGGL(shape_bias_bwd_kernel, dim3((unsigned)(kH + nshapes)), dim3(256), lds, stream, t1, t5, (int)nshapes, z,
                       latent, W1, W5, dW1, dW5, gz);
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

// Per-segment sums of dZ1 and dZ5 (t1, t5: [256][nseg]) from the images and the tile partials of the same sg_sdfnet_bwd call
// (bias_partials as written there: rows [0, 7*256) x sg_sdfnet_bwd_blocks(N) columns).
int sg_sdfnet_segsum(const float* dz, const float* bias_partials, long ldn, long N, const int64_t* seg_off, long nseg, float* t1,
                     float* t5, hipStream_t stream) {
    SG_CHECK_ARG(dz && bias_partials && seg_off && t1 && t5 && N > 0 && ldn >= N && nseg > 0);
    const TilePlan tp = tile_plan(N, SG_BWD_TILE, kBwdSlots);
    hipLaunchKernelGGL(sdfnet_segsum_kernel, dim3((unsigned)(((long)kH * nseg + 3) / 4), 2), dim3(256), 0, stream, dz,
                       bias_partials, ldn, tp.nbig, tp.nbig + tp.nsmall, seg_off, nseg, t1, t5);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"
