// shapegan_amd/csrc/conv3d_edge.hip — the k4/s2/p1 layers with ONE channel on the voxel-grid side (gfx950 only):
//   gan.Discriminator.layers[0]  Conv3d(1 -> 64)   (model/gan.py:49), progressive stage `it` on from_SDF's single real channel
//   (model/progressive_gan.py:38), Autoencoder.encoder[0] Conv3d(1 -> 24) (model/autoencoder.py:16): forward + weight gradient;
//   gan.Generator.layers[9] ConvTranspose3d(64 -> 1) (model/gan.py:21), Autoencoder.decoder's ConvTranspose3d(24 -> 1)
//   (model/autoencoder.py:63): forward = the input gradient of the Conv3d(1 -> C) form.
//
// These layers move 64x more activation bytes than grid bytes and are HBM-bound (28 flop/B): SURVEY.md 8d asks for HBM GB/s
// here.  Each output/operand is touched once; the small 1-channel grid is re-read from L1/L2.
//
//   forward  (conv_fwd_c1_kernel)    GEMM rows = positions, cols = Cout (<= 64), K = 64 taps.  The grid is read IN PLACE: a lane's tap
//            values are dword buffer loads at lane offset + (kd, kh) scalar offset, and a tap that falls into the zero padding
//            carries an out-of-range offset (the hardware returns 0): no padded copy, no masks, no LDS, no VALU on the operands.
//            The 64 x 64 weight matrix lives in registers as MFMA B fragments for the whole kernel (filled through LDS); a wave
//            walks a contiguous run of position tiles of 32, the patch loads of the next tile are issued before the MFMAs of
//            the current one.  Positions are the ROWS of the product so that 4 accumulator registers are 4 consecutive
//            positions of one channel: 16-byte stores (row-per-lane dword stores were store-issue bound).
//   wgrad    (conv_wgrad_c1_kernel)  GEMM M = Cout, N = 64 taps, K = positions (split over all waves, deterministic two-level
//            reduction).  The dy tile of a pass ([Cout][32 positions]) is copied with fully coalesced 16-byte loads, transposed
//            through a wave-private LDS tile and read back as fragments; the patch operand comes from x in place (lane = tap,
//            stride-2 gather served by L1, padding by out-of-range offsets).
//   dgrad    (tapplane_gemm_kernel + col2im_c1_kernel)  out[2q+p] = sum_co sum_t dy[co][q+d] w[co][k]: first the 64 tap planes
//            S[k][q] = sum_co w[co][k] dy[co][q] (a dense GEMM rows = positions, cols = 64 taps, K = Cout; a 1-row GEMM per
//            parity would waste 31/32 of every MFMA), then each output gathers its 8 taps from the planes (every plane element
//            is used exactly once; the planes stay in the 256 MB Infinity Cache between the two kernels).
#include <stdlib.h>

#include "conv_common.h"
#include "../../include/shapegan_hip.h"

namespace sg {

// ---- forward ---------------------------------------------------------------------------------------------------------
struct EdgeFwdArgs {
    const float* x;     // [batch][Cx][ID][IH][IW], channel 0 is read — straight from the tensor, no padded copy
    const float* w;     // [Cout][Cin_total][64], channel 0 used
    const float* bias;  // [Cout] or null
    float* y;           // [batch][Cy][O3]
    int OD, OH, OW, IH, IW, Cout, Cy, Cin_total;
    long x_sample;      // floats between samples of x (Cx * I3)
    int tiles_per_sample, total_tiles;
    FastDiv dtps, dOW, dOH;
    int act;
    float slope;
};

__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float x, float y, float z,
                                           float w) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    v.x = __builtin_bit_cast(unsigned, x);
    v.y = __builtin_bit_cast(unsigned, y);
    v.z = __builtin_bit_cast(unsigned, z);
    v.w = __builtin_bit_cast(unsigned, w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
    // A 16-byte store reads its data registers over several cycles.  hipcc (ROCm 7.2) reuses them in the very next
    // instruction when the store carries an SGPR offset (it only guards the immediate-offset form), and on gfx950 the lanes
    // read last (12-15 of every 16) then pick up the new value: nondeterministic corruption of exactly those lanes was
    // observed.  Two wait states before anything else may issue close the window.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// D[row = position][col = channel]: the patch values are the A operand, the weights the B operand, so that a lane ends up
// with 4 consecutive positions of ONE channel per 4 accumulator registers -> 16-byte stores (4x fewer store instructions than
// channel-major rows, which were store-issue bound).
//
// Round 3 (measured on the MI355X: 67 -> see DESIGN.md 3.2):
//  * no padded copy of the grid.  K is ordered (kd, kh, kw') with lane half 0 owning the taps kw = 1, 2 (w = 2 ow, 2 ow + 1: always
//    inside the row) and lane half 1 the taps kw = 3, 0 (w = 2 ow + 2, 2 ow - 1: outside the row at the right / left border).  Every
//    tap value is one dword buffer load at lane offset + (kd, kh) scalar offset; a lane whose tap falls into the zero padding
//    (left / right by its own ow, top / bottom / front / back by its (od, oh) and the (kd, kh) of the load) carries an
//    out-of-range offset instead and the hardware returns 0 — exact padding semantics, no masks on the values, no stray reads
//    outside the tensor, and the 14 us pad pass per call is gone;
//  * the 64 x 64 weight tile reaches the B fragments through LDS (one coalesced copy per workgroup, conflict-free fragment
//    reads) instead of 64 uncoalesced dword loads per lane (32 cache lines per instruction: ~8 us per CU before the first MFMA);
//  * every wave walks a contiguous run of tiles (consecutive 128-byte lines of each channel row, the patch rows of the next
//    tile already in L1);
//  * bias + LeakyReLU as max(t, slope t) on 2-wide packed adds / multiplies.
#ifndef SG_FWDC1_ABL
#define SG_FWDC1_ABL 0   // ablation builds only (scripts/ab_build.sh): 1 no global stores, 2 no MFMAs, 4 no patch loads
#endif
template <int NT, int ACT>   // NT: column tiles of 32 output channels (1 or 2); ACT: 0 none, 1 LeakyReLU, 2 any (sg_apply_act)
__global__ void __launch_bounds__(256) conv_fwd_c1_kernel(EdgeFwdArgs a) {
    constexpr int kWL = 65;   // LDS row stride of the weight tile: lane = channel row -> 32 distinct banks per fragment read
    constexpr int kTL = 36;   // LDS row stride of the output staging tile [channel][32 positions + 4]: 16-byte aligned rows
    __shared__ float wl[NT * 32 * kWL];
    __shared__ __attribute__((aligned(16))) float tl[4][NT * 32 * kTL];   // one staging tile per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kh2 = lane >> 5;
    {
        // all loads of the copy are issued before the first LDS write (a load / wait / write loop serialises NT * 8 memory
        // latencies: measured 10+ us of every launch); rows beyond Cout become zero columns
        const __amdgpu_buffer_rsrc_t wres = make_rsrc(a.w);
        f32x4 wv[NT * 2];
#pragma unroll
        for (int i = 0; i < NT * 2; ++i) {
            const int e4 = threadIdx.x + 256 * i, co = e4 >> 4, t4 = e4 & 15;     // 16 float4 per weight row
            wv[i] = buf_load4v(wres, co < a.Cout ? (unsigned)(((long)co * a.Cin_total * 64 + t4 * 4) * 4) : kBufOutside, 0);
        }
#pragma unroll
        for (int i = 0; i < NT * 2; ++i) {
            const int e4 = threadIdx.x + 256 * i, co = e4 >> 4, t4 = e4 & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) wl[co * kWL + t4 * 4 + j] = wv[i][j];
        }
    }
    __syncthreads();
    // B fragments: wfr[nt][kd*4+kh][j] = W[32 nt + r][kd][kh][kw], kw = 1, 2 (lane half 0) / 3, 0 (lane half 1) for j = 0, 1
    float wfr[NT][16][2], bl[NT];
    const int kw0 = kh2 ? 3 : 1, kw1 = kh2 ? 0 : 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 32 + r;
        bl[nt] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            wfr[nt][g][0] = wl[co * kWL + g * 4 + kw0];
            wfr[nt][g][1] = wl[co * kWL + g * 4 + kw1];
        }
    }
    // x is addressed from (IH + 1) rows before its start, so that the (kd, kh) = (0, 0) scalar offset is 0 and lane offsets are
    // non-negative; nothing below x is ever dereferenced (those lanes carry the out-of-range offset)
    const __amdgpu_buffer_rsrc_t xres = make_rsrc(a.x - ((long)a.IH * a.IW + a.IW));
    const __amdgpu_buffer_rsrc_t yres = make_rsrc(a.y);
    const unsigned O3 = (unsigned)(a.OD * a.OH * a.OW);
    unsigned soff[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) soff[g] = (unsigned)(((g >> 2) * a.IH + (g & 3)) * a.IW) * 4u;

    // every wave takes a contiguous run of tiles
    const int nwaves = gridDim.x * 4;
    const int per_wave = (a.total_tiles + nwaves - 1) / nwaves;
    int t = (blockIdx.x * 4 + wave) * per_wave;
    const int t_end = min(a.total_tiles, t + per_wave);
    if (t >= t_end) return;

    // per tile and lane: o0 / o1 = offsets of the lane's two taps in the (kd, kh) = (0, 0) row (or out of range at the left /
    // right border), edge[4] = the lane's position touches the front / back / top / bottom face, yoff = its output rows
    struct TileLane {
        unsigned o0, o1;
        bool d0, d3, h0, h3;
    };
    auto lane_offsets = [&](int tile, TileLane& L, unsigned (&yoff)[1]) __attribute__((always_inline)) {
        uint32_t n, tp, q1, ow, oh, od;
        a.dtps.divmod((uint32_t)tile, n, tp);
        const uint32_t p = tp * 32 + r;
        a.dOW.divmod(p, q1, ow);
        a.dOH.divmod(q1, od, oh);
        const unsigned base = (unsigned)((long)n * a.x_sample + ((long)(2 * od) * a.IH + 2 * oh) * a.IW + 2 * ow) * 4u;   // w = 2 ow
        const bool left = ow == 0, right = (int)ow == a.OW - 1;
        L.o0 = kh2 ? (right ? kBufOutside : base + 8u) : base;          // kw 3: w = 2 ow + 2   | kw 1: w = 2 ow
        L.o1 = kh2 ? (left ? kBufOutside : base - 4u) : base + 4u;      // kw 0: w = 2 ow - 1   | kw 2: w = 2 ow + 1
        L.d0 = od == 0;
        L.d3 = (int)od == a.OD - 1;
        L.h0 = oh == 0;
        L.h3 = (int)oh == a.OH - 1;
        // output: this lane stores 16 bytes of channel row (lane >> 3) + 8 i at positions 4 (lane & 7) .. + 3 of the tile
        yoff[0] = (unsigned)(((long)n * a.Cy + (lane >> 3)) * O3 + tp * 32 + 4 * (lane & 7)) * 4u;
    };
    auto load_patch = [&](const TileLane& L, float (&b0)[16], float (&b1)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int kd = g >> 2, kh = g & 3;
            bool out = false;   // (compile-time structure: the interior (kd, kh) need no test at all)
            if (kd == 0) out = out || L.d0;
            if (kd == 3) out = out || L.d3;
            if (kh == 0) out = out || L.h0;
            if (kh == 3) out = out || L.h3;
            b0[g] = buf_load(xres, out ? kBufOutside : L.o0, soff[g]);
            b1[g] = buf_load(xres, out ? kBufOutside : L.o1, soff[g]);
        }
    };
    TileLane L;
    unsigned yoff[1];
    lane_offsets(t, L, yoff);
    float* const tw = tl[wave];
    float c0[16], c1[16];
    load_patch(L, c0, c1);

    // Software pipeline: the epilogue of tile t-1 (bias + activation, the trip through the LDS staging tile, the stores) is cut
    // into 16 slices that are issued BETWEEN the 16 MFMA groups of tile t — a 32x32x2 MFMA occupies the matrix pipe for 64
    // cycles during which the wave is free to issue its VALU / LDS / VMEM work, so the pipe no longer idles through an
    // epilogue (two waves per SIMD with identical phase structure run in lockstep and do not hide each other's epilogues:
    // the matrix pipe was 39 % busy by the counters).  pv holds the finished accumulators of the previous tile.
    // Output path: registers 4c .. 4c+3 of D are 4 consecutive positions of channel r; written as they are, a store instruction
    // would put 32 bytes into each of 32 channel rows — partial cache lines, which the HBM write path handles worst (1.8 TB/s
    // once the output outgrows the Infinity Cache).  The tile goes through a wave-private LDS tile [channel][32 positions] (16-byte
    // writes and reads, conflict-free with the 36-float row stride) and leaves as whole 128-byte lines: 8 lanes per channel
    // row, 8 rows per store instruction.
    float pv[NT][16];
    unsigned yprev = kBufOutside;
    auto epilogue_slice = [&](int sl) __attribute__((always_inline)) {
        if (sl < NT * 4) {            // slices 0 .. 4 NT - 1: one (column tile, 4-position group) each -> LDS
            const int nt = sl >> 2, c = sl & 3;
            f32x4 v = {pv[nt][4 * c], pv[nt][4 * c + 1], pv[nt][4 * c + 2], pv[nt][4 * c + 3]};
            v = v + bl[nt];
            if (ACT == 1) {
                const f32x4 sv = v * a.slope;           // LeakyReLU with 0 <= slope <= 1: max(t, slope t)
                v = __builtin_elementwise_max(v, sv);
            } else if (ACT == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = sg_apply_act(v[i], a.act, a.slope);
            }
            *reinterpret_cast<f32x4*>(tw + (nt * 32 + r) * kTL + 8 * c + 4 * kh2) = v;
        } else if (sl >= 8 && sl < 8 + NT * 4) {   // slices 8 ..: 8 channel rows each -> memory
            const int i = sl - 8;
            const int row = 8 * i + (lane >> 3);
            const f32x4 v = *reinterpret_cast<const f32x4*>(tw + row * kTL + 4 * (lane & 7));
            buf_store4(yres, row < a.Cout && !(SG_FWDC1_ABL & 1) ? yprev : kBufOutside, (unsigned)(8 * i) * O3 * 4u, v[0], v[1], v[2], v[3]);
        }
    };
    auto tile_mfma = [&](auto with_epilogue, const float (&c0)[16], const float (&c1)[16]) __attribute__((always_inline)) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nt][q] = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (SG_FWDC1_ABL & 2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][g] += c0[g] * wfr[nt][g][0] + c1[g] * wfr[nt][g][1];
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0[g], wfr[nt][g][0], acc[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[g], wfr[nt][g][1], acc[nt], 0, 0, 0);
            }
            if (decltype(with_epilogue)::value) {
                __builtin_amdgcn_sched_barrier(0);
                epilogue_slice(g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[nt][q] = acc[nt][q];
    };
    // issue the patch loads of tile tcur + 1 into (n0, n1).  Behind the last tile the lane offsets are all out of range (nothing
    // is fetched) — NOT skipped by a branch: s_waitcnt vmcnt counts in issue order and at a control-flow join the compiler assumes
    // the path with the fewest younger loads in flight.
    auto advance = [&](int tcur, float (&n0)[16], float (&n1)[16]) __attribute__((always_inline)) {
        TileLane Ln;
        unsigned yn[1];
        lane_offsets(tcur + 1 < t_end ? tcur + 1 : tcur, Ln, yn);
        if (tcur + 1 >= t_end || (SG_FWDC1_ABL & 4)) Ln.o0 = Ln.o1 = kBufOutside;
        load_patch(Ln, n0, n1);
        yprev = yoff[0];       // where tile tcur's output goes (its epilogue runs during tile tcur + 1)
        L = Ln;
        yoff[0] = yn[0];
    };
    // One tile: loads of the next tile into (n0, n1), then this tile's MFMAs on (c0, c1) with the previous tile's epilogue between
    // them.  The loop below alternates two register sets instead of copying next -> current at the end of every tile: the copy
    // needed the loads it had just issued, so every tile ended in s_waitcnt vmcnt(8) — a full memory latency per tile with
    // nothing to cover it (ISA listing, round 4; 45 us at 128 samples = 0.42 of the HBM roofline with the matrix pipe 45 % busy).
    auto tile_step = [&](int tt, auto with_epilogue, const float (&c0)[16], const float (&c1)[16], float (&n0)[16],
                         float (&n1)[16]) __attribute__((always_inline)) {
        const unsigned ydone = yprev;     // output offsets of tile tt - 1, whose epilogue runs inside this tile's MFMA groups
        advance(tt, n0, n1);
        const unsigned ynext = yprev;
        yprev = ydone;
        __builtin_amdgcn_sched_barrier(0);
        tile_mfma(with_epilogue, c0, c1);
        yprev = ynext;
    };
    float d0[16], d1[16];
    // first tile: nothing to drain yet
    tile_step(t, IntTag<0>(), c0, c1, d0, d1);
    for (++t; t < t_end; t += 2) {
        tile_step(t, IntTag<1>(), d0, d1, c0, c1);
        if (t + 1 < t_end) tile_step(t + 1, IntTag<1>(), c0, c1, d0, d1);
    }
    // drain: the epilogue of the last tile
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) epilogue_slice(sl);
}

// ---- forward, LDS-staged grid (round 5) -------------------------------------------------------------------------------------
// conv_fwd_c1_kernel gathers its A operand (the patch values of 32 positions) with 32 dword buffer loads per tile that can only be
// issued one tile ahead (two register sets): its ablations (scripts/ab_build.sh -DSG_FWDC1_ABL=.., cold operands) add up — loads
// alone 25 us, + MFMAs 43, + stores 60 us at 128 samples for a 27 us matrix-pipe floor: nothing overlaps, every tile waits out an
// HBM latency.  Here a workgroup owns UNITS of 256 consecutive output positions (BH output rows of one output plane); the
// 4 x (2 BH + 2) input rows a unit needs are copied into LDS by 16-byte coalesced loads issued a whole unit (8 tiles) ahead —
// rows / planes in the zero padding are out-of-range offsets, i.e. arrive as zeros — and the A operand of a tile is 16 LDS reads
// of two consecutive floats per lane (immediate offsets for (kd, kh)).  Everything else — B fragments in registers, the epilogue of
// tile t - 1 sliced between the MFMA groups of tile t, whole-line stores through a wave-private LDS tile — is the scheme above.
template <int IWT>
struct FwdLds {
    static constexpr int OWT = IWT / 2, BH = 256 / OWT, R = 2 * BH + 2, RS = IWT + 8, PLANE = R * RS, BUF = 4 * PLANE;
    static constexpr int PW = IWT / 4, PIECES = 4 * R * PW, NP = (PIECES + 255) / 256;
    static constexpr int TILE_ROWS = IWT == 32 ? 4 : 2;     // input rows between consecutive tiles of a unit
};
struct EdgeFwdLdsArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int OD, OH, IH, Cout, Cy, Cin_total;
    long x_sample;
    int units, units_per_wg, blocks_per_plane;   // blocks_per_plane = OH / BH
    FastDiv dbpp, dOD;
    int act;
    float slope;
};

template <int NT, int ACT, int IWT>
__global__ void __launch_bounds__(256) conv_fwd_c1_lds_kernel(EdgeFwdLdsArgs a) {
    using G = FwdLds<IWT>;
    constexpr int kWL = 65, kTL = 36;
    __shared__ __attribute__((aligned(16))) float xs[2 * G::BUF];
    __shared__ __attribute__((aligned(16))) float tl[4][32 * kTL];          // one staging tile of 32 channels per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kh2 = lane >> 5;
    int u = blockIdx.x * a.units_per_wg;
    const int u_end = min(a.units, u + a.units_per_wg);
    if (u >= u_end) return;
    // ---- staging bookkeeping: piece e = tid + 256 f is 16 bytes of staged row (plane p, row rr) ----
    const __amdgpu_buffer_rsrc_t xres = make_rsrc(a.x - ((long)a.IH * IWT + IWT));    // (see conv_fwd_c1_kernel: offsets stay >= 0)
    const __amdgpu_buffer_rsrc_t yres = make_rsrc(a.y);
    // one word per piece: byte offset inside the unit's box (< 2^26) | padding classes in bits 26..30: 1 plane 0 (padding when
    // od = 0), 2 plane 3 (od = OD - 1), 4 row 0 (first row block), 8 row R - 1 (last row block), 16 not a piece at all
    unsigned pword[G::NP];
    lds_f32x4* pdst[G::NP];
#pragma unroll
    for (int f = 0; f < G::NP; ++f) {
        const int e = tid + 256 * f;
        const int p = e / (G::R * G::PW), rem = e - p * (G::R * G::PW), rr = rem / G::PW, c4 = rem - rr * G::PW;
        const unsigned cls = e >= G::PIECES ? 16u : ((p == 0 ? 1u : 0u) | (p == 3 ? 2u : 0u) | (rr == 0 ? 4u : 0u) | (rr == G::R - 1 ? 8u : 0u));
        pword[f] = (unsigned)(((p * a.IH + rr) * IWT + 4 * c4) * 4) | (cls << 26);
        pdst[f] = (lds_f32x4*)((lds_float*)xs + (e >= G::PIECES ? 0 : p * G::PLANE + rr * G::RS + 4 + 4 * c4));
    }
    const unsigned O3 = (unsigned)(a.OD * a.OH * G::OWT);
    f32x4 sv[G::NP];
    unsigned ybase = 0;       // byte offset of the unit's first output position in channel row 0 of its sample
    auto stage_issue = [&](int uu, bool real) __attribute__((always_inline)) {
        uint32_t q, ohb, n, od;
        a.dbpp.divmod((uint32_t)uu, q, ohb);
        a.dOD.divmod(q, n, od);
        const unsigned mask = 0x03ffffffu | ((16u | (od == 0 ? 1u : 0u) | ((int)od == a.OD - 1 ? 2u : 0u) | (ohb == 0 ? 4u : 0u) |
                                              ((int)ohb == a.blocks_per_plane - 1 ? 8u : 0u)) << 26);
        const unsigned sbase = (unsigned)(((long)n * a.x_sample + ((long)(2 * od) * a.IH + 2 * ohb * G::BH) * IWT) * 4);
#pragma unroll
        for (int f = 0; f < G::NP; ++f) {
            const unsigned t = pword[f] & mask;
            sv[f] = buf_load4v(xres, (!real || (t >> 26) || (SG_FWDC1_ABL & 4)) ? kBufOutside : t, sbase);
        }
    };
    // the first unit's rows are requested before anything else: their HBM latency passes under the weight staging below
    stage_issue(u, true);
    // ---- weights -> B fragments (through xs, which is not in use yet) ----
    {
        float* wl = xs;
        const __amdgpu_buffer_rsrc_t wres = make_rsrc(a.w);
        f32x4 wv[NT * 2];
#pragma unroll
        for (int i = 0; i < NT * 2; ++i) {
            const int e4 = tid + 256 * i, co = e4 >> 4, t4 = e4 & 15;
            wv[i] = buf_load4v(wres, co < a.Cout ? (unsigned)(((long)co * a.Cin_total * 64 + t4 * 4) * 4) : kBufOutside, 0);
        }
#pragma unroll
        for (int i = 0; i < NT * 2; ++i) {
            const int e4 = tid + 256 * i, co = e4 >> 4, t4 = e4 & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) wl[co * kWL + t4 * 4 + j] = wv[i][j];
        }
    }
    __syncthreads();
    // lane half 0 owns the taps kw = 0, 1, lane half 1 the taps kw = 2, 3: two consecutive floats of the staged row
    float wfr[NT][16][2], bl[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 32 + r;
        bl[nt] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            wfr[nt][g][0] = xs[co * kWL + g * 4 + 2 * kh2];
            wfr[nt][g][1] = xs[co * kWL + g * 4 + 2 * kh2 + 1];
        }
    }
    __syncthreads();
    // the pad columns (w = -1 and w = IW) of every staged row stay zero for the whole kernel
    for (int e = tid; e < 2 * 4 * G::R * 2; e += 256) {
        const int row = e >> 1;
        xs[row * G::RS + ((e & 1) ? 4 + IWT : 3)] = 0.f;
    }
    auto stage_write = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < G::NP; ++f)
            if (f + 1 < G::NP || tid < G::PIECES - 256 * (G::NP - 1)) pdst[f][buf * (G::BUF / 4)] = sv[f];
    };
    auto unit_ybase = [&](int uu) __attribute__((always_inline)) {
        uint32_t q, ohb, n, od;
        a.dbpp.divmod((uint32_t)uu, q, ohb);
        a.dOD.divmod(q, n, od);
        return (unsigned)(((long)n * a.Cy * O3 + ((long)od * a.OH + ohb * G::BH) * G::OWT) * 4);
    };
    // A-operand address of this lane inside a staged buffer for tile 2 wave + j: immediate (kd, kh, j) offsets are added at the reads
    const int oh_in = IWT == 32 ? 2 * (r >> 4) : 0, ow = IWT == 32 ? (r & 15) : r;
    const lds_float* const abase = (const lds_float*)xs + (2 * wave * G::TILE_ROWS + oh_in) * G::RS + 3 + 2 * ow + 2 * kh2;
    const unsigned ylane = (unsigned)(((lane >> 3) * O3 + 64 * wave + 4 * (lane & 7)) * 4);   // row (lane >> 3), tile 2 wave, 16 bytes
    float* const tw = tl[wave];

    float pv[NT][16];
    unsigned yprev = kBufOutside;
    // epilogue of the previous tile in 16 slices: [nt = 0: 4 x registers -> LDS, 4 x LDS -> 8 channel rows of 128 B] [nt = 1: the same]
    auto epilogue_slice = [&](int sl) __attribute__((always_inline)) {
        const int nt = sl >> 3, k = sl & 7;
        if (nt >= NT) return;
        if (k < 4) {
            f32x4 v = {pv[nt][4 * k], pv[nt][4 * k + 1], pv[nt][4 * k + 2], pv[nt][4 * k + 3]};
            v = v + bl[nt];
            if (ACT == 1) {
                const f32x4 sv2 = v * a.slope;
                v = __builtin_elementwise_max(v, sv2);
            } else if (ACT == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = sg_apply_act(v[i], a.act, a.slope);
            }
            *reinterpret_cast<f32x4*>(tw + r * kTL + 8 * k + 4 * kh2) = v;
        } else {
            const int i = k - 4;
            const int row = nt * 32 + 8 * i + (lane >> 3);
            const f32x4 v = *reinterpret_cast<const f32x4*>(tw + (8 * i + (lane >> 3)) * kTL + 4 * (lane & 7));
            buf_store4(yres, row < a.Cout && !(SG_FWDC1_ABL & 1) ? yprev : kBufOutside, (unsigned)(nt * 32 + 8 * i) * O3 * 4u, v[0], v[1], v[2], v[3]);
        }
    };
    auto tile = [&](auto with_epilogue, int buf, int j) __attribute__((always_inline)) {
        const lds_float* ap = abase + buf * G::BUF + j * G::TILE_ROWS * G::RS;
        // operand values two groups ahead of their MFMAs (an LDS read is ~100 cycles, a group of 2 NT MFMAs 128 NT cycles)
        float ca[3], cb[3];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            ca[g] = ap[(g >> 2) * G::PLANE + (g & 3) * G::RS];
            cb[g] = ap[(g >> 2) * G::PLANE + (g & 3) * G::RS + 1];
        }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nt][q] = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g + 2 < 16) {
                ca[(g + 2) % 3] = ap[((g + 2) >> 2) * G::PLANE + ((g + 2) & 3) * G::RS];
                cb[(g + 2) % 3] = ap[((g + 2) >> 2) * G::PLANE + ((g + 2) & 3) * G::RS + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (SG_FWDC1_ABL & 2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][g] += ca[g % 3] * wfr[nt][g][0] + cb[g % 3] * wfr[nt][g][1];
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[g % 3], wfr[nt][g][0], acc[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[g % 3], wfr[nt][g][1], acc[nt], 0, 0, 0);
            }
            if (decltype(with_epilogue)::value) {
                __builtin_amdgcn_sched_barrier(0);
                epilogue_slice(g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[nt][q] = acc[nt][q];
    };

    stage_write(0);
    __syncthreads();
    int buf = 0;
    ybase = unit_ybase(u);
    // first unit: its first tile has no epilogue to carry
    stage_issue(u + 1, u + 1 < u_end);
    tile(IntTag<0>(), buf, 0);
    yprev = ybase + ylane;
    tile(IntTag<1>(), buf, 1);
    yprev = ybase + ylane + 128u;
    stage_write(buf ^ 1);
    __syncthreads();
    for (++u; u < u_end; ++u) {
        buf ^= 1;
        ybase = unit_ybase(u);
        stage_issue(u + 1, u + 1 < u_end);     // (unconditional: out-of-range offsets behind the last unit, see the vmcnt note above)
        tile(IntTag<1>(), buf, 0);
        yprev = ybase + ylane;
        tile(IntTag<1>(), buf, 1);
        yprev = ybase + ylane + 128u;
        stage_write(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) epilogue_slice(sl);
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------
constexpr int kEdgePartial = 4096 + 64;   // floats per workgroup partial
struct EdgeWgradArgs {
    const float* dy;    // [batch][Cy][O3]
    const float* y;     // FUSE: the layer's activated output, same layout as dy: dz = dy * act'(y) is formed on the fly
    float slope;
    const float* x;     // [batch][Cx][ID][IH][IW], channel 0 is read in place (no padded copy)
    float* partial;     // [gridDim.x][kEdgePartial]: 64 x 64 weight-gradient tile (+ 64 bias-gradient sums with FUSE)
    int OD, OH, OW, IH, IW, Cout, Cy;
    long x_sample;      // floats between samples of x (Cx * I3)
    int passes_per_sample, total_passes;   // a pass = 32 consecutive positions of one sample
    FastDiv dpps, dOW16, dOH;               // dOW16: OW / 16 row segments per row
};

// dy rows are staged through LDS: the MFMA wants A[row = channel][k = position] with only two k per instruction, so direct
// fragment loads would touch one 128-byte line per lane; instead the wave copies its [32 MT channels][32 positions] tile with
// fully coalesced 16-byte loads (8 lanes per 128-byte row segment), stores it as [row][36] (conflict-free for the fragment
// reads) and reads 16 consecutive positions per lane back as four ds_read_b128.
// FUSE (0: off, SG_ACT_LEAKY, SG_ACT_RELU): the incoming gradient is taken through the layer's activation here — dz = dy * act'(y),
// read from the sign of the activated output as sg_act_bwd does — and the bias gradient (channel sums of dz) comes out of the same
// pass.  For the critic's first layer, whose input needs no gradient, dz then never exists in memory: the separate activation
// backward (read dy + y, write dz: 402 MB at 128 x 64 x 16^3) and this kernel's read of dz become one read of dy + y.
#ifndef SG_WGRAD_C1_WAVES
#define SG_WGRAD_C1_WAVES 1   // (A/B: waves per SIMD the register allocation must admit)
#endif
template <int MT, int FUSE>
__global__ void __launch_bounds__(256, SG_WGRAD_C1_WAVES) conv_wgrad_c1_kernel(EdgeWgradArgs a) {
    constexpr int kLd = 36;                       // floats per staged row
    constexpr int kStage = MT * 32 * kLd;         // floats per wave
    __shared__ __attribute__((aligned(16))) float lds[4 * 4096];   // staging (<= 4 x 2304 floats), then the cross-wave sum
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kh2 = lane >> 5;
    const __amdgpu_buffer_rsrc_t dres = make_rsrc(a.dy);
    const __amdgpu_buffer_rsrc_t yres = make_rsrc(FUSE ? a.y : a.dy);
    // the patch operand comes straight from x: the resource starts (IH + 1) rows + 1 element before it, so that tap (0, 0, 0) is
    // offset 0; a lane (= tap) whose row falls into the zero padding for this pass's (od, oh), or whose column does at the
    // first / last position of a row, carries the out-of-range offset and receives 0 (nothing outside x is dereferenced)
    const __amdgpu_buffer_rsrc_t xres = make_rsrc(a.x - ((long)a.IH * a.IW + a.IW + 1));
    const unsigned O3 = (unsigned)(a.OD * a.OH * a.OW);
    float* stage = lds + wave * kStage;
    __shared__ float bsh[4][64];   // FUSE: per-wave channel sums
    // copy lane -> (row within a group of 8, 16-byte column)
    const int crow = lane >> 3, ccol = (lane & 7) * 4;
    unsigned tapoff[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int tap = nt * 32 + r, kd = tap >> 4, kh = (tap >> 2) & 3, kw = tap & 3;
        tapoff[nt] = (unsigned)((kd * a.IH + kh) * a.IW + kw) * 4u;
    }
    // this lane's taps: nt 0 -> kd in {0, 1}, nt 1 -> kd in {2, 3}; (kh, kw) are the same for both
    const int lkh = (r >> 2) & 3, lkw = r & 3, lkd0 = r >> 4;
    const int nseg = a.OW >> 4;
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;

    const int nwaves = gridDim.x * 4;
    f32x4 cv[MT * 4];     // this lane's pieces of the dy tile
    f32x4 yv[FUSE ? MT * 4 : 1];
    float rs[MT * 4];     // FUSE: running sums of dz over this lane's pieces (channel i*8 + crow)
#pragma unroll
    for (int i = 0; i < MT * 4; ++i) rs[i] = 0.f;
    float bv[2][16];      // the patch operand: 16 positions of this lane's two taps
    // The dy / y pieces of the next pass are requested before the MFMA block and land in cv / yv (which the staging copy has
    // just freed); the patch values of the next pass are requested INSIDE the block, each right behind the MFMA group that read
    // its register last, and the A fragments are read from the staging tile one 4-position group ahead.  (The first version
    // kept a second copy of the patch operand and all 32 A-fragment registers across the block: 290 registers with FUSE = one
    // wave per SIMD; a register cap instead of this restructuring spilled 56 of them and took 115 instead of 76 us.)
    unsigned xo[2], xl[2], xr[2];     // next pass's patch offsets: middle positions, first (left edge), last (right edge)
    // (valid = false: behind the last pass every offset is out of range and nothing is fetched.  NOT an `if (more)` around the
    // call: s_waitcnt vmcnt counts in issue order and at the join behind a skipped block of loads the compiler assumes the path
    // with the fewest younger loads in flight — every pass then began with vmcnt(0), i.e. waited for the dy / y pieces of the NEXT
    // pass it had just requested: no overlap of the 285 MB stream with the MFMAs at all, ISA listing of round 4)
    auto issue_dy = [&](int ps, bool valid) __attribute__((always_inline)) {
        uint32_t n, pp;
        a.dpps.divmod((uint32_t)(valid ? ps : 0), n, pp);
        const unsigned dbase = (unsigned)((long)n * a.Cy * O3 + pp * 32 + ccol) * 4u;
        // the invalid case as a bit OR-ed into the offsets (top bit set = out of range), hidden from the optimizer: written as a
        // select on `valid` the compiler unswitched it back into two copies of the loads under a branch
        unsigned inval = valid ? 0u : kBufOutside;
        asm volatile("" : "+v"(inval));
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            const int co = i * 8 + crow;
            const unsigned off = (co < a.Cout ? dbase + (unsigned)co * O3 * 4u : kBufOutside) | inval;
            cv[i] = buf_load4v(dres, off, 0);
            if (FUSE) yv[i] = buf_load4v(yres, off, 0);
        }
    };
    auto plan_x = [&](int ps, bool valid) __attribute__((always_inline)) {
        uint32_t n, pp, seg, q1, od, oh;
        a.dpps.divmod((uint32_t)ps, n, pp);
        const uint32_t p0 = pp * 32 + 16 * kh2;   // this lane half's 16 positions: one row segment (od, oh, ow0 .. ow0+15)
        a.dOW16.divmod(p0 >> 4, q1, seg);
        a.dOH.divmod(q1, od, oh);
        const unsigned xbase = (unsigned)((long)n * a.x_sample + ((long)(2 * od) * a.IH + 2 * oh) * a.IW + 32 * seg) * 4u;
        const bool hout = !valid || (lkh == 0 && oh == 0) || (lkh == 3 && (int)oh == a.OH - 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int kd = 2 * nt + lkd0;
            const bool out = hout || (kd == 0 && od == 0) || (kd == 3 && (int)od == a.OD - 1);
            xo[nt] = out ? kBufOutside : xbase + tapoff[nt];
            // w = 2 (16 seg + j) + kw - 1: left of the row for (kw 0, first position), right of it for (kw 3, last position)
            xl[nt] = (lkw == 0 && seg == 0) ? kBufOutside : xo[nt];
            xr[nt] = (lkw == 3 && (int)seg == nseg - 1) ? kBufOutside : xo[nt];
        }
    };
    auto load_x = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bv[nt][j] = buf_load(xres, j == 0 ? xl[nt] : (j == 15 ? xr[nt] : xo[nt]), 8u * j);
    };
    int ps = blockIdx.x * 4 + wave;
    if (ps < a.total_passes) {
        issue_dy(ps, true);
        plan_x(ps, true);
#pragma unroll
        for (int j = 0; j < 16; ++j) load_x(j);
    }
    for (; ps < a.total_passes; ps += nwaves) {
        if (FUSE) {
            const float neg = FUSE == SG_ACT_LEAKY ? a.slope : 0.f;
#pragma unroll
            for (int i = 0; i < MT * 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) cv[i][j] = yv[i][j] > 0.f ? cv[i][j] : cv[i][j] * neg;
                rs[i] += (cv[i][0] + cv[i][1]) + (cv[i][2] + cv[i][3]);
            }
        }
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) *reinterpret_cast<f32x4*>(stage + (i * 8 + crow) * kLd + ccol) = cv[i];
        const bool more = ps + nwaves < a.total_passes;
        issue_dy(ps + nwaves, more);              // next pass's dy / y fly during the MFMAs
        plan_x(more ? ps + nwaves : ps, more);    // (no next pass: every offset out of range, the loads below fetch nothing)
        f32x4 avc[MT], avn[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) avc[mt] = *reinterpret_cast<const f32x4*>(stage + (mt * 32 + r) * kLd + 16 * kh2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    avn[mt] = *reinterpret_cast<const f32x4*>(stage + (mt * 32 + r) * kLd + 16 * kh2 + 4 * (c + 1));
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * c + jj;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(avc[mt][jj], bv[nt][j], acc[mt][nt], 0, 0, 0);
                load_x(j);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) avc[mt] = avn[mt];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // cross-wave sum in a fixed order, then one partial tile per workgroup
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh2;
                lds[wave * 4096 + row * 64 + nt * 32 + r] = acc[mt][nt][q];
            }
    if (FUSE) {
        // the 8 lanes that share `crow` hold the pieces of one 128-byte row segment: sum them (quad swaps + half-row mirror)
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            float v = rs[i];
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
            if ((lane & 7) == 0) bsh[wave][i * 8 + crow] = v;
        }
    }
    __syncthreads();
    float* out = a.partial + (long)blockIdx.x * kEdgePartial;
    for (int e = threadIdx.x; e < MT * 32 * 64; e += 256)
        out[e] = (lds[e] + lds[4096 + e]) + (lds[2 * 4096 + e] + lds[3 * 4096 + e]);
    if (FUSE && threadIdx.x < MT * 32) out[4096 + threadIdx.x] = (bsh[0][threadIdx.x] + bsh[1][threadIdx.x]) + (bsh[2][threadIdx.x] + bsh[3][threadIdx.x]);
}

// dw[co][0][tap] = sum over workgroup partials.  256 threads = 16 partial groups x 16 outputs; every thread keeps 8
// independent loads in flight (a serial chain of 128 loads per thread took 32 us); fixed summation order.
// (blocks beyond the 256 of the weight tile: the 64 bias-gradient sums of the fused form -> db)
__global__ void __launch_bounds__(256) wgrad_c1_finalize_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                int nparts, int Cout, int Cin_total, float* __restrict__ db) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + o;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    for (int p0 = grp; p0 < nparts; p0 += 128) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 16 * u;
            s[u] += p < nparts ? partial[(long)p * kEdgePartial + e] : 0.f;
        }
    }
    red[grp][o] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (grp == 0) {
        float t = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) t += red[g2][o];
        const int co = e >> 6, tap = e & 63;
        if (e >= 4096) {
            if (e - 4096 < Cout) db[e - 4096] = t;
        } else if (co < Cout) {
            dw[((long)co * Cin_total) * 64 + tap] = t;
        }
    }
}

// ---- input gradient: tap planes + col2im ----------------------------------------------------------------------------------
struct TapPlaneArgs {
    const float* dy;   // [batch][Cy][O3]
    const float* w;    // [Cout][Cin_total][64], channel 0
    float* S;          // [batch][64][O3]
    int Cout, Cy, Cin_total;
    unsigned O3;
    int tiles_per_sample, total_tiles;
    FastDiv dtps;
};

template <int NS>   // channel pairs: Cout <= 2 NS
__global__ void __launch_bounds__(256) tapplane_gemm_kernel(TapPlaneArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kh2 = lane >> 5;
    // B fragments (columns = taps): wfr[nt][s] = W[co = 2s + kh2][0][tap = 32 nt + r]
    float wfr[2][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int co = 2 * s + kh2, coc = co < a.Cout ? co : a.Cout - 1;
        const float keep = co < a.Cout ? 1.f : 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wfr[nt][s] = keep * a.w[(long)coc * a.Cin_total * 64 + nt * 32 + r];
    }
    const __amdgpu_buffer_rsrc_t dres = make_rsrc(a.dy);
    const __amdgpu_buffer_rsrc_t sres = make_rsrc(a.S);
    const int nwaves = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + wave; t < a.total_tiles; t += nwaves) {
        uint32_t n, tp;
        a.dtps.divmod((uint32_t)t, n, tp);
        // A operand (rows = positions): dy[n][co = 2s + kh2][tile*32 + r]
        const unsigned doff = (unsigned)(((long)n * a.Cy + kh2) * a.O3 + tp * 32 + r) * 4u;
        float av[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) av[s] = buf_load(dres, 2 * s + kh2 < a.Cout ? doff : kBufOutside, (unsigned)(2 * s) * a.O3 * 4u);
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nt][q] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wfr[0][s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wfr[1][s], acc[1], 0, 0, 0);
        }
        // S[n][tap = 32 nt + r][tile*32 + 8c + 4 kh2 + (0..3)]
        const unsigned soff = (unsigned)(((long)n * 64 + r) * a.O3 + tp * 32 + 4 * kh2) * 4u;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                buf_store4(sres, soff, (unsigned)(nt * 32) * a.O3 * 4u + 32u * c, acc[nt][4 * c], acc[nt][4 * c + 1],
                           acc[nt][4 * c + 2], acc[nt][4 * c + 3]);
    }
}

// out[n][2q + p] = act(bias + sum over the 8 taps of parity p of S[n][k][q + delta]); per dimension: parity 0 takes tap 1 at
// q and tap 3 at q-1, parity 1 takes tap 0 at q+1 and tap 2 at q (conv3d.hip, dgrad_out1_kernel).  One thread per q.
__global__ void __launch_bounds__(256) col2im_c1_kernel(const float* __restrict__ S, const float* __restrict__ bias,
                                                        float* __restrict__ dx, ConvGeom g, long dx_sample, int total, int act,
                                                        float slope) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    int n, qd, qh, qw;
    decode_pos(g, (uint32_t)j, n, qd, qh, qw);
    const int OHW = g.OH * g.OW;
    const long O3 = (long)g.OD * OHW;
    const float* Sn = S + (long)n * 64 * O3;
    float acc[2][2][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) (&acc[0][0][0])[i] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
        const int pd = (kd & 1) ? 0 : 1, dd = (kd == 0) ? 1 : (kd == 3 ? -1 : 0);
        const int od = qd + dd;
        if ((unsigned)od >= (unsigned)g.OD) continue;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            const int ph = (kh & 1) ? 0 : 1, dh = (kh == 0) ? 1 : (kh == 3 ? -1 : 0);
            const int oh = qh + dh;
            if ((unsigned)oh >= (unsigned)g.OH) continue;
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int pw = (kw & 1) ? 0 : 1, dw_ = (kw == 0) ? 1 : (kw == 3 ? -1 : 0);
                const int ow = qw + dw_;
                if ((unsigned)ow >= (unsigned)g.OW) continue;
                acc[pd][ph][pw] += Sn[(long)(kd * 16 + kh * 4 + kw) * O3 + (long)od * OHW + oh * g.OW + ow];
            }
        }
    }
    const float b0 = bias ? bias[0] : 0.f;
    float* out = dx + (long)n * dx_sample;
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float2 o;
            o.x = sg_apply_act(acc[pd][ph][0] + b0, act, slope);
            o.y = sg_apply_act(acc[pd][ph][1] + b0, act, slope);
            *reinterpret_cast<float2*>(out + ((long)(2 * qd + pd) * g.IH + (2 * qh + ph)) * g.IW + 2 * qw) = o;
        }
}

// ---- input gradient, fused: tap groups in LDS ------------------------------------------------------------------------------
// One workgroup per (sample, qd): its two output d-planes 2qd, 2qd+1 need exactly four (kd, source plane) tap groups —
//   kd = 1, 2 of plane qd,  kd = 3 of plane qd-1,  kd = 0 of plane qd+1 — 16 (kh, kw) taps x OH*OW positions each (64 KB at
// 16 x 16).  They are computed here (every tap-plane element is consumed by exactly one workgroup, so nothing is computed
// twice), parked in LDS and gathered into the outputs: the 2 x 67 MB round trip of the tap planes through the Infinity Cache
// and the second launch are gone; dy is read three times but by neighbouring workgroups of the same XCD (blockIdx -> (sample,
// qd) keeps a sample's planes on one XCD's L2).
//   plane qd   : rows = 32 positions, cols = [kd 1 | kd 2] taps, v_mfma_f32_32x32x2_f32, K = channel pairs
//   planes qd-1 / qd+1 : only 16 columns are needed -> v_mfma_f32_16x16x4_f32 (rows = 16 positions, K = channel quads), so no
//                half-empty 32-wide tiles: 512 instead of 768 MFMA-equivalents per workgroup (the kernel is MFMA-bound otherwise)
struct ConvTFusedArgs {
    const float* dy;     // [batch][Cy][OD][P2]
    const float* w;      // [Cout][Cin_total][64], channel 0
    const float* bias;   // optional [1]
    float* dx;           // [batch][dx_sample], channel 0 written
    int Cout, Cy, Cin_total;
    int OD, OH, OW, P2;
    long dx_sample;
    int batch, act;
    float slope;
};
constexpr int kFusedTapStride = 260;                    // floats per tap row in LDS (256 positions + 4: b128 writes of 16 taps hit 64 banks)
constexpr int kFusedGroup = 16 * kFusedTapStride;       // one kd group

// Round 3, second form.  The first form asked for 96 loads per lane "up front", but with the weight fragments in 64 registers
// the compiler (128-register budget of two workgroups per CU) re-serialised them: 8 loads in flight, one more per MFMA — every
// MFMA waited for memory (ISA listing: buffer_load / s_waitcnt / v_mfma triples).  Here
//   * the weights live in LDS ([64 channels][kFusedWStride], zero rows beyond Cout: no masks, no clamps on the B side) and every
//     MFMA reads its B fragment from there: the 96 load destinations + accumulators fit the budget, so ALL loads of the workgroup
//     are in flight before the first MFMA (a sched_barrier keeps them there);
//   * the neighbour planes come first (their loads are issued first): their two tap groups go to LDS, are gathered into the output
//     registers, and the same 33 KB then take the plane's own two groups — 53 KB of LDS instead of 66, and the plane's 32 loads
//     are still landing while the neighbour planes compute.
constexpr int kFusedWStride = 80;   // floats per channel row of the LDS weight image: the four channel rows a 16x16x4 B fragment
                                    // reads (kq) fall into four disjoint quarters of the 64 banks
template <bool ALLCH>   // ALLCH: Cout == 64 (no channel clamps on the loads)
__global__ void __launch_bounds__(512, 4) convT_c1_fused_kernel(ConvTFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float S[];   // [2 slots][16 taps][kFusedTapStride] | weights [64][kFusedWStride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kh2 = lane >> 5;     // 32x32x2 fragments
    const int i16 = lane & 15, kq = lane >> 4;    // 16x16x4 fragments
    // XCD-aware decode: workgroup b runs on XCD b % 8; all planes of a sample stay on one XCD, neighbours in dispatch order
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int n = (j / a.OD) * 8 + xcd, qd = j % a.OD;
    if (n >= a.batch) return;     // (the whole workgroup)
    const int P2 = a.P2, npt = (P2 + 31) >> 5;    // <= 8 position tiles: one per wave
    const bool has_prev = qd > 0, has_next = qd + 1 < a.OD;
    const bool active = wave < npt;
    const int tp = wave;
    lds_float* const Sl = (lds_float*)S;
    lds_float* const Wl = Sl + 2 * kFusedGroup;
    typedef float f32x4v __attribute__((ext_vector_type(4)));

    // ---- every load of this workgroup: the weight image (64 x 64 floats, rows beyond Cout zero; coalesced rows, all eight
    // loads of a thread issued together), then the neighbour planes, then the plane ----
    float wv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 512 * k, co = e >> 6, tap = e & 63;
        wv[k] = co < a.Cout ? a.w[(long)co * a.Cin_total * 64 + tap] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t dres = make_rsrc(a.dy + (long)n * a.Cy * a.OD * P2);
    const unsigned chan = (unsigned)(a.OD * P2) * 4u;     // bytes between channels of a sample
    float bv[2][2][16];   // [plane: prev / next][16-position half][k step]
    float av[32];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const bool have = active && (pl == 0 ? has_prev : has_next);
        const int plane = pl == 0 ? qd - 1 : qd + 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pB = tp * 32 + h * 16 + i16;
            const unsigned offB = (have && pB < P2) ? (unsigned)((long)plane * P2 + pB) * 4u : kBufOutside;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (ALLCH)
                    bv[pl][h][s] = buf_load(dres, offB + (unsigned)kq * chan, (unsigned)(4 * s) * chan);
                else   // channels beyond Cout: the lane reads channel Cout-1 instead, its weight row is zero
                    bv[pl][h][s] = buf_load(dres, offB + (unsigned)min(4 * s + kq, a.Cout - 1) * chan, 0u);
            }
        }
    }
    {
        const int pA = tp * 32 + r;
        const unsigned offA = (active && pA < P2) ? (unsigned)((long)qd * P2 + pA) * 4u : kBufOutside;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (ALLCH)
                av[s] = buf_load(dres, offA + (unsigned)kh2 * chan, (unsigned)(2 * s) * chan);
            else
                av[s] = buf_load(dres, offA + (unsigned)min(2 * s + kh2, a.Cout - 1) * chan, 0u);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 512 * k;
        Wl[(e >> 6) * kFusedWStride + (e & 63)] = wv[k];
    }
    __syncthreads();

    // gather of one slot pair into the thread's 2 x 2 outputs: thread = (q position of the plane, output d-parity pd); slot pd
    // holds the tap group this parity takes in the current phase
    const int qi = tid & 255, pd = tid >> 8;
    const int qh = qi / a.OW, qw = qi - qh * a.OW;
    float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    auto gather = [&](bool have) {
        if (qi < P2 && have) {
            const lds_float* G = Sl + pd * kFusedGroup;
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int ph = (kh & 1) ? 0 : 1, dh = (kh == 0) ? 1 : (kh == 3 ? -1 : 0);
                const int oh = qh + dh;
                if ((unsigned)oh >= (unsigned)a.OH) continue;
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int pw = (kw & 1) ? 0 : 1, dw_ = (kw == 0) ? 1 : (kw == 3 ? -1 : 0);
                    const int ow = qw + dw_;
                    if ((unsigned)ow >= (unsigned)a.OW) continue;
                    o[ph][pw] += G[(kh * 4 + kw) * kFusedTapStride + oh * a.OW + ow];
                }
            }
        }
    };

    // ---- planes qd-1 (kd 3 -> slot 0) and qd+1 (kd 0 -> slot 1): [16 positions] x [16 taps], K = 4 channels per step ----
    if (active) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const bool have = pl == 0 ? has_prev : has_next;
            if (!have) continue;     // (workgroup-uniform)
            const lds_float* wb = Wl + kq * kFusedWStride + (pl == 0 ? 48 : 0) + i16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4v c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[pl][h][s], wb[4 * s * kFusedWStride], c4, 0, 0, 0);
                // column i16 = tap row, fragment rows 4 kq + (0..3) = positions
                lds_float* dst = Sl + pl * kFusedGroup + i16 * kFusedTapStride + tp * 32 + h * 16 + 4 * kq;
                *(__attribute__((address_space(3))) f32x4v*)dst = c4;
            }
        }
    }
    __syncthreads();
    // d-parity 0 takes kd 3 of plane qd-1, d-parity 1 takes kd 0 of plane qd+1
    gather(pd == 0 ? has_prev : has_next);
    __syncthreads();
    // ---- plane qd: [32 positions] x [kd 1 (slot 0) | kd 2 (slot 1)] ----
    if (active) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const lds_float* wa = Wl + kh2 * kFusedWStride + 16 + r;
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wa[2 * s * kFusedWStride], acc, 0, 0, 0);
        // column r = tap: slot r >> 4, tap row r & 15; rows of the fragment = positions 8c + 4 kh2 + (0..3)
        lds_float* dst = Sl + (r >> 4) * kFusedGroup + (r & 15) * kFusedTapStride + tp * 32 + 4 * kh2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4v v = {acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
            *(__attribute__((address_space(3))) f32x4v*)(dst + 8 * c) = v;
        }
    }
    __syncthreads();
    gather(true);     // d-parity 0: kd 1, d-parity 1: kd 2
    if (qi < P2) {
        const float b0 = a.bias ? a.bias[0] : 0.f;
        const int IH = 2 * a.OH, IW = 2 * a.OW;
        float* out = a.dx + (long)n * a.dx_sample;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float2 v;
            v.x = sg_apply_act(o[ph][0] + b0, a.act, a.slope);
            v.y = sg_apply_act(o[ph][1] + b0, a.act, a.slope);
            *reinterpret_cast<float2*>(out + ((long)(2 * qd + pd) * IH + (2 * qh + ph)) * IW + 2 * qw) = v;
        }
    }
}

// ---- input gradient, plane-streaming form (round 4) --------------------------------------------------------------------------
// The fused kernel above gives every (sample, plane) its own workgroup: all 512 resident workgroups load together, compute
// together, gather together, and each plane is fetched by three of them (0.30 of the HBM roofline, MFMA-busy 0.33: neither
// bandwidth- nor matrix-bound).  Here a workgroup owns one OUTPUT PARITY PAIR (pd, ph) of one sample and walks all its
// input planes qd = 0 .. OD-1:
//   * the output planes of d-parity pd take, per input plane, exactly two kd taps ("cur": kd = 1 + pd, lands on output plane
//     2 qd + pd; "far": kd = 3 (pd 0) -> output plane 2 qd + 2, kd = 0 (pd 1) -> output plane 2 qd - 1), output rows of h-parity ph
//     two kh taps, every kw: 16 of the 64 taps.  So the four workgroups of a sample split the tap planes without overlap — no
//     halo plane is recomputed, no partial result crosses workgroups — and each computes S[16 taps][P2 positions] per plane as
//     [16 positions] x [16 taps] x K = 4 channels v_mfma_f32_16x16x4_f32 tiles;
//   * the plane's 16 tap rows go to one of two LDS buffers (16.6 KB each), ONE barrier per plane, every thread gathers one output
//     (position q, column parity pw) from them: 4 reads for the cur taps, 4 for the far taps; the far (pd 0) / cur (pd 1) sum is
//     carried in a register to the next plane, where the output plane it belongs to is completed and stored as whole rows;
//   * the A fragments of plane qd + 1 (32 dword loads per lane) are requested before the MFMAs of plane qd: loads, matrix work
//     and the gather of consecutive planes overlap inside every workgroup, not by luck of the dispatch;
//   * a sample's activations are read by its four workgroups (same XCD, dispatched together: the L2 serves three of the four).
// Optional input transform (PRE): dy is taken through  act_in(dy * in_scale[c] + in_shift[c])  on the way into the fragments — a
// BatchNorm3d (+ LeakyReLU) between the producing layer and this one (model/gan.py:18-21) never becomes a pass of its own.
struct ConvTStreamArgs {
    const float* dy;     // [batch][Cy][OD][P2]
    const float* w;      // [Cout][Cin_total][64], channel 0
    const float* bias;   // optional [1]
    float* dx;           // [batch][dx_sample], channel 0 written
    const float* in_scale;   // PRE: [Cout] scale / shift of the input transform
    const float* in_shift;
    float in_slope;          // PRE: max(t, in_slope * t) after the affine map (LeakyReLU slope; ReLU 0; none 1)
    int spg;                 // samples per group: sample n uses in_scale / in_shift row n / spg ([groups][Cout]) ...
    long out_group_stride;   // ... and is written at dx + (n / spg) * out_group_stride + (n % spg) * dx_sample
    int Cout, Cy, Cin_total;
    int OD, OH, OW, P2;
    long dx_sample;
    int batch, act;
    float slope;
    int splits;              // 1, or 2: the input planes of a (sample, pd, ph) are walked by two workgroups, [0, OD/2) and [OD/2 - 1, OD)
};

// ALLCH: Cout == 64 (sixteen full k-steps, no channel clamps); FULL: P2 == 256 (every wave owns two whole position tiles).
// The plane loop is kept free of data-dependent branches between the prefetch and the MFMAs: s_waitcnt vmcnt counts loads in
// issue order, and at a control-flow join the compiler must assume the path with the FEWEST younger loads in flight — with an
// `if (more planes) prefetch` it waited for the just-issued loads of plane qd + 1 before the first MFMA of plane qd (ISA listing:
// vmcnt(31) .. vmcnt(0) instead of vmcnt(63) .. vmcnt(32)), i.e. no overlap at all.  The prefetch behind the last plane is issued
// anyway with an out-of-range scalar offset (returns zeros without touching memory).
#ifndef SG_CONVT_ABL
#define SG_CONVT_ABL 0   // ablation builds only (scripts/ab_build.sh): 1 no plane loads, 2 no MFMAs, 4 no global stores,
                         // 8 no gather epilogue, 16 no LDS tile writes, 32 no per-plane barrier, 64 no load instructions at all
#endif
template <bool ALLCH, bool PRE, bool FULL, int EPI>   // EPI: SG_ACT_NONE, SG_ACT_TANH (inlined, branch-free) or -1 (a.act at run time)
__global__ void __launch_bounds__(512, 2) convT_c1_stream_kernel(ConvTStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float S[];   // [2 buffers][16 taps][stride]
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    // XCD-aware decode: workgroup b runs on XCD b % 8; the four workgroups of a sample are neighbours in dispatch order on one XCD
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    // Plane split (round 5): at batch 64 the grid is 256 workgroups — ONE per CU, two waves per SIMD — and the plane walk is a
    // latency chain (prefetch distance one plane): with two workgroups per (sample, pd, ph) the CU holds two such chains.  The
    // second one starts one plane early (plane OD/2 - 1 again: only for its carried sum, nothing is stored for it).
    const int sub = a.splits == 2 ? (j & 7) : (j & 3), half = sub >> 2;
    const int n = (a.splits == 2 ? (j >> 3) : (j >> 2)) * 8 + xcd, pd = (sub >> 1) & 1, ph = sub & 1;
    if (n >= a.batch) return;     // (the whole workgroup)
    const int P2 = a.P2, OW = a.OW, OH = a.OH, OD = a.OD;
    const int qs = a.splits == 2 && half ? OD / 2 - 1 : 0, qe = a.splits == 2 && !half ? OD / 2 : OD;   // planes [qs, qe)
    // Positions are handled in blocks of 32 (one per wave): lane (i16, kq) loads TWO consecutive positions 2 i16, 2 i16 + 1 of
    // channel 4 s + kq with one 8-byte load — a wave instruction covers one whole 128-byte line of each of four channel rows (with
    // 4-byte loads it touched eight half lines for the same data and the texture addresser, not the matrix pipe, paced the plane)
    // — and the two components feed two MFMAs whose row r is position 2 r + j of the block.  The tap rows are stored in that
    // permuted order, [block][j][r], and the gather indexes them accordingly.
    const int nblocks = (P2 + 31) >> 5;                // <= 8
    const int stride = nblocks * 32 + 4;               // floats per tap row
    lds_float* const Sl = (lds_float*)S;
    const int nks = ALLCH ? 16 : (a.Cout + 3) >> 2;    // k-steps of 4 channels

    const __amdgpu_buffer_rsrc_t dres = make_rsrc_bytes(a.dy + (long)n * a.Cy * OD * P2, (long)a.Cy * OD * P2 * 4);
    const unsigned chan = (unsigned)(OD * P2) * 4u;     // bytes between channels of a sample
    // lane offset of the wave's block at plane 0 (out of range beyond the plane)
    const bool block_on = wave < nblocks;
    const int p0 = wave * 32 + 2 * i16;
    const unsigned voff = (block_on && p0 < P2) ? (unsigned)p0 * 4u + (unsigned)kq * chan : kBufOutside;
    // one k-step of a plane's A fragments (both MFMA tiles of the block); behind the last plane nothing is fetched (out-of-range
    // scalar offset)
    auto load_step = [&](int qd, int s, float (&dst)[2][16]) __attribute__((always_inline)) {
        const unsigned pshift = qd < qe && !(SG_CONVT_ABL & 1) ? (unsigned)(qd * P2) * 4u : kBufOutside;
        // (bit_cast the WHOLE result of the builtin: component access on its own vector type narrows the load to one dword)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v;
        if (ALLCH) {
            v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dres, (int)voff, (int)((unsigned)(4 * s) * chan + pshift), 0));
        } else {   // channels beyond Cout: the lane reads channel Cout-1 instead, its weight is zero
            const int co = min(4 * s + kq, a.Cout - 1);
            const unsigned off = voff == kBufOutside ? kBufOutside : voff - (unsigned)kq * chan + (unsigned)co * chan;
            v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dres, (int)(s < nks ? off : kBufOutside), (int)pshift, 0));
        }
        dst[0][s] = v.x;
        dst[1][s] = v.y;
    };
    float A0[2][16], A1[2][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) load_step(qs, s, A0);
    // B fragments: column i16 = tap (g2 = cur / far, khi = same row / neighbour row, kw), k row kq = channel 4 s + kq
    // (requested after plane 0's A fragments below have been: one memory round trip for both)
    const int g2 = i16 >> 3, khi = (i16 >> 2) & 1, kw = i16 & 3;
    const int kd = g2 == 0 ? (pd == 0 ? 1 : 2) : (pd == 0 ? 3 : 0);
    const int kh = khi == 0 ? (ph == 0 ? 1 : 2) : (ph == 0 ? 3 : 0);
    float wfr[16], psc[PRE ? 16 : 1], psh[PRE ? 16 : 1];
    {
        // branch-free: channels beyond Cout carry an out-of-range offset (weight, scale and shift read as 0)
        const __amdgpu_buffer_rsrc_t wres = make_rsrc(a.w);
        const long grow = (long)(n / a.spg) * a.Cout;     // this sample's group row of the input transform
        const __amdgpu_buffer_rsrc_t sres = make_rsrc(PRE ? a.in_scale + grow : a.w), hres = make_rsrc(PRE ? a.in_shift + grow : a.w);
        const unsigned wtap = (unsigned)(kd * 16 + kh * 4 + kw) * 4u, wrow = (unsigned)a.Cin_total * 256u;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int co = 4 * s + kq;
            const bool have = s < nks && co < a.Cout;
            wfr[s] = buf_load(wres, have ? (unsigned)co * wrow + wtap : kBufOutside, 0);
            if (PRE) {
                psc[s] = buf_load(sres, have ? (unsigned)co * 4u : kBufOutside, 0);
                psh[s] = buf_load(hres, have ? (unsigned)co * 4u : kBufOutside, 0);
            }
        }
    }
    // gather role of this thread: output (qh, qw, pw) of the plane pair row 2 qh + ph
    const int q = tid >> 1, pw = tid & 1;
    const int qh = q / OW, qw = q - qh * OW;
    const bool gather_on = q < P2;
    // the four (khi, kwi) taps of a group: row qh (+ dh for khi 1), column qw (+ dw for kwi 1)
    const int dh = ph == 0 ? -1 : 1, dw = pw == 0 ? -1 : 1;
    const int kw_same = pw == 0 ? 1 : 2, kw_nb = pw == 0 ? 3 : 0;
    const bool row_nb = (unsigned)(qh + dh) < (unsigned)OH, col_nb = (unsigned)(qw + dw) < (unsigned)OW;
    // LDS offsets (floats) of the cur group's taps (far group: + 8 * stride).  A tap whose source position lies outside the plane
    // reads the thread's own position instead and is multiplied by 0: no branches in the gather.
    const int qc = gather_on ? q : 0;
    const int qrow = row_nb ? dh * OW : 0, qcol = col_nb ? dw : 0;
    auto slot = [](int pos) { return (pos & ~31) + (pos & 1) * 16 + ((pos & 31) >> 1); };   // position -> index in a tap row
    int goff[4];
    goff[0] = (0 * 4 + kw_same) * stride + slot(qc);
    goff[1] = (0 * 4 + kw_nb) * stride + slot(qc + qcol);
    goff[2] = (1 * 4 + kw_same) * stride + slot(qc + qrow);
    goff[3] = (1 * 4 + kw_nb) * stride + slot(qc + qrow + qcol);
    const float gmul[4] = {1.f, col_nb ? 1.f : 0.f, row_nb ? 1.f : 0.f, (row_nb && col_nb) ? 1.f : 0.f};
    const float b0 = a.bias ? a.bias[0] : 0.f;
    const int IH = 2 * OH, IW = 2 * OW;
    // outputs leave through a raw buffer on the sample: lane offset = (row 2 qh + ph, column 2 qw + pw), scalar offset = output
    // plane; a lane / plane with nothing to store carries an out-of-range offset instead of a branch
    const __amdgpu_buffer_rsrc_t ores = make_rsrc(a.dx + (long)(n / a.spg) * a.out_group_stride + (long)(n % a.spg) * a.dx_sample);
    const unsigned ovoff = gather_on ? (unsigned)((2 * qh + ph) * IW + 2 * qw + pw) * 4u : kBufOutside;
    const unsigned oplane = (unsigned)(IH * IW) * 4u;
    float carry = 0.f;
    // Epilogue of plane p (its 16 tap rows are in LDS buffer p & 1, behind plane p's barrier): gather, complete one output plane
    // with the sum carried from plane p - 1, activation, store, carry the other sum.  d-parity 0: output plane 2 p = carried far
    // taps (kd 3 of plane p - 1) + cur taps (kd 1); d-parity 1: output plane 2 p - 1 = carried cur taps (kd 2 of plane p - 1) + far
    // taps (kd 0), nothing to store at p = 0.  It is issued in three slices INSIDE plane p + 1's MFMA loop (the matrix pipe runs
    // 32 cycles per MFMA during which the wave is free to issue LDS / VALU / VMEM work): with the epilogue behind the barrier the
    // pipe idled a third of every plane (25.4 us at 64 samples).  Everything in it is branch-free — selects on the uniform
    // parities, the tanh as exp / rcp with a series near 0 — so the loop stays one basic block and the compiler's s_waitcnt
    // bookkeeping exact.
    float etc[4], etf[4], eval = 0.f;
    auto epilogue_slice = [&](int p, int slice) __attribute__((always_inline)) {
        const lds_float* pb = Sl + (p & 1) * 16 * stride;
        if (slice == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                etc[g] = pb[goff[g]];
                etf[g] = pb[goff[g] + 8 * stride];
            }
        } else if (slice == 1) {
            const float sc = (etc[0] + etc[1] * gmul[1]) + (etc[2] * gmul[2] + etc[3] * gmul[3]);
            const float sf = (etf[0] + etf[1] * gmul[1]) + (etf[2] * gmul[2] + etf[3] * gmul[3]);
            const float fin = pd == 0 ? sc : sf, keep = pd == 0 ? sf : sc;
            float v = carry + fin + b0;
            carry = keep;
            if (EPI == SG_ACT_TANH) {
                const float ax = fabsf(v), x2 = v * v;
                const float e = __expf(2.f * ax);
                const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);                       // |err| ~ 1e-7
                const float small = ax * (1.f + x2 * (-0.33333334f + x2 * 0.13333334f));          // |v| < 0.06: rel. err < 1e-8
                v = copysignf(ax < 0.06f ? small : big, v);
            } else if (EPI != SG_ACT_NONE) {
                v = sg_apply_act(v, a.act, a.slope);
            }
            eval = v;
        } else {
            const bool skip = (p == qs && (pd == 1 || qs > 0)) || (SG_CONVT_ABL & 4);    // nothing complete yet at the first plane of a walk
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, eval), ores, (int)(skip ? kBufOutside : ovoff),
                                                  (int)((unsigned)(2 * p - pd) * oplane), 0);
        }
    };

    auto plane = [&](int qd, auto with_epi, float (&cur)[2][16], float (&nxt)[2][16]) __attribute__((always_inline)) {
        // The loads of plane qd + 1 are issued BETWEEN the MFMAs of plane qd, one k-step (two dword loads) per MFMA pair: all
        // eight waves of the workgroup run in lockstep (one barrier per plane), so a block of 32 loads per wave up front was a
        // phase in which the texture addresser worked and the matrix pipe idled (first version: 36.8 us at 64 samples, no better
        // than one workgroup per plane).
        lds_float* const buf = Sl + (qd & 1) * 16 * stride;
        f32x4v c4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (!(SG_CONVT_ABL & 64)) load_step(qd + 1, s, nxt);
            if (decltype(with_epi)::value && !(SG_CONVT_ABL & 8)) {
                // The barrier that publishes plane qd - 1's tap rows sits HERE, one MFMA pair into plane qd, not behind the LDS
                // writes at the end of plane qd - 1: the drain of that plane's last MFMAs, the write latency and the arrival
                // skew of the eight waves then pass under this plane's first MFMAs (counters with the barrier at the end of the
                // plane: matrix pipe 51 % busy, 30 % of the wave cycles parked).  Safe with two buffers: a wave writes buffer b
                // again only at the end of the plane after next, behind a barrier every reader of b has passed.
                if (s == 1) {
                    if (!(SG_CONVT_ABL & 32)) __syncthreads();
                    epilogue_slice(qd - 1, 0);
                }
                if (s == 5) epilogue_slice(qd - 1, 1);
                if (s == 9) epilogue_slice(qd - 1, 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v = cur[t][s];
                if (PRE) {
                    v = fmaf(v, psc[s], psh[s]);
                    v = fmaxf(v, v * a.in_slope);     // LeakyReLU with 0 <= slope <= 1 (ReLU: 0, none: 1) as max(t, slope t)
                }
                if (SG_CONVT_ABL & 2) c4[t][s & 3] += v * wfr[s];
                else c4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, wfr[s], c4[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // column i16 = tap row, fragment rows r = 4 kq + (0..3) of tile j = positions 2 r + j of the block: slots j * 16 + r
        if ((FULL || block_on) && !(SG_CONVT_ABL & 16)) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                *(__attribute__((address_space(3))) f32x4v*)(buf + i16 * stride + wave * 32 + t * 16 + 4 * kq) = c4[t];
        } else if (SG_CONVT_ABL & 16) {
            asm volatile("" ::"v"(c4[0]), "v"(c4[1]));       // (keep the accumulators alive)
        }
    };
    plane(qs, IntTag<0>(), A0, A1);
    int qd = qs + 1;
    // (buffer parity = plane parity; the register sets alternate from the walk's first plane)
    for (; qd + 1 < qe; qd += 2) {
        plane(qd, IntTag<1>(), A1, A0);
        plane(qd + 1, IntTag<1>(), A0, A1);
    }
    if (qd < qe) plane(qd, IntTag<1>(), A1, A0);
    // the last plane's epilogue, and for d-parity 1 the output plane 2 OD - 1 (cur taps of the last plane alone)
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) epilogue_slice(qe - 1, sl);
    if (pd == 1 && qe == OD) {
        float v = carry + b0;
        v = EPI == SG_ACT_NONE ? v : sg_apply_act(v, a.act, a.slope);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ores, (int)ovoff, (int)((unsigned)(2 * OD - 1) * oplane), 0);
    }
}

// convT_c1_stream2_kernel (round 5): the same plane walk with BOTH h parities in one workgroup.  The ablations of the kernel above
// (profiles/r05_convT_c1_ablation.json) show its load stream — 16 eight-byte load instructions per wave and plane, the same 64 KB
// plane pulled through the texture addresser by each of the four (pd, ph) workgroups of a sample — NOT hiding under the MFMAs: 102 us
// at 256 samples, 61 without the load instructions, 55 for the MFMAs alone.  Here a workgroup owns (sample, pd): every A fragment it
// loads feeds the 16 taps of ph = 0 AND the 16 taps of ph = 1 (two B-fragment sets, 32 tap rows in LDS), a thread gathers one output of
// each h parity per plane: half the load instructions and half the L2 traffic per output, one barrier per plane for twice the outputs.
template <bool ALLCH, bool PRE, bool FULL, int EPI>
__global__ void __launch_bounds__(512, 1) convT_c1_stream2_kernel(ConvTStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float S[];   // [2 buffers][2 ph][16 taps][stride]
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    // XCD-aware decode: workgroup b runs on XCD b % 8; the four workgroups of a sample are neighbours in dispatch order on one XCD
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    // Plane split (round 5): at batch 64 the grid is 256 workgroups — ONE per CU, two waves per SIMD — and the plane walk is a
    // latency chain (prefetch distance one plane): with two workgroups per (sample, pd, ph) the CU holds two such chains.  The
    // second one starts one plane early (plane OD/2 - 1 again: only for its carried sum, nothing is stored for it).
    const int sub = a.splits == 2 ? (j & 3) : (j & 1), half = sub >> 1;
    const int n = (a.splits == 2 ? (j >> 2) : (j >> 1)) * 8 + xcd, pd = sub & 1;
    if (n >= a.batch) return;     // (the whole workgroup)
    const int P2 = a.P2, OW = a.OW, OH = a.OH, OD = a.OD;
    const int qs = a.splits == 2 && half ? OD / 2 - 1 : 0, qe = a.splits == 2 && !half ? OD / 2 : OD;   // planes [qs, qe)
    // Positions are handled in blocks of 32 (one per wave): lane (i16, kq) loads TWO consecutive positions 2 i16, 2 i16 + 1 of
    // channel 4 s + kq with one 8-byte load — a wave instruction covers one whole 128-byte line of each of four channel rows (with
    // 4-byte loads it touched eight half lines for the same data and the texture addresser, not the matrix pipe, paced the plane)
    // — and the two components feed two MFMAs whose row r is position 2 r + j of the block.  The tap rows are stored in that
    // permuted order, [block][j][r], and the gather indexes them accordingly.
    const int nblocks = (P2 + 31) >> 5;                // <= 8
    const int stride = nblocks * 32 + 4;               // floats per tap row
    lds_float* const Sl = (lds_float*)S;
    const int nks = ALLCH ? 16 : (a.Cout + 3) >> 2;    // k-steps of 4 channels

    const __amdgpu_buffer_rsrc_t dres = make_rsrc_bytes(a.dy + (long)n * a.Cy * OD * P2, (long)a.Cy * OD * P2 * 4);
    const unsigned chan = (unsigned)(OD * P2) * 4u;     // bytes between channels of a sample
    // lane offset of the wave's block at plane 0 (out of range beyond the plane)
    const bool block_on = wave < nblocks;
    const int p0 = wave * 32 + 2 * i16;
    const unsigned voff = (block_on && p0 < P2) ? (unsigned)p0 * 4u + (unsigned)kq * chan : kBufOutside;
    // one k-step of a plane's A fragments (both MFMA tiles of the block); behind the last plane nothing is fetched (out-of-range
    // scalar offset)
    auto load_step = [&](int qd, int s, float (&dst)[2][16]) __attribute__((always_inline)) {
        const unsigned pshift = qd < qe ? (unsigned)(qd * P2) * 4u : kBufOutside;
        // (bit_cast the WHOLE result of the builtin: component access on its own vector type narrows the load to one dword)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v;
        if (ALLCH) {
            v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dres, (int)voff, (int)((unsigned)(4 * s) * chan + pshift), 0));
        } else {   // channels beyond Cout: the lane reads channel Cout-1 instead, its weight is zero
            const int co = min(4 * s + kq, a.Cout - 1);
            const unsigned off = voff == kBufOutside ? kBufOutside : voff - (unsigned)kq * chan + (unsigned)co * chan;
            v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dres, (int)(s < nks ? off : kBufOutside), (int)pshift, 0));
        }
        dst[0][s] = v.x;
        dst[1][s] = v.y;
    };
    float A0[2][16], A1[2][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) load_step(qs, s, A0);
    // B fragments: column i16 = tap (g2 = cur / far, khi = same row / neighbour row, kw), k row kq = channel 4 s + kq
    // (requested after plane 0's A fragments below have been: one memory round trip for both)
    const int g2 = i16 >> 3, khi = (i16 >> 2) & 1, kw = i16 & 3;
    const int kd = g2 == 0 ? (pd == 0 ? 1 : 2) : (pd == 0 ? 3 : 0);
    const int khp[2] = {khi == 0 ? 1 : 3, khi == 0 ? 2 : 0};     // kh of this tap column for ph = 0 / 1
    float wfr[2][16], psc[PRE ? 16 : 1], psh[PRE ? 16 : 1];
    {
        // branch-free: channels beyond Cout carry an out-of-range offset (weight, scale and shift read as 0)
        const __amdgpu_buffer_rsrc_t wres = make_rsrc(a.w);
        const long grow = (long)(n / a.spg) * a.Cout;     // this sample's group row of the input transform
        const __amdgpu_buffer_rsrc_t sres = make_rsrc(PRE ? a.in_scale + grow : a.w), hres = make_rsrc(PRE ? a.in_shift + grow : a.w);
        const unsigned wtap0 = (unsigned)(kd * 16 + khp[0] * 4 + kw) * 4u, wtap1 = (unsigned)(kd * 16 + khp[1] * 4 + kw) * 4u;
        const unsigned wrow = (unsigned)a.Cin_total * 256u;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int co = 4 * s + kq;
            const bool have = s < nks && co < a.Cout;
            wfr[0][s] = buf_load(wres, have ? (unsigned)co * wrow + wtap0 : kBufOutside, 0);
            wfr[1][s] = buf_load(wres, have ? (unsigned)co * wrow + wtap1 : kBufOutside, 0);
            if (PRE) {
                psc[s] = buf_load(sres, have ? (unsigned)co * 4u : kBufOutside, 0);
                psh[s] = buf_load(hres, have ? (unsigned)co * 4u : kBufOutside, 0);
            }
        }
    }
    // gather role of this thread: output (qh, qw, pw) of the plane pair row 2 qh + ph
    const int q = tid >> 1, pw = tid & 1;
    const int qh = q / OW, qw = q - qh * OW;
    const bool gather_on = q < P2;
    // the four (khi, kwi) taps of a group: row qh (+ dh for khi 1), column qw (+ dw for kwi 1); per h parity
    const int dw = pw == 0 ? -1 : 1;
    const int kw_same = pw == 0 ? 1 : 2, kw_nb = pw == 0 ? 3 : 0;
    const bool col_nb = (unsigned)(qw + dw) < (unsigned)OW;
    const int qc = gather_on ? q : 0;
    const int qcol = col_nb ? dw : 0;
    auto slot = [](int pos) { return (pos & ~31) + (pos & 1) * 16 + ((pos & 31) >> 1); };   // position -> index in a tap row
    int goff[2][4];
    float gmul[2][4];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const int dh = ph == 0 ? -1 : 1;
        const bool row_nb = (unsigned)(qh + dh) < (unsigned)OH;
        const int qrow = row_nb ? dh * OW : 0;
        const int base = ph * 16 * stride;                 // this parity's 16 tap rows
        goff[ph][0] = base + (0 * 4 + kw_same) * stride + slot(qc);
        goff[ph][1] = base + (0 * 4 + kw_nb) * stride + slot(qc + qcol);
        goff[ph][2] = base + (1 * 4 + kw_same) * stride + slot(qc + qrow);
        goff[ph][3] = base + (1 * 4 + kw_nb) * stride + slot(qc + qrow + qcol);
        gmul[ph][0] = 1.f;
        gmul[ph][1] = col_nb ? 1.f : 0.f;
        gmul[ph][2] = row_nb ? 1.f : 0.f;
        gmul[ph][3] = (row_nb && col_nb) ? 1.f : 0.f;
    }
    const float b0 = a.bias ? a.bias[0] : 0.f;
    const int IH = 2 * OH, IW = 2 * OW;
    // outputs leave through a raw buffer on the sample: lane offset = (row 2 qh + ph, column 2 qw + pw), scalar offset = output
    // plane; a lane / plane with nothing to store carries an out-of-range offset instead of a branch
    const __amdgpu_buffer_rsrc_t ores = make_rsrc(a.dx + (long)(n / a.spg) * a.out_group_stride + (long)(n % a.spg) * a.dx_sample);
    const unsigned ovoff = gather_on ? (unsigned)((2 * qh) * IW + 2 * qw + pw) * 4u : kBufOutside;      // row of ph = 0; ph = 1: + IW
    const unsigned oplane = (unsigned)(IH * IW) * 4u;
    float carry[2] = {0.f, 0.f};
    // Epilogue of plane p (its 16 tap rows are in LDS buffer p & 1, behind plane p's barrier): gather, complete one output plane
    // with the sum carried from plane p - 1, activation, store, carry the other sum.  d-parity 0: output plane 2 p = carried far
    // taps (kd 3 of plane p - 1) + cur taps (kd 1); d-parity 1: output plane 2 p - 1 = carried cur taps (kd 2 of plane p - 1) + far
    // taps (kd 0), nothing to store at p = 0.  It is issued in three slices INSIDE plane p + 1's MFMA loop (the matrix pipe runs
    // 32 cycles per MFMA during which the wave is free to issue LDS / VALU / VMEM work): with the epilogue behind the barrier the
    // pipe idled a third of every plane (25.4 us at 64 samples).  Everything in it is branch-free — selects on the uniform
    // parities, the tanh as exp / rcp with a series near 0 — so the loop stays one basic block and the compiler's s_waitcnt
    // bookkeeping exact.
    float etc[2][4], etf[2][4], eval[2] = {0.f, 0.f};
    auto epilogue_slice = [&](int p, int slice) __attribute__((always_inline)) {
        const lds_float* pb = Sl + (p & 1) * 32 * stride;
        if (slice == 0) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    etc[ph][g] = pb[goff[ph][g]];
                    etf[ph][g] = pb[goff[ph][g] + 8 * stride];
                }
        } else if (slice == 1) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const float sc = (etc[ph][0] + etc[ph][1] * gmul[ph][1]) + (etc[ph][2] * gmul[ph][2] + etc[ph][3] * gmul[ph][3]);
                const float sf = (etf[ph][0] + etf[ph][1] * gmul[ph][1]) + (etf[ph][2] * gmul[ph][2] + etf[ph][3] * gmul[ph][3]);
                const float fin = pd == 0 ? sc : sf, keep = pd == 0 ? sf : sc;
                float v = carry[ph] + fin + b0;
                carry[ph] = keep;
                if (EPI == SG_ACT_TANH) {
                    const float ax = fabsf(v), x2 = v * v;
                    const float e = __expf(2.f * ax);
                    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);                       // |err| ~ 1e-7
                    const float small = ax * (1.f + x2 * (-0.33333334f + x2 * 0.13333334f));          // |v| < 0.06: rel. err < 1e-8
                    v = copysignf(ax < 0.06f ? small : big, v);
                } else if (EPI != SG_ACT_NONE) {
                    v = sg_apply_act(v, a.act, a.slope);
                }
                eval[ph] = v;
            }
        } else {
            const bool skip = (p == qs && (pd == 1 || qs > 0)) || (SG_CONVT_ABL & 4);    // nothing complete yet at the first plane of a walk
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, eval[ph]), ores, (int)(skip ? kBufOutside : ovoff),
                                                      (int)((unsigned)(2 * p - pd) * oplane + (unsigned)(ph * IW) * 4u), 0);
        }
    };

    auto plane = [&](int qd, auto with_epi, float (&cur)[2][16], float (&nxt)[2][16]) __attribute__((always_inline)) {
        // The loads of plane qd + 1 are issued BETWEEN the MFMAs of plane qd, one k-step (two dword loads) per MFMA pair: all
        // eight waves of the workgroup run in lockstep (one barrier per plane), so a block of 32 loads per wave up front was a
        // phase in which the texture addresser worked and the matrix pipe idled (first version: 36.8 us at 64 samples, no better
        // than one workgroup per plane).
        lds_float* const buf = Sl + (qd & 1) * 32 * stride;
        f32x4v c4[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            load_step(qd + 1, s, nxt);
            if (decltype(with_epi)::value) {
                // The barrier that publishes plane qd - 1's tap rows sits HERE, one MFMA pair into plane qd, not behind the LDS
                // writes at the end of plane qd - 1: the drain of that plane's last MFMAs, the write latency and the arrival
                // skew of the eight waves then pass under this plane's first MFMAs (counters with the barrier at the end of the
                // plane: matrix pipe 51 % busy, 30 % of the wave cycles parked).  Safe with two buffers: a wave writes buffer b
                // again only at the end of the plane after next, behind a barrier every reader of b has passed.
                if (s == 1) {
                    __syncthreads();
                    epilogue_slice(qd - 1, 0);
                }
                if (s == 5) epilogue_slice(qd - 1, 1);
                if (s == 9) epilogue_slice(qd - 1, 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v = cur[t][s];
                if (PRE) {
                    v = fmaf(v, psc[s], psh[s]);
                    v = fmaxf(v, v * a.in_slope);     // LeakyReLU with 0 <= slope <= 1 (ReLU: 0, none: 1) as max(t, slope t)
                }
                c4[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, wfr[0][s], c4[t][0], 0, 0, 0);
                c4[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, wfr[1][s], c4[t][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // column i16 = tap row, fragment rows r = 4 kq + (0..3) of tile j = positions 2 r + j of the block: slots j * 16 + r
        if (FULL || block_on) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
                    *(__attribute__((address_space(3))) f32x4v*)(buf + (ph * 16 + i16) * stride + wave * 32 + t * 16 + 4 * kq) = c4[t][ph];
        }
    };
    plane(qs, IntTag<0>(), A0, A1);
    int qd = qs + 1;
    // (buffer parity = plane parity; the register sets alternate from the walk's first plane)
    for (; qd + 1 < qe; qd += 2) {
        plane(qd, IntTag<1>(), A1, A0);
        plane(qd + 1, IntTag<1>(), A0, A1);
    }
    if (qd < qe) plane(qd, IntTag<1>(), A1, A0);
    // the last plane's epilogue, and for d-parity 1 the output plane 2 OD - 1 (cur taps of the last plane alone)
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) epilogue_slice(qe - 1, sl);
    if (pd == 1 && qe == OD) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float v = carry[ph] + b0;
            v = EPI == SG_ACT_NONE ? v : sg_apply_act(v, a.act, a.slope);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ores, (int)ovoff,
                                                  (int)((unsigned)(2 * OD - 1) * oplane + (unsigned)(ph * IW) * 4u), 0);
        }
    }
}

// convT_c1_all_kernel (round 6): the plane walk with ALL 64 TAPS in one workgroup — what the ablations of the two kernels above
// asked for (profiles/r05_convT_c1_ablation.json).  A workgroup owns (sample, a range of input planes): every A
// fragment it loads feeds four B-fragment sets, set = 2 pd + ph (the 16 taps of each output parity pair), 64 tap rows in LDS, and a
// thread gathers the four outputs (pd, ph) of its (position, column parity) per plane: a QUARTER of the load instructions and of
// the L2 traffic per output of convT_c1_stream_kernel, one barrier per plane for four times the outputs.  The plane range is what
// fills the chip: `splits` workgroups per sample walk OD / splits planes each, every one but the first starting one plane early
// for the carried sums (nothing is stored for that plane) — 1.25 x the loads and MFMAs at 4 splits, 1.0 x at one.
// The sums are those of the other forms, in the same order (same k order per tap, same gather expression): bit-identical outputs.
// LDS: tap row (g2, khi, kw) of a set sits at row 8 g2 + 4 khi + {kw 1: 0, kw 3: 1, kw 2: 2, kw 0: 3}: the two column parities of a
// gather instruction then read rows two apart (8 banks at a row pitch of 260 floats), i.e. the 32 lanes of a half-wave — 16
// positions x 2 column parities — hit 32 different banks (the kw-major order of the kernels above: rows one apart, 33 % conflicts).
#ifndef SG_CONVT_ALL_ILV
#define SG_CONVT_ALL_ILV 6      // other instructions asked for in front of each MFMA of a k-step (tuning)
#endif
template <bool ALLCH, bool PRE, bool FULL, int EPI>
__global__ void __launch_bounds__(512, 1) convT_c1_all_kernel(ConvTStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float S[];   // [2 buffers][4 sets][16 taps][stride]
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int part = j % a.splits, n = (j / a.splits) * 8 + xcd;
    if (n >= a.batch) return;     // (the whole workgroup)
    const int P2 = a.P2, OW = a.OW, OH = a.OH, OD = a.OD;
    const int pp = OD / a.splits;
    const int qs = part ? part * pp - 1 : 0, qe = (part + 1) * pp;   // planes [qs, qe); plane qs of a later part only feeds the carries
    const int nblocks = (P2 + 31) >> 5;                // <= 8
    const int stride = nblocks * 32 + 4;               // floats per tap row
    lds_float* const Sl = (lds_float*)S;
    const int nks = ALLCH ? 16 : (a.Cout + 3) >> 2;    // k-steps of 4 channels

    const __amdgpu_buffer_rsrc_t dres = make_rsrc_bytes(a.dy + (long)n * a.Cy * OD * P2, (long)a.Cy * OD * P2 * 4);
    const unsigned chan = (unsigned)(OD * P2) * 4u;     // bytes between channels of a sample
    const bool block_on = wave < nblocks;
    const int p0 = wave * 32 + 2 * i16;
    const unsigned voff = (block_on && p0 < P2) ? (unsigned)p0 * 4u + (unsigned)kq * chan : kBufOutside;
    auto load_step = [&](int qd, int s, float (&dst)[2][16]) __attribute__((always_inline)) {
        const unsigned pshift = qd < qe ? (unsigned)(qd * P2) * 4u : kBufOutside;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v;
        if (ALLCH) {
            v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dres, (int)voff, (int)((unsigned)(4 * s) * chan + pshift), 0));
        } else {   // channels beyond Cout: the lane reads channel Cout-1 instead, its weight is zero
            const int co = min(4 * s + kq, a.Cout - 1);
            const unsigned off = voff == kBufOutside ? kBufOutside : voff - (unsigned)kq * chan + (unsigned)co * chan;
            v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dres, (int)(s < nks ? off : kBufOutside), (int)pshift, 0));
        }
        dst[0][s] = v.x;
        dst[1][s] = v.y;
    };
    float A0[2][16], A1[2][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) load_step(qs, s, A0);
    // B fragments: column i16 = tap row 8 g2 + 4 khi + kwpos of each of the four sets, k row kq = channel 4 s + kq
    const int g2 = i16 >> 3, khi = (i16 >> 2) & 1, kwpos = i16 & 3;
    const int kw = kwpos == 0 ? 1 : (kwpos == 1 ? 3 : (kwpos == 2 ? 2 : 0));
    float wfr[4][16], psc[PRE ? 16 : 1], psh[PRE ? 16 : 1];
    {
        const __amdgpu_buffer_rsrc_t wres = make_rsrc(a.w);
        const long grow = (long)(n / a.spg) * a.Cout;     // this sample's group row of the input transform
        const __amdgpu_buffer_rsrc_t sres = make_rsrc(PRE ? a.in_scale + grow : a.w), hres = make_rsrc(PRE ? a.in_shift + grow : a.w);
        unsigned wtap[4];
#pragma unroll
        for (int set = 0; set < 4; ++set) {
            const int pd = set >> 1, ph = set & 1;
            const int kd = g2 == 0 ? (pd == 0 ? 1 : 2) : (pd == 0 ? 3 : 0);
            const int kh = khi == 0 ? (ph == 0 ? 1 : 2) : (ph == 0 ? 3 : 0);
            wtap[set] = (unsigned)(kd * 16 + kh * 4 + kw) * 4u;
        }
        const unsigned wrow = (unsigned)a.Cin_total * 256u;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int co = 4 * s + kq;
            const bool have = s < nks && co < a.Cout;
#pragma unroll
            for (int set = 0; set < 4; ++set) wfr[set][s] = buf_load(wres, have ? (unsigned)co * wrow + wtap[set] : kBufOutside, 0);
            if (PRE) {
                psc[s] = buf_load(sres, have ? (unsigned)co * 4u : kBufOutside, 0);
                psh[s] = buf_load(hres, have ? (unsigned)co * 4u : kBufOutside, 0);
            }
        }
    }
    // gather role of this thread: outputs (pd, ph) at (qh, qw, pw)
    const int q = tid >> 1, pw = tid & 1;
    const int qh = q / OW, qw = q - qh * OW;
    const bool gather_on = q < P2;
    const int dw = pw == 0 ? -1 : 1;
    const int kp_same = pw == 0 ? 0 : 2, kp_nb = pw == 0 ? 1 : 3;     // row positions of kw 1 / 2 (same column) and kw 3 / 0 (neighbour)
    const bool col_nb = (unsigned)(qw + dw) < (unsigned)OW;
    const int qc = gather_on ? q : 0;
    const int qcol = col_nb ? dw : 0;
    auto slot = [](int pos) { return (pos & ~31) + (pos & 1) * 16 + ((pos & 31) >> 1); };   // position -> index in a tap row
    int goff[2][4];
    float gmul[2][4];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const int dh = ph == 0 ? -1 : 1;
        const bool row_nb = (unsigned)(qh + dh) < (unsigned)OH;
        const int qrow = row_nb ? dh * OW : 0;
        const int base = ph * 16 * stride;                 // (set = 2 pd + ph: + pd * 32 * stride at the use)
        goff[ph][0] = base + (0 + kp_same) * stride + slot(qc);
        goff[ph][1] = base + (0 + kp_nb) * stride + slot(qc + qcol);
        goff[ph][2] = base + (4 + kp_same) * stride + slot(qc + qrow);
        goff[ph][3] = base + (4 + kp_nb) * stride + slot(qc + qrow + qcol);
        gmul[ph][0] = 1.f;
        gmul[ph][1] = col_nb ? 1.f : 0.f;
        gmul[ph][2] = row_nb ? 1.f : 0.f;
        gmul[ph][3] = (row_nb && col_nb) ? 1.f : 0.f;
    }
    const float b0 = a.bias ? a.bias[0] : 0.f;
    const int IH = 2 * OH, IW = 2 * OW;
    const __amdgpu_buffer_rsrc_t ores = make_rsrc(a.dx + (long)(n / a.spg) * a.out_group_stride + (long)(n % a.spg) * a.dx_sample);
    const unsigned ovoff = gather_on ? (unsigned)((2 * qh) * IW + 2 * qw + pw) * 4u : kBufOutside;      // row of ph = 0; ph = 1: + IW
    const unsigned oplane = (unsigned)(IH * IW) * 4u;
    float carry[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // Epilogue of plane p for d parity pd (the plane's 64 tap rows are in LDS buffer p & 1, behind plane p's barrier) — the
    // expressions of convT_c1_stream_kernel: pd 0 completes output plane 2 p = carried far taps (kd 3 of plane p - 1) + cur taps
    // (kd 1); pd 1 completes output plane 2 p - 1 = carried cur taps (kd 2 of plane p - 1) + far taps (kd 0).  Five slices inside
    // plane p + 1's MFMA loop: reads pd 0 | sums pd 0 | stores pd 0 + reads pd 1 | sums pd 1 | stores pd 1.
    float etc[2][4], etf[2][4], eval[2] = {0.f, 0.f};
    auto epi_read = [&](int p, int pd) __attribute__((always_inline)) {
        const lds_float* pb = Sl + ((p & 1) * 64 + pd * 32) * stride;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                etc[ph][g] = pb[goff[ph][g]];
                etf[ph][g] = pb[goff[ph][g] + 8 * stride];
            }
    };
    auto epi_sum = [&](int pd) __attribute__((always_inline)) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const float sc = (etc[ph][0] + etc[ph][1] * gmul[ph][1]) + (etc[ph][2] * gmul[ph][2] + etc[ph][3] * gmul[ph][3]);
            const float sf = (etf[ph][0] + etf[ph][1] * gmul[ph][1]) + (etf[ph][2] * gmul[ph][2] + etf[ph][3] * gmul[ph][3]);
            const float fin = pd == 0 ? sc : sf, keep = pd == 0 ? sf : sc;
            float v = carry[pd][ph] + fin + b0;
            carry[pd][ph] = keep;
            if (EPI == SG_ACT_TANH) {
                const float ax = fabsf(v), x2 = v * v;
                const float e = __expf(2.f * ax);
                const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);                       // |err| ~ 1e-7
                const float small = ax * (1.f + x2 * (-0.33333334f + x2 * 0.13333334f));          // |v| < 0.06: rel. err < 1e-8
                v = copysignf(ax < 0.06f ? small : big, v);
            } else if (EPI != SG_ACT_NONE) {
                v = sg_apply_act(v, a.act, a.slope);
            }
            eval[ph] = v;
        }
    };
    auto epi_store = [&](int p, int pd) __attribute__((always_inline)) {
        const bool skip = p == qs && (pd == 1 || qs > 0);    // nothing complete yet at the first plane of a walk
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, eval[ph]), ores, (int)(skip ? kBufOutside : ovoff),
                                                  (int)((unsigned)(2 * p - pd) * oplane + (unsigned)(ph * IW) * 4u), 0);
    };

    auto plane = [&](int qd, auto with_epi, float (&cur)[2][16], float (&nxt)[2][16]) __attribute__((always_inline)) {
        lds_float* const buf = Sl + (qd & 1) * 64 * stride;
        f32x4v c4[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int set = 0; set < 4; ++set) c4[t][set] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // One scheduling region per k-step: the next plane's A load, a slice of the previous plane's epilogue, the input
            // transform and the eight MFMAs.  The workgroup's two waves per SIMD move in lock-step (one barrier per plane, one
            // workgroup per CU), so nothing but this wave's own instruction order can put the epilogue's LDS reads / VALU / stores
            // UNDER the MFMAs: the sched_group_barrier sequence below asks for "a few other instructions, one MFMA" eight times
            // (with the slices in front of the MFMAs as a block the matrix pipe idled through every one of them: 92 us at 256
            // samples for 62 us of MFMAs).
            if (decltype(with_epi)::value && s == 1) {
                // (the barrier that publishes plane qd - 1's tap rows sits one k-step into plane qd: see convT_c1_stream_kernel)
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
            load_step(qd + 1, s, nxt);
            if (decltype(with_epi)::value) {
                if (s == 1) epi_read(qd - 1, 0);
                if (s == 3) epi_sum(0);
                if (s == 5) {
                    epi_store(qd - 1, 0);
                    epi_read(qd - 1, 1);
                }
                if (s == 7) epi_sum(1);
                if (s == 9) epi_store(qd - 1, 1);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v = cur[t][s];
                if (PRE) {
                    v = fmaf(v, psc[s], psh[s]);
                    v = fmaxf(v, v * a.in_slope);     // LeakyReLU with 0 <= slope <= 1 (ReLU: 0, none: 1) as max(t, slope t)
                }
#pragma unroll
                for (int set = 0; set < 4; ++set) c4[t][set] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, wfr[set][s], c4[t][set], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // the A load first
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x096, SG_CONVT_ALL_ILV, 0);   // VALU | SALU | VMEM | DS: a piece of everything else
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // column i16 = tap row, fragment rows r = 4 kq + (0..3) of tile j = positions 2 r + j of the block: slots j * 16 + r
        if (FULL || block_on) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int set = 0; set < 4; ++set)
                    *(__attribute__((address_space(3))) f32x4v*)(buf + (set * 16 + i16) * stride + wave * 32 + t * 16 + 4 * kq) = c4[t][set];
        }
    };
    plane(qs, IntTag<0>(), A0, A1);
    int qd = qs + 1;
    // (buffer parity = plane parity; the register sets alternate from the walk's first plane)
    for (; qd + 1 < qe; qd += 2) {
        plane(qd, IntTag<1>(), A1, A0);
        plane(qd + 1, IntTag<1>(), A0, A1);
    }
    if (qd < qe) plane(qd, IntTag<1>(), A1, A0);
    // the last plane's epilogue, and for d-parity 1 the output plane 2 OD - 1 (cur taps of the last plane alone)
    __syncthreads();
#pragma unroll
    for (int pd = 0; pd < 2; ++pd) {
        epi_read(qe - 1, pd);
        epi_sum(pd);
        epi_store(qe - 1, pd);
    }
    if (qe == OD) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float v = carry[1][ph] + b0;
            v = EPI == SG_ACT_NONE ? v : sg_apply_act(v, a.act, a.slope);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ores, (int)ovoff,
                                                  (int)((unsigned)(2 * OD - 1) * oplane + (unsigned)(ph * IW) * 4u), 0);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
size_t edge_fwd_workspace_bytes(int, int, int, int) { return 0; }   // the forward reads the grid in place
size_t edge_wgrad_workspace_bytes(int, int, int, int) { return (size_t)512 * kEdgePartial * sizeof(float); }   // partial tiles
size_t edge_dgrad_workspace_bytes(int batch, int OD, int OH, int OW) { return (size_t)batch * 64 * OD * OH * OW * sizeof(float); }

// Conv3d(1 -> Cout <= 64) forward.  Returns 1 if handled, 0 if not eligible.  Needs no workspace (the grid is read in place).
int edge_fwd_try(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                 const ConvGeom& g, int Cout, int act, float slope, void* workspace, size_t workspace_bytes,
                 hipStream_t stream, int force) {
    (void)workspace;
    (void)workspace_bytes;
    const long O3 = g.O3();
    if (Cin != 1 || Cout > 64 || O3 % 32 != 0) return 0;
    if (!force && (long)batch * O3 < 65536) return 0;   // small problems: the generic kernel's launch is as good
    // 32-bit buffer offsets: x (plus the (IH + 1) rows the resource starts before it) and y inside the 2 GiB window
    if (((size_t)batch * g.Cx * g.I3() + (size_t)(g.IH + 1) * g.IW + 4) * 4 >= (size_t)kBufRange ||
        (size_t)batch * g.Cy * O3 * 4 >= (size_t)kBufRange)
        return 0;
    // LeakyReLU runs as max(t, slope t): only for slopes in [0, 1]; anything else takes the generic epilogue
    const int actk = act == SG_ACT_NONE ? 0 : ((act == SG_ACT_LEAKY && slope >= 0.f && slope <= 1.f) ? 1 : 2);
    EdgeFwdArgs a;
    a.x = x;
    a.w = w;
    a.bias = bias;
    a.y = y;
    a.OD = g.OD;
    a.OH = g.OH;
    a.OW = g.OW;
    a.IH = g.IH;
    a.IW = g.IW;
    a.Cout = Cout;
    a.Cy = g.Cy;
    a.Cin_total = Cin_total;
    a.x_sample = (long)g.Cx * g.I3();
    a.tiles_per_sample = (int)(O3 / 32);
    a.total_tiles = batch * a.tiles_per_sample;
    a.dtps = FastDiv((uint32_t)a.tiles_per_sample);
    a.dOW = FastDiv((uint32_t)g.OW);
    a.dOH = FastDiv((uint32_t)g.OH);
    a.act = act;
    a.slope = slope;
    // round 5: the LDS-staged form for 32- / 64-wide grids (the critic's first layer, the progressive discriminator's 64^3 stage)
    {
        const int bh = g.IW == 32 ? 16 : 8;
        if (SG_FWD_C1_LDS && (g.IW == 32 || g.IW == 64) && g.OH % bh == 0) {
            EdgeFwdLdsArgs l;
            l.x = x;
            l.w = w;
            l.bias = bias;
            l.y = y;
            l.OD = g.OD;
            l.OH = g.OH;
            l.IH = g.IH;
            l.Cout = Cout;
            l.Cy = g.Cy;
            l.Cin_total = Cin_total;
            l.x_sample = (long)g.Cx * g.I3();
            l.blocks_per_plane = g.OH / bh;
            l.units = batch * g.OD * l.blocks_per_plane;
            int lwgs = l.units < 512 ? l.units : 512;
            l.units_per_wg = (l.units + lwgs - 1) / lwgs;
            lwgs = (l.units + l.units_per_wg - 1) / l.units_per_wg;
            l.dbpp = FastDiv((uint32_t)l.blocks_per_plane);
            l.dOD = FastDiv((uint32_t)g.OD);
            l.act = act;
            l.slope = slope;
#define SG_FWD_L(NT_, ACT_, IW_) hipLaunchKernelGGL((conv_fwd_c1_lds_kernel<NT_, ACT_, IW_>), dim3(lwgs), dim3(256), 0, stream, l)
#define SG_FWD_LA(NT_, IW_) do { if (actk == 0) SG_FWD_L(NT_, 0, IW_); else if (actk == 1) SG_FWD_L(NT_, 1, IW_); else SG_FWD_L(NT_, 2, IW_); } while (0)
            if (g.IW == 32) { if (Cout > 32) SG_FWD_LA(2, 32); else SG_FWD_LA(1, 32); }
            else            { if (Cout > 32) SG_FWD_LA(2, 64); else SG_FWD_LA(1, 64); }
#undef SG_FWD_LA
#undef SG_FWD_L
            return 1;
        }
    }
    int wgs = (a.total_tiles + 3) / 4;
    if (wgs > 512) wgs = 512;
#define SG_FWD_C1(NT_, ACT_) hipLaunchKernelGGL((conv_fwd_c1_kernel<NT_, ACT_>), dim3(wgs), dim3(256), 0, stream, a)
    if (Cout > 32) {
        if (actk == 0) SG_FWD_C1(2, 0); else if (actk == 1) SG_FWD_C1(2, 1); else SG_FWD_C1(2, 2);
    } else {
        if (actk == 0) SG_FWD_C1(1, 0); else if (actk == 1) SG_FWD_C1(1, 1); else SG_FWD_C1(1, 2);
    }
#undef SG_FWD_C1
    return 1;
}

// Conv3d(1 -> Cout <= 64) weight gradient (channel 0 of dw; the caller zeroes the rest when Cin_total > 1).
// y != NULL: fused activation backward (act = SG_ACT_LEAKY / SG_ACT_RELU), db receives the bias gradient
int edge_wgrad_try(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, const ConvGeom& g, int Cout,
                   void* workspace, size_t workspace_bytes, hipStream_t stream, int force, const float* y, int act, float slope,
                   float* db) {
    const long O3 = g.O3();
    if (Cin != 1 || Cout > 64 || g.OW % 16 != 0) return 0;
    if (y && ((act != SG_ACT_LEAKY && act != SG_ACT_RELU) || !db)) return 0;
    if (!force && (long)batch * O3 < 65536) return 0;
    if (!workspace || workspace_bytes < edge_wgrad_workspace_bytes(batch, g.OD, g.OH, g.OW)) return 0;
    if (((size_t)batch * g.Cx * g.I3() + (size_t)(g.IH + 1) * g.IW + 4) * 4 >= (size_t)kBufRange ||
        (size_t)batch * g.Cy * O3 * 4 >= (size_t)kBufRange)
        return 0;
    float* partial = (float*)workspace;
    EdgeWgradArgs a;
    a.dy = dy;
    a.y = y;
    a.slope = slope;
    a.x = x;
    a.partial = partial;
    a.OD = g.OD;
    a.OH = g.OH;
    a.OW = g.OW;
    a.IH = g.IH;
    a.IW = g.IW;
    a.Cout = Cout;
    a.Cy = g.Cy;
    a.x_sample = (long)g.Cx * g.I3();
    a.passes_per_sample = (int)(O3 / 32);
    a.total_passes = batch * a.passes_per_sample;
    a.dpps = FastDiv((uint32_t)a.passes_per_sample);
    a.dOW16 = FastDiv((uint32_t)(g.OW / 16));
    a.dOH = FastDiv((uint32_t)g.OH);
    int wgs = (a.total_passes + 3) / 4;
    if (wgs > 512) wgs = 512;
    const int fuse = y ? act : 0;
#define SG_LAUNCH_WGRAD_C1(MT, F) hipLaunchKernelGGL((conv_wgrad_c1_kernel<MT, F>), dim3(wgs), dim3(256), 0, stream, a)
    if (Cout > 32) {
        if (fuse == SG_ACT_LEAKY) SG_LAUNCH_WGRAD_C1(2, SG_ACT_LEAKY);
        else if (fuse == SG_ACT_RELU) SG_LAUNCH_WGRAD_C1(2, SG_ACT_RELU);
        else SG_LAUNCH_WGRAD_C1(2, 0);
    } else {
        if (fuse == SG_ACT_LEAKY) SG_LAUNCH_WGRAD_C1(1, SG_ACT_LEAKY);
        else if (fuse == SG_ACT_RELU) SG_LAUNCH_WGRAD_C1(1, SG_ACT_RELU);
        else SG_LAUNCH_WGRAD_C1(1, 0);
    }
#undef SG_LAUNCH_WGRAD_C1
    hipLaunchKernelGGL(wgrad_c1_finalize_kernel, dim3(fuse ? 260 : 256), dim3(256), 0, stream, (const float*)partial, dw, wgs, Cout,
                       Cin_total, db);
    return 1;
}

// Conv3d(1 -> Cout <= 64) input gradient = ConvTranspose3d(Cout -> 1) forward through the plane-streaming kernel; optionally with
// the input transform act_in(dy * in_scale[c] + in_shift[c]) folded into the loads (in_act: none / LeakyReLU / ReLU).
int edge_dgrad_stream_try(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin, int Cin_total,
                          const ConvGeom& g, int Cout, int act, float slope, hipStream_t stream, const float* in_scale,
                          const float* in_shift, int in_act, float in_slope, int samples_per_group, long out_group_stride, int form) {
    const long O3 = g.O3();
    if (Cin != 1 || Cout > 64 || g.OH * g.OW > 256 || (size_t)g.Cy * O3 * 4 >= (size_t)kBufRange) return 0;
    const bool pre = in_scale != nullptr;
    if (pre && (!in_shift || (in_act != SG_ACT_NONE && in_act != SG_ACT_LEAKY && in_act != SG_ACT_RELU) ||
                (in_act == SG_ACT_LEAKY && (in_slope < 0.f || in_slope > 1.f))))
        return 0;
    ConvTStreamArgs f;
    f.dy = dy;
    f.w = w;
    f.bias = bias;
    f.dx = dx;
    f.in_scale = in_scale;
    f.in_shift = in_shift;
    f.in_slope = in_act == SG_ACT_LEAKY ? in_slope : (in_act == SG_ACT_RELU ? 0.f : 1.f);
    f.Cout = Cout;
    f.Cy = g.Cy;
    f.Cin_total = Cin_total;
    f.OD = g.OD;
    f.OH = g.OH;
    f.OW = g.OW;
    f.P2 = g.OH * g.OW;
    f.dx_sample = (long)g.Cx * g.I3();
    f.batch = batch;
    f.act = act;
    f.slope = slope;
    f.spg = samples_per_group > 0 ? samples_per_group : batch;
    f.out_group_stride = samples_per_group > 0 ? out_group_stride : (long)batch * f.dx_sample;
    const int nblocks = (f.P2 + 31) / 32;
    // Both h parities per workgroup (convT_c1_stream2_kernel, round 5): half the plane loads per output.  Workgroups = samples x 2
    // (x 2 plane walks while that leaves CUs without one).
    // Measured (cold): 256 samples 100.5 -> 86.8 us; 64 samples 27.5 - 31.5 -> 25.9 us with two plane walks (the one-parity kernel
    // fills the chip with twice the workgroups there and is as fast warm): taken from 192 samples on — the grouped generator pass of
    // WGANTrainer.step runs at 256.
    // `form` (sg_convT3d_k4s2p1_to1_pre_impl; 0 = this dispatch rule): 1 / 2 = one h parity per workgroup with one / two plane walks,
    // 3 / 4 = both h parities with one / two walks, 5 = all 64 taps per workgroup with the plane range chosen here, 6 / 7 / 8 = the
    // same with 1 / 2 / 4 plane ranges per sample.  Tests and tuning select a form through the ABI, not through the environment.
    {
        // all taps per workgroup (convT_c1_all_kernel, round 6): enough plane ranges per sample for one workgroup per CU
        const long padded = (long)(batch + 7) / 8 * 8;
        int splits = form == 6 ? 1 : (form == 7 ? 2 : (form == 8 ? 4 : (padded >= 192 ? 1 : (padded >= 96 ? 2 : 4))));
        while (splits > 1 && (g.OD % splits != 0 || g.OD / splits < 2)) splits >>= 1;
        // Measured (cold, input transform + tanh; scripts/edge_cold.py convT_forms, profiles/r06_convT_forms.json): 256 samples
        // 128 (one parity) / 100 (both h parities) / 87 us (all taps, one plane range); 128 samples 62 / 52 / 52 (two ranges);
        // 64 samples 33 / 50 / 32 (four ranges).  TCC requests 1.16 x the algorithmic bytes (3.9 x / 2.1 x), no LDS bank
        // conflicts (33 %), MFMA busy 0.61 (0.44 / 0.53) — profiles/r06_convT_c1_counters.txt.  It wins where one plane range
        // per sample fills the chip; below that the one-parity kernel's 4 workgroups per sample are as fast: taken from 192 on.
        const bool all_default = SG_CONVT_ALL && padded >= 192;
        if (form >= 5 || (form == 0 && all_default)) {
            f.splits = splits;
            const size_t lds4 = (size_t)2 * 64 * (nblocks * 32 + 4) * sizeof(float);
            const unsigned wgs4 = (unsigned)(padded * splits);
            static SgPerDeviceOnce once4;
            if (once4.begin()) {      // 133 KB of dynamic LDS: the attribute, once per device, for every instantiation that may be launched
#define SG_ATTR4(ALL_, PRE_, FULL_, EPI_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convT_c1_all_kernel<ALL_, PRE_, FULL_, EPI_>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 260 * 4)
#define SG_ATTR4E(ALL_, PRE_, FULL_) do { SG_ATTR4(ALL_, PRE_, FULL_, SG_ACT_TANH); SG_ATTR4(ALL_, PRE_, FULL_, SG_ACT_NONE); SG_ATTR4(ALL_, PRE_, FULL_, -1); } while (0)
                SG_ATTR4E(true, true, true);
                SG_ATTR4E(true, false, true);
                SG_ATTR4E(false, true, false);
                SG_ATTR4E(false, false, false);
#undef SG_ATTR4E
#undef SG_ATTR4
                once4.end();
            }
#define SG_CONVT_ALLK(ALL_, PRE_, FULL_, EPI_) \
    hipLaunchKernelGGL((convT_c1_all_kernel<ALL_, PRE_, FULL_, EPI_>), dim3(wgs4), dim3(512), lds4, stream, f)
#define SG_CONVT_ALLK_EPI(ALL_, PRE_, FULL_)                          \
    do {                                                              \
        if (act == SG_ACT_TANH) SG_CONVT_ALLK(ALL_, PRE_, FULL_, SG_ACT_TANH); \
        else if (act == SG_ACT_NONE) SG_CONVT_ALLK(ALL_, PRE_, FULL_, SG_ACT_NONE); \
        else SG_CONVT_ALLK(ALL_, PRE_, FULL_, -1);                     \
    } while (0)
            if (Cout == 64 && f.P2 == 256) {
                if (pre) SG_CONVT_ALLK_EPI(true, true, true); else SG_CONVT_ALLK_EPI(true, false, true);
            } else {
                if (pre) SG_CONVT_ALLK_EPI(false, true, false); else SG_CONVT_ALLK_EPI(false, false, false);
            }
#undef SG_CONVT_ALLK_EPI
#undef SG_CONVT_ALLK
            return 1;
        }
    }
    const long base_wgs = (long)((batch + 7) / 8 * 8) * 2;
    const bool both = form ? form >= 3 : base_wgs >= 384;
    const bool can_split = g.OD >= 4 && g.OD % 2 == 0;
    if (both) {
        f.splits = (form ? form == 4 : base_wgs < 256) && can_split ? 2 : 1;
        const size_t lds2 = (size_t)2 * 32 * (nblocks * 32 + 4) * sizeof(float);
        const unsigned wgs2 = (unsigned)(base_wgs * f.splits);
        static SgPerDeviceOnce once2;
        if (once2.begin()) {      // 66.5 KB of dynamic LDS: the attribute, once per device, for every instantiation that may be launched
#define SG_ATTR2(ALL_, PRE_, FULL_, EPI_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convT_c1_stream2_kernel<ALL_, PRE_, FULL_, EPI_>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 260 * 4)
#define SG_ATTR2E(ALL_, PRE_, FULL_) do { SG_ATTR2(ALL_, PRE_, FULL_, SG_ACT_TANH); SG_ATTR2(ALL_, PRE_, FULL_, SG_ACT_NONE); SG_ATTR2(ALL_, PRE_, FULL_, -1); } while (0)
            SG_ATTR2E(true, true, true);
            SG_ATTR2E(true, false, true);
            SG_ATTR2E(false, true, false);
            SG_ATTR2E(false, false, false);
#undef SG_ATTR2E
#undef SG_ATTR2
            once2.end();
        }
#define SG_CONVT_STREAM2(ALL_, PRE_, FULL_, EPI_) \
    hipLaunchKernelGGL((convT_c1_stream2_kernel<ALL_, PRE_, FULL_, EPI_>), dim3(wgs2), dim3(512), lds2, stream, f)
#define SG_CONVT_STREAM2_EPI(ALL_, PRE_, FULL_)                          \
    do {                                                                 \
        if (act == SG_ACT_TANH) SG_CONVT_STREAM2(ALL_, PRE_, FULL_, SG_ACT_TANH); \
        else if (act == SG_ACT_NONE) SG_CONVT_STREAM2(ALL_, PRE_, FULL_, SG_ACT_NONE); \
        else SG_CONVT_STREAM2(ALL_, PRE_, FULL_, -1);                     \
    } while (0)
        if (Cout == 64 && f.P2 == 256) {
            if (pre) SG_CONVT_STREAM2_EPI(true, true, true); else SG_CONVT_STREAM2_EPI(true, false, true);
        } else {
            if (pre) SG_CONVT_STREAM2_EPI(false, true, false); else SG_CONVT_STREAM2_EPI(false, false, false);
        }
#undef SG_CONVT_STREAM2_EPI
#undef SG_CONVT_STREAM2
        return 1;
    }
    const size_t lds = (size_t)2 * 16 * (nblocks * 32 + 4) * sizeof(float);
    // Two walks per (sample, pd, ph) — measured in round 5 and NOT the default: at 64 samples (256 -> 512 workgroups) the cold time
    // went from 27.7 to 31.5 us, i.e. the plane walk is not a latency chain that a second workgroup per CU would hide (the four
    // workgroups of a sample already pull every input line through L2 four times; a second walk adds a recomputed plane to that);
    // only at 32 samples does it win (18.4 us against 25.6 unsplit and 19.4 - 20.9 for the per-plane kernel that serves < 48
    // samples).  Form 2 of sg_convT3d_k4s2p1_to1_pre_impl; tests/test_gpu_ops.py runs every form.
    f.splits = form == 2 && can_split ? 2 : 1;
    const unsigned wgs = (unsigned)((batch + 7) / 8 * 8 * 4 * f.splits);
#define SG_CONVT_STREAM(ALL_, PRE_, FULL_, EPI_) \
    hipLaunchKernelGGL((convT_c1_stream_kernel<ALL_, PRE_, FULL_, EPI_>), dim3(wgs), dim3(512), lds, stream, f)
#define SG_CONVT_STREAM_EPI(ALL_, PRE_, FULL_)                          \
    do {                                                                \
        if (act == SG_ACT_TANH) SG_CONVT_STREAM(ALL_, PRE_, FULL_, SG_ACT_TANH); \
        else if (act == SG_ACT_NONE) SG_CONVT_STREAM(ALL_, PRE_, FULL_, SG_ACT_NONE); \
        else SG_CONVT_STREAM(ALL_, PRE_, FULL_, -1);                     \
    } while (0)
    if (Cout == 64 && f.P2 == 256) {
        if (pre) SG_CONVT_STREAM_EPI(true, true, true); else SG_CONVT_STREAM_EPI(true, false, true);
    } else {
        if (pre) SG_CONVT_STREAM_EPI(false, true, false); else SG_CONVT_STREAM_EPI(false, false, false);
    }
#undef SG_CONVT_STREAM_EPI
#undef SG_CONVT_STREAM
    return 1;
}

// Conv3d(1 -> Cout <= 64) input gradient = ConvTranspose3d(Cout -> 1) forward: dx [batch][Cx][I3] (channel 0 written).
int edge_dgrad_try(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin, int Cin_total,
                   const ConvGeom& g, int Cout, int act, float slope, void* workspace, size_t workspace_bytes,
                   hipStream_t stream, int force) {
    const long O3 = g.O3();
    if (Cin != 1 || Cout > 64) return 0;
    // plane-streaming kernel (round 4); -DSG_NO_EDGE=16 builds restore the per-plane fused kernel below (A/B)
    constexpr bool stream_off = (SG_NO_EDGE & 16) != 0;
    // (the streaming kernel runs four workgroups per sample for the whole depth of the grid: below ~48 samples it leaves CUs
    // idle and the one-workgroup-per-plane kernel is faster — 17.8 vs 22.6 us at 32 samples, 32.0 vs 22.7 at 64, 114 vs 92 at 256)
    constexpr int stream_min_batch = SG_CONVT_MIN_BATCH;
    if (!stream_off && g.OH * g.OW <= 256 && (force || (long)batch * O3 >= 512) && batch >= stream_min_batch &&
        edge_dgrad_stream_try(dy, w, bias, dx, batch, Cin, Cin_total, g, Cout, act, slope, stream, nullptr, nullptr, 0, 0.f, 0, 0) == 1)
        return 1;
    // fused kernel: a whole (OH x OW) plane of the four tap groups fits in LDS
    constexpr bool fused_off = (SG_NO_EDGE & 8) != 0;
    if (!fused_off && g.OH * g.OW <= 256 && (size_t)g.Cy * O3 * 4 < (size_t)kBufRange && (force || (long)batch * O3 >= 512)) {
        ConvTFusedArgs f;
        f.dy = dy;
        f.w = w;
        f.bias = bias;
        f.dx = dx;
        f.Cout = Cout;
        f.Cy = g.Cy;
        f.Cin_total = Cin_total;
        f.OD = g.OD;
        f.OH = g.OH;
        f.OW = g.OW;
        f.P2 = g.OH * g.OW;
        f.dx_sample = (long)g.Cx * g.I3();
        f.batch = batch;
        f.act = act;
        f.slope = slope;
        const size_t lds = (size_t)(2 * kFusedGroup + 64 * kFusedWStride) * sizeof(float);
        static SgPerDeviceOnce attr_once;   // > 48 KB of dynamic LDS needs the attribute once per DEVICE
        if (attr_once.begin()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convT_c1_fused_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convT_c1_fused_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_once.end();
        }
        const unsigned wgs = (unsigned)((batch + 7) / 8 * 8 * g.OD);
        if (Cout == 64)
            hipLaunchKernelGGL((convT_c1_fused_kernel<true>), dim3(wgs), dim3(512), lds, stream, f);
        else
            hipLaunchKernelGGL((convT_c1_fused_kernel<false>), dim3(wgs), dim3(512), lds, stream, f);
        return 1;
    }
    if (O3 % 32 != 0) return 0;
    if (!force && (long)batch * O3 < 65536) return 0;
    if (!workspace || workspace_bytes < edge_dgrad_workspace_bytes(batch, g.OD, g.OH, g.OW)) return 0;
    if ((size_t)batch * 64 * O3 * 4 >= (size_t)kBufRange || (size_t)batch * g.Cy * O3 * 4 >= (size_t)kBufRange) return 0;
    TapPlaneArgs a;
    a.dy = dy;
    a.w = w;
    a.S = (float*)workspace;
    a.Cout = Cout;
    a.Cy = g.Cy;
    a.Cin_total = Cin_total;
    a.O3 = (unsigned)O3;
    a.tiles_per_sample = (int)(O3 / 32);
    a.total_tiles = batch * a.tiles_per_sample;
    a.dtps = FastDiv((uint32_t)a.tiles_per_sample);
    int wgs = (a.total_tiles + 3) / 4;
    if (wgs > 1024) wgs = 1024;
    if (Cout <= 16)
        hipLaunchKernelGGL((tapplane_gemm_kernel<8>), dim3(wgs), dim3(256), 0, stream, a);
    else if (Cout <= 32)
        hipLaunchKernelGGL((tapplane_gemm_kernel<16>), dim3(wgs), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((tapplane_gemm_kernel<32>), dim3(wgs), dim3(256), 0, stream, a);
    const int total = (int)((long)batch * O3);
    hipLaunchKernelGGL(col2im_c1_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, (const float*)workspace, bias, dx, g,
                       (long)g.Cx * g.I3(), total, act, slope);
    return 1;
}

}  // namespace sg
