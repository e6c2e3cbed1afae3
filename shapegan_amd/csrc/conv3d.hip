// shapegan_amd/csrc/conv3d.hip — Conv3d / ConvTranspose3d, kernel 4, stride 2, padding 1 (K1/K2).
//
// Replaces the ATen convolution / convolution_backward calls behind
//   nn.Conv3d(k=4,s=2,p=1)          model/gan.py:49-53, model/autoencoder.py:16-24, model/progressive_gan.py:38
//   nn.ConvTranspose3d(k=4,s=2,p=1) model/gan.py:13-21, model/autoencoder.py:55-63
// Three implicit-GEMM forms on the shared f32-MFMA tile skeleton (mfma_tile.h), all NCDHW fp32:
//
//   fwd   : y[n,co,o]   = b[co] + sum_{ci,tap} W[co,ci,tap] * x[n,ci,2o+tap-1]
//           GEMM  M=Cout  N=batch*O^3  K=Cin*64      (A = W row-major, B = input patches)
//   dgrad : dx[n,ci,2q+p] = sum_{co,t in {0,1}^3} W[co,ci,tap(p,t)] * dy[n,co,q+p-t]
//           8 output-parity classes p (no zero insertion), each a GEMM M=Cin N=batch*O^3 K=Cout*8
//           (A = parity-packed weights Wt[p][co*8+t][ci], B = dy patches)
//   wgrad : dW[co,ci,tap] = sum_{n,o} dy[n,co,o] * x[n,ci,2o+tap-1]
//           GEMM  M=Cout  N=Cin*64  K=batch*O^3, split-K over the batch*positions axis
//
// ConvTranspose3d is the adjoint: its forward is `dgrad`, its input gradient is `fwd`, and its
// weight gradient is `wgrad` with the roles of (x, dy) swapped; the ConvTranspose weight layout
// [Cin_T, Cout_T, 4,4,4] is exactly the Conv weight layout of the adjoint conv (Cout=Cin_T).
// Double backward (WGAN-GP, train_hybrid_progressive_gan.py:102-111) needs nothing else because the
// convolution is bilinear in (x, W).
#include "mfma_tile.h"
#include <stdlib.h>

#include "conv_common.h"
#include "../../include/shapegan_hip.h"

namespace sg {

// patch of x around output position pos=(n,od,oh,ow): element (ci,kd,kh,kw) = x[n,ci,2od-1+kd,2oh-1+kh,2ow-1+kw].
// `base` is the (possibly negative) element offset of tap (0,0,0) of channel 0; mask bits 0-3 / 4-7 / 8-11 say which
// kd / kh / kw taps fall inside the tensor.  Offsets are 32-bit: the entry points reject tensors >= 2^31 elements.
struct PatchCtx {
    int base;
    int mask;
    __device__ __forceinline__ void set(const ConvGeom& g, uint32_t pos) {
        int n, od, oh, ow;
        decode_pos(g, pos, n, od, oh, ow);
        const int id0 = 2 * od - 1, ih0 = 2 * oh - 1, iw0 = 2 * ow - 1;
        base = n * g.Cx * g.ID * g.IH * g.IW + (id0 * g.IH + ih0) * g.IW + iw0;
        int m = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((unsigned)(id0 + t) < (unsigned)g.ID) m |= 1 << t;
            if ((unsigned)(ih0 + t) < (unsigned)g.IH) m |= 1 << (4 + t);
            if ((unsigned)(iw0 + t) < (unsigned)g.IW) m |= 1 << (8 + t);
        }
        mask = m;
    }
};

// ---- fwd -------------------------------------------------------------------------------------
// B(k=(ci,kd,kh,kw), j=pos), lanes along positions.  A 16-aligned k-tile has ONE (ci,kd) and all 16 (kh,kw):
// per thread the (kh,kw) of its E elements never change, so their offsets/validity are precomputed once and the
// per-tile part (ci*I^3 + kd*IH*IW, the kd validity bit) is wave-uniform.
struct FwdPatchLoader {
    const float* x;
    ConvGeom g;
    int base, mask, tid_;
    int off[8];
    unsigned okbits, pend;
    template <int BR>
    __device__ void init(int tid, int j0, int N) {
        constexpr int E = BR * kBK / 256, STEP = 256 / BR;
        tid_ = tid;
        const int j = j0 + tid % BR, kq = tid / BR;
        PatchCtx c;
        c.set(g, (uint32_t)(j < N ? j : 0));
        base = c.base;
        mask = j < N ? c.mask : 0;
        okbits = 0;
#pragma unroll
        for (int it = 0; it < E; ++it) {
            const int kk = kq + it * STEP, kh = kk >> 2, kw = kk & 3;
            off[it] = kh * g.IW + kw;
            if (((mask >> (4 + kh)) & (mask >> (8 + kw)) & 1) != 0) okbits |= 1u << it;
        }
    }
    template <int BR>
    __device__ void load(int k0, int kend, float (&r)[BR * kBK / 256]) {
        const int ci = k0 >> 6, kd = (k0 >> 4) & 3;
        const int toff = base + (ci * g.ID + kd) * g.IH * g.IW;
        const unsigned ok = ((mask >> kd) & 1) ? okbits : 0u;
#pragma unroll
        for (int it = 0; it < BR * kBK / 256; ++it) r[it] = x[((ok >> it) & 1u) ? toff + off[it] : 0];
        pend = ok;
    }
    template <int BR, int LO, int HI>
    __device__ void store(float (*S)[BR + 1], const float (&r)[BR * kBK / 256]) const {
        StageRowFast<BR>::template store<LO, HI>(S, r, tid_, pend);
    }
};
struct FwdEpi {  // y[n][co][o] = act(v + bias[co])
    float* y;
    const float* bias;
    long O3;
    int Cy;
    FastDiv dO3;
    int act;
    float slope;
    struct Col {
        long off;
    };
    __device__ Col col(int j) const {
        uint32_t n, o;
        dO3.divmod((uint32_t)j, n, o);
        return Col{(long)n * Cy * O3 + o};
    }
    __device__ void store(const Col& c, int i, int j, float v) const {
        if (bias) v += bias[i];
        y[c.off + (long)i * O3] = sg_apply_act(v, act, slope);
    }
};

// ---- dgrad -----------------------------------------------------------------------------------
struct DgradWeightLoader {  // A(i=ci, k=co*8+t) = Wt[parity][k][ci]  (ci contiguous)
    const float* wt;
    int Cin;
    long pstride;
    MatColMajor m;
    template <int BR>
    __device__ void init(int tid, int row0, int nrows) {
        m.p = wt + (long)blockIdx.z * pstride;
        m.ld = Cin;
        m.template init<BR>(tid, row0, nrows);
    }
    template <int BR>
    __device__ void load(int k0, int kend, float (&r)[BR * kBK / 256]) {
        m.template load<BR>(k0, kend, r);
    }
    template <int BR, int LO, int HI>
    __device__ void store(float (*S)[BR + 1], const float (&r)[BR * kBK / 256]) const {
        m.template store<BR, LO, HI>(S, r);
    }
};
// B(k=co*8+t, j=(n,qd,qh,qw)) = dy[n,co,qd+pd-td,qh+ph-th,qw+pw-tw], lanes along positions.  A 16-aligned k-tile
// covers two output channels x 8 taps: per-thread tap offsets/validity are precomputed, the tile part is co0*O^3.
struct DgradPatchLoader {
    const float* dy;
    ConvGeom g;
    int base, tid_, kq;
    int off[8];
    unsigned okbits, pend;
    template <int BR>
    __device__ void init(int tid, int j0, int N) {
        constexpr int E = BR * kBK / 256, STEP = 256 / BR;
        tid_ = tid;
        const int j = j0 + tid % BR;
        kq = tid / BR;
        int n, qd, qh, qw;
        decode_pos(g, (uint32_t)(j < N ? j : 0), n, qd, qh, qw);
        const int p = blockIdx.z;
        const int d1 = qd + ((p >> 2) & 1), h1 = qh + ((p >> 1) & 1), w1 = qw + (p & 1);
        const int O3 = g.OD * g.OH * g.OW;
        base = n * g.Cy * O3 + (d1 * g.OH + h1) * g.OW + w1;
        okbits = 0;
#pragma unroll
        for (int it = 0; it < E; ++it) {
            const int kk = kq + it * STEP, t = kk & 7, dco = kk >> 3;
            const int td = (t >> 2) & 1, th = (t >> 1) & 1, tw = t & 1;
            off[it] = dco * O3 - (td * g.OH + th) * g.OW - tw;
            const bool ok = j < N && (unsigned)(d1 - td) < (unsigned)g.OD && (unsigned)(h1 - th) < (unsigned)g.OH &&
                            (unsigned)(w1 - tw) < (unsigned)g.OW;
            if (ok) okbits |= 1u << it;
        }
    }
    template <int BR>
    __device__ void load(int k0, int kend, float (&r)[BR * kBK / 256]) {
        constexpr int STEP = 256 / BR;
        const int toff = base + (k0 >> 3) * g.OD * g.OH * g.OW;
        pend = 0;
#pragma unroll
        for (int it = 0; it < BR * kBK / 256; ++it) {
            const bool e = ((okbits >> it) & 1u) && (k0 + kq + it * STEP) < kend;
            r[it] = dy[e ? toff + off[it] : 0];
            pend |= (e ? 1u : 0u) << it;
        }
    }
    template <int BR, int LO, int HI>
    __device__ void store(float (*S)[BR + 1], const float (&r)[BR * kBK / 256]) const {
        StageRowFast<BR>::template store<LO, HI>(S, r, tid_, pend);
    }
};
struct DgradEpi {  // dx[n][ci][2qd+pd][2qh+ph][2qw+pw] = act(v + bias[ci])
    float* dx;
    const float* bias;
    ConvGeom g;
    int act;
    float slope;
    struct Col {
        long off;
    };
    __device__ Col col(int j) const {
        int n, qd, qh, qw;
        decode_pos(g, (uint32_t)j, n, qd, qh, qw);
        const int p = blockIdx.z;
        const int d = 2 * qd + ((p >> 2) & 1), h = 2 * qh + ((p >> 1) & 1), w = 2 * qw + (p & 1);
        return Col{(long)n * g.Cx * g.ID * g.IH * g.IW + ((long)d * g.IH + h) * g.IW + w};
    }
    __device__ void store(const Col& c, int i, int j, float v) const {
        if (bias) v += bias[i];
        dx[c.off + (long)i * g.ID * g.IH * g.IW] = sg_apply_act(v, act, slope);
    }
};

// Wt[p][co*8+t][ci] = W[co][ci][kd][kh][kw], kd = 1 - pd + 2*td (same for h, w)
__global__ void __launch_bounds__(256) pack_dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                 int Cout, int Cin_total, int Cin) {
    const long total = 8L * Cout * 8 * Cin;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ci = (int)(e % Cin);
        long r = e / Cin;
        const int t = (int)(r & 7);
        r >>= 3;
        const int co = (int)(r % Cout);
        const int p = (int)(r / Cout);
        const int kd = 1 - ((p >> 2) & 1) + 2 * ((t >> 2) & 1);
        const int kh = 1 - ((p >> 1) & 1) + 2 * ((t >> 1) & 1);
        const int kw = 1 - (p & 1) + 2 * (t & 1);
        wt[e] = w[((long)co * Cin_total + ci) * 64 + kd * 16 + kh * 4 + kw];
    }
}

// dgrad-form with ONE output channel (G's last ConvTranspose 64->1, D's first conv dgrad, the progressive D's
// from_SDF stage): a 1-row GEMM would waste 63/64 of every MFMA, so this is a VALU kernel.  One thread owns the 2x2x2
// output block of position q: all 8 outputs read the same 3x3x3 neighbourhood of dy per channel (27 coalesced
// loads, lanes along qw) and the channel's 64 weights from LDS as 16 broadcast ds_read_b128.
//   output i = 2q + p:   p = 0 -> taps (o = q, k = 1), (o = q-1, k = 3);   p = 1 -> (o = q+1, k = 0), (o = q, k = 2)
__global__ void __launch_bounds__(256) dgrad_out1_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ dx,
                                                        ConvGeom g, int Cout, int Cin_total, int total, int act,
                                                        float slope) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cout][64] (ci = 0 slice)
    for (int e = threadIdx.x; e < Cout * 64; e += 256) wl[e] = w[((long)(e >> 6) * Cin_total) * 64 + (e & 63)];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    int n, qd, qh, qw;
    decode_pos(g, (uint32_t)j, n, qd, qh, qw);
    const int OHW = g.OH * g.OW, O3 = g.OD * OHW;
    // neighbour offsets / validity, a in {-1, 0, +1} -> index a + 1
    int offd[3], offh[3], offw[3];
    bool vd[3], vh[3], vw[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vd[a] = (unsigned)(qd + a - 1) < (unsigned)g.OD;
        vh[a] = (unsigned)(qh + a - 1) < (unsigned)g.OH;
        vw[a] = (unsigned)(qw + a - 1) < (unsigned)g.OW;
        offd[a] = (qd + a - 1) * OHW;
        offh[a] = (qh + a - 1) * g.OW;
        offw[a] = qw + a - 1;
    }
    float acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 8; ++a) (&acc[0][0][0])[a] = 0.f;
    const float* dyn = dy + (long)n * g.Cy * O3;
    for (int co = 0; co < Cout; ++co) {
        const float* dyc = dyn + (long)co * O3;
        float v[3][3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v[a][b][c] = (vd[a] && vh[b] && vw[c]) ? dyc[offd[a] + offh[b] + offw[c]] : 0.f;
        const float4* wc = reinterpret_cast<const float4*>(wl + co * 64);
        // neighbour index for (parity p, slot s): p=0: s0 -> a=1 (k=1), s1 -> a=0 (k=3); p=1: s0 -> a=2 (k=0), s1 -> a=1 (k=2)
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) {
            const int pd = (kd & 1) ? 0 : 1, ad = (kd == 0) ? 2 : (kd == 3 ? 0 : 1);
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int ph = (kh & 1) ? 0 : 1, ah = (kh == 0) ? 2 : (kh == 3 ? 0 : 1);
                const float4 w4 = wc[kd * 4 + kh];
                // kw = 0 -> pw 1, a 2 ; kw = 1 -> pw 0, a 1 ; kw = 2 -> pw 1, a 1 ; kw = 3 -> pw 0, a 0
                acc[pd][ph][1] = fmaf(v[ad][ah][2], w4.x, acc[pd][ph][1]);
                acc[pd][ph][0] = fmaf(v[ad][ah][1], w4.y, acc[pd][ph][0]);
                acc[pd][ph][1] = fmaf(v[ad][ah][1], w4.z, acc[pd][ph][1]);
                acc[pd][ph][0] = fmaf(v[ad][ah][0], w4.w, acc[pd][ph][0]);
            }
        }
    }
    const float b0 = bias ? bias[0] : 0.f;
    float* out = dx + (long)n * g.Cx * g.ID * g.IH * g.IW;
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

// The same operator with LDS staging and register blocking, for grids that tile by 4 x 8 x 16 (every 16^3 / 32^3 layer):
// a workgroup owns a 4x8x16 block of positions q of one sample; per stage it copies the (4+2)x(8+2)x(16+2) box of dy
// of 8 channels into LDS (buffer loads, the hardware returns 0 outside the grid; double buffered), a thread owns TWO
// neighbouring q along w = 2x2x4 outputs x 2: per channel it reads 9 rows of 4 consecutive dy values (18 ds_read_b64)
// and takes the channel's 64 weights from scalar registers (s_load), 128 FMAs — the FMA pipe, not the load path, is
// the limit.  Output rows are written as float4.
constexpr int kO1D = 4, kO1H = 8, kO1W = 16;                       // q block
constexpr int kO1BD = kO1D + 2, kO1BH = kO1H + 2, kO1BW = kO1W + 2;  // box
constexpr int kO1CH = kO1BD * kO1BH * kO1BW;                       // 1080 floats per channel
constexpr int kO1CC = 8;                                           // channels per stage
constexpr int kO1NF = (kO1CH + 255) / 256;                         // copy elements per thread per channel

__global__ void __launch_bounds__(256) dgrad_out1_tile_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ dx,
                                                             ConvGeom g, int Cout, int Cin_total, int ntd, int nth, int ntw,
                                                             int act, float slope) {
    extern __shared__ __attribute__((aligned(16))) float box[];  // [2][kO1CC][kO1CH]
    lds_float* const bl = (lds_float*)box;
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int twi = t % ntw; t /= ntw;
    const int thi = t % nth; t /= nth;
    const int tdi = t % ntd;
    const int n = t / ntd;
    const int qd0 = tdi * kO1D, qh0 = thi * kO1H, qw0 = twi * kO1W;
    const int OHW = g.OH * g.OW, O3 = g.OD * OHW;
    const __amdgpu_buffer_rsrc_t dres = make_rsrc(dy + (long)n * g.Cy * O3);
    const unsigned chan_bytes = (unsigned)O3 * 4u;

    unsigned goff[kO1NF];
    lds_float* sdst[kO1NF];
#pragma unroll
    for (int f = 0; f < kO1NF; ++f) {
        const int e = tid + 256 * f;
        const int bd = e / (kO1BH * kO1BW), rem = e - bd * (kO1BH * kO1BW), bh = rem / kO1BW, bw = rem - bh * kO1BW;
        const int od = qd0 - 1 + bd, oh = qh0 - 1 + bh, ow = qw0 - 1 + bw;
        const bool ok = e < kO1CH && (unsigned)od < (unsigned)g.OD && (unsigned)oh < (unsigned)g.OH && (unsigned)ow < (unsigned)g.OW;
        goff[f] = ok ? (unsigned)((od * g.OH + oh) * g.OW + ow) * 4u : kBufOutside;
        sdst[f] = bl + (e < kO1CH ? e : kO1CH);   // one spare float behind each channel box takes the overshoot
    }
    // this thread's two positions: qw = 2 wq, 2 wq + 1
    const int wq = tid & 7, hq = (tid >> 3) & 7, dq = tid >> 6;
    const lds_float* rbase = bl + dq * (kO1BH * kO1BW) + hq * kO1BW + 2 * wq;

    float acc[2][2][2][2];   // [q][pd][ph][pw]
#pragma unroll
    for (int i = 0; i < 16; ++i) (&acc[0][0][0][0])[i] = 0.f;

    constexpr int kBUF = kO1CC * (kO1CH + 8);   // channel stride kO1CH + 8 keeps 8-byte alignment of the rows
    constexpr int kCS = kO1CH + 8;
    const int nstage = (Cout + kO1CC - 1) / kO1CC;
    float fv[kO1CC][kO1NF];
    auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < kO1CC; ++c) {
            const int co = s * kO1CC + c;
#pragma unroll
            for (int f = 0; f < kO1NF; ++f) fv[c][f] = buf_load(dres, co < Cout ? goff[f] : kBufOutside, (unsigned)co * chan_bytes);
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < kO1CC; ++c)
#pragma unroll
            for (int f = 0; f < kO1NF; ++f) sdst[f][buf * kBUF + c * kCS] = fv[c][f];
    };
    issue(0);
    commit(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nstage;
        if (more) issue(s + 1);
#pragma unroll 2
        for (int c = 0; c < kO1CC; ++c) {
            const int co = s * kO1CC + c;
            if (co >= Cout) break;
            const float* wc = w + (long)co * Cin_total * 64;   // uniform address: scalar loads
            float v[3][3][4];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const lds_float* rp = rbase + cur * kBUF + c * kCS + a * (kO1BH * kO1BW) + b * kO1BW;
                    v[a][b][0] = rp[0];
                    v[a][b][1] = rp[1];
                    v[a][b][2] = rp[2];
                    v[a][b][3] = rp[3];
                }
            // neighbour index for (parity p, tap k): k=0 -> p 1, a 2 ; k=1 -> p 0, a 1 ; k=2 -> p 1, a 1 ; k=3 -> p 0, a 0
#pragma unroll
            for (int kd = 0; kd < 4; ++kd) {
                const int pd = (kd & 1) ? 0 : 1, ad = (kd == 0) ? 2 : (kd == 3 ? 0 : 1);
#pragma unroll
                for (int kh = 0; kh < 4; ++kh) {
                    const int ph = (kh & 1) ? 0 : 1, ah = (kh == 0) ? 2 : (kh == 3 ? 0 : 1);
                    const float w0 = wc[kd * 16 + kh * 4 + 0], w1 = wc[kd * 16 + kh * 4 + 1], w2 = wc[kd * 16 + kh * 4 + 2],
                                w3 = wc[kd * 16 + kh * 4 + 3];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc[q][pd][ph][1] = fmaf(v[ad][ah][q + 2], w0, acc[q][pd][ph][1]);
                        acc[q][pd][ph][0] = fmaf(v[ad][ah][q + 1], w1, acc[q][pd][ph][0]);
                        acc[q][pd][ph][1] = fmaf(v[ad][ah][q + 1], w2, acc[q][pd][ph][1]);
                        acc[q][pd][ph][0] = fmaf(v[ad][ah][q + 0], w3, acc[q][pd][ph][0]);
                    }
                }
            }
        }
        if (more) commit(cur ^ 1);
        __syncthreads();
    }
    const float b0 = bias ? bias[0] : 0.f;
    float* out = dx + (long)n * g.Cx * g.ID * g.IH * g.IW;
    const int qd = qd0 + dq, qh = qh0 + hq, qw = qw0 + 2 * wq;
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float4 o;
            o.x = sg_apply_act(acc[0][pd][ph][0] + b0, act, slope);
            o.y = sg_apply_act(acc[0][pd][ph][1] + b0, act, slope);
            o.z = sg_apply_act(acc[1][pd][ph][0] + b0, act, slope);
            o.w = sg_apply_act(acc[1][pd][ph][1] + b0, act, slope);
            *reinterpret_cast<float4*>(out + ((long)(2 * qd + pd) * g.IH + (2 * qh + ph)) * g.IW + 2 * qw) = o;
        }
}

// ---- wgrad -----------------------------------------------------------------------------------
// A(i=co, k=(n,o)) = dy[n][co][o], lanes along k (positions).  Per k-tile each thread decodes its one position.
struct WgradDyLoader {
    const float* dy;
    int O3, Cy;
    FastDiv dO3;
    int tid_, kk, rowoff0, rbase, nrows_;
    unsigned pend;
    template <int BR>
    __device__ void init(int tid, int row0, int nrows) {
        tid_ = tid;
        kk = tid & 15;
        rbase = row0 + (tid >> 4);
        nrows_ = nrows;
        rowoff0 = rbase * O3;
    }
    template <int BR>
    __device__ void load(int k0, int kend, float (&r)[BR * kBK / 256]) {
        const int k = k0 + kk;
        const bool kok = k < kend;
        uint32_t n, o;
        dO3.divmod((uint32_t)(kok ? k : 0), n, o);
        const int base = (int)n * Cy * O3 + (int)o + rowoff0;
        pend = 0;
#pragma unroll
        for (int it = 0; it < BR * kBK / 256; ++it) {
            const bool e = kok && (rbase + it * 16) < nrows_;
            r[it] = dy[e ? base + it * 16 * O3 : 0];
            pend |= (e ? 1u : 0u) << it;
        }
    }
    template <int BR, int LO, int HI>
    __device__ void store(float (*S)[BR + 1], const float (&r)[BR * kBK / 256]) const {
        StageKFast<BR>::template store<LO, HI>(S, r, tid_, pend);
    }
};
// B(k=pos, j=(ci,tap)), lanes along k (positions).  Rows of one thread: j = j0 + tid/16 + 16*it with j0 % 64 == 0, so
// (kh,kw) = tid/16 is a thread constant, kd = it & 3 and ci = j0/64 + it/4 are compile-time/uniform.
struct WgradPatchLoader {
    const float* x;
    ConvGeom g;
    int tid_, kk, ci0, ncin, hw, off0;
    unsigned pend;
    template <int BR>
    __device__ void init(int tid, int j0, int N) {
        tid_ = tid;
        kk = tid & 15;
        hw = tid >> 4;  // kh*4 + kw
        ci0 = j0 >> 6;
        ncin = N >> 6;
        off0 = (hw >> 2) * g.IW + (hw & 3);
    }
    template <int BR>
    __device__ void load(int k0, int kend, float (&r)[BR * kBK / 256]) {
        const int k = k0 + kk;
        const bool kok = k < kend;
        PatchCtx c;
        c.set(g, (uint32_t)(kok ? k : 0));
        const bool hwok = kok && (((c.mask >> (4 + (hw >> 2))) & (c.mask >> (8 + (hw & 3))) & 1) != 0);
        const int I3 = g.ID * g.IH * g.IW, IHW = g.IH * g.IW;
        const int b = c.base + off0 + ci0 * I3;
        pend = 0;
#pragma unroll
        for (int it = 0; it < BR * kBK / 256; ++it) {
            const bool ok = hwok && ((c.mask >> (it & 3)) & 1) && (ci0 + (it >> 2)) < ncin;
            r[it] = x[ok ? b + (it >> 2) * I3 + (it & 3) * IHW : 0];
            pend |= (ok ? 1u : 0u) << it;
        }
    }
    template <int BR, int LO, int HI>
    __device__ void store(float (*S)[BR + 1], const float (&r)[BR * kBK / 256]) const {
        StageKFast<BR>::template store<LO, HI>(S, r, tid_, pend);
    }
};
struct WgradEpi {  // dW[co][j], row stride ldw (= Cin_total*64)
    float* dw;
    long ldw;
    struct Col {
        int j;
    };
    __device__ Col col(int j) const { return Col{j}; }
    __device__ void store(const Col& c, int i, int j, float v) const { dw[(long)i * ldw + c.j] = v; }
};

}  // namespace sg

using namespace sg;

extern "C" {

// A/B switch for the edge-layer kernels of conv3d_edge.hip (SG_NO_EDGE=1 restores the previous kernels)
static bool edge_enabled(int which = 7) {   // SG_NO_EDGE = bit mask: 1 forward, 2 input gradient, 4 weight gradient
    constexpr int off = SG_NO_EDGE;
    return (off & which) == 0;
}

size_t sg_conv3d_k4s2p1_dgrad_workspace_bytes(int Cout, int Cin) {
    const size_t a = (size_t)8 * Cout * 8 * Cin * sizeof(float), b = halo_dgrad_workspace_bytes(Cin, Cout);
    return a > b ? a : b;
}

size_t sg_conv3d_k4s2p1_dgrad_workspace_bytes_for(int batch, int Cin, int Cout, int OD, int OH, int OW) {
    const size_t a = sg_conv3d_k4s2p1_dgrad_workspace_bytes(Cout, Cin);
    const size_t b = (Cin == 1 && Cout <= 64) ? edge_dgrad_workspace_bytes(batch, OD, OH, OW) : 0;
    return a > b ? a : b;
}

size_t sg_conv3d_k4s2p1_wgrad_workspace_bytes(int batch, int Cin, int Cout, int OD, int OH, int OW) {
    // gather kernel: up to 16 split-K partials of the [Cout, Cin*64] weight gradient; LDS-halo kernel: packed dy + partials
    const size_t a = (size_t)16 * Cout * Cin * 64 * sizeof(float);
    const size_t b = (Cin >= 2 && Cout >= 32) ? halo_wgrad_workspace_bytes(batch, Cin, Cout, OD, OH, OW) : 0;   // halo_wgrad_try's own gate
    const size_t c = (Cin == 1 && Cout <= 64) ? edge_wgrad_workspace_bytes(batch, OD, OH, OW) : 0;
    const size_t ab = a > b ? a : b;
    return ab > c ? ab : c;
}

size_t sg_conv3d_k4s2p1_fwd_workspace_bytes(int batch, int Cin, int Cout, int OD, int OH, int OW) {
    // (a) the LDS-halo kernel's packed weight image, (b) split-K partials of the gather kernel, used only when
    // batch*O^3 x Cout gives fewer than 512 tiles of 64x64 (<= 8 partials)
    const size_t per = (size_t)batch * Cout * OD * OH * OW;
    const size_t tiles = ((size_t)batch * OD * OH * OW + 63) / 64 * ((Cout + 63) / 64);
    const size_t splitk = tiles >= 512 ? 0 : per * 8 * sizeof(float);
    const size_t pack = halo_fwd_workspace_bytes(Cin, Cout);
    const size_t edge = (Cin == 1 && Cout <= 64) ? edge_fwd_workspace_bytes(batch, OD, OH, OW) : 0;
    // (4^3 outputs at small batches: the halo kernel's image AND its channel-split partials — at most 8 — side by side)
    const size_t sp = (OD == 4 && OH == 4 && OW == 4) ? pack + splitk : (splitk > pack ? splitk : pack);
    return sp > edge ? sp : edge;
}

static int check_sizes(const ConvGeom& g, int batch, const char* who) {
    const long npos = (long)batch * g.O3();
    if (npos >= (1L << 31) || (long)batch * g.Cx * g.I3() >= (1L << 31) || (long)batch * g.Cy * g.O3() >= (1L << 31))
        SG_FAIL(SG_ERR_ARG, "%s: tensors of 2^31 elements or more are not supported (32-bit gather offsets)", who);
    return 0;
}

}  // extern "C"

// packed_already: the LDS-halo kernel's weight image of this call is still in `workspace` (sg_conv3d_k4s2p1_fwd_keep)
static int fwd_call(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                    int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* workspace,
                    size_t workspace_bytes, hipStream_t stream, bool packed_already) {
    SG_CHECK_ARG(x && w && y && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_fwd: spatial dims must be even and >= 2");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_fwd")) return SG_ERR_ARG;
    const long npos = (long)batch * g.O3();
    if (Cin == 1 && edge_enabled(1) &&
        edge_fwd_try(x, w, bias, y, batch, Cin, Cin_total, g, Cout, act, slope, workspace, workspace_bytes, stream) == 1) {
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
    {
        const int rc = halo_fwd_try(x, w, bias, y, batch, Cin, Cin_total, g, Cout, act, slope, workspace, workspace_bytes,
                                    stream, 0, 0, packed_already);
        if (rc < 0) return rc;
        if (rc == 1) {
            SG_CHECK_LAUNCH();
            return SG_OK;
        }
        if (rc > 16) {      // the 4^3 halo kernel with its input channels split over rc - 16 workgroups: partial sums behind the image
            const int csplit = rc - 16;
            FwdEpi epi{y, bias, g.O3(), Cout, FastDiv((uint32_t)g.O3()), act, slope};
            const float* partial = reinterpret_cast<const float*>(static_cast<const char*>(workspace) + halo_fwd_workspace_bytes(Cin, Cout));
            const long fb = (long)Cout * ((npos + 1023) >> 10);
            hipLaunchKernelGGL((splitk_finalize_kernel<FwdEpi>), dim3((unsigned)fb), dim3(256), 0, stream, partial, epi, Cout, (int)npos,
                               csplit);
            SG_CHECK_LAUNCH();
            return SG_OK;
        }
    }
    MatRowMajor la;
    la.p = w;
    la.ld = (long)Cin_total * 64;
    FwdPatchLoader lb;
    lb.x = x;
    lb.g = g;
    FwdEpi epi{y, bias, g.O3(), Cout, FastDiv((uint32_t)g.O3()), act, slope};
    // the split-K plan depends on how many partial images fit: use at most what the size query asks for, so that the same call
    // sums in the same order whether it is given a workspace of exactly that size (a kept one) or a larger shared one
    size_t ws_use = workspace ? workspace_bytes : 0;
    const size_t ws_query = sg_conv3d_k4s2p1_fwd_workspace_bytes(batch, Cin, Cout, g.OD, g.OH, g.OW);
    if (ws_use > ws_query) ws_use = ws_query;
    launch_tile_gemm(la, lb, epi, Cout, (int)npos, Cin * 64, (float*)workspace, ws_use, stream);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

extern "C" {

int sg_conv3d_k4s2p1_fwd(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                         int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
    return fwd_call(x, w, bias, y, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, act, slope, workspace, workspace_bytes, stream, false);
}
// The same with the weight image KEPT in a workspace the caller dedicates to this weight (see sg_conv3d_k4s2p1_dgrad_keep):
// weights_unchanged != 0 promises that the image of exactly this call (same weight values, same shapes) is in place — left by
// an earlier call on the workspace or by sg_conv3d_k4s2p1_pack_images.
int sg_conv3d_k4s2p1_fwd_keep(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                              int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* workspace,
                              size_t workspace_bytes, int weights_unchanged, hipStream_t stream) {
    return fwd_call(x, w, bias, y, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, act, slope, workspace, workspace_bytes, stream,
                    weights_unchanged != 0);
}

// Forces one implementation of the forward (testing / tuning): impl 0 = gather kernel, 1 = LDS-halo kernel
// (SG_ERR_ARG if the shape is not eligible).  `debug` is for timing experiments only.
int sg_conv3d_k4s2p1_fwd_impl(const float* x, const float* w, const float* bias, float* y, int batch, int Cin,
                              int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                              void* workspace, size_t workspace_bytes, int impl, int debug, hipStream_t stream) {
    SG_CHECK_ARG(x && w && y && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_fwd_impl: bad spatial dims");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_fwd_impl")) return SG_ERR_ARG;
    if (impl == 1) {
        const int rc = halo_fwd_try(x, w, bias, y, batch, Cin, Cin_total, g, Cout, act, slope, workspace, workspace_bytes,
                                    stream, 1, debug);
        if (rc != 1) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_fwd_impl: shape not eligible for the LDS-halo kernel");
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
    const long npos = (long)batch * g.O3();
    MatRowMajor la;
    la.p = w;
    la.ld = (long)Cin_total * 64;
    FwdPatchLoader lb;
    lb.x = x;
    lb.g = g;
    FwdEpi epi{y, bias, g.O3(), Cout, FastDiv((uint32_t)g.O3()), act, slope};
    launch_tile_gemm(la, lb, epi, Cout, (int)npos, Cin * 64, (float*)workspace, workspace ? workspace_bytes : 0, stream);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"

// `packed_already`: the weight image a kernel of this call would build in the workspace is still there (same weights, same
// shapes, workspace untouched since): the packing launch is skipped.  Which kernel serves a call depends on the shapes only.
static int dgrad_call(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin, int Cin_total, int Cx,
                      int Cout, int ID, int IH, int IW, int act, float slope, void* workspace, size_t workspace_bytes,
                      hipStream_t stream, bool packed_already) {
    SG_CHECK_ARG(dy && w && dx && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_dgrad: spatial dims must be even and >= 2");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_dgrad")) return SG_ERR_ARG;
    const long npos = (long)batch * g.O3();
    if (Cin == 1 && edge_enabled(2) &&
        edge_dgrad_try(dy, w, bias, dx, batch, Cin, Cin_total, g, Cout, act, slope, workspace, workspace_bytes, stream) == 1) {
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
    if (Cin == 1 && g.OD % kO1D == 0 && g.OH % kO1H == 0 && g.OW % kO1W == 0 && (long)g.Cy * g.O3() * 4 < (long)kBufRange) {
        const int ntd = g.OD / kO1D, nth = g.OH / kO1H, ntw = g.OW / kO1W;
        const size_t lds = (size_t)(2 * kO1CC * (kO1CH + 8) + 8) * sizeof(float);
        static SgPerDeviceOnce attr_once;   // > 48 KB of dynamic LDS needs the attribute once per DEVICE
        if (attr_once.begin()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_out1_tile_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_once.end();
        }
        hipLaunchKernelGGL(dgrad_out1_tile_kernel, dim3((unsigned)((long)batch * ntd * nth * ntw)), dim3(256), lds, stream, dy, w,
                           bias, dx, g, Cout, Cin_total, ntd, nth, ntw, act, slope);
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
    if (Cin == 1 && Cout <= 512) {
        hipLaunchKernelGGL(dgrad_out1_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256),
                           (size_t)Cout * 64 * sizeof(float), stream, dy, w, bias, dx, g, Cout, Cin_total, (int)npos, act,
                           slope);
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
    {
        const int rc = halo_dgrad_try(dy, w, bias, dx, batch, Cin, Cin_total, g, Cout, act, slope, workspace,
                                      workspace_bytes, stream, 0, packed_already);
        if (rc < 0) return rc;
        if (rc == 1) {
            SG_CHECK_LAUNCH();
            return SG_OK;
        }
    }
    const size_t need = (size_t)8 * Cout * 8 * Cin * sizeof(float);
    if (!workspace || workspace_bytes < need)
        SG_FAIL(SG_ERR_WORKSPACE, "sg_conv3d_k4s2p1_dgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    float* wt = (float*)workspace;
    if (!packed_already) {
        const long total = 8L * Cout * 8 * Cin;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(pack_dgrad_weights_kernel, dim3(blocks), dim3(256), 0, stream, w, wt, Cout, Cin_total, Cin);
    }
    DgradWeightLoader la;
    la.wt = wt;
    la.Cin = Cin;
    la.pstride = (long)Cout * 8 * Cin;
    DgradPatchLoader lb;
    lb.dy = dy;
    lb.g = g;
    DgradEpi epi{dx, bias, g, act, slope};
    const int M = Cin, N = (int)npos, K = Cout * 8;
    const TilePlan p = plan_tiles(M, N, K, 0, 8);
    dim3 grid(sg_cdiv(N, 64 * p.tn), sg_cdiv(M, 64 * p.tm), 8);
#define SG_DGRAD_LAUNCH(TM_, TN_)                                                                                     \
    hipLaunchKernelGGL((tile_gemm_kernel<TM_, TN_, DgradWeightLoader, DgradPatchLoader, DgradEpi>), grid, dim3(256), 0, \
                       stream, la, lb, epi, M, N, K, 0)
    if (p.tm == 2 && p.tn == 2)
        SG_DGRAD_LAUNCH(2, 2);
    else if (p.tm == 2)
        SG_DGRAD_LAUNCH(2, 1);
    else if (p.tn == 2)
        SG_DGRAD_LAUNCH(1, 2);
    else
        SG_DGRAD_LAUNCH(1, 1);
#undef SG_DGRAD_LAUNCH
    SG_CHECK_LAUNCH();
    return SG_OK;
}

extern "C" {

int sg_conv3d_k4s2p1_dgrad(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                           int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return dgrad_call(dy, w, bias, dx, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, act, slope, workspace, workspace_bytes, stream,
                      false);
}
// The same with the packed weight image KEPT between calls (a ConvTranspose3d forward whose weights did not change since the
// last call: the WGAN generator is evaluated six times per training unit and updated once): `workspace` is a buffer the
// caller dedicates to this weight tensor; weights_unchanged != 0 promises that the previous call on it had the same weight
// values and the same shapes and that nothing else wrote to it.
int sg_conv3d_k4s2p1_dgrad_keep(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                                int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                                void* workspace, size_t workspace_bytes, int weights_unchanged, hipStream_t stream) {
    return dgrad_call(dy, w, bias, dx, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, act, slope, workspace, workspace_bytes, stream,
                      weights_unchanged != 0);
}

// The weight images of up to 8 forthcoming calls in ONE launch.  Call i is described by kinds[i] (0: sg_conv3d_k4s2p1_fwd_keep, 1:
// sg_conv3d_k4s2p1_dgrad_keep) and dims[8 i ..] = {batch, Cin, Cin_total, Cx, Cout, ID, IH, IW} exactly as it will be made, with the
// workspace it will be given.  served[i] = 1: the image is in place, make the call with weights_unchanged = 1;  0: that call is
// not served by a kernel with a kept image (one-channel layers, small shapes on the gather kernels) — make it with
// weights_unchanged = 0.  Planning is done by the same code that serves the calls.
int sg_conv3d_k4s2p1_pack_images(int n, const int* kinds, const float* const* weights, void* const* workspaces,
                                 const size_t* workspace_bytes, const int* dims, int* served, hipStream_t stream) {
    SG_CHECK_ARG(n > 0 && n <= 8 && kinds && weights && workspaces && workspace_bytes && dims && served);
    PackJobs jobs;
    static const float dummy = 0.f;       // operand pointers are only stored in collect mode, never dereferenced
    for (int i = 0; i < n; ++i) {
        const int* d = dims + 8 * i;
        const int batch = d[0], Cin = d[1], Cin_total = d[2], Cx = d[3], Cout = d[4];
        served[i] = 0;
        SG_CHECK_ARG(weights[i] && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0 && (kinds[i] == 0 || kinds[i] == 1));
        ConvGeom g;
        if (make_geom(g, d[5], d[6], d[7], Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_pack_images: bad spatial dims");
        if (check_sizes(g, batch, "sg_conv3d_k4s2p1_pack_images")) return SG_ERR_ARG;
        if (Cin == 1) continue;          // the one-channel kernels read the weights in place
        int rc;
        if (kinds[i] == 0)
            rc = halo_fwd_try(&dummy, weights[i], nullptr, const_cast<float*>(&dummy), batch, Cin, Cin_total, g, Cout, SG_ACT_NONE, 0.f,
                              workspaces[i], workspace_bytes[i], stream, 0, 0, false, &jobs);
        else
            rc = halo_dgrad_try(&dummy, weights[i], nullptr, const_cast<float*>(&dummy), batch, Cin, Cin_total, g, Cout, SG_ACT_NONE, 0.f,
                                workspaces[i], workspace_bytes[i], stream, 0, false, &jobs);
        if (rc < 0) return rc;
        served[i] = rc == 1;
    }
    halo_pack_jobs_launch(jobs, stream);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// What the kept image of a call looks like, as one number: 0 if the call {kind, dims} (as in sg_conv3d_k4s2p1_pack_images) on a
// workspace of `workspace_bytes` is not served by a kernel with a kept image; otherwise a value that is equal for two calls exactly
// when they read the SAME image (same packed form, row-tile count, channel counts and place in the workspace).  The batch size and
// the grid do not enter the image of the LDS-halo kernels: the critic's images serve its 128-sample update passes and the 64-sample
// pass of the generator update alike (a caller keying its images on the full shape rebuilt them at every change of the batch size).
// Pure host code (the planning half of the calls themselves).
long long sg_conv3d_k4s2p1_image_layout(int kind, const int* dims, size_t workspace_bytes) {
    if (!dims || (kind != 0 && kind != 1)) return 0;
    const int batch = dims[0], Cin = dims[1], Cin_total = dims[2], Cx = dims[3], Cout = dims[4];
    if (batch <= 0 || Cin <= 1 || Cin > Cin_total || Cin > Cx || Cout <= 0) return 0;
    ConvGeom g;
    if (make_geom(g, dims[5], dims[6], dims[7], Cx, Cout)) return 0;
    if ((long)batch * g.O3() >= (1L << 31) || (long)batch * g.I3() >= (1L << 31)) return 0;
    PackJobs jobs;
    static const float dummy = 0.f;               // operand pointers are only stored in collect mode, never dereferenced
    char* const base = reinterpret_cast<char*>(uintptr_t(1) << 20);
    const int rc = kind == 0 ? halo_fwd_try(&dummy, &dummy, nullptr, const_cast<float*>(&dummy), batch, Cin, Cin_total, g, Cout, SG_ACT_NONE,
                                            0.f, base, workspace_bytes, nullptr, 0, 0, false, &jobs)
                             : halo_dgrad_try(&dummy, &dummy, nullptr, const_cast<float*>(&dummy), batch, Cin, Cin_total, g, Cout,
                                              SG_ACT_NONE, 0.f, base, workspace_bytes, nullptr, 0, false, &jobs);
    if (rc != 1 || jobs.n != 1) return 0;
    const PackJob& j = jobs.job[0];
    unsigned long long h = 1469598103934665603ull;                       // FNV-1a over the fields that define the image
    const long long f[7] = {j.kind + 1, j.Cout, j.Cin_total, j.Cin, j.nt, (long long)(reinterpret_cast<char*>(j.wp) - base), 0x5347};
    for (long long v : f)
        for (int b = 0; b < 8; ++b) {
            h ^= (unsigned long long)(v >> (8 * b)) & 0xffull;
            h *= 1099511628211ull;
        }
    h &= 0x7fffffffffffffffull;
    return h ? (long long)h : 1;
}

// testing / tuning: impl 1 forces the LDS-halo dgrad kernel (SG_ERR_ARG if the shape is not eligible)
int sg_conv3d_k4s2p1_dgrad_impl(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                                int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                                void* workspace, size_t workspace_bytes, int impl, hipStream_t stream) {
    SG_CHECK_ARG(dy && w && dx && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0 && (impl & 1) &&
                 (impl == 1 || impl == 3 || impl == 9 || impl == 17 || impl == 33));
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_dgrad_impl: bad spatial dims");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_dgrad_impl")) return SG_ERR_ARG;
    const int rc = halo_dgrad_try(dy, w, bias, dx, batch, Cin, Cin_total, g, Cout, act, slope, workspace, workspace_bytes,
                                  stream, impl);   // 1: forced, 3: one parity per workgroup, 1 + 4 ppw: ppw parities
    if (rc != 1) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_dgrad_impl: shape not eligible for the LDS-halo kernel");
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// testing / tuning: impl 1 forces the LDS-halo wgrad kernel (SG_ERR_ARG if the shape is not eligible)
int sg_conv3d_k4s2p1_wgrad_impl(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                                int Cout, int ID, int IH, int IW, void* workspace, size_t workspace_bytes, int impl,
                                hipStream_t stream) {
    SG_CHECK_ARG(dy && x && dw && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0 && impl == 1);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad_impl: bad spatial dims");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_wgrad_impl")) return SG_ERR_ARG;
    const int rc = halo_wgrad_try(dy, x, dw, batch, Cin, Cin_total, g, Cout, workspace, workspace_bytes, stream, 1);
    if (rc != 1) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad_impl: shape not eligible for the LDS-halo kernel");
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// Weight + bias gradient of act(conv(x) + b) taken straight from the gradient w.r.t. the activated output: dz = dy * act'(y) is
// formed inside the weight-gradient kernel (one-channel layers, LeakyReLU / ReLU), so the activation backward is not a pass of
// its own.  sg_conv3d_k4s2p1_wgrad_act_eligible says whether a shape is served (host code, no GPU needed).
// The incoming gradient's A-fragment image of the LDS-halo weight-gradient kernel, written by the kernel that produces the
// gradient instead of by a packing pass of its own (conv3d_halo.hip).  sg_conv3d_k4s2p1_wgrad_dy_image: 1 if the call (batch, Cin,
// Cout, O^3 grid of dy, the workspace it will be given) is served that way — then the image occupies the START of that workspace,
// [mt_total][nslice][8][64] float4 — else 0.  sg_act_bwd_rowsum_pack8 = sg_act_bwd_rowsum for [N][C][8^3] tensors that also writes the
// image; sg_head_dot_bwd takes the image pointer for 4^3 grids.  sg_conv3d_k4s2p1_wgrad_prepacked = sg_conv3d_k4s2p1_wgrad that
// trusts the image in its workspace.
int sg_conv3d_k4s2p1_wgrad_dy_image(int batch, int Cin, int Cout, int OD, int OH, int OW, size_t workspace_bytes, int* mt_total,
                                    long* nslice) {
    if (batch <= 0 || Cin <= 0 || Cout <= 0 || !mt_total || !nslice) return 0;
    ConvGeom g;
    if (make_geom(g, 2 * OD, 2 * OH, 2 * OW, Cin, Cout)) return 0;
    if (Cin == 1 && Cout <= 64) return 0;      // the one-channel kernels
    return halo_wgrad_dy_image_plan(batch, Cin, Cout, g, workspace_bytes, mt_total, nslice);
}
int sg_act_bwd_rowsum_pack8(const float* y, const float* dy, float* dz, float* rowsum, void* dz_image, long N, int C, long nslice,
                            int act, float slope, hipStream_t stream) {
    SG_CHECK_ARG(y && dy && dz && rowsum && dz_image && N > 0 && C > 0 && C % 128 == 0 && nslice == N * 8);
    SG_CHECK_ARG(act == SG_ACT_LEAKY || act == SG_ACT_RELU);
    SG_CHECK_ARG((((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dz | (uintptr_t)dz_image) & 15) == 0);
    halo_act_bwd_pack8_launch(y, dy, dz, rowsum, dz_image, N * C, C, nslice, act, slope, stream);
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_conv3d_k4s2p1_wgrad_prepacked(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                                     int Cout, int ID, int IH, int IW, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(dy && x && dw && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad_prepacked: bad spatial dims");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_wgrad_prepacked")) return SG_ERR_ARG;
    int mt;
    long ns;
    if (!halo_wgrad_dy_image_plan(batch, Cin, Cout, g, workspace_bytes, &mt, &ns))
        SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad_prepacked: this call is not served with a producer-written image");
    if (halo_wgrad_try(dy, x, dw, batch, Cin, Cin_total, g, Cout, workspace, workspace_bytes, stream, 0, true) != 1)
        SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad_prepacked: not served");
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_conv3d_k4s2p1_wgrad_act_eligible(int batch, int Cin, int Cout, int OD, int OH, int OW, int act) {
    // the same refusal conditions as edge_wgrad_try: 32-bit buffer ranges of the one-channel grid (read in place; the
    // resource starts (IH + 1) rows + 1 element before it) and of dy / y
    const size_t x_bytes = ((size_t)batch * 8 * OD * OH * OW + (size_t)(2 * OH + 1) * 2 * OW + 4) * 4;
    const size_t dy_bytes = (size_t)batch * Cout * OD * OH * OW * 4;
    return Cin == 1 && Cout <= 64 && OW % 16 == 0 && (long)batch * OD * OH * OW >= 65536 && (act == SG_ACT_LEAKY || act == SG_ACT_RELU) &&
           x_bytes < (size_t)kBufRange && dy_bytes < (size_t)kBufRange && edge_enabled(4);
}
int sg_conv3d_k4s2p1_wgrad_act(const float* dy, const float* y, const float* x, float* dw, float* db, int batch, int Cin,
                               int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(dy && y && x && dw && db && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad_act: spatial dims must be even and >= 2");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_wgrad_act")) return SG_ERR_ARG;
    if (!sg_conv3d_k4s2p1_wgrad_act_eligible(batch, Cin, Cout, g.OD, g.OH, g.OW, act) ||
        edge_wgrad_try(dy, x, dw, batch, Cin, Cin_total, g, Cout, workspace, workspace ? workspace_bytes : 0, stream, 0, y, act, slope,
                       db) != 1)
        SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad_act: shape not served (see sg_conv3d_k4s2p1_wgrad_act_eligible) or workspace too small");
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_conv3d_k4s2p1_wgrad(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                           int Cout, int ID, int IH, int IW, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
    SG_CHECK_ARG(dy && x && dw && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad: spatial dims must be even and >= 2");
    if (check_sizes(g, batch, "sg_conv3d_k4s2p1_wgrad")) return SG_ERR_ARG;
    const long npos = (long)batch * g.O3();
    if (Cin == 1 && edge_enabled(4) &&
        edge_wgrad_try(dy, x, dw, batch, Cin, Cin_total, g, Cout, workspace, workspace ? workspace_bytes : 0, stream) == 1) {
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
    {
        const int rc = halo_wgrad_try(dy, x, dw, batch, Cin, Cin_total, g, Cout, workspace, workspace ? workspace_bytes : 0,
                                      stream, 0);
        if (rc < 0) return rc;
        if (rc == 1) {
            SG_CHECK_LAUNCH();
            return SG_OK;
        }
    }
    WgradDyLoader la;
    la.dy = dy;
    la.O3 = (int)g.O3();
    la.Cy = Cout;
    la.dO3 = FastDiv((uint32_t)g.O3());
    WgradPatchLoader lb;
    lb.x = x;
    lb.g = g;
    WgradEpi epi{dw, (long)Cin_total * 64};
    launch_tile_gemm(la, lb, epi, Cout, Cin * 64, (int)npos, (float*)workspace, workspace ? workspace_bytes : 0, stream);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

// ConvTranspose3d(k4,s2,p1): forward = conv dgrad-form, input gradient = conv fwd-form, weight gradient =
// conv wgrad-form with (x, dy) swapped.  Channel naming follows nn.ConvTranspose3d: weight [Cin_T, Cout_T, 4,4,4],
// x [batch, Cin_T, I^3] -> y [batch, Cout_T, (2I)^3].
int sg_convT3d_k4s2p1_fwd(const float* x, const float* w, const float* bias, float* y, int batch, int Cin_T, int Cout_T,
                          int ID, int IH, int IW, int act, float slope, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
    return sg_conv3d_k4s2p1_dgrad(x, w, bias, y, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW, act,
                                  slope, workspace, workspace_bytes, stream);
}
// ConvTranspose3d(C -> 1, k4 s2 p1) with the INPUT taken through act_in(x * in_scale[c] + in_shift[c]) on the way into the kernel
// (a BatchNorm3d + LeakyReLU between the producing layer and this one, model/gan.py:18-21, folded into the loads of
// convT_c1_stream_kernel): x [batch, C, I^3] -> y [batch, 1, (2I)^3].  Served for C <= 64 and planes of at most 256 positions.
int sg_convT3d_k4s2p1_to1_pre_eligible(int batch, int C, int ID, int IH, int IW) {
    return batch > 0 && C > 0 && C <= 64 && ID > 0 && IH > 0 && IW > 0 && IH * IW <= 256 &&
           (size_t)C * ID * IH * IW * 4 < (size_t)kBufRange && (long)batch * 8 * ID * IH * IW < (1L << 31);
}
static int convT_to1_pre(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                         const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW,
                         int act, float slope, int samples_per_group, long y_group_stride, int form, hipStream_t stream) {
    SG_CHECK_ARG(x && w && y && in_scale && in_shift && samples_per_group > 0 && batch % samples_per_group == 0 && form >= 0 && form <= 8);
    SG_CHECK_ARG(y_group_stride >= (long)samples_per_group * 8 * ID * IH * IW);
    if (!sg_convT3d_k4s2p1_to1_pre_eligible(batch, C, ID, IH, IW))
        SG_FAIL(SG_ERR_ARG, "sg_convT3d_k4s2p1_to1_pre: shape not served (C <= 64, IH * IW <= 256)");
    SG_CHECK_ARG(in_act == SG_ACT_NONE || in_act == SG_ACT_RELU || (in_act == SG_ACT_LEAKY && in_slope >= 0.f && in_slope <= 1.f));
    ConvGeom g;
    if (make_geom(g, 2 * ID, 2 * IH, 2 * IW, 1, C)) SG_FAIL(SG_ERR_ARG, "sg_convT3d_k4s2p1_to1_pre: bad spatial dims");
    if (edge_dgrad_stream_try(x, w, bias, y, batch, 1, 1, g, C, act, slope, stream, in_scale, in_shift, in_act, in_slope,
                              samples_per_group, y_group_stride, form) != 1)
        SG_FAIL(SG_ERR_ARG, "sg_convT3d_k4s2p1_to1_pre: not served");
    SG_CHECK_LAUNCH();
    return SG_OK;
}
int sg_convT3d_k4s2p1_to1_pre_grouped(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                                      const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW,
                                      int act, float slope, int samples_per_group, long y_group_stride, hipStream_t stream) {
    return convT_to1_pre(x, w, bias, y, in_scale, in_shift, in_act, in_slope, batch, C, ID, IH, IW, act, slope, samples_per_group,
                         y_group_stride, 0, stream);
}
// the same with the kernel form chosen by the caller (tests / tuning): 0 the dispatch rule, 1 / 2 one h parity per workgroup with
// one / two plane walks, 3 / 4 both h parities per workgroup with one / two plane walks
int sg_convT3d_k4s2p1_to1_pre_impl(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                                   const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW,
                                   int act, float slope, int form, hipStream_t stream) {
    return convT_to1_pre(x, w, bias, y, in_scale, in_shift, in_act, in_slope, batch, C, ID, IH, IW, act, slope, batch,
                         (long)batch * 8 * ID * IH * IW, form, stream);
}
int sg_convT3d_k4s2p1_to1_pre(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                              const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW, int act,
                              float slope, hipStream_t stream) {
    return sg_convT3d_k4s2p1_to1_pre_grouped(x, w, bias, y, in_scale, in_shift, in_act, in_slope, batch, C, ID, IH, IW, act, slope,
                                             batch, (long)batch * 8 * ID * IH * IW, stream);
}
int sg_convT3d_k4s2p1_dgrad(const float* dy, const float* w, float* dx, int batch, int Cin_T, int Cout_T, int ID, int IH,
                            int IW, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return sg_conv3d_k4s2p1_fwd(dy, w, nullptr, dx, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW,
                                SG_ACT_NONE, 0.f, workspace, workspace_bytes, stream);
}
int sg_convT3d_k4s2p1_wgrad(const float* dy, const float* x, float* dw, int batch, int Cin_T, int Cout_T, int ID, int IH,
                            int IW, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return sg_conv3d_k4s2p1_wgrad(x, dy, dw, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW, workspace,
                                  workspace_bytes, stream);
}

}  // extern "C"
