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
#include "../../include/shapegan_hip.h"

namespace sg {

// n / d for n < 2^31 with precomputed magic (round-up method)
struct FastDiv {
    uint32_t m, s, d;
    FastDiv() : m(0), s(0), d(1) {}
    explicit FastDiv(uint32_t dd) : d(dd) {
        s = 0;
        while ((1u << s) < dd) ++s;
        m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << s) - dd)) / dd + 1);
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return (__umulhi(n, m) + n) >> s; }
    __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
        q = div(n);
        r = n - q * d;
    }
};

struct ConvGeom {
    int ID, IH, IW;   // spatial size of the stride-1 side (x of the conv)
    int OD, OH, OW;   // spatial size of the stride-2 side (y of the conv) = I/2
    int Cx, Cy;       // channels physically present in x / y tensors (batch strides)
    FastDiv dOW, dOH, dOD;
    long I3() const { return (long)ID * IH * IW; }
    long O3() const { return (long)OD * OH * OW; }
};

// decode a flat (n,od,oh,ow) position of the O grid
__device__ __forceinline__ void decode_pos(const ConvGeom& g, uint32_t j, int& n, int& od, int& oh, int& ow) {
    uint32_t t1, t2, t3, a, b, c;
    g.dOW.divmod(j, t1, a);
    g.dOH.divmod(t1, t2, b);
    g.dOD.divmod(t2, t3, c);
    ow = (int)a;
    oh = (int)b;
    od = (int)c;
    n = (int)t3;
}

// patch of x around output position pos=(n,od,oh,ow): element (ci,kd,kh,kw) = x[n,ci,2od-1+kd,2oh-1+kh,2ow-1+kw]
struct PatchCtx {
    long base;
    int mask;
    __device__ __forceinline__ void set(const ConvGeom& g, uint32_t pos) {
        int n, od, oh, ow;
        decode_pos(g, pos, n, od, oh, ow);
        const int id0 = 2 * od - 1, ih0 = 2 * oh - 1, iw0 = 2 * ow - 1;
        base = (long)n * g.Cx * g.ID * g.IH * g.IW + ((long)id0 * g.IH + ih0) * g.IW + iw0;
        int m = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((unsigned)(id0 + t) < (unsigned)g.ID) m |= 1 << t;
            if ((unsigned)(ih0 + t) < (unsigned)g.IH) m |= 1 << (4 + t);
            if ((unsigned)(iw0 + t) < (unsigned)g.IW) m |= 1 << (8 + t);
        }
        mask = m;
    }
    // c = ci*64 + kd*16 + kh*4 + kw
    __device__ __forceinline__ float get(const ConvGeom& g, const float* x, int c) const {
        const int ci = c >> 6, kd = (c >> 4) & 3, kh = (c >> 2) & 3, kw = c & 3;
        const bool ok = ((mask >> kd) & (mask >> (4 + kh)) & (mask >> (8 + kw)) & 1) != 0;
        return ok ? x[base + ((long)ci * g.ID + kd) * g.IH * g.IW + kh * g.IW + kw] : 0.f;
    }
};

// ---- fwd -------------------------------------------------------------------------------------
struct FwdPatchLoader {  // B(k=(ci,tap), j=pos), lanes along positions
    static constexpr bool K_FAST = false;
    const float* x;
    ConvGeom g;
    PatchCtx c;
    __device__ void fix(int j) { c.set(g, (uint32_t)j); }
    __device__ float get(int k) const { return c.get(g, x, k); }
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
struct DgradWeightLoader {  // A(i=ci, k=co*8+t) = Wt[parity][k][ci]
    static constexpr bool K_FAST = false;
    const float* wt;
    int Cin;
    long pstride;
    int rr;
    __device__ void fix(int row) { rr = row; }
    __device__ float get(int k) const { return wt[(long)blockIdx.z * pstride + (long)k * Cin + rr]; }
};
struct DgradPatchLoader {  // B(k=co*8+t, j=(n,qd,qh,qw)) = dy[n,co,qd+pd-td,qh+ph-th,qw+pw-tw]
    static constexpr bool K_FAST = false;
    const float* dy;
    ConvGeom g;
    long base;
    int mask;
    __device__ void fix(int j) {
        int n, qd, qh, qw;
        decode_pos(g, (uint32_t)j, n, qd, qh, qw);
        const int p = blockIdx.z;
        const int d1 = qd + ((p >> 2) & 1), h1 = qh + ((p >> 1) & 1), w1 = qw + (p & 1);
        base = (long)n * g.Cy * g.OD * g.OH * g.OW + ((long)d1 * g.OH + h1) * g.OW + w1;
        int m = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if ((unsigned)(d1 - t) < (unsigned)g.OD) m |= 1 << t;
            if ((unsigned)(h1 - t) < (unsigned)g.OH) m |= 1 << (2 + t);
            if ((unsigned)(w1 - t) < (unsigned)g.OW) m |= 1 << (4 + t);
        }
        mask = m;
    }
    __device__ float get(int k) const {
        const int co = k >> 3, td = (k >> 2) & 1, th = (k >> 1) & 1, tw = k & 1;
        const bool ok = ((mask >> td) & (mask >> (2 + th)) & (mask >> (4 + tw)) & 1) != 0;
        return ok ? dy[base + ((long)co * g.OD - td) * g.OH * g.OW - th * g.OW - tw] : 0.f;
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

// dgrad-form with ONE output channel (G's last ConvTranspose 64->1, D's first conv dgrad): a 1-row
// GEMM would waste 63/64 of every MFMA, so this is a plain VALU gather: one thread per output voxel,
// weights [Cout][64] staged in LDS, dy re-reads served by L1/L2 (each dy element feeds 8 outputs).
__global__ void __launch_bounds__(256) dgrad_out1_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ dx,
                                                        ConvGeom g, int Cout, int Cin_total, long total, int act,
                                                        float slope) {
    extern __shared__ float wl[];  // [Cout][64] (ci = 0 slice)
    for (int e = threadIdx.x; e < Cout * 64; e += 256) wl[e] = w[((long)(e >> 6) * Cin_total) * 64 + (e & 63)];
    __syncthreads();
    const long O3 = (long)g.OD * g.OH * g.OW;
    const int IHW = g.IH * g.IW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int iw = (int)(e % g.IW);
        long r = e / g.IW;
        const int ih = (int)(r % g.IH);
        r /= g.IH;
        const int id = (int)(r % g.ID);
        const int n = (int)(r / g.ID);
        // x index i = 2*o + k - 1  ->  o = (i + 1 - k) / 2 for the two k of matching parity
        int od[2], kd[2], oh[2], kh[2], ow[2], kw[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kd[t] = ((id + 1) & 1) + 2 * t;
            od[t] = (id + 1 - kd[t]) >> 1;
            kh[t] = ((ih + 1) & 1) + 2 * t;
            oh[t] = (ih + 1 - kh[t]) >> 1;
            kw[t] = ((iw + 1) & 1) + 2 * t;
            ow[t] = (iw + 1 - kw[t]) >> 1;
        }
        float acc = 0.f;
        const float* dyn = dy + (long)n * g.Cy * O3;
        for (int co = 0; co < Cout; ++co) {
            const float* dyc = dyn + (long)co * O3;
            const float* wc = wl + co * 64;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if ((unsigned)od[a] >= (unsigned)g.OD) continue;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if ((unsigned)oh[b] >= (unsigned)g.OH) continue;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if ((unsigned)ow[c] >= (unsigned)g.OW) continue;
                        acc = fmaf(dyc[((long)od[a] * g.OH + oh[b]) * g.OW + ow[c]], wc[kd[a] * 16 + kh[b] * 4 + kw[c]],
                                   acc);
                    }
                }
            }
        }
        if (bias) acc += bias[0];
        dx[(long)n * g.Cx * g.ID * IHW + ((long)id * g.IH + ih) * g.IW + iw] = sg_apply_act(acc, act, slope);
    }
}

// ---- wgrad -----------------------------------------------------------------------------------
struct WgradDyLoader {  // A(i=co, k=(n,o)) = dy[n][co][o], lanes along k
    static constexpr bool K_FAST = true;
    const float* dy;
    long O3;
    int Cy;
    FastDiv dO3;
    long base;
    __device__ void fix(int k) {
        uint32_t n, o;
        dO3.divmod((uint32_t)k, n, o);
        base = (long)n * Cy * O3 + o;
    }
    __device__ float get(int row) const { return dy[base + (long)row * O3]; }
};
struct WgradPatchLoader {  // B(k=pos, j=(ci,tap)), lanes along positions
    static constexpr bool K_FAST = true;
    const float* x;
    ConvGeom g;
    PatchCtx c;
    __device__ void fix(int k) { c.set(g, (uint32_t)k); }
    __device__ float get(int j) const { return c.get(g, x, j); }
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

static int make_geom(ConvGeom& g, int ID, int IH, int IW, int Cx, int Cy) {
    if (ID < 2 || IH < 2 || IW < 2 || (ID & 1) || (IH & 1) || (IW & 1)) return -1;
    g.ID = ID;
    g.IH = IH;
    g.IW = IW;
    g.OD = ID / 2;
    g.OH = IH / 2;
    g.OW = IW / 2;
    g.Cx = Cx;
    g.Cy = Cy;
    g.dOW = FastDiv(g.OW);
    g.dOH = FastDiv(g.OH);
    g.dOD = FastDiv(g.OD);
    return 0;
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_conv3d_k4s2p1_dgrad_workspace_bytes(int Cout, int Cin) { return (size_t)8 * Cout * 8 * Cin * sizeof(float); }

size_t sg_conv3d_k4s2p1_wgrad_workspace_bytes(int Cout, int Cin) {
    // up to 64 split-K partials of the [Cout, Cin*64] weight gradient
    return (size_t)64 * Cout * Cin * 64 * sizeof(float);
}

int sg_conv3d_k4s2p1_fwd(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                         int Cx, int Cout, int ID, int IH, int IW, int act, float slope, hipStream_t stream) {
    SG_CHECK_ARG(x && w && y && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_fwd: spatial dims must be even and >= 2");
    const long npos = (long)batch * g.O3();
    SG_CHECK_ARG(npos < (1L << 31) && (long)batch * Cx * g.I3() < (1L << 40));
    MatRowMajor la{w, (long)Cin_total * 64, 0};
    FwdPatchLoader lb{x, g, {}};
    FwdEpi epi{y, bias, g.O3(), Cout, FastDiv((uint32_t)g.O3()), act, slope};
    launch_tile_gemm(la, lb, epi, Cout, (int)npos, Cin * 64, nullptr, 0, stream);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_conv3d_k4s2p1_dgrad(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                           int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
    SG_CHECK_ARG(dy && w && dx && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_dgrad: spatial dims must be even and >= 2");
    const long npos = (long)batch * g.O3();
    SG_CHECK_ARG(npos < (1L << 31));
    if (Cin == 1 && Cout <= 256) {
        const long total = (long)batch * g.I3();
        int blocks = (int)((total + 255) / 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(dgrad_out1_kernel, dim3(blocks), dim3(256), (size_t)Cout * 64 * sizeof(float), stream, dy, w,
                           bias, dx, g, Cout, Cin_total, total, act, slope);
        SG_CHECK_LAUNCH();
        return SG_OK;
    }
    const size_t need = sg_conv3d_k4s2p1_dgrad_workspace_bytes(Cout, Cin);
    if (!workspace || workspace_bytes < need)
        SG_FAIL(SG_ERR_WORKSPACE, "sg_conv3d_k4s2p1_dgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    float* wt = (float*)workspace;
    {
        const long total = 8L * Cout * 8 * Cin;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(pack_dgrad_weights_kernel, dim3(blocks), dim3(256), 0, stream, w, wt, Cout, Cin_total, Cin);
    }
    DgradWeightLoader la{wt, Cin, (long)Cout * 8 * Cin, 0};
    DgradPatchLoader lb{dy, g, 0, 0};
    DgradEpi epi{dx, bias, g, act, slope};
    const int M = Cin, N = (int)npos, K = Cout * 8;
    const int tm = M > 64 ? 2 : 1;
    dim3 grid(sg_cdiv(N, 128), sg_cdiv(M, 64 * tm), 8);
    if (tm == 2)
        hipLaunchKernelGGL((tile_gemm_kernel<2, 2, DgradWeightLoader, DgradPatchLoader, DgradEpi>), grid, dim3(256), 0,
                           stream, la, lb, epi, M, N, K, 0);
    else
        hipLaunchKernelGGL((tile_gemm_kernel<1, 2, DgradWeightLoader, DgradPatchLoader, DgradEpi>), grid, dim3(256), 0,
                           stream, la, lb, epi, M, N, K, 0);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

int sg_conv3d_k4s2p1_wgrad(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                           int Cout, int ID, int IH, int IW, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
    SG_CHECK_ARG(dy && x && dw && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0);
    ConvGeom g;
    if (make_geom(g, ID, IH, IW, Cx, Cout)) SG_FAIL(SG_ERR_ARG, "sg_conv3d_k4s2p1_wgrad: spatial dims must be even and >= 2");
    const long npos = (long)batch * g.O3();
    SG_CHECK_ARG(npos < (1L << 31));
    WgradDyLoader la{dy, g.O3(), Cout, FastDiv((uint32_t)g.O3()), 0};
    WgradPatchLoader lb{x, g, {}};
    WgradEpi epi{dw, (long)Cin_total * 64};
    launch_tile_gemm(la, lb, epi, Cout, Cin * 64, (int)npos, (float*)workspace, workspace_bytes, stream);
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
int sg_convT3d_k4s2p1_dgrad(const float* dy, const float* w, float* dx, int batch, int Cin_T, int Cout_T, int ID, int IH,
                            int IW, hipStream_t stream) {
    return sg_conv3d_k4s2p1_fwd(dy, w, nullptr, dx, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW,
                                SG_ACT_NONE, 0.f, stream);
}
int sg_convT3d_k4s2p1_wgrad(const float* dy, const float* x, float* dw, int batch, int Cin_T, int Cout_T, int ID, int IH,
                            int IW, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return sg_conv3d_k4s2p1_wgrad(x, dy, dw, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW, workspace,
                                  workspace_bytes, stream);
}

}  // extern "C"
