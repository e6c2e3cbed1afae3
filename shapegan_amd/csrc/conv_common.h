// shapegan_amd/csrc/conv_common.h — geometry helpers shared by the k4/s2/p1 convolution kernels.
#pragma once
#include "common.h"

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

static inline int make_geom(ConvGeom& g, int ID, int IH, int IW, int Cx, int Cy) {
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


// Packing jobs of the LDS-halo kernels' weight images, collected by halo_fwd_try / halo_dgrad_try in collect mode and launched
// together (conv3d_halo.hip: pack_images_kernel, sg_conv3d_k4s2p1_pack_images)
struct PackJob {
    const float* w;
    float4* wp;
    int kind, Cout, Cin_total, Cin, nt;     // kind 0: forward image (nt = 32-row tiles), 1: input-gradient image (nt = MT)
};
struct PackJobs {
    PackJob job[8];
    int n = 0;
    int add(int kind, const float* w, float4* wp, int Cout, int Cin_total, int Cin, int nt) {
        if (n >= 8) return 0;
        job[n++] = PackJob{w, wp, kind, Cout, Cin_total, Cin, nt};
        return 1;
    }
};
int halo_pack_jobs_launch(const PackJobs& jobs, hipStream_t stream);

// conv3d_halo.hip: LDS-halo forward kernel.  Returns 1 if it handled the call, 0 if the shape is not eligible
// (the caller then uses the generic gather kernel), <0 on error.
size_t halo_fwd_workspace_bytes(int Cin, int Cout);
int halo_fwd_try(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                 const ConvGeom& g, int Cout, int act, float slope, void* workspace, size_t workspace_bytes,
                 hipStream_t stream, int force = 0, int debug = 0, bool packed_already = false, PackJobs* collect = nullptr);

size_t halo_dgrad_workspace_bytes(int Cin, int Cout);
// (packed_already: the workspace still holds this weight's fragment image from an earlier call — sg_conv3d_k4s2p1_dgrad_keep)
int halo_dgrad_try(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin, int Cin_total,
                   const ConvGeom& g, int Cout, int act, float slope, void* workspace, size_t workspace_bytes,
                   hipStream_t stream, int force = 0, bool packed_already = false, PackJobs* collect = nullptr);

size_t halo_wgrad_workspace_bytes(int batch, int Cin, int Cout, int OD, int OH, int OW);
int halo_wgrad_try(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, const ConvGeom& g, int Cout,
                   void* workspace, size_t workspace_bytes, hipStream_t stream, int force = 0, bool dy_packed = false);
// the packed-dy image of the halo weight-gradient kernel written by dy's producer (conv3d_halo.hip)
int halo_wgrad_dy_image_plan(int batch, int Cin, int Cout, const ConvGeom& g, size_t workspace_bytes, int* mt_total, long* nslice);
int halo_act_bwd_pack8_launch(const float* y, const float* dy, float* dz, float* rowsum, void* ap, long rows, int C, long nslice,
                              int act, float slope, hipStream_t stream);

// conv3d_edge.hip: layers with one channel on the voxel-grid side (Cin == 1).  Same return convention as the halo_*_try.
size_t edge_fwd_workspace_bytes(int batch, int OD, int OH, int OW);
size_t edge_wgrad_workspace_bytes(int batch, int OD, int OH, int OW);
size_t edge_dgrad_workspace_bytes(int batch, int OD, int OH, int OW);
int edge_fwd_try(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                 const ConvGeom& g, int Cout, int act, float slope, void* workspace, size_t workspace_bytes,
                 hipStream_t stream, int force = 0);
int edge_wgrad_try(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, const ConvGeom& g, int Cout,
                   void* workspace, size_t workspace_bytes, hipStream_t stream, int force = 0, const float* y = nullptr,
                   int act = 0, float slope = 0.f, float* db = nullptr);
int edge_dgrad_try(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin, int Cin_total,
                   const ConvGeom& g, int Cout, int act, float slope, void* workspace, size_t workspace_bytes,
                   hipStream_t stream, int force = 0);

int edge_dgrad_stream_try(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin, int Cin_total,
                          const ConvGeom& g, int Cout, int act, float slope, hipStream_t stream, const float* in_scale,
                          const float* in_shift, int in_act, float in_slope, int samples_per_group = 0, long out_group_stride = 0,
                          int form = 0);

}  // namespace sg

// ---- tuning switches: COMPILE-TIME constants (scripts/ab_build.sh builds variants with -D...); the product library reads no
// environment variable and keeps no mutable global state (include/shapegan_hip.h) ----------------------------------------------------
#ifndef SG_NO_EDGE
#define SG_NO_EDGE 0            // bits: which one-channel ("edge") kernels are switched off (A/B builds)
#endif
#ifndef SG_FWD_C1_LDS
#define SG_FWD_C1_LDS 1         // the LDS-staged Conv3d(1 -> C) forward at 32- / 64-wide grids (0: the gather form)
#endif
#ifndef SG_CONVT_ALL
#define SG_CONVT_ALL 1          // convT_c1_all_kernel (all 64 taps per workgroup) from 192 samples on (0: the both-parity kernel, A/B)
#endif
#ifndef SG_CONVT_MIN_BATCH
#define SG_CONVT_MIN_BATCH 48   // the plane-streaming ConvT(C -> 1) from this many samples on
#endif
