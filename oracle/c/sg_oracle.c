/* oracle/c/sg_oracle.c — TEST INFRASTRUCTURE (checker only; never linked into libshapegan_hip.so).
 *
 * Plain-C restatement of the ATen op semantics the reference's nn.Modules invoke on the hot path:
 *   nn.Conv3d(k=4,s=2,p=1)          model/gan.py:49-53, model/autoencoder.py:16-24, model/progressive_gan.py:38
 *   nn.ConvTranspose3d(k=4,s=2,p=1) model/gan.py:13-21, model/autoencoder.py:55-63
 *   SDFNet.forward                  model/sdf_net.py:56-61
 * written as the defining sums (no tiling, no reordering tricks), double accumulation, fp32 in/out.
 * Pinned against torch CPU ops and the golden SDFNet known answers in tests/test_oracle_pins.py.
 * Build: make -C oracle/c   (gcc -O2 -fopenmp -shared) -> oracle/_build/libsg_oracle.so
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define IDX5(n, c, d, h, w, C, D, H, W) (((((size_t)(n) * (C) + (c)) * (D) + (d)) * (H) + (h)) * (W) + (w))

/* y[n,co,od,oh,ow] = b[co] + sum_{ci,kd,kh,kw} x[n,ci,2od+kd-1,2oh+kh-1,2ow+kw-1] * w[co,ci,kd,kh,kw] */
void oracle_conv3d_k4s2p1_fwd(const float* x, const float* w, const float* b, float* y, int N, int Ci, int Co, int D,
                              int H, int W) {
    const int OD = D / 2, OH = H / 2, OW = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int od = 0; od < OD; ++od)
                for (int oh = 0; oh < OH; ++oh)
                    for (int ow = 0; ow < OW; ++ow) {
                        double acc = b ? b[co] : 0.0;
                        for (int ci = 0; ci < Ci; ++ci)
                            for (int kd = 0; kd < 4; ++kd) {
                                const int id = 2 * od + kd - 1;
                                if (id < 0 || id >= D) continue;
                                for (int kh = 0; kh < 4; ++kh) {
                                    const int ih = 2 * oh + kh - 1;
                                    if (ih < 0 || ih >= H) continue;
                                    for (int kw = 0; kw < 4; ++kw) {
                                        const int iw = 2 * ow + kw - 1;
                                        if (iw < 0 || iw >= W) continue;
                                        acc += (double)x[IDX5(n, ci, id, ih, iw, Ci, D, H, W)] *
                                               (double)w[((((size_t)co * Ci + ci) * 4 + kd) * 4 + kh) * 4 + kw];
                                    }
                                }
                            }
                        y[IDX5(n, co, od, oh, ow, Co, OD, OH, OW)] = (float)acc;
                    }
}

/* dx = adjoint of the above w.r.t. x (scatter form): dx[n,ci,2od+kd-1,...] += dy[n,co,od,...] * w[co,ci,kd,kh,kw].
 * This is also ConvTranspose3d(k4,s2,p1).forward with weight [Cin_T=Co, Cout_T=Ci, 4,4,4] (+ bias[ci]). */
void oracle_conv3d_k4s2p1_dgrad(const float* dy, const float* w, const float* b, float* dx, int N, int Ci, int Co, int D,
                                int H, int W) {
    const int OD = D / 2, OH = H / 2, OW = W / 2;
    const size_t vol = (size_t)D * H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Ci; ++ci) {
            double* acc = (double*)calloc(vol, sizeof(double));
            for (int co = 0; co < Co; ++co)
                for (int od = 0; od < OD; ++od)
                    for (int oh = 0; oh < OH; ++oh)
                        for (int ow = 0; ow < OW; ++ow) {
                            const double g = dy[IDX5(n, co, od, oh, ow, Co, OD, OH, OW)];
                            for (int kd = 0; kd < 4; ++kd) {
                                const int id = 2 * od + kd - 1;
                                if (id < 0 || id >= D) continue;
                                for (int kh = 0; kh < 4; ++kh) {
                                    const int ih = 2 * oh + kh - 1;
                                    if (ih < 0 || ih >= H) continue;
                                    for (int kw = 0; kw < 4; ++kw) {
                                        const int iw = 2 * ow + kw - 1;
                                        if (iw < 0 || iw >= W) continue;
                                        acc[((size_t)id * H + ih) * W + iw] +=
                                            g * (double)w[((((size_t)co * Ci + ci) * 4 + kd) * 4 + kh) * 4 + kw];
                                    }
                                }
                            }
                        }
            for (size_t e = 0; e < vol; ++e) dx[((size_t)n * Ci + ci) * vol + e] = (float)(acc[e] + (b ? b[ci] : 0.0));
            free(acc);
        }
}

/* dw[co,ci,kd,kh,kw] = sum_{n,od,oh,ow} dy[n,co,od,oh,ow] * x[n,ci,2od+kd-1,2oh+kh-1,2ow+kw-1] */
void oracle_conv3d_k4s2p1_wgrad(const float* dy, const float* x, float* dw, int N, int Ci, int Co, int D, int H, int W) {
    const int OD = D / 2, OH = H / 2, OW = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int kd = 0; kd < 4; ++kd)
                for (int kh = 0; kh < 4; ++kh)
                    for (int kw = 0; kw < 4; ++kw) {
                        double acc = 0.0;
                        for (int n = 0; n < N; ++n)
                            for (int od = 0; od < OD; ++od) {
                                const int id = 2 * od + kd - 1;
                                if (id < 0 || id >= D) continue;
                                for (int oh = 0; oh < OH; ++oh) {
                                    const int ih = 2 * oh + kh - 1;
                                    if (ih < 0 || ih >= H) continue;
                                    for (int ow = 0; ow < OW; ++ow) {
                                        const int iw = 2 * ow + kw - 1;
                                        if (iw < 0 || iw >= W) continue;
                                        acc += (double)dy[IDX5(n, co, od, oh, ow, Co, OD, OH, OW)] *
                                               (double)x[IDX5(n, ci, id, ih, iw, Ci, D, H, W)];
                                    }
                                }
                            }
                        dw[((((size_t)co * Ci + ci) * 4 + kd) * 4 + kh) * 4 + kw] = (float)acc;
                    }
}

/* SDFNet.forward (model/sdf_net.py:56-61): params in state_dict order W1,b1,...,W8,b8; latent per point. */
void oracle_sdfnet_fwd(const float* points, const float* latent, int L, const float* const* params, float* out, long N) {
    const int KIN = 3 + L, Hd = 256;
#pragma omp parallel
    {
        float* in = (float*)malloc(sizeof(float) * (size_t)(KIN));
        float* a = (float*)malloc(sizeof(float) * (size_t)(Hd + KIN));
        float* t = (float*)malloc(sizeof(float) * (size_t)Hd);
#pragma omp for schedule(static)
        for (long p = 0; p < N; ++p) {
            for (int c = 0; c < 3; ++c) in[c] = points[p * 3 + c];
            for (int k = 0; k < L; ++k) in[3 + k] = latent[p * (long)L + k];
            int width = KIN;
            memcpy(a, in, sizeof(float) * (size_t)KIN);
            for (int layer = 0; layer < 7; ++layer) {
                const float* Wl = params[2 * layer];
                const float* bl = params[2 * layer + 1];
                if (layer == 4) { /* x = cat(x, input), model/sdf_net.py:59 */
                    memcpy(a + Hd, in, sizeof(float) * (size_t)KIN);
                    width = Hd + KIN;
                }
                for (int o = 0; o < Hd; ++o) {
                    double s = bl[o];
                    for (int k = 0; k < width; ++k) s += (double)Wl[(size_t)o * width + k] * (double)a[k];
                    t[o] = s > 0.0 ? (float)s : 0.f;
                }
                memcpy(a, t, sizeof(float) * (size_t)Hd);
                width = Hd;
            }
            double s = params[15][0];
            for (int k = 0; k < Hd; ++k) s += (double)params[14][k] * (double)a[k];
            out[p] = (float)tanh(s);
        }
        free(in);
        free(a);
        free(t);
    }
}

/* per-channel batch statistics of x[N,C,S]: mean and biased variance (nn.BatchNorm normalisation statistics) */
void oracle_bn_stats(const float* x, double* mean, double* var, int N, int C, long S) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s = 0, q = 0;
        for (int n = 0; n < N; ++n)
            for (long e = 0; e < S; ++e) s += x[((size_t)n * C + c) * S + e];
        const double m = s / ((double)N * S);
        for (int n = 0; n < N; ++n)
            for (long e = 0; e < S; ++e) {
                const double d = x[((size_t)n * C + c) * S + e] - m;
                q += d * d;
            }
        mean[c] = m;
        var[c] = q / ((double)N * S);
    }
}
