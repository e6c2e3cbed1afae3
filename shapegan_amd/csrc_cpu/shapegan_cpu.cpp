// shapegan_amd/csrc_cpu/shapegan_cpu.cpp — libshapegan_cpu.so: the plain-C++ twin of the C ABI (include/shapegan_hip.h).
//
// SURVEY.md 8b asks for "a CPU twin (`*_cpu`) of each [entry point] for the no-GPU config and unit tests"; BASELINE configs[0]
// is `train_autoencoder.py classic ... on CPU (plumbing, no GPU)`.  Every function here is `sg_<name>_cpu` with the argument
// list of `sg_<name>` (the stream and workspace arguments are accepted and ignored; pointers are host pointers).  It is a
// SEPARATE implementation written against the header's contracts — straightforward loops with OpenMP over the outer index,
// no tiling heroics: correctness plumbing, not a performance path — and it shares no code with oracle/ (test infrastructure)
// or with the HIP kernels.  The Python shells pick it only for tensors that live on the CPU (shapegan_amd/lib.py); GPU
// tensors never come here and there is no fallback in either direction.
//
// Opaque-buffer contracts it keeps compatible with the sizes the HIP library reports (host-side helpers, callable without a
// GPU): sg_sdfnet_packed_floats (this twin stores the plain weights at the front of the buffer), sg_sdfnet_bwd_blocks
// (bias partials: column 0 carries the row sums, the other columns zeros).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef void* hipStream_t_;
#define SG_OK 0
#define SG_SDFNET_PARTIAL_ROW (14 * 256 + 32)   // include/shapegan_hip.h (the twin does not include the HIP header)
#define SG_SDFGEN_PARTIAL_ROW (SG_SDFNET_PARTIAL_ROW + 14 * 256)
#define SG_ERR_ARG (-1)
enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_RELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };

static thread_local char g_err[512];
#define CPU_CHECK(cond)                                                                  \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            snprintf(g_err, sizeof(g_err), "%s: bad argument: %s", __func__, #cond);     \
            return SG_ERR_ARG;                                                           \
        }                                                                                \
    } while (0)

static inline float apply_act(float v, int act, float slope) {
    switch (act) {
        case ACT_LEAKY: return v > 0.f ? v : v * slope;
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_TANH: return tanhf(v);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}
static inline float act_grad_out(float y, float dy, int act, float slope) {   // derivative through the activation OUTPUT
    switch (act) {
        case ACT_LEAKY: return y > 0.f ? dy : dy * slope;
        case ACT_RELU: return y > 0.f ? dy : 0.f;
        case ACT_TANH: return dy * (1.f - y * y);
        case ACT_SIGMOID: return dy * y * (1.f - y);
        default: return dy;
    }
}

extern "C" {

const char* sg_cpu_last_error(void) { return g_err; }

// ---- K1 / K2: Conv3d / ConvTranspose3d (kernel 4, stride 2, padding 1) ------------------------------------------------
int sg_conv3d_k4s2p1_fwd_cpu(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                             int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void*, size_t, void*) {
    CPU_CHECK(x && w && y && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0 && !(ID & 1) && !(IH & 1) && !(IW & 1));
    const int OD = ID / 2, OH = IH / 2, OW = IW / 2;
    const long I3 = (long)ID * IH * IW, O3 = (long)OD * OH * OW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < batch; ++n)
        for (int co = 0; co < Cout; ++co) {
            float* yo = y + ((long)n * Cout + co) * O3;
            const float b = bias ? bias[co] : 0.f;
            for (long e = 0; e < O3; ++e) yo[e] = b;
            for (int ci = 0; ci < Cin; ++ci) {
                const float* xi = x + ((long)n * Cx + ci) * I3;
                const float* wk = w + ((long)co * Cin_total + ci) * 64;
                for (int kd = 0; kd < 4; ++kd)
                    for (int kh = 0; kh < 4; ++kh)
                        for (int kw = 0; kw < 4; ++kw) {
                            const float wv = wk[kd * 16 + kh * 4 + kw];
                            const int ow0 = kw == 0 ? 1 : 0, ow1 = kw == 3 ? OW - 1 : OW;   // 0 <= 2ow + kw - 1 < IW
                            for (int od = 0; od < OD; ++od) {
                                const int id = 2 * od + kd - 1;
                                if ((unsigned)id >= (unsigned)ID) continue;
                                for (int oh = 0; oh < OH; ++oh) {
                                    const int ih = 2 * oh + kh - 1;
                                    if ((unsigned)ih >= (unsigned)IH) continue;
                                    const float* xr = xi + ((long)id * IH + ih) * IW + kw - 1;
                                    float* yr = yo + ((long)od * OH + oh) * OW;
                                    for (int ow = ow0; ow < ow1; ++ow) yr[ow] += wv * xr[2 * ow];
                                }
                            }
                        }
            }
            if (act != ACT_NONE)
                for (long e = 0; e < O3; ++e) yo[e] = apply_act(yo[e], act, slope);
        }
    return SG_OK;
}

int sg_conv3d_k4s2p1_dgrad_cpu(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                               int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void*, size_t,
                               void*) {
    CPU_CHECK(dy && w && dx && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0 && !(ID & 1) && !(IH & 1) && !(IW & 1));
    const int OD = ID / 2, OH = IH / 2, OW = IW / 2;
    const long I3 = (long)ID * IH * IW, O3 = (long)OD * OH * OW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < batch; ++n)
        for (int ci = 0; ci < Cin; ++ci) {
            float* xo = dx + ((long)n * Cx + ci) * I3;
            const float b = bias ? bias[ci] : 0.f;
            for (long e = 0; e < I3; ++e) xo[e] = b;
            for (int co = 0; co < Cout; ++co) {
                const float* yo = dy + ((long)n * Cout + co) * O3;
                const float* wk = w + ((long)co * Cin_total + ci) * 64;
                for (int kd = 0; kd < 4; ++kd)
                    for (int kh = 0; kh < 4; ++kh)
                        for (int kw = 0; kw < 4; ++kw) {
                            const float wv = wk[kd * 16 + kh * 4 + kw];
                            const int ow0 = kw == 0 ? 1 : 0, ow1 = kw == 3 ? OW - 1 : OW;
                            for (int od = 0; od < OD; ++od) {
                                const int id = 2 * od + kd - 1;
                                if ((unsigned)id >= (unsigned)ID) continue;
                                for (int oh = 0; oh < OH; ++oh) {
                                    const int ih = 2 * oh + kh - 1;
                                    if ((unsigned)ih >= (unsigned)IH) continue;
                                    float* xr = xo + ((long)id * IH + ih) * IW + kw - 1;
                                    const float* yr = yo + ((long)od * OH + oh) * OW;
                                    for (int ow = ow0; ow < ow1; ++ow) xr[2 * ow] += wv * yr[ow];
                                }
                            }
                        }
            }
            if (act != ACT_NONE)
                for (long e = 0; e < I3; ++e) xo[e] = apply_act(xo[e], act, slope);
        }
    return SG_OK;
}

int sg_conv3d_k4s2p1_wgrad_cpu(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx, int Cout,
                               int ID, int IH, int IW, void*, size_t, void*) {
    CPU_CHECK(dy && x && dw && batch > 0 && Cin > 0 && Cin <= Cin_total && Cin <= Cx && Cout > 0 && !(ID & 1) && !(IH & 1) && !(IW & 1));
    const int OD = ID / 2, OH = IH / 2, OW = IW / 2;
    const long I3 = (long)ID * IH * IW, O3 = (long)OD * OH * OW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            double acc[64];
            for (int k = 0; k < 64; ++k) acc[k] = 0.0;
            for (int n = 0; n < batch; ++n) {
                const float* yo = dy + ((long)n * Cout + co) * O3;
                const float* xi = x + ((long)n * Cx + ci) * I3;
                for (int kd = 0; kd < 4; ++kd)
                    for (int kh = 0; kh < 4; ++kh)
                        for (int kw = 0; kw < 4; ++kw) {
                            const int ow0 = kw == 0 ? 1 : 0, ow1 = kw == 3 ? OW - 1 : OW;
                            float s = 0.f;
                            for (int od = 0; od < OD; ++od) {
                                const int id = 2 * od + kd - 1;
                                if ((unsigned)id >= (unsigned)ID) continue;
                                for (int oh = 0; oh < OH; ++oh) {
                                    const int ih = 2 * oh + kh - 1;
                                    if ((unsigned)ih >= (unsigned)IH) continue;
                                    const float* xr = xi + ((long)id * IH + ih) * IW + kw - 1;
                                    const float* yr = yo + ((long)od * OH + oh) * OW;
                                    float r = 0.f;
                                    for (int ow = ow0; ow < ow1; ++ow) r += yr[ow] * xr[2 * ow];
                                    s += r;
                                }
                            }
                            acc[kd * 16 + kh * 4 + kw] += (double)s;
                        }
            }
            float* d = dw + ((long)co * Cin_total + ci) * 64;
            for (int k = 0; k < 64; ++k) d[k] = (float)acc[k];
        }
    return SG_OK;
}

int sg_conv3d_k4s2p1_dgrad_keep_cpu(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                                    int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* ws,
                                    size_t wb, int, void* st) {     // (nothing is packed here: nothing to keep)
    return sg_conv3d_k4s2p1_dgrad_cpu(dy, w, bias, dx, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, act, slope, ws, wb, st);
}
int sg_conv3d_k4s2p1_fwd_keep_cpu(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                                  int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* ws, size_t wb, int, void* st) {
    return sg_conv3d_k4s2p1_fwd_cpu(x, w, bias, y, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, act, slope, ws, wb, st);
}
int sg_conv3d_k4s2p1_pack_images_cpu(int n, const int* kinds, const float* const* weights, void* const* workspaces,
                                     const size_t* workspace_bytes, const int* dims, int* served, void*) {
    CPU_CHECK(n > 0 && n <= 8 && kinds && weights && workspaces && workspace_bytes && dims && served);
    for (int i = 0; i < n; ++i) served[i] = 0;     // the twin reads the weights in place: there is no image to keep
    return SG_OK;
}
// (the twin's weight-gradient loops read dy in place: sg_conv3d_k4s2p1_wgrad_dy_image is a host query of the HIP library and
// answers for GPU tensors only; these two exist so that every entry point has its twin)
int sg_act_bwd_rowsum_pack8_cpu(const float* y, const float* dy, float* dz, float* rowsum, void*, long N, int C, long, int act,
                                float slope, void*) {
    CPU_CHECK(y && dy && dz && rowsum && N > 0 && C > 0);
    for (long r = 0; r < N * C; ++r) {
        double s = 0;
        for (int e = 0; e < 512; ++e) {
            const float v = act_grad_out(y[r * 512 + e], dy[r * 512 + e], act, slope);
            dz[r * 512 + e] = v;
            s += (double)v;
        }
        rowsum[r] = (float)s;
    }
    return SG_OK;
}
int sg_conv3d_k4s2p1_wgrad_prepacked_cpu(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                                         int Cout, int ID, int IH, int IW, void* ws, size_t wb, void* st) {
    return sg_conv3d_k4s2p1_wgrad_cpu(dy, x, dw, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, ws, wb, st);
}
int sg_convT3d_k4s2p1_fwd_cpu(const float* x, const float* w, const float* bias, float* y, int batch, int Cin_T, int Cout_T,
                              int ID, int IH, int IW, int act, float slope, void* ws, size_t wb, void* st) {
    return sg_conv3d_k4s2p1_dgrad_cpu(x, w, bias, y, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW, act, slope, ws,
                                      wb, st);
}
int sg_convT3d_k4s2p1_to1_pre_cpu(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                                  const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW,
                                  int act, float slope, void* st) {
    CPU_CHECK(x && w && y && in_scale && in_shift && batch > 0 && C > 0 && C <= 64 && IH * IW <= 256);
    CPU_CHECK(in_act == ACT_NONE || in_act == ACT_LEAKY || in_act == ACT_RELU);
    const long S = (long)ID * IH * IW, total = (long)batch * C * S;
    std::vector<float> t((size_t)total);
#pragma omp parallel for schedule(static)
    for (long e = 0; e < total; ++e) {
        const int c = (int)((e / S) % C);
        t[(size_t)e] = apply_act(x[e] * in_scale[c] + in_shift[c], in_act, in_slope);
    }
    return sg_conv3d_k4s2p1_dgrad_cpu(t.data(), w, bias, y, batch, 1, 1, 1, C, 2 * ID, 2 * IH, 2 * IW, act, slope, nullptr, 0, st);
}
int sg_convT3d_k4s2p1_to1_pre_grouped_cpu(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                                          const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH,
                                          int IW, int act, float slope, int spg, long y_group_stride, void* st) {
    CPU_CHECK(spg > 0 && batch % spg == 0);
    const long S = (long)ID * IH * IW;
    for (int g = 0; g < batch / spg; ++g) {
        const int rc = sg_convT3d_k4s2p1_to1_pre_cpu(x + (long)g * spg * C * S, w, bias, y + (long)g * y_group_stride,
                                                     in_scale + (long)g * C, in_shift + (long)g * C, in_act, in_slope, spg, C, ID, IH,
                                                     IW, act, slope, st);
        if (rc) return rc;
    }
    return SG_OK;
}
int sg_convT3d_k4s2p1_dgrad_cpu(const float* dy, const float* w, float* dx, int batch, int Cin_T, int Cout_T, int ID, int IH,
                                int IW, void* ws, size_t wb, void* st) {
    return sg_conv3d_k4s2p1_fwd_cpu(dy, w, nullptr, dx, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW, ACT_NONE, 0.f,
                                    ws, wb, st);
}
int sg_convT3d_k4s2p1_wgrad_cpu(const float* dy, const float* x, float* dw, int batch, int Cin_T, int Cout_T, int ID, int IH,
                                int IW, void* ws, size_t wb, void* st) {
    return sg_conv3d_k4s2p1_wgrad_cpu(x, dy, dw, batch, Cout_T, Cout_T, Cout_T, Cin_T, 2 * ID, 2 * IH, 2 * IW, ws, wb, st);
}

// ---- K3 / K6: GEMM family -------------------------------------------------------------------------------------------------
int sg_gemm_cpu(const float* A, long sai, long sak, const float* B, long sbk, long sbj, float* C, long sci, long scj,
                const float* bias_i, const float* bias_j, int bias_j_shift, int M, int N, int K, int act, float slope, void*,
                size_t, void*) {
    CPU_CHECK(A && B && C && M > 0 && N > 0 && K > 0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; ++i) {
        std::vector<double> row(N, 0.0);
        for (int k = 0; k < K; ++k) {
            const double a = A[i * sai + k * sak];
            const float* bk = B + k * sbk;
            for (int j = 0; j < N; ++j) row[j] += a * (double)bk[j * sbj];
        }
        for (int j = 0; j < N; ++j) {
            float v = (float)row[j] + (bias_i ? bias_i[i] : 0.f) + (bias_j ? bias_j[j >> bias_j_shift] : 0.f);
            C[i * sci + j * scj] = apply_act(v, act, slope);
        }
    }
    return SG_OK;
}
int sg_gemm_nt_cpu(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, long K, void*, size_t,
                   void*) {
    CPU_CHECK(A && B && C && M > 0 && N > 0 && K > 0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            const float *a = A + i * lda, *b = B + j * ldb;
            double s = 0;
            for (long k = 0; k < K; ++k) s += (double)a[k] * (double)b[k];
            C[i * ldc + j] = (float)s;
        }
    return SG_OK;
}
int sg_gemm_nt_batched_cpu(const float* A, const long* a_off, long lda, const float* B, const long* b_off, long ldb, float* C,
                           const long* c_off, const long* ldc, int batch, int M, int N, long K, void* ws, size_t wb, void* st) {
    CPU_CHECK(A && B && C && a_off && b_off && c_off && ldc && batch > 0 && batch <= 8);
    for (int b = 0; b < batch; ++b) {
        const int rc = sg_gemm_nt_cpu(A + a_off[b], lda, B + b_off[b], ldb, C + c_off[b], ldc[b], M, N, K, ws, wb, st);
        if (rc) return rc;
    }
    return SG_OK;
}
int sg_gemm_nt_batched_lnrelu_cpu(const float* A, const long* a_off, long lda, const float* B, const long* b_off, long ldb,
                                  const float* gamma, const float* beta, const long* g_off, float* C, const long* c_off, const long* ldc,
                                  int batch, int M, int N, long K, void*, size_t, void*) {
    CPU_CHECK(A && B && C && a_off && b_off && c_off && ldc && gamma && beta && g_off && batch > 0 && batch <= 8 && M > 0 && N > 0 && K > 0);
    for (int m = 0; m < batch; ++m) {
        const float *Am = A + a_off[m], *Bm = B + b_off[m], *gm = gamma + g_off[m], *bt = beta + g_off[m];
        float* Cm = C + c_off[m];
#pragma omp parallel for collapse(2) schedule(static)
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                const float *a = Am + i * lda, *b = Bm + j * ldb;
                double s = 0;
                for (long k = 0; k < K; ++k) {
                    const float h = gm[j] * b[k] + bt[j];
                    s += (double)a[k] * (double)(h > 0.f ? h : 0.f);
                }
                Cm[i * ldc[m] + j] = (float)s;
            }
    }
    return SG_OK;
}
int sg_colsum_cpu(const float* x, float* out, int rows, int cols, long ld, void*) {
    CPU_CHECK(x && out && rows > 0 && cols > 0);
    for (int j = 0; j < cols; ++j) {
        double s = 0;
        for (int i = 0; i < rows; ++i) s += x[i * ld + j];
        out[j] = (float)s;
    }
    return SG_OK;
}
int sg_rowsum_cpu(const float* x, float* out, long rows, long len, long ld, void*) {
    CPU_CHECK(x && out && rows > 0 && len > 0);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        double s = 0;
        for (long e = 0; e < len; ++e) s += x[r * ld + e];
        out[r] = (float)s;
    }
    return SG_OK;
}
int sg_rowsum_multi_cpu(const float* x, float* const* outs, const long* out_strides, int ndst, long rows_per_dst, long len,
                        long ld, void*) {
    CPU_CHECK(x && outs && ndst > 0 && ndst <= 8 && rows_per_dst > 0 && len > 0);
    for (int d = 0; d < ndst; ++d)
        for (long r = 0; r < rows_per_dst; ++r) {
            double s = 0;
            const float* p = x + ((long)d * rows_per_dst + r) * ld;
            for (long e = 0; e < len; ++e) s += p[e];
            outs[d][r * (out_strides ? out_strides[d] : 1)] = (float)s;
        }
    return SG_OK;
}
int sg_segsum_cpu(const float* x, float* out, long rows, long ld, const int64_t* seg_off, long nseg, void*) {
    CPU_CHECK(x && out && seg_off && rows > 0 && nseg > 0);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r)
        for (long s = 0; s < nseg; ++s) {
            double acc = 0;
            for (long e = seg_off[s]; e < seg_off[s + 1]; ++e) acc += x[r * ld + e];
            out[r * nseg + s] = (float)acc;
        }
    return SG_OK;
}
int sg_colsum_tall_cpu(const float* x, float* out, long batch, long batch_stride, long rows, int cols, long ld, void*, size_t,
                       void*) {
    CPU_CHECK(x && out && batch > 0 && rows > 0 && cols > 0);
    for (long b = 0; b < batch; ++b)
        for (int c = 0; c < cols; ++c) {
            double s = 0;
            for (long r = 0; r < rows; ++r) s += x[b * batch_stride + r * ld + c];
            out[b * cols + c] = (float)s;
        }
    return SG_OK;
}

// ---- K4: BatchNorm ---------------------------------------------------------------------------------------------------------
int sg_bn_train_fwd_cpu(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                        float* running_mean, float* running_var, long long* num_batches_tracked, int N, int C, long S, float eps,
                        float momentum, int act, float slope, void*, size_t, void*) {
    CPU_CHECK(x && gamma && beta && y && save_mean && save_invstd && N > 0 && C > 0 && S > 0);
    const double cnt = (double)N * S;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s = 0, s2 = 0;
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) s += p[e];
        }
        const double mu = s / cnt;
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) s2 += (p[e] - mu) * (p[e] - mu);
        }
        const double var = s2 / cnt;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[c] = (float)mu;
        save_invstd[c] = is;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(cnt > 1 ? s2 / (cnt - 1) : var);
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((long)n * C + c) * S;
            float* q = y + ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) q[e] = apply_act(gamma[c] * ((p[e] - (float)mu) * is) + beta[c], act, slope);
        }
    }
    if (num_batches_tracked) *num_batches_tracked += 1;
    return SG_OK;
}
int sg_bn_train_stats_cpu(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float* scale, float* shift,
                          int N, int C, long S, float eps, float momentum, void*, size_t, void*) {
    CPU_CHECK(x && gamma && beta && save_mean && save_invstd && scale && shift && N > 0 && C > 0 && S > 0);
    const double cnt = (double)N * S;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s = 0, s2 = 0;
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) s += p[e];
        }
        const double mu = s / cnt;
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) s2 += (p[e] - mu) * (p[e] - mu);
        }
        const double var = s2 / cnt;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[c] = (float)mu;
        save_invstd[c] = is;
        scale[c] = gamma[c] * is;
        shift[c] = beta[c] - (float)mu * scale[c];
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(cnt > 1 ? s2 / (cnt - 1) : var);
    }
    if (num_batches_tracked) *num_batches_tracked += 1;
    return SG_OK;
}
int sg_bn_train_fwd_grouped_cpu(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                                float* save_invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                                int groups, int N, int C, long S, float eps, float momentum, int act, float slope, void* ws, size_t wb,
                                void* st) {
    CPU_CHECK(groups > 0);
    for (int g = 0; g < groups; ++g) {     // the groups are independent batches, applied one after the other
        const long off = (long)g * N * C * S;
        const int rc = sg_bn_train_fwd_cpu(x + off, gamma, beta, y + off, save_mean + (long)g * C, save_invstd + (long)g * C,
                                           running_mean, running_var, num_batches_tracked, N, C, S, eps, momentum, act, slope, ws, wb,
                                           st);
        if (rc) return rc;
    }
    return SG_OK;
}
int sg_bn_train_stats_grouped_cpu(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                                  float* running_mean, float* running_var, long long* num_batches_tracked, float* scale,
                                  float* shift, int groups, int N, int C, long S, float eps, float momentum, void* ws, size_t wb,
                                  void* st) {
    CPU_CHECK(groups > 0);
    for (int g = 0; g < groups; ++g) {
        const int rc = sg_bn_train_stats_cpu(x + (long)g * N * C * S, gamma, beta, save_mean + (long)g * C, save_invstd + (long)g * C,
                                             running_mean, running_var, num_batches_tracked, scale + (long)g * C,
                                             shift + (long)g * C, N, C, S, eps, momentum, ws, wb, st);
        if (rc) return rc;
    }
    return SG_OK;
}
int sg_bn_eval_fwd_cpu(const float* x, const float* gamma, const float* beta, float* y, const float* running_mean,
                       const float* running_var, float* save_mean, float* save_invstd, int N, int C, long S, float eps, int act,
                       float slope, void*) {
    CPU_CHECK(x && gamma && beta && y && running_mean && running_var && save_mean && save_invstd && N > 0 && C > 0 && S > 0);
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float mu = running_mean[c], is = 1.f / sqrtf(running_var[c] + eps);
        save_mean[c] = mu;
        save_invstd[c] = is;
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((long)n * C + c) * S;
            float* q = y + ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) q[e] = apply_act(gamma[c] * ((p[e] - mu) * is) + beta[c], act, slope);
        }
    }
    return SG_OK;
}
int sg_bn_bwd_cpu(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                  const float* save_invstd, float* dx, float* dgamma, float* dbeta, int N, int C, long S, int train, int act,
                  float slope, void*, size_t, void*) {
    CPU_CHECK(dy && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && N > 0 && C > 0 && S > 0);
    const double cnt = (double)N * S;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float mu = save_mean[c], is = save_invstd[c], g = gamma[c], b = beta[c];
        double s1 = 0, s2 = 0;
        for (int n = 0; n < N; ++n) {
            const long base = ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) {
                const float xh = (x[base + e] - mu) * is;
                const float gg = act_grad_out(apply_act(g * xh + b, act, slope), dy[base + e], act, slope);
                s1 += gg;
                s2 += (double)gg * xh;
            }
        }
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        const float m1 = (float)(s1 / cnt), m2 = (float)(s2 / cnt);
        for (int n = 0; n < N; ++n) {
            const long base = ((long)n * C + c) * S;
            for (long e = 0; e < S; ++e) {
                const float xh = (x[base + e] - mu) * is;
                const float gg = act_grad_out(apply_act(g * xh + b, act, slope), dy[base + e], act, slope);
                dx[base + e] = g * is * (train ? gg - m1 - xh * m2 : gg);
            }
        }
    }
    return SG_OK;
}

// ---- K5: activations ---------------------------------------------------------------------------------------------------------
int sg_act_fwd_cpu(const float* x, float* y, long n, int act, float slope, void*) {
    CPU_CHECK(x && y && n > 0);
#pragma omp parallel for schedule(static)
    for (long e = 0; e < n; ++e) y[e] = apply_act(x[e], act, slope);
    return SG_OK;
}
int sg_act_bwd_cpu(const float* y, const float* dy, float* dx, long n, int act, float slope, void*) {
    CPU_CHECK(y && dy && dx && n > 0);
#pragma omp parallel for schedule(static)
    for (long e = 0; e < n; ++e) dx[e] = act_grad_out(y[e], dy[e], act, slope);
    return SG_OK;
}
int sg_act_bwd_rowsum_cpu(const float* y, const float* dy, float* dx, float* rowsum, long rows, long S, int act, float slope, void*) {
    CPU_CHECK(y && dy && dx && rowsum && rows > 0 && S > 0);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        double s = 0;
        for (long e = r * S; e < (r + 1) * S; ++e) {
            dx[e] = act_grad_out(y[e], dy[e], act, slope);
            s += dx[e];
        }
        rowsum[r] = (float)s;
    }
    return SG_OK;
}
// weight + bias gradient through the activation (header: sg_conv3d_k4s2p1_wgrad_act): dz = dy * act'(y), then the plain forms
int sg_conv3d_k4s2p1_wgrad_act_cpu(const float* dy, const float* y, const float* x, float* dw, float* db, int batch, int Cin,
                                   int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void*, size_t,
                                   void*) {
    CPU_CHECK(dy && y && x && dw && db && batch > 0 && Cout > 0 && !(ID & 1) && !(IH & 1) && !(IW & 1));
    const long O3 = (long)(ID / 2) * (IH / 2) * (IW / 2), n = (long)batch * Cout * O3;
    std::vector<float> dz((size_t)n);
    for (long e = 0; e < n; ++e) dz[e] = act_grad_out(y[e], dy[e], act, slope);
    for (int co = 0; co < Cout; ++co) {
        double t = 0;
        for (int b = 0; b < batch; ++b)
            for (long e = 0; e < O3; ++e) t += dz[((long)b * Cout + co) * O3 + e];
        db[co] = (float)t;
    }
    return sg_conv3d_k4s2p1_wgrad_cpu(dz.data(), x, dw, batch, Cin, Cin_total, Cx, Cout, ID, IH, IW, nullptr, 0, nullptr);
}
int sg_act_bwd_dy_cpu(const float* y, const float* dy, const float* ggx, float* out, long n, int act, void*) {
    CPU_CHECK(y && dy && ggx && out && n > 0 && (act == ACT_TANH || act == ACT_SIGMOID));
    for (long e = 0; e < n; ++e) out[e] = ggx[e] * dy[e] * (act == ACT_TANH ? -2.f * y[e] : 1.f - 2.f * y[e]);
    return SG_OK;
}

// ---- K7: SDFNet ------------------------------------------------------------------------------------------------------------
// packed buffer of this twin (all row-major): W1k [256][KU] | W2 W3 W4 [256][256] | W5x [256][256] | W5i [256][KU] | W6 W7 |
// w8 [256] | b1..b7 [7][256] | b8
struct CpuSdf {
    int KU;
    const float *W1k, *W2, *W3, *W4, *W5x, *W5i, *W6, *W7, *w8, *b, *b8;
};
static CpuSdf sdf_view(const float* p, int KU) {
    CpuSdf v;
    v.KU = KU;
    v.W1k = p;
    p += 256L * KU;
    v.W2 = p; p += 65536;
    v.W3 = p; p += 65536;
    v.W4 = p; p += 65536;
    v.W5x = p; p += 65536;
    v.W5i = p; p += 256L * KU;
    v.W6 = p; p += 65536;
    v.W7 = p; p += 65536;
    v.w8 = p; p += 256;
    v.b = p; p += 7 * 256;
    v.b8 = p;
    return v;
}
int sg_sdfnet_pack_cpu(const float* const* params, int latent, int kin_used, float* packed, void*) {
    CPU_CHECK(params && packed && latent >= 0 && kin_used >= 3 && kin_used <= 3 + latent);
    const int kin = 3 + latent;
    float* p = packed;
    for (int o = 0; o < 256; ++o) memcpy(p + (long)o * kin_used, params[0] + (long)o * kin, sizeof(float) * kin_used);
    p += 256L * kin_used;
    for (int l = 1; l <= 3; ++l, p += 65536) memcpy(p, params[2 * l], sizeof(float) * 65536);
    for (int o = 0; o < 256; ++o) memcpy(p + o * 256, params[8] + (long)o * (256 + kin), sizeof(float) * 256);
    p += 65536;
    for (int o = 0; o < 256; ++o) memcpy(p + (long)o * kin_used, params[8] + (long)o * (256 + kin) + 256, sizeof(float) * kin_used);
    p += 256L * kin_used;
    for (int l = 5; l <= 6; ++l, p += 65536) memcpy(p, params[2 * l], sizeof(float) * 65536);
    memcpy(p, params[14], sizeof(float) * 256);
    p += 256;
    for (int l = 0; l < 7; ++l, p += 256) memcpy(p, params[2 * l + 1], sizeof(float) * 256);
    p[0] = params[15][0];
    return SG_OK;
}
// out[o] = relu?(b[o] + sum_k W[o][k] in[k]) for one point
static inline void dense(const float* W, int K, const float* in, const float* b, float* out, bool relu, bool accumulate) {
    for (int o = 0; o < 256; ++o) {
        const float* w = W + (long)o * K;
        float s = accumulate ? out[o] : (b ? b[o] : 0.f);
        for (int k = 0; k < K; ++k) s += w[k] * in[k];
        out[o] = s;
    }
    if (relu)
        for (int o = 0; o < 256; ++o) out[o] = out[o] > 0.f ? out[o] : 0.f;
}
int sg_sdfnet_fwd_cpu(const float* points, long points_period, const float* latent, const int64_t* latent_idx, int latent_size,
                      const float* packed, int kin_used, const float* zb1, const float* zb5, long points_per_shape,
                      const int* shape_index, float* out, float* acts, long ldn, long N, void*) {
    CPU_CHECK(points && packed && out && N > 0 && kin_used >= 3 && (kin_used == 3 || latent) && ((zb1 != nullptr) == (zb5 != nullptr)));
    CPU_CHECK(!zb1 || shape_index || points_per_shape > 0);
    const CpuSdf v = sdf_view(packed, kin_used);
    const int KU = kin_used, L = latent_size;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < N; ++p) {
        float xin[3 + 1024], h[256], h2[256];
        const long pi = points_period > 0 ? p % points_period : p;
        for (int c = 0; c < 3; ++c) xin[c] = points[pi * 3 + c];
        if (KU > 3) {
            const long row = latent_idx ? latent_idx[p] : p;
            for (int k = 0; k < KU - 3; ++k) xin[3 + k] = latent[row * L + k];
        }
        const long shape = zb1 ? (shape_index ? shape_index[p] : p / points_per_shape) : 0;
        auto save = [&](int layer, const float* a) {
            if (acts)
                for (int o = 0; o < 256; ++o) acts[((long)layer * 256 + o) * ldn + p] = a[o];
        };
        dense(v.W1k, KU, xin, zb1 ? zb1 + shape * 256 : v.b, h, true, false);
        save(0, h);
        dense(v.W2, 256, h, v.b + 256, h2, true, false);
        save(1, h2);
        dense(v.W3, 256, h2, v.b + 512, h, true, false);
        save(2, h);
        dense(v.W4, 256, h, v.b + 768, h2, true, false);
        save(3, h2);
        dense(v.W5x, 256, h2, zb5 ? zb5 + shape * 256 : v.b + 1024, h, false, false);
        dense(v.W5i, KU, xin, nullptr, h, true, true);
        save(4, h);
        dense(v.W6, 256, h, v.b + 1280, h2, true, false);
        save(5, h2);
        dense(v.W7, 256, h2, v.b + 1536, h, true, false);
        save(6, h);
        float s = v.b8[0];
        for (int k = 0; k < 256; ++k) s += v.w8[k] * h[k];
        out[p] = tanhf(s);
    }
    return SG_OK;
}
// dH_in[k] = sum_o W[o][k] dZ[o]
static inline void dense_t(const float* W, int K, const float* dz, float* dh, bool accumulate) {
    if (!accumulate)
        for (int k = 0; k < K; ++k) dh[k] = 0.f;
    for (int o = 0; o < 256; ++o) {
        const float g = dz[o];
        if (g == 0.f) continue;
        const float* w = W + (long)o * K;
        for (int k = 0; k < K; ++k) dh[k] += w[k] * g;
    }
}
// the header's tile layout of the backward partials (sg_sdfnet_bwd_blocks / sg_sdfnet_bwd_tile_start)
static void sdf_bwd_plan(long N, long& nbig, long& nsmall) {
    const long tiles = (N + 63) / 64, rem = tiles % 512, full = tiles - rem;
    if (rem > 256 && rem <= 384) {
        nbig = full + 256;
        nsmall = (N - nbig * 64 + 31) / 32;
    } else if (full == 0 || rem == 0 || rem > 384) {
        nbig = tiles;
        nsmall = 0;
    } else {
        nbig = full;
        nsmall = (N - full * 64 + 31) / 32;
    }
}
static long sdf_bwd_tiles(long N) {
    long nbig, nsmall;
    sdf_bwd_plan(N, nbig, nsmall);
    return nbig + nsmall;
}
static long sdf_bwd_tile_start(long N, long t) {
    long nbig, nsmall;
    sdf_bwd_plan(N, nbig, nsmall);
    const long p = t <= nbig ? t * 64 : nbig * 64 + (t - nbig) * 32;
    return p < N ? p : N;
}
int sg_sdfnet_bwd_cpu(const float* dout, const float* out, const float* acts, float* dz, float* dz8, float* bias_partials,
                      const float* points, long points_period, float* dx, long dx_ld, const float* packed, int kin_used,
                      long ldn, long N, void*) {
    CPU_CHECK(dout && out && acts && dz && dz8 && packed && N > 0 && kin_used >= 3 && kin_used <= 3 + 1024);
    const CpuSdf v = sdf_view(packed, kin_used);
    const float* Wt[7] = {nullptr, v.W2, v.W3, v.W4, v.W5x, v.W6, v.W7};   // W of layer l+1 maps H_l -> Z_{l+1}
#pragma omp parallel for schedule(static)
    for (long p = 0; p < N; ++p) {
        float g[256], gn[256], dxa[3 + 1024];
        const float o = out[p];
        const float d8 = dout[p] * (1.f - o * o);
        dz8[p] = d8;
        auto masked_store = [&](int layer, float* gz) {
            for (int r = 0; r < 256; ++r) {
                gz[r] = acts[((long)layer * 256 + r) * ldn + p] > 0.f ? gz[r] : 0.f;
                dz[((long)layer * 256 + r) * ldn + p] = gz[r];
            }
        };
        for (int r = 0; r < 256; ++r) g[r] = v.w8[r] * d8;
        masked_store(6, g);
        for (int layer = 5; layer >= 0; --layer) {       // dH_layer = W_{layer+1}^T dZ_{layer+1}
            dense_t(Wt[layer + 1], 256, g, gn, false);
            if (dx && layer == 3) dense_t(v.W5i, kin_used, g, dxa, false);   // g is dZ5 here: the skip-connection input gradient
            memcpy(g, gn, sizeof(g));
            masked_store(layer, g);
        }
        if (dx) {
            dense_t(v.W1k, kin_used, g, dxa, true);
            for (int k = 0; k < kin_used; ++k) dx[p * dx_ld + k] = dxa[k];
        }
    }
    if (bias_partials) {
        // tile-major partial rows (header: SG_SDFNET_PARTIAL_ROW): 14 blocks of 256 row sums + the tile's sum of dz8
        const long nblk = sdf_bwd_tiles(N);
        const long nrows = points ? 14 * 256 : 7 * 256;
#pragma omp parallel for schedule(static)
        for (long t = 0; t < nblk; ++t) {
            const long p0 = sdf_bwd_tile_start(N, t), p1 = sdf_bwd_tile_start(N, t + 1);
            float* prow = bias_partials + t * SG_SDFNET_PARTIAL_ROW;
            for (long r = 0; r < nrows; ++r) {
                double s = 0;
                if (r < 7 * 256) {
                    for (long p = p0; p < p1; ++p) s += dz[r * ldn + p];
                } else if (r < 8 * 256) {                                   // w8 gradient: sum_p dz8[p] H7[row][p]
                    const long row = r - 7 * 256;
                    for (long p = p0; p < p1; ++p) s += (double)dz8[p] * acts[(6L * 256 + row) * ldn + p];
                } else {                                                    // point columns of dW1 (dZ1) / dW5 (dZ5)
                    const long q = r - 8 * 256, blk = q / (3 * 256), c = (q / 256) % 3, row = q % 256;
                    const long layer = blk == 0 ? 0 : 4;
                    for (long p = p0; p < p1; ++p) {
                        const long pi = points_period > 0 ? p % points_period : p;
                        s += (double)dz[(layer * 256 + row) * ldn + p] * points[pi * 3 + c];
                    }
                }
                prow[r] = (float)s;
            }
            double s8 = 0;
            for (long p = p0; p < p1; ++p) s8 += dz8[p];
            prow[14 * 256] = (float)s8;
        }
    }
    return SG_OK;
}
// the per-shape latent fold (header: sg_sdfnet_shape_bias)
int sg_sdfnet_shape_bias_cpu(const float* z, long nshapes, int latent, const float* W1, const float* b1, const float* W5,
                             const float* b5, float* zb1, float* zb5, void*) {
    CPU_CHECK(z && W1 && b1 && W5 && b5 && zb1 && zb5 && nshapes > 0 && latent > 0);
    const long ld1 = 3 + latent, ld5 = 259 + latent;
#pragma omp parallel for schedule(static)
    for (long e = 0; e < nshapes * 256; ++e) {
        const long s = e / 256, o = e % 256;
        double a = b1[o], b = b5[o];
        for (int k = 0; k < latent; ++k) {
            a += (double)z[s * latent + k] * W1[o * ld1 + 3 + k];
            b += (double)z[s * latent + k] * W5[o * ld5 + 259 + k];
        }
        zb1[e] = (float)a;
        zb5[e] = (float)b;
    }
    return SG_OK;
}
// backward of the per-shape latent fold (header: sg_sdfnet_shape_bias_bwd)
int sg_sdfnet_pack_shape_bias_cpu(const float* const* params, int latent, float* packed, const float* z, long nshapes, float* zb1,
                                  float* zb5, void* st) {
    CPU_CHECK(params && packed && latent > 0 && z && zb1 && zb5 && nshapes > 0);
    const int rc = sg_sdfnet_pack_cpu(params, latent, 3, packed, st);
    return rc ? rc : sg_sdfnet_shape_bias_cpu(z, nshapes, latent, params[0], params[1], params[8], params[9], zb1, zb5, st);
}
int sg_sdfnet_shape_bias_bwd_cpu(const float* t1, const float* t5, long nshapes, const float* z, int latent, const float* W1,
                                 const float* W5, float* dW1, float* dW5, float* gz, const float* reg_weight, float reg_scale, void*) {
    CPU_CHECK(t1 && t5 && z && W1 && W5 && nshapes > 0 && latent > 0 && (dW1 == nullptr) == (dW5 == nullptr));
    const long ld1 = 3 + latent, ld5 = 259 + latent;
    if (dW1) {
#pragma omp parallel for schedule(static)
        for (int o = 0; o < 256; ++o)
            for (int k = 0; k < latent; ++k) {
                double a = 0, b = 0;
                for (long s = 0; s < nshapes; ++s) {
                    a += (double)t1[o * nshapes + s] * z[s * latent + k];
                    b += (double)t5[o * nshapes + s] * z[s * latent + k];
                }
                dW1[o * ld1 + 3 + k] = (float)a;
                dW5[o * ld5 + 259 + k] = (float)b;
            }
    }
    if (gz) {
#pragma omp parallel for schedule(static)
        for (long s = 0; s < nshapes; ++s)
            for (int k = 0; k < latent; ++k) {
                double a = 0;
                for (int o = 0; o < 256; ++o)
                    a += (double)t1[o * nshapes + s] * W1[o * ld1 + 3 + k] + (double)t5[o * nshapes + s] * W5[o * ld5 + 259 + k];
                float v = (float)a;
                if (reg_scale != 0.f) v += ((reg_weight ? reg_weight[s] : 1.f) * reg_scale) * z[s * latent + k];
                gz[s * latent + k] = v;
            }
    }
    return SG_OK;
}
// the sums derived from one backward's partials (header: sg_sdfnet_bwd_finish): column sums over the tiles, segment sums
// straight from the images
int sg_sdfnet_bwd_finish_cpu(const float* dz, const float* bias_partials, long ldn, long N, int extended, float* const* bias_grads,
                             float* w8_grad, float* b8_grad, float* w1_cols, long w1_ld, float* w5_cols, long w5_ld,
                             const int64_t* seg_off, long nseg, float* t1, float* t5, void*, size_t, unsigned*, void*) {
    CPU_CHECK(dz && bias_partials && N > 0 && ldn >= N && nseg >= 0 && (!bias_grads || b8_grad));
    CPU_CHECK(!extended || !bias_grads || (w8_grad && w1_cols && w5_cols));
    CPU_CHECK(nseg == 0 || (seg_off && t1 && t5));
    const long nblk = sdf_bwd_tiles(N);
    if (bias_grads) {
        const int ngroups = extended ? 14 : 7;
#pragma omp parallel for schedule(static)
        for (long e = 0; e < (long)ngroups * 256; ++e) {
            const long g = e / 256, row = e % 256;
            double s = 0;
            for (long t = 0; t < nblk; ++t) s += bias_partials[t * SG_SDFNET_PARTIAL_ROW + e];
            if (g < 7)
                bias_grads[g][row] = (float)s;
            else if (g == 7)
                w8_grad[row] = (float)s;
            else if (g < 11)
                w1_cols[row * w1_ld + (g - 8)] = (float)s;
            else
                w5_cols[row * w5_ld + (g - 11)] = (float)s;
        }
        double s8 = 0;
        for (long t = 0; t < nblk; ++t) s8 += bias_partials[t * SG_SDFNET_PARTIAL_ROW + 14 * 256];
        b8_grad[0] = (float)s8;
    }
#pragma omp parallel for schedule(static)
    for (long pair = 0; pair < 256 * nseg; ++pair) {
        const long row = pair / nseg, sgm = pair % nseg;
        double a = 0, b = 0;
        for (long e = seg_off[sgm]; e < seg_off[sgm + 1]; ++e) {
            a += dz[row * ldn + e];
            b += dz[(4L * 256 + row) * ldn + e];
        }
        t1[pair] = (float)a;
        t5[pair] = (float)b;
    }
    return SG_OK;
}

// ---- K7b: the LayerNorm form (SDFGenerator, model/point_sdf_net.py:49-119; header: sg_sdfgen_*) ------------------------------------
// The LayerNorm vectors sit where sg_sdfgen_packed_norm_offset (host code of the HIP library, shared) says: float offsets of the
// HIP image's layout for kin_used = 3, far behind this twin's own 462 337 floats.
static const long kGenG = 809216, kGenBe = 811008;
int sg_sdfgen_pack_cpu(const float* const* params, const float* const* norm_params, float* packed, void* st) {
    CPU_CHECK(params && norm_params && packed);
    const int rc = sg_sdfnet_pack_cpu(params, 0, 3, packed, st);
    if (rc) return rc;
    for (int l = 0; l < 7; ++l) {
        memcpy(packed + kGenG + l * 256, norm_params[2 * l], sizeof(float) * 256);
        memcpy(packed + kGenBe + l * 256, norm_params[2 * l + 1], sizeof(float) * 256);
    }
    return SG_OK;
}
// x (256 pre-LayerNorm values) -> xhat, rstd; h = relu(gamma xhat + beta)
static inline float ln_relu(const float* x, const float* g, const float* b, float eps, float* xhat, float* h) {
    double m = 0;
    for (int o = 0; o < 256; ++o) m += x[o];
    m /= 256;
    double v = 0;
    for (int o = 0; o < 256; ++o) v += ((double)x[o] - m) * ((double)x[o] - m);
    const float rstd = (float)(1.0 / sqrt(v / 256 + (double)eps));
    for (int o = 0; o < 256; ++o) {
        xhat[o] = (float)((double)x[o] - m) * rstd;
        const float y = g[o] * xhat[o] + b[o];
        h[o] = y > 0.f ? y : 0.f;
    }
    return rstd;
}
int sg_sdfgen_fwd_cpu(const float* points, const float* packed, const float* zb1, const float* zb5, long points_per_shape,
                      const int* shape_index, float eps, float* out, float* acts, long ldn, long N, void*) {
    CPU_CHECK(points && packed && zb1 && zb5 && out && N > 0 && eps > 0.f && (shape_index || points_per_shape > 0));
    CPU_CHECK(!acts || ldn >= N);
    const CpuSdf v = sdf_view(packed, 3);
    const float *G = packed + kGenG, *Be = packed + kGenBe;
    float* rstd_img = acts ? acts + 7L * 256 * ldn + 56L * ldn : nullptr;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < N; ++p) {
        float x[256], xh[256], h[256];
        const float* xin = points + p * 3;
        const long shape = shape_index ? shape_index[p] : p / points_per_shape;
        auto norm = [&](int layer) {
            const float r = ln_relu(x, G + layer * 256, Be + layer * 256, eps, xh, h);
            if (acts) {
                for (int o = 0; o < 256; ++o) acts[((long)layer * 256 + o) * ldn + p] = xh[o];
                rstd_img[(long)layer * ldn + p] = r;
            }
        };
        dense(v.W1k, 3, xin, zb1 + shape * 256, x, false, false);
        norm(0);
        dense(v.W2, 256, h, v.b + 256, x, false, false);
        norm(1);
        dense(v.W3, 256, h, v.b + 512, x, false, false);
        norm(2);
        dense(v.W4, 256, h, v.b + 768, x, false, false);
        norm(3);
        dense(v.W5x, 256, h, zb5 + shape * 256, x, false, false);
        dense(v.W5i, 3, xin, nullptr, x, false, true);
        norm(4);
        dense(v.W6, 256, h, v.b + 1280, x, false, false);
        norm(5);
        dense(v.W7, 256, h, v.b + 1536, x, false, false);
        norm(6);
        float s = v.b8[0];
        for (int k = 0; k < 256; ++k) s += v.w8[k] * h[k];
        out[p] = s;
    }
    return SG_OK;
}
int sg_sdfgen_bwd_cpu(const float* dout, const float* acts, float* dz, float* dz8, float* partials, const float* points,
                      const float* packed, long ldn, long N, void*) {
    CPU_CHECK(dout && acts && dz && dz8 && partials && points && packed && N > 0 && ldn >= N);
    const CpuSdf v = sdf_view(packed, 3);
    const float *G = packed + kGenG, *Be = packed + kGenBe;
    const float* Wt[7] = {nullptr, v.W2, v.W3, v.W4, v.W5x, v.W6, v.W7};
    const float* rstd_img = acts + 7L * 256 * ldn + 56L * ldn;
    const long nblk = (N + 31) / 32;
#pragma omp parallel for schedule(static)
    for (long t = 0; t < nblk; ++t) {
        std::vector<double> acc(28 * 256, 0.0);      // the tile's partial row (blocks 0..13, then the 14 LayerNorm blocks)
        double s8 = 0;
        const long p0 = t * 32, p1 = p0 + 32 < N ? p0 + 32 : N;
        for (long p = p0; p < p1; ++p) {
            float g[256], gn[256];
            const float d8 = dout[p];
            dz8[p] = d8;
            s8 += d8;
            // dH of image `layer` in g -> dZ (through ReLU and LayerNorm), stored; the LayerNorm parameter sums
            auto through_norm = [&](int layer) {
                const float *gm = G + layer * 256, *bt = Be + layer * 256;
                const float rs = rstd_img[(long)layer * ldn + p];
                double m1 = 0, m2 = 0;
                float dyh[256];
                for (int r = 0; r < 256; ++r) {
                    const float xh = acts[((long)layer * 256 + r) * ldn + p];
                    const float dy = gm[r] * xh + bt[r] > 0.f ? g[r] : 0.f;
                    acc[(14 + layer) * 256 + r] += (double)dy * xh;
                    acc[(21 + layer) * 256 + r] += dy;
                    dyh[r] = dy * gm[r];
                    m1 += dyh[r];
                    m2 += (double)dyh[r] * xh;
                }
                m1 /= 256;
                m2 /= 256;
                for (int r = 0; r < 256; ++r) {
                    const float xh = acts[((long)layer * 256 + r) * ldn + p];
                    g[r] = (float)(rs * ((double)dyh[r] - m1 - (double)xh * m2));
                    dz[((long)layer * 256 + r) * ldn + p] = g[r];
                    acc[layer * 256 + r] += g[r];
                }
            };
            for (int r = 0; r < 256; ++r) {
                const float xh = acts[(6L * 256 + r) * ldn + p];
                const float y = G[6 * 256 + r] * xh + Be[6 * 256 + r];
                acc[7 * 256 + r] += (double)d8 * (y > 0.f ? y : 0.f);      // w8 gradient: sum_p dz8 H7
                g[r] = v.w8[r] * d8;
            }
            through_norm(6);
            for (int layer = 5; layer >= 0; --layer) {
                dense_t(Wt[layer + 1], 256, g, gn, false);
                if (layer == 3)
                    for (int c = 0; c < 3; ++c)
                        for (int r = 0; r < 256; ++r) acc[(11 + c) * 256 + r] += (double)g[r] * points[p * 3 + c];   // g = dZ5
                memcpy(g, gn, sizeof(g));
                through_norm(layer);
            }
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 256; ++r) acc[(8 + c) * 256 + r] += (double)g[r] * points[p * 3 + c];           // g = dZ1
        }
        float* prow = partials + t * SG_SDFGEN_PARTIAL_ROW;
        for (int e = 0; e < 14 * 256; ++e) prow[e] = (float)acc[e];
        prow[14 * 256] = (float)s8;
        for (int e = 0; e < 14 * 256; ++e) prow[SG_SDFNET_PARTIAL_ROW + e] = (float)acc[14 * 256 + e];
    }
    return SG_OK;
}
int sg_sdfgen_bwd_finish_cpu(const float* dz, const float* partials, long ldn, long N, float* const* bias_grads, float* w8_grad,
                             float* b8_grad, float* w1_cols, long w1_ld, float* w5_cols, long w5_ld, float* const* norm_grads,
                             const int64_t* seg_off, long nseg, float* t1, float* t5, void*, size_t, unsigned*, void*) {
    CPU_CHECK(dz && partials && N > 0 && ldn >= N && nseg >= 0 && bias_grads && norm_grads && w8_grad && b8_grad && w1_cols && w5_cols);
    CPU_CHECK(nseg == 0 || (seg_off && t1 && t5));
    const long nblk = (N + 31) / 32;
#pragma omp parallel for schedule(static)
    for (long e = 0; e < 28L * 256; ++e) {
        const long g = e / 256, row = e % 256;
        const long src = g < 14 ? e : SG_SDFNET_PARTIAL_ROW + (e - 14 * 256);
        double s = 0;
        for (long t = 0; t < nblk; ++t) s += partials[t * SG_SDFGEN_PARTIAL_ROW + src];
        if (g < 7)
            bias_grads[g][row] = (float)s;
        else if (g == 7)
            w8_grad[row] = (float)s;
        else if (g < 11)
            w1_cols[row * w1_ld + (g - 8)] = (float)s;
        else if (g < 14)
            w5_cols[row * w5_ld + (g - 11)] = (float)s;
        else
            norm_grads[g - 14][row] = (float)s;
    }
    double s8 = 0;
    for (long t = 0; t < nblk; ++t) s8 += partials[t * SG_SDFGEN_PARTIAL_ROW + 14 * 256];
    b8_grad[0] = (float)s8;
#pragma omp parallel for schedule(static)
    for (long pair = 0; pair < 256 * nseg; ++pair) {
        const long row = pair / nseg, sgm = pair % nseg;
        double a = 0, b = 0;
        for (long e = seg_off[sgm]; e < seg_off[sgm + 1]; ++e) {
            a += dz[row * ldn + e];
            b += dz[(4L * 256 + row) * ldn + e];
        }
        t1[pair] = (float)a;
        t5[pair] = (float)b;
    }
    return SG_OK;
}

// ---- PointNet.nn1 + max as a selection pass (header: sg_pointnet_pack / sg_pointnet_select) -----------------------------------------
// packed buffer of this twin: the four weight matrices row-major, one after the other
int sg_pointnet_pack_cpu(const float* const* weights, float* packed, void*) {
    CPU_CHECK(weights && packed && weights[0] && weights[1] && weights[2] && weights[3]);
    const long n[4] = {64 * 4, 128 * 64, 256 * 128, 512 * 256};
    float* p = packed;
    for (int i = 0; i < 4; ++i, p += n[i - 1]) memcpy(p, weights[i], sizeof(float) * n[i]);
    return SG_OK;
}
int sg_pointnet_select_cpu(const float* x, const float* packed, const float* const* biases, long B, long P, float* out, int* idx,
                           void*, size_t, void*) {
    CPU_CHECK(x && packed && biases && out && idx && B > 0 && P > 0 && P % 32 == 0);
    const float *W1 = packed, *W2 = W1 + 256, *W3 = W2 + 8192, *W4 = W3 + 32768;
#pragma omp parallel for schedule(static)
    for (long b = 0; b < B; ++b) {
        std::vector<float> best(512, -INFINITY);
        std::vector<int> bp(512, 0);
        for (long p = 0; p < P; ++p) {
            const float* in = x + (b * P + p) * 4;
            float h1[64], h2[128], h3[256];
            for (int o = 0; o < 64; ++o) {
                float s = biases[0][o];
                for (int k = 0; k < 4; ++k) s += W1[o * 4 + k] * in[k];
                h1[o] = s > 0.f ? s : 0.f;
            }
            for (int o = 0; o < 128; ++o) {
                float s = biases[1][o];
                for (int k = 0; k < 64; ++k) s += W2[o * 64 + k] * h1[k];
                h2[o] = s > 0.f ? s : 0.f;
            }
            for (int o = 0; o < 256; ++o) {
                float s = biases[2][o];
                for (int k = 0; k < 128; ++k) s += W3[o * 128 + k] * h2[k];
                h3[o] = s > 0.f ? s : 0.f;
            }
            for (int o = 0; o < 512; ++o) {
                float s = biases[3][o];
                for (int k = 0; k < 256; ++k) s += W4[o * 256 + k] * h3[k];
                if (s > best[o]) best[o] = s, bp[o] = (int)p;
            }
        }
        for (int o = 0; o < 512; ++o) out[b * 512 + o] = best[o], idx[b * 512 + o] = bp[o];
    }
    return SG_OK;
}

// ---- K8 - K11: elementwise, reductions, table rows, optimizers ---------------------------------------------------------------
int sg_axpby_cpu(const float* x, const float* y, float* out, long n, float a, float b, void*) {
    CPU_CHECK(x && out && n > 0);
    for (long e = 0; e < n; ++e) out[e] = y ? a * x[e] + b * y[e] : a * x[e];
    return SG_OK;
}
int sg_reduce_sum_cpu(const float* x, float* out, long n, float scale, void*, size_t, void*) {
    CPU_CHECK(x && out && n > 0);
    double s = 0;
    for (long e = 0; e < n; ++e) s += x[e];
    out[0] = (float)(s * (double)scale);
    return SG_OK;
}
int sg_gather_rows_cpu(const float* table, const int64_t* idx, float* out, long n, int L, void*) {
    CPU_CHECK(table && idx && out && n > 0 && L > 0);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) memcpy(out + i * L, table + idx[i] * L, sizeof(float) * L);
    return SG_OK;
}
int sg_scatter_add_rows_cpu(const float* rows, long rows_ld, const int64_t* idx, float* table_grad, long n, int L, void*) {
    CPU_CHECK(rows && idx && table_grad && n > 0 && L > 0 && rows_ld >= L);
    for (long i = 0; i < n; ++i)
        for (int k = 0; k < L; ++k) table_grad[idx[i] * L + k] += rows[i * rows_ld + k];
    return SG_OK;
}
// stable counting sort of the batch on the shape id (header: sg_sdf_batch_sort)
int sg_sdf_batch_sort_cpu(const int64_t* indices, long n, long pointcloud_size, long nshapes, const float* points, const float* sdf,
                          float* out_points, float* out_sdf, int* out_shape, int64_t* seg_off, float* counts, int* bad_index_flag,
                          int* bad_index_device, int bad_index_value, void*, size_t, void*) {
    CPU_CHECK(indices && points && sdf && out_points && out_sdf && out_shape && seg_off && counts && bad_index_flag);
    CPU_CHECK(n > 0 && pointcloud_size > 0 && nshapes > 0 && bad_index_value != 0);
    std::vector<int64_t> next(nshapes + 1, 0);
    auto shape_of = [&](int64_t i) {
        int64_t k = i >= 0 ? i / pointcloud_size : -1;
        if (k < 0 || k >= nshapes) {
            if (!bad_index_device) {
                *bad_index_flag = bad_index_value;
            } else if (*bad_index_device == 0 || *bad_index_device == bad_index_value) {
                *bad_index_device = bad_index_value;
                *bad_index_flag = bad_index_value;
            }
            k = k < 0 ? 0 : nshapes - 1;
        }
        return k;
    };
    for (long e = 0; e < n; ++e) next[shape_of(indices[e]) + 1] += 1;
    for (long s = 0; s < nshapes; ++s) {
        counts[s] = (float)next[s + 1];
        next[s + 1] += next[s];
    }
    for (long s = 0; s <= nshapes; ++s) seg_off[s] = next[s];
    for (long e = 0; e < n; ++e) {
        const int64_t k = shape_of(indices[e]), p = next[k]++, rows = nshapes * pointcloud_size;
        const int64_t i = indices[e] < 0 ? 0 : (indices[e] < rows ? indices[e] : rows - 1);   // (flagged above)
        memcpy(out_points + p * 3, points + i * 3, 3 * sizeof(float));
        out_sdf[p] = sdf[i];
        out_shape[p] = (int)k;
    }
    return SG_OK;
}
int sg_rmsprop_step_cpu(float* p, const float* g, float* sq, long n, float lr, float alpha, float eps, float gscale, float clip, void*) {
    CPU_CHECK(p && g && sq && n > 0);
    for (long e = 0; e < n; ++e) {
        const float gg = g[e] * gscale;
        const float s = alpha * sq[e] + (1.f - alpha) * gg * gg;
        sq[e] = s;
        float v = p[e] - lr * (gg / (sqrtf(s) + eps));
        if (clip > 0.f) v = std::min(std::max(v, -clip), clip);
        p[e] = v;
    }
    return SG_OK;
}
static void adam_update(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float bc1,
                        float bc2_sqrt, float gscale) {
    const float step = lr / bc1;
    for (long e = 0; e < n; ++e) {
        const float gg = g[e] * gscale;
        const float mm = b1 * m[e] + (1.f - b1) * gg;
        const float vv = b2 * v[e] + (1.f - b2) * gg * gg;
        m[e] = mm;
        v[e] = vv;
        p[e] = p[e] - step * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
}
int sg_adam_step_guarded_cpu(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                             long step, float gscale, const int* skip, void*) {
    CPU_CHECK(p && g && m && v && n > 0 && step > 0);
    if (skip && *skip) return SG_OK;
    adam_update(p, g, m, v, n, lr, b1, b2, eps, (float)(1.0 - pow((double)b1, (double)step)),
                (float)sqrt(1.0 - pow((double)b2, (double)step)), gscale);
    return SG_OK;
}
int sg_adam_step_cpu(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, long step,
                     float gscale, void* stream) {
    return sg_adam_step_guarded_cpu(p, g, m, v, n, lr, b1, b2, eps, step, gscale, nullptr, stream);
}
int sg_adam_step_dev_multi_cpu(int nsets, float* const* p, const float* const* g, float* const* m, float* const* v, const long* n,
                               const float* lr, const float* b1, const float* b2, const float* eps, long long* const* step_dev,
                               float* const* corr_dev, const float* gscale, const int* skip, void* stream);
int sg_adam_step_dev_guarded_cpu(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                                 long long* step_dev, float* corr_dev, float gscale, const int* skip, void*) {
    CPU_CHECK(p && g && m && v && n > 0 && step_dev && corr_dev);
    if (skip && *skip) return SG_OK;
    const long long t = ++*step_dev;
    corr_dev[0] = (float)(1.0 - pow((double)b1, (double)t));
    corr_dev[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
    adam_update(p, g, m, v, n, lr, b1, b2, eps, corr_dev[0], corr_dev[1], gscale);
    return SG_OK;
}
int sg_adam_step_dev_multi_cpu(int nsets, float* const* p, const float* const* g, float* const* m, float* const* v, const long* n,
                               const float* lr, const float* b1, const float* b2, const float* eps, long long* const* step_dev,
                               float* const* corr_dev, const float* gscale, const int* skip, void* stream) {
    CPU_CHECK(nsets > 0 && nsets <= 4 && p && g && m && v && n && lr && b1 && b2 && eps && step_dev && corr_dev && gscale);
    for (int i = 0; i < nsets; ++i) {
        const int rc = sg_adam_step_dev_guarded_cpu(p[i], g[i], m[i], v[i], n[i], lr[i], b1[i], b2[i], eps[i], step_dev[i], corr_dev[i],
                                                    gscale[i], skip, stream);
        if (rc) return rc;
    }
    return SG_OK;
}
int sg_adam_step_dev_cpu(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                         long long* step_dev, float* corr_dev, float gscale, void* stream) {
    return sg_adam_step_dev_guarded_cpu(p, g, m, v, n, lr, b1, b2, eps, step_dev, corr_dev, gscale, nullptr, stream);
}
int sg_clamp_cpu(float* p, long n, float lo, float hi, void*) {
    CPU_CHECK(p && n > 0);
    for (long e = 0; e < n; ++e) p[e] = std::min(std::max(p[e], lo), hi);
    return SG_OK;
}
int sg_clamp_multi_cpu(float* const* tensors, const long* counts, int ntensors, float lo, float hi, void*) {
    CPU_CHECK(tensors && counts && ntensors > 0);
    for (int i = 0; i < ntensors; ++i) {
        const int rc = sg_clamp_cpu(tensors[i], counts[i], lo, hi, nullptr);
        if (rc != SG_OK) return rc;
    }
    return SG_OK;
}
int sg_voxel_prepare_cpu(const float* x, float* out, long n, float c, float divisor, void*) {
    CPU_CHECK(x && out && n > 0 && c >= 0.f);
    for (long e = 0; e < n; ++e) {
        float v = x[e];
        v = v < -c ? -c : (v > c ? c : v);
        out[e] = divisor > 0.f ? v / divisor : v;
    }
    return SG_OK;
}

// ---- losses / blends ----------------------------------------------------------------------------------------------------------
int sg_loss_weighted_l1_fwd_cpu(const float* o, const float* t, long n, float negw, float* loss, void*, size_t, void*) {
    CPU_CHECK(o && t && loss && n > 0);
    double s = 0;
    for (long e = 0; e < n; ++e) s += fabs((double)((o[e] - t[e]) * (t[e] < 0.f ? negw : 1.f)));
    loss[0] = (float)(s / (double)n);
    return SG_OK;
}
int sg_loss_weighted_l1_bwd_cpu(const float* o, const float* t, const float* gloss, float* d_o, long n, float negw, void*) {
    CPU_CHECK(o && t && gloss && d_o && n > 0);
    const float g = gloss[0] * (float)(1.0 / (double)n);
    for (long e = 0; e < n; ++e) {
        const float w = t[e] < 0.f ? negw : 1.f, d = (o[e] - t[e]) * w;
        d_o[e] = g * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
    }
    return SG_OK;
}
int sg_count_sign_mismatch_cpu(const float* a, const float* b, long n, long long* count, void*, size_t, void*) {
    CPU_CHECK(a && b && count && n > 0);
    long long c = 0;
    for (long e = 0; e < n; ++e) {
        volatile float prod = a[e] * b[e];     // the rounded fp32 product, as `(input * target) < 0` sees it
        c += prod < 0.f ? 1 : 0;
    }
    count[0] = c;
    return SG_OK;
}
int sg_loss_mean_split_fwd_cpu(const float* x, long n, long n_first, float w_first, float w_rest, float* loss, float* dx_unit,
                               void*) {
    CPU_CHECK(x && loss && n > 0 && n_first >= 0 && n_first <= n);
    if (dx_unit) {
        const float ca = n_first > 0 ? (float)((double)w_first / (double)n_first) : 0.f;
        const float cb = n > n_first ? (float)((double)w_rest / (double)(n - n_first)) : 0.f;
        for (long e = 0; e < n; ++e) dx_unit[e] = e < n_first ? ca : cb;
    }
    double a = 0, b = 0;
    for (long e = 0; e < n; ++e) (e < n_first ? a : b) += x[e];
    double v = 0;
    if (n_first > 0) v += (double)w_first * a / (double)n_first;
    if (n > n_first) v += (double)w_rest * b / (double)(n - n_first);
    loss[0] = (float)v;
    return SG_OK;
}
int sg_loss_mean_split_bwd_cpu(const float* gloss, float* dx, long n, long n_first, float w_first, float w_rest, void*) {
    CPU_CHECK(gloss && dx && n > 0 && n_first >= 0 && n_first <= n);
    const float ca = n_first > 0 ? (float)((double)w_first / (double)n_first) : 0.f;
    const float cb = n > n_first ? (float)((double)w_rest / (double)(n - n_first)) : 0.f;
    for (long e = 0; e < n; ++e) dx[e] = gloss[0] * (e < n_first ? ca : cb);
    return SG_OK;
}
// ---- classic-GAN losses on the [B] vector of discriminator outputs; VAE reparameterisation; the critic's last layer ---------
int sg_loss_bce_fwd_cpu(const float* p, long n, float t, float* loss, void*) {
    CPU_CHECK(p && loss && n > 0);
    double s = 0;
    for (long e = 0; e < n; ++e)
        s -= (double)(t * std::max(logf(p[e]), -100.f) + (1.f - t) * std::max(logf(1.f - p[e]), -100.f));
    loss[0] = (float)(s / (double)n);
    return SG_OK;
}
int sg_loss_bce_bwd_cpu(const float* p, const float* gloss, float* dp, long n, float t, void*) {
    CPU_CHECK(p && gloss && dp && n > 0);
    const float g = gloss[0] * (float)(1.0 / (double)n);
    for (long e = 0; e < n; ++e) dp[e] = g * (p[e] - t) / std::max((1.f - p[e]) * p[e], 1e-12f);
    return SG_OK;
}
int sg_loss_neg_mean_log_fwd_cpu(const float* p, long n, float* loss, void*) {
    CPU_CHECK(p && loss && n > 0);
    double s = 0;
    for (long e = 0; e < n; ++e) s -= (double)logf(p[e]);
    loss[0] = (float)(s / (double)n);
    return SG_OK;
}
int sg_loss_neg_mean_log_bwd_cpu(const float* p, const float* gloss, float* dp, long n, void*) {
    CPU_CHECK(p && gloss && dp && n > 0);
    const float g = gloss[0] * (float)(1.0 / (double)n);
    for (long e = 0; e < n; ++e) dp[e] = -g / p[e];
    return SG_OK;
}
int sg_vae_reparam_fwd_cpu(const float* mu, const float* lv, const float* eps, float* z, long n, void*) {
    CPU_CHECK(mu && lv && eps && z && n > 0);
    for (long e = 0; e < n; ++e) {
        const float sd = expf(lv[e] * 0.5f);
        volatile float prod = sd * eps[e];     // two roundings, as the reference's `mean + standard_deviation * eps`
        z[e] = mu[e] + prod;
    }
    return SG_OK;
}
int sg_vae_reparam_bwd_cpu(const float* lv, const float* eps, const float* gz, float* dlv, long n, void*) {
    CPU_CHECK(lv && eps && gz && dlv && n > 0);
    for (long e = 0; e < n; ++e) dlv[e] = gz[e] * eps[e] * (0.5f * expf(lv[e] * 0.5f));
    return SG_OK;
}
static inline float head_act_cpu(float v, int act, float slope) {
    return act == ACT_LEAKY ? (v > 0.f ? v : v * slope) : (act == ACT_RELU ? (v > 0.f ? v : 0.f) : v);
}
int sg_head_dot_fwd_cpu(const float* z, const float* w, const float* bias, float* y, int N, long K, int act, float slope, void*) {
    CPU_CHECK(z && w && y && N > 0 && K > 0 && K % 4 == 0);
    CPU_CHECK(act == ACT_NONE || act == ACT_LEAKY || act == ACT_RELU);
#pragma omp parallel for
    for (int n = 0; n < N; ++n) {
        double s = 0;
        for (long k = 0; k < K; ++k) s += (double)head_act_cpu(z[(long)n * K + k], act, slope) * (double)w[k];
        y[n] = (float)s + (bias ? bias[0] : 0.f);
    }
    return SG_OK;
}
int sg_head_dot_bwd_cpu(const float* z, const float* w, const float* gy, float* gz, float* gw, float* gb, float* gbz, void*, int N,
                        int C, int S, int act, float slope, void*) {
    CPU_CHECK(z && w && gy && gz && N > 0 && C > 0 && S == 64);
    CPU_CHECK(act == ACT_NONE || act == ACT_LEAKY || act == ACT_RELU);
    const long K = (long)C * S;
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        double cz = 0;
        for (int s = 0; s < S; ++s) {
            const long k = (long)c * S + s;
            double aw = 0;
            for (int n = 0; n < N; ++n) {
                const float v = z[(long)n * K + k];
                const float d = act == ACT_LEAKY ? (v > 0.f ? 1.f : slope) : (act == ACT_RELU ? (v > 0.f ? 1.f : 0.f) : 1.f);
                const float o = gy[n] * w[k] * d;
                gz[(long)n * K + k] = o;
                cz += (double)o;
                aw += (double)gy[n] * (double)head_act_cpu(v, act, slope);
            }
            if (gw) gw[k] = (float)aw;
        }
        if (gbz) gbz[c] = (float)cz;
    }
    if (gb) {
        double t = 0;
        for (int n = 0; n < N; ++n) t += (double)gy[n];
        gb[0] = (float)t;
    }
    return SG_OK;
}
int sg_loss_kld_fwd_cpu(const float* mu, const float* lv, long n, float* loss, void*, size_t, void*) {
    CPU_CHECK(mu && lv && loss && n > 0);
    double s = 0;
    for (long e = 0; e < n; ++e) s += (double)(1.f + lv[e] - mu[e] * mu[e] - expf(lv[e]));
    loss[0] = (float)(-0.5 * s / (double)n);
    return SG_OK;
}
int sg_loss_kld_bwd_cpu(const float* mu, const float* lv, const float* gloss, float* dmu, float* dlv, long n, void*) {
    CPU_CHECK(mu && lv && gloss && dmu && dlv && n > 0);
    const float g = gloss[0] * (float)(1.0 / (double)n);
    for (long e = 0; e < n; ++e) {
        dmu[e] = g * mu[e];
        dlv[e] = -0.5f * g * (1.f - expf(lv[e]));
    }
    return SG_OK;
}
int sg_loss_meansq_fwd_cpu(const float* x, const float* roww, long rows, int L, double denom, float* loss, void*, size_t, void*) {
    CPU_CHECK(x && loss && rows > 0 && L > 0 && denom > 0);
    double s = 0;
    for (long e = 0; e < rows * L; ++e) s += (double)((roww ? roww[e / L] : 1.f) * x[e] * x[e]);
    loss[0] = (float)(s / denom);
    return SG_OK;
}
int sg_loss_meansq_bwd_cpu(const float* x, const float* roww, const float* gloss, float* dx, long rows, int L, double denom, void*) {
    CPU_CHECK(x && gloss && dx && rows > 0 && L > 0 && denom > 0);
    const float g = gloss[0] * (float)(2.0 / denom);
    for (long e = 0; e < rows * L; ++e) dx[e] = (roww ? roww[e / L] : 1.f) * g * x[e];
    return SG_OK;
}
int sg_loss_deepsdf_fwd_cpu(const float* o, const float* t, long n, const float* z, const float* roww, long rows, int L, double denom,
                            float* loss, void*, size_t, void*) {
    CPU_CHECK(o && t && z && loss && n > 0 && rows > 0 && L > 0 && denom > 0);
    double s1 = 0, s2 = 0;
    for (long e = 0; e < n; ++e) s1 += fabs((double)(o[e] - t[e]));
    for (long e = 0; e < rows * L; ++e) s2 += (double)((roww ? roww[e / L] : 1.f) * z[e] * z[e]);
    loss[0] = (float)(s1 / (double)n) + (float)(s2 / denom);
    return SG_OK;
}
int sg_loss_deepsdf_bwd_cpu(const float* o, const float* t, long n, const float* z, const float* roww, long rows, int L, double denom,
                            const float* gloss, float* d_o, float* dz, void*) {
    CPU_CHECK(o && t && z && gloss && d_o && dz && n > 0 && rows > 0 && L > 0 && denom > 0);
    const float g1 = gloss[0] * (float)(1.0 / (double)n), g2 = gloss[0] * (float)(2.0 / denom);
    for (long e = 0; e < n; ++e) {
        const float d = o[e] - t[e];
        d_o[e] = g1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    for (long e = 0; e < rows * L; ++e) dz[e] = (roww ? roww[e / L] : 1.f) * g2 * z[e];
    return SG_OK;
}
int sg_loss_deepsdf_fused_cpu(const float* o, const float* t, long n, const float* z, const float* roww, long rows, int L, double denom,
                              float* loss, float* d_o, float* dz, void*, size_t, unsigned*, void*) {
    CPU_CHECK(o && t && z && loss && n > 0 && rows > 0 && L > 0 && denom > 0);
    const int rc = sg_loss_deepsdf_fwd_cpu(o, t, n, z, roww, rows, L, denom, loss, nullptr, 0, nullptr);
    if (rc != SG_OK) return rc;
    const float g1 = (float)(1.0 / (double)n), g2 = (float)(2.0 / denom);
    if (d_o)
        for (long e = 0; e < n; ++e) {
            const float d = o[e] - t[e];
            d_o[e] = g1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
    if (dz)
        for (long e = 0; e < rows * L; ++e) dz[e] = (roww ? roww[e / L] : 1.f) * g2 * z[e];
    return SG_OK;
}
int sg_gradient_penalty_fwd_cpu(const float* grad, long B, long M, float weight, float* norms, float* loss, void*) {
    CPU_CHECK(grad && norms && loss && B > 0 && M > 0);
    double acc = 0;
    for (long b = 0; b < B; ++b) {
        double s = 0;
        for (long e = 0; e < M; ++e) s += (double)grad[b * M + e] * grad[b * M + e];
        norms[b] = (float)sqrt(s);
        acc += ((double)norms[b] - 1.0) * ((double)norms[b] - 1.0);
    }
    loss[0] = (float)(acc / (double)B * (double)weight);
    return SG_OK;
}
int sg_gradient_penalty_bwd_cpu(const float* grad, const float* norms, const float* gloss, float* dgrad, long B, long M, float weight, void*) {
    CPU_CHECK(grad && norms && gloss && dgrad && B > 0 && M > 0);
    const float coef = (float)(2.0 * (double)weight / (double)B);
    for (long b = 0; b < B; ++b) {
        const float c = norms[b] > 0.f ? gloss[0] * coef * (norms[b] - 1.f) / norms[b] : 0.f;
        for (long e = 0; e < M; ++e) dgrad[b * M + e] = c * grad[b * M + e];
    }
    return SG_OK;
}
int sg_lerp_rows_cpu(const float* a, const float* b, const float* alpha, float* out, long B, long M, void*) {
    CPU_CHECK(a && b && alpha && out && B > 0 && M > 0);
    for (long r = 0; r < B; ++r) {
        const float al = alpha[r], be = 1.f - al;
        for (long e = 0; e < M; ++e) {
            const float t1 = al * a[r * M + e], t2 = be * b[r * M + e];
            out[r * M + e] = t1 + t2;
        }
    }
    return SG_OK;
}
int sg_fade_blend_cpu(const float* x, const float* half, float* out, long B, int C, long S, float fade, float hs, void*) {
    CPU_CHECK(half && out && B > 0 && C > 0 && S > 0);
    for (long b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (long e = 0; e < S; ++e) {
                float v = x ? fade * x[(b * C + c) * S + e] : 0.f;
                if (c == 0) v = v + hs * half[b * S + e];
                out[(b * C + c) * S + e] = v;
            }
    return SG_OK;
}
int sg_channel0_cpu(const float* g, float* out, long B, int C, long S, float scale, void*) {
    CPU_CHECK(g && out && B > 0 && C > 0 && S > 0);
    for (long b = 0; b < B; ++b)
        for (long e = 0; e < S; ++e) out[b * S + e] = scale * g[b * C * S + e];
    return SG_OK;
}
int sg_subsample2_cpu(const float* x, float* out, long B, int R, void*) {
    CPU_CHECK(x && out && B > 0 && R >= 2 && !(R & 1));
    const int h = R / 2;
    for (long b = 0; b < B; ++b)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < h; ++j)
                for (int k = 0; k < h; ++k) out[((b * h + i) * h + j) * h + k] = x[((b * R + 2 * i) * R + 2 * j) * R + 2 * k];
    return SG_OK;
}
int sg_subsample2_adjoint_cpu(const float* g, float* out, long B, int R, void*) {
    CPU_CHECK(g && out && B > 0 && R >= 2 && !(R & 1));
    const int h = R / 2;
    memset(out, 0, sizeof(float) * B * R * R * R);
    for (long b = 0; b < B; ++b)
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < h; ++j)
                for (int k = 0; k < h; ++k) out[((b * R + 2 * i) * R + 2 * j) * R + 2 * k] = g[((b * h + i) * h + j) * h + k];
    return SG_OK;
}

// ---- PointNet family ------------------------------------------------------------------------------------------------------------
int sg_layernorm_fwd_cpu(const float* x, long ldx, const float* rowbias, long rows_per_shape, const float* gamma,
                         const float* beta, float* y, long ldy, float* mean, float* rstd, long R, int C, float eps, int act, void*) {
    CPU_CHECK(x && gamma && beta && y && mean && rstd && R > 0 && C > 0 && (act == ACT_NONE || act == ACT_RELU));
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        const float* zb = rowbias ? rowbias + (r / rows_per_shape) * C : nullptr;
        double s = 0, s2 = 0;
        for (int c = 0; c < C; ++c) s += x[r * ldx + c] + (zb ? zb[c] : 0.f);
        const double mu = s / C;
        for (int c = 0; c < C; ++c) {
            const double d = x[r * ldx + c] + (zb ? zb[c] : 0.f) - mu;
            s2 += d * d;
        }
        const float rs = (float)(1.0 / sqrt(s2 / C + (double)eps));
        mean[r] = (float)mu;
        rstd[r] = rs;
        for (int c = 0; c < C; ++c) {
            const float v = gamma[c] * ((x[r * ldx + c] + (zb ? zb[c] : 0.f) - (float)mu) * rs) + beta[c];
            y[r * ldy + c] = act == ACT_RELU ? (v > 0.f ? v : 0.f) : v;
        }
    }
    return SG_OK;
}
int sg_layernorm_bwd_cpu(const float* x, long ldx, const float* rowbias, long rows_per_shape, const float* gamma, const float* y,
                         long ldy, const float* dy, long lddy, const float* mean, const float* rstd, float* dz, long lddz,
                         float* dgamma, float* dbeta, long R, int C, int act, void*, size_t, void*) {
    CPU_CHECK(x && gamma && dy && mean && rstd && dz && dgamma && dbeta && R > 0 && C > 0);
    std::vector<double> ag(C, 0.0), ab(C, 0.0);
    for (long r = 0; r < R; ++r) {
        const float* zb = rowbias ? rowbias + (r / rows_per_shape) * C : nullptr;
        const float mu = mean[r], rs = rstd[r];
        double s1 = 0, s2 = 0;
        std::vector<float> xh(C), gh(C);
        for (int c = 0; c < C; ++c) {
            float g = dy[r * lddy + c];
            if (act == ACT_RELU) g = y[r * ldy + c] > 0.f ? g : 0.f;
            xh[c] = (x[r * ldx + c] + (zb ? zb[c] : 0.f) - mu) * rs;
            gh[c] = g * gamma[c];
            ag[c] += (double)g * xh[c];
            ab[c] += g;
            s1 += gh[c];
            s2 += (double)gh[c] * xh[c];
        }
        const float m1 = (float)(s1 / C), m2 = (float)(s2 / C);
        for (int c = 0; c < C; ++c) dz[r * lddz + c] = rs * (gh[c] - m1 - xh[c] * m2);
    }
    for (int c = 0; c < C; ++c) {
        dgamma[c] = (float)ag[c];
        dbeta[c] = (float)ab[c];
    }
    return SG_OK;
}
int sg_segmax_fwd_cpu(const float* x, float* out, int* idx, long B, long P, int C, void*, size_t, void*) {
    CPU_CHECK(x && out && idx && B > 0 && P > 0 && C > 0);
    // torch.max semantics: first occurrence of the maximum; a NaN wins and the first NaN is reported
    for (long b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float best = x[(b * P) * C + c];
            int arg = 0;
            for (long p = 1; p < P && best == best; ++p) {
                const float v = x[(b * P + p) * C + c];
                if (v != v || v > best) {
                    best = v;
                    arg = (int)p;
                }
            }
            out[b * C + c] = best;
            idx[b * C + c] = arg;
        }
    return SG_OK;
}
int sg_segmax_scatter_cpu(const float* dy, const int* idx, float* dx, long B, long P, int C, void*) {
    CPU_CHECK(dy && idx && dx && B > 0 && P > 0 && C > 0);
    memset(dx, 0, sizeof(float) * B * P * C);
    for (long b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) dx[(b * P + idx[b * C + c]) * C + c] = dy[b * C + c];
    return SG_OK;
}
int sg_scatter_rows_grouped_cpu(const float* g, const int64_t* rows, float* dx, long B, int C, int K, void*) {
    CPU_CHECK(g && rows && dx && B > 0 && C > 0 && C <= 1024 && K > 0);
    for (long b = 0; b < B; ++b)
        for (int c2 = 0; c2 < C; ++c2) {
            const int64_t mine = rows[b * C + c2];
            bool led = false;
            for (int e = 0; e < c2 && !led; ++e) led = rows[b * C + e] == mine;
            if (led) continue;
            for (int k = 0; k < K; ++k) {
                float s = g[(b * C + c2) * K + k];
                for (int e = c2 + 1; e < C; ++e)
                    if (rows[b * C + e] == mine) s += g[(b * C + e) * K + k];
                dx[mine * K + k] = s;
            }
        }
    return SG_OK;
}
int sg_rowdot_cpu(const float* h, const float* w, const float* bias, float* out, long B, int C, int K, void*) {
    CPU_CHECK(h && w && out && B > 0 && C > 0 && K > 0);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < B * C; ++r) {
        const int c = (int)(r % C);
        double s = bias ? bias[c] : 0.0;
        for (int k = 0; k < K; ++k) s += (double)h[r * K + k] * w[(long)c * K + k];
        out[r] = (float)s;
    }
    return SG_OK;
}
int sg_rowscale_cpu(const float* g, const float* w, float* out, long B, int C, int K, void*) {
    CPU_CHECK(g && w && out && B > 0 && C > 0 && K > 0);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < B * C; ++r)
        for (int k = 0; k < K; ++k) out[r * K + k] = g[r] * w[(long)(r % C) * K + k];
    return SG_OK;
}
int sg_rowouter_cpu(const float* g, const float* h, float* out, long B, int C, int K, void*) {
    CPU_CHECK(g && h && out && B > 0 && C > 0 && K > 0);
#pragma omp parallel for schedule(static)
    for (long e = 0; e < (long)C * K; ++e) {
        const long c = e / K, k = e % K;
        double s = 0;
        for (long b = 0; b < B; ++b) s += (double)g[b * C + c] * h[(b * C + c) * K + k];
        out[e] = (float)s;
    }
    return SG_OK;
}
int sg_segmax_gather_cpu(const float* x, const int* idx, float* out, long B, long P, int C, void*) {
    CPU_CHECK(x && idx && out && B > 0 && P > 0 && C > 0);
    for (long b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) out[b * C + c] = x[(b * P + idx[b * C + c]) * C + c];
    return SG_OK;
}
int sg_scatter_max_fwd_cpu(const float* x, const int64_t* batch, float* out, int* arg, long N, long B, int C, void*, size_t, void*) {
    CPU_CHECK(x && batch && out && arg && N > 0 && B > 0 && C > 0);
    for (long e = 0; e < B * C; ++e) {
        arg[e] = -1;
        out[e] = 0.f;
    }
    for (long i = 0; i < N; ++i) {
        const long b = batch[i];
        if (b < 0 || b >= B) continue;
        for (int c = 0; c < C; ++c)
            if (arg[b * C + c] < 0 || x[i * C + c] > out[b * C + c]) {
                out[b * C + c] = x[i * C + c];
                arg[b * C + c] = (int)i;
            }
    }
    return SG_OK;
}
int sg_scatter_max_scatter_cpu(const float* dy, const int* arg, float* dx, long N, long B, int C, void*) {
    CPU_CHECK(dy && arg && dx && N > 0 && B > 0 && C > 0);
    memset(dx, 0, sizeof(float) * N * C);
    for (long e = 0; e < B * C; ++e)
        if (arg[e] >= 0) dx[(long)arg[e] * C + e % C] = dy[e];
    return SG_OK;
}
int sg_scatter_max_gather_cpu(const float* x, const int* arg, float* out, long N, long B, int C, void*) {
    CPU_CHECK(x && arg && out && N > 0 && B > 0 && C > 0);
    for (long e = 0; e < B * C; ++e) out[e] = arg[e] >= 0 ? x[(long)arg[e] * C + e % C] : 0.f;
    return SG_OK;
}

}  // extern "C"
