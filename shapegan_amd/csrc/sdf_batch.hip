// shapegan_amd/csrc/sdf_batch.hip — batch assembly of the DeepSDF auto-decoder step (K10), one counting sort.
//
// Replaces, for the shape-sorted data flow of SDFAutoDecoderTrainer.step_sorted, the index arithmetic of
// train_sdf_autodecoder.py:78-85
//     model_indices = indices // POINTCLOUD_SIZE ; batch_points = points[indices, :] ; batch_sdf = sdf[indices]
// plus the grouping by shape that lets the fused MLP read latent_codes[model_indices] as per-shape bias rows: round 1 did
// this with torch.argsort + bincount + cumsum + three gathers (27 launches and three host synchronisations, 0.54 ms of a
// 5.3 ms step).  Here: keys + per-chunk histograms, a prefix over chunks, a stable scatter that gathers the table rows
// straight to their sorted position — three launches, no host round trip, and a fully deterministic order (points of a
// shape keep the order they have in `indices`).
//
// A chunk is 512 batch entries handled by ONE wave: the position of an entry inside its shape's run is
//     seg_base[shape] (scan of the shape totals) + base[chunk][shape] (entries of that shape in earlier chunks)
//   + its rank among the equal keys of the chunk (64 at a time: a readlane loop, order = lane order).
#include "common.h"
#include "../../include/shapegan_hip.h"

namespace sg {

// Chunk = ROUNDS x 64 entries.  Eight rounds keep the per-chunk tables (hist / base: [chunks][S] ints) small for large batches and
// many shapes; the reference's own batch (20 000 entries of 64 shapes) takes ONE round per wave — 313 independent waves instead of
// 40 chains of eight dependent rounds (the scatter was 26 us of that step's 37 us sort) — when the tables then stay below 128 KB:
// the chunk prefix is one workgroup per 64 shapes walking the chunks (3 125 one-round chunks of a 200 000-entry batch took it 82 us).
constexpr int kSortMaxShapes = 16384;   // one int of LDS per shape
constexpr long kSortSmallTableInts = 1L << 15;

template <int kSortRounds>
__global__ void __launch_bounds__(64) sdf_sort_hist_kernel(const int64_t* __restrict__ idx, long n, long pc, int S,
                                                           int* __restrict__ keys, int* __restrict__ hist,
                                                           int* __restrict__ flag, int* __restrict__ flag_dev, int flag_value) {
    constexpr int kSortChunk = kSortRounds * 64;
    extern __shared__ int cnt[];
    const int lane = threadIdx.x;
    const long chunk = blockIdx.x;
    for (int s = lane; s < S; s += 64) cnt[s] = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kSortRounds; ++it) {
        const long e = chunk * kSortChunk + it * 64 + lane;
        if (e < n) {
            const long i = idx[e];
            long k = i >= 0 ? (long)((unsigned long long)i / (unsigned long long)pc) : -1;
            if (k < 0 || k >= S) {   // reference: an index error of latent_codes[model_indices]; here a sticky flag, memory stays safe
                // flag_dev: the word the guarded optimizer kernels of the same stream read (sg_adam_step_guarded).  Both words keep
                // the value of the FIRST launch that met a bad index (launches are stream-ordered; within a launch every writer
                // stores the same value)
                if (!flag_dev) {
                    *flag = flag_value;
                } else if (*flag_dev == 0 || *flag_dev == flag_value) {
                    *flag_dev = flag_value;
                    *flag = flag_value;
                }
                k = k < 0 ? 0 : S - 1;
            }
            keys[e] = (int)k;
            atomicAdd(&cnt[k], 1);
        }
    }
    __syncthreads();
    for (int s = lane; s < S; s += 64) hist[chunk * S + s] = cnt[s];
}

// base[c][s] = sum_{c' < c} hist[c'][s], total[s] = sum_c hist[c][s]; one workgroup per 64 shapes, 16 waves split the chunks
__global__ void __launch_bounds__(1024) sdf_sort_prefix_kernel(const int* __restrict__ hist, int* __restrict__ base,
                                                               int* __restrict__ total, long nchunk, int S) {
    __shared__ int part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    const long per = (nchunk + 15) / 16;
    const long c0 = wave * per, c1 = c0 + per < nchunk ? c0 + per : nchunk;
    // (eight loads in flight per thread: a wave's range is a chain of dependent additions, not of dependent loads)
    int run = 0;
    if (s < S) {
        long c = c0;
        for (; c + 8 <= c1; c += 8) {
            int h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = hist[(c + u) * S + s];
#pragma unroll
            for (int u = 0; u < 8; ++u) run += h[u];
        }
        for (; c < c1; ++c) run += hist[c * S + s];
    }
    part[wave][lane] = run;
    __syncthreads();
    int pre = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int v = part[w][lane];
        pre += w < wave ? v : 0;
        all += v;
    }
    if (s < S) {
        run = pre;
        long c = c0;
        for (; c + 8 <= c1; c += 8) {
            int h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = hist[(c + u) * S + s];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                base[(c + u) * S + s] = run;
                run += h[u];
            }
        }
        for (; c < c1; ++c) {
            base[c * S + s] = run;
            run += hist[c * S + s];
        }
        if (wave == 0) total[s] = all;
    }
}

template <int kSortRounds>
__global__ void __launch_bounds__(64) sdf_sort_scatter_kernel(const int64_t* __restrict__ idx, const int* __restrict__ keys,
                                                              const int* __restrict__ base, const int* __restrict__ total,
                                                              long n, int S, long table_rows, const float* __restrict__ points,
                                                              const float* __restrict__ sdf, float* __restrict__ out_points,
                                                              float* __restrict__ out_sdf, int* __restrict__ out_shape,
                                                              int64_t* __restrict__ seg_off, float* __restrict__ counts) {
    constexpr int kSortChunk = kSortRounds * 64;
    extern __shared__ int pos[];     // [S] next free position of every shape for this chunk
    __shared__ int lsum[64];
    const int lane = threadIdx.x;
    const long chunk = blockIdx.x;
    // The wave's work is a chain of memory round trips, so everything whose address is known is requested up front: the keys and
    // batch indices of the chunk's 8 rounds together with the shape totals / chunk bases of the scan below (one round trip), then
    // the table rows as soon as the indices are there (second round trip) — the scan and the ranks (LDS and lane traffic only)
    // run while those are in flight.
    int key[kSortRounds];
    long src[kSortRounds];
#pragma unroll
    for (int it = 0; it < kSortRounds; ++it) {
        const long e = chunk * kSortChunk + it * 64 + lane;
        const bool ok = e < n;
        key[it] = ok ? keys[e] : -1 - lane;   // distinct negative keys: match nothing
        const long i = ok ? idx[e] : 0;
        src[it] = i < 0 ? 0 : (i < table_rows ? i : table_rows - 1);   // (a bad index was flagged by the histogram pass)
    }
    // exclusive scan of the shape totals: lane owns the bins [lane R, lane R + R)
    const int R = (S + 63) / 64;
    int mine = 0;
    for (int j = 0; j < R; ++j) {
        const int s = lane * R + j;
        mine += s < S ? total[s] : 0;
    }
    float x[kSortRounds], y[kSortRounds], z[kSortRounds], d[kSortRounds];
#pragma unroll
    for (int it = 0; it < kSortRounds; ++it) {
        const float* q = points + src[it] * 3;   // (row 0 for the lanes beyond n: in range, never stored)
        x[it] = q[0];
        y[it] = q[1];
        z[it] = q[2];
        d[it] = sdf[src[it]];
    }
    lsum[lane] = mine;
    __syncthreads();
    int run = 0;
    for (int l = 0; l < 64; ++l) run += l < lane ? lsum[l] : 0;
    for (int j = 0; j < R; ++j) {
        const int s = lane * R + j;
        if (s < S) {
            const int t = total[s];
            pos[s] = run + base[chunk * S + s];
            if (chunk == 0) {
                seg_off[s] = run;
                counts[s] = (float)t;
            }
            run += t;
        }
    }
    if (chunk == 0 && lane == 63) seg_off[S] = n;
    __syncthreads();
    int p[kSortRounds];
#pragma unroll
    for (int it = 0; it < kSortRounds; ++it) {
        const bool ok = key[it] >= 0;
        int rank = 0, later = 0;
#pragma unroll
        for (int l = 0; l < 64; ++l) {
            const int kl = __builtin_amdgcn_readlane(key[it], l);
            rank += (kl == key[it] && l < lane) ? 1 : 0;
            later |= (kl == key[it] && l > lane) ? 1 : 0;
        }
        p[it] = ok ? pos[key[it]] + rank : 0;
        __syncthreads();
        if (ok && !later) pos[key[it]] = p[it] + 1;
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < kSortRounds; ++it) {
        if (key[it] >= 0) {
            float* dst = out_points + (long)p[it] * 3;
            dst[0] = x[it];
            dst[1] = y[it];
            dst[2] = z[it];
            out_sdf[p[it]] = d[it];
            out_shape[p[it]] = key[it];
        }
    }
}

static int sort_rounds(long n, long S) { return ((n + 63) / 64) * S <= kSortSmallTableInts ? 1 : 8; }
static long sort_chunks(long n, long S) {
    const long c = 64L * sort_rounds(n, S);
    return (n + c - 1) / c;
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_sdf_batch_sort_max_shapes(void) { return kSortMaxShapes; }

// workspace: keys[n] | hist[nchunk][S] | base[nchunk][S] | total[S] | flag   (ints)
size_t sg_sdf_batch_sort_workspace_bytes(long n, long nshapes) {
    return (size_t)(n + 2 * sort_chunks(n, nshapes) * nshapes + nshapes + 4) * sizeof(int);
}

int sg_sdf_batch_sort(const int64_t* indices, long n, long pointcloud_size, long nshapes, const float* points,
                      const float* sdf, float* out_points, float* out_sdf, int* out_shape, int64_t* seg_off, float* counts,
                      int* bad_index_flag, int* bad_index_device, int bad_index_value, void* workspace, size_t workspace_bytes,
                      hipStream_t stream) {
    SG_CHECK_ARG(indices && points && sdf && out_points && out_sdf && out_shape && seg_off && counts && bad_index_flag);
    SG_CHECK_ARG(bad_index_value != 0);
    SG_CHECK_ARG(n > 0 && n < (1L << 31) && pointcloud_size > 0 && nshapes > 0 && nshapes <= kSortMaxShapes);
    if (!workspace || workspace_bytes < sg_sdf_batch_sort_workspace_bytes(n, nshapes))
        SG_FAIL(SG_ERR_WORKSPACE, "sg_sdf_batch_sort: workspace too small");
    const long nc = sort_chunks(n, nshapes);
    const bool one = sort_rounds(n, nshapes) == 1;
    const int S = (int)nshapes;
    int* keys = (int*)workspace;
    int* hist = keys + n;
    int* base = hist + nc * S;
    int* total = base + nc * S;
    const size_t lds = (size_t)S * sizeof(int);
    if (lds > 48 * 1024) {     // (only with thousands of shapes: the eight-round form)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sdf_sort_hist_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(sdf_sort_scatter_kernel<8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(sdf_sort_hist_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(sdf_sort_scatter_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            SG_FAIL(SG_ERR_HIP, "sg_sdf_batch_sort: cannot reserve %zu B LDS", lds);
    }
    if (one)
        hipLaunchKernelGGL(sdf_sort_hist_kernel<1>, dim3((unsigned)nc), dim3(64), lds, stream, indices, n, pointcloud_size, S, keys,
                           hist, bad_index_flag, bad_index_device, bad_index_value);
    else
        hipLaunchKernelGGL(sdf_sort_hist_kernel<8>, dim3((unsigned)nc), dim3(64), lds, stream, indices, n, pointcloud_size, S, keys,
                           hist, bad_index_flag, bad_index_device, bad_index_value);
    hipLaunchKernelGGL(sdf_sort_prefix_kernel, dim3((unsigned)((S + 63) / 64)), dim3(1024), 0, stream, hist, base, total, nc, S);
    if (one)
        hipLaunchKernelGGL(sdf_sort_scatter_kernel<1>, dim3((unsigned)nc), dim3(64), lds, stream, indices, keys, base, total, n, S,
                           nshapes * pointcloud_size, points, sdf, out_points, out_sdf, out_shape, seg_off, counts);
    else
        hipLaunchKernelGGL(sdf_sort_scatter_kernel<8>, dim3((unsigned)nc), dim3(64), lds, stream, indices, keys, base, total, n, S,
                           nshapes * pointcloud_size, points, sdf, out_points, out_sdf, out_shape, seg_off, counts);
    SG_CHECK_LAUNCH();
    return SG_OK;
}

}  // extern "C"
