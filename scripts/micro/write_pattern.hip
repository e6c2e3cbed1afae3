// Write-pattern microbenchmark (round 5): how fast does MI355X take 134 MB of output written the way conv_fwd_c1_kernel writes it
// (every wave appends 128 B to each of 64 channel rows that lie 16 KB apart), against longer runs per row and a plain fill?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/write_pattern.hip -o scripts/micro/write_pattern && scripts/micro/write_pattern
// Output rotates over 6 buffers (> 2 x the 256 MB Infinity Cache): cold numbers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// rows of ROWB bytes; a "tile" = RUN bytes of each of 64 consecutive rows (one sample's 64 channel rows)
// MODE 0: wave per tile, 8 instr x (8 rows x 128 B)                        [RUN = 128]   = conv_fwd_c1 today
// MODE 1: wave per tile of RUN = 512: 32 instr x (2 rows x 512 B)
// MODE 2: wave per 16 rows x 512 B (a workgroup of 4 waves covers 64 rows x 512 B): 8 instr x (2 rows x 512 B)
// MODE 3: plain fill, wave writes 1 KB contiguous per instruction
template <int MODE>
__global__ void __launch_bounds__(256) wr(float* out, long rowb, long nrows, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wave;
    u32x4 v = {1u, 2u, 3u, (unsigned)gw};
    char* base = (char*)out;
    if (MODE == 0) {
        const long runs_per_row = rowb / 128;                 // tiles per sample
        for (int t = 0; t < tiles_per_wave; ++t) {
            const long tile = gw * tiles_per_wave + t;
            const long n = tile / runs_per_row, tp = tile % runs_per_row;
            if (n * 64 >= nrows) return;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long row = n * 64 + 8 * i + (lane >> 3);
                *(u32x4*)(base + row * rowb + tp * 128 + (lane & 7) * 16) = v;
            }
        }
    } else if (MODE == 1) {
        const long runs_per_row = rowb / 512;
        for (int t = 0; t < tiles_per_wave / 4; ++t) {
            const long tile = gw * (tiles_per_wave / 4) + t;
            const long n = tile / runs_per_row, tp = tile % runs_per_row;
            if (n * 64 >= nrows) return;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const long row = n * 64 + 2 * i + (lane >> 5);
                *(u32x4*)(base + row * rowb + tp * 512 + (lane & 31) * 16) = v;
            }
        }
    } else if (MODE == 2) {
        const long runs_per_row = rowb / 512;
        for (int t = 0; t < tiles_per_wave; ++t) {
            const long tile = (long)blockIdx.x * tiles_per_wave + t;     // workgroup tile: 64 rows x 512 B
            const long n = tile / runs_per_row, tp = tile % runs_per_row;
            if (n * 64 >= nrows) return;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long row = n * 64 + wave * 16 + 2 * i + (lane >> 5);
                *(u32x4*)(base + row * rowb + tp * 512 + (lane & 31) * 16) = v;
            }
        }
    } else {
        const long total = rowb * nrows / 1024;
        for (long c = gw; c < total; c += (long)gridDim.x * 4) *(u32x4*)(base + c * 1024 + lane * 16) = v;
    }
}

int main() {
    const long rowb = 16384, nrows = 128 * 64;                 // 128 samples x 64 channels x 16^3 floats = 134 MB
    const long bytes = rowb * nrows;
    const int NB = 6;
    std::vector<float*> bufs(NB);
    for (auto& b : bufs) hipMalloc(&b, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long tiles = nrows / 64 * (rowb / 128);               // 16384 tiles of 128 B x 64 rows
    for (int grid : {512, 1024, 2048}) {
        const int tpw = (int)(tiles / ((long)grid * 4));
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&](float* o) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(256), 0, 0, o, rowb, nrows, tpw); break;
                    case 1: hipLaunchKernelGGL(wr<1>, dim3(grid), dim3(256), 0, 0, o, rowb, nrows, tpw); break;
                    case 2: hipLaunchKernelGGL(wr<2>, dim3(grid), dim3(256), 0, 0, o, rowb, nrows, tpw); break;
                    default: hipLaunchKernelGGL(wr<3>, dim3(grid), dim3(256), 0, 0, o, rowb, nrows, tpw); break;
                }
            };
            for (int i = 0; i < 12; ++i) launch(bufs[i % NB]);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int iters = 24;
            for (int i = 0; i < iters; ++i) launch(bufs[i % NB]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms / iters * 1e3;
            printf("{\"grid\": %d, \"mode\": %d, \"us\": %.1f, \"tb_per_s\": %.2f}\n", grid, mode, us, bytes / us / 1e6);
            // warm: the same buffer again and again (134 MB fits the Infinity Cache)
            for (int i = 0; i < 6; ++i) launch(bufs[0]);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) launch(bufs[0]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("{\"grid\": %d, \"mode\": %d, \"warm_us\": %.1f}\n", grid, mode, ms / iters * 1e3);
        }
    }
    return 0;
}
