// shapegan_amd/csrc/common.h — shared host/device helpers for libshapegan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>

#define SG_OK 0
#define SG_ERR_ARG (-1)
#define SG_ERR_HIP (-2)
#define SG_ERR_WORKSPACE (-3)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// thread-local last-error string, read through sg_last_error()
char* sg_err_buf();
#define SG_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(sg_err_buf(), 512, __VA_ARGS__);            \
        return (code);                                       \
    } while (0)

#define SG_CHECK_ARG(cond)                                                       \
    do {                                                                         \
        if (!(cond)) SG_FAIL(SG_ERR_ARG, "%s: bad argument: %s", __func__, #cond); \
    } while (0)

#define SG_CHECK_LAUNCH()                                                                  \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess)                                                             \
            SG_FAIL(SG_ERR_HIP, "%s: kernel launch failed: %s", __func__, hipGetErrorString(e__)); \
    } while (0)

// activation codes shared by every entry point that takes an `act` argument
enum { SG_ACT_NONE = 0, SG_ACT_LEAKY = 1, SG_ACT_RELU = 2, SG_ACT_TANH = 3, SG_ACT_SIGMOID = 4 };

__device__ __forceinline__ float sg_apply_act(float v, int act, float slope) {
    switch (act) {
        case SG_ACT_LEAKY: return v > 0.f ? v : v * slope;
        case SG_ACT_RELU: return v > 0.f ? v : 0.f;
        case SG_ACT_TANH: return tanhf(v);
        case SG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize (> 48 KB of dynamic LDS) is a property of (function, DEVICE): a process that drives
// several devices (nn.DataParallel, train_hybrid_progressive_gan.py:62-68: one host thread per device) needs it once per device,
// and two host threads may reach the same launch site at the same time.  `SgPerDeviceOnce once; if (once.begin()) { ...set...;
// once.end(); }`: the fast path is one relaxed-cost atomic load of the current device's bit.
struct SgPerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mu;
    unsigned long long bit = 0;
    bool begin() {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long b = (d >= 0 && d < 64) ? (1ull << d) : 0ull;   // device ids beyond 63: set on every call
        if (b && (done.load(std::memory_order_acquire) & b)) return false;
        mu.lock();
        if (b && (done.load(std::memory_order_relaxed) & b)) {
            mu.unlock();
            return false;
        }
        bit = b;
        return true;      // the caller sets its attributes, then calls end()
    }
    void end() {
        done.fetch_or(bit, std::memory_order_release);
        mu.unlock();
    }
};

static inline int sg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float sg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double sg_wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- zero-VALU addressing helpers (see conv3d_halo.hip for the rationale) ------------------------------------------
namespace sg {
typedef __attribute__((address_space(3))) float lds_float;
constexpr unsigned kBufRange = 0x80000000u;     // num_records of the buffer resources
constexpr unsigned kBufOutside = 0x80000000u;   // byte offset that is out of range -> the load returns 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)kBufRange, 0x00020000);
}
// the same with an exact extent: every dword at or beyond `bytes` reads as 0 WITHOUT touching memory (range checking of raw
// buffers is per dword on gfx950, also inside a 16-byte load: measured).  For operands whose vector loads may overshoot their
// last element.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_bytes(const void* base, long bytes) {
    const unsigned n = bytes <= 0 ? 0u : (bytes < (long)kBufRange ? (unsigned)bytes : kBufRange);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    // bit_cast the WHOLE result: component access on the builtin's own vector type narrows the load to one dword
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ f32x4 buf_load4v(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
template <class T>
__device__ __forceinline__ void pin_vgpr(T& v) {   // the value stays in its register (no rematerialising v_add in the loop)
    asm volatile("" : "+v"(v));
}
template <int V>
struct IntTag {
    static constexpr int value = V;
};
}  // namespace sg
