// Shared device helpers for the kalle-audio MI355X (gfx950) kernels.
// Wave = 64 lanes; all cross-lane helpers below hard-code that.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#define KALLE_OK 0
#define KALLE_ERR_ARG -1      // bad shape / alignment / null pointer
#define KALLE_ERR_LAUNCH -2   // hipLaunch failure
#define KALLE_ERR_UNSUPPORTED -3

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef unsigned short bf16_t;  // raw storage type at the C-ABI

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __builtin_bit_cast(float, ((uint32_t)v) << 16);
}
// round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for <= 16 waves; `red` is LDS scratch of >= 16 floats. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }

// Buffer descriptor with hardware range check (out-of-range loads return 0, stores are dropped).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

// hipGetLastError() is sticky per thread: an earlier, unrelated HIP call of the host application (e.g. a
// hipEventQuery that returned hipErrorNotReady) must not be reported as this launch's failure - clear it first.
#define KALLE_LAUNCH(...)            \
    do {                             \
        (void)hipGetLastError();     \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

// Kernels that ask for more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, which is a
// per-DEVICE property of the function: set once per (kernel, device), any thread.  `done` is the launch site's own
// bitmask over device ordinals (a benign race sets the attribute twice).
static inline void kalle_allow_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_relaxed);
}

extern "C" void kalle_set_last_error(const char* what);
static inline int kalle_check_launch() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return KALLE_OK;
    kalle_set_last_error(hipGetErrorName(e));
    return KALLE_ERR_LAUNCH;
}
