// cb_device.h — device-side element access shared by the kernel files. Activation kernels are templated on the storage type T
// (float or bf16_t): loads widen to fp32, all arithmetic is fp32, stores round to nearest even (v_cvt_pk_bf16_f32).
// A "quad" is 4 consecutive elements: 16 bytes of fp32 or 8 bytes of bf16.
#pragma once
#include "cb_common.h"

namespace cb {

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)v; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    bf16x4_t o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    *reinterpret_cast<bf16x4_t*>(p) = o;
}
// two fp32 -> one dword of two bf16 (low half = a)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, v);
}

// ---- "the last block to arrive finishes the job" (cdna_hip_programming.md §6 Guideline 16, form R1): every block has written its
// partial results WRITE-THROUGH (sc1 / relaxed agent-scope atomic stores: the bytes leave the XCD's L2, so no release fence). Every
// storing wave drains its stores (asm wait: the compiler does not count on it), barrier, ONE lane draws a ticket (relaxed, agent scope).
// The block that draws n - 1 does ONE agent-scope acquire (drops this CU's L1), puts the counter back to zero for the next launch and
// returns true (uniformly for its threads): it may now read every block's partials with plain loads. `flag`: one int of LDS the block
// no longer uses (never a second __shared__ object in a kernel with a counted LDS-DMA pipeline). `cnt` must be zero at launch.
__device__ __forceinline__ bool grid_arrive_last(int* cnt, int n, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == n - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = ticket;
    }
    __syncthreads();
    return *flag == n - 1;
}

#define CB_DT_SWITCH(dt, T, ...)                                        \
    do {                                                                \
        if ((dt) == DT_BF16) { typedef bf16_t T; __VA_ARGS__; }         \
        else { typedef float T; __VA_ARGS__; }                          \
    } while (0)

}  // namespace cb
