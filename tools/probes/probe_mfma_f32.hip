// tools/probes/probe_mfma_f32.hip — what the fp32 matrix instructions sustain on their own (round 6): a register-only loop of independent
// accumulators, 16x16x4 against 32x32x2, with 0 / 1 / 2 plain VALU instructions and 0 / 1 LDS reads per MFMA mixed in.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/probe_mfma_f32 tools/probes/probe_mfma_f32.hip && gpurun_out/probe_mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int VALU, int LDSR>
__global__ __launch_bounds__(256) void k(float* out, int iters, float x) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = x * i;
    __syncthreads();
    float a = x * threadIdx.x, b = x + threadIdx.x, v0 = x, v1 = x * 2;
    const float* lp = lds + (threadIdx.x & 63) * 33;
    float sum = 0.f;
    if constexpr (KIND == 16 && LDSR >= 2) {
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
        float cur[8], nxt[8];
        auto ld = [&](float (&r)[8], int it) {
            if constexpr (LDSR == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = lp[((it & 3) * 8 + j)];
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float4 v = *reinterpret_cast<const float4*>(lds + (threadIdx.x & 63) * 36 + (it & 3) * 8 + j * 4);
                    r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
                }
            }
        };
        auto mm = [&](float (&r)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[j], b, acc[j], 0, 0, 0);
                if constexpr (VALU >= 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(x));
                if constexpr (VALU >= 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(x));
            }
        };
        ld(cur, 0);
        for (int it = 0; it < iters; it += 2) {
            ld(nxt, it + 1); mm(cur);
            ld(cur, it + 2); mm(nxt);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else if constexpr (KIND == 16) {
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float aa = a;
                if constexpr (LDSR) aa = lp[(it * 8 + j) & 31];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, b, acc[j], 0, 0, 0);
                if constexpr (VALU >= 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(x));
                if constexpr (VALU >= 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(x));
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float aa = a;
                if constexpr (LDSR) aa = lp[(it * 4 + j) & 31];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, b, acc[j], 0, 0, 0);
                if constexpr (VALU >= 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(x));
                if constexpr (VALU >= 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(x));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) sum += acc[j][e];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum + v0 + v1;
}

template <int KIND, int VALU, int LDSR>
static void run(float* out, int blocks_per_cu, const char* name) {
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, VALU, LDSR><<<blocks, 256>>>(out, 100, 0.f);
    hipEventRecord(e0);
    k<KIND, VALU, LDSR><<<blocks, 256>>>(out, iters, 0.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (KIND == 16 ? 8.0 * 2048 : 4.0 * 4096) * 2 /*flops per fma*/ * iters * 4.0 * blocks;   // per wave: 16x16x4 = 1024 fma
    printf("{\"kernel\": \"%s\", \"valu_per_mfma\": %d, \"lds_read_per_mfma\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f}\n", name, VALU, LDSR,
           blocks_per_cu, ms, mf / 2 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<16, 0, 0>(out, w, "mfma_f32_16x16x4"); run<16, 1, 0>(out, w, "mfma_f32_16x16x4"); run<16, 2, 0>(out, w, "mfma_f32_16x16x4");
        run<16, 0, 1>(out, w, "mfma_f32_16x16x4"); run<16, 1, 1>(out, w, "mfma_f32_16x16x4");
        run<16, 0, 2>(out, w, "mfma_f32_16x16x4 b32 ahead"); run<16, 1, 2>(out, w, "mfma_f32_16x16x4 b32 ahead");
        run<16, 0, 3>(out, w, "mfma_f32_16x16x4 b128 ahead"); run<16, 1, 3>(out, w, "mfma_f32_16x16x4 b128 ahead");
        run<32, 0, 0>(out, w, "mfma_f32_32x32x2"); run<32, 1, 0>(out, w, "mfma_f32_32x32x2"); run<32, 2, 0>(out, w, "mfma_f32_32x32x2");
        run<32, 0, 1>(out, w, "mfma_f32_32x32x2"); run<32, 1, 1>(out, w, "mfma_f32_32x32x2");
    }
    return 0;
}
