// cb_kernels.hip — HBM-bound kernels of the CelebBasis training step for gfx950 (MI355X):
// GroupNorm(+SiLU), LayerNorm, softmax, GEGLU, activations, layout changes, DDPM glue (q_sample, loss),
// face warp/resize, the CelebBasis coefficient head (L2-norm, basis contraction, row surgery, EMA),
// AdamW, and one-time weight packing. Activations are NHWC ([tokens][channels]) in fp32 or bf16 storage (kernels templated on
// the element type, arithmetic always fp32); parameters, statistics and the coefficient head are fp32.
// wave = 64 lanes everywhere; reductions use 64-wide shuffles.
#include "cb_device.h"
#include <cstdarg>

namespace cb {

#define LAUNCH(kernel, grid, block, L, ...)                                                   \
    do {                                                                                      \
        (L).count();                                                                          \
        if (!(L).dry) {                                                                       \
            hipLaunchKernelGGL(kernel, grid, block, 0, (L).stream, __VA_ARGS__);              \
            CB_HIP(hipGetLastError());                                                        \
        }                                                                                     \
    } while (0)

static inline dim3 grid1d(long n, int per_block = 256, int cap = 8192) {
    long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return dim3((unsigned)b);
}

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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// block-wide sum for 256-thread blocks (4 waves); result valid in all threads
__device__ __forceinline__ float block_sum256(float v, float* sh /*>=4 floats*/) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max256(float v, float* sh) {
    v = wave_max(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ------------------------------------------------------------------ simple elementwise
template <typename T> __global__ void fill_kernel(T* p, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) st1(p + i, v);
}
void k_fill(const Launch& L, DType dt, void* p, long n, float v) {
    CB_DT_SWITCH(dt, T, LAUNCH((fill_kernel<T>), grid1d(n), dim3(256), L, (T*)p, n, v));
}

// dst = a*src + b*dst ; quads when everything is 4-aligned (the common case: copies and gradient accumulation)
template <typename T> __global__ void axpby_kernel(T* __restrict__ dst, const T* __restrict__ src, long n, float a, float b) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = a * ld1(src + i);
        if (b != 0.f) v += b * ld1(dst + i);
        st1(dst + i, v);
    }
}
template <typename T> __global__ void axpby4_kernel(T* __restrict__ dst, const T* __restrict__ src, long nq, float a, float b) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nq; i += (long)gridDim.x * blockDim.x) {
        float4 v = ld4(src + i * 4);
        v.x *= a; v.y *= a; v.z *= a; v.w *= a;
        if (b != 0.f) { const float4 o = ld4(dst + i * 4); v.x += b * o.x; v.y += b * o.y; v.z += b * o.z; v.w += b * o.w; }
        st4(dst + i * 4, v);
    }
}
void k_axpby(const Launch& L, DType dt, void* dst, const void* src, long n, float a, float b) {
    const bool q = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    if (q) CB_DT_SWITCH(dt, T, LAUNCH((axpby4_kernel<T>), grid1d(n / 4), dim3(256), L, (T*)dst, (const T*)src, n / 4, a, b));
    else CB_DT_SWITCH(dt, T, LAUNCH((axpby_kernel<T>), grid1d(n), dim3(256), L, (T*)dst, (const T*)src, n, a, b));
}

template <typename T>
__global__ void copy2d_kernel(T* __restrict__ dst, long ldd, const T* __restrict__ src, long lds, long rows, int cols, float beta) {
    long total = rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / cols;
        int c = (int)(i - r * cols);
        float v = ld1(src + r * lds + c);
        T* d = dst + r * ldd + c;
        if (beta != 0.f) v += beta * ld1(d);
        st1(d, v);
    }
}
template <typename T>
__global__ void copy2d4_kernel(T* __restrict__ dst, long ldd, const T* __restrict__ src, long lds, long rows, int qcols, float beta) {
    long total = rows * qcols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / qcols;
        int c = (int)(i - r * qcols) * 4;
        float4 v = ld4(src + r * lds + c);
        T* d = dst + r * ldd + c;
        if (beta != 0.f) { const float4 o = ld4(d); v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w; }
        st4(d, v);
    }
}
void k_copy2d(const Launch& L, DType dt, void* dst, long ldd, const void* src, long lds, long rows, int cols, float beta) {
    const bool q = (cols % 4 == 0) && (ldd % 4 == 0) && (lds % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
    if (q) CB_DT_SWITCH(dt, T, LAUNCH((copy2d4_kernel<T>), grid1d(rows * (cols / 4)), dim3(256), L, (T*)dst, ldd, (const T*)src, lds, rows, cols / 4, beta));
    else CB_DT_SWITCH(dt, T, LAUNCH((copy2d_kernel<T>), grid1d(rows * cols), dim3(256), L, (T*)dst, ldd, (const T*)src, lds, rows, cols, beta));
}

// element-type conversion with optional zero padding of the row (cols_dst >= cols_src)
template <typename TD, typename TS>
__global__ void convert_kernel(TD* __restrict__ dst, int cd, const TS* __restrict__ src, int cs, long rows, float beta) {
    long total = rows * cd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / cd;
        int c = (int)(i - r * cd);
        float v = c < cs ? ld1(src + r * cs + c) : 0.f;
        if (beta != 0.f) v += beta * ld1(dst + i);
        st1(dst + i, v);
    }
}
void k_convert(const Launch& L, void* dst, DType ddt, int cols_dst, const void* src, DType sdt, int cols_src, long rows, float beta) {
    CB_CHECK(cols_dst >= cols_src && cols_src >= 1, CB_ERR_SHAPE, "convert: %d -> %d columns", cols_src, cols_dst);
    const dim3 g = grid1d(rows * cols_dst);
    if (ddt == DT_BF16 && sdt == DT_F32) LAUNCH((convert_kernel<bf16_t, float>), g, dim3(256), L, (bf16_t*)dst, cols_dst, (const float*)src, cols_src, rows, beta);
    else if (ddt == DT_F32 && sdt == DT_BF16) LAUNCH((convert_kernel<float, bf16_t>), g, dim3(256), L, (float*)dst, cols_dst, (const bf16_t*)src, cols_src, rows, beta);
    else if (ddt == DT_F32) LAUNCH((convert_kernel<float, float>), g, dim3(256), L, (float*)dst, cols_dst, (const float*)src, cols_src, rows, beta);
    else LAUNCH((convert_kernel<bf16_t, bf16_t>), g, dim3(256), L, (bf16_t*)dst, cols_dst, (const bf16_t*)src, cols_src, rows, beta);
}

// dst[b][c][r] = src[b][r][c] through a 32x32 LDS tile (V^T for the materialised bf16 attention)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(T* __restrict__ dst, const T* __restrict__ src, int rows, int cols, long ld_src,
                                                         long sb_src) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const T* s = src + (long)b * sb_src;
    for (int j = ty; j < 32; j += 8) {
        int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? ld1(s + (long)r * ld_src + c) : 0.f;
    }
    __syncthreads();
    T* d = dst + (long)b * rows * cols;
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) st1(d + (long)c * rows + r, tile[tx][j]);
    }
}
void k_transpose(const Launch& L, DType dt, void* dst, const void* src, int batch, int rows, int cols, long ld_src, long sb_src) {
    CB_DT_SWITCH(dt, T, LAUNCH((transpose_kernel<T>), dim3(cdiv(cols, 32), cdiv(rows, 32), batch), dim3(256), L, (T*)dst, (const T*)src, rows,
                               cols, ld_src, sb_src));
}

template <typename T> __global__ void silu_kernel(T* __restrict__ dst, const T* __restrict__ src, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x = ld1(src + i);
        st1(dst + i, x * sigmoidf_(x));
    }
}
void k_silu(const Launch& L, DType dt, void* dst, const void* src, long n) {
    CB_DT_SWITCH(dt, T, LAUNCH((silu_kernel<T>), grid1d(n), dim3(256), L, (T*)dst, (const T*)src, n));
}

// GEGLU (attention.py:37-44): x[M][2D] -> y[M][D] = x[:, :D] * gelu(x[:, D:]), exact erf GELU. D % 4 == 0: one quad per thread.
__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.f + erff(g * 0.70710678118654752f)); }
template <typename T> __global__ void geglu_fwd_kernel(T* __restrict__ y, const T* __restrict__ x, long M, int D) {
    const int Q = D >> 2;
    long total = M * Q;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / Q;
        int c = (int)(i - r * Q) * 4;
        const float4 a = ld4(x + r * 2 * D + c), g = ld4(x + r * 2 * D + D + c);
        st4(y + r * D + c, make_float4(a.x * gelu_erf(g.x), a.y * gelu_erf(g.y), a.z * gelu_erf(g.z), a.w * gelu_erf(g.w)));
    }
}
void k_geglu_fwd(const Launch& L, DType dt, void* y, const void* x, long M, int D) {
    CB_CHECK(D % 4 == 0, CB_ERR_SHAPE, "geglu: D=%d must be a multiple of 4", D);
    CB_DT_SWITCH(dt, T, LAUNCH((geglu_fwd_kernel<T>), grid1d(M * (D / 4)), dim3(256), L, (T*)y, (const T*)x, M, D));
}
__device__ __forceinline__ void geglu_d(float a, float g, float d, float* da, float* dg) {
    float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * __expf(-0.5f * g * g);
    *da = d * g * cdf;
    *dg = d * a * (cdf + g * pdf);
}
template <typename T>
__global__ void geglu_bwd_kernel(T* __restrict__ dx, const T* __restrict__ x, const T* __restrict__ dy, long M, int D) {
    const int Q = D >> 2;
    long total = M * Q;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / Q;
        int c = (int)(i - r * Q) * 4;
        const float4 a = ld4(x + r * 2 * D + c), g = ld4(x + r * 2 * D + D + c), d = ld4(dy + r * D + c);
        float4 da, dg;
        geglu_d(a.x, g.x, d.x, &da.x, &dg.x); geglu_d(a.y, g.y, d.y, &da.y, &dg.y);
        geglu_d(a.z, g.z, d.z, &da.z, &dg.z); geglu_d(a.w, g.w, d.w, &da.w, &dg.w);
        st4(dx + r * 2 * D + c, da);
        st4(dx + r * 2 * D + D + c, dg);
    }
}
void k_geglu_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, long M, int D) {
    CB_CHECK(D % 4 == 0, CB_ERR_SHAPE, "geglu: D=%d must be a multiple of 4", D);
    CB_DT_SWITCH(dt, T, LAUNCH((geglu_bwd_kernel<T>), grid1d(M * (D / 4)), dim3(256), L, (T*)dx, (const T*)x, (const T*)dy, M, D));
}

// quick-GELU x*sigmoid(1.702x) (transformers 4.18 activations.QuickGELUActivation)
template <typename T> __global__ void quickgelu_fwd_kernel(T* __restrict__ y, const T* __restrict__ x, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = ld1(x + i);
        st1(y + i, v * sigmoidf_(1.702f * v));
    }
}
void k_quickgelu_fwd(const Launch& L, DType dt, void* y, const void* x, long n) {
    CB_DT_SWITCH(dt, T, LAUNCH((quickgelu_fwd_kernel<T>), grid1d(n), dim3(256), L, (T*)y, (const T*)x, n));
}
template <typename T>
__global__ void quickgelu_bwd_kernel(T* __restrict__ dx, const T* __restrict__ x, const T* __restrict__ dy, long n, float beta) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = ld1(x + i);
        float s = sigmoidf_(1.702f * v);
        float g = ld1(dy + i) * s * (1.f + 1.702f * v * (1.f - s));
        if (beta != 0.f) g += beta * ld1(dx + i);
        st1(dx + i, g);
    }
}
void k_quickgelu_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, long n, float beta) {
    CB_DT_SWITCH(dt, T, LAUNCH((quickgelu_bwd_kernel<T>), grid1d(n), dim3(256), L, (T*)dx, (const T*)x, (const T*)dy, n, beta));
}

__global__ void leakyrelu_fwd_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float s) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        y[i] = v > 0.f ? v : s * v;
    }
}
void k_leakyrelu_fwd(const Launch& L, float* y, const float* x, long n, float s) {
    LAUNCH(leakyrelu_fwd_kernel, grid1d(n), dim3(256), L, y, x, n, s);
}
__global__ void leakyrelu_bwd_kernel(float* __restrict__ dx, const float* __restrict__ x, const float* __restrict__ dy, long n,
                                     float s) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dx[i] = x[i] > 0.f ? dy[i] : s * dy[i];
}
void k_leakyrelu_bwd(const Launch& L, float* dx, const float* x, const float* dy, long n, float s) {
    LAUNCH(leakyrelu_bwd_kernel, grid1d(n), dim3(256), L, dx, x, dy, n, s);
}

// layout changes at the ABI edge: the reference's public latents / noise / images are fp32 NCHW
template <typename T> __global__ void nchw_to_nhwc_kernel(T* __restrict__ dst, const float* __restrict__ src, int N, int C, int H, int W) {
    long total = (long)N * C * H * W;
    long HW = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes dst (n, hw, c)
        int c = (int)(i % C);
        long t = i / C;
        long hw = t % HW;
        long n = t / HW;
        st1(dst + i, src[(n * C + c) * HW + hw]);
    }
}
void k_nchw_to_nhwc(const Launch& L, DType ddt, void* dst, const float* src, int N, int C, int H, int W) {
    CB_DT_SWITCH(ddt, T, LAUNCH((nchw_to_nhwc_kernel<T>), grid1d((long)N * C * H * W), dim3(256), L, (T*)dst, src, N, C, H, W));
}
template <typename T> __global__ void nhwc_to_nchw_kernel(float* __restrict__ dst, const T* __restrict__ src, int N, int C, int H, int W) {
    long total = (long)N * C * H * W;
    long HW = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes dst (n, c, hw)
        long hw = i % HW;
        long t = i / HW;
        int c = (int)(t % C);
        long n = t / C;
        dst[i] = ld1(src + (n * HW + hw) * C + c);
    }
}
void k_nhwc_to_nchw(const Launch& L, DType sdt, float* dst, const void* src, int N, int C, int H, int W) {
    CB_DT_SWITCH(sdt, T, LAUNCH((nhwc_to_nchw_kernel<T>), grid1d((long)N * C * H * W), dim3(256), L, dst, (const T*)src, N, C, H, W));
}

// y = x*scale[c] + shift[c], then optional PReLU(slope[c]) (iresnet.py:44-63: eval BatchNorm + PReLU)
// (y and x may be the SAME buffer — the de-fused PReLU of cb_gemm.hip runs in place — so neither carries __restrict__)
__global__ void channel_affine_kernel(float* y, const float* x, const float* __restrict__ scale,
                                      const float* __restrict__ shift, const float* __restrict__ prelu, long rows, int C) {
    long total = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        float v = x[i];
        if (scale) v = v * scale[c] + shift[c];
        if (prelu) v = v > 0.f ? v : prelu[c] * v;
        y[i] = v;
    }
}
void k_channel_affine(const Launch& L, float* y, const float* x, const float* scale, const float* shift, const float* prelu,
                      long rows, int C) {
    LAUNCH(channel_affine_kernel, grid1d(rows * C), dim3(256), L, y, x, scale, shift, prelu, rows, C);
}

// backward of nearest-2x upsample: dx[n,y,x,c] (+)= sum of the 2x2 block of dy (dy is [N,2H,2W,C]); C % 4 == 0: quads
template <typename T>
__global__ void upsample2x_bwd_kernel(T* __restrict__ dx, const T* __restrict__ dy, int N, int H, int W, int C, float beta) {
    const int Q = C >> 2;
    long total = (long)N * H * W * Q;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % Q) * 4;
        long t = i / Q;
        int x = (int)(t % W);
        t /= W;
        int y = (int)(t % H);
        long n = t / H;
        const T* p = dy + (((n * 2 * H + 2 * y) * 2 * W) + 2 * x) * (long)C + c;
        const float4 a = ld4(p), b = ld4(p + C), d = ld4(p + (long)2 * W * C), e = ld4(p + (long)2 * W * C + C);
        float4 v = make_float4(a.x + b.x + d.x + e.x, a.y + b.y + d.y + e.y, a.z + b.z + d.z + e.z, a.w + b.w + d.w + e.w);
        T* o = dx + i * 4;
        if (beta != 0.f) { const float4 u = ld4(o); v.x += beta * u.x; v.y += beta * u.y; v.z += beta * u.z; v.w += beta * u.w; }
        st4(o, v);
    }
}
void k_upsample2x_bwd(const Launch& L, DType dt, void* dx, const void* dy, int N, int H, int W, int C, float beta) {
    CB_CHECK(C % 4 == 0, CB_ERR_SHAPE, "upsample bwd: C=%d must be a multiple of 4", C);
    CB_DT_SWITCH(dt, T, LAUNCH((upsample2x_bwd_kernel<T>), grid1d((long)N * H * W * (C / 4)), dim3(256), L, (T*)dx, (const T*)dy, N, H, W, C, beta));
}

// ------------------------------------------------------------------ GroupNorm (NHWC)
// util.py:214-216 GroupNorm32 (eps 1e-5), attention.py:76-77 / model.py:38-39 Normalize (eps 1e-6).
// HBM-bound: fwd reads x twice and writes y once, bwd reads x,dy twice and writes dx once; all global accesses are 16-byte
// per lane over the contiguous channel axis. Statistics are deterministic (no global atomics): stage 1 gives per-(image,
// row-chunk, group) partial sums (shifted by the first element of the group so E[x^2]-E[x]^2 stays well conditioned),
// stage 2 reduces the chunks with one wave per (image, group) in double and expands per-channel constants.
constexpr int GN_MAXQ = 3;   // quads (float4 columns) per thread: C <= 3*256*4 = 3072

static int gn_rows_per_block(int N, int HW) {
    // ~1024 statistics blocks per call (4 per CU); CB_GN_BLOCKS overrides the target for sweeps
    static const long target = [] { const char* e = getenv("CB_GN_BLOCKS"); long v = e ? atol(e) : 1024; return v > 0 ? v : 1024; }();
    long rows = (long)N * HW;
    long rpb = (rows + target - 1) / target;
    if (rpb < 8) rpb = 8;
    if (rpb > 256) rpb = 256;
    return (int)rpb;
}
size_t gn_ws_floats(int N, int HW, int C, int G) {
    int nchunk = cdiv(HW, gn_rows_per_block(N, HW));
    return (size_t)N * nchunk * G * 2 + (size_t)N * C * 4 + (size_t)N * C * 2 + 64;
}

// thread -> (quad slots, first row, row step) for a [rows][C] tile processed by 256 threads
struct GnMap { int Q, rif, rr; int q[GN_MAXQ]; int nq; };
__device__ __forceinline__ GnMap gn_map(int C) {
    GnMap m; m.Q = C >> 2;
    if (m.Q <= 256) {
        m.rif = 256 / m.Q; m.rr = threadIdx.x / m.Q; m.nq = (m.rr < m.rif) ? 1 : 0; m.q[0] = threadIdx.x % m.Q; m.q[1] = m.q[2] = 0;
    } else {
        m.rif = 1; m.rr = 0; m.nq = 0;
#pragma unroll
        for (int s = 0; s < GN_MAXQ; ++s) { int q = threadIdx.x + s * 256; m.q[s] = q; if (q < m.Q) m.nq = s + 1; }
    }
    return m;
}

// deterministic block reduction: every (row-slot, channel) cell of the LDS tile is written by exactly one thread, then
// one thread per group sums its cells in a fixed order and writes part[blockIdx.x][g][2]
__device__ __forceinline__ void gn_block_reduce(float* __restrict__ part, float* chs, const GnMap& m, const float (*sa)[4],
                                                const float (*sb)[4], int C, int G) {
#pragma unroll
    for (int s = 0; s < GN_MAXQ; ++s)
        if (s < m.nq) {
            float* cell = chs + ((long)m.rr * C + m.q[s] * 4) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cell[2 * j] = sa[s][j]; cell[2 * j + 1] = sb[s][j]; }
        }
    __syncthreads();
    int cpg = C / G;
    if (threadIdx.x < G) {
        int g = threadIdx.x;
        float a = 0.f, b = 0.f;
        for (int rr = 0; rr < m.rif; ++rr)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += chs[((long)rr * C + c) * 2]; b += chs[((long)rr * C + c) * 2 + 1]; }
        float* o = part + ((long)blockIdx.x * G + g) * 2;
        // write-through (sc1): the launch's last-arriving block reads these from another CU / XCD (grid_arrive_last)
        __hip_atomic_store(o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
static size_t gn_lds_bytes(int C) { int Q = C / 4; int rif = Q <= 256 ? 256 / Q : 1; return (size_t)rif * C * 2 * sizeof(float); }

// one wave per (n, g): reduce chunk partials in double -> stats (mean, rstd) and per-channel {mean, rstd, gamma, beta}
template <typename T>
__device__ __forceinline__ void gn_final_one(int w, int lane, float* __restrict__ stats, float* __restrict__ chn, const float* __restrict__ part,
                                             const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C,
                                             int G, int nchunk, float eps) {
    int n = w / G, g = w % G, cpg = C / G;
    double a = 0, b = 0;
    for (int ch = lane; ch < nchunk; ch += 64) {
        const float* o = part + (((long)n * nchunk + ch) * G + g) * 2;
        a += o[0]; b += o[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    double cnt = (double)HW * cpg;
    double sh = ld1(x + (long)n * HW * C + g * cpg);
    double mloc = a / cnt;
    double var = b / cnt - mloc * mloc;
    if (var < 0) var = 0;
    float mean = (float)(mloc + sh), rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (lane == 0) { stats[w * 2] = mean; stats[w * 2 + 1] = rstd; }
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64)
        *reinterpret_cast<float4*>(chn + ((long)n * C + c) * 4) = make_float4(mean, rstd, gamma[c], beta[c]);
}
// cnt != nullptr (experiment, CB_GN_INLAUNCH=1): the statistics pass finishes itself — the launch's last-arriving block
// (grid_arrive_last) reduces all chunk partials and writes stats + the per-channel constants: no gn_final launch between the two passes
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(float* __restrict__ part, const T* __restrict__ x, int HW, int C,
                                                          int G, int nchunk, int rpb, int* cnt, float* __restrict__ stats,
                                                          float* __restrict__ chn, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int N, float eps) {
    extern __shared__ __attribute__((aligned(16))) float chs[];
    int n = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    int r0 = chunk * rpb, r1 = min(HW, r0 + rpb);
    int cpg = C / G;
    const T* xn = x + (long)n * HW * C;
    GnMap m = gn_map(C);
    float sa[GN_MAXQ][4], sb[GN_MAXQ][4], sh[GN_MAXQ][4];
#pragma unroll
    for (int s = 0; s < GN_MAXQ; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sa[s][j] = 0.f; sb[s][j] = 0.f; sh[s][j] = 0.f; }
#pragma unroll
    for (int s = 0; s < GN_MAXQ; ++s)
        if (s < m.nq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sh[s][j] = ld1(xn + ((m.q[s] * 4 + j) / cpg) * cpg);   // shift: row 0, first channel of the group
        }
    for (int r = r0 + m.rr; r < r1; r += m.rif) {
#pragma unroll
        for (int s = 0; s < GN_MAXQ; ++s)
            if (s < m.nq) {
                float4 v = ld4(xn + (long)r * C + m.q[s] * 4);
                float d0 = v.x - sh[s][0], d1 = v.y - sh[s][1], d2 = v.z - sh[s][2], d3 = v.w - sh[s][3];
                sa[s][0] += d0; sa[s][1] += d1; sa[s][2] += d2; sa[s][3] += d3;
                sb[s][0] += d0 * d0; sb[s][1] += d1 * d1; sb[s][2] += d2 * d2; sb[s][3] += d3 * d3;
            }
    }
    gn_block_reduce(part, chs, m, sa, sb, C, G);
    if (!cnt) return;
    if (!grid_arrive_last(cnt, (int)gridDim.x, reinterpret_cast<int*>(chs))) return;
    for (int w = threadIdx.x >> 6; w < N * G; w += 4) gn_final_one<T>(w, threadIdx.x & 63, stats, chn, part, x, gamma, beta, HW, C, G, nchunk, eps);
}
template <typename T>
__global__ __launch_bounds__(256) void gn_final_kernel(float* __restrict__ stats, float* __restrict__ chn,
                                                        const float* __restrict__ part, const T* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, int N, int HW,
                                                        int C, int G, int nchunk, float eps) {
    int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= N * G) return;
    gn_final_one<T>(w, lane, stats, chn, part, x, gamma, beta, HW, C, G, nchunk, eps);
}
// rebuild the per-channel constants from saved stats (backward entry through the ABI)
__global__ void gn_expand_kernel(float* __restrict__ chn, const float* __restrict__ stats, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, int N, int C, int G) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    int n = i / C, c = i % C, g = c / (C / G);
    *reinterpret_cast<float4*>(chn + (long)i * 4) = make_float4(stats[(n * G + g) * 2], stats[(n * G + g) * 2 + 1], gamma[c], beta[c]);
}
// one 16-byte access per thread and pass: 1 quad of fp32, 2 quads of bf16 (C % 8 == 0; otherwise the single-quad form)
template <typename T> struct GnVec { static constexpr int VQ = 1; };
template <> struct GnVec<bf16_t> { static constexpr int VQ = 2; };
__device__ __forceinline__ void gn_ldq(const float* p, float4 (&v)[1]) { v[0] = ld4(p); }
__device__ __forceinline__ void gn_ldq(const bf16_t* p, float4 (&v)[1]) { v[0] = ld4(p); }
__device__ __forceinline__ void gn_ldq(const bf16_t* p, float4 (&v)[2]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    v[1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
}
__device__ __forceinline__ void gn_stq(float* p, const float4 (&v)[1]) { st4(p, v[0]); }
__device__ __forceinline__ void gn_stq(bf16_t* p, const float4 (&v)[1]) { st4(p, v[0]); }
__device__ __forceinline__ void gn_stq(bf16_t* p, const float4 (&v)[2]) {
    uint4 u;
    u.x = pack_bf16x2(v[0].x, v[0].y); u.y = pack_bf16x2(v[0].z, v[0].w); u.z = pack_bf16x2(v[1].x, v[1].y); u.w = pack_bf16x2(v[1].z, v[1].w);
    *reinterpret_cast<uint4*>(p) = u;
}
// Apply pass. The grid's total thread count is a multiple of the units per row (host: gn_apply_grid), so a thread visits the SAME
// channels in every iteration of its grid-stride loop: the per-channel constants (mean, rstd * gamma, beta) live in registers and are
// re-read only when the walk crosses into the next image. (Round 2's form fetched a 16-byte constant record per ELEMENT — eight
// table loads per 16-byte payload load in bf16 — and topped out at 2.1 TB/s on the VAE's 268 MB tensors.)
template <typename T, int VQ>
__global__ __launch_bounds__(256) void gn_apply_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                        const float* __restrict__ chn, long nquads, int HW, int C, int silu) {
    const int Q = C >> 2, UPR = Q / VQ;
    const long nunits = nquads / VQ, per_img = (long)HW * UPR, stride = (long)gridDim.x * 256;
    const long u0 = blockIdx.x * 256L + threadIdx.x;
    if (u0 >= nunits) return;
    const int c0 = (int)(u0 % UPR) * 4 * VQ;
    long n = u0 / per_img, rem = u0 - n * per_img;
    float mean[4 * VQ], sc[4 * VQ], sh[4 * VQ];
    bool fresh = true;
    for (long u = u0; u < nunits; u += stride) {
        if (fresh) {
            fresh = false;
#pragma unroll
            for (int j = 0; j < 4 * VQ; ++j) {
                const float4 kk = *reinterpret_cast<const float4*>(chn + (n * C + c0 + j) * 4);
                mean[j] = kk.x; sc[j] = kk.y * kk.z; sh[j] = kk.w;
            }
        }
        float4 v[VQ], r[VQ];
        gn_ldq(x + u * VQ * 4, v);
#pragma unroll
        for (int w = 0; w < VQ; ++w) {
            const float in[4] = {v[w].x, v[w].y, v[w].z, v[w].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = (in[j] - mean[4 * w + j]) * sc[4 * w + j] + sh[4 * w + j];
                o[j] = silu ? t * sigmoidf_(t) : t;
            }
            r[w] = make_float4(o[0], o[1], o[2], o[3]);
        }
        gn_stq(y + u * VQ * 4, r);
        rem += stride;
        while (rem >= per_img) { rem -= per_img; ++n; fresh = true; }
    }
}
// grid for the apply kernels: as many blocks as grid1d would give, rounded down to a multiple of UPR / gcd(UPR, 256)
static dim3 gn_apply_grid(long nunits, int UPR) {
    long g = std::min<long>((nunits + 255) / 256, 4096);
    int a = UPR, b2 = 256;
    while (b2) { int t = a % b2; a = b2; b2 = t; }
    const long m = UPR / a;
    g = std::max<long>(m, g / m * m);
    return dim3((unsigned)g);
}

// backward: t = x_hat*gamma+beta ; dt = dy * silu'(t) (or dy) ; s1 = mean(dt*gamma) ; s2 = mean(dt*gamma*x_hat) (per n,g)
// dx = rstd * (dt*gamma - s1 - x_hat*s2)
__device__ __forceinline__ float gn_dt(float xh, float gam, float bet, float dy, int silu) {
    if (!silu) return dy;
    float t = xh * gam + bet;
    float s = sigmoidf_(t);
    return dy * s * (1.f + t * (1.f - s));
}
// ---- one block per (image, group), two passes over the group's [HW][cpg] slab (second pass from L2): the UNet-sized tensors
// (the 16x16 and 8x8 levels: <= 48 KB per group) are launch-latency bound in the three-kernel form above, so statistics,
// the per-channel constants and the normalisation go into ONE launch. Element pairs (cpg is even for every width used:
// C % 64 == 0) are walked in a fixed thread order and reduced by a fixed shuffle / LDS tree: bitwise reproducible.
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const bf16_t* p) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void st2(bf16_t* p, float2 v) { *reinterpret_cast<unsigned*>(p) = pack_bf16x2(v.x, v.y); }
constexpr int GNF_THREADS = 1024;
// block-wide sums of two values; every thread returns the totals
__device__ __forceinline__ void gnf_reduce2(double& a, double& b, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                    // sh may still be read from a previous call
    if (lane == 0) { sh[2 * wave] = a; sh[2 * wave + 1] = b; }
    __syncthreads();
    a = 0; b = 0;
#pragma unroll
    for (int w = 0; w < GNF_THREADS / 64; ++w) { a += sh[2 * w]; b += sh[2 * w + 1]; }
}
template <typename T>
__global__ __launch_bounds__(GNF_THREADS) void gn_fused_fwd_kernel(T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ stats,
                                                                    float* __restrict__ chn, int HW, int C, int G, float eps, int silu) {
    __shared__ double sh[2 * GNF_THREADS / 64];
    const int n = blockIdx.x / G, g = blockIdx.x - n * G, cpg = C / G, P = cpg >> 1;
    const long base = (long)n * HW * C + g * cpg;
    const int total = HW * P, ra = GNF_THREADS / P, rb = GNF_THREADS - ra * P;
    const float shift = ld1(x + base);
    float sa = 0.f, sb = 0.f;
    int row = threadIdx.x / P, pp = threadIdx.x - row * P;
    for (int i = threadIdx.x; i < total; i += GNF_THREADS) {
        const float2 v = ld2(x + base + (long)row * C + 2 * pp);
        const float d0 = v.x - shift, d1 = v.y - shift;
        sa += d0 + d1; sb += d0 * d0 + d1 * d1;
        row += ra; pp += rb; if (pp >= P) { pp -= P; ++row; }
    }
    double a = sa, b = sb;
    gnf_reduce2(a, b, sh);
    const double cnt = (double)HW * cpg, mloc = a / cnt;
    double var = b / cnt - mloc * mloc;
    if (var < 0) var = 0;
    const float mean = (float)(mloc + (double)shift), rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { stats[blockIdx.x * 2] = mean; stats[blockIdx.x * 2 + 1] = rstd; }
    if (threadIdx.x < cpg) {
        const int c = g * cpg + threadIdx.x;
        *reinterpret_cast<float4*>(chn + ((long)n * C + c) * 4) = make_float4(mean, rstd, gamma[c], beta[c]);
    }
    row = threadIdx.x / P; pp = threadIdx.x - row * P;
    for (int i = threadIdx.x; i < total; i += GNF_THREADS) {
        const long off = base + (long)row * C + 2 * pp;
        const float2 v = ld2(x + off);
        const int c = g * cpg + 2 * pp;
        float t0 = (v.x - mean) * rstd * gamma[c] + beta[c], t1 = (v.y - mean) * rstd * gamma[c + 1] + beta[c + 1];
        if (silu) { t0 = t0 * sigmoidf_(t0); t1 = t1 * sigmoidf_(t1); }
        st2(y + off, make_float2(t0, t1));
        row += ra; pp += rb; if (pp >= P) { pp -= P; ++row; }
    }
}
template <typename T>
__global__ __launch_bounds__(GNF_THREADS) void gn_fused_bwd_kernel(T* __restrict__ dx, const T* __restrict__ x, const T* __restrict__ dy,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ stats, int HW, int C, int G, int silu,
                                                                    float acc_beta) {
    __shared__ double sh[2 * GNF_THREADS / 64];
    const int n = blockIdx.x / G, g = blockIdx.x - n * G, cpg = C / G, P = cpg >> 1;
    const long base = (long)n * HW * C + g * cpg;
    const int total = HW * P, ra = GNF_THREADS / P, rb = GNF_THREADS - ra * P;
    const float mean = stats[blockIdx.x * 2], rstd = stats[blockIdx.x * 2 + 1];
    float sa = 0.f, sb = 0.f;
    int row = threadIdx.x / P, pp = threadIdx.x - row * P;
    for (int i = threadIdx.x; i < total; i += GNF_THREADS) {
        const long off = base + (long)row * C + 2 * pp;
        const float2 xv = ld2(x + off), dv = ld2(dy + off);
        const int c = g * cpg + 2 * pp;
        const float xh0 = (xv.x - mean) * rstd, xh1 = (xv.y - mean) * rstd;
        const float d0 = gn_dt(xh0, gamma[c], beta[c], dv.x, silu) * gamma[c], d1 = gn_dt(xh1, gamma[c + 1], beta[c + 1], dv.y, silu) * gamma[c + 1];
        sa += d0 + d1; sb += d0 * xh0 + d1 * xh1;
        row += ra; pp += rb; if (pp >= P) { pp -= P; ++row; }
    }
    double a = sa, b = sb;
    gnf_reduce2(a, b, sh);
    const double cnt = (double)HW * cpg;
    const float s1 = (float)(a / cnt), s2 = (float)(b / cnt);
    row = threadIdx.x / P; pp = threadIdx.x - row * P;
    for (int i = threadIdx.x; i < total; i += GNF_THREADS) {
        const long off = base + (long)row * C + 2 * pp;
        const float2 xv = ld2(x + off), dv = ld2(dy + off);
        const int c = g * cpg + 2 * pp;
        const float xh0 = (xv.x - mean) * rstd, xh1 = (xv.y - mean) * rstd;
        const float d0 = gn_dt(xh0, gamma[c], beta[c], dv.x, silu) * gamma[c], d1 = gn_dt(xh1, gamma[c + 1], beta[c + 1], dv.y, silu) * gamma[c + 1];
        float o0 = rstd * (d0 - s1 - xh0 * s2), o1 = rstd * (d1 - s1 - xh1 * s2);
        if (acc_beta != 0.f) { const float2 old = ld2(dx + off); o0 += acc_beta * old.x; o1 += acc_beta * old.y; }
        st2(dx + off, make_float2(o0, o1));
        row += ra; pp += rb; if (pp >= P) { pp -= P; ++row; }
    }
}
// the fused form pays while a group's slab stays small (one block per group = 32 blocks per image, 40-byte row pieces): measured on
// the config-2 step, threshold 0 / 16 / 48 / 96 / 192 / 512 KB -> 40.72 / 40.54 / 40.10 / 40.51 / 41.17 / 41.83 ms (profiles/README.md)
static bool gn_use_fused(int HW, int C, int G, size_t esz) {
    static const long kb = [] { const char* e = getenv("CB_GN_FUSED_KB"); return e ? atol(e) : 48L; }();
    const int cpg = C / G;
    return kb > 0 && cpg % 2 == 0 && cpg / 2 <= GNF_THREADS && (long)HW * cpg * (long)esz <= kb * 1024 && (long)HW * (cpg / 2) < (1L << 31);
}
void k_groupnorm_fwd(const Launch& L, DType dt, void* y, const void* x, const float* gamma, const float* beta, float* stats, float* ws,
                     int N, int HW, int C, int G, float eps, int silu) {
    CB_CHECK(C % G == 0 && C % 4 == 0 && C <= GN_MAXQ * 1024 && G <= 256, CB_ERR_SHAPE, "groupnorm: C=%d G=%d unsupported", C, G);
    int rpb = gn_rows_per_block(N, HW), nchunk = cdiv(HW, rpb);
    float* part = ws;
    float* chn = ws + (size_t)N * nchunk * G * 2;
    const bool fused = gn_use_fused(HW, C, G, dt_size(dt));
    // chip-wide form, experiment (CB_GN_INLAUNCH=1): the statistics launch finishes itself in its last-arriving block instead of a
    // gn_final launch. Measured on the config-2 step and NOT adopted (36.75 against 36.09 ms, profiles/README.md round 6): one block
    // reducing N x 32 groups x up to 1024 chunk partials is a longer serial tail than the 8-block gn_final launch plus its boundary.
    static const bool inl = [] { const char* e = getenv("CB_GN_INLAUNCH"); return e && e[0] == '1'; }();
    int* const cnt = (!fused && inl && L.cnt) ? L.cnt->take(1) : nullptr;
    const double esz = (double)dt_size(dt);
    // A tensor beyond the 256 MiB Infinity Cache (config 3's VAE levels at bs = 4: 268 MB in bf16) is normalised in image groups of at
    // most CB_GN_SPLIT_MB (128) MB: the statistics pass pulls a group through the cache and the apply pass that follows finds it there
    // — 1R + 1W of HBM traffic instead of 2R + 1W. Same kernels on the same per-image data: bitwise the one-shot result.
    static const long split_mb = [] { const char* e = getenv("CB_GN_SPLIT_MB"); return e ? atol(e) : 128L; }();
    const double img_mb = (double)HW * C * esz / 1e6;
    int per = N;
    if (!fused && split_mb > 0 && N > 1 && img_mb * N > 2.0 * split_mb) per = std::max(1, (int)(split_mb / img_mb));
    (void)L.count(fused ? 1 : (cnt ? 2 : 3) * (long)cdiv(N, per));
    if (L.dry) return;
    // records are split by size class: tensors beyond the 256 MB Infinity Cache really stream from HBM, the UNet-sized ones are
    // launch-latency bound and would otherwise drown the achieved-bandwidth figure
    const bool big = (double)N * HW * C * esz >= 64e6;
    ProfRec rec; const bool prof = L.prof && L.prof->begin(&rec, silu ? (big ? "groupnorm_silu_fwd[>=64MB]" : "groupnorm_silu_fwd[<64MB]")
                                                                   : (big ? "groupnorm_fwd[>=64MB]" : "groupnorm_fwd[<64MB]"),
                                                             0.0, 3.0 * N * HW * (double)C * esz, L.stream);
    if (fused) {
        CB_DT_SWITCH(dt, T, {
            hipLaunchKernelGGL((gn_fused_fwd_kernel<T>), dim3(N * G), dim3(GNF_THREADS), 0, L.stream, (T*)y, (const T*)x, gamma, beta, stats, chn, HW, C, G, eps, silu);
        });
    } else {
        for (int n0 = 0; n0 < N; n0 += per) {                 // image groups (one group unless the tensor exceeds the Infinity Cache, above)
            const int Ng = std::min(per, N - n0);
            const long eoff = (long)n0 * HW * C;
            const long nquads = (long)Ng * HW * (C / 4);
            float* part_g = part + (size_t)n0 * nchunk * G * 2;
            float* stats_g = stats + (size_t)n0 * G * 2;
            float* chn_g = chn + (size_t)n0 * C * 4;
            CB_DT_SWITCH(dt, T, {
                const T* xg = (const T*)x + eoff; T* yg = (T*)y + eoff;
                hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(Ng * nchunk), dim3(256), gn_lds_bytes(C), L.stream, part_g, xg, HW, C, G, nchunk, rpb,
                                   cnt, stats_g, chn_g, gamma, beta, Ng, eps);
                if (!cnt) hipLaunchKernelGGL((gn_final_kernel<T>), dim3(cdiv(Ng * G, 4)), dim3(256), 0, L.stream, stats_g, chn_g, part_g, xg, gamma, beta, Ng, HW, C, G, nchunk, eps);
                if (GnVec<T>::VQ == 2 && C % 8 == 0)
                    hipLaunchKernelGGL((gn_apply_kernel<T, GnVec<T>::VQ>), gn_apply_grid(nquads / 2, C / 8), dim3(256), 0, L.stream, yg, xg, chn_g, nquads, HW, C, silu);
                else
                    hipLaunchKernelGGL((gn_apply_kernel<T, 1>), gn_apply_grid(nquads, C / 4), dim3(256), 0, L.stream, yg, xg, chn_g, nquads, HW, C, silu);
            });
        }
    }
    if (prof) L.prof->end(&rec, L.stream);
    CB_HIP(hipGetLastError());
}

// one wave per (n, g): chunk partials -> per-channel {s1, s2} (already divided by the group size)
__device__ __forceinline__ void gn_bwd_final_one(int w, int lane, float* __restrict__ gs, const float* __restrict__ part, int HW, int C, int G,
                                                 int nchunk) {
    int n = w / G, g = w % G, cpg = C / G;
    double a = 0, b = 0;
    for (int ch = lane; ch < nchunk; ch += 64) {
        const float* o = part + (((long)n * nchunk + ch) * G + g) * 2;
        a += o[0]; b += o[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    double cnt = (double)HW * cpg;
    float s1 = (float)(a / cnt), s2 = (float)(b / cnt);
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) { gs[((long)n * C + c) * 2] = s1; gs[((long)n * C + c) * 2 + 1] = s2; }
}
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(float* __restrict__ part, const T* __restrict__ x,
                                                              const T* __restrict__ dy, const float* __restrict__ chn, int HW,
                                                              int C, int G, int nchunk, int rpb, int silu, int* cnt, float* __restrict__ gs,
                                                              int N) {
    extern __shared__ __attribute__((aligned(16))) float chs[];
    int n = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
    int r0 = chunk * rpb, r1 = min(HW, r0 + rpb);
    const T* xn = x + (long)n * HW * C;
    const T* dn = dy + (long)n * HW * C;
    GnMap m = gn_map(C);
    float sa[GN_MAXQ][4], sb[GN_MAXQ][4];
    float4 kk[GN_MAXQ][4];
#pragma unroll
    for (int s = 0; s < GN_MAXQ; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sa[s][j] = 0.f; sb[s][j] = 0.f; kk[s][j] = make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int s = 0; s < GN_MAXQ; ++s)
        if (s < m.nq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) kk[s][j] = *reinterpret_cast<const float4*>(chn + ((long)n * C + m.q[s] * 4 + j) * 4);
        }
    for (int r = r0 + m.rr; r < r1; r += m.rif) {
#pragma unroll
        for (int s = 0; s < GN_MAXQ; ++s)
            if (s < m.nq) {
                float4 xv = ld4(xn + (long)r * C + m.q[s] * 4);
                float4 dv = ld4(dn + (long)r * C + m.q[s] * 4);
                float xi[4] = {xv.x, xv.y, xv.z, xv.w}, di[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xh = (xi[j] - kk[s][j].x) * kk[s][j].y;
                    float dt = gn_dt(xh, kk[s][j].z, kk[s][j].w, di[j], silu) * kk[s][j].z;
                    sa[s][j] += dt; sb[s][j] += dt * xh;
                }
            }
    }
    gn_block_reduce(part, chs, m, sa, sb, C, G);
    if (!cnt) return;                                  // (cnt: as in gn_partial_kernel — the last-arriving block finishes the statistics)
    if (!grid_arrive_last(cnt, (int)gridDim.x, reinterpret_cast<int*>(chs))) return;
    for (int w = threadIdx.x >> 6; w < N * G; w += 4) gn_bwd_final_one(w, threadIdx.x & 63, gs, part, HW, C, G, nchunk);
}
// one wave per (n, g): chunk partials -> per-channel {s1, s2} (already divided by the group size)
__global__ __launch_bounds__(256) void gn_bwd_final_kernel(float* __restrict__ gs, const float* __restrict__ part, int N, int HW,
                                                            int C, int G, int nchunk) {
    int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= N * G) return;
    gn_bwd_final_one(w, lane, gs, part, HW, C, G, nchunk);
}
template <typename T, int VQ>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(T* __restrict__ dx, const T* __restrict__ x,
                                                            const T* __restrict__ dy, const float* __restrict__ chn,
                                                            const float* __restrict__ gs, long nquads, int HW, int C, int silu,
                                                            float acc_beta) {
    // same walk as gn_apply_kernel: the six per-channel constants (mean, rstd, gamma, beta, s1, s2) stay in registers
    const int Q = C >> 2, UPR = Q / VQ;
    const long nunits = nquads / VQ, per_img = (long)HW * UPR, stride = (long)gridDim.x * 256;
    const long u0 = blockIdx.x * 256L + threadIdx.x;
    if (u0 >= nunits) return;
    const int c0 = (int)(u0 % UPR) * 4 * VQ;
    long n = u0 / per_img, rem = u0 - n * per_img;
    float4 kc[4 * VQ]; float2 sc[4 * VQ];
    bool fresh = true;
    for (long u = u0; u < nunits; u += stride) {
        if (fresh) {
            fresh = false;
#pragma unroll
            for (int j = 0; j < 4 * VQ; ++j) {
                kc[j] = *reinterpret_cast<const float4*>(chn + (n * C + c0 + j) * 4);
                sc[j] = *reinterpret_cast<const float2*>(gs + (n * C + c0 + j) * 2);
            }
        }
        const long i = u * VQ;
        float4 xv[VQ], dv[VQ], old[VQ], r[VQ];
        gn_ldq(x + i * 4, xv);
        gn_ldq(dy + i * 4, dv);
        if (acc_beta != 0.f) gn_ldq(dx + i * 4, old);
#pragma unroll
        for (int w = 0; w < VQ; ++w) {
            const float xi[4] = {xv[w].x, xv[w].y, xv[w].z, xv[w].w}, di[4] = {dv[w].x, dv[w].y, dv[w].z, dv[w].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 kk = kc[4 * w + j]; const float2 ss = sc[4 * w + j];
                const float xh = (xi[j] - kk.x) * kk.y;
                const float dt = gn_dt(xh, kk.z, kk.w, di[j], silu) * kk.z;
                o[j] = kk.y * (dt - ss.x - xh * ss.y);
            }
            if (acc_beta != 0.f) { o[0] += acc_beta * old[w].x; o[1] += acc_beta * old[w].y; o[2] += acc_beta * old[w].z; o[3] += acc_beta * old[w].w; }
            r[w] = make_float4(o[0], o[1], o[2], o[3]);
        }
        gn_stq(dx + i * 4, r);
        rem += stride;
        while (rem >= per_img) { rem -= per_img; ++n; fresh = true; }
    }
}
void k_groupnorm_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, const float* gamma, const float* beta,
                     const float* stats, float* ws, int N, int HW, int C, int G, int silu, float acc_beta, bool chn_valid) {
    CB_CHECK(C % G == 0 && C % 4 == 0 && C <= GN_MAXQ * 1024 && G <= 256, CB_ERR_SHAPE, "groupnorm bwd: C=%d G=%d unsupported", C, G);
    int rpb = gn_rows_per_block(N, HW), nchunk = cdiv(HW, rpb);
    float* part = ws;
    float* chn = ws + (size_t)N * nchunk * G * 2;
    float* gs = chn + (size_t)N * C * 4;
    const bool fused = gn_use_fused(HW, C, G, dt_size(dt));
    static const bool inl = [] { const char* e = getenv("CB_GN_INLAUNCH"); return e && e[0] == '1'; }();       // experiment, see k_groupnorm_fwd
    int* const cnt = (!fused && inl && L.cnt) ? L.cnt->take(1) : nullptr;
    (void)L.count(fused ? 1 : (chn_valid ? 3 : 4) - (cnt ? 1 : 0));
    if (L.dry) return;
    const double esz = (double)dt_size(dt);
    const bool big = (double)N * HW * C * esz >= 64e6;
    ProfRec rec; const bool prof = L.prof && L.prof->begin(&rec, silu ? (big ? "groupnorm_silu_bwd[>=64MB]" : "groupnorm_silu_bwd[<64MB]")
                                                                   : (big ? "groupnorm_bwd[>=64MB]" : "groupnorm_bwd[<64MB]"),
                                                             0.0, 5.0 * N * HW * (double)C * esz, L.stream);
    if (fused) {
        CB_DT_SWITCH(dt, T, {
            hipLaunchKernelGGL((gn_fused_bwd_kernel<T>), dim3(N * G), dim3(GNF_THREADS), 0, L.stream, (T*)dx, (const T*)x, (const T*)dy, gamma, beta,
                               stats, HW, C, G, silu, acc_beta);
        });
        if (prof) L.prof->end(&rec, L.stream);
        CB_HIP(hipGetLastError());
        return;
    }
    // chn_valid: `ws` is the forward call's workspace and still holds its per-channel constants (the tape's case)
    if (!chn_valid) hipLaunchKernelGGL(gn_expand_kernel, dim3(cdiv((long)N * C, 256)), dim3(256), 0, L.stream, chn, stats, gamma, beta, N, C, G);
    long nquads = (long)N * HW * (C / 4);
    CB_DT_SWITCH(dt, T, {
        hipLaunchKernelGGL((gn_bwd_partial_kernel<T>), dim3(N * nchunk), dim3(256), gn_lds_bytes(C), L.stream, part, (const T*)x, (const T*)dy,
                           chn, HW, C, G, nchunk, rpb, silu, cnt, gs, N);
        if (!cnt) hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(cdiv(N * G, 4)), dim3(256), 0, L.stream, gs, part, N, HW, C, G, nchunk);
        if (GnVec<T>::VQ == 2 && C % 8 == 0)
            hipLaunchKernelGGL((gn_bwd_apply_kernel<T, GnVec<T>::VQ>), gn_apply_grid(nquads / 2, C / 8), dim3(256), 0, L.stream, (T*)dx, (const T*)x,
                               (const T*)dy, chn, gs, nquads, HW, C, silu, acc_beta);
        else
            hipLaunchKernelGGL((gn_bwd_apply_kernel<T, 1>), gn_apply_grid(nquads, C / 4), dim3(256), 0, L.stream, (T*)dx, (const T*)x, (const T*)dy, chn,
                               gs, nquads, HW, C, silu, acc_beta);
    });
    if (prof) L.prof->end(&rec, L.stream);
    CB_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ LayerNorm: one wave per row, row held in registers
// attention.py:203-205 norm1..3, HF CLIP layer norms. HBM-bound: 16-byte accesses (C % 4 == 0), C <= 64*4*LN_MAXQ.
constexpr int LN_MAXQ = 6;
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ stats, long rows, int C, float eps) {
    long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, Q = C >> 2;
    const T* xr = x + row * C;
    float4 v[LN_MAXQ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        int q = lane + j * 64;
        v[j] = (q < Q) ? ld4(xr + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mean = wave_sum(s) / C;
    float qd = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        if (lane + j * 64 < Q) {
            float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            qd += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(qd) / C + eps);
    if (lane == 0 && stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    T* yr = y + row * C;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        int q = lane + j * 64;
        if (q < Q) {
            float4 g = g4[q], b = b4[q];
            st4(yr + q * 4, make_float4((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y,
                                        (v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w));
        }
    }
}
void k_layernorm_fwd(const Launch& L, DType dt, void* y, const void* x, const float* gamma, const float* beta, float* stats, long rows,
                     int C, float eps) {
    CB_CHECK(C % 4 == 0 && C <= 256 * LN_MAXQ, CB_ERR_SHAPE, "layernorm: C=%d unsupported", C);
    ProfRec rec; const bool prof = !L.dry && L.prof && L.prof->begin(&rec, "layernorm_fwd", 0.0, 2.0 * rows * (double)C * (double)dt_size(dt), L.stream);
    CB_DT_SWITCH(dt, T, LAUNCH((layernorm_fwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), L, (T*)y, (const T*)x, gamma, beta, stats, rows, C, eps));
    if (prof) L.prof->end(&rec, L.stream);
}
// dx = rstd * (dy*gamma - mean(dy*gamma) - x_hat * mean(dy*gamma*x_hat))
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(T* __restrict__ dx, const T* __restrict__ x,
                                                             const T* __restrict__ dy, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, long rows, int C, float acc_beta) {
    long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, Q = C >> 2;
    const T* xr = x + row * C;
    const T* dr = dy + row * C;
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float4 xh[LN_MAXQ], dg[LN_MAXQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        int q = lane + j * 64;
        if (q < Q) {
            float4 xv = ld4(xr + q * 4), dv = ld4(dr + q * 4), g = g4[q];
            xh[j] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            dg[j] = make_float4(dv.x * g.x, dv.y * g.y, dv.z * g.z, dv.w * g.w);
        } else { xh[j] = make_float4(0.f, 0.f, 0.f, 0.f); dg[j] = xh[j]; }
        s1 += (dg[j].x + dg[j].y) + (dg[j].z + dg[j].w);
        s2 += (dg[j].x * xh[j].x + dg[j].y * xh[j].y) + (dg[j].z * xh[j].z + dg[j].w * xh[j].w);
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
    T* o = dx + row * C;
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        int q = lane + j * 64;
        if (q < Q) {
            float4 r = make_float4(rstd * (dg[j].x - s1 - xh[j].x * s2), rstd * (dg[j].y - s1 - xh[j].y * s2),
                                   rstd * (dg[j].z - s1 - xh[j].z * s2), rstd * (dg[j].w - s1 - xh[j].w * s2));
            if (acc_beta != 0.f) { float4 old = ld4(o + q * 4); r.x += acc_beta * old.x; r.y += acc_beta * old.y; r.z += acc_beta * old.z; r.w += acc_beta * old.w; }
            st4(o + q * 4, r);
        }
    }
}
void k_layernorm_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, const float* gamma, const float* stats,
                     long rows, int C, float acc_beta) {
    CB_CHECK(C % 4 == 0 && C <= 256 * LN_MAXQ, CB_ERR_SHAPE, "layernorm bwd: C=%d unsupported", C);
    ProfRec rec; const bool prof = !L.dry && L.prof && L.prof->begin(&rec, "layernorm_bwd", 0.0, 3.0 * rows * (double)C * (double)dt_size(dt), L.stream);
    CB_DT_SWITCH(dt, T, LAUNCH((layernorm_bwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), L, (T*)dx, (const T*)x, (const T*)dy, gamma, stats, rows, C, acc_beta));
    if (prof) L.prof->end(&rec, L.stream);
}

// ------------------------------------------------------------------ row softmax (materialised attention path)
// one 256-thread block per row; n <= 256*16
constexpr int SM_PER_T = 16;
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(T* __restrict__ S, int n, int ld, int causal, int rows_per_mat) {
    __shared__ float sh[4];
    long row = blockIdx.x;
    T* r = S + row * ld;
    int lim = causal ? (int)(row % rows_per_mat) + 1 : n;  // columns [0, lim) are visible
    if (lim > n) lim = n;
    float v[SM_PER_T];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SM_PER_T; ++j) {
        int c = threadIdx.x + j * 256;
        v[j] = (c < lim) ? ld1(r + c) : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = block_max256(mx, sh);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SM_PER_T; ++j) {
        int c = threadIdx.x + j * 256;
        v[j] = (c < lim) ? __expf(v[j] - mx) : 0.f;
        s += v[j];
    }
    s = block_sum256(s, sh);
    float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < SM_PER_T; ++j) {
        int c = threadIdx.x + j * 256;
        if (c < n) st1(r + c, v[j] * inv);
    }
}
// short rows (cross-attention ctx = 77, CLIP 77, 8x8 / 16x16 self-attention): one wave per row, 4 rows per block
constexpr int SMW_PER_L = 8;   // n <= 512
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_wave_kernel(T* __restrict__ S, long rows, int n, int ld, int causal,
                                                                int rows_per_mat) {
    long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    T* r = S + row * ld;
    int lim = causal ? (int)(row % rows_per_mat) + 1 : n;
    if (lim > n) lim = n;
    float v[SMW_PER_L];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SMW_PER_L; ++j) {
        int c = lane + j * 64;
        v[j] = (c < lim) ? ld1(r + c) : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SMW_PER_L; ++j) {
        int c = lane + j * 64;
        v[j] = (c < lim) ? __expf(v[j] - mx) : 0.f;
        s += v[j];
    }
    float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int j = 0; j < SMW_PER_L; ++j) {
        int c = lane + j * 64;
        if (c < n) st1(r + c, v[j] * inv);
    }
}
__global__ __launch_bounds__(256) void softmax_bwd_wave_kernel(float* __restrict__ dP, const float* __restrict__ P, long rows, int n,
                                                                int ld, float scale) {
    long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    float* d = dP + row * ld;
    const float* p = P + row * ld;
    float pv[SMW_PER_L], dv[SMW_PER_L];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SMW_PER_L; ++j) {
        int c = lane + j * 64;
        pv[j] = (c < n) ? p[c] : 0.f;
        dv[j] = (c < n) ? d[c] : 0.f;
        s += pv[j] * dv[j];
    }
    s = wave_sum(s);
#pragma unroll
    for (int j = 0; j < SMW_PER_L; ++j) {
        int c = lane + j * 64;
        if (c < n) d[c] = scale * pv[j] * (dv[j] - s);
    }
}
void k_softmax_fwd(const Launch& L, DType dt, void* S, long rows, int n, int ld, int causal, int rows_per_mat) {
    CB_CHECK(n <= 256 * SM_PER_T, CB_ERR_SHAPE, "softmax: n=%d too large", n);
    if (n <= 64 * SMW_PER_L) CB_DT_SWITCH(dt, T, LAUNCH((softmax_fwd_wave_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), L, (T*)S, rows, n, ld, causal, rows_per_mat));
    else CB_DT_SWITCH(dt, T, LAUNCH((softmax_fwd_kernel<T>), dim3((unsigned)rows), dim3(256), L, (T*)S, n, ld, causal, rows_per_mat));
}
__global__ __launch_bounds__(256) void softmax_bwd_kernel(float* __restrict__ dP, const float* __restrict__ P, int n, int ld,
                                                           float scale) {
    __shared__ float sh[4];
    long row = blockIdx.x;
    float* d = dP + row * ld;
    const float* p = P + row * ld;
    float pv[SM_PER_T], dv[SM_PER_T];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SM_PER_T; ++j) {
        int c = threadIdx.x + j * 256;
        pv[j] = (c < n) ? p[c] : 0.f;
        dv[j] = (c < n) ? d[c] : 0.f;
        s += pv[j] * dv[j];
    }
    s = block_sum256(s, sh);
#pragma unroll
    for (int j = 0; j < SM_PER_T; ++j) {
        int c = threadIdx.x + j * 256;
        if (c < n) d[c] = scale * pv[j] * (dv[j] - s);
    }
}
void k_softmax_bwd(const Launch& L, float* dP, const float* P, long rows, int n, int ld, float scale) {
    CB_CHECK(n <= 256 * SM_PER_T, CB_ERR_SHAPE, "softmax bwd: n=%d too large", n);
    if (n <= 64 * SMW_PER_L) LAUNCH(softmax_bwd_wave_kernel, dim3(cdiv(rows, 4)), dim3(256), L, dP, P, rows, n, ld, scale);
    else LAUNCH(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), L, dP, P, n, ld, scale);
}

// ------------------------------------------------------------------ diffusion glue
// util.py:151-171 timestep_embedding: [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(10000) * i / half)
template <typename T> __global__ void timestep_embedding_kernel(T* __restrict__ out, const int64_t* __restrict__ t, int B, int dim) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int half = dim / 2;
    if (i >= B * half) return;
    int b = i / half, j = i % half;
    float freq = expf(-logf(10000.f) * (float)j / (float)half);
    float arg = (float)t[b] * freq;
    st1(out + (long)b * dim + j, cosf(arg));
    st1(out + (long)b * dim + half + j, sinf(arg));
    if ((dim & 1) && j == 0) st1(out + (long)b * dim + dim - 1, 0.f);
}
void k_timestep_embedding(const Launch& L, DType dt, void* out, const int64_t* t, int B, int dim) {
    CB_DT_SWITCH(dt, T, LAUNCH((timestep_embedding_kernel<T>), dim3(cdiv((long)B * (dim / 2), 256)), dim3(256), L, (T*)out, t, B, dim));
}

// ddpm.py:289-292 q_sample. x0/xt NHWC [B,HW,C]; eps is the caller's NCHW noise.
template <typename T>
__global__ void qsample_kernel(T* __restrict__ xt, const T* __restrict__ x0, const float* __restrict__ eps,
                               const int64_t* __restrict__ t, const float* __restrict__ sa, const float* __restrict__ sb, int B,
                               int C, int HW) {
    long total = (long)B * C * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        long hw = r % HW;
        long b = r / HW;
        float e = eps[(b * C + c) * HW + hw];
        int64_t tt = t[b];
        st1(xt + i, sa[tt] * ld1(x0 + i) + sb[tt] * e);
    }
}
void k_qsample(const Launch& L, DType dt, void* xt, const void* x0, const float* eps_nchw, const int64_t* t, const float* sqrt_ac,
               const float* sqrt_1mac, int B, int C, int HW) {
    CB_DT_SWITCH(dt, T, LAUNCH((qsample_kernel<T>), grid1d((long)B * C * HW), dim3(256), L, (T*)xt, (const T*)x0, eps_nchw, t, sqrt_ac, sqrt_1mac, B, C, HW));
}

// ddim.py:166-204 p_sample_ddim, eps-parameterisation. x (NHWC [n,HW,C]) is updated in place; eps holds the UNet output:
// [n] rows when cfg == 0, or [2n] rows (unconditional half first, ddim.py:175-179) when cfg != 0:
//   e      = e_u + s (e_c - e_u)
//   pred_x0 = (x - sqrt(1-a_t) e) / sqrt(a_t)                       (:189)
//   x_prev  = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) e + sigma * noise     (:193-198)
// The scalar coefficients are evaluated on the host in fp32 exactly as the reference's torch.full(...) tensors are.
template <typename T>
__global__ void ddim_update_kernel(float* __restrict__ x, float* __restrict__ pred_x0, const T* __restrict__ eps,
                                   const float* __restrict__ noise_nchw, long half, int C, int HW, int cfg, float scale,
                                   float sqrt_1m_at, float sqrt_at, float sqrt_aprev, float dir_coef, float sigma) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < half; i += (long)gridDim.x * blockDim.x) {
        float e = ld1(eps + (cfg ? half + i : i));
        if (cfg) { float eu = ld1(eps + i); e = eu + scale * (e - eu); }
        float xv = x[i];
        float p0 = __fdiv_rn(xv - sqrt_1m_at * e, sqrt_at);
        float dir = dir_coef * e;
        float nz = 0.f;
        if (noise_nchw) {
            int c = (int)(i % C);
            long r = i / C;
            long hw = r % HW, b = r / HW;
            nz = sigma * noise_nchw[(b * C + c) * HW + hw];
        }
        x[i] = sqrt_aprev * p0 + dir + nz;
        if (pred_x0) pred_x0[i] = p0;
    }
}
void k_ddim_update(const Launch& L, DType dt, float* x, float* pred_x0, const void* eps, const float* noise_nchw, int n, int C, int HW,
                   int cfg, float scale, float sqrt_1m_at, float sqrt_at, float sqrt_aprev, float dir_coef, float sigma) {
    long half = (long)n * C * HW;
    CB_DT_SWITCH(dt, T, LAUNCH((ddim_update_kernel<T>), grid1d(half), dim3(256), L, x, pred_x0, (const T*)eps, noise_nchw, half, C, HW, cfg, scale,
                               sqrt_1m_at, sqrt_at, sqrt_aprev, dir_coef, sigma));
}
__global__ void fill_i64_kernel(int64_t* p, long n, int64_t v) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void k_fill_i64(const Launch& L, int64_t* p, long n, int64_t v) {
    LAUNCH(fill_i64_kernel, dim3(cdiv(n, 256)), dim3(256), L, p, n, v);
}
// stable_txt2img.py:343: clamp((x + 1) / 2, 0, 1), NHWC -> NCHW
template <typename T> __global__ void image_out_kernel(float* __restrict__ dst, const T* __restrict__ src, int N, int C, int HW, int to01) {
    long total = (long)N * C * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long hw = i % HW;
        long r = i / HW;
        int c = (int)(r % C);
        long b = r / C;
        float v = ld1(src + (b * HW + hw) * C + c);
        if (to01) v = fminf(fmaxf((v + 1.0f) / 2.0f, 0.f), 1.f);
        dst[i] = v;
    }
}
void k_image_out(const Launch& L, DType sdt, float* dst_nchw, const void* src_nhwc, int N, int C, int HW, int to01) {
    CB_DT_SWITCH(sdt, T, LAUNCH((image_out_kernel<T>), grid1d((long)N * C * HW), dim3(256), L, dst_nchw, (const T*)src_nhwc, N, C, HW, to01));
}

// ddpm.py:1069-1116 with l2 / eps-parameterisation / logvar==0: loss = mean((pred - eps)^2); dpred = 2 (pred-eps)/numel
template <typename T>
__global__ __launch_bounds__(256) void mse_loss_kernel(float* __restrict__ loss, T* __restrict__ dpred,
                                                        const T* __restrict__ pred, const float* __restrict__ target, int B,
                                                        int C, int HW) {
    __shared__ float sh[4];
    long total = (long)B * C * HW;
    float inv = 1.f / (float)total;
    float acc = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        long hw = r % HW;
        long b = r / HW;
        float d = ld1(pred + i) - target[(b * C + c) * HW + hw];
        acc += d * d;
        if (dpred) st1(dpred + i, 2.f * d * inv);
    }
    acc = block_sum256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(loss, acc * inv);
}
// the step's latent (<= 256K elements): ONE block of 1024 threads, a pixel per thread and iteration (32-bit index math, the C channels
// of the NHWC prediction contiguous), fixed shuffle / LDS reduction tree -> the scalar is bitwise reproducible and is written directly
// (no zero-fill launch, no atomic). Round 2's 256-thread grid-stride form with 64-bit div / mod per element took 44 us (170 us at bs=4).
template <typename T>
__global__ __launch_bounds__(1024) void mse_loss_block_kernel(float* __restrict__ loss, T* __restrict__ dpred, const T* __restrict__ pred,
                                                              const float* __restrict__ target, int B, int C, int HW) {
    __shared__ float sh[16];
    const int npix = B * HW;
    const float inv = 1.f / ((float)npix * (float)C);
    float acc = 0.f;
    for (int p = threadIdx.x; p < npix; p += 1024) {
        const int b = p / HW, hw = p - b * HW;
        for (int c = 0; c < C; ++c) {
            const float d = ld1(pred + (long)p * C + c) - target[((long)b * C + c) * HW + hw];
            acc += d * d;
            if (dpred) st1(dpred + (long)p * C + c, 2.f * d * inv);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += sh[w];
        *loss = t * inv;
    }
}
void k_mse_loss(const Launch& L, DType dt, float* loss, void* dpred, const void* pred, const float* target_nchw, int B, int C, int HW) {
    const long total = (long)B * C * HW;
    if (total <= (1L << 18)) {
        CB_DT_SWITCH(dt, T, LAUNCH((mse_loss_block_kernel<T>), dim3(1), dim3(1024), L, loss, (T*)dpred, (const T*)pred, target_nchw, B, C, HW));
        return;
    }
    LAUNCH((fill_kernel<float>), dim3(1), dim3(64), L, loss, 1L, 0.f);
    // larger tensors (op-level callers): several blocks + float atomicAdd — the last bits of the scalar then depend on the arrival order
    CB_DT_SWITCH(dt, T, LAUNCH((mse_loss_kernel<T>), grid1d(total, 256, 64), dim3(256), L, loss, (T*)dpred,
                               (const T*)pred, target_nchw, B, C, HW));
}

// distributions.py:24-37 + ddpm.py:597: z = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale
// moments NHWC [B,HW,2*Cz]; noise NCHW [B,Cz,HW]; z NHWC [B,HW,Cz]
template <typename T>
__global__ void vae_sample_kernel(T* __restrict__ z, const T* __restrict__ mom, const float* __restrict__ noise, int B,
                                  int HW, int Cz, float scale) {
    long total = (long)B * HW * Cz;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % Cz);
        long r = i / Cz;
        long hw = r % HW;
        long b = r / HW;
        float mean = ld1(mom + r * 2 * Cz + c);
        float lv = fminf(fmaxf(ld1(mom + r * 2 * Cz + Cz + c), -30.f), 20.f);
        float nz = noise ? noise[(b * Cz + c) * HW + hw] : 0.f;
        st1(z + i, (mean + expf(0.5f * lv) * nz) * scale);
    }
}
void k_vae_sample(const Launch& L, DType dt, void* z, const void* moments, const float* noise_nchw, int B, int HW, int Cz, float scale) {
    CB_DT_SWITCH(dt, T, LAUNCH((vae_sample_kernel<T>), grid1d((long)B * HW * Cz), dim3(256), L, (T*)z, (const T*)moments, noise_nchw, B, HW, Cz, scale));
}

__global__ void gather_rows_kernel(float* __restrict__ out, const float* __restrict__ table, const int64_t* __restrict__ idx,
                                   long rows, int C, long table_rows) {
    long total = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C;
        int c = (int)(i - r * C);
        int64_t id = idx[r];
        out[i] = (id >= 0 && id < table_rows) ? table[id * C + c] : 0.f;
    }
}
void k_gather_rows(const Launch& L, float* out, const float* table, const int64_t* idx, long rows, int C, long table_rows) {
    LAUNCH(gather_rows_kernel, grid1d(rows * C), dim3(256), L, out, table, idx, rows, C, table_rows);
}
// dst[r][c] = x[r][c] + pos[r % period][c]
template <typename T>
__global__ void add_rowbcast_kernel(T* __restrict__ dst, const T* __restrict__ x, const float* __restrict__ pos,
                                    long rows, int period, int C) {
    long total = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C;
        int c = (int)(i - r * C);
        st1(dst + i, ld1(x + i) + pos[(r % period) * C + c]);
    }
}
void k_add_rowbcast(const Launch& L, DType dt, void* dst, const void* x, const float* pos, long rows, int period, int C) {
    CB_DT_SWITCH(dt, T, LAUNCH((add_rowbcast_kernel<T>), grid1d(rows * C), dim3(256), L, (T*)dst, (const T*)x, pos, rows, period, C));
}

// ------------------------------------------------------------------ face pre-processing (meta_net.py:250-260)
// faces: [B, S, S, 3*nface] HWC; out: [nface*B, S, S, 3] with face f of sample b at index f*B + b
// (forward_multi_faces chunks along channels then cats along batch, meta_net.py:336-337).
// affine_grid(theta, align_corners=True) + grid_sample(bilinear, zeros, align_corners=True).
// `grid` (optional, [S][S][2]): the sampling grid as the caller's host evaluated it — the reference builds it with
// F.affine_grid(trans_matrix, size, align_corners=True) from a CONSTANT matrix and a constant 512x512 size (meta_net.py:131-142,
// 253-257), i.e. it is a constant table like the beta schedule. With the table the kernel samples at the reference's own fp32
// coordinates (the fp32 ulp of a 512-pixel coordinate is 6e-5 pixel: re-deriving the grid in another evaluation order moves a
// sample of a white-noise face by 3e-5, profiles/r04_warp_rounding.json) and follows grid_sample's unnormalise / weight order
// (GridSampler.h grid_sampler_unnormalize: ((g + 1) / 2) * (S - 1); corner weights (1-wy)(1-wx) ..., summed nw + ne + sw + se).
__global__ void affine_warp_kernel(float* __restrict__ out, const float* __restrict__ faces, const float* __restrict__ grid, int B,
                                   int S, int nface, float t00, float t01, float t02, float t10, float t11, float t12) {
#pragma clang fp contract(off)
    long total = (long)nface * B * S * S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int x = (int)(i % S);
        long r = i / S;
        int y = (int)(r % S);
        long fb = r / S;
        int f = (int)(fb / B), b = (int)(fb % B);
        float gx, gy;
        if (grid) {
            const float2 gg = *reinterpret_cast<const float2*>(grid + ((long)y * S + x) * 2);
            gx = gg.x; gy = gg.y;
        } else {
            float xn = (2.f * x) / (S - 1) - 1.f;
            float yn = (2.f * y) / (S - 1) - 1.f;
            gx = t00 * xn + t01 * yn + t02;
            gy = t10 * xn + t11 * yn + t12;
        }
        const float half = 0.5f * (float)(S - 1);
        float ix = ((gx + 1.f) * half);
        float iy = ((gy + 1.f) * half);
        float fx0 = floorf(ix), fy0 = floorf(iy);
        int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        float wx1 = ix - fx0, wx0 = 1.f - wx1, wy1 = iy - fy0, wy0 = 1.f - wy1;
        const float* base = faces + (long)b * S * S * 3 * nface + 3 * f;
        long pstride = 3L * nface;
        float acc[3] = {0.f, 0.f, 0.f};
        bool okx0 = x0 >= 0 && x0 < S, okx1 = x1 >= 0 && x1 < S, oky0 = y0 >= 0 && y0 < S, oky1 = y1 >= 0 && y1 < S;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // torch's CPU grid_sample: nw_val * nw + ne_val * ne + sw_val * sw + se_val * se, compiled to one multiply and three
            // fused multiply-adds in that order (bit-exact against F.grid_sample on the same grid: tests/test_gpu_ops.py)
            const float vnw = (oky0 && okx0) ? base[((long)y0 * S + x0) * pstride + c] : 0.f;
            const float vne = (oky0 && okx1) ? base[((long)y0 * S + x1) * pstride + c] : 0.f;
            const float vsw = (oky1 && okx0) ? base[((long)y1 * S + x0) * pstride + c] : 0.f;
            const float vse = (oky1 && okx1) ? base[((long)y1 * S + x1) * pstride + c] : 0.f;
            float v = vnw * (wy0 * wx0);
            v = __builtin_fmaf(vne, wy0 * wx1, v);
            v = __builtin_fmaf(vsw, wy1 * wx0, v);
            v = __builtin_fmaf(vse, wy1 * wx1, v);
            acc[c] = v;
        }
        float* o = out + i * 3;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}
void k_affine_warp(const Launch& L, float* out, const float* faces, int B, int S, int nface, const float* th, const float* grid) {
    LAUNCH(affine_warp_kernel, grid1d((long)nface * B * S * S), dim3(256), L, out, faces, grid, B, S, nface, th[0], th[1], th[2], th[3],
           th[4], th[5]);
}
// ------------------------------------------------------------------ feeder augmentations on the device
// face_id.py:526-532 `trans` after the resize: RandomHorizontalFlip + ColorJitter(b, c, s in [0.8,1.2], hue 0.01) + Normalize,
// on HWC images in [-1,1]. torchvision's tensor formulas, applied in the fixed order brightness -> contrast -> saturation -> hue
// (torchvision draws a random order; bit-exact parity with its PIL path is not a goal).
// params[b] = {flip, brightness, contrast, saturation, hue_shift}
constexpr int AUG_NCHUNK = 64;
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
// pass 1: per (sample, chunk) sum of the grayscale of the brightness-adjusted image (the contrast pivot is its mean)
__global__ __launch_bounds__(256) void jitter_gray_partial_kernel(float* __restrict__ part, const float* __restrict__ in,
                                                                   const float* __restrict__ params, int HW) {
    __shared__ float sh[4];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float br = params[b * 5 + 1];
    const int per = (HW + AUG_NCHUNK - 1) / AUG_NCHUNK;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    float acc = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        const float* px = in + ((long)b * HW + p) * 3;
        float r = clamp01((px[0] + 1.f) * 0.5f * br), g = clamp01((px[1] + 1.f) * 0.5f * br), bl = clamp01((px[2] + 1.f) * 0.5f * br);
        acc += gray_of(r, g, bl);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[b * AUG_NCHUNK + chunk] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void jitter_apply_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                            const float* __restrict__ params, const float* __restrict__ part, int S) {
    __shared__ float mean_s;
    const int b = blockIdx.y, HW = S * S;
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int i = 0; i < AUG_NCHUNK; ++i) m += part[b * AUG_NCHUNK + i];       // fixed order: deterministic
        mean_s = m / (float)HW;
    }
    __syncthreads();
    const float flip = params[b * 5], br = params[b * 5 + 1], ct = params[b * 5 + 2], sa = params[b * 5 + 3], hu = params[b * 5 + 4];
    const float mean = mean_s;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const int y = p / S, x = p - y * S;
        const int xs = flip != 0.f ? S - 1 - x : x;
        const float* px = in + ((long)b * HW + (long)y * S + xs) * 3;
        float r = clamp01((px[0] + 1.f) * 0.5f * br), g = clamp01((px[1] + 1.f) * 0.5f * br), bl = clamp01((px[2] + 1.f) * 0.5f * br);
        r = clamp01(ct * r + (1.f - ct) * mean); g = clamp01(ct * g + (1.f - ct) * mean); bl = clamp01(ct * bl + (1.f - ct) * mean);
        const float gr = gray_of(r, g, bl);
        r = clamp01(sa * r + (1.f - sa) * gr); g = clamp01(sa * g + (1.f - sa) * gr); bl = clamp01(sa * bl + (1.f - sa) * gr);
        if (hu != 0.f) {   // rgb -> hsv, shift h, hsv -> rgb (torchvision _rgb2hsv / _hsv2rgb)
            const float mx = fmaxf(r, fmaxf(g, bl)), mn = fminf(r, fminf(g, bl));
            const float d = mx - mn, v = mx;
            const float sat = mx > 0.f ? d / mx : 0.f;
            float h = 0.f;
            if (d > 0.f) {
                if (mx == r) h = (g - bl) / d;
                else if (mx == g) h = 2.f + (bl - r) / d;
                else h = 4.f + (r - g) / d;
                h = h / 6.f;
                h = h - floorf(h);
            }
            h = h + hu; h = h - floorf(h);
            const float h6 = h * 6.f;
            const float fi = floorf(h6), f = h6 - fi;
            const int i = ((int)fi) % 6;
            const float pp = clamp01(v * (1.f - sat)), q = clamp01(v * (1.f - f * sat)), t = clamp01(v * (1.f - (1.f - f) * sat));
            switch (i) {
                case 0: r = v; g = t; bl = pp; break;
                case 1: r = q; g = v; bl = pp; break;
                case 2: r = pp; g = v; bl = t; break;
                case 3: r = pp; g = q; bl = v; break;
                case 4: r = t; g = pp; bl = v; break;
                default: r = v; g = pp; bl = q; break;
            }
        }
        float* o = out + ((long)b * HW + p) * 3;
        o[0] = r * 2.f - 1.f; o[1] = g * 2.f - 1.f; o[2] = bl * 2.f - 1.f;
    }
}
void k_face_augment(const Launch& L, float* out, const float* in, const float* params_dev, float* ws, int B, int S) {
    L.count(2);
    if (L.dry) return;
    hipLaunchKernelGGL(jitter_gray_partial_kernel, dim3(AUG_NCHUNK, B), dim3(256), 0, L.stream, ws, in, params_dev, S * S);
    hipLaunchKernelGGL(jitter_apply_kernel, dim3(cdiv((long)S * S, 256 * 4), B), dim3(256), 0, L.stream, out, in, params_dev, ws, S);
    CB_HIP(hipGetLastError());
}
size_t face_augment_ws_floats(int B) { return (size_t)B * AUG_NCHUNK; }

// face_id.py:712-730 `_add_bg`: the face bilinearly rescaled (align_corners=True) to rh x rw and pasted at (pos_h, pos_w) on a
// canvas of -1. geom[b] = {rh, rw, pos_h, pos_w} (int32, drawn on the host like the reference's np.random calls).
__global__ void paste_bg_kernel(float* __restrict__ out, const float* __restrict__ img, const int* __restrict__ geom, int B, int S,
                                int C) {
    long total = (long)B * S * S * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        int x = (int)(r % S);
        r /= S;
        int y = (int)(r % S);
        int b = (int)(r / S);
        const int rh = geom[b * 4], rw = geom[b * 4 + 1], ph = geom[b * 4 + 2], pw = geom[b * 4 + 3];
        float v = -1.f;
        const int yy = y - ph, xx = x - pw;
        if (yy >= 0 && yy < rh && xx >= 0 && xx < rw) {
            const float sy = rh > 1 ? (float)(S - 1) / (float)(rh - 1) : 0.f, sx = rw > 1 ? (float)(S - 1) / (float)(rw - 1) : 0.f;
            const float fy = sy * yy, fx = sx * xx;
            int y0 = (int)fy, x0 = (int)fx;
            int y1 = min(y0 + 1, S - 1), x1 = min(x0 + 1, S - 1);
            const float wy = fy - y0, wx = fx - x0;
            const float* base = img + (long)b * S * S * C + c;
            const float v00 = base[((long)y0 * S + x0) * C], v01 = base[((long)y0 * S + x1) * C];
            const float v10 = base[((long)y1 * S + x0) * C], v11 = base[((long)y1 * S + x1) * C];
            v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
        }
        out[i] = v;
    }
}
void k_paste_bg(const Launch& L, float* out, const float* img, const int* geom_dev, int B, int S, int C) {
    LAUNCH(paste_bg_kernel, grid1d((long)B * S * S * C), dim3(256), L, out, img, geom_dev, B, S, C);
}

// F.interpolate(mode='bilinear', align_corners=True), NHWC
__global__ void resize_bilinear_ac_kernel(float* __restrict__ out, const float* __restrict__ in, int N, int Hin, int Win,
                                          int Hout, int Wout, int C) {
    // every product / sum below is a SEPARATE fp32 rounding unless written as an explicit fma (HIP's __fmul_rn / __fsub_rn are plain
    // operators compiled in the header's own contraction mode: with them, or without this pragma, `fy - y0` still contracts into fma(sy, y, -y0) on the unrounded product)
#pragma clang fp contract(off)
    long total = (long)N * Hout * Wout * C;
    float sy = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
    float sx = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        int x = (int)(r % Wout);
        r /= Wout;
        int y = (int)(r % Hout);
        long n = r / Hout;
        // the source index is a ROUNDED fp32 product in torch (area_pixel_compute_source_index); left to the compiler, `fy - y0` contracts
        // into fma(sy, y, -y0) on the unrounded product and the weights move by up to an ulp of the index (7e-6 at 96 -> 112)
        float fy = sy * (float)y, fx = sx * (float)x;
        int y0 = min((int)fy, Hin - 1), x0 = min((int)fx, Win - 1);
        int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
        float ly = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
        const float* p = in + n * (long)Hin * Win * C + c;
        float v00 = p[((long)y0 * Win + x0) * C], v01 = p[((long)y0 * Win + x1) * C];
        float v10 = p[((long)y1 * Win + x0) * C], v11 = p[((long)y1 * Win + x1) * C];
        // torch's CPU upsample_bilinear2d (UpSampleKernel.cpp Interpolate<2>): out = t0 * w0; out += t1 * w1 per dimension, compiled
        // as fma(t0, w0, t1 * w1) — bit-exact against F.interpolate(align_corners=True) (tests/test_gpu_ops.py)
        const float lx0 = 1.f - lx, ly0 = 1.f - ly;
        const float top = __builtin_fmaf(v00, lx0, v01 * lx);
        const float bot = __builtin_fmaf(v10, lx0, v11 * lx);
        out[i] = __builtin_fmaf(top, ly0, bot * ly);
    }
}
void k_resize_bilinear_ac(const Launch& L, float* out, const float* in, int N, int Hin, int Win, int Hout, int Wout, int C) {
    LAUNCH(resize_bilinear_ac_kernel, grid1d((long)N * Hout * Wout * C), dim3(256), L, out, in, N, Hin, Win, Hout, Wout, C);
}

// ------------------------------------------------------------------ coefficient head (meta_net.py:266-289)
// F.normalize(x, dim=-1): y = x / max(||x||, 1e-12); one wave per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(float* __restrict__ y, float* __restrict__ norms,
                                                          const float* __restrict__ x, long rows, int D) {
    long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { float v = x[row * D + c]; s += v * v; }
    float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    if (lane == 0 && norms) norms[row] = nrm;
    for (int c = lane; c < D; c += 64) y[row * D + c] = x[row * D + c] / nrm;
}
void k_l2norm_fwd(const Launch& L, float* y, float* norms, const float* x, long rows, int D) {
    LAUNCH(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), L, y, norms, x, rows, D);
}
// dx = (dy - y * (y . dy)) / norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(float* __restrict__ dx, const float* __restrict__ y,
                                                          const float* __restrict__ norms, const float* __restrict__ dy, long rows,
                                                          int D) {
    long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += y[row * D + c] * dy[row * D + c];
    s = wave_sum(s);
    float inv = 1.f / norms[row];
    for (int c = lane; c < D; c += 64) dx[row * D + c] = (dy[row * D + c] - y[row * D + c] * s) * inv;
}
void k_l2norm_bwd(const Launch& L, float* dx, const float* y, const float* norms, const float* dy, long rows, int D) {
    LAUNCH(l2norm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), L, dx, y, norms, dy, rows, D);
}
// z[r,e,c] = sum_k coef[r,e,k] * basis[e,1+k,c] + basis[e,0,c]   (einsum 'behk,ekc->behc' with h=1)
// block = 64 columns c x 4 k-slices (one wave each, coalesced over c), partial sums combined through LDS
__global__ __launch_bounds__(256) void basis_fwd_kernel(float* __restrict__ z, const float* __restrict__ coef,
                                                         const float* __restrict__ basis, int R, int E, int Kc, int Cz) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long re = blockIdx.y;
    const int e = (int)(re % E);
    const float* bs = basis + (long)e * (Kc + 1) * Cz;
    const float* cf = coef + re * Kc;
    float acc = 0.f;
    if (c < Cz) {
        int k = w;
        for (; k + 12 < Kc; k += 16) {
            float b0 = bs[(long)(1 + k) * Cz + c], b1 = bs[(long)(5 + k) * Cz + c], b2 = bs[(long)(9 + k) * Cz + c],
                  b3 = bs[(long)(13 + k) * Cz + c];
            acc += cf[k] * b0; acc += cf[k + 4] * b1; acc += cf[k + 8] * b2; acc += cf[k + 12] * b3;
        }
        for (; k < Kc; k += 4) acc += cf[k] * bs[(long)(1 + k) * Cz + c];
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < Cz) z[re * Cz + c] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) + bs[c];
}
void k_basis_fwd(const Launch& L, float* z, const float* coef, const float* basis, int R, int E, int Kc, int Cz) {
    LAUNCH(basis_fwd_kernel, dim3(cdiv(Cz, 64), R * E), dim3(256), L, z, coef, basis, R, E, Kc, Cz);
}
// dcoef[r,e,k] = sum_c dz[r,e,c] * basis[e,1+k,c] : one wave per (r,e,k)
__global__ __launch_bounds__(256) void basis_bwd_kernel(float* __restrict__ dcoef, const float* __restrict__ dz,
                                                         const float* __restrict__ basis, int R, int E, int Kc, int Cz) {
    long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    long total = (long)R * E * Kc;
    if (w >= total) return;
    int lane = threadIdx.x & 63;
    int k = (int)(w % Kc);
    long re = w / Kc;
    int e = (int)(re % E);
    const float* bs = basis + ((long)e * (Kc + 1) + 1 + k) * Cz;
    const float* d = dz + re * Cz;
    float s = 0.f;
    for (int c = lane; c < Cz; c += 64) s += d[c] * bs[c];
    s = wave_sum(s);
    if (lane == 0) dcoef[w] = s;
}
void k_basis_bwd(const Launch& L, float* dcoef, const float* dz, const float* basis, int R, int E, int Kc, int Cz) {
    LAUNCH(basis_bwd_kernel, dim3(cdiv((long)R * E * Kc, 4)), dim3(256), L, dcoef, dz, basis, R, E, Kc, Cz);
}

// Row surgery (helpers.py:13-41 shift_tensor_dim0 + embedding_manager.py:347-360), one block per sample.
// map[b][o] = source row i >= 0 (copy in[b][i]) or -(1+j) (take z[b][j], j < reps).
// Up to CB_MAX_PLACEHOLDERS placeholder tokens per sample (embedding_manager.py:323-391: one / two / three persons in a prompt):
// `placeholders` is a device array [B][nph] (negative = unused); an occurrence of placeholder k takes rows z[b][k*reps + j].
__global__ __launch_bounds__(256) void row_surgery_fwd_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                               const float* __restrict__ z, const int64_t* __restrict__ tokens,
                                                               int32_t* __restrict__ map, int T, int C, int reps,
                                                               int64_t placeholder, const int64_t* __restrict__ placeholders, int nph) {
    // grid (B, slices): every block rebuilds the (tiny) row map from an LDS copy of the sample's tokens and copies its
    // slice of output rows; the serial part no longer waits on T dependent global loads
    __shared__ int smap[256];
    __shared__ unsigned char hit[256];          // 0 = ordinary token, 1 + k = occurrence of placeholder k
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const int64_t tk = tokens[(long)b * T + i];
        int h = 0;
        if (placeholders) {
            for (int k = nph - 1; k >= 0; --k) { const int64_t ph = placeholders[(long)b * nph + k]; if (ph >= 0 && tk == ph) h = 1 + k; }
        } else if (tk == placeholder) h = 1;
        hit[i] = (unsigned char)h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int cnt = 0;
        for (int i = 0; i < T; ++i) cnt += hit[i] != 0;
        int keep = T - cnt * (reps - 1);  // rows of the source that survive ("drop the last")
        for (int o = 0; o < T; ++o) smap[o] = o;  // untouched rows keep their own content
        int shift = 0;
        for (int i = 0; i < keep && i < T; ++i) {
            int tgt = i + shift;
            if (tgt < T) smap[tgt] = i;
            if (hit[i]) {
                const int k = hit[i] - 1;
                for (int j = 0; j < reps; ++j) if (tgt + j < T) smap[tgt + j] = -(1 + k * reps + j);
                shift += reps - 1;
            }
        }
    }
    __syncthreads();
    if (blockIdx.y == 0)
        for (int o = threadIdx.x; o < T; o += blockDim.x) map[(long)b * T + o] = smap[o];
    const int zrows = reps * (placeholders ? nph : 1);
    for (int o = blockIdx.y; o < T; o += gridDim.y) {
        const int s = smap[o];
        const float* src = (s >= 0) ? in + ((long)b * T + s) * C : z + ((long)b * zrows + (-s - 1)) * C;
        float* dst = out + ((long)b * T + o) * C;
        for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = src[c];
    }
}
void k_row_surgery_fwd(const Launch& L, float* out, const float* in, const float* z, const int64_t* tokens, int32_t* map, int B,
                       int T, int C, int reps, int64_t placeholder) {
    CB_CHECK(T <= 256, CB_ERR_SHAPE, "row surgery: T=%d > 256", T);
    LAUNCH(row_surgery_fwd_kernel, dim3(B, T < 16 ? T : 16), dim3(256), L, out, in, z, tokens, map, T, C, reps, placeholder,
           (const int64_t*)nullptr, 1);
}
void k_row_surgery_multi(const Launch& L, float* out, const float* in, const float* z, const int64_t* tokens, int32_t* map, int B,
                         int T, int C, int reps, const int64_t* placeholders_dev, int nph) {
    CB_CHECK(T <= 256, CB_ERR_SHAPE, "row surgery: T=%d > 256", T);
    CB_CHECK(nph >= 1 && nph <= 8 && nph * reps <= 120, CB_ERR_ARG, "row surgery: nph=%d reps=%d unsupported", nph, reps);
    LAUNCH(row_surgery_fwd_kernel, dim3(B, T < 16 ? T : 16), dim3(256), L, out, in, z, tokens, map, T, C, reps, (int64_t)-1,
           placeholders_dev, nph);
}
// dz[b][j][c] = sum over output rows o with map == -(1+j) of dout[b][o][c]
__global__ void row_surgery_bwd_kernel(float* __restrict__ dz, const float* __restrict__ dout, const int32_t* __restrict__ map,
                                       int B, int T, int C, int reps) {
    long total = (long)B * reps * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long bj = i / C;
        int j = (int)(bj % reps);
        long b = bj / reps;
        float acc = 0.f;
        for (int o = 0; o < T; ++o)
            if (map[b * T + o] == -(1 + j)) acc += dout[(b * T + o) * C + c];
        dz[i] = acc;
    }
}
void k_row_surgery_bwd(const Launch& L, float* dz, const float* dout, const int32_t* map, int B, int T, int C, int reps) {
    LAUNCH(row_surgery_bwd_kernel, grid1d((long)B * reps * C), dim3(256), L, dz, dout, map, B, T, C, reps);
}

// embedding_manager.py:484-489: dict[id] = m*dict[id] + (1-m)*pred, samples applied in batch order
__global__ void ema_update_kernel(float* __restrict__ id_emb, float* __restrict__ id_coef, const float* __restrict__ z,
                                  const float* __restrict__ coef, const int64_t* __restrict__ ids, int ids_stride, int B,
                                  int embN, int coefN, int max_ids, float m) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int b = 0; b < B; ++b) {
        int64_t id = ids[(long)b * ids_stride];
        if (id < 0 || id >= max_ids) continue;
        if (i < embN) id_emb[id * embN + i] = m * id_emb[id * embN + i] + (1.f - m) * z[(long)b * embN + i];
        if (i < coefN) id_coef[id * coefN + i] = m * id_coef[id * coefN + i] + (1.f - m) * coef[(long)b * coefN + i];
    }
}
void k_ema_update(const Launch& L, float* id_emb, float* id_coef, const float* z, const float* coef, const int64_t* ids,
                  int ids_stride, int B, int embN, int coefN, int max_ids, float m) {
    int n = embN > coefN ? embN : coefN;
    LAUNCH(ema_update_kernel, dim3(cdiv(n, 256)), dim3(256), L, id_emb, id_coef, z, coef, ids, ids_stride, B, embN, coefN, max_ids,
           m);
}

// use_rm_mlp (meta_net.py:173-175, 272-273): x = self.coef[id_idx] with id_idx = torch.flatten(id_multi) (meta_net.py:338-340), i.e.
// row r of the face-chunk-major batch (r = f*B + b) takes the r-th entry of the ROW-MAJOR [B][nface] id table, ids[r / nface][r % nface]
// — the reference's own pairing (it equals ids[b][f] for B == 1 and whenever all rows carry one identity).
__global__ void coef_gather_kernel(float* __restrict__ out, const float* __restrict__ coef, const int64_t* __restrict__ ids,
                                   int ids_stride, int B, int nface, int D, int max_ids) {
    long total = (long)nface * B * D;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % D);
        int r = (int)(i / D);
        int64_t id = ids[(long)(r / nface) * ids_stride + (r % nface)];
        out[i] = (id >= 0 && id < max_ids) ? coef[id * D + c] : 0.f;
    }
}
void k_coef_gather(const Launch& L, float* out, const float* coef, const int64_t* ids, int ids_stride, int B, int nface, int D, int max_ids) {
    LAUNCH(coef_gather_kernel, grid1d((long)nface * B * D), dim3(256), L, out, coef, ids, ids_stride, B, nface, D, max_ids);
}
// dense gradient of the gather (what autograd's index backward produces): one thread per (identity, column) sums its rows in a
// fixed order — deterministic, identities absent from the batch get an explicit zero
__global__ void coef_scatter_grad_kernel(float* __restrict__ g, const float* __restrict__ dout, const int64_t* __restrict__ ids,
                                         int ids_stride, int B, int nface, int D, int max_ids) {
    long total = (long)max_ids * D;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % D);
        int64_t id = i / D;
        float acc = 0.f;
        for (int r = 0; r < nface * B; ++r)
            if (ids[(long)(r / nface) * ids_stride + (r % nface)] == id) acc += dout[(long)r * D + c];
        g[i] = acc;
    }
}
void k_coef_scatter_grad(const Launch& L, float* gcoef, const float* dout, const int64_t* ids, int ids_stride, int B, int nface, int D,
                         int max_ids) {
    LAUNCH(coef_scatter_grad_kernel, grid1d((long)max_ids * D), dim3(256), L, gcoef, dout, ids, ids_stride, B, nface, D, max_ids);
}

// dW[o][i] = sum_r dpre[r][o] * v[r][i] ; db[o] = sum_r dpre[r][o]   (EqualLinear wgrad, meta_net.py:41-48)
__global__ void outer_grad_kernel(float* __restrict__ dW, float* __restrict__ db, const float* __restrict__ dpre,
                                  const float* __restrict__ v, int R, int O, int I) {
    long total = (long)O * I;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int o = (int)(idx / I), i = (int)(idx - (long)o * I);
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += dpre[(long)r * O + o] * v[(long)r * I + i];
        dW[idx] = acc;
        if (i == 0) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += dpre[(long)r * O + o];
            db[o] = s;
        }
    }
}
void k_outer_grad(const Launch& L, float* dW, float* db, const float* dpre, const float* v, int R, int O, int I) {
    LAUNCH(outer_grad_kernel, grid1d((long)O * I), dim3(256), L, dW, db, dpre, v, R, O, I);
}

// torch.optim.AdamW single-tensor update (ddpm.py:1454; defaults betas (0.9,0.999), eps 1e-8, wd 1e-2).
// The step counter lives on the device so the whole training step can be replayed from a hipGraph.
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                             long n, const int* __restrict__ step, float lr_arg, const float* __restrict__ lr_dev, float b1, float b2,
                             float eps, float wd) {
    const float lr = lr_dev ? *lr_dev : lr_arg;       // a captured step graph reads the learning rate from a device scalar: one graph serves every lr
    int t = *step;
    double bc1 = 1.0 - pow((double)b1, (double)t);
    double bc2 = 1.0 - pow((double)b2, (double)t);
    float step_size = (float)((double)lr / bc1);
    float bc2s = (float)sqrt(bc2);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);
        float mi = m[i] + (gi - m[i]) * (1.f - b1);   // exp_avg.lerp_(grad, 1-beta1)
        float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pi - step_size * (mi / denom);
        m[i] = mi; v[i] = vi;
    }
}
void k_adamw(const Launch& L, float* p, float* m, float* v, const float* g, long n, const int* step, float lr, float b1, float b2,
             float eps, float wd, const float* lr_dev) {
    LAUNCH(adamw_kernel, grid1d(n), dim3(256), L, p, m, v, g, n, step, lr, lr_dev, b1, b2, eps, wd);
}
__global__ void step_inc_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }
void k_step_inc(const Launch& L, int* step) { LAUNCH(step_inc_kernel, dim3(1), dim3(64), L, step); }

// y[m][n] = sum_k x[m][k] W[n][k] + bias[n], M <= 8: one wave per output column, weight-streaming bound
template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(T* __restrict__ y, const T* __restrict__ x, const T* __restrict__ W,
                                                    const float* __restrict__ bias, int M, int N, int K) {
    int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    int lane = threadIdx.x & 63;
    const T* w = W + (long)n * K;
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    if ((K & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            float4 wv = ld4(w + k);
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (m < M) {
                    float4 xv = ld4(x + (long)m * K + k);
                    acc[m] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
                }
        }
    } else {
        for (int k = lane; k < K; k += 64) {
            float wv = ld1(w + k);
#pragma unroll
            for (int m = 0; m < 8; ++m) if (m < M) acc[m] += wv * ld1(x + (long)m * K + k);
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
        if (m < M) {
            float s = wave_sum(acc[m]);
            if (lane == 0) st1(y + (long)m * N + n, s + (bias ? bias[n] : 0.f));
        }
}
void k_gemv(const Launch& L, DType dt, void* y, const void* x, const void* W, const float* bias, int M, int N, int K) {
    CB_CHECK(M >= 1 && M <= 8, CB_ERR_SHAPE, "gemv: M=%d must be in 1..8", M);
    if (L.stats) L.stats->gemm_flops += 2.0 * M * N * (double)K;
    CB_DT_SWITCH(dt, T, LAUNCH((gemv_kernel<T>), dim3(cdiv(N, 4)), dim3(256), L, (T*)y, (const T*)x, (const T*)W, bias, M, N, K));
}

// ------------------------------------------------------------------ one-time weight packing
// OIHW -> [O][ky][kx][I]  (B operand of the implicit GEMM, k = (ky*ks+kx)*I + ci)
template <typename T> __global__ void pack_conv_kernel(T* __restrict__ out, const float* __restrict__ w, int O, int I, int ks, int Ip) {
    long total = (long)O * Ip * ks * ks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int ci = (int)(i % Ip);
        long r = i / Ip;
        int kx = (int)(r % ks);
        r /= ks;
        int ky = (int)(r % ks);
        long o = r / ks;
        st1(out + i, ci < I ? w[((o * I + ci) * ks + ky) * ks + kx] : 0.f);
    }
}
void k_pack_conv(const Launch& L, DType dt, void* out, const float* w, int O, int I, int ks, int Ip) {
    CB_DT_SWITCH(dt, T, LAUNCH((pack_conv_kernel<T>), grid1d((long)O * Ip * ks * ks), dim3(256), L, (T*)out, w, O, I, ks, Ip));
}
// OIHW -> [I][ky'][kx'][O] with the kernel flipped: out[ci][ky'][kx'][co] = w[co][ci][ks-1-ky'][ks-1-kx']
template <typename T> __global__ void pack_conv_dgrad_kernel(T* __restrict__ out, const float* __restrict__ w, int O, int I, int ks, int Op) {
    long total = (long)Op * I * ks * ks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int co = (int)(i % Op);
        long r = i / Op;
        int kx = (int)(r % ks);
        r /= ks;
        int ky = (int)(r % ks);
        long ci = r / ks;
        st1(out + i, co < O ? w[(((long)co * I + ci) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)] : 0.f);
    }
}
void k_pack_conv_dgrad(const Launch& L, DType dt, void* out, const float* w, int O, int I, int ks, int Op) {
    CB_DT_SWITCH(dt, T, LAUNCH((pack_conv_dgrad_kernel<T>), grid1d((long)Op * I * ks * ks), dim3(256), L, (T*)out, w, O, I, ks, Op));
}
// Input gradient of "nearest x2 upsample + conv3x3" (openaimodel.py Upsample) as ONE 4x4 stride-2 pad-1 convolution over dy:
// output pixel i of the low-resolution input receives dy rows {2i-1, 2i, 2i+1, 2i+2} with the 3 kernel rows combined as
// {w2, w1+w2, w0+w1, w0} (same along x): 16 taps per input pixel instead of 4 positions x 9 taps, 2.25x fewer MACs, no
// full-resolution temporary and no pooling pass. OIHW -> [I][ty][tx][O] (O padded to Op): the B operand of that conv's GEMM.
template <typename T> __global__ void pack_conv_up_dgrad_kernel(T* __restrict__ out, const float* __restrict__ w, int O, int I, int Op) {
    long total = (long)Op * I * 16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int co = (int)(i % Op);
        long r = i / Op;
        int tx = (int)(r % 4);
        r /= 4;
        int ty = (int)(r % 4);
        long ci = r / 4;
        float v = 0.f;
        if (co < O) {
            const float* wk = w + ((long)co * I + ci) * 9;
            // taps of combined row t: t = 0 -> {2}, 1 -> {1, 2}, 2 -> {0, 1}, 3 -> {0}
            const int y0 = ty == 0 ? 2 : (ty == 1 ? 1 : 0), y1 = ty == 3 ? 0 : (ty == 2 ? 1 : 2);
            const int x0 = tx == 0 ? 2 : (tx == 1 ? 1 : 0), x1 = tx == 3 ? 0 : (tx == 2 ? 1 : 2);
            for (int ky = y0; ky <= y1; ++ky)
                for (int kx = x0; kx <= x1; ++kx) v += wk[ky * 3 + kx];
        }
        st1(out + i, v);
    }
}
void k_pack_conv_up_dgrad(const Launch& L, DType dt, void* out, const float* w, int O, int I, int Op) {
    CB_DT_SWITCH(dt, T, LAUNCH((pack_conv_up_dgrad_kernel<T>), grid1d((long)Op * I * 16), dim3(256), L, (T*)out, w, O, I, Op));
}
// fc over torch.flatten(NCHW): w[o][c*HW + hw] -> out[o][hw*C + c]   (iresnet.py:176-179)
__global__ void pack_fc_nhwc_kernel(float* __restrict__ out, const float* __restrict__ w, int O, int C, int HW) {
    long total = (long)O * C * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        int hw = (int)(r % HW);
        long o = r / HW;
        out[i] = w[(o * C + c) * HW + hw];
    }
}
void k_pack_fc_nhwc(const Launch& L, float* out, const float* w, int O, int C, int HW) {
    LAUNCH(pack_fc_nhwc_kernel, grid1d((long)O * C * HW), dim3(256), L, out, w, O, C, HW);
}
// fold an eval BatchNorm that FOLLOWS a bias-free conv/linear into its packed weights [O][per_o] and a bias
__global__ void fold_bn_kernel(float* __restrict__ wp, float* __restrict__ bias_out, const float* __restrict__ bw,
                               const float* __restrict__ bb, const float* __restrict__ bm, const float* __restrict__ bv, float eps,
                               int O, long per_o) {
    long total = (long)O * per_o;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long o = i / per_o;
        float s = bw[o] / sqrtf(bv[o] + eps);
        wp[i] *= s;
        if (i % per_o == 0) bias_out[o] = bb[o] - bm[o] * s;
    }
}
void k_fold_bn(const Launch& L, float* wp, float* bias_out, const float* bw, const float* bb, const float* bm, const float* bv,
               float eps, int O, long per_o) {
    LAUNCH(fold_bn_kernel, grid1d((long)O * per_o), dim3(256), L, wp, bias_out, bw, bb, bm, bv, eps, O, per_o);
}
__global__ void bn_to_affine_kernel(float* __restrict__ scale, float* __restrict__ shift, const float* __restrict__ w,
                                    const float* __restrict__ b, const float* __restrict__ m, const float* __restrict__ v, float eps,
                                    int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = w[c] / sqrtf(v[c] + eps);
    scale[c] = s;
    shift[c] = b[c] - m[c] * s;
}
void k_bn_to_affine(const Launch& L, float* scale, float* shift, const float* w, const float* b, const float* m, const float* v,
                    float eps, int C) {
    LAUNCH(bn_to_affine_kernel, dim3(cdiv(C, 256)), dim3(256), L, scale, shift, w, b, m, v, eps, C);
}

}  // namespace cb
