// cb_gemm_tiles.h — pieces shared by the fp32 (cb_gemm.hip) and bf16 (cb_gemm_glds.hip) MFMA GEMM / implicit-GEMM kernels:
// kernel arguments, the branch-free 16-byte buffer loaders for k-contiguous operands, the swizzled LDS store, the XCD-aware
// block -> tile remap, the split-K reduce and the tile / split planner. Everything is templated on the element size ESZ
// (bytes): a 16-byte chunk holds EPC = 16 / ESZ elements of k and a k-tile is 4 chunks per row (64 bytes), so the thread ->
// (row, chunk) mapping, the LDS image and its XOR swizzle are the same for both element types.
#pragma once
#include "cb_device.h"
#include <type_traits>
#include <unordered_map>

namespace cb {

// tap index -> kernel row for ks = 1 .. 4 (3: (t * 11) >> 5 == t / 3 for t < 9)
__host__ __device__ __forceinline__ int conv_tap_row(int tap, int ks) {
    return ks == 3 ? ((tap * 11) >> 5) : (ks == 4 ? (tap >> 2) : (ks == 2 ? (tap >> 1) : 0));
}

struct KArgs {
    GemmArgs a;
    int tiles_m, tiles_n;
    int splitk, k_per_split;
    float* ws;          // split-K partials [split][batch][M][N] fp32 (null when splitk == 1)
    int a_vec, b_vec;   // 1 when the operand may use 16-byte global loads
    int safe = 0;       // LDS-DMA kernel: 1 = drain the DMA queue (vmcnt(0)) + full barrier per k-tile instead of the counted wait
    int* cnt = nullptr; // LDS-DMA kernels, splitk > 1: arrival counters [batch][tile] (zero between launches) -> the tile's LAST block folds the
                        // partials in the same launch (splitk_arrive below); nullptr = partials only, splitk_reduce_kernel follows
};

// ---- branch-free 16-byte loaders: buffer loads with hardware range checking ---------------------------------------------
// Preconditions checked on the host: 16-byte aligned bases/strides, every operand slice < 2 GiB, K a multiple of the chunk
// (ROW/CONV), so a chunk is either fully inside or fully outside the operand. Each operand gets a raw buffer descriptor
// (wave-uniform, num_records = 2 GiB); a lane whose chunk is out of range (row/k tail, conv padding, transposed stride-2
// holes) issues the load at byte offset 0xFFFFFFF0, which the hardware answers with zeros. Consequences:
//  * no divergent branch and no select sits between the global loads and their consumers, so the prefetch of tile t+1 is
//    issued before the MFMA block of tile t and waited for only at the LDS stores;
//  * address math is 32-bit and the zero-fill costs no VALU (the fp32 MFMA shares issue with the vector ALU: every VALU
//    instruction removed from the k loop is MFMA time won — PMC in profiles/r01_pmc_conv.md).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = -16;   // 0xFFFFFFF0 >= num_records -> zeros
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t srd, int byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// ---- in-launch split-K reduction (cdna_hip_programming.md §6 Guideline 16, form R1; MI355X_MICROARCH.md §Workgroup dispatch: per-XCD
// L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores). Every block of a tile writes its fp32 partial slab
// WRITE-THROUGH (`sc1` buffer stores: the bytes leave the XCD's L2, so no release fence — the agent-scope release of the first version
// of this code, one `buffer_wbl2` per block, cost 10 us per launch under load: 39.5 against 36.9 ms per step). Publish: every storing
// wave drains its stores (asm wait: the compiler does not count on it), barrier, ONE lane draws the ticket (relaxed, agent scope).
// The block that draws splitk - 1 is the reducer: ONE agent-scope acquire (drops this CU's L1), counter back to zero for the next
// launch, barrier, and then every thread reads ALL slabs (its own included) in split order: the sum does not depend on which block
// arrived last — deterministic, and the same order as splitk_reduce_kernel. `flag` is one int of the block's (idle) tile ring: no
// second __shared__ object (that would de-pipeline the k loop, guide §5 "three .s-level traps"). Returns true in the reducer block.
__device__ __forceinline__ bool splitk_arrive(int* cnt, int splitk, int* flag) { return grid_arrive_last(cnt, splitk, flag); }
// 8 consecutive fp32 partials of one output row -> the block's slab, write-through. `srd` = the slab, `off` = byte offset in it.
__device__ __forceinline__ void splitk_store8(__amdgpu_buffer_rsrc_t srd, float* slab, int off, const float (&v)[8], bool v4, int nvalid) {
    if (v4) {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, srd, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, srd, off + 16, 0, 16);
    } else {
        float* w = reinterpret_cast<float*>(reinterpret_cast<char*>(slab) + off);
        for (int e = 0; e < 8 && e < nvalid; ++e) __hip_atomic_store(w + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// sum of the splitk partial values of 8 consecutive outputs, split order (v4: 16-byte aligned pieces)
__device__ __forceinline__ void splitk_fold8(float (&v)[8], const float* __restrict__ w0, long slab_stride, int splitk, bool v4, int nvalid) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (v4) {
#pragma unroll 4
        for (int sp = 0; sp < splitk; ++sp) {
            const float* w = w0 + sp * slab_stride;
            const float4 x0 = *reinterpret_cast<const float4*>(w), x1 = *reinterpret_cast<const float4*>(w + 4);
            v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
        }
    } else {
        for (int sp = 0; sp < splitk; ++sp) {
            const float* w = w0 + sp * slab_stride;
            for (int e = 0; e < 8 && e < nvalid; ++e) v[e] += w[e];
        }
    }
}

// ---- epilogue pieces shared by the LDS-DMA kernels (cb_gemm_glds.hip, cb_gemm_bf16_p8.hip): 8 consecutive outputs of one row
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8_t*>(p) = o;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// one 8-column piece of an output row: v = alpha*acc + bias, + rowbias / residual / beta*C, rounded once to T. `vec`: the piece lies
// inside N; the 16-byte path needs aligned C / rowbias / R pieces, everything else goes element by element.
template <typename T>
__device__ __forceinline__ void glds_finish_piece(const GemmArgs& a, float (&v)[8], const float (&bcol)[8], T* __restrict__ Cp, const T* __restrict__ Rp,
                                                  const T* __restrict__ rowbias, long ldrb, int row, int col, bool vec) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a.alpha * v[e] + bcol[e];
    T* c = Cp + (long)row * a.ldc + col;
    const T* rb = rowbias ? rowbias + (long)(row / a.rows_per_img) * ldrb + col : nullptr;
    const T* rr = Rp ? Rp + (long)row * a.ldr + col : nullptr;
    if (vec && al16(c) && (!rb || al16(rb)) && (!rr || al16(rr))) {
        float t[8];
        if (rb) { ld8(rb, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e]; }
        if (rr) { ld8(rr, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e]; }
        if (a.beta != 0.f) { ld8(c, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a.beta * t[e]; }
        if (a.prelu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : a.prelu[col + e] * v[e]; }
        st8(c, v);
        if (a.C2) {                                                   // second output: the next layer's per-channel affine of this one
            T* c2 = static_cast<T*>(a.C2) + (long)row * a.ldc + col;
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = a.c2_scale[col + e] * v[e] + a.c2_shift[col + e];
            if (al16(c2)) st8(c2, w);
            else for (int e = 0; e < 8; ++e) c2[e] = (T)w[e];
        }
    } else {
        for (int e = 0; e < 8 && col + e < a.N; ++e) {
            float x = v[e];
            if (rb) x += (float)rb[e];
            if (rr) x += (float)rr[e];
            if (a.beta != 0.f) x += a.beta * (float)c[e];
            if (a.prelu) x = x > 0.f ? x : a.prelu[col + e] * x;
            c[e] = (T)x;
            if (a.C2) (static_cast<T*>(a.C2) + (long)row * a.ldc + col)[e] = (T)(a.c2_scale[col + e] * x + a.c2_shift[col + e]);
        }
    }
}
// ROW operand: element (r, k) at byte (r*ld + k)*ESZ. Per-slot row offsets are loop invariant.
template <int ROWS, int ESZ>
struct RowOffs {
    int off[ROWS / 64];     // (r*ld + chunk*EPC)*ESZ or OOB
    __device__ __forceinline__ void init(long ld, int r0, int R, int tid) {
#pragma unroll
        for (int i = 0; i < ROWS / 64; ++i) {
            int r = r0 + ((tid + i * 256) >> 2);
            off[i] = (r < R) ? (int)(((long)r * ld + (tid & 3) * (16 / ESZ)) * ESZ) : OOB;
        }
    }
};
template <int ROWS, int ESZ>
__device__ __forceinline__ void offs_row(int (&off)[ROWS / 64], const RowOffs<ROWS, ESZ>& ro, int k0, int Kend, int tid) {
    const bool kok = (k0 + (tid & 3) * (16 / ESZ)) < Kend;
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) off[i] = (kok && ro.off[i] != OOB) ? ro.off[i] + k0 * ESZ : OOB;
}

// Per-thread conv gather state: the rows a thread loads are fixed for the whole k loop, so the row -> (image, oy, ox)
// decode (three integer divisions) is hoisted out of it; the k -> (tap, ci) decode advances incrementally by one k-tile.
template <int ROWS, int ESZ>
struct ConvRows {
    static constexpr int EPC = 16 / ESZ, BKE = 4 * EPC;
    int nb[ROWS / 64];    // n*H*W (pixel offset of the image), or -1 when the row is out of range
    int vy0[ROWS / 64];   // oy*stride - pad
    int vx0[ROWS / 64];   // ox*stride - pad
    int tap, ci;          // decode of this thread's first k of the current tile
    __device__ __forceinline__ void init(const ConvGeom& g, int r0, int R, int kfirst, int tid) {
        const int HWo = g.Ho * g.Wo;
#pragma unroll
        for (int i = 0; i < ROWS / 64; ++i) {
            int r = r0 + ((tid + i * 256) >> 2);
            if (r < R) {
                int n = r / HWo;
                int rem = r - n * HWo;
                int oy = rem / g.Wo, ox = rem - oy * g.Wo;
                nb[i] = n * g.H * g.W; vy0[i] = oy * g.stride - g.pad; vx0[i] = ox * g.stride - g.pad;
            } else { nb[i] = -1; vy0[i] = 0; vx0[i] = 0; }
        }
        int k = kfirst + (tid & 3) * EPC;
        tap = k / g.C; ci = k - tap * g.C;
    }
    __device__ __forceinline__ void advance(const ConvGeom& g) {
        ci += BKE;
        if (g.C >= BKE) { bool w = ci >= g.C; ci -= w ? g.C : 0; tap += w ? 1 : 0; }   // wave-uniform, one wrap at most
        else while (ci >= g.C) { ci -= g.C; ++tap; }
    }
};
template <int ROWS, int ESZ>
__device__ __forceinline__ void offs_conv(int (&off)[ROWS / 64], const ConvGeom& g, const ConvRows<ROWS, ESZ>& cr, int k0, int Kend, int tid) {
    const int k = k0 + (tid & 3) * (16 / ESZ);
    const int sh = (g.up | g.tr);
    const int Hv = g.H << sh, Wv = g.W << sh;
    const int ky = (g.ks == 3) ? ((cr.tap * 11) >> 5) : 0;
    const int kx = cr.tap - ky * g.ks;
    const bool kok = k < Kend;
    const int c4 = g.C * ESZ, ci4 = cr.ci * ESZ;
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        int vy = cr.vy0[i] + ky, vx = cr.vx0[i] + kx;
        bool ok = kok && (cr.nb[i] >= 0) && ((unsigned)vy < (unsigned)Hv) && ((unsigned)vx < (unsigned)Wv);
        ok = ok && !(g.tr && ((vy | vx) & 1));
        int pix = cr.nb[i] + (vy >> sh) * g.W + (vx >> sh);
        off[i] = ok ? pix * c4 + ci4 : OOB;
    }
}

// Fast conv gather (plain 1x1 / 3x3, stride 1 or 2, no upsample / transposed holes, C a multiple of the k-tile): a k-tile
// then lies inside ONE tap for the whole workgroup, so the tap decode lives in scalar registers (uniform k0) and each lane
// only needs    byte offset = base_off[slot] + delta(tap, ci)     valid = bit `tap` of a per-slot 9-bit mask
// i.e. 4 VALU per 16-byte load instead of ~25 (bounds compares, shifts, 2 multiplies).
template <int ROWS, int ESZ>
struct ConvFast {
    int base[ROWS / 64];      // ((n*H + oy*stride - pad)*W + ox*stride - pad)*C*ESZ + chunk*16   (may be "negative": only used when valid)
    unsigned vmask[ROWS / 64];
    __device__ __forceinline__ void init(const ConvGeom& g, int r0, int R, int tid) {
        const int HWo = g.Ho * g.Wo;
#pragma unroll
        for (int i = 0; i < ROWS / 64; ++i) {
            int r = r0 + ((tid + i * 256) >> 2);
            unsigned m = 0; int b = 0;
            if (r < R) {
                int n = r / HWo;
                int rem = r - n * HWo;
                int oy = rem / g.Wo, ox = rem - oy * g.Wo;
                int vy0 = oy * g.stride - g.pad, vx0 = ox * g.stride - g.pad;
                for (int ky = 0; ky < g.ks; ++ky)
                    for (int kx = 0; kx < g.ks; ++kx)
                        if ((unsigned)(vy0 + ky) < (unsigned)g.H && (unsigned)(vx0 + kx) < (unsigned)g.W) m |= 1u << (ky * g.ks + kx);
                b = (((n * g.H + vy0) * g.W + vx0) * g.C + (tid & 3) * (16 / ESZ)) * ESZ;
            }
            vmask[i] = m; base[i] = b;
        }
    }
};
template <int ROWS, int ESZ>
__device__ __forceinline__ void offs_conv_fast(int (&off)[ROWS / 64], const ConvFast<ROWS, ESZ>& cf, int tap_u, int delta_u) {
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        bool ok = (cf.vmask[i] >> tap_u) & 1u;
        off[i] = ok ? cf.base[i] + delta_u : OOB;
    }
}

// k-contiguous operand: LDS layout [row][4 chunks of 16 bytes] without padding, chunk c of row r stored at chunk
// c ^ ((r >> 2) & 3). Both the store (a 16-lane group = 4 rows x 4 chunks) and the MFMA fragment read (a 16-lane group =
// 16 rows x 1 chunk) then touch 16 distinct 4-bank groups: conflict-free ds_write_b128 / ds_read_b128 with loop-invariant
// per-lane addresses (no address arithmetic left in the k loop). `buf` is the tile base in floats (16 per row).
template <int ROWS>
__device__ __forceinline__ void store_kc(float* buf, const float4 (&reg)[ROWS / 64], int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        int idx = tid + i * 256;
        int r = idx >> 2;
        int c = (idx & 3) ^ ((r >> 2) & 3);
        *reinterpret_cast<float4*>(buf + r * 16 + c * 4) = reg[i];
    }
}

__device__ __forceinline__ int xcd_remap(int b, int n) {
    // bijective "XCD x gets a contiguous chunk" remap (guide §5: XCD swizzle must be bijective)
    int q = n >> 3, r = n & 7;
    int xcd = b & 7, j = b >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + j;
}

// split-K reduce + epilogue: one thread per output element (T = element type of C, R and rowbias)
template <typename T>
__global__ void splitk_reduce_kernel(const KArgs p) {
    const GemmArgs& a = p.a;
    long MN = (long)a.M * a.N;
    long total = MN * a.batch;
    const T* Rb = static_cast<const T*>(a.R);
    const T* rowbias = static_cast<const T*>(a.rowbias);
    T* Cb = static_cast<T*>(a.C);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int z = (int)(i / MN);
        long rem = i - (long)z * MN;
        int row = (int)(rem / a.N), col = (int)(rem - (long)row * a.N);
        float s = 0.f;
        for (int sp = 0; sp < p.splitk; ++sp) s += p.ws[((long)sp * a.batch + z) * MN + rem];
        int zb = z / a.nh, zh = z - zb * a.nh;
        float v = a.alpha * s;
        if (a.bias) v += a.bias[col];
        if (rowbias) v += ld1(rowbias + (long)(row / a.rows_per_img) * (a.ld_rowbias ? a.ld_rowbias : a.N) + col);
        if (Rb) v += ld1(Rb + zb * a.sRb + zh * a.sRh + (long)row * a.ldr + col);
        T* c = Cb + zb * a.sCb + zh * a.sCh + (long)row * a.ldc + col;
        if (a.beta != 0.f) v += a.beta * ld1(c);
        if (a.prelu) v = v > 0.f ? v : a.prelu[col] * v;
        st1(c, v);
        if (a.C2) st1(static_cast<T*>(a.C2) + zb * a.sCb + zh * a.sCh + (long)row * a.ldc + col, a.c2_scale[col] * v + a.c2_shift[col]);
    }
}

// ---------------------------------------------------------------- host side: tile / split-K planner
struct Plan { int BM, BN, tiles_m, tiles_n, splitk, k_per_split; };

// Planner constants (unit = one MAC on one CU at the fp32 MFMA rate), each overridable from the environment for sweeps on the
// GPU box (profiles/README.md, runs 26-29):
//   CB_SPLITK_COST     fixed cost of the extra reduce launch            (0.25e6 ~ 1 us; 1e6 was 1.2 % slower per step)
//   CB_PLAN_TRAFFIC    cost per (2*split+1) partial element moved
//   CB_PLAN_BLOCK_OVH  per-block prologue/epilogue, in k-tiles
//   CB_PLAN_MINK       fewest k-tiles a split may get
//   CB_PLAN_EFF        relative MFMA efficiency of the 128x128,128x64,64x128,64x64 tiles
//   CB_PLAN_BF16_SPEED how many times faster than the fp32 kernel the bf16 kernel retires MACs (scales the MFMA term only)
struct PlanKnobs { double splitk_cost, traffic, block_ovh; int mink; double eff[4]; double bf16_speed; double eff80 = 0.70; };
inline const PlanKnobs& plan_knobs() {
    static const PlanKnobs k = [] {
        PlanKnobs k{0.25e6, 0.41, 0.0, 4, {1.0, 0.80, 0.80, 0.58}, 6.0};
        if (const char* e = getenv("CB_SPLITK_COST")) k.splitk_cost = atof(e);
        if (const char* e = getenv("CB_PLAN_TRAFFIC")) k.traffic = atof(e);
        if (const char* e = getenv("CB_PLAN_BLOCK_OVH")) k.block_ovh = atof(e);
        if (const char* e = getenv("CB_PLAN_MINK")) k.mink = atoi(e) > 0 ? atoi(e) : 1;
        if (const char* e = getenv("CB_PLAN_EFF")) sscanf(e, "%lf,%lf,%lf,%lf", &k.eff[0], &k.eff[1], &k.eff[2], &k.eff[3]);
        if (const char* e = getenv("CB_PLAN_BF16_SPEED")) k.bf16_speed = atof(e) > 0 ? atof(e) : 6.0;
        if (const char* e = getenv("CB_PLAN_EFF80")) k.eff80 = atof(e) > 0 ? atof(e) : 0.70;
        return k;
    }();
    return k;
}
// ---- measured plan table. The analytic model below is within 5-8 % of the per-shape optimum over the step's launches and no
// setting of its constants closes the gap (profiles/README.md, round 2), so the shapes of the SD v1-4 step (config 2 at bs = 1,
// config 3 at bs = 4, config 5 at bs = 2, the DDIM batches) carry MEASURED plans: cb_plan_table.inc is generated by
// tools/make_plan_table.py from an offline autotune run (CB_AUTOTUNE=<file>: every launch of the LDS-DMA kernel times all
// (tile, split-K) candidates on its own operands into scratch outputs and records the winner; the run itself keeps executing
// the analytic plans). Unknown shapes fall back to the model. CB_PLAN_TABLE=0 disables the table, CB_PLAN_FILE=<file> adds one.
struct PlanEntry { int BM, BN, split; };
inline std::string plan_key(const GemmArgs& a, int bke) {
    char b[160];
    const int cf = a.a_kind == LD_CONV ? (a.g.up | (a.g.tr << 1) | (a.g.stride << 2) | (a.g.ks << 5)) : 0;
    snprintf(b, sizeof b, "%d|%d|%d|%d|%d|%d|%d|%d", bke, (int)a.dt, (int)a.a_kind, a.M, a.N, a.K, a.batch, cf);
    return b;
}
inline void plan_table_parse(std::unordered_map<std::string, PlanEntry>& t, const char* text) {
    // lines: "<key> <BM> <BN> <split> [anything]"
    const char* p = text;
    while (*p) {
        const char* e = strchr(p, '\n');
        std::string line(p, e ? (size_t)(e - p) : strlen(p));
        char key[160]; int bm, bn, sp;
        const bool ok = sscanf(line.c_str(), "%159s %d %d %d", key, &bm, &bn, &sp) == 4 && sp >= 1;
        const bool pow2 = ((bm == 64 || bm == 128 || bm == 256) && (bn == 64 || bn == 128) && (bm != 256 || bn == 128)) ||
                          (bm == 256 && bn == 256) || (bm == 512 && bn == 128);   // the phase-pipelined bf16 tiles (cb_gemm_bf16_p8.hip)
        const bool n80 = bn == 80 && (bm == 16 || bm == 32 || bm == 64);          // fp32 80-column tiles (cb_gemm_glds.hip)
        if (ok && (pow2 || n80))
            t[key] = PlanEntry{bm, bn, sp};
        if (!e) break;
        p = e + 1;
    }
}
inline const std::unordered_map<std::string, PlanEntry>& plan_table() {
    static const std::unordered_map<std::string, PlanEntry> t = [] {
        std::unordered_map<std::string, PlanEntry> m;
        const char* off = getenv("CB_PLAN_TABLE");
        if (off && off[0] == '0') return m;
        static const char* kEmbedded =
#include "cb_plan_table.inc"
            ;
        plan_table_parse(m, kEmbedded);
        if (const char* f = getenv("CB_PLAN_FILE")) {
            if (FILE* fp = fopen(f, "r")) {
                std::string all; char buf[4096]; size_t n;
                while ((n = fread(buf, 1, sizeof buf, fp)) > 0) all.append(buf, n);
                fclose(fp);
                plan_table_parse(m, all.c_str());
            }
        }
        return m;
    }();
    return t;
}

// diagnostics (tools/tune_gemm.py): restrict the planner to one tile shape / split count; 0 = free. Not thread safe, not for production.
struct PlanForce { int BM = 0, BN = 0, split = 0; int glds_safe = 0; int splitk_inlaunch = 1;
                   long gen = 0;   /* bumped by every cb_debug_* call that changes it: captured step graphs carry the value they were made under */ };
inline PlanForce& plan_force() {
    static PlanForce f = [] {
        PlanForce g;
        const char* e = getenv("CB_GLDS_SAFE"); g.glds_safe = (e && e[0] == '1') ? 1 : 0;
        const char* s = getenv("CB_SPLITK_INLAUNCH"); g.splitk_inlaunch = (s && s[0] == '0') ? 0 : 1;     // cb_debug_splitk_inlaunch
        return g;
    }();
    return f;
}
// the phase-pipelined bf16 tiles, 256 x 256 and 512 x 128 (cb_gemm_bf16_p8.hip): eligibility of a problem, launch (bn selects the tile)
bool gemm_bf16_p8_ok(const GemmArgs& a);
void gemm_bf16_p8_launch(int bn, dim3 grid, hipStream_t s, const KArgs& ka);
inline bool is_p8_tile(int BM, int BN) { return (BM == 256 && BN == 256) || (BM == 512 && BN == 128); }

// bke = elements of k per tile (16 for fp32, 32 for bf16); speed = MFMA rate relative to the fp32 kernel
// n80: the caller can run the fp32 80-column tiles (64 / 32 / 16 rows x 80 columns, k-groups inside the block) on this problem
// p8: the caller can run the phase-pipelined bf16 tiles on this problem (a table entry / pinned plan naming one is ignored otherwise)
inline Plan make_plan(const GemmArgs& a, int bke, double speed, bool n80 = false, bool p8 = false) {
    // Pick the (tile, split-K) pair with the smallest modelled time. Unit = one MAC on one CU at the fp32 MFMA rate.
    // Measured on MI355X (profiles/r01): 128x128 tiles ~78 TF, 64x64 tiles ~45 TF on the same conv shapes, so small tiles
    // are only worth it when the big ones cannot fill the 256 CUs even with split-K.
    // the fifth shape (8-wave 256x128 tile of the bf16 LDS-DMA kernel) is never chosen by the model: only a measured table entry
    // or cb_debug_force_plan selects it
    static const int tiles[10][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {256, 128}, {64, 80}, {32, 80}, {16, 80}, {256, 256}, {512, 128}};
    const PlanKnobs& kn = plan_knobs();
    const double eff[10] = {kn.eff[0], kn.eff[1], kn.eff[2], kn.eff[3], 1.0, kn.eff80, kn.eff80, kn.eff80, 1.0, 1.0};
    static const int splits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    const int ksteps = cdiv(a.K, bke);
    double best = 1e300;
    Plan pl{128, 128, 1, 1, 1, ksteps * bke};
    PlanForce pf = plan_force();
    if (pf.BN == 80 && !n80) { const PlanForce keep = pf; pf = PlanForce{}; pf.glds_safe = keep.glds_safe; pf.splitk_inlaunch = keep.splitk_inlaunch; pf.gen = keep.gen; }   // a pinned 80-column tile only binds the problems that can run it
    if (is_p8_tile(pf.BM, pf.BN) && !p8) { const PlanForce keep = pf; pf = PlanForce{}; pf.glds_safe = keep.glds_safe; pf.splitk_inlaunch = keep.splitk_inlaunch; pf.gen = keep.gen; }   // likewise the 256 x 256 bf16 tile
    if (!pf.BM && !pf.split) {
        const auto& tab = plan_table();
        if (!tab.empty()) {
            auto it = tab.find(plan_key(a, bke));
            if (it != tab.end() && it->second.split <= ksteps && (it->second.BN != 80 || n80) && (!is_p8_tile(it->second.BM, it->second.BN) || p8)) {
                const PlanEntry& e = it->second;
                const int kps = cdiv(ksteps, e.split) * bke;
                return Plan{e.BM, e.BN, (int)cdiv(a.M, e.BM), (int)cdiv(a.N, e.BN), (int)cdiv(a.K, kps), kps};
            }
        }
    }
    for (int t = 0; t < 10; ++t) {
        const int BM = tiles[t][0], BN = tiles[t][1];
        if ((t == 4 || t >= 8) && pf.BM != BM) continue;             // the 8-wave tiles: never by model, only pinned (or from the table, above)
        if (t >= 8 && !p8) continue;
        if (t > 4 && t < 8 && !n80) continue;
        if (pf.BM && (pf.BM != BM || pf.BN != BN)) continue;
        if (!pf.BM && BM == 128 && a.M <= 64) continue;
        if (!pf.BM && BN == 128 && a.N <= 64) continue;
        const long tm = cdiv(a.M, BM), tn = cdiv(a.N, BN);
        for (int sp : splits) {
            if (pf.split && sp != pf.split) continue;
            if (!pf.split && sp > 1 && ksteps / sp < kn.mink) break;      // keep >= mink k-tiles per split
            int kps = cdiv(ksteps, sp) * bke;
            int split = cdiv(a.K, kps);
            double blocks = (double)tm * tn * a.batch * split;
            double waves = blocks / 256.0;
            if (blocks <= 1024) waves = (double)cdiv((long)blocks, 256);
            double tcost = waves * (double)BM * BN * (kps + kn.block_ovh * bke) / eff[t] / speed;
            // split-K traffic: partial writes + reduce reads/writes ~ (2*split+1)*M*N*4 B at ~3 TB/s, plus one more launch
            if (split > 1) tcost += kn.splitk_cost + kn.traffic * (2.0 * split + 1.0) * (double)a.M * a.N * a.batch;
            if (tcost < best) { best = tcost; pl = Plan{BM, BN, (int)tm, (int)tn, split, kps}; }
        }
    }
    return pl;
}

}  // namespace cb
