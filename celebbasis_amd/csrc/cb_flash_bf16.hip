// cb_flash_bf16.hip — fused (flash-style) multi-head attention forward + backward with bf16 operands on the gfx950 matrix cores
// (v_mfma_f32_16x16x32_bf16, fp32 accumulation; scores, softmax statistics and the probabilities' normalisation are fp32).
// BASELINE config 3 twin of cb_flash.hip; replaces the materialised softmax(QK^T)V of ldm/modules/attention.py:170-193 for
// head dims d <= 160 (UNet self/cross attention d = 40/80/160, CLIP causal attention d = 64).
//
// Same transposed mapping as the fp32 kernels, re-tiled for the 16x16x32 instruction (A lane (i = lane&15, g = lane>>4) holds
// A[i][8g..8g+7], B lane (j, g) holds B[8g..8g+7][j], C/D: col = lane&15, row = 4*(lane>>4) + reg):
//   S^T[key][query] = K . Q^T     A = 16-byte chunks of a row-major K tile in LDS, B = Q rows in registers (loaded once).
//   A query's scores of one 16-key sub-tile sit in its lane quartet {j, j+16, j+32, j+48}: softmax needs two xor-shuffles.
//   O^T[d][query] += V^T . P^T    B = the probabilities AS THEY LIE in the S^T accumulators: lane (query, g) holds keys
//   4g..4g+3 of sub-tile 2G and of sub-tile 2G+1 — taken as the 8 k-slots of its MFMA k-group. Any assignment of keys to
//   k-slots is a valid contraction as long as A uses the same one, so the V tile is written to LDS TRANSPOSED ([d][key]) with
//   the keys of a 32-key group permuted to pos = g*8 + sub*4 + e: the A operand is then one ds_read_b128.
//   The transpose happens in registers on the way global -> LDS: a thread owns a 4-key x 8-d block (four 16-byte loads),
//   v_perm_b32 pairs the 16-bit halves, eight ds_write_b64 store it.
// Backward = dQ kernel (block owns 64 queries, streams K/V; needs K row-major and transposed) + dK/dV kernel (block owns 64
// keys, streams Q/dO in both layouts); both recompute P from the saved log-sum-exp; no atomics, deterministic.
// LDS images: rows of 16-byte chunks with an odd chunk pitch, chunk index XOR ((row>>2 ^ row>>3) & 1): the 16-lane groups of
// ds_read_b128 ({0-3,12-15,20-27}, ...) then hit 16 distinct bank quads for both the row-major and the transposed read.
#include "cb_device.h"

namespace cb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

constexpr int FB_ROWS = 64;      // rows (queries or keys) owned by a block: 4 waves x 16
constexpr int FB_OOB = -16;

__device__ __forceinline__ float fb_quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float fb_quad_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fb_srd(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}

template <int KS, int TKV> struct FbCfg {
    static constexpr int TK = TKV;                  // rows of a streamed tile
    static constexpr int DP = 32 * KS;              // head dim padded to the MFMA k depth
    static constexpr int CPD = DP / 8;              // 16-byte chunks per row-major row
    static constexpr int CPR = CPD | 1;             // odd chunk pitch of the row-major image
    static constexpr int CPT = TK / 8 + 1;          // odd chunk pitch of the transposed image (TK positions per row)
    static constexpr int NT = TK / 16;              // 16-row sub-tiles
    static constexpr int NG = TK / 32;              // 32-row groups (PV-style contraction depth)
    static constexpr int RBYTES = TK * CPR * 16;    // row-major image
    static constexpr int TBYTES = DP * CPT * 16;    // transposed image: DP rows (head-dim index)
    static constexpr int NU = ((TK / 4) * CPD + 255) / 256;   // 4-row x 8-column units per thread
};

template <int KS, int TK> using FbRegs = float4[FbCfg<KS, TK>::NU][4];     // a thread's 4-row x 8-column units of one tile

// ---- streamed tiles: global -> registers (4-row x 8-col units) -> LDS (row-major and / or transposed image) ---------------
template <int KS, int TK> struct FbStream {
    using C = FbCfg<KS, TK>;
    int goff[C::NU];     // (4*kq*ld + 8*dc) * 2 bytes, FB_OOB when the unit is unused
    int row4[C::NU];     // 4*kq
    int lo_r[C::NU];     // row-major image: byte offset of row 4*kq, chunk dc
    int lo_ta[C::NU];    // transposed image: byte offset of row 8*dc (+0..3), position block of rows 4*kq..4*kq+3
    int lo_tb[C::NU];    // same for rows 8*dc+4 .. 8*dc+7 (the chunk swizzle bit flips at 4-row granularity)
    __device__ __forceinline__ void init(long ld, int d) {
        const int dp8 = d >> 3;
#pragma unroll
        for (int j = 0; j < C::NU; ++j) {
            const int u = threadIdx.x + j * 256;
            const int kq = u / dp8, dc = u - kq * dp8;
            const bool used = u < (TK / 4) * dp8;
            row4[j] = 4 * kq;
            goff[j] = used ? (int)(((long)4 * kq * ld + 8 * dc) * 2) : FB_OOB;
            const int gsw = (kq ^ (kq >> 1)) & 1;                                   // swizzle bit of rows 4kq..4kq+3
            lo_r[j] = (4 * kq * C::CPR + (dc ^ gsw)) * 16;
            const int G = kq >> 3, sub = (kq >> 2) & 1, qq = kq & 3;                // 32-row group, sub-tile, k-group
            const int ch = G * 4 + qq;
            const int base = 8 * dc * C::CPT * 16 + sub * 8;
            lo_ta[j] = base + (ch ^ (dc & 1)) * 16;                                 // rows 8dc+0..3: bit = dc & 1
            lo_tb[j] = base + (ch ^ ((dc & 1) ^ 1)) * 16;                           // rows 8dc+4..7: bit flipped
        }
    }
};
template <int KS, int TK>
__device__ __forceinline__ void fb_prefetch(FbRegs<KS, TK>& reg, __amdgpu_buffer_rsrc_t srd, const FbStream<KS, TK>& st,
                                            int row0, int limit, long ld) {
    const int base = (int)((long)row0 * ld * 2);          // wave-uniform
    const int ldb = (int)(ld * 2);
#pragma unroll
    for (int j = 0; j < FbCfg<KS, TK>::NU; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = (st.goff[j] != FB_OOB) && (row0 + st.row4[j] + e < limit);
            u32x4b v = __builtin_amdgcn_raw_buffer_load_b128(srd, ok ? st.goff[j] + base + e * ldb : FB_OOB, 0, 0);
            reg[j][e] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
}
template <int KS, int TK>
__device__ __forceinline__ void fb_store_rows(char* img, const FbRegs<KS, TK>& reg, const FbStream<KS, TK>& st) {
#pragma unroll
    for (int j = 0; j < FbCfg<KS, TK>::NU; ++j)
        if (st.goff[j] != FB_OOB) {
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<float4*>(img + st.lo_r[j] + e * FbCfg<KS, TK>::CPR * 16) = reg[j][e];
        }
}
// 4 rows x 8 columns of bf16 -> 8 rows x 4 positions: dword pairs built with v_perm_b32
template <int KS, int TK>
__device__ __forceinline__ void fb_store_tr(char* img, const FbRegs<KS, TK>& reg, const FbStream<KS, TK>& st) {
    constexpr int pitch = FbCfg<KS, TK>::CPT * 16;
#pragma unroll
    for (int j = 0; j < FbCfg<KS, TK>::NU; ++j)
        if (st.goff[j] != FB_OOB) {
            const unsigned r0[4] = {__float_as_uint(reg[j][0].x), __float_as_uint(reg[j][0].y), __float_as_uint(reg[j][0].z), __float_as_uint(reg[j][0].w)};
            const unsigned r1[4] = {__float_as_uint(reg[j][1].x), __float_as_uint(reg[j][1].y), __float_as_uint(reg[j][1].z), __float_as_uint(reg[j][1].w)};
            const unsigned r2[4] = {__float_as_uint(reg[j][2].x), __float_as_uint(reg[j][2].y), __float_as_uint(reg[j][2].z), __float_as_uint(reg[j][2].w)};
            const unsigned r3[4] = {__float_as_uint(reg[j][3].x), __float_as_uint(reg[j][3].y), __float_as_uint(reg[j][3].z), __float_as_uint(reg[j][3].w)};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned sel = (i & 1) ? 0x07060302u : 0x05040100u;      // high / low 16-bit halves of (b, a)
                uint2 v;
                v.x = __builtin_amdgcn_perm(r1[i >> 1], r0[i >> 1], sel);     // rows 4kq+0, 4kq+1 at column 8dc+i
                v.y = __builtin_amdgcn_perm(r3[i >> 1], r2[i >> 1], sel);     // rows 4kq+2, 4kq+3
                *reinterpret_cast<uint2*>(img + (i < 4 ? st.lo_ta[j] : st.lo_tb[j]) + i * pitch) = v;
            }
        }
}
__device__ __forceinline__ void fb_zero_lds(char* lds, int bytes) {
    for (int i = threadIdx.x * 16; i < bytes; i += 256 * 16) *reinterpret_cast<float4*>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// per-lane register copy of one row as MFMA B operand: reg[s] = src[row][32s + 8g .. +7], zero beyond d / limit
template <int KS>
__device__ __forceinline__ void fb_load_rowregs(bf16x8_t (&reg)[KS], const bf16_t* __restrict__ src, long ld, int row, int limit, int d, int g) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * g;
        u32x4b v = {0u, 0u, 0u, 0u};
        if (row < limit && c < d) v = *reinterpret_cast<const u32x4b*>(src + (long)row * ld + c);
        reg[s] = __builtin_bit_cast(bf16x8_t, v);
    }
}
__device__ __forceinline__ bf16x8_t fb_pack8(const f32x4& a, const f32x4& b) {
    bf16x8_t r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    return r;
}
// lane-invariant part of an MFMA A-operand read: row (lane & 15) of a 16-row block, chunk (4*grp + g) with the image swizzle
__device__ __forceinline__ int fb_lane_off(int l15, int g, int pitch_chunks) {
    const int gsw = ((l15 >> 2) ^ (l15 >> 3)) & 1;
    return (l15 * pitch_chunks + (g ^ gsw)) * 16;
}
__device__ __forceinline__ void fb_store_out4(bf16_t* p, const f32x4& v, float beta) {
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (beta != 0.f) { const float4 old = ld4(p); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
    st4(p, o);
}

// ---------------------------------------------------------------------------------------------- forward
template <int KS, int TK, int DTV>
__global__ __launch_bounds__(256) void flash_bf16_fwd_kernel(const FlashArgs a) {
    using C = FbCfg<KS, TK>;
    constexpr int DT = DTV;                          // 16-wide output tiles over the head dim: ceil(d / 16), not the padded 2 * KS
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Ks = lds;                                  // row-major K tile
    char* Vt = lds + C::RBYTES;                      // transposed V tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * FB_ROWS;
    const int qrow = q0 + wave * 16 + l15;
    const bf16_t* Q = static_cast<const bf16_t*>(a.q) + b * a.sqb + h * a.sqh;
    const bf16_t* K = static_cast<const bf16_t*>(a.k) + b * a.skb + h * a.skh;
    const bf16_t* V = static_cast<const bf16_t*>(a.v) + b * a.skb + h * a.skh;

    bf16x8_t qreg[KS];
    fb_load_rowregs<KS>(qreg, Q, a.ldq, qrow, a.Nq, a.d, g);
    const float scale2 = a.scale * 1.4426950408889634f;          // scores are kept in the log2 domain
    const int ndt = (a.d + 15) >> 4;                             // output tiles that hold real columns (wave-uniform)
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    int kend = a.Nk;
    if (a.causal) kend = min(a.Nk, q0 + FB_ROWS);      // keys beyond the last query of the block are fully masked
    const __amdgpu_buffer_rsrc_t srdK = fb_srd(K), srdV = fb_srd(V);
    FbStream<KS, TK> st; st.init(a.ldk, a.d);
    FbRegs<KS, TK> pk, pv;
    fb_prefetch<KS, TK>(pk, srdK, st, 0, a.Nk, a.ldk);
    fb_prefetch<KS, TK>(pv, srdV, st, 0, a.Nk, a.ldk);
    fb_zero_lds(lds, C::RBYTES + C::TBYTES);           // head-dim padding chunks of K must read as zeros
    const int offK = fb_lane_off(l15, g, C::CPR), offV = fb_lane_off(l15, g, C::CPT);
    for (int k0 = 0; k0 < kend; k0 += TK) {
        __syncthreads();
        fb_store_rows<KS, TK>(Ks, pk, st);
        fb_store_tr<KS, TK>(Vt, pv, st);
        __syncthreads();
        if (k0 + TK < kend) {
            fb_prefetch<KS, TK>(pk, srdK, st, k0 + TK, a.Nk, a.ldk);
            fb_prefetch<KS, TK>(pv, srdV, st, k0 + TK, a.Nk, a.ldk);
        }
        f32x4 s[C::NT];
#pragma unroll
        for (int kt = 0; kt < C::NT; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < C::NT; ++kt) {
                const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(Ks + kt * 16 * C::CPR * 16 + ks * 64 + offK);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, qreg[ks], s[kt], 0, 0, 0);
            }
        const bool need_mask = a.causal || k0 + TK > a.Nk;                     // wave-uniform
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[kt][r] * scale2;
                if (need_mask) {
                    const int key = k0 + kt * 16 + g * 4 + r;
                    const bool ok = key < a.Nk && (!a.causal || key <= qrow);
                    v = ok ? v : -INFINITY;
                }
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fb_quad_max(mx);
        const float mnew = fmaxf(m, mx);
        const float muse = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = __builtin_amdgcn_exp2f(m - muse);            // m = -inf -> 0
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kt][r] - muse);
                s[kt][r] = p;
                ps += p;
            }
        l = l * alpha + fb_quad_sum(ps);
        m = mnew;
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t] *= alpha;
        // O^T += V^T P^T : one MFMA per (32-key group, 16 output columns)
#pragma unroll
        for (int G = 0; G < C::NG; ++G) {
            const bf16x8_t pb = fb_pack8(s[2 * G], s[2 * G + 1]);
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (t < ndt) {
                    const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(Vt + t * 16 * C::CPT * 16 + G * 64 + offV);
                    o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb, o[t], 0, 0, 0);
                }
        }
    }
    if (qrow < a.Nq) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        bf16_t* O = static_cast<bf16_t*>(a.o) + b * a.sob + h * a.soh + (long)qrow * a.ldo;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int c = t * 16 + g * 4;
            if (c < a.d) st4(O + c, make_float4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv));
        }
        if (g == 0 && a.lse) a.lse[(long)bh * a.Nq + qrow] = m + __builtin_amdgcn_logf(l);   // log2 domain
    }
}

// D[bh][q] = sum_d dO[q][d] * O[q][d] : one wave per query row (only when no dQ kernel runs)
__global__ __launch_bounds__(256) void flash_bf16_dvec_kernel(const FlashArgs a, int BH) {
    long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (w >= (long)BH * a.Nq) return;
    int lane = threadIdx.x & 63;
    int bh = (int)(w / a.Nq), qi = (int)(w - (long)bh * a.Nq), b = bh / a.H, h = bh - b * a.H;
    const bf16_t* O = static_cast<const bf16_t*>(a.o) + b * a.sob + h * a.soh + (long)qi * a.ldo;
    const bf16_t* dO = static_cast<const bf16_t*>(a.dout) + b * a.sob + h * a.soh + (long)qi * a.ldo;
    float s = 0.f;
    for (int c = lane; c < a.d; c += 64) s += (float)O[c] * (float)dO[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) a.dvec[w] = s;
}

// ---------------------------------------------------------------------------------------------- backward: dQ
template <int KS, int TK, int DTV>
__global__ __launch_bounds__(256) void flash_bf16_dq_kernel(const FlashArgs a) {
    using C = FbCfg<KS, TK>;
    constexpr int DT = DTV;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Ks = lds;                                  // row-major K
    char* Vs = lds + C::RBYTES;                      // row-major V
    char* Kt = lds + 2 * C::RBYTES;                  // transposed K
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * FB_ROWS;
    const int qrow = q0 + wave * 16 + l15;
    const bf16_t* Q = static_cast<const bf16_t*>(a.q) + b * a.sqb + h * a.sqh;
    const bf16_t* K = static_cast<const bf16_t*>(a.k) + b * a.skb + h * a.skh;
    const bf16_t* V = static_cast<const bf16_t*>(a.v) + b * a.skb + h * a.skh;
    const bf16_t* dO = static_cast<const bf16_t*>(a.dout) + b * a.sob + h * a.soh;

    bf16x8_t qreg[KS], dreg[KS];
    fb_load_rowregs<KS>(qreg, Q, a.ldq, qrow, a.Nq, a.d, g);
    fb_load_rowregs<KS>(dreg, dO, a.ldo, qrow, a.Nq, a.d, g);
    const float scale2 = a.scale * 1.4426950408889634f;
    const float lse = qrow < a.Nq ? a.lse[(long)bh * a.Nq + qrow] : 0.f;
    const int ndt = (a.d + 15) >> 4;
    float Dq = 0.f;     // D = rowsum(dO * O), computed here and published for the dK/dV kernel that runs next
    {
        bf16x8_t oreg[KS];
        fb_load_rowregs<KS>(oreg, static_cast<const bf16_t*>(a.o) + b * a.sob + h * a.soh, a.ldo, qrow, a.Nq, a.d, g);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
            for (int e = 0; e < 8; ++e) Dq += (float)oreg[s_][e] * (float)dreg[s_][e];
        Dq = fb_quad_sum(Dq);
        if (g == 0 && qrow < a.Nq) a.dvec[(long)bh * a.Nq + qrow] = Dq;
    }
    f32x4 gq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) gq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kend = a.Nk;
    if (a.causal) kend = min(a.Nk, q0 + FB_ROWS);
    const __amdgpu_buffer_rsrc_t srdK = fb_srd(K), srdV = fb_srd(V);
    FbStream<KS, TK> st; st.init(a.ldk, a.d);
    FbRegs<KS, TK> pk, pv;
    fb_prefetch<KS, TK>(pk, srdK, st, 0, a.Nk, a.ldk);
    fb_prefetch<KS, TK>(pv, srdV, st, 0, a.Nk, a.ldk);
    fb_zero_lds(lds, 2 * C::RBYTES + C::TBYTES);
    const int offR = fb_lane_off(l15, g, C::CPR), offT = fb_lane_off(l15, g, C::CPT);
    for (int k0 = 0; k0 < kend; k0 += TK) {
        __syncthreads();
        fb_store_rows<KS, TK>(Ks, pk, st);
        fb_store_tr<KS, TK>(Kt, pk, st);
        fb_store_rows<KS, TK>(Vs, pv, st);
        __syncthreads();
        if (k0 + TK < kend) {
            fb_prefetch<KS, TK>(pk, srdK, st, k0 + TK, a.Nk, a.ldk);
            fb_prefetch<KS, TK>(pv, srdV, st, k0 + TK, a.Nk, a.ldk);
        }
        f32x4 s[C::NT], dp[C::NT];
#pragma unroll
        for (int kt = 0; kt < C::NT; ++kt) { s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < C::NT; ++kt) {
                const int off = kt * 16 * C::CPR * 16 + ks * 64 + offR;
                const bf16x8_t ak = *reinterpret_cast<const bf16x8_t*>(Ks + off);
                const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(Vs + off);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, qreg[ks], s[kt], 0, 0, 0);      // S^T  = K Q^T
                dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, dreg[ks], dp[kt], 0, 0, 0);    // dP^T = V dO^T
            }
        const bool need_mask = a.causal || k0 + TK > a.Nk;          // wave-uniform
#pragma unroll
        for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s[kt][r] * scale2 - lse);
                if (need_mask) {
                    const int key = k0 + kt * 16 + g * 4 + r;
                    const bool ok = key < a.Nk && (!a.causal || key <= qrow);
                    p = ok ? p : 0.f;
                }
                s[kt][r] = p * (dp[kt][r] - Dq) * a.scale;                                        // d(q.k)
            }
        // dQ^T += K^T dS^T
#pragma unroll
        for (int G = 0; G < C::NG; ++G) {
            const bf16x8_t sb = fb_pack8(s[2 * G], s[2 * G + 1]);
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (t < ndt) {
                    const bf16x8_t ak = *reinterpret_cast<const bf16x8_t*>(Kt + t * 16 * C::CPT * 16 + G * 64 + offT);
                    gq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, sb, gq[t], 0, 0, 0);
                }
        }
    }
    if (qrow < a.Nq) {
        bf16_t* Gp = static_cast<bf16_t*>(a.dq) + b * a.sqb + h * a.sqh + (long)qrow * a.ldq;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int c = t * 16 + g * 4;
            if (c < a.d) fb_store_out4(Gp + c, gq[t], a.beta_q);
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward: dK, dV
// block owns 64 keys (wave: 16 keys, key = lane&15 is the MFMA column); streams Q / dO tiles.
//   S[q][key]    = Q K^T         A = Q tile (row-major), B = K regs (unscaled; scale applied after)
//   dP[q][key]   = dO V^T        A = dO tile (row-major), B = V regs
//   dV^T[d][key] += dO^T P       A = dO tile (transposed), B = P regs
//   dK^T[d][key] += Q^T dS       A = Q tile (transposed),  B = dS regs
template <int KS, int TK, int DTV>
__global__ __launch_bounds__(256) void flash_bf16_dkv_kernel(const FlashArgs a) {
    using C = FbCfg<KS, TK>;
    constexpr int DT = DTV;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Qs = lds;
    char* Ds = lds + C::RBYTES;
    char* Qt = lds + 2 * C::RBYTES;
    char* Dt = lds + 2 * C::RBYTES + C::TBYTES;
    float* Ls = reinterpret_cast<float*>(lds + 2 * C::RBYTES + 2 * C::TBYTES);          // lse[TK], D[TK]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int k0 = blockIdx.x * FB_ROWS;
    const int krow = k0 + wave * 16 + l15;
    const bf16_t* Q = static_cast<const bf16_t*>(a.q) + b * a.sqb + h * a.sqh;
    const bf16_t* K = static_cast<const bf16_t*>(a.k) + b * a.skb + h * a.skh;
    const bf16_t* V = static_cast<const bf16_t*>(a.v) + b * a.skb + h * a.skh;
    const bf16_t* dO = static_cast<const bf16_t*>(a.dout) + b * a.sob + h * a.soh;

    bf16x8_t kreg[KS], vreg[KS];
    fb_load_rowregs<KS>(kreg, K, a.ldk, krow, a.Nk, a.d, g);
    fb_load_rowregs<KS>(vreg, V, a.ldk, krow, a.Nk, a.d, g);
    const int ndt = (a.d + 15) >> 4;
    f32x4 gk[DT], gv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { gk[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; gv[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // cross-attention has few keys and many queries: blockIdx.z splits the query range, partial dK/dV go to a workspace
    const int qchunk = ((a.Nq + a.nsplit - 1) / a.nsplit + TK - 1) / TK * TK;
    int qbeg = blockIdx.z * qchunk;
    const int qlim = min(a.Nq, qbeg + qchunk);
    if (a.causal) qbeg = max(qbeg, (k0 / TK) * TK);   // queries before the first key of the block see none of its keys
    const __amdgpu_buffer_rsrc_t srdQ = fb_srd(Q), srdD = fb_srd(dO);
    FbStream<KS, TK> stq, sto; stq.init(a.ldq, a.d); sto.init(a.ldo, a.d);
    const float scale2 = a.scale * 1.4426950408889634f;
    FbRegs<KS, TK> pq, pd;
    fb_prefetch<KS, TK>(pq, srdQ, stq, qbeg, a.Nq, a.ldq);
    fb_prefetch<KS, TK>(pd, srdD, sto, qbeg, a.Nq, a.ldo);
    fb_zero_lds(lds, 2 * C::RBYTES + 2 * C::TBYTES);
    const int offR = fb_lane_off(l15, g, C::CPR), offT = fb_lane_off(l15, g, C::CPT);
    for (int q0 = qbeg; q0 < qlim; q0 += TK) {
        __syncthreads();
        fb_store_rows<KS, TK>(Qs, pq, stq);
        fb_store_tr<KS, TK>(Qt, pq, stq);
        fb_store_rows<KS, TK>(Ds, pd, sto);
        fb_store_tr<KS, TK>(Dt, pd, sto);
        if (threadIdx.x < TK) {
            const int qi = q0 + threadIdx.x;
            Ls[threadIdx.x] = qi < a.Nq ? a.lse[(long)bh * a.Nq + qi] : 0.f;
            Ls[TK + threadIdx.x] = qi < a.Nq ? a.dvec[(long)bh * a.Nq + qi] : 0.f;
        }
        __syncthreads();
        if (q0 + TK < qlim) {
            fb_prefetch<KS, TK>(pq, srdQ, stq, q0 + TK, a.Nq, a.ldq);
            fb_prefetch<KS, TK>(pd, srdD, sto, q0 + TK, a.Nq, a.ldo);
        }
        f32x4 s[C::NT], dp[C::NT];
#pragma unroll
        for (int qt = 0; qt < C::NT; ++qt) { s[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int qt = 0; qt < C::NT; ++qt) {
                const int off = qt * 16 * C::CPR * 16 + ks * 64 + offR;
                const bf16x8_t aq = *reinterpret_cast<const bf16x8_t*>(Qs + off);
                const bf16x8_t ad = *reinterpret_cast<const bf16x8_t*>(Ds + off);
                s[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, kreg[ks], s[qt], 0, 0, 0);
                dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, vreg[ks], dp[qt], 0, 0, 0);
            }
        // rows of the C layout are queries 4*g + r of sub-tile qt, the column is this lane's key
        const bool need_mask = a.causal || q0 + TK > a.Nq || k0 + FB_ROWS > a.Nk;   // wave-uniform
#pragma unroll
        for (int qt = 0; qt < C::NT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = qt * 16 + g * 4 + r, qi = q0 + ql;
                float p = __builtin_amdgcn_exp2f(s[qt][r] * scale2 - Ls[ql]);
                if (need_mask) {
                    const bool ok = qi < a.Nq && krow < a.Nk && (!a.causal || krow <= qi);
                    p = ok ? p : 0.f;
                }
                const float ds = p * (dp[qt][r] - Ls[TK + ql]) * a.scale;
                s[qt][r] = p;
                dp[qt][r] = ds;
            }
#pragma unroll
        for (int G = 0; G < C::NG; ++G) {
            const bf16x8_t pb = fb_pack8(s[2 * G], s[2 * G + 1]);
            const bf16x8_t sb = fb_pack8(dp[2 * G], dp[2 * G + 1]);
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (t < ndt) {
                    const int off = t * 16 * C::CPT * 16 + G * 64 + offT;
                    const bf16x8_t ad = *reinterpret_cast<const bf16x8_t*>(Dt + off);
                    const bf16x8_t aq = *reinterpret_cast<const bf16x8_t*>(Qt + off);
                    gv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, pb, gv[t], 0, 0, 0);
                    gk[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, sb, gk[t], 0, 0, 0);
                }
        }
    }
    if (a.nsplit > 1) {
        if (krow < a.Nk) {
            const long slab = (long)gridDim.y * a.Nk * a.d;
            float* WK = a.ws + ((long)blockIdx.z * gridDim.y + bh) * a.Nk * a.d + (long)krow * a.d;
            float* WV = WK + (long)a.nsplit * slab;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int c = t * 16 + g * 4;
                if (c < a.d) {
                    *reinterpret_cast<float4*>(WK + c) = make_float4(gk[t][0], gk[t][1], gk[t][2], gk[t][3]);
                    *reinterpret_cast<float4*>(WV + c) = make_float4(gv[t][0], gv[t][1], gv[t][2], gv[t][3]);
                }
            }
        }
        return;
    }
    if (krow < a.Nk) {
        bf16_t* GK = static_cast<bf16_t*>(a.dk) + b * a.skb + h * a.skh + (long)krow * a.ldk;
        bf16_t* GV = static_cast<bf16_t*>(a.dv) + b * a.skb + h * a.skh + (long)krow * a.ldk;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int c = t * 16 + g * 4;
            if (c < a.d) { fb_store_out4(GK + c, gk[t], a.beta_kv); fb_store_out4(GV + c, gv[t], a.beta_kv); }
        }
    }
}

// sum the query-split partials: one thread per (bh, key, 4 head-dim columns), dK and dV
__global__ __launch_bounds__(256) void flash_bf16_dkv_reduce_kernel(const FlashArgs a, int BH) {
    const int dq4 = a.d / 4;
    long total = (long)BH * a.Nk * dq4;
    const long slab = (long)BH * a.Nk * a.d;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c = (int)(i % dq4) * 4;
        long r = i / dq4;
        int key = (int)(r % a.Nk);
        int bh = (int)(r / a.Nk), b = bh / a.H, h = bh - b * a.H;
        const float* wk = a.ws + ((long)bh * a.Nk + key) * a.d + c;
        const float* wv = wk + (long)a.nsplit * slab;
        f32x4 sk = {0.f, 0.f, 0.f, 0.f}, sv = sk;
        for (int sp = 0; sp < a.nsplit; ++sp) {
            float4 x = *reinterpret_cast<const float4*>(wk + sp * slab), y = *reinterpret_cast<const float4*>(wv + sp * slab);
            sk[0] += x.x; sk[1] += x.y; sk[2] += x.z; sk[3] += x.w;
            sv[0] += y.x; sv[1] += y.y; sv[2] += y.z; sv[3] += y.w;
        }
        bf16_t* GK = static_cast<bf16_t*>(a.dk) + b * a.skb + h * a.skh + (long)key * a.ldk + c;
        bf16_t* GV = static_cast<bf16_t*>(a.dv) + b * a.skb + h * a.skh + (long)key * a.ldk + c;
        fb_store_out4(GK, sk, a.beta_kv);
        fb_store_out4(GV, sv, a.beta_kv);
    }
}

// ---------------------------------------------------------------------------------------------- host
static int fb_ks(int d) { return (d + 31) / 32; }
static int fb_tk(int ks) { return ks <= 3 ? 64 : 32; }

size_t flash_bf16_bwd_ws_floats(const FlashArgs& a, int B, int* nsplit) {
    const int BH = B * a.H;
    long blocks0 = (long)cdiv(a.Nk, FB_ROWS) * BH;
    int ns = 1;
    if (blocks0 < 256 && a.Nq >= 128 && !a.causal) {     // same rule as the fp32 kernel (cb_flash.hip): aim at ~2 blocks per CU
        ns = (int)((512 + blocks0 - 1) / blocks0);
        int maxs = a.Nq / 64;                       // at least one 64-query tile (two 32-query tiles) per split
        if (ns > maxs) ns = maxs;
        if (ns < 1) ns = 1;
    }
    *nsplit = ns;
    return ns > 1 ? (size_t)2 * ns * BH * a.Nk * a.d : 0;
}

bool flash_bf16_supported(int d, const FlashArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool strides = (a.ldq % 8 == 0) && (a.ldk % 8 == 0) && (a.ldo % 8 == 0) && (a.sqb % 8 == 0) && (a.sqh % 8 == 0) &&
                   (a.skb % 8 == 0) && (a.skh % 8 == 0) && (a.sob % 8 == 0) && (a.soh % 8 == 0);
    const long lim = (1L << 31) - 64;
    bool small = (long)a.Nq * a.ldq * 2 < lim && (long)a.Nk * a.ldk * 2 < lim && (long)a.Nq * a.ldo * 2 < lim;   // 32-bit buffer offsets
    return d >= 8 && d <= 160 && d % 8 == 0 && strides && small && al(a.q) && al(a.k) && al(a.v) && al(a.o);
}

template <int KS, int TK> static size_t fb_lds(int kind) {     // 0 fwd, 1 dq, 2 dkv
    using C = FbCfg<KS, TK>;
    if (kind == 0) return (size_t)C::RBYTES + C::TBYTES;
    if (kind == 1) return (size_t)2 * C::RBYTES + C::TBYTES;
    return (size_t)2 * C::RBYTES + 2 * C::TBYTES + 2 * TK * sizeof(float);
}

// KS = 32-wide k groups of the head dim (contraction depth of S / dP), DT = 16-wide output tiles actually needed (d = 40: 3, not 4;
// d = 80: 5, not 6): the accumulate phases (P.V, dS.K, dS^T.Q, P^T.dO) skip the padding tiles
#define FB_GO(KSV, TKV, DTV, KERNEL, KIND, GRID, ...) \
    hipLaunchKernelGGL((KERNEL<KSV, TKV, DTV>), GRID, dim3(256), (fb_lds<KSV, TKV>(KIND)), L.stream, __VA_ARGS__)
#define FB_DISPATCH(KSV, KERNEL, KIND, GRID, ...)                                                        \
    switch (KSV) {                                                                                       \
        case 1: FB_GO(1, 64, 2, KERNEL, KIND, GRID, __VA_ARGS__); break;                                 \
        case 2: if (a.d <= 48) FB_GO(2, 64, 3, KERNEL, KIND, GRID, __VA_ARGS__); else FB_GO(2, 64, 4, KERNEL, KIND, GRID, __VA_ARGS__); break; \
        case 3: if (a.d <= 80) FB_GO(3, 64, 5, KERNEL, KIND, GRID, __VA_ARGS__); else FB_GO(3, 64, 6, KERNEL, KIND, GRID, __VA_ARGS__); break; \
        case 4: FB_GO(4, 32, 8, KERNEL, KIND, GRID, __VA_ARGS__); break;                                 \
        default: FB_GO(5, 32, 10, KERNEL, KIND, GRID, __VA_ARGS__); break;                               \
    }

void flash_bf16_forward(const Launch& L, const FlashArgs& a, int B) {
    const int BH = B * a.H, ks = fb_ks(a.d);
    if (L.stats) { L.stats->gemm_flops += 4.0 * BH * (double)a.Nq * a.Nk * a.d; L.stats->gemm_launches++; }
    L.count(a.tws ? 2 : 1);
    if (L.dry) return;
    dim3 grid(cdiv(a.Nq, FB_ROWS), BH);
    ProfRec rec; bool prof = L.prof && L.prof->on;
    if (prof) {
        rec.variant = L.prof->id_of(a.tws ? "flash_bf16_v2_fwd_kernel" : "flash_bf16_fwd_kernel"); rec.flops = 4.0 * BH * (double)a.Nq * a.Nk * a.d;
        CB_HIP(hipEventCreate(&rec.e0)); CB_HIP(hipEventCreate(&rec.e1)); CB_HIP(hipEventRecord(rec.e0, L.stream));
    }
    if (a.tws) flash_bf16_v2_forward(L, a, B, a.tws);
    else FB_DISPATCH(ks, flash_bf16_fwd_kernel, 0, grid, a)
    CB_HIP(hipGetLastError());
    if (prof) { CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec); }
}

void flash_bf16_backward(const Launch& L, const FlashArgs& a, int B, bool need_dq, bool need_dkv) {
    const int BH = B * a.H, ks = fb_ks(a.d);
    // algorithmic (reference-equivalent) work: dP, dQ | dK, dV — the recomputation of S inside both kernels is not counted
    double fl = 2.0 + (need_dq ? 2.0 : 0.0) + (need_dkv ? 4.0 : 0.0);
    if (L.stats) { L.stats->gemm_flops += fl * BH * (double)a.Nq * a.Nk * a.d; L.stats->gemm_launches++; }
    L.count(a.tws ? 2 + (need_dkv ? 3 : 0) : 1 + (need_dkv ? (a.nsplit > 1 ? 2 : 1) : 0));
    if (L.dry) return;
    ProfRec rec; bool prof = L.prof && L.prof->on;
    if (prof) {
        rec.variant = L.prof->id_of(a.tws ? "flash_bf16_v2_bwd_kernels" : "flash_bf16_bwd_kernels"); rec.flops = fl * BH * (double)a.Nq * a.Nk * a.d;
        CB_HIP(hipEventCreate(&rec.e0)); CB_HIP(hipEventCreate(&rec.e1)); CB_HIP(hipEventRecord(rec.e0, L.stream));
    }
    if (a.tws) {
        flash_bf16_v2_backward(L, a, B, need_dq, need_dkv, a.tws);
        CB_HIP(hipGetLastError());
        if (prof) { CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec); }
        return;
    }
    if (!need_dq) hipLaunchKernelGGL(flash_bf16_dvec_kernel, dim3(cdiv((long)BH * a.Nq, 4)), dim3(256), 0, L.stream, a, BH);
    if (need_dq) {
        dim3 grid(cdiv(a.Nq, FB_ROWS), BH);
        FB_DISPATCH(ks, flash_bf16_dq_kernel, 1, grid, a)
    }
    if (need_dkv) {
        CB_CHECK(a.nsplit == 1 || a.ws, CB_ERR_STATE, "flash backward: query-split workspace missing");
        dim3 grid(cdiv(a.Nk, FB_ROWS), BH, a.nsplit);
        FB_DISPATCH(ks, flash_bf16_dkv_kernel, 2, grid, a)
        if (a.nsplit > 1) {
            long total = (long)BH * a.Nk * (a.d / 4);
            hipLaunchKernelGGL(flash_bf16_dkv_reduce_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, L.stream, a, BH);
        }
    }
    CB_HIP(hipGetLastError());
    if (prof) { CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec); }
}

}  // namespace cb
