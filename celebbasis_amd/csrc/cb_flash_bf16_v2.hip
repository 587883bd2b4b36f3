// cb_flash_bf16_v2.hip — second-generation fused attention for the bf16 step (BASELINE config 3), head dims 40 .. 80: the shape
// that carries 85 % of the UNet's attention FLOPs (the five 64x64-level self-attention layers: N = 4096, d = 40, 8 heads;
// ldm/modules/attention.py:170-193) and the 32x32 level (N = 1024, d = 80). Same arithmetic as cb_flash_bf16.hip (S^T = K Q^T in 16x16x32 bf16 MFMA tiles, log2-domain
// online softmax in fp32, P consumed from the accumulators as the next MFMA's B operand), re-built around what bounded it:
//
//   * LDS traffic, not MFMA, set the pace: a block of 4 waves x 16 rows re-staged every 64-row K / V tile through VGPRs
//     (ds_write_b128 + v_perm + ds_write_b64: ~510 LDS-array cycles per tile) and every wave re-read the whole tile for its 16
//     rows (~220 cycles) against 224 cycles of MFMA per SIMD. Here a wave owns 16*QB rows (QB = 2 .. 4 row blocks held in
//     registers) and every LDS fragment it reads feeds QB MFMAs.
//   * No register staging: every streamed tile is fetched by `buffer_load_dwordx4 ... lds` (LDS-DMA) into a two-stage ring —
//     one counted wait + ONE barrier per tile, no ds_write, no v_perm, no staging VGPRs. The DMA image is lane-linear, so the
//     bank swizzle (f2_swz: chunk ^ (row >> 1) & 7 on 128-byte rows, the scheme of cb_gemm_glds.hip; a 2-bit permutation on the
//     192-byte rows of d <= 96) is applied to the SOURCE address.
//   * LDS-DMA cannot transpose, and the second MFMA of each phase (P.V, dS.K, dS^T.Q, P^T.dO) wants its streamed operand
//     transposed ([head dim][row], with the rows of a 32-row group permuted to the k-slot order in which the probabilities lie
//     in the accumulators). Those operands are therefore transposed ONCE per layer by a small prepass kernel into a
//     [B*H][16*DT][N] scratch image (10 MB at N = 4096: ~5 us) and streamed from there like any other tile.
// Not taken by v2 (falls back to cb_flash_bf16.hip): causal masks, head dims outside (32, 80], fewer than 256 rows on either
// side (cross-attention, the small levels), the query-split dK/dV path. Measured (tools/bench_flash.py, bs = 4, forward / backward):
// d = 40, N = 4096: 165.6 / 433.6 us against 290.7 / 874.9 us for the round-2 kernels; d = 80, N = 1024: 33.6 / 107.4 us against
// 43.0 / 113.5 us (one block per CU there: 8 row blocks x 32 heads, and 98 KB of LDS per dK/dV block).
#include "cb_device.h"
#include <type_traits>

namespace cb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int F2_TK = 64;             // rows of a streamed tile
constexpr int F2_OOB = -16;           // 0xFFFFFFF0 >= num_records: the DMA writes zeros
// Tile images (all in 16-byte chunks, filled by LDS-DMA in 4 KiB slots = 256 threads x 16 bytes):
//   row-major image  : 64 rows (keys / queries) x 4 KS chunks — chunk c = head-dim elements 8c .. 8c+7; KS = 32-wide k groups of the head
//                      dim: 2 (d <= 64: 128-byte rows) or 3 (d <= 96: 192-byte rows)
//   transposed image : 16 DT rows (head-dim index) x 8 chunks — chunk c = permuted positions 8c .. 8c+7 of the 64-row tile
template <int KS> struct F2R { static constexpr int CPR = 4 * KS, NS = CPR * 64 / 256, BYTES = NS * 4096; };
template <int DT> struct F2T { static constexpr int NS = (16 * DT * 8 + 255) / 256, BYTES = NS * 4096; };
// bank swizzle of a row-major image, applied to the LOGICAL chunk index c of row r (an involution within aligned groups of 8 / 4 chunks):
// 128-byte rows (8 chunks): c ^ ((r >> 1) & 7), the scheme of cb_gemm_glds.hip; 192-byte rows (12 chunks): consecutive rows already start
// 12 bank quads apart, so only the low two chunk bits are permuted, by {0,3,2,1}[(r >> 2) & 3] — with either, the 16-lane groups of the
// MFMA operand read (ds_read_b128: rows {0-3,12-15} at chunk 4ks+g and rows {4-11} at chunk 4ks+g+1, ...) hit 16 distinct bank quads.
template <int KS> __device__ __forceinline__ int f2_swz(int c, int r) {
    if constexpr (KS == 2) return c ^ ((r >> 1) & 7);
    else return c ^ ((0x6C >> (2 * ((r >> 2) & 3))) & 3);          // 0x6C = 01 10 11 00b : {0, 3, 2, 1}
}

__device__ __forceinline__ float f2_quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float f2_quad_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f2_srd(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ bf16x8_t f2_pack8(const f32x4& a, const f32x4& b) {
    bf16x8_t r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    return r;
}
// a row of the head as MFMA B operand: reg[s] = src[row][32s + 8g .. +7], zero beyond d / limit
template <int KS>
__device__ __forceinline__ void f2_rowregs(bf16x8_t (&reg)[KS], const bf16_t* __restrict__ src, long ld, int row, int limit, int d, int g) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 32 * s + 8 * g;
        u32x4b v = {0u, 0u, 0u, 0u};
        if (row < limit && c < d) v = *reinterpret_cast<const u32x4b*>(src + (long)row * ld + c);
        reg[s] = __builtin_bit_cast(bf16x8_t, v);
    }
}
__device__ __forceinline__ void f2_store4(bf16_t* p, const f32x4& v, float beta) {
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (beta != 0.f) { const float4 old = ld4(p); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
    st4(p, o);
}

// ---- LDS-DMA of one tile image: thread slot idx = tid + 256 i owns LDS chunk idx (lane-linear image) = (row, physical chunk) and fetches
// the LOGICAL chunk the swizzle maps there.
template <int NS> struct F2Slots {
    int off[NS];                       // byte offset of (row, logical chunk) relative to the tile's first row, or F2_OOB
    int row[NS];
    // row-major source: element (r, 8c) of a [rows][ld] matrix; chunks at or beyond the head dim d are never fetched (zeros)
    template <int KS> __device__ __forceinline__ void init_rows(long ld, int d, int tid) {
        constexpr int CPR = F2R<KS>::CPR;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = tid + 256 * i, r = idx / CPR, c = f2_swz<KS>(idx - r * CPR, r);
            row[i] = r;
            off[i] = (8 * c < d) ? (int)(((long)r * ld + 8 * c) * 2) : F2_OOB;
        }
    }
    // transposed source: row = head-dim index (< DR) of a [DR][Np] image, 8 chunks of 8 positions per row (128-byte rows)
    __device__ __forceinline__ void init_tr(int Np, int DR, int tid) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = tid + 256 * i, r = idx >> 3, c = f2_swz<2>(idx & 7, r);
            row[i] = r;
            off[i] = (r < DR) ? (r * Np + 8 * c) * 2 : F2_OOB;
        }
    }
};
// dst = this wave's 1 KiB piece of the image (wave-uniform): image + 1024 * wave; slot i lands 4 KiB further
template <int NS>
__device__ __forceinline__ void f2_dma_rows(__amdgpu_buffer_rsrc_t srd, char* dst, const F2Slots<NS>& s, int r0, int limit, long ld) {
    const int base = (int)((long)r0 * ld * 2);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int o = (s.off[i] != F2_OOB && r0 + s.row[i] < limit) ? s.off[i] + base : F2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(dst + i * 4096), 16, o, 0, 0, 0);
    }
}
template <int NS>
__device__ __forceinline__ void f2_dma_tr(__amdgpu_buffer_rsrc_t srd, char* dst, const F2Slots<NS>& s, int p0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int o = (s.off[i] != F2_OOB) ? s.off[i] + p0 * 2 : F2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(dst + i * 4096), 16, o, 0, 0, 0);
    }
}
// MFMA A-operand reads: 16-row block `blk` of an image, 32-wide k group `kg` (lane (l15, g) holds k-slots 8g .. 8g+7 of the group).
// lor[kg] / lot[kg] = this lane's byte offset inside a block of the row-major / transposed image (f2_lane_offsets)
template <int KS> __device__ __forceinline__ bf16x8_t f2_frag_r(const char* img, int blk, const int (&lor)[KS], int kg) {
    return *reinterpret_cast<const bf16x8_t*>(img + blk * (16 * F2R<KS>::CPR * 16) + lor[kg]);
}
__device__ __forceinline__ bf16x8_t f2_frag_t(const char* img, int blk, const int (&lot)[2], int kg) {
    return *reinterpret_cast<const bf16x8_t*>(img + blk * 2048 + lot[kg]);
}
template <int KS> __device__ __forceinline__ void f2_lane_offsets(int (&lor)[KS], int (&lot)[2], int l15, int g) {
#pragma unroll
    for (int x = 0; x < KS; ++x) lor[x] = (l15 * F2R<KS>::CPR + f2_swz<KS>(4 * x + g, l15)) * 16;
#pragma unroll
    for (int x = 0; x < 2; ++x) lot[x] = (l15 * 8 + f2_swz<2>(4 * x + g, l15)) * 16;
}

// ---------------------------------------------------------------------------------------------- prepass: transpose + permute
// XT[bh][dd][p] = X[b][n(p)][h*sh + dd] (zero for dd >= d, n >= N), p = 64 T + 32 G + 8 g + 4 sub + e  <->  n = 64 T + 32 G + 16 sub + 4 g + e:
// position order = the k-slot order of a 32-row group in the S^T accumulators (lane (col, g) holds rows 4g .. 4g+3 of sub-tiles 2G, 2G+1).
// ones_row != 0 (forward V^T only, d < DR): row d of the image is 1.0 for every real key — the P.V MFMA that contracts P with V^T then also
// produces the softmax denominator sum_k P in its accumulator row d, for free (f2_fwd_kernel<.., ONES = true>).
// LSE definition with ONES (d % 16 != 0, e.g. d = 40): that denominator is the sum of the bf16-ROUNDED probabilities (the MFMA's B operand),
// not of the fp32 ones the round-2 kernels add up, so the stored log-sum-exp differs from theirs by up to 2^-9 relative. The backward
// kernels (v2, and the v1 fallback when v2 is not eligible) recompute P in fp32 against whichever LSE the forward stored: with a v2 forward
// a recomputed row then sums to 1 only to ~2^-9 — inside every bf16 tolerance of tests/test_gpu_bf16_*.py, and the same order as the
// rounding of P itself that the forward output already carries.
__global__ __launch_bounds__(256) void f2_transpose_kernel(const bf16_t* __restrict__ X, long ld, long sb, long sh, int H, int N, int d,
                                                           int DR, int Np, bf16_t* __restrict__ XT, int ones_row) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][98];      // 64 rows x DR <= 96 head-dim values (+2: the row pitch stays a multiple of 4 bytes and off the banks' period)
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, n0 = blockIdx.x * 64;
    const bf16_t* src = X + b * sb + h * sh;
    const int cpr = DR >> 3;
    for (int i = threadIdx.x; i < 64 * cpr; i += 256) {
        const int key = i / cpr, ch = i - key * cpr, n = n0 + key;
        u32x4b v = {0u, 0u, 0u, 0u};
        if (n < N && 8 * ch < d) v = *reinterpret_cast<const u32x4b*>(src + (long)n * ld + 8 * ch);
        else if (ones_row && n < N && 8 * ch == d) v.x = 0x3F80u;                       // bf16 1.0 at head-dim index d
        unsigned* t = reinterpret_cast<unsigned*>(&tile[key][8 * ch]);
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DR * 8; i += 256) {
        const int dd = i >> 3, pc = i & 7;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = 8 * pc + j, G = p >> 5, q = p & 31;
            const int key = 32 * G + 16 * ((q >> 2) & 1) + 4 * (q >> 3) + (q & 3);
            e[j] = tile[key][dd];
        }
        u32x4b o = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                    (unsigned)e[6] | ((unsigned)e[7] << 16)};
        *reinterpret_cast<u32x4b*>(XT + ((long)bh * DR + dd) * Np + n0 + 8 * pc) = o;
    }
}

// ---------------------------------------------------------------------------------------------- forward
// block: 4 waves x (16 QB) queries; ring of two stages {K tile (row-major), V^T tile (transposed)}
// VALU is what bounds this kernel at d = 40 (one exp per score against 160 MFMA FLOPs), so the softmax is cut to v_max, v_fma, v_exp and
// half a v_cvt_pk per score: the maximum is taken on the raw scores (scale > 0), the scale and the running maximum enter the exponent as
// one FMA, and with ONES the row sum comes out of the P.V MFMA (row d of the V^T image is all ones) instead of 16 v_add per lane and tile;
// the accumulator rescale is skipped while no lane's running maximum moves.
template <int KS, int DT, int QB, bool ONES, bool RG>
__global__ __launch_bounds__(256, 2) void f2_fwd_kernel(const FlashArgs a, const bf16_t* __restrict__ VTg, int Np) {
    constexpr int RIMG = F2R<KS>::BYTES, STAGE = RIMG + F2T<DT>::BYTES;
    __shared__ __attribute__((aligned(16))) char ring[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * (64 * QB) + wave * (16 * QB);
    const bf16_t* Q = static_cast<const bf16_t*>(a.q) + b * a.sqb + h * a.sqh;
    const bf16_t* K = static_cast<const bf16_t*>(a.k) + b * a.skb + h * a.skh;
    constexpr int DR = 16 * DT;

    bf16x8_t qreg[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) f2_rowregs<KS>(qreg[qb], Q, a.ldq, q0 + 16 * qb + l15, a.Nq, a.d, g);
    const float scale2 = a.scale * 1.4426950408889634f;          // scores are kept in the log2 domain
    f32x4 o[QB][DT];
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = -INFINITY; l[qb] = 0.f;
#pragma unroll
        for (int t = 0; t < DT; ++t) o[qb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const __amdgpu_buffer_rsrc_t srdK = f2_srd(K), srdT = f2_srd(VTg + (long)bh * DR * Np);
    F2Slots<F2R<KS>::NS> sk;
    F2Slots<F2T<DT>::NS> st;
    sk.template init_rows<KS>(a.ldk, a.d, tid);
    st.init_tr(Np, DR, tid);
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 1024);
    int lor[KS], lot[2];
    f2_lane_offsets<KS>(lor, lot, l15, g);

    const int ntiles = (a.Nk + F2_TK - 1) / F2_TK;
    f2_dma_rows(srdK, ring + wbase, sk, 0, a.Nk, a.ldk);
    f2_dma_tr(srdT, ring + RIMG + wbase, st, 0);
    for (int t = 0; t < ntiles; ++t) {
        // tile t has landed for this wave (nothing else is in flight) ... and, after the barrier, for every wave; the barrier also says
        // that everybody has finished reading the other slot, which is refilled next
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntiles) {
            char* nx = ring + ((t + 1) & 1) * STAGE + wbase;
            f2_dma_rows(srdK, nx, sk, (t + 1) * F2_TK, a.Nk, a.ldk);
            f2_dma_tr(srdT, nx + RIMG, st, (t + 1) * F2_TK);
        }
        // the key mask exists only in the last tile of a ragged key count: two instantiations of the tile body behind a wave-uniform branch
        // (left to the compiler, the masking turns into 64 v_cndmask per row block in EVERY tile)
        auto tile_body = [&](auto mask_tag) {
            constexpr bool MK = decltype(mask_tag)::value;
            const char* Ks = ring + (t & 1) * STAGE;
            const char* Vt = Ks + RIMG;
            const int k0 = t * F2_TK;
            f32x4 s[QB][4];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[qb][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const bf16x8_t av = f2_frag_r<KS>(Ks, kt, lor, ks);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, qreg[qb][ks], s[qb][kt], 0, 0, 0);
                }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (MK) s[qb][kt][r] = (k0 + kt * 16 + g * 4 + r < a.Nk) ? s[qb][kt][r] : -INFINITY;
                        mx = fmaxf(mx, s[qb][kt][r]);
                    }
                mx = f2_quad_max(mx) * scale2;                                       // log2-domain maximum of this tile (scale2 > 0)
                const float mnew = fmaxf(m[qb], mx);
                const float muse = (mnew == -INFINITY) ? 0.f : mnew;
                float ps = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kt][r], scale2, -muse));
                        s[qb][kt][r] = p;
                        if (!ONES) ps += p;
                    }
                if (!__all(mnew == m[qb])) {                                          // some lane's running maximum moved: rescale (wave-uniform branch)
                    const float alpha = __builtin_amdgcn_exp2f(m[qb] - muse);         // m = -inf -> 0
#pragma unroll
                    for (int t2 = 0; t2 < DT; ++t2) o[qb][t2] *= alpha;
                    l[qb] *= alpha;
                    m[qb] = mnew;
                }
                if (!ONES) l[qb] += f2_quad_sum(ps);
            }
            // O^T += V^T P^T : the V^T fragment of (16 output columns, 32-key group) is read once and feeds QB MFMAs
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                bf16x8_t pb[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) pb[qb] = f2_pack8(s[qb][2 * G], s[qb][2 * G + 1]);
#pragma unroll
                for (int t2 = 0; t2 < DT; ++t2) {
                    const bf16x8_t av = f2_frag_t(Vt, t2, lot, G);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) o[qb][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[qb], o[qb][t2], 0, 0, 0);
                }
            }
        };
        if constexpr (RG) {          // RG = the launch has a ragged last key tile (host: Nk % 64 != 0); otherwise the masked body is not even compiled
            if (t == ntiles - 1) tile_body(std::true_type{}); else tile_body(std::false_type{});
        } else tile_body(std::false_type{});
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0 + 16 * qb + l15;
        float lq = l[qb];
        if (ONES) {        // the denominator sits in accumulator row d of this query's column: tile d / 16, lane group (d % 16) / 4, register 0
            const float v = o[qb][DT - 1][0];
            lq = __shfl(v, l15 + 16 * ((a.d & 15) >> 2), 64);
        }
        if (qrow < a.Nq) {
            const float inv = lq > 0.f ? 1.f / lq : 0.f;
            bf16_t* O = static_cast<bf16_t*>(a.o) + b * a.sob + h * a.soh + (long)qrow * a.ldo;
#pragma unroll
            for (int t2 = 0; t2 < DT; ++t2) {
                const int c = t2 * 16 + g * 4;
                if (c < a.d) st4(O + c, make_float4(o[qb][t2][0] * inv, o[qb][t2][1] * inv, o[qb][t2][2] * inv, o[qb][t2][3] * inv));
            }
            if (g == 0 && a.lse) a.lse[(long)bh * a.Nq + qrow] = m[qb] + __builtin_amdgcn_logf(lq);   // log2 domain
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward: dQ
// block: 4 waves x (16 QB) queries; stage = {K (row-major), V (row-major), K^T (transposed)}; also publishes D = rowsum(dO * O)
template <int KS, int DT> constexpr int f2_dq_stage() { return 2 * F2R<KS>::BYTES + F2T<DT>::BYTES; }
template <int KS, int DT, int QB, bool RG>
__global__ __launch_bounds__(256, 2) void f2_dq_kernel(const FlashArgs a, const bf16_t* __restrict__ KTg, int Np) {
    constexpr int RIMG = F2R<KS>::BYTES, STAGE = f2_dq_stage<KS, DT>();
    extern __shared__ __attribute__((aligned(16))) char ring_dq[];          // 2 x STAGE (48 KB at d <= 64, 72 KB at d <= 96: dynamic)
    char* ring = ring_dq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * (64 * QB) + wave * (16 * QB);
    const bf16_t* Q = static_cast<const bf16_t*>(a.q) + b * a.sqb + h * a.sqh;
    const bf16_t* K = static_cast<const bf16_t*>(a.k) + b * a.skb + h * a.skh;
    const bf16_t* V = static_cast<const bf16_t*>(a.v) + b * a.skb + h * a.skh;
    const bf16_t* dO = static_cast<const bf16_t*>(a.dout) + b * a.sob + h * a.soh;
    const bf16_t* Oo = static_cast<const bf16_t*>(a.o) + b * a.sob + h * a.soh;
    constexpr int DR = 16 * DT;

    bf16x8_t qreg[QB][KS], dreg[QB][KS];
    float lse[QB], Dq[QB];
    f32x4 gq[QB][DT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0 + 16 * qb + l15;
        f2_rowregs<KS>(qreg[qb], Q, a.ldq, qrow, a.Nq, a.d, g);
        f2_rowregs<KS>(dreg[qb], dO, a.ldo, qrow, a.Nq, a.d, g);
        lse[qb] = qrow < a.Nq ? a.lse[(long)bh * a.Nq + qrow] : 0.f;
        bf16x8_t oreg[KS];
        f2_rowregs<KS>(oreg, Oo, a.ldo, qrow, a.Nq, a.d, g);
        float dsum = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += (float)oreg[s_][e] * (float)dreg[qb][s_][e];
        dsum = f2_quad_sum(dsum);
        Dq[qb] = dsum;
        if (g == 0 && qrow < a.Nq) a.dvec[(long)bh * a.Nq + qrow] = dsum;
#pragma unroll
        for (int t = 0; t < DT; ++t) gq[qb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float scale2 = a.scale * 1.4426950408889634f;
    const __amdgpu_buffer_rsrc_t srdK = f2_srd(K), srdV = f2_srd(V), srdT = f2_srd(KTg + (long)bh * DR * Np);
    F2Slots<F2R<KS>::NS> sr;
    F2Slots<F2T<DT>::NS> st;
    sr.template init_rows<KS>(a.ldk, a.d, tid);
    st.init_tr(Np, DR, tid);
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 1024);
    int lor[KS], lot[2];
    f2_lane_offsets<KS>(lor, lot, l15, g);

    const int ntiles = (a.Nk + F2_TK - 1) / F2_TK;
    f2_dma_rows(srdK, ring + wbase, sr, 0, a.Nk, a.ldk);
    f2_dma_rows(srdV, ring + RIMG + wbase, sr, 0, a.Nk, a.ldk);
    f2_dma_tr(srdT, ring + 2 * RIMG + wbase, st, 0);
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntiles) {
            char* nx = ring + ((t + 1) & 1) * STAGE + wbase;
            f2_dma_rows(srdK, nx, sr, (t + 1) * F2_TK, a.Nk, a.ldk);
            f2_dma_rows(srdV, nx + RIMG, sr, (t + 1) * F2_TK, a.Nk, a.ldk);
            f2_dma_tr(srdT, nx + 2 * RIMG, st, (t + 1) * F2_TK);
        }
        auto tile_body = [&](auto mask_tag) {              // ragged last key tile only: see f2_fwd_kernel
            constexpr bool MK = decltype(mask_tag)::value;
            const char* Ks = ring + (t & 1) * STAGE;
            const char* Vs = Ks + RIMG;
            const char* Kt = Ks + 2 * RIMG;
            const int k0 = t * F2_TK;
#pragma unroll
            for (int G = 0; G < 2; ++G) {                               // one 32-key group at a time keeps S / dP at 2 sub-tiles per row block
                f32x4 s[QB][2], dp[QB][2];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) { s[qb][sub] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[qb][sub] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const bf16x8_t ak = f2_frag_r<KS>(Ks, 2 * G + sub, lor, ks);
                        const bf16x8_t av = f2_frag_r<KS>(Vs, 2 * G + sub, lor, ks);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            s[qb][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, qreg[qb][ks], s[qb][sub], 0, 0, 0);      // S^T  = K Q^T
                            dp[qb][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, dreg[qb][ks], dp[qb][sub], 0, 0, 0);    // dP^T = V dO^T
                        }
                    }
                bf16x8_t sb[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][sub][r], scale2, -lse[qb]));
                            if constexpr (MK) p = (k0 + (2 * G + sub) * 16 + g * 4 + r < a.Nk) ? p : 0.f;
                            s[qb][sub][r] = p * (dp[qb][sub][r] - Dq[qb]);                    // d(q.k) / scale: the scale multiplies dQ once, at the end
                        }
                    sb[qb] = f2_pack8(s[qb][0], s[qb][1]);
                }
                // dQ^T += K^T dS^T
#pragma unroll
                for (int t2 = 0; t2 < DT; ++t2) {
                    const bf16x8_t ak = f2_frag_t(Kt, t2, lot, G);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) gq[qb][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, sb[qb], gq[qb][t2], 0, 0, 0);
                }
            }
        };
        if constexpr (RG) {          // RG = the launch has a ragged last key tile (host: Nk % 64 != 0); otherwise the masked body is not even compiled
            if (t == ntiles - 1) tile_body(std::true_type{}); else tile_body(std::false_type{});
        } else tile_body(std::false_type{});
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0 + 16 * qb + l15;
        if (qrow < a.Nq) {
            bf16_t* Gp = static_cast<bf16_t*>(a.dq) + b * a.sqb + h * a.sqh + (long)qrow * a.ldq;
#pragma unroll
            for (int t2 = 0; t2 < DT; ++t2) {
                const int c = t2 * 16 + g * 4;
                if (c < a.d) f2_store4(Gp + c, gq[qb][t2] * a.scale, a.beta_q);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward: dK, dV
// block: 4 waves x (16 QB) keys (key = lane & 15 is the MFMA column); stage = {Q, dO (row-major), Q^T, dO^T (transposed), lse | D (64 + 64 floats)}
//   S[q][key]    = Q K^T         A = Q tile,  B = K regs (unscaled; scale applied after)
//   dP[q][key]   = dO V^T        A = dO tile, B = V regs
//   dV^T[d][key] += dO^T P       A = dO^T tile, B = P regs
//   dK^T[d][key] += Q^T dS       A = Q^T tile,  B = dS regs
template <int KS, int DT> constexpr int f2_dkv_stage() { return 2 * F2R<KS>::BYTES + 2 * F2T<DT>::BYTES + 1024; }
template <int KS, int DT, int QB, bool RG>
__global__ __launch_bounds__(256, 2) void f2_dkv_kernel(const FlashArgs a, const bf16_t* __restrict__ QTg, const bf16_t* __restrict__ DTg, int Np) {
    constexpr int RIMG = F2R<KS>::BYTES, TIMG = F2T<DT>::BYTES, STAGE = f2_dkv_stage<KS, DT>();
    extern __shared__ __attribute__((aligned(16))) char ring2[];
    char* ring = ring2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int k0 = blockIdx.x * (64 * QB) + wave * (16 * QB);
    const bf16_t* Q = static_cast<const bf16_t*>(a.q) + b * a.sqb + h * a.sqh;
    const bf16_t* K = static_cast<const bf16_t*>(a.k) + b * a.skb + h * a.skh;
    const bf16_t* V = static_cast<const bf16_t*>(a.v) + b * a.skb + h * a.skh;
    const bf16_t* dO = static_cast<const bf16_t*>(a.dout) + b * a.sob + h * a.soh;
    constexpr int DR = 16 * DT;

    bf16x8_t kreg[QB][KS], vreg[QB][KS];
    f32x4 gk[QB][DT], gv[QB][DT];
#pragma unroll
    for (int kb = 0; kb < QB; ++kb) {
        f2_rowregs<KS>(kreg[kb], K, a.ldk, k0 + 16 * kb + l15, a.Nk, a.d, g);
        f2_rowregs<KS>(vreg[kb], V, a.ldk, k0 + 16 * kb + l15, a.Nk, a.d, g);
#pragma unroll
        for (int t = 0; t < DT; ++t) { gk[kb][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; gv[kb][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    const float scale2 = a.scale * 1.4426950408889634f;
    const __amdgpu_buffer_rsrc_t srdQ = f2_srd(Q), srdD = f2_srd(dO), srdQT = f2_srd(QTg + (long)bh * DR * Np), srdDT = f2_srd(DTg + (long)bh * DR * Np);
    const __amdgpu_buffer_rsrc_t srdL = f2_srd(a.lse + (long)bh * a.Nq), srdS = f2_srd(a.dvec + (long)bh * a.Nq);
    F2Slots<F2R<KS>::NS> sq, so;
    F2Slots<F2T<DT>::NS> st;
    sq.template init_rows<KS>(a.ldq, a.d, tid);
    so.template init_rows<KS>(a.ldo, a.d, tid);
    st.init_tr(Np, DR, tid);
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 1024);
    int lor[KS], lot[2];
    f2_lane_offsets<KS>(lor, lot, l15, g);

    // lse / D of a query tile: 64 + 64 floats = 32 chunks, fetched by wave 0 (lanes 0-15: lse, 16-31: D, the rest zero-fill their slots)
    auto issue = [&](int q0t, int slot) {
        char* nx = ring + slot * STAGE + wbase;
        f2_dma_rows(srdQ, nx, sq, q0t, a.Nq, a.ldq);
        f2_dma_rows(srdD, nx + RIMG, so, q0t, a.Nq, a.ldo);
        f2_dma_tr(srdQT, nx + 2 * RIMG, st, q0t);
        f2_dma_tr(srdDT, nx + 2 * RIMG + TIMG, st, q0t);
        if (wave == 0) {
            const int c = lane & 15;
            const int o = (lane < 32 && q0t + 4 * c < a.Nq) ? (q0t + 4 * c) * 4 : F2_OOB;
            if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdL, (lds_ptr_t)(ring + slot * STAGE + 2 * RIMG + 2 * TIMG), 16, o, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(srdS, (lds_ptr_t)(ring + slot * STAGE + 2 * RIMG + 2 * TIMG), 16, o, 0, 0, 0);
        }
    };
    const int ntiles = (a.Nq + F2_TK - 1) / F2_TK;
    issue(0, 0);
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ntiles) issue((t + 1) * F2_TK, (t + 1) & 1);
        auto tile_body = [&](auto mask_tag) {              // ragged query tile / ragged key block only: see f2_fwd_kernel
            constexpr bool MK = decltype(mask_tag)::value;
            const char* Qs = ring + (t & 1) * STAGE;
            const char* Ds = Qs + RIMG;
            const char* Qt = Qs + 2 * RIMG;
            const char* Dt = Qs + 2 * RIMG + TIMG;
            const float* Ls = reinterpret_cast<const float*>(Qs + 2 * RIMG + 2 * TIMG);          // lse[64], D[64]
            const int q0t = t * F2_TK;
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                f32x4 s[QB][2], dp[QB][2];
#pragma unroll
                for (int kb = 0; kb < QB; ++kb)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) { s[kb][sub] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[kb][sub] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const bf16x8_t aq = f2_frag_r<KS>(Qs, 2 * G + sub, lor, ks);
                        const bf16x8_t ad = f2_frag_r<KS>(Ds, 2 * G + sub, lor, ks);
#pragma unroll
                        for (int kb = 0; kb < QB; ++kb) {
                            s[kb][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, kreg[kb][ks], s[kb][sub], 0, 0, 0);
                            dp[kb][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, vreg[kb][ks], dp[kb][sub], 0, 0, 0);
                        }
                    }
                // rows of the C layout are queries 4*g + r of sub-tile 2G + sub, the column is this lane's key
                bf16x8_t pb[QB], sb[QB];
#pragma unroll
                for (int kb = 0; kb < QB; ++kb) {
                    const int krow = k0 + 16 * kb + l15;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ql = (2 * G + sub) * 16 + g * 4 + r;
                            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][sub][r], scale2, -Ls[ql]));
                            if constexpr (MK) p = (q0t + ql < a.Nq && krow < a.Nk) ? p : 0.f;
                            const float ds = p * (dp[kb][sub][r] - Ls[64 + ql]);                 // / scale: applied to dK once, at the end
                            s[kb][sub][r] = p;
                            dp[kb][sub][r] = ds;
                        }
                    pb[kb] = f2_pack8(s[kb][0], s[kb][1]);
                    sb[kb] = f2_pack8(dp[kb][0], dp[kb][1]);
                }
#pragma unroll
                for (int t2 = 0; t2 < DT; ++t2) {
                    const bf16x8_t ad = f2_frag_t(Dt, t2, lot, G);
                    const bf16x8_t aq = f2_frag_t(Qt, t2, lot, G);
#pragma unroll
                    for (int kb = 0; kb < QB; ++kb) {
                        gv[kb][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, pb[kb], gv[kb][t2], 0, 0, 0);
                        gk[kb][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, sb[kb], gk[kb][t2], 0, 0, 0);
                    }
                }
            }
        };
        if constexpr (RG) {          // RG = ragged query tile or ragged key block somewhere in the launch (host)
            if ((t + 1) * F2_TK > a.Nq || (int)blockIdx.x * (64 * QB) + 64 * QB > a.Nk) tile_body(std::true_type{}); else tile_body(std::false_type{});
        } else tile_body(std::false_type{});
    }
#pragma unroll
    for (int kb = 0; kb < QB; ++kb) {
        const int krow = k0 + 16 * kb + l15;
        if (krow < a.Nk) {
            bf16_t* GK = static_cast<bf16_t*>(a.dk) + b * a.skb + h * a.skh + (long)krow * a.ldk;
            bf16_t* GV = static_cast<bf16_t*>(a.dv) + b * a.skb + h * a.skh + (long)krow * a.ldk;
#pragma unroll
            for (int t2 = 0; t2 < DT; ++t2) {
                const int c = t2 * 16 + g * 4;
                if (c < a.d) { f2_store4(GK + c, gk[kb][t2] * a.scale, a.beta_kv); f2_store4(GV + c, gv[kb][t2], a.beta_kv); }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host
static bool f2_enabled() {
    static const bool on = [] { const char* e = getenv("CB_FLASH_BF16_V2"); return !(e && e[0] == '0'); }();
    return on;
}
static int f2_np(int n) { return (n + F2_TK - 1) / F2_TK * F2_TK; }
static int f2_dt(int d) { return (d + 15) / 16; }

// Shapes v2 takes: 32 < d <= 64 (two 32-wide k groups: the 64x64-level self-attention, d = 40) and 64 < d <= 80 (three k groups, five
// output tiles: the 32x32 level, d = 80); no mask; at least 256 rows on both sides; row counts multiples of 8 (16-byte lse / D chunks).
bool flash_bf16_v2_ok(const FlashArgs& a) {
    if (!f2_enabled() || a.dt != DT_BF16 || a.causal) return false;
    if (a.d <= 32 || a.d > 80 || a.d % 8) return false;
    if (a.Nq < 256 || a.Nk < 256 || a.Nq % 8 || a.Nk % 8) return false;
    return true;       // strides / alignment: flash_bf16_supported (checked by the caller first)
}
// bytes of the transposed scratch images: forward = V^T; backward = K^T | Q^T | dO^T
size_t flash_bf16_v2_tws_bytes(const FlashArgs& a, int B, bool backward) {
    if (!flash_bf16_v2_ok(a)) return 0;
    const size_t BH = (size_t)B * a.H, DR = 16 * f2_dt(a.d);
    const size_t kimg = BH * DR * f2_np(a.Nk) * 2, qimg = BH * DR * f2_np(a.Nq) * 2;
    return backward ? kimg + 2 * qimg : kimg;
}
static void f2_transpose(const Launch& L, const void* X, long ld, long sb, long sh, int H, int N, int d, int BH, bf16_t* XT, int ones_row = 0) {
    const int DR = 16 * f2_dt(d), Np = f2_np(N);
    hipLaunchKernelGGL(f2_transpose_kernel, dim3(Np / 64, BH), dim3(256), 0, L.stream, static_cast<const bf16_t*>(X), ld, sb, sh, H, N, d, DR, Np, XT,
                       ones_row);
}

// rows per block = 64 QB: take the widest row block that still gives every CU two blocks (experiments: CB_F2_QB="fwd,dq,dkv" caps them)
static int f2_pick_qb(int rows, int BH, int qbmax, int which) {
    static int envcap[3] = {0, 0, 0};
    static const bool once = [] { const char* e = getenv("CB_F2_QB"); if (e) sscanf(e, "%d,%d,%d", &envcap[0], &envcap[1], &envcap[2]); return true; }();
    (void)once;
    int qb = qbmax;
    if (envcap[which] == 1 || envcap[which] == 2 || envcap[which] == 4) qb = std::min(qb, envcap[which]);
    while (qb > 1 && (long)cdiv(rows, 64 * qb) * BH < 512) qb >>= 1;
    return qb;
}
// Instantiations: (KS, DT) = (2, 3) d <= 48, (2, 4) d <= 64, (3, 5) d <= 80. QB = 4 exists for (2, 3) and whole tiles only — four row
// blocks x four / five output tiles, or a second (masked) copy of the tile body, would spill, and celebbasis_amd/build.py fails the build
// on any scratch in this file.
template <int KS, int DT, int QB, bool ONES, bool RG>
static void f2_go_fwd(dim3 grid, hipStream_t st, const FlashArgs& a, const bf16_t* VT, int Np) {
    hipLaunchKernelGGL((f2_fwd_kernel<KS, DT, QB, ONES, RG>), grid, dim3(256), 0, st, a, VT, Np);
}
template <int KS, int DT, bool ONES>
static void f2_launch_fwd(int qb, bool rg, dim3 grid, hipStream_t st, const FlashArgs& a, const bf16_t* VT, int Np) {
    if (rg) { if (qb == 2) f2_go_fwd<KS, DT, 2, ONES, true>(grid, st, a, VT, Np); else f2_go_fwd<KS, DT, 1, ONES, true>(grid, st, a, VT, Np); return; }
    if constexpr (KS == 2 && DT == 3) { if (qb == 4) { f2_go_fwd<2, 3, 4, ONES, false>(grid, st, a, VT, Np); return; } }
    if (qb == 2) f2_go_fwd<KS, DT, 2, ONES, false>(grid, st, a, VT, Np); else f2_go_fwd<KS, DT, 1, ONES, false>(grid, st, a, VT, Np);
}
template <int KS, int DT, int QB, bool RG>
static void f2_go_dq(dim3 grid, hipStream_t st, const FlashArgs& a, const bf16_t* KT, int Np) {
    constexpr int LDS = 2 * f2_dq_stage<KS, DT>();
    set_dyn_lds(reinterpret_cast<const void*>(&f2_dq_kernel<KS, DT, QB, RG>), LDS);
    hipLaunchKernelGGL((f2_dq_kernel<KS, DT, QB, RG>), grid, dim3(256), LDS, st, a, KT, Np);
}
template <int KS, int DT>
static void f2_launch_dq(int qb, bool rg, dim3 grid, hipStream_t st, const FlashArgs& a, const bf16_t* KT, int Np) {
    if (rg) { if (qb == 2) f2_go_dq<KS, DT, 2, true>(grid, st, a, KT, Np); else f2_go_dq<KS, DT, 1, true>(grid, st, a, KT, Np); return; }
    if constexpr (KS == 2 && DT == 3) { if (qb == 4) { f2_go_dq<2, 3, 4, false>(grid, st, a, KT, Np); return; } }
    if (qb == 2) f2_go_dq<KS, DT, 2, false>(grid, st, a, KT, Np); else f2_go_dq<KS, DT, 1, false>(grid, st, a, KT, Np);
}
template <int KS, int DT, int QB, bool RG>
static void f2_go_dkv(dim3 grid, hipStream_t st, const FlashArgs& a, const bf16_t* QT, const bf16_t* DTp, int Np) {
    constexpr int LDS = 2 * f2_dkv_stage<KS, DT>();
    set_dyn_lds(reinterpret_cast<const void*>(&f2_dkv_kernel<KS, DT, QB, RG>), LDS);
    hipLaunchKernelGGL((f2_dkv_kernel<KS, DT, QB, RG>), grid, dim3(256), LDS, st, a, QT, DTp, Np);
}
template <int KS, int DT>
static void f2_launch_dkv(int qb, bool rg, dim3 grid, hipStream_t st, const FlashArgs& a, const bf16_t* QT, const bf16_t* DTp, int Np) {
    if (rg) {
        if constexpr (DT == 3) { if (qb == 2) { f2_go_dkv<KS, 3, 2, true>(grid, st, a, QT, DTp, Np); return; } }
        f2_go_dkv<KS, DT, 1, true>(grid, st, a, QT, DTp, Np);
    } else {
        if constexpr (DT <= 4) { if (qb == 2) { f2_go_dkv<KS, DT, 2, false>(grid, st, a, QT, DTp, Np); return; } }
        f2_go_dkv<KS, DT, 1, false>(grid, st, a, QT, DTp, Np);
    }
}
static int f2_qbmax(int dt, bool rg, int which) {      // which: 0 forward, 1 dQ, 2 dK/dV — must mirror the instantiations above
    if (which == 2) return (dt == 3 || (dt == 4 && !rg)) ? 2 : 1;
    return (dt == 3 && !rg) ? 4 : 2;
}

void flash_bf16_v2_forward(const Launch& L, const FlashArgs& a, int B, void* tws) {
    const int BH = B * a.H, Np = f2_np(a.Nk), dt = f2_dt(a.d);
    CB_CHECK(tws, CB_ERR_STATE, "flash v2: transposed-operand scratch missing");
    bf16_t* VT = static_cast<bf16_t*>(tws);
    const bool ones = (a.d & 15) != 0;       // a spare row in the last 16-row block of the V^T image: the softmax denominator rides on the P.V MFMA
    const bool rg = (a.Nk % F2_TK) != 0;
    f2_transpose(L, a.v, a.ldk, a.skb, a.skh, a.H, a.Nk, a.d, BH, VT, ones ? 1 : 0);
    const int qb = f2_pick_qb(a.Nq, BH, f2_qbmax(dt, rg, 0), 0);
    dim3 grid(cdiv(a.Nq, 64 * qb), BH);
    if (dt == 3) { if (ones) f2_launch_fwd<2, 3, true>(qb, rg, grid, L.stream, a, VT, Np); else f2_launch_fwd<2, 3, false>(qb, rg, grid, L.stream, a, VT, Np); }
    else if (dt == 4) { if (ones) f2_launch_fwd<2, 4, true>(qb, rg, grid, L.stream, a, VT, Np); else f2_launch_fwd<2, 4, false>(qb, rg, grid, L.stream, a, VT, Np); }
    else { if (ones) f2_launch_fwd<3, 5, true>(qb, rg, grid, L.stream, a, VT, Np); else f2_launch_fwd<3, 5, false>(qb, rg, grid, L.stream, a, VT, Np); }
}

void flash_bf16_v2_backward(const Launch& L, const FlashArgs& a, int B, bool need_dq, bool need_dkv, void* tws) {
    const int BH = B * a.H, Npk = f2_np(a.Nk), Npq = f2_np(a.Nq), dt = f2_dt(a.d), DR = 16 * dt;
    CB_CHECK(tws, CB_ERR_STATE, "flash v2: transposed-operand scratch missing");
    CB_CHECK(need_dq, CB_ERR_STATE, "flash v2 backward: the dK/dV kernel takes D = rowsum(dO * O) from the dQ kernel");
    bf16_t* KT = static_cast<bf16_t*>(tws);
    bf16_t* QT = KT + (size_t)BH * DR * Npk;
    bf16_t* DT = QT + (size_t)BH * DR * Npq;
    f2_transpose(L, a.k, a.ldk, a.skb, a.skh, a.H, a.Nk, a.d, BH, KT);
    {
        const bool rg = (a.Nk % F2_TK) != 0;
        const int qb = f2_pick_qb(a.Nq, BH, f2_qbmax(dt, rg, 1), 1);
        dim3 grid(cdiv(a.Nq, 64 * qb), BH);
        if (dt == 3) f2_launch_dq<2, 3>(qb, rg, grid, L.stream, a, KT, Npk);
        else if (dt == 4) f2_launch_dq<2, 4>(qb, rg, grid, L.stream, a, KT, Npk);
        else f2_launch_dq<3, 5>(qb, rg, grid, L.stream, a, KT, Npk);
    }
    if (need_dkv) {
        f2_transpose(L, a.q, a.ldq, a.sqb, a.sqh, a.H, a.Nq, a.d, BH, QT);
        f2_transpose(L, a.dout, a.ldo, a.sob, a.soh, a.H, a.Nq, a.d, BH, DT);
        // ragged = a partial query tile or a partial key block somewhere in the launch; the widest key block depends on it and vice versa
        int qb = f2_pick_qb(a.Nk, BH, f2_qbmax(dt, false, 2), 2);
        bool rg = (a.Nq % F2_TK) != 0 || (a.Nk % (64 * qb)) != 0;
        if (rg && qb > f2_qbmax(dt, true, 2)) { qb = f2_qbmax(dt, true, 2); rg = (a.Nq % F2_TK) != 0 || (a.Nk % (64 * qb)) != 0; }
        dim3 grid(cdiv(a.Nk, 64 * qb), BH);
        if (dt == 3) f2_launch_dkv<2, 3>(qb, rg, grid, L.stream, a, QT, DT, Npq);
        else if (dt == 4) f2_launch_dkv<2, 4>(qb, rg, grid, L.stream, a, QT, DT, Npq);
        else f2_launch_dkv<3, 5>(qb, rg, grid, L.stream, a, QT, DT, Npq);
    }
}

}  // namespace cb
