// cb_flash.hip — fused (flash-style) multi-head attention forward + backward in exact fp32 on the gfx950 matrix cores.
//
// Replaces the materialised softmax(QK^T)V of the reference (ldm/modules/attention.py:170-193 keeps the (8B, N, N) score
// tensor and its softmax alive for backward: 5 x 512 MiB at bs=1) for head dims d <= 160:
// UNet self/cross attention (d = 40/80/160, ctx = N or 77) and the CLIP causal attention (d = 64, 77x77).
//
// Mapping (v_mfma_f32_16x16x4_f32, wave64): every contraction is computed TRANSPOSED so that the softmax axis of a
// query lives inside one lane quartet:
//   S^T[key][query] = K . Q^T     A = K tile (LDS, K-style read), B = Q^T (registers, loaded once per block)
//   C/D layout: col = query = lane&15, row = key = 4*(lane>>4) + reg  ->  a query's 16 keys of a 16x16 tile sit in the
//   4 lanes {j, j+16, j+32, j+48}: row max / row sum = 16 in-register ops + two 64-wide xor-shuffles, no LDS.
//   O^T[d][query] += V^T . P^T    A = V tile (LDS), B = P^T = the S^T accumulator registers AS THEY ARE (the MFMA k index
//   = lane>>4 selects key 4*(lane>>4)+t at step t): probabilities never leave the register file and the running
//   max/sum/rescale factors of query j are already in the lanes that hold O^T[:, j].
// Backward = two kernels with the same trick (no atomics, deterministic): dQ (block owns 64 queries, loops over K/V tiles)
// and dK/dV (block owns 64 keys, loops over Q/dO tiles); both recompute P from the saved log-sum-exp.
// LDS tiles are row-major [row][S2] with S2 = DP + 4 (a multiple of 4: 16-byte chunks for the LDS-DMA and the float4 stores). ds_read_b32 is
// served in two 32-lane halves over 32 banks: the V-style fragment read (row 4*quarter + r, 16 columns per quarter) is conflict-free
// (4*S2 = 16 mod 32 puts the two quarters of a half 16 banks apart), the K-style read (16 rows x one column per quarter) is 2-way (S2*l15
// mod 32 takes 8 values). Rounds 1 - 5 used S2 = 2 (mod 32), which has it the other way round. Measured neutral (707 vs 704 us on the 64x64-level
// backward); bank-conflict cycles 34 -> 29 % of the LDS-active ones in dK/dV and the forward, 30 -> 38 % in dQ (profiles/r05_ / r06_pmc_flash_f32.txt):
// the loop does not wait for LDS either way (SQ_WAIT_INST_LDS = 2.9 M of 416 M wave cycles).
#include "cb_common.h"
#include <type_traits>

namespace cb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FL_ROWS = 64;      // rows (queries or keys) owned by a block: 4 waves x 16
// Timing ablations of the two backward kernels (build with CB_FLASH_ABL=<bits>; results are WRONG under them, tools/bench_flash.py only):
// 1 = no softmax arithmetic, 2 = no tile staging in the loop (no LDS stores, barriers, prefetch), 4 = MFMA A operands from registers (no LDS reads)
#ifndef FL_ABL
#define FL_ABL 0
#endif

__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

template <int DT> struct FlCfg {
    static constexpr int DP = DT * 16;                          // padded head dim
    static constexpr int KS = (DT == 3) ? 10 : DT * 4;          // MFMA k-steps (4 wide) over the head dim; DT = 3 serves d <= 40
                                                                // (SD's 64x64 attention): 10 steps, not 12 (the dispatcher sends 40 < d <= 48 to DT = 4)
    static constexpr int S2 = DP + 4;                           // LDS row stride (file header: V-style reads conflict-free, K-style 2-way; 16-byte aligned rows)
    static constexpr int TK = (DT >= 8) ? 32 : 64;              // rows of the streamed tile
    static constexpr int NT = TK / 16;                          // 16-row sub-tiles of the streamed tile
};

// Streamed tiles (TK rows x d floats, row-major source) go global -> registers -> LDS T[TK][S2]. The register stage is a
// software prefetch: the buffer loads of tile t+1 are issued before the MFMA work of tile t and land in LDS after it.
// Raw buffer loads zero-fill out-of-range lanes in hardware (rows beyond the tensor, head-dim padding columns), and the
// per-thread (row, column) decode is loop invariant, so a tile costs ~3 VALU per 16-byte load (the fp32 MFMA shares issue
// with the vector ALU: VALU in the loop is MFMA time lost).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int FL_OOB = (int)0x80000000;     // stays out of range (>= num_records = 2^31 - 1, unsigned) after any in-slice base is added
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fl_srd(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
template <int DT> struct FlTile {
    static constexpr int Q = FlCfg<DT>::DP / 4;                               // float4 per row
    static constexpr int N = (FlCfg<DT>::TK * Q + 255) / 256;                 // float4 per thread
};
template <int DT> struct FlStream {
    int goff[FlTile<DT>::N];      // (r*ld + c)*4 inside the source slice, FL_OOB for padding columns / unused slots
    int loff[FlTile<DT>::N];      // r*S2 + c in the LDS tile, -1 for unused slots
    int row[FlTile<DT>::N];
    __device__ __forceinline__ void init(long ld, int d) {
        constexpr int Q = FlTile<DT>::Q;
#pragma unroll
        for (int j = 0; j < FlTile<DT>::N; ++j) {
            int i = threadIdx.x + j * 256;
            int r = i / Q, c = (i - r * Q) * 4;
            bool used = i < FlCfg<DT>::TK * Q;
            row[j] = r;
            loff[j] = used ? r * FlCfg<DT>::S2 + c : -1;
            goff[j] = (used && c < d) ? (int)(((long)r * ld + c) * 4) : FL_OOB;           // d % 4 == 0
        }
    }
};
template <int DT>
__device__ __forceinline__ void fl_prefetch(float4 (&reg)[FlTile<DT>::N], __amdgpu_buffer_rsrc_t srd, const FlStream<DT>& st, int row0,
                                            int limit, long ld) {
    const int base = (int)((long)row0 * ld * 4);          // wave-uniform
    // with the fp32 MFMA every VALU instruction in the loop is MFMA time lost, so a full tile (the common case; wave-uniform test)
    // costs ONE add per load: padding slots carry the sentinel 2^31, which stays out of range when the base is added
    if (row0 + FlCfg<DT>::TK <= limit) {
#pragma unroll
        for (int j = 0; j < FlTile<DT>::N; ++j) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, (int)((unsigned)st.goff[j] + (unsigned)base), 0, 0);
            reg[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < FlTile<DT>::N; ++j) {
        bool ok = (st.goff[j] != FL_OOB) && (row0 + st.row[j] < limit);
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, ok ? st.goff[j] + base : FL_OOB, 0, 0);
        reg[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
}
template <int DT>
__device__ __forceinline__ void fl_store(float* T, const float4 (&reg)[FlTile<DT>::N], const FlStream<DT>& st) {
    constexpr bool all_used = (FlCfg<DT>::TK * FlTile<DT>::Q) % 256 == 0;     // every thread slot holds a real float4: no per-lane test
#pragma unroll
    for (int j = 0; j < FlTile<DT>::N; ++j)
        if (all_used || st.loff[j] >= 0) {
            *reinterpret_cast<float4*>(T + st.loff[j]) = reg[j];        // S2 and the column are multiples of 4: 16-byte aligned
        }
}

// ---- LDS-DMA staging (round 6; d <= 48, i.e. the 64x64-level UNet attention, d = 40): the streamed tiles go global -> LDS directly
// (`buffer_load ... lds`, 16 bytes per lane = 1 KiB per wave instruction) into TWO buffer pairs. Per tile the loop is
//     { s_waitcnt vmcnt(0); s_barrier;  issue the DMA of tile t+1 into the other pair;  compute tile t from this pair }
// — one barrier instead of two, no staging registers (12 .. 24 per thread), no ds_write instructions, no VALU address arithmetic: the
// per-lane source offsets are loop invariant and the tile's base rides in the instruction's scalar offset. The barrier orders both
// hazards: every wave's DMA of tile t has landed (RAW) and every wave has finished reading tile t-1, whose pair the next DMA overwrites
// (WAR). Measured on the 64x64-level self-attention (N = 4096, d = 40; profiles/README.md round 6): the register staging with its two
// barriers cost 73 of 704 us of the backward. The image is T[TK][S2] as before, S2 / 4 chunks of 16 bytes per row of which the last is
// the row padding (loaded out of range = zeros, like the head-dim padding columns); the last 1 KiB instruction may run past the TK rows, so a
// tile occupies TILE_BYTES = a whole number of KiB. Larger head dims keep the register staging: two pairs of their 21 KiB tiles do not fit
// two blocks per CU.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int DT> struct FlDma {
    static constexpr bool ON = DT <= 3;                               // d <= 48 (two pairs of the d = 64 tiles would need > 64 KiB of dynamic LDS)
    static constexpr int CPR = FlCfg<DT>::S2 / 4;                     // 16-byte chunks per LDS row
    static constexpr int NI = (FlCfg<DT>::TK * CPR + 63) / 64;        // 1 KiB instructions per tile
    static constexpr int NW = (NI + 3) / 4;                           // ... per wave (wave w issues instructions w, w + 4, ...)
    static constexpr int TILE_BYTES = NI * 1024;
    int off[NW];        // this lane's source byte offset inside the slice for instruction wave + 4 j; FL_OOB: padding chunk / beyond the tile
    int row[NW];
    __device__ __forceinline__ void init(long ld, int d, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int f = 64 * (wave + 4 * j) + lane, r = f / CPR, c = (f - r * CPR) * 4;
            row[j] = r;
            off[j] = (r < FlCfg<DT>::TK && c < d) ? (int)(((long)r * ld + c) * 4) : FL_OOB;
        }
    }
    // tile = LDS address of the tile image, wave = this wave's index (both wave-uniform)
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t srd, char* tile, int wave, int row0, int limit, long ld) const {
        const int base = (int)((long)row0 * ld * 4);
        const bool full = row0 + FlCfg<DT>::TK <= limit;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if (NI % 4 != 0 && j == NW - 1 && wave + 4 * j >= NI) break;                      // wave-uniform
            // (range checking looks at the vector offset alone: the FL_OOB sentinel stays out of range whatever the scalar base is)
            const int o = (full || row0 + row[j] < limit) ? off[j] : FL_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(tile + (wave + 4 * j) * 1024), 16, o, base, 0, 0);
        }
    }
};
#define FL_DMA_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)

// per-lane register copy of row `row` (B operand of a transposed MFMA): reg[s] = src[row][4*s + quarter] * mul, 0 beyond d / limit
template <int DT>
__device__ __forceinline__ void fl_load_rowregs(float (&reg)[FlCfg<DT>::KS], const float* __restrict__ src, long ld, int row, int limit,
                                                int d, int quarter, float mul) {
#pragma unroll
    for (int s = 0; s < FlCfg<DT>::KS; ++s) {
        int c = 4 * s + quarter;
        reg[s] = (row < limit && c < d) ? src[(long)row * ld + c] * mul : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- forward
template <int DT>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const FlashArgs a) {
    using C = FlCfg<DT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks0 = lds;                        // [TK][S2]   (register staging; the DMA form lays out two pairs of KiB-rounded images)
    float* Vs0 = lds + C::TK * C::S2;        // [TK][S2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, quarter = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * FL_ROWS;
    const int qrow = q0 + wave * 16 + l15;
    const float* Q = static_cast<const float*>(a.q) + b * a.sqb + h * a.sqh;
    const float* K = static_cast<const float*>(a.k) + b * a.skb + h * a.skh;
    const float* V = static_cast<const float*>(a.v) + b * a.skb + h * a.skh;

    // scores are kept in the log2 domain: q is pre-scaled by scale*log2(e) so that softmax needs a bare v_exp_f32
    float qreg[C::KS];
    fl_load_rowregs<DT>(qreg, Q, a.ldq, qrow, a.Nq, a.d, quarter, a.scale * 1.4426950408889634f);
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    int kbeg = 0, kend = a.Nk;
    if (a.causal) kend = min(a.Nk, q0 + FL_ROWS);      // keys beyond the last query of the block are fully masked
    if (a.ksplit > 1) {                                // blockIdx.z owns a key range (never combined with causal)
        const int kchunk = ((a.Nk + a.ksplit - 1) / a.ksplit + C::TK - 1) / C::TK * C::TK;
        kbeg = blockIdx.z * kchunk; kend = min(a.Nk, kbeg + kchunk);
    }
    const __amdgpu_buffer_rsrc_t srdK = fl_srd(K), srdV = fl_srd(V);
    auto tile = [&](const float* Ks, const float* Vs, int k0) {
            f32x4 s[C::NT];
#pragma unroll
            for (int kt = 0; kt < C::NT; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < C::KS; ++st)
#pragma unroll
                for (int kt = 0; kt < C::NT; ++kt) {
                    float av = Ks[(kt * 16 + l15) * C::S2 + 4 * st + quarter];
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, qreg[st], s[kt], 0, 0, 0);
                }
            // mask (only on the ragged last tile / causal diagonal: wave-uniform branch) + online softmax per lane quartet
            if (a.causal || k0 + C::TK > a.Nk) {
#pragma unroll
                for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int key = k0 + kt * 16 + quarter * 4 + r;
                        bool ok = key < a.Nk && (!a.causal || key <= qrow);
                        s[kt][r] = ok ? s[kt][r] : -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            mx = quad_max(mx);
            float mnew = fmaxf(m, mx);
            float muse = (mnew == -INFINITY) ? 0.f : mnew;
            float alpha = __builtin_amdgcn_exp2f(m - muse);            // m = -inf -> 0
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(s[kt][r] - muse);
                    s[kt][r] = p;
                    ps += p;
                }
            l = l * alpha + quad_sum(ps);
            m = mnew;
#pragma unroll
            for (int t = 0; t < DT; ++t) o[t] *= alpha;
            // O^T += V^T P^T
#pragma unroll
            for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        float av = Vs[(kt * 16 + quarter * 4 + r) * C::S2 + t * 16 + l15];
                        o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s[kt][r], o[t], 0, 0, 0);
                    }
    };
    if constexpr (FlDma<DT>::ON) {
        constexpr int TB = FlDma<DT>::TILE_BYTES;
        char* img = reinterpret_cast<char*>(lds);                  // [2 pairs][K tile, V tile]
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        FlDma<DT> dm; dm.init(a.ldk, a.d, wv, lane);
        if (kbeg < kend) {                                         // (never leave a DMA in flight behind the block: its LDS is handed on)
            dm.issue(srdK, img, wv, kbeg, a.Nk, a.ldk);
            dm.issue(srdV, img + TB, wv, kbeg, a.Nk, a.ldk);
        }
        int pair = 0;
        for (int k0 = kbeg; k0 < kend; k0 += C::TK, pair ^= 1) {
            FL_DMA_SYNC();
            if (k0 + C::TK < kend) {
                dm.issue(srdK, img + (pair ^ 1) * 2 * TB, wv, k0 + C::TK, a.Nk, a.ldk);
                dm.issue(srdV, img + (pair ^ 1) * 2 * TB + TB, wv, k0 + C::TK, a.Nk, a.ldk);
            }
            tile(reinterpret_cast<const float*>(img + pair * 2 * TB), reinterpret_cast<const float*>(img + pair * 2 * TB + TB), k0);
        }
    } else {
        FlStream<DT> st; st.init(a.ldk, a.d);
        float4 pk[FlTile<DT>::N], pv[FlTile<DT>::N];
        fl_prefetch<DT>(pk, srdK, st, kbeg, a.Nk, a.ldk);
        fl_prefetch<DT>(pv, srdV, st, kbeg, a.Nk, a.ldk);
        for (int k0 = kbeg; k0 < kend; k0 += C::TK) {
            __syncthreads();
            fl_store<DT>(Ks0, pk, st);
            fl_store<DT>(Vs0, pv, st);
            __syncthreads();
            if (k0 + C::TK < kend) {
                fl_prefetch<DT>(pk, srdK, st, k0 + C::TK, a.Nk, a.ldk);
                fl_prefetch<DT>(pv, srdV, st, k0 + C::TK, a.Nk, a.ldk);
            }
            tile(Ks0, Vs0, k0);
        }
    }
    if (a.ksplit > 1) {                                // partial: unnormalised O relative to the running max m, and (m, l)
        if (qrow < a.Nq) {
            const long BHN = (long)gridDim.y * a.Nq, row = (long)bh * a.Nq + qrow;
            float* W = a.wsk + ((long)blockIdx.z * BHN + row) * a.d;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                int c = t * 16 + quarter * 4;
                if (c < a.d) *reinterpret_cast<float4*>(W + c) = make_float4(o[t][0], o[t][1], o[t][2], o[t][3]);
            }
            if (quarter == 0) {
                float* ML = a.wsk + (long)a.ksplit * BHN * a.d + ((long)blockIdx.z * BHN + row) * 2;
                ML[0] = m; ML[1] = l;
            }
        }
        return;
    }
    if (qrow < a.Nq) {
        float inv = l > 0.f ? 1.f / l : 0.f;
        float* O = static_cast<float*>(a.o) + b * a.sob + h * a.soh + (long)qrow * a.ldo;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            int c = t * 16 + quarter * 4;
            if (c < a.d) *reinterpret_cast<float4*>(O + c) = make_float4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
        }
        if (quarter == 0 && a.lse) a.lse[(long)bh * a.Nq + qrow] = m + __builtin_amdgcn_logf(l);   // log2 domain
    }
}


// D[bh][q] = sum_d dO[q][d] * O[q][d] : one wave per query row
__global__ __launch_bounds__(256) void flash_dvec_kernel(const FlashArgs a, int BH) {
    long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (w >= (long)BH * a.Nq) return;
    int lane = threadIdx.x & 63;
    int bh = (int)(w / a.Nq), qi = (int)(w - (long)bh * a.Nq), b = bh / a.H, h = bh - b * a.H;
    const float* O = static_cast<const float*>(a.o) + b * a.sob + h * a.soh + (long)qi * a.ldo;
    const float* dO = static_cast<const float*>(a.dout) + b * a.sob + h * a.soh + (long)qi * a.ldo;
    float s = 0.f;
    for (int c = lane; c < a.d; c += 64) s += O[c] * dO[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) a.dvec[w] = s;
}

// ---------------------------------------------------------------------------------------------- backward: dQ
template <int DT>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const FlashArgs a) {
    using C = FlCfg<DT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks0 = lds;
    float* Vs0 = lds + C::TK * C::S2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, quarter = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = blockIdx.x * FL_ROWS;
    const int qrow = q0 + wave * 16 + l15;
    const float* Q = static_cast<const float*>(a.q) + b * a.sqb + h * a.sqh;
    const float* K = static_cast<const float*>(a.k) + b * a.skb + h * a.skh;
    const float* V = static_cast<const float*>(a.v) + b * a.skb + h * a.skh;
    const float* dO = static_cast<const float*>(a.dout) + b * a.sob + h * a.soh;

    float qreg[C::KS], dreg[C::KS];
    fl_load_rowregs<DT>(qreg, Q, a.ldq, qrow, a.Nq, a.d, quarter, a.scale * 1.4426950408889634f);
    fl_load_rowregs<DT>(dreg, dO, a.ldo, qrow, a.Nq, a.d, quarter, 1.f);
    const float lse = qrow < a.Nq ? a.lse[(long)bh * a.Nq + qrow] : 0.f;
    float Dq = 0.f;     // D = rowsum(dO * O), computed here and published for the dK/dV kernel that runs next
    {
        const float* O = static_cast<const float*>(a.o) + b * a.sob + h * a.soh;
#pragma unroll
        for (int s_ = 0; s_ < C::KS; ++s_) {
            int c = 4 * s_ + quarter;
            float ov = (qrow < a.Nq && c < a.d) ? O[(long)qrow * a.ldo + c] : 0.f;
            Dq += ov * dreg[s_];
        }
        Dq = quad_sum(Dq);
        if (quarter == 0 && qrow < a.Nq && blockIdx.z == 0) a.dvec[(long)bh * a.Nq + qrow] = Dq;
    }
    f32x4 g[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) g[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kbeg = 0, kend = a.Nk;
    if (a.causal) kend = min(a.Nk, q0 + FL_ROWS);
    if (a.ksplit > 1) {
        const int kchunk = ((a.Nk + a.ksplit - 1) / a.ksplit + C::TK - 1) / C::TK * C::TK;
        kbeg = blockIdx.z * kchunk; kend = min(a.Nk, kbeg + kchunk);
    }
    const __amdgpu_buffer_rsrc_t srdK = fl_srd(K), srdV = fl_srd(V);
    auto tile = [&](const float* Ks, const float* Vs, int k0) {
            // the ragged / causal tile is its own copy of the body: left to itself the compiler if-converts the wave-uniform `need_mask` and
            // every tile pays the per-element compares and selects (~9 VALU / SALU per score; fp32 MFMAs issue on the vector ALU's own
            // pipe, so VALU instructions in this loop are matrix time lost one for one)
            auto body = [&](auto masked) {
                constexpr bool MASK = decltype(masked)::value;
                f32x4 s[C::NT], dp[C::NT];
#pragma unroll
                for (int kt = 0; kt < C::NT; ++kt) { s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int st = 0; st < C::KS; ++st)
#pragma unroll
                    for (int kt = 0; kt < C::NT; ++kt) {
                        float ak = (FL_ABL & 4) ? dreg[st] : Ks[(kt * 16 + l15) * C::S2 + 4 * st + quarter];
                        float av = (FL_ABL & 4) ? qreg[st] : Vs[(kt * 16 + l15) * C::S2 + 4 * st + quarter];
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak, qreg[st], s[kt], 0, 0, 0);      // S^T  = K Q^T
                        dp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dreg[st], dp[kt], 0, 0, 0);    // dP^T = V dO^T
                    }
#pragma unroll
                for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (FL_ABL & 1) { s[kt][r] += dp[kt][r]; continue; }
                        float p = __builtin_amdgcn_exp2f(s[kt][r] - lse);
                        if constexpr (MASK) {
                            int key = k0 + kt * 16 + quarter * 4 + r;
                            bool ok = key < a.Nk && (!a.causal || key <= qrow);
                            p = ok ? p : 0.f;
                        }
                        s[kt][r] = p * (dp[kt][r] - Dq) * a.scale;                                        // d(q.k)
                    }
                // dQ^T += K^T dS^T
#pragma unroll
                for (int kt = 0; kt < C::NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < DT; ++t) {
                            float ak = (FL_ABL & 4) ? qreg[t] : Ks[(kt * 16 + quarter * 4 + r) * C::S2 + t * 16 + l15];
                            g[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ak, s[kt][r], g[t], 0, 0, 0);
                        }
            };
            if (a.causal || k0 + C::TK > a.Nk) body(std::true_type{}); else body(std::false_type{});     // wave-uniform
    };
    if constexpr (FlDma<DT>::ON) {
        constexpr int TB = FlDma<DT>::TILE_BYTES;
        char* img = reinterpret_cast<char*>(lds);                  // [2 pairs][K tile, V tile]
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        FlDma<DT> dm; dm.init(a.ldk, a.d, wv, lane);
        if (kbeg < kend) {                                         // (never leave a DMA in flight behind the block: its LDS is handed on)
            dm.issue(srdK, img, wv, kbeg, a.Nk, a.ldk);
            dm.issue(srdV, img + TB, wv, kbeg, a.Nk, a.ldk);
        }
        int pair = 0;
        for (int k0 = kbeg; k0 < kend; k0 += C::TK, pair ^= 1) {
            if (!(FL_ABL & 2) || k0 == kbeg) {
            FL_DMA_SYNC();
            if (k0 + C::TK < kend) {
                dm.issue(srdK, img + (pair ^ 1) * 2 * TB, wv, k0 + C::TK, a.Nk, a.ldk);
                dm.issue(srdV, img + (pair ^ 1) * 2 * TB + TB, wv, k0 + C::TK, a.Nk, a.ldk);
            }
            }
            tile(reinterpret_cast<const float*>(img + pair * 2 * TB), reinterpret_cast<const float*>(img + pair * 2 * TB + TB), k0);
        }
    } else {
        FlStream<DT> st; st.init(a.ldk, a.d);
        float4 pk[FlTile<DT>::N], pv[FlTile<DT>::N];
        fl_prefetch<DT>(pk, srdK, st, kbeg, a.Nk, a.ldk);
        fl_prefetch<DT>(pv, srdV, st, kbeg, a.Nk, a.ldk);
        for (int k0 = kbeg; k0 < kend; k0 += C::TK) {
            if (!(FL_ABL & 2) || k0 == kbeg) {
            __syncthreads();
            fl_store<DT>(Ks0, pk, st);
            fl_store<DT>(Vs0, pv, st);
            __syncthreads();
            if (k0 + C::TK < kend) {
                fl_prefetch<DT>(pk, srdK, st, k0 + C::TK, a.Nk, a.ldk);
                fl_prefetch<DT>(pv, srdV, st, k0 + C::TK, a.Nk, a.ldk);
            }
            }
            tile(Ks0, Vs0, k0);
        }
    }
    if (a.ksplit > 1) {
        if (qrow < a.Nq) {
            float* W = a.wsk + ((long)blockIdx.z * gridDim.y * a.Nq + (long)bh * a.Nq + qrow) * a.d;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                int c = t * 16 + quarter * 4;
                if (c < a.d) *reinterpret_cast<float4*>(W + c) = make_float4(g[t][0], g[t][1], g[t][2], g[t][3]);
            }
        }
        return;
    }
    if (qrow < a.Nq) {
        float* G = static_cast<float*>(a.dq) + b * a.sqb + h * a.sqh + (long)qrow * a.ldq;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            int c = t * 16 + quarter * 4;
            if (c < a.d) {
                float4 v = make_float4(g[t][0], g[t][1], g[t][2], g[t][3]);
                if (a.beta_q != 0.f) { float4 old = *reinterpret_cast<float4*>(G + c); v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
                *reinterpret_cast<float4*>(G + c) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward: dK, dV
// block owns 64 keys (wave: 16 keys, key = lane&15 is the MFMA column); streams Q / dO tiles.
//   S[q][key]    = Q K^T         A = Q tile (K-style), B = K^T regs (unscaled; scale applied after)
//   dP[q][key]   = dO V^T        A = dO tile (K-style), B = V^T regs
//   dV^T[d][key] += dO^T P       A = dO tile (V-style), B = P regs
//   dK^T[d][key] += Q^T dS       A = Q tile (V-style),  B = dS regs
template <int DT>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const FlashArgs a) {
    using C = FlCfg<DT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Qs0 = lds;                             // [TK][S2]   (register staging)
    float* Ds0 = lds + C::TK * C::S2;             // dO tile [TK][S2]
    float* Ls0 = lds + 2 * C::TK * C::S2;         // lse[TK], D[TK]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, quarter = lane >> 4;
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int k0 = blockIdx.x * FL_ROWS;
    const int krow = k0 + wave * 16 + l15;
    const float* Q = static_cast<const float*>(a.q) + b * a.sqb + h * a.sqh;
    const float* K = static_cast<const float*>(a.k) + b * a.skb + h * a.skh;
    const float* V = static_cast<const float*>(a.v) + b * a.skb + h * a.skh;
    const float* dO = static_cast<const float*>(a.dout) + b * a.sob + h * a.soh;

    float kreg[C::KS], vreg[C::KS];
    fl_load_rowregs<DT>(kreg, K, a.ldk, krow, a.Nk, a.d, quarter, 1.f);
    fl_load_rowregs<DT>(vreg, V, a.ldk, krow, a.Nk, a.d, quarter, 1.f);
    f32x4 gk[DT], gv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { gk[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; gv[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // cross-attention has few keys and many queries: blockIdx.z splits the query range, partial dK/dV go to a workspace
    const int qchunk = ((a.Nq + a.nsplit - 1) / a.nsplit + C::TK - 1) / C::TK * C::TK;
    int qbeg = blockIdx.z * qchunk;
    const int qlim = min(a.Nq, qbeg + qchunk);
    if (a.causal) qbeg = max(qbeg, (k0 / C::TK) * C::TK);   // queries before the first key of the block see none of its keys
    const __amdgpu_buffer_rsrc_t srdQ = fl_srd(Q), srdD = fl_srd(dO);
    const float scale2 = a.scale * 1.4426950408889634f;
    auto tile = [&](const float* Qs, const float* Ds, const float* Ls, int q0) {
            // rows of the C layout are queries 4*quarter + r of sub-tile qt, the column is this lane's key; masked tiles are a separate copy of the
            // body (see the dQ kernel)
            auto body = [&](auto masked) {
                constexpr bool MASK = decltype(masked)::value;
                f32x4 s[C::NT], dp[C::NT];
#pragma unroll
                for (int qt = 0; qt < C::NT; ++qt) { s[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int st = 0; st < C::KS; ++st)
#pragma unroll
                    for (int qt = 0; qt < C::NT; ++qt) {
                        float aq = (FL_ABL & 4) ? vreg[st] : Qs[(qt * 16 + l15) * C::S2 + 4 * st + quarter];
                        float ad = (FL_ABL & 4) ? kreg[st] : Ds[(qt * 16 + l15) * C::S2 + 4 * st + quarter];
                        s[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, kreg[st], s[qt], 0, 0, 0);
                        dp[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad, vreg[st], dp[qt], 0, 0, 0);
                    }
#pragma unroll
                for (int qt = 0; qt < C::NT; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int ql = qt * 16 + quarter * 4 + r, qi = q0 + ql;
                        if (FL_ABL & 1) continue;
                        float p = __builtin_amdgcn_exp2f(s[qt][r] * scale2 - Ls[ql]);
                        if constexpr (MASK) {
                            bool ok = qi < a.Nq && krow < a.Nk && (!a.causal || krow <= qi);
                            p = ok ? p : 0.f;
                        }
                        float ds = p * (dp[qt][r] - Ls[C::TK + ql]) * a.scale;
                        s[qt][r] = p;
                        dp[qt][r] = ds;
                    }
#pragma unroll
                for (int qt = 0; qt < C::NT; ++qt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < DT; ++t) {
                            float ad = (FL_ABL & 4) ? kreg[t] : Ds[(qt * 16 + quarter * 4 + r) * C::S2 + t * 16 + l15];
                            float aq = (FL_ABL & 4) ? vreg[t] : Qs[(qt * 16 + quarter * 4 + r) * C::S2 + t * 16 + l15];
                            gv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad, s[qt][r], gv[t], 0, 0, 0);
                            gk[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, dp[qt][r], gk[t], 0, 0, 0);
                        }
            };
            if (a.causal || q0 + C::TK > a.Nq || k0 + FL_ROWS > a.Nk) body(std::true_type{}); else body(std::false_type{});   // wave-uniform
    };
    if constexpr (FlDma<DT>::ON) {
        constexpr int TB = FlDma<DT>::TILE_BYTES;
        char* img = reinterpret_cast<char*>(lds);                  // [2 pairs][Q tile, dO tile], then [2][lse[TK], D[TK]]
        float* Lsb = reinterpret_cast<float*>(img + 4 * TB);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        FlDma<DT> dq_, do_; dq_.init(a.ldq, a.d, wv, lane); do_.init(a.ldo, a.d, wv, lane);
        // the per-query scalars of a tile: one 256-byte DMA each (wave 0: log-sum-exp, wave 1: D); rows beyond Nq read as zero (num_records)
        const __amdgpu_buffer_rsrc_t srdL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((wv == 0 ? a.lse : a.dvec) + (long)bh * a.Nq), 0, a.Nq * 4, 0x00020000);
        auto scalars = [&](int pair, int q0) {
            if (wv < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdL, (lds_ptr_t)(Lsb + pair * 2 * C::TK + wv * C::TK), 4, (q0 + lane) * 4, 0, 0, 0);
        };
        static_assert(C::TK == 64, "one DMA instruction per scalar row");
        if (qbeg < qlim) {                                         // (never leave a DMA in flight behind the block: its LDS is handed on)
            dq_.issue(srdQ, img, wv, qbeg, a.Nq, a.ldq);
            do_.issue(srdD, img + TB, wv, qbeg, a.Nq, a.ldo);
            scalars(0, qbeg);
        }
        int pair = 0;
        for (int q0 = qbeg; q0 < qlim; q0 += C::TK, pair ^= 1) {
            if (!(FL_ABL & 2) || q0 == qbeg) {
            FL_DMA_SYNC();
            if (q0 + C::TK < qlim) {
                dq_.issue(srdQ, img + (pair ^ 1) * 2 * TB, wv, q0 + C::TK, a.Nq, a.ldq);
                do_.issue(srdD, img + (pair ^ 1) * 2 * TB + TB, wv, q0 + C::TK, a.Nq, a.ldo);
                scalars(pair ^ 1, q0 + C::TK);
            }
            }
            tile(reinterpret_cast<const float*>(img + pair * 2 * TB), reinterpret_cast<const float*>(img + pair * 2 * TB + TB), Lsb + pair * 2 * C::TK, q0);
        }
    } else {
        FlStream<DT> stq, sto; stq.init(a.ldq, a.d); sto.init(a.ldo, a.d);
        float4 pq[FlTile<DT>::N], pd[FlTile<DT>::N];
        fl_prefetch<DT>(pq, srdQ, stq, qbeg, a.Nq, a.ldq);
        fl_prefetch<DT>(pd, srdD, sto, qbeg, a.Nq, a.ldo);
        for (int q0 = qbeg; q0 < qlim; q0 += C::TK) {
            if (!(FL_ABL & 2) || q0 == qbeg) {
            __syncthreads();
            fl_store<DT>(Qs0, pq, stq);
            fl_store<DT>(Ds0, pd, sto);
            if (threadIdx.x < C::TK) {
                int qi = q0 + threadIdx.x;
                Ls0[threadIdx.x] = qi < a.Nq ? a.lse[(long)bh * a.Nq + qi] : 0.f;
                Ls0[C::TK + threadIdx.x] = qi < a.Nq ? a.dvec[(long)bh * a.Nq + qi] : 0.f;
            }
            __syncthreads();
            if (q0 + C::TK < qlim) {
                fl_prefetch<DT>(pq, srdQ, stq, q0 + C::TK, a.Nq, a.ldq);
                fl_prefetch<DT>(pd, srdD, sto, q0 + C::TK, a.Nq, a.ldo);
            }
            }
            tile(Qs0, Ds0, Ls0, q0);
        }
    }
    if (a.nsplit > 1) {
        if (krow < a.Nk) {
            const long slab = (long)gridDim.y * a.Nk * a.d;
            float* WK = a.ws + ((long)blockIdx.z * gridDim.y + bh) * a.Nk * a.d + (long)krow * a.d;
            float* WV = WK + (long)a.nsplit * slab;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                int c = t * 16 + quarter * 4;
                if (c < a.d) {
                    *reinterpret_cast<float4*>(WK + c) = make_float4(gk[t][0], gk[t][1], gk[t][2], gk[t][3]);
                    *reinterpret_cast<float4*>(WV + c) = make_float4(gv[t][0], gv[t][1], gv[t][2], gv[t][3]);
                }
            }
        }
        return;
    }
    if (krow < a.Nk) {
        float* GK = static_cast<float*>(a.dk) + b * a.skb + h * a.skh + (long)krow * a.ldk;
        float* GV = static_cast<float*>(a.dv) + b * a.skb + h * a.skh + (long)krow * a.ldk;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            int c = t * 16 + quarter * 4;
            if (c < a.d) {
                float4 vk = make_float4(gk[t][0], gk[t][1], gk[t][2], gk[t][3]);
                float4 vv = make_float4(gv[t][0], gv[t][1], gv[t][2], gv[t][3]);
                if (a.beta_kv != 0.f) {
                    float4 ok = *reinterpret_cast<float4*>(GK + c), ov = *reinterpret_cast<float4*>(GV + c);
                    vk.x += ok.x; vk.y += ok.y; vk.z += ok.z; vk.w += ok.w;
                    vv.x += ov.x; vv.y += ov.y; vv.z += ov.z; vv.w += ov.w;
                }
                *reinterpret_cast<float4*>(GK + c) = vk;
                *reinterpret_cast<float4*>(GV + c) = vv;
            }
        }
    }
}

// sum the query-split partials: one thread per (bh, key, 4 head-dim columns), dK and dV
__global__ __launch_bounds__(256) void flash_dkv_reduce_kernel(const FlashArgs a, int BH) {
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
        float4 sk = make_float4(0.f, 0.f, 0.f, 0.f), sv = sk;
        for (int sp = 0; sp < a.nsplit; ++sp) {
            float4 x = *reinterpret_cast<const float4*>(wk + sp * slab), y = *reinterpret_cast<const float4*>(wv + sp * slab);
            sk.x += x.x; sk.y += x.y; sk.z += x.z; sk.w += x.w;
            sv.x += y.x; sv.y += y.y; sv.z += y.z; sv.w += y.w;
        }
        float* GK = static_cast<float*>(a.dk) + b * a.skb + h * a.skh + (long)key * a.ldk + c;
        float* GV = static_cast<float*>(a.dv) + b * a.skb + h * a.skh + (long)key * a.ldk + c;
        if (a.beta_kv != 0.f) {
            float4 ok = *reinterpret_cast<float4*>(GK), ov = *reinterpret_cast<float4*>(GV);
            sk.x += ok.x; sk.y += ok.y; sk.z += ok.z; sk.w += ok.w;
            sv.x += ov.x; sv.y += ov.y; sv.z += ov.z; sv.w += ov.w;
        }
        *reinterpret_cast<float4*>(GK) = sk;
        *reinterpret_cast<float4*>(GV) = sv;
    }
}

// key-split combines: one thread per (bh, query, 4 head-dim columns).
// forward: O = sum_i O_i 2^(m_i - M) / sum_i l_i 2^(m_i - M), lse = M + log2(sum)   (log2 domain, like the unsplit kernel)
__global__ __launch_bounds__(256) void flash_fwd_combine_kernel(const FlashArgs a, int BH) {
    const int dq4 = a.d / 4;
    const long BHN = (long)BH * a.Nq, total = BHN * dq4;
    const float* MLs = a.wsk + (long)a.ksplit * BHN * a.d;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % dq4) * 4;
        const long row = i / dq4;
        float M = -INFINITY;
        for (int sp = 0; sp < a.ksplit; ++sp) M = fmaxf(M, MLs[(sp * BHN + row) * 2]);
        const float Mu = (M == -INFINITY) ? 0.f : M;
        float Lsum = 0.f; float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < a.ksplit; ++sp) {
            const float* ml = MLs + (sp * BHN + row) * 2;
            const float w = __builtin_amdgcn_exp2f(ml[0] - Mu);             // m_i = -inf (empty range) -> 0
            Lsum += ml[1] * w;
            const float4 x = *reinterpret_cast<const float4*>(a.wsk + (sp * BHN + row) * a.d + c);
            acc.x += x.x * w; acc.y += x.y * w; acc.z += x.z * w; acc.w += x.w * w;
        }
        const float inv = Lsum > 0.f ? 1.f / Lsum : 0.f;
        const int bh = (int)(row / a.Nq), qi = (int)(row - (long)bh * a.Nq), b = bh / a.H, h = bh - b * a.H;
        float* O = static_cast<float*>(a.o) + b * a.sob + h * a.soh + (long)qi * a.ldo + c;
        *reinterpret_cast<float4*>(O) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if (c == 0 && a.lse) a.lse[row] = M + __builtin_amdgcn_logf(Lsum);
    }
}
__global__ __launch_bounds__(256) void flash_dq_combine_kernel(const FlashArgs a, int BH) {
    const int dq4 = a.d / 4;
    const long BHN = (long)BH * a.Nq, total = BHN * dq4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % dq4) * 4;
        const long row = i / dq4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < a.ksplit; ++sp) {
            const float4 x = *reinterpret_cast<const float4*>(a.wsk + (sp * BHN + row) * a.d + c);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        const int bh = (int)(row / a.Nq), qi = (int)(row - (long)bh * a.Nq), b = bh / a.H, h = bh - b * a.H;
        float* G = static_cast<float*>(a.dq) + b * a.sqb + h * a.sqh + (long)qi * a.ldq + c;
        if (a.beta_q != 0.f) { float4 old = *reinterpret_cast<float4*>(G); acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w; }
        *reinterpret_cast<float4*>(G) = acc;
    }
}

// ---------------------------------------------------------------------------------------------- host
static int fl_tk(int d) { return d > 80 ? 32 : 64; }           // FlCfg<DT>::TK of the variant fl_dt(d) picks
// key split of the forward / dQ kernels: aim at ~2 blocks per CU, every split at least two streamed tiles
size_t flash_ksplit_ws_floats(const FlashArgs& a, int B, bool forward, int* ksplit) {
    const int BH = B * a.H;
    const long blocks0 = (long)cdiv(a.Nq, FL_ROWS) * BH;
    int ks = 1;
    if (blocks0 < 256 && !a.causal) {
        ks = (int)((512 + blocks0 - 1) / blocks0);
        const int maxs = a.Nk / (2 * fl_tk(a.d));
        if (ks > maxs) ks = maxs;
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
    }
    if (getenv("CB_FLASH_KSPLIT")) ks = std::max(1, std::min(atoi(getenv("CB_FLASH_KSPLIT")), std::max(1, a.Nk / fl_tk(a.d))));
    if (a.causal) ks = 1;
    *ksplit = ks;
    return ks > 1 ? (size_t)ks * BH * a.Nq * (a.d + (forward ? 2 : 0)) : 0;
}

size_t flash_bwd_ws_floats(const FlashArgs& a, int B, int* nsplit) {
    const int BH = B * a.H;
    long blocks0 = (long)cdiv(a.Nk, FL_ROWS) * BH;
    int ns = 1;
    if (blocks0 < 256 && a.Nq >= 128 && !a.causal) {
        ns = (int)((512 + blocks0 - 1) / blocks0);
        int maxs = a.Nq / 64;                       // at least one 64-query tile (two 32-query tiles) per split
        if (ns > maxs) ns = maxs;
        if (ns < 1) ns = 1;
    }
    *nsplit = ns;
    return ns > 1 ? (size_t)2 * ns * BH * a.Nk * a.d : 0;
}

bool flash_supported(int d, const FlashArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool strides = (a.ldq % 4 == 0) && (a.ldk % 4 == 0) && (a.ldo % 4 == 0) && (a.sqb % 4 == 0) && (a.sqh % 4 == 0) &&
                   (a.skb % 4 == 0) && (a.skh % 4 == 0) && (a.sob % 4 == 0) && (a.soh % 4 == 0);
    const long lim = (1L << 31) - 64;
    bool small = (long)a.Nq * a.ldq * 4 < lim && (long)a.Nk * a.ldk * 4 < lim && (long)a.Nq * a.ldo * 4 < lim;   // 32-bit buffer offsets
    return d >= 4 && d <= 160 && d % 4 == 0 && strides && small && al(a.q) && al(a.k) && al(a.v) && al(a.o);
}

template <int DT> static size_t fl_lds_bytes(bool dkv) {
    using C = FlCfg<DT>;
    if (FlDma<DT>::ON) return (size_t)4 * FlDma<DT>::TILE_BYTES + (dkv ? 4 * C::TK * sizeof(float) : 0);       // two pairs of KiB-rounded images (+ two scalar rows pairs)
    return (size_t)(2 * C::TK * C::S2 + (dkv ? 2 * C::TK : 0)) * sizeof(float);
}

#define FL_DISPATCH(DTV, KERNEL, GRID, LDSB, ...)                                                                     \
    switch (DTV) {                                                                                                   \
        case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(256), LDSB(1), L.stream, __VA_ARGS__); break;               \
        case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), LDSB(2), L.stream, __VA_ARGS__); break;               \
        case 3: hipLaunchKernelGGL(KERNEL<3>, GRID, dim3(256), LDSB(3), L.stream, __VA_ARGS__); break;               \
        case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), LDSB(4), L.stream, __VA_ARGS__); break;               \
        case 5: hipLaunchKernelGGL(KERNEL<5>, GRID, dim3(256), LDSB(5), L.stream, __VA_ARGS__); break;               \
        default: hipLaunchKernelGGL(KERNEL<10>, GRID, dim3(256), LDSB(10), L.stream, __VA_ARGS__); break;            \
    }

static int fl_dt(int d) { int dt = (d + 15) / 16; if (dt == 3 && d > 40) dt = 4; return dt <= 5 ? dt : 10; }

static size_t lds_fwd(int dt) {
    switch (dt) { case 1: return fl_lds_bytes<1>(false); case 2: return fl_lds_bytes<2>(false); case 3: return fl_lds_bytes<3>(false);
                  case 4: return fl_lds_bytes<4>(false); case 5: return fl_lds_bytes<5>(false); default: return fl_lds_bytes<10>(false); }
}
static size_t lds_dkv(int dt) {
    switch (dt) { case 1: return fl_lds_bytes<1>(true); case 2: return fl_lds_bytes<2>(true); case 3: return fl_lds_bytes<3>(true);
                  case 4: return fl_lds_bytes<4>(true); case 5: return fl_lds_bytes<5>(true); default: return fl_lds_bytes<10>(true); }
}

void flash_forward(const Launch& L, const FlashArgs& a, int B) {
    const int BH = B * a.H, dt = fl_dt(a.d);
    if (L.stats) { L.stats->gemm_flops += 4.0 * BH * (double)a.Nq * a.Nk * a.d; L.stats->gemm_launches++; }
    L.count(a.ksplit > 1 ? 2 : 1);
    if (L.dry) return;
    CB_CHECK(a.ksplit == 1 || a.wsk, CB_ERR_STATE, "flash forward: key-split workspace missing");
    dim3 grid(cdiv(a.Nq, FL_ROWS), BH, a.ksplit);
    ProfRec rec; bool prof = L.prof && L.prof->on;
    if (prof) {
        rec.variant = L.prof->id_of("flash_fwd_kernel"); rec.flops = 4.0 * BH * (double)a.Nq * a.Nk * a.d;
        CB_HIP(hipEventCreate(&rec.e0)); CB_HIP(hipEventCreate(&rec.e1)); CB_HIP(hipEventRecord(rec.e0, L.stream));
    }
#define LDSF(x) lds_fwd(x)
    // Two experiments that did NOT move this kernel (profiles/README.md, round 2): threading the online softmax by hand through the
    // next tile's score MFMAs (one MFMA + one softmax slice per fenced step: 260 vs 262 us at N = 4096, d = 40) and doubling /
    // quadrupling the resident waves with a key split at N = 4096 (252 .. 257 us). The MFMA pipe stays 64 % busy either way.
    FL_DISPATCH(dt, flash_fwd_kernel, grid, LDSF, a)
    if (a.ksplit > 1) {
        long total = (long)BH * a.Nq * (a.d / 4);
        hipLaunchKernelGGL(flash_fwd_combine_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, L.stream, a, BH);
    }
    CB_HIP(hipGetLastError());
    if (prof) { CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec); }
}

void flash_backward(const Launch& L, const FlashArgs& a, int B, bool need_dq, bool need_dkv) {
    const int BH = B * a.H, dt = fl_dt(a.d);
    // algorithmic (reference-equivalent) work: dP, dQ | dK, dV — the recomputation of S inside both kernels is not counted
    double fl = 2.0 + (need_dq ? 2.0 : 0.0) + (need_dkv ? 4.0 : 0.0);
    if (L.stats) { L.stats->gemm_flops += fl * BH * (double)a.Nq * a.Nk * a.d; L.stats->gemm_launches++; }
    L.count(1 + (need_dq && a.ksplit > 1 ? 1 : 0) + (need_dkv ? (a.nsplit > 1 ? 2 : 1) : 0));
    if (L.dry) return;
    ProfRec rec; bool prof = L.prof && L.prof->on;
    if (prof) {
        rec.variant = L.prof->id_of("flash_bwd_kernels"); rec.flops = fl * BH * (double)a.Nq * a.Nk * a.d;
        CB_HIP(hipEventCreate(&rec.e0)); CB_HIP(hipEventCreate(&rec.e1)); CB_HIP(hipEventRecord(rec.e0, L.stream));
    }
    if (!need_dq) hipLaunchKernelGGL(flash_dvec_kernel, dim3(cdiv((long)BH * a.Nq, 4)), dim3(256), 0, L.stream, a, BH);
    if (need_dq) {
        CB_CHECK(a.ksplit == 1 || a.wsk, CB_ERR_STATE, "flash backward: key-split workspace missing");
        dim3 grid(cdiv(a.Nq, FL_ROWS), BH, a.ksplit);
        FL_DISPATCH(dt, flash_bwd_dq_kernel, grid, LDSF, a)
        if (a.ksplit > 1) {
            long total = (long)BH * a.Nq * (a.d / 4);
            hipLaunchKernelGGL(flash_dq_combine_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, L.stream, a, BH);
        }
    }
    if (need_dkv) {
        CB_CHECK(a.nsplit == 1 || a.ws, CB_ERR_STATE, "flash backward: query-split workspace missing");
        dim3 grid(cdiv(a.Nk, FL_ROWS), BH, a.nsplit);
#define LDSK(x) lds_dkv(x)
        FL_DISPATCH(dt, flash_bwd_dkv_kernel, grid, LDSK, a)
        if (a.nsplit > 1) {
            long total = (long)BH * a.Nk * (a.d / 4);
            hipLaunchKernelGGL(flash_dkv_reduce_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, L.stream, a, BH);
        }
    }
    CB_HIP(hipGetLastError());
    if (prof) { CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec); }
}

}  // namespace cb
