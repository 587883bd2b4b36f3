// cb_gemm_glds.hip — the LDS-DMA MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X): bf16 (BASELINE config 3) and fp32
// (config 2: every problem with two k-contiguous operands — all convolutions, Linear forwards, and Linear input gradients through
// the transposed weight copy the engine keeps) and the measured-plan autotuner.
//
//   C[M,N] = alpha * A[M,K] . B[N,K]^T + bias[n] + rowbias[img(m)][n] + R[m][n] + beta * C[m][n]
//
// A, B, C, R and rowbias are T (bf16 or fp32) in HBM, the accumulator is fp32 (v_mfma_f32_32x32x16_bf16: 2.5 PF dense chip peak;
// v_mfma_f32_32x32x2_f32: 157.3 TF), bias and the split-K partials are fp32. Both operands are k-contiguous (activations / conv
// gathers as A, [out][in] weights as B), 16-byte aligned with K a whole number of 16-byte chunks, so every load is one branch-free
// `buffer_load_dwordx4` with hardware zero fill. File layout:
//   1. `gemm_glds_kernel<T, ...>` / `gemm_glds256_kernel` — the LDS-DMA kernels. (Round 2's register-staged bf16 kernel, kept for a while
//      as the A/B baseline — 384 vs 525 TF over the config-3 mix, profiles/r02_bf16_gemm_variants.jsonl — is deleted.)
//   2. host side: variant selection, eligibility of fp32 problems, offline autotuner (CB_AUTOTUNE), launch + split-K reduce.
// profiles/README.md (round 2) has the measurements behind every choice.
#include "cb_gemm_tiles.h"

namespace cb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { AK_ROW = 0, AK_CONV = 1, AK_CONVF = 2 };

// ================================================================= the LDS-DMA kernel (bf16 AND fp32)
// 4 wave64 in 2x2 per tile (8 for the 256x128 tile), staging by `buffer_load_dwordx4 ... lds`: it writes the 16-byte chunks of a k-tile straight into LDS (no VGPR
// staging, no ds_write pass — at bf16 MFMA rates the VGPR->LDS store pass of the register-staged kernel above costs more
// LDS-pipe time than the MFMAs it feeds; at fp32 rates it removes the ~45 VALU + 16 ds_write per k-tile that compete with the
// MFMAs for issue slots, profiles/r01_pmc_conv.md), and a k-tile is one 128-byte row per tile row: 64 bf16 or 32 fp32.
// An LDS-DMA instruction writes wave-uniform base + lane * 16, i.e. the LDS image is lane-linear: thread slot idx = tid + i*256
// owns LDS chunk idx (row = idx >> 3, physical chunk = idx & 7). The XOR swizzle therefore goes on the SOURCE address: the lane
// fetches logical chunk (idx & 7) ^ ((row >> 1) & 7) of its row. With that, the 16-lane groups of the MFMA operand read
// (ds_read_b128: rows {0-3,12-15,20-27} / {4-11,16-19,28-31} of a 32-row block at one logical chunk) hit 16 distinct 16-byte
// bank quads. Out-of-range lanes (row / k tails, conv padding, stride-2 holes) use offset 0xFFFFFFF0: the DMA then writes zeros
// (tools/probes/probe_glds.hip).
// Element type T: one 16-byte chunk per lane and k-group feeds ONE v_mfma_f32_32x32x16_bf16 (8 bf16 = the lane's 8 k-slots) or
// FOUR v_mfma_f32_32x32x2_f32 (float e of the chunk is the lane's k-slot of step e: the contraction visits k in the order
// {4c + e} x {half-wave}, the same for A and B — any assignment of k to MFMA k-slots is valid when both operands agree).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct El;
template <> struct El<bf16_t> { static constexpr int ESZ = 2, EPC = 8; typedef bf16x8_t Frag; };
template <> struct El<float> { static constexpr int ESZ = 4, EPC = 4; typedef f32x4_t Frag; };

// geometry of the LDS-DMA image: rows of RB bytes = CPR 16-byte chunks; thread slot idx = tid + i*256 owns LDS chunk idx
template <int RB, int ESZ> struct GGeo {
    static constexpr int CPR = RB / 16;              // chunks per row: 4 (64-byte rows) or 8 (128-byte rows)
    static constexpr int LOG = CPR == 4 ? 2 : 3;
    static constexpr int BK = RB / ESZ;              // k elements per tile
    static constexpr int EPC = 16 / ESZ;             // elements per chunk
    __device__ static __forceinline__ int row(int idx) { return idx >> LOG; }
    __device__ static __forceinline__ int swz(int row) { return CPR == 4 ? ((row >> 2) & 3) : ((row >> 1) & 7); }
    __device__ static __forceinline__ int chunk(int idx) { return (idx & (CPR - 1)) ^ swz(idx >> LOG); }   // logical chunk fetched by slot idx
};

template <int ROWS, int RB, int ESZ, int NTHR = 256> struct GRows {        // ROW operand: per-slot byte offset of (row, swizzled chunk) or OOB
    using G = GGeo<RB, ESZ>;
    static constexpr int N = ROWS * G::CPR / NTHR;
    int off[N];
    __device__ __forceinline__ void init(long ld, int r0, int R, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = tid + i * NTHR, row = G::row(idx);
            off[i] = (r0 + row < R) ? (int)(((long)(r0 + row) * ld + G::chunk(idx) * G::EPC) * ESZ) : OOB;
        }
    }
};
template <int ROWS, int RB, int ESZ, int NTHR = 256> struct GConvFast {   // plain conv, C % BK == 0: a k-tile lies inside one tap (scalar tap walk)
    using G = GGeo<RB, ESZ>;
    static constexpr int N = ROWS * G::CPR / NTHR;
    int base[N]; unsigned vmask[N];
    __device__ __forceinline__ void init(const ConvGeom& g, int r0, int R, int tid) {
        const int HWo = g.Ho * g.Wo;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = tid + i * NTHR, r = r0 + G::row(idx);
            unsigned m = 0; int b = 0;
            if (r < R) {
                int n = r / HWo, rem = r - n * HWo, oy = rem / g.Wo, ox = rem - oy * g.Wo;
                int vy0 = oy * g.stride - g.pad, vx0 = ox * g.stride - g.pad;
                unsigned xb = 0, yb = 0;                             // validity of the ks x ks taps = (row bits) x (column bits), ks <= 4
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < g.ks) {
                        xb |= ((unsigned)(vx0 + k) < (unsigned)g.W) ? (1u << k) : 0u;
                        yb |= ((unsigned)(vy0 + k) < (unsigned)g.H) ? (1u << k) : 0u;
                    }
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
                    if (ky < g.ks && ((yb >> ky) & 1u)) m |= xb << (g.ks * ky);
                b = (((n * g.H + vy0) * g.W + vx0) * g.C + G::chunk(idx) * G::EPC) * ESZ;
            }
            vmask[i] = m; base[i] = b;
        }
    }
};
template <int ROWS, int RB, int ESZ, int NTHR = 256> struct GConvGen {    // general gather: per-slot (tap, channel) walk
    using G = GGeo<RB, ESZ>;
    static constexpr int N = ROWS * G::CPR / NTHR;
    int nb[N], vy0[N], vx0[N], tap[N], ci[N];
    __device__ __forceinline__ void init(const ConvGeom& g, int r0, int R, int kfirst, int tid) {
        const int HWo = g.Ho * g.Wo;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = tid + i * NTHR, r = r0 + G::row(idx);
            if (r < R) {
                int n = r / HWo, rem = r - n * HWo, oy = rem / g.Wo, ox = rem - oy * g.Wo;
                nb[i] = n * g.H * g.W; vy0[i] = oy * g.stride - g.pad; vx0[i] = ox * g.stride - g.pad;
            } else { nb[i] = -1; vy0[i] = 0; vx0[i] = 0; }
            const int k = kfirst + G::chunk(idx) * G::EPC;
            tap[i] = k / g.C; ci[i] = k - tap[i] * g.C;
        }
    }
    __device__ __forceinline__ void offs(int (&off)[N], const ConvGeom& g, int k0, int Kend, int tid) {
        const int sh = (g.up | g.tr), Hv = g.H << sh, Wv = g.W << sh;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int k = k0 + G::chunk(tid + i * NTHR) * G::EPC;
            const int ky = conv_tap_row(tap[i], g.ks), kx = tap[i] - ky * g.ks;
            const int vy = vy0[i] + ky, vx = vx0[i] + kx;
            bool ok = (k < Kend) && (nb[i] >= 0) && ((unsigned)vy < (unsigned)Hv) && ((unsigned)vx < (unsigned)Wv);
            ok = ok && !(g.tr && ((vy | vx) & 1));
            off[i] = ok ? ((nb[i] + (vy >> sh) * g.W + (vx >> sh)) * g.C + ci[i]) * ESZ : OOB;
            ci[i] += G::BK;                                   // advance to the next k-tile
            while (ci[i] >= g.C) { ci[i] -= g.C; ++tap[i]; }
        }
    }
};

// ---- LDS-staged epilogue. The MFMA C layout gives a lane ONE column and 16 scattered rows: written straight to HBM that is 64
// narrow stores per lane in 64/128-byte row pieces, which at K = 320 (the 16 384-row Linear layers) costs more than the whole
// k loop. Each wave therefore parks its (BM/2)x(BN/2) fp32 sub-tile in its own slice of the (now idle) tile ring — conflict-free
// ds_write_b32, lanes = consecutive columns — and reads it back as 8-column pieces: a lane then owns 8 consecutive outputs of
// one row, adds bias / per-image row bias / residual / beta*C with 16- and 32-byte loads, rounds ONCE to T and issues one
// (bf16) or two (fp32) 16-byte stores; 8 / 4 lanes cover a whole 128-byte line of C.
template <typename T, int BM, int BN, int WM = 2>
__device__ __forceinline__ void gemm_glds_epilogue(const KArgs& p, f32x16 (&acc)[BM / WM / 32][BN / 64], char* lds, int m0, int n0, int wm,
                                                   int wn, int lane, int wave, int split, int z, int zb, int zh) {
    constexpr int TM = BM / WM / 32, TN = BN / 64, WR = BM / WM, WC = BN / 2, CPRW = WC / 8, RPP = 64 / CPRW;
    const GemmArgs& a = p.a;
    const int kl = lane >> 5, ml = lane & 31;
    float* S = reinterpret_cast<float*>(lds) + wave * (WR * WC);
    __syncthreads();                                   // every wave is done reading the tile ring (and all DMAs have landed)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                S[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl) * WC + j * 32 + ml] = acc[i][j][r];
    __syncthreads();
    const int rsub = lane / CPRW, cc = (lane - rsub * CPRW) * 8;
    const int col = n0 + wn * WC + cc;
    const bool vec = (col + 8 <= a.N);
    float bcol[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bcol[e] = (a.bias && col + e < a.N) ? a.bias[col + e] : 0.f;
    const T* __restrict__ rowbias = static_cast<const T*>(a.rowbias);
    const long ldrb = a.ld_rowbias ? a.ld_rowbias : a.N;
    if (p.splitk > 1) {
        const long MN = (long)a.M * a.N;
        float* W = p.ws + ((long)split * a.batch + z) * MN;
        const bool v4 = vec && (a.N % 4 == 0);
        const __amdgpu_buffer_rsrc_t srdW = make_srd(W);
#pragma unroll
        for (int ps = 0; ps < WR / RPP; ++ps) {
            const int rl = ps * RPP + rsub, row = m0 + wm * WR + rl;
            if (row >= a.M || col >= a.N) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(S + rl * WC + cc), v1 = *reinterpret_cast<const float4*>(S + rl * WC + cc + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (p.cnt) { splitk_store8(srdW, W, (int)(((long)row * a.N + col) * 4), v, v4, a.N - col); continue; }     // host: M * N * 4 < 2 GiB
            float* w = W + (long)row * a.N + col;
            if (v4) { *reinterpret_cast<float4*>(w) = v0; *reinterpret_cast<float4*>(w + 4) = v1; }
            else for (int e = 0; e < 8 && col + e < a.N; ++e) w[e] = v[e];
        }
        if (!p.cnt) return;                               // partials only: splitk_reduce_kernel follows
        const int tile = (m0 / BM) * p.tiles_n + n0 / BN;
        if (!splitk_arrive(p.cnt + (long)z * (p.tiles_m * p.tiles_n) + tile, p.splitk, reinterpret_cast<int*>(lds))) return;
        T* __restrict__ Cq = static_cast<T*>(a.C) + zb * a.sCb + zh * a.sCh;
        const T* __restrict__ Rq = a.R ? static_cast<const T*>(a.R) + zb * a.sRb + zh * a.sRh : nullptr;
        const float* W0 = p.ws + (long)z * MN;
        for (int ps = 0; ps < WR / RPP; ++ps) {
            const int rl = ps * RPP + rsub, row = m0 + wm * WR + rl;
            if (row >= a.M || col >= a.N) continue;
            float v[8];
            splitk_fold8(v, W0 + (long)row * a.N + col, (long)a.batch * MN, p.splitk, v4, a.N - col);
            glds_finish_piece<T>(a, v, bcol, Cq, Rq, rowbias, ldrb, row, col, vec);
        }
        return;
    }
    T* __restrict__ Cp = static_cast<T*>(a.C) + zb * a.sCb + zh * a.sCh;
    const T* __restrict__ Rp = a.R ? static_cast<const T*>(a.R) + zb * a.sRb + zh * a.sRh : nullptr;
#pragma unroll
    for (int ps = 0; ps < WR / RPP; ++ps) {
        const int rl = ps * RPP + rsub, row = m0 + wm * WR + rl;
        if (row >= a.M || col >= a.N) continue;
        const float4 v0 = *reinterpret_cast<const float4*>(S + rl * WC + cc), v1 = *reinterpret_cast<const float4*>(S + rl * WC + cc + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        glds_finish_piece<T>(a, v, bcol, Cp, Rp, rowbias, ldrb, row, col, vec);
    }
}

// Pipeline: a ring of STAGES tile buffers with STAGES-1 tiles in flight. The LDS-DMA round trip (issue -> landed) is ~1.1 us on
// MI355X (MI355X_MICROARCH.md, ldsdma-fill) while the MFMAs of one 128x128 k-tile take 0.21 us (bf16, 64 k) or 1.9 us (fp32, 32 k:
// the fp32 kernel hides the whole round trip behind one tile's MFMAs with two stages). Every iteration issues exactly
// one tile (past the end of K the lanes are out of range and the DMA writes zeros into a slot nobody reads), so the number of
// outstanding DMA instructions is constant and "tile t has landed" is the counted wait vmcnt((STAGES-2) * loads per tile)
// followed by a raw s_barrier (a __syncthreads() would drain the whole queue with vmcnt(0)).
// FAST (two-stage rings, K a whole number of k-tiles, not the general gather): the k advance rides in the buffer instruction's SCALAR
// offset — excluded from the hardware range check, so the 0xFFFFFFF0 lanes of out-of-range rows keep reading zeros — and the conv
// gather's per-lane validity is re-evaluated only when the tap changes: a DMA piece then costs no vector instruction (5 before), and
// with fp32 MFMAs every VALU instruction of the k loop is matrix-pipe time lost (profiles/r04_pmc_n80.txt, README round 4).
template <typename T, int BM, int BN, int AK, int RB, int STAGES, int NTHR = 256, bool FAST = false>
__device__ __forceinline__ void gemm_glds_body(const KArgs& p, char* lds) {
    static_assert(!FAST || (STAGES == 2 && AK != AK_CONV), "FAST: vmcnt(0) rings, ROW or plain-conv A operand");
    constexpr int ESZ = El<T>::ESZ;
    using G = GGeo<RB, ESZ>;
    using Frag = typename El<T>::Frag;
    constexpr int WM = NTHR / 128;                       // waves along M (x 2 along N): 2 for the 4-wave tiles, 4 for the 8-wave 256x128 tile
    constexpr int TM = BM / WM / 32, TN = BN / 64, NA = BM * G::CPR / NTHR, NB = BN * G::CPR / NTHR, BKT = G::BK;
    const GemmArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y, z = blockIdx.z;
    const int zb = z / a.nh, zh = z - zb * a.nh;
    const T* __restrict__ Ap = static_cast<const T*>(a.A) + zb * a.sAb + zh * a.sAh;
    const T* __restrict__ Bp = static_cast<const T*>(a.B) + zb * a.sBb + zh * a.sBh;
    const int kbeg = split * p.k_per_split;
    const int kend = min(a.K, kbeg + p.k_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    GRows<BM, RB, ESZ, NTHR> ra; GRows<BN, RB, ESZ, NTHR> rb; GConvFast<BM, RB, ESZ, NTHR> cf; GConvGen<BM, RB, ESZ, NTHR> cg;
    int tap_u = 0, ci_u = 0;
    if constexpr (AK == AK_ROW) ra.init(a.lda, m0, a.M, tid);
    else if constexpr (AK == AK_CONVF) { cf.init(a.g, m0, a.M, tid); tap_u = kbeg / a.g.C; ci_u = kbeg - tap_u * a.g.C; }
    else cg.init(a.g, m0, a.M, kbeg, tid);
    rb.init(a.ldb, n0, a.N, tid);
    // per-slot k offset of the fetched chunk (loop invariant): the K tail test is one compare per slot
    int kca[NA], kcb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) kca[i] = G::chunk(tid + i * NTHR) * G::EPC;
#pragma unroll
    for (int i = 0; i < NB; ++i) kcb[i] = G::chunk(tid + i * NTHR) * G::EPC;
    // FAST conv: the range check sees voffset alone, so the most negative row base (pad up and left of image 0) moves into the descriptor
    const int abias = (FAST && AK == AK_CONVF) ? (a.g.pad * a.g.W + a.g.pad) * a.g.C * ESZ : 0;
    const __amdgpu_buffer_rsrc_t srdA = make_srd(reinterpret_cast<const char*>(Ap) - abias), srdB = make_srd(Bp);
    int a_vo[NA];                                     // FAST conv: the slot's current voffset = valid(tap) ? base + bias : OOB
    int delta_u = 0;                                  // FAST conv: byte offset of the current (tap, channel), scalar
    if constexpr (FAST && AK == AK_CONVF) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a_vo[i] = ((cf.vmask[i] >> tap_u) & 1u) ? cf.base[i] + abias : OOB;
        const int ky = conv_tap_row(tap_u, a.g.ks), kx = tap_u - ky * a.g.ks;
        delta_u = ((ky * a.g.W + kx) * a.g.C + ci_u) * ESZ;
    }
    constexpr int A_BUF = BM * RB, B_BUF = BN * RB;
    char* const As = lds;
    char* const Bs = lds + STAGES * A_BUF;
    // this wave's LDS-DMA destination of slot i: chunks (i*256 + 64*wave) .. +63 of the tile image
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 64 * 16);

    auto issue = [&](int k0, int buf) {
        if constexpr (FAST) {
            char* ad = As + buf * A_BUF + wbase;
            char* bd = Bs + buf * B_BUF + wbase;
            if constexpr (AK == AK_ROW) {
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(ad + i * (NTHR * 16)), 16, ra.off[i], k0 * ESZ, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(ad + i * (NTHR * 16)), 16, a_vo[i], delta_u, 0, 0);
                ci_u += BKT; delta_u += BKT * ESZ;                         // the (tap, channel) byte offset walks with the k-tiles ...
                if (ci_u >= a.g.C) {                                       // ... and is re-derived, like the lanes' validity, only at a tap change
                    ci_u -= a.g.C; ++tap_u;
                    const int ky = conv_tap_row(tap_u, a.g.ks), kx = tap_u - ky * a.g.ks;
                    delta_u = ((ky * a.g.W + kx) * a.g.C + ci_u) * ESZ;
#pragma unroll
                    for (int i = 0; i < NA; ++i) a_vo[i] = ((cf.vmask[i] >> (tap_u & 31)) & 1u) ? cf.base[i] + abias : OOB;
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(bd + i * (NTHR * 16)), 16, rb.off[i], k0 * ESZ, 0, 0);
            return;
        }
        int oa[NA], ob[NB];
        if constexpr (AK == AK_ROW) {
#pragma unroll
            for (int i = 0; i < NA; ++i) oa[i] = (ra.off[i] != OOB && k0 + kca[i] < kend) ? ra.off[i] + k0 * ESZ : OOB;
        } else if constexpr (AK == AK_CONVF) {
            const int ky = conv_tap_row(tap_u, a.g.ks), kx = tap_u - ky * a.g.ks;
            const int delta = ((ky * a.g.W + kx) * a.g.C + ci_u) * ESZ;
            const bool in_k = k0 < kend;                       // C % BK == 0 and K = taps * C: a tile is entirely inside or outside K
#pragma unroll
            for (int i = 0; i < NA; ++i) oa[i] = (in_k && ((cf.vmask[i] >> tap_u) & 1u)) ? cf.base[i] + delta : OOB;
            ci_u += BKT;
            if (ci_u >= a.g.C) { ci_u -= a.g.C; ++tap_u; }
        } else cg.offs(oa, a.g, k0, kend, tid);
#pragma unroll
        for (int i = 0; i < NB; ++i) ob[i] = (rb.off[i] != OOB && k0 + kcb[i] < kend) ? rb.off[i] + k0 * ESZ : OOB;
        char* ad = As + buf * A_BUF + wbase;
        char* bd = Bs + buf * B_BUF + wbase;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(ad + i * (NTHR * 16)), 16, oa[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(bd + i * (NTHR * 16)), 16, ob[i], 0, 0, 0);
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue(kbeg + s * BKT, s);          // prologue: STAGES-1 tiles in flight

    int cur = 0, nxt = STAGES - 1;
    const int kl = lane >> 5, ml = lane & 31;
    const int sw = G::swz(ml);                                     // sub-tile bases are multiples of 32: swz(row) == swz(ml)
    for (int k0 = kbeg; k0 < kend; k0 += BKT) {
        // tile k0 has landed for this wave (all but the newest STAGES-2 tiles are complete) ... and for every wave after the barrier,
        // which also says that everybody is done reading the slot that is refilled next (it was read in the previous iteration)
        // The count is exact only while every k-tile issues NA + NB VMEM instructions and the compiler adds none of its own (spills,
        // re-materialised loads): celebbasis_amd/build.py fails the build if any kernel of this file has scratch. p.safe (CB_GLDS_SAFE=1
        // / cb_debug_glds_safe) is the reference pipeline: drain the queue, full barrier — bitwise the same results
        // (tests/test_gpu_ops.py::test_glds_counted_wait_matches_safe_mode).
        if (p.safe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        else { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (NA + NB)) : "memory"); __builtin_amdgcn_s_barrier(); }
        // (FAST has no per-lane k limit: the tile past the end of K is not issued; its ring wait is vmcnt(0), nothing counts pieces)
        if (!FAST || k0 + (STAGES - 1) * BKT < kend) issue(k0 + (STAGES - 1) * BKT, nxt);
        const char* arow = As + cur * A_BUF + (wm * (BM / WM) + ml) * RB;
        const char* brow = Bs + cur * B_BUF + (wn * (BN / 2) + ml) * RB;
#pragma unroll
        for (int s = 0; s < G::CPR / 2; ++s) {                     // one 16-byte chunk per half-wave: 16 bf16 / 8 fp32 of k per step
            const int co = ((2 * s + kl) ^ sw) * 16;
            Frag av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const Frag*>(arow + i * 32 * RB + co);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const Frag*>(brow + j * 32 * RB + co);
            if constexpr (ESZ == 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
            }
        }
        cur = (cur + 1 == STAGES) ? 0 : cur + 1;
        nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-fill tiles issued past the end of K
    gemm_glds_epilogue<T, BM, BN, WM>(p, acc, lds, m0, n0, wm, wn, lane, wave, split, z, zb, zh);
}

// 8-wave variant: 256x128 tile (4 x 2 waves of 64x64), 128-byte rows, THREE stages = 144 KB of LDS = one block per CU with two
// k-tiles in flight and 25 % fewer operand bytes per FLOP than the 128x128 tile. For the bf16 problems large enough to fill the
// chip with it (the planner / plan table decide); dynamic LDS (> 64 KB).
template <typename T, int AK>
__global__ __launch_bounds__(512) void gemm_glds256_kernel(const KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds256[];
    gemm_glds_body<T, 256, 128, AK, 128, 3, 512>(p, lds256);
}
constexpr int GLDS256_LDS = 3 * (256 + 128) * 128;

template <typename T, int BM, int BN, int AK, int RB, int STAGES, bool FAST = false>
__global__ __launch_bounds__(256) void gemm_glds_kernel(const KArgs p) {
    constexpr int RING = STAGES * (BM + BN) * RB, CTILE = BM * BN * 4;        // the epilogue parks the fp32 output tile in the ring
    __shared__ __attribute__((aligned(16))) char lds[RING > CTILE ? RING : CTILE];
    gemm_glds_body<T, BM, BN, AK, RB, STAGES, 256, FAST>(p, lds);
}


// ================================================================= the 80-column tile (fp32): SD v1-4's widths are 320 * 2^j
// Every Linear / conv output of the UNet is 320, 640, 1280 (or a small multiple) columns wide and, at bs = 1, 4096 / 1024 / 256 / 64
// rows tall. In 64-wide tiles that is 5 * 2^j column tiles: 320 blocks at 64x64 on 256 CUs put two tiles on 64 of them (the critical
// CU runs 2.0 tile-times for 1.25 tiles of work), and the 32x32 ... 8x8 levels need a split-K with its reduce launch to fill the
// chip at all. An 80-column tile makes the column count 4 * 2^j, and the block's four waves trade rows for k-groups as M shrinks:
//      BM = 64: 4 waves x 16 rows, one k-group      4096 x 320 -> 64 x 4 = 256 blocks
//      BM = 32: 2 x 16 rows, 2 k-groups             1024 x 640 -> 32 x 8 = 256 blocks
//      BM = 16: 1 x 16 rows, 4 k-groups              256 x 1280 -> 16 x 16 = 256 blocks
// i.e. one block per CU, one wave per SIMD, at every level, with the k-split INSIDE the block: the KW partial strips meet in LDS (fixed
// order: deterministic), no partials in HBM, no reduce launch. A wave owns a 16 x 80 strip = five 16x16 accumulators of
// v_mfma_f32_16x16x4_f32 (same FLOP rate as 32x32x2; one A fragment feeds five MFMAs), k-group g walks its own contiguous slice of K.
// Staging is the LDS-DMA ring of the kernel above (128-byte rows, source-side XOR swizzle, counted vmcnt): the A image is 64 rows
// (KW groups x BM rows), the B image KW x 80 rows padded to whole 256-chunk slots. A 16-lane group of the ds_read_b128 fragment read
// covers rows {0-3, 12-15} at logical chunk c and rows {4-11} at chunk c ^ 1: with the (row >> 1) & 7 swizzle 16 distinct bank quads.
typedef float f32x4a __attribute__((ext_vector_type(4)));
template <int BM, int STAGES> struct G80 {
    static constexpr int KW = 64 / BM, WMV = BM / 16, NA = 2, BROWS = KW * 80, NB = (BROWS * 8 + 255) / 256;
    static constexpr int A_BUF = 64 * 128, B_BUF = NB * 256 * 16, STAGE = A_BUF + B_BUF, LDS = STAGES * STAGE, SLD = 84;
    static_assert(64 * SLD * 4 <= LDS, "the epilogue parks KW x BM x 80 fp32 partials in the ring");
};

// FAST: every k-group of every split owns the same whole number of k-tiles (host: (k range) % (32 * KW) == 0 — all SD v1-4 shapes at
// BM = 64 / 32, most at 16). The k advance of a ROW operand then rides in the buffer instruction's SCALAR offset (excluded from the
// range check, so the 0xFFFFFFF0 lanes of out-of-range rows still read zeros) and a piece costs no VALU at all: with fp32 MFMAs every
// vector instruction in the k loop is matrix-pipe time lost (profiles/r04_pmc_n80.txt: 30 - 50 VALU per 40 MFMAs before this).
template <int BM, int AK, int STAGES, bool FAST>
__device__ __forceinline__ void gemm_glds80_body(const KArgs& p, char* lds80) {
    using Q = G80<BM, STAGES>;
    using G = GGeo<128, 4>;
    constexpr int KW = Q::KW, WMV = Q::WMV, NA = Q::NA, NB = Q::NB, BKT = 32;
    const GemmArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * 80;
    const int split = blockIdx.y;
    const float* __restrict__ Ap = static_cast<const float*>(a.A);
    const float* __restrict__ Bp = static_cast<const float*>(a.B);
    const int kbeg = split * p.k_per_split;
    const int kend = min(a.K, kbeg + p.k_per_split);
    const int ktiles = (kend - kbeg + BKT - 1) / BKT;
    const int ktg = (ktiles + KW - 1) / KW;                 // k-tiles per group (the last group may get fewer: zero tiles past its end)

    // ---- per-slot source state. Slot i of a thread is chunk idx = tid + i * 256 of the stage image (lane-linear LDS-DMA destination).
    int a_off[NA], a_klim[NA], b_off[NB], b_klim[NB];
    unsigned a_vmask[NA]; int a_tap[NA], a_ci[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = tid + i * 256, irow = idx >> 3;
        const int g = __builtin_amdgcn_readfirstlane(((wave * 64 + i * 256) >> 3) / BM);      // wave-uniform: 64 lanes = 8 image rows
        const int r = irow - g * BM, lc = G::chunk(idx) * 4;
        const int kg0 = kbeg + g * ktg * BKT, kge = min(kend, kg0 + ktg * BKT);
        a_klim[i] = kge - kg0 - lc;
        a_vmask[i] = 0; a_tap[i] = 0; a_ci[i] = 0;
        const int row = m0 + r;
        if constexpr (AK == AK_ROW) {
            a_off[i] = (row < a.M) ? (int)(((long)row * a.lda + kg0 + lc) * 4) : OOB;
        } else {
            const ConvGeom& cg = a.g;
            unsigned m = 0; int b = 0;
            if (row < a.M) {
                const int HWo = cg.Ho * cg.Wo;
                const int n = row / HWo, rem = row - n * HWo, oy = rem / cg.Wo, ox = rem - oy * cg.Wo;
                const int vy0 = oy * cg.stride - cg.pad, vx0 = ox * cg.stride - cg.pad;
                unsigned xb = 0, yb = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cg.ks) {
                        xb |= ((unsigned)(vx0 + k) < (unsigned)cg.W) ? (1u << k) : 0u;
                        yb |= ((unsigned)(vy0 + k) < (unsigned)cg.H) ? (1u << k) : 0u;
                    }
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
                    if (ky < cg.ks && ((yb >> ky) & 1u)) m |= xb << (cg.ks * ky);
                b = (((n * cg.H + vy0) * cg.W + vx0) * cg.C + lc) * 4;
            }
            a_vmask[i] = m; a_off[i] = b;
            a_tap[i] = kg0 / cg.C; a_ci[i] = kg0 - a_tap[i] * cg.C;          // scalar (g is wave-uniform): the group's tap walk
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * 256, irow = idx >> 3;
        const int g = irow / 80, r = irow - g * 80, lc = G::chunk(idx) * 4;
        const int kg0 = kbeg + g * ktg * BKT, kge = min(kend, kg0 + ktg * BKT);
        const bool ok = irow < Q::BROWS && n0 + r < a.N;
        b_klim[i] = ok ? kge - kg0 - lc : 0;
        b_off[i] = ok ? (int)(((long)(n0 + r) * a.ldb + kg0 + lc) * 4) : OOB;
    }
    // FAST conv gather: voffset = the row's (tap-independent) base, soffset = the tap / channel walk. The range check sees voffset alone,
    // so it must not be negative for a valid lane: the most negative base (row 0, pad up and left) is folded into the descriptor's base.
    const int abias = (FAST && AK != AK_ROW) ? (a.g.pad * a.g.W + a.g.pad) * a.g.C * 4 : 0;
    const __amdgpu_buffer_rsrc_t srdA = make_srd(reinterpret_cast<const char*>(Ap) - abias), srdB = make_srd(Bp);
    const int wbase = __builtin_amdgcn_readfirstlane(wave * 64 * 16);
    int a_vo[NA], a_dl[NA];                                  // FAST conv: current voffset of the slot = valid(tap) ? base + bias : OOB; tap/channel offset
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        a_vo[i] = ((a_vmask[i] >> (a_tap[i] & 31)) & 1u) ? a_off[i] + abias : OOB;
        const int ky = conv_tap_row(a_tap[i], a.g.ks), kx = a_tap[i] - ky * a.g.ks;
        a_dl[i] = (AK == AK_ROW) ? 0 : ((ky * a.g.W + kx) * a.g.C + a_ci[i]) * 4;
    }

    // one LDS-DMA piece (slot `pc`: A slots first, then B) of k-tile kt -> ring slot `stage`
    auto piece = [&](int pc, int kt, char* stage) {
        const int ko = kt * BKT;
        if (pc < NA) {
            const int i = pc;
            int oa;
            if constexpr (AK == AK_ROW && FAST) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(stage + wbase + i * 4096), 16, a_off[i], ko * 4, 0, 0);
                return;
            } else if constexpr (AK == AK_ROW) oa = (a_off[i] != OOB && ko < a_klim[i]) ? a_off[i] + ko * 4 : OOB;
            else if constexpr (FAST) {
                // a_dl = byte offset of (tap, channel) — scalar (the k-group is wave-uniform), advanced by one k-tile per piece and
                // re-derived only when the tap changes, like the per-lane validity: two scalar adds and a compare per tile otherwise
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(stage + wbase + i * 4096), 16, a_vo[i], a_dl[i], 0, 0);
                a_ci[i] += BKT; a_dl[i] += BKT * 4;
                if (a_ci[i] >= a.g.C) {
                    a_ci[i] -= a.g.C; ++a_tap[i];
                    const int ky = conv_tap_row(a_tap[i], a.g.ks), kx = a_tap[i] - ky * a.g.ks;
                    a_dl[i] = ((ky * a.g.W + kx) * a.g.C + a_ci[i]) * 4;
                    a_vo[i] = ((a_vmask[i] >> (a_tap[i] & 31)) & 1u) ? a_off[i] + abias : OOB;
                }
                return;
            } else {
                const int ky = conv_tap_row(a_tap[i], a.g.ks), kx = a_tap[i] - ky * a.g.ks;
                const int delta = ((ky * a.g.W + kx) * a.g.C + a_ci[i]) * 4;
                oa = (ko < a_klim[i] && ((a_vmask[i] >> (a_tap[i] & 31)) & 1u)) ? a_off[i] + delta : OOB;
                a_ci[i] += BKT;
                if (a_ci[i] >= a.g.C) { a_ci[i] -= a.g.C; ++a_tap[i]; }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(stage + wbase + i * 4096), 16, oa, 0, 0, 0);
        } else {
            const int i = pc - NA;
            if constexpr (FAST) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(stage + Q::A_BUF + wbase + i * 4096), 16, b_off[i], ko * 4, 0, 0);
                return;
            }
            const int ob = (ko < b_klim[i]) ? b_off[i] + ko * 4 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(stage + Q::A_BUF + wbase + i * 4096), 16, ob, 0, 0, 0);
        }
    };

    f32x4a acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = f32x4a{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (!FAST || s < ktg) {                               // (FAST pieces carry no k limit: a tile past the group's end is never issued)
#pragma unroll
            for (int pc = 0; pc < NA + NB; ++pc) piece(pc, s, lds80 + s * Q::STAGE);
        }

    const int g = wave / WMV, wm = wave - g * WMV;
    const int lg = lane >> 4, lr = lane & 15, sw = lr >> 1;
    const int arow_off = (g * BM + wm * 16 + lr) * 128, brow_off = Q::A_BUF + (g * 80 + lr) * 128;
    static_assert(STAGES == 2 || STAGES == 3, "ring slots are compile-time constants: the k loop is unrolled by STAGES");
    // one k-tile; `slot` = kt % STAGES is a compile-time constant at every call site (the loop runs in groups of STAGES), so every LDS
    // address of the tile is an immediate offset: no vector adds for the fragment reads, no scalar ring bookkeeping.
    // STAGES = 3 puts TWO k-tiles in flight per CU (round-6 experiment for the launches with at most one block per CU, measured flat:
    // see launch_n80_go).
    auto tile = [&](int kt, auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        // tile kt has landed when all but the newest STAGES - 2 tiles are complete; the tiles past the group's end are not issued in the
        // FAST form, so near the end fewer are outstanding and the wait is a full drain (wave-uniform branch)
        if (p.safe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        else {
            if (STAGES == 3 && (!FAST || kt + 1 < ktg)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const char* st = lds80 + slot * Q::STAGE;
        // the next tile's pieces go out in one clump: spreading them between the MFMAs was measured and changes nothing (profiles/README.md,
        // round 4). FAST has no per-lane k limit: the tile past the group's end is simply not issued
        if (!FAST || kt + STAGES - 1 < ktg) {
            constexpr int nslot = (slot + STAGES - 1) % STAGES;           // the slot read in the previous iteration: every wave is past its reads
#pragma unroll
            for (int pc = 0; pc < NA + NB; ++pc) piece(pc, kt + STAGES - 1, lds80 + nslot * Q::STAGE);
        }
        // all twelve fragment reads of the k-tile first (chunk 4s + lg is the lane's k-slot quadruple of half tile s), then the 40 MFMAs:
        // with one wave per SIMD nobody else covers the LDS latency, so it is paid once per k-tile instead of once per half tile
        f32x4a av[2], bv[2][5];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int co = ((4 * s + lg) ^ sw) * 16;
            av[s] = *reinterpret_cast<const f32x4a*>(st + arow_off + co);
#pragma unroll
            for (int j = 0; j < 5; ++j) bv[s][j] = *reinterpret_cast<const f32x4a*>(st + brow_off + j * 2048 + co);
        }
        __builtin_amdgcn_sched_barrier(0);                    // keep the reads ahead of the MFMAs (the scheduler sinks half of them otherwise)
#pragma unroll
        for (int q = 0; q < 40; ++q) {
            const int s = q / 20, e = (q / 5) & 3, j = q % 5;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][e], bv[s][j][e], acc[j], 0, 0, 0);
        }
    };
    int kt = 0;
    if constexpr (STAGES == 2) {
        for (; kt + 1 < ktg; kt += 2) { tile(kt, std::integral_constant<int, 0>{}); tile(kt + 1, std::integral_constant<int, 1>{}); }
        if (kt < ktg) tile(kt, std::integral_constant<int, 0>{});
    } else {
        for (; kt + 2 < ktg; kt += 3) {
            tile(kt, std::integral_constant<int, 0>{}); tile(kt + 1, std::integral_constant<int, 1>{}); tile(kt + 2, std::integral_constant<int, 2>{});
        }
        if (kt < ktg) tile(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < ktg) tile(kt + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- the KW partial strips meet in LDS; D layout of 16x16x4: lane holds column lr, rows 4 * lg + r
    float* S = reinterpret_cast<float*>(lds80);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(g * BM + wm * 16 + 4 * lg + r) * Q::SLD + j * 16 + lr] = acc[j][r];
    __syncthreads();
    float* __restrict__ Cp = static_cast<float*>(a.C);
    const float* __restrict__ Rp = static_cast<const float*>(a.R);
    const float* __restrict__ rowbias = static_cast<const float*>(a.rowbias);
    const long ldrb = a.ld_rowbias ? a.ld_rowbias : a.N;
    const long MN = (long)a.M * a.N;
    for (int q = tid; q < BM * 10; q += 256) {
        const int rl = q / 10, cc = (q - rl * 10) * 8;
        const int row = m0 + rl, col = n0 + cc;
        if (row >= a.M || col >= a.N) continue;
        float v[8];
        {
            const float4 v0 = *reinterpret_cast<const float4*>(S + rl * Q::SLD + cc), v1 = *reinterpret_cast<const float4*>(S + rl * Q::SLD + cc + 4);
            v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        }
#pragma unroll
        for (int gg = 1; gg < KW; ++gg) {
            const float* sp = S + (gg * BM + rl) * Q::SLD + cc;
            const float4 v0 = *reinterpret_cast<const float4*>(sp), v1 = *reinterpret_cast<const float4*>(sp + 4);
            v[0] += v0.x; v[1] += v0.y; v[2] += v0.z; v[3] += v0.w; v[4] += v1.x; v[5] += v1.y; v[6] += v1.z; v[7] += v1.w;
        }
        const bool vec = (col + 8 <= a.N);
        if (p.splitk > 1) {
            float* w = p.ws + (long)split * MN + (long)row * a.N + col;
            if (p.cnt) { splitk_store8(make_srd(p.ws + (long)split * MN), p.ws + (long)split * MN, (int)(((long)row * a.N + col) * 4), v, vec && (a.N % 4 == 0), a.N - col); continue; }
            if (vec && (a.N % 4 == 0)) {
                *reinterpret_cast<float4*>(w) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(w + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else
                for (int e = 0; e < 8 && col + e < a.N; ++e) w[e] = v[e];
            continue;
        }
        float bcol[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bcol[e] = (a.bias && col + e < a.N) ? a.bias[col + e] : 0.f;
        glds_finish_piece<float>(a, v, bcol, Cp, Rp, rowbias, ldrb, row, col, vec);
    }
    if (p.splitk <= 1 || !p.cnt) return;                  // (no counters: partials only, splitk_reduce_kernel follows)
    // in-launch reduction: the tile's last-arriving block folds the splitk slabs in split order (cb_gemm_tiles.h, splitk_arrive)
    if (!splitk_arrive(p.cnt + tm * p.tiles_n + tn, p.splitk, reinterpret_cast<int*>(lds80))) return;
    for (int q = tid; q < BM * 10; q += 256) {
        const int rl = q / 10, cc = (q - rl * 10) * 8;
        const int row = m0 + rl, col = n0 + cc;
        if (row >= a.M || col >= a.N) continue;
        const bool vec = (col + 8 <= a.N);
        float v[8], bcol[8];
        splitk_fold8(v, p.ws + (long)row * a.N + col, MN, p.splitk, vec && (a.N % 4 == 0), a.N - col);
#pragma unroll
        for (int e = 0; e < 8; ++e) bcol[e] = (a.bias && col + e < a.N) ? a.bias[col + e] : 0.f;
        glds_finish_piece<float>(a, v, bcol, Cp, Rp, rowbias, ldrb, row, col, vec);
    }
}
// (the body lives in a __device__ function: target builtins used directly inside a __global__ template make the HOST pass drop the stub)
template <int BM, int AK, int STAGES, bool FAST>
__global__ __launch_bounds__(256) void gemm_glds80_kernel(const KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds80_dyn[];
    gemm_glds80_body<BM, AK, STAGES, FAST>(p, lds80_dyn);
}

// ---------------------------------------------------------------- host side
// Variant selection (experiments): CB_BF16_GEMM = 128x2 | 128x3 | 64x2 | 64x3  (row bytes x ring stages; default 128x2)
struct GVariant { bool glds; int rb, stages; };
static const GVariant& gvariant() {
    static const GVariant v = [] {
        GVariant g{true, 128, 2};
        if (const char* e = getenv("CB_BF16_GEMM")) {
            std::string s(e);
            int rb = 0, st = 0;
            if (sscanf(e, "%dx%d", &rb, &st) == 2 && (rb == 64 || rb == 128) && st >= 2 && st <= 3) g = {true, rb, st};
        }
        return g;
    }();
    return v;
}
static int bf16_bk() { return gvariant().rb / 2; }
static bool bf16_p8(const GemmArgs& a) { return bf16_bk() == 64 && gemm_bf16_p8_ok(a); }
template <typename T, int AK, int RB, int ST, bool FAST>
static void launch_glds_go(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka) {
    if (pl.BM == 128 && pl.BN == 128) hipLaunchKernelGGL((gemm_glds_kernel<T, 128, 128, AK, RB, ST, FAST>), grid, dim3(256), 0, s, ka);
    else if (pl.BM == 128 && pl.BN == 64) hipLaunchKernelGGL((gemm_glds_kernel<T, 128, 64, AK, RB, ST, FAST>), grid, dim3(256), 0, s, ka);
    else if (pl.BM == 64 && pl.BN == 128) hipLaunchKernelGGL((gemm_glds_kernel<T, 64, 128, AK, RB, ST, FAST>), grid, dim3(256), 0, s, ka);
    else hipLaunchKernelGGL((gemm_glds_kernel<T, 64, 64, AK, RB, ST, FAST>), grid, dim3(256), 0, s, ka);
}
static bool glds_fast_on() {
    static const bool on = [] { const char* e = getenv("CB_GLDS_FAST"); return !(e && e[0] == '0'); }();
    return on;
}
template <typename T, int AK, int RB, int ST>
static void launch_glds(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka) {
    if constexpr (ST == 2 && RB == 128 && AK != AK_CONV) {        // the default ring only: the experiment variants keep the per-lane path
        constexpr int BK = RB / (int)sizeof(T);
        if (glds_fast_on() && !ka.safe && ka.a.K % BK == 0) { launch_glds_go<T, AK, RB, ST, true>(pl, grid, s, ka); return; }
    }
    if (pl.BM == 128 && pl.BN == 128) hipLaunchKernelGGL((gemm_glds_kernel<T, 128, 128, AK, RB, ST>), grid, dim3(256), 0, s, ka);
    else if (pl.BM == 128 && pl.BN == 64) hipLaunchKernelGGL((gemm_glds_kernel<T, 128, 64, AK, RB, ST>), grid, dim3(256), 0, s, ka);
    else if (pl.BM == 64 && pl.BN == 128) hipLaunchKernelGGL((gemm_glds_kernel<T, 64, 128, AK, RB, ST>), grid, dim3(256), 0, s, ka);
    else hipLaunchKernelGGL((gemm_glds_kernel<T, 64, 64, AK, RB, ST>), grid, dim3(256), 0, s, ka);
}
template <int AK>
static void launch_tiles_bf16(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka) {
    const GVariant& v = gvariant();
    if (is_p8_tile(pl.BM, pl.BN)) { gemm_bf16_p8_launch(pl.BN, grid, s, ka); return; }      // the phase-pipelined tiles (cb_gemm_bf16_p8.hip)
    if (pl.BM == 256) {
        CB_CHECK(v.glds && v.rb == 128 && pl.BN == 128, CB_ERR_STATE, "gemm_bf16: the 256x128 tile exists for the default LDS-DMA variant only");
        set_dyn_lds(reinterpret_cast<const void*>(&gemm_glds256_kernel<bf16_t, AK>), GLDS256_LDS);
        hipLaunchKernelGGL((gemm_glds256_kernel<bf16_t, AK>), grid, dim3(512), GLDS256_LDS, s, ka);
        return;
    }
    if (v.rb == 128 && v.stages == 2) launch_glds<bf16_t, AK, 128, 2>(pl, grid, s, ka);
    else if (v.rb == 128 && v.stages == 3) launch_glds<bf16_t, AK, 128, 3>(pl, grid, s, ka);
    else if (v.rb == 64 && v.stages == 3) launch_glds<bf16_t, AK, 64, 3>(pl, grid, s, ka);
    else launch_glds<bf16_t, AK, 64, 2>(pl, grid, s, ka);
}

// ---------------------------------------------------------------- fp32 through the LDS-DMA kernel
// Eligible: both operands k-contiguous (ROW / CONV x ROW: activations or conv gathers against [out][in] weights), every row a whole
// number of 16-byte chunks (K, lda, ldb, conv C multiples of 4), operand slices below 2 GiB. Everything else — m-contiguous
// operands (Linear input gradients, attention P.V), the 3-channel image conv, misaligned op-level pointers — stays on the
// register-staged kernel of cb_gemm.hip. CB_F32_GLDS=0 switches it off (A/B measurements); CB_F32_GLDS=64x3 selects 64-byte rows.
static int f32_glds_mode() {
    static const int m = [] {
        const char* e = getenv("CB_F32_GLDS");
        if (!e) return 1;
        if (!strcmp(e, "0")) return 0;
        if (!strcmp(e, "64x3")) return 2;
        return 1;
    }();
    return m;
}
static int f32_glds_bk() { return f32_glds_mode() == 2 ? 16 : 32; }
bool gemm_f32_glds_shape_ok(const GemmArgs& a) {
    // CB_F32_GLDS=0 (A/B measurements) hands everything the register-staged kernel CAN run to it; the 2x2 / 4x4 kernels (the Upsample
    // layers' input-gradient conv) exist here only and stay
    const bool only_here = a.a_kind == LD_CONV && a.g.ks != 1 && a.g.ks != 3;
    if ((!f32_glds_mode() && !only_here) || a.dt != DT_F32 || a.b_kind != LD_ROW) return false;
    if (a.K % 4 || a.ldb % 4 || a.sBb % 4 || a.sBh % 4) return false;
    const long lim = ((1L << 31) - 64) / 4;
    if ((long)a.N * a.ldb >= lim) return false;
    if (a.a_kind == LD_CONV) {
        const ConvGeom& g = a.g;
        if (g.C % 4 || g.ks < 1 || g.ks > 4 || a.K != g.ks * g.ks * g.C) return false;
        // (+ the FAST path's descriptor bias: its voffsets are shifted up by the most negative row base; + its largest SCALAR offset, the
        // (tap, channel) walk up to (ks * W + ks) * C: the hardware adds soffset after the range check, the sum must stay a 31-bit offset)
        return (long)(a.M / (g.Ho * g.Wo)) * g.H * g.W * g.C + (long)(g.pad * g.W + g.pad) * g.C + (long)(g.ks * g.W + g.ks) * g.C < lim;
    }
    if (a.a_kind != LD_ROW || a.lda % 4 || a.sAb % 4 || a.sAh % 4) return false;
    return (long)a.M * a.lda + a.K < lim && (long)a.N * a.ldb + a.K < lim;       // + the FAST path's scalar k offset
}
static double f32_glds_speed() {
    static const double s = [] { const char* e = getenv("CB_PLAN_F32G_SPEED"); return e && atof(e) > 0 ? atof(e) : 1.25; }();
    return s;
}
// the 80-column tiles take what the fp32 LDS-DMA kernel takes, minus batches and the general gather (upsample / transposed holes / C % 32)
static bool n80_ok(const GemmArgs& a) {
    static const bool on = [] { const char* e = getenv("CB_N80"); return !(e && e[0] == '0'); }();
    if (!on || f32_glds_mode() != 1 || a.batch != 1 || a.N < 40) return false;
    if (a.a_kind == LD_CONV) return !a.g.up && !a.g.tr && a.g.C % 32 == 0;
    return a.a_kind == LD_ROW;
}
size_t gemm_f32_glds_ws_bytes(const GemmArgs& a) {
    Plan pl = make_plan(a, f32_glds_bk(), f32_glds_speed(), n80_ok(a));
    return pl.splitk <= 1 ? 0 : (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
}
// the 80-column tiles: dynamic LDS (36 / 56 / 96 KB for BM = 64 / 32 / 16), two ring stages. Three stages were measured twice and are not
// instantiated: round 4 on the 4096-row conv (two blocks per CU become one: 87 -> 106 us), round 6 as a per-launch policy — three stages
// exactly where the grid has at most 256 blocks, i.e. for the M <= 256 weight-streaming problems, whose CUs then hold two k-tiles in
// flight instead of one: 36.40 - 36.45 ms per fp32 step against 36.39 - 36.45 (three stages everywhere: 37.6; profiles/README.md). Those
// launches are not waiting for bytes. The kernel body keeps the general ring (STAGES is a template parameter, slots are compile-time).
template <int BM, int AK, bool FAST>
static void launch_n80_go(dim3 grid, hipStream_t s, const KArgs& ka) {
    constexpr int LDS = G80<BM, 2>::LDS;
    set_dyn_lds(reinterpret_cast<const void*>(&gemm_glds80_kernel<BM, AK, 2, FAST>), LDS);
    hipLaunchKernelGGL((gemm_glds80_kernel<BM, AK, 2, FAST>), grid, dim3(256), LDS, s, ka);
}
template <int BM, int AK>
static void launch_n80_one(dim3 grid, hipStream_t s, const KArgs& ka) {
    // FAST: every split's k range is a whole number of k-tiles per k-group (the last split included)
    static const bool allow = [] { const char* e = getenv("CB_N80_FAST"); return !(e && e[0] == '0'); }();
    const int unit = 32 * (64 / BM), last = ka.a.K - (ka.splitk - 1) * ka.k_per_split;
    if (allow && ka.k_per_split % unit == 0 && last % unit == 0 && !ka.safe) launch_n80_go<BM, AK, true>(grid, s, ka);
    else launch_n80_go<BM, AK, false>(grid, s, ka);
}
template <int AK>
static void launch_n80(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka) {
    CB_CHECK(AK != AK_CONV, CB_ERR_STATE, "gemm: the 80-column tile takes ROW and plain-conv (C %% 32 == 0) operands only");
    constexpr int AKK = AK == AK_CONV ? AK_CONVF : AK;
    if (pl.BM == 64) launch_n80_one<64, AKK>(grid, s, ka);
    else if (pl.BM == 32) launch_n80_one<32, AKK>(grid, s, ka);
    else launch_n80_one<16, AKK>(grid, s, ka);
}
template <int AK>
static void launch_tiles_f32(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka) {
    CB_CHECK(pl.BM != 256, CB_ERR_STATE, "gemm: the 256x128 tile is a bf16 variant");
    if (pl.BN == 80) { launch_n80<AK>(pl, grid, s, ka); return; }
    if (f32_glds_mode() == 2) launch_glds<float, AK, 64, 3>(pl, grid, s, ka);
    else launch_glds<float, AK, 128, 2>(pl, grid, s, ka);
}

// In-launch split-K reduction (cb_gemm_tiles.h, splitk_arrive): on unless CB_SPLITK_INLAUNCH=0 / cb_debug_splitk_inlaunch(0) (A/B and the
// bitwise comparison of tests/test_gpu_ops.py: partials + splitk_reduce_kernel).
static bool splitk_inlaunch_on() { return plan_force().splitk_inlaunch != 0; }
static int* splitk_counters(const Launch& L, const Plan& pl, int batch) {
    if (!splitk_inlaunch_on() || pl.splitk <= 1 || !L.cnt) return nullptr;
    if ((long)pl.tiles_m * pl.BM * (long)pl.tiles_n * pl.BN * 4 >= (1L << 31) - 64) return nullptr;     // slab offsets are 31-bit buffer offsets
    return L.cnt->take((size_t)pl.tiles_m * pl.tiles_n * batch);
}

// ---------------------------------------------------------------- offline autotuner (CB_AUTOTUNE=<file>)
// For every distinct problem that reaches the LDS-DMA kernel on a real context, time all (tile, split-K) candidates on the
// launch's own operands — outputs and partials go to scratch buffers, so the step itself is undisturbed and keeps running the
// analytic plan (the arena was sized for it) — and append "<plan_key> <BM> <BN> <split> <us> | <analytic BM BN split us>" to the file.
// tools/make_plan_table.py turns the file into cb_plan_table.inc. Batched (attention) problems are left to the model.
template <typename T> static void launch_problem(hipStream_t s, const GemmArgs& a, const Plan& pl, float* ws, int bk, int* cnt);
struct Tuner { FILE* f = nullptr; std::unordered_map<std::string, int> seen; };
static Tuner* tuner() {
    static Tuner* t = [] {
        const char* p = getenv("CB_AUTOTUNE");
        if (!p || !*p) return (Tuner*)nullptr;
        Tuner* t = new Tuner;
        t->f = fopen(p, "a");
        return t->f ? t : (Tuner*)nullptr;
    }();
    return t;
}
template <typename T>
static void autotune(const Launch& L, const GemmArgs& a, int bk, double speed) {
    Tuner* t = tuner();
    if (!t || L.dry || a.batch != 1) return;
    const std::string key = plan_key(a, bk);
    if (t->seen.count(key)) return;
    t->seen[key] = 1;
    static const int tiles[10][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {256, 128}, {64, 80}, {32, 80}, {16, 80}, {256, 256}, {512, 128}};
    static const int splits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    const bool n80 = sizeof(T) == 4 && n80_ok(a);
    const bool p8 = sizeof(T) == 2 && bf16_p8(a);
    const int ntile = p8 ? 10 : (n80 ? 8 : ((sizeof(T) == 2 && bk == 64) ? 5 : 4));           // the 8-wave tiles: bf16, default ring only
    const int ksteps = cdiv(a.K, bk);
    const size_t out_elems = (size_t)a.M * (size_t)std::max<long>(a.ldc, a.N);
    size_t ws_max = 0;
    for (int sp : splits) if (sp > 1 && ksteps / sp >= 2) ws_max = std::max(ws_max, (size_t)sp * a.M * a.N * sizeof(float));
    ws_max = std::min(ws_max, (size_t)1 << 30);
    void* ctmp = nullptr; void* wtmp = nullptr; void* cnttmp = nullptr;
    CB_HIP(hipDeviceSynchronize());                      // the other streams' work would disturb the timings
    CB_HIP(hipMalloc(&ctmp, out_elems * sizeof(T)));
    if (ws_max) CB_HIP(hipMalloc(&wtmp, ws_max));
    if (ws_max && splitk_inlaunch_on()) {       // candidates are timed the way they would run: in-launch reduction
        const size_t nc = (size_t)cdiv(a.M, 16) * cdiv(a.N, 64) + 64;
        CB_HIP(hipMalloc(&cnttmp, nc * sizeof(int))); CB_HIP(hipMemset(cnttmp, 0, nc * sizeof(int)));
    }
    GemmArgs b = a;
    b.C = ctmp; b.beta = 0.f;
    hipEvent_t e0, e1;
    CB_HIP(hipEventCreate(&e0)); CB_HIP(hipEventCreate(&e1));
    auto time_plan = [&](const Plan& pl) {
        for (int i = 0; i < 2; ++i) launch_problem<T>(L.stream, b, pl, static_cast<float*>(wtmp), bk, static_cast<int*>(cnttmp));
        CB_HIP(hipEventRecord(e0, L.stream));
        for (int i = 0; i < 5; ++i) launch_problem<T>(L.stream, b, pl, static_cast<float*>(wtmp), bk, static_cast<int*>(cnttmp));
        CB_HIP(hipEventRecord(e1, L.stream));
        CB_HIP(hipEventSynchronize(e1));
        float ms = 0.f; CB_HIP(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms * 1e3 / 5.0;
    };
    const Plan model = make_plan(a, bk, speed, n80, p8);
    const double t_model = (model.splitk <= 1 || (size_t)model.splitk * a.M * a.N * sizeof(float) <= ws_max) ? time_plan(model) : -1.0;
    double best = 1e30; Plan bp = model;
    for (int ti = 0; ti < ntile; ++ti) {
        const int BM = tiles[ti][0], BN = tiles[ti][1];
        if (BM == 256 && !(sizeof(T) == 2 && bk == 64)) continue;
        if (BN == 80 && !n80) continue;
        if (BN == 256 && (!p8 || a.N <= 128)) continue;
        if (BM == 512 && (!p8 || a.M <= 256)) continue;
        if ((BM >= 128 && a.M <= 64) || (BN == 128 && a.N <= 64) || (BM >= 256 && a.M <= 128)) continue;
        for (int sp : splits) {
            if (sp > 1 && ksteps / sp < 2) break;
            if (sp > 1 && (size_t)sp * a.M * a.N * sizeof(float) > ws_max) break;
            const long blocks = (long)cdiv(a.M, BM) * cdiv(a.N, BN) * sp;
            if (sp > 1 && blocks > 8192) break;
            const int kps = (int)cdiv(ksteps, sp) * bk;
            Plan pl{BM, BN, (int)cdiv(a.M, BM), (int)cdiv(a.N, BN), (int)cdiv(a.K, kps), kps};
            const double us = time_plan(pl);
            if (us < best) { best = us; bp = pl; }
        }
    }
    fprintf(t->f, "%s %d %d %d %.2f | %d %d %d %.2f\n", key.c_str(), bp.BM, bp.BN, bp.splitk, best, model.BM, model.BN, model.splitk, t_model);
    fflush(t->f);
    CB_HIP(hipEventDestroy(e0)); CB_HIP(hipEventDestroy(e1));
    CB_HIP(hipFree(ctmp)); if (wtmp) CB_HIP(hipFree(wtmp)); if (cnttmp) CB_HIP(hipFree(cnttmp));
}

template <> void launch_problem<bf16_t>(hipStream_t s, const GemmArgs& a, const Plan& pl, float* ws, int bk, int* cnt) {
    KArgs ka;
    ka.a = a; ka.tiles_m = pl.tiles_m; ka.tiles_n = pl.tiles_n; ka.splitk = pl.splitk; ka.k_per_split = pl.k_per_split;
    ka.ws = ws; ka.a_vec = 1; ka.b_vec = 1; ka.safe = plan_force().glds_safe; ka.cnt = pl.splitk > 1 ? cnt : nullptr;
    dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk, a.batch);
    const bool convf = a.a_kind == LD_CONV && !a.g.up && !a.g.tr && (a.g.C % bk == 0);
    if (a.a_kind == LD_ROW) launch_tiles_bf16<AK_ROW>(pl, grid, s, ka);
    else if (convf) launch_tiles_bf16<AK_CONVF>(pl, grid, s, ka);
    else launch_tiles_bf16<AK_CONV>(pl, grid, s, ka);
    if (pl.splitk > 1 && !ka.cnt) {
        long total = (long)a.M * a.N * a.batch;
        hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, ka);
    }
}
template <> void launch_problem<float>(hipStream_t s, const GemmArgs& a, const Plan& pl, float* ws, int bk, int* cnt) {
    KArgs ka;
    ka.a = a; ka.tiles_m = pl.tiles_m; ka.tiles_n = pl.tiles_n; ka.splitk = pl.splitk; ka.k_per_split = pl.k_per_split;
    ka.ws = ws; ka.a_vec = 1; ka.b_vec = 1; ka.safe = plan_force().glds_safe; ka.cnt = pl.splitk > 1 ? cnt : nullptr;
    dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk, a.batch);
    const bool convf = a.a_kind == LD_CONV && !a.g.up && !a.g.tr && (a.g.C % bk == 0);
    if (a.a_kind == LD_ROW) launch_tiles_f32<AK_ROW>(pl, grid, s, ka);
    else if (convf) launch_tiles_f32<AK_CONVF>(pl, grid, s, ka);
    else launch_tiles_f32<AK_CONV>(pl, grid, s, ka);
    if (pl.splitk > 1 && !ka.cnt) {
        long total = (long)a.M * a.N * a.batch;
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, ka);
    }
}

// CB_GEMM_LOG=<file>: one "<plan_key> <BM> <BN> <split>" line per problem handed to the LDS-DMA kernels, dry contexts included — the
// launch histogram of a step without a GPU (tools/gemm_histogram.py joins it with the autotune records' measured times).
static void gemm_log(const GemmArgs& a, const Plan& pl, int bk) {
    static FILE* f = [] { const char* p = getenv("CB_GEMM_LOG"); return (p && *p) ? fopen(p, "a") : (FILE*)nullptr; }();
    if (!f) return;
    fprintf(f, "%s %d %d %d\n", plan_key(a, bk).c_str(), pl.BM, pl.BN, pl.splitk);
    fflush(f);
}

// returns false (nothing launched, nothing counted) when the problem is not eligible
bool gemm_f32_glds(const Launch& L, const GemmArgs& a, float* ws, size_t ws_bytes) {
    if (!gemm_f32_glds_shape_ok(a)) return false;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!L.dry && !(aligned16(a.A) && aligned16(a.B))) return false;
    if (a.a_kind == LD_CONV) {
        CB_CHECK(a.M % (a.g.Ho * a.g.Wo) == 0, CB_ERR_SHAPE, "conv: M=%d not a multiple of Ho*Wo=%d", a.M, a.g.Ho * a.g.Wo);
        CB_CHECK(!(a.g.up && a.g.tr), CB_ERR_ARG, "conv: up and tr are exclusive");
    }
    const int bk = f32_glds_bk();
    Plan pl = make_plan(a, bk, f32_glds_speed(), n80_ok(a));
    if (L.stats) { L.stats->gemm_flops += 2.0 * a.M * a.N * (double)a.K * a.batch; L.stats->gemm_launches++; }
    int* const cnt = splitk_counters(L, pl, a.batch);
    L.count(pl.splitk > 1 && !cnt ? 2 : 1);
    gemm_log(a, pl, bk);
    if (pl.splitk > 1) {
        size_t need = (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
        CB_CHECK(L.dry || (ws && ws_bytes >= need), CB_ERR_OOM, "gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
    }
    if (L.dry) return true;
    autotune<float>(L, a, bk, f32_glds_speed());
    KArgs ka;
    ka.a = a;
    ka.tiles_m = pl.tiles_m; ka.tiles_n = pl.tiles_n;
    ka.splitk = pl.splitk; ka.k_per_split = pl.k_per_split;
    ka.ws = ws; ka.a_vec = 1; ka.b_vec = 1; ka.safe = plan_force().glds_safe; ka.cnt = cnt;
    dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk, a.batch);
    const bool convf = a.a_kind == LD_CONV && !a.g.up && !a.g.tr && (a.g.C % bk == 0);
    ProfRec rec; bool prof = L.prof && L.prof->on;
    if (prof) {
        rec.flops = 2.0 * a.M * a.N * (double)a.K * a.batch;
        double ea_ = a.a_kind == LD_CONV ? (double)(a.M / (a.g.Ho * a.g.Wo)) * a.g.H * a.g.W * a.g.C : (double)a.M * a.K * a.batch;
        double eb_ = (double)a.N * a.K * (a.sBb || a.sBh ? a.batch : 1);
        rec.bytes = 4.0 * (ea_ + eb_ + (double)a.M * a.N * a.batch);
        CB_HIP(hipEventCreate(&rec.e0)); CB_HIP(hipEventCreate(&rec.e1));
        CB_HIP(hipEventRecord(rec.e0, L.stream));
    }
    if (a.a_kind == LD_ROW) launch_tiles_f32<AK_ROW>(pl, grid, L.stream, ka);
    else if (convf) launch_tiles_f32<AK_CONVF>(pl, grid, L.stream, ka);
    else launch_tiles_f32<AK_CONV>(pl, grid, L.stream, ka);
    CB_HIP(hipGetLastError());
    if (prof) {
        char nm[96];
        snprintf(nm, sizeof nm, "gemm_glds%s_kernel<float,%d,%d,%s>", pl.BN == 80 ? "80" : "", pl.BM, pl.BN, a.a_kind == LD_ROW ? "ROW" : (convf ? "CONVF" : "CONV"));
        rec.variant = L.prof->id_of(nm);
        CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec);
    }
    if (pl.splitk > 1 && !cnt) {
        long total = (long)a.M * a.N * a.batch;
        unsigned nb = (unsigned)std::min<long>((total + 255) / 256, 4096);
        ProfRec r2;
        if (prof) {
            r2.variant = L.prof->id_of("splitk_reduce_kernel");
            r2.flops = 0;
            r2.bytes = 4.0 * total * (pl.splitk + 1 + (a.R ? 1 : 0) + (a.beta != 0.f ? 1 : 0));
            CB_HIP(hipEventCreate(&r2.e0)); CB_HIP(hipEventCreate(&r2.e1));
            CB_HIP(hipEventRecord(r2.e0, L.stream));
        }
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(nb), dim3(256), 0, L.stream, ka);
        CB_HIP(hipGetLastError());
        if (prof) { CB_HIP(hipEventRecord(r2.e1, L.stream)); L.prof->recs.push_back(r2); }
    }
    return true;
}

size_t gemm_bf16_ws_bytes(const GemmArgs& a) {
    Plan pl = make_plan(a, bf16_bk(), plan_knobs().bf16_speed, false, bf16_p8(a));
    if (pl.splitk <= 1) return 0;
    return (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
}

void gemm_bf16(const Launch& L, const GemmArgs& a, float* ws, size_t ws_bytes) {
    CB_CHECK(a.dt == DT_BF16, CB_ERR_ARG, "gemm_bf16: the bf16 kernel was handed an fp32 problem");
    CB_CHECK(a.M > 0 && a.N > 0 && a.K > 0, CB_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    CB_CHECK(a.batch >= 1 && a.nh >= 1 && a.batch % a.nh == 0, CB_ERR_SHAPE, "gemm: bad batch %d/%d", a.batch, a.nh);
    CB_CHECK(a.beta == 0.f || a.beta == 1.f, CB_ERR_ARG, "gemm: beta must be 0 or 1");
    CB_CHECK(a.b_kind == LD_ROW && (a.a_kind == LD_ROW || a.a_kind == LD_CONV), CB_ERR_ARG,
             "gemm_bf16: operands must be k-contiguous (ROW / CONV x ROW), got kinds %d/%d", a.a_kind, a.b_kind);
    CB_CHECK(a.K % 8 == 0, CB_ERR_SHAPE, "gemm_bf16: K=%d must be a multiple of 8 (16-byte chunks)", a.K);
    if (a.a_kind == LD_CONV) {
        const ConvGeom& g = a.g;
        CB_CHECK(g.ks >= 1 && g.ks <= 4, CB_ERR_SHAPE, "conv: ks=%d", g.ks);
        CB_CHECK(a.K == g.ks * g.ks * g.C, CB_ERR_SHAPE, "conv: K=%d != %d*%d*%d", a.K, g.ks, g.ks, g.C);
        CB_CHECK(g.C % 8 == 0, CB_ERR_SHAPE, "conv bf16: C=%d must be a multiple of 8 (pad the channel axis)", g.C);
        CB_CHECK(a.M % (g.Ho * g.Wo) == 0, CB_ERR_SHAPE, "conv: M=%d not a multiple of Ho*Wo=%d", a.M, g.Ho * g.Wo);
        CB_CHECK(!(g.up && g.tr), CB_ERR_ARG, "conv: up and tr are exclusive");
    } else {
        CB_CHECK(a.lda % 8 == 0 && a.sAb % 8 == 0 && a.sAh % 8 == 0, CB_ERR_SHAPE, "gemm_bf16: A strides must be multiples of 8 elements");
    }
    CB_CHECK(a.ldb % 8 == 0 && a.sBb % 8 == 0 && a.sBh % 8 == 0, CB_ERR_SHAPE, "gemm_bf16: B strides must be multiples of 8 elements");
    Plan pl = make_plan(a, bf16_bk(), plan_knobs().bf16_speed, false, bf16_p8(a));
    if (L.stats) { L.stats->gemm_flops += 2.0 * a.M * a.N * (double)a.K * a.batch; L.stats->gemm_launches++; }
    int* const cnt = splitk_counters(L, pl, a.batch);
    L.count(pl.splitk > 1 && !cnt ? 2 : 1);
    gemm_log(a, pl, bf16_bk());
    if (pl.splitk > 1) {
        size_t need = (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
        CB_CHECK(L.dry || (ws && ws_bytes >= need), CB_ERR_OOM, "gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
    }
    if (L.dry) return;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    CB_CHECK(aligned16(a.A) && aligned16(a.B), CB_ERR_ARG, "gemm_bf16: operands must be 16-byte aligned");
    {   // 32-bit byte offsets inside one operand slice (buffer loads)
        const long lim = (1L << 31) - 64;
        long ea = a.a_kind == LD_CONV ? (long)(a.M / (a.g.Ho * a.g.Wo)) * a.g.H * a.g.W * a.g.C + (long)(a.g.pad * a.g.W + a.g.pad) * a.g.C
                                            + (long)(a.g.ks * a.g.W + a.g.ks) * a.g.C
                                      : (long)a.M * a.lda + a.K;     // conv: + the FAST path's descriptor bias; both: + its largest scalar offset
        long eb = (long)a.N * a.ldb + a.K;
        CB_CHECK(ea * 2 < lim && eb * 2 < lim, CB_ERR_SHAPE, "gemm_bf16: an operand slice exceeds 2 GiB");
    }
    autotune<bf16_t>(L, a, bf16_bk(), plan_knobs().bf16_speed);
    KArgs ka;
    ka.a = a;
    ka.tiles_m = pl.tiles_m; ka.tiles_n = pl.tiles_n;
    ka.splitk = pl.splitk; ka.k_per_split = pl.k_per_split;
    ka.ws = ws; ka.a_vec = 1; ka.b_vec = 1; ka.safe = plan_force().glds_safe; ka.cnt = cnt;
    dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk, a.batch);
    const bool convf = a.a_kind == LD_CONV && !a.g.up && !a.g.tr && (a.g.C % bf16_bk() == 0);
    ProfRec rec; bool prof = L.prof && L.prof->on;
    if (prof) {
        rec.flops = 2.0 * a.M * a.N * (double)a.K * a.batch;
        double ea_ = a.a_kind == LD_CONV ? (double)(a.M / (a.g.Ho * a.g.Wo)) * a.g.H * a.g.W * a.g.C : (double)a.M * a.K * a.batch;
        double eb_ = (double)a.N * a.K * (a.sBb || a.sBh ? a.batch : 1);
        rec.bytes = 2.0 * (ea_ + eb_ + (double)a.M * a.N * a.batch);
        CB_HIP(hipEventCreate(&rec.e0)); CB_HIP(hipEventCreate(&rec.e1));
        CB_HIP(hipEventRecord(rec.e0, L.stream));
    }
    if (a.a_kind == LD_ROW) launch_tiles_bf16<AK_ROW>(pl, grid, L.stream, ka);
    else if (convf) launch_tiles_bf16<AK_CONVF>(pl, grid, L.stream, ka);
    else launch_tiles_bf16<AK_CONV>(pl, grid, L.stream, ka);
    CB_HIP(hipGetLastError());
    if (prof) {
        char nm[96];
        snprintf(nm, sizeof nm, "gemm_%skernel<bf16,%d,%d,%s>", is_p8_tile(pl.BM, pl.BN) ? "bf16_p8_" : (pl.BM == 256 ? "glds256_" : "glds_"), pl.BM, pl.BN,
                 a.a_kind == LD_ROW ? "ROW" : (convf ? "CONVF" : "CONV"));
        rec.variant = L.prof->id_of(nm);
        CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec);
    }
    if (pl.splitk > 1 && !cnt) {
        long total = (long)a.M * a.N * a.batch;
        unsigned nb = (unsigned)std::min<long>((total + 255) / 256, 4096);
        ProfRec r2;
        if (prof) {
            r2.variant = L.prof->id_of("splitk_reduce_kernel<bf16>");
            r2.flops = 0;
            r2.bytes = 4.0 * total * pl.splitk + 2.0 * total * (1 + (a.R ? 1 : 0) + (a.beta != 0.f ? 1 : 0));
            CB_HIP(hipEventCreate(&r2.e0)); CB_HIP(hipEventCreate(&r2.e1));
            CB_HIP(hipEventRecord(r2.e0, L.stream));
        }
        hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(nb), dim3(256), 0, L.stream, ka);
        CB_HIP(hipGetLastError());
        if (prof) { CB_HIP(hipEventRecord(r2.e1, L.stream)); L.prof->recs.push_back(r2); }
    }
}

}  // namespace cb
