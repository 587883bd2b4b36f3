// cb_gemm.hip — fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// One kernel family covers every dense contraction on the CelebBasis training step
// (SURVEY.md Appendix A.1, A.4, A.5, A.6): Linear fwd/dgrad, 1x1 and 3x3 conv fwd/dgrad
// (NHWC implicit GEMM, optional fused nearest-2x upsample or transposed stride-2 gather),
// and the batched attention contractions (QK^T, PV, and their backward forms).
//
//   C[M,N] = alpha * A[M,K] . B[K,N] + bias[n] + rowbias[img(m)][n] + R[m][n] + beta * C[m][n]
//
// Design (CDNA4): 256 threads = 4 wave64 in a 2x2 arrangement; each wave owns a (BM/2)x(BN/2)
// sub-tile made of 32x32 fp32 MFMA tiles (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF chip peak); BK = 16.
// LDS staging depends on how the operand lies in memory:
//  * k-contiguous (activations, conv gathers, [out][in] weights; the 16-byte-load kernels): [row][16 k], the four 16-byte
//    chunks of a row XOR-swizzled by (row >> 2) & 3. One conflict-free ds_write_b128 per global 16-byte load; a lane's
//    fragments for all 8 MFMA k-steps of the tile are two conflict-free ds_read_b128 at loop-invariant addresses. MFMA
//    step s contracts the k-pair (s, s + 8) — any pairing is a valid contraction as long as A and B agree.
//  * m-contiguous (transposed operands) and the scalar fallback loaders: k-major [BK][BM+4]; the fragment read
//    (lane l -> A[m = l & 31][k = s + 8 * (l >> 5)]) is a conflict-free ds_read_b32 per half-wave, no shuffle needed.
// Global->register prefetch of tile t+1 is issued before the MFMA block of tile t (one barrier per k-tile,
// double-buffered LDS). Split-K (partials in a workspace + a chip-wide reduce/epilogue kernel) fills the 256 CUs for the
// small-M / deep-K layers (8x8 and 16x16 UNet levels, CLIP, attention bwd). blockIdx -> tile mapping is XCD-aware
// (bijective remap) so that tiles sharing an A row-panel land on the same XCD L2. Measurements behind these choices:
// profiles/r01_pmc_conv.md (PMC) and profiles/README.md (per-step history, rejected experiments).
#include "cb_gemm_tiles.h"
#include <cstdarg>
#include <mutex>

namespace cb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

// ---- tile loaders: global -> registers -------------------------------------------------
// ROW: element (r, k) at p[r*ld + k]  (k contiguous)   -> NV = ROWS/64 float4 per thread (4 k's of one row)
// COL: element (r, k) at p[k*ld + r]  (r contiguous)   -> NV = ROWS/64 float4 per thread (4 r's of one k)
// CONV: row r = output pixel, k = (ky*ks + kx)*C + ci gathered from NHWC input.

template <int ROWS>
__device__ __forceinline__ void load_row(float4 (&reg)[ROWS / 64], const float* __restrict__ p, long ld, int r0, int R,
                                         int k0, int Kend, int vec, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        int idx = tid + i * 256;
        int r = r0 + (idx >> 2);
        int k = k0 + ((idx & 3) << 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) {
            const float* q = p + (long)r * ld + k;
            if (vec && k + 3 < Kend) {
                v = *reinterpret_cast<const float4*>(q);
            } else {
                if (k + 0 < Kend) v.x = q[0];
                if (k + 1 < Kend) v.y = q[1];
                if (k + 2 < Kend) v.z = q[2];
                if (k + 3 < Kend) v.w = q[3];
            }
        }
        reg[i] = v;
    }
}

template <int ROWS>
__device__ __forceinline__ void load_col(float4 (&reg)[ROWS / 64], const float* __restrict__ p, long ld, int r0, int R,
                                         int k0, int Kend, int vec, int tid) {
    constexpr int QPR = ROWS / 4;  // float4 per k-row
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        int idx = tid + i * 256;
        int k = k0 + idx / QPR;
        int r = r0 + (idx % QPR) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < Kend) {
            const float* q = p + (long)k * ld + r;
            if (vec && r + 3 < R) {
                v = *reinterpret_cast<const float4*>(q);
            } else {
                if (r + 0 < R) v.x = q[0];
                if (r + 1 < R) v.y = q[1];
                if (r + 2 < R) v.z = q[2];
                if (r + 3 < R) v.w = q[3];
            }
        }
        reg[i] = v;
    }
}

__device__ __forceinline__ float conv_fetch(const float* __restrict__ p, const ConvGeom& g, int n, int oy, int ox, int k) {
    int tap = k / g.C;
    int ci = k - tap * g.C;
    int ky = tap / g.ks, kx = tap - ky * g.ks;
    int vy = oy * g.stride + ky - g.pad;
    int vx = ox * g.stride + kx - g.pad;
    int sh = (g.up | g.tr);
    int Hv = g.H << sh, Wv = g.W << sh;
    if (vy < 0 || vx < 0 || vy >= Hv || vx >= Wv) return 0.f;
    if (g.tr && ((vy | vx) & 1)) return 0.f;
    int iy = vy >> sh, ix = vx >> sh;
    return p[(((long)n * g.H + iy) * g.W + ix) * g.C + ci];
}

template <int ROWS>
__device__ __forceinline__ void load_conv(float4 (&reg)[ROWS / 64], const float* __restrict__ p, const ConvGeom& g,
                                          const ConvRows<ROWS, 4>& cr, int k0, int Kend, int vec, int tid) {
    const int k = k0 + ((tid & 3) << 2);
    const int sh = (g.up | g.tr);
    const int Hv = g.H << sh, Wv = g.W << sh;
    if (vec && k + 3 < Kend) {      // C % 4 == 0: the 4 k's of this thread share one tap
        const int ky = (g.ks == 3) ? ((cr.tap * 11) >> 5) : 0;
        const int kx = cr.tap - ky * g.ks;
#pragma unroll
        for (int i = 0; i < ROWS / 64; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int vy = cr.vy0[i] + ky, vx = cr.vx0[i] + kx;
            bool ok = (cr.nb[i] >= 0) && (vy >= 0) && (vx >= 0) && (vy < Hv) && (vx < Wv);
            if (g.tr && ((vy | vx) & 1)) ok = false;
            if (ok) v = *reinterpret_cast<const float4*>(p + ((long)(cr.nb[i] + (vy >> sh) * g.W + (vx >> sh))) * g.C + cr.ci);
            reg[i] = v;
        }
    } else {                        // generic path (Cin = 3, K tails): per-element decode
#pragma unroll
        for (int i = 0; i < ROWS / 64; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cr.nb[i] >= 0 && k < Kend) {
                int n = cr.nb[i] / (g.H * g.W);
                int oy = (cr.vy0[i] + g.pad) / g.stride, ox = (cr.vx0[i] + g.pad) / g.stride;
                v.x = conv_fetch(p, g, n, oy, ox, k);
                if (k + 1 < Kend) v.y = conv_fetch(p, g, n, oy, ox, k + 1);
                if (k + 2 < Kend) v.z = conv_fetch(p, g, n, oy, ox, k + 2);
                if (k + 3 < Kend) v.w = conv_fetch(p, g, n, oy, ox, k + 3);
            }
            reg[i] = v;
        }
    }
}

// ---- branch-free 16-byte loaders: cb_gemm_tiles.h (shared with the bf16 kernel); the m-contiguous (COL) form is fp32 only
// COL operand: element (r, k) at byte (k*ld + r)*4; a thread's slots differ in k
template <int ROWS>
__device__ __forceinline__ void offs_col(int (&off)[ROWS / 64], long ld, int r0, int R, int k0, int Kend, int tid) {
    constexpr int QPR = ROWS / 4;
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        int idx = tid + i * 256;
        int k = k0 + idx / QPR;
        int r = r0 + (idx % QPR) * 4;
        off[i] = ((k < Kend) && (r < R)) ? (int)(((long)k * ld + r) * 4) : OOB;
    }
}
// ---- registers -> LDS (k-major tile [BK][ROWS+4]) -----------------------------------------
template <int ROWS>
__device__ __forceinline__ void store_kcontig(float (*T)[ROWS + 4], const float4 (&reg)[ROWS / 64], int tid, unsigned mask) {
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        int idx = tid + i * 256;
        int r = idx >> 2;
        int k = (idx & 3) << 2;
        const bool ok = (mask >> i) & 1u;
        T[k + 0][r] = ok ? reg[i].x : 0.f;
        T[k + 1][r] = ok ? reg[i].y : 0.f;
        T[k + 2][r] = ok ? reg[i].z : 0.f;
        T[k + 3][r] = ok ? reg[i].w : 0.f;
    }
}
template <int ROWS>
__device__ __forceinline__ void store_rcontig(float (*T)[ROWS + 4], const float4 (&reg)[ROWS / 64], int tid, unsigned mask) {
    constexpr int QPR = ROWS / 4;
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i) {
        int idx = tid + i * 256;
        int k = idx / QPR;
        int r = (idx % QPR) * 4;
        const bool ok = (mask >> i) & 1u;
        *reinterpret_cast<float4*>(&T[k][r]) = ok ? reg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int BM, int BN, int AK, int BKIND, bool VEC>
__device__ __forceinline__ void gemm_body(const KArgs& p, float (*As)[BK][BM + 4], float (*Bs)[BK][BN + 4]) {
    constexpr int TM = BM / 64, TN = BN / 64;

    const GemmArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int z = blockIdx.z;
    const int zb = z / a.nh, zh = z - zb * a.nh;

    const float* __restrict__ Ap = static_cast<const float*>(a.A) + zb * a.sAb + zh * a.sAh;
    const float* __restrict__ Bp = static_cast<const float*>(a.B) + zb * a.sBb + zh * a.sBh;

    const int kbeg = split * p.k_per_split;
    const int kend = min(a.K, kbeg + p.k_per_split);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[BM / 64], rb[BN / 64];
    // LD_CONVF is the host's choice for VEC && !up && !tr && C % BK == 0 convs: the tap / channel walk of the k loop is then
    // wave-uniform (SGPRs). It is a separate instantiation so that the general gather's per-row state does not cost the
    // common kernel registers.
    constexpr bool conv_fast = (AK == LD_CONVF);
    static_assert(!conv_fast || VEC, "LD_CONVF needs the vector loaders");
    ConvRows<BM, 4> cr;
    ConvFast<BM, 4> cf;
    int tap_u = 0, ci_u = 0;                                                                   // scalar decode of k0
    if constexpr (conv_fast) { cf.init(a.g, m0, a.M, tid); tap_u = kbeg / a.g.C; ci_u = kbeg - tap_u * a.g.C; }
    else if constexpr (AK == LD_CONV) cr.init(a.g, m0, a.M, kbeg, tid);

    const unsigned ma = ~0u, mb = ~0u;
    __amdgpu_buffer_rsrc_t srdA = make_srd(Ap), srdB = make_srd(Bp);
    RowOffs<BM, 4> roA; RowOffs<BN, 4> roB;
    if constexpr (VEC && AK == LD_ROW) roA.init(a.lda, m0, a.M, tid);
    if constexpr (VEC && BKIND == LD_ROW) roB.init(a.ldb, n0, a.N, tid);
    // VEC: all byte offsets are computed first, then the 16-byte buffer loads are issued back to back (pinned by a
    // sched_barrier): interleaving offset math with the loads made the register allocator reuse a load destination as an
    // address temporary, which costs an s_waitcnt vmcnt(0) in the middle of the prefetch.
    auto loadAB = [&](int k0) {
        if constexpr (VEC) {
            int oa[BM / 64], ob[BN / 64];
            if constexpr (AK == LD_ROW) offs_row<BM, 4>(oa, roA, k0, kend, tid);
            else if constexpr (AK == LD_COL) offs_col<BM>(oa, a.lda, m0, a.M, k0, kend, tid);
            else if constexpr (conv_fast) {
                const int ky = (a.g.ks == 3) ? ((tap_u * 11) >> 5) : 0;
                const int kx = tap_u - ky * a.g.ks;
                offs_conv_fast<BM, 4>(oa, cf, tap_u, ((ky * a.g.W + kx) * a.g.C + ci_u) * 4);
                ci_u += BK;
                if (ci_u >= a.g.C) { ci_u -= a.g.C; ++tap_u; }
            } else { offs_conv<BM, 4>(oa, a.g, cr, k0, kend, tid); cr.advance(a.g); }
            if constexpr (BKIND == LD_ROW) offs_row<BN, 4>(ob, roB, k0, kend, tid);
            else offs_col<BN>(ob, a.ldb, n0, a.N, k0, kend, tid);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BM / 64; ++i) ra[i] = buf_ld4(srdA, oa[i]);
#pragma unroll
            for (int i = 0; i < BN / 64; ++i) rb[i] = buf_ld4(srdB, ob[i]);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            if constexpr (AK == LD_ROW) load_row<BM>(ra, Ap, a.lda, m0, a.M, k0, kend, p.a_vec, tid);
            else if constexpr (AK == LD_COL) load_col<BM>(ra, Ap, a.lda, m0, a.M, k0, kend, p.a_vec, tid);
            else { load_conv<BM>(ra, Ap, a.g, cr, k0, kend, p.a_vec, tid); cr.advance(a.g); }
            if constexpr (BKIND == LD_ROW) load_row<BN>(rb, Bp, a.ldb, n0, a.N, k0, kend, p.b_vec, tid);
            else load_col<BN>(rb, Bp, a.ldb, n0, a.N, k0, kend, p.b_vec, tid);
        }
    };
    constexpr bool A_KC = VEC && (AK != LD_COL);          // k-contiguous operands use the swizzled [row][16] layout
    constexpr bool B_KC = VEC && (BKIND == LD_ROW);
    constexpr int A_BUF = BK * (BM + 4), B_BUF = BK * (BN + 4);
    float* const AsF = &As[0][0][0];
    float* const BsF = &Bs[0][0][0];
    auto storeA = [&](int buf) {
        if constexpr (A_KC) store_kc<BM>(AsF + buf * A_BUF, ra, tid);
        else if constexpr (AK == LD_COL) store_rcontig<BM>(As[buf], ra, tid, ma);
        else store_kcontig<BM>(As[buf], ra, tid, ma);
    };
    auto storeB = [&](int buf) {
        if constexpr (B_KC) store_kc<BN>(BsF + buf * B_BUF, rb, tid);
        else if constexpr (BKIND == LD_COL) store_rcontig<BN>(Bs[buf], rb, tid, mb);
        else store_kcontig<BN>(Bs[buf], rb, tid, mb);
    };

    if (kbeg < kend) {
        loadAB(kbeg);
        storeA(0); storeB(0);
    }
    __syncthreads();

    int cur = 0;
    const int kl = lane >> 5, ml = lane & 31;
    const int swz = (ml >> 2) & 3;                                         // == ((row >> 2) & 3): sub-tile bases are multiples of 32
    const int clo = ((2 * kl) ^ swz) * 4, chi = ((2 * kl + 1) ^ swz) * 4;  // float offsets of the lane's two 16-byte chunks
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) loadAB(k0 + BK);

        // MFMA step kk contracts the k-pair (kk, kk + BK/2) — any pairing of the tile's k indices is valid as long as A and B
        // agree. With the row pitch BM+4 (== 4 mod 8) the two half-waves then sit 8*pitch == 32 banks apart, so a fragment read
        // is conflict-free (the adjacent pairing (2kk, 2kk+1) cost one conflict cycle per dword: PMC SQ_LDS_BANK_CONFLICT).
        // all MFMA fragments of the tile are requested from LDS up front; the compiler then waits with counted lgkmcnt(N)
        // in front of each MFMA group instead of a full drain per k-step
        float av[BK / 2][TM], bv[BK / 2][TN];
        // k-contiguous operands: lane (ml, kl) owns k = 8*kl .. 8*kl+7 of its row = chunks 2*kl ("lo", MFMA steps 0-3) and
        // 2*kl+1 ("hi", steps 4-7), one ds_read_b128 each; the lo chunks of A and B are requested first
        const float* arow = AsF + cur * A_BUF + (wm * (BM / 2) + ml) * 16;
        const float* brow = BsF + cur * B_BUF + (wn * (BN / 2) + ml) * 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if constexpr (A_KC) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(arow + i * 512 + (half ? chi : clo));
                    av[4 * half + 0][i] = v.x; av[4 * half + 1][i] = v.y; av[4 * half + 2][i] = v.z; av[4 * half + 3][i] = v.w;
                }
            } else {
#pragma unroll
                for (int kk = 4 * half; kk < 4 * half + 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[kk][i] = As[cur][kk + (BK / 2) * kl][wm * (BM / 2) + i * 32 + ml];
            }
            if constexpr (B_KC) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float4 v = *reinterpret_cast<const float4*>(brow + j * 512 + (half ? chi : clo));
                    bv[4 * half + 0][j] = v.x; bv[4 * half + 1][j] = v.y; bv[4 * half + 2][j] = v.z; bv[4 * half + 3][j] = v.w;
                }
            } else {
#pragma unroll
                for (int kk = 4 * half; kk < 4 * half + 4; ++kk)
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[kk][j] = Bs[cur][kk + (BK / 2) * kl][wn * (BN / 2) + j * 32 + ml];
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the reads above: the scheduler otherwise sinks each pair next to its MFMAs
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][i], bv[kk][j], acc[i][j], 0, 0, 0);

        if (more) { storeA(cur ^ 1); storeB(cur ^ 1); }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue. C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (p.splitk > 1) {
        float* W = p.ws + ((long)split * a.batch + z) * (long)a.M * a.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int col = n0 + wn * (BN / 2) + j * 32 + ml;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                    if (row < a.M && col < a.N) W[(long)row * a.N + col] = acc[i][j][r];
                }
            }
        return;
    }
    float* __restrict__ Cp = static_cast<float*>(a.C) + zb * a.sCb + zh * a.sCh;
    const float* __restrict__ Rp = a.R ? static_cast<const float*>(a.R) + zb * a.sRb + zh * a.sRh : nullptr;
    const float* __restrict__ rowbias = static_cast<const float*>(a.rowbias);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int col = n0 + wn * (BN / 2) + j * 32 + ml;
            if (col >= a.N) continue;
            float bcol = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                if (row < a.M) {
                    float v = a.alpha * acc[i][j][r] + bcol;
                    if (rowbias) v += rowbias[(long)(row / a.rows_per_img) * (a.ld_rowbias ? a.ld_rowbias : a.N) + col];
                    if (Rp) v += Rp[(long)row * a.ldr + col];
                    float* c = Cp + (long)row * a.ldc + col;
                    if (a.beta != 0.f) v += a.beta * (*c);
                    *c = v;
                }
            }
        }
}

template <int BM, int BN, int AK, int BKIND, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(const KArgs p) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + 4];
    gemm_body<BM, BN, AK, BKIND, VEC>(p, As, Bs);
}
// Same body, register budget capped at 128 VGPRs (accumulators included) so that 4 blocks instead of 3 are resident per CU:
// an experiment switch for the 128x128 tile (CB_GEMM_OCC4=1), which otherwise needs 152 registers.
template <int AK, int BKIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_kernel_occ4(const KArgs p) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][128 + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][128 + 4];
    gemm_body<128, 128, AK, BKIND, true>(p, As, Bs);
}

// ---------------------------------------------------------------- host side (planner: cb_gemm_tiles.h)
void set_dyn_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, uint64_t> done;       // kernel -> bit per device ordinal
    int dev = 0;
    CB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    uint64_t& m = done[kernel];
    if (dev < 64 && ((m >> dev) & 1ull)) return;
    CB_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev < 64) m |= 1ull << dev;
}

size_t gemm_ws_bytes(const GemmArgs& a) {
    Plan pl = make_plan(a, BK, 1.0);
    size_t own = pl.splitk <= 1 ? 0 : (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
    // a shape-eligible problem normally runs on the LDS-DMA kernel (its own plan); a misaligned op-level pointer sends it back here
    return gemm_f32_glds_shape_ok(a) ? std::max(own, gemm_f32_glds_ws_bytes(a)) : own;
}

template <int AK, int BKIND, bool VEC>
static bool launch_tiles2(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka) {
    // 128x128 tile: 152 registers -> 3 blocks per CU (768 slots); the register-capped twin holds 4 (1024 slots). Measured on
    // MI355X (profiles/README.md, run 21): the twin wins when the launch then needs fewer rounds of blocks (1024 blocks +7 %,
    // 2048 blocks +12 %) and loses ~1 % otherwise; the variants that spill under the cap (general conv gather, ROW x COL) never use it.
    // CB_GEMM_OCC4=0/1 forces the choice (experiments).
    static const int occ4_env = [] { const char* e = getenv("CB_GEMM_OCC4"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    if (pl.BM == 128 && pl.BN == 128) {
        if constexpr (VEC && (AK == LD_CONVF || (AK == LD_ROW && BKIND == LD_ROW))) {
            const long blocks = (long)grid.x * grid.y * grid.z;
            const long r4 = (blocks + 1023) / 1024 * 4, r3 = (blocks + 767) / 768 * 3;
            const bool use4 = occ4_env >= 0 ? occ4_env == 1 : (blocks >= 1024 && r4 < r3);
            if (use4) { hipLaunchKernelGGL((gemm_kernel_occ4<AK, BKIND>), grid, dim3(256), 0, s, ka); return true; }
        }
        hipLaunchKernelGGL((gemm_kernel<128, 128, AK, BKIND, VEC>), grid, dim3(256), 0, s, ka);
    }
    else if (pl.BM == 128 && pl.BN == 64) hipLaunchKernelGGL((gemm_kernel<128, 64, AK, BKIND, VEC>), grid, dim3(256), 0, s, ka);
    else if (pl.BM == 64 && pl.BN == 128) hipLaunchKernelGGL((gemm_kernel<64, 128, AK, BKIND, VEC>), grid, dim3(256), 0, s, ka);
    else hipLaunchKernelGGL((gemm_kernel<64, 64, AK, BKIND, VEC>), grid, dim3(256), 0, s, ka);
    return false;
}
template <int AK, int BKIND>
static bool launch_tiles(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka, bool vec) {
    return vec ? launch_tiles2<AK, BKIND, true>(pl, grid, s, ka) : launch_tiles2<AK, BKIND, false>(pl, grid, s, ka);
}

void gemm(const Launch& L, const GemmArgs& a, float* ws, size_t ws_bytes) {
    CB_CHECK(a.M > 0 && a.N > 0 && a.K > 0, CB_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    CB_CHECK(a.batch >= 1 && a.nh >= 1 && a.batch % a.nh == 0, CB_ERR_SHAPE, "gemm: bad batch %d/%d", a.batch, a.nh);
    CB_CHECK(a.beta == 0.f || a.beta == 1.f, CB_ERR_ARG, "gemm: beta must be 0 or 1");
    if (a.a_kind == LD_CONV) {
        const ConvGeom& g = a.g;
        CB_CHECK(g.ks >= 1 && g.ks <= 4, CB_ERR_SHAPE, "conv: ks=%d", g.ks);
        CB_CHECK(a.K == g.ks * g.ks * g.C, CB_ERR_SHAPE, "conv: K=%d != %d*%d*%d", a.K, g.ks, g.ks, g.C);
        CB_CHECK(a.M % (g.Ho * g.Wo) == 0, CB_ERR_SHAPE, "conv: M=%d not a multiple of Ho*Wo=%d", a.M, g.Ho * g.Wo);
        CB_CHECK(!(g.up && g.tr), CB_ERR_ARG, "conv: up and tr are exclusive");
        CB_CHECK(a.b_kind == LD_ROW, CB_ERR_ARG, "conv: packed weights must be ROW");
    }
    CB_CHECK(a.dt == DT_F32, CB_ERR_ARG, "gemm: the fp32 kernel was handed a bf16 problem");
    if (gemm_f32_glds(L, a, ws, ws_bytes)) return;          // k-contiguous operands: the LDS-DMA kernel (cb_gemm_glds.hip)
    if (a.prelu || a.C2) {
        // the PReLU / second-output epilogue exists on the LDS-DMA kernels only: a problem they decline (CB_F32_GLDS=0, C % 4 != 0,
        // misaligned operands, slices over 2 GiB) runs de-fused — the plain GEMM, then the per-channel passes it replaced
        CB_CHECK(a.batch == 1 && a.ldc == a.N, CB_ERR_ARG, "gemm: de-fused PReLU / second output needs one dense [M][N] output");
        GemmArgs b = a;
        b.prelu = nullptr; b.C2 = nullptr; b.c2_scale = nullptr; b.c2_shift = nullptr;
        gemm(L, b, ws, ws_bytes);
        if (a.prelu) k_channel_affine(L, static_cast<float*>(a.C), static_cast<const float*>(a.C), nullptr, nullptr, a.prelu, a.M, a.N);
        if (a.C2) k_channel_affine(L, static_cast<float*>(a.C2), static_cast<const float*>(a.C), a.c2_scale, a.c2_shift, nullptr, a.M, a.N);
        return;
    }
    if (a.a_kind == LD_CONV)
        CB_CHECK(a.g.ks == 1 || a.g.ks == 3, CB_ERR_SHAPE, "conv: ks=%d exists on the LDS-DMA kernel only (16-byte aligned operands, C %% 4 == 0)", a.g.ks);
    Plan pl = make_plan(a, BK, 1.0);
    if (L.stats) { L.stats->gemm_flops += 2.0 * a.M * a.N * (double)a.K * a.batch; L.stats->gemm_launches++; }
    L.count(pl.splitk > 1 ? 2 : 1);
    if (pl.splitk > 1) {
        size_t need = (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
        CB_CHECK(L.dry || (ws && ws_bytes >= need), CB_ERR_OOM, "gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
    }
    if (L.dry) return;

    KArgs ka;
    ka.a = a;
    ka.tiles_m = pl.tiles_m; ka.tiles_n = pl.tiles_n;
    ka.splitk = pl.splitk; ka.k_per_split = pl.k_per_split;
    ka.ws = ws;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    auto strides_ok = [](long a, long b) { return (a % 4 == 0) && (b % 4 == 0); };
    if (a.a_kind == LD_CONV) ka.a_vec = aligned16(a.A) && (a.g.C % 4 == 0);
    else ka.a_vec = aligned16(a.A) && (a.lda % 4 == 0) && strides_ok(a.sAb, a.sAh);
    ka.b_vec = aligned16(a.B) && (a.ldb % 4 == 0) && strides_ok(a.sBb, a.sBh);

    // branch-free 16-byte kernels need every float4 to be fully in or fully out of range
    bool vec = ka.a_vec && ka.b_vec;
    {   // 32-bit byte offsets inside one operand slice (buffer loads)
        const long lim = (1L << 31) - 64;
        long ea = a.a_kind == LD_CONV ? (long)(a.M / (a.g.Ho * a.g.Wo)) * a.g.H * a.g.W * a.g.C
                                      : (a.a_kind == LD_ROW ? (long)a.M * a.lda : (long)a.K * a.lda);
        long eb = a.b_kind == LD_ROW ? (long)a.N * a.ldb : (long)a.K * a.ldb;
        vec = vec && ea * 4 < lim && eb * 4 < lim;
    }
    if (a.a_kind == LD_COL) vec = vec && (a.M % 4 == 0); else vec = vec && (a.K % 4 == 0);
    if (a.b_kind == LD_COL) vec = vec && (a.N % 4 == 0); else vec = vec && (a.K % 4 == 0);
    dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk, a.batch);
    ProfRec rec; bool prof = L.prof && L.prof->on;
    const bool convf = a.a_kind == LD_CONV && vec && !a.g.up && !a.g.tr && (a.g.C % BK == 0);
    if (prof) {
        rec.flops = 2.0 * a.M * a.N * (double)a.K * a.batch;
        // algorithmic HBM bytes: each operand once + the output once (conv A = the input image, not its im2col expansion)
        double ea_ = a.a_kind == LD_CONV ? (double)(a.M / (a.g.Ho * a.g.Wo)) * a.g.H * a.g.W * a.g.C : (double)a.M * a.K * a.batch;
        double eb_ = (double)a.N * a.K * (a.sBb || a.sBh ? a.batch : 1);
        rec.bytes = 4.0 * (ea_ + eb_ + (double)a.M * a.N * a.batch);
        CB_HIP(hipEventCreate(&rec.e0)); CB_HIP(hipEventCreate(&rec.e1));
        CB_HIP(hipEventRecord(rec.e0, L.stream));
    }
    bool occ4 = false;
    if (a.a_kind == LD_ROW && a.b_kind == LD_ROW) occ4 = launch_tiles<LD_ROW, LD_ROW>(pl, grid, L.stream, ka, vec);
    else if (a.a_kind == LD_ROW && a.b_kind == LD_COL) launch_tiles<LD_ROW, LD_COL>(pl, grid, L.stream, ka, vec);
    else if (a.a_kind == LD_COL && a.b_kind == LD_COL) launch_tiles<LD_COL, LD_COL>(pl, grid, L.stream, ka, vec);
    else if (a.a_kind == LD_COL && a.b_kind == LD_ROW) launch_tiles<LD_COL, LD_ROW>(pl, grid, L.stream, ka, vec);
    else if (a.a_kind == LD_CONV) {
        if (convf) occ4 = launch_tiles2<LD_CONVF, LD_ROW, true>(pl, grid, L.stream, ka);
        else launch_tiles<LD_CONV, LD_ROW>(pl, grid, L.stream, ka, vec);
    }
    else fail(CB_ERR_ARG, "gemm: unsupported operand kinds %d/%d", a.a_kind, a.b_kind);
    CB_HIP(hipGetLastError());
    // the event pair brackets the GEMM kernel alone and the record is named after the symbol that ran, so a variant's average
    // matches the rocprofv3 per-symbol average; the split-K reduce is its own (HBM-bound) record
    if (prof) {
        static const char* kn[4] = {"ROW", "COL", "CONV", "CONVF"};
        char nm[96];
        const char* ak = kn[convf ? 3 : a.a_kind];
        if (occ4) snprintf(nm, sizeof nm, "gemm_kernel_occ4<%s,%s>", ak, kn[a.b_kind]);
        else snprintf(nm, sizeof nm, "gemm_kernel<%d,%d,%s,%s%s>", pl.BM, pl.BN, ak, kn[a.b_kind], vec ? "" : ",generic");
        rec.variant = L.prof->id_of(nm);
        CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec);
    }
    // The partials are combined by a separate, chip-wide kernel. A fused "last block to arrive sums its tile" variant was
    // built and measured (profiles/README.md, runs 23/24): correct, but 46 -> 78-87 ms per step — agent-scope visibility needs
    // either whole-L2 write-back/invalidate fences or L2-bypassing sc1 accesses, and the fix-up serialises on one CU per tile.
    if (pl.splitk > 1) {
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
}

}  // namespace cb
