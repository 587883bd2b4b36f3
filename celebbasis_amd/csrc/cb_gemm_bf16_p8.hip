// cb_gemm_bf16_p8.hip — the phase-pipelined 256 x 256 bf16 GEMM / implicit-GEMM convolution tile for gfx950 (BASELINE config 3).
//
//   C[M,N] = alpha * A[M,K] . B[N,K]^T + bias[n] + rowbias[img(m)][n] + R[m][n] + beta * C[m][n]        (bf16 in HBM, fp32 accumulate)
//
// Why a second bf16 kernel. gemm_glds_kernel<bf16,128,128> (cb_gemm_glds.hip) is the "128 x 128 tile, two barriers per k-step"
// structure: all four waves stage, wait, barrier, read, compute in lockstep, so the LDS-DMA round trip and the ds_read latency are paid
// in the open once per k-tile — cdna_hip_programming.md §5 prices that structure at 839 - 890 TF on this chip (0.36 of the 2.5 PF dense
// peak) and it measures 675 - 815 TF on the config-3 convolutions. What breaks the ceiling (same section, "The 256^2 8-phase template"):
//   * a 256 x 256 x 64 tile — 128 FLOP per byte filled into LDS instead of 64 —, 8 waves as 2 (M) x 4 (N), each owning 128 x 64
//     outputs = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs);
//   * the k-tile cut into four PHASES of 8 MFMAs, every phase = {ds_read the phase's fragments, issue ONE half-tile of LDS-DMA,
//     s_barrier, MFMAs, s_barrier}; the two wave groups (wr = 0 / 1: one wave of each on every SIMD) run ONE BARRIER APART, so on every
//     SIMD one wave is in its matrix segment while its partner reads LDS / issues DMA (MI355X_MICROARCH.md §Two waves per SIMD);
//   * LDS-DMA pieces stay in flight ACROSS the barriers: the only VMEM wait of the loop is one counted `s_waitcnt vmcnt(4)` per
//     k-tile, never 0;
//   * `s_setprio 1` around the MFMA clusters (T5: has something to arbitrate only in such a role-split schedule).
//
// LDS: two k-tile buffers x (A 256 rows + B 256 rows) x 128 bytes = 128 KiB, one block per CU. The image of a half-tile (128 rows of
// one operand, 16 KiB) is the lane-linear LDS-DMA image of cb_gemm_glds.hip: slot idx = tid + i * 512 owns physical chunk idx & 7 of
// row idx >> 3 and FETCHES logical chunk (idx & 7) ^ ((row >> 1) & 7) (source-side XOR swizzle: the 16-lane groups of the fragment
// ds_read_b128 then hit 16 distinct bank quads).
//
// Schedule (k-tile t lives in buffer t & 1; HA0 / HA1 = A rows 0-127 / 128-255, HB0 / HB1 likewise; a wave (wr, wc) reads HA[wr] and
// HB[wc >> 1]; its fragments: a0 / a1 = its rows 0-63 / 64-127, b0 / b1 = its columns 0-31 / 32-63, four 16-k steps each):
//      phase   ds_read            LDS-DMA issue      MFMA (8)         wait
//      P0      a0, b0, b1 (16)    HA0 of tile t+1    a0 x b0
//      P1      —                  HA1 of tile t+1    a0 x b1
//      P2      a1 (8)             HB0 of tile t+2    a1 x b1
//      P3      —                  HB1 of tile t+2    a1 x b0          vmcnt(4) before the phase's first barrier
// Hazards, with the groups one barrier apart (interval I_j between barriers b_{j-1}, b_j; phase k: group 0 reads / issues in I_2k and
// computes in I_2k+1, group 1 one interval later):
//   RAW  a half-tile is read one phase AFTER the wait that retires it: tile t+1 is complete when every wave has passed P3's vmcnt(4)
//        (outstanding then: the two B halves of tile t+2 = 4 pieces per wave) and the barrier behind it; its first read is P0 of t+1.
//   WAR  a slot is re-staged at least TWO phases after its last read (the later group's reads are issued one interval late and
//        retire only at its lgkmcnt wait behind the next barrier): HB slots are last read in P0 and re-staged in P2 / P3 of the same
//        tile, HA slots are last read in P2 and re-staged in P0 / P1 of the next tile.
// Past the end of K every piece is still issued — with out-of-range offsets, so the DMA writes zeros into a slot nobody reads — and
// the number of outstanding pieces stays what the counted wait assumes. The build fails if this file has scratch (celebbasis_amd/build.py:
// a spill's scratch loads would be counted too).
//
// Second geometry, 512 x 128 (WM = 4: waves 4 (M) x 2 (N), the same 128 x 64 per wave): for the VAE's N = 128 layers at 512 x 512 — the
// largest FLOP share of config 3 —, where a 256-column tile would idle half its waves. A k-tile is four A units + one B unit (80 KiB; two
// buffers = all 160 KiB of LDS). b1 is read in P1, so the B slot is last read in P1 and re-staged in P3:
//      phase   ds_read            LDS-DMA issue               MFMA (8)         wait
//      P0      a0, b0 (12)        HA0, HA1 of tile t+1        a0 x b0
//      P1      b1 (4)             HA2, HA3 of tile t+1        a0 x b1
//      P2      a1 (8)             —                           a1 x b1
//      P3      —                  HB of tile t+2              a1 x b0          vmcnt(2)
// (same RAW / WAR rules: HA slots last read in P2 of tile t-1, re-staged from P0 of t; HB last read in P1 of t, re-staged in P3 of t.)
//
// Eligibility (host: gemm_bf16_p8_ok): bf16, both operands k-contiguous, K (and every split's k range) a multiple of 64, conv gathers
// of the plain kind (no upsample / transposed holes) with C % 64 == 0 — every UNet / VAE layer of SD v1-4 except the 3- / 4-channel
// ends. Everything else stays on gemm_glds_kernel. The planner never picks this tile by model: a measured plan-table entry or
// cb_debug_force_plan(256, 256, split) does.
#include "cb_gemm_tiles.h"

namespace cb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
enum { P8_ROW = 0, P8_CONVF = 2 };

namespace p8 {
constexpr int BK = 64, RB = 128;                                      // k-tile: 64 bf16 = one 128-byte row per tile row
constexpr int HALF = 128 * RB;                                        // a unit: 128 rows of one operand = 16 KiB = 1024 chunks
constexpr int NTHR = 512;
template <int WM> struct Geo {                                        // WM waves along M x 8 / WM along N, 128 x 64 outputs per wave
    static constexpr int WN = 8 / WM, BM = 128 * WM, BN = 64 * WN, NA = WM, NB = BN / 128;
    static constexpr int A_TILE = BM * RB, B_TILE = BN * RB, STAGE = A_TILE + B_TILE, LDS = 2 * STAGE;
    static constexpr int WAIT = 2 * NB;                              // pieces per wave still in flight behind P3's wait: the B units of tile t+2
};
}  // namespace p8

#define P8_FENCE() asm volatile("" ::: "memory")
#define P8_BARRIER() do { P8_FENCE(); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); P8_FENCE(); } while (0)

// ABL (diagnostics, CB_P8_ABL, WRONG results: timing only): 1 = no LDS-DMA in the k loop, 2 = no fragment reads in the k loop,
// 3 = both, 4 = both and no barriers (MFMAs only). profiles/r06_p8_ablation.json has the measurements.
template <int AK, int WM, int ABL>
__device__ __forceinline__ void gemm_bf16_p8_body(const KArgs& p, char* lds) {
    using namespace p8;
    using G = Geo<WM>;
    constexpr int BM = G::BM, BN = G::BN, NA = G::NA, NB = G::NB, A_TILE = G::A_TILE, STAGE = G::STAGE;
    const GemmArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = WM == 2 ? (wave >> 2) : (wave >> 1), wc = WM == 2 ? (wave & 3) : (wave & 1);
    const int grp = wave >> 2;                                        // waves 0-3 / 4-7: one of each on every SIMD
    const int ntiles = p.tiles_m * p.tiles_n;
    const int t_ = xcd_remap(blockIdx.x, ntiles);
    const int tm = t_ / p.tiles_n, tn = t_ - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y, z = blockIdx.z;
    const int zb = z / a.nh, zh = z - zb * a.nh;
    const bf16_t* __restrict__ Ap = static_cast<const bf16_t*>(a.A) + zb * a.sAb + zh * a.sAh;
    const bf16_t* __restrict__ Bp = static_cast<const bf16_t*>(a.B) + zb * a.sBb + zh * a.sBh;
    const int kbeg = split * p.k_per_split;
    const int kend = min(a.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg) / BK;                                 // host: a whole number of k-tiles

    // ---- per-slot source offsets: slot (unit u, i) = chunk idx = tid + i * 512 of the unit's image
    int a_vo[NA][2], b_vo[NB][2];                                     // current voffset of the slot (OOB = zeros)
    int a_base[NA][2]; unsigned a_vmask[NA][2];                       // conv: tap-independent row base, validity of the ks x ks taps
    int tap_u = 0, ci_u = 0, delta_u = 0;                             // conv: (tap, channel) of the next A tile to stage, and its byte offset
    const int abias = (AK == P8_CONVF) ? (a.g.pad * a.g.W + a.g.pad) * a.g.C * 2 : 0;     // most negative row base -> into the descriptor
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NTHR, row = idx >> 3;
        const int lc = ((idx & 7) ^ ((row >> 1) & 7)) * 8;             // logical chunk's first k element
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int rn = n0 + h * 128 + row;
            b_vo[h][i] = (rn < a.N) ? (int)(((long)rn * a.ldb + lc) * 2) : OOB;
        }
#pragma unroll
        for (int h = 0; h < NA; ++h) {
            const int r = m0 + h * 128 + row;
            if constexpr (AK == P8_ROW) {
                a_vo[h][i] = (r < a.M) ? (int)(((long)r * a.lda + lc) * 2) : OOB;
                a_base[h][i] = 0; a_vmask[h][i] = 0;
            } else {
                const ConvGeom& g = a.g;
                unsigned m = 0; int b = 0;
                if (r < a.M) {
                    const int HWo = g.Ho * g.Wo;
                    const int n = r / HWo, rem = r - n * HWo, oy = rem / g.Wo, ox = rem - oy * g.Wo;
                    const int vy0 = oy * g.stride - g.pad, vx0 = ox * g.stride - g.pad;
                    unsigned xb = 0, yb = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < g.ks) {
                            xb |= ((unsigned)(vx0 + k) < (unsigned)g.W) ? (1u << k) : 0u;
                            yb |= ((unsigned)(vy0 + k) < (unsigned)g.H) ? (1u << k) : 0u;
                        }
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky)
                        if (ky < g.ks && ((yb >> ky) & 1u)) m |= xb << (g.ks * ky);
                    b = (((n * g.H + vy0) * g.W + vx0) * g.C + lc) * 2 + abias;
                }
                a_vmask[h][i] = m; a_base[h][i] = b;
            }
        }
    }
    if constexpr (AK == P8_CONVF) {
        tap_u = kbeg / a.g.C; ci_u = kbeg - tap_u * a.g.C;
        const int ky = conv_tap_row(tap_u, a.g.ks), kx = tap_u - ky * a.g.ks;
        delta_u = ((ky * a.g.W + kx) * a.g.C + ci_u) * 2;
#pragma unroll
        for (int h = 0; h < NA; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) a_vo[h][i] = ((a_vmask[h][i] >> (tap_u & 31)) & 1u) ? a_base[h][i] : OOB;
    }
    const __amdgpu_buffer_rsrc_t srdA = make_srd(reinterpret_cast<const char*>(Ap) - abias), srdB = make_srd(Bp);
    const int wbase = wave * 1024;                                     // this wave's 64 chunks of a 512-chunk slot row

    // one unit of A / B of k-tile `tile` (relative to kbeg) -> its slot in buffer tile & 1. Two pieces per wave, always.
    auto stageA = [&](int tile, int h) {
        char* d = lds + (tile & 1) * STAGE + h * HALF + wbase;
        const bool live = tile < nk;
        const int so = (AK == P8_ROW) ? (kbeg + tile * BK) * 2 : delta_u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(d + i * (NTHR * 16)), 16, live ? a_vo[h][i] : OOB, so, 0, 0);
    };
    auto advanceA = [&]() {                                            // conv: the (tap, channel) walk, one k-tile per call
        if constexpr (AK == P8_CONVF) {
            ci_u += BK; delta_u += BK * 2;
            if (ci_u >= a.g.C) {
                ci_u -= a.g.C; ++tap_u;
                const int ky = conv_tap_row(tap_u, a.g.ks), kx = tap_u - ky * a.g.ks;
                delta_u = ((ky * a.g.W + kx) * a.g.C + ci_u) * 2;
#pragma unroll
                for (int h = 0; h < NA; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i) a_vo[h][i] = ((a_vmask[h][i] >> (tap_u & 31)) & 1u) ? a_base[h][i] : OOB;
            }
        }
    };
    auto stageB = [&](int tile, int h) {
        char* d = lds + (tile & 1) * STAGE + A_TILE + h * HALF + wbase;
        const bool live = tile < nk;
        const int so = (kbeg + tile * BK) * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(d + i * (NTHR * 16)), 16, live ? b_vo[h][i] : OOB, so, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane (kl, ml) reads row ml of a 32-row block at logical chunk 2 s + kl of 16-k step s
    const int kl = lane >> 5, ml = lane & 31, sw = (ml >> 1) & 7;
    const int aoff = (wr * 128 + ml) * RB, boff = A_TILE + (wc * 64 + ml) * RB;
    int co[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) co[s] = ((2 * s + kl) ^ sw) * 16;

    // ---- prologue: all of tile 0, the B unit(s) of tile 1
#pragma unroll
    for (int h = 0; h < NB; ++h) stageB(0, h);
#pragma unroll
    for (int h = 0; h < NA; ++h) stageA(0, h);
    advanceA();
#pragma unroll
    for (int h = 0; h < NB; ++h) stageB(1, h);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WAIT) : "memory");
    P8_BARRIER();
    if (grp == 1) P8_BARRIER();                                       // the stagger: group 1 runs one barrier behind group 0

    constexpr bool DMA = !(ABL & 1) && ABL != 4, RD = !(ABL & 2) && ABL != 4, BAR = ABL != 4;
    bf16x8_t av[2][4], b0[4], b1[4];
    auto rd_b0 = [&](const char* bb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) b0[s] = *reinterpret_cast<const bf16x8_t*>(bb + co[s]);
    };
    auto rd_b1 = [&](const char* bb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) b1[s] = *reinterpret_cast<const bf16x8_t*>(bb + 32 * RB + co[s]);
    };
    auto rd_a = [&](const char* ab, int i0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) av[i][s] = *reinterpret_cast<const bf16x8_t*>(ab + (i0 + i) * 32 * RB + co[s]);
    };
    auto mm = [&](int i0, int j, const bf16x8_t (&b)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][s], b[s], acc[i0 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    if constexpr (!RD) { rd_b0(lds + boff); rd_b1(lds + boff); rd_a(lds + aoff, 0); }
    auto tile = [&](int t, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const char* ab = lds + buf * STAGE + aoff;
        const char* bb = lds + buf * STAGE + boff;
        // ---- P0
        if constexpr (RD) { rd_b0(bb); if constexpr (WM == 2) rd_b1(bb); rd_a(ab, 0); }
        if constexpr (DMA) { stageA(t + 1, 0); if constexpr (WM == 4) stageA(t + 1, 1); }
        if constexpr (BAR) P8_BARRIER();
        mm(0, 0, b0);
        if constexpr (BAR) P8_BARRIER();
        // ---- P1
        if constexpr (RD && WM == 4) rd_b1(bb);
        if constexpr (DMA) { if constexpr (WM == 2) stageA(t + 1, 1); else { stageA(t + 1, 2); stageA(t + 1, 3); } }
        advanceA();
        if constexpr (BAR) P8_BARRIER();
        mm(0, 1, b1);
        if constexpr (BAR) P8_BARRIER();
        // ---- P2
        if constexpr (RD) rd_a(ab, 2);
        if constexpr (DMA && WM == 2) stageB(t + 2, 0);
        if constexpr (BAR) P8_BARRIER();
        mm(2, 1, b1);
        if constexpr (BAR) P8_BARRIER();
        // ---- P3
        if constexpr (DMA) stageB(t + 2, NB - 1);
        if (p.safe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // reference pipeline (cb_debug_glds_safe): drained queue
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WAIT) : "memory");   // tile t+1 complete; the B unit(s) of tile t+2 stay in flight
        if constexpr (BAR) P8_BARRIER();
        mm(2, 0, b0);
        if constexpr (BAR) P8_BARRIER();
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) { tile(t, std::integral_constant<int, 0>{}); tile(t + 1, std::integral_constant<int, 1>{}); }
    if (t < nk) tile(t, std::integral_constant<int, 0>{});
    if (grp == 0) P8_BARRIER();                                       // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the zero-fill pieces issued past the end of K
    __syncthreads();

    // ---- epilogue: each wave parks its 128 x 64 fp32 sub-tile in its own 16 KiB of the idle ring, 64 rows at a time, and reads it
    // back as 8-column pieces (a lane then owns 8 consecutive outputs of one row: one 16-byte bf16 store; cb_gemm_glds.hip's form)
    float* S = reinterpret_cast<float*>(lds) + wave * (64 * 64);
    const int rsub = lane >> 3, cc = (lane & 7) * 8;
    const int col = n0 + wc * 64 + cc;
    const bool vec = (col + 8 <= a.N);
    float bcol[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bcol[e] = (a.bias && col + e < a.N) ? a.bias[col + e] : 0.f;
    const bf16_t* __restrict__ rowbias = static_cast<const bf16_t*>(a.rowbias);
    const long ldrb = a.ld_rowbias ? a.ld_rowbias : a.N;
    const long MN = (long)a.M * a.N;
    bf16_t* __restrict__ Cp = static_cast<bf16_t*>(a.C) + zb * a.sCb + zh * a.sCh;
    const bf16_t* __restrict__ Rp = a.R ? static_cast<const bf16_t*>(a.R) + zb * a.sRb + zh * a.sRh : nullptr;
    float* W = p.splitk > 1 ? p.ws + ((long)split * a.batch + z) * MN : nullptr;
    const bool v4 = vec && (a.N % 4 == 0);
    const __amdgpu_buffer_rsrc_t srdW = make_srd(W);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    S[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl) * 64 + j * 32 + ml] = acc[2 * hf + i][j][r];
        P8_FENCE();                                                   // (one wave's LDS accesses execute in order: no barrier between its own write and read)
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int rl = ps * 8 + rsub, row = m0 + wr * 128 + hf * 64 + rl;
            if (row >= a.M || col >= a.N) continue;
            const float4 v0 = *reinterpret_cast<const float4*>(S + rl * 64 + cc), v1 = *reinterpret_cast<const float4*>(S + rl * 64 + cc + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (p.splitk > 1) {
                if (p.cnt) { splitk_store8(srdW, W, (int)(((long)row * a.N + col) * 4), v, v4, a.N - col); continue; }
                float* w = W + (long)row * a.N + col;
                if (v4) { *reinterpret_cast<float4*>(w) = v0; *reinterpret_cast<float4*>(w + 4) = v1; }
                else for (int e = 0; e < 8 && col + e < a.N; ++e) w[e] = v[e];
                continue;
            }
            glds_finish_piece<bf16_t>(a, v, bcol, Cp, Rp, rowbias, ldrb, row, col, vec);
        }
        P8_FENCE();
    }
    if (p.splitk <= 1 || !p.cnt) return;
    // in-launch split-K reduction: the tile's last-arriving block folds the slabs in split order (cb_gemm_tiles.h, splitk_arrive)
    if (!splitk_arrive(p.cnt + (long)z * ntiles + t_, p.splitk, reinterpret_cast<int*>(lds))) return;
    const float* W0 = p.ws + (long)z * MN;
    for (int hf = 0; hf < 2; ++hf)
        for (int ps = 0; ps < 8; ++ps) {
            const int rl = ps * 8 + rsub, row = m0 + wr * 128 + hf * 64 + rl;
            if (row >= a.M || col >= a.N) continue;
            float v[8];
            splitk_fold8(v, W0 + (long)row * a.N + col, (long)a.batch * MN, p.splitk, v4, a.N - col);
            glds_finish_piece<bf16_t>(a, v, bcol, Cp, Rp, rowbias, ldrb, row, col, vec);
        }
}

template <int AK, int WM, int ABL>
__global__ __launch_bounds__(512) void gemm_bf16_p8_kernel(const KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds_p8[];
    gemm_bf16_p8_body<AK, WM, ABL>(p, lds_p8);
}

// ---------------------------------------------------------------- host side
bool gemm_bf16_p8_ok(const GemmArgs& a) {          // (every split's k range is a whole number of 64-element k-tiles by construction of the plans)
    if (a.dt != DT_BF16 || a.b_kind != LD_ROW || a.K % p8::BK) return false;
    if (a.a_kind == LD_CONV) return !a.g.up && !a.g.tr && a.g.C % p8::BK == 0;
    return a.a_kind == LD_ROW;
}
template <int AK, int WM, int ABL>
static void p8_go(dim3 grid, hipStream_t s, const KArgs& ka) {
    set_dyn_lds(reinterpret_cast<const void*>(&gemm_bf16_p8_kernel<AK, WM, ABL>), p8::Geo<WM>::LDS);
    hipLaunchKernelGGL((gemm_bf16_p8_kernel<AK, WM, ABL>), grid, dim3(p8::NTHR), p8::Geo<WM>::LDS, s, ka);
}
// bn = 256: the 256 x 256 tile; bn = 128: the 512 x 128 tile
void gemm_bf16_p8_launch(int bn, dim3 grid, hipStream_t s, const KArgs& ka) {
    static const int abl = [] { const char* e = getenv("CB_P8_ABL"); return e ? atoi(e) : 0; }();       // diagnostics: WRONG results
    const bool conv = ka.a.a_kind == LD_CONV;
    if (bn == 128) { if (conv) p8_go<P8_CONVF, 4, 0>(grid, s, ka); else p8_go<P8_ROW, 4, 0>(grid, s, ka); return; }
    if (conv) { p8_go<P8_CONVF, 2, 0>(grid, s, ka); return; }
    if (abl == 1) p8_go<P8_ROW, 2, 1>(grid, s, ka);
    else if (abl == 2) p8_go<P8_ROW, 2, 2>(grid, s, ka);
    else if (abl == 3) p8_go<P8_ROW, 2, 3>(grid, s, ka);
    else if (abl == 4) p8_go<P8_ROW, 2, 4>(grid, s, ka);
    else p8_go<P8_ROW, 2, 0>(grid, s, ka);
}

}  // namespace cb
