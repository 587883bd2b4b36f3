// cb_gemm_bf16.hip — bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X): BASELINE config 3.
//
//   C[M,N] = alpha * A[M,K] . B[N,K]^T + bias[n] + rowbias[img(m)][n] + R[m][n] + beta * C[m][n]
//
// A, B, C, R and rowbias are bf16 in HBM, the accumulator is fp32 (v_mfma_f32_32x32x16_bf16: 2.5 PF dense chip peak), bias
// and the split-K partials are fp32. Both operands are k-contiguous (activations / conv gathers as A, [out][in] weights as B:
// the host keeps a transposed copy of a Linear weight for its input gradient instead of an m-contiguous loader), 16-byte
// aligned with K % 8 == 0, so every load is one branch-free `buffer_load_dwordx4` with hardware zero fill (cb_gemm_tiles.h).
//
// Structure = the fp32 kernel's (cb_gemm.hip): 256 threads = 4 wave64 in 2x2, each wave a (BM/2)x(BN/2) sub-tile of 32x32
// MFMA tiles; a k-tile is 64 bytes per row = 32 bf16 (BK = 32); LDS image [row][4 x 16-byte chunks] XOR-swizzled by
// (row >> 2) & 3, one conflict-free ds_write_b128 per global load and one conflict-free ds_read_b128 per MFMA operand.
// A lane (row = lane & 31, kl = lane >> 5) owns the chunks 2*kl and 2*kl+1 of its row; MFMA step `half` feeds chunk 2*kl + half,
// i.e. the 8 k-slots of half-wave kl hold k = 16*kl + 8*half .. +7 — any assignment of k to MFMA k-slots is a valid contraction
// as long as A and B use the same one, and they do by construction.
// What differs from fp32 is the balance: one k-tile is 2 x TM x TN MFMAs of 32 cycles (fp32: 8 x TM x TN of 64), so the kernel
// is bound by the global->LDS staging, not by the matrix pipe; profiles/README.md (round 2) has the measurements.
#include "cb_gemm_tiles.h"

namespace cb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BKE = 32;          // k elements per tile
constexpr int ROWB = 64;         // bytes per LDS row

enum { AK_ROW = 0, AK_CONV = 1, AK_CONVF = 2 };

template <int BM, int BN, int AK>
__device__ __forceinline__ void gemm_bf16_body(const KArgs& p, char* lds) {
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

    const bf16_t* __restrict__ Ap = static_cast<const bf16_t*>(a.A) + zb * a.sAb + zh * a.sAh;
    const bf16_t* __restrict__ Bp = static_cast<const bf16_t*>(a.B) + zb * a.sBb + zh * a.sBh;

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
    ConvRows<BM, 2> cr;
    ConvFast<BM, 2> cf;
    int tap_u = 0, ci_u = 0;                                                                   // scalar decode of k0 (AK_CONVF)
    if constexpr (AK == AK_CONVF) { cf.init(a.g, m0, a.M, tid); tap_u = kbeg / a.g.C; ci_u = kbeg - tap_u * a.g.C; }
    else if constexpr (AK == AK_CONV) cr.init(a.g, m0, a.M, kbeg, tid);

    __amdgpu_buffer_rsrc_t srdA = make_srd(Ap), srdB = make_srd(Bp);
    RowOffs<BM, 2> roA; RowOffs<BN, 2> roB;
    if constexpr (AK == AK_ROW) roA.init(a.lda, m0, a.M, tid);
    roB.init(a.ldb, n0, a.N, tid);
    auto loadAB = [&](int k0) {
        int oa[BM / 64], ob[BN / 64];
        if constexpr (AK == AK_ROW) offs_row<BM, 2>(oa, roA, k0, kend, tid);
        else if constexpr (AK == AK_CONVF) {
            const int ky = (a.g.ks == 3) ? ((tap_u * 11) >> 5) : 0;
            const int kx = tap_u - ky * a.g.ks;
            offs_conv_fast<BM, 2>(oa, cf, tap_u, ((ky * a.g.W + kx) * a.g.C + ci_u) * 2);
            ci_u += BKE;
            if (ci_u >= a.g.C) { ci_u -= a.g.C; ++tap_u; }
        } else { offs_conv<BM, 2>(oa, a.g, cr, k0, kend, tid); cr.advance(a.g); }
        offs_row<BN, 2>(ob, roB, k0, kend, tid);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < BM / 64; ++i) ra[i] = buf_ld4(srdA, oa[i]);
#pragma unroll
        for (int i = 0; i < BN / 64; ++i) rb[i] = buf_ld4(srdB, ob[i]);
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int A_BUF = BM * ROWB, B_BUF = BN * ROWB;            // bytes per buffer
    char* const As = lds;
    char* const Bs = lds + 2 * A_BUF;

    if (kbeg < kend) {
        loadAB(kbeg);
        store_kc<BM>(reinterpret_cast<float*>(As), ra, tid);
        store_kc<BN>(reinterpret_cast<float*>(Bs), rb, tid);
    }
    __syncthreads();

    int cur = 0;
    const int kl = lane >> 5, ml = lane & 31;
    const int swz = (ml >> 2) & 3;                                           // == ((row >> 2) & 3): sub-tile bases are multiples of 32
    const int clo = ((2 * kl) ^ swz) * 16, chi = ((2 * kl + 1) ^ swz) * 16;  // byte offsets of the lane's two chunks
    for (int k0 = kbeg; k0 < kend; k0 += BKE) {
        const bool more = (k0 + BKE) < kend;
        if (more) loadAB(k0 + BKE);

        const char* arow = As + cur * A_BUF + (wm * (BM / 2) + ml) * ROWB;
        const char* brow = Bs + cur * B_BUF + (wn * (BN / 2) + ml) * ROWB;
        bf16x8_t av[2][TM], bv[2][TN];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[half][i] = *reinterpret_cast<const bf16x8_t*>(arow + i * 32 * ROWB + (half ? chi : clo));
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[half][j] = *reinterpret_cast<const bf16x8_t*>(brow + j * 32 * ROWB + (half ? chi : clo));
        }
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[half][i], bv[half][j], acc[i][j], 0, 0, 0);

        if (more) {
            store_kc<BM>(reinterpret_cast<float*>(As + (cur ^ 1) * A_BUF), ra, tid);
            store_kc<BN>(reinterpret_cast<float*>(Bs + (cur ^ 1) * B_BUF), rb, tid);
        }
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
    bf16_t* __restrict__ Cp = static_cast<bf16_t*>(a.C) + zb * a.sCb + zh * a.sCh;
    const bf16_t* __restrict__ Rp = a.R ? static_cast<const bf16_t*>(a.R) + zb * a.sRb + zh * a.sRh : nullptr;
    const bf16_t* __restrict__ rowbias = static_cast<const bf16_t*>(a.rowbias);
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
                    if (rowbias) v += (float)rowbias[(long)(row / a.rows_per_img) * (a.ld_rowbias ? a.ld_rowbias : a.N) + col];
                    if (Rp) v += (float)Rp[(long)row * a.ldr + col];
                    bf16_t* c = Cp + (long)row * a.ldc + col;
                    if (a.beta != 0.f) v += a.beta * (float)(*c);
                    *c = (bf16_t)v;
                }
            }
        }
}

template <int BM, int BN, int AK>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const KArgs p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * ROWB];
    gemm_bf16_body<BM, BN, AK>(p, lds);
}

// ---------------------------------------------------------------- host side
size_t gemm_bf16_ws_bytes(const GemmArgs& a) {
    Plan pl = make_plan(a, BKE, plan_knobs().bf16_speed);
    if (pl.splitk <= 1) return 0;
    return (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
}

template <int AK>
static void launch_tiles_bf16(const Plan& pl, dim3 grid, hipStream_t s, const KArgs& ka) {
    if (pl.BM == 128 && pl.BN == 128) hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, AK>), grid, dim3(256), 0, s, ka);
    else if (pl.BM == 128 && pl.BN == 64) hipLaunchKernelGGL((gemm_bf16_kernel<128, 64, AK>), grid, dim3(256), 0, s, ka);
    else if (pl.BM == 64 && pl.BN == 128) hipLaunchKernelGGL((gemm_bf16_kernel<64, 128, AK>), grid, dim3(256), 0, s, ka);
    else hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, AK>), grid, dim3(256), 0, s, ka);
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
        CB_CHECK(g.ks == 1 || g.ks == 3, CB_ERR_SHAPE, "conv: ks=%d", g.ks);
        CB_CHECK(a.K == g.ks * g.ks * g.C, CB_ERR_SHAPE, "conv: K=%d != %d*%d*%d", a.K, g.ks, g.ks, g.C);
        CB_CHECK(g.C % 8 == 0, CB_ERR_SHAPE, "conv bf16: C=%d must be a multiple of 8 (pad the channel axis)", g.C);
        CB_CHECK(a.M % (g.Ho * g.Wo) == 0, CB_ERR_SHAPE, "conv: M=%d not a multiple of Ho*Wo=%d", a.M, g.Ho * g.Wo);
        CB_CHECK(!(g.up && g.tr), CB_ERR_ARG, "conv: up and tr are exclusive");
    } else {
        CB_CHECK(a.lda % 8 == 0 && a.sAb % 8 == 0 && a.sAh % 8 == 0, CB_ERR_SHAPE, "gemm_bf16: A strides must be multiples of 8 elements");
    }
    CB_CHECK(a.ldb % 8 == 0 && a.sBb % 8 == 0 && a.sBh % 8 == 0, CB_ERR_SHAPE, "gemm_bf16: B strides must be multiples of 8 elements");
    Plan pl = make_plan(a, BKE, plan_knobs().bf16_speed);
    if (L.stats) { L.stats->gemm_flops += 2.0 * a.M * a.N * (double)a.K * a.batch; L.stats->gemm_launches++; }
    L.count(pl.splitk > 1 ? 2 : 1);
    if (pl.splitk > 1) {
        size_t need = (size_t)pl.splitk * a.batch * (size_t)a.M * a.N * sizeof(float);
        CB_CHECK(L.dry || (ws && ws_bytes >= need), CB_ERR_OOM, "gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
    }
    if (L.dry) return;
    auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    CB_CHECK(aligned16(a.A) && aligned16(a.B), CB_ERR_ARG, "gemm_bf16: operands must be 16-byte aligned");
    {   // 32-bit byte offsets inside one operand slice (buffer loads)
        const long lim = (1L << 31) - 64;
        long ea = a.a_kind == LD_CONV ? (long)(a.M / (a.g.Ho * a.g.Wo)) * a.g.H * a.g.W * a.g.C : (long)a.M * a.lda;
        long eb = (long)a.N * a.ldb;
        CB_CHECK(ea * 2 < lim && eb * 2 < lim, CB_ERR_SHAPE, "gemm_bf16: an operand slice exceeds 2 GiB");
    }
    KArgs ka;
    ka.a = a;
    ka.tiles_m = pl.tiles_m; ka.tiles_n = pl.tiles_n;
    ka.splitk = pl.splitk; ka.k_per_split = pl.k_per_split;
    ka.ws = ws; ka.a_vec = 1; ka.b_vec = 1;
    dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk, a.batch);
    const bool convf = a.a_kind == LD_CONV && !a.g.up && !a.g.tr && (a.g.C % BKE == 0);
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
        snprintf(nm, sizeof nm, "gemm_bf16_kernel<%d,%d,%s>", pl.BM, pl.BN, a.a_kind == LD_ROW ? "ROW" : (convf ? "CONVF" : "CONV"));
        rec.variant = L.prof->id_of(nm);
        CB_HIP(hipEventRecord(rec.e1, L.stream)); L.prof->recs.push_back(rec);
    }
    if (pl.splitk > 1) {
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
