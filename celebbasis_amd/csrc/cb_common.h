// cb_common.h — shared declarations for the CelebBasis MI355X engine (gfx950 only).
// Internal header: the public C ABI lives in include/celebbasis_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <stdexcept>
#include <cstdarg>
#include <vector>

namespace cb {

// ---------------------------------------------------------------- errors
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
enum { CB_OK = 0, CB_ERR_ARG = -1, CB_ERR_HIP = -2, CB_ERR_SHAPE = -3, CB_ERR_MISSING = -4, CB_ERR_OOM = -5,
       CB_ERR_STATE = -6 };

[[noreturn]] inline void fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw Error(code, buf);
}
#define CB_CHECK(cond, code, ...) do { if (!(cond)) ::cb::fail(code, __VA_ARGS__); } while (0)
#define CB_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    ::cb::fail(::cb::CB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// ---------------------------------------------------------------- element types
// Activations and frozen weights of UNet / VAE / CLIP are stored either as fp32 (BASELINE config 2: the reference's own
// arithmetic) or as bf16 (config 3: bf16 storage + bf16 MFMA with fp32 accumulation; normalisation statistics, softmax, the
// coefficient head, its gradient and AdamW stay fp32). A tensor's element type travels with it (Ten::dt, GemmArgs::dt, ...).
enum DType { DT_F32 = 0, DT_BF16 = 1 };
typedef __bf16 bf16_t;
inline size_t dt_size(DType d) { return d == DT_BF16 ? 2 : 4; }
inline void* ptr_off(void* p, DType dt, long elems) { return static_cast<char*>(p) + elems * (long)dt_size(dt); }
inline const void* ptr_off(const void* p, DType dt, long elems) { return static_cast<const char*>(p) + elems * (long)dt_size(dt); }

// ---------------------------------------------------------------- launch context
// Every kernel launcher takes a Launch. In dry mode (no device) launchers only do shape checks and
// accounting: this is how the host-side graph logic is tested on a box without a GPU.
struct Stats {
    double gemm_flops = 0;      // 2*M*N*K summed over GEMM-class launches
    long   launches = 0;
    long   gemm_launches = 0;
};
// per-launch HIP-event timing of the GEMM-class kernels (bench.py's live roofline measurement)
struct ProfRec { int variant; double flops; hipEvent_t e0, e1; double bytes = 0; };
struct Profiler {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<std::string> names;
    // helpers for non-GEMM launch groups: begin() returns false when profiling is off
    bool begin(ProfRec* r, const char* name, double flops, double bytes, hipStream_t s) {
        if (!on) return false;
        r->variant = id_of(name); r->flops = flops; r->bytes = bytes;
        if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) return false;
        (void)hipEventRecord(r->e0, s);
        return true;
    }
    void end(ProfRec* r, hipStream_t s) { (void)hipEventRecord(r->e1, s); recs.push_back(*r); }
    int id_of(const std::string& n) {
        for (size_t i = 0; i < names.size(); ++i) if (names[i] == n) return (int)i;
        names.push_back(n);
        return (int)names.size() - 1;
    }
};
// Arrival counters of the in-launch split-K reduction (cb_gemm_glds.hip): device ints that are ZERO between launches (zeroed once
// when the pool is created; the last-arriving block of a tile puts its counter back to zero). A launch takes a fresh range, the
// ranges are rewound together with the step arena (same ranges every step: replay-stable like the arena's addresses).
struct CounterPool {
    int* base = nullptr; size_t cap = 0, off = 0;
    bool dry = false;            // dry context: a placeholder address (nothing is launched), so that launch accounting matches a real one
    int* take(size_t n) {
        if (dry) return reinterpret_cast<int*>(0x300000);
        n = (n + 31) / 32 * 32;
        if (!base || off + n > cap) return nullptr;      // exhausted: the caller falls back to the separate reduce launch
        int* p = base + off; off += n; return p;
    }
};
struct Launch {
    hipStream_t stream = nullptr;
    bool dry = false;
    Stats* stats = nullptr;
    Profiler* prof = nullptr;
    CounterPool* cnt = nullptr;
    void count(long n = 1) const { if (stats) stats->launches += n; }
};

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute belongs to the
// CURRENT device: remembered per (kernel, device), so a second context on another GPU of the same process sets it again.
void set_dyn_lds(const void* kernel, int bytes);

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }
inline long rup(long a, long b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------- conv geometry
// Implicit-GEMM gather description (NHWC input). Virtual input = source upsampled 2x when `up`.
struct ConvGeom {
    int H = 0, W = 0, C = 0;    // source dims (NHWC), C = channels of the gathered tensor
    int Ho = 0, Wo = 0;         // output spatial dims
    int ks = 3;                 // kernel size 1 or 3 (2 and 4 on the LDS-DMA kernel: the 4x4 stride-2 up-conv input gradient)
    int stride = 1;             // step of the virtual coordinate per output pixel
    int pad = 1;                // top/left padding in virtual coordinates
    int up = 0;                 // 1: nearest-2x upsample fused: source pixel = virtual >> 1
    int tr = 0;                 // 1: transposed stride-2 (dgrad of a stride-2 conv): virtual must be even, source = virtual >> 1
};

// ---------------------------------------------------------------- GEMM arguments
enum { LD_ROW = 0, LD_COL = 1, LD_CONV = 2, LD_CONVF = 3 /* kernel-internal: conv with the scalar tap walk (cb_gemm.hip) */ };
struct GemmArgs {
    DType dt = DT_F32;          // element type of A, B, C, R and rowbias (bias is always fp32; split-K partials are fp32)
    const void* A = nullptr; const void* B = nullptr; void* C = nullptr;
    int M = 0, N = 0, K = 0;
    long lda = 0, ldb = 0, ldc = 0;
    int a_kind = LD_ROW, b_kind = LD_ROW;
    // batch: z = b*nh + h ; offset = b*s?b + h*s?h
    int batch = 1, nh = 1;
    long sAb = 0, sAh = 0, sBb = 0, sBh = 0, sCb = 0, sCh = 0;
    // epilogue: C = alpha*acc + bias[n] + rowbias[m / rows_per_img][n] + R[m][n] + beta*C
    const float* bias = nullptr;
    const void* rowbias = nullptr; int rows_per_img = 1; long ld_rowbias = 0;   // ld 0 = N
    const void* R = nullptr; long ldr = 0; long sRb = 0, sRh = 0;
    float alpha = 1.f, beta = 0.f;
    // LDS-DMA kernels only (the frozen FR backbone, iresnet.py:44-63): v = PReLU(v, prelu[n]) before the store, and a second output
    // C2[m][n] = c2_scale[n] * v + c2_shift[n] (leading dimension ldc) — the NEXT block's eval BatchNorm of this block's output
    const float* prelu = nullptr;
    void* C2 = nullptr; const float* c2_scale = nullptr; const float* c2_shift = nullptr;
    ConvGeom g;
};

void gemm(const Launch& L, const GemmArgs& a, float* splitk_ws, size_t splitk_ws_bytes);
size_t gemm_ws_bytes(const GemmArgs& a);
// LDS-DMA kernels (cb_gemm_glds.hip): k-contiguous operands only (ROW / CONV x ROW), 16-byte aligned rows.
// bf16: K % 8 == 0. fp32: gemm() itself routes every eligible problem there (gemm_f32_glds returns false otherwise).
void gemm_bf16(const Launch& L, const GemmArgs& a, float* splitk_ws, size_t splitk_ws_bytes);
size_t gemm_bf16_ws_bytes(const GemmArgs& a);
bool gemm_f32_glds_shape_ok(const GemmArgs& a);
size_t gemm_f32_glds_ws_bytes(const GemmArgs& a);
bool gemm_f32_glds(const Launch& L, const GemmArgs& a, float* splitk_ws, size_t splitk_ws_bytes);

// ---------------------------------------------------------------- fused attention (cb_flash.hip)
struct FlashArgs {
    DType dt = DT_F32;               // element type of q, k, v, o, dout, dq, dk, dv (lse, dvec and the workspace are fp32)
    const void* q = nullptr; long ldq = 0, sqb = 0, sqh = 0;
    const void* k = nullptr; const void* v = nullptr; long ldk = 0, skb = 0, skh = 0;
    void* o = nullptr; long ldo = 0, sob = 0, soh = 0;
    float* lse = nullptr;            // [B*H][Nq]
    int H = 1, Nq = 0, Nk = 0, d = 0;
    float scale = 1.f; int causal = 0;
    const void* dout = nullptr;      // layout of o
    float* dvec = nullptr;           // [B*H][Nq]  rowsum(dO * O)
    void* dq = nullptr; float beta_q = 0.f;                      // layout of q
    void* dk = nullptr; void* dv = nullptr; float beta_kv = 0.f; // layout of k / v
    float* ws = nullptr; int nsplit = 1;   // dK/dV query-split partials [2][nsplit][B*H][Nk][d] (cross-attention: few keys, many queries)
    // key-split of the forward / dQ kernels when Nq/64 * B*H blocks cannot fill the chip (the 32x32 .. 8x8 levels): blockIdx.z owns
    // a key range; forward partials [ksplit][B*H][Nq][d] (unnormalised O) + [ksplit][B*H][Nq][2] (max, sum), dQ partials
    // [ksplit][B*H][Nq][d]; one small combine kernel each (fp32 kernels only)
    float* wsk = nullptr; int ksplit = 1;
    // bf16, second-generation kernels (cb_flash_bf16_v2.hip): scratch for the operands that are streamed TRANSPOSED
    // ([B*H][16*ceil(d/16)][N] bf16 each: forward V^T; backward K^T | Q^T | dO^T), sized by flash_bf16_v2_tws_bytes. Non-null selects v2.
    void* tws = nullptr;
};
size_t flash_ksplit_ws_floats(const FlashArgs& a, int B, bool forward, int* ksplit);
size_t flash_bwd_ws_floats(const FlashArgs& a, int B, int* nsplit);
bool flash_supported(int d, const FlashArgs& a);
void flash_forward(const Launch& L, const FlashArgs& a, int B);
void flash_backward(const Launch& L, const FlashArgs& a, int B, bool need_dq, bool need_dkv);
// bf16 twins (cb_flash_bf16.hip): d % 8 == 0, d <= 160
size_t flash_bf16_bwd_ws_floats(const FlashArgs& a, int B, int* nsplit);
bool flash_bf16_supported(int d, const FlashArgs& a);
void flash_bf16_forward(const Launch& L, const FlashArgs& a, int B);
void flash_bf16_backward(const Launch& L, const FlashArgs& a, int B, bool need_dq, bool need_dkv);
// v2 (LDS-DMA staged, 16*QB rows per wave, pre-transposed second operands): 0 bytes = not applicable to this problem
size_t flash_bf16_v2_tws_bytes(const FlashArgs& a, int B, bool backward);
void flash_bf16_v2_forward(const Launch& L, const FlashArgs& a, int B, void* tws);
void flash_bf16_v2_backward(const Launch& L, const FlashArgs& a, int B, bool need_dq, bool need_dkv, void* tws);

// ---------------------------------------------------------------- other kernels (cb_kernels.hip)
// Launchers whose tensors are activations take the element type `dt` of those tensors (void pointers); parameters, statistics,
// workspaces, losses and everything of the coefficient head are fp32 and keep typed pointers.
void k_fill(const Launch& L, DType dt, void* p, long n, float v);
void k_axpby(const Launch& L, DType dt, void* dst, const void* src, long n, float a, float b);          // dst = a*src + b*dst
void k_copy2d(const Launch& L, DType dt, void* dst, long ldd, const void* src, long lds, long rows, int cols, float beta);
// dst[rows][cols_dst] = src[rows][cols_src] (+ zero columns) with a change of element type: fp32 <-> bf16 casts at the edges of
// the bf16 graph, and the channel padding (3 or 4 -> 8) that gives the first convolutions 16-byte rows
void k_convert(const Launch& L, void* dst, DType ddt, int cols_dst, const void* src, DType sdt, int cols_src, long rows, float beta = 0.f);
void k_transpose(const Launch& L, DType dt, void* dst, const void* src, int batch, int rows, int cols, long ld_src, long sb_src);  // dst[b][cols][rows]
void k_silu(const Launch& L, DType dt, void* dst, const void* src, long n);
void k_geglu_fwd(const Launch& L, DType dt, void* y, const void* x, long M, int D);
void k_geglu_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, long M, int D);
void k_quickgelu_fwd(const Launch& L, DType dt, void* y, const void* x, long n);
void k_quickgelu_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, long n, float beta);
void k_leakyrelu_fwd(const Launch& L, float* y, const float* x, long n, float slope);
void k_leakyrelu_bwd(const Launch& L, float* dx, const float* x, const float* dy, long n, float slope);
void k_nchw_to_nhwc(const Launch& L, DType ddt, void* dst, const float* src, int N, int C, int H, int W);   // public NCHW tensors are fp32
void k_nhwc_to_nchw(const Launch& L, DType sdt, float* dst, const void* src, int N, int C, int H, int W);
void k_channel_affine(const Launch& L, float* y, const float* x, const float* scale, const float* shift,
                      const float* prelu, long rows, int C);
void k_upsample2x_bwd(const Launch& L, DType dt, void* dx, const void* dy, int N, int H, int W, int C, float beta);

// GroupNorm (NHWC, G groups). stats = [N][G][2] (mean, rstd). partial ws sized gn_ws_floats(). x, y, dy, dx are `dt`.
size_t gn_ws_floats(int N, int HW, int C, int G);
void k_groupnorm_fwd(const Launch& L, DType dt, void* y, const void* x, const float* gamma, const float* beta,
                     float* stats, float* ws, int N, int HW, int C, int G, float eps, int silu);
void k_groupnorm_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, const float* gamma, const float* beta,
                     const float* stats, float* ws, int N, int HW, int C, int G, int silu, float acc_beta, bool chn_valid = false);
// LayerNorm over last dim C; stats = [rows][2]
void k_layernorm_fwd(const Launch& L, DType dt, void* y, const void* x, const float* gamma, const float* beta,
                     float* stats, long rows, int C, float eps);
void k_layernorm_bwd(const Launch& L, DType dt, void* dx, const void* x, const void* dy, const float* gamma,
                     const float* stats, long rows, int C, float acc_beta);
// softmax over rows of length n (leading dim ld), in place. causal: row i (mod rows_per_mat) sees cols <= i.
void k_softmax_fwd(const Launch& L, DType dt, void* S, long rows, int n, int ld, int causal, int rows_per_mat);
// dS = scale * P * (dP - sum(dP*P)), in place in dP (fp32 materialised path only)
void k_softmax_bwd(const Launch& L, float* dP, const float* P, long rows, int n, int ld, float scale);

// diffusion-specific
void k_timestep_embedding(const Launch& L, DType dt, void* out, const int64_t* t, int B, int dim);
void k_qsample(const Launch& L, DType dt, void* xt, const void* x0, const float* eps_nchw, const int64_t* t,
               const float* sqrt_ac, const float* sqrt_1mac, int B, int C, int HW);
void k_mse_loss(const Launch& L, DType dt, float* loss, void* dpred, const void* pred, const float* target_nchw,
                int B, int C, int HW);
// x and pred_x0 are the fp32 sampler state; eps is the UNet output (`dt`)
void k_ddim_update(const Launch& L, DType dt, float* x, float* pred_x0, const void* eps, const float* noise_nchw, int n, int C, int HW,
                   int cfg, float scale, float sqrt_1m_at, float sqrt_at, float sqrt_aprev, float dir_coef, float sigma);
void k_fill_i64(const Launch& L, int64_t* p, long n, int64_t v);
void k_image_out(const Launch& L, DType sdt, float* dst_nchw, const void* src_nhwc, int N, int C, int HW, int to01);
void k_vae_sample(const Launch& L, DType dt, void* z, const void* moments, const float* noise_nchw, int B, int HW, int Cz, float scale);
void k_gather_rows(const Launch& L, float* out, const float* table, const int64_t* idx, long rows, int C, long table_rows);
void k_add_rowbcast(const Launch& L, DType dt, void* dst, const void* x, const float* pos, long rows, int period, int C);
// face pre-processing (meta_net.py:250-260)
void k_affine_warp(const Launch& L, float* out, const float* faces, int B, int S, int nface, const float* theta,
                   const float* grid = nullptr /* [S][S][2] sampling grid of the caller (Ctx::warp_grid), or the analytic one */);
void k_resize_bilinear_ac(const Launch& L, float* out, const float* in, int N, int Hin, int Win, int Hout, int Wout, int C);
// feeder augmentations (face_id.py:526-532, 712-730)
size_t face_augment_ws_floats(int B);
void k_face_augment(const Launch& L, float* out, const float* in, const float* params_dev, float* ws, int B, int S);
void k_paste_bg(const Launch& L, float* out, const float* img, const int* geom_dev, int B, int S, int C);
// manager (fp32)
void k_l2norm_fwd(const Launch& L, float* y, float* norms, const float* x, long rows, int D);
void k_l2norm_bwd(const Launch& L, float* dx, const float* y, const float* norms, const float* dy, long rows, int D);
void k_basis_fwd(const Launch& L, float* z, const float* coef, const float* basis, int R, int E, int Kc, int Cz);
void k_basis_bwd(const Launch& L, float* dcoef, const float* dz, const float* basis, int R, int E, int Kc, int Cz);
void k_row_surgery_fwd(const Launch& L, float* out, const float* in, const float* z, const int64_t* tokens,
                       int32_t* map, int B, int T, int C, int reps, int64_t placeholder);
void k_row_surgery_multi(const Launch& L, float* out, const float* in, const float* z, const int64_t* tokens, int32_t* map, int B,
                         int T, int C, int reps, const int64_t* placeholders_dev, int nph);
void k_row_surgery_bwd(const Launch& L, float* dz, const float* dout, const int32_t* map, int B, int T, int C, int reps);
void k_ema_update(const Launch& L, float* id_emb, float* id_coef, const float* z, const float* coef,
                  const int64_t* ids, int ids_stride, int B, int embN, int coefN, int max_ids, float m);
// use_rm_mlp (meta_net.py:272-273): out[f*B + b][:] = coef[ids[b][f]][:] ; gcoef = dense index-backward of the same gather
void k_coef_gather(const Launch& L, float* out, const float* coef, const int64_t* ids, int ids_stride, int B, int nface, int D, int max_ids);
void k_coef_scatter_grad(const Launch& L, float* gcoef, const float* dout, const int64_t* ids, int ids_stride, int B, int nface, int D,
                         int max_ids);
void k_outer_grad(const Launch& L, float* dW, float* db, const float* dpre, const float* v, int R, int O, int I);
void k_adamw(const Launch& L, float* p, float* m, float* v, const float* g, long n, const int* step,
             float lr, float b1, float b2, float eps, float wd, const float* lr_dev = nullptr /* non-null: the rate is read from this device scalar */);
void k_step_inc(const Launch& L, int* step);
// y[M][N] = x[M][K] . W[N][K]^T + bias, M <= 8 (x, W, y are `dt`; bias fp32)
void k_gemv(const Launch& L, DType dt, void* y, const void* x, const void* W, const float* bias, int M, int N, int K);
// weight packing (source: the fp32 state-dict tensor; packs are `dt` with the channel counts padded to Ip / Op >= I / O)
void k_pack_conv(const Launch& L, DType dt, void* out, const float* w_oihw, int O, int I, int ks, int Ip);          // -> [O][ky][kx][Ip]
void k_pack_conv_up_dgrad(const Launch& L, DType dt, void* out, const float* w, int O, int I, int Op);   // [I][4][4][Op], see cb_kernels.hip
void k_pack_conv_dgrad(const Launch& L, DType dt, void* out, const float* w_oihw, int O, int I, int ks, int Op);   // -> [I][ky'][kx'][Op], flipped
void k_pack_fc_nhwc(const Launch& L, float* out, const float* w, int O, int C, int HW);            // [O][C*HW] -> [O][HW*C]
void k_fold_bn(const Launch& L, float* w_packed, float* bias_out, const float* bn_w, const float* bn_b,
               const float* bn_m, const float* bn_v, float eps, int O, long per_o);
void k_bn_to_affine(const Launch& L, float* scale, float* shift, const float* w, const float* b, const float* m,
                    const float* v, float eps, int C);

// fp32 conveniences for the coefficient head, the FR-net and the weight loaders
inline void k_fill(const Launch& L, float* p, long n, float v) { k_fill(L, DT_F32, p, n, v); }
inline void k_axpby(const Launch& L, float* dst, const float* src, long n, float a, float b) { k_axpby(L, DT_F32, dst, src, n, a, b); }

}  // namespace cb
