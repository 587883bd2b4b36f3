// cb_abi.hip — extern "C" surface of libcelebbasis_hip.so (declared in include/celebbasis_hip.h).
#include "../../include/celebbasis_hip.h"
#include "cb_models.h"
#include "cb_gemm_tiles.h"

using namespace cb;

static thread_local std::string g_err;

#define CB_TRY try {
#define CB_END                                                                     \
    }                                                                              \
    catch (const cb::Error& e) { g_err = e.what(); return e.code; }                \
    catch (const std::exception& e) { g_err = e.what(); return CB_STATUS_INTERNAL; } \
    catch (...) { g_err = "unknown error"; return CB_STATUS_INTERNAL; }            \
    return CB_STATUS_OK;

struct cb_ctx { Ctx c; };
struct cb_wtab { WTab t; };
struct cb_unet { Ctx* ctx; UNet U; std::unique_ptr<Graph> G; Ten* out = nullptr; Ten* context = nullptr; int B = 0, H = 0, W = 0, L = 0; };
struct cb_vae { Ctx* ctx; VaeEnc V; };
struct cb_vae_dec { Ctx* ctx; VaeDec V; };
struct cb_clip { Ctx* ctx; Clip C; std::unique_ptr<Graph> G; Ten* out = nullptr; Ten* in = nullptr; };
struct cb_frnet { Ctx* ctx; FrNet F; };
struct cb_trainer { Ctx* ctx; Trainer T; };

static void require(const void* p, const char* what) { CB_CHECK(p != nullptr, CB_ERR_ARG, "%s must not be NULL", what); }

extern "C" {

const char* cb_last_error(void) { return g_err.c_str(); }
int cb_version(void) { return CB_ABI_VERSION; }

// ------------------------------------------------------------------ context
int cb_ctx_create(int device, void* hip_stream, cb_ctx** out) { return cb_ctx_create_dtype(device, hip_stream, CB_DTYPE_F32, out); }
int cb_ctx_dtype(cb_ctx* ctx, int* dtype) {
    CB_TRY
    require(ctx, "ctx"); require(dtype, "dtype");
    *dtype = (int)ctx->c.dtype;
    CB_END
}
int cb_ctx_aux_stream(cb_ctx* ctx, void** hip_stream) {
    CB_TRY
    require(ctx, "ctx"); require(hip_stream, "hip_stream");
    *hip_stream = static_cast<void*>(ctx->c.aux);
    CB_END
}
int cb_ctx_create_dtype(int device, void* hip_stream, int dtype, cb_ctx** out) {
    CB_TRY
    require(out, "out");
    CB_CHECK(dtype == CB_DTYPE_F32 || dtype == CB_DTYPE_BF16, CB_ERR_ARG, "unknown dtype %d (0 = f32, 1 = bf16)", dtype);
    std::unique_ptr<cb_ctx> h(new cb_ctx());
    Ctx& c = h->c;
    c.dtype = c.load_dt = (DType)dtype;
    c.device = device;
    c.dry = device < 0;
    c.arena.dry = c.dry; c.arena.cnt.dry = c.dry; c.pool.dry = c.dry;
    if (!c.dry) {
        int n = 0;
        CB_HIP(hipGetDeviceCount(&n));
        CB_CHECK(device < n, CB_ERR_ARG, "device %d out of range (%d visible)", device, n);
        CB_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        CB_HIP(hipGetDeviceProperties(&prop, device));
        CB_CHECK(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0, CB_ERR_HIP,
                 "libcelebbasis_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
        c.stream = static_cast<hipStream_t>(hip_stream);
        // (a high-priority side stream was measured: 39.23 / 39.09 ms against 39.24 / 39.26 ms per fp32 step, 34.64 / 34.41 against 34.55 / 34.56 ms
        //  per bf16 step — no effect, plain stream kept; profiles/README.md round 3)
        {
            const char* e = getenv("CB_AUX_PRIORITY");          // experiment switch: 1 = side stream at the highest priority, -1 = lowest
            int least = 0, greatest = 0;
            CB_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            if (e && e[0] == '1') CB_HIP(hipStreamCreateWithPriority(&c.aux, hipStreamNonBlocking, greatest));
            else if (e && e[0] == '-') CB_HIP(hipStreamCreateWithPriority(&c.aux, hipStreamNonBlocking, least));
            else CB_HIP(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
        }
        CB_HIP(hipStreamCreateWithFlags(&c.aux2, hipStreamNonBlocking));
        CB_HIP(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
        CB_HIP(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
        CB_HIP(hipEventCreateWithFlags(&c.ev_tail, hipEventDisableTiming));
        CB_HIP(hipEventCreateWithFlags(&c.ev_tail_fork, hipEventDisableTiming));
    }
    *out = h.release();
    CB_END
}
int cb_ctx_destroy(cb_ctx* ctx) {
    CB_TRY
    delete ctx;
    CB_END
}
int cb_ctx_sync(cb_ctx* ctx) {
    CB_TRY
    require(ctx, "ctx");
    ctx->c.join_tail();          // a training step's deferred tail (side stream) is part of "everything enqueued on this context"
    if (!ctx->c.dry) CB_HIP(hipStreamSynchronize(ctx->c.stream));
    CB_END
}
int cb_ctx_stats(cb_ctx* ctx, double* gemm_flops, int64_t* launches, int64_t* arena_peak_bytes, int64_t* pool_bytes) {
    CB_TRY
    require(ctx, "ctx");
    if (gemm_flops) *gemm_flops = ctx->c.stats.gemm_flops;
    if (launches) *launches = ctx->c.stats.launches;
    if (arena_peak_bytes) *arena_peak_bytes = (int64_t)ctx->c.arena.peak;
    if (pool_bytes) *pool_bytes = (int64_t)ctx->c.pool.total;
    CB_END
}
int cb_ctx_reset_stats(cb_ctx* ctx) {
    CB_TRY
    require(ctx, "ctx");
    ctx->c.stats = Stats();
    ctx->c.arena.peak = 0;
    CB_END
}
int cb_ctx_set_warp_grid(cb_ctx* ctx, const float* host_grid, int S) {
    CB_TRY
    require(ctx, "ctx");
    Ctx& c = ctx->c;
    ++c.warp_gen;                         // captured step graphs made under the previous table are dropped at their next use
    if (!host_grid) {
        // the prefetch stream and a replaying step graph read the table too and the context does not know them: wait for the device
        if (c.warp_grid && !c.dry) { c.join_tail(); CB_HIP(hipDeviceSynchronize()); CB_HIP(hipFree(c.warp_grid)); }
        c.warp_grid = nullptr; c.warp_S = 0;
    } else {
        CB_CHECK(S > 1 && S <= 4096, CB_ERR_ARG, "warp grid: S = %d", S);
        // the table must be a plausible affine grid: finite, inside a few image widths
        const size_t n = (size_t)S * S * 2;
        for (size_t i = 0; i < n; i += (size_t)S * 2 - 1)
            CB_CHECK(host_grid[i] == host_grid[i] && host_grid[i] > -8.f && host_grid[i] < 8.f, CB_ERR_ARG,
                     "warp grid: element %zu = %g is not a normalised coordinate", i, (double)host_grid[i]);
        if (c.dry) {
            c.warp_grid = reinterpret_cast<float*>(0x200000); c.warp_S = S;
        } else {
            c.join_tail();
            CB_HIP(hipDeviceSynchronize());            // rare configuration call: every stream that may be sampling the old table
            if (c.warp_grid && c.warp_S != S) { CB_HIP(hipFree(c.warp_grid)); c.warp_grid = nullptr; }
            if (!c.warp_grid) CB_HIP(hipMalloc(&c.warp_grid, n * sizeof(float)));
            CB_HIP(hipMemcpy(c.warp_grid, host_grid, n * sizeof(float), hipMemcpyHostToDevice));
            c.warp_S = S;
        }
    }
    CB_END
}
int cb_ctx_set_debug_taps(cb_ctx* ctx, int enable) {
    CB_TRY
    require(ctx, "ctx");
    ctx->c.debug_taps = enable != 0;
    if (!enable) ctx->c.taps.clear();
    CB_END
}
int cb_ctx_tap(cb_ctx* ctx, const char* name, float* host_dst, int64_t max_elems, int64_t* numel) {
    CB_TRY
    require(ctx, "ctx"); require(name, "name");
    auto it = ctx->c.taps.find(name);
    CB_CHECK(it != ctx->c.taps.end(), CB_ERR_MISSING, "no tap named '%s'", name);
    if (numel) *numel = it->second.n;
    if (host_dst && !ctx->c.dry) {
        int64_t n = std::min<int64_t>(max_elems, it->second.n);
        ctx->c.join_tail();
        CB_HIP(hipStreamSynchronize(ctx->c.stream));
        if (it->second.dt == DT_BF16) {        // taps are always handed out as fp32
            std::vector<uint16_t> tmp((size_t)n);
            CB_HIP(hipMemcpy(tmp.data(), it->second.p, (size_t)n * 2, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)tmp[(size_t)i] << 16; memcpy(host_dst + i, &u, 4); }
        } else {
            CB_HIP(hipMemcpy(host_dst, it->second.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        }
    }
    CB_END
}

int cb_ctx_profile_begin(cb_ctx* ctx) {
    CB_TRY
    require(ctx, "ctx");
    CB_CHECK(!ctx->c.dry, CB_ERR_STATE, "profiling needs a real device context");
    ctx->c.join_tail();
    for (auto& r : ctx->c.prof.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    ctx->c.prof.recs.clear();
    ctx->c.prof.on = true;
    CB_END
}
int cb_ctx_profile_end(cb_ctx* ctx, char* json, int64_t cap) {
    CB_TRY
    require(ctx, "ctx"); require(json, "json");
    Profiler& P = ctx->c.prof;
    P.on = false;
    ctx->c.join_tail();
    CB_HIP(hipStreamSynchronize(ctx->c.stream));
    std::vector<double> ms(P.names.size(), 0.0), fl(P.names.size(), 0.0), by(P.names.size(), 0.0);
    std::vector<long> cnt(P.names.size(), 0);
    for (auto& r : P.recs) {
        float t = 0.f;
        CB_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
        ms[r.variant] += t; fl[r.variant] += r.flops; by[r.variant] += r.bytes; cnt[r.variant]++;
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    P.recs.clear();
    std::string out = "[";
    for (size_t i = 0; i < P.names.size(); ++i) {
        if (!cnt[i]) continue;
        char buf[320];
        snprintf(buf, sizeof buf, "%s{\"kernel\":\"%s\",\"launches\":%ld,\"flops\":%.6e,\"bytes\":%.6e,\"ms\":%.6f}",
                 out.size() > 1 ? "," : "", P.names[i].c_str(), cnt[i], fl[i], by[i], ms[i]);
        out += buf;
    }
    out += "]";
    CB_CHECK((int64_t)out.size() + 1 <= cap, CB_ERR_ARG, "profile_end: buffer too small (%zu needed)", out.size() + 1);
    memcpy(json, out.c_str(), out.size() + 1);
    CB_END
}

// ------------------------------------------------------------------ weight table
int cb_wtab_create(cb_wtab** out) {
    CB_TRY
    require(out, "out");
    *out = new cb_wtab();
    CB_END
}
int cb_wtab_add(cb_wtab* wt, const char* name, const float* data, int on_device, int ndim, const int64_t* dims) {
    CB_TRY
    require(wt, "wt"); require(name, "name");
    CB_CHECK(ndim >= 0 && ndim <= 8 && (ndim == 0 || dims), CB_ERR_ARG, "bad dims for '%s'", name);
    WEntry e; e.data = data; e.on_device = on_device;
    for (int i = 0; i < ndim; ++i) e.dims.push_back(dims[i]);
    wt->t.m[name] = e;
    CB_END
}
int cb_wtab_destroy(cb_wtab* wt) {
    CB_TRY
    delete wt;
    CB_END
}

// ------------------------------------------------------------------ diagnostics
int cb_debug_force_plan(int BM, int BN, int split) {
    CB_TRY
    CB_CHECK((BM == 0 && BN == 0) || ((BM == 64 || BM == 128) && (BN == 64 || BN == 128)) || (BM == 256 && (BN == 128 || BN == 256)) || (BM == 512 && BN == 128) ||
                 (BN == 80 && (BM == 16 || BM == 32 || BM == 64)), CB_ERR_ARG,
             "force_plan: tile %dx%d", BM, BN);
    PlanForce& f = plan_force();
    f.BM = BM; f.BN = BN; f.split = split > 0 ? split : 0; ++f.gen;
    CB_END
}

int cb_debug_glds_safe(int on) {
    CB_TRY
    plan_force().glds_safe = on ? 1 : 0; ++plan_force().gen;
    CB_END
}

int cb_debug_splitk_inlaunch(int on) {
    CB_TRY
    plan_force().splitk_inlaunch = on ? 1 : 0; ++plan_force().gen;
    CB_END
}

// ------------------------------------------------------------------ op level
int cb_gemm(cb_ctx* ctx, const cb_gemm_desc* d) {
    CB_TRY
    require(ctx, "ctx"); require(d, "desc");
    CB_CHECK(d->struct_size == (int)sizeof(cb_gemm_desc), CB_ERR_ARG, "gemm: desc.struct_size %d != %d — caller built against another "
             "header (ABI %d)", d->struct_size, (int)sizeof(cb_gemm_desc), CB_ABI_VERSION);
    CB_CHECK(d->dtype == CB_DTYPE_F32 || d->dtype == CB_DTYPE_BF16, CB_ERR_ARG, "gemm: unknown desc.dtype %d", d->dtype);
    GemmArgs a;
    a.dt = (DType)d->dtype;
    a.A = d->A; a.B = d->B; a.C = d->C; a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.a_kind = d->a_layout == CB_LD_COL ? LD_COL : LD_ROW;
    a.b_kind = d->b_layout == CB_LD_COL ? LD_COL : LD_ROW;
    a.batch = d->batch > 0 ? d->batch : 1; a.nh = 1;
    a.sAb = d->strideA; a.sBb = d->strideB; a.sCb = d->strideC;
    a.bias = d->bias; a.R = d->R; a.ldr = d->ldr; a.sRb = d->strideR;
    a.alpha = d->alpha; a.beta = d->beta;
    Ctx* c = &ctx->c;
    plan_and_run(c, [&]() {
        Launch L = c->launch();
        const bool bf = a.dt == DT_BF16;
        size_t wsb = bf ? gemm_bf16_ws_bytes(a) : gemm_ws_bytes(a);
        float* ws = wsb ? static_cast<float*>(c->arena.alloc(wsb)) : nullptr;
        if (bf) gemm_bf16(L, a, ws, wsb); else gemm(L, a, ws, wsb);
    });
    CB_END
}
static int pad8c(DType dt, int c) { return dt == DT_BF16 ? (c + 7) / 8 * 8 : c; }
int cb_pack_conv_weight(cb_ctx* ctx, const float* w, void* out, int O, int I, int ks, int dgrad) {
    CB_TRY
    require(ctx, "ctx");
    Launch L = ctx->c.launch();
    const DType dt = ctx->c.dtype;
    if (dgrad == 2) { CB_CHECK(ks == 3, CB_ERR_ARG, "pack: the up-conv gradient pack is defined for 3x3 kernels"); k_pack_conv_up_dgrad(L, dt, out, w, O, I, pad8c(dt, O)); }
    else if (dgrad) k_pack_conv_dgrad(L, dt, out, w, O, I, ks, pad8c(dt, O)); else k_pack_conv(L, dt, out, w, O, I, ks, pad8c(dt, I));
    CB_END
}
int cb_conv2d_fwd(cb_ctx* ctx, const void* x, const void* w_packed, const float* bias, const void* rowbias,
                  const void* residual, void* y, int N, int H, int W, int Cin, int Cout, int ks, int mode) {
    CB_TRY
    require(ctx, "ctx");
    Ctx* c = &ctx->c;
    const DType dt = c->dtype;
    Conv cv; cv.w = const_cast<void*>(w_packed); cv.b = const_cast<float*>(bias); cv.cin = Cin; cv.cout = Cout; cv.ks = ks;
    cv.dt = dt; cv.cin_p = pad8c(dt, Cin); cv.cout_p = pad8c(dt, Cout);
    plan_and_run(c, [&]() {
        Graph G(c);
        Ten* xt = G.wrap(x, N, H, W, Cin, false, dt);
        int Ho, Wo; ConvGeom g;
        conv_out_dims(cv, (ConvMode)mode, H, W, &Ho, &Wo, &g);
        Ten* rb = rowbias ? G.wrap(rowbias, N, 1, 1, Cout, false, dt) : nullptr;
        Ten* rs = residual ? G.wrap(residual, N, Ho, Wo, Cout, false, dt) : nullptr;
        Ten* o = G.conv(xt, cv, (ConvMode)mode, rb, rs);
        k_axpby(G.L, dt, y, o->p, o->numel(), 1.f, 0.f);
    });
    CB_END
}
int cb_conv2d_dgrad(cb_ctx* ctx, const void* dy, const void* wd, void* dx, int N, int H, int W, int Cin, int Cout, int ks,
                    int mode, float beta) {
    CB_TRY
    require(ctx, "ctx");
    Ctx* c = &ctx->c;
    Conv cv; cv.wd = const_cast<void*>(wd); cv.cin = Cin; cv.cout = Cout; cv.ks = ks;
    if (mode == CB_CONV_UP_DGRAD4) { cv.wu = cv.wd; mode = CB_CONV_UP; }       // the dgrad = 2 pack: one 4x4 stride-2 conv over dy
    cv.dt = c->dtype; cv.cin_p = pad8c(cv.dt, Cin); cv.cout_p = pad8c(cv.dt, Cout);
    plan_and_run(c, [&]() {
        Graph G(c);
        conv_dgrad(&G, dy, cv, (ConvMode)mode, N, H, W, dx, beta);
    });
    CB_END
}
int cb_groupnorm_silu_fwd(cb_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, float* stats, int N,
                          int HW, int C, float eps, int silu) {
    CB_TRY
    require(ctx, "ctx");
    Ctx* c = &ctx->c;
    plan_and_run(c, [&]() {
        float* ws = c->arena.allocf(gn_ws_floats(N, HW, C, 32) + (size_t)N * 64);
        k_groupnorm_fwd(c->launch(), c->dtype, y, x, gamma, beta, stats, ws, N, HW, C, 32, eps, silu);
    });
    CB_END
}
int cb_groupnorm_silu_bwd(cb_ctx* ctx, const void* x, const void* dy, const float* gamma, const float* beta,
                          const float* stats, void* dx, int N, int HW, int C, int silu, float acc_beta) {
    CB_TRY
    require(ctx, "ctx");
    Ctx* c = &ctx->c;
    plan_and_run(c, [&]() {
        float* ws = c->arena.allocf(gn_ws_floats(N, HW, C, 32) + (size_t)N * 64);
        k_groupnorm_bwd(c->launch(), c->dtype, dx, x, dy, gamma, beta, stats, ws, N, HW, C, 32, silu, acc_beta);
    });
    CB_END
}
int cb_layernorm_fwd(cb_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, float* stats, int64_t rows,
                     int C, float eps) {
    CB_TRY
    require(ctx, "ctx");
    k_layernorm_fwd(ctx->c.launch(), ctx->c.dtype, y, x, gamma, beta, stats, rows, C, eps);
    CB_END
}
int cb_layernorm_bwd(cb_ctx* ctx, const void* x, const void* dy, const float* gamma, const float* stats, void* dx,
                     int64_t rows, int C, float acc_beta) {
    CB_TRY
    require(ctx, "ctx");
    k_layernorm_bwd(ctx->c.launch(), ctx->c.dtype, dx, x, dy, gamma, stats, rows, C, acc_beta);
    CB_END
}
int cb_geglu_fwd(cb_ctx* ctx, const void* x, void* y, int64_t M, int D) {
    CB_TRY
    require(ctx, "ctx");
    k_geglu_fwd(ctx->c.launch(), ctx->c.dtype, y, x, M, D);
    CB_END
}
int cb_geglu_bwd(cb_ctx* ctx, const void* x, const void* dy, void* dx, int64_t M, int D) {
    CB_TRY
    require(ctx, "ctx");
    k_geglu_bwd(ctx->c.launch(), ctx->c.dtype, dx, x, dy, M, D);
    CB_END
}
int cb_attention(cb_ctx* ctx, const cb_attention_desc* d) {
    CB_TRY
    require(ctx, "ctx"); require(d, "desc");
    CB_CHECK(d->struct_size == (int)sizeof(cb_attention_desc), CB_ERR_ARG, "attention: desc.struct_size %d != %d — caller built against "
             "another header (ABI %d)", d->struct_size, (int)sizeof(cb_attention_desc), CB_ABI_VERSION);
    Ctx* c = &ctx->c;
    const bool bwd = d->dout != nullptr;
    if (bwd) CB_CHECK(d->dqbuf && d->dkvbuf, CB_ERR_ARG, "attention: backward needs dqbuf and dkvbuf");
    CB_CHECK(d->dtype == CB_DTYPE_F32 || d->dtype == CB_DTYPE_BF16, CB_ERR_ARG, "attention: unknown desc.dtype %d", d->dtype);
    const DType dt = (DType)d->dtype;
    plan_and_run(c, [&]() {
        Graph G(c);
        Ten* qt = G.wrap(d->qbuf, d->B, d->Nq, 1, (int)d->ldq, bwd, dt);
        Ten* kt = (d->kvbuf == d->qbuf) ? qt : G.wrap(d->kvbuf, d->B, d->Nk, 1, (int)d->ldkv, bwd, dt);
        Ten* o = G.attention(qt, d->q_off, kt, d->k_off, d->v_off, d->C, d->heads, d->scale, d->causal != 0);
        k_axpby(G.L, dt, d->out, o->p, o->numel(), 1.f, 0.f);
        if (bwd) {
            float beta; void* go = G.grad(o, &beta);
            k_axpby(G.L, dt, go, d->dout, o->numel(), 1.f, 0.f);
            qt->g = d->dqbuf;
            if (kt != qt) kt->g = d->dkvbuf;
            G.backward();
        }
    });
    CB_END
}
int cb_timestep_embedding(cb_ctx* ctx, const int64_t* t, void* out, int B, int dim) {
    CB_TRY
    require(ctx, "ctx");
    k_timestep_embedding(ctx->c.launch(), ctx->c.dtype, out, t, B, dim);
    CB_END
}
int cb_q_sample(cb_ctx* ctx, const void* x0, const float* eps, const int64_t* t, const float* sa, const float* sb, void* xt, int B,
                int C, int HW) {
    CB_TRY
    require(ctx, "ctx");
    k_qsample(ctx->c.launch(), ctx->c.dtype, xt, x0, eps, t, sa, sb, B, C, HW);
    CB_END
}
int cb_mse_loss(cb_ctx* ctx, const void* pred, const float* target, float* loss, void* dpred, int B, int C, int HW) {
    CB_TRY
    require(ctx, "ctx");
    k_mse_loss(ctx->c.launch(), ctx->c.dtype, loss, dpred, pred, target, B, C, HW);
    CB_END
}
int cb_face_warp_resize(cb_ctx* ctx, const float* faces, float* warped, float* out, int B, int S, int nface, int out_size) {
    CB_TRY
    require(ctx, "ctx");
    static const float th[6] = {1.07695457f, -0.03625215f, -1.56352194f / 512.f, 0.03625215f, 1.07695457f, -5.32134629f / 512.f};
    Launch L = ctx->c.launch();
    k_affine_warp(L, warped, faces, B, S, nface, th, ctx->c.warp_grid_for(S));
    k_resize_bilinear_ac(L, out, warped, nface * B, S, S, out_size, out_size, 3);
    CB_END
}
// meta_net.py:262-289 after the FR backbone: F.normalize(v) -> EqualLinear + LeakyReLU(0.2) -> 'b (e h d) -> b e h d' ->
// L2-normalise over d -> einsum with the celeb basis + mean. feat [N,I] (raw FR features), W [es*K, I], b [es*K],
// basis [es, 1+K, Wd] -> coef [N, es*K] (unit rows per e), z [N, es*Wd]. The inference-time `test_mode: image` path.
int cb_celeb_head_forward(cb_ctx* ctx, const float* feat, const float* W, const float* b, const float* basis, int N, int I, int es,
                          int K, int Wd, float* coef, float* z) {
    CB_TRY
    require(ctx, "ctx"); require(feat, "feat"); require(W, "W"); require(b, "b"); require(basis, "basis"); require(coef, "coef");
    require(z, "z");
    Ctx* c = &ctx->c;
    plan_and_run(c, [&]() {
        Graph G(c);
        const int O = es * K;
        Ten* v = G.make(N, 1, 1, I, false, DT_F32);
        k_l2norm_fwd(G.L, v->pf(), nullptr, feat, N, I);
        Lin head; head.w = const_cast<float*>(W); head.b = const_cast<float*>(b); head.in = I; head.out = O;
        Ten* pre = G.linear(v, head);
        Ten* act = G.make(N, 1, 1, O, false, DT_F32);
        k_leakyrelu_fwd(G.L, act->pf(), pre->pf(), act->numel(), 0.2f);
        k_l2norm_fwd(G.L, coef, nullptr, act->pf(), (long)N * es, K);
        k_basis_fwd(G.L, z, coef, basis, N, es, K, Wd);
    });
    CB_END
}
int cb_face_augment(cb_ctx* ctx, const float* in, const float* params, float* out, int B, int S) {
    CB_TRY
    require(ctx, "ctx"); require(in, "in"); require(params, "params"); require(out, "out");
    CB_CHECK(in != out, CB_ERR_ARG, "face augment: in-place is not supported (the flip reads mirrored pixels)");
    Ctx* c = &ctx->c;
    plan_and_run(c, [&]() {
        float* ws = c->arena.allocf(face_augment_ws_floats(B));
        k_face_augment(c->launch(), out, in, params, ws, B, S);
    });
    CB_END
}
int cb_paste_on_canvas(cb_ctx* ctx, const float* img, const int32_t* geom, float* out, int B, int S, int C) {
    CB_TRY
    require(ctx, "ctx"); require(img, "img"); require(geom, "geom"); require(out, "out");
    k_paste_bg(ctx->c.launch(), out, img, geom, B, S, C);
    CB_END
}
int cb_row_surgery(cb_ctx* ctx, const float* embedded, const float* z, const int64_t* tokens, float* out, int32_t* map, int B,
                   int T, int C, int reps, int64_t placeholder) {
    CB_TRY
    require(ctx, "ctx");
    k_row_surgery_fwd(ctx->c.launch(), out, embedded, z, tokens, map, B, T, C, reps, placeholder);
    CB_END
}
int cb_row_surgery_multi(cb_ctx* ctx, const float* embedded, const float* z, const int64_t* tokens, float* out, int32_t* map, int B,
                         int T, int C, int reps, const int64_t* placeholders, int nph) {
    CB_TRY
    require(ctx, "ctx"); require(placeholders, "placeholders");
    k_row_surgery_multi(ctx->c.launch(), out, embedded, z, tokens, map, B, T, C, reps, placeholders, nph);
    CB_END
}
int cb_adamw_step(cb_ctx* ctx, float* p, float* m, float* v, const float* g, int64_t n, int* step_dev, float lr, float b1,
                  float b2, float eps, float wd) {
    CB_TRY
    require(ctx, "ctx");
    Launch L = ctx->c.launch();
    k_step_inc(L, step_dev);
    k_adamw(L, p, m, v, g, n, step_dev, lr, b1, b2, eps, wd);
    CB_END
}
int cb_nchw_to_nhwc(cb_ctx* ctx, const float* src, void* dst, int N, int C, int H, int W) {
    CB_TRY
    require(ctx, "ctx");
    k_nchw_to_nhwc(ctx->c.launch(), ctx->c.dtype, dst, src, N, C, H, W);
    CB_END
}
int cb_nhwc_to_nchw(cb_ctx* ctx, const void* src, float* dst, int N, int C, int H, int W) {
    CB_TRY
    require(ctx, "ctx");
    k_nhwc_to_nchw(ctx->c.launch(), ctx->c.dtype, dst, src, N, C, H, W);
    CB_END
}

// ------------------------------------------------------------------ graph level: UNet
static UNetCfg to_cfg(const cb_unet_cfg* c) {
    UNetCfg u;
    u.in_channels = c->in_channels; u.out_channels = c->out_channels; u.model_channels = c->model_channels;
    u.num_res_blocks = c->num_res_blocks; u.n_mult = c->n_mult; u.n_attn = c->n_attn;
    CB_CHECK(c->n_mult >= 1 && c->n_mult <= 8 && c->n_attn >= 0 && c->n_attn <= 8, CB_ERR_ARG, "unet cfg: bad list sizes");
    for (int i = 0; i < 8; ++i) { u.channel_mult[i] = c->channel_mult[i]; u.attention_resolutions[i] = c->attention_resolutions[i]; }
    u.num_heads = c->num_heads; u.context_dim = c->context_dim;
    CB_CHECK(u.model_channels % 32 == 0 && u.num_heads > 0, CB_ERR_ARG, "unet cfg: model_channels must be a multiple of 32");
    return u;
}
static VaeCfg to_cfg(const cb_vae_cfg* c) {
    VaeCfg v;
    v.ch = c->ch; v.n_mult = c->n_mult; v.num_res_blocks = c->num_res_blocks; v.in_channels = c->in_channels; v.z_channels = c->z_channels;
    CB_CHECK(c->n_mult >= 1 && c->n_mult <= 8, CB_ERR_ARG, "vae cfg: bad n_mult");
    for (int i = 0; i < 8; ++i) v.ch_mult[i] = c->ch_mult[i];
    return v;
}
static ClipCfg to_cfg(const cb_clip_cfg* c) {
    ClipCfg k; k.vocab = c->vocab; k.width = c->width; k.layers = c->layers; k.heads = c->heads; k.ctx_len = c->ctx_len; k.mlp = c->mlp;
    return k;
}

int cb_unet_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_unet_cfg* cfg, cb_unet** out) {
    CB_TRY
    require(ctx, "ctx"); require(wt, "wt"); require(cfg, "cfg"); require(out, "out");
    std::unique_ptr<cb_unet> u(new cb_unet());
    u->ctx = &ctx->c;
    unet_load(&ctx->c, wt->t, prefix ? prefix : "", to_cfg(cfg), &u->U);
    *out = u.release();
    CB_END
}
int cb_unet_destroy(cb_unet* u) {
    CB_TRY
    delete u;
    CB_END
}
int cb_unet_forward(cb_unet* u, const float* x_nchw, const int64_t* t, const float* context, int B, int H, int W, int L,
                    float* eps_nchw, int need_grad) {
    CB_TRY
    require(u, "unet");
    Ctx* c = u->ctx;
    u->B = B; u->H = H; u->W = W; u->L = L;
    plan_and_run(c, [&]() {
        u->G.reset(new Graph(c));
        Graph& G = *u->G;
        const DType dt = u->U.dt;
        Ten* x = G.make(B, H, W, u->U.cfg.in_channels, false, dt);
        k_nchw_to_nhwc(G.L, dt, x->p, x_nchw, B, u->U.cfg.in_channels, H, W);
        Ten* cx = G.wrap(context, B, L, 1, u->U.cfg.context_dim, need_grad != 0, DT_F32);      // the public context is fp32
        Ten* o = unet_forward(G, u->U, x, t, G.cast(cx, dt));
        k_nhwc_to_nchw(G.L, dt, eps_nchw, o->p, B, u->U.cfg.out_channels, H, W);
        u->out = o; u->context = cx;
        if (c->dry && need_grad) { float beta; G.grad(o, &beta); G.backward(); }   // size the arena for the backward as well
    });
    CB_END
}
int cb_unet_backward_context(cb_unet* u, const float* deps_nchw, float* dcontext) {
    CB_TRY
    require(u, "unet");
    CB_CHECK(u->G && u->out && u->context && u->context->rg, CB_ERR_STATE, "unet backward: call cb_unet_forward(need_grad=1) first");
    Graph& G = *u->G;
    float beta; void* go = G.grad(u->out, &beta);
    k_nchw_to_nhwc(G.L, u->out->dt, go, deps_nchw, u->B, u->U.cfg.out_channels, u->H, u->W);
    G.backward();
    if (!u->ctx->dry) CB_CHECK(u->context->ginit, CB_ERR_STATE, "unet backward: no gradient reached the context");
    if (u->context->g) k_axpby(G.L, dcontext, static_cast<const float*>(u->context->g), u->context->numel(), 1.f, 0.f);
    CB_END
}

// DDIM (ddim.py:113-204). The scalar coefficients follow the reference's fp32 tensor arithmetic: a_t.sqrt(),
// (1 - a_prev - sigma_t**2).sqrt(), a_prev.sqrt() on torch.full(..., float32) tensors.
namespace {
struct DdimCoef { float sqrt_at, sqrt_aprev, dir; };
DdimCoef ddim_coef(float a_t, float a_prev, float sigma) {
    DdimCoef c;
    c.sqrt_at = sqrtf(a_t);
    c.sqrt_aprev = sqrtf(a_prev);
    float s2 = sigma * sigma;
    float one_m = 1.0f - a_prev;
    c.dir = sqrtf(one_m - s2);
    return c;
}
}  // namespace
int cb_ddim_update(cb_ctx* ctx, float* x_nchw, const float* eps_nchw, const float* noise_nchw, int n, int C, int HW, int cfg_rows,
                   float guidance_scale, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, float* pred_x0_nchw) {
    CB_TRY
    require(ctx, "ctx"); require(x_nchw, "x"); require(eps_nchw, "eps");
    CB_CHECK(sigma_t == 0.f || noise_nchw, CB_ERR_ARG, "ddim update with sigma != 0 needs a noise tensor");
    Launch L = ctx->c.launch();
    DdimCoef k = ddim_coef(a_t, a_prev, sigma_t);
    // elementwise, so the NCHW buffers can be walked linearly (C=1 decode keeps the noise index identical)
    k_ddim_update(L, DT_F32, x_nchw, pred_x0_nchw, eps_nchw, noise_nchw, n, 1, C * HW, cfg_rows != 0, guidance_scale, sqrt_one_minus_at,
                  k.sqrt_at, k.sqrt_aprev, k.dir, sigma_t);
    CB_END
}
int cb_unet_ddim_sample(cb_unet* u, float* x_nchw, const float* cond, const float* uncond, int n, int H, int W, int L, int S,
                        const int64_t* timesteps, const float* alphas, const float* alphas_prev, const float* sigmas,
                        const float* sqrt_one_minus_alphas, float guidance_scale, const float* noise_nchw, float* pred_x0_nchw) {
    CB_TRY
    require(u, "unet"); require(timesteps, "timesteps");
    if (!u->ctx->dry) { require(x_nchw, "x"); require(cond, "cond"); }
    require(alphas, "alphas"); require(alphas_prev, "alphas_prev"); require(sigmas, "sigmas"); require(sqrt_one_minus_alphas, "sqrt_one_minus_alphas");
    CB_CHECK(n > 0 && S > 0, CB_ERR_ARG, "ddim: n and S must be positive");
    for (int i = 0; i < S; ++i) CB_CHECK(sigmas[i] == 0.f || noise_nchw, CB_ERR_ARG, "ddim: eta != 0 needs a noise tensor [S][n,C,H,W]");
    Ctx* c = u->ctx;
    const bool cfg = (uncond != nullptr || c->dry) && guidance_scale != 1.0f;   // ddim.py:172 (a dry context plans the guided shape)
    const int Cx = u->U.cfg.in_channels, D = u->U.cfg.context_dim;
    CB_CHECK(Cx == u->U.cfg.out_channels, CB_ERR_ARG, "ddim: eps-parameterisation needs out_channels == in_channels");
    const int Bn = cfg ? 2 * n : n;
    const long per = (long)n * H * W * Cx;
    // loop-carried buffers live at the bottom of the step arena; every iteration rewinds the arena to `mark`
    // the sampler state x (and pred_x0) stays fp32 across the steps; the UNet input / context / output are in the UNet's dtype
    const DType dt = u->U.dt;
    Ten X, XIN, CIN, P0;
    Ten *x = &X, *xin = &XIN, *cin = &CIN, *p0 = &P0;
    auto shape = [](Ten* t, int n_, int h_, int w_, int c_, DType d_) { t->n = n_; t->h = h_; t->w = w_; t->c = c_; t->dt = d_; };
    shape(x, n, H, W, Cx, DT_F32); shape(xin, Bn, H, W, Cx, dt); shape(p0, n, H, W, Cx, DT_F32); shape(cin, Bn, L, 1, D, dt);
    int64_t* tin = nullptr;
    size_t mark = 0;
    auto iteration = [&](int i) {
        const int index = S - 1 - i;                                       // ddim.py:141
        Graph G(c);
        k_fill_i64(G.L, tin, Bn, timesteps[index]);
        k_convert(G.L, xin->p, dt, Cx, x->p, DT_F32, Cx, per / Cx);
        if (cfg) k_convert(G.L, xin->at(per), dt, Cx, x->p, DT_F32, Cx, per / Cx);
        Ten* e = unet_forward(G, u->U, xin, tin, cin);
        DdimCoef k = ddim_coef(alphas[index], alphas_prev[index], sigmas[index]);
        const float* nz = (noise_nchw && sigmas[index] != 0.f) ? noise_nchw + (long)i * per : nullptr;
        k_ddim_update(G.L, dt, x->pf(), (pred_x0_nchw && i == S - 1) ? p0->pf() : nullptr, e->p, nz, n, Cx, H * W, cfg, guidance_scale,
                      sqrt_one_minus_alphas[index], k.sqrt_at, k.sqrt_aprev, k.dir, sigmas[index]);
    };
    plan_and_run(c, [&]() {
        Launch L0 = c->launch();
        x->p = c->arena.alloc(x->bytes()); xin->p = c->arena.alloc(xin->bytes());
        p0->p = c->arena.alloc(p0->bytes()); cin->p = c->arena.alloc(cin->bytes());
        tin = static_cast<int64_t*>(c->arena.alloc(sizeof(int64_t) * (size_t)Bn));
        k_nchw_to_nhwc(L0, DT_F32, x->p, x_nchw, n, Cx, H, W);
        const long cn = (long)n * L * D;
        if (cfg) { k_convert(L0, cin->p, dt, D, uncond, DT_F32, D, cn / D); k_convert(L0, cin->at(cn), dt, D, cond, DT_F32, D, cn / D); }   // cat([uc, c])
        else k_convert(L0, cin->p, dt, D, cond, DT_F32, D, cn / D);
        mark = c->arena.off;
        iteration(0);
    });
    for (int i = 1; i < S; ++i) { c->arena.off = mark; iteration(i); }
    Launch Lx = c->launch();
    k_nhwc_to_nchw(Lx, DT_F32, x_nchw, x->p, n, Cx, H, W);
    if (pred_x0_nchw) k_nhwc_to_nchw(Lx, DT_F32, pred_x0_nchw, p0->p, n, Cx, H, W);
    CB_END
}

// ------------------------------------------------------------------ VAE
int cb_vae_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_vae_cfg* cfg, cb_vae** out) {
    CB_TRY
    require(ctx, "ctx"); require(wt, "wt"); require(cfg, "cfg"); require(out, "out");
    std::unique_ptr<cb_vae> v(new cb_vae());
    v->ctx = &ctx->c;
    vae_load(&ctx->c, wt->t, prefix ? prefix : "", to_cfg(cfg), &v->V);
    *out = v.release();
    CB_END
}
int cb_vae_destroy(cb_vae* v) {
    CB_TRY
    delete v;
    CB_END
}
int cb_vae_encode(cb_vae* v, const float* image, const float* post_noise, int B, int H, int W, float scale, float* moments,
                  float* z_nchw) {
    CB_TRY
    require(v, "vae");
    Ctx* c = v->ctx;
    plan_and_run(c, [&]() {
        Graph G(c);
        const DType dt = v->V.dt;
        Ten* img = G.cast(G.wrap(image, B, H, W, v->V.cfg.in_channels, false, DT_F32), dt);      // the dataset image is fp32 HWC
        Ten* mom = vae_encode_moments(G, v->V, img);
        const int zc = v->V.cfg.z_channels;
        if (moments) k_convert(G.L, moments, DT_F32, mom->c, mom->p, dt, mom->c, mom->rows());
        if (z_nchw) {
            Ten* z = G.make(B, mom->h, mom->w, zc, false, dt);
            k_vae_sample(G.L, dt, z->p, mom->p, post_noise, B, mom->h * mom->w, zc, scale);
            k_nhwc_to_nchw(G.L, dt, z_nchw, z->p, B, zc, mom->h, mom->w);
        }
    });
    CB_END
}

int cb_vae_decoder_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_vae_cfg* cfg, cb_vae_dec** out) {
    CB_TRY
    require(ctx, "ctx"); require(wt, "wt"); require(cfg, "cfg"); require(out, "out");
    std::unique_ptr<cb_vae_dec> v(new cb_vae_dec());
    v->ctx = &ctx->c;
    vae_dec_load(&ctx->c, wt->t, prefix ? prefix : "", to_cfg(cfg), &v->V);
    *out = v.release();
    CB_END
}
int cb_vae_decoder_destroy(cb_vae_dec* v) {
    CB_TRY
    delete v;
    CB_END
}
int cb_vae_decode(cb_vae_dec* v, const float* z_nchw, int B, int h, int w, float scale_factor, int to01, float* image_nchw) {
    CB_TRY
    require(v, "vae decoder");
    if (!v->ctx->dry) { require(z_nchw, "z"); require(image_nchw, "image"); }
    CB_CHECK(scale_factor != 0.f, CB_ERR_ARG, "scale_factor must not be 0");
    Ctx* c = v->ctx;
    plan_and_run(c, [&]() {
        Graph G(c);
        const int zc = v->V.cfg.z_channels;
        Ten* z32 = G.make(B, h, w, zc, false, DT_F32);
        k_nchw_to_nhwc(G.L, DT_F32, z32->p, z_nchw, B, zc, h, w);
        k_axpby(G.L, z32->pf(), z32->pf(), z32->numel(), 1.f / scale_factor, 0.f);        // ddpm.py:772 z = 1/scale_factor * z
        Ten* img = vae_decode(G, v->V, G.cast(z32, v->V.dt));
        k_image_out(G.L, img->dt, image_nchw, img->p, B, img->c, img->h * img->w, to01);
    });
    CB_END
}

// ------------------------------------------------------------------ CLIP
int cb_clip_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_clip_cfg* cfg, cb_clip** out) {
    CB_TRY
    require(ctx, "ctx"); require(wt, "wt"); require(cfg, "cfg"); require(out, "out");
    std::unique_ptr<cb_clip> k(new cb_clip());
    k->ctx = &ctx->c;
    clip_load(&ctx->c, wt->t, prefix ? prefix : "", to_cfg(cfg), &k->C);
    *out = k.release();
    CB_END
}
int cb_clip_destroy(cb_clip* c) {
    CB_TRY
    delete c;
    CB_END
}
int cb_clip_embed_tokens(cb_clip* c, const int64_t* tokens, int B, int L, float* embedded) {
    CB_TRY
    require(c, "clip");
    k_gather_rows(c->ctx->launch(), embedded, c->C.tok_emb, tokens, (long)B * L, c->C.cfg.width, c->C.cfg.vocab);
    CB_END
}
int cb_clip_forward(cb_clip* k, const float* embedded, int B, int L, float* hidden, int need_grad) {
    CB_TRY
    require(k, "clip");
    Ctx* c = k->ctx;
    plan_and_run(c, [&]() {
        k->G.reset(new Graph(c));
        Graph& G = *k->G;
        Ten* e = G.wrap(embedded, B, L, 1, k->C.cfg.width, need_grad != 0, DT_F32);
        Ten* o = clip_encode(G, k->C, e);
        k_convert(G.L, hidden, DT_F32, o->c, o->p, o->dt, o->c, o->rows());
        k->out = o; k->in = e;
        if (c->dry && need_grad) { float beta; G.grad(o, &beta); G.backward(); }
    });
    CB_END
}
int cb_clip_backward_embeds(cb_clip* k, const float* dhidden, float* dembedded) {
    CB_TRY
    require(k, "clip");
    CB_CHECK(k->G && k->out && k->in && k->in->rg, CB_ERR_STATE, "clip backward: call cb_clip_forward(need_grad=1) first");
    Graph& G = *k->G;
    float beta; void* go = G.grad(k->out, &beta);
    k_convert(G.L, go, k->out->dt, k->out->c, dhidden, DT_F32, k->out->c, k->out->rows());
    G.backward();
    if (k->in->g) k_axpby(G.L, dembedded, static_cast<const float*>(k->in->g), k->in->numel(), 1.f, 0.f);
    CB_END
}

// ------------------------------------------------------------------ FR-net
int cb_frnet_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const int* layers4, int width, int in_size, cb_frnet** out) {
    CB_TRY
    require(ctx, "ctx"); require(wt, "wt"); require(layers4, "layers"); require(out, "out");
    CB_CHECK(in_size % 16 == 0 && width > 0, CB_ERR_ARG, "frnet: in_size must be a multiple of 16");
    std::unique_ptr<cb_frnet> f(new cb_frnet());
    f->ctx = &ctx->c;
    frnet_load(&ctx->c, wt->t, prefix ? prefix : "", &f->F, layers4, width, in_size);
    *out = f.release();
    CB_END
}
int cb_frnet_destroy(cb_frnet* f) {
    CB_TRY
    delete f;
    CB_END
}
int cb_frnet_forward(cb_frnet* f, const float* img, int N, float* features) {
    CB_TRY
    require(f, "frnet");
    Ctx* c = f->ctx;
    CB_CHECK(N >= 1 && N <= 16, CB_ERR_ARG, "frnet: N=%d must be in 1..16", N);
    plan_and_run(c, [&]() {
        Graph G(c);
        Ten* x = G.wrap(img, N, f->F.in_size, f->F.in_size, 3, false, DT_F32);
        Ten* o = frnet_forward(G, f->F, x);
        k_axpby(G.L, features, o->pf(), o->numel(), 1.f, 0.f);
    });
    CB_END
}

// ------------------------------------------------------------------ trainer
int cb_trainer_cfg_default(cb_trainer_cfg* cfg) {
    CB_TRY
    require(cfg, "cfg");
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int)sizeof(*cfg);
    cb_unet_cfg& u = cfg->unet;     // aigc_id.yaml:39-54
    u.in_channels = 4; u.out_channels = 4; u.model_channels = 320; u.num_res_blocks = 2;
    u.channel_mult[0] = 1; u.channel_mult[1] = 2; u.channel_mult[2] = 4; u.channel_mult[3] = 4; u.n_mult = 4;
    u.attention_resolutions[0] = 4; u.attention_resolutions[1] = 2; u.attention_resolutions[2] = 1; u.n_attn = 3;
    u.num_heads = 8; u.context_dim = 768;
    cb_vae_cfg& v = cfg->vae;       // aigc_id.yaml:56-77
    v.ch = 128; v.ch_mult[0] = 1; v.ch_mult[1] = 2; v.ch_mult[2] = 4; v.ch_mult[3] = 4; v.n_mult = 4; v.num_res_blocks = 2;
    v.in_channels = 3; v.z_channels = 4;
    cb_clip_cfg& k = cfg->clip;     // openai/clip-vit-large-patch14 text tower
    k.vocab = 49408; k.width = 768; k.layers = 12; k.heads = 12; k.ctx_len = 77; k.mlp = 3072;
    cfg->fr_layers[0] = 3; cfg->fr_layers[1] = 13; cfg->fr_layers[2] = 30; cfg->fr_layers[3] = 3; cfg->fr_width = 64; cfg->fr_in = 112;
    cfg->image_size = 512; cfg->es = 2; cfg->inner_dim = 512; cfg->max_ids = 10;
    cfg->momentum = 0.99f; cfg->scale_factor = 0.18215f; cfg->linear_start = 0.00085f; cfg->linear_end = 0.012f; cfg->timesteps = 1000;
    cfg->placeholder_token = 48136;
    cfg->adam_b1 = 0.9f; cfg->adam_b2 = 0.999f; cfg->adam_eps = 1e-8f; cfg->weight_decay = 1e-2f;
    cfg->dtype = CB_DTYPE_F32;
    CB_END
}
int cb_trainer_create(cb_ctx* ctx, const cb_wtab* wt, const cb_trainer_cfg* cfg, cb_trainer** out) {
    CB_TRY
    require(ctx, "ctx"); require(wt, "wt"); require(cfg, "cfg"); require(out, "out");
    CB_CHECK(cfg->struct_size == (int)sizeof(cb_trainer_cfg), CB_ERR_ARG, "trainer: cfg.struct_size %d != %d — start from "
             "cb_trainer_cfg_default() of this library (ABI %d)", cfg->struct_size, (int)sizeof(cb_trainer_cfg), CB_ABI_VERSION);
    TrainerCfg tc;
    tc.unet = to_cfg(&cfg->unet); tc.vae = to_cfg(&cfg->vae); tc.clip = to_cfg(&cfg->clip);
    for (int i = 0; i < 4; ++i) tc.fr_layers[i] = cfg->fr_layers[i];
    tc.fr_width = cfg->fr_width; tc.fr_in = cfg->fr_in;
    tc.image_size = cfg->image_size; tc.es = cfg->es; tc.inner_dim = cfg->inner_dim; tc.max_ids = cfg->max_ids;
    tc.momentum = cfg->momentum; tc.scale_factor = cfg->scale_factor; tc.linear_start = cfg->linear_start;
    tc.linear_end = cfg->linear_end; tc.timesteps = cfg->timesteps; tc.placeholder_token = cfg->placeholder_token;
    tc.adam_b1 = cfg->adam_b1; tc.adam_b2 = cfg->adam_b2; tc.adam_eps = cfg->adam_eps; tc.weight_decay = cfg->weight_decay;
    tc.use_rm_mlp = cfg->use_rm_mlp != 0;
    CB_CHECK(tc.image_size % 8 == 0 && tc.es >= 1 && tc.inner_dim >= 1 && tc.max_ids >= 1 && tc.timesteps >= 1, CB_ERR_ARG,
             "trainer cfg: bad sizes");
    CB_CHECK(cfg->dtype == (int)ctx->c.dtype, CB_ERR_ARG, "trainer cfg: dtype %d differs from the context's %d (create the context with "
             "cb_ctx_create_dtype)", cfg->dtype, (int)ctx->c.dtype);
    std::unique_ptr<cb_trainer> t(new cb_trainer());
    t->ctx = &ctx->c;
    trainer_load(&ctx->c, wt->t, tc, &t->T);
    *out = t.release();
    CB_END
}
// Views of the trainer's frozen networks for the sampling path: the returned objects alias the trainer's device weights
// (which live in the context's pool), so personalisation and txt2img of the same run hold one copy of SD v1-4.
int cb_trainer_share_unet(cb_trainer* t, cb_unet** out) {
    CB_TRY
    require(t, "trainer"); require(out, "out");
    std::unique_ptr<cb_unet> u(new cb_unet());
    u->ctx = t->ctx; u->U = t->T.unet;
    *out = u.release();
    CB_END
}
int cb_trainer_share_clip(cb_trainer* t, cb_clip** out) {
    CB_TRY
    require(t, "trainer"); require(out, "out");
    std::unique_ptr<cb_clip> c(new cb_clip());
    c->ctx = t->ctx; c->C = t->T.clip;
    *out = c.release();
    CB_END
}
int cb_trainer_destroy(cb_trainer* t) {
    CB_TRY
    if (t && t->ctx && !t->ctx->dry) { t->ctx->join_tail(); (void)hipStreamSynchronize(t->ctx->stream); }
    delete t;
    CB_END
}
int cb_trainer_step(cb_trainer* t, const cb_batch* b, float lr) {
    CB_TRY
    require(t, "trainer"); require(b, "batch");
    if (!t->ctx->dry) {
        require(b->image, "batch.image"); require(b->faces, "batch.faces"); require(b->ids, "batch.ids");
        require(b->tokens, "batch.tokens"); require(b->t, "batch.t"); require(b->noise, "batch.noise");
        require(b->post_noise, "batch.post_noise");
    }
    Batch bt;
    bt.image = b->image; bt.faces = b->faces; bt.ids = b->ids; bt.tokens = b->tokens; bt.t = b->t; bt.noise = b->noise;
    bt.post_noise = b->post_noise; bt.B = b->B;
    trainer_step(&t->T, bt, lr, false);
    CB_END
}

int cb_trainer_set_graph_replay(cb_trainer* t, int enable) {
    CB_TRY
    require(t, "trainer");
    trainer_set_graph_replay(&t->T, enable != 0);
    CB_END
}
int cb_trainer_set_tail_overlap(cb_trainer* t, int enable) {
    CB_TRY
    require(t, "trainer");
    t->ctx->join_tail();
    t->T.tail_overlap = enable != 0;
    CB_END
}
int cb_trainer_graph_stats(cb_trainer* t, int64_t* captures, int64_t* replays) {
    CB_TRY
    require(t, "trainer");
    if (captures) *captures = t->T.graph.captures;
    if (replays) *replays = t->T.graph.replays;
    CB_END
}
int cb_trainer_prefetch(cb_trainer* t, const cb_batch* b) {
    CB_TRY
    require(t, "trainer"); require(b, "batch");
    if (!t->ctx->dry) { require(b->image, "batch.image"); require(b->faces, "batch.faces"); require(b->post_noise, "batch.post_noise"); }
    Batch bt;
    bt.image = b->image; bt.faces = b->faces; bt.ids = b->ids; bt.tokens = b->tokens; bt.t = b->t; bt.noise = b->noise;
    bt.post_noise = b->post_noise; bt.B = b->B;
    trainer_prefetch(&t->T, bt);
    CB_END
}
int cb_trainer_set_prefetch_stream(cb_trainer* t, void* hip_stream) {
    CB_TRY
    require(t, "trainer");
    trainer_set_prefetch_stream(&t->T, static_cast<hipStream_t>(hip_stream));
    CB_END
}
int cb_trainer_prefetch_stream(cb_trainer* t, void** hip_stream) {
    CB_TRY
    require(t, "trainer"); require(hip_stream, "hip_stream");
    *hip_stream = static_cast<void*>(trainer_prefetch_stream(&t->T));
    CB_END
}

int cb_trainer_loss_to(cb_trainer* t, float* loss_dev) {
    CB_TRY
    require(t, "trainer");
    if (!t->ctx->dry) require(loss_dev, "loss_dev");
    // (the loss is written on the context stream during the UNet phase: no need to join a deferred tail for it)
    k_axpby(t->ctx->launch_nojoin(), loss_dev, t->T.loss, 1, 1.f, 0.f);
    CB_END
}

struct StateRef { void* p; long n; bool is_int; };
static StateRef find_state(cb_trainer* t, const char* name) {
    Trainer& T = t->T;
    const TrainerCfg& c = T.cfg;
    const long O = (long)c.es * c.inner_dim, I = T.fr.feat, w = c.clip.width;
    std::string s(name);
    if (s == "loss") return {T.loss, 1, false};
    if (s == "mlp.weight") return {T.W, O * I, false};
    if (s == "mlp.bias") return {T.b, O, false};
    if (s == "mlp.grad_weight") return {T.gW, O * I, false};
    if (s == "mlp.grad_bias") return {T.gb, O, false};
    if (s == "adam.m_weight") return {T.mW, O * I, false};
    if (s == "adam.v_weight") return {T.vW, O * I, false};
    if (s == "adam.m_bias") return {T.mb, O, false};
    if (s == "adam.v_bias") return {T.vb, O, false};
    if (s == "id_embeddings") return {T.id_emb, (long)c.max_ids * c.es * w, false};
    if (s == "id_coefficients") return {T.id_coef, (long)c.max_ids * O, false};
    if (s == "step") return {T.step, 1, true};
    if (s.rfind("rm.", 0) == 0 || s == "adam.m_coef" || s == "adam.v_coef") {
        CB_CHECK(T.coef != nullptr, CB_ERR_STATE, "trainer state '%s' exists only with use_rm_mlp", name);
        if (s == "rm.coef") return {T.coef, (long)c.max_ids * O, false};
        if (s == "rm.grad_coef") return {T.gC, (long)c.max_ids * O, false};
        if (s == "adam.m_coef") return {T.mC, (long)c.max_ids * O, false};
        if (s == "adam.v_coef") return {T.vC, (long)c.max_ids * O, false};
    }
    fail(CB_ERR_MISSING, "unknown trainer state '%s'", name);
}
int cb_trainer_get_state(cb_trainer* t, const char* name, float* host_dst, int64_t max_elems, int64_t* numel) {
    CB_TRY
    require(t, "trainer"); require(name, "name");
    StateRef r = find_state(t, name);
    if (numel) *numel = r.n;
    if (host_dst && !t->ctx->dry) {
        CB_CHECK(max_elems >= r.n, CB_ERR_ARG, "get_state '%s': buffer too small (%ld < %ld)", name, (long)max_elems, r.n);
        t->ctx->join_tail();
        CB_HIP(hipStreamSynchronize(t->ctx->stream));
        if (r.is_int) {
            int v = 0;
            CB_HIP(hipMemcpy(&v, r.p, sizeof(int), hipMemcpyDeviceToHost));
            host_dst[0] = (float)v;
        } else {
            CB_HIP(hipMemcpy(host_dst, r.p, (size_t)r.n * sizeof(float), hipMemcpyDeviceToHost));
        }
    }
    CB_END
}
int cb_trainer_set_state(cb_trainer* t, const char* name, const float* host_src, int64_t numel) {
    CB_TRY
    require(t, "trainer"); require(name, "name"); require(host_src, "host_src");
    StateRef r = find_state(t, name);
    CB_CHECK(numel == r.n, CB_ERR_SHAPE, "set_state '%s': %ld elements given, %ld expected", name, (long)numel, r.n);
    if (!t->ctx->dry) {
        t->ctx->join_tail();
        CB_HIP(hipStreamSynchronize(t->ctx->stream));
        if (r.is_int) {
            int v = (int)host_src[0];
            CB_HIP(hipMemcpy(r.p, &v, sizeof(int), hipMemcpyHostToDevice));
        } else {
            CB_HIP(hipMemcpy(r.p, host_src, (size_t)r.n * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    CB_END
}

}  // extern "C"
