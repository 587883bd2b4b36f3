// cb_engine.hip — memory pools, weight loading/packing and the autograd-tape ops of the native runtime.
#include "cb_engine.h"

namespace cb {

// ---------------------------------------------------------------- pools
void* Pool::alloc(size_t bytes) {
    bytes = (size_t)rup((long)bytes, 256);
    total += bytes;
    if (dry) return reinterpret_cast<void*>(0x200000 + (total - bytes));
    if (bytes > left) {
        size_t sz = bytes > slab_bytes ? bytes : slab_bytes;
        void* p = nullptr;
        CB_HIP(hipMalloc(&p, sz));
        slabs.push_back(p);
        cur = static_cast<char*>(p);
        left = sz;
    }
    void* r = cur;
    cur += bytes;
    left -= bytes;
    return r;
}
void Pool::release() {
    for (void* p : slabs) (void)hipFree(p);
    slabs.clear(); cur = nullptr; left = 0; total = 0;
}

constexpr size_t kArenaCounters = (size_t)1 << 20;      // 4 MB of split-K arrival counters per arena (a step takes ~1e5)
void Ctx::ensure_arena(size_t bytes) {
    if (dry) { arena.dry = true; arena.cnt.dry = true; return; }
    if (!arena.cnt.base) {
        void* c = nullptr;
        CB_HIP(hipMalloc(&c, kArenaCounters * sizeof(int)));
        CB_HIP(hipMemset(c, 0, kArenaCounters * sizeof(int)));
        arena.cnt.base = static_cast<int*>(c); arena.cnt.cap = kArenaCounters; arena.cnt.off = 0;
    }
    if (arena.cap >= bytes) return;
    join_tail();                                       // a deferred step tail still reads the arena that is about to be freed
    CB_HIP(hipStreamSynchronize(stream));
    if (arena.base) CB_HIP(hipFree(arena.base));
    arena.base = nullptr; arena.cap = 0;
    size_t want = bytes + bytes / 16 + ((size_t)64 << 20);
    void* p = nullptr;
    CB_HIP(hipMalloc(&p, want));
    arena.base = static_cast<char*>(p);
    arena.cap = want;
    arena.off = 0;
}
Ctx::~Ctx() {
    if (!dry) {
        (void)hipStreamSynchronize(stream);
        if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); }
        if (aux2) { (void)hipStreamSynchronize(aux2); (void)hipStreamDestroy(aux2); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_tail) (void)hipEventDestroy(ev_tail);
        if (ev_tail_fork) (void)hipEventDestroy(ev_tail_fork);
        if (arena.base) (void)hipFree(arena.base);
        if (arena.cnt.base) (void)hipFree(arena.cnt.base);
        if (warp_grid) (void)hipFree(warp_grid);
        pool.release();
    }
}

void plan_and_run(Ctx* ctx, const std::function<void()>& fn) {
    if (!ctx->dry) {
        Stats saved = ctx->stats;
        ctx->dry = true; ctx->arena.dry = true; ctx->arena.reset(); ctx->arena.peak = 0;
        try { fn(); } catch (...) { ctx->dry = false; ctx->arena.dry = false; ctx->stats = saved; throw; }
        size_t need = ctx->arena.peak;
        ctx->dry = false; ctx->arena.dry = false; ctx->stats = saved;
        ctx->ensure_arena(need);
    }
    ctx->arena.reset();
    fn();
}

static long numel_of(const std::vector<int64_t>& d) { long n = 1; for (auto v : d) n *= v; return n; }

float* Ctx::upload(const WTab& wt, const std::string& name, std::initializer_list<int64_t> expect) {
    const WEntry& e = wt.get(name);
    CB_CHECK(e.dtype == 0, CB_ERR_ARG, "weight '%s' must be float32", name.c_str());
    std::vector<int64_t> ex(expect);
    if (!ex.empty()) {
        bool ok = ex.size() == e.dims.size();
        for (size_t i = 0; ok && i < ex.size(); ++i) ok = (ex[i] < 0) || (ex[i] == e.dims[i]);
        if (!ok) {
            std::string got, want;
            for (auto v : e.dims) got += std::to_string(v) + ",";
            for (auto v : ex) want += std::to_string(v) + ",";
            fail(CB_ERR_SHAPE, "weight '%s' has shape [%s] but [%s] was expected", name.c_str(), got.c_str(), want.c_str());
        }
    }
    long n = numel_of(e.dims);
    float* d = pool.allocf((size_t)n);
    if (!dry) {
        CB_CHECK(e.data != nullptr, CB_ERR_ARG, "weight '%s' has no data", name.c_str());
        CB_HIP(hipMemcpyAsync(d, e.data, (size_t)n * sizeof(float), e.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              stream));
        if (!e.on_device) CB_HIP(hipStreamSynchronize(stream));  // the caller may free host memory afterwards
    }
    return d;
}

static int pad8(DType dt, int c) { return dt == DT_BF16 ? (int)rup(c, 8) : c; }

void* Ctx::operand(const float* src, long rows, int cols, int cols_p) {
    void* d = pool.alloc((size_t)rows * cols_p * dt_size(load_dt));
    k_convert(launch(), d, load_dt, cols_p, src, DT_F32, cols, rows);
    return d;
}
// finish a Lin whose fp32 [out][in] weight already sits on the device: fp32 graphs use it as it is; bf16 graphs get a bf16 copy
// (rows padded to 8 columns). Both keep its transpose [in][out] for the input gradient dx = dy . W: with it that GEMM has two
// k-contiguous operands and runs on the LDS-DMA kernel (cb_gemm_glds.hip) instead of an m-contiguous register-staged loader.
static void finish_linear(Ctx* ctx, Lin* l, float* w32) {
    l->dt = ctx->load_dt;
    l->in_p = pad8(l->dt, l->in);
    if (l->dt == DT_F32) {
        l->w = w32;
        if (l->in % 4 == 0 && l->out % 4 == 0) {
            l->wT = ctx->pool.alloc((size_t)l->in * l->out * sizeof(float));
            k_transpose(ctx->launch(), DT_F32, l->wT, l->w, 1, l->out, l->in, l->in, 0);
        }
        return;
    }
    l->w = ctx->operand(w32, l->out, l->in, l->in_p);
    if (l->in_p == l->in && l->out % 8 == 0) {
        l->wT = ctx->pool.alloc((size_t)l->in * l->out * dt_size(l->dt));
        k_transpose(ctx->launch(), l->dt, l->wT, l->w, 1, l->out, l->in, l->in, 0);
    }
}

Conv load_conv(Ctx* ctx, const WTab& wt, const std::string& prefix, int cin, int cout, int ks, bool need_dgrad, bool bias, bool up_dgrad) {
    Conv c; c.cin = cin; c.cout = cout; c.ks = ks;
    c.dt = ctx->load_dt; c.cin_p = pad8(c.dt, cin); c.cout_p = pad8(c.dt, cout);
    Launch L = ctx->launch();
    float* raw = ctx->upload(wt, prefix + ".weight", {cout, cin, ks, ks});
    c.w = ctx->pool.alloc((size_t)cout * ks * ks * c.cin_p * dt_size(c.dt));
    k_pack_conv(L, c.dt, c.w, raw, cout, cin, ks, c.cin_p);
    if (need_dgrad) {
        c.wd = ctx->pool.alloc((size_t)cin * ks * ks * c.cout_p * dt_size(c.dt));
        k_pack_conv_dgrad(L, c.dt, c.wd, raw, cout, cin, ks, c.cout_p);
    }
    if (up_dgrad && ks == 3) {
        c.wu = ctx->pool.alloc((size_t)cin * 16 * c.cout_p * dt_size(c.dt));
        k_pack_conv_up_dgrad(L, c.dt, c.wu, raw, cout, cin, c.cout_p);
    }
    if (bias) c.b = ctx->upload(wt, prefix + ".bias", {cout});
    return c;
}
Lin load_linear(Ctx* ctx, const WTab& wt, const std::string& prefix, int in, int out, bool bias) {
    Lin l; l.in = in; l.out = out;
    finish_linear(ctx, &l, ctx->upload(wt, prefix + ".weight", {out, in}));
    if (bias) l.b = ctx->upload(wt, prefix + ".bias", {out});
    return l;
}
Lin load_conv1x1_as_linear(Ctx* ctx, const WTab& wt, const std::string& prefix, int in, int out) {
    Lin l; l.in = in; l.out = out;
    finish_linear(ctx, &l, ctx->upload(wt, prefix + ".weight", {out, in, 1, 1}));
    l.b = ctx->upload(wt, prefix + ".bias", {out});
    return l;
}
// concatenate several [out_each, in] projections into one [n*out_each, in] weight (fused QKV / KV)
Lin load_linear_cat(Ctx* ctx, const WTab& wt, const std::vector<std::string>& prefixes, int in, int out_each, bool bias,
                    bool conv1x1) {
    Lin l; l.in = in; l.out = out_each * (int)prefixes.size();
    float* w32 = ctx->pool.allocf((size_t)l.out * in);
    if (bias) l.b = ctx->pool.allocf((size_t)l.out);
    Launch L = ctx->launch();
    for (size_t i = 0; i < prefixes.size(); ++i) {
        float* w = conv1x1 ? ctx->upload(wt, prefixes[i] + ".weight", {out_each, in, 1, 1})
                           : ctx->upload(wt, prefixes[i] + ".weight", {out_each, in});
        k_axpby(L, w32 + i * (size_t)out_each * in, w, (long)out_each * in, 1.f, 0.f);
        if (bias) {
            float* b = ctx->upload(wt, prefixes[i] + ".bias", {out_each});
            k_axpby(L, l.b + i * (size_t)out_each, b, out_each, 1.f, 0.f);
        }
    }
    finish_linear(ctx, &l, w32);
    return l;
}
Lin load_linear_cat_var(Ctx* ctx, const WTab& wt, const std::vector<std::string>& prefixes, int in, const std::vector<int>& outs,
                        bool bias) {
    CB_CHECK(prefixes.size() == outs.size(), CB_ERR_ARG, "load_linear_cat_var: size mismatch");
    Lin l; l.in = in; l.out = 0;
    for (int o : outs) l.out += o;
    float* w32 = ctx->pool.allocf((size_t)l.out * in);
    if (bias) l.b = ctx->pool.allocf((size_t)l.out);
    Launch L = ctx->launch();
    size_t row = 0;
    for (size_t i = 0; i < prefixes.size(); ++i) {
        float* w = ctx->upload(wt, prefixes[i] + ".weight", {outs[i], in});
        k_axpby(L, w32 + row * in, w, (long)outs[i] * in, 1.f, 0.f);
        if (bias) k_axpby(L, l.b + row, ctx->upload(wt, prefixes[i] + ".bias", {outs[i]}), outs[i], 1.f, 0.f);
        row += outs[i];
    }
    finish_linear(ctx, &l, w32);
    return l;
}
Norm load_norm(Ctx* ctx, const WTab& wt, const std::string& prefix, int c, float eps) {
    Norm n; n.c = c; n.eps = eps;
    n.g = ctx->upload(wt, prefix + ".weight", {c});
    n.b = ctx->upload(wt, prefix + ".bias", {c});
    return n;
}

// ---------------------------------------------------------------- graph basics
Ten* Graph::make(int n, int h, int w, int c, bool rg, DType dt) {
    tens.emplace_back();
    Ten* t = &tens.back();
    t->n = n; t->h = h; t->w = w; t->c = c; t->rg = rg; t->dt = dt;
    t->p = ctx->arena.alloc(t->bytes());
    return t;
}
Ten* Graph::wrap(const void* p, int n, int h, int w, int c, bool rg, DType dt) {
    tens.emplace_back();
    Ten* t = &tens.back();
    t->n = n; t->h = h; t->w = w; t->c = c; t->rg = rg; t->dt = dt; t->p = const_cast<void*>(p);
    return t;
}
void* Graph::grad(Ten* t, float* beta) {
    if (!t->g) t->g = ctx->arena.alloc(t->bytes());
    *beta = t->ginit ? 1.f : 0.f;
    t->ginit = true;
    return t->g;
}
void Graph::tap(const std::string& name, const Ten* t) {
    if (ctx->debug_taps) ctx->taps[name] = Ctx::Tap{t->p, t->numel(), t->dt};
}
void Graph::backward() { backward_until(0); }
void Graph::backward_until(size_t mark) {
    while (tape.size() > mark) {
        std::function<void()> fn = std::move(tape.back());
        tape.pop_back();
        fn();
    }
}
// fp32 <-> bf16 at the edges of a bf16 graph (the public tensors of the step, the coefficient head and the FR-net are fp32)
Ten* Graph::cast(Ten* x, DType dt) {
    if (x->dt == dt) return x;
    Ten* y = make(x->n, x->h, x->w, x->c, x->rg, dt);
    k_convert(L, y->p, dt, x->c, x->p, x->dt, x->c, x->rows());
    if (x->rg) {
        tape.push_back([this, x, y]() {
            if (!y->ginit) return;
            float beta; void* gx = grad(x, &beta);
            k_convert(L, gx, x->dt, x->c, y->g, y->dt, y->c, x->rows(), beta);
        });
    }
    return y;
}

static void run_gemm(Graph* G, const GemmArgs& a) {
    if (a.dt == DT_BF16) {
        size_t wsb = gemm_bf16_ws_bytes(a);
        float* ws = wsb ? static_cast<float*>(G->ctx->arena.alloc(wsb)) : nullptr;
        gemm_bf16(G->L, a, ws, wsb);
        return;
    }
    size_t wsb = gemm_ws_bytes(a);
    float* ws = wsb ? static_cast<float*>(G->ctx->arena.alloc(wsb)) : nullptr;
    gemm(G->L, a, ws, wsb);
}

static void accumulate(Graph* G, Ten* t, const void* src) {
    float beta;
    void* g = G->grad(t, &beta);
    k_axpby(G->L, t->dt, g, src, t->numel(), 1.f, beta);
}
// y = f(...) + residual: d residual = dy. When nothing has been written to residual.g yet, the residual simply ADOPTS
// y.g as its gradient buffer (no copy): y.g is dead after y's own backward closure — the tape runs each closure once, in
// reverse order, and only the closure of the op that produced y reads y.g.
static void pass_grad_to_residual(Graph* G, Ten* residual, Ten* y) {
    if (!residual->g && !residual->ginit) { residual->g = y->g; residual->ginit = true; return; }
    accumulate(G, residual, y->g);
}
// zero-pad the channel axis of an activation (bf16 graphs: 16-byte rows for the 3 / 4-channel ends of the networks)
static const void* pad_channels(Graph* G, const void* x, DType dt, long rows, int c, int cp) {
    if (cp == c) return x;
    void* t = G->ctx->arena.alloc((size_t)rows * cp * dt_size(dt));
    k_convert(G->L, t, dt, cp, x, dt, c, rows);
    return t;
}

// ---------------------------------------------------------------- conv
void conv_out_dims(const Conv& cv, ConvMode mode, int H, int W, int* Ho, int* Wo, ConvGeom* g) {
    g->H = H; g->W = W; g->C = cv.cin_p ? cv.cin_p : cv.cin; g->ks = cv.ks; g->up = 0; g->tr = 0;
    *Ho = H; *Wo = W;
    switch (mode) {
        case CONV_SAME: g->stride = 1; g->pad = cv.ks / 2; break;
        case CONV_DOWN: CB_CHECK(cv.ks == 3, CB_ERR_ARG, "conv down: ks"); g->stride = 2; g->pad = 1;
            *Ho = (H + 2 - 3) / 2 + 1; *Wo = (W + 2 - 3) / 2 + 1; break;
        case CONV_DOWN_ASYM: CB_CHECK(cv.ks == 3, CB_ERR_ARG, "conv down: ks"); g->stride = 2; g->pad = 0;
            *Ho = (H + 1 - 3) / 2 + 1; *Wo = (W + 1 - 3) / 2 + 1; break;
        case CONV_UP: CB_CHECK(cv.ks == 3, CB_ERR_ARG, "conv up: ks"); g->stride = 1; g->pad = 1; g->up = 1; *Ho = 2 * H; *Wo = 2 * W; break;
        case CONV_S2_1x1: CB_CHECK(cv.ks == 1, CB_ERR_ARG, "conv s2 1x1: ks"); g->stride = 2; g->pad = 0;
            *Ho = (H - 1) / 2 + 1; *Wo = (W - 1) / 2 + 1; break;
        default: fail(CB_ERR_ARG, "conv: unknown mode %d", (int)mode);
    }
    g->Ho = *Ho; g->Wo = *Wo;
}
// dx[N,H,W,cin] (beta-accumulated) from dy[N,Ho,Wo,cout] using the flipped/transposed pack cv.wd
void conv_dgrad(Graph* G, const void* dy, const Conv& cv, ConvMode mode, int N, int H, int W, void* dx, float beta) {
    CB_CHECK(cv.wd, CB_ERR_STATE, "conv: dgrad weights were not packed");
    CB_CHECK(mode == CONV_SAME || mode == CONV_DOWN || mode == CONV_UP, CB_ERR_ARG, "conv: mode %d has no dgrad", (int)mode);
    int Ho, Wo; ConvGeom fg;
    conv_out_dims(cv, mode, H, W, &Ho, &Wo, &fg);
    const int cop = cv.cout_p ? cv.cout_p : cv.cout;
    GemmArgs d;
    d.dt = cv.dt;
    d.A = pad_channels(G, dy, cv.dt, (long)N * Ho * Wo, cv.cout, cop); d.B = cv.wd;
    d.N = cv.cin; d.K = cv.ks * cv.ks * cop;
    d.a_kind = LD_CONV; d.b_kind = LD_ROW; d.ldb = d.K; d.ldc = cv.cin;
    ConvGeom dg; dg.H = Ho; dg.W = Wo; dg.C = cop; dg.ks = cv.ks; dg.stride = 1; dg.pad = cv.ks - 1 - cv.ks / 2;
    if (mode == CONV_SAME) {
        dg.Ho = H; dg.Wo = W;
        d.C = dx; d.beta = beta; d.M = N * H * W; d.g = dg;
        run_gemm(G, d);
    } else if (mode == CONV_DOWN) {      // transposed stride-2 gather
        dg.Ho = H; dg.Wo = W; dg.tr = 1; dg.pad = 1;
        d.C = dx; d.beta = beta; d.M = N * H * W; d.g = dg;
        run_gemm(G, d);
    } else if (cv.wu && (cv.dt == DT_BF16 || cop % 4 == 0)) {
        // CONV_UP with the combined pack: one 4x4 stride-2 pad-1 conv over dy (2H x 2W) straight into dx (cb_kernels.hip,
        // pack_conv_up_dgrad_kernel): 16 * Cout MACs per input element instead of 36 * Cout, no temporary, no pooling pass
        ConvGeom ug; ug.H = Ho; ug.W = Wo; ug.C = cop; ug.ks = 4; ug.stride = 2; ug.pad = 1; ug.Ho = H; ug.Wo = W;
        d.B = cv.wu; d.K = 16 * cop; d.ldb = d.K;
        d.C = dx; d.beta = beta; d.M = N * H * W; d.g = ug;
        run_gemm(G, d);
    } else {                              // CONV_UP: dgrad at the upsampled resolution, then 2x2 sum-pool
        dg.Ho = Ho; dg.Wo = Wo;
        void* tmp = G->ctx->arena.alloc((size_t)N * Ho * Wo * cv.cin * dt_size(cv.dt));
        d.C = tmp; d.beta = 0.f; d.M = N * Ho * Wo; d.g = dg;
        run_gemm(G, d);
        k_upsample2x_bwd(G->L, cv.dt, dx, tmp, N, H, W, cv.cin, beta);
    }
}

Ten* Graph::conv(Ten* x, const Conv& cv, ConvMode mode, Ten* rowbias, Ten* residual, int rowbias_col, const ConvEpi* epi) {
    CB_CHECK(x->c == cv.cin, CB_ERR_SHAPE, "conv: input has %d channels, weight expects %d", x->c, cv.cin);
    CB_CHECK(x->dt == cv.dt, CB_ERR_ARG, "conv: activation / weight element types differ");
    const int H = x->h, W = x->w;
    ConvGeom g; int Ho, Wo;
    conv_out_dims(cv, mode, H, W, &Ho, &Wo, &g);
    bool rg = x->rg || (residual && residual->rg);
    Ten* y = make(x->n, Ho, Wo, cv.cout, rg, x->dt);
    if (residual) CB_CHECK(residual->numel() == y->numel() && residual->c == y->c && residual->dt == y->dt, CB_ERR_SHAPE, "conv: residual shape");
    if (rowbias) CB_CHECK(rowbias->rows() == x->n && rowbias_col >= 0 && rowbias->c >= rowbias_col + cv.cout && rowbias->dt == y->dt,
                          CB_ERR_SHAPE, "conv: rowbias shape");
    GemmArgs a;
    a.dt = x->dt;
    a.A = pad_channels(this, x->p, x->dt, x->rows(), cv.cin, g.C); a.B = cv.w; a.C = y->p;
    a.M = (int)y->rows(); a.N = cv.cout; a.K = cv.ks * cv.ks * g.C;
    a.a_kind = LD_CONV; a.b_kind = LD_ROW; a.ldb = a.K; a.ldc = cv.cout;
    a.bias = cv.b;
    if (rowbias) { a.rowbias = rowbias->at(rowbias_col); a.ld_rowbias = rowbias->c; a.rows_per_img = Ho * Wo; }   // column view of a wider table
    if (residual) { a.R = residual->p; a.ldr = cv.cout; }
    a.g = g;
    if (epi) {
        CB_CHECK(!rg && x->dt == DT_F32, CB_ERR_STATE, "conv: the PReLU / second-output epilogue is a no-grad fp32 path (frozen FR-net)");
        a.prelu = epi->prelu;
        if (epi->c2_out) {
            Ten* y2 = make(x->n, Ho, Wo, cv.cout, false, x->dt);
            a.C2 = y2->p; a.c2_scale = epi->c2_scale; a.c2_shift = epi->c2_shift;
            *epi->c2_out = y2;
        }
    }
    run_gemm(this, a);
    if (rg) {
        CB_CHECK(!x->rg || cv.wd, CB_ERR_STATE, "conv: dgrad weights were not packed for a layer on the gradient path");
        CB_CHECK(!x->rg || mode == CONV_SAME || mode == CONV_DOWN || mode == CONV_UP, CB_ERR_ARG, "conv: mode has no dgrad");
        Conv cvc = cv;
        tape.push_back([this, x, y, residual, cvc, mode]() {
            if (!y->ginit) return;
            if (residual && residual->rg) pass_grad_to_residual(this, residual, y);
            if (!x->rg) return;
            float beta; void* gx = grad(x, &beta);
            conv_dgrad(this, y->g, cvc, mode, x->n, x->h, x->w, gx, beta);
        });
    }
    return y;
}

// ---------------------------------------------------------------- linear
Ten* Graph::linear(Ten* x, const Lin& l, Ten* residual) {
    CB_CHECK(x->c == l.in, CB_ERR_SHAPE, "linear: input has %d features, weight expects %d", x->c, l.in);
    CB_CHECK(x->dt == l.dt, CB_ERR_ARG, "linear: activation / weight element types differ");
    bool rg = x->rg || (residual && residual->rg);
    Ten* y = make(x->n, x->h, x->w, l.out, rg, x->dt);
    long rows = x->rows();
    const int inp = l.in_p ? l.in_p : l.in;
    if (residual) CB_CHECK(residual->numel() == y->numel() && residual->dt == y->dt, CB_ERR_SHAPE, "linear: residual shape");
    // few rows: a one-wave-per-output-column GEMV streams the weight once — but it re-reads all `rows` input rows per weight chunk, so it
    // only pays while those stay cache-resident (time embedding, emb_layers, the MLP head: <= 1280 inputs per row). The FR backbone's
    // 25088 -> 512 fc at 8 faces ran 220 us that way (233 GB/s of weight traffic); the split-K GEMM takes it instead.
    if (rows <= 8 && !residual && inp == l.in && l.in <= 8192) {
        k_gemv(L, x->dt, y->p, x->p, l.w, l.b, (int)rows, l.out, l.in);
    } else {
        GemmArgs a;
        a.dt = x->dt;
        a.A = pad_channels(this, x->p, x->dt, rows, l.in, inp); a.B = l.w; a.C = y->p; a.M = (int)rows; a.N = l.out; a.K = inp;
        a.lda = inp; a.ldb = inp; a.ldc = l.out; a.bias = l.b;
        if (residual) { a.R = residual->p; a.ldr = l.out; }
        run_gemm(this, a);
    }
    if (rg) {
        Lin lc = l;
        CB_CHECK(!x->rg || x->dt == DT_F32 || lc.wT, CB_ERR_STATE, "linear: no transposed bf16 weight for a layer on the gradient path");
        tape.push_back([this, x, y, residual, lc, rows]() {
            if (!y->ginit) return;
            if (residual && residual->rg) pass_grad_to_residual(this, residual, y);
            if (!x->rg) return;
            GemmArgs d;   // dx[rows,in] = dy[rows,out] . W[out,in]
            d.dt = x->dt;
            d.A = y->g; d.lda = lc.out; d.a_kind = LD_ROW;
            if (lc.wT) { d.B = lc.wT; d.ldb = lc.out; d.b_kind = LD_ROW; }                // W^T [in][out]: k = out contiguous
            else { d.B = lc.w; d.ldb = lc.in; d.b_kind = LD_COL; }
            d.M = (int)rows; d.N = lc.in; d.K = lc.out; d.ldc = lc.in;
            float beta; d.C = grad(x, &beta); d.beta = beta;
            run_gemm(this, d);
        });
    }
    return y;
}

// ---------------------------------------------------------------- norms
Ten* Graph::groupnorm(Ten* x, const Norm& nm, bool silu) {
    CB_CHECK(x->c == nm.c, CB_ERR_SHAPE, "groupnorm: channels %d vs %d", x->c, nm.c);
    const int G = 32, N = x->n, HW = x->h * x->w, C = x->c;
    Ten* y = make(x->n, x->h, x->w, x->c, x->rg, x->dt);
    float* stats = ctx->arena.allocf((size_t)N * G * 2);
    float* ws = ctx->arena.allocf(gn_ws_floats(N, HW, C, G) + (size_t)N * G * 2);
    k_groupnorm_fwd(L, x->dt, y->p, x->p, nm.g, nm.b, stats, ws, N, HW, C, G, nm.eps, silu ? 1 : 0);
    if (x->rg) {
        Norm nc = nm;
        tape.push_back([this, x, y, nc, stats, ws, N, HW, C, silu]() {
            if (!y->ginit) return;
            float beta; void* gx = grad(x, &beta);
            k_groupnorm_bwd(L, x->dt, gx, x->p, y->g, nc.g, nc.b, stats, ws, N, HW, C, 32, silu ? 1 : 0, beta, /*chn_valid=*/true);
        });
    }
    return y;
}
Ten* Graph::layernorm(Ten* x, const Norm& nm) {
    CB_CHECK(x->c == nm.c, CB_ERR_SHAPE, "layernorm: channels %d vs %d", x->c, nm.c);
    Ten* y = make(x->n, x->h, x->w, x->c, x->rg, x->dt);
    long rows = x->rows();
    float* stats = ctx->arena.allocf((size_t)rows * 2);
    k_layernorm_fwd(L, x->dt, y->p, x->p, nm.g, nm.b, stats, rows, x->c, nm.eps);
    if (x->rg) {
        Norm nc = nm;
        tape.push_back([this, x, y, nc, stats, rows]() {
            if (!y->ginit) return;
            float beta; void* gx = grad(x, &beta);
            k_layernorm_bwd(L, x->dt, gx, x->p, y->g, nc.g, stats, rows, x->c, beta);
        });
    }
    return y;
}

// ---------------------------------------------------------------- attention: fused (cb_flash*.hip) or materialised scores
// attention.py:170-193 CrossAttention (scale after QK^T, softmax over context), model.py:186-198 AttnBlock,
// transformers-4.18 CLIPAttention (causal additive mask).
Ten* Graph::attention(Ten* qt, int qoff, Ten* kt, int koff, int voff, int C, int heads, float scale, bool causal) {
    const int B = qt->n, Nq = qt->h * qt->w, Nk = kt->h * kt->w, d = C / heads;
    CB_CHECK(kt->n == B && C % heads == 0, CB_ERR_SHAPE, "attention: batch/head mismatch");
    CB_CHECK(qoff + C <= qt->c && koff + C <= kt->c && voff + C <= kt->c, CB_ERR_SHAPE, "attention: column views out of range");
    CB_CHECK(qt->dt == kt->dt, CB_ERR_ARG, "attention: q and k/v element types differ");
    const long ldq = qt->c, ldk = kt->c;
    const DType dt = qt->dt;
    const bool bf = dt == DT_BF16;
    bool rg = qt->rg || kt->rg;
    // fused path for head dims <= 160: scores/probabilities never touch HBM
    FlashArgs f;
    f.dt = dt;
    f.q = qt->at(qoff); f.ldq = ldq; f.sqb = (long)Nq * ldq; f.sqh = d;
    f.k = kt->at(koff); f.v = kt->at(voff); f.ldk = ldk; f.skb = (long)Nk * ldk; f.skh = d;
    f.ldo = C; f.sob = (long)Nq * C; f.soh = d;
    f.H = heads; f.Nq = Nq; f.Nk = Nk; f.d = d; f.scale = scale; f.causal = causal ? 1 : 0;
    Ten* o = make(B, qt->h, qt->w, C, rg, dt);
    f.o = o->p;
    if (use_flash && (bf ? flash_bf16_supported(d, f) : flash_supported(d, f))) {
        f.lse = ctx->arena.allocf((size_t)B * heads * Nq);
        if (!bf) { size_t wf = flash_ksplit_ws_floats(f, B, true, &f.ksplit); if (wf) f.wsk = ctx->arena.allocf(wf); }
        if (bf) { size_t tb = flash_bf16_v2_tws_bytes(f, B, false); if (tb) f.tws = ctx->arena.alloc(tb); }
        if (bf) flash_bf16_forward(L, f, B); else flash_forward(L, f, B);
        if (rg) {
            tape.push_back([this, f, o, qt, kt, qoff, koff, voff, B, Nq, bf]() mutable {
                if (!o->ginit) return;
                f.dout = o->g;
                f.dvec = ctx->arena.allocf((size_t)B * f.H * Nq);
                float bq = 0.f, bk = 0.f; void* gq = nullptr; void* gk = nullptr;
                if (qt->rg) gq = grad(qt, &bq);
                if (kt->rg) { if (kt == qt) { gk = gq; bk = bq; } else gk = grad(kt, &bk); }
                if (gq) { f.dq = ptr_off(gq, f.dt, qoff); f.beta_q = bq; }
                f.ksplit = 1; f.wsk = nullptr;
                if (gq && !bf) { size_t wq = flash_ksplit_ws_floats(f, B, false, &f.ksplit); if (wq) f.wsk = ctx->arena.allocf(wq); }
                f.tws = nullptr;
                if (bf && gq) { size_t tb = flash_bf16_v2_tws_bytes(f, B, true); if (tb) f.tws = ctx->arena.alloc(tb); }   // v2 needs the dQ kernel's D
                if (gk) {
                    f.dk = ptr_off(gk, f.dt, koff); f.dv = ptr_off(gk, f.dt, voff); f.beta_kv = bk;
                    f.nsplit = 1; f.ws = nullptr;
                    if (!f.tws) {
                        size_t wsf = bf ? flash_bf16_bwd_ws_floats(f, B, &f.nsplit) : flash_bwd_ws_floats(f, B, &f.nsplit);
                        if (wsf) f.ws = ctx->arena.allocf(wsf);
                    }
                }
                if (bf) flash_bf16_backward(L, f, B, gq != nullptr, gk != nullptr);
                else flash_backward(L, f, B, gq != nullptr, gk != nullptr);
            });
        }
        return o;
    }
    // otherwise (VAE AttnBlock, d = 512) the scores are materialised, reusing `o`
    return attention_materialised(qt, qoff, kt, koff, voff, C, heads, scale, causal, o);
}

Ten* Graph::attention_materialised(Ten* qt, int qoff, Ten* kt, int koff, int voff, int C, int heads, float scale, bool causal, Ten* o) {
    const int B = qt->n, Nq = qt->h * qt->w, Nk = kt->h * kt->w, d = C / heads;
    const long ldq = qt->c, ldk = kt->c;
    const DType dt = qt->dt;
    const int ldS = (int)rup(Nk, dt == DT_BF16 ? 8 : 4);
    const int BH = B * heads;
    bool rg = qt->rg || kt->rg;
    void* P = ctx->arena.alloc((size_t)BH * Nq * ldS * dt_size(dt));

    GemmArgs s;   // S = scale * Q K^T
    s.dt = dt;
    s.A = qt->at(qoff); s.lda = ldq; s.sAb = (long)Nq * ldq; s.sAh = d;
    s.B = kt->at(koff); s.ldb = ldk; s.sBb = (long)Nk * ldk; s.sBh = d;
    s.C = P; s.ldc = ldS; s.sCb = (long)heads * Nq * ldS; s.sCh = (long)Nq * ldS;
    s.M = Nq; s.N = Nk; s.K = d; s.alpha = scale; s.batch = BH; s.nh = heads;
    run_gemm(this, s);
    k_softmax_fwd(L, dt, P, (long)BH * Nq, Nk, ldS, causal ? 1 : 0, Nq);
    GemmArgs pv;  // O = P V
    pv.dt = dt;
    pv.A = P; pv.lda = ldS; pv.sAb = s.sCb; pv.sAh = s.sCh;
    if (dt == DT_BF16) {
        // the bf16 GEMM takes k-contiguous operands: V^T [B][heads][d][Nk] (the forward-only VAE AttnBlock, one head of 512)
        CB_CHECK(!rg, CB_ERR_STATE, "materialised bf16 attention is forward-only");
        CB_CHECK(Nk % 8 == 0, CB_ERR_SHAPE, "materialised bf16 attention: Nk=%d must be a multiple of 8", Nk);
        void* Vt = ctx->arena.alloc((size_t)BH * d * Nk * dt_size(dt));
        for (int h = 0; h < heads; ++h)      // [B][Nk][d-slice of head h] -> [B][heads][d][Nk]
            for (int b = 0; b < B; ++b)
                k_transpose(L, dt, ptr_off(Vt, dt, ((long)b * heads + h) * d * Nk), kt->at((long)b * Nk * ldk + voff + (long)h * d), 1, Nk, d, ldk, 0);
        pv.B = Vt; pv.ldb = Nk; pv.b_kind = LD_ROW; pv.sBb = (long)heads * d * Nk; pv.sBh = (long)d * Nk;
        if (s.K % 8) fail(CB_ERR_SHAPE, "materialised bf16 attention: head dim %d must be a multiple of 8", d);
        // the zero columns of P beyond Nk (ldS padding) are never read: K = Nk
    } else {
        pv.B = kt->at(voff); pv.ldb = ldk; pv.b_kind = LD_COL; pv.sBb = (long)Nk * ldk; pv.sBh = d;
    }
    pv.C = o->p; pv.ldc = C; pv.sCb = (long)Nq * C; pv.sCh = d;
    pv.M = Nq; pv.N = d; pv.K = Nk; pv.batch = BH; pv.nh = heads;
    run_gemm(this, pv);

    if (rg) {
        tape.push_back([=]() {
            if (!o->ginit) return;
            float* Pf = static_cast<float*>(P);
            float* dP = ctx->arena.allocf((size_t)BH * Nq * ldS);
            GemmArgs a;   // dP = dO V^T
            a.A = o->g; a.lda = C; a.sAb = (long)Nq * C; a.sAh = d;
            a.B = kt->at(voff); a.ldb = ldk; a.sBb = (long)Nk * ldk; a.sBh = d;
            a.C = dP; a.ldc = ldS; a.sCb = (long)heads * Nq * ldS; a.sCh = (long)Nq * ldS;
            a.M = Nq; a.N = Nk; a.K = d; a.batch = BH; a.nh = heads;
            run_gemm(this, a);
            k_softmax_bwd(L, dP, Pf, (long)BH * Nq, Nk, ldS, scale);   // dP now holds d(QK^T)
            float bq = 0.f, bk = 0.f; void* gq = nullptr; void* gk = nullptr;
            if (qt->rg) gq = grad(qt, &bq);
            if (kt->rg) { if (kt == qt) { gk = gq; bk = bq; } else gk = grad(kt, &bk); }
            if (gq) {     // dQ = dS K
                GemmArgs q;
                q.A = dP; q.lda = ldS; q.sAb = a.sCb; q.sAh = a.sCh;
                q.B = kt->at(koff); q.ldb = ldk; q.b_kind = LD_COL; q.sBb = (long)Nk * ldk; q.sBh = d;
                q.C = ptr_off(gq, DT_F32, qoff); q.ldc = ldq; q.sCb = (long)Nq * ldq; q.sCh = d; q.beta = bq;
                q.M = Nq; q.N = d; q.K = Nk; q.batch = BH; q.nh = heads;
                run_gemm(this, q);
            }
            if (gk) {     // dK = dS^T Q ; dV = P^T dO
                GemmArgs k;
                k.A = dP; k.lda = ldS; k.a_kind = LD_COL; k.sAb = a.sCb; k.sAh = a.sCh;
                k.B = qt->at(qoff); k.ldb = ldq; k.b_kind = LD_COL; k.sBb = (long)Nq * ldq; k.sBh = d;
                k.C = ptr_off(gk, DT_F32, koff); k.ldc = ldk; k.sCb = (long)Nk * ldk; k.sCh = d; k.beta = bk;
                k.M = Nk; k.N = d; k.K = Nq; k.batch = BH; k.nh = heads;
                run_gemm(this, k);
                GemmArgs v;
                v.A = Pf; v.lda = ldS; v.a_kind = LD_COL; v.sAb = a.sCb; v.sAh = a.sCh;
                v.B = o->g; v.ldb = C; v.b_kind = LD_COL; v.sBb = (long)Nq * C; v.sBh = d;
                v.C = ptr_off(gk, DT_F32, voff); v.ldc = ldk; v.sCb = (long)Nk * ldk; v.sCh = d; v.beta = bk;
                v.M = Nk; v.N = d; v.K = Nq; v.batch = BH; v.nh = heads;
                run_gemm(this, v);
            }
        });
    }
    return o;
}

// ---------------------------------------------------------------- activations / glue
Ten* Graph::geglu(Ten* x) {
    CB_CHECK(x->c % 2 == 0, CB_ERR_SHAPE, "geglu: odd width");
    const int D = x->c / 2;
    Ten* y = make(x->n, x->h, x->w, D, x->rg, x->dt);
    k_geglu_fwd(L, x->dt, y->p, x->p, x->rows(), D);
    if (x->rg) {
        tape.push_back([this, x, y, D]() {
            if (!y->ginit) return;
            float beta; void* gx = grad(x, &beta);
            if (beta == 0.f) k_geglu_bwd(L, x->dt, gx, x->p, y->g, x->rows(), D);
            else {
                void* tmp = ctx->arena.alloc(x->bytes());
                k_geglu_bwd(L, x->dt, tmp, x->p, y->g, x->rows(), D);
                k_axpby(L, x->dt, gx, tmp, x->numel(), 1.f, 1.f);
            }
        });
    }
    return y;
}
Ten* Graph::quickgelu(Ten* x) {
    Ten* y = make(x->n, x->h, x->w, x->c, x->rg, x->dt);
    k_quickgelu_fwd(L, x->dt, y->p, x->p, x->numel());
    if (x->rg) {
        tape.push_back([this, x, y]() {
            if (!y->ginit) return;
            float beta; void* gx = grad(x, &beta);
            k_quickgelu_bwd(L, x->dt, gx, x->p, y->g, x->numel(), beta);
        });
    }
    return y;
}
Ten* Graph::silu(Ten* x) {
    CB_CHECK(!x->rg, CB_ERR_STATE, "silu: only used on the no-grad timestep path");
    Ten* y = make(x->n, x->h, x->w, x->c, false, x->dt);
    k_silu(L, x->dt, y->p, x->p, x->numel());
    return y;
}
Ten* Graph::concat(Ten* a, Ten* b) {
    CB_CHECK(a->rows() == b->rows() && a->h == b->h && a->w == b->w && a->dt == b->dt, CB_ERR_SHAPE, "concat: spatial / type mismatch");
    const int ca = a->c, cb_ = b->c;
    const DType dt = a->dt;
    Ten* y = make(a->n, a->h, a->w, ca + cb_, a->rg || b->rg, dt);
    long rows = a->rows();
    k_copy2d(L, dt, y->p, ca + cb_, a->p, ca, rows, ca, 0.f);
    k_copy2d(L, dt, y->at(ca), ca + cb_, b->p, cb_, rows, cb_, 0.f);
    if (y->rg) {
        tape.push_back([this, a, b, y, ca, cb_, rows, dt]() {
            if (!y->ginit) return;
            if (a->rg) { float beta; void* g = grad(a, &beta); k_copy2d(L, dt, g, ca, y->g, ca + cb_, rows, ca, beta); }
            if (b->rg) { float beta; void* g = grad(b, &beta); k_copy2d(L, dt, g, cb_, y->gat(ca), ca + cb_, rows, cb_, beta); }
        });
    }
    return y;
}
Ten* Graph::channel_affine(Ten* x, const float* scale, const float* shift, const float* prelu) {
    CB_CHECK(!x->rg && x->dt == DT_F32, CB_ERR_STATE, "channel_affine: no-grad fp32 path only (frozen FR-net)");
    Ten* y = make(x->n, x->h, x->w, x->c, false, DT_F32);
    k_channel_affine(L, y->pf(), x->pf(), scale, shift, prelu, x->rows(), x->c);
    return y;
}

}  // namespace cb
