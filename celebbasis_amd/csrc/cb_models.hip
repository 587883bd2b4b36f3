// cb_models.hip — UNet / VAE encoder / CLIP text transformer / iresnet / CelebBasis head / training step,
// expressed on the tape runtime. Weight names follow the reference state-dict keys.
#include "cb_models.h"
#include "cb_gemm_tiles.h"
#include <cmath>

namespace cb {

static std::string S(const std::string& a, int i) { return a + std::to_string(i); }

// ================================================================= UNet
// Every ResBlock's emb_layers (SiLU -> Linear on the shared time embedding) is stacked into one weight and evaluated by one
// launch per step; the blocks read column views of the result as their per-image row bias.
struct CatCollect { std::vector<std::string> prefixes; std::vector<int> outs; int cols = 0; };
typedef CatCollect KvCollect;
static ResBlockW load_resblock(Ctx* ctx, const WTab& wt, const std::string& p, int cin, int cout, CatCollect* embc) {
    ResBlockW r; r.cin = cin; r.cout = cout;
    r.n1 = load_norm(ctx, wt, p + ".in_layers.0", cin, 1e-5f);              // GroupNorm32, util.py:214-216
    r.c1 = load_conv(ctx, wt, p + ".in_layers.2", cin, cout, 3, true);
    r.emb_col = embc->cols;
    embc->prefixes.push_back(p + ".emb_layers.1"); embc->outs.push_back(cout); embc->cols += cout;
    r.n2 = load_norm(ctx, wt, p + ".out_layers.0", cout, 1e-5f);
    r.c2 = load_conv(ctx, wt, p + ".out_layers.3", cout, cout, 3, true);
    if (cin != cout) { r.has_skip = true; r.skip = load_conv1x1_as_linear(ctx, wt, p + ".skip_connection", cin, cout); }
    return r;
}
// The 32 cross-attention K/V projections depend only on the conditioning: they are stacked into ONE weight at load time and
// evaluated by one GEMM per step (and one dgrad GEMM back to the context) instead of 16 + 16 tiny M=77 launches.
static XfmrW load_xfmr(Ctx* ctx, const WTab& wt, const std::string& p, int ch, int heads, int ctx_dim, KvCollect* kvc) {
    XfmrW x; x.ch = ch; x.heads = heads;
    x.norm = load_norm(ctx, wt, p + ".norm", ch, 1e-6f);                    // attention.py:76-77 Normalize eps 1e-6
    x.proj_in = load_conv1x1_as_linear(ctx, wt, p + ".proj_in", ch, ch);
    const std::string tb = p + ".transformer_blocks.0";
    x.ln1 = load_norm(ctx, wt, tb + ".norm1", ch, 1e-5f);
    x.ln2 = load_norm(ctx, wt, tb + ".norm2", ch, 1e-5f);
    x.ln3 = load_norm(ctx, wt, tb + ".norm3", ch, 1e-5f);
    x.qkv1 = load_linear_cat(ctx, wt, {tb + ".attn1.to_q", tb + ".attn1.to_k", tb + ".attn1.to_v"}, ch, ch, false);
    x.out1 = load_linear(ctx, wt, tb + ".attn1.to_out.0", ch, ch);
    x.q2 = load_linear(ctx, wt, tb + ".attn2.to_q", ch, ch, false);
    x.kv_col = kvc->cols;
    kvc->prefixes.push_back(tb + ".attn2.to_k"); kvc->outs.push_back(ch);
    kvc->prefixes.push_back(tb + ".attn2.to_v"); kvc->outs.push_back(ch);
    kvc->cols += 2 * ch;
    x.out2 = load_linear(ctx, wt, tb + ".attn2.to_out.0", ch, ch);
    x.ff1 = load_linear(ctx, wt, tb + ".ff.net.0.proj", ch, 8 * ch);
    x.ff2 = load_linear(ctx, wt, tb + ".ff.net.2", 4 * ch, ch);
    x.proj_out = load_conv1x1_as_linear(ctx, wt, p + ".proj_out", ch, ch);
    return x;
}

void unet_load(Ctx* ctx, const WTab& wt, const std::string& pre, const UNetCfg& cfg, UNet* U) {
    U->cfg = cfg;
    ctx->load_dt = U->dt = ctx->dtype;
    const int mc = cfg.model_channels, ted = 4 * mc;
    auto in_attn = [&](int ds) { for (int i = 0; i < cfg.n_attn; ++i) if (cfg.attention_resolutions[i] == ds) return true; return false; };
    U->te0 = load_linear(ctx, wt, pre + "time_embed.0", mc, ted);
    U->te2 = load_linear(ctx, wt, pre + "time_embed.2", ted, ted);
    U->conv_in = load_conv(ctx, wt, pre + "input_blocks.0.0", cfg.in_channels, mc, 3, false);
    std::vector<int> chans{mc};
    KvCollect kvc;
    CatCollect embc;
    int ch = mc, ds = 1, idx = 1;
    for (int level = 0; level < cfg.n_mult; ++level) {
        int mult = cfg.channel_mult[level];
        for (int r = 0; r < cfg.num_res_blocks; ++r) {
            UBlock b;
            b.has_res = true;
            b.res = load_resblock(ctx, wt, pre + S("input_blocks.", idx) + ".0", ch, mult * mc, &embc);
            ch = mult * mc;
            if (in_attn(ds)) { b.has_xf = true; b.xf = load_xfmr(ctx, wt, pre + S("input_blocks.", idx) + ".1", ch, cfg.num_heads, cfg.context_dim, &kvc); }
            U->in_blocks.push_back(b);
            chans.push_back(ch); ++idx;
        }
        if (level != cfg.n_mult - 1) {
            UBlock b;
            b.has_down = true;
            b.down = load_conv(ctx, wt, pre + S("input_blocks.", idx) + ".0.op", ch, ch, 3, true);
            U->in_blocks.push_back(b);
            chans.push_back(ch); ds *= 2; ++idx;
        }
    }
    U->mid1 = load_resblock(ctx, wt, pre + "middle_block.0", ch, ch, &embc);
    U->midx = load_xfmr(ctx, wt, pre + "middle_block.1", ch, cfg.num_heads, cfg.context_dim, &kvc);
    U->mid2 = load_resblock(ctx, wt, pre + "middle_block.2", ch, ch, &embc);
    int oidx = 0;
    for (int level = cfg.n_mult - 1; level >= 0; --level) {
        int mult = cfg.channel_mult[level];
        for (int i = 0; i <= cfg.num_res_blocks; ++i) {
            int ich = chans.back(); chans.pop_back();
            UBlock b;
            const std::string bp = pre + S("output_blocks.", oidx);
            b.has_res = true;
            b.res = load_resblock(ctx, wt, bp + ".0", ch + ich, mc * mult, &embc);
            ch = mc * mult;
            int sub = 1;
            if (in_attn(ds)) { b.has_xf = true; b.xf = load_xfmr(ctx, wt, bp + ".1", ch, cfg.num_heads, cfg.context_dim, &kvc); sub = 2; }
            if (level && i == cfg.num_res_blocks) {
                b.has_up = true;
                b.up = load_conv(ctx, wt, bp + "." + std::to_string(sub) + ".conv", ch, ch, 3, true, true, /*up_dgrad=*/true);
                ds /= 2;
            }
            U->out_blocks.push_back(b);
            ++oidx;
        }
    }
    U->out_norm = load_norm(ctx, wt, pre + "out.0", ch, 1e-5f);
    U->out_conv = load_conv(ctx, wt, pre + "out.2", mc, cfg.out_channels, 3, true);
    U->kv_all = load_linear_cat_var(ctx, wt, kvc.prefixes, cfg.context_dim, kvc.outs);
    // Precision lever of the bf16 step (config 3): the conditioning reaches the UNet only through these 32 projections
    // and its gradient leaves it only through their input gradient, so running that one GEMM pair in fp32 (77 rows: 3 GFLOP per
    // sample) removes the bf16 rounding of the conditioning, of K/V before the cast and of d context after the K = 24960 sum.
    static const bool kv32 = [] { const char* e = getenv("CB_BF16_KV_F32"); return e && e[0] == '1'; }();
    if (U->dt == DT_BF16 && kv32) {
        ctx->load_dt = DT_F32;
        U->kv_all32 = load_linear_cat_var(ctx, wt, kvc.prefixes, cfg.context_dim, kvc.outs);
        ctx->load_dt = U->dt;
    }
    U->emb_all = load_linear_cat_var(ctx, wt, embc.prefixes, ted, embc.outs, true);
}

// openaimodel.py:255-275 ResBlock._forward (use_scale_shift_norm=False, no up/down, dropout 0)
static Ten* resblock(Graph& G, const ResBlockW& r, Ten* x, Ten* emb_all) {
    Ten* h = G.groupnorm(x, r.n1, true);
    h = G.conv(h, r.c1, CONV_SAME, emb_all, nullptr, r.emb_col);   // h + emb_out[..., None, None] fused as a per-image row bias
    h = G.groupnorm(h, r.n2, true);
    Ten* skip = r.has_skip ? G.linear(x, r.skip) : x;
    return G.conv(h, r.c2, CONV_SAME, nullptr, skip);  // skip_connection(x) + h fused as a residual epilogue
}
// All cross-attention keys / values of a UNet pass come from ONE GEMM on the conditioning (U.kv_all: the 32 projections stacked at load time);
// the layers read column views of it. It is issued at the first cross-attention, not at the top of the UNet: everything before it — time
// embedding, conv_in, the first ResBlock, the first block's self-attention — does not read the conditioning, and in the training step the
// conditioning comes from the side stream (FR backbone -> head -> CLIP), which is the LONGER branch of the step's first phase at bs = 1
// (CB_JOIN_PROBE: the context stream sat 0.54 - 0.68 ms in the join when it came ahead of the UNet). Same kernels, same operands: bitwise.
// The backward order does not move either: nothing recorded before the first cross-attention carries a gradient.
struct LazyKV {
    Graph& G; const UNet& U; Ten* context; Ten* context_f32; const std::function<void()>* before;
    Ten* kv = nullptr;
    Ten* get() {
        if (kv) return kv;
        if (before && *before) (*before)();
        kv = (context_f32 && U.kv_all32.w) ? G.cast(G.linear(context_f32, U.kv_all32), U.dt) : G.linear(context, U.kv_all);
        if (kv->rg) {      // zero-filled up front: the 16 layers each accumulate into their own disjoint column block of its gradient
            float beta; void* g = G.grad(kv, &beta);
            k_fill(G.L, kv->dt, g, kv->numel(), 0.f);
        }
        return kv;
    }
};
// attention.py:250-261 SpatialTransformer + 211-215 BasicTransformerBlock
static Ten* xfmr(Graph& G, const XfmrW& w, Ten* x, LazyKV& kv_all) {
    const int ch = w.ch, d = ch / w.heads;
    const float scale = 1.f / sqrtf((float)d);
    Ten* h = G.groupnorm(x, w.norm, false);
    h = G.linear(h, w.proj_in);
    Ten* n1 = G.layernorm(h, w.ln1);
    Ten* qkv = G.linear(n1, w.qkv1);
    Ten* a = G.attention(qkv, 0, qkv, ch, 2 * ch, ch, w.heads, scale, false);
    h = G.linear(a, w.out1, h);
    Ten* n2 = G.layernorm(h, w.ln2);
    Ten* q = G.linear(n2, w.q2);
    a = G.attention(q, 0, kv_all.get(), w.kv_col, w.kv_col + ch, ch, w.heads, scale, false);
    h = G.linear(a, w.out2, h);
    Ten* n3 = G.layernorm(h, w.ln3);
    Ten* f = G.linear(n3, w.ff1);
    Ten* g = G.geglu(f);
    h = G.linear(g, w.ff2, h);
    return G.linear(h, w.proj_out, x);
}

Ten* unet_forward(Graph& G, const UNet& U, Ten* x, const int64_t* t_dev, Ten* context, Ten* context_f32, const std::function<void()>* before_context) {
    const int B = x->n, mc = U.cfg.model_channels;
    CB_CHECK(x->c == U.cfg.in_channels, CB_ERR_SHAPE, "unet: input channels %d vs %d", x->c, U.cfg.in_channels);
    CB_CHECK(context->c == U.cfg.context_dim && context->n == B, CB_ERR_SHAPE, "unet: context shape");
    CB_CHECK(x->dt == U.dt && context->dt == U.dt, CB_ERR_ARG, "unet: inputs must have the element type of the weights");
    Ten* temb = G.make(B, 1, 1, mc, false, U.dt);
    k_timestep_embedding(G.L, U.dt, temb->p, t_dev, B, mc);
    Ten* e = G.linear(temb, U.te0);
    e = G.silu(e);
    Ten* emb = G.linear(e, U.te2);
    Ten* semb = G.linear(G.silu(emb), U.emb_all);      // emb_layers of all ResBlocks: SiLU(emb) is shared, one stacked Linear
    LazyKV kv_all{G, U, context, context_f32, before_context};
    std::vector<Ten*> hs;
    Ten* h = G.conv(x, U.conv_in, CONV_SAME);
    hs.push_back(h);
    int bi = 1;
    for (const UBlock& b : U.in_blocks) {
        if (b.has_res) h = resblock(G, b.res, h, semb);
        if (b.has_xf) h = xfmr(G, b.xf, h, kv_all);
        if (b.has_down) h = G.conv(h, b.down, CONV_DOWN);
        hs.push_back(h);
        G.tap("unet.input_blocks." + std::to_string(bi++), h);
    }
    h = resblock(G, U.mid1, h, semb);
    h = xfmr(G, U.midx, h, kv_all);
    h = resblock(G, U.mid2, h, semb);
    G.tap("unet.middle_block", h);
    bi = 0;
    for (const UBlock& b : U.out_blocks) {
        Ten* skip = hs.back(); hs.pop_back();
        h = G.concat(h, skip);
        h = resblock(G, b.res, h, semb);
        if (b.has_xf) h = xfmr(G, b.xf, h, kv_all);
        if (b.has_up) h = G.conv(h, b.up, CONV_UP);
        G.tap("unet.output_blocks." + std::to_string(bi++), h);
    }
    h = G.groupnorm(h, U.out_norm, true);
    h = G.conv(h, U.out_conv, CONV_SAME);
    G.tap("unet.out", h);
    kv_all.get();          // a UNet without cross-attention still owes its caller the `before_context` call
    return h;
}

// ================================================================= VAE encoder
static VResW load_vres(Ctx* ctx, const WTab& wt, const std::string& p, int cin, int cout) {
    VResW r;
    r.n1 = load_norm(ctx, wt, p + ".norm1", cin, 1e-6f);
    r.c1 = load_conv(ctx, wt, p + ".conv1", cin, cout, 3, false);
    r.n2 = load_norm(ctx, wt, p + ".norm2", cout, 1e-6f);
    r.c2 = load_conv(ctx, wt, p + ".conv2", cout, cout, 3, false);
    if (cin != cout) { r.has_nin = true; r.nin = load_conv1x1_as_linear(ctx, wt, p + ".nin_shortcut", cin, cout); }
    return r;
}
void vae_load(Ctx* ctx, const WTab& wt, const std::string& pre, const VaeCfg& cfg, VaeEnc* V) {
    V->cfg = cfg;
    ctx->load_dt = V->dt = ctx->dtype;
    const std::string e = pre + "encoder.";
    V->conv_in = load_conv(ctx, wt, e + "conv_in", cfg.in_channels, cfg.ch, 3, false);
    int block_in = cfg.ch;
    for (int l = 0; l < cfg.n_mult; ++l) {
        int block_out = cfg.ch * cfg.ch_mult[l];
        std::vector<VResW> blocks;
        for (int b = 0; b < cfg.num_res_blocks; ++b) {
            blocks.push_back(load_vres(ctx, wt, e + S("down.", l) + S(".block.", b), block_in, block_out));
            block_in = block_out;
        }
        V->down.push_back(blocks);
        bool hd = l != cfg.n_mult - 1;
        V->has_down.push_back(hd);
        V->downsample.push_back(hd ? load_conv(ctx, wt, e + S("down.", l) + ".downsample.conv", block_in, block_in, 3, false) : Conv());
    }
    V->mid1 = load_vres(ctx, wt, e + "mid.block_1", block_in, block_in);
    V->attn_norm = load_norm(ctx, wt, e + "mid.attn_1.norm", block_in, 1e-6f);
    V->attn_qkv = load_linear_cat(ctx, wt, {e + "mid.attn_1.q", e + "mid.attn_1.k", e + "mid.attn_1.v"}, block_in, block_in, true, true);
    V->attn_proj = load_conv1x1_as_linear(ctx, wt, e + "mid.attn_1.proj_out", block_in, block_in);
    V->mid2 = load_vres(ctx, wt, e + "mid.block_2", block_in, block_in);
    V->norm_out = load_norm(ctx, wt, e + "norm_out", block_in, 1e-6f);
    V->conv_out = load_conv(ctx, wt, e + "conv_out", block_in, 2 * cfg.z_channels, 3, false);
    V->quant = load_conv1x1_as_linear(ctx, wt, pre + "quant_conv", 2 * cfg.z_channels, 2 * cfg.z_channels);
}
static Ten* vres(Graph& G, const VResW& r, Ten* x) {   // model.py:121-141, temb is None
    Ten* h = G.groupnorm(x, r.n1, true);
    h = G.conv(h, r.c1, CONV_SAME);
    h = G.groupnorm(h, r.n2, true);
    Ten* skip = r.has_nin ? G.linear(x, r.nin) : x;
    return G.conv(h, r.c2, CONV_SAME, nullptr, skip);
}
Ten* vae_encode_moments(Graph& G, const VaeEnc& V, Ten* img) {
    Ten* h = G.conv(img, V.conv_in, CONV_SAME);
    for (size_t l = 0; l < V.down.size(); ++l) {
        for (const VResW& r : V.down[l]) h = vres(G, r, h);
        if (V.has_down[l]) h = G.conv(h, V.downsample[l], CONV_DOWN_ASYM);   // pad (0,1,0,1) + stride 2, model.py:74-76
        G.tap("vae.down." + std::to_string(l), h);
    }
    h = vres(G, V.mid1, h);
    {   // AttnBlock model.py:176-202: single head, scale C^-0.5
        const int C = h->c;
        Ten* n = G.groupnorm(h, V.attn_norm, false);
        Ten* qkv = G.linear(n, V.attn_qkv);
        Ten* a = G.attention(qkv, 0, qkv, C, 2 * C, C, 1, 1.f / sqrtf((float)C), false);
        h = G.linear(a, V.attn_proj, h);
    }
    h = vres(G, V.mid2, h);
    G.tap("vae.mid", h);
    h = G.groupnorm(h, V.norm_out, true);
    h = G.conv(h, V.conv_out, CONV_SAME);
    h = G.linear(h, V.quant);
    G.tap("vae.moments", h);
    return h;
}

// ================================================================= VAE decoder (sampling path)
void vae_dec_load(Ctx* ctx, const WTab& wt, const std::string& pre, const VaeCfg& cfg, VaeDec* V) {
    V->cfg = cfg;
    ctx->load_dt = V->dt = ctx->dtype;
    const std::string d = pre + "decoder.";
    V->post_quant = load_conv1x1_as_linear(ctx, wt, pre + "post_quant_conv", cfg.z_channels, cfg.z_channels);
    int block_in = cfg.ch * cfg.ch_mult[cfg.n_mult - 1];
    V->conv_in = load_conv(ctx, wt, d + "conv_in", cfg.z_channels, block_in, 3, false);
    V->mid1 = load_vres(ctx, wt, d + "mid.block_1", block_in, block_in);
    V->attn_norm = load_norm(ctx, wt, d + "mid.attn_1.norm", block_in, 1e-6f);
    V->attn_qkv = load_linear_cat(ctx, wt, {d + "mid.attn_1.q", d + "mid.attn_1.k", d + "mid.attn_1.v"}, block_in, block_in, true, true);
    V->attn_proj = load_conv1x1_as_linear(ctx, wt, d + "mid.attn_1.proj_out", block_in, block_in);
    V->mid2 = load_vres(ctx, wt, d + "mid.block_2", block_in, block_in);
    V->up.resize(cfg.n_mult); V->upsample.resize(cfg.n_mult); V->has_up.assign(cfg.n_mult, false);
    for (int l = cfg.n_mult - 1; l >= 0; --l) {            // model.py:507-526: built from the coarsest level down, num_res_blocks+1 each
        int block_out = cfg.ch * cfg.ch_mult[l];
        for (int b = 0; b < cfg.num_res_blocks + 1; ++b) {
            V->up[l].push_back(load_vres(ctx, wt, d + S("up.", l) + S(".block.", b), block_in, block_out));
            block_in = block_out;
        }
        if (l != 0) { V->has_up[l] = true; V->upsample[l] = load_conv(ctx, wt, d + S("up.", l) + ".upsample.conv", block_in, block_in, 3, false); }
    }
    V->norm_out = load_norm(ctx, wt, d + "norm_out", block_in, 1e-6f);
    V->conv_out = load_conv(ctx, wt, d + "conv_out", block_in, cfg.in_channels, 3, false);
}
Ten* vae_decode(Graph& G, const VaeDec& V, Ten* z) {
    Ten* h = G.linear(z, V.post_quant);                    // autoencoder.py:331
    h = G.conv(h, V.conv_in, CONV_SAME);
    h = vres(G, V.mid1, h);
    {   // AttnBlock model.py:176-202
        const int C = h->c;
        Ten* n = G.groupnorm(h, V.attn_norm, false);
        Ten* qkv = G.linear(n, V.attn_qkv);
        Ten* a = G.attention(qkv, 0, qkv, C, 2 * C, C, 1, 1.f / sqrtf((float)C), false);
        h = G.linear(a, V.attn_proj, h);
    }
    h = vres(G, V.mid2, h);
    G.tap("vaedec.mid", h);
    for (int l = V.cfg.n_mult - 1; l >= 0; --l) {
        for (const VResW& r : V.up[l]) h = vres(G, r, h);
        if (V.has_up[l]) h = G.conv(h, V.upsample[l], CONV_UP);   // nearest x2 + conv3x3, model.py:42-57
        G.tap("vaedec.up." + std::to_string(l), h);
    }
    h = G.groupnorm(h, V.norm_out, true);
    return G.conv(h, V.conv_out, CONV_SAME);
}

// ================================================================= CLIP text transformer
void clip_load(Ctx* ctx, const WTab& wt, const std::string& pre, const ClipCfg& cfg, Clip* C) {
    C->cfg = cfg;
    // The text transformer sits on the COEFFICIENT side of the step: the gradient of the loss reaches the two pseudo-word rows
    // through its backward. Config 3 is "bf16 UNet + fp32 coefficient gradient" (BASELINE.json), so it stays fp32 in both modes
    // (26 GFLOP of 2.9 TFLOP per sample, on the side stream under the VAE encode); its output is cast once for the UNet.
    ctx->load_dt = C->dt = DT_F32;
    const int w = cfg.width;
    C->tok_emb = ctx->upload(wt, pre + "embeddings.token_embedding.weight", {cfg.vocab, w});
    C->pos_emb = ctx->upload(wt, pre + "embeddings.position_embedding.weight", {cfg.ctx_len, w});
    for (int i = 0; i < cfg.layers; ++i) {
        const std::string p = pre + S("encoder.layers.", i);
        ClipLayerW l;
        l.ln1 = load_norm(ctx, wt, p + ".layer_norm1", w, 1e-5f);
        l.ln2 = load_norm(ctx, wt, p + ".layer_norm2", w, 1e-5f);
        l.qkv = load_linear_cat(ctx, wt, {p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, w, w, true);
        l.out = load_linear(ctx, wt, p + ".self_attn.out_proj", w, w);
        l.fc1 = load_linear(ctx, wt, p + ".mlp.fc1", w, cfg.mlp);
        l.fc2 = load_linear(ctx, wt, p + ".mlp.fc2", cfg.mlp, w);
        C->layers.push_back(l);
    }
    C->final_ln = load_norm(ctx, wt, pre + "final_layer_norm", w, 1e-5f);
}
Ten* clip_encode(Graph& G, const Clip& C, Ten* embedded) {
    const int w = C.cfg.width, L = embedded->h * embedded->w;
    CB_CHECK(embedded->c == w && L <= C.cfg.ctx_len, CB_ERR_SHAPE, "clip: embedded shape");
    // embeddings = inputs_embeds + position_embedding (modules.py:295-296)
    embedded = G.cast(embedded, C.dt);          // the manager hook and its gradient are fp32; the transformer runs in C.dt
    Ten* x = G.make(embedded->n, embedded->h, embedded->w, w, embedded->rg, C.dt);
    k_add_rowbcast(G.L, C.dt, x->p, embedded->p, C.pos_emb, embedded->rows(), L, w);
    if (embedded->rg) {
        Graph* g = &G;
        G.tape.push_back([g, x, embedded]() {
            if (!x->ginit) return;
            float beta; void* ge = g->grad(embedded, &beta);
            k_axpby(g->L, x->dt, ge, x->g, x->numel(), 1.f, beta);
        });
    }
    const float scale = 1.f / sqrtf((float)(w / C.cfg.heads));
    for (const ClipLayerW& l : C.layers) {       // HF CLIPEncoderLayer: pre-LN, causal self-attention, quick-GELU MLP
        Ten* h = G.layernorm(x, l.ln1);
        Ten* qkv = G.linear(h, l.qkv);
        Ten* a = G.attention(qkv, 0, qkv, w, 2 * w, w, C.cfg.heads, scale, true);
        x = G.linear(a, l.out, x);
        h = G.layernorm(x, l.ln2);
        h = G.linear(h, l.fc1);
        h = G.quickgelu(h);
        x = G.linear(h, l.fc2, x);
    }
    x = G.layernorm(x, C.final_ln);
    G.tap("clip.last_hidden_state", x);
    return x;
}

// ================================================================= iresnet (frozen, eval)
static void bn_affine(Ctx* ctx, const WTab& wt, const std::string& p, int c, float** s, float** b) {
    float* w = ctx->upload(wt, p + ".weight", {c});
    float* bb = ctx->upload(wt, p + ".bias", {c});
    float* m = ctx->upload(wt, p + ".running_mean", {c});
    float* v = ctx->upload(wt, p + ".running_var", {c});
    *s = ctx->pool.allocf(c); *b = ctx->pool.allocf(c);
    k_bn_to_affine(ctx->launch(), *s, *b, w, bb, m, v, 1e-5f, c);
}
// conv (no bias) followed by an eval BatchNorm: fold the BN into the packed weights + a bias
static Conv load_conv_bn(Ctx* ctx, const WTab& wt, const std::string& conv, const std::string& bn, int cin, int cout, int ks) {
    Conv c = load_conv(ctx, wt, conv, cin, cout, ks, false, false);
    float* w = ctx->upload(wt, bn + ".weight", {cout});
    float* bb = ctx->upload(wt, bn + ".bias", {cout});
    float* m = ctx->upload(wt, bn + ".running_mean", {cout});
    float* v = ctx->upload(wt, bn + ".running_var", {cout});
    c.b = ctx->pool.allocf(cout);
    k_fold_bn(ctx->launch(), static_cast<float*>(c.w), c.b, w, bb, m, v, 1e-5f, cout, (long)ks * ks * cin);
    return c;
}
void frnet_load(Ctx* ctx, const WTab& wt, const std::string& pre, FrNet* F, const int* layers, int width, int in_size) {
    ctx->load_dt = DT_F32;      // the face-recognition backbone feeds the fp32 coefficient head: always fp32 (SURVEY §8 d, config 3)
    for (int i = 0; i < 4; ++i) F->layers[i] = layers[i];
    F->width = width; F->in_size = in_size;
    F->conv1 = load_conv_bn(ctx, wt, pre + "conv1", pre + "bn1", 3, width, 3);
    F->prelu1 = ctx->upload(wt, pre + "prelu.weight", {width});
    int inpl = width;
    for (int l = 0; l < 4; ++l) {
        int planes = width << l;
        for (int i = 0; i < layers[l]; ++i) {
            const std::string p = pre + S("layer", l + 1) + S(".", i);
            IBlockW b;
            b.stride = (i == 0) ? 2 : 1;
            bn_affine(ctx, wt, p + ".bn1", inpl, &b.bn1_s, &b.bn1_b);
            b.c1 = load_conv_bn(ctx, wt, p + ".conv1", p + ".bn2", inpl, planes, 3);
            b.prelu = ctx->upload(wt, p + ".prelu.weight", {planes});
            b.c2 = load_conv_bn(ctx, wt, p + ".conv2", p + ".bn3", planes, planes, 3);
            if (i == 0) { b.has_ds = true; b.ds = load_conv_bn(ctx, wt, p + ".downsample.0", p + ".downsample.1", inpl, planes, 1); }
            F->blocks.push_back(b);
            inpl = planes;
        }
    }
    const int C = width * 8;
    bn_affine(ctx, wt, pre + "bn2", C, &F->bn2_s, &F->bn2_b);
    int fs = in_size / 16;   // four stride-2 stages
    const int HW = fs * fs;
    F->feat = (int)wt.get(pre + "fc.weight").dims[0];
    float* fcw = ctx->upload(wt, pre + "fc.weight", {F->feat, (int64_t)C * HW});
    float* fcb = ctx->upload(wt, pre + "fc.bias", {F->feat});
    F->fc.in = C * HW; F->fc.out = F->feat;
    float* fcp = ctx->pool.allocf((size_t)F->feat * C * HW);
    F->fc.w = fcp;
    k_pack_fc_nhwc(ctx->launch(), fcp, fcw, F->feat, C, HW);
    // features = BatchNorm1d after fc (iresnet.py:180): y = (W x + b) * s + t
    float *fs_, *ft_;
    bn_affine(ctx, wt, pre + "features", F->feat, &fs_, &ft_);
    float* dummy = ctx->pool.allocf(F->feat);
    {   // scale rows of W by s (fold_bn with mean=0,var=1-eps would be clumsy: reuse channel math directly)
        float* w = ctx->upload(wt, pre + "features.weight", {F->feat});
        float* bb = ctx->upload(wt, pre + "features.bias", {F->feat});
        float* m = ctx->upload(wt, pre + "features.running_mean", {F->feat});
        float* v = ctx->upload(wt, pre + "features.running_var", {F->feat});
        k_fold_bn(ctx->launch(), fcp, dummy, w, bb, m, v, 1e-5f, F->feat, (long)C * HW);
    }
    F->fc.b = ctx->pool.allocf(F->feat);
    k_channel_affine(ctx->launch(), F->fc.b, fcb, fs_, ft_, nullptr, 1, F->feat);   // b*s + t
}
Ten* frnet_forward(Graph& G, const FrNet& F, Ten* img) {
    // iresnet.py:52-63 IBasicBlock: x -> bn1 -> conv1 -> bn2 -> PReLU -> conv2 -> bn3, + identity. bn2 / bn3 are folded into the conv
    // weights at load; the PReLU runs in conv1's epilogue; bn1 cannot be folded (the conv pads bn1(x) with zeros, not bn1(0)), so the
    // conv2 that PRODUCES x also writes the next block's bn1(x) as a second output. Two channel_affine launches are left (100 before).
    Ten* x = G.conv(img, F.conv1, CONV_SAME);                         // 3-channel gather: register-staged kernel, no fused epilogue
    x = G.channel_affine(x, nullptr, nullptr, F.prelu1);
    Ten* o = F.blocks.empty() ? nullptr : G.channel_affine(x, F.blocks[0].bn1_s, F.blocks[0].bn1_b, nullptr);
    for (size_t i = 0; i < F.blocks.size(); ++i) {
        const IBlockW& b = F.blocks[i];
        Graph::ConvEpi e1; e1.prelu = b.prelu;
        Ten* h = G.conv(o, b.c1, CONV_SAME, nullptr, nullptr, 0, &e1);
        Ten* idn = b.has_ds ? G.conv(x, b.ds, CONV_S2_1x1) : x;
        Graph::ConvEpi e2;                                             // next block's bn1, or the backbone's final bn2
        const bool last = i + 1 == F.blocks.size();
        e2.c2_scale = last ? F.bn2_s : F.blocks[i + 1].bn1_s; e2.c2_shift = last ? F.bn2_b : F.blocks[i + 1].bn1_b; e2.c2_out = &o;
        x = G.conv(h, b.c2, b.stride == 2 ? CONV_DOWN : CONV_SAME, nullptr, idn, 0, &e2);
    }
    x = F.blocks.empty() ? G.channel_affine(x, F.bn2_s, F.bn2_b, nullptr) : o;
    Ten* flat = G.wrap(x->p, x->n, 1, 1, x->h * x->w * x->c, false, DT_F32);
    Ten* f = G.linear(flat, F.fc);
    G.tap("frnet.features", f);
    return f;
}

// ================================================================= training step
void trainer_load(Ctx* ctx, const WTab& wt, const TrainerCfg& cfg, Trainer* T) {
    T->ctx = ctx; T->cfg = cfg;
    { const char* e = getenv("CB_STEP_GRAPH"); T->graph.enabled = e && e[0] == '1'; }
    { const char* e = getenv("CB_TAIL_OVERLAP"); T->tail_overlap = !(e && e[0] == '0'); }          // default on; cb_trainer_set_tail_overlap      // opt-in: measured flat on MI355X (profiles/README.md, round 5)
    unet_load(ctx, wt, "model.diffusion_model.", cfg.unet, &T->unet);
    vae_load(ctx, wt, "first_stage_model.", cfg.vae, &T->vae);
    clip_load(ctx, wt, "cond_stage_model.transformer.text_model.", cfg.clip, &T->clip);
    frnet_load(ctx, wt, "embedding_manager.meta_id_net.id_model.", &T->fr, cfg.fr_layers, cfg.fr_width, cfg.fr_in);
    const int w = cfg.clip.width, O = cfg.es * cfg.inner_dim, I = T->fr.feat;
    T->basis = ctx->upload(wt, "celeb_embeddings", {cfg.es, 1 + cfg.inner_dim, w});
    T->W = ctx->upload(wt, "embedding_manager.meta_id_net.stylegan_mlp.net.0.weight", {O, I});
    T->b = ctx->upload(wt, "embedding_manager.meta_id_net.stylegan_mlp.net.0.bias", {O});
    Launch L = ctx->launch();
    auto zeros = [&](size_t n) { float* p = ctx->pool.allocf(n); k_fill(L, p, (long)n, 0.f); return p; };
    T->mW = zeros((size_t)O * I); T->vW = zeros((size_t)O * I); T->gW = zeros((size_t)O * I);
    T->mb = zeros(O); T->vb = zeros(O); T->gb = zeros(O);
    if (cfg.use_rm_mlp) {
        const size_t n = (size_t)cfg.max_ids * O;
        T->coef = ctx->upload(wt, "embedding_manager.meta_id_net.coef", {cfg.max_ids, cfg.es, 1, cfg.inner_dim});
        T->mC = zeros(n); T->vC = zeros(n); T->gC = zeros(n);
    }
    T->step = reinterpret_cast<int*>(zeros(64));
    T->id_emb = wt.has("id_embeddings") ? ctx->upload(wt, "id_embeddings", {cfg.max_ids, cfg.es * w}) : zeros((size_t)cfg.max_ids * cfg.es * w);
    T->id_coef = wt.has("id_coefficients") ? ctx->upload(wt, "id_coefficients", {cfg.max_ids, cfg.es * cfg.inner_dim})
                                           : zeros((size_t)cfg.max_ids * cfg.es * cfg.inner_dim);
    T->loss = zeros(64);
    T->lr_dev = zeros(64);
    // ddpm.py:126-178 register_schedule with util.py:21-25 "linear" (sqrt-space linspace), float64 -> float32
    std::vector<float> sa(cfg.timesteps), sb(cfg.timesteps);
    double s0 = std::sqrt((double)cfg.linear_start), s1 = std::sqrt((double)cfg.linear_end), ac = 1.0;
    for (int i = 0; i < cfg.timesteps; ++i) {
        double r = cfg.timesteps > 1 ? s0 + (s1 - s0) * (double)i / (double)(cfg.timesteps - 1) : s0;
        double beta = r * r;
        ac *= (1.0 - beta);
        sa[i] = (float)std::sqrt(ac);
        sb[i] = (float)std::sqrt(1.0 - ac);
    }
    T->sqrt_ac = ctx->pool.allocf(cfg.timesteps); T->sqrt_1mac = ctx->pool.allocf(cfg.timesteps);
    if (!ctx->dry) {
        CB_HIP(hipMemcpy(T->sqrt_ac, sa.data(), sa.size() * sizeof(float), hipMemcpyHostToDevice));
        CB_HIP(hipMemcpy(T->sqrt_1mac, sb.data(), sb.size() * sizeof(float), hipMemcpyHostToDevice));
    }
}

constexpr int kMaxBatch = 16;
static const float kTheta[6] = {1.07695457f, -0.03625215f, -1.56352194f / 512.f,     // meta_net.py:131-142 trans_matrix
                                0.03625215f, 1.07695457f, -5.32134629f / 512.f};

// phase boundaries for the live profiler (cb_ctx_profile_begin/end): one HIP-event pair per phase of the step
struct Phase {
    const Launch& L; ProfRec rec; bool on;
    Phase(const Launch& l, const char* name) : L(l) { on = !L.dry && L.prof && L.prof->begin(&rec, name, 0.0, 0.0, L.stream); }
    ~Phase() { if (on) L.prof->end(&rec, L.stream); }
};

// first stage, part A: warp + resize + iresnet on both face chunks (meta_net.py:250-262), no grad -> raw features [2B, 512]
static Ten* first_stage_faces(Graph& G, Trainer* T, const Batch& bt) {
    const TrainerCfg& c = T->cfg;
    const int NF = 2 * bt.B, S_ = c.image_size;
    Ten* warped = G.make(NF, S_, S_, 3, false, DT_F32);
    k_affine_warp(G.L, warped->pf(), bt.faces, bt.B, S_, 2, kTheta, T->ctx->warp_grid_for(S_));
    Ten* small = G.make(NF, c.fr_in, c.fr_in, 3, false, DT_F32);
    k_resize_bilinear_ac(G.L, small->pf(), warped->pf(), NF, S_, S_, c.fr_in, c.fr_in, 3);
    return frnet_forward(G, T->fr, small);
}
// first stage, part B: VAE encode -> posterior sample * scale_factor (ddpm.py:702-759, 590-597), no grad -> z0 [B,h,w,zc]
static Ten* first_stage_latent(Graph& G, Trainer* T, const Batch& bt) {
    const TrainerCfg& c = T->cfg;
    Ten* img = G.cast(G.wrap(bt.image, bt.B, c.image_size, c.image_size, c.vae.in_channels, false, DT_F32), T->vae.dt);
    Ten* mom = vae_encode_moments(G, T->vae, img);
    Ten* z0 = G.make(bt.B, mom->h, mom->w, c.vae.z_channels, false, mom->dt);
    k_vae_sample(G.L, mom->dt, z0->p, mom->p, bt.post_noise, bt.B, mom->h * mom->w, c.vae.z_channels, c.scale_factor);
    return z0;
}

static void prefetch_impl(Trainer* T, const Batch& bt, Trainer::Prefetch::Slot& sl) {
    Ctx* ctx = T->ctx;
    Graph G(ctx, /*join_tail=*/false);          // own stream, own arena, frozen weights only: a deferred step tail is no business of the prefetch
    G.L.stream = T->pf.stream;
    if (!T->cfg.use_rm_mlp) {
        Ten* feat = first_stage_faces(G, T, bt);
        k_axpby(G.L, sl.feat, feat->pf(), feat->numel(), 1.f, 0.f);
    }
    Ten* z0 = first_stage_latent(G, T, bt);
    k_axpby(G.L, z0->dt, sl.z0, z0->p, z0->numel(), 1.f, 0.f);
}
hipStream_t trainer_prefetch_stream(Trainer* T) {
    Ctx* ctx = T->ctx;
    Trainer::Prefetch& pf = T->pf;
    const TrainerCfg& c = T->cfg;
    if (ctx->dry) return nullptr;
    if (!pf.ready) {
        if (!pf.stream) { CB_HIP(hipStreamCreateWithFlags(&pf.stream, hipStreamNonBlocking)); pf.own_stream = true; }
        CB_HIP(hipEventCreateWithFlags(&pf.ready, hipEventDisableTiming));
        const int lat = c.image_size / (1 << (c.vae.n_mult - 1));
        for (auto& sl : pf.slot) {
            CB_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
            sl.z0 = ctx->pool.allocf((size_t)kMaxBatch * lat * lat * c.vae.z_channels);      // sized for the largest batch
            sl.feat = ctx->pool.allocf((size_t)2 * kMaxBatch * T->fr.feat);
        }
    }
    return pf.stream;
}

void trainer_set_prefetch_stream(Trainer* T, hipStream_t s) {
    Trainer::Prefetch& pf = T->pf;
    CB_CHECK(s != nullptr, CB_ERR_ARG, "prefetch stream: null");
    CB_CHECK(!pf.ready, CB_ERR_STATE, "prefetch stream: must be set before the first cb_trainer_prefetch");
    pf.stream = s; pf.own_stream = false;
}

void trainer_prefetch(Trainer* T, const Batch& bt) {
    Ctx* ctx = T->ctx;
    CB_CHECK(bt.B >= 1 && bt.B <= kMaxBatch, CB_ERR_ARG, "trainer: batch %d unsupported (1..%d)", bt.B, kMaxBatch);
    Trainer::Prefetch& pf = T->pf;
    const TrainerCfg& c = T->cfg;
    if (ctx->dry) return;
    trainer_prefetch_stream(T);
    Trainer::Prefetch::Slot& sl = pf.slot[pf.next];
    pf.next ^= 1;
    sl.valid = false;
    // the prefetch graph allocates from its own arena: the step arena is rewound while these kernels may still be running
    std::swap(ctx->arena, pf.arena);
    try {
        if (pf.planned_B != bt.B) {
            Stats saved = ctx->stats;
            ctx->dry = true; ctx->arena.dry = true; ctx->arena.reset(); ctx->arena.peak = 0;
            try { prefetch_impl(T, bt, sl); } catch (...) { ctx->dry = false; ctx->arena.dry = false; ctx->stats = saved; throw; }
            pf.need = ctx->arena.peak;
            ctx->dry = false; ctx->arena.dry = false; ctx->stats = saved;
            CB_HIP(hipStreamSynchronize(pf.stream));
            ctx->ensure_arena(pf.need);
            pf.planned_B = bt.B;
        }
        ctx->arena.reset();
        // everything enqueued on the caller's stream so far — the step that consumed this slot two prefetches ago and whatever
        // produced the batch buffers — happens before the prefetch; the step enqueued after this call overlaps with it
        CB_HIP(hipEventRecord(pf.ready, ctx->stream));
        CB_HIP(hipStreamWaitEvent(pf.stream, pf.ready, 0));
        prefetch_impl(T, bt, sl);
        CB_HIP(hipEventRecord(sl.done, pf.stream));
    } catch (...) { std::swap(ctx->arena, pf.arena); throw; }
    std::swap(ctx->arena, pf.arena);
    sl.image = bt.image; sl.faces = bt.faces; sl.post_noise = bt.post_noise; sl.B = bt.B; sl.valid = true;
}

static void step_impl(Trainer* T, const Batch& bt, float lr) {
    Ctx* ctx = T->ctx;
    const TrainerCfg& c = T->cfg;
    // no join of the previous step's deferred tail here: it runs on its own stream; this step's VAE encode (context stream) and FR
    // backbone (side stream) touch neither its arena cells nor the trainable tensors, the head below waits for it (see the end of this function)
    Graph G(ctx, /*join_tail=*/false);
    const Launch& L = G.L;
    const int B = bt.B, S_ = c.image_size, w = c.clip.width, es = c.es, K = c.inner_dim, T77 = c.clip.ctx_len;
    const int zc = c.vae.z_channels;

    // (2) identity branch: warp + resize + iresnet (no grad) -> MLP -> L2-norm -> basis   (meta_net.py:250-289)
    // It is independent of the VAE encode, so it is enqueued on the library's side stream and overlaps with (1): the ~250 small
    // FR-net launches fill the tails of the big VAE convolutions instead of serialising 3 ms behind them.
    const int NF = 2 * B;
    const bool fork = !ctx->dry && ctx->aux != nullptr;
    hipStream_t main_stream = G.L.stream;
    // a batch whose first stage was enqueued by trainer_prefetch: wait for it instead of recomputing (bitwise the same values)
    Trainer::Prefetch::Slot* ps = nullptr;
    if (!ctx->dry)
        for (auto& sl : T->pf.slot)
            if (sl.valid && sl.image == bt.image && sl.faces == bt.faces && sl.post_noise == bt.post_noise && sl.B == B) ps = &sl;
    const bool prefetched = ps != nullptr;
    // The previous step's deferred tail may stay in flight only under a step with the SAME arena layout (same batch size — checked by
    // trainer_step — and the same first-stage form): a prefetched step allocates neither the FR backbone's nor the VAE's activations, so
    // everything behind them sits at other offsets, and this step's first stage would write where that tail still reads.
    const int layout = 2 * B + (prefetched ? 1 : 0);
    if (ctx->tail_pending && ctx->tail_layout != layout) ctx->join_tail();
    if (fork) {
        CB_HIP(hipEventRecord(ctx->ev_fork, main_stream));        // everything enqueued so far (incl. the previous step) ...
        CB_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));    // ... happens before the side branch
        G.L.stream = ctx->aux;
    }
    if (prefetched) {
        CB_HIP(hipStreamWaitEvent(G.L.stream, ps->done, 0)); CB_HIP(hipStreamWaitEvent(main_stream, ps->done, 0));
        ps->valid = false;                              // consumed; the slot is rewritten no earlier than the prefetch after next
    }
    std::unique_ptr<Phase> ph_id(new Phase(L, "phase:frnet_head"));
    const bool rm = c.use_rm_mlp != 0;
    const int I = T->fr.feat, O = es * K;
    Ten *v = nullptr, *pre = nullptr;
    Ten* xn = G.make(NF * es, 1, 1, K, false, DT_F32);    // 'b (e h d) -> b e h d', h = 1
    float* cn = ctx->arena.allocf((size_t)NF * es);
    if (rm) {
        // meta_net.py:272-273 x = self.coef[id_idx], id_idx = flatten(ids) over the chunk-major rows (forward_multi_faces :338-340)
        k_coef_gather(L, xn->pf(), T->coef, bt.ids, 2, B, 2, O, c.max_ids);
    } else {
        Ten* feat = prefetched ? G.wrap(ps->feat, NF, 1, 1, T->fr.feat, false, DT_F32) : first_stage_faces(G, T, bt);
        // the previous step's deferred tail (third stream) ends in AdamW on W / b, and it still reads the arena cells that are rewritten
        // from here on (xn, cn, v, pre, the CLIP activations); the FR backbone above touched neither
        if (fork && ctx->tail_pending) CB_HIP(hipStreamWaitEvent(G.L.stream, ctx->ev_tail, 0));
        v = G.make(NF, 1, 1, I, false, DT_F32);
        k_l2norm_fwd(L, v->pf(), nullptr, feat->pf(), NF, I);
        Lin head; head.w = T->W; head.b = T->b; head.in = I; head.out = O;
        pre = G.linear(v, head);                              // EqualLinear with lr_mul = 1 (meta_net.py:41-48)
        Ten* act = G.make(NF, 1, 1, O, false, DT_F32);
        k_leakyrelu_fwd(L, act->pf(), pre->pf(), act->numel(), 0.2f);
        k_l2norm_fwd(L, xn->pf(), cn, act->pf(), (long)NF * es, K);
    }
    Ten* zz = G.make(NF, es, 1, w, false, DT_F32);
    k_basis_fwd(L, zz->pf(), xn->pf(), T->basis, NF, es, K, w);
    G.tap("coef", xn); G.tap("zz", zz);
    // momentum dictionaries (embedding_manager.py:484-489) use the first face chunk (meta1 / cef1)
    k_ema_update(L, T->id_emb, T->id_coef, zz->pf(), xn->pf(), bt.ids, 2, B, es * w, es * K, c.max_ids, c.momentum);
    ph_id.reset();

    // (3) conditioning: token embedding -> manager hook (row surgery) -> CLIP   (modules.py:176-298, 345-404). It needs the
    // identity embedding but not the latent, so it stays on the side stream: ~200 M=77 launches hidden under the VAE encode.
    std::unique_ptr<Phase> ph_clip(new Phase(L, "phase:clip_fwd"));
    Ten* emb0 = G.make(B, T77, 1, w, false, DT_F32);
    k_gather_rows(L, emb0->pf(), T->clip.tok_emb, bt.tokens, (long)B * T77, w, c.clip.vocab);
    Ten* emb1 = G.make(B, T77, 1, w, true, DT_F32);
    int32_t* map = static_cast<int32_t*>(ctx->arena.alloc((size_t)B * T77 * sizeof(int32_t)));
    k_row_surgery_fwd(L, emb1->pf(), emb0->pf(), zz->pf(), bt.tokens, map, B, T77, w, es, c.placeholder_token);
    Ten* cond32 = clip_encode(G, T->clip, emb1);
    Ten* cond = G.cast(cond32, T->unet.dt);
    ph_clip.reset();
    const size_t tape_mark = G.tape.size();          // closures below this index: conditioning -> CLIP -> pseudo-word rows (the step's tail)
    const size_t id_region_end = ctx->arena.off;     // arena [0, id_region_end): identity branch + CLIP activations (what the tail reads)

    if (fork) { CB_HIP(hipEventRecord(ctx->ev_join, ctx->aux)); G.L.stream = main_stream; }

    // (1) first stage: VAE encode -> posterior sample * scale_factor   (ddpm.py:702-759, 590-597) — no grad, main stream
    Ten* z0;
    {
        Phase ph(L, "phase:vae_encode");
        z0 = prefetched ? G.wrap(ps->z0, B, S_ >> (c.vae.n_mult - 1), S_ >> (c.vae.n_mult - 1), zc, false, T->vae.dt) : first_stage_latent(G, T, bt);
        G.tap("z0", z0);
    }

    const size_t vae_region_end = ctx->arena.off;    // arena [id_region_end, vae_region_end): the first stage (context stream, before the join)
    // join: the UNet needs the conditioning — from its first cross-attention on (unet_forward calls `join` there; what runs before it reads
    // frozen weights, z0 and the caller's noise / t, and writes the arena cells right above the first stage's: nothing the previous step's
    // deferred tail touches). ev_join was recorded on the side stream AFTER its wait for that tail, so this wait also orders the tail before
    // everything enqueued behind it.
    // CB_JOIN_PROBE=1 (diagnostics; blocks the host once per step) prints how long the context stream sits in the wait.
    static const bool join_probe = getenv("CB_JOIN_PROBE") != nullptr;
    static hipEvent_t jp0 = nullptr, jp1 = nullptr;
    bool joined = false;
    const std::function<void()> join = [&]() {
        if (!fork || joined) return;
        joined = true;
        const bool probe = join_probe && !T->capturing;
        if (probe) {
            if (!jp0) { CB_HIP(hipEventCreate(&jp0)); CB_HIP(hipEventCreate(&jp1)); }
            CB_HIP(hipEventRecord(jp0, main_stream));
        }
        CB_HIP(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
        ctx->tail_pending = false;
        if (probe) {
            CB_HIP(hipEventRecord(jp1, main_stream));
            CB_HIP(hipEventSynchronize(jp1));
            float ms = 0.f;
            CB_HIP(hipEventElapsedTime(&ms, jp0, jp1));
            fprintf(stderr, "[cb join probe] context stream waited %.3f ms for the side branch\n", ms);
        }
    };
    static const bool join_early = [] { const char* e = getenv("CB_JOIN_EARLY"); return e && e[0] == '1'; }();      // A/B: the round-5 position
    if (join_early) join();

    // (4-6) q_sample, UNet, loss   (ddpm.py:926-936, 1069-1116)
    std::unique_ptr<Phase> ph_uf(new Phase(L, "phase:unet_fwd"));
    Ten* xt = G.make(B, z0->h, z0->w, zc, false, z0->dt);
    k_qsample(L, z0->dt, xt->p, z0->p, bt.noise, bt.t, T->sqrt_ac, T->sqrt_1mac, B, zc, z0->h * z0->w);
    Ten* eps_hat = unet_forward(G, T->unet, xt, bt.t, cond, cond32->dt == DT_F32 && cond32 != cond ? cond32 : nullptr, &join);
    join();
    float beta;
    void* deps = G.grad(eps_hat, &beta);
    k_mse_loss(L, eps_hat->dt, T->loss, deps, eps_hat->p, bt.noise, B, zc, z0->h * z0->w);
    ph_uf.reset();

    // (7) backward: UNet dgrad -> context -> CLIP dgrad -> rows p, p+1 -> basis^T -> norm -> LeakyReLU -> dW, db
    // The tail — everything after d context: ~190 launches of 77-row problems, the head's backward, AdamW — reads only the identity /
    // CLIP region of the arena and what the backward allocates above the UNet's, writes only trainer-owned state, and is needed by
    // nothing before the NEXT step's head (FR-net -> HEAD -> CLIP forward: the head reads the updated W). It is therefore enqueued on a
    // third stream and NOT joined: the next step's VAE encode (context stream; arena region [id_region_end, vae_region_end), frozen
    // weights, the caller's image) and FR backbone (side stream; its own activations, frozen weights, the caller's faces) run beside it,
    // and the side stream waits for it right before the head. (On the side stream itself it bought nothing: FR-net + CLIP forward already
    // take as long as the VAE encode, so a tail queued in front of them only moved the critical path there.) Whoever else touches the
    // context joins first (Ctx::join_tail). Off under the profiler / debug taps / graph capture / use_rm_mlp (whose tail reads the
    // caller's `ids`) and with CB_TAIL_OVERLAP=0: the tail then runs on the context stream as before. Same kernels, same order: bitwise.
    const bool defer_tail = fork && T->tail_overlap && !c.use_rm_mlp && !ctx->debug_taps && !ctx->prof.on && !T->capturing;
    {
        Phase ph(L, "phase:backward_unet_clip");
        if (defer_tail) {
            G.backward_until(tape_mark);
            CB_CHECK(id_region_end <= vae_region_end && vae_region_end <= ctx->arena.off, CB_ERR_STATE, "step arena regions out of order");
            CB_HIP(hipEventRecord(ctx->ev_tail_fork, main_stream));
            CB_HIP(hipStreamWaitEvent(ctx->aux2, ctx->ev_tail_fork, 0));
            G.L.stream = ctx->aux2;
        }
        G.backward();
    }
    struct TailDone {      // record the tail's end on every exit path of the function below (use_rm_mlp returns early, but never defers)
        Ctx* ctx; Graph& G; hipStream_t main; bool on;
        const void* owner; int layout;
        ~TailDone() {
            if (on) { (void)hipEventRecord(ctx->ev_tail, ctx->aux2); ctx->tail_pending = true; ctx->tail_owner = owner; ctx->tail_layout = layout; G.L.stream = main; }
        }
    } tail_done{ctx, G, main_stream, defer_tail, T, layout};
    Phase ph_tail(L, "phase:head_bwd_adamw");
    const float* lr_dev = T->capturing ? T->lr_dev : nullptr;      // a captured step reads the rate from the device: one graph per batch size
    CB_CHECK(ctx->dry || emb1->ginit, CB_ERR_STATE, "no gradient reached the text embeddings");
    if (!emb1->g) emb1->g = ctx->arena.allocf((size_t)emb1->numel());
    float* dzz = ctx->arena.allocf((size_t)NF * es * w);
    k_fill(L, dzz, (long)NF * es * w, 0.f);
    k_row_surgery_bwd(L, dzz, static_cast<float*>(emb1->g), map, B, T77, w, es);          // fills the first B samples
    float* dxn = ctx->arena.allocf((size_t)NF * es * K);
    k_basis_bwd(L, dxn, dzz, T->basis, NF, es, K, w);
    k_step_inc(L, T->step);
    if (rm) {
        // d coef[id] = sum of dxn over the rows that gathered it (dense gradient: absent identities get zero, like autograd's
        // index backward; torch's AdamW then still applies weight decay to those rows)
        k_coef_scatter_grad(L, T->gC, dxn, bt.ids, 2, B, 2, O, c.max_ids);
        k_adamw(L, T->coef, T->mC, T->vC, T->gC, (long)c.max_ids * O, T->step, lr, c.adam_b1, c.adam_b2, c.adam_eps, c.weight_decay, lr_dev);
        return;
    }
    float* dact = ctx->arena.allocf((size_t)NF * O);
    k_l2norm_bwd(L, dact, xn->pf(), cn, dxn, (long)NF * es, K);
    float* dpre = ctx->arena.allocf((size_t)NF * O);
    k_leakyrelu_bwd(L, dpre, pre->pf(), dact, (long)NF * O, 0.2f);
    k_outer_grad(L, T->gW, T->gb, dpre, v->pf(), NF, O, I);

    // (8) AdamW on the 525 312 trainable floats (ddpm.py:1442-1454)
    k_adamw(L, T->W, T->mW, T->vW, T->gW, (long)O * I, T->step, lr, c.adam_b1, c.adam_b2, c.adam_eps, c.weight_decay, lr_dev);
    k_adamw(L, T->b, T->mb, T->vb, T->gb, (long)O, T->step, lr, c.adam_b1, c.adam_b2, c.adam_eps, c.weight_decay, lr_dev);
}

void trainer_set_graph_replay(Trainer* T, bool on) { T->graph.enabled = on; }

// stage the batch, then replay (or first capture) the step's graph
static void step_graph(Trainer* T, const Batch& bt, float lr) {
    Ctx* ctx = T->ctx;
    const TrainerCfg& c = T->cfg;
    Trainer::StepGraph& g = T->graph;
    constexpr int kMaxGraphs = 8;
    const int B = bt.B, S_ = c.image_size, lat = S_ >> (c.vae.n_mult - 1), zc = c.vae.z_channels, T77 = c.clip.ctx_len;
    if (!g.image) {
        const size_t n_img = (size_t)kMaxBatch * S_ * S_ * 3, n_faces = (size_t)kMaxBatch * S_ * S_ * 6, n_lat = (size_t)kMaxBatch * zc * lat * lat;
        const size_t n_ids = (size_t)kMaxBatch * 2, n_tok = (size_t)kMaxBatch * T77, n_t = kMaxBatch;
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t bytes = up(n_img * 4) + up(n_faces * 4) + 2 * up(n_lat * 4) + up(n_ids * 8) + up(n_tok * 8) + up(n_t * 8);
        CB_HIP(hipMalloc(&g.stage_base, bytes));
        char* p = static_cast<char*>(g.stage_base);
        auto take = [&](size_t b) { char* r = p; p += up(b); return r; };
        g.image = reinterpret_cast<float*>(take(n_img * 4)); g.faces = reinterpret_cast<float*>(take(n_faces * 4));
        g.noise = reinterpret_cast<float*>(take(n_lat * 4)); g.post_noise = reinterpret_cast<float*>(take(n_lat * 4));
        g.ids = reinterpret_cast<int64_t*>(take(n_ids * 8)); g.tokens = reinterpret_cast<int64_t*>(take(n_tok * 8)); g.t = reinterpret_cast<int64_t*>(take(n_t * 8));
    }
    hipStream_t s = ctx->stream;
    ctx->join_tail();                 // an eager step may have left its tail on the side stream
    auto stage = [&](void* dst, const void* src, size_t bytes) { CB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); };
    stage(g.image, bt.image, (size_t)B * S_ * S_ * 3 * sizeof(float));
    stage(g.faces, bt.faces, (size_t)B * S_ * S_ * 6 * sizeof(float));
    stage(g.noise, bt.noise, (size_t)B * zc * lat * lat * sizeof(float));
    stage(g.post_noise, bt.post_noise, (size_t)B * zc * lat * lat * sizeof(float));
    stage(g.ids, bt.ids, (size_t)B * 2 * sizeof(int64_t));
    stage(g.tokens, bt.tokens, (size_t)B * T77 * sizeof(int64_t));
    stage(g.t, bt.t, (size_t)B * sizeof(int64_t));
    // a graph is valid for the arena it was captured on (a larger batch size in between re-allocates the step arena and every address
    // inside the older graphs is gone with it), for the warp-grid table the context held (its device pointer is a captured kernel
    // argument; cb_ctx_set_warp_grid may have freed or replaced it since) and for the GEMM plans / pipeline switches in force
    // (cb_debug_force_plan and friends change which kernels a step launches): anything captured under other stamps is dropped
    const long plan_gen = plan_force().gen;
    for (size_t i = 0; i < g.cache.size();) {
        if (g.cache[i].arena_base != ctx->arena.base || g.cache[i].warp_gen != ctx->warp_gen || g.cache[i].plan_gen != plan_gen) {
            CB_HIP(hipStreamSynchronize(s));
            (void)hipGraphExecDestroy(g.cache[i].exec);
            g.cache.erase(g.cache.begin() + i);
        } else ++i;
    }
    Trainer::StepGraph::Entry* e = nullptr;
    for (auto& x : g.cache) if (x.B == B) e = &x;
    if (!e) {
        Batch sb;
        sb.image = g.image; sb.faces = g.faces; sb.ids = g.ids; sb.tokens = g.tokens; sb.t = g.t; sb.noise = g.noise; sb.post_noise = g.post_noise;
        sb.B = B;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        ctx->arena.reset();
        // the caller's stream may be the legacy default stream (PyTorch's current stream usually is), which cannot be captured: the
        // step's host code runs once against a library-owned capture stream; the instantiated graph is launched on the caller's.
        // A runtime that refuses the capture or the instantiation must not take the training run down: the trainer falls back to
        // eager launches for good (graph.broken) and this step runs eagerly.
        bool ok = true;
        if (!g.cap) ok = hipStreamCreateWithFlags(&g.cap, hipStreamNonBlocking) == hipSuccess;
        if (ok) {
            ctx->stream = g.cap;
            ok = hipStreamBeginCapture(g.cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                T->capturing = true;          // every forked stream must rejoin the capture stream: no deferred tail
                try { step_impl(T, sb, lr); }
                catch (...) { T->capturing = false; ctx->stream = s; (void)hipStreamEndCapture(g.cap, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
                T->capturing = false;
                ok = hipStreamEndCapture(g.cap, &graph) == hipSuccess && graph != nullptr;
            }
            ctx->stream = s;
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (!ok) {
            (void)hipGetLastError();
            g.broken = true;
            ctx->arena.reset();
            step_impl(T, bt, lr);            // eager, on the caller's own buffers (nothing was launched by the failed capture)
            return;
        }
        if ((int)g.cache.size() >= kMaxGraphs) {
            size_t lru = 0;
            for (size_t i = 1; i < g.cache.size(); ++i) if (g.cache[i].used < g.cache[lru].used) lru = i;
            CB_HIP(hipStreamSynchronize(s));                 // the evicted graph may still be in flight
            (void)hipGraphExecDestroy(g.cache[lru].exec);
            g.cache.erase(g.cache.begin() + lru);
        }
        Trainer::StepGraph::Entry ne; ne.B = B; ne.exec = exec; ne.arena_base = ctx->arena.base; ne.warp_gen = ctx->warp_gen; ne.plan_gen = plan_gen;
        g.cache.push_back(ne);
        e = &g.cache.back();
        g.captures++;
    }
    {   // the rate of THIS step: a 4-byte memset node ahead of the graph (no host buffer whose lifetime would matter)
        unsigned bits; memcpy(&bits, &lr, sizeof bits);
        CB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(T->lr_dev), (int)bits, 1, s));
    }
    e->used = ++g.tick;
    CB_HIP(hipGraphLaunch(e->exec, s));
    g.replays++;
}

void trainer_step(Trainer* T, const Batch& bt, float lr, bool dry_only) {
    Ctx* ctx = T->ctx;
    CB_CHECK(bt.B >= 1 && bt.B <= kMaxBatch, CB_ERR_ARG, "trainer: batch %d unsupported (1..%d)", bt.B, kMaxBatch);
    // graph replay from the second step at a batch size on (the first one runs eagerly: it sizes the arena and latches the per-kernel
    // dynamic-LDS attributes); not while the live profiler is recording (its event pairs are host calls) or debug taps are collected (they are
    // registered by the host code of an eager step), and not for a batch whose first stage was prefetched (that path waits on a prefetch event the captured graph does not know)
    if (T->graph.enabled && !T->graph.broken && !ctx->dry && !dry_only && T->arena_B == bt.B && !ctx->prof.on && !ctx->debug_taps) {
        bool prefetched = false;
        for (auto& sl : T->pf.slot)
            if (sl.valid && sl.image == bt.image && sl.faces == bt.faces && sl.post_noise == bt.post_noise && sl.B == bt.B) prefetched = true;
        if (!prefetched) { step_graph(T, bt, lr); return; }
    }
    // a deferred tail is left alone only by the NEXT step of the SAME trainer at the SAME batch size (same arena offsets)
    if (ctx->tail_pending && (ctx->tail_owner != T || T->arena_B != bt.B || dry_only)) ctx->join_tail();
    if (ctx->dry || dry_only || T->arena_B != bt.B) {
        bool was_dry = ctx->dry;
        Stats saved = ctx->stats;
        ctx->dry = true; ctx->arena.dry = true; ctx->arena.reset(); ctx->arena.peak = 0;
        try { step_impl(T, bt, lr); } catch (...) { ctx->dry = was_dry; ctx->arena.dry = was_dry; throw; }
        T->arena_need = ctx->arena.peak;
        ctx->dry = was_dry; ctx->arena.dry = was_dry;
        if (was_dry || dry_only) return;       // keep the accounting of the dry pass
        ctx->stats = saved;
        ctx->ensure_arena(T->arena_need);
        T->arena_B = bt.B;
    }
    ctx->arena.reset();
    step_impl(T, bt, lr);
}

}  // namespace cb
