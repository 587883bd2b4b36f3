// cb_models.h — the frozen networks of the CelebBasis training step expressed on the native tape runtime.
#pragma once
#include "cb_engine.h"

namespace cb {

// ---------------------------------------------------------------- UNet (openaimodel.py:413-742)
struct UNetCfg {
    int in_channels = 4, out_channels = 4, model_channels = 320;
    int num_res_blocks = 2;
    int channel_mult[8] = {1, 2, 4, 4, 0, 0, 0, 0}; int n_mult = 4;
    int attention_resolutions[8] = {4, 2, 1, 0, 0, 0, 0, 0}; int n_attn = 3;
    int num_heads = 8;
    int context_dim = 768;
};
struct ResBlockW { Norm n1; Conv c1; Norm n2; Conv c2; bool has_skip = false; Lin skip; int cin = 0, cout = 0;
                   int emb_col = 0; /* column of this block's emb_layers output inside UNet::emb_all */ };
struct XfmrW { Norm norm; Lin proj_in; Norm ln1, ln2, ln3; Lin qkv1, out1, q2, out2, ff1, ff2, proj_out; int ch = 0, heads = 0;
               int kv_col = 0; /* column of this layer's K block inside UNet::kv_all (V follows at kv_col + ch) */ };
struct UBlock { bool has_res = false; ResBlockW res; bool has_xf = false; XfmrW xf; bool has_down = false; Conv down;
                bool has_up = false; Conv up; };
struct UNet {
    UNetCfg cfg;
    DType dt = DT_F32;   // element type of activations and GEMM weights (the context's dtype at load)
    Lin te0, te2;
    Conv conv_in;
    std::vector<UBlock> in_blocks;      // input_blocks[1..]
    ResBlockW mid1, mid2; XfmrW midx;
    std::vector<UBlock> out_blocks;
    Norm out_norm; Conv out_conv;
    Lin emb_all;         // the emb_layers.1 Linear of all 22 ResBlocks stacked: [sum cout, 4*model_channels] (+ bias)
    Lin kv_all;          // every cross-attention to_k / to_v of the UNet stacked: [sum 2*ch, context_dim]
    Lin kv_all32;        // bf16 UNets under CB_BF16_KV_F32=1: fp32 copy of kv_all (projection + its input gradient run in fp32)
};
void unet_load(Ctx* ctx, const WTab& wt, const std::string& prefix, const UNetCfg& cfg, UNet* U);
// x: [B,H,W,Cin] NHWC, t: device int64[B], context: [B,L,ctx_dim]; returns eps_hat [B,H,W,Cout]
// context_f32 (optional): the conditioning BEFORE its cast to the UNet's element type; used by the fp32 K/V-projection lever
// `before_context` (optional) is called once, on the graph's stream, right before the first kernel that READS the conditioning (the stacked
// K/V projection, issued at the first cross-attention): the trainer joins its side branch there instead of ahead of the whole UNet
Ten* unet_forward(Graph& G, const UNet& U, Ten* x, const int64_t* t_dev, Ten* context, Ten* context_f32 = nullptr,
                  const std::function<void()>* before_context = nullptr);

// ---------------------------------------------------------------- VAE encoder (model.py:368-459, autoencoder.py:324-328)
struct VaeCfg { int ch = 128; int ch_mult[8] = {1, 2, 4, 4, 0, 0, 0, 0}; int n_mult = 4; int num_res_blocks = 2; int in_channels = 3;
                int z_channels = 4; };
struct VResW { Norm n1; Conv c1; Norm n2; Conv c2; bool has_nin = false; Lin nin; };
struct VaeEnc {
    VaeCfg cfg;
    DType dt = DT_F32;
    Conv conv_in;
    std::vector<std::vector<VResW>> down; std::vector<Conv> downsample; std::vector<bool> has_down;
    VResW mid1, mid2; Norm attn_norm; Lin attn_qkv, attn_proj;
    Norm norm_out; Conv conv_out; Lin quant;
};
void vae_load(Ctx* ctx, const WTab& wt, const std::string& prefix, const VaeCfg& cfg, VaeEnc* V);
Ten* vae_encode_moments(Graph& G, const VaeEnc& V, Ten* img /*[B,H,W,3]*/);   // -> [B,H/8,W/8,2*z]

// ---------------------------------------------------------------- VAE decoder (model.py:462-568, autoencoder.py:330-333) — sampling path
struct VaeDec {
    VaeCfg cfg;
    DType dt = DT_F32;
    Lin post_quant;
    Conv conv_in;
    VResW mid1, mid2; Norm attn_norm; Lin attn_qkv, attn_proj;
    std::vector<std::vector<VResW>> up;      // indexed by level like the reference's self.up (level 0 = full resolution)
    std::vector<Conv> upsample; std::vector<bool> has_up;
    Norm norm_out; Conv conv_out;
};
void vae_dec_load(Ctx* ctx, const WTab& wt, const std::string& prefix, const VaeCfg& cfg, VaeDec* V);
Ten* vae_decode(Graph& G, const VaeDec& V, Ten* z /*[B,h,w,z_channels], already divided by scale_factor*/);   // -> [B,8h,8w,3]

// ---------------------------------------------------------------- CLIP text transformer (modules.py:345-404 + HF 4.18 modeling_clip)
struct ClipCfg { int vocab = 49408, width = 768, layers = 12, heads = 12, ctx_len = 77, mlp = 3072; };
struct ClipLayerW { Norm ln1, ln2; Lin qkv, out, fc1, fc2; };
struct Clip {
    ClipCfg cfg;
    DType dt = DT_F32;
    float* tok_emb = nullptr; float* pos_emb = nullptr;
    std::vector<ClipLayerW> layers;
    Norm final_ln;
};
void clip_load(Ctx* ctx, const WTab& wt, const std::string& prefix, const ClipCfg& cfg, Clip* C);
// embedded: [B,L,width] token embeddings AFTER the manager hook (before position embedding)
Ten* clip_encode(Graph& G, const Clip& C, Ten* embedded);

// ---------------------------------------------------------------- face-recognition backbone (iresnet.py:67-181), frozen/eval
struct IBlockW { float* bn1_s; float* bn1_b; Conv c1; float* prelu; Conv c2; bool has_ds = false; Conv ds; int stride = 1; };
struct FrNet {
    int layers[4] = {3, 13, 30, 3};
    int width = 64;          // channels of layer1 (x1, x2, x4, x8)
    int in_size = 112;
    int feat = 512;
    Conv conv1; float* prelu1 = nullptr;
    std::vector<IBlockW> blocks;
    float* bn2_s = nullptr; float* bn2_b = nullptr;
    Lin fc;
};
void frnet_load(Ctx* ctx, const WTab& wt, const std::string& prefix, FrNet* F, const int* layers, int width, int in_size);
Ten* frnet_forward(Graph& G, const FrNet& F, Ten* img /*[N,S,S,3]*/);   // -> [N,1,1,512] (not normalised)

// ---------------------------------------------------------------- the training step
struct TrainerCfg {
    UNetCfg unet; VaeCfg vae; ClipCfg clip;
    int fr_layers[4] = {3, 13, 30, 3}; int fr_width = 64; int fr_in = 112;
    int image_size = 512;
    int es = 2;                 // num_embeds_per_token
    int inner_dim = 512;        // meta_inner_dim (number of basis vectors)
    int max_ids = 10;
    float momentum = 0.99f;
    float scale_factor = 0.18215f;
    float linear_start = 0.00085f, linear_end = 0.012f; int timesteps = 1000;
    int64_t placeholder_token = 48136;   // 'sks' (modules.py:198-262)
    float adam_b1 = 0.9f, adam_b2 = 0.999f, adam_eps = 1e-8f, weight_decay = 1e-2f;
    // meta_net.py:173-175, 272-273 (ablation switch `use_rm_mlp`): the (max_ids, es, 1, inner) coefficients are the trainable
    // parameter themselves — x = coef[id_idx], no FR-net / MLP / L2-norm — "optimise the 1024 coefficients directly"
    int use_rm_mlp = 0;
};
struct Trainer {
    Ctx* ctx = nullptr;
    TrainerCfg cfg;
    UNet unet; VaeEnc vae; Clip clip; FrNet fr;
    float* basis = nullptr;                       // [es][1+inner][width]
    float* W = nullptr; float* b = nullptr;       // EqualLinear(512 -> es*inner) — the only trainable tensors
    float* mW = nullptr; float* vW = nullptr; float* mb = nullptr; float* vb = nullptr;
    float* gW = nullptr; float* gb = nullptr;
    float* coef = nullptr; float* mC = nullptr; float* vC = nullptr; float* gC = nullptr;   // use_rm_mlp: [max_ids][es*inner]
    int* step = nullptr;
    float* id_emb = nullptr; float* id_coef = nullptr;   // [max_ids][es*width], [max_ids][es*inner]
    float* sqrt_ac = nullptr; float* sqrt_1mac = nullptr;
    float* loss = nullptr;                        // device scalar
    float* lr_dev = nullptr;                      // device scalar: the learning rate of a REPLAYED step (written before every hipGraphLaunch)
    size_t arena_need = 0; int arena_B = 0;
    bool capturing = false;                       // step_impl runs under hipStreamBeginCapture (no deferred tail)
    bool tail_overlap = true;                     // cb_trainer_set_tail_overlap / CB_TAIL_OVERLAP=0: the step's tail on the side stream, joined lazily
    // ---- optional first-stage prefetch (trainer_prefetch): the frozen, gradient-free stages of the NEXT batch (VAE encode ->
    // z0, FR backbone -> features) run on their own stream and arena while the current step's UNet is in flight
    struct Prefetch {
        hipStream_t stream = nullptr; bool own_stream = false; hipEvent_t ready = nullptr;
        Arena arena; size_t need = 0; int planned_B = 0;
        // two result slots: prefetch(n+1) is enqueued before step(n) consumes slot n
        struct Slot { void* z0 = nullptr; float* feat = nullptr; hipEvent_t done = nullptr;
                      const void *image = nullptr, *faces = nullptr, *post_noise = nullptr; int B = 0; bool valid = false; } slot[2];
        int next = 0;
        ~Prefetch() {
            if (stream) { (void)hipStreamSynchronize(stream); if (own_stream) (void)hipStreamDestroy(stream); }
            if (ready) (void)hipEventDestroy(ready);
            for (auto& sl : slot) if (sl.done) (void)hipEventDestroy(sl.done);
            if (arena.base) (void)hipFree(arena.base);
            if (arena.cnt.base) (void)hipFree(arena.cnt.base);
        }
    } pf;
    // ---- whole-step HIP-graph replay (SURVEY §8 B3 `cb_train_step`): the fixed-shape step — ~1 650 launches on two streams — is captured
    // once per (batch size, learning rate) and replayed with one hipGraphLaunch. The caller's batch is copied into trainer-owned staging
    // buffers first (7 small device copies), so the captured kernel arguments never change; the step arena is a bump allocator that
    // hands out the same addresses every step (cb_engine.h). Results are bitwise those of the eager path: same kernels, same order.
    struct StepGraph {
        bool enabled = false;
        // a graph bakes in: the arena's addresses, the context's warp-grid table pointer, the (tile, split-K) plans and pipeline
        // variants in force at capture. Each has a generation stamp; an entry captured under other stamps is dropped, not replayed.
        struct Entry { int B = 0; hipGraphExec_t exec = nullptr; long used = 0; const void* arena_base = nullptr; long warp_gen = 0, plan_gen = 0; };
        bool broken = false;                      // a capture / instantiate failed once: the trainer stays on eager launches
        std::vector<Entry> cache;                 // at most kMaxGraphs, least recently used evicted
        float* image = nullptr; float* faces = nullptr; int64_t* ids = nullptr; int64_t* tokens = nullptr; int64_t* t = nullptr;
        float* noise = nullptr; float* post_noise = nullptr;       // staging, sized for kMaxBatch: carved out of ONE allocation the trainer owns
        void* stage_base = nullptr;                                // (not the context's bump pool: a graph-enabled trainer gives its ~150 MB back when it is destroyed)
        long captures = 0, replays = 0, tick = 0;
        hipStream_t cap = nullptr;                // capture stream (the caller's may be the uncapturable legacy default stream)
        ~StepGraph() {
            for (auto& e : cache) if (e.exec) (void)hipGraphExecDestroy(e.exec);
            if (cap) (void)hipStreamDestroy(cap);
            if (stage_base) (void)hipFree(stage_base);
        }
    } graph;
};
void trainer_load(Ctx* ctx, const WTab& wt, const TrainerCfg& cfg, Trainer* T);
struct Batch {
    const float* image = nullptr;     // [B,S,S,3] HWC in [-1,1]        (face_id.py:598-644 'image')
    const float* faces = nullptr;     // [B,S,S,6]                      ('image_ori'['faces'])
    const int64_t* ids = nullptr;     // [B,2]                          ('image_ori'['ids'])
    const int64_t* tokens = nullptr;  // [B,77] CLIP BPE ids
    const int64_t* t = nullptr;       // [B] diffusion timesteps
    const float* noise = nullptr;     // [B,4,64,64] NCHW epsilon target
    const float* post_noise = nullptr;// [B,4,64,64] NCHW VAE posterior noise
    int B = 1;
};
void trainer_step(Trainer* T, const Batch& bt, float lr, bool dry_only);
void trainer_set_graph_replay(Trainer* T, bool on);   // default: off unless the environment says CB_STEP_GRAPH=1
// Enqueue the first stage of `next` (image -> z0, faces -> FR features) on the prefetch stream. A following trainer_step on the
// same buffers consumes the results instead of recomputing them; any other batch recomputes inline. Results are identical.
void trainer_prefetch(Trainer* T, const Batch& next);
hipStream_t trainer_prefetch_stream(Trainer* T);      // created on first use; nullptr on a dry context
void trainer_set_prefetch_stream(Trainer* T, hipStream_t s);   // caller-owned stream instead (before the first prefetch)

}  // namespace cb
