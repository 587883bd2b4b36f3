/* celebbasis_hip.h — C ABI of libcelebbasis_hip.so: the MI355X-native (gfx950) replacement for the
 * CelebBasis personalization training step.
 *
 * The reference (ygtxr1997/CelebBasis) has no FFI: its seam is the config-driven Python plugin boundary
 * `instantiate_from_config` (ldm/util.py:78-93) that resolves the YAML `target:` strings of
 * configs/stable-diffusion/aigc_id.yaml:3,24,40,57,80 to Python classes. This header is the boundary the
 * build's own `celebbasis_amd.ldm.*` classes bind (ctypes). Each entry point cites the
 * reference function it replaces.
 *
 * Conventions
 *  - every function returns 0 on success or a negative cb_status; cb_last_error() returns a thread-local
 *    message. No C++ exception crosses the boundary.
 *  - all tensor arguments are raw DEVICE pointers to float32 unless stated (int64 for token / id / timestep
 *    tensors, as in the reference). The caller owns every input/output buffer; the library never frees caller
 *    memory and owns only its handles, its weight copies and its per-context step arena.
 *  - element type. A context has one ACTIVATION dtype, fixed at creation (cb_ctx_create = f32, cb_ctx_create_dtype):
 *      CB_DTYPE_F32  — BASELINE config 2: the reference's own arithmetic (fp32 storage, exact-fp32 MFMA).
 *      CB_DTYPE_BF16 — BASELINE config 3: bf16 storage of the UNet / VAE / CLIP activations and frozen weights, bf16 MFMA
 *                      with fp32 accumulation; GroupNorm / LayerNorm / softmax statistics, the FR-net, the coefficient head,
 *                      its gradient (dW, db) and AdamW stay fp32.
 *    Op-level entry points take their activation tensors (`void*`) in the context's dtype (cb_gemm / cb_attention carry
 *    their own `dtype` in the descriptor, so fp32 head-path products can run on a bf16 context); norm parameters, biases,
 *    statistics, losses and everything of the coefficient head are float32 in both modes. Graph-level entry points keep
 *    float32 for the reference's PUBLIC tensors (images, latents, noise, conditioning, features) in both modes and convert
 *    at the boundary, so the host classes above this ABI do not change with the dtype.
 *  - work is enqueued on the context's stream (borrowed from the caller, e.g. PyTorch's current HIP stream);
 *    calls are asynchronous with respect to the host except the _create, _destroy and _get_state calls.
 *  - activation layout inside the library is NHWC ("tokens x channels"); NCHW appears only where the
 *    reference's public tensors are NCHW (latents, noise, UNet I/O) and is converted at the boundary.
 *  - a context created with device < 0 is a DRY context: no GPU is touched, kernels are not launched, but all
 *    shape checks, weight-name lookups, arena sizing and FLOP accounting run. It exists so the host logic can
 *    be tested on machines without a GPU. It computes nothing: there is no CPU fallback in this library.
 */
#ifndef CELEBBASIS_HIP_H
#define CELEBBASIS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CB_STATUS_OK = 0, CB_STATUS_ARG = -1, CB_STATUS_HIP = -2, CB_STATUS_SHAPE = -3, CB_STATUS_MISSING = -4,
               CB_STATUS_OOM = -5, CB_STATUS_STATE = -6, CB_STATUS_INTERNAL = -7 } cb_status;

typedef enum { CB_DTYPE_F32 = 0, CB_DTYPE_BF16 = 1 } cb_dtype;

typedef struct cb_ctx cb_ctx;
typedef struct cb_wtab cb_wtab;
typedef struct cb_unet cb_unet;
typedef struct cb_vae cb_vae;
typedef struct cb_vae_dec cb_vae_dec;
typedef struct cb_clip cb_clip;
typedef struct cb_frnet cb_frnet;
typedef struct cb_trainer cb_trainer;

const char* cb_last_error(void);
/* ABI version of this header. It changes whenever a struct layout or the meaning of an argument changes: 100 = round 1 (float-only
 * descriptors), 300 = this header (dtype-carrying descriptors, `struct_size` guards). cb_version() returns the library's value;
 * a binding compares the two at load time (celebbasis_amd/lib.py does). Every descriptor struct that has grown since 100 starts with
 * `struct_size` = sizeof(the struct as the CALLER compiled it); the library rejects any other size with CB_ERR_ARG instead of
 * reading fields the caller never wrote. */
#define CB_ABI_VERSION 303
int cb_version(void);

/* ---- context ------------------------------------------------------------------------------------ */
int cb_ctx_create(int device, void* hip_stream, cb_ctx** out);                           /* activation dtype f32 */
int cb_ctx_create_dtype(int device, void* hip_stream, int dtype, cb_ctx** out);         /* dtype: cb_dtype */
int cb_ctx_dtype(cb_ctx* ctx, int* dtype);
/* The library-owned side stream (hipStream_t) on which cb_trainer_step runs the identity branch — it reads the caller's `faces`,
 * `ids` and `tokens` buffers there, forked from AND JOINED BACK INTO the context stream with events inside the call: when the call
 * returns every read of CALLER memory is ordered before later context-stream work, so no record_stream is needed for it (exposed for
 * profilers). The step's deferred tail (cb_trainer_set_tail_overlap) runs on a third library stream; it touches library-owned
 * memory only. */
int cb_ctx_aux_stream(cb_ctx* ctx, void** hip_stream);
int cb_ctx_destroy(cb_ctx* ctx);
int cb_ctx_sync(cb_ctx* ctx);
/* accounting since the last reset: sum of 2*M*N*K over GEMM-class launches, kernel launches,
 * peak step-arena bytes, bytes held by the persistent pool (weights + packs + optimiser state) */
int cb_ctx_stats(cb_ctx* ctx, double* gemm_flops, int64_t* launches, int64_t* arena_peak_bytes, int64_t* pool_bytes);
int cb_ctx_reset_stats(cb_ctx* ctx);
/* The face warp's sampling grid as a constant table (meta_net.py:131-142, 253-257: F.affine_grid of a constant matrix at a constant
 * S x S size). host_grid = [S][S][2] fp32 (x, y) normalised coordinates exactly as the caller's host evaluates them; copied to the
 * device, kept by the context and consumed by every later warp at that S (cb_face_warp_resize, cb_trainer_step / _prefetch). Without
 * it (or at another S) the kernel derives the grid analytically from the same matrix — equal up to two fp32 ulps of a coordinate.
 * host_grid = NULL drops the table. */
int cb_ctx_set_warp_grid(cb_ctx* ctx, const float* host_grid, int S);
/* debug taps: named intermediates of the last graph-level call (valid until the next call) */
int cb_ctx_set_debug_taps(cb_ctx* ctx, int enable);
int cb_ctx_tap(cb_ctx* ctx, const char* name, float* host_dst, int64_t max_elems, int64_t* numel);

/* per-launch HIP-event timing of the GEMM-class kernels between begin/end (events are recorded on the context stream);
 * profile_end writes a JSON array [{"kernel": name, "launches": n, "flops": algorithmic 2MNK sum, "ms": summed duration}] */
int cb_ctx_profile_begin(cb_ctx* ctx);
int cb_ctx_profile_end(cb_ctx* ctx, char* json, int64_t cap);

/* ---- weight table: mirrors a torch state_dict (names = the reference's state-dict keys) */
int cb_wtab_create(cb_wtab** out);
int cb_wtab_add(cb_wtab* wt, const char* name, const float* data, int on_device, int ndim, const int64_t* dims);
int cb_wtab_destroy(cb_wtab* wt);

/* ---- op level (each replaces one torch op family on the reference hot path) ---------------------- */
typedef enum { CB_CONV_SAME = 0, CB_CONV_DOWN = 1, CB_CONV_DOWN_ASYM = 2, CB_CONV_UP = 3, CB_CONV_S2_1X1 = 4,
               CB_CONV_UP_DGRAD4 = 5 /* cb_conv2d_dgrad only: CB_CONV_UP's input gradient from the dgrad = 2 pack */ } cb_conv_mode;
typedef enum { CB_LD_ROW = 0, CB_LD_COL = 1 } cb_operand_layout;

/* C[M,N] = alpha * A.B + bias[n] + R + beta*C, batched; ROW = K contiguous, COL = M/N contiguous.
 * Replaces nn.Linear / einsum / bmm (ldm/modules/attention.py:170-193, openaimodel.py linear()). */
typedef struct {
    int struct_size;                             /* = sizeof(cb_gemm_desc) */
    const void* A; const void* B; void* C;      /* element type = dtype */
    int M, N, K;
    int64_t lda, ldb, ldc;
    int a_layout, b_layout;                      /* bf16: both operands must be CB_LD_ROW (K contiguous), K % 8 == 0, 16-byte aligned */
    int batch;
    int64_t strideA, strideB, strideC;
    const float* bias;                           /* float32 in both modes */
    const void* R; int64_t ldr; int64_t strideR;
    float alpha, beta;
    int dtype;                                   /* cb_dtype of A, B, C, R (descriptor-level: independent of the context's) */
} cb_gemm_desc;
int cb_gemm(cb_ctx* ctx, const cb_gemm_desc* d);
/* diagnostics (tools/tune_gemm.py, tools/pmc_n80.py): pin the GEMM planner to one tile shape — 64/128 x 64/128, 256 x 128 (bf16), or
 * the fp32 80-column tiles 64/32/16 x 80, which bind only the problems that can run them — and split-K count for every later launch
 * of this process; (0, 0, 0) releases it. Process-global, not thread safe — never call it in production code. */
int cb_debug_force_plan(int BM, int BN, int split);
/* diagnostics: the LDS-DMA GEMM / conv kernel synchronises its tile ring with a COUNTED `s_waitcnt vmcnt(N)` + raw `s_barrier`; on != 0
 * makes every later launch of this process drain the queue (`vmcnt(0)`) and use a full barrier instead — the slow reference pipeline the
 * counted one must agree with bit for bit (also: environment CB_GLDS_SAFE=1). Process-global, not thread safe. */
int cb_debug_glds_safe(int on);
/* diagnostics: a split-K launch of the LDS-DMA kernels folds its partial slabs INSIDE the launch — every block publishes its fp32 slab
 * (agent-scope release), draws a ticket from a per-tile arrival counter, and the tile's last block sums all slabs in split order and
 * runs the epilogue (deterministic: the order does not depend on who arrives last). on == 0 restores partials + the separate
 * splitk_reduce_kernel launch, which sums in the same order: the two must agree bit for bit (tests/test_gpu_ops.py; also environment
 * CB_SPLITK_INLAUNCH=0). Process-global, not thread safe. */
int cb_debug_splitk_inlaunch(int on);

/* OIHW -> implicit-GEMM packs. dgrad=0: [O][ky][kx][I]; dgrad=1: [I][ky'][kx'][O] with the kernel flipped; dgrad=2 (3x3 only):
 * [I][4][4][O], the input gradient of "nearest x2 upsample + conv3x3" (openaimodel.py Upsample) as ONE 4x4 stride-2 pad-1 conv over dy
 * (kernel rows combined as {w2, w1+w2, w0+w1, w0}): 2.25x fewer MACs than differentiating at the upsampled resolution. */
/* the source is the float32 state-dict tensor; the pack is in the context's dtype. bf16 packs pad the contiguous channel axis to a
 * multiple of 8: [O][ky][kx][ceil8(I)] resp. [I][ky'][kx'][ceil8(O)] (zero filled). */
int cb_pack_conv_weight(cb_ctx* ctx, const float* w_oihw, void* out, int O, int I, int ks, int dgrad);
/* nn.Conv2d forward on NHWC (openaimodel.py conv_nd / model.py Conv2d), optional fused per-image row bias
 * (ResBlock emb add, openaimodel.py:262-273) and residual (skip_connection(x) + h, openaimodel.py:275). */
int cb_conv2d_fwd(cb_ctx* ctx, const void* x, const void* w_packed, const float* bias, const void* rowbias,
                  const void* residual, void* y, int N, int H, int W, int Cin, int Cout, int ks, int mode);
/* input gradient of the same conv (autograd of F.conv2d wrt input; weights are frozen, ddpm.py:491-500) */
int cb_conv2d_dgrad(cb_ctx* ctx, const void* dy, const void* w_packed_dgrad, void* dx, int N, int H, int W, int Cin,
                    int Cout, int ks, int mode, float beta);

/* GroupNorm(32)(+SiLU): util.py:214-216 GroupNorm32 / attention.py:76-77 Normalize + nn.SiLU. stats: [N][32][2]. */
int cb_groupnorm_silu_fwd(cb_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                          int N, int HW, int C, float eps, int silu);
int cb_groupnorm_silu_bwd(cb_ctx* ctx, const void* x, const void* dy, const float* gamma, const float* beta,
                          const float* stats, void* dx, int N, int HW, int C, int silu, float acc_beta);
/* nn.LayerNorm over the last dim (attention.py:203-205, HF CLIP layer norms). stats: [rows][2]. */
int cb_layernorm_fwd(cb_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                     int64_t rows, int C, float eps);
int cb_layernorm_bwd(cb_ctx* ctx, const void* x, const void* dy, const float* gamma, const float* stats, void* dx,
                     int64_t rows, int C, float acc_beta);
/* GEGLU (attention.py:37-44) */
int cb_geglu_fwd(cb_ctx* ctx, const void* x, void* y, int64_t M, int D);
int cb_geglu_bwd(cb_ctx* ctx, const void* x, const void* dy, void* dx, int64_t M, int D);

/* Multi-head attention softmax(scale * Q K^T [+causal]) V  (attention.py:170-193 CrossAttention, model.py:176-202
 * AttnBlock, HF CLIPAttention). q lives in columns [q_off, q_off+C) of `qbuf` [B,Nq,ldq]; k and v in columns
 * k_off / v_off of `kvbuf` [B,Nk,ldkv] (kvbuf may equal qbuf for fused QKV). out: [B,Nq,C].
 * If dout != NULL the call also runs the backward and writes dq into dqbuf (same layout as qbuf, only the q columns)
 * and dk/dv into dkvbuf (same layout as kvbuf; may equal dqbuf). */
typedef struct {
    int struct_size;                             /* = sizeof(cb_attention_desc) */
    const void* qbuf; int64_t ldq; int q_off;
    const void* kvbuf; int64_t ldkv; int k_off, v_off;
    int B, Nq, Nk, C, heads;
    float scale; int causal;
    void* out;
    const void* dout; void* dqbuf; void* dkvbuf;
    int dtype;                                   /* cb_dtype of every tensor above (descriptor-level) */
} cb_attention_desc;
int cb_attention(cb_ctx* ctx, const cb_attention_desc* d);

/* small ops of the step, exported for parity tests */
int cb_timestep_embedding(cb_ctx* ctx, const int64_t* t, void* out, int B, int dim);                   /* util.py:151-171 */
int cb_q_sample(cb_ctx* ctx, const void* x0_nhwc, const float* eps_nchw, const int64_t* t, const float* sqrt_ac,
                const float* sqrt_1mac, void* xt_nhwc, int B, int C, int HW);                         /* ddpm.py:289-292 */
int cb_mse_loss(cb_ctx* ctx, const void* pred_nhwc, const float* target_nchw, float* loss, void* dpred, int B, int C,
                int HW);                                                                              /* ddpm.py:294-307 */
int cb_face_warp_resize(cb_ctx* ctx, const float* faces, float* warped, float* out, int B, int S, int nface, int out_size); /* meta_net.py:253-260 */
/* MetaIdNet head after the FR backbone (meta_net.py:262-289): normalize -> EqualLinear+LeakyReLU(0.2) -> per-(e) L2 norm ->
 * celeb-basis contraction. feat [N,I], W [es*K,I], b [es*K], basis [es,1+K,Wd] -> coef [N,es*K], z [N,es*Wd]. Forward only: the
 * training-time head (with its backward and AdamW) lives inside cb_trainer_step; this entry serves `test_mode: image`. */
int cb_celeb_head_forward(cb_ctx* ctx, const float* feat, const float* W, const float* b, const float* basis, int N, int I, int es,
                          int K, int Wd, float* coef, float* z);
/* feeder augmentations on the device (face_id.py:526-532 `trans` after the resize; :712-730 `_add_bg`), HWC images in [-1,1]:
 * params[B][5] = {flip, brightness, contrast, saturation, hue shift} (device floats, drawn by the host like the reference's
 * transforms); geom[B][4] = {rh, rw, pos_h, pos_w} (device int32). The colour ops follow torchvision's tensor formulas in the
 * fixed order b -> c -> s -> h. */
int cb_face_augment(cb_ctx* ctx, const float* in, const float* params, float* out, int B, int S);
int cb_paste_on_canvas(cb_ctx* ctx, const float* img, const int32_t* geom, float* out, int B, int S, int C);
int cb_row_surgery(cb_ctx* ctx, const float* embedded, const float* z, const int64_t* tokens, float* out, int32_t* map,
                   int B, int T, int C, int reps, int64_t placeholder);                               /* helpers.py:13-41 */
/* the two / three person prompts of embedding_manager.py:323-345, 363-391: placeholders = device int64 [B][nph] (negative =
 * unused slot), z = [B][nph*reps][C]; an occurrence of placeholder k is replaced by rows z[b][k*reps .. k*reps+reps-1] */
int cb_row_surgery_multi(cb_ctx* ctx, const float* embedded, const float* z, const int64_t* tokens, float* out, int32_t* map,
                         int B, int T, int C, int reps, const int64_t* placeholders, int nph);
int cb_adamw_step(cb_ctx* ctx, float* p, float* m, float* v, const float* g, int64_t n, int* step_dev, float lr, float b1,
                  float b2, float eps, float wd);                                                     /* ddpm.py:1454 */
int cb_nchw_to_nhwc(cb_ctx* ctx, const float* src, void* dst, int N, int C, int H, int W);   /* public fp32 NCHW -> activation NHWC */
int cb_nhwc_to_nchw(cb_ctx* ctx, const void* src, float* dst, int N, int C, int H, int W);

/* ---- graph level --------------------------------------------------------------------------------- */
typedef struct {
    int in_channels, out_channels, model_channels, num_res_blocks;
    int channel_mult[8]; int n_mult;
    int attention_resolutions[8]; int n_attn;
    int num_heads, context_dim;
} cb_unet_cfg;
/* UNetModel (openaimodel.py:413-742). Weights are looked up as prefix + state-dict key. */
int cb_unet_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_unet_cfg* cfg, cb_unet** out);
int cb_unet_destroy(cb_unet* u);
/* UNetModel.forward(x, timesteps, context) (openaimodel.py:710-742): x/eps NCHW [B,C,H,W], t int64[B],
 * context [B,L,context_dim]. need_grad=1 keeps the tape for cb_unet_backward_context. */
int cb_unet_forward(cb_unet* u, const float* x_nchw, const int64_t* t, const float* context, int B, int H, int W, int L,
                    float* eps_nchw, int need_grad);
/* d(sum(eps * deps)) / d context : the input-gradient-only backward the reference gets from autograd */
int cb_unet_backward_context(cb_unet* u, const float* deps_nchw, float* dcontext);
/* DDIMSampler.ddim_sampling + p_sample_ddim (ddim.py:113-204), eps-parameterisation, whole loop on the device.
 * x_nchw [n,C,H,W]: x_T on entry, the sample on return. cond [n,L,ctx]; uncond [n,L,ctx] or NULL (no guidance, ddim.py:172-173).
 * The S-entry HOST tables are the buffers DDIMSampler.make_schedule registers (ddim.py:25-54): timesteps = ddim_timesteps
 * (ascending; the loop runs them flipped, ddim.py:133), alphas = ddim_alphas, alphas_prev = ddim_alphas_prev,
 * sigmas = ddim_sigmas, sqrt_one_minus_alphas = ddim_sqrt_one_minus_alphas. noise_nchw: [S][n,C,H,W] device noise used when
 * sigmas != 0 (iteration order; NULL requires eta == 0). pred_x0_nchw (optional): the last step's pred_x0.
 * Classifier-free guidance runs as one batch-2n UNet pass per step (torch.cat([x]*2), ddim.py:175-179). */
int cb_unet_ddim_sample(cb_unet* u, float* x_nchw, const float* cond, const float* uncond, int n, int H, int W, int L, int S,
                        const int64_t* timesteps, const float* alphas, const float* alphas_prev, const float* sigmas,
                        const float* sqrt_one_minus_alphas, float guidance_scale, const float* noise_nchw, float* pred_x0_nchw);
/* one p_sample_ddim update given the (guided or 2n-row unguided) UNet output, in place on x_nchw (ddim.py:181-199) */
int cb_ddim_update(cb_ctx* ctx, float* x_nchw, const float* eps_nchw, const float* noise_nchw, int n, int C, int HW, int cfg_rows,
                   float guidance_scale, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, float* pred_x0_nchw);

typedef struct { int ch; int ch_mult[8]; int n_mult; int num_res_blocks; int in_channels; int z_channels; } cb_vae_cfg;
/* AutoencoderKL.encode (autoencoder.py:324-328) + DiagonalGaussianDistribution.sample * scale (ddpm.py:590-597).
 * image: [B,H,W,3] HWC (the dataset layout, face_id.py:598-644); moments_nhwc (optional) [B,H/8,W/8,2z];
 * z_nchw [B,z,H/8,W/8]; post_noise_nchw may be NULL (mode()). */
int cb_vae_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_vae_cfg* cfg, cb_vae** out);
int cb_vae_destroy(cb_vae* v);
int cb_vae_encode(cb_vae* v, const float* image_hwc, const float* post_noise_nchw, int B, int H, int W, float scale,
                  float* moments_nhwc, float* z_nchw);

typedef struct { int vocab, width, layers, heads, ctx_len, mlp; } cb_clip_cfg;
/* CLIP text transformer as patched by FrozenCLIPEmbedder (modules.py:176-404): takes the token embeddings AFTER the
 * embedding-manager hook, adds position embeddings, 12 causal layers, final LayerNorm. */
int cb_clip_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_clip_cfg* cfg, cb_clip** out);
int cb_clip_destroy(cb_clip* c);
int cb_clip_embed_tokens(cb_clip* c, const int64_t* tokens, int B, int L, float* embedded);   /* token_embedding lookup */
int cb_clip_forward(cb_clip* c, const float* embedded, int B, int L, float* hidden, int need_grad);
int cb_clip_backward_embeds(cb_clip* c, const float* dhidden, float* dembedded);

/* iresnet (iresnet.py:67-181), frozen eval: faces [N,S,S,3] NHWC -> features [N,feat] (before F.normalize) */
int cb_frnet_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const int* layers4, int width, int in_size, cb_frnet** out);
int cb_frnet_destroy(cb_frnet* f);
int cb_frnet_forward(cb_frnet* f, const float* img_nhwc, int N, float* features);

/* ---- the whole personalization training step (LatentDiffusion.training_step, ddpm.py:357-370 -> shared_step ->
 *      get_input -> forward -> p_losses, loss.backward(), AdamW.step(); SURVEY §3.1 hot loop) ------------- */
/* AutoencoderKL.decode (autoencoder.py:330-333: post_quant_conv -> Decoder.forward model.py:535-568) behind
 * LatentDiffusion.decode_first_stage (ddpm.py:761-819: z / scale_factor first). z_nchw [B,z,h,w] -> image_nchw [B,3,8h,8w].
 * to01 != 0 additionally applies clamp((x+1)/2, 0, 1) (stable_txt2img.py:343). */
int cb_vae_decoder_create(cb_ctx* ctx, const cb_wtab* wt, const char* prefix, const cb_vae_cfg* cfg, cb_vae_dec** out);
int cb_vae_decoder_destroy(cb_vae_dec* v);
int cb_vae_decode(cb_vae_dec* v, const float* z_nchw, int B, int h, int w, float scale_factor, int to01, float* image_nchw);

typedef struct {
    int struct_size;      /* = sizeof(cb_trainer_cfg); filled by cb_trainer_cfg_default */
    cb_unet_cfg unet; cb_vae_cfg vae; cb_clip_cfg clip;
    int fr_layers[4]; int fr_width; int fr_in;
    int image_size, es, inner_dim, max_ids;
    float momentum, scale_factor, linear_start, linear_end; int timesteps;
    int64_t placeholder_token;
    float adam_b1, adam_b2, adam_eps, weight_decay;
    int dtype;            /* cb_dtype of the frozen UNet / VAE (must equal the context's); CLIP, head + AdamW are always f32 */
    int use_rm_mlp;       /* meta_net.py:173-175, 272-273: the (max_ids, es, 1, inner) coefficients are the trainable parameter
                             (`embedding_manager.meta_id_net.coef` in the weight table); no FR-net / MLP / L2-norm in the step */
} cb_trainer_cfg;
int cb_trainer_cfg_default(cb_trainer_cfg* cfg);   /* SD v1-4 + aigc_id.yaml values */
/* wt must hold: model.diffusion_model.*, first_stage_model.*, cond_stage_model.transformer.text_model.*,
 * embedding_manager.meta_id_net.id_model.*, embedding_manager.meta_id_net.stylegan_mlp.net.0.{weight,bias},
 * celeb_embeddings [es,1+inner,width]; optional id_embeddings [max_ids, es*width], id_coefficients [max_ids, es*inner] */
int cb_trainer_create(cb_ctx* ctx, const cb_wtab* wt, const cb_trainer_cfg* cfg, cb_trainer** out);
int cb_trainer_destroy(cb_trainer* t);
/* cb_unet / cb_clip views that alias the trainer's device weights (no second copy of SD v1-4): the same run can personalise
 * (cb_trainer_step) and then sample (cb_clip_forward + cb_unet_ddim_sample), as stable_txt2img.py does with the trained model.
 * Destroy them with cb_unet_destroy / cb_clip_destroy; they must not outlive the cb_ctx. */
int cb_trainer_share_unet(cb_trainer* t, cb_unet** out);
int cb_trainer_share_clip(cb_trainer* t, cb_clip** out);
typedef struct {
    const float* image;       /* [B,S,S,3]  batch['image']                 */
    const float* faces;       /* [B,S,S,6]  batch['image_ori']['faces']    */
    const int64_t* ids;       /* [B,2]      batch['image_ori']['ids']      */
    const int64_t* tokens;    /* [B,77]     tokenizer(caption).input_ids   */
    const int64_t* t;         /* [B]        torch.randint(0, 1000)         */
    const float* noise;       /* [B,4,64,64] NCHW  randn_like(x_start)     */
    const float* post_noise;  /* [B,4,64,64] NCHW  posterior.sample noise  */
    int B;
} cb_batch;
int cb_trainer_step(cb_trainer* t, const cb_batch* batch, float lr);
/* Whole-step HIP-graph replay. From the second step at a batch size on, cb_trainer_step copies the batch into
 * trainer-owned staging buffers and replays ONE captured graph of the step's ~1 400 launches (both streams) instead of enqueuing them
 * one by one; ONE graph per batch size is kept (8 most recent): the learning rate is read from a device scalar that a 4-byte memset
 * node writes ahead of every replay, so a schedule (warm-up, decay) replays the same graph. Same kernels in the same order on the same
 * addresses: results are bitwise those of the eager path. A graph bakes in the step arena's addresses, the context's warp-grid table
 * pointer (cb_ctx_set_warp_grid) and the GEMM plans / pipeline variants in force (cb_debug_*): each carries a generation stamp, and
 * a graph captured under other stamps is dropped and re-captured at its next use. If the runtime refuses a capture or an
 * instantiation, the step runs eagerly and the trainer stays on eager launches. OPT-IN (this call, or environment CB_STEP_GRAPH=1):
 * on MI355X the step is GPU-bound — the replayed step measures the eager step's time within 0.3 % when the graph is captured in a
 * process that has created no other streams / contexts, and 2.2x that (83 ms against 37) when it is captured after a prefetch stream
 * or a second context exist: the runtime's graph-internal streams then take the hardware queues the step's own side streams use. That
 * dependence on process history is the runtime's, not this library's to fix; it is why replay is not the default. With replay on, the
 * eager path still serves the first step at a batch size, a step under cb_ctx_profile_begin or debug taps, and a batch consumed from
 * cb_trainer_prefetch. */
int cb_trainer_set_graph_replay(cb_trainer* t, int enable);
/* Deferred step tail (default ON; environment CB_TAIL_OVERLAP=0 or enable = 0 turns it off). Everything of a step after the UNet's
 * input gradient has reached the conditioning — the CLIP text transformer's input-gradient chain (~190 launches of 77-row problems),
 * the coefficient head's backward and AdamW (ddpm.py:1442-1454) — reads no caller memory and is needed by nothing before the NEXT
 * step's identity branch (which reads the updated weights). cb_trainer_step therefore enqueues it on the library's side stream and
 * does NOT make the context stream wait for it: the next step's VAE encode (context stream, frozen weights only) runs beside it, the
 * next step's identity branch queues behind it in side-stream order. Same kernels in the same order: results are bitwise those of
 * the joined form. What changes for the CALLER: when cb_trainer_step returns, the trainable state (mlp.*, adam.*, step) may still be
 * in flight on the side stream although the context stream has drained. Every library entry that could observe it — cb_ctx_sync,
 * cb_trainer_get_state / _set_state, cb_ctx_tap, every op- and graph-level call on the context, the next cb_trainer_step — joins
 * first (one hipStreamWaitEvent). cb_trainer_loss_to does not need to (the loss is written on the context stream). The tail is
 * not deferred under cb_ctx_profile_begin, debug taps, graph capture, or use_rm_mlp (whose tail reads the caller's `ids`). */
int cb_trainer_set_tail_overlap(cb_trainer* t, int enable);
int cb_trainer_graph_stats(cb_trainer* t, int64_t* captures, int64_t* replays);
/* Optional software pipelining across steps: enqueue the gradient-free first stage of the NEXT batch (VAE encode -> latent,
 * FR backbone -> face features; only image / faces / post_noise / B of `next` are read) on a library-owned stream so that it
 * overlaps the UNet of the step that is enqueued afterwards. A later cb_trainer_step on the same buffers consumes the result
 * (bitwise identical to computing it in the step); any other batch is computed inline as usual. Call order per iteration:
 * cb_trainer_prefetch(batch[i+1]); cb_trainer_step(batch[i]).
 * BUFFER LIFETIME: this is the one entry point whose reads of caller memory are NOT ordered on the context stream when it returns —
 * image / faces / post_noise are read on the prefetch stream until the cb_trainer_step that consumes them has been enqueued (that
 * step makes the context stream wait for the prefetch). Keep the three buffers alive until then, or tell a stream-ordered
 * allocator about the second reader (PyTorch: hand the library a torch-owned stream with cb_trainer_set_prefetch_stream before
 * the first prefetch and tensor.record_stream() the three tensors on it — what celebbasis_amd.lib.Trainer.prefetch does; a
 * library-owned stream would be destroyed with the trainer while the allocator still holds events on it). Every other entry point
 * joins its internal side stream into the context stream before it returns, so freeing a buffer in context-stream order after the
 * call is safe (tools/probes/probe_temporaries.py). */
int cb_trainer_prefetch(cb_trainer* t, const cb_batch* next);
int cb_trainer_set_prefetch_stream(cb_trainer* t, void* hip_stream); /* caller-owned hipStream_t; before the first prefetch */
int cb_trainer_prefetch_stream(cb_trainer* t, void** hip_stream);   /* the stream in use (created on first use); NULL on a dry context */

/* enqueue a copy of the last step's loss (device scalar) into caller-owned DEVICE memory: `training_step` can return the loss as a
 * 0-dim device tensor without a host synchronisation (ddpm.py:357-370 returns `loss`) */
int cb_trainer_loss_to(cb_trainer* t, float* loss_dev);

/* state names: "loss"(1) "mlp.weight" "mlp.bias" "mlp.grad_weight" "mlp.grad_bias" "adam.m_weight" "adam.v_weight"
 * "adam.m_bias" "adam.v_bias" "id_embeddings" "id_coefficients" "step"(1, as float);
 * with use_rm_mlp additionally "rm.coef" "rm.grad_coef" "adam.m_coef" "adam.v_coef" ([max_ids, es*inner]) */
int cb_trainer_get_state(cb_trainer* t, const char* name, float* host_dst, int64_t max_elems, int64_t* numel);
int cb_trainer_set_state(cb_trainer* t, const char* name, const float* host_src, int64_t numel);

#ifdef __cplusplus
}
#endif
#endif
