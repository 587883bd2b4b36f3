// cb_engine.h — native runtime of the CelebBasis MI355X training step: device memory pools, a weight table
// keyed by the reference's state-dict names, and a tiny input-gradient-only autograd tape (every pretrained
// weight is frozen, ddpm.py:491-500, so backward needs dgrad only; pruning of branches that do not depend
// on the conditioning falls out of requires-grad propagation exactly as in torch autograd).
#pragma once
#include "cb_common.h"
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace cb {

// ---------------------------------------------------------------- memory
// Bump allocator. The step arena is reset at the start of every step and never reuses memory inside a
// step (288 GB of HBM3E: ~10 GB of live activations is nothing), which makes every intermediate a valid
// debug tap until the next step and makes addresses replay-stable for hipGraph capture.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = false;
    CounterPool cnt;             // split-K arrival counters, rewound with the arena
    void* alloc(size_t bytes) {
        off = (size_t)rup((long)off, 256);
        if (!dry) CB_CHECK(off + bytes <= cap, CB_ERR_OOM, "step arena exhausted: need %zu, cap %zu", off + bytes, cap);
        void* p = (dry ? reinterpret_cast<char*>(0x100000) : base) + off;
        off += bytes;
        if (off > peak) peak = off;
        return p;
    }
    float* allocf(size_t n) { return static_cast<float*>(alloc(n * sizeof(float))); }
    void reset() { off = 0; cnt.off = 0; }
};

// Persistent pool (weights, packed weights, optimiser state): grows in hipMalloc'ed slabs.
struct Pool {
    bool dry = false;
    std::vector<void*> slabs;
    char* cur = nullptr; size_t left = 0; size_t total = 0;
    size_t slab_bytes = (size_t)1 << 30;
    void* alloc(size_t bytes);
    float* allocf(size_t n) { return static_cast<float*>(alloc(n * sizeof(float))); }
    void release();
};

// ---------------------------------------------------------------- weight table (mirrors a torch state_dict)
struct WEntry { const void* data = nullptr; int on_device = 0; std::vector<int64_t> dims; int dtype = 0; /*0 f32, 1 i64*/ };
struct WTab {
    std::map<std::string, WEntry> m;
    const WEntry& get(const std::string& name) const {
        auto it = m.find(name);
        CB_CHECK(it != m.end(), CB_ERR_MISSING, "weight '%s' missing from the table", name.c_str());
        return it->second;
    }
    bool has(const std::string& name) const { return m.count(name) != 0; }
};

struct Ctx {
    int device = -1;             // -1: dry (no GPU): shape checks and accounting only
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;           // library-owned side stream: independent branches of the step (FR-net || VAE encode)
    hipStream_t aux2 = nullptr;          // library-owned: a training step's deferred tail (below)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Deferred tail of a training step (cb_models.hip, step_impl): the CLIP input-gradient chain, the coefficient head's backward and
    // AdamW of step i are enqueued on `aux2` and recorded in ev_tail instead of being joined into `stream` — step i+1's VAE encode
    // (main stream) and FR backbone (`aux`), both independent of the trainable tensors, then run beside them; step i+1's head waits
    // for ev_tail on `aux`. EVERY other consumer of this context joins first: launch() (all op- and graph-level entries),
    // ensure_arena, the state / tap / sync entries (join_tail is one hipStreamWaitEvent).
    hipEvent_t ev_tail = nullptr, ev_tail_fork = nullptr;
    bool tail_pending = false; const void* tail_owner = nullptr; int tail_layout = 0;    // layout: arena layout class of the step that left the tail
    void join_tail() {
        if (!tail_pending) return;
        // (`device`, not `dry`: plan_and_run flips `dry` for its sizing pass on a REAL context, and that pass must not swallow the join)
        if (device >= 0 && ev_tail) CB_HIP(hipStreamWaitEvent(stream, ev_tail, 0));
        tail_pending = false;
    }
    bool dry = true;
    Arena arena;                 // per-step activations
    Pool pool;                   // persistent
    Stats stats;
    Profiler prof;
    bool debug_taps = false;
    struct Tap { const void* p; long n; DType dt; };
    std::map<std::string, Tap> taps;
    // element type of the activations and frozen weights of the UNet / VAE / CLIP handles created on this context (fixed at
    // creation: cb_ctx_create_dtype). The FR-net, the coefficient head, its gradient and AdamW are always fp32.
    DType dtype = DT_F32;
    DType load_dt = DT_F32;      // element type the weight loaders currently pack to (set by the *_load functions)
    // the face warp's constant sampling grid [S][S][2] as the caller's host evaluated it (cb_ctx_set_warp_grid); nullptr = analytic
    float* warp_grid = nullptr; int warp_S = 0;
    long warp_gen = 0;           // bumped by every cb_ctx_set_warp_grid: a captured step graph holds the table's device pointer
    const float* warp_grid_for(int S) const { return (warp_grid && warp_S == S) ? warp_grid : nullptr; }
    Launch launch() { join_tail(); return launch_nojoin(); }
    Launch launch_nojoin() { Launch L; L.stream = stream; L.dry = dry; L.stats = &stats; L.prof = &prof; L.cnt = &arena.cnt; return L; }   // the trainer's own steps only
    void ensure_arena(size_t bytes);
    // upload a table entry to the device (raw fp32 copy); dims are checked against `expect` (-1 = any)
    float* upload(const WTab& wt, const std::string& name, std::initializer_list<int64_t> expect);
    // a GEMM operand in load_dt made from a device fp32 matrix [rows][cols], columns zero-padded to cols_p
    void* operand(const float* src, long rows, int cols, int cols_p);
    ~Ctx();
};

// ---------------------------------------------------------------- tensors and tape
struct Ten {
    void* p = nullptr;
    void* g = nullptr;
    int n = 1, h = 1, w = 1, c = 1;   // NHWC; sequences are (n=B, h=L, w=1, c)
    bool rg = false;                  // requires grad
    bool ginit = false;               // g holds a value (first writer uses beta=0)
    DType dt = DT_F32;                // element type of p and g
    long rows() const { return (long)n * h * w; }
    long numel() const { return rows() * c; }
    float* pf() const { return static_cast<float*>(p); }             // fp32 tensors only
    void* at(long elem) const { return ptr_off(p, dt, elem); }       // element offset into p
    void* gat(long elem) const { return ptr_off(g, dt, elem); }
    size_t bytes() const { return (size_t)numel() * dt_size(dt); }
};

// Frozen weights. `dt` is the element type of the GEMM operands (w, wd, wT); biases and norm parameters are always fp32.
// bf16 operands must be 16-byte chunks of 8 channels: the packs of a conv whose channel count is not a multiple of 8 (the RGB /
// latent ends of the networks) are zero-padded to cin_p / cout_p and the op pads the activation to match.
struct Conv { void* w = nullptr; void* wd = nullptr; float* b = nullptr; int cin = 0, cout = 0, ks = 3;
              DType dt = DT_F32; int cin_p = 0, cout_p = 0;
              void* wu = nullptr; /* CONV_UP layers on the gradient path: [I][4][4][O] pack of the 4x4 stride-2 input-gradient conv */ };
struct Lin { void* w = nullptr; float* b = nullptr; int in = 0, out = 0;
             DType dt = DT_F32; int in_p = 0;
             void* wT = nullptr; /* [in][out]: bf16 dgrad operand (the bf16 GEMM takes k-contiguous operands only) */ };
struct Norm { float* g = nullptr; float* b = nullptr; int c = 0; float eps = 1e-5f; };

enum ConvMode { CONV_SAME = 0, CONV_DOWN = 1, CONV_DOWN_ASYM = 2, CONV_UP = 3, CONV_S2_1x1 = 4 };

struct Graph {
    Ctx* ctx;
    Launch L;
    std::deque<Ten> tens;
    std::vector<std::function<void()>> tape;
    explicit Graph(Ctx* c, bool join_tail = true) : ctx(c), L(join_tail ? c->launch() : c->launch_nojoin()) {}

    Ten* make(int n, int h, int w, int c, bool rg, DType dt);
    Ten* wrap(const void* p, int n, int h, int w, int c, bool rg, DType dt);   // external buffer
    void* grad(Ten* t, float* beta);       // grad buffer for a writer; *beta = 0 on first write, 1 afterwards
    Ten* cast(Ten* x, DType dt);           // element-type change (differentiable); returns x itself when the type already matches
    void tap(const std::string& name, const Ten* t);
    void backward();                       // run (and drop) the whole tape, newest closure first
    void backward_until(size_t mark);      // ... only the closures pushed at or after index `mark`

    // ops (forward now, backward pushed on the tape when an input requires grad)
    // epi (no-grad fp32 layers: the frozen FR backbone): PReLU on the output and / or a second output tensor c2 = scale[c] * y + shift[c]
    struct ConvEpi { const float* prelu = nullptr; const float* c2_scale = nullptr; const float* c2_shift = nullptr; Ten** c2_out = nullptr; };
    Ten* conv(Ten* x, const Conv& cv, ConvMode mode, Ten* rowbias = nullptr, Ten* residual = nullptr, int rowbias_col = 0, const ConvEpi* epi = nullptr);
    Ten* linear(Ten* x, const Lin& l, Ten* residual = nullptr);
    Ten* groupnorm(Ten* x, const Norm& nm, bool silu);
    Ten* layernorm(Ten* x, const Norm& nm);
    // q: columns [qoff, qoff+C) of qt ; k/v: columns of kt at koff/voff. Output [B, Nq, C].
    Ten* attention(Ten* qt, int qoff, Ten* kt, int koff, int voff, int C, int heads, float scale, bool causal);
    Ten* attention_materialised(Ten* qt, int qoff, Ten* kt, int koff, int voff, int C, int heads, float scale, bool causal, Ten* o);
    bool use_flash = true;                  // fused attention for head dims <= 160 (cb_flash.hip); false = materialised path
    Ten* geglu(Ten* x);
    Ten* quickgelu(Ten* x);
    Ten* silu(Ten* x);                      // no grad path only
    Ten* concat(Ten* a, Ten* b);            // channel concat
    Ten* channel_affine(Ten* x, const float* scale, const float* shift, const float* prelu);   // no grad
};

// conv geometry / dgrad shared by the tape ops and the op-level C ABI
void conv_out_dims(const Conv& cv, ConvMode mode, int H, int W, int* Ho, int* Wo, ConvGeom* g);
void conv_dgrad(Graph* G, const void* dy, const Conv& cv, ConvMode mode, int N, int H, int W, void* dx, float beta);

// run fn once in dry mode to size the step arena, then for real (no-op split on a dry context)
void plan_and_run(Ctx* ctx, const std::function<void()>& fn);

// weight loading helpers (cb_engine.hip)
Conv load_conv(Ctx* ctx, const WTab& wt, const std::string& prefix, int cin, int cout, int ks, bool need_dgrad, bool bias = true,
               bool up_dgrad = false);
Lin load_linear(Ctx* ctx, const WTab& wt, const std::string& prefix, int in, int out, bool bias = true);
Lin load_conv1x1_as_linear(Ctx* ctx, const WTab& wt, const std::string& prefix, int in, int out);
Lin load_linear_cat(Ctx* ctx, const WTab& wt, const std::vector<std::string>& prefixes, int in, int out_each, bool bias,
                    bool conv1x1 = false);
Norm load_norm(Ctx* ctx, const WTab& wt, const std::string& prefix, int c, float eps);
// concatenate bias-free [out_i, in] projections of different widths into one [sum(out_i), in] weight
Lin load_linear_cat_var(Ctx* ctx, const WTab& wt, const std::vector<std::string>& prefixes, int in, const std::vector<int>& outs,
                        bool bias = false);

}  // namespace cb
