// pdengine: engine object behind the C ABI of include/pdengine.h (host-side orchestration).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pdengine.h"
#include "pd_common.h"

void pd_set_error(const char* fmt, ...);
// timestep_embedding (util.py:154-174) of n timesteps on the host: [cos | sin] halves, fp32 like the reference
void pd_host_timestep_embedding(const int64_t* t, int n, int dim, std::vector<float>& out);

#define HIP_OK(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            pd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)
#define PD_TRY(expr)            \
    do {                        \
        int _r = (expr);        \
        if (_r) return _r;      \
    } while (0)

// A GEMM weight matrix [Nalloc][Kpad] in the compute type, K-contiguous, plus its fp32 bias.
struct WMat {
    void* w = nullptr;
    float* bias = nullptr;
    int N = 0;       // weight rows the kernel sees (virtual columns; GEGLU: interleaved blocks)
    int Nout = 0;    // logical output columns
    int K = 0;       // logical reduction length taps*cin_pad
    int Kpad = 0;
    int taps = 1;
    int cin = 0, cin_pad = 0;
    int fan_in = 1;
    bool geglu = false;
    // LayerNorm folded into this layer (pd_engine::fold_layernorms): W * diag(gamma), its column sums, bias + beta . W^T
    void* w_ln = nullptr;
    float* colsum = nullptr;
    float* bias_ln = nullptr;
    // e4m3 copy with one scale per output row (pd_engine::sd3_quantize): the PREC_FP8 form of this layer
    void* w8 = nullptr;
    float* wscale = nullptr;
    int Kpad8 = 0;
    float wnorm_max = 0.f, bias_max = 0.f;   // max_n |W_n|_2 and max |bias|: bound of this layer's outputs from its input row's norm
};

// LayerNorm statistics handed from the GEMM that writes a residual-stream tensor to the GEMM that consumes its LayerNorm
struct LnStats {
    float* stats = nullptr;   // [rows][parts][2] {sum, sum of squares}
    int parts = 0;
    int C = 0;
    int cap_parts = 0;        // partials per row the buffer holds (producers check it before writing)
};

struct ConvW {
    WMat m;
    int cin = 0, cout = 0, k = 1, stride = 1;
};

struct ResW {
    int cin = 0, cout = 0;
    float *gn1_g = nullptr, *gn1_b = nullptr, *gn2_g = nullptr, *gn2_b = nullptr;
    ConvW conv1, conv2, skip;
    WMat emb;
    bool has_skip = false;
    float eps = 1e-5f;  // GroupNorm eps (1e-6 in the VAE's ResnetBlock)
    int emb_slot = -1;  // index into the per-net table of projected time embeddings
};

struct STW {
    int C = 0;
    float *gn_g = nullptr, *gn_b = nullptr;
    float* ln_g[3] = {nullptr, nullptr, nullptr};
    float* ln_b[3] = {nullptr, nullptr, nullptr};
    ConvW proj_in, proj_out;
    WMat qkv, out1, q2, kv2, out2, ff1, ff2;
    int kv_slot = -1;  // index into the per-net table of hoisted context K / V^T
    // fused tail (st_tail.hip): the block's matrices after self-attention in MFMA-fragment order + its fp32 vectors
    void* tail_w = nullptr;
    float* tail_vec = nullptr;
    void* front_w = nullptr;    // proj_in + to_q/k/v (norm1 folded) in the same order, for the fused front kernel
    float* front_vec = nullptr;
};

struct EncBlock {
    int kind = 0;  // 0 conv_in, 1 res(+attn), 2 down
    ConvW conv;
    ResW res;
    bool attn = false;
    STW st;
    int cout = 0, ds = 1;
};

struct DecBlock {
    ResW res;
    bool attn = false, up = false;
    STW st;
    ConvW upconv;
    int skip_c = 0, cout = 0;
};

struct NetW {
    WMat te0, te2;
    std::vector<EncBlock> enc;
    ResW mid0, mid2;
    STW mid1;
    // UNet only
    std::vector<DecBlock> dec;
    float *out_g = nullptr, *out_b = nullptr;
    ConvW outconv;
    // ControlNet only
    std::vector<ConvW> zero;
    ConvW mid_out;
    std::vector<ConvW> hint_pair, hint_query;
    int n_emb = 0, n_kv = 0;
    std::vector<ResW*> res_list;  // in emb_slot order
    std::vector<STW*> st_list;    // in kv_slot order
};

// First-stage decoder: ldm/modules/diffusionmodules/model.py:546-653 + post_quant_conv (autoencoder.py:34)
struct VaeLevel {
    std::vector<ResW> blocks;
    bool up = false;
    ConvW upconv;
    int ch = 0;
};
struct VaeW {
    bool built = false;
    ConvW post_quant, conv_in, conv_out, proj_out;
    ResW mid1, mid2;
    float *attn_g = nullptr, *attn_b = nullptr, *out_g = nullptr, *out_b = nullptr;
    WMat qkv;
    std::vector<VaeLevel> levels;  // execution order (highest resolution level last)
    int top = 0;
};

// CLIP text transformer
struct TextLayerW {
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    WMat qkv, out, fc1, fc2;
};
struct TextW {
    bool built = false;
    WMat tok, pos;
    std::vector<TextLayerW> layers;
    float *fln_g = nullptr, *fln_b = nullptr;
};

// SD3 / MMDiT: one JointTransformerBlock
struct Sd3BlockW {
    WMat qkv, qkv_c;     // to_q|to_k|to_v and add_q_proj|add_k_proj|add_v_proj, rows [0,D) q, [D,2D) k, [2D,3D) v
    WMat out, out_c;     // attn.to_out.0, attn.to_add_out (absent when pre_only)
    WMat ff1, ff2, ffc1, ffc2;
    int mod_off = 0, mod_c_off = 0;   // first row of norm1.linear / norm1_context.linear in the net's modulation matrix
    bool pre_only = false;            // context_pre_only (last transformer block): context gives keys / values only
    bool single = false;              // SD3SingleTransformerBlock: no context stream at all (ControlNet with joint_attention_dim = None)
    // qk_norm = "rms_norm": RMSNorm weights [head_dim] of the image / context queries and keys
    float *nq = nullptr, *nk = nullptr, *naq = nullptr, *nak = nullptr;
    // dual_attention_layers: attn2, self-attention over the image tokens (norm1.linear has 9 chunks: shift2 / scale2 / gate2 last)
    bool dual = false;
    WMat qkv2, out2;
    float *nq2 = nullptr, *nk2 = nullptr;
};
struct Sd3NetW {
    bool built = false, single = false;
    int layers = 0, pos_max = 0;
    WMat pe, pe_in, t1, t2, p1, p2, ctx_emb, mod, proj_out;
    ConvW down_proj;                  // ControlNet only: Conv2d(6, 3, 3, padding 1) of encode_support_pair
    float* pos = nullptr;             // pos_embed.pos_embed [pos_max * pos_max][D] fp32
    std::vector<Sd3BlockW> blocks;
    std::vector<WMat> zero;           // controlnet_blocks
    int mod_rows = 0, norm_out_off = 0;
};

struct Param {
    std::string name;
    std::vector<int64_t> shape;
    int kind = 0;  // 0 fp32 vector, 1 matrix rows (linear / conv)
    // vector destination
    float* vdst = nullptr;
    bool geglu_vec = false;
    int geglu_half = 0;  // 4C for GEGLU permutation
    // matrix destination
    WMat* mat = nullptr;
    int row_off = 0;
    bool conv = false;  // OIHW source
    char init = 'w';    // recipe class for pd_init_random_weights: w, b, g(amma), e(beta)
    bool loaded = false;
    int group = 0;      // 0: UNet + ControlNet (needed to sample), 1: VAE decoder, 2: text transformer, 3: SD3 networks
};

struct Act {
    void* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    int dt = DT_F32;
    long long rows() const { return (long long)B * H * W; }
    size_t bytes() const { return (size_t)rows() * C * dt_size(dt); }
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, top = 0, peak = 0;
    bool dry = false;
    bool overflow = false;   // an allocation ran past cap: every enqueue refuses to launch until the arena is reset
    void* alloc(size_t bytes) {
        size_t a = (top + 255) & ~(size_t)255;
        top = a + bytes;
        if (top > peak) peak = top;
        if (dry) return reinterpret_cast<void*>((size_t)0x1000 + a);
        if (top > cap) { overflow = true; return base; }   // never hand out memory beyond the workspace
        return base + a;
    }
    size_t mark() const { return top; }
    void release(size_t m) { top = m; }
};

struct KVSlot { void* K = nullptr; void* VT = nullptr; void* P = nullptr; };   // P: K / V^T in st_tail.hip's fragment order (fused blocks)

struct Session {
    bool active = false;
    pd_sample_args a{};
    int Bf = 0;
    int S = 0;
    std::vector<int64_t> timesteps;  // ascending (ddim_timesteps)
    std::vector<int64_t> custom_ts;  // copy of pd_sample_args.timesteps (descending), empty = uniform grid
    std::vector<float> alphas, alphas_prev, sigmas, sqrt_1m;
    std::vector<float> scales_step;  // [S][13]
    // device state
    float* x_state = nullptr;   // [B, HW, 8] fp32
    float* x_in = nullptr;      // [Bf, HW, 8] fp32
    float* pred_x0 = nullptr;   // [B, HW, C]
    float* eps_g = nullptr;     // [B, HW, C]
    float* noise = nullptr;     // [S][B, C, HW] or null
    void* ctx = nullptr;        // [Bf, L, Dpad] compute type
    Act hint;                   // guided_hint [Bf, h, w, C0]
    std::vector<KVSlot> kv_u, kv_c;
    std::vector<float*> emb_u, emb_c;  // per ResBlock [rows, Cout]
    int emb_rows = 0;
    float* per_step = nullptr;  // optional [S+1][B,C,h,w]
    Act control[PD_NUM_CONTROL];
    size_t session_top = 0;
    // shared CFG front (pd_engine::forward_eps): both halves of the batch see the same guided_hint; this evaluation runs the UNet's /
    // the ControlNet's layers in front of the first cross-attention once for both halves
    bool hint_shared = false, share_u = false, share_c = false;
    bool cn_cond_only = false;   // guess mode under guidance: this evaluation runs the ControlNet on the conditional half only (run_controlnet)
};

struct pd_engine {
    pd_config cfg{};
    int device = 0;
    int ncu = 256;      // compute units of the device (hipDeviceProp_t::multiProcessorCount): the tile / split-K rules scale with it
    hipStream_t stream = nullptr;
    void* comm = nullptr;   // ncclComm_t of pd_comm_init (comm.cpp); the final all-gather of a sharded batch
    int comm_world = 1, comm_rank = 0;
    bool f32 = false;   // fp32 storage of activations and weights (PD_PREC_F32 and PD_PREC_F16X2)
    int P = DT_BF16;    // MFMA precision code handed to the contraction launchers: T, or PREC_F16X2
    int T = DT_BF16;  // compute (MFMA operand) type: DT_BF16, DT_F16 or DT_F32
    int S = DT_BF16;  // residual-stream type: T, or DT_F32 with stream_f32
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    NetW unet, cnet;
    VaeW vae;
    TextW text;
    // captured step loops (option "graph"): key = everything a step's kernel arguments depend on
    struct GraphEntry { uint64_t key; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    void clear_graphs();
    int run_steps_graph();
    int reg_group = 0;
    std::vector<void*> owned;  // device allocations (weights)
    bool alloc_failed = false; // a hipMalloc in dmalloc() failed (reported by build() / the bench hooks)
    int check_arena();         // non-zero (with the error set) when a workspace allocation overflowed
    size_t weight_bytes = 0;
    Arena arena;
    // second context for the ControlNet pass (own stream / workspace / GroupNorm scratch), see forward_eps
    Arena arena2;
    hipStream_t stream2 = nullptr;
    double* gn_partial2 = nullptr;
    int* tile_cnt2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool cn_pending = false;
    bool opt_two_streams = true;
    bool opt_cfg_share = true;   // option "cfg_share": the layers in front of the first cross-attention once per CFG pair (forward_eps)
    void swap_context();
    int join_controlnet();
    Session ses;
    int verbose = 0;
    bool opt_splitk_fused = false; // split-K sums + epilogue run in the last-arriving slice instead of a finalize kernel
    int opt_splitk_max = 16;   // (8 until the end of round 4: the 8x8 level at batch 1 -- M = 128, 8 tiles -- then ran on 64 blocks; 16: +1.5 % at batch 1, nothing at batch 8)
    int opt_splitk_big = 0;       // split-K conv3x3 on 256 x 160 tiles where that still fills the chip (option "splitk_big")
    int opt_splitk_tiles = 256;   // (x ncu / 256 at build)   // split K when the 128x160 tile grid has fewer blocks than this (one tile per CU needs no split:
                                  // 384 sent the 16x16 level's 5120 -> 1280 feed-forward-out GEMM -- 256 tiles -- to split-K 2 + a finalize pass, 86 us; the ring
                                  // GEMM takes it unsplit in ~60: +0.5 % end to end)
    bool opt_attn_legacy = false;  // debug: single-buffered attention kernel
    bool opt_wide = true;      // 256 x 320 GEMM tiles for large-M linear layers
    int opt_dense_tiles = 128;
    bool opt_gn_single = true; // GroupNorm as one LDS-slab kernel where a sample's group bundle fits (32x32 and below)
    bool opt_graph = false;    // pd_ddim_sample: capture the step loop in a hipGraph and replay it on later calls with the same arguments
    int opt_ring = 80;         // linear layers with at most this many K steps (0: off) take gemm_ring.hip's persistent LDS-DMA ring kernel
    int opt_ring_tile = -1;    // its tile: -1 auto, 0 = 128 x 160, 1 = 256 x 160
    int opt_ring_pp = 1;       // its ping-pong form where it measures faster (long K, or one 256-row tile per CU)
    int opt_ring_geglu = 1;    // GEGLU projections too (256 x 160 on 8 x 1 waves)
    int opt_ring_small = 4;    // small-M linear layers on 64 x 80 ring tiles instead of split-K (option "ring_small"): 0 off, d: where the 128 x 160 grid fills at most 1 / d of the chip
    int opt_short_k = 20;      // linear layers with at most this many K steps: 8-wave 128x160 tile at 16 waves per CU
    bool opt_patch_split = true;      // LDS-patch conv with the channel chunks split over 2-4 slices (16x16 level)
    int opt_patch_split_tiles = 64;
    int opt_patch_split_min = 4;      // ... each slice at least this many 128-byte channel chunks.  (min 1 / tiles 32 measured +2.3 % at batch 1 and
                                      // neutral at batch 8, but re-associates the sums of the 256x256 parity fixture's convs: its step-0 latent error
                                      // moves from 7.4e-4 to 1.02e-3 -- same arithmetic, other rounding pattern -- so the defaults stay)
    int opt_patch_split_fill = 256;   // slices are chosen to reach about this many blocks
    int opt_dense_k = 40;      // linear layers with at most this many K steps and >= opt_dense_tiles tiles: one 8-wave block per CU, no split-K
    bool opt_bigtile = true;  // 256-row GEMM tiles where the grid still fills the chip
    bool opt_tile192 = true;  // 256 x 192 GEMM tiles for widths that divide by 192 but not by 160 (MMDiT)
    // apply GroupNorm(+SiLU) inside the patch conv's staging.  Measured neutral-to-negative in round 1 (the SiLU VALU work
    // lands on the MFMA waves and the halo is transformed 1.27x redundantly), so it is off by default.
    bool opt_gn_fuse = false;
    bool opt_patch = true;  // use the LDS-patch conv3x3 kernel where eligible
    bool opt_patch2 = true; // 2-byte modes: the wave-specialised second-generation patch kernel (conv_patch2.hip)
    bool opt_patch4 = true; // 2-byte modes: the 4-wave patch kernel (conv_patch4.hip) for unsplit launches
    int opt_patch2_tiles = 768;   // ... for launches of at least this many blocks (and every split-K patch launch)
    long long launches = 0;
    long long gn_from_slabs = 0;   // GroupNorm launches that summed a split-K GEMM's slabs (stat "gn_from_slabs")
    long long ring_launches = 0;   // launches that took gemm_ring.hip (stat "ring_launches")
    // optional per-launch timing (bench.py roofline leg): HIP events around every contraction launch
    struct ProfRec { hipEvent_t a, b; int klass; double flops; int M, N, K, taps; };
    bool profiling = false;
    float prof_overhead_ms = 0.f;   // elapsed time of an EMPTY event bracket on this stream (calibrated when profiling is switched on)
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    hipEvent_t next_event();
    void prof_begin(ProfRec& r, int klass, double flops);
    void prof_end(ProfRec& r);
    double* gn_partial = nullptr;  // scratch for GroupNorm partial sums
    int* tile_cnt = nullptr;       // split-K arrival counters (kTileCnt ints, zero between GEMMs), one set per stream
    static constexpr int kTileCnt = 4096;
    size_t gn_partial_cap = 0;

    // construction
    int build();
    void* dmalloc(size_t bytes);
    void make_mat(WMat& m, int n_rows, int k_logical, int taps, int cin, bool bias, bool geglu = false);
    void reg_mat(const std::string& name, std::vector<int64_t> shape, WMat* m, int row_off, bool conv);
    void reg_vec(const std::string& name, int n, float** dst, char init);
    void reg_bias(const std::string& name, WMat* m, int off, int n, bool geglu = false);
    void build_conv(const std::string& prefix, ConvW& c, int cin, int cout, int k, int stride);
    void build_res(const std::string& prefix, ResW& r, int cin, int cout, NetW& net);
    void build_st(const std::string& prefix, STW& s, int ch, NetW& net);
    void build_encoder(const std::string& prefix, NetW& net);
    void build_middle(const std::string& prefix, NetW& net);
    void build_vres(const std::string& prefix, ResW& r, int cin, int cout);
    void build_vae();
    void build_text();
    int text_forward(const int* ids_dev, int B, float* out_dev, int clip_skip);
    int vae_forward(const float* latents_dev, int B, int h, int w, float* out_dev);
    int vae_attention(const Act& x, Act& out);

    // SD3 / MMDiT path (sd3.cpp)
    pd_sd3_config sd3{};
    Sd3NetW sd3_tr, sd3_cn;
    std::vector<hipEvent_t> sd3_ev;   // ControlNet residual i written (recorded on the second stream)
    struct Sd3Io {   // device pointers of one evaluation
        const float *latents, *context, *pooled, *cond, *pair;
        const float* cn_pooled;   // ControlNet's pooled projections; null: zeros
        const float* t_host;
        int B, H, W, S;
        float scale;
    };
    void build_sd3_net(const std::string& prefix, Sd3NetW& net, bool controlnet);
    int sd3_embed(Sd3NetW& net, const Sd3Io& io, bool controlnet, Act& hs, Act& c, Act& modbuf);
    int sd3_block(const Sd3BlockW& b, Act& x, Act& c, const Act& modbuf, const Act& qk, const Act& vt, const Act* pre_add = nullptr);
    int sd3_forward(const Sd3Io& io, float* v_out_dev, int control_index, float* control_out_dev);
    // one-shot extras of the next gemm() call (MMDiT: gated residual, joint-buffer row remap)
    // A split-K conv whose only consumer is a single-kernel GroupNorm (resblock conv1 -> norm2 at the 16x16 / 8x8 levels) leaves its
    // slabs un-summed: `allow` is set by the caller of gemm(), `active` by gemm() when it did skip the finalize pass; the slabs
    // stay on the workspace stack until the caller's own mark is released.
    struct SlabDefer { bool allow = false, active = false; const float* slabs = nullptr; int nslab = 0; const float* bias = nullptr; const float* rowvec = nullptr; int rowvec_stride = 0; };
    SlabDefer* gx_defer = nullptr;   // one-shot, like gx: consumed (and cleared) by the next gemm()
    int opt_slab_gn = 1;             // the fusion above (option "slab_gn")
    struct GemmExtra { const float* a_scale = nullptr; const float* c_scale = nullptr; const float* gate = nullptr; int gate_stride = 0, c_sample_rows = 0, c_row_off = 0, vt_tok_off = 0, a_sample_rows = 0, a_row_off = 0; } gx;
    int opt_sd3_fp8 = 0;       // 0 off, 1: the AdaLN-fed projections, 2: also the feed-forward-out projections (e4m3 GELU output under a norm bound)  // SD3 path: QKV and feed-forward-in projections in PREC_FP8 (e4m3 operands, per-row scales)
    bool sd3_fp8_dirty = true;
    int sd3_quantize();        // (re)builds the e4m3 weights of those layers after a weight change
    bool opt_gemv = true;     // Linear over <= 4 fp32 rows with a wide output (MMDiT modulation): weight-streaming kernel instead of a GEMM tile

    // weights
    int load(const char* name, const void* data, const int64_t* shape, int ndim, int dtype);
    int init_random(uint64_t seed);
    int upload_vec(float* dst_dev, const float* src, int n, bool geglu, int half);
    int upload_rows(WMat& m, int row_off, const float* src, int rows, bool conv);

    // primitive ops (enqueue on stream; honour arena.dry)
    Act new_act(int B, int H, int W, int C, int dt);
    int gemm(const WMat& m, const Act& in, Act& out, int taps_stride, int ups, int act, float scale, const Act* R,
             const float* rowvec, int rowvec_stride, bool a_silu, void* VT, int vt_begin, int vt_ld, int ldc_override = 0,
             const float* gn_coef = nullptr, bool gn_silu = false, const LnStats* ln_in = nullptr, LnStats* ln_out = nullptr);
    int fold_layernorms();     // (re)builds the folded weights of every transformer block after a weight change
    bool ln_dirty = true;
    int opt_ln_fuse = -1;      // -1: on in the 2-byte modes, off in the fp32-storage modes; 0 / 1: forced
    bool opt_st_fuse = true;   // 320-channel SpatialTransformer blocks: one kernel for everything after self-attention (2-byte modes)
    bool st_tail_on(const STW& s, int rows_per_sample) const;
    int gn_stats(const Act& x, int& nchunk);
    int conv_gn(const ConvW& c, const Act& x, Act& out, const float* g, const float* b, float eps, bool silu, const Act* R,
                const float* rowvec, int rowvec_stride, SlabDefer* out_defer = nullptr, const SlabDefer* in_slabs = nullptr);
    int conv(const ConvW& c, const Act& in, Act& out, int act = 0, float scale = 1.f, const Act* R = nullptr,
             const float* rowvec = nullptr, int rowvec_stride = 0, int ups = 0);
    int groupnorm(const Act& x, Act& y, const float* g, const float* b, float eps, bool silu, const SlabDefer* from_slabs = nullptr);
    int layernorm(const Act& x, Act& y, const float* g, const float* b);
    int resblock(const ResW& r, const Act& x, Act& out, const float* embrow, int emb_stride);
    int transformer(const STW& s, const Act& x, Act& out, const KVSlot& kv, int out_B = 0);
    int repeat_act(const Act& src, int reps, Act& dst);
    static Act batch_view(const Act& a, int first, int count);
    int attention(const void* Q, int ldq, const void* K, int ldk, const void* VT, int vt_ld, void* O, int ldo, int B, int Nq,
                  int Nk, int C, int heads = 0, bool causal = false, long long q_bs = 0, long long k_bs = 0, long long o_bs = 0);

    // networks
    int run_controlnet(const Act& x_in, int emb_row, int emb_stride, const float* scales);
    int run_unet(const Act& x_in, int emb_row, int emb_stride, bool only_mid, Act& eps);
    int forward_eps(int emb_row, int emb_stride, const float* scales, Act& eps);

    // sessions
    int ensure_arena(int Bf, int h, int w, int rows, bool per_step);
    int session_setup(const pd_sample_args& a, const int64_t* t_rows, int n_rows, bool per_sample_t, bool want_per_step);
    int compute_emb(NetW& net, std::vector<float*>& tabs, const int64_t* t, int n, int row0);
    int begin(const pd_sample_args* a, bool want_per_step);
    int step(int i);
    int make_schedule(int steps, float eta, std::vector<int64_t>& ts, std::vector<float>& a, std::vector<float>& ap,
                      std::vector<float>& sg, std::vector<float>& s1m, const int64_t* custom_desc = nullptr);
};
