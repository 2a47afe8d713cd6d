/* pdengine.h -- C ABI of the MI355X-native Prompt-Diffusion DDIM sampling engine.
 *
 * This is the drop-in boundary for the reference's hot path: the work the
 * reference does, per denoising step, inside
 *     DDIMSampler.sample / ddim_sampling / p_sample_ddim      cldm/ddim_hacked.py:55-234
 *     ControlLDM.apply_model                                   cldm/cldm.py:369-382
 *     ControlNet.forward / ControlledUnetModel.forward          cldm/cldm.py:302-325, :23-45
 * and, under diffusers naming, inside PromptDiffusionPipeline.__call__'s loop
 *     pipeline_prompt_diffusion.py:1210-1290 + promptdiffusioncontrolnet.py:188-391.
 *
 * Plain C: opaque handle, pointers and sizes only; no torch types.  All tensors that cross
 * the boundary are float32 in the reference's own layouts (NCHW images/latents,
 * [B, L, D] text context).  Every entry point returns 0 on success, non-zero on error and
 * never throws; pd_last_error() returns the thread-local message of the last failure.
 * One engine <-> one GPU <-> one HIP stream; calls on one engine must be serialised by the
 * caller (the reference pipeline is not thread-safe either, pipeline_prompt_diffusion.py:1065).
 */
#ifndef PDENGINE_H
#define PDENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_ABI_VERSION 2

/* arithmetic mode of the engine */
#define PD_PREC_BF16 0 /* bf16 MFMA operands, fp32 accumulate / norm statistics / softmax */
#define PD_PREC_F32 1  /* fp32 MFMA (v_mfma_f32_16x16x4_f32): bit-faithful fp32 arithmetic */
#define PD_PREC_F16 2  /* fp16 MFMA operands (the reference's own GPU dtype, README.md:44-45 torch_dtype=torch.float16;
                          same MFMA rate as bf16, 3 more mantissa bits), fp32 accumulate / norm statistics / softmax */
#define PD_PREC_F16X2 3 /* fp32 storage; every MFMA operand split into fp16 hi + lo in registers, three fp16 MFMAs per
                           8 K-elements (hi*hi + hi*lo + lo*hi): ~22-bit operands, fp32-class results at ~5x the fp32 MFMA rate */

/* where caller-owned I/O buffers live */
#define PD_MEM_HOST 0
#define PD_MEM_DEVICE 1

/* dtype tags for pd_load_weights */
#define PD_DT_F32 0
#define PD_DT_F16 1
#define PD_DT_BF16 2

#define PD_MAX_LEVELS 8
#define PD_NUM_CONTROL 13 /* 12 input blocks + middle, cldm/cldm.py:313-323 */

typedef struct pd_engine pd_engine;

/* Hyper-parameters = models/cldm_v15.yaml:30-62 (ControlNet and UNet share them). */
typedef struct pd_config {
    int32_t in_channels;      /* 4 */
    int32_t out_channels;     /* 4 */
    int32_t hint_channels;    /* 6: example pair, cldm/cldm.py:147 */
    int32_t query_channels;   /* 3: query image, cldm/cldm.py:166 */
    int32_t model_channels;   /* 320 */
    int32_t num_levels;       /* 4 */
    int32_t channel_mult[PD_MAX_LEVELS]; /* 1,2,4,4 */
    int32_t num_res_blocks;   /* 2 */
    int32_t num_attn_res;     /* 3 */
    int32_t attention_resolutions[PD_MAX_LEVELS]; /* 4,2,1 */
    int32_t num_heads;        /* 8 */
    int32_t context_dim;      /* 768 */
    int32_t context_len;      /* 77 */
    int32_t hint_widths[7];   /* 16,16,32,32,96,96,256: cldm/cldm.py:147-163 */
    int32_t timesteps;        /* 1000 */
    double linear_start;      /* 0.00085 (double: the schedule is derived in float64, util.py:22-25) */
    double linear_end;        /* 0.0120 */
    int32_t precision;        /* PD_PREC_* */
    int32_t stream_f32;       /* PD_PREC_BF16 / PD_PREC_F16: keep the residual stream (block outputs) in fp32 */
    /* first-stage KL-VAE decoder (models/cldm_v15.yaml:64-85).  vae_ch = 0: not built */
    int32_t vae_ch;           /* 128 */
    int32_t vae_num_levels;   /* 4 */
    int32_t vae_ch_mult[PD_MAX_LEVELS]; /* 1,2,4,4 */
    int32_t vae_num_res_blocks; /* 2 */
    int32_t vae_out_ch;       /* 3 */
    double scale_factor;      /* 0.18215, cldm_v15.yaml:17 */
    /* cond-stage CLIP text transformer (ldm/modules/encoders/modules.py:88-131).  Width = context_dim,
     * length = context_len.  text_layers = 0: not built */
    int32_t text_vocab;       /* 49408 */
    int32_t text_layers;      /* 12 */
    int32_t text_heads;       /* 12 */
    int32_t text_ff;          /* 3072 */
    int32_t reserved[2];
} pd_config;

/* Arguments of one sampling call (replaces DDIMSampler.sample's arguments,
 * cldm/ddim_hacked.py:55-79, and the loop state of PromptDiffusionPipeline.__call__). */
typedef struct pd_sample_args {
    int32_t batch;            /* B images; the engine runs the CFG-doubled batch 2B when use_cfg */
    int32_t h, w;             /* latent height/width = image/8 */
    int32_t steps;            /* S */
    float eta;                /* ddim eta */
    float cfg_scale;          /* unconditional_guidance_scale / guidance_scale */
    int32_t use_cfg;          /* (L): unconditional_conditioning is not None, ddim_hacked.py:188;
                                 (D): guidance_scale > 1, pipeline_prompt_diffusion.py:878 */
    int32_t guess_mode;       /* (D) :1220-1224,:1248-1253: ControlNet sees the cond half only */
    int32_t only_mid_control; /* cldm/cldm.py:38 */
    float temperature;        /* ddim_hacked.py:230 */
    int32_t mem;              /* PD_MEM_* of every pointer below */
    const float* x_T;         /* [B, in_ch, h, w] initial latents (required) */
    const float* ctx_cond;    /* [B, L, D] */
    const float* ctx_uncond;  /* [B, L, D] (use_cfg) */
    const float* pair;        /* [B, hint_ch, 8h, 8w] example pair */
    const float* query;       /* [B, query_ch, 8h, 8w] query image */
    const float* pair_uncond; /* optional: unconditional example pair (NULL = same as pair) */
    const float* query_uncond;/* optional */
    const float* control_scales;      /* [13] or NULL (= all 1.0), cldm/cldm.py:335,379 */
    const float* control_scales_step; /* optional [steps][13]: per-step scales in sampling order
                                         (controlnet_keep gating, pipeline :1196-1202,:1229-1235) */
    const float* noise;       /* eta > 0: [steps][B, in_ch, h, w] standard normal draws (required then: ddim_hacked.py:230) */
    const int64_t* timesteps; /* optional, HOST memory whatever `mem` says: [steps] custom DDIM timesteps in sampling order
                                 (strictly descending) replacing the uniform grid of make_ddim_timesteps -- the (D) pipeline's
                                 `timesteps=` argument (pipeline_prompt_diffusion.py:101-142) and diffusers' leading-spaced
                                 grid for step counts that do not divide 1000 */
    int32_t reserved[6];
} pd_sample_args;

const char* pd_last_error(void);
int pd_abi_version(void);

/* lifecycle (replaces create_model + load_state_dict + .to(device), cldm/model.py:12-28) */
int pd_engine_create(const pd_config* cfg, int device_id, pd_engine** out);
void pd_engine_destroy(pd_engine* e);

/* parameter registry: names/shapes are the reference checkpoint's (model.diffusion_model.*,
 * control_model.*; tool_add_control.py:36-45) */
int pd_param_count(pd_engine* e);
int pd_param_info(pd_engine* e, int index, const char** name, int32_t* ndim, int64_t shape[4]);
/* copy one tensor from HOST memory; the engine repacks it (NHWC taps, bf16, fused QKV). */
int pd_load_weights(pd_engine* e, const char* name, const void* data, const int64_t* shape,
                    int32_t ndim, int32_t dtype);
/* device-side seeded N(0, 1/fan_in)-style initialisation of every tensor (benchmarks only) */
int pd_init_random_weights(pd_engine* e, uint64_t seed);
/* number of UNet + ControlNet tensors not loaded yet (0 = ready to sample); the VAE decoder is counted separately */
int pd_weights_missing(pd_engine* e);
int pd_vae_weights_missing(pd_engine* e);

/* first-stage decode, LatentDiffusion.decode_first_stage (ldm/models/diffusion/ddpm.py:820-828) ->
 * AutoencoderKL.decode (ldm/models/autoencoder.py:89-92) -> Decoder.forward (ldm/modules/diffusionmodules/model.py:619-653):
 * latents [B, in_ch, h, w] -> images [B, vae_out_ch, 8h, 8w] in roughly [-1, 1] (fp32, NCHW).  Call after the sampling
 * session has ended; parameters are the checkpoint's first_stage_model.decoder.* / first_stage_model.post_quant_conv.* */
int pd_vae_decode(pd_engine* e, const float* latents, int32_t B, int32_t h, int32_t w, int32_t mem, float* images_out);

/* operator boundary: eps = apply_model(x, t, cond), cldm/cldm.py:369-382.
 *   x [Bf,in_ch,h,w], t [Bf] (int64), ctx [Bf,L,D], pair [Bf,hint_ch,8h,8w], query [Bf,q_ch,8h,8w],
 *   scales [13] or NULL.  eps_out [Bf,out_ch,h,w].  residuals_out (optional): the 13 scaled
 *   control tensors, NCHW, concatenated in list order (sizes from pd_control_shape). */
int pd_eps(pd_engine* e, const float* x, const int64_t* t, const float* ctx, const float* pair,
           const float* query, const float* scales, int32_t Bf, int32_t h, int32_t w, int32_t mem,
           float* eps_out, float* residuals_out);
int pd_control_shape(pd_engine* e, int index, int32_t h, int32_t w, int32_t* C, int32_t* H, int32_t* W);

/* the fused loop: begin + S steps + read-back; blocking (stream-synchronised on return).
 *   latents_out [B,in_ch,h,w]; per_step_out optional [S+1][B,in_ch,h,w] = x_inter incl. x_T
 *   (intermediates with log_every_t=1, ddim_hacked.py:143,174-176). */
int pd_ddim_sample(pd_engine* e, const pd_sample_args* args, int32_t mem_out, float* latents_out,
                   float* per_step_out);

/* stepwise form, for callbacks that inspect or replace latents between steps
 * (callback/img_callback ddim_hacked.py:171-172; callback_on_step_end pipeline :1275-1283) */
int pd_sample_begin(pd_engine* e, const pd_sample_args* args);
int pd_sample_step(pd_engine* e, int32_t i); /* i = 0..steps-1, asynchronous on the engine stream */
#define PD_GET_LATENTS 0
#define PD_GET_PRED_X0 1
#define PD_GET_EPS 2 /* guided noise prediction of the last step */
int pd_sample_get(pd_engine* e, int32_t what, int32_t mem, float* out);
int pd_sample_set_latents(pd_engine* e, int32_t mem, const float* latents);
/* unconditional_guidance_scale of the following steps: DDIMSampler.ddim_sampling's ucg_schedule (cldm/ddim_hacked.py:159-161)
 * replaces the scale before every p_sample_ddim */
int pd_sample_set_guidance(pd_engine* e, float scale);
/* guided eps at an arbitrary timestep for the current latents, no update: lets a host-side
 * scheduler (pipeline :1273 scheduler.step) drive the engine */
int pd_sample_eps_at(pd_engine* e, int64_t t, const float* scales13);
int pd_sample_end(pd_engine* e);

/* schedule exactly as DDIMSampler.make_schedule derives it (cldm/ddim_hacked.py:23-52):
 * fills timesteps[S] (ascending), alphas[S], alphas_prev[S], sigmas[S], sqrt_one_minus_alphas[S] */
int pd_make_schedule(pd_engine* e, int32_t steps, float eta, int64_t* timesteps, float* alphas,
                     float* alphas_prev, float* sigmas, float* sqrt_one_minus_alphas);

/* instrumentation */
int pd_synchronize(pd_engine* e);
void* pd_stream(pd_engine* e);              /* hipStream_t the engine launches on */
/* PD_MEM_DEVICE inputs: make the engine's (non-blocking) streams wait for the work already enqueued on `producer`, the
 * hipStream_t that wrote those buffers (NULL = the default stream) -- the engine-side half of what `.to(device)` ordering
 * gives the reference (everything on torch's current stream).  Call before handing device buffers over; outputs need no
 * counterpart because every call that fills a caller buffer synchronises the engine stream before it returns. */
int pd_wait_stream(pd_engine* e, void* producer);
/* Multi-GPU: images are independent through every denoising step, so a batch shards over the GPUs of a node with
 * no per-step exchange -- one process and one engine per GPU, exactly how the reference shards its evaluation set
 * (eval/distributed.py:25-27 init_process_group, eval/evaluate_gen.py:55-57 batch_ids[rank::world_size]).  The one collective
 * is an all-gather of the final latents over an RCCL communicator the engine owns (librccl is opened on first use):
 *   rank 0: pd_comm_new_id(id) and hands the 128 bytes to every rank by any host channel (file, pipe, MPI, torch.distributed);
 *   every rank: pd_comm_init(e, id, world, rank) (collective: returns once all ranks have joined), sample its shard, then
 *   pd_comm_all_gather(e, my_latents, all_latents, count, mem): all_latents[world][count] in rank order, equal `count` on
 *   every rank (pad ragged shards).  Without a communicator the gather is a copy (world 1). */
#define PD_COMM_ID_BYTES 128
int pd_comm_new_id(uint8_t id[PD_COMM_ID_BYTES]);
int pd_comm_init(pd_engine* e, const uint8_t id[PD_COMM_ID_BYTES], int32_t world, int32_t rank);
int pd_comm_world(pd_engine* e, int32_t* world, int32_t* rank);
int pd_comm_all_gather(pd_engine* e, const float* send, float* recv, int64_t count, int32_t mem);
int pd_comm_destroy(pd_engine* e); /* also done by pd_engine_destroy */
/* Tuning / instrumentation knobs (defaults are the measured best; tests and tools/ flip them for A/B runs):
 *   "verbose" (1: workspace sizes; 2: one stderr line per contraction launch saying which kernel family / tile / split-K count it takes),
 *   "profile" (HIP events around every contraction launch, see pd_profile_read),
 *   "two_streams" (ControlNet on a second stream beside the UNet encoder, default 1),
 *   "cfg_share" (classifier-free guidance feeds both halves of the doubled batch the same latent and timestep, ddim_hacked.py:189-192, and
 *   -- unless pair_uncond / query_uncond say otherwise -- the same example pair and query: the layers in front of the first
 *   cross-attention (conv_in, the first ResBlock, the first SpatialTransformer up to attn2.to_q, in the UNet and in the ControlNet)
 *   run once on the B samples the halves have in common instead of twice; exact, default 1; stat "cfg_shared" tells whether the
 *   last evaluation did: bit 0 UNet, bit 1 ControlNet; bit 2: guess mode under guidance ran the ControlNet on the conditional half only,
 *   as pipeline_prompt_diffusion.py:1220-1224 does, instead of on the doubled batch with the unconditional residuals zeroed afterwards),
 *   "graph" (pd_ddim_sample captures its step loop in a hipGraph and replays it on later calls with equal arguments, 0),
 *   "conv_patch" (LDS-patch conv3x3 kernel, 1), "conv_patch2" (its wave-specialised second generation in the 2-byte modes, 1), "patch4" (the
 *   fourth generation -- 4 waves per block, one per SIMD, 32x32x16 MFMAs, LDS-DMA operands -- for every unsplit 2-byte patch conv, 1;
 *   bit-identical to the others), "patch_split" / "patch_split_tiles" (that kernel with the channel chunks
 *   split over 2-4 slices when it has fewer tiles than CUs but at least this many, 1 / 64), "gn_fuse" (GroupNorm applied while the patch is staged, 0),
 *   "ln_fuse" (norm1 / norm2 of a transformer block folded into the to_q/k/v and attn2.to_q GEMMs, row statistics carried
 *   from the producing GEMM's epilogue: -1 = on in the 2-byte modes and off in the fp32-storage modes, 0 / 1 forced),
 *   "gn_single" (single-kernel LDS-slab GroupNorm where a sample's group bundle fits, 1), "gn_reg" (process-wide: its register-resident
 *   form -- one block per (sample, group), the group's slab in registers, one barrier -- for 2-byte tensors of 64 / 256 / 1024 pixels, 1),
 *   "big_tile" / "wide_tile" (256x160 / 256x320 GEMM tiles, 1), "dense_k" / "dense_tiles" (8-wave unsplit tile for
 *   linear layers with at most that many K steps, 40 / 128), "short_k" (8-wave 128x160 tile at 16 waves per CU for
 *   linear layers with at most that many K steps, 20), "splitk_tiles" (split K below this many tiles, 256),
 *   "splitk_fused" (in-kernel split-K finalize, 0), "tile192" (256x192 GEMM tile for widths that divide by 192 but not by
 *   160: the MMDiT's 1536 / 4608 / 6144, 1), "gemv" (weight-streaming kernel for a Linear over <= 4 fp32 rows with >= 4096
 *   outputs: the MMDiT modulation matrix, 1),
 *   "sd3_fp8" (SD3 path, 2-byte modes: 0 off; 1 the projections fed by an AdaLN output -- q/k/v of both streams, ff / ff_context
 *   net.0 -- take e4m3 operands with one scale per token and per output channel on the block-scaled K = 128 MFMA; 2 also the
 *   feed-forward-out projections, their input stored as e4m3 under a row bound; default 0),
 *   "attn_legacy" (single-buffered attention kernel, 0),
 *   "st_fuse" (320-channel SpatialTransformer blocks: st_front / st_tail fused kernels in the 2-byte modes, 1),
 *   "ring" / "ring_tile" / "ring_geglu" (gemm_ring.hip: linear layers over 2-byte operands with at most that many 64-element K steps
 *   take the persistent LDS-DMA ring GEMM, 80 / its tile -1 auto, 0 = 128x160, 1 = 256x160 / GEGLU projections too, 1; results are
 *   bit-identical to the igemm tiles'), "ring_small" (small-M linear layers -- at most a quarter chip of 128x160 tiles, the 8x8 level's M = 1024 --
 *   on 64x80 ring tiles without split-K slabs and a finalize pass: 0 off, d = where the 128x160 grid fills at most 1/d of the chip, 4), "ring_pp" (its ping-pong form -- two wave groups half a K step apart -- for K >= 2560 and for one
 *   256-row tile per CU, 1; bit-identical), "slab_gn" (a ResBlock conv1 that runs split-K hands its fp32 slabs to the single-kernel
 *   GroupNorm that reads them instead of running a finalize pass, 1; bit-identical), "patch_split_min" (patch-conv split-K: at
 *   least this many 128-byte channel chunks per slice, 4). */
int pd_set_option(pd_engine* e, const char* key, int64_t value);
/* "workspace_bytes", "weight_bytes", "launches" (engine launches, split-K finalize passes not counted), "ring_launches" / "gn_from_slabs"
 * (of which: gemm_ring.hip / GroupNorm fed by split-K slabs), "steps", "event_overhead_ns", "cfg_shared" (see option "cfg_share") */
int64_t pd_get_stat(pd_engine* e, const char* key);
/* Per-launch timing: while option "profile" is 1 the engine brackets every contraction launch with HIP
 * events on its stream.  klass 0 = igemm_kernel on a conv3x3, 1 = igemm_kernel / rgemm_kernel on a conv1x1/linear,
 * 2 = attention, 3 = the conv3x3 patch kernels (all generations), 4 = st_front / st_tail, -1 = all.  One bracket = one launch (split-K finalize excluded).
 * Returns summed device time, launch count and algorithmic FLOPs (2*M*N*K, logical channel counts).  The elapsed time of
 * a bracket around an empty one-block kernel, calibrated when "profile" is switched on (stat "event_overhead_ns"), is
 * taken off every bracket. */
int pd_profile_read(pd_engine* e, int32_t klass, double* total_ms, int64_t* n_launches, double* flops);
int pd_profile_dump(pd_engine* e, const char* csv_path); /* one row per profiled launch */
/* Micro-benchmark hook used by bench.py's roofline leg: times `iters` launches of the dominant
 * conv3x3 implicit-GEMM kernel (Cin->Cout at HxW, batch Bf) with HIP events on the engine
 * stream; returns average ms per launch in *ms. */
int pd_bench_conv3x3(pd_engine* e, int32_t Bf, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                     int32_t iters, float* ms);

/* Cond stage (replaces FrozenCLIPEmbedder.forward after tokenisation, ldm/modules/encoders/modules.py:118-128, and
 * text_encoder(ids)[0] in PromptDiffusionPipeline.encode_prompt, pipeline_prompt_diffusion.py:308-487): token ids
 * [B, context_len] int32 -> last_hidden_state [B, context_len, context_dim] fp32.  mem: PD_MEM_HOST / PD_MEM_DEVICE for both
 * buffers.  Needs the cond_stage_model.transformer.text_model.* weights (pd_text_weights_missing() == 0). */
int pd_text_encode(pd_engine* e, const int32_t* ids, int32_t B, int32_t mem, float* out);
/* clip_skip k > 0 (pipeline_prompt_diffusion.py:398-413): the hidden state after block text_layers - k, passed through
 * final_layer_norm (= hidden_states[-(k+1)] of transformers' CLIPTextModel); k = 0 is pd_text_encode */
int pd_text_encode_ex(pd_engine* e, const int32_t* ids, int32_t B, int32_t mem, int32_t clip_skip, float* out);
int pd_text_weights_missing(pd_engine* e);

/* Same for one Linear / conv1x1 layer ([M,K] x [N,K]^T, optional residual add) in isolation. */
int pd_bench_linear(pd_engine* e, int32_t M, int32_t K, int32_t N, int32_t residual, int32_t iters, float* ms);

/* ---------------------------------------------------------------------------------------------------------------
 * SD3 / MMDiT variant of the path.  Replaces, per denoising step,
 *     SD3PromptDiffusionModel.forward                       promptdiffusioncontrolnet_sd3.py:362-483
 *     self.transformer(..., block_controlnet_hidden_states)  promptdiffusioncontrolnetpipeline_sd3.py:1226-1234
 *     CFG + scheduler.step (FlowMatchEuler)                  promptdiffusioncontrolnetpipeline_sd3.py:1237-1243
 * The block arithmetic lives in diffusers (absent offline): PARITY UNPINNED, checked against oracle/sd3_oracle.py only.
 * Parameter names are diffusers' state-dict names under the prefixes "transformer." (SD3Transformer2DModel) and
 * "controlnet." (SD3PromptDiffusionModel).  The example-pair / query conditions arrive as VAE latents: pd_sd3_down_proj is the
 * Conv2d(6, 3) half of encode_support_pair (promptdiffusioncontrolnet_sd3.py:189-198); vae.encode stays with the caller.
 * Not built: use_pos_embed = False (the ControlNet fed pre-embedded tokens, which the reference's own pipeline never does),
 * joint_attention_kwargs / LoRA scale. */
typedef struct pd_sd3_config {
    int32_t in_channels;        /* 16 */
    int32_t out_channels;       /* 16 */
    int32_t patch_size;         /* 2 */
    int32_t heads;              /* 24 (SD3-medium); hidden = heads * head_dim */
    int32_t head_dim;           /* 64 */
    int32_t layers;             /* transformer blocks (24); the last one is context_pre_only */
    int32_t cn_layers;          /* ControlNet blocks (reference default 18, promptdiffusioncontrolnet_sd3.py:59); 0: no ControlNet */
    int32_t joint_dim;          /* joint_attention_dim 4096 */
    int32_t pooled_dim;         /* pooled_projection_dim 2048 */
    int32_t pos_embed_max_size; /* 192: side of the transformer's sin/cos table "transformer.pos_embed.pos_embed" */
    int32_t cn_pos_embed_max_size; /* the ControlNet's own table (promptdiffusioncontrolnet_sd3.py:102 default 96); 0: same */
    int32_t cn_zero_pooled;     /* force_zeros_for_pooled_projection (promptdiffusioncontrolnet_sd3.py:108, pipeline :1164-1168):
                                   the ControlNet sees zero pooled projections; 0: cn_pooled, or pooled when that is NULL */
    int32_t qk_norm;            /* 0: none; 1: "rms_norm" -- RMSNorm(head_dim, eps 1e-6) on the queries and keys of every head
                                   (attn.norm_q / norm_k / norm_added_q / norm_added_k), promptdiffusioncontrolnet_sd3.py:105,140 */
    uint32_t dual_mask;         /* bit i: transformer block i carries attn2, a second attention over the image tokens alone
                                   (dual_attention_layers, :104,141; norm1.linear then has 9 chunks) */
    uint32_t cn_dual_mask;      /* the same for the ControlNet's blocks */
    int32_t cn_single;          /* 1: the ControlNet consists of SD3SingleTransformerBlocks (joint_attention_dim = None, :147-160): no
                                   context_embedder, no context stream; its blocks see the image tokens alone */
} pd_sd3_config;

typedef struct pd_sd3_args {
    int32_t batch;            /* B: rows of every tensor below (<= 32 per call) */
    int32_t height, width;    /* latent size (multiples of patch_size) */
    int32_t context_len;      /* S tokens of `context` */
    int32_t mem;              /* PD_MEM_HOST / PD_MEM_DEVICE of all tensor pointers (timestep is always host) */
    float conditioning_scale; /* controlnet_conditioning_scale */
    const float* latents;     /* [B, C, H, W] */
    const float* timestep;    /* [B] (sigma * 1000), host */
    const float* context;     /* [B, S, joint_dim]  encoder_hidden_states */
    const float* pooled;      /* [B, pooled_dim]    pooled_projections */
    const float* cond;        /* [B, C, H, W] controlnet_cond latents, or NULL: transformer alone */
    const float* pair;        /* [B, C, H, W] controlnet_example_pair_cond latents */
    const float* cn_pooled;   /* [B, pooled_dim] controlnet_pooled_projections, or NULL (see cn_zero_pooled) */
    int64_t reserved[3];
} pd_sd3_args;

/* Registers the SD3 networks' parameters on an existing engine (any pd_config; the UNet path keeps working) and allocates
 * their weights.  Once per engine. */
int pd_sd3_configure(pd_engine* e, const pd_sd3_config* cfg);
int pd_sd3_weights_missing(pd_engine* e);
/* One evaluation: velocity [B, out_channels, H, W] (fp32, args->mem) = transformer(latents | ControlNet residuals). */
int pd_sd3_forward(pd_engine* e, const pd_sd3_args* args, float* v_out);
/* ControlNet alone (the model-level call): residual i as [B, (H/p)(W/p), hidden] fp32, i in [0, cn_layers)
 * (controlnet_block_samples); pooled projections = cn_pooled, or pooled when that is NULL -- cn_zero_pooled is the
 * pipeline's choice and does not apply here. */
int pd_sd3_control(pd_engine* e, const pd_sd3_args* args, int32_t index, float* out);
/* The whole loop: sigmas[steps + 1] (host, descending, last = 0 for a full schedule); guidance > 1 runs the doubled batch
 * [negative ; positive]: context / pooled / cn_pooled then hold 2B rows, latents / cond / pair B rows.
 * step_scales: per-step conditioning scale [steps] (host) = controlnet_conditioning_scale * controlnet_keep[i]
 * (pipeline :1155-1162, :1202-1208), or NULL: args->conditioning_scale at every step; a step with scale 0 skips the
 * ControlNet (its residuals would be all zero).  latents_out: [B, C, H, W]. */
int pd_sd3_sample(pd_engine* e, const pd_sd3_args* args, const float* sigmas, int32_t steps, float guidance,
                  const float* step_scales, float* latents_out);
/* SD3PromptDiffusionModel.down_proj (promptdiffusioncontrolnet_sd3.py:114, :189-194): Conv2d(6, 3, kernel 3, padding 1) over
 * pair = cat([cond, gt], 1) [B, 6, H, W] -> out [B, 3, H, W] (fp32, `mem` as in pd_sd3_args). */
int pd_sd3_down_proj(pd_engine* e, const float* pair, int32_t B, int32_t H, int32_t W, int32_t mem, float* out);

#ifdef __cplusplus
}
#endif
#endif /* PDENGINE_H */
