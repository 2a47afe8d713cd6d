/* pdengine_ops.h -- per-operator parity hooks of libpdengine.so (test surface, not the product path).
 *
 * Each hook runs exactly one kernel family of the hot path on host fp32 arrays in the reference's
 * layouts, in the engine's precision mode, so tests/ can compare it with the oracle:
 *   pd_op_conv2d     nn.Conv2d 3x3/1x1 (stride 1/2, fused nearest-x2 upsample)   openaimodel.py:90-159,200-240
 *   pd_op_linear     nn.Linear / GEGLU / SiLU->Linear (emb_layers)                 attention.py:49-76, openaimodel.py:217-223
 *   pd_op_linear_fp8 the e4m3 form of nn.Linear used by the SD3 path's sd3_fp8 option
 *   pd_op_groupnorm  GroupNorm32 (+SiLU)                                           util.py:217-219, attention.py:88-89
 *   pd_op_layernorm  nn.LayerNorm                                                  attention.py:263-265
 *   pd_op_attention  softmax(q k^T dh^-0.5) v, heads from the engine config        attention.py:171-193
 *   pd_op_spatial_transformer  one SpatialTransformer block of the loaded networks  attention.py:271-275,321-340
 *   pd_op_time_embed timestep_embedding + the time_embed MLP of a loaded network    util.py:154-174, openaimodel.py:526-531
 */
#ifndef PDENGINE_OPS_H
#define PDENGINE_OPS_H
#include "pdengine.h"
#ifdef __cplusplus
extern "C" {
#endif
int pd_op_conv2d(pd_engine* e, const float* x, const float* w, const float* bias, const float* residual, int B, int Cin, int H,
                 int W, int Cout, int k, int stride, int upsample, int act_silu, float scale, int stream_out, float* y);
int pd_op_linear(pd_engine* e, const float* x, const float* w, const float* bias, int M, int K, int N, int geglu, int a_silu,
                 float* y);
/* the SD3 path's PREC_FP8 linear layer (option sd3_fp8): e4m3 operands quantised on the device with one scale per row of x
 * and of w, block-scaled MFMA, scales applied in the epilogue; act 0 or 4 (tanh-GELU).  2-byte engine modes only. */
int pd_op_linear_fp8(pd_engine* e, const float* x, const float* w, const float* bias, int M, int K, int N, int act, float* y);
int pd_op_groupnorm(pd_engine* e, const float* x, const float* gamma, const float* beta, int B, int C, int H, int W, float eps,
                    int silu, float* y);
int pd_op_layernorm(pd_engine* e, const float* x, const float* gamma, const float* beta, int rows, int C, float* y);
int pd_op_attention(pd_engine* e, const float* q, const float* k, const float* v, int B, int Nq, int Nk, int C, float* o);
/* SpatialTransformer.forward of the block whose weights were loaded under `prefix` (reference state-dict prefix ending in '.'),
 * x [B, C, H, W], context [B, context_len, context_dim], y [B, C, H, W]; the same code path as a sampling step (2-byte modes at
 * 320 channels: self-attention + the fused tail kernel). */
int pd_op_spatial_transformer(pd_engine* e, const char* prefix, const float* x, const float* context, int B, int H, int W, float* y);
/* timestep_embedding(t, model_channels) (util.py:154-174: [cos | sin], max_period 10000, frequencies i / half) -> temb [n][model_channels],
 * and time_embed(temb) = Linear -> SiLU -> Linear of the loaded network (net 0: UNet, 1: ControlNet; openaimodel.py:526-531) -> emb
 * [n][4 * model_channels]; exactly what the sampler computes once per call for all its steps (pd_engine::compute_emb). */
int pd_op_time_embed(pd_engine* e, int net, const int64_t* t, int n, float* temb, float* emb);
#ifdef __cplusplus
}
#endif
#endif
