// Shared declarations of the pdengine HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// Storage / MFMA operand types.  DT_BF16 and DT_F16 share one byte-level data path (2-byte elements); the engine's
// compute type T is one of the three and selects the MFMA instruction and the converts.
enum PdDType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2, DT_FP8 = 4 };   // DT_FP8: OCP e4m3fn bytes + one fp32 scale per row
static inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : dt == DT_FP8 ? 1 : 2; }
// MFMA precision codes handed to the contraction launchers: the operand PdDType, or PREC_F16X2 -- fp32 storage like
// DT_F32, but every operand is split into fp16 hi + lo halves in registers and the product runs as two fp16 MFMAs
// (all four hi/lo cross terms): ~22-bit operands at a quarter of the fp16 MFMA rate, 4x the fp32 MFMA rate.
constexpr int PREC_F16X2 = 3;
// PREC_FP8 (== DT_FP8): both operands e4m3 with per-row scales (A: per token, W: per output channel), one
// v_mfma_scale_f32_16x16x128_f8f6f4 (block scales 1.0) per accumulator and 128-byte K step -- twice the f16 MFMA rate at
// half the LDS bytes per FLOP; the scales multiply the accumulator in the epilogue.  Linear layers of the SD3 path only.
constexpr int PREC_FP8 = 4;
constexpr bool prec_f32_storage(int P) { return P == DT_F32 || P == PREC_F16X2; }

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#if defined(__HIPCC__)
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) instead of two converts + shift + or
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ float h2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // v_cvt_f16_f32, RNE
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
// 16-bit flavour helpers: DT is DT_BF16 or DT_F16 (compile-time in the kernels that are templated on it)
template <int DT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (DT == DT_F16) return pack2h(lo, hi); else return pack2bf(lo, hi);
}
template <int DT> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
    if constexpr (DT == DT_F16) {
        const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
        lo = (float)h[0]; hi = (float)h[1];
    } else {
        lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
    }
}
template <int DT> __device__ __forceinline__ uint16_t cvt16(float f) {
    if constexpr (DT == DT_F16) return f2h(f); else return f2bf(f);
}
template <int DT> __device__ __forceinline__ float cvt32(uint16_t v) {
    if constexpr (DT == DT_F16) return h2f(v); else return bf2f(v);
}
// runtime-tagged forms (dt is wave-uniform: a kernel argument)
__device__ __forceinline__ uint16_t cvt16_rt(float f, int dt) { return dt == DT_F16 ? f2h(f) : f2bf(f); }
__device__ __forceinline__ float cvt32_rt(uint16_t v, int dt) { return dt == DT_F16 ? h2f(v) : bf2f(v); }
// 8 packed 16-bit values (one uint4) <-> 8 floats
template <int DT> __device__ __forceinline__ void unpack8(const uint4& r, float* f) {
    unpack2<DT>(r.x, f[0], f[1]); unpack2<DT>(r.y, f[2], f[3]); unpack2<DT>(r.z, f[4], f[5]); unpack2<DT>(r.w, f[6], f[7]);
}
template <int DT> __device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 u;
    u.x = pack2<DT>(f[0], f[1]); u.y = pack2<DT>(f[2], f[3]); u.z = pack2<DT>(f[4], f[5]); u.w = pack2<DT>(f[6], f[7]);
    return u;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;  // single instruction: hipcc otherwise canonicalises MFMA outputs with extra v_max before fmaxf
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): ~12 instructions instead of
// libm erff's ~45; used where the result is rounded to bf16 anyway
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = poly * t * __expf(-z * z);   // 1 - erf(z)
    const float erfz = 1.0f - e;
    return 0.5f * x * (1.0f + copysignf(erfz, x));
}

// 4 consecutive elements of a [.., C] row, as floats, from an f32 or bf16 buffer
__device__ __forceinline__ f32x4 load4(const void* base, size_t idx, int dt) {
    f32x4 r;
    if (dt == DT_F32) {
        r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
    } else {
        uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
        float a, b, c, d;
        if (dt == DT_F16) {
            unpack2<DT_F16>(u.x, a, b);
            unpack2<DT_F16>(u.y, c, d);
        } else {
            unpack2<DT_BF16>(u.x, a, b);
            unpack2<DT_BF16>(u.y, c, d);
        }
        r = f32x4{a, b, c, d};
    }
    return r;
}
__device__ __forceinline__ void store4(void* base, size_t idx, int dt, f32x4 v) {
    if (dt == DT_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
    } else {
        uint2 u;
        if (dt == DT_F16) {
            u.x = pack2h(v[0], v[1]);
            u.y = pack2h(v[2], v[3]);
        } else {
            u.x = pack2bf(v[0], v[1]);
            u.y = pack2bf(v[2], v[3]);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + idx) = u;
    }
}
#endif

// One-time (per device) opt-in to more than 64 KB of dynamic LDS for a kernel.  `done` is a static bit mask owned by the
// launcher (one bit per device id: engines on different devices share the process).
static inline int ensure_dyn_smem(const void* fn, int bytes, unsigned long long* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    const unsigned long long bit = 1ull << (dev & 63);
    if (*done & bit) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
    *done |= bit;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution / linear:  C[M,N] = epilogue( A_gather[M,K] x W[N,K]^T )
// ---------------------------------------------------------------------------------------------
struct GemmParams {
    const void* A;        // activations, NHWC (conv) or [M, lda] rows (linear)
    const void* W;        // weights [Nw][Kpad] in the compute type, K-contiguous (taps x Cin)
    const float* bias;    // [Nw] fp32 or null
    void* C;              // output [M, ldc]
    const void* R;        // residual [M, ldr] or null (added after scale)
    const float* rowvec;  // per-sample broadcast row (time-embedding projection) or null
    void* VT;             // optional transposed output for columns >= vt_begin: [B][N-vt_begin][vt_ld]
    int M, N, K;          // K: logical reduction length (taps*Cin); N: weight rows (virtual cols)
    int Kpad;             // weight row stride in elements (multiple of the K tile)
    int lda, ldc, ldr;    // element strides (lda = pixel stride for conv)
    int a_dt, c_dt, r_dt; // PdDType of A / C / R
    int taps;             // 1 (linear / conv1x1) or 9 (conv3x3, pad 1)
    int Cin;              // channels per tap
    int Hin, Win, Hout, Wout, stride, ups;  // conv geometry; ups=1: nearest x2 upsample fused in the gather
    int rows_per_sample;  // Hout*Wout (conv) or tokens per sample (linear)
    int rowvec_stride;    // elements between samples in rowvec (0: one row for all)
    int act;              // 0 none, 1 SiLU, 2 GEGLU (weights pre-interleaved in 80+80 blocks), 3 quick-GELU, 4 tanh-GELU
    int a_silu;           // apply SiLU to A on load (emb_layers)
    float out_scale;      // applied to (acc + bias [+ rowvec]) before the residual
    int vt_begin, vt_ld;  // see VT
    int Nout;             // GEGLU: logical output columns (N/2 rounded), else == N
    int splitk;           // >1: grid.y slices K; each slice stores an fp32 slab, a finalize pass sums them + epilogue
    void* slab;           // [splitk][M][N] fp32 workspace
    int* tile_cnt;        // split-K: one zeroed counter per 128x160 tile -> finalize fused into the last-arriving slice; null: separate pass
    int defer_finalize;   // split-K without tile counters: 1 = no finalize pass here -- the consumer (norm.hip gn_fused_kernel, slab mode) sums the slabs
    int big_tile;         // tile: 0 128 x 160 (4 waves), 1 256 x 160, 2 128 x 160 on 8 waves (short K), 3 256 x 320, 4 256 x 192
    const float* gn_coef; // conv_patch only: [B][Cin][2] GroupNorm coefficients applied (+SiLU) while staging A; null = none
    int gn_silu;
    int ldw;              // weight row stride in elements (0: Kpad) -- lets a device activation act as the W operand
    // LayerNorm folded into this GEMM (consumer side): A is the RAW residual-stream tensor, W holds W * diag(gamma), and the
    // epilogue turns acc = x . W'^T into LN(x) . W^T = rstd[m] * (acc - mean[m] * colsum[n]) (+ bias, which carries beta . W^T).
    // ln_stats: [M][ln_parts][2] fp32 {sum, sum of squares} partials of each row of A over disjoint column ranges.
    const float* ln_stats;
    const float* ln_colsum;   // [N] sum_k W'[n][k] of the stored (rounded) weights
    int ln_parts;
    int ln_C;                 // row length of A (number of normalised channels)
    float ln_eps;
    // producer side: this GEMM writes a residual-stream tensor that a LayerNorm reads next -- every (row, column range of
    // one wave) leaves its {sum, sum of squares} in stats_out[M][stats_parts][2] (igemm_kernel epilogue only)
    float* stats_out;
    int stats_parts;
    // MMDiT (sd3.cpp): per-sample gate row multiplied into (acc + bias) * out_scale before the residual (AdaLN-Zero), and a
    // row remap of the stores so that two token streams land in one joint [sample][c_sample_rows] buffer:
    // row = sample * c_sample_rows + c_row_off + tok (c_sample_rows = 0: row = gm); V^T tokens shift by vt_tok_off.
    const float* gate;
    int gate_stride;
    int c_sample_rows, c_row_off, vt_tok_off;
    // ... and of the A rows (linear, MM instantiations): row = sample * a_sample_rows + a_row_off + tok -- one token stream
    // read out of a joint buffer (a_sample_rows = 0: row = m)
    int a_sample_rows, a_row_off;
    // PREC_FP8: acc * a_scale[m] * w_scale[n] before everything else in the epilogue
    const float* a_scale;
    const float* w_scale;
    // c_dt == DT_FP8 (MM instantiations): the output rows are written as e4m3 of value / c_scale[row]; the caller guarantees
    // |value| <= 448 c_scale[row] (sd3.cpp: a Cauchy-Schwarz bound from the input row's norm)
    const float* c_scale;
};

// element-wise / norm / attention launchers (definitions in the .hip files)
struct AttnParams {
    const void* Q; const void* K; const void* VT; void* O;
    int ldq, ldk, ldo;   // element strides between tokens
    int vt_ld;           // VT row stride (keys, padded)
    long long q_bs, k_bs, vt_bs, o_bs;  // element strides between samples
    int Nq, Nk, heads, dh;
    float scale;
    int B;
    int legacy;  // 1: use the single-buffered reference kernel (debug)
    int causal;  // keys after the query are masked (CLIP text transformer)
};

int launch_gemm(const GemmParams& p, int prec, hipStream_t s, hipEvent_t mid = nullptr);   // prec: the compute type (DT_*)
// gemm_ring.hip: persistent LDS-DMA ring GEMM for plain linear layers over 2-byte operands (tile 0: 128 x 160, 1: 256 x 160)
bool ring_gemm_eligible(const GemmParams& p, int prec);
int launch_ring_gemm(const GemmParams& p, int prec, int tile, hipStream_t s);
// fp8 (e4m3) rows with one scale per row: dst[r][k] = e4m3(src[r][k] / scale[r]), scale[r] = max_k |src[r][k]| / 448 (weights of
// the SD3 linear layers that run in PREC_FP8); src in dtype src_dt with row stride src_ld, dst row stride dst_ld >= K (pad zeroed)
int launch_quant_rows(const void* src, int src_dt, int src_ld, void* dst, int dst_ld, float* scale, int rows, int K, hipStream_t s);
// y[b][n] = bias[n] + sum_k W[n][k] * f(a[b][k]) for B <= 4 fp32 rows (f = SiLU when a_silu): the weight-streaming form of
// a Linear over a handful of rows (MMDiT modulation vectors); w_dt: DT_F32 / DT_F16 / DT_BF16, K % 8 == 0, K <= 2048
int launch_gemv(const float* a, int lda, const void* W, int w_dt, int Kpad, const float* bias, float* y, int ldy, int B, int N, int K,
                int a_silu, hipStream_t s);
int gemm_tiles(int M, int N);
int launch_splitk_finalize(const GemmParams& p, hipStream_t s);   // sums p.splitk fp32 slabs in slice order + epilogue
int conv_patch_tiles(const GemmParams& p, int prec);  // 0: shape not eligible for the LDS-patch conv kernel
int launch_conv_patch(const GemmParams& p, int prec, hipStream_t s);
int launch_conv_patch2(const GemmParams& p, int prec, hipStream_t s);   // conv_patch2.hip: compute / loader wave specialisation (2-byte types)
bool conv_patch4_eligible(const GemmParams& p, int prec);              // conv_patch4.hip: 4 waves per block, one per SIMD, 32x32x16 MFMAs, LDS-DMA operands (2-byte types)
int launch_conv_patch4(const GemmParams& p, int prec, hipStream_t s);
int launch_attention(const AttnParams& p, int prec, hipStream_t s);
// st_tail.hip: everything after the self-attention product of a 320-channel SpatialTransformer block in one kernel (attn1.to_out +
// residual, norm2, attn2 against the hoisted context K / V, norm3, GEGLU feed-forward, proj_out + the block residual)
bool st_tail_eligible(int prec, int C, int heads, int rows_per_sample, int Nk);
size_t st_tail_weight_bytes();
size_t st_tail_vec_floats();
size_t st_tail_kv_bytes(int B);
double st_tail_flops(long long M, int Nk);
// ... and its front: GroupNorm apply + proj_in + norm1 + to_q / to_k / to_v (h, q | k and V^T out)
size_t st_front_weight_bytes();
size_t st_front_vec_floats();
double st_front_flops(long long M);
int launch_st_front_pack(const void* wpi, const void* wqkv_ln, int ld, const float* bpi, const float* bqkv_ln, void* wdst, float* vdst, hipStream_t s);
int launch_st_front(const void* x, const float* coef, void* h, void* qk, void* vt, const void* wpk, const float* vec, long long M, int rows_per_sample,
                    int vt_ld, int prec, hipStream_t s);
int launch_st_tail_pack(const void* wo1, const void* wq_ln, const void* wo2, const void* w1_ln, const void* w2, const void* wp, int ld_c, int ld_w2,
                        void* dst, hipStream_t s);
int launch_st_tail_vec(const float* bo1, const float* bq_ln, const float* bo2, const float* b1_ln, const float* b2, const float* bp, float* dst,
                       hipStream_t s);
int launch_st_tail_kv_pack(const void* K, const void* VT, void* dst, int B, int Nk, int lpad, hipStream_t s);
int launch_st_tail(const void* att, const void* h, const void* x_in, void* out, const void* wpk, const float* vec, const void* kvp, long long M,
                   int rows_per_sample, int Nk, int s_dt, float scale, int prec, hipStream_t s, long long in_rows = 0);
int launch_gn_stats(const void* x, int x_dt, double* partial, int B, int HW, int C, int groups, int nchunk, hipStream_t s);
int launch_gn_apply(const void* x, int x_dt, void* y, int y_dt, const double* partial, const float* gamma,
                    const float* beta, int B, int HW, int C, int groups, int nchunk, float eps, int silu, hipStream_t s);
extern int g_gn_reg;   // norm.hip
int gn_fused_bundle(int x_dt, int HW, int C, int groups);   // norm.hip: > 0 when the single-kernel GroupNorm applies
int launch_gn_fused(const void* x, int x_dt, void* y, int y_dt, const float* gamma, const float* beta, int B, int HW, int C, int groups,
                    float eps, int do_silu, hipStream_t s);
// the same kernel fed by a split-K GEMM's fp32 slabs instead of a stored tensor: x[row][c] = round_T(sum_s slab[s][row][c] + bias[c] +
// rowvec[sample][c]) -- splitk_finalize_kernel's arithmetic and rounding, so the result is bit-identical to finalize + launch_gn_fused
int launch_gn_fused_slabs(const float* slabs, int nslab, const float* bias, const float* rowvec, int rowvec_stride, int x_dt, void* y, int y_dt,
                          const float* gamma, const float* beta, int B, int HW, int C, int groups, float eps, int do_silu, hipStream_t s);
int launch_gn_coef(const double* partial, const float* gamma, const float* beta, float* coef, int B, int HW, int C, int groups,
                   int nchunk, float eps, hipStream_t s);
// {sum, sum of squares} of every row: the single-part form of the LayerNorm statistics above (producers whose epilogue
// cannot emit them: split-K finalize, patch conv)
int launch_row_stats(const void* x, int x_dt, float* stats, int rows, int C, hipStream_t s);
// W' = W * diag(gamma) in the compute type, colsum[n] = sum_k W'[n][k], bias_out[n] = bias_in[n] + sum_k beta[k] * W[n][k]
int launch_ln_fold(const void* W, void* Wout, int dt, int N, int K, int Kpad, const float* gamma, const float* beta, const float* bias_in,
                   float* colsum, float* bias_out, hipStream_t s);
int launch_layernorm(const void* x, int x_dt, void* y, int y_dt, const float* gamma, const float* beta,
                     int rows, int C, float eps, hipStream_t s);
int launch_nchw_to_nhwc(const float* in, void* out, int out_dt, int B, int C, int H, int W, int Cpad, hipStream_t s,
                        float scale = 1.0f);
int launch_softmax_rows(const float* in, void* out, int out_dt, int rows, int n, hipStream_t s);
// out[b,l,:] = tok[ids[b,l]] + pos[l]  (CLIP text embeddings); tables are [rows][ld] in dtype dt
int launch_embed_tokens(const int* ids, const void* tok, int tok_ld, const void* pos, int pos_ld, int dt, void* out, int out_dt,
                        int B, int L, int C, int vocab, hipStream_t s);
int launch_nhwc_to_nchw(const void* in, int in_dt, float* out, int B, int C, int H, int W, int Cpad, float scale, hipStream_t s);
int launch_cast_rows(const float* in, void* out, int out_dt, long long rows, int C, int Cpad, hipStream_t s);
int launch_concat_add(const void* a, const void* a_add, const void* b, const void* b_add, void* out, int dt,
                      long long rows, int Ca, int Cb, hipStream_t s, long long b_rows = 0, long long b_add_rows = 0);
int launch_add_inplace(void* a, const void* b, int dt, long long n, hipStream_t s);
struct DdimCoef { float sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma, cfg_scale; };
int launch_cfg_ddim(const void* eps, int eps_dt, int eps_C, float* x_state, float* pred_x0, float* eps_guided,
                    void* x_in, const float* noise, int B, int HW, int C, int Cpad, int use_cfg, DdimCoef k,
                    float temperature, int do_update, hipStream_t s);
int launch_fill_random(void* p, int dt, long long n, float scale, float shift, uint64_t seed, hipStream_t s);
// sd3_kernels.hip: element-wise pieces of the MMDiT path
// y_dt == DT_FP8: y holds e4m3 bytes and y_scale[row] the row's scale (max |value| / 448); add: x <- x + add first (written back)
int launch_adaln(const void* x, int x_dt, void* y, int y_dt, const float* mod, int mod_stride, int shift_off, int scale_off, int rows,
                 int rows_per_sample, int C, float eps, hipStream_t s, float* y_scale = nullptr, const void* add = nullptr,
                 float* bound_out = nullptr, float bound_mul = 0.f, float bound_add = 0.f);   // bound_out[row] = |y_row|_2 * mul + add
// max over rows of the row L2 norm of W (dtype dt, row stride ld) and max |bias| -> out[0], out[1] (fp32, device, zeroed by the caller)
int launch_rows_norm_max(const void* W, int dt, int ld, const float* bias, int rows, int K, float* out, hipStream_t s);
// per-head RMSNorm (qk_norm = "rms_norm") of the q and k halves of a [rows][2 * heads * dh] buffer, in place: row r of sample b uses (wq, wk) when
// its token index < n_first, else (wq2, wk2) (the context stream's norm_added_q / norm_added_k); dh <= 64, dh % 8 == 0
int launch_qk_rmsnorm(void* qk, int dt, long long rows, int rows_per_sample, int n_first, int heads, int dh, const float* wq, const float* wk,
                      const float* wq2, const float* wk2, float eps, hipStream_t s);
int launch_patchify(const float* nchw, void* out, int out_dt, int B, int C, int H, int W, int patch, int Cpad, int Kpad, hipStream_t s);
// Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0) of B <= 64 host-side timesteps: out[b] = [cos(t f_i) | sin(t f_i)],
// f_i = exp(-ln(10000) i / 128); the values are passed by value (no host buffer has to outlive the launch)
int launch_timestep_embedding(const float* t_host, int B, float* out, hipStream_t s);
int launch_pos_crop(const float* table, float* out, int B, int h, int w, int max_size, int D, hipStream_t s);
int launch_unpatchify(const void* in, int in_dt, int ld, float* nchw, int B, int C, int h, int w, int patch, hipStream_t s);
int launch_cfg_euler(const float* v, float* x, int B, long long n, float guidance, float dsigma, int use_cfg, hipStream_t s);
int launch_fill_x_in(const float* x_state_nchw, float* x_in, int B, int dup, int C, int Cpad, int HW, hipStream_t s);
