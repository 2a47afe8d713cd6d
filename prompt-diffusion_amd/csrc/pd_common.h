// Shared declarations of the pdengine HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

enum PdDType : int { DT_F32 = 0, DT_BF16 = 1 };
static inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

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
        r[0] = __uint_as_float(u.x << 16);
        r[1] = __uint_as_float(u.x & 0xffff0000u);
        r[2] = __uint_as_float(u.y << 16);
        r[3] = __uint_as_float(u.y & 0xffff0000u);
    }
    return r;
}
__device__ __forceinline__ void store4(void* base, size_t idx, int dt, f32x4 v) {
    if (dt == DT_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
    } else {
        uint2 u;
        u.x = pack2bf(v[0], v[1]);
        u.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + idx) = u;
    }
}
#endif

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
    int act;              // 0 none, 1 SiLU, 2 GEGLU (weights pre-interleaved in 80+80 blocks)
    int a_silu;           // apply SiLU to A on load (emb_layers)
    float out_scale;      // applied to (acc + bias [+ rowvec]) before the residual
    int vt_begin, vt_ld;  // see VT
    int Nout;             // GEGLU: logical output columns (N/2 rounded), else == N
    int splitk;           // >1: grid.y slices K; each slice stores an fp32 slab, a finalize pass sums them + epilogue
    void* slab;           // [splitk][M][N] fp32 workspace
    int* tile_cnt;        // split-K: one zeroed counter per 128x160 tile -> finalize fused into the last-arriving slice; null: separate pass
    int big_tile;         // 1: 256 x 160 block tile (8 waves) instead of 128 x 160
    const float* gn_coef; // conv_patch only: [B][Cin][2] GroupNorm coefficients applied (+SiLU) while staging A; null = none
    int gn_silu;
    int ldw;              // weight row stride in elements (0: Kpad) -- lets a device activation act as the W operand
    int diag;             // timing diagnostic: every tile row reads row 0 (operands served from L1); results are wrong
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

int launch_gemm(const GemmParams& p, bool f32mode, hipStream_t s, hipEvent_t mid = nullptr);
int gemm_tiles(int M, int N);
int launch_splitk_finalize(const GemmParams& p, hipStream_t s);   // sums p.splitk fp32 slabs in slice order + epilogue
int conv_patch_tiles(const GemmParams& p, bool f32mode);  // 0: shape not eligible for the LDS-patch conv kernel
int launch_conv_patch(const GemmParams& p, bool f32mode, hipStream_t s);
bool gemm8_eligible(const GemmParams& p);          // gemm8.hip: 256 x 256 LDS-DMA tile, bf16 linear layers
int launch_gemm8(const GemmParams& p, hipStream_t s);
int launch_attention(const AttnParams& p, bool f32mode, hipStream_t s);
int launch_gn_stats(const void* x, int x_dt, double* partial, int B, int HW, int C, int groups, int nchunk, hipStream_t s);
int launch_gn_apply(const void* x, int x_dt, void* y, int y_dt, const double* partial, const float* gamma,
                    const float* beta, int B, int HW, int C, int groups, int nchunk, float eps, int silu, hipStream_t s);
int gn_fused_bundle(int x_dt, int HW, int C, int groups);   // norm.hip: > 0 when the single-kernel GroupNorm applies
int launch_gn_fused(const void* x, int x_dt, void* y, int y_dt, const float* gamma, const float* beta, int B, int HW, int C, int groups,
                    float eps, int do_silu, hipStream_t s);
int launch_gn_coef(const double* partial, const float* gamma, const float* beta, float* coef, int B, int HW, int C, int groups,
                   int nchunk, float eps, hipStream_t s);
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
                      long long rows, int Ca, int Cb, hipStream_t s);
int launch_add_inplace(void* a, const void* b, int dt, long long n, hipStream_t s);
struct DdimCoef { float sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma, cfg_scale; };
int launch_cfg_ddim(const void* eps, int eps_dt, int eps_C, float* x_state, float* pred_x0, float* eps_guided,
                    void* x_in, const float* noise, int B, int HW, int C, int Cpad, int use_cfg, DdimCoef k,
                    float temperature, int do_update, hipStream_t s);
int launch_fill_random(void* p, int dt, long long n, float scale, float shift, uint64_t seed, hipStream_t s);
int launch_fill_x_in(const float* x_state_nchw, float* x_in, int B, int dup, int C, int Cpad, int HW, hipStream_t s);
