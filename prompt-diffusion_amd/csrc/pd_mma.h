// Device helpers shared by the contraction kernels (gemm.hip, conv_patch.hip).
#pragma once
#include <type_traits>
#include <utility>

#include "pd_common.h"

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// MFMA operand fragments.  For the plain modes a fragment is the 16 bytes read from LDS.  PREC_F16X2 converts the
// 4 floats of a fragment once per K step: A -> [hi0..3 | lo0..3] (8 halfs), W -> {[hi|hi], [lo|lo]}; two
// v_mfma_f32_16x16x32_f16 then sum wh*ah + wh*al + wl*ah + wl*al.  hi is the truncated value (v_cvt_pkrtz), so
// x - hi is exact in fp32 and lo carries the next 11 bits.
struct FragW2 { uint4 hi, lo; };
template <int P> struct Frag { using A = uint4; using W = uint4; };
template <> struct Frag<PREC_F16X2> { using A = uint4; using W = FragW2; };

__device__ __forceinline__ void split_f16(const uint4& v, uint32_t& h01, uint32_t& h23, uint32_t& l01, uint32_t& l23) {
    const float* f = reinterpret_cast<const float*>(&v);
    const auto a = __builtin_amdgcn_cvt_pkrtz(f[0], f[1]);
    const auto b = __builtin_amdgcn_cvt_pkrtz(f[2], f[3]);
    h01 = __builtin_bit_cast(uint32_t, a);
    h23 = __builtin_bit_cast(uint32_t, b);
    l01 = pack2h(f[0] - (float)a[0], f[1] - (float)a[1]);
    l23 = pack2h(f[2] - (float)b[0], f[3] - (float)b[1]);
}
template <int P> __device__ __forceinline__ typename Frag<P>::A prep_a(const uint4& v) {
    if constexpr (P == PREC_F16X2) {
        uint4 r;
        split_f16(v, r.x, r.y, r.z, r.w);
        return r;
    } else {
        return v;
    }
}
template <int P> __device__ __forceinline__ typename Frag<P>::W prep_w(const uint4& v) {
    if constexpr (P == PREC_F16X2) {
        uint32_t h01, h23, l01, l23;
        split_f16(v, h01, h23, l01, l23);
        FragW2 r;
        r.hi = make_uint4(h01, h23, h01, h23);
        r.lo = make_uint4(l01, l23, l01, l23);
        return r;
    } else {
        return v;
    }
}

// one 16x16 MFMA step over a 16-byte K fragment of each operand; P = DT_F32 / DT_BF16 / DT_F16 / PREC_F16X2
template <int P>
__device__ __forceinline__ void mma(const typename Frag<P>::W& w, const typename Frag<P>::A& a, f32x4& acc) {
    if constexpr (P == DT_F32) {
        const float* wf = reinterpret_cast<const float*>(&w);
        const float* af = reinterpret_cast<const float*>(&a);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[j], acc, 0, 0, 0);
    } else if constexpr (P == PREC_F16X2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w.hi), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w.lo), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
    } else if constexpr (P == DT_F16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a),
                                                      acc, 0, 0, 0);
    }
}
// PREC_F16X2 over a whole K step (two 16-byte fp32 fragments per operand = 8 K elements): hi / lo planes of 8 halfs each and
// THREE MFMAs per accumulator -- wh*ah + wh*al + wl*ah; the dropped lo*lo term is 2^-22 of the product -- instead of the four
// that two independent half-steps issue.
struct FragX2 { uint4 hi, lo; };
__device__ __forceinline__ FragX2 prep_x2(const uint4& f0, const uint4& f1) {
    FragX2 r;
    split_f16(f0, r.hi.x, r.hi.y, r.lo.x, r.lo.y);
    split_f16(f1, r.hi.z, r.hi.w, r.lo.z, r.lo.w);
    return r;
}
__device__ __forceinline__ void mma_x2(const FragX2& w, const FragX2& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w.hi), __builtin_bit_cast(f16x8, a.hi), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w.hi), __builtin_bit_cast(f16x8, a.lo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w.lo), __builtin_bit_cast(f16x8, a.hi), acc, 0, 0, 0);
}

// PREC_FP8 over a whole 128-byte K step: lane (fr, fq) holds 32 e4m3 values of its row -- LDS chunks fq and 4 + fq, the same
// k permutation on both operands -- and one block-scaled MFMA (all block scales 2^0) sums the 128 products per accumulator
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ i32x8 prep_f8(const uint4& c0, const uint4& c1) {
    i32x8 r;
    r[0] = c0.x; r[1] = c0.y; r[2] = c0.z; r[3] = c0.w; r[4] = c1.x; r[5] = c1.y; r[6] = c1.z; r[7] = c1.w;
    return r;
}
__device__ __forceinline__ void mma_f8(const i32x8& w, const i32x8& a, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, acc, 0 /*A: e4m3*/, 0 /*B: e4m3*/, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// the same from two raw 16-byte fragments (converts per call in PREC_F16X2: attention only)
template <int P>
__device__ __forceinline__ void mma_raw(const uint4& w, const uint4& a, f32x4& acc) {
    mma<P>(prep_w<P>(w), prep_a<P>(a), acc);
}

// convert 8 fp32 (two uint4) to 8 bf16 / fp16 (one uint4)
template <int P>
__device__ __forceinline__ uint4 cvt8(const uint4& lo, const uint4& hi, bool do_silu) {
    float f[8];
    const float* a = reinterpret_cast<const float*>(&lo);
    const float* b = reinterpret_cast<const float*>(&hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
    if (do_silu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
    }
    return pack8<P>(f);
}

// LayerNorm statistics of row gm of a folded GEMM's A operand: {mean, rstd} from the producer's column-range partials
__device__ __forceinline__ void ln_row_stats(const GemmParams& p, int gm, float& mean, float& rstd) {
    // fp32 is enough here: <= 16 partials of <= 1280 channels each, and var = E[x^2] - mean^2 loses precision only when
    // |mean| >> std (relative error ~ 1e-7 * mean^2 / var); the residual stream this normalises is far from that regime
    float s = 0.f, q = 0.f;
    const float* st = p.ln_stats + (size_t)gm * p.ln_parts * 2;
    for (int i = 0; i < p.ln_parts; ++i) {
        const float2 t = *reinterpret_cast<const float2*>(st + 2 * i);
        s += t.x;
        q += t.y;
    }
    const float inv = __builtin_amdgcn_rcpf((float)p.ln_C);
    mean = s * inv;
    const float var = fmaxf(q * inv - mean * mean, 0.f);
    rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
}
__device__ __forceinline__ f32x4 ln_apply4(const GemmParams& p, int gn, f32x4 v, float mean, float rstd) {
    const f32x4 cs = *reinterpret_cast<const f32x4*>(p.ln_colsum + gn);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rstd * (v[j] - mean * cs[j]);
    return v;
}

// (folded LayerNorm) / bias / time-embedding row / activation / scale / residual / (transposed) store of 4 consecutive
// channels, in two halves: epilogue4_value does every read and returns the value to store (before rounding to the output
// type), epilogue4_store writes it.  A kernel runs the value half over ALL of a lane's accumulators before the first store:
// loads and stores retire through one in-order counter (vmcnt), so a bias / residual read issued behind a store cannot be
// consumed before that store is acknowledged -- interleaved, every 4-channel group pays a store round trip (measured on the
// ring GEMM: 23 k of a block's 53 k cycles, 13 k once the reads went first).
// MM: the MMDiT extras (tanh-GELU, per-sample gate, joint-buffer row remap) -- compile-time, because even as untaken branches
// they push the 256 x 320 tile and the wave-specialised patch conv (both at the VGPR cap) into scratch.
template <bool MM = false>
__device__ __forceinline__ f32x4 epilogue4_value(const GemmParams& p, int gm, int gn, int sample, f32x4 v, float ln_mean = 0.f, float ln_rstd = 0.f) {
    if constexpr (MM) {
        if (p.a_scale) {   // PREC_FP8: per-token x per-channel operand scales
            const f32x4 ws = *reinterpret_cast<const f32x4*>(p.w_scale + gn);
            const float as = p.a_scale[gm];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= as * ws[j];
        }
    }
    if (p.ln_stats) v = ln_apply4(p, gn, v, ln_mean, ln_rstd);
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + gn);
    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (size_t)sample * p.rowvec_stride + gn);
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
    } else if (p.act == 3) {   // quick_gelu: x * sigmoid(1.702 x)  (CLIP MLP)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.0f + __expf(-1.702f * v[j]));
    }
    if constexpr (MM) {
        if (p.act == 4) {   // tanh-GELU (MMDiT feed-forward): 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u))
            // 96 values per lane on the 256 x 192 tile: with an IEEE division this epilogue is ~10 % of the K = 1536 launch;
            // outputs rounded to 2 bytes take v_rcp_f32 (1 ulp) instead
            const bool fast = p.c_dt != DT_F32;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = 0.7978845608028654f * fmaf(0.044715f * v[j] * v[j], v[j], v[j]);
                const float d = 1.0f + __expf(-2.0f * u);
                v[j] = fast ? v[j] * __builtin_amdgcn_rcpf(d) : v[j] / d;
            }
        }
    }
    v *= p.out_scale;
    if constexpr (MM) {
        if (p.gate) v *= *reinterpret_cast<const f32x4*>(p.gate + (size_t)sample * p.gate_stride + gn);
    }
    if (p.R) v += load4(p.R, (size_t)gm * p.ldr + gn, p.r_dt);
    return v;
}
template <bool MM = false>
__device__ __forceinline__ void epilogue4_store(const GemmParams& p, int gm, int gn, int sample, int tok, const f32x4& v) {
    if (gn >= p.vt_begin) {
        // transposed store (attention V^T): [sample][channel][token]
        size_t base = ((size_t)sample * (p.N - p.vt_begin) + (gn - p.vt_begin)) * p.vt_ld + tok;
        if constexpr (MM) base += p.vt_tok_off;
        if (p.c_dt == DT_F32) {
            float* o = reinterpret_cast<float*>(p.VT);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[base + (size_t)j * p.vt_ld] = v[j];
        } else {
            uint16_t* o = reinterpret_cast<uint16_t*>(p.VT);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[base + (size_t)j * p.vt_ld] = cvt16_rt(v[j], p.c_dt);
        }
    } else {
        size_t crow = (size_t)gm;
        if constexpr (MM) {
            if (p.c_sample_rows) crow = (size_t)sample * p.c_sample_rows + p.c_row_off + tok;
            if (p.c_dt == DT_FP8) {   // e4m3 of value / c_scale[row] (the bound makes the clamp a no-op up to rounding)
                const float inv = __builtin_amdgcn_rcpf(p.c_scale[crow]);
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0] * inv, -448.f), 448.f), fminf(fmaxf(v[1] * inv, -448.f), 448.f), 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2] * inv, -448.f), 448.f), fminf(fmaxf(v[3] * inv, -448.f), 448.f), w, true);
                *reinterpret_cast<int*>(reinterpret_cast<char*>(p.C) + crow * p.ldc + gn) = w;
                return;
            }
        }
        store4(p.C, crow * p.ldc + gn, p.c_dt, v);
    }
}
// both halves for one group (the split-K finalize pass: one group per thread)
template <bool MM = false>
__device__ __forceinline__ f32x4 epilogue4(const GemmParams& p, int gm, int gn, int sample, int tok, f32x4 v, float ln_mean = 0.f,
                                           float ln_rstd = 0.f) {
    v = epilogue4_value<MM>(p, gm, gn, sample, v, ln_mean, ln_rstd);
    epilogue4_store<MM>(p, gm, gn, sample, tok, v);
    return v;
}
