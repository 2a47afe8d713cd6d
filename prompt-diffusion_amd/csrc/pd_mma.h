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

template <bool F32>
__device__ __forceinline__ void mma(const uint4& w, const uint4& a, f32x4& acc) {
    if constexpr (F32) {
        const float* wf = reinterpret_cast<const float*>(&w);
        const float* af = reinterpret_cast<const float*>(&a);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[j], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a),
                                                      acc, 0, 0, 0);
    }
}

// convert 8 fp32 (two uint4) to 8 bf16 (one uint4)
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
    uint4 r;
    r.x = pack2bf(f[0], f[1]); r.y = pack2bf(f[2], f[3]); r.z = pack2bf(f[4], f[5]); r.w = pack2bf(f[6], f[7]);
    return r;
}

// bias / time-embedding row / activation / scale / residual / (transposed) store of 4 consecutive channels
__device__ __forceinline__ void epilogue4(const GemmParams& p, int gm, int gn, int sample, int tok, f32x4 v) {
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + gn);
    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (size_t)sample * p.rowvec_stride + gn);
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
    } else if (p.act == 3) {   // quick_gelu: x * sigmoid(1.702 x)  (CLIP MLP)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.0f + __expf(-1.702f * v[j]));
    }
    v *= p.out_scale;
    if (p.R) v += load4(p.R, (size_t)gm * p.ldr + gn, p.r_dt);
    if (gn >= p.vt_begin) {
        // transposed store (attention V^T): [sample][channel][token]
        const size_t base = ((size_t)sample * (p.N - p.vt_begin) + (gn - p.vt_begin)) * p.vt_ld + tok;
        if (p.c_dt == DT_F32) {
            float* o = reinterpret_cast<float*>(p.VT);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[base + (size_t)j * p.vt_ld] = v[j];
        } else {
            uint16_t* o = reinterpret_cast<uint16_t*>(p.VT);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[base + (size_t)j * p.vt_ld] = f2bf(v[j]);
        }
    } else {
        store4(p.C, (size_t)gm * p.ldc + gn, p.c_dt, v);
    }
}

