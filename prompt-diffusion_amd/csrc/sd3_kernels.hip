// pdengine: element-wise kernels of the SD3 / MMDiT path (SURVEY.md §8f "next" row N4; host side in sd3.cpp).
// All HBM-bound single passes; the contractions run on the igemm / attention kernels of the UNet path.
#include "pd_common.h"

// v_cvt_pk_fp8_f32 turns |x| > 448 into NaN: saturate first (the scales map the row maximum onto 448, so this only absorbs rounding)
__device__ __forceinline__ float clamp448(float x) { return fminf(fmaxf(x, -448.0f), 448.0f); }

// AdaLayerNormZero / AdaLayerNormContinuous body: y = LN(x) (no affine, eps) * (1 + scale[b]) + shift[b].
// One wave per row, the row held in registers between the two passes; C <= 2048, C % 4 == 0.
// mod: fp32 [B][mod_stride] (output of the modulation GEMM), the chunk of this norm at shift_off / scale_off.
// add (optional, dtype of x): x <- x + add first, written back -- the ControlNet residual that lands on the residual stream
// between two blocks, taken in by the next block's first norm instead of a pass of its own.
template <int XD, int YD>
__global__ __launch_bounds__(256) void adaln_kernel(const void* __restrict__ x, void* __restrict__ y, const float* __restrict__ mod,
                                                     int mod_stride, int shift_off, int scale_off, int rows, int rows_per_sample,
                                                     int C, float eps, float* __restrict__ y_scale, void* __restrict__ x_rw,
                                                     const void* __restrict__ add, float* __restrict__ bound_out, float bound_mul,
                                                     float bound_add) {
    constexpr int MAXV = 8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        v[k] = vi < nv ? load4(x, (size_t)row * C + (size_t)vi * 4, XD) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (add && vi < nv) {
            v[k] += load4(add, (size_t)row * C + (size_t)vi * 4, XD);
            store4(x_rw, (size_t)row * C + (size_t)vi * 4, XD, v[k]);
            if constexpr (XD != DT_F32) {   // normalise what the stream now holds (the rounded sum), like a separate add pass would
                uint2 u;
                u.x = pack2<XD>(v[k][0], v[k][1]);
                u.y = pack2<XD>(v[k][2], v[k][3]);
                float a0, a1, a2, a3;
                unpack2<XD>(u.x, a0, a1);
                unpack2<XD>(u.y, a2, a3);
                v[k] = f32x4{a0, a1, a2, a3};
            }
        }
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (lane + 64 * k < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[k][j] - mean; q = fmaf(d, d, q); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    const float* mrow = mod + (size_t)(row / rows_per_sample) * mod_stride;
    float amax = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nv) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(mrow + scale_off + vi * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(mrow + shift_off + vi * 4);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf((v[k][j] - mean) * rstd, 1.0f + sc[j], sh[j]);
            if constexpr (YD == DT_FP8) {
                v[k] = o;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                sq += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            } else {
                store4(y, (size_t)row * C + (size_t)vi * 4, YD, o);
            }
        }
    }
    if constexpr (YD == DT_FP8) {   // e4m3 bytes with the row's scale: value = byte * scale, |byte| <= 448
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
        const float inv = 1.0f / scale;
        if (lane == 0) y_scale[row] = scale;
        if (bound_out) {   // |W y|_inf <= |y|_2 max_n |W_n|_2: the row scale of the consuming layer's e4m3 OUTPUT
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            if (lane == 0) bound_out[row] = fmaf(sqrtf(sq), bound_mul, bound_add);
        }
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int vi = lane + 64 * k;
            if (vi < nv) {
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[k][0] * inv), clamp448(v[k][1] * inv), 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[k][2] * inv), clamp448(v[k][3] * inv), w, true);
                reinterpret_cast<int*>(y)[(size_t)row * (C / 4) + vi] = w;
            }
        }
    }
}

int launch_adaln(const void* x, int x_dt, void* y, int y_dt, const float* mod, int mod_stride, int shift_off, int scale_off, int rows,
                 int rows_per_sample, int C, float eps, hipStream_t s, float* y_scale, const void* add, float* bound_out,
                 float bound_mul, float bound_add) {
    void* x_rw = const_cast<void*>(x);   // written only when add is given
    if (C % 4 || C > 2048 || rows < 1 || rows_per_sample < 1 || (mod_stride | shift_off | scale_off) % 4) return 1;
    if (y_dt == DT_FP8 && !y_scale) return 1;
    const dim3 grid((rows + 3) / 4);
#define PD_ADALN(XD, YD)                                                                                                       \
    hipLaunchKernelGGL((adaln_kernel<XD, YD>), grid, dim3(256), 0, s, x, y, mod, mod_stride, shift_off, scale_off, rows, \
                       rows_per_sample, C, eps, y_scale, x_rw, add, bound_out, bound_mul, bound_add)
    if (x_dt == DT_F32 && y_dt == DT_F32) PD_ADALN(DT_F32, DT_F32);
    else if (x_dt == DT_F32 && y_dt == DT_F16) PD_ADALN(DT_F32, DT_F16);
    else if (x_dt == DT_F32 && y_dt == DT_BF16) PD_ADALN(DT_F32, DT_BF16);
    else if (x_dt == DT_F16 && y_dt == DT_F16) PD_ADALN(DT_F16, DT_F16);
    else if (x_dt == DT_BF16 && y_dt == DT_BF16) PD_ADALN(DT_BF16, DT_BF16);
    else if (x_dt == DT_F32 && y_dt == DT_FP8) PD_ADALN(DT_F32, DT_FP8);
    else if (x_dt == DT_F16 && y_dt == DT_FP8) PD_ADALN(DT_F16, DT_FP8);
    else if (x_dt == DT_BF16 && y_dt == DT_FP8) PD_ADALN(DT_BF16, DT_FP8);
    else return 1;
#undef PD_ADALN
    return hipGetLastError() != hipSuccess;
}

// PatchEmbed's Conv2d(C, D, kernel = stride = patch) as a GEMM: rows [B * (H/p) * (W/p)][Kpad], k = (py * p + px) * Cpad + c
// (the engine's tap-major conv weight layout, upload_rows), pad columns zero.
template <int YD>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ in, void* __restrict__ out, int B, int C, int H, int W,
                                                        int patch, int Cpad, int Kpad) {
    const long long total = (long long)B * (H / patch) * (W / patch) * Kpad;
    const int wq = W / patch, hq = H / patch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % Kpad);
        const long long r = i / Kpad;
        const int tap = k / Cpad, c = k - tap * Cpad;
        float val = 0.f;
        if (tap < patch * patch && c < C) {
            const int px = tap % patch, py = tap / patch;
            const int tx = (int)(r % wq), ty = (int)((r / wq) % hq), b = (int)(r / ((long long)wq * hq));
            val = in[(((size_t)b * C + c) * H + (size_t)ty * patch + py) * W + (size_t)tx * patch + px];
        }
        if constexpr (YD == DT_F32) reinterpret_cast<float*>(out)[i] = val;
        else reinterpret_cast<uint16_t*>(out)[i] = cvt16<YD>(val);
    }
}

int launch_patchify(const float* nchw, void* out, int out_dt, int B, int C, int H, int W, int patch, int Cpad, int Kpad, hipStream_t s) {
    if (patch < 1 || H % patch || W % patch || Cpad < C || Kpad < Cpad * patch * patch) return 1;
    const long long total = (long long)B * (H / patch) * (W / patch) * Kpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (out_dt == DT_F32) hipLaunchKernelGGL((patchify_kernel<DT_F32>), dim3(blocks), dim3(256), 0, s, nchw, out, B, C, H, W, patch, Cpad, Kpad);
    else if (out_dt == DT_F16) hipLaunchKernelGGL((patchify_kernel<DT_F16>), dim3(blocks), dim3(256), 0, s, nchw, out, B, C, H, W, patch, Cpad, Kpad);
    else hipLaunchKernelGGL((patchify_kernel<DT_BF16>), dim3(blocks), dim3(256), 0, s, nchw, out, B, C, H, W, patch, Cpad, Kpad);
    return hipGetLastError() != hipSuccess;
}

// PatchEmbed.cropped_pos_embed, broadcast over the batch: out[b][y * w + x][:] = table[(top + y) * max + left + x][:]
__global__ __launch_bounds__(256) void pos_crop_kernel(const float* __restrict__ table, float* __restrict__ out, int B, int h, int w,
                                                        int max_size, int D) {
    const int top = (max_size - h) / 2, left = (max_size - w) / 2;
    const int dv = D / 4;
    const long long total = (long long)B * h * w * dv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int d = (int)(i % dv);
        const long long r = i / dv;
        const int x = (int)(r % w), y = (int)((r / w) % h);
        const size_t src = ((size_t)(top + y) * max_size + left + x) * D + (size_t)d * 4;
        *reinterpret_cast<f32x4*>(out + (size_t)r * D + (size_t)d * 4) = *reinterpret_cast<const f32x4*>(table + src);
    }
}

int launch_pos_crop(const float* table, float* out, int B, int h, int w, int max_size, int D, hipStream_t s) {
    if (h > max_size || w > max_size || D % 4 || B < 1) return 1;
    const long long total = (long long)B * h * w * (D / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pos_crop_kernel, dim3(blocks), dim3(256), 0, s, table, out, B, h, w, max_size, D);
    return hipGetLastError() != hipSuccess;
}

// proj_out rows [B * h * w][ld], column (py * p + px) * C + c  ->  fp32 NCHW [B, C, h p, w p]  ("nhwpqc->nchpwq")
__global__ __launch_bounds__(256) void unpatchify_kernel(const void* __restrict__ in, int in_dt, int ld, float* __restrict__ out, int B,
                                                          int C, int h, int w, int patch) {
    const int H = h * patch, W = w * patch;
    const long long total = (long long)B * C * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((long long)W * H)) % C), b = (int)(i / ((long long)W * H * C));
        const size_t row = ((size_t)b * h + y / patch) * w + x / patch;
        const size_t col = (size_t)((y % patch) * patch + (x % patch)) * C + c;
        out[i] = in_dt == DT_F32 ? reinterpret_cast<const float*>(in)[row * ld + col]
                                 : cvt32_rt(reinterpret_cast<const uint16_t*>(in)[row * ld + col], in_dt);
    }
}

int launch_unpatchify(const void* in, int in_dt, int ld, float* nchw, int B, int C, int h, int w, int patch, hipStream_t s) {
    const long long total = (long long)B * C * h * w * patch * patch;
    if (total < 1) return 1;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(unpatchify_kernel, dim3(blocks), dim3(256), 0, s, in, in_dt, ld, nchw, B, C, h, w, patch);
    return hipGetLastError() != hipSuccess;
}

// classifier-free guidance + flow-matching Euler step: v holds [negative ; positive] halves of n elements each when use_cfg,
// x <- x + dsigma * (v_neg + g * (v_pos - v_neg))
__global__ __launch_bounds__(256) void cfg_euler_kernel(const float* __restrict__ v, float* __restrict__ x, long long n, float g, float ds,
                                                         int use_cfg) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float vv = v[i];
        if (use_cfg) vv = vv + g * (v[n + i] - vv);
        x[i] = x[i] + ds * vv;
    }
}

int launch_cfg_euler(const float* v, float* x, int B, long long n, float guidance, float dsigma, int use_cfg, hipStream_t s) {
    (void)B;
    if (n < 1) return 1;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(blocks), dim3(256), 0, s, v, x, n, guidance, dsigma, use_cfg);
    return hipGetLastError() != hipSuccess;
}

// Linear over B <= 4 fp32 rows: y[b][n] = bias[n] + sum_k W[n][k] * f(a[b][k]).  HBM-bound on W (the MMDiT modulation matrix
// is 1.36 GB in fp16 and is read once per evaluation): one wave per output row at a time, the activation slices a lane
// multiplies with held in registers for the whole kernel, UNROLL rows of 16-byte weight loads in flight per wave.
template <int WD, int NB, int KC>   // KC: 512-element chunks of K (K <= 512 * KC)
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ a, int lda, const void* __restrict__ W, int Kpad,
                                                         const float* __restrict__ bias, float* __restrict__ y, int ldy, int N, int K,
                                                         int a_silu) {
    constexpr int EB = WD == DT_F32 ? 4 : 2;
    constexpr int UNROLL = 4;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    // activation slices of this lane: vector loads from a clamped address (lanes past K hold zeros, which also makes the
    // weight values they fetch below irrelevant)
    float av[NB][KC][8];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = c * 512 + lane * 8;
            const bool ok = k < K;
            const float* ap = a + (size_t)b * lda + (ok ? k : 0);
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ap), hi = *reinterpret_cast<const f32x4*>(ap + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = j < 4 ? lo[j] : hi[j - 4];
                if (a_silu) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                av[b][c][j] = ok ? v : 0.f;
            }
        }
    const char* Wb = reinterpret_cast<const char*>(W);
    int koff[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) koff[c] = (c * 512 + lane * 8 < K ? c * 512 + lane * 8 : 0) * EB;
    for (int n0 = wave * UNROLL; n0 < N; n0 += nwaves * UNROLL) {
        uint4 wq[UNROLL][KC][EB / 2];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int n = n0 + u < N ? n0 + u : N - 1;   // clamped rows are computed and dropped
            const char* wr = Wb + (size_t)n * Kpad * EB;
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int h = 0; h < EB / 2; ++h) wq[u][c][h] = *reinterpret_cast<const uint4*>(wr + koff[c] + 16 * h);
        }
        float acc[UNROLL][NB];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[u][b] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                float w[8];
                if constexpr (WD == DT_F32) {
                    const float* f0 = reinterpret_cast<const float*>(&wq[u][c][0]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = f0[j];
                } else {
                    unpack8<WD>(wq[u][c][0], w);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[u][b] = fmaf(w[j], av[b][c][j], acc[u][b]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float s = acc[u][b];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                if (lane == 0 && n0 + u < N) y[(size_t)b * ldy + n0 + u] = s + (bias ? bias[n0 + u] : 0.f);
            }
    }
}

template <int WD, int NB>
static int gemv_dispatch(const float* a, int lda, const void* W, int Kpad, const float* bias, float* y, int ldy, int N, int K, int a_silu,
                         hipStream_t s) {
    const int kc = (K + 511) / 512;
    int blocks = (N / 4 + 3) / 4;           // 4 rows per wave pass, 4 waves per block
    if (blocks > 1024) blocks = 1024;       // 4 blocks per CU: every wave amortises its activation prologue over many rows
    if (blocks < 1) blocks = 1;
#define PD_GEMV(KC) hipLaunchKernelGGL((gemv_rows_kernel<WD, NB, KC>), dim3(blocks), dim3(256), 0, s, a, lda, W, Kpad, bias, y, ldy, N, K, a_silu)
    switch (kc) {
        case 1: PD_GEMV(1); break;
        case 2: PD_GEMV(2); break;
        case 3: PD_GEMV(3); break;
        case 4: PD_GEMV(4); break;
        default: return 1;
    }
#undef PD_GEMV
    return hipGetLastError() != hipSuccess;
}

int launch_gemv(const float* a, int lda, const void* W, int w_dt, int Kpad, const float* bias, float* y, int ldy, int B, int N, int K,
                int a_silu, hipStream_t s) {
    if (B < 1 || B > 4 || K % 8 || K > 2048 || N < 1) return 1;
#define PD_GEMV_B(WD)                                                                                              \
    switch (B) {                                                                                                   \
        case 1: return gemv_dispatch<WD, 1>(a, lda, W, Kpad, bias, y, ldy, N, K, a_silu, s);                       \
        case 2: return gemv_dispatch<WD, 2>(a, lda, W, Kpad, bias, y, ldy, N, K, a_silu, s);                       \
        case 3: return gemv_dispatch<WD, 3>(a, lda, W, Kpad, bias, y, ldy, N, K, a_silu, s);                       \
        default: return gemv_dispatch<WD, 4>(a, lda, W, Kpad, bias, y, ldy, N, K, a_silu, s);                      \
    }
    if (w_dt == DT_F32) { PD_GEMV_B(DT_F32) }
    if (w_dt == DT_F16) { PD_GEMV_B(DT_F16) }
    PD_GEMV_B(DT_BF16)
#undef PD_GEMV_B
}

// Weight rows -> e4m3 with one scale per row (one wave per row, two passes over the row: it is read from L2 the second time)
template <int XD>
__global__ __launch_bounds__(256) void quant_rows_kernel(const void* __restrict__ src, int src_ld, uint8_t* __restrict__ dst, int dst_ld,
                                                          float* __restrict__ scale, int rows, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float amax = 0.f;
    for (int v = lane; v < K / 4; v += 64) {
        const f32x4 t = load4(src, (size_t)row * src_ld + (size_t)v * 4, XD);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    for (int v = lane; v < dst_ld / 4; v += 64) {
        int w = 0;
        if (v < K / 4) {
            const f32x4 t = load4(src, (size_t)row * src_ld + (size_t)v * 4, XD);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(t[0] * inv), clamp448(t[1] * inv), 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(t[2] * inv), clamp448(t[3] * inv), w, true);
        }
        reinterpret_cast<int*>(dst)[(size_t)row * (dst_ld / 4) + v] = w;
    }
}

int launch_quant_rows(const void* src, int src_dt, int src_ld, void* dst, int dst_ld, float* scale, int rows, int K, hipStream_t s) {
    if (rows < 1 || K % 4 || dst_ld % 4 || dst_ld < K || src_ld % 4) return 1;
    const dim3 grid((rows + 3) / 4);
    uint8_t* d = reinterpret_cast<uint8_t*>(dst);
    if (src_dt == DT_F32) hipLaunchKernelGGL((quant_rows_kernel<DT_F32>), grid, dim3(256), 0, s, src, src_ld, d, dst_ld, scale, rows, K);
    else if (src_dt == DT_F16) hipLaunchKernelGGL((quant_rows_kernel<DT_F16>), grid, dim3(256), 0, s, src, src_ld, d, dst_ld, scale, rows, K);
    else if (src_dt == DT_BF16) hipLaunchKernelGGL((quant_rows_kernel<DT_BF16>), grid, dim3(256), 0, s, src, src_ld, d, dst_ld, scale, rows, K);
    else return 1;
    return hipGetLastError() != hipSuccess;
}

// max_n |W_n|_2 and max |bias| (one wave per row; non-negative floats compare like their bit patterns)
template <int XD>
__global__ __launch_bounds__(256) void rows_norm_max_kernel(const void* __restrict__ W, int ld, const float* __restrict__ bias, int rows, int K,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float sq = 0.f;
    for (int v = lane; v < K / 4; v += 64) {
        const f32x4 t = load4(W, (size_t)row * ld + (size_t)v * 4, XD);
        sq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) {
        atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(sqrtf(sq)));
        if (bias) atomicMax(reinterpret_cast<unsigned int*>(out) + 1, __float_as_uint(fabsf(bias[row])));
    }
}

int launch_rows_norm_max(const void* W, int dt, int ld, const float* bias, int rows, int K, float* out, hipStream_t s) {
    if (rows < 1 || K % 4 || ld % 4) return 1;
    const dim3 grid((rows + 3) / 4);
    if (dt == DT_F32) hipLaunchKernelGGL((rows_norm_max_kernel<DT_F32>), grid, dim3(256), 0, s, W, ld, bias, rows, K, out);
    else if (dt == DT_F16) hipLaunchKernelGGL((rows_norm_max_kernel<DT_F16>), grid, dim3(256), 0, s, W, ld, bias, rows, K, out);
    else if (dt == DT_BF16) hipLaunchKernelGGL((rows_norm_max_kernel<DT_BF16>), grid, dim3(256), 0, s, W, ld, bias, rows, K, out);
    else return 1;
    return hipGetLastError() != hipSuccess;
}

struct TimestepArgs { float t[64]; };
__global__ __launch_bounds__(128) void timestep_embedding_kernel(TimestepArgs a, float* __restrict__ out) {
    const int b = blockIdx.x, i = threadIdx.x;
    const float f = expf(-logf(10000.0f) * (float)i / 128.0f);
    const float x = a.t[b] * f;
    out[(size_t)b * 256 + i] = cosf(x);
    out[(size_t)b * 256 + 128 + i] = sinf(x);
}

int launch_timestep_embedding(const float* t_host, int B, float* out, hipStream_t s) {
    if (B < 1 || B > 64) return 1;
    TimestepArgs a{};
    for (int b = 0; b < B; ++b) a.t[b] = t_host[b];
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(128), 0, s, a, out);
    return hipGetLastError() != hipSuccess;
}
// qk_norm = "rms_norm": one thread per (row, head, q|k) group of dh <= 64 values of the joint q|k buffer
template <int DT>
__global__ __launch_bounds__(256) void qk_rmsnorm_kernel(void* __restrict__ qk, long long rows, int rows_per_sample, int n_first, int heads, int dh,
                                                         const float* __restrict__ wq, const float* __restrict__ wk,
                                                         const float* __restrict__ wq2, const float* __restrict__ wk2, float eps) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = rows * heads * 2;
    if (gid >= total) return;
    const int which = (int)(gid % 2), head = (int)((gid / 2) % heads);
    const long long row = gid / (2LL * heads);
    const int tok = (int)(row % rows_per_sample);
    const float* w = tok < n_first ? (which ? wk : wq) : (which ? wk2 : wq2);
    uint16_t* p = reinterpret_cast<uint16_t*>(qk) + (size_t)row * (2 * heads * dh) + (size_t)which * heads * dh + (size_t)head * dh;
    float v[64];
    float ss = 0.f;
    for (int i = 0; i < dh; i += 8) {
        float f[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(p + i), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[i + j] = f[j]; ss = fmaf(f[j], f[j], ss); }
    }
    const float r = 1.0f / sqrtf(ss / (float)dh + eps);
    for (int i = 0; i < dh; i += 8) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = v[i + j] * r * w[i + j];
        *reinterpret_cast<uint4*>(p + i) = pack8<DT>(f);
    }
}
__global__ __launch_bounds__(256) void qk_rmsnorm_f32_kernel(float* __restrict__ qk, long long rows, int rows_per_sample, int n_first, int heads, int dh,
                                                             const float* __restrict__ wq, const float* __restrict__ wk,
                                                             const float* __restrict__ wq2, const float* __restrict__ wk2, float eps) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = rows * heads * 2;
    if (gid >= total) return;
    const int which = (int)(gid % 2), head = (int)((gid / 2) % heads);
    const long long row = gid / (2LL * heads);
    const int tok = (int)(row % rows_per_sample);
    const float* w = tok < n_first ? (which ? wk : wq) : (which ? wk2 : wq2);
    float* p = qk + (size_t)row * (2 * heads * dh) + (size_t)which * heads * dh + (size_t)head * dh;
    float ss = 0.f;
    for (int i = 0; i < dh; ++i) ss = fmaf(p[i], p[i], ss);
    const float r = 1.0f / sqrtf(ss / (float)dh + eps);
    for (int i = 0; i < dh; ++i) p[i] = p[i] * r * w[i];
}

int launch_qk_rmsnorm(void* qk, int dt, long long rows, int rows_per_sample, int n_first, int heads, int dh, const float* wq, const float* wk,
                      const float* wq2, const float* wk2, float eps, hipStream_t s) {
    if (dh > 64 || dh % 8 || !wq || !wk) return 1;
    if (!wq2) wq2 = wq;
    if (!wk2) wk2 = wk;
    const long long total = rows * heads * 2;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dt == DT_F32) hipLaunchKernelGGL(qk_rmsnorm_f32_kernel, grid, dim3(256), 0, s, reinterpret_cast<float*>(qk), rows, rows_per_sample, n_first, heads, dh, wq, wk, wq2, wk2, eps);
    else if (dt == DT_F16) hipLaunchKernelGGL(qk_rmsnorm_kernel<DT_F16>, grid, dim3(256), 0, s, qk, rows, rows_per_sample, n_first, heads, dh, wq, wk, wq2, wk2, eps);
    else hipLaunchKernelGGL(qk_rmsnorm_kernel<DT_BF16>, grid, dim3(256), 0, s, qk, rows, rows_per_sample, n_first, heads, dh, wq, wk, wq2, wk2, eps);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
