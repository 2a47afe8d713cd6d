// GroupNorm (+SiLU) and LayerNorm for NHWC activations on gfx950.  HBM-bound passes.
//
// GroupNorm32 (reference util.py:217-219: 32 groups, fp32 statistics) runs as two coalesced sweeps:
//   gn_stats : grid (nchunk, B); each block reads a contiguous slab of pixels x all channels with
//              16-byte loads and reduces per-group {sum, sum of squares} in fp64 -> partial[B][nchunk][G][2]
//   gn_apply : folds the partials (fp64 mean / biased variance, rstd = 1/sqrt(var+eps)), then one more
//              coalesced sweep writes (x-mean)*rstd*gamma+beta (optionally SiLU'd) in the compute type.
// LayerNorm (nn.LayerNorm, eps 1e-5, attention.py:263-265): one wave per token row, row held in
// registers, two-pass variance in fp32.
#include <type_traits>

#include "pd_common.h"

int g_gn_reg = 1;   // option "gn_reg": the register-resident single-kernel GroupNorm where it applies (process-wide)

namespace {

constexpr int GN_THREADS = 256;

template <int XD>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const void* __restrict__ x, double* __restrict__ partial,
                                                               int HW, int C, int groups, int nchunk) {
    constexpr bool XF32 = XD == DT_F32;
    constexpr int VEC = XF32 ? 4 : 8;
    __shared__ double s_sum[64], s_sq[64];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < 64) { s_sum[tid] = 0.0; s_sq[tid] = 0.0; }
    __syncthreads();
    const int ppc = (HW + nchunk - 1) / nchunk;
    const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
    const int nvec = C / VEC;
    const int lanes = min(nvec, GN_THREADS);
    const int rows_par = GN_THREADS / lanes;
    const int pr = tid / lanes, vc = tid - pr * lanes;
    const int cpg = C / groups;
    if (pr < rows_par) {
        for (int v = vc; v < nvec; v += lanes) {
            float s[VEC], q[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; }
            double ds[VEC], dq[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { ds[j] = 0.0; dq[j] = 0.0; }
            int cnt = 0;
            for (int p = p0 + pr; p < p1; p += rows_par) {
                const size_t idx = ((size_t)b * HW + p) * C + (size_t)v * VEC;
                float f[VEC];
                if constexpr (XF32) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + idx);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = t[j];
                } else {
                    const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x) + idx);
                    unpack8<XD>(t, f);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
                if (++cnt == 16) {  // flush short fp32 runs into fp64
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { ds[j] += s[j]; dq[j] += q[j]; s[j] = 0.f; q[j] = 0.f; }
                    cnt = 0;
                }
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                ds[j] += s[j];
                dq[j] += q[j];
                const int g = (v * VEC + j) / cpg;
                atomicAdd(&s_sum[g], ds[j]);
                atomicAdd(&s_sq[g], dq[j]);
            }
        }
    }
    __syncthreads();
    if (tid < groups) {
        double* o = partial + (((size_t)b * nchunk + chunk) * groups + tid) * 2;
        o[0] = s_sum[tid];
        o[1] = s_sq[tid];
    }
}

// y = x * a[c] + b[c] with a = rstd[g] * gamma[c], b = beta[c] - mean[g] * a: each thread owns one 16-byte
// channel vector (its a/b live in registers) and walks the pixels of the block's slab -- no integer
// division or per-element group lookup in the loop.
template <int XD, int YD>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                               const double* __restrict__ partial,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int HW, int C, int groups, int nchunk, int napply,
                                                               float eps, int do_silu) {
    constexpr bool XF32 = XD == DT_F32, YF32 = YD == DT_F32;
    constexpr int VEC = XF32 ? 4 : 8;  // elements per 16-byte input vector
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ double s_part[4][64][2];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    // fold the chunk partials: 4 threads per group each sum a quarter of the chunks (every block repeats this fold, so
    // its serial length -- 64 dependent fp64 loads in round 1 -- sat in front of every block's few pixels), then a
    // fixed-order sum of the 4 quarters: deterministic
    {
        const int g = tid & 63, part = tid >> 6;
        if (g < groups) {
            double s = 0.0, q = 0.0;
            for (int c = part; c < nchunk; c += 4) {
                const double* o = partial + (((size_t)b * nchunk + c) * groups + g) * 2;
                s += o[0];
                q += o[1];
            }
            s_part[part][g][0] = s;
            s_part[part][g][1] = q;
        }
    }
    __syncthreads();
    if (tid < groups) {
        const double s = (s_part[0][tid][0] + s_part[1][tid][0]) + (s_part[2][tid][0] + s_part[3][tid][0]);
        const double q = (s_part[0][tid][1] + s_part[1][tid][1]) + (s_part[2][tid][1] + s_part[3][tid][1]);
        const double n = (double)HW * (double)(C / groups);
        const double mean = s / n;
        double var = q / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / groups;
    const int nvec = C / VEC;
    const int lanes = min(nvec, GN_THREADS);
    const int rows_par = GN_THREADS / lanes;
    const int pr = tid / lanes, vc = tid - pr * lanes;
    if (pr >= rows_par) return;
    const int ppc = (HW + napply - 1) / napply;
    const int p0 = blockIdx.x * ppc, p1 = min(HW, p0 + ppc);
    const char* xb = reinterpret_cast<const char*>(x) + (size_t)b * HW * C * (XF32 ? 4 : 2);
    char* yb = reinterpret_cast<char*>(y) + (size_t)b * HW * C * (YF32 ? 4 : 2);
    for (int v = vc; v < nvec; v += lanes) {
        float ka[VEC], kb[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = v * VEC + j;
            const int g = c / cpg;
            ka[j] = s_rstd[g] * gamma[c];
            kb[j] = beta[c] - s_mean[g] * ka[j];
        }
        for (int p = p0 + pr; p < p1; p += rows_par) {
            const size_t e = (size_t)p * C + (size_t)v * VEC;   // element index inside the sample
            const uint4 raw = *reinterpret_cast<const uint4*>(xb + e * (XF32 ? 4 : 2));
            float f[VEC];
            if constexpr (XF32) {
                const float* t = reinterpret_cast<const float*>(&raw);
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = t[j];
            } else {
                unpack8<XD>(raw, f);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float r = fmaf(f[j], ka[j], kb[j]);
                f[j] = do_silu ? silu_f(r) : r;
            }
            if constexpr (YF32) {
                float* o = reinterpret_cast<float*>(yb) + e;
#pragma unroll
                for (int h = 0; h < VEC / 4; ++h) *reinterpret_cast<f32x4*>(o + 4 * h) = f32x4{f[4 * h], f[4 * h + 1], f[4 * h + 2], f[4 * h + 3]};
            } else {
                uint16_t* o = reinterpret_cast<uint16_t*>(yb) + e;
                if constexpr (VEC == 8) {
                    *reinterpret_cast<uint4*>(o) = pack8<YD>(f);
                } else {
                    uint2 u;
                    u.x = pack2<YD>(f[0], f[1]); u.y = pack2<YD>(f[2], f[3]);
                    *reinterpret_cast<uint2*>(o) = u;
                }
            }
        }
    }
}

// Single-kernel GroupNorm for the levels where one sample's slab of a few groups fits in LDS (32x32 and below):
// a block owns (sample, bundle of groups whose channels align to 16-byte vectors: lcm(channels per group, VEC)),
// loads the HW x bundle slab once into LDS, reduces the statistics in a fixed order (deterministic), then
// normalises out of LDS.  One read + one write of the tensor and one launch instead of two reads + one write in two.
constexpr int GNF_THREADS = 960;   // 15 waves; divisible by 5, 10, 15 (vectors per pixel of the bundles that occur)

// SLAB: the input is not a stored tensor but the fp32 partial sums of a split-K GEMM (nslab slabs of [rows][C], slab_stride
// floats apart) plus its bias and per-sample row (the time-embedding projection): the value a thread would have loaded is
// rebuilt as round_XD(slab_0 + slab_1 + ... + bias + row) -- splitk_finalize_kernel's order and rounding, bit for bit -- so the
// GEMM's finalize pass and the tensor it would have written disappear (x is unused).
struct GnSlabSrc { const float* slabs; long long slab_stride; const float* bias; const float* row; int nslab, row_stride; };

template <int XD, int YD, bool SLAB = false>
__global__ __launch_bounds__(GNF_THREADS) void gn_fused_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int HW, int C, int groups, int BC, float eps, int do_silu, GnSlabSrc src) {
    constexpr bool XF32 = XD == DT_F32, YF32 = YD == DT_F32;
    constexpr int VEC = XF32 ? 4 : 8;
    constexpr int EX = XF32 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) char slab[];   // [HW][BC] raw input, then [waves][4][2] doubles
    const int nb = C / BC;
    // consecutive bundles of a sample (which share 128-byte lines) on the same XCD: blocks b and b+8 share an L2
    int bid = blockIdx.x;
    {
        const int total = gridDim.x, q = total >> 3, r = total & 7, xk = bid & 7;
        bid = (xk < r ? xk * (q + 1) : r * (q + 1) + (xk - r) * q) + (bid >> 3);
    }
    const int b = bid / nb, bundle = bid - b * nb;
    const int tid = threadIdx.x;
    const int nvec = BC / VEC;              // vectors per pixel in this bundle
    const int v = tid % nvec, pr = tid / nvec;
    const int rows_par = GNF_THREADS / nvec;
    const int cpg = C / groups, gpb = BC / cpg;   // groups per bundle (<= 4)
    const int c0 = bundle * BC + v * VEC;         // first channel of this thread's vector
    double* red = reinterpret_cast<double*>(slab + (size_t)HW * BC * EX);
    const char* xb = reinterpret_cast<const char*>(x) + ((size_t)b * HW * C + c0) * EX;

    // ---- pass 1: global -> LDS, per-thread per-channel sums (short fp32 runs flushed into fp64)
    double ds[VEC], dq[VEC];
    float s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { ds[j] = 0.0; dq[j] = 0.0; s[j] = 0.f; q[j] = 0.f; }
    int cnt = 0;
    for (int p = pr; p < HW; p += rows_par) {
        uint4 raw;
        if constexpr (SLAB) {
            const float* sp = src.slabs + ((size_t)b * HW + p) * C + c0;
            float acc[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(sp + j);
                for (int k = 1; k < src.nslab; ++k) t += *reinterpret_cast<const f32x4*>(sp + (size_t)k * src.slab_stride + j);
                if (src.bias) t += *reinterpret_cast<const f32x4*>(src.bias + c0 + j);
                if (src.row) t += *reinterpret_cast<const f32x4*>(src.row + (size_t)b * src.row_stride + c0 + j);
                acc[j] = t[0]; acc[j + 1] = t[1]; acc[j + 2] = t[2]; acc[j + 3] = t[3];
            }
            if constexpr (XF32) raw = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3]));
            else raw = pack8<XD>(acc);
        } else {
            raw = *reinterpret_cast<const uint4*>(xb + (size_t)p * C * EX);
        }
        *reinterpret_cast<uint4*>(slab + ((size_t)p * nvec + v) * 16) = raw;
        float f[VEC];
        if constexpr (XF32) {
            const float* t = reinterpret_cast<const float*>(&raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = t[j];
        } else {
            unpack8<XD>(raw, f);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
        if (++cnt == 16) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { ds[j] += s[j]; dq[j] += q[j]; s[j] = 0.f; q[j] = 0.f; }
            cnt = 0;
        }
    }
    // this thread's contribution to each group of the bundle
    double gs[4] = {0.0, 0.0, 0.0, 0.0}, gq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int g = (v * VEC + j) / cpg;
        const double a = ds[j] + s[j], bq = dq[j] + q[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k == g) { gs[k] += a; gq[k] += bq; }
    }
    // wave butterfly (fixed order), then the 15 wave partials summed in wave order: bit-reproducible
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { gs[k] += __shfl_xor(gs[k], o); gq[k] += __shfl_xor(gq[k], o); }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[((tid >> 6) * 4 + k) * 2] = gs[k]; red[((tid >> 6) * 4 + k) * 2 + 1] = gq[k]; }
    }
    __syncthreads();
    __shared__ float s_mean[4], s_rstd[4];
    if (tid < gpb) {
        double S = 0.0, Q = 0.0;
        for (int t = 0; t < GNF_THREADS / 64; ++t) { S += red[(t * 4 + tid) * 2]; Q += red[(t * 4 + tid) * 2 + 1]; }
        const double n = (double)HW * (double)cpg;
        const double mean = S / n;
        double var = Q / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    // ---- pass 2: LDS -> normalise -> global
    float ka[VEC], kb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int g = (v * VEC + j) / cpg;
        ka[j] = s_rstd[g] * gamma[c0 + j];
        kb[j] = beta[c0 + j] - s_mean[g] * ka[j];
    }
    char* yb = reinterpret_cast<char*>(y) + ((size_t)b * HW * C + c0) * (YF32 ? 4 : 2);
    for (int p = pr; p < HW; p += rows_par) {
        const uint4 raw = *reinterpret_cast<const uint4*>(slab + ((size_t)p * nvec + v) * 16);
        float f[VEC];
        if constexpr (XF32) {
            const float* t = reinterpret_cast<const float*>(&raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = t[j];
        } else {
            unpack8<XD>(raw, f);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float r = fmaf(f[j], ka[j], kb[j]);
            f[j] = do_silu ? silu_f(r) : r;
        }
        if constexpr (YF32) {
            float* o = reinterpret_cast<float*>(yb + (size_t)p * C * 4);
#pragma unroll
            for (int h = 0; h < VEC / 4; ++h) *reinterpret_cast<f32x4*>(o + 4 * h) = f32x4{f[4 * h], f[4 * h + 1], f[4 * h + 2], f[4 * h + 3]};
        } else {
            uint16_t* o = reinterpret_cast<uint16_t*>(yb + (size_t)p * C * 2);
            if constexpr (VEC == 8) {
                *reinterpret_cast<uint4*>(o) = pack8<YD>(f);
            } else {
                uint2 u;
                u.x = pack2<YD>(f[0], f[1]); u.y = pack2<YD>(f[2], f[3]);
                *reinterpret_cast<uint2*>(o) = u;
            }
        }
    }
}

// Folds the GroupNorm statistics and affine parameters into per-(sample, channel) coefficients
// coef[b][c] = {a, b} with y = x * a + b, consumed by the conv kernels that normalise while staging.
__global__ __launch_bounds__(256) void gn_coef_kernel(const double* __restrict__ partial, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ coef, int HW, int C,
                                                      int groups, int nchunk, float eps) {
    __shared__ float s_mean[64], s_rstd[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int g0 = 0; g0 < groups; g0 += 32) {
        // 8 lanes per group walk the chunk partials in parallel (a serial walk of 64 dependent double loads took 18 us), then a
        // fixed-order fold across the 8 lanes: deterministic
        const int g = g0 + (tid >> 3), l8 = tid & 7;
        double s = 0.0, q = 0.0;
        if (g < groups) {
            for (int c = l8; c < nchunk; c += 8) {
                const double* o = partial + (((size_t)b * nchunk + c) * groups + g) * 2;
                s += o[0];
                q += o[1];
            }
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            s += __shfl_xor(s, d);
            q += __shfl_xor(q, d);
        }
        if (g < groups && l8 == 0) {
            const double n = (double)HW * (double)(C / groups);
            const double mean = s / n;
            double var = q / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mean[g] = (float)mean;
            s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float a = s_rstd[g] * gamma[c];
        coef[((size_t)b * C + c) * 2] = a;
        coef[((size_t)b * C + c) * 2 + 1] = beta[c] - s_mean[g] * a;
    }
}

// one wave per row; C <= 64*MAXV*4
template <int XD, int YD, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec4 = C / 4;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec4) {
            v[k] = load4(x, (size_t)row * C + (size_t)vi * 4, XD);
            s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        } else {
            v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[k][j] - mean; q = fmaf(d, d, q); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nvec4) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + vi * 4);
            const f32x4 be = *reinterpret_cast<const f32x4*>(beta + vi * 4);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[k][j] - mean) * rstd * g[j] + be[j];
            store4(y, (size_t)row * C + (size_t)vi * 4, YD, o);
        }
    }
}

// {sum, sum of squares} of each row (one wave per row): LayerNorm statistics for the GEMMs that fold the normalisation
template <int XD>
__global__ __launch_bounds__(256) void row_stats_kernel(const void* __restrict__ x, float* __restrict__ stats, int rows, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f, q = 0.f;
    for (int v = lane; v < C / 4; v += 64) {
        const f32x4 t = load4(x, (size_t)row * C + (size_t)v * 4, XD);
        s += (t[0] + t[1]) + (t[2] + t[3]);
        q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (lane == 0) { stats[(size_t)row * 2] = s; stats[(size_t)row * 2 + 1] = q; }
}

// LayerNorm folded into the consumer's weights (one wave per weight row n, fixed summation order):
//   Wout[n][k] = round_T(W[n][k] * gamma[k]),  colsum[n] = sum_k Wout[n][k],  bias_out[n] = bias_in[n] + sum_k beta[k] * W[n][k]
template <int DT>
__global__ __launch_bounds__(256) void ln_fold_kernel(const void* __restrict__ W, void* __restrict__ Wout, int N, int K, int Kpad,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias_in, float* __restrict__ colsum,
                                                      float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float cs = 0.f, bs = 0.f;
    for (int k = lane; k < Kpad; k += 64) {
        float w, wg = 0.f;
        const size_t i = (size_t)n * Kpad + k;
        if constexpr (DT == DT_F32) w = reinterpret_cast<const float*>(W)[i];
        else w = cvt32<DT>(reinterpret_cast<const uint16_t*>(W)[i]);
        if (k < K) {
            wg = w * gamma[k];
            bs += beta[k] * w;
        }
        if constexpr (DT == DT_F32) {
            reinterpret_cast<float*>(Wout)[i] = wg;
        } else {
            const uint16_t r = cvt16<DT>(wg);
            reinterpret_cast<uint16_t*>(Wout)[i] = r;
            wg = cvt32<DT>(r);      // the column sum must be that of the weights the MFMA will see
        }
        cs += wg;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cs += __shfl_xor(cs, o); bs += __shfl_xor(bs, o); }
    if (lane == 0) {
        colsum[n] = cs;
        bias_out[n] = (bias_in ? bias_in[n] : 0.f) + bs;
    }
}

// calls f(integral_constant<XD>, integral_constant<YD>) for the (input, output) type pairs the engine produces:

// Register-resident single-kernel GroupNorm (round 4) for the 2-byte modes: ONE block per (sample, group), the group's HW x cpg slab
// lives in the block's registers -- thread (pr, v) owns vector v (VB bytes = VB / 2 channels) of the pixels pr + 64 i, i < NL = HW / 64 --
// so every load is issued before the first use, there is no LDS slab and one barrier (the wave partials).  gn_fused_kernel above takes
// 15-16 us on a 2.6 MB tensor (the 8x8 level): 960-thread blocks, an LDS round trip and two barriers, whatever the size.  Statistics
// as there: fp32 runs of at most 16 values per (thread, channel), fp64 from then on, fixed order (bit-reproducible).  SLAB: the input is
// rebuilt from a split-K GEMM's fp32 slabs in splitk_finalize_kernel's order and rounding (see GnSlabSrc).
template <int XD, int VB, int NL, bool SLAB>
__global__ __launch_bounds__(NL == 16 ? 512 : 1024) void gn_reg_kernel(const void* __restrict__ x, void* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int HW, int C, int groups, int nv, float eps,
                                                       int do_silu, GnSlabSrc src) {
    constexpr int EV = VB / 2;           // channels per vector
    constexpr int NW = VB / 4;           // dwords per vector
    __shared__ double red[16][2];
    int bid = blockIdx.x;
    {   // consecutive groups of a sample (they share 128-byte lines) on the same XCD: blocks b and b + 8 share an L2
        const int total = gridDim.x, q = total >> 3, r = total & 7, xk = bid & 7;
        bid = (xk < r ? xk * (q + 1) : r * (q + 1) + (xk - r) * q) + (bid >> 3);
    }
    const int b = bid / groups, g = bid - b * groups;
    const int tid = threadIdx.x;
    const int v = tid % nv, pr = tid / nv;          // pr < 64
    const int cpg = C / groups;
    const int c0 = g * cpg + v * EV;
    unsigned raw[NL][NW];
    const char* xb = reinterpret_cast<const char*>(x) + ((size_t)b * HW * C + c0) * 2;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int p = pr + 64 * i;
        if constexpr (SLAB) {
            static_assert(!SLAB || EV >= 4, "slab input comes in 4-channel pieces");
            const float* sp = src.slabs + ((size_t)b * HW + p) * C + c0;
            float acc[EV];
#pragma unroll
            for (int j = 0; j < EV; j += 4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(sp + j);
                for (int k = 1; k < src.nslab; ++k) t += *reinterpret_cast<const f32x4*>(sp + (size_t)k * src.slab_stride + j);
                if (src.bias) t += *reinterpret_cast<const f32x4*>(src.bias + c0 + j);
                if (src.row) t += *reinterpret_cast<const f32x4*>(src.row + (size_t)b * src.row_stride + c0 + j);
                acc[j] = t[0]; acc[j + 1] = t[1]; acc[j + 2] = t[2]; acc[j + 3] = t[3];
            }
#pragma unroll
            for (int w = 0; w < NW; ++w) raw[i][w] = pack2<XD>(acc[2 * w], acc[2 * w + 1]);
        } else {
            const char* q = xb + (size_t)p * C * 2;
            if constexpr (NW == 4) {
                const uint4 t = *reinterpret_cast<const uint4*>(q);
                raw[i][0] = t.x; raw[i][1] = t.y; raw[i][2] = t.z; raw[i][3] = t.w;
            } else if constexpr (NW == 2) {
                const uint2 t = *reinterpret_cast<const uint2*>(q);
                raw[i][0] = t.x; raw[i][1] = t.y;
            } else {
                raw[i][0] = *reinterpret_cast<const unsigned*>(q);
            }
        }
    }
    // per-(thread, channel) fp32 sums over NL <= 16 values, then fp64
    float s[EV], q2[EV];
#pragma unroll
    for (int j = 0; j < EV; ++j) { s[j] = 0.f; q2[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            float lo, hi;
            unpack2<XD>(raw[i][w], lo, hi);
            s[2 * w] += lo; q2[2 * w] = fmaf(lo, lo, q2[2 * w]);
            s[2 * w + 1] += hi; q2[2 * w + 1] = fmaf(hi, hi, q2[2 * w + 1]);
        }
    double gs = 0.0, gq = 0.0;
#pragma unroll
    for (int j = 0; j < EV; ++j) { gs += (double)s[j]; gq += (double)q2[j]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { gs += __shfl_xor(gs, o); gq += __shfl_xor(gq, o); }
    if ((tid & 63) == 0) { red[tid >> 6][0] = gs; red[tid >> 6][1] = gq; }
    __syncthreads();
    double S = 0.0, Q = 0.0;
    for (int t = 0; t < nv; ++t) { S += red[t][0]; Q += red[t][1]; }   // 64 nv threads = nv waves, in wave order
    const double n = (double)HW * (double)cpg;
    const double mean = S / n;
    double var = Q / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)eps));
    float ka[EV], kb[EV];
#pragma unroll
    for (int j = 0; j < EV; ++j) {
        ka[j] = rstd_f * gamma[c0 + j];
        kb[j] = beta[c0 + j] - mean_f * ka[j];
    }
    char* yb = reinterpret_cast<char*>(y) + ((size_t)b * HW * C + c0) * 2;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        unsigned out[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            float lo, hi;
            unpack2<XD>(raw[i][w], lo, hi);
            float r0 = fmaf(lo, ka[2 * w], kb[2 * w]), r1 = fmaf(hi, ka[2 * w + 1], kb[2 * w + 1]);
            if (do_silu) { r0 = silu_f(r0); r1 = silu_f(r1); }
            out[w] = pack2<XD>(r0, r1);
        }
        char* o = yb + (size_t)(pr + 64 * i) * C * 2;
        if constexpr (NW == 4) *reinterpret_cast<uint4*>(o) = make_uint4(out[0], out[1], out[2], out[3]);
        else if constexpr (NW == 2) *reinterpret_cast<uint2*>(o) = make_uint2(out[0], out[1]);
        else *reinterpret_cast<unsigned*>(o) = out[0];
    }
}

// vector width (bytes) of gn_reg_kernel for this shape, or 0: 2-byte tensors, HW in {64, 256, 1024}, a group = nv <= 16 whole vectors
static int gn_reg_vb(int x_dt, int y_dt, int HW, int C, int groups, bool slab) {
    if (x_dt != y_dt || (x_dt != DT_F16 && x_dt != DT_BF16) || C % groups) return 0;
    if (HW != 64 && HW != 256 && HW != 1024) return 0;
    const int gb = (C / groups) * 2;   // bytes of one pixel's group
    const int nv_max = HW == 1024 ? 8 : 16;   // 16 loads per thread: blocks of at most 512 threads (256 registers each)
    for (int vb = 16; vb >= (slab ? 8 : 4); vb >>= 1)
        if (gb % vb == 0 && gb / vb <= nv_max && (C * 2) % vb == 0) return vb;
    return 0;
}

template <bool SLAB>
static int launch_gn_reg(const void* x, int dt, void* y, const float* gamma, const float* beta, int B, int HW, int C, int groups, float eps,
                         int do_silu, const GnSlabSrc& src, hipStream_t s) {
    const int vb = gn_reg_vb(dt, dt, HW, C, groups, true);
    if (!vb) return 1;
    const int nv = (C / groups) * 2 / vb, nl = HW / 64;
    const dim3 grid(B * groups), block(64 * nv);
    bool done = false;
    auto go = [&](auto XD, auto VB, auto NL) {
        if constexpr (SLAB && decltype(VB)::value < 8) return;
        else {
            hipLaunchKernelGGL((gn_reg_kernel<decltype(XD)::value, decltype(VB)::value, decltype(NL)::value, SLAB>), grid, block, 0, s, x, y, gamma, beta,
                               HW, C, groups, nv, eps, do_silu, src);
            done = true;
        }
    };
    auto by_nl = [&](auto XD, auto VB) {
        if (nl == 1) go(XD, VB, std::integral_constant<int, 1>{});
        else if (nl == 4) go(XD, VB, std::integral_constant<int, 4>{});
        else go(XD, VB, std::integral_constant<int, 16>{});
    };
    auto by_vb = [&](auto XD) {
        if (vb == 16) by_nl(XD, std::integral_constant<int, 16>{});
        else if (vb == 8) by_nl(XD, std::integral_constant<int, 8>{});
        else by_nl(XD, std::integral_constant<int, 4>{});
    };
    if (dt == DT_F16) by_vb(std::integral_constant<int, DT_F16>{});
    else by_vb(std::integral_constant<int, DT_BF16>{});
    return done && hipGetLastError() == hipSuccess ? 0 : 1;
}

// fp32 on either side, or the same 2-byte flavour on both
template <typename F>
bool dispatch_xy(int x_dt, int y_dt, F&& f) {
    using IF = std::integral_constant<int, DT_F32>;
    using IB = std::integral_constant<int, DT_BF16>;
    using IH = std::integral_constant<int, DT_F16>;
    if (x_dt == DT_F32 && y_dt == DT_F32) f(IF{}, IF{});
    else if (x_dt == DT_F32 && y_dt == DT_BF16) f(IF{}, IB{});
    else if (x_dt == DT_F32 && y_dt == DT_F16) f(IF{}, IH{});
    else if (x_dt == DT_BF16 && y_dt == DT_F32) f(IB{}, IF{});
    else if (x_dt == DT_F16 && y_dt == DT_F32) f(IH{}, IF{});
    else if (x_dt == DT_BF16 && y_dt == DT_BF16) f(IB{}, IB{});
    else if (x_dt == DT_F16 && y_dt == DT_F16) f(IH{}, IH{});
    else return false;
    return true;
}

}  // namespace

int launch_gn_stats(const void* x, int x_dt, double* partial, int B, int HW, int C, int groups, int nchunk, hipStream_t s) {
    if (groups > 64 || C % groups || C % 8) return 1;
    dim3 grid(nchunk, B);
    if (x_dt == DT_F32)
        hipLaunchKernelGGL(gn_stats_kernel<DT_F32>, grid, dim3(GN_THREADS), 0, s, x, partial, HW, C, groups, nchunk);
    else if (x_dt == DT_F16)
        hipLaunchKernelGGL(gn_stats_kernel<DT_F16>, grid, dim3(GN_THREADS), 0, s, x, partial, HW, C, groups, nchunk);
    else
        hipLaunchKernelGGL(gn_stats_kernel<DT_BF16>, grid, dim3(GN_THREADS), 0, s, x, partial, HW, C, groups, nchunk);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_gn_apply(const void* x, int x_dt, void* y, int y_dt, const double* partial, const float* gamma,
                    const float* beta, int B, int HW, int C, int groups, int nchunk, float eps, int silu, hipStream_t s) {
    // pixel slabs per sample: aim at >= 1024 blocks, at least 16 pixels per thread row (every block folds the statistics)
    int napply = (1024 + B - 1) / B;
    if (napply > HW / 16) napply = HW / 16;
    if (napply < 1) napply = 1;
    dim3 grid(napply, B);
    const bool ok = dispatch_xy(x_dt, y_dt, [&](auto XD, auto YD) {
        hipLaunchKernelGGL((gn_apply_kernel<decltype(XD)::value, decltype(YD)::value>), grid, dim3(GN_THREADS), 0, s, x, y, partial,
                           gamma, beta, HW, C, groups, nchunk, napply, eps, silu);
    });
    return ok && hipGetLastError() == hipSuccess ? 0 : 1;
}

// bundle width (channels) for gn_fused_kernel, or 0 when the shape is not eligible (slab beyond the LDS budget)
static int gn_slab_bundle(int x_dt, int HW, int C, int groups);
// > 0 when a single-kernel GroupNorm applies to this shape (either input form): the register-resident kernel, or gn_fused_kernel's bundle
int gn_fused_bundle(int x_dt, int HW, int C, int groups) {
    if (C % groups) return 0;
    if (g_gn_reg && gn_reg_vb(x_dt, x_dt, HW, C, groups, true)) return C / groups;
    return gn_slab_bundle(x_dt, HW, C, groups);
}
static int gn_slab_bundle(int x_dt, int HW, int C, int groups) {
    const int VEC = x_dt == DT_F32 ? 4 : 8, EX = x_dt == DT_F32 ? 4 : 2;
    if (C % groups) return 0;
    const int cpg = C / groups;
    int BC = cpg;
    while (BC % VEC) BC += cpg;           // lcm(cpg, VEC)
    if (C % BC || BC / cpg > 4 || GNF_THREADS % (BC / VEC)) return 0;
    const size_t bytes = (size_t)HW * BC * EX + (GNF_THREADS / 64) * 4 * 2 * sizeof(double);
    return bytes <= 100 * 1024 ? BC : 0;
}

int launch_gn_fused(const void* x, int x_dt, void* y, int y_dt, const float* gamma, const float* beta, int B, int HW, int C, int groups,
                    float eps, int do_silu, hipStream_t s) {
    // (the same eligibility rule for both input forms: a shape must take the same kernel -- the same summation order -- whether its input
    // is a stored tensor or a split-K GEMM's slabs: test_split_k_slabs_into_groupnorm_is_bit_identical)
    if (g_gn_reg && gn_reg_vb(x_dt, y_dt, HW, C, groups, true))
        return launch_gn_reg<false>(x, x_dt, y, gamma, beta, B, HW, C, groups, eps, do_silu, GnSlabSrc{}, s);
    const int BC = gn_slab_bundle(x_dt, HW, C, groups);
    if (!BC) return 1;
    const int EX = x_dt == DT_F32 ? 4 : 2;
    const size_t smem = (size_t)HW * BC * EX + (GNF_THREADS / 64) * 4 * 2 * sizeof(double);
    const dim3 grid(B * (C / BC));
    bool fail = false;
    const bool ok = dispatch_xy(x_dt, y_dt, [&](auto XD, auto YD) {
        static unsigned long long attr_done = 0;   // one per (XD, YD) instantiation of this lambda
        auto kfn = gn_fused_kernel<decltype(XD)::value, decltype(YD)::value>;
        if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), 104 * 1024, &attr_done)) { fail = true; return; }
        hipLaunchKernelGGL(kfn, grid, dim3(GNF_THREADS), smem, s, x, y, gamma, beta, HW, C, groups, BC, eps, do_silu, GnSlabSrc{});
    });
    return ok && !fail && hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_gn_fused_slabs(const float* slabs, int nslab, const float* bias, const float* rowvec, int rowvec_stride, int x_dt, void* y, int y_dt,
                          const float* gamma, const float* beta, int B, int HW, int C, int groups, float eps, int do_silu, hipStream_t s) {
    if (!slabs || nslab < 1 || C % 4) return 1;
    if (g_gn_reg && gn_reg_vb(x_dt, y_dt, HW, C, groups, true))
        return launch_gn_reg<true>(nullptr, x_dt, y, gamma, beta, B, HW, C, groups, eps, do_silu,
                                   GnSlabSrc{slabs, (long long)B * HW * C, bias, rowvec, nslab, rowvec_stride}, s);
    const int BC = gn_slab_bundle(x_dt, HW, C, groups);
    if (!BC) return 1;
    const int EX = x_dt == DT_F32 ? 4 : 2;
    const size_t smem = (size_t)HW * BC * EX + (GNF_THREADS / 64) * 4 * 2 * sizeof(double);
    const dim3 grid(B * (C / BC));
    const GnSlabSrc src{slabs, (long long)B * HW * C, bias, rowvec, nslab, rowvec_stride};
    bool fail = false;
    const bool ok = dispatch_xy(x_dt, y_dt, [&](auto XD, auto YD) {
        static unsigned long long attr_done = 0;   // one per (XD, YD) instantiation of this lambda
        auto kfn = gn_fused_kernel<decltype(XD)::value, decltype(YD)::value, true>;
        if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), 104 * 1024, &attr_done)) { fail = true; return; }
        hipLaunchKernelGGL(kfn, grid, dim3(GNF_THREADS), smem, s, static_cast<const void*>(nullptr), y, gamma, beta, HW, C, groups, BC, eps, do_silu, src);
    });
    return ok && !fail && hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_gn_coef(const double* partial, const float* gamma, const float* beta, float* coef, int B, int HW, int C, int groups,
                   int nchunk, float eps, hipStream_t s) {
    hipLaunchKernelGGL(gn_coef_kernel, dim3(B), dim3(256), 0, s, partial, gamma, beta, coef, HW, C, groups, nchunk, eps);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_row_stats(const void* x, int x_dt, float* stats, int rows, int C, hipStream_t s) {
    if (C % 4) return 1;
    dim3 grid((rows + 3) / 4);
    if (x_dt == DT_F32) hipLaunchKernelGGL(row_stats_kernel<DT_F32>, grid, dim3(256), 0, s, x, stats, rows, C);
    else if (x_dt == DT_F16) hipLaunchKernelGGL(row_stats_kernel<DT_F16>, grid, dim3(256), 0, s, x, stats, rows, C);
    else hipLaunchKernelGGL(row_stats_kernel<DT_BF16>, grid, dim3(256), 0, s, x, stats, rows, C);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_ln_fold(const void* W, void* Wout, int dt, int N, int K, int Kpad, const float* gamma, const float* beta, const float* bias_in,
                   float* colsum, float* bias_out, hipStream_t s) {
    dim3 grid((N + 3) / 4);
    if (dt == DT_F32) hipLaunchKernelGGL(ln_fold_kernel<DT_F32>, grid, dim3(256), 0, s, W, Wout, N, K, Kpad, gamma, beta, bias_in, colsum, bias_out);
    else if (dt == DT_F16) hipLaunchKernelGGL(ln_fold_kernel<DT_F16>, grid, dim3(256), 0, s, W, Wout, N, K, Kpad, gamma, beta, bias_in, colsum, bias_out);
    else hipLaunchKernelGGL(ln_fold_kernel<DT_BF16>, grid, dim3(256), 0, s, W, Wout, N, K, Kpad, gamma, beta, bias_in, colsum, bias_out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int launch_layernorm(const void* x, int x_dt, void* y, int y_dt, const float* gamma, const float* beta, int rows,
                     int C, float eps, hipStream_t s) {
    if (C % 4 || C > 64 * 4 * 8) return 1;
    dim3 grid((rows + 3) / 4);
    const int nv = (C / 4 + 63) / 64;
    const bool ok = dispatch_xy(x_dt, y_dt, [&](auto XD, auto YD) {
        constexpr int xd = decltype(XD)::value, yd = decltype(YD)::value;
        if (nv <= 2) hipLaunchKernelGGL((layernorm_kernel<xd, yd, 2>), grid, dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
        else if (nv <= 5) hipLaunchKernelGGL((layernorm_kernel<xd, yd, 5>), grid, dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
        else hipLaunchKernelGGL((layernorm_kernel<xd, yd, 8>), grid, dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
    });
    return ok && hipGetLastError() == hipSuccess ? 0 : 1;
}
