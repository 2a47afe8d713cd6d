// Layout conversion, concat, CFG + DDIM update and device-side weight initialisation (gfx950).
// All of these are HBM-trivial next to the contractions; they exist to keep the whole denoising loop
// on the device with no host round trip per step.
#include "pd_common.h"

namespace {

constexpr int TPB = 256;
inline int nblocks(long long n, int per = TPB, int cap = 65535 * 16) {
    long long b = (n + per - 1) / per;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, void* __restrict__ out, int out_dt, int B, int C,
                                    int HW, int Cpad, float scale) {
    const long long total = (long long)B * HW * Cpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long long bp = i / Cpad;
        const int p = (int)(bp % HW);
        const int b = (int)(bp / HW);
        const float v = c < C ? scale * in[((long long)b * C + c) * HW + p] : 0.f;
        if (out_dt == DT_F32) reinterpret_cast<float*>(out)[i] = v;
        else reinterpret_cast<uint16_t*>(out)[i] = cvt16_rt(v, out_dt);
    }
}

// row softmax of fp32 logits (one wave per row), output in the compute type: the VAE's single-head attention
// (AttnBlock, model.py:171-202) materialises its [N, N] scores like the reference does
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in, void* __restrict__ out, int out_dt,
                                                            int rows, int n) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = in + (size_t)row * n;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += __expf(x[i] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int i = lane; i < n; i += 64) {
        const float v = __expf(x[i] - m) * inv;
        if (out_dt == DT_F32) reinterpret_cast<float*>(out)[(size_t)row * n + i] = v;
        else reinterpret_cast<uint16_t*>(out)[(size_t)row * n + i] = cvt16_rt(v, out_dt);
    }
}

__global__ void embed_tokens_kernel(const int* __restrict__ ids, const void* __restrict__ tok, int tok_ld, const void* __restrict__ pos,
                                    int pos_ld, int dt, void* __restrict__ out, int out_dt, int B, int L, int C, int vocab) {
    const long long total = (long long)B * L * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long bl = i / C;
        const int l = (int)(bl % L);
        int id = ids[bl];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        float v;
        if (dt == DT_F32)
            v = reinterpret_cast<const float*>(tok)[(size_t)id * tok_ld + c] + reinterpret_cast<const float*>(pos)[(size_t)l * pos_ld + c];
        else
            v = cvt32_rt(reinterpret_cast<const uint16_t*>(tok)[(size_t)id * tok_ld + c], dt) + cvt32_rt(reinterpret_cast<const uint16_t*>(pos)[(size_t)l * pos_ld + c], dt);
        if (out_dt == DT_F32) reinterpret_cast<float*>(out)[i] = v;
        else reinterpret_cast<uint16_t*>(out)[i] = cvt16_rt(v, out_dt);
    }
}

__global__ void nhwc_to_nchw_kernel(const void* __restrict__ in, int in_dt, float* __restrict__ out, int B, int C,
                                    int HW, int Cpad, float scale) {
    const long long total = (long long)B * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % C);
        const int b = (int)(bc / C);
        const long long j = ((long long)b * HW + p) * Cpad + c;
        const float v = in_dt == DT_F32 ? reinterpret_cast<const float*>(in)[j] : cvt32_rt(reinterpret_cast<const uint16_t*>(in)[j], in_dt);
        out[i] = v * scale;
    }
}

__global__ void cast_rows_kernel(const float* __restrict__ in, void* __restrict__ out, int out_dt, long long rows, int C,
                                 int Cpad) {
    const long long total = rows * Cpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long long r = i / Cpad;
        const float v = c < C ? in[r * C + c] : 0.f;
        if (out_dt == DT_F32) reinterpret_cast<float*>(out)[i] = v;
        else reinterpret_cast<uint16_t*>(out)[i] = cvt16_rt(v, out_dt);
    }
}

// out[r, 0:Ca] = a[r] (+ a_add[r]);  out[r, Ca:Ca+Cb] = b[r] (+ b_add[r])   -- the skip concat of
// ControlledUnetModel.forward (cldm/cldm.py:35,41) with the control residual adds folded in.
__global__ void concat_add_kernel(const void* __restrict__ a, const void* __restrict__ a_add, const void* __restrict__ b,
                                  const void* __restrict__ b_add, void* __restrict__ out, int dt, long long rows, int Ca,
                                  int Cb, long long b_rows, long long b_add_rows) {
    const int C4 = (Ca + Cb) / 4;
    const long long total = rows * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i % C4);
        const long long r = i / C4;
        const int c = v * 4;
        f32x4 x;
        if (c < Ca) {
            x = load4(a, (size_t)r * Ca + c, dt);
            if (a_add) x += load4(a_add, (size_t)r * Ca + c, dt);
        } else {
            // b / b_add may hold only the first b_rows / b_add_rows rows (tensors shared by the two halves of a CFG batch)
            // (rows <= 2 * b_rows is checked by the launcher: one conditional subtraction instead of a 64-bit modulo)
            x = load4(b, (size_t)(r >= b_rows ? r - b_rows : r) * Cb + (c - Ca), dt);
            if (b_add) x += load4(b_add, (size_t)(r >= b_add_rows ? r - b_add_rows : r) * Cb + (c - Ca), dt);
        }
        store4(out, (size_t)r * (Ca + Cb) + c, dt, x);
    }
}

__global__ void add_inplace_kernel(void* __restrict__ a, const void* __restrict__ b, int dt, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 x = load4(a, (size_t)i * 4, dt);
        x += load4(b, (size_t)i * 4, dt);
        store4(a, (size_t)i * 4, dt, x);
    }
}

// Classifier-free guidance + DDIM update, DDIMSampler.p_sample_ddim (cldm/ddim_hacked.py:193,218,229-233),
// in the reference's own fp32 operation order (no FMA contraction).
//   eps      [Bf, HW, eps_C] fp32/bf16: UNet output, uncond half first (ddim_hacked.py:189-192)
//   x_state  [B, HW, Cpad] fp32 (channels >= C are zero)    -> updated in place
//   x_in     [dup*B, HW, Cpad]: the CFG-duplicated latents the next step's conv_in reads
__global__ void cfg_ddim_kernel(const void* __restrict__ eps, int eps_dt, int eps_C, float* __restrict__ x_state,
                                float* __restrict__ pred_x0, float* __restrict__ eps_guided, void* __restrict__ x_in,
                                int x_in_dt, const float* __restrict__ noise, int B, int HW, int C, int Cpad, int use_cfg,
                                DdimCoef k, float temperature, int do_update) {
    const long long total = (long long)B * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long bp = i / C;
        const int p = (int)(bp % HW);
        const int b = (int)(bp / HW);
        auto ld = [&](long long j) {
            return eps_dt == DT_F32 ? reinterpret_cast<const float*>(eps)[j] : cvt32_rt(reinterpret_cast<const uint16_t*>(eps)[j], eps_dt);
        };
        float e;
        if (use_cfg) {
            const float eu = ld(((long long)b * HW + p) * eps_C + c);
            const float ec = ld(((long long)(B + b) * HW + p) * eps_C + c);
            e = __fadd_rn(eu, __fmul_rn(k.cfg_scale, __fsub_rn(ec, eu)));
        } else {
            e = ld(((long long)b * HW + p) * eps_C + c);
        }
        eps_guided[i] = e;
        if (!do_update) continue;
        const long long xi = ((long long)b * HW + p) * Cpad + c;
        const float x = x_state[xi];
        const float pred = __fdiv_rn(__fsub_rn(x, __fmul_rn(k.sqrt_one_minus_at, e)), k.sqrt_at);
        const float dir = __fmul_rn(k.dir_coef, e);
        float xp = __fadd_rn(__fmul_rn(k.sqrt_a_prev, pred), dir);
        if (noise) {
            const float nz = __fmul_rn(__fmul_rn(k.sigma, noise[((long long)b * C + c) * HW + p]), temperature);
            xp = __fadd_rn(xp, nz);
        }
        pred_x0[i] = pred;
        x_state[xi] = xp;
        if (x_in_dt == DT_F32) {
            float* xo = reinterpret_cast<float*>(x_in);
            xo[xi] = xp;
            if (use_cfg) xo[(long long)B * HW * Cpad + xi] = xp;
        } else {
            uint16_t* xo = reinterpret_cast<uint16_t*>(x_in);
            xo[xi] = cvt16_rt(xp, x_in_dt);
            if (use_cfg) xo[(long long)B * HW * Cpad + xi] = cvt16_rt(xp, x_in_dt);
        }
    }
}

// x_in[d*B + b] = x_state[b] for d < dup   (torch.cat([x]*2), ddim_hacked.py:189)
__global__ void dup_rows_kernel(const float* __restrict__ x_state, float* __restrict__ x_in, long long n, int dup) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x_state[i];
        for (int d = 0; d < dup; ++d) x_in[(long long)d * n + i] = v;
    }
}

__device__ __forceinline__ uint64_t splitmix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// value = shift + scale * N(0,1)  (Box-Muller on a counter hash); benchmark-only weight init
__global__ void fill_random_kernel(void* __restrict__ p, int dt, long long n, float scale, float shift, uint64_t seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint64_t h = splitmix(seed ^ splitmix((uint64_t)i));
        const float u1 = ((float)((h >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777217.0f);
        const float u2 = (float)((h >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
        const float z = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
        const float v = shift + scale * z;
        if (dt == DT_F32) reinterpret_cast<float*>(p)[i] = v;
        else reinterpret_cast<uint16_t*>(p)[i] = cvt16_rt(v, dt);
    }
}

}  // namespace

#define CHECK_LAUNCH() return hipGetLastError() == hipSuccess ? 0 : 1

int launch_nchw_to_nhwc(const float* in, void* out, int out_dt, int B, int C, int H, int W, int Cpad, hipStream_t s, float scale) {
    const long long n = (long long)B * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, in, out, out_dt, B, C, H * W, Cpad, scale);
    CHECK_LAUNCH();
}
int launch_softmax_rows(const float* in, void* out, int out_dt, int rows, int n, hipStream_t s) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in, out, out_dt, rows, n);
    CHECK_LAUNCH();
}
int launch_embed_tokens(const int* ids, const void* tok, int tok_ld, const void* pos, int pos_ld, int dt, void* out, int out_dt,
                        int B, int L, int C, int vocab, hipStream_t s) {
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(nblocks((long long)B * L * C)), dim3(TPB), 0, s, ids, tok, tok_ld, pos, pos_ld, dt,
                       out, out_dt, B, L, C, vocab);
    CHECK_LAUNCH();
}
int launch_nhwc_to_nchw(const void* in, int in_dt, float* out, int B, int C, int H, int W, int Cpad, float scale, hipStream_t s) {
    const long long n = (long long)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, in, in_dt, out, B, C, H * W, Cpad, scale);
    CHECK_LAUNCH();
}
int launch_cast_rows(const float* in, void* out, int out_dt, long long rows, int C, int Cpad, hipStream_t s) {
    hipLaunchKernelGGL(cast_rows_kernel, dim3(nblocks(rows * Cpad)), dim3(TPB), 0, s, in, out, out_dt, rows, C, Cpad);
    CHECK_LAUNCH();
}
int launch_concat_add(const void* a, const void* a_add, const void* b, const void* b_add, void* out, int dt, long long rows,
                      int Ca, int Cb, hipStream_t s, long long b_rows, long long b_add_rows) {
    if (Ca % 4 || Cb % 4) return 1;
    if (b_rows <= 0) b_rows = rows;
    if (b_add_rows <= 0) b_add_rows = rows;
    if (rows > 2 * b_rows || rows > 2 * b_add_rows) return 1;
    hipLaunchKernelGGL(concat_add_kernel, dim3(nblocks(rows * ((Ca + Cb) / 4), TPB, 8192)), dim3(TPB), 0, s, a, a_add, b, b_add,
                       out, dt, rows, Ca, Cb, b_rows, b_add_rows);
    CHECK_LAUNCH();
}
int launch_add_inplace(void* a, const void* b, int dt, long long n, hipStream_t s) {
    if (n % 4) return 1;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(nblocks(n / 4, TPB, 8192)), dim3(TPB), 0, s, a, b, dt, n / 4);
    CHECK_LAUNCH();
}
int launch_cfg_ddim(const void* eps, int eps_dt, int eps_C, float* x_state, float* pred_x0, float* eps_guided, void* x_in,
                    const float* noise, int B, int HW, int C, int Cpad, int use_cfg, DdimCoef k, float temperature,
                    int do_update, hipStream_t s) {
    const long long n = (long long)B * HW * C;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, eps, eps_dt, eps_C, x_state, pred_x0, eps_guided,
                       x_in, (int)DT_F32, noise, B, HW, C, Cpad, use_cfg, k, temperature, do_update);
    CHECK_LAUNCH();
}
int launch_fill_x_in(const float* x_state, float* x_in, int B, int dup, int C, int Cpad, int HW, hipStream_t s) {
    (void)C;
    const long long n = (long long)B * HW * Cpad;
    hipLaunchKernelGGL(dup_rows_kernel, dim3(nblocks(n)), dim3(TPB), 0, s, x_state, x_in, n, dup);
    CHECK_LAUNCH();
}
int launch_fill_random(void* p, int dt, long long n, float scale, float shift, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_random_kernel, dim3(nblocks(n, TPB, 16384)), dim3(TPB), 0, s, p, dt, n, scale, shift, seed);
    CHECK_LAUNCH();
}
