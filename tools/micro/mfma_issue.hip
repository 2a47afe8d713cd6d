// How fast does ONE wave issue MFMAs?  cycles per instruction (s_memtime) for v_mfma_f32_16x16x32 and v_mfma_f32_32x32x16 (f16 / bf16),
// 1 or 2 waves per SIMD, with one shared operand pair or distinct operand registers per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_issue.hip -o tools/micro/bin/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int SHAPE, int BF, int DISTINCT, int W>
__global__ __launch_bounds__(256 * W) void k(const float* in, float* out, unsigned long long* stamps, int iters) {
    f16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) { a[j][i] = (_Float16)in[(threadIdx.x * 8 + i + 13 * j) & 1023]; b[j][i] = (_Float16)in[(threadIdx.x * 8 + i + 7 + 29 * j) & 1023]; }
    float res = 0.f;
    unsigned long long t0, t1;
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ia = DISTINCT ? (i & 3) : 0, ib = DISTINCT ? (i >> 2) : 0;
                if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[ia]), __builtin_bit_cast(bf16x8, b[ib]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia], b[ib], acc[i], 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ia = DISTINCT ? (i & 3) : 0, ib = DISTINCT ? (i >> 2) : 0;
                if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ia]), __builtin_bit_cast(bf16x8, b[ib]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[i], 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int BF, int DISTINCT, int W>
void run(const float* in, float* out, unsigned long long* st) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, BF, DISTINCT, W>), dim3(blocks), dim3(256 * W), 0, 0, in, out, st, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE, BF, DISTINCT, W>), dim3(blocks), dim3(256 * W), 0, 0, in, out, st, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < blocks; ++i) cyc += (double)h[i * 8];
    cyc /= blocks;
    const int per = SHAPE == 16 ? 16 : 8;
    const double flops = (double)blocks * 4 * W * iters * per * (SHAPE == 16 ? 16384.0 : 32768.0);
    printf("%s %s %s operands, %d wave(s)/SIMD: %.1f cycles per MFMA per wave (%.1f per SIMD), %.0f TF/s\n", SHAPE == 16 ? "16x16x32" : "32x32x16", BF ? "bf16" : "f16 ",
           DISTINCT ? "distinct" : "shared  ", W, cyc / (iters * per), cyc / (iters * per) / W, flops / ms / 1e9);
}

int main() {
    float *in, *out; unsigned long long* st;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 8);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<16, 0, 0, 1>(in, out, st); run<16, 0, 1, 1>(in, out, st); run<16, 1, 1, 1>(in, out, st); run<16, 0, 1, 2>(in, out, st);
    run<32, 0, 0, 1>(in, out, st); run<32, 0, 1, 1>(in, out, st); run<32, 1, 1, 1>(in, out, st); run<32, 0, 1, 2>(in, out, st);
    return 0;
}
