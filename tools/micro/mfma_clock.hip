// Micro-benchmark: what a dense fp16 MFMA loop actually sustains on this MI355X, and at which shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
// Every wave runs ITERS x 16 independent v_mfma_f32_16x16x32_f16 on random register operands (no memory traffic);
// clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void mfma_loop(const float* in, float* out, unsigned long long* stamps, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(threadIdx.x * 8 + i) & 1023]; b[i] = (_Float16)in[(threadIdx.x * 8 + i + 7) & 1023]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
    for (int i = 1; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int W>
void run(const float* in, float* out, unsigned long long* st, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_loop<W>, dim3(blocks), dim3(256 * W), 0, 0, in, out, st, iters);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(mfma_loop<W>, dim3(blocks), dim3(256 * W), 0, 0, in, out, st, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    double clk = 0;
    for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
    clk /= blocks;
    const double flops = (double)blocks * 4 * W * iters * 16 * 16384.0;
    printf("waves/SIMD %d: %.3f ms  %.0f TF/s  shader clock %.0f MHz  -> peak at this clock %.0f TF/s\n", W, ms, flops / ms / 1e9, clk,
           256.0 * 4 * 1024 * clk * 1e6 / 1e12);
}

int main() {
    float *in, *out; unsigned long long* st;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4 * 4); hipMalloc(&st, 4096 * 16);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<1>(in, out, st, 256, 20000);
    run<2>(in, out, st, 256, 20000);
    run<1>(in, out, st, 512, 20000);
    return 0;
}
