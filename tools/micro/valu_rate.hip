// Issue rate of the vector instructions the attention softmax is made of, one wave per SIMD and two: cycles per instruction (s_memtime / count).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body, setup)                                                                      \
    __global__ __launch_bounds__(512) void name(float* out, unsigned long long* cyc, int iters) {      \
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;                  \
        setup;                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int i = 0; i < iters; ++i) { REP8(body) }                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                                \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                               \
    }

KERNEL(k_exp32, asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, )
KERNEL(k_exp16, asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, )
KERNEL(k_rcp32, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, )
KERNEL(k_fma32, asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, )
KERNEL(k_pkfma32, asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1" : "+v"(d0), "+v"(d1));, double d0 = a0; double d1 = a1; a2 += (float)d0 + (float)d1)
KERNEL(k_pkfma16, asm volatile("v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, )
KERNEL(k_cvtpk, asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, )
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, )

template <typename K>
static void run(const char* name, K k, int per_body, int waves_per_simd) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    // s_memtime runs at 100 MHz on this chip; report ticks per instruction and let the fp32 FMA (4 cycles per wave instruction) calibrate
    printf("%-10s %d wave(s)/SIMD: %.4f ticks per instruction and wave\n", name, waves_per_simd, s / 256 / ((double)iters * 8 * per_body));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run("fma_f32", k_fma32, 4, w); run("pk_fma_f32", k_pkfma32, 2, w); run("pk_fma_f16", k_pkfma16, 4, w); run("exp_f32", k_exp32, 4, w); run("exp_f16", k_exp16, 4, w);
        run("rcp_f32", k_rcp32, 4, w); run("cvt_pk_f16", k_cvtpk, 4, w); run("max3_f32", k_max3, 4, w);
    }
    return 0;
}
