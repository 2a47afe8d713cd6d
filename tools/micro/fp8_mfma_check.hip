#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// C[16x16] = A[16x128] * B[16x128]^T in fp8 e4m3; A as MFMA "A" operand rows, B as "B" operand cols
__global__ void k(const uint8_t* A, const uint8_t* B, float* C) {
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    i32x8 a, b;
    const int* ar = reinterpret_cast<const int*>(A + fr * 128 + fq * 32);
    const int* br = reinterpret_cast<const int*>(B + fr * 128 + fq * 32);
    for (int j = 0; j < 8; ++j) { a[j] = ar[j]; b[j] = br[j]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    // C/D: col = lane&15, row = (lane>>4)*4 + reg
    for (int r = 0; r < 4; ++r) C[(fq * 4 + r) * 16 + fr] = c[r];
}
__global__ void cvt(const float* x, uint8_t* y, int n) {
    int i = threadIdx.x;
    if (2 * i + 1 < n) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
        y[2 * i] = w & 0xff; y[2 * i + 1] = (w >> 8) & 0xff;
    }
}
static float e4m3_to_f(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
    if (e == 15 && m == 7) f = NAN;
    return s ? -f : f;
}
int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    for (int i = 0; i < 16 * 128; ++i) { A[i] = (uint8_t)((i * 37 + 11) % 120); B[i] = (uint8_t)(((i * 53 + 7) % 120) | ((i % 3 == 0) ? 0x80 : 0)); }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    double maxerr = 0, maxerrT = 0, maxref = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double r = 0;
        for (int kk = 0; kk < 128; ++kk) r += (double)e4m3_to_f(A[i * 128 + kk]) * e4m3_to_f(B[j * 128 + kk]);
        maxerr = fmax(maxerr, fabs(r - C[i * 16 + j])); maxerrT = fmax(maxerrT, fabs(r - C[j * 16 + i])); maxref = fmax(maxref, fabs(r));
    }
    printf("C[i=A row][j=B row]: maxerr %g   transposed: %g   (max |ref| %g)\n", maxerr, maxerrT, maxref);
    // convert check
    std::vector<float> x = {0.f, 1.f, -1.5f, 447.f, 448.f, 500.f, 1e6f, 0.0019f, 0.001f, -0.06f, 17.3f, 240.f, 0.4375f, 3.3f, -464.f, 460.f};
    float* dx; uint8_t* dy; hipMalloc(&dx, 64); hipMalloc(&dy, 16);
    hipMemcpy(dx, x.data(), 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(1), dim3(8), 0, 0, dx, dy, 16);
    uint8_t y[16]; hipMemcpy(y, dy, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%g -> 0x%02x = %g\n", x[i], y[i], e4m3_to_f(y[i]));
    return 0;
}
