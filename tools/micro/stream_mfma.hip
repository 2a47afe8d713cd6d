// Micro-benchmark for the wave-chain structure of st_tail.hip: one wave per SIMD (4 waves per workgroup, one workgroup per CU), each wave
// multiplying a stream of 1-KB weight fragments (v_mfma_f32_32x32x16_f16 A operands) with its own register-resident B fragments.
//   mode 0: MFMAs only (operands in registers)           -> what one wave per SIMD issues
//   mode 1: + one ds_read_b128 per MFMA from a static LDS image, 4 reads in flight
//   mode 2: + the fragment stream arriving by LDS-DMA (global_load_lds_dwordx4) through a 6 x 20 KB ring, 5 steps ahead, one barrier
//           per 20 fragments; every workgroup streams the same STREAM_MB buffer (L2 / Infinity-Cache resident weights)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/stream_mfma.hip -o /tmp/stream_mfma && /tmp/stream_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
constexpr int SF = 20, NS = 6, D = 5;
extern __shared__ __attribute__((aligned(16))) char smem[];
__device__ __forceinline__ uint4 ldsr(unsigned addr) { return *reinterpret_cast<const uint4*>(smem + addr); }

template <int MODE, bool BAR = true, bool READ = true, bool ONEW = false, int SPREAD = 0>
__global__ __launch_bounds__(256, 1) void k(const char* __restrict__ w, int nsteps_stream, const uint4* __restrict__ x, float* out, int iters,
                                              unsigned long long* stamps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[10];
    uint4 y[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) y[i] = x[(size_t)i * 256 + threadIdx.x];
#pragma unroll
    for (int t = 0; t < 10; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const char* src = w + lane * 16;
    int step = 0;
    auto issue1 = [&](int st, int q) __attribute__((always_inline)) {   // one of the 5 pieces of this wave
        const unsigned slot = (unsigned)st % NS;
        const int ss = st % nsteps_stream;
        const int f = q * 4 + wave;
        glds16(src + ((size_t)ss * SF + f) * 1024, __builtin_amdgcn_readfirstlane((slot * SF + f) * 1024));
    };
    auto issue = [&](int st) __attribute__((always_inline)) {
        const unsigned slot = (unsigned)st % NS;
        const int ss = st % nsteps_stream;
        if constexpr (ONEW) {   // all 20 pieces of a step from wave (st % 4)
            if (wave == (st & 3)) {
#pragma unroll
                for (int f = 0; f < 20; ++f) glds16(src + ((size_t)ss * SF + f) * 1024, __builtin_amdgcn_readfirstlane((slot * SF + f) * 1024));
            }
        } else {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int f = q * 4 + wave;
            glds16(src + ((size_t)ss * SF + f) * 1024, __builtin_amdgcn_readfirstlane((slot * SF + f) * 1024));
        }
        }
    };
    if constexpr (MODE == 2) {
        for (int st = 0; st < D; ++st) issue(st);
    } else {
        for (int i = threadIdx.x; i < NS * SF * 64; i += 256) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(w)[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned cur = lane * 16, nxt = SF * 1024 + lane * 16;
    uint4 w0 = ldsr(cur), w1 = ldsr(cur + 1024), w2 = ldsr(cur + 2048), w3 = ldsr(cur + 3072);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        static_for<10>([&](auto TN) __attribute__((always_inline)) {
            constexpr int tn = decltype(TN)::value;
            static_for<SF>([&](auto I) __attribute__((always_inline)) {
                constexpr int f = decltype(I)::value;
                uint4& wf = (f % 4 == 0) ? w0 : (f % 4 == 1) ? w1 : (f % 4 == 2) ? w2 : w3;
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, y[f]), acc[tn], 0, 0, 0);
                if constexpr (MODE >= 1 && READ) {
                    if constexpr (f + 4 < SF) wf = ldsr(cur + (f + 4) * 1024); else wf = ldsr(nxt + (f + 4 - SF) * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if constexpr (MODE == 2 && SPREAD > 0 && f > 9 && (f - 9) % SPREAD == 0 && (f - 9) / SPREAD < 5) issue1(step + D, (f - 9) / SPREAD);
                if constexpr (MODE == 2 && f == 9) {
                    if constexpr (ONEW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                    if constexpr (BAR) __builtin_amdgcn_s_barrier();
                    if constexpr (SPREAD == 0) issue(step + D); else issue1(step + D, 0);
                }
            });
            step++;
            cur = nxt;
            nxt = ((unsigned)(step + 1) % NS) * (SF * 1024) + lane * 16;
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 10; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}


// mode 3: the same stream staged through registers: at the middle of step i every wave stores the 5 pieces of step i+2 it requested a step
// earlier (ds_write_b128 into a 3-slot ring) and requests its pieces of step i+3 with plain global_load_dwordx4
__global__ __launch_bounds__(256, 1) void k3(const char* __restrict__ w, int nsteps_stream, const uint4* __restrict__ x, float* out, int iters,
                                             unsigned long long* stamps) {
    constexpr int NS3 = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[10];
    uint4 y[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) y[i] = x[(size_t)i * 256 + threadIdx.x];
#pragma unroll
    for (int t = 0; t < 10; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const char* src = w + lane * 16;
    int step = 0;
    uint4 g0, g1, g2, g3, g4;
    auto gload = [&](int st) __attribute__((always_inline)) {
        const char* p = src + ((size_t)(st % nsteps_stream) * SF + wave) * 1024;
        g0 = *reinterpret_cast<const uint4*>(p); g1 = *reinterpret_cast<const uint4*>(p + 4096); g2 = *reinterpret_cast<const uint4*>(p + 8192);
        g3 = *reinterpret_cast<const uint4*>(p + 12288); g4 = *reinterpret_cast<const uint4*>(p + 16384);
    };
    auto lstore = [&](int st) __attribute__((always_inline)) {
        char* d = smem + (((unsigned)st % NS3) * SF + wave) * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(d) = g0; *reinterpret_cast<uint4*>(d + 4096) = g1; *reinterpret_cast<uint4*>(d + 8192) = g2;
        *reinterpret_cast<uint4*>(d + 12288) = g3; *reinterpret_cast<uint4*>(d + 16384) = g4;
    };
    gload(0); lstore(0); gload(1); lstore(1); gload(2);
    __syncthreads();
    unsigned cur = lane * 16, nxt = SF * 1024 + lane * 16;
    uint4 w0 = ldsr(cur), w1 = ldsr(cur + 1024), w2 = ldsr(cur + 2048), w3 = ldsr(cur + 3072);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        static_for<10>([&](auto TN) __attribute__((always_inline)) {
            constexpr int tn = decltype(TN)::value;
            static_for<SF>([&](auto I) __attribute__((always_inline)) {
                constexpr int f = decltype(I)::value;
                uint4& wf = (f % 4 == 0) ? w0 : (f % 4 == 1) ? w1 : (f % 4 == 2) ? w2 : w3;
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, y[f]), acc[tn], 0, 0, 0);
                if constexpr (f + 4 < SF) wf = ldsr(cur + (f + 4) * 1024); else wf = ldsr(nxt + (f + 4 - SF) * 1024);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if constexpr (f == 9) {
                    __builtin_amdgcn_s_barrier();
                    lstore(step + 2);
                    gload(step + 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            step++;
            cur = nxt;
            nxt = ((unsigned)(step + 1) % NS3) * (SF * 1024) + lane * 16;
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = g0.x + g1.x + g2.x + g3.x + g4.x;
#pragma unroll
    for (int t = 0; t < 10; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE, bool BAR = true, bool READ = true, bool ONEW = false, int SPREAD = 0>
void run(const char* w, int nsteps_stream, const uint4* x, float* out, unsigned long long* st, int blocks, int iters) {
    const int smem_bytes = NS * SF * 1024;
    auto kf = k<MODE == 3 ? 2 : MODE, BAR, READ, ONEW, SPREAD>;
    if constexpr (MODE == 3) kf = k3;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), smem_bytes, 0, w, nsteps_stream, x, out, iters, st);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), smem_bytes, 0, w, nsteps_stream, x, out, iters, st);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    double clk = 0, cyc = 0;
    for (int i = 0; i < blocks; ++i) { clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; cyc += (double)h[2 * i]; }
    clk /= blocks; cyc /= blocks;
    const double mfmas = (double)iters * 200.0;
    const double flops = (double)blocks * 4 * mfmas * 32768.0;
    printf("mode %d bar %d read %d onew %d spread %d blocks %4d stream %5.2f MB: %8.3f ms  %6.0f TF/s  clock %4.0f MHz  %.1f cycles/MFMA in-kernel  (L2->LDS %.1f TB/s)\n", MODE, (int)BAR, (int)READ, (int)ONEW, SPREAD, blocks,
           nsteps_stream * 20.0 / 1024, ms, flops / ms / 1e9, clk, cyc / mfmas, MODE >= 2 ? (double)blocks * mfmas * 1024 / ms / 1e9 : 0.0);
}

int main() {
    char* w; uint4* x; float* out; unsigned long long* st;
    const int max_steps = 3200;   // 64 MB
    hipMalloc(&w, (size_t)max_steps * SF * 1024); hipMalloc(&x, 20 * 256 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 16);
    std::vector<_Float16> h((size_t)max_steps * SF * 512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)((float)((i * 2654435761u) % 2001) / 1000.f - 1.f);
    hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(x, h.data(), 20 * 256 * 16, hipMemcpyHostToDevice);
    const int iters = 16;   // 3200 MFMAs per wave ~ one st_tail workgroup
    run<0>(w, 160, x, out, st, 256, iters);
    run<1>(w, 160, x, out, st, 256, iters);
    run<2>(w, 160, x, out, st, 512, iters);                              // burst of 5 LDS-DMAs per wave behind the barrier
    run<2, true, true, false, 1>(w, 160, x, out, st, 512, iters);       // one LDS-DMA per MFMA after the barrier
    run<2, true, true, false, 2>(w, 160, x, out, st, 512, iters);       // one every 2 MFMAs
    run<2>(w, 160, x, out, st, 512, iters);
    run<2, true, true, false, 2>(w, 160, x, out, st, 512, iters);
    return 0;
}
