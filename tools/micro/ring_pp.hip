// Ping-pong form of the ring GEMM against the lockstep form: bit-for-bit comparison of the outputs and interleaved timings.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iprompt-diffusion_amd/csrc tools/micro/ring_pp.hip -o /tmp/ring_pp && /tmp/ring_pp
#include "../../prompt-diffusion_amd/csrc/gemm_ring.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

static void fill_random(void* p, size_t bytes, unsigned seed) {
    std::vector<uint16_t> h(bytes / 2);
    unsigned x = seed;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((int)(x >> 9 & 0x3fff) - 8192) / 16384.0f; _Float16 hf = (_Float16)f; v = *reinterpret_cast<uint16_t*>(&hf); }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
}

int main() {
    struct Shape { int M, K, N, res, tile; };
    const Shape shapes[] = {{16384, 640, 640, 0, 0}, {16384, 640, 640, 0, 1}, {16384, 640, 640, 1, 1}, {4096, 1280, 1280, 1, 0}, {4096, 1280, 1280, 1, 1}, {16384, 2560, 640, 1, 1},
                            {16384, 640, 1920, 0, 1}, {4096, 5120, 1280, 1, 0}, {4096, 1280, 3840, 0, 0}, {65536, 320, 320, 1, 1}, {16384, 5120, 2560, 0, 1}, {8192, 8192, 8192, 0, 1},
                            {1000, 320, 324, 1, 0}, {777, 1280, 1920, 1, 1}};
    int bad = 0;
    for (const Shape& sh : shapes) {
        void *a, *w, *c0, *c1, *r;
        float* bias;
        const size_t cb = (size_t)sh.M * sh.N * 2;
        hipMalloc(&a, (size_t)sh.M * sh.K * 2); hipMalloc(&w, (size_t)sh.N * sh.K * 2); hipMalloc(&c0, cb); hipMalloc(&c1, cb); hipMalloc(&r, cb);
        hipMalloc(&bias, sh.N * 4);
        fill_random(a, (size_t)sh.M * sh.K * 2, 1); fill_random(w, (size_t)sh.N * sh.K * 2, 2); fill_random(r, cb, 3); hipMemset(bias, 0, sh.N * 4);
        hipMemset(c0, 0xff, cb); hipMemset(c1, 0xee, cb);
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = a; p.W = w; p.R = sh.res ? r : nullptr; p.bias = bias;
        p.M = sh.M; p.N = sh.N; p.K = p.Kpad = sh.K; p.lda = sh.K; p.ldc = p.ldr = sh.N;
        p.a_dt = p.c_dt = p.r_dt = DT_F16; p.taps = 1; p.Cin = sh.K; p.rows_per_sample = sh.M; p.out_scale = 1.f; p.vt_begin = sh.N; p.Nout = sh.N; p.splitk = 1;
        GemmParams p0 = p, p1 = p;
        p0.C = c0; p1.C = c1;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float t[2] = {1e9f, 1e9f};
        for (int round = 0; round < 5; ++round)
            for (int v = 0; v < 2; ++v) {
                const GemmParams& q = v ? p1 : p0;
                const int tile = sh.tile + 2 * v;
                launch_ring_gemm(q, DT_F16, tile, 0);
                hipEventRecord(e0, 0);
                for (int rep = 0; rep < 10; ++rep) launch_ring_gemm(q, DT_F16, tile, 0);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                t[v] = std::min(t[v], ms * 100.0f);
            }
        std::vector<uint16_t> h0(cb / 2), h1(cb / 2);
        hipMemcpy(h0.data(), c0, cb, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), c1, cb, hipMemcpyDeviceToHost);
        size_t diff = 0;
        for (size_t i = 0; i < h0.size(); ++i) diff += h0[i] != h1[i];
        const double fl = 2.0 * sh.M * sh.K * sh.N;
        printf("M=%6d K=%5d N=%5d res=%d tile=%d: lockstep %7.1f us (%5.0f TF/s)  ping-pong %7.1f us (%5.0f TF/s)  %+5.1f %%  differing outputs %zu\n", sh.M, sh.K, sh.N, sh.res,
               sh.tile, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, 100.0 * (t[1] / t[0] - 1.0), diff);
        fflush(stdout);
        bad += diff != 0;
        hipFree(a); hipFree(w); hipFree(c0); hipFree(c1); hipFree(r); hipFree(bias);
    }
    return bad ? 1 : 0;
}
