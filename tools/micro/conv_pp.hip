// The 4-wave patch conv (conv_patch4.hip; column "ping-pong" is the retired 8-wave ping-pong experiment, now skipped) against the first two generations: bit-for-bit comparison of the outputs on random fp16
// operands (image borders, split-K slabs, the fused upsample) and interleaved timings.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iprompt-diffusion_amd/csrc tools/micro/conv_pp.hip -Lprompt-diffusion_amd/csrc -lpdengine
//         -Wl,-rpath,'$ORIGIN/../../../prompt-diffusion_amd/csrc' -o tools/micro/bin/conv_pp
#include "pd_common.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

static void fill_random(void* p, size_t bytes, unsigned seed, float scale) {
    std::vector<uint16_t> h(bytes / 2);
    unsigned x = seed;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = scale * ((int)(x >> 9 & 0x3fff) - 8192) / 16384.0f; _Float16 hf = (_Float16)f; v = *reinterpret_cast<uint16_t*>(&hf); }
    (void)hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
}

int main() {
    struct Shape { int B, H, Cin, Cout, res, ups, splitk; };
    const Shape shapes[] = {{16, 64, 320, 320, 1, 0, 1}, {16, 64, 320, 320, 0, 0, 1}, {16, 32, 640, 640, 1, 0, 1}, {16, 64, 640, 320, 1, 0, 1}, {16, 64, 960, 320, 1, 0, 1},
                            {16, 32, 1280, 640, 1, 0, 1}, {16, 32, 320, 640, 0, 0, 1}, {16, 16, 1280, 1280, 1, 0, 4}, {16, 16, 2560, 1280, 1, 0, 4}, {16, 16, 640, 1280, 0, 0, 2},
                            {16, 64, 640, 640, 0, 1, 1}, {16, 32, 1280, 1280, 0, 1, 1}, {2, 64, 320, 320, 1, 0, 1}, {2, 32, 640, 640, 1, 0, 1}, {1, 16, 128, 324, 1, 0, 1}, {3, 16, 192, 40, 0, 1, 1},
                            // M = 1024 (the 8x8 level's row count, as 4 images of 16x16): how would this kernel family do there, by split
                            {4, 16, 1280, 1280, 1, 0, 4}, {4, 16, 1280, 1280, 1, 0, 5}, {4, 16, 1280, 1280, 1, 0, 10}, {4, 16, 2560, 1280, 1, 0, 8}, {4, 16, 2560, 1280, 1, 0, 10}};
    int bad = 0;
    for (const Shape& sh : shapes) {
        const int Hin = sh.ups ? sh.H / 2 : sh.H;
        const size_t M = (size_t)sh.B * sh.H * sh.H, K = 9 * (size_t)sh.Cin;
        const size_t ab = (size_t)sh.B * Hin * Hin * sh.Cin * 2, wb = (size_t)sh.Cout * K * 2, cb = M * sh.Cout * 2, sb = sh.splitk > 1 ? (size_t)sh.splitk * M * sh.Cout * 4 : 0;
        void *a, *w, *r, *c[4], *slab[4] = {nullptr, nullptr, nullptr, nullptr};
        float* bias;
        (void)hipMalloc(&a, ab); (void)hipMalloc(&w, wb); (void)hipMalloc(&r, cb); (void)hipMalloc(&bias, sh.Cout * 4);
        for (int v = 0; v < 4; ++v) { (void)hipMalloc(&c[v], cb); (void)hipMemset(c[v], 0xe0 + v, cb); if (sb) { (void)hipMalloc(&slab[v], sb); (void)hipMemset(slab[v], 0xe0 + v, sb); } }
        fill_random(a, ab, 1, 2.f); fill_random(w, wb, 2, 0.1f); fill_random(r, cb, 3, 2.f);
        std::vector<float> hb(sh.Cout);
        for (int i = 0; i < sh.Cout; ++i) hb[i] = 0.01f * (i % 37);
        (void)hipMemcpy(bias, hb.data(), sh.Cout * 4, hipMemcpyHostToDevice);
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = a; p.W = w; p.R = sh.res ? r : nullptr; p.bias = bias;
        p.M = (int)M; p.N = sh.Cout; p.K = p.Kpad = (int)K; p.lda = sh.Cin; p.ldc = p.ldr = sh.Cout;
        p.a_dt = p.c_dt = p.r_dt = DT_F16; p.taps = 9; p.Cin = sh.Cin; p.Hin = p.Win = Hin; p.Hout = p.Wout = sh.H; p.stride = 1; p.ups = sh.ups;
        p.rows_per_sample = sh.H * sh.H; p.out_scale = 1.f; p.vt_begin = 0x7fffffff; p.Nout = sh.Cout; p.splitk = sh.splitk;
        const int tiles = conv_patch_tiles(p, DT_F16);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float t[4] = {1e9f, 1e9f, 1e9f, 1e9f};
#ifdef COLD   // every launch reads operands that left the caches long ago: NB copies of input, weights and residual in turn (> 256 MB together)
        constexpr int NB = 12;
        static void *as[NB], *ws[NB], *rs[NB];
        for (int i = 0; i < NB; ++i) {
            (void)hipMalloc(&as[i], ab); (void)hipMalloc(&ws[i], wb); (void)hipMalloc(&rs[i], cb);
            (void)hipMemcpy(as[i], a, ab, hipMemcpyDeviceToDevice); (void)hipMemcpy(ws[i], w, wb, hipMemcpyDeviceToDevice); (void)hipMemcpy(rs[i], r, cb, hipMemcpyDeviceToDevice);
        }
        int turn = 0;
#endif
        auto run = [&](int v) {
            GemmParams q = p;
            q.C = c[v]; q.slab = slab[v];
#ifdef COLD
            q.A = as[turn % NB]; q.W = ws[turn % NB]; if (sh.res) q.R = rs[turn % NB];
            ++turn;
#endif
            return v == 0 ? launch_conv_patch(q, DT_F16, 0) : v == 1 ? launch_conv_patch2(q, DT_F16, 0) : v == 2 ? 0 : launch_conv_patch4(q, DT_F16, 0);
        };
        int fail = 0;
        for (int round = 0; round < 4; ++round)
            for (int v = 0; v < 4; ++v) {
                fail |= run(v);
                (void)hipEventRecord(e0, 0);
                for (int rep = 0; rep < 5; ++rep) fail |= run(v);
                (void)hipEventRecord(e1, 0);
                if (hipDeviceSynchronize() != hipSuccess) { printf("device error: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                t[v] = std::min(t[v], ms * 200.0f);
            }
        std::vector<uint16_t> h0(cb / 2), h1(cb / 2);
        (void)hipMemcpy(h0.data(), c[0], cb, hipMemcpyDeviceToHost);
        size_t diff[4] = {0, 0, 0, 0};
        double maxd = 0, maxv = 0;   // ping-pong (32x32x16 MFMAs) vs gen1: same exact products, another fp32 summation grouping
        auto h2f = [](uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (double)(float)h; };
        for (int v = 1; v < 4; ++v) {
            (void)hipMemcpy(h1.data(), c[v], cb, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < h0.size(); ++i) {
                diff[v] += h0[i] != h1[i];
                if (v == 3) { const double a = h2f(h0[i]), b = h2f(h1[i]); maxd = std::max(maxd, b != b ? 1e30 : std::abs(a - b)); maxv = std::max(maxv, std::abs(a)); }
            }
        }
        const double fl = 2.0 * M * sh.Cout * K;
        printf("B=%2d %3dx%-3d Cin=%4d Cout=%4d res=%d ups=%d splitk=%d tiles=%4d fail=%d: gen1 %7.1f us (%5.0f TF/s)  gen2 %7.1f (%5.0f)  ping-pong %7.1f (%5.0f)  4-wave %7.1f (%5.0f)  %+5.1f %% vs best of 1/2  diff gen2 %zu pp %zu w4 %zu (max %.1e of %.1f)\n",
               sh.B, sh.H, sh.H, sh.Cin, sh.Cout, sh.res, sh.ups, sh.splitk, tiles, fail, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[2], fl / t[2] / 1e6, t[3], fl / t[3] / 1e6,
               100.0 * (t[3] / std::min(t[0], t[1]) - 1.0), diff[1], diff[2], diff[3], maxd, maxv);
        fflush(stdout);
        bad += diff[3] != 0 || fail;   // (column 2, the retired ping-pong generation, is not run)
#ifdef COLD
        for (int i = 0; i < NB; ++i) { (void)hipFree(as[i]); (void)hipFree(ws[i]); (void)hipFree(rs[i]); }
#endif
        (void)hipFree(a); (void)hipFree(w); (void)hipFree(r); (void)hipFree(bias);
        for (int v = 0; v < 4; ++v) { (void)hipFree(c[v]); if (slab[v]) (void)hipFree(slab[v]); }
    }
    return bad ? 1 : 0;
}
