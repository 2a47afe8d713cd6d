// Diagnostic build of conv_patch4.hip with s_memtime accumulators (wave 0 of every block): the counted wait, the barrier, the epilogue,
// the in-kernel clock.    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPD_STAMP -w -Iprompt-diffusion_amd/csrc tools/micro/conv_w4_stamp.hip -o tools/micro/bin/conv_w4_stamp
#include "../../prompt-diffusion_amd/csrc/conv_patch4.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

int launch_splitk_finalize(const GemmParams&, hipStream_t) { return 0; }   // (gemm.hip's; never reached: no split-K below)

static void fill_random(void* p, size_t bytes, unsigned seed) {
    std::vector<uint16_t> h(bytes / 2);
    unsigned x = seed;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((int)(x >> 9 & 0x3fff) - 8192) / 16384.0f; _Float16 hf = (_Float16)f; v = *reinterpret_cast<uint16_t*>(&hf); }
    (void)hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
}

int main() {
    struct Shape { int B, H, Cin, Cout, res; };
    const Shape shapes[] = {{16, 64, 320, 320, 0}, {16, 32, 640, 640, 1}, {16, 64, 640, 320, 1}};
    unsigned long long* stamps;
    (void)hipMalloc(&stamps, 4096 * 8 * 8);
#ifdef PD_STAMP
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w4_stamps), &stamps, sizeof(stamps));
#endif
    for (const Shape& sh : shapes) {
        const long long M = (long long)sh.B * sh.H * sh.H;
        const int K = 9 * sh.Cin;
        void *a, *w, *c, *r;
        float* bias;
        (void)hipMalloc(&a, M * sh.Cin * 2); (void)hipMalloc(&w, (size_t)sh.Cout * K * 2); (void)hipMalloc(&c, M * sh.Cout * 2); (void)hipMalloc(&r, M * sh.Cout * 2); (void)hipMalloc(&bias, sh.Cout * 4);
        fill_random(a, M * sh.Cin * 2, 1); fill_random(w, (size_t)sh.Cout * K * 2, 2); fill_random(r, M * sh.Cout * 2, 3);
        (void)hipMemset(bias, 0, sh.Cout * 4);
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = a; p.W = w; p.C = c; p.R = sh.res ? r : nullptr; p.bias = bias;
        p.M = (int)M; p.N = sh.Cout; p.K = p.Kpad = K; p.lda = sh.Cin; p.ldc = p.ldr = sh.Cout;
        p.a_dt = p.c_dt = p.r_dt = DT_F16; p.taps = 9; p.Cin = sh.Cin; p.Hin = p.Win = p.Hout = p.Wout = sh.H; p.stride = 1;
        p.rows_per_sample = sh.H * sh.H; p.out_scale = 1.f; p.vt_begin = 0x7fffffff; p.Nout = sh.Cout; p.splitk = 1;
        const int blocks = (int)(M / 256) * ((sh.Cout + 159) / 160);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) launch_conv_patch4(p, DT_F16, 0);
        (void)hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep) launch_conv_patch4(p, DT_F16, 0);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * M * sh.Cout * K;
        printf("conv3x3 B=%d %dx%d Cin=%d Cout=%d res=%d: %.1f us per launch (%.0f TF/s), %d blocks\n", sh.B, sh.H, sh.H, sh.Cin, sh.Cout, sh.res, ms * 100.0, fl / (ms * 1e-4) / 1e12, blocks);
#ifdef PD_STAMP
        std::vector<unsigned long long> h((size_t)blocks * 8);
        (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        auto med = [&](int slot) {
            std::vector<double> d;
            for (int b = 0; b < blocks; ++b) { const unsigned long long* o = &h[(size_t)b * 8]; d.push_back(slot < 0 ? (double)(o[1] - o[0]) : (double)o[slot]); }
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
        const double life = med(-1), U = med(5);
        printf("  block life (median) %.0f ticks = %.0f per unit (%.0f units); in-kernel clock %.0f MHz\n", life, life / U, U, life / med(6) * 100.0);
        printf("    counted vmcnt wait %8.0f ticks %5.1f %% (%5.0f per unit)\n    barrier            %8.0f ticks %5.1f %% (%5.0f per unit)\n    epilogue           %8.0f ticks %5.1f %% (%5.0f per unit)\n    rest (MFMA stream) %8.0f ticks %5.1f %% (%5.0f per unit)\n",
               med(2), 100 * med(2) / life, med(2) / U, med(3), 100 * med(3) / life, med(3) / U, med(4), 100 * med(4) / life, med(4) / U, life - med(2) - med(3) - med(4),
               100 * (life - med(2) - med(3) - med(4)) / life, (life - med(2) - med(3) - med(4)) / U);
#endif
        fflush(stdout);
        (void)hipFree(a); (void)hipFree(w); (void)hipFree(c); (void)hipFree(r); (void)hipFree(bias);
    }
    return 0;
}
