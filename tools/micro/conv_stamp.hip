// Diagnostic build of conv_patch.hip with s_memtime accumulators (wave 0 of every block): where do a block's cycles go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPD_STAMP -w -Iprompt-diffusion_amd/csrc tools/micro/conv_stamp.hip -o /tmp/conv_stamp && /tmp/conv_stamp
// Read the SHARES, not the lengths: the stamps' fences forbid overlaps the product build has.
// Without -DPD_STAMP the kernel is the product build and only the launch time is printed.
#include "../../prompt-diffusion_amd/csrc/conv_patch.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

int launch_splitk_finalize(const GemmParams&, hipStream_t) { return 0; }   // (gemm.hip's; not linked here, never reached: no split-K below)

int main() {
    struct Shape { int B, H, Cin, Cout, res; };
    const Shape shapes[] = {{16, 64, 320, 320, 1}, {16, 64, 320, 320, 0}, {16, 32, 640, 640, 1}, {16, 64, 640, 320, 1}, {16, 16, 1280, 1280, 1}};
    unsigned long long* stamps;
    hipMalloc(&stamps, 4096 * 10 * 8);
#ifdef PD_STAMP
    hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &stamps, sizeof(stamps));
#endif
    for (const Shape& sh : shapes) {
        const long long M = (long long)sh.B * sh.H * sh.H;
        const int K = 9 * sh.Cin;
        void *a, *w, *c, *r;
        float* bias;
        hipMalloc(&a, M * sh.Cin * 2); hipMalloc(&w, (size_t)sh.Cout * K * 2); hipMalloc(&c, M * sh.Cout * 2); hipMalloc(&r, M * sh.Cout * 2); hipMalloc(&bias, sh.Cout * 4);
        hipMemset(a, 0x11, M * sh.Cin * 2); hipMemset(w, 0x12, (size_t)sh.Cout * K * 2); hipMemset(r, 0x13, M * sh.Cout * 2); hipMemset(bias, 0, sh.Cout * 4);
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = a; p.W = w; p.C = c; p.R = sh.res ? r : nullptr; p.bias = bias;
        p.M = (int)M; p.N = sh.Cout; p.K = p.Kpad = K; p.lda = sh.Cin; p.ldc = p.ldr = sh.Cout;
        p.a_dt = p.c_dt = p.r_dt = DT_F16; p.taps = 9; p.Cin = sh.Cin; p.Hin = p.Win = p.Hout = p.Wout = sh.H; p.stride = 1;
        p.rows_per_sample = sh.H * sh.H; p.out_scale = 1.f; p.vt_begin = sh.Cout; p.Nout = sh.Cout; p.splitk = 1;
        const int blocks = conv_patch_tiles(p, DT_F16);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) launch_conv_patch(p, DT_F16, 0);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep) launch_conv_patch(p, DT_F16, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 10);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        auto med = [&](int slot) {
            std::vector<double> d;
            for (int b = 0; b < blocks; ++b) d.push_back(slot < 0 ? (double)(h[b * 10 + 1] - h[b * 10]) : (double)h[b * 10 + slot]);
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
#ifndef PD_STAMP
        printf("conv3x3 B=%d %dx%d Cin=%d Cout=%d res=%d: %.1f us per launch (%.0f TF/s)\n", sh.B, sh.H, sh.H, sh.Cin, sh.Cout, sh.res, ms * 100.0, 2.0 * M * sh.Cout * K / (ms * 1e-4) / 1e12);
        continue;
#endif
        const double life = med(-1), units = (double)h[8];
        const double flops = 2.0 * M * sh.Cout * K;
        printf("conv3x3 B=%d %dx%d Cin=%d Cout=%d res=%d: %.1f us per launch (stamped build, %.0f TF/s), %d blocks, %.0f units per block; block life (median) %.0f ticks = %.0f per unit\n", sh.B, sh.H,
               sh.H, sh.Cin, sh.Cout, sh.res, ms * 100.0, flops / (ms * 1e-4) / 1e12, blocks, units, life, life / units);
        const char* names[] = {"", "", "prologue", "unit top: requests + piece store", "ds_read + MFMA", "weight wait + ds_write", "barrier", "epilogue"};
        for (int s = 2; s <= 7; ++s) printf("    %-36s %8.0f ticks %5.1f %% of life  (%5.0f per unit)\n", names[s], med(s), 100.0 * med(s) / life, med(s) / units);
        hipFree(a); hipFree(w); hipFree(c); hipFree(r); hipFree(bias);
    }
    return 0;
}
