// Launch timing of conv_patch.hip (first-generation patch conv) on random / constant, warm / cold operands.  Its s_memtime stamps were
// retired in round 4 (the stamped numbers are kept in profiles/r03_conv_stamp.txt; git history has the stamped kernel); -DPD_STAMP no longer
// has an effect here.  The stamped generation is conv_patch4.hip: tools/micro/conv_w4_stamp.hip.  Original header:
// Diagnostic build of conv_patch.hip with s_memtime accumulators (wave 0 of every block): where do a block's cycles go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPD_STAMP -w -Iprompt-diffusion_amd/csrc tools/micro/conv_stamp.hip -o /tmp/conv_stamp && /tmp/conv_stamp
// Read the SHARES, not the lengths: the stamps' fences forbid overlaps the product build has.
// Without -DPD_STAMP the kernel is the product build and only the launch time is printed.  -DCOLD: every launch reads operands that
// left the caches long ago (10 operand sets in turn); -DCONST_DATA: hipMemset operands instead of random ones (10-15 % faster: the
// matrix pipe's power draw, and with it the clock, depends on the data -- never benchmark a kernel on constant fills).
#ifdef CONV_SRC_OLD   // A/B against another version of the kernel: put it at tools/micro/conv_patch_old.hip (not committed)
#include "conv_patch_old.hip"
#else
#include "../../prompt-diffusion_amd/csrc/conv_patch.hip"
#endif
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

int launch_splitk_finalize(const GemmParams&, hipStream_t) { return 0; }   // (gemm.hip's; not linked here, never reached: no split-K below)

// random fp16 operands in [-0.5, 0.5): the matrix pipe's power draw -- and with it the clock the chip settles at -- depends on the data;
// constant fills (hipMemset) run 10-15 % faster than anything the network ever sees
static void fill_random(void* p, size_t bytes, unsigned seed) {
    std::vector<uint16_t> h(bytes / 2);
    unsigned x = seed;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((int)(x >> 9 & 0x3fff) - 8192) / 16384.0f; _Float16 hf = (_Float16)f; v = *reinterpret_cast<uint16_t*>(&hf); }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
}

int main() {
    struct Shape { int B, H, Cin, Cout, res; };
    const Shape shapes[] = {{16, 64, 320, 320, 1}, {16, 64, 320, 320, 0}, {16, 32, 640, 640, 1}, {16, 64, 640, 320, 1}, {16, 16, 1280, 1280, 1}};
    unsigned long long* stamps;
    hipMalloc(&stamps, 4096 * 18 * 8);
    for (const Shape& sh : shapes) {
        const long long M = (long long)sh.B * sh.H * sh.H;
        const int K = 9 * sh.Cin;
        void *a, *w, *c, *r;
        float* bias;
        hipMalloc(&a, M * sh.Cin * 2); hipMalloc(&w, (size_t)sh.Cout * K * 2); hipMalloc(&c, M * sh.Cout * 2); hipMalloc(&r, M * sh.Cout * 2); hipMalloc(&bias, sh.Cout * 4);
#ifdef CONST_DATA
        hipMemset(a, 0x11, M * sh.Cin * 2); hipMemset(w, 0x12, (size_t)sh.Cout * K * 2); hipMemset(r, 0x13, M * sh.Cout * 2);
#else
        fill_random(a, M * sh.Cin * 2, 1); fill_random(w, (size_t)sh.Cout * K * 2, 2); fill_random(r, M * sh.Cout * 2, 3);
#endif
        hipMemset(bias, 0, sh.Cout * 4);
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = a; p.W = w; p.C = c; p.R = sh.res ? r : nullptr; p.bias = bias;
        p.M = (int)M; p.N = sh.Cout; p.K = p.Kpad = K; p.lda = sh.Cin; p.ldc = p.ldr = sh.Cout;
        p.a_dt = p.c_dt = p.r_dt = DT_F16; p.taps = 9; p.Cin = sh.Cin; p.Hin = p.Win = p.Hout = p.Wout = sh.H; p.stride = 1;
        p.rows_per_sample = sh.H * sh.H; p.out_scale = 1.f; p.vt_begin = sh.Cout; p.Nout = sh.Cout; p.splitk = 1;
        const int blocks = conv_patch_tiles(p, DT_F16);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef COLD   // every launch reads inputs that left the caches long ago: NB operand sets (> the 256 MB Infinity Cache together) in turn
        constexpr int NB = 10;
        void *as[NB], *rs[NB], *cs[NB];
        for (int i = 0; i < NB; ++i) { hipMalloc(&as[i], M * sh.Cin * 2); hipMalloc(&rs[i], M * sh.Cout * 2); hipMalloc(&cs[i], M * sh.Cout * 2); hipMemcpy(as[i], a, M * sh.Cin * 2, hipMemcpyDeviceToDevice); hipMemcpy(rs[i], r, M * sh.Cout * 2, hipMemcpyDeviceToDevice); }
        auto launch_i = [&](int i) { GemmParams q = p; q.A = as[i % NB]; q.C = cs[i % NB]; if (q.R) q.R = rs[i % NB]; launch_conv_patch(q, DT_F16, 0); };
        for (int rep = 0; rep < NB; ++rep) launch_i(rep);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep) launch_i(rep);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        for (int i = 0; i < NB; ++i) { hipFree(as[i]); hipFree(rs[i]); hipFree(cs[i]); }
#else
        for (int rep = 0; rep < 3; ++rep) launch_conv_patch(p, DT_F16, 0);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep) launch_conv_patch(p, DT_F16, 0);
        hipEventRecord(e1, 0);
#endif
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 18);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        auto med = [&](int slot) {
            std::vector<double> d;
            for (int b = 0; b < blocks; ++b) d.push_back(slot < 0 ? (double)(h[b * 10 + 1] - h[b * 10]) : (double)h[b * 10 + slot]);
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
#if 1
        printf("conv3x3 B=%d %dx%d Cin=%d Cout=%d res=%d: %.1f us per launch (%.0f TF/s)\n", sh.B, sh.H, sh.H, sh.Cin, sh.Cout, sh.res, ms * 100.0, 2.0 * M * sh.Cout * K / (ms * 1e-4) / 1e12);
        continue;
#endif
        const double life = med(-1), units = (double)h[8];
        const double flops = 2.0 * M * sh.Cout * K;
        printf("conv3x3 B=%d %dx%d Cin=%d Cout=%d res=%d: %.1f us per launch (stamped build, %.0f TF/s), %d blocks, %.0f units per block; block life (median) %.0f ticks = %.0f per unit\n", sh.B, sh.H,
               sh.H, sh.Cin, sh.Cout, sh.res, ms * 100.0, flops / (ms * 1e-4) / 1e12, blocks, units, life, life / units);
        const char* names[] = {"", "", "prologue", "unit top: requests + piece store", "ds_read + MFMA", "weight wait + ds_write", "barrier", "epilogue"};
        {   // per wave: cycles between a barrier's release and the wave's arrival at the next one, per unit (median over blocks)
            printf("    per-wave busy cycles per unit (waves 0-7):");
            for (int w = 0; w < 8; ++w) {
                std::vector<double> d;
                for (int b = 0; b < blocks; ++b) d.push_back((double)h[(size_t)blocks * 10 + (size_t)b * 8 + w]);
                std::sort(d.begin(), d.end());
                printf(" %5.0f", d[d.size() / 2] / units);
            }
            printf("\n");
        }
        for (int s = 2; s <= 7; ++s) printf("    %-36s %8.0f ticks %5.1f %% of life  (%5.0f per unit)\n", names[s], med(s), 100.0 * med(s) / life, med(s) / units);
        hipFree(a); hipFree(w); hipFree(c); hipFree(r); hipFree(bias);
    }
    return 0;
}
