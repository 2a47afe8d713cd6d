// Diagnostic build of gemm_ring.hip with s_memtime accumulators (wave 0 of every block): where do a block's cycles go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPD_STAMP -Iprompt-diffusion_amd/csrc tools/micro/ring_stamp.hip -o /tmp/ring_stamp && /tmp/ring_stamp
// Read the SHARES, not the lengths: the stamps' fences forbid overlaps the product build has.
// Without -DPD_STAMP: the product kernel, launch times only.  Knock-outs (timing only, results are garbage): -DPD_KO_DMA, -DPD_KO_DSREAD,
// -DPD_KO_COALESCED (every store instruction writes 512 contiguous bytes).
#include "../../prompt-diffusion_amd/csrc/gemm_ring.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

// random fp16 operands in [-0.5, 0.5) (constant fills run 10-15 % faster than real data: the clock follows the matrix pipe's power draw)
static void fill_random(void* p, size_t bytes, unsigned seed) {
    std::vector<uint16_t> h(bytes / 2);
    unsigned x = seed;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((int)(x >> 9 & 0x3fff) - 8192) / 16384.0f; _Float16 hf = (_Float16)f; v = *reinterpret_cast<uint16_t*>(&hf); }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
}

int main() {
    struct Shape { int M, K, N, res, tile; };
    const Shape shapes[] = {{16384, 640, 640, 0, 0}, {16384, 640, 640, 0, 1}, {16384, 640, 640, 1, 1}, {4096, 1280, 1280, 1, 0}, {16384, 2560, 640, 1, 1}, {16384, 640, 1920, 0, 1}, {16384, 5120, 2560, 0, 1}, {8192, 8192, 8192, 0, 1}};
    unsigned long long* stamps;
    hipMalloc(&stamps, 256 * 8 * 8);
#ifdef PD_STAMP
    hipMemcpyToSymbol(HIP_SYMBOL(g_ring_stamps), &stamps, sizeof(stamps));
#endif
    for (const Shape& sh : shapes) {
        void *a, *w, *c, *r;
        float* bias;
        hipMalloc(&a, (size_t)sh.M * sh.K * 2); hipMalloc(&w, (size_t)sh.N * sh.K * 2); hipMalloc(&c, (size_t)sh.M * sh.N * 2); hipMalloc(&r, (size_t)sh.M * sh.N * 2);
        hipMalloc(&bias, sh.N * 4);
        fill_random(a, (size_t)sh.M * sh.K * 2, 1); fill_random(w, (size_t)sh.N * sh.K * 2, 2); fill_random(r, (size_t)sh.M * sh.N * 2, 3); hipMemset(bias, 0, sh.N * 4);
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = a; p.W = w; p.C = c; p.R = sh.res ? r : nullptr; p.bias = bias;
        p.M = sh.M; p.N = sh.N; p.K = p.Kpad = sh.K; p.lda = sh.K; p.ldc = p.ldr = sh.N;
        p.a_dt = p.c_dt = p.r_dt = DT_F16; p.taps = 1; p.Cin = sh.K; p.rows_per_sample = sh.M; p.out_scale = 1.f; p.vt_begin = sh.N; p.Nout = sh.N; p.splitk = 1;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) launch_ring_gemm(p, DT_F16, sh.tile, 0);
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep) launch_ring_gemm(p, DT_F16, sh.tile, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
#ifndef PD_STAMP
        printf("M=%d K=%d N=%d res=%d tile=%d: %.1f us per launch (%.0f TF/s)\n", sh.M, sh.K, sh.N, sh.res, sh.tile, ms * 100.0, 2.0 * sh.M * sh.K * sh.N / (ms * 1e-4) / 1e12);
        continue;
#endif
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        auto med = [&](int slot, bool life) {
            std::vector<double> d;
            for (int b = 0; b < 256; ++b) d.push_back(life ? (double)(h[b * 8 + 1] - h[b * 8]) : (double)h[b * 8 + slot]);
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
        unsigned long long first = ~0ull, last = 0;
        for (int b = 0; b < 256; ++b) { first = std::min(first, h[b * 8]); last = std::max(last, h[b * 8 + 1]); }
        const double life = med(0, true);
        printf("M=%d K=%d N=%d res=%d tile=%d: %.1f us per launch (stamped build); tiles per block %llu; s_memtime ticks: launch span %llu, block life (median) %.0f\n", sh.M, sh.K, sh.N,
               sh.res, sh.tile, ms * 100.0, h[7], last - first, life);
        const char* names[] = {"", "", "DMA wait (vmcnt)", "barrier", "DMA issue (+ next tile's addresses)", "ds_read + MFMA", "epilogue"};
        double sum = 0;
        for (int s = 2; s <= 6; ++s) sum += med(s, false);
        for (int s = 2; s <= 6; ++s) printf("    %-38s %8.0f ticks %5.1f %% of life\n", names[s], med(s, false), 100.0 * med(s, false) / life);
        printf("    %-38s %8.0f ticks %5.1f %% of life\n", "prologue + unaccounted", life - sum, 100.0 * (life - sum) / life);
        hipFree(a); hipFree(w); hipFree(c); hipFree(r); hipFree(bias);
    }
    return 0;
}
