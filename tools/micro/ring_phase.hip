// Are the short-reduction GEMMs limited by every CU reaching its store burst at the same moment?  One [M, K] x [N, K]^T launch against the same
// work cut into 2 / 4 row slices launched back to back on one stream and side by side on 2 / 4 streams (product build of gemm_ring.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iprompt-diffusion_amd/csrc tools/micro/ring_phase.hip -o /tmp/ring_phase && /tmp/ring_phase
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
    const Shape shapes[] = {{16384, 640, 640, 0, 1}, {16384, 640, 640, 1, 1}, {16384, 640, 640, 1, 0}, {4096, 1280, 1280, 1, 0}, {16384, 2560, 640, 1, 1}, {16384, 640, 1920, 0, 1}, {16384, 640, 5120, 0, 1}, {65536, 320, 320, 1, 1}};
    hipStream_t st[4];
    hipEvent_t ev[4], e0, e1;
    for (int i = 0; i < 4; ++i) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
    hipEventCreate(&e0); hipEventCreate(&e1);
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
        // mode: slices (1, 2, 4) x streams (1 = back to back on st[0], else one stream per slice); 12 chained rounds per timing, best of 5
        auto run = [&](int slices, bool par) {
            float best = 1e9f;
            for (int t = 0; t < 6; ++t) {
                hipEventRecord(e0, st[0]);
                for (int i = 1; i < slices && par; ++i) hipStreamWaitEvent(st[i], e0, 0);
                for (int rep = 0; rep < 12; ++rep)
                    for (int s = 0; s < slices; ++s) {
                        GemmParams q = p;
                        const int rows = sh.M / slices;
                        q.M = rows; q.rows_per_sample = rows;
                        q.A = (char*)a + (size_t)s * rows * sh.K * 2; q.C = (char*)c + (size_t)s * rows * sh.N * 2;
                        if (sh.res) q.R = (char*)r + (size_t)s * rows * sh.N * 2;
                        if (launch_ring_gemm(q, DT_F16, sh.tile, par ? st[s] : st[0])) { printf("launch refused\n"); exit(1); }
                    }
                for (int i = 1; i < slices && par; ++i) { hipEventRecord(ev[i], st[i]); hipStreamWaitEvent(st[0], ev[i], 0); }
                hipEventRecord(e1, st[0]);
                hipStreamSynchronize(st[0]);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (t) best = std::min(best, ms * 1000.f / 12.f);
            }
            return best;
        };
        const float t1 = run(1, false), t2s = run(2, false), t2p = run(2, true), t4s = run(4, false), t4p = run(4, true);
        printf("M=%5d K=%4d N=%4d res=%d tile=%d: whole %6.1f us | 2 slices: in turn %6.1f, side by side %6.1f | 4 slices: in turn %6.1f, side by side %6.1f\n", sh.M, sh.K, sh.N, sh.res,
               sh.tile, t1, t2s, t2p, t4s, t4p);
        fflush(stdout);
        hipFree(a); hipFree(w); hipFree(c); hipFree(r); hipFree(bias);
    }
    return 0;
}
