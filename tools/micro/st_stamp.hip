// Diagnostic build of the fused transformer kernels with s_memtime stamps at their phase boundaries (wave 0 of every workgroup):
// where do the cycles of a workgroup go?  Random operands, SD1.5's 64 x 64 level: M = 65536 tokens, forward batch 16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPD_STAMP -Iprompt-diffusion_amd/csrc tools/micro/st_stamp.hip -o /tmp/st_stamp && /tmp/st_stamp
// Read the SHARES, not the lengths: the stamps' fences forbid overlaps the product build has.
#include "../../prompt-diffusion_amd/csrc/st_tail.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

static void fill(void* p, size_t bytes, unsigned seed) {
    std::vector<uint16_t> h(bytes / 2);
    unsigned x = seed;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; const float f = ((int)(x >> 9 & 0x3fff) - 8192) / 16384.0f; _Float16 hf = (_Float16)f; v = *reinterpret_cast<uint16_t*>(&hf); }
    hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
}
static void report(const char* name, unsigned long long* dstamps, int blocks, int nslot, const char* const* labels) {
    std::vector<unsigned long long> h((size_t)blocks * 16);
    hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%s: median cycles per phase over %d workgroups (wave 0)\n", name, blocks);
    double total = 0;
    std::vector<double> med(nslot);
    for (int s = 1; s < nslot; ++s) {
        std::vector<double> d;
        for (int b = 0; b < blocks; ++b) d.push_back((double)(h[(size_t)b * 16 + s] - h[(size_t)b * 16 + s - 1]));
        std::sort(d.begin(), d.end());
        med[s] = d[d.size() / 2];
        total += med[s];
    }
    for (int s = 1; s < nslot; ++s) printf("  %-44s %9.0f cycles  %5.1f %%\n", labels[s], med[s], 100.0 * med[s] / total);
    // spread of the start stamps: how far apart do the workgroups of the launch begin?
    std::vector<unsigned long long> st;
    for (int b = 0; b < blocks; ++b) st.push_back(h[(size_t)b * 16]);
    std::sort(st.begin(), st.end());
    {
        std::vector<double> a, b, c;
        for (int bk = 0; bk < blocks; ++bk) { a.push_back((double)h[(size_t)bk * 16 + 8]); b.push_back((double)h[(size_t)bk * 16 + 9]); c.push_back((double)h[(size_t)bk * 16 + 10]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
        if (c[c.size() / 2] > 0)
            printf("  of which, summed over the ring steps: DMA wait (vmcnt) %.0f, barrier %.0f, DMA issue %.0f cycles\n", a[a.size() / 2], b[b.size() / 2], c[c.size() / 2]);
    }
    printf("  total %.0f cycles per workgroup; start stamps span %.0f cycles (first to last workgroup; 2 rounds of 256)\n", total, (double)(st.back() - st.front()));
}

int main() {
    const int B = 16, N = 4096, C = 320;
    const long long M = (long long)B * N;
    void *x, *h, *qk, *vt, *att, *out, *wf, *wt, *kv;
    float *coef, *vf, *vtl;
    unsigned long long* stamps;
    hipMalloc(&x, M * C * 2); hipMalloc(&h, M * C * 2); hipMalloc(&qk, M * C * 4); hipMalloc(&vt, M * C * 2); hipMalloc(&att, M * C * 2); hipMalloc(&out, M * C * 2);
    hipMalloc(&wf, st_front_weight_bytes()); hipMalloc(&wt, st_tail_weight_bytes()); hipMalloc(&kv, st_tail_kv_bytes(B));
    hipMalloc(&coef, B * C * 2 * 4); hipMalloc(&vf, st_front_vec_floats() * 4); hipMalloc(&vtl, st_tail_vec_floats() * 4);
    hipMalloc(&stamps, 512 * 16 * 8);
    fill(x, M * C * 2, 1); fill(att, M * C * 2, 2); fill(wf, st_front_weight_bytes(), 3); fill(wt, st_tail_weight_bytes(), 4); fill(kv, st_tail_kv_bytes(B), 5);
    std::vector<float> ones(B * C * 2, 0.5f), zeros(8192, 0.01f);
    hipMemcpy(coef, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(vf, zeros.data(), st_front_vec_floats() * 4, hipMemcpyHostToDevice);
    hipMemcpy(vtl, zeros.data(), st_tail_vec_floats() * 4, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &stamps, sizeof(stamps));
    static const char* fl[] = {"", "prologue (ring fill, x rows, coefficients)", "GroupNorm apply + proj_in (200 MFMAs)", "h store + norm1", "to_q/k/v (600 MFMAs) + q/k/V^T stores", "drain"};
    static const char* tl[] = {"", "prologue (ring fill, att / h rows)", "attn1.to_out (200 MFMAs)", "norm2 + cross-attention (648 MFMAs)", "norm3 + feed-forward (2400 MFMAs)", "h3 frags, x_in rows, proj_out (200 MFMAs)", "drain + out store"};
    for (int rep = 0; rep < 3; ++rep) {
        launch_st_front(x, coef, h, qk, vt, wf, vf, M, N, N, DT_F16, 0);
        hipDeviceSynchronize();
    }
    report("st_front_kernel", stamps, 512, 6, fl);
    for (int rep = 0; rep < 3; ++rep) {
        launch_st_tail(att, h, x, out, wt, vtl, kv, M, N, 77, DT_F16, 0.158f, DT_F16, 0);
        hipDeviceSynchronize();
    }
    report("st_tail_kernel", stamps, 512, 7, tl);
    return 0;
}
