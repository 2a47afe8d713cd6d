// Per-operator parity hooks (declared in include/pdengine_ops.h): run ONE kernel of the hot path on
// host arrays in the reference's own layouts, so tests/ can compare each HIP kernel with the oracle.
// Not used by the sampling path.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pdengine_ops.h"
#include "engine.h"

namespace {
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) p = nullptr; else { hipMemset(p, 0, bytes ? bytes : 16); hipDeviceSynchronize(); } }
    ~DevBuf() { if (p) hipFree(p); }
};
struct TempMat {
    pd_engine* e;
    size_t owned0;
    explicit TempMat(pd_engine* e_) : e(e_), owned0(e_->owned.size()) {}
    ~TempMat() {
        while (e->owned.size() > owned0) { hipFree(e->owned.back()); e->owned.pop_back(); }
    }
};
int to_dev_nhwc(pd_engine* e, const float* host, void* dst, int dt, int B, int C, int H, int W, int Cpad) {
    DevBuf tmp((size_t)B * C * H * W * 4);
    if (!tmp.p) return 1;
    HIP_OK(hipMemcpy(tmp.p, host, (size_t)B * C * H * W * 4, hipMemcpyHostToDevice));
    if (launch_nchw_to_nhwc(reinterpret_cast<const float*>(tmp.p), dst, dt, B, C, H, W, Cpad, e->stream)) return 1;
    HIP_OK(hipStreamSynchronize(e->stream));
    return 0;
}
int from_dev_nhwc(pd_engine* e, const void* src, int dt, float* host, int B, int C, int H, int W, int Cpad) {
    DevBuf tmp((size_t)B * C * H * W * 4);
    if (!tmp.p) return 1;
    if (launch_nhwc_to_nchw(src, dt, reinterpret_cast<float*>(tmp.p), B, C, H, W, Cpad, 1.f, e->stream)) return 1;
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy(host, tmp.p, (size_t)B * C * H * W * 4, hipMemcpyDeviceToHost));
    return 0;
}
int round_up(int x, int m) { return (x + m - 1) / m * m; }
}  // namespace

extern "C" {

// y = conv2d(x, w, b, stride, padding = k/2) [+ SiLU] [* scale] [+ residual]; x NCHW fp32, w OIHW.
int pd_op_conv2d(pd_engine* e, const float* x, const float* w, const float* bias, const float* residual, int B, int Cin, int H,
                 int W, int Cout, int k, int stride, int upsample, int act_silu, float scale, int stream_out, float* y) {
    if (!e || !x || !w || !y) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    TempMat guard(e);
    ConvW c;
    c.cin = Cin; c.cout = Cout; c.k = k; c.stride = stride;
    e->make_mat(c.m, Cout, Cin * k * k, k * k, Cin, true);
    if (!c.m.w) { pd_set_error("allocation failed"); return 1; }
    PD_TRY(e->upload_rows(c.m, 0, w, Cout, true));
    if (bias) PD_TRY(e->upload_vec(c.m.bias, bias, Cout, false, 0));
    const int Hv = H << upsample, Wv = W << upsample;
    const int Ho = stride == 2 ? (Hv + 1) / 2 : Hv, Wo = stride == 2 ? (Wv + 1) / 2 : Wv;
    const int cpad = c.m.cin_pad, odt = stream_out ? e->S : e->T, copad = round_up(Cout, 4);
    DevBuf in((size_t)B * H * W * cpad * dt_size(e->T)), out((size_t)B * Ho * Wo * copad * dt_size(odt)),
        res((size_t)B * Ho * Wo * copad * dt_size(e->S));
    if (!in.p || !out.p || !res.p) { pd_set_error("allocation failed"); return 1; }
    PD_TRY(to_dev_nhwc(e, x, in.p, e->T, B, Cin, H, W, cpad));
    Act a, o, r;
    a.p = in.p; a.B = B; a.H = H; a.W = W; a.C = cpad; a.dt = e->T;
    o.p = out.p; o.B = B; o.H = Ho; o.W = Wo; o.C = copad; o.dt = odt;
    r = o; r.p = res.p; r.dt = e->S;
    if (residual) PD_TRY(to_dev_nhwc(e, residual, res.p, e->S, B, Cout, Ho, Wo, copad));
    PD_TRY(e->conv(c, a, o, act_silu, scale, residual ? &r : nullptr, nullptr, 0, upsample));
    return from_dev_nhwc(e, out.p, odt, y, B, Cout, Ho, Wo, copad);
}

// y[M,N] = act(x[M,K] @ w[N,K]^T + b) ; geglu: w [2*N, K] -> y[M,N] = (x w_a + b_a) * gelu(x w_g + b_g)
int pd_op_linear(pd_engine* e, const float* x, const float* w, const float* bias, int M, int K, int N, int geglu, int a_silu,
                 float* y) {
    if (!e || !x || !w || !y) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    TempMat guard(e);
    WMat m;
    const int rows = geglu ? 2 * N : N;
    e->make_mat(m, rows, K, 1, K, true, geglu != 0);
    if (!m.w) { pd_set_error("allocation failed"); return 1; }
    if (K % 8) { pd_set_error("K must be a multiple of 8"); return 1; }
    PD_TRY(e->upload_rows(m, 0, w, rows, false));
    if (bias) PD_TRY(e->upload_vec(m.bias, bias, rows, geglu != 0, N));
    const int adt = a_silu ? DT_F32 : e->T;
    DevBuf in((size_t)M * K * 4), out((size_t)M * round_up(N, 4) * 4);
    if (!in.p || !out.p) { pd_set_error("allocation failed"); return 1; }
    {
        DevBuf tmp((size_t)M * K * 4);
        HIP_OK(hipMemcpy(tmp.p, x, (size_t)M * K * 4, hipMemcpyHostToDevice));
        if (launch_cast_rows(reinterpret_cast<const float*>(tmp.p), in.p, adt, M, K, K, e->stream)) return 1;
        HIP_OK(hipStreamSynchronize(e->stream));
    }
    Act a, o;
    a.p = in.p; a.B = M; a.H = 1; a.W = 1; a.C = K; a.dt = adt;
    o.p = out.p; o.B = M; o.H = 1; o.W = 1; o.C = round_up(N, 4); o.dt = DT_F32;
    PD_TRY(e->gemm(m, a, o, 1, 0, 0, 1.f, nullptr, nullptr, 0, a_silu != 0, nullptr, 0, 0));
    HIP_OK(hipStreamSynchronize(e->stream));
    std::vector<float> host((size_t)M * o.C);
    HIP_OK(hipMemcpy(host.data(), out.p, host.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < M; ++i) memcpy(y + (size_t)i * N, host.data() + (size_t)i * o.C, (size_t)N * 4);
    return 0;
}

// y[M,N] = (e4m3(x / sx) @ e4m3(w / sw)^T) * sx * sw + b with per-row scales (max |row| / 448): the PREC_FP8 linear layer of
// the SD3 path (option sd3_fp8) on host fp32 arrays; act 4: tanh-GELU epilogue.  2-byte engine modes only.
int pd_op_linear_fp8(pd_engine* e, const float* x, const float* w, const float* bias, int M, int K, int N, int act, float* y) {
    if (!e || !x || !w || !y) { pd_set_error("null argument"); return 1; }
    if (e->f32) { pd_set_error("pd_op_linear_fp8: 2-byte engine modes only"); return 1; }
    if (K % 8 || (act != 0 && act != 4)) { pd_set_error("pd_op_linear_fp8: K must be a multiple of 8, act 0 or 4"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    TempMat guard(e);
    WMat m;
    e->make_mat(m, N, K, 1, K, true);
    if (!m.w) { pd_set_error("allocation failed"); return 1; }
    if (bias) PD_TRY(e->upload_vec(m.bias, bias, N, false, 0));
    m.Kpad8 = round_up(K, 128);
    DevBuf w32((size_t)N * K * 4), w8((size_t)m.N * m.Kpad8), ws((size_t)(m.N + 4) * 4), x32((size_t)M * K * 4), x8((size_t)M * m.Kpad8),
        xs((size_t)M * 4 + 16), out((size_t)M * round_up(N, 4) * 4);
    if (!w32.p || !w8.p || !ws.p || !x32.p || !x8.p || !xs.p || !out.p) { pd_set_error("allocation failed"); return 1; }
    HIP_OK(hipMemset(w8.p, 0, (size_t)m.N * m.Kpad8));
    HIP_OK(hipMemcpy(w32.p, w, (size_t)N * K * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(x32.p, x, (size_t)M * K * 4, hipMemcpyHostToDevice));
    HIP_OK(hipDeviceSynchronize());
    if (launch_quant_rows(w32.p, DT_F32, K, w8.p, m.Kpad8, reinterpret_cast<float*>(ws.p), N, K, e->stream) ||
        launch_quant_rows(x32.p, DT_F32, K, x8.p, m.Kpad8, reinterpret_cast<float*>(xs.p), M, K, e->stream)) {
        pd_set_error("quantisation launch failed");
        return 1;
    }
    m.w8 = w8.p; m.wscale = reinterpret_cast<float*>(ws.p);
    m.cin_pad = m.Kpad8;   // the activation rows are padded to the fp8 K step
    Act a, o;
    a.p = x8.p; a.B = M; a.H = 1; a.W = 1; a.C = m.Kpad8; a.dt = DT_FP8;
    o.p = out.p; o.B = M; o.H = 1; o.W = 1; o.C = round_up(N, 4); o.dt = DT_F32;
    e->gx.a_scale = reinterpret_cast<float*>(xs.p);
    PD_TRY(e->gemm(m, a, o, 1, 0, act, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    HIP_OK(hipStreamSynchronize(e->stream));
    std::vector<float> host((size_t)M * o.C);
    HIP_OK(hipMemcpy(host.data(), out.p, host.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < M; ++i) memcpy(y + (size_t)i * N, host.data() + (size_t)i * o.C, (size_t)N * 4);
    return 0;
}

int pd_op_groupnorm(pd_engine* e, const float* x, const float* gamma, const float* beta, int B, int C, int H, int W, float eps,
                    int silu, float* y) {
    if (!e || !x || !gamma || !beta || !y) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    DevBuf in((size_t)B * H * W * C * dt_size(e->S)), out((size_t)B * H * W * C * dt_size(e->T)), g((size_t)C * 4 + 16), b((size_t)C * 4 + 16);
    if (!in.p || !out.p || !g.p || !b.p) { pd_set_error("allocation failed"); return 1; }
    HIP_OK(hipMemcpy(g.p, gamma, (size_t)C * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(b.p, beta, (size_t)C * 4, hipMemcpyHostToDevice));
    PD_TRY(to_dev_nhwc(e, x, in.p, e->S, B, C, H, W, C));
    Act a, o;
    a.p = in.p; a.B = B; a.H = H; a.W = W; a.C = C; a.dt = e->S;
    o = a; o.p = out.p; o.dt = e->T;
    PD_TRY(e->groupnorm(a, o, reinterpret_cast<float*>(g.p), reinterpret_cast<float*>(b.p), eps, silu != 0));
    return from_dev_nhwc(e, out.p, e->T, y, B, C, H, W, C);
}

int pd_op_layernorm(pd_engine* e, const float* x, const float* gamma, const float* beta, int rows, int C, float* y) {
    if (!e || !x || !gamma || !beta || !y) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    DevBuf tmp((size_t)rows * C * 4), in((size_t)rows * C * dt_size(e->S)), out((size_t)rows * C * 4), g((size_t)C * 4 + 16), b((size_t)C * 4 + 16);
    if (!tmp.p || !in.p || !out.p || !g.p || !b.p) { pd_set_error("allocation failed"); return 1; }
    HIP_OK(hipMemcpy(g.p, gamma, (size_t)C * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(b.p, beta, (size_t)C * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(tmp.p, x, (size_t)rows * C * 4, hipMemcpyHostToDevice));
    if (launch_cast_rows(reinterpret_cast<const float*>(tmp.p), in.p, e->S, rows, C, C, e->stream)) return 1;
    Act a, o;
    a.p = in.p; a.B = rows; a.H = 1; a.W = 1; a.C = C; a.dt = e->S;
    o = a; o.p = out.p; o.dt = DT_F32;
    PD_TRY(e->layernorm(a, o, reinterpret_cast<float*>(g.p), reinterpret_cast<float*>(b.p)));
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy(y, out.p, (size_t)rows * C * 4, hipMemcpyDeviceToHost));
    return 0;
}

// softmax(q k^T dh^-0.5) v per head; q [B,Nq,C], k/v [B,Nk,C] fp32, C = heads*dh (heads from the engine config)
int pd_op_attention(pd_engine* e, const float* q, const float* k, const float* v, int B, int Nq, int Nk, int C, float* o) {
    if (!e || !q || !k || !v || !o) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    const int lpad = round_up(Nk, 8);
    const size_t eb = dt_size(e->T);
    std::vector<float> vt((size_t)B * C * lpad, 0.f);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < Nk; ++n)
            for (int c = 0; c < C; ++c) vt[((size_t)b * C + c) * lpad + n] = v[((size_t)b * Nk + n) * C + c];
    DevBuf tq((size_t)B * Nq * C * 4), tk((size_t)B * Nk * C * 4), tv(vt.size() * 4);
    DevBuf dq((size_t)B * Nq * C * eb), dk((size_t)B * Nk * C * eb), dv(vt.size() * eb), dout((size_t)B * Nq * C * eb), fo((size_t)B * Nq * C * 4);
    if (!tq.p || !tk.p || !tv.p || !dq.p || !dk.p || !dv.p || !dout.p || !fo.p) { pd_set_error("allocation failed"); return 1; }
    HIP_OK(hipMemcpy(tq.p, q, (size_t)B * Nq * C * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(tk.p, k, (size_t)B * Nk * C * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(tv.p, vt.data(), vt.size() * 4, hipMemcpyHostToDevice));
    if (launch_cast_rows(reinterpret_cast<const float*>(tq.p), dq.p, e->T, (long long)B * Nq, C, C, e->stream)) return 1;
    if (launch_cast_rows(reinterpret_cast<const float*>(tk.p), dk.p, e->T, (long long)B * Nk, C, C, e->stream)) return 1;
    if (launch_cast_rows(reinterpret_cast<const float*>(tv.p), dv.p, e->T, (long long)B * C, lpad, lpad, e->stream)) return 1;
    PD_TRY(e->attention(dq.p, C, dk.p, C, dv.p, lpad, dout.p, C, B, Nq, Nk, C));
    if (launch_nhwc_to_nchw(dout.p, e->T, reinterpret_cast<float*>(fo.p), 1, 1, 1, B * Nq * C, 1, 1.f, e->stream)) return 1;
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy(o, fo.p, (size_t)B * Nq * C * 4, hipMemcpyDeviceToHost));
    return 0;
}

// One whole SpatialTransformer block of the engine's own networks (GroupNorm eps 1e-6, proj_in, BasicTransformerBlock, proj_out, + x;
// attention.py:321-340 and :271-275) with the weights loaded under `prefix` (e.g. "model.diffusion_model.input_blocks.1.1.") on
// x [B, C, H, W] and context [B, context_len, context_dim], through exactly the code path a sampling step takes -- including the fused
// tail kernel (st_tail.hip) where it is eligible.
int pd_op_spatial_transformer(pd_engine* e, const char* prefix, const float* x, const float* ctx, int B, int H, int W, float* y) {
    if (!e || !prefix || !x || !ctx || !y) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    const std::string pre(prefix);
    auto it = e->index.find(pre + "norm.weight");
    if (it == e->index.end()) { pd_set_error("no SpatialTransformer under '%s'", prefix); return 1; }
    const float* gn_g = e->params[it->second].vdst;
    STW* st = nullptr;
    for (NetW* net : {&e->unet, &e->cnet})
        for (STW* s : net->st_list)
            if (s->gn_g == gn_g) st = s;
    if (!st) { pd_set_error("no SpatialTransformer under '%s'", prefix); return 1; }
    for (const Param& p : e->params)
        if (!p.loaded && p.name.compare(0, pre.size(), pre) == 0) { pd_set_error("weights not loaded: '%s'", p.name.c_str()); return 1; }
    e->ln_dirty = true;
    PD_TRY(e->fold_layernorms());
    const int C = st->C, L = e->cfg.context_len, D = e->cfg.context_dim, Dp = round_up(D, 8), lpad = round_up(L, 8), N = H * W;
    const size_t eb = dt_size(e->T);
    // a private workspace for this call (op hooks run outside a session)
    const size_t ws = (size_t)64 << 20, act = (size_t)B * N * C * 4;
    DevBuf work(ws + 48 * act), cin((size_t)B * L * D * 4), cdev((size_t)B * L * Dp * eb), kbuf((size_t)B * L * C * eb), vtbuf((size_t)B * C * lpad * eb),
        kvp(st_tail_kv_bytes(B)), xin((size_t)B * N * C * dt_size(e->S));
    if (!work.p || !cin.p || !cdev.p || !kbuf.p || !vtbuf.p || !kvp.p || !xin.p) { pd_set_error("allocation failed"); return 1; }
    const Arena saved = e->arena;
    e->arena = Arena{};
    e->arena.base = reinterpret_cast<char*>(work.p);
    e->arena.cap = ws + 48 * act;
    int rc = 0;
    do {
        if ((rc = to_dev_nhwc(e, x, xin.p, e->S, B, C, H, W, C))) break;
        if (hipMemcpy(cin.p, ctx, (size_t)B * L * D * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = 1; break; }
        if ((rc = launch_cast_rows(reinterpret_cast<const float*>(cin.p), cdev.p, e->T, (long long)B * L, D, Dp, e->stream))) break;
        Act c, k;
        c.p = cdev.p; c.B = B; c.H = L; c.W = 1; c.C = Dp; c.dt = e->T;
        k.p = kbuf.p; k.B = B; k.H = L; k.W = 1; k.C = C; k.dt = e->T;
        if ((rc = e->gemm(st->kv2, c, k, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, vtbuf.p, C, lpad))) break;
        KVSlot kv;
        kv.K = kbuf.p; kv.VT = vtbuf.p;
        if (e->st_tail_on(*st, N) && st->tail_w) {
            if ((rc = launch_st_tail_kv_pack(kbuf.p, vtbuf.p, kvp.p, B, L, lpad, e->stream))) break;
            kv.P = kvp.p;
        }
        Act a, o;
        a.p = xin.p; a.B = B; a.H = H; a.W = W; a.C = C; a.dt = e->S;
        if ((rc = e->transformer(*st, a, o, kv))) break;
        rc = from_dev_nhwc(e, o.p, e->S, y, B, C, H, W, C);
    } while (0);
    (void)hipStreamSynchronize(e->stream);
    e->arena = saved;
    return rc;
}


// timestep_embedding + time_embed MLP of a loaded network: the first two stages of pd_engine::compute_emb
int pd_op_time_embed(pd_engine* e, int net, const int64_t* t, int n, float* temb, float* emb) {
    if (!e || !t || n < 1 || (!temb && !emb)) { pd_set_error("null argument"); return 1; }
    HIP_OK(hipSetDevice(e->device));
    NetW& nw = net ? e->cnet : e->unet;
    const int mc = e->cfg.model_channels, td = mc * 4;
    std::vector<float> host;
    pd_host_timestep_embedding(t, n, mc, host);
    if (temb) memcpy(temb, host.data(), host.size() * sizeof(float));
    if (!emb) return 0;
    DevBuf a((size_t)n * mc * 4), b((size_t)n * td * 4), c((size_t)n * td * 4);
    if (!a.p || !b.p || !c.p) { pd_set_error("allocation failed"); return 1; }
    HIP_OK(hipMemcpy(a.p, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    Act te, e1, e2;
    te.p = a.p; te.B = n; te.H = te.W = 1; te.C = mc; te.dt = DT_F32;
    e1 = te; e1.p = b.p; e1.C = td;
    e2 = e1; e2.p = c.p;
    PD_TRY(e->gemm(nw.te0, te, e1, 1, 0, /*silu*/ 1, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    PD_TRY(e->gemm(nw.te2, e1, e2, 1, 0, 0, 1.f, nullptr, nullptr, 0, false, nullptr, 0, 0));
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy(emb, c.p, (size_t)n * td * 4, hipMemcpyDeviceToHost));
    return 0;
}
}  // extern "C"
