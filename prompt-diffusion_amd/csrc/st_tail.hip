// Fused tail of a SpatialTransformer block at the 320-channel level, gfx950 (MI355X, CDNA4).
//
// Everything BasicTransformerBlock._forward does after the self-attention product (attention.py:271-275) plus proj_out and
// the block's residual (attention.py:338-340) for 128 tokens per workgroup, without touching HBM in between:
//     h1 = att . Wo1^T + b + h            (attn1.to_out + residual)
//     q  = LN2(h1) . Wq^T                 (attn2.to_q; gamma / beta folded into the weights / a bias)
//     o  = softmax(q k^T dh^-0.5) v       (8 heads x 40, the <= 96 context keys of this sample, fp32 logits and statistics)
//     h2 = o . Wo2^T + b + h1
//     h3 = (x * gelu(gate)) . W2^T + b + h2,   [x | gate] = LN3(h2) . W1^T + b      (GEGLU feed-forward, exact-erf GELU)
//     out = h3 . Wp^T + b + x_in          (proj_out + the SpatialTransformer residual)
// Unfused this is 8 launches that each stream a [65536, 320..2560] tensor through HBM at 13-30 % of the MFMA rate.
//
// Structure ("wave chain"): a workgroup is 4 waves, ONE per SIMD with the whole 512-entry register file; a wave owns 32 tokens
// for the whole chain.  Activations never leave registers: the 32 x 320 fp32 residual stream is ten 32x32 MFMA accumulator
// tiles (token on the lane, channel on the register), LayerNorm statistics are an in-lane sum plus one lane^32 exchange, and
// an accumulator tile IS the next product's B operand after a pairwise convert (v_mfma_f32_32x32x16: the k order inside a
// 16-step is a fixed permutation, which the weight packing absorbs).  The weights are the A operand: all matrices of the
// block are repacked once per weight load into the exact order the MFMAs consume them, 1 KB per fragment (64 lanes x 16 B), and
// streamed through a 6 x 20 KB LDS ring by LDS-DMA (global_load_lds_dwordx4), 5 steps ahead of their use, one barrier per 20
// fragments placed in the MIDDLE of a step so that the fragment reads (4 in flight) run across step boundaries.  Every fragment
// feeds exactly one MFMA of each of the 4 waves.  The context K / V^T of the sample travel through the same ring
// (st_tail_kv_pack_kernel lays them out per head pair).  tests/st_tail_emul.py restates every index map of this file in NumPy.
#include "pd_common.h"
#include "pd_mma.h"

namespace {

// Diagnostic build only (tools/micro/st_stamp.hip compiles this file with -DPD_STAMP): s_memtime at the phase boundaries of wave 0 of
// every workgroup, written to a buffer of its own.  No stamp executes in the product build.
#ifdef PD_STAMP
__device__ unsigned long long* g_stamp_buf = nullptr;
#define PD_STAMP_AT(slot)                                                                                          \
    do {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        unsigned long long t_;                                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (threadIdx.x == 0 && g_stamp_buf) g_stamp_buf[(size_t)blockIdx.x * 16 + (slot)] = t_;                    \
    } while (0)
#else
#define PD_STAMP_AT(slot) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef ST_NV
#define ST_NV 0   // vector instructions placed behind each MFMA of the feed-forward's first product (x * gelu(gate) of the previous chunk)
#endif
constexpr int TC = 320;          // channels
constexpr int TNT = 10;          // 32-channel tiles
constexpr int TKS = 20;          // k16 steps over TC
constexpr int THD = 8, TDH = 40; // heads x head dim
constexpr int THID = 1280;       // feed-forward hidden units (x and gate each)
constexpr int TCHUNK = 32;       // hidden units per feed-forward chunk
constexpr int TNCHUNK = THID / TCHUNK;
constexpr int SF = 20;           // fragments per ring step
constexpr int NS = 6;            // ring slots
constexpr int PD = 5;            // steps requested ahead
constexpr int STEP_BYTES = SF * 1024;
constexpr int RING_BYTES = NS * STEP_BYTES;
// weight stream (fragments): A attn1.to_out | B 4 x [q 60 | attn2.to_out 60] | C feed-forward, 40 chunks of [ff.net.0 40 | ff.net.2 20] in
// software-pipelined order: W1(0), {W1(c+1), W2(c)} for c = 0..38, W2(39) | D proj_out
constexpr int WF_A = TNT * TKS, WF_PAIR = 120, WF_B = 4 * WF_PAIR, WF_CHUNK = 60, WF_C = TNCHUNK * WF_CHUNK, WF_D = TNT * TKS;
constexpr int WF_TOTAL = WF_A + WF_B + WF_C + WF_D;   // 3280
constexpr int KV_PAIR = 60;      // fragments per head pair: [K h0 9][V h0 12][K h1 9][V h1 12][pad 18]
constexpr int STEPS_A = 10, STEPS_PAIR = 9, STEPS_CHUNK = 3, STEPS_D = 10;
constexpr int STEPS_TOTAL = STEPS_A + 4 * STEPS_PAIR + TNCHUNK * STEPS_CHUNK + STEPS_D;   // 176
// fp32 vectors in LDS behind the ring
constexpr int V_BO1 = 0, V_BQ = 320, V_BO2 = V_BQ + 384, V_B1 = V_BO2 + 320, V_B2 = V_B1 + TNCHUNK * 2 * 32, V_BP = V_B2 + 320, V_TOTAL = V_BP + 320;

__host__ __device__ constexpr int sigma(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
__host__ __device__ constexpr int kmap0(int ks, int hk) { return 32 * (ks >> 1) + 16 * hk + 8 * (ks & 1); }

// LDS-DMA of 64 x 16 B: wave-uniform 64-bit base in SGPRs + a 32-bit lane offset.  M0 carries the LDS destination; it is
// compiler-reserved, so it is saved / restored inside the statement.
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

extern __shared__ __attribute__((aligned(16))) char smem[];
__device__ __forceinline__ uint4 ldsr(unsigned addr) { return *reinterpret_cast<const uint4*>(smem + addr); }

template <int P> __device__ __forceinline__ void mfma32(const uint4& a, const uint4& b, f32x16& c) {
    if constexpr (P == DT_F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// registers 8s .. 8s+7 of an accumulator tile as the B fragment of k16 step s of the next product
template <int P, int S> __device__ __forceinline__ uint4 acc_frag(const f32x16& a) {
    uint4 u;
    u.x = pack2<P>(a[8 * S + 0], a[8 * S + 1]); u.y = pack2<P>(a[8 * S + 2], a[8 * S + 3]);
    u.z = pack2<P>(a[8 * S + 4], a[8 * S + 5]); u.w = pack2<P>(a[8 * S + 6], a[8 * S + 7]);
    return u;
}
// 16 consecutive fp32 of the vector area: floats [index + 16 * (lane >> 5), +16).  The lane half comes out of lane16 (= lane * 16, live
// in every step anyway) behind an opaque asm: a hoisted per-lane address would be one more register alive across the whole kernel, and
// hipcc spilled exactly that one -- its reload put an `s_waitcnt vmcnt(0)` at the head of the feed-forward loop, which drained the
// five steps of LDS-DMA prefetch in every iteration (85 cycles per MFMA there instead of 60).
__device__ __forceinline__ f32x16 lds_vec16(unsigned float_index, unsigned lane16) {
    unsigned l = lane16;
    asm volatile("" : "+v"(l));
    const f32x4* p = reinterpret_cast<const f32x4*>(smem + RING_BYTES + (l >> 9) * 64) + (float_index >> 2);
    const f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
    return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }

struct StTailArgs {
    const void* att;      // [M][320] compute type: self-attention output
    const void* h;        // [M][320] stream type: the residual of attn1 (proj_in output)
    const void* x_in;     // [M][320] stream type: the SpatialTransformer's input
    void* out;            // [M][320] stream type
    const char* wpk;      // WF_TOTAL fragments
    const float* vec;     // V_TOTAL floats
    const char* kvp;      // [B][4][KV_PAIR] fragments
    int rows_per_sample, Nk;
    float scale_log2e;    // dh^-0.5 * log2(e)
    int in_rows;          // att / h / x_in hold this many rows; output row r reads input row r % in_rows (the shared front of a CFG pair: pd_engine::transformer)
};

// front kernel (GroupNorm apply + proj_in + norm1 + to_q/k/v): [proj_in 200 | qkv 600] fragments, a linear stream of 40 steps
constexpr int FR_STEPS = 40, FR_WF = FR_STEPS * SF;
constexpr int FV_BPI = 0, FV_BQKV = 320, FV_COEF = FV_BQKV + 960, FV_TOTAL = FV_COEF + 640;

// the ring: one wave-uniform cursor; step st lives in slot st % NS
template <int KIND>
struct Pipe {
    const char* wsrc;   // weight stream (wave-uniform)
    const char* kvsrc;  // this sample's K / V fragments (wave-uniform)
    int st;             // step being consumed
    unsigned cur, nxt;  // LDS address of (slot, lane) of the current / next step
    unsigned lane16;
    unsigned lds0;      // LDS byte address of smem[0] (0 unless static LDS precedes the dynamic region)
    int wave;
    uint4 w0, w1, w2, w3;
#ifdef PD_STAMP
    unsigned long long t_vm = 0, t_bar = 0, t_dma = 0;   // cycles in the step's DMA wait, barrier and DMA issue (diagnostic build)
#endif
    template <int I> __device__ __forceinline__ uint4& reg() {
        if constexpr (I == 0) return w0; else if constexpr (I == 1) return w1; else if constexpr (I == 2) return w2; else return w3;
    }
    __device__ __forceinline__ const char* step_src(int s) const {
        if constexpr (KIND == 1) return wsrc + (size_t)(s < FR_STEPS ? s : FR_STEPS - 1) * STEP_BYTES;
        if (s >= STEPS_TOTAL) s = STEPS_TOTAL - 1;   // past the end: keep the request count per step uniform (harmless re-reads)
        if (s < STEPS_A) return wsrc + (size_t)s * STEP_BYTES;
        if (s < STEPS_A + 4 * STEPS_PAIR) {
            const int p = (s - STEPS_A) / STEPS_PAIR, r = (s - STEPS_A) - p * STEPS_PAIR;
            if (r < 3) return wsrc + (size_t)(WF_A + p * WF_PAIR + r * SF) * 1024;
            if (r < 6) return kvsrc + (size_t)(p * KV_PAIR + (r - 3) * SF) * 1024;
            return wsrc + (size_t)(WF_A + p * WF_PAIR + 60 + (r - 6) * SF) * 1024;
        }
        return wsrc + (size_t)(WF_A + WF_B + (s - STEPS_A - 4 * STEPS_PAIR) * SF) * 1024;
    }
    __device__ __forceinline__ void issue(int s) {
        s = sgpr(s);
        const char* src = step_src(s) + wave * 1024;
        const unsigned dst = lds0 + (((unsigned)s % NS) * SF + wave) * 1024;
#pragma unroll
        for (int q = 0; q < SF / 4; ++q) glds16(src + q * 4096, lane16, (unsigned)sgpr((int)dst + q * 4096));
    }
    __device__ __forceinline__ void advance() {
        st = sgpr(st + 1);
        cur = nxt;
        nxt = (unsigned)sgpr((int)(((unsigned)(st + 1) % NS) * STEP_BYTES)) + lane16;
    }
};

// One ring step: 20 fragments, fragment f handed to op(f, fragment) in order while fragment f + 4 is being read; the step's
// barrier sits after fragment 9: every wave's own requests for step st + 1 have landed (all but the 15 youngest of its
// LDS-DMAs are complete), all waves are past step st - 1, so its slot takes the requests for step st + 5.
// NV > 0: the step's ops carry independent vector work (x * gelu(gate) of the previous feed-forward chunk): up to NV of those
// instructions are placed behind every MFMA, where they issue under its 32 cycles
template <int NV = 0, class PIPE, class F> __device__ __forceinline__ void run_step(PIPE& pp, F&& op) {
    static_for<SF>([&](auto I) __attribute__((always_inline)) {
        constexpr int f = decltype(I)::value;
        uint4& w = pp.template reg<f % 4>();
        const bool used = op(I, w);
        if constexpr (f + 4 < SF) w = ldsr(pp.cur + (f + 4) * 1024); else w = ldsr(pp.nxt + (f + 4 - SF) * 1024);
        if (used) __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x2, NV, 0);
        if constexpr (f == 9) {
#ifdef PD_STAMP
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long s0_ = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            const unsigned long long s1_ = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const unsigned long long s2_ = __builtin_amdgcn_s_memtime();
            pp.issue(pp.st + PD);
            const unsigned long long s3_ = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            pp.t_vm += s1_ - s0_; pp.t_bar += s2_ - s1_; pp.t_dma += s3_ - s2_;
#else
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            pp.issue(pp.st + PD);
#endif
        }
    });
    pp.advance();
}

// two-pass LayerNorm statistics of the lane pair's row over the ten accumulator tiles -> (x - mean) * rstd as 20 B fragments
template <int P>
__device__ __forceinline__ void ln_frags(const f32x16 (&acc)[TNT], uint4 (&yf)[TKS]) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TNT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    s += xor32(s);
    const float mean = s * (1.0f / TC);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < TNT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mean; q = fmaf(d, d, q); }
    q += xor32(q);
    const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / TC) + 1e-5f);
#pragma unroll
    for (int t = 0; t < TNT; ++t) {
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = (acc[t][r] - mean) * rstd;
        yf[2 * t] = acc_frag<P, 0>(y);
        yf[2 * t + 1] = acc_frag<P, 1>(y);
    }
}

// 16 consecutive channels of a row of a stream-type tensor (fp32 or the 2-byte compute type) as floats
template <int P, bool F32> __device__ __forceinline__ f32x16 load_row16(const void* base, size_t elem) {
    f32x16 v;
    if constexpr (F32) {
        const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const f32x4 t = p[i]; v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3]; }
    } else {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + elem);
        float f[16];
        unpack8<P>(p[0], f);
        unpack8<P>(p[1], f + 8);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = f[i];
    }
    return v;
}
template <int P, bool F32> __device__ __forceinline__ void store_row16(void* base, size_t elem, const f32x16& v) {
    if constexpr (F32) {
        f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem);
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    } else {
        uint4* p = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + elem);
        p[0] = acc_frag<P, 0>(v);
        p[1] = acc_frag<P, 1>(v);
    }
}

template <int P, bool SF32>
__global__ __launch_bounds__(256, 1) void st_tail_kernel(StTailArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + (lane & 31);
    {   // inputs shared by the two halves of a CFG batch: a workgroup of the second half reads the first half's rows (scalar pointer offset)
        const unsigned blk0 = blockIdx.x * 128u;
        const long long back = (long long)(blk0 - blk0 % (unsigned)a.in_rows);
        a.att = reinterpret_cast<const char*>(a.att) - back * TC * (long long)sizeof(uint16_t);
        a.h = reinterpret_cast<const char*>(a.h) - back * TC * (long long)(SF32 ? 4 : 2);
        a.x_in = reinterpret_cast<const char*>(a.x_in) - back * TC * (long long)(SF32 ? 4 : 2);
    }

    PD_STAMP_AT(0);
    Pipe<0> pp;
    pp.wsrc = a.wpk;
    pp.kvsrc = a.kvp + (size_t)((blockIdx.x * 128) / a.rows_per_sample) * (4 * KV_PAIR * 1024);
    pp.st = 0;
    pp.lane16 = lane * 16;
    pp.lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    pp.wave = sgpr(wave);
    for (int s = 0; s < PD; ++s) pp.issue(s);
    {   // the block's fp32 vectors -> LDS
        const f32x4* src = reinterpret_cast<const f32x4*>(a.vec);
        f32x4* dst = reinterpret_cast<f32x4*>(smem + RING_BYTES);
        for (int i = tid; i < V_TOTAL / 4; i += 256) dst[i] = src[i];
    }
    // ---- 1. h1 = att . Wo1^T + b + h: att as 20 B fragments (lane = token, 8 consecutive channels per k16 half step)
    uint4 yf[TKS];
    {
        const uint16_t* ar = reinterpret_cast<const uint16_t*>(a.att) + (size_t)row * TC;
#pragma unroll
        for (int ks = 0; ks < TKS; ++ks) yf[ks] = *reinterpret_cast<const uint4*>(ar + kmap0(ks, 0) + 16 * hh);
    }
    f32x16 acc[TNT];
#pragma unroll
    for (int t = 0; t < TNT; ++t) acc[t] = load_row16<P, SF32>(a.h, (size_t)row * TC + 32 * t + 16 * hh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TNT; ++t) acc[t] += lds_vec16(V_BO1 + 32 * t, pp.lane16);
    pp.cur = pp.lane16;
    pp.nxt = STEP_BYTES + pp.lane16;
    pp.w0 = ldsr(pp.cur); pp.w1 = ldsr(pp.cur + 1024); pp.w2 = ldsr(pp.cur + 2048); pp.w3 = ldsr(pp.cur + 3072);

    PD_STAMP_AT(1);
    static_for<TNT>([&](auto TN) __attribute__((always_inline)) {
        constexpr int tn = decltype(TN)::value;
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            mfma32<P>(w, yf[decltype(I)::value], acc[tn]);
            return true;
        });
    });
    PD_STAMP_AT(2);

    // ---- 2-5. cross-attention, two heads at a time
    ln_frags<P>(acc, yf);   // norm2 (gamma / beta live in the q weights / bias)
    for (int pr = 0; pr < 4; ++pr) {
        uint4 qf[6];
        static_for<3>([&](auto TL) __attribute__((always_inline)) {
            constexpr int tl = decltype(TL)::value;
            f32x16 qa = lds_vec16(V_BQ + 32 * (3 * pr + tl), pp.lane16);
            run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
                mfma32<P>(w, yf[decltype(I)::value], qa);
                return true;
            });
            qf[2 * tl] = acc_frag<P, 0>(qa);
            qf[2 * tl + 1] = acc_frag<P, 1>(qa);
        });
        // per head: S^T = K . q^T (keys on the accumulator rows: key 32 kt + 16 hh + r; tokens on the lanes), softmax over the
        // keys, O^T = V^T . P^T (head dim on the rows: d = 32 dt + 16 hh + r, 40 of the 64 real).  The pair's 60 fragments are
        // [K h0 9][V h0 12][K h1 9][V h1 12][pad 18]; the softmax of a head sits in front of its first V fragment.
        f32x16 S[3], O[2];
        uint4 pf[6], of[2][3];
        float linv = 0.f;
        auto zero_s = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
        };
        auto softmax = [&]() __attribute__((always_inline)) {
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = 32 * kt + 16 * hh + r < a.Nk;
                    S[kt][r] = ok ? S[kt][r] : -INFINITY;
                    m = fmaxf(m, S[kt][r]);
                }
            m = fmaxf(m, xor32(m));
            const float mc = m * a.scale_log2e;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(S[kt][r], a.scale_log2e, -mc));
                    S[kt][r] = e;
                    l += e;
                }
            l += xor32(l);
            linv = __builtin_amdgcn_rcpf(l);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                pf[2 * kt] = acc_frag<P, 0>(S[kt]);
                pf[2 * kt + 1] = acc_frag<P, 1>(S[kt]);
            }
        };
        auto finish_o = [&](auto HL) __attribute__((always_inline)) {
            constexpr int hl = decltype(HL)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) O[0][r] *= linv;
#pragma unroll
            for (int r = 0; r < 8; ++r) O[1][r] *= linv;
            of[hl][0] = acc_frag<P, 0>(O[0]);
            of[hl][1] = acc_frag<P, 1>(O[0]);
            of[hl][2] = acc_frag<P, 0>(O[1]);
        };
        zero_s();
        static_for<3>([&](auto SG) __attribute__((always_inline)) {
            constexpr int sg = decltype(SG)::value;
            run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
                constexpr int o = sg * SF + decltype(I)::value;
                if constexpr (o < 42) {
                    constexpr int hl = o / 21, q = o % 21;
                    if constexpr (q < 9) {
                        if constexpr (q == 0 && hl == 1) {
                            finish_o(std::integral_constant<int, 0>{});
                            zero_s();
                        }
                        mfma32<P>(w, qf[3 * hl + q % 3], S[q / 3]);
                    } else {
                        if constexpr (q == 9) softmax();
                        mfma32<P>(w, pf[(q - 9) % 6], O[(q - 9) / 6]);
                    }
                    return true;
                } else {
                    if constexpr (o == 42) finish_o(std::integral_constant<int, 1>{});
                    return false;
                }
            });
        });
        // h2 += o . Wo2^T for the two heads (3 k16 steps each)
        static_for<3>([&](auto SG) __attribute__((always_inline)) {
            constexpr int sg = decltype(SG)::value;
            run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
                constexpr int o = sg * SF + decltype(I)::value;
                constexpr int hl = o / 30, tn = (o % 30) / 3, ksl = o % 3;
                mfma32<P>(w, of[hl][ksl], acc[tn]);
                return true;
            });
        });
    }
#pragma unroll
    for (int t = 0; t < TNT; ++t) acc[t] += lds_vec16(V_BO2 + 32 * t, pp.lane16);
    PD_STAMP_AT(3);

    // ---- 6-8. GEGLU feed-forward in chunks of 32 hidden units, software-pipelined: while the matrix pipe runs ff.net.0 of chunk
    // c + 1 ([x | gate] tiles, 40 MFMAs) the vector pipe does x * gelu(gate) of chunk c, 2 values per 5 MFMAs; then ff.net.2 of
    // chunk c (2 k16 steps into each of the 10 output tiles)
    ln_frags<P>(acc, yf);   // norm3
#pragma unroll
    for (int t = 0; t < TNT; ++t) acc[t] += lds_vec16(V_B2 + 32 * t, pp.lane16);
    f32x16 a1[2];
    static_for<2>([&](auto TI) __attribute__((always_inline)) {
        constexpr int ti = decltype(TI)::value;
        a1[ti] = lds_vec16(V_B1 + ti * 32, pp.lane16);
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            mfma32<P>(w, yf[decltype(I)::value], a1[ti]);
            return true;
        });
    });
    for (int cc = 0; cc < TNCHUNK - 1; ++cc) {
        f32x16 a1n[2], g;
        static_for<2>([&](auto TI) __attribute__((always_inline)) {
            constexpr int ti = decltype(TI)::value;
            a1n[ti] = lds_vec16(V_B1 + ((cc + 1) * 2 + ti) * 32, pp.lane16);
            run_step<ST_NV>(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
                constexpr int o = ti * SF + decltype(I)::value;
                mfma32<P>(w, yf[decltype(I)::value], a1n[ti]);
                if constexpr (o % 5 == 0 || o % 5 == 2) {
                    constexpr int gi = (o / 5) * 2 + (o % 5 == 2 ? 1 : 0);
                    g[gi] = a1[0][gi] * gelu_fast(a1[1][gi]);
                }
                return true;
            });
        });
        const uint4 gf0 = acc_frag<P, 0>(g), gf1 = acc_frag<P, 1>(g);
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            constexpr int o = decltype(I)::value;
            mfma32<P>(w, (o & 1) ? gf1 : gf0, acc[o / 2]);
            return true;
        });
        a1[0] = a1n[0];
        a1[1] = a1n[1];
    }
    {
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = a1[0][r] * gelu_fast(a1[1][r]);
        const uint4 gf0 = acc_frag<P, 0>(g), gf1 = acc_frag<P, 1>(g);
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            constexpr int o = decltype(I)::value;
            mfma32<P>(w, (o & 1) ? gf1 : gf0, acc[o / 2]);
            return true;
        });
    }

    PD_STAMP_AT(4);
    // ---- 9. out = h3 . Wp^T + b + x_in
#pragma unroll
    for (int t = 0; t < TNT; ++t) {
        yf[2 * t] = acc_frag<P, 0>(acc[t]);
        yf[2 * t + 1] = acc_frag<P, 1>(acc[t]);
    }
#pragma unroll
    for (int t = 0; t < TNT; ++t) acc[t] = load_row16<P, SF32>(a.x_in, (size_t)row * TC + 32 * t + 16 * hh) + lds_vec16(V_BP + 32 * t, pp.lane16);
    static_for<TNT>([&](auto TN) __attribute__((always_inline)) {
        constexpr int tn = decltype(TN)::value;
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            mfma32<P>(w, yf[decltype(I)::value], acc[tn]);
            return true;
        });
    });
    PD_STAMP_AT(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (clamped) requests must have landed before the LDS is released
#pragma unroll
    for (int t = 0; t < TNT; ++t) store_row16<P, SF32>(a.out, (size_t)row * TC + 32 * t + 16 * hh, acc[t]);
    PD_STAMP_AT(6);
#ifdef PD_STAMP
    if (threadIdx.x == 0 && g_stamp_buf) { g_stamp_buf[(size_t)blockIdx.x * 16 + 8] = pp.t_vm; g_stamp_buf[(size_t)blockIdx.x * 16 + 9] = pp.t_bar; g_stamp_buf[(size_t)blockIdx.x * 16 + 10] = pp.t_dma; }
#endif
}


// ------------------------------------------------------------------------------------------------ front of the block
// h = proj_in(GroupNorm(x)) (attention.py:331-333; GroupNorm statistics arrive as per-(sample, channel) {scale, shift}), then
// [q | k | v] = norm1(h) . Wqkv^T (attention.py:271: attn1 of norm1(x); gamma / beta folded into the weights / a bias).
// Outputs: h [M][320] (the residual the tail kernel adds), q | k [M][640] row-major, V^T [B][320][vt_ld] for the attention kernel.

// Two accumulator tiles (64 channels x the wave's 32 tokens) -> memory through a wave-private 4 KB LDS buffer, so that one store
// instruction writes whole 128-byte row segments (8 rows x 128 B) instead of 64 16-byte pieces of 64 different lines.
template <int P>
__device__ __forceinline__ void store_rows64(unsigned buf, int lane, const f32x16& o0, const f32x16& o1, uint16_t* dst_row0, int ld_elems) {
    const unsigned wa = buf + (lane & 31) * 128 + (lane >> 5) * 32;
    *reinterpret_cast<uint4*>(smem + wa) = acc_frag<P, 0>(o0);
    *reinterpret_cast<uint4*>(smem + wa + 16) = acc_frag<P, 1>(o0);
    *reinterpret_cast<uint4*>(smem + wa + 64) = acc_frag<P, 0>(o1);
    *reinterpret_cast<uint4*>(smem + wa + 80) = acc_frag<P, 1>(o1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = k * 64 + lane;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + buf + idx * 16);
        *reinterpret_cast<uint4*>(dst_row0 + (size_t)(idx >> 3) * ld_elems + (idx & 7) * 8) = v;
    }
}
// ... and transposed: [channel][32 tokens] rows of 64 bytes (V^T for the attention kernel)
template <int P>
__device__ __forceinline__ void store_cols64(unsigned buf, int lane, const f32x16& o0, const f32x16& o1, uint16_t* dst_ch0, int ld_elems) {
    const unsigned wa = buf + ((lane >> 5) * 16) * 64 + (lane & 31) * 2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        *reinterpret_cast<uint16_t*>(smem + wa + r * 64) = cvt16<P>(o0[r]);
        *reinterpret_cast<uint16_t*>(smem + wa + (32 + r) * 64) = cvt16<P>(o1[r]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = k * 64 + lane;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + buf + idx * 16);
        *reinterpret_cast<uint4*>(dst_ch0 + (size_t)(idx >> 2) * ld_elems + (idx & 3) * 8) = v;
    }
}

struct StFrontArgs {
    const void* x;        // [M][320] stream type
    const float* coef;    // [B][320][2] GroupNorm {scale, shift}
    void* h;              // [M][320] stream type
    void* qk;             // [M][640] compute type
    void* vt;             // [B][320][vt_ld] compute type
    const char* wpk;      // FR_WF fragments
    const float* vec;     // b_proj_in[320] | b_qkv[960]
    int rows_per_sample, vt_ld;
};

template <int P>
__global__ __launch_bounds__(256, 1) void st_front_kernel(StFrontArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int sample = (blockIdx.x * 128) / a.rows_per_sample;

    PD_STAMP_AT(0);
    Pipe<1> pp;
    pp.wsrc = a.wpk;
    pp.kvsrc = a.wpk;
    pp.st = 0;
    pp.lane16 = lane * 16;
    pp.lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    pp.wave = sgpr(wave);
    for (int s = 0; s < PD; ++s) pp.issue(s);
    {
        f32x4* dst = reinterpret_cast<f32x4*>(smem + RING_BYTES);
        const f32x4* src = reinterpret_cast<const f32x4*>(a.vec);
        for (int i = tid; i < FV_COEF / 4; i += 256) dst[i] = src[i];
        const f32x4* cs = reinterpret_cast<const f32x4*>(a.coef + (size_t)sample * TC * 2);
        for (int i = tid; i < 640 / 4; i += 256) dst[FV_COEF / 4 + i] = cs[i];
    }
    uint4 xr[TKS];
    {
        const uint16_t* xp = reinterpret_cast<const uint16_t*>(a.x) + (size_t)row * TC;
#pragma unroll
        for (int ks = 0; ks < TKS; ++ks) xr[ks] = *reinterpret_cast<const uint4*>(xp + kmap0(ks, hh));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PD_STAMP_AT(1);
    // GroupNorm apply while the rows become B fragments: y = x * scale[c] + shift[c], rounded to the compute type like the
    // per-layer path's GroupNorm output
    uint4 yf[TKS];
#pragma unroll
    for (int ks = 0; ks < TKS; ++ks) {
        float f[8];
        unpack8<P>(xr[ks], f);
        const f32x4* cf = reinterpret_cast<const f32x4*>(smem + RING_BYTES) + (FV_COEF + 2 * kmap0(ks, hh)) / 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 c = cf[j];   // {scale, shift} of two channels
            f[2 * j] = fmaf(f[2 * j], c[0], c[1]);
            f[2 * j + 1] = fmaf(f[2 * j + 1], c[2], c[3]);
        }
        yf[ks] = pack8<P>(f);
    }
    f32x16 acc[TNT];
#pragma unroll
    for (int t = 0; t < TNT; ++t) acc[t] = lds_vec16(FV_BPI + 32 * t, pp.lane16);
    pp.cur = pp.lane16;
    pp.nxt = STEP_BYTES + pp.lane16;
    pp.w0 = ldsr(pp.cur); pp.w1 = ldsr(pp.cur + 1024); pp.w2 = ldsr(pp.cur + 2048); pp.w3 = ldsr(pp.cur + 3072);
    static_for<TNT>([&](auto TN) __attribute__((always_inline)) {
        constexpr int tn = decltype(TN)::value;
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            mfma32<P>(w, yf[decltype(I)::value], acc[tn]);
            return true;
        });
    });
    PD_STAMP_AT(2);
    // the residual stream is stored in the (2-byte) stream type; norm1 sees the same rounded values the per-layer path reads back
    const unsigned sbuf = RING_BYTES + FV_TOTAL * 4 + wave * 4096;   // this wave's staging buffer for coalesced stores
    const int row0 = blockIdx.x * 128 + wave * 32;
#pragma unroll
    for (int t = 0; t < TNT; ++t) {
        float f[16];
        unpack8<P>(acc_frag<P, 0>(acc[t]), f);
        unpack8<P>(acc_frag<P, 1>(acc[t]), f + 8);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = f[r];
    }
#pragma unroll
    for (int t = 0; t < TNT; t += 2)
        store_rows64<P>(sbuf, lane, acc[t], acc[t + 1], reinterpret_cast<uint16_t*>(a.h) + (size_t)row0 * TC + 32 * t, TC);
    ln_frags<P>(acc, yf);   // norm1
    PD_STAMP_AT(3);
    // q | k | v: 30 output tiles, one ring step each
    uint16_t* qk0 = reinterpret_cast<uint16_t*>(a.qk) + (size_t)row0 * (2 * TC);
    uint16_t* vt0 = reinterpret_cast<uint16_t*>(a.vt) + (size_t)sample * TC * a.vt_ld + (row0 - sample * a.rows_per_sample);
    for (int tg = 0; tg < 30; tg += 2) {   // two tiles per trip keeps the loop body at 40 MFMAs
        f32x16 o0 = lds_vec16(FV_BQKV + 32 * tg, pp.lane16), o1 = lds_vec16(FV_BQKV + 32 * (tg + 1), pp.lane16);
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            mfma32<P>(w, yf[decltype(I)::value], o0);
            return true;
        });
        run_step(pp, [&](auto I, const uint4& w) __attribute__((always_inline)) {
            mfma32<P>(w, yf[decltype(I)::value], o1);
            return true;
        });
        if (tg < 20) store_rows64<P>(sbuf, lane, o0, o1, qk0 + 32 * tg, 2 * TC);
        else store_cols64<P>(sbuf, lane, o0, o1, vt0 + (size_t)(32 * (tg - 20)) * a.vt_ld, a.vt_ld);
    }
    PD_STAMP_AT(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PD_STAMP_AT(5);
}

struct StFrontPackArgs { const uint16_t *wpi, *wqkv; int ld; uint16_t* dst; };
__global__ __launch_bounds__(256) void st_front_pack_kernel(StFrontPackArgs a) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int F = gid >> 6, lane = gid & 63;
    if (F >= FR_WF) return;
    const int i = lane & 31, hk = lane >> 5;
    const int tn = F / TKS, ks = F % TKS;   // tiles 0..9: proj_in, 10..39: the fused to_q | to_k | to_v rows
    const uint16_t* src = tn < TNT ? a.wpi + (size_t)(32 * tn + sigma(i)) * a.ld + kmap0(ks, hk)
                                   : a.wqkv + (size_t)(32 * (tn - TNT) + sigma(i)) * a.ld + kmap0(ks, hk);
    reinterpret_cast<uint4*>(a.dst)[gid] = *reinterpret_cast<const uint4*>(src);
}
__global__ __launch_bounds__(256) void st_front_vec_kernel(const float* bpi, const float* bqkv, float* dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < FV_BQKV) dst[i] = bpi[i];
    else if (i < FV_COEF) dst[i] = bqkv[i - FV_BQKV];
}

// ------------------------------------------------------------------------------------------------ packing
// weight stream: one thread per (fragment, lane): 8 consecutive input channels of one output row (or zeros)
struct StPackArgs {
    const uint16_t *wo1, *wq, *wo2, *w1, *w2, *wp;   // engine layouts [N][ld], compute type; wq / w1 with the LayerNorm gamma folded in
    int ld_c, ld_w2;                                 // row strides (elements): the C-wide matrices, ff.net.2
    uint16_t* dst;
};
__device__ __forceinline__ int geglu_row(int u, int gate) { return (u / 80) * 160 + (u % 80) + (gate ? 80 : 0); }   // engine's interleaved ff.net.0 rows

__global__ __launch_bounds__(256) void st_tail_pack_kernel(StPackArgs a) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int F = gid >> 6, lane = gid & 63;
    if (F >= WF_TOTAL) return;
    const int i = lane & 31, hk = lane >> 5;
    const uint16_t* src = nullptr;   // 8 elements
    if (F < WF_A) {
        const int tn = F / TKS, ks = F % TKS;
        src = a.wo1 + (size_t)(32 * tn + sigma(i)) * a.ld_c + kmap0(ks, hk);
    } else if (F < WF_A + WF_B) {
        const int pr = (F - WF_A) / WF_PAIR, r = (F - WF_A) % WF_PAIR;
        if (r < 60) {
            const int tq = 3 * pr + r / TKS, ks = r % TKS;
            const int sl = sigma(i), h2 = sl >> 4, rr = sl & 15;
            const int ksq = 2 * tq + (rr >> 3), head = ksq / 3, d = 16 * (ksq % 3) + 8 * h2 + (rr & 7);
            if (d < TDH) src = a.wq + (size_t)(head * TDH + d) * a.ld_c + kmap0(ks, hk);
        } else {
            const int o = r - 60, hl = o / 30, tn = (o % 30) / 3, ksl = o % 3;
            const int d0 = 32 * (ksl >> 1) + 16 * hk + 8 * (ksl & 1);
            if (d0 < TDH) src = a.wo2 + (size_t)(32 * tn + sigma(i)) * a.ld_c + (2 * pr + hl) * TDH + d0;
        }
    } else if (F < WF_A + WF_B + WF_C) {
        const int r = F - WF_A - WF_B;
        int cc, idx;      // chunk and index inside its ff.net.0 (idx < 40) or ff.net.2 (idx >= 40) fragments
        if (r < 40) { cc = 0; idx = r; }
        else if (r >= WF_C - 20) { cc = TNCHUNK - 1; idx = 40 + r - (WF_C - 20); }
        else {
            const int b = (r - 40) / 60, q = (r - 40) % 60;
            if (q < 40) { cc = b + 1; idx = q; } else { cc = b; idx = q; }
        }
        if (idx < 40) {
            const int gate = idx / TKS, ks = idx % TKS;
            src = a.w1 + (size_t)geglu_row(TCHUNK * cc + sigma(i), gate) * a.ld_c + kmap0(ks, hk);
        } else {
            const int o = idx - 40, tn = o / 2, ksl = o % 2;
            src = a.w2 + (size_t)(32 * tn + sigma(i)) * a.ld_w2 + TCHUNK * cc + 16 * hk + 8 * ksl;
        }
    } else {
        const int r = F - WF_A - WF_B - WF_C, tn = r / TKS, ks = r % TKS;
        src = a.wp + (size_t)(32 * tn + sigma(i)) * a.ld_c + kmap0(ks, hk);
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (src) v = *reinterpret_cast<const uint4*>(src);
    reinterpret_cast<uint4*>(a.dst)[gid] = v;
}

struct StVecArgs { const float *bo1, *bq, *bo2, *b1, *b2, *bp; float* dst; };   // bq / b1: the LayerNorm-folded biases (b1 in the engine's interleaved order)
__global__ __launch_bounds__(256) void st_tail_vec_kernel(StVecArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V_TOTAL) return;
    float v = 0.f;
    if (i < V_BQ) v = a.bo1[i];
    else if (i < V_BO2) {
        const int sl = i - V_BQ, tq = sl >> 5, h2 = (sl >> 4) & 1, rr = sl & 15;
        const int ksq = 2 * tq + (rr >> 3), head = ksq / 3, d = 16 * (ksq % 3) + 8 * h2 + (rr & 7);
        v = d < TDH ? a.bq[head * TDH + d] : 0.f;
    } else if (i < V_B1) v = a.bo2[i - V_BO2];
    else if (i < V_B2) {
        const int k = i - V_B1, cc = k / 64, gate = (k % 64) / 32, c = k % 32;
        v = a.b1[geglu_row(TCHUNK * cc + c, gate)];
    } else if (i < V_BP) v = a.b2[i - V_B2];
    else v = a.bp[i - V_BP];
    a.dst[i] = v;
}

// context K [B][Nk][320] / V^T [B][320][lpad] -> [B][4][KV_PAIR] fragments; one thread per (fragment, lane)
__global__ __launch_bounds__(256) void st_tail_kv_pack_kernel(const uint16_t* K, const uint16_t* VT, uint16_t* dst, int B, int Nk, int lpad) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(gid & 63);
    const long long FF = gid >> 6;
    if (FF >= (long long)B * 4 * KV_PAIR) return;
    const int F = (int)(FF % KV_PAIR), pr = (int)((FF / KV_PAIR) % 4), b = (int)(FF / (4 * KV_PAIR));
    const int i = lane & 31, hk = lane >> 5;
    uint16_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int hl = F / 21, q = F % 21;
    if (F < 42 && q < 9) {
        const int kt = q / 3, ks = q % 3;
        const int key = 32 * kt + sigma(i), d0 = 16 * ks + 8 * hk;
        if (key < Nk && d0 < TDH) {
            const uint16_t* s = K + ((size_t)b * Nk + key) * TC + (2 * pr + hl) * TDH + d0;
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = s[j];
        }
    } else if (F < 42) {
        const int o = q - 9, dt = o / 6, kk = o % 6;
        const int d = 32 * dt + sigma(i), key0 = 32 * (kk >> 1) + 16 * hk + 8 * (kk & 1);
        if (d < TDH) {
            const uint16_t* s = VT + ((size_t)b * TC + (2 * pr + hl) * TDH + d) * lpad;
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = key0 + j < Nk ? s[key0 + j] : (uint16_t)0;
        }
    }
    uint4 v;
    v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16); v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
    reinterpret_cast<uint4*>(dst)[gid] = v;
}

}  // namespace


size_t st_front_weight_bytes() { return (size_t)FR_WF * 1024; }
size_t st_front_vec_floats() { return FV_COEF; }
double st_front_flops(long long M) { return 2.0 * (double)M * TC * (4.0 * TC); }   // proj_in + to_q / to_k / to_v
int launch_st_front_pack(const void* wpi, const void* wqkv_ln, int ld, const float* bpi, const float* bqkv_ln, void* wdst, float* vdst, hipStream_t s) {
    StFrontPackArgs a{reinterpret_cast<const uint16_t*>(wpi), reinterpret_cast<const uint16_t*>(wqkv_ln), ld, reinterpret_cast<uint16_t*>(wdst)};
    hipLaunchKernelGGL(st_front_pack_kernel, dim3(FR_WF * 64 / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(st_front_vec_kernel, dim3((FV_COEF + 255) / 256), dim3(256), 0, s, bpi, bqkv_ln, vdst);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_st_front(const void* x, const float* coef, void* h, void* qk, void* vt, const void* wpk, const float* vec, long long M, int rows_per_sample,
                    int vt_ld, int prec, hipStream_t s) {
    if (M % 128 || rows_per_sample % 128) return 1;
    StFrontArgs a{x, coef, h, qk, vt, reinterpret_cast<const char*>(wpk), vec, rows_per_sample, vt_ld};
    constexpr int SMEM = RING_BYTES + FV_TOTAL * 4 + 4 * 4096;
    static unsigned long long done[2] = {0, 0};
    void (*kfn)(StFrontArgs) = nullptr;
    int slot = 0;
    if (prec == DT_F16) { kfn = st_front_kernel<DT_F16>; slot = 0; }
    else if (prec == DT_BF16) { kfn = st_front_kernel<DT_BF16>; slot = 1; }
    else return 1;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), SMEM, &done[slot])) return 1;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(M / 128)), dim3(256), SMEM, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

size_t st_tail_weight_bytes() { return (size_t)WF_TOTAL * 1024; }
size_t st_tail_vec_floats() { return V_TOTAL; }
size_t st_tail_kv_bytes(int B) { return (size_t)B * 4 * KV_PAIR * 1024; }
double st_tail_flops(long long M, int Nk) {   // algorithmic: the six linear layers + QK^T and PV over the real keys and head dim
    return 2.0 * (double)M * TC * (4.0 * TC + 8.0 * TC + 4.0 * TC) + 4.0 * (double)M * Nk * TC;
}
bool st_tail_eligible(int prec, int C, int heads, int rows_per_sample, int Nk) {
    return (prec == DT_F16 || prec == DT_BF16) && C == TC && heads == THD && rows_per_sample % 128 == 0 && Nk >= 1 && Nk <= 96;
}

int launch_st_tail_pack(const void* wo1, const void* wq_ln, const void* wo2, const void* w1_ln, const void* w2, const void* wp, int ld_c, int ld_w2,
                        void* dst, hipStream_t s) {
    StPackArgs a{reinterpret_cast<const uint16_t*>(wo1), reinterpret_cast<const uint16_t*>(wq_ln), reinterpret_cast<const uint16_t*>(wo2),
                 reinterpret_cast<const uint16_t*>(w1_ln), reinterpret_cast<const uint16_t*>(w2), reinterpret_cast<const uint16_t*>(wp), ld_c, ld_w2,
                 reinterpret_cast<uint16_t*>(dst)};
    hipLaunchKernelGGL(st_tail_pack_kernel, dim3(WF_TOTAL * 64 / 256), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_st_tail_vec(const float* bo1, const float* bq_ln, const float* bo2, const float* b1_ln, const float* b2, const float* bp, float* dst,
                       hipStream_t s) {
    StVecArgs a{bo1, bq_ln, bo2, b1_ln, b2, bp, dst};
    hipLaunchKernelGGL(st_tail_vec_kernel, dim3((V_TOTAL + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_st_tail_kv_pack(const void* K, const void* VT, void* dst, int B, int Nk, int lpad, hipStream_t s) {
    const long long n = (long long)B * 4 * KV_PAIR * 64;
    hipLaunchKernelGGL(st_tail_kv_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(K),
                       reinterpret_cast<const uint16_t*>(VT), reinterpret_cast<uint16_t*>(dst), B, Nk, lpad);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_st_tail(const void* att, const void* h, const void* x_in, void* out, const void* wpk, const float* vec, const void* kvp, long long M,
                   int rows_per_sample, int Nk, int s_dt, float scale, int prec, hipStream_t s, long long in_rows) {
    if (in_rows <= 0) in_rows = M;
    if (M % 128 || M > 0x7fffffff || rows_per_sample % 128 || Nk < 1 || Nk > 96 || in_rows % 128 || M % in_rows) return 1;
    StTailArgs a{att, h, x_in, out, reinterpret_cast<const char*>(wpk), vec, reinterpret_cast<const char*>(kvp), rows_per_sample, Nk,
                 scale * 1.4426950408889634f, (int)in_rows};
    constexpr int SMEM = RING_BYTES + V_TOTAL * 4;
    static unsigned long long done[2] = {0, 0};
    if (s_dt != prec) return 1;   // the stream type is the compute type here (option stream_f32 keeps the per-layer path: an fp32 stream's
                                  // loads in flight next to the 240 resident registers spill)
    void (*kfn)(StTailArgs) = nullptr;
    int slot = 0;
    if (prec == DT_F16) { kfn = st_tail_kernel<DT_F16, false>; slot = 0; }
    else if (prec == DT_BF16) { kfn = st_tail_kernel<DT_BF16, false>; slot = 1; }
    else return 1;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), SMEM, &done[slot])) return 1;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(M / 128)), dim3(256), SMEM, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
