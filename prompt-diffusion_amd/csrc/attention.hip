// Fused (flash-style) attention for gfx950: softmax(Q K^T * dh^-0.5) V without materialising the
// [N,N] score matrix the reference builds (ldm/modules/attention.py:171-193).  fp32 logits, fp32
// online softmax (the reference forces fp32 logits, attention.py:173-177).
//
// Layouts: Q [B, Nq, heads*dh] and K [B, Nk, heads*dh] token-major (they are column slices of the
// fused QKV / KV projection outputs); V arrives TRANSPOSED, VT [B, heads*dh, Nk_pad], written that
// way by the projection GEMM's epilogue so every query block can stage it with 16-byte loads.
// Block = 4 waves = 128 queries of one (sample, head); each wave owns 32 queries; K / V^T tiles of 64
// keys are staged in LDS and shared by the 4 waves.
// Both products are issued "swapped" so that the query index lives on the MFMA lane (column) axis:
//   S^T[key, query] = K . Q^T            (A = K rows from LDS, B = Q rows kept in registers)
//   O^T[d,   query] = V^T . P^T          (A = V^T rows from LDS, B = P^T straight from S^T's registers)
// The S^T accumulator of a 16-key tile holds keys 4*(lane>>4)+j for query lane&15, which is exactly the
// k-slot the next MFMA's B operand wants (k order permuted identically on the V^T read), so P never
// goes through LDS, softmax row statistics are lane-local plus two cross-lane steps (xor 16, 32), and
// the O^T rescale uses the lane's own alpha.
#include "pd_common.h"
#include "pd_mma.h"

namespace {

constexpr int TK = 64;  // keys per tile

// 2-byte V^T tiles: 128-byte rows (64 keys) holding the keys PERMUTED inside each group of 32, so that the 8 keys one lane
// feeds to a K = 32 PV MFMA (keys 4*fq + j of the two 16-key S^T tiles 2u and 2u+1 -- the order P already has in
// registers) are 16 contiguous bytes, with the 16-byte chunks XOR-swizzled like the GEMM's operand rows: one conflict-free
// ds_read_b128 per fragment.  (Round 1 read two 8-byte halves from 144-byte rows; hipcc merged them into a ds_read2_b64,
// which is banked mod 32: 2-way conflicts, 47 % of the kernel's LDS cycles.)
//   slot(key) = 32*(key/32) + 8*((key%16)/4) + 4*((key%32)/16) + key%4        (a 2-byte element index inside the row)
constexpr int VT_ROW = 128;
__device__ __forceinline__ int vt_slot(int k0) { return (k0 & ~31) + (((k0 & 15) >> 2) << 3) + (((k0 & 31) >> 4) << 2); }
__device__ __forceinline__ int vt_chunk(int row, int chunk) { return row * VT_ROW + (((chunk ^ (row >> 1)) & 7) << 4); }
// a staged 16-byte chunk = 8 consecutive keys starting at k0 (multiple of 8) lands as two 8-byte halves
__device__ __forceinline__ void vt_store(char* tile, int row, int k0, const uint4& v) {
    const int s0 = vt_slot(k0), s1 = vt_slot(k0 + 4);
    *reinterpret_cast<uint2*>(tile + vt_chunk(row, s0 >> 3) + (s0 & 7) * 2) = make_uint2(v.x, v.y);
    *reinterpret_cast<uint2*>(tile + vt_chunk(row, s1 >> 3) + (s1 & 7) * 2) = make_uint2(v.z, v.w);
}
// fragment of d-row `row` for the PV step u (keys 32u .. 32u+31), lane quarter fq
__device__ __forceinline__ uint4 vt_frag(const char* tile, int row, int u, int fq) {
    return *reinterpret_cast<const uint4*>(tile + vt_chunk(row, 4 * u + fq));
}

template <int P>
__device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x4& acc) { mma_raw<P>(a, b, acc); }

template <int P, int DH>
struct AttnCfg {
    static constexpr bool F32 = prec_f32_storage(P);
    static constexpr int EB = F32 ? 4 : 2;
    static constexpr int VEC = 16 / EB;
    static constexpr int NCH = DH * EB / 16;      // 16-byte chunks per head row
    static constexpr int KS = (NCH + 3) / 4;      // k-steps (4 chunks each) of the QK^T product
    static constexpr int NTD = (DH + 15) / 16;    // 16-wide d tiles of the output
    static constexpr int KROW = KS * 64 + 16;     // K tile row stride (odd multiple of 16 B: conflict-free b128 reads)
    static constexpr int VROW = F32 ? TK * EB + 16 : VT_ROW;   // V^T tile row stride (2-byte: swizzled 128-byte rows)
    static constexpr int VCH = TK * EB / 16;      // chunks per V^T row
    static constexpr int SMEM = TK * KROW + NTD * 16 * VROW;
    static_assert((DH * EB) % 16 == 0, "head dim must fill whole 16-byte chunks");
};

template <int P, int DH>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
    using Cfg = AttnCfg<P, DH>;
    constexpr bool F32 = prec_f32_storage(P);
    constexpr int EB = Cfg::EB, VEC = Cfg::VEC, NCH = Cfg::NCH, KS = Cfg::KS, NTD = Cfg::NTD;
    constexpr int KROW = Cfg::KROW, VROW = Cfg::VROW, VCH = Cfg::VCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + TK * KROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = blockIdx.x * 128 + wave * 32;

    const char* Qb = reinterpret_cast<const char*>(p.Q) + ((size_t)b * p.q_bs + (size_t)h * DH) * EB;
    const char* Kb = reinterpret_cast<const char*>(p.K) + ((size_t)b * p.k_bs + (size_t)h * DH) * EB;
    const char* Vb = reinterpret_cast<const char*>(p.VT) + ((size_t)b * p.vt_bs + (size_t)h * DH * p.vt_ld) * EB;

    // Q^T fragments (B operand): lane holds query (lane&15), k = chunk*VEC..
    uint4 qf[KS][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = q0 + qt * 16 + fr;
        q = q < p.Nq ? q : p.Nq - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 4 + fq;
            qf[ks][qt] = c < NCH ? *reinterpret_cast<const uint4*>(Qb + (size_t)q * p.ldq * EB + c * 16) : make_uint4(0, 0, 0, 0);
        }
    }

    f32x4 o[NTD][2];
#pragma unroll
    for (int n = 0; n < NTD; ++n) { o[n][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[n][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float mrow[2] = {-INFINITY, -INFINITY};
    float lrow[2] = {0.f, 0.f};
    const float sl2 = p.scale * 1.4426950408889634f;

    for (int t0 = 0; t0 < p.Nk; t0 += TK) {
        __syncthreads();
        // ---- stage K tile [64 keys][KS*4 chunks]
        for (int idx = tid; idx < TK * KS * 4; idx += 256) {
            const int r = idx / (KS * 4), c = idx - r * (KS * 4);
            const int key = t0 + r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < NCH && key < p.Nk) v = *reinterpret_cast<const uint4*>(Kb + (size_t)key * p.ldk * EB + c * 16);
            *reinterpret_cast<uint4*>(sK + r * KROW + c * 16) = v;
        }
        // ---- stage V^T tile [NTD*16 d][64 keys]
        const bool ragged = t0 + TK > p.Nk;
        for (int idx = tid; idx < NTD * 16 * VCH; idx += 256) {
            const int d = idx / VCH, c = idx - d * VCH;
            const int key = t0 + c * VEC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d < DH && key < p.Nk) {
                v = *reinterpret_cast<const uint4*>(Vb + ((size_t)d * p.vt_ld + key) * EB);
                if (ragged && key + VEC > p.Nk) {  // zero the pad keys inside the chunk (0 * garbage must stay 0)
                    if constexpr (F32) {
                        uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (key + j >= p.Nk) w[j] = 0;
                    } else {
                        uint16_t* w = reinterpret_cast<uint16_t*>(&v);
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (key + j >= p.Nk) w[j] = 0;
                    }
                }
            }
            if constexpr (F32) *reinterpret_cast<uint4*>(sV + d * VROW + c * 16) = v;
            else vt_store(sV, d, c * 8, v);
        }
        __syncthreads();

        // ---- S^T = K Q^T : 4 key tiles x 2 query tiles
        f32x4 s[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { s[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 kf[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) kf[kt] = *reinterpret_cast<const uint4*>(sK + (kt * 16 + fr) * KROW + (ks * 4 + fq) * 16);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                mma16<P>(kf[kt], qf[ks][0], s[kt][0]);
                mma16<P>(kf[kt], qf[ks][1], s[kt][1]);
            }
        }
        // ---- online softmax (base-2), query = lane&15 of each query tile
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = t0 + kt * 16 + fq * 4 + j;
                    float v = s[kt][qt][j] * sl2;
                    v = (key < p.Nk && (!p.causal || key <= q0 + qt * 16 + fr)) ? v : -INFINITY;
                    s[kt][qt][j] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrow[qt], mx);
            const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);
            mrow[qt] = mnew;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = __builtin_amdgcn_exp2f(s[kt][qt][j] - mnew);
                    s[kt][qt][j] = e;
                    ps += e;
                }
            lrow[qt] = lrow[qt] * alpha + ps;
#pragma unroll
            for (int n = 0; n < NTD; ++n) o[n][qt] *= alpha;
        }
        // ---- O^T += V^T P^T
        if constexpr (F32) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                uint4 pf[2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) pf[qt] = __builtin_bit_cast(uint4, s[kt][qt]);
#pragma unroll
                for (int n = 0; n < NTD; ++n) {
                    const uint4 vf = *reinterpret_cast<const uint4*>(sV + (n * 16 + fr) * VROW + (kt * 16 + fq * 4) * 4);
                    mma16<P>(vf, pf[0], o[n][0]);
                    mma16<P>(vf, pf[1], o[n][1]);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint4 pf[2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    pf[qt].x = pack2<P>(s[2 * u][qt][0], s[2 * u][qt][1]);
                    pf[qt].y = pack2<P>(s[2 * u][qt][2], s[2 * u][qt][3]);
                    pf[qt].z = pack2<P>(s[2 * u + 1][qt][0], s[2 * u + 1][qt][1]);
                    pf[qt].w = pack2<P>(s[2 * u + 1][qt][2], s[2 * u + 1][qt][3]);
                }
#pragma unroll
                for (int n = 0; n < NTD; ++n) {
                    const uint4 vf = vt_frag(sV, n * 16 + fr, u, fq);
                    mma16<P>(vf, pf[0], o[n][0]);
                    mma16<P>(vf, pf[1], o[n][1]);
                }
            }
        }
    }

    // ---- normalise and store: lane holds d = n*16 + 4*fq + j of query (lane&15)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = lrow[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int q = q0 + qt * 16 + fr;
        if (q >= p.Nq) continue;
#pragma unroll
        for (int n = 0; n < NTD; ++n) {
            const int d = n * 16 + fq * 4;
            if (d >= DH) continue;
            f32x4 v = o[n][qt] * inv;
            store4(p.O, (size_t)b * p.o_bs + (size_t)q * p.ldo + (size_t)h * DH + d, F32 ? (int)DT_F32 : P, v);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// bf16 fast path (dh <= 80): same math and fragment mapping as attn_kernel, but the next K / V^T tile is
// requested into registers BEFORE the current tile's MFMAs and written to the other LDS buffer after them
// (one barrier per tile, global latency hidden), the softmax scale is folded into one FMA feeding
// v_exp_f32, and key masking runs only on the ragged last tile.
template <int DH>
struct Attn2Cfg {
    static constexpr int NCH = DH * 2 / 16;
    static constexpr int KS = (NCH + 3) / 4;
    static constexpr int NTD = (DH + 15) / 16;
    // K rows: 128-byte rows with the GEMM's XOR swizzle when dh fits them (conflict-free ds_read_b128 fragments),
    // else padded rows
    static constexpr bool KSWZ = KS == 2;
    static constexpr int KROW = KSWZ ? 128 : KS * 64 + 16;
    static constexpr int VROW = VT_ROW;
    static constexpr int K_BYTES = TK * KROW, V_BYTES = NTD * 16 * VROW;
    static constexpr int K_SLOTS = TK * NCH;        // valid 16-byte chunks of a K tile
    static constexpr int V_SLOTS = DH * 8;          // valid chunks of a V^T tile (8 per row of 64 keys)
    static constexpr int K_IT = (K_SLOTS + 255) / 256, V_IT = (V_SLOTS + 255) / 256;
    static constexpr int SMEM = 2 * (K_BYTES + V_BYTES);
    static_assert(K_IT <= 5 && V_IT <= 5, "staging registers cover 5 + 5 pieces");
    // dh padded up to the MFMA shapes leaves free slots, used to move softmax VALU work onto the matrix pipe:
    //  SUBM: a free K-dim slot of S^T = K.Q^T holds K'[key][DH] = 1 and Q'[q][DH] = -m_ref[q], so the MFMA
    //        itself delivers s - m_ref (Q is pre-scaled by scale*log2e): no per-score FMA.  m_ref is a lazily
    //        updated bf16-representable reference (softmax is shift-invariant; it only has to bound exp2's
    //        argument), raised -- with one rescale of O -- when a tile exceeds it by more than LAZY_THR.
    //  ONES: a free row of V^T is all ones, so row DH of O^T = V^T.P^T accumulates sum_k P (rescaled with O).
    static constexpr bool SUBM = KS * 32 > DH;
    static constexpr bool ONES = NTD * 16 > DH;
};
constexpr float LAZY_THR = 8.0f;   // exp2 arguments stay <= 8: P <= 256 in bf16, fp32 sums far from overflow

// smallest bf16-representable value >= x (x finite)
__device__ __forceinline__ float bf16_ceil(float x) {
    unsigned u = __float_as_uint(x);
    u = x >= 0.f ? (u + 0xFFFFu) & 0xFFFF0000u : u & 0xFFFF0000u;
    return __uint_as_float(u);
}
// an fp16-representable value >= x within two ulps of it (|x| far below 65504: logits in exp2 units)
__device__ __forceinline__ float f16_ceil(float x) {
    float r = h2f(f2h(x));
    if (r < x) r = h2f(f2h(x + fabsf(x) * 0x1p-10f + 1e-7f));
    return r;
}
template <int P> __device__ __forceinline__ float ref_ceil(float x) {
    if constexpr (P == DT_F16) return f16_ceil(x); else return bf16_ceil(x);
}
template <int P> struct One16 { static constexpr unsigned v = P == DT_F16 ? 0x3C00u : 0x3F80u; };   // 1.0 in the 2-byte type

// P: DT_BF16 or DT_F16.  Non-causal only (the CLIP text transformer's causal attention runs attn_kernel).
// The hot loop is written for VALU issue, which bounds this kernel (round-2 counters: 169 vector instructions per wave and
// 64-key tile, 1028 issue cycles against 448 MFMA cycles): K / V^T tiles come in through raw buffer loads (per-thread
// 32-bit offsets computed once, the tile advance is a scalar offset; rows past Nk read as zero through the descriptor's
// range check), LDS addresses are per-thread constants with the double-buffer toggle folded into immediates (the loop is
// unrolled by two), and everything that only the ragged last tile needs sits behind scalar branches.
template <int DH, int P, int WPS>
__global__ __launch_bounds__(256, WPS) void attn2_kernel(AttnParams p) {
    using Cfg = Attn2Cfg<DH>;
    constexpr int NCH = Cfg::NCH, KS = Cfg::KS, NTD = Cfg::NTD, KROW = Cfg::KROW;
    constexpr int BUF = Cfg::K_BYTES + Cfg::V_BYTES;
    constexpr int K_IT = Cfg::K_IT, V_IT = Cfg::V_IT;
    constexpr bool SUBM = Cfg::SUBM, ONES = Cfg::ONES;
    constexpr int KSM = DH / 32, FQM = (DH % 32) / 8;    // fragment slot of K-dim index DH
    auto kpos = [](int row, int c) __attribute__((always_inline)) { return Cfg::KSWZ ? (c ^ ((row >> 1) & 7)) : c; };
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = blockIdx.x * 128 + wave * 32;

    const char* Qb = reinterpret_cast<const char*>(p.Q) + ((size_t)b * p.q_bs + (size_t)h * DH) * 2;
    // one descriptor per operand, based at this (sample, head): K rows >= Nk fall outside num_records and load as zero
    const char* Kb = reinterpret_cast<const char*>(p.K) + ((size_t)b * p.k_bs + (size_t)h * DH) * 2;
    const char* Vb = reinterpret_cast<const char*>(p.VT) + ((size_t)b * p.vt_bs + (size_t)h * DH * p.vt_ld) * 2;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Kb), 0, (int)(((size_t)(p.Nk - 1) * p.ldk + DH) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Vb), 0, (int)((size_t)DH * p.vt_ld * 2), 0x00020000);   // whole rows (vt_ld is a multiple of 8): the range check works on dwords

    uint4 qf[KS][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = q0 + qt * 16 + fr;
        q = q < p.Nq ? q : p.Nq - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 4 + fq;
            qf[ks][qt] = c < NCH ? *reinterpret_cast<const uint4*>(Qb + (size_t)q * p.ldq * 2 + c * 16) : make_uint4(0, 0, 0, 0);
        }
    }
    const float sl2 = p.scale * 1.4426950408889634f;
    if constexpr (SUBM) {   // Q <- Q * scale * log2(e): logits leave the MFMA in exp2 units
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                unsigned* w = reinterpret_cast<unsigned*>(&qf[ks][qt]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo, hi;
                    unpack2<P>(w[j], lo, hi);
                    w[j] = pack2<P>(lo * sl2, hi * sl2);
                }
            }
    }
    // zero the pad chunks / pad rows of both LDS buffers once (they never change)
    for (int i = tid; i < Cfg::SMEM / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

    // ---- per-thread staging slots, computed once: buffer offsets (bytes, tile 0) and LDS byte offsets (buffer 0)
    unsigned k_off[K_IT], v_off[V_IT];
    int k_lds[K_IT], v_lds0[V_IT], v_lds1[V_IT];
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
        const int s = tid + 256 * i;
        const int r = s / NCH, c = s - r * NCH;
        k_off[i] = (unsigned)((r * p.ldk + c * 8) * 2);
        k_lds[i] = r * KROW + kpos(r, c) * 16;
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
        const int s = tid + 256 * i;
        const int d = s >> 3, c = s & 7;
        v_off[i] = (unsigned)((d * p.vt_ld + c * 8) * 2);
        const int s0 = vt_slot(c * 8), s1 = vt_slot(c * 8 + 4);
        v_lds0[i] = Cfg::K_BYTES + vt_chunk(d, s0 >> 3) + (s0 & 7) * 2;
        v_lds1[i] = Cfg::K_BYTES + vt_chunk(d, s1 >> 3) + (s1 & 7) * 2;
    }
    // piece i of a tile exists for the waves with wave*64 + 256*i < SLOTS (slot counts are multiples of 64: wave-uniform)
    uint4 kr0, kr1, kr2, kr3, kr4, vr0, vr1, vr2, vr3, vr4;   // staging registers (named: see gemm.hip); dh <= 80 uses three of each
    auto KR = [&](auto I) __attribute__((always_inline)) -> uint4& {
        constexpr int i = decltype(I)::value;
        if constexpr (i == 0) return kr0; else if constexpr (i == 1) return kr1; else if constexpr (i == 2) return kr2; else if constexpr (i == 3) return kr3; else return kr4;
    };
    auto VR = [&](auto I) __attribute__((always_inline)) -> uint4& {
        constexpr int i = decltype(I)::value;
        if constexpr (i == 0) return vr0; else if constexpr (i == 1) return vr1; else if constexpr (i == 2) return vr2; else if constexpr (i == 3) return vr3; else return vr4;
    };
    auto load_tile = [&](int t0) __attribute__((always_inline)) {
        const unsigned sk = (unsigned)t0 * (unsigned)p.ldk * 2u, sv = (unsigned)t0 * 2u;   // scalar tile offsets
        static_for<K_IT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if ((i + 1) * 256 <= Cfg::K_SLOTS || wave * 64 + 256 * i < Cfg::K_SLOTS)
                KR(I) = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rk, k_off[i], sk, 0));
        });
        static_for<V_IT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if ((i + 1) * 256 <= Cfg::V_SLOTS || wave * 64 + 256 * i < Cfg::V_SLOTS)
                VR(I) = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rv, v_off[i], sv, 0));
        });
    };
    // buf is a compile-time constant at every call site: the LDS offsets fold into the store's immediate
    auto store_tile = [&](int buf, int t0) __attribute__((always_inline)) {
        char* base = smem + buf * BUF;
        if (t0 + TK > p.Nk) {   // ragged tile (the last one): V^T pad keys hold whatever follows the row -- 0 * garbage must stay 0
            static_for<V_IT>([&](auto I) __attribute__((always_inline)) {
                const int s = tid + 256 * decltype(I)::value;
                const int key = t0 + (s & 7) * 8;
                uint4 v = VR(I);
                unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w[j] &= (key + 2 * j < p.Nk ? 0x0000ffffu : 0u) | (key + 2 * j + 1 < p.Nk ? 0xffff0000u : 0u);
                VR(I) = v;
            });
        }
        static_for<K_IT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if ((i + 1) * 256 <= Cfg::K_SLOTS || wave * 64 + 256 * i < Cfg::K_SLOTS)
                *reinterpret_cast<uint4*>(base + k_lds[i]) = KR(I);
        });
        static_for<V_IT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if ((i + 1) * 256 <= Cfg::V_SLOTS || wave * 64 + 256 * i < Cfg::V_SLOTS) {
                const uint4 v = VR(I);
                *reinterpret_cast<uint2*>(base + v_lds0[i]) = make_uint2(v.x, v.y);
                *reinterpret_cast<uint2*>(base + v_lds1[i]) = make_uint2(v.z, v.w);
            }
        });
    };

    f32x4 o[NTD][2];
#pragma unroll
    for (int n = 0; n < NTD; ++n) { o[n][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[n][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float mrow[2] = {-INFINITY, -INFINITY};   // running max of the RAW logits (scale > 0)
    float lrow[2] = {0.f, 0.f};
    float mref[2] = {0.f, 0.f};               // SUBM: reference already subtracted by the MFMA (representable in P)

    load_tile(0);
    __syncthreads();   // zero fill done before the first tile lands on top of it
    if constexpr (SUBM) {   // K'[key][DH] = 1.0 in both buffers
        for (int i = tid; i < 2 * TK; i += 256)
            *reinterpret_cast<unsigned*>(smem + (i / TK) * BUF + (i % TK) * KROW + kpos(i % TK, NCH) * 16) = One16<P>::v;
    }
    if constexpr (ONES) {   // V^T row DH = 1.0 for all 64 keys, both buffers (a whole row: the key permutation does not matter)
        for (int i = tid; i < 2 * (TK / 2); i += 256)
            *reinterpret_cast<unsigned*>(smem + (i / (TK / 2)) * BUF + Cfg::K_BYTES + DH * VT_ROW + (i % (TK / 2)) * 4) = One16<P>::v * 0x10001u;
    }
    store_tile(0, 0);
    __syncthreads();

    // fragment read offsets (buffer 0): K rows kt*16+fr / V^T rows n*16+fr differ from these by compile-time immediates
    int kf_off[KS], vf_off[2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf_off[ks] = fr * KROW + kpos(fr, ks * 4 + fq) * 16;
#pragma unroll
    for (int u = 0; u < 2; ++u) vf_off[u] = Cfg::K_BYTES + vt_chunk(fr, 4 * u + fq);

    const int ntiles = (p.Nk + TK - 1) / TK;
    auto tile = [&](auto BUFC, int t) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUFC)::value;
        const int t0 = t * TK;
        if (t + 1 < ntiles) load_tile(t0 + TK);
        const char* sb = smem + buf * BUF;

        f32x4 s[4][2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const uint4 kf = *reinterpret_cast<const uint4*>(sb + kf_off[ks] + kt * 16 * KROW);
                if (ks == 0) {
                    s[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    s[kt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                mma16<P>(kf, qf[ks][0], s[kt][0]);
                mma16<P>(kf, qf[ks][1], s[kt][1]);
            }
        }
        if (t0 + TK > p.Nk) {  // ragged last tile: mask the pad keys
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = t0 + kt * 16 + fq * 4 + j;
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        if (key >= p.Nk) s[kt][qt][j] = -INFINITY;
                }
        }
        if constexpr (SUBM) {
            float mx[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float m = max3f(s[0][qt][0], s[0][qt][1], s[0][qt][2]);
                m = max3f(m, s[0][qt][3], s[1][qt][0]);
                m = max3f(m, s[1][qt][1], s[1][qt][2]);
                m = max3f(m, s[1][qt][3], s[2][qt][0]);
                m = max3f(m, s[2][qt][1], s[2][qt][2]);
                m = max3f(m, s[2][qt][3], s[3][qt][0]);
                m = max3f(m, s[3][qt][1], s[3][qt][2]);
                mx[qt] = fmaxf(m, s[3][qt][3]);
            }
            // s already is (logit - m_ref) in exp2 units.  Rare path (always the first tile): move the reference.
            if (t == 0 || __any(fmaxf(mx[0], mx[1]) > LAZY_THR)) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    float m = mx[qt];
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    const float mnew = (t == 0 || m > 0.f) ? ref_ceil<P>(mref[qt] + m) : mref[qt];
                    const float delta = mnew - mref[qt];
                    mref[qt] = mnew;
                    if (fq == FQM) qf[KSM][qt].x = cvt16<P>(-mnew);   // Q'[q][DH] = -m_ref (Q'[q][DH+1] = 0)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[kt][qt][j] -= delta;
                    if (t > 0) {
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        if constexpr (!ONES) lrow[qt] *= alpha;
#pragma unroll
                        for (int n = 0; n < NTD; ++n) o[n][qt] *= alpha;
                    }
                }
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float ps = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float e = __builtin_amdgcn_exp2f(s[kt][qt][j]);
                        s[kt][qt][j] = e;
                        if constexpr (!ONES) ps += e;
                    }
                if constexpr (!ONES) lrow[qt] += ps;
            }
        } else {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float mx = max3f(s[0][qt][0], s[0][qt][1], s[0][qt][2]);
                mx = max3f(mx, s[0][qt][3], s[1][qt][0]);
                mx = max3f(mx, s[1][qt][1], s[1][qt][2]);
                mx = max3f(mx, s[1][qt][3], s[2][qt][0]);
                mx = max3f(mx, s[2][qt][1], s[2][qt][2]);
                mx = max3f(mx, s[2][qt][3], s[3][qt][0]);
                mx = max3f(mx, s[3][qt][1], s[3][qt][2]);
                mx = max3f(mx, mx, s[3][qt][3]);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mnew = fmaxf(mrow[qt], mx);
                const float alpha = __builtin_amdgcn_exp2f((mrow[qt] - mnew) * sl2);
                mrow[qt] = mnew;
                const float nm = -mnew * sl2;
                float ps = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][j], sl2, nm));
                        s[kt][qt][j] = e;
                        ps += e;
                    }
                lrow[qt] = fmaf(lrow[qt], alpha, ps);
#pragma unroll
                for (int n = 0; n < NTD; ++n) o[n][qt] *= alpha;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint4 pf[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                pf[qt].x = pack2<P>(s[2 * u][qt][0], s[2 * u][qt][1]);
                pf[qt].y = pack2<P>(s[2 * u][qt][2], s[2 * u][qt][3]);
                pf[qt].z = pack2<P>(s[2 * u + 1][qt][0], s[2 * u + 1][qt][1]);
                pf[qt].w = pack2<P>(s[2 * u + 1][qt][2], s[2 * u + 1][qt][3]);
            }
#pragma unroll
            for (int n = 0; n < NTD; ++n) {
                const uint4 vf = *reinterpret_cast<const uint4*>(sb + vf_off[u] + n * 16 * VT_ROW);
                mma16<P>(vf, pf[0], o[n][0]);
                mma16<P>(vf, pf[1], o[n][1]);
            }
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1, t0 + TK);   // buffer buf^1 was last read in tile t-1 (barrier below)
        __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    }

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l;
        if constexpr (ONES) {   // row DH of O^T: held by the lanes with 4*fq == DH % 16
            l = __shfl(o[DH / 16][qt][0], ((DH % 16) / 4) * 16 + fr);
        } else {
            l = lrow[qt];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        const float inv = 1.0f / l;
        const int q = q0 + qt * 16 + fr;
        if (q >= p.Nq) continue;
#pragma unroll
        for (int n = 0; n < NTD; ++n) {
            const int d = n * 16 + fq * 4;
            if (d >= DH) continue;
            store4(p.O, (size_t)b * p.o_bs + (size_t)q * p.ldo + (size_t)h * DH + d, P, o[n][qt] * inv);
        }
    }
}

template <int DH, int P>
int launch_attn2(const AttnParams& p, hipStream_t s) {
    using Cfg = Attn2Cfg<DH>;
    auto kfn = attn2_kernel<DH, P, (DH <= 40 ? 3 : DH <= 80 ? 2 : 1)>;
    static unsigned long long attr_done = 0;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), Cfg::SMEM, &attr_done)) return 1;
    dim3 grid((p.Nq + 127) / 128, p.B * p.heads);
    hipLaunchKernelGGL(kfn, grid, dim3(256), Cfg::SMEM, s, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <int P, int DH>
int launch_dh(const AttnParams& p, hipStream_t s) {
    using Cfg = AttnCfg<P, DH>;
    auto kfn = attn_kernel<P, DH>;
    static unsigned long long attr_done = 0;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), Cfg::SMEM, &attr_done)) return 1;
    dim3 grid((p.Nq + 127) / 128, p.B * p.heads);
    hipLaunchKernelGGL(kfn, grid, dim3(256), Cfg::SMEM, s, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <int P>
int launch_prec(const AttnParams& p, hipStream_t s) {
    if constexpr (!prec_f32_storage(P)) {
        if (!p.legacy && !p.causal) {
            switch (p.dh) {
                case 8: return launch_attn2<8, P>(p, s);
                case 16: return launch_attn2<16, P>(p, s);
                case 32: return launch_attn2<32, P>(p, s);
                case 40: return launch_attn2<40, P>(p, s);
                case 64: return launch_attn2<64, P>(p, s);
                case 80: return launch_attn2<80, P>(p, s);
                case 160: return launch_attn2<160, P>(p, s);   // the 16x16 / 8x8 levels (round 4)
                default: break;
            }
        }
    }
    switch (p.dh) {
        case 8: return launch_dh<P, 8>(p, s);
        case 16: return launch_dh<P, 16>(p, s);
        case 32: return launch_dh<P, 32>(p, s);
        case 40: return launch_dh<P, 40>(p, s);
        case 64: return launch_dh<P, 64>(p, s);
        case 80: return launch_dh<P, 80>(p, s);
        case 160: return launch_dh<P, 160>(p, s);
        default: return 2;
    }
}

}  // namespace

int launch_attention(const AttnParams& p, int prec, hipStream_t s) {
    if (p.Nq <= 0 || p.Nk <= 0) return 0;
    switch (prec) {
        case DT_F32: return launch_prec<DT_F32>(p, s);
        case PREC_F16X2: return launch_prec<PREC_F16X2>(p, s);
        case DT_BF16: return launch_prec<DT_BF16>(p, s);
        case DT_F16: return launch_prec<DT_F16>(p, s);
        default: return 1;
    }
}
