// conv3x3 (stride 1, pad 1, optional fused nearest-x2 upsample) with an LDS-staged input patch -- second generation,
// 2-byte compute types (bf16 / fp16), gfx950.
//
// Same tiling as conv_patch.hip (a block owns a 16x16 patch of output pixels of one sample x 160 output channels; per
// 128-byte channel chunk the (16+2)^2 input patch sits in LDS once and all 9 taps run from it; the [160 x 128 B] weight
// tile of each (chunk, tap) unit streams from L2), but the 8 waves are SPECIALISED:
//   waves 0-3  (one per SIMD) compute: each owns 4 patch rows x all 160 channels = 4 x 10 MFMA tiles, 160 accumulator
//              registers, 80 v_mfma_f32_16x16x32 per unit.  They never touch global memory.
//   waves 4-7  (their SIMD partners) load: weight tile of unit u+2 and the next chunk's patch, global -> VGPR -> LDS.
// Round 2's counters on the first-generation kernel (all 8 waves doing both jobs in lockstep, one barrier per unit) put
// 42 % of every wave's life in s_waitcnt / s_barrier and the MFMA pipe at 44 %: both SIMD partners reach their LDS read
// burst, their MFMA burst and the barrier together, so nothing overlaps.  Here the MFMA pipe of a SIMD belongs to one wave
// whose own stream interleaves fragment reads with MFMAs, and the fragments of the NEXT half-unit are always requested
// before the current one's MFMAs (across the unit barrier too: three weight buffers make tile u+1 complete one barrier
// before unit u+1 starts), so the pipe does not drain at a barrier.
#include <type_traits>

#include "pd_common.h"
#include "pd_mma.h"

#ifndef PATCH2_DIAG
#define PATCH2_DIAG 0
#endif

namespace {

constexpr int TP = 16;             // patch is TP x TP output pixels
constexpr int BN = 160;
constexpr int NT = 512;            // threads: 4 compute waves + 4 loader waves
constexpr int NL = 256;            // loader threads
constexpr int ROWB = 128;          // bytes of K per LDS row (64 two-byte channels)
constexpr int BKE = 64;
constexpr int W_TILE = BN * ROWB;  // 20480
constexpr int W_IT = BN * 8 / NL;  // 5 16-byte pieces per loader thread and weight tile
constexpr int NWB = 3;             // weight tile buffers

__device__ __forceinline__ int swz2(int row, int chunk) { return (row * ROWB) + (((chunk ^ (row >> 1)) & 7) << 4); }

template <int UPS>
struct Geom2 {
    static constexpr int PW = UPS ? TP / 2 + 2 : TP + 2;   // patch rows/cols held in LDS (source resolution)
    static constexpr int PROWS = PW * PW;
    static constexpr int P_SLOTS = PROWS * 8;
    static constexpr int P_IT = (P_SLOTS + NL - 1) / NL;   // 11 (plain) / 4 (upsampling) pieces per loader thread
    static constexpr int PA = P_IT < 6 ? P_IT : 6;         // first batch (requested at tap 0, stored at tap 2)
    static constexpr int PB = P_IT - PA;                   // second batch (requested at tap 2, stored at tap 4)
    static constexpr int P_BYTES = PROWS * ROWB;
    static constexpr int SMEM = 2 * P_BYTES + NWB * W_TILE;
};

template <int P, int UPS>
__global__ __launch_bounds__(NT) void conv3x3_patch2_kernel(GemmParams p) {
    using G = Geom2<UPS>;
    constexpr int PW = G::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                    // [2][PROWS][128]
    char* sW = smem + 2 * G::P_BYTES;   // [3][160][128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    const int ptx = p.Wout / TP, pty = p.Hout / TP;
    const int mtiles = (p.M / (p.Hout * p.Wout)) * ptx * pty, ntiles = (p.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (gemm.hip)
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int bm = bid / ntiles, bn = bid % ntiles;
    const int sample = bm / (ptx * pty);
    const int prem = bm - sample * (ptx * pty);
    const int y0 = (prem / ptx) * TP, x0 = (prem - (prem / ptx) * ptx) * TP;  // patch origin (output coords)
    const int sy0 = (y0 - 1) >> UPS, sx0 = (x0 - 1) >> UPS;                   // source-resolution origin of LDS patch index 0

    // split-K (blockIdx.y): this slice owns the channel chunks [c0, c0 + nchunks)
    const int chunks_all = p.Cin / BKE;
    int c0 = 0, nchunks = chunks_all;
    if (p.splitk > 1) {
        const int per = (chunks_all + p.splitk - 1) / p.splitk;
        c0 = blockIdx.y * per;
        nchunks = min(chunks_all, c0 + per) - c0;
    }
    const int U = nchunks * 9;  // (chunk, tap) units; weights of unit (lc, tap) start at element tap*Cin + (c0+lc)*BKE

    if (wave >= 4) {
        // =============================================================== loader waves
        const int lt = tid - NL;
        const char* Ab = reinterpret_cast<const char*>(p.A);
        const char* Wb = reinterpret_cast<const char*>(p.W);
        unsigned w_off[W_IT];
        int w_lds[W_IT];
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int s = lt + NL * j;
            const int row = s >> 3, ch = s & 7;
            int n = bn * BN + row;
            n = n < p.N ? n : p.N - 1;      // channels >= N are never stored
            w_off[j] = (unsigned)(((size_t)n * p.Kpad + ch * 8) * 2);
            w_lds[j] = swz2(row, ch);
        }
        // patch slot j of this thread: LDS offset (-1: none), in-image flag, byte offset of channel chunk 0 in A
        auto patch_slot = [&](int j, int& lds, bool& ok, unsigned& off) __attribute__((always_inline)) {
            const int s = lt + NL * j;
            const int prow = s >> 3, ch = s & 7;
            const int iy = prow / PW, ix = prow - iy * PW;
            const int gy = sy0 + iy, gx = sx0 + ix;
            ok = s < G::P_SLOTS && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
            off = ok ? (unsigned)((((size_t)(sample * p.Hin + gy) * p.Win + gx) * p.lda + ch * 8) * 2) : 0u;
            lds = s < G::P_SLOTS ? swz2(prow, ch) : -1;
        };
        unsigned p_off[G::P_IT];
#pragma unroll
        for (int j = 0; j < G::P_IT; ++j) {
            int lds; bool ok;
            patch_slot(j, lds, ok, p_off[j]);
        }
        // staging registers are NAMED scalars (hipcc leaves indexed uint4 arrays of such loops in scratch memory)
        uint4 wr0, wr1, wr2, wr3, wr4, pa0, pa1, pa2, pa3, pa4, pa5, pb0, pb1, pb2, pb3, pb4;
#define PD_W5(X) X(0, wr0) X(1, wr1) X(2, wr2) X(3, wr3) X(4, wr4)
#define PD_PA(X) X(0, pa0) X(1, pa1) X(2, pa2) X(3, pa3) X(4, pa4) X(5, pa5)
#define PD_PB(X) X(0, pb0) X(1, pb1) X(2, pb2) X(3, pb3) X(4, pb4)
#define W_LD(j, r) r = *reinterpret_cast<const uint4*>(bw + w_off[j]);
#define W_ST(j, r) *reinterpret_cast<uint4*>(dw + w_lds[j]) = r;
#define PA_LD(j, r) if constexpr (j < G::PA) r = *reinterpret_cast<const uint4*>(an + p_off[j]);
#define PB_LD(j, r) if constexpr (j < G::PB) r = *reinterpret_cast<const uint4*>(an + p_off[G::PA + j]);
#define P_ST(jj, r)                                                       \
    {                                                                     \
        int lds; bool ok; unsigned off;                                   \
        patch_slot(jj, lds, ok, off);                                     \
        uint4 v = r;                                                      \
        if (!ok) v = make_uint4(0, 0, 0, 0);                              \
        if (lds >= 0) *reinterpret_cast<uint4*>(pn + lds) = v;            \
    }
#define PA_ST(j, r) if constexpr (j < G::PA) P_ST(j, r)
#define PB_ST(j, r) if constexpr (j < G::PB) P_ST(G::PA + j, r)
        // weight tile of unit v (clamped to the last unit: requests stay unconditional, see conv_patch.hip)
        auto w_base = [&](int v) __attribute__((always_inline)) {
            v = v < U ? v : U - 1;
            const int lc = v / 9, tap = v - lc * 9;
            return Wb + ((size_t)tap * p.Cin + (size_t)(c0 + lc) * BKE) * 2;
        };
        // ---- prologue: patch of chunk c0 and the weight tiles of units 0 and 1
        {
            const char* an = Ab + (size_t)c0 * BKE * 2;
            char* pn = sP;
            PD_PA(PA_LD) PD_PB(PB_LD)
            const char* bw = w_base(0);
            char* dw = sW;
            PD_W5(W_LD)
            PD_PA(PA_ST) PD_PB(PB_ST)
            PD_W5(W_ST)
            bw = w_base(1);
            dw = sW + W_TILE;
            PD_W5(W_LD)
            PD_W5(W_ST)
        }
        __syncthreads();
        int wbuf = 2;   // buffer of tile u + 2
        for (int lc = 0; lc < nchunks; ++lc) {
            const bool nextc = lc + 1 < nchunks;
            const char* an = Ab + (size_t)(c0 + (nextc ? lc + 1 : lc)) * BKE * 2;
            char* pn = sP + ((lc + 1) & 1) * G::P_BYTES;   // next chunk's patch buffer (last read in chunk lc-1)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int u = lc * 9 + tap;
#if PATCH2_DIAG == 1   /* timing diagnostic: loaders only keep the barrier count */
                __syncthreads();
                continue;
#endif
                // weight tile of unit u+2 first (L2), HBM-latency patch pieces after it: vmcnt retires in issue order
                const char* bw = w_base(u + 2);
                PD_W5(W_LD)
                if (tap == 0) { PD_PA(PA_LD) }
                if (tap == 2) {
                    if (nextc) { PD_PA(PA_ST) }
                    PD_PB(PB_LD)
                }
                if (tap == 4 && nextc) { PD_PB(PB_ST) }
                // tile u+2 -> buffer (u+2) % 3, last read by the compute waves in unit u-1
                char* dw = sW + wbuf * W_TILE;
                PD_W5(W_ST)
                wbuf = wbuf == NWB - 1 ? 0 : wbuf + 1;
                __syncthreads();
            }
        }
#undef PD_W5
#undef PD_PA
#undef PD_PB
#undef W_LD
#undef W_ST
#undef PA_LD
#undef PB_LD
#undef P_ST
#undef PA_ST
#undef PB_ST
        return;
    }

    // =================================================================== compute waves
    __builtin_amdgcn_s_setprio(1);
    const int wm = wave;   // patch rows wm*4 .. wm*4+3
    f32x4 acc[10][4];
#pragma unroll
    for (int n = 0; n < 10; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A fragment of patch row py = wm*4+m for tap (ky,kx), k-step ks: 16 consecutive LDS patch rows (16 x of one patch row)
    auto af_off = [&](int m, int ky, int kx, int ks, int frv) __attribute__((always_inline)) {
        const int py = wm * 4 + m;
        const int prow = (((y0 - 1 + py + ky) >> UPS) - sy0) * PW + (((x0 - 1 + frv + kx) >> UPS) - sx0);
        return swz2(prow, ks * 4 + fq);
    };
    // W fragment of channel tile n: (row>>1)&7 of row n*16+fr does not depend on n -> one base + n*2048
    auto wf_off = [&](int ks, int frv) __attribute__((always_inline)) { return swz2(frv, ks * 4 + fq); };

    uint4 afA[4], afB[4];   // A fragments of the ks = 0 / ks = 1 half-unit
    // weight fragments: a ring of 4, requested 3 MFMA groups (192 MFMA cycles) ahead of their use -- a ds_read_b128 issued
    // one group ahead (64 cycles) still exposes most of its latency.  Fragment g of a unit: ks = g / 10, n = g % 10.
    uint4 w0, w1, w2, w3;
    auto WR = [&](auto G4) __attribute__((always_inline)) -> uint4& {
        constexpr int r = decltype(G4)::value & 3;
        if constexpr (r == 0) return w0; else if constexpr (r == 1) return w1; else if constexpr (r == 2) return w2; else return w3;
    };

    __syncthreads();        // prologue of the loader waves: patch c0 and weight tiles 0, 1 are in LDS
    {
        int frv = fr;
        asm volatile("" : "+v"(frv));
#pragma unroll
        for (int m = 0; m < 4; ++m) afA[m] = *reinterpret_cast<const uint4*>(sP + af_off(m, 0, 0, 0, frv));
        const int wo0 = wf_off(0, frv);
        w0 = *reinterpret_cast<const uint4*>(sW + wo0);
        w1 = *reinterpret_cast<const uint4*>(sW + wo0 + 1 * 16 * ROWB);
        w2 = *reinterpret_cast<const uint4*>(sW + wo0 + 2 * 16 * ROWB);
    }
    int wbuf = 0;
    auto unit = [&](auto TAPC, int lc) __attribute__((always_inline)) {
        constexpr int tap = decltype(TAPC)::value;
        constexpr int ky = tap / 3, kx = tap % 3;
        constexpr int ntap = tap == 8 ? 0 : tap + 1, nky = ntap / 3, nkx = ntap % 3;
        const int u = lc * 9 + tap;
        const char* pa = sP + (lc & 1) * G::P_BYTES;
        const char* wa = sW + wbuf * W_TILE;
        const int nwbuf = wbuf == NWB - 1 ? 0 : wbuf + 1;
        const char* pan = sP + ((tap == 8 ? lc + 1 : lc) & 1) * G::P_BYTES;   // patch / weight buffers of unit u+1
        const char* wan = sW + nwbuf * W_TILE;
        const bool more = u + 1 < U;
        // keep the per-tap fragment addresses out of loop-invariant hoisting (they would take the accumulators' registers)
        int frv = fr;
        asm volatile("" : "+v"(frv));
        const int wo0 = wf_off(0, frv), wo1 = wf_off(1, frv);
        // fragments afA and ring slots 0..2 were requested during the previous unit.  The first fragments of unit u+1 are
        // requested in this unit's second half, BEFORE the barrier (its weight tile was completed one barrier ago, its
        // patch -- another buffer only at tap 8 -- at tap 4).
        // requests are unconditional (the last unit re-reads valid LDS of its own buffers): a branch would split the
        // scheduling region that the sched_group_barrier pattern below lays out
        const char* pan2 = more ? pan : pa;
        const char* wan2 = more ? wan : wa;
        static_for<20>([&](auto GI) __attribute__((always_inline)) {
            constexpr int g = decltype(GI)::value;
            constexpr int ks = g / 10, n = g % 10;
            constexpr int g3 = g + 3;            // fragment requested now
            if constexpr (g3 < 20) {
                constexpr int ks3 = g3 / 10, n3 = g3 % 10;
                WR(std::integral_constant<int, g3>{}) = *reinterpret_cast<const uint4*>(wa + (ks3 ? wo1 : wo0) + n3 * 16 * ROWB);
            } else {
                WR(std::integral_constant<int, g3>{}) = *reinterpret_cast<const uint4*>(wan2 + wo0 + (g3 - 20) * 16 * ROWB);
            }
            if constexpr (g < 4) afB[g] = *reinterpret_cast<const uint4*>(pa + af_off(g, ky, kx, 1, frv));
            if constexpr (g >= 12 && g < 16) afA[g - 12] = *reinterpret_cast<const uint4*>(pan2 + af_off(g - 12, nky, nkx, 0, frv));
            const uint4 wc = WR(GI);
#if PATCH2_DIAG != 2   /* 2: timing diagnostic, compute waves issue no MFMAs */
#pragma unroll
            for (int m = 0; m < 4; ++m) mma<P>(wc, ks ? afB[m] : afA[m], acc[n][m]);
#else
            acc[n][0][0] += __uint_as_float(wc.x);
#endif
        });
        // pin the interleave: per MFMA group of 4, the LDS reads issued in front of it (1 weight fragment 3 groups ahead,
        // plus one A fragment in groups 0-3 and 12-15).  hipcc otherwise sinks every read next to its first use (it is at
        // the VGPR cap) and the MFMA pipe idles for an LDS latency every 8 MFMAs.
        static_for<20>([&](auto GI) __attribute__((always_inline)) {
            constexpr int g = decltype(GI)::value;
            __builtin_amdgcn_sched_group_barrier(0x100, (g < 4 || (g >= 12 && g < 16)) ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        });
        wbuf = nwbuf;
        __syncthreads();
    };
    for (int lc = 0; lc < nchunks; ++lc) {
        unit(std::integral_constant<int, 0>{}, lc);
        unit(std::integral_constant<int, 1>{}, lc);
        unit(std::integral_constant<int, 2>{}, lc);
        unit(std::integral_constant<int, 3>{}, lc);
        unit(std::integral_constant<int, 4>{}, lc);
        unit(std::integral_constant<int, 5>{}, lc);
        unit(std::integral_constant<int, 6>{}, lc);
        unit(std::integral_constant<int, 7>{}, lc);
        unit(std::integral_constant<int, 8>{}, lc);
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue (split-K: this slice's fp32 partial goes to its slab; splitk_finalize_kernel sums and finishes)
    float* slab = p.splitk > 1 ? reinterpret_cast<float*>(p.slab) + (size_t)blockIdx.y * p.M * p.N : nullptr;
    if (!slab) {   // pass 1: every read of the epilogue before the first store (pd_mma.h epilogue4_value)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int gm = sample * p.rows_per_sample + (y0 + wm * 4 + m) * p.Wout + x0 + fr;
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                const int gn = min(bn * BN + n * 16 + fq * 4, p.N - 4);
                acc[n][m] = epilogue4_value(p, gm, gn, sample, acc[n][m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int oy = y0 + wm * 4 + m, ox = x0 + fr;
        const int tok = oy * p.Wout + ox;
        const int gm = sample * p.rows_per_sample + tok;
#pragma unroll
        for (int n = 0; n < 10; ++n) {
            const int gn = bn * BN + n * 16 + fq * 4;
            if (gn >= p.N) continue;
            if (slab) *reinterpret_cast<f32x4*>(slab + (size_t)gm * p.N + gn) = acc[n][m];
            else epilogue4_store(p, gm, gn, sample, tok, acc[n][m]);
        }
    }
}

template <int P, int UPS>
int launch_patch2(const GemmParams& p, hipStream_t s) {
    using G = Geom2<UPS>;
    static unsigned long long attr_done = 0;
    auto kfn = conv3x3_patch2_kernel<P, UPS>;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), G::SMEM, &attr_done)) return 1;
    const int mtiles = (p.M / (p.Hout * p.Wout)) * (p.Hout / TP) * (p.Wout / TP), ntiles = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kfn, dim3(mtiles * ntiles, p.splitk > 1 ? p.splitk : 1), dim3(NT), G::SMEM, s, p);
    if (hipGetLastError() != hipSuccess) return 1;
    return (p.splitk > 1 && !p.defer_finalize) ? launch_splitk_finalize(p, s) : 0;
}

}  // namespace

// same eligibility as conv_patch_tiles(); 2-byte compute types only, no fused GroupNorm
int launch_conv_patch2(const GemmParams& p, int prec, hipStream_t s) {
    if (p.gn_coef || (prec != DT_BF16 && prec != DT_F16)) return 1;
    if (prec == DT_F16) return p.ups ? launch_patch2<DT_F16, 1>(p, s) : launch_patch2<DT_F16, 0>(p, s);
    return p.ups ? launch_patch2<DT_BF16, 1>(p, s) : launch_patch2<DT_BF16, 0>(p, s);
}
