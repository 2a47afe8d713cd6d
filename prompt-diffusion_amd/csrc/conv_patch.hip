// conv3x3 (stride 1, pad 1, optional fused nearest-x2 upsample) with an LDS-staged input patch, gfx950.
//
// The generic implicit GEMM (gemm.hip) re-gathers the activation tile from L2/HBM for each of the 9 taps
// and is bound by operand traffic (~70 FLOP per byte fetched) and by the first-touch latency of those
// gathers.  Here a block owns a 16x16 patch of output pixels of one sample (M = 256) x 160 output
// channels.  Per 128-byte channel chunk it stages the (16+2)x(16+2) input patch in LDS ONCE and runs all
// 9 taps against it, streaming only the [160 x 128 B] weight tile per tap (always L2-resident: every block
// walks the same weights).  ~3x fewer bytes fetched per FLOP; the patch of chunk c+1 is requested at
// piece by piece during taps 0-4 of chunk c and lands in LDS at taps 2-7 (one 16-byte piece per thread per tap),
// so its HBM latency hides under >= 2 taps of MFMAs; the next tap's weight tile is requested at the top of each tap.
//
// 512 threads = 8 waves as 4 (patch rows) x 2 (80 channels); a wave owns 4 patch rows x 80 channels =
// 4 x 5 MFMA tiles of 16x16, i.e. 40 v_mfma_f32_16x16x32_bf16 per tap and chunk.  An MFMA tile's 16
// pixels are 16 consecutive x of one patch row, so its fragment read for tap (ky,kx) is 16 consecutive
// rows of the LDS patch starting at (py+ky)*PW + kx: the same XOR swizzle as gemm.hip is conflict-free
// for any start row.  Swapped operands / 4-channel-per-lane epilogue / fp32 mode as in gemm.hip.
#include <type_traits>

#include "pd_common.h"
#include "pd_mma.h"

namespace {

constexpr int TP = 16;             // patch is TP x TP output pixels
constexpr int BN = 160;
constexpr int NT = 512;            // threads
constexpr int ROWB = 128;          // bytes of K per LDS row
constexpr int W_TILE = BN * ROWB;  // 20480
constexpr int W_SLOTS = BN * 8;    // 16-byte chunks of a weight tile
constexpr int W_ITERS = (W_SLOTS + NT - 1) / NT;  // 3 (last one half masked)

// LDS rows are 128 B = 8 chunks of 16 B; a row's chunks are permuted by XORing bits 1-2 of the chunk index with (row >> 1) & 3.
// A ds_read_b128 is served in four groups of 16 lanes, and a group of this kernel's fragment reads is 16 CONSECUTIVE rows of
// which 8 read chunk c and 8 chunk c + 1 (c even); bit 0 is left alone, so the two halves never meet, and inside a half the four
// rows of either parity differ in (row >> 1) & 3 -- conflict-free for ANY start row.  (gemm.hip's swizzle XORs all three bits with
// (row >> 1) & 7: conflict-free for start rows that are multiples of 16, as in a GEMM tile and the weight tile here, but 2-way
// conflicts on two patch reads in three -- the start row moves with the tap: 6.7 instead of 4 LDS cycles per read on average.
// conv_patch2.hip keeps the old form: with its loader / compute wave split the new one measured 0.5 % slower end to end.)
__device__ __forceinline__ int swzp(int row, int chunk) { return (row * ROWB) + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

template <int UPS>
struct PatchGeom {
    static constexpr int PW = UPS ? TP / 2 + 2 : TP + 2;   // patch rows/cols held in LDS (source resolution)
    static constexpr int PROWS = PW * PW;
    static constexpr int P_SLOTS = PROWS * 8;
    static constexpr int P_ITERS = (P_SLOTS + NT - 1) / NT;
    static constexpr int P_HALF = (P_ITERS + 1) / 2;   // pieces staged per half (registers are reused)
    static constexpr int P_BYTES = PROWS * ROWB;
    static constexpr int SMEM = 2 * P_BYTES + 2 * W_TILE;
};

template <int P, int UPS, bool GN>
__global__ __launch_bounds__(NT) void conv3x3_patch_kernel(GemmParams p) {
    using G = PatchGeom<UPS>;
    constexpr bool F32 = prec_f32_storage(P);
    constexpr int EB = F32 ? 4 : 2;
    constexpr int VEC = 16 / EB;
    constexpr int BKE = ROWB / EB;
    constexpr int PW = G::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                    // [2][PROWS][128]
    char* sW = smem + 2 * G::P_BYTES;   // [2][160][128]
    float* sCoef = reinterpret_cast<float*>(smem + G::SMEM);  // GN: [Cin][2] coefficients of this block's sample

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    const int ptx = p.Wout / TP, pty = p.Hout / TP;
    const int mtiles = (p.M / (p.Hout * p.Wout)) * ptx * pty, ntiles = (p.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int bm = bid / ntiles, bn = bid % ntiles;
    const int sample = bm / (ptx * pty);
    const int prem = bm - sample * (ptx * pty);
    const int y0 = (prem / ptx) * TP, x0 = (prem - (prem / ptx) * ptx) * TP;  // patch origin (output coords)
    // source-resolution origin of the staged patch (row/col of LDS patch index 0)
    const int sy0 = (y0 - 1) >> UPS, sx0 = (x0 - 1) >> UPS;

    // ---- staging assignments (32-bit byte offsets; masks and LDS addresses of the patch are recomputed
    //      at store time, once per chunk, to keep registers for the accumulators)
    auto patch_slot = [&](int j, int& lds, bool& ok, unsigned& off) __attribute__((always_inline)) {
        const int s = tid + NT * j;
        const int prow = s >> 3, ch = s & 7;
        const int iy = prow / PW, ix = prow - iy * PW;
        const int gy = sy0 + iy, gx = sx0 + ix;
        ok = s < G::P_SLOTS && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
        off = ok ? (unsigned)((((size_t)(sample * p.Hin + gy) * p.Win + gx) * p.lda + ch * VEC) * EB) : 0u;
        lds = s < G::P_SLOTS ? swzp(prow, ch) : -1;
    };
    static_assert(G::P_ITERS <= 6, "piece schedule covers 6 pieces");
    unsigned p_off[G::P_ITERS];
#pragma unroll
    for (int j = 0; j < G::P_ITERS; ++j) {
        int lds; bool ok;
        patch_slot(j, lds, ok, p_off[j]);
    }
    unsigned w_off[W_ITERS];
    int w_lds[W_ITERS];
#pragma unroll
    for (int j = 0; j < W_ITERS; ++j) {
        const int s = tid + NT * j;
        const int row = s >> 3, ch = s & 7;
        int n = bn * BN + row;
        n = n < p.N ? n : p.N - 1;
        w_off[j] = (unsigned)(((size_t)n * p.Kpad + ch * VEC) * EB);
        w_lds[j] = s < W_SLOTS ? swzp(row, ch) : -1;
    }
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    // split-K (blockIdx.y): this slice owns the channel chunks [c0, c0 + nchunks); units are counted from the slice's start
    const int chunks_all = p.Cin / BKE;
    int c0 = 0, nchunks = chunks_all;
    if (p.splitk > 1) {
        const int per = (chunks_all + p.splitk - 1) / p.splitk;
        c0 = blockIdx.y * per;
        nchunks = min(chunks_all, c0 + per) - c0;
    }
    const int U = nchunks * 9;  // (chunk, tap) units of this slice; weights of a unit start at element tap*Cin + chunk*BKE

    // staging registers as named scalars (hipcc leaves small indexed arrays captured by these lambdas in scratch)
    static_assert(W_ITERS == 3, "weight staging below is written for 3 pieces");
    uint4 pr0, pr1, pr2, wr0, wr1, wr2;

    // The patch of the next chunk is staged piece by piece through three registers: piece j is requested at tap
    // (j < 3 ? 0 : j - 1) and written to LDS at tap j + 2, so at most one piece is converted/stored per tap and every
    // request has >= 2 taps of MFMAs to land.
    auto PR = [&](auto J) __attribute__((always_inline)) -> uint4& {
        constexpr int r = decltype(J)::value % 3;
        if constexpr (r == 0) return pr0; else if constexpr (r == 1) return pr1; else return pr2;
    };
    auto load_piece = [&](auto J, int c) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < G::P_ITERS) PR(J) = *reinterpret_cast<const uint4*>(Ab + (size_t)c * BKE * EB + p_off[j]);
    };
    // GroupNorm(+SiLU) fused into the staging: y = silu(x * a[c] + b[c]); pad positions stay 0 (the reference
    // pads the normalised tensor).  A thread's 16-byte piece always covers channels chunk*BKE + (tid&7)*VEC ..
    auto store_piece = [&](char* d, int j, const uint4& r, int c) __attribute__((always_inline)) {
        int lds; bool ok; unsigned off;
        patch_slot(j, lds, ok, off);
        uint4 v = r;
        if constexpr (GN) {
            const float* cf = sCoef + (size_t)(c * BKE + (tid & 7) * VEC) * 2;
            float f[VEC];
            if constexpr (F32) {
                const float* t = reinterpret_cast<const float*>(&r);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = t[e];
            } else {
                unpack8<P>(r, f);
            }
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                const f32x4 ab = *reinterpret_cast<const f32x4*>(cf + 2 * e);   // a[e], b[e], a[e+1], b[e+1]
                float y0 = fmaf(f[e], ab[0], ab[1]), y1 = fmaf(f[e + 1], ab[2], ab[3]);
                if (p.gn_silu) { y0 = silu_f(y0); y1 = silu_f(y1); }
                f[e] = y0; f[e + 1] = y1;
            }
            if constexpr (F32) {
                float* t = reinterpret_cast<float*>(&v);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = f[e];
            } else {
                v = pack8<P>(f);
            }
        }
        if (!ok) v = make_uint4(0, 0, 0, 0);
        if (lds >= 0) *reinterpret_cast<uint4*>(d + lds) = v;
    };
    auto store_one = [&](auto J, int buf, int c) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < G::P_ITERS) store_piece(sP + buf * G::P_BYTES, j, PR(J), c);
    };
    auto load_w = [&](int c, int tap) __attribute__((always_inline)) {
        const char* base = Wb + ((size_t)tap * p.Cin + (size_t)c * BKE) * EB;
        wr0 = *reinterpret_cast<const uint4*>(base + w_off[0]);
        wr1 = *reinterpret_cast<const uint4*>(base + w_off[1]);
        wr2 = *reinterpret_cast<const uint4*>(base + w_off[2]);
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) {
        char* d = sW + buf * W_TILE;
        *reinterpret_cast<uint4*>(d + w_lds[0]) = wr0;
        *reinterpret_cast<uint4*>(d + w_lds[1]) = wr1;
        if (w_lds[2] >= 0) *reinterpret_cast<uint4*>(d + w_lds[2]) = wr2;
    };

    f32x4 acc[5][4];
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: patch 0, weight tile of unit 0 in LDS; unit 1's weights in flight
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    load_piece(I0{}, c0); load_piece(I1{}, c0); load_piece(I2{}, c0);
    load_w(c0, 0);
    if constexpr (GN) {
        const float* src = p.gn_coef + (size_t)sample * p.Cin * 2;
        for (int i = tid; i < p.Cin * 2; i += NT) sCoef[i] = src[i];
        __syncthreads();
    }
    store_one(I0{}, 0, c0); store_one(I1{}, 0, c0); store_one(I2{}, 0, c0);
    load_piece(I3{}, c0); load_piece(I4{}, c0); load_piece(I5{}, c0);
    store_w(0);
    store_one(I3{}, 0, c0); store_one(I4{}, 0, c0); store_one(I5{}, 0, c0);
    __syncthreads();

    // one (chunk, tap) unit.  Weight tile of unit u sits in LDS buffer u&1; the tile of unit u+1 is requested
    // at the top of the unit and written to the other buffer (last read in unit u-1) after the MFMAs.
    auto unit = [&](auto TAPC, int lc) __attribute__((always_inline)) {
        constexpr int tap = decltype(TAPC)::value;
        constexpr int ky = tap / 3, kx = tap % 3;
        const int c = c0 + lc;          // channel chunk (addresses); lc counts this slice's chunks (buffers, parity)
        const int u = lc * 9 + tap;
        const int par = u & 1;
        const bool nextc = lc + 1 < nchunks;
        // Request order matters: vmcnt retires in issue order, so the wait in front of store_w (bottom of the unit) also
        // waits for every load issued BEFORE this unit's load_w.  The weight tile (L2-resident) is requested first and the
        // HBM-latency patch pieces after it: a piece is then only waited for by the NEXT unit's store_w -- two units of
        // MFMAs to land instead of one (round 1 issued the pieces first / at the bottom and stalled on HBM in 4 taps of 9).
        // Every request below is UNCONDITIONAL (the last chunk / unit re-request addresses of their own that are never
        // stored): hipcc's waitcnt insertion merges the counter state of both sides of a branch around a load, and a load
        // that "may not have been issued" turns the counted vmcnt(N) in front of store_w into vmcnt(0) -- the stall again.
        const int cn = nextc ? c + 1 : c;
        load_w(tap + 1 >= 9 ? cn : c, tap + 1 >= 9 ? 0 : tap + 1);
        // next chunk's patch (two halves through the same registers) -> the other patch buffer, last read in chunk c-1
        {
            const int nb = (lc + 1) & 1;
            if constexpr (tap == 0) { load_piece(I0{}, cn); load_piece(I1{}, cn); load_piece(I2{}, cn); }
            if constexpr (tap == 2) { if (nextc) store_one(I0{}, nb, cn); load_piece(I3{}, cn); }
            if constexpr (tap == 3) { if (nextc) store_one(I1{}, nb, cn); load_piece(I4{}, cn); }
            if constexpr (tap == 4) { if (nextc) store_one(I2{}, nb, cn); load_piece(I5{}, cn); }
            if constexpr (tap == 5) { if (nextc) store_one(I3{}, nb, cn); }
            if constexpr (tap == 6) { if (nextc) store_one(I4{}, nb, cn); }
            if constexpr (tap == 7) { if (nextc) store_one(I5{}, nb, cn); }
        }
        const char* pa = sP + (lc & 1) * G::P_BYTES;
        const char* wa = sW + par * W_TILE;
        // keep the per-tap fragment addresses out of loop-invariant hoisting (they would otherwise live in
        // registers for the whole kernel and spill the accumulators)
        int frv = fr;
        asm volatile("" : "+v"(frv));
        if constexpr (P == PREC_F16X2) {   // both half-steps at once: 3 MFMAs per accumulator and unit (pd_mma.h)
            FragX2 af[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int py = wm * 4 + m;
                const int prow = (((y0 - 1 + py + ky) >> UPS) - sy0) * PW + (((x0 - 1 + frv + kx) >> UPS) - sx0);
                af[m] = prep_x2(*reinterpret_cast<const uint4*>(pa + swzp(prow, fq)), *reinterpret_cast<const uint4*>(pa + swzp(prow, 4 + fq)));
            }
            const char* w0 = wa + swzp(wn * 80 + frv, fq);
            const char* w1 = wa + swzp(wn * 80 + frv, 4 + fq);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const FragX2 wf = prep_x2(*reinterpret_cast<const uint4*>(w0 + n * 16 * ROWB), *reinterpret_cast<const uint4*>(w1 + n * 16 * ROWB));
#pragma unroll
                for (int m = 0; m < 4; ++m) mma_x2(wf, af[m], acc[n][m]);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            typename Frag<P>::A af[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int py = wm * 4 + m;
                const int prow = (((y0 - 1 + py + ky) >> UPS) - sy0) * PW + (((x0 - 1 + frv + kx) >> UPS) - sx0);
                af[m] = prep_a<P>(*reinterpret_cast<const uint4*>(pa + swzp(prow, ks * 4 + fq)));
            }
            // the swizzle term (row>>1)&3 of a weight row wn*80 + n*16 + fr does not depend on n or wn: one base + n*2048
            const char* wrow = wa + swzp(wn * 80 + frv, ks * 4 + fq);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const typename Frag<P>::W wf = prep_w<P>(*reinterpret_cast<const uint4*>(wrow + n * 16 * ROWB));
#pragma unroll
                for (int m = 0; m < 4; ++m) mma<P>(wf, af[m], acc[n][m]);
            }
        }
        if (u + 1 < U) store_w(par ^ 1);
        __syncthreads();

    };
    for (int c = 0; c < nchunks; ++c) {
        unit(std::integral_constant<int, 0>{}, c);
        unit(std::integral_constant<int, 1>{}, c);
        unit(std::integral_constant<int, 2>{}, c);
        unit(std::integral_constant<int, 3>{}, c);
        unit(std::integral_constant<int, 4>{}, c);
        unit(std::integral_constant<int, 5>{}, c);
        unit(std::integral_constant<int, 6>{}, c);
        unit(std::integral_constant<int, 7>{}, c);
        unit(std::integral_constant<int, 8>{}, c);
    }

    // ---- epilogue (split-K: this slice's fp32 partial goes to its slab; splitk_finalize_kernel sums and finishes)
    float* slab = p.splitk > 1 ? reinterpret_cast<float*>(p.slab) + (size_t)blockIdx.y * p.M * p.N : nullptr;
    if (!slab) {   // pass 1: every read of the epilogue before the first store (pd_mma.h epilogue4_value)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int gm = sample * p.rows_per_sample + (y0 + wm * 4 + m) * p.Wout + x0 + fr;
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const int gn = min(bn * BN + wn * 80 + n * 16 + fq * 4, p.N - 4);
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
        for (int n = 0; n < 5; ++n) {
            const int gn = bn * BN + wn * 80 + n * 16 + fq * 4;
            if (gn >= p.N) continue;
            if (slab) *reinterpret_cast<f32x4*>(slab + (size_t)gm * p.N + gn) = acc[n][m];
            else epilogue4_store(p, gm, gn, sample, tok, acc[n][m]);
        }
    }
}

constexpr int COEF_BYTES_MAX = 24 * 1024;   // [Cin <= 3072][2] floats behind the staging buffers

template <int P, int UPS, bool GN>
int launch_patch(const GemmParams& p, hipStream_t s) {
    using G = PatchGeom<UPS>;
    static unsigned long long attr_done = 0;
    auto kfn = conv3x3_patch_kernel<P, UPS, GN>;
    const int smem = G::SMEM + (GN ? COEF_BYTES_MAX : 0);
    if (GN && p.Cin * 8 > COEF_BYTES_MAX) return 1;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), smem, &attr_done)) return 1;
    const int mtiles = (p.M / (p.Hout * p.Wout)) * (p.Hout / TP) * (p.Wout / TP), ntiles = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kfn, dim3(mtiles * ntiles, p.splitk > 1 ? p.splitk : 1), dim3(NT), smem, s, p);
    if (hipGetLastError() != hipSuccess) return 1;
    return (p.splitk > 1 && !p.defer_finalize) ? launch_splitk_finalize(p, s) : 0;
}

}  // namespace

// number of blocks the patch kernel would launch, or 0 when the shape does not qualify
int conv_patch_tiles(const GemmParams& p, int prec) {
    const int bke = prec_f32_storage(prec) ? 32 : 64;
    if (p.taps != 9 || p.stride != 1) return 0;
    if (p.Hout % TP || p.Wout % TP || p.Cin % bke || p.K != 9 * p.Cin || p.act == 2 || p.vt_begin < p.N) return 0;
    if (p.a_dt != (prec_f32_storage(prec) ? (int)DT_F32 : prec) || p.a_silu) return 0;
    if (p.Cin * 8 > 24 * 1024) return 0;
    if ((p.Hin << p.ups) != p.Hout || (p.Win << p.ups) != p.Wout) return 0;
    return (p.M / (p.Hout * p.Wout)) * (p.Hout / TP) * (p.Wout / TP) * ((p.N + BN - 1) / BN);
}

namespace {
template <int P>
int launch_patch_prec(const GemmParams& p, hipStream_t s) {
    if (p.gn_coef) {
        if (p.ups) return 1;   // GroupNorm never feeds an upsampling conv in this network
        return launch_patch<P, 0, true>(p, s);
    }
    return p.ups ? launch_patch<P, 1, false>(p, s) : launch_patch<P, 0, false>(p, s);
}
}  // namespace

int launch_conv_patch(const GemmParams& p, int prec, hipStream_t s) {
    switch (prec) {
        case DT_F32: return launch_patch_prec<DT_F32>(p, s);
        case PREC_F16X2: return launch_patch_prec<PREC_F16X2>(p, s);
        case DT_BF16: return launch_patch_prec<DT_BF16>(p, s);
        case DT_F16: return launch_patch_prec<DT_F16>(p, s);
        default: return 1;
    }
}
