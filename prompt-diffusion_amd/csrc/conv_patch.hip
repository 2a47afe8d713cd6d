// conv3x3 (stride 1, pad 1, optional fused nearest-x2 upsample) with an LDS-staged input patch, gfx950.
//
// The generic implicit GEMM (gemm.hip) re-gathers the activation tile from L2/HBM for each of the 9 taps
// and is bound by operand traffic (~70 FLOP per byte fetched) and by the first-touch latency of those
// gathers.  Here a block owns a 16x16 patch of output pixels of one sample (M = 256) x 160 output
// channels.  Per 128-byte channel chunk it stages the (16+2)x(16+2) input patch in LDS ONCE and runs all
// 9 taps against it, streaming only the [160 x 128 B] weight tile per tap (always L2-resident: every block
// walks the same weights).  ~3x fewer bytes fetched per FLOP; the patch of chunk c+1 is requested at
// taps 0/3 of chunk c (two halves through the same registers) and lands in LDS at taps 3/6, so its
// HBM latency hides under 3 taps of MFMAs; the next tap's weight tile is requested at the top of each tap.
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

__device__ __forceinline__ int swzp(int row, int chunk) { return (row * ROWB) + (((chunk ^ (row >> 1)) & 7) << 4); }

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

template <bool F32, int UPS>
__global__ __launch_bounds__(NT) void conv3x3_patch_kernel(GemmParams p) {
    using G = PatchGeom<UPS>;
    constexpr int EB = F32 ? 4 : 2;
    constexpr int VEC = 16 / EB;
    constexpr int BKE = ROWB / EB;
    constexpr int PW = G::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                    // [2][PROWS][128]
    char* sW = smem + 2 * G::P_BYTES;   // [2][160][128]

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
    unsigned p_off[2 * G::P_HALF];
#pragma unroll
    for (int j = 0; j < 2 * G::P_HALF; ++j) {
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
    const int nchunks = p.Cin / BKE;
    const int U = nchunks * 9;  // (chunk, tap) units; weights for unit u start at element (u%9)*Cin + (u/9)*BKE

    // staging registers as named scalars (hipcc leaves small indexed arrays captured by these lambdas in scratch)
    static_assert(G::P_HALF <= 3 && W_ITERS == 3, "staging code below is written for <= 3 pieces");
    uint4 pr0, pr1, pr2, wr0, wr1, wr2;

    // the patch is staged in two halves through the same registers
    auto load_patch = [&](int c, int half) __attribute__((always_inline)) {
        const char* base = Ab + (size_t)c * BKE * EB;
        pr0 = *reinterpret_cast<const uint4*>(base + (half ? p_off[G::P_HALF + 0] : p_off[0]));
        if constexpr (G::P_HALF > 1) pr1 = *reinterpret_cast<const uint4*>(base + (half ? p_off[G::P_HALF + 1] : p_off[1]));
        if constexpr (G::P_HALF > 2) pr2 = *reinterpret_cast<const uint4*>(base + (half ? p_off[G::P_HALF + 2] : p_off[2]));
    };
    auto store_piece = [&](char* d, int j, const uint4& r) __attribute__((always_inline)) {
        int lds; bool ok; unsigned off;
        patch_slot(j, lds, ok, off);
        const uint4 v = ok ? r : make_uint4(0, 0, 0, 0);
        if (lds >= 0) *reinterpret_cast<uint4*>(d + lds) = v;
    };
    auto store_patch = [&](int buf, int half) __attribute__((always_inline)) {
        char* d = sP + buf * G::P_BYTES;
        store_piece(d, half * G::P_HALF + 0, pr0);
        if constexpr (G::P_HALF > 1) store_piece(d, half * G::P_HALF + 1, pr1);
        if constexpr (G::P_HALF > 2) store_piece(d, half * G::P_HALF + 2, pr2);
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
    load_patch(0, 0);
    load_w(0, 0);
    store_patch(0, 0);
    load_patch(0, 1);
    store_w(0);
    store_patch(0, 1);
    __syncthreads();

    // one (chunk, tap) unit.  Weight tile of unit u sits in LDS buffer u&1; the tile of unit u+1 is requested
    // at the top of the unit and written to the other buffer (last read in unit u-1) after the MFMAs.
    auto unit = [&](auto TAPC, int c) __attribute__((always_inline)) {
        constexpr int tap = decltype(TAPC)::value;
        constexpr int ky = tap / 3, kx = tap % 3;
        const int u = c * 9 + tap;
        const int par = u & 1;
        if (tap == 0 && c + 1 < nchunks) load_patch(c + 1, 0);
        if (u + 1 < U) {
            const int c2 = tap + 1 >= 9 ? c + 1 : c;
            load_w(c2, u + 1 - c2 * 9);
        }
        const char* pa = sP + (c & 1) * G::P_BYTES;
        const char* wa = sW + par * W_TILE;
        // keep the per-tap fragment addresses out of loop-invariant hoisting (they would otherwise live in
        // registers for the whole kernel and spill the accumulators)
        int frv = fr;
        asm volatile("" : "+v"(frv));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int py = wm * 4 + m;
                const int prow = (((y0 - 1 + py + ky) >> UPS) - sy0) * PW + (((x0 - 1 + frv + kx) >> UPS) - sx0);
                af[m] = *reinterpret_cast<const uint4*>(pa + swzp(prow, ks * 4 + fq));
            }
            // (row>>1)&7 of a weight row wn*80 + n*16 + fr does not depend on n or wn: one base + n*2048
            const char* wrow = wa + swzp(wn * 80 + frv, ks * 4 + fq);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const uint4 wf = *reinterpret_cast<const uint4*>(wrow + n * 16 * ROWB);
#pragma unroll
                for (int m = 0; m < 4; ++m) mma<F32>(wf, af[m], acc[n][m]);
            }
        }
        if (u + 1 < U) store_w(par ^ 1);
        // next chunk's patch (two halves through the same registers) -> the other patch buffer, last read in chunk c-1
        if (tap == 3 && c + 1 < nchunks) { store_patch((c + 1) & 1, 0); load_patch(c + 1, 1); }
        if (tap == 6 && c + 1 < nchunks) store_patch((c + 1) & 1, 1);
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

    // ---- epilogue
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int oy = y0 + wm * 4 + m, ox = x0 + fr;
        const int tok = oy * p.Wout + ox;
        const int gm = sample * p.rows_per_sample + tok;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int gn = bn * BN + wn * 80 + n * 16 + fq * 4;
            if (gn >= p.N) continue;
            epilogue4(p, gm, gn, sample, tok, acc[n][m]);
        }
    }
}

template <bool F32, int UPS>
int launch_patch(const GemmParams& p, hipStream_t s) {
    using G = PatchGeom<UPS>;
    static bool attr_done = false;
    auto kfn = conv3x3_patch_kernel<F32, UPS>;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM) !=
            hipSuccess)
            return 1;
        attr_done = true;
    }
    const int mtiles = (p.M / (p.Hout * p.Wout)) * (p.Hout / TP) * (p.Wout / TP), ntiles = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kfn, dim3(mtiles * ntiles), dim3(NT), G::SMEM, s, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace

// number of blocks the patch kernel would launch, or 0 when the shape does not qualify
int conv_patch_tiles(const GemmParams& p, bool f32mode) {
    const int bke = f32mode ? 32 : 64;
    if (p.taps != 9 || p.stride != 1 || p.splitk > 1) return 0;
    if (p.Hout % TP || p.Wout % TP || p.Cin % bke || p.K != 9 * p.Cin || p.act == 2 || p.vt_begin < p.N) return 0;
    if (p.a_dt != (f32mode ? DT_F32 : DT_BF16) || p.a_silu) return 0;
    if ((p.Hin << p.ups) != p.Hout || (p.Win << p.ups) != p.Wout) return 0;
    return (p.M / (p.Hout * p.Wout)) * (p.Hout / TP) * (p.Wout / TP) * ((p.N + BN - 1) / BN);
}

int launch_conv_patch(const GemmParams& p, bool f32mode, hipStream_t s) {
    if (p.ups) return f32mode ? launch_patch<true, 1>(p, s) : launch_patch<false, 1>(p, s);
    return f32mode ? launch_patch<true, 0>(p, s) : launch_patch<false, 0>(p, s);
}
