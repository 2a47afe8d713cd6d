// conv3x3 (stride 1, pad 1, optional fused nearest-x2 upsample) with an LDS-staged input patch -- third generation ("ping-pong"),
// 2-byte compute types (bf16 / fp16), gfx950.
//
// Same block tiling as conv_patch.hip (a block owns a 16x16 patch of output pixels of one sample x 160 output channels; per 128-byte
// channel chunk the (16+2)^2 input patch sits in LDS once and all 9 taps run from it; the [160 x 128 B] weight tile of each
// (chunk, tap) unit streams from L2), but
//   * nothing is staged through registers: weight tiles and patches travel L2/HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, the
//     lane fetches the 16-byte chunk that belongs in its swizzled slot), weights through a ring of 3 tiles two units ahead, the
//     next chunk's patch one piece per wave and unit; out-of-image pixels are lanes masked off over a patch buffer zeroed once
//     (the padding positions of a block are the same for every chunk);
//   * the 8 waves form two groups (waves 0-3: patch rows 0-7, waves 4-7: rows 8-15; every SIMD hosts one wave of each) that run
//     half a unit apart: a unit is a LOAD phase (all 24 fragment reads of the unit, the LDS-DMA requests, the counted wait for
//     the next unit's operands) and an MFMA phase (20 v_mfma_f32_32x32x16 back to back) with a barrier after each, and group 1
//     enters the loop one barrier late.  On every SIMD one wave owns the matrix pipe while its partner talks to the LDS and the
//     memory system.  (The first generation runs all 8 waves in lockstep -- request, read, MFMA, store, barrier -- and its stamps
//     put 2.6 k cycles into a unit whose MFMAs need 1.28 k: profiles/r03_conv_stamp.txt.)
//   * the MFMA is v_mfma_f32_32x32x16: ONE wave issues it at the matrix pipe's full rate (32 cycles each), which a single wave does
//     not reach with the 16x16x32 shape (29 cycles instead of 16: tools/micro/mfma_clock.hip; the first ping-pong build on 16x16x32
//     tiles ran a unit in 2.4 k cycles -- 2 x 40 x 29).  A wave owns 2 patch rows (32 pixels) x all 160 channels: 5 accumulator
//     tiles, 4 + 20 fragment reads and 20 MFMAs per unit; results differ from the 16x16x32 kernels in the last fp32 bits only
//     (another summation grouping of the same exact products).
// Ordering rules of the hand-off (cdna_hip_programming.md, "Read a staged buffer one phase after the wait that retires it"):
//   RAW  weights of unit u + 1 are waited for (counted vmcnt, every wave for its own pieces) at the end of load phase u of BOTH
//        groups; group 0 reads them in its load phase u + 1, which begins behind the barrier that ends group 1's load phase u.
//   WAR  the tile of unit u + 2 goes into the slot of unit u - 1, last read in group 1's load phase u - 1, whose reads are
//        complete (lgkmcnt(0)) before the barrier that every later request is behind.
#include <type_traits>

#include "pd_common.h"
#include "pd_mma.h"

namespace {

constexpr int TP = 16;             // patch is TP x TP output pixels
constexpr int BN = 160;
constexpr int NT = 512;            // threads
constexpr int ROWB = 128;          // bytes of K per LDS row (64 two-byte channels)
constexpr int BKE = 64;
constexpr int W_TILE = BN * ROWB;  // 20480
constexpr int W_PIECES = BN / 8;   // 20 LDS-DMA pieces (8 rows x 128 B) per weight tile
constexpr int NWB = 3;             // weight tile ring

// 16-byte chunk index ^= (row >> 1) & 7: a ds_read_b128 of 32 CONSECUTIVE rows at one chunk per lane half (the 32x32x16 operand) is
// conflict-free from any start row (each of the instruction's four 16-lane groups meets 8 rows of either parity whose (row >> 1) & 7
// are distinct); the pixel operand is two runs of 16 rows PW apart and pays one 2-way pair per group (4 of a unit's 24 reads)
__device__ __forceinline__ int swz3(int row, int chunk) { return (row * ROWB) + (((chunk ^ (row >> 1)) & 7) << 4); }
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int P> __device__ __forceinline__ void mma32(const uint4& w, const uint4& a, f32x16& acc) {
    if constexpr (P == DT_F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

template <int UPS>
struct Geom3 {
    static constexpr int PW = UPS ? TP / 2 + 2 : TP + 2;   // patch rows/cols held in LDS (source resolution)
    static constexpr int PROWS = PW * PW;
    static constexpr int P_PIECES = (PROWS + 7) / 8;       // 41 (plain) / 13 (upsampling)
    static constexpr int P_PER_WAVE = (P_PIECES + 7) / 8;  // 6 / 2: piece wave + 8 t is requested in the load phase of tap t
    static constexpr int P_BYTES = P_PIECES * 8 * ROWB;    // whole pieces (the last one's rows past PROWS are never written)
    static constexpr int SMEM = 2 * P_BYTES + NWB * W_TILE;
};

// LDS-DMA of 64 x 16 B (gemm_ring.hip): wave-uniform 64-bit base + 32-bit lane offset, LDS destination = M0 base + 16 * lane;
// lanes with EXEC off write nothing.  The compiler does not see the request; every wait on it is an explicit s_waitcnt below.
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* q) {   // provably wave-uniform for the "s" constraint
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const char*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ void wait_vm_n(int n) {   // n is wave-uniform and small
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Diagnostic build only (tools/micro/conv_pp_stamp.hip compiles this file with -DPD_STAMP): cycles waves 0 and 4 of every block spend in the
// parts of a unit.  No stamp executes in the product build.
#ifdef PD_STAMP
__device__ unsigned long long* g_pp_stamps = nullptr;
#define PT_NOW() ([]() { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); return t_; }())
#define PT_ADD(acc, a, b) acc += (b) - (a)
#else
#define PT_NOW() 0ull
#define PT_ADD(acc, a, b) do { } while (0)
#endif

template <int P, int UPS>
__global__ __launch_bounds__(NT) void conv3x3_pp_kernel(GemmParams p) {
    using G = Geom3<UPS>;
    constexpr int PW = G::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                    // [2][P_PIECES * 8][128]
    char* sW = smem + 2 * G::P_BYTES;   // [3][160][128]
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l31 = lane & 31, lh = lane >> 5;   // operand row / column of the 32x32x16 MFMA, K half

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

    // split-K (blockIdx.y): this slice owns the channel chunks [c0, c0 + nchunks); units are counted from the slice's start
    const int chunks_all = p.Cin / BKE;
    int c0 = 0, nchunks = chunks_all;
    if (p.splitk > 1) {
        const int per = (chunks_all + p.splitk - 1) / p.splitk;
        c0 = blockIdx.y * per;
        nchunks = min(chunks_all, c0 + per) - c0;
    }
    const int U = nchunks * 9;

    // ---- this lane's LDS-DMA sources.  A piece is 8 LDS rows; lane l writes row 8 * piece + (l >> 3), slot l & 7, and fetches the
    // logical chunk that the swizzle keeps in that slot.
    const int lrow = lane >> 3, lslot = lane & 7;
    // weights: pieces wave, wave + 8 and (waves 0-3) wave + 16; rows past N re-read the last row (never stored)
    const int kw = wave < W_PIECES - 16 ? 3 : 2;
    unsigned w_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = (wave + 8 * j) * 8 + lrow;
        const int chunk = (lslot ^ (row >> 1)) & 7;
        const int n = min(bn * BN + row, p.N - 1);
        w_off[j] = (unsigned)(((size_t)n * p.Kpad + chunk * 8) * 2);
    }
    // patch: pieces wave + 8 t, t < P_PER_WAVE; pixels outside the image (and rows past the patch) are lanes that stay off
    unsigned p_off[G::P_PER_WAVE];
    unsigned p_ok = 0, p_any = 0;   // bit t: this lane is on in piece t / some lane of the wave is (wave-uniform)
#pragma unroll
    for (int t = 0; t < G::P_PER_WAVE; ++t) {
        const int prow = (wave + 8 * t) * 8 + lrow;
        const int chunk = (lslot ^ (prow >> 1)) & 7;
        const int iy = prow / PW, ix = prow - iy * PW;
        const int gy = sy0 + iy, gx = sx0 + ix;
        const bool ok = prow < G::PROWS && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
        p_off[t] = ok ? (unsigned)((((size_t)(sample * p.Hin + gy) * p.Win + gx) * p.lda + chunk * 8) * 2) : 0u;
        p_ok |= ok ? 1u << t : 0u;
        p_any |= __builtin_amdgcn_ballot_w64(ok) != 0 ? 1u << t : 0u;
    }
    p_any = (unsigned)__builtin_amdgcn_readfirstlane((int)p_any);
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);

    auto issue_w = [&](int c, int tap, int slot) __attribute__((always_inline)) {   // weight tile of (chunk c, tap) -> ring slot
        const char* base = uniform_ptr(Wb + ((size_t)tap * p.Cin + (size_t)c * BKE) * 2);
        const unsigned dst = lds0 + 2 * G::P_BYTES + (unsigned)slot * W_TILE + (unsigned)wave * 8 * ROWB;
        glds16(base, w_off[0], (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
        glds16(base, w_off[1], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 64 * ROWB)));
        if (kw == 3) glds16(base, w_off[2], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 128 * ROWB)));
    };
    auto issue_p = [&](auto T, int c, int buf) __attribute__((always_inline)) {     // piece wave + 8 t of chunk c's patch
        constexpr int t = decltype(T)::value;
        if constexpr (t < G::P_PER_WAVE) {
            if (p_any >> t & 1) {   // (a piece entirely outside the image / the patch is not requested at all)
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)buf * G::P_BYTES + (unsigned)(wave + 8 * t) * 8 * ROWB));
                const char* base = uniform_ptr(Ab + (size_t)c * BKE * 2);
                if (p_ok >> t & 1) glds16(base, p_off[t], dst);
            }
        }
    };
    // 1 when this wave requests a patch piece in the load phase of tap t (one entry in its memory queue)
    auto has_p = [&](int t) __attribute__((always_inline)) -> int { return (t >= 0 && t < G::P_PER_WAVE) ? (int)(p_any >> t & 1) : 0; };

    f32x16 acc[5];   // 5 tiles of 32 channels x this wave's 32 pixels: lane = pixel l31, register r = channel (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // ---- prologue: both patch buffers zeroed (the padding), then patch c0 and the weight tiles of units 0 and 1 requested
    for (int i = tid * 16; i < 2 * G::P_BYTES; i += NT * 16) *reinterpret_cast<uint4*>(sP + i) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pp_barrier();
    static_for<G::P_PER_WAVE>([&](auto T) { issue_p(T, c0, 0); });   // (older than the weight tiles in the queue)
    issue_w(c0, 0, 0);
    if (U > 1) issue_w(c0, 1, 1);
    wait_vm_n(U > 1 ? kw : 0);   // everything but unit 1's tile has landed
    pp_barrier();
    if (grp) pp_barrier();       // group 1 runs one phase behind

    [[maybe_unused]] unsigned long long c_rd = 0, c_is = 0, c_wt = 0, c_b1 = 0, c_mm = 0, c_b2 = 0;
    [[maybe_unused]] const unsigned long long t_begin = PT_NOW();
    uint4 af[4], wf[4][5];
    auto unit = [&](auto TAPC, int lc) __attribute__((always_inline)) {
        constexpr int tap = decltype(TAPC)::value;
        constexpr int ky = tap / 3, kx = tap % 3;
        const int c = c0 + lc;
        const int u = lc * 9 + tap;
        const bool nextc = lc + 1 < nchunks;
        // ---- load phase: the unit's fragments (weight slot tap % 3: 9 units per chunk keep unit % 3 == tap % 3)
        [[maybe_unused]] const unsigned long long t0 = PT_NOW();
        {
            const char* pa = sP + (lc & 1) * G::P_BYTES;
            const char* wa = sW + (tap % NWB) * W_TILE;
            int lv = l31;
            asm volatile("" : "+v"(lv));   // keep the per-tap addresses out of loop-invariant hoisting (they would occupy registers for the whole kernel)
            // pixel l31 of this wave: patch row 2 * wave + (l31 >> 4), column l31 & 15
            const int prow = (((y0 - 1 + 2 * wave + (lv >> 4) + ky) >> UPS) - sy0) * PW + (((x0 - 1 + (lv & 15) + kx) >> UPS) - sx0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                af[ks] = *reinterpret_cast<const uint4*>(pa + swz3(prow, ks * 2 + lh));
                const char* wrow = wa + swz3(lv, ks * 2 + lh);   // (the swizzle term of row 32 n + l31 does not depend on n)
#pragma unroll
                for (int n = 0; n < 5; ++n) wf[ks][n] = *reinterpret_cast<const uint4*>(wrow + n * 32 * ROWB);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        [[maybe_unused]] const unsigned long long t1 = PT_NOW();
        // requests: the weight tile of unit u + 2 into the slot of unit u - 1, then the next chunk's patch piece of this tap.  The
        // counted wait retires this wave's pieces of unit u + 1 and leaves what is younger in flight: the previous unit's patch piece
        // (two units to land: it may come from HBM), this unit's weight pieces and patch piece
        int inflight = (nextc && tap > 0) ? has_p(tap - 1) : 0;
        if (u + 2 < U) {
            const int t2 = tap + 2;
            issue_w(t2 >= 9 ? c + 1 : c, t2 >= 9 ? t2 - 9 : t2, t2 % NWB);
            inflight += kw;
        } else {
            inflight = 0;   // the last two units: nothing younger than unit u + 1's tile
        }
        if (nextc && has_p(tap)) { issue_p(TAPC, c + 1, (lc + 1) & 1); ++inflight; }
        [[maybe_unused]] const unsigned long long t2 = PT_NOW();
        wait_vm_n(inflight);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        [[maybe_unused]] const unsigned long long t3 = PT_NOW();
        pp_barrier();
        [[maybe_unused]] const unsigned long long t4 = PT_NOW();
        // ---- MFMA phase
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int n = 0; n < 5; ++n) mma32<P>(wf[ks][n], af[ks], acc[n]);
#ifdef PD_STAMP
        asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[4]));   // the unit's MFMAs have completed before the stamp
#endif
        [[maybe_unused]] const unsigned long long t5 = PT_NOW();
        if (!(grp && u + 1 == U)) pp_barrier();   // (group 1 entered one barrier late and leaves one early)
        else __builtin_amdgcn_sched_barrier(0);
        [[maybe_unused]] const unsigned long long t6 = PT_NOW();
        PT_ADD(c_rd, t0, t1); PT_ADD(c_is, t1, t2); PT_ADD(c_wt, t2, t3); PT_ADD(c_b1, t3, t4); PT_ADD(c_mm, t4, t5); PT_ADD(c_b2, t5, t6);
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

    // ---- epilogue (split-K: this slice's fp32 partial goes to its slab; splitk_finalize_kernel or the consumer sums and finishes)
    [[maybe_unused]] const unsigned long long t_epi = PT_NOW();
    float* slab = p.splitk > 1 ? reinterpret_cast<float*>(p.slab) + (size_t)blockIdx.y * p.M * p.N : nullptr;
    // this lane's pixel and its 20 groups of 4 consecutive channels: tile n, group g -> channel 32 n + 8 g + 4 lh
    const int oy = y0 + 2 * wave + (l31 >> 4), ox = x0 + (l31 & 15);
    const int tok = oy * p.Wout + ox;
    const int gm = sample * p.rows_per_sample + tok;
    if (!slab) {   // pass 1: every read of the epilogue before the first store (pd_mma.h epilogue4_value)
#pragma unroll
        for (int n = 0; n < 5; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gn = min(bn * BN + n * 32 + g * 8 + lh * 4, p.N - 4);
                const f32x4 v = epilogue4_value(p, gm, gn, sample, f32x4{acc[n][4 * g], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]});
                acc[n][4 * g] = v[0]; acc[n][4 * g + 1] = v[1]; acc[n][4 * g + 2] = v[2]; acc[n][4 * g + 3] = v[3];
            }
    }
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int gn = bn * BN + n * 32 + g * 8 + lh * 4;
            if (gn >= p.N) continue;
            const f32x4 v = f32x4{acc[n][4 * g], acc[n][4 * g + 1], acc[n][4 * g + 2], acc[n][4 * g + 3]};
            if (slab) *reinterpret_cast<f32x4*>(slab + (size_t)gm * p.N + gn) = v;
            else epilogue4_store(p, gm, gn, sample, tok, v);
        }
#ifdef PD_STAMP
    if (lane == 0 && (wave & 3) == 0 && g_pp_stamps) {
        unsigned long long* o = g_pp_stamps + ((size_t)blockIdx.x * 2 + grp) * 12;
        const unsigned long long t_end = PT_NOW();
        o[0] = t_begin; o[1] = t_end; o[2] = c_rd; o[3] = c_is; o[4] = c_wt; o[5] = c_b1; o[6] = c_mm; o[7] = c_b2; o[8] = t_end - t_epi; o[9] = (unsigned long long)U;
    }
#endif
}

template <int P, int UPS>
int launch_pp(const GemmParams& p, hipStream_t s) {
    using G = Geom3<UPS>;
    static_assert(G::SMEM <= 160 * 1024, "LDS");
    static unsigned long long attr_done = 0;
    auto kfn = conv3x3_pp_kernel<P, UPS>;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), G::SMEM, &attr_done)) return 1;
    const int mtiles = (p.M / (p.Hout * p.Wout)) * (p.Hout / TP) * (p.Wout / TP), ntiles = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kfn, dim3(mtiles * ntiles, p.splitk > 1 ? p.splitk : 1), dim3(NT), G::SMEM, s, p);
    if (hipGetLastError() != hipSuccess) return 1;
    return (p.splitk > 1 && !p.defer_finalize) ? launch_splitk_finalize(p, s) : 0;
}

}  // namespace

// shapes conv_patch_tiles() accepts, 2-byte compute types, no fused GroupNorm; 32-bit byte offsets inside each operand
bool conv_patch3_eligible(const GemmParams& p, int prec) {
    if (prec != DT_F16 && prec != DT_BF16) return false;
    if (p.gn_coef || p.a_dt != prec || p.Cin % BKE || p.K != 9 * p.Cin) return false;
    const unsigned long long a_bytes = (unsigned long long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * (unsigned)p.lda * 2ull;
    const unsigned long long w_bytes = (unsigned long long)p.N * (unsigned)p.Kpad * 2ull;
    const unsigned long long c_bytes = (unsigned long long)p.M * (unsigned)(p.ldc > p.ldr ? p.ldc : p.ldr) * 4ull;
    return a_bytes < (1ull << 31) && w_bytes < (1ull << 32) && c_bytes < (1ull << 31);   // (buffer descriptors of conv_patch4.hip: 31-bit ranges)
}

int launch_conv_patch3(const GemmParams& p, int prec, hipStream_t s) {
    if (!conv_patch3_eligible(p, prec)) return 1;
    if (prec == DT_F16) return p.ups ? launch_pp<DT_F16, 1>(p, s) : launch_pp<DT_F16, 0>(p, s);
    return p.ups ? launch_pp<DT_BF16, 1>(p, s) : launch_pp<DT_BF16, 0>(p, s);
}
