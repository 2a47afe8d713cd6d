// conv3x3 (stride 1, pad 1, optional fused nearest-x2 upsample) with an LDS-staged input patch -- fourth generation: FOUR waves per
// block, one per SIMD with the whole 512-entry register file, v_mfma_f32_32x32x16, 2-byte compute types (bf16 / fp16), gfx950.
//
// Why: on this chip ONE wave issues v_mfma_f32_16x16x32 every 27 cycles (16 would be the pipe's rate; two waves per SIMD reach 17)
// but v_mfma_f32_32x32x16 every 32-33 -- the pipe's full rate (tools/micro/mfma_issue.hip: 1.11-1.19 against 1.59-1.65 PFLOP/s).
// The 8-wave generations keep two waves per SIMD in lockstep (request, read, MFMA, barrier: 2.6 k cycles per (chunk, tap) unit whose
// MFMAs need 1.28 k) or half a unit apart (conv_patch3.hip: the load phase of 24 fragment reads + LDS-DMA requests takes 1.1 k cycles
// against 0.68 k of MFMAs).  Here a wave owns 4 patch rows (64 pixels) x all 160 channels = 2 x 5 accumulator tiles of 32 x 32
// (160 registers), so a weight fragment feeds two MFMAs and a pixel fragment five (28 fragment reads per 40 MFMAs, 112 KB of LDS reads
// per unit and block instead of 144-192), and its own instruction stream overlaps everything: the fragments of K step t + 2 (16
// channels of one tap) are read while the MFMAs of step t issue, the LDS-DMA requests of weight tile u + 2 and of the next chunk's
// patch sit between MFMAs behind the unit's one barrier.
//
// Padding: the patch pieces are buffer loads to LDS (buffer_load_dwordx4 ... lds) whose lanes outside the image carry an offset past
// the descriptor's range -- the hardware writes zeros for them; no lane masks, no zeroed buffer.
// Same block tiling, LDS images and K order as conv_patch3.hip (a block owns a 16x16 patch of output pixels of one
// sample x 160 output channels; per 128-byte channel chunk the (16+2)^2 input patch sits in LDS once and all 9 taps run from it; the
// [160 x 128 B] weight tile of each unit streams from L2 through a ring of 3); results are bit-identical to the other generations (the
// MFMA shapes sum the same exact products in the same order).
//
// Ordering.  Global K step t = 4 u + s (unit u, 16-channel step s).  The barrier B_u sits in front of step 4 u + 2:
//   RAW  before B_u every wave waits (counted vmcnt) for its own pieces of weight tile u + 1 and of any patch piece requested two
//        units ago; the reads of unit u + 1 (issued from step 4 u + 2 on) come behind B_u.
//   WAR  behind B_u tile u + 2 is requested into the slot of unit u - 1, whose fragments every wave had in registers before its MFMAs of
//        step 4 u - 1; patch pieces of chunk c + 1 go to the other patch buffer, last read in chunk c - 1.
#include <type_traits>

#include "pd_common.h"
#include "pd_mma.h"
#include "pd_stamp.h"

namespace {

constexpr int TP = 16;             // patch is TP x TP output pixels
constexpr int BN = 160;
constexpr int NT = 256;            // threads: 4 waves, one per SIMD
constexpr int ROWB = 128;          // bytes of K per LDS row (64 two-byte channels)
constexpr int BKE = 64;
constexpr int W_TILE = BN * ROWB;  // 20480
constexpr int NWB = 3;             // weight tile ring

// 16-byte chunk index ^= (row >> 1) & 7 (conv_patch3.hip)
__device__ __forceinline__ int swz4(int row, int chunk) { return (row * ROWB) + (((chunk ^ (row >> 1)) & 7) << 4); }
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int P> __device__ __forceinline__ void mma32(const uint4& w, const uint4& a, f32x16& acc) {
    if constexpr (P == DT_F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
}

template <int UPS>
struct Geom4 {
    static constexpr int PW = UPS ? TP / 2 + 2 : TP + 2;   // patch rows/cols held in LDS (source resolution)
    static constexpr int PROWS = PW * PW;
    static constexpr int P_PIECES = (PROWS + 7) / 8;       // 41 (plain) / 13 (upsampling) LDS-DMA pieces of 8 rows
    static constexpr int P_PER_WAVE = (P_PIECES + 3) / 4;  // 11 / 4: wave w requests pieces w + 4 j
    static constexpr int P_BYTES = 4 * P_PER_WAVE * 8 * ROWB;   // whole pieces for every (wave, j): the ones past the patch are lanes out of range (zeros)
    static constexpr int SMEM = 2 * P_BYTES + NWB * W_TILE;
    // the next chunk's patch: taps 1-7 request per_tap(tap) <= 2 pieces per wave, the larger counts first (11 = 2 2 2 2 1 1 1); the
    // counted waits retire the last one in front of the barrier of tap 8
    static constexpr int per_tap(int tap) { return tap >= 1 && tap <= 7 ? (P_PER_WAVE + 7 - tap) / 7 : 0; }
    static constexpr int first_of(int tap) { int f = 0; for (int t = 0; t < tap; ++t) f += per_tap(t); return f; }
};

// LDS-DMA of 64 x 16 B: wave-uniform 64-bit base + 32-bit lane offset -> LDS at M0 + 16 * lane.  M0 is written in the statement that
// uses it and not restored: nothing else in this kernel reads M0 (no other LDS-DMA, no GWS, no movrel), and three scalar instructions
// instead of five matter when ONE wave per SIMD issues everything.  The compiler does not see the request; waits are explicit.
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void bufds16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_dst) {   // the buffer form: lanes out of range write zeros
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rs), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* q) {   // provably wave-uniform for the "s" constraint
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const char*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void w4_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

struct Frag4 { uint4 a[2]; uint4 w[5]; };   // one 16-channel K step: 2 pixel tiles, 5 channel tiles

// Diagnostic build only (tools/micro/conv_w4_stamp.hip compiles this file with -DPD_STAMP): where wave 0 of every block spends its cycles
// (pd_stamp.h; nothing of it exists in the product build).
PD_T_ONLY(__device__ unsigned long long* g_w4_stamps = nullptr;)

template <int P, int UPS>
__global__ __launch_bounds__(NT, 1) void conv3x3_w4_kernel(GemmParams p) {
    using G = Geom4<UPS>;
    constexpr int PW = G::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];


    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    [[maybe_unused]] const int U = nchunks * 9;

    // ---- this lane's LDS-DMA sources.  A piece is 8 LDS rows; lane l writes row 8 * piece + (l >> 3), slot l & 7, and fetches the
    // logical chunk that the swizzle keeps in that slot.
    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned w_off[5];   // weights: pieces wave + 4 j; rows past N re-read the last row (never stored)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int row = (wave + 4 * j) * 8 + lrow;
        const int chunk = (lslot ^ (row >> 1)) & 7;
        const int n = min(bn * BN + row, p.N - 1);
        w_off[j] = (unsigned)(((size_t)n * p.Kpad + chunk * 8) * 2);
    }
    // patch: pieces wave + 4 j; pixels outside the image (and rows past the patch) read at an offset outside the buffer descriptor: zeros
    constexpr unsigned OOB = 0xF0000000u;   // + the chunk offset (< 2^16) stays out of range and does not wrap (operands are < 2^31 bytes)
    unsigned p_off[G::P_PER_WAVE];
#pragma unroll
    for (int j = 0; j < G::P_PER_WAVE; ++j) {
        const int prow = (wave + 4 * j) * 8 + lrow;
        const int chunk = (lslot ^ (prow >> 1)) & 7;
        const int iy = prow / PW, ix = prow - iy * PW;
        const int gy = sy0 + iy, gx = sx0 + ix;
        const bool ok = prow < G::PROWS && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
        p_off[j] = ok ? (unsigned)((((size_t)(sample * p.Hin + gy) * p.Win + gx) * p.lda + chunk * 8) * 2) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((size_t)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * p.lda * 2), 0x00020000);

    const char* Wb = reinterpret_cast<const char*>(p.W);

    // wave-uniform LDS destinations of this wave's pieces (piece wave + 4 j of a weight slot / a patch buffer)
    const unsigned wdst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + 2 * G::P_BYTES + (unsigned)wave * 8 * ROWB));
    const unsigned pdst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)wave * 8 * ROWB));
    auto issue_w = [&](auto J, const char* base, int slot) __attribute__((always_inline)) {   // piece wave + 4 j of a weight tile -> ring slot
        constexpr int j = decltype(J)::value;
        glds16(base, w_off[j], wdst0 + (unsigned)slot * W_TILE + j * 4 * 8 * ROWB);
    };
    auto w_base = [&](int c, int tap) __attribute__((always_inline)) { return uniform_ptr(Wb + ((size_t)tap * p.Cin + (size_t)c * BKE) * 2); };
    auto issue_p = [&](auto J, int c, int buf) __attribute__((always_inline)) {                // piece wave + 4 j of chunk c's patch
        constexpr int j = decltype(J)::value;
        if constexpr (j < G::P_PER_WAVE) bufds16(rsA, p_off[j], (unsigned)(c * BKE * 2), pdst0 + (unsigned)buf * G::P_BYTES + j * 4 * 8 * ROWB);
    };

    f32x16 acc[5][2];   // [channel tile][pixel tile]: lane = pixel l31 of the tile, register r = channel (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

    // ---- prologue: patch c0 and the weight tiles of units 0 and 1
    static_for<G::P_PER_WAVE>([&](auto J) { issue_p(J, c0, 0); });
    {
        const char* b0 = w_base(c0, 0);
        static_for<5>([&](auto J) { issue_w(J, b0, 0); });
        const char* b1 = w_base(c0, 1);
        static_for<5>([&](auto J) { issue_w(J, b1, 1); });
    }
    wait_vm<0>();
    w4_barrier();

    // One K step in 10 slots of one MFMA each (MFMA i = channel tile i / 2 x pixel tile i % 2 of the step's fragments), each followed by
    // the instructions that travel under it: slots 0-6 issue the 7 fragment reads of step t + 2 (2 pixel tiles, 5 channel tiles),
    // slots 7-9 the LDS-DMA requests.  sched_barrier pins the slots: with one wave per SIMD nothing else fills the matrix pipe while
    // this wave issues something else, so the order IS the schedule.
    // Fragment addresses.  Chunk 2 s + lh of a row whose swizzle term is x sits at ((2 s + lh) ^ x) & 7 = (2 s) ^ ((lh ^ x) & 7): per row one
    // base and one pre-shifted y = ((lh ^ x) & 7) << 4, per step one v_xad (y ^ 32 s) + base.
    struct ReadCtx { unsigned pb[2], py[2], wb; };   // LDS byte addresses
    const unsigned wy = (unsigned)(((lh ^ (l31 >> 1)) & 7) << 4);   // channel row 32 n + l31: (row >> 1) & 7 = (l31 >> 1) & 7 for every n
    auto read_ctx = [&](auto TAPC, int lc) __attribute__((always_inline)) -> ReadCtx {
        constexpr int tap = decltype(TAPC)::value;
        constexpr int ky = tap / 3, kx = tap % 3;
        ReadCtx r;
        int lv = l31;
        asm volatile("" : "+v"(lv));   // keep the per-tap addresses out of loop-invariant hoisting (they would occupy registers for the whole kernel)
#pragma unroll
        for (int m = 0; m < 2; ++m) {  // pixel l31 of tile m: patch row 4 * wave + 2 m + (l31 >> 4), column l31 & 15
            const int prow = (((y0 - 1 + 4 * wave + 2 * m + (lv >> 4) + ky) >> UPS) - sy0) * PW + (((x0 - 1 + (lv & 15) + kx) >> UPS) - sx0);
            r.pb[m] = (unsigned)((lc & 1) * G::P_BYTES + prow * ROWB);
            r.py[m] = (unsigned)(((lh ^ (prow >> 1)) & 7) << 4);
        }
        r.wb = (unsigned)(2 * G::P_BYTES + (tap % NWB) * W_TILE + lv * ROWB);   // 9 units per chunk keep unit % 3 == tap % 3
        return r;
    };
    auto read_one = [&](auto IC, auto SC, const ReadCtx& r, Frag4& f) __attribute__((always_inline)) {
        constexpr int i = decltype(IC)::value, s = decltype(SC)::value;
        if constexpr (i < 2) f.a[i] = *reinterpret_cast<const uint4*>(smem + ((r.py[i] ^ (32u * s)) + r.pb[i]));
        else f.w[i - 2] = *reinterpret_cast<const uint4*>(smem + ((wy ^ (32u * s)) + r.wb) + (i - 2) * 32 * ROWB);
    };

    // three fragment sets: step t lives in set t % 3 (36 steps per chunk: the assignment repeats chunk after chunk)
    Frag4 f0, f1, f2;
    auto FS = [&](auto T) __attribute__((always_inline)) -> Frag4& {
        constexpr int r = decltype(T)::value % 3;
        if constexpr (r == 0) return f0; else if constexpr (r == 1) return f1; else return f2;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    const char* wbase = nullptr;      // source of the weight tile being requested (set behind each barrier)
    ReadCtx rc = read_ctx(I0{}, 0);   // of the unit whose fragments are being read: recomputed behind each barrier
    static_for<7>([&](auto IC) { read_one(IC, I0{}, rc, f0); });
    static_for<7>([&](auto IC) { read_one(IC, I1{}, rc, f1); });

    [[maybe_unused]] unsigned long long c_wait = 0, c_bar = 0;
    [[maybe_unused]] const unsigned long long t_begin = PD_T_NOW();
    PD_T_ONLY(const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();)
    // The steady state has NO branches (with one wave per SIMD every taken branch is ~15 cycles in which the matrix pipe idles: the first
    // build, with its "last unit?" tests around the reads and requests, lost 500 cycles per unit to them): reads past the slice's end
    // fetch stale LDS bytes that nothing uses, requests past it repeat the last tile / chunk into buffers nobody reads any more, and
    // everything is drained in front of the epilogue.
    const int last_c = c0 + nchunks - 1;
    auto step = [&](auto TAPC, auto SC, int lc) __attribute__((always_inline)) {
        constexpr int tap = decltype(TAPC)::value, s = decltype(SC)::value;
        constexpr int tl = tap * 4 + s;   // step index inside the chunk
        const int c = c0 + lc;
        if constexpr (s == 2) {
            // B_u: tile u + 1 and the previous unit's patch pieces have landed for every wave; this unit's own patch pieces (requested in
            // steps 0 and 1, the youngest entries of the queue) stay in flight
            [[maybe_unused]] const unsigned long long t0 = PD_T_NOW();
            wait_vm<G::per_tap(tap)>();
            [[maybe_unused]] const unsigned long long t1 = PD_T_NOW();
            w4_barrier();
            [[maybe_unused]] const unsigned long long t2 = PD_T_NOW();
            PD_T_ADD(c_wait, t0, t1); PD_T_ADD(c_bar, t1, t2);
        }
        // step t + 2: (tap2, s2) of this chunk or the next
        constexpr int s2 = (s + 2) % 4, tap2 = (tap + (s + 2) / 4) % 9;
        constexpr bool wrap = tap + (s + 2) / 4 >= 9;
        if constexpr (s == 2) rc = read_ctx(std::integral_constant<int, tap2>{}, wrap ? lc + 1 : lc);   // steps t + 2 ... t + 5 are unit u + 1
        Frag4& fn = FS(std::integral_constant<int, tl + 2>{});
        const Frag4& fc = FS(std::integral_constant<int, tl>{});
        constexpr int t2 = tap + 2;
        const char* wb = nullptr;
        if constexpr (s == 2) wbase = w_base(min(t2 >= 9 ? c + 1 : c, last_c), t2 >= 9 ? t2 - 9 : t2);   // tile u + 2 (past the end: a valid tile, unused)
        if constexpr (s >= 2) wb = wbase;
        const int cn = min(c + 1, last_c);
        __builtin_amdgcn_sched_barrier(0);
        static_for<10>([&](auto IC) {
            constexpr int i = decltype(IC)::value;
            mma32<P>(fc.w[i / 2], fc.a[i % 2], acc[i / 2][i % 2]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i < 7) read_one(IC, std::integral_constant<int, s2>{}, rc, fn);
            // LDS-DMA requests.  Behind B_u (steps 2 and 3): the 5 pieces of weight tile u + 2, into the slot of unit u - 1, one every
            // fourth MFMA.  Steps 0 and 1 of taps 1-7: the next chunk's patch pieces of this tap (behind B of tap 0 every wave has the
            // previous chunk's last fragments in registers).
            if constexpr (s >= 2) {
                constexpr int slot = 10 * (s - 2) + i;
                if constexpr (slot % 4 == 1) issue_w(std::integral_constant<int, slot / 4>{}, wb, t2 % NWB);
            } else if constexpr (i == 8 && s < G::per_tap(tap)) {
                issue_p(std::integral_constant<int, G::first_of(tap) + s>{}, cn, (lc + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto unit = [&](auto TAPC, int lc) __attribute__((always_inline)) {
        step(TAPC, std::integral_constant<int, 0>{}, lc);
        step(TAPC, std::integral_constant<int, 1>{}, lc);
        step(TAPC, std::integral_constant<int, 2>{}, lc);
        step(TAPC, std::integral_constant<int, 3>{}, lc);
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

    wait_vm<0>();   // the requests past the slice's end have landed: no LDS-DMA may outlive the block
    // ---- epilogue (split-K: this slice's fp32 partial goes to its slab; splitk_finalize_kernel or the consumer sums and finishes)
    [[maybe_unused]] const unsigned long long t_epi = PD_T_NOW();
    float* slab = p.splitk > 1 ? reinterpret_cast<float*>(p.slab) + (size_t)blockIdx.y * p.M * p.N : nullptr;
    // this lane's two pixels and their 20 groups of 4 consecutive channels each: tile n, group g -> channel 32 n + 8 g + 4 lh
    const bool fast = !slab && p.act == 0 && p.c_dt == P && (!p.R || p.r_dt == P) && !p.ln_stats;
    if (fast) {
        // pd_mma.h epilogue4's arithmetic -- (acc + bias + time-embedding row) * scale + residual -- with EVERY read issued before the first
        // use: written as 40 calls of epilogue4_value, hipcc waits vmcnt(0) behind each optional read (120 dependent round trips per
        // lane: 54 k cycles per tile on this kernel).  Absent operands read a valid address and are not added (select, no branch).
        const bool hb = p.bias != nullptr, hv = p.rowvec != nullptr, hr = p.R != nullptr;
        // buffer loads / stores: one lane offset per operand + compile-time offsets per channel group (40 64-bit addresses per operand
        // would not fit beside the accumulators); channel groups past N read zeros or a neighbour row and are not stored
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hb ? p.bias : reinterpret_cast<const float*>(p.W)), 0, p.N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(hv ? p.rowvec + (size_t)sample * p.rowvec_stride : reinterpret_cast<const float*>(p.W)), 0, p.N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(hr ? p.R : p.C), 0, (int)((size_t)p.M * (hr ? p.ldr : p.ldc) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rc_ = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((size_t)p.M * p.ldc * 2), 0x00020000);
        const int ldr = hr ? p.ldr : 0;
        const int gn0 = bn * BN + lh * 4;   // + 32 n + 8 g
        // wide = 16-byte accesses: the two lane halves of a pixel exchange the halves of two neighbouring channel groups (v_permlane32_swap),
        // after which a lane owns 8 consecutive channels -- one store instruction then writes 32 contiguous bytes per pixel (as the 16x16
        // kernels do; the 8-byte form measured 2.6x the output bytes at the memory side) in half as many instructions, and the residual
        // comes in the same way.  Needs whole 16-byte pieces: N, ldc, ldr multiples of 8.
        const bool wide = p.N % 8 == 0 && p.ldc % 8 == 0 && (!hr || p.ldr % 8 == 0);
        const int gn1 = bn * BN + lh * 8;   // wide: + 32 n + 8 g (g even)
        unsigned ro[2], co[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int tok = (y0 + 4 * wave + 2 * m + (l31 >> 4)) * p.Wout + x0 + (l31 & 15);
            const unsigned gm = (unsigned)(sample * p.rows_per_sample + tok);
            ro[m] = (gm * (unsigned)ldr + (unsigned)(wide ? gn1 : gn0)) * 2u;
            co[m] = (gm * (unsigned)p.ldc + (unsigned)(wide ? gn1 : gn0)) * 2u;
        }
        auto swap2 = [](uint2& a, uint2& b) __attribute__((always_inline)) {   // lanes 32-63 of a <-> lanes 0-31 of b, both dwords
            auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
            a.x = rx[0]; b.x = rx[1]; a.y = ry[0]; b.y = ry[1];
        };
        // two halves of 10 channel groups (both pixels): the second half's bias / time-embedding rows are requested once the first half's
        // results are packed, still in front of the first store -- 120 + 40 instead of 240 operand registers
        f32x4 eb[10], ev[10];
        uint2 er[2][20], out[2][10];
        auto load_vec = [&](auto H) __attribute__((always_inline)) {
            constexpr int h = decltype(H)::value;
            static_for<10>([&](auto Q) {
                constexpr int qq = 10 * h + decltype(Q)::value, q = decltype(Q)::value;
                constexpr int coff = ((qq >> 2) * 32 + (qq & 3) * 8) * 4;
                eb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)gn0 * 4u, coff, 0));
                ev[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)gn0 * 4u, coff, 0));
            });
        };
        auto value = [&](int m, int qq) __attribute__((always_inline)) -> uint2 {   // qq: 0..19; eb / ev hold its half
            const int n = qq >> 2, g = qq & 3, q = qq % 10;
            f32x4 v = f32x4{acc[n][m][4 * g], acc[n][m][4 * g + 1], acc[n][m][4 * g + 2], acc[n][m][4 * g + 3]};
            const f32x4 vb = v + eb[q];
            v = hb ? vb : v;
            const f32x4 vv = v + ev[q];
            v = hv ? vv : v;
            float r0, r1, r2, r3;
            unpack2<P>(er[m][qq].x, r0, r1); unpack2<P>(er[m][qq].y, r2, r3);
            // scale, THEN the residual: two roundings, as the reference does it (cldm.py:379 multiplies the control tensor, :41 adds it) and as
            // epilogue4_value ends up in the other generations.  Left to itself hipcc contracts the two statements into one fused
            // multiply-add here (0.015 % of the outputs of a scaled launch then differ by one ulp of the storage type): the empty asm
            // keeps the product a value of its own
            const f32x4 sc = f32x4{p.out_scale, p.out_scale, p.out_scale, p.out_scale};
            f32x4 vs = v * sc;
            asm volatile("" : "+v"(vs));
            const f32x4 vr = vs + f32x4{r0, r1, r2, r3};
            v = hr ? vr : vs;
            uint2 o;
            o.x = pack2<P>(v[0], v[1]); o.y = pack2<P>(v[2], v[3]);
            return o;
        };
        auto store_half = [&](auto H) __attribute__((always_inline)) {
            constexpr int h = decltype(H)::value;
            static_for<2>([&](auto MC) {
                constexpr int m = decltype(MC)::value;
                static_for<10>([&](auto Q) {
                    constexpr int qq = 10 * h + decltype(Q)::value, q = decltype(Q)::value;
                    constexpr int coff = (qq >> 2) * 32 + (qq & 3) * 8;
                    if (!wide) {
                        if (gn0 + coff < p.N) { typedef unsigned int u32x2_t __attribute__((__vector_size__(8))); __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, out[m][q]), rc_, co[m], coff * 2, 0); }
                    } else if constexpr ((qq & 1) == 0) {
                        // groups g (even) and g + 1: lower lanes end with [own g | upper's g] = channels 8 g .. 8 g + 7, upper lanes with
                        // [lower's g + 1 | own g + 1] = the next 8
                        uint2 a = out[m][q], b = out[m][q + 1];
                        swap2(a, b);
                        if (gn1 + coff < p.N) __builtin_amdgcn_raw_buffer_store_b128(u32x4{a.x, a.y, b.x, b.y}, rc_, co[m], coff * 2, 0);
                    }
                });
            });
        };
        load_vec(std::integral_constant<int, 0>{});
        static_for<2>([&](auto MC) {
            constexpr int m = decltype(MC)::value;
            static_for<20>([&](auto Q) {
                constexpr int qq = decltype(Q)::value;
                constexpr int coff = (qq >> 2) * 32 + (qq & 3) * 8;
                if (!wide) {
                    er[m][qq] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rr, ro[m], coff * 2, 0));
                } else if constexpr ((qq & 1) == 0) {   // 16 bytes = this lane's 8 channels of the pair (g, g + 1); un-swapped below
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rr, ro[m], coff * 2, 0);
                    er[m][qq] = make_uint2(t[0], t[1]);
                    er[m][qq + 1] = make_uint2(t[2], t[3]);
                }
            });
        });
        if (wide) {   // X = first halves, Y = second halves: after the swap X is group g for both lane halves, Y group g + 1
            static_for<2>([&](auto MC) {
                static_for<10>([&](auto Q) { swap2(er[decltype(MC)::value][2 * decltype(Q)::value], er[decltype(MC)::value][2 * decltype(Q)::value + 1]); });
            });
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 10; ++q) out[m][q] = value(m, q);
        __builtin_amdgcn_sched_barrier(0);
        load_vec(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        store_half(std::integral_constant<int, 0>{});
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 10; ++q) out[m][q] = value(m, 10 + q);
        store_half(std::integral_constant<int, 1>{});
    } else {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int oy = y0 + 4 * wave + 2 * m + (l31 >> 4), ox = x0 + (l31 & 15);
        const int tok = oy * p.Wout + ox;
        const int gm = sample * p.rows_per_sample + tok;
#pragma unroll
        for (int n = 0; n < 5; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gn = bn * BN + n * 32 + g * 8 + lh * 4;
                if (gn >= p.N) continue;
                f32x4 v = f32x4{acc[n][m][4 * g], acc[n][m][4 * g + 1], acc[n][m][4 * g + 2], acc[n][m][4 * g + 3]};
                if (slab) *reinterpret_cast<f32x4*>(slab + (size_t)gm * p.N + gn) = v;
                else epilogue4(p, gm, gn, sample, tok, v);
            }
    }
    }
    PD_T_ONLY(if (tid == 0 && g_w4_stamps) {
        unsigned long long* o = g_w4_stamps + (size_t)blockIdx.x * 8;
        const unsigned long long t_end = PD_T_NOW();
        o[0] = t_begin; o[1] = t_end; o[2] = c_wait; o[3] = c_bar; o[4] = t_end - t_epi; o[5] = (unsigned long long)U; o[6] = __builtin_amdgcn_s_memrealtime() - r_begin;
    })
}

template <int P, int UPS>
int launch_w4(const GemmParams& p, hipStream_t s) {
    using G = Geom4<UPS>;
    static_assert(G::SMEM <= 160 * 1024, "LDS");
    static unsigned long long attr_done = 0;
    auto kfn = conv3x3_w4_kernel<P, UPS>;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), G::SMEM, &attr_done)) return 1;
    const int mtiles = (p.M / (p.Hout * p.Wout)) * (p.Hout / TP) * (p.Wout / TP), ntiles = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kfn, dim3(mtiles * ntiles, p.splitk > 1 ? p.splitk : 1), dim3(NT), G::SMEM, s, p);
    if (hipGetLastError() != hipSuccess) return 1;
    return (p.splitk > 1 && !p.defer_finalize) ? launch_splitk_finalize(p, s) : 0;
}

}  // namespace

// shapes conv_patch_tiles() accepts, 2-byte compute types, no fused GroupNorm, plain epilogue; 31-bit byte ranges inside each operand
// (buffer descriptors) and 32-bit weight offsets
bool conv_patch4_eligible(const GemmParams& p, int prec) {
    if (prec != DT_F16 && prec != DT_BF16) return false;
    if (p.gn_coef || p.a_dt != prec || p.Cin % BKE || p.K != 9 * p.Cin || p.act != 0) return false;
    const unsigned long long a_bytes = (unsigned long long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * (unsigned)p.lda * 2ull;
    const unsigned long long w_bytes = (unsigned long long)p.N * (unsigned)p.Kpad * 2ull;
    const unsigned long long c_bytes = (unsigned long long)p.M * (unsigned)(p.ldc > p.ldr ? p.ldc : p.ldr) * 4ull;
    return a_bytes < (1ull << 31) && w_bytes < (1ull << 32) && c_bytes < (1ull << 31);
}

int launch_conv_patch4(const GemmParams& p, int prec, hipStream_t s) {
    if (!conv_patch4_eligible(p, prec)) return 1;
    if (prec == DT_F16) return p.ups ? launch_w4<DT_F16, 1>(p, s) : launch_w4<DT_F16, 0>(p, s);
    return p.ups ? launch_w4<DT_BF16, 1>(p, s) : launch_w4<DT_BF16, 0>(p, s);
}
