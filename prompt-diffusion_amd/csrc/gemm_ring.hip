// Linear layers with a short reduction (K <= a few thousand) on gfx950: a persistent GEMM whose operands travel
// HBM/L2 -> LDS by LDS-DMA through a ring of NS stages.
//
// Why a second GEMM kernel: gemm.hip's igemm_kernel stages global -> VGPR -> LDS one K step (64 elements) ahead.  A K step
// of a 128 x 160 tile is 20 MFMAs per wave (~0.15 us), a memory round trip under load 1-2 us, so with K = 640 / 1280 (10 / 20
// steps) every step waits out most of a round trip: 16384 x 640 x 640 takes 27 us where the MFMAs need 5 and the traffic floor
// is 13 (tools/bench_linear.py).  Here nothing is staged through registers, so the prefetch distance is bounded by LDS, not
// by VGPRs: NS - 1 stages (108-156 KB per CU) are in flight while one is consumed, and because the block is persistent the
// ring runs on across tile boundaries -- the first stages of the next tile land under the epilogue of the current one.
//
//   C[M, N] = epilogue(A[M, K] . W[N, K]^T), A and W 2-byte operands of the compute type, K a multiple of 64.
//   Block: WM x WN waves, tile BM x BN, one block per CU (launch bound), each block walks its XCD's run of tiles.
//   Stage: BM + BN rows of 128 B (64 K elements), the same 16-byte-chunk XOR swizzle as gemm.hip; one wave-instruction of
//   LDS-DMA fills 8 rows (64 lanes x 16 B, LDS destination linear in the lane, global source address per lane -- the lane
//   fetches the chunk that belongs in its swizzled slot).
//   Step g: wait until this wave's own requests for stage g have landed (counted vmcnt: the younger stages stay in
//   flight), one barrier (everyone's stage g is there, everyone is done with stage g - 1), request stage g + NS - 1 into
//   the slot of g - 1, then the step's MFMAs.
// The MFMA / epilogue side is igemm_kernel's (16x16x32 "swapped" MFMAs, a lane owns 4 consecutive output channels).
#include "pd_common.h"
#include "pd_mma.h"
#include "pd_stamp.h"

namespace {

constexpr int BKB = 128;  // bytes of K per LDS row
__device__ __forceinline__ int swz(int row, int chunk) { return (row * BKB) + (((chunk ^ (row >> 1)) & 7) << 4); }

// LDS-DMA of 64 x 16 B: wave-uniform 64-bit base in SGPRs + a 32-bit lane offset; M0 (compiler-reserved) carries the LDS
// destination and is saved / restored inside the statement.  The compiler does not see the request: every wait on it is an
// explicit s_waitcnt below (its own counted waits stay correct -- foreign entries in the queue only make them wait longer).
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

extern __shared__ __attribute__((aligned(16))) char smem[];

// Diagnostic build only (tools/micro/ring_stamp.hip compiles this file with -DPD_STAMP): cycles wave 0 of every block spends in
// each part of its step loop (pd_stamp.h; nothing of it exists in the product build).
PD_T_ONLY(__device__ unsigned long long* g_ring_stamps = nullptr;)

// GG: the GEGLU epilogue (act == 2; weights pre-interleaved so that virtual columns [0, 80) of each 160-column block are x and
// [80, 160) the gate): WN = 1, a wave owns whole blocks and writes x * gelu(gate), 80 columns per block.
// PP ("ping-pong"): the waves of a block form two groups (waves >= WM * WN / 2 share their SIMDs one-to-one with the waves below) that
// run half a K step apart: a step is a load phase (ring request, all fragment reads of the step) and an MFMA phase with a barrier
// after each, and group 1 enters the loop one barrier late, so on every SIMD one wave issues MFMAs while its partner reads LDS and
// issues LDS-DMA.  Same MFMA order per accumulator as the lockstep form (bit-identical results).
template <int P, int BM, int BN, int WM, int WN, int NS, bool GG = false, bool PP = false>
__global__ __launch_bounds__(WM * WN * 64, 1) void rgemm_kernel(GemmParams p) {
    constexpr int NTHREADS = WM * WN * 64, RPI = NTHREADS / 8;   // RPI: tile rows one pass of all threads covers
    constexpr int A_IT = BM / RPI, B_IT = (BN + RPI - 1) / RPI, L = A_IT + B_IT;   // L: LDS-DMAs per wave and stage
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    static_assert(BM % RPI == 0 && RPI % 16 == 0 && BN % 8 == 0, "an 8-row DMA group is entirely inside or outside the tile");
    static_assert(NS >= 2 && NS <= 5 && (NS - 2) * L <= 63, "vmcnt immediates");
    static_assert(!GG || (WN == 1 && BN == 160), "GEGLU: a wave's columns are one whole 160-column block");

    const int tid = threadIdx.x, lane = tid & 63, wave = sgpr(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // This block's tiles: XCD x = blockIdx & 7 owns a contiguous run of the (n fastest) tile order -- neighbours sharing A
    // rows meet in one L2 -- and its blocks take the run's tiles round-robin.
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN, nblk = mtiles * ntiles;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int cnt = q + (x < r ? 1 : 0), start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const int nmy = cnt > j ? (cnt - j + nbx - 1) / nbx : 0;
    if (nmy == 0) return;
    const int KT = p.K / 64, G = nmy * KT;

    // ---- producer: the ring runs over the block's whole step sequence (tile after tile)
    const int row0 = tid >> 3;
    const unsigned lchunk = (unsigned)(((tid & 7) ^ ((row0 >> 1) & 7)) << 4);   // the logical chunk this lane's swizzled slot holds
    const unsigned ldw = (unsigned)(p.ldw ? p.ldw : p.Kpad);
    unsigned a_off[A_IT], w_off[B_IT];
    auto setup = [&](int i) __attribute__((always_inline)) {
        const int t = start + j + i * nbx, bm = t / ntiles, bn = t - bm * ntiles;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int m = min(bm * BM + row0 + RPI * it, p.M - 1);   // rows past M: a clamped copy, never stored
            a_off[it] = (unsigned)m * (unsigned)p.lda * 2u + lchunk;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int n = min(bn * BN + row0 + RPI * it, p.N - 1);
            w_off[it] = (unsigned)n * ldw * 2u + lchunk;
        }
    };
    int pj = 0, pkt = 0, pslot = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        const char* As = reinterpret_cast<const char*>(p.A) + (size_t)pkt * BKB;
        const char* Ws = reinterpret_cast<const char*>(p.W) + (size_t)pkt * BKB;
        const unsigned dst = lds0 + (unsigned)pslot * STAGE + (unsigned)wave * 8 * BKB;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) glds16(As, a_off[it], (unsigned)sgpr((int)(dst + it * RPI * BKB)));
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            // the last pass covers BN only partly: its waves outside repeat their previous group (same bytes, same place),
            // so that every wave has exactly L requests per stage in its queue
            const bool in = (BN % RPI == 0) || it + 1 < B_IT || wave * 8 + RPI * it < BN;
            const int u = in ? it : it - 1;
            const unsigned off = in ? w_off[it] : w_off[it > 0 ? it - 1 : 0];
            glds16(Ws, off, (unsigned)sgpr((int)(dst + (BM + RPI * u) * BKB)));
        }
        if (++pkt == KT) {
            pkt = 0;
            if (++pj < nmy) setup(pj);
        }
        pslot = pslot + 1 == NS ? 0 : pslot + 1;
    };
    auto wait_stage = [&](int ahead) __attribute__((always_inline)) {   // all but the `ahead` youngest stages of this wave
        if (ahead >= NS - 2) wait_vm<(NS - 2) * L>();
        else if (ahead == 2) wait_vm<2 * L>();
        else if (ahead == 1) wait_vm<L>();
        else wait_vm<0>();
    };

    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[NT][MT];
    auto compute = [&](int slot) __attribute__((always_inline)) {
        const char* sa = smem + slot * STAGE;
        const char* sb = sa + BM * BKB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[MT], wf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                af[m] = *reinterpret_cast<const uint4*>(sa + swz(wm * WTM + m * 16 + fr, ks * 4 + fq));
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                wf[n] = *reinterpret_cast<const uint4*>(sb + swz(wn * WTN + n * 16 + fr, ks * 4 + fq));
            }
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) mma<P>(wf[n], af[m], acc[n][m]);
        }
    };

    // PP: a step's fragments are all read in its load phase (both 32-element halves) and consumed in its MFMA phase
    [[maybe_unused]] uint4 paf[PP ? 2 : 1][MT], pwf[PP ? 2 : 1][NT];
    auto load_frags = [&](int slot) __attribute__((always_inline)) {
        const char* sa = smem + slot * STAGE;
        const char* sb = sa + BM * BKB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) paf[PP ? ks : 0][m] = *reinterpret_cast<const uint4*>(sa + swz(wm * WTM + m * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int n = 0; n < NT; ++n) pwf[PP ? ks : 0][n] = *reinterpret_cast<const uint4*>(sb + swz(wn * WTN + n * 16 + fr, ks * 4 + fq));
        }
    };
    auto mma_frags = [&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) mma<P>(pwf[PP ? ks : 0][n], paf[PP ? ks : 0][m], acc[n][m]);
    };
    auto pp_barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // {mean, rstd} of the tile's rows (folded LayerNorm); PP: one buffer per tile parity (group 0 writes the next tile's while group 1
    // still has the current tile's epilogue ahead of it)
    float2* sLn = reinterpret_cast<float2*>(smem + NS * STAGE);
    [[maybe_unused]] unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_mma = 0, c_epi = 0;
    [[maybe_unused]] const unsigned long long t_begin = PD_T_NOW();
    setup(0);
    int issued = 0;
    for (; issued < NS - 1 && issued < G; ++issued) issue();
    int cslot = 0;
    [[maybe_unused]] const int grp = wave >= WM * WN / 2 ? 1 : 0;
    if constexpr (PP) {
        wait_stage(issued - 1);   // stage 0 of every wave has landed
        pp_barrier();
        if (grp) pp_barrier();    // group 1 runs one phase behind
    }
    for (int i = 0; i < nmy; ++i) {
        const int t = start + j + i * nbx, bm = t / ntiles, bn = t - bm * ntiles;
        if constexpr (PP) sLn = reinterpret_cast<float2*>(smem + NS * STAGE) + (i & 1) * BM;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the epilogue's operands (this lane's bias / column sums / residual values): requested in the tile's last K step, behind
        // that step's ring request, so that they travel under its MFMAs and nothing is read once the tile's stores have begun
        // -- loads and stores retire through one in-order counter, and a read issued behind a store waits for that store
        f32x4 e_bias[NT], e_cs[NT];
        uint2 e_res[MT][NT];
        if constexpr (PP) {
            for (int kt = 0; kt < KT; ++kt) {
                const int g = i * KT + kt;
                // ---- load phase (the partner wave on this SIMD is in its MFMA phase)
                if (kt == 0 && p.ln_stats && tid < BM) {   // (before the ring request: the compiler waits vmcnt(0) for these reads)
                    const int gm = bm * BM + tid;
                    float mean = 0.f, rstd = 0.f;
                    if (gm < p.M) ln_row_stats(p, gm, mean, rstd);
                    sLn[tid] = make_float2(mean, rstd);
                }
                // the step's fragment reads first (their latency runs under the DMA issue); stage g + NS - 1 goes into the slot of stage
                // g - 1, which both groups have read before the barrier this phase began with
                load_frags(cslot);
                __builtin_amdgcn_sched_barrier(0);
                if (issued < G) { issue(); ++issued; }
                wait_stage(issued - 2 - g);   // this wave's pieces of stage g + 1 have landed (read from the next barrier-but-one on)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                pp_barrier();
                // ---- MFMA phase
                mma_frags(std::integral_constant<int, 0>{});
                if (!GG && kt == KT - 1) {    // the epilogue's residual values: requested between the two halves of the tile's last MFMA phase,
                    if (p.R) {                // into the registers of the first half's fragments, so that they travel under the second half
                        // buffer loads: one lane offset + a scalar offset per (m, n) instead of 20 64-bit addresses held across the K loop; rows
                        // past M read as zero, column groups past N read the next row (neither is stored)
                        __builtin_amdgcn_sched_barrier(0);
                        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.R), 0, (int)((size_t)p.M * p.ldr * 2), 0x00020000);
                        const unsigned r_lane = ((unsigned)(bm * BM + wm * WTM + fr) * (unsigned)p.ldr + (unsigned)(bn * BN + wn * WTN + fq * 4)) * 2u;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                e_res[m][n] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rres, r_lane, (m * 16 * p.ldr + n * 16) * 2, 0));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                mma_frags(std::integral_constant<int, 1>{});
                cslot = cslot + 1 == NS ? 0 : cslot + 1;
                if (!(grp && g + 1 == G)) pp_barrier();   // (group 1 entered one barrier late and leaves one early)
                else __builtin_amdgcn_sched_barrier(0);
            }
        } else
        for (int kt = 0; kt < KT; ++kt) {
            [[maybe_unused]] const unsigned long long t0 = PD_T_NOW();
            wait_stage(issued - 1 - (i * KT + kt));
            [[maybe_unused]] const unsigned long long t1 = PD_T_NOW();
            asm volatile("s_barrier" ::: "memory");
            [[maybe_unused]] const unsigned long long t2 = PD_T_NOW();
            if (kt == 0 && p.ln_stats && tid < BM) {   // behind the barrier: every wave has left the previous tile's epilogue
                const int gm = bm * BM + tid;
                float mean = 0.f, rstd = 0.f;
                if (gm < p.M) ln_row_stats(p, gm, mean, rstd);
                sLn[tid] = make_float2(mean, rstd);
            }
            if (issued < G) { issue(); ++issued; }
            if (!GG && kt == KT - 1) {
                if (p.R) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int gm = min(bm * BM + wm * WTM + m * 16 + fr, p.M - 1);
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const int gn = min(bn * BN + wn * WTN + n * 16 + fq * 4, p.N - 4);
                            e_res[m][n] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.R) + (size_t)gm * p.ldr + gn);
                        }
                    }
                }
            }
            [[maybe_unused]] const unsigned long long t3 = PD_T_NOW();
            compute(cslot);
            cslot = cslot + 1 == NS ? 0 : cslot + 1;
            PD_T_ONLY(asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[NT - 1][MT - 1]));)   // the step\'s MFMAs are issued before the stamp
            [[maybe_unused]] const unsigned long long t4 = PD_T_NOW();
            PD_T_ADD(c_wait, t0, t1); PD_T_ADD(c_bar, t1, t2); PD_T_ADD(c_issue, t2, t3); PD_T_ADD(c_mma, t3, t4);
        }
        [[maybe_unused]] const unsigned long long t5 = PD_T_NOW();
        if constexpr (GG) {
            // ---- GEGLU epilogue: out[:, 80 bn + j] = (x_j + b_j) * gelu(g_j + b_{80 + j}); the bias reads go first (see above)
#pragma unroll
            for (int n = 0; n < NT; ++n) e_bias[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + bn * BN + n * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gm = bm * BM + wm * WTM + m * 16 + fr;
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    const int on = bn * 80 + n * 16 + fq * 4;
                    const f32x4 x = acc[n][m] + e_bias[n], g = acc[n + 5][m] + e_bias[n + 5];
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = x[c] * gelu_fast(g[c]);
                    if (gm < p.M && on < p.Nout) store4(p.C, (size_t)gm * p.ldc + on, p.c_dt, o);
                }
            }
            [[maybe_unused]] const unsigned long long t6g = PD_T_NOW();
            PD_T_ADD(c_epi, t5, t6g);
            continue;
        }
        // ---- epilogue (pd_mma.h epilogue4's arithmetic): a lane holds channels gn .. gn + 3 of row gm
#pragma unroll
        for (int n = 0; n < NT; ++n) {   // (the per-column vectors: cache hits, read once the step's fragment registers are free)
            const int gn = min(bn * BN + wn * WTN + n * 16 + fq * 4, p.N - 4);
            e_bias[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
            e_cs[n] = p.ln_stats ? *reinterpret_cast<const f32x4*>(p.ln_colsum + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int gm = bm * BM + wm * WTM + m * 16 + fr;
            const bool row_ok = gm < p.M;
            const int sample = gm / p.rows_per_sample;
            const int tok = gm - sample * p.rows_per_sample;
            float ln_mean = 0.f, ln_rstd = 0.f;
            if (p.ln_stats) {
                const float2 st = sLn[wm * WTM + m * 16 + fr];
                ln_mean = st.x;
                ln_rstd = st.y;
            }
            float rs = 0.f, rq = 0.f;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int gn = bn * BN + wn * WTN + n * 16 + fq * 4;
                f32x4 v = acc[n][m];
                if (p.ln_stats) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = ln_rstd * (v[c] - ln_mean * e_cs[n][c]);
                }
                v += e_bias[n];
                if (p.act == 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = silu_f(v[c]);
                } else if (p.act == 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = v[c] / (1.0f + __expf(-1.702f * v[c]));
                }
                v *= p.out_scale;
                if (p.R) {
                    float r0, r1, r2, r3;
                    if (p.r_dt == DT_F16) { unpack2<DT_F16>(e_res[m][n].x, r0, r1); unpack2<DT_F16>(e_res[m][n].y, r2, r3); }
                    else { unpack2<DT_BF16>(e_res[m][n].x, r0, r1); unpack2<DT_BF16>(e_res[m][n].y, r2, r3); }
                    v += f32x4{r0, r1, r2, r3};
                }
                if (!row_ok || gn >= p.N) continue;
                if (gn >= p.vt_begin) {   // transposed store (attention V^T): [sample][channel][token]
                    const size_t base = ((size_t)sample * (p.N - p.vt_begin) + (gn - p.vt_begin)) * p.vt_ld + tok;
                    if (p.c_dt == DT_F32) {
                        float* o = reinterpret_cast<float*>(p.VT);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[base + (size_t)c * p.vt_ld] = v[c];
                    } else {
                        uint16_t* o = reinterpret_cast<uint16_t*>(p.VT);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[base + (size_t)c * p.vt_ld] = cvt16_rt(v[c], p.c_dt);
                    }
                } else {
                    store4(p.C, (size_t)gm * p.ldc + gn, p.c_dt, v);
                }
                rs += (v[0] + v[1]) + (v[2] + v[3]);
                rq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
            if (p.stats_out) {   // the row's 4 lane quarters hold disjoint channels: fold them, quarter 0 writes
                rs += __shfl_xor(rs, 16); rq += __shfl_xor(rq, 16);
                rs += __shfl_xor(rs, 32); rq += __shfl_xor(rq, 32);
                if (fq == 0 && row_ok) {
                    float* o = p.stats_out + ((size_t)gm * p.stats_parts + bn * WN + wn) * 2;
                    o[0] = rs;
                    o[1] = rq;
                }
            }
        }
        [[maybe_unused]] const unsigned long long t6 = PD_T_NOW();
        PD_T_ADD(c_epi, t5, t6);
    }
    PD_T_ONLY(if (threadIdx.x == 0 && g_ring_stamps) {
        unsigned long long* o = g_ring_stamps + (size_t)blockIdx.x * 8;
        o[0] = t_begin; o[1] = PD_T_NOW(); o[2] = c_wait; o[3] = c_bar; o[4] = c_issue; o[5] = c_mma; o[6] = c_epi; o[7] = (unsigned long long)nmy;
    })
}

template <int P, int BM, int BN, int WM, int WN, int NS, bool GG = false, bool PP = false>
int launch_ring(const GemmParams& p, int ncu, hipStream_t s) {
    constexpr int SMEM_BYTES = NS * (BM + BN) * BKB + BM * 8 * (PP ? 2 : 1);
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
    static unsigned long long attr_done = 0;
    auto kfn = rgemm_kernel<P, BM, BN, WM, WN, NS, GG, PP>;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), SMEM_BYTES, &attr_done)) return 1;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN, nblk = mtiles * ntiles;
    if (p.stats_out && p.stats_parts != ntiles * WN) return 1;
    int grid = nblk < ncu ? (nblk + 7) / 8 * 8 : ncu / 8 * 8;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WM * WN * 64), SMEM_BYTES, s, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace

// what rgemm_kernel covers: plain linear layers over 2-byte operands of the compute type
bool ring_gemm_eligible(const GemmParams& p, int prec) {
    if (prec != DT_F16 && prec != DT_BF16) return false;
    if (p.taps != 1 || p.a_dt != prec || p.a_silu || p.splitk > 1 || p.act == 4 || p.rowvec) return false;
    // GEGLU: whole 160-column blocks, nothing else in the epilogue
    if (p.act == 2 && (p.N % 160 || p.R || p.ln_stats || p.stats_out || p.vt_begin < p.N || p.out_scale != 1.f)) return false;
    if (p.R && (dt_size(p.r_dt) != 2 || (unsigned long long)p.M * (unsigned)p.ldr * 2ull >= (1ull << 31))) return false;   // the prefetched residual values are 2-byte, 32-bit offsets
    if (p.gate || p.c_sample_rows || p.a_sample_rows || p.a_scale || p.c_scale || p.gn_coef) return false;
    if (p.K % 64 || p.K < 128 || p.K != p.Kpad || p.M < 1 || p.N % 4 || p.N < 4) return false;
    // 32-bit byte offsets inside each operand
    if ((unsigned long long)p.M * (unsigned)p.lda * 2ull >= (1ull << 32) || (unsigned long long)p.N * (unsigned)(p.ldw ? p.ldw : p.Kpad) * 2ull >= (1ull << 32)) return false;
    return true;
}

// tile: 0 = 128 x 160 (4 stages), 1 = 256 x 160 (3 stages); both 4 x 2 waves; 2 / 3 their ping-pong forms; 4 = 64 x 80 on 4 x 1 waves (5 stages).
// GEGLU layers (act 2): 256 x 160 on 8 x 1 waves.
int launch_ring_gemm(const GemmParams& p, int prec, int tile, hipStream_t s) {
    if (!ring_gemm_eligible(p, prec)) return 1;
    static int ncu_of[64] = {};   // per device id (engines on different devices share the process)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    int& ncu = ncu_of[dev & 63];
    if (!ncu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
        ncu = prop.multiProcessorCount;
        if (ncu < 8) return 1;
    }
    if (p.act == 2) {   // 256 x 160 on 8 x 1 waves (a wave: 32 rows x one GEGLU block)
        return prec == DT_F16 ? launch_ring<DT_F16, 256, 160, 8, 1, 3, true>(p, ncu, s) : launch_ring<DT_BF16, 256, 160, 8, 1, 3, true>(p, ncu, s);
    }
    if (tile == 4) {   // 64 x 80 on 4 x 1 waves (16 rows x 80 columns per wave), 5 stages: small-M layers (the 8x8 level's M = 1024: 256 tiles instead of 64)
        return prec == DT_F16 ? launch_ring<DT_F16, 64, 80, 4, 1, 5>(p, ncu, s) : launch_ring<DT_BF16, 64, 80, 4, 1, 5>(p, ncu, s);
    }
    if (tile >= 2) {   // ping-pong forms: 2 = 128 x 160, 3 = 256 x 160
        if (prec == DT_F16) return tile == 3 ? launch_ring<DT_F16, 256, 160, 4, 2, 3, false, true>(p, ncu, s) : launch_ring<DT_F16, 128, 160, 4, 2, 4, false, true>(p, ncu, s);
        return tile == 3 ? launch_ring<DT_BF16, 256, 160, 4, 2, 3, false, true>(p, ncu, s) : launch_ring<DT_BF16, 128, 160, 4, 2, 4, false, true>(p, ncu, s);
    }
    if (prec == DT_F16) return tile ? launch_ring<DT_F16, 256, 160, 4, 2, 3>(p, ncu, s) : launch_ring<DT_F16, 128, 160, 4, 2, 4>(p, ncu, s);
    return tile ? launch_ring<DT_BF16, 256, 160, 4, 2, 3>(p, ncu, s) : launch_ring<DT_BF16, 128, 160, 4, 2, 4>(p, ncu, s);
}
