// Implicit-GEMM convolution / linear kernel for gfx950 (MI355X, CDNA4).
//
// One kernel covers every contraction of the hot path: conv3x3 (stride 1/2,
// optional fused nearest-x2 upsample gather), conv1x1, nn.Linear.  C[M,N] = A[M,K] * W[N,K]^T with
//   M = B*Hout*Wout output pixels / tokens (NHWC rows), N = Cout, K = taps*Cin.
// Tile: 128(M) x 160(N) x 128 bytes of K per step, 256 threads = 4 waves.  160 divides every channel
// count of SD1.5 (320/640/960/1280/1920/2560/5120/10240), so no N padding is wasted.
// Operands are staged global -> VGPR -> LDS (the A gather needs per-row halo masks), double-buffered,
// one barrier per K step.  All global loads of a K step are issued back to back with NO divergent
// control flow around them (out-of-image / out-of-range rows load a clamped address and are zeroed by
// a select), so the whole step's 9 x 16 B per lane is in flight under the previous step's MFMAs.
// LDS rows are 128 B with a 16-B-chunk XOR swizzle (chunk ^= (row>>1)&7) that makes the ds_read_b128
// fragment reads conflict-free.
// The MFMA is issued "swapped" (weights as the A operand, activations as B) so every lane ends up with
// 4 consecutive output channels of one pixel: bias/residual/time-embedding reads and the store are
// 8-16 B vectors along the NHWC channel axis.
// Precision modes share the byte-level data path: bf16 -> v_mfma_f32_16x16x32_bf16 (8 k per lane),
// fp32 -> 4 x v_mfma_f32_16x16x4_f32 on the same 16-byte fragment (k order permuted identically on
// both operands, which leaves the dot product unchanged).
// Small-M layers (the 8x8 and 16x16 levels) split K across blockIdx.y: each slice writes an fp32 slab
// with plain stores; the slice arriving last at the tile's counter (or, without counters,
// splitk_finalize_kernel) sums the slabs in slice order (deterministic) and applies
// the epilogue.
#include "pd_common.h"
#include "pd_mma.h"

namespace {

constexpr int BKB = 128;  // bytes of K per LDS row
// Block shapes (BM x BN): 128 x 160 (4 or 8 waves), 256 x 160 (8 waves) and 256 x 320 (8 waves, 160 accumulator
// registers per lane).  The generic kernel is bound by operand bytes fetched per FLOP (~1/BM + 1/BN) against the ~6 TB/s
// the CUs can pull from L2, so large-M layers use the biggest tile that still fills the chip.
// Two block shapes: 128 x 160 with 4 waves (2 blocks per CU) and 256 x 160 with 8 waves (1 block per CU,
// 1.4x fewer operand bytes per FLOP) for layers with enough rows to fill the chip with the big tile.

__device__ __forceinline__ int swz(int row, int chunk) { return (row * BKB) + (((chunk ^ (row >> 1)) & 7) << 4); }

// P: compute type (DT_F32: fp32 MFMA mode; DT_BF16 / DT_F16: 2-byte operands).  CONV: 3x3 gather (else rows of A
// are contiguous).  AF32: 2-byte compute with an fp32 A source (converted while staging; only meaningful when P != DT_F32).
// GG: the GEGLU epilogue (act == 2) instead of the plain one -- a wave's WTN columns are whole [80 x | 80 gate] blocks.
// MM: the MMDiT epilogue extras (pd_mma.h epilogue4<true>), linear layers of the SD3 path only.
template <int P, int BM, int BN, int WM, int WN, bool CONV, bool AF32, bool GG = false, bool MM = false>
__global__ __launch_bounds__(WM * WN * 64, (BM == 128 && WM * WN == 8) ? 4 : 2) void igemm_kernel(GemmParams p) {
    constexpr bool F32 = prec_f32_storage(P);
    // register prefetch depth: two K steps ahead (two named staging sets) unless the fp32->bf16 staging
    // path already doubles the A registers
    constexpr int DEPTH = (AF32 || BN > 192 || (BM == 128 && WM * WN == 8) || P == PREC_F16X2 || P == PREC_FP8) ? 1 : 2;   // the 16-waves-per-CU shape has 128 VGPRs per wave
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int A_ITERS = BM * 8 / NTHREADS;                   // 16-byte chunks per thread per K step
    constexpr int B_ITERS = (BN * 8 + NTHREADS - 1) / NTHREADS;  // last one masked when it does not divide
    constexpr int ROWS_PER_IT = NTHREADS / 8;
    constexpr int EB = F32 ? 4 : (P == PREC_FP8 ? 1 : 2);
    constexpr int VEC = 16 / EB;    // elements per 16-byte chunk
    constexpr int BKE = BKB / EB;   // elements of K per step
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int AEB = (F32 || AF32) ? 4 : EB;  // bytes per element of the A source
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: blocks b and b+8 share an XCD/L2; give each XCD a contiguous run of
    // tiles (n fastest) so neighbouring tiles that share the A rows hit the same L2.
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    // (a 4 x 8 super-tile order inside each XCD's run -- the ~32 blocks in flight sharing 4 A panels and 8 weight panels --
    // measured within noise on the widest layers, 24-32 tiles across, and was dropped)
    const int bm = bid / ntiles, bn = bid % ntiles;
    const int kslice = blockIdx.y;

    // ---- per-thread staging assignment
    const int chunk = tid & 7;
    const int row0 = tid >> 3;  // + ROWS_PER_IT*i
    int a_pix[A_ITERS];         // CONV: sample * Hin (row base); else unused
    int a_y[A_ITERS], a_x[A_ITERS];
    bool a_ok[A_ITERS];
    size_t a_base[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int m = bm * BM + row0 + ROWS_PER_IT * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if constexpr (CONV) {
            const int b = mm / p.rows_per_sample;
            const int rem = mm - b * p.rows_per_sample;
            const int oy = rem / p.Wout;
            a_pix[i] = b * p.Hin;
            a_y[i] = oy * p.stride - 1;
            a_x[i] = (rem - oy * p.Wout) * p.stride - 1;
            a_base[i] = 0;
        } else {
            a_base[i] = (size_t)mm * p.lda;
            if constexpr (MM) {
                if (p.a_sample_rows) {   // this token stream's rows inside a joint [sample][a_sample_rows] buffer
                    const int b = mm / p.rows_per_sample;
                    a_base[i] = (size_t)(b * p.a_sample_rows + p.a_row_off + (mm - b * p.rows_per_sample)) * p.lda;
                }
            }
            a_pix[i] = a_y[i] = a_x[i] = 0;
        }
    }
    const char* w_ptr[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        int n = bn * BN + row0 + ROWS_PER_IT * i;
        n = n < p.N ? n : p.N - 1;  // clamp: columns >= N are never stored
        w_ptr[i] = reinterpret_cast<const char*>(p.W) + ((size_t)n * (p.ldw ? p.ldw : p.Kpad) + chunk * VEC) * EB;
    }

    // folded LayerNorm: {mean, rstd} of this block's BM rows, combined once from the producer's partials into LDS behind the
    // staging buffers (the epilogue reads two floats per row instead of walking the partials in every lane)
    constexpr bool LNF = !GG && !MM;   // not in the GEGLU instantiations (nor the MMDiT ones: no LayerNorm fold on that path) (160 accumulator registers at the VGPR cap): norm3 stays a kernel
    float2* sLn = reinterpret_cast<float2*>(smem + 2 * (BM + BN) * BKB);
    if constexpr (LNF) {
        if (p.ln_stats && tid < BM) {
            const int gm = bm * BM + tid;
            float mean = 0.f, rstd = 0.f;
            if (gm < p.M) ln_row_stats(p, gm, mean, rstd);
            sLn[tid] = make_float2(mean, rstd);
        }
    }
    const int ktiles_all = p.Kpad / BKE;
    int kt0 = 0, kt1 = ktiles_all;
    if (p.splitk > 1) {
        const int per = (ktiles_all + p.splitk - 1) / p.splitk;
        kt0 = kslice * per;
        kt1 = min(ktiles_all, kt0 + per);
    }

    // Staging registers are NAMED scalars reached through constexpr selectors: hipcc leaves small indexed
    // arrays captured by these lambdas in scratch memory, which serialises every prefetch behind a wait.
    static_assert(A_ITERS <= 4 && B_ITERS <= 6, "staging registers below cover 4 A and 6 B pieces");
    uint4 ra0, ra1, ra2, ra3, rh0, rh1, rh2, rh3, rb0, rb1, rb2, rb3, rb4, rb5;   // set 0
    uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3, sb4, sb5;                  // set 1 (DEPTH 2 only)
    auto RA = [&](auto S, auto I) __attribute__((always_inline)) -> uint4& {
        constexpr int i = decltype(I)::value;
        if constexpr (decltype(S)::value == 0) {
            if constexpr (i == 0) return ra0; else if constexpr (i == 1) return ra1; else if constexpr (i == 2) return ra2; else return ra3;
        } else {
            if constexpr (i == 0) return sa0; else if constexpr (i == 1) return sa1; else if constexpr (i == 2) return sa2; else return sa3;
        }
    };
    auto RH = [&](auto I) __attribute__((always_inline)) -> uint4& {  // second 16 B of an fp32 A source (AF32, set 0 only)
        constexpr int i = decltype(I)::value;
        if constexpr (i == 0) return rh0; else if constexpr (i == 1) return rh1; else if constexpr (i == 2) return rh2; else return rh3;
    };
    auto RB = [&](auto S, auto I) __attribute__((always_inline)) -> uint4& {
        constexpr int i = decltype(I)::value;
        if constexpr (decltype(S)::value == 0) {
            if constexpr (i == 0) return rb0; else if constexpr (i == 1) return rb1; else if constexpr (i == 2) return rb2;
            else if constexpr (i == 3) return rb3; else if constexpr (i == 4) return rb4; else return rb5;
        } else {
            if constexpr (i == 0) return sb0; else if constexpr (i == 1) return sb1; else if constexpr (i == 2) return sb2;
            else if constexpr (i == 3) return sb3; else if constexpr (i == 4) return sb4; else return sb5;
        }
    };
    unsigned rokmask1 = 0;
    unsigned rokmask = 0;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const char* Ab = reinterpret_cast<const char*>(p.A);

    // issue every global load of K step kt; nothing here branches per lane
    auto gload = [&](auto S, int kt) __attribute__((always_inline)) {
        unsigned& rmask = decltype(S)::value == 0 ? rokmask : rokmask1;
        const int k0 = kt * BKE + chunk * VEC;
        const bool kok = k0 < p.K;
        int ky = 0, kx = 0, cof = k0;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;
            cof = k0 - tap * p.Cin;
            ky = tap / 3;
            kx = tap - ky * 3;
        }
        rmask = 0;
        static_for<A_ITERS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            bool ok = a_ok[i] && kok;
            size_t idx;
            if constexpr (CONV) {
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                ok = ok && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                const int pix = (a_pix[i] + (iy >> p.ups)) * p.Win + (ix >> p.ups);
                idx = (size_t)(ok ? pix : 0) * p.lda + cof;
            } else {
                idx = a_base[i] + cof;
            }
            idx = ok ? idx : 0;
            rmask |= ok ? (1u << i) : 0u;
            const uint4* s = reinterpret_cast<const uint4*>(Ab + idx * AEB);
            RA(S, I) = s[0];
            if constexpr (AEB == 4 && !F32) RH(I) = s[1];
        });
        static_for<B_ITERS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            RB(S, I) = *reinterpret_cast<const uint4*>(w_ptr[i] + (size_t)kt * BKE * EB);
        });
    };
    auto lstore = [&](auto S, int buf) __attribute__((always_inline)) {
        const unsigned rmask = decltype(S)::value == 0 ? rokmask : rokmask1;
        char* sa = smem + buf * (BM + BN) * BKB;
        char* sb = sa + BM * BKB;
        static_for<A_ITERS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            uint4 v;
            if constexpr (F32) {
                v = RA(S, I);
                if (p.a_silu) {
                    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = silu_f(f[j]);
                }
            } else if constexpr (AF32) {
                v = cvt8<P>(RA(S, I), RH(I), p.a_silu != 0);
            } else {
                v = RA(S, I);
            }
            if (!(rmask & (1u << i))) v = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(sa + swz(row0 + ROWS_PER_IT * i, chunk)) = v;
        });
        static_for<B_ITERS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if (BN % ROWS_PER_IT == 0 || row0 + ROWS_PER_IT * i < BN)
                *reinterpret_cast<uint4*>(sb + swz(row0 + ROWS_PER_IT * i, chunk)) = RB(S, I);
        });
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* sa = smem + buf * (BM + BN) * BKB;
        const char* sb = sa + BM * BKB;
        if constexpr (P == PREC_FP8) {   // one K = 128 block-scaled MFMA per accumulator and K step
            // fragments are 32 bytes per lane here: all of A's stay live, the weights' are read one column block at a time
            i32x8 af[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                af[m] = prep_f8(*reinterpret_cast<const uint4*>(sa + swz(wm * WTM + m * 16 + fr, fq)),
                                *reinterpret_cast<const uint4*>(sa + swz(wm * WTM + m * 16 + fr, 4 + fq)));
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const i32x8 wf = prep_f8(*reinterpret_cast<const uint4*>(sb + swz(wn * WTN + n * 16 + fr, fq)),
                                         *reinterpret_cast<const uint4*>(sb + swz(wn * WTN + n * 16 + fr, 4 + fq)));
#pragma unroll
                for (int m = 0; m < MT; ++m) mma_f8(wf, af[m], acc[n][m]);
            }
            return;
        }
        if constexpr (P == PREC_F16X2) {   // both half-steps at once: 3 MFMAs per accumulator and K step (pd_mma.h)
            FragX2 af[MT], wf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                af[m] = prep_x2(*reinterpret_cast<const uint4*>(sa + swz(wm * WTM + m * 16 + fr, fq)),
                                *reinterpret_cast<const uint4*>(sa + swz(wm * WTM + m * 16 + fr, 4 + fq)));
#pragma unroll
            for (int n = 0; n < NT; ++n)
                wf[n] = prep_x2(*reinterpret_cast<const uint4*>(sb + swz(wn * WTN + n * 16 + fr, fq)),
                                *reinterpret_cast<const uint4*>(sb + swz(wn * WTN + n * 16 + fr, 4 + fq)));
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) mma_x2(wf[n], af[m], acc[n][m]);
            return;
        }
        if constexpr (P != PREC_FP8 && P != PREC_F16X2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            typename Frag<P>::A af[MT];
            typename Frag<P>::W wf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = prep_a<P>(*reinterpret_cast<const uint4*>(sa + swz(wm * WTM + m * 16 + fr, ks * 4 + fq)));
#pragma unroll
            for (int n = 0; n < NT; ++n) wf[n] = prep_w<P>(*reinterpret_cast<const uint4*>(sb + swz(wn * WTN + n * 16 + fr, ks * 4 + fq)));
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) mma<P>(wf[n], af[m], acc[n][m]);
        }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if constexpr (DEPTH == 1) {
        if (kt0 < kt1) {
            gload(S0{}, kt0);
            lstore(S0{}, 0);
        }
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            const bool more = kt + 1 < kt1;
            if (more) gload(S0{}, kt + 1);
            compute(buf);
            if (more) lstore(S0{}, buf ^ 1);
            __syncthreads();
        }
    } else {
        // tile kt is computed from LDS buffer (kt-kt0)&1 while tile kt+1 waits in the other register set and
        // tile kt+2 is being requested: every global load has two K steps of MFMAs to land
        if (kt0 < kt1) {
            gload(S0{}, kt0);
            lstore(S0{}, 0);
            if (kt0 + 1 < kt1) gload(S1{}, kt0 + 1);
        }
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt += 2) {
            if (kt + 2 < kt1) gload(S0{}, kt + 2);
            compute(0);
            if (kt + 1 < kt1) lstore(S1{}, 1);
            __syncthreads();
            if (kt + 1 >= kt1) break;
            if (kt + 3 < kt1) gload(S1{}, kt + 3);
            compute(1);
            if (kt + 2 < kt1) lstore(S0{}, 0);
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds channels n..n+3 (rows of the swapped MFMA) of pixel m
    if (p.splitk > 1) {
        if (!p.tile_cnt) {   // plain slab stores; splitk_finalize_kernel sums them after this launch
            float* slab = reinterpret_cast<float*>(p.slab) + (size_t)kslice * p.M * p.N;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gm = bm * BM + wm * WTM + m * 16 + fr;
                if (gm >= p.M) continue;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int gn = bn * BN + wn * WTN + n * 16 + fq * 4;
                    if (gn >= p.N) continue;
                    *reinterpret_cast<f32x4*>(slab + (size_t)gm * p.N + gn) = acc[n][m];
                }
            }
            return;
        }
        // Fused finalize: the slice that arrives last at this tile's counter sums all slabs in slice order
        // (bit-identical to splitk_finalize_kernel, independent of arrival order) and runs the epilogue.
        // Cross-XCD visibility without L2 flushes: every slab byte is stored write-through (sc1) and every
        // slab load is an sc1 load; the counter is a relaxed agent-scope atomic.  The "last" flag travels
        // through the staging LDS (free after the K loop's closing barrier).
        if constexpr (BM == 128 && WM == 2 && WN == 2) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.slab, 0, (int)((size_t)p.splitk * p.M * p.N * 4), 0x00020000);
            constexpr int SC1 = 16;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gm = bm * BM + wm * WTM + m * 16 + fr;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int gn = bn * BN + wn * WTN + n * 16 + fq * 4;
                    if (gm >= p.M || gn >= p.N) continue;
                    const unsigned off = (unsigned)(((kslice * p.M + gm) * p.N + gn) * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[n][m]), rs, off, 0, SC1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores
            __syncthreads();
            int* s_last = reinterpret_cast<int*>(smem);
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(p.tile_cnt + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == p.splitk - 1;
                if (last) __hip_atomic_store(p.tile_cnt + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next GEMM
                *s_last = last;
            }
            __syncthreads();
            if (!*s_last) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: keep the slab loads below the counter
            for (int s = 0; s < p.splitk; ++s) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int gm = bm * BM + wm * WTM + m * 16 + fr;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int gn = bn * BN + wn * WTN + n * 16 + fq * 4;
                        if (gm >= p.M || gn >= p.N) continue;
                        const unsigned off = (unsigned)(((s * p.M + gm) * p.N + gn) * 4);
                        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, SC1));
                        acc[n][m] = s == 0 ? t : acc[n][m] + t;
                    }
                }
            }
        } else {
            return;
        }
    }
    // (the 256 x 320 tile keeps reads and stores interleaved: 160 accumulator registers at the VGPR cap, the hoisted reads spill)
    constexpr bool TWO_PASS = !GG && !(BM == 256 && BN == 320);
    if constexpr (TWO_PASS) {
        // pass 1 -- values: every read of the epilogue (bias, time-embedding row, residual, LayerNorm column sums), for all of
        // the lane's groups, before the first store (pd_mma.h: a read issued behind a store waits for that store)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int gm = min(bm * BM + wm * WTM + m * 16 + fr, p.M - 1);   // rows past M: a clamped copy, never stored
            const int sample = gm / p.rows_per_sample;
            // folded LayerNorm / row statistics: every tile except the GEGLU-capable ones (8 x 1 waves, 160 accumulator
            // registers at the VGPR cap: the extra epilogue state spills their accumulators -- norm3 stays a kernel of its own)
            float ln_mean = 0.f, ln_rstd = 0.f;
            if constexpr (LNF) {
                if (p.ln_stats) {
                    const float2 t = sLn[wm * WTM + m * 16 + fr];
                    ln_mean = t.x;
                    ln_rstd = t.y;
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int gn = min(bn * BN + wn * WTN + n * 16 + fq * 4, p.N - 4);
                acc[n][m] = epilogue4_value<MM>(p, gm, gn, sample, acc[n][m], ln_mean, ln_rstd);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int gm = bm * BM + wm * WTM + m * 16 + fr;
        if (gm >= p.M) continue;
        const int sample = gm / p.rows_per_sample;
        const int tok = gm - sample * p.rows_per_sample;
        if constexpr (GG) {
            // GEGLU: virtual columns [0,80) of each 160-column block are x, [80,160) the gate (weights interleaved at load)
            static_assert(NT % 10 == 0, "a wave's columns must be whole 160-column GEGLU blocks");
#pragma unroll
            for (int sb = 0; sb < NT / 10; ++sb) {
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    const int vn = bn * BN + wn * WTN + sb * 160 + n * 16 + fq * 4;
                    const int on = ((bn * BN + wn * WTN) / 160 + sb) * 80 + n * 16 + fq * 4;
                    if (on >= p.Nout) continue;
                    f32x4 x = acc[sb * 10 + n][m], g = acc[sb * 10 + n + 5][m];
                    if (p.bias) {
                        x += *reinterpret_cast<const f32x4*>(p.bias + vn);
                        g += *reinterpret_cast<const f32x4*>(p.bias + vn + 80);
                    }
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = x[j] * (F32 ? gelu_f(g[j]) : gelu_fast(g[j]));
                    store4(p.C, (size_t)gm * p.ldc + on, p.c_dt, o);
                }
            }
            continue;
        }
        // pass 2 -- stores (+ producer side: this row's {sum, sum of squares} over the wave's column range)
        float rs = 0.f, rq = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int gn = bn * BN + wn * WTN + n * 16 + fq * 4;
            if (gn >= p.N) continue;
            f32x4 v = acc[n][m];
            if constexpr (TWO_PASS) {
                epilogue4_store<MM>(p, gm, gn, sample, tok, v);
            } else {
                float ln_mean = 0.f, ln_rstd = 0.f;
                if constexpr (LNF) {
                    if (p.ln_stats) {
                        const float2 t = sLn[wm * WTM + m * 16 + fr];
                        ln_mean = t.x;
                        ln_rstd = t.y;
                    }
                }
                v = epilogue4<MM>(p, gm, gn, sample, tok, v, ln_mean, ln_rstd);
            }
            if constexpr (LNF) {
                if (p.stats_out) {
                    rs += (v[0] + v[1]) + (v[2] + v[3]);
                    rq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
        }
        if (LNF && p.stats_out) {   // the row's 4 lane quarters (fq) hold disjoint channels: fold them, lane quarter 0 writes
            rs += __shfl_xor(rs, 16); rq += __shfl_xor(rq, 16);
            rs += __shfl_xor(rs, 32); rq += __shfl_xor(rq, 32);
            if (fq == 0) {
                float* o = p.stats_out + ((size_t)gm * p.stats_parts + bn * WN + wn) * 2;
                o[0] = rs;
                o[1] = rq;
            }
        }
    }
}

// sums the split-K slabs in slice order and applies the epilogue; one thread per 4 channels
__global__ __launch_bounds__(256) void splitk_finalize_kernel(GemmParams p) {
    const int n4 = p.N / 4;
    const long long total = (long long)p.M * n4;
    const float* slab = reinterpret_cast<const float*>(p.slab);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int gm = (int)(i / n4);
        const int gn = (int)(i - (long long)gm * n4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + (size_t)gm * p.N + gn);
        for (int s = 1; s < p.splitk; ++s) v += *reinterpret_cast<const f32x4*>(slab + ((size_t)s * p.M + gm) * p.N + gn);
        const int sample = gm / p.rows_per_sample;
        float ln_mean = 0.f, ln_rstd = 0.f;
        if (p.ln_stats) ln_row_stats(p, gm, ln_mean, ln_rstd);
        epilogue4<true>(p, gm, gn, sample, gm - sample * p.rows_per_sample, v, ln_mean, ln_rstd);
    }
}

template <int P, int BM, int BN, int WM, int WN, bool CONV, bool AF32, bool GG = false, bool MM = false>
int launch_one(const GemmParams& p, hipStream_t s, hipEvent_t mid) {
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int SMEM_BYTES = 2 * (BM + BN) * BKB + ((GG || MM) ? 0 : BM * 8);   // staging buffers + {mean, rstd} of the block's rows (LayerNorm fold)
    static unsigned long long attr_done = 0;
    auto kfn = igemm_kernel<P, BM, BN, WM, WN, CONV, AF32, GG, MM>;
    if (ensure_dyn_smem(reinterpret_cast<const void*>(kfn), SMEM_BYTES, &attr_done)) return 1;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
    if (p.stats_out && p.stats_parts != ntiles * WN) return 1;   // the caller sized the statistics rows for another tile
    dim3 grid(mtiles * ntiles, p.splitk > 1 ? p.splitk : 1);
    hipLaunchKernelGGL(kfn, grid, dim3(NTHREADS), SMEM_BYTES, s, p);
    if (hipGetLastError() != hipSuccess) return 1;
    if (mid) (void)hipEventRecord(mid, s);   // profiling: end of the contraction kernel proper
    if (p.splitk > 1 && !p.tile_cnt && !p.defer_finalize) {
        long long total = (long long)p.M * (p.N / 4);
        int nb = (int)((total + 255) / 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(splitk_finalize_kernel, dim3(nb), dim3(256), 0, s, p);
        if (hipGetLastError() != hipSuccess) return 1;
    }
    return 0;
}

}  // namespace

int launch_splitk_finalize(const GemmParams& p, hipStream_t s) {
    if (p.splitk <= 1 || !p.slab || p.N % 4 || p.act == 2 || p.vt_begin < p.N) return 1;
    long long total = (long long)p.M * (p.N / 4);
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(nb), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int gemm_tiles(int M, int N) { return ((M + 127) / 128) * ((N + 159) / 160); }

namespace {
// PREC_FP8: linear layers with the MMDiT epilogue only (the SD3 path's QKV and feed-forward-in projections)
int launch_fp8(const GemmParams& p, hipStream_t s, hipEvent_t mid) {
    if (p.taps != 1 || p.a_dt != DT_FP8 || !p.a_scale || !p.w_scale || p.act == 2 || p.Kpad % 128) return 1;
    if (p.splitk > 1) return launch_one<PREC_FP8, 128, 160, 2, 2, false, false, false, true>(p, s, mid);   // fp32 slabs; the finalize pass applies the scales
    if (p.big_tile == 4) return launch_one<PREC_FP8, 256, 192, 4, 2, false, false, false, true>(p, s, mid);
    if (p.big_tile == 1 || p.big_tile == 3) return launch_one<PREC_FP8, 256, 160, 4, 2, false, false, false, true>(p, s, mid);
    return launch_one<PREC_FP8, 128, 160, 2, 2, false, false, false, true>(p, s, mid);
}

template <int P>
int launch_prec(const GemmParams& p, hipStream_t s, hipEvent_t mid) {
    constexpr bool F = prec_f32_storage(P);
    const bool conv = p.taps != 1;
    const bool af32 = !F && p.a_dt == DT_F32;          // 2-byte compute reading an fp32 A (converted while staging)
    // 0: 128x160, 1: 256x160, 2: 128x160 on 8 waves, 3: 256x320; split-K launches: 128x160, or 256x160 when the engine asks for it
    // (plain slab stores work from any tile; the in-kernel finalize of option splitk_fused exists for the 128x160 tile only)
    const int tile = p.splitk == 1 ? p.big_tile : ((p.big_tile == 1 && !p.tile_cnt) ? 1 : 0);
    if (p.act == 2) {
        if (conv || af32 || p.ln_stats || p.stats_out) return 1;
        // 256 x 320 on 4 x 2 waves (wave tile 64 x 160 = one GEGLU block): 14 fragment reads per 40 MFMAs.  Round 1 ran it on
        // 8 x 1 waves (32 x 320: 22 reads per 40 MFMAs, every wave re-reading all weight fragments), which made the tile
        // LDS-read bound (352 KB per K step = 1375 LDS cycles against 1280 MFMA cycles).
        if (tile == 3) return launch_one<P, 256, 320, 4, 2, false, false, true>(p, s, mid);
        if (tile == 1) return launch_one<P, 256, 160, 8, 1, false, false, true>(p, s, mid);
        return launch_one<P, 128, 160, 4, 1, false, false, true>(p, s, mid);
    }
    if constexpr (!F) {
        // 256 x 192 on 4 x 2 waves: the tile of widths that are multiples of 192 but not of 160 (MMDiT: 1536 = 8 x 192,
        // 4608, 6144) -- no padded columns, and 8192 rows x 1536 columns is exactly one block per CU
        if (tile == 4) {
            if (conv || af32) return 1;
            if (p.act == 4 || p.gate || p.c_sample_rows || p.a_sample_rows) return launch_one<P, 256, 192, 4, 2, false, false, false, true>(p, s, mid);
            return launch_one<P, 256, 192, 4, 2, false, false, false, false>(p, s, mid);
        }
    }
    // (measured and dropped: 128 x 192 four-wave blocks, two per CU so that one block's epilogue runs under the other's K loop
    // -- 43 % more operand bytes per FLOP through L2 cost more than the overlap gains: SD3 step 36.1 -> 37.4 ms)
    if (p.act == 4 || p.gate || p.c_sample_rows || p.a_sample_rows) {   // MMDiT epilogue extras
        if (p.ln_stats || p.stats_out) return 1;   // no LayerNorm fold in the MM instantiations: linear layers over operands of the compute type
        if (conv || af32) return 1;
        if constexpr (F) {
            if (tile == 1 || tile == 3) return launch_one<P, 256, 160, 4, 2, false, false, false, true>(p, s, mid);
            return launch_one<P, 128, 160, 2, 2, false, false, false, true>(p, s, mid);
        } else {
            // (the 256 x 320 tile spills 896 B per lane with the extras compiled in: its launches take the 256 x 160 tile)
            if (tile == 2) return launch_one<P, 128, 160, 4, 2, false, false, false, true>(p, s, mid);
            if (tile == 1 || tile == 3) return launch_one<P, 256, 160, 4, 2, false, false, false, true>(p, s, mid);
            return launch_one<P, 128, 160, 2, 2, false, false, false, true>(p, s, mid);
        }
    }
    if constexpr (F) {   // fp32 mode: the 128x160 / 256x160 four-wave-group tiles only
        if (tile == 1 || tile == 3) return conv ? launch_one<P, 256, 160, 4, 2, true, false>(p, s, mid) : launch_one<P, 256, 160, 4, 2, false, false>(p, s, mid);
        return conv ? launch_one<P, 128, 160, 2, 2, true, false>(p, s, mid) : launch_one<P, 128, 160, 2, 2, false, false>(p, s, mid);
    } else {
        if (tile == 3 && !conv && !af32) return launch_one<P, 256, 320, 4, 2, false, false>(p, s, mid);
        if (tile == 2 && !conv && !af32) return launch_one<P, 128, 160, 4, 2, false, false>(p, s, mid);
        if (tile == 1 || tile == 3) {
            if (conv) return af32 ? launch_one<P, 256, 160, 4, 2, true, true>(p, s, mid) : launch_one<P, 256, 160, 4, 2, true, false>(p, s, mid);
            return af32 ? launch_one<P, 256, 160, 4, 2, false, true>(p, s, mid) : launch_one<P, 256, 160, 4, 2, false, false>(p, s, mid);
        }
        if (conv) return af32 ? launch_one<P, 128, 160, 2, 2, true, true>(p, s, mid) : launch_one<P, 128, 160, 2, 2, true, false>(p, s, mid);
        return af32 ? launch_one<P, 128, 160, 2, 2, false, true>(p, s, mid) : launch_one<P, 128, 160, 2, 2, false, false>(p, s, mid);
    }
}
}  // namespace

int launch_gemm(const GemmParams& p, int prec, hipStream_t s, hipEvent_t mid) {
    if (p.M <= 0 || p.N <= 0) return 0;
    if (p.splitk > 1 && (p.act == 2 || p.vt_begin < p.N || !p.slab || p.N % 4)) return 1;
    if (prec == PREC_FP8) return launch_fp8(p, s, mid);
    if (prec_f32_storage(prec) ? p.a_dt != DT_F32 : (p.a_dt != DT_F32 && p.a_dt != prec)) return 1;   // operand type must match the mode
    switch (prec) {
        case DT_F32: return launch_prec<DT_F32>(p, s, mid);
        case PREC_F16X2: return launch_prec<PREC_F16X2>(p, s, mid);
        case DT_BF16: return launch_prec<DT_BF16>(p, s, mid);
        case DT_F16: return launch_prec<DT_F16>(p, s, mid);
        default: return 1;
    }
}
