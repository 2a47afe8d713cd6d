// Large-tile bf16 GEMM for the big Linear / conv1x1 layers:  C[M,N] = epilogue( A[M,K] x W[N,K]^T ).
//
// 256 x 256 block tile, K step 64 (128-byte LDS rows), 8 waves as 2 (M) x 4 (N): each wave owns 128 rows x 64
// channels = 32 accumulator fragments (128 VGPRs).  Both operands reach LDS by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, no ds_write): a wave-instruction fills 8 consecutive 128-byte rows, and the XOR swizzle that
// makes the ds_read_b128 fragment reads conflict-free (chunk ^ (row>>1)&7, same as gemm.hip) is applied to the
// per-lane SOURCE address.  Two LDS stages of 64 KB (A lo/hi, W lo/hi halves of 128 rows): the DMA of K tile k+1 is
// issued right after the barrier that opens K tile k and is waited for, with a counted vmcnt, a whole K tile of
// MFMAs later.  One raw s_barrier per K tile; inside a K tile the wave walks its 4 accumulator quadrants
// (64 rows x 32 channels, 16 MFMAs each) reloading only the operand that changes, so the two waves of a SIMD
// drift into load / MFMA ping-pong on their own.
// Like gemm.hip the MFMA is issued "swapped" (weights = A operand): a lane ends with 4 consecutive output channels
// of one row, and the epilogue (bias / time-embedding row / SiLU / scale / residual / V^T store) is the shared one.
#include "pd_common.h"
#include "pd_mma.h"

namespace {

constexpr int T8 = 256;                    // block tile edge (rows and channels)
constexpr int ROWB = 128;                  // bytes per LDS row = 64 bf16 of K
constexpr int HALF_BYTES = 128 * ROWB;     // 16 KB: 128 rows of one operand
constexpr int STAGE_BYTES = 4 * HALF_BYTES;

using lds_t = __attribute__((address_space(3))) char*;
using gptr_t = const __attribute__((address_space(1))) char*;

__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmParams p) {
    // Two DISTINCT LDS objects, one per stage: the compiler tags accesses with the object they touch, which is what
    // lets it keep a ds_read of one stage from waiting (vmcnt(0)) for the LDS-DMA still filling the other.
    __shared__ __attribute__((aligned(1024))) char stage0[STAGE_BYTES];
    __shared__ __attribute__((aligned(1024))) char stage1[STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    // XCD-aware tile order (see gemm.hip)
    const int mtiles = (p.M + T8 - 1) / T8, ntiles = (p.N + T8 - 1) / T8;
    const int nblk = mtiles * ntiles;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int bm = bid / ntiles, bn = bid % ntiles;

    // ---- LDS-DMA sources: per half-tile (128 rows) every wave moves 2 pieces of 8 rows; lane -> (row, chunk slot)
    const int ldw = p.ldw ? p.ldw : p.Kpad;
    gptr_t srcA[2][2], srcW[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wave * 2 + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int m = bm * T8 + h * 128 + r;
            m = m < p.M ? m : p.M - 1;          // rows >= M are never stored
            int n = bn * T8 + h * 128 + r;
            n = n < p.N ? n : p.N - 1;
            srcA[h][j] = (gptr_t)(reinterpret_cast<const char*>(p.A) + ((size_t)m * p.lda + c * 8) * 2);
            srcW[h][j] = (gptr_t)(reinterpret_cast<const char*>(p.W) + ((size_t)n * ldw + c * 8) * 2);
        }
    auto issue = [&](int kt, char* stage) __attribute__((always_inline)) {
        char* base = stage + wave * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                __builtin_amdgcn_global_load_lds(srcA[h][j] + (size_t)kt * ROWB, (lds_t)(base + h * HALF_BYTES + j * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(srcW[h][j] + (size_t)kt * ROWB, (lds_t)(base + (2 + h) * HALF_BYTES + j * 1024), 16, 0, 0);
            }
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment offsets inside a half-tile: row fr (+16 per fragment), swizzled chunk for ks = 0, 1
    const int sw = (fr >> 1) & 7;
    const int foff0 = fr * ROWB + ((fq ^ sw) * 16);
    const int foff1 = fr * ROWB + (((4 + fq) ^ sw) * 16);
    const int aoff = wm * HALF_BYTES;                                   // this wave's 128 rows
    const int woff = (2 + (wn >> 1)) * HALF_BYTES + (wn & 1) * 64 * ROWB;   // this wave's 64 channels

    auto compute = [&](const char* st) __attribute__((always_inline)) {
        const char* sA = st + aoff;
        const char* sW = st + woff;
        uint4 af[4][2], wf[2][2];
        auto loadA = [&](int mq) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *reinterpret_cast<const uint4*>(sA + (mq * 4 + i) * 16 * ROWB + foff0);
                af[i][1] = *reinterpret_cast<const uint4*>(sA + (mq * 4 + i) * 16 * ROWB + foff1);
            }
        };
        auto loadW = [&](int nq) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i][0] = *reinterpret_cast<const uint4*>(sW + (nq * 2 + i) * 16 * ROWB + foff0);
                wf[i][1] = *reinterpret_cast<const uint4*>(sW + (nq * 2 + i) * 16 * ROWB + foff1);
            }
        };
        auto quad = [&](auto MQ, auto NQ) __attribute__((always_inline)) {
            constexpr int mq = decltype(MQ)::value, nq = decltype(NQ)::value;
            __builtin_amdgcn_s_setprio(1);   // the wave that has its fragments keeps the matrix pipe; its SIMD partner loads
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 4; ++m) mma<false>(wf[n][ks], af[m][ks], acc[nq * 2 + n][mq * 4 + m]);
            __builtin_amdgcn_s_setprio(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        loadW(0); loadA(0); quad(I0{}, I0{});
        loadW(1);           quad(I0{}, I1{});
        loadA(1);           quad(I1{}, I1{});
        loadW(0);           quad(I1{}, I0{});
    };
    // K tile kt has landed (this wave's pieces: vmcnt; everybody's: barrier) and the other stage is free again
    auto open_tile = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only (a real S_WAITCNT: the compiler's own wait tracking sees it)
        __builtin_amdgcn_s_barrier();
    };

    const int nkt = p.Kpad / 64;
    issue(0, stage0);
    for (int kt = 0; kt < nkt; kt += 2) {
        open_tile();
        if (kt + 1 < nkt) issue(kt + 1, stage1);
        compute(stage0);
        if (kt + 1 >= nkt) break;
        open_tile();
        if (kt + 2 < nkt) issue(kt + 2, stage0);
        compute(stage1);
    }

    // ---- epilogue: lane holds channels gn..gn+3 of row gm
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int gm = bm * T8 + wm * 128 + m * 16 + fr;
        if (gm >= p.M) continue;
        const int sample = gm / p.rows_per_sample;
        const int tok = gm - sample * p.rows_per_sample;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int gn = bn * T8 + wn * 64 + n * 16 + fq * 4;
            if (gn >= p.N) continue;
            epilogue4(p, gm, gn, sample, tok, acc[n][m]);
        }
    }
}

}  // namespace

bool gemm8_eligible(const GemmParams& p) {
    const int ldw = p.ldw ? p.ldw : p.Kpad;
    return p.taps == 1 && p.a_dt == DT_BF16 && p.act != 2 && p.splitk <= 1 && p.K == p.Kpad && p.Kpad % 64 == 0 && p.lda >= p.Kpad &&
           p.lda % 8 == 0 && ldw % 8 == 0 && p.N % 4 == 0 && !p.diag && !p.gn_coef;
}

int launch_gemm8(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + T8 - 1) / T8) * ((p.N + T8 - 1) / T8);
    hipLaunchKernelGGL(gemm8_kernel, dim3(tiles), dim3(512), 0, s, p);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
