"""Per-kernel parity on the GPU: each HIP kernel family (through the C ABI's pd_op_* hooks) against
the NumPy oracle on the same seeded inputs.  Tolerances: fp32 mode 2e-5 of the tensor's max
(different summation order only); bf16 mode 2e-2 (8-bit mantissa operands, fp32 accumulate); fp16 mode 2.5e-3
(11-bit mantissa operands: 8x tighter than bf16); split-fp16 mode (f16x2: hi + lo operand pairs over fp32 storage) is held
to the fp32 bounds."""
import numpy as np
import pytest

from oracle import pd_oracle as O
from prompt_diffusion_amd import engine as E
from prompt_diffusion_amd import weights as W

pytestmark = pytest.mark.gpu
TOL = {"f32": 2e-5, "bf16": 2e-2, "f16": 2.5e-3, "f16x2": 2e-5}
NORM_TOL = {"f32": 1e-5, "bf16": 1e-2, "f16": 1.5e-3, "f16x2": 1e-5}   # output rounding of the 2-byte modes
ATTN_TOL = {"f32": 3e-5, "bf16": 2e-2, "f16": 2.5e-3, "f16x2": 3e-5}


def round_like(x, prec):
    """the kernel sees the 2-byte-rounded stream; give the oracle the same input"""
    if prec == "bf16":
        import torch
        return torch.from_numpy(x).bfloat16().float().numpy()
    if prec == "f16":
        return x.astype(np.float16).astype(np.float32)
    return x


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module", params=["f32", "bf16", "f16", "f16x2"])
def eng(request):
    e = E.Engine(W.TINY, precision=request.param)
    e.prec = request.param
    yield e
    e.close()


def rng(seed):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride,ups", [
    (2, 64, 8, 8, 128, 3, 1, False),     # plain 3x3
    (1, 4, 16, 16, 64, 3, 1, False),     # conv_in: Cin padded 4 -> 8
    (2, 6, 32, 32, 16, 3, 1, False),     # hint conv: tiny channels, K = 72
    (2, 32, 10, 10, 32, 3, 2, False),    # Downsample (stride 2, even)
    (1, 32, 9, 7, 48, 3, 2, False),      # stride 2, odd sizes
    (2, 32, 6, 6, 32, 3, 1, True),       # Upsample: nearest x2 fused in the gather
    (2, 96, 12, 12, 96, 1, 1, False),    # 1x1
    (3, 320, 16, 16, 320, 3, 1, False),  # SD1.5 shape, N = 2 x 160 tiles, M = 768
    (1, 320, 8, 8, 4, 3, 1, False),      # out conv: N = 4
    (1, 200, 5, 5, 168, 3, 1, False),    # ragged M and N tiles, Cin not a multiple of the K tile
    (12, 64, 32, 32, 320, 3, 1, False),  # LDS-patch kernel: 48 patches x 2 n-tiles... (>=192 blocks: 12*4*4)
    (13, 128, 16, 48, 168, 3, 1, False), # LDS-patch kernel, non-square, ragged N tile
    (50, 64, 16, 16, 96, 3, 1, True),    # LDS-patch kernel with fused nearest-x2 upsample (input 16x16 -> 32x32)
    (16, 1280, 16, 16, 1280, 3, 1, False),  # the headline workload's 16x16-level conv: LDS-patch kernel, channel chunks split over 2 slices
])
def test_conv2d(eng, B, Cin, H, W, Cout, k, stride, ups):
    g = rng(1)
    x = g.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = (g.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = g.standard_normal(Cout, dtype=np.float32) * 0.1
    xin = np.repeat(np.repeat(x, 2, axis=2), 2, axis=3) if ups else x
    ref = O.conv2d(xin, w, b, stride=stride, padding=k // 2)
    got = eng.op_conv2d(x, w, b, stride=stride, upsample=ups)
    assert got.shape == ref.shape
    assert relerr(got, ref) < TOL[eng.prec]


def test_conv2d_patch_kernel_split_k(eng):
    """LDS-patch conv with the channel chunks split over slices (the 16x16 level of the headline workload), forced onto a
    small shape: fp32 slabs + finalize must match the oracle, epilogue (scale + residual) included, and be reproducible."""
    g = rng(15)
    B, Cin, H, Cout = 8, 512, 16, 168
    x = g.standard_normal((B, Cin, H, H), dtype=np.float32)
    w = (g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(Cin * 9)).astype(np.float32)
    b = g.standard_normal(Cout, dtype=np.float32) * 0.1
    r = g.standard_normal((B, Cout, H, H), dtype=np.float32)
    ref = O.conv2d(x, w, b)
    try:
        eng.set_option("patch_split_tiles", 1)
        got = eng.op_conv2d(x, w, b)
        got_r = eng.op_conv2d(x, w, b, scale=0.5, residual=r, stream_out=True)
        assert np.array_equal(eng.op_conv2d(x, w, b), got)
        eng.set_option("patch_split", 0)
        base = eng.op_conv2d(x, w, b)
    finally:
        eng.set_option("patch_split", 1)
        eng.set_option("patch_split_tiles", 64)
    assert relerr(got, ref) < TOL[eng.prec]
    assert relerr(got_r, ref * 0.5 + r) < TOL[eng.prec]
    assert relerr(base, ref) < TOL[eng.prec] and not np.array_equal(base, got)   # really a different kernel path


@pytest.mark.parametrize("B,Cin,H,Cout,ups", [
    (48, 64, 32, 168, False),    # 192 patches x 2 channel tiles (the second one ragged: 8 of 160), 16-byte epilogue accesses (N % 8 == 0)
    (48, 128, 32, 164, False),   # N % 8 != 0: the 8-byte epilogue form; two channel chunks (the patch double buffer and the ring wrap)
    (50, 64, 16, 96, True),      # fused nearest-x2 upsample: 200 patches, source patch 10 x 10
])
def test_conv2d_patch4_matches_first_generation(eng, B, Cin, H, Cout, ups):
    """The 4-wave patch conv (conv_patch4.hip: one wave per SIMD, 32x32x16 MFMAs, LDS-DMA operands, zero padding by out-of-range
    buffer loads) against the first generation (option patch4 = 0) on launches of >= 192 blocks: image borders in every block row /
    column, ragged channel tile, bias / scale / residual epilogue.  The two MFMA shapes sum the same products in the same order:
    bit-identical in the 2-byte modes (the fp32-storage modes never take the new kernel).  Both against the oracle."""
    g = rng(21)
    x = g.standard_normal((B, Cin, H, H), dtype=np.float32)
    w = (g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(Cin * 9)).astype(np.float32)
    b = g.standard_normal(Cout, dtype=np.float32) * 0.1
    Ho = 2 * H if ups else H
    r = g.standard_normal((B, Cout, Ho, Ho), dtype=np.float32)
    xin = np.repeat(np.repeat(x, 2, axis=2), 2, axis=3) if ups else x
    ref = O.conv2d(xin, w, b)
    try:
        eng.set_option("patch4", 1)
        y1 = eng.op_conv2d(x, w, b, upsample=ups)
        y1r = eng.op_conv2d(x, w, b, upsample=ups, scale=0.75, residual=r, stream_out=True)
        eng.set_option("patch4", 0)
        y0 = eng.op_conv2d(x, w, b, upsample=ups)
        y0r = eng.op_conv2d(x, w, b, upsample=ups, scale=0.75, residual=r, stream_out=True)
    finally:
        eng.set_option("patch4", 1)
    assert relerr(y1, ref) < TOL[eng.prec] and relerr(y1r, ref * 0.75 + r) < TOL[eng.prec]
    assert np.array_equal(y1, y0) and np.array_equal(y1r, y0r)


def test_conv2d_epilogues(eng):
    g = rng(2)
    x = g.standard_normal((2, 64, 8, 8), dtype=np.float32)
    w = (g.standard_normal((64, 64, 3, 3), dtype=np.float32) / 24).astype(np.float32)
    b = g.standard_normal(64, dtype=np.float32) * 0.1
    r = g.standard_normal((2, 64, 8, 8), dtype=np.float32)
    ref = O.conv2d(x, w, b)
    assert relerr(eng.op_conv2d(x, w, b, silu=True), O.silu(ref)) < TOL[eng.prec]
    assert relerr(eng.op_conv2d(x, w, b, scale=0.825, residual=r, stream_out=True), ref * 0.825 + r) < TOL[eng.prec]


def test_splitk_fused_finalize_is_bit_identical(eng):
    """Split-K layers (small M, long K): the slabs are summed in slice order either by the slice that arrives
    last at the tile counter (default) or by splitk_finalize_kernel (option splitk_fused=0).  Both orders are
    fixed, so the results must be bit-identical, run after run (the counters reset themselves)."""
    g = rng(11)
    x = g.standard_normal((3, 320, 16, 16), dtype=np.float32)
    w = (g.standard_normal((320, 320, 3, 3), dtype=np.float32) / 54).astype(np.float32)
    b = g.standard_normal(320, dtype=np.float32) * 0.1
    r = g.standard_normal((3, 320, 16, 16), dtype=np.float32)
    xl = g.standard_normal((1024, 2560), dtype=np.float32)
    wl = (g.standard_normal((1288, 2560), dtype=np.float32) / 50).astype(np.float32)
    try:
        eng.set_option("splitk_fused", 0)
        c0 = eng.op_conv2d(x, w, b, scale=0.5, residual=r, stream_out=True)
        l0 = eng.op_linear(xl, wl, None)
        eng.set_option("splitk_fused", 1)
        for _ in range(3):
            assert np.array_equal(eng.op_conv2d(x, w, b, scale=0.5, residual=r, stream_out=True), c0)
            assert np.array_equal(eng.op_linear(xl, wl, None), l0)
    finally:
        eng.set_option("splitk_fused", 0)     # default: the separate finalize pass (faster here)
    assert relerr(c0, O.conv2d(x, w, b) * 0.5 + r) < TOL[eng.prec]
    assert relerr(l0, O.linear(xl, wl)) < TOL[eng.prec]


@pytest.mark.parametrize("M,K,N", [(300, 64, 256), (7, 1280, 320), (128, 320, 960), (77, 96, 512), (1, 320, 1280)])
def test_linear(eng, M, K, N):
    g = rng(3)
    x = g.standard_normal((M, K), dtype=np.float32)
    w = (g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = g.standard_normal(N, dtype=np.float32) * 0.1
    assert relerr(eng.op_linear(x, w, b), O.linear(x, w, b)) < TOL[eng.prec]
    assert relerr(eng.op_linear(x, w, None), O.linear(x, w)) < TOL[eng.prec]
    # emb_layers: Linear(SiLU(emb)), openaimodel.py:217-223
    assert relerr(eng.op_linear(x, w, b, a_silu=True), O.linear(O.silu(x), w, b)) < TOL[eng.prec]


@pytest.mark.parametrize("tile", [0, 1, 4])   # 128 x 160, 256 x 160, 64 x 80 (the small-M tile)
def test_linear_ring_kernel(eng, tile):
    """gemm_ring.hip (persistent LDS-DMA ring GEMM, 2-byte modes): against the oracle and bit-identical to igemm_kernel (same MFMA
    order per accumulator) on ragged M / N, one and several tiles per block, fewer tiles than XCDs, K from 2 to 40 steps (shapes
    the engine does not split along K: few tiles with K >= 1024 go to the split-K path instead)."""
    if eng.prec not in ("f16", "bf16"):
        pytest.skip("the ring kernel stages 2-byte operands")
    g = rng(21)
    try:
        for M, K, N in [(128, 128, 160), (130, 192, 164), (1000, 320, 320), (777, 960, 1920), (3000, 2560, 1920), (2048 * 9 + 5, 128, 480), (65, 640, 40), (40000, 320, 320)]:
            x = g.standard_normal((M, K), dtype=np.float32)
            w = (g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
            b = g.standard_normal(N, dtype=np.float32) * 0.1
            eng.set_option("ring", 1000); eng.set_option("ring_tile", tile)
            n0 = eng.stat("ring_launches")
            y = eng.op_linear(x, w, b)
            assert eng.stat("ring_launches") == n0 + 1, ("the ring kernel did not take this layer", M, K, N)
            # its ping-pong form (two wave groups half a K step apart; default where it measures faster: K >= 2560, or one 256-row
            # tile per CU) and its lockstep form are the same arithmetic in the same order
            eng.set_option("ring_pp", 0)
            assert np.array_equal(eng.op_linear(x, w, b), y), ("ping-pong vs lockstep", M, K, N)
            eng.set_option("ring_pp", 1)
            n0 += 1
            eng.set_option("ring", 0)
            y0 = eng.op_linear(x, w, b)
            assert eng.stat("ring_launches") == n0 + 1
            assert relerr(y, O.linear(x, w, b)) < TOL[eng.prec], (M, K, N)
            assert np.array_equal(y, y0), (M, K, N)
        # not eligible (K not a multiple of the 64-element stage; SiLU on load): falls back to igemm_kernel, silently and correctly
        eng.set_option("ring", 1000)
        n0 = eng.stat("ring_launches")
        x = g.standard_normal((300, 72), dtype=np.float32); w = g.standard_normal((160, 72), dtype=np.float32)
        assert relerr(eng.op_linear(x, w, None), O.linear(x, w)) < TOL[eng.prec]
        assert eng.stat("ring_launches") == n0
        # GEGLU projections (256 x 160 on 8 x 1 waves): whole 160-column blocks of the interleaved weights
        for M, Cc in [(4096, 320), (1000 + 3, 640)]:
            x = g.standard_normal((M, Cc), dtype=np.float32)
            w = (g.standard_normal((8 * Cc, Cc), dtype=np.float32) / np.sqrt(Cc)).astype(np.float32)
            b = g.standard_normal(8 * Cc, dtype=np.float32) * 0.1
            eng.set_option("ring_geglu", 1)
            n0 = eng.stat("ring_launches")
            y = eng.op_linear(x, w, b, geglu=True)
            assert eng.stat("ring_launches") == n0 + 1
            eng.set_option("ring_geglu", 0)
            y0 = eng.op_linear(x, w, b, geglu=True)
            assert eng.stat("ring_launches") == n0 + 1
            a_, gate = np.split(O.linear(x, w, b), 2, axis=-1)
            assert relerr(y, a_ * O.gelu(gate)) < TOL[eng.prec] and np.array_equal(y, y0), (M, Cc)
    finally:
        eng.set_option("ring", 80); eng.set_option("ring_tile", -1); eng.set_option("ring_geglu", 1); eng.set_option("ring_pp", 1)


@pytest.mark.parametrize("M,C", [(200, 64), (64, 320), (33, 40)])
def test_geglu(eng, M, C):
    g = rng(4)
    x = g.standard_normal((M, C), dtype=np.float32)
    w = (g.standard_normal((8 * C, C), dtype=np.float32) / np.sqrt(C)).astype(np.float32)
    b = g.standard_normal(8 * C, dtype=np.float32) * 0.1
    h = O.linear(x, w, b)
    a, gate = np.split(h, 2, axis=-1)
    assert relerr(eng.op_linear(x, w, b, geglu=True), a * O.gelu(gate)) < TOL[eng.prec]


def test_headline_gemm_tiles_against_oracle(eng):
    """The big-tile instantiations only chosen at the headline workload's sizes: GEGLU on the 256 x 320 tile (M x N large
    enough for >= 256 such tiles) and a K = 2560 projection on the 256 x 160 tile."""
    g = rng(16)
    M, K, N = 14336, 320, 2560
    x = g.standard_normal((M, K), dtype=np.float32)
    w = (g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = g.standard_normal(N, dtype=np.float32) * 0.1
    a, gate = np.split(O.linear(x, w, b), 2, axis=-1)
    assert relerr(eng.op_linear(x, w, b, geglu=True), a * O.gelu(gate)) < TOL[eng.prec]
    M, K, N = 16384, 2560, 640
    x = g.standard_normal((M, K), dtype=np.float32)
    w = (g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    assert relerr(eng.op_linear(x, w, None), O.linear(x, w)) < TOL[eng.prec]
    # 256 x 192 tile (2-byte modes): widths that divide by 192 but not by 160 -- the MMDiT hidden size 1536; ragged M
    M, K, N = 6144 + 37, 448, 1536
    x = g.standard_normal((M, K), dtype=np.float32)
    w = (g.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
    b = g.standard_normal(N, dtype=np.float32) * 0.1
    assert relerr(eng.op_linear(x, w, b), O.linear(x, w, b)) < TOL[eng.prec]


@pytest.mark.parametrize("B,C,H,W,eps,silu", [(2, 64, 8, 8, 1e-5, True), (3, 320, 16, 16, 1e-6, False),
                                               (1, 960, 4, 4, 1e-5, True), (2, 128, 5, 3, 1e-5, True),
                                               (1, 2560, 8, 8, 1e-5, True)])
def test_groupnorm(eng, B, C, H, W, eps, silu):
    g = rng(5)
    x = (g.standard_normal((B, C, H, W), dtype=np.float32) * 2 + 0.5).astype(np.float32)
    ga = 1 + 0.1 * g.standard_normal(C, dtype=np.float32)
    be = 0.1 * g.standard_normal(C, dtype=np.float32)
    x = round_like(x, eng.prec)
    ref = O.group_norm(x, ga, be, eps=eps)
    ref = O.silu(ref) if silu else ref
    assert relerr(eng.op_groupnorm(x, ga, be, eps, silu), ref) < NORM_TOL[eng.prec]


@pytest.mark.parametrize("rows,C", [(100, 64), (9, 320), (5, 1280), (257, 640)])
def test_layernorm(eng, rows, C):
    g = rng(6)
    x = (g.standard_normal((rows, C), dtype=np.float32) * 3 - 1).astype(np.float32)
    ga = 1 + 0.1 * g.standard_normal(C, dtype=np.float32)
    be = 0.1 * g.standard_normal(C, dtype=np.float32)
    x = round_like(x, eng.prec)
    assert relerr(eng.op_layernorm(x, ga, be), O.layer_norm(x, ga, be)) < 1e-5


def _attn_ref(q, k, v, heads):
    B, Nq, C = q.shape
    dh = C // heads
    sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, dh).transpose(0, 2, 1, 3).astype(np.float64)
    qh, kh, vh = sp(q), sp(k), sp(v)
    s = qh @ kh.transpose(0, 1, 3, 2) * dh ** -0.5
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p @ vh).transpose(0, 2, 1, 3).reshape(B, Nq, C).astype(np.float32)


@pytest.mark.parametrize("B,Nq,Nk,C", [
    (2, 256, 256, 64),    # dh 8, self
    (1, 64, 64, 256),     # dh 32
    (2, 100, 77, 320),    # dh 40, cross (ragged key tile, ragged query block)
    (1, 1024, 1024, 320), # dh 40, self, many key tiles
    (1, 192, 192, 640),   # dh 80
    (2, 48, 77, 1280),    # dh 160, cross
    (2, 256, 256, 1280),  # dh 160, self, four key tiles (the 16x16 level: attn2_kernel<160> in the 2-byte modes)
    (1, 12, 12, 128),     # dh 16, tiny ragged
])
def test_attention(eng, B, Nq, Nk, C):
    g = rng(7)
    q = g.standard_normal((B, Nq, C), dtype=np.float32)
    k = g.standard_normal((B, Nk, C), dtype=np.float32)
    v = g.standard_normal((B, Nk, C), dtype=np.float32)
    ref = _attn_ref(q, k, v, eng.cfg.num_heads)
    got = eng.op_attention(q, k, v)
    assert relerr(got, ref) < ATTN_TOL[eng.prec]


def test_attention_peaked_softmax(eng):
    """One key dominates each row at a chosen tile: exercises the online-softmax rescale branch."""
    g = rng(8)
    B, N, C = 1, 320, 64
    q = g.standard_normal((B, N, C), dtype=np.float32)
    k = g.standard_normal((B, N, C), dtype=np.float32)
    v = g.standard_normal((B, N, C), dtype=np.float32)
    k[0, 200] = q[0, 5] * 6.0   # spike for query 5 in the 4th key tile
    k[0, 70] = q[0, 300] * 6.0  # spike in the 2nd key tile
    ref = _attn_ref(q, k, v, eng.cfg.num_heads)
    assert relerr(eng.op_attention(q, k, v), ref) < 1.5 * ATTN_TOL[eng.prec]
