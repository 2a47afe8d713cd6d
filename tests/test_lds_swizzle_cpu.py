"""Bank-conflict model of the contraction kernels' LDS images (CPU, no GPU): ds_read_b128 is served in four fixed groups of 16 lanes
(MI355X_MICROARCH.md, LDS table); a group takes one LDS cycle when its 16 x 16 B hit 64 distinct banks.  The fragment reads of
gemm.hip / gemm_ring.hip start at multiples of 16 rows, those of conv_patch.hip at (py + ky) * PW + kx -- any row."""
import itertools

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def cycles(addr_of_lane):
    """LDS cycles of one wave-wide ds_read_b128 (4 = conflict-free)."""
    total = 0
    for g in GROUPS:
        banks = {}
        for lane in g:
            a = addr_of_lane(lane)
            assert a % 16 == 0
            for d in range(4):
                banks.setdefault((a // 4 + d) % 64, set()).add(a + 4 * d)
        total += max(len(v) for v in banks.values())
    return total


def swz_gemm(row, chunk):      # gemm.hip / gemm_ring.hip (and conv_patch2.hip): chunk ^ ((row >> 1) & 7)
    return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4)


def swz_patch(row, chunk):     # conv_patch.hip: bits 1-2 of the chunk ^ ((row >> 1) & 3), bit 0 untouched
    return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4)


def frag(swz, start_row, ks):  # lane (fr = lane & 15, fq = lane >> 4) reads chunk ks*4 + fq of row start_row + fr
    return lambda lane: swz(start_row + (lane & 15), ks * 4 + (lane >> 4))


def test_gemm_swizzle_is_conflict_free_on_16_row_boundaries():
    for start, ks in itertools.product(range(0, 512, 16), range(2)):
        assert cycles(frag(swz_gemm, start, ks)) == 4
    # ... and only there: a start row of 2 mod 4 collides (this is why the patch image has its own swizzle)
    assert max(cycles(frag(swz_gemm, s, 0)) for s in range(2, 64, 4)) == 8


def test_patch_swizzle_is_conflict_free_for_any_start_row():
    for start, ks in itertools.product(range(0, 400), range(2)):
        assert cycles(frag(swz_patch, start, ks)) == 4


def test_patch_reads_of_the_nine_taps():
    """the 18 x 18 patch image of conv_patch.hip: tap (ky, kx) of pixel row py reads rows (py + ky) * 18 + kx .. + 15"""
    for ky, kx, py, ks in itertools.product(range(3), range(3), range(16), range(2)):
        assert cycles(frag(swz_patch, (py + ky) * 18 + kx, ks)) == 4
    old = [cycles(frag(swz_gemm, (py + ky) * 18 + kx, ks)) for ky, kx, py, ks in itertools.product(range(3), range(3), range(16), range(2))]
    assert 6.5 < sum(old) / len(old) < 6.8     # what the round-2 swizzle cost there: 6.7 cycles per read on average


def test_row_writes_are_conflict_free_under_both():
    """staging writes (ds_write_b128 / LDS-DMA): 8 lanes write the 8 chunks of one row -- a permutation inside 128 contiguous bytes"""
    for swz in (swz_gemm, swz_patch):
        for row in range(64):
            assert sorted(swz(row, c) for c in range(8)) == [row * 128 + 16 * c for c in range(8)]
