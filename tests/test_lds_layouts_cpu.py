"""Bank-conflict model of the three LDS layouts changed at the end of round 3 (DESIGN.md "Round-3 findings": LDS bank conflicts).

The kernels' address formulas are restated here and run through the gfx950 banking rules (64 banks of 4 bytes; the lane groups that
share one LDS cycle per instruction: two 32-lane halves for ds_read_b64_tr_b16, four fixed 16-lane groups for ds_read_b128).  The
counters on the GPU (tools/gpu/lds_conflicts.sh, profiles/r03_lds_conflicts.txt) are the measurement; this is the arithmetic behind
them, kept as a test so that a change of a swizzle key or a row pitch has to stay conflict-free on paper as well."""
import itertools

B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]
HALVES = [list(range(32)), list(range(32, 64))]


def worst_way(addr_of_lane, groups, nbytes):
    """largest number of DISTINCT addresses that meet on one bank inside one lane group (1 = conflict-free)"""
    worst = 1
    for g in groups:
        per_bank = {}
        for lane in g:
            a = addr_of_lane(lane)
            for b in range(a // 4, (a + nbytes) // 4):
                per_bank.setdefault(b % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


# ---- wgrad_v3.hip: row buffers [pixel][C channels], ds_read_b64_tr_b16; quarter q of slot pixel P stored at q ^ key(P)
def wg3_key(C, P):
    return (P & 3) if C == 128 else ((P >> 1) & 1)


def wg3_addr(C, lane, kw, ct, swizzled):
    pxb = 2 * C
    g4, tq, tp = lane >> 4, (lane & 15) >> 2, lane & 3
    hh, blk = g4 >> 1, g4 & 1
    quarter = ct ^ (wg3_key(C, kw + tq) if swizzled else 0)
    return (kw + 8 * hh + tq) * pxb + quarter * 64 + blk * 32 + tp * 8


def test_wg3_row_buffers_plain_layout_conflicts_as_measured():
    # 128 channels: 4-way (the four pixels of a half sit 256 bytes apart); 64 channels: 2-way -- what the counter showed (0.69)
    assert worst_way(lambda l: wg3_addr(128, l, 0, 0, False), HALVES, 8) == 4
    assert worst_way(lambda l: wg3_addr(64, l, 0, 0, False), HALVES, 8) == 2


def test_wg3_row_buffers_swizzled_conflict_free_for_every_tap_and_tile():
    for C in (64, 128):
        for kw, ct, hi in itertools.product(range(3), range(C // 32), (0, 4)):
            # (the second read of a fragment addresses pixel + 4: the same key, since 4 keeps P & 3 and bit 1 of P)
            assert worst_way(lambda l: wg3_addr(C, l, kw + hi, ct, True), HALVES, 8) == 1, (C, kw, ct, hi)


def test_wg3_swizzle_is_a_permutation_of_each_pixel():
    for C in (64, 128):
        nq = 2 * C // 64
        for P in range(16):
            assert sorted(q ^ wg3_key(C, P) for q in range(nq)) == list(range(nq))


# ---- conv_v5.hip band kernel: rows of 64 bytes, ds_read_b128, chunk c of row r in slot c ^ key(r); read at row r0 + fr + shift
def band_addr(lane, shift, key):
    fr, fq = lane & 15, lane >> 4
    row = 16 + fr + shift
    return row * 64 + ((fq ^ key(row)) * 16)


def test_band_swizzle_keyed_on_row_bit_2_is_conflict_free_at_every_alignment():
    new = lambda r: 2 if r & 4 else 0
    old = lambda r: 3 if r & 8 else 0
    for shift in range(16):
        assert worst_way(lambda l: band_addr(l, shift, new), B128_GROUPS, 16) == 1, shift
    assert worst_way(lambda l: band_addr(l, 0, old), B128_GROUPS, 16) == 1            # the old key: fine when 16-aligned ...
    assert worst_way(lambda l: band_addr(l, 1, old), B128_GROUPS, 16) == 2            # ... 2-way on the shifted column taps
    assert worst_way(lambda l: band_addr(l, 2, old), B128_GROUPS, 16) == 2


# ---- conv_px.hip: weight rows of 2 K bytes + padding, ds_read_b128 at col * PITCH + 16 g
def px_addr(lane, K, pad):
    col, g = lane & 15, lane >> 4
    return col * (2 * K + pad) + 16 * g


def test_px_weight_rows_two_slots_of_padding():
    for K in (64, 128, 256, 320):
        assert worst_way(lambda l: px_addr(l, K, 32), B128_GROUPS, 16) == 1, K
        assert worst_way(lambda l: px_addr(l, K, 16), B128_GROUPS, 16) == 2, K       # one slot: a pair per group on the same slot (0.50)
