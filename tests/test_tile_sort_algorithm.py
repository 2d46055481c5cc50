"""CPU restatement of the second sort stage (csrc/lg_binning.h: lg_tile_sort_wave / _wg / lg_tile_sort_long), step
by step as the kernels take them -- stable counting passes of ceil(D / P)-bit digits over the depth field with ranks from the
per-(item, digit) lane sets, the lowest digit finished by run insertion with the all-digits fallback, tile ranges from the
ends of each tile's stretch inside a radix-sort tile -- against numpy's sort of the whole keys.  No GPU involved: this pins
the ALGORITHM (stability argument, fallback condition, padding); tests/test_gpu_sort.py pins the kernels."""
import numpy as np
import pytest

RUN = 8          # LG_TS_RUN
DIGIT = 9        # LG_TS_DIGIT


def _width(depth_bits):
    passes = (depth_bits + DIGIT - 1) // DIGIT
    return (depth_bits + passes - 1) // passes


def _counting_pass(keys, shift, nbits, gid_bits, lanes=64):
    """One stable pass as the kernels do it: list order = (item, lane); within an item the lanes of equal digit form a set, the
    rank of a key = keys of its digit in earlier items (the digit's counter) + members of its set at lower lanes."""
    n = len(keys)
    d = ((keys >> np.uint64(gid_bits + shift)) & np.uint64((1 << nbits) - 1)).astype(np.int64)
    cnt = np.zeros(1 << nbits, np.int64)
    rank = np.zeros(n, np.int64)
    for i0 in range(0, n, lanes):
        di = d[i0:i0 + lanes]
        for lane, dig in enumerate(di):
            peers = np.nonzero(di == dig)[0]                       # the lane set read back from the mask word
            rank[i0 + lane] = cnt[dig] + int((peers < lane).sum())
        for dig in np.unique(di):                                   # the lowest lane of each set bumps the counter
            cnt[dig] += int((di == dig).sum())
    base = np.concatenate(([0], np.cumsum(cnt)[:-1]))
    out = np.empty_like(keys)
    out[base[d] + rank] = keys
    return out


def _tile_sort(keys, depth_bits, gid_bits, finish=True):
    """keys of ONE tile in id order (what the stable tile-bit passes leave) -> (depth, id) order."""
    w = _width(depth_bits)
    keys = keys.copy()
    for attempt in range(2):
        first = w if (finish and depth_bits > w and attempt == 0) else 0
        for shift in range(first, depth_bits, w):
            keys = _counting_pass(keys, shift, min(w, depth_bits - shift), gid_bits)
        if first == 0:
            return keys, attempt
        up = np.uint64(gid_bits + w)
        top = keys >> up
        too_long = False
        i = 0
        while i < len(keys):
            e = i + 1
            while e < len(keys) and top[e] == top[i]:
                e += 1
            if e - i > RUN:
                too_long = True                                    # (the kernel leaves such a run alone and falls back)
            elif e - i > 1:
                keys[i:e] = np.sort(keys[i:e])                      # insertion sort on the whole key
            i = e
        if not too_long:
            return keys, attempt
    raise AssertionError("unreachable")


def _make(n, depth_bits, gid_bits, rs, mode):
    ids = np.sort(rs.choice(1 << gid_bits, n, replace=False)).astype(np.uint64)      # id order = emission order
    if mode == "random":
        depth = rs.randint(0, 1 << depth_bits, n, dtype=np.int64)
    elif mode == "slab":                                            # coplanar: one depth (+ a few within the lowest digit)
        depth = np.full(n, 12345 << 9, np.int64)
        depth[::7] += rs.randint(0, 1 << 9, len(depth[::7]))
    elif mode == "pairs":                                           # many short runs of equal upper bits
        depth = (rs.randint(0, 1 << (depth_bits - 9), n, dtype=np.int64) // 3 * 3 << 9) + rs.randint(0, 1 << 9, n)
    else:                                                           # clustered: long runs of equal upper bits, distinct low bits
        depth = (rs.randint(0, 4, n, dtype=np.int64) << 9) + rs.randint(0, 1 << 9, n)
    return (depth.astype(np.uint64) << np.uint64(gid_bits)) | ids


@pytest.mark.parametrize("mode", ["random", "slab", "pairs", "clustered"])
@pytest.mark.parametrize("n,depth_bits", [(2, 27), (63, 27), (64, 26), (507, 27), (1024, 27), (700, 18), (300, 9), (450, 32)])
def test_tile_list_ends_in_depth_then_id_order(mode, n, depth_bits):
    rs = np.random.RandomState(n * 31 + depth_bits)
    gid_bits = 21
    if depth_bits < 20 and mode != "random":
        pytest.skip("pattern needs a wide depth field")
    keys = _make(n, depth_bits, gid_bits, rs, mode)
    want = np.sort(keys)                                            # unique keys: the u64 order IS (depth, id)
    got, attempt = _tile_sort(keys, depth_bits, gid_bits)
    assert np.array_equal(got, want)
    all_digits, _ = _tile_sort(keys, depth_bits, gid_bits, finish=False)
    assert np.array_equal(all_digits, want)
    if mode in ("slab", "clustered") and n > 64:
        assert attempt == 1                                         # runs beyond LG_TS_RUN: the all-digits fallback ran
    if mode == "random" and depth_bits >= 26 and n <= 1024:
        assert attempt == 0


def test_equal_depths_keep_id_order():
    """A stable sort on depth alone must leave ties in id order (the radix passes before it are stable and lg_duplicate emits in id
    order): here EVERY key has the same depth."""
    ids = np.arange(5, 905, 3, dtype=np.uint64)
    keys = (np.uint64(777) << np.uint64(21)) | ids
    got, _ = _tile_sort(keys, 27, 21)
    assert np.array_equal(got, keys)


def test_ranges_from_the_stretches_of_a_tile_inside_each_sort_tile():
    """lg_onesweep_pass, last pass: keys of one tile id are contiguous in a sort tile's output order; atomicMin / atomicMax of the
    ends of every stretch over {0xFFFFFFFF, 0} give {begin, end}; tiles without keys keep the initial pair (lg_tile_sort -> {0, 0})."""
    rs = np.random.RandomState(5)
    ntiles, n, sort_tile = 40, 5000, 512
    t = np.sort(rs.choice([x for x in range(ntiles) if x not in (0, 7, 39)], n))     # sorted tile ids; three tiles stay empty
    lo = np.full(ntiles, 0xFFFFFFFF, np.int64); hi = np.zeros(ntiles, np.int64)
    for base in range(0, n, sort_tile):                                              # one workgroup per sort tile
        chunk = t[base:base + sort_tile]
        for q, tile in enumerate(chunk):
            if q == 0 or chunk[q - 1] != tile:
                lo[tile] = min(lo[tile], base + q)
            if q + 1 == len(chunk) or chunk[q + 1] != tile:
                hi[tile] = max(hi[tile], base + q + 1)
    for tile in range(ntiles):
        idx = np.nonzero(t == tile)[0]
        if len(idx) == 0:
            assert lo[tile] == 0xFFFFFFFF and hi[tile] == 0
        else:
            assert (lo[tile], hi[tile]) == (idx[0], idx[-1] + 1)
