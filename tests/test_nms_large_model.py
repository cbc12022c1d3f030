"""CPU models of the two algorithms behind csrc/nms_large.h (BoxOutput / NMS / final stage beyond 4032 boxes), checked against
the plain definitions -- the same role tests/test_x3_model.py plays for the split-fp16 arithmetic.  The device code itself is
checked bit-for-bit against the oracle in tests/test_gpu_ops.py (-m gpu); these run everywhere and pin the index arithmetic and
the tiling argument: (1) the global bitonic network addressed by GLOBAL index (LDS kernel for strides <= 1024 + one
compare-exchange kernel per larger stride) sorts descending with the zero padding last; (2) greedy NMS cut into tiles -- cross
suppression by the kept boxes of earlier tiles, then the in-tile greedy scan seeded with that bitmap -- keeps exactly the boxes
the sequential definition (box_output_layer.cpp:38-63) keeps, for every tile size, IoU mode and tie pattern."""
import numpy as np
import pytest

LOCAL = 2048       # kBigSortLocal


def _local(keys, size_lo, size_hi):
    for blk in range(len(keys) // LOCAL):
        gbase = blk * LOCAL
        sk = keys[gbase:gbase + LOCAL].copy()
        size = size_lo
        while size <= size_hi:
            stride = min(size >> 1, LOCAL >> 1)
            while stride > 0:
                tid = np.arange(LOCAL // 2)
                lo = (tid // stride) * (stride << 1) + (tid % stride)
                hi = lo + stride
                desc = ((gbase + lo) & size) == 0
                a, b = sk[lo], sk[hi]
                swap = np.where(desc, a < b, a > b)
                sk[lo], sk[hi] = np.where(swap, b, a), np.where(swap, a, b)
                stride >>= 1
            size <<= 1
        keys[gbase:gbase + LOCAL] = sk


def _step(keys, size, stride):
    t = np.arange(len(keys) // 2)
    lo = (t // stride) * (stride << 1) + (t % stride)
    hi = lo + stride
    desc = (lo & size) == 0
    a, b = keys[lo], keys[hi]
    swap = np.where(desc, a < b, a > b)
    keys[lo], keys[hi] = np.where(swap, b, a), np.where(swap, a, b)


def big_sort_desc(keys):
    """big_sort_desc of nms_large.h, launch for launch."""
    P = len(keys)
    _local(keys, 2, LOCAL)
    size = 2 * LOCAL
    while size <= P:
        stride = size >> 1
        while stride >= LOCAL:
            _step(keys, size, stride)
            stride >>= 1
        _local(keys, size, size)
        size <<= 1


@pytest.mark.parametrize("P,n", [(2048, 1), (2048, 2048), (4096, 2049), (8192, 5000), (32768, 20000), (131072, 81600)])
def test_global_bitonic_network_sorts_descending_with_padding_last(P, n):
    rng = np.random.default_rng(P + n)
    keys = np.zeros(P, np.uint64)
    keys[:n] = rng.permutation(np.arange(1, 4 * n + 1, dtype=np.uint64))[:n] << np.uint64(20)     # unique, non-zero
    want = np.sort(keys)[::-1]
    big_sort_desc(keys)
    assert np.array_equal(keys, want)
    assert np.all(keys[n:] == 0)


def _over(a, b, thr, mode):
    """BoxIOU > thr with a the earlier box (math_functions.cpp:12-35)."""
    if a[2] <= 0 or a[3] <= 0 or b[2] <= 0 or b[3] <= 0:
        return False
    tlx, tly = max(a[0], b[0]), max(a[1], b[1])
    brx, bry = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
    over = 0.0 if (tlx >= brx or tly >= bry) else (brx - tlx) * (bry - tly)
    u = min(a[2] * a[3], b[2] * b[3]) if mode == 1 else a[2] * a[3] if mode == 2 else a[2] * a[3] + b[2] * b[3] - over
    return over / u > thr


def greedy(boxes, thr, mode):
    n = len(boxes)
    keep = np.ones(n, bool)
    for i in range(n):
        if keep[i]:
            for j in range(i + 1, n):
                if keep[j] and _over(boxes[i], boxes[j], thr, mode):
                    keep[j] = False
    return keep


def tiled(boxes, thr, mode, tile):
    """big_nms_tiles: per tile (1) removed_init from the kept boxes of earlier tiles, (2) the tile's own upper-triangular
    matrix, (3) the scan seeded with removed_init; kept boxes appended in order."""
    n = len(boxes)
    kept = []
    for base in range(0, n, tile):
        idx = range(base, min(n, base + tile))
        removed = {j: any(_over(boxes[k], boxes[j], thr, mode) for k in kept) for j in idx}       # cross kernel
        for i in idx:                                                                             # greedy_scan
            if removed[i]:
                continue
            kept.append(i)
            for j in idx:
                if j > i and _over(boxes[i], boxes[j], thr, mode):
                    removed[j] = True
    keep = np.zeros(n, bool)
    keep[kept] = True
    return keep, kept


@pytest.mark.parametrize("tile", [1, 7, 64, 100])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_tiled_greedy_nms_equals_sequential_greedy(tile, mode):
    rng = np.random.default_rng(10 * tile + mode)
    n = 260
    centers = rng.uniform(0, 300, (12, 2))
    c = centers[rng.integers(0, 12, n)] + rng.normal(0, 10, (n, 2))
    wh = rng.uniform(15, 90, (n, 2))
    boxes = np.concatenate([c, wh], 1)
    boxes[5, 2] = 0.0                         # degenerate: IoU 0 with everything
    boxes[40:44] = boxes[40]                  # identical boxes: only the first survives
    want = greedy(boxes, 0.5, mode)
    got, order = tiled(boxes, 0.5, mode, tile)
    assert np.array_equal(got, want)
    assert order == sorted(order)             # the kept list is in sorted (score) order: what the emit kernels rely on
    assert 20 < want.sum() < n - 20
