"""How a grouped convolution launch is cut over the 8 XCDs (conv_group_finalize behind gtx_op_conv_xcd_ranges; host
arithmetic only, runs on CPU). The kernels give XCD x the logical blocks [xcd_begin[x], xcd_begin[x + 1]); with equal COUNTS the
detection head's first stage left XCD 7 with 2.8 x the mean work, so the ranges hold equal WORK."""
import numpy as np
import pytest


def _work(xcd_begin, blocks, cin):
    per_block = np.repeat(np.asarray(cin, np.int64), blocks)
    return np.array([per_block[xcd_begin[k]:xcd_begin[k + 1]].sum() for k in range(8)])


@pytest.mark.parametrize("blocks,cin", [
    ([2700, 675, 171], [128, 256, 512]),                          # YOLOv8s head stage 1 at 1920x1920, batch 2
    ([900, 1800, 225, 450, 57, 114], [64, 128, 64, 128, 64, 128]),  # head stage 2
    ([171, 675, 2700], [512, 256, 128]),                          # the same members the other way round
    ([57, 57, 57], [512, 256, 128]),
])
def test_grouped_launch_ranges_partition_the_blocks_and_balance_the_work(blocks, cin):
    from geotrax_amd import ops

    xb, grid = ops.conv_xcd_ranges(blocks, cin)
    total = int(np.sum(blocks))
    assert xb[0] == 0 and xb[8] == total and np.all(np.diff(xb) >= 0)
    assert grid == 8 * int(np.diff(xb).max()) >= total
    w = _work(xb, blocks, cin)
    assert w.sum() == int(np.dot(blocks, cin))
    # no XCD carries more than the mean plus one block of the deepest member
    assert w.max() <= w.mean() + max(cin)
    # equal counts, for comparison: what the kernels did before
    q, r = divmod(total, 8)
    eq = np.concatenate([[0], np.cumsum([q + (1 if k < r else 0) for k in range(8)])])
    if len(set(cin)) > 1 and min(blocks) > 8:
        assert _work(eq, blocks, cin).max() > w.max()


@pytest.mark.parametrize("n", [1, 5, 8, 9, 1003, 4096])
def test_single_member_ranges_are_equal_counts(n):
    from geotrax_amd import ops

    xb, grid = ops.conv_xcd_ranges([n], [64])
    d = np.diff(xb)
    assert xb[8] == n and d.max() - d.min() <= 1 and grid == 8 * d.max()


def test_bad_arguments_are_refused():
    from geotrax_amd import ops
    from geotrax_amd._lib import GtxError

    with pytest.raises(GtxError):
        ops.conv_xcd_ranges([0], [64])
    with pytest.raises(GtxError):
        ops.conv_xcd_ranges([1] * 9, [64] * 9)
