"""Step.PARTIAL flush (maxPartialMemory) and the reference's intermediate-state format on the device.

  * HashAggregationOperator's partial flush state machine (…/operator/HashAggregationOperator.java:366-378, 476-513) with the
    reference's own case TestHashAggregationOperator.testMultiplePartialFlushes (…/TestHashAggregationOperator.java:511-592);
  * PA_STATES_REFERENCE: one block per aggregate typed as the reference's AccumulatorStateSerializers type it
    (StateCompiler.java:127-185), against the oracle's restatement (oracle.states_to_reference / states_from_reference),
    PARTIAL -> FINAL = SINGLE across both formats and both directions (device PARTIAL -> oracle FINAL, oracle PARTIAL -> device FINAL)."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import AggregationOperator, HashAggregationOperator, to_pages, upload_page
from presto_amd.page import Block, Page, sequence_page
from tests.util import rows_equal_ignore_order

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hash_enabled", [False, True])
def test_multiple_partial_flushes(gpu, oracle, hash_enabled):
    """testMultiplePartialFlushes: 4 sequence pages of 500 BIGINT rows, Step.PARTIAL min(bigint), maxPartialMemory 1 kB: the
    operator stops taking input once it is full, a drain yields pages, then it needs input again; all partial pages together are
    (i, i) for i in [0, 2000) -- as rows (key, min state), the reference's serialized NullableLongState being a plain BIGINT."""
    pages = [sequence_page(500, [(abi.BIGINT, 500 * k)]) for k in range(4)]
    types = [abi.BIGINT]
    hash_channel = -1
    if hash_enabled:
        pages = [Page(p.blocks + [Block.bigint(oracle.hash_page(p, [0]))], 500) for p in pages]
        types, hash_channel = [abi.BIGINT, abi.BIGINT], 1
    op = HashAggregationOperator(types, [0], [(abi.AGG_MIN, 0, abi.BIGINT)], hash_channel=hash_channel, expected_groups=100_000,
                                 step=abi.STEP_PARTIAL, max_partial_memory=1024, state_format=abi.STATES_REFERENCE)
    it = iter(pages)
    fed = 0
    while op.needsInput():          # "Fill up the aggregation"
        page = next(it, None)
        if page is None:
            break
        op.addInput(page)
        fed += 1
    assert 0 < fed < 4              # full before the input ran out
    out = []
    while True:                     # "Drain the output (partial flush)"
        page = op.getOutput()
        if page is None:
            break
        out.append(page)
    assert out                      # "There should be some pages that were drained"
    assert op.needsInput()          # "The operator need input again since this was a partial flush"
    out += to_pages(op, list(it))
    rows = [r for p in out for r in p.to_rows()]
    if hash_enabled:
        assert all(r[1] == oracle.hash_page(Page([Block.bigint([r[0]])], 1), [0])[0] for r in rows[:50])
        rows = [(r[0], r[2]) for r in rows]
    assert sorted(rows) == [(i, i) for i in range(2000)]
    op.close()


def mixed_page(rng, n, keys):
    return Page([
        Block.bigint(rng.integers(0, keys, n)),
        Block.double(rng.random(n), rng.random(n) < 0.1),
        Block.bigint(rng.integers(-1000, 1000, n), rng.random(n) < 0.05),
        Block.integer(rng.integers(-50, 50, n), rng.random(n) < 0.3),
        Block.date(rng.integers(8000, 11000, n)),
        Block.boolean(rng.random(n) < 0.5, rng.random(n) < 0.5),
    ], n)


TYPES = [abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.INTEGER, abi.DATE, abi.BOOLEAN]
AGGS = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_COUNT, 1, abi.DOUBLE), (abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.BIGINT),
        (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_AVG, 2, abi.BIGINT), (abi.AGG_MIN, 2, abi.BIGINT), (abi.AGG_MAX, 3, abi.INTEGER),
        (abi.AGG_MIN, 4, abi.DATE), (abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_MIN, 5, abi.BOOLEAN)]
# the serialized type of every state (what a Java FINAL operator reads)
STATE_TYPES = [abi.BIGINT, abi.BIGINT, abi.ROW, abi.ROW, abi.ROW, abi.ROW, abi.BIGINT, abi.BIGINT, abi.BIGINT, abi.DOUBLE, abi.BOOLEAN]


def final_aggs(leading):
    """Step.FINAL aggregates over reference-format states: aggregate k reads channel leading + k; input type = the value's type."""
    out = []
    for k, a in enumerate(AGGS):
        t = a[2]
        if a[0] == abi.AGG_AVG or (a[0] == abi.AGG_SUM and t == abi.DOUBLE):
            t = abi.DOUBLE
        elif a[0] == abi.AGG_SUM:
            t = abi.BIGINT
        elif a[0] in (abi.AGG_COUNT, abi.AGG_COUNT_STAR):
            t = abi.BIGINT
        out.append((a[0], leading + k, t))
    return out


@pytest.mark.parametrize("device_pages", [False, True])
def test_reference_state_format_both_directions(gpu, oracle, device_pages):
    rng = np.random.default_rng(33)
    halves = [mixed_page(rng, 40_011, 700), mixed_page(rng, 25_003, 900)]
    single = oracle.HashAggregation(TYPES, [0], AGGS)
    for h in halves:
        single.add_page(h)
    expected = single.build_result().to_rows()
    # device PARTIAL in the reference's format == the oracle's PARTIAL reshaped
    device_partials, oracle_partials = [], []
    for h in halves:
        op = HashAggregationOperator(TYPES, [0], AGGS, step=abi.STEP_PARTIAL, state_format=abi.STATES_REFERENCE,
                                     output_mem=abi.MEM_DEVICE if device_pages else abi.MEM_HOST)
        out = to_pages(op, [upload_page(h) if device_pages else h])
        if device_pages:
            from presto_amd.operators import download_page
            out = [download_page(p) for p in out]
        assert [b.type for b in out[0].blocks] == [abi.BIGINT] + STATE_TYPES
        ref = oracle.HashAggregation(TYPES, [0], AGGS, step=abi.STEP_PARTIAL)
        ref.add_page(h)
        want = oracle.states_to_reference(ref.build_result(), 1, AGGS)
        rows_equal_ignore_order([r for p in out for r in p.to_rows()], want.to_rows(), rel=1e-12)
        assert all(r[3][1] is True and r[3][3] is True for r in out[0].to_rows()[:20])   # firstNull / secondNull: always true
        device_partials += out
        oracle_partials.append(want)
        op.close()
    faggs = final_aggs(1)
    # device PARTIAL -> oracle FINAL, oracle PARTIAL -> device FINAL, device -> device: all equal SINGLE
    from presto_amd.exchange import partial_layout
    ptypes, flat_final = partial_layout([abi.BIGINT], AGGS)
    ofinal = oracle.HashAggregation(ptypes, [0], flat_final, step=abi.STEP_FINAL)
    for p in device_partials:
        ofinal.add_page(oracle.states_from_reference(p, 1, AGGS))
    rows_equal_ignore_order(ofinal.build_result().to_rows(), expected, rel=1e-12)
    for name, partials in (("oracle partials", oracle_partials), ("device partials", device_partials)):
        op = HashAggregationOperator([abi.BIGINT] + STATE_TYPES, [0], faggs, step=abi.STEP_FINAL, state_format=abi.STATES_REFERENCE)
        rows = [r for p in to_pages(op, [upload_page(p) if device_pages else p for p in partials]) for r in p.to_rows()]
        rows_equal_ignore_order(rows, expected, rel=1e-12)
        op.close()


def test_global_aggregation_states_in_reference_format(gpu, oracle):
    rng = np.random.default_rng(5)
    page = mixed_page(rng, 30_001, 10)
    aggs = AGGS
    (expected,) = [r for p in to_pages(AggregationOperator(TYPES, aggs), [page]) for r in p.to_rows()]
    parts = []
    for region in (page.get_region(0, 10_000), page.get_region(10_000, 20_001)):
        parts += to_pages(AggregationOperator(TYPES, aggs, step=abi.STEP_PARTIAL, state_format=abi.STATES_REFERENCE), [region])
    assert [b.type for b in parts[0].blocks] == STATE_TYPES and all(p.position_count == 1 for p in parts)
    (final,) = [r for p in to_pages(AggregationOperator(STATE_TYPES, final_aggs(0), step=abi.STEP_FINAL, state_format=abi.STATES_REFERENCE), parts)
                for r in p.to_rows()]
    rows_equal_ignore_order([final], [expected], rel=1e-12)


def test_operator_blocks_on_the_hbm_budget_and_resumes(gpu, oracle, monkeypatch):
    """Operator.isBlocked on memory (Operator.java:69-80; HashAggregationOperator's unfinishedWork, :435-438): with an HBM budget
    set (pa_memory_set_limit) a second aggregation that cannot get its table puts its (stable) page aside, refuses input and
    reports blocked; closing the first operator unblocks it, and its result is unaffected."""
    import ctypes as C
    from presto_amd._lib import check, lib
    from tests.test_gpu_small_pages import stable_regions
    monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", "1")   # no gathering: a page is launched (and its table allocated) as it arrives
    monkeypatch.setenv("PRESTO_AMD_NO_PARTITIONED", "1")  # the HBM-table tier alone: one table, sized once by the planner's estimate
    rng = np.random.default_rng(2)
    n = 2_300_003                                        # above the small-page bound, so it is not copied into an arena either
    host = Page([Block.bigint(rng.integers(0, 300_000, n)), Block.bigint(rng.integers(-5, 5, n))], n)
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None)]
    ref = oracle.HashAggregation([abi.BIGINT, abi.BIGINT], [0], aggs)
    ref.add_page(host)
    expected = ref.build_result().to_rows()
    dev = upload_page(host)
    (page,) = stable_regions(dev, [0, n])
    in_use = C.c_int64()
    check(lib().pa_memory_stats(C.byref(in_use), None, None))
    try:
        # room for ONE HBM table of 8 M expected groups (16 M slots x 24 B ~ 400 MB; beyond the partition-owned tier's range, so
        # the table is allocated with the first page) next to what is already held
        check(lib().pa_memory_set_limit(in_use.value + (640 << 20)))
        first = HashAggregationOperator([abi.BIGINT, abi.BIGINT], [0], aggs, expected_groups=8_000_000)
        first.addInput(page)
        second = HashAggregationOperator([abi.BIGINT, abi.BIGINT], [0], aggs, expected_groups=8_000_000)
        assert second.needsInput()
        check(lib().pa_op_add_input(second._h, C.byref(page.to_c()[0])))   # taken, but put aside: no HBM for its table
        assert second.isBlocked() and not second.needsInput()
        assert second.isBlocked()                                          # still blocked: nothing was released
        first.finish()
        rows_first = [r for p in to_pages(first, []) for r in p.to_rows()]
        first.close()                                                      # releases its table
        assert not second.isBlocked() and second.needsInput()
        second.finish()
        rows_second = []
        while not second.isFinished():
            out = second.getOutput()
            if out is not None:
                rows_second += out.to_rows()
        second.close()
        rows_equal_ignore_order(rows_first, expected)
        rows_equal_ignore_order(rows_second, expected)
    finally:
        check(lib().pa_memory_set_limit(0))
