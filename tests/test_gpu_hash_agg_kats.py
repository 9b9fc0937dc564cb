"""TestHashAggregationOperator's known-answer tests through the device operator (pa_hash_aggregation_create), input for input as
tests/test_oracle_operators.py restates them for the oracle:
    testHashAggregation              …/TestHashAggregationOperator.java:160-219   all six aggregates, hashEnabled both ways
    testHashAggregationWithGlobals   :221-272   globalAggregationGroupIds / groupIdChannel / produceDefaultOutput
    testHashBuilderResize            :360-399   a 200 000-byte key between two small pages
    testMultiSliceAggregationOutput  :477-510   49 152 groups
Rows are compared ignoring order (assertPagesEqualIgnoreOrder); page boundaries are not part of parity on the device path (SURVEY a3 /
9.2: MergePages re-chunks) -- the two-page shape of testMultiSliceAggregationOutput is asserted on the oracle's operator."""
import pytest

from presto_amd import abi
from presto_amd.operators import HashAggregationOperator, download_page, to_pages, upload_page
from presto_amd.page import Block, Page
from tests.test_oracle_operators import (GLOBALS_AGGREGATES, GLOBALS_EXPECTED, GLOBALS_TYPES, HAGG1_AGGREGATES, MULTI_SLICE_POSITIONS, check_hagg1_rows,
                                         check_hash_builder_resize_rows, drive, globals_hash, hagg1_input, hash_builder_resize_pages, multi_slice_input)

pytestmark = pytest.mark.gpu


def rows_of(pages):
    return [r for p in pages for r in (download_page(p) if p.mem == abi.MEM_DEVICE else p).to_rows()]


@pytest.mark.parametrize("hashed", [False, True])
@pytest.mark.parametrize("device_pages", [False, True])
def test_hagg1_all_six_aggregates(gpu, oracle, hashed, device_pages):
    types, pages, hc = hagg1_input(oracle, hashed)
    if device_pages:
        pages = [upload_page(p) for p in pages]
    op = HashAggregationOperator(types, [1], HAGG1_AGGREGATES, hash_channel=hc, expected_groups=100_000,
                                 output_mem=abi.MEM_DEVICE if device_pages else abi.MEM_HOST)
    check_hagg1_rows(oracle, rows_of(to_pages(op, pages)), hashed)
    op.close()


@pytest.mark.parametrize("hashed", [False, True])
@pytest.mark.parametrize("output_mem", [abi.MEM_HOST, abi.MEM_DEVICE])
def test_hash_aggregation_with_globals(gpu, oracle, hashed, output_mem):
    types, hc = (GLOBALS_TYPES + [abi.BIGINT], 7) if hashed else (GLOBALS_TYPES, -1)
    kw = dict(hash_channel=hc, expected_groups=100_000, global_aggregation_group_ids=[42, 49], group_id_channel=1)
    op = HashAggregationOperator(types, [1, 2], GLOBALS_AGGREGATES, produce_default_output=True, output_mem=output_mem, **kw)
    rows = rows_of(to_pages(op, []))
    op.close()
    ref = oracle.HashAggregationOperator(types, [1, 2], GLOBALS_AGGREGATES, produce_default_output=True, **kw)
    assert rows == [r for p in drive(ref, []) for r in p.to_rows()]
    if hashed:
        assert [r[2] for r in rows] == [globals_hash(oracle, 42), globals_hash(oracle, 49)]
        rows = [r[:2] + r[3:] for r in rows]
    assert rows == GLOBALS_EXPECTED


def test_default_rows_only_without_input_and_only_when_asked(gpu, oracle):
    kw = dict(global_aggregation_group_ids=[42, 49], group_id_channel=1)
    page = Page([Block.varchar(["a"]), Block.varchar(["k"]), Block.bigint([7]), Block.bigint([1]), Block.bigint([5]), Block.boolean([True]),
                 Block.varchar(["z"])], 1)
    op = HashAggregationOperator(GLOBALS_TYPES, [1, 2], GLOBALS_AGGREGATES, produce_default_output=True, **kw)
    assert rows_of(to_pages(op, [page])) == [(b"k", 7, 1, 5, 5.0, b"z", 1, 1)]   # inputProcessed (HashAggregationOperator.java:386, 488)
    op.close()
    op = HashAggregationOperator(GLOBALS_TYPES, [1, 2], GLOBALS_AGGREGATES, **kw)
    assert to_pages(op, []) == []
    op.close()
    op = HashAggregationOperator(GLOBALS_TYPES, [1, 2], GLOBALS_AGGREGATES, produce_default_output=True)   # no global grouping sets: output.isEmpty()
    assert to_pages(op, []) == []
    op.close()


@pytest.mark.parametrize("state_format", [abi.STATES_FLAT, abi.STATES_REFERENCE])
def test_partial_step_emits_the_empty_intermediate_states(gpu, oracle, state_format):
    """evaluateIntermediate over fresh accumulators (HashAggregationOperator.java:576-578), in both intermediate formats; a FINAL step fed
    with those rows of two PARTIAL operators gives the SINGLE step's default rows back."""
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_AVG, 3, abi.DOUBLE), (abi.AGG_MIN, 2, abi.BIGINT), (abi.AGG_COUNT, 0, abi.VARCHAR)]
    types = [abi.VARCHAR, abi.BIGINT, abi.BIGINT, abi.DOUBLE]
    kw = dict(global_aggregation_group_ids=[7, 9], group_id_channel=1, produce_default_output=True)
    part = HashAggregationOperator(types, [0, 1], aggs, step=abi.STEP_PARTIAL, state_format=state_format, **kw)
    pages = to_pages(part, [])
    part.close()
    assert len(pages) == 1 and pages[0].position_count == 2
    if state_format == abi.STATES_FLAT:
        ref = oracle.HashAggregationOperator(types, [0, 1], aggs, step=abi.STEP_PARTIAL, **kw)
        assert rows_of(pages) == [r for p in drive(ref, []) for r in p.to_rows()]
        from presto_amd.exchange import partial_layout
        ptypes, faggs = partial_layout([abi.VARCHAR, abi.BIGINT], aggs)
        final = HashAggregationOperator(ptypes, [0, 1], faggs, step=abi.STEP_FINAL)
        assert sorted(rows_of(to_pages(final, pages + pages)), key=lambda r: r[1]) == [(None, 7, 0, None, None, None, 0), (None, 9, 0, None, None, None, 0)]
        final.close()
    else:
        assert [r[:3] for r in rows_of(pages)] == [(None, 7, 0), (None, 9, 0)]


@pytest.mark.parametrize("hashed", [False, True])
def test_hash_builder_resize(gpu, oracle, hashed):
    types, pages, hc = hash_builder_resize_pages(oracle, hashed)
    op = HashAggregationOperator(types, [0], [(abi.AGG_COUNT_STAR, -1, None)], hash_channel=hc, expected_groups=100_000)
    check_hash_builder_resize_rows(rows_of(to_pages(op, pages)), hashed)
    op.close()


@pytest.mark.parametrize("hashed", [False, True])
def test_multi_slice_aggregation_output(gpu, oracle, hashed):
    types, pages, hc = multi_slice_input(oracle, hashed)
    op = HashAggregationOperator(types, [1], [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_AVG, 1, abi.BIGINT)], hash_channel=hc, expected_groups=100_000)
    rows = rows_of(to_pages(op, pages))
    op.close()
    assert sorted((r[0], r[-2], r[-1]) for r in rows) == [(i, 1, float(i)) for i in range(MULTI_SLICE_POSITIONS)]
    if hashed:
        assert all(r[1] == oracle.hash_bigint(r[0]) for r in rows)


def test_group_by_hash_append_to(gpu, oracle):
    """TestGroupByHash.testAppendTo / testAppendToMultipleTuplesPerGroup (…/operator/TestGroupByHash.java:151-200): the keys (and, with a
    hash channel, their $hashvalue) that appendValuesTo writes per group -- on the device: the output of a HashAggregation without aggregates.
    VARCHAR "0" .. "99" with precomputed hashes -> 100 rows (value, hash); BIGINT i % 50 over 100 rows -> the 50 values 0 .. 49.  (Group ids are
    first-seen ordinals on the reference; the device has no group ids to show, rows are compared as a set.)"""
    from presto_amd.page import sequence_page
    values = sequence_page(100, [(abi.VARCHAR, 0)])
    hashes = oracle.hash_page(values, [0])
    page = Page(values.blocks + [Block.bigint(hashes)], 100)
    op = HashAggregationOperator([abi.VARCHAR, abi.BIGINT], [0], [], hash_channel=1, expected_groups=100)
    rows = rows_of(to_pages(op, [page]))
    op.close()
    assert sorted(rows) == sorted((str(i).encode(), int(hashes[i])) for i in range(100))
    ref = oracle.HashAggregation([abi.VARCHAR, abi.BIGINT], [0], [], hash_channel=1, expected_groups=100)
    ids = ref.add_page(page, want_group_ids=True)
    assert ids.tolist() == list(range(100)) and ref.build_result().to_rows() == [(str(i).encode(), int(hashes[i])) for i in range(100)]
    many = Page([Block.bigint([i % 50 for i in range(100)])], 100)
    many = Page(many.blocks + [Block.bigint(oracle.hash_page(many, [0]))], 100)
    op = HashAggregationOperator([abi.BIGINT, abi.BIGINT], [0], [], hash_channel=1, expected_groups=100)
    rows = rows_of(to_pages(op, [many]))
    op.close()
    assert sorted(r[0] for r in rows) == list(range(50)) and all(r[1] == oracle.hash_bigint(r[0]) for r in rows)
