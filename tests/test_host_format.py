"""Hit-line number formatting: the "%.1f" / "%.3g" fast paths of the host mirror (DSSAligner::AppendUserField) must give
printf's text for every value; rsk_selftest_format compares them with snprintf on pseudo-random values drawn from the
distributions hit lines carry, decimal ties and their floating-point neighbours included.  CPU only."""
from reseek_amd import capi


def test_fast_formatters_equal_printf():
    lib = capi.lib()
    for seed in (1, 2, 3):
        bad = lib.rsk_selftest_format(seed, 1500000)
        assert bad == 0, lib.rsk_last_error().decode()


def test_literal_fastdb_reference_run_equals_the_split_route_golden():
    """tests/golden/full11211_fastdb.md5.txt (the golden tests/test_gpu_full_golden.py requires of rsk_search) was assembled
    from one-thread reference processes over 96 target ranges; the literal one-piece command `reseek -search Q -db Q -fast
    -keeptmp -threads 1` (search.cpp:62-111; 5.5 CPU-hours, tests/golden/make_db_goldens.py literal) gives the same input md5,
    hand-off file (bytes and md5), row count and sorted-table md5."""
    import json
    import os
    import fixtures as fx
    lit = json.load(open(os.path.join(fx.GOLDEN, "full11211_fastdb_literal.md5.txt")))
    gold = json.load(open(os.path.join(fx.GOLDEN, "full11211_fastdb.md5.txt")))
    assert lit["bca_md5"] == gold["bca_md5"]
    assert (lit["handoff_bytes"], lit["handoff_md5"]) == (gold["handoff_bytes"], gold["handoff_md5"])
    assert (lit["rows"], lit["sorted_table_md5"]) == (gold["rows"], gold["sorted_table_md5"])
    assert all(lit["equals_split_route_golden"].values())
