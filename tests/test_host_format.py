"""Hit-line number formatting: the "%.1f" / "%.3g" fast paths of the host mirror (DSSAligner::AppendUserField) must give
printf's text for every value; rsk_selftest_format compares them with snprintf on pseudo-random values drawn from the
distributions hit lines carry, decimal ties and their floating-point neighbours included.  CPU only."""
from reseek_amd import capi


def test_fast_formatters_equal_printf():
    lib = capi.lib()
    for seed in (1, 2, 3):
        bad = lib.rsk_selftest_format(seed, 1500000)
        assert bad == 0, lib.rsk_last_error().decode()
