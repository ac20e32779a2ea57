"""Host-side tables of hdp_amd.calendar against the reference-generated fixtures (CPU)."""
import os

import numpy as np
import pytest

from hdp_amd import calendar as cal
from oracle import hdp_oracle as orc
from tests.test_oracle_golden import WINDOW_CASES


@pytest.mark.parametrize("name", sorted(WINDOW_CASES))
def test_window_columns_expand_to_reference_table(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "window_tables.npz"))
    s, e, r = WINDOW_CASES[name]
    dates = orc.noleap_date_range(s, e)
    ti, cols = cal.window_columns(dates, r)
    assert ti.dtype == np.int64 and cols.dtype == np.int32
    assert np.array_equal(cal.expand_window_table(ti, cols), g[name])
    assert np.array_equal(cal.datetimes_to_windows(dates, r), g[name])


def test_reflected_upper_edge_and_first_occurrence_order():
    dates = orc.noleap_date_range("2001-03-15", "2004-03-14")
    ti, cols = cal.window_columns(dates, 7)
    assert ti[0, 0] == 0 and dates[0].dayofyr == 74          # row 0 is the first timestamp's doy
    # window index w gathers row d + r - w; past the end it is reflected to n_doy - (d + r - w)
    assert list(cols[364]) == [359, 360, 361, 362, 363, 364, 0, 364, 363, 362, 361, 360, 359, 358, 357]
    # row 364 of a 365-row table gathers rows {0 (once), 364..357} and the reflected 358..364
    full = cal.datetimes_to_windows(dates, 7)
    assert np.array_equal(full, orc.datetimes_to_windows(dates, 7))


def test_season_tables_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "season_tables.npz"))
    for tag in ("50yr", "midyear", "short"):
        s, e = g[f"{tag}_range"]
        dates = orc.noleap_date_range(str(s), str(e))
        assert np.array_equal(cal.get_range_indices(dates, (5, 1), (10, 1)), g[f"{tag}_north"])
        assert np.array_equal(cal.get_range_indices(dates, (11, 1), (4, 1)), g[f"{tag}_south"])
        assert np.array_equal(cal.build_doy_map(dates)[:400], g[f"{tag}_doy_map_head"])
        n1, s1, y1 = cal.hemisphere_season_tables(dates)
        n2, s2, y2 = orc.hemisphere_ranges(dates)
        assert np.array_equal(n1, n2) and np.array_equal(s1, s2) and np.array_equal(y1, y2)
