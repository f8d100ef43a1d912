"""csrc/table_writer.cpp through geotrax_amd.tables: the same bytes as np.savetxt('%g' / '%.16g' / '%.20g', ',') and as
pandas.DataFrame.to_csv(index=False) (geotrax/extract.py:497-516, georeference.py:868-889). Host code: runs without a GPU."""
import filecmp
import struct
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-trax_amd"))


def _awkward_doubles(rng, n):
    """Values where digit generation and notation choices can go wrong: every binade edge, halfway cases of the rounding,
    subnormals, powers of ten and their neighbours, integers, negative zero, infinities, NaNs of both signs."""
    vals = [0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9999999999999998.0, 1e17, 1e22, 1e23, 5e-324, 2.2250738585072014e-308,
            1.7976931348623157e308, 123456.5, 1234565.0, 0.000123456789, 100000.0, 999999.5, 1000000.0, 0.30000000000000004, 2.5, 3.5, 1e21, 1e-7,
            float("inf"), float("-inf"), float("nan"), struct.unpack("<d", struct.pack("<Q", 0xFFF8000000000001))[0]]
    for p in range(-30, 31):
        for d in (np.nextafter(10.0 ** p, 0), 10.0 ** p, np.nextafter(10.0 ** p, np.inf)):
            vals.append(float(d))
    vals += list(rng.standard_normal(n) * 10.0 ** rng.integers(-12, 20, n))
    vals += list(np.round(rng.uniform(-1e5, 1e5, n), rng.integers(0, 8, n)[0]))
    vals += list(rng.integers(-10 ** 9, 10 ** 9, n).astype(np.float64))
    bits = rng.integers(0, 2 ** 63, n, dtype=np.int64).view(np.float64)
    vals += [float(b) for b in bits if np.isfinite(b)]
    return np.array(vals, dtype=np.float64)


@pytest.mark.parametrize("precision,fmt", [(6, "%g"), (16, "%.16g"), (20, "%.20g"), (3, "%.3g")])
def test_savetxt_float64_is_np_savetxt(tmp_path, precision, fmt):
    from geotrax_amd import tables

    rng = np.random.default_rng(precision)
    v = _awkward_doubles(rng, 4000)
    v = v[: len(v) // 10 * 10].reshape(-1, 10)
    tables.savetxt(tmp_path / "a.txt", v, precision)
    np.savetxt(tmp_path / "b.txt", v, fmt=fmt, delimiter=",")
    assert filecmp.cmp(tmp_path / "a.txt", tmp_path / "b.txt", shallow=False)


def test_savetxt_float32_tracks_are_np_savetxt(tmp_path):
    """The tracks table is float32 and '%g' (extract.py:509): float32 -> Python float -> '%g'. More rows than one chunk of the
    writer's threads, so the chunk seams are in the file."""
    from geotrax_amd import tables

    rng = np.random.default_rng(1)
    n = 70_000
    t = np.empty((n, 12), np.float32)
    t[:, 0] = np.repeat(np.arange(n // 130 + 1), 130)[:n]
    t[:, 1] = rng.integers(1, 3000, n)
    t[:, 2:10] = rng.uniform(-50, 3900, (n, 8))
    t[:, 10] = rng.integers(0, 4, n)
    t[:, 11] = rng.uniform(0.25, 1, n)
    t[::97, 6:10] = np.nan
    t[5, 2], t[6, 3], t[7, 4] = np.inf, -np.inf, -0.0
    t[8, 5] = np.float32(1e-30)
    t[9, 5] = np.float32(3.4e38)
    tables.savetxt(tmp_path / "a.txt", t, 6)
    np.savetxt(tmp_path / "b.txt", t, fmt="%g", delimiter=",")
    assert filecmp.cmp(tmp_path / "a.txt", tmp_path / "b.txt", shallow=False)
    for shape in [(0, 12), (1, 12), (3, 1)]:                  # no rows, one row, one column
        e = rng.standard_normal(shape).astype(np.float32)
        tables.savetxt(tmp_path / "c.txt", e, 6)
        np.savetxt(tmp_path / "d.txt", e, fmt="%g", delimiter=",")
        assert filecmp.cmp(tmp_path / "c.txt", tmp_path / "d.txt", shallow=False), shape
    tables.savetxt(tmp_path / "c.txt", np.arange(5), 6)       # 1-D, integers: one value per line
    np.savetxt(tmp_path / "d.txt", np.arange(5), fmt="%g", delimiter=",")
    assert filecmp.cmp(tmp_path / "c.txt", tmp_path / "d.txt", shallow=False)


def test_savetxt_reports_an_unwritable_path(tmp_path):
    from geotrax_amd import tables

    with pytest.raises(RuntimeError, match="cannot open"):
        tables.savetxt(tmp_path / "no_such_dir" / "a.txt", np.zeros((2, 2)), 6)


def test_dataframe_to_csv_is_pandas_to_csv(tmp_path):
    import pandas as pd
    from geotrax_amd import tables

    rng = np.random.default_rng(2)
    v = _awkward_doubles(rng, 6000)
    n = len(v)
    lane = np.where(rng.random(n) < 0.8, rng.integers(0, 5, n).astype(float), np.nan)
    df = pd.DataFrame({
        "Vehicle_ID": rng.integers(1, 5000, n), "Weird": v, "Rounded_1": np.round(rng.uniform(-2e4, 2e4, n), 1), "Rounded_7": np.round(37 + rng.uniform(0, 0.01, n), 7),
        "Neg": -np.abs(np.round(rng.standard_normal(n), 2)), 'Section, "quoted"': rng.choice(["A", "B,1", 'say "hi"', "", "line\nbreak", "ünï"], n),
        "Lane_Number": pd.Series(lane).apply(lambda x: str(int(x)) if pd.notna(x) else ""), "With_None": rng.choice(np.array(["x", None, "y"], dtype=object), n),
        "Big": rng.integers(-2 ** 62, 2 ** 62, n), "Visibility": rng.integers(0, 2, n),
    })
    assert tables.dataframe_to_csv(tmp_path / "a.csv", df)
    df.to_csv(tmp_path / "b.csv", index=False)
    assert filecmp.cmp(tmp_path / "a.csv", tmp_path / "b.csv", shallow=False)
    assert tables.dataframe_to_csv(tmp_path / "a.csv", df.iloc[:0])          # no rows: the header line only
    df.iloc[:0].to_csv(tmp_path / "b.csv", index=False)
    assert filecmp.cmp(tmp_path / "a.csv", tmp_path / "b.csv", shallow=False)
    big = pd.concat([df] * 12, ignore_index=True)                            # several chunks of the writer's threads
    assert tables.dataframe_to_csv(tmp_path / "a.csv", big)
    big.to_csv(tmp_path / "b.csv", index=False)
    assert filecmp.cmp(tmp_path / "a.csv", tmp_path / "b.csv", shallow=False)
    # frames the writer does not take: pandas writes them (save_georeferenced_data falls through)
    assert not tables.dataframe_to_csv(tmp_path / "a.csv", pd.DataFrame({"a": [True, False], "b": [1, 2]}))
    assert not tables.dataframe_to_csv(tmp_path / "a.csv", pd.DataFrame({"a": pd.to_datetime(["2024-01-01", "2024-01-02"]), "b": [1, 2]}))
    assert not tables.dataframe_to_csv(tmp_path / "a.csv", pd.DataFrame({"a": np.array([1, 2], np.int32), "b": [1, 2]}))
    assert not tables.dataframe_to_csv(tmp_path / "a.csv", pd.DataFrame({"a": [1.5, 2.5]}))


def test_the_georeferenced_table_goes_through_the_library_writer_unchanged(tmp_path):
    """create_and_format_georeferenced_df's real column set (georeference.py:802-866) -> save_georeferenced_data: same file as pandas'."""
    import logging

    from geotrax_amd.georeference import create_and_format_georeferenced_df, save_georeferenced_data

    rng = np.random.default_rng(3)
    n = 40_000
    log = logging.getLogger("t")
    f = lambda s: rng.uniform(0, s, n)                         # noqa: E731
    df = create_and_format_georeferenced_df(np.sort(rng.integers(1, 900, n)), np.array([]), rng.integers(0, 7000, n), f(15000), f(15000), f(500) - 250, f(500), 37 + f(0.01),
                                            126 + f(0.01), (f(6), f(2.5)), rng.integers(0, 4, n), f(60), f(3) - 1.5, rng.choice(["A", "B", "C1"], n),
                                            np.where(rng.random(n) < 0.9, rng.integers(0, 4, n).astype(float), np.nan), rng.integers(0, 2, n), 15,
                                            rng.integers(0, 2, n), logger=log)
    from geotrax_amd import tables

    assert tables.dataframe_to_csv(tmp_path / "direct.csv", df)            # this schema is one the writer takes
    save_georeferenced_data(tmp_path / "a.csv", df, log)
    df.to_csv(tmp_path / "b.csv", index=False)
    assert filecmp.cmp(tmp_path / "a.csv", tmp_path / "b.csv", shallow=False)
