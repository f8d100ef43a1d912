"""The result files of the two stages, written through the C ABI (gtx_write_table_*, gtx_write_csv: csrc/table_writer.cpp)
byte for byte as the reference writes them: np.savetxt with '%g' / '%.16g' / '%.20g' (geotrax/extract.py:497-516,
georeference.py:879-889) and pandas.DataFrame.to_csv(index=False) for the georeferenced table (georeference.py:868-877).

Why: on a 7 000-frame video (~900 k track rows) np.savetxt takes 3 s and DataFrame.to_csv 9 s of interpreter time, next to 7 s
of GPU time for the whole extraction; the same bytes leave the library's threads in a few tenths of a second."""
from __future__ import annotations

import csv
import ctypes as C
import io
from pathlib import Path

import numpy as np

from . import _lib


def savetxt(path, table: np.ndarray, precision: int) -> None:
    """np.savetxt(path, table, fmt='%.<precision>g', delimiter=',') for a 2-D (or 1-D: one value per line, like np.savetxt)
    float32 / float64 array; other dtypes are written as float64, which is what '%g' % value does with them."""
    a = np.asarray(table)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    if a.ndim != 2:
        raise ValueError(f"expected a 1-D or 2-D array, got {a.ndim}-D")
    if a.dtype != np.float32:
        a = a.astype(np.float64, copy=False)
    a = np.ascontiguousarray(a)
    lib = _lib.load()
    fn = lib.gtx_write_table_f32 if a.dtype == np.float32 else lib.gtx_write_table_f64
    if a.shape[1] == 0:                                          # np.savetxt writes one empty line per row
        Path(path).write_text("\n" * a.shape[0])
        return
    _lib.check(fn(str(path).encode(), _lib.ptr(a), a.shape[0], a.shape[1], int(precision), 0))


def _csv_cell(value) -> bytes:
    """One string cell the way csv.writer(QUOTE_MINIMAL) -- pandas' writer -- writes it."""
    if value == "":
        return b""                                               # (csv.writer quotes an empty string only when it is the row's single field)
    buf = io.StringIO()
    csv.writer(buf, lineterminator="\n").writerow([value])      # pandas' dialect: QUOTE_MINIMAL, '"', doubled quotes, "\n" rows
    return buf.getvalue()[:-1].encode("utf-8")


def dataframe_to_csv(path, df) -> bool:
    """df.to_csv(path, index=False) for a frame of int64, float64 and string columns. Returns False -- nothing written -- when
    the frame holds anything else (the caller then uses pandas itself)."""
    import pandas as pd

    n = len(df)
    if len(df.columns) < 2:                                      # (a lone empty-string cell is quoted by csv.writer: not worth a special case)
        return False
    kinds, cols, keep, cats, n_cats = [], [], [], [], []
    for name in df.columns:
        s = df[name]
        if s.dtype == np.int64:
            kinds.append(0)
            cols.append(np.ascontiguousarray(s.to_numpy()))
            cats.append(None)
        elif s.dtype == np.float64:
            kinds.append(1)
            cols.append(np.ascontiguousarray(s.to_numpy()))
            cats.append(None)
        elif s.dtype == object:
            codes, uniques = pd.factorize(s, use_na_sentinel=True)
            if len(uniques) > 65536 or not all(isinstance(u, str) for u in uniques):   # free text rather than a label column: pandas' business
                return False
            kinds.append(2)
            cols.append(np.ascontiguousarray(codes, dtype=np.int32))
            cats.append([_csv_cell(u) for u in uniques])
        else:
            return False
    header = b",".join(_csv_cell(str(c)) for c in df.columns)
    cat_arrays = []
    for c in cats:
        if c is None:
            cat_arrays.append(None)
            n_cats.append(0)
        else:
            arr = (C.c_char_p * max(len(c), 1))(*c)
            keep.append(arr)
            cat_arrays.append(C.cast(arr, C.c_void_p))
            n_cats.append(len(c))
    col_ptrs = (C.c_void_p * len(cols))(*[a.ctypes.data for a in cols])
    cat_ptrs = (C.c_void_p * len(cols))(*[(p.value if p is not None else None) for p in cat_arrays])
    kinds_a = (C.c_int * len(cols))(*kinds)
    ncat_a = (C.c_int * len(cols))(*n_cats)
    _lib.check(_lib.load().gtx_write_csv(str(path).encode(), header, len(cols), kinds_a, col_ptrs, cat_ptrs, ncat_a, n, 0))
    return True
