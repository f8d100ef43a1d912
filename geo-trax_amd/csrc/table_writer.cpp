// Text tables of the extract / georeference stages, written byte for byte like the reference writes them, without the
// interpreter in the loop. Host code only (no HIP call): rows are formatted on a few threads into per-chunk buffers and
// written in order.
//
//   gtx_write_table_f32 / _f64   np.savetxt(path, table, fmt='%.<p>g', delimiter=',')   geotrax/extract.py:497-516
//                                (tracks: '%g' on float32 rows; transforms: '%.16g' on float64 rows). A 7 000-frame video
//                                writes ~900 k x 12 values: 3.1 s through np.savetxt, 0.1-0.3 s here.
//   gtx_write_csv                pandas.DataFrame.to_csv(path, index=False) for a frame whose columns are int64, float64
//                                or strings drawn from a small set (the georeferenced table, georeference.py:802-877):
//                                9 s through pandas for the same video.
//
// Formats. '%.<p>g' % float(x) in Python, printf("%.<p>g", x) and std::to_chars(x, general, p) agree on every finite double (all
// round the exact binary value correctly and write at least two exponent digits) and on +-inf; Python writes every NaN as "nan",
// the C side may write "-nan": handled here. tests/test_tables.py holds both writers against np.savetxt / pandas on binade edges,
// rounding ties, subnormals, powers of ten and their neighbours, and random bit patterns.
// pandas writes a float64 cell as repr(float): the shortest digit string that reads back to the same double, positional
// for 1e-4 <= |x| < 1e16, else d[.ddd]e+XX, integral values with a trailing ".0"; NaN as the empty string.
#include <algorithm>
#include <atomic>
#include <charconv>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gtx.h"
#include "api_guard.hpp"
#include "common.hpp"

namespace {
using gtx::guarded;

// fn(first row, last row + 1, buffer) formats a chunk of rows; chunks are formatted on n_threads threads, each into a buffer it
// reuses, and written in row order (a thread waits for its chunk's turn): memory stays at one chunk per thread.
template <class F>
void write_chunked(const char* path, const std::string& head, int64_t rows, int n_threads, F fn) {
  if (!path) gtx::fail(GTX_ERR_INVALID, "path is null");
  const int64_t per = 8192;
  const int n_chunks = (int)((rows + per - 1) / per);
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = std::max(1, std::min({n_threads > 0 ? n_threads : 8, (int)(hw ? hw : 1), std::max(n_chunks, 1)}));
  FILE* f = std::fopen(path, "wb");
  if (!f) gtx::fail(GTX_ERR_INVALID, "cannot open '%s' for writing", path);
  bool ok = head.empty() || std::fwrite(head.data(), 1, head.size(), f) == head.size();
  std::atomic<int> next{0};
  std::mutex mu;
  std::condition_variable cv;
  int turn = 0;                                                // the chunk whose bytes go to the file next
  bool failed = false;
  std::exception_ptr first_error;
  auto work = [&] {
    std::string buf;
    for (int c = next.fetch_add(1); c < n_chunks; c = next.fetch_add(1)) {
      bool mine_ok = true;
      try {
        buf.clear();
        fn(c * per, std::min(rows, (c + 1) * per), buf);
      } catch (...) {                                          // (out of memory): nothing may leave a worker thread
        mine_ok = false;
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) first_error = std::current_exception();
        failed = true;
      }
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return turn == c; });
      if (mine_ok && !failed && ok) ok = buf.empty() || std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
      ++turn;
      cv.notify_all();
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  ok = (std::fclose(f) == 0) && ok;
  if (failed) std::rethrow_exception(first_error);
  if (!ok) gtx::fail(GTX_ERR_INVALID, "short write to '%s'", path);
}

inline void put_g(std::string& o, double v, int prec) {
  if (std::isnan(v)) { o += "nan"; return; }
  char b[80];                                                  // std::to_chars(general, precision) is printf's %.<p>g in the C locale, without printf
  const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::general, prec);
  o.append(b, (size_t)(r.ptr - b));
}

template <class T>
void write_table(const char* path, const T* data, int64_t rows, int cols, int precision, int n_threads) {
  if (rows < 0 || cols < 1 || precision < 1 || precision > 30) gtx::fail(GTX_ERR_INVALID, "write_table: rows %lld, cols %d, precision %d", (long long)rows, cols, precision);
  if (rows > 0 && !data) gtx::fail(GTX_ERR_INVALID, "data is null");
  write_chunked(path, std::string(), rows, n_threads, [&](int64_t r0, int64_t r1, std::string& o) {
    o.reserve((size_t)(r1 - r0) * (size_t)cols * 10);
    for (int64_t r = r0; r < r1; ++r) {
      const T* row = data + r * cols;
      for (int c = 0; c < cols; ++c) {
        if (c) o.push_back(',');
        put_g(o, (double)row[c], precision);
      }
      o.push_back('\n');
    }
  });
}

// repr(float) of Python / str(numpy.float64): shortest round-trip digits, Python's choice of notation
inline void put_repr(std::string& o, double v) {
  if (std::isnan(v)) return;                                  // pandas: na_rep = ''
  if (std::isinf(v)) { o += v < 0 ? "-inf" : "inf"; return; }
  if (v == 0.0) { o += std::signbit(v) ? "-0.0" : "0.0"; return; }
  char b[40];
  const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest
  const char* p = b;
  if (*p == '-') { o.push_back('-'); ++p; }
  const char* e = p;
  while (e < r.ptr && *e != 'e') ++e;
  char dig[24];
  int nd = 0;
  for (const char* q = p; q < e; ++q)
    if (*q != '.') dig[nd++] = *q;
  int x = 0;
  std::from_chars(e + 1 + (e[1] == '+' ? 1 : 0), r.ptr, x);   // decimal exponent of the first digit
  if (x >= -4 && x < 16) {
    if (x < 0) {
      o += "0.";
      o.append((size_t)(-x - 1), '0');
      o.append(dig, (size_t)nd);
    } else if (nd <= x + 1) {
      o.append(dig, (size_t)nd);
      o.append((size_t)(x + 1 - nd), '0');
      o += ".0";
    } else {
      o.append(dig, (size_t)(x + 1));
      o.push_back('.');
      o.append(dig + x + 1, (size_t)(nd - x - 1));
    }
  } else {
    o.push_back(dig[0]);
    if (nd > 1) { o.push_back('.'); o.append(dig + 1, (size_t)(nd - 1)); }
    o.push_back('e');
    o.push_back(x < 0 ? '-' : '+');
    const int ax = x < 0 ? -x : x;
    if (ax < 10) o.push_back('0');
    o += std::to_string(ax);
  }
}

inline void put_i64(std::string& o, int64_t v) {
  char b[24];
  const auto r = std::to_chars(b, b + sizeof b, v);
  o.append(b, (size_t)(r.ptr - b));
}
}  // namespace

extern "C" {

int gtx_write_table_f32(const char* path, const float* data, int64_t rows, int cols, int precision, int n_threads) {
  return guarded([&] { write_table(path, data, rows, cols, precision, n_threads); });
}

int gtx_write_table_f64(const char* path, const double* data, int64_t rows, int cols, int precision, int n_threads) {
  return guarded([&] { write_table(path, data, rows, cols, precision, n_threads); });
}

int gtx_write_csv(const char* path, const char* header_line, int n_cols, const int* kinds, const void* const* columns,
                  const char* const* const* categories, const int* n_categories, int64_t rows, int n_threads) {
  return guarded([&] {
    if (n_cols < 1 || rows < 0 || !kinds || !columns) gtx::fail(GTX_ERR_INVALID, "write_csv: %d columns, %lld rows", n_cols, (long long)rows);
    for (int c = 0; c < n_cols; ++c) {
      if (kinds[c] < 0 || kinds[c] > 2) gtx::fail(GTX_ERR_INVALID, "write_csv: column %d has kind %d (0 int64, 1 float64, 2 categorical)", c, kinds[c]);
      if (rows > 0 && !columns[c]) gtx::fail(GTX_ERR_INVALID, "write_csv: column %d is null", c);
      if (kinds[c] == 2 && (!categories || !n_categories || (n_categories[c] > 0 && !categories[c])))
        gtx::fail(GTX_ERR_INVALID, "write_csv: column %d is categorical and has no category table", c);
    }
    std::string head = header_line ? header_line : "";
    if (!head.empty() && head.back() != '\n') head.push_back('\n');
    for (int c = 0; c < n_cols; ++c)                           // codes are checked here, on the calling thread: a worker must not throw
      if (kinds[c] == 2) {
        const int32_t* k = static_cast<const int32_t*>(columns[c]);
        for (int64_t r = 0; r < rows; ++r)
          if (k[r] >= n_categories[c]) gtx::fail(GTX_ERR_INVALID, "write_csv: column %d row %lld: code %d of %d categories", c, (long long)r, k[r], n_categories[c]);
      }
    write_chunked(path, head, rows, n_threads, [&](int64_t r0, int64_t r1, std::string& o) {
      o.reserve((size_t)(r1 - r0) * (size_t)n_cols * 8);
      for (int64_t r = r0; r < r1; ++r) {
        for (int c = 0; c < n_cols; ++c) {
          if (c) o.push_back(',');
          if (kinds[c] == 0) {
            put_i64(o, static_cast<const int64_t*>(columns[c])[r]);
          } else if (kinds[c] == 1) {
            put_repr(o, static_cast<const double*>(columns[c])[r]);
          } else {
            const int32_t k = static_cast<const int32_t*>(columns[c])[r];
            if (k >= 0) o += categories[c][k];               // a negative code is a missing value: empty cell
          }
        }
        o.push_back('\n');
      }
    });
  });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// estimate_vehicle_dimensions' walk over a track (geotrax/extract.py:433-452), for every track of a table in one call.
// Per track: an anchor point; the first later observation at least `radius` pixels away becomes the next anchor. The reference
// does this with NumPy float32 scalars, one Python iteration per row (1.5 s of interpreter time on a 7 000-frame video); the
// arithmetic below is the same sequence of float32 operations: x ** 2 of a float32 scalar is glibc's powf(x, 2.0f) -- which is
// NOT x * x in 0.08 % of cases -- so powf is called through a volatile pointer (the compiler would fold it into a multiply),
// then a float32 add, sqrtf, and the comparison against radius in float32 (a Python float is a weak scalar under NumPy 2).
// What the walk does NOT do is the azimuth of an anchor step: NumPy's float32 arctan2 is its own SIMD routine on some CPUs and
// libm's atan2f on others, so the caller computes it with NumPy from the (dx, dy) returned here -- the same ufunc loop the
// reference's scalar call ends in. tests/test_postprocess.py holds the whole thing against the row-by-row form.
namespace {
float (*volatile powf_of_libm)(float, float) = powf;
}

extern "C" int gtx_track_anchor_walk(const float* xc, const float* yc, const int64_t* start, int n_tracks, float radius,
                                     uint8_t* is_anchor, float* step_dx, float* step_dy) {
  return gtx::guarded([&] {
    if (n_tracks < 0 || (n_tracks > 0 && (!xc || !yc || !start || !is_anchor || !step_dx || !step_dy))) gtx::fail(GTX_ERR_INVALID, "anchor walk: null argument");
    for (int t = 0; t < n_tracks; ++t) {
      const int64_t a = start[t], b = start[t + 1];
      if (b < a) gtx::fail(GTX_ERR_INVALID, "anchor walk: track %d has a negative length", t);
      if (b == a) continue;
      std::memset(is_anchor + a, 0, (size_t)(b - a));
      float xp = xc[a], yp = yc[a];
      for (int64_t k = a + 1; k < b; ++k) {
        const float dx = xc[k] - xp, dy = yc[k] - yp;
        const float d = sqrtf(powf_of_libm(dx, 2.0f) + powf_of_libm(dy, 2.0f));
        if (d >= radius) {
          is_anchor[k] = 1;
          step_dx[k] = dx;
          step_dy[k] = dy;
          xp = xc[k];
          yp = yc[k];
        }
      }
    }
  });
}
