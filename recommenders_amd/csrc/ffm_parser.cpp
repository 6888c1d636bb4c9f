// libffm text ingest for FFMTextIterator (host side of libxdfm.so).
//
// Follows the observable behaviour of the reference's pure-Python reader
// (recommenders/models/deeprec/io/iterator.py:90-219):
//   line  := label SP token (SP token)* [ '%' impression_id ]
//   token := field ':' feature ':' value         (1-based field / feature)
// and produces the two COO views of `_convert_data` (:152-219) directly as arrays.
// Text semantics mirrored: universal newlines, str.strip() on the line / words,
// split on the single separator character, empty words skipped (:111-112), the
// impression id only when the line holds exactly one '%' (:101-103), Python's
// int()/float() literal grammar (ASCII subset), float64 -> float32 narrowing on store.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "xdfm_common.h"

namespace xdfm {

namespace {

// whitespace of Python's str.strip() restricted to ASCII
inline bool py_space(unsigned char c) {
  return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f);
}

struct Span {
  const char* b;
  const char* e;
  bool empty() const { return b >= e; }
};

inline Span strip(Span s) {
  while (s.b < s.e && py_space((unsigned char)*s.b)) ++s.b;
  while (s.e > s.b && py_space((unsigned char)s.e[-1])) --s.e;
  return s;
}

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// Python int(str): [ws] [+-] digit ('_'? digit)* [ws]
bool parse_py_int(Span s, int64_t* out) {
  s = strip(s);
  if (s.empty()) return false;
  bool neg = false;
  if (*s.b == '+' || *s.b == '-') {
    neg = (*s.b == '-');
    ++s.b;
  }
  if (s.empty() || !is_digit(*s.b)) return false;
  unsigned long long acc = 0;
  const unsigned long long lim = neg ? 9223372036854775808ull : 9223372036854775807ull;
  bool prev_us = false;
  for (const char* p = s.b; p < s.e; ++p) {
    if (*p == '_') {
      if (prev_us || p + 1 == s.e) return false;
      prev_us = true;
      continue;
    }
    if (!is_digit(*p)) return false;
    prev_us = false;
    unsigned d = (unsigned)(*p - '0');
    if (acc > (lim - d) / 10) return false;  // would not fit int64 (numpy would raise OverflowError)
    acc = acc * 10 + d;
  }
  *out = neg ? (int64_t)(0ull - acc) : (int64_t)acc;
  return true;
}

inline bool ieq(const char* b, const char* e, const char* word) {
  size_t n = strlen(word);
  if ((size_t)(e - b) != n) return false;
  for (size_t i = 0; i < n; ++i) {
    char c = b[i];
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    if (c != word[i]) return false;
  }
  return true;
}

// Python float(str): [ws] [+-] (inf|infinity|nan | digits [. digits?] [exp] | . digits [exp]) [ws]
// digits may carry single underscores between digits.
bool parse_py_float(Span s, double* out) {
  s = strip(s);
  if (s.empty()) return false;
  {
    // fast path (Clinger): [+-] digits [. digits] with at most 15 significant digits is
    // mantissa / 10^k with both exactly representable, so one division rounds correctly --
    // the same double Python's float() returns.  Anything else takes the general path below.
    static const double kPow10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* q = s.b;
    bool neg = false;
    if (*q == '+' || *q == '-') neg = (*q++ == '-');
    unsigned long long mant = 0;
    int nd = 0, frac = 0;
    bool ok = q < s.e, dot = false, any = false;
    for (; q < s.e; ++q) {
      const char ch = *q;
      if (ch >= '0' && ch <= '9') {
        mant = mant * 10 + (unsigned)(ch - '0');
        any = true;
        if (mant != 0 || nd != 0) ++nd;  // leading zeros are not significant
        if (dot) ++frac;
        if (nd > 15 || frac > 22) {
          ok = false;
          break;
        }
      } else if (ch == '.' && !dot) {
        dot = true;
      } else {
        ok = false;
        break;
      }
    }
    if (ok && any) {
      const double v = (double)mant / kPow10[frac];
      *out = neg ? -v : v;
      return true;
    }
  }
  char buf[96];
  size_t n = 0;
  const char* p = s.b;
  if (*p == '+' || *p == '-') buf[n++] = *p++;
  if (p >= s.e) return false;
  if (ieq(p, s.e, "inf") || ieq(p, s.e, "infinity")) {
    *out = (buf[0] == '-' && n == 1) ? -std::numeric_limits<double>::infinity()
                                     : std::numeric_limits<double>::infinity();
    return true;
  }
  if (ieq(p, s.e, "nan")) {
    *out = std::numeric_limits<double>::quiet_NaN();
    return true;
  }
  auto digits = [&](bool* any) -> bool {
    bool prev_us = false;
    *any = false;
    while (p < s.e && (is_digit(*p) || *p == '_')) {
      if (*p == '_') {
        if (!*any || prev_us || p + 1 >= s.e || !is_digit(p[1])) return false;
        prev_us = true;
      } else {
        if (n >= sizeof(buf) - 2) return false;
        buf[n++] = *p;
        *any = true;
        prev_us = false;
      }
      ++p;
    }
    return true;
  };
  bool int_any = false, frac_any = false;
  if (!digits(&int_any)) return false;
  if (p < s.e && *p == '.') {
    if (n >= sizeof(buf) - 2) return false;
    buf[n++] = *p++;
    if (!digits(&frac_any)) return false;
  }
  if (!int_any && !frac_any) return false;
  if (p < s.e && (*p == 'e' || *p == 'E')) {
    if (n >= sizeof(buf) - 4) return false;
    buf[n++] = 'e';
    ++p;
    if (p < s.e && (*p == '+' || *p == '-')) buf[n++] = *p++;
    bool exp_any = false;
    if (!digits(&exp_any) || !exp_any) return false;
  }
  if (p != s.e) return false;
  buf[n] = 0;
  char* endp = nullptr;
  errno = 0;
  *out = strtod(buf, &endp);
  return endp == buf + n;
}

}  // namespace
}  // namespace xdfm

using xdfm::Span;

// what parsing a run of lines produces, in file order
struct Staging {
  std::vector<float> labels;
  std::vector<int64_t> tok_sample, tok_feat, dnn_row, dnn_col;
  std::vector<float> tok_val;
  std::vector<int64_t> imp_off;  // start of each sample's impression id in imp_chars
  std::vector<uint8_t> imp_present;
  std::string imp_chars;
  int64_t dnn_width = 0;
  // per-sample scratch: running count per field (reference: dnn_feat_dic, :178-185)
  std::vector<int64_t> cnt, stamp;                        // fields inside [0, F)
  std::vector<std::pair<int64_t, int64_t>> field_count;   // any other field index
  int err_code = 0;
  std::string err;

  void clear() {
    labels.clear();
    tok_sample.clear();
    tok_feat.clear();
    dnn_row.clear();
    dnn_col.clear();
    tok_val.clear();
    imp_off.clear();
    imp_present.clear();
    imp_chars.clear();
    dnn_width = 0;
    err_code = 0;
    err.clear();
  }
  int fail(int code, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err_code = code;
    err = buf;
    return code;
  }
};

struct xdfm_ffm_reader {
  int fd = -1;
  const char* data = nullptr;
  size_t size = 0;
  size_t pos = 0;
  int64_t line_no = 0;  // lines consumed so far
  int64_t F = 0, N = 0;
  char col_sep = ' ', id_sep = '%';
  // staged batch, file order
  int64_t first_line = 0;
  Staging st;                    // the batch
  std::vector<Staging> workers;  // per-thread pieces of it
  std::vector<xdfm::Span> lines;
  std::vector<uint32_t> order;  // dnn view permutation (stable by (row, col))
  int threads = 1;
};

extern "C" int xdfm_ffm_open(const char* path, int64_t field_count, int64_t feature_count,
                             char col_sep, char id_sep, xdfm_ffm_reader** out) {
  XDFM_REQUIRE(path && out, "xdfm_ffm_open: null argument");
  XDFM_REQUIRE(field_count > 0, "xdfm_ffm_open: FIELD_COUNT must be positive");
  XDFM_REQUIRE(col_sep != ':' && id_sep != ':' && col_sep != id_sep,
               "xdfm_ffm_open: separators must differ from ':' and from each other");
  int fd = open(path, O_RDONLY);
  if (fd < 0) return xdfm::fail(XDFM_E_IO, "cannot open %s: %s", path, strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return xdfm::fail(XDFM_E_IO, "cannot stat %s: %s", path, strerror(errno));
  }
  const char* data = nullptr;
  if (st.st_size > 0) {
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      return xdfm::fail(XDFM_E_IO, "cannot mmap %s: %s", path, strerror(errno));
    }
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    data = (const char*)m;
  }
  auto* r = new xdfm_ffm_reader();
  r->fd = fd;
  r->data = data;
  r->size = (size_t)st.st_size;
  r->F = field_count;
  r->N = feature_count;
  r->col_sep = col_sep;
  r->id_sep = id_sep;
  {
    // parsing threads: XDFM_PARSE_THREADS, else the hardware's, at most 16
    const char* e = getenv("XDFM_PARSE_THREADS");
    int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    r->threads = t < 1 ? 1 : (t > 16 ? 16 : t);
  }
  *out = r;
  return XDFM_OK;
}

extern "C" int xdfm_ffm_close(xdfm_ffm_reader* r) {
  if (!r) return XDFM_OK;
  if (r->data) munmap((void*)r->data, r->size);
  if (r->fd >= 0) close(r->fd);
  delete r;
  return XDFM_OK;
}

namespace {

// one sample (1-based line number ln, index `sample` in the batch); appends to st
int parse_line(const xdfm_ffm_reader* r, Staging& st, Span line, int64_t sample, int64_t ln) {
  using namespace xdfm;
  line = strip(line);
  // words = line.split(ID_spliter); impression id only when len(words) == 2
  const char* first_sep = nullptr;
  int seps = 0;
  for (const char* p = line.b; p < line.e; ++p)
    if (*p == r->id_sep) {
      if (!first_sep) first_sep = p;
      ++seps;
    }
  Span head = line;
  if (first_sep) head.e = first_sep;
  st.imp_off.push_back((int64_t)st.imp_chars.size());
  if (seps == 1) {
    Span imp = strip(Span{first_sep + 1, line.e});
    st.imp_chars.append(imp.b, imp.e);
    st.imp_present.push_back(1);
  } else {
    st.imp_present.push_back(0);
  }
  head = strip(head);

  // cols = head.split(col_spliter)
  const char* p = head.b;
  auto next_word = [&](Span* w) -> bool {
    if (p > head.e) return false;
    const char* q = p;
    while (q < head.e && *q != r->col_sep) ++q;
    *w = Span{p, q};
    p = q + 1;  // past the separator (or past the end -> terminates)
    return true;
  };
  Span w;
  next_word(&w);  // str.split always yields at least one (possibly empty) word
  double label;
  if (!parse_py_float(w, &label))
    return st.fail(XDFM_E_PARSE, "line %lld: could not convert label '%.*s' to float", (long long)ln,
                   (int)std::min<ptrdiff_t>(w.e - w.b, 40), w.b);
  st.labels.push_back((float)label);

  st.field_count.clear();
  if ((int64_t)st.cnt.size() != r->F) {
    st.cnt.assign((size_t)r->F, 0);
    st.stamp.assign((size_t)r->F, -1);
  }
  const int64_t epoch = ln;  // unique per line
  while (next_word(&w)) {
    if (strip(w).empty()) continue;
    // tokens = word.split(':'); tokens[0..2] used, extra ones ignored
    const char* c1 = (const char*)memchr(w.b, ':', (size_t)(w.e - w.b));
    const char* c2 = c1 ? (const char*)memchr(c1 + 1, ':', (size_t)(w.e - c1 - 1)) : nullptr;
    if (!c1 || !c2)
      return st.fail(XDFM_E_PARSE, "line %lld: token '%.*s' is not field:feature:value", (long long)ln,
                     (int)std::min<ptrdiff_t>(w.e - w.b, 40), w.b);
    const char* c3 = (const char*)memchr(c2 + 1, ':', (size_t)(w.e - c2 - 1));
    int64_t field, feat;
    double val;
    if (!parse_py_int(Span{w.b, c1}, &field) || !parse_py_int(Span{c1 + 1, c2}, &feat) ||
        !parse_py_float(Span{c2 + 1, c3 ? c3 : w.e}, &val))
      return st.fail(XDFM_E_PARSE, "line %lld: invalid literal in token '%.*s'", (long long)ln,
                     (int)std::min<ptrdiff_t>(w.e - w.b, 40), w.b);
    field -= 1;
    feat -= 1;
    // running count of this field inside the sample (column of the dnn view)
    int64_t col = 0;
    if (field >= 0 && field < r->F) {
      if (st.stamp[(size_t)field] == epoch) {
        col = ++st.cnt[(size_t)field];
      } else {
        st.stamp[(size_t)field] = epoch;
        st.cnt[(size_t)field] = 0;
      }
    } else {
      bool seen = false;
      for (auto& fc : st.field_count)
        if (fc.first == field) {
          col = ++fc.second;
          seen = true;
          break;
        }
      if (!seen) st.field_count.emplace_back(field, 0);
    }
    st.tok_sample.push_back(sample);
    st.tok_feat.push_back(feat);
    st.tok_val.push_back((float)val);
    st.dnn_row.push_back(sample * r->F + field);
    st.dnn_col.push_back(col);
    if (col + 1 > st.dnn_width) st.dnn_width = col + 1;
  }
  return XDFM_OK;
}

template <typename T>
void append(std::vector<T>& dst, const std::vector<T>& src) {
  dst.insert(dst.end(), src.begin(), src.end());
}

}  // namespace

extern "C" int xdfm_ffm_next(xdfm_ffm_reader* r, int64_t batch_size, xdfm_ffm_batch_info* info) {
  XDFM_REQUIRE(r && info, "xdfm_ffm_next: null argument");
  XDFM_REQUIRE(batch_size > 0, "xdfm_ffm_next: batch_size must be positive");
  Staging& st = r->st;
  st.clear();
  r->first_line = r->line_no + 1;

  // 1. the next batch_size lines (universal newlines)
  r->lines.clear();
  const char* end = r->data + r->size;
  while ((int64_t)r->lines.size() < batch_size && r->pos < r->size) {
    const char* b = r->data + r->pos;
    const char* e = b;
    while (e < end && *e != '\n' && *e != '\r') ++e;
    size_t next = (size_t)(e - r->data);
    if (e < end) next += (*e == '\r' && e + 1 < end && e[1] == '\n') ? 2 : 1;
    r->pos = next;
    r->lines.push_back(Span{b, e});
  }
  const int64_t n = (int64_t)r->lines.size();
  if (n == 0) return xdfm::fail(XDFM_E_EOF, "end of file");

  // 2. parse: contiguous runs of lines on up to r->threads threads, each into its own staging
  int T = r->threads;
  if (T > n / 256) T = (int)(n / 256);
  if (T < 1) T = 1;
  const int64_t base_line = r->line_no;
  if (T == 1) {
    for (int64_t i = 0; i < n; ++i)
      if (parse_line(r, st, r->lines[(size_t)i], i, base_line + i + 1) != XDFM_OK) break;
  } else {
    if ((int)r->workers.size() < T) r->workers.resize((size_t)T);
    std::vector<std::thread> pool;
    const int64_t per = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) {
      pool.emplace_back([r, t, per, n, base_line]() {
        Staging& w = r->workers[(size_t)t];
        w.clear();
        const int64_t lo = t * per, hi = std::min<int64_t>(n, lo + per);
        for (int64_t i = lo; i < hi; ++i)
          if (parse_line(r, w, r->lines[(size_t)i], i, base_line + i + 1) != XDFM_OK) break;
      });
    }
    for (auto& th : pool) th.join();
    // 3. stitch the pieces together in file order; the first failing piece decides the error
    for (int t = 0; t < T; ++t) {
      Staging& w = r->workers[(size_t)t];
      const int64_t imp_base = (int64_t)st.imp_chars.size();
      append(st.labels, w.labels);
      append(st.tok_sample, w.tok_sample);
      append(st.tok_feat, w.tok_feat);
      append(st.dnn_row, w.dnn_row);
      append(st.dnn_col, w.dnn_col);
      append(st.tok_val, w.tok_val);
      append(st.imp_present, w.imp_present);
      for (int64_t o : w.imp_off) st.imp_off.push_back(o + imp_base);
      st.imp_chars += w.imp_chars;
      if (w.dnn_width > st.dnn_width) st.dnn_width = w.dnn_width;
      if (w.err_code) {
        st.err_code = w.err_code;
        st.err = w.err;
        break;
      }
    }
  }
  r->line_no += n;
  if (st.err_code) return xdfm::fail(st.err_code, "%s", st.err.c_str());
  st.imp_off.push_back((int64_t)st.imp_chars.size());

  // dnn view order: stable by (row, col)   (iterator.py:198-201)
  const size_t nnz = r->st.dnn_row.size();
  r->order.resize(nnz);
  std::iota(r->order.begin(), r->order.end(), 0u);
  bool sorted = true;
  for (size_t k = 1; k < nnz && sorted; ++k)
    sorted = std::make_pair(r->st.dnn_row[k - 1], r->st.dnn_col[k - 1]) <=
             std::make_pair(r->st.dnn_row[k], r->st.dnn_col[k]);
  if (!sorted)
    std::stable_sort(r->order.begin(), r->order.end(), [r](uint32_t a, uint32_t b) {
      if (r->st.dnn_row[a] != r->st.dnn_row[b]) return r->st.dnn_row[a] < r->st.dnn_row[b];
      return r->st.dnn_col[a] < r->st.dnn_col[b];
    });

  info->n_rows = n;
  info->nnz = (int64_t)nnz;
  info->dnn_width = r->st.dnn_width;
  info->imp_bytes = (int64_t)r->st.imp_chars.size();
  info->first_line = r->first_line;
  return XDFM_OK;
}

// fn(lo, hi) over [0, n) on up to T threads (the arrays of a batch are tens of megabytes: one
// core's store bandwidth is what the fill is bound by)
template <typename Fn>
static void par_ranges(size_t n, int T, Fn fn) {
  if (T > 8) T = 8;
  if (T <= 1 || n < ((size_t)1 << 16)) {
    fn((size_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  const size_t per = (n + (size_t)T - 1) / (size_t)T;
  for (int t = 0; t < T; ++t) {
    const size_t lo = (size_t)t * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    pool.emplace_back([fn, lo, hi]() { fn(lo, hi); });
  }
  for (auto& th : pool) th.join();
}

extern "C" int xdfm_ffm_fill(xdfm_ffm_reader* r, float* labels, int64_t* fm_feat_indices,
                             float* fm_feat_values, int64_t* dnn_feat_indices,
                             int64_t* dnn_feat_values, float* dnn_feat_weights,
                             int32_t* row_ptr, int64_t* imp_offsets, char* imp_chars,
                             uint8_t* imp_present) {
  XDFM_REQUIRE(r, "xdfm_ffm_fill: null reader");
  const size_t n = r->st.labels.size();
  const size_t nnz = r->st.dnn_row.size();
  if (labels) memcpy(labels, r->st.labels.data(), n * sizeof(float));
  par_ranges(nnz, r->threads, [=](size_t lo, size_t hi) {
    if (fm_feat_indices)
      for (size_t k = lo; k < hi; ++k) {
        fm_feat_indices[2 * k] = r->st.tok_sample[k];
        fm_feat_indices[2 * k + 1] = r->st.tok_feat[k];
      }
    if (fm_feat_values) memcpy(fm_feat_values + lo, r->st.tok_val.data() + lo, (hi - lo) * sizeof(float));
    for (size_t k = lo; k < hi; ++k) {
      const uint32_t s = r->order[k];
      if (dnn_feat_indices) {
        dnn_feat_indices[2 * k] = r->st.dnn_row[s];
        dnn_feat_indices[2 * k + 1] = r->st.dnn_col[s];
      }
      if (dnn_feat_values) dnn_feat_values[k] = r->st.tok_feat[s];
      if (dnn_feat_weights) dnn_feat_weights[k] = r->st.tok_val[s];
    }
  });
  if (row_ptr) {
    const int64_t rows = (int64_t)n * r->F;
    XDFM_REQUIRE(rows + 1 < (int64_t)1 << 31 && nnz < ((size_t)1 << 31),
                 "xdfm_ffm_fill: batch too large for int32 CSR offsets");
    memset(row_ptr, 0, (size_t)(rows + 1) * sizeof(int32_t));
    for (size_t k = 0; k < nnz; ++k) {
      const int64_t row = r->st.dnn_row[k];
      if (row < 0 || row >= rows)
        return xdfm::fail(XDFM_E_INVALID,
                          "batch starting at line %lld: field index outside [1, FIELD_COUNT=%lld]",
                          (long long)r->first_line, (long long)r->F);
      row_ptr[row + 1] += 1;
    }
    for (int64_t i = 0; i < rows; ++i) row_ptr[i + 1] += row_ptr[i];
  }
  if (imp_offsets) memcpy(imp_offsets, r->st.imp_off.data(), (n + 1) * sizeof(int64_t));
  if (imp_chars && !r->st.imp_chars.empty()) memcpy(imp_chars, r->st.imp_chars.data(), r->st.imp_chars.size());
  if (imp_present) memcpy(imp_present, r->st.imp_present.data(), n);
  return XDFM_OK;
}
