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
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
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
  bool in_order = true;   // tokens already in (row, col) order (the dnn view needs no sort)
  bool in_range = true;   // every field inside [0, F)
  int64_t lo = 0, hi = 0; // the samples of the batch this piece holds
  // per-sample scratch: running count per field (reference: dnn_feat_dic, :178-185)
  std::vector<int64_t> cnt, stamp;                        // fields inside [0, F)
  std::vector<int64_t> f_feat, f_row, f_col;              // one line's tokens (parse_line_fast)
  std::vector<float> f_val;
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
    in_order = in_range = true;
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

namespace {

// Threads that live as long as the reader: a batch is parsed and its arrays are filled by the same
// few threads, so neither step pays for starting them (a batch of 8192 lines lasts about a
// millisecond).  run(n, fn) calls fn(0..n-1), one index per thread, and returns when all are done.
class Pool {
 public:
  explicit Pool(int threads) {
    for (int t = 1; t < threads; ++t) th_.emplace_back([this, t]() { loop(t); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int size() const { return (int)th_.size() + 1; }
  void run(int n, const std::function<void(int)>& fn) {
    if (n <= 1) {
      if (n == 1) fn(0);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      n_ = n;
      pending_ = n - 1;
      ++gen_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this]() { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (t < n_) fn = fn_;
      }
      if (!fn) continue;
      (*fn)(t);
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

}  // namespace

struct xdfm_ffm_reader {
  int fd = -1;
  const char* data = nullptr;
  size_t size = 0;
  size_t pos = 0;
  int64_t line_no = 0;  // lines consumed so far
  int64_t F = 0, N = 0;
  char col_sep = ' ', id_sep = '%';
  // staged batch, file order: `pieces` runs of lines, one per parsing thread
  int64_t first_line = 0;
  int64_t n_rows = 0, nnz = 0;
  std::vector<Staging> workers;
  int pieces = 0;
  std::vector<int64_t> tok_base, imp_base;  // of each piece, [pieces + 1]
  bool in_order = true, in_range = true;
  Staging st;                   // the pieces joined: only when the dnn view has to be sorted
  std::vector<uint32_t> order;  // dnn view permutation (stable by (row, col)) of `st`
  std::vector<xdfm::Span> lines;
  int threads = 1;
  std::unique_ptr<Pool> pool;
  // Line index: positions of the '\n's of the stretch [.., idx_pos) that no batch has consumed yet, found by
  // all threads at once, a stretch of the file at a time (finding 8192 line ends with one thread took as long
  // as parsing them with sixteen).  Only for files without any '\r' (checked once when the file is opened):
  // universal newlines also end a line at a bare '\r', which the one-thread scan below still serves.
  bool plain_newlines = false;
  std::vector<uint64_t> nl;                   // ascending
  size_t nl_head = 0;                         // first not yet consumed
  size_t idx_pos = 0;                         // first byte not yet indexed
  std::vector<std::vector<uint64_t>> nl_part;  // per thread
  // the run of indexed lines that completes the current batch: lines[run_at .. run_at + run_take) end at
  // nl[run_nl ..], the first one starts at run_pos (their spans are written by the parsing threads)
  size_t run_at = 0, run_take = 0, run_nl = 0, run_pos = 0;
  // XDFM_FFM_TIMING=1: seconds spent finding lines / parsing / filling, printed when the reader is closed
  bool timing = false;
  double t_lines = 0, t_parse = 0, t_fill = 0;
  int64_t batches = 0;
};

static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

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
  r->pool.reset(new Pool(r->threads));
  r->timing = getenv("XDFM_FFM_TIMING") != nullptr;
  if (r->size > 0) {
    // any '\r' at all?  (each thread looks through its part of the file)
    const int T = r->threads;
    std::vector<uint8_t> found((size_t)T, 0);
    const size_t per = (r->size + (size_t)T - 1) / (size_t)T;
    r->pool->run(T, [r, per, &found](int t) {
      const size_t lo = std::min(r->size, (size_t)t * per), hi = std::min(r->size, lo + per);
      if (hi > lo && memchr(r->data + lo, '\r', hi - lo)) found[(size_t)t] = 1;
    });
    r->plain_newlines = true;
    for (uint8_t f : found) r->plain_newlines = r->plain_newlines && !f;
  }
  *out = r;
  return XDFM_OK;
}

extern "C" int xdfm_ffm_close(xdfm_ffm_reader* r) {
  if (!r) return XDFM_OK;
  if (r->timing && r->batches > 0)
    fprintf(stderr, "xdfm_ffm: %lld batches on %d threads: lines %.3f ms, parse %.3f ms, fill %.3f ms per batch\n",
            (long long)r->batches, r->threads, 1e3 * r->t_lines / r->batches, 1e3 * r->t_parse / r->batches,
            1e3 * r->t_fill / r->batches);
  r->pool.reset();
  if (r->data) munmap((void*)r->data, r->size);
  if (r->fd >= 0) close(r->fd);
  delete r;
  return XDFM_OK;
}

namespace {

// The canonical line -- `label f:i:v f:i:v ...` with single separators, no impression id, plain decimal
// literals ([+-]digits[.digits], at most 15 significant digits: the case in which mantissa / 10^k is the
// correctly rounded double, as in parse_py_float) and every field inside [1, FIELD_COUNT] -- in ONE pass over
// its bytes: no per-token memchr, no spans, the tokens collected in the piece's scratch and appended in bulk.
// Returns false BEFORE anything is appended when the line is anything else; parse_line then takes it through
// the general path (same results, same messages).
inline bool fast_number(const char*& p, const char* e, char stop, double* out) {
  static const double kPow10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char* q = p;
  bool neg = false;
  if (q < e && (*q == '+' || *q == '-')) neg = (*q++ == '-');
  unsigned long long mant = 0;
  int nd = 0, frac = 0;
  bool any = false;
  while (q < e && (unsigned)(*q - '0') <= 9u) {
    mant = mant * 10 + (unsigned)(*q - '0');
    if (mant != 0) ++nd;
    any = true;
    ++q;
  }
  if (q < e && *q == '.') {
    ++q;
    while (q < e && (unsigned)(*q - '0') <= 9u) {
      mant = mant * 10 + (unsigned)(*q - '0');
      if (mant != 0) ++nd;
      any = true;
      ++frac;
      ++q;
    }
  }
  if (!any || nd > 15 || frac > 22 || (q < e && *q != stop)) return false;
  const double v = (double)mant / kPow10[frac];
  *out = neg ? -v : v;
  p = q;
  return true;
}

inline bool fast_index(const char*& p, const char* e, int64_t* out) {
  const char* q = p;
  unsigned long long acc = 0;
  while (q < e && (unsigned)(*q - '0') <= 9u) {
    acc = acc * 10 + (unsigned)(*q - '0');
    ++q;
  }
  if (q == p || q - p > 18 || q >= e || *q != ':') return false;
  *out = (int64_t)acc;
  p = q + 1;
  return true;
}

bool parse_line_fast(const xdfm_ffm_reader* r, Staging& st, Span line, int64_t sample, int64_t ln) {
  using namespace xdfm;
  if (line.empty() || r->col_sep == ':' || (unsigned)(r->col_sep - '0') <= 9u || r->col_sep == '.' ||
      r->col_sep == '+' || r->col_sep == '-')
    return false;
  if (memchr(line.b, r->id_sep, (size_t)(line.e - line.b))) return false;
  const char* p = line.b;
  const char* const e = line.e;
  double label;
  if (!fast_number(p, e, r->col_sep, &label)) return false;
  if ((int64_t)st.cnt.size() != r->F) {
    st.cnt.assign((size_t)r->F, 0);
    st.stamp.assign((size_t)r->F, -1);
  }
  st.f_feat.clear();
  st.f_row.clear();
  st.f_col.clear();
  st.f_val.clear();
  const int64_t epoch = ln;
  int64_t prev_row = st.dnn_row.empty() ? -1 : st.dnn_row.back();
  int64_t prev_col = st.dnn_col.empty() ? -1 : st.dnn_col.back();
  bool in_order = true;
  int64_t width = 0;
  while (p < e) {
    ++p;  // the separator fast_number stopped at
    int64_t field, feat;
    double val;
    if (!fast_index(p, e, &field) || !fast_index(p, e, &feat) || !fast_number(p, e, r->col_sep, &val)) return false;
    field -= 1;
    if (field < 0 || field >= r->F) return false;
    int64_t col = 0;
    if (st.stamp[(size_t)field] == epoch) {
      col = ++st.cnt[(size_t)field];
    } else {
      st.stamp[(size_t)field] = epoch;
      st.cnt[(size_t)field] = 0;
    }
    const int64_t row = sample * r->F + field;
    if (row < prev_row || (row == prev_row && col < prev_col)) in_order = false;
    prev_row = row;
    prev_col = col;
    st.f_feat.push_back(feat - 1);
    st.f_row.push_back(row);
    st.f_col.push_back(col);
    st.f_val.push_back((float)val);
    if (col + 1 > width) width = col + 1;
  }
  // the line is canonical: commit
  st.imp_off.push_back((int64_t)st.imp_chars.size());
  st.imp_present.push_back(0);
  st.labels.push_back((float)label);
  const size_t k = st.f_feat.size();
  st.tok_sample.insert(st.tok_sample.end(), k, sample);
  st.tok_feat.insert(st.tok_feat.end(), st.f_feat.begin(), st.f_feat.end());
  st.tok_val.insert(st.tok_val.end(), st.f_val.begin(), st.f_val.end());
  st.dnn_row.insert(st.dnn_row.end(), st.f_row.begin(), st.f_row.end());
  st.dnn_col.insert(st.dnn_col.end(), st.f_col.begin(), st.f_col.end());
  if (!in_order) st.in_order = false;
  if (width > st.dnn_width) st.dnn_width = width;
  return true;
}

// one sample (1-based line number ln, index `sample` in the batch); appends to st
int parse_line(const xdfm_ffm_reader* r, Staging& st, Span line, int64_t sample, int64_t ln) {
  using namespace xdfm;
  line = strip(line);
  if (parse_line_fast(r, st, line, sample, ln)) return XDFM_OK;
  // (a failed fast attempt may have stamped fields with this line's epoch: start the counts afresh)
  std::fill(st.stamp.begin(), st.stamp.end(), -1);
  // words = line.split(ID_spliter); impression id only when len(words) == 2
  const char* first_sep = line.empty() ? nullptr : (const char*)memchr(line.b, r->id_sep, (size_t)(line.e - line.b));
  int seps = first_sep ? 1 : 0;
  if (first_sep && first_sep + 1 < line.e && memchr(first_sep + 1, r->id_sep, (size_t)(line.e - first_sep - 1)))
    seps = 2;  // two or more: no impression id (:101-103)
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
    const char* q = (const char*)memchr(p, r->col_sep, (size_t)(head.e - p));
    if (!q) q = head.e;
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
  int64_t prev_row = st.dnn_row.empty() ? -1 : st.dnn_row.back();
  int64_t prev_col = st.dnn_col.empty() ? -1 : st.dnn_col.back();
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
      st.in_range = false;
      bool seen = false;
      for (auto& fc : st.field_count)
        if (fc.first == field) {
          col = ++fc.second;
          seen = true;
          break;
        }
      if (!seen) st.field_count.emplace_back(field, 0);
    }
    const int64_t row = sample * r->F + field;
    if (row < prev_row || (row == prev_row && col < prev_col)) st.in_order = false;
    prev_row = row;
    prev_col = col;
    st.tok_sample.push_back(sample);
    st.tok_feat.push_back(feat);
    st.tok_val.push_back((float)val);
    st.dnn_row.push_back(row);
    st.dnn_col.push_back(col);
    if (col + 1 > st.dnn_width) st.dnn_width = col + 1;
  }
  return XDFM_OK;
}

template <typename T>
void append(std::vector<T>& dst, const std::vector<T>& src) {
  dst.insert(dst.end(), src.begin(), src.end());
}

// the pieces joined into r->st and the stable (row, col) order of the dnn view: only for batches
// whose tokens do not already come in that order (fields out of order inside a line)
void join_and_sort(xdfm_ffm_reader* r) {
  Staging& st = r->st;
  st.clear();
  for (int t = 0; t < r->pieces; ++t) {
    const Staging& w = r->workers[(size_t)t];
    append(st.tok_sample, w.tok_sample);
    append(st.tok_feat, w.tok_feat);
    append(st.dnn_row, w.dnn_row);
    append(st.dnn_col, w.dnn_col);
    append(st.tok_val, w.tok_val);
  }
  const size_t nnz = st.dnn_row.size();
  r->order.resize(nnz);
  std::iota(r->order.begin(), r->order.end(), 0u);
  std::stable_sort(r->order.begin(), r->order.end(), [&st](uint32_t a, uint32_t b) {
    if (st.dnn_row[a] != st.dnn_row[b]) return st.dnn_row[a] < st.dnn_row[b];
    return st.dnn_col[a] < st.dnn_col[b];
  });
}

}  // namespace

extern "C" int xdfm_ffm_next(xdfm_ffm_reader* r, int64_t batch_size, xdfm_ffm_batch_info* info) {
  XDFM_REQUIRE(r && info, "xdfm_ffm_next: null argument");
  XDFM_REQUIRE(batch_size > 0, "xdfm_ffm_next: batch_size must be positive");
  r->first_line = r->line_no + 1;

  // 1. the next batch_size lines (universal newlines)
  const double t0 = r->timing ? now_s() : 0;
  r->lines.clear();
  r->run_take = 0;
  const char* end = r->data + r->size;
  if (r->plain_newlines) {
    while ((int64_t)r->lines.size() < batch_size && r->pos < r->size) {
      if (r->nl_head == r->nl.size() && r->idx_pos < r->size) {
        // index the next stretch: what the lines still wanted should weigh (by the lines read so far, 512
        // bytes each before there are any) and a quarter more, 4 MiB per thread at most
        r->nl.clear();
        r->nl_head = 0;
        const int T = r->threads;
        const size_t want = (size_t)(batch_size - (int64_t)r->lines.size());
        const size_t avg = r->line_no > 0 ? r->pos / (size_t)r->line_no + 16 : 512;
        size_t len = std::max<size_t>(want * avg + want * avg / 4, (size_t)1 << 16);
        len = std::min(len, (size_t)T << 22);
        const size_t lo0 = r->idx_pos, hi0 = std::min(r->size, lo0 + len);
        if ((int)r->nl_part.size() < T) r->nl_part.resize((size_t)T);
        const size_t per = (hi0 - lo0 + (size_t)T - 1) / (size_t)T;
        r->pool->run(T, [r, lo0, hi0, per](int t) {
          std::vector<uint64_t>& out = r->nl_part[(size_t)t];
          out.clear();
          const size_t lo = std::min(hi0, lo0 + (size_t)t * per), hi = std::min(hi0, lo + per);
          const char* q = r->data + lo;
          const char* const qe = r->data + hi;
          while (q < qe) {
            const char* e = (const char*)memchr(q, '\n', (size_t)(qe - q));
            if (!e) break;
            out.push_back((uint64_t)(e - r->data));
            q = e + 1;
          }
        });
        for (int t = 0; t < T; ++t) r->nl.insert(r->nl.end(), r->nl_part[(size_t)t].begin(), r->nl_part[(size_t)t].end());
        r->idx_pos = hi0;
      }
      const char* b = r->data + r->pos;
      if (r->nl_head < r->nl.size()) {
        // a run of indexed lines at once.  The run that completes the batch gets its spans from the parsing
        // threads (each writes the spans of the lines it parses: 8192 spans on one thread took a seventh of a
        // batch's time); a run that ends with the index is written here, before the index is refilled.
        const size_t have = r->nl.size() - r->nl_head;
        const size_t want = (size_t)(batch_size - (int64_t)r->lines.size());
        const size_t take = have < want ? have : want;
        const size_t at = r->lines.size();
        r->lines.resize(at + take);
        if (take == want) {
          r->run_at = at;
          r->run_take = take;
          r->run_nl = r->nl_head;
          r->run_pos = r->pos;
        } else {
          size_t from = r->pos;
          for (size_t k = 0; k < take; ++k) {
            const size_t e = (size_t)r->nl[r->nl_head + k];
            r->lines[at + k] = Span{r->data + from, r->data + e};
            from = e + 1;
          }
        }
        r->pos = (size_t)r->nl[r->nl_head + take - 1] + 1;
        r->nl_head += take;
      } else if (r->idx_pos >= r->size) {
        r->pos = r->size;  // the last line has no newline
        r->lines.push_back(Span{b, end});
      }
      // (else: a stretch without a newline was indexed; the loop indexes the next one)
    }
  } else
  while ((int64_t)r->lines.size() < batch_size && r->pos < r->size) {
    const char* b = r->data + r->pos;
    // the line ends at the first '\n' or '\r': two library scans instead of a byte loop (this
    // part runs on one thread)
    const char* e = (const char*)memchr(b, '\n', (size_t)(end - b));
    if (!e) e = end;
    if (const char* cr = (const char*)memchr(b, '\r', (size_t)(e - b))) e = cr;
    size_t next = (size_t)(e - r->data);
    if (e < end) next += (*e == '\r' && e + 1 < end && e[1] == '\n') ? 2 : 1;
    r->pos = next;
    r->lines.push_back(Span{b, e});
  }
  const int64_t n = (int64_t)r->lines.size();
  if (n == 0) return xdfm::fail(XDFM_E_EOF, "end of file");

  // 2. parse: contiguous runs of lines on up to r->threads threads, each into its own staging
  const double t1 = r->timing ? now_s() : 0;
  int T = r->threads;
  if (T > n / 256) T = (int)(n / 256);
  if (T < 1) T = 1;
  if ((int)r->workers.size() < T) r->workers.resize((size_t)T);
  const int64_t base_line = r->line_no;
  const int64_t per = (n + T - 1) / T;
  r->pool->run(T, [r, per, n, base_line](int t) {
    Staging& w = r->workers[(size_t)t];
    w.clear();
    w.lo = std::min<int64_t>(n, t * per);
    w.hi = std::min<int64_t>(n, w.lo + per);
    if (r->run_take) {  // the spans of this thread's lines out of the newline index
      const int64_t a = std::max<int64_t>(w.lo, (int64_t)r->run_at);
      const int64_t b = std::min<int64_t>(w.hi, (int64_t)(r->run_at + r->run_take));
      for (int64_t i = a; i < b; ++i) {
        const size_t k = (size_t)i - r->run_at;
        const size_t from = k == 0 ? r->run_pos : (size_t)r->nl[r->run_nl + k - 1] + 1;
        r->lines[(size_t)i] = Span{r->data + from, r->data + (size_t)r->nl[r->run_nl + k]};
      }
    }
    for (int64_t i = w.lo; i < w.hi; ++i)
      if (parse_line(r, w, r->lines[(size_t)i], i, base_line + i + 1) != XDFM_OK) break;
  });
  r->line_no += n;
  r->pieces = T;
  if (r->timing) {
    r->t_lines += t1 - t0;
    r->t_parse += now_s() - t1;
    r->batches += 1;
  }

  // 3. the pieces stay where they are; the first failing one (file order) decides the error
  r->tok_base.assign((size_t)T + 1, 0);
  r->imp_base.assign((size_t)T + 1, 0);
  r->in_order = r->in_range = true;
  int64_t width = 0, prev_row = -1, prev_col = -1;
  for (int t = 0; t < T; ++t) {
    Staging& w = r->workers[(size_t)t];
    if (w.err_code) return xdfm::fail(w.err_code, "%s", w.err.c_str());
    r->tok_base[(size_t)t + 1] = r->tok_base[(size_t)t] + (int64_t)w.dnn_row.size();
    r->imp_base[(size_t)t + 1] = r->imp_base[(size_t)t] + (int64_t)w.imp_chars.size();
    width = std::max(width, w.dnn_width);
    r->in_range = r->in_range && w.in_range;
    r->in_order = r->in_order && w.in_order;
    if (!w.dnn_row.empty()) {
      if (w.dnn_row.front() < prev_row || (w.dnn_row.front() == prev_row && w.dnn_col.front() < prev_col))
        r->in_order = false;
      prev_row = w.dnn_row.back();
      prev_col = w.dnn_col.back();
    }
  }
  r->n_rows = n;
  r->nnz = r->tok_base[(size_t)T];
  if (!r->in_order) join_and_sort(r);  // dnn view order: stable by (row, col)   (iterator.py:198-201)

  info->n_rows = n;
  info->nnz = r->nnz;
  info->dnn_width = width;
  info->imp_bytes = r->imp_base[(size_t)T];
  info->first_line = r->first_line;
  return XDFM_OK;
}

extern "C" int xdfm_ffm_fill(xdfm_ffm_reader* r, float* labels, int64_t* fm_feat_indices,
                             float* fm_feat_values, int64_t* dnn_feat_indices,
                             int64_t* dnn_feat_values, float* dnn_feat_weights,
                             int32_t* row_ptr, int64_t* imp_offsets, char* imp_chars,
                             uint8_t* imp_present) {
  XDFM_REQUIRE(r, "xdfm_ffm_fill: null reader");
  const int64_t n = r->n_rows, nnz = r->nnz, F = r->F;
  const int64_t rows = n * F;
  if (row_ptr) {
    XDFM_REQUIRE(rows + 1 < (int64_t)1 << 31 && nnz < ((int64_t)1 << 31),
                 "xdfm_ffm_fill: batch too large for int32 CSR offsets");
    if (!r->in_range)
      return xdfm::fail(XDFM_E_INVALID, "batch starting at line %lld: field index outside [1, FIELD_COUNT=%lld]",
                        (long long)r->first_line, (long long)F);
  }
  // every piece writes its own stretch of the arrays (tens of megabytes per batch: one core's
  // store bandwidth would be the bound)
  const bool sorted = r->in_order;
  const double tf = r->timing ? now_s() : 0;
  r->pool->run(r->pieces, [=](int t) {
    const Staging& w = r->workers[(size_t)t];
    const size_t base = (size_t)r->tok_base[(size_t)t], cnt = w.dnn_row.size();
    if (labels && w.hi > w.lo) memcpy(labels + w.lo, w.labels.data(), (size_t)(w.hi - w.lo) * sizeof(float));
    if (fm_feat_indices)
      for (size_t k = 0; k < cnt; ++k) {
        fm_feat_indices[2 * (base + k)] = w.tok_sample[k];
        fm_feat_indices[2 * (base + k) + 1] = w.tok_feat[k];
      }
    if (fm_feat_values && cnt) memcpy(fm_feat_values + base, w.tok_val.data(), cnt * sizeof(float));
    if (sorted) {
      if (dnn_feat_indices)
        for (size_t k = 0; k < cnt; ++k) {
          dnn_feat_indices[2 * (base + k)] = w.dnn_row[k];
          dnn_feat_indices[2 * (base + k) + 1] = w.dnn_col[k];
        }
      if (dnn_feat_values && cnt) memcpy(dnn_feat_values + base, w.tok_feat.data(), cnt * sizeof(int64_t));
      if (dnn_feat_weights && cnt) memcpy(dnn_feat_weights + base, w.tok_val.data(), cnt * sizeof(float));
    } else {
      // positions [base, base + cnt) of the sorted view, gathered from the joined staging
      const Staging& st = r->st;
      for (size_t k = base; k < base + cnt; ++k) {
        const uint32_t s = r->order[k];
        if (dnn_feat_indices) {
          dnn_feat_indices[2 * k] = st.dnn_row[s];
          dnn_feat_indices[2 * k + 1] = st.dnn_col[s];
        }
        if (dnn_feat_values) dnn_feat_values[k] = st.tok_feat[s];
        if (dnn_feat_weights) dnn_feat_weights[k] = st.tok_val[s];
      }
    }
    if (row_ptr) {
      // fields in range: this piece's tokens fall into rows [lo*F, hi*F) and no other piece's do,
      // so its CSR offsets are its token base plus a running count (entries lo*F+1 .. hi*F)
      const int64_t r0 = w.lo * F, r1 = w.hi * F;
      if (t == 0) row_ptr[0] = 0;
      for (int64_t i = r0 + 1; i <= r1; ++i) row_ptr[i] = 0;
      for (size_t k = 0; k < cnt; ++k) row_ptr[w.dnn_row[k] + 1] += 1;
      int32_t acc = (int32_t)base;
      for (int64_t i = r0 + 1; i <= r1; ++i) {
        acc += row_ptr[i];
        row_ptr[i] = acc;
      }
    }
    if (imp_offsets) {
      const int64_t ib = r->imp_base[(size_t)t];
      for (int64_t i = w.lo; i < w.hi; ++i) imp_offsets[i] = w.imp_off[(size_t)(i - w.lo)] + ib;
      if (imp_chars && !w.imp_chars.empty()) memcpy(imp_chars + ib, w.imp_chars.data(), w.imp_chars.size());
    }
    if (imp_present && w.hi > w.lo) memcpy(imp_present + w.lo, w.imp_present.data(), (size_t)(w.hi - w.lo));
  });
  if (imp_offsets) imp_offsets[n] = r->imp_base[(size_t)r->pieces];
  if (r->timing) r->t_fill += now_s() - tf;
  return XDFM_OK;
}
