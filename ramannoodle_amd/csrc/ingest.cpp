// VASP XDATCAR trajectory reader behind include/rn_ingest.h (host only).
//
// Acceptance rules restate ramannoodle/io/vasp/poscar.py:18-121 and xdatcar.py:21-56 (see the
// header); numbers go through std::from_chars (correctly rounded, like Python's float) and the
// file is memory-mapped, indexed once and parsed frame-parallel.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../../include/rn_ingest.h"

namespace {

using sv = std::string_view;

// str.split() / str.strip() whitespace of Python for ASCII text
inline bool is_space(char c) {
  return c == ' ' || (c >= '\t' && c <= '\r') || (c >= '\x1c' && c <= '\x1f');
}
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

struct Line {
  sv text;     // without the line terminator
  bool ended;  // a terminator was present (Python's readline() result ends with '\n')
  std::string as_python() const {
    std::string s(text);
    if (ended) s.push_back('\n');
    return s;
  }
};

struct Cursor {
  const char *p, *end;
  bool at_end() const { return p >= end; }
  Line readline() {  // '' at EOF, like file.readline()
    if (p >= end) return {sv(), false};
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    const char *stop = nl ? nl : end;
    sv t(p, (size_t)(stop - p));
    if (nl && !t.empty() && t.back() == '\r') t.remove_suffix(1);  // universal newlines
    p = nl ? nl + 1 : end;
    return {t, nl != nullptr};
  }
};

// first `max_tokens` whitespace-separated tokens; returns how many were found (<= max_tokens)
int split(sv s, sv *tok, int max_tokens) {
  int n = 0;
  size_t i = 0;
  while (n < max_tokens) {
    while (i < s.size() && is_space(s[i])) ++i;
    if (i >= s.size()) break;
    size_t j = i;
    while (j < s.size() && !is_space(s[j])) ++j;
    tok[n++] = s.substr(i, j - i);
    i = j;
  }
  return n;
}
int count_tokens(sv s) {
  int n = 0;
  size_t i = 0;
  for (;;) {
    while (i < s.size() && is_space(s[i])) ++i;
    if (i >= s.size()) return n;
    while (i < s.size() && !is_space(s[i])) ++i;
    ++n;
  }
}

bool ieq(sv a, const char *b) {
  size_t n = strlen(b);
  if (a.size() != n) return false;
  for (size_t i = 0; i < n; ++i)
    if ((char)(a[i] | 0x20) != b[i]) return false;
  return true;
}

// Python's float(token) for a token without surrounding whitespace
bool parse_float(sv t, double &out) {
  if (t.empty()) return false;
  char buf[64];
  if (t.find('_') != sv::npos) {  // digit-group underscores: only between two digits
    if (t.size() >= sizeof(buf)) return false;
    size_t m = 0;
    for (size_t i = 0; i < t.size(); ++i) {
      if (t[i] == '_') {
        if (i == 0 || i + 1 >= t.size() || !is_digit(t[i - 1]) || !is_digit(t[i + 1])) return false;
      } else {
        buf[m++] = t[i];
      }
    }
    t = sv(buf, m);
  }
  bool neg = false;
  if (t[0] == '+' || t[0] == '-') {
    neg = t[0] == '-';
    t.remove_prefix(1);
    if (t.empty()) return false;
  }
  if (!is_digit(t[0]) && t[0] != '.') {  // the only words Python accepts
    if (ieq(t, "inf") || ieq(t, "infinity")) {
      out = neg ? -__builtin_inf() : __builtin_inf();
      return true;
    }
    if (ieq(t, "nan")) {
      out = __builtin_nan("");
      return true;
    }
    return false;
  }
  // Fast path (Clinger): a decimal with at most 19 significant digits whose integer mantissa is
  // below 2^53 and whose power of ten is at most 22 is ONE exactly-rounded multiply or divide of
  // two exactly representable doubles -- the correctly rounded result, as float() gives.
  // (libstdc++ 11's from_chars goes through strtod under a temporary locale: 30 ns per number
  // and no scaling over threads.)
  {
    static const double kPow10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    uint64_t mant = 0;
    int digits = 0, exp10 = 0;
    size_t i = 0;
    bool any = false, simple = true;
    for (; i < t.size() && is_digit(t[i]); ++i) {
      any = true;
      if (mant || t[i] != '0') {
        if (++digits > 19) simple = false;
        else mant = mant * 10 + (uint64_t)(t[i] - '0');
      }
    }
    if (i < t.size() && t[i] == '.') {
      for (++i; i < t.size() && is_digit(t[i]); ++i) {
        any = true;
        if (mant || t[i] != '0') {
          if (++digits > 19) simple = false;
          else mant = mant * 10 + (uint64_t)(t[i] - '0');
        }
        --exp10;
      }
    }
    if (any && simple && i < t.size() && (t[i] == 'e' || t[i] == 'E')) {
      size_t j = i + 1;
      bool eneg = false;
      if (j < t.size() && (t[j] == '+' || t[j] == '-')) eneg = t[j++] == '-';
      int e = 0;
      const size_t j0 = j;
      for (; j < t.size() && is_digit(t[j]) && e < 10000; ++j) e = e * 10 + (t[j] - '0');
      if (j > j0 && j == t.size()) {
        exp10 += eneg ? -e : e;
        i = j;
      } else {
        simple = false;
      }
    }
    if (any && simple && i == t.size() && mant < (1ULL << 53)) {
      if (mant == 0) {
        out = neg ? -0.0 : 0.0;
        return true;
      }
      if (exp10 >= -22 && exp10 <= 22) {
        const double m = (double)mant;
        const double v = exp10 < 0 ? m / kPow10[-exp10] : m * kPow10[exp10];
        out = neg ? -v : v;
        return true;
      }
    }
  }
  double v = 0;
  const auto r = std::from_chars(t.data(), t.data() + t.size(), v, std::chars_format::general);
  if (r.ptr != t.data() + t.size()) return false;
  if (r.ec == std::errc::result_out_of_range) {  // Python: inf on overflow, 0 / denormal on underflow
    std::string z(t);
    v = strtod(z.c_str(), nullptr);
  } else if (r.ec != std::errc()) {
    return false;
  }
  out = neg ? -v : v;
  return true;
}

// Python's int(token): optional sign, digits, digit-group underscores
bool parse_int(sv t, long long &out) {
  if (t.empty()) return false;
  bool neg = false;
  size_t i = 0;
  if (t[0] == '+' || t[0] == '-') {
    neg = t[0] == '-';
    i = 1;
  }
  if (i >= t.size()) return false;
  long long v = 0;
  for (size_t k = i; k < t.size(); ++k) {
    if (t[k] == '_') {
      if (k == i || k + 1 >= t.size() || !is_digit(t[k - 1]) || !is_digit(t[k + 1])) return false;
      continue;
    }
    if (!is_digit(t[k])) return false;
    v = v * 10 + (t[k] - '0');
    if (v > (1LL << 40)) return false;
  }
  out = neg ? -v : v;
  return true;
}

const char *const kSymbols[] = {
    "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",  "Cl",
    "Ar", "K",  "Ca", "Sc", "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se",
    "Br", "Kr", "Rb", "Sr", "Y",  "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb",
    "Te", "I",  "Xe", "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er",
    "Tm", "Yb", "Lu", "Hf", "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po", "At",
    "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U",  "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es", "Fm", "Md", "No",
    "Lr", "Rf", "Db", "Sg", "Bh", "Hs", "Mt", "Ds", "Rg", "Cn", "Nh", "Fl", "Mc", "Lv", "Ts", "Og"};
bool known_symbol(sv s) {
  for (const char *k : kSymbols)
    if (s == k) return true;
  return false;
}

struct Frame {
  const char *rows;  // first position row
  uint8_t cartesian;
  uint8_t bad_label;  // unrecognised coordinate format: reading this frame fails
  std::string label;  // kept only for bad labels (message)
};

}  // namespace

struct rn_xdatcar {
  int fd = -1;
  const char *data = nullptr;
  size_t size = 0;
  double lattice[9] = {0};
  std::vector<std::string> symbols;
  std::vector<long long> counts;
  int32_t num_atoms = 0;
  std::vector<Frame> frames;
  std::string error;
  ~rn_xdatcar() {
    if (data && size) munmap(const_cast<char *>(data), size);
    if (fd >= 0) close(fd);
  }
};

namespace {

bool parse_header(rn_xdatcar *h, Cursor &c) {
  c.readline();  // comment
  Line line = c.readline();
  double scale = 0;
  {
    sv tok[2];
    if (split(line.text, tok, 2) != 1 || !parse_float(tok[0], scale)) {
      h->error = "scale factor could not be parsed: " + line.as_python();
      return false;
    }
  }
  for (int r = 0; r < 3; ++r) {
    line = c.readline();
    sv tok[3];
    const int n = split(line.text, tok, 3);
    double v[3];
    bool ok = n == 3;
    for (int k = 0; k < n && ok; ++k) ok = parse_float(tok[k], v[k]);
    if (!ok) {  // an invalid token fails float(), fewer than three numbers the shape test
      h->error = "lattice could not be parsed: " + line.as_python();
      return false;
    }
    for (int k = 0; k < 3; ++k) h->lattice[3 * r + k] = v[k] * scale;
  }
  line = c.readline();
  const int nsym = count_tokens(line.text);
  if (nsym == 0) {
    h->error = "no atom symbols found";
    return false;
  }
  std::vector<sv> sym((size_t)nsym);
  split(line.text, sym.data(), nsym);
  for (sv s : sym) {
    if (!known_symbol(s)) {
      h->error = "unrecognized atom symbol: " + std::string(s);
      return false;
    }
    h->symbols.emplace_back(s);
  }
  line = c.readline();
  const int ncnt = count_tokens(line.text);
  if (ncnt != nsym) {
    h->error = "wrong number of ion counts: " + std::to_string(ncnt) + " != " + std::to_string(nsym);
    return false;
  }
  std::vector<sv> cnt((size_t)ncnt);
  split(line.text, cnt.data(), ncnt);
  long long total = 0;
  for (sv s : cnt) {
    long long v;
    if (!parse_int(s, v)) {
      h->error = "could not parse counts: " + line.as_python();
      return false;
    }
    h->counts.push_back(v);
    total += std::max<long long>(v, 0);  // [symbol] * negative == []
  }
  if (total > (1LL << 30)) {
    h->error = "could not parse counts: " + line.as_python();
    return false;
  }
  h->num_atoms = (int32_t)total;
  return true;
}

void index_frames(rn_xdatcar *h, Cursor &c) {
  for (;;) {
    Line label = c.readline();
    // empty / whitespace-only / leading whitespace: the reference's end-of-trajectory signal
    if (label.text.empty() || is_space(label.text[0])) return;
    char first = (char)(label.text[0] | 0x20);
    if (first == 's') {  // selective dynamics: the coordinate label follows
      label = c.readline();
      first = label.text.empty() ? '\0' : (char)(label.text[0] | 0x20);
    }
    Frame f{c.p, (uint8_t)(first == 'c'), (uint8_t)(first != 'c' && first != 'd'), std::string()};
    if (f.bad_label) {
      f.label = label.as_python();
      h->frames.push_back(std::move(f));
      return;  // the reference raises here
    }
    h->frames.push_back(std::move(f));
    // (a truncated last frame is reported when it is parsed: its missing rows read as '')
    for (int32_t r = 0; r < h->num_atoms; ++r) c.readline();
  }
}

// parses one frame; on failure fills `err` (reference wording) and returns false
bool parse_frame(const rn_xdatcar *h, const Frame &f, double *out, std::string &err) {
  if (f.bad_label) {
    err = "unrecognized coordinate format: " + f.label;
    return false;
  }
  Cursor c{f.rows, h->data + h->size};
  for (int32_t r = 0; r < h->num_atoms; ++r) {
    const Line line = c.readline();
    sv tok[3];
    const int n = split(line.text, tok, 3);
    bool ok = n == 3;
    for (int k = 0; k < n && ok; ++k) ok = parse_float(tok[k], out[3 * (size_t)r + k]);
    if (!ok) {
      err = "positions could not be parsed: " + line.as_python();
      return false;
    }
  }
  return true;
}

}  // namespace

extern "C" {

int rn_xdatcar_open(const char *path, rn_xdatcar **out) {
  if (!path || !out) return RN_INGEST_INVALID_ARGUMENT;
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return RN_INGEST_FILE_NOT_FOUND;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
    close(fd);
    return RN_INGEST_FILE_NOT_FOUND;
  }
  rn_xdatcar *h = new rn_xdatcar();
  h->fd = fd;
  h->size = (size_t)st.st_size;
  if (h->size) {
    void *m = mmap(nullptr, h->size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      delete h;
      return RN_INGEST_FILE_NOT_FOUND;
    }
    h->data = static_cast<const char *>(m);
    (void)madvise(m, h->size, MADV_SEQUENTIAL);
  }
  *out = h;
  Cursor c{h->data, h->data + h->size};
  if (!parse_header(h, c)) return RN_INGEST_INVALID_FILE;
  index_frames(h, c);
  return RN_INGEST_OK;
}

void rn_xdatcar_close(rn_xdatcar *h) { delete h; }

int rn_xdatcar_info(const rn_xdatcar *h, int64_t *num_frames, int32_t *num_atoms, double *lattice,
                    int32_t *num_species) {
  if (!h) return RN_INGEST_INVALID_ARGUMENT;
  if (num_frames) *num_frames = (int64_t)h->frames.size();
  if (num_atoms) *num_atoms = h->num_atoms;
  if (lattice) std::memcpy(lattice, h->lattice, sizeof(h->lattice));
  if (num_species) *num_species = (int32_t)h->symbols.size();
  return RN_INGEST_OK;
}

int rn_xdatcar_species(const rn_xdatcar *h, int32_t index, char *symbol, int32_t *count) {
  if (!h || index < 0 || (size_t)index >= h->symbols.size() || !symbol) return RN_INGEST_INVALID_ARGUMENT;
  std::strncpy(symbol, h->symbols[(size_t)index].c_str(), 7);
  symbol[7] = '\0';
  if (count) *count = (int32_t)std::max<long long>(h->counts[(size_t)index], 0);
  return RN_INGEST_OK;
}

int rn_xdatcar_read(rn_xdatcar *h, int64_t first, int64_t count, double *positions, uint8_t *cartesian,
                    int num_threads) {
  if (!h || first < 0 || count < 0 || first + count > (int64_t)h->frames.size() || (count > 0 && !positions))
    return RN_INGEST_INVALID_ARGUMENT;
  if (count == 0) return RN_INGEST_OK;
  {
    // Map the block's pages in one call: faulting them in one by one from many threads
    // serialises on the address-space lock (measured: no speed-up from threads at all).
    const char *b = h->frames[(size_t)first].rows;
    const char *e = (first + count < (int64_t)h->frames.size()) ? h->frames[(size_t)(first + count)].rows
                                                                : h->data + h->size;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = (uintptr_t)b & ~(page - 1);
#ifdef MADV_POPULATE_READ
    (void)madvise(reinterpret_cast<void *>(lo), (size_t)((uintptr_t)e - lo), MADV_POPULATE_READ);
#else
    (void)madvise(reinterpret_cast<void *>(lo), (size_t)((uintptr_t)e - lo), MADV_WILLNEED);
#endif
  }
  if (num_threads <= 0) num_threads = (int)std::min<int64_t>(16, (count + 63) / 64);
  num_threads = std::max(1, std::min<int>(num_threads, (int)std::min<int64_t>(count, 64)));
  const size_t stride = (size_t)h->num_atoms * 3;
  std::atomic<int64_t> next{0}, first_bad{count};
  std::vector<std::string> errors((size_t)num_threads);
  std::vector<int64_t> bad_at((size_t)num_threads, count);
  auto work = [&](int t) {
    for (;;) {
      const int64_t k0 = next.fetch_add(16);
      if (k0 >= count || k0 >= first_bad.load()) return;
      const int64_t k1 = std::min(count, k0 + 16);
      for (int64_t k = k0; k < k1; ++k) {
        const Frame &f = h->frames[(size_t)(first + k)];
        if (cartesian) cartesian[k] = f.cartesian;
        std::string err;
        if (!parse_frame(h, f, positions + (size_t)k * stride, err)) {
          if (k < bad_at[(size_t)t]) {
            bad_at[(size_t)t] = k;
            errors[(size_t)t] = std::move(err);
          }
          int64_t cur = first_bad.load();
          while (k < cur && !first_bad.compare_exchange_weak(cur, k)) {
          }
          return;
        }
      }
    }
  };
  if (num_threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < num_threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
  }
  const int64_t bad = first_bad.load();
  if (bad < count) {  // the reference stops at the first offending frame: report that one
    for (int t = 0; t < num_threads; ++t)
      if (bad_at[(size_t)t] == bad) h->error = errors[(size_t)t];
    return RN_INGEST_INVALID_FILE;
  }
  return RN_INGEST_OK;
}

const char *rn_xdatcar_last_error(const rn_xdatcar *h) { return h ? h->error.c_str() : "null handle"; }

}  // extern "C"
