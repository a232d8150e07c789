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
#include "ingest_common.hpp"

namespace {

using namespace rn_ingest;

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
