// VASP vasprun.xml reader behind include/rn_ingest.h (host only): the positions of an MD run
// (the un-named <structure> children of the root, ramannoodle/io/vasp/vasprun.py:298-330), the
// initial structure (lattice, species, positions: vasprun.py:33-74, 94-115, 244-275) and the
// time step (vasprun.py:278-295).
//
// The reference hands the whole file to ElementTree and walks the tree in Python.  Here the file
// is memory-mapped and tokenised ONCE (tags, nesting and names are checked, so an ill-formed
// document is refused as ElementTree refuses it); that pass indexes the children of the root.
// Everything else is found by re-scanning the few small sub-ranges the reference's paths name, and
// the frames' rows are parsed on demand, frame-parallel, straight into the caller's buffer.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../../include/rn_ingest.h"
#include "ingest_common.hpp"

using namespace rn_ingest;

namespace {

// XML white space (S production)
inline bool xml_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool name_char(char c) {
  return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == ':' ||
         c == '-' || c == '.' || (unsigned char)c >= 0x80;
}

struct Node {
  sv tag;
  sv name;  // value of the attribute called "name" (ElementTree's [@name='...'] tests)
  bool has_name = false;
  const char *content = nullptr;      // first byte after the start tag
  const char *content_end = nullptr;  // the '<' of the end tag (== content for <x/>)
  const char *end = nullptr;          // first byte after the element
};

struct Scanner {
  const char *p, *end;
  std::string error;  // set on an ill-formed document

  bool fail(const char *what) {
    if (error.empty()) error = what;
    return false;
  }
  // skips "<!-- -->", "<? ?>", "<!DOCTYPE >" and "<![CDATA[ ]]>" starting at p (which points at '<');
  // returns false when p is not at one of them
  bool skip_misc(bool &ok) {
    ok = true;
    const size_t left = (size_t)(end - p);
    if (left >= 4 && !memcmp(p, "<!--", 4)) {
      const char *q = p + 4;
      for (;;) {
        q = static_cast<const char *>(memchr(q, '-', (size_t)(end - q)));
        if (!q || q + 2 >= end) { ok = fail("unterminated comment"); return true; }
        if (q[1] == '-') {
          if (q[2] != '>') { ok = fail("'--' inside a comment"); return true; }
          p = q + 3;
          return true;
        }
        ++q;
      }
    }
    if (left >= 2 && p[1] == '?') {
      const char *q = p + 2;
      for (;;) {
        q = static_cast<const char *>(memchr(q, '?', (size_t)(end - q)));
        if (!q || q + 1 >= end) { ok = fail("unterminated processing instruction"); return true; }
        if (q[1] == '>') { p = q + 2; return true; }
        ++q;
      }
    }
    if (left >= 9 && !memcmp(p, "<![CDATA[", 9)) {
      const char *q = p + 9;
      for (;;) {
        q = static_cast<const char *>(memchr(q, ']', (size_t)(end - q)));
        if (!q || q + 2 >= end) { ok = fail("unterminated CDATA section"); return true; }
        if (q[1] == ']' && q[2] == '>') { p = q + 3; return true; }
        ++q;
      }
    }
    if (left >= 2 && p[1] == '!') {  // DOCTYPE and friends: skip to the matching '>'
      int depth = 0;
      for (const char *q = p; q < end; ++q) {
        if (*q == '<') ++depth;
        else if (*q == '>' && --depth == 0) { p = q + 1; return true; }
      }
      ok = fail("unterminated declaration");
      return true;
    }
    return false;
  }

  // p at '<' of a start tag: parses name + attributes; on return p is after '>' ;
  // `selfclosed` tells <x/>
  bool start_tag(Node &n, bool &selfclosed) {
    const char *q = p + 1;
    const char *b = q;
    while (q < end && name_char(*q)) ++q;
    if (q == b) return fail("not well-formed (invalid token)");
    n.tag = sv(b, (size_t)(q - b));
    n.has_name = false;
    for (;;) {
      const char *ws = q;
      while (q < end && xml_space(*q)) ++q;
      if (q >= end) return fail("unclosed token");
      if (*q == '>') { selfclosed = false; p = q + 1; break; }
      if (*q == '/') {
        if (q + 1 >= end || q[1] != '>') return fail("not well-formed (invalid token)");
        selfclosed = true;
        p = q + 2;
        break;
      }
      if (q == ws) return fail("not well-formed (invalid token)");  // attributes need white space before them
      const char *ab = q;
      while (q < end && name_char(*q)) ++q;
      if (q == ab) return fail("not well-formed (invalid token)");
      const sv attr(ab, (size_t)(q - ab));
      while (q < end && xml_space(*q)) ++q;
      if (q >= end || *q != '=') return fail("not well-formed (invalid token)");
      ++q;
      while (q < end && xml_space(*q)) ++q;
      if (q >= end || (*q != '"' && *q != '\'')) return fail("not well-formed (invalid token)");
      const char quote = *q++;
      const char *vb = q;
      q = static_cast<const char *>(memchr(q, quote, (size_t)(end - q)));
      if (!q) return fail("unclosed token");
      if (memchr(vb, '<', (size_t)(q - vb))) return fail("not well-formed (invalid token)");
      if (attr == "name") {
        if (n.has_name) return fail("duplicate attribute");
        n.name = sv(vb, (size_t)(q - vb));
        n.has_name = true;
      }
      ++q;
    }
    n.content = p;
    return true;
  }

  // p just after a start tag of `tag`: consumes the element's content and its end tag, checking the
  // nesting of everything inside; returns the '<' of the end tag in *content_end
  bool skip_content(sv tag, const char **content_end) {
    std::vector<sv> stack;
    stack.push_back(tag);
    while (!stack.empty()) {
      const char *lt = static_cast<const char *>(memchr(p, '<', (size_t)(end - p)));
      if (!lt) return fail("no element found");  // EOF inside an element
      p = lt;
      bool ok;
      if (skip_misc(ok)) {
        if (!ok) return false;
        continue;
      }
      if (p + 1 < end && p[1] == '/') {
        const char *q = p + 2, *b = q;
        while (q < end && name_char(*q)) ++q;
        const sv closing(b, (size_t)(q - b));
        while (q < end && xml_space(*q)) ++q;
        if (q >= end || *q != '>') return fail("not well-formed (invalid token)");
        if (closing != stack.back()) return fail("mismatched tag");
        stack.pop_back();
        if (stack.empty()) *content_end = p;
        p = q + 1;
        continue;
      }
      Node child;
      bool selfclosed;
      if (!start_tag(child, selfclosed)) return false;
      if (!selfclosed) stack.push_back(child.tag);
    }
    return true;
  }
};

// Child elements directly inside [b, e) of an already validated document.
std::vector<Node> children(const char *b, const char *e) {
  std::vector<Node> out;
  Scanner s{b, e, {}};
  while (s.p < e) {
    const char *lt = static_cast<const char *>(memchr(s.p, '<', (size_t)(e - s.p)));
    if (!lt) break;
    s.p = lt;
    bool ok;
    if (s.skip_misc(ok)) {
      if (!ok) break;
      continue;
    }
    if (s.p + 1 < e && s.p[1] == '/') break;  // (the parent's end tag: not inside [b, e) normally)
    Node n;
    bool selfclosed;
    if (!s.start_tag(n, selfclosed)) break;
    if (selfclosed) {
      n.content_end = n.content = s.p;
      n.end = s.p;
    } else {
      if (!s.skip_content(n.tag, &n.content_end)) break;
      n.end = s.p;
    }
    out.push_back(n);
  }
  return out;
}

// ElementTree's element.text: the character data before the first child element (comments and
// processing instructions dropped, CDATA kept, the five predefined and numeric entities decoded).
// Returns false for "text is None" (no character data at all).
bool element_text(const Node &n, std::string &scratch, sv &text) {
  const char *p = n.content, *e = n.content_end;
  const char *lt = static_cast<const char *>(memchr(p, '<', (size_t)(e - p)));
  const bool amp = memchr(p, '&', (size_t)((lt ? lt : e) - p)) != nullptr;
  if (!lt && !amp) {  // the common case: plain text up to the end tag
    text = sv(p, (size_t)(e - p));
    return !text.empty();
  }
  scratch.clear();
  Scanner s{p, e, {}};
  while (s.p < e) {
    if (*s.p == '<') {
      if ((size_t)(e - s.p) >= 9 && !memcmp(s.p, "<![CDATA[", 9)) {
        const char *b = s.p + 9;
        bool ok;
        s.skip_misc(ok);
        if (!ok) break;
        scratch.append(b, (size_t)(s.p - 3 - b));
        continue;
      }
      bool ok;
      if (s.skip_misc(ok)) {
        if (!ok) break;
        continue;
      }
      break;  // first child element: text ends here
    }
    if (*s.p == '&') {
      const char *semi = static_cast<const char *>(memchr(s.p, ';', (size_t)(e - s.p)));
      if (semi) {
        const sv ent(s.p + 1, (size_t)(semi - s.p - 1));
        char c = 0;
        if (ent == "lt") c = '<';
        else if (ent == "gt") c = '>';
        else if (ent == "amp") c = '&';
        else if (ent == "quot") c = '"';
        else if (ent == "apos") c = '\'';
        else if (ent.size() > 1 && ent[0] == '#') {
          const long v = (ent[1] == 'x') ? strtol(std::string(ent.substr(2)).c_str(), nullptr, 16)
                                         : strtol(std::string(ent.substr(1)).c_str(), nullptr, 10);
          if (v > 0 && v < 128) c = (char)v;
        }
        if (c) {
          scratch.push_back(c);
          s.p = semi + 1;
          continue;
        }
      }
    }
    scratch.push_back(*s.p++);
  }
  text = scratch;
  return !scratch.empty();
}

const Node *find(const std::vector<Node> &nodes, sv tag, const char *name = nullptr) {
  for (const Node &n : nodes)
    if (n.tag == tag && (!name || (n.has_name && n.name == name))) return &n;
  return nullptr;
}

inline sv strip(sv s) {  // str.strip()
  while (!s.empty() && is_space(s.front())) s.remove_prefix(1);
  while (!s.empty() && is_space(s.back())) s.remove_suffix(1);
  return s;
}

struct FrameRef {
  const char *rows = nullptr, *rows_end = nullptr;  // content of the structure's first <varray>
  bool has_varray = false;
};

}  // namespace

struct rn_vasprun {
  int fd = -1;
  const char *data = nullptr;
  size_t size = 0;
  std::vector<Node> top;          // children of the root element
  std::vector<FrameRef> frames;   // un-named <structure> children, in document order
  int32_t num_atoms = -1;         // rows of the first frame (every frame must match)
  std::string error;
  int error_kind = 0;  // RN_INGEST_INVALID_FILE or RN_INGEST_VALUE_ERROR of the last failure
  ~rn_vasprun() {
    if (data && size) munmap(const_cast<char *>(data), size);
    if (fd >= 0) close(fd);
  }
};

namespace {

int invalid(rn_vasprun *h, std::string msg) {
  h->error = std::move(msg);
  h->error_kind = RN_INGEST_INVALID_FILE;
  return RN_INGEST_INVALID_FILE;
}
int value_error(rn_vasprun *h, std::string msg) {
  h->error = std::move(msg);
  h->error_kind = RN_INGEST_VALUE_ERROR;
  return RN_INGEST_VALUE_ERROR;
}

// [float(i) for i in text.split()] of every child of a <varray>, `width` numbers per row expected.
// rc: 0 ok, RN_INGEST_INVALID_FILE (child without text), RN_INGEST_VALUE_ERROR (float() fails / ragged)
int parse_rows(const char *b, const char *e, int width, int64_t max_rows, double *out, int64_t *rows_found,
               std::string &err) {
  int64_t r = 0;
  std::string scratch;
  for (const Node &child : children(b, e)) {
    sv text;
    if (!element_text(child, scratch, text)) {
      err = "varray child text not found";
      return RN_INGEST_INVALID_FILE;
    }
    int k = 0;
    size_t i = 0;
    for (;;) {
      while (i < text.size() && is_space(text[i])) ++i;
      if (i >= text.size()) break;
      size_t j = i;
      while (j < text.size() && !is_space(text[j])) ++j;
      double v;
      if (!parse_float(text.substr(i, j - i), v)) {
        err = "could not convert string to float: '" + std::string(text.substr(i, j - i)) + "'";
        return RN_INGEST_VALUE_ERROR;
      }
      if (out && k < width && r < max_rows) out[(size_t)r * (size_t)width + (size_t)k] = v;
      ++k;
      i = j;
    }
    if (k != width) {
      err = "setting an array element with a sequence: a row of " + std::to_string(k) + " numbers where " +
            std::to_string(width) + " are expected";
      return RN_INGEST_VALUE_ERROR;
    }
    ++r;
  }
  *rows_found = r;
  return RN_INGEST_OK;
}

int64_t count_children(const char *b, const char *e) { return (int64_t)children(b, e).size(); }

}  // namespace

extern "C" {

int rn_vasprun_open(const char *path, rn_vasprun **out) {
  if (!path || !out) return RN_INGEST_INVALID_ARGUMENT;
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return RN_INGEST_FILE_NOT_FOUND;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
    close(fd);
    return RN_INGEST_FILE_NOT_FOUND;
  }
  rn_vasprun *h = new rn_vasprun();
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
  const char *const kNoRoot = "root xml element could not be found";  // vasprun.py:22-29
  Scanner s{h->data, h->data + h->size, {}};
  if (h->size >= 3 && !memcmp(s.p, "\xEF\xBB\xBF", 3)) s.p += 3;
  // prolog: white space, declaration, comments, DOCTYPE
  Node root;
  for (;;) {
    while (s.p < s.end && xml_space(*s.p)) ++s.p;
    if (s.p >= s.end) return invalid(h, kNoRoot);  // "no element found"
    if (*s.p != '<') return invalid(h, kNoRoot);   // text before the root
    bool ok;
    if (s.skip_misc(ok)) {
      if (!ok) return invalid(h, kNoRoot);
      continue;
    }
    break;
  }
  bool selfclosed;
  if (!s.start_tag(root, selfclosed)) return invalid(h, kNoRoot);
  if (!selfclosed) {
    // one pass over the document: direct children of the root, everything inside them validated
    for (;;) {
      const char *lt = static_cast<const char *>(memchr(s.p, '<', (size_t)(s.end - s.p)));
      if (!lt) return invalid(h, kNoRoot);
      s.p = lt;
      bool ok;
      if (s.skip_misc(ok)) {
        if (!ok) return invalid(h, kNoRoot);
        continue;
      }
      if (s.p + 1 < s.end && s.p[1] == '/') {
        const char *q = s.p + 2, *b = q;
        while (q < s.end && name_char(*q)) ++q;
        const sv closing(b, (size_t)(q - b));
        while (q < s.end && xml_space(*q)) ++q;
        if (q >= s.end || *q != '>' || closing != root.tag) return invalid(h, kNoRoot);
        s.p = q + 1;
        break;
      }
      Node n;
      bool sc;
      if (!s.start_tag(n, sc)) return invalid(h, kNoRoot);
      if (sc) {
        n.content_end = n.content;
        n.end = s.p;
      } else {
        if (!s.skip_content(n.tag, &n.content_end)) return invalid(h, kNoRoot);
        n.end = s.p;
      }
      h->top.push_back(n);
    }
  }
  // epilog: only white space, comments and processing instructions may follow the root
  for (;;) {
    while (s.p < s.end && xml_space(*s.p)) ++s.p;
    if (s.p >= s.end) break;
    bool ok;
    if (*s.p == '<' && s.skip_misc(ok) && ok) continue;
    return invalid(h, kNoRoot);  // "junk after document element"
  }
  // index the trajectory: un-named <structure> children of the root (vasprun.py:314-322)
  for (const Node &n : h->top) {
    if (n.tag != "structure" || n.has_name) continue;
    FrameRef f;
    for (const Node &c : children(n.content, n.content_end))
      if (c.tag == "varray") {
        f.rows = c.content;
        f.rows_end = c.content_end;
        f.has_varray = true;
        break;
      }
    h->frames.push_back(f);
  }
  if (!h->frames.empty() && h->frames[0].has_varray)
    h->num_atoms = (int32_t)count_children(h->frames[0].rows, h->frames[0].rows_end);
  return RN_INGEST_OK;
}

void rn_vasprun_close(rn_vasprun *h) { delete h; }

int rn_vasprun_info(const rn_vasprun *h, int64_t *num_frames, int32_t *num_atoms) {
  if (!h) return RN_INGEST_INVALID_ARGUMENT;
  if (num_frames) *num_frames = (int64_t)h->frames.size();
  if (num_atoms) *num_atoms = h->num_atoms;
  return RN_INGEST_OK;
}

int rn_vasprun_read(rn_vasprun *h, int64_t first, int64_t count, double *positions, int num_threads) {
  if (!h || first < 0 || count < 0 || first + count > (int64_t)h->frames.size() || (count > 0 && !positions))
    return RN_INGEST_INVALID_ARGUMENT;
  if (count == 0) return RN_INGEST_OK;
  if (num_threads <= 0) num_threads = (int)std::min<int64_t>(16, (count + 63) / 64);
  num_threads = std::max(1, std::min<int>(num_threads, (int)std::min<int64_t>(count, 64)));
  const int32_t n = h->num_atoms;
  const size_t stride = (size_t)std::max(n, 0) * 3;
  std::atomic<int64_t> next{0}, first_bad{count};
  struct Failure { int64_t at; int rc; std::string msg; };
  std::vector<Failure> failures((size_t)num_threads, Failure{count, 0, {}});
  auto work = [&](int t) {
    for (;;) {
      const int64_t k0 = next.fetch_add(8);
      if (k0 >= count || k0 >= first_bad.load()) return;
      const int64_t k1 = std::min(count, k0 + 8);
      for (int64_t k = k0; k < k1; ++k) {
        const FrameRef &f = h->frames[(size_t)(first + k)];
        std::string err;
        int rc = RN_INGEST_OK;
        int64_t rows = 0;
        if (!f.has_varray) {
          rc = RN_INGEST_INVALID_FILE;
          err = "structure varray not found";
        } else {
          rc = parse_rows(f.rows, f.rows_end, 3, n, positions + (size_t)k * stride, &rows, err);
          if (rc == RN_INGEST_OK && rows != n) {
            rc = RN_INGEST_VALUE_ERROR;
            err = "setting an array element with a sequence: frame " + std::to_string(first + k) + " has " +
                  std::to_string(rows) + " atoms, the first frame " + std::to_string(n);
          }
        }
        if (rc != RN_INGEST_OK) {
          if (k < failures[(size_t)t].at) failures[(size_t)t] = Failure{k, rc, std::move(err)};
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
    for (const Failure &f : failures)
      if (f.at == bad) return f.rc == RN_INGEST_VALUE_ERROR ? value_error(h, f.msg) : invalid(h, f.msg);
  }
  return RN_INGEST_OK;
}

int rn_vasprun_timestep(rn_vasprun *h, double *timestep) {
  if (!h || !timestep) return RN_INGEST_INVALID_ARGUMENT;
  // ./parameters/separator[@name='ionic']/i/[@name='POTIM']: first match in document order
  for (const Node &par : h->top) {
    if (par.tag != "parameters") continue;
    for (const Node &sep : children(par.content, par.content_end)) {
      if (sep.tag != "separator" || !sep.has_name || sep.name != "ionic") continue;
      for (const Node &i : children(sep.content, sep.content_end)) {
        if (i.tag != "i" || !i.has_name || i.name != "POTIM") continue;
        std::string scratch;
        sv text;
        if (!element_text(i, scratch, text)) return invalid(h, "potim element has no text");
        const sv t = strip(text);
        if (!parse_float(t, *timestep))
          return value_error(h, "could not convert string to float: '" + std::string(t) + "'");
        return RN_INGEST_OK;
      }
    }
  }
  return invalid(h, "timestep not found");
}

int rn_vasprun_initial_structure(rn_vasprun *h, int32_t *num_atoms, double *lattice, double *positions,
                                 int64_t positions_capacity, char *symbols, int64_t symbols_capacity) {
  if (!h) return RN_INGEST_INVALID_ARGUMENT;
  // atomic symbols: ./atominfo/array/set, child[0].text.strip() of every child (vasprun.py:33-50)
  if (symbols) {
    const Node *set = nullptr;
    std::vector<Node> keep;  // (nodes point into the mapped file, the vectors only hold the descriptors)
    for (const Node &ai : h->top) {
      if (ai.tag != "atominfo" || set) continue;
      for (const Node &arr : children(ai.content, ai.content_end)) {
        if (arr.tag != "array" || set) continue;
        keep = children(arr.content, arr.content_end);
        set = find(keep, "set");
      }
    }
    if (!set) return invalid(h, "atomic symbols not found");
    int64_t used = 0, count = 0;
    for (const Node &rc : children(set->content, set->content_end)) {
      const std::vector<Node> cells = children(rc.content, rc.content_end);
      if (cells.empty()) return value_error(h, "child index out of range");  // IndexError in the reference
      std::string scratch;
      sv text;
      if (!element_text(cells[0], scratch, text)) return invalid(h, "child has no text");
      const sv sym = strip(text);
      if (used + (int64_t)sym.size() + 1 > symbols_capacity) return RN_INGEST_INVALID_ARGUMENT;
      memcpy(symbols + used, sym.data(), sym.size());
      used += (int64_t)sym.size();
      symbols[used++] = '\n';
      ++count;
    }
    if (used < symbols_capacity) symbols[used] = '\0';
    else return RN_INGEST_INVALID_ARGUMENT;
    if (num_atoms) *num_atoms = (int32_t)count;
  }
  const Node *initial = find(h->top, "structure", "initialpos");
  if (lattice) {  // ./structure[@name='initialpos']/crystal/varray[@name='basis'] (vasprun.py:94-115)
    const Node *basis = nullptr;
    std::vector<Node> keep;
    for (const Node &st : h->top) {
      if (st.tag != "structure" || !st.has_name || st.name != "initialpos" || basis) continue;
      for (const Node &cr : children(st.content, st.content_end)) {
        if (cr.tag != "crystal" || basis) continue;
        keep = children(cr.content, cr.content_end);
        basis = find(keep, "varray", "basis");
      }
    }
    if (!basis) return invalid(h, "lattice not found");
    std::string err;
    int64_t rows = 0;
    double tmp[9];
    const int rc = parse_rows(basis->content, basis->content_end, 3, 3, tmp, &rows, err);
    if (rc == RN_INGEST_INVALID_FILE) return invalid(h, err);
    if (rc != RN_INGEST_OK) return value_error(h, err);
    if (rows != 3) return value_error(h, "lattice has " + std::to_string(rows) + " rows");
    memcpy(lattice, tmp, sizeof(tmp));
  }
  if (positions) {  // ./structure[@name='initialpos']/varray (vasprun.py:53-74, 236-241)
    const Node *varray = nullptr;
    std::vector<Node> keep;
    if (initial) {
      keep = children(initial->content, initial->content_end);
      varray = find(keep, "varray");
    }
    if (!varray) return invalid(h, "initial positions not found");
    const int64_t n = count_children(varray->content, varray->content_end);
    if (n * 3 > positions_capacity) {
      if (num_atoms) *num_atoms = (int32_t)n;
      return RN_INGEST_INVALID_ARGUMENT;
    }
    std::string err;
    int64_t rows = 0;
    const int rc = parse_rows(varray->content, varray->content_end, 3, n, positions, &rows, err);
    if (rc == RN_INGEST_INVALID_FILE) return invalid(h, err);
    if (rc != RN_INGEST_OK) return value_error(h, err);
    if (num_atoms && !symbols) *num_atoms = (int32_t)rows;
  }
  return RN_INGEST_OK;
}

int64_t rn_vasprun_initial_num_atoms(rn_vasprun *h) {
  if (!h) return -1;
  const Node *initial = find(h->top, "structure", "initialpos");
  if (!initial) return -1;
  const std::vector<Node> kids = children(initial->content, initial->content_end);
  const Node *varray = find(kids, "varray");
  return varray ? count_children(varray->content, varray->content_end) : -1;
}

const char *rn_vasprun_last_error(const rn_vasprun *h) { return h ? h->error.c_str() : "null handle"; }

}  // extern "C"
