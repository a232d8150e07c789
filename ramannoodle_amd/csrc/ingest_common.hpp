// Text helpers shared by the trajectory readers (ingest.cpp: XDATCAR, ingest_vasprun.cpp:
// vasprun.xml): Python-compatible whitespace splitting and float()/int() parsing.
#pragma once
#include <charconv>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>

namespace rn_ingest {

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
inline int split(sv s, sv *tok, int max_tokens) {
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
inline int count_tokens(sv s) {
  int n = 0;
  size_t i = 0;
  for (;;) {
    while (i < s.size() && is_space(s[i])) ++i;
    if (i >= s.size()) return n;
    while (i < s.size() && !is_space(s[i])) ++i;
    ++n;
  }
}

inline bool ieq(sv a, const char *b) {
  size_t n = strlen(b);
  if (a.size() != n) return false;
  for (size_t i = 0; i < n; ++i)
    if ((char)(a[i] | 0x20) != b[i]) return false;
  return true;
}

// Python's float(token) for a token without surrounding whitespace
inline bool parse_float(sv t, double &out) {
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
inline bool parse_int(sv t, long long &out) {
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

}  // namespace rn_ingest
