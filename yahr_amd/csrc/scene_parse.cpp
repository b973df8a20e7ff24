// scene_parse.cpp -- loader for the .yahrr scene text format.
//
// The format is whatever GHC's *derived* `Read` instance accepts for `Scene` (Scene.hs:42,50,58;
// Vectors.hs:6; Cameras.hs:56; Culling.hs:18-19; Integrators.hs:20; Lights.hs:7), as used by
// `read sceneFile :: S.Scene` (main.hs:117).  This is a recursive-descent restatement of that grammar:
//   * Haskell lexing: free whitespace, identifiers, numeric and string literals, ( ) [ ] { } , =
//   * any value may be wrapped in redundant parentheses; a constructor applied to arguments needs them
//     when it is itself an argument (precedence 11), a record or nullary constructor does not
//   * record syntax needs every field, in declaration order
//   * a leading '-' is accepted before a numeric literal at any precedence (GHC.Read.readNumber), which
//     is why `Vec3 0.4 -0.3 1` (scene.yahrr:6) and `(Vec3 -1000.000000 ...)` (compat/yahr.py:142) parse
//   * integer literals are accepted for Float fields; Int fields take integer literals only
// `lenient` additionally accepts the stale dialect of the bundled scene.yahrr (SURVEY finding 2).
#include <cerrno>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <array>
#include <iterator>
#include <functional>
#include <limits>
#include <thread>
#include <vector>

#include "host_scene.hpp"

namespace yahr {

// Decimal literal -> binary32, correctly rounded (what GHC's `read :: Float` yields: the literal is read as a
// Rational and converted with fromRational).  Fast path for the common shape -- at most 15 significant digits and a
// decimal exponent within +-22: mantissa and 10^|e| are then exact doubles, so ONE IEEE multiply or divide gives the
// correctly rounded double d of the literal x.  float(d) equals the correctly rounded float of x unless a midpoint
// of two adjacent floats lies between x and d; such a midpoint is itself a double (29 low mantissa bits 1000...0) and
// |x - d| <= ulp(d)/2, so that can only happen when d IS that midpoint (its two neighbours are excluded as well,
// for good measure).  Everything else -- long mantissas, big exponents, results outside the normal float range,
// hexadecimal / octal forms -- goes to strtof, which is correctly rounded too but several times slower.
static bool fast_decimal_f32(const char *s, size_t len, float &out) {
  static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  uint64_t m = 0; int nd = 0, e10 = 0; size_t i = 0;
  for (; i < len && s[i] >= '0' && s[i] <= '9'; ++i) {
    if (nd == 0 && s[i] == '0') continue;              // leading zeros
    if (++nd > 15) return false;
    m = m * 10 + (uint64_t)(s[i] - '0');
  }
  if (i < len && s[i] == '.') {
    for (++i; i < len && s[i] >= '0' && s[i] <= '9'; ++i) {
      --e10;
      if (nd == 0 && s[i] == '0') continue;
      if (++nd > 15) return false;
      m = m * 10 + (uint64_t)(s[i] - '0');
    }
  }
  if (i < len && (s[i] == 'e' || s[i] == 'E')) {
    ++i; bool neg = false;
    if (i < len && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
    int e = 0; size_t digits = 0;
    for (; i < len && s[i] >= '0' && s[i] <= '9'; ++i, ++digits) { if (e < 10000) e = e * 10 + (s[i] - '0'); }
    if (digits == 0) return false;
    e10 += neg ? -e : e;
  }
  if (i != len) return false;                          // not a plain decimal literal
  if (m == 0) { out = 0.0f; return true; }
  if (e10 < -22 || e10 > 22) return false;
  const double d = e10 >= 0 ? (double)m * p10[e10] : (double)m / p10[-e10];
  if (!(d >= 1e-37 && d <= 1e37)) return false;        // keep clear of subnormal floats and overflow
  uint64_t bits; std::memcpy(&bits, &d, sizeof bits);
  const uint64_t low = bits & 0x1FFFFFFFULL;           // the 29 mantissa bits a float does not have
  if (low >= 0x0FFFFFFFULL && low <= 0x10000001ULL) return false;
  out = (float)d;
  return true;
}

float parse_float_literal(const char *s, size_t len) {
  float v;
  if (fast_decimal_f32(s, len, v)) return v;
  std::string z(s, len);
  if (z.size() > 2 && z[0] == '0' && (z[1] == 'o' || z[1] == 'O')) return (float)std::strtoull(z.c_str() + 2, nullptr, 8);
  return std::strtof(z.c_str(), nullptr);              // correctly rounded, like fromRational
}

namespace {

thread_local int g_parallel_lists = 0;

enum class Tok { End, Ident, Number, String, Punct };

struct Token {
  Tok kind = Tok::End;
  size_t pos = 0, len = 0;
  char punct = 0;
  bool integral = false;      // Number: no fraction and no exponent
  std::string str;            // String: decoded
};

class Parser {
 public:
  Parser(const char *t, size_t n, bool lenient, int threads = 1) : text_(t), n_(n), lenient_(lenient), threads_(threads) {}

  std::unique_ptr<HostScene> scene() {
    auto hs = std::make_unique<HostScene>();
    parens(0, 11, [&] {
      expect_ident("Scene"); expect('{');
      field("integrator"); hs->depth = integrator(0); expect(',');
      field("cullingMode"); culling(0, *hs); expect(',');
      field("camera"); hs->camera = camera(0); expect(',');
      field("materials"); list([&] { hs->materials.push_back(material(0)); }); expect(',');
      field("lights"); list([&] { hs->lights.push_back(light(0)); }); expect(',');
      field("objects"); list_into(hs->objects, [](Parser &p, std::vector<SceneObject> &o) { o.push_back(p.object(0)); });
      expect('}');
    });
    if (peek().kind != Tok::End) fail("trailing input after Scene");   // `read` requires a full parse
    return hs;
  }

 private:
  const char *text_; size_t n_; bool lenient_;
  int threads_ = 1;            // > 1: long lists are parsed by that many threads (list_into)
  size_t pos_ = 0;
  bool has_peek_ = false; Token peek_; size_t after_peek_ = 0;

  [[noreturn]] void fail(const std::string &m) { throw ParseError{m, pos_}; }

  static bool ident_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
  static bool ident_char(char c) { return ident_start(c) || (c >= '0' && c <= '9') || c == '\''; }
  static bool digit(char c) { return c >= '0' && c <= '9'; }

  Token lex(size_t &p) {
    while (p < n_ && (text_[p] == ' ' || text_[p] == '\n' || text_[p] == '\t' || text_[p] == '\r' ||
                      text_[p] == '\f' || text_[p] == '\v'))
      ++p;
    Token t; t.pos = p;
    if (p >= n_) { t.kind = Tok::End; return t; }
    char c = text_[p];
    if (ident_start(c)) {
      size_t q = p + 1;
      while (q < n_ && ident_char(text_[q])) ++q;
      t.kind = Tok::Ident; t.len = q - p; p = q; return t;
    }
    if (digit(c)) {
      size_t q = p; t.integral = true;
      if (c == '0' && q + 2 < n_ + 0 && (text_[q + 1] == 'x' || text_[q + 1] == 'X' || text_[q + 1] == 'o' ||
                                         text_[q + 1] == 'O') && q + 2 < n_ && std::isxdigit((unsigned char)text_[q + 2])) {
        q += 2;
        while (q < n_ && std::isxdigit((unsigned char)text_[q])) ++q;
      } else {
        while (q < n_ && digit(text_[q])) ++q;
        if (q + 1 < n_ && text_[q] == '.' && digit(text_[q + 1])) {   // "5." is not a fractional literal
          t.integral = false; ++q;
          while (q < n_ && digit(text_[q])) ++q;
        }
        if (q < n_ && (text_[q] == 'e' || text_[q] == 'E')) {
          size_t r = q + 1;
          if (r < n_ && (text_[r] == '+' || text_[r] == '-')) ++r;
          if (r < n_ && digit(text_[r])) {
            t.integral = false;
            while (r < n_ && digit(text_[r])) ++r;
            q = r;
          }
        }
      }
      t.kind = Tok::Number; t.len = q - p; p = q; return t;
    }
    if (c == '"') {
      size_t q = p + 1; t.kind = Tok::String;
      while (true) {
        if (q >= n_) { pos_ = p; fail("unterminated string literal"); }
        char d = text_[q];
        if (d == '"') { ++q; break; }
        if (d == '\n') { pos_ = q; fail("newline in string literal"); }
        if (d != '\\') { t.str.push_back(d); ++q; continue; }
        ++q;
        if (q >= n_) { pos_ = q; fail("bad escape"); }
        char e = text_[q];
        switch (e) {
          case '\\': t.str.push_back('\\'); ++q; break;
          case '"': t.str.push_back('"'); ++q; break;
          case '\'': t.str.push_back('\''); ++q; break;
          case 'n': t.str.push_back('\n'); ++q; break;
          case 't': t.str.push_back('\t'); ++q; break;
          case 'r': t.str.push_back('\r'); ++q; break;
          case 'a': t.str.push_back('\a'); ++q; break;
          case 'b': t.str.push_back('\b'); ++q; break;
          case 'f': t.str.push_back('\f'); ++q; break;
          case 'v': t.str.push_back('\v'); ++q; break;
          case '&': ++q; break;                                  // empty escape
          default:
            if (digit(e)) {                                      // \DDD decimal code point (ASCII subset)
              unsigned v = 0;
              while (q < n_ && digit(text_[q])) { v = v * 10 + (unsigned)(text_[q] - '0'); ++q; }
              if (v > 0x10FFFF) { pos_ = q; fail("bad numeric escape"); }
              append_utf8(t.str, v);
            } else if (e == 'x' && q + 1 < n_ && std::isxdigit((unsigned char)text_[q + 1])) {
              ++q; unsigned v = 0;
              while (q < n_ && std::isxdigit((unsigned char)text_[q])) {
                char h = text_[q]; v = v * 16 + (unsigned)(digit(h) ? h - '0' : (std::tolower(h) - 'a' + 10)); ++q;
              }
              append_utf8(t.str, v);
            } else if (e == ' ' || e == '\n' || e == '\t' || e == '\r') {   // string gap \  ...  \ //
              while (q < n_ && (text_[q] == ' ' || text_[q] == '\n' || text_[q] == '\t' || text_[q] == '\r')) ++q;
              if (q >= n_ || text_[q] != '\\') { pos_ = q; fail("bad string gap"); }
              ++q;
            } else { pos_ = q; fail("unsupported string escape"); }
        }
      }
      t.len = q - p; p = q; return t;
    }
    if (std::strchr("{}[](),=-", c)) { t.kind = Tok::Punct; t.punct = c; t.len = 1; ++p; return t; }
    pos_ = p; fail(std::string("unexpected character '") + c + "'");
  }

  static void append_utf8(std::string &s, unsigned v) {
    if (v < 0x80) s.push_back((char)v);
    else if (v < 0x800) { s.push_back((char)(0xC0 | (v >> 6))); s.push_back((char)(0x80 | (v & 0x3F))); }
    else if (v < 0x10000) { s.push_back((char)(0xE0 | (v >> 12))); s.push_back((char)(0x80 | ((v >> 6) & 0x3F))); s.push_back((char)(0x80 | (v & 0x3F))); }
    else { s.push_back((char)(0xF0 | (v >> 18))); s.push_back((char)(0x80 | ((v >> 12) & 0x3F))); s.push_back((char)(0x80 | ((v >> 6) & 0x3F))); s.push_back((char)(0x80 | (v & 0x3F))); }
  }

  const Token &peek() {
    if (!has_peek_) { size_t p = pos_; peek_ = lex(p); after_peek_ = p; has_peek_ = true; }
    return peek_;
  }
  Token next() { peek(); Token t = std::move(peek_); pos_ = after_peek_; has_peek_ = false; return t; }
  size_t mark() const { return pos_; }
  void reset(size_t m) { pos_ = m; has_peek_ = false; }

  bool is_punct(char c) { const Token &t = peek(); return t.kind == Tok::Punct && t.punct == c; }
  bool is_ident(const char *s) {
    const Token &t = peek();
    return t.kind == Tok::Ident && t.len == std::strlen(s) && std::memcmp(text_ + t.pos, s, t.len) == 0;
  }
  void expect(char c) {
    if (!is_punct(c)) fail(std::string("expected '") + c + "'");
    next();
  }
  void expect_ident(const char *s) {
    if (!is_ident(s)) fail(std::string("expected ") + s);
    next();
  }
  void field(const char *name) { expect_ident(name); expect('='); }

  // GHC.Read `parens` + `prec`: optional redundant parentheses; outside them the body needs
  // current precedence <= body_prec (10 = constructor application, 11 = record/atom).
  template <class F> void parens(int prec, int body_prec, F body) {
    if (is_punct('(')) { next(); parens(0, body_prec, body); expect(')'); return; }
    if (prec > body_prec) fail("constructor application needs parentheses here");
    body();
  }

  template <class F> void list(F elem) {          // GHC.Read.list, elements at precedence 0
    parens(0, 11, [&] {
      expect('[');
      if (is_punct(']')) { next(); return; }
      while (true) {
        elem();
        if (is_punct(',')) { next(); continue; }
        expect(']'); break;
      }
    });
  }

  // ---- long lists in parallel --------------------------------------------------------------------------
  // A scene of a million triangles is one list of a million records (or three lists of points, normals and index
  // triples): 150 MB of text that a single thread reads at 170 MB/s.  list_into() parses such a list with several
  // threads: ONE pass over its bytes finds the closing bracket and the commas between its elements (bracket depth,
  // string literals skipped -- nothing else in this grammar can hide a bracket or a comma), the elements are dealt
  // to the threads in contiguous runs of about equal size, every thread runs an ordinary Parser over its run, and
  // the runs are appended in order.  Whatever goes wrong in there (a syntax error, brackets that do not balance)
  // makes the list fall back to the sequential code, so the error reported -- message and offset -- is the
  // sequential parser's.
  static constexpr size_t kParallelMinBytes = (size_t)1 << 20;
  static size_t parallel_min_bytes() {
    const char *e = std::getenv("YAHR_PARSE_PARALLEL_MIN");          // tests: force the parallel path on small lists
    return e ? (size_t)std::strtoull(e, nullptr, 10) : kParallelMinBytes;
  }

  // `from`: just behind the list's '[' or behind one of its commas.  Finds the list's ']' and the commas at its depth;
  // false if the brackets do not balance
  bool scan_list(size_t from, size_t &close, std::vector<size_t> &commas) const {
    static const auto cls = [] {
      std::array<uint8_t, 256> t{};
      t[(unsigned char)'('] = t[(unsigned char)'['] = t[(unsigned char)'{'] = 1;
      t[(unsigned char)')'] = t[(unsigned char)']'] = t[(unsigned char)'}'] = 2;
      t[(unsigned char)','] = 3; t[(unsigned char)'"'] = 4;
      return t;
    }();
    int depth = 1;
    size_t p = from;
    while (p < n_) {
      const uint8_t k = cls[(unsigned char)text_[p]];
      if (k == 0) { ++p; continue; }
      if (k == 1) { ++depth; ++p; continue; }
      if (k == 2) {
        if (--depth == 0) { close = p; return text_[p] == ']'; }
        ++p; continue;
      }
      if (k == 3) { if (depth == 1) commas.push_back(p); ++p; continue; }
      // string literal, with lex()'s own rules: to the closing quote; a backslash takes the next character with it
      // (\" and \\; the digits of a numeric escape are ordinary characters here), except that backslash + white
      // space opens a string gap that runs to the next backslash
      ++p;
      while (p < n_ && text_[p] != '"') {
        if (text_[p] != '\\') { ++p; continue; }
        if (++p >= n_) return false;
        const char e = text_[p];
        if (e == ' ' || e == '\n' || e == '\t' || e == '\r') {
          while (p < n_ && (text_[p] == ' ' || text_[p] == '\n' || text_[p] == '\t' || text_[p] == '\r')) ++p;
          if (p >= n_ || text_[p] != '\\') return false;
        }
        ++p;
      }
      if (p >= n_) return false;
      ++p;
    }
    return false;
  }

  // The same scan by several threads, for lists too long to walk twice with one: the text is cut into chunks (never
  // right behind a backslash).  Pass 1 walks every chunk under both hypotheses -- it starts outside / inside a string
  // literal -- and keeps the state it ends in, the net bracket depth and the lowest depth reached; chaining the chunks
  // gives every chunk's true start state and depth and tells which chunk holds the closing bracket.  Pass 2 walks the
  // chunks again with that knowledge and records the commas at depth 1.  Chunks are taken in waves of one per thread
  // until the closing bracket is found, so the work follows the length of the list, not of the file.  A backslash
  // followed by white space (a string gap: the one construct whose meaning depends on more than one preceding
  // character) makes the whole thing return false; the caller then uses the sequential scan_list.
  struct ChunkEnd { bool in_str; int delta, low; bool gap; };
  // walks [b, e); depth counts from 0 at b; commas != nullptr: records commas at depth == want and stops at the
  // bracket that takes the depth below `floor_depth` (returns its position in *stop)
  ChunkEnd walk_chunk(size_t b, size_t e, bool in_str, std::vector<size_t> *commas, int want, int floor_depth, size_t *stop) const {
    // per byte outside a string: a table lookup for the bracket depth and one rarely taken branch for comma, quote and backslash
    static const auto tab = [] {
      std::array<int8_t, 256> d{}; std::array<uint8_t, 256> sp{};
      d[(unsigned char)'('] = d[(unsigned char)'['] = d[(unsigned char)'{'] = 1;
      d[(unsigned char)')'] = d[(unsigned char)']'] = d[(unsigned char)'}'] = -1;
      sp[(unsigned char)','] = sp[(unsigned char)'"'] = sp[(unsigned char)'\\'] = 1;
      return std::make_pair(d, sp);
    }();
    const auto &delta = tab.first; const auto &special = tab.second;
    ChunkEnd r{in_str, 0, 0, false};
    size_t p = b;
    int depth = 0;
    while (p < e) {
      if (r.in_str) {
        const char c = text_[p];
        if (c == '"') { r.in_str = false; ++p; continue; }
        if (c == '\\') {
          const char d = p + 1 < n_ ? text_[p + 1] : ' ';
          if (d == ' ' || d == '\n' || d == '\t' || d == '\r') { r.gap = true; return r; }
          p += 2; continue;                      // (never across e: no chunk starts right behind a backslash)
        }
        ++p; continue;
      }
      while (p < e) {
        const unsigned char c = (unsigned char)text_[p];
        depth += delta[c];
        if (depth < r.low) {
          r.low = depth;
          if (commas && depth < floor_depth) { *stop = p; r.delta = depth; return r; }
        }
        if (special[c]) {
          if (c == ',') { if (commas && depth == want) commas->push_back(p); }
          else if (c == '"') { r.in_str = true; ++p; break; }
          else { r.gap = true; return r; }      // a backslash outside a string: a lexical error, the sequential code reports it
        }
        ++p;
      }
    }
    r.delta = depth;
    return r;
  }

  enum class Scan { Done, TooFewElements, NeedSequential };
  Scan scan_list_parallel(size_t from, size_t &close, std::vector<size_t> &commas) const {
    const int T = threads_;
    size_t at = from;                 // start of the next wave
    bool in_str = false;              // state and depth (1 = inside the list) at `at`
    int depth = 1;
    std::vector<std::vector<size_t>> found;
    size_t n_found = 0;
    for (int wave = 0; at < n_; ++wave) {
      const size_t chunk = (size_t)1 << (20 + std::min(wave, 3));     // 1, 2, 4, 8, 8, ... MiB: few thread starts
      std::vector<size_t> cut(1, at);
      for (int r = 0; r < T && cut.back() < n_; ++r) {
        size_t b = std::min(n_, cut.back() + chunk);
        while (b < n_ && text_[b - 1] == '\\') ++b;
        cut.push_back(b);
      }
      const int C = (int)cut.size() - 1;
      std::vector<ChunkEnd> out_h(C), in_h(C);
      auto pass1 = [&](int r) {
        out_h[r] = walk_chunk(cut[r], cut[r + 1], false, nullptr, 0, 0, nullptr);
        // started inside a string: without a quote or a backslash in the chunk (the rule) it never leaves it
        const char *c0 = text_ + cut[r]; const size_t len = cut[r + 1] - cut[r];
        if (!std::memchr(c0, '"', len) && !std::memchr(c0, '\\', len)) in_h[r] = ChunkEnd{true, 0, 0, false};
        else in_h[r] = walk_chunk(cut[r], cut[r + 1], true, nullptr, 0, 0, nullptr);
      };
      {
        std::vector<std::thread> pool;
        for (int r = 1; r < C; ++r) pool.emplace_back(pass1, r);
        pass1(0);
        for (auto &t : pool) t.join();
      }
      // chain: start state / depth of every chunk, and the first chunk in which the depth reaches 0
      std::vector<char> s_in(C); std::vector<int> s_depth(C);
      int closing = -1;
      for (int r = 0; r < C; ++r) {
        s_in[r] = in_str; s_depth[r] = depth;
        const ChunkEnd &h = in_str ? in_h[r] : out_h[r];
        if (h.gap) return Scan::NeedSequential;
        if (depth + h.low <= 0) { closing = r; break; }
        in_str = h.in_str; depth += h.delta;
      }
      const int last = closing >= 0 ? closing : C - 1;
      std::vector<std::vector<size_t>> part(last + 1);
      std::vector<size_t> stop(last + 1, (size_t)-1);
      auto pass2 = [&](int r) {
        // depth counts from 0 inside walk_chunk: the list's own depth is 1 - s_depth[r] there
        walk_chunk(cut[r], cut[r + 1], s_in[r] != 0, &part[r], 1 - s_depth[r], 1 - s_depth[r], &stop[r]);
      };
      {
        std::vector<std::thread> pool;
        for (int r = 1; r <= last; ++r) pool.emplace_back(pass2, r);
        pass2(0);
        for (auto &t : pool) t.join();
      }
      for (int r = 0; r <= last; ++r) { n_found += part[r].size(); found.push_back(std::move(part[r])); }
      if (closing >= 0) {
        if (stop[closing] == (size_t)-1 || text_[stop[closing]] != ']') return Scan::NeedSequential;
        close = stop[closing];
        commas.reserve(n_found);
        for (auto &v : found) commas.insert(commas.end(), v.begin(), v.end());
        return Scan::Done;
      }
      // elements of several MiB each: the long lists are inside them, and those get their own threads
      if (n_found < (size_t)(wave + 1) * (size_t)T / 4) return Scan::TooFewElements;
      at = cut[C];
    }
    return Scan::NeedSequential;
  }

  // Out: a std::vector; elem(Parser &, Out &) parses one element and appends what it yields
  template <class Out, class F> bool parallel_list(size_t from, Out &out, F elem, size_t per_elem) {
    size_t close = 0;
    std::vector<size_t> commas;
    const Scan sc = scan_list_parallel(from, close, commas);
    if (sc == Scan::TooFewElements) return false;
    if (sc == Scan::NeedSequential) {
      commas.clear();
      if (!scan_list(from, close, commas)) return false;
    }
    if (close - from < parallel_min_bytes() || commas.size() + 1 < (size_t)threads_ * 4) return false;
    const size_t n_elem = commas.size() + 1;
    const int T = threads_;
    // run r = elements [first[r], first[r + 1]): cut at the comma nearest to r/T of the bytes
    std::vector<size_t> first(T + 1, 0);
    first[T] = n_elem;
    for (int r = 1; r < T; ++r) {
      const size_t target = from + (close - from) / (size_t)T * (size_t)r;
      const size_t k = (size_t)(std::lower_bound(commas.begin(), commas.end(), target) - commas.begin());
      first[r] = std::max(first[r - 1], std::min(k + 1, n_elem));
    }
    std::vector<Out> parts(T);
    std::vector<char> ok(T, 1);
    auto work = [&](int r) {
      const size_t e0 = first[r], e1 = first[r + 1];
      if (e0 >= e1) return;
      const size_t begin = e0 == 0 ? from : commas[e0 - 1] + 1;
      const size_t end = e1 == n_elem ? close : commas[e1 - 1];
      try {
        Parser sub(text_, end, lenient_, 1);
        sub.pos_ = begin;
        Out mine;                                        // (not parts[r] directly: the vector headers share cache lines)
        mine.reserve((e1 - e0) * per_elem);
        for (size_t e = e0; e < e1; ++e) {
          elem(sub, mine);
          if (e + 1 < e1) sub.expect(',');
        }
        if (sub.peek().kind != Tok::End) ok[r] = 0;
        parts[r] = std::move(mine);
      } catch (const ParseError &) {
        ok[r] = 0;
      } catch (...) {
        ok[r] = 0;
      }
    };
    std::vector<std::thread> pool;
    for (int r = 1; r < T; ++r) pool.emplace_back(work, r);
    work(0);
    for (auto &t : pool) t.join();
    for (int r = 0; r < T; ++r) if (!ok[r]) return false;
    size_t total = out.size();
    for (int r = 0; r < T; ++r) total += parts[r].size();
    out.reserve(total);
    for (int r = 0; r < T; ++r) out.insert(out.end(), std::make_move_iterator(parts[r].begin()), std::make_move_iterator(parts[r].end()));
    pos_ = close + 1; has_peek_ = false;
    ++g_parallel_lists;
    return true;
  }

  template <class Out, class F> void list_into(Out &out, F elem, size_t per_elem = 1) {     // GHC.Read.list, elements at precedence 0;
                                                                                              // per_elem: entries of `out` per element
    parens(0, 11, [&] {
      expect('[');
      if (is_punct(']')) { next(); return; }
      bool try_threads = threads_ > 1;
      while (true) {
        const size_t e0 = peek().pos;
        if (try_threads) {                             // the rest of the list, from this element on, by several threads?
          try_threads = false;
          if (n_ - e0 >= parallel_min_bytes() && parallel_list(e0, out, elem, per_elem)) return;
        }
        elem(*this, out);
        // after an element of megabytes (a mesh) a long run of small ones may still follow: look again
        if (threads_ > 1 && pos_ - e0 >= parallel_min_bytes()) try_threads = true;
        if (is_punct(',')) { next(); continue; }
        expect(']'); break;
      }
    });
  }

  // readNumber: optional parens, optional '-', then a numeric lexeme (or NaN / Infinity for Float)
  float number_f() {
    float out = 0;
    parens(0, 11, [&] {
      bool neg = false;
      if (is_punct('-')) { next(); neg = true; }
      Token t = next();
      float v;
      if (t.kind == Tok::Number) {
        v = parse_float_literal(text_ + t.pos, t.len);
      } else if (t.kind == Tok::Ident && t.len == 3 && !std::memcmp(text_ + t.pos, "NaN", 3)) {
        v = std::numeric_limits<float>::quiet_NaN();
      } else if (t.kind == Tok::Ident && t.len == 8 && !std::memcmp(text_ + t.pos, "Infinity", 8)) {
        v = std::numeric_limits<float>::infinity();
      } else { pos_ = t.pos; fail("expected a number"); }
      out = neg ? -v : v;
    });
    return out;
  }
  int32_t number_i() {
    int32_t out = 0;
    parens(0, 11, [&] {
      bool neg = false;
      if (is_punct('-')) { next(); neg = true; }
      Token t = next();
      if (t.kind != Tok::Number || !t.integral) { pos_ = t.pos; fail("expected an integer"); }
      std::string s(text_ + t.pos, t.len);
      long long v;
      if (s.size() > 2 && s[0] == '0' && (s[1] == 'o' || s[1] == 'O')) v = std::strtoll(s.c_str() + 2, nullptr, 8);
      else v = std::strtoll(s.c_str(), nullptr, 0);
      out = (int32_t)(neg ? -v : v);
    });
    return out;
  }
  std::string string_lit() {
    std::string out;
    parens(0, 11, [&] {
      Token t = next();
      if (t.kind != Tok::String) { pos_ = t.pos; fail("expected a string literal"); }
      out = std::move(t.str);
    });
    return out;
  }
  bool boolean() {
    bool out = false;
    parens(0, 11, [&] {
      if (is_ident("True")) { next(); out = true; }
      else if (is_ident("False")) { next(); out = false; }
      else fail("expected True or False");
    });
    return out;
  }

  Vec3 vec3(int prec) {                             // data Vec3 = Vec3 !Float !Float !Float  (Vectors.hs:5)
    Vec3 v;
    parens(prec, 10, [&] {
      expect_ident("Vec3");
      v.x = number_f(); v.y = number_f(); v.z = number_f();
    });
    return v;
  }

  int integrator(int prec) {                        // Integrators.hs:18-20
    int d = 0;
    parens(prec, 11, [&] {
      expect_ident("WhittedIntegrator"); expect('{');
      field("recursionDepth"); d = number_i();
      expect('}');
    });
    return d;
  }

  void culling(int prec, HostScene &hs) {           // Culling.hs:18-19
    parens(prec, 10, [&] {
      expect_ident("BVH");
      hs.bvh_depth = number_i();
      parens(11, 11, [&] {
        if (is_ident("Midpoint")) { next(); hs.split = YAHR_SPLIT_MIDPOINT; }
        else if (is_ident("SurfaceAreaHeuristic")) { next(); hs.split = YAHR_SPLIT_SAH; }
        else fail("expected Midpoint or SurfaceAreaHeuristic");
      });
    });
  }

  Camera camera(int prec) {                         // Cameras.hs:54-56
    Camera c;
    parens(prec, 11, [&] {
      expect_ident("Camera"); expect('{');
      field("imW"); c.imW = number_f(); expect(',');
      field("imH"); c.imH = number_f(); expect(',');
      field("focalLength"); c.focal = number_f(); expect(',');
      field("lookDir"); c.look = vec3(0); expect(',');
      field("upDir"); c.up = vec3(0); expect(',');
      field("position"); c.pos = vec3(0);
      expect('}');
    });
    return c;
  }

  Material material(int prec) {                     // Scene.hs:45-50
    Material m;
    parens(prec, 11, [&] {
      expect_ident("BlinnPhongMaterial"); expect('{');
      field("id"); m.id = string_lit(); expect(',');
      field("ambient"); m.ambient = vec3(0); expect(',');
      field("diffuse"); m.diffuse = vec3(0); expect(',');
      field("specular"); m.specular = vec3(0); expect(',');
      field("shininess"); m.shininess = number_f();
      expect('}');
    });
    return m;
  }

  Light light(int prec) {                           // Lights.hs:7
    Light l;
    parens(prec, 10, [&] {
      expect_ident("PointLight");
      l.pos = vec3(11);
      bool one_arg = is_punct(',') || is_punct(']') || is_punct(')');
      if (one_arg && lenient_) {
        // stale dialect (scene.yahrr:34): no spectrum.  The value the old code used is unknown; (100,100,100)
        // is this build's documented guess (DESIGN.md "scene.yahrr repair").
        l.spectrum = Vec3{100.f, 100.f, 100.f};
      } else {
        l.spectrum = vec3(11);
      }
    });
    return l;
  }

  template <class F> void maybe(F just) {           // Maybe a: Nothing | Just a (argument at precedence 11)
    bool done = false;
    // `Nothing` is nullary (any precedence); `Just x` is an application (precedence 10) -- field values
    // are read at precedence 0, so both are legal bare.
    parens(0, 10, [&] {
      if (is_ident("Nothing")) { next(); done = true; return; }
      expect_ident("Just");
      just();
    });
    (void)done;
  }

  void tri_tuple(std::vector<int32_t> &out) {       // (Int, Int, Int), possibly inside redundant parens
    size_t m = mark();
    expect('(');
    try {
      int32_t a = number_i(); expect(',');
      int32_t b = number_i(); expect(',');
      int32_t c = number_i(); expect(')');
      out.push_back(a); out.push_back(b); out.push_back(c);
      return;
    } catch (const ParseError &) {
      reset(m);
    }
    expect('('); tri_tuple(out); expect(')');       // redundant parentheses around the whole tuple
  }

  SceneObject object(int prec) {                    // Scene.hs:15-42
    SceneObject o;
    parens(prec, 11, [&] {
      if (is_ident("Sphere")) {
        next(); expect('{'); o.kind = SceneObject::Sphere;
        field("position"); o.position = vec3(0); expect(',');
        field("radius"); o.radius = number_f(); expect(',');
        field("materialId"); o.material_id = string_lit();
        expect('}');
      } else if (is_ident("Triangle")) {
        next(); expect('{'); o.kind = SceneObject::Triangle;
        field("p0"); o.p0 = vec3(0); expect(',');
        field("p1"); o.p1 = vec3(0); expect(',');
        field("p2"); o.p2 = vec3(0); expect(',');
        field("n0"); o.n0 = vec3(0); expect(',');
        field("n1"); o.n1 = vec3(0); expect(',');
        field("n2"); o.n2 = vec3(0); expect(',');
        field("materialId"); o.material_id = string_lit();
        expect('}');
      } else if (is_ident("TriangleMesh")) {
        next(); expect('{'); o.kind = SceneObject::TriangleMesh;
        field("triangleMeshPoints"); list_into(o.points, [](Parser &p, std::vector<Vec3> &v) { v.push_back(p.vec3(0)); }); expect(',');
        if (!lenient_ || is_ident("triangleMeshNormals")) {
          field("triangleMeshNormals");
          maybe([&] { o.has_normals = true; parens(11, 11, [&] { list_into(o.normals, [](Parser &p, std::vector<Vec3> &v) { v.push_back(p.vec3(0)); }); }); });
          expect(',');
        }
        field("triangleMeshTriangles"); list_into(o.tris, [](Parser &p, std::vector<int32_t> &v) { p.tri_tuple(v); }, 3); expect(',');
        if (!lenient_ || is_ident("triangleMeshSmooth")) {
          field("triangleMeshSmooth");
          maybe([&] { o.has_smooth = true; parens(11, 11, [&] { list_into(o.smooth, [](Parser &p, std::vector<uint8_t> &v) { v.push_back(p.boolean() ? 1 : 0); }); }); });
          expect(',');
        }
        field("materialId"); o.material_id = string_lit();
        expect('}');
      } else if (is_ident("Subsampled")) {
        next(); expect('{'); o.kind = SceneObject::Subsampled;
        field("subsampleSize"); o.subsample_size = number_f(); expect(',');
        field("subsampledObjects"); list_into(o.children, [](Parser &p, std::vector<SceneObject> &c) { c.push_back(p.object(0)); });
        expect('}');
      } else {
        fail("expected Sphere, Triangle, TriangleMesh or Subsampled");
      }
    });
    return o;
  }
};

}  // namespace

int parse_parallel_lists() { return g_parallel_lists; }

std::unique_ptr<HostScene> parse_scene(const char *text, size_t len, bool lenient) {
  int threads = 1;
  if (len >= ((size_t)1 << 20) || std::getenv("YAHR_PARSE_PARALLEL_MIN")) {
    const char *e = std::getenv("YAHR_PARSE_THREADS");
    threads = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
    if (threads > 16) threads = 16;
    if (threads < 1) threads = 1;
  }
  g_parallel_lists = 0;
  Parser p(text, len, lenient, threads);
  return p.scene();
}

}  // namespace yahr
