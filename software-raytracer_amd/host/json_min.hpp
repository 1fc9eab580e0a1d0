// json_min.hpp — a small JSON DOM for the scene files.
//
// The reference reads and writes scenes through nlohmann/json 3.11.2 (vendored there,
// Raytracer/json.hpp); that library is not part of this repo.  This is an independent,
// minimal implementation of the two behaviours the scene format depends on:
//   * parse(): RFC 8259 JSON -> DOM; numbers keep "integer vs float" like nlohmann does
//     (an integer literal converts to float exactly the way `float f = json` does).
//   * dump(indent): object keys in sorted (std::map) order, `indent` spaces per level,
//     floats printed as the shortest decimal that round-trips the double, in nlohmann's
//     layout ("1.0", "0.20000000298023224", "1e-05"), empty array "[]", empty object "{}".
#pragma once

#include <charconv>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace srt_host {

class JsonError : public std::runtime_error {
   public:
    explicit JsonError(const std::string& m) : std::runtime_error(m) {}
};


// ---- Grisu2 (Loitsch, PLDI 2010): shortest-or-nearly-shortest digits of a double --------
// The reference writes scenes with nlohmann/json 3.11.2, whose number writer is Grisu2 with
// alpha = -60, gamma = -32 and cached powers 10^k, k = -300 + 8j.  To reproduce its files byte
// for byte we need the same digit CHOICE where several 17-digit strings round-trip — a correctly
// rounding printer would differ — so this block is a condensed RE-EXPRESSION of that library's
// `dtoa_impl` (Raytracer/json.hpp:16865-17588, itself Loitsch's Grisu2): the same DiyFp multiply
// (32-bit limbs p0..p3), the same k-from-exponent estimate (e * 78913 >> 18), boundary computation,
// digit generation and grisu2_round condition, because byte identity forces those choices.  It is not
// an independent design.  What is this project's own: the cached-power table is generated with big-
// integer arithmetic by host/tools/gen_grisu_table.py (grisu_powers.inc), and the output formatting
// below.  Differentially tested against the real json.hpp (tests/test_json_vs_reference.py).
namespace grisu {

struct DiyFp {
    uint64_t f;
    int e;
};
struct CachedPower {
    uint64_t f;
    int e;
    int k;
};
static const CachedPower kPowers[] = {
#include "grisu_powers.inc"
};

inline DiyFp mul(DiyFp x, DiyFp y) {  // upper 64 bits of the 128-bit product, rounded
    const uint64_t u_lo = x.f & 0xFFFFFFFFu, u_hi = x.f >> 32, v_lo = y.f & 0xFFFFFFFFu, v_hi = y.f >> 32;
    const uint64_t p0 = u_lo * v_lo, p1 = u_lo * v_hi, p2 = u_hi * v_lo, p3 = u_hi * v_hi;
    uint64_t q = (p0 >> 32) + (p1 & 0xFFFFFFFFu) + (p2 & 0xFFFFFFFFu);
    q += uint64_t{1} << 31;
    return {p3 + (p2 >> 32) + (p1 >> 32) + (q >> 32), x.e + y.e + 64};
}
inline DiyFp normalize(DiyFp x) {
    while ((x.f >> 63) == 0) {
        x.f <<= 1;
        x.e--;
    }
    return x;
}
inline DiyFp normalize_to(DiyFp x, int e) { return {x.f << (x.e - e), e}; }

// digits of v (> 0, finite) into buf; value = digits * 10^decimal_exponent
inline void shortest(double value, char* buf, int& len, int& decimal_exponent) {
    uint64_t bits;
    std::memcpy(&bits, &value, 8);
    const uint64_t F = bits & ((uint64_t{1} << 52) - 1), E = bits >> 52;
    const int kBias = 1075, kMinExp = 1 - kBias;
    const uint64_t hidden = uint64_t{1} << 52;
    const DiyFp v = E == 0 ? DiyFp{F, kMinExp} : DiyFp{F + hidden, (int)E - kBias};
    // boundaries m- and m+ (half-way to the neighbouring doubles)
    const bool lower_closer = F == 0 && E > 1;
    const DiyFp m_plus{2 * v.f + 1, v.e - 1};
    const DiyFp m_minus = lower_closer ? DiyFp{4 * v.f - 1, v.e - 2} : DiyFp{2 * v.f - 1, v.e - 1};
    const DiyFp w_plus = normalize(m_plus);
    const DiyFp w_minus = normalize_to(m_minus, w_plus.e);
    const DiyFp w_v = normalize(v);
    // cached power c = 10^-k with alpha <= e_c + e + 64 <= gamma  (alpha = -60, gamma = -32)
    const int e = w_plus.e;
    const int f = -60 - e - 1;
    const int k = (f * 78913) / (1 << 18) + (f > 0 ? 1 : 0);
    const int index = (300 + k + 7) / 8;
    const CachedPower cp = kPowers[index];
    const DiyFp c{cp.f, cp.e};
    const DiyFp w = mul(w_v, c), wm = mul(w_minus, c), wp = mul(w_plus, c);
    const DiyFp M_minus{wm.f + 1, wm.e}, M_plus{wp.f - 1, wp.e};  // safe interval
    decimal_exponent = -cp.k;
    // digit generation
    uint64_t delta = M_plus.f - M_minus.f, dist = M_plus.f - w.f;
    const DiyFp one{uint64_t{1} << -M_plus.e, M_plus.e};
    uint32_t p1 = (uint32_t)(M_plus.f >> -one.e);
    uint64_t p2 = M_plus.f & (one.f - 1);
    uint32_t pow10 = 1;
    int n = 1;
    {
        static const uint32_t tens[] = {1000000000u, 100000000u, 10000000u, 1000000u, 100000u, 10000u, 1000u, 100u, 10u, 1u};
        for (int i = 0; i < 10; ++i)
            if (p1 >= tens[i]) {
                pow10 = tens[i];
                n = 10 - i;
                break;
            }
    }
    len = 0;
    auto round_weed = [&](uint64_t rest, uint64_t ten_k) {
        while (rest < dist && delta - rest >= ten_k && (rest + ten_k < dist || dist - rest > rest + ten_k - dist)) {
            buf[len - 1]--;
            rest += ten_k;
        }
    };
    while (n > 0) {
        const uint32_t d = p1 / pow10, r = p1 % pow10;
        buf[len++] = (char)('0' + d);
        p1 = r;
        n--;
        const uint64_t rest = (uint64_t{p1} << -one.e) + p2;
        if (rest <= delta) {
            decimal_exponent += n;
            round_weed(rest, uint64_t{pow10} << -one.e);
            return;
        }
        pow10 /= 10;
    }
    int m = 0;
    for (;;) {
        p2 *= 10;
        const uint64_t d = p2 >> -one.e, r = p2 & (one.f - 1);
        buf[len++] = (char)('0' + d);
        p2 = r;
        m++;
        delta *= 10;
        dist *= 10;
        if (p2 <= delta) break;
    }
    decimal_exponent -= m;
    round_weed(p2, one.f);
}

}  // namespace grisu


class Json {
   public:
    enum class Kind { Null, Bool, Int, Uint, Float, String, Array, Object };
    using Array = std::vector<Json>;
    using Object = std::map<std::string, Json>;

    Json() = default;
    Json(std::nullptr_t) {}
    Json(bool b) : kind_(Kind::Bool), b_(b) {}
    Json(int v) : kind_(Kind::Int), i_(v) {}
    Json(int64_t v) : kind_(Kind::Int), i_(v) {}
    Json(uint64_t v) : kind_(Kind::Uint), u_(v) {}
    Json(double v) : kind_(Kind::Float), d_(v) {}
    Json(float v) : kind_(Kind::Float), d_((double)v) {}  // float -> double, as nlohmann stores it
    Json(const char* s) : kind_(Kind::String), s_(s) {}
    Json(const std::string& s) : kind_(Kind::String), s_(s) {}
    static Json array() {
        Json j;
        j.kind_ = Kind::Array;
        return j;
    }
    static Json array(std::initializer_list<Json> v) {
        Json j = array();
        j.a_ = v;
        return j;
    }
    static Json object() {
        Json j;
        j.kind_ = Kind::Object;
        return j;
    }

    Kind kind() const { return kind_; }
    bool is_null() const { return kind_ == Kind::Null; }
    bool is_number() const { return kind_ == Kind::Int || kind_ == Kind::Uint || kind_ == Kind::Float; }
    bool is_string() const { return kind_ == Kind::String; }
    bool is_array() const { return kind_ == Kind::Array; }
    bool is_object() const { return kind_ == Kind::Object; }

    // ---- typed access; wrong type throws like nlohmann's type_error ------------------
    float as_float() const {
        switch (kind_) {
            case Kind::Float: return (float)d_;
            case Kind::Int: return (float)i_;
            case Kind::Uint: return (float)u_;
            case Kind::Bool: return b_ ? 1.0f : 0.0f;  // nlohmann converts booleans to arithmetic types
            default: throw JsonError(std::string("type must be number, but is ") + type_name());
        }
    }
    double as_double() const {
        switch (kind_) {
            case Kind::Float: return d_;
            case Kind::Int: return (double)i_;
            case Kind::Uint: return (double)u_;
            case Kind::Bool: return b_ ? 1.0 : 0.0;
            default: throw JsonError(std::string("type must be number, but is ") + type_name());
        }
    }
    const std::string& as_string() const {
        if (kind_ != Kind::String) throw JsonError(std::string("type must be string, but is ") + type_name());
        return s_;
    }
    const Array& items() const {
        if (kind_ != Kind::Array) throw JsonError(std::string("type must be array, but is ") + type_name());
        return a_;
    }
    const Object& members() const {
        if (kind_ != Kind::Object) throw JsonError(std::string("type must be object, but is ") + type_name());
        return o_;
    }

    bool contains(const std::string& key) const { return kind_ == Kind::Object && o_.count(key) != 0; }

    // value["key"] on a non-const nlohmann json inserts null for a missing key and turns a
    // null value into an object; on any other non-object it throws.
    Json& operator[](const std::string& key) {
        if (kind_ == Kind::Null) kind_ = Kind::Object;
        if (kind_ != Kind::Object)
            throw JsonError(std::string("cannot use operator[] with a string argument with ") + type_name());
        return o_[key];
    }
    // value[i] on an array; nlohmann fills with nulls when i >= size (null becomes array).
    Json& operator[](size_t i) {
        if (kind_ == Kind::Null) kind_ = Kind::Array;
        if (kind_ != Kind::Array)
            throw JsonError(std::string("cannot use operator[] with a numeric argument with ") + type_name());
        if (i >= a_.size()) a_.resize(i + 1);
        return a_[i];
    }
    void push_back(const Json& v) {
        if (kind_ == Kind::Null) kind_ = Kind::Array;
        if (kind_ != Kind::Array) throw JsonError("cannot use push_back() with " + std::string(type_name()));
        a_.push_back(v);
    }
    size_t size() const { return kind_ == Kind::Array ? a_.size() : kind_ == Kind::Object ? o_.size() : kind_ == Kind::Null ? 0 : 1; }

    const char* type_name() const {
        switch (kind_) {
            case Kind::Null: return "null";
            case Kind::Bool: return "boolean";
            case Kind::String: return "string";
            case Kind::Array: return "array";
            case Kind::Object: return "object";
            default: return "number";
        }
    }

    // ---- parse ---------------------------------------------------------------------------
    static Json parse(const std::string& text) {
        Parser p{text.data(), text.data() + text.size()};
        // nlohmann skips a UTF-8 byte order mark
        if (text.size() >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF)
            p.cur += 3;
        Json v = p.value(0);
        p.ws();
        if (p.cur != p.end) p.fail("unexpected trailing characters");
        return v;
    }

    // ---- dump ----------------------------------------------------------------------------
    std::string dump(int indent = -1) const {
        std::string out;
        dump_to(out, indent, 0);
        return out;
    }

    // Grisu2 digits of a double in nlohmann's layout (min_exp = -4, max_exp = 15)
    static std::string format_double(double v) {
        if (!std::isfinite(v)) return "null";  // nlohmann dumps NaN/inf as null
        if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
        char digits[32];
        int k = 0, dexp = 0;
        grisu::shortest(std::fabs(v), digits, k, dexp);
        std::string ds(digits, (size_t)k);
        const int n = k + dexp;  // position of the decimal point relative to the digits
        std::string out = v < 0 ? "-" : "";
        const int min_exp = -4, max_exp = 15;
        if (k <= n && n <= max_exp) {  // digits[000].0
            out += ds;
            out.append((size_t)(n - k), '0');
            out += ".0";
        } else if (0 < n && n <= max_exp) {  // dig.its
            out += ds.substr(0, (size_t)n);
            out += ".";
            out += ds.substr((size_t)n);
        } else if (min_exp < n && n <= 0) {  // 0.[000]digits
            out += "0.";
            out.append((size_t)(-n), '0');
            out += ds;
        } else {  // d[.igits]e+-XX
            out += ds.substr(0, 1);
            if (k > 1) {
                out += ".";
                out += ds.substr(1);
            }
            out += "e";
            int e = n - 1;
            out += e < 0 ? "-" : "+";
            e = e < 0 ? -e : e;
            char buf[8];
            std::snprintf(buf, sizeof buf, e < 10 ? "0%d" : "%d", e);
            out += buf;
        }
        return out;
    }

   private:
    Kind kind_ = Kind::Null;
    bool b_ = false;
    int64_t i_ = 0;
    uint64_t u_ = 0;
    double d_ = 0;
    std::string s_;
    Array a_;
    Object o_;

    static void dump_string(std::string& out, const std::string& s) {
        out.push_back('"');
        for (unsigned char c : s) {
            switch (c) {
                case '"': out += "\\\""; break;
                case '\\': out += "\\\\"; break;
                case '\b': out += "\\b"; break;
                case '\f': out += "\\f"; break;
                case '\n': out += "\\n"; break;
                case '\r': out += "\\r"; break;
                case '\t': out += "\\t"; break;
                default:
                    if (c < 0x20) {
                        char buf[8];
                        std::snprintf(buf, sizeof buf, "\\u%04x", c);
                        out += buf;
                    } else {
                        out.push_back((char)c);
                    }
            }
        }
        out.push_back('"');
    }

    void dump_to(std::string& out, int indent, int level) const {
        const bool pretty = indent >= 0;
        auto nl = [&](int lvl) {
            if (pretty) {
                out.push_back('\n');
                out.append((size_t)(lvl * indent), ' ');
            }
        };
        switch (kind_) {
            case Kind::Null: out += "null"; break;
            case Kind::Bool: out += b_ ? "true" : "false"; break;
            case Kind::Int: out += std::to_string(i_); break;
            case Kind::Uint: out += std::to_string(u_); break;
            case Kind::Float: out += format_double(d_); break;
            case Kind::String: dump_string(out, s_); break;
            case Kind::Array:
                if (a_.empty()) {
                    out += "[]";
                    break;
                }
                out.push_back('[');
                for (size_t i = 0; i < a_.size(); ++i) {
                    nl(level + 1);
                    a_[i].dump_to(out, indent, level + 1);
                    if (i + 1 < a_.size()) out.push_back(',');
                }
                nl(level);
                out.push_back(']');
                break;
            case Kind::Object: {
                if (o_.empty()) {
                    out += "{}";
                    break;
                }
                out.push_back('{');
                size_t i = 0;
                for (const auto& kv : o_) {
                    nl(level + 1);
                    dump_string(out, kv.first);
                    out += pretty ? ": " : ":";
                    kv.second.dump_to(out, indent, level + 1);
                    if (++i < o_.size()) out.push_back(',');
                }
                nl(level);
                out.push_back('}');
                break;
            }
        }
    }

    struct Parser {
        const char* cur;
        const char* end;
        [[noreturn]] void fail(const char* what) const { throw JsonError(std::string("parse error: ") + what); }
        void ws() {
            while (cur != end && (*cur == ' ' || *cur == '\t' || *cur == '\n' || *cur == '\r')) ++cur;
        }
        bool lit(const char* s) {
            size_t n = std::strlen(s);
            if ((size_t)(end - cur) >= n && std::memcmp(cur, s, n) == 0) {
                cur += n;
                return true;
            }
            return false;
        }
        static void utf8(std::string& out, unsigned cp) {
            if (cp < 0x80)
                out.push_back((char)cp);
            else if (cp < 0x800) {
                out.push_back((char)(0xC0 | (cp >> 6)));
                out.push_back((char)(0x80 | (cp & 0x3F)));
            } else if (cp < 0x10000) {
                out.push_back((char)(0xE0 | (cp >> 12)));
                out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                out.push_back((char)(0x80 | (cp & 0x3F)));
            } else {
                out.push_back((char)(0xF0 | (cp >> 18)));
                out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
                out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                out.push_back((char)(0x80 | (cp & 0x3F)));
            }
        }
        unsigned hex4() {
            if (end - cur < 4) fail("truncated \\u escape");
            unsigned v = 0;
            for (int i = 0; i < 4; ++i) {
                char c = *cur++;
                v <<= 4;
                if (c >= '0' && c <= '9')
                    v |= (unsigned)(c - '0');
                else if (c >= 'a' && c <= 'f')
                    v |= (unsigned)(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F')
                    v |= (unsigned)(c - 'A' + 10);
                else
                    fail("bad \\u escape");
            }
            return v;
        }
        std::string string() {
            if (cur == end || *cur != '"') fail("expected string");
            ++cur;
            std::string out;
            for (;;) {
                if (cur == end) fail("unterminated string");
                unsigned char c = (unsigned char)*cur++;
                if (c == '"') break;
                if (c < 0x20) fail("control character in string");
                if (c != '\\') {
                    out.push_back((char)c);
                    if (c >= 0x80) {
                        // well-formed UTF-8 only, like nlohmann's lexer (the reference's Scene::Load
                        // rejects the whole file otherwise): lead byte -> allowed range of the 2nd byte
                        int more;
                        unsigned char lo = 0x80, hi = 0xBF;
                        if (c >= 0xC2 && c <= 0xDF) more = 1;
                        else if (c == 0xE0) more = 2, lo = 0xA0;
                        else if ((c >= 0xE1 && c <= 0xEC) || c == 0xEE || c == 0xEF) more = 2;
                        else if (c == 0xED) more = 2, hi = 0x9F;
                        else if (c == 0xF0) more = 3, lo = 0x90;
                        else if (c >= 0xF1 && c <= 0xF3) more = 3;
                        else if (c == 0xF4) more = 3, hi = 0x8F;
                        else fail("invalid UTF-8 in string");
                        for (int k = 0; k < more; ++k) {
                            if (cur == end) fail("invalid UTF-8 in string");
                            const unsigned char t = (unsigned char)*cur;
                            if (t < (k == 0 ? lo : 0x80) || t > (k == 0 ? hi : 0xBF)) fail("invalid UTF-8 in string");
                            out.push_back((char)t);
                            ++cur;
                        }
                    }
                    continue;
                }
                if (cur == end) fail("unterminated escape");
                char e = *cur++;
                switch (e) {
                    case '"': out.push_back('"'); break;
                    case '\\': out.push_back('\\'); break;
                    case '/': out.push_back('/'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'n': out.push_back('\n'); break;
                    case 'r': out.push_back('\r'); break;
                    case 't': out.push_back('\t'); break;
                    case 'u': {
                        unsigned cp = hex4();
                        if (cp >= 0xD800 && cp <= 0xDBFF) {
                            if (end - cur < 2 || cur[0] != '\\' || cur[1] != 'u') fail("unpaired surrogate");
                            cur += 2;
                            unsigned lo = hex4();
                            if (lo < 0xDC00 || lo > 0xDFFF) fail("bad low surrogate");
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                            fail("unpaired surrogate");
                        }
                        utf8(out, cp);
                        break;
                    }
                    default: fail("bad escape");
                }
            }
            return out;
        }
        Json number() {
            const char* s = cur;
            bool is_float = false;
            if (cur != end && *cur == '-') ++cur;
            if (cur == end) fail("bad number");
            if (*cur == '0') {
                ++cur;
            } else if (*cur >= '1' && *cur <= '9') {
                while (cur != end && *cur >= '0' && *cur <= '9') ++cur;
            } else {
                fail("bad number");
            }
            if (cur != end && *cur == '.') {
                is_float = true;
                ++cur;
                if (cur == end || *cur < '0' || *cur > '9') fail("bad fraction");
                while (cur != end && *cur >= '0' && *cur <= '9') ++cur;
            }
            if (cur != end && (*cur == 'e' || *cur == 'E')) {
                is_float = true;
                ++cur;
                if (cur != end && (*cur == '+' || *cur == '-')) ++cur;
                if (cur == end || *cur < '0' || *cur > '9') fail("bad exponent");
                while (cur != end && *cur >= '0' && *cur <= '9') ++cur;
            }
            if (!is_float) {
                if (*s == '-') {
                    int64_t v = 0;
                    auto r = std::from_chars(s, cur, v);
                    if (r.ec == std::errc() && r.ptr == cur) return Json(v);
                } else {
                    uint64_t v = 0;
                    auto r = std::from_chars(s, cur, v);
                    if (r.ec == std::errc() && r.ptr == cur) return v <= (uint64_t)INT64_MAX ? Json((int64_t)v) : Json(v);
                }
                // out of integer range: falls through to double, like nlohmann
            }
            double d = 0;
            auto r = std::from_chars(s, cur, d);
            if (r.ec == std::errc::result_out_of_range) {
                // nlohmann (strtod, then !isfinite -> error) rejects overflow but takes an underflow
                // as the (possibly zero or subnormal) value it rounds to
                d = std::strtod(std::string(s, cur).c_str(), nullptr);
                if (!std::isfinite(d)) fail("number overflow");
                return Json(d);
            }
            if (r.ec != std::errc() || r.ptr != cur) fail("bad number");
            return Json(d);
        }
        Json value(int depth) {
            if (depth > 512) fail("nesting too deep");
            ws();
            if (cur == end) fail("unexpected end of input");
            char c = *cur;
            if (c == '{') {
                ++cur;
                Json o = Json::object();
                ws();
                if (cur != end && *cur == '}') {
                    ++cur;
                    return o;
                }
                for (;;) {
                    ws();
                    std::string k = string();
                    ws();
                    if (cur == end || *cur != ':') fail("expected ':'");
                    ++cur;
                    o.o_[k] = value(depth + 1);  // duplicate keys: last one wins (nlohmann default)
                    ws();
                    if (cur != end && *cur == ',') {
                        ++cur;
                        continue;
                    }
                    if (cur != end && *cur == '}') {
                        ++cur;
                        return o;
                    }
                    fail("expected ',' or '}'");
                }
            }
            if (c == '[') {
                ++cur;
                Json a = Json::array();
                ws();
                if (cur != end && *cur == ']') {
                    ++cur;
                    return a;
                }
                for (;;) {
                    a.a_.push_back(value(depth + 1));
                    ws();
                    if (cur != end && *cur == ',') {
                        ++cur;
                        continue;
                    }
                    if (cur != end && *cur == ']') {
                        ++cur;
                        return a;
                    }
                    fail("expected ',' or ']'");
                }
            }
            if (c == '"') return Json(string());
            if (lit("true")) return Json(true);
            if (lit("false")) return Json(false);
            if (lit("null")) return Json(nullptr);
            if (c == '-' || (c >= '0' && c <= '9')) return number();
            fail("unexpected character");
        }
    };
};

}  // namespace srt_host
