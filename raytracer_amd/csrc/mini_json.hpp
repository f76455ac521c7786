// mini_json.hpp -- a small RFC 8259 reader for the scene grammar.
//
// The reference parses scenes with nlohmann/json 3.11.3, a third-party single header that is
// vendored inside the reference tree and not installed on this image.  Rather than copy it, the
// ingest uses this purpose-built reader (DESIGN.md "JSON").  What the ingest relies on is kept
// equal to nlohmann's behaviour:
//   - numbers: integers that fit int64/uint64 stay integral, everything else goes through
//     strtod (nlohmann's lexer does the same), so every double has identical bits;
//   - value(key, default): the default when the key is absent, else a converting get;
//   - number -> int conversion truncates like static_cast<int>;
//   - objects keep one value per key (last one wins).
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mini_json {

struct Value;
using Array = std::vector<Value>;
using Object = std::map<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, Int, UInt, Float, String, Arr, Obj } kind = Null;
    bool b = false;
    int64_t i = 0;
    uint64_t u = 0;
    double f = 0.0;
    std::string s;
    std::shared_ptr<Array> arr;
    std::shared_ptr<Object> obj;

    bool is_null() const { return kind == Null; }
    bool is_number() const { return kind == Int || kind == UInt || kind == Float; }
    bool is_object() const { return kind == Obj; }
    bool is_array() const { return kind == Arr; }

    bool contains(const std::string& key) const { return kind == Obj && obj->count(key) != 0; }

    // operator[] on a missing key yields null (what nlohmann's non-const operator[] leaves behind)
    const Value& operator[](const std::string& key) const {
        static const Value null_value;
        if (kind != Obj) return null_value;
        auto it = obj->find(key);
        return it == obj->end() ? null_value : it->second;
    }
    const Value& operator[](size_t idx) const {
        if (kind != Arr || idx >= arr->size()) throw std::runtime_error("json: array index out of range");
        return (*arr)[idx];
    }
    size_t size() const { return kind == Arr ? arr->size() : (kind == Obj ? obj->size() : 0); }

    double as_double() const {
        switch (kind) {
            case Int: return static_cast<double>(i);
            case UInt: return static_cast<double>(u);
            case Float: return f;
            case Bool: return b ? 1.0 : 0.0;
            default: throw std::runtime_error("json: value is not a number");
        }
    }
    int as_int() const {
        switch (kind) {
            case Int: return static_cast<int>(i);
            case UInt: return static_cast<int>(u);
            case Float: return static_cast<int>(f);
            case Bool: return b ? 1 : 0;
            default: throw std::runtime_error("json: value is not a number");
        }
    }
    bool as_bool() const {
        if (kind != Bool) throw std::runtime_error("json: value is not a boolean");
        return b;
    }
    const std::string& as_string() const {
        if (kind != String) throw std::runtime_error("json: value is not a string");
        return s;
    }

    double value(const std::string& key, double dflt) const { return contains(key) ? (*this)[key].as_double() : dflt; }
    int value(const std::string& key, int dflt) const { return contains(key) ? (*this)[key].as_int() : dflt; }
    bool value(const std::string& key, bool dflt) const { return contains(key) ? (*this)[key].as_bool() : dflt; }
    std::string value(const std::string& key, const char* dflt) const {
        return contains(key) ? (*this)[key].as_string() : std::string(dflt);
    }
};

class Parser {
public:
    explicit Parser(const std::string& text) : p_(text.data()), end_(text.data() + text.size()) {}

    Value parse() {
        // tolerate a UTF-8 BOM like nlohmann does
        if (end_ - p_ >= 3 && (unsigned char)p_[0] == 0xEF && (unsigned char)p_[1] == 0xBB && (unsigned char)p_[2] == 0xBF) p_ += 3;
        Value v = parse_value();
        skip_ws();
        if (p_ != end_) fail("trailing characters after the top-level value");
        return v;
    }

private:
    const char* p_;
    const char* end_;
    int depth_ = 0;  // nesting of objects / arrays: capped so that a hostile file cannot overflow the stack
    static constexpr int kMaxDepth = 256;
    struct Nest {
        Parser& p;
        explicit Nest(Parser& q) : p(q) { if (++p.depth_ > kMaxDepth) p.fail("nesting deeper than 256 levels"); }
        ~Nest() { --p.depth_; }
    };

    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json parse error: ") + what); }
    void skip_ws() {
        while (p_ != end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    bool consume(const char* lit) {
        size_t n = std::strlen(lit);
        if ((size_t)(end_ - p_) >= n && std::memcmp(p_, lit, n) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }

    Value parse_value() {
        skip_ws();
        if (p_ == end_) fail("unexpected end of input");
        Value v;
        switch (*p_) {
            case '{': { Nest n(*this); return parse_object(); }
            case '[': { Nest n(*this); return parse_array(); }
            case '"':
                v.kind = Value::String;
                v.s = parse_string();
                return v;
            case 't':
                if (!consume("true")) fail("bad literal");
                v.kind = Value::Bool;
                v.b = true;
                return v;
            case 'f':
                if (!consume("false")) fail("bad literal");
                v.kind = Value::Bool;
                v.b = false;
                return v;
            case 'n':
                if (!consume("null")) fail("bad literal");
                return v;
            default: return parse_number();
        }
    }

    Value parse_object() {
        Value v;
        v.kind = Value::Obj;
        v.obj = std::make_shared<Object>();
        ++p_;
        skip_ws();
        if (p_ != end_ && *p_ == '}') {
            ++p_;
            return v;
        }
        for (;;) {
            skip_ws();
            if (p_ == end_ || *p_ != '"') fail("object key expected");
            std::string key = parse_string();
            skip_ws();
            if (p_ == end_ || *p_ != ':') fail("':' expected");
            ++p_;
            (*v.obj)[key] = parse_value();
            skip_ws();
            if (p_ == end_) fail("unterminated object");
            if (*p_ == ',') { ++p_; continue; }
            if (*p_ == '}') { ++p_; return v; }
            fail("',' or '}' expected");
        }
    }

    Value parse_array() {
        Value v;
        v.kind = Value::Arr;
        v.arr = std::make_shared<Array>();
        ++p_;
        skip_ws();
        if (p_ != end_ && *p_ == ']') {
            ++p_;
            return v;
        }
        for (;;) {
            v.arr->push_back(parse_value());
            skip_ws();
            if (p_ == end_) fail("unterminated array");
            if (*p_ == ',') { ++p_; continue; }
            if (*p_ == ']') { ++p_; return v; }
            fail("',' or ']' expected");
        }
    }

    static void append_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out.push_back((char)cp);
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
    uint32_t parse_hex4() {
        if (end_ - p_ < 4) fail("bad \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string parse_string() {
        ++p_;  // opening quote
        std::string out;
        for (;;) {
            if (p_ == end_) fail("unterminated string");
            char c = *p_++;
            if (c == '"') return out;
            if ((unsigned char)c < 0x20) fail("control character in string");
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (p_ == end_) fail("unterminated escape");
            char e = *p_++;
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
                    uint32_t cp = parse_hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        if (end_ - p_ < 2 || p_[0] != '\\' || p_[1] != 'u') fail("lone surrogate");
                        p_ += 2;
                        uint32_t lo = parse_hex4();
                        if (lo < 0xDC00 || lo > 0xDFFF) fail("bad surrogate pair");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    append_utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
    }

    Value parse_number() {
        const char* start = p_;
        bool is_float = false;
        if (p_ != end_ && *p_ == '-') ++p_;
        if (p_ == end_) fail("bad number");
        if (*p_ == '0') ++p_;
        else if (*p_ >= '1' && *p_ <= '9') {
            while (p_ != end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        } else fail("unexpected character");
        if (p_ != end_ && *p_ == '.') {
            is_float = true;
            ++p_;
            if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("digit expected after '.'");
            while (p_ != end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (p_ != end_ && (*p_ == 'e' || *p_ == 'E')) {
            is_float = true;
            ++p_;
            if (p_ != end_ && (*p_ == '+' || *p_ == '-')) ++p_;
            if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("digit expected in exponent");
            while (p_ != end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        std::string tok(start, p_);
        Value v;
        if (!is_float) {
            errno = 0;
            char* endp = nullptr;
            if (tok[0] == '-') {
                long long x = std::strtoll(tok.c_str(), &endp, 10);
                if (errno == 0 && *endp == '\0') {
                    v.kind = Value::Int;
                    v.i = x;
                    return v;
                }
            } else {
                unsigned long long x = std::strtoull(tok.c_str(), &endp, 10);
                if (errno == 0 && *endp == '\0') {
                    if (x <= (unsigned long long)INT64_MAX) {
                        v.kind = Value::Int;
                        v.i = (int64_t)x;
                    } else {
                        v.kind = Value::UInt;
                        v.u = x;
                    }
                    return v;
                }
            }
        }
        v.kind = Value::Float;
        v.f = std::strtod(tok.c_str(), nullptr);
        return v;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace mini_json
