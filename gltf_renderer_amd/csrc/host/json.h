// json.h -- a small JSON reader for the glTF loader (the reference gets nlohmann-json through tinygltf; neither is available
// here, and only /root/repo travels to the GPU box, so the loader carries its own).  RFC 8259: objects, arrays, strings with
// all escapes (\uXXXX incl. surrogate pairs -> UTF-8), numbers (kept as double + "was written as an integer"), true/false/null.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace hostjson {

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0;
    bool is_int = false;                       // number token had no '.', 'e', 'E'
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;   // insertion order kept

    bool is_object() const { return type == Object; }
    bool is_array() const { return type == Array; }
    bool is_number() const { return type == Number; }
    bool is_string() const { return type == String; }
    bool has(const char* k) const { return find(k) != nullptr; }
    const Value* find(const char* k) const {
        if (type != Object) return nullptr;
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    // tinygltf::Value::Get on a missing key returns a null value; mirror that so call sites read like the reference's
    const Value& get(const char* k) const { static const Value null_value; const Value* v = find(k); return v ? *v : null_value; }
    const Value& at(size_t i) const { static const Value null_value; return (type == Array && i < arr.size()) ? arr[i] : null_value; }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
    double number_or(double d) const { return type == Number ? num : d; }
    int int_or(int d) const { return type == Number ? (int)num : d; }
    std::string string_or(const char* d) const { return type == String ? str : std::string(d); }
};

class Parser {
  public:
    bool parse(const char* text, size_t n, Value& out, std::string& err) {
        p_ = text; end_ = text + n; err_.clear();
        skip_bom();
        if (!value(out, 0)) { err = err_; return false; }
        ws();
        if (p_ != end_) { err = "trailing characters after JSON value"; return false; }
        return true;
    }

  private:
    const char *p_ = nullptr, *end_ = nullptr;
    std::string err_;
    bool fail(const char* m) { if (err_.empty()) err_ = m; return false; }
    void skip_bom() { if (end_ - p_ >= 3 && (uint8_t)p_[0] == 0xEF && (uint8_t)p_[1] == 0xBB && (uint8_t)p_[2] == 0xBF) p_ += 3; }
    void ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++; }
    bool lit(const char* s) { size_t n = strlen(s); if ((size_t)(end_ - p_) >= n && !memcmp(p_, s, n)) { p_ += n; return true; } return false; }
    static void utf8(std::string& s, uint32_t c) {
        if (c < 0x80) s += (char)c;
        else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
        else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 0x3F)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t& v) {
        if (end_ - p_ < 4) return fail("truncated \\u escape");
        v = 0;
        for (int i = 0; i < 4; i++) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string& s) {
        if (p_ >= end_ || *p_ != '"') return fail("expected string");
        p_++;
        s.clear();
        while (p_ < end_) {
            char c = *p_++;
            if (c == '"') return true;
            if ((uint8_t)c < 0x20) return fail("control character in string");
            if (c != '\\') { s += c; continue; }
            if (p_ >= end_) break;
            c = *p_++;
            switch (c) {
                case '"': s += '"'; break;
                case '\\': s += '\\'; break;
                case '/': s += '/'; break;
                case 'b': s += '\b'; break;
                case 'f': s += '\f'; break;
                case 'n': s += '\n'; break;
                case 'r': s += '\r'; break;
                case 't': s += '\t'; break;
                case 'u': {
                    uint32_t u;
                    if (!hex4(u)) return false;
                    if (u >= 0xD800 && u <= 0xDBFF && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        uint32_t lo;
                        if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo <= 0xDFFF) u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00);
                        else { utf8(s, 0xFFFD); u = lo; }
                    }
                    utf8(s, u);
                } break;
                default: return fail("bad escape");
            }
        }
        return fail("unterminated string");
    }
    bool number(Value& v) {
        const char* s = p_;
        bool integer = true;
        if (p_ < end_ && *p_ == '-') p_++;
        if (p_ >= end_ || !(*p_ >= '0' && *p_ <= '9')) return fail("bad number");
        while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
        if (p_ < end_ && *p_ == '.') { integer = false; p_++; while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++; }
        if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
            integer = false; p_++;
            if (p_ < end_ && (*p_ == '+' || *p_ == '-')) p_++;
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
        }
        std::string tok(s, p_ - s);
        v.type = Value::Number;
        v.num = strtod(tok.c_str(), nullptr);
        v.is_int = integer;
        return true;
    }
    bool value(Value& v, int depth) {
        if (depth > 256) return fail("nesting too deep");
        ws();
        if (p_ >= end_) return fail("unexpected end of input");
        char c = *p_;
        if (c == '{') {
            p_++;
            v.type = Value::Object;
            ws();
            if (p_ < end_ && *p_ == '}') { p_++; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!string(k)) return false;
                ws();
                if (p_ >= end_ || *p_ != ':') return fail("expected ':'");
                p_++;
                v.obj.emplace_back(std::move(k), Value());
                if (!value(v.obj.back().second, depth + 1)) return false;
                ws();
                if (p_ < end_ && *p_ == ',') { p_++; continue; }
                if (p_ < end_ && *p_ == '}') { p_++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            p_++;
            v.type = Value::Array;
            ws();
            if (p_ < end_ && *p_ == ']') { p_++; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back(), depth + 1)) return false;
                ws();
                if (p_ < end_ && *p_ == ',') { p_++; continue; }
                if (p_ < end_ && *p_ == ']') { p_++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.type = Value::String; return string(v.str); }
        if (lit("true")) { v.type = Value::Bool; v.b = true; return true; }
        if (lit("false")) { v.type = Value::Bool; v.b = false; return true; }
        if (lit("null")) { v.type = Value::Null; return true; }
        return number(v);
    }
};

}  // namespace hostjson
