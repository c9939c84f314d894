// Minimal JSON DOM reader for the reference's `convert database -f json` output
// (ports/cli/src/cmds/convert.rs:161-205: serde_json::to_writer_pretty of `Tree`).
// Numbers keep their source text so that u64 ids/hashes beyond 2^53 stay exact.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace cls {

struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    std::string s;  // Str: decoded text; Num: source text
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;  // insertion order kept

    const JVal* get(const char* key) const {
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_null() const { return kind == Null; }
    uint64_t as_u64() const {
        if (kind != Num && kind != Str) throw std::runtime_error("json: expected a number");
        return strtoull(s.c_str(), nullptr, 10);
    }
    double as_f64() const {
        if (kind != Num) throw std::runtime_error("json: expected a number");
        return strtod(s.c_str(), nullptr);
    }
};

class JParser {
public:
    JParser(const char* p, size_t n) : p_(p), e_(p + n) {}
    JVal parse() {
        JVal v = value();
        ws();
        if (p_ != e_) fail("trailing characters");
        return v;
    }

private:
    const char* p_;
    const char* e_;
    [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m); }
    void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_; }
    JVal value() {
        ws();
        if (p_ >= e_) fail("unexpected end");
        JVal v;
        char c = *p_;
        if (c == '{') {
            v.kind = JVal::Obj;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == '}') { ++p_; return v; }
            for (;;) {
                ws();
                if (p_ >= e_ || *p_ != '"') fail("expected a key");
                std::string k = string();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = JVal::Arr;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == ']') { ++p_; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JVal::Str;
            v.s = string();
        } else if (c == 't' && e_ - p_ >= 4 && std::string(p_, 4) == "true") { v.kind = JVal::Bool; v.b = true; p_ += 4; }
        else if (c == 'f' && e_ - p_ >= 5 && std::string(p_, 5) == "false") { v.kind = JVal::Bool; p_ += 5; }
        else if (c == 'n' && e_ - p_ >= 4 && std::string(p_, 4) == "null") { p_ += 4; }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            v.kind = JVal::Num;
            const char* s = p_;
            while (p_ < e_ && (*p_ == '-' || *p_ == '+' || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || (*p_ >= '0' && *p_ <= '9'))) ++p_;
            v.s.assign(s, p_);
        } else fail("unexpected character");
        return v;
    }
    static void utf8(std::string& o, uint32_t cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    uint32_t hex4() {
        if (e_ - p_ < 4) fail("bad \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p_++;
            v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : (fail("bad hex"), 0));
        }
        return v;
    }
    std::string string() {
        ++p_;  // opening quote
        std::string o;
        while (p_ < e_ && *p_ != '"') {
            char c = *p_++;
            if (c != '\\') { o.push_back(c); continue; }
            if (p_ >= e_) fail("bad escape");
            char x = *p_++;
            switch (x) {
                case 'n': o.push_back('\n'); break;
                case 't': o.push_back('\t'); break;
                case 'r': o.push_back('\r'); break;
                case 'b': o.push_back('\b'); break;
                case 'f': o.push_back('\f'); break;
                case '/': o.push_back('/'); break;
                case '\\': o.push_back('\\'); break;
                case '"': o.push_back('"'); break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        uint32_t lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(o, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        if (p_ >= e_) fail("unterminated string");
        ++p_;
        return o;
    }
};

}  // namespace cls
