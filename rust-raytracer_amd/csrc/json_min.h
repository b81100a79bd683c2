// json_min.h -- a small strict JSON reader (RFC 8259 subset: no surrogate-pair
// validation) used by the scene loader.  Objects keep their members in file
// order so that "later key wins" lookups behave like a map insert loop.
#pragma once

#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace jsonmin {

struct Value;
using Member = std::pair<std::string, Value>;

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;  // String payload
    std::vector<Value> arr;
    std::vector<Member> obj;

    bool is_object() const { return type == Object; }
    bool is_array() const { return type == Array; }
    bool is_string() const { return type == String; }
    bool is_number() const { return type == Number; }

    // First member with any of the given names (aliases), or nullptr.
    const Value *find(std::initializer_list<const char *> names) const {
        if (type != Object) return nullptr;
        for (const auto &m : obj)
            for (const char *n : names)
                if (m.first == n) return &m.second;
        return nullptr;
    }
};

class Parser {
  public:
    explicit Parser(const char *text) : p_(text) {}

    Value parse_document() {
        Value v = parse_value();
        skip_ws();
        if (*p_ != '\0') fail("trailing characters after JSON document");
        return v;
    }

  private:
    const char *p_;
    int depth_ = 0;

    [[noreturn]] void fail(const char *msg) { throw std::runtime_error(std::string("JSON: ") + msg); }

    void skip_ws() {
        while (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r') ++p_;
    }

    Value parse_value() {
        skip_ws();
        if (++depth_ > 128) fail("nesting too deep");
        Value v;
        switch (*p_) {
        case '{': v = parse_object(); break;
        case '[': v = parse_array(); break;
        case '"':
            v.type = Value::String;
            v.str = parse_string();
            break;
        case 't':
            expect_word("true");
            v.type = Value::Bool;
            v.b = true;
            break;
        case 'f':
            expect_word("false");
            v.type = Value::Bool;
            v.b = false;
            break;
        case 'n':
            expect_word("null");
            v.type = Value::Null;
            break;
        default: v = parse_number(); break;
        }
        --depth_;
        return v;
    }

    void expect_word(const char *w) {
        size_t n = std::strlen(w);
        if (std::strncmp(p_, w, n) != 0) fail("invalid literal");
        p_ += n;
    }

    Value parse_number() {
        const char *s = p_;
        if (*p_ == '-') ++p_;
        if (*p_ == '0') {
            ++p_;
        } else if (*p_ >= '1' && *p_ <= '9') {
            while (*p_ >= '0' && *p_ <= '9') ++p_;
        } else {
            fail("invalid value");
        }
        if (*p_ == '.') {
            ++p_;
            if (!(*p_ >= '0' && *p_ <= '9')) fail("digit expected after decimal point");
            while (*p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (*p_ == 'e' || *p_ == 'E') {
            ++p_;
            if (*p_ == '+' || *p_ == '-') ++p_;
            if (!(*p_ >= '0' && *p_ <= '9')) fail("digit expected in exponent");
            while (*p_ >= '0' && *p_ <= '9') ++p_;
        }
        Value v;
        v.type = Value::Number;
        v.num = std::strtod(std::string(s, p_).c_str(), nullptr);
        return v;
    }

    static void append_utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) {
            out.push_back(char(cp));
        } else if (cp < 0x800) {
            out.push_back(char(0xC0 | (cp >> 6)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        } else {
            out.push_back(char(0xE0 | (cp >> 12)));
            out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        }
    }

    std::string parse_string() {
        ++p_;  // opening quote
        std::string out;
        for (;;) {
            unsigned char c = (unsigned char)*p_;
            if (c == '\0') fail("unterminated string");
            if (c == '"') {
                ++p_;
                return out;
            }
            if (c < 0x20) fail("control character in string");
            if (c != '\\') {
                out.push_back(char(c));
                ++p_;
                continue;
            }
            ++p_;
            switch (*p_) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                unsigned cp = 0;
                for (int i = 1; i <= 4; ++i) {
                    char h = p_[i];
                    cp <<= 4;
                    if (h >= '0' && h <= '9') cp |= unsigned(h - '0');
                    else if (h >= 'a' && h <= 'f') cp |= unsigned(h - 'a' + 10);
                    else if (h >= 'A' && h <= 'F') cp |= unsigned(h - 'A' + 10);
                    else fail("bad \\u escape");
                }
                append_utf8(out, cp);
                p_ += 4;
                break;
            }
            default: fail("bad escape");
            }
            ++p_;
        }
    }

    Value parse_array() {
        ++p_;
        Value v;
        v.type = Value::Array;
        skip_ws();
        if (*p_ == ']') {
            ++p_;
            return v;
        }
        for (;;) {
            v.arr.push_back(parse_value());
            skip_ws();
            if (*p_ == ',') {
                ++p_;
                continue;
            }
            if (*p_ == ']') {
                ++p_;
                return v;
            }
            fail("',' or ']' expected in array");
        }
    }

    Value parse_object() {
        ++p_;
        Value v;
        v.type = Value::Object;
        skip_ws();
        if (*p_ == '}') {
            ++p_;
            return v;
        }
        for (;;) {
            skip_ws();
            if (*p_ != '"') fail("object key expected");
            std::string key = parse_string();
            skip_ws();
            if (*p_ != ':') fail("':' expected after object key");
            ++p_;
            Value member = parse_value();
            v.obj.emplace_back(std::move(key), std::move(member));
            skip_ws();
            if (*p_ == ',') {
                ++p_;
                continue;
            }
            if (*p_ == '}') {
                ++p_;
                return v;
            }
            fail("',' or '}' expected in object");
        }
    }
};

inline Value parse(const char *text) { return Parser(text).parse_document(); }

}  // namespace jsonmin
