// Minimal JSON reader for glTF files: objects, arrays, strings (with \uXXXX kept
// as UTF-8), numbers, true/false/null.  No third-party dependency so that the
// host layer builds with nothing but a C++17 compiler.
#pragma once
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptamd {

struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    bool is(Type t) const { return type == t; }
    const Json* find(const std::string& key) const
    {
        if (type != Object) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& at(const std::string& key) const
    {
        const Json* j = find(key);
        if (!j) throw std::runtime_error("json: missing key '" + key + "'");
        return *j;
    }
    const Json& at(size_t i) const
    {
        if (type != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
        return arr[i];
    }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
    double number(double dflt) const { return type == Number ? num : dflt; }
    double get(const std::string& key, double dflt) const
    {
        const Json* j = find(key);
        return j && j->type == Number ? j->num : dflt;
    }
    int64_t geti(const std::string& key, int64_t dflt) const
    {
        double v = get(key, (double)dflt);
        if (!(v == v)) return dflt;                                   // NaN
        const double lim = 4611686018427387904.0;                    // 2^62: the cast below is defined for every value left
        return (int64_t)(v > lim ? lim : (v < -lim ? -lim : v));
    }
    std::string gets(const std::string& key, const std::string& dflt) const
    {
        const Json* j = find(key);
        return j && j->type == String ? j->str : dflt;
    }
};

class JsonParser {
public:
    explicit JsonParser(const std::string& text) : s(text) {}
    Json parse()
    {
        Json j = value();
        ws();
        if (p != s.size()) fail("trailing characters");
        return j;
    }

private:
    const std::string& s;
    size_t p = 0;
    [[noreturn]] void fail(const char* what) const
    {
        throw std::runtime_error(std::string("json: ") + what + " at byte " + std::to_string(p));
    }
    void ws()
    {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++;
    }
    Json value()
    {
        ws();
        if (p >= s.size()) fail("unexpected end");
        char c = s[p];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { Json j; j.type = Json::String; j.str = string(); return j; }
        if (c == 't' && s.compare(p, 4, "true") == 0) { p += 4; Json j; j.type = Json::Bool; j.b = true; return j; }
        if (c == 'f' && s.compare(p, 5, "false") == 0) { p += 5; Json j; j.type = Json::Bool; return j; }
        if (c == 'n' && s.compare(p, 4, "null") == 0) { p += 4; return Json(); }
        return number();
    }
    Json number()
    {
        const char* start = s.c_str() + p;
        char* end = nullptr;
        double d = std::strtod(start, &end);
        if (end == start) fail("bad number");
        p += (size_t)(end - start);
        Json j; j.type = Json::Number; j.num = d;
        return j;
    }
    static void utf8(std::string& out, unsigned cp)
    {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    std::string string()
    {
        std::string out;
        p++;   // opening quote
        for (;;) {
            if (p >= s.size()) fail("unterminated string");
            char c = s[p++];
            if (c == '"') break;
            if (c != '\\') { out += c; continue; }
            if (p >= s.size()) fail("bad escape");
            char e = s[p++];
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                if (p + 4 > s.size()) fail("bad \\u escape");
                unsigned cp = (unsigned)std::strtoul(s.substr(p, 4).c_str(), nullptr, 16);
                p += 4;
                utf8(out, cp);
                break;
            }
            default: fail("bad escape");
            }
        }
        return out;
    }
    Json array()
    {
        Json j; j.type = Json::Array;
        p++;
        ws();
        if (p < s.size() && s[p] == ']') { p++; return j; }
        for (;;) {
            j.arr.push_back(value());
            ws();
            if (p >= s.size()) fail("unterminated array");
            if (s[p] == ',') { p++; continue; }
            if (s[p] == ']') { p++; break; }
            fail("expected , or ]");
        }
        return j;
    }
    Json object()
    {
        Json j; j.type = Json::Object;
        p++;
        ws();
        if (p < s.size() && s[p] == '}') { p++; return j; }
        for (;;) {
            ws();
            if (p >= s.size() || s[p] != '"') fail("expected key");
            std::string k = string();
            ws();
            if (p >= s.size() || s[p] != ':') fail("expected :");
            p++;
            j.obj.emplace_back(std::move(k), value());
            ws();
            if (p >= s.size()) fail("unterminated object");
            if (s[p] == ',') { p++; continue; }
            if (s[p] == '}') { p++; break; }
            fail("expected , or }");
        }
        return j;
    }
};

} // namespace ptamd
