// Minimal JSON DOM (parser + pretty writer) for network files ("layers" / "weights" sections,
// autosave-free).  Stands in for the vendored rapidjson of the reference
// (currennt_lib/src/rapidjson/*, helpers/JsonClasses.hpp); written from scratch.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace currennt_hip {
namespace json {

class Value {
public:
    enum Type { Null, Bool, Number, String, Array, Object };

    Value() : m_type(Null), m_num(0), m_bool(false) {}
    explicit Value(Type t) : m_type(t), m_num(0), m_bool(false) {}
    Value(double d) : m_type(Number), m_num(d), m_bool(false) {}
    Value(int i) : m_type(Number), m_num(i), m_bool(false) {}
    Value(bool b) : m_type(Bool), m_num(0), m_bool(b) {}
    Value(const std::string &s) : m_type(String), m_num(0), m_bool(false), m_str(s) {}
    Value(const char *s) : m_type(String), m_num(0), m_bool(false), m_str(s) {}

    Type type() const { return m_type; }
    bool isObject() const { return m_type == Object; }
    bool isArray() const { return m_type == Array; }
    bool isNumber() const { return m_type == Number; }
    bool isString() const { return m_type == String; }

    double getDouble() const { need(Number, "number"); return m_num; }
    int getInt() const { need(Number, "number"); return (int)m_num; }
    bool getBool() const { if (m_type == Number) return m_num != 0; need(Bool, "bool"); return m_bool; }
    const std::string &getString() const { need(String, "string"); return m_str; }

    // arrays
    size_t size() const { return m_type == Array ? m_arr.size() : m_obj.size(); }
    const Value &operator[](size_t i) const { need(Array, "array"); return m_arr.at(i); }
    Value &operator[](size_t i) { need(Array, "array"); return m_arr.at(i); }
    void pushBack(const Value &v) { need(Array, "array"); m_arr.push_back(v); }
    void reserve(size_t n) { m_arr.reserve(n); }
    const std::vector<Value> &items() const { return m_arr; }

    // objects (insertion ordered, like the files the reference writes)
    bool hasMember(const std::string &k) const { return find(k) >= 0; }
    const Value &operator[](const std::string &k) const
    {
        int i = find(k);
        if (i < 0) throw std::runtime_error("Missing value '" + k + "'");
        return m_obj[i].second;
    }
    Value &operator[](const std::string &k)
    {
        need(Object, "object");
        int i = find(k);
        if (i < 0) { m_obj.push_back(std::make_pair(k, Value())); i = (int)m_obj.size() - 1; }
        return m_obj[i].second;
    }
    void addMember(const std::string &k, const Value &v) { (*this)[k] = v; }
    const std::vector<std::pair<std::string, Value> > &members() const { return m_obj; }

    // ---- parsing -----------------------------------------------------------------------------
    static Value parse(const std::string &text)
    {
        const char *p = text.c_str();
        Value v = parseValue(p);
        skipWs(p);
        if (*p) throw std::runtime_error("JSON: trailing characters");
        return v;
    }
    static Value parseFile(const std::string &filename)
    {
        FILE *f = fopen(filename.c_str(), "rb");
        if (!f) throw std::runtime_error("Cannot open file '" + filename + "'");   // main.cpp:553-554
        std::string buf;
        char tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n);
        fclose(f);
        try { return parse(buf); }
        catch (const std::exception &e) { throw std::runtime_error(std::string("Parsing failed: ") + e.what()); }
    }

    // ---- writing -----------------------------------------------------------------------------
    void write(std::string &out, int indent = 0) const
    {
        const std::string pad(indent * 4, ' '), pad2((indent + 1) * 4, ' ');
        char buf[64];
        switch (m_type) {
        case Null: out += "null"; break;
        case Bool: out += m_bool ? "true" : "false"; break;
        case Number:
            if (m_num == (double)(long long)m_num && m_num > -1e15 && m_num < 1e15) snprintf(buf, sizeof(buf), "%lld", (long long)m_num);
            else snprintf(buf, sizeof(buf), "%.9g", m_num);      // lossless for fp32 weights
            out += buf; break;
        case String: writeString(out, m_str); break;
        case Array:
            if (m_arr.empty()) { out += "[]"; break; }
            out += "[\n";
            for (size_t i = 0; i < m_arr.size(); ++i) {
                out += pad2; m_arr[i].write(out, indent + 1);
                out += (i + 1 < m_arr.size()) ? ",\n" : "\n";
            }
            out += pad + "]"; break;
        case Object:
            if (m_obj.empty()) { out += "{}"; break; }
            out += "{\n";
            for (size_t i = 0; i < m_obj.size(); ++i) {
                out += pad2; writeString(out, m_obj[i].first); out += ": ";
                m_obj[i].second.write(out, indent + 1);
                out += (i + 1 < m_obj.size()) ? ",\n" : "\n";
            }
            out += pad + "}"; break;
        }
    }
    void writeFile(const std::string &filename) const
    {
        std::string s; write(s); s += "\n";
        FILE *f = fopen(filename.c_str(), "w");
        if (!f) throw std::runtime_error("Cannot open file");                       // main.cpp:690-691
        fwrite(s.data(), 1, s.size(), f);
        fclose(f);
    }

private:
    Type m_type;
    double m_num;
    bool m_bool;
    std::string m_str;
    std::vector<Value> m_arr;
    std::vector<std::pair<std::string, Value> > m_obj;

    void need(Type t, const char *what) const
    {
        if (m_type != t) throw std::runtime_error(std::string("JSON value is not a") + (t == Array || t == Object ? "n " : " ") + what);
    }
    int find(const std::string &k) const
    {
        for (size_t i = 0; i < m_obj.size(); ++i) if (m_obj[i].first == k) return (int)i;
        return -1;
    }
    static void writeString(std::string &out, const std::string &s)
    {
        out += '"';
        for (size_t i = 0; i < s.size(); ++i) {
            char c = s[i];
            if (c == '"' || c == '\\') { out += '\\'; out += c; }
            else if (c == '\n') out += "\\n";
            else if (c == '\t') out += "\\t";
            else out += c;
        }
        out += '"';
    }
    static void skipWs(const char *&p) { while (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t') ++p; }
    static Value parseValue(const char *&p)
    {
        skipWs(p);
        switch (*p) {
        case '{': {
            Value v(Object); ++p; skipWs(p);
            if (*p == '}') { ++p; return v; }
            for (;;) {
                skipWs(p);
                if (*p != '"') throw std::runtime_error("JSON: expected member name");
                std::string k = parseString(p);
                skipWs(p);
                if (*p != ':') throw std::runtime_error("JSON: expected ':'");
                ++p;
                v.m_obj.push_back(std::make_pair(k, parseValue(p)));
                skipWs(p);
                if (*p == ',') { ++p; continue; }
                if (*p == '}') { ++p; return v; }
                throw std::runtime_error("JSON: expected ',' or '}'");
            } }
        case '[': {
            Value v(Array); ++p; skipWs(p);
            if (*p == ']') { ++p; return v; }
            for (;;) {
                v.m_arr.push_back(parseValue(p));
                skipWs(p);
                if (*p == ',') { ++p; continue; }
                if (*p == ']') { ++p; return v; }
                throw std::runtime_error("JSON: expected ',' or ']'");
            } }
        case '"': return Value(parseString(p));
        case 't': if (!strncmp(p, "true", 4)) { p += 4; return Value(true); } break;
        case 'f': if (!strncmp(p, "false", 5)) { p += 5; return Value(false); } break;
        case 'n': if (!strncmp(p, "null", 4)) { p += 4; return Value(); } break;
        default: {
            char *end = 0;
            double d = strtod(p, &end);
            if (end != p) { p = end; return Value(d); }
            break; }
        }
        throw std::runtime_error("JSON: unexpected character");
    }
    static std::string parseString(const char *&p)
    {
        std::string s; ++p;
        while (*p && *p != '"') {
            if (*p == '\\') {
                ++p;
                switch (*p) {
                case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
                case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                case 'u': { unsigned cp = (unsigned)strtoul(std::string(p + 1, 4).c_str(), 0, 16); s += (char)(cp < 128 ? cp : '?'); p += 4; break; }
                default: s += *p; break;
                }
                ++p;
            } else s += *p++;
        }
        if (*p != '"') throw std::runtime_error("JSON: unterminated string");
        ++p;
        return s;
    }
};

}  // namespace json
}  // namespace currennt_hip
