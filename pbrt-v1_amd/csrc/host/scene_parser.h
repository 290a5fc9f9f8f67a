// scene_parser.h -- hand-written tokenizer + recursive-descent parser for the
// pbrt-v1 scene description language.
//
// The reference builds its front end from a flex lexer (core/pbrtlex.l:95-178)
// and a bison grammar (core/pbrtparse.y:150-468) that turns every statement
// into one pbrt* API call.  Neither tool exists in this environment, and the
// product wants the same language, so this is an independent implementation of
// the same token rules and statement set.  It talks to an abstract
// DirectiveSink instead of to a concrete API so that the very same parser can
// drive (a) the MI355X host scene builder (scene_api.h) and (b) the compiled
// reference itself in oracle/ref (through an adapter onto core/api.h:29-85).
//
// Token rules honoured (pbrtlex.l):  '#' comments to end of line; numbers
// [-+]?digits[.digits][e[-+]digits]; quoted strings with the escapes n t r b f " and backslash,
// plus 3-digit decimal escapes; '[' ']' ; bare identifiers for directives; Include "file".
// Parameter typing rules (pbrtparse.y:470-600): "type name" with type one of
// float integer bool point vector normal string texture color; integers are
// parsed as floats and truncated; bools are the strings "true"/"false" (every
// element takes the FIRST string, as the reference does); a bare (unbracketed)
// single string given to a non-string parameter re-types it as a texture.
#pragma once
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace pbrthip {

enum class ParamType { Float, Int, Bool, Point, Vector, Normal, Color, String, Texture };

struct Param {
    ParamType type;
    std::string name;
    std::vector<float> nums;          // Float/Point/Vector/Normal/Color (flat) and Int (pre-truncation)
    std::vector<std::string> strs;    // String/Texture/Bool source strings
};
using ParamList = std::vector<Param>;

// One virtual per statement of the grammar (pbrtparse.y:294-468).
struct DirectiveSink {
    virtual ~DirectiveSink() {}
    virtual void Identity() = 0;
    virtual void Translate(float x, float y, float z) = 0;
    virtual void Rotate(float angle, float x, float y, float z) = 0;
    virtual void Scale(float x, float y, float z) = 0;
    virtual void LookAt(const float v[9]) = 0;
    virtual void ConcatTransform(const float m[16]) = 0;
    virtual void Transform(const float m[16]) = 0;
    virtual void CoordinateSystem(const std::string &n) = 0;
    virtual void CoordSysTransform(const std::string &n) = 0;
    virtual void PixelFilter(const std::string &n, const ParamList &p) = 0;
    virtual void Film(const std::string &n, const ParamList &p) = 0;
    virtual void Sampler(const std::string &n, const ParamList &p) = 0;
    virtual void Accelerator(const std::string &n, const ParamList &p) = 0;
    virtual void SurfaceIntegrator(const std::string &n, const ParamList &p) = 0;
    virtual void VolumeIntegrator(const std::string &n, const ParamList &p) = 0;
    virtual void Camera(const std::string &n, const ParamList &p) = 0;
    virtual void SearchPath(const std::string &n) = 0;
    virtual void WorldBegin() = 0;
    virtual void AttributeBegin() = 0;
    virtual void AttributeEnd() = 0;
    virtual void TransformBegin() = 0;
    virtual void TransformEnd() = 0;
    virtual void Texture(const std::string &name, const std::string &type,
                         const std::string &cls, const ParamList &p) = 0;
    virtual void Material(const std::string &n, const ParamList &p) = 0;
    virtual void LightSource(const std::string &n, const ParamList &p) = 0;
    virtual void AreaLightSource(const std::string &n, const ParamList &p) = 0;
    virtual void Shape(const std::string &n, const ParamList &p) = 0;
    virtual void ReverseOrientation() = 0;
    virtual void Volume(const std::string &n, const ParamList &p) = 0;
    virtual void ObjectBegin(const std::string &n) = 0;
    virtual void ObjectEnd() = 0;
    virtual void ObjectInstance(const std::string &n) = 0;
    virtual void WorldEnd() = 0;
    // diagnostics (reference: core/util.cpp:36-97 print "file(line): msg")
    virtual void Diagnostic(int severity, const std::string &msg) {
        std::fprintf(stderr, "%s: %s\n", severity ? "Error" : "Warning", msg.c_str());
    }
};

class SceneParser {
  public:
    explicit SceneParser(DirectiveSink &s) : sink(s) {}

    bool ParseFile(const std::string &path) {
        std::ifstream in(path.c_str(), std::ios::binary);
        if (!in) { Report(1, "Unable to open scene file \"" + path + "\""); return false; }
        std::stringstream ss; ss << in.rdbuf();
        return ParseString(ss.str(), path);
    }

    bool ParseString(const std::string &text, const std::string &name = "<string>") {
        Source src; src.text = text; src.name = name;
        size_t slash = name.find_last_of('/');
        src.dir = slash == std::string::npos ? std::string() : name.substr(0, slash + 1);
        stack.push_back(src);
        bool ok = Run();
        stack.clear();
        return ok && nErrors == 0;
    }

    int errors() const { return nErrors; }

  private:
    enum Tok { T_EOF, T_NUM, T_STR, T_ID, T_LB, T_RB };
    struct Source { std::string text, name, dir; size_t pos = 0; int line = 1; };
    DirectiveSink &sink;
    std::vector<Source> stack;
    int nErrors = 0;
    // one-token lookahead
    bool havePeek = false; Tok peekTok = T_EOF; std::string peekText; float peekNum = 0;

    void Report(int sev, const std::string &msg) {
        std::string where;
        if (!stack.empty()) {
            char buf[64]; std::snprintf(buf, sizeof buf, "(%d): ", stack.back().line);
            where = stack.back().name + buf;
        }
        if (sev) ++nErrors;
        sink.Diagnostic(sev, where + msg);
    }

    Tok Lex(std::string &text, float &num) {
        for (;;) {
            if (stack.empty()) return T_EOF;
            Source &s = stack.back();
            const std::string &t = s.text;
            while (s.pos < t.size()) {
                char c = t[s.pos];
                if (c == '\n') { ++s.line; ++s.pos; }
                else if (c == ' ' || c == '\t' || c == '\r') ++s.pos;
                else if (c == '#') { while (s.pos < t.size() && t[s.pos] != '\n') ++s.pos; }
                else break;
            }
            if (s.pos >= t.size()) {
                if (stack.size() > 1) { stack.pop_back(); continue; }
                return T_EOF;
            }
            char c = t[s.pos];
            if (c == '[') { ++s.pos; return T_LB; }
            if (c == ']') { ++s.pos; return T_RB; }
            if (c == '"') {
                ++s.pos; text.clear();
                while (s.pos < t.size() && t[s.pos] != '"') {
                    char ch = t[s.pos++];
                    if (ch == '\n') { Report(1, "Unterminated string!"); ++s.line; return T_STR; }
                    if (ch == '\\' && s.pos < t.size()) {
                        char e = t[s.pos++];
                        if (e == 'n') text += '\n'; else if (e == 't') text += '\t';
                        else if (e == 'r') text += '\r'; else if (e == 'b') text += '\b';
                        else if (e == 'f') text += '\f'; else if (e == '\n') ++s.line;
                        else if (std::isdigit((unsigned char)e) && s.pos + 1 < t.size() &&
                                 std::isdigit((unsigned char)t[s.pos]) && std::isdigit((unsigned char)t[s.pos + 1])) {
                            int v = (e - '0') * 100 + (t[s.pos] - '0') * 10 + (t[s.pos + 1] - '0');
                            s.pos += 2; while (v > 256) v -= 256; text += (char)v;
                        } else text += e;
                    } else text += ch;
                }
                if (s.pos < t.size()) ++s.pos;  // closing quote
                return T_STR;
            }
            if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
                size_t b = s.pos, p = s.pos;
                if (t[p] == '-' || t[p] == '+') ++p;
                size_t digits = 0;
                while (p < t.size() && std::isdigit((unsigned char)t[p])) { ++p; ++digits; }
                if (p < t.size() && t[p] == '.') { ++p; while (p < t.size() && std::isdigit((unsigned char)t[p])) { ++p; ++digits; } }
                if (digits == 0) { Report(1, std::string("Illegal character: ") + c); ++s.pos; continue; }
                if (p < t.size() && (t[p] == 'e' || t[p] == 'E')) {
                    size_t q = p + 1;
                    if (q < t.size() && (t[q] == '-' || t[q] == '+')) ++q;
                    if (q < t.size() && std::isdigit((unsigned char)t[q])) { while (q < t.size() && std::isdigit((unsigned char)t[q])) ++q; p = q; }
                }
                text = t.substr(b, p - b); s.pos = p;
                num = (float)std::atof(text.c_str());   // pbrtlex.l: (float) atof(yytext)
                return T_NUM;
            }
            if (std::isalpha((unsigned char)c) || c == '_') {
                size_t b = s.pos;
                while (s.pos < t.size() && (std::isalnum((unsigned char)t[s.pos]) || t[s.pos] == '_')) ++s.pos;
                text = t.substr(b, s.pos - b);
                return T_ID;
            }
            Report(1, std::string("Illegal character: ") + c);
            ++s.pos;
        }
    }
    Tok Peek() { if (!havePeek) { peekTok = Lex(peekText, peekNum); havePeek = true; } return peekTok; }
    Tok Next(std::string &text, float &num) {
        if (havePeek) { havePeek = false; text = peekText; num = peekNum; return peekTok; }
        return Lex(text, num);
    }

    bool WantNums(float *out, int n, const char *what) {
        std::string tx; float v;
        for (int i = 0; i < n; ++i) {
            if (Peek() != T_NUM) { Report(1, std::string("parse error: expected number in ") + what); return false; }
            Next(tx, v); out[i] = v;
        }
        return true;
    }
    bool WantString(std::string &out, const char *what) {
        float v;
        if (Peek() != T_STR) { Report(1, std::string("parse error: expected quoted string after ") + what); return false; }
        Next(out, v); return true;
    }
    // '[' n n n ... ']'  or a single bare number
    bool WantNumArray(std::vector<float> &out, const char *what) {
        std::string tx; float v; out.clear();
        if (Peek() == T_NUM) { Next(tx, v); out.push_back(v); return true; }
        if (Peek() != T_LB) { Report(1, std::string("parse error: expected array in ") + what); return false; }
        Next(tx, v);
        while (Peek() == T_NUM) { Next(tx, v); out.push_back(v); }
        if (Peek() != T_RB) { Report(1, std::string("parse error: unterminated array in ") + what); return false; }
        Next(tx, v); return true;
    }

    static bool DecodeType(const std::string &decl, ParamType &type, std::string &name) {
        size_t p = 0; while (p < decl.size() && std::isspace((unsigned char)decl[p])) ++p;
        static const struct { const char *kw; ParamType t; } kinds[] = {
            {"float", ParamType::Float}, {"integer", ParamType::Int}, {"bool", ParamType::Bool},
            {"point", ParamType::Point}, {"vector", ParamType::Vector}, {"normal", ParamType::Normal},
            {"string", ParamType::String}, {"texture", ParamType::Texture}, {"color", ParamType::Color}};
        for (auto &k : kinds) {
            size_t n = std::strlen(k.kw);
            if (decl.compare(p, n, k.kw) == 0) {
                p += n; while (p < decl.size() && std::isspace((unsigned char)decl[p])) ++p;
                type = k.t; name = decl.substr(p); return true;
            }
        }
        return false;
    }

    void ReadParamList(ParamList &pl) {
        pl.clear();
        std::string tx; float v;
        while (Peek() == T_STR) {
            std::string decl; Next(decl, v);
            Param prm; bool bareString = false, isStrings = false;
            std::vector<float> nums; std::vector<std::string> strs;
            Tok t = Peek();
            if (t == T_NUM) { Next(tx, v); nums.push_back(v); }
            else if (t == T_STR) { Next(tx, v); strs.push_back(tx); bareString = true; isStrings = true; }
            else if (t == T_LB) {
                Next(tx, v);
                if (Peek() == T_STR) { isStrings = true; while (Peek() == T_STR) { Next(tx, v); strs.push_back(tx); } }
                else while (Peek() == T_NUM) { Next(tx, v); nums.push_back(v); }
                if (Peek() != T_RB) { Report(1, "parse error: unterminated parameter array for \"" + decl + "\""); return; }
                Next(tx, v);
            } else { Report(1, "parse error: parameter \"" + decl + "\" has no value"); return; }
            if (!DecodeType(decl, prm.type, prm.name)) {
                Report(0, "Type of parameter \"" + decl + "\" is unknown"); continue;
            }
            if (bareString && prm.type != ParamType::Texture && prm.type != ParamType::String) {
                Report(0, "Bad type for " + prm.name + ". Changing it to a texture.");
                prm.type = ParamType::Texture;
            }
            (void)isStrings;
            prm.nums.swap(nums); prm.strs.swap(strs);
            pl.push_back(prm);
        }
    }

    bool Run() {
        std::string id, name; float v; ParamList pl; float f[16];
        for (;;) {
            Tok t = Next(id, v);
            if (t == T_EOF) return true;
            if (t != T_ID) { Report(1, "parse error: expected a directive, got \"" + id + "\""); continue; }
            if (id == "Include") {
                std::string fn; if (!WantString(fn, "Include")) continue;
                Source inc; inc.name = (fn.size() && fn[0] == '/') ? fn : stack.back().dir + fn;
                std::ifstream in(inc.name.c_str(), std::ios::binary);
                if (!in) { Report(1, "Unable to open included scene file \"" + inc.name + "\""); continue; }
                std::stringstream ss; ss << in.rdbuf(); inc.text = ss.str();
                size_t slash = inc.name.find_last_of('/');
                inc.dir = slash == std::string::npos ? std::string() : inc.name.substr(0, slash + 1);
                stack.push_back(inc);
            }
            else if (id == "Identity") sink.Identity();
            else if (id == "Translate") { if (WantNums(f, 3, "Translate")) sink.Translate(f[0], f[1], f[2]); }
            else if (id == "Rotate") { if (WantNums(f, 4, "Rotate")) sink.Rotate(f[0], f[1], f[2], f[3]); }
            else if (id == "Scale") { if (WantNums(f, 3, "Scale")) sink.Scale(f[0], f[1], f[2]); }
            else if (id == "LookAt") { if (WantNums(f, 9, "LookAt")) sink.LookAt(f); }
            else if (id == "ConcatTransform" || id == "Transform") {
                std::vector<float> a;
                if (!WantNumArray(a, id.c_str())) continue;
                if (a.size() != 16) { Report(1, "Array argument to " + id + " isn't 16 elements long!"); continue; }
                if (id == "Transform") sink.Transform(a.data()); else sink.ConcatTransform(a.data());
            }
            else if (id == "CoordinateSystem") { if (WantString(name, id.c_str())) sink.CoordinateSystem(name); }
            else if (id == "CoordSysTransform") { if (WantString(name, id.c_str())) sink.CoordSysTransform(name); }
            else if (id == "SearchPath") { if (WantString(name, id.c_str())) sink.SearchPath(name); }
            else if (id == "ObjectBegin") { if (WantString(name, id.c_str())) sink.ObjectBegin(name); }
            else if (id == "ObjectInstance") { if (WantString(name, id.c_str())) sink.ObjectInstance(name); }
            else if (id == "ObjectEnd") sink.ObjectEnd();
            else if (id == "WorldBegin") sink.WorldBegin();
            else if (id == "WorldEnd") sink.WorldEnd();
            else if (id == "AttributeBegin") sink.AttributeBegin();
            else if (id == "AttributeEnd") sink.AttributeEnd();
            else if (id == "TransformBegin") sink.TransformBegin();
            else if (id == "TransformEnd") sink.TransformEnd();
            else if (id == "ReverseOrientation") sink.ReverseOrientation();
            else if (id == "Texture") {
                std::string ty, cls;
                if (!WantString(name, "Texture") || !WantString(ty, "Texture") || !WantString(cls, "Texture")) continue;
                ReadParamList(pl); sink.Texture(name, ty, cls, pl);
            }
            else if (id == "PixelFilter" || id == "Film" || id == "Sampler" || id == "Accelerator" ||
                     id == "SurfaceIntegrator" || id == "VolumeIntegrator" || id == "Camera" ||
                     id == "Material" || id == "LightSource" || id == "AreaLightSource" ||
                     id == "Shape" || id == "Volume") {
                if (!WantString(name, id.c_str())) continue;
                ReadParamList(pl);
                if (id == "PixelFilter") sink.PixelFilter(name, pl);
                else if (id == "Film") sink.Film(name, pl);
                else if (id == "Sampler") sink.Sampler(name, pl);
                else if (id == "Accelerator") sink.Accelerator(name, pl);
                else if (id == "SurfaceIntegrator") sink.SurfaceIntegrator(name, pl);
                else if (id == "VolumeIntegrator") sink.VolumeIntegrator(name, pl);
                else if (id == "Camera") sink.Camera(name, pl);
                else if (id == "Material") sink.Material(name, pl);
                else if (id == "LightSource") sink.LightSource(name, pl);
                else if (id == "AreaLightSource") sink.AreaLightSource(name, pl);
                else if (id == "Shape") sink.Shape(name, pl);
                else sink.Volume(name, pl);
            }
            else Report(1, "parse error: unknown directive \"" + id + "\"");
        }
    }
};

}  // namespace pbrthip
