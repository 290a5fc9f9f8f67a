// paramset.h -- typed name/value parameter bags, the host mirror of the reference's ParamSet
// (core/paramset.h:57-133, core/paramset.cpp).  Same method names, same lookup semantics:
// FindOne* returns the default unless an item of that name AND exactly one value exists
// (paramset.h:44-55 LOOKUP_ONE macro); Find* returns a pointer + count; every successful lookup
// marks the item so ReportUnused() can warn about misspelt parameters (paramset.cpp:242-254);
// adding a name twice replaces the earlier item (Add* erase first, paramset.cpp:60-130).
#pragma once
#include <string>
#include <vector>
#include "scene_parser.h"

namespace pbrthip {

struct Float3 { float x, y, z; };

void Warning(const char *fmt, ...);
void Error(const char *fmt, ...);

class ParamSet {
  public:
    ParamSet() {}
    explicit ParamSet(const ParamList &pl) {
        for (const Param &p : pl) {
            switch (p.type) {
            case ParamType::Float: AddFloat(p.name, p.nums.data(), int(p.nums.size())); break;
            case ParamType::Int: {
                std::vector<int> iv(p.nums.size());
                for (size_t i = 0; i < iv.size(); ++i) iv[i] = int(p.nums[i]);      // pbrtparse.y:484-492
                AddInt(p.name, iv.data(), int(iv.size())); break; }
            case ParamType::Bool: {
                std::vector<bool> bv;
                for (size_t i = 0; i < p.strs.size(); ++i) {
                    const std::string &s0 = p.strs[0];                                  // pbrtparse.y:499 (first string for all)
                    if (s0 == "true") bv.push_back(true); else if (s0 == "false") bv.push_back(false);
                    else { Warning("Value \"%s\" unknown for boolean parameter \"%s\".Using \"false\".", s0.c_str(), p.name.c_str()); bv.push_back(false); }
                }
                Item it; it.kind = K_BOOL; it.name = p.name; for (bool b : bv) it.ints.push_back(b ? 1 : 0); it.count = int(bv.size());
                Put(it); break; }
            case ParamType::Point: AddTriples(K_POINT, p.name, p.nums); break;
            case ParamType::Vector: AddTriples(K_VECTOR, p.name, p.nums); break;
            case ParamType::Normal: AddTriples(K_NORMAL, p.name, p.nums); break;
            case ParamType::Color: AddTriples(K_SPECTRUM, p.name, p.nums); break;
            case ParamType::String: AddString(p.name, p.strs.data(), int(p.strs.size())); break;
            case ParamType::Texture:
                if (p.strs.size() == 1) AddTexture(p.name, p.strs[0]);
                else Error("Only one string allowed for \"texture\" parameter \"%s\"", p.name.c_str());
                break;
            }
        }
    }
    void AddFloat(const std::string &n, const float *d, int c = 1) { Item it; it.kind = K_FLOAT; it.name = n; it.floats.assign(d, d + c); it.count = c; Put(it); }
    void AddInt(const std::string &n, const int *d, int c = 1) { Item it; it.kind = K_INT; it.name = n; it.ints.assign(d, d + c); it.count = c; Put(it); }
    void AddBool(const std::string &n, const bool *d, int c = 1) { Item it; it.kind = K_BOOL; it.name = n; for (int i = 0; i < c; ++i) it.ints.push_back(d[i]); it.count = c; Put(it); }
    void AddPoint(const std::string &n, const Float3 *d, int c = 1) { AddTriples(K_POINT, n, std::vector<float>((const float *)d, (const float *)d + 3 * c)); }
    void AddVector(const std::string &n, const Float3 *d, int c = 1) { AddTriples(K_VECTOR, n, std::vector<float>((const float *)d, (const float *)d + 3 * c)); }
    void AddNormal(const std::string &n, const Float3 *d, int c = 1) { AddTriples(K_NORMAL, n, std::vector<float>((const float *)d, (const float *)d + 3 * c)); }
    void AddSpectrum(const std::string &n, const Float3 *d, int c = 1) { AddTriples(K_SPECTRUM, n, std::vector<float>((const float *)d, (const float *)d + 3 * c)); }
    void AddString(const std::string &n, const std::string *d, int c = 1) { Item it; it.kind = K_STRING; it.name = n; it.strs.assign(d, d + c); it.count = c; Put(it); }
    void AddTexture(const std::string &n, const std::string &v) { Item it; it.kind = K_TEXTURE; it.name = n; it.strs.push_back(v); it.count = 1; Put(it); }

    float FindOneFloat(const std::string &n, float d) const { const Item *it = One(K_FLOAT, n); return it ? it->floats[0] : d; }
    int FindOneInt(const std::string &n, int d) const { const Item *it = One(K_INT, n); return it ? it->ints[0] : d; }
    bool FindOneBool(const std::string &n, bool d) const { const Item *it = One(K_BOOL, n); return it ? it->ints[0] != 0 : d; }
    Float3 FindOnePoint(const std::string &n, Float3 d) const { return OneTriple(K_POINT, n, d); }
    Float3 FindOneVector(const std::string &n, Float3 d) const { return OneTriple(K_VECTOR, n, d); }
    Float3 FindOneNormal(const std::string &n, Float3 d) const { return OneTriple(K_NORMAL, n, d); }
    Float3 FindOneSpectrum(const std::string &n, Float3 d) const { return OneTriple(K_SPECTRUM, n, d); }
    std::string FindOneString(const std::string &n, const std::string &d) const { const Item *it = One(K_STRING, n); return it ? it->strs[0] : d; }
    std::string FindTexture(const std::string &n) const { const Item *it = One(K_TEXTURE, n); return it ? it->strs[0] : std::string(); }
    const float *FindFloat(const std::string &n, int *c) const { const Item *it = Any(K_FLOAT, n); if (!it) return nullptr; *c = it->count; return it->floats.data(); }
    const int *FindInt(const std::string &n, int *c) const { const Item *it = Any(K_INT, n); if (!it) return nullptr; *c = it->count; return it->ints.data(); }
    const Float3 *FindPoint(const std::string &n, int *c) const { const Item *it = Any(K_POINT, n); if (!it) return nullptr; *c = it->count; return (const Float3 *)it->floats.data(); }
    const Float3 *FindNormal(const std::string &n, int *c) const { const Item *it = Any(K_NORMAL, n); if (!it) return nullptr; *c = it->count; return (const Float3 *)it->floats.data(); }
    const Float3 *FindVector(const std::string &n, int *c) const { const Item *it = Any(K_VECTOR, n); if (!it) return nullptr; *c = it->count; return (const Float3 *)it->floats.data(); }

    void ReportUnused() const {
        for (const Item &it : items)
            if (!it.looked_up) Warning("Parameter \"%s\" not used", it.name.c_str());
    }
    void Clear() { items.clear(); }

  private:
    enum Kind { K_INT, K_BOOL, K_FLOAT, K_POINT, K_VECTOR, K_NORMAL, K_SPECTRUM, K_STRING, K_TEXTURE };
    struct Item {
        Kind kind; std::string name; int count = 0; mutable bool looked_up = false;
        std::vector<float> floats; std::vector<int> ints; std::vector<std::string> strs;
    };
    std::vector<Item> items;
    void Put(const Item &it) {
        for (size_t i = 0; i < items.size(); ++i)
            if (items[i].kind == it.kind && items[i].name == it.name) { items.erase(items.begin() + i); break; }
        items.push_back(it);
    }
    void AddTriples(Kind k, const std::string &n, const std::vector<float> &v) {
        Item it; it.kind = k; it.name = n; it.count = int(v.size() / 3); it.floats.assign(v.begin(), v.begin() + size_t(it.count) * 3); Put(it);
    }
    const Item *One(Kind k, const std::string &n) const {
        for (const Item &it : items) if (it.kind == k && it.name == n && it.count == 1) { it.looked_up = true; return &it; }
        return nullptr;
    }
    const Item *Any(Kind k, const std::string &n) const {
        for (const Item &it : items) if (it.kind == k && it.name == n) { it.looked_up = true; return &it; }
        return nullptr;
    }
    Float3 OneTriple(Kind k, const std::string &n, Float3 d) const {
        const Item *it = One(k, n); if (!it) return d;
        Float3 r = {it->floats[0], it->floats[1], it->floats[2]}; return r;
    }
};

}  // namespace pbrthip
