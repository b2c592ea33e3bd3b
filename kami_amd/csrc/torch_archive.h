// torch_archive.h — reads the checkpoints the reference writes with NN::write (kami/nn/nn.cpp:189-202):
//     serialize::OutputArchive a; mod->save(a); a.write("generation", IValue(generation)); a.save_to(path);
// A libtorch archive is a zip; `<root>/data.pkl` is a protocol-2 pickle of the module tree whose leaves are
// tensors (torch._utils._rebuild_tensor_v2 over a persistent storage id), and the tensor bytes are the STORED
// zip members `<root>/data/<id>`.  No libtorch here and nothing from the file is executed: a zip
// central-directory walk plus a stack machine for the opcodes libtorch's pickler emits; anything else is an
// error.  Python twin: kami_amd/torch_archive.py.  Plain C++17, host only.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace kh_archive {

struct Tensor {
    std::vector<int64_t> shape;
    std::vector<float> data;         // fp32 tensors only (every parameter / statistic of the kami network)
};

struct Checkpoint {
    std::map<std::string, Tensor> tensors;     // dotted names: "residual0.conv1.weight"
    std::map<std::string, int64_t> ints;       // top-level integer attributes: "generation"
};

namespace detail {

[[noreturn]] inline void bad(const std::string& what) { throw std::runtime_error("torch archive: " + what); }

struct Member { uint64_t offset = 0, size = 0; uint16_t method = 0; };

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | (uint64_t)rd32(p + 4) << 32; }

// name -> (offset of the member's bytes, size, method) from the central directory (zip64 extra fields honoured)
inline std::map<std::string, Member> zip_members(const std::vector<uint8_t>& z)
{
    const size_t n = z.size();
    if (n < 22) bad("file too short");
    size_t eocd = n - 22;
    for (;; --eocd) {
        if (rd32(&z[eocd]) == 0x06054b50) break;
        if (eocd == 0 || n - eocd > 65557) bad("no end-of-central-directory record");
    }
    uint64_t count = rd16(&z[eocd + 10]), cdsize = rd32(&z[eocd + 12]), cdoff = rd32(&z[eocd + 16]);
    if (count == 0xffff || cdoff == 0xffffffffu) {                       // zip64
        if (eocd < 20 || rd32(&z[eocd - 20]) != 0x07064b50) bad("zip64 locator missing");
        const uint64_t e64 = rd64(&z[eocd - 20 + 8]);
        if (e64 > n || n - e64 < 56 || rd32(&z[e64]) != 0x06064b50) bad("zip64 end record missing");
        count = rd64(&z[e64 + 32]); cdsize = rd64(&z[e64 + 40]); cdoff = rd64(&z[e64 + 48]);
    }
    if (cdoff > n || cdsize > n - cdoff) bad("central directory out of range");      // (no sums of untrusted 64-bit fields)
    std::map<std::string, Member> out;
    size_t p = cdoff;
    for (uint64_t i = 0; i < count; ++i) {
        if (p + 46 > n || rd32(&z[p]) != 0x02014b50) bad("bad central directory entry");
        Member m;
        m.method = rd16(&z[p + 10]);
        uint64_t csize = rd32(&z[p + 20]), usize = rd32(&z[p + 24]), lho = rd32(&z[p + 42]);
        const size_t nl = rd16(&z[p + 28]), xl = rd16(&z[p + 30]), cl = rd16(&z[p + 32]);
        if (p + 46 + nl + xl + cl > n) bad("central directory entry out of range");
        std::string name((const char*)&z[p + 46], nl);
        for (size_t x = p + 46 + nl, xe = x + xl; x + 4 <= xe;) {       // zip64 extended information
            const uint16_t id = rd16(&z[x]), len = rd16(&z[x + 2]);
            if (id == 1) {
                size_t q = x + 4;
                if (usize == 0xffffffffu && q + 8 <= xe) { usize = rd64(&z[q]); q += 8; }
                if (csize == 0xffffffffu && q + 8 <= xe) { csize = rd64(&z[q]); q += 8; }
                if (lho == 0xffffffffu && q + 8 <= xe) { lho = rd64(&z[q]); q += 8; }
            }
            x += 4 + len;
        }
        if (lho > n || n - lho < 30 || rd32(&z[lho]) != 0x04034b50) bad("bad local header for " + name);
        m.offset = lho + 30 + rd16(&z[lho + 26]) + rd16(&z[lho + 28]);
        m.size = m.method == 0 ? usize : csize;
        if (m.offset > n || m.size > n - m.offset) bad("member " + name + " out of range");
        out[name] = m;
        p += 46 + nl + xl + cl;
    }
    return out;
}

// ---- values of the pickle machine
struct Value;
using VP = std::shared_ptr<Value>;
struct Value {
    enum Kind { NONE, INT, BOOL, FLOAT, STR, TUPLE, LIST, DICT, GLOBAL, OBJECT, STORAGE, TENSOR, MARK } kind = NONE;
    int64_t i = 0;
    double f = 0;
    std::string s, s2;                                   // STR; GLOBAL: module / name; STORAGE: dtype / key
    std::vector<VP> items;                               // TUPLE / LIST
    std::vector<std::pair<VP, VP>> dict;                 // DICT and OBJECT state (insertion order)
    std::vector<int64_t> size, stride;                   // TENSOR
    int64_t offset = 0;
    // Hostile files: `depth` bounds the nesting (the shared_ptr graph is torn down recursively: 300 000 nested TUPLE1s
    // used to overflow the stack in the destructor), `sealed` is set once a value has become a child of a container
    // and forbids changing it afterwards — what libtorch's pickler never does, and the only way to close a cycle.
    int depth = 0;
    bool sealed = false;
};
constexpr int MAX_NESTING = 64;
inline VP mk(Value::Kind k) { auto v = std::make_shared<Value>(); v->kind = k; return v; }

inline VP unpickle(const uint8_t* d, size_t n)
{
    std::vector<VP> st;
    std::map<uint32_t, VP> memo;
    size_t i = 0;
    auto need = [&](size_t k) { if (i + k > n) bad("pickle truncated"); };
    auto pop = [&]() { if (st.empty()) bad("pickle stack underflow"); VP v = st.back(); st.pop_back(); return v; };
    auto pop_mark = [&]() {
        size_t k = st.size();
        while (k > 0 && st[k - 1]->kind != Value::MARK) --k;
        if (k == 0) bad("pickle: MARK missing");
        std::vector<VP> items(st.begin() + k, st.end());
        st.resize(k - 1);
        return items;
    };
    auto integer = [&](int64_t v) { VP x = mk(Value::INT); x->i = v; st.push_back(x); };
    // child -> container: depth bookkeeping, nesting bound, and the child may not be changed any more
    auto adopt = [&](Value& box, const VP& child) {
        if (child.get() == &box) bad("pickle: a container inside itself");
        if (child->depth + 1 > MAX_NESTING) bad("pickle: nesting deeper than 64 levels");
        if (child->depth + 1 > box.depth) box.depth = child->depth + 1;
        child->sealed = true;
    };
    auto open_box = [&](Value::Kind k, const char* what) -> Value& {
        if (st.empty() || st.back()->kind != k) bad(std::string("pickle: ") + what);
        if (st.back()->sealed) bad(std::string("pickle: ") + what + " on a container that is already part of another");
        return *st.back();
    };
    if (n > ((size_t)64 << 20)) bad("pickle larger than 64 MB");
    while (i < n) {
        const uint8_t op = d[i++];
        switch (op) {
        case 0x80: need(1); if (d[i] > 5) bad("pickle protocol"); ++i; break;                                   // PROTO
        case 0x63: {                                                                                         // GLOBAL
            const void* e1 = memchr(d + i, '\n', n - i);
            if (!e1) bad("pickle: GLOBAL");
            const size_t a = (const uint8_t*)e1 - d;
            const void* e2 = memchr(d + a + 1, '\n', n - a - 1);
            if (!e2) bad("pickle: GLOBAL");
            const size_t b = (const uint8_t*)e2 - d;
            VP g = mk(Value::GLOBAL);
            g->s.assign((const char*)d + i, a - i); g->s2.assign((const char*)d + a + 1, b - a - 1);
            st.push_back(g); i = b + 1; break;
        }
        case 0x71: need(1); if (st.empty()) bad("pickle: BINPUT"); memo[d[i]] = st.back(); ++i; break;
        case 0x72: need(4); if (st.empty()) bad("pickle: LONG_BINPUT"); memo[rd32(d + i)] = st.back(); i += 4; break;
        case 0x68: { need(1); auto it = memo.find(d[i]); if (it == memo.end()) bad("pickle: BINGET"); st.push_back(it->second); ++i; break; }
        case 0x6a: { need(4); auto it = memo.find(rd32(d + i)); if (it == memo.end()) bad("pickle: LONG_BINGET"); st.push_back(it->second); i += 4; break; }
        case 0x29: st.push_back(mk(Value::TUPLE)); break;
        case 0x7d: st.push_back(mk(Value::DICT)); break;
        case 0x5d: st.push_back(mk(Value::LIST)); break;
        case 0x28: st.push_back(mk(Value::MARK)); break;
        case 0x58: { need(4); const uint32_t ln = rd32(d + i); i += 4; need(ln); VP s = mk(Value::STR); s->s.assign((const char*)d + i, ln); st.push_back(s); i += ln; break; }
        case 0x4b: need(1); integer(d[i]); ++i; break;
        case 0x4d: need(2); integer(rd16(d + i)); i += 2; break;
        case 0x4a: need(4); integer((int32_t)rd32(d + i)); i += 4; break;
        case 0x8a: {                                                                                         // LONG1
            need(1); const uint8_t ln = d[i++]; need(ln);
            if (ln > 8) bad("pickle: integer too wide");
            int64_t v = 0;
            for (int k = 0; k < ln; ++k) v |= (int64_t)d[i + k] << (8 * k);
            if (ln && ln < 8 && (d[i + ln - 1] & 0x80)) v |= -((int64_t)1 << (8 * ln));
            integer(v); i += ln; break;
        }
        case 0x47: { need(8); uint64_t b = 0; for (int k = 0; k < 8; ++k) b = b << 8 | d[i + k]; VP f = mk(Value::FLOAT); memcpy(&f->f, &b, 8); st.push_back(f); i += 8; break; }
        case 0x88: { VP b = mk(Value::BOOL); b->i = 1; st.push_back(b); break; }
        case 0x89: st.push_back(mk(Value::BOOL)); break;
        case 0x4e: st.push_back(mk(Value::NONE)); break;
        case 0x74: { VP t = mk(Value::TUPLE); t->items = pop_mark(); for (auto& v : t->items) adopt(*t, v); st.push_back(t); break; }
        case 0x85: case 0x86: case 0x87: {
            const size_t k = op - 0x84;
            if (st.size() < k) bad("pickle: TUPLEn");
            VP t = mk(Value::TUPLE); t->items.assign(st.end() - k, st.end()); st.resize(st.size() - k);
            for (auto& v : t->items) adopt(*t, v);
            st.push_back(t); break;
        }
        case 0x81: {                                                                                         // NEWOBJ
            VP args = pop(), cls = pop();
            if (cls->kind != Value::GLOBAL || cls->s.rfind("__torch__", 0) != 0 || args->kind != Value::TUPLE || !args->items.empty())
                bad("pickle: NEWOBJ of something that is not a script module class");
            st.push_back(mk(Value::OBJECT)); break;
        }
        case 0x51: {                                                                                         // BINPERSID
            VP pid = pop();
            if (pid->kind != Value::TUPLE || pid->items.size() != 5 || pid->items[0]->kind != Value::STR || pid->items[0]->s != "storage" ||
                pid->items[1]->kind != Value::GLOBAL || pid->items[1]->s != "torch" || pid->items[2]->kind != Value::STR ||
                pid->items[4]->kind != Value::INT)
                bad("pickle: unexpected persistent id");
            VP s = mk(Value::STORAGE);
            s->s = pid->items[1]->s2; s->s2 = pid->items[2]->s; s->i = pid->items[4]->i;
            st.push_back(s); break;
        }
        case 0x52: {                                                                                         // REDUCE
            VP args = pop(), fn = pop();
            if (fn->kind != Value::GLOBAL || args->kind != Value::TUPLE) bad("pickle: REDUCE of a non-global");
            if (fn->s == "collections" && fn->s2 == "OrderedDict" && args->items.empty()) { st.push_back(mk(Value::DICT)); break; }
            if (fn->s == "torch._utils" && fn->s2 == "_rebuild_tensor_v2") {
                if (args->items.size() < 4 || args->items[0]->kind != Value::STORAGE || args->items[1]->kind != Value::INT ||
                    args->items[2]->kind != Value::TUPLE || args->items[3]->kind != Value::TUPLE)
                    bad("pickle: malformed tensor");
                VP t = mk(Value::TENSOR);
                t->s = args->items[0]->s; t->s2 = args->items[0]->s2; t->i = args->items[0]->i; t->offset = args->items[1]->i;
                for (auto& v : args->items[2]->items) { if (v->kind != Value::INT) bad("pickle: tensor size"); t->size.push_back(v->i); }
                for (auto& v : args->items[3]->items) { if (v->kind != Value::INT) bad("pickle: tensor stride"); t->stride.push_back(v->i); }
                st.push_back(t); break;
            }
            bad("pickle: refusing to call " + fn->s + "." + fn->s2);
        }
        case 0x75: {                                                                                         // SETITEMS
            auto items = pop_mark();
            Value& box = open_box(Value::DICT, "SETITEMS");
            if (items.size() & 1) bad("pickle: SETITEMS");
            for (size_t k = 0; k < items.size(); k += 2) { adopt(box, items[k]); adopt(box, items[k + 1]); box.dict.emplace_back(items[k], items[k + 1]); }
            break;
        }
        case 0x73: { VP v = pop(), k = pop(); Value& box = open_box(Value::DICT, "SETITEM"); adopt(box, k); adopt(box, v); box.dict.emplace_back(k, v); break; }
        case 0x65: { auto items = pop_mark(); Value& box = open_box(Value::LIST, "APPENDS"); for (auto& v : items) { adopt(box, v); box.items.push_back(v); } break; }
        case 0x61: { VP v = pop(); Value& box = open_box(Value::LIST, "APPEND"); adopt(box, v); box.items.push_back(v); break; }
        case 0x62: {                                                                                         // BUILD
            VP state = pop();
            Value& box = open_box(Value::OBJECT, "BUILD");
            if (state->kind != Value::DICT) bad("pickle: BUILD");
            for (auto& kv : state->dict) { adopt(box, kv.first); adopt(box, kv.second); }     // (the state's children become the object's)
            box.dict = state->dict; break;
        }
        case 0x2e: return pop();                                                                             // STOP
        default: { char b[80]; snprintf(b, sizeof b, "pickle opcode 0x%02x is not one libtorch's module pickler writes", op); bad(b); }
        }
    }
    bad("pickle: no STOP");
}

}  // namespace detail

inline Checkpoint read_checkpoint(const std::string& path)
{
    using namespace detail;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) bad("cannot open " + path);
    std::vector<uint8_t> z;
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (len < 0) { fclose(f); bad("cannot size " + path); }
    z.resize((size_t)len);
    const size_t got = fread(z.data(), 1, z.size(), f);
    fclose(f);
    if (got != z.size()) bad("short read of " + path);
    auto members = zip_members(z);
    std::string root;
    for (auto& kv : members) {
        const std::string& n = kv.first;
        const size_t slash = n.find('/');
        if (slash != std::string::npos && n.compare(slash, std::string::npos, "/data.pkl") == 0) { root = n.substr(0, slash); break; }
    }
    if (root.empty()) bad("no <root>/data.pkl (not a libtorch module archive)");
    const Member& pk = members[root + "/data.pkl"];
    if (pk.method != 0) bad("data.pkl is compressed");
    VP top = unpickle(&z[pk.offset], pk.size);
    if (top->kind != Value::OBJECT) bad("archive root is not a module");
    Checkpoint ck;
    struct Walker {
        const std::vector<uint8_t>& z; std::map<std::string, Member>& members; const std::string& root; Checkpoint& ck;
        void walk(const Value& obj, const std::string& prefix, int depth = 0)
        {
            if (depth > 32) bad("module tree too deep (a cycle?)");        // a memo reference can make a state contain its own object
            for (auto& kv : obj.dict) {
                if (kv.first->kind != Value::STR) bad("attribute name is not a string");
                const std::string name = prefix + kv.first->s;
                const Value& v = *kv.second;
                if (v.kind == Value::OBJECT) walk(v, name + ".", depth + 1);
                else if (v.kind == Value::TENSOR) {
                    if (name.size() > 19 && name.compare(name.size() - 19, 19, "num_batches_tracked") == 0) continue;   // int64 counters nn.cpp never reads
                    if (v.s != "FloatStorage") bad(name + " is not an fp32 tensor");
                    auto it = members.find(root + "/data/" + v.s2);
                    if (it == members.end()) bad("storage " + v.s2 + " missing");
                    if (it->second.method != 0) bad("tensor data is compressed");
                    if (v.stride.size() != v.size.size()) bad(name + ": sizes and strides differ in rank");
                    const uint64_t cap = it->second.size / 4;          // elements the storage holds: every product stays below it
                    uint64_t numel = 1;
                    for (int64_t s : v.size) {
                        if (s < 0) bad("negative size");
                        if (s != 0 && numel > cap / (uint64_t)s) bad(name + " exceeds its storage");
                        numel *= (uint64_t)s;
                    }
                    uint64_t want = 1;                                  // contiguous tensors only (what module.save writes)
                    for (size_t k = v.size.size(); k-- > 0;) { if (v.size[k] != 1 && (uint64_t)v.stride[k] != want) bad(name + " is not contiguous"); want *= (uint64_t)v.size[k]; }
                    if (v.offset < 0 || (uint64_t)v.offset > cap || numel > cap - (uint64_t)v.offset) bad(name + " exceeds its storage");
                    Tensor t;
                    t.shape = v.size;
                    t.data.resize((size_t)numel);
                    memcpy(t.data.data(), &z[it->second.offset + (size_t)v.offset * 4], (size_t)numel * 4);
                    ck.tensors[name] = std::move(t);
                } else if (v.kind == Value::INT && prefix.empty()) ck.ints[name] = v.i;
            }
        }
    } w{ z, members, root, ck };
    w.walk(*top, "");
    return ck;
}

}  // namespace kh_archive
