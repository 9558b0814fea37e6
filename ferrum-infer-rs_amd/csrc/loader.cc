// Checkpoint reader: HF `config.json` → runner config, safetensors shards (mmap), GPTQ tensor fusion.
//
// Mirrors the reference's host-side loader so a real checkpoint directory can be handed to the runner:
//   * shard discovery and the name → shard index         ferrum-quantization/src/native_safetensors.rs:142-195
//   * dtype conversion (f32 / f16 / bf16 → f32, i32 raw)  native_safetensors.rs:197-330
//   * fused GPTQ linears (q|k|v → qkv, gate|up → gate_up: row-interleaved concat along N, one shared g_idx,
//     symmetric 4-bit qzeros canonicalised to 0x77777777)  native_safetensors.rs:887-1000,1242-1246,1288-1324
//   * quantize_config.json or config.json "quantization_config"   native_safetensors.rs:1475-1530, config.rs:26-45
//   * config.json field extraction and per-architecture defaults  ferrum-models/src/definition.rs:225-375,
//     models/llama_family.rs:596-680,733-810, moe_config.rs:91-130
//   * tensor names                                        models/llama_family.rs:900-945, qwen3_moe/load.rs:178-260
// Pure host code: nothing here touches the GPU except through ferrum_hip_model_set_* (runner.hip).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ferrum_hip.h"
#include "common.h"

namespace fh {
namespace json {

struct Value {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;   // insertion order kept

    const Value* get(const std::string& key) const {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_num() const { return kind == Num; }
    bool is_u64() const { return kind == Num && num >= 0.0 && std::floor(num) == num; }
};

struct Parser {
    const char* p;
    const char* end;
    std::string err;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    }
    bool fail(const char* m) {
        if (err.empty()) err = m;
        return false;
    }
    bool parse_string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        p++;
        out.clear();
        while (p < end && *p != '"') {
            char c = *p++;
            if (c != '\\') { out.push_back(c); continue; }
            if (p >= end) return fail("bad escape");
            char e = *p++;
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
                if (end - p < 4) return fail("bad \\u escape");
                unsigned cp = 0;
                for (int i = 0; i < 4; i++) {
                    char h = *p++;
                    cp <<= 4;
                    if (h >= '0' && h <= '9') cp |= h - '0';
                    else if (h >= 'a' && h <= 'f') cp |= h - 'a' + 10;
                    else if (h >= 'A' && h <= 'F') cp |= h - 'A' + 10;
                    else return fail("bad \\u escape");
                }
                if (cp < 0x80) out.push_back((char)cp);
                else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                break;
            }
            default: return fail("bad escape");
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool parse_value(Value& v, int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        char c = *p;
        if (c == '{') {
            v.kind = Value::Obj;
            p++;
            ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                p++;
                Value child;
                if (!parse_value(child, depth + 1)) return false;
                v.obj.emplace_back(std::move(key), std::move(child));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = Value::Arr;
            p++;
            ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                Value child;
                if (!parse_value(child, depth + 1)) return false;
                v.arr.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = Value::Str; return parse_string(v.str); }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = Value::Bool; v.b = true; p += 4; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = Value::Bool; v.b = false; p += 5; return true; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { v.kind = Value::Null; p += 4; return true; }
        // number (also accepts the non-standard NaN/Infinity python's json may emit: treated as null)
        if (end - p >= 3 && !strncmp(p, "NaN", 3)) { p += 3; return true; }
        if (end - p >= 8 && !strncmp(p, "Infinity", 8)) { p += 8; return true; }
        const char* s = p;
        if (p < end && (*p == '-' || *p == '+')) p++;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) p++;
        if (p == s) return fail("unexpected character");
        v.kind = Value::Num;
        v.num = strtod(std::string(s, p).c_str(), nullptr);
        return true;
    }
};

static bool parse(const char* data, size_t len, Value& out, std::string& err) {
    Parser ps{data, data + len, {}};
    if (!ps.parse_value(out)) { err = ps.err; return false; }
    ps.ws();
    if (ps.p != ps.end) { err = "trailing characters"; return false; }
    return true;
}

}  // namespace json

static bool read_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
    fclose(f);
    return got == out.size();
}
static bool file_exists(const std::string& path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

enum class StDtype { F32, F16, BF16, I32, I64, Other };

struct TensorInfo {
    StDtype dtype = StDtype::Other;
    std::vector<int64_t> shape;
    size_t begin = 0, end = 0;   // byte range inside the shard's data section
    int shard = 0;
    size_t count() const {
        size_t c = 1;
        for (int64_t d : shape) c *= (size_t)d;
        return c;
    }
};

struct Shard {
    std::string path;
    int fd = -1;
    const uint8_t* map = nullptr;
    size_t size = 0, data_off = 0;
    ~Shard() {
        if (map) munmap(const_cast<uint8_t*>(map), size);
        if (fd >= 0) close(fd);
    }
};

static float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else {   // subnormal
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) u = sign | 0x7F800000u | (man << 13);
    else u = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace fh

using fh::json::Value;

struct FerrumHipCheckpoint {
    std::string dir;
    std::vector<std::unique_ptr<fh::Shard>> shards;
    std::map<std::string, fh::TensorInfo> index;
    Value config;        // config.json with text_config flattened over the root (Gemma-3 style nesting)
    bool has_quant = false;
    std::string quant_method;
    int bits = 0, group_size = 0;
    bool desc_act = false, sym = false;
};

namespace fh {

static int open_shard(FerrumHipCheckpoint* ck, const std::string& path) {
    auto sh = std::make_unique<Shard>();
    sh->path = path;
    sh->fd = open(path.c_str(), O_RDONLY);
    FH_REQUIRE(sh->fd >= 0, "checkpoint: cannot open %s", path.c_str());
    struct stat st;
    FH_REQUIRE(fstat(sh->fd, &st) == 0 && st.st_size >= 8, "checkpoint: %s is not a safetensors file", path.c_str());
    sh->size = (size_t)st.st_size;
    void* m = mmap(nullptr, sh->size, PROT_READ, MAP_PRIVATE, sh->fd, 0);
    FH_REQUIRE(m != MAP_FAILED, "checkpoint: mmap of %s failed", path.c_str());
    sh->map = static_cast<const uint8_t*>(m);
    uint64_t hlen;
    memcpy(&hlen, sh->map, 8);
    FH_REQUIRE(hlen <= sh->size - 8 && hlen < (1ull << 31), "checkpoint: %s: bad header length", path.c_str());
    sh->data_off = 8 + (size_t)hlen;
    Value hdr;
    std::string err;
    FH_REQUIRE(json::parse(reinterpret_cast<const char*>(sh->map + 8), (size_t)hlen, hdr, err) && hdr.kind == Value::Obj,
               "checkpoint: %s: header json: %s", path.c_str(), err.c_str());
    const int shard_id = (int)ck->shards.size();
    for (const auto& kv : hdr.obj) {
        if (kv.first == "__metadata__") continue;
        const Value* dt = kv.second.get("dtype");
        const Value* shp = kv.second.get("shape");
        const Value* off = kv.second.get("data_offsets");
        FH_REQUIRE(dt && dt->kind == Value::Str && shp && shp->kind == Value::Arr && off && off->kind == Value::Arr && off->arr.size() == 2,
                   "checkpoint: %s: malformed entry '%s'", path.c_str(), kv.first.c_str());
        TensorInfo ti;
        ti.shard = shard_id;
        const std::string& d = dt->str;
        ti.dtype = d == "F32" ? StDtype::F32 : d == "F16" ? StDtype::F16 : d == "BF16" ? StDtype::BF16 : d == "I32" ? StDtype::I32
                 : d == "I64" ? StDtype::I64 : StDtype::Other;
        // JSON numbers are doubles: only finite, non-negative integers below 2^53 are offsets / extents
        auto exact = [](const Value& v) { return v.kind == Value::Num && v.num >= 0.0 && v.num < 9007199254740992.0 && v.num == (double)(uint64_t)v.num; };
        for (const Value& s : shp->arr) {
            FH_REQUIRE(exact(s), "checkpoint: %s: '%s' has a non-integer / negative shape entry", path.c_str(), kv.first.c_str());
            ti.shape.push_back((int64_t)s.num);
        }
        FH_REQUIRE(exact(off->arr[0]) && exact(off->arr[1]), "checkpoint: %s: '%s' data_offsets are not non-negative integers", path.c_str(),
                   kv.first.c_str());
        ti.begin = (size_t)off->arr[0].num;
        ti.end = (size_t)off->arr[1].num;
        FH_REQUIRE(ti.begin <= ti.end && sh->data_off + ti.end <= sh->size, "checkpoint: %s: '%s' outside the file", path.c_str(),
                   kv.first.c_str());
        ck->index[kv.first] = ti;
    }
    ck->shards.push_back(std::move(sh));
    return 0;
}

static const TensorInfo* find(const FerrumHipCheckpoint* ck, const std::string& name) {
    auto it = ck->index.find(name);
    return it == ck->index.end() ? nullptr : &it->second;
}
static const uint8_t* bytes_of(const FerrumHipCheckpoint* ck, const TensorInfo& t) {
    const Shard& s = *ck->shards[t.shard];
    return s.map + s.data_off + t.begin;
}

static int read_f32(const FerrumHipCheckpoint* ck, const std::string& name, std::vector<float>& out, std::vector<int64_t>* shape) {
    const TensorInfo* t = find(ck, name);
    FH_REQUIRE(t, "checkpoint: tensor '%s' not in index", name.c_str());
    const size_t n = t->count();
    const uint8_t* src = bytes_of(ck, *t);
    out.resize(n);
    if (t->dtype == StDtype::F32) {
        FH_REQUIRE(t->end - t->begin == n * 4, "checkpoint: '%s' byte size mismatch", name.c_str());
        memcpy(out.data(), src, n * 4);
    } else if (t->dtype == StDtype::F16 || t->dtype == StDtype::BF16) {
        FH_REQUIRE(t->end - t->begin == n * 2, "checkpoint: '%s' byte size mismatch", name.c_str());
        const bool bf = t->dtype == StDtype::BF16;
        for (size_t i = 0; i < n; i++) {
            uint16_t h;
            memcpy(&h, src + 2 * i, 2);
            out[i] = bf ? bf16_to_f32(h) : f16_to_f32(h);
        }
    } else {
        set_error("checkpoint: '%s': expected F32/F16/BF16", name.c_str());
        return FERRUM_HIP_INVALID;
    }
    if (shape) *shape = t->shape;
    return 0;
}

static int read_i32(const FerrumHipCheckpoint* ck, const std::string& name, std::vector<int32_t>& out, std::vector<int64_t>* shape) {
    const TensorInfo* t = find(ck, name);
    FH_REQUIRE(t, "checkpoint: tensor '%s' not in index", name.c_str());
    FH_REQUIRE(t->dtype == StDtype::I32, "checkpoint: '%s': expected I32", name.c_str());
    const size_t n = t->count();
    FH_REQUIRE(t->end - t->begin == n * 4, "checkpoint: '%s' byte size mismatch", name.c_str());
    out.resize(n);
    memcpy(out.data(), bytes_of(ck, *t), n * 4);
    if (shape) *shape = t->shape;
    return 0;
}

// gptq_g_idx_is_desc_act: anything but the trivial i / group_size order (native_safetensors.rs:1288-1324)
static int validate_g_idx(const FerrumHipCheckpoint* ck, const std::string& name, const std::vector<int32_t>* g_idx, int k,
                          bool* is_desc_act) {
    *is_desc_act = false;
    FH_REQUIRE(!(ck->desc_act && !g_idx), "%s: quantize_config desc_act=true but no g_idx tensor was found", name.c_str());
    if (!g_idx) return 0;
    FH_REQUIRE(ck->group_size > 0, "%s: GPTQ g_idx present but group_size is 0", name.c_str());
    FH_REQUIRE((int)g_idx->size() == k, "%s: g_idx length %zu must match K=%d", name.c_str(), g_idx->size(), k);
    const int groups = (k + ck->group_size - 1) / ck->group_size;
    for (int i = 0; i < k; i++) {
        const int g = (*g_idx)[i];
        FH_REQUIRE(g >= 0 && g < groups, "%s: g_idx[%d]=%d outside expected group range 0..%d", name.c_str(), i, g, groups - 1);
        if (g != i / ck->group_size) *is_desc_act = true;
    }
    return 0;
}

struct FusedGptq {
    std::vector<int32_t> qweight, qzeros, g_idx;
    std::vector<float> scales;
    bool has_g_idx = false;   // a non-trivial act-order that must be honoured
    int k = 0, n = 0;
};

// load_gptq_linear_fused (native_safetensors.rs:887-1000): parts share K; concat along N row by row.
static int read_gptq_fused(const FerrumHipCheckpoint* ck, const std::vector<std::string>& parts, FusedGptq* out) {
    FH_REQUIRE(ck->has_quant && ck->quant_method == "gptq", "GPTQ load requires quantize_config (quant_method=%s)",
               ck->has_quant ? ck->quant_method.c_str() : "none");
    FH_REQUIRE(!parts.empty(), "GPTQ fusion: no parts");
    struct Part { std::vector<int32_t> qw, qz; std::vector<float> sc; int64_t qw_cols, sc_cols, qz_cols; };
    std::vector<Part> ps(parts.size());
    int64_t qw_rows = 0, sc_rows = 0, qz_rows = 0, total_n = 0, total_sc = 0, total_qz = 0;
    std::vector<int32_t> g_idx;
    int with_g = 0;
    for (size_t i = 0; i < parts.size(); i++) {
        std::vector<int64_t> s_qw, s_sc, s_qz;
        if (int rc = read_i32(ck, parts[i] + ".qweight", ps[i].qw, &s_qw)) return rc;
        if (int rc = read_f32(ck, parts[i] + ".scales", ps[i].sc, &s_sc)) return rc;
        if (int rc = read_i32(ck, parts[i] + ".qzeros", ps[i].qz, &s_qz)) return rc;
        if (ck->sym && ck->bits == 4) std::fill(ps[i].qz.begin(), ps[i].qz.end(), 0x77777777);   // canonicalize_gptq_qzeros_for_sym
        FH_REQUIRE(s_qw.size() == 2 && s_sc.size() == 2 && s_qz.size() == 2, "GPTQ fusion '%s': expected 2D tensors", parts[i].c_str());
        if (i == 0) { qw_rows = s_qw[0]; sc_rows = s_sc[0]; qz_rows = s_qz[0]; }
        FH_REQUIRE(s_qw[0] == qw_rows && s_sc[0] == sc_rows && s_qz[0] == qz_rows, "GPTQ fusion row mismatch on '%s'", parts[i].c_str());
        ps[i].qw_cols = s_qw[1]; ps[i].sc_cols = s_sc[1]; ps[i].qz_cols = s_qz[1];
        total_n += s_qw[1]; total_sc += s_sc[1]; total_qz += s_qz[1];
        if (find(ck, parts[i] + ".g_idx")) {
            std::vector<int32_t> gx;
            std::vector<int64_t> s_gx;
            if (int rc = read_i32(ck, parts[i] + ".g_idx", gx, &s_gx)) return rc;
            FH_REQUIRE(s_gx.size() == 1 && s_gx[0] == qw_rows * 8, "GPTQ fusion '%s': g_idx shape incompatible with K=%ld",
                       parts[i].c_str(), (long)(qw_rows * 8));
            if (with_g == 0) g_idx = gx;
            else FH_REQUIRE(g_idx == gx, "GPTQ fusion '%s': g_idx mismatch with first part; fused qkv/gate_up requires identical "
                                         "act-order across parts", parts[i].c_str());
            with_g++;
        }
    }
    FH_REQUIRE(with_g == 0 || with_g == (int)parts.size(), "GPTQ fusion requires all parts to carry g_idx when any part does");
    FH_REQUIRE(total_sc == total_n && total_qz * 8 == total_n, "GPTQ fusion: scales/qzeros widths do not match N=%ld", (long)total_n);
    out->k = (int)(qw_rows * 8);
    out->n = (int)total_n;
    out->qweight.clear(); out->scales.clear(); out->qzeros.clear();
    out->qweight.reserve((size_t)qw_rows * total_n);
    for (int64_t r = 0; r < qw_rows; r++)
        for (const Part& p : ps) out->qweight.insert(out->qweight.end(), p.qw.begin() + r * p.qw_cols, p.qw.begin() + (r + 1) * p.qw_cols);
    for (int64_t r = 0; r < sc_rows; r++)
        for (const Part& p : ps) out->scales.insert(out->scales.end(), p.sc.begin() + r * p.sc_cols, p.sc.begin() + (r + 1) * p.sc_cols);
    for (int64_t r = 0; r < qz_rows; r++)
        for (const Part& p : ps) out->qzeros.insert(out->qzeros.end(), p.qz.begin() + r * p.qz_cols, p.qz.begin() + (r + 1) * p.qz_cols);
    std::string fused = "GPTQ fusion";
    for (const std::string& p : parts) fused += " " + p;
    bool desc = false;
    if (int rc = validate_g_idx(ck, fused, with_g ? &g_idx : nullptr, out->k, &desc)) return rc;
    out->has_g_idx = desc;
    out->g_idx = desc ? g_idx : std::vector<int32_t>();
    return 0;
}

// "…qkv_proj" / "…gate_up_proj": the fused tensor if the checkpoint has it, else its split parts
static std::vector<std::string> linear_parts(const FerrumHipCheckpoint* ck, const std::string& prefix, const char* fused,
                                             std::initializer_list<const char*> split) {
    if (find(ck, prefix + fused + ".qweight")) return {prefix + fused};
    std::vector<std::string> parts;
    for (const char* s : split) parts.push_back(prefix + s);
    return parts;
}

static double num_or(const Value* v, double dflt) { return v && v->kind == Value::Num ? v->num : dflt; }
static const Value* cfg_get(const FerrumHipCheckpoint* ck, const char* key) { return ck->config.get(key); }

static std::string architecture(const FerrumHipCheckpoint* ck) {
    const Value* a = cfg_get(ck, "architectures");
    if (a && a->kind == Value::Arr && !a->arr.empty() && a->arr[0].kind == Value::Str) return a->arr[0].str;
    const Value* mt = cfg_get(ck, "model_type");
    return mt && mt->kind == Value::Str ? mt->str : "";
}

}  // namespace fh

using namespace fh;

extern "C" {

int ferrum_hip_checkpoint_open(FerrumHipCheckpoint** out, const char* model_dir) {
    FH_REQUIRE(out && model_dir, "checkpoint_open: null argument");
    *out = nullptr;
    auto ck = std::make_unique<FerrumHipCheckpoint>();
    ck->dir = model_dir;
    const std::string d = ck->dir + "/";
    std::vector<std::string> files;
    if (file_exists(d + "model.safetensors")) {
        files.push_back(d + "model.safetensors");
    } else if (file_exists(d + "model.safetensors.index.json")) {
        std::string txt, err;
        FH_REQUIRE(read_file(d + "model.safetensors.index.json", txt), "checkpoint: cannot read the shard index");
        Value idx;
        FH_REQUIRE(json::parse(txt.data(), txt.size(), idx, err), "checkpoint: index json: %s", err.c_str());
        const Value* wm = idx.get("weight_map");
        FH_REQUIRE(wm && wm->kind == Value::Obj, "checkpoint: index missing weight_map");
        for (const auto& kv : wm->obj)
            if (kv.second.kind == Value::Str) files.push_back(d + kv.second.str);
        std::sort(files.begin(), files.end());
        files.erase(std::unique(files.begin(), files.end()), files.end());
    } else {
        set_error("checkpoint: no safetensors files in %s", model_dir);
        return FERRUM_HIP_INVALID;
    }
    for (const std::string& f : files)
        if (int rc = open_shard(ck.get(), f)) return rc;

    // config.json (Gemma-3 style nested text_config is flattened over the root, definition.rs:193-228)
    std::string txt, err;
    if (read_file(d + "config.json", txt)) {
        Value root;
        FH_REQUIRE(json::parse(txt.data(), txt.size(), root, err) && root.kind == Value::Obj, "checkpoint: config.json: %s", err.c_str());
        ck->config = root;
        if (const Value* tc = root.get("text_config"); tc && tc->kind == Value::Obj) {
            for (const auto& kv : tc->obj) {
                bool replaced = false;
                for (auto& own : ck->config.obj)
                    if (own.first == kv.first) { own.second = kv.second; replaced = true; }
                if (!replaced) ck->config.obj.push_back(kv);
            }
        }
    } else {
        ck->config.kind = Value::Obj;
    }
    // quantize_config.json, else config.json "quantization_config" (native_safetensors.rs:1475-1530)
    const Value* qc = nullptr;
    Value qroot;
    if (read_file(d + "quantize_config.json", txt)) {
        FH_REQUIRE(json::parse(txt.data(), txt.size(), qroot, err) && qroot.kind == Value::Obj, "checkpoint: quantize_config.json: %s",
                   err.c_str());
        qc = &qroot;
    } else {
        qc = ck->config.get("quantization_config");
        if (qc && qc->kind != Value::Obj) qc = nullptr;
    }
    if (qc) {
        const Value* m = qc->get("quant_method");
        if (!m) m = qc->get("method");
        std::string method = m && m->kind == Value::Str ? m->str : "none";
        std::transform(method.begin(), method.end(), method.begin(), [](unsigned char c) { return (char)tolower(c); });
        ck->quant_method = method;
        ck->has_quant = method == "gptq" || method == "awq" || method == "gguf";
        ck->bits = (int)num_or(qc->get("bits"), 0);
        ck->group_size = (int)std::max(0.0, num_or(qc->get("group_size"), qc == &qroot ? 0 : 128));
        const Value* da = qc->get("desc_act");
        const Value* sy = qc->get("sym");
        ck->desc_act = da && da->kind == Value::Bool && da->b;
        ck->sym = sy && sy->kind == Value::Bool && sy->b;
    }
    *out = ck.release();
    return 0;
}

int ferrum_hip_checkpoint_close(FerrumHipCheckpoint* ck) {
    delete ck;
    return 0;
}

int ferrum_hip_checkpoint_num_tensors(const FerrumHipCheckpoint* ck) { return ck ? (int)ck->index.size() : 0; }

int ferrum_hip_checkpoint_tensor_info(const FerrumHipCheckpoint* ck, const char* name, int* dtype, int* ndim, int64_t* shape4) {
    FH_REQUIRE(ck && name, "checkpoint_tensor_info: null argument");
    const TensorInfo* t = find(ck, name);
    if (!t) { set_error("checkpoint: tensor '%s' not in index", name); return FERRUM_HIP_INVALID; }
    if (dtype) *dtype = (int)t->dtype;
    if (ndim) *ndim = (int)t->shape.size();
    if (shape4)
        for (size_t i = 0; i < 4; i++) shape4[i] = i < t->shape.size() ? t->shape[i] : 1;
    return 0;
}

int ferrum_hip_checkpoint_read_f32(const FerrumHipCheckpoint* ck, const char* name, float* out, size_t capacity) {
    FH_REQUIRE(ck && name && out, "checkpoint_read_f32: null argument");
    std::vector<float> v;
    if (int rc = read_f32(ck, name, v, nullptr)) return rc;
    FH_REQUIRE(v.size() <= capacity, "checkpoint_read_f32: '%s' has %zu elements, buffer holds %zu", name, v.size(), capacity);
    memcpy(out, v.data(), v.size() * 4);
    return 0;
}

int ferrum_hip_checkpoint_read_i32(const FerrumHipCheckpoint* ck, const char* name, int32_t* out, size_t capacity) {
    FH_REQUIRE(ck && name && out, "checkpoint_read_i32: null argument");
    std::vector<int32_t> v;
    if (int rc = read_i32(ck, name, v, nullptr)) return rc;
    FH_REQUIRE(v.size() <= capacity, "checkpoint_read_i32: '%s' has %zu elements, buffer holds %zu", name, v.size(), capacity);
    memcpy(out, v.data(), v.size() * 4);
    return 0;
}

int ferrum_hip_checkpoint_quant_config(const FerrumHipCheckpoint* ck, int* is_gptq, int* bits, int* group_size, int* desc_act,
                                       int* sym) {
    FH_REQUIRE(ck, "checkpoint_quant_config: null argument");
    if (is_gptq) *is_gptq = ck->has_quant && ck->quant_method == "gptq";
    if (bits) *bits = ck->bits;
    if (group_size) *group_size = ck->group_size;
    if (desc_act) *desc_act = ck->desc_act;
    if (sym) *sym = ck->sym;
    return 0;
}

/* Fused GPTQ read.  `parts` are tensor-name stems ("model.layers.0.self_attn.q_proj", …).  Call with null outputs to get
 * k / n / has_g_idx, then with buffers of qweight [k/8·n], scales [k/group·n], qzeros [k/group·n/8], g_idx [k]. */
int ferrum_hip_checkpoint_read_gptq_fused(const FerrumHipCheckpoint* ck, const char* const* parts, int num_parts, int32_t* qweight,
                                          float* scales, int32_t* qzeros, int32_t* g_idx, int* k, int* n, int* has_g_idx) {
    FH_REQUIRE(ck && parts && num_parts > 0, "checkpoint_read_gptq_fused: bad argument");
    std::vector<std::string> ps;
    for (int i = 0; i < num_parts; i++) ps.emplace_back(parts[i]);
    FusedGptq f;
    if (int rc = read_gptq_fused(ck, ps, &f)) return rc;
    if (k) *k = f.k;
    if (n) *n = f.n;
    if (has_g_idx) *has_g_idx = f.has_g_idx;
    if (qweight) memcpy(qweight, f.qweight.data(), f.qweight.size() * 4);
    if (scales) memcpy(scales, f.scales.data(), f.scales.size() * 4);
    if (qzeros) memcpy(qzeros, f.qzeros.data(), f.qzeros.size() * 4);
    if (g_idx && f.has_g_idx) memcpy(g_idx, f.g_idx.data(), f.g_idx.size() * 4);
    return 0;
}

/* config.json → runner config.  Fills the architecture fields; max_seq_len is min(max_position_embeddings, max_seq_len_cap)
 * (cap 0 = no cap); kv_num_blocks / max_seqs / max_tokens / tp_* are left to the caller (zeroed). */
int ferrum_hip_checkpoint_model_config(const FerrumHipCheckpoint* ck, int max_seq_len_cap, FerrumHipModelConfig* cfg,
                                       char* arch_out, size_t arch_cap, int* tied_lm_head) {
    FH_REQUIRE(ck && cfg, "checkpoint_model_config: null argument");
    memset(cfg, 0, sizeof(*cfg));
    const std::string arch = architecture(ck);
    if (arch_out && arch_cap) snprintf(arch_out, arch_cap, "%s", arch.c_str());
    enum { Llama, Qwen3, Qwen3Moe, Mistral, Gemma3, Qwen2 } fam;
    if (arch == "LlamaForCausalLM" || arch == "llama") fam = Llama;
    else if (arch == "Gemma3ForCausalLM" || arch == "Gemma3ForConditionalGeneration" || arch == "gemma3_text" || arch == "gemma3") fam = Gemma3;
    else if (arch == "Qwen3ForCausalLM" || arch == "qwen3") fam = Qwen3;
    else if (arch == "Qwen2ForCausalLM" || arch == "qwen2") fam = Qwen2;      // structurally Llama + q/k/v biases, θ default 1e6
    else if (arch == "Qwen3MoeForCausalLM" || arch == "qwen3_moe") fam = Qwen3Moe;
    else if (arch == "MistralForCausalLM" || arch == "mistral") fam = Mistral;
    else {
        set_error("checkpoint: architecture '%s' is not supported by the HIP runner (Llama, Mistral, Qwen2, Qwen3, Qwen3-MoE, Gemma-3)", arch.c_str());
        return FERRUM_HIP_UNSUPPORTED;
    }
    auto u = [&](const char* key, const char* alt, double dflt) {
        const Value* v = cfg_get(ck, key);
        if (!(v && v->is_u64()) && alt) v = cfg_get(ck, alt);
        return v && v->is_u64() ? v->num : dflt;
    };
    cfg->hidden = (int)u("hidden_size", nullptr, 4096);
    cfg->intermediate = (int)u("intermediate_size", "ffn_dim", 11008);
    cfg->vocab = (int)u("vocab_size", nullptr, 0);
    cfg->num_layers = (int)u("num_hidden_layers", "n_layer", 32);
    cfg->num_heads = (int)u("num_attention_heads", "n_head", 32);
    cfg->num_kv_heads = (int)u("num_key_value_heads", nullptr, cfg->num_heads);
    cfg->head_dim = (int)u("head_dim", nullptr, cfg->num_heads ? cfg->hidden / cfg->num_heads : 0);
    int max_pos = (int)u("max_position_embeddings", "n_positions", 2048);
    cfg->max_seq_len = max_seq_len_cap > 0 ? std::min(max_pos, max_seq_len_cap) : max_pos;
    const Value* eps = cfg_get(ck, "rms_norm_eps");
    if (!(eps && eps->is_num())) eps = cfg_get(ck, "layer_norm_eps");
    if (!(eps && eps->is_num())) eps = cfg_get(ck, "layer_norm_epsilon");
    cfg->rms_eps = (float)num_or(eps, 1e-6);
    const Value* act = cfg_get(ck, "hidden_act");
    if (!act) act = cfg_get(ck, "hidden_activation");
    cfg->activation = act && act->kind == Value::Str && act->str == "gelu_pytorch_tanh" ? 1 : 0;
    cfg->has_qk_norm = fam == Qwen3 || fam == Qwen3Moe || fam == Gemma3;
    // rope theta: checkpoint value, else the family default (llama_family.rs:654-680)
    const Value* th = cfg_get(ck, "rope_theta");
    if (!(th && th->is_num())) th = cfg_get(ck, "rotary_emb_base");
    if (!(th && th->is_num()))
        if (const Value* rp = cfg_get(ck, "rope_parameters")) th = rp->get("rope_theta");
    cfg->rope_theta = num_or(th, fam == Llama ? 500000.0 : fam == Mistral ? 10000.0 : 1000000.0);
    // rope scaling (llama_family.rs:768-810): linear{factor} | llama3{factor, low, high, original_max}
    if (const Value* rs = cfg_get(ck, "rope_scaling"); rs && rs->kind == Value::Obj) {
        const Value* ty = rs->get("rope_type");
        if (!ty) ty = rs->get("type");
        const std::string type = ty && ty->kind == Value::Str ? ty->str : "";
        const double factor = num_or(rs->get("factor"), 0.0);
        if (type == "linear" && factor > 0.0) {
            cfg->rope_scaling_kind = 1;
            cfg->rope_p0 = factor;
        } else if (type == "llama3") {
            const double lo = num_or(rs->get("low_freq_factor"), 0.0), hi = num_or(rs->get("high_freq_factor"), 0.0);
            double orig = num_or(rs->get("original_max_position_embeddings"), 0.0);
            if (orig <= 0.0) orig = num_or(cfg_get(ck, "original_max_position_embeddings"), 8192.0);
            if (factor > 0.0 && lo > 0.0 && hi > lo && orig > 0.0) {
                cfg->rope_scaling_kind = 2;
                cfg->rope_p0 = factor; cfg->rope_p1 = lo; cfg->rope_p2 = hi; cfg->rope_p3 = orig;
            }
        }
    }
    if (const Value* sw = cfg_get(ck, "sliding_window"); sw && sw->is_u64()) cfg->sliding_window = (int)sw->num;
    if (fam == Qwen3Moe) {
        const Value* ne = cfg_get(ck, "num_experts");
        const Value* mi = cfg_get(ck, "moe_intermediate_size");
        FH_REQUIRE(ne && ne->is_u64(), "qwen3_moe config.json missing num_experts");
        FH_REQUIRE(mi && mi->is_u64(), "qwen3_moe config.json missing moe_intermediate_size");
        cfg->num_experts = (int)ne->num;
        cfg->expert_inter = (int)mi->num;
        cfg->top_k = (int)u("num_experts_per_tok", nullptr, 8);
        const Value* nt = cfg_get(ck, "norm_topk_prob");
        cfg->norm_topk_prob = nt && nt->kind == Value::Bool ? nt->b : 1;
        cfg->intermediate = 0;
    }
    if (fam == Gemma3) {
        // gemma3_from_def (llama_family.rs:683-703): 5:1 local/global schedule, own local θ, sandwich norms, √hidden embeddings
        // rounded through bf16 like HF; the norm-weight folds (+1, q_norm × √(head_dim/query_pre_attn_scalar)) happen at load.
        cfg->sliding_window_pattern = (int)num_or(cfg_get(ck, "sliding_window_pattern"), 6.0);
        cfg->rope_local_theta = num_or(cfg_get(ck, "rope_local_base_freq"), 10000.0);
        cfg->sandwich_norms = 1;
        float es = (float)std::sqrt((double)cfg->hidden);
        uint32_t bits;
        memcpy(&bits, &es, 4);
        bits = (bits + 0x7FFFu + ((bits >> 16) & 1u)) & 0xFFFF0000u;     // bf16 round-to-nearest-even (finite input)
        memcpy(&es, &bits, 4);
        cfg->embed_scale = es;
    }
    cfg->group_size = ck->group_size > 0 ? ck->group_size : 128;
    cfg->tp_world = 1;
    if (tied_lm_head) *tied_lm_head = find(ck, "lm_head.weight") == nullptr;
    return 0;
}

/* Hand every weight of the checkpoint to a created (not yet finalized) runner model, with the reference's tensor names
 * and fusions.  The model must have been created from ferrum_hip_checkpoint_model_config's dimensions. */
int ferrum_hip_model_load_checkpoint(FerrumHipModel* model, const FerrumHipCheckpoint* ck) {
    FH_REQUIRE(model && ck, "model_load_checkpoint: null argument");
    FerrumHipModelConfig c;
    int tied = 0;
    if (int rc = ferrum_hip_checkpoint_model_config(ck, 0, &c, nullptr, 0, &tied)) return rc;
    FH_REQUIRE(ck->has_quant && ck->quant_method == "gptq" && ck->bits == 4,
               "model_load_checkpoint: the HIP runner loads GPTQ INT4 checkpoints (quant_method=%s bits=%d)",
               ck->has_quant ? ck->quant_method.c_str() : "none", ck->bits);
    std::vector<float> buf;
    // fold_norm_weight (llama_family.rs:949-967): Gemma RMSNorm is x̂·(1+w) → w' = (w + 1)·scale at load time
    const bool unit_offset = c.sandwich_norms != 0;
    double q_scale = 1.0;
    if (unit_offset)
        if (const Value* s = cfg_get(ck, "query_pre_attn_scalar"); s && s->is_num() && s->num > 0.0)
            q_scale = std::sqrt((double)c.head_dim / s->num);
    auto fold = [&](float scale) {
        if (!unit_offset && scale == 1.0f) return;
        for (float& x : buf) x = (x + (unit_offset ? 1.0f : 0.0f)) * scale;
    };
    auto dense_global = [&](int which, const std::string& name, bool is_norm) -> int {
        if (int rc = read_f32(ck, name, buf, nullptr)) return rc;
        if (is_norm) fold(1.0f);
        return ferrum_hip_model_set_global_f32(model, which, buf.data());
    };
    if (int rc = dense_global(0, "model.embed_tokens.weight", false)) return rc;
    if (!tied)
        if (int rc = dense_global(1, "lm_head.weight", false)) return rc;
    if (int rc = dense_global(2, "model.norm.weight", true)) return rc;
    FusedGptq f;
    auto gptq = [&](int layer, int which, int expert, const std::vector<std::string>& parts) -> int {
        if (int rc = read_gptq_fused(ck, parts, &f)) return rc;
        return ferrum_hip_model_set_gptq(model, layer, which, expert, f.qweight.data(), f.scales.data(), f.qzeros.data(),
                                         f.has_g_idx ? f.g_idx.data() : nullptr, f.k, f.n);
    };
    for (int li = 0; li < c.num_layers; li++) {
        const std::string p = "model.layers." + std::to_string(li) + ".";
        auto dense_layer = [&](int which, const std::string& name, float norm_scale = 0.0f) -> int {
            if (int rc = read_f32(ck, p + name, buf, nullptr)) return rc;
            if (norm_scale != 0.0f) fold(norm_scale);
            return ferrum_hip_model_set_layer_dense_f32(model, li, which, buf.data());
        };
        if (int rc = dense_layer(0, "input_layernorm.weight", 1.0f)) return rc;
        if (c.sandwich_norms) {   // llama_family.rs:913-921: Gemma naming
            if (int rc = dense_layer(1, "pre_feedforward_layernorm.weight", 1.0f)) return rc;
            if (int rc = dense_layer(5, "post_attention_layernorm.weight", 1.0f)) return rc;
            if (int rc = dense_layer(6, "post_feedforward_layernorm.weight", 1.0f)) return rc;
        } else {
            if (int rc = dense_layer(1, "post_attention_layernorm.weight", 1.0f)) return rc;
        }
        if (c.has_qk_norm && find(ck, p + "self_attn.q_norm.weight") && find(ck, p + "self_attn.k_norm.weight")) {
            if (int rc = dense_layer(2, "self_attn.q_norm.weight", (float)q_scale)) return rc;
            if (int rc = dense_layer(3, "self_attn.k_norm.weight", 1.0f)) return rc;
        }
        // cat_optional_biases (native_safetensors.rs:246-290): q|k|v biases are all-or-none and concatenate like the weights
        {
            const std::vector<std::string> bparts = find(ck, p + "self_attn.qkv_proj.bias")
                ? std::vector<std::string>{p + "self_attn.qkv_proj.bias"}
                : std::vector<std::string>{p + "self_attn.q_proj.bias", p + "self_attn.k_proj.bias", p + "self_attn.v_proj.bias"};
            int present = 0;
            for (const std::string& b : bparts) present += find(ck, b) != nullptr;
            FH_REQUIRE(present == 0 || present == (int)bparts.size(),
                       "dense fusion bias mix in layer %d: some of q/k/v carry a bias and others do not", li);
            if (present) {
                std::vector<float> fused, part;
                for (const std::string& b : bparts) {
                    if (int rc = read_f32(ck, b, part, nullptr)) return rc;
                    fused.insert(fused.end(), part.begin(), part.end());
                }
                FH_REQUIRE((int)fused.size() == (c.num_heads + 2 * c.num_kv_heads) * c.head_dim,
                           "dense fusion bias length %zu != qkv width (layer %d)", fused.size(), li);
                if (int rc = ferrum_hip_model_set_layer_dense_f32(model, li, 7, fused.data())) return rc;
            }
        }
        if (int rc = gptq(li, 0, 0, linear_parts(ck, p + "self_attn.", "qkv_proj", {"q_proj", "k_proj", "v_proj"}))) return rc;
        if (int rc = gptq(li, 1, 0, {p + "self_attn.o_proj"})) return rc;
        if (c.num_experts > 0) {
            if (int rc = dense_layer(4, "mlp.gate.weight")) return rc;   // router: not a norm, no fold
            for (int e = 0; e < c.num_experts; e++) {
                const std::string ep = p + "mlp.experts." + std::to_string(e) + ".";
                if (int rc = gptq(li, 4, e, linear_parts(ck, ep, "gate_up_proj", {"gate_proj", "up_proj"}))) return rc;
                if (int rc = gptq(li, 5, e, {ep + "down_proj"})) return rc;
            }
        } else {
            if (int rc = gptq(li, 2, 0, linear_parts(ck, p + "mlp.", "gate_up_proj", {"gate_proj", "up_proj"}))) return rc;
            if (int rc = gptq(li, 3, 0, {p + "mlp.down_proj"})) return rc;
        }
    }
    return 0;
}

}  // extern "C"
