// norma_assets.hpp -- local checkpoint assets for the host layer: config.json -> Config,
// tokenizer.json -> special-token ids + byte-level BPE detokeniser, model.safetensors -> tensors.
//
// Stands in for what the reference does after its hf-hub download (src/models/whisper/monolingual.rs):
//   :347      serde_json::from_str::<Config>(config.json)
//   :349      Tokenizer::from_file(tokenizer.json); token_id() lookups :376-384, :419-420 (mod.rs:86-90)
//   :371-373  VarBuilder::from_mmaped_safetensors(model.safetensors)
//   model.rs:147  tokenizer.decode(tokens, skip_special_tokens = true)
// The download itself is out of scope (no network); files are read from a local directory.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace norma {
namespace assets {

// ---- a small JSON reader (objects, arrays, strings with escapes, numbers, literals) ------------------------
struct Json {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;  // insertion order kept (safetensors headers are large)
    const Json *get(const std::string &k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

class JsonParser {
  public:
    JsonParser(const char *p, size_t n) : p_(p), e_(p + n) {}
    bool parse(Json &out) { ws(); bool ok = value(out); ws(); return ok; }
    std::string error;
  private:
    const char *p_, *e_;
    void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) p_++; }
    bool fail(const char *m) { error = m; return false; }
    static void utf8(std::string &s, unsigned cp) {
        if (cp < 0x80) s += (char)cp;
        else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
        else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    }
    bool hex4(unsigned &v) {
        if (e_ - p_ < 4) return false;
        v = 0;
        for (int i = 0; i < 4; i++) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else return false;
        }
        return true;
    }
    bool string(std::string &s) {
        if (p_ >= e_ || *p_ != '"') return fail("expected string");
        p_++;
        while (p_ < e_ && *p_ != '"') {
            if (*p_ == '\\') {
                if (++p_ >= e_) return fail("bad escape");
                char c = *p_++;
                switch (c) {
                    case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
                    case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                    case 'u': {
                        unsigned cp;
                        if (!hex4(cp)) return fail("bad \\u");
                        if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                            p_ += 2; unsigned lo;
                            if (!hex4(lo)) return fail("bad surrogate");
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(s, cp);
                        break;
                    }
                    default: s += c;
                }
            } else s += *p_++;
        }
        if (p_ >= e_) return fail("unterminated string");
        p_++;
        return true;
    }
    bool value(Json &v) {
        if (p_ >= e_) return fail("unexpected end");
        if (*p_ == '{') {
            v.type = Json::Obj; p_++; ws();
            if (p_ < e_ && *p_ == '}') { p_++; return true; }
            while (true) {
                std::string k; ws();
                if (!string(k)) return false;
                ws();
                if (p_ >= e_ || *p_ != ':') return fail("expected ':'");
                p_++; ws();
                v.obj.emplace_back(std::move(k), Json());
                if (!value(v.obj.back().second)) return false;
                ws();
                if (p_ < e_ && *p_ == ',') { p_++; continue; }
                if (p_ < e_ && *p_ == '}') { p_++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p_ == '[') {
            v.type = Json::Arr; p_++; ws();
            if (p_ < e_ && *p_ == ']') { p_++; return true; }
            while (true) {
                v.arr.emplace_back();
                ws();
                if (!value(v.arr.back())) return false;
                ws();
                if (p_ < e_ && *p_ == ',') { p_++; continue; }
                if (p_ < e_ && *p_ == ']') { p_++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p_ == '"') { v.type = Json::Str; return string(v.str); }
        if (!strncmp(p_, "true", 4) && e_ - p_ >= 4) { v.type = Json::Bool; v.b = true; p_ += 4; return true; }
        if (!strncmp(p_, "false", 5) && e_ - p_ >= 5) { v.type = Json::Bool; v.b = false; p_ += 5; return true; }
        if (!strncmp(p_, "null", 4) && e_ - p_ >= 4) { v.type = Json::Null; p_ += 4; return true; }
        char *end = nullptr;
        v.num = strtod(p_, &end);
        if (end == p_) return fail("bad value");
        v.type = Json::Num; p_ = end;
        return true;
    }
};

// ---- read-only mmap -------------------------------------------------------------------------------------------
class MappedFile {
  public:
    ~MappedFile() { if (p_) munmap(const_cast<char *>(p_), n_); }
    bool open(const std::string &path, std::string &err) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "cannot open " + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size == 0) { ::close(fd); err = "cannot stat " + path; return false; }
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) { err = "cannot mmap " + path; return false; }
        p_ = (const char *)m; n_ = (size_t)st.st_size;
        return true;
    }
    const char *data() const { return p_; }
    size_t size() const { return n_; }
  private:
    const char *p_ = nullptr; size_t n_ = 0;
};

// ---- safetensors: u64 header length, JSON header {name: {dtype, shape, data_offsets}}, raw little-endian data ---
struct StTensor {
    std::string name; std::string dtype; std::vector<int64_t> shape; const void *data; size_t bytes;
    size_t n_elem() const { size_t n = 1; for (int64_t d : shape) n *= (size_t)d; return n; }
};
// bytes per element of the dtypes a Whisper checkpoint can hold (candle's VarBuilder casts any of them to m::DTYPE = f32)
inline size_t st_elem_size(const std::string &dt) { return dt == "F32" ? 4 : (dt == "F16" || dt == "BF16") ? 2 : 0; }
// widen one tensor to f32 (F32 copy, F16 / BF16 exact)
inline bool st_to_f32(const StTensor &t, std::vector<float> &out) {
    // dtype and size first: the shape of a tensor of any OTHER dtype has not been checked against its byte range as
    // n * elem_size (SafeTensors::open only bounds it), so nothing may be sized from it
    const size_t es = st_elem_size(t.dtype);
    const size_t n = t.n_elem();
    if (!es || n > (size_t)-1 / es || n * es != t.bytes) return false;
    out.resize(n);
    if (t.dtype == "F32") memcpy(out.data(), t.data, n * 4);
    else if (t.dtype == "F16") { const _Float16 *h = reinterpret_cast<const _Float16 *>(t.data); for (size_t i = 0; i < n; i++) out[i] = (float)h[i]; }
    else if (t.dtype == "BF16") {
        const uint16_t *h = reinterpret_cast<const uint16_t *>(t.data);
        for (size_t i = 0; i < n; i++) { uint32_t u = (uint32_t)h[i] << 16; memcpy(&out[i], &u, 4); }
    } else return false;
    return true;
}

// A checkpoint is untrusted input: every size the file declares is checked against the mapping before use (header length,
// data offsets, and bytes == prod(shape) * sizeof(dtype)), with overflow-safe arithmetic.
class SafeTensors {
  public:
    bool open(const std::string &path, std::string &err) {
        if (!f_.open(path, err)) return false;
        if (f_.size() < 8) { err = "safetensors: file too small"; return false; }
        uint64_t hl; memcpy(&hl, f_.data(), 8);
        if (hl > f_.size() - 8) { err = "safetensors: bad header length"; return false; }
        Json h; JsonParser jp(f_.data() + 8, (size_t)hl);
        if (!jp.parse(h) || h.type != Json::Obj) { err = "safetensors: " + jp.error; return false; }
        const char *base = f_.data() + 8 + hl;
        const size_t avail = f_.size() - 8 - (size_t)hl;
        for (auto &kv : h.obj) {
            if (kv.first == "__metadata__") continue;
            const Json *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
            if (!dt || !sh || !off || dt->type != Json::Str || sh->type != Json::Arr || off->arr.size() != 2) { err = "safetensors: malformed entry " + kv.first; return false; }
            StTensor t; t.name = kv.first; t.dtype = dt->str;
            size_t n = 1;
            for (auto &d : sh->arr) {
                if (d.type != Json::Num || d.num < 0 || d.num > 9e15) { err = "safetensors: bad shape for " + kv.first; return false; }
                const size_t dim = (size_t)d.num;
                if (dim != 0 && n > (size_t)-1 / dim) { err = "safetensors: shape overflows for " + kv.first; return false; }
                n *= dim;
                t.shape.push_back((int64_t)dim);
            }
            if (off->arr[0].num < 0 || off->arr[1].num < 0 || off->arr[0].num > 9e15 || off->arr[1].num > 9e15) { err = "safetensors: bad offsets for " + kv.first; return false; }
            size_t b = (size_t)off->arr[0].num, e = (size_t)off->arr[1].num;
            if (e < b || e > avail) { err = "safetensors: offsets out of range for " + kv.first; return false; }
            const size_t es = st_elem_size(t.dtype);
            if (es && (n > (size_t)-1 / es || e - b != n * es)) { err = "safetensors: " + kv.first + " declares " + std::to_string(n) + " " + t.dtype + " elements over " + std::to_string(e - b) + " bytes"; return false; }
            // a dtype this reader does not widen (I64, U8, F8_E4M3 ... -- never read by Whisper): listed, never converted; its
            // shape must still be plausible for its byte range (the smallest safetensors element is one byte)
            if (!es && n > e - b) { err = "safetensors: " + kv.first + " declares " + std::to_string(n) + " " + t.dtype + " elements over " + std::to_string(e - b) + " bytes"; return false; }
            t.data = base + b; t.bytes = e - b;
            tensors.push_back(std::move(t));
        }
        return true;
    }
    std::vector<StTensor> tensors;
  private:
    MappedFile f_;
};

// ---- GGUF (v2 / v3): the quantised checkpoints of `lmz/candle-whisper` (model-tiny-q80.gguf, model-tiny-en-q80.gguf) that
// multilingual.rs:195-199, 227-232 / monolingual.rs load through candle's quantized_var_builder::VarBuilder::from_gguf.
// Layout: "GGUF", u32 version, u64 n_tensors, u64 n_kv; n_kv x {string key, u32 type, value}; n_tensors x {string name,
// u32 n_dims, u64 dims[] (innermost first), u32 ggml type, u64 offset}; padding to general.alignment (default 32); data.
// Tensor types read here: F32 (0), F16 (1), Q8_0 (8: blocks of 32 int8 preceded by one f16 scale, 34 bytes).
// The MI355X path keeps no int8 arithmetic: weights are dequantised at load time (value = scale * q, then rounded to
// fp16 like every other weight) and run through the same fp16 MFMA kernels.  candle's CPU path for these models also
// quantises the ACTIVATIONS of every matmul to Q8_0 blocks (k_quants vec_dot_q8_0_q8_0); that is a property of its CPU
// kernels, not of the checkpoint, and is not reproduced (its own CUDA path does not do it either).
struct GgufTensor { std::string name; uint32_t type = 0; std::vector<int64_t> shape /* outermost first */; const char *data = nullptr; size_t n_elem = 0; };

class GgufFile {
  public:
    bool open(const std::string &path, std::string &err) {
        if (!f_.open(path, err)) return false;
        p_ = f_.data(); end_ = p_ + f_.size();
        uint32_t magic = 0, version = 0; uint64_t n_tensors = 0, n_kv = 0;
        if (!rd(magic) || magic != 0x46554747u) { err = "gguf: bad magic in " + path; return false; }
        if (!rd(version) || version < 2 || version > 3) { err = "gguf: unsupported version " + std::to_string(version); return false; }
        if (!rd(n_tensors) || !rd(n_kv) || n_tensors > (1u << 20) || n_kv > (1u << 20)) { err = "gguf: bad header"; return false; }
        uint64_t alignment = 32;
        for (uint64_t i = 0; i < n_kv; i++) {
            std::string key; uint32_t vt = 0;
            if (!rd_str(key) || !rd(vt)) { err = "gguf: truncated metadata"; return false; }
            uint64_t as_u64 = 0; bool have_u = false;
            if (!skip_value(vt, 0, &as_u64, &have_u)) { err = "gguf: bad metadata value for " + key; return false; }
            if (key == "general.alignment" && have_u && as_u64 > 0) alignment = as_u64;
        }
        std::vector<uint64_t> offs;
        for (uint64_t i = 0; i < n_tensors; i++) {
            GgufTensor t; uint32_t nd = 0; uint64_t off = 0;
            if (!rd_str(t.name) || !rd(nd) || nd > 4) { err = "gguf: truncated tensor info"; return false; }
            std::vector<uint64_t> ne(nd);
            t.n_elem = 1;
            for (uint32_t d = 0; d < nd; d++) {
                if (!rd(ne[d])) { err = "gguf: truncated dims"; return false; }
                if (ne[d] != 0 && t.n_elem > ((size_t)1 << 40) / ne[d]) { err = "gguf: element count of " + t.name + " overflows"; return false; }
                t.n_elem *= (size_t)ne[d];
            }
            for (uint32_t d = 0; d < nd; d++) t.shape.push_back((int64_t)ne[nd - 1 - d]);   // ggml lists the contiguous dim first
            if (!rd(t.type) || !rd(off)) { err = "gguf: truncated tensor info"; return false; }
            if (t.type == 8 && (nd == 0 || ne[0] % 32 != 0)) { err = "gguf: Q8_0 tensor " + t.name + " with a row length that is not a multiple of 32"; return false; }
            offs.push_back(off);
            tensors.push_back(std::move(t));
        }
        const size_t hdr = (size_t)(p_ - f_.data());
        const size_t base = (hdr + alignment - 1) / alignment * alignment;
        for (size_t i = 0; i < tensors.size(); i++) {
            GgufTensor &t = tensors[i];
            size_t bytes;
            if (t.type == 0) bytes = t.n_elem * 4;
            else if (t.type == 1) bytes = t.n_elem * 2;
            else if (t.type == 8) bytes = t.n_elem / 32 * 34;
            else { err = "gguf: unsupported ggml type " + std::to_string(t.type) + " for " + t.name; return false; }
            if (base > f_.size() || offs[i] > f_.size() - base || bytes > f_.size() - base - offs[i]) { err = "gguf: data of " + t.name + " runs past the end of the file"; return false; }
            t.data = f_.data() + base + offs[i];
        }
        return true;
    }
    // f32 copy of a tensor (Q8_0: scale * q per 32-block)
    static void to_f32(const GgufTensor &t, std::vector<float> &out) {
        out.resize(t.n_elem);
        if (t.type == 0) memcpy(out.data(), t.data, t.n_elem * 4);
        else if (t.type == 1) { const _Float16 *h = reinterpret_cast<const _Float16 *>(t.data); for (size_t i = 0; i < t.n_elem; i++) out[i] = (float)h[i]; }
        else {
            const char *b = t.data;
            for (size_t blk = 0; blk < t.n_elem / 32; blk++, b += 34) {
                _Float16 d; memcpy(&d, b, 2);
                const float scale = (float)d;
                const int8_t *q = reinterpret_cast<const int8_t *>(b + 2);
                for (int j = 0; j < 32; j++) out[blk * 32 + j] = scale * (float)q[j];
            }
        }
    }
    std::vector<GgufTensor> tensors;

  private:
    template <typename T> bool rd(T &v) { if ((size_t)(end_ - p_) < sizeof(T)) return false; memcpy(&v, p_, sizeof(T)); p_ += sizeof(T); return true; }
    bool rd_str(std::string &s) { uint64_t n = 0; if (!rd(n) || n > (size_t)(end_ - p_)) return false; s.assign(p_, (size_t)n); p_ += n; return true; }
    bool skip_value(uint32_t vt, int depth, uint64_t *as_u64, bool *have_u) {
        static const int sz[] = {1, 1, 2, 2, 4, 4, 4, 1, -1, -2, 8, 8, 8};  // u8 i8 u16 i16 u32 i32 f32 bool string array u64 i64 f64
        if (vt > 12) return false;
        if (sz[vt] > 0) {
            if ((size_t)(end_ - p_) < (size_t)sz[vt]) return false;
            if (as_u64 && (vt == 4 || vt == 10)) { uint64_t v = 0; memcpy(&v, p_, (size_t)sz[vt]); *as_u64 = v; *have_u = true; }
            p_ += sz[vt];
            return true;
        }
        if (vt == 8) { std::string s; return rd_str(s); }
        uint32_t et = 0; uint64_t n = 0;
        if (depth > 2 || !rd(et) || !rd(n)) return false;
        for (uint64_t i = 0; i < n; i++) if (!skip_value(et, depth + 1, nullptr, nullptr)) return false;
        return true;
    }
    MappedFile f_;
    const char *p_ = nullptr, *end_ = nullptr;
};

// ---- config.json: the fields candle's Config reads -------------------------------------------
struct ConfigJson {
    int num_mel_bins = 0, max_source_positions = 0, d_model = 0, encoder_attention_heads = 0, encoder_layers = 0,
        vocab_size = 0, max_target_positions = 0, decoder_attention_heads = 0, decoder_layers = 0;
    std::vector<int32_t> suppress_tokens;
    bool load(const std::string &path, std::string &err) {
        MappedFile f;
        if (!f.open(path, err)) return false;
        Json j; JsonParser jp(f.data(), f.size());
        if (!jp.parse(j) || j.type != Json::Obj) { err = "config.json: " + jp.error; return false; }
        auto geti = [&](const char *k, int &dst) { const Json *v = j.get(k); if (!v || v->type != Json::Num) { err = std::string("config.json: missing ") + k; return false; } dst = (int)v->num; return true; };
        if (!geti("num_mel_bins", num_mel_bins) || !geti("max_source_positions", max_source_positions) || !geti("d_model", d_model) ||
            !geti("encoder_attention_heads", encoder_attention_heads) || !geti("encoder_layers", encoder_layers) ||
            !geti("vocab_size", vocab_size) || !geti("max_target_positions", max_target_positions) ||
            !geti("decoder_attention_heads", decoder_attention_heads) || !geti("decoder_layers", decoder_layers)) return false;
        if (const Json *s = j.get("suppress_tokens")) for (auto &t : s->arr) suppress_tokens.push_back((int32_t)t.num);
        return true;
    }
};

// ---- tokenizer.json: token -> id (model.vocab + added_tokens) and the byte-level BPE decoder -------------------
class TokenizerJson {
  public:
    bool load(const std::string &path, std::string &err) {
        MappedFile f;
        if (!f.open(path, err)) return false;
        Json j; JsonParser jp(f.data(), f.size());
        if (!jp.parse(j) || j.type != Json::Obj) { err = "tokenizer.json: " + jp.error; return false; }
        if (const Json *m = j.get("model"))
            if (const Json *v = m->get("vocab"))
                for (auto &kv : v->obj) put(kv.first, (int)kv.second.num, false);
        if (const Json *a = j.get("added_tokens"))
            for (auto &t : a->arr) {
                const Json *id = t.get("id"), *c = t.get("content"), *sp = t.get("special");
                if (id && c) put(c->str, (int)id->num, sp ? sp->b : true);
            }
        if (id_to_tok_.empty()) { err = "tokenizer.json: no vocabulary"; return false; }
        // GPT-2 bytes_to_unicode: printable bytes map to themselves, the rest to U+0100 + n
        int n = 0;
        for (int b = 0; b < 256; b++) {
            bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
            uni_to_byte_[keep ? (unsigned)b : 256u + (unsigned)n++] = (unsigned char)b;
        }
        return true;
    }
    // Tokenizer::token_to_id (whisper/mod.rs:86-90); -1 = Error::TokenId
    int token_to_id(const std::string &tok) const { auto it = tok_to_id_.find(tok); return it == tok_to_id_.end() ? -1 : it->second; }
    // Tokenizer::decode(ids, skip_special_tokens) for a ByteLevel BPE model: special added tokens are dropped when asked,
    // the remaining token strings are mapped back to bytes (GPT-2 bytes_to_unicode) and the byte string is turned into
    // text like Rust's String::from_utf8_lossy does (the ByteLevel decoder of the `tokenizers` crate): every maximal
    // invalid sequence becomes U+FFFD.
    std::string decode(const uint32_t *ids, size_t n, bool skip_special = true) const {
        std::string bytes;
        for (size_t i = 0; i < n; i++) {
            if (ids[i] >= id_to_tok_.size() || (skip_special && special_[ids[i]])) continue;
            const std::string &t = id_to_tok_[ids[i]];
            for (size_t p = 0; p < t.size();) {  // UTF-8 code points -> original bytes
                unsigned c = (unsigned char)t[p], cp; int len;
                if (c < 0x80) { cp = c; len = 1; } else if ((c >> 5) == 6) { cp = c & 0x1F; len = 2; }
                else if ((c >> 4) == 14) { cp = c & 0x0F; len = 3; } else { cp = c & 0x07; len = 4; }
                for (int k = 1; k < len && p + k < t.size(); k++) cp = (cp << 6) | ((unsigned char)t[p + k] & 0x3F);
                p += len;
                auto it = uni_to_byte_.find(cp);
                if (it != uni_to_byte_.end()) bytes += (char)it->second;
            }
        }
        return utf8_lossy(bytes);
    }
    // String::from_utf8_lossy: valid sequences are kept; each maximal invalid prefix (per the Unicode "substitution of
    // maximal subparts" rule Rust follows) is replaced by one U+FFFD
    static std::string utf8_lossy(const std::string &b) {
        std::string out;
        const size_t n = b.size();
        size_t i = 0;
        auto cont = [&](size_t k, unsigned lo, unsigned hi) { return k < n && (unsigned char)b[k] >= lo && (unsigned char)b[k] <= hi; };
        while (i < n) {
            const unsigned c = (unsigned char)b[i];
            size_t len = 0;   // length of a valid sequence starting here, or 0
            size_t bad = 1;   // bytes consumed by one replacement when invalid
            if (c < 0x80) len = 1;
            else if (c >= 0xC2 && c <= 0xDF) { if (cont(i + 1, 0x80, 0xBF)) len = 2; }
            else if (c >= 0xE0 && c <= 0xEF) {
                const unsigned lo = c == 0xE0 ? 0xA0 : 0x80, hi = c == 0xED ? 0x9F : 0xBF;
                if (cont(i + 1, lo, hi)) { if (cont(i + 2, 0x80, 0xBF)) len = 3; else bad = 2; }
            } else if (c >= 0xF0 && c <= 0xF4) {
                const unsigned lo = c == 0xF0 ? 0x90 : 0x80, hi = c == 0xF4 ? 0x8F : 0xBF;
                if (cont(i + 1, lo, hi)) {
                    if (cont(i + 2, 0x80, 0xBF)) { if (cont(i + 3, 0x80, 0xBF)) len = 4; else bad = 3; }
                    else bad = 2;
                }
            }
            if (len) { out.append(b, i, len); i += len; }
            else { out += "\xEF\xBF\xBD"; i += bad; }
        }
        return out;
    }
    size_t size() const { return id_to_tok_.size(); }
  private:
    void put(const std::string &tok, int id, bool special) {
        if (id < 0) return;
        if ((size_t)id >= id_to_tok_.size()) { id_to_tok_.resize(id + 1); special_.resize(id + 1, false); }
        id_to_tok_[id] = tok; special_[id] = special; tok_to_id_[tok] = id;
    }
    std::vector<std::string> id_to_tok_;
    std::vector<bool> special_;
    std::unordered_map<std::string, int> tok_to_id_;
    std::unordered_map<unsigned, unsigned char> uni_to_byte_;
};

}  // namespace assets
}  // namespace norma
