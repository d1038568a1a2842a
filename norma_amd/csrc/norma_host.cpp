// norma_host.cpp -- C shim (include/norma_host.h) over norma_host.hpp.
#include "../../include/norma_host.h"

#include <string.h>

#include "norma_host.hpp"

using namespace norma;
using namespace norma::whisper;

struct nm_definition { Definition def; };
struct nm_model { Model *m; std::string last_text; };
struct nm_tensors { std::vector<TensorView> v; };

static void put_err(char *err, int n, const std::string &s) {
    if (!err || n <= 0) return;
    strncpy(err, s.c_str(), (size_t)n - 1);
    err[n - 1] = 0;
}

// No exception may cross the C boundary (norma_hip.h: "never throws or aborts"): the readers below size buffers from what an
// untrusted file declares, so std::bad_alloc / std::length_error are reachable.  Every entry point that parses a file, or
// allocates in proportion to its arguments, runs inside this guard and reports the failure through its error channel.
#define NM_TRY try {
#define NM_CATCH(on_fail)                                                             \
    } catch (const std::exception &ex_) { const std::string what_ = ex_.what(); on_fail; } \
    catch (...) { const std::string what_ = "unknown exception"; on_fail; }

extern "C" {

nm_definition *nm_definition_new(int model_type, int device_kind, size_t ordinal) {
    SelectedDevice dev;
    dev.kind = (SelectedDevice::Kind)device_kind;
    dev.ordinal = ordinal;
    return new nm_definition{Definition((ModelType)model_type, dev)};
}
void nm_definition_free(nm_definition *d) { delete d; }
int nm_definition_set_responsiveness(nm_definition *d, uint64_t period_ms) { return d->def.set_responsiveness(period_ms) ? 1 : 0; }
size_t nm_definition_max_chunk_len(const nm_definition *d) { return d->def.common_params().max_chunk_len(); }
size_t nm_definition_data_buffer_size(const nm_definition *d) { return d->def.common_params().data_buffer_size(); }
void nm_definition_set_data_buffer_size(nm_definition *d, size_t n) { d->def.set_data_buffer_size(n); }

nm_tensors *nm_tensors_new(void) { return new nm_tensors(); }
void nm_tensors_add(nm_tensors *t, const char *name, int dtype, const int64_t *shape, int ndim, const void *data) {
    t->v.push_back(TensorView{name, dtype, std::vector<int64_t>(shape, shape + ndim), data});
}
void nm_tensors_free(nm_tensors *t) { delete t; }

nm_model *nm_definition_blocking_try_to_model(const nm_definition *d, const nh_config *cfg, const nh_tokens *tk,
                                              const int32_t *suppress, int n_suppress, const float *mel_filters,
                                              int n_mel, const nm_tensors *tensors, char *err, int err_len) {
    NM_TRY
    Model *m = nullptr;
    std::vector<int32_t> sup(suppress, suppress + (n_suppress > 0 ? n_suppress : 0));
    Error e = d->def.blocking_try_to_model(*cfg, *tk, sup, mel_filters, n_mel, tensors->v, &m);
    if (e) { put_err(err, err_len, e.message); return nullptr; }
    return new nm_model{m, std::string()};
    NM_CATCH({ put_err(err, err_len, "blocking_try_to_model: " + what_); return nullptr; })
}

nm_model *nm_definition_blocking_try_to_model_from_dir(const nm_definition *d, const char *dir, const float *mel_filters,
                                                       int n_mel, const char *language, int translate, char *err,
                                                       int err_len) {
    NM_TRY
    Model *m = nullptr;
    const bool detect = language == nullptr || language[0] == 0;  // multilingual::Definition: infer the language
    Error e = d->def.blocking_try_to_model_from_dir(dir, mel_filters, n_mel, &m, detect ? "" : language, translate != 0, detect);
    if (e) { put_err(err, err_len, e.message); return nullptr; }
    return new nm_model{m, std::string()};
    NM_CATCH({ put_err(err, err_len, "blocking_try_to_model_from_dir: " + what_); return nullptr; })
}

void nm_model_enable_language_detection(nm_model *m, const int32_t *lang_tokens, int n) {
    m->m->enable_language_detection(std::vector<int32_t>(lang_tokens, lang_tokens + n));
}
int nm_model_language_token(const nm_model *m) { return m->m->language_token(); }

int nm_model_last_text(const nm_model *m, char *buf, int cap) {
    if (!buf || cap <= 0) return (int)m->last_text.size();
    strncpy(buf, m->last_text.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
    return (int)m->last_text.size();
}
void nm_model_free(nm_model *m) { if (m) { delete m->m; delete m; } }

int nm_model_transcribe(nm_model *m, const float *data, size_t n, int final_chunk, int32_t *out_tokens, int cap,
                        int *n_out, size_t *buffered, char *err, int err_len) {
    NM_TRY
    std::vector<float> v(data, data + n);
    std::vector<Segment> segs;
    m->last_text.clear();
    Error e = m->m->transcribe(v, final_chunk != 0, segs, &m->last_text);
    if (buffered) *buffered = m->m->buffered_samples();
    if (e) { put_err(err, err_len, e.message); return 1; }
    int w = 0;
    for (const auto &s : segs) {
        if (w + (int)s.tokens.size() + 1 > cap) { put_err(err, err_len, "output buffer too small"); return 1; }
        for (uint32_t t : s.tokens) out_tokens[w++] = (int32_t)t;
        out_tokens[w++] = -1;
    }
    if (n_out) *n_out = w;
    return 0;
    NM_CATCH({ put_err(err, err_len, "transcribe: " + what_); return 1; })
}

int nm_gguf_list(const char *path, char *buf, int cap) {
    NM_TRY
    norma::assets::GgufFile g; std::string err;
    if (!path || !g.open(path, err)) { put_err(buf, cap, err.empty() ? "nm_gguf_list: no path" : err); return -1; }
    std::string out;
    for (const auto &t : g.tensors) {
        std::vector<float> v; norma::assets::GgufFile::to_f32(t, v);
        double s = 0; for (float x : v) s += x;
        out += t.name + " " + std::to_string(t.type) + " ";
        for (size_t i = 0; i < t.shape.size(); i++) out += (i ? "x" : "") + std::to_string(t.shape[i]);
        char num[64]; snprintf(num, sizeof num, " %.9g\n", s);
        out += num;
    }
    put_err(buf, cap, out);
    return (int)g.tensors.size();
    NM_CATCH({ put_err(buf, cap, "nm_gguf_list: " + what_); return -1; })
}

// Language::iter() order (languages.rs:7-107): code i of the table the host layer resolves "<|code|>" tokens from
const char *nm_language_code(int i) { return i >= 0 && i < 99 ? LANGUAGE_CODES[i] : nullptr; }

// ---- asset readers alone (no GPU): what tests/test_assets_cpu.py pins against fixtures written by the `tokenizers` and
// `safetensors` Python bindings of the crates the reference uses (model.rs:147, mod.rs:86-90, monolingual.rs:237-239)
struct nm_tokenizer { norma::assets::TokenizerJson t; };
nm_tokenizer *nm_tokenizer_open(const char *path, char *err, int err_len) {
    nm_tokenizer *t = nullptr;
    NM_TRY
    t = new nm_tokenizer();
    std::string e;
    if (!path || !t->t.load(path, e)) { put_err(err, err_len, e.empty() ? "nm_tokenizer_open: no path" : e); delete t; return nullptr; }
    return t;
    NM_CATCH({ put_err(err, err_len, "nm_tokenizer_open: " + what_); delete t; return nullptr; })
}
void nm_tokenizer_free(nm_tokenizer *t) { delete t; }
int nm_tokenizer_token_to_id(const nm_tokenizer *t, const char *token) { return t && token ? t->t.token_to_id(token) : -1; }
int nm_tokenizer_decode(const nm_tokenizer *t, const uint32_t *ids, size_t n, int skip_special_tokens, char *buf, int cap) {
    if (!t) return -1;
    NM_TRY
    const std::string s = t->t.decode(ids, n, skip_special_tokens != 0);
    if (buf && cap > 0) { const size_t m = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1; memcpy(buf, s.data(), m); buf[m] = 0; }
    return (int)s.size();
    NM_CATCH({ (void)what_; return -1; })
}

int nm_safetensors_list(const char *path, char *buf, int cap) {
    NM_TRY
    norma::assets::SafeTensors st; std::string err;
    if (!path || !st.open(path, err)) { put_err(buf, cap, err.empty() ? "nm_safetensors_list: no path" : err); return -1; }
    std::string out;
    for (const auto &t : st.tensors) {
        std::vector<float> v;
        double s = 0, sa = 0;
        if (norma::assets::st_to_f32(t, v)) for (float x : v) { s += x; sa += x < 0 ? -(double)x : (double)x; }
        out += t.name + " " + t.dtype + " ";
        for (size_t i = 0; i < t.shape.size(); i++) out += (i ? "x" : "") + std::to_string(t.shape[i]);
        if (t.shape.empty()) out += "scalar";
        char num[96]; snprintf(num, sizeof num, " %.17g %.17g\n", s, sa);
        out += num;
    }
    put_err(buf, cap, out);
    return (int)st.tensors.size();
    NM_CATCH({ put_err(buf, cap, "nm_safetensors_list: " + what_); return -1; })
}

void nm_model_set_temperature_fallback(nm_model *m, int enable, uint64_t seed) {
    if (m && m->m) m->m->set_temperature_fallback(enable != 0, seed);
}

void nm_model_last_result(const nm_model *m, double *avg_logprob, double *no_speech_prob, int *needed_fallback,
                          int *n_tokens) {
    const DecodingResult &r = m->m->last_result();
    if (avg_logprob) *avg_logprob = r.avg_logprob;
    if (no_speech_prob) *no_speech_prob = r.no_speech_prob;
    if (needed_fallback) *needed_fallback = m->m->last_needed_fallback() ? 1 : 0;
    if (n_tokens) *n_tokens = (int)r.tokens.size();
}

}  // extern "C"
