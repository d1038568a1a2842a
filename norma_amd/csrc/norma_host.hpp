// norma_host.hpp -- C++ host layer above the C ABI (include/norma_hip.h) that mirrors norma's plugin
// interface for the Whisper path: same names, argument meaning and error behaviour as the Rust items
// it stands in for (the reference is compiled Rust; no Rust toolchain exists in this environment, so
// the host side is written in C++; the Rust binding itself is in INTEGRATION.md).
//
//   reference (MikeIvanichev/norma @ 2024_10_08)                 here
//   ------------------------------------------------------------ ---------------------------------
//   models::SelectedDevice              src/models/mod.rs:36-55   norma::SelectedDevice (+ Rocm(ord))
//   models::CommonModelParams           src/models/mod.rs:58-117  norma::CommonModelParams
//   models::ModelDefinition / Model     src/models/mod.rs:13-34   norma::whisper::Definition / Model
//   whisper::Error                      whisper/mod.rs:64-84      norma::whisper::Error
//   whisper::monolingual::ModelType     monolingual.rs:32-111     norma::whisper::ModelType
//   whisper::Model::transcribe          model.rs:55-159           Model::transcribe
//   Model::decode_with_fallback (t=0)   model.rs:164-191          Model::decode_with_fallback
//   SliceExt::inclusive_boxed_by        src/utils.rs:22-76        norma::inclusive_boxed_by
//
// Text decoding needs a tokenizer (`tokenizers` crate in the reference, model.rs:147); none is
// available offline, so transcribe() returns the token ids of each segment and, when a detokenizer
// callback is installed, the text as well.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <functional>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "../../include/norma_hip.h"
#include "norma_assets.hpp"

namespace norma {

// ---- src/utils.rs:22-76 ----------------------------------------------------------------------------
// Consecutive sub-slices that start AND end (inclusive) on an element matching pred.
template <typename T, typename P>
std::vector<std::pair<size_t, size_t>> inclusive_boxed_by(const std::vector<T> &v, P pred) {
    std::vector<std::pair<size_t, size_t>> out;  // [begin, end)
    size_t base = 0;
    while (base < v.size()) {
        size_t s = base;
        while (s < v.size() && !pred(v[s])) s++;
        if (s >= v.size()) break;
        size_t e = s + 1;
        while (e < v.size() && !pred(v[e])) e++;
        if (e >= v.size()) break;
        out.emplace_back(s, e + 1);
        base = e + 1;
    }
    return out;
}

// ---- src/models/mod.rs:36-55 ------------------------------------------------------------------------
struct SelectedDevice {
    enum Kind { Cpu, Cuda, Metal, Rocm } kind = Cpu;  // Default = Cpu
    size_t ordinal = 0;
    static SelectedDevice cpu() { return {}; }
    static SelectedDevice cuda(size_t n) { return {Cuda, n}; }
    static SelectedDevice metal() { return {Metal, 0}; }
    static SelectedDevice rocm(size_t n) { return {Rocm, n}; }  // NEW: MI355X ordinal, selects the HIP backend
};

// ---- src/models/mod.rs:58-117 -----------------------------------------------------------------------
class CommonModelParams {
    static constexpr size_t MIN_CHUNK_LEN = 100, MIN_STRING_BUF_SIZE = 1;
    size_t max_chunk_len_, data_buffer_size_, string_buffer_size_;
  public:
    CommonModelParams(size_t max_chunk_len, size_t data_buffer_size, size_t string_buffer_size)
        : max_chunk_len_(std::max(max_chunk_len, MIN_CHUNK_LEN)), data_buffer_size_(data_buffer_size + 2),
          string_buffer_size_(std::max(string_buffer_size, MIN_STRING_BUF_SIZE)) {}
    size_t max_chunk_len() const { return std::max(max_chunk_len_, MIN_CHUNK_LEN); }
    size_t data_buffer_size() const { return data_buffer_size_; }
    size_t string_buffer_size() const { return string_buffer_size_; }
    void set_max_chunk_len(size_t v) { max_chunk_len_ = std::max(v, MIN_CHUNK_LEN); }
    void set_data_buffer_size(size_t v) { data_buffer_size_ = v + 2; }
    void set_string_buffer_size(size_t v) { string_buffer_size_ = std::max(v, MIN_STRING_BUF_SIZE); }
};

namespace whisper {

constexpr uint32_t SAMPLE_RATE = 16000;       // Model::SAMPLE_RATE (model.rs:52)
constexpr size_t N_SAMPLES = NH_N_SAMPLES;    // candle m::N_SAMPLES
constexpr double NO_SPEECH_THRESHOLD = 0.6, LOGPROB_THRESHOLD = -1.0;
constexpr int N_TEMPERATURES = 6;
constexpr float TEMPERATURES[N_TEMPERATURES] = {0.0f, 0.2f, 0.4f, 0.6f, 0.8f, 1.0f};  // candle m::TEMPERATURES (model.rs:175)

// whisper/mod.rs:64-84 (variants that can occur on this path) + the run-time error of model.rs:44-46
struct Error {
    enum Kind { None, TokenId, Backend /* TranscriberError */, MelBins, Respnsivness, UnsupportedDevice } kind = None;
    std::string message;
    explicit operator bool() const { return kind != None; }
};

enum class VocabVersion { V1, V2, EnV1, EnV2 };  // whisper/mod.rs:57-62

// monolingual.rs:32-46 (ids/revisions are download metadata, out of scope offline)
// monolingual::ModelType (monolingual.rs:32-46) followed by multilingual::ModelType (multilingual.rs:47-57); one Definition
// class serves both families here (language fixed = monolingual / MultiAsMono, language detected = multilingual)
enum class ModelType { TinyEn, BaseEn, SmallEn, MediumEn, DistilMediumEn, DistilLargeEnV2, DistilLargeEnV3,
                       QuantizedTinyEn, QuantizedTiny,   // the q8_0 GGUF checkpoints
                       Tiny, Base, Small, Medium, Large, LargeV2, LargeV3 };
inline VocabVersion vocab_version(ModelType m) {  // monolingual.rs:99-110, multilingual.rs:73-84
    switch (m) {
        case ModelType::TinyEn: case ModelType::BaseEn: case ModelType::SmallEn: case ModelType::MediumEn:
        case ModelType::QuantizedTinyEn: return VocabVersion::EnV1;
        case ModelType::DistilLargeEnV3: case ModelType::LargeV3: return VocabVersion::V2;
        default: return VocabVersion::V1;
    }
}
// multilingual.rs:87-92 / the monolingual twin: file-name infix of the quantised checkpoints, nullptr for the others
inline const char *quantized_ext(ModelType m) {
    return m == ModelType::QuantizedTiny ? "tiny" : m == ModelType::QuantizedTinyEn ? "tiny-en" : nullptr;
}

// whisper::Language (languages.rs:7-107): the 99 languages in `Language::iter()` order; token = "<|code|>" (:119-222)
static const char *const LANGUAGE_CODES[99] = {
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi", "vi",
    "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk",
    "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw",
    "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo",
    "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su"};

// crate::dtype::DType (src/dtype.rs:10-45): the sample types a capture stream may deliver.  `valid` marks the four that are
// candle DTypes in the reference (u8, u32, f32, f64: "These are valid DTypes"); the others "can (will) be converted".  Here
// every one of them maps to an NH_SAMPLE_* code of the C ABI, whose nh_logmel_samples does dasp_sample's conversion to
// Model::Data (f32) on the GPU.
template <typename T> struct DType;
#define NORMA_DTYPE(T, CODE, VALID) template <> struct DType<T> { static constexpr int sample = CODE; static constexpr bool valid = VALID; }
NORMA_DTYPE(uint8_t, NH_SAMPLE_U8, true);   NORMA_DTYPE(uint32_t, NH_SAMPLE_U32, true);
NORMA_DTYPE(float, NH_SAMPLE_F32, true);    NORMA_DTYPE(double, NH_SAMPLE_F64, true);
NORMA_DTYPE(int8_t, NH_SAMPLE_I8, false);   NORMA_DTYPE(int16_t, NH_SAMPLE_I16, false);
NORMA_DTYPE(int32_t, NH_SAMPLE_I32, false); NORMA_DTYPE(int64_t, NH_SAMPLE_I64, false);
NORMA_DTYPE(uint16_t, NH_SAMPLE_U16, false); NORMA_DTYPE(uint64_t, NH_SAMPLE_U64, false);
#undef NORMA_DTYPE

enum class Task { Transcribe, Translate };  // multilingual.rs:19-25

struct DecodingResult {  // model.rs:494-499
    std::vector<uint32_t> tokens;
    double avg_logprob = 0, no_speech_prob = 0, compression_ratio = 0;
};

struct Segment { std::vector<uint32_t> tokens; std::string text; };  // one `<ts> text <ts|eot>` span, model.rs:100-149

// The loaded model: owns one nh_ctx (batch 1, like the reference) and the carried-over PCM buffer.
class Model {
  public:
    typedef float Data;                               // Model::Data = f32 (model.rs:49)
    Model(nh_ctx *ctx, nh_config cfg, nh_tokens tk) : ctx_(ctx), cfg_(cfg), tk_(tk) {}
    Model(Model &&o) noexcept : ctx_(o.ctx_), cfg_(o.cfg_), tk_(o.tk_), buf_(std::move(o.buf_)), detok_(std::move(o.detok_)) { o.ctx_ = nullptr; }
    Model(const Model &) = delete;
    ~Model() { if (ctx_) nh_destroy(ctx_); }

    void set_detokenizer(std::function<std::string(const uint32_t *, size_t)> f) { detok_ = std::move(f); }
    // LanguageState::Detect (model.rs:393-440): the language is inferred on the first slice of a transcription and
    // cleared on final_chunk; `language_tokens` in Language::iter() order (multilingual.rs:395-398)
    void enable_language_detection(std::vector<int32_t> language_tokens) { detect_ = true; lang_tokens_ = std::move(language_tokens); lang_token_ = -1; }
    int32_t language_token() const { return detect_ ? lang_token_ : tk_.lang; }
    size_t buffered_samples() const { return buf_.size(); }

    // Model::transcribe (model.rs:55-159).  `data` is consumed (swapped/appended into the model's buffer).
    // Returns an Error with kind != None on a backend failure (the reference's TranscriberError, which
    // ends the transcriber thread: src/lib.rs:466-477).
    Error transcribe(std::vector<float> &data, bool final_chunk, std::vector<Segment> &out, std::string *text = nullptr) {
        if (buf_.empty()) std::swap(buf_, data);                      // :60-64
        else { buf_.insert(buf_.end(), data.begin(), data.end()); data.clear(); }
        bool stop = false;
        while (!buf_.empty() && !stop) {                              // 'new_chunk, :68
            const size_t slice_len = std::min(buf_.size(), N_SAMPLES);
            DecodingResult dr;
            bool have = false;
            Error e = decode_with_fallback(buf_.data(), slice_len, dr, have);
            if (e) return e;
            if (!have) { drain(slice_len); continue; }                // :90-93
            if (dr.no_speech_prob > NO_SPEECH_THRESHOLD && dr.avg_logprob < LOGPROB_THRESHOLD) { drain(slice_len); continue; }  // :95-98
            bool drained = false;
            for (auto seg : inclusive_boxed_by(dr.tokens, [&](uint32_t t) { return t > (uint32_t)tk_.no_timestamps || t == (uint32_t)tk_.eot; })) {
                const uint32_t *tok = dr.tokens.data() + seg.first;
                const size_t n = seg.second - seg.first;
                const uint32_t s_timestamp = tok[0] - (uint32_t)tk_.no_timestamps - 1;  // :103
                const uint32_t e_tok = tok[n - 1];
                if (e_tok == (uint32_t)tk_.eot) {
                    if (s_timestamp == 0 || final_chunk) {
                        if (slice_len == N_SAMPLES || final_chunk) { drain(slice_len); drained = true; }  // :109-115
                        else { stop = true; break; }                                                     // :116-123
                    } else {
                        const size_t pre = buf_.size();
                        drain(std::min((size_t)s_timestamp * 320, slice_len));                           // :125-127
                        drained = true;
                        if (pre > slice_len) break;                                                       // :129-136
                        stop = true; break;                                                               // :138-143
                    }
                }
                Segment sg;
                sg.tokens.assign(tok + 1, tok + n - 1);                                                   // tokens[1..len-1], :147
                if (detok_) { sg.text = detok_(sg.tokens.data(), sg.tokens.size()); if (text) *text += sg.text; }
                out.push_back(std::move(sg));
            }
            // H1: a result without any drained segment (e.g. the no-speech early return, whose
            // tokens hold no timestamp, or segments that all close on a timestamp) leaves `buf` untouched in the
            // reference, which then spins forever on the same slice.  Deviation: drain the slice and go on.
            if (!stop && !drained) drain(slice_len);
        }
        if (final_chunk) {                                            // :153-156
            if (detect_) lang_token_ = -1;                            // self.lang.clear()
            int rc = nh_reset(ctx_);
            if (rc) return backend_error();
        }
        return Error{};
    }

    // decode_with_fallback (model.rs:164-191).  The t = 0 pass is deterministic.  The sampled attempts (t = 0.2 .. 1.0)
    // draw from an entropy-seeded RNG in the reference (monolingual.rs:433-439) and cannot be reproduced draw for draw;
    // here they follow the seeded sampling contract of include/norma_hip.h (same distribution).  Default = the
    // reference's behaviour: the loop is ON and the seed comes from std::random_device (StdRng::from_entropy), so a
    // hopeless slice is dropped exactly when the reference would drop it.  set_temperature_fallback(true, seed) fixes the
    // seed (reproducible tests); set_temperature_fallback(false, _) returns the t = 0 result even when the reference would
    // have gone on, and last_needed_fallback() says so.
    void set_temperature_fallback(bool enable, uint64_t seed) { fallback_ = enable; seed_ = seed; slices_ = 0; }
    Error decode_with_fallback(const float *pcm, size_t n, DecodingResult &dr, bool &have) {
        int32_t ns = (int32_t)n;
        if (nh_logmel(ctx_, pcm, &ns, (int64_t)n, 1)) return backend_error();   // audio::pcm_to_mel + narrow, :74-88
        if (nh_encode(ctx_)) return backend_error();                            // encoder_forward(mel, true), :168
        if (detect_) {                                                          // :170-173
            if (lang_token_ < 0) {
                if (nh_detect_language(ctx_, lang_tokens_.data(), (int)lang_tokens_.size(), &lang_token_, nullptr)) return backend_error();
            } else if (nh_set_languages(ctx_, &lang_token_)) return backend_error();
        }
        std::vector<int32_t> toks(cfg_.max_target_positions);
        have = false;
        const uint32_t clip = slices_++;
        for (int a = 0; a < N_TEMPERATURES && !have; a++) {                      // for &t in m::TEMPERATURES, :175
            nh_decode_result r{};
            if (a == 0) { if (nh_decode_greedy(ctx_, toks.data(), &r, 0)) return backend_error(); }   // decode(.., 0.0), :176
            else if (nh_decode_sampled(ctx_, toks.data(), &r, 0, TEMPERATURES[a], seed_, clip, (uint32_t)a)) return backend_error();
            dr.tokens.assign(toks.begin(), toks.begin() + r.n_tokens);
            dr.avg_logprob = r.avg_logprob; dr.no_speech_prob = r.no_speech_prob;
            dr.compression_ratio = 0.0 / 0.0;                                    // f64::NAN, :387
            // :177-187 -- accept unless avg_logprob < -1 (compression_ratio is NaN, so that test is always false)
            const bool needs_fallback = dr.avg_logprob < LOGPROB_THRESHOLD;
            if (a == 0) needs_fallback_ = needs_fallback && !(dr.no_speech_prob > NO_SPEECH_THRESHOLD);
            if (!needs_fallback || dr.no_speech_prob > NO_SPEECH_THRESHOLD || !fallback_) have = true;
        }
        last_ = dr;
        return Error{};                                                          // have == false: Ok(None), :189-190
    }
    const DecodingResult &last_result() const { return last_; }
    bool has_detokenizer() const { return (bool)detok_; }
    bool last_needed_fallback() const { return needs_fallback_; }

  private:
    static uint64_t entropy_seed() { std::random_device rd; return ((uint64_t)rd() << 32) ^ (uint64_t)rd(); }  // StdRng::from_entropy, monolingual.rs:433
    void drain(size_t n) { buf_.erase(buf_.begin(), buf_.begin() + (long)std::min(n, buf_.size())); }
    Error backend_error() const { return Error{Error::Backend, nh_last_error(ctx_)}; }
    nh_ctx *ctx_;
    nh_config cfg_;
    nh_tokens tk_;
    std::vector<float> buf_;
    std::function<std::string(const uint32_t *, size_t)> detok_;
    DecodingResult last_;
    bool needs_fallback_ = false;
    bool fallback_ = true;
    uint64_t seed_ = entropy_seed();
    uint32_t slices_ = 0;
    bool detect_ = false;
    std::vector<int32_t> lang_tokens_;
    int32_t lang_token_ = -1;
};

// One tensor handed to the loader (HF name, f32 or f16 data) -- stands in for the safetensors mmap of
// monolingual.rs:237-239; the caller decides where the bytes come from.
struct TensorView { std::string name; int dtype; std::vector<int64_t> shape; const void *data; };

// monolingual::Definition (monolingual.rs:113-174, 176-452).  The hf-hub download is out of scope (no
// network); the loaded pieces (config, special-token ids, suppress list, mel filters, tensors) are passed in.
class Definition {
  public:
    Definition(ModelType model, SelectedDevice device)  // Definition::new, monolingual.rs:124-130
        : model_(model), device_(device), common_params_(SAMPLE_RATE * 25, 3, 3) {}
    const CommonModelParams &common_params() const { return common_params_; }
    Error set_responsiveness(uint64_t period_ms) {      // monolingual.rs:147-156
        if (period_ms >= 1000 && period_ms <= 30000) { common_params_.set_max_chunk_len((size_t)SAMPLE_RATE * period_ms / 1000); return Error{}; }
        return Error{Error::Respnsivness, "The respnsivness must be over 1 second and under 30"};
    }
    void set_data_buffer_size(size_t n) { common_params_.set_data_buffer_size(n); }
    void set_string_buffer_size(size_t n) { common_params_.set_string_buffer_size(n); }
    ModelType model() const { return model_; }

    // blocking_try_to_model (monolingual.rs:320-451) for SelectedDevice::Rocm.
    Error blocking_try_to_model(const nh_config &cfg, const nh_tokens &tk, const std::vector<int32_t> &suppress,
                                const float *mel_filters, int n_mel, const std::vector<TensorView> &tensors,
                                Model **out) const {
        if (device_.kind != SelectedDevice::Rocm)
            return Error{Error::UnsupportedDevice, "this build only implements SelectedDevice::Rocm(ord)"};
        if (n_mel != 80 && n_mel != 128)   // whisper::Error::MelBins, monolingual.rs:351-355
            return Error{Error::MelBins, "Unexpected number of mel bins (num_mel_bins), got: " + std::to_string(n_mel)};
        nh_ctx *ctx = nullptr;
        if (nh_create((int)device_.ordinal, &cfg, 1, &ctx)) return Error{Error::Backend, nh_last_error(nullptr)};
        auto fail = [&]() { Error e{Error::Backend, nh_last_error(ctx)}; nh_destroy(ctx); return e; };
        for (const auto &t : tensors)
            if (nh_load_tensor(ctx, t.name.c_str(), t.dtype, t.shape.data(), (int)t.shape.size(), t.data)) return fail();
        if (nh_missing_tensors(ctx) != 0) { nh_destroy(ctx); return Error{Error::Backend, "checkpoint is missing tensors"}; }
        if (nh_set_mel_filters(ctx, mel_filters, n_mel)) return fail();
        if (nh_set_tokens(ctx, &tk, suppress.data(), (int)suppress.size())) return fail();
        *out = new Model(ctx, cfg, tk);
        return Error{};
    }

    // The same from a local checkpoint directory holding config.json, tokenizer.json and model.safetensors
    // (what monolingual.rs:323-373 reads after the hf-hub download).  `language` is the `<|xx|>` token of
    // ModelType::language() (monolingual.rs:85-97, :384); translate selects <|translate|> (multilingual.rs:383-386).
    Error blocking_try_to_model_from_dir(const std::string &dir, const float *mel_filters, int n_mel, Model **out,
                                         const std::string &language = "<|en|>", bool translate = false,
                                         bool detect_language = false) const {
        std::string err;
        // quantised models: config-{ext}.json / tokenizer-{ext}.json / model-{ext}-q80.gguf (multilingual.rs:195-199)
        const char *qext = quantized_ext(model_);
        const std::string sfx = qext ? std::string("-") + qext : std::string();
        assets::ConfigJson cj;
        if (!cj.load(dir + "/config" + sfx + ".json", err)) return Error{Error::Backend, err};
        auto tok = std::make_shared<assets::TokenizerJson>();
        if (!tok->load(dir + "/tokenizer" + sfx + ".json", err)) return Error{Error::Backend, err};
        auto id = [&](const char *t, int &dst) { dst = tok->token_to_id(t); return dst >= 0; };
        nh_tokens tk{};
        // candle constants m::SOT_TOKEN, EOT_TOKEN, TRANSCRIBE_TOKEN, TRANSLATE_TOKEN, NO_TIMESTAMPS_TOKEN, NO_SPEECH_TOKENS
        if (!id("<|startoftranscript|>", tk.sot)) return Error{Error::TokenId, "Failed to get token ID for: <|startoftranscript|>"};
        if (!id("<|endoftext|>", tk.eot)) return Error{Error::TokenId, "Failed to get token ID for: <|endoftext|>"};
        if (!id(translate ? "<|translate|>" : "<|transcribe|>", tk.task)) return Error{Error::TokenId, "Failed to get token ID for the task token"};
        if (!id("<|nocaptions|>", tk.no_speech) && !id("<|nospeech|>", tk.no_speech)) return Error{Error::TokenId, "Failed to get token ID for: <|nocaptions|> nor <|nospeech|>"};
        if (!id("<|notimestamps|>", tk.no_timestamps)) return Error{Error::TokenId, "Failed to get token ID for: <|notimestamps|>"};
        std::vector<int32_t> lang_tokens;
        if (detect_language) {  // multilingual::Definition: LanguageState::Detect (multilingual.rs:395-398, :463-466)
            tk.lang = -1;
            for (const char *code : LANGUAGE_CODES) {
                int t = tok->token_to_id(std::string("<|") + code + "|>");
                if (t < 0) return Error{Error::TokenId, std::string("Failed to get token ID for: <|") + code + "|>"};
                lang_tokens.push_back(t);
            }
        } else if (!id(language.c_str(), tk.lang)) return Error{Error::TokenId, "Failed to get token ID for: " + language};
        if (!id("<|0.00|>", tk.zero_sec)) return Error{Error::TokenId, "Failed to get token ID for: <|0.00|>"};
        if (!id("<|1.00|>", tk.one_sec)) return Error{Error::TokenId, "Failed to get token ID for: <|1.00|>"};
        assets::SafeTensors st;
        assets::GgufFile gg;
        std::vector<std::vector<float>> deq;   // dequantised GGUF tensors (kept alive until the upload below)
        std::vector<TensorView> tv;
        std::vector<std::pair<size_t, size_t>> bf16_fix;
        if (qext) {
            if (!gg.open(dir + "/model-" + qext + "-q80.gguf", err)) return Error{Error::Backend, err};
            deq.resize(gg.tensors.size());
            for (size_t i = 0; i < gg.tensors.size(); i++) {
                assets::GgufFile::to_f32(gg.tensors[i], deq[i]);
                tv.push_back(TensorView{gg.tensors[i].name, NH_DTYPE_F32, gg.tensors[i].shape, deq[i].data()});
            }
        } else {
            if (!st.open(dir + "/model.safetensors", err)) return Error{Error::Backend, err};
            for (const auto &t : st.tensors) {
                if (t.dtype == "F16") tv.push_back(TensorView{t.name, NH_DTYPE_F16, t.shape, t.data});
                else if (t.dtype == "F32") tv.push_back(TensorView{t.name, NH_DTYPE_F32, t.shape, t.data});
                else if (t.dtype == "BF16") {   // widened on the host (exact); candle casts every dtype to m::DTYPE the same way
                    deq.emplace_back();
                    assets::st_to_f32(t, deq.back());
                    tv.push_back(TensorView{t.name, NH_DTYPE_F32, t.shape, nullptr});
                    bf16_fix.push_back({tv.size() - 1, deq.size() - 1});
                } else return Error{Error::Backend, "model.safetensors: unsupported dtype " + t.dtype + " for " + t.name};
            }
            for (auto &f : bf16_fix) tv[f.first].data = deq[f.second].data();   // deq may have reallocated while growing
        }
        nh_config cfg{cj.num_mel_bins, cj.max_source_positions, cj.d_model, cj.encoder_attention_heads, cj.encoder_layers,
                      cj.vocab_size, cj.max_target_positions, cj.decoder_attention_heads, cj.decoder_layers};
        Error e = blocking_try_to_model(cfg, tk, cj.suppress_tokens, mel_filters, n_mel, tv, out);
        if (e) return e;
        (*out)->set_detokenizer([tok](const uint32_t *ids, size_t n) { return tok->decode(ids, n); });  // model.rs:147
        if (detect_language) (*out)->enable_language_detection(std::move(lang_tokens));
        return Error{};
    }

  private:
    ModelType model_;
    SelectedDevice device_;
    CommonModelParams common_params_;
};

}  // namespace whisper
}  // namespace norma
