/*
 * norma_host.h -- C shim over the C++ host layer (norma_amd/csrc/norma_host.hpp), which mirrors norma's
 * Whisper plugin interface (ModelDefinition / Model::transcribe, src/models/mod.rs:13-34,
 * src/models/whisper/model.rs:55-191) on top of the kernel-level C ABI of norma_hip.h.
 * It exists so that the host layer can be driven from tests (ctypes) exactly like the Rust crate would
 * drive it; a Rust integration binds norma_hip.h directly and keeps this policy in Rust.
 */
#ifndef NORMA_HOST_H
#define NORMA_HOST_H
#include <stddef.h>
#include <stdint.h>

#include "norma_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SelectedDevice kinds (src/models/mod.rs:36-43 + the new Rocm variant) */
#define NM_DEVICE_CPU 0
#define NM_DEVICE_CUDA 1
#define NM_DEVICE_METAL 2
#define NM_DEVICE_ROCM 3

/* monolingual::ModelType (monolingual.rs:32-46): the fp16 checkpoints first, then the quantised ones, then multilingual::ModelType */
#define NM_MODEL_TINY_EN 0
#define NM_MODEL_BASE_EN 1
#define NM_MODEL_SMALL_EN 2
#define NM_MODEL_MEDIUM_EN 3
#define NM_MODEL_DISTIL_MEDIUM_EN 4
#define NM_MODEL_DISTIL_LARGE_EN_V2 5
#define NM_MODEL_DISTIL_LARGE_EN_V3 6
/* the q8_0 GGUF checkpoints of lmz/candle-whisper (monolingual QuantizedTinyEn, multilingual QuantizedTiny): loaded from
 * config-{ext}.json / tokenizer-{ext}.json / model-{ext}-q80.gguf, ext = "tiny-en" / "tiny"; weights are dequantised at load */
#define NM_MODEL_QUANTIZED_TINY_EN 7
#define NM_MODEL_QUANTIZED_TINY 8
/* multilingual::ModelType (multilingual.rs:47-57): load with language = NULL / "" (detected) or a fixed "<|xx|>" (MultiAsMono) */
#define NM_MODEL_TINY 9
#define NM_MODEL_BASE 10
#define NM_MODEL_SMALL 11
#define NM_MODEL_MEDIUM 12
#define NM_MODEL_LARGE 13
#define NM_MODEL_LARGE_V2 14
#define NM_MODEL_LARGE_V3 15

typedef struct nm_definition nm_definition; /* whisper::monolingual::Definition */
typedef struct nm_model nm_model;           /* whisper::Model */
typedef struct nm_tensors nm_tensors;       /* the tensors a checkpoint provides (caller-owned memory) */

nm_definition *nm_definition_new(int model_type, int device_kind, size_t ordinal); /* Definition::new */
void nm_definition_free(nm_definition *d);
int nm_definition_set_responsiveness(nm_definition *d, uint64_t period_ms); /* 0 ok, 1 = Error::Respnsivness */
size_t nm_definition_max_chunk_len(const nm_definition *d);                 /* common_params().max_chunk_len() */
size_t nm_definition_data_buffer_size(const nm_definition *d);
void nm_definition_set_data_buffer_size(nm_definition *d, size_t n);

nm_tensors *nm_tensors_new(void);
void nm_tensors_add(nm_tensors *t, const char *name, int dtype, const int64_t *shape, int ndim, const void *data);
void nm_tensors_free(nm_tensors *t);

/* ModelDefinition::blocking_try_to_model; NULL on failure with the message in err */
nm_model *nm_definition_blocking_try_to_model(const nm_definition *d, const nh_config *cfg, const nh_tokens *tk,
                                              const int32_t *suppress, int n_suppress, const float *mel_filters,
                                              int n_mel, const nm_tensors *tensors, char *err, int err_len);
/* The same from a local directory with config.json, tokenizer.json, model.safetensors (monolingual.rs:323-373 after
 * its hf-hub download); installs the byte-level BPE detokeniser so transcribe also produces text (model.rs:147). */
nm_model *nm_definition_blocking_try_to_model_from_dir(const nm_definition *d, const char *dir, const float *mel_filters,
                                                       int n_mel, const char *language, int translate, char *err,
                                                       int err_len);
/* multilingual models: LanguageState::Detect (model.rs:393-440).  lang_tokens in Language::iter() order
 * (languages.rs:7-107); the language is inferred on the first slice and cleared on final_chunk.
 * (nm_definition_blocking_try_to_model_from_dir enables it itself when `language` is NULL or "".) */
void nm_model_enable_language_detection(nm_model *m, const int32_t *lang_tokens, int n);
/* GGUF reader check (no GPU needed): writes one line per tensor "name type d0xd1.. sum_of_dequantised_values\n" into buf,
 * returns the number of tensors or -1 (message in buf). */
int nm_gguf_list(const char *path, char *buf, int cap);
/* i-th language code in Language::iter() order (languages.rs:7-107, used at model.rs:204 and multilingual.rs:395-398);
 * NULL past the 99th.  The language token is "<|code|>". */
const char *nm_language_code(int i);
/* The asset readers alone (no GPU): tokenizer.json (Tokenizer::from_file, token_to_id, decode -- monolingual.rs:349,
 * mod.rs:86-90, model.rs:147) and safetensors (VarBuilder::from_mmaped_safetensors, monolingual.rs:237-239). */
typedef struct nm_tokenizer nm_tokenizer;
nm_tokenizer *nm_tokenizer_open(const char *path, char *err, int err_len);
void nm_tokenizer_free(nm_tokenizer *t);
int nm_tokenizer_token_to_id(const nm_tokenizer *t, const char *token); /* -1 = whisper::Error::TokenId */
/* writes the UTF-8 text (NUL-terminated, truncated to cap - 1) and returns its full length in bytes */
int nm_tokenizer_decode(const nm_tokenizer *t, const uint32_t *ids, size_t n, int skip_special_tokens, char *buf, int cap);
/* one line per tensor "name dtype d0xd1.. sum sum_abs\n" (f64 sums of the values widened to f32); returns the number of
 * tensors or -1 with the message in buf.  F32, F16 and BF16 are read. */
int nm_safetensors_list(const char *path, char *buf, int cap);
int nm_model_language_token(const nm_model *m); /* current language token, -1 = not detected yet */
/* decode_with_fallback's sampled attempts at t = 0.2 .. 1.0 (model.rs:175-188).  ON by default with an entropy seed, as in the
 * reference (StdRng::from_entropy): a slice whose every attempt has avg_logprob < -1 is dropped.  enable = 1 with a seed
 * fixes the draws (seeded sampling contract of norma_hip.h; the reference's draws cannot be reproduced, only their
 * distribution); enable = 0 returns the t = 0 result and nm_model_last_result reports needed_fallback. */
void nm_model_set_temperature_fallback(nm_model *m, int enable, uint64_t seed);
/* text of the last nm_model_transcribe call (empty without a tokenizer); returns its length */
int nm_model_last_text(const nm_model *m, char *buf, int cap);
void nm_model_free(nm_model *m);

/* Model::transcribe(&mut data, final_chunk).  out_tokens receives the token ids of every emitted segment
 * (what the reference hands to tokenizer.decode, model.rs:147), segments separated by -1; *n_out = ints
 * written; *buffered = samples the model keeps for the next call.  Returns 0, or 1 on a backend error. */
int nm_model_transcribe(nm_model *m, const float *data, size_t n, int final_chunk, int32_t *out_tokens, int cap,
                        int *n_out, size_t *buffered, char *err, int err_len);
/* DecodingResult of the last decoded slice + whether the reference would have entered its sampled fallback */
void nm_model_last_result(const nm_model *m, double *avg_logprob, double *no_speech_prob, int *needed_fallback,
                          int *n_tokens);

#ifdef __cplusplus
}
#endif
#endif
