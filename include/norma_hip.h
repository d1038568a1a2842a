/*
 * norma_hip.h -- C ABI of libnorma_hip.so: the MI355X (gfx950) Whisper hot path for norma.
 *
 * This is the drop-in boundary a `norma-hip-sys` FFI crate binds.  Plain
 * pointers and sizes only; no C++ or torch types.  Every function returns an int status
 * (NH_OK == 0) and never throws or aborts; nh_last_error() gives the message for the last
 * failure on that context.  A context is owned by one host thread (the reference's `Model` is
 * `Send`, not `Sync`: src/models/mod.rs:24, src/lib.rs:377,462-464).  One process may hold any number of contexts, on one
 * device or several, each driven by its own thread; they share nothing mutable.  Several contexts on ONE device are how the
 * latency-bound decode of one batch is overlapped with the encoder of the next (bench.py keeps three batches in flight).
 *
 * All citations are relative to the reference repository MikeIvanichev/norma @ 2024_10_08.
 *
 * What each entry point replaces:
 *   nh_create / nh_destroy        SelectedDevice -> device (src/models/mod.rs:47-55) and
 *                                 Whisper::load (src/models/whisper/monolingual.rs:371-373)
 *   nh_create_shared              (no counterpart: the reference is single-stream) further contexts over one weight set
 *   nh_load_tensor                VarBuilder::from_mmaped_safetensors tensor reads by HF name
 *                                 (monolingual.rs:237-239)
 *   nh_set_mel_filters            the include_bytes! filterbank (monolingual.rs:351-362)
 *   nh_set_tokens                 special-token ids + the four vocab masks (monolingual.rs:376-430)
 *   nh_logmel                     audio::pcm_to_mel + Tensor::from_vec + narrow
 *                                 (src/models/whisper/model.rs:74-88)
 *   nh_encode                     Type::encoder_forward (model.rs:168, :455-464)
 *   nh_decode_greedy              Model::decode at t = 0 (model.rs:279-389) including the logit
 *                                 rules (model.rs:212-277), batched, on device
 *   nh_encoder_output, nh_decoder_forward, nh_final_linear, nh_apply_rules
 *                                 fine-grained views of the same state for layer-level parity:
 *                                 Type::decoder_forward / decoder_final_linear (model.rs:466-483)
 *   nh_decode_sampled             Model::decode at t > 0 (model.rs:340-348) under the seeded sampling contract below
 *   nh_detect_language            Model::detect_language (model.rs:194-210)
 *   nh_reset                      Type::reset_kv_cache (model.rs:485-490)
 */
#ifndef NORMA_HIP_H
#define NORMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NH_OK 0
#define NH_ERR_INVALID 1   /* bad argument / unknown tensor name / wrong shape */
#define NH_ERR_HIP 2       /* a HIP runtime call failed (message has the hipError string) */
#define NH_ERR_STATE 3     /* call sequence error (e.g. decode before encode) */
#define NH_ERR_NOMEM 4

#define NH_DTYPE_F32 0
#define NH_DTYPE_F16 1

#define NH_MAX_BATCH 96     /* rows (clips) one context decodes together */
#define NH_N_SAMPLES 480000 /* candle m::N_SAMPLES (model.rs:69)  */
#define NH_N_FRAMES 3000    /* candle m::N_FRAMES  (model.rs:88)  */

typedef struct nh_ctx nh_ctx; /* opaque; owns all device memory and one HIP stream */

/* candle `Config` fields read from HF config.json (monolingual.rs:347). */
typedef struct nh_config {
    int32_t num_mel_bins;
    int32_t max_source_positions; /* 1500 */
    int32_t d_model;
    int32_t encoder_attention_heads;
    int32_t encoder_layers;
    int32_t vocab_size;
    int32_t max_target_positions; /* 448 */
    int32_t decoder_attention_heads;
    int32_t decoder_layers;
} nh_config;

/* norma `Model` token fields (model.rs:37-41) + ids used for first_token_supress. */
typedef struct nh_tokens {
    int32_t sot, eot;
    int32_t lang;  /* language token pushed after sot (model.rs:286-288); < 0: none */
    int32_t task;  /* transcribe / translate */
    int32_t no_speech, no_timestamps;
    int32_t zero_sec, one_sec; /* <|0.00|>, <|1.00|> */
} nh_tokens;

/* Result of one greedy decode (model.rs:494-499 DecodingResult, compression_ratio is always NaN). */
typedef struct nh_decode_result {
    int32_t n_tokens;       /* tokens written for this sequence, prompt and eot included */
    int32_t no_speech_exit; /* 1: early return of model.rs:308-315 */
    double avg_logprob;
    double no_speech_prob;
} nh_decode_result;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* device_ordinal: SelectedDevice::Rocm(ord).  max_batch: chunks processed per call (>= 1). */
int nh_create(int device_ordinal, const nh_config *cfg, int max_batch, nh_ctx **out);
/* Another context on the same device over the SAME weights as `parent` (the reference runs one stream per model,
 * src/lib.rs:462-464; a chunk-parallel caller keeps several batches in flight per GPU and must not pay 1.5 - 3.1 GB of HBM and a
 * separate weight stream per batch).  The new context has its own streams, workspaces, K/V caches, tokens (nh_set_tokens) and
 * decode state; everything nh_load_tensor / nh_set_mel_filters fill in is shared and reference counted: it is freed when the
 * last context of the family is destroyed, in any order.  Loading through ANY context of a family is seen by all of them and
 * must not overlap with a running call on another one (load once, then run); the run-time calls of different contexts may
 * overlap freely, one host thread per context. */
int nh_create_shared(nh_ctx *parent, int max_batch, nh_ctx **out);
void nh_destroy(nh_ctx *ctx);
const char *nh_last_error(const nh_ctx *ctx); /* ctx may be NULL: error of a failed nh_create */
/* 1 when built for and running on a gfx950 device visible to HIP, else 0 (no side effects). */
int nh_device_count(void);

/* ---- model state ----------------------------------------------------------------------------- */
/* name: HF tensor name, e.g. "model.encoder.layers.3.self_attn.q_proj.weight".  data: host
 * pointer, row-major, dtype NH_DTYPE_*.  Tensors candle does not read
 * ("model.encoder.embed_positions.weight", "proj_out.weight") are accepted and ignored. */
int nh_load_tensor(nh_ctx *ctx, const char *name, int dtype, const int64_t *shape, int ndim,
                   const void *data);
/* filters: f32 [num_mel_bins][201] */
int nh_set_mel_filters(nh_ctx *ctx, const float *filters, int n_mel);
int nh_set_tokens(nh_ctx *ctx, const nh_tokens *tk, const int32_t *suppress_tokens, int n_suppress);
/* Number of tensors still missing before the model can run (0 = complete). */
int nh_missing_tensors(const nh_ctx *ctx);

/* ---- the hot path ---------------------------------------------------------------------------- */
/* pcm: host f32 mono 16 kHz, `batch` clips back to back with stride `stride` samples, clip b has
 * n_samples[b] <= 480000 valid samples (the rest is treated as absent, exactly as pcm_to_mel pads
 * with zeros).  Computes the log-mel of every clip on device. */
int nh_logmel(nh_ctx *ctx, const float *pcm, const int32_t *n_samples, int64_t stride, int batch);
/* The sample types norma's `DType` trait admits (src/dtype.rs:37-45): the capture side hands the transcriber samples of the
 * device's native type and converts them with dasp_sample (`Sample::to_sample::<T::Data>`, src/lib.rs:180,207) to the model's
 * Data type -- f32 for Whisper (model.rs:49).  nh_logmel_samples takes the native samples (host memory) and does that
 * conversion on the GPU, so a 16-bit microphone stream crosses PCIe at 2 bytes per sample.  Conversions are dasp_sample's:
 *   i8/i16/i32/i64 -> f32:  (s as f32) / 2^(bits-1)          u8/u16/u32/u64 -> f32: the same after subtracting 2^(bits-1)
 *   f64 -> f32: s as f32 (round to nearest even)             f32: unchanged
 * (integer -> f32 conversions round to nearest even, as Rust's `as` does). */
#define NH_SAMPLE_F32 0
#define NH_SAMPLE_F64 1
#define NH_SAMPLE_I8 2
#define NH_SAMPLE_I16 3
#define NH_SAMPLE_I32 4
#define NH_SAMPLE_I64 5
#define NH_SAMPLE_U8 6
#define NH_SAMPLE_U16 7
#define NH_SAMPLE_U32 8
#define NH_SAMPLE_U64 9
/* Bytes per sample of an NH_SAMPLE_* type (0: unknown). */
int nh_sample_size(int sample_dtype);
/* Like nh_logmel, with `pcm` holding samples of type `sample_dtype` (host memory, `stride` SAMPLES between clips). */
int nh_logmel_samples(nh_ctx *ctx, const void *pcm, int sample_dtype, const int32_t *n_samples, int64_t stride, int batch);
/* Same, but pcm is a DEVICE pointer (already resident in HBM; no copy). */
int nh_logmel_device(nh_ctx *ctx, const float *pcm_dev, const int32_t *n_samples, int64_t stride, int batch);
/* Encoder forward over the mel of the last nh_logmel call (flush = true semantics: the cross
 * K/V of every decoder layer is recomputed). */
int nh_encode(nh_ctx *ctx);
/* Several encoder batches, ONE decode (r03).  The decode step streams the decoder weights and the tied embedding once per
 * token for all rows of the context, so decoding 64 or 96 clips together reads 15 - 20 % fewer bytes per clip than two or
 * three 32-clip decodes -- while the ENCODER side can still be fed 32 clips at a time as they arrive:
 *   nh_logmel_rows(ctx, pcm, n, stride, 32, 0);  nh_encode_rows(ctx, 0, 32);      first 32 clips -> rows 0 .. 31
 *   nh_logmel_rows(ctx, pcm2, n2, stride, 32, 32); nh_encode_rows(ctx, 32, 32);   next 32 clips  -> rows 32 .. 63
 *   nh_decode_greedy(ctx, ...)                                                          all rows filled so far, in lockstep
 * row0 = 0 starts a new set of rows; row0 > 0 must continue where the previous call ended (no gaps) with clips of the same
 * mel length; every row's arithmetic is what it is in a batch of its own (bit-identical results).  The reference decodes one
 * stream at a time (src/lib.rs:462-464): no counterpart. */
int nh_logmel_rows(nh_ctx *ctx, const float *pcm, const int32_t *n_samples, int64_t stride, int batch, int row0);        /* host PCM */
int nh_logmel_device_rows(nh_ctx *ctx, const float *pcm_dev, const int32_t *n_samples, int64_t stride, int batch, int row0); /* PCM in HBM */
int nh_encode_rows(nh_ctx *ctx, int row0, int batch);
/* Greedy decode of all `batch` sequences.  out_tokens: host i32 [batch][max_target_positions],
 * results: [batch].  max_new_tokens <= 0: reference behaviour (cap at max_target_positions - 1). */
int nh_decode_greedy(nh_ctx *ctx, int32_t *out_tokens, nh_decode_result *results, int max_new_tokens);
/* Decode pool (r03): sequences that join and leave a running decode.  The reference's loop ends per sequence at eot
 * (model.rs:317); decoded as one lockstep batch the short sequences wait for the longest.  Here rows [0, rows) of the context
 * decode, every row at its own position, and the rows above them are encoder staging:
 *   nh_pool_begin(ctx, 64, 0, 0);
 *   nh_logmel_rows(ctx, pcm, n, stride, 32, 64);  nh_encode_rows(ctx, 64, 32);      32 clips -> staging rows 64 .. 95
 *   nh_pool_admit(ctx, 64 + i, free_row, -1);                                      clip i joins the decode at position 0
 *   nh_pool_step(ctx, 16, done);                                                   16 tokens for every busy row
 *   nh_pool_collect(ctx, rows_done, k, tokens, results);                           finished rows out, their slots are free again
 * A row's prompt ([sot, lang?, task], model.rs:285-289), no-speech probe and exit (:293-315), rules, length cap (:367) and
 * result (:373-381) are those of nh_decode_greedy, and so are its bits: the same step kernels run, only the position is read
 * per row.  All clips of a pool must produce the same number of mel frames.  nh_logmel* with row0 = 0 ends the pool.
 * per_clip_language != 0: the prompt carries a language token given per clip at nh_pool_admit (LanguageState::Detect);
 * otherwise nh_tokens.lang decides, as in nh_decode_greedy, and `lang` must be -1. */
int nh_pool_begin(nh_ctx *ctx, int rows, int max_new_tokens, int per_clip_language);
/* The clip encoded at staging row src_row (>= rows) starts decoding in the free row dst_row (< rows): its cross K/V move
 * (device-to-device, 4 * S * d_model bytes per decoder layer), its decode state starts over.  Asynchronous. */
int nh_pool_admit(nh_ctx *ctx, int src_row, int dst_row, int32_t lang);
/* The same, the clip coming from row src_row of ANOTHER context of the same weight set (nh_create_shared) that ran
 * nh_logmel* + nh_encode / nh_encode_rows on it: one context decodes without ever stalling for an encoder submission while others
 * encode.  Ordered on the device (the copy waits for that encoder, that context's next encoder submission waits for the copy); on
 * the host the caller keeps this call apart from calls ON `enc` that change its rows (nh_logmel*, nh_encode*). */
int nh_pool_admit_from(nh_ctx *ctx, nh_ctx *enc, int src_row, int dst_row, int32_t lang);
/* n_steps decode steps (one token per busy, unfinished row each), then done_out[rows]: 0 running, 1 finished, 2 finished by
 * the no-speech exit, 3 empty.  Returns when the steps have run. */
int nh_pool_step(nh_ctx *ctx, int n_steps, int32_t *done_out);
/* Results of the n finished rows `rows[]`: out_tokens host i32 [n][max_target_positions], results [n]; the rows are free
 * for nh_pool_admit afterwards. */
int nh_pool_collect(nh_ctx *ctx, const int32_t *rows, int n, int32_t *out_tokens, nh_decode_result *results);
/* Model::decode at t > 0 (model.rs:340-348): every token is SAMPLED from softmax(q / t), q = the rule-masked
 * probabilities.  The reference draws with rand::WeightedIndex from an entropy-seeded StdRng (model.rs:30), so only its
 * distribution can be reproduced; this build fixes a seeded SAMPLING CONTRACT (the C oracle implements the same, bit for bit):
 *   weights  w_i = sexp((q_i - max q) * (1.0f / t)), sexp = exp from IEEE f32 operations only (Cephes polynomial),
 *            masked entries (-inf) weigh 0; this is softmax(q / t) up to the normalisation WeightedIndex ignores;
 *   uniform  u = (philox4x32-10(key = {seed lo, seed hi}, counter = {step, clip, attempt, 0x6e6f726d})[0] >> 8) * 2^-24,
 *            step = tokens in the sequence so far, clip = clip0 + index in the batch, attempt = index into
 *            TEMPERATURES (decode_with_fallback, model.rs:175);
 *   choice   the first token whose cumulative weight exceeds u * total (WeightedIndex::sample's partition_point),
 *            cumulated in f64 over 1024 chunks of ceil(V / 1024) consecutive tokens, then inside the chunk;
 *   all masked: eot is pushed and the sequence stops (model.rs:343-346).  Log-prob bookkeeping as for t = 0. */
int nh_decode_sampled(nh_ctx *ctx, int32_t *out_tokens, nh_decode_result *results, int max_new_tokens,
                      float temperature, uint64_t seed, uint32_t clip0, uint32_t attempt);
/* Model::detect_language (model.rs:194-210) for every clip of the batch: one decoder step on [sot], softmax over the
 * n language-token logits (lang_tokens in `Language::iter()` order, multilingual.rs:395-398), first maximum.
 * out_lang: host i32 [batch]; out_probs: host f32 [batch][n] or NULL.  The detected tokens become the per-sequence
 * language tokens of the next nh_decode_greedy (LanguageState::set_language_token). */
int nh_detect_language(nh_ctx *ctx, const int32_t *lang_tokens, int n, int32_t *out_lang, float *out_probs);
/* Per-sequence language tokens for the next decode (host i32 [batch]); NULL: back to nh_tokens.lang for all. */
int nh_set_languages(nh_ctx *ctx, const int32_t *langs);
/* Device-resident log-mel -> encoder -> decode without intermediate host syncs (bench path). */
int nh_transcribe_batch(nh_ctx *ctx, const float *pcm_dev, const int32_t *n_samples, int64_t stride,
                        int batch, int32_t *out_tokens, nh_decode_result *results, int max_new_tokens);
int nh_reset(nh_ctx *ctx);
int nh_synchronize(nh_ctx *ctx);

/* ---- fine-grained views for layer-level parity tests ------------------------------------------- */
/* mel of clip b in candle layout: f32 [num_mel_bins][3000] */
int nh_get_mel(nh_ctx *ctx, int b, float *out);
/* Upload a mel directly (f32 [batch][num_mel_bins][3000]) instead of nh_logmel. */
int nh_set_mel(nh_ctx *ctx, const float *mel, int batch);
/* encoder output of clip b: f32 [1500][d_model] */
int nh_encoder_output(nh_ctx *ctx, int b, float *out);
/* TextDecoder::forward for every clip over a teacher-forced prefix: tokens i32 [batch][T] ->
 * hidden f32 [batch][T][d_model] (after the final LayerNorm). */
int nh_decoder_forward(nh_ctx *ctx, const int32_t *tokens, int T, float *hidden_out);
/* TextDecoder::final_linear on host-provided rows: x f32 [rows][d_model] -> logits f32 [rows][V] */
int nh_final_linear(nh_ctx *ctx, const float *x, int rows, float *logits_out);
/* The logit rules on device: probs f32 [V] (already soft-maxed), tokens so far, last timestamp
 * (< 0: first generated token).  Returns the masked probabilities (model.rs:333-338). */
int nh_apply_rules(nh_ctx *ctx, const float *probs, const int32_t *tokens, int n_tokens,
                   int last_timestamp, float *masked_out, int32_t *argmax_out);

/* The sampler alone: rules + one draw on a soft-maxed probability vector (token_out = -1: everything masked). */
int nh_sample_rules(nh_ctx *ctx, const float *probs, const int32_t *tokens, int n_tokens, int last_timestamp,
                    float temperature, uint64_t seed, uint32_t clip, uint32_t attempt, int32_t *token_out);

/* ---- instrumentation --------------------------------------------------------------------------- */
/* Milliseconds (HIP events on the context's stream) spent in the phases of the last
 * nh_transcribe_batch / nh_logmel+nh_encode+nh_decode_greedy sequence. */
typedef struct nh_timings {
    float mel_ms, encoder_ms, cross_kv_ms, decode_ms;
    int32_t decode_steps;
    float gemm_ms;      /* summed duration of the dominant encoder GEMM kernel launches */
    int32_t gemm_launches;
    double gemm_flops;  /* algorithmic FLOPs of those launches */
} nh_timings;
int nh_get_timings(nh_ctx *ctx, nh_timings *out);
/* 1: bracket every encoder GEMM launch with a pair of HIP event records on its stream (no synchronisation; bench roofline). */
int nh_set_profile_gemm(nh_ctx *ctx, int enable);

/* A/B switches for the bit-exactness screens in tests/ (the defaults are the product configuration; options 0 and 1 change
 * only how the decode step is launched, never its results). */
#define NH_OPT_DECODE_GRAPHS 0          /* 1 (default): replay the captured decode step; 0: launch every kernel eagerly */
#define NH_OPT_FUSE_DECODE_LAYERNORM 1  /* 1 (default): LayerNorm inside the consuming GEMV; 0: stand-alone LayerNorm kernel */
/* Parity view, the one option that DOES change results: n > 0 runs only the first n decoder blocks of TextDecoder::forward
 * (model.rs:466-476) before the final LayerNorm, so a test can compare the hidden state against an oracle built with n decoder
 * layers and see how the fp16 error grows with depth; 0 (default) = all of them. */
#define NH_OPT_DECODER_LAYER_LIMIT 2
/* Cross-attention computed on the encoder output itself (u = Wk^T q, z = p^T xa, o = Wv z + bv), lockstep
 * decodes only.  1: slow prototype kernels that place the fp16 roundings where the one-pass kernel does (the numerics experiment);
 * 2: the one-pass kernels (xa streamed once per decoder layer; d_model 512 / 768 / 1024 / 1280, other widths fall back to 1).
 * 0 (default): K and V as the reference computes them. */
#define NH_OPT_ABSORBED_XATTN 3
int nh_set_option(nh_ctx *ctx, int option, int value);

#ifdef __cplusplus
}
#endif
#endif /* NORMA_HIP_H */
