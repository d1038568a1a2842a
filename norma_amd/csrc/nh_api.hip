// nh_api.hip -- implementation of the C ABI declared in include/norma_hip.h.
//
// Owns the device state of one Whisper model on one MI355X (weights in fp16, f32 LayerNorm/bias
// parameters, activation workspaces sized for `max_batch` 30-second clips) and sequences the gfx950
// kernels on one HIP stream.  No torch, no candle, no CPU fallback: every entry point either runs
// the HIP path or returns an error.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <algorithm>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/norma_hip.h"
#include "nh_kernels.h"

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            return ctx->fail(NH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                     \
    } while (0)

// message of the last failed nh_create on THIS thread (contexts are created from one thread per GPU)
static thread_local std::string g_create_error;

struct LinW { half_t *w = nullptr; float *b = nullptr; half_t *wt = nullptr; /* tile-major repack (decoder GEMVs) */ };
struct LnW { float *w = nullptr, *b = nullptr; };
struct EncLayer { LnW ln1, ln2; LinW qkv, o, fc1, fc2; };
struct DecLayer { LnW ln1, ln2, ln3; LinW qkv, o, cq, ckv, co, fc1, fc2; half_t *ck = nullptr, *cv = nullptr, *sk = nullptr, *sv = nullptr; };

// The read-only half of a context: everything nh_load_tensor / nh_set_mel_filters fill in.  Owned through a shared_ptr, so
// that several contexts on one device (nh_create_shared) run on ONE copy of the weights: bench.py keeps three batches in
// flight per GPU, and three private copies meant 3 x 1.5 GB of HBM and three different address ranges for the 225 MB of
// decoder weights + embedding every in-flight decode streams per token.  The contexts hold plain pointer VIEWS of these
// tables (the same LinW / LnW structs as before); `version` tells a context that the tables moved under it.
struct nh_model {
    int dev = 0;
    nh_config c{};
    std::vector<void *> allocs;
    LinW conv1, conv2;
    float *enc_pos = nullptr;
    std::vector<EncLayer> enc;
    LnW ln_post, dec_ln;
    half_t *tok_emb = nullptr, *dec_pos = nullptr;
    half_t *tok_emb_t = nullptr;     // tile-major repack of the tied embedding (logits GEMV)
    std::vector<DecLayer> dec;       // weights only (ck / cv / sk / sv stay per context)
    bool dec_tiled_valid = false;    // the repacks mirror the row-major decoder weights loaded so far
    std::set<std::string> expected, loaded;
    MelTables mt{};
    int32_t *mel_grp = nullptr;
    bool have_filters = false;
    int version = 1;                 // bumped whenever a pointer in here changes (lazy repack allocation, new filters)
    std::mutex mu;                   // guards loaded / dec_tiled_valid / version and the lazy repack
    ~nh_model() {
        hipSetDevice(dev);
        for (void *p : allocs) hipFree(p);
    }
};

struct nh_ctx {
    int dev = 0;
    std::shared_ptr<nh_model> mdl;
    int view_version = 0;       // nh_model::version the pointer views below were copied at
    hipStream_t st = nullptr;   // the context's stream: log-mel, encoder, cross K/V, decode loop
    hipStream_t sd = nullptr;   // alias of st (the decode loop's launches are written against sd; see build_context)
    hipEvent_t enc_done = nullptr;
    // nh_pool_admit_from: decode pools of OTHER contexts copy cross K/V out of this context's rows on THEIR streams; the next
    // encoder submission here must not overwrite those rows before the copies have run
    struct EventBox { hipEvent_t e = nullptr; ~EventBox() { if (e) hipEventDestroy(e); } };
    std::shared_ptr<EventBox> kv_copied;       // recorded on this context's stream after it copied K/V out of another context
    std::mutex readers_mu;
    std::vector<std::shared_ptr<EventBox>> kv_readers;   // kv_copied of the pools that read this context's K/V since its last encode (kept alive here)
    nh_config c{};
    int B = 1;
    std::string err;
    std::vector<void *> allocs;
    // weights: VIEWS of mdl's tables (refresh_views); dec[i] also carries this context's own K/V caches
    LinW conv1, conv2;
    float *enc_pos = nullptr;
    std::vector<EncLayer> enc;
    LnW ln_post, dec_ln;
    half_t *tok_emb = nullptr, *dec_pos = nullptr;
    half_t *tok_emb_t = nullptr;
    std::vector<DecLayer> dec;
    // mel
    float *pcm = nullptr;
    void *raw = nullptr; size_t raw_bytes = 0;   // native-sample staging of nh_logmel_samples
    int32_t *nsamp = nullptr;
    float *mel32 = nullptr;
    unsigned *chunk_max = nullptr;
    half_t *mel_img = nullptr;
    // encoder workspaces
    half_t *h1 = nullptr, *xn = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *att = nullptr, *hid = nullptr,
           *xa16 = nullptr;
    float *x = nullptr, *xa32 = nullptr;
    // decoder workspaces
    float *dx = nullptr, *dy32 = nullptr, *logits = nullptr;
    half_t *dxn = nullptr, *dq = nullptr, *datt = nullptr, *dhid = nullptr;
    DecodeState ds{};
    uint8_t *suppress = nullptr;
    float *lpart = nullptr;
    unsigned *ltick = nullptr;
    int32_t *d_pos = nullptr;  // [max_batch] device-side decode position of every sequence (hipGraph replays read and advance it)
    // decode pool (nh_pool_*): rows [0, pool_rows) decode, each at its own position; rows above are encoder staging
    int pool_rows = 0, pool_max_new = 0, pool_prompt = 0;
    bool pool_per_clip_language = false;
    std::vector<char> pool_busy;
    hipGraphExec_t step_graph = nullptr;   // one decode step
    hipGraphExec_t multi_graph = nullptr;  // NH_GRAPH_STEPS consecutive steps (one launch gap instead of NH_GRAPH_STEPS)
    int graph_key[6] = {-1, -1, -1, -1, -1, -1};
    int token_gen = 0;  // bumped by nh_set_tokens; part of the graph key
    bool opt_graphs = true, opt_fuse_ln = true;  // nh_set_option
    int opt_absorbed = 0;        // NH_OPT_ABSORBED_XATTN: 1 = numerics prototype, 2 = one-pass kernels
    half_t *xabs_u = nullptr;    // [max_batch][32][d] scratch (heads padded to 32, pad rows zero)
    float *xabs_z = nullptr, *xabs_ml = nullptr;   // key-range partials of the one-pass form
    int dec_layer_limit = 0;    // parity view (NH_OPT_DECODER_LAYER_LIMIT): run only the first n decoder blocks; 0 = all
    std::vector<int32_t> seq_lang;  // per-sequence language tokens (LanguageState::Detect), empty = tk.lang for all
    int32_t *d_lang_tokens = nullptr, *d_lang_out = nullptr;
    float *d_lang_probs = nullptr;
    RuleTokens tk{};
    bool have_tokens = false;
    int VP = 0;
    // pinned host staging
    int32_t *h_done = nullptr;
    // state
    int cur_batch = 0, frames = 0, S = 0, last_frames = -1;
    bool have_mel = false, have_enc = false;
    // timings
    hipEvent_t ev[7]{};
    nh_timings tm{};
    bool profile_gemm = false;
    std::vector<hipEvent_t> gemm_ev;
    size_t gemm_ev_used = 0;
    double gemm_flops_acc = 0.0;

    int fail(int code, const std::string &msg) { err = msg; return code; }
};

// the captured decode-step graphs bake in everything the step kernels take by value (batch, encoder length, the rule
// token ids, the max_new_tokens knob): whoever changes one of those drops the graphs, the next greedy decode re-captures
static void drop_graphs(nh_ctx *ctx) {
    if (ctx->step_graph) { hipGraphExecDestroy(ctx->step_graph); ctx->step_graph = nullptr; }
    if (ctx->multi_graph) { hipGraphExecDestroy(ctx->multi_graph); ctx->multi_graph = nullptr; }
    for (int &k : ctx->graph_key) k = -1;
}

template <typename T>
static T *dalloc_into(std::vector<void *> &allocs, size_t n, bool zero = true) {
    void *p = nullptr;
    if (n == 0) n = 1;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
    if (zero) hipMemset(p, 0, n * sizeof(T));
    allocs.push_back(p);
    return reinterpret_cast<T *>(p);
}
template <typename T>
static T *dalloc(nh_ctx *ctx, size_t n, bool zero = true) { return dalloc_into<T>(ctx->allocs, n, zero); }

// ---- expected tensor names (the set candle reads) ---------------------------------
static void add_lin(std::set<std::string> &s, const std::string &p, bool bias = true) {
    s.insert(p + ".weight");
    if (bias) s.insert(p + ".bias");
}
static void add_attn(std::set<std::string> &s, const std::string &p) {
    add_lin(s, p + ".q_proj"); add_lin(s, p + ".k_proj", false); add_lin(s, p + ".v_proj"); add_lin(s, p + ".out_proj");
}
static void build_expected(nh_model *m) {
    auto &s = m->expected;
    add_lin(s, "model.encoder.conv1"); add_lin(s, "model.encoder.conv2");
    for (int i = 0; i < m->c.encoder_layers; i++) {
        std::string p = "model.encoder.layers." + std::to_string(i);
        add_attn(s, p + ".self_attn"); add_lin(s, p + ".self_attn_layer_norm");
        add_lin(s, p + ".fc1"); add_lin(s, p + ".fc2"); add_lin(s, p + ".final_layer_norm");
    }
    add_lin(s, "model.encoder.layer_norm");
    s.insert("model.decoder.embed_tokens.weight"); s.insert("model.decoder.embed_positions.weight");
    for (int i = 0; i < m->c.decoder_layers; i++) {
        std::string p = "model.decoder.layers." + std::to_string(i);
        add_attn(s, p + ".self_attn"); add_lin(s, p + ".self_attn_layer_norm");
        add_attn(s, p + ".encoder_attn"); add_lin(s, p + ".encoder_attn_layer_norm");
        add_lin(s, p + ".fc1"); add_lin(s, p + ".fc2"); add_lin(s, p + ".final_layer_norm");
    }
    add_lin(s, "model.decoder.layer_norm");
}

static long mel_frames_for(long n) {  // candle pcm_to_mel frame count (SURVEY.md 3.3[A]-2)
    long n_len = n / 160, pad = 1500;
    if (n_len % pad != 0) n_len = (n_len / pad + 1) * pad;
    return n_len + pad;
}

extern "C" int nh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *nh_last_error(const nh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" void nh_destroy(nh_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->dev);
    if (ctx->st) hipStreamSynchronize(ctx->st);
    for (void *p : ctx->allocs) hipFree(p);
    drop_graphs(ctx);
    if (ctx->h_done) hipHostFree(ctx->h_done);
    for (auto &e : ctx->ev) if (e) hipEventDestroy(e);
    for (auto &e : ctx->gemm_ev) hipEventDestroy(e);
    if (ctx->enc_done) hipEventDestroy(ctx->enc_done);
    if (ctx->st) hipStreamDestroy(ctx->st);   // sd is the same stream
    ctx->mdl.reset();  // the last context of a model frees its weights (~nh_model)
    delete ctx;
}

static bool build_mel_tables(nh_model *m, std::string &err) {
    // host tables with libm, the same f32 expressions candle evaluates (see k_mel.hip)
    std::vector<float> hann(400), dc(625), dsn(625), twc(375), tws(375);
    const float two_pi = (float)M_PI + (float)M_PI;
    for (int i = 0; i < 400; i++) hann[i] = 0.5f * (1.0f - cosf((two_pi * (float)i) / 400.0f));
    for (int k = 0; k < 25; k++)
        for (int j = 0; j < 25; j++) {
            float angle = two_pi * (float)k * (float)j / 25.0f;
            dc[k * 25 + j] = cosf(angle); dsn[k * 25 + j] = sinf(angle);
        }
    int off = 0;
    for (int h = 25; h <= 200; h *= 2) {
        float n_t = (float)(2 * h);
        for (int k = 0; k < h; k++) {
            float theta = two_pi * (float)k / n_t;
            twc[off + k] = cosf(theta); tws[off + k] = -sinf(theta);
        }
        off += h;
    }
    float *d_h = dalloc_into<float>(m->allocs, 400), *d_dc = dalloc_into<float>(m->allocs, 625), *d_ds = dalloc_into<float>(m->allocs, 625);
    float *d_tc = dalloc_into<float>(m->allocs, 375), *d_ts = dalloc_into<float>(m->allocs, 375);
    if (!d_h || !d_dc || !d_ds || !d_tc || !d_ts) { err = "hipMalloc(mel tables)"; return false; }
    if (hipMemcpy(d_h, hann.data(), 400 * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_dc, dc.data(), 625 * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ds, dsn.data(), 625 * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_tc, twc.data(), 375 * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ts, tws.data(), 375 * 4, hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy(mel tables)"; return false; }
    m->mt.hann = d_h; m->mt.dft_cos = d_dc; m->mt.dft_sin = d_ds; m->mt.tw_cos = d_tc; m->mt.tw_sin = d_ts;
    return true;
}

// the weight tables of one model on one device (zero-filled until nh_load_tensor fills them)
static std::shared_ptr<nh_model> build_model(int device_ordinal, const nh_config *cfg, std::string &err, int &code) {
    auto m = std::make_shared<nh_model>();
    m->dev = device_ordinal; m->c = *cfg;
    const int d = cfg->d_model, V = cfg->vocab_size, nm = cfg->num_mel_bins, ctxlen = cfg->max_target_positions;
    build_expected(m.get());
    bool ok = true;
#define MA(field, T, n) ok = ok && ((m->field = dalloc_into<T>(m->allocs, (size_t)(n))) != nullptr)
#define ML(f, T, n) ok = ok && ((L.f = dalloc_into<T>(m->allocs, (size_t)(n))) != nullptr)
    MA(conv1.w, half_t, (long)d * 3 * NH_MELP); MA(conv1.b, float, d);
    MA(conv2.w, half_t, (long)d * 3 * d); MA(conv2.b, float, d);
    MA(enc_pos, float, 1500L * d);
    m->enc.resize(cfg->encoder_layers);
    for (auto &L : m->enc) {
        ML(ln1.w, float, d); ML(ln1.b, float, d); ML(ln2.w, float, d); ML(ln2.b, float, d);
        ML(qkv.w, half_t, 3L * d * d); ML(qkv.b, float, 3 * d); ML(o.w, half_t, (long)d * d); ML(o.b, float, d);
        ML(fc1.w, half_t, 4L * d * d); ML(fc1.b, float, 4 * d); ML(fc2.w, half_t, 4L * d * d); ML(fc2.b, float, d);
    }
    MA(ln_post.w, float, d); MA(ln_post.b, float, d); MA(dec_ln.w, float, d); MA(dec_ln.b, float, d);
    MA(tok_emb, half_t, (long)V * d); MA(dec_pos, half_t, (long)ctxlen * d);
    m->dec.resize(cfg->decoder_layers);
    for (auto &L : m->dec) {
        ML(ln1.w, float, d); ML(ln1.b, float, d); ML(ln2.w, float, d); ML(ln2.b, float, d); ML(ln3.w, float, d); ML(ln3.b, float, d);
        ML(qkv.w, half_t, 3L * d * d); ML(qkv.b, float, 3 * d); ML(o.w, half_t, (long)d * d); ML(o.b, float, d);
        ML(cq.w, half_t, (long)d * d); ML(cq.b, float, d); ML(ckv.w, half_t, 2L * d * d); ML(ckv.b, float, 2 * d);
        ML(co.w, half_t, (long)d * d); ML(co.b, float, d);
        ML(fc1.w, half_t, 4L * d * d); ML(fc1.b, float, 4 * d); ML(fc2.w, half_t, 4L * d * d); ML(fc2.b, float, d);
    }
    MA(mel_grp, int32_t, 2 * nm);
#undef ML
#undef MA
    if (!ok) { err = "hipMalloc failed while sizing the model (out of device memory?)"; code = NH_ERR_NOMEM; return nullptr; }
    {   // encoder sinusoids, recomputed in f32 exactly as candle's sinusoids()
        std::vector<float> pos(1500L * d);
        int half = d / 2;
        float inc = logf(10000.0f) / (float)(half - 1);
        for (int p = 0; p < 1500; p++)
            for (int i = 0; i < half; i++) {
                float st = (float)p * expf((float)i * (-inc));
                pos[(long)p * d + i] = sinf(st);
                pos[(long)p * d + half + i] = cosf(st);
            }
        if (hipMemcpy(m->enc_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            err = "hipMemcpy(enc_pos) failed"; code = NH_ERR_HIP; return nullptr;
        }
    }
    if (!build_mel_tables(m.get(), err)) { code = NH_ERR_NOMEM; return nullptr; }
    return m;
}

// copy the model's pointer tables into the context's views (keeping the context's own K/V caches in dec[i])
static void refresh_views(nh_ctx *ctx) {
    nh_model &m = *ctx->mdl;
    std::lock_guard<std::mutex> lk(m.mu);
    ctx->conv1 = m.conv1; ctx->conv2 = m.conv2; ctx->enc_pos = m.enc_pos; ctx->enc = m.enc;
    ctx->ln_post = m.ln_post; ctx->dec_ln = m.dec_ln; ctx->tok_emb = m.tok_emb; ctx->dec_pos = m.dec_pos; ctx->tok_emb_t = m.tok_emb_t;
    ctx->dec.resize(m.dec.size());
    for (size_t i = 0; i < m.dec.size(); i++) {
        DecLayer v = m.dec[i];
        v.ck = ctx->dec[i].ck; v.cv = ctx->dec[i].cv; v.sk = ctx->dec[i].sk; v.sv = ctx->dec[i].sv;
        ctx->dec[i] = v;
    }
    ctx->view_version = m.version;
}
static inline void ensure_views(nh_ctx *ctx) { if (ctx->view_version != ctx->mdl->version) refresh_views(ctx); }

// stream, workspaces and caches of one context over an existing model
static int build_context(std::shared_ptr<nh_model> mdl, int max_batch, nh_ctx **out) {
    const nh_config *cfg = &mdl->c;
    const int d = cfg->d_model;
    nh_ctx *ctx = new nh_ctx();
    ctx->dev = mdl->dev; ctx->c = *cfg; ctx->B = max_batch; ctx->mdl = mdl;
    auto bail = [&](int code) { g_create_error = ctx->err; nh_destroy(ctx); return code; };
    if (hipSetDevice(ctx->dev) != hipSuccess) { ctx->err = "hipSetDevice failed"; return bail(NH_ERR_HIP); }
    // ONE stream per context (sd aliases st).  r02 gave the decode loop a stream of its own at the highest priority; r03
    // measured what that costs: the runtime backs every stream with an HSA queue, the queues are spread over the command
    // processor's pipes in CREATION ORDER, and when the decode streams of two contexts land on one pipe their kernel chains
    // take turns instead of overlapping -- three batches in flight ran at 4810 audio-s/s instead of 6330 after a harmless
    // reordering of nh_create (weights allocated before the streams), with no other change (profiles/r03_stream_order.txt:
    // two streams per context in r02's order 6332, without priorities 6332, one or two throw-away streams in front 6338 /
    // 6326, three 4782, decode stream created first 4785, ONE stream per context 6358).  Encoder and decode of one
    // context are sequential anyway; with one queue per context three contexts plus the null stream fit the four pipes.
    if (hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->enc_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&(ctx->kv_copied = std::make_shared<nh_ctx::EventBox>())->e, hipEventDisableTiming) != hipSuccess) {
        ctx->err = "hipStreamCreateWithFlags failed"; return bail(NH_ERR_HIP);
    }
    ctx->sd = ctx->st;
    for (auto &e : ctx->ev) hipEventCreate(&e);
    const int B = max_batch, V = cfg->vocab_size, nm = cfg->num_mel_bins, ctxlen = cfg->max_target_positions;
    const long M = (long)B * 1500;
    ctx->VP = (V + 63) & ~63;
    bool ok = true;
#define DA(field, T, n) ok = ok && ((ctx->field = dalloc<T>(ctx, (size_t)(n))) != nullptr)
    // this context's K/V caches: cross K/V of the current batch, self-attention cache
    ctx->dec.resize(cfg->decoder_layers);
    for (auto &L : ctx->dec) {
#define DL(f, T, n) ok = ok && ((L.f = dalloc<T>(ctx, (size_t)(n))) != nullptr)
        DL(ck, half_t, M * d); DL(cv, half_t, M * d);
        DL(sk, half_t, (long)B * ctxlen * d); DL(sv, half_t, (long)B * ctxlen * d);
#undef DL
    }
    // mel
    DA(pcm, float, (long)B * NH_N_SAMPLES); DA(nsamp, int32_t, B); DA(mel32, float, (long)B * nm * NH_N_FRAMES);
    DA(chunk_max, unsigned, B); DA(mel_img, half_t, (long)B * (NH_N_FRAMES + 2) * NH_MELP);
    // encoder
    DA(h1, half_t, (long)B * (NH_N_FRAMES + 2) * d); DA(x, float, M * d); DA(xn, half_t, M * d);
    DA(q, half_t, M * d); DA(k, half_t, M * d); DA(vt, half_t, (long)B * d * NH_SP); DA(att, half_t, M * d);
    DA(hid, half_t, M * 4 * d); DA(xa16, half_t, M * d); DA(xa32, float, M * d);
    // decoder
    DA(dx, float, (long)B * d); DA(dy32, float, (long)B * d); DA(logits, float, (long)B * ctx->VP);
    DA(dxn, half_t, (long)B * d); DA(dq, half_t, (long)B * d); DA(datt, half_t, (long)B * d); DA(dhid, half_t, (long)B * 4 * d);
    DA(ds.tokens, int32_t, (long)B * ctxlen); DA(ds.n_tokens, int32_t, B); DA(ds.done, int32_t, B);
    DA(ds.have_last, int32_t, B); DA(ds.last_ts, int32_t, B); DA(ds.sum_logprob, double, B); DA(ds.no_speech, double, B);
    DA(ds.n_active, int32_t, 1); DA(suppress, uint8_t, V); DA(lpart, float, (long)B * 64); DA(ltick, unsigned, B); DA(d_pos, int32_t, B); DA(d_lang_tokens, int32_t, 256); DA(d_lang_out, int32_t, B); DA(d_lang_probs, float, (long)B * 256);
#undef DA
    if (!ok) { ctx->err = "hipMalloc failed while sizing the context (out of device memory?)"; return bail(NH_ERR_NOMEM); }
    ctx->ds.suppress = ctx->suppress;
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_done), sizeof(int32_t) * 256, 0) != hipSuccess) {
        ctx->err = "hipHostMalloc failed"; return bail(NH_ERR_NOMEM);
    }
    refresh_views(ctx);
    *out = ctx;
    return NH_OK;
}

extern "C" int nh_create(int device_ordinal, const nh_config *cfg, int max_batch, nh_ctx **out) {
    if (!cfg || !out || max_batch < 1) { g_create_error = "nh_create: bad arguments"; return NH_ERR_INVALID; }
    const int d = cfg->d_model;
    if (d % 128 != 0 || d > 1280 || d / cfg->encoder_attention_heads != NH_DH ||
        d / cfg->decoder_attention_heads != NH_DH || cfg->max_source_positions != 1500 ||
        (cfg->num_mel_bins != 80 && cfg->num_mel_bins != 128) || max_batch > NH_MAX_BATCH) {
        g_create_error = "nh_create: unsupported config (need d_model % 128 == 0, d_model <= 1280, head dim 64, "
                         "max_source_positions 1500, num_mel_bins 80|128, max_batch <= 96)";
        return NH_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) {
        g_create_error = "nh_create: no HIP device with ordinal " + std::to_string(device_ordinal) +
                         " (SelectedDevice::Rocm needs a visible MI355X; there is no CPU fallback)";
        return NH_ERR_HIP;
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) { g_create_error = "hipSetDevice failed"; return NH_ERR_HIP; }
    std::string err; int code = NH_ERR_HIP;
    std::shared_ptr<nh_model> mdl = build_model(device_ordinal, cfg, err, code);
    if (!mdl) { g_create_error = err; return code; }
    return build_context(mdl, max_batch, out);
}

// A second (third ...) context on the SAME device over the SAME weights: own stream, workspaces, K/V caches, tokens and
// decode state; the model tables are shared and reference counted (freed with the last context).
extern "C" int nh_create_shared(nh_ctx *parent, int max_batch, nh_ctx **out) {
    if (!parent || !out || max_batch < 1 || max_batch > NH_MAX_BATCH) { g_create_error = "nh_create_shared: bad arguments (1 <= max_batch <= 96)"; return NH_ERR_INVALID; }
    return build_context(parent->mdl, max_batch, out);
}

// ---- weight loading ----------------------------------------------------------------------------------
static float host_elem(const void *data, int dtype, size_t i) {
    return dtype == NH_DTYPE_F32 ? reinterpret_cast<const float *>(data)[i]
                                 : (float)reinterpret_cast<const _Float16 *>(data)[i];
}
static int up_f16(nh_ctx *ctx, half_t *dst, const void *data, int dtype, size_t n) {
    if (dtype == NH_DTYPE_F16) { HIPCHK(hipMemcpy(dst, data, n * 2, hipMemcpyHostToDevice)); return NH_OK; }
    std::vector<_Float16> tmp(n);
    const float *s = reinterpret_cast<const float *>(data);
    for (size_t i = 0; i < n; i++) tmp[i] = (_Float16)s[i];
    HIPCHK(hipMemcpy(dst, tmp.data(), n * 2, hipMemcpyHostToDevice));
    return NH_OK;
}
static int up_f32(nh_ctx *ctx, float *dst, const void *data, int dtype, size_t n) {
    if (dtype == NH_DTYPE_F32) { HIPCHK(hipMemcpy(dst, data, n * 4, hipMemcpyHostToDevice)); return NH_OK; }
    std::vector<float> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = host_elem(data, dtype, i);
    HIPCHK(hipMemcpy(dst, tmp.data(), n * 4, hipMemcpyHostToDevice));
    return NH_OK;
}
// conv weight [co][ci][3] -> [co][kk * cpad + ci] fp16 (zero padded channels)
static int up_conv(nh_ctx *ctx, half_t *dst, const void *data, int dtype, int co, int ci, int cpad) {
    std::vector<_Float16> tmp((size_t)co * 3 * cpad, (_Float16)0.f);
    for (int o = 0; o < co; o++)
        for (int c = 0; c < ci; c++)
            for (int kk = 0; kk < 3; kk++)
                tmp[((size_t)o * 3 + kk) * cpad + c] = (_Float16)host_elem(data, dtype, ((size_t)o * ci + c) * 3 + kk);
    HIPCHK(hipMemcpy(dst, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    return NH_OK;
}

static bool starts(const std::string &s, const char *p, std::string &rest) {
    size_t n = strlen(p);
    if (s.compare(0, n, p) != 0) return false;
    rest = s.substr(n);
    return true;
}

extern "C" int nh_load_tensor(nh_ctx *ctx, const char *name_c, int dtype, const int64_t *shape, int ndim,
                              const void *data) {
    if (!ctx || !name_c || !data || !shape || (dtype != NH_DTYPE_F32 && dtype != NH_DTYPE_F16))
        return ctx ? ctx->fail(NH_ERR_INVALID, "nh_load_tensor: bad arguments") : NH_ERR_INVALID;
    hipSetDevice(ctx->dev);
    const std::string name(name_c);
    if (name == "model.encoder.embed_positions.weight" || name == "proj_out.weight") return NH_OK;  // not read by candle
    if (!ctx->mdl->expected.count(name)) return ctx->fail(NH_ERR_INVALID, "nh_load_tensor: unknown tensor name " + name);
    ensure_views(ctx);
    size_t n = 1;
    for (int i = 0; i < ndim; i++) n *= (size_t)shape[i];
    const int d = ctx->c.d_model;
    auto want = [&](size_t expect) -> int {
        if (n != expect)
            return ctx->fail(NH_ERR_INVALID, "nh_load_tensor: " + name + " has " + std::to_string(n) +
                                                 " elements, expected " + std::to_string(expect));
        return NH_OK;
    };
    int rc = NH_OK;
    std::string rest;
    auto lin = [&](LinW &L, const std::string &leaf, size_t n_out, size_t n_in, size_t row_off) -> int {
        if (leaf == "weight") { if ((rc = want(n_out * n_in))) return rc; return up_f16(ctx, L.w + row_off * n_in, data, dtype, n); }
        if (leaf == "bias") { if ((rc = want(n_out))) return rc; return up_f32(ctx, L.b + row_off, data, dtype, n); }
        return ctx->fail(NH_ERR_INVALID, "nh_load_tensor: unknown leaf in " + name);
    };
    auto ln = [&](LnW &L, const std::string &leaf) -> int {
        if ((rc = want(d))) return rc;
        if (leaf == "weight") return up_f32(ctx, L.w, data, dtype, n);
        if (leaf == "bias") return up_f32(ctx, L.b, data, dtype, n);
        return ctx->fail(NH_ERR_INVALID, "nh_load_tensor: unknown leaf in " + name);
    };
    if (name == "model.encoder.conv1.weight") { if (!(rc = want((size_t)d * ctx->c.num_mel_bins * 3))) rc = up_conv(ctx, ctx->conv1.w, data, dtype, d, ctx->c.num_mel_bins, NH_MELP); }
    else if (name == "model.encoder.conv1.bias") { if (!(rc = want(d))) rc = up_f32(ctx, ctx->conv1.b, data, dtype, n); }
    else if (name == "model.encoder.conv2.weight") { if (!(rc = want((size_t)d * d * 3))) rc = up_conv(ctx, ctx->conv2.w, data, dtype, d, d, d); }
    else if (name == "model.encoder.conv2.bias") { if (!(rc = want(d))) rc = up_f32(ctx, ctx->conv2.b, data, dtype, n); }
    else if (starts(name, "model.encoder.layer_norm.", rest)) rc = ln(ctx->ln_post, rest);
    else if (starts(name, "model.decoder.layer_norm.", rest)) rc = ln(ctx->dec_ln, rest);
    else if (name == "model.decoder.embed_tokens.weight") { if (!(rc = want((size_t)ctx->c.vocab_size * d))) rc = up_f16(ctx, ctx->tok_emb, data, dtype, n); }
    else if (name == "model.decoder.embed_positions.weight") { if (!(rc = want((size_t)ctx->c.max_target_positions * d))) rc = up_f16(ctx, ctx->dec_pos, data, dtype, n); }
    else {
        bool is_enc = starts(name, "model.encoder.layers.", rest);
        bool is_dec = !is_enc && starts(name, "model.decoder.layers.", rest);
        if (!is_enc && !is_dec) return ctx->fail(NH_ERR_INVALID, "nh_load_tensor: unhandled tensor " + name);
        size_t dot = rest.find('.');
        int idx = atoi(rest.substr(0, dot).c_str());
        std::string sub = rest.substr(dot + 1), leaf;
        if (is_enc) {
            EncLayer &L = ctx->enc[idx];
            if (starts(sub, "self_attn.q_proj.", leaf)) rc = lin(L.qkv, leaf, d, d, 0);
            else if (starts(sub, "self_attn.k_proj.", leaf)) rc = lin(L.qkv, leaf, d, d, d);
            else if (starts(sub, "self_attn.v_proj.", leaf)) rc = lin(L.qkv, leaf, d, d, 2 * d);
            else if (starts(sub, "self_attn.out_proj.", leaf)) rc = lin(L.o, leaf, d, d, 0);
            else if (starts(sub, "self_attn_layer_norm.", leaf)) rc = ln(L.ln1, leaf);
            else if (starts(sub, "fc1.", leaf)) rc = lin(L.fc1, leaf, 4 * d, d, 0);
            else if (starts(sub, "fc2.", leaf)) rc = lin(L.fc2, leaf, d, 4 * d, 0);
            else if (starts(sub, "final_layer_norm.", leaf)) rc = ln(L.ln2, leaf);
            else return ctx->fail(NH_ERR_INVALID, "nh_load_tensor: unhandled tensor " + name);
        } else {
            DecLayer &L = ctx->dec[idx];
            if (starts(sub, "self_attn.q_proj.", leaf)) rc = lin(L.qkv, leaf, d, d, 0);
            else if (starts(sub, "self_attn.k_proj.", leaf)) rc = lin(L.qkv, leaf, d, d, d);
            else if (starts(sub, "self_attn.v_proj.", leaf)) rc = lin(L.qkv, leaf, d, d, 2 * d);
            else if (starts(sub, "self_attn.out_proj.", leaf)) rc = lin(L.o, leaf, d, d, 0);
            else if (starts(sub, "self_attn_layer_norm.", leaf)) rc = ln(L.ln1, leaf);
            else if (starts(sub, "encoder_attn.q_proj.", leaf)) rc = lin(L.cq, leaf, d, d, 0);
            else if (starts(sub, "encoder_attn.k_proj.", leaf)) rc = lin(L.ckv, leaf, d, d, 0);
            else if (starts(sub, "encoder_attn.v_proj.", leaf)) rc = lin(L.ckv, leaf, d, d, d);
            else if (starts(sub, "encoder_attn.out_proj.", leaf)) rc = lin(L.co, leaf, d, d, 0);
            else if (starts(sub, "encoder_attn_layer_norm.", leaf)) rc = ln(L.ln2, leaf);
            else if (starts(sub, "fc1.", leaf)) rc = lin(L.fc1, leaf, 4 * d, d, 0);
            else if (starts(sub, "fc2.", leaf)) rc = lin(L.fc2, leaf, d, 4 * d, 0);
            else if (starts(sub, "final_layer_norm.", leaf)) rc = ln(L.ln3, leaf);
            else return ctx->fail(NH_ERR_INVALID, "nh_load_tensor: unhandled tensor " + name);
        }
    }
    if (rc == NH_OK) {
        std::lock_guard<std::mutex> lk(ctx->mdl->mu);
        ctx->mdl->loaded.insert(name); ctx->mdl->dec_tiled_valid = false;
    }
    return rc;
}

// The decoder's GEMVs stream every weight once per token: repack them tile-major (launch_repack_tiles) so that a wave
// instruction reads 1 KiB contiguous instead of 16 row pieces of 64 B.  Done lazily before the first decoder use and
// again after any nh_load_tensor; the row-major originals stay (embedding lookup, cross-K/V GEMM, re-loading).
static int ensure_decoder_repack(nh_ctx *ctx) {
    nh_model &m = *ctx->mdl;
    {
        std::lock_guard<std::mutex> lk(m.mu);   // contexts that share the model may get here together: one of them repacks
        if (!m.dec_tiled_valid) {
            const int d = ctx->c.d_model, V = ctx->c.vocab_size;
            bool moved = false;
            auto one = [&](half_t *&dst, const half_t *src, int N, int K) -> bool {
                if (!dst) { dst = dalloc_into<half_t>(m.allocs, (size_t)((N + 15) / 16) * 16 * K, false); moved = true; }
                if (!dst) return false;
                launch_repack_tiles(src, dst, N, K, ctx->sd);
                return true;
            };
            bool ok = one(m.tok_emb_t, m.tok_emb, V, d);
            for (auto &L : m.dec) {
                ok = ok && one(L.qkv.wt, L.qkv.w, 3 * d, d) && one(L.o.wt, L.o.w, d, d) && one(L.cq.wt, L.cq.w, d, d) &&
                     one(L.co.wt, L.co.w, d, d) && one(L.fc1.wt, L.fc1.w, 4 * d, d) && one(L.fc2.wt, L.fc2.w, d, 4 * d);
                // the cross K projection transposed ([feature][head dim]; NH_OPT_ABSORBED_XATTN reads Wk_h^T q from it)
                if (ok && !L.ckv.wt) { L.ckv.wt = dalloc_into<half_t>(m.allocs, (size_t)d * d, false); moved = true; }
                ok = ok && L.ckv.wt;
                if (ok) launch_transpose_sq(L.ckv.w, L.ckv.wt, d, ctx->sd);
            }
            if (!ok) return ctx->fail(NH_ERR_NOMEM, "hipMalloc(tile-major decoder weights)");
            HIPCHK(hipStreamSynchronize(ctx->sd));
            HIPCHK(hipGetLastError());
            m.dec_tiled_valid = true;
            if (moved) m.version++;
        }
    }
    ensure_views(ctx);
    return NH_OK;
}

extern "C" int nh_missing_tensors(const nh_ctx *ctx) {
    if (!ctx) return -1;
    std::lock_guard<std::mutex> lk(ctx->mdl->mu);
    return (int)(ctx->mdl->expected.size() - ctx->mdl->loaded.size());
}

extern "C" int nh_set_mel_filters(nh_ctx *ctx, const float *filters, int n_mel) {
    if (!ctx || !filters) return NH_ERR_INVALID;
    if (n_mel != ctx->c.num_mel_bins) return ctx->fail(NH_ERR_INVALID, "Unexpected number of mel bins (num_mel_bins), got: " + std::to_string(n_mel));
    hipSetDevice(ctx->dev);
    nh_model &m = *ctx->mdl;
    std::lock_guard<std::mutex> lk(m.mu);
    float *df = dalloc_into<float>(m.allocs, (size_t)n_mel * 201);
    if (!df) return ctx->fail(NH_ERR_NOMEM, "hipMalloc(mel filters)");
    HIPCHK(hipMemcpy(df, filters, (size_t)n_mel * 201 * 4, hipMemcpyHostToDevice));
    std::vector<int32_t> grp(2 * n_mel);
    for (int m = 0; m < n_mel; m++) {
        int g0 = 50, g1 = 0;
        for (int g = 0; g < 50; g++) {
            bool nz = false;
            for (int k = 4 * g; k < 4 * g + 4; k++) nz = nz || filters[(size_t)m * 201 + k] != 0.f;
            if (nz) { if (g < g0) g0 = g; g1 = g + 1; }
        }
        if (g0 > g1) g0 = g1 = 0;
        grp[2 * m] = g0; grp[2 * m + 1] = g1;
    }
    HIPCHK(hipMemcpy(m.mel_grp, grp.data(), grp.size() * 4, hipMemcpyHostToDevice));
    m.mt.filters = df;
    m.have_filters = true;
    return NH_OK;
}

extern "C" int nh_set_tokens(nh_ctx *ctx, const nh_tokens *tk, const int32_t *suppress_tokens, int n_suppress) {
    if (!ctx || !tk || (n_suppress > 0 && !suppress_tokens)) return NH_ERR_INVALID;
    const int V = ctx->c.vocab_size;
    auto inr = [&](int t) { return t >= 0 && t < V; };
    if (!inr(tk->sot) || !inr(tk->eot) || !inr(tk->task) || !inr(tk->no_speech) || !inr(tk->no_timestamps) ||
        !inr(tk->zero_sec) || !inr(tk->one_sec) || (tk->lang >= V))
        return ctx->fail(NH_ERR_INVALID, "nh_set_tokens: token id outside the vocabulary");
    hipSetDevice(ctx->dev);
    // monolingual.rs:386-395: suppress_tokens = config list U {no_timestamps}
    std::vector<uint8_t> sup(V, 0);
    for (int i = 0; i < n_suppress; i++) if (inr(suppress_tokens[i])) sup[suppress_tokens[i]] = 1;
    sup[tk->no_timestamps] = 1;
    HIPCHK(hipMemcpy(ctx->suppress, sup.data(), V, hipMemcpyHostToDevice));
    ctx->tk = RuleTokens{tk->sot, tk->eot, tk->lang, tk->task, tk->no_speech, tk->no_timestamps, tk->zero_sec, tk->one_sec};
    ctx->have_tokens = true;
    ctx->token_gen++;  // the captured step graphs carry the old ids by value (logit_step_kernel): re-capture
    drop_graphs(ctx);
    return NH_OK;
}

// ---- log-mel -------------------------------------------------------------------------------------------
// row0 > 0: the clips join the rows already filled (several encoder batches of one joint decode, nh_logmel_device_rows)
static int prepare_batch(nh_ctx *ctx, const int32_t *n_samples, int batch, int row0 = 0) {
    if (batch < 1 || row0 < 0 || row0 + batch > ctx->B) return ctx->fail(NH_ERR_INVALID, "rows [row0, row0 + batch) must lie in [0, max_batch]");
    long fr = -1;
    for (int b = 0; b < batch; b++) {
        if (n_samples[b] < 1 || n_samples[b] > NH_N_SAMPLES)
            return ctx->fail(NH_ERR_INVALID, "clip length must be in [1, 480000] samples");
        long f = mel_frames_for(n_samples[b]);
        if (f > NH_N_FRAMES) f = NH_N_FRAMES;  // narrow(2, 0, min(3000, frames)), model.rs:88
        if (fr < 0) fr = f;
        else if (fr != f) return ctx->fail(NH_ERR_INVALID, "clips of one batch must produce the same number of mel frames");
    }
    if (ctx->pool_rows > 0 && row0 >= ctx->pool_rows) {  // decode pool: encoder staging rows above the decoding ones
        if (ctx->frames < 0) {
            ctx->frames = (int)fr; ctx->S = (int)((fr + 2 - 3) / 2 + 1);
            if (ctx->frames != ctx->last_frames) {
                HIPCHK(hipMemsetAsync(ctx->mel_img, 0, sizeof(half_t) * (size_t)ctx->B * (NH_N_FRAMES + 2) * NH_MELP, ctx->st));
                HIPCHK(hipMemsetAsync(ctx->h1, 0, sizeof(half_t) * (size_t)ctx->B * (NH_N_FRAMES + 2) * ctx->c.d_model, ctx->st));
                ctx->last_frames = ctx->frames;
            }
        } else if ((int)fr != ctx->frames) return ctx->fail(NH_ERR_INVALID, "all clips of one decode pool must produce the same number of mel frames");
        if (row0 + batch > ctx->cur_batch) ctx->cur_batch = row0 + batch;
        ctx->have_enc = false;
        return NH_OK;
    }
    if (row0 > 0) {
        if (row0 > ctx->cur_batch) return ctx->fail(NH_ERR_STATE, "row0 leaves a gap after the rows filled so far");
        if ((int)fr != ctx->frames) return ctx->fail(NH_ERR_INVALID, "all rows of one joint decode must produce the same number of mel frames");
        if (row0 + batch > ctx->cur_batch) ctx->cur_batch = row0 + batch;
        ctx->have_enc = false;
        return NH_OK;
    }
    ctx->cur_batch = batch; ctx->frames = (int)fr; ctx->S = (int)((fr + 2 - 3) / 2 + 1);
    ctx->have_mel = false; ctx->have_enc = false;
    ctx->pool_rows = 0;  // a fresh batch ends a decode pool
    ctx->seq_lang.clear();
    if (ctx->frames != ctx->last_frames) {  // the zero rows framing each clip move with the frame count
        HIPCHK(hipMemsetAsync(ctx->mel_img, 0, sizeof(half_t) * (size_t)ctx->B * (NH_N_FRAMES + 2) * NH_MELP, ctx->st));
        HIPCHK(hipMemsetAsync(ctx->h1, 0, sizeof(half_t) * (size_t)ctx->B * (NH_N_FRAMES + 2) * ctx->c.d_model, ctx->st));
        ctx->last_frames = ctx->frames;
    }
    return NH_OK;
}

static int run_logmel(nh_ctx *ctx, const float *pcm_dev, const int32_t *n_samples, int64_t stride, int batch, int row0 = 0) {
    if (!ctx->mdl->have_filters) return ctx->fail(NH_ERR_STATE, "nh_logmel: mel filters not set");
    int rc = prepare_batch(ctx, n_samples, batch, row0);
    if (rc) return rc;
    const int nm = ctx->c.num_mel_bins;
    int32_t *nsamp = ctx->nsamp + row0;
    unsigned *cmax = ctx->chunk_max + row0;
    float *mel32 = ctx->mel32 + (size_t)row0 * nm * ctx->frames;
    half_t *img = ctx->mel_img + (size_t)row0 * (ctx->frames + 2) * NH_MELP;
    HIPCHK(hipMemcpyAsync(nsamp, n_samples, sizeof(int32_t) * batch, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->st));
    HIPCHK(hipMemsetAsync(cmax, 0, sizeof(unsigned) * batch, ctx->st));
    launch_logmel_grp(pcm_dev, nsamp, stride, ctx->mdl->mt, ctx->mdl->mel_grp, nm, ctx->frames, mel32, cmax, batch, ctx->st);
    launch_mel_finish_ex(mel32, cmax, img, batch, nm, ctx->frames, 1, ctx->st);
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->st));
    HIPCHK(hipGetLastError());
    ctx->have_mel = true;
    return NH_OK;
}

extern "C" int nh_logmel_device_rows(nh_ctx *ctx, const float *pcm_dev, const int32_t *n_samples, int64_t stride, int batch, int row0) {
    if (!ctx || !pcm_dev || !n_samples) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_logmel_device_rows: bad arguments") : NH_ERR_INVALID;
    hipSetDevice(ctx->dev);
    return run_logmel(ctx, pcm_dev, n_samples, stride, batch, row0);
}

extern "C" int nh_logmel_device(nh_ctx *ctx, const float *pcm_dev, const int32_t *n_samples, int64_t stride, int batch) {
    if (!ctx || !pcm_dev || !n_samples) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_logmel_device: bad arguments") : NH_ERR_INVALID;
    hipSetDevice(ctx->dev);
    return run_logmel(ctx, pcm_dev, n_samples, stride, batch);
}

static int logmel_host_rows(nh_ctx *ctx, const float *pcm, const int32_t *n_samples, int64_t stride, int batch, int row0) {
    hipSetDevice(ctx->dev);
    if (batch < 1 || row0 < 0 || row0 + batch > ctx->B) return ctx->fail(NH_ERR_INVALID, "rows [row0, row0 + batch) must lie in [0, max_batch]");
    float *dst = ctx->pcm + (size_t)row0 * NH_N_SAMPLES;
    for (int b = 0; b < batch; b++) {
        if (n_samples[b] < 1 || n_samples[b] > NH_N_SAMPLES) return ctx->fail(NH_ERR_INVALID, "clip length must be in [1, 480000] samples");
        HIPCHK(hipMemcpyAsync(dst + (size_t)b * NH_N_SAMPLES, pcm + (size_t)b * stride, sizeof(float) * n_samples[b],
                              hipMemcpyHostToDevice, ctx->st));
    }
    return run_logmel(ctx, dst, n_samples, NH_N_SAMPLES, batch, row0);
}

extern "C" int nh_logmel(nh_ctx *ctx, const float *pcm, const int32_t *n_samples, int64_t stride, int batch) {
    if (!ctx || !pcm || !n_samples) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_logmel: bad arguments") : NH_ERR_INVALID;
    return logmel_host_rows(ctx, pcm, n_samples, stride, batch, 0);
}

extern "C" int nh_logmel_rows(nh_ctx *ctx, const float *pcm, const int32_t *n_samples, int64_t stride, int batch, int row0) {
    if (!ctx || !pcm || !n_samples) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_logmel_rows: bad arguments") : NH_ERR_INVALID;
    return logmel_host_rows(ctx, pcm, n_samples, stride, batch, row0);
}

extern "C" int nh_sample_size(int dt) {
    switch (dt) {
        case NH_SAMPLE_F32: case NH_SAMPLE_I32: case NH_SAMPLE_U32: return 4;
        case NH_SAMPLE_F64: case NH_SAMPLE_I64: case NH_SAMPLE_U64: return 8;
        case NH_SAMPLE_I16: case NH_SAMPLE_U16: return 2;
        case NH_SAMPLE_I8: case NH_SAMPLE_U8: return 1;
        default: return 0;
    }
}

// src/dtype.rs + dasp_sample's Sample::to_sample::<f32> (src/lib.rs:180,207), on the device: the native samples cross PCIe
// as they are and become Model::Data (f32) in HBM
extern "C" int nh_logmel_samples(nh_ctx *ctx, const void *pcm, int sample_dtype, const int32_t *n_samples, int64_t stride, int batch) {
    if (!ctx || !pcm || !n_samples) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_logmel_samples: bad arguments") : NH_ERR_INVALID;
    const size_t es = (size_t)nh_sample_size(sample_dtype);
    if (!es) return ctx->fail(NH_ERR_INVALID, "nh_logmel_samples: unknown sample type " + std::to_string(sample_dtype));
    if (sample_dtype == NH_SAMPLE_F32) return nh_logmel(ctx, reinterpret_cast<const float *>(pcm), n_samples, stride, batch);
    hipSetDevice(ctx->dev);
    if (batch < 1 || batch > ctx->B) return ctx->fail(NH_ERR_INVALID, "batch must be in [1, max_batch]");
    const size_t need = (size_t)ctx->B * NH_N_SAMPLES * es;
    if (ctx->raw_bytes < need) {   // staging for the native samples, sized for the widest type seen so far
        void *p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) return ctx->fail(NH_ERR_NOMEM, "hipMalloc(native sample staging)");
        ctx->allocs.push_back(p);
        ctx->raw = p; ctx->raw_bytes = need;
    }
    for (int b = 0; b < batch; b++) {
        if (n_samples[b] < 1 || n_samples[b] > NH_N_SAMPLES) return ctx->fail(NH_ERR_INVALID, "clip length must be in [1, 480000] samples");
        char *dst = reinterpret_cast<char *>(ctx->raw) + (size_t)b * NH_N_SAMPLES * es;
        HIPCHK(hipMemcpyAsync(dst, reinterpret_cast<const char *>(pcm) + (size_t)b * (size_t)stride * es, (size_t)n_samples[b] * es,
                              hipMemcpyHostToDevice, ctx->st));
        launch_convert_samples(dst, ctx->pcm + (size_t)b * NH_N_SAMPLES, n_samples[b], sample_dtype, ctx->st);
    }
    HIPCHK(hipGetLastError());
    return run_logmel(ctx, ctx->pcm, n_samples, NH_N_SAMPLES, batch);
}

extern "C" int nh_get_mel(nh_ctx *ctx, int b, float *out) {
    if (!ctx || !out) return NH_ERR_INVALID;
    if (!ctx->have_mel || b < 0 || b >= ctx->cur_batch) return ctx->fail(NH_ERR_STATE, "nh_get_mel: no mel for that clip");
    hipSetDevice(ctx->dev);
    size_t per = (size_t)ctx->c.num_mel_bins * ctx->frames;
    HIPCHK(hipMemcpyAsync(out, ctx->mel32 + per * b, per * 4, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    return NH_OK;
}

extern "C" int nh_set_mel(nh_ctx *ctx, const float *mel, int batch) {
    if (!ctx || !mel) return NH_ERR_INVALID;
    hipSetDevice(ctx->dev);
    std::vector<int32_t> ns(batch > 0 ? batch : 1, NH_N_SAMPLES);
    int rc = prepare_batch(ctx, ns.data(), batch);
    if (rc) return rc;
    size_t per = (size_t)ctx->c.num_mel_bins * NH_N_FRAMES;
    HIPCHK(hipMemcpyAsync(ctx->mel32, mel, per * batch * 4, hipMemcpyHostToDevice, ctx->st));
    launch_mel_finish_ex(ctx->mel32, ctx->chunk_max, ctx->mel_img, batch, ctx->c.num_mel_bins, ctx->frames, 0, ctx->st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->st));
    ctx->have_mel = true;
    return NH_OK;
}

// ---- encoder ---------------------------------------------------------------------------------------------
static void gemm_prof_begin(nh_ctx *ctx) {
    if (!ctx->profile_gemm) return;
    if (ctx->gemm_ev_used + 2 > ctx->gemm_ev.size()) {
        for (int i = 0; i < 64; i++) { hipEvent_t e; hipEventCreate(&e); ctx->gemm_ev.push_back(e); }
    }
    hipEventRecord(ctx->gemm_ev[ctx->gemm_ev_used], ctx->st);
}
static void gemm_prof_end(nh_ctx *ctx, const GemmParams &p) {
    if (!ctx->profile_gemm) return;
    hipEventRecord(ctx->gemm_ev[ctx->gemm_ev_used + 1], ctx->st);
    ctx->gemm_ev_used += 2;
    ctx->gemm_flops_acc += 2.0 * (double)p.M * (double)p.N * (double)p.K;
}

static void gemm_plain(nh_ctx *ctx, const half_t *A, long lda, const LinW &W, int M, int N, int K, int epi, void *o0,
                       void *o1, void *o2, int seg_n, long ldo, int vt_seg, int head_major = 0, float seg0_scale = 0.f) {
    GemmParams p{};
    p.head_major = head_major; p.seg0_scale = seg0_scale;
    p.A = A; p.lda = lda; p.a_rpb = M; p.a_bstride = 0; p.W = W.w; p.bias = W.b; p.M = M; p.N = N; p.K = K; p.epi = epi;
    p.out[0] = o0; p.out[1] = o1; p.out[2] = o2; p.seg_n = seg_n; p.ldo = ldo; p.o_rpb = M; p.o_bstride = 0; p.o_off = 0;
    p.vt_seg = vt_seg; p.S = ctx->S; p.H = ctx->c.encoder_attention_heads; p.pos = nullptr;
    gemm_prof_begin(ctx);
    launch_gemm(p, ctx->st);
    gemm_prof_end(ctx, p);
}

// Type::encoder_forward + the cross K/V of every decoder layer for the clips in rows [row0, row0 + B) of the context
static int encode_rows(nh_ctx *ctx, int row0, int B) {
    if (!ctx->have_mel) return ctx->fail(NH_ERR_STATE, "nh_encode: call nh_logmel first");
    if (nh_missing_tensors(ctx) != 0) return ctx->fail(NH_ERR_STATE, "nh_encode: " + std::to_string(nh_missing_tensors(ctx)) + " tensors not loaded");
    if (row0 < 0 || B < 1 || row0 + B > ctx->cur_batch) return ctx->fail(NH_ERR_INVALID, "nh_encode_rows: rows outside the clips given to nh_logmel");
    hipSetDevice(ctx->dev);
    ensure_views(ctx);
    const int d = ctx->c.d_model, F = ctx->frames, S = ctx->S, H = ctx->c.encoder_attention_heads;
    const int M = B * S;
    const size_t r0 = (size_t)row0;
    // this group's slices of the per-clip workspaces
    half_t *const mel_img = ctx->mel_img + r0 * (F + 2) * NH_MELP, *const h1 = ctx->h1 + r0 * (F + 2) * d;
    float *const x = ctx->x + r0 * S * d, *const xa32 = ctx->xa32 + r0 * S * d;
    half_t *const xn = ctx->xn + r0 * S * d, *const q = ctx->q + r0 * S * d, *const k = ctx->k + r0 * S * d, *const att = ctx->att + r0 * S * d;
    half_t *const vt = ctx->vt + r0 * d * NH_SP, *const hid = ctx->hid + r0 * S * 4 * d, *const xa16 = ctx->xa16 + r0 * S * d;
    ctx->gemm_ev_used = 0; ctx->gemm_flops_acc = 0.0;
    {   // decode pools of other contexts that are still copying K/V out of these rows (nh_pool_admit_from) go first
        std::lock_guard<std::mutex> lk(ctx->readers_mu);
        for (auto &e : ctx->kv_readers) HIPCHK(hipStreamWaitEvent(ctx->st, e->e, 0));
        ctx->kv_readers.clear();
    }
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->st));
    {   // conv1 + GELU: A rows overlap (lda = 128, K = 3 * 128) inside the zero-framed mel image
        GemmParams p{};
        p.A = mel_img; p.lda = NH_MELP; p.a_rpb = F; p.a_bstride = (long)(F + 2) * NH_MELP;
        p.W = ctx->conv1.w; p.bias = ctx->conv1.b; p.M = B * F; p.N = d; p.K = 3 * NH_MELP; p.epi = EPI_GELU_F16;
        p.out[0] = h1; p.seg_n = d; p.ldo = d; p.o_rpb = F; p.o_bstride = F + 2; p.o_off = 1; p.vt_seg = -1;
        p.S = S; p.H = H;
        gemm_prof_begin(ctx); launch_gemm(p, ctx->st); gemm_prof_end(ctx, p);
    }
    {   // conv2 (stride 2) + GELU + transpose + sinusoid positions -> f32 residual stream
        GemmParams p{};
        p.A = h1; p.lda = 2L * d; p.a_rpb = S; p.a_bstride = (long)(F + 2) * d;
        p.W = ctx->conv2.w; p.bias = ctx->conv2.b; p.M = M; p.N = d; p.K = 3 * d; p.epi = EPI_CONV2_F32;
        p.out[0] = x; p.seg_n = d; p.ldo = d; p.o_rpb = M; p.o_bstride = 0; p.o_off = 0; p.vt_seg = -1;
        p.S = S; p.H = H; p.pos = ctx->enc_pos;
        gemm_prof_begin(ctx); launch_gemm(p, ctx->st); gemm_prof_end(ctx, p);
    }
    for (auto &L : ctx->enc) {
        launch_layernorm(x, L.ln1.w, L.ln1.b, xn, nullptr, M, d, ctx->st);
        // q leaves the GEMM as (x W_q + b_q) * dh^-1/2 * log2(e): candle's q * dh^-1/4 and k * dh^-1/4 and the
        // exp -> exp2 change of base, applied once in f32 before the one rounding to fp16 (k_attn_enc.hip)
        gemm_plain(ctx, xn, d, L.qkv, M, 3 * d, d, EPI_F16, q, k, vt, d, d, 2, 0, NH_ENC_Q_SCALE);
        launch_enc_attention(q, k, d, vt, att, d, B, S, H, ctx->st);
        gemm_plain(ctx, att, d, L.o, M, d, d, EPI_RESID_F32, x, nullptr, nullptr, d, d, -1);
        launch_layernorm(x, L.ln2.w, L.ln2.b, xn, nullptr, M, d, ctx->st);
        gemm_plain(ctx, xn, d, L.fc1, M, 4 * d, d, EPI_GELU_F16, hid, nullptr, nullptr, 4 * d, 4 * d, -1);
        gemm_plain(ctx, hid, 4 * d, L.fc2, M, d, 4 * d, EPI_RESID_F32, x, nullptr, nullptr, d, d, -1);
    }
    launch_layernorm(x, ctx->ln_post.w, ctx->ln_post.b, xa16, xa32, M, d, ctx->st);
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->st));
    // cross-attention K/V of every decoder layer (the flush = true work of MultiHeadAttention::forward), head-major
    // [b][h][S][64]: clip row0 starts at row0 * S * d
    for (auto &L : ctx->dec)
        gemm_plain(ctx, xa16, d, L.ckv, M, 2 * d, d, EPI_F16, L.ck + r0 * S * d, L.cv + r0 * S * d, nullptr, d, d, -1, 1);
    HIPCHK(hipEventRecord(ctx->ev[4], ctx->st));
    HIPCHK(hipEventRecord(ctx->enc_done, ctx->st));
    HIPCHK(hipGetLastError());
    ctx->have_enc = true;
    return NH_OK;
}

extern "C" int nh_encode(nh_ctx *ctx) {
    if (!ctx) return NH_ERR_INVALID;
    return encode_rows(ctx, 0, ctx->cur_batch);
}

extern "C" int nh_encode_rows(nh_ctx *ctx, int row0, int batch) {
    if (!ctx) return NH_ERR_INVALID;
    return encode_rows(ctx, row0, batch);
}

extern "C" int nh_encoder_output(nh_ctx *ctx, int b, float *out) {
    if (!ctx || !out) return NH_ERR_INVALID;
    if (!ctx->have_enc || b < 0 || b >= ctx->cur_batch) return ctx->fail(NH_ERR_STATE, "nh_encoder_output: no encoder output for that clip");
    hipSetDevice(ctx->dev);
    size_t per = (size_t)ctx->S * ctx->c.d_model;
    HIPCHK(hipMemcpyAsync(out, ctx->xa32 + per * b, per * 4, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    return NH_OK;
}

// ---- decoder ---------------------------------------------------------------------------------------------
#define NH_GRAPH_STEPS 8
static void skinny(nh_ctx *ctx, const half_t *x, long ldx, const LinW &W, int R, int N, int K, int epi, void *o0, void *o1,
                   void *o2, long ldo, int t0, int ctxlen, const int32_t *pos_ptr = nullptr, const float *ln_x = nullptr,
                   const float *ln_w = nullptr, const float *ln_b = nullptr) {
    SkinnyParams p{};
    p.pos_ptr = pos_ptr; p.ln_x = ln_x; p.ln_w = ln_w; p.ln_b = ln_b;
    p.x = x; p.ldx = ldx; p.W = W.w; p.Wt = W.wt; p.bias = W.b; p.R = R; p.N = N; p.K = K; p.epi = epi;
    p.out[0] = o0; p.out[1] = o1; p.out[2] = o2; p.ldo = ldo; p.d = ctx->c.d_model; p.t0 = t0; p.Tn = 1; p.ctx = ctxlen;
    launch_skinny(p, ctx->sd);
}

// every decoder LayerNorm uses the "sliced" summation tree (nh_kernels.h) when the width allows, so that the fused and
// the stand-alone forms, and every batch size, give bit-identical rows
static void dec_layernorm(nh_ctx *ctx, const LnW &ln, half_t *y, float *y32, int R, int K) {
    if (!launch_layernorm_sliced(ctx->dx, ln.w, ln.b, y, y32, R, K, ctx->sd)) launch_layernorm(ctx->dx, ln.w, ln.b, y, y32, R, K, ctx->sd);
}

// LayerNorm + projection of one decode step: fused into the skinny GEMM when the shape allows (the separate
// LayerNorm launch is pure latency at B rows), otherwise LayerNorm into dxn first
static void ln_skinny(nh_ctx *ctx, const LnW &ln, const LinW &W, int R, int N, int K, int epi, void *o0, void *o1, void *o2,
                      long ldo, int t0, int ctxlen, const int32_t *pos_ptr) {
    if (ctx->opt_fuse_ln && skinny_ln_supported(R, N, K)) {
        skinny(ctx, nullptr, K, W, R, N, K, epi, o0, o1, o2, ldo, t0, ctxlen, pos_ptr, ctx->dx, ln.w, ln.b);
    } else {
        dec_layernorm(ctx, ln, ctx->dxn, nullptr, R, K);
        skinny(ctx, ctx->dxn, K, W, R, N, K, epi, o0, o1, o2, ldo, t0, ctxlen, pos_ptr);
    }
}

// one decoder position for the whole batch: consumes tokens[b][pos], leaves the residual stream in dx.
// final_ln: also LN(dx) -> dxn (fp16) / dy32 (f32) (the teacher-forced view; the step path fuses it into the logits).
// pos_ptr != nullptr: the position comes from device memory (the step is being captured into a hipGraph).
// skip_done: finished sequences skip their attention (only inside decode_impl, where ds.done is live).
static void decoder_step(nh_ctx *ctx, int pos, const int32_t *pos_ptr = nullptr, bool final_ln = true, bool skip_done = false) {
    const int32_t *done = skip_done ? ctx->ds.done : nullptr;
    const int d = ctx->c.d_model, B = ctx->pool_rows > 0 ? ctx->pool_rows : ctx->cur_batch, H = ctx->c.decoder_attention_heads, C = ctx->c.max_target_positions;
    launch_embed(ctx->ds.tokens, C, ctx->tok_emb, ctx->dec_pos, ctx->dx, B, 1, pos, pos_ptr, d, ctx->sd);
    int nl = 0;
    for (auto &L : ctx->dec) {
        if (ctx->dec_layer_limit > 0 && nl++ >= ctx->dec_layer_limit) break;  // depth profile of the parity tests
        ln_skinny(ctx, L.ln1, L.qkv, B, 3 * d, d, SK_QKV, ctx->dq, L.sk, L.sv, d, pos, C, pos_ptr);
        launch_dec_attention(ctx->dq, L.sk, L.sv, ctx->datt, B, 1, H, d, C, pos + 1, pos_ptr, ctx->sd, 1, done);  // head-major cache
        skinny(ctx, ctx->datt, d, L.o, B, d, d, SK_RESID_F32, ctx->dx, nullptr, nullptr, d, 0, C);
        ln_skinny(ctx, L.ln2, L.cq, B, d, d, SK_F16, ctx->dq, nullptr, nullptr, d, 0, C, nullptr);
        if (ctx->opt_absorbed == 2) launch_xabs_attention_fast(ctx->dq, L.ckv.wt, L.ckv.w, L.ckv.b, ctx->xa16, ctx->xabs_u, ctx->xabs_z, ctx->xabs_ml, ctx->datt, B, H, d, ctx->S, done, ctx->sd);
        else if (ctx->opt_absorbed) launch_xabs_attention(ctx->dq, L.ckv.w, L.ckv.b, ctx->xa16, ctx->xabs_u, ctx->datt, B, H, d, ctx->S, done, ctx->sd);
        else launch_dec_attention(ctx->dq, L.ck, L.cv, ctx->datt, B, 1, H, d, ctx->S, ctx->S, nullptr, ctx->sd, 1, done);  // head-major cross K/V
        skinny(ctx, ctx->datt, d, L.co, B, d, d, SK_RESID_F32, ctx->dx, nullptr, nullptr, d, 0, C);
        ln_skinny(ctx, L.ln3, L.fc1, B, 4 * d, d, SK_GELU_F16, ctx->dhid, nullptr, nullptr, 4 * d, 0, C, nullptr);
        skinny(ctx, ctx->dhid, 4 * d, L.fc2, B, d, 4 * d, SK_RESID_F32, ctx->dx, nullptr, nullptr, d, 0, C);
    }
    if (final_ln) dec_layernorm(ctx, ctx->dec_ln, ctx->dxn, ctx->dy32, B, d);
}

// TextDecoder::final_linear on LN(dx) of the R rows of the last decoder_step(..., final_ln = false)
static void logits_from_dx(nh_ctx *ctx, int R) {
    LinW E; E.w = ctx->tok_emb; E.wt = ctx->tok_emb_t; E.b = nullptr;  // tied embedding, no bias (final_linear)
    const int d = ctx->c.d_model, V = ctx->c.vocab_size;
    if (ctx->opt_fuse_ln && skinny_ln_supported(R, V, d)) {
        skinny(ctx, nullptr, d, E, R, V, d, SK_F32, ctx->logits, nullptr, nullptr, ctx->VP, 0, 0, nullptr, ctx->dx, ctx->dec_ln.w, ctx->dec_ln.b);
    } else {
        dec_layernorm(ctx, ctx->dec_ln, ctx->dxn, nullptr, R, d);
        skinny(ctx, ctx->dxn, d, E, R, V, d, SK_F32, ctx->logits, nullptr, nullptr, ctx->VP, 0, 0);
    }
}

static void logits_from_dxn(nh_ctx *ctx, int R) {
    LinW E; E.w = ctx->tok_emb; E.wt = ctx->tok_emb_t; E.b = nullptr;
    skinny(ctx, ctx->dxn, ctx->c.d_model, E, R, ctx->c.vocab_size, ctx->c.d_model, SK_F32, ctx->logits, nullptr, nullptr,
           ctx->VP, 0, 0);
}

// what Model::decode returns for one sequence from the state its loop left behind (model.rs:308-315, 373-381)
static void finish_sequence(nh_ctx *ctx, int32_t *t, int n, int done, double slp, double nsp, int32_t *out_tokens, nh_decode_result &r) {
    const int C = ctx->c.max_target_positions;
    r.no_speech_prob = nsp;
    r.no_speech_exit = (done == 2);
    if (done == 2) { r.avg_logprob = 0.0; }  // model.rs:308-315
    else {
        r.avg_logprob = slp / (double)n;  // model.rs:373 (prompt and eot count)
        while (n >= 2 && t[n - 2] > ctx->tk.no_timestamps) { t[n - 2] = t[n - 1]; n--; }  // :375-381
    }
    r.n_tokens = n;
    memcpy(out_tokens, t, sizeof(int32_t) * C);
    for (int i = n; i < C; i++) out_tokens[i] = 0;
}

static int capture_step_graphs(nh_ctx *ctx, int B, int max_new_tokens, int P, int mode);

// Model::decode (model.rs:279-389) for the whole batch.  inv_t == 0: t = 0, greedy (hipGraph replay); inv_t > 0: every
// token is sampled at temperature 1 / inv_t under the seeded contract (eager launches: the fallback path is rare).
static int decode_impl(nh_ctx *ctx, int32_t *out_tokens, nh_decode_result *results, int max_new_tokens, float inv_t,
                       unsigned long long seed, unsigned clip0, unsigned attempt) {
    if (!ctx || !out_tokens || !results) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_decode: bad arguments") : NH_ERR_INVALID;
    if (!ctx->have_enc) return ctx->fail(NH_ERR_STATE, "nh_decode: call nh_encode first");
    if (!ctx->have_tokens) return ctx->fail(NH_ERR_STATE, "nh_decode: call nh_set_tokens first");
    if (ctx->pool_rows > 0) return ctx->fail(NH_ERR_STATE, "nh_decode: the context runs a decode pool (nh_pool_begin); a batch submitted with row0 = 0 ends it");
    hipSetDevice(ctx->dev);
    if (int rc = ensure_decoder_repack(ctx)) return rc;
    const int B = ctx->cur_batch, C = ctx->c.max_target_positions, cap = C - 1, V = ctx->c.vocab_size;
    // model.rs:285-289: prompt = [sot, lang?, task]
    const bool per_seq = (int)ctx->seq_lang.size() == B;
    const int P = (per_seq || ctx->tk.lang >= 0) ? 3 : 2;  // 2 or 3, so position 0 is never a generation step
    std::vector<int32_t> toks((size_t)B * C, 0), nt(B, P);
    for (int b = 0; b < B; b++) {
        int32_t *t = toks.data() + (size_t)b * C;
        int i = 0;
        t[i++] = ctx->tk.sot;
        if (P == 3) t[i++] = per_seq ? ctx->seq_lang[b] : ctx->tk.lang;
        t[i++] = ctx->tk.task;
    }
    HIPCHK(hipStreamWaitEvent(ctx->sd, ctx->enc_done, 0));  // the encoder and cross K/V ran on the other stream
    HIPCHK(hipMemcpyAsync(ctx->ds.tokens, toks.data(), toks.size() * 4, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipMemcpyAsync(ctx->ds.n_tokens, nt.data(), B * 4, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->ds.done, 0, B * 4, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->ltick, 0, B * 4, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->ds.have_last, 0, B * 4, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->ds.last_ts, 0, B * 4, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->ds.sum_logprob, 0, B * 8, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->ds.no_speech, 0, B * 8, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));  // toks/nt are stack-owned host buffers
    HIPCHK(hipEventRecord(ctx->ev[5], ctx->sd));
    int steps = 0;
    // Prompt phase (eager): position pos consumes tokens[pos]; pos 0 also yields no_speech_prob
    // (model.rs:293-305: logits at position 0 of the flush = true pass).
    for (int pos = 0; pos < P - 1; pos++) {
        decoder_step(ctx, pos, nullptr, true, true);
        steps++;
        if (pos == 0) {
            logits_from_dxn(ctx, B);
            launch_logit_step(ctx->logits, V, ctx->ds, ctx->tk, B, C, cap, max_new_tokens, P, 0, ctx->lpart, ctx->ltick, nullptr, ctx->sd);
        }
    }
    // Generation phase: one token per step from pos = P-1 on.  The length cap (model.rs:367) forces eot once
    // pos + 2 >= cap, so pos never exceeds cap - 2.  The ~20-launch step is captured (once, and 8 steps back to back) into hipGraphs that
    // reads the position from device memory (the eager loop is host-launch-bound at ~5 us per tiny kernel).
    const bool no_graph = !ctx->opt_graphs || inv_t > 0.f;
    const int key[6] = {B, ctx->S, max_new_tokens, P, ctx->token_gen, 0};
    if (!no_graph && memcmp(key, ctx->graph_key, sizeof(key)) != 0) {
        drop_graphs(ctx);
        if (int rc = capture_step_graphs(ctx, B, max_new_tokens, P, 1)) return rc;
        memcpy(ctx->graph_key, key, sizeof(key));
    }
    const int32_t first_pos = P - 1;
    for (int b = 0; b < B; b++) ctx->h_done[128 + b] = first_pos;  // every sequence of a batch starts generating at the same position
    HIPCHK(hipMemcpyAsync(ctx->d_pos, ctx->h_done + 128, sizeof(int32_t) * B, hipMemcpyHostToDevice, ctx->sd));
    // positions first_pos .. cap - 2; the host looks at the done flags every 16 steps (and after the last one)
    for (int pos = first_pos; pos <= cap - 2;) {
        int n = 1;
        if (no_graph) {
            decoder_step(ctx, pos, nullptr, false, true);
            logits_from_dx(ctx, B);
            if (inv_t > 0.f) launch_sample_step(ctx->logits, V, ctx->ds, ctx->tk, B, C, cap, max_new_tokens, P, inv_t, seed, clip0, attempt, ctx->sd);
            else launch_logit_step(ctx->logits, V, ctx->ds, ctx->tk, B, C, cap, max_new_tokens, P, 1, ctx->lpart, ctx->ltick, nullptr, ctx->sd);
        } else if (pos + NH_GRAPH_STEPS - 1 <= cap - 2) {
            HIPCHK(hipGraphLaunch(ctx->multi_graph, ctx->sd));
            n = NH_GRAPH_STEPS;
        } else {
            HIPCHK(hipGraphLaunch(ctx->step_graph, ctx->sd));
        }
        steps += n;
        const int before = pos - first_pos;
        pos += n;
        if ((before >> 4) != ((pos - first_pos) >> 4) || pos > cap - 2) {
            HIPCHK(hipMemcpyAsync(ctx->h_done, ctx->ds.done, B * 4, hipMemcpyDeviceToHost, ctx->sd));
            HIPCHK(hipStreamSynchronize(ctx->sd));
            bool all = true;
            for (int b = 0; b < B; b++) all = all && ctx->h_done[b] != 0;
            if (all) break;
        }
    }
    HIPCHK(hipEventRecord(ctx->ev[6], ctx->sd));
    std::vector<int32_t> done(B), hl(B);
    std::vector<double> slp(B), nsp(B);
    HIPCHK(hipMemcpyAsync(toks.data(), ctx->ds.tokens, toks.size() * 4, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(nt.data(), ctx->ds.n_tokens, B * 4, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(done.data(), ctx->ds.done, B * 4, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(slp.data(), ctx->ds.sum_logprob, B * 8, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(nsp.data(), ctx->ds.no_speech, B * 8, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    HIPCHK(hipGetLastError());
    for (int b = 0; b < B; b++)
        finish_sequence(ctx, toks.data() + (size_t)b * C, nt[b], done[b], slp[b], nsp[b], out_tokens + (size_t)b * C, results[b]);
    ctx->tm.decode_steps = steps;
    return NH_OK;
}

extern "C" int nh_decode_greedy(nh_ctx *ctx, int32_t *out_tokens, nh_decode_result *results, int max_new_tokens) {
    return decode_impl(ctx, out_tokens, results, max_new_tokens, 0.f, 0, 0, 0);
}

extern "C" int nh_decode_sampled(nh_ctx *ctx, int32_t *out_tokens, nh_decode_result *results, int max_new_tokens,
                                 float temperature, uint64_t seed, uint32_t clip0, uint32_t attempt) {
    if (ctx && !(temperature > 0.f)) return ctx->fail(NH_ERR_INVALID, "nh_decode_sampled: temperature must be > 0 (use nh_decode_greedy for t = 0)");
    return decode_impl(ctx, out_tokens, results, max_new_tokens, 1.0f / temperature, seed, clip0, attempt);
}

// ---- decode pool -------------------------------------------------------------------------------------------
// The reference's loop ends per sequence at eot (model.rs:317), so the sequences of a batch do not finish together.
// Rows [0, rows) of the context decode, every row at its own position; a finished row is handed back (nh_pool_collect)
// and refilled (nh_pool_admit) from the encoder staging rows [rows, max_batch) while the others go on.  Every row's
// arithmetic is what it is in nh_decode_greedy -- the step kernels are the same, only the position is per row.
extern "C" int nh_pool_begin(nh_ctx *ctx, int rows, int max_new_tokens, int per_clip_language) {
    if (!ctx) return NH_ERR_INVALID;
    if (rows < 1 || rows >= ctx->B) return ctx->fail(NH_ERR_INVALID, "nh_pool_begin: rows must lie in [1, max_batch - 1] (the rows above are encoder staging)");
    if (!ctx->have_tokens) return ctx->fail(NH_ERR_STATE, "nh_pool_begin: call nh_set_tokens first");
    hipSetDevice(ctx->dev);
    if (int rc = ensure_decoder_repack(ctx)) return rc;
    if (ctx->opt_absorbed) return ctx->fail(NH_ERR_STATE, "nh_pool_begin: the NH_OPT_ABSORBED_XATTN prototype covers lockstep decodes only");
    ctx->pool_rows = rows; ctx->pool_max_new = max_new_tokens;
    ctx->pool_prompt = (per_clip_language || ctx->tk.lang >= 0) ? 3 : 2;
    ctx->pool_per_clip_language = per_clip_language != 0;
    ctx->pool_busy.assign(rows, 0);
    ctx->cur_batch = rows; ctx->frames = -1; ctx->S = 0; ctx->have_mel = false; ctx->have_enc = false;
    ctx->seq_lang.clear();
    for (int b = 0; b < rows; b++) ctx->h_done[128 + b] = 3;  // 3: empty row (skipped like a finished one)
    HIPCHK(hipMemcpyAsync(ctx->ds.done, ctx->h_done + 128, sizeof(int32_t) * rows, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->d_pos, 0, sizeof(int32_t) * rows, ctx->sd));
    HIPCHK(hipMemsetAsync(ctx->ltick, 0, sizeof(unsigned) * rows, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    return NH_OK;
}

static int pool_admit_impl(nh_ctx *ctx, nh_ctx *src, int src_row, int dst_row, int32_t lang) {
    if (dst_row < 0 || dst_row >= ctx->pool_rows || ctx->pool_busy[dst_row]) return ctx->fail(NH_ERR_INVALID, "nh_pool_admit: dst_row is not a free row of the pool");
    const int P = ctx->pool_prompt;
    if (!ctx->pool_per_clip_language && lang >= 0) return ctx->fail(NH_ERR_INVALID, "nh_pool_admit: the pool was begun without per-clip languages");
    int32_t lg = lang >= 0 ? lang : ctx->tk.lang;
    if (P == 3 && (lg < 0 || lg >= ctx->c.vocab_size)) return ctx->fail(NH_ERR_INVALID, "nh_pool_admit: language token outside the vocabulary");
    hipSetDevice(ctx->dev);
    const size_t per = (size_t)ctx->S * ctx->c.d_model;  // cross K / V of one clip and layer, head-major [h][S][64]
    if (src != ctx) HIPCHK(hipStreamWaitEvent(ctx->sd, src->enc_done, 0));   // the other context's encoder ran on its own stream
    for (size_t l = 0; l < ctx->dec.size(); l++) {
        auto &L = ctx->dec[l]; auto &Ls = src->dec[l];
        HIPCHK(hipMemcpyAsync(L.ck + per * dst_row, Ls.ck + per * src_row, per * sizeof(half_t), hipMemcpyDeviceToDevice, ctx->sd));
        HIPCHK(hipMemcpyAsync(L.cv + per * dst_row, Ls.cv + per * src_row, per * sizeof(half_t), hipMemcpyDeviceToDevice, ctx->sd));
    }
    if (src != ctx) {   // src's next encoder submission waits for these copies
        HIPCHK(hipEventRecord(ctx->kv_copied->e, ctx->sd));
        std::lock_guard<std::mutex> lk(src->readers_mu);
        if (std::find(src->kv_readers.begin(), src->kv_readers.end(), ctx->kv_copied) == src->kv_readers.end()) src->kv_readers.push_back(ctx->kv_copied);
    }
    // model.rs:285-289: prompt = [sot, lang?, task]
    launch_pool_admit(ctx->ds, ctx->d_pos, ctx->ltick, dst_row, ctx->c.max_target_positions, ctx->tk.sot, P == 3 ? lg : ctx->tk.task,
                      ctx->tk.task, P, ctx->sd);
    HIPCHK(hipGetLastError());
    ctx->pool_busy[dst_row] = 1;
    return NH_OK;
}

extern "C" int nh_pool_admit(nh_ctx *ctx, int src_row, int dst_row, int32_t lang) {
    if (!ctx) return NH_ERR_INVALID;
    if (ctx->pool_rows < 1) return ctx->fail(NH_ERR_STATE, "nh_pool_admit: no decode pool (nh_pool_begin)");
    if (!ctx->have_enc || src_row < ctx->pool_rows || src_row >= ctx->cur_batch) return ctx->fail(NH_ERR_STATE, "nh_pool_admit: src_row is not an encoded staging row (nh_encode_rows)");
    return pool_admit_impl(ctx, ctx, src_row, dst_row, lang);
}

extern "C" int nh_pool_admit_from(nh_ctx *ctx, nh_ctx *enc, int src_row, int dst_row, int32_t lang) {
    if (!ctx || !enc) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_pool_admit_from: bad arguments") : NH_ERR_INVALID;
    if (enc == ctx) return nh_pool_admit(ctx, src_row, dst_row, lang);
    if (ctx->pool_rows < 1) return ctx->fail(NH_ERR_STATE, "nh_pool_admit_from: no decode pool (nh_pool_begin)");
    if (enc->mdl != ctx->mdl || enc->dev != ctx->dev) return ctx->fail(NH_ERR_INVALID, "nh_pool_admit_from: the encoder context must share this context's weights (nh_create_shared)");
    if (enc->pool_rows > 0) return ctx->fail(NH_ERR_INVALID, "nh_pool_admit_from: the encoder context runs a pool of its own");
    if (!enc->have_enc || src_row < 0 || src_row >= enc->cur_batch) return ctx->fail(NH_ERR_STATE, "nh_pool_admit_from: src_row is not an encoded row of the encoder context (nh_encode / nh_encode_rows)");
    if (ctx->frames < 0) { ctx->frames = enc->frames; ctx->S = enc->S; }   // the pool's clip length is its first clip's
    else if (enc->frames != ctx->frames) return ctx->fail(NH_ERR_INVALID, "all clips of one decode pool must produce the same number of mel frames");
    return pool_admit_impl(ctx, enc, src_row, dst_row, lang);
}

static int capture_step_graphs(nh_ctx *ctx, int B, int max_new_tokens, int P, int mode) {
    const int C = ctx->c.max_target_positions, cap = C - 1, V = ctx->c.vocab_size;
    for (int which = 0; which < 2; which++) {
        hipGraph_t g = nullptr;
        hipError_t ge = hipStreamBeginCapture(ctx->sd, hipStreamCaptureModeThreadLocal);
        if (ge != hipSuccess) return ctx->fail(NH_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(ge));
        for (int i = 0; i < (which ? NH_GRAPH_STEPS : 1); i++) {  // every step reads and advances the device-side positions
            decoder_step(ctx, 0, ctx->d_pos, false, true);
            logits_from_dx(ctx, B);
            launch_logit_step(ctx->logits, V, ctx->ds, ctx->tk, B, C, cap, max_new_tokens, P, mode, ctx->lpart, ctx->ltick, ctx->d_pos, ctx->sd);
        }
        // the stream must leave capture mode whatever happened in between; a failed capture leaves no graph behind
        ge = hipStreamEndCapture(ctx->sd, &g);
        if (ge == hipSuccess && !g) ge = hipErrorStreamCaptureInvalidated;
        if (ge == hipSuccess) {
            ge = hipGraphInstantiate(which ? &ctx->multi_graph : &ctx->step_graph, g, nullptr, nullptr, 0);
            if (ge != hipSuccess) (which ? ctx->multi_graph : ctx->step_graph) = nullptr;
        }
        if (g) hipGraphDestroy(g);
        if (ge != hipSuccess) {
            (void)hipGetLastError();
            drop_graphs(ctx);
            return ctx->fail(NH_ERR_HIP, std::string("decode-step graph capture: ") + hipGetErrorString(ge));
        }
    }
    return NH_OK;
}

extern "C" int nh_pool_step(nh_ctx *ctx, int n_steps, int32_t *done_out) {
    if (!ctx || !done_out) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_pool_step: bad arguments") : NH_ERR_INVALID;
    if (ctx->pool_rows < 1) return ctx->fail(NH_ERR_STATE, "nh_pool_step: no decode pool (nh_pool_begin)");
    if (n_steps < 0) return ctx->fail(NH_ERR_INVALID, "nh_pool_step: n_steps < 0");
    const int B = ctx->pool_rows, C = ctx->c.max_target_positions, cap = C - 1, V = ctx->c.vocab_size, P = ctx->pool_prompt;
    hipSetDevice(ctx->dev);
    bool any = false;
    for (int b = 0; b < B; b++) any = any || ctx->pool_busy[b];
    if (any && n_steps > 0) {
        if (ctx->S < 1) return ctx->fail(NH_ERR_STATE, "nh_pool_step: rows are busy but nothing was ever encoded");
        if (ctx->opt_graphs) {
            const int key[6] = {B, ctx->S, ctx->pool_max_new, P, ctx->token_gen, 1};
            if (memcmp(key, ctx->graph_key, sizeof(key)) != 0) {
                drop_graphs(ctx);
                if (int rc = capture_step_graphs(ctx, B, ctx->pool_max_new, P, 2)) return rc;
                memcpy(ctx->graph_key, key, sizeof(key));
            }
        }
        for (int left = n_steps; left > 0;) {
            if (!ctx->opt_graphs) {
                decoder_step(ctx, 0, ctx->d_pos, false, true);
                logits_from_dx(ctx, B);
                launch_logit_step(ctx->logits, V, ctx->ds, ctx->tk, B, C, cap, ctx->pool_max_new, P, 2, ctx->lpart, ctx->ltick, ctx->d_pos, ctx->sd);
                left--;
            } else if (left >= NH_GRAPH_STEPS) { HIPCHK(hipGraphLaunch(ctx->multi_graph, ctx->sd)); left -= NH_GRAPH_STEPS; }
            else { HIPCHK(hipGraphLaunch(ctx->step_graph, ctx->sd)); left--; }
        }
        ctx->tm.decode_steps += n_steps;
    }
    HIPCHK(hipMemcpyAsync(ctx->h_done, ctx->ds.done, B * 4, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    HIPCHK(hipGetLastError());
    for (int b = 0; b < B; b++) done_out[b] = ctx->pool_busy[b] ? ctx->h_done[b] : 3;
    return NH_OK;
}

extern "C" int nh_pool_collect(nh_ctx *ctx, const int32_t *rows, int n, int32_t *out_tokens, nh_decode_result *results) {
    if (!ctx || !rows || !out_tokens || !results || n < 1) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_pool_collect: bad arguments") : NH_ERR_INVALID;
    if (ctx->pool_rows < 1) return ctx->fail(NH_ERR_STATE, "nh_pool_collect: no decode pool (nh_pool_begin)");
    const int B = ctx->pool_rows, C = ctx->c.max_target_positions;
    for (int i = 0; i < n; i++)
        if (rows[i] < 0 || rows[i] >= B || !ctx->pool_busy[rows[i]]) return ctx->fail(NH_ERR_INVALID, "nh_pool_collect: not a busy row of the pool");
    hipSetDevice(ctx->dev);
    std::vector<int32_t> toks((size_t)n * C), nt(B), done(B);
    std::vector<double> slp(B), nsp(B);
    for (int i = 0; i < n; i++)
        HIPCHK(hipMemcpyAsync(toks.data() + (size_t)i * C, ctx->ds.tokens + (size_t)rows[i] * C, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(nt.data(), ctx->ds.n_tokens, B * 4, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(done.data(), ctx->ds.done, B * 4, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(slp.data(), ctx->ds.sum_logprob, B * 8, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(nsp.data(), ctx->ds.no_speech, B * 8, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    for (int i = 0; i < n; i++)
        if (done[rows[i]] != 1 && done[rows[i]] != 2) return ctx->fail(NH_ERR_STATE, "nh_pool_collect: that row has not finished (see nh_pool_step's done flags)");
    for (int i = 0; i < n; i++) {
        const int b = rows[i];
        finish_sequence(ctx, toks.data() + (size_t)i * C, nt[b], done[b], slp[b], nsp[b], out_tokens + (size_t)i * C, results[i]);
        ctx->pool_busy[b] = 0;
    }
    return NH_OK;
}

extern "C" int nh_sample_rules(nh_ctx *ctx, const float *probs, const int32_t *tokens, int n_tokens, int last_timestamp,
                               float temperature, uint64_t seed, uint32_t clip, uint32_t attempt, int32_t *token_out) {
    if (!ctx || !probs || !tokens || !token_out || n_tokens < 1 || !(temperature > 0.f))
        return ctx ? ctx->fail(NH_ERR_INVALID, "nh_sample_rules: bad arguments") : NH_ERR_INVALID;
    if (!ctx->have_tokens) return ctx->fail(NH_ERR_STATE, "nh_sample_rules: call nh_set_tokens first");
    hipSetDevice(ctx->dev);
    const int V = ctx->c.vocab_size;
    HIPCHK(hipMemcpyAsync(ctx->logits, probs, sizeof(float) * V, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipMemcpyAsync(ctx->ds.tokens, tokens, sizeof(int32_t) * n_tokens, hipMemcpyHostToDevice, ctx->sd));
    launch_sample_rules(ctx->logits, ctx->ds.n_active, ctx->ds.tokens, n_tokens, last_timestamp, ctx->suppress, ctx->tk, V,
                        1.0f / temperature, seed, clip, attempt, ctx->sd);
    HIPCHK(hipMemcpyAsync(token_out, ctx->ds.n_active, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    HIPCHK(hipGetLastError());
    return NH_OK;
}

extern "C" int nh_set_languages(nh_ctx *ctx, const int32_t *langs) {
    if (!ctx) return NH_ERR_INVALID;
    ctx->seq_lang.clear();
    if (langs) {
        for (int b = 0; b < ctx->cur_batch; b++) {
            if (langs[b] < 0 || langs[b] >= ctx->c.vocab_size) { ctx->seq_lang.clear(); return ctx->fail(NH_ERR_INVALID, "nh_set_languages: token id outside the vocabulary"); }
            ctx->seq_lang.push_back(langs[b]);
        }
    }
    return NH_OK;
}

extern "C" int nh_detect_language(nh_ctx *ctx, const int32_t *lang_tokens, int n, int32_t *out_lang, float *out_probs) {
    if (!ctx || !lang_tokens || !out_lang || n < 1 || n > 256) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_detect_language: bad arguments (1 <= n <= 256)") : NH_ERR_INVALID;
    if (!ctx->have_enc) return ctx->fail(NH_ERR_STATE, "nh_detect_language: call nh_encode first");
    if (!ctx->have_tokens) return ctx->fail(NH_ERR_STATE, "nh_detect_language: call nh_set_tokens first");
    hipSetDevice(ctx->dev);
    if (int rc = ensure_decoder_repack(ctx)) return rc;
    const int B = ctx->cur_batch, C = ctx->c.max_target_positions, V = ctx->c.vocab_size;
    for (int i = 0; i < n; i++) if (lang_tokens[i] < 0 || lang_tokens[i] >= V) return ctx->fail(NH_ERR_INVALID, "nh_detect_language: token id outside the vocabulary");
    std::vector<int32_t> toks((size_t)B * C, 0);
    for (int b = 0; b < B; b++) toks[(size_t)b * C] = ctx->tk.sot;  // tokens = [[sot]], model.rs:195
    HIPCHK(hipStreamWaitEvent(ctx->sd, ctx->enc_done, 0));
    HIPCHK(hipMemcpyAsync(ctx->ds.tokens, toks.data(), toks.size() * 4, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipMemcpyAsync(ctx->d_lang_tokens, lang_tokens, n * 4, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    decoder_step(ctx, 0);
    logits_from_dxn(ctx, B);
    launch_lang_detect(ctx->logits, V, ctx->d_lang_tokens, n, out_probs ? ctx->d_lang_probs : nullptr, ctx->d_lang_out, B, ctx->sd);
    HIPCHK(hipMemcpyAsync(out_lang, ctx->d_lang_out, B * 4, hipMemcpyDeviceToHost, ctx->sd));
    if (out_probs) HIPCHK(hipMemcpyAsync(out_probs, ctx->d_lang_probs, (size_t)B * n * 4, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    HIPCHK(hipGetLastError());
    ctx->seq_lang.assign(out_lang, out_lang + B);
    return NH_OK;
}

extern "C" int nh_transcribe_batch(nh_ctx *ctx, const float *pcm_dev, const int32_t *n_samples, int64_t stride, int batch,
                                   int32_t *out_tokens, nh_decode_result *results, int max_new_tokens) {
    int rc = nh_logmel_device(ctx, pcm_dev, n_samples, stride, batch);
    if (rc) return rc;
    if ((rc = nh_encode(ctx))) return rc;
    return nh_decode_greedy(ctx, out_tokens, results, max_new_tokens);
}

extern "C" int nh_reset(nh_ctx *ctx) {  // Type::reset_kv_cache (model.rs:485-490)
    if (!ctx) return NH_ERR_INVALID;
    ctx->have_enc = false;
    return NH_OK;
}

extern "C" int nh_synchronize(nh_ctx *ctx) {
    if (!ctx) return NH_ERR_INVALID;
    hipSetDevice(ctx->dev);
    HIPCHK(hipStreamSynchronize(ctx->st));   // sd is the same stream
    return NH_OK;
}

extern "C" int nh_decoder_forward(nh_ctx *ctx, const int32_t *tokens, int T, float *hidden_out) {
    if (!ctx || !tokens || !hidden_out) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_decoder_forward: bad arguments") : NH_ERR_INVALID;
    if (!ctx->have_enc) return ctx->fail(NH_ERR_STATE, "nh_decoder_forward: call nh_encode first");
    const int B = ctx->cur_batch, C = ctx->c.max_target_positions, d = ctx->c.d_model, V = ctx->c.vocab_size;
    if (T < 1 || T > C) return ctx->fail(NH_ERR_INVALID, "nh_decoder_forward: T out of range");
    hipSetDevice(ctx->dev);
    if (int rc = ensure_decoder_repack(ctx)) return rc;
    HIPCHK(hipStreamWaitEvent(ctx->sd, ctx->enc_done, 0));
    std::vector<int32_t> toks((size_t)B * C, 0);
    for (int b = 0; b < B; b++)
        for (int i = 0; i < T; i++) {
            int t = tokens[(size_t)b * T + i];
            if (t < 0 || t >= V) return ctx->fail(NH_ERR_INVALID, "nh_decoder_forward: token id outside the vocabulary");
            toks[(size_t)b * C + i] = t;
        }
    HIPCHK(hipMemcpyAsync(ctx->ds.tokens, toks.data(), toks.size() * 4, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    std::vector<float> row((size_t)B * d);
    for (int pos = 0; pos < T; pos++) {
        decoder_step(ctx, pos);
        HIPCHK(hipMemcpyAsync(row.data(), ctx->dy32, row.size() * 4, hipMemcpyDeviceToHost, ctx->sd));
        HIPCHK(hipStreamSynchronize(ctx->sd));
        for (int b = 0; b < B; b++) memcpy(hidden_out + ((size_t)b * T + pos) * d, row.data() + (size_t)b * d, sizeof(float) * d);
    }
    HIPCHK(hipGetLastError());
    return NH_OK;
}

extern "C" int nh_final_linear(nh_ctx *ctx, const float *x, int rows, float *logits_out) {
    if (!ctx || !x || !logits_out) return ctx ? ctx->fail(NH_ERR_INVALID, "nh_final_linear: bad arguments") : NH_ERR_INVALID;
    if (rows < 1) return ctx->fail(NH_ERR_INVALID, "nh_final_linear: rows must be >= 1");
    hipSetDevice(ctx->dev);
    if (int rc = ensure_decoder_repack(ctx)) return rc;
    const int d = ctx->c.d_model, V = ctx->c.vocab_size;
    std::vector<_Float16> h((size_t)ctx->B * d);
    for (int r0 = 0; r0 < rows; r0 += ctx->B) {  // the workspace holds max_batch rows at a time
        const int nr = rows - r0 < ctx->B ? rows - r0 : ctx->B;
        for (size_t i = 0; i < (size_t)nr * d; i++) h[i] = (_Float16)x[(size_t)r0 * d + i];
        HIPCHK(hipMemcpyAsync(ctx->dxn, h.data(), (size_t)nr * d * 2, hipMemcpyHostToDevice, ctx->sd));
        HIPCHK(hipStreamSynchronize(ctx->sd));
        logits_from_dxn(ctx, nr);
        for (int r = 0; r < nr; r++)
            HIPCHK(hipMemcpyAsync(logits_out + (size_t)(r0 + r) * V, ctx->logits + (size_t)r * ctx->VP, sizeof(float) * V,
                                  hipMemcpyDeviceToHost, ctx->sd));
        HIPCHK(hipStreamSynchronize(ctx->sd));
    }
    HIPCHK(hipGetLastError());
    return NH_OK;
}

extern "C" int nh_apply_rules(nh_ctx *ctx, const float *probs, const int32_t *tokens, int n_tokens, int last_timestamp,
                              float *masked_out, int32_t *argmax_out) {
    if (!ctx || !probs || !tokens || !masked_out || !argmax_out || n_tokens < 1)
        return ctx ? ctx->fail(NH_ERR_INVALID, "nh_apply_rules: bad arguments") : NH_ERR_INVALID;
    if (!ctx->have_tokens) return ctx->fail(NH_ERR_STATE, "nh_apply_rules: call nh_set_tokens first");
    hipSetDevice(ctx->dev);
    const int V = ctx->c.vocab_size;
    float *d_in = ctx->logits, *d_out = ctx->logits + ctx->VP * (ctx->B > 1 ? 1 : 0);
    float *tmp_out = nullptr;
    if (ctx->B == 1) { if (hipMalloc(reinterpret_cast<void **>(&tmp_out), sizeof(float) * V) != hipSuccess) return ctx->fail(NH_ERR_NOMEM, "hipMalloc"); d_out = tmp_out; }
    int32_t *d_tok = ctx->ds.tokens;
    HIPCHK(hipMemcpyAsync(d_in, probs, sizeof(float) * V, hipMemcpyHostToDevice, ctx->sd));
    HIPCHK(hipMemcpyAsync(d_tok, tokens, sizeof(int32_t) * n_tokens, hipMemcpyHostToDevice, ctx->sd));
    launch_rules_only(d_in, d_out, ctx->ds.n_active, d_tok, n_tokens, last_timestamp, ctx->suppress, ctx->tk, V, ctx->sd);
    HIPCHK(hipMemcpyAsync(masked_out, d_out, sizeof(float) * V, hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipMemcpyAsync(argmax_out, ctx->ds.n_active, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->sd));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    if (tmp_out) hipFree(tmp_out);
    HIPCHK(hipGetLastError());
    return NH_OK;
}

// ---- instrumentation -------------------------------------------------------------------------------------
extern "C" int nh_set_profile_gemm(nh_ctx *ctx, int enable) {
    if (!ctx) return NH_ERR_INVALID;
    ctx->profile_gemm = enable != 0;
    return NH_OK;
}

extern "C" int nh_set_option(nh_ctx *ctx, int option, int value) {
    if (!ctx) return NH_ERR_INVALID;
    if (option == NH_OPT_DECODE_GRAPHS) ctx->opt_graphs = value != 0;
    else if (option == NH_OPT_FUSE_DECODE_LAYERNORM) { ctx->opt_fuse_ln = value != 0; drop_graphs(ctx); }
    else if (option == NH_OPT_DECODER_LAYER_LIMIT) {
        if (value < 0 || value > ctx->c.decoder_layers) return ctx->fail(NH_ERR_INVALID, "nh_set_option: layer limit outside [0, decoder_layers]");
        ctx->dec_layer_limit = value; drop_graphs(ctx);
    }
    else if (option == NH_OPT_ABSORBED_XATTN) {
        if (value < 0 || value > 2) return ctx->fail(NH_ERR_INVALID, "nh_set_option: NH_OPT_ABSORBED_XATTN takes 0, 1 or 2");
        if (value == 2 && !xabs_fast_supported(ctx->c.d_model, ctx->c.decoder_attention_heads)) value = 1;   // widths the one-pass kernel is not built for
        hipSetDevice(ctx->dev);
        if (value && !ctx->xabs_u) {
            ctx->xabs_u = dalloc<half_t>(ctx, (size_t)ctx->B * 32 * ctx->c.d_model);
            if (!ctx->xabs_u) return ctx->fail(NH_ERR_NOMEM, "nh_set_option: hipMalloc failed");
        }
        if (value == 2 && !ctx->xabs_z) {
            ctx->xabs_z = dalloc<float>(ctx, (size_t)ctx->B * 4 * ctx->c.decoder_attention_heads * ctx->c.d_model);
            ctx->xabs_ml = dalloc<float>(ctx, (size_t)ctx->B * 4 * 32 * 2);
            if (!ctx->xabs_z || !ctx->xabs_ml) return ctx->fail(NH_ERR_NOMEM, "nh_set_option: hipMalloc failed");
        }
        ctx->opt_absorbed = value; drop_graphs(ctx);
    }
    else return ctx->fail(NH_ERR_INVALID, "nh_set_option: unknown option " + std::to_string(option));
    return NH_OK;
}

extern "C" int nh_get_timings(nh_ctx *ctx, nh_timings *out) {
    if (!ctx || !out) return NH_ERR_INVALID;
    hipSetDevice(ctx->dev);
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->sd));
    nh_timings t = ctx->tm;
    hipEventElapsedTime(&t.mel_ms, ctx->ev[0], ctx->ev[1]);
    hipEventElapsedTime(&t.encoder_ms, ctx->ev[2], ctx->ev[3]);
    hipEventElapsedTime(&t.cross_kv_ms, ctx->ev[3], ctx->ev[4]);
    hipEventElapsedTime(&t.decode_ms, ctx->ev[5], ctx->ev[6]);
    (void)hipGetLastError();  // a phase that has not run yet (a decode pool records no decode interval) leaves its time at 0, not an error behind
    t.gemm_ms = 0.f; t.gemm_launches = 0; t.gemm_flops = ctx->gemm_flops_acc;
    for (size_t i = 0; i + 1 < ctx->gemm_ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->gemm_ev[i], ctx->gemm_ev[i + 1]) == hipSuccess) { t.gemm_ms += ms; t.gemm_launches++; }
    }
    *out = t;
    return NH_OK;
}
