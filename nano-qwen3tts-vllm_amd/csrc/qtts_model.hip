// libqtts_hip - the decoder stack (talker / code predictor): per-layer launch sequences for decode steps, prompts and
// the burst's forms, workspace layout, hipGraph helpers.  C ABI in include/qtts_hip.h.
#include "qtts_common.h"

// ------------------------------------------------------------------ decoder stack
struct LayerW {
    const void *in_norm = nullptr, *wqkv = nullptr, *q_norm = nullptr, *k_norm = nullptr, *wo = nullptr,
               *post_norm = nullptr, *wgu = nullptr, *wdown = nullptr;
    // fp8 weights: scale per packed row of the four GEMM weights (all null: bf16 weights)
    const float *s_qkv = nullptr, *s_o = nullptr, *s_gu = nullptr, *s_down = nullptr;
};

// The decoder stack's GEMMs with bf16 or fp8 weights (wscale != null).
// How a prompt GEMM's activations are laid out (all zero: row-major everywhere)
struct ActLayout {
    int xpacked = 0, opks = 0, nnpks = 0;
    int rpks = 0, opks0 = 0;            // plain epilogue: residual rows / output in fragment order
    float *ssq_out = nullptr;           // plain epilogue: the stored rows' sums of squares per 16 columns
    const float *ssq_in = nullptr;      // with a norm weight: the rows are normalised from these partials (prologue 3)
};

static int stack_linear(const void *x, int ldx, const void *w, const float *wscale, const void *residual, int ldr,
                        const void *norm_w, float eps, void *out, int ldo, int M, int N, int K, float *scratch,
                        size_t scratch_bytes, qtts_stream_t stream, const void *next_norm_w = nullptr,
                        void *next_xn = nullptr, int ldnn = 0, const int64_t *x_ids = nullptr,
                        const int64_t *res_ids = nullptr, int id_rows = 0, ActLayout lay = ActLayout()) {
    LinArgs a;
    a.xpacked = lay.xpacked; a.opks = lay.opks; a.nnpks = lay.nnpks;
    a.next_nw = (const bf16_t *)next_norm_w; a.next_xn = (bf16_t *)next_xn; a.ldnn = ldnn;
    a.res_ids = res_ids; a.res_id_rows = res_ids ? id_rows : 0;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = x_ids; a.id_rows = x_ids ? id_rows : 0; a.wp = (const uint4 *)w; a.wscale = wscale;
    a.bias = nullptr; a.res = (const bf16_t *)residual; a.ldr = ldr; a.nw = (const bf16_t *)norm_w; a.eps = eps;
    a.xnorm_out = nullptr; a.ldxn = 0; a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.tiles_per_wg = 1;
    a.rpks = lay.rpks; a.opks0 = lay.opks0; a.ssq_out = lay.ssq_out; a.ssq_in = lay.ssq_in;
    return launch_linear2(a, norm_w ? (lay.ssq_in ? 3 : 1) : 0, (hipStream_t)stream, false, scratch, scratch_bytes);
}

static int stack_swiglu(const void *x, int ldx, const void *wgu, const float *wscale, const void *norm_w, float eps,
                        void *out, int ldo, int M, int I, int K, qtts_stream_t stream, ActLayout lay = ActLayout()) {
    LinArgs a;
    a.xpacked = lay.xpacked; a.opks = lay.opks; a.nnpks = lay.nnpks;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = nullptr; a.id_rows = 0; a.wp = (const uint4 *)wgu; a.wscale = wscale;
    a.bias = nullptr; a.res = nullptr; a.ldr = 0; a.nw = (const bf16_t *)norm_w; a.eps = eps; a.xnorm_out = nullptr;
    a.ldxn = 0; a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = 2 * I; a.K = K; a.tiles_per_wg = 1;
    a.ssq_in = lay.ssq_in;
    return launch_linear2(a, norm_w ? (lay.ssq_in ? 3 : 1) : 0, (hipStream_t)stream, true);
}

struct qtts_model {
    int hidden, n_layers, n_heads, n_kv_heads, head_dim, inter;
    float eps;
    std::vector<LayerW> layers;
    const void *final_norm = nullptr;
    const void *cosT = nullptr, *sinT = nullptr;
    int max_pos = 0;
    char *kv = nullptr;
    int num_blocks = 0;
    char *ws = nullptr;
    size_t ws_bytes = 0;
    int ws_rows = 0, ws_ctx = 0;
    // measurement hook (qtts_model_set_layer_events): events recorded around one launch of every layer
    int ev_which = 0;
    std::vector<hipEvent_t> ev_begin, ev_end;
};

// Hands the layer's begin / end event to the decode GEMM launcher for launch number `which` of the layer body when
// the hook is set: the launch then goes through hipExtLaunchKernelGGL, which binds the two events to the kernel's
// own start and end (the times a profiler reports for it).
struct LayerEventScope {
    bool on;
    LayerEventScope(const qtts_model *m, int layer, int which, qtts_stream_t)
        : on(m->ev_which == which && which != 0) {
        if (on) g_kernel_events = {m->ev_begin[layer], m->ev_end[layer]};
    }
    ~LayerEventScope() { if (on) g_kernel_events = {nullptr, nullptr}; }    // (a launch that took another kernel)
};

extern "C" qtts_model_t *qtts_model_create(int hidden, int n_layers, int n_heads, int n_kv_heads, int head_dim,
                                           int intermediate, float rms_eps) {
    if (hidden <= 0 || n_layers <= 0 || n_heads <= 0 || n_kv_heads <= 0 || intermediate <= 0 ||
        head_dim != QTTS_HEAD_DIM || n_heads % n_kv_heads != 0 || hidden % 32 != 0 || intermediate % 32 != 0) {
        snprintf(g_err, sizeof(g_err), "model_create: unsupported shape (head_dim must be 128, sizes multiples of 32)");
        return nullptr;
    }
    qtts_model *m = new qtts_model();
    m->hidden = hidden;
    m->n_layers = n_layers;
    m->n_heads = n_heads;
    m->n_kv_heads = n_kv_heads;
    m->head_dim = head_dim;
    m->inter = intermediate;
    m->eps = rms_eps;
    m->layers.resize(n_layers);
    return m;
}

extern "C" void qtts_model_destroy(qtts_model_t *m) { delete m; }

extern "C" int qtts_model_set_layer(qtts_model_t *m, int layer, const void *input_norm_w, const void *wqkv_packed,
                                    const void *q_norm_w, const void *k_norm_w, const void *wo_packed,
                                    const void *post_norm_w, const void *w_gate_up_packed, const void *w_down_packed) {
    QCHECK(m && layer >= 0 && layer < m->n_layers, "model_set_layer: bad layer index");
    QCHECK(input_norm_w && wqkv_packed && q_norm_w && k_norm_w && wo_packed && post_norm_w && w_gate_up_packed &&
               w_down_packed, "model_set_layer: null weight");
    LayerW &L = m->layers[layer];
    L.in_norm = input_norm_w;
    L.wqkv = wqkv_packed;
    L.q_norm = q_norm_w;
    L.k_norm = k_norm_w;
    L.wo = wo_packed;
    L.post_norm = post_norm_w;
    L.wgu = w_gate_up_packed;
    L.wdown = w_down_packed;
    L.s_qkv = L.s_o = L.s_gu = L.s_down = nullptr;
    return QTTS_OK;
}

extern "C" int qtts_model_set_layer_fp8(qtts_model_t *m, int layer, const void *input_norm_w, const void *wqkv_fp8,
                                        const float *wqkv_scales, const void *q_norm_w, const void *k_norm_w,
                                        const void *wo_fp8, const float *wo_scales, const void *post_norm_w,
                                        const void *w_gate_up_fp8, const float *w_gate_up_scales, const void *w_down_fp8,
                                        const float *w_down_scales) {
    QCHECK(wqkv_scales && wo_scales && w_gate_up_scales && w_down_scales, "model_set_layer_fp8: null scales");
    QCHECK(m && m->hidden % 512 == 0 && m->inter % 512 == 0, "model_set_layer_fp8: hidden and intermediate size must be multiples of 512");
    int rc = qtts_model_set_layer(m, layer, input_norm_w, wqkv_fp8, q_norm_w, k_norm_w, wo_fp8, post_norm_w, w_gate_up_fp8,
                                  w_down_fp8);
    if (rc) return rc;
    LayerW &L = m->layers[layer];
    L.s_qkv = wqkv_scales; L.s_o = wo_scales; L.s_gu = w_gate_up_scales; L.s_down = w_down_scales;
    return QTTS_OK;
}

extern "C" int qtts_model_set_layer_events(qtts_model_t *m, int which, void *const *begin, void *const *end) {
    QCHECK(m, "model_set_layer_events: null model");
    QCHECK(which >= 0 && which <= QTTS_LAUNCH_HEAD, "model_set_layer_events: unknown launch %d", which);
    QCHECK(which == 0 || (begin && end), "model_set_layer_events: null event arrays");
    m->ev_which = 0;
    m->ev_begin.clear();
    m->ev_end.clear();
    if (which) {
        for (int l = 0; l < m->n_layers; ++l) {
            QCHECK(begin[l] && end[l], "model_set_layer_events: null event for layer %d", l);
            m->ev_begin.push_back((hipEvent_t)begin[l]);
            m->ev_end.push_back((hipEvent_t)end[l]);
        }
        m->ev_which = which;
    }
    return QTTS_OK;
}

extern "C" int qtts_model_set_final_norm(qtts_model_t *m, const void *w) {
    QCHECK(m && w, "model_set_final_norm: null");
    m->final_norm = w;
    return QTTS_OK;
}

extern "C" int qtts_model_set_rope(qtts_model_t *m, const void *cos, const void *sin, int max_pos) {
    QCHECK(m && cos && sin && max_pos > 0, "model_set_rope: bad arguments");
    m->cosT = cos;
    m->sinT = sin;
    m->max_pos = max_pos;
    return QTTS_OK;
}

extern "C" int qtts_model_set_kv_cache(qtts_model_t *m, void *kv, int num_blocks) {
    QCHECK(m && kv && num_blocks > 0, "model_set_kv_cache: bad arguments");
    m->kv = (char *)kv;
    m->num_blocks = num_blocks;
    return QTTS_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
#define QTTS_MAX_DECODE_ROWS 512   // sequences per decode step: the reference's max_num_seqs (P/config.py:45)

struct WsLayout {
    size_t qkv, attn, act, xa, xb, xn, ssq, parts, parts_bytes, att_ws, att_cnt, total;
};

static WsLayout ws_layout(const qtts_model *m, int rows, int max_context) {
    WsLayout L;
    size_t off = 0;
    const size_t qkvw = (size_t)(m->n_heads + 2 * m->n_kv_heads) * 128;
    L.qkv = off; off += align256((size_t)rows * qkvw * 2);
    const size_t rows16 = ((size_t)rows + 15) & ~(size_t)15;      // fragment-order buffers hold whole 16-row tiles
    L.attn = off; off += align256(rows16 * m->n_heads * 128 * 2);
    L.act = off; off += align256(rows16 * m->inter * 2);
    L.xa = off; off += align256(rows16 * m->hidden * 2);              // (residual stream: fragment order inside a prompt's layers)
    L.xb = off; off += align256(rows16 * m->hidden * 2);
    L.xn = off; off += align256(rows16 * m->hidden * 2);              // normalised rows of a prompt (prefill GEMMs)
    L.ssq = off; off += align256(rows16 * (m->hidden / 16) * sizeof(float));   // residual rows' sums of squares per 16 columns
    L.parts_bytes = linear_rows_scratch_bytes(rows, m->hidden, m->inter);   // K parts of a prompt's down projection
    L.parts = off; off += align256(L.parts_bytes);
    // prefill and decode attention share the partials region; the decode kernel's arrival counters follow it.
    // Partials exist per (row, head, 128-token split), 520 bytes each: sized for the rows that can need them, not
    // for every row of the largest prompt batch.  Decode: at most QTTS_MAX_DECODE_ROWS sequences per step.
    // Prompts: with 1 or 2 query heads per KV head only launches below the MFMA kernel's row threshold use the
    // per-row kernel (and its partials).
    const int dec_rows = std::min(rows, QTTS_MAX_DECODE_ROWS);
#ifdef QTTS_TUNING
    const int pre_rows = rows;                          // the row threshold is a run-time variable in this build
#else
    const int pre_rows = m->n_heads / m->n_kv_heads <= 2 ? std::min(rows, g_mfma_attn_rows) : rows;
#endif
    const size_t pre = qtts_attn_workspace_bytes(pre_rows, m->n_heads, max_context);
    const size_t dec = attn2_partial_bytes(dec_rows, m->n_heads, max_context);
    L.att_ws = off; off += align256(pre > dec ? pre : dec);
    L.att_cnt = off; off += align256((size_t)dec_rows * m->n_kv_heads * sizeof(int));
    L.total = off;
    return L;
}

extern "C" size_t qtts_model_workspace_bytes(const qtts_model_t *m, int max_rows, int max_context) {
    if (!m || max_rows <= 0 || max_context <= 0) return 0;
    return ws_layout(m, max_rows, max_context).total;
}

extern "C" int qtts_model_set_workspace(qtts_model_t *m, void *ws, size_t bytes, int max_rows, int max_context) {
    QCHECK(m && ws && max_rows > 0 && max_context > 0, "model_set_workspace: bad arguments");
    QCHECK(((uintptr_t)ws & 255) == 0, "model_set_workspace: workspace must be 256-byte aligned");
    QCHECK(bytes >= ws_layout(m, max_rows, max_context).total, "model_set_workspace: %zu bytes < required %zu", bytes,
           ws_layout(m, max_rows, max_context).total);
    // the decode attention's arrival counters must start at zero (the kernel leaves them at zero)
    const WsLayout W = ws_layout(m, max_rows, max_context);
    hipError_t e = hipMemset((char *)ws + W.att_cnt, 0, W.total - W.att_cnt);
    if (e != hipSuccess) QFAIL((int)e, "model_set_workspace: memset failed: %s", hipGetErrorString(e));
    m->ws = (char *)ws;
    m->ws_bytes = bytes;
    m->ws_rows = max_rows;
    m->ws_ctx = max_context;
    return QTTS_OK;
}

static int model_forward_impl(qtts_model_t *m, const void *x, const int64_t *positions, const int32_t *slot_mapping,
                              const int32_t *row_seq, const int32_t *context_lens, const int32_t *block_tables,
                              int max_blocks, int max_context, int N, void *hidden_out, const void *head_w, int vocab,
                              void *logits_out, qtts_stream_t stream, const int64_t *x_ids = nullptr, int x_id_rows = 0,
                              const void *qkv0_rows = nullptr, bool pairs = false) {
    QCHECK(m, "model_forward: null model");
    // x_ids: x is a table of x_id_rows rows and row r of the step is x[x_ids[r]] (read in place by layer 0's qkv GEMM
    // and o_proj residual); qkv0_rows: a table of the same rows after layer 0's input norm and qkv GEMM, which then
    // is not launched
    QCHECK(!x_ids || (x_id_rows > 0 && !row_seq && (N <= 16 || qkv0_rows)), "model_forward: gathered input rows are for decode steps (beyond 16 rows: with a layer-0 qkv table)");
    QCHECK(!qkv0_rows || x_ids, "model_forward: a layer-0 qkv table needs row ids");
    if (!m->final_norm || !m->cosT || !m->kv || !m->ws) QFAIL(QTTS_ESTATE, "model_forward: model not fully configured");
    for (const LayerW &L : m->layers)
        if (!L.wqkv) QFAIL(QTTS_ESTATE, "model_forward: layer weights missing");
    // uniform decode rows: no metadata arrays (row r = sequence r in KV block r at context max_context), see
    // attention_decode_fused_impl
    const bool uniform = !pairs && !row_seq && !positions && !slot_mapping && !context_lens && !block_tables;
    QCHECK(x && hidden_out && (uniform || (positions && (pairs || (context_lens && block_tables)))), "model_forward: null pointer");
    QCHECK(!uniform || N <= m->num_blocks, "model_forward: rows without metadata need a KV block each (N=%d)", N);
    // pairs: rows (2i, 2i+1) are the two positions of a fresh sequence i; the head (required) runs on the second rows
    QCHECK(!pairs || (N % 2 == 0 && N <= 16 && !row_seq && !x_ids && head_w), "model_forward: bad 2-position prompt arguments");
    QCHECK(N > 0 && N <= m->ws_rows, "model_forward: N=%d exceeds workspace rows %d", N, m->ws_rows);
    QCHECK(row_seq || N <= QTTS_MAX_DECODE_ROWS, "model_forward: a decode step takes at most %d rows (got %d)", QTTS_MAX_DECODE_ROWS, N);
    QCHECK(max_context > 0 && max_context <= m->ws_ctx, "model_forward: max_context=%d exceeds workspace %d", max_context, m->ws_ctx);
    QCHECK(pairs || uniform || (max_blocks > 0 && max_context <= max_blocks * QTTS_KV_BLOCK), "model_forward: max_context exceeds block table");
    const WsLayout W = ws_layout(m, m->ws_rows, m->ws_ctx);
    bf16_t *qkv = (bf16_t *)(m->ws + W.qkv);
    bf16_t *attn = (bf16_t *)(m->ws + W.attn);
    bf16_t *act = (bf16_t *)(m->ws + W.act);
    bf16_t *xa = (bf16_t *)(m->ws + W.xa);
    bf16_t *xb = (bf16_t *)(m->ws + W.xb);
    bf16_t *xn = (bf16_t *)(m->ws + W.xn);
    void *att_ws = m->ws + W.att_ws;
    const int H = m->hidden, Hq = m->n_heads, Hkv = m->n_kv_heads;
    // More than one 16-row tile (prefill): the rows are normalised by their own launch and the GEMMs run without
    // the norm prologue, whose registers would leave room for one workgroup per CU only; up to 16 rows the norm
    // rides in the GEMM prologue.
    const bool split_norm = N > 16;
    const int qkvw = (Hq + 2 * Hkv) * 128;
    // A prompt whose four GEMMs per layer all run as the row-stationary prompt kernel hands the activations between
    // the kernels of a layer (normalised rows, attention output, SwiGLU output) over in fragment order: that kernel
    // then reads one contiguous KiB per request.  The residual stream, qkv and everything returned stay row-major.
    const bool frag = split_norm && !pairs &&
                      linear_rows_covers(N, qkvw, H, false, false, false, 0) &&
                      linear_rows_covers(N, H, Hq * 128, false, false, false, 0) &&
                      linear_rows_covers(N, 2 * m->inter, H, true, false, false, 0) &&
                      linear_rows_covers(N, H, m->inter, false, false, true, W.parts_bytes);
    const int pkH = frag ? H / 32 : 0, pkA = frag ? Hq * 128 / 32 : 0, pkI = frag ? m->inter / 32 : 0;
    ActLayout lay_in;                                  // a GEMM reading fragment-order rows
    lay_in.xpacked = frag;
    // Such a prompt also keeps its residual stream in fragment order from the first o_proj on, and the GEMMs that
    // finish residual rows (o_proj, a single-pass down_proj) leave every row's sum of squares per 16 columns behind:
    // the GEMM that consumes the rows normalised (gate/up, the next layer's qkv, the head) then computes the row
    // scale from those partials and normalises in registers, and the RMSNorm launch between the two is gone.
    // Up to 64 rows at a hidden size of 2048 (beyond that the four-row-tile workgroups, whose registers leave no room
    // for the normalisation, stream the weights faster than two-row-tile ones save launches: a 200-token prompt 2.89
    // against 3.10 ms) and up to 256 rows at 1024.
    const bool n3 = frag && g_rows_norm && N <= g_rows_norm_max && linear_rows_norm_covers(H) && N <= (H > 1024 ? 64 : 256);
    float *ssq = (float *)(m->ws + W.ssq);
    const float scale = 1.0f / sqrtf((float)m->head_dim);
    const size_t layer_kv = (size_t)m->num_blocks * QTTS_KV_BLOCK * Hkv * 128 * 2;   // bytes per layer per K or V
    const bf16_t *cur = (const bf16_t *)x;
    bool cur_frag = false, cur_ssq = false;            // `cur` in fragment order / its sums of squares are in `ssq`
    int rc;
    // A step of more than 16 rows whose input rows come from a table (x_ids; with layer 0's qkv rows from their table):
    // the prompt GEMM kernels take no row ids, so the rows are gathered once, into the buffer of the normalised rows
    // (free until the first post-attention norm, by when layer 0's o_proj has read them as its residual rows), and
    // the ids are used by layer 0's attention only.
    const int64_t *res_ids0 = x_ids;
    if (x_ids && split_norm) {
        if ((rc = qtts_embedding_gather(x, x_ids, xn, H, N, H, x_id_rows, stream))) return rc;
        cur = xn;
        res_ids0 = nullptr;
    }
    const bool decode = row_seq == nullptr;           // one row per sequence: fused norm/RoPE/store/attention
    // Decode steps of 9..16 rows: every lane of a GEMM's activation requests carries a row of its own (no lanes to
    // share the prologue with, as up to 8 rows), so the two activations a layer produces for a plain prologue -
    // attention output for o_proj, SwiGLU output for down_proj - are handed over in fragment order: one contiguous
    // KiB per request instead of 64 bytes from each of 16 rows.
    // (The 2-position prompts of up to 8 fresh sequences are such a step too, and so is a prompt of 9..16 tokens.)
    const bool dfrag = N > 8 && N <= 16 && m->ws_rows >= 16;
    ActLayout lay_dx, lay_dgu;
    lay_dx.xpacked = dfrag;
    lay_dgu.opks = dfrag ? m->inter / 32 : 0;
    bool xn_ready = false, final_ready = false;       // a norm already applied by the producing launch
    for (int l = 0; l < m->n_layers; ++l) {
        const LayerW &L = m->layers[l];
        char *kc = m->kv + (size_t)l * layer_kv;
        char *vc = m->kv + ((size_t)m->n_layers + l) * layer_kv;
        if (split_norm && l == 0 && qkv0_rows) {
            // layer 0's normalised, projected rows come from the table
        } else if (split_norm) {
            if (!xn_ready && cur_ssq) {
                // the rows are normalised inside the GEMM
                ActLayout lay = lay_in;
                lay.ssq_in = ssq;
                if ((rc = stack_linear(cur, H, L.wqkv, L.s_qkv, nullptr, 0, L.in_norm, m->eps, qkv, qkvw, N, qkvw, H, nullptr, 0, stream,
                                       nullptr, nullptr, 0, nullptr, nullptr, 0, lay)))
                    return rc;
            } else {
                if (!xn_ready && (rc = rmsnorm_impl(cur, H, L.in_norm, m->eps, xn, H, N, H, pkH, stream, cur_frag ? pkH : 0))) return rc;
                if ((rc = stack_linear(xn, H, L.wqkv, L.s_qkv, nullptr, 0, nullptr, 0.f, qkv, qkvw, N, qkvw, H, nullptr, 0, stream,
                                       nullptr, nullptr, 0, nullptr, nullptr, 0, lay_in)))
                    return rc;
            }
            xn_ready = false;
        } else if (l == 0 && qkv0_rows) {
            // layer 0's normalised, projected rows come from the table
        } else {
            LayerEventScope ev(m, l, QTTS_LAUNCH_QKV, stream);
            if ((rc = stack_linear(cur, H, L.wqkv, L.s_qkv, nullptr, 0, L.in_norm, m->eps, qkv, qkvw, N, qkvw, H, nullptr, 0,
                                   stream, nullptr, nullptr, 0, l == 0 ? x_ids : nullptr, nullptr, x_id_rows))) return rc;
        }
        const bool tab0 = l == 0 && qkv0_rows;
        if (pairs) {
            if ((rc = attention_pairs_impl(qkv, qkvw, L.q_norm, L.k_norm, m->eps, m->cosT, m->sinT, m->max_pos, positions,
                                           slot_mapping, kc, vc, scale, attn, Hq * 128, N, Hq, Hkv, (hipStream_t)stream,
                                           dfrag ? Hq * 128 / 32 : 0)))
                return rc;
        } else if (decode) {
            LayerEventScope ev(m, l, QTTS_LAUNCH_ATTENTION, stream);
            if ((rc = attention_decode_fused_impl(tab0 ? qkv0_rows : qkv, qkvw, L.q_norm, L.k_norm, m->eps, m->cosT, m->sinT,
                                                  m->max_pos, positions, slot_mapping, kc, vc, block_tables, max_blocks,
                                                  context_lens, max_context, scale, attn, Hq * 128, N, Hq, Hkv, att_ws,
                                                  (int *)(m->ws + W.att_cnt), (hipStream_t)stream,
                                                  tab0 ? x_ids : nullptr, x_id_rows, dfrag ? Hq * 128 / 32 : pkA))) return rc;
        } else {
            if ((rc = qtts_qknorm_rope_store(qkv, qkvw, L.q_norm, L.k_norm, m->eps, m->cosT, m->sinT, m->max_pos,
                                             positions, slot_mapping, kc, vc, N, Hq, Hkv, stream))) return rc;
            if ((rc = paged_attention_impl(qkv, qkvw, kc, vc, block_tables, max_blocks, row_seq, context_lens,
                                           max_context, scale, attn, Hq * 128, N, Hq, Hkv, att_ws, stream,
                                           dfrag ? Hq * 128 / 32 : pkA))) return rc;
        }
        {
            LayerEventScope ev(m, l, QTTS_LAUNCH_O_PROJ, stream);
            ActLayout lay = split_norm ? lay_in : lay_dx;
            if (n3) {
                lay.rpks = cur_frag ? pkH : 0;
                lay.opks0 = pkH;
                lay.ssq_out = ssq;
            }
            if ((rc = stack_linear(attn, Hq * 128, L.wo, L.s_o, cur, H, nullptr, 0.f, xa, H, N, H, Hq * 128, nullptr, 0, stream,
                                   nullptr, nullptr, 0, nullptr, l == 0 ? res_ids0 : nullptr, x_id_rows, lay)))
                return rc;
        }
        if (split_norm) {
            ActLayout lay_gu = lay_in;
            lay_gu.opks = pkI;
            if (n3) {
                lay_gu.ssq_in = ssq;
                if ((rc = stack_swiglu(xa, H, L.wgu, L.s_gu, L.post_norm, m->eps, act, m->inter, N, m->inter, H, stream, lay_gu))) return rc;
            } else {
                if ((rc = rmsnorm_impl(xa, H, L.post_norm, m->eps, xn, H, N, H, pkH, stream))) return rc;
                if ((rc = stack_swiglu(xn, H, L.wgu, L.s_gu, nullptr, 0.f, act, m->inter, N, m->inter, H, stream, lay_gu))) return rc;
            }
        } else {
            LayerEventScope ev(m, l, QTTS_LAUNCH_GATE_UP, stream);
            if ((rc = stack_swiglu(xa, H, L.wgu, L.s_gu, L.post_norm, m->eps, act, m->inter, N, m->inter, H, stream, lay_dgu))) return rc;
        }
        // a prompt's K-split down_proj also normalises the finished rows for whoever reads them next: the next
        // layer's input norm, or the final norm (the returned hidden state)
        const bool last = l + 1 == m->n_layers;
        const bool fuse_next = split_norm && linear_rows_is_ksplit(N, H, m->inter, W.parts_bytes);
        LayerEventScope ev_down(m, l, QTTS_LAUNCH_DOWN, stream);
        ActLayout lay_down;
        lay_down.xpacked = split_norm ? frag : dfrag;
        lay_down.nnpks = fuse_next && !last ? pkH : 0;   // the next layer's normalised rows; the returned hidden state stays row-major
        if (n3) {
            lay_down.rpks = lay_down.opks0 = pkH;
            lay_down.ssq_out = fuse_next ? nullptr : ssq;
        }
        if ((rc = stack_linear(act, m->inter, L.wdown, L.s_down, xa, H, nullptr, fuse_next ? m->eps : 0.f, xb, H, N, H, m->inter,
                               (float *)(m->ws + W.parts), W.parts_bytes, stream,
                               fuse_next ? (last ? m->final_norm : m->layers[l + 1].in_norm) : nullptr,
                               fuse_next ? (last ? (bf16_t *)hidden_out : xn) : nullptr, H, nullptr, nullptr, 0, lay_down))) return rc;
        if (fuse_next) (last ? final_ready : xn_ready) = true;
        // Fixed roles: xa = post-attention residual stream, xb = layer output.  The next layer reads xb
        // (qkv GEMM, o_proj residual) before its down_proj overwrites it, all stream-ordered.
        cur = xb;
        cur_frag = n3;
        cur_ssq = n3 && !fuse_next;
    }
    if (head_w) {
        if (split_norm) {
            if (!final_ready && cur_ssq && vocab % 16 == 0) {
                // the final norm inside the head GEMM; its first column block writes the normalised rows (the hidden state)
                LinArgs a;
                a.xpacked = 1; a.ssq_in = ssq;
                a.x = cur; a.ldx = H; a.ids = nullptr; a.id_rows = 0; a.wp = (const uint4 *)head_w; a.bias = nullptr; a.res = nullptr;
                a.ldr = 0; a.nw = (const bf16_t *)m->final_norm; a.eps = m->eps; a.xnorm_out = (bf16_t *)hidden_out; a.ldxn = H;
                a.out = (bf16_t *)logits_out; a.ldo = vocab; a.M = N; a.N = vocab; a.K = H; a.tiles_per_wg = 1;
                return launch_linear2(a, 3, (hipStream_t)stream);
            }
            if (!final_ready && (rc = rmsnorm_impl(cur, H, m->final_norm, m->eps, hidden_out, H, N, H, 0, stream, cur_frag ? pkH : 0))) return rc;
            return qtts_linear_fused(hidden_out, H, nullptr, 0, head_w, nullptr, nullptr, 0, nullptr, 0.f, nullptr, 0, 0,
                                     logits_out, vocab, N, vocab, H, stream);
        }
        if (pairs)      // the second row of every pair: a strided view; hidden_out gets those N/2 rows
            return qtts_linear_fused(cur + H, 2 * H, nullptr, 0, head_w, nullptr, nullptr, 0, m->final_norm, m->eps, hidden_out, H,
                                     0, logits_out, vocab, N / 2, vocab, H, stream);
        // final RMSNorm folded into the head GEMM's prologue; the normalised rows are the returned hidden state
        LayerEventScope ev(m, 0, QTTS_LAUNCH_HEAD, stream);
        if ((rc = qtts_linear_fused(cur, H, nullptr, 0, head_w, nullptr, nullptr, 0, m->final_norm, m->eps, hidden_out, H, 0,
                                    logits_out, vocab, N, vocab, H, stream))) return rc;
        return QTTS_OK;
    }
    if (!final_ready && (rc = rmsnorm_impl(cur, H, m->final_norm, m->eps, hidden_out, H, N, H, 0, stream, cur_frag ? pkH : 0))) return rc;
    return QTTS_OK;
}

extern "C" int qtts_model_forward(qtts_model_t *m, const void *x, const int64_t *positions, const int32_t *slot_mapping,
                                  const int32_t *row_seq, const int32_t *context_lens, const int32_t *block_tables,
                                  int max_blocks, int max_context, int N, void *hidden_out, qtts_stream_t stream) {
    return model_forward_impl(m, x, positions, slot_mapping, row_seq, context_lens, block_tables, max_blocks, max_context,
                              N, hidden_out, nullptr, 0, nullptr, stream);
}

extern "C" int qtts_model_forward_logits(qtts_model_t *m, const void *x, const int64_t *positions,
                                         const int32_t *slot_mapping, const int32_t *row_seq,
                                         const int32_t *context_lens, const int32_t *block_tables, int max_blocks,
                                         int max_context, int N, void *hidden_out, const void *head_w_packed, int vocab,
                                         void *logits_out, qtts_stream_t stream) {
    QCHECK(head_w_packed && logits_out && vocab > 0 && vocab % 16 == 0, "model_forward_logits: bad head arguments");
    return model_forward_impl(m, x, positions, slot_mapping, row_seq, context_lens, block_tables, max_blocks, max_context,
                              N, hidden_out, head_w_packed, vocab, logits_out, stream);
}

extern "C" int qtts_model_forward_logits_gathered(qtts_model_t *m, const void *x_table, const void *qkv0_table,
                                                  const int64_t *ids, int table_rows, const int64_t *positions,
                                                  const int32_t *slot_mapping, const int32_t *context_lens,
                                                  const int32_t *block_tables, int max_blocks, int max_context, int N,
                                                  void *hidden_out, const void *head_w_packed, int vocab,
                                                  void *logits_out, qtts_stream_t stream) {
    QCHECK(head_w_packed && logits_out && vocab > 0 && vocab % 16 == 0, "model_forward_logits_gathered: bad head arguments");
    QCHECK(x_table && ids && table_rows > 0, "model_forward_logits_gathered: null table or ids");
    return model_forward_impl(m, x_table, positions, slot_mapping, nullptr, context_lens, block_tables, max_blocks, max_context,
                              N, hidden_out, head_w_packed, vocab, logits_out, stream, ids, table_rows, qkv0_table);
}

extern "C" int qtts_model_forward_logits_pairs(qtts_model_t *m, const void *x, const int64_t *positions,
                                               const int32_t *slot_mapping, int N, void *hidden_out,
                                               const void *head_w_packed, int vocab, void *logits_out,
                                               qtts_stream_t stream) {
    QCHECK(head_w_packed && logits_out && vocab > 0 && vocab % 16 == 0, "model_forward_logits_pairs: bad head arguments");
    return model_forward_impl(m, x, positions, slot_mapping, nullptr, nullptr, nullptr, 0, 2, N, hidden_out, head_w_packed,
                              vocab, logits_out, stream, nullptr, 0, nullptr, true);
}

#ifdef QTTS_TUNING
#include "qtts_tuning.h"
// ------------------------------------------------------------------ grid barrier probe (tuning experiments only)
// `iters` device-wide barriers inside one resident kernel.  mode 0: every workgroup adds to one counter and polls
// it.  mode 1: one arrival counter, the last arriver publishes the round number to per-group flags (256 B apart)
// and a workgroup polls its group's flag only.  mode 2: arrival through per-group counters, the group's last
// arriver adds to the top counter, flags as in mode 1.  Every wait is bounded: a workgroup that gives up raises
// the error word and all leave.
#define GB_STRIDE 64      // ints: one 256 B line per counter / flag
__device__ __forceinline__ int gb_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gb_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int gb_add(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(512) void grid_barrier_probe_kernel(int *mem, int iters, int mode, int gsize, int sleep) {
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int ngroups = (nwg + gsize - 1) / gsize, grp = wg / gsize;
    const int gmembers = min(gsize, nwg - grp * gsize);
    int *err = mem, *top = mem + GB_STRIDE, *flags = mem + 2 * GB_STRIDE, *gcnt = mem + (2 + 64) * GB_STRIDE;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int round = it + 1;
            const int *watch;
            int want;
            if (mode == 0) {
                gb_add(top, 1);
                watch = top;
                want = round * nwg;
            } else {
                bool last;
                if (mode == 1) {
                    last = gb_add(top, 1) == round * nwg - 1;
                } else {
                    last = false;
                    if (gb_add(gcnt + grp * GB_STRIDE, 1) == round * gmembers - 1)
                        last = gb_add(top, 1) == round * ngroups - 1;
                }
                if (last)
                    for (int g = 0; g < ngroups; ++g) gb_st(flags + g * GB_STRIDE, round);
                watch = flags + grp * GB_STRIDE;
                want = round;
            }
            int spins = 0;
            while (gb_ld(watch) < want) {
                if (++spins > 2000000) {
                    gb_st(err, 1);
                    break;
                }
                if (sleep) __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (gb_ld(err)) return;
    }
}

extern "C" int qtts_debug_grid_barrier(int *mem, int iters, int nwg, int threads, int mode, int gsize, int sleep,
                                       qtts_stream_t stream) {
    // mem: (2 + 64 + 64) * 256 bytes
    QCHECK(mem && iters > 0 && nwg > 0 && nwg <= 256 && threads > 0 && threads <= 512 && gsize > 0 &&
               (nwg + gsize - 1) / gsize <= 64, "grid barrier probe: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(mem, 0, (size_t)(2 + 64 + 64) * GB_STRIDE * sizeof(int), s);
    if (e != hipSuccess) QFAIL((int)e, "grid barrier probe: memset failed");
    hipLaunchKernelGGL(grid_barrier_probe_kernel, dim3(nwg), dim3(threads), 0, s, mem, iters, mode, gsize, sleep);
    QLAUNCH();
    return QTTS_OK;
}

#endif  // QTTS_TUNING

// ------------------------------------------------------------------ hipGraph helpers
struct qtts_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

extern "C" int qtts_graph_begin(qtts_stream_t stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) QFAIL((int)e, "graph_begin: %s", hipGetErrorString(e));
    return QTTS_OK;
}

extern "C" qtts_graph_t *qtts_graph_end(qtts_stream_t stream) {
    qtts_graph *g = new qtts_graph();
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g->graph);
    if (e != hipSuccess || !g->graph) {
        snprintf(g_err, sizeof(g_err), "graph_end: %s", hipGetErrorString(e));
        delete g;
        return nullptr;
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "graph_instantiate: %s", hipGetErrorString(e));
        (void)hipGraphDestroy(g->graph);
        delete g;
        return nullptr;
    }
    return g;
}

extern "C" int qtts_graph_launch(qtts_graph_t *g, qtts_stream_t stream) {
    QCHECK(g && g->exec, "graph_launch: null graph");
    hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
    if (e != hipSuccess) QFAIL((int)e, "graph_launch: %s", hipGetErrorString(e));
    return QTTS_OK;
}

extern "C" void qtts_graph_destroy(qtts_graph_t *g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
