/*
 * qtts_hip.h — C ABI of libqtts_hip.so: the MI355X (gfx950) kernels behind the
 * Qwen3-TTS talker + code-predictor `run_step` hot path.
 *
 * The reference (tsdocode/nano-qwen3tts-vllm) is pure Python and has no FFI of its
 * own; each entry point below cites the reference function it replaces
 * (P = nano-qwen3tts-vllm/ in the reference tree).  All pointers are DEVICE
 * pointers unless stated otherwise, all tensors are dense row-major, `bf16` is
 * the 16-bit brain float stored as uint16_t.  Every function that takes a
 * `stream` (a hipStream_t passed as void*) is asynchronous on it, allocates no
 * device memory, never synchronises, and is therefore legal under
 * hipStreamBeginCapture.  The set-up calls without a stream argument are not:
 * qtts_model_create / qtts_graph_end allocate host memory for their handles,
 * qtts_model_set_workspace clears the workspace's counters with a synchronous
 * hipMemset; call them outside capture.
 *
 * Return value: 0 on success, a negative QTTS_E* code on a rejected argument
 * (nothing was launched), or a positive hipError_t from the launch.
 * qtts_last_error() returns a thread-local message for the last failure.
 */
#ifndef QTTS_HIP_H
#define QTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QTTS_OK 0
#define QTTS_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define QTTS_ESTATE (-2)   /* model handle not fully configured */

#define QTTS_KV_BLOCK 256  /* tokens per KV block (P/engine/sequence.py:16, P/config.py:52,60) */
#define QTTS_HEAD_DIM 128  /* the only head size the released checkpoints use */

typedef void *qtts_stream_t;

int qtts_version(void);
const char *qtts_last_error(void);

/* ---------------------------------------------------------------- weights
 * F.linear weights [N,K] (P/layers/linear.py:25) are re-laid once at load time
 * into MFMA fragment order so that one wave-instruction reads 1 KiB contiguous:
 *   packed[nt][ks][lane][j] = W[16*nt + (lane&15)][32*ks + 8*(lane>>4) + j]
 * nt < N/16, ks < K/32, lane < 64, j < 8.  Requires N % 16 == 0, K % 32 == 0.
 * `gate_up` interleaves two [I,K] matrices so that tile t holds gate rows 8t..8t+7 in its
 * rows 0..7 and up rows 8t..8t+7 in rows 8..15 (I % 8 == 0): every MFMA tile then carries
 * both operands of 8 SwiGLU outputs. */
size_t qtts_packed_weight_bytes(int N, int K);
int qtts_pack_weight(const void *w_rowmajor, void *packed, int N, int K, qtts_stream_t stream);
int qtts_pack_gate_up(const void *gate, const void *up, void *packed, int I, int K, qtts_stream_t stream);

/* ---------------------------------------------------------------- operators */

/* out[M,N] = x[M,K] @ W^T (+bias[N]) (+residual[M,N]); fp32 accumulate, one
 * rounding to bf16 after the bias, a second after the residual add — the
 * rounding points of F.linear followed by `residual + y`
 * (P/layers/linear.py:72-73,147-153; P/models/qwen3_tts_share.py:171,175).
 * If norm_w != NULL the rows of x are RMS-normalised first exactly as
 * P/layers/layernorm.py:17-27 (fp32 statistics, bf16, then * norm_w).
 * ldx/ldo/ldr are row strides in elements. */
int qtts_linear(const void *x, int ldx, const void *w_packed, const void *bias,
                const void *residual, int ldr, const void *norm_w, float eps,
                void *out, int ldo, int M, int N, int K, qtts_stream_t stream);

/* qtts_linear without the norm prologue, with scratch memory for prompts: when K is long
 * (more than 12 x 256) and M > 16 the product is formed as K parts whose fp32 partial sums
 * go through `workspace` (qtts_linear_workspace_bytes(M, N, K) bytes, 0 when not needed)
 * before the same bias/residual epilogue.  Without a sufficient workspace the call
 * computes the same result by the single-pass kernel.  Same call sites as qtts_linear
 * (P/layers/linear.py:147-153 for down_proj over a prompt's rows). */
size_t qtts_linear_workspace_bytes(int M, int N, int K);
int qtts_linear_ws(const void *x, int ldx, const void *w_packed, const void *bias,
                   const void *residual, int ldr, void *out, int ldo, int M, int N, int K,
                   void *workspace, size_t workspace_bytes, qtts_stream_t stream);

/* ---- fp8 (e4m3, OCP) weights: an extension, the reference has bf16 weights only.
 * A weight row is stored as q * scale, q in e4m3, scale the smallest power of two with
 * max|row| / scale <= 448 — so q * scale is a bf16 number and every fp8 entry point below
 * returns, bit for bit, what its bf16 counterpart returns on the dequantised weights
 * bf16(q * scale) (that is how it is tested; against the reference's bf16 weights the
 * difference is the quantisation itself).  Halves the weight bytes a step streams.
 * Packed layout: [N/16][K/64][64 lanes][16 B], lane (g, r) = row 16*tile + r, bytes 0..7
 * the k values 64j + 8g .. +8, bytes 8..15 the k values 64j + 32 + 8g .. +8; N % 16 == 0,
 * K % 64 == 0 (K % 512 == 0 for the GEMMs); scales[N] per packed row (for gate/up: the
 * interleaved row order of qtts_pack_gate_up). */
size_t qtts_packed_weight_fp8_bytes(int N, int K);
int qtts_pack_weight_fp8(const void *w_rowmajor, void *packed, float *scales, int N, int K,
                         qtts_stream_t stream);
int qtts_pack_gate_up_fp8(const void *gate, const void *up, void *packed, float *scales,
                          int I, int K, qtts_stream_t stream);
/* qtts_linear / qtts_linear_ws over fp8 weights (workspace as qtts_linear_ws, may be NULL). */
int qtts_linear_fp8(const void *x, int ldx, const void *w_fp8_packed, const float *w_scales,
                    const void *bias, const void *residual, int ldr, const void *norm_w,
                    float eps, void *out, int ldo, int M, int N, int K, void *workspace,
                    size_t workspace_bytes, qtts_stream_t stream);
/* qtts_linear_swiglu over fp8 weights. */
int qtts_linear_swiglu_fp8(const void *x, int ldx, const void *w_gate_up_fp8_packed,
                           const float *w_scales, const void *norm_w, float eps, void *out,
                           int ldo, int M, int I, int K, qtts_stream_t stream);

/* out[M,I] = silu(x@Wg^T) * (x@Wu^T) with the reference's two roundings
 * (P/models/qwen3_tts_share.py:131-135, P/layers/activation.py:11-13);
 * optional fused RMSNorm of x as above. */
int qtts_linear_swiglu(const void *x, int ldx, const void *w_gate_up_packed,
                       const void *norm_w, float eps, void *out, int ldo,
                       int M, int I, int K, qtts_stream_t stream);

/* The decoder stack's GEMM with its fusions (same arithmetic and rounding points as
 * the calls above, one launch):
 *   gather_ids != NULL : row m of x is x[gather_ids[m]] (an embedding table as input,
 *                        P/models/qwen3_tts_predictor.py:94-95 feeding :106);
 *   norm_w != NULL     : RMSNorm prologue; if xnorm_out != NULL the normalised rows are
 *                        also written there (the post-final-norm hidden state,
 *                        P/models/qwen3_tts_talker.py:61, when the GEMM is the codec head);
 *   swiglu_input != 0  : x is [M, 2K] = [gate | up] and the GEMM consumes
 *                        silu(gate)*up (P/models/qwen3_tts_share.py:131-135), i.e. the
 *                        activation is applied on the way into down_proj.
 * out = bf16(bf16(acc + bias) + residual). */
int qtts_linear_fused(const void *x, int ldx, const int64_t *gather_ids, int gather_rows,
                      const void *w_packed, const void *bias, const void *residual, int ldr,
                      const void *norm_w, float eps, void *xnorm_out, int ldxn, int swiglu_input,
                      void *out, int ldo, int M, int N, int K, qtts_stream_t stream);

/* y = rmsnorm(x) * w  (P/layers/layernorm.py:17-27).  x,y [M,H]. */
int qtts_rmsnorm(const void *x, int ldx, const void *w, float eps, void *y, int ldy,
                 int M, int H, qtts_stream_t stream);

/* Per-head RMSNorm of q and k, rotary embedding and KV-cache store in one pass
 * (P/models/qwen3_tts_share.py:96-103, P/layers/rotary_embedding.py:33-34,
 * P/layers/attention.py:10-30).  qkv is [N, (Hq+2*Hkv)*128] (q heads, then k
 * heads, then v heads); q and k are rewritten in place.  rope_cos/rope_sin are
 * bf16 tables [max_pos, 64] (the two halves of the reference's cos/sin rows are
 * equal).  slot_mapping[i] == -1 skips the store; k_cache/v_cache may be NULL. */
int qtts_qknorm_rope_store(void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w,
                           float eps, const void *rope_cos, const void *rope_sin, int max_pos,
                           const int64_t *positions, const int32_t *slot_mapping,
                           void *k_cache, void *v_cache, int N, int Hq, int Hkv,
                           qtts_stream_t stream);

/* KV-cache scatter alone (P/layers/attention.py:33-54): k,v [N,Hkv*128] rows with
 * row strides ldk/ldv; caches [num_blocks,256,Hkv,128]; slot -1 skipped. */
int qtts_store_kvcache(const void *k, int ldk, const void *v, int ldv, void *k_cache, void *v_cache,
                       const int32_t *slot_mapping, int N, int Hkv, qtts_stream_t stream);

/* Paged attention: row i of q attends to the first context_lens[i] cached tokens
 * of sequence row_seq[i] (row_seq == NULL: sequence i) through
 * block_tables[seq][max_blocks].  Serves both call sites of
 * P/layers/attention.py:88-95: decode (one row per sequence, context_lens =
 * cache_seqlens) and causal prefill (context_lens[i] = position+1 after the
 * rows' K/V were stored).  context_lens[i] == 0 writes zeros.  fp32 softmax.
 * With row_seq != NULL and at least 112 rows (prompts) 16 rows share a workgroup and
 * the two contractions run on MFMA; otherwise one workgroup per (row, KV head).
 * workspace: qtts_attn_workspace_bytes(N, Hq, max_context) bytes of device scratch. */
size_t qtts_attn_workspace_bytes(int N, int Hq, int max_context);
int qtts_paged_attention(const void *q, int ldq, const void *k_cache, const void *v_cache,
                         const int32_t *block_tables, int max_blocks, const int32_t *row_seq,
                         const int32_t *context_lens, int max_context, float scale,
                         void *out, int ldo, int N, int Hq, int Hkv, void *workspace,
                         qtts_stream_t stream);

/* Decode-step attention block in one launch (P/models/qwen3_tts_share.py:96-106 with
 * P/layers/attention.py:73-96 in decode mode): per-head RMSNorm + RoPE of q and k read
 * from the raw qkv rows, the new token's K/V stored at slot_mapping[b], attention of
 * row b over context_lens[b] tokens of block_tables[b] (the newest from registers).
 * One row per sequence.  Rows longer than 128 tokens are split over workgroups whose
 * partial results meet in `workspace` (qtts_attn_decode_workspace_bytes(B, Hq, Hkv,
 * max_context) bytes of device scratch; may be NULL when max_context <= 128); the call
 * clears the few counter bytes it needs, so the scratch may be reused freely between calls. */
size_t qtts_attn_decode_workspace_bytes(int B, int Hq, int Hkv, int max_context);
int qtts_attention_decode_fused(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w,
                                float eps, const void *rope_cos, const void *rope_sin, int max_pos,
                                const int64_t *positions, const int32_t *slot_mapping, void *k_cache,
                                void *v_cache, const int32_t *block_tables, int max_blocks,
                                const int32_t *context_lens, int max_context, float scale, void *out,
                                int ldo, int B, int Hq, int Hkv, void *workspace, qtts_stream_t stream);

/* qtts_attention_decode_fused for callers that cannot allocate (code under hipGraph capture that has no workspace
 * to pass, tight serving loops, harnesses that forbid hipMalloc): the same kernels, same arguments and same results,
 * but the split partials and arrival counters live in a static device array inside the library (about 1.02 MiB on
 * every GPU the library runs on, held until the process ends) instead of a caller-provided workspace.  No
 * counterpart in the reference.  Limits: B * Hq <= 256 and max_context <= 1024, larger calls are rejected with
 * QTTS_EINVAL.  Calls no hipMalloc / hipFree; the first call on a device looks the array's address up (make it
 * outside stream capture).  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams)
 * race — serialise them, or use the workspace form. */
int qtts_attention_decode_fused_static(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w,
                                       float eps, const void *rope_cos, const void *rope_sin, int max_pos,
                                       const int64_t *positions, const int32_t *slot_mapping, void *k_cache,
                                       void *v_cache, const int32_t *block_tables, int max_blocks,
                                       const int32_t *context_lens, int max_context, float scale, void *out,
                                       int ldo, int B, int Hq, int Hkv, qtts_stream_t stream);

/* Greedy pick: out[i] = argmax_j float(logits[i,j]), lowest j on ties (the build's
 * do_sample=False path; P/sampling_params.py:9 carries the flag). */
int qtts_argmax(const void *logits, int ld, int64_t *out, int M, int V, qtts_stream_t stream);

/* Stochastic pick of P/layers/sampler.py:15-24: logits/temperature, keep values >=
 * the top_k-th largest, softmax, one multinomial draw (counter-based RNG keyed
 * by seed, offset and row — not torch's generator stream). V <= 4096. */
int qtts_sample_topk(const void *logits, int ld, const float *temperatures, int top_k,
                     uint64_t seed, uint64_t offset, int64_t *out, int M, int V,
                     qtts_stream_t stream);

/* Both picks behind one launch, for graphs that serve mixed batches: row i is greedy
 * when greedy_flags != NULL && greedy_flags[i] != 0, otherwise drawn as
 * qtts_sample_topk does.  offset_dev (nullable, device) is added to `offset` so a
 * captured graph can advance its RNG counter between replays.  out[i*ldout]. */
int qtts_pick_token(const void *logits, int ld, const float *temperatures,
                    const uint8_t *greedy_flags, int top_k, uint64_t seed, uint64_t offset,
                    const uint64_t *offset_dev, int64_t *out, int ldout, int M, int V,
                    qtts_stream_t stream);

/* out[i,:] = table[ids[i],:] (P/models/qwen3_tts_predictor.py:94-95). */
int qtts_embedding_gather(const void *table, const int64_t *ids, void *out, int ldo,
                          int M, int H, int vocab, qtts_stream_t stream);

/* Per-frame glue kept on the device (P/interface.py:1242-1252,1276-1285):
 * pred_in[b] = [hidden[b]; talker_emb[token0[b]]]                              ([B,2,H])
 * next[b]    = bf16( sum_{c<16} emb_c[code_c(b)] (fp32, c ascending) + text_add_b (fp32) )
 * where code_0 = token0[b], code_c = pred_tokens[(c-1)*ldp + b], and text_add_b is
 * the fp32 row text_add + b*ld_text + frame_idx[b]*H (frame_idx NULL: row 0) — the
 * trailing text hidden or tts pad embedding the API adds each frame.  The 16 ids
 * are also written to codes[b*16 + c] when codes != NULL.  qtts_frame_glue reads
 * and writes 16 bytes at a time: H % 8 == 0, ld_text % 4 == 0, every table, text_add
 * and next 16-byte aligned. */
int qtts_predictor_input(const void *hidden, const void *talker_emb, const int64_t *token0,
                         void *pred_in, int B, int H, int vocab, qtts_stream_t stream);
int qtts_frame_glue(const void *talker_emb, const void *const *pred_emb_tables /* device array of 15 */,
                    const int64_t *token0, const int64_t *pred_tokens, int ldp,
                    const float *text_add, long ld_text, const int32_t *frame_idx,
                    void *next /* [B,H] bf16 */, int64_t *codes /* [B,16] or NULL */,
                    int B, int H, int talker_vocab, int pred_vocab, qtts_stream_t stream);

/* Device-side equivalent of re-running prepare_decode_talker for the next token
 * (P/engine/model_runner/talker_mode_runner.py:116-140) for every active row
 * (context_lens > 0): position+1, context+1, slot from the row's block table,
 * frame_idx+1; *rng_counter += 64 (nullable).  A row whose context has reached
 * min(max_context, context_limits[b]) (context_limits nullable: max_context alone) is
 * frozen — the host sets the limit to the tokens the row's KV blocks were reserved for,
 * so a row stepped past its budget rewrites its own last slot and never walks into a
 * block-table entry that is not its own.  No counterpart in the reference (it rebuilds
 * the lists on the host and has no length guard, P/engine/llm_engine/talker_llm_engine.py:68-83). */
int qtts_advance_decode(int64_t *positions, int32_t *slot_mapping, int32_t *context_lens,
                        const int32_t *block_tables, int max_blocks, int32_t *frame_idx,
                        uint64_t *rng_counter, const int32_t *context_limits, int max_context, int B,
                        qtts_stream_t stream);

/* ---------------------------------------------------------------- decoder stack
 * One handle per model (talker or code predictor): the layer loop of
 * P/models/qwen3_tts_talker.py:53-62 / qwen3_tts_predictor.py:30-44 with the
 * layer body of P/models/qwen3_tts_share.py:79-113,163-176 as five launches. */
typedef struct qtts_model qtts_model_t;

qtts_model_t *qtts_model_create(int hidden, int n_layers, int n_heads, int n_kv_heads,
                                int head_dim, int intermediate, float rms_eps);
void qtts_model_destroy(qtts_model_t *m);
/* wqkv_packed = qtts_pack_weight of [q_proj; k_proj; v_proj] rows, w_gate_up_packed =
 * qtts_pack_gate_up(gate_proj, up_proj). */
int qtts_model_set_layer(qtts_model_t *m, int layer, const void *input_norm_w,
                         const void *wqkv_packed, const void *q_norm_w, const void *k_norm_w,
                         const void *wo_packed, const void *post_norm_w,
                         const void *w_gate_up_packed, const void *w_down_packed);
/* qtts_model_set_layer with the four GEMM weights as fp8 (qtts_pack_weight_fp8 / qtts_pack_gate_up_fp8) and their
 * row scales; hidden and intermediate size must be multiples of 512.  No counterpart in the reference. */
int qtts_model_set_layer_fp8(qtts_model_t *m, int layer, const void *input_norm_w, const void *wqkv_fp8,
                             const float *wqkv_scales, const void *q_norm_w, const void *k_norm_w,
                             const void *wo_fp8, const float *wo_scales, const void *post_norm_w,
                             const void *w_gate_up_fp8, const float *w_gate_up_scales,
                             const void *w_down_fp8, const float *w_down_scales);

int qtts_model_set_final_norm(qtts_model_t *m, const void *w);
/* Measurement hook: from now on every decode step of `m` (up to 16 rows) launches kernel number `which` of layer l with
 * begin[l] / end[l] (hipEvent_t, n_layers of each, created by the caller with timing enabled; the arrays are copied)
 * as the kernel's own start / stop events (hipExtLaunchKernelGGL), so hipEventElapsedTime(begin[l], end[l]) after
 * the stream has drained is that kernel's execution time in place, between the launches it really follows and
 * precedes - the figure a kernel trace reports for it.  which == 0 switches the hook off.  Not for forwards under
 * stream capture.  The reference times whole steps with time.time() around run_model
 * (P/engine/model_runner/talker_mode_runner.py:53,83-84); it has no per-launch figure. */
#define QTTS_LAUNCH_QKV 1        /* input norm + q/k/v projection */
#define QTTS_LAUNCH_O_PROJ 2     /* o_proj + residual */
#define QTTS_LAUNCH_GATE_UP 3    /* post-attention norm + gate/up projection + SwiGLU */
#define QTTS_LAUNCH_DOWN 4       /* down_proj + residual */
#define QTTS_LAUNCH_ATTENTION 5  /* q/k norm + RoPE + KV store + paged decode attention (the fused decode launch) */
#define QTTS_LAUNCH_HEAD 6       /* final norm + output head of qtts_model_forward_logits* (events of index 0) */
int qtts_model_set_layer_events(qtts_model_t *m, int which, void *const *begin, void *const *end);
int qtts_model_set_rope(qtts_model_t *m, const void *cos, const void *sin, int max_pos);
/* kv: [2, n_layers, num_blocks, 256, n_kv_heads, 128] bf16 (P/engine/model_runner/base.py:158-164). */
int qtts_model_set_kv_cache(qtts_model_t *m, void *kv, int num_blocks);
/* Device scratch of a model: activations for max_rows rows (the largest packed prompt batch,
 * max_num_batched_tokens) and the attention partials of the rows that can need them (a decode step takes at most 512
 * rows, the reference's max_num_seqs, P/config.py:45).  set_workspace clears the counters in it (synchronous). */
size_t qtts_model_workspace_bytes(const qtts_model_t *m, int max_rows, int max_context);
int qtts_model_set_workspace(qtts_model_t *m, void *ws, size_t bytes, int max_rows, int max_context);

/* hidden_out[N,hidden] = final_norm(layers(x)).  Rows are tokens; see
 * qtts_paged_attention for row_seq/context_lens (decode: row_seq NULL,
 * context_lens as prepare_decode builds them; prefill: per-token).
 * P/engine/model_runner/talker_mode_runner.py:51-114, predictor_model_runner.py:66-131. */
int qtts_model_forward(qtts_model_t *m, const void *x, const int64_t *positions,
                       const int32_t *slot_mapping, const int32_t *row_seq,
                       const int32_t *context_lens, const int32_t *block_tables, int max_blocks,
                       int max_context, int N, void *hidden_out, qtts_stream_t stream);
/* Same, with the output head folded in: logits_out[N,vocab] = head(final_norm(layers(x)))
 * and hidden_out = final_norm(layers(x)) from the head GEMM's prologue
 * (compute_logits, P/models/qwen3_tts_talker.py:131-136 / qwen3_tts_predictor.py:110-124). */
int qtts_model_forward_logits(qtts_model_t *m, const void *x, const int64_t *positions,
                              const int32_t *slot_mapping, const int32_t *row_seq,
                              const int32_t *context_lens, const int32_t *block_tables, int max_blocks,
                              int max_context, int N, void *hidden_out, const void *head_w_packed,
                              int vocab, void *logits_out, qtts_stream_t stream);
/* A decode step (one row per sequence) whose input rows are rows of a table: row r = x_table[ids[r]],
 * x_table [table_rows, hidden] bf16; ids are device data and clamped to the table.  Up to 16 rows the table rows are
 * read in place by layer 0 (qkv GEMM input, o_proj residual); a step of more than 16 rows needs qkv0_table and
 * copies its x rows into the model's workspace first (one gather launch).
 * qkv0_table (may be NULL up to 16 rows) [table_rows, (Hq+2Hkv)*128] holds, per table row,
 * layer 0's input norm + qkv projection of that row; with it layer 0's qkv GEMM is not launched and its
 * attention reads the table rows (needs max_context <= 32 unless Hq/Hkv == 4).  Both tables are functions of the
 * weights only: for the code predictor x_table[i] = small_to_mtp_projection(codec_embedding[i].weight) and
 * qkv0_table[i] = qkv_proj(input_layernorm(x_table[i])) of layer 0, built once at load with the library's own
 * GEMMs, so a step returns the same bits as gathering and projecting on the fly.
 * positions, slot_mapping, context_lens and block_tables may ALL be NULL: row r is then sequence r with its tokens in KV
 * block r, every row at context max_context (<= 32; position max_context - 1, slot r * 256 + max_context - 1) - the
 * burst of a batch whose sequences were given block r each; the attention launches then read no metadata at all.
 * (get_input_embeddings P/models/qwen3_tts_predictor.py:87-99, projection :58-61,106, layer body
 * P/models/qwen3_tts_share.py:96-103,160-176). */
int qtts_model_forward_logits_gathered(qtts_model_t *m, const void *x_table, const void *qkv0_table,
                                       const int64_t *ids, int table_rows, const int64_t *positions,
                                       const int32_t *slot_mapping, const int32_t *context_lens,
                                       const int32_t *block_tables, int max_blocks, int max_context, int N,
                                       void *hidden_out, const void *head_w_packed, int vocab,
                                       void *logits_out, qtts_stream_t stream);
/* The 2-position prompt of N/2 fresh sequences (the code predictor's burst starts with [talker hidden, first
 * codebook embedding] per sequence, P/workers/predictor_worker.py:128-134 -> predictor_model_runner.py:134-157 in
 * prefill mode): rows (2i, 2i+1) of x [N,hidden] are positions[2i], positions[2i+1] of sequence i, contexts 1
 * and 2, N even and <= 16.  q/k norm, RoPE, KV store (slot_mapping, -1 = skip) and both rows' attention are one
 * launch per layer.  logits_out [N/2, vocab] and hidden_out [N/2, hidden] belong to the second row of each pair
 * (the last position, talker_mode_runner.py:107-110 / compute_logits). */
int qtts_model_forward_logits_pairs(qtts_model_t *m, const void *x, const int64_t *positions,
                                    const int32_t *slot_mapping, int N, void *hidden_out,
                                    const void *head_w_packed, int vocab, void *logits_out,
                                    qtts_stream_t stream);

/* ---------------------------------------------------------------- hipGraph helpers
 * Capture a sequence of the calls above once per batch size and replay it
 * (the role of torch.cuda.CUDAGraph in talker_mode_runner.py:157-192). */
typedef struct qtts_graph qtts_graph_t;
int qtts_graph_begin(qtts_stream_t stream);
qtts_graph_t *qtts_graph_end(qtts_stream_t stream);
int qtts_graph_launch(qtts_graph_t *g, qtts_stream_t stream);
void qtts_graph_destroy(qtts_graph_t *g);

#ifdef __cplusplus
}
#endif
#endif /* QTTS_HIP_H */
