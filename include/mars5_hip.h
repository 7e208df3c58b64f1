/*
 * mars5_hip.h  --  C ABI of libmars5_hip.so: hand-written HIP kernels (gfx950 / MI355X)
 * for the MARS5-TTS hot path (AR decode loop + multinomial-DDPM NAR refinement).
 *
 * Conventions: plain pointers + explicit sizes/strides, no C++ or torch types;
 * every function returns an int status (M5_OK = 0, negative = error, never throws);
 * the CALLER owns every buffer (weights, KV cache, workspaces); all pointers are DEVICE
 * pointers unless stated; `stream` is a hipStream_t passed as void*; nothing here
 * synchronises the host.  Strides are in ELEMENTS unless stated.  The reference is pure
 * Python/PyTorch (no FFI of its own), so each entry point cites the reference Python
 * function whose device arithmetic it replaces (paths relative to the reference root).
 *
 * dtype codes select the GEMM/attention OPERAND type; accumulation is always fp32 and the
 * residual stream is always fp32:  M5_F32 = exact-fp32 parity mode (reference CPU/NAR
 * numerics), M5_F16 = reference GPU autocast numerics (ar_generate.py:67), M5_BF16 =
 * BASELINE.json config.  M5_F32X3 (m5_gemm, m5_attention): fp32 operands in memory exactly as for
 * M5_F32, products on the f16 matrix pipe as three split-operand terms (x = hi + lo 2^-11):
 * operand error 2^-22 relative, i.e. fp32-grade results at several times the fp32 matrix
 * rate, NOT bitwise an fmaf chain; |A| < 4094, |W| < 255 or the result is inf / NaN
 * (csrc/gemm.hip, "X3").
 */
#ifndef MARS5_HIP_H
#define MARS5_HIP_H
#include <stdint.h>

/* The library is built with -fvisibility=hidden: only the entry points below are exported. */
#define M5_API __attribute__((visibility("default")))
#ifdef __cplusplus
extern "C" {
#endif

#define M5_OK 0
#define M5_ERR_ARG (-1)
#define M5_ERR_LAUNCH (-2)
#define M5_ERR_UNSUPPORTED (-3)

#define M5_F32 0
#define M5_F16 1
#define M5_BF16 2
#define M5_F32X3 3

M5_API int m5_version(void);                 /* ABI version, currently 1 */
M5_API const char* m5_build_info(void);      /* "gfx950 ..." */

/* ------------------------------------------------------------------------------------
 * GEMM  C[M,N] = A[M,K] . W[N,K]^T (+bias)   -- both operands K-contiguous (nn.Linear
 * layout), MFMA 16x16x32 (f16/bf16) or 16x16x4 (f32).  Replaces every nn.Linear on the
 * path at M > 1: nn_future.py:241,274,298,398 (AR prefill), model.py:61-65,179-203,339,342
 * (speaker encoders, NAR encoder/decoder, heads).  K must be a multiple of 64.
 * ------------------------------------------------------------------------------------ */
#define M5_EPI_F32 0          /* C fp32 = acc + bias                                      */
#define M5_EPI_DT 1           /* C dtype = acc + bias                                     */
#define M5_EPI_RESIDUAL 2     /* C fp32 += acc + bias   (x = x + linear(...))             */
#define M5_EPI_SWIGLU 3       /* W rows interleaved (W_i, V_i): C dtype[M, N/2] = silu(c_2i) * c_2i+1
                                 (nn_future.py:21-29, :297-298)                           */
#define M5_EPI_QKV 4          /* scatter columns into head-major Q / K / V^T (M5QkvScatter) */
#define M5_EPI_SILU_DT 5      /* C dtype = silu(acc + bias)   (timestep MLP, model.py:206-215) */

typedef struct {
    void* q;                  /* [b][h][s][hd]  or NULL when the columns hold no Q section */
    void* k;                  /* [b][h][s][hd]  or NULL                                    */
    void* vt;                 /* [b][h][hd][s]  (V transposed: s contiguous) or NULL       */
    int32_t rows_per_batch;   /* row m -> b = m / rows_per_batch, s = m % rows_per_batch  */
    int32_t n_heads, head_dim;
    int64_t q_bs, q_hs, q_rs; /* element strides: batch, head, row                        */
    int64_t k_bs, k_hs, k_rs;
    int64_t vt_bs, vt_hs, vt_ds; /* batch, head, d-row (s contiguous)                     */
} M5QkvScatter;               /* columns are laid out [present sections in q,k,v order] x (n_heads*head_dim) */

M5_API int m5_gemm(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
            void* C, int64_t ldc, int M, int N, int K, int epi, const M5QkvScatter* sc,
            int batch, int64_t sA, int64_t sW, int64_t sC, int64_t sBias, void* stream);

/* LayerNorm rows (fp32 in): y = (x-mean)*rsqrt(var+eps)*gamma+beta, out fp32 or dtype.
 * n_affine > 1 applies several (gamma,beta) sets to the same normalised row (the 8 NAR
 * heads, model.py:236-242,342).  D % 64 == 0, D <= 2048.  nn.LayerNorm on the path. */
M5_API int m5_layernorm(int out_dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                 void* y, int64_t ldy, int M, int D, int n_affine, int64_t affine_stride, int64_t y_affine_stride,
                 void* stream);

/* RMSNorm rows (nn_future.py:301-312): y = dtype((x * rsqrt(mean(x^2)+eps)) * w). */
M5_API int m5_rmsnorm(int out_dtype, const float* x, int64_t ldx, const float* w, float eps, void* y, int64_t ldy,
               int M, int D, void* stream);

/* ------------------------------------------------------------------------------------
 * Flash attention over head-major Q/K and transposed V, head_dim 64, online softmax,
 * never materialises scores.  F.scaled_dot_product_attention at nn_future.py:272 (AR
 * prefill, causal) and inside nn.MultiheadAttention (model.py:61-67,179-203: key-padding
 * masks).  O is row-major [b][s][h*64+d] (ready to be the next GEMM's A operand).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const void* q; int64_t q_bs, q_hs, q_rs;
    const void* k; int64_t k_bs, k_hs, k_rs;
    const void* vt; int64_t vt_bs, vt_hs, vt_ds;
    void* o; int64_t o_bs, o_rs;
    int32_t B, H, Sq, Sk;
    const int32_t* key_len;     /* per-batch valid key count (device) or NULL = Sk          */
    int32_t causal;             /* 1: key j visible to query i iff j <= i                    */
    float scale;                /* 1/sqrt(head_dim)                                          */
    const int32_t* kv_index;    /* optional device int: K base += (*kv_index) * kv_index_stride_k,   */
    int64_t kv_index_stride_k;  /* V^T base += (*kv_index) * kv_index_stride_v (selects the          */
    int64_t kv_index_stride_v;  /* pre-computed cross-attention memory of the current DDPM step)     */
    const int32_t* q_len;       /* per-batch valid query count (device) or NULL = Sq: query blocks past it are not computed
                                   (sequences of different lengths in one padded layout; their rows of O stay untouched) */
} M5AttnArgs;
M5_API int m5_attention(int dtype, const M5AttnArgs* a, void* stream);

/* out[r][:] = table[idx[r]][:] * 1 + alpha * pe[pos[r]][:] + add[add_idx[r]][:]
 * (pe / add optional).  nn.Embedding + SinePositionalEmbedding (nn_future.py:78-83) +
 * timestep-embedding add (model.py:329,337); also the AR prefill embedding (model.py:106,129). */
M5_API int m5_gather_rows(float* out, int64_t ldo, int R, int D, const float* table, const int64_t* idx,
                   const float* alpha, const float* pe, const int32_t* pos,
                   const float* add, const int32_t* add_idx, void* stream);

/* ChunkedEmbedding (model.py:147-159) + optional leading identity row + sine positional
 * embedding + optional add row selected by a DEVICE index (the DDPM step):
 * out[rep][r][:] = concat_q tables[q][codes[r-lead][q]] + alpha*pe[r] + add[*add_index]. */
M5_API int m5_chunked_embed(float* out, int64_t ld_rep, int n_rep, int R, int D, int n_q, int n_codes,
                     const float* tables, const int64_t* codes, const float* lead_row,
                     const float* alpha, const float* pe, const float* add, const int32_t* add_index,
                     void* stream);

/* AR prefill: rotate q,k (interleaved-pair RoPE, nn_future.py:181-191) of a row-major
 * [M][3D] qkv buffer at positions pos0..pos0+M-1; write Q head-major, K and V into the KV
 * cache ([h][slot][hd], slot = pos % window, nn_future.py:249-252) and V^T scratch. */
M5_API int m5_rope_cache(int dtype, const void* qkv, int M, int n_heads, int pos0, const float* rope,
                  void* q_out, void* kcache, void* vcache, int64_t cache_hs, int window,
                  void* vt_out, int64_t vt_hs, int64_t vt_ds, void* stream);

/* Cross-attention against a short memory with the projections absorbed into the memory (16-bit engines; NAR decoder,
 * model.py:179-203).  Per head h: scores = x (K_h Wq_h / 8)^T + K_h bq_h / 8 =: x A_h^T + c_h and
 * out = sum_h softmax(scores_h) (V_h Wo_h^T) + bo =: P B + bo, so a layer needs two GEMMs of width n_heads * Lp instead of
 * q-projection + attention + out-projection.
 *  - m5_xattn_absorb builds, for the memory block of the device step index *step, A [n_heads*Lp][D] (16-bit), c [n_heads*Lp]
 *    (fp32; padded keys -1e30) and B^T [D][n_heads*Lp] (16-bit) of every (layer, sequence) in ONE launch.
 *    tab_seq [n_layers*n_seq][8] (device int64): {K base, V base (both [H][Le][64] inside a step block), Le, step stride
 *    (elements), A out, c out, B^T out, 0};  tab_layer [n_layers][4]: {WqT ([H][D][64]: Wq[h*64+d][n] at [h][n][d]),
 *    Wo ([D][D] row-major), bq (fp32 [D]) or 0, 0}.  Lp = 48 or 64 >= every Le.
 *  - m5_xattn_scores: P[b] = per-head softmax(X[b] A[b]^T + c[b]) (16-bit [M][n_heads*Lp], row stride ldp) for `batch`
 *    sequences (strides sX, sA_tab, sc_tab, sP in elements).  Then m5_gemm(P, B^T, EPI_RESIDUAL, batch) finishes the block.
 *    Deferred LayerNorm (below): tab_seq[..][7] = s out (fp32 [n_heads*Lp]) and tab_layer[..][3] = fp32 [D] row sums of the
 *    query weights make m5_xattn_absorb also write s[h Lp + j] = scale K_h[j] . rowsum(Wq_h) = sum_k A[h Lp + j][k]. */
M5_API int m5_xattn_absorb(int dtype, const int64_t* tab_seq, const int64_t* tab_layer, int n_layers, int n_seq, int n_heads,
                    int D, int Lp, const int32_t* step, float scale, void* stream);
M5_API int m5_xattn_scores(int dtype, const void* X, int64_t ldx, int64_t sX, const void* A, int64_t sA_tab, const float* c, int64_t sc_tab,
                    void* P, int64_t ldp, int64_t sP, int M, int n_heads, int Lp, int K, int batch, void* stream);

/* LayerNorm DEFERRED into the GEMM that consumes it (NAR decoder, model.py:179-203: norm1 / norm2 / norm3 of the pre-LN
 * nn.TransformerDecoderLayer).  For rows x (fp32 residual stream, D features) and a Linear (W, b) behind LayerNorm(gamma, beta):
 *     LN(x) W^T + b = r (xt W'^T - d s) + b',   xt = dtype(x - cen),  W' = dtype(W diag gamma),  s[n] = sum_k W'[n][k],
 *     b' = b + W beta,  d = mean(x - cen),  r = 1 / sqrt(var(x) + eps)           (cen: any per-row constant near the mean)
 * so the LayerNorm launch, its read of x and the normalised copy disappear:
 *  - mode 1, the PRODUCER: m5_gemm_ex(M5_EPI_RESIDUAL) updates x in place as m5_gemm does and also writes xt (row stride
 *    ld_xt) and, per row and 128-column tile tn, part[row][tn] = {sum, sum of squares} of (x_new - cen_in[row]) over the tile
 *    (np = N / 128, even, <= 8).  The row's centre is cen_in[row] + delta[row] (NULL = 0) and is left in cen_out[row].
 *  - mode 2, a CONSUMER: m5_gemm_ex(M5_EPI_QKV / M5_EPI_SWIGLU) and m5_xattn_scores_ex take A = xt and W = W', bias = b',
 *    derive (d, r) per row from the np partial pairs (n_feat = D, eps) and apply the formula in their epilogues; the column
 *    tile 0 workgroups also write delta[row] = d (centre + d = the row's mean: the next producer's centre; a consumer
 *    never READS a centre, so nothing in it waits on another launch's small stores).
 * Rows of xt / part / cen are numbered bz * rows_bs + m for batched launches.  m5_layernorm_mean starts a chain (an explicit
 * LayerNorm that also leaves the row means).  16-bit operands only; M5_ERR_UNSUPPORTED for shapes / tilings without the
 * vector epilogues (N % 128 for a producer).  Exact in exact arithmetic; in 16 bits the rounding moves from LN(x) to x - cen and
 * from W to W diag gamma (same count and magnitude: tests/test_gpu_kernels.py, tests/test_gpu_parity16.py). */
typedef struct {
    int32_t mode;             /* 1 producer, 2 consumer                                           */
    int32_t np;               /* partial pairs per row                                            */
    void* xt; int64_t ld_xt;  /* producer: centred operand-type copy of the updated rows          */
    float* part;              /* [rows][np][2] fp32                                               */
    const float* cen_in;      /* producer: [rows] fp32 or NULL (0)                                */
    float* cen_out;           /* producer: [rows] fp32 or NULL: the centre it used (another buffer than cen_in) */
    float* delta;             /* [rows] fp32 or NULL: consumer writes d, the next producer reads it */
    const float* s;           /* consumer: [N] fp32 row sums of W'                                */
    int64_t s_bs;             /* consumer: batch stride of s (elements)                           */
    float eps;                /* consumer                                                         */
    int32_t n_feat;           /* consumer: D                                                      */
    int32_t rows_bs;          /* rows per batch entry                                             */
} M5DeferredLN;
/* Row-tile lists: a batch of sequences of DIFFERENT lengths in one padded layout (sequence b = rows b * rows_per_seq ..,
 * its first len_b rows real, rows_per_seq a multiple of 384).  For each tile height BM in {96, 128, 192}, map[i] lists the
 * row tiles b * (rows_per_seq / BM) + t with t * BM < len_b (device int32, n[i] entries, ascending); a launch that gets the
 * lists runs only those tiles (the kernel picks the list of its own tile height), so the padding costs nothing and the
 * XCD-contiguous tile order stays balanced.  Flat launches (batch = 1, M = all rows) and batched ones (batch = sequences,
 * M = rows_per_seq) use the same lists.  Rows of tiles that are not listed are neither read nor written.
 * The three lists cover DIFFERENT pad rows of a sequence (ceil(len / BM) * BM differs per height), so a pad row may be read by a
 * consumer of one height after only a producer of another height -- or none -- wrote it: every buffer of the layout must
 * start out finite (the host zero-fills), and with seq_len given a deferred-LayerNorm consumer treats the rows >= len_b of
 * its tiles as d = r = 0 (output = b' alone: finite whatever partials the row holds), so nothing a pad row carries can grow
 * through the 1 / sqrt(eps) of a stale statistic into an Inf that p = 0 would turn into NaN inside a real row's attention. */
typedef struct {
    const int32_t* map[3];    /* BM = 96, 128, 192                                                */
    int32_t n[3];
    int32_t rows_per_seq;
    const int32_t* seq_len;   /* [sequences] device int32: len_b (may be NULL: no pad-row treatment) */
} M5RowTiles;
/* m5_gemm / m5_xattn_scores with either or both of the two (NULL = without). */
M5_API int m5_gemm_ex(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
               void* C, int64_t ldc, int M, int N, int K, int epi, const M5QkvScatter* sc,
               int batch, int64_t sA, int64_t sW, int64_t sC, int64_t sBias, const M5DeferredLN* dl, const M5RowTiles* rt, void* stream);
M5_API int m5_xattn_scores_ex(int dtype, const void* X, int64_t ldx, int64_t sX, const void* A, int64_t sA_tab, const float* c, int64_t sc_tab,
                       void* P, int64_t ldp, int64_t sP, int M, int n_heads, int Lp, int K, int batch, const M5DeferredLN* dl,
                       const M5RowTiles* rt, void* stream);
/* y = normalise(LayerNorm(x; gamma, beta, eps); eps2) without a second affine, one pass: the NAR output heads' LayerNorm on top of
 * the decoder's final LayerNorm (model.py:236-242,342) -- the seven heads share the statistics, their gamma / beta are folded
 * into the head weights, so one normalised copy serves all heads.  Rows: n_seq runs of rows_per_seq rows; run s of x starts
 * x_seq_stride rows after run s - 1, y is packed.  Vector rows only (D % 256 == 0, 16-byte aligned). */
M5_API int m5_layernorm_twice(int out_dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float eps2,
                       void* y, int64_t ldy, int rows_per_seq, int n_seq, int64_t x_seq_stride, int D, void* stream);
M5_API int m5_layernorm_mean(int out_dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                      void* y, int64_t ldy, int M, int D, float* mean_out, void* stream);

/* ------------------------------------------------------------------------------------
 * AR decode step (one token, batch 1): weight-streaming GEMV with fused prologue/epilogue.
 * Device-resident state lets one hipGraph replay serve every step.
 * ------------------------------------------------------------------------------------ */
#define M5_ST_POS 0        /* RoPE position of the current query (= prefix length)          */
#define M5_ST_NGEN 1       /* tokens generated so far                                       */
#define M5_ST_DONE 2       /* 1 after EOS or when max_len is reached: later steps no-op     */
#define M5_ST_NTOK 3       /* total tokens stored (prompt + generated)                      */
#define M5_ST_LAST 4       /* last sampled token id                                         */
#define M5_ST_WORDS 8

#define M5_PRO_RMS 0       /* xs = dtype((x*rsqrt(mean(x^2)+eps))*w)  (nn_future.py:307-312) */
#define M5_PRO_DT 1        /* xs = x (dtype vector)                                         */
#define M5_PRO_ATTN 2      /* xs = combine of split-KV attention partials                   */

#define M5_GEPI_QKV_ROPE 0 /* rows q|k|v: RoPE q,k at state.pos, q -> qbuf, k,v -> cache slot */
#define M5_GEPI_RESIDUAL 1 /* xres[n] += acc                                                */
#define M5_GEPI_SWIGLU 2   /* rows interleaved (w1_i, w3_i): y[i] = silu(a)*b               */
#define M5_GEPI_F32 3      /* y_f32[n] = acc  (logits)                                      */

#define M5_ATTN_PART 66    /* floats per (head, split) partial: o[64], m, l                 */

/* Same-stream weight prefetch (batch-1 decode).  A decode launch lives ~5 us of which the HBM pipe is busy for
 * 1-3: `wgs` extra workgroups appended to a launch touch (one 4-byte load per 64 bytes, result discarded) the
 * weights a LATER launch of the step will stream, so they are on their way into L2 / Infinity Cache while this
 * launch's own dependency chain runs.  The region is `n_chunks` chunks of `chunk_bytes` starting at chunk
 * `first_chunk` of `ptr`, chunk j being what workgroup j of the consuming launch reads; a prefetching workgroup
 * only touches chunks j with j % 8 == its own (block id % 8), i.e. (observed placement, speed only) the chunks its
 * own XCD's L2 will be asked for.  ptr == NULL or wgs == 0: off.  Never changes any result. */
typedef struct {
    const void* ptr; int64_t chunk_bytes; int32_t first_chunk, n_chunks, wgs, pad_;
} M5Prefetch;

typedef struct {
    const void* W; int64_t ldw; int32_t N, K;
    const float* x_f32; const float* norm_w; float eps;   /* PRO_RMS                        */
    const void* x_dt;                                      /* PRO_DT                         */
    const float* part; int32_t nsplit, n_heads;            /* PRO_ATTN                       */
    void* y_dt; float* y_f32; float* xres;
    const float* rope; const int32_t* state;
    void* kcache; void* vcache; void* qbuf;                /* this layer's cache [h][W][64]  */
    int32_t w_alloc, window, dim;
    unsigned long long* dbg;                               /* diagnostics: phase stamps of workgroup 0 (NULL = off) */
    M5Prefetch pf;                                         /* optional: the NEXT launch's weights (streaming geometry only) */
} M5GemvArgs;
M5_API int m5_ar_gemv(int dtype, int pro, int epi, const M5GemvArgs* a, void* stream);

typedef struct {
    const void* qbuf; const void* kcache; const void* vcache;
    float* part; const int32_t* state;
    int32_t n_heads, w_alloc, window, nsplit;
    float scale;
    /* batched decode (several sequences per launch; 0 / 1 = one sequence, strides ignored):
     * sequence b uses qbuf + b*q_bs, k/vcache + b*cache_bs (elements), part + b*part_bs (floats),
     * state + b*state_bs (int32 words). */
    int32_t batch, state_bs;
    int64_t q_bs, cache_bs, part_bs;
    M5Prefetch pf;                                         /* optional: a later launch's weights (batch 1 only) */
} M5AttnDecodeArgs;
/* nn_future.py:257-272 decode branch: q . K[:min(pos+1,W)] softmax . V, split over keys. */
M5_API int m5_ar_attn_decode(int dtype, const M5AttnDecodeArgs* a, void* stream);

/* Batched decode step (BASELINE config 3: B sequences advance one token per step; the projections run
 * as M = B row GEMMs through m5_gemm, whose M <= 32 path streams every weight once for the whole batch).
 * Per-sequence state words as above at state + b*state_bs; finished sequences (ST_DONE) are skipped.
 *  - m5_ar_rope_cache_batch: row b of the [B][3D] qkv buffer is rotated at ITS position (state POS),
 *    q -> qbuf + b*q_bs ([h][64]), k / v -> this layer's cache of sequence b at slot pos % window
 *    (same arithmetic as m5_rope_cache, nn_future.py:181-191,249-252).
 *  - m5_ar_attn_combine_batch: merges the split-KV partials of m5_ar_attn_decode into out[b][D] (dtype). */
M5_API int m5_ar_rope_cache_batch(int dtype, const void* qkv, int B, int n_heads, const float* rope, const int32_t* state,
                           int32_t state_bs, void* qbuf, int64_t q_bs, void* kcache, void* vcache, int64_t cache_bs,
                           int64_t cache_hs, int window, void* stream);
/*  - m5_ar_qkv_rope_batch: the QKV projection of the batched step with that rotation and the cache write fused
 *    into its epilogue: xn [B][K] (dtype) x wqkv [3D][K] -> qbuf / caches; qkv_tmp [B][3D] is scratch for the fp32
 *    (parity mode) path, which runs m5_gemm + m5_ar_rope_cache_batch. */
M5_API int m5_ar_qkv_rope_batch(int dtype, const void* xn, int64_t lda, const void* wqkv, int64_t ldw, int B, int n_heads, int K,
                         const float* rope, const int32_t* state, int32_t state_bs, void* qbuf, int64_t q_bs,
                         void* kcache, void* vcache, int64_t cache_bs, int64_t cache_hs, int window, void* qkv_tmp, void* stream);
M5_API int m5_ar_attn_combine_batch(int dtype, const float* part, int64_t part_bs, int B, int n_heads, int nsplit,
                             const int32_t* state, int32_t state_bs, void* out, int64_t out_bs, void* stream);

/* The n_layers Mistral layers of one decode step as ONE persistent launch (csrc/ar_mega.hip): 256 co-resident workgroups,
 * the five launches of a layer become phases with the same row / lane / k mapping (bit-identical results), dependent
 * vectors cross between workgroups as tagged 8-byte granules, later phases' weight rows stream in underneath the edges.
 * 16-bit operands, CodecLM geometry only (dim 1536, hidden 3584, 24 heads, 8 key splits); needs >= 256 CUs; the stacked
 * per-layer weights are contiguous: wqkv [L][3D][D], wo [L][D][D], w13 [L][2F][D] (rows interleaved W1_i, W3_i), w2 [L][D][F],
 * norms [L][D] fp32, caches [L][H][w_alloc][64].  gran: M5_AR_MEGA_GRANULES 8-byte words, zeroed by the host at every
 * prefill (tags are unique within an utterance only).  err[0] != 0 after a launch: a workgroup gave up waiting (the grid was
 * not co-resident); sticky -- later launches return at once -- the caller re-runs the step with the per-launch entry points.
 * M5_ERR_UNSUPPORTED (nothing launched) for other geometries / fewer CUs.  nn_future.py:235-274,326-333. */
#define M5_AR_MEGA_GRANULES 21760
typedef struct {
    const void* wqkv; const void* wo; const void* w13; const void* w2;
    const float* attn_norm; const float* ffn_norm; float eps;
    int32_t dim, hidden, n_heads, layer0, layer1;          /* layers [layer0, layer1)                */
    float* xres;                                           /* [dim] residual stream in / out         */
    const float* rope; const int32_t* state;
    void* kcache; void* vcache; int32_t w_alloc, window;
    float scale;
    uint64_t* gran; uint32_t* err;
    unsigned long long* dbg;                               /* diagnostics (tools build): phase stamps of workgroup 0, or NULL */
} M5ArMegaArgs;
M5_API int m5_ar_layers_persistent(int dtype, const M5ArMegaArgs* a, void* stream);

/* Sampler chain of ar_generate.py:74-115 + samplers.py:20-93 on device, then the
 * multinomial draw argmax(p / q) with caller-supplied Exp(1) noise, EOS / max_len
 * handling (ar_generate.py:62,121) and the next token's embedding load (model.py:106). */
typedef struct {
    const float* logits; int32_t V;
    int32_t* state; int64_t* tokens; int32_t max_len;
    float alpha_frequency, alpha_presence; int32_t penalty_window;
    int32_t n_text; int32_t eos_idx;
    int32_t n_est; const float* eos_table;      /* eos_table[n], n = 0..n_est, or NULL       */
    float temperature; int32_t div_mode;         /* 0: z / T (CPU reference), 1: z * (1/T)    */
    int32_t top_k; float top_p;
    float typical_p;                             /* samplers.py:96-122; > 0.999 disables (reference default 1.0) */
    const float* noise; int64_t noise_stride;    /* Exp(1) draws [step][V]                    */
    const float* embed; int32_t dim; float* xres;
    /* batched decode: one workgroup per sequence (0 / 1 = one sequence).  Sequence b uses
     * logits + b*logits_bs, state + b*state_bs, tokens + b*tokens_bs, noise + b*noise_bs,
     * xres + b*xres_bs, eos_table + b*eos_table_bs, and n_est_b[b] / max_len_b[b] when given. */
    int32_t batch, state_bs;
    int64_t logits_bs, tokens_bs, noise_bs, xres_bs, eos_table_bs;
    const int32_t* n_est_b; const int32_t* max_len_b;
    /* noise = NULL: the Exp(1) row of sampler call i (= n_gen) is generated here, bit-identical to the i-th
     * `torch.empty(V).exponential_(1, generator=g)` of a generator that stood at (seed, offset0) before call 0 (torch.multinomial's
     * draw, ar_generate.py:115): rng = device {seed, offset0} (sequence b: rng + b * rng_bs words), call i draws at offset0 + i * noise_inc,
     * noise_grid = torch's launch width for V values (see m5_nar_uniforms).  Only the kept tokens' values are ever computed. */
    const uint64_t* rng; int64_t rng_bs;
    uint32_t noise_inc, noise_grid;
} M5SampleArgs;
M5_API int m5_ar_sample(const M5SampleArgs* a, void* stream);

/* ------------------------------------------------------------------------------------
 * NAR reverse-diffusion step tail (diffuser.py:364-393 + :467-468): CFG mix, temperature,
 * log_softmax, multinomial posterior, Gumbel-argmax with caller-supplied uniforms, known
 * branch q_sample, inpaint merge and L0 override, for every (frame, codebook).
 * ------------------------------------------------------------------------------------ */
#define M5_NAR_CONSTS 8   /* per step: lca[t-1], l1mca[t-1]-lnK, la[t], l1ma[t]-lnK, lca[t], l1mca[t]-lnK, t, unused */
typedef struct {
    const float* logits_c; const float* logits_u;   /* [S-row_offset][n_q-1][ldk] (codebooks 1..)  */
    int64_t ld_row, ld_q;
    int32_t S, n_q, K, row_offset;
    int64_t* x; const int64_t* x_known; const uint8_t* m;
    const float* u1; const float* u2;               /* [S][n_q][K] uniforms                   */
    const float* consts; const int32_t* step;       /* consts[*step][M5_NAR_CONSTS]            */
    float guidance_w, temperature, log_eps;         /* log_eps = log(1e-7f)                    */
    int32_t div_mode, q0_override_steps;
} M5NarSampleArgs;
M5_API int m5_nar_sample(const M5NarSampleArgs* a, void* stream);

/* The reverse step's uniforms, bit-identical to what `torch.rand((1, S, n_q, K), device=..., generator=g)` draws on this device
 * (reference diffuser.py:219-228, 380-390 draws them with torch.rand_like, twice per step, once at t = 0), generated inside this
 * library so that the whole reverse step can be one captured hipGraph and the loop holds no ATen launch.  torch's Philox4x32-10
 * geometry: grid_threads = 256 * min(CUs * (maxThreadsPerMultiProcessor / 256), ceil(n / 256)); a draw of n values advances the
 * generator offset by inc = ceil(n / (4 * grid_threads)) * 4.  rng (DEVICE memory, so that a captured graph serves every run) =
 * {seed, offset0}: reverse step i = *step draws at offset0 + 2 i inc and, unless consts[i][6] (= t) is 0, at offset0 + (2 i + 1) inc
 * (offset0 = the generator's offset when step 0 of the session ran; a multiple of 4 like every torch offset).
 * out[e] = m[e / K] ? second draw : first draw -- m5_nar_sample reads the first draw on the rows it samples from the model and the
 * second on the known rows, so one merged buffer serves as its u1 AND u2.  m = NULL: out = the first draw alone (= torch.rand).
 * k_magic / k_shift: e / K = (e * k_magic) >> (32 + k_shift) for every e < n (0 = divide); the host computes and the call checks it. */
typedef struct {
    float* out; int64_t n;                          /* n = S * n_q * K < 2^32                  */
    int32_t K; uint32_t k_magic, k_shift;
    const uint8_t* m;                               /* [S * n_q] or NULL                       */
    const uint64_t* rng;                            /* device: {seed, offset0}                 */
    uint32_t inc, grid_threads;
    const int32_t* step; const float* consts;       /* as in M5NarSampleArgs (may be NULL with m = NULL) */
    int32_t transform;                              /* 0: torch.rand (uniform [0, 1));  1: Tensor.exponential_(1) of the same draw */
} M5NarUniformArgs;
M5_API int m5_nar_uniforms(const M5NarUniformArgs* a, void* stream);

/* AR -> NAR hand-off on device (reference inference.py:272-275 with speechtok.decode_int, minbpe/codebook.py:88-126):
 * tokens[i] (global AR ids, i < n) -> speech-vocabulary id max(tokens[i] - n_text, 0) -> the run of codebook-0 codes that
 * BPE token was merged from: vals[off[id] .. off[id + 1]) (CSR over n_vocab ids; special tokens have empty runs),
 * concatenated in order into out[0 .. *total) (at most out_cap are written; *total is the full length). */
M5_API int m5_expand_tokens(const int64_t* tokens, int n, int n_text, const int32_t* off, const int64_t* vals, int n_vocab,
                     int64_t* out, int out_cap, int32_t* total, void* stream);

/* Silence trim on device (reference mars5/trim.py:110-178, librosa.effects.trim semantics with centred reflect-padded
 * frames): y mono fp32 [n]; power: scratch of n_frames = 1 + n / hop floats (left holding the frame powers);
 * bounds[0..1] = [start, end) in samples (0, 0 when everything is silent).  Needs n > frame_length / 2. */
M5_API int m5_trim_bounds(const float* y, int n, int frame_length, int hop, float top_db, float* power, int n_frames,
                   int32_t* bounds, void* stream);

M5_API int m5_add_int(int32_t* p, int32_t delta, void* stream);

/* ------------------------------------------------------------------------------------
 * Stage-level entry points: one call enqueues a whole stage from a caller-filled plan (csrc/stage_plan.hip).
 *   m5_nar_step        one NAR reverse step = forward of both guidance branches + the step's uniforms + posterior / sample +
 *                      step counter (reference mars5/diffuser.py:345-394, :451-468 loop body; model.py:264-343): ~190 launches
 *   m5_ar_decode_step  one AR decode step = the 26 layers + final norm / head + sampler (nn_future.py:369-398,
 *                      ar_generate.py:74-121): 3 launches with the persistent layer kernel, 132 in the per-launch form
 *   m5_stage_run       any op list (conditioning, prefill)
 * A plan is an array of ops in launch order; op.fn names an entry point of THIS header and op.a[] holds its arguments except
 * the trailing stream: integers by value, device pointers by address, floats as IEEE bits in the low 32, pointers to argument
 * structures (M5QkvScatter, M5DeferredLN, M5RowTiles, M5AttnArgs, M5GemvArgs, M5AttnDecodeArgs, M5ArMegaArgs, M5SampleArgs,
 * M5NarSampleArgs, M5NarUniformArgs) as (byte offset into `arena` + 1; 8-byte aligned), 0 = NULL.  Every argument of a step is a
 * pointer, stride or size (the step index, positions and RNG state live in device memory), so a session's plan is filled once
 * and serves every step; the caller owns ops / arena / all buffers and may capture the call in a hipGraph.  Returns the first
 * non-zero status of an op (the ops before it are already enqueued; *failed_op = its index, -1 on success, may be NULL);
 * M5_ERR_ARG for an unknown fn, a wrong argument count, an op that does not belong to the stage kind or an arena offset out of
 * range. */
#define M5_PLAN_MAX_ARGS 24
enum {
    M5_FN_GEMM = 1, M5_FN_GEMM_EX, M5_FN_LAYERNORM, M5_FN_LAYERNORM_TWICE, M5_FN_LAYERNORM_MEAN, M5_FN_RMSNORM, M5_FN_ATTENTION,
    M5_FN_GATHER_ROWS, M5_FN_CHUNKED_EMBED, M5_FN_XATTN_ABSORB, M5_FN_XATTN_SCORES, M5_FN_XATTN_SCORES_EX, M5_FN_NAR_UNIFORMS,
    M5_FN_NAR_SAMPLE, M5_FN_ADD_INT, M5_FN_COPY_D2D, M5_FN_AR_GEMV, M5_FN_AR_ATTN_DECODE, M5_FN_AR_LAYERS_PERSISTENT, M5_FN_AR_SAMPLE,
    M5_FN_AR_ROPE_CACHE_BATCH, M5_FN_AR_QKV_ROPE_BATCH, M5_FN_AR_ATTN_COMBINE_BATCH
};
typedef struct {
    int32_t fn, n_args;                 /* M5_FN_*, number of slots used (= the entry point's parameters without the stream) */
    int64_t a[M5_PLAN_MAX_ARGS];
} M5PlanOp;
typedef struct {
    const M5PlanOp* ops; int32_t n_ops;
    const unsigned char* arena; int64_t arena_bytes;      /* copies of the argument structures the ops point to */
    int32_t* failed_op;                                    /* host int32, may be NULL */
} M5StagePlan;
M5_API int m5_nar_step(const M5StagePlan* plan, void* stream);
M5_API int m5_ar_decode_step(const M5StagePlan* plan, void* stream);
M5_API int m5_stage_run(const M5StagePlan* plan, void* stream);
/* dst[0 .. bytes) = src[0 .. bytes), device to device, stream-ordered (capturable) */
M5_API int m5_copy_d2d(void* dst, const void* src, int64_t bytes, void* stream);

/* hipGraph helpers (capture the launches issued between begin/end on `stream`). */
M5_API int m5_graph_begin(void* stream);
M5_API int m5_graph_end(void* stream, void** graph_exec);
M5_API int m5_graph_launch(void* graph_exec, void* stream);
M5_API int m5_graph_destroy(void* graph_exec);

/* HIP-event timing on an arbitrary stream (bench roofline measurements). */
M5_API int m5_event_create(void** ev);
M5_API int m5_event_record(void* ev, void* stream);
M5_API int m5_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronises on ev_stop */
M5_API int m5_event_destroy(void* ev);
/* In-graph timing (bench roofline leg): a one-lane launch that stores the 100 MHz wall clock into *slot (device memory).
 * Captured between two launches of a hipGraph, consecutive stamps bracket the launch between them as the replay runs it. */
M5_API int m5_clock_stamp(uint64_t* slot, void* stream);

#ifdef M5_TOOLS
/* ---- tools library only (libmars5_hip_tools.so, built with -DM5_TOOLS; the scripts under tools/ load it with M5_HIP_TOOLS=1).  The
 * product library libmars5_hip.so exports none of these, reads no environment variable and contains no ablation kernel. */

/* ---- Two fusions that were built, tested and measured SLOWER inside the NAR step on MI355X: they are kept as
 * A/B instruments of the tools library (tools/nar_step_bench.py "M5_GEMM_XATTN=1" / "M5_GEMM_LN=1"), not shipped. */
/* Cross-attention query projection with the attention fused into its epilogue (nn.MultiheadAttention of the NAR
 * decoder against the text memory, model.py:179-203): out[M][H*64] = softmax(scale (A W^T + bias)_h K_h^T) V_h per head,
 * for memories of at most 64 keys whose K / V^T were projected beforehand.  Rows are grouped in sequences of
 * `rows_per_seq` (a multiple of 16); sequence s attends to the memory described by mem_table[s][6] (device int64):
 * {K base address, V^T base address, Le, Lep, step stride of K, step stride of V^T (elements)} with K [H][Le][64] and
 * V^T [H][64][Lep] inside the block selected by the device index *step (the DDPM step).  Returns M5_ERR_UNSUPPORTED
 * (nothing launched; use m5_gemm(EPI_QKV) + m5_attention) unless 16-bit operands, max_le <= 64, even n_heads. */
M5_API int m5_gemm_q_cross_attn(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                         int M, int n_heads, int K, const int64_t* mem_table, int max_le, int rows_per_seq,
                         const int32_t* step, float scale, void* out, int64_t ld_out, void* stream);

/* x[M][N] += A . W^T + bias (the RESIDUAL epilogue of m5_gemm) with the LayerNorm that follows it in every pre-LN
 * block (model.py:179-203: norm2 / norm3 / the next layer's norm1) fused into the same launch:
 * xn = LayerNorm(x_new; gamma, beta, eps) in the operand type.  The workgroups of a row tile exchange per-tile
 * (mean, M2) words through `scratch` (M2 carries an 8-bit launch tag in its low mantissa bits), so the WHOLE grid
 * must be co-resident: returns M5_ERR_UNSUPPORTED (caller falls back to m5_gemm + m5_layernorm) unless 16-bit
 * operands, N % 128 == 0, N <= 2048 and ceil(M/96) * N/128 <= number of CUs.
 * scratch: zero-initialised once by the caller, >= 256 + 768 ceil(M/96) N/128 bytes, reusable by later calls on the
 * same stream PROVIDED consecutive calls carry different launch tags: tag = 1 + (*tag_step * 64 + tag) % 255 with
 * tag_step a device int32 (or NULL = 0) -- under hipGraph replay the kernel arguments are frozen, so the varying part
 * must live in device memory (the DDPM step counter).  scratch[0] (uint32) counts wait timeouts (0 in a healthy run;
 * waits are bounded, a launch never hangs). */
M5_API int m5_gemm_residual_ln(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                        float* C, int64_t ldc, int M, int N, int K, const float* ln_gamma, const float* ln_beta,
                        float ln_eps, void* xn, int64_t ld_xn, void* scratch, int64_t scratch_bytes,
                        const int32_t* tag_step, int tag, void* stream);

/* Diagnostics: placement census of a grid (nblocks x threads, lds_bytes of LDS per workgroup);
 * out[6 * block] = {XCC_ID, HW_ID, start clock lo/hi, end clock lo/hi}.  tools/census.py. */
M5_API int m5_debug_census(uint32_t* out, int nblocks, int threads, int lds_bytes, int spin, void* stream);

/* Diagnostics: n dependent trivial launches (blocks x threads; touch = 1: one read-modify-write of
 * buf[0] per launch) on `stream` -- measures the per-launch floor, eager vs hipGraph.  tools/launch_floor.py. */
M5_API int m5_debug_launch_chain(int32_t* buf, int n, int blocks, int threads, int touch, void* stream);

/* Diagnostics: ONE launch of `blocks` co-resident workgroups that cross `iters` device-wide barriers (agent-scope
 * atomic arrive + bounded acquire spin; mode 1 also passes one word per workgroup across each barrier).
 * scratch: blocks + 4 words; after the run scratch[1] = spin timeouts, scratch[2] = stale reads.  tools/grid_barrier.py. */
M5_API int m5_debug_grid_barrier(uint32_t* scratch, int blocks, int threads, int iters, int mode, void* stream);

/* Diagnostics: subsequent 16-bit m5_gemm launches record {shader clock, 100 MHz wall clock} at the entry
 * and at the end of the main loop of workgroup 0 into buf[0..3] (device memory); NULL disables. */
M5_API int m5_debug_gemm_clock(unsigned long long* buf);

/* Diagnostics: operand-feed probe -- every workgroup stages the same L2-resident panel into LDS `iters`
 * times; mode 0 LDS-DMA, 1 global_load -> ds_write, 2 global loads only.  tools/feed_probe.py. */
M5_API int m5_debug_feed_probe(const void* src, int64_t panel_bytes, int iters, int row_bytes, int mode, int blocks, int threads,
                        float* sink, void* stream);

/* Diagnostics: workgroup b streams chunk (b + shift) % blocks of buf (chunk_bytes each; nt = non-temporal loads): pairs of
 * launches with equal / different shifts measure whether an XCD's L2 keeps lines across a kernel boundary.  tools/l2_retention.py. */
M5_API int m5_debug_l2_touch(const void* buf, int64_t chunk_bytes, int blocks, int threads, int shift, int nt, float* sink, void* stream);
/* All-gather edge probe (tools/edge_probe.py): `blocks` co-resident workgroups each publish `per` values of an n-vector
 * as 8-byte {fp32, tag} granules and sweep the whole vector, `iters` times in one launch, optionally under a weight
 * stream of stream_kb KiB per workgroup and edge: the price of one dependency edge of a persistent decode step. */
M5_API int m5_debug_edge_probe(uint64_t* gran, int n, int per, int blocks, int threads, int iters, uint32_t base_tag,
                        const void* wbuf, int stream_kb, uint32_t* err, float* sums, void* stream);
#endif /* M5_TOOLS */

#ifdef __cplusplus
}
#endif
#endif /* MARS5_HIP_H */
