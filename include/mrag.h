/*
 * mrag.h -- C ABI of the MI355X-native dense-retrieval hot path (libmrag_hip.so).
 *
 * Drop-in boundary for the embedding + cosine top-k path of
 * AndyUkJ/A-Modular-RAG-Framework.  The reference is pure Python and has no FFI of
 * its own; each entry point below names the reference code it replaces
 * (paths relative to the reference checkout).  Binding a maintainer would add:
 * INTEGRATION.md (ctypes).
 *
 * Conventions
 *   - every function returns an int status: MRAG_OK (0) or a negative MRAG_ERR_*;
 *     mrag_last_error() returns a thread-local message for the last failure.
 *   - handles are opaque uint64 values; 0 is never a valid handle.
 *   - all buffers are caller-owned, plain pointers + sizes.  A pointer argument is
 *     a HOST pointer unless the matching *_is_device flag is non-zero, in which
 *     case it is a device pointer on the handle's device (e.g. a torch tensor's
 *     data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls
 *     with device outputs are asynchronous on that stream; calls with host outputs
 *     return after the results have landed.
 *   - one call in flight per handle (the reference is single-threaded).
 *   - there is NO CPU fallback: without a usable GPU every compute entry point
 *     fails with MRAG_ERR_NO_DEVICE.
 */
#ifndef MRAG_H_
#define MRAG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRAG_ABI_VERSION 1

#define MRAG_OK 0
#define MRAG_ERR_INVALID (-1)     /* bad argument / bad handle */
#define MRAG_ERR_NO_DEVICE (-2)   /* no HIP device, or device index out of range */
#define MRAG_ERR_HIP (-3)         /* a HIP runtime call failed */
#define MRAG_ERR_OOM (-4)         /* device or host allocation failed */
#define MRAG_ERR_UNSUPPORTED (-5) /* valid request this build cannot serve (e.g. k too large) */

typedef uint64_t mrag_handle;

enum mrag_dtype { MRAG_F32 = 0, MRAG_F16 = 1, MRAG_BF16 = 2, MRAG_F64 = 3 };
enum mrag_metric { MRAG_METRIC_COSINE = 0, MRAG_METRIC_IP = 1 };
enum mrag_pool { MRAG_POOL_MEAN = 0, MRAG_POOL_CLS = 1 };

/* ---- library ------------------------------------------------------------------ */
int mrag_abi_version(void);
const char* mrag_last_error(void);
int mrag_device_count(int* out_count);

/* ---- a1/a2: pairwise cosine, replaces DenseReranker._cosine x N ------------------
 * app/modules/retrieval/retrieval_backend.py:192-197 (formula, zero/length guards)
 * and :245 (one query against every candidate).  fp64 on the device:
 * out[i] = dot(q, c_i) / (|q| * |c_i|), 0.0 when either norm is 0.  `cands` is
 * row-major [n, dim] fp64.  A dimension mismatch between query and candidates is the
 * caller's to map to 0.0 (the reference compares list lengths, :193). */
int mrag_cosine_f64(int device, const double* query, const double* cands, int64_t n, int dim,
                    double* out_scores, int is_device, void* stream);

/* all-pairs form (SURVEY 8f "next" rows): out[i*n+j] = cosine(x_i, x_j), fp64, same zero-norm rule.
 * Serves the semantic-edge pass of graph construction (app/modules/graph_construction/
 * edge_builder.py:146-169: per pair three norms + one dot, edge if sim >= threshold) and the
 * pairwise term of MMR (app/utils/similarity.py:33-62) in one launch instead of n^2 Python calls. */
int mrag_cosine_matrix_f64(int device, const double* x, int64_t n, int dim, double* out, int is_device, void* stream);
/* adjacent rows (embed-mode segmentation, app/modules/graph_construction/segmenter.py:32-50):
 * out[i] = dot(x_i, x_{i+1}) / (|x_i| * |x_{i+1}| + eps) for i < n-1 -- eps = 1e-9 sits INSIDE the denominator
 * there and a zero vector gives 0/eps = 0, not a special case.  Host buffers, fp64. */
int mrag_cosine_adjacent_f64(int device, const double* x, int64_t n, int dim, double eps, double* out, void* stream);

/* ---- a7 at corpus scale: brute-force cosine / inner-product top-k ----------------
 * Replaces "score every candidate, sort descending, truncate" --
 * retrieval_backend.py:245 + :371-372 (and retrieval_adapter.py:129-131) -- over the
 * whole docs.jsonl corpus (ingest row order, my_code/ingest_hotpotqa.py:73-81).
 * Rows are stored in HBM as fp16/bf16 [n, dim padded to 64]; tie-break is
 * (score desc, row id asc). */
int mrag_index_create(int dim, int metric, int storage_dtype, int device, mrag_handle* out);
int mrag_index_destroy(mrag_handle h);
int mrag_index_reserve(mrag_handle h, int64_t n_rows);
/* append n rows (row-major [n, dim] of src_dtype).  normalize != 0: L2-normalise each
 * row (fp64 norm, zero rows stay zero -> cosine 0.0 like :197) before rounding once to
 * the storage dtype; normalize == 0: rows are rounded/stored as given. */
int mrag_index_add(mrag_handle h, const void* rows, int64_t n, int src_dtype, int normalize,
                   int rows_is_device, void* stream);
int mrag_index_size(mrag_handle h, int64_t* out_rows);
int mrag_index_dim(mrag_handle h, int* out_dim);
/* ids reported by search = id_base + local row (row-sharded corpora) */
int mrag_index_set_id_base(mrag_handle h, int64_t id_base);
/* copy stored (rounded) rows [row0, row0+n) back as fp32 [n, dim] -- for tests/caches */
int mrag_index_get_rows(mrag_handle h, int64_t row0, int64_t n, float* out, int out_is_device, void* stream);
/* largest k mrag_index_search serves for a batch of nq queries (256: the reference's dense pool is 200
 * candidates per question, config/settings.yaml:101-102 / retrieval_backend.py:218,276).  k <= 64 runs
 * the fused batch kernel; 64 < k <= 256 runs the streaming kernel 8 queries per launch (exact, HBM-bound) for up
 * to 32 queries, and for larger batches the batch kernel at k' = 64 per corpus split + merge + an exactness check,
 * with the few queries the check flags redone by the streaming kernel (such a search synchronises the stream once). */
int mrag_index_max_k(int64_t nq, int* out_k);
/* measurement hook: queries the LAST 64 < k <= 256 batch search (any index of this process) had to redo through the
 * streaming kernel because one corpus split held more than 64 of their true top k (0 for every other kind of search). */
int mrag_index_last_wide_redone(int64_t* out_queries);
/* top-k of every query against every stored row.  out_scores [nq,k] fp32 descending,
 * out_ids [nq,k] int64; slots past the corpus size hold (-inf, -1).  k > mrag_index_max_k:
 * MRAG_ERR_UNSUPPORTED (never a silent clamp). */
int mrag_index_search(mrag_handle h, const void* queries, int64_t nq, int q_dtype, int normalize,
                      int queries_is_device, int k, float* out_scores, int64_t* out_ids,
                      int out_is_device, void* stream);
/* the re-ranker's candidate lookup by row id.  Replaces the 1 + ceil(N/bs) embed calls +
 * N x _cosine of DenseReranker.score (retrieval_backend.py:227-245) for candidates whose rows are already
 * in the index: out_scores[i] = <query, stored row row_ids[i]> (the cosine when rows and query are
 * normalised); an id outside [0, size) scores 0.0 like an unusable vector (:193-197).  Host buffers. */
int mrag_index_score_rows(mrag_handle h, const void* query, int q_dtype, int normalize, const int64_t* row_ids,
                          int64_t n, float* out_scores, void* stream);
/* timing hooks for bench.py: device time in ms of the dominant kernel (fused similarity
 * GEMM + top-k) and of the whole search of the LAST mrag_index_search call, measured with
 * hipEvents on the stream the kernels were launched on.  Blocks until that call is done. */
int mrag_index_last_timing(mrag_handle h, float* out_gemm_ms, float* out_total_ms);
/* held shader clock under the fused kernel (SURVEY 8d "record held clock"): with the reading enabled, workgroup 0
 * of the batch kernel stamps s_memtime / s_memrealtime at entry and exit; last_clock = d(cycles) / d(100 MHz ticks).
 * MRAG_ERR_UNSUPPORTED when the last search ran the streaming kernel (no stamps). */
int mrag_index_measure_clock(mrag_handle h, int enable);
int mrag_index_last_clock(mrag_handle h, float* out_ghz);

/* ---- 8e: host-side merge of per-shard partial top-k ------------------------------
 * scores [nparts, nq, k] fp32 and ids [nparts, nq, k] int64 (ids < 0 = empty slot), as
 * gathered by the RCCL all-gather.  Result ordered by (score desc, id asc). */
int mrag_topk_merge(const float* scores, const int64_t* ids, int nparts, int64_t nq, int k,
                    float* out_scores, int64_t* out_ids, int nthreads);

/* the same merge for buffers that already sit in HBM (the RCCL all-gather output): every pointer is a
 * device pointer on `device`, asynchronous on `stream`.  Each part must be sorted (score desc, id asc)
 * with its empty slots last, as mrag_index_search produces them.  Serves nparts <= 64 and
 * nparts * k <= 2048; larger requests return MRAG_ERR_UNSUPPORTED (use the host merge). */
int mrag_topk_merge_device(int device, const float* scores, const int64_t* ids, int nparts, int64_t nq, int k,
                           float* out_scores, int64_t* out_ids, void* stream);
/* the exchange's packed form (ONE all-gather of 8-byte words): word = (fp32 score bits << 32) | shard-local row
 * (id - id_base as 32 bits; 0xFFFFFFFF = empty slot).  pack: n = nq * k device entries -> words; an id outside
 * [id_base, id_base + 2^32 - 1) ORs 1 into *bad_flag (device int, may be NULL) -- nothing is wrapped silently.
 * merge_packed: the gathered words [nparts, nq, k] + the parts' first global rows bases[nparts] (device) -> [nq, k], same order
 * and limits as mrag_topk_merge_device (the words are unpacked on their way into LDS: no pass over the gathered buffer). */
int mrag_pack_partial_device(int device, const float* scores, const int64_t* ids, int64_t n, int64_t id_base,
                             int64_t* words, int* bad_flag, void* stream);
int mrag_topk_merge_packed_device(int device, const int64_t* words, const int64_t* bases, int nparts, int64_t nq, int k,
                                  float* out_scores, int64_t* out_ids, void* stream);

/* ---- IVF-flat (BASELINE.json config 5; no counterpart in the reference) ----------- */
int mrag_ivf_create(int dim, int nlist, int metric, int storage_dtype, int device, mrag_handle* out);
int mrag_ivf_destroy(mrag_handle h);
/* spherical k-means on the device, seeded: init = rows picked by the host from `seed` */
int mrag_ivf_train(mrag_handle h, const void* rows, int64_t n, int src_dtype, int normalize,
                   int rows_is_device, int iters, uint64_t seed, void* stream);
int mrag_ivf_set_centroids(mrag_handle h, const void* centroids, int src_dtype, int normalize,
                           int is_device, void* stream);
int mrag_ivf_get_centroids(mrag_handle h, float* out, int out_is_device, void* stream);
int mrag_ivf_add(mrag_handle h, const void* rows, int64_t n, int src_dtype, int normalize,
                 int rows_is_device, void* stream);
int mrag_ivf_size(mrag_handle h, int64_t* out_rows);
int mrag_ivf_set_id_base(mrag_handle h, int64_t id_base);
/* list id of every stored row, in insertion order */
int mrag_ivf_get_assignments(mrag_handle h, int32_t* out, int out_is_device, void* stream);
/* top-k over the `nprobe` nearest lists of every query (nprobe 1..256, or nlist = exhaustive); k <= 64.
 * Order and tie-break as mrag_index_search: (score desc, original row asc); missing entries (-inf, -1).
 * With queries AND results in device memory the call is asynchronous on `stream`; with host buffers it
 * returns when they are complete.  Batches are cut internally so that a chunk's fp32 score segments fit the
 * score buffer (MRAG_IVF_SCORES_MB, default 4096; 0 = always the fused GEMM + top-k kernel). */
int mrag_ivf_search(mrag_handle h, const void* queries, int64_t nq, int q_dtype, int normalize,
                    int queries_is_device, int nprobe, int k, float* out_scores, int64_t* out_ids,
                    int out_is_device, void* stream);
/* measurement hooks for bench.py (like mrag_index_last_timing): device ms of the list scan (scan + per-query selection kernels, or the fused kernel) and of the
 * whole LAST mrag_ivf_search (hipEvents on its stream), the rows that scan streamed (sum over its
 * workgroups of their list's length: x ld x 2 = the algorithmic HBM bytes) and its workgroup count. */
int mrag_ivf_last_timing(mrag_handle h, float* out_scan_ms, float* out_total_ms, int64_t* out_scanned_rows, int* out_n_wg);

/* ---- a5: sentence-encoder forward, fills the LLMProvider.embed slot ---------------
 * app/core/providers/base.py:6, called as embed(model=, texts=, require=) from
 * app/core/llm_router.py:115.  BERT-family encoder (MiniLM-L6 / bge-base shapes,
 * SURVEY 8c): embeddings + LayerNorm, L x {MHA, erf-GELU FFN, post-LN}, pooling,
 * optional L2 normalisation.  Tokenisation stays on the host. */
typedef struct mrag_encoder_config {
  int32_t vocab_size;
  int32_t hidden;
  int32_t layers;
  int32_t heads;
  int32_t intermediate;
  int32_t max_position;
  int32_t type_vocab_size;
  float layer_norm_eps;
  int32_t compute_dtype; /* MRAG_F16 or MRAG_BF16 */
} mrag_encoder_config;

int mrag_encoder_create(const mrag_encoder_config* cfg, int device, mrag_handle* out);
int mrag_encoder_destroy(mrag_handle h);
/* upload one named fp32 parameter (HF BertModel state_dict names, e.g.
 * "embeddings.word_embeddings.weight", "encoder.layer.3.attention.self.query.bias") */
int mrag_encoder_set_param(mrag_handle h, const char* name, const float* data, int64_t numel,
                           int is_device, void* stream);
/* number of parameters still missing (0 = ready) */
int mrag_encoder_missing_params(mrag_handle h, int* out_missing);
/* ids/mask int32 [B,S]; out fp32 [B,hidden] */
int mrag_encoder_forward(mrag_handle h, const int32_t* ids, const int32_t* mask, int B, int S,
                         float* out, int pool, int normalize, int io_is_device, void* stream);
/* measurement hook for bench.py: device ms of the LAST forward's kernels (embeddings .. pooling), from
 * hipEvents on its stream; the ids/mask H2D and the output D2H are outside the bracket. */
int mrag_encoder_last_timing(mrag_handle h, float* out_ms);

/* ---- 8f-3: the BM25 text channel on the device -------------------------------------
 * Replaces BM25LiteIndex.search (app/modules/retrieval/text_index.py:59-97: per candidate document
 * _score_doc over the query tokens, max / sum merge over the expanded queries, sort descending, top_k) and
 * feeds the fusion of retrieval_backend.py:353-372.  The index is built on the host exactly like
 * text_index.py:36-52 (same tokeniser, tf / df / doc_lens / avgdl) and uploaded as CSR postings:
 *   indptr[n_terms+1], post_doc[nnz] (ascending inside a term), post_tf[nnz],
 *   doc_norm[d] = k1 * (1 - b + b * dl_d / avgdl)   (the length term of :66), k1_plus_1 = k1 + 1.
 * search: n_queries token lists (q_ptr[n_queries+1] into q_terms / q_idf; a token outside the vocabulary is
 * id -1; q_idf = the reference's _idf of that token, :54-56).  fp64 throughout; a document's score is the
 * reference's left-to-right sum bit for bit.  merge_sum = 0: max over the queries (alpha_merge="max"), 1: sum.
 * Output: out_n <= k (doc, score) pairs, (score desc, doc asc), scores > 0 only.  Host buffers. */
int mrag_bm25_create(int device, int64_t n_docs, int64_t n_terms, const int64_t* indptr, const int32_t* post_doc,
                     const int32_t* post_tf, const double* doc_norm, double k1_plus_1, mrag_handle* out);
int mrag_bm25_destroy(mrag_handle h);
int mrag_bm25_search(mrag_handle h, int n_queries, const int32_t* q_ptr, const int32_t* q_terms, const double* q_idf,
                     int merge_sum, int k, int64_t* out_docs, double* out_scores, int* out_n, void* stream);

/* a7 on the device: the fusion arithmetic of HybridRetrievalBackend.run (retrieval_backend.py:336-372).
 * keys / scores hold the text, graph and dense channel entries back to back (n_text, n_graph, n_dense); a key is
 * the rank of the hit's NORMALISED id (:283-294) among all ids of the call, so ascending key = ascending id.
 * Per channel: dedupe by key (strictly larger score wins, :336-348), min-max (:296-301); union:
 * alpha_t*ts + alpha_g*gs + alpha_d*ds (:363); order (score desc, key asc); top_k (:371-372).  fp64, the
 * reference's values bit for bit.  Outputs (host, length >= min(top_k, entries)): key, fused score and the three
 * score_*_norm values per kept hit; *out_n = hits written.  At most 4096 entries. */
int mrag_fuse_topk(int device, const int32_t* keys, const double* scores, int n_text, int n_graph, int n_dense,
                   double alpha_text, double alpha_graph, double alpha_dense, int top_k, int32_t* out_keys,
                   double* out_scores, double* out_text_norm, double* out_graph_norm, double* out_dense_norm,
                   int* out_n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRAG_H_ */
