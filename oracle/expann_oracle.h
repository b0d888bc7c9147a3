/*
 * expann_oracle.h -- CPU restatement of expANN's distance + top-k hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call into it, and only as the checker / the
 * reported CPU baseline.  Nothing under expann_amd/ links, loads or falls back to it.
 *
 * Every function cites the reference file:line (relative to the upstream
 * jacketsj/expANN tree) whose arithmetic it restates in plain C.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - top-k selection rule (oracle_topk_*, brute-force admission/eviction/tie order):
 *     PINNED against the reference's own src/topk_t.h compiled unmodified
 *     (oracle/ref/topk_ref.cpp -> oracle/_ref/libtopk_ref.so) and against the
 *     golden vectors generated from it (tests/golden/topk_*.json).
 *   - distance values (oracle_l2_f32 etc.): PARITY UNPINNED.  src/distance.h includes
 *     <Eigen/Dense>, an un-vendored submodule that is absent from this image, so the
 *     reference kernels cannot be built here without writing a stand-in header, which
 *     is not allowed.  The restatement follows the documented semantics of the AVX-512
 *     intrinsics the reference uses, lane for lane.
 */
#ifndef EXPANN_ORACLE_H
#define EXPANN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar distance kernels ------------------------------------------------------ */

/* fp32 squared L2.  src/distance.h:136-147 (distance_compare_avx512f_f16) and, bit for
 * bit the same value, :86-111 (..._batch128) and :112-134 (..._prefetched): 16 lane
 * accumulators, lane l sums dims i == l (mod 16) in increasing i by fused multiply-add
 * of (a-b), then the _mm512_reduce_add_ps tree (l + l+8, then +4, then +2, then +1).
 * d must be a multiple of 16. */
float oracle_l2_f32(const float* a, const float* b, size_t d);

/* fp32 inner product.  src/distance.h:181-190 (dot_avx512f_f16): same lane scheme. */
float oracle_dot_f32(const float* a, const float* b, size_t d);

/* int8 "L2" exactly as the reference computes it.  src/distance.h:29-53
 * (distance_compare_avx512f_i64): the 8-bit subtraction wraps and the difference is
 * ZERO-extended, so each term is ((uint8_t)(a_i - b_i))^2.  d multiple of 64. */
int32_t oracle_l2_i8_refcompat(const int8_t* a, const int8_t* b, size_t d);

/* true int8 squared L2, sum (a_i - b_i)^2 in int32 (what :29-53 intended). */
int32_t oracle_l2_i8(const int8_t* a, const int8_t* b, size_t d);

/* int16 squared L2 as the reference computes it.  src/distance.h:14-27
 * (distance_compare_avx512f_i32): 16-bit wrapping subtract, mullo_epi16 keeps the low 16
 * bits of the square, madd_epi16 with 1 sign-extends and adds.  d multiple of 32. */
int32_t oracle_l2_i16_refcompat(const int16_t* a, const int16_t* b, size_t d);

/* int8 inner product, sum a_i*b_i in int32.  Not in the reference ("There
 * is no int8 inner-product in the reference"); plain integer arithmetic is the oracle. */
int32_t oracle_ip_i8(const int8_t* a, const int8_t* b, size_t d);

/* uint8-compressed squared L2.  src/antitopo_engine.h:38-61 (dist2_compressed) with the
 * query swizzle of :726-737: sum over i of ((uint32_t)q_i - row_i)^2 in wrapping 32-bit
 * arithmetic, returned as int.  q_i is truncated float->uint32 (defined for 0<=q_i<2^32).
 * d multiple of 64. */
int32_t oracle_l2_u8_compressed(const float* q, const uint8_t* row, size_t d);

/* ---- quantisers ------------------------------------------------------------------- */

/* src/quantizer.h:132-141 quantizer_simple<uint8_t>::build: plain cast, no scaling. */
void oracle_quantize_simple_u8(const float* rows, size_t n_values, uint8_t* out);

/* src/quantizer.h:213-232 quantizer_ranged_q8::build + :196-208 convert: global affine
 * int8.  Returns scale_factor and offset through the pointers.  max starts at
 * FLT_MIN (smallest positive normal), as the reference does (:217). */
void oracle_quantize_ranged_q8(const float* rows, size_t n, size_t d, int8_t* out,
                               float* scale_factor, float* offset);

/* ---- top-k ------------------------------------------------------------------------ */

typedef struct oracle_topk oracle_topk;
/* src/topk_t.h:9-67.  Bounded max-heap on (dist, id) pairs + set of known ids. */
oracle_topk* oracle_topk_create(size_t k);
void oracle_topk_destroy(oracle_topk* t);
/* :24-35 consider(d, v): admit iff id unseen and (size<k or top.first > d); evict the
 * lexicographic maximum when size>k.  Returns is_good. */
int oracle_topk_consider(oracle_topk* t, float d, uint64_t v);
/* :36-41 */
void oracle_topk_discard_until_size(oracle_topk* t, size_t goal);
size_t oracle_topk_size(const oracle_topk* t);
/* :42-44 (undefined on an empty heap, like the reference) */
uint64_t oracle_topk_worst(const oracle_topk* t);
float oracle_topk_worst_val(const oracle_topk* t);
int oracle_topk_at_capacity(const oracle_topk* t);
/* :45-66 to_combined_vector(): ascending (dist, id).  Returns the count written. */
size_t oracle_topk_to_combined(const oracle_topk* t, uint64_t* ids, float* dists);

/* ---- brute force ------------------------------------------------------------------ */

enum {
	ORACLE_METRIC_L2_F32 = 0,           /* oracle_l2_f32                              */
	ORACLE_METRIC_IP_F32 = 1,           /* -oracle_dot_f32 (largest dot first)        */
	ORACLE_METRIC_L2_I8 = 2,            /* oracle_l2_i8                               */
	ORACLE_METRIC_L2_I8_REFCOMPAT = 3,  /* oracle_l2_i8_refcompat                     */
	ORACLE_METRIC_IP_I8 = 4,            /* -oracle_ip_i8                              */
	ORACLE_METRIC_L2_U8 = 5,            /* oracle_l2_u8_compressed (fp32 query)       */
	ORACLE_METRIC_L2_I16_REFCOMPAT = 6  /* oracle_l2_i16_refcompat (int16 rows)       */
};

/* src/brute_force_engine.h:28-46 _query_k over a dense row-major base (the layout of
 * std::vector<vec<T>> with fixed DIM, src/vec.h:17-23): scan rows in increasing index,
 * admit iff size<k or top.first > d, pop the lexicographic max when size>k; output
 * ascending.  Result length min(k, n) is returned; dists may be NULL.
 * base/query element type follows the metric (float, int8_t, or uint8_t rows + float q). */
size_t oracle_brute_force_query_k(const void* base, size_t n, size_t d, const void* query,
                                  size_t k, int metric, uint64_t* ids, float* dists);

/* m queries, partitioned over n_threads POSIX threads against one shared read-only base
 * (n_threads = 1 is the reference's execution model, src/basic_bench.h:83-84). ids/dists
 * are [m][k], rows shorter than k are padded with UINT64_MAX / +inf. */
void oracle_brute_force_batch(const void* base, size_t n, size_t d, const void* queries,
                              size_t m, size_t k, int metric, int n_threads, uint64_t* ids,
                              float* dists);

/* src/quantizer.h:20-59 filter_by_score: for ids in order, d = score(id); keep (id, d)
 * with d < cutoff.  Returns the number kept. */
size_t oracle_filter_by_score(const void* base, size_t d, const void* query, int metric,
                              const uint64_t* ids, size_t n_ids, float cutoff,
                              uint64_t* kept_ids, float* kept_dists);

/* src/basic_bench.h:116-121,143 recall = #(ans ∩ expected) / (m*k). */
double oracle_recall(const uint64_t* ans, const uint64_t* expected, size_t m, size_t k);

/* ---- graph search (antitopo_engine, query side) ----------------------------------------
 * PARITY UNPINNED: src/antitopo_engine.h needs Eigen + nlohmann json and cannot be built
 * here.  The restatement follows the source line by line, including libstdc++'s
 * push_heap / pop_heap / make_heap element movement (the reference's priority queues compare
 * .first only, so equal distances come out in heap order). */
typedef struct oracle_graph oracle_graph;
/* Load an index in the reference's binary layout (src/antitopo_engine.h:994-1074). */
oracle_graph* oracle_graph_load(const char* path);
void oracle_graph_destroy(oracle_graph* g);
size_t oracle_graph_size(const oracle_graph* g);
size_t oracle_graph_dim(const oracle_graph* g);
const float* oracle_graph_vectors(const oracle_graph* g);
/* src/antitopo_engine.h:853-928 _query_k: greedy descent through the upper layers (:863-902),
 * then query_k_at_layer<true,false,false> (:495-708) or, with use_compression,
 * query_k_bottom_compressed (:710-851) over quantizer_simple<uint8_t> rows (:132-141) with the
 * final fp32 re-score (:845-848); truncated to k.  Returns the number of ids written; dists
 * receive the .first values of the returned pairs; *n_distcomps the RECORD_STATS counter. */
size_t oracle_graph_query_k(oracle_graph* g, const float* q, size_t k, size_t ef_search,
                            int use_compression, uint64_t* ids, float* dists,
                            uint64_t* n_distcomps);


/* Test hook: a trace of priority-queue operations through the heap code the graph search above
 * uses (libstdc++'s make_heap / push_heap / pop_heap as restated in expann_oracle_graph.c),
 * comparator on the distance only (src/antitopo_engine.h:540-545).  Pinned against the image's real
 * std::priority_queue by tests/golden/heap_ref.json (oracle/ref/heap_ref.cpp).  ops[i]: 1 = push
 * (op_d[i], op_id[i]), 0 = pop.  out_*[0] = state after construction from the init range,
 * out_*[i + 1] = state after op i; the drain arrays need n_init + n_ops slots.  Returns the
 * number of drained elements. */
size_t oracle_heap_trace(int max_heap, size_t n_init, const float* init_d, const uint64_t* init_id,
                         size_t n_ops, const int* ops, const float* op_d, const uint64_t* op_id,
                         uint64_t* out_size, float* out_top_d, uint64_t* out_top_id, float* drain_d,
                         uint64_t* drain_id);

#ifdef __cplusplus
}
#endif
#endif
