/*
 * expann_hip.h -- C ABI of the MI355X-native distance + top-k engine (libexpann_hip.so).
 *
 * This is the drop-in boundary for expANN's hot path: the entry points are what a C++
 * engine class satisfying the reference's CRTP `ann_engine<T, Derived>` interface
 * (upstream src/ann_engine.h:16-29) binds to.  Plain pointers and sizes only; no C++
 * types, no exceptions, no torch types cross this boundary.  Every call returns an int
 * status (EXPANN_OK == 0); the message of the last failure on a handle is available from
 * expann_last_error().  A handle is single-caller; distinct handles may be used from
 * different host threads concurrently (the reference runs one private engine per job
 * thread, src/bench_runner.h:30-57,78-87).
 *
 * There is no CPU fallback behind this ABI: without a usable HIP device every compute
 * entry point fails with EXPANN_ERR_NO_DEVICE.
 */
#ifndef EXPANN_HIP_H
#define EXPANN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: expann_profile grew (deferred_searches); expann_sharded_* gained the device-resident in-process
 * search, the all-to-all exchange pattern and its caller-transport hook */
#define EXPANN_ABI_VERSION 2

enum expann_status {
	EXPANN_OK = 0,
	EXPANN_ERR_INVALID_ARG = 1,
	EXPANN_ERR_NO_DEVICE = 2,
	EXPANN_ERR_HIP = 3,
	EXPANN_ERR_NOT_BUILT = 4,
	EXPANN_ERR_UNSUPPORTED = 5,
	EXPANN_ERR_OVERFLOW = 6 /* internal candidate buffers exhausted even after retries */
};

/* element type of the stored rows */
enum expann_dtype {
	EXPANN_DTYPE_F32 = 0, /* vec<float> rows, src/vec.h:17-23                         */
	EXPANN_DTYPE_U8 = 1,  /* quantizer_simple<uint8_t> rows, src/quantizer.h:127      */
	EXPANN_DTYPE_I8 = 2,  /* quantizer_ranged_q8 rows, src/quantizer.h:152-238        */
	EXPANN_DTYPE_I16 = 3  /* int16 rows scored by src/distance.h:14-27 bit for bit
	                       * (distance_compare_avx512f_i32: 16-bit wrapping arithmetic,
	                       * exact only while |a_i - b_i| <= 181); EXPANN_METRIC_L2 only,
	                       * dim 64 or 128; queries int16                            */
};

/* how a (query, row) pair is scored; smaller score = nearer */
enum expann_metric {
	EXPANN_METRIC_L2 = 0, /* f32: squared L2 in the lane order of src/distance.h:136-147;
	                         u8: src/antitopo_engine.h:38-61 (dist2_compressed);
	                         i8: true sum (a-b)^2                                      */
	EXPANN_METRIC_IP = 1, /* score = -dot (src/distance.h:181-190 for f32); largest
	                         inner product first, ties by lower id                      */
	EXPANN_METRIC_L2_I8_REFCOMPAT = 2 /* i8 only: src/distance.h:29-53 bit for bit,
	                         including its zero-extension of the wrapped difference     */
};

typedef struct expann_index expann_index;

/* library / device ------------------------------------------------------------------ */
int expann_abi_version(void);
/* number of visible HIP devices (0 when there is none or the runtime is unusable). */
int expann_device_count(void);

/* lifecycle (replaces: engine construction, src/bench_runner.h:33) -------------------- */
int expann_create(int dim, int dtype, int metric, int device, expann_index** out);
void expann_destroy(expann_index* h);
/* message of the last error on h (or of the last failed expann_create when h == NULL). */
const char* expann_last_error(const expann_index* h);

/* store_vector x n (src/ann_engine.h:23-25, src/brute_force_engine.h:20-22): rows are
 * COPIED into host staging; the caller keeps ownership.  Row i of the j-th call gets id
 * (rows stored so far) + i, i.e. insertion order, like all_entries.push_back. */
int expann_add(expann_index* h, const void* rows, size_t n);

/* build() (src/ann_engine.h:26, src/brute_force_engine.h:24-26): upload the staged rows
 * to HBM.  Fails with EXPANN_ERR_INVALID_ARG on an empty index (the reference asserts). */
int expann_build(expann_index* h);

/* Adopt rows that already live in device memory (no copy; the caller keeps them alive
 * until expann_destroy).  id_offset is added to every returned id: rank r of a sharded
 * job passes the global number of its first row.  Replaces add+build. */
int expann_set_base_device(expann_index* h, const void* d_rows, size_t n, uint64_t id_offset);

size_t expann_size(const expann_index* h);

/* query_k for a batch (src/ann_engine.h:27-29, src/brute_force_engine.h:28-46; the
 * reference has no batch API -- m = 1 is query_k).  Host buffers.  For each query the k
 * rows with the smallest (score, id), ascending; ids[m][k], dists[m][k] (dists may be
 * NULL).  When fewer than k rows exist the tail is padded with UINT64_MAX / +inf. */
int expann_search(expann_index* h, const void* queries, size_t m, size_t k, uint64_t* ids,
                  float* dists);

/* same with device-resident queries and outputs, enqueued on `stream` (a hipStream_t;
 * NULL = the index's own stream).  The call returns after the stream has drained: the one host
 * synchronisation of a search is the check that no internal candidate list overflowed (the
 * search is re-run with larger lists / the tie-exact kernel if one did). */
int expann_search_device(expann_index* h, const void* d_queries, size_t m, size_t k,
                         uint64_t* d_ids, float* d_dists, void* stream);

/* Deferred check (option "async_search" = 1, device-pointer searches only): expann_search_device
 * enqueues the whole search and returns WITHOUT waiting, so the caller can enqueue what follows
 * (the next batch, the RCCL exchange of a sharded job) while the GPU still scans.  expann_sync
 * waits for the stream of the last such search and reports on all searches since the previous
 * sync: EXPANN_OK, or EXPANN_ERR_OVERFLOW when one of them would have needed the synchronous
 * retry (overflowed candidate lists, queries outside the fp16 range of the index): its results
 * are then invalid and the caller repeats it with "async_search" = 0.  At most 256 searches may
 * be outstanding (further ones wait by themselves).
 * ONE stream at a time: all searches of a handle share one workspace, so a search enqueued on a
 * different stream than the outstanding deferred ones first waits (on the host) until those have
 * drained; deferred searches overlap with each other only in stream order. */
int expann_sync(expann_index* h);

/* k-way merge of per-shard results after an all-gather (RCCL): in_ids/in_dists are
 * [n_lists][m][k], each row ascending by (score, id) and padded as above; out is [m][k]. */
int expann_merge_topk_device(int device, const uint64_t* d_in_ids, const float* d_in_dists,
                             size_t n_lists, size_t m, size_t k, uint64_t* d_out_ids,
                             float* d_out_dists, void* stream);

/* the same with explicit distances (in elements) between the lists: list g is at
 * d_in_ids + g*ids_stride and d_in_dists + g*dists_stride.  For all-gathered chunks that hold
 * [ids | dists] of one rank each (one collective per exchange instead of two). */
int expann_merge_topk_strided_device(int device, const uint64_t* d_in_ids, const float* d_in_dists,
                                     size_t ids_stride, size_t dists_stride, size_t n_lists, size_t m,
                                     size_t k, uint64_t* d_out_ids, float* d_out_dists, void* stream);

/* row-sharded brute force over the GPUs of one node (SURVEY 8b "create(..., devices, n_dev, ...)",
 * 8e; csrc/expann_sharded.hip) -------------------------------------------------------------------
 * Shard r holds the contiguous rows [r * ceil(N/G), min(N, (r+1) * ceil(N/G))) and searches ALL
 * queries on its own device; the per-shard results [m][k] of (score, global id) are exchanged over
 * RCCL (xGMI) and merged in the reference's (score, id) order, so ids and distances are bit-identical
 * to the single-device index (contiguous ranges + global ids).  Two exchange patterns (option
 * "exchange_pattern"): 2 (default) = an all-to-all of QUERY SLICES -- rank j receives, from every
 * shard, the results of the queries [j * ceil(m/G), ...) and merges those (m/G queries, G lists);
 * the rank form then all-gathers the merged slices; 1 = ONE ncclAllGather of every shard's whole
 * [ids m*k u64 | dists m*k f32] chunk, every rank merges all m queries.  The reference has no
 * counterpart (one engine, one thread: src/basic_bench.h:83-84); the calls mirror the
 * single-device ones above (engine construction src/bench_runner.h:33, store_vector / build /
 * query_k src/ann_engine.h:23-29). */
typedef struct expann_sharded expann_sharded;
/* In-process form: ONE handle drives n_dev devices of this node (one stream and one enqueue thread
 * per device, ncclCommInitAll at build).  devices[] may name a device more than once (several shards
 * on one GPU; the exchange then runs as device copies, RCCL refuses duplicate devices). */
int expann_sharded_create(int dim, int dtype, int metric, const int* devices, int n_dev,
                          expann_sharded** out);
/* One-process-per-GPU form: rank `rank` of `world` ranks on `device`; id128 = the 128 bytes rank 0
 * got from expann_sharded_unique_id (ncclGetUniqueId), distributed by the launcher.  Collective:
 * every rank calls it (ncclCommInitRank).  id128 == NULL: no RCCL communicator -- one rank needs
 * none, more ranks exchange through expann_sharded_set_exchange_fn / _set_alltoallv_fn. */
int expann_sharded_unique_id(void* id128);
int expann_sharded_create_rank(int dim, int dtype, int metric, int device, int rank, int world,
                               const void* id128, expann_sharded** out);
void expann_sharded_destroy(expann_sharded* h);
const char* expann_sharded_last_error(const expann_sharded* h);
/* in-process form: store_vector x n into host staging, then build() cuts the rows into the ranges
 * above and uploads each to its device (as expann_add / expann_build). */
int expann_sharded_add(expann_sharded* h, const void* rows, size_t n);
int expann_sharded_build(expann_sharded* h);
/* Adopt rows already in the shard's device memory (as expann_set_base_device): the rank form's
 * only way to receive rows (shard = 0, id_offset = global number of the rank's first row; n = 0
 * with d_rows = NULL declares the rank's range empty -- the ceil partition leaves trailing ranks
 * without rows when N < G * (G-1) -- and the rank still takes part in every exchange);
 * in-process form: shards in order 0, 1, ... instead of add + build. */
int expann_sharded_set_shard_device(expann_sharded* h, int shard, const void* d_rows, size_t n,
                                    uint64_t id_offset);
size_t expann_sharded_size(const expann_sharded* h);  /* rows over all local shards */
int expann_sharded_shards(const expann_sharded* h);   /* shards in use (rank form: world) */
int expann_sharded_exchange(const expann_sharded* h); /* transport: 0 none (one shard), 1 RCCL, 2 device
                                                         copies, 3 the caller's function */
int expann_sharded_exchange_pattern(const expann_sharded* h); /* 0 none, 1 all-gather of chunks, 2 all-to-all
                                                                 of query slices */
/* ranks of the RCCL communicator as RCCL reports them (ncclCommCount); 0 = no communicator */
int expann_sharded_comm_ranks(const expann_sharded* h);
/* host time the last in-process search spent enqueuing (scan + exchange + merge of all shards, up to
 * the point where the host starts waiting), milliseconds */
double expann_sharded_last_enqueue_ms(const expann_sharded* h);
/* Rank form: the exchange through the caller's transport instead of RCCL (another fabric; ranks
 * that share one GPU, which RCCL refuses; tests).  expann_exchange_fn (pattern 1) gathers `bytes`
 * bytes of device memory d_send from every rank into d_recv (rank r's chunk at r * bytes);
 * expann_alltoallv_fn (pattern 2) sends send_bytes[j] bytes at d_send + send_off[j] to rank j and
 * receives recv_bytes[j] bytes from rank j at d_recv + recv_off[j] (arrays of `world` entries; entry
 * [rank] is always 0 bytes: the library copies what stays).  Both: ordered after the work already
 * on `stream`, complete -- as far as `stream` is concerned -- on return or in stream order; 0 = ok.
 * NULL goes back to the communicator.  With both set, "exchange_pattern" chooses. */
typedef int (*expann_exchange_fn)(void* ctx, const void* d_send, void* d_recv, size_t bytes, int rank,
                                  int world, void* stream);
typedef int (*expann_alltoallv_fn)(void* ctx, const void* d_send, const size_t* send_off,
                                   const size_t* send_bytes, void* d_recv, const size_t* recv_off,
                                   const size_t* recv_bytes, int rank, int world, void* stream);
int expann_sharded_set_exchange_fn(expann_sharded* h, expann_exchange_fn fn, void* ctx);
int expann_sharded_set_alltoallv_fn(expann_sharded* h, expann_alltoallv_fn fn, void* ctx);
/* in-process form, host buffers: as expann_search. */
int expann_sharded_search(expann_sharded* h, const void* queries, size_t m, size_t k, uint64_t* ids,
                          float* dists);
/* in-process form, everything resident: d_queries[r] = the m queries in the memory of shard r's
 * device; shard r merges the query slice expann_sharded_slice(h, m, r, &lo, &hi) and leaves it in
 * d_ids[r][(hi-lo)][k] / d_dists[r] on its device (ordered on the shard's own stream).  With
 * "async_search" = 1 (default) the call returns once everything is enqueued on every device;
 * expann_sharded_sync waits for all shards and validates the searches since the last sync
 * (EXPANN_ERR_OVERFLOW: repeat them with "async_search" = 0). */
int expann_sharded_search_devices(expann_sharded* h, const void* const* d_queries, size_t m, size_t k,
                                  uint64_t* const* d_ids, float* const* d_dists);
/* query slice [*q_lo, *q_hi) that shard (rank) `shard` merges under pattern 2 and in
 * expann_sharded_search_devices: [min(m, shard * ceil(m/G)), min(m, (shard+1) * ceil(m/G))) */
int expann_sharded_slice(const expann_sharded* h, size_t m, int shard, size_t* q_lo, size_t* q_hi);
/* rank form, device buffers on `stream` (NULL = the handle's own): local search, exchange, merge;
 * every rank ends with the full ids[m][k] / dists[m][k].  Collective.  With the option
 * "async_search" = 1 the call returns without a host wait (expann_sharded_sync, as expann_sync). */
int expann_sharded_search_device(expann_sharded* h, const void* d_queries, size_t m, size_t k,
                                 uint64_t* d_ids, float* d_dists, void* stream);
int expann_sharded_sync(expann_sharded* h);
/* "exchange" (in-process form, transport: 0 auto, 1 RCCL, 2 device copies), "exchange_pattern" (0 auto
 * = 2, 1 all-gather of whole chunks, 2 all-to-all of query slices), "threads" (in-process form: 1
 * (default) = one enqueue thread per shard, 0 = the calling thread enqueues every device in turn),
 * "async_search" (in-process form: 1 (default) = the shards' searches are enqueued on all devices
 * before the host waits for any); every other option goes to the shards' indexes (expann_set_option). */
int expann_sharded_set_option(expann_sharded* h, const char* name, long value);

/* batched candidate scoring (quantized_scorer::filter_by_score, src/quantizer.h:20-59):
 * for each of n_ids row ids (order kept) score against ONE query; keep (id, score) with
 * score < cutoff.  Host buffers; *n_kept receives the count. */
int expann_score_ids(expann_index* h, const void* query, const uint64_t* ids, size_t n_ids,
                     float cutoff, uint64_t* kept_ids, float* kept_scores, size_t* n_kept);

/* graph search (antitopo_engine, query side) --------------------------------------------- */
typedef struct expann_graph expann_graph;
/* Upload a built graph (replaces the tail of antitopo_engine::_build, src/antitopo_engine.h:
 * 467-493, after deserialize :994-1074): `vectors` [n][dim] fp32 (all_entries), CSR adjacency
 * per layer as flattened by include/expann/antitopo_index.h (hadj_flat): layer_offsets is
 * [n_layers][n+1] into `neighbours`; layer 0 is hadj_bottom.  Host arrays, copied. */
int expann_graph_create(int dim, int device, const float* vectors, size_t n, uint32_t n_layers,
                        uint32_t starting_vertex, const uint64_t* layer_offsets,
                        const uint32_t* neighbours, expann_graph** out);
void expann_graph_destroy(expann_graph* g);
const char* expann_graph_last_error(const expann_graph* g);
/* antitopo_engine::_query_k for a batch (src/antitopo_engine.h:853-928): greedy descent, then
 * the bottom-layer best-first search with queue size ef_search (:495-708), or with
 * use_compression != 0 over uint8 rows + final fp32 re-score (:710-851).  Host buffers:
 * ids[m][k] / dists[m][k] padded with UINT64_MAX / +inf; distcomps[m] (RECORD_STATS'
 * num_distcomps per query, :125-129) may be NULL. */
int expann_graph_search(expann_graph* g, const float* queries, size_t m, size_t k,
                        size_t ef_search, int use_compression, uint64_t* ids, float* dists,
                        uint32_t* distcomps);
/* device time of the last expann_graph_search's traversal kernel, milliseconds */
double expann_graph_last_kernel_ms(const expann_graph* g);

/* GPU-assisted batched construction of the graph (csrc/graph_build.hpp; replaces the inner loop of
 * antitopo_engine::_store_vector / prune_edges, src/antitopo_engine.h:263-465, for the vectors
 * [n_built, n) -- the first n_built come with their rows already built, by the serial host builder
 * of include/expann/antitopo_index.h).  Vectors are inserted in BATCHES against the graph of the
 * vectors before them: ef_construction searches per layer, prune_edges' rule on the candidate lists,
 * reverse edges, and one more prune of every row that outgrew M / M0.  `levels[v]` = the level the
 * reference's draw gives vertex v (:323).  Adjacency arrays are host memory, fixed row strides:
 * layer 0 ids0 / d0 [n][stride0] + deg0[n]; layers 1.. idsu / du [(l-1) * U + upper_idx[v]][strideu] +
 * degu, upper_idx[v] = -1 for level-0 vertices.  On return rows hold at most M0 / M edges (id,
 * reference-order distance).  stats[4] (optional): batches, reverse edges dropped for want of slack,
 * rows re-pruned, 0.  ortho_count = 1 only (the reference's sweep, src/bench_runner.h:138). */
int expann_graph_build_batched(int dim, int device, const float* vectors, size_t n, const uint8_t* levels,
                               size_t n_built, uint32_t* max_layer_io, uint32_t* starting_vertex_io, size_t M,
                               size_t M0, size_t ef_construction, size_t prune_overflow, float ortho_factor,
                               float ortho_bias, size_t max_batch, uint32_t* ids0, float* d0, uint32_t* deg0,
                               size_t stride0, const int32_t* upper_idx, size_t U, size_t n_upper_layers,
                               uint32_t* idsu, float* du, uint32_t* degu, size_t strideu, uint64_t* stats);

/* the whole graph engine behind one handle (what src/pyrunner.cpp:56-90 binds: ctor,
 * store_vector / store_many_vectors, build, query_k, set_ef_search).  Construction runs on the
 * host (include/expann/antitopo_index.h), queries on the GPU (expann_graph_search). */
typedef struct expann_antitopo expann_antitopo;
/* antitopo_engine(M, ef_construction, ortho_count, prune_overflow, use_compression),
 * src/antitopo_engine.h:157-166: M0 = 2M, ortho_factor = 0.5, ortho_bias = 0, ef_search_mult = 1 */
int expann_antitopo_create(int dim, int device, size_t M, size_t ef_construction,
                           size_t ortho_count, size_t prune_overflow, int use_compression,
                           expann_antitopo** out);
void expann_antitopo_destroy(expann_antitopo* e);
const char* expann_antitopo_last_error(const expann_antitopo* e);
int expann_antitopo_store(expann_antitopo* e, const float* rows, size_t n);  /* _store_vector x n */
/* the same rows through the batched GPU builder (expann_graph_build_batched); the first
 * min(n_serial, n) of an empty engine are still inserted serially and seed the batches */
int expann_antitopo_store_batched(expann_antitopo* e, const float* rows, size_t n, size_t n_serial);
int expann_antitopo_build(expann_antitopo* e);                               /* _build (:467-493) */
int expann_antitopo_set_ef_search(expann_antitopo* e, size_t ef_search);     /* :189-195 */
/* query_k for a batch; ef_search defaults to k * ef_search_mult and is sticky (:858-859) */
int expann_antitopo_query(expann_antitopo* e, const float* queries, size_t m, size_t k,
                          uint64_t* ids, float* dists);
int expann_antitopo_save(expann_antitopo* e, const char* index_path);  /* serialize, :932-991 */
int expann_antitopo_load(expann_antitopo* e, const char* index_path);  /* deserialize + upload */
size_t expann_antitopo_size(const expann_antitopo* e);
uint64_t expann_antitopo_num_distcomps(const expann_antitopo* e);

/* test hook: replays a trace of priority-queue operations through the device's wave-cooperative heap code
 * (csrc/graph_search.hpp coop_push / coop_pop, the walk's queues) on device 0.  Trace format of
 * tests/golden/heap_ref.json: ops[i] 1 = push (op_d[i], op_id[i]), 0 = pop; entry 0 of the
 * out_* arrays is the state after the range constructor, entry i + 1 the state after op i; returns the
 * drain length, (size_t)-1 on failure.  tests/test_heap_pin.py checks it against the traces the image's
 * real std::priority_queue answered (tests/golden/heap_ref.json). */
size_t expann_device_heap_trace(int max_heap, size_t n_init, const float* init_d, const uint64_t* init_id,
                                size_t n_ops, const int* ops, const float* op_d, const uint64_t* op_id,
                                uint64_t* out_size, float* out_top_d, uint64_t* out_top_id, float* drain_d,
                                uint64_t* drain_id);

/* quantiser builds on device buffers (src/quantizer.h) -------------------------------- */
/* quantizer_simple<uint8_t>::build (src/quantizer.h:132-141): out[i] = uint8_t(in[i]), no
 * scaling; defined for 0 <= in[i] < 256.  Asynchronous on `stream`. */
int expann_quantize_simple_u8_device(int device, const float* d_rows, size_t n_values,
                                     uint8_t* d_out, void* stream);
/* quantizer_ranged_q8::build (src/quantizer.h:213-232, :196-200): global affine int8 in
 * [0,127]; d_scale_offset[0..1] receive scale_factor and offset.  Synchronises `stream`. */
int expann_quantize_ranged_q8_device(int device, const float* d_rows, size_t n_values,
                                     int8_t* d_out, float* d_scale_offset, void* stream);

/* profiling ------------------------------------------------------------------------- */
typedef struct expann_profile {
	uint64_t scan_launches;   /* launches of the full-base scan kernel since reset      */
	double scan_ms;           /* their summed device time (HIP events on their stream)  */
	uint64_t scan_rows;       /* rows streamed by those launches (per launch: n)        */
	uint64_t scan_query_tiles;/* query tiles (passes over the base) by those launches   */
	uint32_t query_tile;      /* Q_t: queries per pass of the last launch               */
	uint32_t levels;          /* threshold levels of the last search                    */
	uint64_t candidates;      /* candidates kept by the last full scan (all queries)    */
	uint64_t retries;         /* overflow retries since reset                           */
	char scan_kernel[64];     /* name of the full-scan kernel last launched             */
	uint64_t deferred_searches; /* searches since reset whose flag check was deferred to
	                             * expann_sync ("async_search"); the others waited on the host */
} expann_profile;
/* enable != 0 brackets every full-scan launch with HIP events. */
int expann_set_profiling(expann_index* h, int enable);
/* synchronises the recorded events, fills *out and resets the accumulators. */
int expann_get_profile(expann_index* h, expann_profile* out);
/* the same for every shard / for one shard of a sharded handle */
int expann_sharded_set_profiling(expann_sharded* h, int enable);
int expann_sharded_get_profile(expann_sharded* h, int shard, expann_profile* out);

/* integer options: "query_tile" (0 = auto), "cand_capacity" (0 = auto),
 * "scan_kernel" (0 = auto, 1 = direct VALU scan, 2 = GEMM form on the fp32 / int8 matrix
 * cores, 3 = GEMM form on the bf16 matrix cores with the 3-term split, 4 = GEMM form with one
 * scaled fp16 product, 5 = 8-bit rows: int8 MFMA form with per-wave hit queues (d = 128,
 * 256, 768, 832, 960); the final ids and distances are identical for every choice),
 * "sample_ratio" (rows ratio between the levels of the threshold ladder, default 32),
 * "sample_pass" (fp16 form: 1 = one sampled pass gives the threshold (default), 0 = ladder),
 * "sample_frac" (the sampled pass reads 1/frac of the rows; 0 = chosen from k (default)),
 * "u8_exact" (1 (default): an fp32 L2 index of dim 128 / 256 whose values are all integers in
 * [0, 255] keeps a uint8 copy, and batches of 8-bit integer queries are searched through the
 * exact 8-bit kernels -- same ids and fp32 distances; 0: never),
 * "latency_mode" (1 (default): expann_search with few queries (m*k <= 16384) stages them in
 * pinned memory and lets the select kernels store the results there -- one host sync per
 * search and no pageable copies; 0: always the plain copy path),
 * "async_search" (1: expann_search_device returns without the final host wait, see expann_sync),
 * "xcd_tolerance" (percent of modelled launch cost given up for an XCD-aligned row-chunk count,
 * default 3), "scan_chunks" (experiments: force the row-chunk count of the fp16 scan; 0 = model),
 * "sample_run", "debug" (bench / ablation switches),
 * "persist" (1 (default): the d = 128 fp16 scan runs as resident workgroups pulling work per XCD; 0: plain launch),
 * "ip_rescale" (1 (default): inner product, fp16 form: the filter sees each query times a power of two that brings
 * its norm to the largest row's -- ranks unchanged, results exact; 0: rounds 1-2's unscaled filter). */
int expann_set_option(expann_index* h, const char* name, long value);

#ifdef __cplusplus
}
#endif
#endif
