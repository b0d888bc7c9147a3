/*
 * expann_oracle.c -- CPU restatement of expANN's distance + top-k hot path in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see expann_oracle.h).  Never linked into the product.
 * Each function cites the reference file:line it follows.
 */
#include "expann_oracle.h"

#include <assert.h>
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * _mm512_reduce_add_ps as GCC 11 defines it (avx512fintrin.h): upper 256 + lower 256,
 * then upper 128 + lower 128, then lanes {2,3} + {0,1}, then lane 0 + lane 1.
 * ---------------------------------------------------------------------------------- */
static inline float reduce_add_16(const float acc[16]) {
	float t8[8], t4[4], t2[2];
	for (int l = 0; l < 8; ++l)
		t8[l] = acc[l + 8] + acc[l];
	for (int l = 0; l < 4; ++l)
		t4[l] = t8[l + 4] + t8[l];
	for (int l = 0; l < 2; ++l)
		t2[l] = t4[l] + t4[l + 2];
	return t2[0] + t2[1];
}

/* src/distance.h:136-147 */
float oracle_l2_f32(const float* a, const float* b, size_t d) {
	assert(d % 16 == 0);
	float acc[16];
	for (int l = 0; l < 16; ++l)
		acc[l] = 0.0f;
	for (size_t i = 0; i < d; i += 16)
		for (int l = 0; l < 16; ++l) {
			float diff = a[i + l] - b[i + l];
			acc[l] = fmaf(diff, diff, acc[l]);
		}
	return reduce_add_16(acc);
}

/* src/distance.h:181-190 */
float oracle_dot_f32(const float* a, const float* b, size_t d) {
	assert(d % 16 == 0);
	float acc[16];
	for (int l = 0; l < 16; ++l)
		acc[l] = 0.0f;
	for (size_t i = 0; i < d; i += 16)
		for (int l = 0; l < 16; ++l)
			acc[l] = fmaf(a[i + l], b[i + l], acc[l]);
	return reduce_add_16(acc);
}

/* src/distance.h:29-53: sub_epi8 wraps, unpack with zero zero-extends, madd squares.
 * All accumulation is 32-bit wrapping, so the lane partition does not matter. */
int32_t oracle_l2_i8_refcompat(const int8_t* a, const int8_t* b, size_t d) {
	assert(d % 64 == 0);
	uint32_t sum = 0;
	for (size_t i = 0; i < d; ++i) {
		uint8_t diff = (uint8_t)((uint8_t)a[i] - (uint8_t)b[i]);
		sum += (uint32_t)diff * (uint32_t)diff;
	}
	return (int32_t)sum;
}

int32_t oracle_l2_i8(const int8_t* a, const int8_t* b, size_t d) {
	uint32_t sum = 0;
	for (size_t i = 0; i < d; ++i) {
		int32_t diff = (int32_t)a[i] - (int32_t)b[i];
		sum += (uint32_t)(diff * diff);
	}
	return (int32_t)sum;
}

/* src/distance.h:14-27 */
int32_t oracle_l2_i16_refcompat(const int16_t* a, const int16_t* b, size_t d) {
	assert(d % 32 == 0);
	uint32_t sum = 0;
	for (size_t i = 0; i < d; ++i) {
		uint16_t diff = (uint16_t)((uint16_t)a[i] - (uint16_t)b[i]);
		int16_t sdiff = (int16_t)diff;
		/* mullo_epi16: low 16 bits of the signed product */
		uint16_t sq = (uint16_t)((int32_t)sdiff * (int32_t)sdiff);
		/* madd_epi16(1, sq): sign-extend to 32 bits */
		sum += (uint32_t)(int32_t)(int16_t)sq;
	}
	return (int32_t)sum;
}

int32_t oracle_ip_i8(const int8_t* a, const int8_t* b, size_t d) {
	uint32_t sum = 0;
	for (size_t i = 0; i < d; ++i)
		sum += (uint32_t)((int32_t)a[i] * (int32_t)b[i]);
	return (int32_t)sum;
}

/* src/antitopo_engine.h:38-61 with the swizzle of :726-737 undone: lane w of pass p
 * sees uint32(q[64b+4w+p]) against byte p of dword w of the row, i.e. dimension
 * 64b+4w+p on both sides. */
int32_t oracle_l2_u8_compressed(const float* q, const uint8_t* row, size_t d) {
	assert(d % 64 == 0);
	uint32_t sum = 0;
	for (size_t i = 0; i < d; ++i) {
		uint32_t qi = (uint32_t)q[i];
		uint32_t diff = qi - (uint32_t)row[i];
		sum += diff * diff;
	}
	return (int32_t)sum;
}

/* src/quantizer.h:132-141 */
void oracle_quantize_simple_u8(const float* rows, size_t n_values, uint8_t* out) {
	for (size_t i = 0; i < n_values; ++i)
		out[i] = (uint8_t)rows[i];
}

/* src/quantizer.h:213-232, :196-208 */
void oracle_quantize_ranged_q8(const float* rows, size_t n, size_t d, int8_t* out,
                               float* scale_factor, float* offset) {
	const size_t q_min = 0, q_max = 127;           /* :189-190 */
	float max_val = FLT_MIN, min_val = FLT_MAX;    /* :217-218 */
	for (size_t i = 0; i < n * d; ++i) {
		if (rows[i] > max_val)
			max_val = rows[i];
		if (rows[i] < min_val)
			min_val = rows[i];
	}
	float scale = (float)(q_max - q_min + 1) / (max_val - min_val); /* :226 */
	float off = -scale * min_val - (float)q_min;                    /* :227 */
	for (size_t i = 0; i < n * d; ++i) {
		/* :196-200; negative rounded values are UB in the reference (float -> size_t),
		 * the oracle clamps them to q_min. */
		float r = roundf(scale * rows[i] + off);
		size_t v = r <= 0.0f ? 0 : (size_t)r;
		if (v < q_min)
			v = q_min;
		if (v > q_max)
			v = q_max;
		out[i] = (int8_t)v;
	}
	*scale_factor = scale;
	*offset = off;
}

/* ------------------------------------------------------------------------------------
 * max-heap of (dist, id) pairs ordered like std::pair<float, size_t>::operator<
 * (what std::priority_queue<std::pair<T,size_t>> uses, src/topk_t.h:11,
 * src/brute_force_engine.h:29).
 * ---------------------------------------------------------------------------------- */
typedef struct {
	float d;
	uint64_t id;
} pair_t;

static inline int pair_less(pair_t a, pair_t b) {
	return a.d < b.d || (!(b.d < a.d) && a.id < b.id);
}

typedef struct {
	pair_t* v;
	size_t n, cap;
} heap_t;

static void heap_init(heap_t* h, size_t cap) {
	h->v = (pair_t*)malloc(sizeof(pair_t) * (cap + 2));
	h->n = 0;
	h->cap = cap + 2;
}
static void heap_push(heap_t* h, pair_t p) {
	if (h->n == h->cap) {
		h->cap *= 2;
		h->v = (pair_t*)realloc(h->v, sizeof(pair_t) * h->cap);
	}
	size_t i = h->n++;
	while (i > 0) {
		size_t par = (i - 1) / 2;
		if (!pair_less(h->v[par], p))
			break;
		h->v[i] = h->v[par];
		i = par;
	}
	h->v[i] = p;
}
static void heap_pop(heap_t* h) {
	pair_t last = h->v[--h->n];
	size_t i = 0;
	for (;;) {
		size_t c = 2 * i + 1;
		if (c >= h->n)
			break;
		if (c + 1 < h->n && pair_less(h->v[c], h->v[c + 1]))
			++c;
		if (!pair_less(last, h->v[c]))
			break;
		h->v[i] = h->v[c];
		i = c;
	}
	if (h->n)
		h->v[i] = last;
}

/* ---- topk_t ------------------------------------------------------------------------ */
struct oracle_topk {
	heap_t h;
	size_t k;
	/* known-id set (src/topk_t.h:18); a linear scan is fine for an oracle */
	uint64_t* known;
	size_t n_known, cap_known;
};

oracle_topk* oracle_topk_create(size_t k) {
	oracle_topk* t = (oracle_topk*)calloc(1, sizeof(*t));
	heap_init(&t->h, k);
	t->k = k;
	t->cap_known = k + 2;
	t->known = (uint64_t*)malloc(sizeof(uint64_t) * t->cap_known);
	return t;
}
void oracle_topk_destroy(oracle_topk* t) {
	if (!t)
		return;
	free(t->h.v);
	free(t->known);
	free(t);
}
static int known_contains(const oracle_topk* t, uint64_t v) {
	for (size_t i = 0; i < t->n_known; ++i)
		if (t->known[i] == v)
			return 1;
	return 0;
}
static void known_erase(oracle_topk* t, uint64_t v) {
	for (size_t i = 0; i < t->n_known; ++i)
		if (t->known[i] == v) {
			t->known[i] = t->known[--t->n_known];
			return;
		}
}
/* src/topk_t.h:24-35 */
int oracle_topk_consider(oracle_topk* t, float d, uint64_t v) {
	int is_good = !known_contains(t, v) && (t->h.n < t->k || t->h.v[0].d > d);
	if (is_good) {
		pair_t p = {d, v};
		heap_push(&t->h, p);
		if (t->n_known == t->cap_known) {
			t->cap_known *= 2;
			t->known = (uint64_t*)realloc(t->known, sizeof(uint64_t) * t->cap_known);
		}
		t->known[t->n_known++] = v;
	}
	if (t->h.n > t->k) {
		known_erase(t, t->h.v[0].id);
		heap_pop(&t->h);
	}
	return is_good;
}
/* src/topk_t.h:36-41 */
void oracle_topk_discard_until_size(oracle_topk* t, size_t goal) {
	while (t->h.n > goal) {
		known_erase(t, t->h.v[0].id);
		heap_pop(&t->h);
	}
}
size_t oracle_topk_size(const oracle_topk* t) { return t->h.n; }
uint64_t oracle_topk_worst(const oracle_topk* t) { return t->h.v[0].id; }
float oracle_topk_worst_val(const oracle_topk* t) { return t->h.v[0].d; }
int oracle_topk_at_capacity(const oracle_topk* t) { return t->h.n == t->k; }

static size_t heap_drain_ascending(const heap_t* src, uint64_t* ids, float* dists) {
	heap_t h;
	heap_init(&h, src->n);
	memcpy(h.v, src->v, sizeof(pair_t) * src->n);
	h.n = src->n;
	size_t n = h.n;
	for (size_t i = n; i-- > 0;) { /* pop max first, fill from the back = reverse() */
		if (ids)
			ids[i] = h.v[0].id;
		if (dists)
			dists[i] = h.v[0].d;
		heap_pop(&h);
	}
	free(h.v);
	return n;
}
/* src/topk_t.h:45-66 */
size_t oracle_topk_to_combined(const oracle_topk* t, uint64_t* ids, float* dists) {
	return heap_drain_ascending(&t->h, ids, dists);
}

/* ---- brute force ------------------------------------------------------------------ */
static size_t elem_size(int metric) {
	switch (metric) {
	case ORACLE_METRIC_L2_F32:
	case ORACLE_METRIC_IP_F32:
		return 4;
	case ORACLE_METRIC_L2_I16_REFCOMPAT:
		return 2;
	default:
		return 1;
	}
}

static inline float score_row(const void* base, size_t row, size_t d, const void* q,
                              int metric) {
	switch (metric) {
	case ORACLE_METRIC_L2_F32:
		return oracle_l2_f32((const float*)q, (const float*)base + row * d, d);
	case ORACLE_METRIC_IP_F32:
		return -oracle_dot_f32((const float*)q, (const float*)base + row * d, d);
	case ORACLE_METRIC_L2_I8:
		return (float)oracle_l2_i8((const int8_t*)q, (const int8_t*)base + row * d, d);
	case ORACLE_METRIC_L2_I8_REFCOMPAT:
		return (float)oracle_l2_i8_refcompat((const int8_t*)q,
		                                     (const int8_t*)base + row * d, d);
	case ORACLE_METRIC_IP_I8:
		return -(float)oracle_ip_i8((const int8_t*)q, (const int8_t*)base + row * d, d);
	case ORACLE_METRIC_L2_U8:
		return (float)oracle_l2_u8_compressed((const float*)q,
		                                      (const uint8_t*)base + row * d, d);
	case ORACLE_METRIC_L2_I16_REFCOMPAT:
		return (float)oracle_l2_i16_refcompat((const int16_t*)q, (const int16_t*)base + row * d, d);
	}
	return NAN;
}

/* src/brute_force_engine.h:28-46 */
size_t oracle_brute_force_query_k(const void* base, size_t n, size_t d, const void* query,
                                  size_t k, int metric, uint64_t* ids, float* dists) {
	heap_t top_k;
	heap_init(&top_k, k);
	for (size_t i = 0; i < n; ++i) {
		float dist = score_row(base, i, d, query, metric);
		if (top_k.n < k || top_k.v[0].d > dist) { /* :33 */
			pair_t p = {dist, (uint64_t)i};
			heap_push(&top_k, p);
		}
		if (top_k.n > k) /* :36-37 */
			heap_pop(&top_k);
	}
	size_t cnt = heap_drain_ascending(&top_k, ids, dists); /* :39-45 */
	free(top_k.v);
	return cnt;
}

typedef struct {
	const void* base;
	size_t n, d;
	const char* queries;
	size_t q_stride, q_begin, q_end, k;
	int metric;
	uint64_t* ids;
	float* dists;
} bf_job;

static void* bf_worker(void* arg) {
	bf_job* j = (bf_job*)arg;
	for (size_t q = j->q_begin; q < j->q_end; ++q) {
		uint64_t* ids = j->ids + q * j->k;
		float* dists = j->dists ? j->dists + q * j->k : NULL;
		size_t cnt = oracle_brute_force_query_k(j->base, j->n, j->d,
		                                        j->queries + q * j->q_stride, j->k,
		                                        j->metric, ids, dists);
		for (size_t i = cnt; i < j->k; ++i) {
			ids[i] = UINT64_MAX;
			if (dists)
				dists[i] = INFINITY;
		}
	}
	return NULL;
}

void oracle_brute_force_batch(const void* base, size_t n, size_t d, const void* queries,
                              size_t m, size_t k, int metric, int n_threads, uint64_t* ids,
                              float* dists) {
	if (n_threads < 1)
		n_threads = 1;
	if ((size_t)n_threads > m)
		n_threads = m ? (int)m : 1;
	size_t q_elem = (metric == ORACLE_METRIC_L2_U8) ? 4 : elem_size(metric);
	bf_job* jobs = (bf_job*)calloc((size_t)n_threads, sizeof(bf_job));
	pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
	for (int t = 0; t < n_threads; ++t) {
		bf_job j = {base, n, d, (const char*)queries, d * q_elem,
		            m * (size_t)t / (size_t)n_threads,
		            m * (size_t)(t + 1) / (size_t)n_threads, k, metric, ids, dists};
		jobs[t] = j;
		if (n_threads == 1)
			bf_worker(&jobs[t]);
		else
			pthread_create(&th[t], NULL, bf_worker, &jobs[t]);
	}
	if (n_threads > 1)
		for (int t = 0; t < n_threads; ++t)
			pthread_join(th[t], NULL);
	free(jobs);
	free(th);
}

/* src/quantizer.h:20-59 (the prefetching only changes timing, not values) */
size_t oracle_filter_by_score(const void* base, size_t d, const void* query, int metric,
                              const uint64_t* ids, size_t n_ids, float cutoff,
                              uint64_t* kept_ids, float* kept_dists) {
	size_t kept = 0;
	for (size_t i = 0; i < n_ids; ++i) {
		float s = score_row(base, ids[i], d, query, metric);
		if (s < cutoff) {
			kept_ids[kept] = ids[i];
			kept_dists[kept] = s;
			++kept;
		}
	}
	return kept;
}

/* src/basic_bench.h:116-121,143 */
double oracle_recall(const uint64_t* ans, const uint64_t* expected, size_t m, size_t k) {
	size_t found = 0;
	for (size_t q = 0; q < m; ++q)
		for (size_t i = 0; i < k; ++i) {
			uint64_t e = expected[q * k + i];
			if (e == UINT64_MAX)
				continue;
			for (size_t j = 0; j < k; ++j)
				if (ans[q * k + j] == e) {
					++found;
					break;
				}
		}
	return (double)found / (double)(m * k);
}
