/*
 * expann_oracle_graph.c -- CPU restatement of the query side of the reference's graph engine.
 * TEST INFRASTRUCTURE ONLY (see expann_oracle.h).  Each function cites src/antitopo_engine.h.
 */
#include "expann_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct oracle_graph {
	size_t n, dim, starting_vertex, max_layer;
	float* vectors;        /* all_entries */
	uint8_t* compressed;   /* quantizer_simple<uint8_t>::stored, built lazily (:485-486) */
	size_t* n_layers;      /* per vertex */
	size_t** n_edges;      /* [v][layer] */
	uint64_t*** adj;       /* [v][layer][edge] = hadj_flat */
	char* visited;
	size_t* visited_recent;
	size_t n_recent;
};

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

/* src/antitopo_engine.h:994-1074 */
oracle_graph* oracle_graph_load(const char* path) {
	FILE* f = fopen(path, "rb");
	if (!f)
		return NULL;
	oracle_graph* g = (oracle_graph*)calloc(1, sizeof(*g));
	uint64_t sv, M, M0, efm, tmp, efc, oc, po, ml, n;
	uint8_t has_ef, uc, ul;
	float of, ob;
	int ok = rd(f, &sv, 8) && rd(f, &M, 8) && rd(f, &M0, 8) && rd(f, &efm, 8) && rd(f, &has_ef, 1);
	if (ok && has_ef)
		ok = rd(f, &tmp, 8);
	ok = ok && rd(f, &efc, 8) && rd(f, &oc, 8) && rd(f, &of, 4) && rd(f, &ob, 4) && rd(f, &po, 8) &&
	     rd(f, &uc, 1) && rd(f, &ul, 1) && rd(f, &ml, 8) && rd(f, &n, 8);
	if (!ok)
		goto fail;
	g->starting_vertex = sv;
	g->max_layer = ml;
	g->n = n;
	for (uint64_t i = 0; i < n; ++i) {
		uint64_t len;
		if (!rd(f, &len, 8))
			goto fail;
		if (i == 0) {
			g->dim = len;
			g->vectors = (float*)malloc(sizeof(float) * n * len);
		}
		if (len != g->dim || !rd(f, g->vectors + i * g->dim, 4 * len))
			goto fail;
	}
	uint64_t nv;
	if (!rd(f, &nv, 8) || nv != n)
		goto fail;
	g->n_layers = (size_t*)calloc(n, sizeof(size_t));
	g->n_edges = (size_t**)calloc(n, sizeof(size_t*));
	g->adj = (uint64_t***)calloc(n, sizeof(uint64_t**));
	for (uint64_t v = 0; v < n; ++v) {
		uint64_t nl;
		if (!rd(f, &nl, 8))
			goto fail;
		g->n_layers[v] = nl;
		g->n_edges[v] = (size_t*)calloc(nl ? nl : 1, sizeof(size_t));
		g->adj[v] = (uint64_t**)calloc(nl ? nl : 1, sizeof(uint64_t*));
		for (uint64_t l = 0; l < nl; ++l) {
			uint64_t ne;
			if (!rd(f, &ne, 8))
				goto fail;
			g->n_edges[v][l] = ne;
			g->adj[v][l] = (uint64_t*)malloc(sizeof(uint64_t) * (ne ? ne : 1));
			for (uint64_t e = 0; e < ne; ++e) {
				float len;
				if (!rd(f, &len, 4) || !rd(f, &g->adj[v][l][e], 8))
					goto fail;
			}
		}
	}
	fclose(f);
	g->visited = (char*)calloc(n, 1);
	g->visited_recent = (size_t*)malloc(sizeof(size_t) * (n + 1));
	return g;
fail:
	fclose(f);
	oracle_graph_destroy(g);
	return NULL;
}

void oracle_graph_destroy(oracle_graph* g) {
	if (!g)
		return;
	if (g->adj)
		for (size_t v = 0; v < g->n; ++v) {
			if (g->adj[v])
				for (size_t l = 0; l < g->n_layers[v]; ++l)
					free(g->adj[v][l]);
			free(g->adj[v]);
			if (g->n_edges)
				free(g->n_edges[v]);
		}
	free(g->adj);
	free(g->n_edges);
	free(g->n_layers);
	free(g->vectors);
	free(g->compressed);
	free(g->visited);
	free(g->visited_recent);
	free(g);
}
size_t oracle_graph_size(const oracle_graph* g) { return g->n; }
size_t oracle_graph_dim(const oracle_graph* g) { return g->dim; }
const float* oracle_graph_vectors(const oracle_graph* g) { return g->vectors; }

/* ---- libstdc++ heap algorithms on (dist, id) pairs, comparator on .first only ------------- */
typedef struct {
	float d;
	uint64_t id;
} md_t;
typedef struct {
	md_t* v;
	size_t n, cap;
	int max_heap; /* 1: worst_elem (a.first < b.first), top = largest; 0: best_elem, top = smallest */
} pq_t;
static inline int pq_less(const pq_t* h, md_t a, md_t b) {
	return h->max_heap ? (a.d < b.d) : (a.d > b.d);
}
static void pq_init(pq_t* h, int max_heap) {
	h->cap = 64;
	h->v = (md_t*)malloc(sizeof(md_t) * h->cap);
	h->n = 0;
	h->max_heap = max_heap;
}
static void pq_push_up(pq_t* h, size_t hole, size_t top, md_t value) { /* std::__push_heap */
	while (hole > top) {
		size_t parent = (hole - 1) / 2;
		if (!pq_less(h, h->v[parent], value))
			break;
		h->v[hole] = h->v[parent];
		hole = parent;
	}
	h->v[hole] = value;
}
static void pq_adjust(pq_t* h, size_t hole, size_t len, md_t value) { /* std::__adjust_heap */
	const size_t top = hole;
	size_t child = hole;
	while (len > 1 && child < (len - 1) / 2) {
		child = 2 * (child + 1);
		if (pq_less(h, h->v[child], h->v[child - 1]))
			--child;
		h->v[hole] = h->v[child];
		hole = child;
	}
	if ((len & 1) == 0 && len >= 2 && child == (len - 2) / 2) {
		child = 2 * (child + 1);
		h->v[hole] = h->v[child - 1];
		hole = child - 1;
	}
	pq_push_up(h, hole, top, value);
}
static void pq_push(pq_t* h, md_t e) {
	if (h->n == h->cap) {
		h->cap *= 2;
		h->v = (md_t*)realloc(h->v, sizeof(md_t) * h->cap);
	}
	h->v[h->n++] = e;
	pq_push_up(h, h->n - 1, 0, e);
}
static void pq_pop(pq_t* h) {
	if (h->n > 1) {
		md_t value = h->v[h->n - 1];
		h->v[h->n - 1] = h->v[0];
		pq_adjust(h, 0, h->n - 1, value);
	}
	h->n--;
}

static void pq_make(pq_t* h) { /* std::__make_heap (priority_queue's range constructor, :551-558) */
	const size_t len = h->n;
	if (len < 2)
		return;
	for (size_t parent = (len - 2) / 2;; --parent) {
		md_t value = h->v[parent];
		pq_adjust(h, parent, len, value);
		if (parent == 0)
			break;
	}
}

/* Test hook: run a trace of queue operations through pq_* (the heap code search_bottom uses) so
 * that tests/test_oracle.py can compare it with the same trace run through the image's real
 * libstdc++ (tests/golden/heap_ref.json, oracle/ref/heap_ref.cpp).  ops[i] 1 = push (op_d[i],
 * op_id[i]), 0 = pop.  Entry 0 of the out_* arrays is the state after construction from the
 * init range, entry i + 1 the state after op i (size 0: top fields 0).  Returns the drain length. */
size_t oracle_heap_trace(int max_heap, size_t n_init, const float* init_d, const uint64_t* init_id,
                         size_t n_ops, const int* ops, const float* op_d, const uint64_t* op_id,
                         uint64_t* out_size, float* out_top_d, uint64_t* out_top_id, float* drain_d,
                         uint64_t* drain_id) {
	pq_t h;
	pq_init(&h, max_heap);
	for (size_t i = 0; i < n_init; ++i) {
		if (h.n == h.cap) {
			h.cap *= 2;
			h.v = (md_t*)realloc(h.v, sizeof(md_t) * h.cap);
		}
		h.v[h.n].d = init_d[i];
		h.v[h.n].id = init_id[i];
		h.n++;
	}
	pq_make(&h);
	for (size_t i = 0;; ++i) {
		out_size[i] = h.n;
		out_top_d[i] = h.n ? h.v[0].d : 0.0f;
		out_top_id[i] = h.n ? h.v[0].id : 0;
		if (i == n_ops)
			break;
		if (ops[i] == 1) {
			md_t e = {op_d[i], op_id[i]};
			pq_push(&h, e);
		} else if (h.n) {
			pq_pop(&h);
		}
	}
	size_t nd = 0;
	while (h.n) {
		drain_d[nd] = h.v[0].d;
		drain_id[nd] = h.v[0].id;
		++nd;
		pq_pop(&h);
	}
	free(h.v);
	return nd;
}

/* src/antitopo_engine.h:25-37 (DIM % 128 == 0 -> src/distance.h:86-111) */
static inline float g_dist2(const oracle_graph* g, const float* a, const float* b) {
	return oracle_l2_f32(a, b, g->dim);
}

/* Bottom-layer best-first search shared by :495-708 (fp32) and :710-851 (uint8). */
static size_t search_bottom(oracle_graph* g, const float* q, uint64_t entry_point, size_t k,
                            int compressed, md_t* out, uint64_t* n_distcomps) {
	pq_t candidates, nearest;
	pq_init(&candidates, 0);
	pq_init(&nearest, 1);
#define SCORE(idx)                                                                               \
	(++*n_distcomps, compressed ? (float)oracle_l2_u8_compressed(q, g->compressed + (idx)*g->dim, g->dim) \
	                            : g_dist2(g, q, g->vectors + (idx)*g->dim))
	md_t e0 = {SCORE(entry_point), entry_point};
	pq_push(&candidates, e0); /* one entry point: make_heap of one element */
	pq_push(&nearest, e0);
	while (nearest.n > k)
		pq_pop(&nearest);
	g->visited[entry_point] = 1;
	g->n_recent = 0;
	g->visited_recent[g->n_recent++] = entry_point;
	size_t nl_cap = 256, nl_n;
	uint64_t* neighbour_list = (uint64_t*)malloc(sizeof(uint64_t) * nl_cap);
	while (candidates.n) {
		md_t cur = candidates.v[0];
		pq_pop(&candidates);
		if (cur.d > nearest.v[0].d && nearest.n == k) /* :588-590 / :774-776 */
			break;
		nl_n = 0;
		const size_t ne = g->n_layers[cur.id] ? g->n_edges[cur.id][0] : 0;
		for (size_t i = 0; i < ne; ++i) {
			uint64_t nb = g->adj[cur.id][0][i];
			if (!g->visited[nb]) {
				if (nl_n == nl_cap) {
					nl_cap *= 2;
					neighbour_list = (uint64_t*)realloc(neighbour_list, sizeof(uint64_t) * nl_cap);
				}
				neighbour_list[nl_n++] = nb;
				g->visited[nb] = 1;
				g->visited_recent[g->n_recent++] = nb;
			}
		}
		for (size_t i = 0; i < nl_n; ++i) { /* :636-689 / :795-835 */
			uint64_t next = neighbour_list[i];
			float dn = SCORE(next);
			if (nearest.n < k || dn < nearest.v[0].d) {
				md_t e = {dn, next};
				pq_push(&candidates, e);
				pq_push(&nearest, e);
				if (nearest.n > k)
					pq_pop(&nearest);
			}
		}
	}
#undef SCORE
	for (size_t i = 0; i < g->n_recent; ++i)
		g->visited[g->visited_recent[i]] = 0;
	g->n_recent = 0;
	size_t cnt = nearest.n;
	for (size_t i = cnt; i-- > 0;) { /* drain (worst first) then reverse */
		out[i] = nearest.v[0];
		pq_pop(&nearest);
	}
	free(neighbour_list);
	free(candidates.v);
	free(nearest.v);
	return cnt;
}

/* src/antitopo_engine.h:853-928 */
size_t oracle_graph_query_k(oracle_graph* g, const float* q, size_t k, size_t ef_search,
                            int use_compression, uint64_t* ids, float* dists,
                            uint64_t* n_distcomps) {
	uint64_t dc = 0;
	if (use_compression && !g->compressed) { /* :485-486 -> src/quantizer.h:132-141 */
		g->compressed = (uint8_t*)malloc(g->n * g->dim);
		oracle_quantize_simple_u8(g->vectors, g->n * g->dim, g->compressed);
	}
	uint64_t entry_point = g->starting_vertex;
	++dc;
	float ep_dist = g_dist2(g, g->vectors + entry_point * g->dim, q); /* :866-869 */
	for (size_t layer = g->max_layer - 1; layer > 0; --layer) {       /* :879-893 */
		int changed = 1;
		while (changed) {
			changed = 0;
			const uint64_t* nbrs = g->adj[entry_point][layer]; /* list bound at loop start */
			const size_t ne = g->n_edges[entry_point][layer];
			for (size_t i = 0; i < ne; ++i) {
				++dc;
				float nd = g_dist2(g, g->vectors + nbrs[i] * g->dim, q);
				if (nd < ep_dist) {
					entry_point = nbrs[i];
					ep_dist = nd;
					changed = 1;
				}
			}
		}
	}
	md_t* ret = (md_t*)malloc(sizeof(md_t) * (ef_search + 1));
	size_t cnt = search_bottom(g, q, entry_point, ef_search, use_compression, ret, &dc);
	if (use_compression) /* :845-848 final re-score, order kept */
		for (size_t i = 0; i < cnt; ++i)
			ret[i].d = g_dist2(g, g->vectors + ret[i].id * g->dim, q);
	if (cnt > k)
		cnt = k; /* :914-919 */
	for (size_t i = 0; i < cnt; ++i) {
		ids[i] = ret[i].id;
		if (dists)
			dists[i] = ret[i].d;
	}
	free(ret);
	if (n_distcomps)
		*n_distcomps = dc;
	return cnt;
}
