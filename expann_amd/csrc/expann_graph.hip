// expann_graph.hip -- C ABI of the graph path (expann_graph_*, expann_antitopo_*) and of the
// quantiser builds (expann_quantize_*): SURVEY 8 rows a-7..a-11, a-13, f-2, f-3.  The brute-force
// index lives in expann_hip.hip; both files share host_common.hpp.
#include "../../include/expann_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "graph_search.hpp"
#include "host_common.hpp"
#include "quantize.hpp"

using namespace expann;

struct expann_graph {
	int dim = 0, device = 0;
	size_t n = 0;
	uint32_t n_layers = 0, starting_vertex = 0, max_degree0 = 0;
	float* d_vectors = nullptr;
	uint8_t* d_compressed = nullptr;
	uint32_t* d_layer_off = nullptr;
	uint32_t* d_neighbours = nullptr;
	uint8_t* d_visited = nullptr;
	uint32_t* d_epochs = nullptr;
	uint32_t* d_error = nullptr;
	uint32_t slots = 0;
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	double last_ms = 0;
	mutable std::string err;
	int fail(int code, const std::string& msg) const {
		err = msg;
		return code;
	}
};

namespace {
using GraphFn = void (*)(GraphSearchParams);
struct GraphVariant {
	int d;
	bool compressed;
	GraphFn fn;
};
#define GRAPH_V(D) {D, false, graph_search_kernel<D, false>}, {D, true, graph_search_kernel<D, true>}
const GraphVariant kGraph[] = {GRAPH_V(64), GRAPH_V(128), GRAPH_V(256), GRAPH_V(832), GRAPH_V(960)};
#undef GRAPH_V
}  // namespace

extern "C" {

int expann_graph_create(int dim, int device, const float* vectors, size_t n, uint32_t n_layers,
                        uint32_t starting_vertex, const uint64_t* layer_offsets,
                        const uint32_t* neighbours, expann_graph** out) {
	if (!out) {
		g_create_error = "out == NULL";
		return EXPANN_ERR_INVALID_ARG;
	}
	*out = nullptr;
	if (!vectors || !layer_offsets || n == 0 || n_layers == 0 || starting_vertex >= n ||
	    n >= (1ull << 32) - 64 || (!neighbours && layer_offsets[(size_t)n_layers * (n + 1) - 1])) {
		g_create_error = "expann_graph_create: bad arguments";
		return EXPANN_ERR_INVALID_ARG;
	}
	// CSR offsets: start at 0, never decrease (a corrupt index file must not turn into
	// out-of-bounds reads on the device)
	if (layer_offsets[0] != 0) {
		g_create_error = "expann_graph_create: layer_offsets[0] != 0";
		return EXPANN_ERR_INVALID_ARG;
	}
	for (size_t i = 1; i < (size_t)n_layers * (n + 1); ++i)
		if (layer_offsets[i] < layer_offsets[i - 1]) {
			g_create_error = "expann_graph_create: layer_offsets decrease";
			return EXPANN_ERR_INVALID_ARG;
		}
	bool dim_ok = false;
	for (const auto& v : kGraph)
		dim_ok |= v.d == dim;
	if (!dim_ok) {
		g_create_error = "graph search is built for dim 64, 128, 256, 832, 960";
		return EXPANN_ERR_UNSUPPORTED;
	}
	const uint64_t n_edges = layer_offsets[(size_t)n_layers * (n + 1) - 1];
	if (n_edges >= (1ull << 32)) {
		g_create_error = "more than 2^32 edges";
		return EXPANN_ERR_UNSUPPORTED;
	}
	int ndev = expann_device_count();
	if (ndev <= 0) {
		g_create_error = "no HIP device visible: libexpann_hip has no CPU fallback";
		return EXPANN_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= ndev) {
		g_create_error = "device index out of range";
		return EXPANN_ERR_INVALID_ARG;
	}
	expann_graph* g = new expann_graph();
	g->dim = dim;
	g->device = device;
	g->n = n;
	g->n_layers = n_layers;
	g->starting_vertex = starting_vertex;
	std::vector<uint32_t> off32((size_t)n_layers * (n + 1));
	for (size_t i = 0; i < off32.size(); ++i)
		off32[i] = (uint32_t)layer_offsets[i];
	for (size_t v = 0; v < n; ++v)
		g->max_degree0 = std::max(g->max_degree0, off32[v + 1] - off32[v]);
	// the validity of every neighbour id is the caller's contract; check it once here so a
	// corrupt index cannot make the kernel read out of bounds
	for (uint64_t e = 0; e < n_edges; ++e)
		if (neighbours[e] >= n) {
			g_create_error = "neighbour id out of range";
			delete g;
			return EXPANN_ERR_INVALID_ARG;
		}
	auto bail = [&](const char* what) {
		g_create_error = std::string("expann_graph_create: ") + what;
		expann_graph_destroy(g);
		return EXPANN_ERR_HIP;
	};
	if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&g->stream) != hipSuccess)
		return bail("hipSetDevice/hipStreamCreate");
	if (hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess)
		return bail("hipEventCreate");
	const size_t vbytes = n * (size_t)dim * sizeof(float);
	if (hipMalloc(&g->d_vectors, vbytes) != hipSuccess ||
	    hipMalloc(&g->d_layer_off, off32.size() * sizeof(uint32_t)) != hipSuccess ||
	    hipMalloc(&g->d_neighbours, std::max<uint64_t>(n_edges, 1) * sizeof(uint32_t)) != hipSuccess ||
	    hipMalloc(&g->d_error, sizeof(uint32_t)) != hipSuccess)
		return bail("hipMalloc");
	if (hipMemcpy(g->d_vectors, vectors, vbytes, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(g->d_layer_off, off32.data(), off32.size() * sizeof(uint32_t),
	              hipMemcpyHostToDevice) != hipSuccess ||
	    (n_edges && hipMemcpy(g->d_neighbours, neighbours, n_edges * sizeof(uint32_t),
	                          hipMemcpyHostToDevice) != hipSuccess))
		return bail("hipMemcpy");
	// one visited array per resident workgroup; bounded to ~1/8 of a 288 GB HBM
	const int cus = num_cus(device);
	uint64_t slots = (uint64_t)cus * 8;
	while (slots > 64 && slots * n > (32ull << 30))
		slots /= 2;
	g->slots = (uint32_t)slots;
	if (hipMalloc(&g->d_visited, slots * n) != hipSuccess ||
	    hipMalloc(&g->d_epochs, slots * sizeof(uint32_t)) != hipSuccess)
		return bail("hipMalloc(visited)");
	if (hipMemset(g->d_visited, 0, slots * n) != hipSuccess ||
	    hipMemset(g->d_epochs, 0, slots * sizeof(uint32_t)) != hipSuccess)
		return bail("hipMemset");
	*out = g;
	return EXPANN_OK;
}

void expann_graph_destroy(expann_graph* g) {
	if (!g)
		return;
	hipSetDevice(g->device);
	if (g->stream) hipStreamSynchronize(g->stream);
	if (g->d_vectors) hipFree(g->d_vectors);
	if (g->d_compressed) hipFree(g->d_compressed);
	if (g->d_layer_off) hipFree(g->d_layer_off);
	if (g->d_neighbours) hipFree(g->d_neighbours);
	if (g->d_visited) hipFree(g->d_visited);
	if (g->d_epochs) hipFree(g->d_epochs);
	if (g->d_error) hipFree(g->d_error);
	if (g->ev0) hipEventDestroy(g->ev0);
	if (g->ev1) hipEventDestroy(g->ev1);
	if (g->stream) hipStreamDestroy(g->stream);
	delete g;
}

const char* expann_graph_last_error(const expann_graph* g) {
	return g ? g->err.c_str() : g_create_error.c_str();
}

double expann_graph_last_kernel_ms(const expann_graph* g) { return g ? g->last_ms : 0.0; }

int expann_graph_search(expann_graph* g, const float* queries, size_t m, size_t k,
                        size_t ef_search, int use_compression, uint64_t* ids, float* dists,
                        uint32_t* distcomps) {
	if (!g)
		return EXPANN_ERR_INVALID_ARG;
	if (k == 0 || ef_search == 0)
		return g->fail(EXPANN_ERR_INVALID_ARG, "k == 0 or ef_search == 0");
	if (m == 0)
		return EXPANN_OK;
	if (!queries || !ids || !dists)
		return g->fail(EXPANN_ERR_INVALID_ARG, "NULL pointer");
	if (ef_search > 4096)
		return g->fail(EXPANN_ERR_UNSUPPORTED, "ef_search > 4096");
	HIP_TRY(g, hipSetDevice(g->device));
	const GraphVariant* gv = nullptr;
	for (const auto& v : kGraph)
		if (v.d == g->dim && v.compressed == (use_compression != 0))
			gv = &v;
	if (!gv)
		return g->fail(EXPANN_ERR_UNSUPPORTED, "no graph kernel for this dim");
	if (use_compression && !g->d_compressed) {  // quantizer_simple<uint8_t>::build, :485-486
		HIP_TRY(g, hipMalloc(&g->d_compressed, g->n * (size_t)g->dim));
		const size_t nv = g->n * (size_t)g->dim;
		hipLaunchKernelGGL(quantize_simple_u8_kernel, dim3((uint32_t)((nv + kBlock - 1) / kBlock)),
		                   dim3(kBlock), 0, g->stream, (const float*)g->d_vectors, nv,
		                   g->d_compressed);
		HIP_TRY(g, hipGetLastError());
	}
	DevBuf b_q, b_ids, b_d, b_dc;
	const size_t qb = m * (size_t)g->dim * sizeof(float);
	HIP_TRY(g, b_q.alloc(qb));
	HIP_TRY(g, b_ids.alloc(sizeof(uint64_t) * m * k));
	HIP_TRY(g, b_d.alloc(sizeof(float) * m * k));
	HIP_TRY(g, b_dc.alloc(sizeof(uint32_t) * m));
	float* d_q = b_q.as<float>();
	uint64_t* d_ids = b_ids.as<uint64_t>();
	float* d_d = b_d.as<float>();
	uint32_t* d_dc = b_dc.as<uint32_t>();
	HIP_TRY(g, hipMemcpyAsync(d_q, queries, qb, hipMemcpyHostToDevice, g->stream));
	HIP_TRY(g, hipMemsetAsync(g->d_error, 0, sizeof(uint32_t), g->stream));
	uint32_t err_host = 0;
	uint32_t cand_cap = 256;
	while (cand_cap < 16 * ef_search && cand_cap < 8192)
		cand_cap *= 2;
	for (;;) {  // until no candidates heap overflows, or its LDS capacity limit (8192) is reached
		GraphSearchParams p{};
		p.vectors = g->d_vectors;
		p.compressed = g->d_compressed;
		p.layer_off = g->d_layer_off;
		p.neighbours = g->d_neighbours;
		p.n = (uint32_t)g->n;
		p.n_layers = g->n_layers;
		p.starting_vertex = g->starting_vertex;
		p.queries = d_q;
		p.m = (uint32_t)m;
		p.k = (uint32_t)k;
		p.ef = (uint32_t)ef_search;
		p.cand_cap = cand_cap;
		p.max_degree = g->max_degree0;
		p.list_cap = std::max<uint32_t>(std::max<uint32_t>(g->max_degree0, (uint32_t)ef_search), 4);
		p.visited = g->d_visited;
		p.epochs = g->d_epochs;
		p.out_ids = d_ids;
		p.out_dists = d_d;
		p.out_distcomps = d_dc;
		p.error = g->d_error;
		const size_t lds = sizeof(md_pair) * (p.ef + 1 + p.cand_cap + 1) +
		                   (sizeof(uint32_t) + sizeof(float)) * p.list_cap + 8 * sizeof(uint32_t);
		if (lds > 160 * 1024)
			return g->fail(EXPANN_ERR_UNSUPPORTED, "graph search working set exceeds LDS");
		HIP_TRY(g, hipFuncSetAttribute((const void*)gv->fn, hipFuncAttributeMaxDynamicSharedMemorySize,
		                               (int)lds));
		const uint32_t grid = (uint32_t)std::min<size_t>(m, g->slots);
		HIP_TRY(g, hipEventRecord(g->ev0, g->stream));
		hipLaunchKernelGGL(gv->fn, dim3(grid), dim3(64), lds, g->stream, p);
		HIP_TRY(g, hipEventRecord(g->ev1, g->stream));
		HIP_TRY(g, hipGetLastError());
		HIP_TRY(g, hipMemcpyAsync(&err_host, g->d_error, sizeof(uint32_t), hipMemcpyDeviceToHost,
		                          g->stream));
		HIP_TRY(g, hipStreamSynchronize(g->stream));
		float ms = 0;
		HIP_TRY(g, hipEventElapsedTime(&ms, g->ev0, g->ev1));
		g->last_ms = ms;
		if (!err_host || cand_cap >= 8192)
			break;
		cand_cap *= 4;  // a candidates heap overflowed: retry with a larger one
		if (cand_cap > 8192)
			cand_cap = 8192;
		HIP_TRY(g, hipMemsetAsync(g->d_error, 0, sizeof(uint32_t), g->stream));
	}
	HIP_TRY(g, hipMemcpy(ids, d_ids, sizeof(uint64_t) * m * k, hipMemcpyDeviceToHost));
	HIP_TRY(g, hipMemcpy(dists, d_d, sizeof(float) * m * k, hipMemcpyDeviceToHost));
	if (distcomps)
		HIP_TRY(g, hipMemcpy(distcomps, d_dc, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
	if (err_host)
		return g->fail(EXPANN_ERR_OVERFLOW, "graph search: candidates queue overflowed its LDS capacity");
	return EXPANN_OK;
}

// ---- the graph engine behind one handle (host build + device queries) --------------------
}  // extern "C"

#include "expann/gpu_antitopo_engine.h"

struct expann_antitopo {
	gpu_antitopo_engine<float>* eng = nullptr;
	int dim = 0;
	mutable std::string err;
	int fail(int code, const std::string& msg) const {
		err = msg;
		return code;
	}
};

#define ANTITOPO_TRY(e, body)                                              \
	try {                                                                  \
		body;                                                              \
	} catch (const std::exception& ex) {                                   \
		return (e)->fail(EXPANN_ERR_INVALID_ARG, ex.what());               \
	}

extern "C" {

int expann_antitopo_create(int dim, int device, size_t M, size_t ef_construction,
                           size_t ortho_count, size_t prune_overflow, int use_compression,
                           expann_antitopo** out) {
	if (!out || dim <= 0 || dim % 64 != 0 || M < 2 || ef_construction == 0) {
		g_create_error = "expann_antitopo_create: bad arguments (dim % 64 == 0, M >= 2)";
		return EXPANN_ERR_INVALID_ARG;
	}
	if (expann_device_count() <= 0) {
		g_create_error = "no HIP device visible: libexpann_hip has no CPU fallback";
		return EXPANN_ERR_NO_DEVICE;
	}
	gpu_antitopo_engine_config cfg(M, 2 * M, 1, ef_construction, ortho_count, 0.5f, 0.0f,
	                               prune_overflow, use_compression != 0);
	cfg.device = device;
	expann_antitopo* e = new expann_antitopo();
	e->dim = dim;
	e->eng = new gpu_antitopo_engine<float>(cfg);
	e->eng->index.dim = (size_t)dim;
	*out = e;
	return EXPANN_OK;
}

void expann_antitopo_destroy(expann_antitopo* e) {
	if (!e)
		return;
	delete e->eng;
	delete e;
}

const char* expann_antitopo_last_error(const expann_antitopo* e) {
	return e ? e->err.c_str() : g_create_error.c_str();
}

int expann_antitopo_store(expann_antitopo* e, const float* rows, size_t n) {
	if (!e || (!rows && n))
		return EXPANN_ERR_INVALID_ARG;
	ANTITOPO_TRY(e, for (size_t i = 0; i < n; ++i) e->eng->index.insert(rows + i * (size_t)e->dim));
	return EXPANN_OK;
}

int expann_antitopo_build(expann_antitopo* e) {
	if (!e)
		return EXPANN_ERR_INVALID_ARG;
	ANTITOPO_TRY(e, e->eng->_build());
	return EXPANN_OK;
}

int expann_antitopo_set_ef_search(expann_antitopo* e, size_t ef_search) {
	if (!e || ef_search == 0)
		return EXPANN_ERR_INVALID_ARG;
	e->eng->set_ef_search(ef_search);
	return EXPANN_OK;
}

int expann_antitopo_query(expann_antitopo* e, const float* queries, size_t m, size_t k,
                          uint64_t* ids, float* dists) {
	if (!e || !queries || !ids || !dists || k == 0)
		return e ? e->fail(EXPANN_ERR_INVALID_ARG, "bad arguments") : EXPANN_ERR_INVALID_ARG;
	ANTITOPO_TRY(e, e->eng->query_k_batch(queries, m, k, ids, dists));
	return EXPANN_OK;
}

int expann_antitopo_save(expann_antitopo* e, const char* index_path) {
	if (!e || !index_path)
		return EXPANN_ERR_INVALID_ARG;
	ANTITOPO_TRY(e, e->eng->index.write_index(index_path));
	return EXPANN_OK;
}

int expann_antitopo_load(expann_antitopo* e, const char* index_path) {
	if (!e || !index_path)
		return EXPANN_ERR_INVALID_ARG;
	ANTITOPO_TRY(e, {
		e->eng->index.read_index(index_path);
		if ((int)e->eng->index.dim != e->dim)
			throw std::runtime_error("index dimension differs from the engine's");
		e->eng->upload();
	});
	return EXPANN_OK;
}

size_t expann_antitopo_size(const expann_antitopo* e) { return e ? e->eng->index.size() : 0; }
uint64_t expann_antitopo_num_distcomps(const expann_antitopo* e) {
	return e ? e->eng->num_distcomps : 0;
}

int expann_quantize_simple_u8_device(int device, const float* d_rows, size_t n_values,
                                     uint8_t* d_out, void* stream) {
	if (!d_rows || !d_out) {
		g_create_error = "expann_quantize_simple_u8_device: NULL pointer";
		return EXPANN_ERR_INVALID_ARG;
	}
	if (n_values == 0)
		return EXPANN_OK;
	if (hipSetDevice(device) != hipSuccess) {
		g_create_error = "hipSetDevice failed";
		return EXPANN_ERR_HIP;
	}
	hipLaunchKernelGGL(quantize_simple_u8_kernel, dim3((uint32_t)((n_values + kBlock - 1) / kBlock)),
	                   dim3(kBlock), 0, (hipStream_t)stream, d_rows, n_values, d_out);
	if (hipGetLastError() != hipSuccess) {
		g_create_error = "quantize_simple_u8_kernel launch failed";
		return EXPANN_ERR_HIP;
	}
	return EXPANN_OK;
}

int expann_quantize_ranged_q8_device(int device, const float* d_rows, size_t n_values,
                                     int8_t* d_out, float* d_scale_offset, void* stream) {
	if (!d_rows || !d_out || !d_scale_offset || n_values == 0) {
		g_create_error = "expann_quantize_ranged_q8_device: bad arguments";
		return EXPANN_ERR_INVALID_ARG;
	}
	if (hipSetDevice(device) != hipSuccess) {
		g_create_error = "hipSetDevice failed";
		return EXPANN_ERR_HIP;
	}
	hipStream_t st = (hipStream_t)stream;
	uint32_t* d_mm = nullptr;
	if (hipMalloc(&d_mm, 2 * sizeof(uint32_t)) != hipSuccess) {
		g_create_error = "hipMalloc failed";
		return EXPANN_ERR_HIP;
	}
	// min starts at FLT_MAX, max at FLT_MIN (smallest positive normal): src/quantizer.h:217-218
	const uint32_t init[2] = {float_to_ordered(3.402823466e+38f), float_to_ordered(1.175494351e-38f)};
	hipError_t e = hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);  // `init` is a stack buffer
	if (e == hipSuccess) {
		const uint32_t blocks = (uint32_t)std::min<size_t>((n_values + kBlock - 1) / kBlock, 4096);
		hipLaunchKernelGGL(minmax_f32_kernel, dim3(blocks), dim3(kBlock), 0, st, d_rows, n_values, d_mm);
		hipLaunchKernelGGL(quantize_ranged_q8_kernel, dim3((uint32_t)((n_values + kBlock - 1) / kBlock)),
		                   dim3(kBlock), 0, st, d_rows, n_values, (const uint32_t*)d_mm, d_out,
		                   d_scale_offset);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipStreamSynchronize(st);
	}
	hipFree(d_mm);
	if (e != hipSuccess) {
		g_create_error = std::string("expann_quantize_ranged_q8_device: ") + hipGetErrorString(e);
		return EXPANN_ERR_HIP;
	}
	return EXPANN_OK;
}

}  // extern "C"
