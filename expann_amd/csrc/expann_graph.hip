// expann_graph.hip -- C ABI of the graph path (expann_graph_*, expann_antitopo_*) and of the
// quantiser builds (expann_quantize_*): SURVEY 8 rows a-7..a-11, a-13, f-2, f-3.  The brute-force
// index lives in expann_hip.hip; both files share host_common.hpp.
#include "../../include/expann_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "graph_build.hpp"
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
	uint32_t* d_adj0 = nullptr;   // [n][stride0] layer-0 lists at a fixed stride, padded with UINT32_MAX
	uint32_t stride0 = 0;
	uint8_t* d_visited = nullptr;  // epoch bytes [slots][n], or -- vis_words != 0 -- bitsets [slots][vis_words]
	uint32_t vis_words = 0;
	uint32_t* d_epochs = nullptr;
	uint32_t* d_error = nullptr;   // [2]: overflow flag, query counter of the launch
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
const GraphVariant kGraph[] = {GRAPH_V(64),  GRAPH_V(128), GRAPH_V(256), GRAPH_V(512),
                               GRAPH_V(768), GRAPH_V(832), GRAPH_V(960)};
#undef GRAPH_V
// the instrumented instance (EXPANN_GRAPH_STAMPS=1, d = 128): per-phase shader clocks of a hop
const GraphVariant kGraphDbg[] = {{128, false, graph_search_kernel<128, false, 1>}, {128, true, graph_search_kernel<128, true, 1>}};
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
		g_create_error = "graph search is built for dim 64, 128, 256, 512, 768, 832, 960";
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
	    hipMalloc(&g->d_error, 2 * sizeof(uint32_t)) != hipSuccess)
		return bail("hipMalloc");
	if (hipMemcpy(g->d_vectors, vectors, vbytes, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(g->d_layer_off, off32.data(), off32.size() * sizeof(uint32_t),
	              hipMemcpyHostToDevice) != hipSuccess ||
	    (n_edges && hipMemcpy(g->d_neighbours, neighbours, n_edges * sizeof(uint32_t),
	                          hipMemcpyHostToDevice) != hipSuccess))
		return bail("hipMemcpy");
	{
		// layer 0 once more at a fixed stride: a hop then reads its list at cur * stride without first
		// fetching the row's offsets (one dependent HBM round trip less per hop)
		g->stride0 = std::max<uint32_t>(4, (g->max_degree0 + 3) / 4 * 4);
		std::vector<uint32_t> adj((size_t)n * g->stride0, 0xFFFFFFFFu);
		for (size_t v = 0; v < n; ++v)
			std::copy(neighbours + off32[v], neighbours + off32[v + 1], adj.begin() + v * g->stride0);
		if (hipMalloc(&g->d_adj0, adj.size() * sizeof(uint32_t)) != hipSuccess)
			return bail("hipMalloc(adj0)");
		if (hipMemcpy(g->d_adj0, adj.data(), adj.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
			return bail("hipMemcpy(adj0)");
	}
	// one visited array per workgroup of the (persistent) search grid; bounded to ~1/8 of a 288 GB
	// HBM.  The walk is a chain of dependent memory round trips: what hides them is waves per CU
	// (the kernel needs ~60 VGPRs and ~10 KB of LDS at ef = 60: 16 workgroups fit a CU).
	const int cus = num_cus(device);
	uint64_t slots = (uint64_t)cus * 16;
	// visited sets: bitsets (cleared after every query) up to 8 M vertices, epoch bytes beyond
	g->vis_words = n <= (8u << 20) ? (uint32_t)((n + 8191) / 8192 * 256) : 0u;
	if (std::getenv("EXPANN_GRAPH_VISITED_BYTES"))  // (A/B switch: round 2's epoch bytes at any size)
		g->vis_words = 0;
	const uint64_t per_slot = g->vis_words ? (uint64_t)g->vis_words * 4 : (uint64_t)n;
	while (slots > 64 && slots * per_slot > (32ull << 30))
		slots /= 2;
	g->slots = (uint32_t)slots;
	if (hipMalloc(&g->d_visited, slots * per_slot) != hipSuccess ||
	    hipMalloc(&g->d_epochs, slots * sizeof(uint32_t)) != hipSuccess)
		return bail("hipMalloc(visited)");
	if (hipMemset(g->d_visited, 0, slots * per_slot) != hipSuccess ||
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
	if (g->d_adj0) hipFree(g->d_adj0);
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
	const bool stamps = std::getenv("EXPANN_GRAPH_STAMPS") != nullptr && g->dim == 128;
	if (stamps)
		gv = &kGraphDbg[use_compression ? 1 : 0];
	DevBuf b_stamps;
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
	HIP_TRY(g, hipMemsetAsync(g->d_error, 0, 2 * sizeof(uint32_t), g->stream));
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
		p.adj0 = g->d_adj0;
		p.stride0 = g->stride0;
		p.n = (uint32_t)g->n;
		p.n_layers = g->n_layers;
		p.starting_vertex = g->starting_vertex;
		p.queries = d_q;
		p.m = (uint32_t)m;
		p.k = (uint32_t)k;
		p.ef = (uint32_t)ef_search;
		p.cand_cap = cand_cap;
		p.max_degree = g->max_degree0;
		p.list_cap = std::max<uint32_t>(std::max<uint32_t>(g->stride0, (uint32_t)ef_search), 4);
		p.visited = g->d_visited;
		p.vis_bits = reinterpret_cast<uint32_t*>(g->d_visited);
		p.vis_words = g->vis_words;
		p.epochs = g->d_epochs;
		p.out_ids = d_ids;
		p.out_dists = d_d;
		p.out_distcomps = d_dc;
		p.error = g->d_error;
		p.next_query = g->d_error + 1;
		if (const char* e = std::getenv("EXPANN_GRAPH_DEBUG"))
			p.debug = (uint32_t)std::atol(e);
		if (stamps) {
			if (!b_stamps.p)
				HIP_TRY(g, b_stamps.alloc(sizeof(unsigned long long) * 8 * g->slots));
			HIP_TRY(g, hipMemsetAsync(b_stamps.p, 0, sizeof(unsigned long long) * 8 * g->slots, g->stream));
			p.stamps = b_stamps.as<unsigned long long>();
		}
		const size_t lds = sizeof(md_pair) * (p.ef + 1 + p.cand_cap + 1) +
		                   (sizeof(uint32_t) + sizeof(float)) * p.list_cap;
		if (lds > 160 * 1024)
			return g->fail(EXPANN_ERR_UNSUPPORTED, "graph search working set exceeds LDS");
		HIP_TRY(g, hipFuncSetAttribute((const void*)gv->fn, hipFuncAttributeMaxDynamicSharedMemorySize,
		                               (int)lds));
		// as many workgroups as are resident at once (registers and LDS of this instance), each with a
		// visited array of its own; they pull queries from a counter
		int per_cu = 0;
		HIP_TRY(g, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gv->fn, 64, lds));
		const uint32_t resident = (uint32_t)std::max(1, per_cu) * (uint32_t)num_cus(g->device);
		const uint32_t grid = (uint32_t)std::min<size_t>(m, std::min<uint32_t>(g->slots, resident));
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
		if (stamps) {  // mean over the workgroups of the grid, per hop
			std::vector<unsigned long long> st(8 * (size_t)grid);
			HIP_TRY(g, hipMemcpy(st.data(), b_stamps.p, st.size() * 8, hipMemcpyDeviceToHost));
			double tot[8] = {0};
			for (uint32_t b = 0; b < grid; ++b)
				for (int i = 0; i < 8; ++i)
					tot[i] += (double)st[8 * (size_t)b + i];
			const double hops = tot[6] > 0 ? tot[6] : 1, all = tot[0] + tot[1] + tot[2] + tot[3] + tot[4] + tot[5];
			std::fprintf(stderr,
			             "graph_search<%d,%s> ef %zu: %.3f ms, %u workgroups, %.1f hops / query, %.1f queue insertions / hop; "
			             "shader clocks per hop: pop + broadcast %.0f, adjacency + visited %.0f, gathers + scoring %.0f, "
			             "serial queue update %.0f; per query: setup + descent %.0f, output %.0f; shares %.1f / %.1f / %.1f / "
			             "%.1f / %.1f / %.1f %%\n",
			             g->dim, use_compression ? "u8" : "f32", ef_search, ms, grid, tot[6] / (double)m, tot[7] / hops,
			             tot[1] / hops, tot[2] / hops, tot[3] / hops, tot[4] / hops, tot[0] / (double)m, tot[5] / (double)m,
			             100 * tot[1] / all, 100 * tot[2] / all, 100 * tot[3] / all, 100 * tot[4] / all, 100 * tot[0] / all,
			             100 * tot[5] / all);
		}
		if (!err_host || cand_cap >= 8192)
			break;
		cand_cap *= 4;  // a candidates heap overflowed: retry with a larger one
		if (cand_cap > 8192)
			cand_cap = 8192;
		HIP_TRY(g, hipMemsetAsync(g->d_error, 0, 2 * sizeof(uint32_t), g->stream));
	}
	HIP_TRY(g, hipMemcpy(ids, d_ids, sizeof(uint64_t) * m * k, hipMemcpyDeviceToHost));
	HIP_TRY(g, hipMemcpy(dists, d_d, sizeof(float) * m * k, hipMemcpyDeviceToHost));
	if (distcomps)
		HIP_TRY(g, hipMemcpy(distcomps, d_dc, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
	if (err_host)
		return g->fail(EXPANN_ERR_OVERFLOW, "graph search: candidates queue overflowed its LDS capacity");
	return EXPANN_OK;
}

// test hook: a queue trace through the device's wave-cooperative heap code (tests/test_heap_pin.py);
// the signature of oracle_heap_trace / std_heap_trace, ids must fit 32 bits
size_t expann_device_heap_trace(int max_heap, size_t n_init, const float* init_d, const uint64_t* init_id, size_t n_ops,
                                const int* ops, const float* op_d, const uint64_t* op_id, uint64_t* out_size,
                                float* out_top_d, uint64_t* out_top_id, float* drain_d, uint64_t* drain_id) {
	const size_t cap = n_init + n_ops + 1;
	if (cap * sizeof(md_pair) > (60u << 10) || hipSetDevice(0) != hipSuccess)
		return (size_t)-1;
	std::vector<uint32_t> id32(n_init), opid32(n_ops);
	for (size_t i = 0; i < n_init; ++i)
		id32[i] = (uint32_t)init_id[i];
	for (size_t i = 0; i < n_ops; ++i)
		opid32[i] = (uint32_t)op_id[i];
	DevBuf b_id, b_ii, b_ops, b_od, b_oi, b_sz, b_td, b_ti, b_dd, b_di, b_n;
	if (b_id.alloc(4 * n_init) != hipSuccess || b_ii.alloc(4 * n_init) != hipSuccess || b_ops.alloc(4 * n_ops) != hipSuccess ||
	    b_od.alloc(4 * n_ops) != hipSuccess || b_oi.alloc(4 * n_ops) != hipSuccess || b_sz.alloc(4 * (n_ops + 1)) != hipSuccess ||
	    b_td.alloc(4 * (n_ops + 1)) != hipSuccess || b_ti.alloc(4 * (n_ops + 1)) != hipSuccess ||
	    b_dd.alloc(4 * cap) != hipSuccess || b_di.alloc(4 * cap) != hipSuccess || b_n.alloc(4) != hipSuccess)
		return (size_t)-1;
	hipMemcpy(b_id.p, init_d, 4 * n_init, hipMemcpyHostToDevice);
	hipMemcpy(b_ii.p, id32.data(), 4 * n_init, hipMemcpyHostToDevice);
	hipMemcpy(b_ops.p, ops, 4 * n_ops, hipMemcpyHostToDevice);
	hipMemcpy(b_od.p, op_d, 4 * n_ops, hipMemcpyHostToDevice);
	hipMemcpy(b_oi.p, opid32.data(), 4 * n_ops, hipMemcpyHostToDevice);
	HeapTraceParams p{max_heap, (uint32_t)n_init, (uint32_t)n_ops, b_id.as<float>(), b_ii.as<uint32_t>(), b_ops.as<int>(),
	                  b_od.as<float>(), b_oi.as<uint32_t>(), b_sz.as<uint32_t>(), b_td.as<float>(), b_ti.as<uint32_t>(),
	                  b_dd.as<float>(), b_di.as<uint32_t>(), b_n.as<uint32_t>(), 0u};
	if (const char* e = std::getenv("EXPANN_GRAPH_DEBUG"))
		p.serial = (uint32_t)std::atol(e) & 1u;
	hipLaunchKernelGGL(heap_trace_kernel, dim3(1), dim3(64), cap * sizeof(md_pair), nullptr, p);
	if (hipDeviceSynchronize() != hipSuccess)
		return (size_t)-1;
	std::vector<uint32_t> sz(n_ops + 1), ti(n_ops + 1), di(cap);
	uint32_t nd = 0;
	hipMemcpy(sz.data(), b_sz.p, 4 * (n_ops + 1), hipMemcpyDeviceToHost);
	hipMemcpy(out_top_d, b_td.p, 4 * (n_ops + 1), hipMemcpyDeviceToHost);
	hipMemcpy(ti.data(), b_ti.p, 4 * (n_ops + 1), hipMemcpyDeviceToHost);
	hipMemcpy(&nd, b_n.p, 4, hipMemcpyDeviceToHost);
	hipMemcpy(drain_d, b_dd.p, 4 * (size_t)nd, hipMemcpyDeviceToHost);
	hipMemcpy(di.data(), b_di.p, 4 * (size_t)nd, hipMemcpyDeviceToHost);
	for (size_t i = 0; i <= n_ops; ++i) {
		out_size[i] = sz[i];
		out_top_id[i] = ti[i];
	}
	for (uint32_t i = 0; i < nd; ++i)
		drain_id[i] = di[i];
	return nd;
}

// ---- GPU-assisted batched construction (graph_build.hpp) -------------------------------------
namespace {
using BuildSearchFn = void (*)(BuildSearchParams);
using BuildPruneFn = void (*)(BuildPruneParams);
struct BuildVariant {
	int d;
	BuildSearchFn search;
	BuildPruneFn prune;
};
#define BUILD_V(D) {D, build_search_kernel<D>, build_prune_kernel<D>}
const BuildVariant kBuild[] = {BUILD_V(64),  BUILD_V(128), BUILD_V(256), BUILD_V(512),
                               BUILD_V(768), BUILD_V(832), BUILD_V(960)};
#undef BUILD_V
struct BuildFail {
	std::string msg;
};
}  // namespace

int expann_graph_build_batched(int dim, int device, const float* vectors, size_t n, const uint8_t* levels,
                               size_t n_built, uint32_t* max_layer_io, uint32_t* starting_vertex_io, size_t M,
                               size_t M0, size_t ef_construction, size_t prune_overflow, float ortho_factor,
                               float ortho_bias, size_t max_batch, uint32_t* ids0, float* d0, uint32_t* deg0,
                               size_t stride0, const int32_t* upper_idx, size_t U, size_t n_upper_layers,
                               uint32_t* idsu, float* du, uint32_t* degu, size_t strideu, uint64_t* stats) {
	const BuildVariant* bv = nullptr;
	for (const auto& v : kBuild)
		if (v.d == dim)
			bv = &v;
	if (!bv) {
		g_create_error = "graph build is compiled for dim 64, 128, 256, 512, 768, 832, 960";
		return EXPANN_ERR_UNSUPPORTED;
	}
	if (!vectors || !levels || !max_layer_io || !starting_vertex_io || !ids0 || !d0 || !deg0 || !upper_idx ||
	    n == 0 || n_built == 0 || n_built > n || n >= (1ull << 32) - 64 || M < 2 || M0 < M || stride0 < M0 ||
	    (n_upper_layers && (!idsu || !du || !degu || strideu < M)) || ef_construction == 0 ||
	    ef_construction > (size_t)kPruneMaxCand || stride0 > (size_t)kPruneMaxCand ||
	    strideu > (size_t)kPruneMaxCand || *max_layer_io == 0 || *max_layer_io > n_upper_layers + 1 ||
	    *starting_vertex_io >= n_built) {
		g_create_error = "expann_graph_build_batched: bad arguments (ef_construction and row strides <= 1024)";
		return EXPANN_ERR_INVALID_ARG;
	}
	if (expann_device_count() <= device || device < 0) {
		g_create_error = "no such HIP device: libexpann_hip has no CPU fallback";
		return EXPANN_ERR_NO_DEVICE;
	}
	struct Fail {
		int fail(int code, const std::string& m) const {
			g_create_error = "expann_graph_build_batched: " + m;
			return code;
		}
	} F;
	Fail* fh = &F;
	HIP_TRY(fh, hipSetDevice(device));
	struct StreamGuard {  // (every early return below leaves through this)
		hipStream_t s = nullptr;
		~StreamGuard() {
			if (s)
				(void)hipStreamDestroy(s);
		}
	} sg;
	HIP_TRY(fh, hipStreamCreate(&sg.s));
	hipStream_t st = sg.s;
	DevBuf b_vec, b_lvl, b_up, b_id0, b_d0, b_deg0, b_idu, b_du, b_degu, b_vis, b_out, b_outc, b_tasks, b_upslot,
	    b_dirty, b_ctr;
	const size_t n_up_rows = U * n_upper_layers;
	HIP_TRY(fh, b_vec.alloc(n * (size_t)dim * 4));
	HIP_TRY(fh, b_lvl.alloc(n));
	HIP_TRY(fh, b_up.alloc(n * 4));
	HIP_TRY(fh, b_id0.alloc(n * stride0 * 4));
	HIP_TRY(fh, b_d0.alloc(n * stride0 * 4));
	HIP_TRY(fh, b_deg0.alloc(n * 4));
	HIP_TRY(fh, b_idu.alloc(std::max<size_t>(1, n_up_rows * strideu) * 4));
	HIP_TRY(fh, b_du.alloc(std::max<size_t>(1, n_up_rows * strideu) * 4));
	HIP_TRY(fh, b_degu.alloc(std::max<size_t>(1, n_up_rows) * 4));
	HIP_TRY(fh, hipMemcpy(b_vec.p, vectors, n * (size_t)dim * 4, hipMemcpyHostToDevice));
	HIP_TRY(fh, hipMemcpy(b_lvl.p, levels, n, hipMemcpyHostToDevice));
	HIP_TRY(fh, hipMemcpy(b_up.p, upper_idx, n * 4, hipMemcpyHostToDevice));
	HIP_TRY(fh, hipMemcpy(b_id0.p, ids0, n * stride0 * 4, hipMemcpyHostToDevice));
	HIP_TRY(fh, hipMemcpy(b_d0.p, d0, n * stride0 * 4, hipMemcpyHostToDevice));
	HIP_TRY(fh, hipMemcpy(b_deg0.p, deg0, n * 4, hipMemcpyHostToDevice));
	if (n_up_rows) {
		HIP_TRY(fh, hipMemcpy(b_idu.p, idsu, n_up_rows * strideu * 4, hipMemcpyHostToDevice));
		HIP_TRY(fh, hipMemcpy(b_du.p, du, n_up_rows * strideu * 4, hipMemcpyHostToDevice));
		HIP_TRY(fh, hipMemcpy(b_degu.p, degu, n_up_rows * 4, hipMemcpyHostToDevice));
	}
	BuildGraph g{};
	g.vec = b_vec.as<float>();
	g.n = (uint32_t)n;
	g.level = b_lvl.as<uint8_t>();
	g.upper_idx = b_up.as<int32_t>();
	g.id0 = b_id0.as<uint32_t>();
	g.d0 = b_d0.as<float>();
	g.deg0 = b_deg0.as<uint32_t>();
	g.cap0 = (uint32_t)M0;
	g.stride0 = (uint32_t)stride0;
	g.idu = b_idu.as<uint32_t>();
	g.du = b_du.as<float>();
	g.degu = b_degu.as<uint32_t>();
	g.capu = (uint32_t)M;
	g.strideu = (uint32_t)strideu;
	g.U = (uint32_t)U;

	const int cus = num_cus(device);
	if (max_batch == 0)
		max_batch = 32768;
	// search: one wave per new vertex; LDS = nearest heap + candidates heap + a hop's neighbour list
	const uint32_t list_cap = (uint32_t)std::max(stride0, std::max(strideu, ef_construction));
	uint32_t cand_cap = 8192;
	const size_t search_lds = sizeof(md_pair) * (ef_construction + 1 + cand_cap + 1) +
	                          (sizeof(uint32_t) + sizeof(float)) * list_cap + 8 * sizeof(uint32_t);
	HIP_TRY(fh, hipFuncSetAttribute((const void*)bv->search, hipFuncAttributeMaxDynamicSharedMemorySize, (int)search_lds));
	uint64_t slots = (uint64_t)cus * 2;  // resident search workgroups (two 72 KB workgroups per CU)
	// one visited bitset per resident workgroup, all zero between searches (graph_search.hpp)
	const uint32_t vis_words = (uint32_t)((n + 8191) / 8192 * 256);
	HIP_TRY(fh, b_vis.alloc(slots * vis_words * 4));
	HIP_TRY(fh, hipMemset(b_vis.p, 0, slots * vis_words * 4));
	const size_t max_up_in_batch = max_batch * (n_upper_layers ? 1 : 0) + 64;  // (bounded below per batch)
	HIP_TRY(fh, b_out.alloc((max_batch + max_up_in_batch) * ef_construction * sizeof(md_pair)));
	HIP_TRY(fh, b_outc.alloc((max_batch + max_up_in_batch) * 4));
	HIP_TRY(fh, b_tasks.alloc((max_batch + max_up_in_batch) * sizeof(PruneTask)));
	HIP_TRY(fh, b_upslot.alloc(max_batch * 4));
	const size_t dirty_cap = n + n_up_rows + 1;
	HIP_TRY(fh, b_dirty.alloc(dirty_cap * sizeof(uint2)));
	HIP_TRY(fh, b_ctr.alloc(8 * 4));  // [0] n_dirty, [1] dropped, [2] search overflow, [3] n_tasks
	HIP_TRY(fh, hipMemset(b_ctr.p, 0, 8 * 4));
	uint32_t* ctr = b_ctr.as<uint32_t>();

	uint32_t max_layer = *max_layer_io, starting_vertex = *starting_vertex_io;
	uint64_t n_batches = 0, n_dirty_total = 0;
	std::vector<PruneTask> tasks;
	std::vector<int32_t> up_slot;
	size_t b0 = n_built;
	while (b0 < n) {
		// a batch: at most 1/16 of the graph so far (its members do not see each other: measured
		// recall@10 at ef = 10, 12 k rows, M = 16: serial 0.384, batches of 1/8 0.354); a vertex that
		// opens a new layer goes alone
		size_t b1 = std::min(n, b0 + std::min(max_batch, std::max<size_t>(1, b0 / 16)));
		for (size_t v = b0; v < b1; ++v)
			if (levels[v] >= max_layer) {
				b1 = v == b0 ? v + 1 : v;
				break;
			}
		const uint32_t B = (uint32_t)(b1 - b0);
		tasks.clear();
		up_slot.assign(B, -1);
		uint32_t n_up = 0;
		for (uint32_t i = 0; i < B; ++i)
			tasks.push_back(PruneTask{(uint32_t)(b0 + i), 0u, (int32_t)i});
		for (uint32_t i = 0; i < B; ++i) {
			const uint32_t lv = std::min<uint32_t>(levels[b0 + i], max_layer - 1);
			if (lv >= 1) {
				up_slot[i] = (int32_t)n_up;
				for (uint32_t l = 1; l <= lv; ++l)
					tasks.push_back(PruneTask{(uint32_t)(b0 + i), l, (int32_t)(B + n_up + l - 1)});
				n_up += lv;
			}
		}
		if (n_up > max_up_in_batch)
			return F.fail(EXPANN_ERR_UNSUPPORTED, "more upper-layer vertices in a batch than provisioned");
		const uint32_t n_tasks = (uint32_t)tasks.size();
		HIP_TRY(fh, hipMemcpyAsync(b_tasks.p, tasks.data(), sizeof(PruneTask) * n_tasks, hipMemcpyHostToDevice, st));
		HIP_TRY(fh, hipMemcpyAsync(b_upslot.p, up_slot.data(), 4 * B, hipMemcpyHostToDevice, st));
		HIP_TRY(fh, hipMemcpyAsync(ctr + 3, &n_tasks, 4, hipMemcpyHostToDevice, st));
		HIP_TRY(fh, hipMemsetAsync(ctr, 0, 4, st));
		BuildSearchParams sp{};
		sp.g = g;
		sp.b0 = (uint32_t)b0;
		sp.b1 = (uint32_t)b1;
		sp.max_layer = max_layer;
		sp.starting_vertex = starting_vertex;
		sp.ef = (uint32_t)ef_construction;
		sp.cand_cap = cand_cap;
		sp.list_cap = list_cap;
		sp.vis_bits = b_vis.as<uint32_t>();
		sp.vis_words = vis_words;
		sp.up_slot = b_upslot.as<int32_t>();
		sp.out = b_out.as<md_pair>();
		sp.out_cnt = b_outc.as<uint32_t>();
		sp.error = ctr + 2;
		hipLaunchKernelGGL(bv->search, dim3((uint32_t)std::min<uint64_t>(B, slots)), dim3(64), search_lds, st, sp);
		BuildPruneParams pp{};
		pp.g = g;
		pp.tasks = b_tasks.as<PruneTask>();
		pp.n_tasks = ctr + 3;
		pp.lists = b_out.as<md_pair>();
		pp.list_cnt = b_outc.as<uint32_t>();
		pp.ef = (uint32_t)ef_construction;
		pp.ortho_factor = ortho_factor;
		pp.ortho_bias = ortho_bias;
		pp.prune_overflow = (uint32_t)prune_overflow;
		hipLaunchKernelGGL(bv->prune, dim3(std::min<uint32_t>(n_tasks, (uint32_t)cus * 16)), dim3(kPruneThreads), 0, st, pp);
		BuildReverseParams rp{};
		rp.g = g;
		rp.tasks = b_tasks.as<PruneTask>();
		rp.n_tasks = n_tasks;
		rp.dirty = b_dirty.as<uint2>();
		rp.n_dirty = ctr;
		rp.dirty_cap = (uint32_t)dirty_cap;
		rp.dropped = ctr + 1;
		hipLaunchKernelGGL(build_reverse_kernel, dim3((n_tasks + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, rp);
		// the rows that outgrew their cap: clamp to the slack, prune (grid-stride over the dirty list)
		hipLaunchKernelGGL(build_clamp_kernel, dim3((uint32_t)((dirty_cap + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, g,
		                   (const uint2*)b_dirty.p, (const uint32_t*)ctr);
		BuildPruneParams dp = pp;
		dp.tasks = nullptr;
		dp.n_tasks = ctr;
		dp.dirty = b_dirty.as<uint2>();
		hipLaunchKernelGGL(bv->prune, dim3((uint32_t)cus * 16), dim3(kPruneThreads), 0, st, dp);
		HIP_TRY(fh, hipGetLastError());
		uint32_t h_ctr[4];
		HIP_TRY(fh, hipMemcpyAsync(h_ctr, ctr, sizeof(h_ctr), hipMemcpyDeviceToHost, st));
		HIP_TRY(fh, hipStreamSynchronize(st));
		if (h_ctr[2])
			return F.fail(EXPANN_ERR_OVERFLOW, "construction search: candidates queue overflowed its LDS capacity");
		n_dirty_total += h_ctr[0];
		++n_batches;
		for (size_t v = b0; v < b1; ++v)  // :461-464
			while (levels[v] >= max_layer) {
				++max_layer;
				starting_vertex = (uint32_t)v;
			}
		b0 = b1;
	}
	HIP_TRY(fh, hipMemcpy(ids0, b_id0.p, n * stride0 * 4, hipMemcpyDeviceToHost));
	HIP_TRY(fh, hipMemcpy(d0, b_d0.p, n * stride0 * 4, hipMemcpyDeviceToHost));
	HIP_TRY(fh, hipMemcpy(deg0, b_deg0.p, n * 4, hipMemcpyDeviceToHost));
	if (n_up_rows) {
		HIP_TRY(fh, hipMemcpy(idsu, b_idu.p, n_up_rows * strideu * 4, hipMemcpyDeviceToHost));
		HIP_TRY(fh, hipMemcpy(du, b_du.p, n_up_rows * strideu * 4, hipMemcpyDeviceToHost));
		HIP_TRY(fh, hipMemcpy(degu, b_degu.p, n_up_rows * 4, hipMemcpyDeviceToHost));
	}
	uint32_t h_ctr[4];
	HIP_TRY(fh, hipMemcpy(h_ctr, ctr, sizeof(h_ctr), hipMemcpyDeviceToHost));
	*max_layer_io = max_layer;
	*starting_vertex_io = starting_vertex;
	if (stats) {
		stats[0] = n_batches;
		stats[1] = h_ctr[1];
		stats[2] = n_dirty_total;
		stats[3] = h_ctr[2];
	}
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

int expann_antitopo_store_batched(expann_antitopo* e, const float* rows, size_t n, size_t n_serial) {
	if (!e || (!rows && n))
		return EXPANN_ERR_INVALID_ARG;
	ANTITOPO_TRY(e, e->eng->store_rows_batched(rows, n, n_serial ? n_serial : 2048));
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
