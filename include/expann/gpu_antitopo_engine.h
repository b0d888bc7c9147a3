// gpu_antitopo_engine.h -- drop-in MI355X counterpart of the reference's graph engine
// `antitopo_engine<T>` (upstream src/antitopo_engine.h:104-260) behind the same CRTP shape
// (ann_engine<T, Derived>, src/ann_engine.h:16-29) and the same `config` convention
// (src/bench_runner.h:33,116,120-122; antitopo_engine_config, src/antitopo_engine.h:72-101).
//
// Division of labour: the graph is BUILT on the host (antitopo_index.h, a restatement of
// _store_vector / prune_edges; or read from an index file written by the reference itself,
// read_index/write_index), the QUERY side -- greedy descent, best-first bottom-layer search with
// its candidate scoring and queues, uint8 path and re-score -- runs on the GPU through
// expann_graph_search.  _query_k keeps the reference's "sticky" ef_search (:858-859).
#pragma once

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "ann_engine.h"
#include "antitopo_index.h"
#include "expann_hip.h"

struct gpu_antitopo_engine_config : public expann::antitopo_config {
	int device = 0;
	std::string index_filename;
	bool read_index = false, write_index = false;
	gpu_antitopo_engine_config() = default;
	// same argument order as antitopo_engine_config (src/antitopo_engine.h:88-101)
	gpu_antitopo_engine_config(size_t _M, size_t _M0, size_t _ef_search_mult, size_t _ef_construction,
	                           size_t _ortho_count, float _ortho_factor, float _ortho_bias,
	                           size_t _prune_overflow, bool _use_compression = false,
	                           bool _use_largest_direction_filtering = false,
	                           std::string _index_filename = "", bool _read_index = false,
	                           bool _write_index = false) {
		M = _M; M0 = _M0; ef_search_mult = _ef_search_mult; ef_construction = _ef_construction;
		ortho_count = _ortho_count; ortho_factor = _ortho_factor; ortho_bias = _ortho_bias;
		prune_overflow = _prune_overflow; use_compression = _use_compression;
		use_largest_direction_filtering = _use_largest_direction_filtering;
		index_filename = _index_filename; read_index = _read_index; write_index = _write_index;
	}
};

template <typename T>
struct gpu_antitopo_engine : public ann_engine<T, gpu_antitopo_engine<T>> {
	using config = gpu_antitopo_engine_config;
	config conf;
	expann::antitopo_index index;
	expann_graph* graph = nullptr;
	std::optional<size_t> ef_search;
	size_t num_distcomps = 0;  // RECORD_STATS counter (src/antitopo_engine.h:125-129)

	explicit gpu_antitopo_engine(config c) : conf(c) { index.conf = c; }
	gpu_antitopo_engine(const gpu_antitopo_engine&) = delete;
	gpu_antitopo_engine& operator=(const gpu_antitopo_engine&) = delete;
	~gpu_antitopo_engine() { expann_graph_destroy(graph); }

	void set_ef_search(size_t e) {  // src/antitopo_engine.h:189-195
		ef_search = e;
		index.has_ef_search = true;
		index.ef_search = e;
	}

	void _store_vector(const vec<T>& v) {
		if (conf.read_index)
			return;  // :312-313
		if (index.dim == 0)
			index.dim = v.size();
		std::vector<float> row(index.dim);
		for (size_t i = 0; i < index.dim; ++i)
			row[i] = float(v.at(i));
		index.insert(row.data());
	}
	// Extension: `n` rows through the batched GPU builder (expann_graph_build_batched,
	// csrc/graph_build.hpp).  The first rows of an empty engine -- until n_serial are stored -- go
	// through the serial restatement and seed the batches; level draws stay the reference's sequence.
	// Returns the builder's statistics {batches, dropped reverse edges, rows re-pruned, 0}.
	std::vector<uint64_t> store_rows_batched(const float* rows, size_t n, size_t n_serial = 2048, size_t max_batch = 0) {
		std::vector<uint64_t> stats(4, 0);
		if (conf.read_index || n == 0)
			return stats;
		if (index.dim == 0)
			throw std::runtime_error("gpu_antitopo_engine: dimension not set");
		if (conf.ortho_count != 1) {  // (several ortho entry points: serial path only)
			for (size_t i = 0; i < n; ++i)
				index.insert(rows + i * index.dim);
			return stats;
		}
		std::vector<uint8_t> lv(n);
		for (size_t i = 0; i < n; ++i)
			lv[i] = (uint8_t)std::min<size_t>(index.draw_level(), 250);
		size_t i = 0;
		for (; i < n && index.size() < n_serial; ++i)
			index.insert_with_level(rows + i * index.dim, lv[i]);
		if (i == n)
			return stats;
		const size_t built = index.size();
		auto g = index.to_strided(std::vector<uint8_t>(lv.begin() + i, lv.end()), 64);
		std::vector<float> all(index.vectors);
		all.insert(all.end(), rows + i * index.dim, rows + n * index.dim);
		uint32_t ml = (uint32_t)index.max_layer, sv = (uint32_t)index.starting_vertex;
		int rc = expann_graph_build_batched(int(index.dim), conf.device, all.data(), g.n, g.levels.data(), built, &ml, &sv,
		                                    conf.M, conf.M0, conf.ef_construction, conf.prune_overflow,
		                                    conf.ortho_factor, conf.ortho_bias, max_batch, g.ids0.data(), g.d0.data(),
		                                    g.deg0.data(), g.stride0, g.upper_idx.data(), g.U, g.n_upper_layers,
		                                    g.idsu.data(), g.du.data(), g.degu.data(), g.strideu, stats.data());
		if (rc != EXPANN_OK)
			throw std::runtime_error(std::string("expann_graph_build_batched: ") + expann_graph_last_error(nullptr));
		std::vector<float>().swap(all);
		index.from_strided(g, rows + i * index.dim, ml, sv);
		return stats;
	}
	void _build() {  // :467-493
		if (conf.write_index && !conf.index_filename.empty())
			index.write_index(conf.index_filename);
		if (conf.read_index)
			index.read_index(conf.index_filename);
		if (index.size() == 0)
			throw std::runtime_error("gpu_antitopo_engine: build() on an empty index");
		upload();
		num_distcomps = 0;
	}
	void upload() {
		const auto fg = index.flatten();
		expann_graph_destroy(graph);
		graph = nullptr;
		int rc = expann_graph_create(int(index.dim), conf.device, index.vectors.data(), index.size(),
		                             fg.n_layers, fg.starting_vertex, fg.layer_offsets.data(),
		                             fg.neighbours.data(), &graph);
		if (rc != EXPANN_OK)
			throw std::runtime_error(std::string("expann_graph_create: ") +
			                         expann_graph_last_error(nullptr));
	}
	std::vector<size_t> _query_k(const vec<T>& v, size_t k) {
		std::vector<float> q(index.dim);
		for (size_t i = 0; i < index.dim; ++i)
			q[i] = float(v.at(i));
		std::vector<uint64_t> ids(k);
		std::vector<float> dists(k);
		query_k_batch(q.data(), 1, k, ids.data(), dists.data());
		std::vector<size_t> ret;
		for (size_t i = 0; i < k && ids[i] != UINT64_MAX; ++i)
			ret.push_back(size_t(ids[i]));
		return ret;
	}
	// Extension: m queries in one launch.  distcomps (per query) may be nullptr.
	void query_k_batch(const float* queries, size_t m, size_t k, uint64_t* ids, float* dists,
	                   uint32_t* distcomps = nullptr) {
		if (!graph)
			throw std::runtime_error("gpu_antitopo_engine: query before build()");
		if (!ef_search.has_value())
			set_ef_search(k * conf.ef_search_mult);  // :858-859
		std::vector<uint32_t> dc(m);
		int rc = expann_graph_search(graph, queries, m, k, ef_search.value(),
		                             conf.use_compression ? 1 : 0, ids, dists, dc.data());
		if (rc != EXPANN_OK)
			throw std::runtime_error(std::string("expann_graph_search: ") +
			                         expann_graph_last_error(graph));
		for (size_t i = 0; i < m; ++i) {
			num_distcomps += dc[i];
			if (distcomps)
				distcomps[i] = dc[i];
		}
	}
	const std::string _name() { return "GPU Anti-Topo Engine+ (MI355X)"; }
	const param_list_t _param_list() {  // :242-259
		param_list_t pl;
		pl["M"] = std::to_string(conf.M);
		pl["M0"] = std::to_string(conf.M0);
		pl["ef_search_mult"] = std::to_string(conf.ef_search_mult);
		pl["ef_construction"] = std::to_string(conf.ef_construction);
		pl["ortho_count"] = std::to_string(conf.ortho_count);
		pl["ortho_factor"] = std::to_string(conf.ortho_factor);
		pl["ortho_bias"] = std::to_string(conf.ortho_bias);
		pl["prune_overflow"] = std::to_string(conf.prune_overflow);
		pl["use_compression"] = std::to_string(conf.use_compression);
		pl["use_largest_direction_filtering"] = std::to_string(conf.use_largest_direction_filtering);
		pl["num_distcomps"] = std::to_string(num_distcomps);
		return pl;
	}
};
