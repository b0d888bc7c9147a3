// gpu_brute_force_engine.h -- drop-in MI355X counterpart of the reference's
// brute_force_engine<T> (upstream src/brute_force_engine.h:9-46).
//
// Same CRTP shape (ann_engine<T, Derived>: _name/_param_list/_store_vector/_build/_query_k,
// src/ann_engine.h:16-29) so that basic_bench::get_benchmark_data(ann_engine<T,Engine>&)
// (src/basic_bench.h:58-59) accepts it unchanged, plus what the reference's job machinery
// needs and brute_force_engine lacks: a nested `config` and an Engine(config) constructor
// (src/bench_runner.h:33,116,120-122).  All compute goes through the C ABI of
// include/expann_hip.h; this header holds no arithmetic.
//
// Semantics kept from the reference: store_vector copies the row and assigns ids in
// insertion order (:20-22); build on an empty index is an error (:25 asserts); query_k
// returns the ids of the min(k, n) rows with the smallest (dist2, id), ascending (:28-46).
// Errors from the ABI become std::runtime_error (the job wrapper turns them into the
// variant's error string, src/bench_runner.h:18,27).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ann_engine.h"
#include "expann_hip.h"

struct gpu_brute_force_engine_config {
	int device = 0;
	int metric = EXPANN_METRIC_L2;
	long query_tile = 0;  // 0 = auto
	// more than one entry: the base is row-sharded over these devices behind the same engine
	// (expann_sharded_*: contiguous row ranges, one RCCL all-gather of the per-shard top-k, merge);
	// results are bit-identical to the single-device engine
	std::vector<int> devices;
	gpu_brute_force_engine_config() = default;
	gpu_brute_force_engine_config(int _device, int _metric = EXPANN_METRIC_L2, long _query_tile = 0)
	    : device(_device), metric(_metric), query_tile(_query_tile) {}
	gpu_brute_force_engine_config(std::vector<int> _devices, int _metric = EXPANN_METRIC_L2, long _query_tile = 0)
	    : device(_devices.empty() ? 0 : _devices[0]), metric(_metric), query_tile(_query_tile),
	      devices(std::move(_devices)) {}
};

template <typename T>
struct gpu_brute_force_engine : public ann_engine<T, gpu_brute_force_engine<T>> {
	using config = gpu_brute_force_engine_config;
	config conf;
	expann_index* handle = nullptr;
	expann_sharded* sharded = nullptr;  // conf.devices.size() > 1
	size_t dimension = 0;
	size_t stored = 0;

	gpu_brute_force_engine() = default;
	explicit gpu_brute_force_engine(config c) : conf(c) {}
	gpu_brute_force_engine(const gpu_brute_force_engine&) = delete;
	gpu_brute_force_engine& operator=(const gpu_brute_force_engine&) = delete;
	~gpu_brute_force_engine() {
		expann_destroy(handle);
		expann_sharded_destroy(sharded);
	}

	void _store_vector(const vec<T>& v) {
		if (!opened())
			open(v.size());
		if (v.size() != dimension)
			throw std::runtime_error("gpu_brute_force_engine: row dimension changed");
		std::vector<T> row(dimension);
		for (size_t i = 0; i < dimension; ++i)
			row[i] = v.at(i);
		check(sharded ? expann_sharded_add(sharded, row.data(), 1) : expann_add(handle, row.data(), 1));
		++stored;
	}
	// batch form of store_vector (cf. store_many_vectors, src/pyrunner.cpp:60-82)
	void store_rows(const T* rows, size_t n, size_t dim) {
		if (!opened())
			open(dim);
		check(sharded ? expann_sharded_add(sharded, rows, n) : expann_add(handle, rows, n));
		stored += n;
	}
	void _build() {
		if (!opened())
			throw std::runtime_error("gpu_brute_force_engine: build() on an empty index");
		check(sharded ? expann_sharded_build(sharded) : expann_build(handle));
	}
	std::vector<size_t> _query_k(const vec<T>& v, size_t k) {
		std::vector<T> q(dimension);
		for (size_t i = 0; i < dimension; ++i)
			q[i] = v.at(i);
		std::vector<uint64_t> ids(k);
		query_k_batch(q.data(), 1, k, ids.data(), nullptr);
		std::vector<size_t> ret;
		for (size_t i = 0; i < k && ids[i] != UINT64_MAX; ++i)
			ret.push_back(size_t(ids[i]));
		return ret;
	}
	// Extension (the reference has no batch API): ids[m][k] / dists[m][k], rows ascending,
	// padded with UINT64_MAX / +inf when fewer than k rows exist.  dists may be nullptr.
	void query_k_batch(const T* queries, size_t m, size_t k, uint64_t* ids, float* dists) {
		if (!opened())
			throw std::runtime_error("gpu_brute_force_engine: query before build()");
		check(sharded ? expann_sharded_search(sharded, queries, m, k, ids, dists)
		              : expann_search(handle, queries, m, k, ids, dists));
	}
	const std::string _name() { return "GPU Brute-Force Engine (MI355X)"; }
	const param_list_t _param_list() {
		param_list_t pl;
		pl["device"] = std::to_string(conf.device);
		if (sharded) {
			std::string d;
			for (int x : conf.devices)
				d += (d.empty() ? "" : ",") + std::to_string(x);
			pl["devices"] = d;
			pl["shards"] = std::to_string(expann_sharded_shards(sharded));
		}
		pl["metric"] = conf.metric == EXPANN_METRIC_IP ? "ip" : "l2";
		pl["query_tile"] = std::to_string(conf.query_tile);
		return pl;
	}

private:
	// the row types the library stores (include/expann_hip.h, expann_dtype); anything else is a
	// compile-time error rather than a silently wrong row type
	static constexpr int dtype_of() {
		static_assert(std::is_same<T, float>::value || std::is_same<T, uint8_t>::value ||
		                  std::is_same<T, int8_t>::value || std::is_same<T, int16_t>::value,
		              "gpu_brute_force_engine<T>: T must be float, uint8_t, int8_t or int16_t");
		return std::is_same<T, float>::value     ? EXPANN_DTYPE_F32
		       : std::is_same<T, uint8_t>::value ? EXPANN_DTYPE_U8
		       : std::is_same<T, int8_t>::value  ? EXPANN_DTYPE_I8
		                                         : EXPANN_DTYPE_I16;
	}
	bool opened() const { return handle || sharded; }
	void open(size_t dim) {
		dimension = dim;
		if (conf.devices.size() > 1) {
			int rc = expann_sharded_create(int(dim), dtype_of(), conf.metric, conf.devices.data(),
			                               int(conf.devices.size()), &sharded);
			if (rc != EXPANN_OK)
				throw std::runtime_error(std::string("expann_sharded_create: ") + expann_sharded_last_error(nullptr));
			if (conf.query_tile)
				check(expann_sharded_set_option(sharded, "query_tile", conf.query_tile));
			return;
		}
		int rc = expann_create(int(dim), dtype_of(), conf.metric, conf.devices.empty() ? conf.device : conf.devices[0],
		                       &handle);
		if (rc != EXPANN_OK)
			throw std::runtime_error(std::string("expann_create: ") + expann_last_error(nullptr));
		if (conf.query_tile)
			check(expann_set_option(handle, "query_tile", conf.query_tile));
	}
	void check(int rc) {
		if (rc != EXPANN_OK)
			throw std::runtime_error(std::string("expann_hip: ") +
			                         (sharded ? expann_sharded_last_error(sharded) : expann_last_error(handle)));
	}
};
