// basic_bench.h -- host-side counterpart of the reference's benchmark harness
// (upstream src/basic_bench.h:19-150, src/bench_data.h:8-40, src/dataset.h).
//
// get_benchmark_data() times exactly what the reference times: store_vector x n + build()
// as the build span (:63-71), then m serial query_k calls with the per-query bookkeeping
// (top-1 distance, duplicate-id check, recall against the dataset's ground truth) inside the
// timed span (:82-126).  get_benchmark_data_batched() is the same with ONE query_k_batch
// call inside the span (the reference has no batch API; the bookkeeping stays in the span).
// bench_data carries the reference's result fields and serialises to the same JSON keys.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ann_engine.h"

struct bench_data {
	double time_per_query_ns = 0;
	double time_to_build_ns = 0;
	double average_distance = 0;
	double average_squared_distance = 0;
	double recall = 0;
	std::string engine_name;
	param_list_t param_list;

	// same keys as to_json(bench_data), src/bench_data.h:20-28
	std::string to_string() const {
		std::ostringstream o;
		o.precision(17);
		o << "{\"average_distance\":" << average_distance
		  << ",\"average_squared_distance\":" << average_squared_distance
		  << ",\"engine_name\":\"" << engine_name << "\",\"param_list\":{";
		bool first = true;
		for (const auto& kv : param_list) {
			o << (first ? "" : ",") << "\"" << kv.first << "\":\"" << kv.second << "\"";
			first = false;
		}
		o << "},\"recall\":" << recall << ",\"time_per_query_ns\":" << time_per_query_ns
		  << ",\"time_to_build_ns\":" << time_to_build_ns << "}";
		return o.str();
	}
};

// Dense in-memory test dataset with the reference's accessor names (src/dataset.h:9-30,
// src/in_memory_dataset.h): rows are returned BY VALUE, as in the reference.
template <typename T> struct dense_test_dataset {
	std::string name;
	size_t n = 0, m = 0, k = 0, dim = 0;
	std::vector<T> all_vecs;                       // [n][dim]
	std::vector<T> all_query_vecs;                 // [m][dim]
	std::vector<std::vector<size_t>> all_query_ans;  // [m][<=k]
	vec<T> get_vec(size_t i) const { return vec<T>(all_vecs.data() + i * dim, dim); }
	vec<T> get_query(size_t i) const { return vec<T>(all_query_vecs.data() + i * dim, dim); }
	std::vector<size_t> get_query_ans(size_t i) const { return all_query_ans[i]; }
};

template <typename T, typename test_dataset_t> struct basic_bench {
	const test_dataset_t& ds;
	explicit basic_bench(const test_dataset_t& _ds) : ds(_ds) {}

	template <class Engine> bench_data get_benchmark_data(ann_engine<T, Engine>& eng) const {
		// (takes the CRTP base like src/basic_bench.h:58-59)
		bench_data ret;
		using clk = std::chrono::high_resolution_clock;
		auto b0 = clk::now();
		for (size_t i = 0; i < ds.n; ++i)
			eng.store_vector(ds.get_vec(i));
		eng.build();
		auto b1 = clk::now();
		stats st;
		auto t0 = clk::now();
		for (size_t q = 0; q < ds.m; ++q) {
			std::vector<size_t> ans = eng.query_k(ds.get_query(q), ds.k);
			account(q, ans, st);
		}
		auto t1 = clk::now();
		finish(ret, st, ns(t0, t1), ns(b0, b1), eng.name(), eng.param_list());
		return ret;
	}

	// Engine must additionally offer store_rows() and query_k_batch().
	template <class Engine> bench_data get_benchmark_data_batched(Engine& eng) const {
		bench_data ret;
		using clk = std::chrono::high_resolution_clock;
		auto b0 = clk::now();
		eng.store_rows(ds.all_vecs.data(), ds.n, ds.dim);
		eng.build();
		auto b1 = clk::now();
		stats st;
		std::vector<uint64_t> ids(ds.m * ds.k);
		auto t0 = clk::now();
		eng.query_k_batch(ds.all_query_vecs.data(), ds.m, ds.k, ids.data(), nullptr);
		for (size_t q = 0; q < ds.m; ++q) {
			std::vector<size_t> ans;
			for (size_t i = 0; i < ds.k && ids[q * ds.k + i] != UINT64_MAX; ++i)
				ans.push_back(size_t(ids[q * ds.k + i]));
			account(q, ans, st);
		}
		auto t1 = clk::now();
		finish(ret, st, ns(t0, t1), ns(b0, b1), eng.name(), eng.param_list());
		return ret;
	}

private:
	struct stats {
		double avg_dist = 0, avg_dist2 = 0;
		size_t num_best_found = 0;
	};
	template <class TP> static double ns(TP a, TP b) {
		return double(std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count());
	}
	void account(size_t q, const std::vector<size_t>& ans, stats& st) const {
		if (!ans.empty()) {  // src/basic_bench.h:91-96
			st.avg_dist += dist(ds.get_query(q), ds.get_vec(ans[0]));
			st.avg_dist2 += dist2(ds.get_query(q), ds.get_vec(ans[0]));
		}
		std::set<size_t> ans_s(ans.begin(), ans.end());  // :98-104
		if (ans_s.size() != ans.size())
			throw std::runtime_error("Duplicates detected, engine is buggy.");
		for (size_t e : ds.get_query_ans(q))  // :116-121
			if (ans_s.count(e))
				++st.num_best_found;
	}
	void finish(bench_data& ret, const stats& st, double query_ns, double build_ns,
	            const std::string& name, const param_list_t& pl) const {
		ret.time_per_query_ns = query_ns / double(ds.m);  // :131-135
		ret.time_to_build_ns = build_ns;
		ret.average_distance = st.avg_dist / double(ds.m);
		ret.average_squared_distance = st.avg_dist2 / double(ds.m);
		ret.recall = double(st.num_best_found) / double(ds.m * ds.k);  // :143
		ret.param_list = pl;
		ret.engine_name = name;
	}
};
