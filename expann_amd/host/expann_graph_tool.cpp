// expann_graph_tool.cpp -- host driver for the graph path (config C4): builds an antitopo graph
// on the CPU (include/expann/antitopo_index.h), writes it in the reference's index format, and
// (unless --build-only 1) searches it on the GPU through gpu_antitopo_engine for a list of
// ef_search values, with and without the uint8 "compression" path.
//
//   expann_graph_tool --n 5000 --m 200 --d 128 --k 10 --M 16 --ef_construction 100 \
//       --data sift|gauss --index out.index --queries out.queries --results out.results \
//       [--ef 10,20,40] [--build-only 1] [--read-index 1] [--prune_overflow 0|1] [--batched 2048]
//
// Files: <queries> raw m*d float32; <results> for each (compression in {0,1}) x (ef in list):
// m*k uint64 ids, m*k float32 dists, m uint32 distcomps, in that order.  One JSON line per
// configuration on stdout (time per query, distance evaluations), like the reference's
// bench_data + RECORD_STATS (src/bench_data.h:20-28, src/antitopo_engine.h:254-257).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <sstream>
#include <string>

#include "expann/gpu_antitopo_engine.h"
#include "expann/gpu_brute_force_engine.h"

int main(int argc, char** argv) {
	std::map<std::string, std::string> a;
	for (int i = 1; i + 1 < argc; i += 2)
		a[argv[i] + 2] = argv[i + 1];
	auto get = [&](const char* k, const char* d) { return a.count(k) ? a[k] : std::string(d); };
	const size_t n = std::stoul(get("n", "2000")), m = std::stoul(get("m", "100")),
	             d = std::stoul(get("d", "128")), k = std::stoul(get("k", "10"));
	const size_t M = std::stoul(get("M", "16")), efc = std::stoul(get("ef_construction", "100"));
	const size_t prune_overflow = std::stoul(get("prune_overflow", "0"));
	// --batched S: rows beyond the first S go through the batched GPU builder (0 = serial host build)
	const size_t batched = std::stoul(get("batched", "0"));
	const bool sift = get("data", "sift") == "sift";
	const bool build_only = get("build-only", "0") == "1", read_index = get("read-index", "0") == "1";
	const std::string index_path = get("index", "graph.index");

	std::mt19937 gen(std::stoul(get("seed", "1234")));
	std::normal_distribution<> nd(0, 1);
	auto draw = [&]() -> float {
		const double x = nd(gen);
		if (!sift)
			return float(x);
		double v = std::round(std::fabs(x) * 40.0);  // SIFT-like stand-in
		return float(v < 0 ? 0 : (v > 255 ? 255 : v));
	};
	std::vector<float> base(n * d), queries(m * d);
	for (auto& x : base) x = draw();
	std::uniform_real_distribution<> frac(0.0, 0.99);
	for (auto& x : queries) {
		x = draw();
		if (sift)  // fractional parts: the uint8 path truncates the query (antitopo_engine.h:726-737)
			x = std::min(255.5f, x + float(frac(gen)));
	}

	try {
		gpu_antitopo_engine_config cfg(M, 2 * M, 1, efc, 1, 0.5f, 0.0f, prune_overflow);
		cfg.index_filename = index_path;
		cfg.read_index = read_index;
		cfg.write_index = !read_index;
		gpu_antitopo_engine<float> eng(cfg);
		auto t0 = std::chrono::high_resolution_clock::now();
		std::vector<uint64_t> bstats;
		if (batched && !read_index) {
			eng.index.dim = d;
			bstats = eng.store_rows_batched(base.data(), n, batched);
		} else {
			for (size_t i = 0; i < n; ++i)
				eng.store_vector(vec<float>(base.data() + i * d, d));
		}
		if (build_only) {
			if (!read_index)
				eng.index.write_index(index_path);
			else
				eng.index.read_index(index_path);
		} else {
			eng.build();
		}
		auto t1 = std::chrono::high_resolution_clock::now();
		std::printf("{\"phase\":\"build\",\"n\":%zu,\"d\":%zu,\"M\":%zu,\"ef_construction\":%zu,"
		            "\"max_layer\":%zu,\"time_to_build_ns\":%.0f,\"builder\":\"%s\",\"batches\":%llu,"
		            "\"dropped_reverse_edges\":%llu,\"rows_repruned\":%llu}\n",
		            eng.index.size(), d, M, efc, eng.index.max_layer,
		            double(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count()),
		            bstats.empty() ? "serial host" : "batched gpu", bstats.empty() ? 0ull : (unsigned long long)bstats[0],
		            bstats.empty() ? 0ull : (unsigned long long)bstats[1], bstats.empty() ? 0ull : (unsigned long long)bstats[2]);
		if (a.count("queries")) {
			std::ofstream qf(a["queries"], std::ios::binary);
			qf.write(reinterpret_cast<const char*>(queries.data()), (std::streamsize)(queries.size() * 4));
		}
		if (get("strided-roundtrip", "0") == "1") {
			// the batched builder's view of the graph (fixed-stride rows) and back: must be lossless
			auto g = eng.index.to_strided({}, 64);
			expann::antitopo_index copy(eng.index.dim, eng.index.conf);
			copy.hadj_flat_with_lengths.clear();
			const std::vector<float> rows = eng.index.vectors;
			copy.from_strided(g, rows.data(), eng.index.max_layer, eng.index.starting_vertex);
			copy.has_ef_search = eng.index.has_ef_search;
			copy.ef_search = eng.index.ef_search;
			eng.index.hadj_flat_with_lengths = copy.hadj_flat_with_lengths;
			eng.index.hadj_flat = copy.hadj_flat;
		}
		if (a.count("rewrite")) {  // read_index -> write_index round trip
			eng.index.write_index(a["rewrite"]);
		}
		if (build_only)
			return 0;
		// exact ground truth for the recall column (src/dataset_loader.h:27-38: brute force)
		std::vector<uint64_t> gt(m * k);
		{
			gpu_brute_force_engine<float>::config bcfg(0);
			gpu_brute_force_engine<float> bf(bcfg);
			bf.store_rows(eng.index.vectors.data(), eng.index.size(), d);
			bf.build();
			bf.query_k_batch(queries.data(), m, k, gt.data(), nullptr);
		}
		std::vector<size_t> efs;
		{
			std::stringstream ss(get("ef", "10,20,40"));
			std::string tok;
			while (std::getline(ss, tok, ','))
				efs.push_back(std::stoul(tok));
		}
		std::ofstream rf;
		if (a.count("results"))
			rf.open(a["results"], std::ios::binary);
		for (int comp = 0; comp <= 1; ++comp)
			for (size_t ef : efs) {
				eng.conf.use_compression = comp != 0;
				eng.set_ef_search(ef);
				std::vector<uint64_t> ids(m * k);
				std::vector<float> dists(m * k);
				std::vector<uint32_t> dc(m);
				eng.query_k_batch(queries.data(), m, k, ids.data(), dists.data(), dc.data());  // warm
				auto q0 = std::chrono::high_resolution_clock::now();
				eng.query_k_batch(queries.data(), m, k, ids.data(), dists.data(), dc.data());
				auto q1 = std::chrono::high_resolution_clock::now();
				double evals = 0;
				for (auto x : dc) evals += x;
				size_t found = 0;  // src/basic_bench.h:116-121,143
				for (size_t q = 0; q < m; ++q)
					for (size_t i = 0; i < k; ++i)
						for (size_t j = 0; j < k; ++j)
							if (ids[q * k + j] == gt[q * k + i]) {
								++found;
								break;
							}
				const double recall = double(found) / double(m * k);
				std::printf("{\"phase\":\"query\",\"use_compression\":%d,\"ef_search\":%zu,"
				            "\"time_per_query_ns\":%.1f,\"kernel_ms\":%.4f,\"distcomps_per_query\":%.1f,"
				            "\"recall\":%.4f}\n",
				            comp, ef,
				            double(std::chrono::duration_cast<std::chrono::nanoseconds>(q1 - q0).count()) / double(m),
				            expann_graph_last_kernel_ms(eng.graph), evals / double(m), recall);
				if (rf) {
					rf.write(reinterpret_cast<const char*>(ids.data()), (std::streamsize)(ids.size() * 8));
					rf.write(reinterpret_cast<const char*>(dists.data()), (std::streamsize)(dists.size() * 4));
					rf.write(reinterpret_cast<const char*>(dc.data()), (std::streamsize)(dc.size() * 4));
				}
			}
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
	return 0;
}
