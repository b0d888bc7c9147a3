// expann_bench.cpp -- thin host driver: the counterpart of the reference's CLI
// (upstream src/main.cpp:32-109) for the brute-force path.  It accepts the same config
// JSON keys (dataset, ds_name, num_threads, k and n, m, d for "Synthetic",
// config_synthetic.json:1-9) with the same precedence CLI `--name value` > config file
// (src/main.cpp:32-50), generates the synthetic dataset the way the reference does
// (iid N(0,1) per component, un-normalised: src/randomgeometry.h:87-95; ground truth by
// exact brute force: src/dataset_loader.h:27-38) and prints one bench_data JSON per run
// mode with the reference's field names (src/bench_data.h:20-28).
//
// --devices 0,1,...,7 runs the engine under test row-sharded over several GPUs (BASELINE configs[2]).
//
// Differences, all deliberate: the dimension is a run-time value (the reference bakes
// -DDIM into the binary, CMakeLists.txt:87); the random seed is fixed (1234) instead of
// std::random_device; Sift1M needs *.fvecs files that are not shipped (use --dataset
// Synthetic); results go to stdout (and --out FILE) instead of ./data/<ds_name>/.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <string>

#include "expann/basic_bench.h"
#include "expann/dataset_io.h"
#include "expann/gpu_brute_force_engine.h"

// flat JSON object of string / number values -> map<string,string>
static std::map<std::string, std::string> parse_flat_json(const std::string& text) {
	std::map<std::string, std::string> out;
	size_t i = 0;
	auto skip = [&]() { while (i < text.size() && strchr(" \t\r\n,:{}", text[i])) ++i; };
	auto token = [&]() -> std::string {
		std::string t;
		if (text[i] == '"') {
			for (++i; i < text.size() && text[i] != '"'; ++i) t += text[i];
			++i;
		} else {
			for (; i < text.size() && !strchr(" \t\r\n,:{}", text[i]); ++i) t += text[i];
		}
		return t;
	};
	for (;;) {
		skip();
		if (i >= text.size()) break;
		std::string k = token();
		skip();
		if (i >= text.size()) break;
		out[k] = token();
	}
	return out;
}

int main(int argc, char** argv) {
	std::map<std::string, std::string> cli, cfg;
	for (int a = 1; a + 1 < argc; a += 2) {
		if (strncmp(argv[a], "--", 2) != 0) {
			std::fprintf(stderr, "expected --name value, got %s\n", argv[a]);
			return 2;
		}
		cli[argv[a] + 2] = argv[a + 1];
	}
	std::string config_file = cli.count("config") ? cli["config"] : "config.json";
	{
		std::ifstream f(config_file);
		if (f) {
			std::stringstream ss;
			ss << f.rdbuf();
			cfg = parse_flat_json(ss.str());
		} else if (cli.count("config")) {
			std::fprintf(stderr, "cannot open config %s\n", config_file.c_str());
			return 2;
		}
	}
	auto get = [&](const char* name, const char* dflt) -> std::string {
		if (cli.count(name)) return cli[name];
		if (cfg.count(name)) return cfg[name];
		if (dflt) return dflt;
		std::fprintf(stderr, "missing parameter %s (give --%s or put it in the config)\n", name, name);
		std::exit(2);
	};
	const std::string dataset = get("dataset", "Synthetic");
	dense_test_dataset<float> ds;
	const int device = std::stoi(get("device", "0"));
	// --devices 0,1,2,...: the engine under test shards the base over these GPUs (one handle, RCCL
	// all-gather of the per-shard top-k); the ground truth stays on --device
	std::vector<int> devices;
	{
		std::stringstream ss(get("devices", ""));
		std::string tok;
		while (std::getline(ss, tok, ','))
			if (!tok.empty())
				devices.push_back(std::stoi(tok));
	}
	auto engine_config = [&]() {
		return devices.size() > 1 ? gpu_brute_force_engine<float>::config(devices)
		                          : gpu_brute_force_engine<float>::config(devices.empty() ? device : devices[0]);
	};
	const std::string mode = get("mode", "both");  // serial | batched | both
	bool have_ground_truth = false;
	if (dataset == "Sift1M") {
		// src/main.cpp:72-80: fixed relative paths under datasets/sift/ (override: --sift_dir)
		const std::string dir = get("sift_dir", "datasets/sift");
		try {
			ds = expann::load_sift1m(dir + "/sift_base.fvecs", dir + "/sift_query.fvecs",
			                         dir + "/sift_groundtruth.ivecs", std::stoul(get("k", nullptr)));
		} catch (const std::exception& e) {
			std::fprintf(stderr, "error: %s\n", e.what());
			return 1;
		}
		if (cli.count("m") && std::stoul(cli["m"]) < ds.m) {  // load_sift1m_custom, :170-181
			ds.m = std::stoul(cli["m"]);
			ds.all_query_vecs.resize(ds.m * ds.dim);
			ds.all_query_ans.resize(ds.m);
		}
		have_ground_truth = true;
	} else if (dataset == "Synthetic") {
		ds.n = std::stoul(get("n", nullptr));
		ds.m = std::stoul(get("m", nullptr));
		ds.dim = std::stoul(get("d", nullptr));
		ds.k = std::stoul(get("k", nullptr));
		std::mt19937 gen(1234);
		std::normal_distribution<> nd(0, 1);
		ds.all_vecs.resize(ds.n * ds.dim);
		ds.all_query_vecs.resize(ds.m * ds.dim);
		for (auto& x : ds.all_vecs) x = float(nd(gen));
		for (auto& x : ds.all_query_vecs) x = float(nd(gen));
	} else {
		std::fprintf(stderr, "Invalid dataset type!\n");  // src/main.cpp:90-93
		return 1;
	}
	std::string ds_name = get("ds_name", "");
	if (ds_name.empty())
		ds_name = dataset;  // src/main.cpp:95-99
	ds.name = ds_name;

	try {
		if (!have_ground_truth) {  // exact brute force (src/dataset_loader.h:27-38), on the GPU
			gpu_brute_force_engine<float>::config gcfg(device);
			gpu_brute_force_engine<float> gt(gcfg);
			gt.store_rows(ds.all_vecs.data(), ds.n, ds.dim);
			gt.build();
			std::vector<uint64_t> ids(ds.m * ds.k);
			gt.query_k_batch(ds.all_query_vecs.data(), ds.m, ds.k, ids.data(), nullptr);
			unsigned long long checksum = 1469598103934665603ull;  // FNV-1a over the ids
			for (size_t q = 0; q < ds.m; ++q) {
				std::vector<size_t> ans;
				for (size_t i = 0; i < ds.k; ++i) {
					uint64_t id = ids[q * ds.k + i];
					checksum = (checksum ^ id) * 1099511628211ull;
					if (id != UINT64_MAX) ans.push_back(size_t(id));
				}
				ds.all_query_ans.push_back(ans);
			}
			std::printf("{\"ground_truth_ids_fnv1a\":\"%016llx\",\"n\":%zu,\"m\":%zu,\"d\":%zu,\"k\":%zu}\n",
			            checksum, ds.n, ds.m, ds.dim, ds.k);
		}
		basic_bench<float, dense_test_dataset<float>> bench(ds);
		expann::bench_data_manager bdm;
		std::ofstream out;
		if (cli.count("out")) out.open(cli["out"]);
		if (mode == "serial" || mode == "both") {
			gpu_brute_force_engine<float>::config ecfg = engine_config();
			gpu_brute_force_engine<float> eng(ecfg);
			bench_data bd = bench.get_benchmark_data(eng);
			bd.param_list["mode"] = "serial";
			bdm.add(bd);
			std::printf("%s\n", bd.to_string().c_str());
			if (out) out << bd.to_string() << "\n";
		}
		if (mode == "batched" || mode == "both") {
			gpu_brute_force_engine<float>::config ecfg = engine_config();
			gpu_brute_force_engine<float> eng(ecfg);
			bench_data bd = bench.get_benchmark_data_batched(eng);
			bd.param_list["mode"] = "batched";
			bdm.add(bd);
			std::printf("%s\n", bd.to_string().c_str());
			if (out) out << bd.to_string() << "\n";
		}
		if (get("save", "0") == "1")  // src/main.cpp:101-106: ./data/<ds_name>/data/{latest,all}.json
			bdm.save(get("data_root", "./data/") + ds_name + "/");
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
	return 0;
}
