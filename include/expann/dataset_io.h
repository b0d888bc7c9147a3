// dataset_io.h -- the reference's dataset and result file formats.
//   *.fvecs / *.ivecs  (src/dataset_loader.h:96-125): per vector a 32-bit dimension d followed by
//                      d 4-byte components.  Read into one dense row-major array; sizes are
//                      size_t throughout (the reference's `int n = tellg()/vecsizeof` overflows
//                      beyond 2 GiB, :107-108).
//   result files       (src/bench_data_manager.h:17-42,65-73, src/main.cpp:105-106):
//                      <prefix>data/latest.json is overwritten, <prefix>data/all.json appended;
//                      both are JSON arrays of bench_data objects (src/bench_data.h:20-28), which
//                      is what src/pyplotter.py consumes.
#pragma once

#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "basic_bench.h"

namespace expann {

template <typename T>
std::vector<T> read_tvecs(const std::string& filename, size_t& n, size_t& d) {
	static_assert(sizeof(T) == 4, "fvecs/ivecs components are 4 bytes");
	std::ifstream file(filename, std::ios::binary);
	if (!file.is_open())
		throw std::runtime_error("I/O error: Unable to open the file " + filename);
	int32_t d32 = 0;
	file.read(reinterpret_cast<char*>(&d32), 4);
	if (!file || d32 <= 0)
		throw std::runtime_error("I/O error: bad vecs header in " + filename);
	d = size_t(d32);
	file.seekg(0, std::ios::end);
	const size_t bytes = size_t(file.tellg());
	const size_t vecsizeof = 4 + d * 4;
	n = bytes / vecsizeof;
	file.seekg(0, std::ios::beg);
	std::vector<T> out(n * d);
	std::vector<char> rec(vecsizeof);
	for (size_t i = 0; i < n; ++i) {
		file.read(rec.data(), std::streamsize(vecsizeof));
		if (!file)
			throw std::runtime_error("I/O error: truncated " + filename);
		std::memcpy(out.data() + i * d, rec.data() + 4, d * 4);
	}
	return out;
}

// Sift1M wiring of src/dataset_loader.h:127-168: base/query fvecs + ground-truth ivecs,
// ground truth truncated to k.
inline dense_test_dataset<float> load_sift1m(const std::string& base_file,
                                             const std::string& query_file,
                                             const std::string& gt_file, size_t k) {
	dense_test_dataset<float> ds;
	size_t dq = 0, ngt = 0, dgt = 0;
	ds.all_vecs = read_tvecs<float>(base_file, ds.n, ds.dim);
	ds.all_query_vecs = read_tvecs<float>(query_file, ds.m, dq);
	if (dq != ds.dim)
		throw std::runtime_error("query and base dimensions differ");
	std::vector<int32_t> gt = read_tvecs<int32_t>(gt_file, ngt, dgt);
	if (ngt != ds.m || dgt < k)
		throw std::runtime_error("ground truth does not cover m queries x k answers");
	for (size_t q = 0; q < ds.m; ++q) {
		std::vector<size_t> a;
		for (size_t i = 0; i < k; ++i)
			a.push_back(size_t(gt[q * dgt + i]));
		ds.all_query_ans.push_back(a);
	}
	ds.k = k;
	ds.name = "sift1m_full_k" + std::to_string(k);
	return ds;
}

// bench_data_manager::save counterpart: latest.json overwritten, all.json appended.
struct bench_data_manager {
	std::vector<bench_data> latest;
	void add(const bench_data& bd) { latest.push_back(bd); }
	static std::string items(const std::vector<bench_data>& v) {
		std::string s;
		for (size_t i = 0; i < v.size(); ++i)
			s += (i ? ",\n    " : "    ") + v[i].to_string();
		return s;
	}
	void save(const std::string& prefix) const {
		const std::string dir = prefix + "data/";
		std::filesystem::create_directories(dir);
		{
			std::ofstream f(dir + "latest.json");
			f << "[\n" << items(latest) << "\n]\n";
		}
		std::string old;
		{
			std::ifstream f(dir + "all.json");
			if (f) {
				std::stringstream ss;
				ss << f.rdbuf();
				old = ss.str();
			}
		}
		while (!old.empty() && (old.back() == '\n' || old.back() == ' ' || old.back() == '\r'))
			old.pop_back();
		std::ofstream f(dir + "all.json");
		const size_t open = old.find('[');
		const bool has_items = !old.empty() && old.back() == ']' && open != std::string::npos &&
		                       old.find_first_not_of(" \n\r\t", open + 1) != old.size() - 1;
		if (has_items) {
			old.pop_back();
			while (!old.empty() && (old.back() == '\n' || old.back() == ' '))
				old.pop_back();
			f << old << ",\n" << items(latest) << "\n]\n";
		} else {
			f << "[\n" << items(latest) << "\n]\n";
		}
	}
};

}  // namespace expann
