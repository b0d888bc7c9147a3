// std_heap_hook.cpp -- test hook (compiled by tests/test_heap_pin.py with g++): the trace
// interface of oracle_heap_trace (oracle/expann_oracle.h) over expann::std_heap, the heap the
// host-side graph builder uses (include/expann/antitopo_index.h), so that it can be compared
// with the image's real std::priority_queue (tests/golden/heap_ref.json).
#include <cstddef>
#include <cstdint>
#include <utility>

#include "expann/antitopo_index.h"

using edge = std::pair<float, size_t>;

template <typename Less>
static size_t run(Less less, size_t n_init, const float* init_d, const uint64_t* init_id, size_t n_ops,
                  const int* ops, const float* op_d, const uint64_t* op_id, uint64_t* out_size,
                  float* out_top_d, uint64_t* out_top_id, float* drain_d, uint64_t* drain_id) {
	expann::std_heap<edge, Less> h(less);
	for (size_t i = 0; i < n_init; ++i)
		h.v.push_back(edge(init_d[i], size_t(init_id[i])));
	h.make();
	for (size_t i = 0;; ++i) {
		out_size[i] = h.size();
		out_top_d[i] = h.size() ? h.top().first : 0.0f;
		out_top_id[i] = h.size() ? h.top().second : 0;
		if (i == n_ops)
			break;
		if (ops[i] == 1)
			h.push(edge(op_d[i], size_t(op_id[i])));
		else if (!h.empty())
			h.pop();
	}
	size_t nd = 0;
	while (!h.empty()) {
		drain_d[nd] = h.top().first;
		drain_id[nd] = h.top().second;
		++nd;
		h.pop();
	}
	return nd;
}

extern "C" size_t std_heap_trace(int max_heap, size_t n_init, const float* init_d, const uint64_t* init_id,
                                 size_t n_ops, const int* ops, const float* op_d, const uint64_t* op_id,
                                 uint64_t* out_size, float* out_top_d, uint64_t* out_top_id,
                                 float* drain_d, uint64_t* drain_id) {
	auto worst_elem = [](const edge& a, const edge& b) { return a.first < b.first; };
	auto best_elem = [](const edge& a, const edge& b) { return a.first > b.first; };
	return max_heap ? run(worst_elem, n_init, init_d, init_id, n_ops, ops, op_d, op_id, out_size, out_top_d,
	                      out_top_id, drain_d, drain_id)
	                : run(best_elem, n_init, init_d, init_id, n_ops, ops, op_d, op_id, out_size, out_top_d,
	                      out_top_id, drain_d, drain_id);
}
