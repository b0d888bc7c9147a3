// topk_ref.cpp -- thin C wrapper that instantiates the REFERENCE's own topk_t<float>
// (upstream src/topk_t.h, compiled unmodified from where it lies under /root/reference;
// its only dependency is the vendored src/robin_hood.h).  Built by oracle/Makefile into
// oracle/_ref/libtopk_ref.so; used only to pin oracle_topk_* and to generate
// tests/golden/topk_*.json (oracle/gen_golden.py).  No reference source is copied here.
#include "topk_t.h"

#include <cstddef>
#include <cstdint>

extern "C" {

// Feed (d[i], v[i]) through consider() in order; record is_good per step and size/worst
// after each step; at the end write the ascending (dist,id) list.  If discard_goal >= 0,
// discard_until_size(discard_goal) is applied before the final drain.
size_t ref_topk_run(size_t k, size_t n, const float* d, const uint64_t* v, long discard_goal,
                    uint8_t* is_good, uint64_t* size_after, uint64_t* worst_after,
                    float* worst_val_after, uint8_t* at_capacity_after, uint64_t* out_ids,
                    float* out_dists) {
	topk_t<float> t(k);
	for (size_t i = 0; i < n; ++i) {
		bool g = t.consider(d[i], size_t(v[i]));
		if (is_good) is_good[i] = g ? 1 : 0;
		if (size_after) size_after[i] = t.size();
		if (t.size() > 0) {
			if (worst_after) worst_after[i] = t.worst();
			if (worst_val_after) worst_val_after[i] = t.worst_val();
		}
		if (at_capacity_after) at_capacity_after[i] = t.at_capacity() ? 1 : 0;
	}
	if (discard_goal >= 0) t.discard_until_size(size_t(discard_goal));
	auto comb = t.to_combined_vector();
	auto ids = t.to_vector();
	for (size_t i = 0; i < comb.size(); ++i) {
		out_dists[i] = comb[i].first;
		out_ids[i] = comb[i].second;
		if (ids[i] != comb[i].second) return size_t(-1); // to_vector/to_combined disagree
	}
	return comb.size();
}
}
