// score_ids.hpp -- batched candidate scoring: scores[i] = score(query, base[ids[i]]).
// The device half of quantized_scorer::filter_by_score (src/quantizer.h:20-59) and of the
// neighbour-scoring loop of the graph search (src/antitopo_engine.h:636-689): a gather of
// whole rows by id, 16 lanes per row, same arithmetic order as scan_f32.hpp.
#pragma once
#include "common.hpp"

namespace expann {

struct ScoreIdsParams {
	const void* base;
	const void* query;      // [D]
	const uint64_t* ids;    // [n_ids] global ids
	uint64_t id_offset;
	uint32_t n_ids;
	float* scores;          // [n_ids]
};

template <int D, bool IP>
__global__ __launch_bounds__(kBlock) void score_ids_f32_kernel(ScoreIdsParams p) {
	constexpr int DPL = D / 16;
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t i = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const bool valid = i < p.n_ids;
	const uint64_t row = p.ids[valid ? i : p.n_ids - 1] - p.id_offset;
	const float* __restrict__ r = (const float*)p.base + (size_t)row * D + l;
	const float* __restrict__ q = (const float*)p.query + l;
	float acc = 0.0f;
#pragma unroll
	for (int t = 0; t < DPL; ++t) {
		if (IP) {
			acc = __builtin_fmaf(q[16 * t], r[16 * t], acc);
		} else {
			const float diff = q[16 * t] - r[16 * t];
			acc = __builtin_fmaf(diff, diff, acc);
		}
	}
	acc = reduce16_ref_order(acc);
	if (valid && l == 0)
		p.scores[i] = IP ? -acc : acc;
}

}  // namespace expann
