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

// The `d < cutoff` half of filter_by_score (src/quantizer.h:42-46) on the device: an ORDER-KEEPING
// compaction of (id, score) -- the reference pushes the survivors in input order.  One workgroup
// walks the list in blocks of 1024: ballot + lane rank inside a wave, wave totals through LDS, a
// running offset across blocks; *n_kept receives the count.
struct FilterScoresParams {
	const uint64_t* ids;     // [n_ids]
	const float* scores;     // [n_ids]
	uint32_t n_ids;
	float cutoff;
	uint64_t* kept_ids;      // [n_ids]
	float* kept_scores;      // [n_ids]
	uint32_t* n_kept;        // [1]
};
__global__ __launch_bounds__(1024) void filter_scores_kernel(FilterScoresParams p) {
	__shared__ uint32_t wave_tot[16];
	__shared__ uint32_t running;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0)
		running = 0;
	__syncthreads();
	for (uint32_t i0 = 0; i0 < p.n_ids; i0 += 1024) {
		const uint32_t i = i0 + threadIdx.x;
		const float sc = i < p.n_ids ? p.scores[i] : 0.0f;
		const bool keep = i < p.n_ids && sc < p.cutoff;   // (a NaN score is never kept, as `d < cutoff` on the host)
		const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
		const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		if (lane == 0)
			wave_tot[wave] = (uint32_t)__builtin_popcountll(mask);
		__syncthreads();
		uint32_t off = running;
		for (uint32_t w = 0; w < wave; ++w)
			off += wave_tot[w];
		if (keep) {
			p.kept_ids[off + before] = p.ids[i];
			p.kept_scores[off + before] = sc;
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t t = running;
			for (int w = 0; w < 16; ++w)
				t += wave_tot[w];
			running = t;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0)
		*p.n_kept = running;
}

}  // namespace expann
