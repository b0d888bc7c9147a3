// select.hpp -- exact k-selection over a query's candidate keys, and the k-way merge of
// per-shard results.  The order is the reference's: ascending (score, id), i.e. the
// lexicographic std::pair order of the max-heaps in src/topk_t.h:11 and
// src/brute_force_engine.h:29-46 (k smallest, ties by lower id, output ascending).
#pragma once
#include "common.hpp"

namespace expann {

struct SelectParams {
	const uint64_t* cand;     // [m][cap] keys
	const uint32_t* cand_cnt; // [m], or nullptr: every query has fixed_count keys
	uint32_t fixed_count;
	uint32_t cap;
	uint32_t k;
	uint64_t id_offset;
	uint64_t* out_ids;        // [m][k] or nullptr
	float* out_dists;         // [m][k] or nullptr
	float* tau_out;           // [m] k-th smallest score, or nullptr
	const float* tau_prev;    // [m] carried forward when fewer than k keys (nullptr: +inf)
	// re-rank (GEMM-form scan): when rerank_base != nullptr the keys hold only a row number
	// (low 32 bits); the exact reference-order score is recomputed here before the sort
	const float* rerank_base;     // [n][dim]
	const float* rerank_queries;  // [m][dim]
	uint32_t dim;
	uint32_t metric_ip;
	uint32_t* overflow;       // [1] number of queries whose list overflowed cap
	unsigned long long* total_cand;  // [1] sum of counts (statistics) or nullptr
};

// One workgroup per query: bitonic sort of the (power-of-two padded) key list in LDS.
__global__ __launch_bounds__(kBlock) void select_topk_kernel(SelectParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
	const uint32_t qi = blockIdx.x;
	const uint32_t tid = threadIdx.x;
	uint32_t c = p.cand_cnt ? p.cand_cnt[qi] : p.fixed_count;
	if (tid == 0 && p.total_cand)
		atomicAdd(p.total_cand, (unsigned long long)c);
	if (c > p.cap) {
		if (tid == 0)
			atomicAdd(p.overflow, 1u);
		c = p.cap;
	}
	uint32_t n2 = 2;
	while (n2 < c)
		n2 <<= 1;
	const uint64_t* src = p.cand + (size_t)qi * p.cap;
	if (p.rerank_base) {
		// 16 lanes per candidate row, lane l owns dims l, l+16, ... in increasing order and
		// the partial sums meet in the _mm512_reduce_add_ps tree: the scores written here are
		// bit-identical to scan_filter_f32_kernel's (src/distance.h:136-147 / :181-190).
		const uint32_t l = tid & 15, grp = tid >> 4;  // 16 candidates per pass
		const float* q = p.rerank_queries + (size_t)qi * p.dim + l;
		for (uint32_t i0 = 0; i0 < n2; i0 += kBlock / 16) {
			const uint32_t i = i0 + grp;
			uint64_t key = kSentinelKey;
			if (i < c)
				key = src[i];
			const uint32_t row = (i < c) ? key_idx(key) : 0u;
			const float* r = p.rerank_base + (size_t)row * p.dim + l;
			float acc = 0.0f;
			for (uint32_t t = 0; t < p.dim / 16; ++t) {
				if (p.metric_ip) {
					acc = __builtin_fmaf(q[16 * t], r[16 * t], acc);
				} else {
					const float diff = q[16 * t] - r[16 * t];
					acc = __builtin_fmaf(diff, diff, acc);
				}
			}
			acc = reduce16_ref_order(acc);
			if (l == 0 && i < n2)
				keys[i] = (i < c) ? make_key(p.metric_ip ? -acc : acc, row) : kSentinelKey;
		}
	} else {
		for (uint32_t i = tid; i < n2; i += kBlock)
			keys[i] = i < c ? src[i] : kSentinelKey;
	}
	__syncthreads();
	for (uint32_t size = 2; size <= n2; size <<= 1) {
		for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
			for (uint32_t i = tid; i < (n2 >> 1); i += kBlock) {
				const uint32_t lo = 2 * i - (i & (stride - 1));
				const uint32_t hi = lo + stride;
				const bool up = ((lo & size) == 0);
				const uint64_t a = keys[lo], b = keys[hi];
				if ((a > b) == up) {
					keys[lo] = b;
					keys[hi] = a;
				}
			}
			__syncthreads();
		}
	}
	for (uint32_t i = tid; i < p.k; i += kBlock) {
		const uint64_t key = i < n2 ? keys[i] : kSentinelKey;
		const bool ok = key != kSentinelKey;
		if (p.out_ids)
			p.out_ids[(size_t)qi * p.k + i] = ok ? (uint64_t)key_idx(key) + p.id_offset : ~0ull;
		if (p.out_dists)
			p.out_dists[(size_t)qi * p.k + i] = ok ? key_score(key) : __builtin_inff();
	}
	if (tid == 0 && p.tau_out) {
		const uint64_t key = (p.k - 1 < n2) ? keys[p.k - 1] : kSentinelKey;
		p.tau_out[qi] = key != kSentinelKey ? key_score(key)
		                                    : (p.tau_prev ? p.tau_prev[qi] : __builtin_inff());
	}
}

// k-way merge of n_lists ascending (score, id) lists per query; one thread per query.
// Lists are padded with (+inf, UINT64_MAX).  Order: score, then id (64-bit).
__global__ __launch_bounds__(kBlock) void merge_topk_kernel(const uint64_t* in_ids,
                                                            const float* in_dists,
                                                            uint32_t n_lists, uint32_t m,
                                                            uint32_t k, uint64_t* out_ids,
                                                            float* out_dists) {
	const uint32_t qi = blockIdx.x * kBlock + threadIdx.x;
	if (qi >= m)
		return;
	constexpr int kMaxLists = 64;
	uint32_t pos[kMaxLists];
	for (uint32_t g = 0; g < n_lists; ++g)
		pos[g] = 0;
	for (uint32_t i = 0; i < k; ++i) {
		uint32_t best = n_lists;
		uint32_t bo = 0;
		uint64_t bid = ~0ull;
		for (uint32_t g = 0; g < n_lists; ++g) {
			if (pos[g] >= k)
				continue;
			const size_t off = ((size_t)g * m + qi) * k + pos[g];
			const uint64_t id = in_ids[off];
			if (id == ~0ull)
				continue;  // padding: this list is exhausted
			const uint32_t o = float_to_ordered(in_dists[off]);
			if (best == n_lists || o < bo || (o == bo && id < bid)) {
				best = g;
				bo = o;
				bid = id;
			}
		}
		if (best == n_lists) {
			out_ids[(size_t)qi * k + i] = ~0ull;
			out_dists[(size_t)qi * k + i] = __builtin_inff();
		} else {
			out_ids[(size_t)qi * k + i] = bid;
			out_dists[(size_t)qi * k + i] = ordered_to_float(bo);
			pos[best]++;
		}
	}
}

}  // namespace expann
