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
	uint32_t* tau_row_out;    // [m] row of the k-th smallest key (UINT32_MAX when carried) or nullptr
	const uint32_t* tau_row_prev;
	// re-rank (GEMM-form scan): when rerank_base != nullptr the keys hold only a row number
	// (low 32 bits); the exact reference-order score is recomputed here before the sort
	const float* rerank_base;     // [n][dim]
	const float* rerank_queries;  // [m][dim]
	uint32_t dim;
	uint32_t metric_ip;
	// rerank pruning (L2 GEMM forms): the approximate key A = bn(1-eps) - 2 q.b satisfies
	//   A + qn(1-eps/2)  <=  reference-order score  <=  A + qn(1+eps/2) + 1.5 eps bn
	// (the slack eps is at least twice the worst-case evaluation error, scan_gemm_*.hpp), so a
	// candidate whose A exceeds the k-th smallest A by more than eps*(qn + 1.5*bn_max) cannot be
	// among the k best and need not be re-scored.  prune_eps == 0 disables it.
	float prune_eps;
	float prune_abs;          // absolute slack coefficient (fp16 form): + prune_abs*(|q| + |b|max)
	const float* bn_max;      // [1] max over rows of ||b||^2 (1-eps)
	uint32_t* overflow;       // [1] number of queries whose list overflowed cap
	uint32_t wave_done;       // select_wave_kernel already served the lists of <= wave_done keys
	uint32_t wave0_short;     // select_topk_kernel: lists of <= 2048 keys go through wave 0's wave path
	const float* qnrm;        // [m] ||q||^2 if the caller has it (pruning margin), else nullptr
};

// statistics: *out = sum of counts (one workgroup; one same-address atomic per query in the
// select kernel would cost ~10 ns each, 100 us at 10^4 queries)
__global__ __launch_bounds__(1024) void sum_u32_kernel(const uint32_t* in, uint32_t n,
                                                       unsigned long long* out) {
	__shared__ unsigned long long red[16];
	unsigned long long s = 0;
	for (uint32_t i = threadIdx.x; i < n; i += 1024)
		s += in[i];
	for (int off = 32; off > 0; off >>= 1)
		s += __shfl_xor(s, off);
	if ((threadIdx.x & 63) == 0)
		red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 16; ++w)
			s += red[w];
		out[0] = s;
	}
}

// min over the 64 lanes (result in every lane)
__device__ inline uint32_t wave_min_u32(uint32_t v) {
	// within rows of 16 by DPP rotations, then across the four rows
	v = min(v, (uint32_t)row_ror_i<8>((int)v));
	v = min(v, (uint32_t)row_ror_i<4>((int)v));
	v = min(v, (uint32_t)row_ror_i<2>((int)v));
	v = min(v, (uint32_t)row_ror_i<1>((int)v));
	v = min(v, (uint32_t)__shfl_xor((int)v, 16));
	v = min(v, (uint32_t)__shfl_xor((int)v, 32));
	return v;
}
__device__ inline uint64_t wave_min_u64(uint64_t v) {
	// the high word first (scores), the low word (rows) among the lanes that hold the min score
	const uint32_t hi = (uint32_t)(v >> 32);
	const uint32_t mhi = wave_min_u32(hi);
	const uint32_t lo = hi == mhi ? (uint32_t)v : 0xFFFFFFFFu;
	return ((uint64_t)mhi << 32) | wave_min_u32(lo);
}

// ---- one WAVE per query -------------------------------------------------------------------
// Lists of at most 2048 candidates (what the sampled thresholds leave) are ordered by a single
// wave: the chain count -> keys -> candidate rows -> output is three dependent memory round
// trips whatever the arithmetic, so the win over one 256-thread workgroup per query is 4x the
// queries in flight and no workgroup barriers.  With rerank_base the keys carry approximate
// scores of a GEMM-form scan (pruned, then re-scored exactly); without, final scores.  Longer
// lists are left to select_topk_kernel (which skips the queries served here: p.wave_done).

// the k smallest of the n_s <= 64*NE keys in list[], ascending, to the outputs (+ the next tau).
// Small k: k rounds of wave-min extraction, lane 0 writes.  Larger k: every lane ranks its keys
// by counting (all n_s keys stream past as LDS broadcasts) and writes them to their rank.
template <int NE>
__device__ inline void wave_emit_sorted(const SelectParams& p, const uint64_t* list, uint32_t n_s,
                                        uint32_t qi, int lane) {
	uint64_t e[NE];
#pragma unroll
	for (int j = 0; j < NE; ++j)
		e[j] = lane + 64 * j < (int)n_s ? list[lane + 64 * j] : kSentinelKey;
	if (p.k > 16) {
		uint32_t r[NE];
#pragma unroll
		for (int j = 0; j < NE; ++j)
			r[j] = 0;
		for (uint32_t i = 0; i < n_s; ++i) {
			const uint64_t x = list[i];
#pragma unroll
			for (int j = 0; j < NE; ++j)
				r[j] += x < e[j] ? 1u : 0u;
		}
#pragma unroll
		for (int j = 0; j < NE; ++j) {
			if (e[j] == kSentinelKey || r[j] >= p.k)
				continue;  // (keys are distinct: ranks are a permutation)
			if (p.out_ids)
				p.out_ids[(size_t)qi * p.k + r[j]] = (uint64_t)key_idx(e[j]) + p.id_offset;
			if (p.out_dists)
				p.out_dists[(size_t)qi * p.k + r[j]] = key_score(e[j]);
			if (r[j] == p.k - 1 && p.tau_out) {
				p.tau_out[qi] = key_score(e[j]);
				if (p.tau_row_out)
					p.tau_row_out[qi] = key_idx(e[j]);
			}
		}
		for (uint32_t i = n_s + lane; i < p.k; i += 64) {  // fewer than k keys: padding
			if (p.out_ids)
				p.out_ids[(size_t)qi * p.k + i] = ~0ull;
			if (p.out_dists)
				p.out_dists[(size_t)qi * p.k + i] = __builtin_inff();
		}
		if (lane == 0 && n_s < p.k && p.tau_out) {
			p.tau_out[qi] = p.tau_prev ? p.tau_prev[qi] : __builtin_inff();
			if (p.tau_row_out)
				p.tau_row_out[qi] = p.tau_row_prev ? p.tau_row_prev[qi] : 0xFFFFFFFFu;
		}
		return;
	}
	for (uint32_t r = 0; r < p.k; ++r) {
		uint64_t mn = e[0];
#pragma unroll
		for (int j = 1; j < NE; ++j)
			mn = e[j] < mn ? e[j] : mn;
		mn = wave_min_u64(mn);
		const bool ok = mn != kSentinelKey;
		if (lane == 0) {
			if (p.out_ids)
				p.out_ids[(size_t)qi * p.k + r] = ok ? (uint64_t)key_idx(mn) + p.id_offset : ~0ull;
			if (p.out_dists)
				p.out_dists[(size_t)qi * p.k + r] = ok ? key_score(mn) : __builtin_inff();
			if (r == p.k - 1 && p.tau_out) {
				p.tau_out[qi] = ok ? key_score(mn) : (p.tau_prev ? p.tau_prev[qi] : __builtin_inff());
				if (p.tau_row_out)
					p.tau_row_out[qi] = ok ? key_idx(mn) : (p.tau_row_prev ? p.tau_row_prev[qi] : 0xFFFFFFFFu);
			}
		}
#pragma unroll
		for (int j = 0; j < NE; ++j)
			e[j] = e[j] == mn ? kSentinelKey : e[j];
	}
}

// k-th smallest of the 64*PER values held by a wave (duplicates counted; all-ones = absent):
// bisection on the value, one ballot + popcount per element and step -- 32 steps whatever k
template <int PER> __device__ inline uint32_t wave_kth_smallest_u32(const uint32_t (&v)[PER], uint32_t k) {
	uint32_t lo = 0, hi = 0xFFFFFFFFu;  // smallest x with count(v <= x) >= k
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		uint32_t cnt = 0;
#pragma unroll
		for (int j = 0; j < PER; ++j)
			cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v[j] <= mid));
		if (cnt >= k)
			hi = mid;
		else
			lo = mid + 1;
	}
	return lo;
}

// k-th largest (0 = absent; 0 is returned when fewer than k values are present)
template <int PER> __device__ inline uint32_t wave_kth_largest_u32(const uint32_t (&v)[PER], uint32_t k) {
	uint32_t lo = 0, hi = 0xFFFFFFFFu;  // largest x with count(v >= x) >= k
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1);
		uint32_t cnt = 0;
#pragma unroll
		for (int j = 0; j < PER; ++j)
			cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v[j] >= mid));
		if (cnt >= k)
			lo = mid;
		else
			hi = mid - 1;
	}
	return lo;
}

// The same k-th largest when k <= 64 and PER is large: B = the k-th largest of the 64 per-lane
// maxima is a lower bound of the answer (k values >= B exist) and usually only a few more values
// reach it -- they are compacted into `scratch` (64 words of LDS, this wave's own) and the
// bisection runs over one value per lane instead of PER.  Falls back to the full bisection when
// more than 64 values reach B (ties, k close to the number of values).
template <int PER>
__device__ inline uint32_t wave_kth_largest_sparse_u32(const uint32_t (&v)[PER], uint32_t k, uint32_t* scratch,
                                                       int lane) {
	uint32_t mx[1] = {v[0]};
#pragma unroll
	for (int j = 1; j < PER; ++j)
		mx[0] = v[j] > mx[0] ? v[j] : mx[0];
	const uint32_t B = wave_kth_largest_u32<1>(mx, k);
	uint32_t total = 0;
#pragma unroll
	for (int j = 0; j < PER; ++j)
		total += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(v[j] >= B));
	if (total > 64)
		return wave_kth_largest_u32<PER>(v, k);
	uint32_t base = 0;  // wave-uniform
#pragma unroll
	for (int j = 0; j < PER; ++j) {
		const unsigned long long mask = __builtin_amdgcn_ballot_w64(v[j] >= B);
		if (v[j] >= B)
			scratch[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
			                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = v[j];
		base += (uint32_t)__builtin_popcountll(mask);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	uint32_t w[1] = {lane < (int)total ? scratch[lane] : 0u};
	return wave_kth_largest_u32<1>(w, k);
}

// one wave orders the list of query qi (c <= 64 * PER keys); list = 64 * PER keys of LDS
template <int PER>
__device__ inline void select_wave_body(const SelectParams& p, uint32_t qi, uint32_t c, uint64_t* list,
                                        int lane) {
	const uint64_t* src = p.cand + (size_t)qi * p.cap;
	uint64_t kk[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j)
		kk[j] = lane + 64 * j < (int)c ? src[lane + 64 * j] : kSentinelKey;
	const uint32_t l = lane & 15, grp = lane >> 4;  // 16 lanes per candidate row
	const float* q = p.rerank_queries + (size_t)qi * p.dim + l;
	float cutoff = __builtin_inff();
	// lists with FINAL scores (the 8-bit forms: no re-score) are cut at their k-th smallest score itself --
	// what is left to rank is k keys + ties instead of the whole list (k = 100 on uint8 rows: 535 keys, each
	// ranked against all of them, were 0.31 ms of a 2.4 ms step)
	const bool final_scores = p.rerank_base == nullptr;
	if ((p.prune_eps > 0.0f || final_scores) && c > p.k) {
		float qn = 0.0f;
		if (final_scores) {
		} else if (p.qnrm) {
			qn = p.qnrm[qi];
		} else {
			for (uint32_t t = 0; t < p.dim / 16; ++t)
				qn = __builtin_fmaf(q[16 * t], q[16 * t], qn);
			qn = reduce16_ref_order(qn);
		}
		// k-th smallest approximate score (k-th DISTINCT one if scores repeat: a valid, looser cut)
		uint32_t sc[PER];
#pragma unroll
		for (int j = 0; j < PER; ++j)
			sc[j] = (uint32_t)(kk[j] >> 32);
		uint32_t kth = 0xFFFFFFFFu;
		if (p.k > 24) {
			kth = wave_kth_smallest_u32<PER>(sc, p.k);
		} else
		for (uint32_t r = 0; r < p.k; ++r) {
			uint32_t mn = sc[0];
#pragma unroll
			for (int j = 1; j < PER; ++j)
				mn = min(mn, sc[j]);
			mn = wave_min_u32(mn);
			if (mn == 0xFFFFFFFFu)
				break;
			kth = mn;
#pragma unroll
			for (int j = 0; j < PER; ++j)
				sc[j] = sc[j] == mn ? 0xFFFFFFFFu : sc[j];
		}
		if (final_scores) {
			cutoff = ordered_to_float(kth);
		} else {
			const float bmax = p.bn_max[0];
			cutoff = ordered_to_float(kth) + p.prune_eps * (qn + 1.5f * bmax) +
			         p.prune_abs * (__builtin_sqrtf(qn) + __builtin_sqrtf(bmax));
		}
	}
	uint32_t n_s = 0;  // wave-uniform
#pragma unroll
	for (int j = 0; j < PER; ++j) {
		const bool hit = lane + 64 * j < (int)c && key_score(kk[j]) <= cutoff;
		const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
		const uint32_t pos = n_s + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
		                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		if (hit)
			list[pos] = kk[j];
		n_s += (uint32_t)__builtin_popcountll(mask);
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: LDS ops are in order)
	// exact re-score, two candidates per 16-lane group in flight; slot i is read and rewritten
	// by the one group that owns it.  (Lists with final scores -- the 8-bit forms -- skip it.)
	for (uint32_t i0 = 0; p.rerank_base && i0 < n_s; i0 += 16) {
		constexpr int U = 4;  // candidates per 16-lane group in flight
		float acc[U];
		uint32_t row[U];
		const float* r[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const uint32_t i = i0 + 4 * u + grp;
			row[u] = key_idx(list[i < n_s ? i : 0]);
			r[u] = p.rerank_base + (size_t)row[u] * p.dim + l;
			acc[u] = 0.0f;
		}
		// 8 dims of every candidate are requested before any is consumed (the trip count is a
		// run-time value: without this the loop waits for each load in turn); per candidate the
		// FMA chain still runs over t in increasing order, as the reference's lane does
		const uint32_t nt = p.dim / 16;
		for (uint32_t t0 = 0; t0 < nt; t0 += 8) {
			float qv[8], rv[U][8];
#pragma unroll
			for (int t = 0; t < 8; ++t)
				qv[t] = t0 + t < nt ? q[16 * (t0 + t)] : 0.0f;
#pragma unroll
			for (int u = 0; u < U; ++u)
#pragma unroll
				for (int t = 0; t < 8; ++t)
					rv[u][t] = t0 + t < nt ? r[u][16 * (t0 + t)] : 0.0f;
#pragma unroll
			for (int u = 0; u < U; ++u)
#pragma unroll
				for (int t = 0; t < 8; ++t) {
					if (t0 + t >= nt)
						continue;  // (d = 64: four dims per lane)
					if (p.metric_ip) {
						acc[u] = __builtin_fmaf(qv[t], rv[u][t], acc[u]);
					} else {
						const float diff = qv[t] - rv[u][t];
						acc[u] = __builtin_fmaf(diff, diff, acc[u]);
					}
				}
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			acc[u] = reduce16_ref_order(acc[u]);
			const uint32_t i = i0 + 4 * u + grp;
			if (l == 0 && i < n_s)
				list[i] = make_key(p.metric_ip ? -acc[u] : acc[u], row[u]);
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	if (n_s <= 256)
		wave_emit_sorted<(PER < 4 ? PER : 4)>(p, list, n_s, qi, lane);
	else
		wave_emit_sorted<PER>(p, list, n_s, qi, lane);
}

// lists of at most 64 * PER keys; WAVES queries per workgroup (PER = 32: one, its list is 16 KB)
template <int PER, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void select_wave_kernel(SelectParams p, uint32_t m) {
	constexpr uint32_t kSelectWaveMax = 64 * PER;
	__shared__ uint64_t lists[WAVES][kSelectWaveMax];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t qi = blockIdx.x * WAVES + wave;
	if (qi >= m)
		return;
	const uint32_t c = p.cand_cnt ? p.cand_cnt[qi] : p.fixed_count;
	if (c > kSelectWaveMax || c > p.cap || (c <= p.wave_done && p.wave_done != 0))
		return;  // longer lists: the next size up; shorter ones: already served
	select_wave_body<PER>(p, qi, c, lists[wave], lane);
}

// One workgroup per query: bitonic sort of the (power-of-two padded) key list in LDS.
__global__ __launch_bounds__(kBlock) void select_topk_kernel(SelectParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
	const uint32_t qi = blockIdx.x;
	const uint32_t tid = threadIdx.x;
	uint32_t c = p.cand_cnt ? p.cand_cnt[qi] : p.fixed_count;
	if (p.wave_done && c <= p.wave_done && c <= p.cap)
		return;  // served by a wave kernel (those leave overflowed lists, c > cap, to this one)
	if (p.wave0_short && c <= 2048 && c <= p.cap && 8 * (size_t)p.cap >= 16384) {
		// few queries in the launch: no separate wave kernels; wave 0 of this workgroup takes the
		// short list, the other waves leave
		if (tid < 64) {
			if (c <= 512)
				select_wave_body<8>(p, qi, c, keys, (int)tid);
			else
				select_wave_body<32>(p, qi, c, keys, (int)tid);
		}
		return;
	}
	if (c > p.cap) {
		if (tid == 0)
			atomicAdd(p.overflow, 1u);
		c = p.cap;
	}
	uint32_t n2 = 2;
	while (n2 < c)
		n2 <<= 1;
	const uint64_t* src = p.cand + (size_t)qi * p.cap;
	for (uint32_t i = tid; i < n2; i += kBlock)
		keys[i] = i < c ? src[i] : kSentinelKey;
	__syncthreads();
	auto bitonic = [&](uint32_t len) {
		for (uint32_t size = 2; size <= len; size <<= 1) {
			for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
				for (uint32_t i = tid; i < (len >> 1); i += kBlock) {
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
	};
	if (p.rerank_base) {
		const uint32_t l = tid & 15, grp = tid >> 4;  // 16 lanes per candidate row
		const float* q = p.rerank_queries + (size_t)qi * p.dim + l;
		uint32_t n_rescore = c;
		if (p.prune_eps > 0.0f && c > p.k) {
			// order by the approximate key, keep what can still reach the k best
			bitonic(n2);
			float qn = 0.0f;
			for (uint32_t t = 0; t < p.dim / 16; ++t)
				qn = __builtin_fmaf(q[16 * t], q[16 * t], qn);
			qn = reduce16_ref_order(qn);
			const float bmax = p.bn_max[0];
			const float cutoff = key_score(keys[p.k - 1]) + p.prune_eps * (qn + 1.5f * bmax) +
			                     p.prune_abs * (__builtin_sqrtf(qn) + __builtin_sqrtf(bmax));
			uint32_t cnt = 0;
			for (uint32_t i = tid; i < c; i += kBlock)
				cnt += key_score(keys[i]) <= cutoff ? 1u : 0u;
			// (counter lives behind the keys in the dynamic LDS region: a static __shared__ object
			// would shift the 16-byte alignment of the dynamic base)
			uint32_t* s_cnt = reinterpret_cast<uint32_t*>(keys + p.cap);
			if (tid == 0)
				*s_cnt = 0;
			__syncthreads();
			if (cnt)
				atomicAdd(s_cnt, cnt);
			__syncthreads();
			n_rescore = *s_cnt;  // keys are sorted: exactly the first *s_cnt qualify
			__syncthreads();
		}
		uint32_t n2r = 2;
		while (n2r < n_rescore)
			n2r <<= 1;
		// exact re-score: lane l owns dims l, l+16, ... in increasing order and the partial sums
		// meet in the _mm512_reduce_add_ps tree: bit-identical to scan_filter_f32_kernel's scores
		// (src/distance.h:136-147 / :181-190)
		for (uint32_t i0 = 0; i0 < n2r; i0 += kBlock / 16) {
			const uint32_t i = i0 + grp;
			const bool live = i < n_rescore;
			const uint32_t row = live ? key_idx(keys[i]) : 0u;
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
			__syncthreads();  // every key of this pass has been read before any is overwritten
			if (l == 0 && i < n2r)
				keys[i] = live ? make_key(p.metric_ip ? -acc : acc, row) : kSentinelKey;
		}
		__syncthreads();
		n2 = n2r;
	}
	bitonic(n2);
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
		if (p.tau_row_out)
			p.tau_row_out[qi] = key != kSentinelKey
			                        ? key_idx(key)
			                        : (p.tau_row_prev ? p.tau_row_prev[qi] : 0xFFFFFFFFu);
	}
}

// max of a float array (single workgroup; used once per build for the rerank pruning bound)
__global__ __launch_bounds__(1024) void max_f32_kernel(const float* in, size_t n, float* out) {
	__shared__ float red[16];
	float m = -__builtin_inff();
	for (size_t i = threadIdx.x; i < n; i += 1024)
		m = in[i] > m ? in[i] : m;
	for (int off = 32; off > 0; off >>= 1) {
		const float o = __shfl_xor(m, off);
		m = o > m ? o : m;
	}
	if ((threadIdx.x & 63) == 0)
		red[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 16; ++w)
			m = red[w] > m ? red[w] : m;
		out[0] = m;
	}
}

// k-way merge of n_lists ascending (score, id) lists per query: ONE WAVE per query, every element
// finds its own place.  Lists are padded with (+inf, UINT64_MAX); order: score, then id (64-bit),
// then list number (distinct shards never tie on an id).  List g starts at in_ids + g*ids_stride /
// in_dists + g*dists_stride (elements): m*k each for separate [n_lists][m][k] arrays, the chunk size
// when every rank's chunk is [ids | dists], or the slice stride of the all-to-all exchange.
// Element i of list g has rank i + sum over the other lists of (entries ordered before it) -- a
// binary search per list in the sorted lists, no serial k-step chain (round 2's one-thread-per-query
// merge walked k x n_lists dependent global loads: latency-bound whatever m is, and its per-list
// cursor array lived in scratch).  STAGE: the n_lists x k entries of the query are first copied to
// LDS (12 B each) and searched there; else they are searched where they lie (lists too long for LDS).
template <bool STAGE>
__global__ __launch_bounds__(kWave) void merge_topk_kernel(const uint64_t* in_ids, const float* in_dists,
                                                            size_t ids_stride, size_t dists_stride,
                                                            uint32_t n_lists, uint32_t m, uint32_t k,
                                                            uint64_t* out_ids, float* out_dists) {
	extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
	const uint32_t qi = blockIdx.x, lane = threadIdx.x;
	if (qi >= m)
		return;
	const uint32_t total = n_lists * k;
	uint64_t* const s_id = reinterpret_cast<uint64_t*>(merge_smem);            // [total]
	uint32_t* const s_sc = reinterpret_cast<uint32_t*>(merge_smem + 8 * (size_t)total);  // [total]
	const size_t qoff = (size_t)qi * k;
	auto load = [&](uint32_t g, uint32_t i, uint64_t& id, uint32_t& sc) {
		if (STAGE) {
			id = s_id[g * k + i];
			sc = s_sc[g * k + i];
		} else {
			id = in_ids[g * ids_stride + qoff + i];
			sc = id == ~0ull ? 0xFFFFFFFFu : float_to_ordered(in_dists[g * dists_stride + qoff + i]);
		}
	};
	uint32_t n_valid = 0;
	if (STAGE) {
		for (uint32_t e = lane; e < total; e += kWave) {
			const uint32_t g = e / k, i = e - g * k;
			const uint64_t id = in_ids[g * ids_stride + qoff + i];
			s_id[e] = id;
			s_sc[e] = id == ~0ull ? 0xFFFFFFFFu : float_to_ordered(in_dists[g * dists_stride + qoff + i]);
		}
		__syncthreads();
	}
	for (uint32_t e = lane; e < total; e += kWave) {
		const uint32_t g = e / k, i = e - g * k;
		uint64_t id;
		uint32_t sc;
		load(g, i, id, sc);
		if (id == ~0ull)
			continue;  // padding: behind every real entry
		n_valid++;
		uint32_t rank = i;
		for (uint32_t g2 = 0; g2 < n_lists && rank < k; ++g2) {
			if (g2 == g)
				continue;
			// entries of list g2 ordered before (sc, id, g): none beyond position k - rank matter
			uint32_t lo = 0, hi = k - rank;
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				uint64_t xid;
				uint32_t xsc;
				load(g2, mid, xid, xsc);
				const bool before = xsc < sc || (xsc == sc && (xid < id || (xid == id && g2 < g)));
				if (before)
					lo = mid + 1;
				else
					hi = mid;
			}
			rank += lo;
		}
		if (rank < k) {
			out_ids[qoff + rank] = id;
			out_dists[qoff + rank] = ordered_to_float(sc);
		}
	}
	// fewer than k real entries in all lists together: pad the tail
	for (int off = 32; off > 0; off >>= 1)
		n_valid += __shfl_xor(n_valid, off);
	for (uint32_t r = n_valid + lane; r < k; r += kWave) {
		out_ids[qoff + r] = ~0ull;
		out_dists[qoff + r] = __builtin_inff();
	}
}

}  // namespace expann
