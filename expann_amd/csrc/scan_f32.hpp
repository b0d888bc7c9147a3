// scan_f32.hpp -- fp32 threshold-filter scan of the base set (gfx950).
//
// One launch streams (a strided sample of) the base rows once per query tile and keeps,
// for every query of the tile, the rows whose score is <= that query's threshold tau.
// The scores are bit-identical to the reference's AVX-512 kernels:
//   L2: src/distance.h:136-147 (== :86-111, the live kernel of src/antitopo_engine.h:25-37)
//   IP: src/distance.h:181-190 (score = -dot)
// because the wavefront is cut into 4 DPP rows of 16 lanes and lane l of a row plays the
// role of AVX-512 lane l: it owns dims l, l+16, l+32, ... of ONE base row, accumulates them
// in increasing order with v_fma_f32, and the 16 partial sums are combined by the same
// tree as _mm512_reduce_add_ps (common.hpp: reduce16_ref_order).
//
// Data movement: every lane reads its dims straight from HBM into VGPRs
// (global_load_dword, 16 lanes = one 64-B segment, a wave = 4 rows x D*4 B contiguous);
// the TQ query slices a lane needs (TQ x D/16 floats) live in VGPRs for the whole launch,
// so a base row costs exactly one HBM/L2 read per query tile and no LDS traffic.
// Thresholds are wave-uniform (SGPRs).  Candidates are rare (the host picks thresholds so
// that ~32k of n rows pass) and are appended to a per-query list with one global atomic.
#pragma once
#include "common.hpp"

namespace expann {

struct ScanParams {
	const void* base;         // [n_rows][D]
	uint32_t n_rows;
	uint32_t n_groups_sel;    // groups (of 16 rows) this launch visits ...
	uint32_t group_stride;    // ... group j of the launch is base group j*group_stride
	uint32_t groups_per_block;
	uint32_t n_qtiles;
	const void* queries;      // [m][D]
	uint32_t m;
	const float* tau;         // [m] thresholds, or nullptr: keep everything (level 0)
	const uint32_t* tau_row;  // [m] row of the threshold key: (score, row) <= (tau, tau_row) passes
	uint32_t* cand_cnt;       // [m]
	uint64_t* cand;           // [m][cap] keys (common.hpp: make_key)
	uint32_t cap;
	// level 0 only (tau == nullptr): instead of every key of the sample, keep the SMALLEST key of
	// each class = (workgroup, wave, 16-lane row group), at slot workgroup*16 + wave*4 + group.
	// The k smallest class minima are k different rows, so their k-th is a valid (score, row)
	// threshold -- from a 16x larger sample than a keep-all level can afford, in one launch.
	uint32_t classmin;
};

template <int D, int TQ, bool IP>
__global__ __launch_bounds__(kBlock) void scan_filter_f32_kernel(ScanParams p) {
	static_assert(D % 32 == 0, "the reference kernels need dim % 16 == 0");
	constexpr int DPL = D / 16;  // dims per lane
	constexpr int NP = (TQ + 1) / 2;  // query pairs
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * TQ;
	const float* __restrict__ base = (const float*)p.base;
	const float* __restrict__ queries = (const float*)p.queries;
	const bool level0 = (p.tau == nullptr);

	// query slices -> VGPRs (two queries per register pair, so the per-dim work is packed
	// v_pk_add_f32 / v_pk_fma_f32: plain fp32 VALU ops issue at 16 lanes/clk on this chip,
	// packed ones at 2 x 16), thresholds -> SGPRs.  Slots past m never match (tau = -inf).
	f32x2 q2[NP][DPL];
	float tau[TQ];
#pragma unroll
	for (int pq = 0; pq < NP; ++pq) {
		const uint32_t qa = (q0 + 2 * pq < p.m) ? q0 + 2 * pq : p.m - 1;
		const uint32_t qb = (q0 + 2 * pq + 1 < p.m) ? q0 + 2 * pq + 1 : p.m - 1;
#pragma unroll
		for (int t = 0; t < DPL; ++t)
			q2[pq][t] = f32x2{queries[(size_t)qa * D + l + 16 * t],
			                  queries[(size_t)qb * D + l + 16 * t]};
	}
#pragma unroll
	for (int j = 0; j < TQ; ++j) {
		const uint32_t qi = (q0 + j < p.m) ? q0 + j : p.m - 1;
		float tj = level0 ? __builtin_inff() : p.tau[qi];
		tj = (q0 + j < p.m) ? tj : -__builtin_inff();
		tau[j] = __builtin_bit_cast(float,
		                            __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tj)));
	}

	const uint32_t g0 = chunk * p.groups_per_block;
	uint32_t g1 = g0 + p.groups_per_block;
	if (g1 > p.n_groups_sel)
		g1 = p.n_groups_sel;

	auto row_of = [&](uint32_t g) -> uint32_t {
		return g * p.group_stride * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	};
	// a lane's dims t and t+1 share one 64-bit register pair (op_sel picks the half)
	auto load_row = [&](f32x2 (&r)[DPL / 2], uint32_t row) {
		const uint32_t rr = row < p.n_rows ? row : p.n_rows - 1;
		const float* src = base + (size_t)rr * D + l;
#pragma unroll
		for (int t = 0; t < DPL; t += 2)
			r[t / 2] = f32x2{src[16 * t], src[16 * t + 16]};
	};
	uint64_t best[TQ];  // class minima (classmin level)
#pragma unroll
	for (int j = 0; j < TQ; ++j)
		best[j] = kSentinelKey;
	auto process = [&](const f32x2 (&r)[DPL / 2], uint32_t g) {
		const uint32_t row = row_of(g);
		const bool rvalid = row < p.n_rows;
		// per-pair accumulators; each component is one (row, query) chain of the reference:
		// acc = fma(diff, diff, acc) over this lane's dims in increasing order
		f32x2 acc[NP];
		const f32x2 zero = f32x2{0.0f, 0.0f};
#pragma unroll
		for (int pq = 0; pq < NP; ++pq)
			acc[pq] = zero;
#pragma unroll
		for (int t = 0; t < DPL; t += 2) {
#pragma unroll
			for (int pq = 0; pq < NP; ++pq) {
				if (IP) {
					acc[pq] = pk_fma_bcast<0>(q2[pq][t], r[t / 2], acc[pq]);
				} else {
					const f32x2 diff = pk_sub_bcast<0>(q2[pq][t], r[t / 2]);
					acc[pq] = pk_fma(diff, diff, acc[pq]);
				}
			}
#pragma unroll
			for (int pq = 0; pq < NP; ++pq) {
				if (IP) {
					acc[pq] = pk_fma_bcast<1>(q2[pq][t + 1], r[t / 2], acc[pq]);
				} else {
					const f32x2 diff = pk_sub_bcast<1>(q2[pq][t + 1], r[t / 2]);
					acc[pq] = pk_fma(diff, diff, acc[pq]);
				}
			}
		}
		float s[TQ];
		unsigned long long any = 0;
#pragma unroll
		for (int j = 0; j < TQ; ++j) {
			const float red = reduce16_ref_order(acc[j >> 1][j & 1]);
			s[j] = IP ? -red : red;
			any |= __builtin_amdgcn_ballot_w64(s[j] <= tau[j]);
		}
		if (level0 && p.classmin) {
#pragma unroll
			for (int j = 0; j < TQ; ++j) {
				const uint64_t key = rvalid ? make_key(s[j], row) : kSentinelKey;
				best[j] = key < best[j] ? key : best[j];
			}
		} else if (level0) {
			// keep every row of the sample at its sample position (no atomics)
			const uint32_t slot = g * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
			if (l == 0 && slot < p.cap) {
#pragma unroll
				for (int j = 0; j < TQ; ++j)
					if (q0 + j < p.m)
						p.cand[(size_t)(q0 + j) * p.cap + slot] =
						    rvalid ? make_key(s[j], row) : kSentinelKey;
			}
		} else if (any) {
#pragma unroll
			for (int j = 0; j < TQ; ++j) {
				// exact lexicographic key test (only evaluated on the rare path): ties with the
				// threshold score pass only up to the threshold's row, which bounds the survivors
				// even when many rows are identical
				if (l == 0 && rvalid && q0 + j < p.m &&
				    (s[j] < tau[j] || (s[j] == tau[j] && row <= p.tau_row[q0 + j]))) {
					const uint32_t slot = atomicAdd(&p.cand_cnt[q0 + j], 1u);
					if (slot < p.cap)
						p.cand[(size_t)(q0 + j) * p.cap + slot] = make_key(s[j], row);
				}
			}
		}
	};

	// two row buffers: the loads of step g+1 / g+2 are in flight while step g computes
	f32x2 ra[DPL / 2], rb[DPL / 2];
	uint32_t g = g0;
	if (g < g1)
		load_row(ra, row_of(g));
	if (g + 1 < g1)
		load_row(rb, row_of(g + 1));
	for (; g + 1 < g1; g += 2) {
		process(ra, g);
		if (g + 2 < g1)
			load_row(ra, row_of(g + 2));
		process(rb, g + 1);
		if (g + 3 < g1)
			load_row(rb, row_of(g + 3));
	}
	if (g < g1)
		process(ra, g);
	if (level0 && p.classmin) {
		const uint32_t slot = chunk * 16 + wave * 4 + rg;
		if (l == 0 && slot < p.cap) {
#pragma unroll
			for (int j = 0; j < TQ; ++j)
				if (q0 + j < p.m)
					p.cand[(size_t)(q0 + j) * p.cap + slot] = best[j];
		}
	}
}

}  // namespace expann
