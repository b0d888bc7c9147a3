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
	uint32_t* cand_cnt;       // [m]
	uint64_t* cand;           // [m][cap] keys (common.hpp: make_key)
	uint32_t cap;
};

template <int D, int TQ, bool IP>
__global__ __launch_bounds__(kBlock) void scan_filter_f32_kernel(ScanParams p) {
	static_assert(D % 16 == 0, "the reference kernels need dim % 16 == 0");
	constexpr int DPL = D / 16;  // dims per lane
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * TQ;
	const float* __restrict__ base = (const float*)p.base;
	const float* __restrict__ queries = (const float*)p.queries;
	const bool level0 = (p.tau == nullptr);

	// query slices -> VGPRs, thresholds -> SGPRs.  Slots past m never match (tau = -inf).
	float q[TQ][DPL];
	float tau[TQ];
#pragma unroll
	for (int j = 0; j < TQ; ++j) {
		const uint32_t qi = (q0 + j < p.m) ? q0 + j : p.m - 1;
#pragma unroll
		for (int t = 0; t < DPL; ++t)
			q[j][t] = queries[(size_t)qi * D + l + 16 * t];
		float tj = level0 ? __builtin_inff() : p.tau[qi];
		tau[j] = (q0 + j < p.m) ? tj : -__builtin_inff();
	}

	const uint32_t g0 = chunk * p.groups_per_block;
	uint32_t g1 = g0 + p.groups_per_block;
	if (g1 > p.n_groups_sel)
		g1 = p.n_groups_sel;

	auto row_of = [&](uint32_t g) -> uint32_t {
		return g * p.group_stride * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	};
	auto load_row = [&](float (&r)[DPL], uint32_t row) {
		const uint32_t rr = row < p.n_rows ? row : p.n_rows - 1;
		const float* src = base + (size_t)rr * D + l;
#pragma unroll
		for (int t = 0; t < DPL; ++t)
			r[t] = src[16 * t];
	};
	auto process = [&](const float (&r)[DPL], uint32_t g) {
		const uint32_t row = row_of(g);
		const bool rvalid = row < p.n_rows;
		float s[TQ];
		bool any = false;
#pragma unroll
		for (int j = 0; j < TQ; ++j) {
			float acc = 0.0f;
#pragma unroll
			for (int t = 0; t < DPL; ++t) {
				if (IP) {
					acc = __builtin_fmaf(q[j][t], r[t], acc);
				} else {
					const float diff = q[j][t] - r[t];
					acc = __builtin_fmaf(diff, diff, acc);
				}
			}
			acc = reduce16_ref_order(acc);
			s[j] = IP ? -acc : acc;
			any |= (s[j] <= tau[j]);
		}
		if (level0) {
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
				if (l == 0 && rvalid && q0 + j < p.m && s[j] <= tau[j]) {
					const uint32_t slot = atomicAdd(&p.cand_cnt[q0 + j], 1u);
					if (slot < p.cap)
						p.cand[(size_t)(q0 + j) * p.cap + slot] = make_key(s[j], row);
				}
			}
		}
	};

	// two row buffers: the loads of step g+1 / g+2 are in flight while step g computes
	float ra[DPL], rb[DPL];
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
}

}  // namespace expann
