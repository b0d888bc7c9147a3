// scan_direct_f16.hpp -- the fp16 filter of scan_gemm_f16.hpp for a HANDFUL of queries (the
// reference's own timing mode is one query per call, src/basic_bench.h:82-126): no matrix cores,
// no LDS -- every lane streams its 16-byte chunks of the scaled fp16 rows straight from HBM
// (half the bytes of the fp32 rows the direct scan reads), multiplies them with the query
// slices it keeps in registers (fp32 FMAs on exactly converted fp16 values: one rounding per
// element, covered by gemm_f16_filter_eps(d) like the MFMA accumulation) and the LPR lanes of a
// row combine their partial sums by DPP.  Same test as the MFMA form -- theta' + q16.b16 >= bn'
// -- same approximate keys, same exact re-rank in the select kernel, so ids and distances stay
// bit-identical.  HBM-bound: 2d bytes per row and query tile.
//
// Lanes per row: 16 where a row is a multiple of 16 chunks (d = 128, 256, 512, 768), else 8
// (d = 64, 832, 960); a wave covers 64 / LPR consecutive rows per step = one contiguous stretch.
#pragma once
#include "scan_gemm_f16.hpp"

namespace expann {

template <int CTRL> __device__ inline float dpp_f32(float v) {
	const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
	return __builtin_bit_cast(float, r);
}
// sum over the LPR (8 or 16) lanes of a row group; every lane ends with the total
template <int LPR> __device__ inline float reduce_lanes(float v) {
	v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
	v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
	v += dpp_f32<0x141>(v);  // row_half_mirror: the other quad of the 8 lanes
	if (LPR == 16)
		v += dpp_f32<0x140>(v);  // row_mirror: the other half of the 16 lanes
	return v;
}

template <int D> struct DirectF16Geom {
	static constexpr int CH = D / 8;                       // 16-byte chunks per row
	static constexpr int LPR = (CH % 16 == 0) ? 16 : 8;    // lanes per row
	static constexpr int CPL = CH / LPR;                   // chunks per lane
	static constexpr int RW = 64 / LPR;                    // rows per wave step
	static constexpr int RPS = RW * (kBlock / 64);         // rows per workgroup step
	static constexpr int NB = (RW * D * 2 >= 2048) ? 2 : 4;  // row buffers in flight per wave
	static_assert(CH % 8 == 0 && 64 % RPS == 0, "rows are multiples of 64 dims; steps tile the 64-row padding");
};

// GemmF16Params: n_tiles_sel / tiles_per_block count RPS-row steps here; tile_stride, tile_run,
// xcd_map and the SAMPLE fields are unused
template <int D, int TQ>
__global__ __launch_bounds__(kBlock) void scan_direct_f16_kernel(GemmF16Params p) {
	using G = DirectF16Geom<D>;
	constexpr int LPR = G::LPR, CPL = G::CPL, RW = G::RW, RPS = G::RPS, NB = G::NB;
	constexpr bool QF32 = D <= 512;  // query slices as fp32 pairs in registers (else fp16, converted per use)
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int l = lane % LPR, rg = lane / LPR;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * TQ;

	f32x2 q2[QF32 ? TQ : 1][CPL][4];
	f16x8 qh[QF32 ? 1 : TQ][CPL];
	float theta[TQ];
#pragma unroll
	for (int j = 0; j < TQ; ++j) {
		const uint32_t qi = q0 + j < p.m ? q0 + j : p.m - 1;
		const f16x8* src = reinterpret_cast<const f16x8*>((const _Float16*)p.queries_f16 + (size_t)qi * D) + l;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const f16x8 v = src[LPR * c];
			if (QF32) {
#pragma unroll
				for (int e = 0; e < 4; ++e)
					q2[j][c][e] = f32x2{(float)v[2 * e], (float)v[2 * e + 1]};
			} else {
				qh[j][c] = v;
			}
		}
		const float tj = q0 + j < p.m ? p.theta[qi] : -__builtin_inff();
		theta[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tj)));
	}

	const uint32_t g0 = chunk * p.tiles_per_block;
	uint32_t g1 = g0 + p.tiles_per_block;
	if (g1 > p.n_tiles_sel)
		g1 = p.n_tiles_sel;
	auto row_of = [&](uint32_t g) -> uint32_t { return g * RPS + wave * RW + rg; };
	struct RowBuf {
		f16x8 r[CPL];
		float bn;
	};
	auto load_row = [&](RowBuf& b, uint32_t g) {
		const uint32_t row = row_of(g);  // (inside the 64-row padding of the fp16 copy: bn' = NaN there)
		const f16x8* src = reinterpret_cast<const f16x8*>((const _Float16*)p.base_f16 + (size_t)row * D) + l;
#pragma unroll
		for (int c = 0; c < CPL; ++c)
			b.r[c] = src[LPR * c];
		b.bn = p.bnorm[row];
	};
	auto process = [&](const RowBuf& b, uint32_t g) {
		f32x2 acc[TQ];
#pragma unroll
		for (int j = 0; j < TQ; ++j)
			acc[j] = f32x2{0.0f, 0.0f};
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			f32x2 rf[4];
#pragma unroll
			for (int e = 0; e < 4; ++e)
				rf[e] = f32x2{(float)b.r[c][2 * e], (float)b.r[c][2 * e + 1]};
#pragma unroll
			for (int j = 0; j < TQ; ++j)
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const f32x2 qv = QF32 ? q2[QF32 ? j : 0][c][e]
					                      : f32x2{(float)qh[QF32 ? 0 : j][c][2 * e], (float)qh[QF32 ? 0 : j][c][2 * e + 1]};
					acc[j] = pk_fma(qv, rf[e], acc[j]);
				}
		}
		float cv[TQ];
		unsigned long long any = 0;
#pragma unroll
		for (int j = 0; j < TQ; ++j) {
			cv[j] = theta[j] + reduce_lanes<LPR>(acc[j][0] + acc[j][1]);
			any |= __builtin_amdgcn_ballot_w64(cv[j] >= b.bn);
		}
		if (any) {
			const uint32_t row = row_of(g);
#pragma unroll
			for (int j = 0; j < TQ; ++j)
				if (l == 0 && cv[j] >= b.bn) {  // (slots past m: theta = -inf never passes)
					const uint32_t qi = q0 + j;
					const uint32_t slot = atomicAdd(&p.cand_cnt[qi], 1u);
					if (slot < p.cap)
						p.cand[(size_t)qi * p.cap + slot] =
						    make_key(((b.bn - cv[j]) + theta[j]) * p.two_inv_s2, row);
				}
		}
	};

	// NB row buffers: the loads of the next NB - 1 steps are in flight while one step computes
	RowBuf buf[NB];
#pragma unroll
	for (int i = 0; i < NB; ++i)
		if (g0 + i < g1)
			load_row(buf[i], g0 + i);
	uint32_t g = g0;
	for (; g + NB <= g1; g += NB) {
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			process(buf[i], g + i);
			if (g + NB + i < g1)
				load_row(buf[i], g + NB + i);
		}
	}
#pragma unroll
	for (int i = 0; i < NB; ++i)
		if (g + i < g1)
			process(buf[i], g + i);
}

// The sampled pass + threshold of the handful-of-queries path in ONE launch: every workgroup
// streams its share of the sample (RPS-row steps spread evenly over the index), each (wave, row
// group) keeps the maximum of g = q16.b16 - bns over its rows -- one class per lane group, so the
// k largest class maxima belong to k different rows, as in the MFMA SAMPLE pass -- and the LAST
// workgroup to finish (ticket counter, agent-scope fences) turns the class maxima into tau /
// theta' and zeroes the list counters (sample_tau_query_wg: radix select by the whole workgroup).
struct SampleDirectParams {
	GemmF16Params g;    // base_f16, bnorm = UPPER row terms, queries_f16, m; n_tiles_sel = steps of the
	                    // sample, tile_stride = step stride, tiles_per_block = steps per workgroup
	SampleTauParams t;  // vals = class maxima [m][n_vals], n_vals = gridDim.x * classes per workgroup
	uint32_t* ticket;   // 0 on entry; the last workgroup resets it
};
template <int D, int TQ>
__global__ __launch_bounds__(kBlock) void sample_direct_f16_kernel(SampleDirectParams p) {
	using G = DirectF16Geom<D>;
	constexpr int LPR = G::LPR, CPL = G::CPL, RW = G::RW, RPS = G::RPS;
	// only ~128 workgroups run (one class per lane group, <= 2048 classes), so each wave keeps many
	// steps in flight: 16 KB per wave at d <= 128
	constexpr int NB = CPL >= 4 ? 4 : 16 / CPL;
	constexpr int CPW = RW * (kBlock / 64);  // classes per workgroup
	static_assert(TQ <= kBlock / 64, "one wave per query in the threshold step");
	__shared__ uint32_t scratch[kBlock / 64][64];
	__shared__ uint32_t s_ticket[2];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int l = lane % LPR, rg = lane / LPR;
	const uint32_t chunk = blockIdx.x;
	if (p.g.clk && threadIdx.x == 0 && blockIdx.x == 0)
		p.g.clk[0] = wall_clock64();  // (debug: 100 MHz ticks at the first workgroup's start)

	f32x2 q2[TQ][CPL][4];
#pragma unroll
	for (int j = 0; j < TQ; ++j) {
		const uint32_t qi = (uint32_t)j < p.g.m ? j : p.g.m - 1;
		const f16x8* src = reinterpret_cast<const f16x8*>((const _Float16*)p.g.queries_f16 + (size_t)qi * D) + l;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const f16x8 v = src[LPR * c];
#pragma unroll
			for (int e = 0; e < 4; ++e)
				q2[j][c][e] = f32x2{(float)v[2 * e], (float)v[2 * e + 1]};
		}
	}
	const uint32_t g0 = chunk * p.g.tiles_per_block;
	uint32_t g1 = g0 + p.g.tiles_per_block;
	if (g1 > p.g.n_tiles_sel)
		g1 = p.g.n_tiles_sel;
	auto row_of = [&](uint32_t g) -> uint32_t { return g * p.g.tile_stride * RPS + wave * RW + rg; };
	struct RowBuf {
		f16x8 r[CPL];
		float bn;
	};
	auto load_row = [&](RowBuf& b, uint32_t g) {
		const uint32_t row = row_of(g);
		const f16x8* src = reinterpret_cast<const f16x8*>((const _Float16*)p.g.base_f16 + (size_t)row * D) + l;
#pragma unroll
		for (int c = 0; c < CPL; ++c)
			b.r[c] = src[LPR * c];
		b.bn = p.g.bnorm[row];
	};
	float best[TQ];
#pragma unroll
	for (int j = 0; j < TQ; ++j)
		best[j] = -__builtin_inff();
	auto process = [&](const RowBuf& b) {
		f32x2 acc[TQ];
#pragma unroll
		for (int j = 0; j < TQ; ++j)
			acc[j] = f32x2{0.0f, 0.0f};
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			f32x2 rf[4];
#pragma unroll
			for (int e = 0; e < 4; ++e)
				rf[e] = f32x2{(float)b.r[c][2 * e], (float)b.r[c][2 * e + 1]};
#pragma unroll
			for (int j = 0; j < TQ; ++j)
#pragma unroll
				for (int e = 0; e < 4; ++e)
					acc[j] = pk_fma(q2[j][c][e], rf[e], acc[j]);
		}
#pragma unroll
		for (int j = 0; j < TQ; ++j)  // a NaN bns (padding row) never wins the max
			best[j] = __builtin_fmaxf(best[j], reduce_lanes<LPR>(acc[j][0] + acc[j][1]) - b.bn);
	};
	RowBuf buf[NB];
#pragma unroll
	for (int i = 0; i < NB; ++i)
		if (g0 + i < g1)
			load_row(buf[i], g0 + i);
	uint32_t g = g0;
	for (; g + NB <= g1; g += NB) {
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			process(buf[i]);
			if (g + NB + i < g1)
				load_row(buf[i], g + NB + i);
		}
	}
#pragma unroll
	for (int i = 0; i < NB; ++i)
		if (g + i < g1)
			process(buf[i]);
	if (l == 0) {
#pragma unroll
		for (int j = 0; j < TQ; ++j)
			if ((uint32_t)j < p.g.m)
				const_cast<float*>(p.t.vals)[(size_t)j * p.t.n_vals + chunk * CPW + wave * RW + rg] = best[j];
	}
	// last workgroup: thresholds (the class maxima of the others are visible after the fences)
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0)
		s_ticket[0] = atomicAdd(p.ticket, 1u);
	__syncthreads();
	if (s_ticket[0] != gridDim.x - 1)
		return;
	__threadfence();
	if (p.g.clk && threadIdx.x == 0)
		p.g.clk[1] = wall_clock64();
	if (p.g.m <= 2) {  // the whole workgroup per query (a lone wave's selection costs ~15 us)
		for (uint32_t qi = 0; qi < p.g.m; ++qi)
			sample_tau_query_wg(p.t, qi, &scratch[0][0], s_ticket);
	} else if ((uint32_t)wave < p.g.m) {  // one wave per query, side by side
		sample_tau_query<32>(p.t, (uint32_t)wave, lane, scratch[wave]);
	}
	if (p.g.clk && threadIdx.x == 0)
		p.g.clk[2] = wall_clock64();
	if (threadIdx.x == 0)
		*p.ticket = 0;
}

}  // namespace expann
