// scan_int8.hpp -- threshold-filter scan over 8-bit rows (uint8 "compressed" rows and int8
// quantised rows), exact integer arithmetic.
//
// Scores (all exact 32-bit wrapping integer sums, so any summation order is bit-exact):
//   kU8L2    sum (q_i - b_i)^2, q_i = trunc(query_i) in [0,255], b_i uint8
//            = src/antitopo_engine.h:38-61 (dist2_compressed) with the swizzle of :726-737
//   kI8L2    sum (a_i - b_i)^2 on int8 (what src/distance.h:29-53 intended)
//   kI8L2Ref sum ((uint8)(a_i - b_i))^2: src/distance.h:29-53 bit for bit (8-bit wrapping
//            subtract, ZERO-extended, squared)
//   kI8IP    -(sum a_i*b_i) on int8 (no reference kernel)
// The score handed to the selection is float(int32 score), as in the reference
// (`T d_next = score(next)` with T = float, src/antitopo_engine.h:803).
//
// Layout: same wave geometry as scan_f32.hpp -- 4 DPP rows of 16 lanes, one base row per DPP
// row, lane l owns bytes [l*d/16, (l+1)*d/16) of the row (contiguous, dword loads), partial sums
// meet in a 4-step DPP add.  Per dword: one v_dot4 (u8 or i8).  For the L2 forms the expanded
// identity  sum(a-b)^2 = sum a^2 + sum b^2 - 2 sum ab  is exact in integers.
#pragma once
// (non-template kernels are `static`: this header is included by more than one translation unit)
#include "common.hpp"
#include "scan_f32.hpp"

namespace expann {

// kI16L2Ref: int16 rows, src/distance.h:14-27 bit for bit (distance_compare_avx512f_i32): 16-bit
// wrapping subtract, mullo_epi16 keeps the low 16 bits of the square, madd_epi16 with 1
// sign-extends it and adds -- so a term is wrong once |a_i - b_i| > 181, as in the reference.
// The kernels below are instantiated with D = BYTES per row (2 x the element count).
enum IntMode : int { kU8L2 = 0, kI8L2 = 1, kI8L2Ref = 2, kI8IP = 3, kI16L2Ref = 4 };

template <int MODE> __device__ inline int dot4(int a, int b, int c) {
	if (MODE == kU8L2 || MODE == kI8L2Ref)
		return (int)__builtin_amdgcn_udot4((unsigned)a, (unsigned)b, (unsigned)c, false);
	return __builtin_amdgcn_sdot4(a, b, c, false);
}

// per-byte wrapping a - b of four packed bytes (no borrow across bytes)
__device__ inline int sub_bytes(int a, int b) {
	const unsigned ua = (unsigned)a, ub = (unsigned)b;
	const unsigned d = (ua | 0x80808080u) - (ub & 0x7F7F7F7Fu);
	return (int)(d ^ ((ua ^ ~ub) & 0x80808080u));
}

// (a - b) and (d * d) on two packed int16, low 16 bits each (v_pk_sub_i16 / v_pk_mul_lo_u16)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ inline int i16_refcompat_term(int a, int b) {
	const s16x2 d = __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b);
	const s16x2 sq = d * d;
	return (int)sq.x + (int)sq.y;  // sign-extended, as madd_epi16(1, sq)
}

// the integer score of one (query, row) pair restricted to this lane's dwords; the caller
// reduces over the 16 lanes and adds the query-only term
template <int MODE, int NW>
__device__ inline int partial_score(const int (&q)[NW], const int (&b)[NW], int bself) {
	int acc = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w) {
		if (MODE == kI16L2Ref) {
			acc += i16_refcompat_term(q[w], b[w]);
		} else if (MODE == kI8L2Ref) {
			const int d = sub_bytes(q[w], b[w]);
			acc = dot4<MODE>(d, d, acc);
		} else {
			acc = dot4<MODE>(q[w], b[w], acc);
		}
	}
	if (MODE == kU8L2 || MODE == kI8L2)
		return bself - 2 * acc;  // + sum q^2 after the reduction
	if (MODE == kI8IP)
		return -acc;
	return acc;
}

template <int D, int TQ, int MODE>
__global__ __launch_bounds__(kBlock) void scan_filter_i8_kernel(ScanParams p) {
	static_assert(D % 64 == 0, "the reference kernels need dim % 64 == 0");
	constexpr int NW = D / 64;  // dwords per lane
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * TQ;
	const int* __restrict__ base = (const int*)p.base;
	const int* __restrict__ queries = (const int*)p.queries;
	const bool level0 = (p.tau == nullptr);

	int q[TQ][NW];
	int qself[TQ];  // sum q^2 over the whole query (L2 forms)
	float tau[TQ];
#pragma unroll
	for (int j = 0; j < TQ; ++j) {
		const uint32_t qi = (q0 + j < p.m) ? q0 + j : p.m - 1;
		int self = 0;
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			q[j][w] = queries[(size_t)qi * (D / 4) + l * NW + w];
			self = dot4<MODE>(q[j][w], q[j][w], self);
		}
		qself[j] = (MODE == kU8L2 || MODE == kI8L2) ? reduce16_i32(self) : 0;
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
	auto load_row = [&](int (&r)[NW], uint32_t row) {
		const uint32_t rr = row < p.n_rows ? row : p.n_rows - 1;
		const int* src = base + (size_t)rr * (D / 4) + l * NW;
#pragma unroll
		for (int w = 0; w < NW; ++w)
			r[w] = src[w];
	};
	uint64_t best[TQ];  // class minima (classmin level, scan_f32.hpp)
#pragma unroll
	for (int j = 0; j < TQ; ++j)
		best[j] = kSentinelKey;
	auto process = [&](const int (&r)[NW], uint32_t g) {
		const uint32_t row = row_of(g);
		const bool rvalid = row < p.n_rows;
		int bself = 0;
		if (MODE == kU8L2 || MODE == kI8L2) {
#pragma unroll
			for (int w = 0; w < NW; ++w)
				bself = dot4<MODE>(r[w], r[w], bself);
		}
		float s[TQ];
		unsigned long long any = 0;
#pragma unroll
		for (int j = 0; j < TQ; ++j) {
			const int part = partial_score<MODE, NW>(q[j], r, bself);
			const int tot = reduce16_i32(part) + qself[j];
			s[j] = (float)tot;
			any |= __builtin_amdgcn_ballot_w64(s[j] <= tau[j]);
		}
		if (level0 && p.classmin) {
#pragma unroll
			for (int j = 0; j < TQ; ++j) {
				const uint64_t key = rvalid ? make_key(s[j], row) : kSentinelKey;
				best[j] = key < best[j] ? key : best[j];
			}
		} else if (level0) {
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
				if (l == 0 && rvalid && q0 + j < p.m &&
				    (s[j] < tau[j] || (s[j] == tau[j] && row <= p.tau_row[q0 + j]))) {
					const uint32_t slot = atomicAdd(&p.cand_cnt[q0 + j], 1u);
					if (slot < p.cap)
						p.cand[(size_t)(q0 + j) * p.cap + slot] = make_key(s[j], row);
				}
			}
		}
	};

	int ra[NW], rb[NW];
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

// fp32 query -> uint8 (trunc), as `uint32_t(q[i])` in src/antitopo_engine.h:726-737; values
// outside [0,255] cannot be represented in the 8-bit kernels: *bad counts them.  *frac (optional)
// counts in-range values with a fractional part (the exact-uint8 shortcut of fp32 indexes needs
// integer queries; the uint8 engine itself truncates like the reference).
static __global__ __launch_bounds__(kBlock) void u8_query_prep_kernel(const float* q, size_t n_values,
                                                               uint8_t* out, uint32_t* bad,
                                                               uint32_t* frac) {
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_values)
		return;
	const float v = q[i];
	const bool ok = v >= 0.0f && v < 256.0f;
	if (!ok)
		atomicAdd(bad, 1u);
	else if (frac && v != __builtin_truncf(v))
		atomicAdd(frac, 1u);
	out[i] = ok ? (uint8_t)(uint32_t)v : 0;
}

// count of values that are not integers in [0, 255] (grid-stride; one atomic per workgroup at most)
static __global__ __launch_bounds__(kBlock) void count_non_u8_kernel(const float* x, size_t n, uint32_t* bad) {
	uint32_t c = 0;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
		const float v = x[i];
		c += (v >= 0.0f && v <= 255.0f && v == __builtin_truncf(v)) ? 0u : 1u;
	}
	if (__builtin_amdgcn_ballot_w64(c != 0) != 0 && c != 0)
		atomicAdd(bad, c);
}
// fp32 (known to hold integers in [0, 255]) -> uint8
static __global__ __launch_bounds__(kBlock) void cast_f32_u8_kernel(const float* x, size_t n, uint8_t* out) {
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
		out[i] = (uint8_t)(uint32_t)x[i];
}

// scores[i] = score(query, base[ids[i]]) on 8-bit rows (cf. score_ids.hpp)
struct ScoreIdsI8Params {
	const void* base;
	const void* query;  // [D] bytes
	const uint64_t* ids;
	uint64_t id_offset;
	uint32_t n_ids;
	float* scores;
};
template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void score_ids_i8_kernel(ScoreIdsI8Params p) {
	constexpr int NW = D / 64;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t i = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const bool valid = i < p.n_ids;
	const uint64_t row = p.ids[valid ? i : p.n_ids - 1] - p.id_offset;
	const int* r = (const int*)p.base + (size_t)row * (D / 4) + l * NW;
	const int* qq = (const int*)p.query + l * NW;
	int q[NW], b[NW];
	int qself = 0, bself = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w) {
		q[w] = qq[w];
		b[w] = r[w];
		if (MODE == kU8L2 || MODE == kI8L2) {
			qself = dot4<MODE>(q[w], q[w], qself);
			bself = dot4<MODE>(b[w], b[w], bself);
		}
	}
	const int part = partial_score<MODE, NW>(q, b, bself) + qself;
	const int tot = reduce16_i32(part);
	if (valid && l == 0)
		p.scores[i] = (float)tot;
}

}  // namespace expann
