// common.hpp -- shared device helpers for the expann HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace expann {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves, one per SIMD
constexpr int kRowsPerWaveStep = 4;  // 16 lanes per row (the reference's 16 accumulators)
constexpr int kRowsPerGroup = 16;    // one workgroup step = 4 waves x 4 rows

constexpr uint64_t kSentinelKey = ~0ull;  // sorts after every real (score, id) key

// Monotone map float -> uint32: a < b  <=>  ord(a) < ord(b) for all non-NaN floats
// (negative scores occur for the inner-product metric).  +NaN sorts after +inf.
__host__ __device__ inline uint32_t float_to_ordered(float f) {
	uint32_t u = __builtin_bit_cast(uint32_t, f);
	return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float ordered_to_float(uint32_t o) {
	uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
	return __builtin_bit_cast(float, u);
}
// (score, local row) -> one 64-bit key whose unsigned order is the reference's
// lexicographic (dist, idx) order (std::pair<T,size_t>::operator<, src/topk_t.h:11).
__host__ __device__ inline uint64_t make_key(float score, uint32_t idx) {
	return ((uint64_t)float_to_ordered(score) << 32) | idx;
}
__host__ __device__ inline float key_score(uint64_t key) {
	return ordered_to_float((uint32_t)(key >> 32));
}
__host__ __device__ inline uint32_t key_idx(uint64_t key) { return (uint32_t)key; }

// DPP rotate-right within each row of 16 lanes: lane i reads lane (i - N) mod 16.
template <int N> __device__ inline float row_ror(float v) {
	static_assert(N >= 1 && N <= 15, "row_ror");
	int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false);
	return __builtin_bit_cast(float, r);
}
template <int N> __device__ inline int row_ror_i(int v) {
	return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false);
}

// The _mm512_reduce_add_ps tree (src/distance.h:146) over the 16 lanes of a DPP row:
// (l, l+8), then +4, then +2, then +1.  Every lane ends with the same bits because IEEE
// addition is commutative.
__device__ inline float reduce16_ref_order(float acc) {
	acc = acc + row_ror<8>(acc);
	acc = acc + row_ror<4>(acc);
	acc = acc + row_ror<2>(acc);
	acc = acc + row_ror<1>(acc);
	return acc;
}
__device__ inline int reduce16_i32(int acc) {
	acc = acc + row_ror_i<8>(acc);
	acc = acc + row_ror_i<4>(acc);
	acc = acc + row_ror_i<2>(acc);
	acc = acc + row_ror_i<1>(acc);
	return acc;
}

}  // namespace expann
