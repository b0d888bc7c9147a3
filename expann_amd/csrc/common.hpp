// common.hpp -- shared device helpers for the expann HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace expann {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves, one per SIMD
constexpr int kRowsPerWaveStep = 4;  // 16 lanes per row (the reference's 16 accumulators)
constexpr int kRowsPerGroup = 16;    // one workgroup step = 4 waves x 4 rows

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- packed fp32 (VOP3P) helpers ----------------------------------------------------
// Plain fp32 VALU ops issue at 16 lanes/clk per SIMD on gfx950 (measured: a wave64
// v_fma_f32 costs 4 cycles at any occupancy); v_pk_*_f32 do two IEEE ops per lane in the
// same 4 cycles.  hipcc scalarises <2 x float> arithmetic in the unrolled scan loops, so the
// packed instructions are spelled out.  Each half is exactly the scalar op (same rounding),
// so results stay bit-identical.  `sel` picks which half of the 64-bit register pair `r`
// is broadcast to both halves (0 = low dword, 1 = high dword).
template <int SEL> __device__ inline f32x2 pk_sub_bcast(f32x2 a, f32x2 r) {  // a - {r[SEL], r[SEL]}
	f32x2 d;
	if (SEL == 0)
		asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
		    : "=v"(d) : "v"(a), "v"(r));
	else
		asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]"
		    : "=v"(d) : "v"(a), "v"(r));
	return d;
}
__device__ inline f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {  // {fma(a.x,b.x,c.x), fma(a.y,b.y,c.y)}
	f32x2 d;
	asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
template <int SEL> __device__ inline f32x2 pk_fma_bcast(f32x2 a, f32x2 r, f32x2 c) {  // fma(a, {r[SEL]..}, c)
	f32x2 d;
	if (SEL == 0)
		asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(r), "v"(c));
	else
		asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
		    : "=v"(d) : "v"(a), "v"(r), "v"(c));
	return d;
}

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
