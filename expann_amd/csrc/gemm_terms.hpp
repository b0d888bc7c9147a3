// gemm_terms.hpp -- the row and query terms of the expanded squared-L2 test
//
//        ||b||^2 (1-eps)  -  2 q.b   <=   tau_q  -  ||q||^2 (1-eps)
//
// that the GEMM-form candidate filters evaluate (scan_gemm_bf16.hpp; the fp16 forms carry their own
// scaled terms), and the launch geometry the 8-wave kernels share.  The slack eps dominates every
// rounding error of the expanded evaluation AND of the reference-order evaluation, so
// every row whose reference-order score is <= tau_q survives; survivors are re-scored in the reference's
// exact 16-lane FMA order (select.hpp) before the (score, id) selection.
// (Rounds 1-2 also carried the same filter on v_mfma_f32_32x32x2_f32 -- exact fp32 products at 1/16 of
// the fp16 rate, 0.41 M QPS at C2 against 4.4 M -- as a selectable form; it is gone since round 3.)
#pragma once
#include "common.hpp"

namespace expann {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmTQ = 128;  // queries per workgroup
constexpr int kGemmTB = 128;  // base rows per step

__host__ __device__ inline float gemm_filter_eps(int d) {
	return (float)(4 * d + 128) * 5.9604644775390625e-08f;  // (4d+128) * 2^-24
}

struct GemmScanParams {
	const float* base;       // [n_rows][D]
	const float* bnorm;      // [n_rows] ||b||^2 * (1 - eps)
	uint32_t n_rows;
	uint32_t n_tiles_sel;    // 128-row tiles this launch visits ...
	uint32_t tile_stride;    // ... tile j of the launch is base tile j*tile_stride
	uint32_t tiles_per_block;
	uint32_t n_qtiles;
	const float* queries;    // [m][D]
	const float* theta;      // [m] tau_q - ||q||^2 (1 - eps)
	uint32_t m;
	uint32_t* cand_cnt;      // [m]
	uint64_t* cand;          // [m][cap]; low 32 bits = row, high = ordered(approx score)
	uint32_t cap;
};

// ||b||^2 (1-eps) per row; 16 lanes per row.
template <int D>
__global__ __launch_bounds__(kBlock) void row_norms_kernel(const float* base, uint32_t n_rows,
                                                           float scale, float* out) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t row = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const uint32_t rr = row < n_rows ? row : n_rows - 1;
	const float* src = base + (size_t)rr * D + l;
	float acc = 0.0f;
#pragma unroll
	for (int t = 0; t < D / 16; ++t)
		acc = __builtin_fmaf(src[16 * t], src[16 * t], acc);
	acc = reduce16_ref_order(acc);
	if (row < n_rows && l == 0)
		out[row] = acc * scale;
}

// theta_q = tau_q - ||q||^2 (1-eps); 16 lanes per query.
template <int D>
__global__ __launch_bounds__(kBlock) void query_theta_kernel(const float* queries, uint32_t m,
                                                             const float* tau, float scale,
                                                             float* theta) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t qi = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const uint32_t qq = qi < m ? qi : m - 1;
	const float* src = queries + (size_t)qq * D + l;
	float acc = 0.0f;
#pragma unroll
	for (int t = 0; t < D / 16; ++t)
		acc = __builtin_fmaf(src[16 * t], src[16 * t], acc);
	acc = reduce16_ref_order(acc);
	if (qi < m && l == 0)
		theta[qi] = tau[qi] - acc * scale;
}

constexpr int kGemmThreads = 512;  // 8 waves, 2 per SIMD

}  // namespace expann
