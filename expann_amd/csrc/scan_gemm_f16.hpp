// scan_gemm_f16.hpp -- the GEMM-form fp32 L2 candidate filter with ONE fp16 product per element
// on the matrix cores (v_mfma_f32_32x32x16_f16): a third of the MFMA work of the bf16x3 form
// (scan_gemm_bf16.hpp) and half its tile bytes.
//
// Every value is scaled by a per-index power of two s (so that max|x|*s <= 2^15) and rounded
// once to fp16:  |x*s - fp16(x*s)| <= 2^-11 |x*s| + 2^-25  (normal + subnormal range).  Hence
//   |q.b - q16.b16 / s^2|  <=  (2^-10 + 2^-22) sum|q_i b_i|  +  2^-25/s * sum(|q_i| + |b_i|)
// and the squared-L2 test in expanded form keeps every row whose reference-order score is
// <= tau_q when evaluated with the slack
//      1.125 * 2^-10 * (||q||^2 + ||b||^2)  +  2^-24/s * sqrt(d) * (|q| + |b|)
// (the 12.5 % on top of 2^-10 covers the fp32 accumulation, the norms and the reference-order
// rounding, (4d+226) * 2^-24 at most).  The slack admits ~10 % more candidates than the exact
// test; they are re-scored exactly by the select kernel, so ids and distances stay bit-identical
// to the direct scan.  Queries whose scaled components would leave the fp16 range make the
// engine use the bf16x3 form for that search.
//
// Layout / geometry: as scan_gemm_bf16.hpp with one plane: rows of d fp16 (2d bytes), 256
// queries x 128 rows per workgroup step, 8 waves x (32 queries x 4 MFMA tiles), 32 MFMAs per
// wave-step.
#pragma once
#include "common.hpp"
#include "scan_gemm_bf16.hpp"

namespace expann {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline float gemm_f16_filter_eps() { return 1.125f * 0.0009765625f; }

// fp32 [n_values] * scale -> fp16 (round to nearest even)
__global__ __launch_bounds__(kBlock) void convert_f16_kernel(const float* in, size_t n_values,
                                                             float scale, _Float16* out) {
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if (i < n_values)
		out[i] = (_Float16)(in[i] * scale);
}

// max |x| over an array (single workgroup; NaN never wins)
__global__ __launch_bounds__(1024) void maxabs_f32_kernel(const float* in, size_t n, float* out) {
	__shared__ float red[16];
	float m = 0.0f;
	for (size_t i = threadIdx.x; i < n; i += 1024) {
		const float v = __builtin_fabsf(in[i]);
		m = v > m ? v : m;
	}
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

// out[i] = ||x_i||^2 (fp32, reference lane order); 16 lanes per row
template <int D>
__global__ __launch_bounds__(kBlock) void sqnorm_kernel(const float* x, uint32_t n, float* out) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t i = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const uint32_t ii = i < n ? i : n - 1;
	const float* src = x + (size_t)ii * D + l;
	float acc = 0.0f;
#pragma unroll
	for (int t = 0; t < D / 16; ++t)
		acc = __builtin_fmaf(src[16 * t], src[16 * t], acc);
	acc = reduce16_ref_order(acc);
	if (i < n && l == 0)
		out[i] = acc;
}
// rows:    out = nrm*(1-eps) - abs*sqrt(nrm)                (tau == nullptr)
// queries: out = tau - (nrm*(1-eps) - abs*sqrt(nrm))
__global__ __launch_bounds__(kBlock) void f16_terms_kernel(const float* nrm, uint32_t n, float eps,
                                                           float abs_coef, const float* tau,
                                                           float* out) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float v = nrm[i];
	const float t = v * (1.0f - eps) - abs_coef * __builtin_sqrtf(v);
	out[i] = tau ? tau[i] - t : t;
}

struct GemmF16Params {
	const void* base_f16;    // [n_rows][D] fp16, scaled by s
	const float* bnorm;      // [n_rows] ||b||^2 (1-eps) - abs*|b|
	uint32_t n_rows;
	uint32_t n_tiles_sel;
	uint32_t tile_stride;
	uint32_t tile_run;
	uint32_t tiles_per_block;
	uint32_t n_qtiles;
	const void* queries_f16; // [m][D] fp16, scaled by s
	const float* theta;      // [m] tau - ||q||^2 (1-eps) + abs*|q|
	float neg2_inv_s2;       // -2 / s^2 (exact: s is a power of two)
	uint32_t m;
	uint32_t* cand_cnt;
	uint64_t* cand;
	uint32_t cap;
};

template <int D>
__global__ __launch_bounds__(kGemmThreads, 2) void scan_gemm_f16_kernel(GemmF16Params p) {
	static_assert(D == 128 || D == 64, "built for d = 64, 128");
	constexpr int ROWB = D * 2;      // bytes per fp16 row
	constexpr int CH = ROWB / 16;    // 16-byte chunks per row
	constexpr int KS = D / 16;       // MFMA k-steps; lane half h of k-step s holds chunk h*KS + s
	constexpr int TILE_BYTES = kGemmTB * ROWB;
	constexpr int RPB = (ROWB < 256) ? 256 / ROWB : 1;  // rows per 256-byte LDS bank row
	constexpr int SWM = (CH < 16 ? CH : 16) - 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = wave;  // each wave: 32 queries x all 128 rows of the tile (4 MFMA tiles)
	const int h = lane >> 5, r31 = lane & 31;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * kGemmBf16TQ;

	f16x8 a[KS];
	{
		uint32_t qi = q0 + wr * 32 + r31;
		if (qi >= p.m)
			qi = p.m - 1;
		const f16x8* src = reinterpret_cast<const f16x8*>((const unsigned char*)p.queries_f16 +
		                                                  (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			a[s] = src[h * KS + s];
	}
	float th[16];
#pragma unroll
	for (int reg = 0; reg < 16; ++reg) {
		const uint32_t qi = q0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
		th[reg] = qi < p.m ? p.theta[qi] : -__builtin_inff();
	}
	// per-lane LDS offset of k-step s (row r31 of column tile 0); column tile tc adds tc*32 rows,
	// whose swizzle term is the same
	static_assert((32 / RPB) % (SWM + 1) == 0, "swizzle must repeat every 32 rows");
	uint32_t aoff[KS];
#pragma unroll
	for (int s = 0; s < KS; ++s)
		aoff[s] = r31 * ROWB + (((h * KS + s) ^ ((r31 / RPB) & SWM)) * 16);

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	auto tile_row0 = [&](uint32_t t) -> uint32_t {
		return ((t / p.tile_run) * (p.tile_stride * p.tile_run) + (t % p.tile_run)) * kGemmTB;
	};

	constexpr int ROWS_PER_INSTR = kGemmThreads / CH;
	static_assert(ROWS_PER_INSTR % (16 * RPB) == 0, "swizzle must be instruction-invariant");
	constexpr int N_STAGE = kGemmTB * CH / kGemmThreads;
	const uint32_t lane_row = tid / CH;
	const uint32_t lane_off = lane_row * ROWB + (((tid % CH) ^ ((lane_row / RPB) & SWM)) * 16);
	auto stage = [&](uint32_t t, int buf) {
		const uint32_t row0 = tile_row0(t);
		unsigned char* dst0 = smem + buf * TILE_BYTES + wave * 64 * 16;
		if (row0 + kGemmTB <= p.n_rows) {
			const unsigned char* tb = (const unsigned char*)p.base_f16 + (size_t)row0 * ROWB;
#pragma unroll
			for (int i = 0; i < N_STAGE; ++i)
				__builtin_amdgcn_global_load_lds(
				    (const __attribute__((address_space(1))) void*)(tb + lane_off +
				                                                    (uint32_t)i * ROWS_PER_INSTR * ROWB),
				    (__attribute__((address_space(3))) void*)(dst0 + i * kGemmThreads * 16), 16, 0, 0);
		} else {
#pragma unroll
			for (int i = 0; i < N_STAGE; ++i) {
				uint32_t grow = row0 + i * ROWS_PER_INSTR + lane_row;
				if (grow >= p.n_rows)
					grow = p.n_rows - 1;
				const unsigned char* src = (const unsigned char*)p.base_f16 + (size_t)grow * ROWB +
				                           (lane_off - lane_row * ROWB);
				__builtin_amdgcn_global_load_lds(
				    (const __attribute__((address_space(1))) void*)src,
				    (__attribute__((address_space(3))) void*)(dst0 + i * kGemmThreads * 16), 16, 0, 0);
			}
		}
	};

	auto load_bn = [&](float (&bnv)[4], uint32_t row0) {
#pragma unroll
		for (int tc = 0; tc < 4; ++tc) {
			const uint32_t brow = row0 + tc * 32 + r31;
			bnv[tc] = brow < p.n_rows ? p.bnorm[brow] : __builtin_inff();
		}
	};
	const float c2 = p.neg2_inv_s2;
	auto epilogue = [&](const f32x16 (&accs)[4], uint32_t row0, const float (&bnv)[4]) {
#pragma unroll
		for (int tc = 0; tc < 4; ++tc) {
			const uint32_t brow = row0 + tc * 32 + r31;
			const float bn = bnv[tc];
#pragma unroll
			for (int r4 = 0; r4 < 16; r4 += 4) {
				float tv[4];
				bool any = false;
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					tv[e] = __builtin_fmaf(c2, accs[tc][r4 + e], bn);
					any |= tv[e] <= th[r4 + e];
				}
				if (__builtin_amdgcn_ballot_w64(any) != 0) {
					uint32_t qrow0 = q0 + wr * 32 + 4 * h;  // rare path: arithmetic stays in here
					asm volatile("" : "+v"(qrow0));
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const int reg = r4 + e;
						if (tv[e] <= th[reg]) {
							const uint32_t qi = qrow0 + (reg & 3) + 8 * (reg >> 2);
							const uint32_t slot = atomicAdd(&p.cand_cnt[qi], 1u);
							if (slot < p.cap)
								p.cand[(size_t)qi * p.cap + slot] = make_key(tv[e], brow);
						}
					}
				}
			}
		}
	};

	if (t0 < t1)
		stage(t0, 0);
	__syncthreads();

	// (wave pairing and the ordering against the LDS-DMA queue: see scan_gemm_bf16.hpp)
	const bool deferred = wave >= 4;
	f32x16 acc[4];
	uint32_t prev_row0 = 0;
	bool have_prev = false;
	float bnv[4];
	int buf = 0;
	for (uint32_t t = t0; t < t1; ++t, buf ^= 1) {
		if (deferred) {
			if (have_prev) {
				load_bn(bnv, prev_row0);
				epilogue(acc, prev_row0, bnv);
			}
		} else {
			load_bn(bnv, tile_row0(t));
		}
		if (t + 1 < t1)
			stage(t + 1, buf ^ 1);
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
#pragma unroll
			for (int e = 0; e < 16; ++e)
				acc[tc][e] = 0.0f;
		const uint32_t boff = (uint32_t)buf * TILE_BYTES;
		auto frag = [&](int tc, int s) -> f16x8 {
			return *reinterpret_cast<const f16x8*>(smem + (boff + aoff[s]) + tc * 32 * ROWB);
		};
#pragma unroll
		for (int s = 0; s < KS; ++s) {
#pragma unroll
			for (int tc = 0; tc < 4; ++tc)
				acc[tc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], frag(tc, s), acc[tc], 0, 0, 0);
		}
		const uint32_t row0 = tile_row0(t);
		if (!deferred) {
			epilogue(acc, row0, bnv);
		} else {
			prev_row0 = row0;
			have_prev = true;
		}
		__syncthreads();
	}
	if (deferred && have_prev) {
		load_bn(bnv, prev_row0);
		epilogue(acc, prev_row0, bnv);
	}
}

}  // namespace expann
