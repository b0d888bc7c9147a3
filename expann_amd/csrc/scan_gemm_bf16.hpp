// scan_gemm_bf16.hpp -- the GEMM-form fp32 L2 candidate filter (scan_gemm_f32.hpp) on the bf16
// matrix cores with a 3-term split: every fp32 value x is stored as two bf16 numbers
//      hi = bf16(x),   lo = bf16(x - hi)            (|x - hi - lo| <= 2^-18 |x|)
// and the inner product is evaluated as  q.b ~= qh.bh + qh.bl + ql.bh  with
// v_mfma_f32_32x32x16_bf16 (bf16 products are exact in fp32, accumulation is fp32).  Three
// bf16 MFMAs cost 3/16 of the fp32-input MFMAs they replace.
//
// This is NOT a reduced-precision result: the filter only has to keep every row whose
// reference-order score is <= tau_q, and it does so with a slack that covers the dropped terms
// (ql.bl and the split residuals, <= 3*2^-18 |q_i b_i| per element), the fp32 accumulation of
// the 3d products and everything the fp32 GEMM form already budgets:
//      eps = (10 d + 512) * 2^-24      (needed: ~(4d + 226) * 2^-24; the rest is margin for the
//                                       matrix core's internal summation order)
// Survivors are re-scored in the reference's exact 16-lane FMA order by the select kernel, so
// the final ids and distances are bit-identical to the direct scan.
//
// Layout: base rows and queries are pre-split into [row][hi: d bf16][lo: d bf16] (same 4d bytes
// per row as fp32).  Workgroup step = 128 queries x 128 rows; 8 waves as 4 (query quarters of
// 32) x 2 (row halves of 64): a wave owns 2 MFMA tiles sharing one A fragment pair; the A
// fragments (d/16 k-steps x {hi,lo} x 4 VGPRs) stay in registers.  A lane's k-slice of a part
// is dims [d/2*h, d/2*(h+1)) (h = lane>>5), 8 dims per MFMA.  LDS staging and the source-side XOR
// swizzle are those of scan_gemm_f32.hpp.
#pragma once
#include "common.hpp"
#include "scan_gemm_f32.hpp"

namespace expann {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline float gemm_bf16_filter_eps(int d) {
	return (float)(10 * d + 512) * 5.9604644775390625e-08f;
}

// fp32 [n][D] -> [n][2][D] bf16 (hi plane, lo plane)
__global__ __launch_bounds__(kBlock) void split_bf16_kernel(const float* in, size_t n_rows, int d,
                                                            __bf16* out) {
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_rows * (size_t)d)
		return;
	const size_t row = i / d, col = i % d;
	const float x = in[i];
	const __bf16 hi = (__bf16)x;
	const __bf16 lo = (__bf16)(x - (float)hi);
	out[row * 2 * d + col] = hi;
	out[row * 2 * d + d + col] = lo;
}

struct GemmBf16Params {
	const void* base_split;  // [n_rows][2][D] bf16
	const float* bnorm;      // [n_rows] ||b||^2 * (1 - eps)  (exact fp32 norms, scaled)
	uint32_t n_rows;
	uint32_t n_tiles_sel;
	uint32_t tile_stride;
	uint32_t tile_run;       // sampled tiles come in runs of this many consecutive tiles
	uint32_t tiles_per_block;
	uint32_t n_qtiles;
	const void* queries_split;  // [m][2][D] bf16
	const float* theta;         // [m]
	uint32_t m;
	uint32_t* cand_cnt;
	uint64_t* cand;
	uint32_t cap;
};

template <int D>
__global__ __launch_bounds__(kGemmThreads, 2) void scan_gemm_bf16_kernel(GemmBf16Params p) {
	static_assert(D == 128 || D == 64, "built for d = 64, 128");
	constexpr int ROWB = D * 4;      // bytes per split row
	constexpr int CH = ROWB / 16;    // 16-byte chunks per row (hi: [0,CH/2), lo: [CH/2,CH))
	constexpr int KS = D / 16;       // MFMA k-steps per part
	constexpr int TILE_BYTES = kGemmTB * ROWB;
	constexpr int RPB = (ROWB < 256) ? 256 / ROWB : 1;
	constexpr int SWM = (CH < 16 ? CH : 16) - 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = wave >> 1, wc = wave & 1;  // 4 query quarters of 32 x 2 row halves of 64
	const int h = lane >> 5, r31 = lane & 31;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * kGemmTQ;

	// query fragments: {hi, lo} x KS k-steps.  k-step s of lane half h covers dims
	// [8*(h*KS + s), +8) of a part, i.e. 16-byte chunk h*KS + s of the part's 2*D bytes.
	bf16x8 a_hi[KS], a_lo[KS];
	{
		uint32_t qi = q0 + wr * 32 + r31;
		if (qi >= p.m)
			qi = p.m - 1;
		const bf16x8* src = reinterpret_cast<const bf16x8*>((const unsigned char*)p.queries_split +
		                                                    (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KS; ++s) {
			a_hi[s] = src[h * KS + s];
			a_lo[s] = src[CH / 2 + h * KS + s];
		}
	}
	float th[16];
#pragma unroll
	for (int reg = 0; reg < 16; ++reg) {
		const uint32_t qi = q0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
		th[reg] = qi < p.m ? p.theta[qi] : -__builtin_inff();
	}

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	auto tile_row0 = [&](uint32_t t) -> uint32_t {
		return ((t / p.tile_run) * (p.tile_stride * p.tile_run) + (t % p.tile_run)) * kGemmTB;
	};

	auto stage = [&](uint32_t t, int buf) {
		const uint32_t row0 = tile_row0(t);
#pragma unroll
		for (int i = 0; i < kGemmTB * CH / kGemmThreads; ++i) {
			const int S = i * kGemmThreads + tid;
			const int r = S / CH, pc = S % CH;
			const int c = pc ^ ((r / RPB) & SWM);
			uint32_t grow = row0 + r;
			if (grow >= p.n_rows)
				grow = p.n_rows - 1;
			const unsigned char* src = (const unsigned char*)p.base_split + (size_t)grow * ROWB + c * 16;
			unsigned char* dst = smem + buf * TILE_BYTES + (i * kGemmThreads + wave * 64) * 16;
			__builtin_amdgcn_global_load_lds(
			    (const __attribute__((address_space(1))) void*)src,
			    (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
		}
	};

	auto epilogue = [&](const f32x16& acc0, const f32x16& acc1, uint32_t row0) {
#pragma unroll
		for (int tc = 0; tc < 2; ++tc) {
			const uint32_t brow = row0 + wc * 64 + tc * 32 + r31;
			const float bn = brow < p.n_rows ? p.bnorm[brow] : __builtin_inff();
			const f32x16& acc = tc ? acc1 : acc0;
#pragma unroll
			for (int r4 = 0; r4 < 16; r4 += 4) {
				float tv[4];
				bool any = false;
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					tv[e] = __builtin_fmaf(-2.0f, acc[r4 + e], bn);
					any |= tv[e] <= th[r4 + e];
				}
				if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const int reg = r4 + e;
						if (tv[e] <= th[reg]) {
							const uint32_t qi = q0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
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

	const bool deferred = wave >= 4;  // SIMD partners (w, w+4) alternate MFMA and epilogue phases
	f32x16 acc0, acc1;
	uint32_t prev_row0 = 0;
	bool have_prev = false;

	int buf = 0;
	for (uint32_t t = t0; t < t1; ++t, buf ^= 1) {
		if (t + 1 < t1)
			stage(t + 1, buf ^ 1);
		if (deferred && have_prev)
			epilogue(acc0, acc1, prev_row0);
#pragma unroll
		for (int e = 0; e < 16; ++e) {
			acc0[e] = 0.0f;
			acc1[e] = 0.0f;
		}
		const unsigned char* bt = smem + buf * TILE_BYTES;
		const int rb0 = wc * 64 + r31, rb1 = rb0 + 32;
		auto frag = [&](int rb, int part, int s) -> bf16x8 {
			const int c = (part * (CH / 2) + h * KS + s) ^ ((rb / RPB) & SWM);
			return *reinterpret_cast<const bf16x8*>(bt + rb * ROWB + c * 16);
		};
		bf16x8 bh0 = frag(rb0, 0, 0), bl0 = frag(rb0, 1, 0), bh1 = frag(rb1, 0, 0), bl1 = frag(rb1, 1, 0);
#pragma unroll
		for (int s = 0; s < KS; ++s) {
			bf16x8 nbh0 = bh0, nbl0 = bl0, nbh1 = bh1, nbl1 = bl1;
			if (s + 1 < KS) {
				nbh0 = frag(rb0, 0, s + 1);
				nbl0 = frag(rb0, 1, s + 1);
				nbh1 = frag(rb1, 0, s + 1);
				nbl1 = frag(rb1, 1, s + 1);
			}
			// small cross terms first, the dominant hi.hi product last
			acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[s], bh0, acc0, 0, 0, 0);
			acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[s], bh1, acc1, 0, 0, 0);
			acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bl0, acc0, 0, 0, 0);
			acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bl1, acc1, 0, 0, 0);
			acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bh0, acc0, 0, 0, 0);
			acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bh1, acc1, 0, 0, 0);
			bh0 = nbh0;
			bl0 = nbl0;
			bh1 = nbh1;
			bl1 = nbl1;
		}
		const uint32_t row0 = tile_row0(t);
		if (!deferred) {
			epilogue(acc0, acc1, row0);
		} else {
			prev_row0 = row0;
			have_prev = true;
		}
		__syncthreads();
	}
	if (deferred && have_prev)
		epilogue(acc0, acc1, prev_row0);
}

}  // namespace expann
