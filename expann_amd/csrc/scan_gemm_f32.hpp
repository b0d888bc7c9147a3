// scan_gemm_f32.hpp -- large-batch fp32 L2 candidate filter in GEMM form on the matrix cores.
//
// For a tile of 128 queries x 128 base rows the workgroup computes the dense Q x B^T inner
// products with v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: an exact k-ordered fmaf
// chain, MI355X_MICROARCH.md "FP32-input MFMA") and tests
//
//        ||b||^2 (1-eps)  -  2 q.b   <=   tau_q  -  ||q||^2 (1-eps)          (*)
//
// in the epilogue.  (*) is the reference's squared-L2 test  dist2(q,b) <= tau_q  written in
// expanded form with a slack eps = (4d+128) * 2^-24 that dominates every rounding error of the
// expanded evaluation AND of the reference-order evaluation, so every row whose
// reference-order score is <= tau_q survives.  Survivors are only CANDIDATES: the select kernel
// re-scores them in the reference's exact 16-lane FMA order (select.hpp, rerank) before the
// (score, id) selection, so the final ids and distances are bit-identical to the direct scan.
//
// Data movement per workgroup step: one 128-row base tile (64 KiB at d=128) goes HBM/L2 -> LDS
// with global_load_lds_dwordx4 (two LDS buffers; the next tile streams in while the MFMAs of
// the current one run), XOR-swizzled through the SOURCE address so that the 16-byte fragment
// reads (ds_read_b128, rows 512 B apart) are bank-conflict free.  The query tile's fragments
// (d/2 VGPRs per lane) are loaded once per workgroup.  8 waves = 4 x 2 sub-tiles of 32 queries
// x 64 rows (1 x 2 MFMA tiles of 32x32), two waves per SIMD; a lane's k-slice is dims [0,d/2)
// for lanes 0-31 and [d/2,d) for lanes 32-63 (any k order is a valid dot product; the slack
// covers the rounding).
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

constexpr int kGemmThreads = 512;  // 8 waves: 4 query sub-tiles x 2 row halves, 2 waves per SIMD

template <int D>
__global__ __launch_bounds__(kGemmThreads, 2) void scan_gemm_f32_kernel(GemmScanParams p) {
	static_assert(D % 64 == 0 && D <= 128, "built for d = 64, 128");
	constexpr int CH = D / 4;   // 16-byte chunks per row
	constexpr int KH = D / 2;   // dims per lane half
	constexpr int TILE_BYTES = kGemmTB * D * 4;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// wave tile = 32 queries x 64 base rows (1 x 2 MFMA tiles of 32x32).  Two waves share a
	// SIMD, so one wave's LDS waits / epilogue VALU overlap the other's MFMAs.
	const int wr = wave >> 1, wc = wave & 1;
	const int h = lane >> 5, r31 = lane & 31;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * kGemmTQ;

	// ---- query fragments: A[i = r31][k = h*KH + s] --------------------------------------
	float a[KH];
	{
		uint32_t qi = q0 + wr * 32 + r31;
		if (qi >= p.m)
			qi = p.m - 1;
		const f32x4* src = reinterpret_cast<const f32x4*>(p.queries + (size_t)qi * D + h * KH);
#pragma unroll
		for (int g = 0; g < KH / 4; ++g) {
			const f32x4 v = src[g];
			a[4 * g + 0] = v[0];
			a[4 * g + 1] = v[1];
			a[4 * g + 2] = v[2];
			a[4 * g + 3] = v[3];
		}
	}
	// thresholds of the 16 query rows this lane's accumulators belong to
	// (C/D layout of the 32x32 MFMA: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31)
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

	// ---- stage one base tile into LDS buffer `buf` (LDS-DMA, 16 B per lane) -------------
	// LDS image is lane-linear: slot S = r*CH + pc holds chunk c = pc ^ (r & 15) of tile row r.
	auto stage = [&](uint32_t t, int buf) {
		const uint32_t row0 = t * p.tile_stride * kGemmTB;
#pragma unroll
		for (int i = 0; i < kGemmTB * CH / kGemmThreads; ++i) {
			const int S = i * kGemmThreads + tid;
			const int r = S / CH, pc = S % CH;
			const int c = pc ^ (r & 15);
			uint32_t grow = row0 + r;
			if (grow >= p.n_rows)
				grow = p.n_rows - 1;
			const float* src = p.base + (size_t)grow * D + c * 4;
			unsigned char* dst = smem + buf * TILE_BYTES + (i * kGemmThreads + wave * 64) * 16;
			__builtin_amdgcn_global_load_lds(
			    (const __attribute__((address_space(1))) void*)src,
			    (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
		}
	};

	if (t0 < t1)
		stage(t0, 0);
	__syncthreads();  // (drains the LDS-DMA: hipcc emits vmcnt(0) before the barrier)

	// ---- epilogue: threshold test on one 32x64 accumulator pair, rare candidate append ----
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

	// The two waves of a SIMD (w and w+4) would otherwise run their MFMA phases and their
	// VALU epilogues in lockstep and leave the matrix pipe idle during the epilogues; waves
	// 4-7 defer the epilogue of tile t to the start of step t+1, under the partner's MFMAs.
	const bool deferred = wave >= 4;
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
		const int sw = r31 & 15;
		auto frag = [&](int rb, int g) -> f32x4 {
			const int c = (h * (CH / 2) + g) ^ sw;
			return *reinterpret_cast<const f32x4*>(bt + rb * (D * 4) + c * 16);
		};
		// fragment registers are double-buffered: group g+1 is read while group g multiplies
		f32x4 b0 = frag(rb0, 0), b1 = frag(rb1, 0);
#pragma unroll
		for (int g = 0; g < KH / 4; ++g) {
			f32x4 nb0 = b0, nb1 = b1;
			if (g + 1 < KH / 4) {
				nb0 = frag(rb0, g + 1);
				nb1 = frag(rb1, g + 1);
			}
#pragma unroll
			for (int kk = 0; kk < 4; ++kk) {
				const int s = 4 * g + kk;
				acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[kk], acc0, 0, 0, 0);
				acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[kk], acc1, 0, 0, 0);
			}
			b0 = nb0;
			b1 = nb1;
		}

		const uint32_t row0 = t * p.tile_stride * kGemmTB;
		if (!deferred) {
			epilogue(acc0, acc1, row0);
		} else {
			prev_row0 = row0;
			have_prev = true;
		}
		__syncthreads();  // next tile landed (vmcnt(0)) and everyone is done reading `buf`
	}
	if (deferred && have_prev)
		epilogue(acc0, acc1, prev_row0);
}

}  // namespace expann
