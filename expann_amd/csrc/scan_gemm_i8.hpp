// scan_gemm_i8.hpp -- large-batch threshold filter over 8-bit rows in GEMM form on the int8
// matrix cores (v_mfma_i32_32x32x32_i8), BASELINE config C5 (int8 inner product) and the int8 /
// uint8 L2 forms.
//
// Integer arithmetic is exact, so unlike the fp32 GEMM form (round 1-2's fp32-input MFMA form) no slack and
// no re-rank are needed: the expanded identity  sum(a-b)^2 = sum a^2 + sum b^2 - 2 sum ab  holds
// exactly in wrapping 32-bit integers, and the candidates leave this kernel with their final
// scores.  uint8 rows (src/antitopo_engine.h:38-61) are mapped to int8 by subtracting 128 from
// both sides (x ^ 0x80), which leaves every difference -- hence the score -- unchanged.
//
// Test in the epilogue (all int32):   bias[row] + MULT*acc  <=  theta_q
//   L2 forms: bias = sum b^2, MULT = -2, theta_q = floor(tau_q) - sum q^2
//   IP:       bias = 0,       MULT = -1, theta_q = floor(tau_q)
// (scores are integers, so score <= tau  <=>  score <= floor(tau)).
//
// Geometry: 256 queries x (32*NT) rows per workgroup step (NT = 2/4/8 for d = 768/256/128: 32-48
// KiB tiles); 8 waves (2 per SIMD), each 32 queries against every row of the tile (NT MFMA tiles,
// NT*d/32 MFMAs per step); the wave's query fragments (d/2 bytes per lane) stay in VGPRs; the base
// tile is staged HBM/L2 -> LDS with global_load_lds_dwordx4 into two buffers, XOR-swizzled on
// the source address exactly like the fp32 kernel.
#pragma once
#include "common.hpp"
#include "gemm_terms.hpp"
#include "scan_int8.hpp"

namespace expann {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmI8TQ = 256;  // 8 waves x 32 queries
// rows per step = NT MFMA column tiles x 32; NT keeps the LDS tile at 32-48 KiB
__host__ __device__ constexpr int gemm_i8_nt(int d) { return d >= 512 ? 2 : (d >= 256 ? 4 : 8); }
__host__ __device__ constexpr int gemm_i8_tb(int d) { return 32 * gemm_i8_nt(d); }

struct GemmI8Params {
	const void* base;        // [n_rows][D] bytes
	const int* bias;         // [n_rows] sum b^2 (L2 forms; shifted for uint8) -- unused for IP
	uint32_t n_rows;
	uint32_t n_tiles_sel;    // gemm_i8_tb(D)-row tiles this launch visits ...
	uint32_t tile_stride;
	uint32_t tiles_per_block;
	uint32_t n_qtiles;
	const void* queries;     // [m][D] bytes (uint8 rows: queries already truncated to uint8)
	const int* theta;        // [m]
	const int* qself;        // [m] sum q^2 (L2 forms) or nullptr
	uint32_t m;
	uint32_t* cand_cnt;
	uint64_t* cand;
	uint32_t cap;
};

// sum b^2 per row (uint8 rows: of b-128); 16 lanes per row
template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void row_self_i8_kernel(const void* base, uint32_t n_rows,
                                                             int* out) {
	constexpr int NW = D / 64;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t row = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const uint32_t rr = row < n_rows ? row : n_rows - 1;
	const int* src = (const int*)base + (size_t)rr * (D / 4) + l * NW;
	int acc = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w) {
		int v = src[w];
		if (MODE == kU8L2)
			v ^= (int)0x80808080;
		acc = __builtin_amdgcn_sdot4(v, v, acc, false);
	}
	acc = reduce16_i32(acc);
	if (row < n_rows && l == 0)
		out[row] = acc;
}

// theta_q (and sum q^2) per query; 16 lanes per query
template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void query_theta_i8_kernel(const void* queries, uint32_t m,
                                                                const float* tau, int* theta,
                                                                int* qself) {
	constexpr int NW = D / 64;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t qi = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const uint32_t qq = qi < m ? qi : m - 1;
	const int* src = (const int*)queries + (size_t)qq * (D / 4) + l * NW;
	int acc = 0;
	if (MODE != kI8IP) {
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			int v = src[w];
			if (MODE == kU8L2)
				v ^= (int)0x80808080;
			acc = __builtin_amdgcn_sdot4(v, v, acc, false);
		}
		acc = reduce16_i32(acc);
	}
	if (qi < m && l == 0 && !tau)
		qself[qi] = acc;  // (only the query term is wanted: scan_gemm_i8q.hpp)
	if (qi < m && l == 0 && tau) {
		const float t = tau[qi];
		// floor(tau) clamped into int32 (scores are integers of magnitude < 2^31)
		int ft;
		if (!(t < 2147483520.0f))
			ft = 2147483647;
		else if (t < -2147483520.0f)
			ft = -2147483647 - 1;
		else
			ft = (int)__builtin_floorf(t);
		// theta = floor(tau) - sum q^2 without wrapping
		long long th = (long long)ft - (long long)acc;
		if (th > 2147483647LL)
			th = 2147483647LL;
		if (th < -2147483648LL)
			th = -2147483648LL;
		theta[qi] = (int)th;
		qself[qi] = acc;
	}
}

template <int D, int MODE>
__global__ __launch_bounds__(kGemmThreads, 2) void scan_gemm_i8_kernel(GemmI8Params p) {
	static_assert(D == 128 || D == 256 || D == 768, "built for d = 128, 256, 768");
	static_assert(MODE == kU8L2 || MODE == kI8L2 || MODE == kI8IP, "bilinear forms only");
	constexpr int CH = D / 16;  // 16-byte chunks per row
	constexpr int KH = CH / 2;  // chunks (= MFMA k-steps of 32) per lane half
	constexpr int NT = gemm_i8_nt(D);  // MFMA column tiles per wave = all rows of the tile / 32
	constexpr int TB = gemm_i8_tb(D);
	constexpr int TILE_BYTES = TB * D;
	constexpr bool L2FORM = (MODE != kI8IP);
	// swizzle: rows that share a 256-byte LDS bank row get the same XOR, 16 consecutive bank
	// rows get 16 (or CH, if smaller) different ones -> conflict-free ds_read_b128 fragments
	constexpr int RPB = (D < 256) ? 256 / D : 1;
	constexpr int SWM = (CH < 16 ? CH : 16) - 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// the kernel is bound by the LDS fill rate, so the query tile is as large as the register
	// file allows: 8 waves x 32 queries, each wave against every row of the tile
	const int wr = wave;
	const int h = lane >> 5, r31 = lane & 31;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * kGemmI8TQ;

	// query fragments: row r31 of the wave's 32 queries, bytes [h*D/2, (h+1)*D/2)
	i32x4 a[KH];
	{
		uint32_t qi = q0 + wr * 32 + r31;
		if (qi >= p.m)
			qi = p.m - 1;
		const i32x4* src = reinterpret_cast<const i32x4*>((const unsigned char*)p.queries +
		                                                  (size_t)qi * D + h * (D / 2));
#pragma unroll
		for (int g = 0; g < KH; ++g) {
			i32x4 v = src[g];
			if (MODE == kU8L2)
				v ^= (int)0x80808080;
			a[g] = v;
		}
	}
	int th[16];
	bool qvalid[16];  // a padded query slot must never match
#pragma unroll
	for (int reg = 0; reg < 16; ++reg) {
		const uint32_t qi = q0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
		qvalid[reg] = qi < p.m;
		th[reg] = qi < p.m ? p.theta[qi] : (-2147483647 - 1);
	}
	// per-lane LDS offsets of the fragment chunks (k-step g): row r31, chunk (h*KH+g) ^ swizzle;
	// column tile tc adds tc*32 rows, whose swizzle term is the same (32/RPB is a multiple of 16)
	static_assert((32 / RPB) % (SWM + 1) == 0, "swizzle must repeat every 32 rows");
	uint32_t aoff[KH];
#pragma unroll
	for (int g = 0; g < KH; ++g)
		aoff[g] = r31 * D + (((h * KH + g) ^ ((r31 / RPB) & SWM)) * 16);

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;

	constexpr int N_STAGE = TB * CH / kGemmThreads;  // staging instructions per thread
	static_assert(TB * CH % kGemmThreads == 0, "tile must be a whole number of staging rounds");
	constexpr int SLOTS_PER_ROUND = kGemmThreads;    // 16-byte slots per round
	auto stage = [&](uint32_t t, int buf) {
		const uint32_t row0 = t * p.tile_stride * TB;
		unsigned char* dst0 = smem + buf * TILE_BYTES + wave * 64 * 16;
#pragma unroll
		for (int i = 0; i < N_STAGE; ++i) {
			const int S = i * SLOTS_PER_ROUND + tid;
			const int r = S / CH, pc = S % CH;
			const int c = pc ^ ((r / RPB) & SWM);
			uint32_t grow = row0 + r;
			if (grow >= p.n_rows)
				grow = p.n_rows - 1;
			const unsigned char* src = (const unsigned char*)p.base + (size_t)grow * D + c * 16;
			__builtin_amdgcn_global_load_lds(
			    (const __attribute__((address_space(1))) void*)src,
			    (__attribute__((address_space(3))) void*)(dst0 + i * SLOTS_PER_ROUND * 16), 16, 0, 0);
		}
	};
	auto load_bias = [&](int (&bv)[NT], uint32_t row0) {
#pragma unroll
		for (int tc = 0; tc < NT; ++tc) {
			const uint32_t brow = row0 + tc * 32 + r31;
			bv[tc] = (L2FORM && brow < p.n_rows) ? p.bias[brow] : 0;
		}
	};
	auto epilogue = [&](const i32x16 (&accs)[NT], uint32_t row0, const int (&bv)[NT]) {
#pragma unroll
		for (int tc = 0; tc < NT; ++tc) {
			const uint32_t brow = row0 + tc * 32 + r31;
			const bool bvalid = brow < p.n_rows;
#pragma unroll
			for (int r4 = 0; r4 < 16; r4 += 4) {
				int tv[4];
				bool any = false;
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					tv[e] = L2FORM ? bv[tc] - 2 * accs[tc][r4 + e] : -accs[tc][r4 + e];
					any |= bvalid && qvalid[r4 + e] && tv[e] <= th[r4 + e];
				}
				if (__builtin_amdgcn_ballot_w64(any) != 0) {
					// rare path: keep its address arithmetic inside the branch
					uint32_t qrow0 = q0 + wr * 32 + 4 * h;
					asm volatile("" : "+v"(qrow0));
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const int reg = r4 + e;
						if (bvalid && qvalid[reg] && tv[e] <= th[reg]) {
							const uint32_t qi = qrow0 + (reg & 3) + 8 * (reg >> 2);
							const int qs = L2FORM ? p.qself[qi] : 0;
							const float score = (float)(tv[e] + qs);
							const uint32_t slot = atomicAdd(&p.cand_cnt[qi], 1u);
							if (slot < p.cap)
								p.cand[(size_t)qi * p.cap + slot] = make_key(score, brow);
						}
					}
				}
			}
		}
	};

	if (t0 < t1)
		stage(t0, 0);
	__syncthreads();

	// SIMD partners (w, w+4) alternate MFMA and epilogue phases; vector-memory operations retire
	// in issue order, so the deferred waves run their epilogue before they issue their share of
	// the next tile's LDS-DMA and the others fetch the row terms before staging
	const bool deferred = wave >= 4;
	i32x16 acc[NT];
	int bv[NT];
	uint32_t prev_row0 = 0;
	bool have_prev = false;

	int buf = 0;
	for (uint32_t t = t0; t < t1; ++t, buf ^= 1) {
		const uint32_t row0 = t * p.tile_stride * TB;
		if (deferred) {
			if (have_prev) {
				load_bias(bv, prev_row0);
				epilogue(acc, prev_row0, bv);
			}
		} else {
			load_bias(bv, row0);
		}
		if (t + 1 < t1)
			stage(t + 1, buf ^ 1);
#pragma unroll
		for (int tc = 0; tc < NT; ++tc)
#pragma unroll
			for (int e = 0; e < 16; ++e)
				acc[tc][e] = 0;

		const uint32_t boff = (uint32_t)buf * TILE_BYTES;
		auto frag = [&](int tc, int g) -> i32x4 {
			i32x4 v = *reinterpret_cast<const i32x4*>(smem + (boff + aoff[g]) + tc * 32 * D);
			if (MODE == kU8L2)
				v ^= (int)0x80808080;
			return v;
		};
#pragma unroll
		for (int g = 0; g < KH; ++g) {
#pragma unroll
			for (int tc = 0; tc < NT; ++tc)
				acc[tc] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[g], frag(tc, g), acc[tc], 0, 0, 0);
		}

		if (!deferred) {
			epilogue(acc, row0, bv);
		} else {
			prev_row0 = row0;
			have_prev = true;
		}
		__syncthreads();
	}
	if (deferred && have_prev) {
		load_bias(bv, prev_row0);
		epilogue(acc, prev_row0, bv);
	}
}

}  // namespace expann
