// scan_gemm_bf16.hpp -- the GEMM-form fp32 L2 candidate filter (round 1-2's fp32-input MFMA form) on the bf16
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
// per row as fp32).  Workgroup step = 256 queries x 128 rows -- the kernel is bound by the LDS
// fill rate of a CU (~25 GB/s per CU by LDS-DMA), so the query tile is as large as the register
// file allows: 8 waves x 32 queries, each wave against all 128 rows of the tile (4 MFMA tiles
// sharing one A fragment pair); the A fragments (d/16 k-steps x {hi,lo} x 4 VGPRs) stay in
// registers.  A lane's k-slice of a part
// is dims [d/2*h, d/2*(h+1)) (h = lane>>5), 8 dims per MFMA.  LDS staging and the source-side XOR
// swizzle are those of rounds 1-2's fp32-input form.
#pragma once
#include "common.hpp"
#include "gemm_terms.hpp"

namespace expann {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kGemmBf16TQ = 256;  // queries per workgroup: 8 waves x 32

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
	uint32_t debug;  // timing experiments only: 1 = no epilogue, 2 = no staging, 4 = no MFMA
};

template <int D>
__global__ __launch_bounds__(kGemmThreads, 2) void scan_gemm_bf16x3_kernel(GemmBf16Params p) {
	static_assert(D == 128 || D == 64, "built for d = 64, 128");
	constexpr int ROWB = D * 4;      // bytes per split row
	constexpr int CH = ROWB / 16;    // 16-byte chunks per row (hi: [0,CH/2), lo: [CH/2,CH))
	constexpr int KS = D / 16;       // MFMA k-steps per part
	constexpr int TILE_BYTES = kGemmTB * ROWB;
	constexpr int RPB = (ROWB < 256) ? 256 / ROWB : 1;
	// XOR mask: stays inside one part (hi or lo) of the row; 16 values at d=128 (conflict-free
	// ds_read_b128), 8 at d=64 (2-way)
	constexpr int SWM = (CH / 2 < 16 ? CH / 2 : 16) - 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = wave;  // each wave: 32 queries x all 128 rows of the tile (4 MFMA tiles)
	const int h = lane >> 5, r31 = lane & 31;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * kGemmBf16TQ;

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

	// LDS byte offset (within a tile) of this lane's 16-byte chunk of k-step s, column tile 0,
	// hi part: row r31, chunk (h*KS + s) ^ swizzle(r31).  Column tile tc adds tc*32 rows (the
	// swizzle only looks at row bits below 32*RPB... i.e. is unchanged), the lo part adds CH/2
	// chunks (the XOR never touches that bit: SWM < CH/2).
	static_assert(SWM < CH / 2 && (32 / RPB) % (SWM + 1) == 0, "swizzle layout");
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

	// Per-lane byte offset of this thread's 16-byte chunk inside a tile (the same for every
	// staging instruction: an instruction covers kGemmThreads/CH whole rows, a multiple of 16, so
	// the swizzle term does not depend on the instruction index).
	constexpr int ROWS_PER_INSTR = kGemmThreads / CH;
	static_assert(ROWS_PER_INSTR % (16 * RPB) == 0, "swizzle must be instruction-invariant");
	const uint32_t lane_row = tid / CH;
	const uint32_t lane_off = lane_row * ROWB + (((tid % CH) ^ ((lane_row / RPB) & SWM)) * 16);
	auto stage = [&](uint32_t t, int buf) {
		const uint32_t row0 = tile_row0(t);
		unsigned char* dst0 = smem + buf * TILE_BYTES + wave * 64 * 16;
		if (row0 + kGemmTB <= p.n_rows) {  // whole tile in range: uniform base + lane offset
			const unsigned char* tb = (const unsigned char*)p.base_split + (size_t)row0 * ROWB;
#pragma unroll
			for (int i = 0; i < kGemmTB * CH / kGemmThreads; ++i)
				__builtin_amdgcn_global_load_lds(
				    (const __attribute__((address_space(1))) void*)(tb + lane_off +
				                                                    (uint32_t)i * ROWS_PER_INSTR * ROWB),
				    (__attribute__((address_space(3))) void*)(dst0 + i * kGemmThreads * 16), 16, 0, 0);
		} else {  // ragged last tile: clamp the row
#pragma unroll
			for (int i = 0; i < kGemmTB * CH / kGemmThreads; ++i) {
				uint32_t grow = row0 + i * ROWS_PER_INSTR + lane_row;
				if (grow >= p.n_rows)
					grow = p.n_rows - 1;
				const unsigned char* src = (const unsigned char*)p.base_split + (size_t)grow * ROWB +
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
					tv[e] = __builtin_fmaf(-2.0f, accs[tc][r4 + e], bn);
					any |= tv[e] <= th[r4 + e];
				}
				if (__builtin_amdgcn_ballot_w64(any) != 0) {
					// rare path: keep its address arithmetic inside the branch (hoisted out of the
					// tile loop it would pin ~64 VGPRs of per-query pointers)
					uint32_t qrow0 = q0 + wr * 32 + 4 * h;
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

	// SIMD partners (w, w+4) alternate MFMA and epilogue phases: waves 4-7 test the accumulators
	// of tile t at the start of step t+1, under their partner's MFMAs
	const bool deferred = wave >= 4;
	f32x16 acc[4];
	uint32_t prev_row0 = 0;
	bool have_prev = false;

	// Ordering against the LDS-DMA queue: vector-memory operations retire in issue order, so a
	// load or returning atomic issued AFTER the 16 staging instructions cannot complete before the
	// whole next tile has landed.  The deferred waves therefore run their epilogue (norm loads,
	// rare atomics) BEFORE they issue their share of the staging; the other waves fetch the row
	// norms of the current tile before staging and keep them in 4 VGPRs until their epilogue.
	float bnv[4];
	int buf = 0;
	for (uint32_t t = t0; t < t1; ++t, buf ^= 1) {
		if (deferred) {
			if (have_prev && !(p.debug & 1)) {
				load_bn(bnv, prev_row0);
				epilogue(acc, prev_row0, bnv);
			}
		} else {
			load_bn(bnv, tile_row0(t));
		}
		if (t + 1 < t1 && !(p.debug & 2))
			stage(t + 1, buf ^ 1);
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
#pragma unroll
			for (int e = 0; e < 16; ++e)
				acc[tc][e] = 0.0f;
		// fragment of column tile tc, part (0 = hi, 1 = lo), k-step s: the swizzled chunk index
		// splits into a per-lane term (aoff[s], hoisted) and compile-time offsets
		const uint32_t boff = (uint32_t)buf * TILE_BYTES;
		auto frag = [&](int tc, int part, int s) -> bf16x8 {
			return *reinterpret_cast<const bf16x8*>(smem + (boff + aoff[s]) + tc * 32 * ROWB +
			                                        part * (CH / 2) * 16);
		};
		// 2*KS half-steps (k-step s, column-tile pair `half`); the fragments of half-step i+1 are
		// read from LDS while the six MFMAs of half-step i run
		bf16x8 bha = frag(0, 0, 0), bla = frag(0, 1, 0), bhb = frag(1, 0, 0), blb = frag(1, 1, 0);
		if (!(p.debug & 4))
#pragma unroll
		for (int hs = 0; hs < 2 * KS; ++hs) {
			const int s = hs >> 1, ta = 2 * (hs & 1), tb = ta + 1;
			bf16x8 nha = bha, nla = bla, nhb = bhb, nlb = blb;
			if (hs + 1 < 2 * KS) {
				const int ns = (hs + 1) >> 1, nta = 2 * ((hs + 1) & 1);
				nha = frag(nta, 0, ns);
				nla = frag(nta, 1, ns);
				nhb = frag(nta + 1, 0, ns);
				nlb = frag(nta + 1, 1, ns);
			}
			// small cross terms first, the dominant hi.hi product last
			acc[ta] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[s], bha, acc[ta], 0, 0, 0);
			acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[s], bhb, acc[tb], 0, 0, 0);
			acc[ta] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bla, acc[ta], 0, 0, 0);
			acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], blb, acc[tb], 0, 0, 0);
			acc[ta] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bha, acc[ta], 0, 0, 0);
			acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[s], bhb, acc[tb], 0, 0, 0);
			bha = nha;
			bla = nla;
			bhb = nhb;
			blb = nlb;
			// pin the issue order: the 4 fragment reads of the next half-step first, then the
			// 6 MFMAs of this one (hipcc otherwise sinks the reads to just before their use)
			__builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
			__builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
		}
		const uint32_t row0 = tile_row0(t);
		if (!deferred) {
			if (!(p.debug & 1))
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
