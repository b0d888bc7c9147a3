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
// Layout / geometry: rows of d fp16 (2d bytes), staged by LDS-DMA as in scan_gemm_bf16.hpp; 256
// queries x 128 rows per workgroup step; 8 waves as 4 (query groups of 64) x 2 (row halves of
// 64), each wave 2x2 MFMA tiles with its 64 queries' fragments in registers, so a row fragment
// read from LDS feeds two MFMAs (LDS read traffic = half the MFMA pipe time at d = 128).
//
// The test  bn - 2 q.b <= theta_q  runs in the scaled domain: the accumulators START at
// theta'_q = theta_q s^2/2, the MFMAs add q16.b16, and a row passes when  acc >= bn'_b =
// bn_b s^2/2  -- one compare per pair, folded per lane into a v_max3 tree per 32x32 tile, so
// the common (no candidate) path costs 9 VALU instructions per tile.
#pragma once
#include "common.hpp"
#include "scan_gemm_bf16.hpp"

namespace expann {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline float gemm_f16_filter_eps() { return 1.125f * 0.0009765625f; }

// fp32 [n_values] * scale -> fp16 (round to nearest even); *maxabs_bits (optional, zeroed by the
// caller) receives the bit pattern of max |in| -- for non-negative floats the unsigned integer
// order is the float order, and a NaN (0x7fc00000 and up) wins, which the range check rejects
__global__ __launch_bounds__(kBlock) void convert_f16_kernel(const float* in, size_t n_values,
                                                             float scale, _Float16* out,
                                                             uint32_t* maxabs_bits) {
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	float v = 0.0f;
	if (i < n_values) {
		v = in[i];
		out[i] = (_Float16)(v * scale);
	}
	if (maxabs_bits) {
		uint32_t b = __builtin_bit_cast(uint32_t, v) & 0x7fffffffu;
		for (int off = 32; off > 0; off >>= 1) {
			const uint32_t o = (uint32_t)__shfl_xor((int)b, off);
			b = o > b ? o : b;
		}
		if ((threadIdx.x & 63) == 0 && b != 0)
			atomicMax(maxabs_bits, b);
	}
}

// bit pattern of max |x| over an array (see convert_f16_kernel); *out zeroed by the caller
__global__ __launch_bounds__(kBlock) void maxabs_bits_kernel(const float* in, size_t n, uint32_t* out) {
	uint32_t b = 0;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
		const uint32_t v = __builtin_bit_cast(uint32_t, in[i]) & 0x7fffffffu;
		b = v > b ? v : b;
	}
	for (int off = 32; off > 0; off >>= 1) {
		const uint32_t o = (uint32_t)__shfl_xor((int)b, off);
		b = o > b ? o : b;
	}
	if ((threadIdx.x & 63) == 0 && b != 0)
		atomicMax(out, b);
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
// rows:    out = (nrm*(1-eps) - abs*sqrt(nrm)) * mul                (tau == nullptr)
// queries: out = (tau - (nrm*(1-eps) - abs*sqrt(nrm))) * mul
// mul = s^2/2, a power of two: the scaling is exact
__global__ __launch_bounds__(kBlock) void f16_terms_kernel(const float* nrm, uint32_t n, float eps,
                                                           float abs_coef, const float* tau,
                                                           float mul, float* out) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float v = nrm[i];
	const float t = v * (1.0f - eps) - abs_coef * __builtin_sqrtf(v);
	out[i] = (tau ? tau[i] - t : t) * mul;
}

struct GemmF16Params {
	const void* base_f16;    // [n_rows][D] fp16, scaled by s
	const float* bnorm;      // [n_rows] (||b||^2 (1-eps) - abs*|b|) * s^2/2
	uint32_t n_rows;
	uint32_t n_tiles_sel;
	uint32_t tile_stride;
	uint32_t tile_run;
	uint32_t tiles_per_block;
	uint32_t n_qtiles;
	const void* queries_f16; // [m][D] fp16, scaled by s
	const float* theta;      // [m] (tau - ||q||^2 (1-eps) + abs*|q|) * s^2/2
	float two_inv_s2;        // 2 / s^2 (exact: s is a power of two)
	uint32_t m;
	uint32_t* cand_cnt;
	uint64_t* cand;
	uint32_t cap;
	uint32_t debug;          // ablation switches (bench only): 1 no staging, 2 no MFMA, 4 no epilogue
};

__device__ inline float max3f(float a, float b, float c) {
	return __builtin_fmaxf(__builtin_fmaxf(a, b), c);  // folds to v_max3_f32
}

// Wait until at most N of this wave's vector-memory operations (LDS-DMA stage loads) are still
// in flight, then the workgroup barrier.  (__syncthreads() would drain them all: vmcnt(0).)
template <int N> __device__ inline void wait_vm_then_barrier() {
	asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int kF16Prefetch = 2;              // tiles in flight ahead of the one being multiplied
constexpr int kF16Bufs = kF16Prefetch + 1;   // LDS tile buffers
constexpr int kF16QueueCap = 640;            // candidate queue entries in LDS
constexpr int kF16EntryBytes = 80;
template <int D> constexpr int gemm_f16_lds_bytes() {
	// tiles + per-wave bn' slots + candidate queue + its counter
	return kF16Bufs * (kGemmTB * D * 2 + 8 * 256) + kF16QueueCap * kF16EntryBytes + 16;
}
static_assert(gemm_f16_lds_bytes<128>() <= 160 * 1024, "LDS budget");

template <int D>
__global__ __launch_bounds__(kGemmThreads, 2) void scan_gemm_f16_kernel(GemmF16Params p) {
	static_assert(D == 128 || D == 64, "built for d = 64, 128");
	constexpr int ROWB = D * 2;      // bytes per fp16 row
	constexpr int CH = ROWB / 16;    // 16-byte chunks per row
	constexpr int KS = D / 16;       // MFMA k-steps; lane half h of k-step s holds chunk h*KS + s
	constexpr int TILE_BYTES = kGemmTB * ROWB;
	constexpr int RPB = (ROWB < 256) ? 256 / ROWB : 1;  // rows per 256-byte LDS bank row
	constexpr int SWM = (CH < 16 ? CH : 16) - 1;
	constexpr int PF = kF16Prefetch, NBUF = kF16Bufs;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wq = wave & 3;   // query group: queries [64 wq, 64 wq + 64) of the workgroup's 256
	const int wh = wave >> 2;  // row half: rows [64 wh, 64 wh + 64) of each 128-row tile
	const int h = lane >> 5, r31 = lane & 31;
	const uint32_t qtile = blockIdx.x % p.n_qtiles;
	const uint32_t chunk = blockIdx.x / p.n_qtiles;
	const uint32_t q0 = qtile * kGemmBf16TQ + wq * 64;

	f16x8 a[2][KS];
#pragma unroll
	for (int tq = 0; tq < 2; ++tq) {
		uint32_t qi = q0 + tq * 32 + r31;
		if (qi >= p.m)
			qi = p.m - 1;
		const f16x8* src = reinterpret_cast<const f16x8*>((const unsigned char*)p.queries_f16 +
		                                                  (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			a[tq][s] = src[h * KS + s];
	}
	// accumulator start values: theta' of the query each accumulator register belongs to
	f32x16 th[2];
#pragma unroll
	for (int tq = 0; tq < 2; ++tq)
#pragma unroll
		for (int reg = 0; reg < 16; ++reg) {
			const uint32_t qi = q0 + tq * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
			th[tq][reg] = qi < p.m ? p.theta[qi] : -__builtin_inff();
		}
	// the query fragments and thresholds are in registers before the first stage load is issued:
	// a later wait for them would be a vmcnt(0) inside the loop and drain the prefetch queue
#pragma unroll
	for (int tq = 0; tq < 2; ++tq) {
#pragma unroll
		for (int s = 0; s < KS; ++s)
			asm volatile("" : "+v"(a[tq][s]));
		asm volatile("" : "+v"(th[tq]));
	}
	// per-lane LDS offset of k-step s (row r31 of this wave's first column tile); the second
	// column tile is 32 rows further, where the swizzle term is the same
	static_assert((32 / RPB) % (SWM + 1) == 0, "swizzle must repeat every 32 rows");
	uint32_t aoff[KS];
#pragma unroll
	for (int s = 0; s < KS; ++s)
		aoff[s] = (wh * 64 + r31) * ROWB + (((h * KS + s) ^ ((r31 / RPB) & SWM)) * 16);

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	if (t0 >= t1)
		return;  // (whole workgroup)
	auto tile_row0 = [&](uint32_t t) -> uint32_t {
		return ((t / p.tile_run) * (p.tile_stride * p.tile_run) + (t % p.tile_run)) * kGemmTB;
	};

	// Staging: every wave issues exactly N_STAGE + 1 LDS-DMA loads per tile (its share of the
	// tile and the bn' of its own 64 rows into a private slot), so the vmcnt arithmetic below is
	// the same for all waves.
	constexpr int ROWS_PER_INSTR = kGemmThreads / CH;
	static_assert(ROWS_PER_INSTR % (16 * RPB) == 0, "swizzle must be instruction-invariant");
	constexpr int N_STAGE = kGemmTB * CH / kGemmThreads;
	constexpr int LOADS = N_STAGE + 1;
	const uint32_t lane_row = tid / CH;
	const uint32_t lane_off = lane_row * ROWB + (((tid % CH) ^ ((lane_row / RPB) & SWM)) * 16);
	unsigned char* const bn_slots = smem + NBUF * TILE_BYTES;
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
		uint32_t brow = row0 + wh * 64 + lane;  // rows past the end: a copy of the last row's bn',
		if (brow >= p.n_rows)                   // dropped again by the row check in the epilogue
			brow = p.n_rows - 1;
		__builtin_amdgcn_global_load_lds(
		    (const __attribute__((address_space(1))) void*)(p.bnorm + brow),
		    (__attribute__((address_space(3))) void*)(bn_slots + (buf * 8 + wave) * 256), 4, 0, 0);
	};
	auto read_bn = [&](float (&bnv)[2], int buf) {
		const float* slot = reinterpret_cast<const float*>(bn_slots + (buf * 8 + wave) * 256);
		bnv[0] = slot[r31];
		bnv[1] = slot[32 + r31];
	};
	// Candidates: a lane whose 16 accumulators of a 32x32 tile hold at least one hit (about a
	// third of all tiles do, at ~32k candidates per query) appends the raw 16-vector, its bn', row
	// and first query to a queue in LDS -- one LDS atomic and five 16-byte stores, no per-register
	// branches in the MFMA loop.  The workgroup empties the queue when it is half full and at the
	// end: 16 threads per entry redo the compare and push the hits to the global per-query lists.
	struct QEntry {
		float acc[16];
		float bn;
		uint32_t row;
		uint32_t qrow0;  // query of accumulator register 0; register r is + (r & 3) + 8 (r >> 2)
		uint32_t pad;
	};
	static_assert(sizeof(QEntry) == kF16EntryBytes, "queue entry size");
	QEntry* const queue = reinterpret_cast<QEntry*>(bn_slots + NBUF * 8 * 256);
	uint32_t* const qcount = reinterpret_cast<uint32_t*>(queue + kF16QueueCap);
	if (tid == 0)
		*qcount = 0;  // (ordered before the first push by the barrier after the prologue stages)
	auto push_global = [&](uint32_t qi, uint64_t key) {
		const uint32_t slot = atomicAdd(&p.cand_cnt[qi], 1u);
		if (slot < p.cap)
			p.cand[(size_t)qi * p.cap + slot] = key;
	};
	// approximate key of a hit: bn(1-eps) - abs|b| - 2 q16.b16/s^2 = ((bn' - acc) + theta') * 2/s^2
	auto flush = [&]() {  // whole workgroup, after a barrier
		uint32_t n = *qcount;
		if (n > (uint32_t)kF16QueueCap)
			n = kF16QueueCap;
		for (uint32_t i = tid; i < n * 16; i += kGemmThreads) {
			const QEntry& e = queue[i >> 4];
			const uint32_t reg = i & 15;
			const float c = e.acc[reg], bn = e.bn;
			if (c >= bn) {
				const uint32_t qi = e.qrow0 + (reg & 3) + 8 * (reg >> 2);
				push_global(qi, make_key(((bn - c) + p.theta[qi]) * p.two_inv_s2, e.row));
			}
		}
		wait_vm_then_barrier<0>();
		if (tid == 0)
			*qcount = 0;
		wait_vm_then_barrier<0>();
	};
	auto epilogue = [&](const f32x16 (&accs)[2][2], uint32_t row0, const float (&bnv)[2]) {
#pragma unroll
		for (int tq = 0; tq < 2; ++tq)
#pragma unroll
			for (int tc = 0; tc < 2; ++tc) {
				const f32x16& c = accs[tq][tc];
				const float bn = bnv[tc];
				float m0 = max3f(c[0], c[1], c[2]);
				float m1 = max3f(c[3], c[4], c[5]);
				float m2 = max3f(c[6], c[7], c[8]);
				float m3 = max3f(c[9], c[10], c[11]);
				float m4 = max3f(c[12], c[13], c[14]);
				m0 = max3f(m0, m1, c[15]);
				m2 = max3f(m2, m3, m4);
				m0 = __builtin_fmaxf(m0, m2);
				if (__builtin_amdgcn_ballot_w64(m0 >= bn) != 0 && !(p.debug & 8)) {
					uint32_t qrow0 = q0 + tq * 32 + 4 * h;  // rare path: arithmetic stays in here
					asm volatile("" : "+v"(qrow0));
					const uint32_t brow = row0 + wh * 64 + tc * 32 + r31;
					if (m0 >= bn && brow < p.n_rows) {
						const uint32_t slot = atomicAdd(qcount, 1u);
						if (slot < (uint32_t)kF16QueueCap) {
							QEntry& e = queue[slot];
#pragma unroll
							for (int reg = 0; reg < 16; ++reg)
								e.acc[reg] = c[reg];
							e.bn = bn;
							e.row = brow;
							e.qrow0 = qrow0;
						} else {  // queue full (pathological thresholds): straight to the lists
#pragma unroll 1
							for (int reg = 0; reg < 16; ++reg) {
								float cr = c[0];
#pragma unroll
								for (int j = 1; j < 16; ++j)
									cr = reg == j ? c[j] : cr;
								if (cr >= bn) {
									const uint32_t qi = qrow0 + (reg & 3) + 8 * (reg >> 2);
									push_global(qi, make_key(((bn - cr) + p.theta[qi]) * p.two_inv_s2, brow));
								}
							}
						}
					}
				}
			}
	};
	// before the barrier that ends step t: tile t+1 must have landed; the stages issued after
	// it (tiles t+2 .. min(t+PF, t1-1)) may stay in flight
	auto end_of_step = [&](uint32_t t) {
		const uint32_t last = (t + PF < t1 - 1) ? t + PF : t1 - 1;
		const uint32_t later = last > t + 1 ? last - (t + 1) : 0;
		static_assert(PF == 2, "cases below");
		if (later == 1)
			wait_vm_then_barrier<1 * LOADS>();
		else
			wait_vm_then_barrier<0>();
	};

#pragma unroll
	for (int i = 0; i < PF; ++i)
		if (t0 + i < t1)
			stage(t0 + i, i);
	{  // tile t0 landed (same count as "end of step t0 - 1")
		const uint32_t last = (t0 + PF - 1 < t1 - 1) ? t0 + PF - 1 : t1 - 1;
		const uint32_t later = last - t0;
		if (later == 1)
			wait_vm_then_barrier<1 * LOADS>();
		else
			wait_vm_then_barrier<0>();
	}

	// Waves w and w+4 share a SIMD (and the same queries): the upper four run their epilogue one
	// step late, so that on each SIMD one wave's compares overlap the other's MFMAs.
	const bool deferred = wave >= 4;
	f32x16 acc[2][2];
	uint32_t prev_row0 = 0;
	bool have_prev = false;
	float bnv[2];
	int buf = 0, pbuf = PF;  // pbuf: buffer that tile t+PF goes to (= the one tile t-1 used)
	for (uint32_t t = t0; t < t1; ++t) {
		if (t + PF < t1 && !(p.debug & 1))
			stage(t + PF, pbuf);
		if (deferred && have_prev && !(p.debug & 4))
			epilogue(acc, prev_row0, bnv);
		const uint32_t boff = (uint32_t)buf * TILE_BYTES;
		auto frag = [&](int tc, int s) -> f16x8 {
			return *reinterpret_cast<const f16x8*>(smem + (boff + aoff[s]) + tc * 32 * ROWB);
		};
		if (p.debug & 2) {
#pragma unroll
			for (int tq = 0; tq < 2; ++tq)
				acc[tq][0] = acc[tq][1] = th[tq];
		} else
#pragma unroll
		for (int s = 0; s < KS; ++s) {
			const f16x8 b0 = frag(0, s), b1 = frag(1, s);
#pragma unroll
			for (int tq = 0; tq < 2; ++tq) {
				acc[tq][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tq][s], b0,
				                                                    s == 0 ? th[tq] : acc[tq][0], 0, 0, 0);
				acc[tq][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tq][s], b1,
				                                                    s == 0 ? th[tq] : acc[tq][1], 0, 0, 0);
			}
		}
		read_bn(bnv, buf);
		const uint32_t row0 = tile_row0(t);
		if (!deferred) {
			if (!(p.debug & 4))
				epilogue(acc, row0, bnv);
		} else {
			prev_row0 = row0;
			have_prev = true;
		}
		end_of_step(t);
		if (*qcount >= (uint32_t)kF16QueueCap / 2)
			flush();
		pbuf = buf;
		buf = buf + 1 == NBUF ? 0 : buf + 1;
	}
	if (deferred && have_prev && !(p.debug & 4))
		epilogue(acc, prev_row0, bnv);
	wait_vm_then_barrier<0>();
	flush();
}

}  // namespace expann
