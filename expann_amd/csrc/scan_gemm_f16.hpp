// scan_gemm_f16.hpp -- the GEMM-form fp32 L2 candidate filter with ONE fp16 product per element
// on the matrix cores: a third of the MFMA work of the bf16x3 form (scan_gemm_bf16.hpp) and half its
// tile bytes.  This header holds the analysis and everything the fp16 kernels share; the kernels
// themselves are scan_gemm_f16x.hpp / f16y / f16kx (v_mfma_f32_16x16x32_f16).
//
// Every value is scaled by a per-index power of two s (so that max|x|*s <= 2^15) and rounded
// once to fp16:  |x*s - fp16(x*s)| <= 2^-11 |x*s| + 2^-25  (normal + subnormal range).  Hence
//   |q.b - q16.b16 / s^2|  <=  (2^-10 + 2^-22) sum|q_i b_i|  +  2^-25/s * sum(|q_i| + |b_i|)
// and the squared-L2 test in expanded form keeps every row whose reference-order score is
// <= tau_q when evaluated with the slack
//      1.125 * 2^-10 * (||q||^2 + ||b||^2)  +  2^-24/s * sqrt(d) * (|q| + |b|)
// (the 12.5 % on top of 2^-10 covers the fp32 accumulation, the norms and the reference-order
// rounding, (4.25 d + 256) * 2^-24 at most, up to d = 256; beyond, gemm_f16_filter_eps(d) grows
// with d).  The slack admits ~10 % more candidates than the exact
// test; they are re-scored exactly by the select kernel, so ids and distances stay bit-identical
// to the direct scan.  Queries whose scaled components would leave the fp16 range make the
// engine use the bf16x3 form for that search.
//
// Layout / geometry: rows of d fp16 (2d bytes), staged by LDS-DMA as in scan_gemm_bf16.hpp.  A
// workgroup is 4 waves = 256 queries x 64 rows per step; each wave holds its 64 queries'
// fragments in registers and runs 2x2 MFMA tiles per step, so a row fragment read from LDS
// feeds two MFMAs.  Two workgroups share a CU (<= 80 KB of LDS, <= 256 VGPRs): the two waves on
// a SIMD belong to different workgroups with separate barriers, so one workgroup's barrier /
// epilogue / flush bubbles are covered by the other's MFMAs.
//
// The test  bn - 2 q.b <= theta_q  runs in the scaled domain: the accumulators START at
// theta'_q = theta_q s^2/2, the MFMAs add q16.b16, and a row passes when  acc >= bn'_b =
// bn_b s^2/2  -- one compare per pair, folded per lane into a v_max3 tree per 32x32 tile, so
// the common (no candidate) path costs 9 VALU instructions per tile.
//
// The fp16 copy of the index is padded to a multiple of 64 rows (zero rows whose bn' is NaN:
// no accumulator compares >= NaN), so staging needs no bounds handling.
#pragma once
#include "common.hpp"
#include "scan_gemm_bf16.hpp"
#include "select.hpp"

namespace expann {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// relative slack of the filter on (||q||^2 + ||b||^2).  Beyond the input rounding (2^-10 + 2^-22)
// it has to cover, in units of u = 2^-24: d roundings of the fp32 accumulation at any association,
// each on a running magnitude <= |theta'| + sum|q16 b16| <= 4 (||q||^2 + ||b||^2) mul  (4d), the
// reference-order score (d/8 + 14), the two norms (d/16 + 5 each) and the threshold terms (~20):
// <= 4.25 d + 256.  Up to d = 256 the flat 12.5 % (2048 u) is at least that.
__host__ __device__ inline float gemm_f16_filter_eps(int d) {
	return d <= 256 ? 1.125f * 0.0009765625f
	                : 0.0009765625f + (4.0f + 4.25f * (float)d + 256.0f) * 5.9604644775390625e-08f;
}

// fp32 [n_values] * scale -> fp16 (round to nearest even); *maxabs_bits (optional, zeroed by the
// caller) receives the bit pattern of max |in| -- for non-negative floats the unsigned integer
// order is the float order, and a NaN (0x7fc00000 and up) wins, which the range check rejects
__global__ __launch_bounds__(kBlock) void convert_f16_kernel(const float* in, size_t n_values,
                                                             float scale, _Float16* out,
                                                             uint32_t* maxabs_bits) {
	uint32_t b = 0;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_values;
	     i += (size_t)gridDim.x * kBlock) {
		const float v = in[i];
		out[i] = (_Float16)(v * scale);
		const uint32_t vb = __builtin_bit_cast(uint32_t, v) & 0x7fffffffu;
		b = vb > b ? vb : b;
	}
	if (maxabs_bits) {  // one atomic per workgroup
		__shared__ uint32_t red[kBlock / 64];
		for (int off = 32; off > 0; off >>= 1) {
			const uint32_t o = (uint32_t)__shfl_xor((int)b, off);
			b = o > b ? o : b;
		}
		if ((threadIdx.x & 63) == 0)
			red[threadIdx.x >> 6] = b;
		__syncthreads();
		if (threadIdx.x == 0) {
			for (int w = 1; w < kBlock / 64; ++w)
				b = red[w] > b ? red[w] : b;
			if (b != 0)
				atomicMax(maxabs_bits, b);
		}
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
// Query side of a search in ONE launch: scaled fp16 copy, ||q||^2 in the reference lane order and
// the bit pattern of max |q| (range check).  16 lanes per query row, as sqnorm_kernel.
// Latency mode (m <= 16: a single workgroup): q may be pinned host memory -- the kernel then also
// leaves a device copy in q_copy for the kernels that follow -- and zero_flags (the 8-word flag /
// statistics block that maxabs_bits points into) is cleared here instead of by a memset launch.
// Inner product (ip_ref != nullptr, round 3): the filter's slack eps (|q|^2 + |b|^2) bounds 2 eps |q| |b| tightly
// only when the two norms are alike.  Ranks under the inner product do not change when a query is multiplied by
// a positive constant, so the FILTER sees each query times a power of two c_q (exact) that brings |c_q q|^2 to
// *ip_ref (the largest row norm: the rows that reach a top-k under the inner product are the large ones):
// q16, qnrm and the range check are those of c_q q, qscale[i] = c_q; thresholds that arrive in true units are
// multiplied by it (f16_terms_kernel), the sampled pass's own are in the scaled domain already, and the exact
// re-rank reads the caller's queries.  (L2: c_q = 1.)
template <int D>
__global__ __launch_bounds__(kBlock) void f16_query_prep_kernel(const float* q, uint32_t m, float scale,
                                                                _Float16* q16, float* qnrm,
                                                                uint32_t* maxabs_bits, float* q_copy,
                                                                uint32_t* zero_flags, const float* ip_ref,
                                                                float* qscale) {
	__shared__ uint32_t red[kBlock / 64];
	if (zero_flags) {
		if (threadIdx.x < 8)
			zero_flags[threadIdx.x] = 0;
		__syncthreads();
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l = lane & 15, rg = lane >> 4;
	const uint32_t i = blockIdx.x * kRowsPerGroup + wave * kRowsPerWaveStep + rg;
	const uint32_t ii = i < m ? i : m - 1;
	const float* src = q + (size_t)ii * D + l;
	float acc = 0.0f;
	uint32_t b = 0;
	float vals[D / 16];
#pragma unroll
	for (int t = 0; t < D / 16; ++t) {
		const float v = src[16 * t];
		vals[t] = v;
		acc = __builtin_fmaf(v, v, acc);
		const uint32_t vb = __builtin_bit_cast(uint32_t, v) & 0x7fffffffu;
		b = vb > b ? vb : b;
	}
	acc = reduce16_ref_order(acc);
	acc = __shfl(acc, lane & ~15);  // (the reference order is the first lane's: one value for the 16 lanes)
	float c = 1.0f;
	if (ip_ref) {
		const float ref = ip_ref[0];
		if (acc > 0.0f && ref > 0.0f && acc < 3.0e38f && ref < 3.0e38f) {
			int e = (int)__builtin_rintf(0.5f * (__builtin_log2f(ref) - __builtin_log2f(acc)));
			e = e < -40 ? -40 : (e > 40 ? 40 : e);
			c = __builtin_ldexpf(1.0f, e);
		}
	}
#pragma unroll
	for (int t = 0; t < D / 16; ++t) {
		if (i < m) {
			q16[(size_t)i * D + l + 16 * t] = (_Float16)(vals[t] * c * scale);
			if (q_copy)
				q_copy[(size_t)i * D + l + 16 * t] = vals[t];
		}
	}
	if (i < m && l == 0) {
		qnrm[i] = acc * c * c;
		if (qscale)
			qscale[i] = c;
	}
	b = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, b) * c);  // (max |c_q q|: what the fp16 range check is about)
	for (int off = 32; off > 0; off >>= 1) {
		const uint32_t o = (uint32_t)__shfl_xor((int)b, off);
		b = o > b ? o : b;
	}
	if (lane == 0)
		red[wave] = b;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < kBlock / 64; ++w)
			b = red[w] > b ? red[w] : b;
		if (b != 0)
			atomicMax(maxabs_bits, b);
	}
}

// rows:    out = (nrm*(1-eps) - abs*sqrt(nrm)) * mul                (tau == nullptr)
// queries: out = (tau - (nrm*(1-eps) - abs*sqrt(nrm))) * mul
// mul = s^2/2, a power of two: the scaling is exact
// (eps, abs_coef negated: the UPPER row term (nrm*(1+eps) + abs*sqrt(nrm)) * mul of the sampled pass)
//
// Inner product (score = -q.b, ip != 0): |q.b - q16.b16/s^2| <= E/2 with the same
// E = eps (||q||^2 + ||b||^2) + abs (|q| + |b|), so with e(v) = eps v + abs sqrt(v):
//   rows:    out = -e(nrm) * mul          queries: out = (2 tau + e(nrm)) * mul
// and the kernel's test acc = theta' + q16.b16 >= bn' keeps every row with -q.b <= tau.
// (qscale: the queries' filter-side factors c_q of f16_query_prep_kernel -- inner product only, nullptr = 1 -- :
// nrm is already that of c_q q, a threshold in true units becomes c_q tau)
__global__ __launch_bounds__(kBlock) void f16_terms_kernel(const float* nrm, uint32_t n, float eps,
                                                           float abs_coef, const float* tau,
                                                           float mul, float* out, int ip, const float* qscale) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float v = nrm[i];
	if (ip) {
		const float e = v * eps + abs_coef * __builtin_sqrtf(v);
		out[i] = (tau ? 2.0f * tau[i] * (qscale ? qscale[i] : 1.0f) + e : -e) * mul;
		return;
	}
	const float t = v * (1.0f - eps) - abs_coef * __builtin_sqrtf(v);
	out[i] = (tau ? tau[i] - t : t) * mul;
}

struct GemmF16Params {
	const void* base_f16;    // [n_rows padded to 64][D] fp16, scaled by s
	const float* bnorm;      // [n_rows padded to 64] (||b||^2 (1-eps) - abs*|b|) * s^2/2; NaN padding
	uint32_t n_rows;
	uint32_t n_tiles_sel;
	uint32_t tile_stride;
	uint32_t tile_run;
	uint32_t tiles_per_block;
	uint32_t n_qtiles;
	const void* queries_f16; // [m][D] fp16, scaled by s
	const float* theta;      // [m] (tau - ||q||^2 (1-eps) + abs*|q|) * s^2/2
	float two_inv_s2;        // 2 / s^2 (exact: s is a power of two); inner product: 1 / s^2
	uint32_t m;
	uint32_t* cand_cnt;
	uint64_t* cand;
	uint32_t cap;
	uint32_t debug;          // ablation switches (bench only): 4 no epilogue, 8 no candidate path
	unsigned long long* clk; // [2] debug: shader-clock and 100 MHz ticks of workgroup 0 (or nullptr)
	// SAMPLE variant: out[(q * n_chunks + chunk) * 32 + c] = max over the chunk's rows of class c
	// (row mod 32) of g = q16.b16 - bn'  (larger g = nearer row); theta, cand* unused
	float* sample_out;
	uint32_t n_chunks;
	// blocks b and b+8 share an XCD (observed dispatch order): with xcd_map the 8 row chunks
	// {x, x+8, ...} and all their query tiles go to XCD x, so a tile is fetched into ONE L2
	// instead of eight.  Needs the chunk count to be a multiple of 8.
	uint32_t xcd_map;
	// scan_gemm_f16x_kernel only: row chunks of two sizes.  Chunks 0 .. n_big - 1 hold tiles_per_block
	// tiles, the chunks behind them tiles_small (0 = one size).  Workgroups start in block order, so
	// the launch ends on the small ones and drains in a third of the time (pick_tail_chunks).
	uint32_t n_big, tiles_small;
	// scan_gemm_f16x_kernel only: every wave appends its hits to a log of its own in global memory --
	// log[(4 blockIdx.x + wave) * log_cap + i] = {bn' - acc, row, query}, plain stores in the wave's own order, no
	// atomic and no returned value to wait for inside the MFMA kernel -- and leaves the count in
	// log_cnt; scatter_log_kernel then files the entries into the per-query lists (cand, cand_cnt).
	// *lost is incremented by a wave whose log or LDS queue overflowed (the host repeats the search
	// with scan_gemm_f16_kernel, whose direct appends have no such limit).
	uint4* log;
	uint32_t* log_cnt;
	uint32_t log_cap;
	uint32_t* lost;
	// scan_gemm_f16x_kernel only: persistent launch -- work_ctr[16 * x] = the next item of XCD x's queue
	// (zero at launch, one cache line each), n_items = the plain launch's grid; nullptr = plain launch
	uint32_t* work_ctr;
	uint32_t n_items;
};

// Per-wave hit logs -> per-query candidate lists.  The 64 queries of (query tile, wave w) receive
// hits from exactly the n_chunks logs of wave w in the workgroups (chunk, query tile); a workgroup
// here takes `chunks_per_block` of those logs: it counts its entries per query in LDS, reserves the
// slots of each query with ONE global atomicAdd (64 per workgroup instead of one per hit), then
// files the entries.  cand_cnt is zero when the kernel starts (sample_tau_kernel / the level's memset).
// grid = n_qtiles * 4 * n_groups; block index = (qtile * 4 + w) * n_groups + group.
struct GatherLogParams {
	const uint4* log;
	const uint32_t* log_cnt;
	uint32_t log_cap, n_chunks, n_qtiles, xcd_map, m;
	uint32_t n_groups, chunks_per_block;
	uint32_t* cand_cnt;
	uint64_t* cand;
	uint32_t cap;
	// fp16 logs (scan_gemm_f16x.hpp): an entry is {bits of bn' - acc, row, query}; the key is made HERE,
	// make_key((x + theta[query]) * key_mul, row) -- the arithmetic the scan's flush used to do behind a second
	// LDS round trip.
	const float* theta;
	float key_mul;
	// 8-bit logs (scan_gemm_i8w.hpp): an entry is {raw accumulator, row, query}; the exact integer score is made
	// HERE -- dot = acc - i_thp[query], L2 forms: i_bias[row] - 2 dot + i_qself[query], inner product: -dot --
	// instead of two global gathers per hit inside the MFMA kernel (whose vmcnt they shared with the stage
	// loads).  i_mode: 0 = not an 8-bit log, 1 = L2 forms, 2 = inner product.
	const int* i_thp;
	const int* i_bias;
	const int* i_qself;
	int i_mode;
};
__global__ __launch_bounds__(kBlock) void gather_logs_kernel(GatherLogParams p) {
	__shared__ uint32_t cnt[64], base[64];
	const uint32_t grp = blockIdx.x % p.n_groups, qw = blockIdx.x / p.n_groups;
	const uint32_t qtile = qw >> 2, w = qw & 3;
	const uint32_t q0 = qtile * 256 + w * 64;
	if (threadIdx.x < 64)
		cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t c0 = grp * p.chunks_per_block;
	const uint32_t c1 = c0 + p.chunks_per_block < p.n_chunks ? c0 + p.chunks_per_block : p.n_chunks;
	auto log_of = [&](uint32_t c, uint32_t& n) -> const uint4* {
		const uint32_t bid = p.xcd_map ? ((c >> 3) * p.n_qtiles + qtile) * 8 + (c & 7) : c * p.n_qtiles + qtile;
		const uint32_t li = bid * 4 + w;
		n = p.log_cnt[li];
		if (n > p.log_cap)
			n = p.log_cap;
		return p.log + (size_t)li * p.log_cap;
	};
	for (uint32_t c = c0 + wv; c < c1; c += kBlock / 64) {  // one wave per log
		uint32_t n;
		const uint4* src = log_of(c, n);
		for (uint32_t i = lane; i < n; i += 64)
			atomicAdd(&cnt[(src[i].z - q0) & 63], 1u);
	}
	__syncthreads();
	if (threadIdx.x < 64) {
		const uint32_t n = cnt[threadIdx.x];
		base[threadIdx.x] = n ? atomicAdd(&p.cand_cnt[q0 + threadIdx.x], n) : 0u;
		cnt[threadIdx.x] = 0;
	}
	__syncthreads();
	for (uint32_t c = c0 + wv; c < c1; c += kBlock / 64) {
		uint32_t n;
		const uint4* src = log_of(c, n);
		for (uint32_t i = lane; i < n; i += 64) {
			const uint4 e = src[i];
			const uint32_t ql = (e.z - q0) & 63;
			const uint32_t slot = base[ql] + atomicAdd(&cnt[ql], 1u);
			if (slot < p.cap) {
				uint64_t key;
				if (p.i_mode) {
					const int dot = (int)e.x - p.i_thp[e.z];
					const int score = p.i_mode == 1 ? p.i_bias[e.y] - 2 * dot + p.i_qself[e.z] : -dot;
					key = make_key((float)score, e.y);
				} else {
					key = make_key((__builtin_bit_cast(float, e.x) + p.theta[e.z]) * p.key_mul, e.y);
				}
				p.cand[(size_t)e.z * p.cap + slot] = key;
			}
		}
	}
}

__device__ inline float max3f(float a, float b, float c) {
	return __builtin_fmaxf(__builtin_fmaxf(a, b), c);  // folds to v_max3_f32
}

// Wait until at most N of this wave's vector-memory operations (LDS-DMA stage loads) are still
// in flight and its LDS traffic has drained, then the workgroup barrier.  (__syncthreads() would
// drain the stage loads too: vmcnt(0).)
template <int N> __device__ inline void wait_vm_then_barrier() {
	asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

#ifndef EXPANN_F16_THREADS
#define EXPANN_F16_THREADS 256
#endif
constexpr int kF16Threads = EXPANN_F16_THREADS;  // 256: two workgroups per CU; 512: one
constexpr int kF16WgPerCu = 512 / kF16Threads;
constexpr int kF16Waves = kF16Threads / 64;
constexpr int kF16TQ = 64 * kF16Waves;       // queries per workgroup
static_assert(kF16Waves == 4 && kF16TQ == 256, "gather_logs_kernel files the logs of 4 waves x 64 queries per workgroup");
constexpr int kF16TB = 64;                   // rows per tile (= per workgroup step)
constexpr int kF16Prefetch = 2;              // tiles in flight ahead of the one being multiplied
constexpr int kF16Bufs = kF16Prefetch + 1;   // LDS tile buffers
constexpr int kF16WaveQueue = 88;            // candidate queue entries per wave, in LDS
constexpr int kF16EntryBytes = 80;
constexpr int kF16FlushEvery = 8;            // steps between looks at the queue fill
// Geometry by dimension (the int8 kernels use the same scheme, scan_gemm_i8q.hpp).  d = 64 / 128:
// 4 waves x 64 queries, two workgroups per CU.  d = 256: a wave's fragments of 64 queries would
// be 128 VGPRs, so 8 waves x 32 queries share each staged tile (one workgroup per CU), and the
// 16-byte chunks of a row go to (k-step, lane half) in natural order (2s + h): 8 fragment
// address registers serve all 16 k-steps.
template <int D> struct F16Geom {
	static constexpr int THREADS = D >= 256 ? 512 : kF16Threads;
	static constexpr int WAVES = THREADS / 64;
	static constexpr int TQW = D >= 256 ? 1 : 2;   // 32-query MFMA tiles per wave
	static constexpr int WGQ = WAVES * 32 * TQW;   // queries per workgroup
	static constexpr int NBUF = D >= 512 ? 2 : kF16Bufs;  // d = 512: two 64 KB tile buffers
	// d = 64: fragments of 64 queries are 32 VGPRs; with the accumulator start values read from
	// LDS each step the scan fits 168 VGPRs and THREE workgroups share a CU (as I8qGeom<128>)
	static constexpr bool TH_LDS = D == 64;
	static constexpr int QCAP = D >= 512 ? 32 : ((D >= 256 || TH_LDS) ? 56 : kF16WaveQueue);
	static constexpr bool NATURAL = D >= 256;
	static constexpr int WG_PER_CU = TH_LDS ? 3 : 512 / THREADS;
};
static_assert(F16Geom<128>::WGQ == kF16TQ && F16Geom<256>::WGQ == kF16TQ, "one query-tile size");
template <int D> constexpr int gemm_f16_lds_bytes() {
	using G = F16Geom<D>;
	// tiles + per-wave bn' slots + per-wave candidate queues + theta' + queue fills
	return G::NBUF * (kF16TB * D * 2 + G::WAVES * 256) + G::WAVES * G::QCAP * kF16EntryBytes + G::WGQ * 4 + 64 +
	       (G::TH_LDS ? G::WAVES * 2 * G::TQW * 16 * 4 : 0);
}
static_assert(gemm_f16_lds_bytes<64>() * 3 <= 160 * 1024 &&
                  gemm_f16_lds_bytes<128>() * F16Geom<128>::WG_PER_CU <= 160 * 1024 &&
                  gemm_f16_lds_bytes<256>() <= 160 * 1024 && gemm_f16_lds_bytes<512>() <= 160 * 1024,
              "LDS budget per CU");

// (Rounds 1-2 ran this filter -- full scan and sampled pass -- on v_mfma_f32_32x32x16_f16 in a kernel of this
// file, scan_gemm_f16_kernel<D, SAMPLE>; since round 3 every fp16 stream is one of the 16 x 16 x 32 kernels
// scan_gemm_f16x (d = 64 / 128), f16y (256 / 512), f16kx (768 - 960), each with a SAMPLE instance.  What stays
// here is what they share: parameters, geometry and LDS budget, the index / query conversions, the hit logs'
// gather, the thresholds from the sampled pass.)

// tau[q] = an upper bound of the k-th smallest reference-order score over the sampled rows, from
// the n_vals class maxima of g written by the SAMPLE pass.  The SAMPLE pass subtracts the UPPER
// row term  bns = (||b||^2 (1+eps) + abs |b|) * mul  (the full scan subtracts the lower one), so
// for every row   score <= ||q||^2 (1+eps) + abs |q| - g/mul,   mul = s^2/2,   by the slack
// analysis at the top of this file -- row by row, with no worst-case norm of the index in it
// (one outlier row would otherwise inflate every threshold).  The k largest maxima belong to k
// different rows.  One wave per query.
struct SampleTauParams {
	const float* vals;   // [m][n_vals]
	uint32_t n_vals;
	uint32_t m;
	uint32_t k;
	const float* qnrm;   // [m] ||q||^2
	float eps, abs_coef, inv_mul;
	int ip;              // inner product: score <= e(||q||^2)/2 - g/s^2 (the SAMPLE pass subtracts +e(||b||^2) mul)
	float* tau;          // [m]
	uint32_t* tau_row;   // [m] <- 0xFFFFFFFF (no row tie-break: the GEMM forms do not use it)
	float* theta;        // [m] <- (tau - (||q||^2 (1-eps) - abs |q|)) * mul, as f16_terms_kernel
	float mul;
	uint32_t* cand_cnt;  // [m] <- 0 (the full scan's list counters)
	const float* qscale; // inner product: the queries' filter-side factors c_q (tau is stored in true units: / c_q); or nullptr
};
// one thread: ord = ordered bits of the k-th largest g (0: fewer than k values) -> tau, theta', counter
__device__ inline void sample_tau_finish(const SampleTauParams& p, uint32_t qi, uint32_t ord) {
	float tau = __builtin_inff();
	const float qn = p.qnrm[qi];
	if (ord != 0) {
		const float g = ordered_to_float(ord);
		if (p.ip)
			tau = 0.5f * (qn * p.eps + p.abs_coef * __builtin_sqrtf(qn)) - g * (0.5f * p.inv_mul);
		else
			tau = qn * (1.0f + p.eps) + p.abs_coef * __builtin_sqrtf(qn) - g * p.inv_mul;
		if (!(tau == tau))
			tau = __builtin_inff();
	}
	p.tau[qi] = (p.ip && p.qscale) ? tau / p.qscale[qi] : tau;  // (true units; theta' below stays in the filter's domain)
	p.tau_row[qi] = 0xFFFFFFFFu;
	p.theta[qi] = p.ip ? (2.0f * tau + qn * p.eps + p.abs_coef * __builtin_sqrtf(qn)) * p.mul
	                   : (tau - (qn * (1.0f - p.eps) - p.abs_coef * __builtin_sqrtf(qn))) * p.mul;
	p.cand_cnt[qi] = 0;
}
// The same threshold by a whole 256-thread workgroup (n_vals <= 2048): radix select on the ordered
// bits, 8 bits per pass -- histogram in LDS, suffix sums by an 8-step scan, the thread whose bin
// holds the k-th largest fixes the digit.  A lone wave issues one instruction per >= 4 cycles, so
// the one-wave selection costs ~15 us when nothing else runs (the last workgroup of
// sample_direct_f16_kernel); this is ~2 us.  hist / ctl: 256 + 2 words of LDS.
__device__ inline void sample_tau_query_wg(const SampleTauParams& p, uint32_t qi, uint32_t* hist, uint32_t* ctl) {
	const uint32_t tid = threadIdx.x;
	const float* v = p.vals + (size_t)qi * p.n_vals;
	uint32_t ord[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const uint32_t i = j * kBlock + tid;
		ord[j] = i < p.n_vals ? float_to_ordered(v[i]) : 0u;
	}
	uint32_t prefix = 0, mask = 0, krem = p.k;
	for (int pass = 3; pass >= 0; --pass) {
		const int sh = 8 * pass;
		__syncthreads();  // (the previous pass's readers are done with hist / ctl)
		hist[tid] = 0;
		if (tid == 0)
			ctl[0] = 0xFFFFFFFFu;
		__syncthreads();
#pragma unroll
		for (int j = 0; j < 8; ++j)
			if ((ord[j] & mask) == prefix)
				atomicAdd(&hist[(ord[j] >> sh) & 255u], 1u);
		__syncthreads();
		const uint32_t mine = hist[tid];
		// inclusive suffix sums: hist[t] <- sum of bins >= t
		for (int off = 1; off < 256; off <<= 1) {
			const uint32_t add = tid + off < 256 ? hist[tid + off] : 0u;
			__syncthreads();
			hist[tid] += add;
			__syncthreads();
		}
		const uint32_t above = hist[tid] - mine;  // values with a larger digit
		if (mine != 0 && above < krem && krem <= above + mine) {
			ctl[0] = tid;
			ctl[1] = krem - above;
		}
		__syncthreads();
		const uint32_t digit = ctl[0];
		if (digit == 0xFFFFFFFFu) {  // fewer than k values left: no threshold
			prefix = 0;
			break;
		}
		krem = ctl[1];
		prefix |= digit << sh;
		mask |= 0xFFu << sh;
	}
	if (tid == 0)
		sample_tau_finish(p, qi, prefix);
}
// the work of one wave for query qi; scratch = 64 words of LDS of this wave's own
template <int PER>  // values per lane: n_vals <= 64 * PER
__device__ inline void sample_tau_query(const SampleTauParams& p, uint32_t qi, int lane, uint32_t* scratch) {
	const float* v = p.vals + (size_t)qi * p.n_vals;
	uint64_t keys[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j) {
		const uint32_t i = j * 64 + lane;
		// larger g first: order by ~ordered(g); the index keeps equal values apart
		keys[j] = i < p.n_vals ? ((uint64_t)float_to_ordered(v[i]) << 32) | (0xFFFFFFFFu - i) : 0ull;
	}
	// bisection on the ordered value (32 steps whatever k and the list length; k <= 64: over the few
	// values that reach the k-th largest lane maximum)
	uint32_t ord[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j)
		ord[j] = (uint32_t)(keys[j] >> 32);
	const uint64_t kth = (uint64_t)(p.k <= 64 ? wave_kth_largest_sparse_u32<PER>(ord, p.k, scratch, lane)
	                                          : wave_kth_largest_u32<PER>(ord, p.k))
	                     << 32;
	if (lane == 0)
		sample_tau_finish(p, qi, (uint32_t)(kth >> 32));
}
template <int PER>
__global__ __launch_bounds__(kBlock) void sample_tau_kernel(SampleTauParams p) {
	__shared__ uint32_t scratch[kBlock / 64][64];
	const uint32_t qi = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (qi >= p.m)
		return;  // (whole wave)
	sample_tau_query<PER>(p, qi, threadIdx.x & 63, scratch[threadIdx.x >> 6]);
}

}  // namespace expann
