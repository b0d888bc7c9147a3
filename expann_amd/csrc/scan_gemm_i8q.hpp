// scan_gemm_i8q.hpp -- the 8-bit GEMM-form filter in the geometry of scan_gemm_f16.hpp
// (256-thread workgroups, two per CU; 64 queries x 64 rows per wave and step; LDS-DMA staging; per-wave
// hit queues in LDS; one sampled pass for the threshold) on the int8 matrix cores: the analysis and
// everything the kernels share.  The kernels themselves are scan_gemm_i8w.hpp / scan_gemm_i8x.hpp.
// (I8qGeom below: d = 768 trades the two-workgroups-per-CU layout for 8 waves on one tile.)
//
// Integer arithmetic is exact, so the filter needs no slack and no re-rank.  Everything runs in
// the "g domain":  g(q, b) = q.b - bp[b],  bp[b] = floor(bias[b] / 2),  bias[b] = sum b^2 (L2
// forms; 0 for the inner product).  With score = qself + bias - 2 q.b (L2) or -q.b (IP):
//     score(q, b)  =  qself - 2 g - (bias & 1)    (L2)          score = -g   (IP)
// so larger g = nearer row, up to the parity bit.  The sampled pass returns class maxima of g;
// the k-th largest of them, g_k, belongs to k different rows, and the full scan keeps every
// row with g >= g_k: the accumulators start at -g_k and a row passes when acc >= bp.  A kept row
// may score one unit above the k-th (parity), never the reverse, so no neighbour is lost;
// candidates leave the kernel with their exact integer scores.
// uint8 rows (src/antitopo_engine.h:38-61) are mapped to int8 by x ^ 0x80 on both sides when
// the engine makes its padded copy of the index (differences, hence scores, unchanged).
#pragma once
#include "common.hpp"
#include "scan_gemm_f16.hpp"
#include "scan_gemm_i8.hpp"

namespace expann {

constexpr int kI8qPadBp = 1 << 30;  // bp of the padding rows: no accumulator reaches it (flush checks the row too)

struct GemmI8qParams {
	const void* base;        // [n_rows padded to 64][D] int8 (uint8 rows already ^ 0x80)
	const int* bp;           // [padded] floor(bias/2); kI8qPadBp on the padding
	const int* bias;         // [n_rows] sum b^2 (L2 forms) or nullptr (IP)
	uint32_t n_rows;
	uint32_t n_tiles_sel;
	uint32_t tile_stride;
	uint32_t tile_run;
	uint32_t tiles_per_block;
	uint32_t n_qtiles;
	const void* queries;     // [m][D] int8 (uint8 queries already ^ 0x80)
	const int* thp;          // [m] accumulator start = -g_k
	const int* qself;        // [m] sum q^2 (L2 forms) or nullptr
	uint32_t m;
	uint32_t* cand_cnt;
	uint64_t* cand;
	uint32_t cap;
	int* sample_out;         // SAMPLE: [(q * n_chunks + chunk) * 32 + class] max g
	uint32_t n_chunks;
	uint32_t xcd_map;        // XCD-aware block placement, as GemmF16Params::xcd_map
};

// bp[i] = bias[i] >> 1 for i < n, kI8qPadBp for n <= i < n_pad (bias == nullptr: zeros)
__global__ __launch_bounds__(kBlock) void i8q_bp_kernel(const int* bias, uint32_t n, uint32_t n_pad,
                                                        int* bp) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i < n_pad)
		bp[i] = i < n ? (bias ? bias[i] >> 1 : 0) : kI8qPadBp;
}
// out[i] = in[i] ^ x (bytes, 4 at a time) for i < n_words; zero for n_words <= i < n_words_pad
__global__ __launch_bounds__(kBlock) void i8q_copy_xor_kernel(const uint32_t* in, size_t n_words,
                                                              size_t n_words_pad, uint32_t x,
                                                              uint32_t* out) {
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_words_pad;
	     i += (size_t)gridDim.x * kBlock)
		out[i] = i < n_words ? in[i] ^ x : 0u;
}

// rows of d bytes -> rows of dq >= d bytes: out[r][c] = in[r][c] ^ x for r < n_rows, c < d; else 0
// (4 bytes at a time; d, dq multiples of 4)
__global__ __launch_bounds__(kBlock) void i8q_pad_rows_kernel(const uint32_t* in, size_t n_rows, uint32_t dw,
                                                              uint32_t dqw, size_t n_rows_pad, uint32_t x,
                                                              uint32_t* out) {
	const size_t total = n_rows_pad * dqw;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
		const size_t r = i / dqw;
		const uint32_t c = (uint32_t)(i - r * dqw);
		out[i] = (r < n_rows && c < dw) ? in[r * dw + c] ^ x : 0u;
	}
}

__device__ inline int max3i(int a, int b, int c) { return max(max(a, b), c); }

// Geometry by dimension.  d = 128 / 256: as scan_gemm_f16.hpp (4 waves x 64 queries, two
// workgroups per CU, 3 tile buffers).  d = 768 (BASELINE C5): a wave's fragments of 32 queries
// are already 96 VGPRs, so 8 waves x 32 queries share each staged tile (one workgroup per CU --
// halving the queries per tile would double the L2->LDS traffic that bounds this shape), 2 tile
// buffers of 48 KB, and the 16-byte chunks of a row are assigned to (k-step, lane half) in
// natural order (2s + h), which needs 8 fragment-address registers instead of 24.
// d = 1024 is the PHYSICAL row length of d = 832 / 960 indexes: their 832- / 960-byte rows (64 /
// 192 mod 256) have no conflict-free XOR swizzle, so the engine's padded copy stores them in
// 1024 bytes (zeros in the mapped int8 domain add nothing to q.b); two 64 KB tile buffers.
template <int D> struct I8qGeom {
	static constexpr int THREADS = D >= 768 ? 512 : 256;
	static constexpr int WAVES = THREADS / 64;
	static constexpr int TQW = D >= 768 ? 1 : 2;      // 32-query MFMA tiles per wave
	static constexpr int WGQ = WAVES * 32 * TQW;      // queries per workgroup
	static constexpr int NBUF = D >= 768 ? 2 : 3;
	// d = 128: the query fragments are only 32 VGPRs; with the accumulator start values read from
	// LDS at every step instead of held in 32 more, the scan fits 168 VGPRs and THREE workgroups
	// share a CU (3 waves per SIMD to cover each other's barriers and flushes)
	static constexpr bool TH_LDS = D == 128;
	static constexpr int QCAP = D == 1024 ? 24 : (D == 768 ? 56 : (TH_LDS ? 56 : 88));  // queue entries per wave
	static constexpr bool NATURAL = D >= 768;
	static constexpr int WG_PER_CU = TH_LDS ? 3 : 512 / THREADS;
};
static_assert(I8qGeom<128>::WGQ == kF16TQ && I8qGeom<768>::WGQ == kF16TQ, "one query-tile size");

template <int D> constexpr int gemm_i8q_lds_bytes() {
	using G = I8qGeom<D>;
	// tiles + per-wave bp slots + per-wave queues + accumulator start values + one fill word per wave
	return G::NBUF * (kF16TB * D + G::WAVES * 256) + G::WAVES * G::QCAP * kF16EntryBytes + G::WGQ * 4 + 64 +
	       (G::TH_LDS ? G::WAVES * 2 * G::TQW * 16 * 4 : 0);
}
static_assert(gemm_i8q_lds_bytes<768>() <= 160 * 1024 && gemm_i8q_lds_bytes<1024>() <= 160 * 1024 &&
                  gemm_i8q_lds_bytes<256>() * 2 <= 160 * 1024 &&
                  gemm_i8q_lds_bytes<128>() * 3 <= 160 * 1024,
              "LDS budget per CU");

// (Rounds 1-2 ran this filter -- full scan and sampled pass -- on v_mfma_i32_32x32x32_i8 in a kernel of this
// file, scan_gemm_i8q_kernel<D, L2FORM, SAMPLE, DR>; since round 3 every 8-bit stream is one of the 16 x 16 x 64
// kernels scan_gemm_i8w (d = 128 / 256) and scan_gemm_i8x (d = 768 and the 1024-byte slots of d = 832 / 960),
// each with a SAMPLE instance.  What stays here is what they share: the g-domain arithmetic above, parameters,
// geometry and LDS budget, the index / query conversions, the thresholds from the sampled pass.)

// thp[q] = -g_k, g_k the k-th largest of the n_vals class maxima of the SAMPLE pass (fewer than k
// real values: pass everything); also zeroes the query's list counter.  One wave per query.
struct SampleTauI8Params {
	const int* vals;    // [m][n_vals]
	uint32_t n_vals;
	uint32_t m;
	uint32_t k;
	int* thp;           // [m]
	uint32_t* cand_cnt; // [m] <- 0
};
template <int PER>
__global__ __launch_bounds__(kBlock) void sample_tau_i8_kernel(SampleTauI8Params p) {
	const int lane = threadIdx.x & 63;
	const uint32_t qi = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (qi >= p.m)
		return;  // (whole wave)
	const int* v = p.vals + (size_t)qi * p.n_vals;
	// keys ordered like g: (g ^ sign) << 32 | unique index; 0 = empty / taken
	uint64_t keys[PER];
#pragma unroll
	for (int j = 0; j < PER; ++j) {
		const uint32_t i = j * 64 + lane;
		keys[j] = i < p.n_vals ? ((uint64_t)((uint32_t)v[i] ^ 0x80000000u) << 32) | (0xFFFFFFFFu - i) : 0ull;
	}
	uint64_t kth = 0;
	if (p.k > 24) {
		uint32_t ord[PER];
#pragma unroll
		for (int j = 0; j < PER; ++j)
			ord[j] = (uint32_t)(keys[j] >> 32);
		kth = (uint64_t)wave_kth_largest_u32<PER>(ord, p.k) << 32;
	} else
	for (uint32_t it = 0; it < p.k; ++it) {
		uint64_t best = 0;
#pragma unroll
		for (int j = 0; j < PER; ++j)
			best = keys[j] > best ? keys[j] : best;
		for (int off = 32; off > 0; off >>= 1) {
			const uint64_t o = __shfl_xor(best, off);
			best = o > best ? o : best;
		}
		kth = best;
#pragma unroll
		for (int j = 0; j < PER; ++j)
			keys[j] = keys[j] == best ? 0ull : keys[j];
	}
	if (lane == 0) {
		const int g = (int)((uint32_t)(kth >> 32) ^ 0x80000000u);
		// no k-th value, or a maximum that only padding produced: keep every row
		p.thp[qi] = (kth == 0 || g < -(1 << 29)) ? (1 << 30) : -g;
		p.cand_cnt[qi] = 0;
	}
}

}  // namespace expann
