// scan_gemm_i8q.hpp -- the 8-bit GEMM-form filter in the geometry of scan_gemm_f16.hpp
// (256-thread workgroups, two per CU; 64 queries x 64 rows = 2x2 MFMA tiles per wave and step;
// LDS-DMA staging with counted vmcnt waits; per-wave hit queues in LDS; one sampled pass for the
// threshold), on the int8 matrix cores (v_mfma_i32_32x32x32_i8), for d = 128, 256 and 768
// (I8qGeom below: d = 768 trades the two-workgroups-per-CU layout for 8 waves on one tile).
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

// DR = bytes of a row that hold data (d = 832 / 960 rows live in 1024-byte slots, zero-padded on both
// sides of the product): the k-steps behind them multiply zeros and are left out -- 26 / 30 of 32
// k-steps, and as many query fragments less in registers (at D = DR = 1024 the kernel spilled)
template <int D, bool L2FORM, bool SAMPLE, int DR = D>
__global__ __launch_bounds__(I8qGeom<D>::THREADS, (I8qGeom<D>::TH_LDS && !SAMPLE) ? 3 : 2) void
scan_gemm_i8q_kernel(GemmI8qParams p) {
	static_assert(D == 128 || D == 256 || D == 768 || D == 1024, "built for d = 128, 256, 768, 1024");
	static_assert(DR == D || (I8qGeom<D>::NATURAL && DR < D && DR % 64 == 0), "padded rows: natural chunk order");
	using G = I8qGeom<D>;
	constexpr int THREADS = G::THREADS, WAVES = G::WAVES, TQW = G::TQW, WGQ = G::WGQ, QCAP = G::QCAP;
	constexpr bool NATURAL = G::NATURAL;
	constexpr int ROWB = D;          // bytes per row
	constexpr int CH = ROWB / 16;    // 16-byte chunks per row
	constexpr int KS = D / 32;       // MFMA k-steps of a row slot
	constexpr int KR = (DR + 31) / 32;  // ... that hold data
	constexpr int TILE_BYTES = kF16TB * ROWB;
	constexpr int RPB = (ROWB < 256) ? 256 / ROWB : 1;
	constexpr int SWM = (CH < 16 ? CH : 16) - 1;
	constexpr int NBUF = G::NBUF, PF = NBUF - 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int h = lane >> 5, r31 = lane & 31;
	uint32_t qtile = blockIdx.x % p.n_qtiles;
	uint32_t chunk = blockIdx.x / p.n_qtiles;
	if (p.xcd_map) {
		const uint32_t j = blockIdx.x >> 3;
		qtile = j % p.n_qtiles;
		chunk = (blockIdx.x & 7) + 8 * (j / p.n_qtiles);
	}
	const uint32_t wg_q0 = qtile * WGQ;
	const uint32_t q0 = wg_q0 + wave * 32 * TQW;

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	if (t0 >= t1)
		return;  // (whole workgroup)

	unsigned char* const bn_slots = smem + NBUF * TILE_BYTES;
	struct QEntry {
		int acc[16];
		int bp;
		uint32_t row;
		uint32_t qrow0;
		uint32_t pad;
	};
	static_assert(sizeof(QEntry) == kF16EntryBytes, "queue entry size");
	QEntry* const queue = reinterpret_cast<QEntry*>(bn_slots + NBUF * WAVES * 256) + wave * QCAP;
	int* const thq =
	    reinterpret_cast<int*>(bn_slots + NBUF * WAVES * 256 + WAVES * QCAP * kF16EntryBytes);
	uint32_t* const fills = reinterpret_cast<uint32_t*>(thq + WGQ);
	constexpr bool THL = G::TH_LDS && !SAMPLE;
	// (THL) accumulator start values by (wave, lane half): 16 per query tile, read back as broadcasts
	int* const thl = reinterpret_cast<int*>(fills + 16) + (wave * 2 + h) * (TQW * 16);

	constexpr int kNever = -2147483647 - 1;
	// query fragments; lane half h of k-step s holds chunk 2s + h (natural) or h*KS + s
	i32x4 a[TQW][KR];
#pragma unroll
	for (int tq = 0; tq < TQW; ++tq) {
		uint32_t qi = q0 + tq * 32 + r31;
		if (qi >= p.m)
			qi = p.m - 1;
		const i32x4* src =
		    reinterpret_cast<const i32x4*>((const unsigned char*)p.queries + (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KR; ++s)
			a[tq][s] = src[NATURAL ? 2 * s + h : h * KS + s];
	}
	// accumulator start values -g_k (SAMPLE: zero starts; th holds the running class maxima)
	i32x16 th[TQW];
#pragma unroll
	for (int tq = 0; tq < TQW; ++tq)
#pragma unroll
		for (int reg = 0; reg < 16; ++reg) {
			const uint32_t qi = q0 + tq * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
			// a padded query slot starts at INT_MIN/2: dot + that never reaches a bp >= 0
			th[tq][reg] = SAMPLE ? kNever : (qi < p.m ? p.thp[qi] : kNever / 2);
		}
	if (!SAMPLE && tid < WGQ)
		thq[tid] = wg_q0 + tid < p.m ? p.thp[wg_q0 + tid] : kNever / 2;
	if (THL && r31 == 0) {
#pragma unroll
		for (int tq = 0; tq < TQW; ++tq)
#pragma unroll
			for (int reg = 0; reg < 16; ++reg)
				thl[tq * 16 + reg] = th[tq][reg];
	}
#pragma unroll
	for (int tq = 0; tq < TQW; ++tq) {
#pragma unroll
		for (int s = 0; s < KR; ++s)
			asm volatile("" : "+v"(a[tq][s]));
		if (!THL)
			asm volatile("" : "+v"(th[tq]));
	}
	// LDS offsets of this lane's fragment chunks (row r31 of the first column tile; the second
	// is 32 rows further, same swizzle term).  Natural order: chunk 2s + h = 16 (s >> 3) +
	// (2 (s & 7) + h), and the XOR with the row's swizzle (< 16) only touches the low part.
	static_assert((32 / RPB) % (SWM + 1) == 0, "swizzle must repeat every 32 rows");
	constexpr int NA = NATURAL ? 8 : KS;
	uint32_t aoff[NA];
#pragma unroll
	for (int j = 0; j < NA; ++j)
		aoff[j] = r31 * ROWB + ((((NATURAL ? 2 * j + h : h * KS + j)) ^ ((r31 / RPB) & SWM)) * 16);

	auto tile_row0 = [&](uint32_t t) -> uint32_t {
		return ((t / p.tile_run) * (p.tile_stride * p.tile_run) + (t % p.tile_run)) * kF16TB;
	};
	// staging: piece i of a thread is slot S = i*THREADS + tid of the tile (16 bytes each, LDS
	// order = row-major physical chunks); its source is the logical chunk pc ^ swizzle(row)
	constexpr int N_STAGE = kF16TB * CH / THREADS;
	static_assert(kF16TB * CH % THREADS == 0, "whole staging rounds");
	constexpr int LOADS = N_STAGE + 1;
	uint32_t soff[N_STAGE];
#pragma unroll
	for (int i = 0; i < N_STAGE; ++i) {
		const uint32_t S = i * THREADS + tid;
		const uint32_t r = S / CH, pc = S % CH;
		soff[i] = r * ROWB + ((pc ^ ((r / RPB) & SWM)) * 16);
	}
	auto stage_piece = [&](const unsigned char* tb, uint32_t row0, int buf, int i) {
		if (i < N_STAGE) {
			unsigned char* dst0 = smem + buf * TILE_BYTES + wave * 64 * 16;
			__builtin_amdgcn_global_load_lds(
			    (const __attribute__((address_space(1))) void*)(tb + soff[i < N_STAGE ? i : 0]),
			    (__attribute__((address_space(3))) void*)(dst0 + i * THREADS * 16), 16, 0, 0);
		} else {
			__builtin_amdgcn_global_load_lds(
			    (const __attribute__((address_space(1))) void*)(p.bp + row0 + lane),
			    (__attribute__((address_space(3))) void*)(bn_slots + (buf * WAVES + wave) * 256), 4, 0, 0);
		}
	};
	auto stage_src = [&](uint32_t t, uint32_t& row0) -> const unsigned char* {
		if (t > t1 - 1)
			t = t1 - 1;
		row0 = tile_row0(t);
		return (const unsigned char*)p.base + (size_t)row0 * ROWB;
	};
	auto stage = [&](uint32_t t, int buf) {
		uint32_t row0;
		const unsigned char* tb = stage_src(t, row0);
#pragma unroll
		for (int i = 0; i < LOADS; ++i)
			stage_piece(tb, row0, buf, i);
	};
	auto read_bp = [&](int (&bv)[2], int buf) {
		const int* slot = reinterpret_cast<const int*>(bn_slots + (buf * WAVES + wave) * 256);
		bv[0] = slot[r31];
		bv[1] = slot[32 + r31];
	};

	uint32_t wfill = 0;  // wave-uniform
	auto push_global = [&](uint32_t qi, int acc, uint32_t row) {
		// exact integer score from the accumulator: dot = acc - thp[q]
		const int dot = acc - thq[(qi - wg_q0) & (WGQ - 1)];
		const int score = L2FORM ? p.bias[row] - 2 * dot + p.qself[qi] : -dot;
		const uint32_t slot = atomicAdd(&p.cand_cnt[qi], 1u);
		if (slot < p.cap)
			p.cand[(size_t)qi * p.cap + slot] = make_key((float)score, row);
	};
	auto flush_own = [&]() {
		const uint32_t n = wfill < (uint32_t)QCAP ? wfill : (uint32_t)QCAP;
		constexpr int R = 8;
		for (uint32_t base = 0; base < n * 16; base += 64 * R) {
			bool hit[R];
			uint32_t qi[R], slot[R], row[R];
			int dot[R];
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const uint32_t i = base + j * 64 + lane;
				const QEntry& e = queue[i < n * 16 ? i >> 4 : 0];
				const uint32_t reg = i & 15;
				const int c = e.acc[reg];
				row[j] = e.row;
				hit[j] = i < n * 16 && c >= e.bp && row[j] < p.n_rows;
				qi[j] = e.qrow0 + (reg & 3) + 8 * (reg >> 2);
				dot[j] = c - thq[(qi[j] - wg_q0) & (WGQ - 1)];
			}
			int score[R];
#pragma unroll
			for (int j = 0; j < R; ++j)
				score[j] = !hit[j] ? 0 : (L2FORM ? p.bias[row[j]] - 2 * dot[j] + p.qself[qi[j]] : -dot[j]);
#pragma unroll
			for (int j = 0; j < R; ++j)
				slot[j] = hit[j] ? atomicAdd(&p.cand_cnt[qi[j]], 1u) : 0xFFFFFFFFu;
#pragma unroll
			for (int j = 0; j < R; ++j)
				if (hit[j] && slot[j] < p.cap)
					p.cand[(size_t)qi[j] * p.cap + slot[j]] = make_key((float)score[j], row[j]);
		}
		wfill = 0;
	};
	auto epilogue = [&](const i32x16 (&accs)[TQW][2], uint32_t row0, const int (&bv)[2]) {
#pragma unroll
		for (int tq = 0; tq < TQW; ++tq)
#pragma unroll
			for (int tc = 0; tc < 2; ++tc) {
				const i32x16& c = accs[tq][tc];
				const int bn = bv[tc];
				int m0 = max3i(c[0], c[1], c[2]);
				int m1 = max3i(c[3], c[4], c[5]);
				int m2 = max3i(c[6], c[7], c[8]);
				int m3 = max3i(c[9], c[10], c[11]);
				int m4 = max3i(c[12], c[13], c[14]);
				m0 = max3i(m0, m1, c[15]);
				m2 = max3i(m2, m3, m4);
				m0 = max(m0, m2);
				const unsigned long long mask = __builtin_amdgcn_ballot_w64(m0 >= bn);
				if (mask != 0) {
					uint32_t qrow0 = q0 + tq * 32 + 4 * h;  // rare path: arithmetic stays in here
					asm volatile("" : "+v"(qrow0));
					const uint32_t brow = row0 + tc * 32 + r31;
					const uint32_t slot =
					    wfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
					                                      __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
					if (m0 >= bn) {
						if (slot < (uint32_t)QCAP) {
							QEntry& e = queue[slot];
#pragma unroll
							for (int reg = 0; reg < 16; ++reg)
								e.acc[reg] = c[reg];
							e.bp = bn;
							e.row = brow;
							e.qrow0 = qrow0;
						} else if (brow < p.n_rows) {  // queue full: straight to the lists
#pragma unroll 1
							for (int reg = 0; reg < 16; ++reg) {
								int cr = c[0];
#pragma unroll
								for (int j = 1; j < 16; ++j)
									cr = reg == j ? c[j] : cr;
								if (cr >= bn)
									push_global(qrow0 + (reg & 3) + 8 * (reg >> 2), cr, brow);
							}
						}
					}
					wfill += (uint32_t)__builtin_popcountll(mask);
				}
			}
	};

#pragma unroll
	for (int i = 0; i < PF; ++i)
		stage(t0 + i, i);
	wait_vm_then_barrier<(PF - 1) * LOADS>();

	i32x16 acc[TQW][2];
	i32x16 zero16;
#pragma unroll
	for (int e = 0; e < 16; ++e)
		zero16[e] = 0;
	int bv[2];
	int buf = 0, pbuf = PF;
	uint32_t since_look = 0;
	for (uint32_t t = t0; t < t1; ++t) {
		const uint32_t boff = (uint32_t)buf * TILE_BYTES;
		auto frag = [&](int tc, int s) -> i32x4 {
			const uint32_t o = NATURAL ? aoff[s & 7] + (s >> 3) * 256 : aoff[NATURAL ? 0 : s];
			return *reinterpret_cast<const i32x4*>(smem + (boff + o) + tc * 32 * ROWB);
		};
		uint32_t srow0;
		const unsigned char* stb = stage_src(t + PF, srow0);
		i32x4 fb[KR][2];
		fb[0][0] = frag(0, 0);
		fb[0][1] = frag(1, 0);
		fb[1][0] = frag(0, 1);
		fb[1][1] = frag(1, 1);
		read_bp(bv, buf);
		__builtin_amdgcn_s_setprio(1);  // the MFMA phase outranks the other workgroup's epilogue / flush
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KR; ++s) {
			if (s + 2 < KR) {
				fb[s + 2][0] = frag(0, s + 2);
				fb[s + 2][1] = frag(1, s + 2);
			}
#pragma unroll
			for (int tq = 0; tq < TQW; ++tq) {
				if (THL && s == 0) {
					// start values straight from LDS into the first accumulator, which then seeds
					// both column tiles (second one first: the first is overwritten in place)
					acc[tq][0] = *reinterpret_cast<const i32x16*>(thl + tq * 16);
					acc[tq][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[tq][s], fb[s][1], acc[tq][0], 0, 0, 0);
					acc[tq][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[tq][s], fb[s][0], acc[tq][0], 0, 0, 0);
					continue;
				}
				acc[tq][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(
				    a[tq][s], fb[s][0], s == 0 ? (SAMPLE ? zero16 : th[tq]) : acc[tq][0], 0, 0, 0);
				acc[tq][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(
				    a[tq][s], fb[s][1], s == 0 ? (SAMPLE ? zero16 : th[tq]) : acc[tq][1], 0, 0, 0);
			}
			// the stage loads of tile t+PF go out between the MFMAs
			constexpr int PER = (LOADS + KR - 1) / KR;
#pragma unroll
			for (int j = 0; j < PER; ++j)
				if (s * PER + j < LOADS)
					stage_piece(stb, srow0, pbuf, s * PER + j);
			__builtin_amdgcn_sched_barrier(0);
		}
		__builtin_amdgcn_s_setprio(0);
		if (SAMPLE) {
#pragma unroll
			for (int tq = 0; tq < TQW; ++tq)
#pragma unroll
				for (int tc = 0; tc < 2; ++tc)
#pragma unroll
					for (int reg = 0; reg < 16; ++reg)
						th[tq][reg] = max(th[tq][reg], acc[tq][tc][reg] - bv[tc]);
			wait_vm_then_barrier<(PF - 1) * LOADS>();
		} else {
			epilogue(acc, tile_row0(t), bv);
			if (wfill >= (uint32_t)QCAP * 3 / 4)  // about to overflow: empty it at once
				flush_own();
			const bool look = ++since_look == kF16FlushEvery;
			if (look && lane == 0)
				fills[wave] = wfill;
			wait_vm_then_barrier<(PF - 1) * LOADS>();
			if (look) {
				since_look = 0;
				const uint32_t f = fills[lane & (WAVES - 1)];
				if (__builtin_amdgcn_ballot_w64(f >= (uint32_t)QCAP / 2) != 0)
					flush_own();
			}
		}
		pbuf = buf;
		buf = buf + 1 == NBUF ? 0 : buf + 1;
	}
	if (SAMPLE) {
#pragma unroll
		for (int tq = 0; tq < TQW; ++tq)
#pragma unroll
			for (int reg = 0; reg < 16; ++reg) {
				const uint32_t qi = q0 + tq * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
				if (qi < p.m)
					p.sample_out[((size_t)qi * p.n_chunks + chunk) * 32 + r31] = th[tq][reg];
			}
	} else {
		flush_own();
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-staged tail tiles: LDS must outlive them
}

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
