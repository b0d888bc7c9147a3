// scan_gemm_f16x.hpp -- the fp16 candidate filter of scan_gemm_f16.hpp (same slack analysis, same
// parameters, same LDS tile image and staging, same candidate queues) on v_mfma_f32_16x16x32_f16,
// with the epilogue folded into the MFMA stream.  d = 64 / 128 (the 4-wave, two-workgroups-per-CU
// geometry).
//
// Why another form of the same kernel (round 2, measured on MI355X, C2):
//   * the chip lowers its clock under a dense MFMA stream and holds a higher one on the 16x16x32
//     shape than on 32x32x16 at equal cycles per flop (MI355X_MICROARCH.md, DVFS give-back item 7);
//   * with 16 x 16 tiles the row (base vector) sits on lane & 15 and a lane's 4 accumulators of a
//     tile are 4 QUERIES, so the 16 accumulators a lane holds for one 16-row tile column -- 4 query
//     tiles x 4 -- share ONE row term bn': one max tree and one compare per tile column instead
//     of one per 32 x 32 tile, and no branch inside the MFMA stream: the tile-column loop is
//     outermost (k-steps inside), the max tree of column tc-1 is scheduled between the MFMAs of
//     column tc, and the (rare) queue push happens once per step behind a single wave-uniform test.
//
// Layout of one wave's step: 64 queries x 64 rows = 4 x 4 tiles of 16 x 16; A = queries
// (a[tq][s]: lane l holds query l & 15 of tile tq, 16-byte chunk 4 s + (l >> 4) of its row), B =
// base rows from LDS (lane l: row l & 15 of tile column tc, same chunk), C: lane l, register r =
// query 4 (l >> 4) + r of tile tq against row l & 15 of column tc.
#pragma once
#include <type_traits>

#include "scan_gemm_f16.hpp"

namespace expann {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// DBG = 1 compiles the run-time ablation switches of p.debug in (bench only: 1 no barrier, 2 no
// staging, 4 no epilogue, 8 no candidate path, 32 no LDS fragment reads, 64 flushes drop their hits);
// the production instance has none of their branches, so a step's MFMA stream is ONE basic block.
// LDS of one workgroup: tiles + per-wave bn' slots + per-wave queues + theta' + queue fills
// d = 64: two k-steps per column are too little MFMA per step for two waves per SIMD (measured: 4 %
// slower than scan_gemm_f16_kernel<64>), so -- as scan_gemm_i8w_kernel<128>, the same byte geometry --
// the form stays under 168 VGPRs there and runs THREE workgroups per CU (smaller hit queues to fit)
template <int D> constexpr int f16x_qcap() { return D == 64 ? 64 : kF16WaveQueue; }
template <int D> constexpr int f16x_wg_per_cu() { return D == 64 ? 3 : 2; }
template <int D> constexpr int gemm_f16x_lds_bytes() {
	return kF16Bufs * (kF16TB * D * 2 + kF16Waves * 256) + kF16Waves * f16x_qcap<D>() * kF16EntryBytes + kF16TQ * 4 + 64;
}
static_assert(gemm_f16x_lds_bytes<64>() * 3 <= 160 * 1024, "three workgroups per CU at d = 64");
static_assert(gemm_f16x_lds_bytes<128>() == gemm_f16_lds_bytes<128>(), "same LDS map as scan_gemm_f16_kernel<128>");

template <int D, bool SAMPLE, int DBG = 0>
__global__ __launch_bounds__(kF16Threads, f16x_wg_per_cu<D>()) void scan_gemm_f16x_kernel(GemmF16Params p) {
	const uint32_t dbg = DBG ? p.debug : 0u;
	static_assert(D == 64 || D == 128, "built for d = 64, 128");
	using G = F16Geom<128>;  // (geometry of the 4-wave form; d = 64 shares it here: no TH_LDS variant)
	constexpr int THREADS = kF16Threads, WAVES = kF16Waves, WGQ = kF16TQ, QCAP = f16x_qcap<D>();
	constexpr int ROWB = D * 2, CH = ROWB / 16;
	constexpr int KS = D / 32;  // MFMA k-steps of 32
#ifndef EXPANN_F16X_SPLIT
#define EXPANN_F16X_SPLIT 1
#endif
	constexpr int SPLIT = EXPANN_F16X_SPLIT;  // k-steps of a column issued before the next column's fragment reads
	constexpr int TILE_BYTES = kF16TB * ROWB;
	constexpr int RPB = (ROWB < 256) ? 256 / ROWB : 1;
	constexpr int SWM = (CH < 16 ? CH : 16) - 1;
	constexpr int NBUF = kF16Bufs, PF = NBUF - 1;
	static_assert(G::NBUF == NBUF && G::WGQ == WGQ && (D == 64 || G::QCAP == QCAP), "shares gemm_f16_lds_bytes<128>()");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l15 = lane & 15, lq = lane >> 4;
	// One work item = what a workgroup of the plain launch does: (query tile, row chunk) number `bid` of
	// n_items.  Plain launch: bid = blockIdx.x.  Persistent launch (p.work_ctr != nullptr, round 3): the
	// grid is what is resident at once and every workgroup PULLS items -- from the counter of its own XCD
	// first (item ids = XCD mod 8, as the hardware deals plain blocks: the row chunks {x, x + 8, ..} of a
	// query tile stay on one L2), from the other XCDs' counters when its own is exhausted.  Workgroups of
	// equal work finish 2-4 % apart by XCD (DVFS), so the dealt launch ended on its slowest XCD.
	const uint32_t n_items = p.work_ctr ? p.n_items : gridDim.x;
	auto item_body = [&](uint32_t bid) __attribute__((always_inline)) {
	uint32_t qtile = bid % p.n_qtiles;
	uint32_t chunk = bid / p.n_qtiles;
	if (p.xcd_map) {
		const uint32_t j = bid >> 3;
		qtile = j % p.n_qtiles;
		chunk = (bid & 7) + 8 * (j / p.n_qtiles);
	}
	const uint32_t wg_q0 = qtile * WGQ;
	const uint32_t q0 = wg_q0 + wave * 64;
	// 16-query tiles of this wave that hold real queries (4 everywhere but in the batch's last query
	// tile: m = 10 000 leaves 16 queries for it, i.e. one tile of wave 0 and none of waves 1-3)
	const int n_tq = __builtin_amdgcn_readfirstlane(q0 >= p.m ? 0 : (int)((p.m - q0 + 15) / 16 < 4 ? (p.m - q0 + 15) / 16 : 4));

	uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (p.tiles_small && chunk >= p.n_big) {  // the launch's tail: smaller chunks (GemmF16Params::n_big)
		t0 = p.n_big * p.tiles_per_block + (chunk - p.n_big) * p.tiles_small;
		t1 = t0 + p.tiles_small;
	}
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	uint32_t* const my_log_cnt = p.log_cnt + (size_t)bid * WAVES + wave;
	if (t0 >= t1) {
		if (SAMPLE) {  // (the host plans no empty chunk; if one appears its class maxima are "no row")
			for (uint32_t i = lane; i < 64 * 32; i += 64)
				if (q0 + (i >> 5) < p.m)
					p.sample_out[((size_t)(q0 + (i >> 5)) * p.n_chunks + chunk) * 32 + (i & 31)] = -__builtin_inff();
		} else if (lane == 0) {
			*my_log_cnt = 0;
		}
		return;
	}
	const unsigned long long clk0 = p.clk ? clock64() : 0, wall0 = p.clk ? wall_clock64() : 0;

	// LDS map: as scan_gemm_f16_kernel<128> (tiles, per-wave bn' slots, per-wave queues, theta', fills)
	unsigned char* const bn_slots = smem + NBUF * TILE_BYTES;
	struct QEntry {
		float acc[16];   // value i = query tile i >> 2, register i & 3
		float bn;
		uint32_t row;
		uint32_t qrow0;  // query of value 0; value i is + 16 (i >> 2) + (i & 3)
		uint32_t pad;
	};
	static_assert(sizeof(QEntry) == kF16EntryBytes, "queue entry size");
	QEntry* const queue = reinterpret_cast<QEntry*>(bn_slots + NBUF * WAVES * 256) + wave * QCAP;
	// (behind the queues: WGQ words that held theta' for the flush's keys until round 3 -- the map is shared with f16y)
	uint32_t* const fills = reinterpret_cast<uint32_t*>(bn_slots + NBUF * WAVES * 256 + WAVES * QCAP * kF16EntryBytes) + WGQ;

	f16x8 a[4][KS];
#pragma unroll
	for (int tq = 0; tq < 4; ++tq) {
		uint32_t qi = q0 + tq * 16 + l15;
		if (qi >= p.m)
			qi = p.m - 1;
		const f16x8* src = reinterpret_cast<const f16x8*>((const unsigned char*)p.queries_f16 + (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			a[tq][s] = src[4 * s + lq];
	}
	// accumulator start values: theta' of the query of each accumulator register
	// (SAMPLE: accumulators start at zero, th holds the running class maxima of g)
	f32x4 th[4];
#pragma unroll
	for (int tq = 0; tq < 4; ++tq)
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
			th[tq][r] = (qi < p.m && !SAMPLE) ? p.theta[qi] : -__builtin_inff();
		}
#pragma unroll
	for (int tq = 0; tq < 4; ++tq) {
#pragma unroll
		for (int s = 0; s < KS; ++s)
			asm volatile("" : "+v"(a[tq][s]));
		asm volatile("" : "+v"(th[tq]));
	}
	// per-lane LDS offset of k-step s in tile column 0 (row l15); column tc is 16 rows further,
	// where the swizzle term is the same
	uint32_t aoff[KS];
#pragma unroll
	for (int s = 0; s < KS; ++s)
		aoff[s] = l15 * ROWB + (((4 * s + lq) ^ ((l15 / RPB) & SWM)) * 16);

	auto tile_row0 = [&](uint32_t t) -> uint32_t {
		return ((t / p.tile_run) * (p.tile_stride * p.tile_run) + (t % p.tile_run)) * kF16TB;
	};
	// staging: identical to scan_gemm_f16_kernel (N_STAGE pieces of the tile + the bn' piece per wave)
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
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + soff[i < N_STAGE ? i : 0]),
			                                 (__attribute__((address_space(3))) void*)(dst0 + i * THREADS * 16), 16, 0, 0);
		} else {
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bnorm + row0 + lane),
			                                 (__attribute__((address_space(3))) void*)(bn_slots + (buf * WAVES + wave) * 256),
			                                 4, 0, 0);
		}
	};
	auto stage_src = [&](uint32_t t, uint32_t& row0) -> const unsigned char* {
		if (t > t1 - 1)
			t = t1 - 1;
		row0 = tile_row0(t);
		return (const unsigned char*)p.base_f16 + (size_t)row0 * ROWB;
	};
	auto stage = [&](uint32_t t, int buf) {
		uint32_t row0;
		const unsigned char* tb = stage_src(t, row0);
#pragma unroll
		for (int i = 0; i < LOADS; ++i)
			stage_piece(tb, row0, buf, i);
	};

	// candidates: the hit LANES of a step append their 16 raw accumulators (+ bn', row, first query)
	// to this wave's queue in LDS, as scan_gemm_f16_kernel; when the queues are emptied, 16 lanes per
	// entry redo the compare and the hits go -- ballot-compacted, plain 16-byte stores at a
	// wave-uniform position -- to this wave's LOG in global memory (GemmF16Params::log): nothing the
	// wave has to wait for, and no atomic inside the MFMA kernel (round 1's per-hit atomicAdd on the
	// query's counter cost a round trip to L2 per flush and 98 MB of write traffic per launch).
	uint32_t wfill = 0;   // wave-uniform: entries in the LDS queue
	uint32_t glog_n = 0;  // wave-uniform: entries in this wave's global log
	uint4* const my_log = p.log + ((size_t)bid * WAVES + wave) * p.log_cap;
	auto flush_own = [&]() {
		const uint32_t n = wfill < (uint32_t)QCAP ? wfill : (uint32_t)QCAP;
		// 16 lanes per entry, TWO rounds of 4 entries in flight (their LDS reads share one wait); a log entry is
		// {bn' - acc, row, query}: gather_logs_kernel adds theta' and makes the key (no second LDS round trip here)
		for (uint32_t base = 0; base < n * 16; base += 128) {
			bool hit[2];
			float dv[2];
			uint32_t row[2], qi[2];
			unsigned long long mask[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const uint32_t i = base + 64 * u + lane;
				const QEntry& e = queue[i < n * 16 ? i >> 4 : 0];
				const uint32_t v = i & 15;
				const float c = e.acc[v], bn = e.bn;
				hit[u] = i < n * 16 && c >= bn;
				qi[u] = e.qrow0 + 16 * (v >> 2) + (v & 3);
				row[u] = e.row;
				dv[u] = bn - c;
			}
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				mask[u] = __builtin_amdgcn_ballot_w64(hit[u]);
				if (mask[u] == 0 || (dbg & 64))
					continue;
				const uint32_t pos = glog_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[u] >> 32),
				                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask[u], 0u));
				if (hit[u] && pos < p.log_cap)
					my_log[pos] = make_uint4(__builtin_bit_cast(uint32_t, dv[u]), row[u], qi[u], 0u);
				glog_n += (uint32_t)__builtin_popcountll(mask[u]);
			}
		}
		wfill = 0;
	};
	// the rare path of a step: lanes whose tile column tc holds a hit append their 16 values
	auto push_hits = [&](const f32x4 (&acc)[4][4], int tc, unsigned long long mask, bool mine, float bn,
	                     uint32_t row0) {
		// dense hits (clustered rows, loose thresholds): make room first -- a column adds at most 64
		// entries and the queue holds more, so nothing is ever dropped here
		static_assert(QCAP >= 64, "a tile column's hits fit an empty queue");
		if (wfill + (uint32_t)__builtin_popcountll(mask) > (uint32_t)QCAP)
			flush_own();
		uint32_t qrow0 = q0 + 4 * lq;
		asm volatile("" : "+v"(qrow0));
		const uint32_t brow = row0 + tc * 16 + l15;
		const uint32_t slot = wfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
		                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		if (mine) {
			if (slot < (uint32_t)QCAP) {
				QEntry& e = queue[slot];
#pragma unroll
				for (int tq = 0; tq < 4; ++tq)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						e.acc[tq * 4 + r] = acc[tq][tc][r];
				e.bn = bn;
				e.row = brow;
				e.qrow0 = qrow0;
			}
		}
		wfill += (uint32_t)__builtin_popcountll(mask);
	};

	// ---- the pipeline ---------------------------------------------------------------------------
	// ONE workgroup barrier per step, in the MIDDLE of the step's MFMA stream (after tile column 1):
	//   * behind it tile t+1 has landed for every wave, so the fragments of its column 0 (and its
	//     row terms) are requested while columns 2-3 of tile t still multiply: a step never starts
	//     with an exposed LDS round trip (measured before: ~350-480 cycles per step and wave);
	//   * behind it every wave has left tile t-1, whose buffer then takes the stage loads of tile
	//     t+2 -- all issued in the second half of the step, so the wait in front of the barrier is a
	//     plain vmcnt(0): the only loads outstanding are tile t+1's, issued a whole step earlier;
	//   * a wave parked at the barrier leaves the matrix pipe to the other workgroup's wave on its
	//     SIMD, which is in the middle of an MFMA stream of its own, not at a step boundary.
	// SAMPLE (the threshold pass, 4.6 of DESIGN.md): the same stream without thresholds and candidate
	// path.  A lane's accumulators of one tile column belong to ONE row, so the row term enters as the
	// MFMA's C operand -- acc starts at -bn' -- and the epilogue is the running maximum of g = q16.b16 -
	// bn' per (query register, row class): class = row mod 32 = 16 (tc & 1) + lane & 15, the columns of
	// equal parity folded by one v_max3 (32 per step instead of the 32 x 32 form's subtract + max per
	// accumulator).  A padding row's bn' is NaN: its g is NaN and never wins a max.
	f32x4 smax[2][4];
#pragma unroll
	for (int par = 0; par < 2; ++par)
#pragma unroll
		for (int tq = 0; tq < 4; ++tq)
			smax[par][tq] = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
	stage(t0, 0);
	stage(t0 + 1, 1);
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tiles t0, t0+1 landed, fills visible

	f32x4 acc[4][4];
	int buf = 0;
	uint32_t since_look = 0;
	unsigned long long seg[4] = {0, 0, 0, 0}, ts = (DBG && p.clk) ? clock64() : 0;
	auto stamp = [&](int i) {
		if (DBG && p.clk) {
			const unsigned long long now = clock64();
			seg[i] += now - ts;
			ts = now;
		}
	};
	auto frag_at = [&](int b, int tc, int s) -> f16x8 {
		return *reinterpret_cast<const f16x8*>(smem + ((uint32_t)b * TILE_BYTES + aoff[s]) + tc * 16 * ROWB);
	};
	auto read_bn = [&](float (&bn)[4], int b) {
		const float* slot = reinterpret_cast<const float*>(bn_slots + (b * WAVES + wave) * 256);
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			bn[tc] = slot[tc * 16 + l15];
	};
	auto col_max = [&](int pc) -> float {
		float m0 = max3f(acc[0][pc][0], acc[0][pc][1], acc[0][pc][2]);
		float m1 = max3f(acc[0][pc][3], acc[1][pc][0], acc[1][pc][1]);
		float m2 = max3f(acc[1][pc][2], acc[1][pc][3], acc[2][pc][0]);
		float m3 = max3f(acc[2][pc][1], acc[2][pc][2], acc[2][pc][3]);
		float m4 = max3f(acc[3][pc][0], acc[3][pc][1], acc[3][pc][2]);
		m0 = max3f(m0, m1, acc[3][pc][3]);
		m2 = max3f(m2, m3, m4);
		return __builtin_fmaxf(m0, m2);
	};
	// The step loop, compiled twice: FULL (every 16-query tile of the wave holds queries: the stream
	// below is one basic block per step) and PARTIAL (the batch's last query tile: the MFMAs -- and, for
	// a wave without any query, the fragment reads -- of the empty tiles are skipped behind wave-uniform
	// branches; their accumulators stay at theta' = -inf, so the max tree and the candidate path need no
	// change; staging and barriers are the same in both, the waves of a workgroup may take either).
	auto run = [&](auto partial_tag) __attribute__((always_inline)) {
	constexpr bool PARTIAL = decltype(partial_tag)::value;
	float bnv[4];
	auto fold = [&](int par) {  // SAMPLE: class maxima of the two columns of this parity
#pragma unroll
		for (int tq = 0; tq < 4; ++tq)
#pragma unroll
			for (int r = 0; r < 4; ++r)
				smax[par][tq][r] = max3f(smax[par][tq][r], acc[tq][par][r], acc[tq][par + 2][r]);
	};
	// k-steps [s0, s1) of tile column tc.  A column is issued in two parts -- k-step 0, then the rest -- with
	// the NEXT column's fragment reads between them: the compiler puts an lgkmcnt(0) in front of the first
	// MFMA that follows LDS reads (it cannot see the waits of the barrier's inline asm), and with the reads
	// in front of the column that wait exposed their whole round trip twice per step.
	auto mfma_part = [&](int tc, const f16x8 (&f)[KS], int s0, int s1) {
		if (SAMPLE) {
			const float nb = -bnv[tc];
			const f32x4 c0 = {nb, nb, nb, nb};
#pragma unroll
			for (int s = s0; s < s1; ++s)
#pragma unroll
				for (int tq = 0; tq < 4; ++tq)
					acc[tq][tc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tq][s], f[s], s == 0 ? c0 : acc[tq][tc], 0, 0, 0);
			return;
		}
		if (!PARTIAL) {
#pragma unroll
			for (int s = s0; s < s1; ++s)
#pragma unroll
				for (int tq = 0; tq < 4; ++tq)
					acc[tq][tc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tq][s], f[s], s == 0 ? th[tq] : acc[tq][tc], 0, 0, 0);
			return;
		}
#pragma unroll
		for (int tq = 0; tq < 4; ++tq) {
			if (tq >= n_tq) {
				if (s0 == 0)
					acc[tq][tc] = th[tq];
				continue;
			}
#pragma unroll
			for (int s = s0; s < s1; ++s)
				acc[tq][tc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tq][s], f[s], s == 0 ? th[tq] : acc[tq][tc], 0, 0, 0);
		}
	};
	auto frag = [&](int b, int tc, int s) -> f16x8 {
		if (PARTIAL && n_tq == 0)
			return f16x8{};
		return frag_at(b, tc, s);
	};

	f16x8 fb[2][KS];  // fragments of the column being multiplied / the next one
#pragma unroll
	for (int s = 0; s < KS; ++s)
		fb[0][s] = frag(0, 0, s);
	read_bn(bnv, 0);
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
	for (uint32_t t = t0; t < t1; ++t) {
		const int nbuf = buf + 1 == NBUF ? 0 : buf + 1;   // tile t+1
		const int pbuf = buf == 0 ? NBUF - 1 : buf - 1;   // tile t-1 -> takes tile t+2
		float gmax[4];
		float bnn[4];
		__builtin_amdgcn_s_setprio(1);
		// column 0 (its fragments came in during the previous step), column 1
		mfma_part(0, fb[0], 0, SPLIT);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[1][s] = (dbg & 32) ? fb[0][0] : frag(buf, 1, s);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(0, fb[0], SPLIT, KS);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(1, fb[1], 0, SPLIT);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[0][s] = (dbg & 32) ? fb[1][0] : frag(buf, 2, s);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(1, fb[1], SPLIT, KS);
		if (!SAMPLE && !(dbg & 4)) {
			gmax[0] = col_max(0);
			asm volatile("" : "+v"(gmax[0]));  // (computed HERE: between column 1's MFMAs, not sunk behind the barrier)
		}
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_setprio(0);
		stamp(0);
		// ---- the step's barrier: tile t+1 landed, tile t-1 released ------------------------------
		const bool look = !SAMPLE && ++since_look == kF16FlushEvery;
		if (look && lane == 0)
			fills[wave] = wfill;
		if (!(dbg & 1)) {
			// (builtins, not inline asm: the compiler's own wait insertion then knows that every LDS read issued
			// so far has landed, and puts no lgkmcnt(0) behind the second half's first fragment reads)
			asm volatile("" ::: "memory");
			__builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
			__builtin_amdgcn_s_barrier();
			asm volatile("" ::: "memory");
		}
		stamp(1);
		if (look) {
			since_look = 0;
			const uint32_t f = fills[lane & (WAVES - 1)];
			if (__builtin_amdgcn_ballot_w64(f >= (uint32_t)QCAP / 2) != 0)
				flush_own();
		}
		// ---- second half: columns 2, 3; stage tile t+2; fetch column 0 of tile t+1 ---------------
		uint32_t srow0;
		const unsigned char* stb = stage_src(t + 2, srow0);
		__builtin_amdgcn_s_setprio(1);
		mfma_part(2, fb[0], 0, SPLIT);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[1][s] = (dbg & 32) ? fb[0][0] : frag(buf, 3, s);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(2, fb[0], SPLIT, KS);
		if (!(dbg & 2)) {
#pragma unroll
			for (int i = 0; i < (LOADS + 1) / 2; ++i)
				stage_piece(stb, srow0, pbuf, i);
		}
		if (!SAMPLE && !(dbg & 4))
			gmax[1] = col_max(1);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(3, fb[1], 0, SPLIT);
		__builtin_amdgcn_sched_barrier(0);
		// column 0 of tile t+1 and its row terms (clamped past the end: the re-staged last tile)
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[0][s] = (dbg & 32) ? fb[1][0] : frag(nbuf, 0, s);
		read_bn(bnn, nbuf);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(3, fb[1], SPLIT, KS);
		if (!(dbg & 2)) {
#pragma unroll
			for (int i = (LOADS + 1) / 2; i < LOADS; ++i)
				stage_piece(stb, srow0, pbuf, i);
		}
		if (SAMPLE)
			fold(0);
		else if (!(dbg & 4))
			gmax[2] = col_max(2);
		__builtin_amdgcn_s_setprio(0);
		stamp(2);
		if (SAMPLE) {
			fold(1);
		} else if (!(dbg & 4)) {
			gmax[3] = col_max(3);
			// one wave-uniform test per step; the queue push is the rare path
			const bool h0 = gmax[0] >= bnv[0], h1 = gmax[1] >= bnv[1], h2 = gmax[2] >= bnv[2],
			           h3 = gmax[3] >= bnv[3];
			if (__builtin_amdgcn_ballot_w64(h0 || h1 || h2 || h3) != 0 && !(dbg & 8)) {
				const uint32_t row0 = tile_row0(t);
				const unsigned long long k0 = __builtin_amdgcn_ballot_w64(h0), k1 = __builtin_amdgcn_ballot_w64(h1),
				                         k2 = __builtin_amdgcn_ballot_w64(h2), k3 = __builtin_amdgcn_ballot_w64(h3);
				if (k0) push_hits(acc, 0, k0, h0, bnv[0], row0);
				if (k1) push_hits(acc, 1, k1, h1, bnv[1], row0);
				if (k2) push_hits(acc, 2, k2, h2, bnv[2], row0);
				if (k3) push_hits(acc, 3, k3, h3, bnv[3], row0);
			}
		}
		if (!SAMPLE && wfill >= (uint32_t)QCAP * 3 / 4)
			flush_own();
		// (tile t+1's first fragments and row terms were requested under column 3: they have landed; saying so
		// here keeps the compiler from waiting for them -- and for whatever LDS read it has issued since -- in
		// the middle of the next step's first column)
		__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			bnv[tc] = bnn[tc];
		buf = nbuf;
		stamp(3);
	}
	};
	if (SAMPLE || n_tq == 4)
		run(std::false_type{});
	else
		run(std::true_type{});
	if (SAMPLE) {
#pragma unroll
		for (int par = 0; par < 2; ++par)
#pragma unroll
			for (int tq = 0; tq < 4; ++tq)
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
					if (qi < p.m)
						p.sample_out[((size_t)qi * p.n_chunks + chunk) * 32 + par * 16 + l15] = smax[par][tq][r];
				}
	} else {
		flush_own();
		if (lane == 0) {
			*my_log_cnt = glog_n;
			if (glog_n > p.log_cap)  // (the log is as large as this wave's share of the candidate lists)
				atomicAdd(p.lost, 1u);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (p.clk && (bid == 0 || bid == n_items - 1) && tid == 0) {
		unsigned long long* c = p.clk + (bid == 0 ? 0 : 8);  // (first and last workgroup of the grid)
		c[0] = clock64() - clk0;
		c[1] = wall_clock64() - wall0;
		for (int i = 0; i < 4; ++i)
			c[2 + i] = seg[i];
		c[6] = t1 - t0;
		c[7] = wall0;
	}
	if (p.clk && tid == 0 && bid < 65536) {  // (debug 16: every workgroup's start / end on the 100 MHz clock, and its CU;
	                                                 // the host's buffer holds 65 536 records)
		p.clk[18 + 3 * (size_t)bid] = wall0;
		p.clk[19 + 3 * (size_t)bid] = wall_clock64();
		uint32_t hwid;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
		uint32_t xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		p.clk[20 + 3 * (size_t)bid] = ((unsigned long long)xcc << 32) | hwid;
	}
	};
	// (ONE inlined copy of the item's code per instance: the sampled pass and d = 64 -- three workgroups per
	// CU on 168 registers, none to spare for the loop's state -- are always launched plainly)
	if constexpr (SAMPLE || D != 128) {
		item_body(blockIdx.x);
	} else {
		const bool persist = p.work_ctr != nullptr;
		uint32_t* const pick_slot = reinterpret_cast<uint32_t*>(smem + gemm_f16x_lds_bytes<D>() - 16);  // (behind `fills`)
		uint32_t xcc = 0, live = 0xFFu;  // tid 0: queues that may still hold items
		if (persist && tid == 0) {
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			xcc &= 7u;
		}
		for (bool first = true;; first = false) {
			uint32_t item = blockIdx.x;
			if (persist) {
				if (tid == 0) {
					uint32_t got = 0xFFFFFFFFu;
					for (uint32_t r = 0; r < 8 && got == 0xFFFFFFFFu; ++r) {
						const uint32_t y = (xcc + r) & 7u;
						if (!(live & (1u << y)))
							continue;
						const uint32_t id = 8u * atomicAdd(p.work_ctr + 16 * y, 1u) + y;
						if (id < n_items)
							got = id;
						else
							live &= ~(1u << y);
					}
					*pick_slot = got;
				}
				asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
				item = *pick_slot;
				asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (read before tid 0 picks again)
				if (item == 0xFFFFFFFFu)
					break;
			} else if (!first) {
				break;
			}
			item_body(item);
		}
	}
}

}  // namespace expann
