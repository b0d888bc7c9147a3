// scan_gemm_i8x.hpp -- the full scan of scan_gemm_i8q.hpp's 8-waves-per-tile geometry (d = 768: BASELINE
// C5; d = 1024: the padded rows of d = 832 / 960 indexes) on v_mfma_i32_16x16x64_i8.
//
// The g-domain arithmetic, parameters, LDS map, staging (counted vmcnt waits, two tile buffers) and hit
// queues of scan_gemm_i8q.hpp (rounds 1-2 ran them on v_mfma_i32_32x32x32_i8: removed in round 3).  Measured on
// the fp16 filter (scan_gemm_f16x.hpp): under a dense MFMA stream the chip holds a
// higher clock on the 16 x 16 shapes than on 32 x 32 at equal cycles per operation, and at d = 768 a
// step is 96 % MFMA issue, so the clock is what is left to win.  With 16 x 16 tiles the base row sits
// on lane & 15 and a lane's accumulators of one 16-row column -- 2 query tiles x 4 registers -- share
// ONE row term bp: one max tree + compare per column (4 per step) instead of one per 32 x 32 tile.
//
// A = queries: lane l holds query l & 15 of tile tq, 16-byte chunk 4 s + (l >> 4) of its row (k-step
// s = 64 bytes); B = base rows from LDS: row l & 15 of column tc, same chunk; C: lane l, register r
// = query 4 (l >> 4) + r of tile tq against row l & 15 of column tc.
#pragma once
#include "scan_gemm_i8q.hpp"

namespace expann {

// DR: bytes of a row slot that hold data: 13 / 15 of the 16 k-steps at d = 832 / 960 (the rest multiply zeros)
// SAMPLE (round 3: the threshold pass on the same stream, as scan_gemm_i8w_kernel<D, L2F, true>): a lane's
// accumulators of one tile column belong to ONE row, so they start at -bp (the MFMA's C operand) and the
// epilogue is the running maximum of g = dot - bp per (query register, row class = row mod 32), the two columns of
// equal parity folded by one v_max3; the row terms of tile t+1 are requested behind the barrier with its first
// fragments.  Output: p.sample_out in sample_tau_i8_kernel's layout.
template <int D, bool L2FORM, int DR = D, bool SAMPLE = false>
__global__ __launch_bounds__(I8qGeom<D>::THREADS, 2) void scan_gemm_i8x_kernel(GemmI8qParams p) {
	static_assert(D == 768 || D == 1024, "the 8-waves-per-tile geometry of scan_gemm_i8q.hpp");
	static_assert(DR <= D && DR % 64 == 0, "whole k-steps of data");
	using G = I8qGeom<D>;
	constexpr int THREADS = G::THREADS, WAVES = G::WAVES, WGQ = G::WGQ, QCAP = G::QCAP;
	static_assert(G::TQW == 1 && G::NATURAL && !G::TH_LDS, "32 queries per wave, natural chunk order");
	constexpr int ROWB = D;
	constexpr int CH = ROWB / 16;
	constexpr int KS = DR / 64;  // MFMA k-steps of 64 bytes that hold data
	constexpr int TILE_BYTES = kF16TB * ROWB;
	constexpr int NBUF = G::NBUF, PF = NBUF - 1;
	static_assert(ROWB % 256 == 0, "rows start on an LDS bank row: the XOR swizzle is (row & 15)");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l15 = lane & 15, lq = lane >> 4;
	uint32_t qtile = blockIdx.x % p.n_qtiles;
	uint32_t chunk = blockIdx.x / p.n_qtiles;
	if (p.xcd_map) {
		const uint32_t j = blockIdx.x >> 3;
		qtile = j % p.n_qtiles;
		chunk = (blockIdx.x & 7) + 8 * (j / p.n_qtiles);
	}
	const uint32_t wg_q0 = qtile * WGQ;
	const uint32_t q0 = wg_q0 + wave * 32;

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	constexpr int kNever = -2147483647 - 1;
	if (t0 >= t1) {
		if (SAMPLE) {  // (the host plans no empty chunk; if one appears its class maxima are "no row")
			for (uint32_t i = lane; i < 32 * 32; i += 64)
				if (q0 + (i >> 5) < p.m)
					p.sample_out[((size_t)(q0 + (i >> 5)) * p.n_chunks + chunk) * 32 + (i & 31)] = kNever;
		}
		return;
	}

	unsigned char* const bn_slots = smem + NBUF * TILE_BYTES;
	struct QEntry {
		int acc[16];     // values 0..7: query tile i >> 2, register i & 3 (8..15 unused)
		int bp;
		uint32_t row;
		uint32_t qrow0;  // query of value 0; value i is + 16 (i >> 2) + (i & 3)
		uint32_t pad;
	};
	static_assert(sizeof(QEntry) == kF16EntryBytes, "queue entry size");
	QEntry* const queue = reinterpret_cast<QEntry*>(bn_slots + NBUF * WAVES * 256) + wave * QCAP;
	int* const thq = reinterpret_cast<int*>(bn_slots + NBUF * WAVES * 256 + WAVES * QCAP * kF16EntryBytes);
	uint32_t* const fills = reinterpret_cast<uint32_t*>(thq + WGQ);

	i32x4 a[2][KS];
#pragma unroll
	for (int tq = 0; tq < 2; ++tq) {
		uint32_t qi = q0 + tq * 16 + l15;
		if (qi >= p.m)
			qi = p.m - 1;
		const i32x4* src = reinterpret_cast<const i32x4*>((const unsigned char*)p.queries + (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			a[tq][s] = src[4 * s + lq];
	}
	i32x4 th[2];  // accumulator start values -g_k of the query of each register
#pragma unroll
	for (int tq = 0; tq < 2; ++tq)
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
			th[tq][r] = (qi < p.m && !SAMPLE) ? p.thp[qi] : kNever / 2;  // a padded query slot never reaches a bp >= 0
		}
	if (!SAMPLE && tid < WGQ)
		thq[tid] = wg_q0 + tid < p.m ? p.thp[wg_q0 + tid] : kNever / 2;
#pragma unroll
	for (int tq = 0; tq < 2; ++tq) {
#pragma unroll
		for (int s = 0; s < KS; ++s)
			asm volatile("" : "+v"(a[tq][s]));
		asm volatile("" : "+v"(th[tq]));
	}
	// chunk 4 s + lq = 16 (s >> 2) + (4 (s & 3) + lq); the XOR with the row's swizzle (< 16) only
	// touches the low part: 4 address registers + immediates serve every k-step and column
	uint32_t aoff[4];
#pragma unroll
	for (int j = 0; j < 4; ++j)
		aoff[j] = l15 * ROWB + (((4 * j + lq) ^ l15) * 16);

	auto tile_row0 = [&](uint32_t t) -> uint32_t {
		return ((t / p.tile_run) * (p.tile_stride * p.tile_run) + (t % p.tile_run)) * kF16TB;
	};
	constexpr int N_STAGE = kF16TB * CH / THREADS;
	static_assert(kF16TB * CH % THREADS == 0, "whole staging rounds");
	constexpr int LOADS = N_STAGE + 1;
	// source offset of this thread's piece i of a tile (chunk S = i THREADS + tid: row S / CH, chunk S %
	// CH, swizzled).  1024-byte slots: a round covers 8 rows, so rounds of equal parity see the same
	// swizzle term and two registers + a multiple of 16 rows serve all of them (8 registers fewer: with
	// them the d = 960 instance spilled).
	constexpr bool SOFF2 = CH == 64 && THREADS == 512;
	uint32_t soff[SOFF2 ? 2 : N_STAGE];
#pragma unroll
	for (int i = 0; i < (SOFF2 ? 2 : N_STAGE); ++i) {
		const uint32_t S = i * THREADS + tid;
		const uint32_t r = S / CH, pc = S % CH;
		soff[i] = r * ROWB + ((pc ^ (r & 15)) * 16);
	}
	auto soff_of = [&](int i) -> uint32_t {
		if (SOFF2) {
			uint32_t b = soff[i & 1];
			asm volatile("" : "+v"(b));  // (kept opaque: hipcc would hoist the eight sums out of the tile loop again)
			return b + (uint32_t)(i >> 1) * 16u * ROWB;
		}
		return soff[SOFF2 ? 0 : i];
	};
	auto stage_piece = [&](const unsigned char* tb, uint32_t row0, int buf, int i) {
		if (i < N_STAGE) {
			unsigned char* dst0 = smem + buf * TILE_BYTES + wave * 64 * 16;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + soff_of(i < N_STAGE ? i : 0)),
			                                 (__attribute__((address_space(3))) void*)(dst0 + i * THREADS * 16), 16, 0, 0);
		} else {
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bp + row0 + lane),
			                                 (__attribute__((address_space(3))) void*)(bn_slots + (buf * WAVES + wave) * 256),
			                                 4, 0, 0);
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

	uint32_t wfill = 0;  // wave-uniform
	auto flush_own = [&]() {
		const uint32_t n = wfill < (uint32_t)QCAP ? wfill : (uint32_t)QCAP;
		constexpr int R = 8;
		uint32_t ln = (uint32_t)lane;
		asm volatile("" : "+v"(ln));  // (opaque: the queue addresses are formed HERE -- hoisted out of the tile loop they spilled at d = 960)
		for (uint32_t base = 0; base < n * 16; base += 64 * R) {
			bool hit[R];
			uint32_t qi[R], slot[R], row[R];
			int dot[R];
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const uint32_t i = base + j * 64 + ln;
				const QEntry& e = queue[i < n * 16 ? i >> 4 : 0];
				const uint32_t v = i & 15;
				const int c = e.acc[v & 7];
				row[j] = e.row;
				hit[j] = i < n * 16 && v < 8 && c >= e.bp && row[j] < p.n_rows;
				qi[j] = e.qrow0 + 16 * ((v & 7) >> 2) + (v & 3);
				dot[j] = c - thq[(qi[j] - wg_q0) & (WGQ - 1)];  // exact: acc = dot + thp[q]
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
	// the hit lanes of column tc append their 8 accumulators (a column adds at most 64 entries: the
	// queue is emptied first when they would not fit)
	static_assert(QCAP >= 24, "queue");
	auto push_hits = [&](const i32x4 (&acc)[2][4], int tc, unsigned long long mask, bool mine, int bp, uint32_t row0) {
		uint32_t left = (uint32_t)__builtin_popcountll(mask);
		if (wfill + left > (uint32_t)QCAP)
			flush_own();
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		// (more hit lanes than the whole queue holds: in rounds)
		for (uint32_t done = 0; done < left; done += (uint32_t)QCAP) {
			const bool now = mine && rank >= done && rank < done + (uint32_t)QCAP;
			if (now) {
				QEntry& e = queue[wfill + rank - done];
#pragma unroll
				for (int tq = 0; tq < 2; ++tq)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						e.acc[tq * 4 + r] = acc[tq][tc][r];
				e.bp = bp;
				e.row = row0 + tc * 16 + l15;
				e.qrow0 = q0 + 4 * lq;
			}
			const uint32_t batch = left - done < (uint32_t)QCAP ? left - done : (uint32_t)QCAP;
			wfill += batch;
			if (done + batch < left)
				flush_own();
		}
	};

#pragma unroll
	for (int i = 0; i < PF; ++i)
		stage(t0 + i, i);
	wait_vm_then_barrier<(PF - 1) * LOADS>();

	i32x4 acc[2][4];
	int buf = 0, pbuf = PF;
	uint32_t since_look = 0;
	auto frag_of = [&](int b, int tc, int s) -> i32x4 {
		return *reinterpret_cast<const i32x4*>(smem + ((uint32_t)b * TILE_BYTES + aoff[s & 3] + (s >> 2) * 256) + tc * 16 * ROWB);
	};
	auto read_bp = [&](int (&bp)[4], int b) {
		const int* slot = reinterpret_cast<const int*>(bn_slots + (b * WAVES + wave) * 256);
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			bp[tc] = slot[tc * 16 + l15];
	};
	// The step (round 3): MFMAs of tile t -> barrier (tile t+1 landed) -> the first fragments and row terms of
	// tile t+1 are REQUESTED -> epilogue of tile t (max trees, hit test, queue push) while they travel -> MFMAs of
	// tile t+1.  Before, the epilogue sat in front of the barrier and every step began with an exposed LDS round
	// trip, at the same moment in both waves of a SIMD (one workgroup per CU: they leave the barrier together).
	i32x4 fb0[4];  // k-step 0 of the tile about to be multiplied
	int bv[4], bvn[4];
	i32x4 smax[2][2];  // SAMPLE: running class maxima of g
#pragma unroll
	for (int par = 0; par < 2; ++par)
#pragma unroll
		for (int tq = 0; tq < 2; ++tq)
			smax[par][tq] = i32x4{kNever, kNever, kNever, kNever};
#pragma unroll
	for (int tc = 0; tc < 4; ++tc)
		fb0[tc] = frag_of(0, tc, 0);
	if (SAMPLE)
		read_bp(bvn, 0);
	for (uint32_t t = t0; t < t1; ++t) {
		uint32_t srow0;
		const unsigned char* stb = stage_src(t + PF, srow0);
		i32x4 fb[KS][4];
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			fb[0][tc] = fb0[tc];
		i32x4 c0[4];  // SAMPLE: -bp of this lane's row in every column
		if (SAMPLE) {
#pragma unroll
			for (int tc = 0; tc < 4; ++tc)
				c0[tc] = i32x4{-bvn[tc], -bvn[tc], -bvn[tc], -bvn[tc]};
		} else {
			read_bp(bv, buf);  // (used by the epilogue behind the barrier: the k-steps cover the read)
		}
		__builtin_amdgcn_s_setprio(1);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s) {
			if (s + 1 < KS) {  // one k-step (8 MFMAs, 128 cycles) ahead
#pragma unroll
				for (int tc = 0; tc < 4; ++tc)
					fb[s + 1][tc] = frag_of(buf, tc, s + 1);
			}
#pragma unroll
			for (int tq = 0; tq < 2; ++tq)
#pragma unroll
				for (int tc = 0; tc < 4; ++tc)
					acc[tq][tc] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[tq][s], fb[s][tc],
					                                                    s == 0 ? (SAMPLE ? c0[tc] : th[tq]) : acc[tq][tc], 0, 0, 0);
			constexpr int PER = (LOADS + KS - 1) / KS;
#pragma unroll
			for (int j = 0; j < PER; ++j)
				if (s * PER + j < LOADS)
					stage_piece(stb, srow0, pbuf, s * PER + j);
			__builtin_amdgcn_sched_barrier(0);
		}
		__builtin_amdgcn_s_setprio(0);
		const bool look = !SAMPLE && ++since_look == kF16FlushEvery;
		if (look && lane == 0)
			fills[wave] = wfill;
		wait_vm_then_barrier<(PF - 1) * LOADS>();
		const int nbuf = buf + 1 == NBUF ? 0 : buf + 1;
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			fb0[tc] = frag_of(nbuf, tc, 0);  // (past the last tile: its re-staged copy)
		if (SAMPLE)
			read_bp(bvn, nbuf);
		__builtin_amdgcn_sched_barrier(0);
		if constexpr (SAMPLE) {
#pragma unroll
			for (int par = 0; par < 2; ++par)
#pragma unroll
				for (int tq = 0; tq < 2; ++tq)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						smax[par][tq][r] = max3i(smax[par][tq][r], acc[tq][par][r], acc[tq][par + 2][r]);
			pbuf = buf;
			buf = nbuf;
			continue;
		}
		// one max tree + compare per 16-row column, one wave-uniform test per step
		int gmax[4];
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			gmax[tc] = max(max3i(acc[0][tc][0], acc[0][tc][1], acc[0][tc][2]),
			               max3i(acc[0][tc][3], acc[1][tc][0], max3i(acc[1][tc][1], acc[1][tc][2], acc[1][tc][3])));
		const bool h0 = gmax[0] >= bv[0], h1 = gmax[1] >= bv[1], h2 = gmax[2] >= bv[2], h3 = gmax[3] >= bv[3];
		if (__builtin_amdgcn_ballot_w64(h0 || h1 || h2 || h3) != 0) {
			const uint32_t row0 = tile_row0(t);
			const unsigned long long k0 = __builtin_amdgcn_ballot_w64(h0), k1 = __builtin_amdgcn_ballot_w64(h1),
			                         k2 = __builtin_amdgcn_ballot_w64(h2), k3 = __builtin_amdgcn_ballot_w64(h3);
			if (k0) push_hits(acc, 0, k0, h0, bv[0], row0);
			if (k1) push_hits(acc, 1, k1, h1, bv[1], row0);
			if (k2) push_hits(acc, 2, k2, h2, bv[2], row0);
			if (k3) push_hits(acc, 3, k3, h3, bv[3], row0);
		}
		if (wfill >= (uint32_t)QCAP * 3 / 4)
			flush_own();
		if (look) {
			since_look = 0;
			const uint32_t f = fills[lane & (WAVES - 1)];
			if (__builtin_amdgcn_ballot_w64(f >= (uint32_t)QCAP / 2) != 0)
				flush_own();
		}
		pbuf = buf;
		buf = nbuf;
	}
	if constexpr (SAMPLE) {
#pragma unroll
		for (int par = 0; par < 2; ++par)
#pragma unroll
			for (int tq = 0; tq < 2; ++tq)
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
					if (qi < p.m)
						p.sample_out[((size_t)qi * p.n_chunks + chunk) * 32 + par * 16 + l15] = smax[par][tq][r];
				}
	} else {
		flush_own();
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace expann
