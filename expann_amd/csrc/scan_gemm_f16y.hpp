// scan_gemm_f16y.hpp -- the fp16 candidate filter of scan_gemm_f16.hpp for d = 256 / 512 (the
// 8-waves-per-tile geometry: 8 waves x 32 queries share each staged tile, one workgroup per CU) on
// v_mfma_f32_16x16x32_f16: same slack analysis, parameters, LDS map, staging and hit queues as
// scan_gemm_f16_kernel<D, false>; the MFMA shape is what differs (the chip holds a higher clock on
// the 16 x 16 shapes under a dense MFMA stream: DESIGN.md 4.4x, scan_gemm_i8x.hpp), and a lane's
// accumulators of one 16-row column -- 2 query tiles x 4 registers -- share ONE row term bn': one max
// tree + compare per column.  A = queries: lane l holds query l & 15 of tile tq, 16-byte chunk 4 s +
// (l >> 4) of its row; B = base rows from LDS: row l & 15 of column tc, same chunk; C: lane l,
// register r = query 4 (l >> 4) + r of tile tq against row l & 15 of column tc.
// SAMPLE (round 3): the sampled pass on the same stream, as scan_gemm_f16x_kernel<D, true> -- the row term
// enters as the MFMA's C operand (the accumulators start at -bn'), the epilogue is the running maximum of
// g per (query register, row class = 16 (column & 1) + lane & 15), two columns of equal parity per v_max3.
#pragma once
#include "scan_gemm_f16x.hpp"

namespace expann {

template <int D, bool SAMPLE = false>
__global__ __launch_bounds__(F16Geom<D>::THREADS, 2) void scan_gemm_f16y_kernel(GemmF16Params p) {
	static_assert(D == 256 || D == 512, "the 8-waves-per-tile geometry of scan_gemm_f16.hpp");
	using G = F16Geom<D>;
	constexpr int THREADS = G::THREADS, WAVES = G::WAVES, WGQ = G::WGQ, QCAP = G::QCAP;
	static_assert(G::TQW == 1 && G::NATURAL && !G::TH_LDS, "32 queries per wave, natural chunk order");
	constexpr int ROWB = D * 2;  // bytes per fp16 row
	constexpr int CH = ROWB / 16;
	constexpr int KS = D / 32;  // MFMA k-steps of 32 elements (64 bytes)
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
	if (t0 >= t1)
		return;

	unsigned char* const bn_slots = smem + NBUF * TILE_BYTES;
	struct QEntry {
		float acc[16];   // values 0..7: query tile i >> 2, register i & 3 (8..15 unused)
		float bn;
		uint32_t row;
		uint32_t qrow0;  // query of value 0; value i is + 16 (i >> 2) + (i & 3)
		uint32_t pad;
	};
	static_assert(sizeof(QEntry) == kF16EntryBytes, "queue entry size");
	QEntry* const queue = reinterpret_cast<QEntry*>(bn_slots + NBUF * WAVES * 256) + wave * QCAP;
	float* const thq = reinterpret_cast<float*>(bn_slots + NBUF * WAVES * 256 + WAVES * QCAP * kF16EntryBytes);
	uint32_t* const fills = reinterpret_cast<uint32_t*>(thq + WGQ);

	f16x8 a[2][KS];
#pragma unroll
	for (int tq = 0; tq < 2; ++tq) {
		uint32_t qi = q0 + tq * 16 + l15;
		if (qi >= p.m)
			qi = p.m - 1;
		const f16x8* src = reinterpret_cast<const f16x8*>((const unsigned char*)p.queries_f16 + (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			a[tq][s] = src[4 * s + lq];
	}
	f32x4 th[2];  // accumulator start values theta' of the query of each register
#pragma unroll
	for (int tq = 0; tq < 2; ++tq)
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
			th[tq][r] = (qi < p.m && !SAMPLE) ? p.theta[qi] : -__builtin_inff();
		}
	if (!SAMPLE && tid < WGQ)
		thq[tid] = wg_q0 + tid < p.m ? p.theta[wg_q0 + tid] : -__builtin_inff();
	f32x4 smax[2][2];  // SAMPLE: running class maxima [column parity][query tile]
#pragma unroll
	for (int par = 0; par < 2; ++par)
#pragma unroll
		for (int tq = 0; tq < 2; ++tq)
			smax[par][tq] = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
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
	uint32_t soff[N_STAGE];
#pragma unroll
	for (int i = 0; i < N_STAGE; ++i) {
		const uint32_t S = i * THREADS + tid;
		const uint32_t r = S / CH, pc = S % CH;
		soff[i] = r * ROWB + ((pc ^ (r & 15)) * 16);
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

	uint32_t wfill = 0;  // wave-uniform
	// approximate key of a hit: bn(1-eps) - abs|b| - 2 q16.b16/s^2 = ((bn' - acc) + theta') * 2/s^2
	auto flush_own = [&]() {
		const uint32_t n = wfill < (uint32_t)QCAP ? wfill : (uint32_t)QCAP;
		constexpr int R = 4;
		uint32_t ln = (uint32_t)lane;
		asm volatile("" : "+v"(ln));  // (opaque: the queue addresses are formed HERE, not held across the tile loop)
		for (uint32_t base = 0; base < n * 16; base += 64 * R) {
			bool hit[R];
			uint32_t qi[R], slot[R];
			uint64_t key[R];
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const uint32_t i = base + j * 64 + ln;
				const QEntry& e = queue[i < n * 16 ? i >> 4 : 0];
				const uint32_t v = i & 15;
				const float c = e.acc[v & 7], bn = e.bn;
				hit[j] = i < n * 16 && v < 8 && c >= bn;
				qi[j] = e.qrow0 + 16 * ((v & 7) >> 2) + (v & 3);
				key[j] = make_key(((bn - c) + thq[(qi[j] - wg_q0) & (WGQ - 1)]) * p.two_inv_s2, e.row);
			}
#pragma unroll
			for (int j = 0; j < R; ++j)
				slot[j] = hit[j] ? atomicAdd(&p.cand_cnt[qi[j]], 1u) : 0xFFFFFFFFu;
#pragma unroll
			for (int j = 0; j < R; ++j)
				if (hit[j] && slot[j] < p.cap)
					p.cand[(size_t)qi[j] * p.cap + slot[j]] = key[j];
		}
		wfill = 0;
	};
	// the hit lanes of column tc append their 8 accumulators (a column adds at most 64 entries: the
	// queue is emptied first when they would not fit)
	static_assert(QCAP >= 24, "queue");
	auto push_hits = [&](const f32x4 (&acc)[2][4], int tc, unsigned long long mask, bool mine, float bn, uint32_t row0) {
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
				e.bn = bn;
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

	f32x4 acc[2][4];
	int buf = 0, pbuf = PF;
	uint32_t since_look = 0;
	auto frag_of = [&](int b, int tc, int s) -> f16x8 {
		return *reinterpret_cast<const f16x8*>(smem + ((uint32_t)b * TILE_BYTES + aoff[s & 3] + (s >> 2) * 256) + tc * 16 * ROWB);
	};
	auto read_bn = [&](float (&bn)[4], int b) {
		const float* slot = reinterpret_cast<const float*>(bn_slots + (b * WAVES + wave) * 256);
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			bn[tc] = slot[tc * 16 + l15];
	};
	// The step (round 3, as scan_gemm_i8x.hpp): MFMAs of tile t -> barrier (tile t+1 landed) -> the first
	// fragments of tile t+1 are REQUESTED -> epilogue of tile t while they travel -> MFMAs of tile t+1.
	f16x8 fb0[4];  // k-step 0 of the tile about to be multiplied
	float bv[4];
#pragma unroll
	for (int tc = 0; tc < 4; ++tc)
		fb0[tc] = frag_of(0, tc, 0);
	for (uint32_t t = t0; t < t1; ++t) {
		uint32_t srow0;
		const unsigned char* stb = stage_src(t + PF, srow0);
		f16x8 fb[KS][4];
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			fb[0][tc] = fb0[tc];
		read_bn(bv, buf);  // (SAMPLE: the MFMAs' C operand; else used by the epilogue behind the barrier)
		f32x4 c0[4];  // SAMPLE: -bn' of this lane's row in every column
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			c0[tc] = f32x4{-bv[tc], -bv[tc], -bv[tc], -bv[tc]};
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
					acc[tq][tc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tq][s], fb[s][tc],
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
		__builtin_amdgcn_sched_barrier(0);
		if constexpr (SAMPLE) {
#pragma unroll
			for (int par = 0; par < 2; ++par)
#pragma unroll
				for (int tq = 0; tq < 2; ++tq)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						smax[par][tq][r] = max3f(smax[par][tq][r], acc[tq][par][r], acc[tq][par + 2][r]);
		} else {
			// one max tree + compare per 16-row column, one wave-uniform test per step
			float gmax[4];
#pragma unroll
			for (int tc = 0; tc < 4; ++tc)
				gmax[tc] = __builtin_fmaxf(max3f(acc[0][tc][0], acc[0][tc][1], acc[0][tc][2]),
				                           max3f(acc[0][tc][3], acc[1][tc][0], max3f(acc[1][tc][1], acc[1][tc][2], acc[1][tc][3])));
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
				uint32_t ln = (uint32_t)lane;
				asm volatile("" : "+v"(ln));  // (opaque, as in flush_own: one address register fewer across the tile loop)
				const uint32_t f = fills[ln & (WAVES - 1)];
				if (__builtin_amdgcn_ballot_w64(f >= (uint32_t)QCAP / 2) != 0)
					flush_own();
			}
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
