// scan_gemm_i8w.hpp -- the exact 8-bit filter of scan_gemm_i8q.hpp (g domain, same parameters, same
// LDS tile image and staging) in the step structure of scan_gemm_f16x.hpp, on
// v_mfma_i32_16x16x64_i8: tile-column loop outermost, one max tree and compare per 16-row column,
// ONE barrier in the middle of the step's MFMA stream, hits leaving through per-wave logs in global
// memory (gather_logs_kernel files them).  d = 128 (SIFT: the uint8 shortcut of fp32 rows, uint8 /
// int8 indexes) and d = 256.
//
// Byte geometry: a d = 256 row is the 256 bytes of scan_gemm_f16x_kernel<128>'s fp16 row, a
// d = 128 row the 128 bytes of its d = 64 form; a 16-byte chunk is 16 k-values of one lane
// quarter, so fragment addresses, swizzle and staging are those kernels'.  d = 128 has two
// k-steps per column -- too little MFMA per step for two waves per SIMD to cover each other's
// barrier (what sank the fp16 form at d = 64) -- so this form keeps its registers under 168 and
// runs THREE workgroups per CU (scan_gemm_i8q's answer to the same problem, without its
// start-values-in-LDS detour: 4 x 4 accumulators + 32 fragment registers leave room for them).
#pragma once
#include "scan_gemm_i8q.hpp"

namespace expann {

template <int D> struct I8wGeom {
	static constexpr int WG_PER_CU = D == 128 ? 3 : 2;
	static constexpr int QCAP = D == 128 ? 64 : kF16WaveQueue;  // queue entries per wave
};
template <int D> constexpr int gemm_i8w_lds_bytes() {
	return kF16Bufs * (kF16TB * D + kF16Waves * 256) + kF16Waves * I8wGeom<D>::QCAP * kF16EntryBytes + kF16TQ * 4 + 64;
}
static_assert(gemm_i8w_lds_bytes<128>() * 3 <= 160 * 1024 && gemm_i8w_lds_bytes<256>() * 2 <= 160 * 1024,
              "LDS budget per CU");

// the hit logs (GemmF16Params::log ... of scan_gemm_f16x.hpp) next to scan_gemm_i8q's parameters
struct GemmI8wParams {
	GemmI8qParams q;
	uint4* log;
	uint32_t* log_cnt;
	uint32_t log_cap;
	uint32_t* lost;
};

// SAMPLE (round 3: the threshold pass on the same stream, as scan_gemm_f16x_kernel<D, true>): no thresholds, no
// candidate path; a lane's accumulators of one tile column belong to ONE row, so the row term enters as the
// MFMA's C operand -- they start at -bp -- and the epilogue is the running maximum of g = dot - bp per (query
// register, row class), class = row mod 32 = 16 (column & 1) + lane & 15, the two columns of equal parity
// folded by one v_max3.  A padding row's bp is 2^30: its g never wins.  Output: q.sample_out, the layout
// sample_tau_i8_kernel reads ([(query * n_chunks + chunk) * 32 + class]).
template <int D, bool L2FORM, bool SAMPLE = false>
__global__ __launch_bounds__(kF16Threads, I8wGeom<D>::WG_PER_CU) void scan_gemm_i8w_kernel(GemmI8wParams pw) {
	static_assert(D == 128 || D == 256, "built for d = 128, 256");
	const GemmI8qParams& p = pw.q;
	constexpr int THREADS = kF16Threads, WAVES = kF16Waves, WGQ = kF16TQ, QCAP = I8wGeom<D>::QCAP;
	constexpr int ROWB = D, CH = ROWB / 16;
	constexpr int KS = D / 64;  // MFMA k-steps of 64
	constexpr int TILE_BYTES = kF16TB * ROWB;
	constexpr int RPB = (ROWB < 256) ? 256 / ROWB : 1;
	constexpr int SWM = (CH < 16 ? CH : 16) - 1;
	constexpr int NBUF = kF16Bufs;
	constexpr int kNever = -2147483647 - 1;
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
	const uint32_t q0 = wg_q0 + wave * 64;

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	uint32_t* const my_log_cnt = SAMPLE ? nullptr : pw.log_cnt + (size_t)blockIdx.x * WAVES + wave;
	if (t0 >= t1) {
		if (SAMPLE) {  // (the host plans no empty chunk; if one appears its class maxima are "no row")
			for (uint32_t i = lane; i < 64 * 32; i += 64)
				if (q0 + (i >> 5) < p.m)
					p.sample_out[((size_t)(q0 + (i >> 5)) * p.n_chunks + chunk) * 32 + (i & 31)] = kNever;
		} else if (lane == 0) {
			*my_log_cnt = 0;
		}
		return;
	}

	// LDS map: tiles, per-wave bp slots, per-wave queues, accumulator start values, queue fills
	unsigned char* const bn_slots = smem + NBUF * TILE_BYTES;
	struct QEntry {
		int acc[16];     // value i = query tile i >> 2, register i & 3
		int bp;
		uint32_t row;
		uint32_t qrow0;  // query of value 0; value i is + 16 (i >> 2) + (i & 3)
		uint32_t pad;
	};
	static_assert(sizeof(QEntry) == kF16EntryBytes, "queue entry size");
	QEntry* const queue = reinterpret_cast<QEntry*>(bn_slots + NBUF * WAVES * 256) + wave * QCAP;
	// (behind the queues: WGQ words that held the queries' start values for the flush's scores until round 3)
	uint32_t* const fills = reinterpret_cast<uint32_t*>(bn_slots + NBUF * WAVES * 256 + WAVES * QCAP * kF16EntryBytes) + WGQ;

	// query fragments: lane l holds query l & 15 of tile tq, chunk 4 s + (l >> 4) of its row
	i32x4 a[4][KS];
#pragma unroll
	for (int tq = 0; tq < 4; ++tq) {
		uint32_t qi = q0 + tq * 16 + l15;
		if (qi >= p.m)
			qi = p.m - 1;
		const i32x4* src = reinterpret_cast<const i32x4*>((const unsigned char*)p.queries + (size_t)qi * ROWB);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			a[tq][s] = src[4 * s + lq];
	}
	// accumulator start values -g_k of the query of each register (a padded query slot starts at
	// INT_MIN / 2: dot + that never reaches a bp >= 0)
	i32x4 th[4];
#pragma unroll
	for (int tq = 0; tq < 4; ++tq)
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
			th[tq][r] = (qi < p.m && !SAMPLE) ? p.thp[qi] : kNever / 2;
		}
#pragma unroll
	for (int tq = 0; tq < 4; ++tq) {
#pragma unroll
		for (int s = 0; s < KS; ++s)
			asm volatile("" : "+v"(a[tq][s]));
		asm volatile("" : "+v"(th[tq]));
	}
	uint32_t aoff[KS];
#pragma unroll
	for (int s = 0; s < KS; ++s)
		aoff[s] = l15 * ROWB + (((4 * s + lq) ^ ((l15 / RPB) & SWM)) * 16);
	static_assert((16 / RPB) % (SWM + 1) == 0, "the swizzle term repeats every 16 rows");

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
		soff[i] = r * ROWB + ((pc ^ ((r / RPB) & SWM)) * 16);
	}
	auto stage_piece = [&](const unsigned char* tb, uint32_t row0, int buf, int i) {
		if (i < N_STAGE) {
			unsigned char* dst0 = smem + buf * TILE_BYTES + wave * 64 * 16;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + soff[i < N_STAGE ? i : 0]),
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

	// candidates: hit lanes append their 16 raw accumulators to the wave's LDS queue; a flush redoes
	// the compare with 16 lanes per entry and appends the hits -- exact integer scores, ballot-
	// compacted 16-byte stores at a wave-uniform position -- to this wave's log in global memory
	uint32_t wfill = 0;   // wave-uniform: entries in the LDS queue
	uint32_t glog_n = 0;  // wave-uniform: entries in this wave's global log
	uint4* const my_log = SAMPLE ? nullptr : pw.log + ((size_t)blockIdx.x * WAVES + wave) * pw.log_cap;
	auto flush_own = [&]() {
		const uint32_t n = wfill < (uint32_t)QCAP ? wfill : (uint32_t)QCAP;
		// 16 lanes per entry, two rounds of 4 entries in flight; a log entry is {raw accumulator, row, query}:
		// gather_logs_kernel makes the exact score (GatherLogParams::i_mode)
		for (uint32_t base = 0; base < n * 16; base += 128) {
			bool hit[2];
			int cv[2];
			uint32_t row[2], qi[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const uint32_t i = base + 64 * u + lane;
				const QEntry& e = queue[i < n * 16 ? i >> 4 : 0];
				const uint32_t v = i & 15;
				cv[u] = e.acc[v];
				row[u] = e.row;
				hit[u] = i < n * 16 && cv[u] >= e.bp && row[u] < p.n_rows;
				qi[u] = e.qrow0 + 16 * (v >> 2) + (v & 3);
			}
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit[u]);
				if (mask == 0)
					continue;
				const uint32_t pos = glog_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
				                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
				if (hit[u] && pos < pw.log_cap)
					my_log[pos] = make_uint4((uint32_t)cv[u], row[u], qi[u], 0u);
				glog_n += (uint32_t)__builtin_popcountll(mask);
			}
		}
		wfill = 0;
	};
	auto push_hits = [&](const i32x4 (&acc)[4][4], int tc, unsigned long long mask, bool mine, int bp,
	                     uint32_t row0) {
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
				e.bp = bp;
				e.row = brow;
				e.qrow0 = qrow0;
			}
		}
		wfill += (uint32_t)__builtin_popcountll(mask);
	};

	// ---- the pipeline: scan_gemm_f16x.hpp's step (ONE barrier, after tile column 1) ---------------
	stage(t0, 0);
	stage(t0 + 1, 1);
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tiles t0, t0+1 landed

	i32x4 acc[4][4];
	int buf = 0;
	uint32_t since_look = 0;
	auto frag_at = [&](int b, int tc, int s) -> i32x4 {
		return *reinterpret_cast<const i32x4*>(smem + ((uint32_t)b * TILE_BYTES + aoff[s]) + tc * 16 * ROWB);
	};
	auto read_bp = [&](int (&bp)[4], int b) {
		const int* slot = reinterpret_cast<const int*>(bn_slots + (b * WAVES + wave) * 256);
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			bp[tc] = slot[tc * 16 + l15];
	};
	auto col_max = [&](int pc) -> int {
		int m0 = max3i(acc[0][pc][0], acc[0][pc][1], acc[0][pc][2]);
		int m1 = max3i(acc[0][pc][3], acc[1][pc][0], acc[1][pc][1]);
		int m2 = max3i(acc[1][pc][2], acc[1][pc][3], acc[2][pc][0]);
		int m3 = max3i(acc[2][pc][1], acc[2][pc][2], acc[2][pc][3]);
		int m4 = max3i(acc[3][pc][0], acc[3][pc][1], acc[3][pc][2]);
		m0 = max3i(m0, m1, acc[3][pc][3]);
		m2 = max3i(m2, m3, m4);
		return max(m0, m2);
	};
	// k-steps [s0, s1) of tile column tc; a column is issued as k-step 0, the NEXT column's fragment reads, the
	// rest (scan_gemm_f16x.hpp: no lgkmcnt(0) in front of a column that exposes those reads' round trip)
	int bpv[4];
	i32x4 smax[2][4];  // SAMPLE: running class maxima of g
#pragma unroll
	for (int par = 0; par < 2; ++par)
#pragma unroll
		for (int tq = 0; tq < 4; ++tq)
			smax[par][tq] = i32x4{kNever, kNever, kNever, kNever};
	auto fold = [&](int par) {
#pragma unroll
		for (int tq = 0; tq < 4; ++tq)
#pragma unroll
			for (int r = 0; r < 4; ++r)
				smax[par][tq][r] = max3i(smax[par][tq][r], acc[tq][par][r], acc[tq][par + 2][r]);
	};
	auto mfma_part = [&](int tc, const i32x4 (&f)[KS], int s0, int s1) {
		const int nb = -bpv[tc];
		const i32x4 c0 = {nb, nb, nb, nb};
#pragma unroll
		for (int s = s0; s < s1; ++s)
#pragma unroll
			for (int tq = 0; tq < 4; ++tq)
				acc[tq][tc] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[tq][s], f[s], s == 0 ? (SAMPLE ? c0 : th[tq]) : acc[tq][tc], 0, 0, 0);
	};

	i32x4 fb[2][KS];  // fragments of the column being multiplied / the next one
#pragma unroll
	for (int s = 0; s < KS; ++s)
		fb[0][s] = frag_at(0, 0, s);
	read_bp(bpv, 0);
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
	for (uint32_t t = t0; t < t1; ++t) {
		const int nbuf = buf + 1 == NBUF ? 0 : buf + 1;   // tile t+1
		const int pbuf = buf == 0 ? NBUF - 1 : buf - 1;   // tile t-1 -> takes tile t+2
		int gmax[4];
		int bpn[4];
		__builtin_amdgcn_s_setprio(1);
		mfma_part(0, fb[0], 0, 1);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[1][s] = frag_at(buf, 1, s);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(0, fb[0], 1, KS);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(1, fb[1], 0, 1);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[0][s] = frag_at(buf, 2, s);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(1, fb[1], 1, KS);
		if (!SAMPLE) {
			gmax[0] = col_max(0);
			asm volatile("" : "+v"(gmax[0]));  // (computed here, between column 1's MFMAs)
		}
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_setprio(0);
		// ---- the step's barrier: tile t+1 landed, tile t-1 released ------------------------------
		const bool look = !SAMPLE && ++since_look == kF16FlushEvery;
		if (look && lane == 0)
			fills[wave] = wfill;
		asm volatile("" ::: "memory");
		__builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0), as builtins: the compiler's wait insertion sees them
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");
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
		mfma_part(2, fb[0], 0, 1);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[1][s] = frag_at(buf, 3, s);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(2, fb[0], 1, KS);
#pragma unroll
		for (int i = 0; i < (LOADS + 1) / 2; ++i)
			stage_piece(stb, srow0, pbuf, i);
		if (!SAMPLE)
			gmax[1] = col_max(1);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(3, fb[1], 0, 1);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KS; ++s)
			fb[0][s] = frag_at(nbuf, 0, s);
		read_bp(bpn, nbuf);
		__builtin_amdgcn_sched_barrier(0);
		mfma_part(3, fb[1], 1, KS);
#pragma unroll
		for (int i = (LOADS + 1) / 2; i < LOADS; ++i)
			stage_piece(stb, srow0, pbuf, i);
		if (SAMPLE)
			fold(0);
		else
			gmax[2] = col_max(2);
		__builtin_amdgcn_s_setprio(0);
		if (SAMPLE) {
			fold(1);
		} else {
			gmax[3] = col_max(3);
			// one wave-uniform test per step; the queue push is the rare path
			const bool h0 = gmax[0] >= bpv[0], h1 = gmax[1] >= bpv[1], h2 = gmax[2] >= bpv[2], h3 = gmax[3] >= bpv[3];
			if (__builtin_amdgcn_ballot_w64(h0 || h1 || h2 || h3) != 0) {
				const uint32_t row0 = tile_row0(t);
				const unsigned long long k0 = __builtin_amdgcn_ballot_w64(h0), k1 = __builtin_amdgcn_ballot_w64(h1),
				                         k2 = __builtin_amdgcn_ballot_w64(h2), k3 = __builtin_amdgcn_ballot_w64(h3);
				if (k0) push_hits(acc, 0, k0, h0, bpv[0], row0);
				if (k1) push_hits(acc, 1, k1, h1, bpv[1], row0);
				if (k2) push_hits(acc, 2, k2, h2, bpv[2], row0);
				if (k3) push_hits(acc, 3, k3, h3, bpv[3], row0);
			}
			if (wfill >= (uint32_t)QCAP * 3 / 4)
				flush_own();
		}
		__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): tile t+1's first fragments and row terms, requested under column 3
#pragma unroll
		for (int tc = 0; tc < 4; ++tc)
			bpv[tc] = bpn[tc];
		buf = nbuf;
	}
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
			if (glog_n > pw.log_cap)  // (the log is as large as this wave's share of the candidate lists)
				atomicAdd(pw.lost, 1u);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-staged tail tiles: LDS must outlive them
}

}  // namespace expann
