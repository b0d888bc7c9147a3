// scan_gemm_f16kx.hpp -- scan_gemm_f16k.hpp's k-split filter (512 < d <= 960: the k range of a 32-query
// group split over a pair of waves, 32-row tiles, partial sums handed over through LDS) on
// v_mfma_f32_16x16x32_f16: same geometry (F16kGeom), LDS map, staging, exchange and queues; the MFMA
// shape differs (the higher clock the chip holds on 16 x 16 shapes: DESIGN.md 4.4x).  A wave's step is
// 2 query tiles x 2 row-tile columns x its k-steps of 32 elements; lane l holds query l & 15 of
// tile tq and row l & 15 of column tc, 16-byte chunk 4 s + (l >> 4); the swizzles of the 32-row
// form stay conflict-free for 16-row reads (rows r and r + 16 share a swizzle term; within a
// ds_read_b128 lane group the (row, chunk) pairs {rows 0-3, 12-15: chunk c; rows 4-11: chunk c + 1}
// land on 16 distinct bank quads for 0- and 128-mod-256 row strides alike).
#pragma once
#include "scan_gemm_f16k.hpp"
#include "scan_gemm_f16x.hpp"

namespace expann {

// SAMPLE (round 3): the sampled pass on the same stream -- the first-half wave's accumulators start at -bn'
// (the row term as the MFMA's C operand), after the exchange it keeps the running maximum of g per (query
// register, row class = 16 column + lane & 15); as scan_gemm_f16x_kernel<D, true>.
template <int D, bool SAMPLE = false>
__global__ __launch_bounds__(512, 2) void scan_gemm_f16kx_kernel(GemmF16Params p) {
	static_assert(D == 768 || D == 832 || D == 960, "built for d = 768, 832, 960");
	using G = F16kGeom<D>;
	constexpr int THREADS = G::THREADS, WGQ = G::WGQ, QCAP = G::QCAP, TB = G::TB;
	constexpr int ROWB = G::ROWB, CH = G::CH, SWG = G::SWG;
	// k-steps of 32 elements (4 chunks): the wave pair splits them at a multiple of the swizzle group
	constexpr int KSA = G::KSA / 2, KSB = G::KSB / 2, NA = SWG / 4;
	static_assert(G::KSA % 2 == 0 && G::KSB % 2 == 0 && KSA % NA == 0, "k split of the 16-element form halves");
	constexpr int TILE_BYTES = G::TILE_BYTES;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int qg = wave & 3, kh = wave >> 2;
	const int l15 = lane & 15, lq = lane >> 4, r31 = lane & 31;
	uint32_t qtile = blockIdx.x % p.n_qtiles;
	uint32_t chunk = blockIdx.x / p.n_qtiles;
	if (p.xcd_map) {  // as scan_gemm_f16_kernel: the 8 row chunks {x, x+8, ..} of a query tile on XCD x
		const uint32_t j = blockIdx.x >> 3;
		qtile = j % p.n_qtiles;
		chunk = (blockIdx.x & 7) + 8 * (j / p.n_qtiles);
	}
	const uint32_t wg_q0 = qtile * WGQ;
	const uint32_t q0 = wg_q0 + qg * 32;  // this pair's queries

	const uint32_t t0 = chunk * p.tiles_per_block;
	uint32_t t1 = t0 + p.tiles_per_block;
	if (t1 > p.n_tiles_sel)
		t1 = p.n_tiles_sel;
	if (t0 >= t1)
		return;  // (whole workgroup)

	// LDS map
	unsigned char* const bn_slots = smem + G::NBUF * TILE_BYTES;
	unsigned char* const xch = bn_slots + G::NBUF * 256;
	struct QEntry {
		float acc[16];   // values 0..7: query tile i >> 2, register i & 3 (8..15 unused)
		float bn;
		uint32_t row;
		uint32_t qrow0;  // query of value 0; value i is + 16 (i >> 2) + (i & 3)
		uint32_t pad;
	};
	static_assert(sizeof(QEntry) == kF16EntryBytes, "queue entry size");
	QEntry* const queue = reinterpret_cast<QEntry*>(xch + G::XCH_BYTES) + qg * QCAP;  // (kh = 0 waves)
	float* const thq = reinterpret_cast<float*>(xch + G::XCH_BYTES + G::PAIRS * QCAP * kF16EntryBytes);
	uint32_t* const fills = reinterpret_cast<uint32_t*>(thq + WGQ);

	// this wave's k-steps: kh * KSA + s
	f16x8 a[2][KSB];
#pragma unroll
	for (int tq = 0; tq < 2; ++tq) {
		uint32_t qi = q0 + tq * 16 + l15;
		if (qi >= p.m)
			qi = p.m - 1;
		const f16x8* src =
		    reinterpret_cast<const f16x8*>((const unsigned char*)p.queries_f16 + (size_t)qi * ROWB) + 4 * kh * KSA;
#pragma unroll
		for (int s = 0; s < KSB; ++s)
			a[tq][s] = src[4 * s + lq];  // (kh = 0: the last KSB - KSA are loaded but never multiplied)
	}
	// accumulator start values: theta' with the first k-half, zero with the second
	f32x4 th[2];
#pragma unroll
	for (int tq = 0; tq < 2; ++tq)
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
			th[tq][r] = kh == 0 ? ((qi < p.m && !SAMPLE) ? p.theta[qi] : -__builtin_inff()) : 0.0f;
		}
	if (!SAMPLE && tid < WGQ)
		thq[tid] = wg_q0 + tid < p.m ? p.theta[wg_q0 + tid] : -__builtin_inff();
	f32x4 smax[2][2];  // SAMPLE: running class maxima [column][query tile]
#pragma unroll
	for (int tc = 0; tc < 2; ++tc)
#pragma unroll
		for (int tq = 0; tq < 2; ++tq)
			smax[tc][tq] = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
	if (tid < 8)
		fills[tid] = 0;
#pragma unroll
	for (int tq = 0; tq < 2; ++tq) {
#pragma unroll
		for (int s = 0; s < KSB; ++s)
			asm volatile("" : "+v"(a[tq][s]));  // in registers before the first stage load (see scan_gemm_f16_kernel)
		asm volatile("" : "+v"(th[tq]));
	}

	auto swz = [](uint32_t r) -> uint32_t { return G::HALF ? ((r >> 1) & 7) : (r & 15); };
	// per-lane LDS offset of local k-step s (row l15 of tile column 0; column 1 is 16 rows further,
	// where the swizzle term is the same): chunk 4 (kh KSA + s) + lq = group base + (4 (s % NA) + lq),
	// the XOR with the row's swizzle stays inside the group: NA registers + immediates
	uint32_t aoff[NA];
#pragma unroll
	for (int j = 0; j < NA; ++j)
		aoff[j] = l15 * ROWB + kh * (4 * KSA * 16) + (((4 * j + lq) ^ swz(l15)) * 16);

	auto tile_row0 = [&](uint32_t t) -> uint32_t {
		return ((t / p.tile_run) * (p.tile_stride * p.tile_run) + (t % p.tile_run)) * TB;
	};

	// Staging by LDS-DMA: 16-byte slot S = i * 512 + tid of the tile (LDS order = row-major physical
	// chunks; source = logical chunk pc ^ swizzle(row)); the last round covers only the first
	// REM / 64 waves, and wave 7 brings the tile's 32 bn' in the same round.
	constexpr int N_FULL = TB * CH / THREADS;
	constexpr int REM = TB * CH - N_FULL * THREADS;
	static_assert(REM % 64 == 0 && REM <= 7 * 64, "last staging round: whole waves, wave 7 free");
	constexpr int LOADS = N_FULL + 1;
	static_assert(LOADS <= 2 * KSA, "two stage pieces per k-step at most");
	uint32_t soff[LOADS];
#pragma unroll
	for (int i = 0; i < LOADS; ++i) {
		uint32_t S = i * THREADS + tid;
		if (S >= (uint32_t)(TB * CH))
			S = 0;
		const uint32_t r = S / CH, pc = S % CH;
		soff[i] = r * ROWB + ((pc ^ swz(r)) * 16);
	}
	auto stage_piece = [&](const unsigned char* tb, uint32_t row0, int buf, int i) {
		if (i < N_FULL || wave * 64 < REM) {
			__builtin_amdgcn_global_load_lds(
			    (const __attribute__((address_space(1))) void*)(tb + soff[i]),
			    (__attribute__((address_space(3))) void*)(smem + buf * TILE_BYTES + wave * 64 * 16 + i * THREADS * 16),
			    16, 0, 0);
		} else if (wave == 7) {
			__builtin_amdgcn_global_load_lds(
			    (const __attribute__((address_space(1))) void*)(p.bnorm + row0 + r31),
			    (__attribute__((address_space(3))) void*)(bn_slots + buf * 256), 4, 0, 0);
		}
	};
	auto stage_src = [&](uint32_t t, uint32_t& row0) -> const unsigned char* {
		if (t > t1 - 1)
			t = t1 - 1;
		row0 = tile_row0(t);
		return (const unsigned char*)p.base_f16 + (size_t)row0 * ROWB;
	};

	// candidate queue of the kh = 0 wave (scan_gemm_f16_kernel's, one column tile)
	uint32_t wfill = 0;  // wave-uniform
	auto flush_own = [&]() {
		const uint32_t n = wfill < (uint32_t)QCAP ? wfill : (uint32_t)QCAP;
		constexpr int R = 4;
		for (uint32_t base = 0; base < n * 16; base += 64 * R) {
			bool hit[R];
			uint32_t qi[R], slot[R];
			uint64_t key[R];
#pragma unroll
			for (int j = 0; j < R; ++j) {
				const uint32_t i = base + j * 64 + lane;
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
	// one max tree + compare per 16-row column; the hit lanes append their 8 accumulators (a column
	// adds at most 64 entries: in rounds through the queue when they do not fit)
	auto push_hits = [&](const f32x4 (&c)[2][2], int tc, unsigned long long mask, bool mine, float bn, uint32_t row0) {
		const uint32_t left = (uint32_t)__builtin_popcountll(mask);
		if (wfill + left > (uint32_t)QCAP)
			flush_own();
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		for (uint32_t done = 0; done < left; done += (uint32_t)QCAP) {
			if (mine && rank >= done && rank < done + (uint32_t)QCAP) {
				QEntry& e = queue[wfill + rank - done];
#pragma unroll
				for (int tq = 0; tq < 2; ++tq)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						e.acc[tq * 4 + r] = c[tq][tc][r];
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
	auto epilogue = [&](const f32x4 (&c)[2][2], uint32_t row0, const float (&bn)[2]) {
		float g[2];
#pragma unroll
		for (int tc = 0; tc < 2; ++tc)
			g[tc] = __builtin_fmaxf(max3f(c[0][tc][0], c[0][tc][1], c[0][tc][2]),
			                        max3f(c[0][tc][3], c[1][tc][0], max3f(c[1][tc][1], c[1][tc][2], c[1][tc][3])));
		const bool h0 = g[0] >= bn[0], h1 = g[1] >= bn[1];
		if (__builtin_amdgcn_ballot_w64(h0 || h1) != 0) {
			const unsigned long long k0 = __builtin_amdgcn_ballot_w64(h0), k1 = __builtin_amdgcn_ballot_w64(h1);
			if (k0) push_hits(c, 0, k0, h0, bn[0], row0);
			if (k1) push_hits(c, 1, k1, h1, bn[1], row0);
		}
	};

	{
		uint32_t row0;
		const unsigned char* tb = stage_src(t0, row0);
#pragma unroll
		for (int i = 0; i < LOADS; ++i)
			stage_piece(tb, row0, 0, i);
	}
	wait_vm_then_barrier<0>();  // tile t0 landed, thq / fills visible

	f32x4 acc[2][2];
	int buf = 0;
	uint32_t par = 0, since_look = 0;
	for (uint32_t t = t0; t < t1; ++t) {
		const uint32_t boff = (uint32_t)buf * TILE_BYTES;
		auto frag = [&](int tc, int s) -> f16x8 {
			return *reinterpret_cast<const f16x8*>(smem + (boff + aoff[s % NA]) + (s / NA) * (SWG * 16) + tc * 16 * ROWB);
		};
		uint32_t srow0;
		const unsigned char* stb = stage_src(t + 1, srow0);
		constexpr int FD = 1;  // k-steps (4 MFMAs, 64 cycles) of fragment read-ahead
		f16x8 fb[KSB][2];
#pragma unroll
		for (int s = 0; s < FD; ++s) {
			fb[s][0] = frag(0, s);
			fb[s][1] = frag(1, s);
		}
		// the bn' of this tile: its slot is re-staged during the NEXT step, so it is read now
		const float* bslot = reinterpret_cast<const float*>(bn_slots + buf * 256);
		const float bnv[2] = {bslot[l15], bslot[16 + l15]};
		f32x4 c0[2];  // SAMPLE: the first-half wave starts at -bn' of its lane's rows, the second at zero
#pragma unroll
		for (int tc = 0; tc < 2; ++tc) {
			const float v = kh == 0 ? -bnv[tc] : 0.0f;
			c0[tc] = f32x4{v, v, v, v};
		}
		__builtin_amdgcn_s_setprio(1);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KSA; ++s) {
			if (s + FD < KSA || (s + FD < KSB && kh)) {
				fb[s + FD][0] = frag(0, s + FD);
				fb[s + FD][1] = frag(1, s + FD);
			}
#pragma unroll
			for (int tq = 0; tq < 2; ++tq)
#pragma unroll
				for (int tc = 0; tc < 2; ++tc)
					acc[tq][tc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tq][s], fb[s][tc],
					                                                     s == 0 ? (SAMPLE ? c0[tc] : th[tq]) : acc[tq][tc], 0, 0, 0);
			{
				if (2 * s < LOADS)
					stage_piece(stb, srow0, buf ^ 1, 2 * s);
				if (2 * s + 1 < LOADS)
					stage_piece(stb, srow0, buf ^ 1, 2 * s + 1);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		if (kh) {
#pragma unroll
			for (int s = KSA; s < KSB; ++s) {
				if (s + FD < KSB) {
					fb[s + FD][0] = frag(0, s + FD);
					fb[s + FD][1] = frag(1, s + FD);
				}
#pragma unroll
				for (int tq = 0; tq < 2; ++tq)
#pragma unroll
					for (int tc = 0; tc < 2; ++tc)
						acc[tq][tc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tq][s], fb[s][tc], acc[tq][tc], 0, 0, 0);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		__builtin_amdgcn_s_setprio(0);
		float4* const xp = reinterpret_cast<float4*>(xch + (par * G::PAIRS + qg) * 4096) + lane;
		if (kh) {
#pragma unroll
			for (int j = 0; j < 4; ++j)
				xp[j * 64] = float4{acc[j >> 1][j & 1][0], acc[j >> 1][j & 1][1], acc[j >> 1][j & 1][2], acc[j >> 1][j & 1][3]};
		}
		// tile t+1 landed, the partner's partial sums are visible, everyone is done with this buffer
		wait_vm_then_barrier<0>();
		if (!kh) {
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const float4 v = xp[j * 64];
				acc[j >> 1][j & 1][0] += v.x;
				acc[j >> 1][j & 1][1] += v.y;
				acc[j >> 1][j & 1][2] += v.z;
				acc[j >> 1][j & 1][3] += v.w;
			}
			if constexpr (SAMPLE) {
#pragma unroll
				for (int tc = 0; tc < 2; ++tc)
#pragma unroll
					for (int tq = 0; tq < 2; ++tq)
#pragma unroll
						for (int r = 0; r < 4; ++r)
							smax[tc][tq][r] = __builtin_fmaxf(smax[tc][tq][r], acc[tq][tc][r]);
			} else {
				// queue fills posted at the previous step are visible after this step's barrier: if a
				// queue of the workgroup is half full, every wave empties its own now
				if (since_look == kF16FlushEvery) {
					since_look = 0;
					const uint32_t f = fills[lane & 3];
					if (__builtin_amdgcn_ballot_w64(f >= (uint32_t)QCAP / 2) != 0)
						flush_own();
				}
				epilogue(acc, tile_row0(t), bnv);
				if (wfill >= (uint32_t)QCAP * 3 / 4)
					flush_own();
				if (++since_look == kF16FlushEvery && lane == 0)
					fills[qg] = wfill;
			}
		}
		buf ^= 1;
		par ^= 1;
	}
	if (!kh) {
		if constexpr (SAMPLE) {
#pragma unroll
			for (int tc = 0; tc < 2; ++tc)
#pragma unroll
				for (int tq = 0; tq < 2; ++tq)
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const uint32_t qi = q0 + tq * 16 + 4 * lq + r;
						if (qi < p.m)
							p.sample_out[((size_t)qi * p.n_chunks + chunk) * 32 + tc * 16 + l15] = smax[tc][tq][r];
					}
		} else {
			flush_own();
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-staged tail tile: LDS must outlive it
}

}  // namespace expann
