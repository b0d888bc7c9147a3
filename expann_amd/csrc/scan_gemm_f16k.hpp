// scan_gemm_f16k.hpp -- the fp16 single-product filter of scan_gemm_f16.hpp for 512 < d <= 960
// (d = 768, 832, 960: the reference builds modules for 832 and 960, CMakeLists.txt:137-153).
//
// Beyond d = 512 the fragments of even 32 queries no longer fit one wave's registers (d = 960:
// 240 VGPRs), so the k range is SPLIT over a pair of waves: 8 waves = 4 query groups of 32 x
// 2 k-halves, 128 queries per workgroup, one workgroup per CU.  Wave (g, kh) holds the fragments
// of group g for its half of the k-steps (<= 128 VGPRs), multiplies them with every staged
// 32-row tile and ends a step with a partial 32x32 accumulator; the kh = 1 wave hands its 16
// registers to its partner through LDS (4 KiB per pair, two copies by step parity so that one
// workgroup barrier per step orders both the tile buffers and the exchange), the kh = 0 wave
// adds them and runs the epilogue / candidate queue of scan_gemm_f16.hpp while the kh = 1 wave
// is already multiplying the next tile.  Both waves of a pair sit on the same SIMD (waves w and
// w + 4), so the SIMD's matrix pipe sees all d/16 MFMAs per tile whatever the split.
//
// Sum order: theta' + (first half) and (second half) are accumulated separately and added once;
// the slack gemm_f16_filter_eps(d) covers d roundings at ANY association (see there).
//
// LDS (d = 960): 2 x 60 KiB tiles + 32 KiB exchange + 4 x 22 queue entries.  A 1920- or
// 1664-byte row stride is 128 mod 256, so consecutive rows alternate between the two halves of
// the 256-byte bank row and the XOR swizzle works on groups of 8 chunks with (row >> 1) & 7;
// d = 768 (stride 0 mod 256) swizzles 16 chunks with row & 15 like d = 512.  The k-steps are
// split at a multiple of the swizzle group (24/24, 24/28, 28/32 of 48/52/60): the kh = 0
// waves, which also run the epilogue, take the smaller share.
#pragma once
#include "scan_gemm_f16.hpp"

#ifndef EXPANN_F16K_FD
#define EXPANN_F16K_FD 2
#endif
namespace expann {

template <int D> struct F16kGeom {
	static constexpr int THREADS = 512, WAVES = 8, PAIRS = 4;
	static constexpr int WGQ = 32 * PAIRS;  // queries per workgroup
	static constexpr int TB = 32;           // rows per tile
	static constexpr int ROWB = D * 2, CH = ROWB / 16, KS = D / 16;
	static constexpr bool HALF = ROWB % 256 != 0;
	static constexpr int SWG = HALF ? 8 : 16;       // chunks per swizzle group
	static constexpr int NA = SWG / 2;              // k-steps per swizzle group
	static constexpr int KSA = (KS / 2) / NA * NA;  // k-steps of the kh = 0 waves
	static constexpr int KSB = KS - KSA;            // ... of the kh = 1 waves
	static constexpr int TILE_BYTES = TB * ROWB;
	static constexpr int NBUF = 2;
	static constexpr int XCH_BYTES = 2 * PAIRS * 4096;
	static constexpr int FIXED = NBUF * TILE_BYTES + NBUF * 256 + XCH_BYTES + WGQ * 4 + 32;
	static constexpr int QROOM = (160 * 1024 - FIXED) / (PAIRS * kF16EntryBytes);
	static constexpr int QCAP = QROOM < 56 ? QROOM : 56;
	static constexpr int LDS_BYTES = FIXED + PAIRS * QCAP * kF16EntryBytes;
	static_assert(ROWB % 128 == 0 && KSB >= KSA && KSB - KSA <= NA && QCAP >= 16, "geometry");
};

template <int D, bool SAMPLE>
__global__ __launch_bounds__(512, 2) void scan_gemm_f16k_kernel(GemmF16Params p) {
	static_assert(D == 768 || D == 832 || D == 960, "built for d = 768, 832, 960");
	static_assert(SAMPLE, "round 3: the 32 x 32 x 16 stream serves the sampled pass only; the full scan is scan_gemm_f16kx");
	using G = F16kGeom<D>;
	constexpr int THREADS = G::THREADS, WGQ = G::WGQ, TB = G::TB;
	constexpr int ROWB = G::ROWB, CH = G::CH, KSA = G::KSA, KSB = G::KSB, NA = G::NA, SWG = G::SWG;
	constexpr int TILE_BYTES = G::TILE_BYTES;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int qg = wave & 3, kh = wave >> 2;
	const int h = lane >> 5, r31 = lane & 31;
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
	// this wave's k-steps: kh * KSA + s
	f16x8 a[KSB];
	{
		uint32_t qi = q0 + r31;
		if (qi >= p.m)
			qi = p.m - 1;
		const f16x8* src =
		    reinterpret_cast<const f16x8*>((const unsigned char*)p.queries_f16 + (size_t)qi * ROWB) + 2 * kh * KSA;
#pragma unroll
		for (int s = 0; s < KSB; ++s)
			a[s] = src[2 * s + h];  // (kh = 0: the last KSB - KSA are loaded but never multiplied)
	}
	// the accumulators start at zero; th holds the running class maxima of g
	f32x16 th;
#pragma unroll
	for (int reg = 0; reg < 16; ++reg)
		th[reg] = -__builtin_inff();
#pragma unroll
	for (int s = 0; s < KSB; ++s)
		asm volatile("" : "+v"(a[s]));  // in registers before the first stage load (see scan_gemm_f16_kernel)
	asm volatile("" : "+v"(th));

	auto swz = [](uint32_t r) -> uint32_t { return G::HALF ? ((r >> 1) & 7) : (r & 15); };
	// per-lane LDS offset of local k-step s: chunk 2 (kh KSA + s) + h = group base + (2 (s % NA) + h),
	// the XOR with the row's swizzle stays inside the group: NA registers + immediates
	uint32_t aoff[NA];
#pragma unroll
	for (int j = 0; j < NA; ++j)
		aoff[j] = r31 * ROWB + kh * (2 * KSA * 16) + (((2 * j + h) ^ swz(r31)) * 16);

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
	static_assert(LOADS <= KSA, "one stage piece per k-step");
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


	{
		uint32_t row0;
		const unsigned char* tb = stage_src(t0, row0);
#pragma unroll
		for (int i = 0; i < LOADS; ++i)
			stage_piece(tb, row0, 0, i);
	}
	wait_vm_then_barrier<0>();  // tile t0 landed

	f32x16 zero16;
#pragma unroll
	for (int e = 0; e < 16; ++e)
		zero16[e] = 0.0f;
	f32x16 acc;
	int buf = 0;
	uint32_t par = 0;
	for (uint32_t t = t0; t < t1; ++t) {
		const uint32_t boff = (uint32_t)buf * TILE_BYTES;
		auto frag = [&](int s) -> f16x8 {
			return *reinterpret_cast<const f16x8*>(smem + (boff + aoff[s % NA]) + (s / NA) * (SWG * 16));
		};
		uint32_t srow0;
		const unsigned char* stb = stage_src(t + 1, srow0);
		constexpr int FD = EXPANN_F16K_FD;  // k-steps of fragment read-ahead
		f16x8 fb[KSB];
#pragma unroll
		for (int s = 0; s < FD; ++s)
			fb[s] = frag(s);
		// the bn' of this tile: its slot is re-staged during the NEXT step, so it is read now
		const float bnv = reinterpret_cast<const float*>(bn_slots + buf * 256)[r31];
		__builtin_amdgcn_s_setprio(1);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int s = 0; s < KSA; ++s) {
			if (s + FD < KSA)
				fb[s + FD] = frag(s + FD);
			else if (s + FD < KSB && kh)
				fb[s + FD] = frag(s + FD);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], fb[s], s == 0 ? zero16 : acc, 0, 0, 0);
			if (s < LOADS)
				stage_piece(stb, srow0, buf ^ 1, s);
			__builtin_amdgcn_sched_barrier(0);
		}
		if (kh) {
#pragma unroll
			for (int s = KSA; s < KSB; ++s) {
				if (s + FD < KSB)
					fb[s + FD] = frag(s + FD);
				acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], fb[s], acc, 0, 0, 0);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		__builtin_amdgcn_s_setprio(0);
		float4* const xp = reinterpret_cast<float4*>(xch + (par * G::PAIRS + qg) * 4096) + lane;
		if (kh) {
#pragma unroll
			for (int j = 0; j < 4; ++j)
				xp[j * 64] = float4{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
		}
		// tile t+1 landed, the partner's partial sums are visible, everyone is done with this buffer
		wait_vm_then_barrier<0>();
		if (!kh) {
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const float4 v = xp[j * 64];
				acc[4 * j] += v.x;
				acc[4 * j + 1] += v.y;
				acc[4 * j + 2] += v.z;
				acc[4 * j + 3] += v.w;
			}
#pragma unroll
			for (int reg = 0; reg < 16; ++reg)
				th[reg] = __builtin_fmaxf(th[reg], acc[reg] - bnv);
		}
		buf ^= 1;
		par ^= 1;
	}
	if (!kh) {
#pragma unroll
		for (int reg = 0; reg < 16; ++reg) {
			const uint32_t qi = q0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
			if (qi < p.m)
				p.sample_out[((size_t)qi * p.n_chunks + chunk) * 32 + r31] = th[reg];
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-staged tail tile: LDS must outlive it
}

}  // namespace expann
