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

// (the 32 x 32 x 16 kernel of rounds 1-2 over this geometry is gone: scan_gemm_f16kx.hpp is the kernel, full
// scan and sampled pass)

}  // namespace expann
