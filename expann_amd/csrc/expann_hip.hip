// expann_hip.hip -- the C ABI (include/expann_hip.h) over the gfx950 kernels.
//
// Host side of the hot path: owns device memory, plans the threshold levels, launches the
// scan / select / merge kernels on HIP streams.  No CPU fallback lives here: every compute
// entry point needs a HIP device and fails loudly without one.
#include "../../include/expann_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "scan_f32.hpp"
#include "scan_gemm_bf16.hpp"
#include "scan_gemm_f16.hpp"
#include "scan_gemm_f16k.hpp"
#include "scan_gemm_f16x.hpp"
#include "scan_gemm_f16y.hpp"
#include "scan_gemm_f16kx.hpp"
#include "scan_direct_f16.hpp"
#include "scan_gemm_i8.hpp"
#include "scan_gemm_i8q.hpp"
#include "scan_gemm_i8x.hpp"
#include "scan_gemm_i8w.hpp"
#include "scan_int8.hpp"
#include "score_ids.hpp"
#include "select.hpp"

using namespace expann;

namespace expann {
thread_local std::string g_create_error;
}

namespace {

constexpr uint32_t kMaxCap = 16384;      // keys per query that fit the select kernel's LDS
constexpr size_t kMaxQueriesPerPass = 32768;
constexpr int kEventPairs = 64;
constexpr uint32_t kAsyncRing = 256;  // outstanding searches of the deferred-check mode

struct Level {
	uint32_t n_groups_sel;
	uint32_t group_stride;
	uint32_t classmin_blocks = 0;  // > 0: level 0 keeps class minima (scan_f32.hpp), this many workgroups
};

}  // namespace

struct expann_index {
	int dim = 0, dtype = 0, metric = 0, device = 0;
	size_t elem = 4;    // bytes per stored element
	size_t q_elem = 4;  // bytes per query element at the ABI (f32 for F32/U8 rows, int8 for I8)
	int int_mode = -1;  // IntMode for 8-bit rows
	GrowPtr<uint8_t> d_q8;  // U8 rows: queries truncated to uint8
	std::vector<unsigned char> staging;  // store_vector() rows until build()
	size_t n_staged = 0;
	void* d_base = nullptr;
	bool owns_base = false;
	size_t n = 0;
	uint64_t id_offset = 0;
	hipStream_t stream = nullptr;
	// workspace (grown on demand)
	GrowPtr<uint64_t> d_cand;
	DevPtr<uint32_t> d_cnt;          // [m_alloc]
	DevPtr<float> d_tau[2];
	DevPtr<uint32_t> d_tau_row[2];
	size_t m_alloc = 0;
	DevPtr<uint32_t> d_overflow;     // [4]: overflow count
	unsigned long long* d_total = nullptr;
	PinPtr<uint32_t> h_flags;        // pinned [4]
	// latency mode (expann_search with few queries): pinned staging [queries | ids | dists]; the
	// select kernels store the results straight into it, so a search costs one async H2D copy of
	// the queries, the kernels, and the flag read-back -- one host sync, no pageable copies
	PinPtr<void> h_pin;
	bool q_in_pinned_host = false;  // (expann_search -> search_pass: d_queries is pinned host memory)
	DevPtr<uint32_t> d_ticket;  // last-workgroup counter of sample_direct_f16_kernel (0 between launches)
	size_t h_pin_bytes = 0;
	DevPtr<float> d_bnorm;           // ||b||^2 (1-eps) per row (GEMM-form scan), built lazily
	DevPtr<float> d_bnorm_bf;        // same with the bf16x3 slack
	DevPtr<float> d_bnmax;           // [2]: max of d_bnorm, max of d_bnorm_bf
	DevPtr<void> d_base_split;       // [n][2][dim] bf16 hi/lo planes (bf16x3 GEMM form), lazily
	// fp32 index whose values are all integers in [0, 255] (SIFT): searches with integer queries
	// go through an internal uint8 engine -- exact integer scores, equal to the fp32 ones bit for
	// bit while d * 255^2 < 2^24 -- at the 8-bit kernels' speed
	int u8_exact = 0;                // 0 not examined, 1 yes (shadow built), -1 no
	expann_index* u8_shadow = nullptr;
	DevPtr<void> d_base_u8;
	bool strict_u8 = false;          // (on the shadow) non-8-bit / fractional queries: hand back, no error
	double prof_extra_ms = 0;        // scan time absorbed from the shadow
	long opt_u8_exact = 1;
	void* d_base_i8q = nullptr;      // padded int8 copy of an 8-bit index (uint8 rows ^ 0x80) or alias of d_base
	bool base_i8q_owned = false;
	DevPtr<int> d_bp_i8q;            // [n padded] floor(bias/2), scan_gemm_i8q.hpp
	DevPtr<void> d_log;              // per-wave hit logs of scan_gemm_f16x ([n_logs][log_cap] x 16 B)
	DevPtr<uint32_t> d_log_cnt;      // [n_logs]
	DevPtr<uint32_t> d_work_ctr;     // [8][16]: per-XCD item counters of the persistent scan launch
	size_t log_bytes = 0, log_cnt_n = 0;
	struct {                         // scatter_log_kernel of the scan just launched (run after its timing event)
		uint32_t n_logs = 0, log_cap = 0, cap = 0, n_chunks = 0, n_qtiles = 0, xcd_map = 0, m = 0;
		const float* theta = nullptr;  // fp16 logs: entries hold bn' - acc, the gather adds theta' and scales (GatherLogParams)
		float key_mul = 0.0f;
		const int *i_thp = nullptr, *i_bias = nullptr, *i_qself = nullptr;  // 8-bit logs: raw accumulators, scored by the gather
		int i_mode = 0;
	} pending_scatter;
	GrowPtr<float> d_sample;         // [m][n_chunks][32] class maxima of the fp16 / int8 sample pass
	GrowPtr<void> d_q_split;         // [m][2][dim] bf16 (or [m][dim] fp16)
	DevPtr<void> d_base_f16;         // [n][dim] fp16 rows scaled by f16_scale (fp16 GEMM form), lazily
	DevPtr<float> d_bnorm_f16;       // [n] ||b||^2 (1-eps) - abs |b|
	DevPtr<float> d_bns_f16;         // [n] ||b||^2 (1+eps) + abs |b|: the sampled pass's row term
	DevPtr<float> d_qnrm;            // [m_alloc] ||q||^2 (fp16 form, inner product: of the filter-side query c_q q)
	DevPtr<float> d_qscale;          // [m_alloc] c_q: the power of two the fp16 filter sees a query multiplied by (inner product; L2: 1)
	float f16_scale = 0.0f;          // power of two; 0 = not built
	float f16_bnmax = 0.0f;          // max ||b||^2 (host copy lives in d_bnmax[2])
	DevPtr<float> d_theta;           // [m_alloc] (int32 thetas for the 8-bit GEMM form)
	DevPtr<int> d_bias_i;            // [n] sum b^2 per row (8-bit L2 GEMM form), built lazily
	DevPtr<int> d_qself;             // [m_alloc]
	GrowPtr<void> d_q;               // host-API staging
	GrowPtr<uint64_t> d_ids;
	GrowPtr<float> d_dists;
	// options
	long opt_query_tile = 0, opt_cand_capacity = 0, opt_sample_ratio = 32;
	long opt_scan_chunks = 0;
	// deferred check (expann_sync): flag blocks of the outstanding searches, read back into a ring
	long opt_async = 0;
	bool host_call = false;          // (expann_search: always the synchronous path)
	PinPtr<uint32_t> h_flag_ring;    // pinned [kAsyncRing][8]
	uint32_t async_pending = 0;
	hipStream_t async_stream = nullptr;
	long opt_xcd_tolerance = 3;  // % of modelled cost given up for an XCD-aligned chunk count
	long opt_latency_mode = 1;  // few queries from host buffers: results land in pinned memory
	long opt_debug = 0;
	long opt_sample_frac = 0;        // the sampled pass reads 1/frac of the rows; 0 = by k (sample_frac_for)
	long opt_sample_run = 1;         // consecutive 64-row tiles per sampled stretch.  1: the sample
	                                 // is spread as finely as tiles allow -- with rows stored by
	                                 // cluster, stretches of 16 tiles skipped whole clusters and the
	                                 // full scan drowned in candidates (90 k instead of 2.8 M QPS on
	                                 // 1000 contiguous clusters); on iid rows the two differ by < 1 %
	long opt_sample_pass = 1;        // fp16 form: one sampled class-maxima pass instead of the level ladder
	long opt_tail_chunks = 1;        // scan_gemm_f16x: the last round's row chunks three times finer (pick_tail_chunks)
	long opt_persist = 1;            // scan_gemm_f16x: resident workgroups pull (query tile, row chunk) items per XCD
	long opt_ip_rescale = 1;         // fp16 form, inner product: the filter sees each query times a power of two (f16_query_prep_kernel)
	long opt_scan_kernel = 0;        // 0 auto, 1 direct (scan_filter), 2 GEMM form on fp32 / int8
	                                 // MFMA, 3 GEMM form on bf16 MFMA with the 3-term split
	// profiling
	bool profiling = false;
	hipEvent_t ev[kEventPairs][2];
	bool ev_created = false;
	int ev_used = 0;
	expann_profile prof{};
	mutable std::string err;

	int fail(int code, const std::string& msg) const {
		err = msg;
		return code;
	}
};

namespace expann {
int num_cus(int device) {
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
		return 256;
	return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}
}  // namespace expann

namespace {

// ---- kernel dispatch ---------------------------------------------------------------
using ScanFn = void (*)(ScanParams);
struct ScanVariant {
	int d, tq;
	bool ip;
	ScanFn fn;
	const char* name;
};
#define SCAN_V(D, TQ)                                                                      \
	{D, TQ, false, scan_filter_f32_kernel<D, TQ, false>, "scan_filter_f32<" #D "," #TQ ",L2>"}, \
	{D, TQ, true, scan_filter_f32_kernel<D, TQ, true>, "scan_filter_f32<" #D "," #TQ ",IP>"}
const ScanVariant kScanF32[] = {
    SCAN_V(64, 1),  SCAN_V(64, 4),  SCAN_V(64, 16), SCAN_V(128, 1), SCAN_V(128, 2),
    SCAN_V(128, 4), SCAN_V(128, 8), SCAN_V(128, 16), SCAN_V(256, 1), SCAN_V(256, 4),
    SCAN_V(256, 8), SCAN_V(512, 1), SCAN_V(512, 4),  SCAN_V(768, 1), SCAN_V(768, 2),
    SCAN_V(832, 1), SCAN_V(832, 2), SCAN_V(960, 1),  SCAN_V(960, 2), SCAN_V(1024, 1),
    SCAN_V(1024, 2)};
#undef SCAN_V

const ScanVariant* pick_scan_f32(int d, bool ip, size_t m, long forced_tq) {
	const ScanVariant* best = nullptr;
	for (const auto& v : kScanF32) {
		if (v.d != d || v.ip != ip)
			continue;
		if (forced_tq > 0) {
			if (v.tq == forced_tq)
				return &v;
			continue;
		}
		if (!best) {
			best = &v;
			continue;
		}
		// smallest tile that covers m, else the largest tile
		const bool v_covers = (size_t)v.tq >= m, b_covers = (size_t)best->tq >= m;
		if (v_covers && (!b_covers || v.tq < best->tq))
			best = &v;
		else if (!v_covers && !b_covers && v.tq > best->tq)
			best = &v;
	}
	return best;
}

using ScoreFn = void (*)(ScoreIdsParams);
struct ScoreVariant {
	int d;
	bool ip;
	ScoreFn fn;
};
#define SCORE_V(D) {D, false, score_ids_f32_kernel<D, false>}, {D, true, score_ids_f32_kernel<D, true>}
const ScoreVariant kScoreF32[] = {SCORE_V(64),  SCORE_V(128), SCORE_V(256), SCORE_V(512),
                                  SCORE_V(768), SCORE_V(832), SCORE_V(960), SCORE_V(1024)};
#undef SCORE_V

// 8-bit rows
struct ScanI8Variant {
	int d, tq, mode;
	ScanFn fn;
	const char* name;
};
#define SCAN_I8_M(D, TQ, MODE, MN) {D, TQ, MODE, scan_filter_i8_kernel<D, TQ, MODE>, "scan_filter_i8<" #D "," #TQ "," MN ">"}
#define SCAN_I8(D, TQ) SCAN_I8_M(D, TQ, kU8L2, "U8L2"), SCAN_I8_M(D, TQ, kI8L2, "I8L2"), \
	SCAN_I8_M(D, TQ, kI8L2Ref, "I8L2REF"), SCAN_I8_M(D, TQ, kI8IP, "I8IP")
// int16 rows: d ELEMENTS, kernels instantiated on 2 d bytes per row
#define SCAN_I16(DE, TQ) {DE, TQ, kI16L2Ref, scan_filter_i8_kernel<2 * DE, TQ, kI16L2Ref>, \
	"scan_filter_i16<" #DE "," #TQ ",L2REF>"}
const ScanI8Variant kScanI8[] = {SCAN_I8(64, 1),  SCAN_I8(64, 16), SCAN_I8(128, 1), SCAN_I8(128, 4),
                                 SCAN_I8(128, 16), SCAN_I8(256, 1), SCAN_I8(256, 8), SCAN_I8(768, 1),
                                 SCAN_I8(768, 4),  SCAN_I8(768, 8), SCAN_I8(832, 1), SCAN_I8(832, 4),
                                 SCAN_I8(960, 1),  SCAN_I8(960, 4),
                                 SCAN_I16(64, 1),  SCAN_I16(64, 4), SCAN_I16(64, 16), SCAN_I16(128, 1),
                                 SCAN_I16(128, 8)};
#undef SCAN_I16
#undef SCAN_I8
#undef SCAN_I8_M

const ScanI8Variant* pick_scan_i8(int d, int mode, size_t m, long forced_tq) {
	const ScanI8Variant* best = nullptr;
	for (const auto& v : kScanI8) {
		if (v.d != d || v.mode != mode)
			continue;
		if (forced_tq > 0) {
			if (v.tq == forced_tq)
				return &v;
			continue;
		}
		if (!best) {
			best = &v;
			continue;
		}
		const bool v_covers = (size_t)v.tq >= m, b_covers = (size_t)best->tq >= m;
		if (v_covers && (!b_covers || v.tq < best->tq))
			best = &v;
		else if (!v_covers && !b_covers && v.tq > best->tq)
			best = &v;
	}
	return best;
}

using ScoreI8Fn = void (*)(ScoreIdsI8Params);
struct ScoreI8Variant {
	int d, mode;
	ScoreI8Fn fn;
};
#define SCORE_I8(D) {D, kU8L2, score_ids_i8_kernel<D, kU8L2>}, {D, kI8L2, score_ids_i8_kernel<D, kI8L2>}, \
	{D, kI8L2Ref, score_ids_i8_kernel<D, kI8L2Ref>}, {D, kI8IP, score_ids_i8_kernel<D, kI8IP>}
const ScoreI8Variant kScoreI8[] = {SCORE_I8(64), SCORE_I8(128), SCORE_I8(256), SCORE_I8(768), SCORE_I8(832), SCORE_I8(960),
                                   {64, kI16L2Ref, score_ids_i8_kernel<128, kI16L2Ref>},
                                   {128, kI16L2Ref, score_ids_i8_kernel<256, kI16L2Ref>}};
#undef SCORE_I8

uint32_t pow2ceil(uint32_t x) {
	uint32_t p = 1;
	while (p < x)
		p <<= 1;
	return p;
}


// Threshold levels: level 0 keeps every row of a small strided sample (tau = +inf), each
// later level scans `ratio` times more rows with tau = k-th best score of the level before,
// the last level scans every row.  Expected survivors per query and level ~ ratio * k.
std::vector<Level> plan_levels(size_t n, size_t k, uint32_t cap, long ratio_opt) {
	const uint32_t n_groups = (uint32_t)((n + kRowsPerGroup - 1) / kRowsPerGroup);
	uint32_t s0 = std::max<uint32_t>(64, (uint32_t)((4 * k + kRowsPerGroup - 1) / kRowsPerGroup));
	s0 = std::min(s0, cap / kRowsPerGroup);
	std::vector<Level> lv;
	if (n_groups <= s0) {
		lv.push_back({n_groups, 1});
		return lv;
	}
	// survivors ~ ratio*k must stay well inside cap
	double max_ratio = std::max(2.0, (double)cap / (6.0 * (double)k));
	double ratio = std::min<double>((double)ratio_opt, max_ratio);
	if (ratio < 2.0)
		ratio = 2.0;
	const double span = (double)n_groups / (double)s0;
	int steps = (int)std::ceil(std::log(span) / std::log(ratio) - 1e-9);
	if (steps < 1)
		steps = 1;
	const double r = std::pow(span, 1.0 / steps);
	double cur = s0;
	lv.push_back({s0, n_groups / s0});
	for (int i = 1; i < steps; ++i) {
		cur *= r;
		uint32_t sel = (uint32_t)std::llround(cur);
		sel = std::min(sel, n_groups);
		lv.push_back({sel, std::max<uint32_t>(1, n_groups / sel)});
	}
	lv.push_back({n_groups, 1});
	return lv;
}

// grow one of the handle's shared buffers; deferred searches still in flight read the old block, so
// their stream is drained first (hipFree happens to synchronise the device: stated, not implied)
template <typename T> hipError_t grow_ws(expann_index* h, GrowPtr<T>& b, size_t need) {
	if (h->async_pending > 0 && (need > b.bytes || !b.p)) {
		const hipError_t e = hipStreamSynchronize(h->async_stream);
		if (e != hipSuccess)
			return e;
	}
	return b.ensure(need);
}

int ensure_workspace(expann_index* h, size_t m, uint32_t cap) {
	if (h->async_pending > 0 && m > h->m_alloc)  // (the per-query arrays below: as grow_ws)
		HIP_TRY(h, hipStreamSynchronize(h->async_stream));
	if (m > h->m_alloc) {  // the per-query arrays grow together
		h->m_alloc = 0;
		for (DevPtr<float>* b : {std::addressof(h->d_qnrm), std::addressof(h->d_qscale), std::addressof(h->d_theta),
		                         std::addressof(h->d_tau[0]), std::addressof(h->d_tau[1])}) {
			b->reset();
			HIP_TRY(h, hipMalloc(&*b, sizeof(float) * m));
		}
		for (DevPtr<uint32_t>* b : {std::addressof(h->d_cnt), std::addressof(h->d_tau_row[0]),
		                            std::addressof(h->d_tau_row[1])}) {
			b->reset();
			HIP_TRY(h, hipMalloc(&*b, sizeof(uint32_t) * m));
		}
		h->d_qself.reset();
		HIP_TRY(h, hipMalloc(&h->d_qself, sizeof(int) * m));
		h->m_alloc = m;
	}
	HIP_TRY(h, grow_ws(h, h->d_cand, sizeof(uint64_t) * m * (size_t)cap));
	if (!h->d_overflow) {
		// flags [4 x u32] and statistics [2 x u64] share one allocation: one memset, one read-back
		HIP_TRY(h, hipMalloc(&h->d_overflow, sizeof(uint32_t) * 4 + sizeof(unsigned long long) * 2));
		// every word starts at zero: the paths clear only the words they use, and expann_sync /
		// the strict uint8 check read words 1..3 of searches that never wrote them
		HIP_TRY(h, hipMemset(h->d_overflow, 0, sizeof(uint32_t) * 4 + sizeof(unsigned long long) * 2));
		h->d_total = reinterpret_cast<unsigned long long*>(h->d_overflow + 4);
		HIP_TRY(h, hipHostMalloc((void**)&h->h_flags, sizeof(uint32_t) * 8, 0));
		HIP_TRY(h, hipHostMalloc((void**)&h->h_flag_ring, sizeof(uint32_t) * 8 * kAsyncRing, 0));
		HIP_TRY(h, hipMalloc(&h->d_ticket, sizeof(uint32_t)));
		HIP_TRY(h, hipMemset(h->d_ticket, 0, sizeof(uint32_t)));
		HIP_TRY(h, hipDeviceSynchronize());  // (once per handle: the counter is zero before any stream uses it)
	}
	return EXPANN_OK;
}

// ---- GEMM-form (MFMA) scan: dims it is built for ----------------------------------------
using GemmFn = void (*)(GemmScanParams);
using NormFn = void (*)(const float*, uint32_t, float, float*);
using ThetaFn = void (*)(const float*, uint32_t, const float*, float, float*);
struct GemmVariant {
	int d;
	GemmFn scan;
	NormFn norms;
	ThetaFn theta;
	const char* name;
};
// d = 64 / 128: the row / query terms of the bf16x3 fallback form (gemm_terms.hpp); no scan kernel of its own
const GemmVariant kGemmF32[] = {
    {64, nullptr, row_norms_kernel<64>, query_theta_kernel<64>, "-"},
    {128, nullptr, row_norms_kernel<128>, query_theta_kernel<128>, "-"}};
// dims that only the fp16 form covers: a placeholder without kernels (scan == nullptr) keeps the
// GEMM branch of the level loop alive; without the fp16 form it counts as "no GEMM form"
const GemmVariant kGemmF16Only[] = {{256, nullptr, nullptr, nullptr, "-"}, {512, nullptr, nullptr, nullptr, "-"},
                                    {64, nullptr, nullptr, nullptr, "-"},  {128, nullptr, nullptr, nullptr, "-"},
                                    {768, nullptr, nullptr, nullptr, "-"}, {832, nullptr, nullptr, nullptr, "-"},
                                    {960, nullptr, nullptr, nullptr, "-"}};

using GemmBf16Fn = void (*)(GemmBf16Params);
struct GemmBf16Variant {
	int d;
	GemmBf16Fn scan;
	const char* name;
};
const GemmBf16Variant kGemmBf16[] = {{64, scan_gemm_bf16x3_kernel<64>, "scan_gemm_bf16x3<64>"},
                                     {128, scan_gemm_bf16x3_kernel<128>, "scan_gemm_bf16x3<128>"}};

using GemmF16Fn = void (*)(GemmF16Params);
using SqnormFn = void (*)(const float*, uint32_t, float*);
using F16PrepFn = void (*)(const float*, uint32_t, float, _Float16*, float*, uint32_t*, float*, uint32_t*, const float*, float*);
struct GemmF16Variant {
	int d;
	GemmF16Fn scan;
	GemmF16Fn sample;
	SqnormFn sqnorm;
	F16PrepFn prep;
	const char* name;
	int tb, wgq, threads, wg_per_cu, lds;  // rows per tile, queries per workgroup, launch geometry
	int hit_log;                           // scan writes per-wave hit logs (scan_gemm_f16x.hpp) + scatter_log_kernel
};
// the 16x16x32 form of the full scan and of the sampled pass (scan_gemm_f16x.hpp)
#define F16X_V(D)                                                                                  \
	{D, scan_gemm_f16x_kernel<D, false>, scan_gemm_f16x_kernel<D, true>, sqnorm_kernel<D>,          \
	 f16_query_prep_kernel<D>, "scan_gemm_f16x<" #D ", false>", kF16TB, kF16TQ, kF16Threads,         \
	 f16x_wg_per_cu<D>(), gemm_f16x_lds_bytes<D>(), 1}
// (d = 64 with two workgroups per CU measured slower than scan_gemm_f16_kernel<64>'s three: 6.55 M vs
// 6.83 M QPS at 1 M rows -- two k-steps per column leave too little MFMA per step; hence three here too)
// d = 256 / 512: the 8-waves-per-tile geometry on 16x16x32 (scan_gemm_f16y.hpp), hits appended directly
#define F16Y_V(D)                                                                                  \
	{D, scan_gemm_f16y_kernel<D, false>, scan_gemm_f16y_kernel<D, true>, sqnorm_kernel<D>, f16_query_prep_kernel<D>, \
	 "scan_gemm_f16y<" #D ", false>", kF16TB, F16Geom<D>::WGQ, F16Geom<D>::THREADS, F16Geom<D>::WG_PER_CU, \
	 gemm_f16_lds_bytes<D>(), 0}
// 512 < d <= 960: the k-split geometry on 16x16x32 (scan_gemm_f16kx.hpp)
#define F16KX_V(D)                                                                                 \
	{D, scan_gemm_f16kx_kernel<D, false>, scan_gemm_f16kx_kernel<D, true>, sqnorm_kernel<D>, f16_query_prep_kernel<D>, \
	 "scan_gemm_f16kx<" #D ", false>", F16kGeom<D>::TB, F16kGeom<D>::WGQ, F16kGeom<D>::THREADS, 1,     \
	 F16kGeom<D>::LDS_BYTES, 0}
const GemmF16Variant kGemmF16X[] = {F16X_V(64), F16X_V(128), F16Y_V(256), F16Y_V(512), F16KX_V(768), F16KX_V(832), F16KX_V(960)};
#undef F16KX_V
#undef F16Y_V
#undef F16X_V
// the same with the run-time ablation switches compiled in ("debug" option != 0)
const GemmF16Variant kGemmF16XDbg[] = {{128, scan_gemm_f16x_kernel<128, false, 1>, scan_gemm_f16x_kernel<128, true>,
                                        sqnorm_kernel<128>, f16_query_prep_kernel<128>, "scan_gemm_f16x<128, false>",
                                        kF16TB, kF16TQ, kF16Threads, 2, gemm_f16_lds_bytes<128>(), 1}};
inline bool f16_choice(long opt) { return opt == 0 || opt == 4 || opt == 6; }

// a handful of queries: the same filter streamed from HBM without the matrix cores
// (scan_direct_f16.hpp); tq = queries per pass
using SampleDirectFn = void (*)(SampleDirectParams);
struct DirectF16Variant {
	int d, tq, rps;
	GemmF16Fn fn;
	const char* name;
	SampleDirectFn sample;  // fused sampled pass + thresholds (nullptr: the MFMA SAMPLE pass serves)
	int cpw;                // its classes per workgroup
};
#define DF16_V(D, TQ) {D, TQ, DirectF16Geom<D>::RPS, scan_direct_f16_kernel<D, TQ>, "scan_direct_f16<" #D ", " #TQ ">", \
	sample_direct_f16_kernel<D, TQ>, DirectF16Geom<D>::RW * (kBlock / 64)}
#define DF16_VN(D, TQ) {D, TQ, DirectF16Geom<D>::RPS, scan_direct_f16_kernel<D, TQ>, "scan_direct_f16<" #D ", " #TQ ">", \
	nullptr, 0}
const DirectF16Variant kDirectF16[] = {DF16_V(64, 2),  DF16_V(64, 4),  DF16_V(128, 2), DF16_V(128, 4),
                                       DF16_V(256, 2), DF16_V(256, 4), DF16_V(512, 1), DF16_V(512, 2),
                                       DF16_VN(768, 1), DF16_VN(832, 1), DF16_VN(960, 1)};
#undef DF16_V
#undef DF16_VN
// the variant for m queries: ONE pass (measured at 1M x d128: a second pass, or 8 queries per
// pass -- VALU-bound --, loses to the MFMA form), else the other paths take over
const DirectF16Variant* pick_direct_f16(int d, size_t m) {
	const DirectF16Variant* best = nullptr;
	for (const auto& v : kDirectF16) {
		if (v.d != d)
			continue;
		if (!best || ((size_t)best->tq < m && v.tq > best->tq) || ((size_t)v.tq >= m && v.tq < best->tq))
			best = &v;
	}
	return (best && m <= (size_t)best->tq) ? best : nullptr;
}

// fp16 copy of the base (scaled by a power of two), its slack-adjusted norms, max norm
int ensure_f16(expann_index* h, const GemmF16Variant* gf, hipStream_t st) {
	if (h->d_base_f16 || h->f16_scale < 0.0f)
		return EXPANN_OK;
	const size_t nv = h->n * (size_t)h->dim;
	DevBuf tmp, nrm;
	HIP_TRY(h, tmp.alloc(sizeof(float)));
	HIP_TRY(h, nrm.alloc(sizeof(float) * h->n));
	HIP_TRY(h, hipMemsetAsync(tmp.p, 0, sizeof(uint32_t), st));
	hipLaunchKernelGGL(maxabs_bits_kernel, dim3(2048), dim3(kBlock), 0, st, (const float*)h->d_base, nv,
	                   tmp.as<uint32_t>());
	float maxabs = 0.0f;  // (a NaN pattern stays a NaN and fails the range check below)
	HIP_TRY(h, hipMemcpyAsync(&maxabs, tmp.p, sizeof(float), hipMemcpyDeviceToHost, st));
	HIP_TRY(h, hipStreamSynchronize(st));
	float scale = 1.0f;
	if (maxabs > 0.0f && std::isfinite(maxabs)) {
		int e = (int)std::floor(std::log2(32768.0 / (double)maxabs));
		e = std::max(-40, std::min(40, e));  // s^2 and every scaled norm stay finite in fp32
		scale = std::ldexp(1.0f, e);
	}
	if (!(maxabs * scale <= 32768.0f)) {  // non-finite or astronomically large values
		h->f16_scale = -1.0f;
		return EXPANN_OK;
	}
	// padded to whole 64-row tiles: zero rows whose bn' is NaN (never a candidate)
	const size_t n_pad = (h->n + kF16TB - 1) / kF16TB * kF16TB;
	HIP_TRY(h, hipMalloc(&h->d_base_f16, n_pad * h->dim * 2));
	HIP_TRY(h, hipMalloc(&h->d_bnorm_f16, sizeof(float) * n_pad));
	HIP_TRY(h, hipMalloc(&h->d_bns_f16, sizeof(float) * n_pad));
	HIP_TRY(h, hipMemsetAsync(h->d_base_f16, 0, n_pad * h->dim * 2, st));
	HIP_TRY(h, hipMemsetAsync(h->d_bnorm_f16, 0xFF, sizeof(float) * n_pad, st));  // 0xFFFFFFFF: a NaN
	HIP_TRY(h, hipMemsetAsync(h->d_bns_f16, 0xFF, sizeof(float) * n_pad, st));
	if (!h->d_bnmax)
		HIP_TRY(h, hipMalloc(&h->d_bnmax, 4 * sizeof(float)));
	hipLaunchKernelGGL(convert_f16_kernel, dim3((uint32_t)std::min<size_t>((nv + kBlock - 1) / kBlock, 8192)),
	                   dim3(kBlock), 0, st, (const float*)h->d_base, nv, scale, h->d_base_f16.as<_Float16>(),
	                   (uint32_t*)nullptr);
	const uint32_t blocks16 = (uint32_t)((h->n + kRowsPerGroup - 1) / kRowsPerGroup);
	hipLaunchKernelGGL(gf->sqnorm, dim3(blocks16), dim3(kBlock), 0, st, (const float*)h->d_base,
	                   (uint32_t)h->n, nrm.as<float>());
	hipLaunchKernelGGL(max_f32_kernel, dim3(1), dim3(1024), 0, st, (const float*)nrm.p, h->n,
	                   h->d_bnmax + 2);
	const float abs_coef = std::ldexp(1.0f, -24) / scale * std::sqrt((float)h->dim);
	const int ipm = h->metric == EXPANN_METRIC_IP ? 1 : 0;
	hipLaunchKernelGGL(f16_terms_kernel, dim3((uint32_t)((h->n + kBlock - 1) / kBlock)), dim3(kBlock),
	                   0, st, (const float*)nrm.p, (uint32_t)h->n, gemm_f16_filter_eps(h->dim), abs_coef,
	                   (const float*)nullptr, 0.5f * scale * scale, h->d_bnorm_f16, ipm, (const float*)nullptr);
	hipLaunchKernelGGL(f16_terms_kernel, dim3((uint32_t)((h->n + kBlock - 1) / kBlock)), dim3(kBlock),
	                   0, st, (const float*)nrm.p, (uint32_t)h->n, -gemm_f16_filter_eps(h->dim), -abs_coef,
	                   (const float*)nullptr, 0.5f * scale * scale, h->d_bns_f16, ipm, (const float*)nullptr);
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, hipStreamSynchronize(st));  // tmp/nrm are freed on return
	h->f16_scale = scale;
	return EXPANN_OK;
}

const GemmVariant* pick_gemm(const expann_index* h, size_t m) {
	if (h->opt_scan_kernel == 1 || h->dtype != EXPANN_DTYPE_F32)
		return nullptr;
	if (h->metric == EXPANN_METRIC_IP) {
		// inner product: only the fp16 form has it (a placeholder variant keeps the GEMM branch alive)
		const bool ok = f16_choice(h->opt_scan_kernel) && h->f16_scale >= 0.0f &&
		                !(h->opt_scan_kernel == 0 && ((m < 5 && !pick_direct_f16(h->dim, m)) || h->n < 4096));
		if (ok)
			for (const auto& v : kGemmF16Only)
				if (v.d == h->dim)
					return &v;
		return nullptr;
	}
	// measured crossovers at N = 1M, d = 128.  bf16x3 / fp32 forms: m = 16: 0.34 vs 0.37 ms per
	// step, m = 32: 0.49 vs 0.41 ms -- below ~24 queries the HBM-bound direct scan wins.  fp16
	// form (half the bytes per row, one sampled pass): 0.175 vs 0.175 ms at m = 4, 0.173 vs 0.199
	// at m = 8, 0.185 vs 0.31 at m = 16 -- from 5 queries on it wins.
	bool f16_dims = false;
	for (const auto& v : kGemmF16Only)
		f16_dims = f16_dims || (v.d == h->dim && h->f16_scale >= 0.0f);
	if (h->opt_scan_kernel == 0 &&
	    ((m < (f16_dims ? 5u : 24u) && !(f16_dims && pick_direct_f16(h->dim, m))) || h->n < 4096))
		return nullptr;
	for (const auto& v : kGemmF32)
		if (v.d == h->dim)
			return &v;
	if (f16_choice(h->opt_scan_kernel))
		for (const auto& v : kGemmF16Only)
			if (v.d == h->dim)
				return &v;
	return nullptr;
}

int ensure_bnorm(expann_index* h, const GemmVariant* gv, bool bf16, hipStream_t st) {
	DevPtr<float>& dst = bf16 ? h->d_bnorm_bf : h->d_bnorm;
	const uint32_t blocks = (uint32_t)((h->n + kRowsPerGroup - 1) / kRowsPerGroup);
	if (!dst) {
		HIP_TRY(h, hipMalloc(&dst, sizeof(float) * h->n));
		const float eps = bf16 ? gemm_bf16_filter_eps(h->dim) : gemm_filter_eps(h->dim);
		hipLaunchKernelGGL(gv->norms, dim3(blocks), dim3(kBlock), 0, st, (const float*)h->d_base,
		                   (uint32_t)h->n, 1.0f - eps, dst);
		if (!h->d_bnmax)
			HIP_TRY(h, hipMalloc(&h->d_bnmax, 4 * sizeof(float)));
		hipLaunchKernelGGL(max_f32_kernel, dim3(1), dim3(1024), 0, st, (const float*)dst, h->n,
		                   h->d_bnmax + (bf16 ? 1 : 0));
		HIP_TRY(h, hipGetLastError());
	}
	if (bf16 && !h->d_base_split) {
		const size_t nv = h->n * (size_t)h->dim;
		HIP_TRY(h, hipMalloc(&h->d_base_split, nv * 4));
		hipLaunchKernelGGL(split_bf16_kernel, dim3((uint32_t)((nv + kBlock - 1) / kBlock)),
		                   dim3(kBlock), 0, st, (const float*)h->d_base, h->n, h->dim,
		                   h->d_base_split.as<__bf16>());
		HIP_TRY(h, hipGetLastError());
	}
	return EXPANN_OK;
}

// ---- 8-bit GEMM form (int8 MFMA) -----------------------------------------------------------
using GemmI8Fn = void (*)(GemmI8Params);
using SelfI8Fn = void (*)(const void*, uint32_t, int*);
using ThetaI8Fn = void (*)(const void*, uint32_t, const float*, int*, int*);
struct GemmI8Variant {
	int d, mode;
	GemmI8Fn scan;
	SelfI8Fn self;
	ThetaI8Fn theta;
	const char* name;
};
#define GEMM_I8(D, MODE, MN) {D, MODE, scan_gemm_i8_kernel<D, MODE>, row_self_i8_kernel<D, MODE>, \
	query_theta_i8_kernel<D, MODE>, "scan_gemm_i8<" #D "," MN ">"}
const GemmI8Variant kGemmI8[] = {
    GEMM_I8(128, kU8L2, "U8L2"), GEMM_I8(128, kI8L2, "I8L2"), GEMM_I8(128, kI8IP, "I8IP"),
    GEMM_I8(256, kU8L2, "U8L2"), GEMM_I8(256, kI8L2, "I8L2"), GEMM_I8(256, kI8IP, "I8IP"),
    GEMM_I8(768, kU8L2, "U8L2"), GEMM_I8(768, kI8L2, "I8L2"), GEMM_I8(768, kI8IP, "I8IP")};
#undef GEMM_I8

const GemmI8Variant* pick_gemm_i8(const expann_index* h, size_t m) {
	if (h->opt_scan_kernel == 1 || h->dtype == EXPANN_DTYPE_F32)
		return nullptr;
	if (h->opt_scan_kernel == 0 && (m < 96 || h->n < 4096))
		return nullptr;
	for (const auto& v : kGemmI8)
		if (v.d == h->dim && v.mode == h->int_mode)
			return &v;
	return nullptr;
}

int ensure_bias_i8(expann_index* h, const GemmI8Variant* gv, hipStream_t st) {
	if (h->d_bias_i || h->int_mode == kI8IP)
		return EXPANN_OK;
	HIP_TRY(h, hipMalloc(&h->d_bias_i, sizeof(int) * h->n));
	const uint32_t blocks = (uint32_t)((h->n + kRowsPerGroup - 1) / kRowsPerGroup);
	hipLaunchKernelGGL(gv->self, dim3(blocks), dim3(kBlock), 0, st, (const void*)h->d_base,
	                   (uint32_t)h->n, h->d_bias_i);
	HIP_TRY(h, hipGetLastError());
	return EXPANN_OK;
}

// one wave per query for lists of <= 512 keys, then (when the buffers allow longer lists) for
// <= 1024 and <= 2048; select_topk_kernel takes what is left (sel.wave_done = longest list served)
// `expect` = the list length the thresholds aim at (~1.2 k frac): a stage whose lists would be the rare tail
// is not launched -- an empty launch still costs ~5 us of the stream, 3 % of a step on a 125 k-row shard --
// and select_topk_kernel's workgroup-per-query sort takes those few lists.
void launch_select_wave(SelectParams& sel, size_t m, uint32_t cap, hipStream_t st, uint32_t expect) {
	sel.wave_done = 0;
	hipLaunchKernelGGL((select_wave_kernel<8, 4>), dim3((uint32_t)((m + 3) / 4)), dim3(256), 0, st, sel,
	                   (uint32_t)m);
	sel.wave_done = 512;
	if (expect <= 256)
		return;
	if (cap > 512) {
		// (k = 100: ~920-entry lists -- half the registers and ballots of the 2048-key form)
		hipLaunchKernelGGL((select_wave_kernel<16, 2>), dim3((uint32_t)((m + 1) / 2)), dim3(128), 0, st, sel, (uint32_t)m);
		sel.wave_done = 1024;
	}
	if (cap > 1024 && expect > 512) {
		hipLaunchKernelGGL((select_wave_kernel<32, 1>), dim3((uint32_t)m), dim3(64), 0, st, sel, (uint32_t)m);
		sel.wave_done = 2048;
	}
}

// Per-wave hit logs of the scans that write them (scan_gemm_f16x.hpp, scan_gemm_i8w.hpp): one log
// per wave of the grid, together as large as the candidate lists they are filed into; the launch
// geometry is kept for launch_gather_logs.
int ensure_hit_logs(expann_index* h, uint32_t grid, int waves, size_t m, uint32_t cap, uint32_t n_chunks,
                    uint32_t n_qtiles, uint32_t xcd_map, hipStream_t st, uint4** log, uint32_t** log_cnt,
                    uint32_t* log_cap_out) {
	const uint32_t n_logs = grid * (uint32_t)waves;
	// four times a wave's average share of the candidate lists (queries sorted by cluster send a wave's
	// 64 queries to the same row chunk; memory that is never written costs nothing)
	const uint32_t log_cap = std::max<uint32_t>(
	    4096, pow2ceil((uint32_t)std::min<size_t>(4 * ((m * (size_t)cap + n_logs - 1) / n_logs), 1u << 20)));
	const size_t need = (size_t)n_logs * log_cap * 16;
	if (need > h->log_bytes || n_logs > h->log_cnt_n) {
		HIP_TRY(h, hipStreamSynchronize(st));
		h->d_log.reset();
		h->d_log_cnt.reset();
		h->log_bytes = h->log_cnt_n = 0;
		HIP_TRY(h, hipMalloc(&h->d_log, need));
		HIP_TRY(h, hipMalloc(&h->d_log_cnt, sizeof(uint32_t) * n_logs));
		h->log_bytes = need;
		h->log_cnt_n = n_logs;
	}
	*log = h->d_log.as<uint4>();
	*log_cnt = h->d_log_cnt;
	*log_cap_out = log_cap;
	h->pending_scatter.n_logs = n_logs;
	h->pending_scatter.log_cap = log_cap;
	h->pending_scatter.cap = cap;
	h->pending_scatter.n_chunks = n_chunks;
	h->pending_scatter.n_qtiles = n_qtiles;
	h->pending_scatter.xcd_map = xcd_map;
	h->pending_scatter.m = (uint32_t)m;
	h->pending_scatter.theta = nullptr;
	h->pending_scatter.key_mul = 0.0f;
	h->pending_scatter.i_mode = 0;
	return EXPANN_OK;
}

// after a scan that wrote hit logs: file them into the per-query candidate lists
void launch_gather_logs(expann_index* h, hipStream_t st) {
	if (!h->pending_scatter.n_logs)
		return;
	const auto ps = h->pending_scatter;
	h->pending_scatter.n_logs = 0;
	// ~1024 workgroups: the logs of a (query tile, wave) pair are split over n_groups of them
	const uint32_t want = std::max<uint32_t>(1, (1024 + ps.n_qtiles * 4 - 1) / (ps.n_qtiles * 4));
	const uint32_t cpb = std::max<uint32_t>(1, (ps.n_chunks + want - 1) / want);
	const uint32_t n_groups = (ps.n_chunks + cpb - 1) / cpb;
	GatherLogParams gp{h->d_log.as<const uint4>(), h->d_log_cnt.as<const uint32_t>(), ps.log_cap, ps.n_chunks,
	                   ps.n_qtiles, ps.xcd_map, ps.m, n_groups, cpb, h->d_cnt, h->d_cand, ps.cap, ps.theta, ps.key_mul,
	                   ps.i_thp, ps.i_bias, ps.i_qself, ps.i_mode};
	hipLaunchKernelGGL(gather_logs_kernel, dim3(ps.n_qtiles * 4 * n_groups), dim3(kBlock), 0, st, gp);
}

// Deferred check: the flag block of this search goes to the next ring slot and the call returns
// without waiting; expann_sync looks at the slots.  False when the caller has to wait as usual.
bool defer_flags(expann_index* h, hipStream_t st, int attempt) {
	if (!h->opt_async || h->host_call || attempt != 0 || h->strict_u8 || h->async_pending >= kAsyncRing)
		return false;
	if (hipMemcpyAsync(h->h_flag_ring + 8 * h->async_pending, h->d_overflow,
	                   sizeof(uint32_t) * 4 + sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess)
		return false;
	// ring words 6 / 7 are the host's: 6 = fp16 scale of the search (set by the caller, 0 = none),
	// 7 = 1 when word 1 (query values outside [0, 255]) belongs to this search (uint8 rows)
	h->h_flag_ring[8 * h->async_pending + 6] = 0;
	h->h_flag_ring[8 * h->async_pending + 7] = h->dtype == EXPANN_DTYPE_U8 ? 1u : 0u;
	h->async_pending++;
	h->async_stream = st;
	h->prof.deferred_searches++;
	return true;
}

// Row-chunk count of a tiled scan.  `slots` workgroups are resident at a time, so a launch of
// g chunks x n_qtiles workgroups runs in rounds: minimise rounds x (steps per workgroup +
// `overhead` steps of prologue / tail).  With xcd != nullptr a multiple of 8 whose rounding leaves
// exactly g chunks is preferred when it costs at most tol_pct % more: blocks b and b+8 share an
// XCD, so the kernels can then keep a row chunk's query tiles on ONE L2 (GemmF16Params::xcd_map).
uint32_t pick_row_chunks(uint32_t n_tiles, uint32_t n_qtiles, uint32_t slots, double overhead, uint32_t min_tiles,
                         uint32_t g_limit, long tol_pct, uint32_t* xcd) {
	const uint32_t gmax = std::max<uint32_t>(1, n_tiles / min_tiles);
	uint32_t chunks = 1, g8 = 0;
	double best = 1e300, best8 = 1e300;
	for (uint32_t g = 1; g <= std::min<uint32_t>(gmax, g_limit); ++g) {
		const uint32_t steps = (n_tiles + g - 1) / g;
		const uint64_t rounds = ((uint64_t)g * n_qtiles + slots - 1) / slots;
		const double cost = (double)rounds * (steps + overhead);
		if (cost < best * 0.999) {
			best = cost;
			chunks = g;
		}
		if (xcd && g % 8 == 0 && cost < best8 * 0.999 && (n_tiles + steps - 1) / steps == g) {
			best8 = cost;
			g8 = g;
		}
	}
	if (xcd) {
		*xcd = 0;
		if (g8 && best8 <= best * (1.0 + 0.01 * (double)tol_pct)) {
			chunks = g8;
			*xcd = 1;
		}
	}
	return chunks;
}

// Two chunk sizes for a launch of several rounds (scan_gemm_f16x): workgroups are dispatched in block
// order and a launch ends when its last workgroup does, so with equal chunks the last round drains
// for about half a workgroup's time (profiles/wg_times.py: 480 of 512 resident on average at C2).
// The chunks of all rounds but the last stay as they are; the rows of the last round are cut three
// times finer.  Both counts are multiples of 8 (xcd_map).  Returns false when the launch is too
// short to gain (fewer than 3 rounds) or the chunk count does not suit.
bool pick_tail_chunks(uint32_t n_tiles, uint32_t g, uint32_t n_qtiles, uint32_t slots, uint32_t min_tiles,
                      uint32_t* tiles_big, uint32_t* n_big, uint32_t* tiles_small, uint32_t* n_small) {
	const uint64_t rounds = ((uint64_t)g * n_qtiles + slots - 1) / slots;
	if (rounds < 3 || g % 8 != 0 || g < 24)
		return false;
	uint32_t nb = (uint32_t)(((rounds - 1) * slots / n_qtiles) / 8) * 8;  // chunks of the first rounds - 1 rounds
	nb = std::min(nb, g - 8);
	if (nb < 8)
		return false;
	const uint32_t ns = (3 * (g - nb) + 7) / 8 * 8;
	const uint32_t big = (uint32_t)(((uint64_t)n_tiles * 3 + (3 * nb + ns) - 1) / (3 * nb + ns));  // big = 3 small
	if ((uint64_t)nb * big >= n_tiles)
		return false;
	const uint32_t rest = n_tiles - nb * big;
	const uint32_t small = (rest + ns - 1) / ns;
	if (small < min_tiles || (uint64_t)(ns - 1) * small >= rest)  // (every small chunk must hold rows)
		return false;
	*tiles_big = big;
	*n_big = nb;
	*tiles_small = small;
	*n_small = ns;
	return true;
}

// Rows read by the sampled pass = 1/frac.  Its cost falls with frac, the candidates of the full
// scan (~1.2 k frac per query) grow with it.  Measured optima: 12-16 at k = 10 (flat), 8 at
// k = 100 (3.44 ms per 2500 queries x 5 M rows against 3.65 at 5 and 3.62 at 16).
// (8-bit forms: the scan is twice as fast, the candidates cost the same: 5 at k = 100.)
// Round 2, scan_gemm_f16x with hit logs, C3's per-GPU shape (profiles/sweep_frac_k100.sh): 4, 6 and 8
// are even at k = 100 (3.69-3.70 ms per step: scan 2.67 / 2.80 / 2.91 ms at 468 / 703 / 922 candidates per
// query, ~55 cycles of a workgroup per hit), 10 and up lose (12: 4.17 ms; 24 overflows the lists).
// The sample's cost grows with the bytes of the index, the candidates' does not, so the optimum
// moves up on the largest shapes (profiles/sweep_frac_big.sh, ab_frac.sh: C5 = 30 x C2's bytes: 48
// is 3-4 % faster than 16, 96 overflows the lists; 10 M x d128: +1.7 %): ~bytes^0.3 from 8 x C2's
// bytes on, k <= 32 (between C2 and that, and at k = 100, the denser hits cost the scan what the
// sample saves: 1 M x d960, 4 M x d128, 10 M x d128 k = 100 all within +-1 %).
uint32_t sample_frac_for(const expann_index* h, size_t k) {
	if (h->opt_sample_frac > 0)
		return (uint32_t)h->opt_sample_frac;
	const double expo = h->dtype == EXPANN_DTYPE_F32 ? 0.3 : 0.5;
	const double bytes = (double)h->n * (double)h->dim * (h->dtype == EXPANN_DTYPE_F32 ? 2.0 : 1.0);
	double size = (bytes >= 8 * 2.56e8 && k <= 32) ? std::min(3.0, std::pow(bytes / 2.56e8, 0.3)) : 1.0;
	// (round 3, after the hits got cheaper -- lighter flush, keys made by the gather, pruned selects --
	// profiles/sweep_frac_r3.sh: k = 100 on 1.25 M rows: 6 beats 8 by 2.3 % (3.28 vs 3.36 ms), on 10 M rows 8 - 12 are
	// even: ~bytes^0.25 from C3's per-GPU shard on)
	if (h->dtype == EXPANN_DTYPE_F32 && k > 32)
		size = 0.75 * std::pow(std::max(1.0, bytes / 3.2e8), 0.25);
	const double f = 16.0 * std::pow(10.0 / (double)std::max<size_t>(1, k), expo) * size;
	return (uint32_t)std::min(48.0, std::max(4.0, std::round(f)));
}

// ---- 8-bit GEMM form, queue geometry (scan_gemm_i8q.hpp), d = 128 / 256 / 768 / 832 / 960 -----
using GemmI8qFn = void (*)(GemmI8qParams);
struct GemmI8qVariant {
	int d, mode;
	GemmI8qFn scan, sample;
	SelfI8Fn self;
	ThetaI8Fn theta;
	const char* name;
	int dq;  // physical row bytes of the engine's copy (> d: zero-padded, scan_gemm_i8q.hpp)
	int lds, threads, wg_per_cu;
	// scan_gemm_i8w.hpp: the full scan writes per-wave hit logs (gather_logs_kernel files them)
	void (*scan_w)(GemmI8wParams) = nullptr;
	int lds_w = 0, threads_w = 0, wg_per_cu_w = 0;
	void (*sample_w)(GemmI8wParams) = nullptr;  // the sampled pass on the same stream (SAMPLE instance; same launch geometry)
};
// the 16x16x64 kernels of the 8-waves-per-tile geometries (scan_gemm_i8x.hpp): d = 768, and d = 832 / 960 -- rows in
// 1024-byte slots -- with DR = d (the zero-padded k-steps are left out); scan and SAMPLE instance
#define GEMM_I8X_P(D, DQ, MODE, L2F, MN) {D, MODE, scan_gemm_i8x_kernel<DQ, L2F, D>, \
	scan_gemm_i8x_kernel<DQ, L2F, D, true>, row_self_i8_kernel<D, MODE>, query_theta_i8_kernel<D, MODE>, \
	"scan_gemm_i8x<" #DQ "," MN ">", DQ, gemm_i8q_lds_bytes<DQ>(), I8qGeom<DQ>::THREADS, I8qGeom<DQ>::WG_PER_CU}
const GemmI8qVariant kGemmI8x[] = {
    GEMM_I8X_P(768, 768, kU8L2, true, "U8L2"), GEMM_I8X_P(768, 768, kI8L2, true, "I8L2"), GEMM_I8X_P(768, 768, kI8IP, false, "I8IP"),
    GEMM_I8X_P(832, 1024, kU8L2, true, "U8L2"), GEMM_I8X_P(832, 1024, kI8L2, true, "I8L2"),
    GEMM_I8X_P(832, 1024, kI8IP, false, "I8IP"),
    GEMM_I8X_P(960, 1024, kU8L2, true, "U8L2"), GEMM_I8X_P(960, 1024, kI8L2, true, "I8L2"),
    GEMM_I8X_P(960, 1024, kI8IP, false, "I8IP")};
#undef GEMM_I8X_P
// d = 128 / 256: f16x's step structure on 16x16x64 with hit logs (scan_gemm_i8w.hpp)
#define GEMM_I8W(D, MODE, L2F, MN) {D, MODE, nullptr, nullptr, row_self_i8_kernel<D, MODE>, \
	query_theta_i8_kernel<D, MODE>, "scan_gemm_i8w<" #D "," MN ">", D, gemm_i8q_lds_bytes<D>(), kF16Threads, \
	I8wGeom<D>::WG_PER_CU, scan_gemm_i8w_kernel<D, L2F>, gemm_i8w_lds_bytes<D>(), kF16Threads, I8wGeom<D>::WG_PER_CU, \
	scan_gemm_i8w_kernel<D, L2F, true>}
const GemmI8qVariant kGemmI8w[] = {
    GEMM_I8W(128, kU8L2, true, "U8L2"), GEMM_I8W(128, kI8L2, true, "I8L2"), GEMM_I8W(128, kI8IP, false, "I8IP"),
    GEMM_I8W(256, kU8L2, true, "U8L2"), GEMM_I8W(256, kI8L2, true, "I8L2"), GEMM_I8W(256, kI8IP, false, "I8IP")};
#undef GEMM_I8W
constexpr int kRetryGeneric = -1000;  // internal: the caller falls back to the threshold ladder
constexpr int kStrictReject = -1001;  // internal (uint8 shadow): these queries are not 8-bit integers

const GemmI8qVariant* pick_gemm_i8q(const expann_index* h, size_t m, size_t k) {
	if (h->dtype == EXPANN_DTYPE_F32 || !(h->opt_scan_kernel == 0 || h->opt_scan_kernel == 5))
		return nullptr;
	if (h->opt_scan_kernel == 0 && (m < 96 || h->n < 65536))
		return nullptr;
	if (h->n < 2 * 256 * kF16TB || k > 256)
		return nullptr;
	for (const auto& v : kGemmI8w)
		if (v.d == h->dim && v.mode == h->int_mode)
			return &v;
	for (const auto& v : kGemmI8x)
		if (v.d == h->dim && v.mode == h->int_mode)
			return &v;
	return nullptr;
}

int ensure_i8q(expann_index* h, const GemmI8qVariant* gq, hipStream_t st) {
	if (h->d_base_i8q)
		return EXPANN_OK;
	const size_t n_pad = (h->n + kF16TB - 1) / kF16TB * kF16TB;
	if (h->int_mode != kI8IP && !h->d_bias_i) {
		HIP_TRY(h, hipMalloc(&h->d_bias_i, sizeof(int) * h->n));
		hipLaunchKernelGGL(gq->self, dim3((uint32_t)((h->n + kRowsPerGroup - 1) / kRowsPerGroup)),
		                   dim3(kBlock), 0, st, (const void*)h->d_base, (uint32_t)h->n, h->d_bias_i);
	}
	HIP_TRY(h, hipMalloc(&h->d_bp_i8q, sizeof(int) * n_pad));
	hipLaunchKernelGGL(i8q_bp_kernel, dim3((uint32_t)((n_pad + kBlock - 1) / kBlock)), dim3(kBlock), 0,
	                   st, h->int_mode != kI8IP ? h->d_bias_i.as<const int>() : (const int*)nullptr,
	                   (uint32_t)h->n, (uint32_t)n_pad, h->d_bp_i8q);
	if (gq->dq != h->dim) {
		// own copy with rows padded to dq bytes (zeros in the int8 domain), whole 64-row tiles
		void* copy = nullptr;
		HIP_TRY(h, hipMalloc(&copy, n_pad * (size_t)gq->dq));
		hipLaunchKernelGGL(i8q_pad_rows_kernel, dim3(8192), dim3(kBlock), 0, st, (const uint32_t*)h->d_base, h->n,
		                   (uint32_t)h->dim / 4, (uint32_t)gq->dq / 4, n_pad,
		                   h->int_mode == kU8L2 ? 0x80808080u : 0u, (uint32_t*)copy);
		h->d_base_i8q = copy;
		h->base_i8q_owned = true;
	} else if (h->int_mode == kU8L2 || n_pad != h->n) {
		// own copy: whole 64-row tiles (zero rows behind the end), uint8 rows mapped to int8
		void* copy = nullptr;
		HIP_TRY(h, hipMalloc(&copy, n_pad * h->dim));
		const size_t words = h->n * (size_t)h->dim / 4, words_pad = n_pad * (size_t)h->dim / 4;
		hipLaunchKernelGGL(i8q_copy_xor_kernel, dim3(4096), dim3(kBlock), 0, st, (const uint32_t*)h->d_base,
		                   words, words_pad, h->int_mode == kU8L2 ? 0x80808080u : 0u, (uint32_t*)copy);
		h->d_base_i8q = copy;
		h->base_i8q_owned = true;
	} else {
		h->d_base_i8q = h->d_base;
		h->base_i8q_owned = false;
	}
	HIP_TRY(h, hipGetLastError());
	return EXPANN_OK;
}

// sample pass -> thresholds -> full scan -> select, all in the g domain (scan_gemm_i8q.hpp)
int search_i8q(expann_index* h, const GemmI8qVariant* gq, const void* d_queries, size_t m, size_t k,
               uint64_t* d_ids, float* d_dists, hipStream_t st, uint32_t cap) {
	const int cus = num_cus(h->device);
	const bool dbg = std::getenv("EXPANN_DEBUG_SYNC") != nullptr;
	auto mark = [&](const char* what) {
		if (dbg) {
			const hipError_t e = hipStreamSynchronize(st);
			std::fprintf(stderr, "[i8q] %s: %s\n", what, hipGetErrorString(e));
			std::fflush(stderr);
		}
	};
	int rc = ensure_i8q(h, gq, st);
	mark("ensure_i8q");
	if (rc != EXPANN_OK)
		return rc;
	const void* q8 = d_queries;
	if (gq->dq != h->dim) {  // rows of the padded geometry: queries padded (and mapped) the same way
		const size_t nb = m * (size_t)gq->dq;
		HIP_TRY(h, grow_ws(h, h->d_q_split, nb));
		hipLaunchKernelGGL(i8q_pad_rows_kernel, dim3((uint32_t)std::min<size_t>((nb / 4 + kBlock - 1) / kBlock, 1024)),
		                   dim3(kBlock), 0, st, (const uint32_t*)d_queries, m, (uint32_t)h->dim / 4,
		                   (uint32_t)gq->dq / 4, m, h->int_mode == kU8L2 ? 0x80808080u : 0u,
		                   h->d_q_split.as<uint32_t>());
		q8 = h->d_q_split;
	} else if (h->int_mode == kU8L2) {  // queries ^ 0x80 (d_queries is the uint8 conversion, d_q8)
		const size_t nb = m * (size_t)h->dim;
		HIP_TRY(h, grow_ws(h, h->d_q_split, nb));
		hipLaunchKernelGGL(i8q_copy_xor_kernel, dim3((uint32_t)std::min<size_t>((nb / 4 + kBlock - 1) / kBlock, 1024)),
		                   dim3(kBlock), 0, st, (const uint32_t*)d_queries, nb / 4, nb / 4, 0x80808080u,
		                   h->d_q_split.as<uint32_t>());
		q8 = h->d_q_split;
	}
	const uint32_t nt = (uint32_t)((h->n + kF16TB - 1) / kF16TB);
	const uint32_t run = (uint32_t)h->opt_sample_run;
	const uint32_t t_sel = std::max<uint32_t>(256, nt / sample_frac_for(h, k)) / run * run;
	const uint32_t nqt = (uint32_t)((m + kF16TQ - 1) / kF16TQ);
	const uint32_t wg_slots = (uint32_t)(gq->sample_w ? gq->wg_per_cu_w : gq->wg_per_cu) * (uint32_t)cus;
	uint32_t chunks = std::max<uint32_t>(1, wg_slots / nqt);
	chunks = std::max<uint32_t>(chunks, (uint32_t)((8 * k + 31) / 32));
	chunks = std::min<uint32_t>(chunks, std::min<uint32_t>(64, t_sel / 4));
	if (t_sel * 2 > nt || (size_t)chunks * 32 < 8 * k)
		return kRetryGeneric;
	const int lds = gq->lds;
	const dim3 wg((uint32_t)gq->threads);
	for (int attempt = 0;; ++attempt) {
		rc = ensure_workspace(h, m, cap);
		if (rc != EXPANN_OK)
			return rc;
		HIP_TRY(h, hipMemsetAsync(h->d_overflow, 0, sizeof(uint32_t), st));
		HIP_TRY(h, hipMemsetAsync(h->d_total, 0, sizeof(unsigned long long) * 2, st));
		if (h->int_mode != kI8IP)
			hipLaunchKernelGGL(gq->theta, dim3((uint32_t)((m + kRowsPerGroup - 1) / kRowsPerGroup)),
			                   dim3(kBlock), 0, st, d_queries, (uint32_t)m, (const float*)nullptr,
			                   (int*)nullptr, h->d_qself);
		HIP_TRY(h, grow_ws(h, h->d_sample, m * (size_t)chunks * 32 * sizeof(int)));
		GemmI8qParams sp{};
		sp.base = h->d_base_i8q;
		sp.bp = h->d_bp_i8q;
		sp.bias = h->int_mode != kI8IP ? h->d_bias_i : nullptr;
		sp.n_rows = (uint32_t)h->n;
		sp.n_tiles_sel = t_sel;
		sp.tile_stride = nt / t_sel;
		sp.tile_run = run;
		sp.tiles_per_block = (t_sel + chunks - 1) / chunks;
		const uint32_t schunks = (t_sel + sp.tiles_per_block - 1) / sp.tiles_per_block;
		sp.n_qtiles = nqt;
		sp.queries = q8;
		sp.m = (uint32_t)m;
		sp.sample_out = h->d_sample.as<int>();
		sp.n_chunks = schunks;
		if (dbg)
			std::fprintf(stderr, "[i8q] sample: grid %u x %u, t_sel %u stride %u tpb %u chunks %u nt %u m %zu base %p bp %p q %p out %p (%zu B)\n",
			             schunks, nqt, t_sel, sp.tile_stride, sp.tiles_per_block, schunks, nt, m, sp.base,
			             (const void*)sp.bp, sp.queries, (void*)sp.sample_out, h->d_sample.bytes);
		mark("qself");
		if (gq->sample_w) {
			GemmI8wParams wsp{};
			wsp.q = sp;
			hipLaunchKernelGGL(gq->sample_w, dim3(schunks * nqt), dim3((uint32_t)gq->threads_w), gq->lds_w, st, wsp);
		} else {
			hipLaunchKernelGGL(gq->sample, dim3(schunks * nqt), wg, lds, st, sp);
		}
		mark("sample");
		SampleTauI8Params tp{};
		tp.vals = h->d_sample.as<const int>();
		tp.n_vals = schunks * 32;
		tp.m = (uint32_t)m;
		tp.k = (uint32_t)k;
		tp.thp = h->d_theta.as<int>();
		tp.cand_cnt = h->d_cnt;
		hipLaunchKernelGGL(tp.n_vals <= 512 ? sample_tau_i8_kernel<8>
		                                   : (tp.n_vals <= 1024 ? sample_tau_i8_kernel<16> : sample_tau_i8_kernel<32>),
		                   dim3((uint32_t)((m + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st, tp);
		mark("tau");
		// the full scan
		GemmI8qParams fp = sp;
		fp.sample_out = nullptr;
		fp.n_tiles_sel = nt;
		fp.tile_stride = 1;
		fp.tile_run = 1;
		fp.thp = h->d_theta.as<const int>();
		fp.qself = h->d_qself;
		fp.cand_cnt = h->d_cnt;
		fp.cand = h->d_cand;
		fp.cap = cap;
		// (a retry after overflowed logs / lists stays on the same kernel: the logs grow with the lists)
		const GemmI8qVariant* gs = gq;
		const uint32_t scan_slots = gs->scan_w ? (uint32_t)gs->wg_per_cu_w * (uint32_t)cus : wg_slots;
		uint32_t fchunks = pick_row_chunks(nt, nqt, scan_slots, 4.0, 8, 2048, h->opt_xcd_tolerance,
		                                   (h->opt_debug & 1024) ? nullptr : &fp.xcd_map);
		fp.tiles_per_block = (nt + fchunks - 1) / fchunks;
		fchunks = (nt + fp.tiles_per_block - 1) / fp.tiles_per_block;
		GemmI8wParams wp{};
		if (gs->scan_w) {  // four hit logs per workgroup, one per 64 queries of its tile (may wait for the stream to grow them)
			wp.q = fp;
			rc = ensure_hit_logs(h, fchunks * nqt, 4, m, cap, fchunks, nqt, fp.xcd_map, st, &wp.log, &wp.log_cnt,
			                     &wp.log_cap);
			if (rc != EXPANN_OK)
				return rc;
			wp.lost = h->d_overflow;
			h->pending_scatter.i_thp = fp.thp;
			h->pending_scatter.i_bias = fp.bias;
			h->pending_scatter.i_qself = fp.qself;
			h->pending_scatter.i_mode = fp.bias ? 1 : 2;
		}
		const bool timed = h->profiling && h->ev_used < kEventPairs;
		if (timed)
			HIP_TRY(h, hipEventRecord(h->ev[h->ev_used][0], st));
		if (gs->scan_w)
			hipLaunchKernelGGL(gs->scan_w, dim3(fchunks * nqt), dim3((uint32_t)gs->threads_w), gs->lds_w, st, wp);
		else
			hipLaunchKernelGGL(gs->scan, dim3(fchunks * nqt), dim3((uint32_t)gs->threads), gs->lds, st, fp);
		if (timed) {
			HIP_TRY(h, hipEventRecord(h->ev[h->ev_used][1], st));
			h->ev_used++;
		}
		launch_gather_logs(h, st);
		if (timed || !h->profiling) {
			h->prof.scan_launches++;
			h->prof.scan_rows += h->n;
			h->prof.scan_query_tiles += nqt;
			h->prof.query_tile = kF16TQ;
			h->prof.levels = 2;
			std::snprintf(h->prof.scan_kernel, sizeof(h->prof.scan_kernel), "%s", gs->name);
		}
		HIP_TRY(h, hipGetLastError());
		mark("scan");
		SelectParams sel{};
		sel.cand = h->d_cand;
		sel.cand_cnt = h->d_cnt;
		sel.cap = cap;
		sel.k = (uint32_t)k;
		sel.id_offset = h->id_offset;
		sel.out_ids = d_ids;
		sel.out_dists = d_dists;
		sel.dim = (uint32_t)h->dim;
		sel.overflow = h->d_overflow;
		if (h->profiling)
			hipLaunchKernelGGL(sum_u32_kernel, dim3(1), dim3(1024), 0, st, sel.cand_cnt, (uint32_t)m,
			                   h->d_total);
		launch_select_wave(sel, m, cap, st, (uint32_t)(1.2 * (double)k * sample_frac_for(h, k)));
		mark("select_wave");
		hipLaunchKernelGGL(select_topk_kernel, dim3((uint32_t)m), dim3(kBlock), sizeof(uint64_t) * cap + 16,
		                   st, sel);
		HIP_TRY(h, hipGetLastError());
		if (defer_flags(h, st, attempt)) {  // deferred check: expann_sync reads the flags
			h->h_flag_ring[8 * (h->async_pending - 1) + 6] = 0;
			return EXPANN_OK;
		}
		HIP_TRY(h, hipMemcpyAsync(h->h_flags, h->d_overflow, sizeof(uint32_t) * 4 + sizeof(unsigned long long),
		                          hipMemcpyDeviceToHost, st));
		HIP_TRY(h, hipStreamSynchronize(st));
		unsigned long long tot;
		std::memcpy(&tot, h->h_flags + 4, sizeof(tot));
		h->prof.candidates = tot;
		if (h->strict_u8 && (h->h_flags[1] != 0 || h->h_flags[3] != 0))
			return kStrictReject;
		if (h->dtype == EXPANN_DTYPE_U8 && h->h_flags[1] != 0)
			return h->fail(EXPANN_ERR_UNSUPPORTED,
			               std::to_string(h->h_flags[1]) +
			                   " query values outside [0,255]: the uint8 metric "
			                   "(dist2_compressed) is only defined for 8-bit valued queries");
		if (h->h_flags[0] == 0)
			return EXPANN_OK;
		h->prof.retries++;
		if (cap >= kMaxCap || attempt >= 2)
			return kRetryGeneric;  // massive ties: the ladder's direct kernel breaks them by row
		cap = std::min(cap * 4, kMaxCap);
	}
}

int search_pass(expann_index* h, const void* d_queries, size_t m, size_t k, uint64_t* d_ids,
                float* d_dists, hipStream_t st);
}  // namespace
extern "C" int expann_create(int dim, int dtype, int metric, int device, expann_index** out);
extern "C" int expann_set_base_device(expann_index* h, const void* d_rows, size_t n, uint64_t id_offset);
extern "C" int expann_set_profiling(expann_index* h, int enable);
extern "C" void expann_destroy(expann_index* h);
namespace {

// Exact-uint8 shortcut of an fp32 index (fields of expann_index): examine the rows once; if all
// are integers in [0, 255] keep a uint8 copy behind an internal uint8 engine.
int build_u8_shadow(expann_index* h, hipStream_t st) {
	const size_t nv = h->n * (size_t)h->dim;
	DevBuf bad;
	HIP_TRY(h, bad.alloc(sizeof(uint32_t)));
	HIP_TRY(h, hipMemsetAsync(bad.p, 0, sizeof(uint32_t), st));
	hipLaunchKernelGGL(count_non_u8_kernel, dim3(2048), dim3(kBlock), 0, st, (const float*)h->d_base, nv,
	                   bad.as<uint32_t>());
	uint32_t n_bad = 1;
	HIP_TRY(h, hipMemcpyAsync(&n_bad, bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	HIP_TRY(h, hipStreamSynchronize(st));
	if (n_bad != 0) {
		h->u8_exact = -1;
		return EXPANN_OK;
	}
	HIP_TRY(h, hipMalloc(&h->d_base_u8, nv));
	hipLaunchKernelGGL(cast_f32_u8_kernel, dim3(4096), dim3(kBlock), 0, st, (const float*)h->d_base, nv,
	                   h->d_base_u8.as<uint8_t>());
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, hipStreamSynchronize(st));
	expann_index* sh = nullptr;
	int rc = expann_create(h->dim, EXPANN_DTYPE_U8, EXPANN_METRIC_L2, h->device, &sh);
	if (rc == EXPANN_OK)
		rc = expann_set_base_device(sh, h->d_base_u8, h->n, h->id_offset);
	if (rc == EXPANN_OK && h->profiling)
		rc = expann_set_profiling(sh, 1);
	if (rc != EXPANN_OK) {
		if (sh)
			expann_destroy(sh);
		h->d_base_u8.reset();
		h->u8_exact = -1;  // (not fatal: the fp32 paths serve the index)
		return EXPANN_OK;
	}
	sh->strict_u8 = true;
	h->u8_shadow = sh;
	h->u8_exact = 1;
	return EXPANN_OK;
}
void drop_u8_shadow(expann_index* h) {
	if (h->u8_shadow)
		expann_destroy(h->u8_shadow);
	h->u8_shadow = nullptr;
	h->d_base_u8.reset();
	h->u8_exact = 0;
}
// the shadow's counters and scan time become the parent's
int absorb_shadow_profile(expann_index* h) {
	expann_index* sh = h->u8_shadow;
	double ms = 0;
	for (int i = 0; i < sh->ev_used; ++i) {
		HIP_TRY(h, hipEventSynchronize(sh->ev[i][1]));
		float t = 0;
		HIP_TRY(h, hipEventElapsedTime(&t, sh->ev[i][0], sh->ev[i][1]));
		ms += t;
	}
	sh->ev_used = 0;
	h->prof_extra_ms += ms;
	h->prof.scan_launches += sh->prof.scan_launches;
	h->prof.scan_rows += sh->prof.scan_rows;
	h->prof.scan_query_tiles += sh->prof.scan_query_tiles;
	h->prof.retries += sh->prof.retries;
	h->prof.deferred_searches += sh->prof.deferred_searches;
	h->prof.query_tile = sh->prof.query_tile;
	h->prof.levels = sh->prof.levels;
	h->prof.candidates = sh->prof.candidates;
	std::memcpy(h->prof.scan_kernel, sh->prof.scan_kernel, sizeof(h->prof.scan_kernel));
	sh->prof = expann_profile{};
	return EXPANN_OK;
}

// One sampled pass of the fp16 form gives every query the threshold of its full scan (4.6 of
// DESIGN.md): class maxima of g over 1/frac of the rows, their k-th largest -> tau, theta', and
// the list counters zeroed.  A handful of queries: sampled pass and thresholds in ONE launch
// (scan_direct_f16.hpp); otherwise the MFMA SAMPLE kernel + sample_tau_kernel.  *done stays false
// when the index is too small for a sample (the caller runs the threshold ladder instead).
int sampled_pass_f16(expann_index* h, const GemmF16Variant* gvf, size_t m, size_t k, bool ip, int cus,
                     float* d_tau, uint32_t* d_tau_row, hipStream_t st, bool* done) {
	*done = false;
	auto ensure_sample = [&](size_t need) -> int {
		HIP_TRY(h, grow_ws(h, h->d_sample, need));
		return EXPANN_OK;
	};
	SampleTauParams tp{};
	tp.m = (uint32_t)m;
	tp.k = (uint32_t)k;
	tp.qnrm = h->d_qnrm;
	tp.eps = gemm_f16_filter_eps(h->dim);
	tp.abs_coef = std::ldexp(1.0f, -24) / h->f16_scale * std::sqrt((float)h->dim);
	tp.inv_mul = 2.0f / (h->f16_scale * h->f16_scale);
	tp.ip = ip ? 1 : 0;
	tp.qscale = ip ? h->d_qscale.as<const float>() : nullptr;
	tp.tau = d_tau;
	tp.tau_row = d_tau_row;
	tp.theta = h->d_theta;
	tp.mul = 0.5f * h->f16_scale * h->f16_scale;
	tp.cand_cnt = h->d_cnt;

	const DirectF16Variant* dvs = h->opt_scan_kernel == 0 ? pick_direct_f16(h->dim, m) : nullptr;
	if (dvs && dvs->sample) {
		const uint32_t steps_all = (uint32_t)((h->n + dvs->rps - 1) / dvs->rps);
		const uint32_t sel =
		    std::max<uint32_t>(256u * 64u / (uint32_t)dvs->rps, steps_all / sample_frac_for(h, k));
		uint32_t wgs = std::min<uint32_t>(2048u / (uint32_t)dvs->cpw, sel / 4);
		if (sel * 2 <= steps_all && (size_t)wgs * dvs->cpw >= 8 * k) {
			SampleDirectParams sp{};
			sp.g.base_f16 = h->d_base_f16;
			sp.g.bnorm = h->d_bns_f16;  // upper row term: thresholds hold row by row
			sp.g.n_rows = (uint32_t)h->n;
			sp.g.n_tiles_sel = sel;
			sp.g.tile_stride = steps_all / sel;
			sp.g.tiles_per_block = (sel + wgs - 1) / wgs;
			wgs = (sel + sp.g.tiles_per_block - 1) / sp.g.tiles_per_block;
			sp.g.queries_f16 = h->d_q_split;
			sp.g.m = (uint32_t)m;
			const int ra = ensure_sample(m * (size_t)wgs * dvs->cpw * sizeof(float));
			if (ra != EXPANN_OK)
				return ra;
			sp.t = tp;
			sp.t.vals = h->d_sample;
			sp.t.n_vals = wgs * (uint32_t)dvs->cpw;
			sp.ticket = h->d_ticket;
			DevBuf sclk;
			if (h->opt_debug & 16) {
				HIP_TRY(h, sclk.alloc(3 * 8));
				sp.g.clk = sclk.as<unsigned long long>();
			}
			hipLaunchKernelGGL(dvs->sample, dim3(wgs), dim3(kBlock), 0, st, sp);
			HIP_TRY(h, hipGetLastError());
			if (sp.g.clk) {
				unsigned long long c[3] = {0, 0, 0};
				HIP_TRY(h, hipStreamSynchronize(st));
				HIP_TRY(h, hipMemcpy(c, sp.g.clk, sizeof(c), hipMemcpyDeviceToHost));
				std::fprintf(stderr, "sample_direct_f16: %u workgroups, stream %.2f us, thresholds %.2f us\n", wgs,
				             (double)(c[1] - c[0]) / 100.0, (double)(c[2] - c[1]) / 100.0);
			}
			*done = true;
			return EXPANN_OK;
		}
	}
	const uint32_t nt = (uint32_t)((h->n + gvf->tb - 1) / gvf->tb);
	const uint32_t run = (uint32_t)h->opt_sample_run;
	const uint32_t t_sel = std::max<uint32_t>(256, nt / sample_frac_for(h, k)) / run * run;
	const uint32_t nqt = (uint32_t)((m + gvf->wgq - 1) / gvf->wgq);
	uint32_t chunks = std::max<uint32_t>(1, ((uint32_t)gvf->wg_per_cu * (uint32_t)cus) / nqt);
	chunks = std::max<uint32_t>(chunks, (uint32_t)((8 * k + 31) / 32));
	chunks = std::min<uint32_t>(chunks, std::min<uint32_t>(64, t_sel / 4));
	if (t_sel * 2 > nt || (size_t)chunks * 32 < 8 * k)
		return EXPANN_OK;
	const int ra = ensure_sample(m * (size_t)chunks * 32 * sizeof(float));
	if (ra != EXPANN_OK)
		return ra;
	GemmF16Params fp{};
	fp.base_f16 = h->d_base_f16;
	fp.bnorm = h->d_bns_f16;
	fp.n_rows = (uint32_t)h->n;
	fp.n_tiles_sel = t_sel;
	fp.tile_stride = nt / t_sel;
	fp.tile_run = run;
	fp.tiles_per_block = (t_sel + chunks - 1) / chunks;
	chunks = (t_sel + fp.tiles_per_block - 1) / fp.tiles_per_block;
	fp.n_qtiles = nqt;
	fp.queries_f16 = h->d_q_split;
	fp.m = (uint32_t)m;
	fp.sample_out = h->d_sample;
	fp.n_chunks = chunks;
	hipLaunchKernelGGL(gvf->sample, dim3(chunks * nqt), dim3((uint32_t)gvf->threads), gvf->lds, st, fp);
	tp.vals = h->d_sample;
	tp.n_vals = chunks * 32;
	hipLaunchKernelGGL(tp.n_vals <= 512 ? sample_tau_kernel<8>
	                                   : (tp.n_vals <= 1024 ? sample_tau_kernel<16> : sample_tau_kernel<32>),
	                   dim3((uint32_t)((m + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st, tp);
	HIP_TRY(h, hipGetLastError());
	*done = true;
	return EXPANN_OK;
}

// The fp16-form scan of one threshold level (theta' is in h->d_theta): the MFMA kernel of the
// index's dimension over `rows_sel` sampled rows (last: all rows), or -- a handful of queries on
// the last level -- the same filter streamed from HBM (scan_direct_f16.hpp).
int launch_scan_f16(expann_index* h, const GemmF16Variant* gvf, uint32_t rows_sel, bool last, size_t m, int cus,
                    bool ip, uint32_t cap, hipStream_t st, const char** kname, uint32_t* n_qtiles,
                    uint32_t* query_tile) {
	*query_tile = (uint32_t)gvf->wgq;
	// 64-row tiles, 256 queries per workgroup, two workgroups resident per CU
	GemmF16Params fp{};
	fp.base_f16 = h->d_base_f16;
	fp.bnorm = h->d_bnorm_f16;
	fp.n_rows = (uint32_t)h->n;
	const uint32_t nt = (uint32_t)((h->n + gvf->tb - 1) / gvf->tb);
	fp.n_tiles_sel = last ? nt : std::min(nt, (rows_sel + gvf->tb - 1) / gvf->tb);
	fp.tile_stride = std::max<uint32_t>(1, nt / fp.n_tiles_sel);
	fp.tile_run = 1;
	if (!last && fp.tile_stride >= 16 && fp.n_tiles_sel >= 16) {
		// sampled level: runs of 16 consecutive tiles (one 256 KiB stretch each at d = 128)
		fp.tile_run = 16;
		fp.n_tiles_sel = (fp.n_tiles_sel / 16) * 16;
	}
	fp.n_qtiles = (uint32_t)((m + gvf->wgq - 1) / gvf->wgq);
	uint32_t fchunks = pick_row_chunks(fp.n_tiles_sel, fp.n_qtiles, (uint32_t)gvf->wg_per_cu * (uint32_t)cus,
	                                   4.0, 8, 2048, h->opt_xcd_tolerance,
	                                   (h->opt_debug & 1024) ? nullptr : &fp.xcd_map);
	if (h->opt_scan_chunks > 0) {  // (experiments: force the row-chunk count)
		fchunks = (uint32_t)std::min<long>(h->opt_scan_chunks, std::max<uint32_t>(1, fp.n_tiles_sel / 8));
		fp.xcd_map = (fchunks % 8 == 0 && !(h->opt_debug & 1024)) ? 1 : 0;
		const uint32_t tpb = (fp.n_tiles_sel + fchunks - 1) / fchunks;
		if ((fp.n_tiles_sel + tpb - 1) / tpb != fchunks)
			fp.xcd_map = 0;
	}
	fp.tiles_per_block = (fp.n_tiles_sel + fchunks - 1) / fchunks;
	fchunks = (fp.n_tiles_sel + fp.tiles_per_block - 1) / fp.tiles_per_block;
	if (gvf->hit_log && fp.xcd_map && h->opt_tail_chunks && !h->opt_scan_chunks) {  // (scan_gemm_f16x reads n_big)
		uint32_t big, nb, small, ns;
		// (small chunks of >= 64 steps: at 40 -- 500 k rows -- their prologues cost more than the drain saves)
		if (pick_tail_chunks(fp.n_tiles_sel, fchunks, fp.n_qtiles, (uint32_t)gvf->wg_per_cu * (uint32_t)cus, 64, &big, &nb,
		                     &small, &ns)) {
			fp.tiles_per_block = big;
			fp.n_big = nb;
			fp.tiles_small = small;
			fchunks = nb + ns;
		}
	}
	fp.queries_f16 = h->d_q_split;
	fp.theta = h->d_theta;
	fp.two_inv_s2 = (ip ? 1.0f : 2.0f) / (h->f16_scale * h->f16_scale);
	fp.m = (uint32_t)m;
	fp.cand_cnt = h->d_cnt;
	fp.cand = h->d_cand;
	fp.cap = cap;
	fp.debug = (uint32_t)h->opt_debug & ~16u;
	DevBuf clk;
	if (h->opt_debug & 16) {
		HIP_TRY(h, clk.alloc((18 + 3 * 65536) * 8));  // (+ per-workgroup records of scan_gemm_f16x)
		HIP_TRY(h, hipMemsetAsync(clk.p, 0, (18 + 3 * 65536) * 8, st));
		fp.clk = clk.as<unsigned long long>();
	}
	const DirectF16Variant* dv =
	    (last && h->opt_scan_kernel == 0) ? pick_direct_f16(h->dim, m) : nullptr;
	if (dv) {
		// a handful of queries: stream the fp16 rows without the matrix cores, ~16
		// workgroups per CU, whole RPS-row steps (inside the copy's 64-row padding)
		fp.n_qtiles = (uint32_t)((m + dv->tq - 1) / dv->tq);
		fp.n_tiles_sel = (uint32_t)((h->n + dv->rps - 1) / dv->rps);
		const uint32_t want = std::max<uint32_t>(1, (16u * (uint32_t)cus) / fp.n_qtiles);
		fp.tiles_per_block = std::max<uint32_t>(8, (fp.n_tiles_sel + want - 1) / want);
		fchunks = (fp.n_tiles_sel + fp.tiles_per_block - 1) / fp.tiles_per_block;
		fp.xcd_map = 0;
		hipLaunchKernelGGL(dv->fn, dim3(fchunks * fp.n_qtiles), dim3(kBlock), 0, st, fp);
		*kname = dv->name;
		*query_tile = (uint32_t)dv->tq;
	} else {
		const uint32_t grid = fchunks * fp.n_qtiles;
		if (gvf->hit_log) {
			const int lrc = ensure_hit_logs(h, grid, gvf->threads / 64, m, cap, fchunks, fp.n_qtiles, fp.xcd_map, st,
			                                &fp.log, &fp.log_cnt, &fp.log_cap);
			if (lrc != EXPANN_OK)
				return lrc;
			fp.lost = h->d_overflow;
			h->pending_scatter.theta = fp.theta;
			h->pending_scatter.key_mul = fp.two_inv_s2;
		}
		uint32_t launch_grid = grid;
		const uint32_t resident = (uint32_t)gvf->wg_per_cu * (uint32_t)cus;
		// (built into the d = 128 instance; items shorter than ~160 steps -- a thousand queries over 1 M rows --
		// lose more to the per-item pull and prologue than the even end gains: 2.91 M vs 3.00 M QPS)
		if (gvf->hit_log && gvf->d == 128 && h->opt_persist && grid > resident &&
		    (fp.tiles_per_block >= 160 || h->opt_persist > 1)) {
			// persistent launch: what is resident pulls the `grid` items from per-XCD counters
			if (!h->d_work_ctr)
				HIP_TRY(h, hipMalloc(&h->d_work_ctr, sizeof(uint32_t) * 8 * 16));
			HIP_TRY(h, hipMemsetAsync(h->d_work_ctr, 0, sizeof(uint32_t) * 8 * 16, st));
			fp.work_ctr = h->d_work_ctr;
			fp.n_items = grid;
			launch_grid = resident;
		}
		hipLaunchKernelGGL(gvf->scan, dim3(launch_grid), dim3((uint32_t)gvf->threads), gvf->lds, st, fp);
		*kname = gvf->name;
	}
	*n_qtiles = fp.n_qtiles;
	if (fp.clk) {
		int occ = -1;
		hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)gvf->scan, gvf->threads, gvf->lds);
		std::fprintf(stderr, "scan_gemm_f16: %d workgroups per CU resident, grid %u\n", occ,
		             fchunks * fp.n_qtiles);
		unsigned long long c[18] = {0};
		HIP_TRY(h, hipMemcpy(c, fp.clk, sizeof(c), hipMemcpyDeviceToHost));
		std::fprintf(stderr, "scan_gemm_f16 wg0: %llu shader clocks in %.1f us = %.0f MHz\n", c[0],
		             c[1] / 100.0, c[1] ? c[0] * 100.0 / c[1] : 0.0);
		if (c[6])
			std::fprintf(stderr, "  per step (%llu steps): columns 0-1 %.0f, stage wait + barrier %.0f, columns 2-3 %.0f, tail + queue work %.0f cycles\n",
			             c[6], (double)c[2] / c[6], (double)c[3] / c[6], (double)c[4] / c[6], (double)c[5] / c[6]);
		if (const char* dump = std::getenv("EXPANN_WG_TIMES")) {  // per-workgroup start / end / place, for profiles/wg_times.py
			const uint32_t grid = std::min<uint32_t>(fchunks * fp.n_qtiles, 65536);
			std::vector<unsigned long long> rec(3 * (size_t)grid);
			HIP_TRY(h, hipMemcpy(rec.data(), fp.clk + 18, rec.size() * 8, hipMemcpyDeviceToHost));
			if (FILE* f = std::fopen(dump, "w")) {
				for (uint32_t b = 0; b < grid; ++b)
					std::fprintf(f, "%u %llu %llu %llu\n", b, rec[3 * b], rec[3 * b + 1], rec[3 * b + 2]);
				std::fclose(f);
			}
		}
		if (c[14])  // (scan_gemm_f16x: the last workgroup of the grid as well)
			std::fprintf(stderr, "scan_gemm_f16 last wg: %llu shader clocks in %.1f us = %.0f MHz, started %.1f us after wg0; per step (%llu): %.0f / %.0f / %.0f / %.0f\n",
			             c[8], c[9] / 100.0, c[9] ? c[8] * 100.0 / c[9] : 0.0, (double)(c[15] - c[7]) / 100.0, c[14],
			             (double)c[10] / c[14], (double)c[11] / c[14], (double)c[12] / c[14], (double)c[13] / c[14]);
	}
	return EXPANN_OK;
}

// ---- one pass over <= kMaxQueriesPerPass queries, as a planner + steps -------------------------
// choose_kernels()   which kernel family serves this pass (the planner: row type, metric, batch size,
//                    options, what an earlier attempt learned) and the index's derived copies it needs
// prepare_queries()  the query-side conversions of that family (scaled fp16 / bf16 split)
// plan_thresholds()  threshold levels of this attempt: the ladder, class minima, or one sampled pass
// run_level(li)      scan of level li (direct / fp32 / bf16x3 / fp16 / int8 MFMA filter) + selection
// check(attempt)     the one host wait: flags read back, retry with larger lists / another family
struct ScanSel {
	ScanFn fn = nullptr;
	int tq = 0;
	const char* name = "";
};
struct SearchPass {
	expann_index* h;
	const void* d_queries;
	size_t m, k;
	uint64_t* d_ids;
	float* d_dists;
	hipStream_t st;
	bool ip;
	int cus;
	uint32_t cap;
	ScanSel svs;
	const ScanSel* sv = &svs;
	// the planner's choice
	const GemmVariant* gv = nullptr;
	const GemmI8Variant* gvi = nullptr;
	const GemmBf16Variant* gvb = nullptr;
	const GemmF16Variant* gvf = nullptr;
	// what an attempt learned
	bool flags_clean = false;   // the fp16 prelude has just zeroed the flag block
	bool force_direct = false;  // a GEMM-form filter overflowed: massive near-ties
	bool no_f16 = false;        // the queries do not fit the fp16 range of this index
	// thresholds of the current attempt
	std::vector<Level> levels;
	uint32_t n_qtiles = 0;
	size_t li_start = 0;
	bool theta_ready = false;   // the sampled pass also wrote theta' and zeroed the list counters

	enum Next { kDone, kRestart, kRetry };
	int choose_kernels();
	int prepare_queries(bool* restart);
	int plan_thresholds();
	int run_level(size_t li);
	int check(int attempt, Next* next);
	int run();
};

int SearchPass::choose_kernels() {
	gv = force_direct ? nullptr : pick_gemm(h, m);
	gvi = force_direct ? nullptr : pick_gemm_i8(h, m);
	gvb = nullptr;
	gvf = nullptr;
	if (h->opt_scan_kernel == 2 && !gvi)
		return h->fail(EXPANN_ERR_UNSUPPORTED,
		               "GEMM-form scan (scan_kernel=2): 8-bit L2/IP with dim 128/256/768 (the fp32-input MFMA form of "
		               "rounds 1-2 is gone: scan_kernel 3 = bf16x3, 4 = fp16)");
	if (gvi) {
		int rc = ensure_bias_i8(h, gvi, st);
		if (rc != EXPANN_OK)
			return rc;
	}
	if (gv)
		for (const auto& v : kGemmBf16)
			if (v.d == h->dim)
				gvb = &v;
	if (h->opt_scan_kernel == 3 && !gvb)
		return h->fail(EXPANN_ERR_UNSUPPORTED, "bf16x3 GEMM-form scan: f32 L2 with dim 64 or 128 only");
	// fp16 single-product form: default when available; a search whose queries leave the fp16
	// range after scaling is redone with the bf16x3 form (no_f16)
	if (gv && !no_f16 && f16_choice(h->opt_scan_kernel)) {
		const bool dbg = (h->opt_debug & ~16L) != 0;  // (the debug instance exists for d = 128 only)
		for (const auto& v : kGemmF16X)
			if (v.d == h->dim && !(dbg && v.d == kGemmF16XDbg[0].d))
				gvf = &v;
		if (dbg && kGemmF16XDbg[0].d == h->dim)
			gvf = &kGemmF16XDbg[0];
	}
	if ((h->opt_scan_kernel == 4 || h->opt_scan_kernel == 6) && !gvf && !no_f16)
		return h->fail(EXPANN_ERR_UNSUPPORTED, "fp16 GEMM-form scan: f32 with dim 64, 128, 256, 512, 768, 832 or 960 only");
	if (gvf)
		gvb = nullptr;
	else if ((h->opt_scan_kernel == 0 && m < 24) || !gvb)
		gv = nullptr, gvb = nullptr;  // without the fp16 form only bf16x3 is left (d = 64 / 128; pays from ~24 queries on)
	if (gv && !gvf) {
		int rc = ensure_bnorm(h, gv, gvb != nullptr, st);
		if (rc != EXPANN_OK)
			return rc;
	}
	return EXPANN_OK;
}

int SearchPass::prepare_queries(bool* restart) {
	*restart = false;
	if (gvf) {
		int rc = ensure_f16(h, gvf, st);
		if (rc != EXPANN_OK)
			return rc;
		if (h->f16_scale < 0.0f) {  // the index does not fit the fp16 range at any allowed scale
			no_f16 = true;
			*restart = true;
			return EXPANN_OK;
		}
		rc = ensure_workspace(h, m, cap);
		if (rc != EXPANN_OK)
			return rc;
		const size_t nv = m * (size_t)h->dim;
		HIP_TRY(h, grow_ws(h, h->d_q_split, nv * 4));
		// scaled fp16 queries, ||q||^2, and the largest |q| (range check, read back at the end):
		// one memset of the flag block, one kernel
		// (latency mode, one workgroup: the kernel clears the flag block itself and, when the queries
		// still sit in pinned host memory, leaves the device copy the later kernels read)
		const bool one_wg = m <= (size_t)kRowsPerGroup;
		const bool from_host = h->q_in_pinned_host && one_wg;
		if (!one_wg)
			HIP_TRY(h, hipMemsetAsync(h->d_overflow, 0, 32, st));
		flags_clean = true;
		hipLaunchKernelGGL(gvf->prep, dim3((uint32_t)((m + kRowsPerGroup - 1) / kRowsPerGroup)), dim3(kBlock),
		                   0, st, (const float*)d_queries, (uint32_t)m, h->f16_scale, h->d_q_split.as<_Float16>(),
		                   h->d_qnrm, h->d_overflow + 2, from_host ? h->d_q.as<float>() : (float*)nullptr,
		                   one_wg ? h->d_overflow : (uint32_t*)nullptr,
		                   (ip && h->opt_ip_rescale) ? (const float*)(h->d_bnmax + 2) : (const float*)nullptr,
		                   h->d_qscale.as<float>());
		HIP_TRY(h, hipGetLastError());
		if (from_host) {
			d_queries = h->d_q;
			h->q_in_pinned_host = false;
		}
	}
	if (gvb) {  // queries -> bf16 hi/lo planes
		const size_t nv = m * (size_t)h->dim;
		HIP_TRY(h, grow_ws(h, h->d_q_split, nv * 4));
		hipLaunchKernelGGL(split_bf16_kernel, dim3((uint32_t)((nv + kBlock - 1) / kBlock)),
		                   dim3(kBlock), 0, st, (const float*)d_queries, m, h->dim,
		                   h->d_q_split.as<__bf16>());
		HIP_TRY(h, hipGetLastError());
	}
	return EXPANN_OK;
}

int SearchPass::plan_thresholds() {
	int rc = ensure_workspace(h, m, cap);
	if (rc != EXPANN_OK)
		return rc;
	levels = plan_levels(h->n, k, cap, h->opt_sample_ratio);
	n_qtiles = (uint32_t)((m + sv->tq - 1) / sv->tq);
	if (!gv && !gvi && levels.size() >= 3 && h->opt_sample_pass) {
		// direct path: ONE sampled level of class minima (1/16 of the rows) instead of the
		// first two levels of the ladder -- a launch chain shorter by a scan, a select and a
		// memset, and ~10 k instead of ~32 k candidates in the full scan
		const uint32_t n_groups = (uint32_t)((h->n + kRowsPerGroup - 1) / kRowsPerGroup);
		const uint32_t sel = n_groups / 16;
		uint32_t blocks = std::min<uint32_t>({128u, cap / 16, sel / 8});
		if (blocks * 16 >= 8 * k && blocks >= 16) {
			Level l0;
			l0.n_groups_sel = sel;
			l0.group_stride = 16;
			l0.classmin_blocks = blocks;
			levels.clear();
			levels.push_back(l0);
			levels.push_back(Level{n_groups, 1, 0});
		}
	}
	if (!flags_clean) {  // overflow count and statistics (words 1, 2: uint8 / fp16 range flags, kept)
		HIP_TRY(h, hipMemsetAsync(h->d_overflow, 0, sizeof(uint32_t), st));
		HIP_TRY(h, hipMemsetAsync(h->d_total, 0, sizeof(unsigned long long) * 2, st));
	}
	flags_clean = false;
	// fp16 form on a large index: ONE sampled pass (1/16 of the rows, class maxima per query,
	// scan_gemm_f16.hpp) gives the threshold of the full scan -- no direct level-0 scan, no
	// intermediate candidate lists and selects
	li_start = 0;
	theta_ready = false;
	if (gvf && h->opt_sample_pass && levels.size() >= 2) {
		const int rs = sampled_pass_f16(h, gvf, m, k, ip, cus, h->d_tau[levels.size() & 1],
		                                h->d_tau_row[levels.size() & 1], st, &theta_ready);
		if (rs != EXPANN_OK)
			return rs;
		if (theta_ready)
			li_start = levels.size() - 1;
	}
	return EXPANN_OK;
}

int SearchPass::run_level(size_t li) {
	const Level& L = levels[li];
	const bool first = (li == 0), last = (li + 1 == levels.size());
	ScanParams sp{};
	sp.base = h->d_base;
	sp.n_rows = (uint32_t)h->n;
	sp.n_groups_sel = L.n_groups_sel;
	sp.group_stride = L.group_stride;
	sp.n_qtiles = n_qtiles;
	sp.queries = d_queries;
	sp.m = (uint32_t)m;
	sp.tau = first ? nullptr : h->d_tau[(li + 1) & 1];
	sp.tau_row = first ? nullptr : h->d_tau_row[(li + 1) & 1];
	sp.cand_cnt = h->d_cnt;
	sp.cand = h->d_cand;
	sp.cap = cap;
	// ~16 workgroups per CU in total, at least 8 groups (128 rows) per workgroup
	uint32_t target_chunks = (uint32_t)std::max<long>(1, (16L * cus + n_qtiles - 1) / n_qtiles);
	uint32_t max_chunks = std::max<uint32_t>(1, L.n_groups_sel / 8);
	uint32_t n_chunks = std::min(target_chunks, max_chunks);
	if (first && L.classmin_blocks) {
		n_chunks = L.classmin_blocks;
		sp.classmin = 1;
	}
	sp.groups_per_block = (L.n_groups_sel + n_chunks - 1) / n_chunks;
	n_chunks = (L.n_groups_sel + sp.groups_per_block - 1) / sp.groups_per_block;
	if (!first && !theta_ready)
		HIP_TRY(h, hipMemsetAsync(h->d_cnt, 0, sizeof(uint32_t) * m, st));
	const bool use_gemm = gv && !first;
	const bool timed = last && h->profiling && h->ev_used < kEventPairs;
	uint32_t passes = n_qtiles;
	const char* kname = sv->name;
	uint32_t qt_used = (uint32_t)sv->tq;
	if (use_gemm) {
		// theta_q = tau_q - ||q||^2 (1-eps), then the MFMA filter over 128-row tiles
		const float f16_abs = gvf ? std::ldexp(1.0f, -24) / h->f16_scale * std::sqrt((float)h->dim)
		                          : 0.0f;
		if (gvf && theta_ready)
			;
		else if (gvf)
			hipLaunchKernelGGL(f16_terms_kernel, dim3((uint32_t)((m + kBlock - 1) / kBlock)),
			                   dim3(kBlock), 0, st, h->d_qnrm.as<const float>(), (uint32_t)m,
			                   gemm_f16_filter_eps(h->dim), f16_abs, (const float*)sp.tau,
			                   0.5f * h->f16_scale * h->f16_scale, h->d_theta, ip ? 1 : 0,
			                   ip ? h->d_qscale.as<const float>() : (const float*)nullptr);
		else
			hipLaunchKernelGGL(gv->theta, dim3((uint32_t)((m + kRowsPerGroup - 1) / kRowsPerGroup)),
			                   dim3(kBlock), 0, st, (const float*)d_queries, (uint32_t)m,
			                   (const float*)sp.tau,
			                   1.0f - (gvb ? gemm_bf16_filter_eps(h->dim) : gemm_filter_eps(h->dim)),
			                   h->d_theta);
		GemmScanParams gp{};
		gp.base = (const float*)h->d_base;
		gp.bnorm = h->d_bnorm;
		gp.n_rows = (uint32_t)h->n;
		const uint32_t n_tiles = (uint32_t)((h->n + kGemmTB - 1) / kGemmTB);
		gp.n_tiles_sel = last ? n_tiles
		                      : std::min(n_tiles, (L.n_groups_sel * kRowsPerGroup + kGemmTB - 1) / kGemmTB);
		gp.tile_stride = std::max<uint32_t>(1, n_tiles / gp.n_tiles_sel);
		const uint32_t tq_wg = gvf ? (uint32_t)gvf->wgq : (gvb ? kGemmBf16TQ : kGemmTQ);
		uint32_t tq_small = 0;
		gp.n_qtiles = (uint32_t)((m + tq_wg - 1) / tq_wg);
		gp.queries = (const float*)d_queries;
		gp.theta = h->d_theta;
		gp.m = (uint32_t)m;
		gp.cand_cnt = h->d_cnt;
		gp.cand = h->d_cand;
		gp.cap = cap;
		// One workgroup per CU is resident (128 KiB of LDS), so the launch runs in rounds
		// of `cus` workgroups: pick the row-chunk count that minimises
		// rounds x (steps per workgroup + ~2 steps of prologue).
		uint32_t gchunks = pick_row_chunks(gp.n_tiles_sel, gp.n_qtiles, (uint32_t)cus, 2.0, 4, 1024, 0, nullptr);
		gp.tiles_per_block = (gp.n_tiles_sel + gchunks - 1) / gchunks;
		gchunks = (gp.n_tiles_sel + gp.tiles_per_block - 1) / gp.tiles_per_block;
		if (timed)
			HIP_TRY(h, hipEventRecord(h->ev[h->ev_used][0], st));
		if (gvf) {
			const int rl = launch_scan_f16(h, gvf, L.n_groups_sel * kRowsPerGroup, last, m, cus, ip, cap, st, &kname,
			                               &gp.n_qtiles, &tq_small);
			if (rl != EXPANN_OK)
				return rl;
		} else if (gvb) {
			GemmBf16Params bp{};
			bp.base_split = h->d_base_split;
			bp.bnorm = h->d_bnorm_bf;
			bp.n_rows = gp.n_rows;
			bp.n_tiles_sel = gp.n_tiles_sel;
			bp.tile_stride = gp.tile_stride;
			bp.tile_run = 1;
			if (!last && gp.tile_stride >= 8) {
				// sampled level: runs of 8 consecutive tiles (one 2 MiB page each at d=128)
				// instead of isolated tiles, same number of tiles
				bp.tile_run = 8;
				bp.n_tiles_sel = (gp.n_tiles_sel / 8) * 8;
				if (bp.n_tiles_sel == 0) {
					bp.n_tiles_sel = gp.n_tiles_sel;
					bp.tile_run = 1;
				}
			}
			bp.tiles_per_block = gp.tiles_per_block;
			bp.n_qtiles = gp.n_qtiles;
			bp.queries_split = h->d_q_split;
			bp.theta = gp.theta;
			bp.m = gp.m;
			bp.cand_cnt = gp.cand_cnt;
			bp.cand = gp.cand;
			bp.cap = gp.cap;
			bp.debug = (uint32_t)h->opt_debug;
			hipLaunchKernelGGL(gvb->scan, dim3(gchunks * gp.n_qtiles), dim3(kGemmThreads),
			                   2 * kGemmTB * h->dim * sizeof(float), st, bp);
			kname = gvb->name;
		} else {
			return h->fail(EXPANN_ERR_UNSUPPORTED, "no GEMM-form scan kernel for this index");  // (choose_kernels rules it out)
		}
		passes = gp.n_qtiles;
		qt_used = tq_small ? tq_small : tq_wg;  // (launch_scan_f16 reports the tile it used)
	} else if (gvi && !first) {
		hipLaunchKernelGGL(gvi->theta, dim3((uint32_t)((m + kRowsPerGroup - 1) / kRowsPerGroup)),
		                   dim3(kBlock), 0, st, d_queries, (uint32_t)m, (const float*)sp.tau,
		                   h->d_theta.as<int>(), h->d_qself);
		const uint32_t tb = (uint32_t)gemm_i8_tb(h->dim);
		GemmI8Params gp{};
		gp.base = h->d_base;
		gp.bias = h->d_bias_i;
		gp.n_rows = (uint32_t)h->n;
		const uint32_t n_tiles = (uint32_t)((h->n + tb - 1) / tb);
		gp.n_tiles_sel = last ? n_tiles
		                      : std::min(n_tiles, (L.n_groups_sel * kRowsPerGroup + tb - 1) / tb);
		gp.tile_stride = std::max<uint32_t>(1, n_tiles / gp.n_tiles_sel);
		gp.n_qtiles = (uint32_t)((m + kGemmI8TQ - 1) / kGemmI8TQ);
		gp.queries = d_queries;
		gp.theta = h->d_theta.as<const int>();
		gp.qself = h->d_qself;
		gp.m = (uint32_t)m;
		gp.cand_cnt = h->d_cnt;
		gp.cand = h->d_cand;
		gp.cap = cap;
		uint32_t gchunks = pick_row_chunks(gp.n_tiles_sel, gp.n_qtiles, (uint32_t)cus, 2.0, 4, 4096, 0, nullptr);
		gp.tiles_per_block = (gp.n_tiles_sel + gchunks - 1) / gchunks;
		gchunks = (gp.n_tiles_sel + gp.tiles_per_block - 1) / gp.tiles_per_block;
		if (timed)
			HIP_TRY(h, hipEventRecord(h->ev[h->ev_used][0], st));
		hipLaunchKernelGGL(gvi->scan, dim3(gchunks * gp.n_qtiles), dim3(kGemmThreads),
		                   2 * tb * h->dim, st, gp);
		passes = gp.n_qtiles;
		kname = gvi->name;
		qt_used = kGemmI8TQ;
	} else {
		if (timed)
			HIP_TRY(h, hipEventRecord(h->ev[h->ev_used][0], st));
		hipLaunchKernelGGL(sv->fn, dim3(n_chunks * n_qtiles), dim3(kBlock), 0, st, sp);
	}
	if (timed) {
		HIP_TRY(h, hipEventRecord(h->ev[h->ev_used][1], st));
		h->ev_used++;
	}
	launch_gather_logs(h, st);
	if (last && (timed || !h->profiling)) {
		h->prof.scan_launches++;
		h->prof.scan_rows += h->n;
		h->prof.scan_query_tiles += passes;
		h->prof.query_tile = qt_used;
		h->prof.levels = (uint32_t)(levels.size() - li_start + (li_start ? 1 : 0));
		std::snprintf(h->prof.scan_kernel, sizeof(h->prof.scan_kernel), "%s", kname);
	}
	HIP_TRY(h, hipGetLastError());

	SelectParams sel{};
	sel.cand = h->d_cand;
	sel.cand_cnt = first ? nullptr : h->d_cnt;
	sel.fixed_count = (first && L.classmin_blocks) ? n_chunks * 16 : L.n_groups_sel * kRowsPerGroup;
	sel.cap = cap;
	sel.k = (uint32_t)k;
	sel.id_offset = h->id_offset;
	sel.out_ids = last ? d_ids : nullptr;
	sel.out_dists = last ? d_dists : nullptr;
	sel.tau_out = last ? nullptr : h->d_tau[li & 1];
	sel.tau_prev = first ? nullptr : h->d_tau[(li + 1) & 1];
	sel.tau_row_out = last ? nullptr : h->d_tau_row[li & 1];
	sel.tau_row_prev = first ? nullptr : h->d_tau_row[(li + 1) & 1];
	sel.rerank_base = use_gemm ? (const float*)h->d_base : nullptr;
	sel.rerank_queries = use_gemm ? (const float*)d_queries : nullptr;
	sel.dim = (uint32_t)h->dim;
	sel.metric_ip = ip ? 1u : 0u;
	// (inner product, fp16 form only: the approximate key is off by at most E_q + E_b in
	// total, half the L2 margins)
	const float pr = ip ? 1.0f : 2.0f;
	sel.prune_eps = use_gemm ? (gvf ? pr * gemm_f16_filter_eps(h->dim)
	                                : (gvb ? gemm_bf16_filter_eps(h->dim) : gemm_filter_eps(h->dim)))
	                         : 0.0f;
	sel.prune_abs = (use_gemm && gvf)
	                    ? pr * std::ldexp(1.0f, -24) / h->f16_scale * std::sqrt((float)h->dim)
	                    : 0.0f;
	sel.bn_max = h->d_bnmax ? h->d_bnmax + (gvf ? 2 : (gvb ? 1 : 0)) : nullptr;
	sel.qnrm = (use_gemm && gvf) ? h->d_qnrm : nullptr;
	sel.overflow = h->d_overflow;
	if (last && sel.cand_cnt && h->profiling)  // statistics: candidates of the full scan
		hipLaunchKernelGGL(sum_u32_kernel, dim3(1), dim3(1024), 0, st, sel.cand_cnt, (uint32_t)m,
		                   h->d_total);
	if (m <= 64) {
		sel.wave0_short = 1;  // latency mode: one launch, wave 0 orders the short lists
	} else if (sel.rerank_base && sel.cand_cnt) {
		// short lists (the usual case after a GEMM-form scan): one wave per query
		launch_select_wave(sel, m, cap, st, (uint32_t)(1.2 * (double)k * sample_frac_for(h, k)));
	}
	hipLaunchKernelGGL(select_topk_kernel, dim3((uint32_t)m), dim3(kBlock),
	                   sizeof(uint64_t) * cap + 16, st, sel);
	HIP_TRY(h, hipGetLastError());
	return EXPANN_OK;
}

int SearchPass::check(int attempt, Next* next) {
	*next = kDone;
	// overflow check (the only host sync of a search); flags and statistics are contiguous
	if (defer_flags(h, st, attempt)) {  // deferred check: expann_sync reads the flags
		float sc = gvf ? h->f16_scale : 0.0f;  // (fp16 form: the range check of max |q| needs the scale)
		std::memcpy(&h->h_flag_ring[8 * (h->async_pending - 1) + 6], &sc, sizeof(float));
		return EXPANN_OK;
	}
	HIP_TRY(h, hipMemcpyAsync(h->h_flags, h->d_overflow, sizeof(uint32_t) * 4 + sizeof(unsigned long long),
	                          hipMemcpyDeviceToHost, st));
	HIP_TRY(h, hipStreamSynchronize(st));
	unsigned long long tot;
	std::memcpy(&tot, h->h_flags + 4, sizeof(tot));
	h->prof.candidates = tot;
	if (std::getenv("EXPANN_DEBUG_LISTS")) {  // (diagnostics: the distribution of the candidate lists' lengths)
		std::vector<uint32_t> cnt(m);
		HIP_TRY(h, hipMemcpy(cnt.data(), h->d_cnt, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
		std::sort(cnt.begin(), cnt.end());
		std::fprintf(stderr, "[lists] attempt %d cap %u: min %u median %u p90 %u p99 %u max %u, overflow flag %u\n", attempt, cap,
		             cnt.front(), cnt[m / 2], cnt[m * 9 / 10], cnt[m * 99 / 100], cnt.back(), h->h_flags[0]);
	}
	if (gvf) {  // queries outside the fp16 range of this index: redo with the bf16x3 form
		float qmax;  // bit pattern of max |q| (flags word 2, read back with the overflow flags)
		std::memcpy(&qmax, &h->h_flags[2], sizeof(float));
		if (!(qmax * h->f16_scale <= 60000.0f)) {
			no_f16 = true;
			h->prof.retries++;
			*next = kRestart;
			return EXPANN_OK;
		}
	}
	if (h->strict_u8 && (h->h_flags[1] != 0 || h->h_flags[3] != 0))
		return kStrictReject;
	if (h->dtype == EXPANN_DTYPE_U8 && h->h_flags[1] != 0)
		return h->fail(EXPANN_ERR_UNSUPPORTED,
		               std::to_string(h->h_flags[1]) +
		                   " query values outside [0,255]: the uint8 metric "
		                   "(dist2_compressed) is only defined for 8-bit valued queries");
	if (h->h_flags[0] == 0)
		return EXPANN_OK;
	// some candidate list overflowed: retry with 4x the capacity
	h->prof.retries++;
	// (a hit log of the 16x16x32 form that overflowed counts like a list: the logs grow with the lists)
	if ((cap >= kMaxCap || attempt >= 3) && (gv || gvi) && h->opt_scan_kernel == 0) {
		// the GEMM forms cannot break exact ties by row number; the direct scan can
		force_direct = true;
		*next = kRestart;
		return EXPANN_OK;
	}
	if (cap >= kMaxCap || attempt >= 3)
		return h->fail(EXPANN_ERR_OVERFLOW,
		               "candidate lists overflowed (" + std::to_string(h->h_flags[0]) +
		                   " queries) at capacity " + std::to_string(cap) +
		                   "; the data has more near-ties than the threshold filter supports");
	cap = std::min(cap * 4, kMaxCap);
	*next = kRetry;
	return EXPANN_OK;
}

int SearchPass::run() {
	for (;;) {  // (a restart: another kernel family serves the same queries)
		int rc = choose_kernels();
		if (rc != EXPANN_OK)
			return rc;
		bool restart = false;
		rc = prepare_queries(&restart);
		if (rc != EXPANN_OK)
			return rc;
		if (restart)
			continue;
		Next next = kDone;
		for (int attempt = 0;; ++attempt) {  // (a retry: the same family with larger candidate lists)
			rc = ensure_workspace(h, m, cap);
			if (rc == EXPANN_OK)
				rc = plan_thresholds();
			for (size_t li = li_start; rc == EXPANN_OK && li < levels.size(); ++li)
				rc = run_level(li);
			if (rc == EXPANN_OK)
				rc = check(attempt, &next);
			if (rc != EXPANN_OK || next != kRetry)
				break;
		}
		if (rc != EXPANN_OK || next == kDone)
			return rc;
	}
}

// One pipeline pass over <= kMaxQueriesPerPass queries (device pointers).
int search_pass(expann_index* h, const void* d_queries, size_t m, size_t k, uint64_t* d_ids,
                float* d_dists, hipStream_t st) {
	const bool ip = (h->metric == EXPANN_METRIC_IP);
	if (h->dtype == EXPANN_DTYPE_F32 && !ip && h->opt_scan_kernel == 0 && h->opt_u8_exact &&
	    (h->dim == 128 || h->dim == 256) && m >= 96 && h->n >= 65536 && k <= 256 && h->u8_exact >= 0) {
		// rows that are all integers in [0, 255] (SIFT): with integer queries the uint8 engine's
		// exact integer scores ARE the fp32 scores (d * 255^2 < 2^24: every partial sum of the
		// reference is an exactly represented integer), at the 8-bit kernels' speed
		if (h->u8_exact == 0) {
			const int rb = build_u8_shadow(h, st);
			if (rb != EXPANN_OK)
				return rb;
		}
		if (h->u8_exact == 1) {
			const int rq = search_pass(h->u8_shadow, d_queries, m, k, d_ids, d_dists, st);
			const int ra = absorb_shadow_profile(h);
			if (rq == EXPANN_OK)
				return ra;
			if (rq != kStrictReject)
				return h->fail(rq, h->u8_shadow->err);
		}
	}
	ScanSel svs;
	if (h->dtype == EXPANN_DTYPE_F32) {
		const ScanVariant* v = pick_scan_f32(h->dim, ip, m, h->opt_query_tile);
		if (v)
			svs = ScanSel{v->fn, v->tq, v->name};
	} else {
		const ScanI8Variant* v = pick_scan_i8(h->dim, h->int_mode, m, h->opt_query_tile);
		if (v)
			svs = ScanSel{v->fn, v->tq, v->name};
	}
	if (!svs.fn)
		return h->fail(EXPANN_ERR_UNSUPPORTED,
		               "no scan kernel for dim " + std::to_string(h->dim) +
		                   " / query_tile " + std::to_string(h->opt_query_tile));
	if (h->dtype == EXPANN_DTYPE_U8) {
		// fp32 queries -> uint8 (trunc); values outside [0,255] are counted and rejected below
		const size_t nv = m * (size_t)h->dim;
		HIP_TRY(h, grow_ws(h, h->d_q8, nv));
		int rcw = ensure_workspace(h, m, 2048);
		if (rcw != EXPANN_OK)
			return rcw;
		HIP_TRY(h, hipMemsetAsync(h->d_overflow + 1, 0, sizeof(uint32_t), st));
		HIP_TRY(h, hipMemsetAsync(h->d_overflow + 3, 0, sizeof(uint32_t), st));
		hipLaunchKernelGGL(u8_query_prep_kernel, dim3((uint32_t)((nv + kBlock - 1) / kBlock)),
		                   dim3(kBlock), 0, st, (const float*)d_queries, nv, h->d_q8,
		                   h->d_overflow + 1, h->strict_u8 ? h->d_overflow + 3 : (uint32_t*)nullptr);
		HIP_TRY(h, hipGetLastError());
		d_queries = h->d_q8;
	}
	uint32_t cap = h->opt_cand_capacity > 0
	                   ? (uint32_t)h->opt_cand_capacity
	                   : pow2ceil((uint32_t)std::max<size_t>(2048, 64 * k));
	cap = std::min(pow2ceil(cap), kMaxCap);
	if (h->opt_cand_capacity > 0 && (size_t)cap < 2 * k)  // the option is a starting size, never below 2k
		cap = std::min(pow2ceil((uint32_t)(2 * k)), kMaxCap);
	if ((size_t)cap < 2 * k)
		return h->fail(EXPANN_ERR_UNSUPPORTED, "k too large for the candidate buffers (k <= " +
		                                           std::to_string(kMaxCap / 2) + ")");
	const int cus = num_cus(h->device);
	if (const GemmI8qVariant* gq = pick_gemm_i8q(h, m, k)) {
		const int rq = search_i8q(h, gq, d_queries, m, k, d_ids, d_dists, st, cap);
		if (rq != kRetryGeneric)
			return rq;
	}
	SearchPass pass{h, d_queries, m, k, d_ids, d_dists, st, ip, cus, cap, svs};
	return pass.run();
}

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------
extern "C" {

int expann_abi_version(void) { return EXPANN_ABI_VERSION; }

int expann_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int expann_create(int dim, int dtype, int metric, int device, expann_index** out) {
	if (!out) {
		g_create_error = "out == NULL";
		return EXPANN_ERR_INVALID_ARG;
	}
	*out = nullptr;
	if (dim <= 0 || dim % 16 != 0) {
		g_create_error = "dim must be a positive multiple of 16 (the reference kernels require it)";
		return EXPANN_ERR_INVALID_ARG;
	}
	int int_mode = -1;
	if (dtype == EXPANN_DTYPE_F32) {
		if (metric != EXPANN_METRIC_L2 && metric != EXPANN_METRIC_IP) {
			g_create_error = "metric must be EXPANN_METRIC_L2 or EXPANN_METRIC_IP for f32 rows";
			return EXPANN_ERR_INVALID_ARG;
		}
	} else if (dtype == EXPANN_DTYPE_U8) {
		if (metric != EXPANN_METRIC_L2) {
			g_create_error = "uint8 rows support EXPANN_METRIC_L2 only (dist2_compressed)";
			return EXPANN_ERR_INVALID_ARG;
		}
		int_mode = kU8L2;
	} else if (dtype == EXPANN_DTYPE_I16) {
		if (metric != EXPANN_METRIC_L2) {
			g_create_error = "int16 rows support EXPANN_METRIC_L2 only (src/distance.h:14-27 semantics)";
			return EXPANN_ERR_INVALID_ARG;
		}
		int_mode = kI16L2Ref;
	} else if (dtype == EXPANN_DTYPE_I8) {
		if (metric == EXPANN_METRIC_L2)
			int_mode = kI8L2;
		else if (metric == EXPANN_METRIC_IP)
			int_mode = kI8IP;
		else if (metric == EXPANN_METRIC_L2_I8_REFCOMPAT)
			int_mode = kI8L2Ref;
		else {
			g_create_error = "unknown metric for int8 rows";
			return EXPANN_ERR_INVALID_ARG;
		}
	} else {
		g_create_error = "unknown dtype";
		return EXPANN_ERR_INVALID_ARG;
	}
	if (int_mode >= 0 && dim % 64 != 0) {
		g_create_error = "8-bit rows need dim % 64 == 0 (src/distance.h:29-53, antitopo_engine.h:726)";
		return EXPANN_ERR_INVALID_ARG;
	}
	int ndev = expann_device_count();
	if (ndev <= 0) {
		g_create_error = "no HIP device visible: libexpann_hip has no CPU fallback";
		return EXPANN_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= ndev) {
		g_create_error = "device index out of range";
		return EXPANN_ERR_INVALID_ARG;
	}
	if ((int_mode < 0 && !pick_scan_f32(dim, metric == EXPANN_METRIC_IP, 1, 0)) ||
	    (int_mode >= 0 && !pick_scan_i8(dim, int_mode, 1, 0))) {
		g_create_error = "unsupported dim " + std::to_string(dim) +
		                 " (built: f32 64,128,256,512,768,832,960,1024; 8-bit 64,128,256,768,832,960)";
		return EXPANN_ERR_UNSUPPORTED;
	}
	expann_index* h = new expann_index();
	h->dim = dim;
	h->dtype = dtype;
	h->metric = metric;
	h->device = device;
	h->elem = (dtype == EXPANN_DTYPE_F32) ? 4 : (dtype == EXPANN_DTYPE_I16 ? 2 : 1);
	h->q_elem = (dtype == EXPANN_DTYPE_I8) ? 1 : (dtype == EXPANN_DTYPE_I16 ? 2 : 4);
	h->int_mode = int_mode;
	if (const char* e = std::getenv("EXPANN_TAIL_CHUNKS"))
		h->opt_tail_chunks = std::atol(e);
	if (const char* e = std::getenv("EXPANN_PERSIST"))
		h->opt_persist = std::atol(e);
	if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) {
		g_create_error = "hipSetDevice/hipStreamCreate failed";
		delete h;
		return EXPANN_ERR_HIP;
	}
	// the select kernel sorts up to kMaxCap 8-byte keys in LDS (128 KiB of the CU's 160 KiB)
	if (hipFuncSetAttribute((const void*)select_topk_kernel,
	                        hipFuncAttributeMaxDynamicSharedMemorySize,
	                        (int)(sizeof(uint64_t) * kMaxCap + 16)) != hipSuccess) {
		g_create_error = "hipFuncSetAttribute(select_topk_kernel) failed";
		hipStreamDestroy(h->stream);
		delete h;
		return EXPANN_ERR_HIP;
	}
	for (const auto& v : kGemmF16X)
		if (v.d == dim &&  // (the sampled pass is launched with the scan's LDS size, never less than its own)
		    (hipFuncSetAttribute((const void*)v.scan, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds) != hipSuccess ||
		     hipFuncSetAttribute((const void*)v.sample, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds) != hipSuccess)) {
			g_create_error = "hipFuncSetAttribute(scan_gemm_f16x_kernel) failed";
			hipStreamDestroy(h->stream);
			delete h;
			return EXPANN_ERR_HIP;
		}
	for (const auto* tab : {kGemmF16XDbg})
		if (tab[0].d == dim)
			if (hipFuncSetAttribute((const void*)tab[0].scan, hipFuncAttributeMaxDynamicSharedMemorySize, tab[0].lds) !=
			    hipSuccess) {
				g_create_error = "hipFuncSetAttribute(scan_gemm_f16x_kernel) failed";
				hipStreamDestroy(h->stream);
				delete h;
				return EXPANN_ERR_HIP;
			}
	for (const auto& v : kGemmI8x)
		if (v.d == dim &&
		    (hipFuncSetAttribute((const void*)v.scan, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds) != hipSuccess ||
		     hipFuncSetAttribute((const void*)v.sample, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds) != hipSuccess)) {
			g_create_error = "hipFuncSetAttribute(scan_gemm_i8x_kernel) failed";
			hipStreamDestroy(h->stream);
			delete h;
			return EXPANN_ERR_HIP;
		}
	for (const auto& v : kGemmI8w)
		if (v.d == dim &&
		    (hipFuncSetAttribute((const void*)v.scan_w, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds_w) != hipSuccess ||
		     hipFuncSetAttribute((const void*)v.sample_w, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds_w) != hipSuccess)) {
			g_create_error = "hipFuncSetAttribute(scan_gemm_i8w_kernel) failed";
			hipStreamDestroy(h->stream);
			delete h;
			return EXPANN_ERR_HIP;
		}
	for (const auto& v : kGemmBf16)
		if (v.d == dim)
			if (hipFuncSetAttribute((const void*)v.scan, hipFuncAttributeMaxDynamicSharedMemorySize,
			                        2 * kGemmTB * dim * (int)sizeof(float)) != hipSuccess) {
				g_create_error = "hipFuncSetAttribute(scan_gemm_bf16x3_kernel) failed";
				hipStreamDestroy(h->stream);
				delete h;
				return EXPANN_ERR_HIP;
			}
	for (const auto& v : kGemmI8)
		if (v.d == dim && v.mode == int_mode)
			if (hipFuncSetAttribute((const void*)v.scan, hipFuncAttributeMaxDynamicSharedMemorySize,
			                        2 * gemm_i8_tb(dim) * dim) != hipSuccess) {
				g_create_error = "hipFuncSetAttribute(scan_gemm_i8_kernel) failed";
				hipStreamDestroy(h->stream);
				delete h;
				return EXPANN_ERR_HIP;
			}
	*out = h;
	return EXPANN_OK;
}

void expann_destroy(expann_index* h) {
	if (!h)
		return;
	hipSetDevice(h->device);
	if (h->stream) hipStreamSynchronize(h->stream);
	drop_u8_shadow(h);
	if (h->owns_base && h->d_base) hipFree(h->d_base);
	if (h->d_base_i8q && h->base_i8q_owned) hipFree(h->d_base_i8q);
	// (every other buffer is a DevPtr / PinPtr member: freed by `delete h` below, device still current)
	if (h->ev_created)
		for (int i = 0; i < kEventPairs; ++i) {
			hipEventDestroy(h->ev[i][0]);
			hipEventDestroy(h->ev[i][1]);
		}
	if (h->stream) hipStreamDestroy(h->stream);
	delete h;
}

const char* expann_last_error(const expann_index* h) {
	return h ? h->err.c_str() : g_create_error.c_str();
}

int expann_add(expann_index* h, const void* rows, size_t n) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!rows && n)
		return h->fail(EXPANN_ERR_INVALID_ARG, "rows == NULL");
	if (h->d_base)
		return h->fail(EXPANN_ERR_INVALID_ARG, "index already built");
	const size_t bytes = n * (size_t)h->dim * h->elem;
	const unsigned char* src = static_cast<const unsigned char*>(rows);
	h->staging.insert(h->staging.end(), src, src + bytes);
	h->n_staged += n;
	return EXPANN_OK;
}

int expann_build(expann_index* h) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (h->d_base)
		return h->fail(EXPANN_ERR_INVALID_ARG, "index already built");
	if (h->n_staged == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "build() on an empty index");
	if (h->n_staged >= (1ull << 32) - 32)
		return h->fail(EXPANN_ERR_UNSUPPORTED, "more than 2^32 rows per shard");
	HIP_TRY(h, hipSetDevice(h->device));
	HIP_TRY(h, hipMalloc(&h->d_base, h->staging.size()));
	h->owns_base = true;
	HIP_TRY(h, hipMemcpy(h->d_base, h->staging.data(), h->staging.size(), hipMemcpyHostToDevice));
	h->n = h->n_staged;
	std::vector<unsigned char>().swap(h->staging);
	// the fp16 copy every fp32 search of a built dim filters through: made here, inside the
	// reference's timed build span (basic_bench.h:63-71), not inside the first query
	if (h->dtype == EXPANN_DTYPE_F32 && h->n >= 4096 && h->opt_scan_kernel == 0)
		for (const auto& v : kGemmF16X)
			if (v.d == h->dim) {
				const int rc = ensure_f16(h, &v, h->stream);
				if (rc != EXPANN_OK)
					return rc;
			}
	return EXPANN_OK;
}

int expann_set_base_device(expann_index* h, const void* d_rows, size_t n, uint64_t id_offset) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!d_rows || n == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "empty device base");
	if (n >= (1ull << 32) - 32)
		return h->fail(EXPANN_ERR_UNSUPPORTED, "more than 2^32 rows per shard");
	if (h->owns_base && h->d_base)
		hipFree(h->d_base);
	drop_u8_shadow(h);
	// the derived copies belong to the rows they were made from
	h->d_bnorm.reset();
	h->d_bias_i.reset();
	h->d_bnorm_bf.reset();
	h->d_base_split.reset();
	h->d_base_f16.reset();
	if (h->d_base_i8q && h->base_i8q_owned)
		hipFree(h->d_base_i8q);
	h->d_base_i8q = nullptr;
	h->base_i8q_owned = false;
	h->d_bp_i8q.reset();
	h->d_bnorm_f16.reset();
	h->d_bns_f16.reset();
	h->f16_scale = 0.0f;
	h->d_base = const_cast<void*>(d_rows);
	h->owns_base = false;
	h->n = n;
	h->id_offset = id_offset;
	return EXPANN_OK;
}

size_t expann_size(const expann_index* h) { return h ? (h->d_base ? h->n : h->n_staged) : 0; }

int expann_search_device(expann_index* h, const void* d_queries, size_t m, size_t k,
                         uint64_t* d_ids, float* d_dists, void* stream) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->d_base)
		return h->fail(EXPANN_ERR_NOT_BUILT, "search before build()");
	if (k == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "k == 0");
	if (m == 0)
		return EXPANN_OK;
	if (!d_queries || !d_ids)
		return h->fail(EXPANN_ERR_INVALID_ARG, "NULL query/ids pointer");
	HIP_TRY(h, hipSetDevice(h->device));
	hipStream_t st = stream ? (hipStream_t)stream : h->stream;
	// all searches of a handle share one workspace: deferred searches still running on ANOTHER
	// stream are drained before this one is enqueued (their flag blocks stay in the ring for
	// expann_sync); on one stream the stream order does this
	if (h->async_pending > 0 && h->async_stream != st) {
		HIP_TRY(h, hipStreamSynchronize(h->async_stream));
		h->async_stream = st;
	}
	const size_t qbytes = (size_t)h->dim * h->q_elem;
	for (size_t q0 = 0; q0 < m; q0 += kMaxQueriesPerPass) {
		const size_t mm = std::min(kMaxQueriesPerPass, m - q0);
		int rc = search_pass(h, (const char*)d_queries + q0 * qbytes, mm, k, d_ids + q0 * k,
		                     d_dists ? d_dists + q0 * k : nullptr, st);
		if (rc != EXPANN_OK)
			return rc;
	}
	return EXPANN_OK;
}

int expann_sync(expann_index* h) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (h->async_pending == 0)
		return EXPANN_OK;
	HIP_TRY(h, hipSetDevice(h->device));
	const uint32_t n = h->async_pending;
	h->async_pending = 0;
	HIP_TRY(h, hipStreamSynchronize(h->async_stream));
	uint32_t bad = 0;
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t* f = h->h_flag_ring + 8 * i;
		float qmax, scale;
		std::memcpy(&qmax, &f[2], sizeof(float));
		std::memcpy(&scale, &f[6], sizeof(float));
		if (f[0] != 0 || (f[7] != 0 && f[1] != 0) || (scale > 0.0f && !(qmax * scale <= 60000.0f)))
			++bad;
		unsigned long long tot;
		std::memcpy(&tot, f + 4, sizeof(tot));
		h->prof.candidates = tot;
	}
	if (bad)
		return h->fail(EXPANN_ERR_OVERFLOW, std::to_string(bad) + " of " + std::to_string(n) +
		                                        " deferred searches need the synchronous retry "
		                                        "(candidate overflow or queries outside the filter's range): "
		                                        "repeat them with async_search = 0");
	return EXPANN_OK;
}

int expann_search(expann_index* h, const void* queries, size_t m, size_t k, uint64_t* ids,
                  float* dists) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->d_base)
		return h->fail(EXPANN_ERR_NOT_BUILT, "search before build()");
	if (k == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "k == 0");
	if (m == 0)
		return EXPANN_OK;
	if (!queries || !ids)
		return h->fail(EXPANN_ERR_INVALID_ARG, "NULL query/ids pointer");
	if (h->async_pending) {  // deferred searches first
		const int rs = expann_sync(h);
		if (rs != EXPANN_OK)
			return rs;
	}
	HIP_TRY(h, hipSetDevice(h->device));
	const size_t qbytes = m * (size_t)h->dim * h->q_elem;
	HIP_TRY(h, h->d_q.ensure(qbytes));
	if (m * k <= 16384 && qbytes <= (256u << 10) && m <= kMaxQueriesPerPass && h->opt_latency_mode) {
		const size_t ids_off = (qbytes + 63) / 64 * 64;
		const size_t dists_off = ids_off + sizeof(uint64_t) * m * k;
		const size_t need = dists_off + sizeof(float) * m * k;
		if (need > h->h_pin_bytes) {
			h->h_pin.reset();
			h->h_pin_bytes = 0;
			const size_t want = std::max<size_t>(need, 64u << 10);
			HIP_TRY(h, hipHostMalloc(&h->h_pin, want, hipHostMallocDefault));
			h->h_pin_bytes = want;
		}
		char* pin = h->h_pin.as<char>();
		std::memcpy(pin, queries, qbytes);
		// fp32 index with its fp16 copy in place and at most one workgroup of queries: the query
		// prep kernel reads them straight from pinned memory (and makes the device copy); otherwise
		// one async copy
		const void* dq = h->d_q;
		if (h->dtype == EXPANN_DTYPE_F32 && h->d_base_f16 && h->f16_scale > 0.0f && h->opt_scan_kernel == 0 &&
		    m <= (size_t)kRowsPerGroup && h->n >= 4096 && pick_gemm(h, m) != nullptr) {
			h->q_in_pinned_host = true;
			dq = pin;
		} else {
			HIP_TRY(h, hipMemcpyAsync(h->d_q, pin, qbytes, hipMemcpyHostToDevice, h->stream));
		}
		// (search_pass ends with the flag read-back and a stream sync: the results are complete)
		h->host_call = true;
		int rc = expann_search_device(h, dq, m, k, (uint64_t*)(pin + ids_off), (float*)(pin + dists_off),
		                              h->stream);
		h->host_call = false;
		if (h->q_in_pinned_host) {  // (a path that did not run the prep kernel read the queries over PCIe)
			h->q_in_pinned_host = false;
		}
		if (rc != EXPANN_OK)
			return rc;
		std::memcpy(ids, pin + ids_off, sizeof(uint64_t) * m * k);
		if (dists)
			std::memcpy(dists, pin + dists_off, sizeof(float) * m * k);
		return EXPANN_OK;
	}
	HIP_TRY(h, h->d_ids.ensure(sizeof(uint64_t) * m * k));
	HIP_TRY(h, h->d_dists.ensure(sizeof(float) * m * k));
	HIP_TRY(h, hipMemcpyAsync(h->d_q, queries, qbytes, hipMemcpyHostToDevice, h->stream));
	h->host_call = true;
	int rc = expann_search_device(h, h->d_q, m, k, h->d_ids, h->d_dists, h->stream);
	h->host_call = false;
	if (rc != EXPANN_OK)
		return rc;
	HIP_TRY(h, hipMemcpyAsync(ids, h->d_ids, sizeof(uint64_t) * m * k, hipMemcpyDeviceToHost,
	                          h->stream));
	if (dists)
		HIP_TRY(h, hipMemcpyAsync(dists, h->d_dists, sizeof(float) * m * k,
		                          hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	return EXPANN_OK;
}

int expann_merge_topk_strided_device(int device, const uint64_t* d_in_ids, const float* d_in_dists,
                                     size_t ids_stride, size_t dists_stride, size_t n_lists, size_t m,
                                     size_t k, uint64_t* d_out_ids, float* d_out_dists, void* stream) {
	if (!d_in_ids || !d_in_dists || !d_out_ids || !d_out_dists || n_lists == 0 || n_lists > 64 ||
	    k == 0) {
		g_create_error = "expann_merge_topk_device: bad arguments (1 <= n_lists <= 64)";
		return EXPANN_ERR_INVALID_ARG;
	}
	if (m == 0)
		return EXPANN_OK;
	if (hipSetDevice(device) != hipSuccess) {
		g_create_error = "hipSetDevice failed";
		return EXPANN_ERR_HIP;
	}
	if (m >= (1ull << 31) || n_lists * k >= (1ull << 31)) {
		g_create_error = "expann_merge_topk_device: m or n_lists * k beyond 2^31";
		return EXPANN_ERR_INVALID_ARG;
	}
	// one wave per query; the query's n_lists * k entries staged in LDS while they fit 48 KiB
	const size_t lds = n_lists * k * 12;
	if (lds <= (48u << 10))
		hipLaunchKernelGGL(merge_topk_kernel<true>, dim3((uint32_t)m), dim3(kWave), lds, (hipStream_t)stream, d_in_ids,
		                   d_in_dists, ids_stride, dists_stride, (uint32_t)n_lists, (uint32_t)m, (uint32_t)k, d_out_ids,
		                   d_out_dists);
	else
		hipLaunchKernelGGL(merge_topk_kernel<false>, dim3((uint32_t)m), dim3(kWave), 0, (hipStream_t)stream, d_in_ids,
		                   d_in_dists, ids_stride, dists_stride, (uint32_t)n_lists, (uint32_t)m, (uint32_t)k, d_out_ids,
		                   d_out_dists);
	if (hipGetLastError() != hipSuccess) {
		g_create_error = "merge_topk_kernel launch failed";
		return EXPANN_ERR_HIP;
	}
	return EXPANN_OK;
}

int expann_merge_topk_device(int device, const uint64_t* d_in_ids, const float* d_in_dists,
                             size_t n_lists, size_t m, size_t k, uint64_t* d_out_ids,
                             float* d_out_dists, void* stream) {
	return expann_merge_topk_strided_device(device, d_in_ids, d_in_dists, m * k, m * k, n_lists, m, k,
	                                        d_out_ids, d_out_dists, stream);
}

int expann_score_ids(expann_index* h, const void* query, const uint64_t* ids, size_t n_ids,
                     float cutoff, uint64_t* kept_ids, float* kept_scores, size_t* n_kept) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->d_base)
		return h->fail(EXPANN_ERR_NOT_BUILT, "score_ids before build()");
	if (!n_kept || (n_ids && (!query || !ids || !kept_ids || !kept_scores)))
		return h->fail(EXPANN_ERR_INVALID_ARG, "NULL pointer");
	*n_kept = 0;
	if (n_ids == 0)
		return EXPANN_OK;
	if (n_ids >= (1ull << 32))
		return h->fail(EXPANN_ERR_INVALID_ARG, "more than 2^32 ids");
	for (size_t i = 0; i < n_ids; ++i)
		if (ids[i] < h->id_offset || ids[i] - h->id_offset >= h->n)
			return h->fail(EXPANN_ERR_INVALID_ARG, "id out of range");
	HIP_TRY(h, hipSetDevice(h->device));
	DevBuf b_query, b_idl, b_sc, b_q8s, b_bad;
	const size_t qb = (size_t)h->dim * h->q_elem;
	HIP_TRY(h, b_query.alloc(qb));
	HIP_TRY(h, b_idl.alloc(sizeof(uint64_t) * n_ids));
	HIP_TRY(h, b_sc.alloc(sizeof(float) * n_ids));
	void *d_query = b_query.p, *d_idl = b_idl.p, *d_sc = b_sc.p, *d_q8s = nullptr;
	uint32_t* d_bad = nullptr;
	HIP_TRY(h, hipMemcpyAsync(d_query, query, qb, hipMemcpyHostToDevice, h->stream));
	HIP_TRY(h, hipMemcpyAsync(d_idl, ids, sizeof(uint64_t) * n_ids, hipMemcpyHostToDevice, h->stream));
	const uint32_t blocks = (uint32_t)((n_ids + kRowsPerGroup - 1) / kRowsPerGroup);
	uint32_t bad_host = 0;
	if (h->dtype == EXPANN_DTYPE_F32) {
		const ScoreVariant* sv = nullptr;
		for (const auto& v : kScoreF32)
			if (v.d == h->dim && v.ip == (h->metric == EXPANN_METRIC_IP))
				sv = &v;
		if (!sv)
			return h->fail(EXPANN_ERR_UNSUPPORTED, "no score kernel for this dim");
		ScoreIdsParams sp{h->d_base, d_query, (const uint64_t*)d_idl, h->id_offset,
		                  (uint32_t)n_ids, (float*)d_sc};
		hipLaunchKernelGGL(sv->fn, dim3(blocks), dim3(kBlock), 0, h->stream, sp);
	} else {
		const ScoreI8Variant* sv = nullptr;
		for (const auto& v : kScoreI8)
			if (v.d == h->dim && v.mode == h->int_mode)
				sv = &v;
		if (!sv)
			return h->fail(EXPANN_ERR_UNSUPPORTED, "no 8-bit score kernel for this dim");
		const void* qptr = d_query;
		if (h->dtype == EXPANN_DTYPE_U8) {
			HIP_TRY(h, b_q8s.alloc((size_t)h->dim));
			HIP_TRY(h, b_bad.alloc(sizeof(uint32_t)));
			d_q8s = b_q8s.p;
			d_bad = b_bad.as<uint32_t>();
			HIP_TRY(h, hipMemsetAsync(d_bad, 0, sizeof(uint32_t), h->stream));
			hipLaunchKernelGGL(u8_query_prep_kernel, dim3((uint32_t)((h->dim + kBlock - 1) / kBlock)),
			                   dim3(kBlock), 0, h->stream, (const float*)d_query, (size_t)h->dim,
			                   (uint8_t*)d_q8s, d_bad, (uint32_t*)nullptr);
			HIP_TRY(h, hipMemcpyAsync(&bad_host, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost,
			                          h->stream));
			qptr = d_q8s;
		}
		ScoreIdsI8Params sp{h->d_base, qptr, (const uint64_t*)d_idl, h->id_offset, (uint32_t)n_ids,
		                    (float*)d_sc};
		hipLaunchKernelGGL(sv->fn, dim3(blocks), dim3(kBlock), 0, h->stream, sp);
	}
	// the `d < cutoff` filter, order kept (src/quantizer.h:42-46): compacted on the device, only the
	// survivors and their count come back
	DevBuf b_kid, b_ksc, b_cnt;
	HIP_TRY(h, b_kid.alloc(sizeof(uint64_t) * n_ids));
	HIP_TRY(h, b_ksc.alloc(sizeof(float) * n_ids));
	HIP_TRY(h, b_cnt.alloc(sizeof(uint32_t)));
	FilterScoresParams fp{(const uint64_t*)d_idl, (const float*)d_sc, (uint32_t)n_ids, cutoff, b_kid.as<uint64_t>(),
	                      b_ksc.as<float>(), b_cnt.as<uint32_t>()};
	hipLaunchKernelGGL(filter_scores_kernel, dim3(1), dim3(1024), 0, h->stream, fp);
	HIP_TRY(h, hipGetLastError());
	uint32_t kept = 0;
	HIP_TRY(h, hipMemcpyAsync(&kept, b_cnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	if (bad_host)
		return h->fail(EXPANN_ERR_UNSUPPORTED, "query values outside [0,255] for the uint8 metric");
	if (kept) {
		HIP_TRY(h, hipMemcpyAsync(kept_ids, b_kid.p, sizeof(uint64_t) * kept, hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(h, hipMemcpyAsync(kept_scores, b_ksc.p, sizeof(float) * kept, hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(h, hipStreamSynchronize(h->stream));
	}
	*n_kept = kept;
	return EXPANN_OK;
}

// (graph search, the antitopo engine handle and the quantiser builds: expann_graph.hip)

int expann_set_profiling(expann_index* h, int enable) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	HIP_TRY(h, hipSetDevice(h->device));
	if (enable && !h->ev_created) {
		for (int i = 0; i < kEventPairs; ++i) {
			HIP_TRY(h, hipEventCreate(&h->ev[i][0]));
			HIP_TRY(h, hipEventCreate(&h->ev[i][1]));
		}
		h->ev_created = true;
	}
	h->profiling = enable != 0;
	h->ev_used = 0;
	h->prof = expann_profile{};
	h->prof_extra_ms = 0;
	if (h->u8_shadow)
		return expann_set_profiling(h->u8_shadow, enable);
	return EXPANN_OK;
}

int expann_get_profile(expann_index* h, expann_profile* out) {
	if (!h || !out)
		return EXPANN_ERR_INVALID_ARG;
	HIP_TRY(h, hipSetDevice(h->device));
	double ms = 0;
	for (int i = 0; i < h->ev_used; ++i) {
		HIP_TRY(h, hipEventSynchronize(h->ev[i][1]));
		float t = 0;
		HIP_TRY(h, hipEventElapsedTime(&t, h->ev[i][0], h->ev[i][1]));
		ms += t;
	}
	h->prof.scan_ms = ms + h->prof_extra_ms;
	// (launches beyond the event pool are neither timed nor counted while profiling)
	*out = h->prof;
	h->ev_used = 0;
	h->prof_extra_ms = 0;
	h->prof = expann_profile{};
	return EXPANN_OK;
}

int expann_set_option(expann_index* h, const char* name, long value) {
	if (!h || !name)
		return EXPANN_ERR_INVALID_ARG;
	if (!std::strcmp(name, "query_tile"))
		h->opt_query_tile = value;
	else if (!std::strcmp(name, "cand_capacity"))
		h->opt_cand_capacity = value;
	else if (!std::strcmp(name, "debug"))
		h->opt_debug = value;
	else if (!std::strcmp(name, "scan_kernel"))
		h->opt_scan_kernel = value;
	else if (!std::strcmp(name, "f16x") || !std::strcmp(name, "i8x") || !std::strcmp(name, "i8w")) {
		// (rounds 1-2: A/B switches between the 32x32 kernels and the 16x16 ones; the 32x32 kernels are gone)
		if (value == 0)
			return h->fail(EXPANN_ERR_UNSUPPORTED, std::string(name) + " = 0: the 32 x 32 MFMA forms of rounds 1-2 were removed in round 3");
	}
	else if (!std::strcmp(name, "tail_chunks"))
		h->opt_tail_chunks = value;
	else if (!std::strcmp(name, "persist"))
		h->opt_persist = value;
	else if (!std::strcmp(name, "ip_rescale"))
		h->opt_ip_rescale = value;
	else if (!std::strcmp(name, "sample_pass"))
		h->opt_sample_pass = value;
	else if (!std::strcmp(name, "u8_exact"))
		h->opt_u8_exact = value;
	else if (!std::strcmp(name, "latency_mode"))
		h->opt_latency_mode = value;
	else if (!std::strcmp(name, "async_search"))
		h->opt_async = value;
	else if (!std::strcmp(name, "scan_chunks"))
		h->opt_scan_chunks = value;
	else if (!std::strcmp(name, "xcd_tolerance"))
		h->opt_xcd_tolerance = value < 0 ? 0 : value;
	else if (!std::strcmp(name, "sample_run"))
		h->opt_sample_run = value < 1 ? 1 : (value > 64 ? 64 : value);
	else if (!std::strcmp(name, "sample_frac"))
		h->opt_sample_frac = value < 0 ? 0 : value;
	else if (!std::strcmp(name, "sample_ratio"))
		h->opt_sample_ratio = value < 2 ? 2 : value;
	else
		return h->fail(EXPANN_ERR_INVALID_ARG, std::string("unknown option ") + name);
	return EXPANN_OK;
}

}  // extern "C"
