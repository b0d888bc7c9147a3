// expann_sharded.hip -- the row-sharded brute-force engine behind the C ABI (expann_sharded_*,
// include/expann_hip.h): SURVEY 8b's `create(..., const int* devices, n_dev, ...)` and 8e's
// partition, exchange and merge.
//
//   rows        shard r holds the contiguous range [r * ceil(N/G), min(N, (r+1) * ceil(N/G)));
//               ids are global row numbers, so the per-shard (score, id) order is the global one
//   scan        every shard searches ALL queries on its own device / stream (an expann_index per
//               shard, deferred check: nothing waits on the host between the scan and the exchange)
//   exchange    pattern "slices" (default): the queries are cut into G slices [j * ceil(m/G), ...);
//               an ALL-TO-ALL sends rank j the slice-j part of every shard's per-query top-k (k x 12 B
//               per query), rank j merges its slice (merge_topk, select.hpp) -- m/G queries, G lists --
//               and, in the rank form, the merged slices are all-gathered so that every rank ends
//               with the full result: 2 (G-1)/G x 12 m k bytes arrive per rank instead of the
//               (G-1) x 12 m k of
//               pattern "all-gather" (round 2's only one, still selectable): ONE ncclAllGather of the
//               whole chunk [ids m*k u64 | dists m*k f32] per shard, every rank merges all m queries
//   transport   RCCL over xGMI (ncclSend / ncclRecv groups, ncclAllGather), plain device copies
//               (several shards on one device, which RCCL refuses: the one-GPU test box), or the
//               caller's function (rank form: another fabric, rehearsals)
//
// Two forms, same code path after the communicator exists:
//   * in-process (expann_sharded_create): one handle drives n_dev devices of this node.  Every
//     shard has its OWN host thread: a search posts the enqueue work of all shards to their
//     threads at once, so the host-side launch cost of a search (about a dozen launches per shard)
//     is paid in parallel and does not grow with the device count -- at C2 / G = 8 the per-shard
//     GPU work is ~0.4 ms, less than one thread's serial enqueue over 8 devices.  ncclCommInitAll;
//     device r merges the r-th slice of the queries (host API: and copies it to the caller's host
//     buffers; expann_sharded_search_devices: leaves it in the caller's device buffer on device r).
//   * one rank of a one-process-per-GPU job (expann_sharded_create_rank): ncclCommInitRank with
//     a unique id the launcher distributed; every rank ends with the full [m][k] result in
//     device memory, on the caller's stream.  A rank whose row range is empty (N < G * (G-1) under
//     the ceil partition) still takes part in every collective with an all-padding chunk.
//
// Everything below the exchange is the public C ABI of the single-device index (expann_create,
// expann_set_base_device, expann_search_device, expann_sync, expann_merge_topk_strided_device):
// this file holds no kernels.
#include "../../include/expann_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "host_common.hpp"

using namespace expann;

namespace {

// one host thread per shard of the in-process form: runs the jobs posted to it, one at a time
class Worker {
public:
	Worker() : th_([this] { loop(); }) {}
	~Worker() {
		{
			std::lock_guard<std::mutex> lk(mu_);
			quit_ = true;
		}
		cv_.notify_all();
		th_.join();
	}
	void post(std::function<int()> job) {
		{
			std::lock_guard<std::mutex> lk(mu_);
			job_ = std::move(job);
			busy_ = true;
		}
		cv_.notify_all();
	}
	int wait() {
		std::unique_lock<std::mutex> lk(mu_);
		cv_.wait(lk, [this] { return !busy_; });
		return rc_;
	}

private:
	void loop() {
		for (;;) {
			std::function<int()> job;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [this] { return quit_ || (busy_ && job_); });
				if (quit_)
					return;
				job = std::move(job_);
				job_ = nullptr;
			}
			const int rc = job();
			{
				std::lock_guard<std::mutex> lk(mu_);
				rc_ = rc;
				busy_ = false;
			}
			cv_.notify_all();
		}
	}
	std::mutex mu_;
	std::condition_variable cv_;
	std::function<int()> job_;
	bool busy_ = false, quit_ = false;
	int rc_ = 0;
	std::thread th_;
};

struct Shard {
	int device = 0;
	expann_index* idx = nullptr;
	hipStream_t stream = nullptr;
	void* d_rows = nullptr;     // owned copy of this shard's rows (build()) or nullptr (adopted)
	size_t n = 0;
	uint64_t id_offset = 0;
	bool adopted_empty = false;         // rank form: this rank's row range is empty
	void* d_q = nullptr;        // queries on this device (in-process form, host API)
	size_t q_bytes = 0;
	unsigned char* mine = nullptr;      // [ids m*k u64 | dists m*k f32] of this shard's search
	unsigned char* gathered = nullptr;  // all-gather: G chunks; slices: [G][per*k] ids, then [G][per*k] dists
	unsigned char* merged = nullptr;    // merged slice [ids per*k | dists per*k]
	size_t mine_alloc = 0, gathered_alloc = 0, merged_alloc = 0;
	hipEvent_t ev_search = nullptr, ev_done = nullptr;
	bool done_recorded = false;         // ev_done holds the readers of the previous search's `mine`
	ncclComm_t comm = nullptr;
	std::unique_ptr<Worker> worker;     // in-process form
	std::string err;                    // failure of this shard's last job
	int fail(int code, const std::string& msg) {
		err = msg;
		return code;
	}
};

size_t chunk_bytes(size_t m, size_t k) { return (m * k * 12 + 15) / 16 * 16; }

// slice j of m queries cut G ways: [min(m, j * per), min(m, (j+1) * per)), per = ceil(m / G)
struct Slices {
	size_t m, G, per;
	Slices(size_t m_, size_t G_) : m(m_), G(G_), per((m_ + G_ - 1) / G_) {}
	size_t lo(size_t j) const { return std::min(m, j * per); }
	size_t hi(size_t j) const { return std::min(m, (j + 1) * per); }
	size_t cnt(size_t j) const { return hi(j) - lo(j); }
};

}  // namespace

struct expann_sharded {
	int dim = 0, dtype = 0, metric = 0;
	size_t elem = 4, q_elem = 4;
	std::vector<Shard> shards;          // in-process: one per device; rank form: exactly one
	int rank = 0, world = 1;            // rank form (in-process: world = shards in use)
	expann_exchange_fn exchange_fn = nullptr;      // rank form: the caller's all-gather in place of RCCL's
	void* exchange_ctx = nullptr;
	expann_alltoallv_fn alltoallv_fn = nullptr;    // rank form: the caller's all-to-all-v
	void* alltoallv_ctx = nullptr;
	bool rank_form = false;
	std::vector<unsigned char> staging;  // add() rows until build()
	size_t n_staged = 0, n_total = 0;
	int n_active = 0;                   // shards that hold rows (in-process)
	int exchange = 0;                   // option: 0 auto, 1 RCCL, 2 device copies
	int exchange_used = 0;              // 1 RCCL, 2 copies, 3 the caller's function, 0 none (single shard)
	int pattern = 0;                    // option: 0 auto (slices), 1 all-gather of whole chunks, 2 all-to-all of query slices
	bool comm_ready = false;
	long opt_async = 1;
	long opt_threads = 1;               // in-process form: one enqueue thread per shard (0: the caller's thread does all)
	uint64_t searches = 0, retries = 0;
	double last_enqueue_ms = 0;
	bool devices_pending = false;       // expann_sharded_search_devices calls not yet validated by _sync
	mutable std::string err;
	int fail(int code, const std::string& msg) const {
		err = msg;
		return code;
	}
	int pattern_used() const { return pattern == 1 ? 1 : 2; }
};

#define NCCL_TRY(h, expr)                                                                    \
	do {                                                                                     \
		ncclResult_t _r = (expr);                                                            \
		if (_r != ncclSuccess)                                                               \
			return (h)->fail(EXPANN_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
	} while (0)
#define SUB_TRY(h, s, expr)                                                                  \
	do {                                                                                     \
		int _rc = (expr);                                                                    \
		if (_rc != EXPANN_OK)                                                                \
			return (h)->fail(_rc, std::string("shard on device ") + std::to_string((s).device) + ": " + \
			                          expann_last_error((s).idx));                            \
	} while (0)

namespace {

int common_create(int dim, int dtype, int metric, expann_sharded** out, expann_sharded*& h) {
	if (!out) {
		g_create_error = "out == NULL";
		return EXPANN_ERR_INVALID_ARG;
	}
	*out = nullptr;
	h = new expann_sharded();
	h->dim = dim;
	h->dtype = dtype;
	h->metric = metric;
	h->elem = (dtype == EXPANN_DTYPE_F32) ? 4 : (dtype == EXPANN_DTYPE_I16 ? 2 : 1);
	h->q_elem = (dtype == EXPANN_DTYPE_I8) ? 1 : (dtype == EXPANN_DTYPE_I16 ? 2 : 4);
	return EXPANN_OK;
}

int open_shard(expann_sharded* h, Shard& s, int device) {
	s.device = device;
	int rc = expann_create(h->dim, h->dtype, h->metric, device, &s.idx);
	if (rc != EXPANN_OK)
		return rc;  // (g_create_error holds the message)
	if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&s.stream) != hipSuccess ||
	    hipEventCreateWithFlags(&s.ev_search, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) != hipSuccess) {
		g_create_error = "hipSetDevice / hipStreamCreate / hipEventCreate failed on device " + std::to_string(device);
		return EXPANN_ERR_HIP;
	}
	return EXPANN_OK;
}

void close_shard(Shard& s) {
	s.worker.reset();  // (joins the shard's thread)
	if (s.idx || s.stream)
		(void)hipSetDevice(s.device);
	if (s.stream)
		(void)hipStreamSynchronize(s.stream);
	if (s.comm)
		ncclCommDestroy(s.comm);
	expann_destroy(s.idx);
	if (s.d_rows) (void)hipFree(s.d_rows);
	if (s.d_q) (void)hipFree(s.d_q);
	if (s.mine) (void)hipFree(s.mine);
	if (s.gathered) (void)hipFree(s.gathered);
	if (s.merged) (void)hipFree(s.merged);
	if (s.ev_search) (void)hipEventDestroy(s.ev_search);
	if (s.ev_done) (void)hipEventDestroy(s.ev_done);
	if (s.stream) (void)hipStreamDestroy(s.stream);
	s = Shard{};
}

// one buffer of a shard, grown on demand (the stream drained first: earlier searches may still read it)
int grow(Shard& s, hipStream_t st, unsigned char** p, size_t* have, size_t need) {
	if (need <= *have)
		return EXPANN_OK;
	HIP_TRY(&s, hipStreamSynchronize(st));
	if (st != s.stream)
		HIP_TRY(&s, hipStreamSynchronize(s.stream));
	if (*p) (void)hipFree(*p);
	*p = nullptr;
	*have = 0;
	HIP_TRY(&s, hipMalloc(p, need));
	*have = need;
	return EXPANN_OK;
}

// exchange buffers of one shard for m x k results from G ranks under `pattern`
int ensure_chunks(Shard& s, hipStream_t st, size_t m, size_t k, int G, int pattern) {
	const size_t cb = chunk_bytes(m, k);
	const Slices sl(m, (size_t)G);
	int rc = grow(s, st, &s.mine, &s.mine_alloc, cb);
	if (rc == EXPANN_OK)
		rc = grow(s, st, &s.gathered, &s.gathered_alloc,
		          pattern == 1 ? cb * (size_t)G : chunk_bytes((size_t)G * sl.per, k));
	if (rc == EXPANN_OK)
		rc = grow(s, st, &s.merged, &s.merged_alloc, chunk_bytes(sl.per, k));
	return rc;
}

// the communicator(s), once the set of shards in use is known
int ensure_comm(expann_sharded* h) {
	if (h->comm_ready)
		return EXPANN_OK;
	const int G = h->rank_form ? h->world : h->n_active;
	int mode = h->exchange;
	if (h->rank_form) {
		mode = 1;  // (the communicator was made by expann_sharded_create_rank)
	} else if (mode == 0) {
		bool distinct = true;
		for (int a = 0; a < G; ++a)
			for (int b = a + 1; b < G; ++b)
				distinct = distinct && h->shards[a].device != h->shards[b].device;
		mode = G <= 1 ? 0 : (distinct ? 1 : 2);
	} else if (mode == 2 && G <= 1) {
		mode = 0;
	}
	if (!h->rank_form)
		for (auto& s : h->shards)
			if (s.comm) {  // (the set of shards or the exchange mode changed)
				ncclCommDestroy(s.comm);
				s.comm = nullptr;
			}
	if (mode == 1 && !h->rank_form) {
		std::vector<int> devs;
		for (int r = 0; r < G; ++r)
			devs.push_back(h->shards[r].device);
		std::vector<ncclComm_t> comms((size_t)G);
		NCCL_TRY(h, ncclCommInitAll(comms.data(), G, devs.data()));
		for (int r = 0; r < G; ++r)
			h->shards[r].comm = comms[(size_t)r];
	}
	h->exchange_used = mode;
	h->comm_ready = true;
	return EXPANN_OK;
}

// ---- transports ----------------------------------------------------------------------------------
// RCCL all-to-all-v on one rank's communicator and stream (inside the caller's ncclGroup): rank sends
// send_bytes[j] from send + send_off[j] to peer j and receives recv_bytes[j] from peer j at recv +
// recv_off[j]; the part it keeps for itself is a device copy.
int rccl_alltoallv(Shard& s, hipStream_t st, const unsigned char* send, const size_t* send_off, const size_t* send_bytes,
                   unsigned char* recv, const size_t* recv_off, const size_t* recv_bytes, int rank, int world) {
	for (int j = 0; j < world; ++j) {
		if (j == rank) {
			if (send_bytes[j])
				HIP_TRY(&s, hipMemcpyAsync(recv + recv_off[j], send + send_off[j], send_bytes[j], hipMemcpyDeviceToDevice, st));
			continue;
		}
		if (send_bytes[j])
			NCCL_TRY(&s, ncclSend(send + send_off[j], send_bytes[j], ncclChar, j, s.comm, st));
		if (recv_bytes[j])
			NCCL_TRY(&s, ncclRecv(recv + recv_off[j], recv_bytes[j], ncclChar, j, s.comm, st));
	}
	return EXPANN_OK;
}

// byte ranges of the slices exchange for rank r of G (ids: 8 B per entry, dists: 4)
struct SliceLayout {
	std::vector<size_t> s_off, s_bytes, r_off, r_bytes;
	SliceLayout(const Slices& sl, size_t k, int r, size_t esz) {
		const size_t G = sl.G;
		s_off.resize(G), s_bytes.resize(G), r_off.resize(G), r_bytes.resize(G);
		for (size_t j = 0; j < G; ++j) {
			s_off[j] = sl.lo(j) * k * esz;          // my results for the queries of slice j
			s_bytes[j] = sl.cnt(j) * k * esz;
			r_off[j] = j * sl.per * k * esz;        // list j of my slice
			r_bytes[j] = sl.cnt((size_t)r) * k * esz;
		}
	}
};

// what a shard of the in-process form does for one search; phases run on the shard's thread
struct Call {
	const void* h_queries = nullptr;        // host API
	const void* const* d_queries = nullptr;  // devices API: [shard] device pointers
	size_t m = 0, k = 0;
	uint64_t* h_ids = nullptr;
	float* h_dists = nullptr;
	uint64_t* const* d_ids = nullptr;        // devices API: [shard] the shard's merged slice
	float* const* d_dists = nullptr;
	bool deferred = false;
};

// phase A: the shard's own scan of all queries -> mine
int phase_scan(expann_sharded* h, int r, const Call& c) {
	Shard& s = h->shards[(size_t)r];
	const int G = h->n_active;
	HIP_TRY(&s, hipSetDevice(s.device));
	int rc = ensure_chunks(s, s.stream, c.m, c.k, G, h->pattern_used());
	if (rc != EXPANN_OK)
		return rc;
	if (h->exchange_used == 2)  // device copies: the readers of the previous search's `mine`
		for (int g = 0; g < G; ++g)
			if (g != r && h->shards[(size_t)g].done_recorded)
				HIP_TRY(&s, hipStreamWaitEvent(s.stream, h->shards[(size_t)g].ev_done, 0));
	const void* dq = nullptr;
	if (c.d_queries) {
		dq = c.d_queries[r];
	} else {
		const size_t qb = c.m * (size_t)h->dim * h->q_elem;
		if (qb > s.q_bytes) {
			HIP_TRY(&s, hipStreamSynchronize(s.stream));
			if (s.d_q) (void)hipFree(s.d_q);
			s.d_q = nullptr;
			s.q_bytes = 0;
			HIP_TRY(&s, hipMalloc(&s.d_q, qb));
			s.q_bytes = qb;
		}
		HIP_TRY(&s, hipMemcpyAsync(s.d_q, c.h_queries, qb, hipMemcpyHostToDevice, s.stream));
		dq = s.d_q;
	}
	SUB_TRY(&s, s, expann_set_option(s.idx, "async_search", c.deferred ? 1 : 0));
	SUB_TRY(&s, s, expann_search_device(s.idx, dq, c.m, c.k, reinterpret_cast<uint64_t*>(s.mine),
	                                    reinterpret_cast<float*>(s.mine + c.m * c.k * 8), s.stream));
	if (h->exchange_used == 2)
		HIP_TRY(&s, hipEventRecord(s.ev_search, s.stream));
	return EXPANN_OK;
}

// phase B: exchange, merge of this shard's query slice, hand the slice over
int phase_exchange(expann_sharded* h, int r, const Call& c) {
	Shard& s = h->shards[(size_t)r];
	const int G = h->n_active;
	const size_t m = c.m, k = c.k, cb = chunk_bytes(m, k);
	const Slices sl(m, (size_t)G);
	const size_t q0 = sl.lo((size_t)r), cnt = sl.cnt((size_t)r);
	HIP_TRY(&s, hipSetDevice(s.device));
	uint64_t* out_ids = c.d_ids ? c.d_ids[r] : reinterpret_cast<uint64_t*>(s.merged);
	float* out_d = c.d_ids ? c.d_dists[r] : reinterpret_cast<float*>(s.merged + cnt * k * 8);
	if (G == 1 && h->exchange_used != 1) {  // one shard: its result is the result
		if (c.d_ids) {
			HIP_TRY(&s, hipMemcpyAsync(out_ids, s.mine, sizeof(uint64_t) * m * k, hipMemcpyDeviceToDevice, s.stream));
			HIP_TRY(&s, hipMemcpyAsync(out_d, s.mine + m * k * 8, sizeof(float) * m * k, hipMemcpyDeviceToDevice, s.stream));
		} else {
			HIP_TRY(&s, hipMemcpyAsync(c.h_ids, s.mine, sizeof(uint64_t) * m * k, hipMemcpyDeviceToHost, s.stream));
			if (c.h_dists)
				HIP_TRY(&s, hipMemcpyAsync(c.h_dists, s.mine + m * k * 8, sizeof(float) * m * k, hipMemcpyDeviceToHost, s.stream));
		}
		return EXPANN_OK;
	}
	const uint64_t* in_ids = nullptr;
	const float* in_d = nullptr;
	size_t ids_stride = 0, d_stride = 0;
	if (h->pattern_used() == 1) {
		// ---- all-gather of whole chunks -----------------------------------------------------------
		if (h->exchange_used == 1) {
			NCCL_TRY(&s, ncclAllGather(s.mine, s.gathered, cb, ncclChar, s.comm, s.stream));
		} else {
			for (int g = 0; g < G; ++g) {
				Shard& src = h->shards[(size_t)g];
				if (g != r)
					HIP_TRY(&s, hipStreamWaitEvent(s.stream, src.ev_search, 0));
				if (src.device == s.device)
					HIP_TRY(&s, hipMemcpyAsync(s.gathered + (size_t)g * cb, src.mine, cb, hipMemcpyDeviceToDevice, s.stream));
				else
					HIP_TRY(&s, hipMemcpyPeerAsync(s.gathered + (size_t)g * cb, s.device, src.mine, src.device, cb, s.stream));
			}
			HIP_TRY(&s, hipEventRecord(s.ev_done, s.stream));
			s.done_recorded = true;
		}
		in_ids = reinterpret_cast<const uint64_t*>(s.gathered) + q0 * k;
		in_d = reinterpret_cast<const float*>(s.gathered + m * k * 8) + q0 * k;
		ids_stride = cb / 8;
		d_stride = cb / 4;
	} else {
		// ---- all-to-all of query slices: list j of my slice comes from shard j ---------------------
		unsigned char* g_ids = s.gathered;
		unsigned char* g_d = s.gathered + (size_t)G * sl.per * k * 8;
		if (h->exchange_used == 1) {
			const SliceLayout li(sl, k, r, 8), ld(sl, k, r, 4);
			NCCL_TRY(&s, ncclGroupStart());
			int rc = rccl_alltoallv(s, s.stream, s.mine, li.s_off.data(), li.s_bytes.data(), g_ids, li.r_off.data(),
			                        li.r_bytes.data(), r, G);
			if (rc == EXPANN_OK)
				rc = rccl_alltoallv(s, s.stream, s.mine + m * k * 8, ld.s_off.data(), ld.s_bytes.data(), g_d,
				                    ld.r_off.data(), ld.r_bytes.data(), r, G);
			const ncclResult_t ge = ncclGroupEnd();
			if (rc != EXPANN_OK)
				return rc;
			NCCL_TRY(&s, ge);
		} else {
			for (int g = 0; g < G && cnt; ++g) {
				Shard& src = h->shards[(size_t)g];
				if (g != r)
					HIP_TRY(&s, hipStreamWaitEvent(s.stream, src.ev_search, 0));
				const unsigned char* s_ids = src.mine + q0 * k * 8;
				const unsigned char* s_d = src.mine + m * k * 8 + q0 * k * 4;
				unsigned char* d_i = g_ids + (size_t)g * sl.per * k * 8;
				unsigned char* d_d = g_d + (size_t)g * sl.per * k * 4;
				if (src.device == s.device) {
					HIP_TRY(&s, hipMemcpyAsync(d_i, s_ids, cnt * k * 8, hipMemcpyDeviceToDevice, s.stream));
					HIP_TRY(&s, hipMemcpyAsync(d_d, s_d, cnt * k * 4, hipMemcpyDeviceToDevice, s.stream));
				} else {
					HIP_TRY(&s, hipMemcpyPeerAsync(d_i, s.device, s_ids, src.device, cnt * k * 8, s.stream));
					HIP_TRY(&s, hipMemcpyPeerAsync(d_d, s.device, s_d, src.device, cnt * k * 4, s.stream));
				}
			}
			HIP_TRY(&s, hipEventRecord(s.ev_done, s.stream));
			s.done_recorded = true;
		}
		in_ids = reinterpret_cast<const uint64_t*>(g_ids);
		in_d = reinterpret_cast<const float*>(g_d);
		ids_stride = d_stride = sl.per * k;
	}
	if (cnt == 0)
		return EXPANN_OK;
	if (expann_merge_topk_strided_device(s.device, in_ids, in_d, ids_stride, d_stride, (size_t)G, cnt, k, out_ids, out_d,
	                                     s.stream) != EXPANN_OK)
		return s.fail(EXPANN_ERR_HIP, std::string("merge: ") + expann_last_error(nullptr));
	if (!c.d_ids) {
		HIP_TRY(&s, hipMemcpyAsync(c.h_ids + q0 * k, out_ids, sizeof(uint64_t) * cnt * k, hipMemcpyDeviceToHost, s.stream));
		if (c.h_dists)
			HIP_TRY(&s, hipMemcpyAsync(c.h_dists + q0 * k, out_d, sizeof(float) * cnt * k, hipMemcpyDeviceToHost, s.stream));
	}
	return EXPANN_OK;
}

// phase C: wait for this shard's stream; EXPANN_ERR_OVERFLOW when its deferred search needs the retry
int phase_wait(expann_sharded* h, int r) {
	Shard& s = h->shards[(size_t)r];
	HIP_TRY(&s, hipSetDevice(s.device));
	const int rs = expann_sync(s.idx);
	HIP_TRY(&s, hipStreamSynchronize(s.stream));
	if (rs != EXPANN_OK)
		return s.fail(rs, std::string("shard on device ") + std::to_string(s.device) + ": " + expann_last_error(s.idx));
	return EXPANN_OK;
}

// run fn(r) for every shard in use: on the shards' threads (all at once) or, with the option
// "threads" = 0, one after the other on the caller's thread.  The first failure is the handle's.
int for_shards(expann_sharded* h, const std::function<int(int)>& fn) {
	const int G = h->n_active;
	int bad = EXPANN_OK, bad_r = -1;
	// (RCCL transport: always the shards' own threads.  One thread that calls a collective rank by rank
	// without a group around all of them can block in the first rank's call until its peers join; a group around
	// phase_exchange would defer the sends past the merge kernels enqueued inside it.)
	const bool threaded = (h->opt_threads != 0 || h->exchange_used == 1) && G > 1;
	for (int r = 0; r < G; ++r) {
		Shard& s = h->shards[(size_t)r];
		if (threaded) {
			if (!s.worker)
				s.worker.reset(new Worker());
			s.worker->post([&fn, r] { return fn(r); });
		} else {
			const int rc = fn(r);
			if (rc != EXPANN_OK && bad == EXPANN_OK)
				bad = rc, bad_r = r;
		}
	}
	if (threaded)
		for (int r = 0; r < G; ++r) {
			const int rc = h->shards[(size_t)r].worker->wait();
			if (rc != EXPANN_OK && bad == EXPANN_OK)
				bad = rc, bad_r = r;
		}
	if (bad != EXPANN_OK)
		return h->fail(bad, h->shards[(size_t)bad_r].err);
	return EXPANN_OK;
}

// one search of the in-process form: scan on every shard, exchange + merge, then (unless the caller
// syncs later) the wait.  Between the phases the threads meet: the device-copy transport records the
// events of phase A before any stream of phase B waits for them; RCCL needs no such order, but the
// join costs microseconds.
int search_inprocess(expann_sharded* h, const Call& c, bool wait) {
	const auto t0 = std::chrono::steady_clock::now();
	int rc = for_shards(h, [&](int r) { return phase_scan(h, r, c); });
	if (rc == EXPANN_OK)
		rc = for_shards(h, [&](int r) { return phase_exchange(h, r, c); });
	h->last_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (rc != EXPANN_OK || !wait) {
		if (rc != EXPANN_OK)  // (leave nothing in flight behind a failure)
			(void)for_shards(h, [&](int r) { (void)phase_wait(h, r); return (int)EXPANN_OK; });
		return rc;
	}
	return for_shards(h, [&](int r) { return phase_wait(h, r); });
}

}  // namespace

extern "C" {

int expann_sharded_create(int dim, int dtype, int metric, const int* devices, int n_dev, expann_sharded** out) {
	expann_sharded* h = nullptr;
	int rc = common_create(dim, dtype, metric, out, h);
	if (rc != EXPANN_OK)
		return rc;
	if (!devices || n_dev < 1 || n_dev > 64) {
		g_create_error = "expann_sharded_create: devices == NULL or n_dev outside [1, 64]";
		delete h;
		return EXPANN_ERR_INVALID_ARG;
	}
	h->shards.resize((size_t)n_dev);
	for (int r = 0; r < n_dev; ++r) {
		rc = open_shard(h, h->shards[(size_t)r], devices[r]);
		if (rc != EXPANN_OK) {
			expann_sharded_destroy(h);
			return rc;
		}
	}
	h->world = n_dev;
	*out = h;
	return EXPANN_OK;
}

int expann_sharded_unique_id(void* id128) {
	if (!id128)
		return EXPANN_ERR_INVALID_ARG;
	static_assert(sizeof(ncclUniqueId) == 128, "expann_sharded_unique_id hands out 128 bytes");
	ncclUniqueId id;
	const ncclResult_t r = ncclGetUniqueId(&id);
	if (r != ncclSuccess) {
		g_create_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r);
		return EXPANN_ERR_HIP;
	}
	std::memcpy(id128, &id, sizeof(id));
	return EXPANN_OK;
}

int expann_sharded_create_rank(int dim, int dtype, int metric, int device, int rank, int world, const void* id128,
                               expann_sharded** out) {
	expann_sharded* h = nullptr;
	int rc = common_create(dim, dtype, metric, out, h);
	if (rc != EXPANN_OK)
		return rc;
	if (world < 1 || world > 64 || rank < 0 || rank >= world) {
		g_create_error = "expann_sharded_create_rank: bad rank / world (1 <= world <= 64)";
		delete h;
		return EXPANN_ERR_INVALID_ARG;
	}
	h->rank_form = true;
	h->rank = rank;
	h->world = world;
	h->n_active = 1;
	h->shards.resize(1);
	rc = open_shard(h, h->shards[0], device);
	if (rc != EXPANN_OK) {
		expann_sharded_destroy(h);
		return rc;
	}
	// no id: no RCCL communicator -- one rank needs none, more ranks exchange through the caller's
	// function (expann_sharded_set_exchange_fn / _set_alltoallv_fn); a one-rank id gives the RCCL
	// path on one GPU
	if (id128) {
		ncclUniqueId id;
		std::memcpy(&id, id128, sizeof(id));
		const ncclResult_t r = ncclCommInitRank(&h->shards[0].comm, world, id, rank);
		if (r != ncclSuccess) {
			g_create_error = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
			expann_sharded_destroy(h);
			return EXPANN_ERR_HIP;
		}
		h->exchange_used = 1;
	}
	h->comm_ready = true;
	*out = h;
	return EXPANN_OK;
}

void expann_sharded_destroy(expann_sharded* h) {
	if (!h)
		return;
	for (auto& s : h->shards)
		close_shard(s);
	delete h;
}

const char* expann_sharded_last_error(const expann_sharded* h) {
	return h ? h->err.c_str() : g_create_error.c_str();
}

int expann_sharded_add(expann_sharded* h, const void* rows, size_t n) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "rank form: pass this rank's rows with expann_sharded_set_shard_device");
	if (!rows && n)
		return h->fail(EXPANN_ERR_INVALID_ARG, "rows == NULL");
	if (h->n_total)
		return h->fail(EXPANN_ERR_INVALID_ARG, "index already built");
	const unsigned char* src = static_cast<const unsigned char*>(rows);
	h->staging.insert(h->staging.end(), src, src + n * (size_t)h->dim * h->elem);
	h->n_staged += n;
	return EXPANN_OK;
}

int expann_sharded_build(expann_sharded* h) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (h->n_total)
		return h->fail(EXPANN_ERR_INVALID_ARG, "index already built");
	if (h->n_staged == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "build() on an empty index");
	// shard r = rows [r * per, min(N, (r+1) * per)), per = ceil(N / G)
	const size_t G = h->shards.size(), N = h->n_staged;
	const size_t per = (N + G - 1) / G;
	const size_t rowb = (size_t)h->dim * h->elem;
	h->n_active = 0;
	for (size_t r = 0; r < G && r * per < N; ++r) {
		Shard& s = h->shards[r];
		const size_t lo = r * per, hi = std::min(N, lo + per);
		HIP_TRY(h, hipSetDevice(s.device));
		HIP_TRY(h, hipMalloc(&s.d_rows, (hi - lo) * rowb));
		HIP_TRY(h, hipMemcpyAsync(s.d_rows, h->staging.data() + lo * rowb, (hi - lo) * rowb, hipMemcpyHostToDevice,
		                          s.stream));
		s.n = hi - lo;
		s.id_offset = lo;
		SUB_TRY(h, s, expann_set_base_device(s.idx, s.d_rows, s.n, s.id_offset));
		h->n_active++;
	}
	for (int r = 0; r < h->n_active; ++r) {
		HIP_TRY(h, hipSetDevice(h->shards[(size_t)r].device));
		HIP_TRY(h, hipStreamSynchronize(h->shards[(size_t)r].stream));
	}
	h->n_total = N;
	std::vector<unsigned char>().swap(h->staging);
	return ensure_comm(h);
}

int expann_sharded_set_shard_device(expann_sharded* h, int shard, const void* d_rows, size_t n, uint64_t id_offset) {
	if (!h || shard < 0 || (size_t)shard >= h->shards.size())
		return h ? h->fail(EXPANN_ERR_INVALID_ARG, "shard index out of range") : EXPANN_ERR_INVALID_ARG;
	Shard& s = h->shards[(size_t)shard];
	if (s.d_rows)
		return h->fail(EXPANN_ERR_INVALID_ARG, "shard already holds rows uploaded by build()");
	if (!h->rank_form && shard != h->n_active && s.n == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "in-process form: adopt the shards in order 0, 1, ...");
	if (n == 0) {
		// the ceil partition leaves trailing ranks without rows when N < G * (G - 1): such a rank
		// still takes part in every exchange, with an all-padding chunk
		if (!h->rank_form)
			return h->fail(EXPANN_ERR_INVALID_ARG, "in-process form: leave out the shards without rows");
		h->n_total -= s.n;
		s.n = 0;
		s.id_offset = id_offset;
		s.adopted_empty = true;
		return EXPANN_OK;
	}
	SUB_TRY(h, s, expann_set_base_device(s.idx, d_rows, n, id_offset));
	if (s.n == 0 && !h->rank_form)
		h->n_active++;
	h->n_total += n - s.n;
	s.n = n;
	s.id_offset = id_offset;
	s.adopted_empty = false;
	h->comm_ready = h->rank_form;  // (in-process: the set of shards in use may have grown)
	return EXPANN_OK;
}

size_t expann_sharded_size(const expann_sharded* h) { return h ? (h->n_total ? h->n_total : h->n_staged) : 0; }
int expann_sharded_shards(const expann_sharded* h) { return h ? (h->rank_form ? h->world : h->n_active) : 0; }
int expann_sharded_exchange(const expann_sharded* h) { return h ? h->exchange_used : 0; }
int expann_sharded_exchange_pattern(const expann_sharded* h) {
	if (!h || (h->rank_form ? h->world : h->n_active) <= 1)
		return 0;
	if (h->rank_form && !h->shards[0].comm)  // the caller's transport decides: the function that is set
		return (h->alltoallv_fn && h->pattern != 1) ? 2 : (h->exchange_fn ? 1 : (h->alltoallv_fn ? 2 : 0));
	return h->pattern_used();
}
int expann_sharded_comm_ranks(const expann_sharded* h) {
	if (!h || h->shards.empty() || !h->shards[0].comm)
		return 0;
	int n = 0;
	return ncclCommCount(h->shards[0].comm, &n) == ncclSuccess ? n : -1;
}
double expann_sharded_last_enqueue_ms(const expann_sharded* h) { return h ? h->last_enqueue_ms : 0.0; }

int expann_sharded_search(expann_sharded* h, const void* queries, size_t m, size_t k, uint64_t* ids, float* dists) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "rank form: use expann_sharded_search_device");
	if (h->n_active == 0)
		return h->fail(EXPANN_ERR_NOT_BUILT, "search before build()");
	if (k == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "k == 0");
	if (m == 0)
		return EXPANN_OK;
	if (!queries || !ids)
		return h->fail(EXPANN_ERR_INVALID_ARG, "NULL query/ids pointer");
	int rc = ensure_comm(h);
	if (rc != EXPANN_OK)
		return rc;
	if (h->devices_pending) {  // deferred device searches first
		rc = expann_sharded_sync(h);
		if (rc != EXPANN_OK)
			return rc;
	}
	h->searches++;
	Call c;
	c.h_queries = queries;
	c.m = m;
	c.k = k;
	c.h_ids = ids;
	c.h_dists = dists;
	c.deferred = h->opt_async != 0;
	rc = search_inprocess(h, c, true);
	if (rc == EXPANN_ERR_OVERFLOW && c.deferred) {
		// a deferred search needed the synchronous retry (overflowed lists, queries outside the
		// filter's range): once more, every shard waiting for its own search
		h->retries++;
		c.deferred = false;
		rc = search_inprocess(h, c, true);
	}
	return rc;
}

int expann_sharded_search_devices(expann_sharded* h, const void* const* d_queries, size_t m, size_t k,
                                  uint64_t* const* d_ids, float* const* d_dists) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "rank form: use expann_sharded_search_device");
	if (h->n_active == 0)
		return h->fail(EXPANN_ERR_NOT_BUILT, "search before build()");
	if (k == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "k == 0");
	if (m == 0)
		return EXPANN_OK;
	if (!d_queries || !d_ids || !d_dists)
		return h->fail(EXPANN_ERR_INVALID_ARG, "NULL pointer array");
	const Slices sl(m, (size_t)h->n_active);
	for (int r = 0; r < h->n_active; ++r)
		if (!d_queries[r] || (sl.cnt((size_t)r) && (!d_ids[r] || !d_dists[r])))
			return h->fail(EXPANN_ERR_INVALID_ARG, "NULL query / ids / dists pointer of shard " + std::to_string(r));
	int rc = ensure_comm(h);
	if (rc != EXPANN_OK)
		return rc;
	h->searches++;
	Call c;
	c.d_queries = d_queries;
	c.m = m;
	c.k = k;
	c.d_ids = d_ids;
	c.d_dists = d_dists;
	c.deferred = h->opt_async != 0;
	rc = search_inprocess(h, c, !c.deferred);
	if (rc == EXPANN_OK && c.deferred)
		h->devices_pending = true;
	return rc;
}

int expann_sharded_slice(const expann_sharded* h, size_t m, int shard, size_t* q_lo, size_t* q_hi) {
	if (!h || shard < 0)
		return EXPANN_ERR_INVALID_ARG;
	const int G = h->rank_form ? h->world : h->n_active;
	if (G < 1 || shard >= G)
		return h->fail(EXPANN_ERR_INVALID_ARG, "shard index out of range");
	const Slices sl(m, (size_t)G);
	if (q_lo) *q_lo = sl.lo((size_t)shard);
	if (q_hi) *q_hi = sl.hi((size_t)shard);
	return EXPANN_OK;
}

int expann_sharded_search_device(expann_sharded* h, const void* d_queries, size_t m, size_t k, uint64_t* d_ids,
                                 float* d_dists, void* stream) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "in-process form: use expann_sharded_search / _search_devices");
	Shard& s = h->shards[0];
	if (s.n == 0 && !s.adopted_empty)
		return h->fail(EXPANN_ERR_NOT_BUILT, "search before expann_sharded_set_shard_device");
	if (k == 0)
		return h->fail(EXPANN_ERR_INVALID_ARG, "k == 0");
	if (m == 0)
		return EXPANN_OK;
	if (!d_queries || !d_ids || !d_dists)
		return h->fail(EXPANN_ERR_INVALID_ARG, "NULL query / ids / dists pointer");
	HIP_TRY(h, hipSetDevice(s.device));
	hipStream_t st = stream ? (hipStream_t)stream : s.stream;
	h->searches++;
	const int G = h->world, r = h->rank;
	const bool caller = !s.comm;
	if (caller && !h->exchange_fn && !h->alltoallv_fn) {  // one rank, no communicator: the local result is the result
		if (G > 1)
			return h->fail(EXPANN_ERR_INVALID_ARG,
			               "rank form without a unique id: set the exchange with expann_sharded_set_exchange_fn "
			               "or expann_sharded_set_alltoallv_fn");
		SUB_TRY(h, s, expann_search_device(s.idx, d_queries, m, k, d_ids, d_dists, st));
		return EXPANN_OK;
	}
	// which pattern: the option, limited by what the caller's transport offers
	int pattern = h->pattern_used();
	if (caller && pattern == 2 && !h->alltoallv_fn)
		pattern = 1;
	if (caller && pattern == 1 && !h->exchange_fn)
		pattern = 2;
	const size_t cb = chunk_bytes(m, k);
	int rc = ensure_chunks(s, st, m, k, G, pattern);
	if (rc != EXPANN_OK)
		return h->fail(rc, s.err);
	uint64_t* mine_ids = reinterpret_cast<uint64_t*>(s.mine);
	float* mine_d = reinterpret_cast<float*>(s.mine + m * k * 8);
	if (s.n) {
		SUB_TRY(h, s, expann_search_device(s.idx, d_queries, m, k, mine_ids, mine_d, st));
	} else {  // no rows on this rank: (UINT64_MAX, +inf) everywhere
		HIP_TRY(h, hipMemsetAsync(mine_ids, 0xFF, m * k * 8, st));
		HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)mine_d, 0x7F800000, m * k, st));
	}
	if (pattern == 1) {
		if (h->exchange_fn) {
			if (h->exchange_fn(h->exchange_ctx, s.mine, s.gathered, cb, r, G, (void*)st) != 0)
				return h->fail(EXPANN_ERR_HIP, "the caller's exchange function failed");
		} else {
			NCCL_TRY(h, ncclAllGather(s.mine, s.gathered, cb, ncclChar, s.comm, st));
		}
		if (expann_merge_topk_strided_device(s.device, reinterpret_cast<const uint64_t*>(s.gathered),
		                                     reinterpret_cast<const float*>(s.gathered + m * k * 8), cb / 8, cb / 4,
		                                     (size_t)G, m, k, d_ids, d_dists, st) != EXPANN_OK)
			return h->fail(EXPANN_ERR_HIP, std::string("merge: ") + expann_last_error(nullptr));
		return EXPANN_OK;
	}
	// ---- slices: all-to-all, merge of my slice, all-gather-v of the merged slices ----------------------
	const Slices sl(m, (size_t)G);
	const size_t q0 = sl.lo((size_t)r), cnt = sl.cnt((size_t)r);
	unsigned char* g_ids = s.gathered;
	unsigned char* g_d = s.gathered + (size_t)G * sl.per * k * 8;
	unsigned char* mg_ids = s.merged;
	unsigned char* mg_d = s.merged + sl.per * k * 8;
	const SliceLayout li(sl, k, r, 8), ld(sl, k, r, 4);
	// the second step: my merged slice to everyone, everyone's slice straight into d_ids / d_dists
	std::vector<size_t> zero((size_t)G, 0), gs_i((size_t)G), gs_d((size_t)G), gr_oi((size_t)G), gr_od((size_t)G),
	    gr_i((size_t)G), gr_d((size_t)G);
	for (int j = 0; j < G; ++j) {
		gs_i[(size_t)j] = cnt * k * 8;
		gs_d[(size_t)j] = cnt * k * 4;
		gr_oi[(size_t)j] = sl.lo((size_t)j) * k * 8;
		gr_od[(size_t)j] = sl.lo((size_t)j) * k * 4;
		gr_i[(size_t)j] = sl.cnt((size_t)j) * k * 8;
		gr_d[(size_t)j] = sl.cnt((size_t)j) * k * 4;
	}
	auto a2a = [&](const unsigned char* send, const std::vector<size_t>& so, const std::vector<size_t>& sb,
	               unsigned char* recv, const std::vector<size_t>& ro, const std::vector<size_t>& rb) -> int {
		if (!caller)
			return rccl_alltoallv(s, st, send, so.data(), sb.data(), recv, ro.data(), rb.data(), r, G);
		// the caller's transport moves what crosses ranks; the part that stays is a device copy
		std::vector<size_t> sb2 = sb, rb2 = rb;
		if (sb[(size_t)r])
			HIP_TRY(&s, hipMemcpyAsync(recv + ro[(size_t)r], send + so[(size_t)r], sb[(size_t)r], hipMemcpyDeviceToDevice, st));
		sb2[(size_t)r] = rb2[(size_t)r] = 0;
		if (h->alltoallv_fn(h->alltoallv_ctx, send, so.data(), sb2.data(), recv, ro.data(), rb2.data(), r, G, (void*)st) != 0)
			return s.fail(EXPANN_ERR_HIP, "the caller's all-to-all-v function failed");
		return EXPANN_OK;
	};
	auto grouped = [&](const std::function<int()>& body) -> int {
		if (caller)
			return body();
		NCCL_TRY(&s, ncclGroupStart());
		const int rb = body();
		const ncclResult_t ge = ncclGroupEnd();
		if (rb != EXPANN_OK)
			return rb;
		NCCL_TRY(&s, ge);
		return EXPANN_OK;
	};
	rc = grouped([&]() -> int {
		int x = a2a(s.mine, li.s_off, li.s_bytes, g_ids, li.r_off, li.r_bytes);
		return x != EXPANN_OK ? x : a2a(s.mine + m * k * 8, ld.s_off, ld.s_bytes, g_d, ld.r_off, ld.r_bytes);
	});
	if (rc != EXPANN_OK)
		return h->fail(rc, s.err);
	if (cnt && expann_merge_topk_strided_device(s.device, reinterpret_cast<const uint64_t*>(g_ids),
	                                            reinterpret_cast<const float*>(g_d), sl.per * k, sl.per * k, (size_t)G, cnt, k,
	                                            reinterpret_cast<uint64_t*>(mg_ids), reinterpret_cast<float*>(mg_d),
	                                            st) != EXPANN_OK)
		return h->fail(EXPANN_ERR_HIP, std::string("merge: ") + expann_last_error(nullptr));
	(void)q0;
	rc = grouped([&]() -> int {
		int x = a2a(mg_ids, zero, gs_i, reinterpret_cast<unsigned char*>(d_ids), gr_oi, gr_i);
		return x != EXPANN_OK ? x : a2a(mg_d, zero, gs_d, reinterpret_cast<unsigned char*>(d_dists), gr_od, gr_d);
	});
	if (rc != EXPANN_OK)
		return h->fail(rc, s.err);
	return EXPANN_OK;
}

int expann_sharded_set_exchange_fn(expann_sharded* h, expann_exchange_fn fn, void* ctx) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "the in-process form exchanges between its own devices");
	if (h->shards[0].comm && fn)
		return h->fail(EXPANN_ERR_INVALID_ARG, "this rank exchanges over its RCCL communicator (created with a unique id)");
	h->exchange_fn = fn;
	h->exchange_ctx = ctx;
	h->exchange_used = (fn || h->alltoallv_fn) ? 3 : (h->shards[0].comm ? 1 : 0);
	return EXPANN_OK;
}

int expann_sharded_set_alltoallv_fn(expann_sharded* h, expann_alltoallv_fn fn, void* ctx) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "the in-process form exchanges between its own devices");
	if (h->shards[0].comm && fn)
		return h->fail(EXPANN_ERR_INVALID_ARG, "this rank exchanges over its RCCL communicator (created with a unique id)");
	h->alltoallv_fn = fn;
	h->alltoallv_ctx = ctx;
	h->exchange_used = (fn || h->exchange_fn) ? 3 : (h->shards[0].comm ? 1 : 0);
	return EXPANN_OK;
}

int expann_sharded_sync(expann_sharded* h) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	int bad = EXPANN_OK;
	for (auto& s : h->shards)
		if (s.idx && (s.n || s.adopted_empty)) {
			HIP_TRY(h, hipSetDevice(s.device));
			const int rc = expann_sync(s.idx);
			if (!h->rank_form)
				HIP_TRY(h, hipStreamSynchronize(s.stream));
			if (rc != EXPANN_OK)
				bad = h->fail(rc, std::string("shard on device ") + std::to_string(s.device) + ": " +
				                      expann_last_error(s.idx));
		}
	h->devices_pending = false;
	return bad;
}

int expann_sharded_set_option(expann_sharded* h, const char* name, long value) {
	if (!h || !name)
		return EXPANN_ERR_INVALID_ARG;
	if (!std::strcmp(name, "exchange")) {
		if (value < 0 || value > 2)
			return h->fail(EXPANN_ERR_INVALID_ARG, "exchange: 0 auto, 1 RCCL, 2 device copies");
		if (h->rank_form)
			return h->fail(EXPANN_ERR_INVALID_ARG, "rank form: the transport is RCCL (unique id) or the caller's function");
		h->exchange = (int)value;
		h->comm_ready = false;
		return EXPANN_OK;
	}
	if (!std::strcmp(name, "exchange_pattern")) {
		if (value < 0 || value > 2)
			return h->fail(EXPANN_ERR_INVALID_ARG, "exchange_pattern: 0 auto, 1 all-gather of whole chunks, 2 all-to-all of query slices");
		h->pattern = (int)value;
		return EXPANN_OK;
	}
	if (!std::strcmp(name, "threads") && !h->rank_form) {
		h->opt_threads = value;
		return EXPANN_OK;
	}
	if (!std::strcmp(name, "async_search") && !h->rank_form) {
		h->opt_async = value;  // (in-process form: the shards' mode is set per search)
		return EXPANN_OK;
	}
	for (auto& s : h->shards)
		if (s.idx)
			SUB_TRY(h, s, expann_set_option(s.idx, name, value));
	return EXPANN_OK;
}

int expann_sharded_set_profiling(expann_sharded* h, int enable) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	for (auto& s : h->shards)
		if (s.idx)
			SUB_TRY(h, s, expann_set_profiling(s.idx, enable));
	return EXPANN_OK;
}

int expann_sharded_get_profile(expann_sharded* h, int shard, expann_profile* out) {
	if (!h || !out || shard < 0 || (size_t)shard >= h->shards.size() || !h->shards[(size_t)shard].idx)
		return EXPANN_ERR_INVALID_ARG;
	Shard& s = h->shards[(size_t)shard];
	SUB_TRY(h, s, expann_get_profile(s.idx, out));
	return EXPANN_OK;
}

}  // extern "C"
