// expann_sharded.hip -- the row-sharded brute-force engine behind the C ABI (expann_sharded_*,
// include/expann_hip.h): SURVEY 8b's `create(..., const int* devices, n_dev, ...)` and 8e's
// partition, exchange and merge.
//
//   rows        shard r holds the contiguous range [r * ceil(N/G), min(N, (r+1) * ceil(N/G)));
//               ids are global row numbers, so the per-shard (score, id) order is the global one
//   scan        every shard searches ALL queries on its own device / stream (an expann_index per
//               shard, deferred check: the host enqueues all devices before it waits for any)
//   exchange    ONE ncclAllGather (RCCL over xGMI) of the fixed-size chunk [ids m*k u64 | dists m*k
//               f32] per shard -- 8 B... 12 B per entry; 12 MB per rank at m = 10 k, k = 100
//   merge       merge_topk (select.hpp) over the G gathered lists, (score, id) order
//
// Two forms, same code path after the communicator exists:
//   * in-process (expann_sharded_create): one handle drives n_dev devices of this node, one
//     stream per device, ncclCommInitAll; device r merges the r-th slice of the queries and
//     copies it to the caller's host buffers (merge and D2H run on all devices in parallel);
//   * one rank of a one-process-per-GPU job (expann_sharded_create_rank): ncclCommInitRank with
//     a unique id the launcher distributed; every rank ends with the full [m][k] result in
//     device memory, on the caller's stream.
// Without RCCL the exchange can run as plain device copies (option "exchange" = 2; automatic
// when several shards share one device, which RCCL refuses) -- the form the one-GPU test box
// exercises with 8 shards.
//
// Everything below the exchange is the public C ABI of the single-device index (expann_create,
// expann_set_base_device, expann_search_device, expann_sync, expann_merge_topk_strided_device):
// this file holds no kernels.
#include "../../include/expann_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "host_common.hpp"

using namespace expann;

namespace {

struct Shard {
	int device = 0;
	expann_index* idx = nullptr;
	hipStream_t stream = nullptr;
	void* d_rows = nullptr;     // owned copy of this shard's rows (build()) or nullptr (adopted)
	size_t n = 0;
	uint64_t id_offset = 0;
	void* d_q = nullptr;        // queries on this device (in-process form)
	size_t q_bytes = 0;
	unsigned char* mine = nullptr;      // [ids | dists] of this shard's search
	unsigned char* gathered = nullptr;  // G chunks
	unsigned char* merged = nullptr;    // merged slice (in-process form)
	size_t chunk_alloc = 0;
	int chunk_ranks = 0;                // `gathered` holds this many chunks of chunk_alloc bytes
	hipEvent_t ev_search = nullptr, ev_done = nullptr;
	ncclComm_t comm = nullptr;
};

size_t chunk_bytes(size_t m, size_t k) { return (m * k * 12 + 15) / 16 * 16; }

}  // namespace

struct expann_sharded {
	int dim = 0, dtype = 0, metric = 0;
	size_t elem = 4, q_elem = 4;
	std::vector<Shard> shards;          // in-process: one per device; rank form: exactly one
	int rank = 0, world = 1;            // rank form (in-process: world = shards in use)
	expann_exchange_fn exchange_fn = nullptr;  // rank form: the caller's all-gather in place of RCCL's
	void* exchange_ctx = nullptr;
	bool rank_form = false;
	std::vector<unsigned char> staging;  // add() rows until build()
	size_t n_staged = 0, n_total = 0;
	int n_active = 0;                   // shards that hold rows (in-process)
	int exchange = 0;                   // option: 0 auto, 1 RCCL, 2 device copies
	int exchange_used = 0;              // 1 RCCL, 2 copies, 0 none (single shard)
	bool comm_ready = false;
	long opt_async = 1;
	uint64_t searches = 0, retries = 0;
	mutable std::string err;
	int fail(int code, const std::string& msg) const {
		err = msg;
		return code;
	}
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

// exchange buffers of one shard for chunks of cb bytes from G ranks
int ensure_chunks(expann_sharded* h, Shard& s, size_t cb, int G) {
	if (cb <= s.chunk_alloc && G <= s.chunk_ranks)
		return EXPANN_OK;
	cb = std::max(cb, s.chunk_alloc);
	HIP_TRY(h, hipSetDevice(s.device));
	HIP_TRY(h, hipStreamSynchronize(s.stream));
	if (s.mine) (void)hipFree(s.mine);
	if (s.gathered) (void)hipFree(s.gathered);
	if (s.merged) (void)hipFree(s.merged);
	s.mine = s.gathered = s.merged = nullptr;
	s.chunk_alloc = 0;
	HIP_TRY(h, hipMalloc(&s.mine, cb));
	HIP_TRY(h, hipMalloc(&s.gathered, cb * (size_t)G));
	HIP_TRY(h, hipMalloc(&s.merged, cb));
	s.chunk_alloc = cb;
	s.chunk_ranks = G;
	return EXPANN_OK;
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

// After every shard's search has been enqueued on its stream: all-gather the chunks.  On return
// shard r's `gathered` holds the G chunks back to back (in its stream's order).
int exchange_chunks(expann_sharded* h, size_t cb) {
	const int G = h->n_active;
	if (h->exchange_used == 1) {
		NCCL_TRY(h, ncclGroupStart());
		for (int r = 0; r < G; ++r) {
			Shard& s = h->shards[r];
			ncclResult_t e = ncclAllGather(s.mine, s.gathered, cb, ncclChar, s.comm, s.stream);
			if (e != ncclSuccess) {
				ncclGroupEnd();
				return h->fail(EXPANN_ERR_HIP, std::string("ncclAllGather: ") + ncclGetErrorString(e));
			}
		}
		NCCL_TRY(h, ncclGroupEnd());
		return EXPANN_OK;
	}
	// device copies: every destination stream waits for every source's search, then pulls
	for (int g = 0; g < G; ++g) {
		HIP_TRY(h, hipSetDevice(h->shards[g].device));
		HIP_TRY(h, hipEventRecord(h->shards[g].ev_search, h->shards[g].stream));
	}
	for (int r = 0; r < G; ++r) {
		Shard& d = h->shards[r];
		HIP_TRY(h, hipSetDevice(d.device));
		for (int g = 0; g < G; ++g) {
			Shard& s = h->shards[g];
			if (g != r)
				HIP_TRY(h, hipStreamWaitEvent(d.stream, s.ev_search, 0));
			if (s.device == d.device)
				HIP_TRY(h, hipMemcpyAsync(d.gathered + (size_t)g * cb, s.mine, cb, hipMemcpyDeviceToDevice, d.stream));
			else
				HIP_TRY(h, hipMemcpyPeerAsync(d.gathered + (size_t)g * cb, d.device, s.mine, s.device, cb, d.stream));
		}
		HIP_TRY(h, hipEventRecord(d.ev_done, d.stream));
	}
	// a shard's `mine` may be overwritten by its next search only after every reader is done
	for (int g = 0; g < G; ++g) {
		HIP_TRY(h, hipSetDevice(h->shards[g].device));
		for (int r = 0; r < G; ++r)
			if (r != g)
				HIP_TRY(h, hipStreamWaitEvent(h->shards[g].stream, h->shards[r].ev_done, 0));
	}
	return EXPANN_OK;
}

// in-process search of host queries; deferred = the shards' searches are enqueued without a host
// wait on any device (expann_sync validates them at the end)
int search_inprocess(expann_sharded* h, const void* queries, size_t m, size_t k, uint64_t* ids, float* dists,
                     bool deferred) {
	const int G = h->n_active;
	const size_t cb = chunk_bytes(m, k);
	const size_t qb = m * (size_t)h->dim * h->q_elem;
	for (int r = 0; r < G; ++r) {
		Shard& s = h->shards[r];
		int rc = ensure_chunks(h, s, cb, G);
		if (rc != EXPANN_OK)
			return rc;
		HIP_TRY(h, hipSetDevice(s.device));
		if (qb > s.q_bytes) {
			HIP_TRY(h, hipStreamSynchronize(s.stream));
			if (s.d_q) (void)hipFree(s.d_q);
			s.d_q = nullptr;
			s.q_bytes = 0;
			HIP_TRY(h, hipMalloc(&s.d_q, qb));
			s.q_bytes = qb;
		}
		SUB_TRY(h, s, expann_set_option(s.idx, "async_search", deferred ? 1 : 0));
		HIP_TRY(h, hipMemcpyAsync(s.d_q, queries, qb, hipMemcpyHostToDevice, s.stream));
		SUB_TRY(h, s, expann_search_device(s.idx, s.d_q, m, k, reinterpret_cast<uint64_t*>(s.mine),
		                                   reinterpret_cast<float*>(s.mine + m * k * 8), s.stream));
	}
	if (G == 1 && h->exchange_used != 1) {  // one shard: its result is the result
		Shard& s = h->shards[0];
		HIP_TRY(h, hipSetDevice(s.device));
		if (deferred && expann_sync(s.idx) != EXPANN_OK)
			return EXPANN_ERR_OVERFLOW;
		HIP_TRY(h, hipMemcpyAsync(ids, s.mine, sizeof(uint64_t) * m * k, hipMemcpyDeviceToHost, s.stream));
		if (dists)
			HIP_TRY(h, hipMemcpyAsync(dists, s.mine + m * k * 8, sizeof(float) * m * k, hipMemcpyDeviceToHost, s.stream));
		HIP_TRY(h, hipStreamSynchronize(s.stream));
		return EXPANN_OK;
	}
	int rc = exchange_chunks(h, cb);
	if (rc != EXPANN_OK)
		return rc;
	// device r merges queries [r m / G, (r+1) m / G) out of its gathered copy
	for (int r = 0; r < G; ++r) {
		Shard& s = h->shards[r];
		const size_t q0 = (size_t)r * m / (size_t)G, q1 = (size_t)(r + 1) * m / (size_t)G;
		if (q1 == q0)
			continue;
		const uint64_t* in_ids = reinterpret_cast<const uint64_t*>(s.gathered) + q0 * k;
		const float* in_d = reinterpret_cast<const float*>(s.gathered + m * k * 8) + q0 * k;
		uint64_t* out_ids = reinterpret_cast<uint64_t*>(s.merged);
		float* out_d = reinterpret_cast<float*>(s.merged + (q1 - q0) * k * 8);
		if (expann_merge_topk_strided_device(s.device, in_ids, in_d, cb / 8, cb / 4, (size_t)G, q1 - q0, k, out_ids,
		                                     out_d, s.stream) != EXPANN_OK)
			return h->fail(EXPANN_ERR_HIP, std::string("merge: ") + expann_last_error(nullptr));
	}
	bool bad = false;
	for (int r = 0; r < G; ++r) {  // validate the deferred searches, then fetch the merged slices
		Shard& s = h->shards[r];
		HIP_TRY(h, hipSetDevice(s.device));
		if (deferred && expann_sync(s.idx) != EXPANN_OK)
			bad = true;
	}
	for (int r = 0; r < G && !bad; ++r) {
		Shard& s = h->shards[r];
		const size_t q0 = (size_t)r * m / (size_t)G, q1 = (size_t)(r + 1) * m / (size_t)G;
		if (q1 == q0)
			continue;
		HIP_TRY(h, hipSetDevice(s.device));
		HIP_TRY(h, hipMemcpyAsync(ids + q0 * k, s.merged, sizeof(uint64_t) * (q1 - q0) * k, hipMemcpyDeviceToHost,
		                          s.stream));
		if (dists)
			HIP_TRY(h, hipMemcpyAsync(dists + q0 * k, s.merged + (q1 - q0) * k * 8, sizeof(float) * (q1 - q0) * k,
			                          hipMemcpyDeviceToHost, s.stream));
	}
	for (int r = 0; r < G; ++r) {
		HIP_TRY(h, hipSetDevice(h->shards[r].device));
		HIP_TRY(h, hipStreamSynchronize(h->shards[r].stream));
	}
	return bad ? EXPANN_ERR_OVERFLOW : EXPANN_OK;
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
	if (world < 1 || rank < 0 || rank >= world) {
		g_create_error = "expann_sharded_create_rank: bad rank / world";
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
	// function (expann_sharded_set_exchange_fn); a one-rank id gives the RCCL path on one GPU
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
	SUB_TRY(h, s, expann_set_base_device(s.idx, d_rows, n, id_offset));
	if (s.n == 0 && !h->rank_form)
		h->n_active++;
	h->n_total += n - s.n;
	s.n = n;
	s.id_offset = id_offset;
	h->comm_ready = h->rank_form;  // (in-process: the set of shards in use may have grown)
	return EXPANN_OK;
}

size_t expann_sharded_size(const expann_sharded* h) { return h ? (h->n_total ? h->n_total : h->n_staged) : 0; }
int expann_sharded_shards(const expann_sharded* h) { return h ? (h->rank_form ? h->world : h->n_active) : 0; }
int expann_sharded_exchange(const expann_sharded* h) { return h ? h->exchange_used : 0; }

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
	h->searches++;
	rc = search_inprocess(h, queries, m, k, ids, dists, h->opt_async != 0);
	if (rc == EXPANN_ERR_OVERFLOW && h->opt_async) {
		// a deferred search needed the synchronous retry (overflowed lists, queries outside the
		// filter's range): once more, every shard waiting for its own search
		h->retries++;
		rc = search_inprocess(h, queries, m, k, ids, dists, false);
	}
	return rc;
}

int expann_sharded_search_device(expann_sharded* h, const void* d_queries, size_t m, size_t k, uint64_t* d_ids,
                                 float* d_dists, void* stream) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "in-process form: use expann_sharded_search");
	Shard& s = h->shards[0];
	if (s.n == 0)
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
	if (!s.comm && !h->exchange_fn) {  // one rank, no communicator: the local result is the result
		if (h->world > 1)
			return h->fail(EXPANN_ERR_INVALID_ARG,
			               "rank form without a unique id: set the exchange with expann_sharded_set_exchange_fn");
		SUB_TRY(h, s, expann_search_device(s.idx, d_queries, m, k, d_ids, d_dists, st));
		return EXPANN_OK;
	}
	const size_t cb = chunk_bytes(m, k);
	if (cb > s.chunk_alloc) {
		if (s.chunk_alloc)
			HIP_TRY(h, hipStreamSynchronize(st));
		int rc = ensure_chunks(h, s, cb, h->world);
		if (rc != EXPANN_OK)
			return rc;
	}
	SUB_TRY(h, s, expann_search_device(s.idx, d_queries, m, k, reinterpret_cast<uint64_t*>(s.mine),
	                                   reinterpret_cast<float*>(s.mine + m * k * 8), st));
	if (h->exchange_fn) {
		if (h->exchange_fn(h->exchange_ctx, s.mine, s.gathered, cb, h->rank, h->world, (void*)st) != 0)
			return h->fail(EXPANN_ERR_HIP, "the caller's exchange function failed");
	} else {
		NCCL_TRY(h, ncclAllGather(s.mine, s.gathered, cb, ncclChar, s.comm, st));
	}
	if (expann_merge_topk_strided_device(s.device, reinterpret_cast<const uint64_t*>(s.gathered),
	                                     reinterpret_cast<const float*>(s.gathered + m * k * 8), cb / 8, cb / 4,
	                                     (size_t)h->world, m, k, d_ids, d_dists, st) != EXPANN_OK)
		return h->fail(EXPANN_ERR_HIP, std::string("merge: ") + expann_last_error(nullptr));
	return EXPANN_OK;
}

int expann_sharded_set_exchange_fn(expann_sharded* h, expann_exchange_fn fn, void* ctx) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	if (!h->rank_form)
		return h->fail(EXPANN_ERR_INVALID_ARG, "the in-process form exchanges between its own devices");
	h->exchange_fn = fn;
	h->exchange_ctx = ctx;
	h->exchange_used = fn ? 3 : (h->shards[0].comm ? 1 : 0);
	return EXPANN_OK;
}

int expann_sharded_sync(expann_sharded* h) {
	if (!h)
		return EXPANN_ERR_INVALID_ARG;
	int bad = EXPANN_OK;
	for (auto& s : h->shards)
		if (s.idx && s.n) {
			HIP_TRY(h, hipSetDevice(s.device));
			const int rc = expann_sync(s.idx);
			if (rc != EXPANN_OK)
				bad = h->fail(rc, std::string("shard on device ") + std::to_string(s.device) + ": " +
				                      expann_last_error(s.idx));
		}
	return bad;
}

int expann_sharded_set_option(expann_sharded* h, const char* name, long value) {
	if (!h || !name)
		return EXPANN_ERR_INVALID_ARG;
	if (!std::strcmp(name, "exchange")) {
		if (value < 0 || value > 2)
			return h->fail(EXPANN_ERR_INVALID_ARG, "exchange: 0 auto, 1 RCCL, 2 device copies");
		if (h->rank_form)
			return h->fail(EXPANN_ERR_INVALID_ARG, "rank form always exchanges over RCCL");
		h->exchange = (int)value;
		h->comm_ready = false;
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
