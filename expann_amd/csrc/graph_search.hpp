// graph_search.hpp -- query side of the reference's graph engine on the device:
//   antitopo_engine::_query_k            src/antitopo_engine.h:853-928
//   query_k_at_layer<true,false,false>   src/antitopo_engine.h:495-708   (fp32 bottom layer)
//   query_k_bottom_compressed            src/antitopo_engine.h:710-851   (uint8 bottom layer +
//                                        final fp32 re-score :845-848)
// One wavefront (= one 64-thread workgroup) walks one query; workgroups are persistent and pull
// queries round-robin.  The traversal is a dependent chain, so the parallelism inside a query is
// the neighbour batch of a hop: the "visited" tests of a vertex's <= M0 neighbours run one lane
// per neighbour, and their distances are evaluated 4 rows at a time with 16 lanes per row in the
// reference's exact FMA order (common.hpp) -- the same arithmetic as the brute-force scan, so
// every distance, and therefore every branch of the walk, is bit-identical to the CPU code.
// The two priority queues live in LDS and are updated with exactly libstdc++'s push_heap / pop_heap
// element movement (the reference's comparators look at .first only, so the order among equal
// distances is the heap's; integer-valued uint8 distances tie often) -- by the WHOLE WAVE (round 3):
// the path of a sift is known before anything moves (the ancestors of the hole for __push_heap; the
// chain of larger children for __adjust_heap, which libstdc++ walks to a leaf before it looks at the
// value), so one lane per level reads its element, a ballot finds where the value settles and the
// lanes shift their elements in one step: the same final array as the serial code, in ~3 LDS round
// trips instead of ~3 per level (coop_push / coop_pop below; round 2's lane-0 loops took 40-52 % of a
// hop, profiles/r03_c4_hop_breakdown.txt).
// A neighbour that cannot beat the current worst kept distance is dropped before the queue update
// (all lanes test, ballot-compacted in list order): `nearest.top()` only decreases while the queue is
// full, so the sequential test of :666 / :823 would reject it too.
// Visited set (the reference keeps a byte array and a reset list, :208-209, :692-694): a BITSET per
// resident workgroup -- test-and-set by one returning atomicOr, cleared by streaming stores when the
// query ends (cheaper than a reset list's scattered stores up to a few million vertices); beyond 8 M
// vertices one epoch-stamped byte per vertex and workgroup, never cleared (round 2's only form).
#pragma once
#include <type_traits>

#include "common.hpp"
#include "scan_int8.hpp"

namespace expann {

struct GraphSearchParams {
	const float* vectors;         // [n][D]
	const uint8_t* compressed;    // [n][D] (quantizer_simple<uint8_t>) or nullptr
	const uint32_t* layer_off;    // [n_layers][n+1] (CSR; the walk uses it for the layers above 0)
	const uint32_t* neighbours;
	const uint32_t* adj0;         // [n][stride0] layer-0 lists at a fixed stride, padded with UINT32_MAX: a
	uint32_t stride0;             // hop reads its list without the dependent round trip for the offsets
	uint32_t n, n_layers, starting_vertex;
	const float* queries;         // [m][D]
	uint32_t m;
	uint32_t k, ef;
	uint32_t cand_cap;            // capacity of the candidates heap (LDS)
	uint32_t max_degree;          // longest neighbour list of layer 0 (<= list_cap)
	uint32_t list_cap;
	uint8_t* visited;             // [gridDim.x][n] epoch bytes, zero-initialised (vis_words == 0)
	uint32_t* epochs;             // [gridDim.x] last epoch used by that workgroup
	// vis_words != 0: the visited set is a BITSET per workgroup, vis_bits[gridDim.x][vis_words] (vis_words a
	// multiple of 256, all zero between queries): test-and-set is one returning atomicOr, the set is
	// cleared with streaming stores when the query ends.  An eighth of the bytes' footprint: at 1 M rows
	// the resident workgroups' sets are 0.4 GB instead of 3-4 GB and mostly live in the Infinity Cache.
	uint32_t* vis_bits;
	uint32_t vis_words;
	uint64_t* out_ids;            // [m][k]
	float* out_dists;             // [m][k]
	uint32_t* out_distcomps;      // [m] or nullptr
	uint32_t* error;              // [1] != 0: a candidates heap overflowed cand_cap
	uint32_t* next_query;         // [1] zero at launch: the workgroups of the (resident) grid pull queries from it
	uint32_t debug;               // bisection switch (EXPANN_GRAPH_DEBUG): 1 = lane 0 runs the serial queue code
	// DBG instance only (profiles/r03_c4_hop_breakdown.txt): stamps[blockIdx.x][8] = shader clocks this
	// workgroup spent in {0 per-query setup + descent, 1 pop + broadcast, 2 adjacency + visited tests,
	// 3 row gathers + scoring, 4 serial queue update, 5 output}, 6 = hops, 7 = queue insertions
	unsigned long long* stamps;
};

struct md_pair {
	float d;
	uint32_t id;
};

// libstdc++ heap primitives on an LDS array; MAXH: top = largest .d (worst_elem), else smallest
template <bool MAXH> __device__ inline bool md_less(md_pair a, md_pair b) {
	return MAXH ? (a.d < b.d) : (a.d > b.d);
}
template <bool MAXH>
__device__ inline void heap_push_up(md_pair* v, uint32_t hole, uint32_t top, md_pair value) {
	while (hole > top) {
		const uint32_t parent = (hole - 1) / 2;
		if (!md_less<MAXH>(v[parent], value))
			break;
		v[hole] = v[parent];
		hole = parent;
	}
	v[hole] = value;
}
template <bool MAXH>
__device__ inline void heap_adjust(md_pair* v, uint32_t hole, uint32_t len, md_pair value) {
	const uint32_t top = hole;
	uint32_t child = hole;
	while (len > 1 && child < (len - 1) / 2) {
		child = 2 * (child + 1);
		if (md_less<MAXH>(v[child], v[child - 1]))
			--child;
		v[hole] = v[child];
		hole = child;
	}
	if ((len & 1) == 0 && len >= 2 && child == (len - 2) / 2) {
		child = 2 * (child + 1);
		v[hole] = v[child - 1];
		hole = child - 1;
	}
	heap_push_up<MAXH>(v, hole, top, value);
}
template <bool MAXH> __device__ inline void heap_push(md_pair* v, uint32_t& n, md_pair e) {
	v[n] = e;
	heap_push_up<MAXH>(v, n, 0, e);
	++n;
}
template <bool MAXH> __device__ inline void heap_pop(md_pair* v, uint32_t& n) {
	if (n > 1) {
		const md_pair value = v[n - 1];
		v[n - 1] = v[0];
		heap_adjust<MAXH>(v, 0, n - 1, value);
	}
	--n;
}

// min of a 64-bit key over the wave (all lanes get it)
__device__ inline uint64_t wave_min_key64(uint64_t v) {
	for (int off = 32; off > 0; off >>= 1) {
		const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off);
		const uint64_t o = ((uint64_t)hi << 32) | lo;
		v = o < v ? o : v;
	}
	return v;
}

// ---- the queues, updated by the whole wave --------------------------------------------------------
// (every lane calls with the same arguments; n and the results are wave-uniform)
__device__ inline void wave_lds_sync() {  // lanes exchange data through LDS: keep the compiler's order
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// std::__push_heap(first, hole, top = 0, value): lane j reads the j-th ancestor of the hole; the value
// stops in front of the first ancestor that is not "less" than it; the ancestors below move down one.
template <bool MAXH> __device__ inline void coop_push_up(md_pair* v, uint32_t hole, md_pair value, int lane) {
	asm volatile("" : "+v"(lane));  // (opaque: the lanes' heap addresses are formed here -- hoisted out of the hop loop they
	                                // spilled in the 128-register uint8 instance, a scratch round trip per queue update)
	const uint32_t h1 = hole + 1;
	const int levels = 31 - __builtin_clz(h1);  // ancestors of the hole
	const bool act = lane < levels;
	const md_pair x = v[act ? (h1 >> ((lane + 1) & 31)) - 1 : 0];
	// (every lane's element is in its registers before any lane stores: without the fence the compiler
	// sinks the load of x.id -- the ballot only needs x.d -- into the branch that uses it, behind another
	// lane's store to the same slot)
	wave_lds_sync();
	const unsigned long long stops = __builtin_amdgcn_ballot_w64(act && !md_less<MAXH>(x, value));
	const int stop = stops ? __builtin_ctzll(stops) : levels;
	const uint32_t pos = (h1 >> (lane & 31)) - 1;  // lane 0: the hole; lane j: ancestor j - 1 (lanes <= stop <= 31 write)
	if (lane < stop)
		v[pos] = x;
	else if (lane == stop)
		v[pos] = value;
}
template <bool MAXH> __device__ inline void coop_push(md_pair* v, uint32_t& n, md_pair e, int lane, uint32_t serial = 0) {
	if (serial) {
		wave_lds_sync();
		if (lane == 0) {
			v[n] = e;
			heap_push_up<MAXH>(v, n, 0, e);
		}
		wave_lds_sync();
	} else {
		coop_push_up<MAXH>(v, n, e, lane);
	}
	++n;
}

// std::__adjust_heap(first, hole = 0, len, value) for len <= 128 * NPL + 1: lane i (+ 64 s) compares the
// two children of node i, the chain of larger children is followed with readlanes (one per level),
// then -- as libstdc++ does -- the value is pushed up from the leaf along that very chain: lane t holds
// the chain's element t + 1, a ballot finds where the value settles.
template <bool MAXH, int NPL>
__device__ inline void coop_adjust(md_pair* v, uint32_t len, md_pair value, int lane) {
	asm volatile("" : "+v"(lane));  // (opaque, as in coop_push_up)
	const uint32_t half = (len - 1) / 2;  // nodes < half have two children
	uint32_t big[NPL];
#pragma unroll
	for (int s = 0; s < NPL; ++s) {
		const uint32_t i = (uint32_t)lane + 64u * s;
		big[s] = 0;
		if (i < half) {
			const uint32_t c2 = 2 * i + 2;
			const md_pair a = v[c2 - 1], b = v[c2];
			big[s] = md_less<MAXH>(b, a) ? c2 - 1 : c2;
		}
	}
	uint32_t cur = 0, depth = 0, my_c = 0, my_cn = 0;
	auto step_to = [&](uint32_t nxt) {
		++depth;
		if ((uint32_t)lane == depth)
			my_c = nxt;
		if ((uint32_t)lane + 1 == depth)
			my_cn = nxt;
		cur = nxt;
	};
	while (cur < half) {
		const uint32_t ln = cur & 63u, sl = cur >> 6;
		uint32_t nxt = (uint32_t)__builtin_amdgcn_readlane((int)big[0], (int)ln);
#pragma unroll
		for (int s = 1; s < NPL; ++s)
			if (sl == (uint32_t)s)
				nxt = (uint32_t)__builtin_amdgcn_readlane((int)big[s], (int)ln);
		step_to(nxt);
	}
	if ((len & 1) == 0 && cur == (len - 2) / 2)
		step_to(2 * cur + 1);  // the last node has a left child only
	const bool on_chain = (uint32_t)lane < depth;
	const md_pair z = v[on_chain ? my_cn : 0];
	wave_lds_sync();  // (as in coop_push_up: all loads before any lane's store)
	const unsigned long long rises = __builtin_amdgcn_ballot_w64(on_chain && md_less<MAXH>(z, value));
	const unsigned long long stay = ~rises & ((depth >= 64 ? 0ull : (1ull << depth)) - 1ull);
	const uint32_t settle = stay ? 64u - (uint32_t)__builtin_clzll(stay) : 0u;
	if ((uint32_t)lane < settle)
		v[my_c] = z;
	else if ((uint32_t)lane == settle)
		v[my_c] = value;
}
// the same for any length: the chain of larger children is walked one level at a time (every lane reads
// the same two children: LDS broadcasts), the tail is coop_adjust's
template <bool MAXH> __device__ inline void coop_adjust_deep(md_pair* v, uint32_t len, md_pair value, int lane) {
	const uint32_t half = (len - 1) / 2;
	uint32_t cur = 0, depth = 0, my_c = 0, my_cn = 0;
	auto step_to = [&](uint32_t nxt) {
		++depth;
		if ((uint32_t)lane == depth)
			my_c = nxt;
		if ((uint32_t)lane + 1 == depth)
			my_cn = nxt;
		cur = nxt;
	};
	while (cur < half) {
		const uint32_t c2 = 2 * cur + 2;
		const float a = v[c2 - 1].d, b = v[c2].d;
		step_to((MAXH ? (b < a) : (b > a)) ? c2 - 1 : c2);
	}
	if ((len & 1) == 0 && cur == (len - 2) / 2)
		step_to(2 * cur + 1);
	const bool on_chain = (uint32_t)lane < depth;
	const md_pair z = v[on_chain ? my_cn : 0];
	wave_lds_sync();
	const unsigned long long rises = __builtin_amdgcn_ballot_w64(on_chain && md_less<MAXH>(z, value));
	const unsigned long long stay = ~rises & ((depth >= 64 ? 0ull : (1ull << depth)) - 1ull);
	const uint32_t settle = stay ? 64u - (uint32_t)__builtin_clzll(stay) : 0u;
	if ((uint32_t)lane < settle)
		v[my_c] = z;
	else if ((uint32_t)lane == settle)
		v[my_c] = value;
}
// std::pop_heap + pop_back
template <bool MAXH> __device__ inline void coop_pop(md_pair* v, uint32_t& n, int lane, uint32_t serial = 0) {
	if (n > 1) {
		// (libstdc++ also parks the old top in slot n - 1; nothing ever reads a slot beyond the queue's
		// size, so that store -- and the two fences it would need around it -- is left out)
		const md_pair value = v[n - 1];
		const uint32_t len = n - 1;
		if (serial) {
			wave_lds_sync();
			if (lane == 0)
				heap_adjust<MAXH>(v, 0, len, value);
		} else if (len <= 129) {
			coop_adjust<MAXH, 1>(v, len, value, lane);
		} else if (len <= 513) {
			coop_adjust<MAXH, 4>(v, len, value, lane);
		} else {  // long queues (the builder's candidates at ef_construction = 480): walk the chain level by level
			coop_adjust_deep<MAXH>(v, len, value, lane);
		}
		wave_lds_sync();
	}
	--n;
}

// Test hook (expann_device_heap_trace): one wave replays a trace of queue operations through the very
// coop_push / coop_pop the walk uses, so that tests/golden/heap_ref.json -- answered by the image's real
// std::priority_queue -- pins the device queues directly.  The range constructor's make_heap runs the
// serial __adjust_heap (the walk only ever constructs one-element queues).
struct HeapTraceParams {
	int max_heap;
	uint32_t n_init, n_ops;
	const float* init_d;
	const uint32_t* init_id;
	const int* ops;       // 1 = push (op_d, op_id), 0 = pop
	const float* op_d;
	const uint32_t* op_id;
	uint32_t* out_size;   // [n_ops + 1]
	float* out_top_d;
	uint32_t* out_top_id;
	float* drain_d;       // [n_init + n_ops]
	uint32_t* drain_id;
	uint32_t* n_drain;
	uint32_t serial;
};
template <bool MAXH> __device__ inline void heap_trace_body(const HeapTraceParams& p, md_pair* v, int lane) {
	uint32_t n = p.n_init;
	for (uint32_t i = lane; i < n; i += 64)
		v[i] = md_pair{p.init_d[i], p.init_id[i]};
	wave_lds_sync();
	if (lane == 0 && n >= 2)  // std::__make_heap
		for (uint32_t parent = (n - 2) / 2;; --parent) {
			const md_pair value = v[parent];
			heap_adjust<MAXH>(v, parent, n, value);
			if (parent == 0)
				break;
		}
	wave_lds_sync();
	for (uint32_t i = 0;; ++i) {
		const md_pair top = v[0];
		if (lane == 0) {
			p.out_size[i] = n;
			p.out_top_d[i] = n ? top.d : 0.0f;
			p.out_top_id[i] = n ? top.id : 0u;
		}
		if (i == p.n_ops)
			break;
		if (p.ops[i] == 1)
			coop_push<MAXH>(v, n, md_pair{p.op_d[i], p.op_id[i]}, lane, p.serial);
		else if (n)
			coop_pop<MAXH>(v, n, lane, p.serial);
		wave_lds_sync();
	}
	uint32_t nd = 0;
	while (n) {
		const md_pair top = v[0];
		if (lane == 0) {
			p.drain_d[nd] = top.d;
			p.drain_id[nd] = top.id;
		}
		++nd;
		coop_pop<MAXH>(v, n, lane, p.serial);
		wave_lds_sync();
	}
	if (lane == 0)
		*p.n_drain = nd;
}
__global__ __launch_bounds__(64) void heap_trace_kernel(HeapTraceParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char heap_trace_smem[];
	md_pair* v = reinterpret_cast<md_pair*>(heap_trace_smem);
	if (p.max_heap)
		heap_trace_body<true>(p, v, (int)threadIdx.x);
	else
		heap_trace_body<false>(p, v, (int)threadIdx.x);
}

// rows in flight per 16-lane group while a hop's neighbours are scored
template <int D> constexpr int graph_rows_f32() { return D <= 128 ? 8 : (D <= 256 ? 4 : 2); }
template <int D> constexpr int graph_rows_u8() { return D <= 128 ? 16 : (D <= 256 ? 8 : 4); }

// (the uint8 walk waits on latency, not bandwidth: 128 registers = 16 waves per CU instead of 12)
template <int D, bool COMPRESSED, int DBG = 0>
__global__ __launch_bounds__(64, (COMPRESSED && D <= 128) ? 4 : 1) void graph_search_kernel(GraphSearchParams p) {
	constexpr int DPL = D / 16;
	constexpr int NW = D / 64;
	unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts = DBG ? clock64() : 0;
	auto stamp = [&](int i) {
		if (DBG) {
			const unsigned long long now = clock64();
			seg[i] += now - ts;
			ts = now;
		}
	};
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	md_pair* nearest = reinterpret_cast<md_pair*>(smem_raw);             // [ef + 1]
	md_pair* candidates = nearest + (p.ef + 1);                          // [cand_cap + 1]
	uint32_t* nlist = reinterpret_cast<uint32_t*>(candidates + (p.cand_cap + 1));  // [list_cap]
	float* ndist = reinterpret_cast<float*>(nlist + p.list_cap);          // [list_cap]

	const int lane = threadIdx.x;
	const int l = lane & 15, rg = lane >> 4;
	const bool bits = p.vis_words != 0;
	uint8_t* visited = bits ? nullptr : p.visited + (size_t)blockIdx.x * p.n;
	uint32_t* vbits = bits ? p.vis_bits + (size_t)blockIdx.x * p.vis_words : nullptr;
	uint32_t epoch = bits ? 0u : p.epochs[blockIdx.x];
	// is vertex nb new to this query?  (marks it)
	auto test_and_set = [&](uint32_t nb, uint8_t ep8) -> bool {
		if (bits)
			return (atomicOr(&vbits[nb >> 5], 1u << (nb & 31)) & (1u << (nb & 31))) == 0;
		const bool fresh = visited[nb] != ep8;
		if (fresh)
			visited[nb] = ep8;
		return fresh;
	};
	uint32_t overflowed = 0;

	for (;;) {
		// queries are pulled, not dealt: walks differ in length, and a grid sized to what is resident
		// ends evenly
		uint32_t qi = 0;
		if (lane == 0)
			qi = atomicAdd(p.next_query, 1u);
		qi = (uint32_t)__builtin_amdgcn_readfirstlane((int)qi);
		if (qi >= p.m)
			break;
		// ---- per-query setup ----------------------------------------------------------
		if (!bits && ++epoch > 255) {  // epoch bytes wrapped: clear this workgroup's visited array
			for (uint32_t i = lane; i < p.n; i += 64)
				visited[i] = 0;
			epoch = 1;
		}
		const uint8_t ep8 = (uint8_t)epoch;
		float q[DPL];
#pragma unroll
		for (int t = 0; t < DPL; ++t)
			q[t] = p.queries[(size_t)qi * D + l + 16 * t];
		int q8[NW];  // trunc(query) bytes of this lane (compressed path)
		int q8self = 0;
		if (COMPRESSED) {
#pragma unroll
			for (int w = 0; w < NW; ++w) {
				unsigned packed = 0;
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const float v = p.queries[(size_t)qi * D + l * (D / 16) + 4 * w + b];
					packed |= ((unsigned)(uint8_t)(uint32_t)v) << (8 * b);
				}
				q8[w] = (int)packed;
				q8self = dot4<kU8L2>(q8[w], q8[w], q8self);
			}
			q8self = reduce16_i32(q8self);
		}
		// exact fp32 squared L2 of the query against U rows per 16-lane group (reference order), all
		// rows requested before the first is consumed
		auto dist_f32_rows = [&](auto u_tag, const uint32_t* rows, float* d) {
			constexpr int U = decltype(u_tag)::value;
			float r[U][DPL];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const float* src = p.vectors + (size_t)rows[u] * D + l;
#pragma unroll
				for (int t = 0; t < DPL; ++t)
					r[u][t] = src[16 * t];
			}
#pragma unroll
			for (int u = 0; u < U; ++u) {
				float acc = 0.0f;
#pragma unroll
				for (int t = 0; t < DPL; ++t) {
					const float diff = q[t] - r[u][t];
					acc = __builtin_fmaf(diff, diff, acc);
				}
				d[u] = reduce16_ref_order(acc);
			}
		};
		auto dist_u8_rows = [&](auto u_tag, const uint32_t* rows, float* d) {
			constexpr int U = decltype(u_tag)::value;
			int b[U][NW];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int* r = reinterpret_cast<const int*>(p.compressed + (size_t)rows[u] * D) + l * NW;
#pragma unroll
				for (int w = 0; w < NW; ++w)
					b[u][w] = r[w];
			}
#pragma unroll
			for (int u = 0; u < U; ++u) {
				int bself = 0;
#pragma unroll
				for (int w = 0; w < NW; ++w)
					bself = dot4<kU8L2>(b[u][w], b[u][w], bself);
				const int part = partial_score<kU8L2, NW>(q8, b[u], bself);
				d[u] = (float)(reduce16_i32(part) + q8self);
			}
		};
		auto dist_f32 = [&](uint32_t row) -> float {
			float d1[1];
			dist_f32_rows(std::integral_constant<int, 1>{}, &row, d1);
			return d1[0];
		};
		uint32_t distcomps = 0;

		// ---- upper layers: greedy descent (:863-902) -----------------------------------
		uint32_t entry = p.starting_vertex;
		float ep_dist = dist_f32(entry);  // every 16-lane row computes the same value
		++distcomps;
		for (uint32_t layer = p.n_layers - 1; layer > 0; --layer) {
			const uint32_t* off = p.layer_off + (size_t)layer * (p.n + 1);
			bool changed = true;
			while (changed) {
				changed = false;
				const uint32_t e0 = off[entry], deg = off[entry + 1] - e0;
				// first-improvement chain == first occurrence of the minimum, if it improves: the smallest
				// (distance bits, list position) key of the list (distances are >= 0: their bits order them)
				constexpr int UD = COMPRESSED ? (graph_rows_f32<D>() > 4 ? 4 : graph_rows_f32<D>()) : graph_rows_f32<D>();
				uint64_t best_key = ~0ull;
				uint32_t best_nb = 0;
				for (uint32_t i0 = 0; i0 < deg; i0 += 4 * UD) {
					uint32_t nb[UD];
					float d[UD];
#pragma unroll
					for (int u = 0; u < UD; ++u) {
						const uint32_t i = i0 + 4 * u + rg;
						nb[u] = p.neighbours[e0 + (i < deg ? i : deg - 1)];
					}
					dist_f32_rows(std::integral_constant<int, UD>{}, nb, d);
#pragma unroll
					for (int u = 0; u < UD; ++u) {
						const uint32_t i = i0 + 4 * u + rg;
						const uint64_t key = ((uint64_t)__builtin_bit_cast(uint32_t, d[u]) << 32) | i;
						if (i < deg && key < best_key) {
							best_key = key;
							best_nb = nb[u];
						}
					}
				}
				const uint64_t wmin = wave_min_key64(best_key);
				distcomps += deg;
				const float dmin = __builtin_bit_cast(float, (uint32_t)(wmin >> 32));
				if (wmin != ~0ull && dmin < ep_dist) {
					// the lane(s) that hold the winning key broadcast its vertex
					const unsigned long long who = __builtin_amdgcn_ballot_w64(best_key == wmin);
					entry = (uint32_t)__builtin_amdgcn_readlane((int)best_nb, __builtin_ctzll(who));
					ep_dist = dmin;
					changed = true;
				}
			}
		}
		stamp(0);

		// ---- bottom layer: best-first search --------------------------------------------
		uint32_t n_near = 0, n_cand = 0;  // wave-uniform
		float d_entry;
		{
			float d1[1];
			if (COMPRESSED)
				dist_u8_rows(std::integral_constant<int, 1>{}, &entry, d1);
			else
				dist_f32_rows(std::integral_constant<int, 1>{}, &entry, d1);
			d_entry = d1[0];
		}
		++distcomps;
		{
			const md_pair e{d_entry, entry};
			coop_push<false>(candidates, n_cand, e, lane, p.debug & 1);
			coop_push<true>(nearest, n_near, e, lane, p.debug & 1);
			if (lane == 0)
				(void)test_and_set(entry, ep8);
		}
		wave_lds_sync();
		for (;;) {
			// pop the best candidate (:588-594)
			if (n_cand == 0)
				break;
			const md_pair cur = candidates[0];
			coop_pop<false>(candidates, n_cand, lane, p.debug & 1);
			const float worst0 = nearest[0].d;
			const bool full0 = n_near == p.ef;
			stamp(1);
			if (cur.d > worst0 && full0)
				break;
			if (DBG)
				seg[6]++;
			// unvisited neighbours, in adjacency order (:595-607): the list sits at a fixed stride, both
			// halves of a list longer than a wavefront are requested before either is consumed
			const uint32_t* adj = p.adj0 + (size_t)cur.id * p.stride0;
			uint32_t nbv[2];
			bool fresh[2];
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const uint32_t i = 64u * h + lane;
				nbv[h] = i < p.stride0 ? adj[i] : 0xFFFFFFFFu;
			}
#pragma unroll
			for (int h = 0; h < 2; ++h)
				fresh[h] = nbv[h] != 0xFFFFFFFFu && test_and_set(nbv[h], ep8);
			uint32_t n_list = 0;
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const unsigned long long mask = __builtin_amdgcn_ballot_w64(fresh[h]);
				if (fresh[h])
					nlist[n_list + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = nbv[h];
				n_list += (uint32_t)__builtin_popcountll(mask);
			}
			for (uint32_t i0 = 128; i0 < p.stride0; i0 += 64) {  // (lists beyond 128: M0 > 128)
				const uint32_t i = i0 + lane;
				const uint32_t nb = i < p.stride0 ? adj[i] : 0xFFFFFFFFu;
				const bool fr = nb != 0xFFFFFFFFu && test_and_set(nb, ep8);
				const unsigned long long mask = __builtin_amdgcn_ballot_w64(fr);
				if (fr)
					nlist[n_list + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = nb;
				n_list += (uint32_t)__builtin_popcountll(mask);
			}
			wave_lds_sync();
			stamp(2);
			// score them (:636-689 / :795-835): 4 rows per step (16 lanes each), U steps' rows requested
			// before the first is consumed -- a hop is a dependent chain of HBM round trips otherwise
			constexpr int U = COMPRESSED ? graph_rows_u8<D>() : graph_rows_f32<D>();
			for (uint32_t i0 = 0; i0 < n_list; i0 += 4 * U) {
				uint32_t nb[U];
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const uint32_t i = i0 + 4 * u + rg;
					nb[u] = nlist[i < n_list ? i : n_list - 1];
				}
				float d[U];
				if (COMPRESSED)
					dist_u8_rows(std::integral_constant<int, U>{}, nb, d);
				else
					dist_f32_rows(std::integral_constant<int, U>{}, nb, d);
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const uint32_t i = i0 + 4 * u + rg;
					if (l == 0 && i < n_list)
						ndist[i] = d[u];
				}
			}
			distcomps += n_list;
			wave_lds_sync();
			stamp(3);
			// queue update in list order (:666-670): first every lane drops what cannot pass `d_next <
			// nearest.top().first` any more (the top only falls while the queue is full), the survivors are
			// compacted in place, in order; then one wave-wide queue operation per survivor
			uint32_t n_s = 0;
			for (uint32_t i0 = 0; i0 < n_list; i0 += 64) {  // (a chunk only writes slots it has already read)
				const uint32_t i = i0 + lane;
				const uint32_t id = i < n_list ? nlist[i] : 0u;
				const float dn = i < n_list ? ndist[i] : 0.0f;
				const bool keep = i < n_list && !(full0 && !(dn < worst0));
				wave_lds_sync();
				const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
				const uint32_t pos = n_s + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
				if (keep) {
					nlist[pos] = id;
					ndist[pos] = dn;
				}
				n_s += (uint32_t)__builtin_popcountll(mask);
			}
			wave_lds_sync();
			for (uint32_t j = 0; j < n_s; ++j) {
				const float dn = ndist[j];
				if (n_near < p.ef || dn < nearest[0].d) {
					const md_pair e{dn, nlist[j]};
					if (n_cand >= p.cand_cap)
						overflowed = 1;
					else
						coop_push<false>(candidates, n_cand, e, lane, p.debug & 1);
					coop_push<true>(nearest, n_near, e, lane, p.debug & 1);
					wave_lds_sync();
					if (n_near > p.ef)
						coop_pop<true>(nearest, n_near, lane, p.debug & 1);
					if (DBG)
						seg[7]++;
				}
			}
			wave_lds_sync();
			stamp(4);
		}
		// ---- output: drain nearest, reverse, (re-score), truncate to k (:695-707, :845-848) -
		const uint32_t cnt = n_near;
		for (uint32_t i = cnt; i-- > 0;) {
			const md_pair t = nearest[0];
			coop_pop<true>(nearest, n_near, lane, p.debug & 1);
			if (lane == 0) {
				nlist[i] = t.id;
				ndist[i] = t.d;
			}
		}
		wave_lds_sync();
		const uint32_t n_out = cnt < p.k ? cnt : p.k;
		if (COMPRESSED) {
			for (uint32_t i0 = 0; i0 < n_out; i0 += 4) {
				const uint32_t i = i0 + rg;
				const float d = dist_f32(nlist[i < n_out ? i : n_out - 1]);
				if (l == 0 && i < n_out)
					ndist[i] = d;
			}
			wave_lds_sync();
		}
		for (uint32_t i = lane; i < p.k; i += 64) {
			p.out_ids[(size_t)qi * p.k + i] = i < n_out ? (uint64_t)nlist[i] : ~0ull;
			p.out_dists[(size_t)qi * p.k + i] = i < n_out ? ndist[i] : __builtin_inff();
		}
		if (lane == 0 && p.out_distcomps)
			p.out_distcomps[qi] = distcomps;
		if (bits) {  // the set goes back to all-zero: 16 bytes per lane and store, nothing waits for them
			uint4* w = reinterpret_cast<uint4*>(vbits);
			for (uint32_t i = lane; i < p.vis_words / 4; i += 64)
				w[i] = make_uint4(0u, 0u, 0u, 0u);
		}
		wave_lds_sync();
		stamp(5);
	}
	if (DBG && lane == 0 && p.stamps)
		for (int i = 0; i < 8; ++i)
			p.stamps[(size_t)blockIdx.x * 8 + i] = seg[i];
	if (lane == 0) {
		if (!bits)
			p.epochs[blockIdx.x] = epoch;
		if (overflowed)
			atomicAdd(p.error, 1u);
	}
}

}  // namespace expann
