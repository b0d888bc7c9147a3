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
// The two priority queues live in LDS and are updated by lane 0 with exactly libstdc++'s
// push_heap / pop_heap element movement (the reference's comparators look at .first only, so
// the order among equal distances is the heap's; integer-valued uint8 distances tie often).
// A neighbour that cannot beat the current worst kept distance is dropped before the serial
// heap update: `nearest.top()` only decreases while the queue is full, so the sequential test of
// :666 / :823 would reject it too.
// Visited set: one byte per vertex per resident workgroup in HBM, stamped with a per-query epoch
// (the reference keeps a byte array and a reset list, :208-209, :692-694).
#pragma once
#include "common.hpp"
#include "scan_int8.hpp"

namespace expann {

struct GraphSearchParams {
	const float* vectors;         // [n][D]
	const uint8_t* compressed;    // [n][D] (quantizer_simple<uint8_t>) or nullptr
	const uint32_t* layer_off;    // [n_layers][n+1]
	const uint32_t* neighbours;
	uint32_t n, n_layers, starting_vertex;
	const float* queries;         // [m][D]
	uint32_t m;
	uint32_t k, ef;
	uint32_t cand_cap;            // capacity of the candidates heap (LDS)
	uint32_t max_degree;          // longest neighbour list of layer 0 (<= list_cap)
	uint32_t list_cap;
	uint8_t* visited;             // [gridDim.x][n] epoch bytes, zero-initialised
	uint32_t* epochs;             // [gridDim.x] last epoch used by that workgroup
	uint64_t* out_ids;            // [m][k]
	float* out_dists;             // [m][k]
	uint32_t* out_distcomps;      // [m] or nullptr
	uint32_t* error;              // [1] != 0: a candidates heap overflowed cand_cap
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

template <int D, bool COMPRESSED>
__global__ __launch_bounds__(64) void graph_search_kernel(GraphSearchParams p) {
	constexpr int DPL = D / 16;
	constexpr int NW = D / 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	md_pair* nearest = reinterpret_cast<md_pair*>(smem_raw);             // [ef + 1]
	md_pair* candidates = nearest + (p.ef + 1);                          // [cand_cap + 1]
	uint32_t* nlist = reinterpret_cast<uint32_t*>(candidates + (p.cand_cap + 1));  // [list_cap]
	float* ndist = reinterpret_cast<float*>(nlist + p.list_cap);          // [list_cap]
	uint32_t* ctl = reinterpret_cast<uint32_t*>(ndist + p.list_cap);      // control words

	const int lane = threadIdx.x;
	const int l = lane & 15, rg = lane >> 4;
	uint8_t* visited = p.visited + (size_t)blockIdx.x * p.n;
	uint32_t epoch = p.epochs[blockIdx.x];

	for (uint32_t qi = blockIdx.x; qi < p.m; qi += gridDim.x) {
		// ---- per-query setup ----------------------------------------------------------
		if (++epoch > 255) {  // epoch bytes wrapped: clear this workgroup's visited array
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
		// exact fp32 squared L2 of the query against row `row` (16 lanes, reference order)
		auto dist_f32 = [&](uint32_t row) -> float {
			const float* r = p.vectors + (size_t)row * D + l;
			float acc = 0.0f;
#pragma unroll
			for (int t = 0; t < DPL; ++t) {
				const float diff = q[t] - r[16 * t];
				acc = __builtin_fmaf(diff, diff, acc);
			}
			return reduce16_ref_order(acc);
		};
		auto dist_u8 = [&](uint32_t row) -> float {
			const int* r = reinterpret_cast<const int*>(p.compressed + (size_t)row * D) + l * NW;
			int b[NW];
			int bself = 0;
#pragma unroll
			for (int w = 0; w < NW; ++w) {
				b[w] = r[w];
				bself = dot4<kU8L2>(b[w], b[w], bself);
			}
			const int part = partial_score<kU8L2, NW>(q8, b, bself);
			return (float)(reduce16_i32(part) + q8self);
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
				// first-improvement chain == first occurrence of the minimum, if it improves
				float best = ep_dist;
				uint32_t best_id = entry;
				for (uint32_t i0 = 0; i0 < deg; i0 += 4) {
					const uint32_t i = i0 + rg;
					const uint32_t nb = p.neighbours[e0 + (i < deg ? i : deg - 1)];
					const float d = dist_f32(nb);
					// rows of this step in list order
#pragma unroll
					for (int g = 0; g < 4; ++g) {
						const float dg = __shfl(d, g * 16);
						const uint32_t ng = __shfl(nb, g * 16);
						if (i0 + g < deg && dg < best) {
							best = dg;
							best_id = ng;
							changed = true;
						}
					}
				}
				distcomps += deg;
				entry = best_id;
				ep_dist = best;
			}
		}

		// ---- bottom layer: best-first search --------------------------------------------
		const uint32_t* off0 = p.layer_off;
		uint32_t n_near = 0, n_cand = 0;
		const float d_entry = COMPRESSED ? dist_u8(entry) : dist_f32(entry);
		++distcomps;
		if (lane == 0) {
			const md_pair e{d_entry, entry};
			heap_push<false>(candidates, n_cand, e);
			heap_push<true>(nearest, n_near, e);
			visited[entry] = ep8;
			ctl[0] = 0;  // overflow flag
		}
		__syncthreads();
		for (;;) {
			// pop the best candidate (lane 0), broadcast through LDS
			if (lane == 0) {
				uint32_t go = 0, cur_id = 0;
				if (n_cand > 0) {
					const md_pair cur = candidates[0];
					heap_pop<false>(candidates, n_cand);
					if (!(cur.d > nearest[0].d && n_near == p.ef)) {
						go = 1;
						cur_id = cur.id;
					}
				}
				ctl[1] = go;
				ctl[2] = cur_id;
				ctl[3] = __builtin_bit_cast(uint32_t, nearest[0].d);
				ctl[4] = n_near;
			}
			__syncthreads();
			if (!ctl[1])
				break;
			const uint32_t cur_id = ctl[2];
			const float worst0 = __builtin_bit_cast(float, ctl[3]);
			const bool full0 = ctl[4] == p.ef;
			const uint32_t e0 = off0[cur_id], deg = off0[cur_id + 1] - e0;
			// unvisited neighbours, in adjacency order (:595-607)
			uint32_t n_list = 0;
			for (uint32_t i0 = 0; i0 < deg; i0 += 64) {
				const uint32_t i = i0 + lane;
				uint32_t nb = 0;
				bool fresh = false;
				if (i < deg) {
					nb = p.neighbours[e0 + i];
					fresh = visited[nb] != ep8;
					if (fresh)
						visited[nb] = ep8;
				}
				const unsigned long long mask = __builtin_amdgcn_ballot_w64(fresh);
				if (fresh)
					nlist[n_list + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = nb;
				n_list += (uint32_t)__builtin_popcountll(mask);
			}
			__syncthreads();
			// score them (:636-689 / :795-835): 4 rows per step (16 lanes each), 4 steps' rows requested
			// before the first is consumed -- a hop is a dependent chain of HBM round trips otherwise
			// (30 steps x ~1 us at M0 = 120)
			for (uint32_t i0 = 0; i0 < n_list; i0 += 16) {
				constexpr int U = 4;
				uint32_t nb[U];
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const uint32_t i = i0 + 4 * u + rg;
					nb[u] = nlist[i < n_list ? i : n_list - 1];
				}
				float d[U];
				if (COMPRESSED) {
					int b[U][NW];
#pragma unroll
					for (int u = 0; u < U; ++u) {
						const int* r = reinterpret_cast<const int*>(p.compressed + (size_t)nb[u] * D) + l * NW;
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
				} else {
					float r[U][DPL];
#pragma unroll
					for (int u = 0; u < U; ++u) {
						const float* src = p.vectors + (size_t)nb[u] * D + l;
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
				}
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const uint32_t i = i0 + 4 * u + rg;
					if (l == 0 && i < n_list)
						ndist[i] = d[u];
				}
			}
			distcomps += n_list;
			__syncthreads();
			// serial queue update in list order (lane 0)
			if (lane == 0) {
				for (uint32_t i = 0; i < n_list; ++i) {
					const float dn = ndist[i];
					if (full0 && !(dn < worst0))
						continue;  // cannot pass `d_next < nearest.top().first` later either
					if (n_near < p.ef || dn < nearest[0].d) {
						const md_pair e{dn, nlist[i]};
						if (n_cand >= p.cand_cap) {
							ctl[0] = 1;
						} else {
							heap_push<false>(candidates, n_cand, e);
						}
						heap_push<true>(nearest, n_near, e);
						if (n_near > p.ef)
							heap_pop<true>(nearest, n_near);
					}
				}
			}
			__syncthreads();
		}
		// ---- output: drain nearest, reverse, (re-score), truncate to k (:695-707, :845-848) -
		if (lane == 0) {
			const uint32_t cnt = n_near;
			for (uint32_t i = cnt; i-- > 0;) {
				const md_pair t = nearest[0];
				heap_pop<true>(nearest, n_near);
				nlist[i] = t.id;
				ndist[i] = t.d;
			}
			ctl[4] = cnt;
			if (ctl[0])
				atomicAdd(p.error, 1u);
		}
		__syncthreads();
		const uint32_t cnt = ctl[4];
		const uint32_t n_out = cnt < p.k ? cnt : p.k;
		if (COMPRESSED) {
			for (uint32_t i0 = 0; i0 < n_out; i0 += 4) {
				const uint32_t i = i0 + rg;
				const float d = dist_f32(nlist[i < n_out ? i : n_out - 1]);
				if (l == 0 && i < n_out)
					ndist[i] = d;
			}
			__syncthreads();
		}
		for (uint32_t i = lane; i < p.k; i += 64) {
			p.out_ids[(size_t)qi * p.k + i] = i < n_out ? (uint64_t)nlist[i] : ~0ull;
			p.out_dists[(size_t)qi * p.k + i] = i < n_out ? ndist[i] : __builtin_inff();
		}
		if (lane == 0 && p.out_distcomps)
			p.out_distcomps[qi] = distcomps;
		__syncthreads();
	}
	if (lane == 0)
		p.epochs[blockIdx.x] = epoch;
}

}  // namespace expann
