// graph_build.hpp -- GPU-assisted, batched construction of the reference's graph index
// (antitopo_engine::_store_vector / prune_edges, upstream src/antitopo_engine.h:263-465).
//
// The reference inserts one vector at a time on one core: search the graph so far
// (ef_construction candidates per layer), prune them to M / M0 edges with the "ortho" rule, add the
// reverse edges and re-prune every neighbour list that overflows -- ~10 ms per vector at M = 60,
// hours for SIFT1M, 99 % of it distance evaluations inside prune_edges.  Here the
// vectors of a BATCH are inserted together against the graph of the vectors before them:
//   build_search_kernel    one wavefront per new vector: greedy descent through the layers above
//                          its level, then the ef_construction best-first search of every layer it
//                          joins (the same walk and arithmetic as graph_search.hpp, on the mutable
//                          adjacency arrays), results = sorted candidate lists
//   build_prune_kernel     one workgroup per (vertex, layer) list: sort by (distance, id) and apply
//                          prune_edges' rule (:263-308) -- new vertices: their candidate lists;
//                          old vertices: their adjacency row after the batch appended to it
//   build_reverse_kernel   appends (distance, new vertex) to the row of every neighbour kept
//                          (:442-455), collecting the rows that outgrew M / M0 for the second prune
// Vertices of one batch do not see each other (they are linked through later batches' searches and
// reverse edges), and a row that receives several new edges is pruned once, not once per edge: the
// result is a valid antitopo graph of the same parameters, not the serial builder's graph edge for
// edge.  include/expann/antitopo_index.h keeps the serial restatement (format / parity path) and
// builds the first vertices, whose graph seeds the batches.
//
// Distances use the reference's 16-lane FMA order wherever they are STORED (edge lengths);
// prune_edges' vertex-to-vertex distances are only compared, and are evaluated per thread in the same
// lane order (16 partial sums per distance) so that serial and batched pruning agree on ties.
#pragma once
#include "graph_search.hpp"

namespace expann {

struct BuildGraph {
	const float* vec;          // [n][D]
	uint32_t n;
	const uint8_t* level;      // [n] level of every vertex (host draw, src/antitopo_engine.h:323)
	const int32_t* upper_idx;  // [n] row in the upper-layer arrays, -1 for level-0 vertices
	uint32_t* id0;             // layer 0: [n][stride0]
	float* d0;
	uint32_t* deg0;            // [n]
	uint32_t cap0, stride0;    // M0, M0 + slack for the batch's reverse edges
	uint32_t* idu;             // layers 1..: [(l-1) * U + upper_idx][strideu]
	float* du;
	uint32_t* degu;            // [(l-1) * U + upper_idx]
	uint32_t capu, strideu, U;

	__device__ inline uint32_t* ids(uint32_t l, uint32_t v) const {
		return l == 0 ? id0 + (size_t)v * stride0 : idu + ((size_t)(l - 1) * U + (uint32_t)upper_idx[v]) * strideu;
	}
	__device__ inline float* dists(uint32_t l, uint32_t v) const {
		return l == 0 ? d0 + (size_t)v * stride0 : du + ((size_t)(l - 1) * U + (uint32_t)upper_idx[v]) * strideu;
	}
	__device__ inline uint32_t* deg(uint32_t l, uint32_t v) const {
		return l == 0 ? deg0 + v : degu + ((size_t)(l - 1) * U + (uint32_t)upper_idx[v]);
	}
	__device__ inline uint32_t cap(uint32_t l) const { return l == 0 ? cap0 : capu; }
	__device__ inline uint32_t stride(uint32_t l) const { return l == 0 ? stride0 : strideu; }
};

// ---- search ------------------------------------------------------------------------------------
struct BuildSearchParams {
	BuildGraph g;
	uint32_t b0, b1;              // the batch: vertices [b0, b1)
	uint32_t max_layer, starting_vertex;
	uint32_t ef;                  // ef_construction
	uint32_t cand_cap, list_cap;  // LDS capacities: candidates heap, neighbour list of a hop
	uint32_t* vis_bits;           // [gridDim.x][vis_words] visited bitsets, all zero between searches
	uint32_t vis_words;           // a multiple of 256
	// results: list of (v, layer) = out + slot * ef entries, slot = v - b0 for layer 0, else
	// up_slot[v - b0] + layer - 1 behind the (b1 - b0) layer-0 slots
	const int32_t* up_slot;       // [b1 - b0] or -1
	md_pair* out;
	uint32_t* out_cnt;            // [slots]
	uint32_t* error;              // candidates heap overflow
};

template <int D>
__global__ __launch_bounds__(64) void build_search_kernel(BuildSearchParams p) {
	constexpr int DPL = D / 16;
	constexpr int U = graph_rows_f32<D>();  // rows in flight per 16-lane group
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	md_pair* nearest = reinterpret_cast<md_pair*>(smem_raw);             // [ef + 1]
	md_pair* candidates = nearest + (p.ef + 1);                          // [cand_cap + 1]
	uint32_t* nlist = reinterpret_cast<uint32_t*>(candidates + (p.cand_cap + 1));  // [list_cap]
	float* ndist = reinterpret_cast<float*>(nlist + p.list_cap);
	const int lane = threadIdx.x;
	const int l = lane & 15, rg = lane >> 4;
	// visited set of this workgroup: a bitset, all zero between searches (graph_search.hpp)
	uint32_t* vbits = p.vis_bits + (size_t)blockIdx.x * p.vis_words;
	uint32_t overflowed = 0;

	for (uint32_t v = p.b0 + blockIdx.x; v < p.b1; v += gridDim.x) {
		float q[DPL];
#pragma unroll
		for (int t = 0; t < DPL; ++t)
			q[t] = p.g.vec[(size_t)v * D + l + 16 * t];
		// exact fp32 squared L2 of the new vertex against UU rows per 16-lane group (reference order), all
		// rows requested before the first is consumed
		auto dist_rows = [&](auto u_tag, const uint32_t* rows, float* d) {
			constexpr int UU = decltype(u_tag)::value;
			float r[UU][DPL];
#pragma unroll
			for (int u = 0; u < UU; ++u) {
				const float* src = p.g.vec + (size_t)rows[u] * D + l;
#pragma unroll
				for (int t = 0; t < DPL; ++t)
					r[u][t] = src[16 * t];
			}
#pragma unroll
			for (int u = 0; u < UU; ++u) {
				float acc = 0.0f;
#pragma unroll
				for (int t = 0; t < DPL; ++t) {
					const float diff = q[t] - r[u][t];
					acc = __builtin_fmaf(diff, diff, acc);
				}
				d[u] = reduce16_ref_order(acc);
			}
		};
		auto dist_f32 = [&](uint32_t row) -> float {
			float d1[1];
			dist_rows(std::integral_constant<int, 1>{}, &row, d1);
			return d1[0];
		};
		const uint32_t lv = p.g.level[v];
		// greedy descent through the layers above the new vertex's level (:343-357, one entry point)
		uint32_t entry = p.starting_vertex;
		float ep_dist = dist_f32(entry);
		for (uint32_t layer = p.max_layer - 1; layer > lv && layer < p.max_layer; --layer) {
			bool changed = true;
			while (changed) {
				changed = false;
				const uint32_t* nb_ids = p.g.ids(layer, entry);
				const uint32_t deg = min(*p.g.deg(layer, entry), p.g.stride(layer));
				// first-improvement chain == first occurrence of the minimum, if it improves
				uint64_t best_key = ~0ull;
				uint32_t best_nb = 0;
				for (uint32_t i0 = 0; i0 < deg; i0 += 4 * U) {
					uint32_t nb[U];
					float d[U];
#pragma unroll
					for (int u = 0; u < U; ++u) {
						const uint32_t i = i0 + 4 * u + rg;
						nb[u] = nb_ids[i < deg ? i : deg - 1];
					}
					dist_rows(std::integral_constant<int, U>{}, nb, d);
#pragma unroll
					for (int u = 0; u < U; ++u) {
						const uint32_t i = i0 + 4 * u + rg;
						const uint64_t key = ((uint64_t)__builtin_bit_cast(uint32_t, d[u]) << 32) | i;
						if (i < deg && key < best_key) {
							best_key = key;
							best_nb = nb[u];
						}
					}
				}
				const uint64_t wmin = wave_min_key64(best_key);
				const float dmin = __builtin_bit_cast(float, (uint32_t)(wmin >> 32));
				if (wmin != ~0ull && dmin < ep_dist) {
					const unsigned long long who = __builtin_amdgcn_ballot_w64(best_key == wmin);
					entry = (uint32_t)__builtin_amdgcn_readlane((int)best_nb, __builtin_ctzll(who));
					ep_dist = dmin;
					changed = true;
				}
			}
		}
		// ef_construction search of every layer the vertex joins, top down (:364-387, ortho_count 1:
		// one seed, plain distances; the next layer starts at this layer's nearest).  The queues are
		// updated by the whole wave (graph_search.hpp: coop_push / coop_pop, libstdc++'s movement).
		const uint32_t top = lv < p.max_layer - 1 ? lv : p.max_layer - 1;
		for (uint32_t layer = top; layer <= top; --layer) {
			uint32_t n_near = 0, n_cand = 0;  // wave-uniform
			const float d_entry = dist_f32(entry);
			{
				const md_pair e{d_entry, entry};
				coop_push<false>(candidates, n_cand, e, lane);
				coop_push<true>(nearest, n_near, e, lane);
				if (lane == 0)
					atomicOr(&vbits[entry >> 5], 1u << (entry & 31));
			}
			wave_lds_sync();
			for (;;) {
				if (n_cand == 0)
					break;
				const md_pair cur = candidates[0];
				coop_pop<false>(candidates, n_cand, lane);
				const float worst0 = nearest[0].d;
				const bool full0 = n_near == p.ef;
				if (cur.d > worst0 && full0)
					break;
				const uint32_t* nb_ids = p.g.ids(layer, cur.id);
				const uint32_t deg = min(*p.g.deg(layer, cur.id), p.g.stride(layer));
				uint32_t n_list = 0;
				for (uint32_t i0 = 0; i0 < deg; i0 += 128) {  // two halves requested before either is consumed
					uint32_t nbv[2];
					bool fresh[2];
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const uint32_t i = i0 + 64u * h + lane;
						nbv[h] = i < deg ? nb_ids[i] : 0xFFFFFFFFu;
					}
#pragma unroll
					for (int h = 0; h < 2; ++h)
						fresh[h] = nbv[h] != 0xFFFFFFFFu &&
						           (atomicOr(&vbits[nbv[h] >> 5], 1u << (nbv[h] & 31)) & (1u << (nbv[h] & 31))) == 0;
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const unsigned long long mask = __builtin_amdgcn_ballot_w64(fresh[h]);
						if (fresh[h])
							nlist[n_list + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = nbv[h];
						n_list += (uint32_t)__builtin_popcountll(mask);
					}
				}
				wave_lds_sync();
				for (uint32_t i0 = 0; i0 < n_list; i0 += 4 * U) {
					uint32_t nbu[U];
					float d[U];
#pragma unroll
					for (int u = 0; u < U; ++u) {
						const uint32_t i = i0 + 4 * u + rg;
						nbu[u] = nlist[i < n_list ? i : n_list - 1];
					}
					dist_rows(std::integral_constant<int, U>{}, nbu, d);
#pragma unroll
					for (int u = 0; u < U; ++u) {
						const uint32_t i = i0 + 4 * u + rg;
						if (l == 0 && i < n_list)
							ndist[i] = d[u];
					}
				}
				wave_lds_sync();
				// pre-filter (all lanes), compacted in list order; then one wave-wide queue operation each
				uint32_t n_s = 0;
				for (uint32_t i0 = 0; i0 < n_list; i0 += 64) {
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
							coop_push<false>(candidates, n_cand, e, lane);
						coop_push<true>(nearest, n_near, e, lane);
						wave_lds_sync();
						if (n_near > p.ef)
							coop_pop<true>(nearest, n_near, lane);
					}
				}
				wave_lds_sync();
			}
			// drain (worst first), write ascending
			const uint32_t slot = layer == 0 ? v - p.b0 : (p.b1 - p.b0) + (uint32_t)p.up_slot[v - p.b0] + layer - 1;
			md_pair* out = p.out + (size_t)slot * p.ef;
			const uint32_t cnt = n_near;
			for (uint32_t i = cnt; i-- > 0;) {
				const md_pair t = nearest[0];
				coop_pop<true>(nearest, n_near, lane);
				if (lane == 0)
					out[i] = t;
				if (i == 0)
					entry = t.id;
			}
			if (lane == 0)
				p.out_cnt[slot] = cnt;
			// the visited set goes back to all-zero for the next search
			uint4* w = reinterpret_cast<uint4*>(vbits);
			for (uint32_t i = lane; i < p.vis_words / 4; i += 64)
				w[i] = make_uint4(0u, 0u, 0u, 0u);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the clear lands before the next search's test-and-sets)
			wave_lds_sync();
		}
	}
	if (lane == 0 && overflowed)
		atomicAdd(p.error, 1u);
}

// ---- prune ---------------------------------------------------------------------------------------
// prune_edges (src/antitopo_engine.h:263-308): candidates in (distance, id) order; repeat: take
// the candidate with the smallest score (first of equals), where
//   score(c) = d_c + sum over the kept edges r, in order, with dist(r, c) < d_c, of
//              ortho_factor * (d_c - dist(r, c)) + ortho_bias,
// and a candidate whose (prune_overflow + 1)-th such edge appears scores "pruned" for good; stop at
// `cap` edges or when only pruned candidates remain.  The scores are updated incrementally (one term
// per kept edge, in the order the reference's loop adds them).
struct PruneTask {
	uint32_t vertex, layer;
	int32_t src_slot;  // >= 0: candidates = search list `src_slot` (a new vertex); -1: the vertex's own row
};
struct BuildPruneParams {
	BuildGraph g;
	const PruneTask* tasks;       // or nullptr: tasks are the dirty rows
	const uint32_t* n_tasks;      // [1]
	const uint2* dirty;           // (vertex, layer) rows that outgrew their cap (build_reverse_kernel)
	const md_pair* lists;         // search results, `ef` entries per slot
	const uint32_t* list_cnt;
	uint32_t ef;
	float ortho_factor, ortho_bias;
	uint32_t prune_overflow;
};
constexpr int kPruneThreads = 256;
constexpr int kPruneMaxCand = 1024;  // candidates per list (ef_construction, or cap + slack)

template <int D>
__global__ __launch_bounds__(kPruneThreads) void build_prune_kernel(BuildPruneParams p) {
	constexpr int PER = kPruneMaxCand / kPruneThreads;
	__shared__ uint64_t keys[kPruneMaxCand];   // (ordered distance, id), sorted
	__shared__ float red_v[kPruneThreads / 64];
	__shared__ uint32_t red_i[kPruneThreads / 64];
	__shared__ uint32_t sel_s;
	__shared__ float row_s[D];
	const uint32_t tid = threadIdx.x;
	const uint32_t n_tasks = *p.n_tasks;
	for (uint32_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
		uint32_t vertex, layer;
		int32_t src;
		if (p.tasks) {
			vertex = p.tasks[t].vertex;
			layer = p.tasks[t].layer;
			src = p.tasks[t].src_slot;
		} else {
			vertex = p.dirty[t].x;
			layer = p.dirty[t].y;
			src = -1;
		}
		const uint32_t cap = p.g.cap(layer);
		uint32_t* row_ids = p.g.ids(layer, vertex);
		float* row_d = p.g.dists(layer, vertex);
		uint32_t* row_deg = p.g.deg(layer, vertex);
		uint32_t C;
		if (src >= 0) {
			C = p.list_cnt[src];
			const md_pair* lst = p.lists + (size_t)src * p.ef;
			for (uint32_t i = tid; i < kPruneMaxCand; i += kPruneThreads)
				keys[i] = i < C ? make_key(lst[i].d, lst[i].id) : kSentinelKey;
		} else {
			C = min(*row_deg, p.g.stride(layer));
			for (uint32_t i = tid; i < kPruneMaxCand; i += kPruneThreads)
				keys[i] = i < C ? make_key(row_d[i], row_ids[i]) : kSentinelKey;
		}
		__syncthreads();
		// bitonic sort of the keys (power of two >= C)
		uint32_t n2 = 2;
		while (n2 < C)
			n2 <<= 1;
		for (uint32_t size = 2; size <= n2; size <<= 1)
			for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
				for (uint32_t i = tid; i < (n2 >> 1); i += kPruneThreads) {
					const uint32_t lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
					const bool up = ((lo & size) == 0);
					const uint64_t a = keys[lo], b = keys[hi];
					if ((a > b) == up) {
						keys[lo] = b;
						keys[hi] = a;
					}
				}
				__syncthreads();
			}
		// per-thread candidates: index tid + j * 256 (a std::set of pairs: duplicates cannot occur)
		float basic[PER], res[PER];
		uint32_t cid[PER], len[PER];
		bool alive[PER];
#pragma unroll
		for (int j = 0; j < PER; ++j) {
			const uint32_t i = tid + j * kPruneThreads;
			alive[j] = i < C;
			const uint64_t k = keys[i < C ? i : 0];
			basic[j] = key_score(k);
			cid[j] = key_idx(k);
			res[j] = basic[j];
			len[j] = p.prune_overflow + 1;
		}
		uint32_t kept = 0;
		while (kept < cap) {
			// argmin of res over the alive candidates, first of equals
			float bv = __builtin_inff();
			uint32_t bi = 0xFFFFFFFFu;
#pragma unroll
			for (int j = 0; j < PER; ++j) {
				const uint32_t i = tid + j * kPruneThreads;
				if (alive[j] && (res[j] < bv || (res[j] == bv && i < bi))) {
					bv = res[j];
					bi = i;
				}
			}
			for (int off = 32; off > 0; off >>= 1) {
				const float ov = __shfl_xor(bv, off);
				const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off);
				if (ov < bv || (ov == bv && oi < bi)) {
					bv = ov;
					bi = oi;
				}
			}
			if ((tid & 63) == 0) {
				red_v[tid >> 6] = bv;
				red_i[tid >> 6] = bi;
			}
			__syncthreads();
			if (tid == 0) {
				for (int w = 1; w < kPruneThreads / 64; ++w)
					if (red_v[w] < bv || (red_v[w] == bv && red_i[w] < bi)) {
						bv = red_v[w];
						bi = red_i[w];
					}
				// nothing alive, or only pruned candidates left (score == prune_score, :303-304)
				sel_s = (bi == 0xFFFFFFFFu || bv == 3.402823466e+38f) ? 0xFFFFFFFFu : bi;
			}
			__syncthreads();
			const uint32_t s = sel_s;
			if (s == 0xFFFFFFFFu)
				break;
			const uint64_t ks = keys[s];
			const uint32_t sid = key_idx(ks);
			// the kept edge goes to the row; its vector to LDS
			if (tid == 0) {
				row_ids[kept] = sid;
				row_d[kept] = key_score(ks);
			}
			for (uint32_t i = tid; i < D; i += kPruneThreads)
				row_s[i] = p.g.vec[(size_t)sid * D + i];
#pragma unroll
			for (int j = 0; j < PER; ++j)
				if (tid + j * kPruneThreads == s)
					alive[j] = false;
			__syncthreads();
			++kept;
			if (kept == cap)
				break;
			// every candidate still alive: one more term of its score
#pragma unroll
			for (int j = 0; j < PER; ++j) {
				if (!alive[j] || res[j] == 3.402823466e+38f)
					continue;
				const float* rc = p.g.vec + (size_t)cid[j] * D;
				float acc[16];
#pragma unroll
				for (int ll = 0; ll < 16; ++ll)
					acc[ll] = 0.0f;
				for (int t0 = 0; t0 < D; t0 += 16) {
					const float4* r4 = reinterpret_cast<const float4*>(rc + t0);
					const float4 v0 = r4[0], v1 = r4[1], v2 = r4[2], v3 = r4[3];
					const float rv[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w,
					                      v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
					for (int ll = 0; ll < 16; ++ll) {
						const float diff = row_s[t0 + ll] - rv[ll];  // dist2(all_entries[prev], all_entries[c])
						acc[ll] = __builtin_fmaf(diff, diff, acc[ll]);
					}
				}
				float t8[8], t4[4];
#pragma unroll
				for (int ll = 0; ll < 8; ++ll)
					t8[ll] = acc[ll + 8] + acc[ll];
#pragma unroll
				for (int ll = 0; ll < 4; ++ll)
					t4[ll] = t8[ll + 4] + t8[ll];
				const float co = (t4[0] + t4[2]) + (t4[1] + t4[3]);
				if (co < basic[j]) {
					const float term = p.ortho_factor * (basic[j] - co);
					res[j] += term + p.ortho_bias;
					if (--len[j] == 0)
						res[j] = 3.402823466e+38f;
				}
			}
			__syncthreads();  // row_s is rewritten by the next round
		}
		__syncthreads();
		if (tid == 0)
			*row_deg = kept;
		__syncthreads();
	}
}

// ---- reverse edges ---------------------------------------------------------------------------------
struct BuildReverseParams {
	BuildGraph g;
	const PruneTask* tasks;   // the batch's (new vertex, layer) rows, just pruned
	uint32_t n_tasks;
	uint2* dirty;             // rows that outgrew their cap
	uint32_t* n_dirty;
	uint32_t dirty_cap;
	uint32_t* dropped;        // edges that found no slack slot left (statistics)
};
__global__ __launch_bounds__(kBlock) void build_reverse_kernel(BuildReverseParams p) {
	// one wave per (new vertex, layer) row, one lane per kept edge
	const uint32_t t = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (t >= p.n_tasks)
		return;
	const uint32_t v = p.tasks[t].vertex, layer = p.tasks[t].layer;
	const uint32_t deg = *p.g.deg(layer, v);
	const uint32_t* ids = p.g.ids(layer, v);
	const float* ds = p.g.dists(layer, v);
	const uint32_t cap = p.g.cap(layer), stride = p.g.stride(layer);
	for (uint32_t i = threadIdx.x & 63; i < deg; i += 64) {
		const uint32_t nb = ids[i];
		const uint32_t slot = atomicAdd(p.g.deg(layer, nb), 1u);
		if (slot < stride) {
			p.g.ids(layer, nb)[slot] = v;
			p.g.dists(layer, nb)[slot] = ds[i];
		} else {
			atomicAdd(p.dropped, 1u);
		}
		if (slot == cap) {  // the first edge beyond the cap: the row needs prune_edges (lazy, :269-272)
			const uint32_t k = atomicAdd(p.n_dirty, 1u);
			if (k < p.dirty_cap)
				p.dirty[k] = make_uint2(nb, layer);
		}
	}
}
// rows whose counter ran past the slack keep `stride` entries
__global__ __launch_bounds__(kBlock) void build_clamp_kernel(BuildGraph g, const uint2* dirty, const uint32_t* n_dirty) {
	const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
	if (t >= *n_dirty)
		return;
	uint32_t* d = g.deg(dirty[t].y, dirty[t].x);
	if (*d > g.stride(dirty[t].y))
		*d = g.stride(dirty[t].y);
}

}  // namespace expann
