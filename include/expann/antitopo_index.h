// antitopo_index.h -- host-side graph index of the reference's `antitopo_engine` (an HNSW
// variant): the data structure the GPU traversal consumes, a CPU builder, and the reference's
// on-disk index format.
//
// Scope: graph CONSTRUCTION is not on the scored hot path -- the reference builds
// serially on one core and hides the cost behind index files (src/bench_runner.h:149-162,
// src/antitopo_engine.h:137-155) -- so it stays on the host here too.  What this header provides:
//   * antitopo_index::insert()      restates _store_vector (src/antitopo_engine.h:310-465) and
//                                   prune_edges (:263-308) incl. the "ortho" entry points and the
//                                   mt19937(0) level draw (:159,:323)
//   * write_index()/read_index()    the reference's binary layout (:932-991 / :994-1074), so an
//                                   index built by an unmodified expANN can be loaded and vice versa
//   * flatten()                     CSR arrays for the device (expann_graph_create)
// Distances during construction use the reference's 16-lane FMA order (src/distance.h:86-111 via
// src/antitopo_engine.h:25-37), written as plain loops.  Floating-point contraction of the
// "ortho" score expression is compiler-dependent in the reference; this code does not contract.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <limits>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace expann {

struct antitopo_config {  // src/antitopo_engine.h:72-101
	size_t M = 16, M0 = 32, ef_search_mult = 1, ef_construction = 100, ortho_count = 1;
	float ortho_factor = 0.5f, ortho_bias = 0.0f;
	size_t prune_overflow = 0;
	bool use_compression = false, use_largest_direction_filtering = false;
};

// std::mt19937 + std::uniform_real_distribution<double>(0,1) as libstdc++ evaluates them
// (generate_canonical<double,53> = two 32-bit draws), restated so the level draw does not depend
// on which standard library compiles this header.
class mt19937_ref {
	uint32_t mt[624];
	int idx;

public:
	explicit mt19937_ref(uint32_t seed) {
		mt[0] = seed;
		for (int i = 1; i < 624; ++i)
			mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
		idx = 624;
	}
	uint32_t next() {
		if (idx >= 624) {
			for (int i = 0; i < 624; ++i) {
				uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
				mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
			}
			idx = 0;
		}
		uint32_t y = mt[idx++];
		y ^= y >> 11;
		y ^= (y << 7) & 0x9d2c5680u;
		y ^= (y << 15) & 0xefc60000u;
		y ^= y >> 18;
		return y;
	}
	double uniform01() {
		const double lo = (double)next();
		const double hi = (double)next();
		double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
		if (r >= 1.0)
			r = std::nextafter(1.0, 0.0);
		return r;
	}
};

// fp32 squared L2 in the lane order of src/distance.h:86-111 / :136-147
inline float dist2_ref_order(const float* a, const float* b, size_t d) {
	float acc[16];
	for (int l = 0; l < 16; ++l)
		acc[l] = 0.0f;
	for (size_t i = 0; i < d; i += 16)
		for (int l = 0; l < 16; ++l) {
			const float diff = a[i + l] - b[i + l];
			acc[l] = std::fmaf(diff, diff, acc[l]);
		}
	float t8[8], t4[4];
	for (int l = 0; l < 8; ++l)
		t8[l] = acc[l + 8] + acc[l];
	for (int l = 0; l < 4; ++l)
		t4[l] = t8[l + 4] + t8[l];
	const float t20 = t4[0] + t4[2], t21 = t4[1] + t4[3];
	return t20 + t21;
}

// Binary heap with exactly libstdc++'s std::push_heap / std::pop_heap / std::make_heap element
// movement, so that ties between equal keys come out in the same order as from the reference's
// std::priority_queue (whose comparators look at .first only, src/antitopo_engine.h:540-545).
template <typename E, typename Less> struct std_heap {
	std::vector<E> v;
	Less less;
	explicit std_heap(Less l) : less(l) {}
	size_t size() const { return v.size(); }
	bool empty() const { return v.empty(); }
	const E& top() const { return v.front(); }
	void push_up(size_t hole, size_t top_index, E value) {
		while (hole > top_index) {
			const size_t parent = (hole - 1) / 2;
			if (!less(v[parent], value))
				break;
			v[hole] = v[parent];
			hole = parent;
		}
		v[hole] = value;
	}
	void adjust(size_t hole, size_t len, E value) {
		const size_t top_index = hole;
		size_t child = hole;
		while (len > 1 && child < (len - 1) / 2) {
			child = 2 * (child + 1);
			if (less(v[child], v[child - 1]))
				--child;
			v[hole] = v[child];
			hole = child;
		}
		if ((len & 1) == 0 && len >= 2 && child == (len - 2) / 2) {
			child = 2 * (child + 1);
			v[hole] = v[child - 1];
			hole = child - 1;
		}
		push_up(hole, top_index, value);
	}
	void push(const E& e) {
		v.push_back(e);
		push_up(v.size() - 1, 0, e);
	}
	void pop() {
		const size_t n = v.size();
		if (n > 1) {
			E value = v[n - 1];
			v[n - 1] = v[0];
			adjust(0, n - 1, value);
		}
		v.pop_back();
	}
	void make() {
		const size_t len = v.size();
		if (len < 2)
			return;
		for (size_t parent = (len - 2) / 2;; --parent) {
			E value = v[parent];
			adjust(parent, len, value);
			if (parent == 0)
				break;
		}
	}
};

struct antitopo_index {
	using edge = std::pair<float, size_t>;
	antitopo_config conf;
	size_t dim = 0;
	size_t starting_vertex = 0, max_layer = 0;
	std::vector<float> vectors;                                        // [n][dim] (all_entries)
	std::vector<std::vector<std::vector<edge>>> hadj_flat_with_lengths;  // vertex -> layer -> edges
	std::vector<std::vector<std::vector<size_t>>> hadj_flat;             // vertex -> layer -> ids
	mt19937_ref gen{0};  // src/antitopo_engine.h:159
	std::vector<char> visited;
	std::vector<size_t> visited_recent;
	size_t num_distcomps = 0;
	// the engine's sticky ef_search (src/antitopo_engine.h:189-195, :858-859): only serialised
	// (:942-949), never loaded (":1000 search-time param, don't load")
	bool has_ef_search = false;
	size_t ef_search = 0;

	antitopo_index() = default;
	antitopo_index(size_t _dim, antitopo_config c) : conf(c), dim(_dim) {}

	size_t size() const { return dim ? vectors.size() / dim : 0; }
	const float* row(size_t i) const { return vectors.data() + i * dim; }
	float d2(const float* a, const float* b) const { return dist2_ref_order(a, b, dim); }

	// ---- src/antitopo_engine.h:263-308 ------------------------------------------------
	void update_edges(size_t layer, size_t from) {
		auto& ids = hadj_flat[from][layer];
		ids.clear();
		for (auto& e : hadj_flat_with_lengths[from][layer])
			ids.push_back(e.second);
	}
	void add_new_edges(size_t layer, size_t from) {
		auto& ids = hadj_flat[from][layer];
		auto& el = hadj_flat_with_lengths[from][layer];
		for (size_t i = ids.size(); i < el.size(); ++i)
			ids.push_back(el[i].second);
	}
	void prune_edges(size_t layer, size_t from, bool lazy) {
		auto& to = hadj_flat_with_lengths[from][layer];
		const size_t edge_count_mult = layer == 0 ? conf.M0 : conf.M;
		if (lazy && to.size() <= edge_count_mult) {
			add_new_edges(layer, from);
			return;
		}
		std::sort(to.begin(), to.end());
		std::set<edge> candidates(to.begin(), to.end());
		std::vector<edge> ret;
		const float prune_score = std::numeric_limits<float>::max();
		auto score = [&](const edge& e) -> float {
			const float basic = e.first;
			float res = basic;
			size_t leniency = conf.prune_overflow + 1;
			for (auto& pr : ret) {
				const float co = d2(row(pr.second), row(e.second));
				if (co < basic) {
					const float term = conf.ortho_factor * (basic - co);
					res += term + conf.ortho_bias;
					if (--leniency == 0)
						return prune_score;
				}
			}
			return res;
		};
		while (ret.size() < edge_count_mult && !candidates.empty()) {
			auto best = candidates.begin();
			float best_s = score(*best);
			for (auto it = std::next(candidates.begin()); it != candidates.end(); ++it) {
				const float s = score(*it);
				if (s < best_s) {
					best = it;
					best_s = s;
				}
			}
			if (best_s == prune_score)
				break;
			ret.push_back(*best);
			candidates.erase(best);
		}
		to = ret;
		update_edges(layer, from);
	}

	// ---- src/antitopo_engine.h:495-708 (use_compressed = false) -----------------------
	// score with the "ortho" penalty against ortho_points when use_ortho, plain dist2 otherwise
	std::vector<edge> query_k_at_layer(const float* q, size_t layer,
	                                   const std::vector<size_t>& entry_points, size_t k,
	                                   const std::vector<size_t>& ortho_points, bool use_ortho) {
		auto score = [&](size_t idx) -> float {
			if (use_ortho) {
				const float basic = d2(row(idx), q);
				float res = basic;
				for (size_t prev : ortho_points) {
					const float co = d2(row(prev), row(idx));
					if (co < basic) {
						const float term = conf.ortho_factor * (basic - co);
						res += term + conf.ortho_bias;
					}
				}
				return res;
			}
			++num_distcomps;
			return d2(q, row(idx));
		};
		auto worst_elem = [](const edge& a, const edge& b) { return a.first < b.first; };
		auto best_elem = [](const edge& a, const edge& b) { return a.first > b.first; };
		std_heap<edge, decltype(best_elem)> candidates(best_elem);
		std_heap<edge, decltype(worst_elem)> nearest(worst_elem);
		for (size_t ep : entry_points) {
			const edge e(score(ep), ep);
			candidates.v.push_back(e);
			nearest.v.push_back(e);
		}
		candidates.make();
		nearest.make();
		while (nearest.size() > k)
			nearest.pop();
		for (size_t ep : entry_points) {
			visited[ep] = 1;
			visited_recent.push_back(ep);
		}
		std::vector<size_t> neighbour_list;
		while (!candidates.empty()) {
			const edge cur = candidates.top();
			candidates.pop();
			if (cur.first > nearest.top().first && nearest.size() == k)
				break;
			neighbour_list.clear();
			for (size_t nb : hadj_flat[cur.second][layer])
				if (!visited[nb]) {
					neighbour_list.push_back(nb);
					visited[nb] = 1;
					visited_recent.push_back(nb);
				}
			for (size_t next : neighbour_list) {
				const float dn = score(next);
				if (nearest.size() < k || dn < nearest.top().first) {
					candidates.push(edge(dn, next));
					nearest.push(edge(dn, next));
					if (nearest.size() > k)
						nearest.pop();
				}
			}
		}
		for (size_t v : visited_recent)
			visited[v] = 0;
		visited_recent.clear();
		std::vector<edge> ret;
		while (!nearest.empty()) {
			ret.push_back(nearest.top());
			nearest.pop();
		}
		std::reverse(ret.begin(), ret.end());
		return ret;
	}

	// ---- src/antitopo_engine.h:310-465 ------------------------------------------------
	// the level draw of _store_vector (:323), one per vertex in insertion order
	size_t draw_level() { return (size_t)std::floor(-std::log(gen.uniform01()) * 1 / std::log(double(conf.M))); }
	void insert(const float* v) { insert_with_level(v, draw_level()); }
	void insert_with_level(const float* v, const size_t new_max_layer) {
		if (dim == 0)
			throw std::runtime_error("antitopo_index: dim not set");
		const size_t v_index = size();
		vectors.insert(vectors.end(), v, v + dim);
		visited.push_back(0);
		const float* q = row(v_index);
		hadj_flat_with_lengths.emplace_back(new_max_layer + 1);
		std::vector<std::vector<edge>> kNN_per_layer;
		if (size() > 1) {
			std::vector<size_t> cur;
			{
				std::vector<size_t> entry_points;
				for (size_t i = 0; i < conf.ortho_count; ++i) {
					size_t entry_point = starting_vertex;
					auto score = [&](size_t idx) -> float {
						const float basic = d2(row(idx), q);
						float res = basic;
						for (size_t prev : entry_points) {
							const float co = d2(row(prev), row(idx));
							if (co < basic) {
								const float term = conf.ortho_factor * (basic - co);
								res += term + conf.ortho_bias;
							}
						}
						return res;
					};
					float ep_dist = score(entry_point);
					for (size_t layer = max_layer - 1; layer > new_max_layer; --layer) {
						bool changed = true;
						while (changed) {
							changed = false;
							const std::vector<size_t>& nbrs = hadj_flat[entry_point][layer];
							for (size_t nb : nbrs) {  // the list bound when the loop starts (range-for)
								const float nd = score(nb);
								if (nd < ep_dist) {
									entry_point = nb;
									ep_dist = nd;
									changed = true;
								}
							}
						}
					}
					if (std::find(entry_points.begin(), entry_points.end(), entry_point) ==
					    entry_points.end())
						entry_points.push_back(entry_point);
				}
				cur = entry_points;
			}
			for (int layer = (int)std::min(new_max_layer, max_layer - 1); layer >= 0; --layer) {
				std::vector<std::vector<edge>> result_lists;
				std::vector<size_t> new_cur;
				std::vector<size_t> seeds = cur;
				std::set<size_t> seeds_set(seeds.begin(), seeds.end());
				for (size_t i = 0; i < conf.ortho_count; ++i) {
					result_lists.push_back(
					    query_k_at_layer(q, (size_t)layer, seeds, conf.ef_construction, new_cur, true));
					for (auto& e : result_lists.back())
						if (!seeds_set.count(e.second)) {
							seeds.push_back(e.second);
							seeds_set.insert(e.second);
						}
					const size_t cand = result_lists.back()[0].second;
					if (std::find(new_cur.begin(), new_cur.end(), cand) == new_cur.end())
						new_cur.push_back(cand);
				}
				std::set<edge> combined;
				for (auto& rl : result_lists)
					for (auto& e : rl)
						combined.insert(e);
				kNN_per_layer.emplace_back(combined.begin(), combined.end());
				cur = new_cur;
			}
			std::reverse(kNN_per_layer.begin(), kNN_per_layer.end());
		}
		hadj_flat.emplace_back(new_max_layer + 1);
		for (size_t layer = 0; layer < std::min(max_layer, new_max_layer + 1); ++layer) {
			hadj_flat_with_lengths[v_index][layer] = kNN_per_layer[layer];
			prune_edges(layer, v_index, false);
			// (copy: prune_edges of the neighbour may reallocate other vertices' lists only,
			// but iterating a copy keeps this loop independent of that detail)
			const std::vector<edge> mine = hadj_flat_with_lengths[v_index][layer];
			for (auto& md : mine) {
				bool edge_exists = false;
				for (auto& other : hadj_flat_with_lengths[md.second][layer])
					if (other.second == v_index) {
						edge_exists = true;
						break;
					}
				if (!edge_exists) {
					hadj_flat_with_lengths[md.second][layer].emplace_back(md.first, v_index);
					prune_edges(layer, md.second, true);
				}
			}
		}
		while (new_max_layer >= max_layer) {
			++max_layer;
			starting_vertex = v_index;
		}
	}

	// ---- the reference's index file (src/antitopo_engine.h:932-991 / :994-1074) -----------
	void write_index(const std::string& path) const {
		std::ofstream out(path, std::ios::binary);
		if (!out)
			throw std::runtime_error("cannot write " + path);
		auto W = [&](const void* p, size_t n) { out.write(reinterpret_cast<const char*>(p), (std::streamsize)n); };
		const uint64_t sv = starting_vertex, M = conf.M, M0 = conf.M0, efm = conf.ef_search_mult;
		W(&sv, 8); W(&M, 8); W(&M0, 8); W(&efm, 8);
		const uint8_t has_ef = has_ef_search ? 1 : 0;
		W(&has_ef, 1);
		if (has_ef) {
			const uint64_t efs = ef_search;
			W(&efs, 8);
		}
		const uint64_t efc = conf.ef_construction, oc = conf.ortho_count, po = conf.prune_overflow,
		               ml = max_layer;
		W(&efc, 8); W(&oc, 8); W(&conf.ortho_factor, 4); W(&conf.ortho_bias, 4); W(&po, 8);
		const uint8_t uc = conf.use_compression, ul = conf.use_largest_direction_filtering;
		W(&uc, 1); W(&ul, 1); W(&ml, 8);
		const uint64_t n = size(), dd = dim;
		W(&n, 8);
		for (size_t i = 0; i < n; ++i) {
			W(&dd, 8);
			W(row(i), dim * 4);
		}
		W(&n, 8);
		for (auto& vert : hadj_flat_with_lengths) {
			const uint64_t nl = vert.size();
			W(&nl, 8);
			for (auto& layer : vert) {
				const uint64_t ne = layer.size();
				W(&ne, 8);
				for (auto& e : layer) {
					const uint64_t id = e.second;
					W(&e.first, 4);
					W(&id, 8);
				}
			}
		}
	}
	void read_index(const std::string& path) {
		std::ifstream in(path, std::ios::binary);
		if (!in)
			throw std::runtime_error("cannot read " + path);
		auto R = [&](void* p, size_t n) {
			in.read(reinterpret_cast<char*>(p), (std::streamsize)n);
			if (!in)
				throw std::runtime_error("truncated index file " + path);
		};
		uint64_t sv, M, M0, efm, tmp;
		R(&sv, 8); R(&M, 8); R(&M0, 8); R(&efm, 8);
		uint8_t has_ef;
		R(&has_ef, 1);
		if (has_ef)
			R(&tmp, 8);
		uint64_t efc, oc, po, ml;
		uint8_t uc, ul;
		R(&efc, 8); R(&oc, 8); R(&conf.ortho_factor, 4); R(&conf.ortho_bias, 4); R(&po, 8);
		R(&uc, 1); R(&ul, 1); R(&ml, 8);
		starting_vertex = sv; conf.M = M; conf.M0 = M0;  // (ef_search_mult / use_compression:
		conf.ef_construction = efc; conf.ortho_count = oc; conf.prune_overflow = po;  // search-time,
		conf.use_largest_direction_filtering = ul; max_layer = ml;                     // not loaded)
		uint64_t n;
		R(&n, 8);
		vectors.clear();
		dim = 0;
		for (uint64_t i = 0; i < n; ++i) {
			uint64_t len;
			R(&len, 8);
			if (i == 0) {
				dim = len;
				vectors.resize(n * dim);
			} else if (len != dim) {
				throw std::runtime_error("index file with rows of different length");
			}
			R(vectors.data() + i * dim, dim * 4);
		}
		uint64_t nv;
		R(&nv, 8);
		hadj_flat_with_lengths.assign(nv, {});
		hadj_flat.assign(nv, {});
		for (uint64_t v = 0; v < nv; ++v) {
			uint64_t nl;
			R(&nl, 8);
			hadj_flat_with_lengths[v].resize(nl);
			hadj_flat[v].resize(nl);
			for (uint64_t l = 0; l < nl; ++l) {
				uint64_t ne;
				R(&ne, 8);
				auto& layer = hadj_flat_with_lengths[v][l];
				layer.resize(ne);
				for (auto& e : layer) {
					uint64_t id;
					R(&e.first, 4);
					R(&id, 8);
					e.second = id;
				}
				update_edges(l, v);
			}
		}
		visited.assign(n, 0);
	}

	// ---- fixed-stride adjacency arrays, the batched GPU builder's view (expann_graph_build_batched) --
	struct strided_graph {
		size_t n = 0, U = 0, n_upper_layers = 0, stride0 = 0, strideu = 0;
		std::vector<uint8_t> levels;      // [n]
		std::vector<int32_t> upper_idx;   // [n] row in the upper arrays or -1
		std::vector<uint32_t> ids0, deg0, idsu, degu;
		std::vector<float> d0, du;
	};
	// rows of the vertices built so far + empty rows for `new_levels.size()` vertices to come
	strided_graph to_strided(const std::vector<uint8_t>& new_levels, size_t slack) const {
		strided_graph g;
		const size_t built = size();
		g.n = built + new_levels.size();
		g.levels.resize(g.n);
		for (size_t v = 0; v < built; ++v)
			g.levels[v] = (uint8_t)(hadj_flat_with_lengths[v].size() - 1);
		std::copy(new_levels.begin(), new_levels.end(), g.levels.begin() + built);
		size_t lmax = max_layer ? max_layer - 1 : 0;
		g.upper_idx.assign(g.n, -1);
		for (size_t v = 0; v < g.n; ++v) {
			lmax = std::max<size_t>(lmax, g.levels[v]);
			if (g.levels[v] >= 1)
				g.upper_idx[v] = (int32_t)g.U++;
		}
		g.n_upper_layers = lmax;
		g.stride0 = conf.M0 + slack;
		g.strideu = conf.M + slack;
		g.ids0.assign(g.n * g.stride0, 0);
		g.d0.assign(g.n * g.stride0, 0.0f);
		g.deg0.assign(g.n, 0);
		g.idsu.assign(g.U * g.n_upper_layers * g.strideu, 0);
		g.du.assign(g.U * g.n_upper_layers * g.strideu, 0.0f);
		g.degu.assign(g.U * g.n_upper_layers, 0);
		for (size_t v = 0; v < built; ++v)
			for (size_t l = 0; l < hadj_flat_with_lengths[v].size(); ++l) {
				const auto& el = hadj_flat_with_lengths[v][l];
				const size_t st = l == 0 ? g.stride0 : g.strideu;
				if (el.size() > st)
					throw std::runtime_error("antitopo_index: adjacency row longer than its stride");
				uint32_t* ids = l == 0 ? &g.ids0[v * st] : &g.idsu[((l - 1) * g.U + (size_t)g.upper_idx[v]) * st];
				float* ds = l == 0 ? &g.d0[v * st] : &g.du[((l - 1) * g.U + (size_t)g.upper_idx[v]) * st];
				for (size_t i = 0; i < el.size(); ++i) {
					ids[i] = (uint32_t)el[i].second;
					ds[i] = el[i].first;
				}
				(l == 0 ? g.deg0[v] : g.degu[(l - 1) * g.U + (size_t)g.upper_idx[v]]) = (uint32_t)el.size();
			}
		return g;
	}
	// adopt the arrays the batched builder filled (all n vertices; `rows` = the vectors of the new ones)
	void from_strided(const strided_graph& g, const float* new_rows, size_t new_max_layer, size_t new_start) {
		const size_t built = size();
		vectors.insert(vectors.end(), new_rows, new_rows + (g.n - built) * dim);
		hadj_flat_with_lengths.assign(g.n, {});
		hadj_flat.assign(g.n, {});
		for (size_t v = 0; v < g.n; ++v) {
			const size_t nl = (size_t)g.levels[v] + 1;
			hadj_flat_with_lengths[v].resize(nl);
			hadj_flat[v].resize(nl);
			for (size_t l = 0; l < nl; ++l) {
				const size_t st = l == 0 ? g.stride0 : g.strideu;
				const uint32_t* ids = l == 0 ? &g.ids0[v * st] : &g.idsu[((l - 1) * g.U + (size_t)g.upper_idx[v]) * st];
				const float* ds = l == 0 ? &g.d0[v * st] : &g.du[((l - 1) * g.U + (size_t)g.upper_idx[v]) * st];
				const uint32_t deg = l == 0 ? g.deg0[v] : g.degu[(l - 1) * g.U + (size_t)g.upper_idx[v]];
				auto& el = hadj_flat_with_lengths[v][l];
				el.resize(deg);
				for (uint32_t i = 0; i < deg; ++i)
					el[i] = edge(ds[i], (size_t)ids[i]);
				update_edges(l, v);
			}
		}
		visited.assign(g.n, 0);
		max_layer = new_max_layer;
		starting_vertex = new_start;
	}

	// ---- CSR for the device: layer-major offsets; ids are 32-bit row numbers ---------------
	struct flat_graph {
		uint32_t n = 0, n_layers = 0, starting_vertex = 0;
		std::vector<uint64_t> layer_offsets;  // [n_layers][n + 1] into `neighbours`
		std::vector<uint32_t> neighbours;
	};
	flat_graph flatten() const {
		flat_graph g;
		g.n = (uint32_t)size();
		g.n_layers = (uint32_t)max_layer;
		g.starting_vertex = (uint32_t)starting_vertex;
		g.layer_offsets.assign((size_t)g.n_layers * (g.n + 1), 0);
		for (uint32_t l = 0; l < g.n_layers; ++l) {
			uint64_t* off = g.layer_offsets.data() + (size_t)l * (g.n + 1);
			for (uint32_t v = 0; v < g.n; ++v) {
				off[v] = g.neighbours.size();
				if (l < hadj_flat[v].size())
					for (size_t nb : hadj_flat[v][l])
						g.neighbours.push_back((uint32_t)nb);
			}
			off[g.n] = g.neighbours.size();
		}
		return g;
	}
};

}  // namespace expann
