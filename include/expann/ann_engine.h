// ann_engine.h -- host-side mirror of the reference's engine interface.
//
// Same public surface as upstream src/ann_engine.h:10-29 (the CRTP base every engine
// derives from: name / param_list / store_vector / build / query_k forwarding to the
// derived class's _name / _param_list / _store_vector / _build / _query_k, plus
// param_list_t and the add_param helpers), so that include/expann/gpu_brute_force_engine.h
// compiles both against this mirror and, dropped into the reference tree, against the
// reference's own header.  Written from the interface description, not copied.
#pragma once

#include <map>
#include <string>
#include <vector>

#include "vec.h"

using param_list_t = std::map<std::string, std::string>;

#ifndef add_param
#define add_param(pl, p) (pl)[#p] = std::to_string(p)
#define add_param_str(pl, p) (pl)[#p] = (p)
#define add_sub_param(pl, prefix_name, pname, p) (pl)[std::string(prefix_name) + (pname)] = (p)
#endif

template <typename T, class Derived> class ann_engine {
	Derived& self() { return *static_cast<Derived*>(this); }

public:
	ann_engine() = default;
	std::string name() { return self()._name(); }
	param_list_t param_list() { return self()._param_list(); }
	void store_vector(const vec<T>& v) { self()._store_vector(v); }
	void build() { self()._build(); }
	const std::vector<size_t> query_k(const vec<T>& v, size_t k) { return self()._query_k(v, k); }
};
