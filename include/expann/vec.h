// vec.h -- minimal host-side stand-in for the reference's vec<T> (upstream src/vec.h:29-181),
// which wraps an Eigen vector.  Only the members the hot-path boundary touches are mirrored:
// construction from std::vector<T>, data()/size()/dim()/at()/operator[]/set_dim, and the free
// functions dist2 / dist / dot used by the harness for its top-1 distance statistics
// (src/basic_bench.h:91-96).  The storage is a dense contiguous array, as in the reference
// (Eigen::Matrix<T,DIM,1>, src/vec.h:17-23), so a std::vector<vec<T>> row is d*sizeof(T) bytes.
#pragma once

#include <cmath>
#include <cstddef>
#include <vector>

template <typename T> class vec {
	std::vector<T> internal;

public:
	vec() = default;
	explicit vec(size_t dim) : internal(dim) {}
	vec(const std::vector<T>& v) : internal(v) {}
	vec(const T* p, size_t dim) : internal(p, p + dim) {}
	T* data() { return internal.data(); }
	const T* data() const { return internal.data(); }
	void set_dim(size_t dim) { internal.resize(dim); }
	size_t size() const { return internal.size(); }
	size_t dim() const { return internal.size(); }
	T& operator[](size_t i) { return internal[i]; }
	const T& at(size_t i) const { return internal[i]; }
	std::vector<T> to_vector() const { return internal; }

	friend T dist2(const vec<T>& a, const vec<T>& b) {
		T s = 0;
		for (size_t i = 0; i < a.size(); ++i) {
			T d = a.at(i) - b.at(i);
			s += d * d;
		}
		return s;
	}
	friend T dist(const vec<T>& a, const vec<T>& b) { return std::sqrt(dist2(a, b)); }
	friend T dot(const vec<T>& a, const vec<T>& b) {
		T s = 0;
		for (size_t i = 0; i < a.size(); ++i)
			s += a.at(i) * b.at(i);
		return s;
	}
};
