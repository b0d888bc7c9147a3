// host_common.hpp -- host-side helpers shared by the translation units of libexpann_hip
// (expann_hip.hip: the brute-force index; expann_graph.hip: graph path and quantiser builds).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/expann_hip.h"

namespace expann {

// scoped device allocation for the entry points that need per-call temporaries
struct DevBuf {
	void* p = nullptr;
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() {
		if (p)
			(void)hipFree(p);
	}
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
	template <typename T> T* as() const { return static_cast<T*>(p); }
};

// Device / pinned-host memory OWNED by a handle: frees itself when the handle goes away, converts
// to the raw pointer the kernels and HIP calls take (`hipMalloc(&h->d_x, n)` still works).
// Not copyable; `reset()` frees now.
template <typename T> struct DevPtr {
	T* p = nullptr;
	DevPtr() = default;
	DevPtr(const DevPtr&) = delete;
	DevPtr& operator=(const DevPtr&) = delete;
	~DevPtr() { reset(); }
	void reset() {
		if (p)
			(void)hipFree(p);
		p = nullptr;
	}
	T* get() const { return p; }
	template <typename U> U* as() const { return (U*)p; }  // the same bytes as another element type
	operator T*() const { return p; }
	T** operator&() { return &p; }
	explicit operator bool() const { return p != nullptr; }
};
template <typename T> struct PinPtr {
	T* p = nullptr;
	PinPtr() = default;
	PinPtr(const PinPtr&) = delete;
	PinPtr& operator=(const PinPtr&) = delete;
	~PinPtr() { reset(); }
	void reset() {
		if (p)
			(void)hipHostFree(p);
		p = nullptr;
	}
	T* get() const { return p; }
	template <typename U> U* as() const { return (U*)p; }
	operator T*() const { return p; }
	T** operator&() { return &p; }
	explicit operator bool() const { return p != nullptr; }
};
// a DevPtr that only ever grows: ensure(n) keeps the allocation when it is large enough.  Growing
// frees the old block: work still in flight that reads it must have drained first (the callers with
// deferred searches outstanding synchronise their stream before they grow a shared buffer).
template <typename T> struct GrowPtr : DevPtr<T> {
	size_t bytes = 0;
	hipError_t ensure(size_t need) {
		if (need <= bytes && this->p)
			return hipSuccess;
		this->reset();
		bytes = 0;
		const hipError_t e = hipMalloc((void**)&this->p, need ? need : 1);
		if (e == hipSuccess)
			bytes = need;
		return e;
	}
	void reset_all() {
		this->reset();
		bytes = 0;
	}
};

// what expann_*_last_error(NULL) reports: the failure of the last create call of this thread
extern thread_local std::string g_create_error;

int num_cus(int device);

}  // namespace expann

// every handle type has fail(code, message) -> code
#define HIP_TRY(h, expr)                                                                   \
	do {                                                                                   \
		hipError_t _e = (expr);                                                            \
		if (_e != hipSuccess)                                                              \
			return (h)->fail(EXPANN_ERR_HIP, std::string(#expr) + ": " +                   \
			                                       hipGetErrorString(_e));                 \
	} while (0)
