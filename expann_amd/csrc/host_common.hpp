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
