#include <hip/hip_runtime.h>
#include <cstdio>
#include "../expann_amd/csrc/graph_search.hpp"
using namespace expann;
__global__ void k2(md_pair* out, uint32_t n_in) {
	__shared__ md_pair v[64];
	int lane = threadIdx.x;
	uint32_t n = n_in;
	if (lane < 3) v[lane] = out[lane];
	wave_lds_sync();
	coop_pop<true>(v, n, lane);
	if (lane < 3) out[lane] = v[lane];
}
int main() {
	md_pair h[3] = {{1.f, 0}, {1.f, 1}, {0.f, 2}}, *d;
	hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, 3u); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	printf("(%g,%u) (%g,%u) (%g,%u)\n", h[0].d, h[0].id, h[1].d, h[1].id, h[2].d, h[2].id);
	return 0;
}
