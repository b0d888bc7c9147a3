#include <hip/hip_runtime.h>
#include <cstdio>
#include "../expann_amd/csrc/graph_search.hpp"
using namespace expann;
__global__ void k(int maxh) {
	__shared__ md_pair v[64];
	int lane = threadIdx.x;
	uint32_t n = 0;
	float ds[3] = {1.f, 1.f, 0.f};
	for (int i = 0; i < 3; ++i) {
		coop_push<true>(v, n, md_pair{ds[i], (uint32_t)i}, lane);
		wave_lds_sync();
	}
	if (lane == 0) printf("after pushes: (%g,%u) (%g,%u) (%g,%u)\n", v[0].d, v[0].id, v[1].d, v[1].id, v[2].d, v[2].id);
	wave_lds_sync();
	// manual pop with prints
	{
		const md_pair value = v[n - 1];
		const uint32_t len = n - 1;
		if (lane == 0) v[n - 1] = v[0];
		const uint32_t half = (len - 1) / 2;
		uint32_t cur = 0, depth = 0, my_c = 0, my_cn = 0;
		auto step_to = [&](uint32_t nxt) { ++depth; if ((uint32_t)lane == depth) my_c = nxt; if ((uint32_t)lane + 1 == depth) my_cn = nxt; cur = nxt; };
		if ((len & 1) == 0 && cur == (len - 2) / 2) step_to(2 * cur + 1);
		const bool on_chain = (uint32_t)lane < depth;
		const md_pair z = v[on_chain ? my_cn : 0];
		const unsigned long long rises = __builtin_amdgcn_ballot_w64(on_chain && md_less<true>(z, value));
		const unsigned long long stay = ~rises & ((1ull << depth) - 1ull);
		const uint32_t settle = stay ? 64u - (uint32_t)__builtin_clzll(stay) : 0u;
		if (lane < 2) printf("lane %d: len %u half %u depth %u my_c %u my_cn %u z (%g,%u) value (%g,%u) rises %llx stay %llx settle %u\n", lane, len, half, depth, my_c, my_cn, z.d, z.id, value.d, value.id, rises, stay, settle);
	}
	wave_lds_sync();
	coop_pop<true>(v, n, lane);
	if (lane == 0) printf("after pop: n %u (%g,%u) (%g,%u) (%g,%u)\n", n, v[0].d, v[0].id, v[1].d, v[1].id, v[2].d, v[2].id);
}
int main() { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 1); hipDeviceSynchronize(); return 0; }
