// quantize.hpp -- the reference's quantiser builds on the device.
//   quantize_simple_u8: quantizer_simple<uint8_t>::build, src/quantizer.h:132-141 -- a plain
//     T(float) cast per element, no scaling (meaningful for data already in [0,255]).
//   quantize_ranged_q8: quantizer_ranged_q8::build, src/quantizer.h:213-232 with convert_single
//     :196-200 -- global min/max, scale = 128/(max-min), offset = -scale*min, then
//     clamp(round(scale*x + offset), 0, 127).  The reference starts max at FLT_MIN (:217); kept.
#pragma once
#include "common.hpp"

namespace expann {

__global__ __launch_bounds__(kBlock) void quantize_simple_u8_kernel(const float* in, size_t n_values,
                                                                    uint8_t* out) {
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if (i < n_values)
		out[i] = (uint8_t)(uint32_t)in[i];  // defined for 0 <= x < 256 (UB in C++ beyond that)
}

// minmax[0] = ordered(min), minmax[1] = ordered(max); initialised by the host to
// ordered(FLT_MAX) / ordered(FLT_MIN) as the reference initialises min_val / max_val
__global__ __launch_bounds__(kBlock) void minmax_f32_kernel(const float* in, size_t n_values,
                                                            uint32_t* minmax) {
	float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_values;
	     i += (size_t)gridDim.x * kBlock) {
		const float v = in[i];
		lo = v < lo ? v : lo;
		hi = v > hi ? v : hi;
	}
	for (int off = 32; off > 0; off >>= 1) {
		const float l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
		lo = l2 < lo ? l2 : lo;
		hi = h2 > hi ? h2 : hi;
	}
	if ((threadIdx.x & 63) == 0) {
		atomicMin(&minmax[0], float_to_ordered(lo));
		atomicMax(&minmax[1], float_to_ordered(hi));
	}
}

__global__ __launch_bounds__(kBlock) void quantize_ranged_q8_kernel(const float* in, size_t n_values,
                                                                    const uint32_t* minmax,
                                                                    int8_t* out, float* scale_offset) {
	const float min_val = ordered_to_float(minmax[0]);
	const float max_val = ordered_to_float(minmax[1]);
	const float scale = 128.0f / (max_val - min_val);  // q_range() = 127 - 0 + 1
	const float offset = __fsub_rn(__fmul_rn(-scale, min_val), 0.0f);
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if (i == 0) {
		scale_offset[0] = scale;
		scale_offset[1] = offset;
	}
	if (i >= n_values)
		return;
	// scale*x + offset as a separate multiply and add (no FMA contraction), then std::round
	const float r = roundf(__fadd_rn(__fmul_rn(scale, in[i]), offset));
	// the reference converts through size_t (negative -> UB); negatives clamp to q_min = 0
	const float c = r <= 0.0f ? 0.0f : (r >= 127.0f ? 127.0f : r);
	out[i] = (int8_t)(int)c;
}

}  // namespace expann
