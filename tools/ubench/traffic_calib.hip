// traffic_calib.hip -- known byte counts in the access patterns of the path pipeline, to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md "HBM": only the wide coalesced
// read is calibrated there: x2).  Each kernel touches every byte of a 2 GiB buffer (8x the Infinity
// Cache) exactly once, in one pattern:
//   rd16  float4 per lane, consecutive lanes consecutive elements (ray[], direct[], fs[] reads)
//   rd8   float2 per lane                                           (np[] reads)
//   wr16  float4 stores, wr8 float2 stores                          (the per-level stores)
//   rmw   four 4-byte agent-scope atomic loads of a float4 + one float4 store back (shadow_flush)
//   rd16s float4 per lane at 64-element strides between consecutive k (the accumulate pass)
// usage: traffic_calib <pattern> ; prints the bytes the kernel read and wrote.  Run under
//   rocprofv3 --pmc FETCH_SIZE -- traffic_calib rd16     (and WRITE_SIZE in a separate pass)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_rd16(const float4* p, float* sink, size_t n) {
	float acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 123.456f) *sink = acc;
}
__global__ void calib_rd8(const float2* p, float* sink, size_t n) {
	float acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float2 v = p[i]; acc += v.x + v.y; }
	if (acc == 123.456f) *sink = acc;
}
__global__ void calib_wr16(float4* p, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void calib_wr8(float2* p, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float2(1.f, (float)i);
}
__global__ void calib_rmw(float4* p, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		float* d = reinterpret_cast<float*>(p + i);
		float4 o;
		o.x = __hip_atomic_load(d + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		o.y = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		o.z = __hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		o.w = __hip_atomic_load(d + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		p[i] = make_float4(o.x + 1.f, o.y + 1.f, o.z + 1.f, o.w + 1.f);
	}
}
// one lane per "pixel": element (slot*K + k)*64 + lane for k = 0..K-1 (the accumulate pass's order)
__global__ void calib_rd16s(const float4* p, float* sink, size_t n, unsigned K) {
	const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t slot = gid >> 6, lane = gid & 63u;
	if ((slot + 1) * K * 64 > n) return;
	float acc = 0;
	for (unsigned k = 0; k < K; ++k) { float4 v = p[(slot * K + k) * 64 + lane]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 123.456f) *sink = acc;
}

int main(int argc, char** argv) {
	const char* pat = argc > 1 ? argv[1] : "rd16";
	const size_t bytes = (size_t)2 << 30;
	void* buf; float* sink;
	CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 4));
	CHECK(hipMemset(buf, 0, bytes));
	CHECK(hipDeviceSynchronize());
	const int blocks = 256 * 8;
	size_t rd = 0, wr = 0;
	for (int rep = 0; rep < 3; ++rep) {
		if (!strcmp(pat, "rd16")) { hipLaunchKernelGGL(calib_rd16, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, sink, bytes / 16); rd = bytes; }
		else if (!strcmp(pat, "rd8")) { hipLaunchKernelGGL(calib_rd8, dim3(blocks), dim3(256), 0, 0, (const float2*)buf, sink, bytes / 8); rd = bytes; }
		else if (!strcmp(pat, "wr16")) { hipLaunchKernelGGL(calib_wr16, dim3(blocks), dim3(256), 0, 0, (float4*)buf, bytes / 16); wr = bytes; }
		else if (!strcmp(pat, "wr8")) { hipLaunchKernelGGL(calib_wr8, dim3(blocks), dim3(256), 0, 0, (float2*)buf, bytes / 8); wr = bytes; }
		else if (!strcmp(pat, "rmw")) { hipLaunchKernelGGL(calib_rmw, dim3(blocks), dim3(256), 0, 0, (float4*)buf, bytes / 16); rd = bytes; wr = bytes; }
		else if (!strcmp(pat, "rd16s")) { const unsigned K = 256; const size_t n = bytes / 16, slots = n / (K * 64);
			hipLaunchKernelGGL(calib_rd16s, dim3((unsigned)((slots * 64 + 255) / 256)), dim3(256), 0, 0, (const float4*)buf, sink, n, K); rd = slots * K * 64 * 16; }
		else { fprintf(stderr, "unknown pattern %s\n", pat); return 2; }
		CHECK(hipGetLastError());
		CHECK(hipDeviceSynchronize());
	}
	printf("{\"pattern\": \"%s\", \"read_bytes\": %zu, \"write_bytes\": %zu, \"launches\": 3}\n", pat, rd, wr);
	return 0;
}
