#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
// D[m][n] = sum_k A[m][k] * B[k][n], M=N=32, K=2: lane l supplies A[l%32][l/32] and B[l/32][l%32]
__global__ void k(const float* A, const float* B, float* D) {  // A: 32x2 row-major, B: 2x32 row-major, D: 32x32 row-major
	int l = threadIdx.x;
	float a = A[(l % 32) * 2 + l / 32], b = B[(l / 32) * 32 + l % 32];
	f16v c = {0};
	c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
	for (int i = 0; i < 16; ++i) D[((i / 4) * 8 + (l / 32) * 4 + (i % 4)) * 32 + (l % 32)] = c[i];
}
__global__ void sw(unsigned* out) {
	unsigned l = threadIdx.x;
	unsigned x = 100 + l, y = 200 + l;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
	auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
	out[l] = r[0]; out[64 + l] = r[1];
#else
	out[l] = 0xdead; out[64 + l] = 0xdead;
#endif
}
int main() {
	float hA[64], hB[64], hD[1024], *dA, *dB, *dD; unsigned *dO, hO[128];
	for (int i = 0; i < 64; ++i) { hA[i] = 1 + i; hB[i] = 0.5f * (i % 7) - 1; }
	hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096); hipMalloc(&dO, 512);
	hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
	k<<<1, 64>>>(dA, dB, dD); sw<<<1, 64>>>(dO);
	hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(hO, dO, 512, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float ref = hA[m * 2] * hB[n] + hA[m * 2 + 1] * hB[32 + n]; if (ref != hD[m * 32 + n]) ++bad; }
	printf("mfma layout mismatches: %d\n", bad);
	printf("swap r0: lane0=%u lane31=%u lane32=%u lane63=%u | r1: lane0=%u lane31=%u lane32=%u lane63=%u\n", hO[0], hO[31], hO[32], hO[63], hO[64], hO[95], hO[96], hO[127]);
	return 0;
}
