// Microbenchmark (GPU box): which lane's data survives when several lanes of ONE ds_write hit the
// same LDS address, and in which order the lanes of one returning LDS atomic are served.  The ISA
// manual leaves both open; the parser's tentative-write conflict detection (csnappy_kernels.hip,
// parse_lean, "TW") would be one instruction shorter per step if the answer were "lowest lane".
// Random patterns: every lane picks one of R slots (R = 2 .. 64), many waves, many rounds, 16-bit
// and 32-bit stores, neighbours in one dword included.
// hipcc --offload-arch=gfx950 -O3 -o lds_order lds_order.hip && ./lds_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(64) k_write(const uint16_t *slots, uint16_t *win16, uint32_t *win32, uint32_t *xchg,
						uint32_t *addr)
{
	__shared__ uint16_t t16[4096];
	__shared__ uint32_t t32[4096];
	__shared__ uint32_t tx[4096];
	__shared__ uint32_t ta[4096];
	const uint32_t lane = threadIdx.x, w = blockIdx.x;
	for (uint32_t k = lane; k < 4096; k += 64) {
		t16[k] = 0xffff;
		t32[k] = 0xffffffffu;
		tx[k] = 1000;
		ta[k] = 0;
	}
	__syncthreads();
	const uint32_t s = slots[w * 64 + lane];
	t16[s] = (uint16_t)lane;
	t32[s] = lane;
	const uint32_t old = __hip_atomic_exchange(&tx[s], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	const uint32_t pre = __hip_atomic_fetch_add(&ta[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	__syncthreads();
	win16[w * 64 + lane] = t16[s];
	win32[w * 64 + lane] = t32[s];
	xchg[w * 64 + lane] = old;
	addr[w * 64 + lane] = pre;
}

int main()
{
	const int W = 65536;
	uint16_t *h_s = (uint16_t *)malloc(W * 64 * 2);
	srand(12345);
	for (int w = 0; w < W; ++w) {
		const int kind = w & 7;
		const int R = kind == 0 ? 2 : kind == 1 ? 4 : kind == 2 ? 16 : kind == 3 ? 40 : kind == 4 ? 64 : kind == 5 ? 200 : kind == 6 ? 1000 : 4096;
		const int base = rand() % (4096 - (R < 4096 ? R : 0) + (R == 4096));
		for (int l = 0; l < 64; ++l)
			h_s[w * 64 + l] = (uint16_t)((R == 4096 ? 0 : base) + rand() % R);
	}
	uint16_t *d_s, *d_w16;
	uint32_t *d_w32, *d_x, *d_a;
	hipMalloc(&d_s, W * 64 * 2);
	hipMalloc(&d_w16, W * 64 * 2);
	hipMalloc(&d_w32, W * 64 * 4);
	hipMalloc(&d_x, W * 64 * 4);
	hipMalloc(&d_a, W * 64 * 4);
	hipMemcpy(d_s, h_s, W * 64 * 2, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k_write, dim3(W), dim3(64), 0, 0, d_s, d_w16, d_w32, d_x, d_a);
	uint16_t *w16 = (uint16_t *)malloc(W * 64 * 2);
	uint32_t *w32 = (uint32_t *)malloc(W * 64 * 4), *x = (uint32_t *)malloc(W * 64 * 4), *a = (uint32_t *)malloc(W * 64 * 4);
	hipMemcpy(w16, d_w16, W * 64 * 2, hipMemcpyDeviceToHost);
	hipMemcpy(w32, d_w32, W * 64 * 4, hipMemcpyDeviceToHost);
	hipMemcpy(x, d_x, W * 64 * 4, hipMemcpyDeviceToHost);
	if (hipMemcpy(a, d_a, W * 64 * 4, hipMemcpyDeviceToHost) != hipSuccess) {
		printf("hip error\n");
		return 1;
	}
	long groups = 0, hi16 = 0, lo16 = 0, hi32 = 0, lo32 = 0, xord = 0, xtot = 0, aord = 0, atot = 0;
	for (int w = 0; w < W; ++w)
		for (int l = 0; l < 64; ++l) {
			const int s = h_s[w * 64 + l];
			int lo = 64, hi = -1, prev = -1, rank = 0;
			for (int j = 0; j < 64; ++j)
				if (h_s[w * 64 + j] == s) {
					lo = j < lo ? j : lo;
					hi = j > hi ? j : hi;
					if (j < l) {
						prev = j;
						rank++;
					}
				}
			if (lo != hi && l == lo) {
				groups++;
				hi16 += w16[w * 64 + l] == hi;
				lo16 += w16[w * 64 + l] == lo;
				hi32 += (int)w32[w * 64 + l] == hi;
				lo32 += (int)w32[w * 64 + l] == lo;
			}
			if (lo != hi) {
				xtot++;
				xord += x[w * 64 + l] == (prev < 0 ? 1000u : (uint32_t)prev);
				atot++;
				aord += (int)a[w * 64 + l] == rank;
			}
		}
	printf("slots shared by several lanes of one store: %ld\n", groups);
	printf("  ds_write_b16: highest lane survives %ld, lowest %ld\n", hi16, lo16);
	printf("  ds_write_b32: highest lane survives %ld, lowest %ld\n", hi32, lo32);
	printf("lanes in such slots: %ld\n", xtot);
	printf("  ds_wrxchg_rtn_b32 returns the nearest lower lane's value (lane order): %ld\n", xord);
	printf("  ds_add_rtn_u32 returns the number of lower lanes (lane order): %ld of %ld\n", aord, atot);
	return 0;
}
