// Microbenchmark (GPU box): wave-instructions per cycle and CU that gfx950 issues for the kinds of
// instruction streams the codec's kernels consist of, against the number of waves per CU.
// Every wave runs ITER x 256 instructions of one kind (no memory, no dependencies between
// neighbouring instructions beyond what is stated); time = s_memtime of wave 0 of each CU's
// first workgroup is not enough (waves start at different times), so the whole grid is timed with
// HIP events and many iterations, and converted with the device's reported shader clock.
// hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND> __global__ void __launch_bounds__(64) k(uint32_t *out, int iters)
{
	extern __shared__ uint32_t pad[];
	uint32_t v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
	uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
	unsigned long long m = 0, m2 = 1;
	for (int it = 0; it < iters; ++it) {
		if (KIND == 0) { /* VALU, four independent chains */
			REP64(asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1"
					   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
		} else if (KIND == 1) { /* SALU, four independent chains */
			REP64(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1"
					   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
		} else if (KIND == 2) { /* alternating VALU / SALU, independent */
			REP64(asm volatile("v_add_u32 %0, %0, 1\n\ts_add_u32 %2, %2, 1\n\tv_add_u32 %1, %1, 1\n\ts_add_u32 %3, %3, 1"
					   : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) : : "scc");)
		} else if (KIND == 4) { /* compare into VCC, select on VCC (VALU -> VCC -> VALU), two independent pairs */
			REP64(asm volatile("v_cmp_ne_u32 vcc, 0, %0\n\tv_cndmask_b32 %1, 1, %1, vcc\n\tv_cmp_ne_u32 vcc, 0, %2\n\tv_cndmask_b32 %3, 1, %3, vcc"
					   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");)
		} else if (KIND == 5) { /* v_readlane only (VALU -> SGPR), independent */
			REP64(asm volatile("v_readlane_b32 %0, %4, 5\n\tv_readlane_b32 %1, %5, 6\n\tv_readlane_b32 %2, %4, 7\n\tv_readlane_b32 %3, %5, 8"
					   : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(v0), "v"(v1));)
		} else if (KIND == 6) { /* 64-bit scalar mask arithmetic only */
			REP64(asm volatile("s_and_b64 %0, %0, exec\n\ts_or_b64 %1, %1, %0\n\ts_and_b64 %0, %0, exec\n\ts_or_b64 %1, %1, %0"
					   : "+s"(m), "+s"(m2) : : "scc");)
		} else if (KIND == 7) { /* compares into SGPR pairs only (VOP3), independent */
			REP64(asm volatile("v_cmp_ne_u32 %0, 0, %2\n\tv_cmp_ne_u32 %1, 1, %3\n\tv_cmp_ne_u32 %0, 2, %2\n\tv_cmp_ne_u32 %1, 3, %3"
					   : "=s"(m), "=s"(m2) : "v"(v0), "v"(v1));)
		} else if (KIND == 3) { /* the parser's mix: compare -> mask arithmetic -> select, a readlane */
			REP64(asm volatile("v_cmp_ne_u32 %2, 0, %0\n\ts_and_b64 %2, %2, exec\n\tv_cndmask_b32 %1, 1, %1, %2\n\tv_readlane_b32 %3, %0, 5"
					   : "+v"(v0), "+v"(v1), "+s"(m), "+s"(s0) : : "scc");)
		}
	}
	if (v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 + (uint32_t)m + (uint32_t)m2 == 0x12345678u)
		out[0] = pad[0];
}

template <int KIND> static double run(int waves_per_cu, int cus, uint32_t *d_out)
{
	const int iters = 100;
	/* occupancy by LDS: 160 KiB per CU */
	size_t lds = (160 * 1024 / waves_per_cu) & ~255u;
	if (lds > 65536)
		lds = 65536; /* (one or two waves per CU: they do not share a SIMD) */
	if (hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
		printf("hipFuncSetAttribute(%zu) failed\n", lds);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	double best = 1e30;
	for (int rep = 0; rep < 4; ++rep) {
		hipEventRecord(a);
		hipLaunchKernelGGL(k<KIND>, dim3(cus * waves_per_cu), dim3(64), lds, 0, d_out, iters);
		hipEventRecord(b);
		if (hipEventSynchronize(b) != hipSuccess || hipGetLastError() != hipSuccess) {
			printf("launch failed (kind %d, %d waves per CU)\n", KIND, waves_per_cu);
			fflush(stdout);
			return 0;
		}
		float ms;
		hipEventElapsedTime(&ms, a, b);
		if (ms < best)
			best = ms;
	}
	/* wave-instructions per ns and CU */
	return (double)iters * 256.0 * waves_per_cu / (best * 1e6);
}

int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	uint32_t *d_out;
	hipMalloc(&d_out, 64);
	/* (rates are per nanosecond; divided by the shader clock hipDeviceProp reports, 2.4 GHz on MI355X) */
	const double ghz = p.clockRate / 1e6;
	printf("CUs %d, clock %.2f GHz\n", cus, ghz);
	fflush(stdout);
	printf("%-44s", "wave-instructions per cycle and CU, waves/CU:");
	const int ws[] = { 1, 4, 8, 16, 24, 32 };
	for (int w : ws)
		printf(" %6d", w);
	printf("\n");
	const char *names[] = { "VALU only (v_add_u32)", "SALU only (s_add_u32)", "VALU / SALU alternating",
				"v_cmp, s_and_b64, v_cndmask, v_readlane", "v_cmp vcc, v_cndmask vcc", "v_readlane only",
				"s_and_b64 / s_or_b64 only", "v_cmp into SGPR pairs only" };
	for (int kind = 0; kind < 8; ++kind) {
		printf("%-44s", names[kind]);
		for (int w : ws) {
			const double r = kind == 0 ? run<0>(w, cus, d_out) : kind == 1 ? run<1>(w, cus, d_out)
					 : kind == 2 ? run<2>(w, cus, d_out) : kind == 3 ? run<3>(w, cus, d_out)
					 : kind == 4 ? run<4>(w, cus, d_out) : kind == 5 ? run<5>(w, cus, d_out)
					 : kind == 6 ? run<6>(w, cus, d_out) : run<7>(w, cus, d_out);
			printf(" %6.2f", r / ghz);
			fflush(stdout);
		}
		printf("\n");
	}
	return 0;
}
