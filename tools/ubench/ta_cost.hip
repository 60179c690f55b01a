// Microbenchmark (GPU box): cycles the per-CU texture addresser spends on one wave-instruction of
// the parser's load patterns.  hipcc --offload-arch=gfx950 -O3 -o ta_cost ta_cost.hip && ./ta_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ void __launch_bounds__(64) k(const uint8_t *buf, uint32_t *out, int iters, uint32_t seed, uint32_t rmask, uint32_t stride)
{
	const uint32_t lane = threadIdx.x;
	const uint8_t *base = buf + (size_t)blockIdx.x * stride; // each wave its own region
	uint32_t acc = 0, x = seed ^ (blockIdx.x * 2654435761u) ^ lane * 40503u;
	uint32_t p = 0;
	for (int i = 0; i < iters; ++i) {
		x = x * 1664525u + 1013904223u;
		uint32_t r = (x >> 8) & rmask; // random position in the wave's window
		if (MODE == 0) { // 16 B per lane at consecutive BYTE offsets (the parser's own-bytes load)
			uint4 v;
			__builtin_memcpy(&v, base + ((p + lane) & 32767u), 16);
			acc += v.x ^ v.y ^ v.z ^ v.w;
		} else if (MODE == 1) { // 16 B per lane at random byte positions (the candidate gather)
			uint4 v;
			__builtin_memcpy(&v, base + r, 16);
			acc += v.x ^ v.y ^ v.z ^ v.w;
		} else if (MODE == 2) { // u16 per lane, consecutive (the id load)
			uint16_t v;
			__builtin_memcpy(&v, base + 32768 + 2 * ((p + lane) & 16383u), 2);
			acc += v;
		} else if (MODE == 3) { // dword per lane, consecutive, aligned (a coalesced window fetch)
			uint32_t v;
			__builtin_memcpy(&v, base + ((p & ~3u) & 32767u) + 4 * lane, 4);
			acc += v;
		} else if (MODE == 4) { // 16 B per lane at random positions, half the lanes masked off
			if (lane & 1) {
				uint4 v;
				__builtin_memcpy(&v, base + r, 16);
				acc += v.x ^ v.y ^ v.z ^ v.w;
			}
		} else if (MODE == 5) { // 16 B per lane, consecutive 16 B chunks (fully coalesced 1 KiB)
			uint4 v;
			__builtin_memcpy(&v, base + (((p & ~15u) + 16 * lane) & 32767u), 16);
			acc += v.x ^ v.y ^ v.z ^ v.w;
		} else if (MODE == 6) { // 4 B per lane at random positions
			uint32_t v;
			__builtin_memcpy(&v, base + r, 4);
			acc += v;
		} else if (MODE == 7) { // 8 B record store by 6 lanes
			if (lane < 6)
				*(uint2 *)(const_cast<uint8_t *>(base) + 49152 + 8 * ((p / 8 + lane) & 1023u)) = make_uint2(x, p);
		}
		p += 61;
	}
	out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE> void run(const char *name, const uint8_t *buf, uint32_t *out, int waves, int iters, uint32_t rmask = 32767u, uint32_t stride = 65536)
{
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, buf, out, iters, 12345u, rmask, stride);
	hipEventRecord(a);
	hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, buf, out, iters, 999u, rmask, stride);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	// waves/256 per CU, each iters instructions; at 2.4 GHz
	const double per_cu_instr = (double)waves / 256 * iters;
	printf("%-52s %8.3f ms  %7.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", name, ms,
	       ms * 1e-3 * 2.4e9 / per_cu_instr);
}

int main()
{
	const int waves = 256 * 16, iters = 4096;
	uint8_t *buf;
	uint32_t *out;
	hipMalloc(&buf, (size_t)waves * 65536);
	hipMalloc(&out, (size_t)waves * 64 * 4);
	hipMemset(buf, 1, (size_t)waves * 65536);
	run<0>("own bytes: 16 B/lane, byte stride 1", buf, out, waves, iters);
	run<1>("gather: 16 B/lane, random in 32 KiB", buf, out, waves, iters);
	run<4>("gather, 32 lanes active", buf, out, waves, iters);
	run<6>("gather: 4 B/lane, random in 32 KiB", buf, out, waves, iters);
	run<2>("ids: u16/lane consecutive", buf, out, waves, iters);
	run<3>("dword/lane consecutive aligned", buf, out, waves, iters);
	run<5>("16 B/lane consecutive aligned (1 KiB)", buf, out, waves, iters);
	run<1>("gather 16 B, window 4 KiB/wave (L2-resident)", buf, out, waves, iters, 4095u, 4096);
	run<1>("gather 16 B, window 1 KiB/wave (L1-resident)", buf, out, waves, iters, 1023u, 1024);
	run<1>("gather 16 B, window 32 KiB, waves overlap 8x", buf, out, waves, iters, 32767u, 4096);
	run<4>("gather 16 B 32 lanes, window 4 KiB/wave", buf, out, waves, iters, 4095u, 4096);
	run<7>("record store: 8 B x 6 lanes", buf, out, waves, iters);
	return 0;
}
