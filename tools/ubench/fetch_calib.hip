// Calibration (GPU box, under rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ_sum): known numbers of bytes
// read with the parser's access patterns, so that the counter can be read for THEM (the guide
// calibrates FETCH_SIZE for wide coalesced reads only).  Every kernel reads a 256 MiB region once.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// 16 B per lane, coalesced: every byte of the region exactly once
extern "C" __global__ void calib_coalesced16(const uint4 *p, uint32_t *o, size_t n16)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	for (; i < n16; i += (size_t)gridDim.x * blockDim.x) {
		uint4 v = p[i];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678)
		o[0] = acc;
}

// 16 B per lane, one lane per 64-byte line (the candidate gather's granularity): each line of the
// region is touched once, 16 of its 64 bytes are used
extern "C" __global__ void calib_gather16_per_line(const uint8_t *p, uint32_t *o, size_t nlines)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
		// scatter the lines of a wave across the region so that lanes do not share DRAM pages
		size_t line = (i * 2654435761ull) % nlines;
		uint4 v;
		__builtin_memcpy(&v, p + line * 64 + 20, 16);
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678)
		o[0] = acc;
}

// 2 B per lane, coalesced (the id stream)
extern "C" __global__ void calib_coalesced2(const uint16_t *p, uint32_t *o, size_t n2)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	for (; i < n2; i += (size_t)gridDim.x * blockDim.x)
		acc += p[i];
	if (acc == 0x12345678)
		o[0] = acc;
}

// 16 B per lane at byte stride 1 (the lanes' own bytes): 64 + 15 bytes per wave-instruction
extern "C" __global__ void calib_bytestride16(const uint8_t *p, uint32_t *o, size_t nbytes)
{
	size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
	uint32_t acc = 0;
	const size_t nw = (size_t)gridDim.x * blockDim.x / 64;
	for (size_t base = w * 64; base + 80 <= nbytes; base += nw * 64) {
		uint4 v;
		__builtin_memcpy(&v, p + base + lane, 16);
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678)
		o[0] = acc;
}

int main()
{
	const size_t bytes = 256ull << 20;
	uint8_t *buf;
	uint32_t *out;
	hipMalloc(&buf, bytes);
	hipMalloc(&out, 64);
	hipMemset(buf, 1, bytes);
	hipDeviceSynchronize();
	hipLaunchKernelGGL(calib_coalesced16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, out, bytes / 16);
	hipLaunchKernelGGL(calib_gather16_per_line, dim3(4096), dim3(256), 0, 0, buf, out, bytes / 64);
	hipLaunchKernelGGL(calib_coalesced2, dim3(4096), dim3(256), 0, 0, (const uint16_t *)buf, out, bytes / 2);
	hipLaunchKernelGGL(calib_bytestride16, dim3(4096), dim3(256), 0, 0, buf, out, bytes);
	hipDeviceSynchronize();
	printf("region %zu bytes, %zu lines of 64 B\n", bytes, bytes / 64);
	return 0;
}
