// Microbenchmark (GPU box): do the top registers of a kernel that declares EXACTLY 64 VGPRs keep their values
// when eight waves share a SIMD (8 x 64 = all 512 VGPRs of the SIMD)?  Round 6 found the global-table parser --
// 64 VGPRs declared, v60-v63 used by its hand-written loop, 5 KiB of LDS a workgroup: up to 32 waves per CU --
// bit-exact with one wave per SIMD and wrong beside others, and right again with the register map ended at v59
// or with more than 64 VGPRs declared (seven waves a SIMD).  This kernel is the question on its own: every wave
// writes a value of its own into v56 .. v63 (inline asm, the registers named as clobbers so that the kernel's
// VGPR count is exactly 64), works for a while on lower registers, LDS and global memory, and reads them back.
//   hipcc --offload-arch=gfx950 -O3 -o vgpr_top vgpr_top.hip && ./vgpr_top
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int TOP, int MEM> __global__ void __launch_bounds__(64) k_top(uint32_t *bad, const uint32_t *in, uint32_t *out, uint32_t rounds)
{
	extern __shared__ uint32_t lds[];
	const uint32_t lane = threadIdx.x, w = blockIdx.x;
	const uint32_t tag = w * 64 + lane;
	uint32_t wrong = 0;
	for (uint32_t r = 0; r < rounds; ++r) {
		const uint32_t v = tag * 2654435761u + r;
		if (TOP)
			asm volatile("v_mov_b32 v60, %0\n\tv_add_u32 v61, 1, %0\n\tv_add_u32 v62, 2, %0\n\tv_add_u32 v63, 3, %0"
				     : : "v"(v) : "v60", "v61", "v62", "v63");
		else
			asm volatile("v_mov_b32 v52, %0\n\tv_add_u32 v53, 1, %0\n\tv_add_u32 v54, 2, %0\n\tv_add_u32 v55, 3, %0"
				     : : "v"(v) : "v52", "v53", "v54", "v55", "v63");
		/* a while of ordinary work: loads, LDS traffic, arithmetic */
		uint32_t acc = v;
		for (uint32_t k = 0; k < 16; ++k) {
			const uint32_t x = in[(tag * 17 + k * 64 + r) & 0xfffff];
			lds[lane] = x + acc;
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			acc = acc * 33 + lds[(lane + k) & 63];
		}
		/* ... and the same registers as the destination of memory instructions: a 16-byte global load and an LDS
		 * read that return while the other waves of the SIMD run */
		if (MEM) {
			const uint32_t off = (tag * 16 + r * 64) & 0xffff0;
			if (TOP)
				asm volatile("global_load_dwordx4 v[60:63], %0, %1\n\ts_waitcnt vmcnt(0)\n\t"
					     "v_sub_u32 v60, v60, v61\n\tv_sub_u32 v62, v62, v63\n\tv_or_b32 v60, v60, v62\n\t"
					     "v_add_u32 v60, v60, %2\n\tv_add_u32 v61, 1, %2\n\tv_add_u32 v62, 2, %2\n\tv_add_u32 v63, 3, %2"
					     : : "v"(off), "s"(in), "v"(v) : "v60", "v61", "v62", "v63", "memory");
			else
				asm volatile("global_load_dwordx4 v[52:55], %0, %1\n\ts_waitcnt vmcnt(0)\n\t"
					     "v_sub_u32 v52, v52, v53\n\tv_sub_u32 v54, v54, v55\n\tv_or_b32 v52, v52, v54\n\t"
					     "v_add_u32 v52, v52, %2\n\tv_add_u32 v53, 1, %2\n\tv_add_u32 v54, 2, %2\n\tv_add_u32 v55, 3, %2"
					     : : "v"(off), "s"(in), "v"(v) : "v52", "v53", "v54", "v55", "v63", "memory");
		}
		uint32_t a, b, c, d;
		if (TOP)
			asm volatile("v_mov_b32 %0, v60\n\tv_mov_b32 %1, v61\n\tv_mov_b32 %2, v62\n\tv_mov_b32 %3, v63"
				     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : : "v60", "v61", "v62", "v63");
		else
			asm volatile("v_mov_b32 %0, v52\n\tv_mov_b32 %1, v53\n\tv_mov_b32 %2, v54\n\tv_mov_b32 %3, v55"
				     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : : "v52", "v53", "v54", "v55", "v63");
		wrong += (a != v) + (b != v + 1) + (c != v + 2) + (d != v + 3);
		out[tag] = acc;
	}
	if (wrong)
		atomicAdd(bad, wrong);
}

int main()
{
	uint32_t *bad, *in, *out, h;
	hipMalloc(&bad, 4);
	hipMalloc(&in, 4 << 20);
	hipMalloc(&out, 4 * 64 * 65536);
	hipMemset(in, 1, 4 << 20); /* (every dword 0x01010101: the loaded registers cancel out) */
	hipFuncAttributes fa;
	for (int mem = 0; mem < 2; ++mem)
		for (int top = 0; top < 2; ++top) {
			const void *k = mem ? (top ? (const void *)k_top<1, 1> : (const void *)k_top<0, 1>)
					    : (top ? (const void *)k_top<1, 0> : (const void *)k_top<0, 0>);
			hipFuncGetAttributes(&fa, k);
			for (uint32_t lds_bytes : { 5120u, 40960u }) { /* 32 waves a CU / 4 */
				hipMemset(bad, 0, 4);
				uint32_t rounds = 64;
				void *args[] = { &bad, &in, &out, &rounds };
				hipLaunchKernel(k, dim3(65536), dim3(64), args, lds_bytes, 0);
				hipDeviceSynchronize();
				hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
				printf("%s registers%s, %d VGPRs declared, %5u B of LDS a workgroup: %u wrong read-backs of %u\n",
				       top ? "v60-v63" : "v52-v55", mem ? " (also loaded into)" : "", fa.numRegs, lds_bytes, h,
				       65536u * 64 * 64 * 4);
			}
		}
	return 0;
}
