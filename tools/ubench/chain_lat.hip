// Microbenchmark (GPU box): cycles per link of the dependent chains the codec's kernels are made
// of, one wave alone on a CU (s_memtime around 256 links, best of 5).
// hipcc --offload-arch=gfx950 -O3 -o chain_lat chain_lat.hip && ./chain_lat
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void __launch_bounds__(64) k(unsigned long long *out, const uint32_t *tabg)
{
	__shared__ uint32_t lds[64];
	const uint32_t lane = threadIdx.x;
	uint32_t nx = (lane * 7 + 3) & 63; // a permutation of the lanes
	lds[lane] = nx * 4;
	__syncthreads();
	unsigned long long t0, t1;
	uint32_t cur = tabg[0] & 63, v = lane, acc = 0;
	unsigned long long mask = 0;
	int slot = 0;
#define BEGIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define END(n) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
		if (lane == 0) out[slot] = (t1 - t0); slot++; (void)(n); } while (0)
	// 0: empty
	BEGIN(); END(0);
	// 1: s_add chain (SALU -> SALU)
	BEGIN(); REP64(asm volatile("s_add_u32 %0, %0, 1" : "+s"(cur));) END(64);
	// 2: v_add chain (VALU -> VALU)
	BEGIN(); REP64(asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));) END(64);
	// 3: v_readlane -> v_readlane (lane select from the previous one)
	BEGIN(); REP64(asm volatile("s_nop 3\n\tv_readlane_b32 %0, %1, %0" : "+s"(cur) : "v"(nx));) END(64);
	// 4: v_readlane -> s_bitset -> (next readlane independent of the bitset, dependent on the readlane)
	BEGIN(); REP64(asm volatile("s_bitset1_b64 %1, %0\n\ts_nop 2\n\tv_readlane_b32 %0, %2, %0" : "+s"(cur), "+s"(mask) : "v"(nx));) END(64);
	// 5: v_readlane -> SALU use (s_add on the result), next readlane uses the sum
	BEGIN(); REP64(asm volatile("v_readlane_b32 %0, %1, %0\n\ts_add_u32 %0, %0, 0\n\ts_nop 3" : "+s"(cur) : "v"(nx));) END(64);
	// 6: v_cmp (ballot) -> s_and -> v_cndmask on it (VALU -> SGPR pair -> SALU -> VALU)
	BEGIN(); REP64(asm volatile("v_cmp_ne_u32 %1, 0, %0\n\ts_and_b64 %1, %1, exec\n\tv_cndmask_b32 %0, 1, %0, %1" : "+v"(v), "+s"(mask));) END(64);
	// 7: ds_bpermute chain
	{
		uint32_t a = lane * 4;
		BEGIN(); REP64(asm volatile("ds_bpermute_b32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(nx * 4));) END(64);
		acc += a;
	}
	// 8: ds_read_b32 chain (pointer chase in LDS)
	{
		uint32_t a = lane * 4;
		BEGIN(); REP64(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a));) END(64);
		acc += a;
	}
	// 9: four independent v_readlane then four s_bitset (the decompress walk's group)
	{
		uint32_t a, b, c;
		BEGIN(); REP64(asm volatile("v_readlane_b32 %1, %5, %0\n\tv_readlane_b32 %2, %5, %0\n\tv_readlane_b32 %3, %5, %0\n\t"
					    "s_bitset1_b64 %4, %0\n\ts_bitset1_b64 %4, %1\n\ts_bitset1_b64 %4, %2\n\ts_bitset1_b64 %4, %3\n\t"
					    "v_readlane_b32 %0, %5, %0"
					    : "+s"(cur), "=&s"(a), "=&s"(b), "=&s"(c), "+s"(mask) : "v"(nx));) END(64);
	}
	// 10: global load chain, L2/L1-resident line (pointer chase through one 256-byte table)
	{
		uint32_t a = (lane & 63) * 4;
		BEGIN(); REP16(asm volatile("global_load_dword %0, %0, %1\n\ts_waitcnt vmcnt(0)" : "+v"(a) : "s"(tabg));) END(16);
		acc += a;
	}
	// 11: taken branch (s_cbranch_scc1 over nothing), 64 times
	BEGIN(); REP64(asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" ::: "scc");) END(64);
	// 12: s_nop 3
	BEGIN(); REP64(asm volatile("s_nop 3");) END(64);
	// 13: independent v_mov x64
	BEGIN(); REP64(asm volatile("v_mov_b32 %0, 1" : "=v"(v));) END(64);
	// 14: v_readlane -> v_writelane (VALU only marks) -> v_readlane
	BEGIN(); REP64(asm volatile("s_nop 3\n\tv_writelane_b32 %1, 1, %0\n\tv_readlane_b32 %0, %2, %0" : "+s"(cur), "+v"(v) : "v"(nx));) END(64);
	if (lane == 0)
		out[31] = acc + cur + v + (uint32_t)mask;
}

int main()
{
	unsigned long long *out;
	uint32_t *tab;
	hipMalloc(&out, 32 * 8);
	hipMalloc(&tab, 256);
	uint32_t h[64];
	for (int i = 0; i < 64; ++i)
		h[i] = ((i * 7 + 3) & 63) * 4; // byte offsets: the load's own address for the next link
	hipMemcpy(tab, h, 256, hipMemcpyHostToDevice);
	const char *names[] = { "empty", "s_add chain", "v_add chain", "readlane -> readlane (s_nop 3)", "bitset; nop 2; readlane",
				"readlane -> s_add -> readlane", "v_cmp -> s_and -> v_cndmask", "ds_bpermute chain", "ds_read_b32 chain",
				"4 readlanes + 4 bitsets group", "global_load chain (cached)", "taken branch", "s_nop 3", "independent v_mov",
				"nop 3; writelane; readlane" };
	const int links[] = { 1, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 64, 64, 64, 64 };
	unsigned long long best[15];
	for (int i = 0; i < 15; ++i)
		best[i] = ~0ull;
	for (int rep = 0; rep < 5; ++rep) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, tab);
		unsigned long long r[32];
		hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
		for (int i = 0; i < 15; ++i)
			if (r[i] < best[i])
				best[i] = r[i];
	}
	for (int i = 0; i < 15; ++i)
		printf("%-36s %8llu ticks  %7.1f per link (empty removed)\n", names[i], best[i],
		       i ? (double)(best[i] - best[0]) / links[i] : 0.0);
	return 0;
}
