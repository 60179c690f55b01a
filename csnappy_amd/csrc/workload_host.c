/*
 * workload_host.c -- host form of the synthetic workload generators (see workload_gen.h).
 * Plain C; works without a GPU.  Bench/test input, not part of the reference.
 */
#include <stddef.h>
#include "workload_gen.h"

void csnappy_workload_generate_host(int kind, uint64_t seed, uint64_t first_block,
				    uint32_t nblocks, uint32_t block_len, void *out)
{
	uint32_t b;
	for (b = 0; b < nblocks; b++)
		wg_fill_block(kind, seed, first_block + b, (uint8_t *)out + (size_t)b * block_len,
			      block_len);
}
