/*
 * gather_rccl_example.c -- how a plain-C caller assembles the final stream of a batch that was
 * block-sharded over the GPUs of a node.
 *
 * Every rank has compressed its own block range with csnappy_hip_compress_batch (no collective on
 * that path).  What follows is the only communication there is:
 *
 *   1. csnappy_hip_dense_offsets   exclusive sum of the rank's compressed lengths (device)
 *   2. csnappy_hip_compact_batch   slot-strided output -> one dense stream per rank (device)
 *   3. ncclAllGather               every rank's byte count (8 bytes per rank)
 *   4. csnappy_hip_gather_layout   where each rank's stream lands in the assembled one (host)
 *   5. ncclGroupStart / ncclRecv x (R-1) on the root, ncclSend on the peers / ncclGroupEnd
 *      -- seven peers use seven different xGMI links into the root; nothing is staged or padded
 *
 * The library does not link RCCL; this file shows the calls a caller makes with ITS communicator.
 *
 * It is also a program (main() below): one process, one GPU, one communicator of ONE rank -- all a
 * 1-GPU test box can run.  It compresses NBLOCKS blocks of G_text, gathers "all ranks'" streams
 * to the root and writes the assembled stream to the file named on the command line;
 * the Makefile builds it when ROCm's rccl.h and librccl.so are there (gcc, against
 * include/csnappy_hip.h, libcsnappy.so and $(ROCM)/lib/librccl.so + libamdhip64.so -- a stand-alone C
 * process uses ROCm's own runtime, not the one bundled with torch); tests/test_gpu_parity.py runs it
 * and compares the file with the stream
 * csnappy_amd/shard.py compacts from the same batch.  With one rank the ncclAllGather and the
 * grouped (empty) exchange do run on the device; ncclSend/ncclRecv between ranks stay unmeasured
 * until an 8-GPU node sees this (the gloo tests cover the layout logic).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/csnappy.h"
#include "../include/csnappy_hip.h"

#define MAX_RANKS 64

/* -> 0, or a negative code.  On the root *d_stream_out receives a hipMalloc'ed buffer with the
 * assembled stream and *stream_bytes its length; on the other ranks both are left alone. */
int gather_final_stream(const void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_len, uint32_t nblocks,
			int rank, int nranks, int root, ncclComm_t comm, hipStream_t stream, void **d_stream_out,
			uint64_t *stream_bytes)
{
	uint64_t *d_dense_off = NULL, *d_total = NULL, *d_counts = NULL;
	void *d_scan_ws = NULL, *d_dense = NULL, *d_all = NULL;
	uint64_t counts[MAX_RANKS], offs[MAX_RANKS], total = 0, mine = 0;
	size_t scan_ws = csnappy_hip_dense_offsets_workspace_size(nblocks);
	int rc = -1, r;

	if (nranks > MAX_RANKS)
		return -1;
	if (hipMalloc((void **)&d_dense_off, (size_t)(nblocks + 1) * 8) != hipSuccess ||
	    hipMalloc((void **)&d_total, 8) != hipSuccess || hipMalloc(&d_scan_ws, scan_ws) != hipSuccess ||
	    hipMalloc((void **)&d_counts, (size_t)nranks * 8) != hipSuccess)
		goto out;
	/* 1. offsets of the rank's blocks in its dense stream, and the stream's length */
	if (csnappy_hip_dense_offsets(d_out_len, nblocks, d_dense_off, d_total, d_scan_ws, scan_ws, stream))
		goto out;
	if (hipMemcpyAsync(&mine, d_total, 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
	    hipStreamSynchronize(stream) != hipSuccess)
		goto out;
	/* 2. the dense stream */
	if (hipMalloc(&d_dense, mine ? mine : 1) != hipSuccess ||
	    csnappy_hip_compact_batch(d_out, d_out_off, d_out_len, d_dense_off, nblocks, d_dense, stream))
		goto out;
	/* 3. size exchange */
	if (ncclAllGather(d_total, d_counts, 1, ncclUint64, comm, stream) != ncclSuccess ||
	    hipMemcpyAsync(counts, d_counts, (size_t)nranks * 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
	    hipStreamSynchronize(stream) != hipSuccess)
		goto out;
	/* 4. layout */
	csnappy_hip_gather_layout(counts, (uint32_t)nranks, offs, &total);
	/* 5. one grouped exchange */
	if (rank == root) {
		if (hipMalloc(&d_all, total ? total : 1) != hipSuccess ||
		    hipMemcpyAsync((char *)d_all + offs[root], d_dense, mine, hipMemcpyDeviceToDevice, stream) != hipSuccess)
			goto out;
	}
	if (ncclGroupStart() != ncclSuccess)
		goto out;
	if (rank == root) {
		for (r = 0; r < nranks; r++)
			if (r != root && counts[r])
				(void)ncclRecv((char *)d_all + offs[r], counts[r], ncclUint8, r, comm, stream);
	} else if (mine) {
		(void)ncclSend(d_dense, mine, ncclUint8, root, comm, stream);
	}
	if (ncclGroupEnd() != ncclSuccess || hipStreamSynchronize(stream) != hipSuccess)
		goto out;
	if (rank == root) {
		*d_stream_out = d_all;
		*stream_bytes = total;
		d_all = NULL;
	}
	rc = 0;
out:
	(void)hipFree(d_dense_off);
	(void)hipFree(d_total);
	(void)hipFree(d_scan_ws);
	(void)hipFree(d_counts);
	(void)hipFree(d_dense);
	(void)hipFree(d_all);
	return rc;
}

#ifndef GATHER_EXAMPLE_NO_MAIN
/* usage: gather_rccl_example OUT_FILE [nblocks [seed]]  -- see the header comment */
#define CHECK(x)                                                                                    \
	do {                                                                                        \
		if (!(x)) {                                                                         \
			fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #x,           \
				csnappy_hip_last_error());                                          \
			return 1;                                                                   \
		}                                                                                   \
	} while (0)

int main(int argc, char **argv)
{
	const uint32_t block = 65536, nblocks = argc > 2 ? (uint32_t)strtoul(argv[2], NULL, 0) : 48;
	const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 0) : 0xC5A90001ull;
	const uint32_t slot = csnappy_max_compressed_length(block);
	const size_t ws_bytes = csnappy_hip_compress_workspace_size(nblocks, block);
	uint64_t *h_off = malloc((size_t)nblocks * 8 * 2), total = 0;
	uint32_t *h_len = malloc((size_t)nblocks * 4), b;
	void *d_in, *d_out, *d_ws, *d_in_off, *d_out_off, *d_in_len, *d_out_len, *d_stream = NULL;
	hipStream_t stream;
	ncclUniqueId id;
	ncclComm_t comm;
	char *h_stream;
	FILE *f;

	if (argc < 2 || !h_off || !h_len) {
		fprintf(stderr, "usage: %s OUT_FILE [nblocks [seed]]\n", argv[0]);
		return 2;
	}
	CHECK(csnappy_hip_device_count() > 0);
	CHECK(hipSetDevice(0) == hipSuccess && hipStreamCreate(&stream) == hipSuccess);
	/* the caller's communicator: here of one rank.  (Progress goes to stderr: a caller that gives up
	 * waiting can tell RCCL's start-up, which is the machine's, from the steps that follow.) */
	fprintf(stderr, "stage: creating a one-rank communicator\n");
	fflush(stderr);
	CHECK(ncclGetUniqueId(&id) == ncclSuccess && ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess);
	fprintf(stderr, "stage: communicator up\n");
	fflush(stderr);
	CHECK(hipMalloc(&d_in, (size_t)nblocks * block) == hipSuccess);
	CHECK(hipMalloc(&d_out, (size_t)nblocks * slot) == hipSuccess && hipMalloc(&d_ws, ws_bytes) == hipSuccess);
	CHECK(hipMalloc(&d_in_off, (size_t)nblocks * 8) == hipSuccess && hipMalloc(&d_out_off, (size_t)nblocks * 8) == hipSuccess);
	CHECK(hipMalloc(&d_in_len, (size_t)nblocks * 4) == hipSuccess && hipMalloc(&d_out_len, (size_t)nblocks * 4) == hipSuccess);
	for (b = 0; b < nblocks; b++) {
		h_off[b] = (uint64_t)b * block;
		h_off[nblocks + b] = (uint64_t)b * slot;
		h_len[b] = block;
	}
	CHECK(hipMemcpy(d_in_off, h_off, (size_t)nblocks * 8, hipMemcpyHostToDevice) == hipSuccess);
	CHECK(hipMemcpy(d_out_off, h_off + nblocks, (size_t)nblocks * 8, hipMemcpyHostToDevice) == hipSuccess);
	CHECK(hipMemcpy(d_in_len, h_len, (size_t)nblocks * 4, hipMemcpyHostToDevice) == hipSuccess);
	/* this rank's block range, compressed with no collective */
	CHECK(csnappy_hip_workload_generate(0, seed, 0, nblocks, block, d_in, stream) == 0);
	CHECK(csnappy_hip_compress_batch(d_in, d_in_off, d_in_len, nblocks, block, d_out, d_out_off, d_out_len, 16,
					 CSNAPPY_HIP_STREAM, d_ws, ws_bytes, stream) == 0);
	/* the one collective of the path */
	CHECK(gather_final_stream(d_out, d_out_off, d_out_len, nblocks, 0, 1, 0, comm, stream, &d_stream, &total) == 0);
	CHECK((h_stream = malloc(total ? total : 1)) != NULL);
	CHECK(hipMemcpy(h_stream, d_stream, total, hipMemcpyDeviceToHost) == hipSuccess);
	CHECK((f = fopen(argv[1], "wb")) != NULL && fwrite(h_stream, 1, total, f) == total && fclose(f) == 0);
	printf("gathered %llu bytes of %u blocks from 1 rank\n", (unsigned long long)total, nblocks);
	ncclCommDestroy(comm);
	return 0;
}
#endif
