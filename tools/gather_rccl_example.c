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
 * It is compiled (not run: there is one GPU per test box) by tests/test_abi.py.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>

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
