/*
 * csnappy_hip.h -- the batched C-ABI the HIP kernels sit behind.
 *
 * The reference's API hands the codec one caller-owned buffer per synchronous call
 * (csnappy.h:46-72, 99-119).  One <=64 KiB call cannot feed a GPU, so next to the six legacy
 * symbols (include/csnappy.h) the library exports a batch form of the same four operations:
 *
 *   reference call                                   batch mode            this header
 *   csnappy_compress            (csnappy.h:65-72)    CSNAPPY_HIP_STREAM    csnappy_hip_compress_batch
 *   csnappy_compress_fragment   (csnappy.h:46-52)    CSNAPPY_HIP_FRAGMENT  csnappy_hip_compress_batch
 *   csnappy_decompress          (csnappy.h:99-104)   CSNAPPY_HIP_STREAM    csnappy_hip_decompress_batch
 *   csnappy_decompress_noheader (csnappy.h:114-119)  CSNAPPY_HIP_FRAGMENT  csnappy_hip_decompress_batch
 *
 * Block b of a batch is exactly one reference call on (in + in_off[b], in_len[b]) writing to
 * (out + out_off[b]); results (bytes, lengths, status codes) are identical to the reference's.
 * Plain pointers and sizes only.  All data pointers are DEVICE pointers (hipMalloc'ed, or
 * torch tensors' data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Calls enqueue work and return; they do not synchronise -- with ONE exception: the first
 * csnappy_hip_compress_batch of a process on a device first asks that device, on a stream of the
 * library's own, whether its LDS serves the lanes of one instruction in ascending order (two small
 * launches, a 4-byte copy and a wait of about a millisecond); make
 * that first call outside a stream capture.  Return value: 0, or a negative CSNAPPY_HIP_E_* code
 * when the launch itself could not be made.
 */
#ifndef CSNAPPY_AMD_CSNAPPY_HIP_H_
#define CSNAPPY_AMD_CSNAPPY_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSNAPPY_HIP_STREAM 0   /* varint length prefix + 32 KiB fragments (csnappy_compress.c:621-656) */
#define CSNAPPY_HIP_FRAGMENT 1 /* one raw fragment <= 32 KiB, no prefix (csnappy_compress.c:469-606) */

#define CSNAPPY_HIP_E_ARG (-101)     /* bad argument (p outside 9..16, FRAGMENT block > 32 KiB, ...) */
#define CSNAPPY_HIP_E_RUNTIME (-102) /* a HIP runtime call failed; see csnappy_hip_last_error() */
#define CSNAPPY_HIP_E_WORKSPACE (-103) /* workspace smaller than csnappy_hip_compress_workspace_size() */

/* Number of usable HIP devices (0 when there is none; never fails). */
int csnappy_hip_device_count(void);

/* Text of the last HIP runtime error seen by this library on the calling thread. */
const char *csnappy_hip_last_error(void);

/*
 * Device scratch needed by csnappy_hip_compress_batch for `nblocks` blocks whose lengths are
 * all <= max_in_len: per 32 KiB fragment the parser's 8-byte (literal, copy) records (one per four
 * input bytes at most), the 2-byte bucket ids of its positions (or its global-memory hash table),
 * 4 KiB for the table entries of buckets beyond the LDS table, and a record count.  A batch is
 * processed in launches of at least 32 768 fragments, so the size stops growing there: 4.1 GiB for
 * 64 KiB blocks (1 GiB of input per launch), 0.5 GiB for 4 KiB pages (128 MiB per launch).  256-byte
 * aligned base required.  This is the LEAST the batch call accepts.
 *
 * ..._size_for(.., launch_gib): the scratch for launches of up to launch_gib (1..8) GiB of input
 * whatever the block size.  A launch's ramp and tail cost 5-11 % of a 1 GiB launch (and a tenth of a
 * 128 MiB launch of pages); a caller with HBM to spare (MI355X: 288 GB) passes a workspace of this
 * size and the batch call uses the largest launches it has room for (pages in 1 GiB launches: -13 %,
 * 4.0 GiB of scratch; 4 GiB launches: text -5 %, runs -11 %; 16.5 GiB of scratch).
 */
size_t csnappy_hip_compress_workspace_size(uint32_t nblocks, uint32_t max_in_len);
size_t csnappy_hip_compress_workspace_size_for(uint32_t nblocks, uint32_t max_in_len, uint32_t launch_gib);

/*
 * Compress nblocks independent blocks.
 *   d_in, d_in_off[b], d_in_len[b]   input bytes of block b            (in_len[b] <= max_in_len: the
 *                                    workspace is sized by max_in_len; a longer block is not
 *                                    compressed and gets d_out_len[b] = 0xffffffff)
 *   d_out, d_out_off[b]              start of block b's output slot, which must hold
 *                                    csnappy_max_compressed_length(in_len[b]) bytes (unchecked,
 *                                    as in the reference)
 *   d_out_len[b]                     receives the compressed size of block b
 *   p                                workmem_bytes_power_of_two, 9..16 (changes the bytes)
 *   mode                             CSNAPPY_HIP_STREAM | CSNAPPY_HIP_FRAGMENT
 */
int csnappy_hip_compress_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
			       uint32_t nblocks, uint32_t max_in_len, void *d_out,
			       const uint64_t *d_out_off, uint32_t *d_out_len, int p, int mode,
			       void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * Decompress nblocks independent blocks.
 *   d_out_cap[b]    STREAM: dst_len of csnappy_decompress; FRAGMENT: *dst_len on entry of
 *                   csnappy_decompress_noheader
 *   d_status[b]     CSNAPPY_E_* code the reference call would return (0, -1, -2, -3, -5)
 *   d_produced[b]   bytes produced (FRAGMENT, status 0: *dst_len on exit).  With status -3 / -5
 *                   it is the length of the decoded prefix that is in the slot: the elements in
 *                   front of the failing one, which the reference's write-as-you-go writer has
 *                   stored by then too (csnappy_decompress.c:258-317); 0 with -1 / -2
 * Bytes of the output slot beyond `produced` are never written when status is 0, and never beyond
 * out_cap[b] otherwise.  in_len[b] and out_cap[b] must
 * be below 2^32 - 2^16 (the kernels keep 32-bit cursors like the reference's uint32 API).
 */
int csnappy_hip_decompress_batch(const void *d_in, const uint64_t *d_in_off,
				 const uint32_t *d_in_len, uint32_t nblocks, void *d_out,
				 const uint64_t *d_out_off, const uint32_t *d_out_cap,
				 int32_t *d_status, uint32_t *d_produced, int mode, void *stream);

/*
 * Decompress ONE stream body of any length with the whole device.  Same contract
 * as csnappy_decompress_noheader(d_in, in_len, d_out, &ulength) (csnappy_decompress.c:319-387):
 * `ulength` is *dst_len on entry (the room in d_out), d_status[0] receives the reference's return
 * code, d_produced[0] the bytes produced (status 0: *dst_len on exit, possibly less than ulength;
 * -3 / -5: the length of the decoded prefix in d_out -- the bytes of d_out between `produced` and
 * `ulength` are UNSPECIFIED after an error: the parallel fragment pass has written there before the
 * one-wave decode re-did the prefix; with status 0 nothing beyond `produced` is written).  csnappy_decompress (:390-415) is this call after the length header has been parsed
 * (the header's value is `ulength`).
 *
 * A pre-pass indexes the tags of the body with one wave per 4 KiB and looks for the elements that
 * start at multiples of 32 KiB of output; streams written by csnappy_compress have one at each
 * (csnappy_compress.c:585-616 restarts the matcher there; Snappy's own 32 KiB or 64 KiB blocks
 * are recognised the same way), and their fragments are then decoded as independent blocks by
 * csnappy_hip_decompress_batch's kernel.  Whenever that does not work out
 * -- a foreign compressor that copies across 32 KiB, a damaged stream, a stream that yields more
 * than ulength, any fragment that does not decode cleanly to exactly its size -- the body is
 * decoded by one wave as in the batch call, so
 * the status and the bytes are the reference's for every input; only the time differs.
 * Asynchronous on `stream`.  The workspace (csnappy_hip_decompress_stream_workspace_size bytes,
 * 16-byte aligned, about 2.4 x in_len + 40 bytes per 32 KiB of output) holds nothing across calls.
 * in_len and ulength must be below 2^32 - 2^16.
 */
size_t csnappy_hip_decompress_stream_workspace_size(uint32_t in_len, uint32_t ulength);
int csnappy_hip_decompress_stream(const void *d_in, uint32_t in_len, uint32_t ulength, void *d_out,
				  int32_t *d_status, uint32_t *d_produced, void *d_workspace,
				  size_t workspace_bytes, void *stream);
/* 1 if the last csnappy_hip_decompress_stream call on this workspace kept the fragments' result,
 * 0 if the one-wave decode produced it (waits for `stream`; for tests and tools) */
int csnappy_hip_decompress_stream_took_fast_path(const void *d_workspace, uint32_t in_len,
						  uint32_t ulength, void *stream);

/*
 * Pack the slot-strided output of csnappy_hip_compress_batch into one dense stream:
 * block b's d_out_len[b] bytes go to d_dense + d_dense_off[b] (the caller supplies the exclusive
 * scan of the lengths).  This is what the reference's callers do with their own memcpy after
 * each csnappy_compress call (block_compressor.c:316-334); it precedes the multi-GPU gather.
 */
int csnappy_hip_compact_batch(const void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_len,
			      const uint64_t *d_dense_off, uint32_t nblocks, void *d_dense, void *stream);

/*
 * The pieces a plain-C caller needs to assemble the final stream (blocks shard
 * across the GPUs of a node with no data-path collective; only the finished streams are gathered):
 *
 *   csnappy_hip_dense_offsets   exclusive sum of d_out_len[] on the device -> the d_dense_off[] that
 *                               csnappy_hip_compact_batch takes, and the rank's byte count d_total[0]
 *                               (workspace: csnappy_hip_dense_offsets_workspace_size bytes, 8-aligned)
 *   csnappy_hip_compact_batch   (above) the rank's dense stream
 *   csnappy_hip_gather_layout   host arithmetic: from every rank's byte count (one ncclAllGather of
 *                               8 bytes per rank) each rank's offset in the assembled stream, and
 *                               its total
 *
 * and then, with the caller's RCCL communicator, one grouped exchange: ncclGroupStart; the root
 * posts ncclRecv(dst + rank_off[r], rank_bytes[r], ncclUint8, r, ...) for every peer, every peer
 * ncclSend(its dense stream, ..., root, ...); ncclGroupEnd.  INTEGRATION.md section 4 has the
 * whole sequence as a C function (tools/gather_rccl_example.c, compiled by the tests); the library
 * itself does not link RCCL -- which RCCL (the system's, or the one bundled with a framework) is
 * the caller's choice.
 */
size_t csnappy_hip_dense_offsets_workspace_size(uint32_t nblocks);
int csnappy_hip_dense_offsets(const uint32_t *d_out_len, uint32_t nblocks, uint64_t *d_dense_off,
			      uint64_t *d_total, void *d_workspace, size_t workspace_bytes, void *stream);
void csnappy_hip_gather_layout(const uint64_t *rank_bytes, uint32_t nranks, uint64_t *rank_off,
			       uint64_t *total);

/*
 * Thread safety: the batch calls keep no state between calls and may be issued from several
 * threads (each with its own buffers, workspace and, preferably, stream); the timing list below is
 * mutex-guarded.  csnappy_hip_last_error() is per thread.  The CSNAPPY_HIP_* environment knobs
 * (experiments; see csnappy_kernels.hip) are read and range-checked once, at the first batch call
 * of the process; a bad value makes every compress batch call return CSNAPPY_HIP_E_ARG.
 *
 * Per-kernel timing for bench.py: while enabled, the batch calls record a hipEvent pair on
 * `stream` around each kernel (nothing synchronises in the launch path).
 * csnappy_hip_get_kernel_timing() waits for the recorded events, returns the summed duration
 * (milliseconds) and the number of launches per kernel since the previous read, and resets.
 * slots: [0] snappy_parse_fragments (all its launches)  [1] the emit launches (snappy_emit_*)  [2] snappy_decompress_blocks
 *        [3] the stream call's index kernels (snappy_stream_*), one count per call
 */
void csnappy_hip_set_kernel_timing(int enable);
void csnappy_hip_get_kernel_timing(float ms[4], uint32_t launches[4]);

/*
 * Synthetic workloads of SURVEY.md section 8(d) (not part of the reference; bench/test input).
 * kind: 0 = G_text (URL-like tokens), 1 = G_low (runs / short periods), 2 = G_page (zram-style
 * page mix: zero / heap words / text / random).  Block i is a pure function of (kind, seed, i,
 * block_len), so any range can be generated on any rank.  The _host form fills host memory with
 * the same bytes (used by the CPU tests and the CPU baseline).
 */
int csnappy_hip_workload_generate(int kind, uint64_t seed, uint64_t first_block, uint32_t nblocks,
				  uint32_t block_len, void *d_out, void *stream);
void csnappy_workload_generate_host(int kind, uint64_t seed, uint64_t first_block,
				    uint32_t nblocks, uint32_t block_len, void *out);

#ifdef __cplusplus
}
#endif

#endif /* CSNAPPY_AMD_CSNAPPY_HIP_H_ */
