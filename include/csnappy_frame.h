/*
 * csnappy_frame.h -- the Snappy FRAMING format (stream of chunks) on top of the batch codec.
 *
 * SURVEY section 8(f) row f4.  The reference does not implement framing: it only names the goal
 * ("a streaming/framed format", reference README:11-17), so there is no reference interface to
 * mirror and no reference source to pin against.  What is implemented is google/snappy's public
 * framing_format.txt (revision of 2013-10-25, the current one):
 *
 *   stream      = stream identifier chunk, then any number of chunks
 *   chunk       = type (1 byte) | length of the rest (3 bytes, little endian) | data
 *   0xff        stream identifier, data = "sNaPpY" (6 bytes); may recur, is then skipped
 *   0x00        compressed data: masked CRC-32C of the UNCOMPRESSED bytes (4 bytes LE), then one
 *               Snappy block (the csnappy_compress format: varint length + fragments)
 *   0x01        uncompressed data: masked CRC-32C (4 bytes LE), then the bytes as they are
 *   0xfe        padding: skipped            0x80..0xfd  reserved skippable: skipped
 *   0x02..0x7f  reserved unskippable: the stream is rejected
 *   a chunk's uncompressed data is at most 65536 bytes
 *   masked crc  = ((crc >> 15) | (crc << 17)) + 0xa282ead8,  crc = CRC-32C (Castagnoli, RFC 3720)
 *
 * Each data chunk is exactly one 64 KiB block of the batch API (csnappy_hip.h, STREAM mode): the
 * writer compresses all chunks of a buffer in one batch launch and computes their CRCs with
 * snappy_crc32c_blocks; the reader decompresses all compressed chunks in one batch launch and
 * checks every CRC on the GPU.  Host buffers in, host buffers out (the chunk walk is host code);
 * csnappy_hip_crc32c_batch is the device-pointer building block.
 */
#ifndef CSNAPPY_AMD_CSNAPPY_FRAME_H_
#define CSNAPPY_AMD_CSNAPPY_FRAME_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSNAPPY_FRAME_CHUNK 65536 /* uncompressed bytes per data chunk */

#define CSNAPPY_FRAME_E_OK 0
#define CSNAPPY_FRAME_E_NO_IDENTIFIER (-201) /* the stream does not start with the identifier chunk */
#define CSNAPPY_FRAME_E_BAD_CHUNK (-202)     /* truncated chunk, reserved unskippable type, bad lengths */
#define CSNAPPY_FRAME_E_CRC (-203)           /* a chunk's checksum does not match its data */
#define CSNAPPY_FRAME_E_OUTPUT_INSUF (-204)  /* dst too small */
#define CSNAPPY_FRAME_E_DATA (-205)          /* a compressed chunk's Snappy block is malformed */
#define CSNAPPY_FRAME_E_DEVICE (-206)        /* no usable HIP device / runtime failure */

/* Largest framed size of n input bytes (identifier + per chunk 8 bytes of header/crc + payload). */
size_t csnappy_frame_max_compressed_length(size_t n);

/*
 * Frame `n` bytes: identifier chunk, then one data chunk per 65536 input bytes -- compressed
 * (table power p, 9..16) when that is smaller than the chunk, else uncompressed.  *dst_len is
 * the space available on entry (csnappy_frame_max_compressed_length(n) always suffices) and the
 * framed size on return.  An empty input gives the identifier chunk alone.
 */
int csnappy_frame_compress(const char *src, size_t n, char *dst, size_t *dst_len, int p);

/*
 * csnappy_frame_compress keeps its device buffers between calls (input, slot-strided and framed
 * output, descriptors, the codec's workspace: about 3.3 x the largest n seen plus up to 4.1 GiB of
 * workspace), on the device that was current when they were grown; a call made with another device
 * current moves them there.  This returns them to the runtime.  Calls are serialised internally.
 */
void csnappy_frame_release(void);

/* Walks the chunks (no device work): total uncompressed size of a well-formed stream. */
int csnappy_frame_uncompressed_length(const char *src, size_t n, size_t *result);

/*
 * Decode a framed stream.  *dst_len is the space available on entry and the number of bytes
 * produced on CSNAPPY_FRAME_E_OK.  Every data chunk's checksum is verified.
 */
int csnappy_frame_decompress(const char *src, size_t n, char *dst, size_t *dst_len);

/*
 * Device building block: masked CRC-32C of nblocks byte ranges (device pointers, as in
 * csnappy_hip.h): d_crc[b] = mask(crc32c(d_data + d_off[b], d_len[b])).  Enqueues on `stream`.
 */
int csnappy_hip_crc32c_batch(const void *d_data, const uint64_t *d_off, const uint32_t *d_len,
			     uint32_t nblocks, uint32_t *d_crc, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* CSNAPPY_AMD_CSNAPPY_FRAME_H_ */
