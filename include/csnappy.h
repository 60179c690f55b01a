/*
 * csnappy.h -- the drop-in boundary.
 *
 * These are the six entry points of zeevt/csnappy's public header, with the same C signatures,
 * the same macros and the same return codes, implemented by the MI355X-native codec in
 * libcsnappy.so (csnappy_amd/csrc).  A program written against the reference header links
 * against this library unchanged.  Each declaration cites the reference interface it replaces
 * (file:line into the reference tree).
 *
 * Every call here is executed by the HIP kernels (one H2D copy, one batch launch, one D2H
 * copy); there is no CPU codec in the library.  If no HIP device is usable the compress
 * entry points abort() with a message on stderr and the decompress entry points return
 * CSNAPPY_E_HIP_UNAVAILABLE; they never fall back to host code.  For throughput use the batched API in
 * csnappy_hip.h; these single-buffer calls are plumbing.
 */
#ifndef CSNAPPY_AMD_CSNAPPY_H_
#define CSNAPPY_AMD_CSNAPPY_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference csnappy.h:11-14 */
#define CSNAPPY_VERSION 5
#define CSNAPPY_WORKMEM_BYTES_POWER_OF_TWO 16
#define CSNAPPY_WORKMEM_BYTES (1 << CSNAPPY_WORKMEM_BYTES_POWER_OF_TWO)

/* reference csnappy.h:124-129 */
#define CSNAPPY_E_OK 0
#define CSNAPPY_E_HEADER_BAD (-1)
#define CSNAPPY_E_OUTPUT_INSUF (-2)
#define CSNAPPY_E_OUTPUT_OVERRUN (-3)
#define CSNAPPY_E_INPUT_NOT_CONSUMED (-4)
#define CSNAPPY_E_DATA_MALFORMED (-5)
/* Not in the reference: the decompress calls return this when no HIP device can be used. */
#define CSNAPPY_E_HIP_UNAVAILABLE (-100)

/*
 * reference csnappy.h:30-31, csnappy_compress.c:612-616.
 * 32 + n + n/6 in uint32 arithmetic (wraps for n near 4 GiB, as the reference does).
 */
uint32_t csnappy_max_compressed_length(uint32_t source_len);

/*
 * reference csnappy.h:46-52, csnappy_compress.c:469-606.
 * Compresses one fragment (input_length <= 32768) without the length prefix and returns the end
 * pointer into `output`.  `output` must hold csnappy_max_compressed_length(input_length) bytes
 * (not checked, as in the reference).  `working_memory` is accepted and ignored: the hash table
 * lives in LDS.  `workmem_bytes_power_of_two` (9..16) is honoured because it changes the bytes.
 */
char *csnappy_compress_fragment(const char *input, const uint32_t input_length, char *output,
				void *working_memory, const int workmem_bytes_power_of_two);

/*
 * reference csnappy.h:65-72, csnappy_compress.c:621-656.
 * varint(input_length) followed by the fragments; *out_compressed_length receives the size.
 */
void csnappy_compress(const char *input, uint32_t input_length, char *compressed,
		      uint32_t *out_compressed_length, void *working_memory,
		      const int workmem_bytes_power_of_two);

/*
 * reference csnappy.h:83-87, csnappy_decompress.c:45-71.
 * Returns the number of header bytes (1..5) or CSNAPPY_E_HEADER_BAD; *result is written
 * progressively and is clobbered even on error.  Pure host arithmetic (no device work).
 */
int csnappy_get_uncompressed_length(const char *start, uint32_t n, uint32_t *result);

/*
 * reference csnappy.h:99-104, csnappy_decompress.c:394-411.
 * Header parse (-1), header length > dst_len (-2), then the tag loop with the header length as
 * the output limit.  The produced length is not compared with the header (reference behaviour).
 */
int csnappy_decompress(const char *src, uint32_t src_len, char *dst, uint32_t dst_len);

/*
 * reference csnappy.h:114-119, csnappy_decompress.c:319-387.
 * *dst_len is the space available on entry and the produced length on CSNAPPY_E_OK (untouched
 * on error).
 */
int csnappy_decompress_noheader(const char *src, uint32_t src_len, char *dst, uint32_t *dst_len);

#ifdef __cplusplus
}
#endif

#endif /* CSNAPPY_AMD_CSNAPPY_H_ */
