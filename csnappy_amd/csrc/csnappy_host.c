/*
 * csnappy_host.c -- the six csnappy.h entry points, in plain C, over the batched HIP layer.
 *
 * This file is what a program linked against the reference's libcsnappy.so sees.  Every codec
 * call is one H2D copy, one batch launch of the HIP kernels with nblocks = 1, and one D2H copy
 * (a whole-buffer csnappy_compress still runs its 32 KiB fragments in parallel, one wave each).
 * There is no CPU codec here: without a usable HIP device the compress calls abort() and the
 * decompress calls return CSNAPPY_E_HIP_UNAVAILABLE.
 *
 * Only the two functions that are pure integer arithmetic on a handful of bytes
 * (csnappy_max_compressed_length, csnappy_get_uncompressed_length) run on the host.
 *
 * Reference semantics kept (file:line into the reference tree):
 *   - caller owns all buffers; working_memory is accepted and ignored        csnappy.h:46-72
 *   - compress cannot fail and does not bound-check `output`                 cl_tester.c:120-165
 *   - *dst_len is updated only on CSNAPPY_E_OK                               csnappy_decompress.c:385
 *   - re-entrant: calls are serialised on one mutex-guarded device context (plumbing: one
 *     synchronous H2D -> launch -> D2H per call; the device buffers grow to the largest call and
 *     are kept for the life of the process)
 * Difference a caller can observe (also in INTEGRATION.md):
 *   - a failing HIP runtime (no device, hipMalloc) returns CSNAPPY_E_HIP_UNAVAILABLE (-100)
 */
#ifndef CSNAPPY_HOST_ARITH_ONLY /* (the sanitizer driver of the test tree compiles only the two arithmetic entry points) */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/csnappy_hip.h"
#endif

#include "../../include/csnappy.h"

/* Where every rank's compacted stream lands in the assembled one (include/csnappy_hip.h): the
 * arithmetic between the size exchange and the grouped send/recv of the gather.  Host only. */
void csnappy_hip_gather_layout(const uint64_t *rank_bytes, uint32_t nranks, uint64_t *rank_off, uint64_t *total)
{
	uint64_t run = 0;
	uint32_t r;
	for (r = 0; r < nranks; r++) {
		rank_off[r] = run;
		run += rank_bytes[r];
	}
	*total = run;
}

#ifndef CSNAPPY_HOST_ARITH_ONLY
struct buf {
	void *p;
	size_t cap;
};

static struct {
	pthread_mutex_t mu;
	int ready; /* 0 = not tried, 1 = ok, -1 = no device */
	hipStream_t stream;
	struct buf in, out, ws, desc;
} g = { PTHREAD_MUTEX_INITIALIZER, 0, 0, { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };

static int ctx_init(void)
{
	if (g.ready)
		return g.ready;
	g.ready = -1;
	if (csnappy_hip_device_count() <= 0)
		return g.ready;
	if (hipStreamCreate(&g.stream) != hipSuccess)
		return g.ready;
	g.ready = 1;
	return g.ready;
}

static int grow(struct buf *b, size_t need)
{
	if (need == 0)
		need = 256;
	if (b->cap >= need)
		return 0;
	if (b->p)
		hipFree(b->p);
	b->p = NULL;
	b->cap = 0;
	need = (need + (need >> 2) + 4095) & ~(size_t)4095;
	if (hipMalloc(&b->p, need) != hipSuccess)
		return -1;
	b->cap = need;
	return 0;
}

static void die(const char *what)
{
	fprintf(stderr, "libcsnappy (HIP): %s: %s -- no CPU fallback exists\n", what,
		csnappy_hip_last_error());
	abort();
}

/* descriptor block on the device: in_off, out_off (u64), in_len, out_len/out_cap, status, produced */
struct desc {
	uint64_t in_off, out_off;
	uint32_t in_len, out_len, out_cap;
	int32_t status;
	uint32_t produced, pad;
};

#endif /* !CSNAPPY_HOST_ARITH_ONLY */

uint32_t csnappy_max_compressed_length(uint32_t source_len)
{
	return 32u + source_len + source_len / 6u;
}

int csnappy_get_uncompressed_length(const char *start, uint32_t n, uint32_t *result)
{
	uint32_t shift = 0, k = 0;
	*result = 0;
	for (;;) {
		uint8_t c;
		if (shift >= 32 || k == n)
			return CSNAPPY_E_HEADER_BAD;
		c = (uint8_t)start[k++];
		*result |= (uint32_t)(c & 0x7f) << shift;
		if (c < 128)
			return (int)k;
		shift += 7;
	}
}

#ifndef CSNAPPY_HOST_ARITH_ONLY
static uint32_t compress_on_device(const char *input, uint32_t n, char *output, int p, int mode)
{
	struct desc d;
	size_t out_cap = 32 + (size_t)n + n / 6, ws_need;
	char *dd;
	int rc;

	pthread_mutex_lock(&g.mu);
	if (ctx_init() < 0)
		die("no usable HIP device");
	ws_need = csnappy_hip_compress_workspace_size(1, n);
	if (grow(&g.in, (size_t)n + 64) || grow(&g.out, out_cap + 64) || grow(&g.ws, ws_need) ||
	    grow(&g.desc, sizeof(d)))
		die("hipMalloc failed");
	memset(&d, 0, sizeof(d));
	d.in_len = n;
	dd = (char *)g.desc.p;
	if (hipMemcpyAsync(g.desc.p, &d, sizeof(d), hipMemcpyHostToDevice, g.stream) != hipSuccess ||
	    (n && hipMemcpyAsync(g.in.p, input, n, hipMemcpyHostToDevice, g.stream) != hipSuccess))
		die("hipMemcpy H2D failed");
	rc = csnappy_hip_compress_batch(g.in.p, (const uint64_t *)(dd + offsetof(struct desc, in_off)),
					(const uint32_t *)(dd + offsetof(struct desc, in_len)), 1, n,
					g.out.p, (const uint64_t *)(dd + offsetof(struct desc, out_off)),
					(uint32_t *)(dd + offsetof(struct desc, out_len)), p, mode, g.ws.p,
					g.ws.cap, g.stream);
	if (rc)
		die("csnappy_hip_compress_batch failed");
	if (hipMemcpyAsync(&d, g.desc.p, sizeof(d), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
	    hipStreamSynchronize(g.stream) != hipSuccess)
		die("compress kernels failed");
	if (d.out_len &&
	    hipMemcpy(output, g.out.p, d.out_len, hipMemcpyDeviceToHost) != hipSuccess)
		die("hipMemcpy D2H failed");
	pthread_mutex_unlock(&g.mu);
	return d.out_len;
}

char *csnappy_compress_fragment(const char *input, const uint32_t input_length, char *output,
				void *working_memory, const int workmem_bytes_power_of_two)
{
	(void)working_memory;
	return output + compress_on_device(input, input_length, output, workmem_bytes_power_of_two,
					   CSNAPPY_HIP_FRAGMENT);
}

void csnappy_compress(const char *input, uint32_t input_length, char *compressed,
		      uint32_t *out_compressed_length, void *working_memory,
		      const int workmem_bytes_power_of_two)
{
	(void)working_memory;
	*out_compressed_length = compress_on_device(input, input_length, compressed,
						    workmem_bytes_power_of_two, CSNAPPY_HIP_STREAM);
}

static int decompress_on_device(const char *src, uint32_t src_len, char *dst, uint32_t cap,
				uint32_t alloc, uint32_t *produced, int mode)
{
	struct desc d;
	char *dd;
	int rc, status;

	pthread_mutex_lock(&g.mu);
	if (ctx_init() < 0) {
		pthread_mutex_unlock(&g.mu);
		return CSNAPPY_E_HIP_UNAVAILABLE;
	}
	if (grow(&g.in, (size_t)src_len + 64) || grow(&g.out, (size_t)alloc + 64) ||
	    grow(&g.desc, sizeof(d))) {
		pthread_mutex_unlock(&g.mu);
		return CSNAPPY_E_HIP_UNAVAILABLE;
	}
	memset(&d, 0, sizeof(d));
	d.in_len = src_len;
	d.out_cap = cap;
	dd = (char *)g.desc.p;
	status = CSNAPPY_E_HIP_UNAVAILABLE;
	if (hipMemcpyAsync(g.desc.p, &d, sizeof(d), hipMemcpyHostToDevice, g.stream) != hipSuccess ||
	    (src_len && hipMemcpyAsync(g.in.p, src, src_len, hipMemcpyHostToDevice, g.stream) != hipSuccess))
		goto out;
	rc = csnappy_hip_decompress_batch(g.in.p, (const uint64_t *)(dd + offsetof(struct desc, in_off)),
					  (const uint32_t *)(dd + offsetof(struct desc, in_len)), 1,
					  g.out.p, (const uint64_t *)(dd + offsetof(struct desc, out_off)),
					  (const uint32_t *)(dd + offsetof(struct desc, out_cap)),
					  (int32_t *)(dd + offsetof(struct desc, status)),
					  (uint32_t *)(dd + offsetof(struct desc, produced)), mode,
					  g.stream);
	if (rc)
		goto out;
	if (hipMemcpyAsync(&d, g.desc.p, sizeof(d), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
	    hipStreamSynchronize(g.stream) != hipSuccess)
		goto out;
	/* also after -3 / -5: the reference has written the elements in front of the failing one to
	 * dst by then (csnappy_decompress.c:258-317), and so do we */
	if (d.produced && hipMemcpy(dst, g.out.p, d.produced, hipMemcpyDeviceToHost) != hipSuccess)
		goto out;
	status = d.status;
	*produced = d.produced;
out:
	pthread_mutex_unlock(&g.mu);
	return status;
}

/* csnappy_decompress of a long stream: the body goes to csnappy_hip_decompress_stream, which
 * spreads it over the device when its fragments can be told apart (include/csnappy_hip.h) */
#define STREAM_CALL_MIN_BODY (128u * 1024u)

static uint32_t expansion_bound(uint32_t n);

static int decompress_stream_on_device(const char *body, uint32_t body_len, char *dst, uint32_t room,
				       uint32_t *produced)
{
	struct desc d;
	size_t ws_need;
	/* Room beyond what the body can expand to is never used, whatever the header or the caller
	 * claims: with the room clamped to that, status and bytes are the same and the device
	 * buffers stay proportional to the input. */
	const uint32_t olen = room < expansion_bound(body_len) ? room : expansion_bound(body_len);
	const uint32_t alloc = olen;
	int status = CSNAPPY_E_HIP_UNAVAILABLE;
	char *dd;

	pthread_mutex_lock(&g.mu);
	if (ctx_init() < 0)
		goto out;
	ws_need = csnappy_hip_decompress_stream_workspace_size(body_len, olen);
	if (grow(&g.in, (size_t)body_len + 64) || grow(&g.out, (size_t)alloc + 64) ||
	    grow(&g.ws, ws_need) || grow(&g.desc, sizeof(d)))
		goto out;
	dd = (char *)g.desc.p;
	if (hipMemcpyAsync(g.in.p, body, body_len, hipMemcpyHostToDevice, g.stream) != hipSuccess)
		goto out;
	if (csnappy_hip_decompress_stream(g.in.p, body_len, olen, g.out.p,
					  (int32_t *)(dd + offsetof(struct desc, status)),
					  (uint32_t *)(dd + offsetof(struct desc, produced)), g.ws.p,
					  g.ws.cap, g.stream))
		goto out;
	if (hipMemcpyAsync(&d, g.desc.p, sizeof(d), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
	    hipStreamSynchronize(g.stream) != hipSuccess)
		goto out;
	/* also after -3 / -5: the reference has written the elements in front of the failing one to
	 * dst by then (csnappy_decompress.c:258-317), and so do we */
	if (d.produced && hipMemcpy(dst, g.out.p, d.produced, hipMemcpyDeviceToHost) != hipSuccess)
		goto out;
	status = d.status;
	*produced = d.produced;
out:
	pthread_mutex_unlock(&g.mu);
	return status;
}

/* Most bytes a Snappy body of n bytes can expand to: a 3-byte copy tag yields up to 64 bytes. */
static uint32_t expansion_bound(uint32_t n)
{
	uint64_t b = (uint64_t)n * 22 + 64;
	return b > 0xffffffffull ? 0xffffffffu : (uint32_t)b;
}

int csnappy_decompress(const char *src, uint32_t src_len, char *dst, uint32_t dst_len)
{
	uint32_t produced = 0, olen = 0, alloc;
	int hdr;
	/* header errors are decided on the host, as in the reference, before anything is allocated:
	 * -1 for an unparsable length, -2 when dst is too small (csnappy_decompress.c:399-409) */
	hdr = csnappy_get_uncompressed_length(src, src_len, &olen);
	if (hdr < 0)
		return CSNAPPY_E_HEADER_BAD;
	if (olen > dst_len)
		return CSNAPPY_E_OUTPUT_INSUF;
	if (src_len - (uint32_t)hdr >= STREAM_CALL_MIN_BODY && src_len < 0xffff0000u)
		return decompress_stream_on_device(src + hdr, src_len - (uint32_t)hdr, dst, olen, &produced);
	/* the kernel never writes past the header length, nor can the body expand past its bound */
	alloc = olen < expansion_bound(src_len) ? olen : expansion_bound(src_len);
	return decompress_on_device(src, src_len, dst, dst_len, alloc, &produced, CSNAPPY_HIP_STREAM);
}

int csnappy_decompress_noheader(const char *src, uint32_t src_len, char *dst, uint32_t *dst_len)
{
	uint32_t produced = 0;
	/* *dst_len is "space available" and may be huge: the device buffer is sized by what src_len
	 * bytes can expand to, not by it */
	uint32_t alloc = *dst_len < expansion_bound(src_len) ? *dst_len : expansion_bound(src_len);
	int rc = src_len >= STREAM_CALL_MIN_BODY && src_len < 0xffff0000u
			 ? decompress_stream_on_device(src, src_len, dst, *dst_len, &produced)
			 : decompress_on_device(src, src_len, dst, *dst_len, alloc, &produced, CSNAPPY_HIP_FRAGMENT);
	if (rc == CSNAPPY_E_OK)
		*dst_len = produced;
	return rc;
}
#endif /* !CSNAPPY_HOST_ARITH_ONLY */
