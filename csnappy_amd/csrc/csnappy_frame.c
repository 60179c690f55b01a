/*
 * csnappy_frame.c -- Snappy framing format (include/csnappy_frame.h) in plain C over the batch
 * C-ABI.  The chunk walk is host code; the codec work (one 64 KiB STREAM block per data chunk)
 * and the checksums (snappy_crc32c_blocks) run on the GPU, one batch launch each per call.
 * There is no CPU codec here: without a HIP device the calls return CSNAPPY_FRAME_E_DEVICE.
 *
 * Spec followed: google/snappy framing_format.txt (sections 2-4); the reference tree holds no
 * framing code (reference README:11-17 lists it as a goal only).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/csnappy.h"
#include "../../include/csnappy_frame.h"
#include "../../include/csnappy_hip.h"

static const unsigned char kStreamId[10] = { 0xff, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y' };
#define SLOT 76544u /* >= csnappy_max_compressed_length(65536) = 76490, 64-byte multiple */

struct dev {
	void *p[8];
	int n;
};

static void *dalloc(struct dev *d, size_t bytes)
{
	void *p = NULL;
	if (d->n >= 8 || hipMalloc(&p, bytes ? bytes : 256) != hipSuccess)
		return NULL;
	d->p[d->n++] = p;
	return p;
}

static void dfree(struct dev *d)
{
	while (d->n > 0)
		(void)hipFree(d->p[--d->n]);
}

/* csnappy_frame_compress keeps its device buffers between calls, like the legacy calls' context
 * (csnappy_host.c): they grow to the largest call and live as long as the process; calls are
 * serialised on the mutex.  (Allocating the GiBs of a large call's workspace anew each time, and
 * fetching the slot-strided output whole, made the call six times slower than csnappy_compress.) */
enum { FB_IN, FB_OUT, FB_OFF, FB_LEN, FB_WS, FB_FRAMED, FB_COUNT };
static struct {
	void *p;
	size_t cap;
	int dev; /* the device the buffer lives on */
} g_fb[FB_COUNT];
static pthread_mutex_t g_fb_mu = PTHREAD_MUTEX_INITIALIZER;

/* (g_fb_mu held) a buffer of >= bytes on the CALLING thread's current device: one that was grown by a
 * thread with another device current is released and allocated anew here */
static void *fb_get(int slot, size_t bytes)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return NULL;
	if (g_fb[slot].cap < bytes || g_fb[slot].dev != dev) {
		void *p = NULL;
		if (g_fb[slot].p) {
			int back = dev;
			(void)hipSetDevice(g_fb[slot].dev);
			(void)hipFree(g_fb[slot].p);
			(void)hipSetDevice(back);
		}
		g_fb[slot].p = NULL;
		g_fb[slot].cap = 0;
		if (hipMalloc(&p, bytes) != hipSuccess)
			return NULL;
		g_fb[slot].p = p;
		g_fb[slot].cap = bytes;
		g_fb[slot].dev = dev;
	}
	return g_fb[slot].p;
}

void csnappy_frame_release(void)
{
	int slot, back = 0;
	pthread_mutex_lock(&g_fb_mu);
	(void)hipGetDevice(&back);
	for (slot = 0; slot < FB_COUNT; slot++) {
		if (g_fb[slot].p) {
			(void)hipSetDevice(g_fb[slot].dev);
			(void)hipFree(g_fb[slot].p);
		}
		g_fb[slot].p = NULL;
		g_fb[slot].cap = 0;
	}
	(void)hipSetDevice(back);
	pthread_mutex_unlock(&g_fb_mu);
}

static uint32_t rd_le24(const unsigned char *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

static uint32_t rd_le32(const unsigned char *p)
{
	return rd_le24(p) | ((uint32_t)p[3] << 24);
}

static void wr_chunk_header(unsigned char *q, unsigned type, uint32_t len, uint32_t crc)
{
	q[0] = (unsigned char)type;
	q[1] = (unsigned char)len;
	q[2] = (unsigned char)(len >> 8);
	q[3] = (unsigned char)(len >> 16);
	q[4] = (unsigned char)crc;
	q[5] = (unsigned char)(crc >> 8);
	q[6] = (unsigned char)(crc >> 16);
	q[7] = (unsigned char)(crc >> 24);
}

size_t csnappy_frame_max_compressed_length(size_t n)
{
	const size_t chunks = (n + CSNAPPY_FRAME_CHUNK - 1) / CSNAPPY_FRAME_CHUNK;
	return sizeof(kStreamId) + chunks * 8 + n;
}

int csnappy_frame_compress(const char *src, size_t n, char *dst, size_t *dst_len, int p)
{
	const size_t chunks = (n + CSNAPPY_FRAME_CHUNK - 1) / CSNAPPY_FRAME_CHUNK;
	uint64_t *off = NULL;
	uint32_t *len = NULL;
	unsigned char *q = (unsigned char *)dst;
	size_t i, ws, need = sizeof(kStreamId);
	int rc = CSNAPPY_FRAME_E_DEVICE;
	void *d_in, *d_out, *d_off, *d_len, *d_ws, *d_framed;

	if (p < 9 || p > 16 || chunks > 0x7fffffffu)
		return CSNAPPY_FRAME_E_BAD_CHUNK;
	if (*dst_len < sizeof(kStreamId))
		return CSNAPPY_FRAME_E_OUTPUT_INSUF;
	memcpy(q, kStreamId, sizeof(kStreamId));
	if (chunks == 0) {
		*dst_len = sizeof(kStreamId);
		return CSNAPPY_FRAME_E_OK;
	}
	if (csnappy_hip_device_count() <= 0)
		return CSNAPPY_FRAME_E_DEVICE;
	/* descriptors: in_off | out_off | framed_off (u64), in_len | out_len | crc | compressed_len |
	 * stored_len (u32) */
	off = malloc(3 * chunks * sizeof(uint64_t));
	len = malloc(5 * chunks * sizeof(uint32_t));
	if (!off || !len) {
		free(off);
		free(len);
		return rc;
	}
	for (i = 0; i < chunks; i++) {
		off[i] = (uint64_t)i * CSNAPPY_FRAME_CHUNK;
		off[chunks + i] = (uint64_t)i * SLOT;
		len[i] = (uint32_t)(n - i * CSNAPPY_FRAME_CHUNK < CSNAPPY_FRAME_CHUNK ? n - i * CSNAPPY_FRAME_CHUNK
										       : CSNAPPY_FRAME_CHUNK);
	}
	pthread_mutex_lock(&g_fb_mu);
	ws = csnappy_hip_compress_workspace_size((uint32_t)chunks, CSNAPPY_FRAME_CHUNK);
	d_in = fb_get(FB_IN, n + 64);
	d_out = fb_get(FB_OUT, chunks * (size_t)SLOT);
	d_off = fb_get(FB_OFF, 3 * chunks * sizeof(uint64_t));
	d_len = fb_get(FB_LEN, 5 * chunks * sizeof(uint32_t));
	d_ws = fb_get(FB_WS, ws);
	d_framed = fb_get(FB_FRAMED, csnappy_frame_max_compressed_length(n) + 64);
	if (!d_in || !d_out || !d_off || !d_len || !d_ws || !d_framed)
		goto done;
	if (hipMemcpy(d_in, src, n, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_off, off, 2 * chunks * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_len, len, chunks * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
		goto done;
	/* all chunks: one compress batch (each chunk = one csnappy_compress block) + one CRC batch */
	if (csnappy_hip_compress_batch(d_in, (uint64_t *)d_off, (uint32_t *)d_len, (uint32_t)chunks,
				       CSNAPPY_FRAME_CHUNK, d_out, (uint64_t *)d_off + chunks,
				       (uint32_t *)d_len + chunks, p, CSNAPPY_HIP_STREAM, d_ws, ws, NULL) ||
	    csnappy_hip_crc32c_batch(d_in, (uint64_t *)d_off, (uint32_t *)d_len, (uint32_t)chunks,
				     (uint32_t *)d_len + 2 * chunks, NULL))
		goto done;
	if (hipDeviceSynchronize() != hipSuccess ||
	    hipMemcpy(len + chunks, (uint32_t *)d_len + chunks, 2 * chunks * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
		goto done;
	/* lay the stream out: a chunk is stored compressed only when that is smaller */
	for (i = 0; i < chunks; i++) {
		const uint32_t ilen = len[i], olen = len[chunks + i];
		const int comp = olen < ilen;
		const uint32_t body = comp ? olen : ilen;
		need += 8 + (size_t)body;
		off[2 * chunks + i] = need - body; /* where the chunk's body goes */
		len[3 * chunks + i] = comp ? olen : 0;
		len[4 * chunks + i] = comp ? 0 : ilen;
	}
	if (need > *dst_len) {
		rc = CSNAPPY_FRAME_E_OUTPUT_INSUF;
		goto done;
	}
	/* the bodies are put at their places on the device (compressed ones from the batch's slots,
	 * stored ones from the input), the stream comes back in one copy, the 8-byte chunk headers are
	 * written over the gaps here */
	if (hipMemcpy((uint64_t *)d_off + 2 * chunks, off + 2 * chunks, chunks * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy((uint32_t *)d_len + 3 * chunks, len + 3 * chunks, 2 * chunks * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
	    csnappy_hip_compact_batch(d_out, (uint64_t *)d_off + chunks, (uint32_t *)d_len + 3 * chunks,
				      (uint64_t *)d_off + 2 * chunks, (uint32_t)chunks, d_framed, NULL) ||
	    csnappy_hip_compact_batch(d_in, (uint64_t *)d_off, (uint32_t *)d_len + 4 * chunks,
				      (uint64_t *)d_off + 2 * chunks, (uint32_t)chunks, d_framed, NULL) ||
	    hipDeviceSynchronize() != hipSuccess ||
	    hipMemcpy(dst, d_framed, need, hipMemcpyDeviceToHost) != hipSuccess)
		goto done;
	memcpy(q, kStreamId, sizeof(kStreamId));
	for (i = 0; i < chunks; i++) {
		const int comp = len[3 * chunks + i] != 0 || len[i] == 0;
		const uint32_t body = comp ? len[3 * chunks + i] : len[4 * chunks + i];
		wr_chunk_header((unsigned char *)dst + off[2 * chunks + i] - 8, comp ? 0x00 : 0x01, 4 + body, len[2 * chunks + i]);
	}
	*dst_len = need;
	rc = CSNAPPY_FRAME_E_OK;
done:
	pthread_mutex_unlock(&g_fb_mu);
	free(off);
	free(len);
	return rc;
}

/* One data chunk found by the walk. */
struct piece {
	uint64_t src_off;  /* body (compressed block, or raw bytes) in the stream */
	uint32_t src_len;
	uint32_t ulen;     /* uncompressed bytes */
	uint32_t crc;      /* masked CRC stored in the chunk */
	int compressed;
};

/* Walks a framed stream; calls back for every data chunk.  Returns 0 or a CSNAPPY_FRAME_E_ code. */
static int walk(const unsigned char *s, size_t n, struct piece **pieces, size_t *npieces, size_t *total)
{
	size_t pos = 0, cap = 0, cnt = 0, sum = 0;
	struct piece *v = NULL;
	if (n < sizeof(kStreamId) || memcmp(s, kStreamId, sizeof(kStreamId)))
		return CSNAPPY_FRAME_E_NO_IDENTIFIER;
	while (pos < n) {
		unsigned type;
		uint32_t len;
		const unsigned char *d;
		if (n - pos < 4)
			goto bad;
		type = s[pos];
		len = rd_le24(s + pos + 1);
		if (n - pos - 4 < len)
			goto bad;
		d = s + pos + 4;
		pos += 4 + (size_t)len;
		if (type == 0xff) {
			if (len != 6 || memcmp(d, kStreamId + 4, 6))
				goto bad;
			continue;
		}
		if (type >= 0x80)
			continue; /* padding (0xfe) and reserved skippable chunks */
		if (type > 0x01)
			goto bad; /* reserved unskippable */
		if (len < 4)
			goto bad;
		if (cnt == cap) {
			struct piece *nv = realloc(v, (cap = cap ? 2 * cap : 64) * sizeof(*v));
			if (!nv)
				goto bad;
			v = nv;
		}
		v[cnt].crc = rd_le32(d);
		v[cnt].src_off = (uint64_t)(d + 4 - s);
		v[cnt].src_len = len - 4;
		v[cnt].compressed = type == 0x00;
		if (type == 0x00) {
			uint32_t ulen = 0;
			if (csnappy_get_uncompressed_length((const char *)d + 4, len - 4, &ulen) < 0)
				goto bad;
			v[cnt].ulen = ulen;
		} else {
			v[cnt].ulen = len - 4;
		}
		if (v[cnt].ulen > CSNAPPY_FRAME_CHUNK)
			goto bad; /* framing_format.txt 4.2 / 4.3: at most 65536 uncompressed bytes */
		sum += v[cnt].ulen;
		cnt++;
	}
	*pieces = v;
	*npieces = cnt;
	*total = sum;
	return CSNAPPY_FRAME_E_OK;
bad:
	free(v);
	return CSNAPPY_FRAME_E_BAD_CHUNK;
}

int csnappy_frame_uncompressed_length(const char *src, size_t n, size_t *result)
{
	struct piece *v = NULL;
	size_t cnt = 0;
	int rc = walk((const unsigned char *)src, n, &v, &cnt, result);
	if (rc == CSNAPPY_FRAME_E_OK)
		free(v);
	return rc;
}

int csnappy_frame_decompress(const char *src, size_t n, char *dst, size_t *dst_len)
{
	struct piece *v = NULL;
	size_t cnt = 0, total = 0, i, nc = 0, nr = 0, pos;
	struct dev D = { { 0 }, 0 };
	uint64_t *off = NULL;
	uint32_t *u32 = NULL;
	int rc = walk((const unsigned char *)src, n, &v, &cnt, &total);
	void *d_src, *d_out, *d_off, *d_u32;

	if (rc != CSNAPPY_FRAME_E_OK)
		return rc;
	if (total > *dst_len) {
		free(v);
		return CSNAPPY_FRAME_E_OUTPUT_INSUF;
	}
	if (cnt == 0) {
		free(v);
		*dst_len = 0;
		return CSNAPPY_FRAME_E_OK;
	}
	rc = CSNAPPY_FRAME_E_DEVICE;
	if (cnt > 0x7fffffffu || csnappy_hip_device_count() <= 0)
		goto done;
	for (i = 0; i < cnt; i++)
		if (v[i].compressed)
			nc++;
	nr = cnt - nc;
	/* descriptors.  Compressed chunks first (decompress batch + CRC over the output), raw chunks
	 * after them (CRC over the stream bytes):
	 *   off: [0,cnt) input offsets (into the stream), [cnt,2cnt) output offsets
	 *   u32: [0,cnt) input lengths, [cnt,2cnt) uncompressed lengths (= out_cap),
	 *        [2cnt,3cnt) status, [3cnt,4cnt) produced, [4cnt,5cnt) computed masked CRCs */
	off = malloc(2 * cnt * sizeof(uint64_t));
	u32 = malloc(5 * cnt * sizeof(uint32_t));
	if (!off || !u32)
		goto done;
	{
		size_t ic = 0, ir = nc;
		for (i = 0, pos = 0; i < cnt; i++) {
			const size_t k = v[i].compressed ? ic++ : ir++;
			off[k] = v[i].src_off;
			off[cnt + k] = pos;
			u32[k] = v[i].src_len;
			u32[cnt + k] = v[i].ulen;
			pos += v[i].ulen;
		}
	}
	d_src = dalloc(&D, n + 64);
	d_out = dalloc(&D, total + 64);
	d_off = dalloc(&D, 2 * cnt * sizeof(uint64_t));
	d_u32 = dalloc(&D, 5 * cnt * sizeof(uint32_t));
	if (!d_src || !d_out || !d_off || !d_u32)
		goto done;
	if (hipMemcpy(d_src, src, n, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_off, off, 2 * cnt * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_u32, u32, 2 * cnt * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
		goto done;
	if (nc && (csnappy_hip_decompress_batch(d_src, (uint64_t *)d_off, (uint32_t *)d_u32, (uint32_t)nc, d_out,
						(uint64_t *)d_off + cnt, (uint32_t *)d_u32 + cnt,
						(int32_t *)((uint32_t *)d_u32 + 2 * cnt), (uint32_t *)d_u32 + 3 * cnt,
						CSNAPPY_HIP_STREAM, NULL) ||
		   csnappy_hip_crc32c_batch(d_out, (uint64_t *)d_off + cnt, (uint32_t *)d_u32 + cnt, (uint32_t)nc,
					    (uint32_t *)d_u32 + 4 * cnt, NULL)))
		goto done;
	if (nr && csnappy_hip_crc32c_batch(d_src, (uint64_t *)d_off + nc, (uint32_t *)d_u32 + nc, (uint32_t)nr,
					   (uint32_t *)d_u32 + 4 * cnt + nc, NULL))
		goto done;
	if (hipDeviceSynchronize() != hipSuccess ||
	    hipMemcpy(u32 + 2 * cnt, (uint32_t *)d_u32 + 2 * cnt, 3 * cnt * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
	    (total && hipMemcpy(dst, d_out, total, hipMemcpyDeviceToHost) != hipSuccess))
		goto done;
	{
		size_t ic = 0, ir = nc;
		for (i = 0; i < cnt; i++) {
			const size_t k = v[i].compressed ? ic++ : ir++;
			if (v[i].compressed) {
				if ((int32_t)u32[2 * cnt + k] != CSNAPPY_E_OK || u32[3 * cnt + k] != v[i].ulen) {
					rc = CSNAPPY_FRAME_E_DATA;
					goto done;
				}
			} else {
				memcpy(dst + off[cnt + k], src + v[i].src_off, v[i].ulen);
			}
			if (u32[4 * cnt + k] != v[i].crc) {
				rc = CSNAPPY_FRAME_E_CRC;
				goto done;
			}
		}
	}
	*dst_len = total;
	rc = CSNAPPY_FRAME_E_OK;
done:
	dfree(&D);
	free(off);
	free(u32);
	free(v);
	return rc;
}
