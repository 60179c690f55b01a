/*
 * snappy_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A from-scratch CPU restatement of the Snappy raw-block codec exactly as the
 * reference (zeevt/csnappy, CSNAPPY_VERSION 5) implements it, written as an
 * index-based state machine over the rules of SURVEY.md Appendix A/B/C.  It is
 * the checker that the HIP path is compared against; nothing under
 * csnappy_amd/ may include, link, dlopen or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against
 *   (1) the reference's own fixtures (testdata/urls.10K.snappy is the p=15
 *       stream, unaligned_uint64_test.snappy, baddata3.snappy), committed
 *       under tests/golden/;
 *   (2) golden digests/KATs generated from the compiled reference
 *       (oracle/_ref, tests/golden/make_golden.py);
 *   (3) when oracle/_ref/libcsnappy_ref.so is present, byte-for-byte fuzz
 *       against the compiled reference itself.
 *
 * Reference lines each function follows are cited as file:line into
 * /root/reference.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define ORC_FRAGMENT 32768u /* kBlockSize, csnappy_compress.c:85-86 */
#define ORC_MARGIN 15u      /* kInputMarginBytes, csnappy_compress.c:468 */

#define ORC_E_OK 0
#define ORC_E_HEADER_BAD (-1)
#define ORC_E_OUTPUT_INSUF (-2)
#define ORC_E_OUTPUT_OVERRUN (-3)
#define ORC_E_DATA_MALFORMED (-5)

static uint32_t rd32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
	       ((uint32_t)p[3] << 24);
}

/* csnappy_compress.c:612-616 -- uint32 arithmetic, wraps for huge inputs. */
uint32_t orc_max_compressed_length(uint32_t n)
{
	return 32u + n + n / 6u;
}

/* csnappy_compress.c:46-73 -- little-endian base-128 varint, 1..5 bytes. */
static uint32_t put_varint(uint8_t *out, uint32_t v)
{
	uint32_t k = 0;
	while (v >= 128u) {
		out[k++] = (uint8_t)(v | 128u);
		v >>= 7;
	}
	out[k++] = (uint8_t)v;
	return k;
}

/* csnappy_compress.c:228-236 */
static uint32_t hash4(uint32_t bytes, int shift)
{
	return (bytes * 0x1e35a7bdu) >> shift;
}

/* csnappy_compress.c:332-371.  Only bytes [0,end) are defined by the
 * reference (its 16-byte fast path may scribble further); we write exactly. */
static uint32_t put_literal(uint8_t *out, uint32_t op, const uint8_t *src, uint32_t len)
{
	uint32_t n = len - 1;
	if (n < 60) {
		out[op++] = (uint8_t)(n << 2);
	} else {
		uint32_t tagpos = op++, count = 0;
		while (n > 0) {
			out[op++] = (uint8_t)(n & 0xff);
			n >>= 8;
			count++;
		}
		out[tagpos] = (uint8_t)((59 + count) << 2);
	}
	memcpy(out + op, src, len);
	return op + len;
}

/* csnappy_compress.c:373-393 */
static uint32_t put_copy_piece(uint8_t *out, uint32_t op, uint32_t offset, uint32_t len)
{
	if (len < 12 && offset < 2048) {
		out[op++] = (uint8_t)(1 + ((len - 4) << 2) + ((offset >> 8) << 5));
		out[op++] = (uint8_t)(offset & 0xff);
	} else {
		out[op++] = (uint8_t)(2 + ((len - 1) << 2));
		out[op++] = (uint8_t)(offset & 0xff);
		out[op++] = (uint8_t)(offset >> 8);
	}
	return op;
}

/* csnappy_compress.c:395-415 */
static uint32_t put_copy(uint8_t *out, uint32_t op, uint32_t offset, uint32_t len)
{
	while (len >= 68) {
		op = put_copy_piece(out, op, offset, 64);
		len -= 64;
	}
	if (len > 64) {
		op = put_copy_piece(out, op, offset, 60);
		len -= 60;
	}
	return put_copy_piece(out, op, offset, len);
}

/* Longest common prefix of F[a..] and F[b..], b > a, limited by n - b.
 * csnappy_compress.c:252-295 (the 8-byte stride there is an implementation
 * detail; the value returned is exactly this). */
static uint32_t lcp(const uint8_t *F, uint32_t a, uint32_t b, uint32_t n)
{
	uint32_t m = 0;
	while (b + m < n && F[a + m] == F[b + m])
		m++;
	return m;
}

/*
 * csnappy_compress.c:469-606, restated per SURVEY.md Appendix B rules 2-9.
 * Returns the number of bytes written to out (the reference returns the end
 * pointer).  p = workmem_bytes_power_of_two, 9..16 (16 is what cl_tester
 * passes although the header documents 15 as the maximum).
 */
uint32_t orc_compress_fragment(const uint8_t *F, uint32_t n, uint8_t *out, int p)
{
	static __thread uint16_t T[32768];
	const int shift = 33 - p;
	uint32_t op = 0, next_emit = 0, ip, ip_limit, skip, next_ip, cand, h;

	if (n < ORC_MARGIN) /* rule 2 */
		goto remainder;
	memset(T, 0, (size_t)1 << p); /* rule 3: empty slot == position 0 */
	ip_limit = n - ORC_MARGIN;
	ip = 1; /* rule 4 */
	for (;;) {
		/* rule 5: scan with the skip heuristic */
		skip = 32;
		next_ip = ip;
		for (;;) {
			ip = next_ip;
			next_ip = ip + (skip >> 5);
			skip++;
			if (next_ip > ip_limit)
				goto remainder;
			h = hash4(rd32(F + ip), shift);
			cand = T[h];
			T[h] = (uint16_t)ip;
			if (rd32(F + ip) == rd32(F + cand))
				break;
		}
		/* rule 6 */
		op = put_literal(out, op, F + next_emit, ip - next_emit);
		/* rule 7: copy, then try for back-to-back copies */
		for (;;) {
			uint32_t base = ip;
			uint32_t m = 4 + lcp(F, cand + 4, ip + 4, n);
			ip += m;
			op = put_copy(out, op, base - cand, m);
			next_emit = ip;
			if (ip >= ip_limit)
				goto remainder;
			T[hash4(rd32(F + ip - 1), shift)] = (uint16_t)(ip - 1);
			h = hash4(rd32(F + ip), shift);
			cand = T[h];
			T[h] = (uint16_t)ip;
			if (rd32(F + ip) != rd32(F + cand))
				break;
		}
		ip++; /* rule 8 */
	}
remainder: /* rule 9 */
	if (next_emit < n)
		op = put_literal(out, op, F + next_emit, n - next_emit);
	return op;
}

/* Table power csnappy_compress picks for a fragment of n bytes when the
 * caller passes p.  csnappy_compress.c:638-646. */
int orc_fragment_table_power(uint32_t n, int p)
{
	int ws = p;
	if (n < ORC_FRAGMENT) {
		for (ws = 9; ws < p; ++ws)
			if ((1u << (ws - 1)) >= n)
				break;
	}
	return ws;
}

/* csnappy_compress.c:621-656 */
void orc_compress(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t *out_len, int p)
{
	uint32_t written = put_varint(out, n);
	while (n > 0) {
		uint32_t take = n < ORC_FRAGMENT ? n : ORC_FRAGMENT;
		written += orc_compress_fragment(in, take, out + written,
						 orc_fragment_table_power(take, p));
		in += take;
		n -= take;
	}
	*out_len = written;
}

/* csnappy_decompress.c:45-71.  *result is clobbered progressively, even on
 * error, as in the reference. */
int orc_get_uncompressed_length(const uint8_t *src, uint32_t n, uint32_t *result)
{
	uint32_t shift = 0, k = 0;
	*result = 0;
	for (;;) {
		uint8_t c;
		if (shift >= 32 || k == n)
			return ORC_E_HEADER_BAD;
		c = src[k++];
		*result |= (uint32_t)(c & 0x7f) << shift;
		if (c < 128)
			return (int)k;
		shift += 7;
	}
}

/*
 * csnappy_decompress.c:319-387 with SAW__* (:264-317), restated per
 * SURVEY.md Appendix C.  Differences that are deliberate:
 *   - bytes past `produced` are never written (the reference's fast paths may
 *     scribble up to 16 bytes further inside the limit; those bytes are
 *     unspecified there);
 *   - a tag whose extra bytes are cut off by the end of input returns -5 (the
 *     reference reads stale stack bytes there: undefined, excluded from
 *     parity).
 */
int orc_decompress_noheader(const uint8_t *src, uint32_t n, uint8_t *dst, uint32_t *dst_len)
{
	const uint32_t limit = *dst_len;
	uint32_t ip = 0, op = 0;
	while (ip < n) {
		uint32_t tag = src[ip++];
		uint32_t kind = tag & 3, len, extra, val = 0, i;
		if (kind) {
			uint32_t offset;
			if (kind == 1) {
				len = 4 + ((tag >> 2) & 7);
				extra = 1;
			} else {
				len = (tag >> 2) + 1;
				extra = kind == 2 ? 2 : 4;
			}
			if (n - ip < extra)
				return ORC_E_DATA_MALFORMED;
			for (i = 0; i < extra; i++)
				val |= (uint32_t)src[ip + i] << (8 * i);
			ip += extra;
			offset = kind == 1 ? ((tag >> 5) << 8) | val : val;
			if (offset == 0 || offset > op) /* :301-303, checked first */
				return ORC_E_DATA_MALFORMED;
			if (limit - op < len) /* :311-312 */
				return ORC_E_OUTPUT_OVERRUN;
			for (i = 0; i < len; i++) /* :188-206 semantics */
				dst[op + i] = dst[op + i - offset];
			op += len;
		} else {
			len = (tag >> 2) + 1;
			if (len > 60) { /* :368-373 */
				extra = len - 60;
				if (n - ip < extra)
					return ORC_E_DATA_MALFORMED;
				for (i = 0; i < extra; i++)
					val |= (uint32_t)src[ip + i] << (8 * i);
				ip += extra;
				len = val + 1; /* wraps to 0 for ff ff ff ff, as there */
			}
			/* :374-375: `available < (int32_t)length`; a length that is
			 * negative as int32 skips the -5 test and ends as -3. */
			if ((int32_t)len >= 0 && n - ip < len)
				return ORC_E_DATA_MALFORMED;
			if (limit - op < len) /* :288-289 / :274-275 */
				return ORC_E_OUTPUT_OVERRUN;
			if ((int32_t)len < 0)
				return ORC_E_DATA_MALFORMED; /* unreachable below 2 GiB limits */
			memcpy(dst + op, src + ip, len);
			ip += len;
			op += len;
		}
	}
	*dst_len = op;
	return ORC_E_OK;
}

/* csnappy_decompress.c:394-411: header, -2 check, then the tag loop with the
 * HEADER length as the output limit; produced length is not compared. */
int orc_decompress(const uint8_t *src, uint32_t n, uint8_t *dst, uint32_t dst_len)
{
	uint32_t olen = 0;
	int hdr = orc_get_uncompressed_length(src, n, &olen);
	if (hdr < 0)
		return hdr;
	if (olen > dst_len)
		return ORC_E_OUTPUT_INSUF;
	return orc_decompress_noheader(src + hdr, n - (uint32_t)hdr, dst, &olen);
}

/* ------------------------------------------------------------------------
 * Batch drivers used by the tests and by bench.py's cpu_baseline leg.  They
 * take the codec entry points as function pointers so the same loop can time
 * this restatement ("port") or the compiled reference in oracle/_ref
 * ("reference").  mode: 0 = STREAM (csnappy_compress / csnappy_decompress),
 * 1 = FRAGMENT (csnappy_compress_fragment / csnappy_decompress_noheader).
 * ---------------------------------------------------------------------- */
typedef void (*compress_fn)(const char *, uint32_t, char *, uint32_t *, void *, int);
typedef char *(*fragment_fn)(const char *, uint32_t, char *, void *, int);
typedef int (*decompress_fn)(const char *, uint32_t, char *, uint32_t);
typedef int (*noheader_fn)(const char *, uint32_t, char *, uint32_t *);

static void port_compress(const char *in, uint32_t n, char *out, uint32_t *olen, void *wm, int p)
{
	(void)wm;
	orc_compress((const uint8_t *)in, n, (uint8_t *)out, olen, p);
}

static char *port_fragment(const char *in, uint32_t n, char *out, void *wm, int p)
{
	(void)wm;
	return out + orc_compress_fragment((const uint8_t *)in, n, (uint8_t *)out, p);
}

static int port_decompress(const char *src, uint32_t n, char *dst, uint32_t dst_len)
{
	return orc_decompress((const uint8_t *)src, n, (uint8_t *)dst, dst_len);
}

static int port_noheader(const char *src, uint32_t n, char *dst, uint32_t *dst_len)
{
	return orc_decompress_noheader((const uint8_t *)src, n, (uint8_t *)dst, dst_len);
}

void orc_batch_compress(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len,
			uint32_t first, uint32_t last, uint8_t *out, const uint64_t *out_off,
			uint32_t *out_len, int p, int mode, void *cfn, void *ffn)
{
	compress_fn c = cfn ? (compress_fn)cfn : port_compress;
	fragment_fn f = ffn ? (fragment_fn)ffn : port_fragment;
	void *wm = malloc(65536);
	uint32_t b;
	for (b = first; b < last; b++) {
		const char *src = (const char *)in + in_off[b];
		char *dst = (char *)out + out_off[b];
		if (mode == 0)
			c(src, in_len[b], dst, &out_len[b], wm, p);
		else
			out_len[b] = (uint32_t)(f(src, in_len[b], dst, wm, p) - dst);
	}
	free(wm);
}

void orc_batch_decompress(const uint8_t *in, const uint64_t *in_off, const uint32_t *in_len,
			  uint32_t first, uint32_t last, uint8_t *out, const uint64_t *out_off,
			  const uint32_t *out_cap, int32_t *status, uint32_t *produced, int mode,
			  void *dfn, void *nfn)
{
	decompress_fn d = dfn ? (decompress_fn)dfn : port_decompress;
	noheader_fn nh = nfn ? (noheader_fn)nfn : port_noheader;
	uint32_t b;
	for (b = first; b < last; b++) {
		const char *src = (const char *)in + in_off[b];
		char *dst = (char *)out + out_off[b];
		if (mode == 0) {
			status[b] = d(src, in_len[b], dst, out_cap[b]);
			produced[b] = 0; /* csnappy_decompress does not report it */
		} else {
			uint32_t len = out_cap[b];
			status[b] = nh(src, in_len[b], dst, &len);
			produced[b] = status[b] == 0 ? len : 0;
		}
	}
}
