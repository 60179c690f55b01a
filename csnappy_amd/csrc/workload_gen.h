/*
 * workload_gen.h -- synthetic inputs for the benchmark configurations of SURVEY.md section 8(d).
 *
 * Not part of the reference (it ships no generator); this is bench/test input.  One source,
 * compiled twice: by hipcc for the device generator kernel and by the host compiler for the
 * CPU tests, so both produce the same bytes.  Integer arithmetic only.  Block i depends only on
 * (kind, seed, i, block_len): any block range can be produced on any rank without
 * communication.
 *
 * The recipes are FROZEN (golden ratios are pinned in tests/golden/golden.json); changing
 * anything here changes every benchmark input.
 */
#ifndef CSNAPPY_AMD_WORKLOAD_GEN_H_
#define CSNAPPY_AMD_WORKLOAD_GEN_H_

#include <stdint.h>

#ifdef __HIPCC__
#define WG_HD __host__ __device__ static inline
#else
#define WG_HD static inline
#endif

#define WG_TEXT 0
#define WG_LOW 1
#define WG_PAGE 2

#define WG_DICT 2048

/* splitmix64 */
WG_HD uint64_t wg_next(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

WG_HD uint64_t wg_block_state(uint64_t seed, uint64_t block)
{
	uint64_t s = seed ^ block;
	return wg_next(&s);
}

/* Token k of the shared dictionary: 3..12 lowercase letters, a pure function of (seed, k). */
WG_HD uint32_t wg_token(uint64_t seed, uint32_t k, uint8_t *dst12)
{
	uint64_t s = seed * 0xD1342543DE82EF95ull + k;
	uint64_t a = wg_next(&s), b = wg_next(&s);
	uint32_t len = 3 + (uint32_t)(a % 10), i;
	for (i = 0; i < len; i++) {
		dst12[i] = (uint8_t)('a' + (b % 26));
		b /= 26;
	}
	return len;
}

/* Zipf-like rank draw over WG_DICT tokens: rank = floor(2048^(u)) - 1 with u uniform in [0,1)
 * has density ~ 1/rank; computed as a product of precomputed 2048^(bit/2^k) factors in 16.16
 * fixed point so host and device agree exactly. */
WG_HD uint32_t wg_zipf(uint32_t r)
{
	/* 2048^(1/2), ^(1/4), ... ^(1/256) in 16.16 */
	const uint32_t f[8] = { 2965821u, 440872u, 169983u, 105550u, 83171u, 73825u, 69556u, 67515u };
	uint64_t x = 65536u;
	int i;
	for (i = 0; i < 8; i++)
		if (r & (0x80u >> i))
			x = (x * f[i]) >> 16;
	x = (x >> 16);
	return (uint32_t)(x >= WG_DICT ? WG_DICT - 1 : (x ? x - 1 : 0));
}

/* G_text: URL-like token stream. */
WG_HD void wg_fill_text(uint64_t seed, uint64_t st, uint8_t *out, uint32_t n)
{
	const char seps[8] = { '/', '/', '.', '/', '?', '=', '&', '-' };
	uint32_t pos = 0;
	while (pos < n) {
		uint64_t r = wg_next(&st);
		uint8_t tok[12];
		uint32_t len, i;
		if ((r & 0xf) == 0) {
			/* a short run of digits: ids / ports / timestamps */
			uint64_t d = r >> 8;
			len = 1 + (uint32_t)((r >> 5) & 7);
			for (i = 0; i < len && pos < n; i++) {
				out[pos++] = (uint8_t)('0' + d % 10);
				d /= 10;
			}
		} else {
			len = wg_token(seed, wg_zipf((uint32_t)(r >> 8) & 0xff) ^ ((uint32_t)(r >> 16) & 3), tok);
			for (i = 0; i < len && pos < n; i++)
				out[pos++] = tok[i];
		}
		if (pos < n)
			out[pos++] = (uint8_t)seps[(r >> 40) & 7];
	}
}

/* G_low: runs of one symbol or of a 2..16-byte period; run length 1 + ~geometric(mean 96). */
WG_HD void wg_fill_low(uint64_t st, uint8_t *out, uint32_t n)
{
	uint32_t pos = 0;
	while (pos < n) {
		uint64_t r = wg_next(&st);
		/* geometric-ish: 64 * (number of leading zero bits of a 16-bit draw) + uniform[0,160) */
		uint32_t g = (uint32_t)(r & 0xffff), lz = 0, run, i;
		while (lz < 16 && !(g & 0x8000u)) {
			g <<= 1;
			lz++;
		}
		run = 1 + 64 * lz + (uint32_t)((r >> 16) % 160);
		if (run > n - pos)
			run = n - pos;
		if ((r >> 32) & 1) {
			uint8_t sym = (uint8_t)(0x40 + ((r >> 33) & 15));
			for (i = 0; i < run; i++)
				out[pos++] = sym;
		} else {
			uint64_t pat = wg_next(&st), pat2 = pat * 0x9E3779B97F4A7C15ull;
			uint32_t per = 2 + (uint32_t)((r >> 37) % 15);
			for (i = 0; i < run; i++) {
				uint32_t k = i % per;
				out[pos++] = (uint8_t)((k < 8 ? pat >> (8 * k) : pat2 >> (8 * (k - 8))) & 0xff);
			}
		}
	}
}

/* G_page: zram-style page mix, class chosen by a hash of the page index. */
WG_HD void wg_fill_page(uint64_t seed, uint64_t st, uint8_t *out, uint32_t n)
{
	uint64_t c = wg_next(&st) % 100;
	uint32_t pos = 0, i;
	if (c < 20) {
		for (i = 0; i < n; i++)
			out[i] = 0;
	} else if (c < 60) {
		/* heap words: small integers and pointers that share their upper five bytes */
		uint64_t base = (wg_next(&st) & 0x00007fffff000000ull) | 0x0000500000000000ull;
		uint64_t objs = wg_next(&st);
		while (pos < n) {
			uint64_t r = wg_next(&st), w;
			if (r & 1)
				w = (r >> 8) & 0xff;
			else
				w = base | ((((objs >> ((r >> 1) & 31)) & 0xfff) << 4) + (((r >> 6) & 7) << 16));
			for (i = 0; i < 8 && pos < n; i++)
				out[pos++] = (uint8_t)(w >> (8 * i));
		}
	} else if (c < 85) {
		wg_fill_text(seed, st, out, n);
	} else {
		while (pos < n) {
			uint64_t r = wg_next(&st);
			for (i = 0; i < 8 && pos < n; i++)
				out[pos++] = (uint8_t)(r >> (8 * i));
		}
	}
}

WG_HD void wg_fill_block(int kind, uint64_t seed, uint64_t block, uint8_t *out, uint32_t n)
{
	uint64_t st = wg_block_state(seed, block);
	if (kind == WG_TEXT)
		wg_fill_text(seed, st, out, n);
	else if (kind == WG_LOW)
		wg_fill_low(st, out, n);
	else
		wg_fill_page(seed, st, out, n);
}

#endif /* CSNAPPY_AMD_WORKLOAD_GEN_H_ */
