/*
 * asan_check.c -- TEST INFRASTRUCTURE: runs the CPU restatement (snappy_oracle.c) and the two
 * host-arithmetic entry points of the product's C layer (csnappy_host.c, compiled with
 * CSNAPPY_HOST_ARITH_ONLY: no HIP) under AddressSanitizer + UndefinedBehaviorSanitizer.
 *
 * Replaces the reference's valgrind target (`make check_leaks`, reference Makefile:31-35; valgrind
 * is not in the image).  GPU ASan is not available on the pool, so this covers the CPU side only.
 *
 * usage: asan_check <vector file>      (written by tests/test_oracle.py::test_sanitizer_build)
 * vector file = records of
 *   u32 kind (1 compress stream, 2 compress fragment, 3 decompress stream, 4 decompress noheader,
 *             5 get_uncompressed_length: expected value in p_or_cap), i32 p_or_cap, u32 in_len, i32 want_rc, u32 want_len,
 *   in bytes, want bytes
 * Every buffer is malloc'ed at its exact size so that any out-of-bounds access is reported.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "snappy_oracle.c"
#define CSNAPPY_HOST_ARITH_ONLY 1
#include "../csnappy_amd/csrc/csnappy_host.c"

static uint32_t file_u32(FILE *f)
{
	uint32_t v = 0;
	if (fread(&v, 4, 1, f) != 1)
		v = 0xffffffffu;
	return v;
}

int main(int argc, char **argv)
{
	FILE *f;
	unsigned long n_ok = 0;
	if (argc < 2 || !(f = fopen(argv[1], "rb")))
		return 2;
	for (;;) {
		uint32_t kind = file_u32(f), in_len, want_len;
		int32_t pc, want_rc;
		uint8_t *in, *want;
		if (kind == 0xffffffffu)
			break;
		pc = (int32_t)file_u32(f);
		in_len = file_u32(f);
		want_rc = (int32_t)file_u32(f);
		want_len = file_u32(f);
		in = malloc(in_len ? in_len : 1);
		want = malloc(want_len ? want_len : 1);
		if ((in_len && fread(in, 1, in_len, f) != in_len) || (want_len && fread(want, 1, want_len, f) != want_len))
			return 3;
		if (kind == 1 || kind == 2) {
			uint32_t cap = orc_max_compressed_length(in_len), got = 0;
			uint8_t *out = malloc(cap ? cap : 1);
			if (cap != csnappy_max_compressed_length(in_len))
				return 10;
			if (kind == 1)
				orc_compress(in, in_len, out, &got, pc);
			else
				got = orc_compress_fragment(in, in_len, out, pc);
			if (got != want_len || memcmp(out, want, got))
				return 11;
			{
				/* and back, into a buffer of exactly the original size */
				uint8_t *back = malloc(in_len ? in_len : 1);
				uint32_t blen = in_len;
				int rc = kind == 1 ? orc_decompress(out, got, back, in_len)
						   : orc_decompress_noheader(out, got, back, &blen);
				if (rc != 0 || (kind == 2 && blen != in_len) || memcmp(back, in, in_len))
					return 12;
				free(back);
			}
			free(out);
		} else if (kind == 3 || kind == 4) {
			uint32_t cap = (uint32_t)pc, dl = cap;
			uint8_t *dst = malloc(cap ? cap : 1);
			int rc = kind == 3 ? orc_decompress(in, in_len, dst, cap) : orc_decompress_noheader(in, in_len, dst, &dl);
			if (rc != want_rc)
				return 13;
			if (rc == 0 && kind == 4 && (dl != want_len || memcmp(dst, want, dl)))
				return 14;
			free(dst);
		} else if (kind == 5) {
			uint32_t a = 0xdeadbeefu, b = 0xdeadbeefu;
			int ra = orc_get_uncompressed_length(in, in_len, &a);
			int rb = csnappy_get_uncompressed_length((const char *)in, in_len, &b);
			if (ra != want_rc || rb != want_rc || (ra > 0 && (a != (uint32_t)pc || b != (uint32_t)pc)))
				return 15;
		} else {
			return 4;
		}
		free(in);
		free(want);
		n_ok++;
	}
	fclose(f);
	printf("asan_check ok: %lu vectors\n", n_ok);
	return 0;
}
