/*
 * snappy_cli.c (builds tools/cl_tester) -- command-line harness with the observable behaviour of the reference's
 * cl_tester (reference cl_tester.c:1-304), written from scratch against include/csnappy.h and
 * linked with the MI355X library, so the reference's only end-to-end test
 * (reference Makefile:21-29: compress | decompress | diff, then -S d, then -S c) runs on the
 * HIP path.
 *
 *   cl_tester [-d] infile outfile     [de]compress infile to outfile
 *   cl_tester [-d] -c                 [de]compress stdin to stdout
 *   cl_tester -S c | -S d             self-tests
 *   cl_tester -p N ...                (extension) table power passed to csnappy_compress,
 *                                     default CSNAPPY_WORKMEM_BYTES_POWER_OF_TWO = 16 as in
 *                                     the reference (cl_tester.c:105-106)
 *
 * Exit codes follow the reference: 1 usage, 2/3 cannot open in/out, 4 allocation, 5 input larger
 * than 10 MiB, 6 bad header, 7 decompression error.
 */
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include "csnappy.h"

#define INPUT_LIMIT (10u * 1024u * 1024u) /* reference cl_tester.c:12 */

static int table_power = CSNAPPY_WORKMEM_BYTES_POWER_OF_TWO;

/* Read at most INPUT_LIMIT bytes; returns NULL and sets *rc on failure. */
static char *slurp(FILE *f, uint32_t *len, int *rc)
{
	char *buf = malloc(INPUT_LIMIT);
	if (!buf) {
		fprintf(stderr, "malloc failed to allocate %u.\n", INPUT_LIMIT);
		*rc = 4;
		return NULL;
	}
	*len = (uint32_t)fread(buf, 1, INPUT_LIMIT, f);
	if (!feof(f)) {
		fprintf(stderr, "input was longer than %u, aborting.\n", INPUT_LIMIT);
		free(buf);
		*rc = 5;
		return NULL;
	}
	return buf;
}

static int run_compress(FILE *in, FILE *out)
{
	int rc = 0;
	uint32_t ilen, olen = 0;
	char *ibuf = slurp(in, &ilen, &rc), *obuf, *wm;
	fclose(in);
	if (!ibuf) {
		fclose(out);
		return rc;
	}
	obuf = malloc(csnappy_max_compressed_length(ilen));
	wm = malloc(CSNAPPY_WORKMEM_BYTES);
	if (!obuf || !wm) {
		fprintf(stderr, "malloc failed.\n");
		fclose(out);
		return 4;
	}
	csnappy_compress(ibuf, ilen, obuf, &olen, wm, table_power);
	fwrite(obuf, 1, olen, out);
	fclose(out);
	free(ibuf);
	free(obuf);
	free(wm);
	return 0;
}

static int run_decompress(FILE *in, FILE *out)
{
	int rc = 0, st;
	uint32_t ilen, olen = 0;
	char *ibuf = slurp(in, &ilen, &rc), *obuf;
	fclose(in);
	if (!ibuf) {
		fclose(out);
		return rc;
	}
	st = csnappy_get_uncompressed_length(ibuf, ilen, &olen);
	if (st < 0) {
		fprintf(stderr, "snappy_get_uncompressed_length returned %d.\n", st);
		fclose(out);
		return 6;
	}
	obuf = malloc(olen ? olen : 1);
	if (!obuf) {
		fprintf(stderr, "malloc failed to allocate %u.\n", olen);
		fclose(out);
		return 4;
	}
	st = csnappy_decompress(ibuf, ilen, obuf, olen);
	if (st != CSNAPPY_E_OK) {
		fprintf(stderr, "snappy_decompress returned %d.\n", st);
		fclose(out);
		return 7;
	}
	fwrite(obuf, 1, olen, out);
	fclose(out);
	free(ibuf);
	free(obuf);
	return 0;
}

static void fill_random(char *buf, uint32_t n)
{
	FILE *f = fopen("/dev/urandom", "rb");
	if (!f || fread(buf, 1, n, f) < n) {
		perror("/dev/urandom");
		exit(EXIT_FAILURE);
	}
	fclose(f);
}

/* One writable page followed by a PROT_NONE guard page. */
static char *guarded_page(long page)
{
	char *p = mmap(NULL, 2 * page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (p == MAP_FAILED || mprotect(p + page, page, PROT_NONE)) {
		perror("mmap/mprotect");
		exit(EXIT_FAILURE);
	}
	return p;
}

static void on_segv(int sig)
{
	if (sig == SIGSEGV) {
		static const char msg[] = "compression overwrites out buffer\n";
		if (write(1, msg, sizeof(msg) - 1) < 0)
			_exit(EXIT_FAILURE);
		_exit(EXIT_SUCCESS);
	}
}

/* reference cl_tester.c:127-165: the compressor does no output bounds checking, so writing
 * PAGE+100 incompressible bytes into a one-page buffer must run into the guard page. */
static int selftest_compress(void)
{
	long page = sysconf(_SC_PAGE_SIZE);
	uint32_t ilen = (uint32_t)page + 100, olen = 0;
	char *obuf = guarded_page(page), *ibuf = malloc(ilen), *wm = malloc(CSNAPPY_WORKMEM_BYTES);
	struct sigaction sa;
	fill_random(ibuf, ilen);
	memset(&sa, 0, sizeof(sa));
	sa.sa_handler = on_segv;
	sigemptyset(&sa.sa_mask);
	sigaction(SIGSEGV, &sa, NULL);
	csnappy_compress(ibuf, ilen, obuf, &olen, wm, table_power);
	fprintf(stderr, "ERROR: csnappy_compress did not segfault when should have!\n");
	return EXIT_FAILURE;
}

/* reference cl_tester.c:167-238 */
static int selftest_decompress(void)
{
	static const char cut_literal[] = "\x32\xc4\x66\x6f\x6f\x6f\x6f\x6f\x6f";
	long page = sysconf(_SC_PAGE_SIZE);
	uint32_t ilen = (uint32_t)page + 100, clen = 0, n = 0, olen;
	char *plain = malloc(ilen), *comp = malloc(csnappy_max_compressed_length(ilen));
	char *wm = malloc(CSNAPPY_WORKMEM_BYTES), *obuf;
	int st, hlen;
	fill_random(plain, ilen);
	csnappy_compress(plain, ilen, comp, &clen, wm, table_power);
	obuf = guarded_page(page);
	/* (a) header says PAGE+100 but only one page is offered */
	st = csnappy_decompress(comp, clen, obuf, (uint32_t)page);
	if (st != CSNAPPY_E_OUTPUT_INSUF) {
		fprintf(stderr, "snappy_decompress returned %d.\n", st);
		return EXIT_FAILURE;
	}
	/* (b) body only, one page of room: must stop with -3 and stay off the guard page */
	hlen = csnappy_get_uncompressed_length(comp, clen, &n);
	if (hlen == CSNAPPY_E_HEADER_BAD) {
		fprintf(stderr, "csnappy_get_uncompressed_length, could not obtain header length\n");
		return EXIT_FAILURE;
	}
	olen = (uint32_t)page;
	st = csnappy_decompress_noheader(comp + hlen, clen - hlen, obuf, &olen);
	if (st != CSNAPPY_E_OUTPUT_OVERRUN) {
		fprintf(stderr, "csnappy_decompress_noheader returned %d.\n", st);
		return EXIT_FAILURE;
	}
	munmap(obuf, 2 * page);
	/* (c) stream cut off in the middle of a literal */
	olen = 50;
	obuf = malloc(olen);
	st = csnappy_decompress(cut_literal, 9, obuf, olen);
	if (st == CSNAPPY_E_OK) {
		fprintf(stderr, "csnappy_decompress, stream cut off mid literal: %d\n", st);
		return EXIT_FAILURE;
	}
	st = csnappy_decompress_noheader(cut_literal + 1, 8, obuf, &olen);
	if (st == CSNAPPY_E_OK) {
		fprintf(stderr, "csnappy_decompress_noheader, stream cut off mid literal: %d\n", st);
		return EXIT_FAILURE;
	}
	return 0;
}

static int usage(void)
{
	fprintf(stderr, "Usage:\n"
			"cl_tester [-d] infile outfile\t-\t[de]compress infile to outfile.\n"
			"cl_tester [-d] -c\t\t-\t[de]compress stdin to stdout.\n"
			"cl_tester -S c\t\t\t-\tSelf-test compression.\n"
			"cl_tester -S d\t\t\t-\tSelf-test decompression.\n"
			"cl_tester -p N ...\t\t-\ttable power for compression (9..16, default 16).\n");
	return 1;
}

int main(int argc, char *const argv[])
{
	int c, decompress = 0, use_files = 1, st_c = 0, st_d = 0;
	FILE *in = stdin, *out = stdout;
	while ((c = getopt(argc, argv, "S:dcp:")) != -1) {
		switch (c) {
		case 'S':
			if (optarg[0] == 'c')
				st_c = 1;
			else if (optarg[0] == 'd')
				st_d = 1;
			else
				return usage();
			break;
		case 'd':
			decompress = 1;
			break;
		case 'c':
			use_files = 0;
			break;
		case 'p':
			table_power = atoi(optarg);
			if (table_power < 9 || table_power > 16)
				return usage();
			break;
		default:
			return usage();
		}
	}
	if (st_c)
		return selftest_compress();
	if (st_d)
		return selftest_decompress();
	if (use_files) {
		if (optind > argc - 2)
			return usage();
		if (!(in = fopen(argv[optind], "rb"))) {
			perror("fopen of ifile_name");
			return 2;
		}
		if (!(out = fopen(argv[optind + 1], "wb"))) {
			perror("fopen of ofile_name");
			return 3;
		}
	}
	return decompress ? run_decompress(in, out) : run_compress(in, out);
}
