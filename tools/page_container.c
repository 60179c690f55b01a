/*
 * page_container.c (builds tools/block_compressor) -- the zram-style 4 KiB page path of the
 * reference's block_compressor (reference block_compressor.c:275-394), snappy method only, on the
 * batched HIP API: every page of the input is one FRAGMENT-mode block
 * (csnappy_compress_fragment(page, <=4096, dst, wm, 13) / csnappy_decompress_noheader, reference
 * block_compressor.c:113-135), all pages of the file go through ONE batch launch.
 *
 * On-disk container, byte-identical to the reference's (block_compressor.c:296-334):
 *     u32 nr_pages | u32 len[nr_pages] | payloads back to back
 * A page whose compressed size is >= its input size is stored raw with len = input size
 * (:316-319); on decode len == 4096 means raw (:378-379).
 *
 *   block_compressor -c snappy infile outfile        compress
 *   block_compressor -c snappy -d infile outfile     decompress
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "csnappy.h"
#include "csnappy_hip.h"

#define PAGE 4096u
#define ORDER 13 /* WMSIZE_ORDER = PAGE_SHIFT + 1, reference block_compressor.c:99 */

static void die(const char *what)
{
	fprintf(stderr, "block_compressor: %s (%s)\n", what, csnappy_hip_last_error());
	exit(EXIT_FAILURE);
}

static void *dmalloc(size_t n)
{
	void *p = NULL;
	if (hipMalloc(&p, n ? n : 256) != hipSuccess)
		die("hipMalloc");
	return p;
}

static char *read_all(const char *name, size_t *len)
{
	FILE *f = fopen(name, "rb");
	char *buf;
	long n;
	if (!f || fseek(f, 0, SEEK_END) || (n = ftell(f)) < 0 || fseek(f, 0, SEEK_SET)) {
		perror(name);
		exit(2);
	}
	buf = malloc((size_t)n + 1);
	if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) {
		perror(name);
		exit(2);
	}
	fclose(f);
	*len = (size_t)n;
	return buf;
}

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec + t.tv_nsec * 1e-9;
}

static int compress_file(const char *in_name, const char *out_name, int per_page)
{
	size_t n, slot = csnappy_max_compressed_length(PAGE), ws;
	char *in = read_all(in_name, &n);
	uint32_t nr = (uint32_t)((n + PAGE - 1) / PAGE), i, counts[3] = { 0, 0, 0 };
	uint64_t *off = malloc(2 * (size_t)nr * sizeof(uint64_t) + 16);
	uint32_t *len = malloc(2 * (size_t)nr * sizeof(uint32_t) + 16);
	char *out = malloc((size_t)nr * slot + 16);
	void *d_in, *d_out, *d_off, *d_len, *d_ws;
	double t0, t1;
	FILE *f;

	printf("compressor: snappy\n#pages: %u\n", nr);
	for (i = 0; i < nr; i++) {
		off[i] = (uint64_t)i * PAGE;      /* in_off */
		off[nr + i] = (uint64_t)i * slot; /* out_off */
		len[i] = (uint32_t)(n - (size_t)i * PAGE < PAGE ? n - (size_t)i * PAGE : PAGE);
	}
	/* launches of up to 1 GiB of pages (4.0 GiB of scratch for the largest files; the least the call
	 * accepts, csnappy_hip_compress_workspace_size, is launches of 128 MiB: a tenth slower) */
	ws = csnappy_hip_compress_workspace_size_for(nr, PAGE, 1);
	d_in = dmalloc(n + 64);
	d_out = dmalloc((size_t)nr * slot);
	d_off = dmalloc(2 * (size_t)nr * sizeof(uint64_t));
	d_len = dmalloc(2 * (size_t)nr * sizeof(uint32_t));
	d_ws = dmalloc(ws);
	if (hipMemcpy(d_in, in, n, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_off, off, 2 * (size_t)nr * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_len, len, (size_t)nr * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
		die("hipMemcpy H2D");
	if (per_page) {
		/* -1: the reference's loop (block_compressor.c:307-337): one call per page, its time
		 * taken around that call alone and summed -- here one 1-block launch + wait per page,
		 * pages resident on the device.  This is the latency of a call, not the throughput. */
		double sum = 0;
		for (i = 0; i < nr; i++) {
			t0 = now();
			if (csnappy_hip_compress_batch(d_in, (uint64_t *)d_off + i, (uint32_t *)d_len + i, 1, PAGE, d_out,
						       (uint64_t *)d_off + nr + i, (uint32_t *)d_len + nr + i, ORDER,
						       CSNAPPY_HIP_FRAGMENT, d_ws, ws, NULL))
				die("csnappy_hip_compress_batch");
			if (hipDeviceSynchronize() != hipSuccess)
				die("compress kernels");
			sum += now() - t0;
		}
		t0 = 0;
		t1 = sum;
	} else {
		t0 = now();
		if (nr && csnappy_hip_compress_batch(d_in, (uint64_t *)d_off, (uint32_t *)d_len, nr, PAGE, d_out,
						     (uint64_t *)d_off + nr, (uint32_t *)d_len + nr, ORDER,
						     CSNAPPY_HIP_FRAGMENT, d_ws, ws, NULL))
			die("csnappy_hip_compress_batch");
		if (hipDeviceSynchronize() != hipSuccess)
			die("compress kernels");
		t1 = now();
	}
	if (hipMemcpy(len + nr, (uint32_t *)d_len + nr, (size_t)nr * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
	    hipMemcpy(out, d_out, (size_t)nr * slot, hipMemcpyDeviceToHost) != hipSuccess)
		die("hipMemcpy D2H");

	if (!(f = fopen(out_name, "wb"))) {
		perror(out_name);
		return 3;
	}
	fwrite(&nr, 4, 1, f);
	for (i = 0; i < nr; i++) {
		uint32_t olen = len[nr + i];
		if (olen >= len[i]) { /* incompressible: stored raw, block_compressor.c:316-319 */
			olen = len[i];
			counts[2]++;
		} else if (olen > PAGE / 2) {
			counts[1]++;
		} else {
			counts[0]++;
		}
		fwrite(&olen, 4, 1, f);
	}
	if (nr == 0) /* the reference always grows the file past the table (:300-302) */
		fwrite(&nr, 4, 1, f);
	for (i = 0; i < nr; i++) {
		if (len[nr + i] >= len[i])
			fwrite(in + (size_t)i * PAGE, 1, len[i], f);
		else
			fwrite(out + (size_t)i * slot, 1, len[nr + i], f);
	}
	fclose(f);
	printf("> 100%%\t:%u\n> 50%%\t:%u\n<= 50%%\t:%u\n%.9f seconds\n", counts[2], counts[1], counts[0], t1 - t0);
	if (nr)
		printf("%.0f ns per page (%s)\n", (t1 - t0) * 1e9 / nr,
		       per_page ? "one launch + wait per page, as the reference times it" : "one batch launch / #pages");
	return 0;
}

static int decompress_file(const char *in_name, const char *out_name)
{
	size_t n, pos;
	char *in = read_all(in_name, &n), *out;
	uint32_t nr, i, *len;
	uint64_t *off;
	int32_t *status;
	void *d_in, *d_out, *d_off, *d_len, *d_st;
	FILE *f;

	if (n < 4)
		die("short container");
	memcpy(&nr, in, 4);
	printf("nr_pages: %u\n", nr);
	if ((size_t)nr > (n - 4) / 4) /* (the header is untrusted: no 32-bit arithmetic on it) */
		die("short container");
	len = malloc(4 * (size_t)nr * sizeof(uint32_t) + 16); /* in_len | out_cap | status | produced */
	off = malloc(2 * (size_t)nr * sizeof(uint64_t) + 16);
	out = malloc((size_t)nr * PAGE + 16);
	if (!len || !off || !out)
		die("out of memory");
	status = (int32_t *)(len + 2 * (size_t)nr);
	pos = ((size_t)nr + 1) * 4;
	for (i = 0; i < nr; i++) {
		memcpy(&len[i], in + 4 + 4 * (size_t)i, 4);
		off[i] = pos;
		off[nr + i] = (uint64_t)i * PAGE;
		len[nr + i] = PAGE;
		pos += len[i];
		if (pos > n)
			die("container truncated");
	}
	d_in = dmalloc(n + 64);
	d_out = dmalloc((size_t)nr * PAGE);
	d_off = dmalloc(2 * (size_t)nr * sizeof(uint64_t));
	d_len = dmalloc(4 * (size_t)nr * sizeof(uint32_t));
	d_st = (uint32_t *)d_len + 2 * (size_t)nr;
	if (hipMemcpy(d_in, in, n, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_off, off, 2 * (size_t)nr * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_len, len, 2 * (size_t)nr * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
		die("hipMemcpy H2D");
	/* raw pages (len == PAGE) are decoded too and then overwritten from the container: it keeps
	 * the launch a single uniform batch */
	if (nr && csnappy_hip_decompress_batch(d_in, (uint64_t *)d_off, (uint32_t *)d_len, nr, d_out,
					       (uint64_t *)d_off + nr, (uint32_t *)d_len + nr, (int32_t *)d_st,
					       (uint32_t *)d_st + nr, CSNAPPY_HIP_FRAGMENT, NULL))
		die("csnappy_hip_decompress_batch");
	if (hipDeviceSynchronize() != hipSuccess)
		die("decompress kernel");
	if (hipMemcpy(status, d_st, 2 * (size_t)nr * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess ||
	    hipMemcpy(out, d_out, (size_t)nr * PAGE, hipMemcpyDeviceToHost) != hipSuccess)
		die("hipMemcpy D2H");
	if (!(f = fopen(out_name, "wb"))) {
		perror(out_name);
		return 3;
	}
	for (i = 0; i < nr; i++) {
		uint32_t olen = PAGE;
		if (len[i] == PAGE) { /* raw page, block_compressor.c:378-379 */
			fwrite(in + off[i], 1, PAGE, f);
		} else {
			if (status[i] != CSNAPPY_E_OK) {
				fprintf(stderr, "decompress: page %u returned %d\n", i, status[i]);
				return EXIT_FAILURE;
			}
			olen = ((uint32_t *)status)[nr + i];
			fwrite(out + (size_t)i * PAGE, 1, olen, f);
		}
		printf("%u -> %u\n", len[i], olen);
	}
	fclose(f);
	return 0;
}

int main(int argc, char *const argv[])
{
	int c, decompress = 0, have_method = 0, per_page = 0;
	while ((c = getopt(argc, argv, "d1c:")) != -1) {
		if (c == 'd')
			decompress = 1;
		else if (c == '1')
			per_page = 1;
		else if (c == 'c' && !strcmp(optarg, "snappy"))
			have_method = 1;
		else
			goto usage;
	}
	if (!have_method || optind > argc - 2)
		goto usage;
	if (csnappy_hip_device_count() <= 0)
		die("no usable HIP device");
	return decompress ? decompress_file(argv[optind], argv[optind + 1])
			  : compress_file(argv[optind], argv[optind + 1], per_page);
usage:
	fprintf(stderr, "usage: block_compressor -c snappy [-d] [-1] infile outfile\n");
	return 1;
}
