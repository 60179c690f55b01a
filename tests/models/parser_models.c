/*
 * parser_models.c -- CPU models behind DESIGN.md section 4.1 "Ceiling of this design" (round 4).
 *
 * Test infrastructure (a CPU model like tests/wave_model.py), not product code: it replays the reference's probe loop
 * (csnappy_compress.c:469-606, restated inline as in oracle/snappy_oracle.c) over fragments of the
 * bench workloads and measures what the three restructurings the round-3 verdict proposed for the
 * parser would have to hold in LDS or fetch from HBM:
 *
 *   (a) interval colouring    buckets LIVE at a time (first..last member): the table a colouring
 *                             of the buckets' lifetimes could not go below
 *   (b) predecessor links     the parser keeps only an "inserted" bitmap; a probe's candidate is its
 *                             nearest INSERTED predecessor in its bucket.  With K parse-independent
 *                             links per position, how often is the candidate deeper than K
 *                             (a probe that needs a dependent fetch), per fragment
 *   (c) hot / cold split      buckets by member count: what a table of only the buckets with >= 3
 *                             (>= 4, >= 5) members would hold, and how many buckets are "useful" at all
 *                             (two members share their four bytes; a bucket of pure hash collisions
 *                             can never produce a match)
 *
 * usage: parser_models text|low|page|urls [blocks]     (urls: tests/golden/urls.10K, cwd = repo root)
 * tests/test_models_cpu.py builds and runs it and pins the headline numbers DESIGN.md quotes.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../csnappy_amd/csrc/workload_gen.h"

#define N 32768
#define KMAX 5

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t hash4(uint32_t b, int shift) { return (b * 0x1e35a7bdu) >> shift; }

static uint16_t T[32768];
static int prevpos[N], lastpos[32768], cnt[32768], firstm[32768], lastm[32768], head[32768], nxt[N], delta[N + 2];
static uint8_t ins[N];

static long frags, probes, copies, buckets, useful, maxlive_sum, maxlive_max, deep[KMAX + 1];
static long c2, c3, c4, c5p;

static void probe(uint32_t ip)
{
	int k = 1, j = prevpos[ip];
	while (j >= 0 && !ins[j]) {
		j = prevpos[j];
		k++;
	}
	probes++;
	if (j < 0)
		return; /* no inserted predecessor: the empty slot, candidate position 0 */
	for (int K = 1; K <= KMAX; K++)
		if (k > K)
			deep[K]++;
}

static void fragment(const uint8_t *F, uint32_t n, int p)
{
	const int shift = 33 - p, nb = 1 << (p - 1);
	for (int i = 0; i < nb; i++) {
		lastpos[i] = head[i] = -1;
		cnt[i] = 0;
	}
	for (uint32_t i = 0; i + 4 <= n; i++) {
		const uint32_t h = hash4(rd32(F + i), shift);
		prevpos[i] = lastpos[h];
		lastpos[h] = (int)i;
		if (!cnt[h]++)
			firstm[h] = (int)i;
		lastm[h] = (int)i;
		nxt[i] = head[h];
		head[h] = (int)i;
	}
	memset(delta, 0, sizeof delta);
	for (int h = 0; h < nb; h++) {
		const int c = cnt[h];
		if (c < 2)
			continue;
		buckets++;
		c2 += c == 2;
		c3 += c == 3;
		c4 += c == 4;
		c5p += c >= 5;
		delta[firstm[h]]++;
		delta[lastm[h] + 1]--;
		int use = 0;
		for (int a = head[h]; a >= 0 && !use; a = nxt[a])
			for (int b = nxt[a]; b >= 0; b = nxt[b])
				if (rd32(F + a) == rd32(F + b)) {
					use = 1;
					break;
				}
		useful += use;
	}
	long live = 0, mx = 0;
	for (uint32_t i = 0; i < n; i++) {
		live += delta[i];
		if (live > mx)
			mx = live;
	}
	maxlive_sum += mx;
	if (mx > maxlive_max)
		maxlive_max = mx;
	frags++;

	memset(ins, 0, n);
	memset(T, 0, 2 * (size_t)nb);
	uint32_t ip = 1, ip_limit = n - 15, skip, next_ip, cand, h;
	for (;;) {
		skip = 32;
		next_ip = ip;
		for (;;) {
			ip = next_ip;
			next_ip = ip + (skip >> 5);
			skip++;
			if (next_ip > ip_limit)
				return;
			h = hash4(rd32(F + ip), shift);
			cand = T[h];
			probe(ip);
			T[h] = (uint16_t)ip;
			ins[ip] = 1;
			if (rd32(F + ip) == rd32(F + cand))
				break;
		}
		for (;;) {
			uint32_t m = 4;
			while (ip + m < n && F[cand + m] == F[ip + m])
				m++;
			ip += m;
			copies++;
			if (ip >= ip_limit)
				return;
			T[hash4(rd32(F + ip - 1), shift)] = (uint16_t)(ip - 1);
			ins[ip - 1] = 1;
			h = hash4(rd32(F + ip), shift);
			cand = T[h];
			probe(ip);
			T[h] = (uint16_t)ip;
			ins[ip] = 1;
			if (rd32(F + ip) != rd32(F + cand))
				break;
		}
		ip++;
	}
}

int main(int argc, char **argv)
{
	static uint8_t blk[65536 + 16], buf[1 << 20];
	const char *w = argc > 1 ? argv[1] : "text";
	const int nblocks = argc > 2 ? atoi(argv[2]) : 64, p = 16;
	if (!strcmp(w, "urls")) {
		FILE *f = fopen("tests/golden/urls.10K", "rb");
		if (!f) {
			perror("tests/golden/urls.10K");
			return 1;
		}
		const size_t n = fread(buf, 1, sizeof buf, f);
		fclose(f);
		for (size_t o = 0; o + N <= n; o += N)
			fragment(buf + o, N, p);
	} else {
		const int kind = !strcmp(w, "low") ? WG_LOW : !strcmp(w, "page") ? WG_PAGE : WG_TEXT;
		const uint64_t seed = kind == WG_LOW ? 0xC5A90005ull : kind == WG_PAGE ? 0xC5A90004ull : 0xC5A90001ull;
		for (int b = 0; b < nblocks; b++) {
			wg_fill_block(kind, seed, (uint64_t)b, blk, 65536);
			fragment(blk, N, p);
			fragment(blk + N, N, p);
		}
	}
	const double fr = (double)frags;
	printf("workload %s fragments %ld p %d\n", w, frags, p);
	printf("per_fragment probes %.0f copies %.0f buckets %.0f useful_buckets %.0f\n", probes / fr, copies / fr, buckets / fr, useful / fr);
	printf("a_interval_colouring max_live_buckets avg %.0f max %ld\n", maxlive_sum / fr, maxlive_max);
	printf("b_links deep_probes_per_fragment");
	for (int K = 1; K <= KMAX; K++)
		printf(" K%d %.1f", K, deep[K] / fr);
	printf("\n");
	printf("c_hot_cold buckets_with_members 2: %.0f 3: %.0f 4: %.0f 5+: %.0f\n", c2 / fr, c3 / fr, c4 / fr, c5p / fr);
	return 0;
}
