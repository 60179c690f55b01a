/*
 * csnappy_kernels.hip -- Snappy raw-block codec for MI355X (gfx950, wave64), from scratch.
 *
 * What the kernels replace in the reference (file:line into the reference tree):
 *   snappy_compress_fragments   csnappy_compress_fragment          csnappy_compress.c:469-606
 *                               Hash/HashBytes                     :228-236
 *                               FindMatchLength                    :252-295
 *                               EmitLiteral / EmitCopy(LessThan64) :332-415
 *                               fragment loop + table-size pick    :636-654
 *   snappy_stitch_blocks        the `compressed = p` pointer chain :633-651 (fragment k+1 starts
 *                               where fragment k ended) + encode_varint32 :46-73
 *   snappy_decompress_blocks    csnappy_decompress_noheader        csnappy_decompress.c:319-387
 *                               SAW__Append* / IncrementalCopy*    :200-317
 *                               csnappy_get_uncompressed_length / csnappy_decompress :45-71,394-411
 *
 * Design (DESIGN.md has the long form):
 *   - compress: ONE WAVE PER 32 KiB FRAGMENT.  The fragment window and the uint16 hash table
 *     live in LDS.  The reference's probe loop is a sequential recurrence; we evaluate 64
 *     consecutive probe positions of that recurrence per step (one per lane), resolve the
 *     only intra-step dependency (two lanes with the same hash slot) exactly by truncating the
 *     step at the first such lane, and take the first lane whose candidate matches.  The
 *     result is bit-identical to the sequential loop.  Match extension compares 512 B per
 *     step across the wave.  Emission is deferred: (literal, copy) records are queued in LDS
 *     and 64 of them are encoded at once with a wave prefix sum for the output offsets.
 *   - decompress: one wave per block; 64 candidate tag positions are decoded in parallel,
 *     the true tag chain is walked on the scalar unit with v_readlane, per-element output
 *     offsets come from a wave prefix sum, errors are resolved in element order, then the
 *     elements are executed with wave-wide copies.
 *   - no MFMA: this is byte/integer work bound by LDS latency and HBM, not a contraction.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/csnappy.h"
#include "../../include/csnappy_hip.h"
#include "workload_gen.h"

namespace {

constexpr uint32_t kFragment = 32768;    /* kBlockSize, csnappy_compress.c:85-86 */
constexpr uint32_t kMargin = 15;         /* kInputMarginBytes, csnappy_compress.c:468 */
constexpr uint32_t kHashMul = 0x1e35a7bdu; /* csnappy_compress.c:230 */
constexpr uint32_t kScratchSlot = 38400; /* >= max_compressed_length(32768)=38261, 256-aligned */
constexpr uint32_t kShortLiteral = 24;   /* literals up to this are copied lane-per-record */

#define DEVINL __device__ __forceinline__

struct CompressArgs {
	const uint8_t *in;
	const uint64_t *in_off;
	const uint32_t *in_len;
	uint8_t *out;
	const uint64_t *out_off;
	uint32_t *out_len;
	uint8_t *scratch;   /* (fpb-1) slots of kScratchSlot bytes per block */
	uint32_t *frag_len; /* fpb entries per block */
	uint32_t nblocks;
	uint32_t fpb;       /* fragments per block (upper bound) */
	uint32_t win_bytes; /* LDS bytes reserved for the window */
	uint32_t s_entries; /* conflict-scratch entries (power of two) */
	int p;
	int mode;
};

struct DecompressArgs {
	const uint8_t *in;
	const uint64_t *in_off;
	const uint32_t *in_len;
	uint8_t *out;
	const uint64_t *out_off;
	const uint32_t *out_cap;
	int32_t *status;
	uint32_t *produced;
	uint32_t nblocks;
	int mode;
};

DEVINL uint32_t rdlane(uint32_t v, uint32_t l)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}

DEVINL uint32_t first_lane(uint64_t m)
{
	return (uint32_t)__builtin_ctzll(m);
}

/* 4 bytes at an arbitrary byte index of an LDS array that is addressed as dwords. */
DEVINL uint32_t lds_rd32(const uint32_t *w, uint32_t byte)
{
	const uint32_t d = byte >> 2;
	return __builtin_amdgcn_alignbyte(w[d + 1], w[d], byte & 3);
}

DEVINL uint64_t lds_rd64(const uint32_t *w, uint32_t byte)
{
	const uint32_t d = byte >> 2, sh = byte & 3;
	const uint32_t a = w[d], b = w[d + 1], c = w[d + 2];
	const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, sh);
	const uint32_t hi = __builtin_amdgcn_alignbyte(c, b, sh);
	return ((uint64_t)hi << 32) | lo;
}

/* exclusive prefix sum across the 64 lanes; *total receives the wave sum */
DEVINL uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t *total)
{
	uint32_t x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64);
		if (lane >= (uint32_t)d)
			x += y;
	}
	*total = rdlane(x, 63);
	return x - v;
}

DEVINL uint32_t varint_len(uint32_t v)
{
	return v < (1u << 7) ? 1 : v < (1u << 14) ? 2 : v < (1u << 21) ? 3 : v < (1u << 28) ? 4 : 5;
}

/* Probe i of a scan that starts at position s: the reference advances by (skip++ >> 5) with
 * skip starting at 32 (csnappy_compress.c:535-542), i.e. 32 probes at stride 1, 32 at stride 2.. */
DEVINL uint32_t scan_pos(uint32_t s, uint32_t i)
{
	const uint32_t a = i >> 5, b = i & 31;
	return s + 16u * a * (a + 1) + b * (a + 1);
}

/* Table power csnappy_compress uses for a fragment of n bytes (csnappy_compress.c:638-646). */
DEVINL int fragment_power(uint32_t n, int p, int mode)
{
	int ws = p;
	if (mode == CSNAPPY_HIP_STREAM && n < kFragment) {
		for (ws = 9; ws < p; ++ws)
			if ((1u << (ws - 1)) >= n)
				break;
	}
	return ws;
}

/* ------------------------------------------------------------------------------------------
 * Encoded size of one (literal, copy) record and the EmitCopy chunking of
 * csnappy_compress.c:395-415.
 * ---------------------------------------------------------------------------------------- */
struct CopyPlan {
	uint32_t k64;   /* number of leading 64-byte pieces */
	uint32_t k60;   /* 0/1: one 60-byte piece */
	uint32_t last;  /* final piece length, 4..64 (0 when there is no copy) */
	uint32_t bytes; /* total encoded bytes */
};

DEVINL CopyPlan plan_copy(uint32_t len, uint32_t off)
{
	CopyPlan c = { 0, 0, 0, 0 };
	if (len == 0)
		return c;
	if (len >= 68) {
		c.k64 = (len - 68) / 64 + 1;
		len -= 64 * c.k64;
	}
	if (len > 64) {
		c.k60 = 1;
		len -= 60;
	}
	c.last = len;
	c.bytes = 3 * (c.k64 + c.k60) + ((len < 12 && off < 2048) ? 2 : 3);
	return c;
}

/* ==========================================================================================
 * COMPRESS: one wave per fragment
 * ======================================================================================== */
extern "C" __global__ void __launch_bounds__(64) snappy_compress_fragments(CompressArgs A)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	const uint32_t lane = threadIdx.x;
	const uint32_t id = blockIdx.x;
	const uint32_t blk = id / A.fpb, fi = id - blk * A.fpb;
	const uint32_t len = A.in_len[blk];
	const uint32_t foff = fi * kFragment;
	if (fi > 0 && foff >= len)
		return;
	const uint32_t n = min(len - foff, kFragment);
	const int ws = fragment_power(n, A.p, A.mode);
	const uint32_t shift = 33 - ws;
	const uint8_t *src = A.in + A.in_off[blk] + foff;

	uint8_t *dst;
	uint32_t hdr = 0;
	if (fi == 0) {
		dst = A.out + A.out_off[blk];
		if (A.mode == CSNAPPY_HIP_STREAM) {
			/* encode_varint32, csnappy_compress.c:46-73 */
			hdr = varint_len(len);
			if (lane < hdr)
				dst[lane] = (uint8_t)((len >> (7 * lane)) | (lane + 1 < hdr ? 0x80u : 0u));
			dst += hdr;
		}
	} else {
		dst = A.scratch + (uint64_t)(blk * (uint64_t)(A.fpb - 1) + (fi - 1)) * kScratchSlot;
	}

	/* ---- LDS carve: window | hash table | conflict scratch | record queue ---- */
	uint32_t *win32 = reinterpret_cast<uint32_t *>(smem);
	const uint8_t *win8 = smem;
	uint16_t *tab = reinterpret_cast<uint16_t *>(smem + A.win_bytes);
	uint32_t *S = reinterpret_cast<uint32_t *>(smem + A.win_bytes + (1u << A.p));
	uint32_t *evq = S + A.s_entries; /* 2 dwords per record, 64 records */
	const uint32_t smask = A.s_entries - 1;

	/* window: aligned 16 B chunks; byte i of the fragment sits at win8[wbase + i] */
	const uint32_t wbase = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
	{
		const uint4 *g = reinterpret_cast<const uint4 *>(src - wbase);
		uint4 *l = reinterpret_cast<uint4 *>(smem);
		const uint32_t chunks = (wbase + n + 15) >> 4;
		for (uint32_t k = lane; k < chunks; k += 64)
			l[k] = g[k];
	}
	if (n >= kMargin) {
		/* memset(table, 0), csnappy_compress.c:501: an empty slot means position 0 */
		uint4 *t4 = reinterpret_cast<uint4 *>(tab);
		for (uint32_t k = lane; k < ((1u << ws) >> 4); k += 64)
			t4[k] = make_uint4(0, 0, 0, 0);
		uint4 *s4 = reinterpret_cast<uint4 *>(S);
		for (uint32_t k = lane; k < (A.s_entries >> 2); k += 64)
			s4[k] = make_uint4(~0u, ~0u, ~0u, ~0u);
	}
	__syncthreads();

	uint32_t op = 0;        /* bytes of output already written */
	uint32_t nev = 0;       /* records queued */
	uint32_t next_emit = 0; /* csnappy_compress.c:496 */

	/* Encode and store the queued records (EmitLiteral + EmitCopy). */
	auto flush = [&]() {
		uint32_t lit_start = 0, lit_len = 0, coff = 0, clen = 0;
		if (lane < nev) {
			const uint32_t r0 = evq[2 * lane], r1 = evq[2 * lane + 1];
			lit_start = r0 & 0xffff;
			lit_len = r0 >> 16;
			coff = r1 & 0xffff;
			clen = r1 >> 16;
		}
		const uint32_t lhdr = lit_len == 0 ? 0 : lit_len <= 60 ? 1 : lit_len <= 256 ? 2 : 3;
		const CopyPlan cp = plan_copy(clen, coff);
		uint32_t total;
		const uint32_t mine = lhdr + lit_len + cp.bytes;
		const uint32_t excl = wave_excl_scan(mine, lane, &total);
		uint8_t *o = dst + op + excl;
		/* literal header, csnappy_compress.c:335-368 */
		if (lhdr == 1) {
			o[0] = (uint8_t)((lit_len - 1) << 2);
		} else if (lhdr == 2) {
			o[0] = (uint8_t)(60 << 2);
			o[1] = (uint8_t)(lit_len - 1);
		} else if (lhdr == 3) {
			o[0] = (uint8_t)(61 << 2);
			o[1] = (uint8_t)((lit_len - 1) & 0xff);
			o[2] = (uint8_t)((lit_len - 1) >> 8);
		}
		/* short literal payloads: one lane per record */
		const bool is_short = lit_len <= kShortLiteral;
		for (uint32_t j = 0; __ballot(is_short && j < lit_len); ++j)
			if (is_short && j < lit_len)
				o[lhdr + j] = win8[wbase + lit_start + j];
		/* long literal payloads: the whole wave per record */
		for (uint64_t lm = __ballot(!is_short); lm; lm &= lm - 1) {
			const uint32_t e = first_lane(lm);
			const uint32_t ls = rdlane(lit_start, e), ll = rdlane(lit_len, e);
			uint8_t *ob = dst + op + rdlane(excl, e) + rdlane(lhdr, e);
			for (uint32_t j = lane; j < ll; j += 64)
				ob[j] = win8[wbase + ls + j];
		}
		/* copy tags, csnappy_compress.c:373-415 */
		if (clen) {
			uint8_t *q = o + lhdr + lit_len;
			const uint8_t lo = (uint8_t)(coff & 0xff), hi = (uint8_t)(coff >> 8);
			for (uint32_t k = 0; k < cp.k64; ++k) {
				q[0] = 0xfe; /* COPY_2 | (63 << 2) */
				q[1] = lo;
				q[2] = hi;
				q += 3;
			}
			if (cp.k60) {
				q[0] = 0xee; /* COPY_2 | (59 << 2) */
				q[1] = lo;
				q[2] = hi;
				q += 3;
			}
			if (cp.last < 12 && coff < 2048) {
				q[0] = (uint8_t)(1 + ((cp.last - 4) << 2) + ((coff >> 8) << 5));
				q[1] = lo;
			} else {
				q[0] = (uint8_t)(2 + ((cp.last - 1) << 2));
				q[1] = lo;
				q[2] = hi;
			}
		}
		op += total;
		nev = 0;
	};

	if (n >= kMargin) {
		const uint32_t ip_limit = n - kMargin;
		uint32_t ip = 0;       /* position right after the last copy (spec > 0) */
		uint32_t spec = 0;     /* leading special lanes: 2 = {insert ip-1, probe ip}, 1 = {probe ip} */
		uint32_t s = 1, qi = 0; /* scan start and index of the next scan probe */
		uint32_t epoch = 0x03ffffffu;

		for (;;) {
			/* ---- lane roles for this step ---- */
			uint32_t pos;
			bool valid, probing = true;
			if (lane < spec) {
				const uint32_t k = lane + (2 - spec); /* 0: insert ip-1, 1: probe ip */
				pos = ip - 1 + k;
				valid = true;
				probing = (k == 1);
			} else {
				const uint32_t i = qi + lane - spec;
				pos = scan_pos(s, i);
				/* csnappy_compress.c:542-544: a probe happens only if the NEXT position
				 * is still <= ip_limit */
				valid = scan_pos(s, i + 1) <= ip_limit;
				if (!valid)
					pos = 0;
			}
			const uint32_t w = lds_rd32(win32, wbase + pos);
			const uint32_t h = (w * kHashMul) >> shift;
			const uint32_t key = h & smask;
			if (valid)
				atomicMin(&S[key], (epoch << 6) | lane);
			const uint32_t cand = tab[h];
			__syncthreads();
			const uint32_t first_same = S[key] & 63u; /* lowest valid lane with my slot key */
			const uint32_t cw = lds_rd32(win32, wbase + cand);
			const uint64_t cmask = __ballot(valid && first_same < lane);
			const uint64_t imask = ~__ballot(valid);
			const uint32_t c = cmask ? first_lane(cmask) : 64; /* first lane that depends on an earlier one */
			const uint32_t v = imask ? first_lane(imask) : 64; /* first lane past the scan limit */
			const uint32_t ulim = min(c, v);
			const uint64_t mmask = __ballot(lane < ulim && probing && cw == w);
			epoch--;

			if (mmask == 0) {
				/* no 4-byte match among the usable lanes: commit their table writes
				 * (table[hash] = ip, csnappy_compress.c:550) and move on */
				if (lane < ulim)
					tab[h] = (uint16_t)pos;
				if (ulim == v && v < 64)
					break; /* goto emit_remainder, :543-544 */
				if (ulim < spec) {
					spec -= ulim;
					s = ip + 1;
					qi = 0;
				} else {
					if (spec) {
						s = ip + 1;
						qi = 0;
					}
					qi += ulim - spec;
					spec = 0;
				}
				continue;
			}

			const uint32_t m = first_lane(mmask);
			if (lane <= m)
				tab[h] = (uint16_t)pos;
			const uint32_t base = rdlane(pos, m);
			const uint32_t cnd = rdlane(cand, m);

			/* ---- FindMatchLength(candidate + 4, ip + 4, ip_end), :578 ---- */
			const uint32_t ma = cnd + 4, mb = base + 4, L = n - mb;
			uint32_t done = 0, matched;
			for (;;) {
				const uint32_t o = done + lane * 8;
				uint32_t mm = 0;
				bool term = true;
				if (o < L) {
					const uint64_t x = lds_rd64(win32, wbase + ma + o) ^ lds_rd64(win32, wbase + mb + o);
					mm = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
					mm = min(mm, L - o);
					term = mm < 8 || o + 8 >= L;
				}
				const uint64_t tmask = __ballot(term);
				if (tmask) {
					const uint32_t t = first_lane(tmask);
					matched = 4 + done + 8 * t + rdlane(mm, t);
					break;
				}
				done += 512;
			}

			/* queue {literal [next_emit, base), copy(offset, matched)} */
			if (lane == 0) {
				evq[2 * nev] = next_emit | ((base - next_emit) << 16);
				evq[2 * nev + 1] = (base - cnd) | (matched << 16);
			}
			nev++;
			ip = base + matched;
			next_emit = ip;
			if (nev == 64) {
				__syncthreads();
				flush();
			}
			if (ip >= ip_limit)
				break; /* :585-586 */
			spec = 2; /* :587-594: insert ip-1, probe ip; then the scan restarts at ip+1 (:596-598) */
			s = ip + 1;
			qi = 0;
		}
	}

	/* emit_remainder, csnappy_compress.c:600-605 */
	if (next_emit < n) {
		if (lane == 0) {
			evq[2 * nev] = next_emit | ((n - next_emit) << 16);
			evq[2 * nev + 1] = 0;
		}
		nev++;
	}
	__syncthreads();
	if (nev)
		flush();

	if (lane == 0) {
		A.frag_len[id] = op;
		if (A.fpb == 1)
			A.out_len[blk] = hdr + op;
	}
}

/* ==========================================================================================
 * STITCH: move fragments 1.. of every block behind fragment 0 and write the block length.
 * (The reference gets this for free from its sequential pointer chain, csnappy_compress.c:647-653.)
 * ======================================================================================== */
extern "C" __global__ void __launch_bounds__(256) snappy_stitch_blocks(CompressArgs A)
{
	__shared__ uint32_t red[4];
	const uint32_t tid = threadIdx.x;
	const uint32_t id = blockIdx.x;
	const uint32_t blk = id / A.fpb, fi = id - blk * A.fpb;
	const uint32_t len = A.in_len[blk];
	const uint32_t nfr = len ? (len + kFragment - 1) / kFragment : 1;
	if (fi >= nfr)
		return;
	const uint32_t hdr = A.mode == CSNAPPY_HIP_STREAM ? varint_len(len) : 0;
	const uint32_t *fl = A.frag_len + (uint64_t)blk * A.fpb;
	/* prefix = sum of the lengths of fragments before mine */
	uint32_t part = 0;
	for (uint32_t j = tid; j < fi; j += 256)
		part += fl[j];
	for (int d = 32; d; d >>= 1)
		part += (uint32_t)__shfl_down((int)part, d, 64);
	if ((tid & 63) == 0)
		red[tid >> 6] = part;
	__syncthreads();
	const uint32_t prefix = red[0] + red[1] + red[2] + red[3];
	const uint32_t mylen = fl[fi];
	if (fi > 0) {
		const uint8_t *s = A.scratch + (uint64_t)(blk * (uint64_t)(A.fpb - 1) + (fi - 1)) * kScratchSlot;
		uint8_t *d = A.out + A.out_off[blk] + hdr + prefix;
		/* head bytes until d is 4-aligned, then dwords assembled from aligned source dwords */
		const uint32_t head = min(mylen, (uint32_t)((4 - (reinterpret_cast<uintptr_t>(d) & 3)) & 3));
		if (tid < head)
			d[tid] = s[tid];
		const uint32_t words = (mylen - head) >> 2;
		const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s);
		uint32_t *d32 = reinterpret_cast<uint32_t *>(d + head);
		const uint32_t sh = head & 3, sd = head >> 2;
		for (uint32_t k = tid; k < words; k += 256)
			d32[k] = __builtin_amdgcn_alignbyte(s32[sd + k + 1], s32[sd + k], sh);
		const uint32_t tail = head + 4 * words;
		if (tail + tid < mylen)
			d[tail + tid] = s[tail + tid];
	}
	if (fi == nfr - 1 && tid == 0)
		A.out_len[blk] = hdr + prefix + mylen;
}

/* ==========================================================================================
 * DECOMPRESS: one wave per block
 * ======================================================================================== */
extern "C" __global__ void __launch_bounds__(64) snappy_decompress_blocks(DecompressArgs A)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t blk = blockIdx.x;
	const uint8_t *src = A.in + A.in_off[blk];
	const uint32_t n = A.in_len[blk];
	uint8_t *dst = A.out + A.out_off[blk];
	const uint32_t cap = A.out_cap[blk];

	uint64_t ip = 0;
	uint64_t limit = cap;
	int32_t status = CSNAPPY_E_OK;

	if (A.mode == CSNAPPY_HIP_STREAM) {
		/* csnappy_get_uncompressed_length, csnappy_decompress.c:45-71 (every lane runs the
		 * same scalar loop) and the -2 check of csnappy_decompress, :408-409 */
		uint32_t olen = 0, shift = 0, k = 0;
		for (;;) {
			if (shift >= 32 || k == n) {
				status = CSNAPPY_E_HEADER_BAD;
				break;
			}
			const uint32_t c = src[k++];
			olen |= (c & 0x7f) << shift;
			if (c < 128)
				break;
			shift += 7;
		}
		if (status == CSNAPPY_E_OK && olen > cap)
			status = CSNAPPY_E_OUTPUT_INSUF;
		ip = k;
		limit = olen;
	}

	uint64_t op = 0; /* bytes produced */
	while (status == CSNAPPY_E_OK && ip < n) {
		/* ---- every lane decodes the byte at ip+lane as if it were a tag ---- */
		const uint64_t at = ip + lane;
		uint32_t b0 = 0, tr = 0;
		if (at < n)
			b0 = src[at];
#pragma unroll
		for (int k = 0; k < 4; ++k)
			if (at + 1 + k < n)
				tr |= (uint32_t)src[at + 1 + k] << (8 * k);
		const uint32_t kind = b0 & 3;
		uint32_t l, extra, off = 0;
		if (kind == 0) {
			l = (b0 >> 2) + 1;
			extra = l > 60 ? l - 60 : 0;
			if (extra)
				l = (extra == 4 ? tr : (tr & ((1u << (8 * extra)) - 1))) + 1;
		} else if (kind == 1) {
			l = 4 + ((b0 >> 2) & 7);
			extra = 1;
			off = ((b0 >> 5) << 8) | (tr & 0xff);
		} else {
			l = (b0 >> 2) + 1;
			extra = kind == 2 ? 2 : 4;
			off = kind == 2 ? (tr & 0xffff) : tr;
		}
		const uint32_t hsz = 1 + extra;

		/* ---- walk the real tag chain on the scalar unit ---- */
		uint64_t tmask = 0;
		uint64_t cur = 0;
		while (cur < 64 && ip + cur < n) {
			const uint32_t cl = (uint32_t)cur;
			tmask |= 1ull << cl;
			cur += rdlane(hsz, cl) + (rdlane(kind, cl) == 0 ? (uint64_t)rdlane(l, cl) : 0ull);
		}
		const bool istag = (tmask >> lane) & 1;

		/* ---- per-element checks, in the reference's order (Appendix C of SURVEY.md) ---- */
		const bool trunc = at + hsz > n; /* header bytes cut off: reference is undefined, we say -5 */
		const uint64_t avail = trunc ? 0 : n - (at + hsz);
		const bool lit_short = kind == 0 && (int32_t)l >= 0 && avail < l; /* :374-375 */
		const bool lit_neg = kind == 0 && (int32_t)l < 0;
		const bool inbad = trunc || lit_short || lit_neg;
		const uint32_t eff = (istag && !inbad) ? l : 0;
		uint32_t total;
		const uint32_t excl = wave_excl_scan(eff, lane, &total);
		const uint64_t pb = op + excl; /* bytes produced before this element */
		int32_t err = 0;
		if (istag) {
			const bool overrun = limit - pb < (uint64_t)l;
			if (trunc)
				err = CSNAPPY_E_DATA_MALFORMED;
			else if (kind == 0)
				err = lit_short ? CSNAPPY_E_DATA_MALFORMED
				      : overrun ? CSNAPPY_E_OUTPUT_OVERRUN /* :288-289, :274-275 */
				      : lit_neg ? CSNAPPY_E_DATA_MALFORMED
						: 0;
			else
				err = (off == 0 || (uint64_t)off > pb) ? CSNAPPY_E_DATA_MALFORMED /* :301-303 */
				      : overrun ? CSNAPPY_E_OUTPUT_OVERRUN		       /* :311-312 */
						: 0;
		}
		const uint64_t emask = __ballot(err != 0);
		const uint32_t fe = emask ? first_lane(emask) : 64;

		/* ---- execute the elements before the first failing one, in order ---- */
		uint64_t run = fe < 64 ? (tmask & ((1ull << fe) - 1)) : tmask;
		uint64_t done_bytes = 0;
		for (; run; run &= run - 1) {
			const uint32_t t = first_lane(run);
			const uint32_t k = rdlane(kind, t), L = rdlane(l, t);
			const uint64_t PB = op + rdlane(excl, t);
			if (k == 0) {
				/* SAW__Append / SAW__AppendFastPath, csnappy_decompress.c:264-293 */
				const uint8_t *ps = src + ip + t + rdlane(hsz, t);
				for (uint32_t j = lane; j < L; j += 64)
					dst[PB + j] = ps[j];
			} else {
				/* SAW__AppendFromSelf, :295-317; dst[i] = dst[i - offset] in order
				 * (:200-206) == replicate the `offset`-byte pattern that precedes PB */
				const uint32_t OFF = rdlane(off, t);
				if (lane < L) {
					const uint32_t j = lane < OFF ? lane : lane % OFF;
					dst[PB + lane] = dst[PB - OFF + j];
				}
			}
			done_bytes = (uint64_t)rdlane(excl, t) + L;
		}
		op += done_bytes;
		if (fe < 64) {
			status = (int32_t)rdlane((uint32_t)err, fe);
			break;
		}
		ip += cur;
	}

	if (lane == 0) {
		A.status[blk] = status;
		A.produced[blk] = status == CSNAPPY_E_OK ? (uint32_t)op : 0;
	}
}

/* ==========================================================================================
 * COMPACT: pack the slot-strided outputs into one dense stream (what the reference's callers do
 * with memcpy after each call; needed before the multi-GPU gather)
 * ======================================================================================== */
extern "C" __global__ void __launch_bounds__(256)
snappy_compact_stream(const uint8_t *out, const uint64_t *out_off, const uint32_t *out_len,
		      const uint64_t *dense_off, uint8_t *dense)
{
	const uint32_t blk = blockIdx.x, tid = threadIdx.x;
	const uint8_t *s = out + out_off[blk];
	uint8_t *d = dense + dense_off[blk];
	const uint32_t n = out_len[blk];
	/* destination-aligned dwords assembled from (possibly misaligned) source bytes */
	const uint32_t head = min(n, (uint32_t)((4 - (reinterpret_cast<uintptr_t>(d) & 3)) & 3));
	if (tid < head)
		d[tid] = s[tid];
	const uint32_t words = (n - head) >> 2;
	const uint8_t *sb = s + head;
	const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(sb) & 3);
	const uint32_t *s32 = reinterpret_cast<const uint32_t *>(sb - sh);
	uint32_t *d32 = reinterpret_cast<uint32_t *>(d + head);
	for (uint32_t k = tid; k < words; k += 256)
		d32[k] = sh ? __builtin_amdgcn_alignbyte(s32[k + 1], s32[k], sh) : s32[k];
	const uint32_t tail = head + 4 * words;
	if (tail + tid < n)
		d[tail + tid] = s[tail + tid];
}

/* ==========================================================================================
 * workload generator kernel (bench/test input; see workload_gen.h)
 * ======================================================================================== */
extern "C" __global__ void __launch_bounds__(64)
workload_generate(int kind, uint64_t seed, uint64_t first_block, uint32_t nblocks, uint32_t block_len,
		  uint8_t *out)
{
	const uint32_t b = blockIdx.x * 64 + threadIdx.x;
	if (b < nblocks)
		wg_fill_block(kind, seed, first_block + b, out + (uint64_t)b * block_len, block_len);
}

/* ------------------------------------------------------------------------------------------
 * host side of the C-ABI
 * ---------------------------------------------------------------------------------------- */
thread_local char g_last_error[256] = "";

bool hip_ok(hipError_t e, const char *what)
{
	if (e == hipSuccess)
		return true;
	snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
	return false;
}

/* Per-kernel timing for bench.py: event pairs are recorded on the launch stream around each
 * kernel (no host synchronisation in the launch path) and resolved when the totals are read. */
struct Pending {
	int slot;
	hipEvent_t a, b;
};
bool g_timing = false;
Pending g_pending[4096];
int g_npending = 0;

struct Timer {
	hipStream_t st;
	hipEvent_t a = nullptr, b = nullptr;
	bool on;
	explicit Timer(hipStream_t s) : st(s), on(g_timing && g_npending < 4096) {}
	void start()
	{
		if (!on)
			return;
		(void)hipEventCreate(&a);
		(void)hipEventCreate(&b);
		(void)hipEventRecord(a, st);
	}
	void stop(int slot)
	{
		if (!on)
			return;
		(void)hipEventRecord(b, st);
		g_pending[g_npending++] = Pending{ slot, a, b };
		on = g_timing && g_npending < 4096;
	}
};

uint32_t frags_per_block(uint32_t max_in_len)
{
	return max_in_len ? (max_in_len + kFragment - 1) / kFragment : 1;
}

} // namespace

extern "C" {

int csnappy_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

const char *csnappy_hip_last_error(void)
{
	return g_last_error;
}

void csnappy_hip_set_kernel_timing(int enable)
{
	g_timing = enable != 0;
}

void csnappy_hip_get_kernel_timing(float ms[4], uint32_t launches[4])
{
	for (int i = 0; i < 4; ++i) {
		ms[i] = 0;
		launches[i] = 0;
	}
	for (int i = 0; i < g_npending; ++i) {
		float t = 0;
		(void)hipEventSynchronize(g_pending[i].b);
		(void)hipEventElapsedTime(&t, g_pending[i].a, g_pending[i].b);
		ms[g_pending[i].slot] += t;
		launches[g_pending[i].slot]++;
		(void)hipEventDestroy(g_pending[i].a);
		(void)hipEventDestroy(g_pending[i].b);
	}
	g_npending = 0;
}

size_t csnappy_hip_compress_workspace_size(uint32_t nblocks, uint32_t max_in_len)
{
	const uint64_t fpb = frags_per_block(max_in_len);
	const uint64_t fl = ((uint64_t)nblocks * fpb * sizeof(uint32_t) + 255) & ~255ull;
	return (size_t)(fl + (uint64_t)nblocks * (fpb - 1) * kScratchSlot + 256);
}

int csnappy_hip_compress_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
			       uint32_t nblocks, uint32_t max_in_len, void *d_out,
			       const uint64_t *d_out_off, uint32_t *d_out_len, int p, int mode,
			       void *d_workspace, size_t workspace_bytes, void *stream)
{
	if (p < 9 || p > 16 || (mode != CSNAPPY_HIP_STREAM && mode != CSNAPPY_HIP_FRAGMENT))
		return CSNAPPY_HIP_E_ARG;
	if (mode == CSNAPPY_HIP_FRAGMENT && max_in_len > kFragment)
		return CSNAPPY_HIP_E_ARG;
	if (nblocks == 0)
		return 0;
	if (workspace_bytes < csnappy_hip_compress_workspace_size(nblocks, max_in_len) ||
	    (reinterpret_cast<uintptr_t>(d_workspace) & 255))
		return CSNAPPY_HIP_E_WORKSPACE;
	const uint32_t fpb = frags_per_block(max_in_len);
	if ((uint64_t)nblocks * fpb > 0x7fffffffull)
		return CSNAPPY_HIP_E_ARG;
	hipStream_t st = static_cast<hipStream_t>(stream);

	CompressArgs A;
	A.in = static_cast<const uint8_t *>(d_in);
	A.in_off = d_in_off;
	A.in_len = d_in_len;
	A.out = static_cast<uint8_t *>(d_out);
	A.out_off = d_out_off;
	A.out_len = d_out_len;
	A.frag_len = static_cast<uint32_t *>(d_workspace);
	const uint64_t fl = ((uint64_t)nblocks * fpb * sizeof(uint32_t) + 255) & ~255ull;
	A.scratch = static_cast<uint8_t *>(d_workspace) + fl;
	A.nblocks = nblocks;
	A.fpb = fpb;
	A.win_bytes = ((max_in_len < kFragment ? max_in_len : kFragment) + 16 + 16 + 63) & ~63u;
	A.s_entries = (1u << (p - 1)) < 2048u ? (1u << (p - 1)) : 2048u;
	A.p = p;
	A.mode = mode;

	const size_t lds = (size_t)A.win_bytes + ((size_t)1 << p) + (size_t)A.s_entries * 4 + 64 * 8;
	if (!hip_ok(hipFuncSetAttribute(reinterpret_cast<const void *>(snappy_compress_fragments),
					hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
		    "hipFuncSetAttribute"))
		return CSNAPPY_HIP_E_RUNTIME;
	Timer t(st);
	t.start();
	hipLaunchKernelGGL(snappy_compress_fragments, dim3(nblocks * fpb), dim3(64), lds, st, A);
	t.stop(0);
	if (!hip_ok(hipGetLastError(), "launch snappy_compress_fragments"))
		return CSNAPPY_HIP_E_RUNTIME;
	if (fpb > 1) {
		t.start();
		hipLaunchKernelGGL(snappy_stitch_blocks, dim3(nblocks * fpb), dim3(256), 0, st, A);
		t.stop(1);
		if (!hip_ok(hipGetLastError(), "launch snappy_stitch_blocks"))
			return CSNAPPY_HIP_E_RUNTIME;
	}
	return 0;
}

int csnappy_hip_decompress_batch(const void *d_in, const uint64_t *d_in_off,
				 const uint32_t *d_in_len, uint32_t nblocks, void *d_out,
				 const uint64_t *d_out_off, const uint32_t *d_out_cap,
				 int32_t *d_status, uint32_t *d_produced, int mode, void *stream)
{
	if (mode != CSNAPPY_HIP_STREAM && mode != CSNAPPY_HIP_FRAGMENT)
		return CSNAPPY_HIP_E_ARG;
	if (nblocks == 0)
		return 0;
	hipStream_t st = static_cast<hipStream_t>(stream);
	DecompressArgs A;
	A.in = static_cast<const uint8_t *>(d_in);
	A.in_off = d_in_off;
	A.in_len = d_in_len;
	A.out = static_cast<uint8_t *>(d_out);
	A.out_off = d_out_off;
	A.out_cap = d_out_cap;
	A.status = d_status;
	A.produced = d_produced;
	A.nblocks = nblocks;
	A.mode = mode;
	Timer t(st);
	t.start();
	hipLaunchKernelGGL(snappy_decompress_blocks, dim3(nblocks), dim3(64), 0, st, A);
	t.stop(2);
	if (!hip_ok(hipGetLastError(), "launch snappy_decompress_blocks"))
		return CSNAPPY_HIP_E_RUNTIME;
	return 0;
}

int csnappy_hip_compact_batch(const void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_len,
			      const uint64_t *d_dense_off, uint32_t nblocks, void *d_dense, void *stream)
{
	if (nblocks == 0)
		return 0;
	hipLaunchKernelGGL(snappy_compact_stream, dim3(nblocks), dim3(256), 0,
			   static_cast<hipStream_t>(stream), static_cast<const uint8_t *>(d_out), d_out_off,
			   d_out_len, d_dense_off, static_cast<uint8_t *>(d_dense));
	if (!hip_ok(hipGetLastError(), "launch snappy_compact_stream"))
		return CSNAPPY_HIP_E_RUNTIME;
	return 0;
}

int csnappy_hip_workload_generate(int kind, uint64_t seed, uint64_t first_block, uint32_t nblocks,
				  uint32_t block_len, void *d_out, void *stream)
{
	if (kind < 0 || kind > 2)
		return CSNAPPY_HIP_E_ARG;
	if (nblocks == 0)
		return 0;
	hipLaunchKernelGGL(workload_generate, dim3((nblocks + 63) / 64), dim3(64), 0,
			   static_cast<hipStream_t>(stream), kind, seed, first_block, nblocks, block_len,
			   static_cast<uint8_t *>(d_out));
	if (!hip_ok(hipGetLastError(), "launch workload_generate"))
		return CSNAPPY_HIP_E_RUNTIME;
	return 0;
}

} /* extern "C" */
