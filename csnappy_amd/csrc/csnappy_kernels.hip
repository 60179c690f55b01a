/*
 * csnappy_kernels.hip -- Snappy raw-block codec for MI355X (gfx950, wave64), from scratch.
 *
 * What the kernels replace in the reference (file:line into the reference tree):
 *   snappy_parse_fragments*     csnappy_compress_fragment, the probe loop   csnappy_compress.c:469-606
 *                               Hash/HashBytes                              :228-236
 *                               FindMatchLength                             :252-295
 *                               fragment loop + table-size pick             :636-654
 *   snappy_emit_sizes/_bases/_blocks, snappy_emit_pages
 *                               EmitLiteral / EmitCopy(LessThan64)          :332-415
 *                               encode_varint32 + the `compressed = p` chain :46-73, :633-651
 *   snappy_decompress_blocks    csnappy_decompress_noheader                 csnappy_decompress.c:319-387
 *                               SAW__Append* / IncrementalCopy*             :200-317
 *                               csnappy_get_uncompressed_length / csnappy_decompress :45-71,394-411
 *   snappy_stream_*             the same two calls on ONE long stream: a tag index finds the
 *                               fragments the compressor's 32 KiB restarts left (:585-616), which
 *                               snappy_decompress_blocks then decodes in parallel
 *   snappy_compact_stream       the caller's memcpy of each block behind the last
 *                               (block_compressor.c:316-334)
 *   snappy_crc32c_blocks        masked CRC-32C of the framing format (include/csnappy_frame.h)
 *
 * Design (DESIGN.md has the long form):
 *   - compress: one wave per 32 KiB fragment reproduces the reference's sequential probe loop
 *     exactly, 64 consecutive positions per step: every lane hashes its 4 bytes, gathers the
 *     table entry, measures a lane-local match length against its candidate; a lane that shares
 *     a hash slot with an earlier lane of the step (the only way its table read could be stale;
 *     found with one returning LDS add on the entry, whose lanes the LDS serves in order)
 *     is flagged and resolved from that lane's registers if the chain of copies -- followed on
 *     the scalar unit -- ever probes it.  The table lives in LDS, indexed by dense bucket ids a
 *     prologue of the same wave assigns (only slots hit twice can matter); the window is read
 *     where it lies.  The wave writes 8-byte (literal, copy) records; a separate kernel turns
 *     the records of a block into bytes, all fragments at their final offsets.
 *   - decompress: one wave per block, two phases: 64 candidate tag positions are decoded in
 *     parallel and the true tag chain is walked on the scalar unit until 64 elements are queued;
 *     then one lane per element: output offsets from a DPP prefix sum, errors resolved in
 *     element order, literals and independent copies in parallel, copies that read the batch's
 *     own output in order, all assembled in LDS and flushed with aligned 16 B stores.
 *   - no MFMA: this is byte/integer work bound by instruction issue, latency and cache-line
 *     gathers, not a contraction.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/csnappy.h"
#include "../../include/csnappy_hip.h"
#include "../../include/csnappy_frame.h"
#include "workload_gen.h"

namespace {

constexpr uint32_t kFragment = 32768;    /* kBlockSize, csnappy_compress.c:85-86 */
constexpr uint32_t kMargin = 15;         /* kInputMarginBytes, csnappy_compress.c:468 */
constexpr uint32_t kHashMul = 0x1e35a7bdu; /* csnappy_compress.c:230 */

#ifndef CSNAPPY_PARSE_NOSPILLSTORE
#define CSNAPPY_PARSE_NOSPILLSTORE 0
#endif
#ifndef CSNAPPY_TIMING_TA
#define CSNAPPY_TIMING_TA 0
#endif
#ifndef CSNAPPY_FAST
#define CSNAPPY_FAST 1 /* 0: no hand-written fast path in the dense parser (A/B, and the reference for its logic) */
#endif
#define DEVINL __device__ __forceinline__

struct CompressArgs {
	const uint8_t *in;
	const uint64_t *in_off;
	const uint32_t *in_len;
	uint8_t *out;
	const uint64_t *out_off;
	uint32_t *out_len;
	uint64_t *recs;     /* rec_cap 8-byte records per fragment of the chunk */
	uint32_t *rec_cnt;  /* records of each fragment of the chunk (kNoRecords: not parsed yet) */
	uint8_t *tabs;      /* tab_stride bytes per fragment: dense ids, or the global-memory table */
	unsigned long long *prof; /* debug cycle counters (PROF instantiations only) */
	uint32_t blk_base;  /* first block of this chunk */
	uint32_t fpb;       /* fragments per block (upper bound) */
	uint32_t rec_cap;
	uint32_t tab_stride;
	uint32_t lds0;      /* global-table kernel: LDS bytes in front of its keyed array (the occupancy bitmap) */
	uint32_t dense_cap; /* entries of the dense LDS table */
	uint32_t spill_cap; /* buckets beyond those: a per-fragment table in HBM behind the ids (0 = none) */
	uint32_t spill_off; /* its byte offset inside the fragment's `tabs` region */
	uint32_t s_entries; /* global-table kernel: keys of the array its lanes find slot sharing with (power of two) */
	uint32_t only_unparsed; /* skip fragments that already have records (second and later launches) */
	uint32_t max_in_len; /* the caller's bound on in_len[]: a longer block is refused (out_len = 0xffffffff) */
	uint32_t emit_wave_per_block, emit_blocks; /* emit: one wave per block (fpb == 1, small blocks) */
	uint32_t sample_min;    /* TAB_LDS_DENSE: full fragments with fewer distinct sampled hashes go to TAB_GLOBAL */
	uint32_t no_isa;        /* 1: every step takes parse_lean's C++ (CSNAPPY_HIP_NO_ISA=1; the hand-written loops' reference) */
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
	const uint32_t *skip_if; /* not null and *skip_if != 0: nothing to do (the stream fast path's result stands) */
};

DEVINL uint32_t rdlane(uint32_t v, uint32_t l)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}

/* number of equal leading bytes (0..16) of two 16-byte strings given as the XOR of their halves;
 * branch-free (a lane-divergent branch here would sit next to the joins of the parser's cursor
 * state and drag it off the scalar unit) */
DEVINL uint32_t common_prefix16(uint64_t xlo, uint64_t xhi)
{
	/* (the x ? ctz(x) : 64 form is what the compiler turns into v_ffbl + clamped add + min3) */
	const uint32_t bl = xlo ? (uint32_t)__builtin_ctzll(xlo) : 64u;
	const uint32_t bh = xhi ? (uint32_t)__builtin_ctzll(xhi) : 64u;
	return (bl < 64u ? bl : 64u + bh) >> 3;
}

/* lanes whose predicate holds (the builtin takes the i1 itself: HIP's __ballot(int) makes the compiler
 * materialise 0/1 in a VGPR and compare it again) */
DEVINL uint64_t ballot64(bool p)
{
	return __builtin_amdgcn_ballot_w64(p);
}

DEVINL uint32_t first_lane(uint64_t m)
{
	return (uint32_t)__builtin_ctzll(m);
}

/* inclusive prefix sum across the 64 lanes with DPP row shifts / row broadcasts (no LDS) */
template <int CTRL, int ROW_MASK> DEVINL uint32_t dpp_add(uint32_t x)
{
	return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}

DEVINL uint32_t wave_incl_scan_dpp(uint32_t x)
{
	x = dpp_add<0x111, 0xf>(x); /* row_shr:1 */
	x = dpp_add<0x112, 0xf>(x); /* row_shr:2 */
	x = dpp_add<0x114, 0xf>(x); /* row_shr:4 */
	x = dpp_add<0x118, 0xf>(x); /* row_shr:8 */
	x = dpp_add<0x142, 0xa>(x); /* row_bcast:15 -> rows 1 and 3 */
	x = dpp_add<0x143, 0xc>(x); /* row_bcast:31 -> rows 2 and 3 */
	return x;
}

/* the same with max (values >= 0: lanes a shift leaves without a source contribute 0) */
template <int CTRL, int ROW_MASK> DEVINL uint32_t dpp_max(uint32_t x)
{
	return max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false));
}

DEVINL uint32_t wave_incl_max_dpp(uint32_t x)
{
	x = dpp_max<0x111, 0xf>(x); /* row_shr:1 */
	x = dpp_max<0x112, 0xf>(x); /* row_shr:2 */
	x = dpp_max<0x114, 0xf>(x); /* row_shr:4 */
	x = dpp_max<0x118, 0xf>(x); /* row_shr:8 */
	x = dpp_max<0x142, 0xa>(x); /* row_bcast:15 -> rows 1 and 3 */
	x = dpp_max<0x143, 0xc>(x); /* row_bcast:31 -> rows 2 and 3 */
	return x;
}

/* lane L receives lane L-1's value, lane 0 receives 0 (wave_shr:1) */
DEVINL uint32_t wave_shr1(uint32_t x)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false);
}

/* exclusive prefix sum across the 64 lanes; *total receives the wave sum */
DEVINL uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t *total)
{
	(void)lane;
	const uint32_t x = wave_incl_scan_dpp(v);
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
	/* short last fragment of a stream: the smallest ws in 9..p with 2^(ws-1) >= n */
	if (mode != CSNAPPY_HIP_STREAM || n >= kFragment || n <= 256)
		return (mode == CSNAPPY_HIP_STREAM && n <= 256) ? 9 : p;
	const int ws = 33 - __builtin_clz(n - 1);
	return ws < p ? ws : p;
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
 * COMPRESS, part 1 of 2: snappy_parse_fragments -- ONE wave per 32 KiB fragment
 *
 * The wave reproduces the reference's sequential probe loop exactly (csnappy_compress.c:469-606)
 * and writes what it decided as 8-byte (literal, copy) records to HBM; the emit launches turn
 * the records into bytes.  The parser is one dependent chain per fragment, so throughput is
 * (fragments in flight per CU) / (latency of a step): everything here is about keeping the
 * per-fragment LDS footprint and the step short.
 *
 * Hash table.  The reference's table has 2^(p-1) uint16 slots (64 KiB at p=16), far more than
 * a fragment can use: a slot matters only if at least two positions of the fragment hash to it
 * (a lone position can neither find a candidate nor be found).  A prologue therefore numbers
 * the slots that are hit twice or more (two LDS bitmaps filled with atomicOr, a popcount
 * prefix) and writes every position's dense bucket id to HBM; the table the parser then keeps in
 * LDS has one uint16 entry per such bucket (~4.6 k on URL-like text, ~1 k on runs) and is indexed
 * by the id, which the lanes load with their 16 input bytes.  Same slots, same contents, same
 * order of updates as the reference's table -- only the slots nobody can ever read are gone.
 * The LDS table has 5 120 entries (10 KiB = 16 fragments per CU); a fragment
 * with up to 2 048 buckets more keeps those in a small table in HBM behind its ids (SPILL), and
 * one with still more is handed to a second launch that keeps the full 2^p-byte table in global
 * memory (TAB_GLOBAL); tables of <= 8 KiB are simply indexed by the hash (TAB_LDS_HASH, no
 * prologue).  The window is never staged: the input is read where it lies.
 *
 * Step logic (restated lane by lane in tests/wave_model.py::compress_fragment_v4 and fuzzed
 * against the CPU checker):
 *   dense step   the 64 lanes take 64 CONSECUTIVE positions starting at the cursor.  Every lane
 *                looks its slot up and computes a lane-local match length (up to kLocalMatch
 *                bytes) against its candidate.  A lane that shares its slot with an earlier
 *                lane of the step is flagged and resolved from that lane's registers if the chain
 *                gets there.  The chain of matches through the step -- match at lane i of
 *                length L, insert lane i+L-1, re-match probe at lane i+L, 32 stride-1 scan probes
 *                after it (csnappy_compress.c:535-598) -- is walked on the scalar unit, so
 *                one step usually retires several copies.
 *   sparse step  once a scan has made 32 probes without a match the reference strides by 2, 3..
 *                (:542); lanes then take the next 64 probe positions of that stride rule and a
 *                step ends at its first match.
 * ======================================================================================== */
constexpr uint32_t kLocalMatch = 16;  /* lane-local match length cap */
constexpr uint32_t kBigRecord = 32;   /* records encoding to more than this bypass the staging (but see kMediumLiteral) */
constexpr uint32_t kMediumLiteral = 256; /* longest literal staged by the wave for its lane */
#ifndef CSNAPPY_STAGE_CAP
#define CSNAPPY_STAGE_CAP (64 * kBigRecord)
#endif
constexpr uint32_t kStageCap = CSNAPPY_STAGE_CAP; /* bytes an emit wave stages before it drains: a chunk of small records fits (4 / 8 KiB: +-1 %, round 5) */
static_assert(kStageCap >= 64 * kBigRecord, "a chunk of small records fits the staging");
constexpr uint32_t kStageBytes = 16 + kStageCap + 16 + 32; /* LDS output staging of one emit wave */
constexpr uint32_t kNoRecords = 0xffffffffu;  /* rec_cnt: "not parsed yet: more buckets than this launch's dense table" */
constexpr uint32_t kWantGlobal = 0xfffffffeu; /* rec_cnt: "not parsed yet: repetitive, take the global-table launch" */
constexpr uint32_t kNoBucket = 0;            /* dense id of a position whose slot nobody else hits */
constexpr uint32_t kFirstBucket = 2;         /* ... and of the first bucket: entries 0 and 1 of the dense table are one dummy
                                              * dword that the lanes without a bucket may read, bump and overwrite at will */
constexpr uint32_t kSpillFilterEntries = 128; /* the spilled lanes' conflict filter (u32 tags) ... */
constexpr uint32_t kSpillFilterSlots = kSpillFilterEntries * 2; /* ... takes the place of this many table entries */

enum { TAB_LDS_HASH = 0, TAB_LDS_DENSE = 1, TAB_GLOBAL = 2 };

/* Orders this wave's LDS accesses for the compiler.  The LDS pipeline executes one wave's
 * instructions in issue order, so cross-lane exchange inside a wave needs no hardware wait. */
DEVINL void wave_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

/* record: x = base | cand << 16, y = copy_len | lit_start << 16 (every field <= 32768) */
DEVINL uint2 pack_record(uint32_t lit_start, uint32_t base, uint32_t cnd, uint32_t clen)
{
	return make_uint2(base | (cnd << 16), clen | (lit_start << 16));
}

/* ==========================================================================================
 * The step loop (round 3; round 2's, with three cursor cases and a per-hop state machine, spent
 * 230 scalar instructions per 64-position step on the CU's ONE scalar unit):
 *   - one cursor form.  The state between steps is (s, q1): the scan that starts at position s
 *     has made q1 - 1 probes; q1 == 0 says "a copy just ended at ip = s - 1: insert ip - 1,
 *     probe ip" (csnappy_compress.c:585-594), and that re-match probe is simply scan index -1
 *     (33 stride-1 probes follow a copy, 32 the start of the fragment, :535-552).  Lane 0 of a
 *     dense step is ALWAYS an insert-only lane -- ip - 1 after a copy, otherwise the last position
 *     the previous step probed, whose re-insertion changes nothing.
 *   - the lanes in front of the scan limit are arithmetic, not a ballot.
 *   - the chain loop has one exit code; the last copy and the window it leaves are derived from
 *     the `taken` mask after the walk instead of being tracked through it; a wide match that
 *     leaves the step is a link of the chain like any other (its own lane writes its record).
 *   - a visit of a flagged lane tests for a real slot-sharer first (one readlane, one compare);
 *     the insert mask is only built when there is one.
 *   - block descriptors are read once, before the prologue's stores, so they stay scalar loads.
 * ======================================================================================== */
/* Tags of the small keyed arrays two kinds of lanes still find slot sharing with (parse_lean, "TW": the
 * lanes of a table in LDS find it through the table itself).  A tag is {epoch, slot, lane}; the epoch
 * counts down, so tags of earlier steps never win an atomicMin and nothing is cleared between steps.
 *   - the lanes of a fragment whose bucket lies in the HBM spill-over: one array, atomicMin + read.  The
 *     smallest tag of a key is that of the smallest SLOT among the lanes sharing the key, lowest lane
 *     first: a lane that reads its own slot back knows exactly whether a lower lane shares it; a lane
 *     that reads another slot is flagged to be safe (they are few: that is rare);
 *   - the global-table kernel: one returning exchange (see there).
 * SB = bits of a slot: 13 for dense ids, 15 for hashes; the epoch gets what is left of 32 bits beside
 * them and 7 bits of lane. */
template <int SB> struct FilterTag {
	static constexpr uint32_t kSlots = (1u << SB) - 1;         /* largest slot */
	static constexpr uint32_t kEpochs = (1u << (25 - SB)) - 1; /* steps between two clearings of the array */
	static DEVINL uint32_t tag(uint32_t epoch, uint32_t slot, uint32_t vlane, bool takes_part)
	{
		return takes_part ? (epoch << (7 + SB)) | (slot << 7) | vlane : ~0u;
	}
	/* e: what the array holds for my key after the step's atomicMin */
	static DEVINL bool flags(uint32_t e, uint32_t slot, uint32_t vlane)
	{
		const bool settled = ((e >> 7) & kSlots) == slot;
		/* (lane 0 has no lower lane; a sparse step that is cut in front of its first lane would never end) */
		return (vlane != 0) & (settled ? (e & 127u) < vlane : true);
	}
};

struct Frag {
	const uint8_t *src; /* the fragment's input */
	uint8_t *region;    /* its `tabs` workspace region: dense ids (+ spill-over table) */
	uint2 *R;           /* its records */
	uint32_t n;         /* its bytes */
	uint32_t shift;     /* 33 - table power */
	uint32_t ws;
	uint32_t c;         /* its index in the chunk */
};

DEVINL uint64_t uni64(uint64_t v)
{
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
	       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}

/* false: nothing to parse here (no such fragment, block longer than promised, already parsed) */
DEVINL bool frag_setup(const CompressArgs &A, Frag &F, bool gtab)
{
	const uint32_t c = blockIdx.x; /* (fragments dealt to the XCDs in contiguous eighths instead of round robin: no difference, round 6) */
	const uint32_t blk = A.blk_base + c / A.fpb, fi = c % A.fpb;
	const uint32_t len = A.in_len[blk];
	const uint32_t foff = fi * kFragment;
	if ((fi > 0 && foff >= len) || len > A.max_in_len)
		return false;
	if (A.only_unparsed) {
		const uint32_t state = A.rec_cnt[c];
		if (state != kNoRecords && !(gtab && state == kWantGlobal))
			return false;
	}
	F.c = c;
	F.n = min(len - foff, kFragment);
	F.ws = (uint32_t)fragment_power(F.n, A.p, A.mode);
	F.shift = 33 - F.ws;
	/* (pointers stay derived from the kernel arguments: laundering them through integers would
	 * turn every access into a flat load, which also counts on the LDS counter) */
	F.src = A.in + uni64(A.in_off[blk] + foff);
	F.R = reinterpret_cast<uint2 *>(A.recs + (uint64_t)c * A.rec_cap);
	F.region = A.tabs + (uint64_t)c * A.tab_stride;
	return true;
}

/* The dense placement's prologue (see the header comment above): numbers the hash slots that two or
 * more positions of the fragment hit and writes every position's dense bucket id to the fragment's
 * workspace region (0 = no bucket, the buckets count from kFirstBucket).  Returns the number of ids
 * in use -- buckets + kFirstBucket, the table entries the fragment needs --, or kNoRecords when the
 * fragment was handed to a later launch (too many buckets, or repetitive: the global-table launch is
 * faster for it). */
#ifndef CSNAPPY_PROLOGUE_PAIRS
#define CSNAPPY_PROLOGUE_PAIRS 1 /* 0: the bitmap and its prefix as two arrays (A/B) */
#endif
DEVINL uint32_t dense_prologue(const CompressArgs &A, const Frag &F)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	const uint32_t lane = threadIdx.x;
	const uint32_t n = F.n, shift = F.shift;
	const uint8_t *src = F.src;
	if (n < kMargin)
		return 0;
	const uint32_t nwords = (1u << (F.ws - 1)) >> 5; /* >= 8 */
	uint32_t *seen1 = reinterpret_cast<uint32_t *>(smem), *seen2 = seen1 + nwords;
	if (A.sample_min && n == kFragment) {
		for (uint32_t k = lane; k < 256; k += 64)
			seen1[k] = 0;
		wave_lds_fence();
		/* (all 32 samples requested before the first is used: with the atomic behind each load the
		 * compiler waited for every load on its own, 32 round trips in a row) */
		uint32_t w[32];
#pragma unroll
		for (uint32_t j = 0; j < 32; ++j)
			__builtin_memcpy(&w[j], src + 16 * lane + 1024 * j, 4);
#pragma unroll
		for (uint32_t j = 0; j < 32; ++j) {
			const uint32_t h = (w[j] * kHashMul) >> 19;
			atomicOr(&seen1[h >> 5], 1u << (h & 31));
		}
		wave_lds_fence();
		uint32_t d = 0, distinct;
		for (uint32_t k = 0; k < 4; ++k)
			d += (uint32_t)__builtin_popcount(seen1[lane * 4 + k]);
		(void)wave_excl_scan(d, lane, &distinct);
		wave_lds_fence();
		if (distinct < A.sample_min) {
			if (lane == 0)
				A.rec_cnt[F.c] = kWantGlobal;
			return kNoRecords;
		}
	}
	uint16_t *pref = reinterpret_cast<uint16_t *>(seen2 + nwords);
	for (uint32_t k = lane; k < 2 * nwords; k += 64)
		seen1[k] = 0;
	wave_lds_fence();
	const uint32_t npos = n - 3; /* positions that have four bytes */
	auto load16 = [&](uint32_t i) -> uint4 {
		uint4 v = make_uint4(0, 0, 0, 0);
		if (i + 16 <= n) {
			__builtin_memcpy(&v, src + i, 16);
		} else if (i < n) {
			uint32_t w[4] = { 0, 0, 0, 0 };
			for (uint32_t k = 0; i + k < n; ++k)
				w[k >> 2] |= (uint32_t)src[i + k] << (8 * (k & 3));
			v = make_uint4(w[0], w[1], w[2], w[3]);
		}
		return v;
	};
	auto hash8 = [&](const uint4 &v, uint32_t hh[8]) {
		const uint32_t w0 = v.x, w1 = v.y, w2 = v.z;
		hh[0] = (w0 * kHashMul) >> shift;
		hh[1] = (__builtin_amdgcn_alignbyte(w1, w0, 1) * kHashMul) >> shift;
		hh[2] = (__builtin_amdgcn_alignbyte(w1, w0, 2) * kHashMul) >> shift;
		hh[3] = (__builtin_amdgcn_alignbyte(w1, w0, 3) * kHashMul) >> shift;
		hh[4] = (w1 * kHashMul) >> shift;
		hh[5] = (__builtin_amdgcn_alignbyte(w2, w1, 1) * kHashMul) >> shift;
		hh[6] = (__builtin_amdgcn_alignbyte(w2, w1, 2) * kHashMul) >> shift;
		hh[7] = (__builtin_amdgcn_alignbyte(w2, w1, 3) * kHashMul) >> shift;
	};
	auto sweep = [&](auto &&body) {
		uint4 nxt[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j)
			nxt[j] = load16(512 * j + 8 * lane);
		for (uint32_t b0 = 0; b0 < npos; b0 += 2048) {
			uint4 cur[4];
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) {
				cur[j] = nxt[j];
				if (b0 + 2048 < npos)
					nxt[j] = load16(b0 + 2048 + 512 * j + 8 * lane);
			}
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) {
				const uint32_t i = b0 + 512 * j + 8 * lane;
				if (b0 + 512 * j < npos)
					body(cur[j], i, i < npos ? min(8u, npos - i) : 0u);
			}
		}
	};
	/* Equal hashes next to each other in a lane's eight (runs: all eight) are settled in registers:
	 * one atomic for the group, and the slot is known to be hit twice.  (Without this, runs
	 * serialise on one LDS word: G_low's prologue took 2.5x the time of text's.) */
	sweep([&](const uint4 &v, uint32_t, uint32_t cntp) {
		uint32_t hh[8], old[8];
		bool first[8], twice[8];
		hash8(v, hh);
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			first[k] = k < cntp && (k == 0 || hh[k] != hh[k - 1]);
			twice[k] = k + 1 < cntp && hh[k + 1] == hh[k]; /* (on the group's members but its last) */
		}
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k)
			old[k] = first[k] ? atomicOr(&seen1[hh[k] >> 5], 1u << (hh[k] & 31)) : 0u;
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k)
			if (first[k] && (((old[k] >> (hh[k] & 31)) & 1u) || twice[k]))
				atomicOr(&seen2[hh[k] >> 5], 1u << (hh[k] & 31));
	});
	wave_lds_fence();
	const uint32_t per = max(1u, nwords >> 6);
	uint32_t mine = 0;
	for (uint32_t k = 0; k < per; ++k) {
		const uint32_t w = lane * per + k;
		if (w < nwords)
			mine += (uint32_t)__builtin_popcount(seen2[w]);
	}
	uint32_t nb;
	uint32_t run = wave_excl_scan(mine, lane, &nb) + kFirstBucket;
	nb += kFirstBucket; /* ids in use */
	/* (a fragment with more buckets than the LDS table gives the table's last kSpillFilterSlots
	 * entries to its spilled lanes' filter, see parse_lean) */
	if (nb > A.dense_cap && nb > A.dense_cap + A.spill_cap - (A.spill_cap ? kSpillFilterSlots : 0u)) {
		if (lane == 0)
			A.rec_cnt[F.c] = kNoRecords;
		return kNoRecords;
	}
#if CSNAPPY_PROLOGUE_PAIRS
	/* (round 6) the second sweep reads a word of the bitmap and its prefix for every position: the two as ONE
	 * 8-byte entry, built over the two bitmaps (the first is done with; every lane holds its words of the second in
	 * registers before the first pair is written): half the sweep's LDS reads */
	uint2 *pairs = reinterpret_cast<uint2 *>(smem);
	{
		uint32_t s2w[16]; /* per <= 16: nwords <= 1024 */
#pragma unroll
		for (uint32_t k = 0; k < 16; ++k) {
			const uint32_t w = lane * per + k;
			s2w[k] = (k < per && w < nwords) ? seen2[w] : 0u;
		}
		wave_lds_fence();
#pragma unroll
		for (uint32_t k = 0; k < 16; ++k) {
			const uint32_t w = lane * per + k;
			if (k < per && w < nwords) {
				pairs[w] = make_uint2(s2w[k], run);
				run += (uint32_t)__builtin_popcount(s2w[k]);
			}
		}
	}
	(void)pref;
#else
	for (uint32_t k = 0; k < per; ++k) {
		const uint32_t w = lane * per + k;
		if (w < nwords) {
			pref[w] = (uint16_t)run;
			run += (uint32_t)__builtin_popcount(seen2[w]);
		}
	}
#endif
	wave_lds_fence();
	uint16_t *wids = reinterpret_cast<uint16_t *>(F.region);
	sweep([&](const uint4 &v, uint32_t i, uint32_t cntp) {
		uint32_t hh[8], id[8];
		hash8(v, hh);
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
#if CSNAPPY_PROLOGUE_PAIRS
			const uint2 pr = pairs[hh[k] >> 5];
			const uint32_t s2 = pr.x, pf = pr.y;
#else
			const uint32_t s2 = seen2[hh[k] >> 5], pf = pref[hh[k] >> 5];
#endif
			const uint32_t b = 1u << (hh[k] & 31);
			id[k] = (k < cntp && (s2 & b)) ? pf + (uint32_t)__builtin_popcount(s2 & (b - 1)) : kNoBucket;
		}
		if (cntp == 8) {
#if CSNAPPY_NT & 1 /* experiment: the ids leave the caches behind them */
			typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
			const u32x4 pk = { id[0] | (id[1] << 16), id[2] | (id[3] << 16), id[4] | (id[5] << 16), id[6] | (id[7] << 16) };
			__builtin_nontemporal_store(pk, reinterpret_cast<u32x4 *>(wids + i));
#else
			*reinterpret_cast<uint4 *>(wids + i) =
				make_uint4(id[0] | (id[1] << 16), id[2] | (id[3] << 16), id[4] | (id[5] << 16),
					   id[6] | (id[7] << 16));
#endif
		} else {
#pragma unroll
			for (uint32_t k = 0; k < 7; ++k)
				if (k < cntp)
					wids[i + k] = (uint16_t)id[k];
		}
	});
	if (nb > A.dense_cap) {
		/* (the spill-over table starts out like the table in LDS: every slot holds position 0's entry, the
		 * reference's zeroed table seen through the check bit, parse_lean) */
		uint32_t first4;
		__builtin_memcpy(&first4, src, 4);
		const uint32_t zv = (((first4 * kHashMul) >> (shift - 1)) & 1u) * 0x80008000u;
		uint4 *z = reinterpret_cast<uint4 *>(F.region + A.spill_off);
		for (uint32_t k = lane; k < (A.spill_cap * 2 + 15) / 16; k += 64)
			z[k] = make_uint4(zv, zv, zv, zv);
	}
	wave_lds_fence();
	/* the ids (and the zeroed spill table) are read back by this wave only (same CU, same L1/L2
	 * path, program order); make the stores leave the wave before the first load of them */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	return nb;
}

/* ==========================================================================================
 * The dense parser's step loop in ISA (round 6).  parse_lean's C++ below states the logic -- and
 * still runs the steps this block leaves to it: sparse steps, the last steps of a fragment (a lane
 * beyond the scan limit), fragments with a spill-over table, the other table placements, the
 * ORD = false parsers.  This block is the same logic for the common case -- table in LDS indexed by
 * dense ids, all 64 lanes in front of the scan limit -- written by hand as ONE loop:
 *
 *   12: FRONT   table read + returning add, check bit, candidate gather with the previous step's
 *               record store behind it, flagged lanes, 16-byte comparison, stop / special masks,
 *               next-stop table, the walk's plain hops
 *   13: VISIT   the walk stands at a special lane (a flagged one, or a match of all 16 bytes that
 *               may be longer): settled as parse_lean's visits() does, the walk goes on
 *   14:         no copy in the whole step: the scan goes on behind its last probe
 *   10: BACK    cursor, the next step's loads, records (running maximum of the copies' ends by DPP),
 *               15: commit; leaves at 19 when the next step is not one for this loop
 *
 * 239 instructions on the common path of round 5's compiled step, ~145 here.  The compiler's version
 * spends the difference on boolean round trips (v_cndmask 0/1 + v_cmp for every ballot of a combined
 * predicate), lane-mask tests in vector registers where v_cndmask takes the mask as it is, v_mbcnt,
 * selects where the dummy table entry needs none, copies at the loop's head, back edge and around
 * every visit, and s_nops where independent instructions fit.
 *
 * Wait states are spelled out (gfx940 family; the compiler's hazard recogniser does not look into
 * inline asm): vector-written SGPR -> vector read: 2, -> v_readlane / v_writelane lane select: 4;
 * vector-written VGPR -> DPP read: 2, -> v_readlane: 1.  A scalar instruction may read a
 * vector-written SGPR at once.
 *
 * Registers are fixed (named as clobbers, or bound to register variables where C++ hands values in
 * and out).  v59 is the highest, for a reason that is not understood: a build of the global-table kernel
 * that declared exactly 64 VGPRs with this block using v60-v63 was bit-exact with one workgroup per CU
 * and computed wrong match lengths beside others (the extension's shift amounts, which lived in v62 /
 * v63: every fragment behind the first four per CU, every run); with the map ended at v59 (60 declared) or
 * with more than 64 declared it is exact, and so has every build since.  It reproduces on demand (the map
 * moved up by four again: blocks 512.. of 64 MiB of G_low wrong, every run), it is not the register count
 * (the same build with v60-v63 and v36-v39 exchanged -- the match lengths, ends, candidates and next stops up
 * there, the temporaries down here -- is exact at 64 VGPRs), and a stand-alone kernel that computes in,
 * loads into and reads back v60-v63 of a 64-VGPR allocation at 32 waves per CU finds nothing
 * (tools/ubench/vgpr_top.hip, a billion read-backs).  tests/test_isa_hazards.py pins both precautions.
 *   v32 mlen   v33 cl   v34 cand   v35 nx   v36 entry address   v37 its dword   v38 my 1   v39 my entry
 *   v40 pos    v41 probes left per lane (32; lane 0: what is left of the scan)   v42 id / slot   v43 record offset
 *   v44-v47 own 16 bytes   v48-v49, v52-v59 scratch   v50-v51 the step's record
 *   s[60:61] lanes with a bucket   s[62:63] flagged   s[64:65] stops   s[66:67] special   s[68:69] taken
 *   s[70:71] candidate can match / inside a copy   s72-s79, s84-s91 scratch   s80 t   s81 lim0   s82 go
 *   s83 the compiler's m0   s92-s93 the global table's epoch field
 * ======================================================================================== */
#ifndef CSNAPPY_NT
#define CSNAPPY_NT 0 /* experiments: streaming hints (1: the prologue's id stores, 2: the records' stores, 4: the id loads) */
#endif
#if CSNAPPY_NT & 2
#define CSNAPPY_NT_REC " nt"
#else
#define CSNAPPY_NT_REC ""
#endif
#if CSNAPPY_NT & 4
#define CSNAPPY_NT_IDL " nt"
#else
#define CSNAPPY_NT_IDL ""
#endif
#ifndef CSNAPPY_ISA_PROF
#define CSNAPPY_ISA_PROF 0
#endif
#if CSNAPPY_ISA_PROF
/* development builds only (tools/build_variant.sh <name> -DCSNAPPY_ISA_PROF=1, tools/phase_isa.py): the dense loop's phases
 * by s_memtime, summed per wave in v65.. (every stamp costs a scalar-memory round trip of its own: the same for every phase) */
__device__ unsigned long long g_isa_prof[16];
#define CSNAPPY_ISA_P(r) "s_memtime s[96:97]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s98, s96, s99\n\ts_mov_b32 s99, s96\n\tv_add_u32_e32 " r ", s98, " r "\n\t"
#define CSNAPPY_ISA_P0 "s_memtime s[96:97]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s99, s96\n\t"
#define CSNAPPY_ISA_PSTEP "v_add_u32_e32 v71, 1, v71\n\t"
#else
#define CSNAPPY_ISA_PSTEP ""
#define CSNAPPY_ISA_P(r) ""
#define CSNAPPY_ISA_P0 ""
#endif
#if CSNAPPY_TIMING_TA == 8 /* timing experiment: one more 2-byte-per-lane load per step (the ids again) */
#define CSNAPPY_TIMING_EXTRA_SMALL_LOAD "global_load_ushort v57, v48, %[ids]\n\t"
#else
#define CSNAPPY_TIMING_EXTRA_SMALL_LOAD ""
#endif
#if CSNAPPY_OWN_DW4 == 1 /* experiment: the step's own 16 bytes as four dword loads */
#define CSNAPPY_ISA_OWN16 "global_load_dword v44, v49, %[src]\n\tglobal_load_dword v45, v49, %[src] offset:4\n\tglobal_load_dword v46, v49, %[src] offset:8\n\tglobal_load_dword v47, v49, %[src] offset:12\n\t"
#elif CSNAPPY_OWN_DW4 == 2 /* ... as two 8-byte loads */
#define CSNAPPY_ISA_OWN16 "global_load_dwordx2 v[44:45], v49, %[src]\n\tglobal_load_dwordx2 v[46:47], v49, %[src] offset:8\n\t"
#else
#define CSNAPPY_ISA_OWN16 "global_load_dwordx4 v[44:47], v49, %[src]\n\t"
#endif
#ifndef CSNAPPY_DENSE_WINDOW
#define CSNAPPY_DENSE_WINDOW 0 /* 1: the dense loop takes its own bytes and ids out of a window requested a step ahead (A/B: no gain) */
#endif
#if CSNAPPY_TIMING_TA == 1 /* timing experiment: one more 16-byte-per-lane load per step (is the texture addresser the bound?) */
#define CSNAPPY_TIMING_EXTRA_LOAD "global_load_dwordx4 v[44:47], v49, %[src]\n\t"
#elif CSNAPPY_TIMING_TA == 2 /* ... or eight more vector instructions */
#define CSNAPPY_TIMING_EXTRA_LOAD "v_add_u32_e32 v56, 1, v56\n\tv_add_u32_e32 v56, 1, v56\n\tv_add_u32_e32 v56, 1, v56\n\tv_add_u32_e32 v56, 1, v56\n\tv_add_u32_e32 v56, 1, v56\n\tv_add_u32_e32 v56, 1, v56\n\tv_add_u32_e32 v56, 1, v56\n\tv_add_u32_e32 v56, 1, v56\n\t"
#else
#define CSNAPPY_TIMING_EXTRA_LOAD ""
#endif
/* The step's own bytes.  Sixteen bytes per lane at byte stride 1 are 1 KiB of requests for 79 bytes, and
 * the texture addresser is what this kernel saturates (one more such load per step: +13 % time; eight more
 * vector instructions: +0.7 %): the window is loaded ONCE as aligned dwords, lane l dword l, one coalesced
 * request, and every lane picks its five dwords out of the others' registers (ds_bpermute: the LDS crossbar,
 * no memory) and shifts its 16 bytes into place.
 * Measured: it pays around the global table (G_low 4.10 -> 3.99 ms per GiB of compress), and costs the dense
 * table 2 % (text 8.12 -> 8.28, pages 6.16 -> 6.29): there the step's latency counts for more than the
 * addresser's time, and the five permutes sit in front of everything the step does.  So the dense loop keeps
 * its 16 bytes per lane, the global-table loop takes the window. */
#define CSNAPPY_ISA_OWN_BYTES \
	"s_and_b32 s79, %[p0], 3\n\t"                                                                                      \
	"v_add_u32_e32 v59, s79, %[lane]\n\t"              /* where my 16 bytes begin in the window of dwords lane l holds dword l of */\
	"v_and_b32_e32 v48, -4, v59\n\t"                                                                                   \
	"ds_bpermute_b32 v52, v48, v44\n\t"                                                                                \
	"ds_bpermute_b32 v53, v48, v44 offset:4\n\t"                                                                       \
	"ds_bpermute_b32 v56, v48, v44 offset:8\n\t"                                                                       \
	"ds_bpermute_b32 v57, v48, v44 offset:12\n\t"                                                                      \
	"ds_bpermute_b32 v58, v48, v44 offset:16\n\t"                                                                      \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_alignbyte_b32 v44, v53, v52, v59\n\t"           /* my own 16 bytes */                                           \
	"v_alignbyte_b32 v45, v56, v53, v59\n\t"                                                                           \
	"v_alignbyte_b32 v46, v57, v56, v59\n\t"                                                                           \
	"v_alignbyte_b32 v47, v58, v57, v59\n\t"

#ifndef CSNAPPY_HOPS_UNROLL
#define CSNAPPY_HOPS_UNROLL 1
#endif
#define CSNAPPY_ISA_HOP(BR) \
	"s_bitset1_b64 s[68:69], s80\n\t"                                                                                  \
	"v_readlane_b32 s80, v35, s80\n\t"                                                                                 \
	"s_nop 0\n\t"                                                                                                      \
	"s_cmp_lt_u32 s80, 64\n\t"                                                                                         \
	BR
#if CSNAPPY_HOPS_UNROLL
/* (four hops a round: the branch of a hop that goes on falls through) */
#define CSNAPPY_ISA_HOPS /* t = nx[t] until t >= 64, every lane passed marked in `taken` */                            \
	"1:\n\t"                                                                                                           \
	CSNAPPY_ISA_HOP("s_cbranch_scc0 2f\n\t")                                                                           \
	CSNAPPY_ISA_HOP("s_cbranch_scc0 2f\n\t")                                                                           \
	CSNAPPY_ISA_HOP("s_cbranch_scc0 2f\n\t")                                                                           \
	CSNAPPY_ISA_HOP("s_cbranch_scc1 1b\n\t")                                                                           \
	"2:\n\t"
#else
#define CSNAPPY_ISA_HOPS /* t = nx[t] until t >= 64, every lane passed marked in `taken` */                            \
	"1:\n\t"                                                                                                           \
	CSNAPPY_ISA_HOP("s_cbranch_scc1 1b\n\t")
#endif

#define CSNAPPY_ISA_PREFIX16(d, a, b, c) /* d = equal leading bytes (0..16) of the strings whose XOR is d, a, b, c */  \
	"v_ffbl_b32_e32 " d ", " d "\n\t"                                                                                  \
	"v_ffbl_b32_e32 " a ", " a "\n\t"                                                                                  \
	"v_ffbl_b32_e32 " b ", " b "\n\t"                                                                                  \
	"v_ffbl_b32_e32 " c ", " c "\n\t"                                                                                  \
	"v_add_u32_e64 " a ", " a ", 32 clamp\n\t"                                                                         \
	"v_add_u32_e64 " c ", " c ", 32 clamp\n\t"                                                                         \
	"v_min3_u32 " d ", " d ", " a ", 64\n\t"          /* equal low bits of bytes 0..7 (64: all) */                     \
	"v_min3_u32 " b ", " b ", " c ", 64\n\t"          /* ... of bytes 8..15 */                                         \
	"v_lshrrev_b32_e32 " a ", 6, " d "\n\t"                                                                            \
	"v_mad_u32_u24 " d ", " a ", " b ", " d "\n\t"    /* + the high half when the low one is all equal */              \
	"v_lshrrev_b32_e32 " d ", 3, " d "\n\t"

/* the pieces that differ between the table placements: the FRONT's table access (label 12), the test for
 * another round, the next step's loads and the commit (behind label 11) */
#if CSNAPPY_TIMING_TA == 7 /* timing experiment: one more LDS read per step, nobody waits for it */
#define CSNAPPY_TIMING_EXTRA_LDS "ds_read_b32 v60, v37\n\t"
#else
#define CSNAPPY_TIMING_EXTRA_LDS ""
#endif
#if CSNAPPY_TIMING_TA == 5 /* timing experiment (wrong output where compiled steps mix in): no check bit, every lane with a bucket gathers */
#define CSNAPPY_ISA_DENSE_CHECK_BIT "v_mov_b32_e32 v59, 0\n\t"
#elif CSNAPPY_TIMING_TA == 6 /* ... the check bit kept, its multiply paid twice */
#define CSNAPPY_ISA_DENSE_CHECK_BIT "v_mul_lo_u32 v59, v44, %[mul]\n\tv_mul_lo_u32 v59, v44, %[mul]\n\tv_bfe_u32 v59, v59, %[shm1], 1\n\t"
#else
#define CSNAPPY_ISA_DENSE_CHECK_BIT \
	"v_mul_lo_u32 v59, v44, %[mul]\n\t"                                                                                \
	"v_bfe_u32 v59, v59, %[shm1], 1\n\t"               /* check bit: one more bit of my hash */
#endif
#define CSNAPPY_ISA_TABLE_DENSE \
	"s_sub_u32 s81, 33, %[q1]\n\t"                     /* probes the scan in progress has left */                      \
	CSNAPPY_ISA_DENSE_OWN_WAIT                         /* own bytes and id */                                          \
	"v_lshlrev_b32_e32 v36, 1, v42\n\t"                /* my table entry (id 0: the dummy) */                          \
	"ds_read_u16 v56, v36\n\t"                                                                                         \
	"v_and_b32_e32 v37, 0xfffc, v36\n\t"               /* its dword */                                                 \
	"v_lshlrev_b32_e32 v58, 4, v42\n\t"                /* bits 4:0 = 16 * (id & 1) */                                  \
	"v_lshlrev_b32_e64 v38, v58, 1\n\t"                /* 1 in my half */                                              \
	"ds_add_rtn_u32 v57, v37, v38\n\t"                 /* comes back with the lower lanes' ones in it */               \
	"v_writelane_b32 v41, s81, 0\n\t"                  /* lane 0 searches what is left of the scan, the others 33 probes */ \
	CSNAPPY_ISA_DENSE_CHECK_BIT                                                                                        \
	"v_cmp_ne_u32_e64 s[60:61], 0, v42\n\t"            /* lanes with a bucket */                                       \
	"v_lshl_or_b32 v39, v59, 15, v40\n\t"              /* my entry, if I am inserted */                                \
	"s_waitcnt lgkmcnt(1)\n\t"                         /* the entry (the add may be on its way) */                     \
	"v_xor_b32_e32 v59, v56, v39\n\t"                                                                                  \
	"v_and_b32_e32 v34, 0x7fff, v56\n\t"                                                                               \
	"v_cmp_lt_u32_e32 vcc, v59, %[thr]\n\t"            /* check bits agree (thr: 0x8000; lane 0, insert-only: 0) */    \
	"s_and_b64 s[70:71], vcc, s[60:61]\n\t"            /* the candidate can match at all */                            \
	"v_cndmask_b32_e64 v59, 0, v34, s[70:71]\n\t"                                                                      \
	"global_load_dwordx4 v[52:55], v59, %[src]\n\t"    /* candidate gather (no candidate: position 0) */               \
	"global_store_dwordx2 v43, v[50:51], %[R]" CSNAPPY_NT_REC "\n\t" /* the previous step's records, behind the gather */ \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	CSNAPPY_ISA_P("v66")                                                                                               \
	CSNAPPY_TIMING_EXTRA_LDS                                                                                           \
	"v_bfe_u32 v57, v57, v58, 16\n\t"                  /* my half as the add found it */                               \
	"v_cmp_ne_u32_e32 vcc, v57, v56\n\t"               /* not the entry: a lower lane has my slot */                   \
	"s_and_b64 s[62:63], vcc, s[60:61]\n\t"            /* flagged lanes */                                             \
	"s_waitcnt vmcnt(1)\n\t"                           /* the gather (the store may be on its way) */                  \
	CSNAPPY_ISA_P("v67")

#define CSNAPPY_ISA_GO_DENSE \
	"s_cmp_lt_u32 %[p0], %[limit64]\n\t"                                                                               \
	"s_cselect_b32 s82, %[q1], 99\n\t"                 /* next step in this loop too: < 33 */

#if CSNAPPY_TIMING_TA == 3 /* timing experiment: what one more dependent round trip in front of the step's loads costs */
#define CSNAPPY_TIMING_ROUND_TRIP "v_lshlrev_b32_e32 v48, 1, v49\n\tglobal_load_ushort v56, v48, %[ids]\n\ts_waitcnt vmcnt(0)\n\t"
#else
#define CSNAPPY_TIMING_ROUND_TRIP ""
#endif
#if CSNAPPY_DENSE_WINDOW
/* The dense loop's own bytes and ids (round 6, second half).  Until here the next step's 16 bytes per lane and its id
 * were requested when the cursor was known and waited for at the top of the step: a memory round trip on the step's
 * chain, in front of the table read and the candidate gather (a dependent load more in that place costs 20 % of the
 * parser's time: CSNAPPY_TIMING_TA=3).  Now every step requests a WINDOW anchored at its own cursor a -- lane l the
 * aligned dword 4l of the input from a & ~3 (256 bytes) and the dword of ids 2l, 2l + 1 from a & ~1 (128 positions) --
 * which nobody needs before the NEXT cursor is known, a whole step later: the next step starts d <= 63 positions
 * further (its last copy ended inside the step or one behind it), its 79 bytes and 64 ids lie inside the window, and
 * every lane picks them out of the others' registers (ds_bpermute: the LDS crossbar, ~a tenth of the round trip,
 * under the records' DPP chain).  d > 63 (a copy ran on past the step: one step in seven on text): the window just
 * requested for the new cursor is waited for, as before.
 *   v60 / v61  the window being loaded (bytes / ids), anchor s92     v62 / v63  the one in use
 *   s93 / s94  the highest dword the windows may touch (bytes / ids)
 *   v52, v54-v57, v49 the permutes' results, v35 / v33 the byte shift / the id's half: untouched up to label 15 */
#define CSNAPPY_ISA_WINDOW_REQUEST(P) /* the windows anchored at cursor P -> v60, v61; s92 = P */                         \
	"s_and_b32 s84, " P ", -4\n\t"                                                                                     \
	"s_and_b32 s85, " P ", -2\n\t"                                                                                     \
	"v_lshl_add_u32 v48, %[lane], 2, s84\n\t"                                                                          \
	"s_lshl_b32 s85, s85, 1\n\t"                                                                                       \
	"v_min_u32_e32 v48, s93, v48\n\t"                  /* (clamped: a window may reach past the fragment's end) */     \
	"s_mov_b32 s92, " P "\n\t"                                                                                         \
	"global_load_dword v60, v48, %[src]\n\t"                                                                           \
	"v_lshl_add_u32 v48, %[lane], 2, s85\n\t"                                                                          \
	"v_min_u32_e32 v48, s94, v48\n\t"                                                                                  \
	"global_load_dword v61, v48, %[ids]" CSNAPPY_NT_IDL "\n\t"

#define CSNAPPY_ISA_ENTRY_DENSE \
	"s_sub_u32 s93, %[n], 4\n\t"                                                                                       \
	"s_sub_u32 s94, %[n], 2\n\t"                                                                                       \
	"s_and_b32 s93, s93, -4\n\t"                       /* the last whole dword of the input */                         \
	"s_and_b32 s94, s94, -2\n\t"                                                                                       \
	"s_lshl_b32 s94, s94, 1\n\t"                       /* ... and of the ids, as a byte offset */                      \
	CSNAPPY_ISA_WINDOW_REQUEST("%[p0]")

#define CSNAPPY_ISA_LOADS_DENSE \
	"s_sub_u32 s88, %[p0], s92\n\t"                    /* d: how far the cursor moved */                               \
	"s_and_b32 s87, s92, 3\n\t"                                                                                        \
	"s_and_b32 s86, s92, 1\n\t"                                                                                        \
	"s_cmp_gt_u32 s88, 63\n\t"                                                                                         \
	"s_cbranch_scc0 2f\n\t"                                                                                            \
	CSNAPPY_ISA_WINDOW_REQUEST("%[p0]")                /* beyond the old window: the new one, now */                   \
	"s_and_b32 s87, %[p0], 3\n\t"                                                                                      \
	"s_and_b32 s86, %[p0], 1\n\t"                                                                                      \
	"s_mov_b32 s88, 0\n\t"                                                                                             \
	"s_waitcnt vmcnt(0)\n\t"                                                                                           \
	"2:\n\t"                                                                                                           \
	"s_add_u32 s87, s87, s88\n\t"                      /* the cursor's byte in the window of bytes */                  \
	"s_add_u32 s86, s86, s88\n\t"                      /* ... its id in the window of ids */                           \
	"v_add_u32_e32 v35, s87, %[lane]\n\t"              /* where my 16 bytes begin in the window */                     \
	"v_add_u32_e32 v33, s86, %[lane]\n\t"              /* ... and my id */                                             \
	"s_waitcnt vmcnt(1)\n\t"                           /* the window requested a step ago (the records' store, issued  \
	                                                     * behind it, may be on its way) */                            \
	"v_mov_b32_e32 v62, v60\n\t"                                                                                       \
	"v_mov_b32_e32 v63, v61\n\t"                                                                                       \
	"v_and_b32_e32 v34, -4, v35\n\t"                                                                                   \
	"v_lshlrev_b32_e32 v49, 1, v33\n\t"                                                                                \
	"ds_bpermute_b32 v52, v34, v62\n\t"                                                                                \
	"v_and_b32_e32 v49, -4, v49\n\t"                   /* the lane that holds the dword of ids mine is in */           \
	"ds_bpermute_b32 v54, v34, v62 offset:4\n\t"                                                                       \
	"ds_bpermute_b32 v49, v49, v63\n\t"                                                                                \
	"ds_bpermute_b32 v55, v34, v62 offset:8\n\t"                                                                       \
	"ds_bpermute_b32 v56, v34, v62 offset:12\n\t"                                                                      \
	"ds_bpermute_b32 v57, v34, v62 offset:16\n\t"                                                                      \
	"v_lshlrev_b32_e32 v33, 4, v33\n\t"                /* bits 4:0: 16 * (my id is the dword's high half) */           \
	"v_add_u32_e32 v40, %[p0], %[lane]\n\t"            /* the next step's positions */                                 \
	"s_cmp_eq_u32 s92, %[p0]\n\t"                      /* (requested above) */                                         \
	"s_cbranch_scc1 3f\n\t"                                                                                            \
	CSNAPPY_ISA_WINDOW_REQUEST("%[p0]")                /* the window of the step behind the next */                    \
	"3:\n\t"

#define CSNAPPY_ISA_DENSE_OWN_WAIT ""
#define CSNAPPY_ISA_DENSE_OWN_FINISH /* label 15, in front of the commit: the permutes are back */                      \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_alignbyte_b32 v44, v54, v52, v35\n\t"           /* my own 16 bytes */                                           \
	"v_alignbyte_b32 v45, v55, v54, v35\n\t"                                                                           \
	"v_alignbyte_b32 v46, v56, v55, v35\n\t"                                                                           \
	"v_alignbyte_b32 v47, v57, v56, v35\n\t"                                                                           \
	"v_bfe_u32 v42, v49, v33, 16\n\t"                  /* my id */

#else
#define CSNAPPY_ISA_ENTRY_DENSE ""
#define CSNAPPY_ISA_DENSE_OWN_WAIT "s_waitcnt vmcnt(0)\n\t"
#define CSNAPPY_ISA_DENSE_OWN_FINISH ""
#define CSNAPPY_ISA_LOADS_DENSE \
	"v_add_u32_e32 v40, %[p0], %[lane]\n\t"            /* the next step's positions */                                 \
	"v_min_u32_e32 v49, %[safemax], v40\n\t"           /* (clamped: harmless loads when the loop ends here) */         \
	CSNAPPY_TIMING_ROUND_TRIP                                                                      \
	CSNAPPY_ISA_OWN16                                                                              \
	CSNAPPY_TIMING_EXTRA_LOAD                                                                      \
	"v_lshlrev_b32_e32 v48, 1, v49\n\t"                                                                                \
	"global_load_ushort v42, v48, %[ids]" CSNAPPY_NT_IDL "\n\t"                                                        \
	CSNAPPY_TIMING_EXTRA_SMALL_LOAD

#endif
#define CSNAPPY_ISA_COMMIT_DENSE CSNAPPY_ISA_DENSE_OWN_FINISH CSNAPPY_ISA_COMMIT_LDS
#define CSNAPPY_ISA_COMMIT_LDS \
	"v_cmp_ge_u32_e32 vcc, s72, %[lane]\n\t"           /* lanes up to e_final */                                       \
	"ds_sub_u32 v37, v38\n\t"                          /* the adds are taken back */                                   \
	"s_andn2_b64 s[70:71], vcc, s[70:71]\n\t"          /* inserted lanes */                                            \
	"s_mov_b64 exec, s[70:71]\n\t"                                                                                     \
	"ds_write_b16 v36, v39\n\t"                        /* (of several with one slot the highest stays) */              \
	"s_mov_b64 exec, -1\n\t"

/* the table in LDS indexed by the hash itself (parse_lean<TAB_LDS_HASH>: tables of <= 8 KiB, no prologue): every lane
 * has a slot, s[60:61] = all lanes; the test for another round and the commit are the dense table's */
#define CSNAPPY_ISA_TABLE_HASH \
	"s_sub_u32 s81, 33, %[q1]\n\t"                     /* probes the scan in progress has left */                      \
	"s_waitcnt vmcnt(0)\n\t"                           /* own bytes */                                                 \
	"v_mul_lo_u32 v59, v44, %[mul]\n\t"                                                                                \
	"v_lshrrev_b32_e32 v42, %[shift], v59\n\t"         /* my slot: the hash (every lane has one) */                    \
	"v_lshlrev_b32_e32 v36, 1, v42\n\t"                /* my table entry */                                            \
	"ds_read_u16 v56, v36\n\t"                                                                                         \
	"v_and_b32_e32 v37, 0xfffc, v36\n\t"               /* its dword */                                                 \
	"v_lshlrev_b32_e32 v58, 4, v42\n\t"                /* bits 4:0 = 16 * (slot & 1) */                                \
	"v_lshlrev_b32_e64 v38, v58, 1\n\t"                /* 1 in my half */                                              \
	"ds_add_rtn_u32 v57, v37, v38\n\t"                 /* comes back with the lower lanes' ones in it */               \
	"v_writelane_b32 v41, s81, 0\n\t"                  /* lane 0 searches what is left of the scan, the others 33 probes */\
	"v_bfe_u32 v59, v59, %[shm1], 1\n\t"               /* check bit: one more bit of my hash */                        \
	"v_lshl_or_b32 v39, v59, 15, v40\n\t"              /* my entry, if I am inserted */                                \
	"s_waitcnt lgkmcnt(1)\n\t"                         /* the entry (the add may be on its way) */                     \
	"v_xor_b32_e32 v59, v56, v39\n\t"                                                                                  \
	"v_and_b32_e32 v34, 0x7fff, v56\n\t"                                                                               \
	"v_cmp_lt_u32_e32 vcc, v59, %[thr]\n\t"            /* check bits agree (thr: 0x8000; lane 0, insert-only: 0) */    \
	"s_mov_b64 s[70:71], vcc\n\t"                      /* the candidate can match at all */                            \
	"v_cndmask_b32_e64 v59, 0, v34, s[70:71]\n\t"                                                                      \
	"global_load_dwordx4 v[52:55], v59, %[src]\n\t"    /* candidate gather (no candidate: position 0) */               \
	"global_store_dwordx2 v43, v[50:51], %[R]\n\t"     /* the previous step's records, behind the gather */            \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_bfe_u32 v57, v57, v58, 16\n\t"                  /* my half as the add found it */                               \
	"v_cmp_ne_u32_e64 s[62:63], v57, v56\n\t"          /* flagged: not the entry, a lower lane has my slot */          \
	"s_waitcnt vmcnt(1)\n\t"                           /* the gather (the store may be on its way) */

#define CSNAPPY_ISA_LOADS_HASH \
	"v_add_u32_e32 v40, %[p0], %[lane]\n\t"            /* the next step's positions */                                 \
	"v_min_u32_e32 v49, %[safemax], v40\n\t"           /* (clamped: harmless loads when the loop ends here) */         \
	"global_load_dwordx4 v[44:47], v49, %[src]\n\t"

/* the dense table with a spill-over in HBM (parse_lean<TAB_LDS_DENSE, SPILL = true>): the buckets beyond the LDS
 * table (ids >= dcap) keep their entries in a 4 KiB table behind the fragment's ids; s[94:95] = the lanes of such
 * buckets, v35 (behind the walk) = their entries' offsets.  (Leaving the spill-over's stores in flight across the
 * next step's first wait -- vmcnt(1) when one was issued -- was measured: 9.46 against 9.47 ms per GiB of compress on
 * urls.10K, not kept.) */
#define CSNAPPY_ISA_TABLE_SPILL \
	"s_sub_u32 s81, 33, %[q1]\n\t"                     /* probes the scan in progress has left */                      \
	"s_add_u32 s92, %[epoch], -1\n\t"                  /* (one epoch of the spilled lanes' filter per step) */         \
	"s_lshl_b32 s93, %[epoch], 20\n\t"                                                                                 \
	"s_waitcnt vmcnt(0)\n\t"                           /* own bytes and id, and the spill-over's stores of the step before */\
	"v_cmp_le_u32_e64 s[94:95], %[dcap], v42\n\t"      /* lanes whose bucket lives in the spill-over table in HBM */   \
	"v_cmp_ne_u32_e64 s[60:61], 0, v42\n\t"            /* lanes with a bucket */                                       \
	"v_mul_lo_u32 v59, v44, %[mul]\n\t"                                                                                \
	"v_cndmask_b32_e64 v57, v42, 0, s[94:95]\n\t"      /* their LDS accesses go to the dummy entry */                  \
	"v_lshlrev_b32_e32 v36, 1, v57\n\t"                /* my table entry */                                            \
	"ds_read_u16 v56, v36\n\t"                                                                                         \
	"v_and_b32_e32 v37, 0xfffc, v36\n\t"               /* its dword */                                                 \
	"v_lshlrev_b32_e32 v58, 4, v57\n\t"                /* bits 4:0 = 16 * (id & 1) */                                  \
	"v_lshlrev_b32_e64 v38, v58, 1\n\t"                /* 1 in my half */                                              \
	"ds_add_rtn_u32 v57, v37, v38\n\t"                 /* comes back with the lower lanes' ones in it */               \
	"v_writelane_b32 v41, s81, 0\n\t"                  /* lane 0 searches what is left of the scan, the others 33 probes */\
	"v_bfe_u32 v59, v59, %[shm1], 1\n\t"               /* check bit: one more bit of my hash */                        \
	"v_lshl_or_b32 v39, v59, 15, v40\n\t"              /* my entry, if I am inserted */                                \
	"s_cmp_lg_u64 s[94:95], 0\n\t"                                                                                     \
	"s_cbranch_scc0 50f\n\t"                                                                                           \
	/* ---- some lanes' buckets live in HBM: their entries, and one filter among themselves (atomic minimum of
	 * {epoch, id, lane} tags per key: the smallest id of a key settles it, lowest lane first) ---- */\
	"v_subrev_u32_e32 v48, %[dcap], v42\n\t"                                                                           \
	"v_lshl_or_b32 v53, v42, 7, %[lane]\n\t"                                                                           \
	"v_and_b32_e32 v49, 127, v48\n\t"                  /* key */                                                       \
	"v_or_b32_e32 v53, s93, v53\n\t"                   /* tag */                                                       \
	"v_lshl_add_u32 v49, v49, 2, %[sbase]\n\t"                                                                         \
	"v_lshlrev_b32_e32 v48, 1, v48\n\t"                                                                                \
	"s_mov_b64 exec, s[94:95]\n\t"                                                                                     \
	"ds_min_u32 v49, v53\n\t"                                                                                          \
	"global_load_ushort v52, v48, %[spill]\n\t"        /* the entry */                                                 \
	"s_mov_b64 exec, -1\n\t"                                                                                           \
	"s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                                \
	"v_cndmask_b32_e64 v56, v56, v52, s[94:95]\n\t"                                                                    \
	"ds_read_b32 v49, v49\n\t"                         /* what settled my key */                                       \
	"50:\n\t"                                                                                                          \
	"s_waitcnt lgkmcnt(1)\n\t"                         /* the entry (the add, or the filter's read-back, may be on its way) */\
	"v_xor_b32_e32 v59, v56, v39\n\t"                                                                                  \
	"v_and_b32_e32 v34, 0x7fff, v56\n\t"                                                                               \
	"v_cmp_lt_u32_e32 vcc, v59, %[thr]\n\t"            /* check bits agree (thr: 0x8000; lane 0, insert-only: 0) */    \
	"s_and_b64 s[70:71], vcc, s[60:61]\n\t"            /* the candidate can match at all */                            \
	"v_cndmask_b32_e64 v59, 0, v34, s[70:71]\n\t"                                                                      \
	"global_load_dwordx4 v[52:55], v59, %[src]\n\t"    /* candidate gather (no candidate: position 0) */               \
	"global_store_dwordx2 v43, v[50:51], %[R]\n\t"     /* the previous step's records, behind the gather */            \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_bfe_u32 v57, v57, v58, 16\n\t"                  /* my half as the add found it */                               \
	"v_cmp_ne_u32_e32 vcc, v57, v56\n\t"               /* not the entry: a lower lane has my slot */                   \
	"s_and_b64 s[62:63], vcc, s[60:61]\n\t"            /* flagged lanes (of the table in LDS) */                       \
	"s_cmp_lg_u64 s[94:95], 0\n\t"                                                                                     \
	"s_cbranch_scc0 51f\n\t"                                                                                           \
	"v_bfe_u32 v57, v49, 7, 13\n\t"                    /* the id that settled my key */                                \
	"v_and_b32_e32 v58, 127, v49\n\t"                  /* ... and its lowest lane */                                   \
	"v_cmp_ne_u32_e64 s[88:89], v57, v42\n\t"          /* another id: flagged to be safe */                            \
	"v_cmp_lt_u32_e32 vcc, v58, %[lane]\n\t"           /* mine, and a lower lane has it */                             \
	"s_or_b64 vcc, vcc, s[88:89]\n\t"                                                                                  \
	"s_and_b64 vcc, vcc, s[94:95]\n\t"                                                                                 \
	"s_andn2_b64 vcc, vcc, 1\n\t"                      /* (lane 0 has no lower lane) */                                \
	"s_andn2_b64 s[62:63], s[62:63], s[94:95]\n\t"                                                                     \
	"s_or_b64 s[62:63], s[62:63], vcc\n\t"                                                                             \
	"51:\n\t"                                                                                                          \
	"s_waitcnt vmcnt(1)\n\t"                           /* the gather (the store may be on its way) */

#define CSNAPPY_ISA_GO_SPILL \
	"s_cmp_lt_u32 %[p0], %[limit64]\n\t"                                                                               \
	"s_cselect_b32 s82, %[q1], 99\n\t"                 /* next step in this loop too: < 33 */                          \
	"s_cmp_lt_u32 s92, 2\n\t"                          /* ... unless the filter's epoch field runs out: the C++ step starts it over */\
	"s_cselect_b32 s82, 99, s82\n\t"

#define CSNAPPY_ISA_LOADS_SPILL \
	"v_subrev_u32_e32 v35, %[dcap], v42\n\t"           /* (my entry's offset in the spill-over table, for the commit: the id is about to be overwritten) */\
	"v_lshlrev_b32_e32 v35, 1, v35\n\t"                                                                                \
	"v_add_u32_e32 v40, %[p0], %[lane]\n\t"            /* the next step's positions */                                 \
	"v_min_u32_e32 v49, %[safemax], v40\n\t"           /* (clamped: harmless loads when the loop ends here) */         \
	"global_load_dwordx4 v[44:47], v49, %[src]\n\t"                                                                    \
	"v_lshlrev_b32_e32 v48, 1, v49\n\t"                                                                                \
	"global_load_ushort v42, v48, %[ids]\n\t"

#define CSNAPPY_ISA_COMMIT_SPILL \
	"v_cmp_ge_u32_e32 vcc, s72, %[lane]\n\t"           /* lanes up to e_final */                                       \
	"ds_sub_u32 v37, v38\n\t"                          /* the adds are taken back */                                   \
	"s_andn2_b64 s[70:71], vcc, s[70:71]\n\t"          /* inserted lanes */                                            \
	"s_mov_b64 exec, s[70:71]\n\t"                                                                                     \
	"ds_write_b16 v36, v39\n\t"                        /* (of several with one slot the highest stays; lanes of the spill-over write the dummy) */\
	"s_mov_b64 exec, -1\n\t"                                                                                           \
	"s_and_b64 s[84:85], s[70:71], s[94:95]\n\t"       /* inserted lanes of the spill-over table */                    \
	"s_cmp_eq_u64 s[84:85], 0\n\t"                                                                                     \
	"s_cbranch_scc1 43f\n\t"                                                                                           \
	"s_and_b64 s[86:87], s[62:63], s[84:85]\n\t"       /* of several with one bucket only the highest may store (memory keeps no order) */\
	"s_cmp_eq_u64 s[86:87], 0\n\t"                                                                                     \
	"s_cbranch_scc1 42f\n\t"                                                                                           \
	"s_mov_b64 s[88:89], 0\n\t"                                                                                        \
	"40:\n\t"                                                                                                          \
	"s_flbit_i32_b64 s76, s[86:87]\n\t"                                                                                \
	"s_xor_b32 s76, s76, 63\n\t"                                                                                       \
	"v_readlane_b32 s77, v35, s76\n\t"                                                                                 \
	"s_bfm_b64 s[90:91], s76, 0\n\t"                                                                                   \
	"s_nop 0\n\t"                                                                                                      \
	"v_cmp_eq_u32_e32 vcc, s77, v35\n\t"                                                                               \
	"s_and_b64 vcc, vcc, s[84:85]\n\t"                                                                                 \
	"s_andn2_b64 s[86:87], s[86:87], vcc\n\t"                                                                          \
	"s_and_b64 vcc, vcc, s[90:91]\n\t"                                                                                 \
	"s_or_b64 s[88:89], s[88:89], vcc\n\t"                                                                             \
	"s_cmp_lg_u64 s[86:87], 0\n\t"                                                                                     \
	"s_cbranch_scc1 40b\n\t"                                                                                           \
	"s_andn2_b64 s[84:85], s[84:85], s[88:89]\n\t"                                                                     \
	"42:\n\t"                                                                                                          \
	"s_mov_b64 exec, s[84:85]\n\t"                                                                                     \
	"global_store_short v35, v39, %[spill]\n\t"                                                                        \
	"s_mov_b64 exec, -1\n\t"                                                                                           \
	"43:\n\t"                                                                                                          \
	"s_mov_b32 %[epoch], s92\n\t"

/* the full 2^p-byte table in global memory, an occupancy bitmap and the keyed exchange array in LDS
 * (parse_lean<TAB_GLOBAL>): the slot is the hash; v36 = my entry's offset, v37 = my bitmap word, v38 = my bit */
#define CSNAPPY_ISA_TABLE_GTAB \
	"s_sub_u32 s81, 33, %[q1]\n\t"                     /* probes the scan in progress has left */                      \
	"s_add_u32 s92, %[epoch], -1\n\t"                  /* this step's epoch field */                                   \
	"s_lshl_b32 s93, s92, 22\n\t"                                                                                      \
	"s_waitcnt vmcnt(0)\n\t"                           /* the window, and the table store of the step before */        \
	CSNAPPY_ISA_OWN_BYTES                                                                                              \
	"v_mul_lo_u32 v59, v44, %[mul]\n\t"                                                                                \
	"v_lshrrev_b32_e32 v42, %[shift], v59\n\t"         /* my slot: the hash */                                         \
	"v_bfe_u32 v59, v59, %[shm1], 1\n\t"               /* check bit: one more bit of it */                             \
	"v_lshl_or_b32 v39, v59, 15, v40\n\t"              /* my entry, if I am inserted */                                \
	"v_and_b32_e32 v56, %[smask], v42\n\t"             /* key of the exchange array */                                 \
	"v_lshl_or_b32 v57, v42, 7, %[lane]\n\t"                                                                           \
	"v_lshl_add_u32 v56, v56, 2, %[sbase]\n\t"                                                                         \
	"v_or_b32_e32 v57, s93, v57\n\t"                   /* tag: epoch, slot, lane */                                    \
	"ds_wrxchg_rtn_b32 v58, v56, v57\n\t"              /* comes back with the tag the nearest lower lane with my key left */\
	"v_lshrrev_b32_e32 v48, 3, v42\n\t"                                                                                \
	"v_and_b32_e32 v37, 0xffc, v48\n\t"                /* my word of the occupancy bitmap */                           \
	"ds_read_b32 v49, v37\n\t"                                                                                         \
	"v_writelane_b32 v41, s81, 0\n\t"                  /* lane 0 searches what is left of the scan, the others 33 probes */\
	"v_lshlrev_b32_e32 v36, 1, v42\n\t"                /* my entry's offset in the table */                            \
	"v_lshlrev_b32_e64 v38, v42, 1\n\t"                /* my bit of the word */                                        \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
	"v_and_b32_e32 v49, v49, v38\n\t"                                                                                  \
	"v_cmp_ne_u32_e64 s[84:85], 0, v49\n\t"            /* written in this fragment */                                  \
	"v_lshrrev_b32_e32 v58, 22, v58\n\t"                                                                               \
	"v_mov_b32_e32 v57, %[zv]\n\t"                     /* (two wait states between the compare and the select) */      \
	"v_cndmask_b32_e64 v48, 0, v36, s[84:85]\n\t"      /* (an empty slot is not read) */                               \
	"global_load_ushort v56, v48, %[gtab]\n\t"         /* the entry */                                                 \
	"v_cmp_eq_u32_e64 s[62:63], s92, v58\n\t"          /* flagged: a lower lane of this step has my key */             \
	"s_waitcnt vmcnt(0)\n\t"                                                                                           \
	"v_cndmask_b32_e64 v56, v57, v56, s[84:85]\n\t"    /* empty: position 0's entry */                                 \
	"v_xor_b32_e32 v59, v56, v39\n\t"                                                                                  \
	"v_and_b32_e32 v34, 0x7fff, v56\n\t"                                                                               \
	"v_cmp_lt_u32_e32 vcc, v59, %[thr]\n\t"            /* check bits agree (thr: 0x8000; lane 0, insert-only: 0) */    \
	"s_mov_b64 s[70:71], vcc\n\t"                      /* the candidate can match at all */                            \
	"v_cndmask_b32_e64 v59, 0, v34, s[70:71]\n\t"                                                                      \
	"global_load_dwordx4 v[52:55], v59, %[src]\n\t"    /* candidate gather (no candidate: position 0) */               \
	"global_store_dwordx2 v43, v[50:51], %[R]\n\t"     /* the previous step's records, behind the gather */            \
	"s_waitcnt vmcnt(1)\n\t"                           /* the gather (the store may be on its way) */

#define CSNAPPY_ISA_GO_GTAB \
	"s_cmp_lt_u32 %[p0], %[limit64]\n\t"                                                                               \
	"s_cselect_b32 s82, %[q1], 99\n\t"                 /* next step in this loop too: < 33 */                          \
	"s_cmp_lt_u32 s92, 2\n\t"                          /* ... unless the tags' epoch field runs out: the C++ step starts the array over */\
	"s_cselect_b32 s82, 99, s82\n\t"

#if CSNAPPY_TIMING_TA == 4 /* timing experiment: what one more dependent round trip in front of the global-table step's window costs */
#define CSNAPPY_TIMING_ROUND_TRIP_G "global_load_dword v56, v49, %[src]\n\ts_waitcnt vmcnt(0)\n\t"
#else
#define CSNAPPY_TIMING_ROUND_TRIP_G ""
#endif
#define CSNAPPY_ISA_LOADS_GTAB \
	"v_add_u32_e32 v40, %[p0], %[lane]\n\t"            /* the next step's positions */                                 \
	"s_and_b32 s79, %[p0], -4\n\t"                     /* its window: aligned dwords from here, lane l takes dword l */\
	"v_lshl_add_u32 v49, %[lane], 2, s79\n\t"                                                                          \
	"v_min_u32_e32 v49, %[safemax4], v49\n\t"          /* (clamped: lanes >= 21 are not used, and the loads are harmless when the loop ends here) */\
	CSNAPPY_TIMING_ROUND_TRIP_G                                                                                        \
	"global_load_dword v44, v49, %[src]\n\t"

#define CSNAPPY_ISA_COMMIT_GTAB \
	"v_cmp_ge_u32_e32 vcc, s72, %[lane]\n\t"           /* lanes up to e_final */                                       \
	"s_andn2_b64 s[70:71], vcc, s[70:71]\n\t"          /* inserted lanes */                                            \
	"s_and_b64 s[84:85], s[62:63], s[70:71]\n\t"       /* of several with one slot only the highest may store (memory keeps no order) */\
	"s_cmp_eq_u64 s[84:85], 0\n\t"                                                                                     \
	"s_cbranch_scc1 41f\n\t"                                                                                           \
	"s_mov_b64 s[86:87], 0\n\t"                        /* one round per shared slot, highest lane first: the lower ones of its slot lose */\
	"40:\n\t"                                                                                                          \
	"s_flbit_i32_b64 s76, s[84:85]\n\t"                                                                                \
	"s_xor_b32 s76, s76, 63\n\t"                                                                                       \
	"v_readlane_b32 s77, v42, s76\n\t"                                                                                 \
	"s_bfm_b64 s[88:89], s76, 0\n\t"                                                                                   \
	"s_nop 0\n\t"                                                                                                      \
	"v_cmp_eq_u32_e32 vcc, s77, v42\n\t"                                                                               \
	"s_and_b64 vcc, vcc, s[70:71]\n\t"                                                                                 \
	"s_andn2_b64 s[84:85], s[84:85], vcc\n\t"                                                                          \
	"s_and_b64 vcc, vcc, s[88:89]\n\t"                                                                                 \
	"s_or_b64 s[86:87], s[86:87], vcc\n\t"                                                                             \
	"s_cmp_lg_u64 s[84:85], 0\n\t"                                                                                     \
	"s_cbranch_scc1 40b\n\t"                                                                                           \
	"s_andn2_b64 s[70:71], s[70:71], s[86:87]\n\t"                                                                     \
	"41:\n\t"                                                                                                          \
	"s_mov_b64 exec, s[70:71]\n\t"                                                                                     \
	"global_store_short v36, v39, %[gtab]\n\t"         /* table[slot] = my entry */                                    \
	"ds_or_b32 v37, v38\n\t"                           /* ... and the slot is occupied */                              \
	"s_mov_b64 exec, -1\n\t"                                                                                           \
	"s_mov_b32 %[epoch], s92\n\t"                      /* one epoch per step */

#define CSNAPPY_ISA_LOOP(ENTRY, TABLE, GO, LOADS, COMMIT)                                                           \
	"v_mov_b32_e32 v41, 32\n\t"                                                                                        \
	"s_mov_b32 s83, m0\n\t"                            /* (m0 is the compiler's: given back at 19) */                  \
	ENTRY                                                                                                              \
	CSNAPPY_ISA_P0                                                                                                     \
	"s_branch 12f\n\t"                                                                                                 \
	/* ================= BACK: the walk has left the step and took at least one copy ================= */            \
	"10:\n\t"                                                                                                          \
	CSNAPPY_ISA_P("v69")                                                                                               \
	"s_flbit_i32_b64 s72, s[68:69]\n\t"                                                                                \
	"s_xor_b32 s72, s72, 63\n\t"                       /* the last copy's lane */                                      \
	"v_mov_b32_e32 v58, %[nemit]\n\t"                  /* where the pending literal starts */                          \
	"v_readlane_b32 s73, v33, s72\n\t"                 /* c: the lane behind the last copy */                          \
	"v_add_u32_e32 v59, %[p0], v33\n\t"                /* where my copy ends */                                        \
	"s_bcnt1_i32_b64 s78, s[68:69]\n\t"                                                                                \
	"v_cndmask_b32_e64 v59, v58, v59, s[68:69]\n\t"                                                                    \
	"s_add_u32 %[nemit], %[p0], s73\n\t"               /* next_emit: behind the last copy */                           \
	"s_add_u32 s74, s73, 32\n\t"                                                                                       \
	"v_lshl_or_b32 v50, v34, 16, v40\n\t"              /* record: base | cand << 16 */                                 \
	"s_min_u32 s74, s74, 63\n\t"                       /* e: the last lane the step probes */                          \
	"s_add_u32 s75, s73, -1\n\t"                       /* c - 1 */                                                     \
	"v_add_u32_e32 v53, 1, v40\n\t"                                                                                    \
	"s_sub_u32 s76, s74, s75\n\t"                      /* probes of the scan behind the copy: e - c + 1 */             \
	"s_cmp_ge_u32 s73, 64\n\t"                         /* the copy leaves the step: re-match probe next */             \
	"s_cselect_b32 s75, s75, s74\n\t"                  /* lane 0 of the next step */                                   \
	"s_cselect_b32 %[q1], 0, s76\n\t"                                                                                  \
	"s_cselect_b32 s72, s72, s74\n\t"                  /* e_final: the last lane that is inserted */                   \
	"s_add_u32 %[p0], %[p0], s75\n\t"                  /* pz */                                                        \
	/* ---- the cursor is known: the next step's loads go out first (the records and the commit run under them) ---- */\
	GO                                                                                                                 \
	LOADS                                                                                                              \
	/* ---- records: the end of the nearest taken copy below every lane is the running maximum of the copies' ends ---- */\
	"v_max_u32_dpp v59, v59, v59 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"s_mov_b64 vcc, s[68:69]\n\t"                                                                                      \
	"v_mbcnt_lo_u32_b32 v48, vcc_lo, 0\n\t"                                                                            \
	"v_max_u32_dpp v59, v59, v59 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"v_mbcnt_hi_u32_b32 v48, vcc_hi, v48\n\t"          /* taken lanes below me */                                      \
	"s_nop 0\n\t"                                                                                                      \
	"v_max_u32_dpp v59, v59, v59 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"v_add_lshl_u32 v48, v48, %[nev], 3\n\t"           /* my record's byte offset */                                   \
	"s_add_u32 %[nev], %[nev], s78\n\t"                                                                                \
	"v_max_u32_dpp v59, v59, v59 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"v_cndmask_b32_e64 v43, %[norec], v48, s[68:69]\n\t"                                                               \
	"s_nop 0\n\t"                                                                                                      \
	"v_max_u32_dpp v59, v59, v59 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                                          \
	"s_nop 1\n\t"                                                                                                      \
	"v_max_u32_dpp v59, v59, v59 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                                          \
	"s_nop 1\n\t"                                                                                                      \
	"v_mov_b32_dpp v58, v59 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"/* end of the nearest copy below me (none: next_emit) */\
	"v_lshl_or_b32 v51, v58, 16, v32\n\t"              /* record: copy_len | lit_start << 16 */                        \
	"v_cmp_lt_u32_e64 s[70:71], v53, v58\n\t"          /* strictly inside a copy: never inserted */                    \
	/* ---- 15: this step's commit ---- */                                                                             \
	"15:\n\t"                                                                                                          \
	CSNAPPY_ISA_P("v70")                                                                                               \
	COMMIT                                                                                                             \
	"s_cmp_lt_u32 s82, 33\n\t"                                                                                         \
	"s_cbranch_scc0 19f\n\t"                                                                                           \
	/* ================= FRONT ================= */                                                                  \
	"12:\n\t"                                                                                                          \
	CSNAPPY_ISA_P("v65")                                                                                               \
	CSNAPPY_ISA_PSTEP                                                                                                  \
	TABLE                                                                                                              \
	"v_xor_b32_e32 v52, v52, v44\n\t"                                                                                  \
	"v_xor_b32_e32 v53, v53, v45\n\t"                                                                                  \
	"v_xor_b32_e32 v54, v54, v46\n\t"                                                                                  \
	"v_xor_b32_e32 v55, v55, v47\n\t"                                                                                  \
	CSNAPPY_ISA_PREFIX16("v52", "v53", "v54", "v55")                                                                   \
	"v_cndmask_b32_e64 v32, 0, v52, s[70:71]\n\t"      /* lane-local match length, 0..16 */                            \
	"v_cmp_lt_u32_e64 s[64:65], 3, v32\n\t"            /* matches */                                                   \
	"v_cmp_eq_u32_e32 vcc, 16, v32\n\t"                /* may be longer */                                             \
	"v_add_u32_e32 v33, %[lane], v32\n\t"              /* lane of the re-match probe behind my match */                \
	"s_or_b64 s[64:65], s[64:65], s[62:63]\n\t"        /* stops of the chain: matches and flagged lanes */             \
	"s_or_b64 s[66:67], vcc, s[62:63]\n\t"             /* ... that need a visit */                                     \
	"v_lshrrev_b64 v[48:49], v33, s[64:65]\n\t"                                                                        \
	"v_sub_u32_e32 v54, 63, v33\n\t"                   /* lanes left behind my match (negative: none) */               \
	"v_ffbl_b32_e32 v49, v49\n\t"                                                                                      \
	"v_ffbl_b32_e32 v48, v48\n\t"                                                                                      \
	"v_add_u32_e64 v49, v49, 32 clamp\n\t"                                                                             \
	"v_min_i32_e32 v54, v54, v41\n\t"                  /* ... and probes */                                            \
	"v_min3_u32 v48, v48, v49, 64\n\t"                 /* distance to the next stop */                                 \
	"v_add_u32_e32 v55, v33, v48\n\t"                  /* its lane */                                                  \
	"v_cmp_le_i32_e32 vcc, v48, v54\n\t"                                                                               \
	"v_lshrrev_b64 v[52:53], v55, s[66:67]\n\t"                                                                        \
	"v_and_b32_e32 v52, 1, v52\n\t"                                                                                    \
	"v_lshl_or_b32 v56, v52, 7, v55\n\t"               /* lane | 128: a special one */                                 \
	"v_cndmask_b32_e32 v35, 64, v56, vcc\n\t"          /* next stop of the chain if my match is taken (64: none here) */ \
	"s_mov_b64 s[68:69], 0\n\t"                                                                                        \
	CSNAPPY_ISA_P("v68")                                                                                               \
	"v_readlane_b32 s80, v35, 0\n\t"                   /* the walk: lane 0 holds the first stop */                     \
	"s_nop 0\n\t"                                                                                                      \
	"s_cmp_lt_u32 s80, 64\n\t"                                                                                         \
	"s_cbranch_scc0 13f\n\t"                           /* none, or a special lane */                                   \
	CSNAPPY_ISA_HOPS                                                                                                   \
	"s_cmp_eq_u32 s80, 64\n\t"                         /* the walk left the step */                                    \
	"s_cbranch_scc1 10b\n\t"                                                                                           \
	/* ================= VISIT: t >= 128, lane t & 63 is special (parse_lean's visits()) ================= */       \
	"13:\n\t"                                                                                                          \
	"s_cmp_lt_u32 s80, 128\n\t"                                                                                        \
	"s_cbranch_scc1 14f\n\t"                           /* 64 / 65: the walk left the step */                           \
	"s_and_b32 s73, s80, 63\n\t"                       /* i */                                                         \
	"s_mov_b32 m0, s73\n\t"                            /* (v_writelane takes one SGPR: its lane select goes through m0) */ \
	"v_readlane_b32 s74, v32, s73\n\t"                 /* L: its lane-local match length */                            \
	"s_bitcmp1_b64 s[62:63], s73\n\t"                                                                                  \
	"s_cbranch_scc0 30f\n\t"                           /* not flagged: a match of 16 bytes that may be longer */       \
	/* ---- a flagged lane: its candidate is the latest position inserted for its slot -- the highest lane    \
	 * below it with the same slot that this step inserts (not strictly inside a copy of the chain), whose   \
	 * bytes are that lane's own 16 bytes -- else the table entry it compared with ---- */                          \
	"v_readlane_b32 s75, v42, s73\n\t"                 /* its slot */                                                  \
	"s_bfm_b64 s[84:85], s73, 0\n\t"                   /* lanes below i */                                             \
	"s_nop 0\n\t"                                                                                                      \
	"v_cmp_eq_u32_e32 vcc, s75, v42\n\t"                                                                               \
	"s_and_b64 s[86:87], vcc, s[60:61]\n\t"                                                                            \
	"s_and_b64 s[86:87], s[86:87], s[84:85]\n\t"       /* same: lanes below i with its slot */                         \
	"s_cmp_eq_u64 s[86:87], 0\n\t"                                                                                     \
	"s_cbranch_scc1 24f\n\t"                                                                                           \
	"s_cmp_eq_u64 s[68:69], 0\n\t"                     /* no copy taken yet: every lane below was inserted */          \
	"s_cbranch_scc1 23f\n\t"                                                                                           \
	/* the highest of `same` tested on the scalar unit: inside the nearest taken copy below it? */                  \
	"s_flbit_i32_b64 s76, s[86:87]\n\t"                                                                                \
	"s_xor_b32 s76, s76, 63\n\t"                       /* jh */                                                        \
	"s_bfm_b64 s[88:89], s76, 0\n\t"                                                                                   \
	"s_and_b64 s[88:89], s[88:89], s[68:69]\n\t"       /* taken lanes below jh */                                      \
	"s_cmp_eq_u64 s[88:89], 0\n\t"                                                                                     \
	"s_cbranch_scc1 23f\n\t"                                                                                           \
	"s_flbit_i32_b64 s77, s[88:89]\n\t"                                                                                \
	"s_xor_b32 s77, s77, 63\n\t"                                                                                       \
	"v_readlane_b32 s77, v33, s77\n\t"                 /* where the nearest of them ends */                            \
	"s_add_u32 s78, s76, 1\n\t"                                                                                        \
	"s_cmp_lt_u32 s78, s77\n\t"                                                                                        \
	"s_cbranch_scc0 23f\n\t"                           /* jh was inserted */                                           \
	/* it was not: settle all of `same` at once -- the end of the nearest taken copy below every lane is the  \
	 * running maximum of the taken lanes' ends (they grow along the chain) */                                      \
	"v_cndmask_b32_e64 v56, 0, v33, s[68:69]\n\t"                                                                      \
	"v_add_u32_e32 v58, 1, %[lane]\n\t"                                                                                \
	"s_nop 0\n\t"                                                                                                      \
	"v_max_u32_dpp v56, v56, v56 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"s_nop 1\n\t"                                                                                                      \
	"v_max_u32_dpp v56, v56, v56 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"s_nop 1\n\t"                                                                                                      \
	"v_max_u32_dpp v56, v56, v56 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"s_nop 1\n\t"                                                                                                      \
	"v_max_u32_dpp v56, v56, v56 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                                             \
	"s_nop 1\n\t"                                                                                                      \
	"v_max_u32_dpp v56, v56, v56 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                                          \
	"s_nop 1\n\t"                                                                                                      \
	"v_max_u32_dpp v56, v56, v56 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                                          \
	"v_mov_b32_e32 v57, 0\n\t"                                                                                         \
	"s_nop 0\n\t"                                                                                                      \
	"v_mov_b32_dpp v57, v56 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                                 \
	"s_nop 0\n\t"                                                                                                      \
	"v_cmp_lt_u32_e32 vcc, v58, v57\n\t"               /* lane + 1 < that end: never inserted */                       \
	"s_andn2_b64 s[86:87], s[86:87], vcc\n\t"                                                                          \
	"s_cmp_eq_u64 s[86:87], 0\n\t"                                                                                     \
	"s_cbranch_scc1 24f\n\t"                                                                                           \
	"23:\n\t"                                          /* j: the highest lane of `same`; its bytes against everyone's */ \
	"s_flbit_i32_b64 s76, s[86:87]\n\t"                                                                                \
	"s_xor_b32 s76, s76, 63\n\t"                                                                                       \
	"v_readlane_b32 s88, v44, s76\n\t"                                                                                 \
	"v_readlane_b32 s89, v45, s76\n\t"                                                                                 \
	"v_readlane_b32 s90, v46, s76\n\t"                                                                                 \
	"v_readlane_b32 s91, v47, s76\n\t"                                                                                 \
	"s_add_u32 s77, %[p0], s76\n\t"                    /* its position */                                              \
	"s_nop 0\n\t"                                                                                                      \
	"v_xor_b32_e32 v52, s88, v44\n\t"                                                                                  \
	"v_xor_b32_e32 v53, s89, v45\n\t"                                                                                  \
	"v_xor_b32_e32 v54, s90, v46\n\t"                                                                                  \
	"v_xor_b32_e32 v55, s91, v47\n\t"                                                                                  \
	CSNAPPY_ISA_PREFIX16("v52", "v53", "v54", "v55")                                                                   \
	"s_nop 0\n\t"                                                                                                      \
	"v_readlane_b32 s74, v52, s73\n\t"                 /* L: lane i's match length against it */                       \
	"v_writelane_b32 v34, s77, m0\n\t"                 /* and its candidate */                                         \
	"24:\n\t"                                                                                                          \
	"s_cmp_lt_u32 s74, 4\n\t"                                                                                          \
	"s_cbranch_scc0 30f\n\t"                                                                                           \
	/* ---- no match at the flagged lane: on to the next stop of the current window ---- */                         \
	"s_mov_b32 s77, s81\n\t"                           /* probes left: of the scan in progress ... */                  \
	"s_cmp_eq_u64 s[68:69], 0\n\t"                                                                                     \
	"s_cbranch_scc1 25f\n\t"                                                                                           \
	"s_flbit_i32_b64 s77, s[68:69]\n\t"                                                                                \
	"s_xor_b32 s77, s77, 63\n\t"                                                                                       \
	"v_readlane_b32 s77, v33, s77\n\t"                                                                                 \
	"s_add_u32 s77, s77, 32\n\t"                       /* ... or the 33 behind the last copy */                        \
	"25:\n\t"                                                                                                          \
	"s_bitset1_b64 s[84:85], s73\n\t"                  /* lanes up to i */                                             \
	"s_andn2_b64 s[84:85], s[64:65], s[84:85]\n\t"     /* stops behind i */                                            \
	"s_ff1_i32_b64 s78, s[84:85]\n\t"                                                                                  \
	"s_mov_b32 s80, 65\n\t"                                                                                            \
	"s_cmp_lt_i32 s78, 0\n\t"                                                                                          \
	"s_cbranch_scc1 14f\n\t"                           /* none */                                                      \
	"s_cmp_gt_u32 s78, s77\n\t"                                                                                        \
	"s_cbranch_scc1 14f\n\t"                           /* beyond the window */                                         \
	"s_bitcmp1_b64 s[66:67], s78\n\t"                                                                                  \
	"s_cselect_b32 s79, 128, 0\n\t"                                                                                    \
	"s_or_b32 s80, s78, s79\n\t"                                                                                       \
	"s_cmp_lt_u32 s80, 64\n\t"                                                                                         \
	"s_cbranch_scc0 13b\n\t"                                                                                           \
	CSNAPPY_ISA_HOPS                                                                                                   \
	"s_branch 13b\n\t"                                                                                                 \
	/* ---- a match of L >= 4 bytes at lane i; L == 16 may be longer: FindMatchLength beyond the lane-local  \
	 * 16 bytes, 512 bytes a round (csnappy_compress.c:252-295), never reading past the fragment ---- */            \
	"30:\n\t"                                                                                                          \
	"s_cmp_eq_u32 s74, 16\n\t"                                                                                         \
	"s_cbranch_scc0 35f\n\t"                                                                                           \
	"s_add_u32 s76, %[p0], s73\n\t"                                                                                    \
	"s_add_u32 s76, s76, 16\n\t"                       /* mb = base + 16 */                                            \
	"s_cmp_lt_u32 s76, %[n]\n\t"                                                                                       \
	"s_cbranch_scc0 35f\n\t"                                                                                           \
	"v_readlane_b32 s77, v34, s73\n\t"                                                                                 \
	"s_sub_u32 s78, %[n], s76\n\t"                     /* lim: bytes left behind mb */                                 \
	"s_mov_b32 s79, 0\n\t"                             /* done */                                                      \
	"v_lshlrev_b32_e32 v56, 3, %[lane]\n\t"                                                                            \
	"s_add_u32 s77, s77, 16\n\t"                       /* ma = cand + 16 */                                            \
	"31:\n\t"                                                                                                          \
	"v_add_u32_e32 v57, s79, v56\n\t"                  /* o = done + 8 * lane */                                       \
	"v_sub_u32_e32 v58, s78, v57\n\t"                                                                                  \
	"v_cmp_gt_u32_e64 s[84:85], s78, v57\n\t"          /* o < lim: my eight bytes begin inside the fragment */         \
	"v_min_u32_e32 v58, 8, v58\n\t"                    /* r: how many of them are inside */                            \
	"v_sub_u32_e32 v59, 8, v58\n\t"                    /* back: read the eight bytes that END at the fragment's end */ \
	"v_sub_u32_e32 v52, v57, v59\n\t"                                                                                  \
	"v_cndmask_b32_e64 v52, 0, v52, s[84:85]\n\t"                                                                      \
	"v_add_u32_e32 v53, s77, v52\n\t"                                                                                  \
	"v_add_u32_e32 v52, s76, v52\n\t"                                                                                  \
	"global_load_dwordx2 v[48:49], v53, %[src]\n\t"                                                                    \
	"global_load_dwordx2 v[54:55], v52, %[src]\n\t"                                                                    \
	"v_lshlrev_b32_e32 v59, 3, v59\n\t"                                                                                \
	"v_add_u32_e32 v53, 8, v57\n\t"                                                                                    \
	"s_waitcnt vmcnt(0)\n\t"                                                                                           \
	"v_xor_b32_e32 v48, v48, v54\n\t"                                                                                  \
	"v_xor_b32_e32 v49, v49, v55\n\t"                                                                                  \
	"v_lshrrev_b64 v[48:49], v59, v[48:49]\n\t"        /* drop the bytes in front of o */                              \
	"v_cmp_le_u32_e32 vcc, s78, v53\n\t"               /* o + 8 >= lim: the fragment ends in my bytes */               \
	"v_ffbl_b32_e32 v52, v48\n\t"                                                                                      \
	"v_ffbl_b32_e32 v54, v49\n\t"                                                                                      \
	"v_add_u32_e64 v54, v54, 32 clamp\n\t"                                                                             \
	"v_min3_u32 v52, v52, v54, 64\n\t"                                                                                 \
	"v_lshrrev_b32_e32 v52, 3, v52\n\t"                /* equal bytes (8: all) */                                      \
	"v_min_u32_e32 v52, v52, v58\n\t"                                                                                  \
	"v_cndmask_b32_e64 v52, 0, v52, s[84:85]\n\t"      /* m8 (lanes beyond the fragment: 0) */                         \
	"v_cmp_gt_u32_e64 s[86:87], 8, v52\n\t"            /* the match ends in my bytes */                                \
	"s_or_b64 s[86:87], s[86:87], vcc\n\t"                                                                             \
	"s_cmp_lg_u64 s[86:87], 0\n\t"                                                                                     \
	"s_cbranch_scc1 32f\n\t"                                                                                           \
	"s_add_u32 s79, s79, 512\n\t"                                                                                      \
	"s_branch 31b\n\t"                                                                                                 \
	"32:\n\t"                                                                                                          \
	"s_ff1_i32_b64 s88, s[86:87]\n\t"                  /* the first lane it ends in */                                 \
	"v_readlane_b32 s89, v52, s88\n\t"                                                                                 \
	"s_lshl_b32 s88, s88, 3\n\t"                                                                                       \
	"s_add_u32 s79, s79, s88\n\t"                                                                                      \
	"s_add_u32 s79, s79, s89\n\t"                                                                                      \
	"s_add_u32 s74, s79, 16\n\t"                       /* L */                                                         \
	"35:\n\t"                                          /* the copy is taken */                                         \
	"s_add_u32 s75, s73, s74\n\t"                      /* the lane behind it */                                        \
	"v_writelane_b32 v32, s74, m0\n\t"                                                                                 \
	"v_writelane_b32 v33, s75, m0\n\t"                                                                                 \
	"s_bitset1_b64 s[68:69], s73\n\t"                                                                                  \
	/* the next stop behind it, on the scalar unit (64: the re-match probe falls outside the step; 65: none  \
	 * of the 33 probes behind the copy is a stop) */                                                               \
	"s_mov_b32 s80, 64\n\t"                                                                                            \
	"s_cmp_ge_u32 s75, 64\n\t"                                                                                         \
	"s_cbranch_scc1 14f\n\t"                                                                                           \
	"s_lshr_b64 s[84:85], s[64:65], s75\n\t"                                                                           \
	"s_ff1_i32_b64 s76, s[84:85]\n\t"                                                                                  \
	"s_mov_b32 s80, 65\n\t"                                                                                            \
	"s_cmp_gt_u32 s76, 32\n\t"                         /* (none: -1) */                                                \
	"s_cbranch_scc1 14f\n\t"                                                                                           \
	"s_add_u32 s76, s76, s75\n\t"                                                                                      \
	"s_cmp_gt_u32 s76, 63\n\t"                                                                                         \
	"s_cbranch_scc1 14f\n\t"                                                                                           \
	"s_bitcmp1_b64 s[66:67], s76\n\t"                                                                                  \
	"s_cselect_b32 s77, 128, 0\n\t"                                                                                    \
	"s_or_b32 s80, s76, s77\n\t"                                                                                       \
	"s_cmp_lt_u32 s80, 64\n\t"                                                                                         \
	"s_cbranch_scc0 13b\n\t"                                                                                           \
	CSNAPPY_ISA_HOPS                                                                                                   \
	"s_branch 13b\n\t"                                                                                                 \
	/* ================= 14: the walk has left the step ================= */                                        \
	"14:\n\t"                                                                                                          \
	"s_cmp_lg_u64 s[68:69], 0\n\t"                                                                                     \
	"s_cbranch_scc1 10b\n\t"                                                                                           \
	/* no copy in the whole step: the scan goes on behind its last probe (csnappy_compress.c:535-552);       \
	 * nothing to record; every lane up to that probe is inserted */                                                \
	CSNAPPY_ISA_P("v69")                                                                                               \
	"s_min_u32 s72, s81, 63\n\t"                       /* e_final */                                                   \
	"v_mov_b32_e32 v50, 0\n\t"                                                                                         \
	"v_mov_b32_e32 v51, 0\n\t"                                                                                         \
	"v_mov_b32_e32 v43, %[norec]\n\t"                                                                                  \
	"s_add_u32 %[q1], %[q1], s72\n\t"                                                                                  \
	"s_add_u32 %[p0], %[p0], s72\n\t"                                                                                  \
	"s_mov_b64 s[70:71], 0\n\t"                                                                                        \
	GO                                                                                                                 \
	LOADS                                                                                                              \
	"s_branch 15b\n\t"                                                                                                 \
	"19:\n\t"                                                                                                          \
	"s_waitcnt vmcnt(0)\n\t"                           /* (loads into registers that are the block's own) */           \
	"s_mov_b32 m0, s83\n\t"

template <int TAB, bool SPILL, bool PROF = false, bool ORD = true>
DEVINL void parse_lean(const CompressArgs &A, const Frag &F)
{
	/* PROF: s_memtime phase counters (debug kernels only; tools/phase_lean.py) */
	unsigned long long pt[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long pt_last = 0, pt_begin = 0, pn_steps = 0, pn_special = 0, pn_hops = 0, pn_sparse = 0;
	unsigned long long pn_tabbed = 0, pn_gathered = 0, pn_match4 = 0, pn_flagged = 0;
	auto tick = [&](int k) {
		if (PROF) {
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			pt[k] += now - pt_last;
			pt_last = now;
		}
	};
	if (PROF)
		pt_begin = pt_last = __builtin_amdgcn_s_memtime();
	constexpr bool DENSE = TAB == TAB_LDS_DENSE, GTAB = TAB == TAB_GLOBAL;
	static_assert(!SPILL || DENSE, "the spill-over belongs to the dense table");
	using FT = FilterTag<DENSE ? 13 : 15>; /* dense ids are < 8192; a hash can have 15 bits */
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	const uint32_t lane = threadIdx.x;
	const uint32_t n = F.n, shift = F.shift;
	const uint8_t *src = F.src;
	uint2 *R = F.R;
	uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
	uint32_t *tab32 = reinterpret_cast<uint32_t *>(smem);
	/* TAB_GLOBAL: the full 2^p-byte table lies in the fragment's workspace region; LDS holds one bit per
	 * slot, "written in this fragment".  A clear bit means the slot is empty (the reference's zeroed
	 * table: candidate position 0) without touching memory, so the global table is never cleared and
	 * never read for empty slots. */
	uint16_t *gtab = reinterpret_cast<uint16_t *>(F.region);
	uint32_t *occ = reinterpret_cast<uint32_t *>(smem);
	const uint16_t *ids = reinterpret_cast<const uint16_t *>(F.region);
	uint16_t *spill = reinterpret_cast<uint16_t *>(F.region + A.spill_off);
	/* TW (round 5): tables in LDS find the lanes of a step that share a slot through the table itself.
	 * gfx950's LDS serves the lanes of one instruction that hit one address in ascending lane order
	 * (tools/ubench/lds_order.hip; csnappy_hip_compress_batch checks it on the device before the first
	 * launch): every lane reads its entry, then adds 1 to its 16-bit half of the entry's dword with a
	 * RETURNING add -- what comes back is the entry plus the number of LOWER lanes with the same slot, so
	 * "differs from the entry" is exactly "flagged".  The commit puts the entries back (one store) and
	 * writes the inserted lanes' (a second store: of several lanes with one slot the highest survives,
	 * which is the reference's order of updates).  The two conflict filters (two atomics, two reads,
	 * thirty instructions of tag arithmetic per step, false alarms, 1 KiB of LDS) and the commit's
	 * dedupe rounds are gone from those placements; the few lanes of a SPILL fragment whose bucket lives
	 * in HBM keep one small filter among themselves, carved out of the table's tail.  The global table
	 * finds sharing with one returning exchange on a keyed array (see there).
	 * The add may carry from the low half of a dword into the high one (an entry within 63 of 0xffff).
	 * That can only FLAG a lane of the high half that has no sharer -- its half comes back one too high --
	 * never hide one (a half's own count is <= 63, the carry <= 1), and a flagged lane's visit is exact
	 * whatever flagged it.  The commit undoes the adds by SUBTRACTING them again (round 6; commutative,
	 * exact modulo 2^32 whatever carried), so no entry is left changed; until then the lanes that were
	 * not inserted stored their entry back, a carry into a half nobody held was never undone, and the
	 * last steps of a full fragment had to flag every lane instead (`late`).
	 * Lanes without a bucket (id 0) read, bump and restore the dummy dword, entries 0 and 1.
	 * ORD = false (the device's LDS does not keep that order, or CSNAPPY_HIP_NO_LDS_ORDER=1): no add,
	 * no exchange; every lane with a bucket is flagged, so every stop of the chain is resolved from
	 * the lower lanes' registers, and the commit lets one lane per slot store (its dedupe rounds).
	 * Exact on any hardware, several times slower. */
	constexpr bool TW = !GTAB;
	constexpr bool FILT = GTAB || SPILL;
	/* TW + SPILL: the last kSpillFilterSlots entries of the LDS table are the spilled lanes' filter */
	const uint32_t dense_cap = (TW && SPILL) ? A.dense_cap - kSpillFilterSlots : A.dense_cap;
	uint32_t *S = reinterpret_cast<uint32_t *>(smem + (TW ? 2 * dense_cap : A.lds0));
	const uint32_t s_entries = TW ? kSpillFilterEntries : A.s_entries; /* keys of that array (a power of two) */
	const uint32_t smask = s_entries - 1;

	uint32_t nev = 0;       /* records written */
	uint32_t next_emit = 0; /* csnappy_compress.c:496 */
	const uint64_t lt_mask = (1ull << lane) - 1;
	bool stuck = false;

	/* An entry is position | check bit << 15 (one more bit of the position's hash: a candidate whose bit
	 * differs cannot match and is not gathered).  memset(table, 0), csnappy_compress.c:501, makes an
	 * empty slot mean position 0; the LDS tables (TW, no spill-over) are filled with position 0's entry
	 * instead, so that a lane need not tell an empty slot from a written one. */
	constexpr bool CHK0_INIT = TW;
	uint32_t chk0 = 0;
	if (n >= 4) {
		uint32_t first4;
		__builtin_memcpy(&first4, src, 4);
		chk0 = ((first4 * kHashMul) >> (shift - 1)) & 1u;
	}
	if (n >= kMargin) {
		const uint32_t zb = GTAB ? (1u << F.ws) >> 4 : DENSE ? 2 * dense_cap : 1u << F.ws;
		const uint32_t zv = CHK0_INIT ? chk0 * 0x80008000u : 0u;
		uint4 *z4 = reinterpret_cast<uint4 *>(smem);
		for (uint32_t k = lane; k < (zb + 15) >> 4; k += 64)
			z4[k] = make_uint4(zv, zv, zv, zv);
		if (FILT) {
			uint4 *s4 = reinterpret_cast<uint4 *>(S);
			for (uint32_t k = lane; k < (s_entries >> 2); k += 64)
				s4[k] = make_uint4(~0u, ~0u, ~0u, ~0u);
		}
		wave_lds_fence();
	}

	/* FindMatchLength beyond the lane-local 16 bytes: 512 B per iteration, :252-295 */
	auto extend = [&](uint32_t cnd, uint32_t base) __attribute__((always_inline)) -> uint32_t {
		const uint32_t ma = cnd + kLocalMatch, mb = base + kLocalMatch, lim = n - mb;
		uint32_t done = 0;
		for (;;) {
			const uint32_t o = done + lane * 8;
			uint32_t m8 = 0;
			bool term = true;
			if (o < lim) {
				/* the last few bytes of the fragment: never read past the input -- take the eight
				 * bytes that END at the fragment's end and drop the ones in front of o */
				const uint32_t r = min(8u, lim - o), back = 8 - r;
				uint64_t xa, xb;
				__builtin_memcpy(&xa, src + ma + o - back, 8);
				__builtin_memcpy(&xb, src + mb + o - back, 8);
				const uint64_t x = (xa ^ xb) >> (8 * back);
				const uint32_t z = (uint32_t)__ffsll((unsigned long long)x); /* 0 when x == 0 */
				m8 = z ? min((z - 1) >> 3, r) : r;
				term = m8 < 8 || o + 8 >= lim;
			}
			const uint64_t tmask = ballot64(term);
			if (tmask) {
				const uint32_t t = first_lane(tmask);
				return done + 8 * t + rdlane(m8, t);
			}
			done += 512;
		}
	};

	/* (a 15-byte fragment has ip_limit 0: no probe ever happens, it is one literal like n < 15) */
	if (n > kMargin) {
		const uint32_t ip_limit = n - kMargin;
		/* the cursor: q1 = 1 + index of the next probe of the current scan (q1 == 0: a copy just ended,
		 * its re-match probe comes first), and pz = the position lane 0 of the next step takes -- the scan's
		 * start is pz - q1 + 2 (only sparse steps need it) */
		uint32_t pz = 0, q1 = 1;
		uint32_t epoch = FT::kEpochs;
		bool fin = false;

		/* the lanes' 16 bytes and bucket ids are fetched one step ahead, as soon as the next
		 * step's cursor is known */
		uint32_t raw0, raw1, raw2, raw3, sid = kNoBucket;
		uint32_t pos = 0;
		bool valid = false;
		auto place = [&]() __attribute__((always_inline)) {
			pos = pz + lane;
			valid = pos < ip_limit;
			if (__builtin_expect(q1 > 32, 0)) {
				/* sparse: the next 64 probes of the stride rule; a probe happens only if the
				 * NEXT position is still <= ip_limit (:542-544) */
				const uint32_t s = pz - q1 + 2;
				pos = scan_pos(s, q1 - 1 + lane);
				valid = scan_pos(s, q1 + lane) <= ip_limit;
			}
			pos = valid ? pos : 0u;
			uint4 v;
			__builtin_memcpy(&v, src + pos, 16);
			/* (plain loads: consecutive steps' ids share lines, and since the id lines are no longer
			 * touched ahead the L1 is where the second step finds them -- with the streaming hint
			 * round 3 gave them: 9.32 against 9.24 ms per GiB of compress on text) */
			const uint16_t idv = DENSE ? ids[pos] : (uint16_t)kNoBucket;
			raw0 = v.x;
			raw1 = v.y;
			raw2 = v.z;
			raw3 = v.w;
			sid = idv;
		};
		place();

		uint32_t guard = 0; /* every step probes or inserts at least one new position: a logic error must not hang the GPU */
		/* the previous step's records, stored BEHIND this step's candidate gather: gfx9 counts loads and
		 * stores in one vmcnt, so a store issued in front of the gather would have to be acknowledged
		 * before the gather's wait ends.  Every lane stores, the lanes without a record into the last slot
		 * of the fragment's record region (which no record reaches), so that the number of memory
		 * instructions behind the gather is the same on every path */
		uint2 prec = make_uint2(0, 0);
		const uint32_t no_rec_off = (A.rec_cap - 1) * 8; /* byte offset of that slot */
		uint32_t prec_off = no_rec_off;
#define CSNAPPY_FLUSH_PREC()                                                                       \
	(*reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(R) + prec_off) =      \
		 *reinterpret_cast<unsigned long long *>(&prec))

		/* ---- what a dense step knows about its lanes (set by the step's front half -- C++ below, or the
		 * hand-written one of the fast path -- and used by the visits of special lanes and the back half) ---- */
		uint32_t p0 = 0;                            /* position of lane 0 */
		uint32_t me0 = 0, me1 = 0, me2 = 0, me3 = 0; /* the lane's own 16 bytes */
		uint32_t slot = 0;                          /* its table slot (dense id, or hash) */
		uint32_t mlen = 0, cl = 0, cand = 0, nx = 0; /* match length, lane behind the match, candidate, next stop */
		uint32_t lim0 = 0, ulim = 64, t = 0;
		uint64_t tmask = 0, cmask = 0, stopmask = 0, special = 0, taken = 0;

		/* The hops of the walk, t = nx[t] until t >= 64, every lane passed marked in `taken`: the whole loop
		 * is one asm block, so that the wait states are spelled out here and do not hang on what the
		 * compiler happens to put between two of its iterations (its hazard recogniser does not look
		 * inside inline asm).  A v_readlane whose lane select was written by the vector unit -- the
		 * previous hop's v_readlane -- needs four wait states: s_nop 0, s_cmp, s_cbranch and the next
		 * hop's mark are those four; the mark in FRONT of the v_readlane is also the one wait state the
		 * first v_readlane needs behind the vector instruction that wrote nx.  t on entry comes from the
		 * scalar unit (a constant, or scalar arithmetic), which needs none.  Five instructions a hop. */
#define CSNAPPY_HOPS()                                                                             \
	asm volatile("1:\n\ts_bitset1_b64 %0, %1\n\tv_readlane_b32 %1, %2, %1\n\ts_nop 0\n\t"           \
		     "s_cmp_lt_u32 %1, 64\n\ts_cbranch_scc1 1b"                                     \
		     : "+s"(taken), "+s"(t) : "v"(nx) : "scc") /* do { taken |= 1ull << t; t = nx[t]; } while (t < 64) */

		/* the next stop behind a copy that ends in front of lane cc, found on the scalar unit (behind a
		 * visit): 64: the re-match probe falls outside the usable lanes; 65: none of the 33 probes
		 * behind the copy is a stop; lane | 128: that stop is a special lane */
		auto scalar_next = [&](uint32_t cc) __attribute__((always_inline)) -> uint32_t {
			if (cc >= ulim)
				return 64u;
			const uint64_t rest = stopmask >> cc;
			const uint32_t fm = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
			const uint32_t j = cc + fm;
			if (fm > 32 || j > 63)
				return 65u;
			return j | (((uint32_t)(special >> j) & 1u) << 7);
		};

		/* ---- the walk met a special lane (t >= 128): a flagged one, or a match of the lane-local 16 bytes
		 * that may be longer.  Settles it and walks on, until the walk leaves the step (t == 64 / 65). ---- */
		auto visits = [&]() __attribute__((always_inline)) {
			while (__builtin_expect(t >= 128, 0)) {
				const uint32_t i = t & 63u;
				uint32_t L = rdlane(mlen, i);
				if (PROF)
					pn_special++;
				if ((cmask >> i) & 1) {
					/* ---- the chain probes a flagged lane ----
					 * Its candidate is the latest position inserted for its slot: the highest
					 * lane below it that this step inserts (not strictly inside a copy of the
					 * chain) and that has the same slot -- whose bytes are that lane's own 16
					 * bytes -- else the table value it already compared with. */
					const uint32_t slot_i = rdlane(slot, i);
					uint64_t same = ballot64(slot == slot_i) & tmask & ((1ull << i) - 1);
					/* A lane strictly inside the nearest taken copy below it was never inserted.  The
					 * highest of `same` is tested on the scalar unit (it nearly always was inserted:
					 * no LDS round trip then); if it was not, all of them are settled at once with a
					 * ds_bpermute of the copies' ends (runs put the slot on every lane below: one by
					 * one on the scalar unit they cost pages a quarter of their speed) */
					if (same && taken) {
						const uint32_t jh = 63u - (uint32_t)__builtin_clzll(same);
						const uint64_t kb = taken & ((1ull << jh) - 1);
						if (kb && jh + 1 < rdlane(cl, 63u - (uint32_t)__builtin_clzll(kb))) {
							const uint64_t below = taken & lt_mask;
							const uint32_t jprev = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
							const uint32_t cprev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(jprev << 2), (int)cl);
							same &= ~ballot64(below != 0 && lane + 1 < cprev);
						}
					}
					if (same) {
						const uint32_t j = 63u - (uint32_t)__builtin_clzll(same);
						const uint32_t o0 = rdlane(me0, j), o1 = rdlane(me1, j);
						const uint32_t o2 = rdlane(me2, j), o3 = rdlane(me3, j);
						const uint64_t ylo = ((uint64_t)(me1 ^ o1) << 32) | (me0 ^ o0);
						const uint64_t yhi = ((uint64_t)(me3 ^ o3) << 32) | (me2 ^ o2);
						const uint32_t ml = common_prefix16(ylo, yhi);
						L = rdlane(ml, i);
						if (lane == i)
							cand = p0 + j;
					}
					if (L < 4) {
						/* no match: on to the next stop of the current window */
						const uint32_t lim_cur = taken ? rdlane(cl, 63u - (uint32_t)__builtin_clzll(taken)) + 32 : lim0;
						const uint64_t m = i < 63 ? stopmask & ((~0ull) << (i + 1)) : 0;
						const uint32_t i2 = m ? first_lane(m) : 64u;
						t = (i2 > lim_cur || i2 > 63) ? 65u : i2 | (((uint32_t)(special >> (i2 & 63u)) & 1u) << 7);
						if (t < 64)
							CSNAPPY_HOPS();
						continue;
					}
				}
				if (L == kLocalMatch && p0 + i + kLocalMatch < n) {
					/* longer than the lane-local cap: extend it wave-wide (it may leave the step) */
					L = kLocalMatch + extend(rdlane(cand, i), p0 + i);
				}
				if (lane == i) {
					mlen = L;
					cl = lane + L;
				}
				taken |= 1ull << i;
				t = scalar_next(i + L);
				if (t < 64)
					CSNAPPY_HOPS();
			}
		};

		/* the step loop in ISA for the common case (CSNAPPY_ISA_LOOP above) */
		constexpr bool FAST = CSNAPPY_FAST && DENSE && !SPILL && ORD && !PROF;
		constexpr bool FAST_G = CSNAPPY_FAST && GTAB && ORD && !PROF; /* the same loop around the global table */
		constexpr bool FAST_S = CSNAPPY_FAST && DENSE && SPILL && ORD && !PROF; /* ... and around the dense table with a spill-over */
		constexpr bool FAST_H = CSNAPPY_FAST && TAB == TAB_LDS_HASH && ORD && !PROF; /* ... and the table indexed by the hash */
		/* a step is for the loop when it is dense (q1 <= 32) and pz + 68 < ip_limit: every lane is valid, so is
		 * every lane's p0 + lane + 16 < n, and the 21 aligned dwords its window is loaded as end inside the fragment */
		/* (readfirstlane: the compiler computes the saturating subtraction on the vector unit) */
		const uint32_t limit64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ip_limit > 68 && !A.no_isa ? ip_limit - 68 : 0u));

		while (!fin && ++guard <= n) {
			if constexpr (FAST) if (q1 <= 32 && pz < limit64) {
				/* the values C++ hands over, in the registers the block keeps them in: the step's own
				 * bytes, id and position (place() requested them), the pending record */
				register uint32_t x0 asm("v44") = raw0, x1 asm("v45") = raw1, x2 asm("v46") = raw2, x3 asm("v47") = raw3;
				register uint32_t vsid asm("v42") = sid, vpos asm("v40") = pos;
				register uint32_t px asm("v50") = prec.x, py asm("v51") = prec.y, poff asm("v43") = prec_off;
				const uint32_t thr = lane == 0 ? 0u : 0x8000u;
				uint32_t nemit = next_emit;
#if CSNAPPY_ISA_PROF
				register uint32_t a65 asm("v65") = 0, a66 asm("v66") = 0, a67 asm("v67") = 0, a68 asm("v68") = 0;
				register uint32_t a69 asm("v69") = 0, a70 asm("v70") = 0, a71 asm("v71") = 0;
#define CSNAPPY_ISA_PROF_OUT , "+v"(a65), "+v"(a66), "+v"(a67), "+v"(a68), "+v"(a69), "+v"(a70), "+v"(a71)
#define CSNAPPY_ISA_PROF_CLOBBER "s96", "s97", "s98", "s99",
#else
#define CSNAPPY_ISA_PROF_OUT
#define CSNAPPY_ISA_PROF_CLOBBER
#endif
				asm volatile(CSNAPPY_ISA_LOOP(CSNAPPY_ISA_ENTRY_DENSE, CSNAPPY_ISA_TABLE_DENSE, CSNAPPY_ISA_GO_DENSE,
							      CSNAPPY_ISA_LOADS_DENSE, CSNAPPY_ISA_COMMIT_DENSE)
					     : [p0] "+s"(pz), [q1] "+s"(q1), [nemit] "+s"(nemit), [nev] "+s"(nev), "+v"(x0),
					       "+v"(x1), "+v"(x2), "+v"(x3), "+v"(vsid), "+v"(vpos), "+v"(px), "+v"(py), "+v"(poff)
					       CSNAPPY_ISA_PROF_OUT
					     : [src] "s"(src), [R] "s"(R), [ids] "s"(ids), [shm1] "s"(shift - 1), [mul] "s"(kHashMul),
					       [limit64] "s"(limit64), [safemax] "s"(n - 16), [safemax4] "s"(n - 4), [n] "s"(n), [lane] "v"(lane),
					       [thr] "v"(thr),
					       [norec] "v"(no_rec_off)
					     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v41", "v48", "v49", "v52", "v53",
					       "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "s60", "s61",
					       "s62", "s63", "s64", "s65", "s66",
					       "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
					       "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92",
					       "s93", "s94", CSNAPPY_ISA_PROF_CLOBBER "vcc", "scc", "memory");
#if CSNAPPY_ISA_PROF
				if (lane == 0) {
					atomicAdd(&g_isa_prof[0], (unsigned long long)a65);
					atomicAdd(&g_isa_prof[1], (unsigned long long)a66);
					atomicAdd(&g_isa_prof[2], (unsigned long long)a67);
					atomicAdd(&g_isa_prof[3], (unsigned long long)a68);
					atomicAdd(&g_isa_prof[4], (unsigned long long)a69);
					atomicAdd(&g_isa_prof[5], (unsigned long long)a70);
					atomicAdd(&g_isa_prof[6], 1ull);
					atomicAdd(&g_isa_prof[7], (unsigned long long)a71);
				}
#endif
				/* (the loads its last step requested are still on their way: clamped addresses, nobody
				 * wants them) */
				asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(vsid));
				next_emit = nemit;
				prec = make_uint2(px, py);
				prec_off = poff;
				fin = pz + 1 >= ip_limit;
				place();
				continue;
			}
			if constexpr (FAST_S) if (q1 <= 32 && pz < limit64 && epoch >= 2) {
				register uint32_t x0 asm("v44") = raw0, x1 asm("v45") = raw1, x2 asm("v46") = raw2, x3 asm("v47") = raw3;
				register uint32_t vsid asm("v42") = sid, vpos asm("v40") = pos;
				register uint32_t px asm("v50") = prec.x, py asm("v51") = prec.y, poff asm("v43") = prec_off;
				const uint32_t thr = lane == 0 ? 0u : 0x8000u;
				uint32_t nemit = next_emit;
				asm volatile(CSNAPPY_ISA_LOOP("", CSNAPPY_ISA_TABLE_SPILL, CSNAPPY_ISA_GO_SPILL, CSNAPPY_ISA_LOADS_SPILL,
							      CSNAPPY_ISA_COMMIT_SPILL)
					     : [p0] "+s"(pz), [q1] "+s"(q1), [nemit] "+s"(nemit), [nev] "+s"(nev), [epoch] "+s"(epoch),
					       "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(vsid), "+v"(vpos), "+v"(px), "+v"(py), "+v"(poff)
					     : [src] "s"(src), [R] "s"(R), [ids] "s"(ids), [spill] "s"(spill), [shm1] "s"(shift - 1),
					       [mul] "s"(kHashMul), [limit64] "s"(limit64), [safemax] "s"(n - 16), [n] "s"(n),
					       [dcap] "s"(dense_cap), [sbase] "s"(2 * dense_cap), [lane] "v"(lane), [thr] "v"(thr),
					       [norec] "v"(no_rec_off)
					     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v41", "v48", "v49", "v52", "v53",
					       "v54", "v55", "v56", "v57", "v58", "v59", "s60", "s61", "s62", "s63", "s64", "s65", "s66",
					       "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
					       "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92",
					       "s93", "s94", "s95", "vcc", "scc", "memory");
				asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(vsid));
				next_emit = nemit;
				prec = make_uint2(px, py);
				prec_off = poff;
				fin = pz + 1 >= ip_limit;
				place();
				continue;
			}
			if constexpr (FAST_H) if (q1 <= 32 && pz < limit64) {
				register uint32_t x0 asm("v44") = raw0, x1 asm("v45") = raw1, x2 asm("v46") = raw2, x3 asm("v47") = raw3;
				register uint32_t vpos asm("v40") = pos;
				register uint32_t px asm("v50") = prec.x, py asm("v51") = prec.y, poff asm("v43") = prec_off;
				const uint32_t thr = lane == 0 ? 0u : 0x8000u;
				uint32_t nemit = next_emit;
				asm volatile(CSNAPPY_ISA_LOOP("s_mov_b64 s[60:61], -1\n\t", CSNAPPY_ISA_TABLE_HASH, CSNAPPY_ISA_GO_DENSE,
							      CSNAPPY_ISA_LOADS_HASH, CSNAPPY_ISA_COMMIT_LDS)
					     : [p0] "+s"(pz), [q1] "+s"(q1), [nemit] "+s"(nemit), [nev] "+s"(nev), "+v"(x0), "+v"(x1),
					       "+v"(x2), "+v"(x3), "+v"(vpos), "+v"(px), "+v"(py), "+v"(poff)
					     : [src] "s"(src), [R] "s"(R), [shm1] "s"(shift - 1), [shift] "s"(shift), [mul] "s"(kHashMul),
					       [limit64] "s"(limit64), [safemax] "s"(n - 16), [n] "s"(n), [lane] "v"(lane), [thr] "v"(thr),
					       [norec] "v"(no_rec_off)
					     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v41", "v42", "v48", "v49", "v52",
					       "v53", "v54", "v55", "v56", "v57", "v58", "v59", "s60", "s61", "s62", "s63", "s64", "s65",
					       "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78",
					       "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91",
					       "vcc", "scc", "memory");
				asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
				next_emit = nemit;
				prec = make_uint2(px, py);
				prec_off = poff;
				fin = pz + 1 >= ip_limit;
				place();
				continue;
			}
			if constexpr (FAST_G) if (q1 <= 32 && pz < limit64 && epoch >= 2) {
				uint32_t w0;
				__builtin_memcpy(&w0, src + min((pz & ~3u) + 4 * lane, n - 4), 4); /* the step's window as dwords (CSNAPPY_ISA_OWN_BYTES) */
				register uint32_t x0 asm("v44") = w0;
				register uint32_t vpos asm("v40") = pos;
				register uint32_t px asm("v50") = prec.x, py asm("v51") = prec.y, poff asm("v43") = prec_off;
				const uint32_t thr = lane == 0 ? 0u : 0x8000u;
				uint32_t nemit = next_emit;
				asm volatile(CSNAPPY_ISA_LOOP("s_mov_b64 s[60:61], -1\n\t", CSNAPPY_ISA_TABLE_GTAB, CSNAPPY_ISA_GO_GTAB,
							      CSNAPPY_ISA_LOADS_GTAB, CSNAPPY_ISA_COMMIT_GTAB)
					     : [p0] "+s"(pz), [q1] "+s"(q1), [nemit] "+s"(nemit), [nev] "+s"(nev), [epoch] "+s"(epoch),
					       "+v"(x0), "+v"(vpos), "+v"(px), "+v"(py), "+v"(poff)
					     : [src] "s"(src), [R] "s"(R), [gtab] "s"(gtab), [shm1] "s"(shift - 1), [shift] "s"(shift),
					       [mul] "s"(kHashMul), [limit64] "s"(limit64), [safemax] "s"(n - 16), [safemax4] "s"(n - 4), [n] "s"(n),
					       [smask] "s"(smask), [sbase] "s"(A.lds0), [zv] "s"(chk0 << 15), [lane] "v"(lane), [thr] "v"(thr),
					       [norec] "v"(no_rec_off)
					     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v41", "v42", "v45", "v46", "v47", "v48", "v49", "v52",
					       "v53", "v54", "v55", "v56", "v57", "v58", "v59", "s60", "s61", "s62", "s63", "s64", "s65",
					       "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78",
					       "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91",
					       "s92", "s93", "vcc", "scc", "memory",
					       /* (v64 is not used: it makes this kernel declare more than 64 VGPRs.  A build that declared
					        * exactly 64, with the block using v60-v63, computed wrong match lengths whenever other
					        * workgroups shared its CU -- see "Registers are fixed" at CSNAPPY_ISA_LOOP; its SGPR count
					        * holds this kernel to seven waves a SIMD anyway, and seven of 72 parse G_low as fast) */
					       "v64");
				asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0));
				next_emit = nemit;
				prec = make_uint2(px, py);
				prec_off = poff;
				fin = pz + 1 >= ip_limit;
				place();
				continue;
			}
			tick(0); /* (rest of the previous step: commit) */
			if (PROF) {
				asm volatile("" : "+v"(raw0), "+v"(raw1), "+v"(raw2), "+v"(raw3), "+v"(sid));
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				pn_steps++;
			}
			tick(1); /* wait for the step's own bytes and ids */
			const bool sparse_c = q1 > 32;
			p0 = pz; /* dense: position of lane 0 */
			const uint32_t pos_c = pos;
			const bool valid_c = valid;
			me0 = raw0;
			me1 = raw1;
			me2 = raw2;
			me3 = raw3;
			const uint32_t prod = me0 * kHashMul;
			slot = DENSE ? sid : prod >> shift;
			const bool tabbed = DENSE ? valid_c && slot != kNoBucket : valid_c;
			const uint32_t chk = (prod >> (shift - 1)) & 1u;
			const uint32_t mine16 = pos_c | (chk << 15); /* my table entry, if I am inserted */
			const bool spilled = SPILL && tabbed && slot >= dense_cap;
			const bool in_lds = !GTAB && tabbed && !spilled;
			uint32_t raw16 = 0;
			cmask = 0; /* lanes that share their slot with a LOWER lane of the step ("flagged") */
			uint32_t bumped = 0;
			uint32_t key = 0;
			uint32_t xold = 0;
			if (!TW) {
				/* slot sharing inside a step, global table: ONE returning exchange on a small keyed
				 * array.  The LDS serves the lanes in ascending order, so a lane
				 * gets back the tag of the nearest LOWER lane with its key (or an older step's): of
				 * this step and my slot -> flagged, exactly; of this step and another slot -> my slot
				 * may hide behind it, flagged to be safe; of an older step -> no lower lane has my key.
				 * (Until round 5: two atomic minima, two reads and thirty instructions of tag
				 * arithmetic.)  The tag's epoch field is epoch - 1: never that of the ~0 fill. */
				key = slot & smask;
				if (ORD && tabbed)
					xold = atomicExch(&S[key], ((epoch - 1) << (7 + 15)) | (slot << 7) | lane);
				const bool written = tabbed && ((occ[slot >> 5] >> (slot & 31)) & 1u);
				cand = gtab[written ? slot : 0u];
				cand = written ? cand : 0u;
			} else {
				raw16 = tab[in_lds ? slot : 0u];
				if (!CHK0_INIT)
					raw16 = in_lds ? raw16 : 0u;
				if (ORD)
					bumped = atomicAdd(&tab32[in_lds ? slot >> 1 : 0u], in_lds ? 1u << ((slot & 1u) << 4) : 0u);
				cand = raw16;
			}
			if (SPILL && ballot64(spilled)) {
				/* the lanes whose bucket lies in HBM: one filter among themselves (a lane it does not
				 * settle is flagged to be safe; they are few, so that is rare) */
				key = (slot - dense_cap) & smask;
				if (ORD)
					atomicMin(&S[key], FT::tag(epoch, slot, lane, spilled));
				const uint32_t g = spill[spilled ? slot - dense_cap : 0u];
				cand = spilled ? g : cand;
			}
			wave_lds_fence();
			/* the candidate's 16 bytes are requested as soon as the table entry is there, in front of
			 * the read-back (dense steps; lanes without a candidate read position 0 -- one
			 * broadcast line; masking them off the load, here and in the spill-over gather, changes
			 * nothing: measured in round 3) */
			/* the candidate can match at all */
			const bool maybe = tabbed && (CHK0_INIT ? cand >> 15 : (cand ? cand >> 15 : chk0)) == chk;
			cand &= 0x7fffu;
			uint4 w4 = make_uint4(0, 0, 0, 0);
			tmask = 0;
			/* the lanes with a bucket, and those that share it with a lower lane (behind the gather's
			 * issue in either kind of step: the kinds are told apart ONCE per step) */
			auto step_flags = [&]() __attribute__((always_inline)) {
				tmask = ballot64(tabbed);
				if (!ORD) {
					cmask = tmask & ~1ull; /* (lane 0 has no lower lane) */
				} else if (!TW) {
					cmask = ballot64((xold >> (7 + 15)) == epoch - 1) & tmask;
				} else {
					/* my half of the dword as the add found it: the entry + the lower lanes of my slot
					 * (+ 1, rarely, for a carry out of the low half: a false alarm) */
					const uint32_t seen = (bumped >> ((slot & 1u) << 4)) & 0xffffu;
					cmask = ballot64(seen != raw16) & tmask;
					if (SPILL && ballot64(spilled)) {
						const uint32_t fe1 = S[key];
						cmask = (cmask & ~ballot64(spilled)) | ballot64(spilled & FT::flags(fe1, slot, lane));
					}
				}
				if (ORD && FILT && --epoch == 0) {
					/* the tags' epoch field is about to wrap: start over with empty filters */
					wave_lds_fence();
					uint4 *s4 = reinterpret_cast<uint4 *>(S);
					for (uint32_t k = lane; k < (s_entries >> 2); k += 64)
						s4[k] = make_uint4(~0u, ~0u, ~0u, ~0u);
					wave_lds_fence();
					epoch = FT::kEpochs;
				}
			};

			uint32_t e_final = 0;
			bool inside = false;
			uint2 rec = make_uint2(0, 0);
			bool rec_mine = false;
			uint32_t rec_idx = 0;

			if (sparse_c) {
				/* ---- sparse step: the lanes are the next probes of the stride rule; it ends at its
				 * first match, and in front of the first lane that shares a slot with an earlier one */
				step_flags();
				tick(2); /* filters + table */
				const uint64_t imask = ~ballot64(valid_c);
				const uint32_t c1 = cmask ? first_lane(cmask) : 64u;
				const uint32_t v = imask ? first_lane(imask) : 64u;
				const uint32_t ulim_s = min(c1, v);
				const bool gathered = lane < ulim_s && maybe;
				__builtin_memcpy(&w4, src + (gathered ? cand : 0u), 16);
				CSNAPPY_FLUSH_PREC();
				const uint64_t xlo = ((uint64_t)(me1 ^ w4.y) << 32) | (me0 ^ w4.x);
				const uint64_t xhi = ((uint64_t)(me3 ^ w4.w) << 32) | (me2 ^ w4.z);
				const uint32_t mlen_s = gathered ? common_prefix16(xlo, xhi) : 0u;
				const uint64_t matchmask = ballot64(lane < ulim_s && mlen_s >= 4);
				if (matchmask == 0) {
					e_final = ulim_s - 1; /* (ulim_s == 0 cannot happen: lane 0 is never flagged, and an invalid lane 0 ended the scan before) */
					if (ulim_s == v && v < 64)
						fin = true; /* next probe is past ip_limit: goto emit_remainder, :543-544 */
					else {
						q1 += ulim_s;
						pz += ulim_s;
					}
				} else {
					const uint32_t i = first_lane(matchmask);
					e_final = i;
					const uint32_t base = rdlane(pos_c, i), cnd = rdlane(cand, i);
					uint32_t L = rdlane(mlen_s, i);
					if (L == kLocalMatch && base + L < n)
						L += extend(cnd, base);
					if (lane == 0)
						R[nev] = pack_record(next_emit, base, cnd, L);
					++nev;
					const uint32_t ip = base + L;
					next_emit = ip;
					if (ip >= ip_limit)
						fin = true; /* :585-586 */
					pz = ip - 1;
					q1 = 0;
				}
				place();
			} else {
				/* ---- dense step: lane L holds position p0 + L; lane 0 is insert-only ---- */
				__builtin_memcpy(&w4, src + (maybe ? cand : 0u), 16);
				CSNAPPY_FLUSH_PREC();
				step_flags();
				tick(2); /* filters + table */
				ulim = min(64u, ip_limit - p0); /* lanes in front of the scan limit */
				/* (until round 5 the id lines two steps ahead were touched here, behind the gather: 9.00
				 * against 9.14 ms per GiB of text in round 3; with a step a fifth shorter the four
				 * instructions cost more than the touch saves: 9.30 against 9.48.  For small fragments
				 * only -- pages gain 2 % from it -- the test costs text 1.4 %: not kept either.) */
				const uint64_t xlo = ((uint64_t)(me1 ^ w4.y) << 32) | (me0 ^ w4.x);
				const uint64_t xhi = ((uint64_t)(me3 ^ w4.w) << 32) | (me2 ^ w4.z);
				if (PROF) {
					asm volatile("" : "+v"(w4.x), "+v"(w4.y), "+v"(w4.z), "+v"(w4.w));
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				}
				tick(3); /* candidate gather */
				if (PROF) {
					pn_tabbed += __builtin_popcountll(tmask);
					pn_gathered += __builtin_popcountll(ballot64(maybe));
					pn_flagged += __builtin_popcountll(cmask);
				}
				/* (computed by every lane, then selected: as a conditional the compiler wraps it in an
				 * exec-mask region -- three scalar instructions and a branch.  Lane 0 is insert-only.) */
				uint32_t cp16 = common_prefix16(xlo, xhi);
				asm volatile("" : "+v"(cp16));
				mlen = (maybe && lane != 0) ? cp16 : 0u;
				const uint64_t matchmask = ballot64(mlen >= 4);
				if (PROF)
					pn_match4 += __builtin_popcountll(matchmask);
				/* flagged lanes are stops of the chain like matches: what they hold is decided
				 * when (and if) the chain gets there */
				stopmask = matchmask | cmask;
				special = ballot64(mlen == kLocalMatch && p0 + lane + kLocalMatch < n) | cmask;
				cl = lane + mlen; /* lane of the re-match probe after my match */
				/* c = lane behind a copy: the next stop of the chain.  64: the re-match probe falls
				 * outside the usable lanes; 65: none of the 33 probes behind the copy is a stop;
				 * lane | 128: that stop is a special lane */
				/* Lane 0 (insert-only, never a stop itself) holds the chain's FIRST stop: lanes 1 .. lim0
				 * are what is left of the current scan's stride-1 probes, so it is the same search
				 * with cl = 0 and that many probes (the scalar unit did this until round 5: a dozen
				 * instructions of its own per step) */
				lim0 = 33 - q1;
				{
					const uint64_t rest = stopmask >> (cl & 63u);
					uint32_t fm = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
					asm volatile("" : "+v"(fm));
					const uint32_t j = cl + fm;
					const bool in = fm <= (lane == 0 ? lim0 : 32u) && j <= 63;
					const uint32_t sp = (uint32_t)(special >> (j & 63u)) & 1u;
					uint32_t inner = in ? j | (sp << 7) : 65u;
					asm volatile("" : "+v"(inner)); /* (a select below, not an exec-mask region around the above) */
					nx = cl >= ulim ? 64u : inner;
				}
				t = 0; /* (the walk starts AT lane 0, whose entry is the first stop) */
				taken = 0; /* lanes whose match is part of the chain */
				tick(4); /* match lengths, next-stop table */
				/* plain matches: hop from match to match (unrolling this loop four times was
				 * measured in round 3: no gain; the walk of a step without special lanes -- four
				 * steps in five -- is this loop alone) */
				CSNAPPY_HOPS();
				taken &= ~1ull; /* (lane 0 is where the walk starts, not a match) */
				visits();
				/* ---- where the chain left the step (selects, no branches: this is scalar code) ----
				 * t == 64: the last copy ends at or behind the usable lanes: re-match probe next
				 * (:585-594).  t == 65: the current window (the 33 probes behind the last copy, or what
				 * was left of the scan the step started in) has no match: the scan goes on behind its
				 * last probed lane e, or runs into the scan limit (goto emit_remainder, :543-544). */
				tick(5); /* chain walk */
				if (PROF)
					pn_hops += __builtin_popcountll(taken);
				const uint32_t emit0 = next_emit, nev0 = nev;
				const uint32_t last = 63u - (uint32_t)__builtin_clzll(taken | 1);
				const uint32_t c = rdlane(cl, last); /* where the last copy ends (no copy: lane 0's, unused) */
				/* Scalar selects on two conditions -- was a copy taken at all; did the last one leave the
				 * usable lanes (t == 64) -- one compare each (as C the compiler makes them branches, or
				 * three mask operations a select).  The next step's lane 0: the position in front of the
				 * copy's end (its ip - 1 insert), else the last lane e this step probed; either way the
				 * step after finds nothing to probe exactly when that position is the last one in front
				 * of the scan limit (:543-544, :585-586). */
				uint32_t lim, qq, ne;
				asm("s_cmp_lg_u64 %3, 0\n\t"
				    "s_cselect_b32 %0, %4, %5\n\t"
				    "s_cselect_b32 %1, %6, %7\n\t"
				    "s_cselect_b32 %2, %8, %9"
				    : "=&s"(lim), "=&s"(qq), "=&s"(ne)
				    : "s"(taken), "s"(c + 32), "s"(lim0), "s"(1 - c), "s"(q1), "s"(p0 + c), "s"(next_emit)
				    : "scc");
				const uint32_t e = min(lim, ulim - 1);
				uint32_t adv, q1n;
				asm("s_cmp_eq_u32 %3, 64\n\t"
				    "s_cselect_b32 %0, %4, %5\n\t"
				    "s_cselect_b32 %1, 0, %6\n\t"
				    "s_cselect_b32 %2, %7, %5"
				    : "=&s"(adv), "=&s"(q1n), "=&s"(e_final)
				    : "s"(t), "s"(c - 1), "s"(e), "s"(qq + e), "s"(last)
				    : "scc");
				next_emit = ne;
				q1 = q1n;
				pz = p0 + adv;
				fin = pz + 1 >= ip_limit;
				/* the cursor of the next step is known: fetch its bytes now (after the last step the
				 * loads are harmless: an invalid lane reads position 0) */
				place();
				tick(6); /* cursor update, next step's loads issued */
				/* ---- records of the taken matches, built by their own lanes ---- */
				if (taken) {
					const uint64_t below = taken & lt_mask;
					rec_mine = (taken >> lane) & 1;
					/* cprev = where the nearest taken copy below me ends (0: none).  The ends grow
					 * along the chain, so this is a running maximum over the lanes below: DPP row
					 * shifts, no LDS round trip (a bpermute from the nearest taken lane did it before) */
					const uint32_t cprev = wave_shr1(wave_incl_max_dpp(rec_mine ? cl : 0u));
					const bool hasprev = cprev != 0;
					const uint32_t lit_start = hasprev ? p0 + cprev : emit0;
					inside = hasprev && lane + 1 < cprev; /* strictly inside a taken copy: never inserted */
					rec_idx = nev0 + (uint32_t)__builtin_popcountll(below);
					rec = pack_record(lit_start, p0 + lane, cand, mlen);
					nev = nev0 + (uint32_t)__builtin_popcountll(taken);
				}
			}
			/* The next step's own bytes (requested by place() above) are waited for HERE, in front
			 * of this step's stores: gfx9 counts loads and stores in one vmcnt, so a wait placed
			 * behind the stores would also sit out the stores' round trip. */
			tick(7); /* records built */
			asm volatile("" : "+v"(raw0), "+v"(raw1), "+v"(raw2), "+v"(raw3), "+v"(sid));
			if (PROF)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			tick(8); /* wait for the next step's bytes (in front of this step's stores) */
			if (PROF && sparse_c)
				pn_sparse++;
			prec = rec;
			prec_off = rec_mine ? rec_idx * 8 : no_rec_off;
			/* commit table[slot] = position for every lane that was probed or inserted
			 * (:550, :589, :593): lanes 0..e_final except those inside a copy */
			bool commit = lane <= e_final && !inside && tabbed;
			{
				/* of several committed lanes with one slot only the last may write */
				const uint64_t cm = ballot64(commit);
				uint64_t fl = cmask & cm;
				if (TW && ORD) /* (the LDS keeps the highest lane of a slot by itself; only stores to HBM need this) */
					fl = SPILL ? fl & ballot64(spilled) : 0;
				if (fl) {
					/* one round per SLOT that several committed lanes share, highest lane first: it
					 * keeps its write, the lower ones of its slot lose theirs (runs put one slot on
					 * most of the 64 lanes: one round, not one per lane) */
					uint64_t dead = 0;
					do {
						const uint32_t x = 63u - (uint32_t)__builtin_clzll(fl);
						const uint64_t same = ballot64(slot == rdlane(slot, x)) & cm;
						dead |= same & ((1ull << x) - 1);
						fl &= ~same;
					} while (fl);
					if ((dead >> lane) & 1)
						commit = false;
				}
			}
			if (GTAB) {
				if (commit) {
					gtab[slot] = (uint16_t)mine16;
					atomicOr(&occ[slot >> 5], 1u << (slot & 31));
				}
			} else {
				/* TW: the adds are taken back (a subtraction each: whatever carried is undone with it);
				 * then the inserted lanes write their entries -- of several with one slot the LDS keeps
				 * the highest (ascending lane order) */
				if (ORD) {
					(void)atomicSub(&tab32[in_lds ? slot >> 1 : 0u], in_lds ? 1u << ((slot & 1u) << 4) : 0u);
					wave_lds_fence();
				}
				if (in_lds && commit)
					tab[slot] = (uint16_t)mine16;
			}
			if (!CSNAPPY_PARSE_NOSPILLSTORE && SPILL && commit && spilled) /* (macro: timing experiment only) */
				spill[slot - dense_cap] = (uint16_t)mine16;
			wave_lds_fence();
		}
#undef CSNAPPY_HOPS
		CSNAPPY_FLUSH_PREC();
#undef CSNAPPY_FLUSH_PREC
		stuck = !fin;
	}

	/* emit_remainder, csnappy_compress.c:600-605: literal [next_emit, n), no copy */
	if (next_emit < n) {
		if (lane == 0)
			R[nev] = pack_record(next_emit, n, n, 0);
		++nev;
	}
	if (lane == 0)
		A.rec_cnt[F.c] = stuck ? kNoRecords : nev; /* (never parsed: the emit kernel reports the block as failed) */
	if (PROF && lane == 0) {
		const unsigned long long t_end = __builtin_amdgcn_s_memtime();
		atomicAdd(&A.prof[0], t_end - pt_begin);
		for (int k = 0; k < 9; ++k)
			atomicAdd(&A.prof[1 + k], pt[k]);
		atomicAdd(&A.prof[10], pn_steps);
		atomicAdd(&A.prof[11], pn_hops);
		atomicAdd(&A.prof[12], pn_special);
		atomicAdd(&A.prof[13], pn_sparse);
		atomicAdd(&A.prof[14], 1ull);
		atomicAdd(&A.prof[16], pn_tabbed);
		atomicAdd(&A.prof[17], pn_gathered);
		atomicAdd(&A.prof[18], pn_match4);
		atomicAdd(&A.prof[19], pn_flagged);
	}
}

/* table indexed by dense bucket ids, in LDS: prologue, then the parser with or without the
 * spill-over's selects */
extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_dense_lean(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, false))
		return;
	const uint32_t nb = dense_prologue(A, F);
	if (nb == kNoRecords)
		return;
	if (nb > A.dense_cap)
		parse_lean<TAB_LDS_DENSE, true>(A, F);
	else
		parse_lean<TAB_LDS_DENSE, false>(A, F);
}

/* table indexed by the hash, in LDS (tables of <= 8 KiB) */
extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_hash_lean(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, false))
		return;
	parse_lean<TAB_LDS_HASH, false>(A, F);
}

/* full table in global memory: fragments with more buckets than the dense table and its spill-over
 * hold, and those the prologue's sample sent here (repetitive data: few steps, no prologue) */
extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_gtab(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, true))
		return;
	parse_lean<TAB_GLOBAL, false>(A, F);
}

/* The same three for a device whose LDS does not serve one instruction's lanes in ascending order (or
 * CSNAPPY_HIP_NO_LDS_ORDER=1): parse_lean<.., ORD = false>, exact on any hardware, several times slower. */
extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_dense_lean_unordered(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, false))
		return;
	const uint32_t nb = dense_prologue(A, F);
	if (nb == kNoRecords)
		return;
	if (nb > A.dense_cap)
		parse_lean<TAB_LDS_DENSE, true, false, false>(A, F);
	else
		parse_lean<TAB_LDS_DENSE, false, false, false>(A, F);
}

extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_hash_lean_unordered(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, false))
		return;
	parse_lean<TAB_LDS_HASH, false, false, false>(A, F);
}

extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_gtab_unordered(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, true))
		return;
	parse_lean<TAB_GLOBAL, false, false, false>(A, F);
}

/* debug: the dense kernel with s_memtime phase counters (csnappy_hip_debug_set_profile_buffer) */
extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_dense_lean_prof(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, false))
		return;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	const uint32_t nb = dense_prologue(A, F);
	if (nb == kNoRecords)
		return;
	if (threadIdx.x == 0)
		atomicAdd(&A.prof[15], __builtin_amdgcn_s_memtime() - t0);
	if (nb > A.dense_cap)
		parse_lean<TAB_LDS_DENSE, true, true>(A, F);
	else
		parse_lean<TAB_LDS_DENSE, false, true>(A, F);
}

/* debug: the global-table kernel with the same counters (G_low takes this kernel) */
extern "C" __global__ void __launch_bounds__(64, 5) snappy_parse_fragments_gtab_prof(CompressArgs A)
{
	Frag F;
	if (!frag_setup(A, F, true))
		return;
	parse_lean<TAB_GLOBAL, false, true>(A, F);
}

/* ==========================================================================================
 * COMPRESS, part 2 of 2: the emit launches (snappy_emit_sizes / _bases / _blocks; _pages)
 *
 * Turn the parser's records into the block's bytes: EmitLiteral / EmitCopy
 * (csnappy_compress.c:332-415), the varint length prefix (:46-73) and the pointer chain that
 * puts fragment k+1 behind fragment k (:647-653).  The encoded size of every record (one pass,
 * 64 records per wave), a scan of the per-64-record totals and of the fragments' totals, then
 * every wave encodes its 64-record chunks into its own LDS staging and flushes them with aligned
 * 16 B/lane stores at the chunk's final place in the block's slot -- nothing is moved twice.
 * ======================================================================================== */
#ifndef CSNAPPY_EMIT_DECODE_ONCE
#define CSNAPPY_EMIT_DECODE_ONCE 1 /* 0: emit_chunk decodes its records again (A/B: +3 % emit time) */
#endif
struct RecFields {
	uint32_t lit_start, lit_len, coff, clen, lhdr, mine;
	CopyPlan cp;
};

DEVINL RecFields decode_record(uint2 r, bool live)
{
	RecFields f;
	f.lit_start = r.y >> 16;
	f.lit_len = live ? (r.x & 0xffffu) - f.lit_start : 0; /* literal [lit_start, base) */
	f.coff = (r.x & 0xffffu) - (r.x >> 16);              /* base - candidate */
	f.clen = live ? r.y & 0xffffu : 0;
	f.lhdr = f.lit_len == 0 ? 0 : f.lit_len <= 60 ? 1 : f.lit_len <= 256 ? 2 : 3;
	f.cp = plan_copy(f.clen, f.coff);
	f.mine = f.lhdr + f.lit_len + f.cp.bytes;
	return f;
}

/* encode records [first, first + nev) of one fragment to dst (their final place); src = the
 * fragment's input, avail = input bytes readable from src (to the end of the block) */
/* what emit_chunk needs from memory for one chunk, fetched one chunk ahead of its use */
struct ChunkIn {
	uint2 r;     /* the lane's record */
	uint4 la, lb; /* 32 bytes at the record's literal (small records only) */
	uint32_t mw; /* dword `lane` of the literal of the chunk's first record with a literal of up to kMediumLiteral bytes
		      * that is not small (round 6: fetched with the chunk, not when its turn comes) */
#if CSNAPPY_EMIT_DECODE_ONCE
	RecFields f; /* the record decoded (fetch_literal needs it for the literal's address: emit_chunk takes it from here) */
#endif
};

DEVINL bool record_is_small(const RecFields &f, bool live, uint32_t avail)
{
	/* small records are staged in LDS, their literal (< 32 bytes) fetched with two 16-byte loads;
	 * a record is "big" when it encodes to more than kBigRecord bytes -- or when those 32 bytes
	 * would reach past the end of the block's input: big records go straight to HBM */
	return live && f.mine <= kBigRecord && (f.lit_len == 0 || f.lit_start + 32 <= avail);
}

/* The loads of fetch_record and fetch_literal are issued by every lane, whatever the chunk holds (a lane
 * without a record or without a literal reads the fragment's first records instead): gfx9 counts loads in
 * order in one counter, and only when the NUMBER of loads between a chunk's own and its first use is the
 * same on every path can the wait in front of that use leave the next chunk's loads in flight.  With a
 * branch around any of them the compiler has to wait for all of them -- vmcnt(0), once per chunk, a memory
 * round trip that nothing covered (rounds 2 to 5). */
DEVINL uint2 fetch_record(const uint2 *R, uint32_t first, uint32_t nev, uint32_t lane)
{
	const uint2 r = R[first + min(lane, nev - 1)]; /* (nev >= 1) */
	return lane < nev ? r : make_uint2(0, 0);
}

/* the record whose literal travels in ChunkIn::mw: not small, a literal of 1..kMediumLiteral bytes */
DEVINL bool record_has_wide_literal(const RecFields &f, bool live, uint32_t avail)
{
	return live && !record_is_small(f, live, avail) && f.lit_len && f.lit_len <= kMediumLiteral;
}

DEVINL void fetch_literal(ChunkIn &c, uint32_t nev, const uint8_t *src, uint32_t avail, uint32_t lane, const uint2 *R)
{
	const bool live = lane < nev;
	const RecFields f = decode_record(c.r, live);
#if CSNAPPY_EMIT_DECODE_ONCE
	c.f = f;
#endif
	/* (what a lane without a literal loads is never looked at: see `left` and the medium loop of emit_chunk) */
	const uint8_t *scrap = reinterpret_cast<const uint8_t *>(R);
	const uint8_t *pl = record_is_small(f, live, avail) && f.lit_len ? src + f.lit_start : scrap;
	__builtin_memcpy(&c.la, pl, 16);
	__builtin_memcpy(&c.lb, pl + 16, 16);
	/* A literal of 32..256 bytes (text: one record in 150, one chunk in three) used to be fetched where it
	 * is staged: a memory round trip in the middle of the chunk.  Its bytes are one dword per lane: requested
	 * here, a chunk ahead, with the others'. */
	const uint64_t wide = ballot64(record_has_wide_literal(f, live, avail));
	const uint32_t m = wide ? first_lane(wide) : 0u;
	const uint32_t ls = rdlane(f.lit_start, m), ll = wide ? rdlane(f.lit_len, m) : 0u;
	const uint8_t *pw = 4 * lane < ll && ls + 4 * lane + 4 <= avail ? src + ls + 4 * lane : scrap;
	__builtin_memcpy(&c.mw, pw, 4);
}

/* `ahead` = the loads already issued for the wave's next chunk: they are waited for in front of
 * this chunk's stores (gfx9 has one counter for loads and stores; a wait behind the stores would
 * also sit out their round trip) */
#ifndef CSNAPPY_EMIT_NOBIG
#define CSNAPPY_EMIT_NOBIG 0
#endif
#ifndef CSNAPPY_EMIT_EXP
#define CSNAPPY_EMIT_EXP 0
#endif
#ifndef CSNAPPY_EMIT_PROF
#define CSNAPPY_EMIT_PROF 0
#endif
#if CSNAPPY_EMIT_PROF
/* development builds only (tools/build_variant.sh <name> -DCSNAPPY_EMIT_PROF=1, tools/phase_emit.py) */
__device__ unsigned long long g_emit_prof[16];
#define EMIT_TICK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ept[k] += now_ - ept_last; ept_last = now_; } while (0)
#define EMIT_PROF_ARG , unsigned long long *ept
#else
#define EMIT_TICK(k) do { } while (0)
#define EMIT_PROF_ARG
#endif
/* what a wave's staging holds between chunks: consecutive chunks of a wave are consecutive in the
 * output, so the staging is drained when the next chunk would not fit it (and behind the wave's
 * last chunk), not behind every chunk */
struct EmitState {
	uint32_t gpos; /* output bytes already in HBM (relative to dst) */
	uint32_t fill; /* staged bytes */
};

template <class Prefetch>
DEVINL uint32_t emit_chunk(const ChunkIn &in, Prefetch &&prefetch, uint32_t nev, const uint8_t *src, uint32_t avail,
			   uint8_t *dst, uint8_t *stage, uint32_t lane, EmitState &st, bool last_chunk EMIT_PROF_ARG)
{
#if CSNAPPY_EMIT_PROF
	unsigned long long ept_last = __builtin_amdgcn_s_memtime();
	ept[7] += 1;
#endif
	/* staged byte t (t < fill) is output byte gpos + t and sits at stage[sa + t], where
	 * sa = (dst + gpos) & 15, so LDS 16 B chunks line up with global 16 B chunks. */
	uint32_t gpos = st.gpos, fill = st.fill;
	uint32_t sa = (uint32_t)(reinterpret_cast<uintptr_t>(dst + gpos) & 15u);

	auto drain = [&]() {
		wave_lds_fence();
		/* (loads before stores: gfx9 has one counter for both, and a wait behind the stores would also sit out
		 * their round trip.  The wait itself, not a use of the loaded registers that makes the compiler wait: those
		 * would become values of the loop around the drain, copied at its head behind a wait for all of them) */
		asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
		uint8_t *gbase = dst + gpos - sa; /* 16 B aligned */
		const uint32_t end = sa + fill;
		uint32_t first_full = 0;
		if (sa > 0) {
			const uint32_t hend = min(16u, end);
			if (lane >= sa && lane < hend)
				gbase[lane] = stage[lane];
			first_full = 1;
		}
		const uint32_t nfull = end >> 4;
		for (uint32_t cc = first_full + lane; cc < nfull; cc += 64)
			reinterpret_cast<uint4 *>(gbase)[cc] = reinterpret_cast<const uint4 *>(stage)[cc];
		const uint32_t tail0 = nfull << 4;
		const uint32_t tail = (end > tail0 && (nfull >= 1 || sa == 0)) ? end - tail0 : 0;
		if (lane < tail)
			gbase[tail0 + lane] = stage[tail0 + lane];
		gpos += fill;
		fill = 0;
		sa = (uint32_t)(reinterpret_cast<uintptr_t>(dst + gpos) & 15u);
		wave_lds_fence();
	};

	const bool live = lane < nev;
#if CSNAPPY_EMIT_DECODE_ONCE
	const RecFields f = in.f;
#else
	const RecFields f = decode_record(in.r, live);
#endif
	const uint32_t lit_start = f.lit_start, lit_len = f.lit_len, coff = f.coff, clen = f.clen;
	const uint32_t lhdr = f.lhdr, mine = f.mine;
	const CopyPlan cp = f.cp;
	const bool small = record_is_small(f, live, avail);
	uint32_t total;
	const uint32_t excl = wave_excl_scan(mine, lane, &total);
	/* A record with a literal of 32..kMediumLiteral bytes (text: one in 150, but it used to split
	 * its chunk in two stagings and two drains) is staged like a small one when the whole chunk
	 * fits the staging: its header and tags by its lane, its literal's bytes by the wave. */
	const bool medium = live && !small && total <= kStageCap && lit_len <= kMediumLiteral && cp.bytes <= 16;
	const uint64_t medmask = ballot64(medium);
	uint64_t bigmask = ballot64(live && !small && !medium);
	const uint32_t lw[8] = { in.la.x, in.la.y, in.la.z, in.la.w, in.lb.x, in.lb.y, in.lb.z, in.lb.w };
	uint32_t seg_lo = 0; /* first record of the current run of small records */
#if CSNAPPY_EMIT_PROF
	asm volatile("" : : "v"(excl), "v"(lw[0]), "v"(lw[7]));
#endif
	EMIT_TICK(0); /* decode, offsets (and the wait for the chunk's own loads) */
	if (fill && fill + total > kStageCap)
		drain(); /* (a chunk with a big record may exceed the staging by itself: its runs do not) */
	/* The next chunk's loads go out HERE: behind this chunk's own (a chunk old, waited for at its top) and behind the
	 * drain's stores, so that nothing in this chunk waits for memory -- the next chunk's top does, a chunk later,
	 * when loads and stores are long back.  (Issued in front of the drain they were what its stores waited for,
	 * a round trip every third chunk; issued before this chunk's top they made its wait theirs.) */
	prefetch();
	while (nev) {
		const uint32_t seg_hi = bigmask ? first_lane(bigmask) : nev; /* one past the run */
		if (seg_hi > seg_lo) {
			/* stage small records [seg_lo, seg_hi) */
			/* (opaque copies: the loop runs once unless the chunk has a big record, and the
			 * compiler otherwise evaluates every predicate of this block -- a dozen 64-bit lane
			 * masks -- in front of the loop and spills them to a VGPR's lanes) */
			uint32_t lit_len_ = lit_len, lhdr_ = lhdr, clen_ = clen, coff_ = coff;
			uint32_t k64_ = cp.k64, k60_ = cp.k60, last_ = cp.last;
			asm volatile("" : "+v"(lit_len_), "+v"(lhdr_), "+v"(clen_), "+v"(coff_), "+v"(k64_), "+v"(k60_), "+v"(last_));
			const uint32_t run_base = rdlane(excl, seg_lo);
			const uint32_t run_bytes = (seg_hi < 64 ? rdlane(excl, seg_hi & 63) : total) - run_base;
			const bool in_run = lane >= seg_lo && lane < seg_hi;
			uint8_t *o = stage + sa + fill + (excl - run_base);
			/* literal payload first, as whole dwords (unaligned LDS stores): the up to three
			 * bytes a lane writes past its literal land on its own copy tag and, at most, on the
			 * first byte of the next record -- always a tag byte -- and both are written below */
			{
				/* (left = the lane's payload bytes still to store; kept opaque to the compiler, which
				 * otherwise computes the eight rounds' lane masks up front and spills them) */
				uint32_t left = in_run && small ? lit_len_ : 0;
#if CSNAPPY_EMIT_EXP & 4 /* timing experiment only (wrong output): no payload stores */
				left = 0;
#endif
#pragma unroll
				for (uint32_t k = 0; k < 8; ++k) {
					asm volatile("" : "+v"(left));
					if (!ballot64(left > 4 * k))
						break;
#if CSNAPPY_EMIT_EXP & 2 /* timing experiment only (wrong output): aligned dword stores */
					if (left > 4 * k)
						*reinterpret_cast<uint32_t *>(reinterpret_cast<uintptr_t>(o + lhdr_ + 4 * k) & ~(uintptr_t)3) = lw[k];
#else
					if (left > 4 * k)
						__builtin_memcpy(o + lhdr_ + 4 * k, &lw[k], 4);
#endif
				}
			}
			wave_lds_fence();
			{
				const uint64_t run_mask = (seg_hi < 64 ? (1ull << seg_hi) - 1 : ~0ull) & ~((1ull << seg_lo) - 1);
				uint64_t mm = medmask & run_mask;
#if CSNAPPY_EMIT_EXP & 1 /* timing experiment only (wrong output): no medium literals */
				mm = 0;
#endif
				if (mm) {
					/* (the literal that came with the chunk: its first record with such a literal, whole dwords inside the input) */
					const uint64_t wide = ballot64(record_has_wide_literal(f, live, avail));
					const uint32_t wm = wide ? first_lane(wide) : 64u;
					do {
						const uint32_t m = first_lane(mm);
						mm &= mm - 1;
						const uint32_t ls = rdlane(lit_start, m), ll = rdlane(lit_len_, m);
						uint8_t *pd = stage + sa + fill + (rdlane(excl, m) - run_base) + rdlane(lhdr_, m);
						/* (whole dwords, like the small records' payload above: the up to three bytes past the
						 * literal land on the record's own copy tag and the next record's first byte, a header
						 * or a tag, all written below -- so only for a record that has a copy) */
						if (m == wm && ls + ((ll + 3) & ~3u) <= avail && rdlane(clen_, m)) {
							if (4 * lane < ll)
								__builtin_memcpy(pd + 4 * lane, &in.mw, 4);
							continue;
						}
						for (uint32_t j = 4 * lane; j < ll; j += 256) {
							if (j + 4 <= ll) {
								uint32_t w;
								__builtin_memcpy(&w, src + ls + j, 4);
								__builtin_memcpy(pd + j, &w, 4);
							} else {
								for (uint32_t t = j; t < ll; ++t)
									pd[t] = src[ls + t];
							}
						}
					} while (mm);
					wave_lds_fence();
				}
			}
			EMIT_TICK(1); /* literal payload into the staging */
			if (in_run) {
				if (lhdr_ == 1) {
					o[0] = (uint8_t)((lit_len_ - 1) << 2);
				} else if (lhdr_ == 2) {
					o[0] = (uint8_t)(60 << 2);
					o[1] = (uint8_t)(lit_len_ - 1);
				}
			}
			if (in_run && clen_) {
				uint8_t *q = o + lhdr_ + lit_len_;
				const uint8_t lo = (uint8_t)(coff_ & 0xff), hi = (uint8_t)(coff_ >> 8);
				for (uint32_t k = 0; k < k64_; ++k) {
					q[0] = 0xfe; /* COPY_2 | (63 << 2) */
					q[1] = lo;
					q[2] = hi;
					q += 3;
				}
				if (k60_) {
					q[0] = 0xee; /* COPY_2 | (59 << 2) */
					q[1] = lo;
					q[2] = hi;
					q += 3;
				}
				if (last_ < 12 && coff_ < 2048) {
					q[0] = (uint8_t)(1 + ((last_ - 4) << 2) + ((coff_ >> 8) << 5));
					q[1] = lo;
				} else {
					q[0] = (uint8_t)(2 + ((last_ - 1) << 2));
					q[1] = lo;
					q[2] = hi;
				}
			}
			fill += run_bytes;
			EMIT_TICK(2); /* literal headers, copy tags */
		}
#if CSNAPPY_EMIT_NOBIG
		break; /* (timing experiment only: wrong output) */
#endif
		if (!bigmask)
			break;
		/* ---- a big record: drain the staging, write it straight to HBM ----
		 * Its literal's first 256 + 6 bytes are requested BEFORE the drain and waited for in front
		 * of the drain's stores: gfx9 counts loads and stores in one counter, and a load issued
		 * behind a store is not back before the store is acknowledged -- four such waits per big
		 * record (head, words, tail behind the drain's and each other's stores) were more than
		 * half of this kernel's time on text, at 0.4 big records per 64. */
		const uint32_t e = first_lane(bigmask);
		bigmask &= bigmask - 1;
		const uint32_t ls = rdlane(lit_start, e), ll = rdlane(lit_len, e);
		const uint32_t lh = rdlane(lhdr, e), co = rdlane(coff, e), cl = rdlane(clen, e);
		const uint32_t k64 = rdlane(cp.k64, e), k60 = rdlane(cp.k60, e), last = rdlane(cp.last, e);
		const uint32_t cbytes = rdlane(cp.bytes, e);
		uint8_t *o = dst + gpos + fill; /* (where the drain will leave the cursor) */
		uint8_t *d = o + lh;
		/* literal payload: destination-aligned dwords from the input */
		const uint32_t head = min(ll, (uint32_t)((4 - (reinterpret_cast<uintptr_t>(d) & 3)) & 3));
		const uint32_t words = (ll - head) >> 2;
		const uint32_t t0 = head + 4 * words;
		uint32_t hb = 0, w0 = 0, tb = 0;
		if (lane < head)
			hb = src[ls + lane];
		if (lane < words)
			__builtin_memcpy(&w0, src + ls + head + 4 * lane, 4);
		if (t0 + lane < ll)
			tb = src[ls + t0 + lane];
		asm volatile("" : "+v"(hb), "+v"(w0), "+v"(tb));
		drain();
		if (lane == 0) {
			const uint32_t v = ll - 1;
			if (lh == 1) {
				o[0] = (uint8_t)(v << 2);
			} else if (lh == 2) {
				o[0] = (uint8_t)(60 << 2);
				o[1] = (uint8_t)v;
			} else if (lh == 3) {
				o[0] = (uint8_t)(61 << 2);
				o[1] = (uint8_t)(v & 0xff);
				o[2] = (uint8_t)(v >> 8);
			}
		}
		{
			uint32_t *d32 = reinterpret_cast<uint32_t *>(d + head);
			if (lane < head)
				d[lane] = (uint8_t)hb;
			if (lane < words)
				d32[lane] = w0;
			if (t0 + lane < ll)
				d[t0 + lane] = (uint8_t)tb;
			/* longer literals: 1 KiB per round trip */
			for (uint32_t k = 64 + lane; k < words; k += 256) {
				const uint32_t lastw = words - 1;
				uint32_t v0, v1, v2, v3;
				__builtin_memcpy(&v0, src + ls + head + 4 * min(k, lastw), 4);
				__builtin_memcpy(&v1, src + ls + head + 4 * min(k + 64u, lastw), 4);
				__builtin_memcpy(&v2, src + ls + head + 4 * min(k + 128u, lastw), 4);
				__builtin_memcpy(&v3, src + ls + head + 4 * min(k + 192u, lastw), 4);
				d32[k] = v0;
				if (k + 64u < words)
					d32[k + 64u] = v1;
				if (k + 128u < words)
					d32[k + 128u] = v2;
				if (k + 192u < words)
					d32[k + 192u] = v3;
			}
		}
		if (cl) {
			uint8_t *q = o + lh + ll;
			const uint32_t body = 3 * (k64 + k60);
			for (uint32_t t = lane; t < cbytes; t += 64) {
				uint8_t bb;
				if (t < body) {
					const uint32_t rr = t % 3;
					bb = rr == 0 ? (t < 3 * k64 ? 0xfe : 0xee) : rr == 1 ? (uint8_t)(co & 0xff) : (uint8_t)(co >> 8);
				} else {
					const uint32_t rr = t - body;
					const bool two = last < 12 && co < 2048;
					bb = rr == 0 ? (two ? (uint8_t)(1 + ((last - 4) << 2) + ((co >> 8) << 5))
							    : (uint8_t)(2 + ((last - 1) << 2)))
					     : rr == 1 ? (uint8_t)(co & 0xff) : (uint8_t)(co >> 8);
				}
				q[t] = bb;
			}
		}
		gpos += lh + ll + cbytes;
		sa = (uint32_t)(reinterpret_cast<uintptr_t>(dst + gpos) & 15u);
		seg_lo = e + 1;
		EMIT_TICK(3); /* a big record */
	}
	if (last_chunk)
		drain();
	EMIT_TICK(4); /* drain */
	st.gpos = gpos;
	st.fill = fill;
	return total;
}

/* waves per workgroup of the emit kernels.  One since round 5: every wave of a fragment's workgroup
 * takes a run of consecutive chunks and keeps its own staging, so the waves never needed each other --
 * and one wave per workgroup leaves the scheduler the finest grain (1 / 2 / 3 / 4 / 8 waves: 1.15 / 1.18 /
 * 1.24 / 1.30 / 1.68 ms per GiB of text, pages 1.23 / 1.28 / 1.35 / 1.52 / 1.78, G_low 0.22 / 0.23 / 0.27 / 0.31) */
constexpr uint32_t kEmitWaves = 1;
constexpr uint32_t kMaxChunks = (kFragment / 4 + 8 + 63) / 64; /* 64-record chunks of one fragment (<= 8193 records): their totals fit the first 1 KiB of its id region */
static_assert(kMaxChunks <= 254, "chunk offsets + two words live in the smallest id region (1 KiB)");

/* small blocks (pages): one wave per block, chunks in order, no size pass.  (A kernel of its own:
 * inlined beside the workgroup-per-block path below, the two copies of emit_chunk cost the
 * kernel 128 VGPRs and 120 spilled SGPRs.) */
extern "C" __global__ void __launch_bounds__(64 * kEmitWaves, 4) snappy_emit_pages(CompressArgs A)
{
	__shared__ __attribute__((aligned(16))) uint8_t stage_all[kEmitWaves][kStageBytes];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	/* small blocks (pages): one wave per block, chunks in order, no size pass */
	const uint32_t b = blockIdx.x * kEmitWaves + wv;
	if (b >= A.emit_blocks)
		return;
	const uint32_t blk = A.blk_base + b;
	const uint32_t len = A.in_len[blk];
	if (len > A.max_in_len) {
		if (lane == 0)
			A.out_len[blk] = 0xffffffffu;
		return;
	}
	uint8_t *dst = A.out + A.out_off[blk];
	uint32_t pos = 0;
	if (A.mode == CSNAPPY_HIP_STREAM) {
		pos = varint_len(len);
		if (lane < pos)
			dst[lane] = (uint8_t)((len >> (7 * lane)) | (lane + 1 < pos ? 0x80u : 0u));
	}
	const uint32_t cnt = A.rec_cnt[b]; /* fpb == 1 */
	if (cnt >= kWantGlobal) {
		/* (never parsed: see below) */
		if (lane == 0)
			A.out_len[blk] = 0xffffffffu;
		return;
	}
	const uint2 *R = reinterpret_cast<const uint2 *>(A.recs + (uint64_t)b * A.rec_cap);
	const uint8_t *src = A.in + A.in_off[blk];
	/* records are fetched two chunks ahead, literal bytes one chunk ahead */
	if (cnt == 0) {
		if (lane == 0)
			A.out_len[blk] = pos;
		return;
	}
	/* (the loads behind the last chunk are those of the last chunk again: every iteration issues the same number) */
	const uint32_t r_last = (cnt - 1) & ~63u;
	ChunkIn cur, nxt;
	nxt.r = fetch_record(R, 0, min(64u, cnt), lane);
	uint2 r2 = fetch_record(R, min(64u, r_last), min(64u, cnt - min(64u, r_last)), lane);
	fetch_literal(nxt, min(64u, cnt), src, len, lane, R);
	EmitState st = { 0, 0 };
	uint8_t *body = dst + pos;
	for (uint32_t r0 = 0; r0 < cnt; r0 += 64) {
		cur = nxt;
		nxt.r = r2;
		const uint32_t r1 = min(r0 + 64, r_last), r2at = min(r0 + 128, r_last);
		auto prefetch = [&]() __attribute__((always_inline)) {
			r2 = fetch_record(R, r2at, min(64u, cnt - r2at), lane);
			fetch_literal(nxt, min(64u, cnt - r1), src, len, lane, R);
		};
#if CSNAPPY_EMIT_PROF
		unsigned long long ept_pages[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		pos += emit_chunk(cur, prefetch, min(64u, cnt - r0), src, len, body, stage_all[wv], lane, st, r0 + 64 >= cnt, ept_pages);
#else
		pos += emit_chunk(cur, prefetch, min(64u, cnt - r0), src, len, body, stage_all[wv], lane, st, r0 + 64 >= cnt);
#endif
	}
	if (lane == 0)
		A.out_len[blk] = pos;
	return;
}

/* Blocks of more than 8 KiB: three launches over the chunk of the batch.
 *   snappy_emit_sizes   one workgroup per FRAGMENT: the encoded bytes of each of its 64-record
 *                       chunks, turned into the chunks' offsets inside the fragment's output, and
 *                       the fragment's total (the parser's id / table region of the fragment is
 *                       free by now and takes the <= 129 + 2 words)
 *   snappy_emit_bases   one wave per BLOCK: the length prefix, then the fragments' totals turned
 *                       into their offsets in the block's slot, 64 fragments per step (fragment
 *                       k+1 starts where fragment k ended, :647-653); out_len
 *   snappy_emit_blocks  one workgroup per FRAGMENT: every wave encodes its 64-record chunks at
 *                       their final place, no barrier
 * (Until round 3 one workgroup did all of this for a whole block, fragment after fragment, with
 * three barriers per fragment: a block's waves waited for each other, and a long single stream
 * was emitted by one workgroup.) */
struct EmitFrag {
	uint32_t blk, fi, len, cnt, nchunks;
	const uint2 *R;
	uint32_t *base; /* per 64-record chunk: its offset in the fragment's output; [kFragTotal] the fragment's
			 * encoded bytes, [kFragBase] its offset in the block's slot (snappy_emit_bases) */
};
constexpr uint32_t kFragTotal = 254, kFragBase = 255; /* (words of the fragment's 1 KiB: kMaxChunks <= 254) */

/* false: the fragment does not exist, or its block cannot be emitted (snappy_emit_bases reports it) */
DEVINL bool emit_frag_setup(const CompressArgs &A, EmitFrag &F, bool after_bases)
{
	const uint32_t c = blockIdx.x;
	F.blk = A.blk_base + c / A.fpb;
	F.fi = c % A.fpb;
	F.len = A.in_len[F.blk];
	if (F.len > A.max_in_len)
		return false;
	const uint32_t nfr = F.len ? (F.len + kFragment - 1) / kFragment : 1;
	if (F.fi >= nfr)
		return false;
	F.cnt = A.rec_cnt[c];
	if (F.cnt >= kWantGlobal)
		return false;
	/* (a block with an unparsed fragment: snappy_emit_bases said so and laid nothing out) */
	if (after_bases && A.out_len[F.blk] == 0xffffffffu)
		return false;
	F.nchunks = (F.cnt + 63) >> 6;
	F.R = reinterpret_cast<const uint2 *>(A.recs + (uint64_t)c * A.rec_cap);
	F.base = reinterpret_cast<uint32_t *>(A.tabs + (uint64_t)c * A.tab_stride);
	return true;
}

constexpr uint32_t kSizesWaves = 4; /* waves per workgroup of snappy_emit_sizes (they meet at one barrier) */
extern "C" __global__ void __launch_bounds__(64 * kSizesWaves) snappy_emit_sizes(CompressArgs A)
{
	__shared__ uint32_t chunk_tot[kMaxChunks];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	EmitFrag F;
	if (!emit_frag_setup(A, F, false))
		return;
	/* encoded bytes of every 64-record chunk (four chunks' records in flight) */
	for (uint32_t c0 = wv; c0 < F.nchunks; c0 += 4 * kSizesWaves) {
		uint2 rr[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			const uint32_t r = (c0 + j * kSizesWaves) * 64 + lane;
			rr[j] = r < F.cnt ? F.R[r] : make_uint2(0, 0);
		}
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			const uint32_t ch = c0 + j * kSizesWaves;
			if (ch < F.nchunks) {
				const RecFields f = decode_record(rr[j], ch * 64 + lane < F.cnt);
				uint32_t total;
				(void)wave_excl_scan(f.mine, lane, &total);
				if (lane == 0)
					chunk_tot[ch] = total;
			}
		}
	}
	__syncthreads();
	if (wv == 0) {
		/* exclusive scan of the chunk totals (<= 129 entries) */
		uint32_t run = 0;
		for (uint32_t b0 = 0; b0 < F.nchunks; b0 += 64) {
			const uint32_t ch = b0 + lane;
			const uint32_t v = ch < F.nchunks ? chunk_tot[ch] : 0;
			uint32_t total;
			const uint32_t ex = wave_excl_scan(v, lane, &total);
			if (ch < F.nchunks)
				F.base[ch] = run + ex;
			run += total;
		}
		if (lane == 0)
			F.base[kFragTotal] = run;
	}
}

extern "C" __global__ void __launch_bounds__(64) snappy_emit_bases(CompressArgs A)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t blk = A.blk_base + blockIdx.x;
	const uint32_t len = A.in_len[blk];
	if (len > A.max_in_len) {
		/* the precondition in_len[b] <= max_in_len is violated: nothing was parsed for this block */
		if (lane == 0)
			A.out_len[blk] = 0xffffffffu;
		return;
	}
	const uint32_t nfr = len ? (len + kFragment - 1) / kFragment : 1;
	/* a fragment no parser launch finished (an internal error, never seen: the parsers give up
	 * instead of spinning when their cursor stops moving): the block is reported as failed */
	/* (64 fragments per load: one long stream has 32 768 of them per GiB, and a serial loop here was
	 * the one-wave tail the three-launch emit exists to avoid) */
	for (uint32_t f0 = 0; f0 < nfr; f0 += 64) {
		const uint32_t fi = f0 + lane;
		const bool unparsed = fi < nfr && A.rec_cnt[blockIdx.x * A.fpb + fi] >= kWantGlobal;
		if (ballot64(unparsed)) {
			if (lane == 0)
				A.out_len[blk] = 0xffffffffu;
			return;
		}
	}
	uint8_t *dst = A.out + A.out_off[blk];
	uint32_t pos = 0;
	if (A.mode == CSNAPPY_HIP_STREAM) {
		/* encode_varint32, csnappy_compress.c:46-73 */
		pos = varint_len(len);
		if (lane < pos)
			dst[lane] = (uint8_t)((len >> (7 * lane)) | (lane + 1 < pos ? 0x80u : 0u));
	}
	/* exclusive scan of the fragments' totals, 64 fragments per step, the loads of four steps in
	 * flight (they are 64 KiB apart: every step is a round trip to HBM) */
	for (uint32_t f0 = 0; f0 < nfr; f0 += 256) {
		uint32_t *base[4];
		uint32_t v[4];
#pragma unroll
		for (uint32_t u = 0; u < 4; ++u) {
			const uint32_t fi = f0 + 64 * u + lane;
			base[u] = reinterpret_cast<uint32_t *>(A.tabs + (uint64_t)(blockIdx.x * A.fpb + min(fi, nfr - 1)) * A.tab_stride);
			v[u] = fi < nfr ? base[u][kFragTotal] : 0;
		}
#pragma unroll
		for (uint32_t u = 0; u < 4; ++u) {
			const uint32_t fi = f0 + 64 * u + lane;
			uint32_t total;
			const uint32_t ex = wave_excl_scan(v[u], lane, &total);
			if (fi < nfr)
				base[u][kFragBase] = pos + ex;
			pos += total;
		}
	}
	if (lane == 0)
		A.out_len[blk] = pos;
}

#ifndef CSNAPPY_EMIT_OCC
#define CSNAPPY_EMIT_OCC 4 /* waves per SIMD the register allocation aims at */
#endif
extern "C" __global__ void __launch_bounds__(64 * kEmitWaves, CSNAPPY_EMIT_OCC) snappy_emit_blocks(CompressArgs A)
{
	__shared__ __attribute__((aligned(16))) uint8_t stage_all[kEmitWaves][kStageBytes];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	EmitFrag F;
	if (!emit_frag_setup(A, F, true))
		return;
	const uint8_t *src = A.in + A.in_off[F.blk] + F.fi * kFragment;
	const uint32_t avail = F.len - F.fi * kFragment;
	uint8_t *dst = A.out + A.out_off[F.blk] + F.base[kFragBase];
	const uint32_t cnt = F.cnt, nchunks = F.nchunks;
	const uint2 *R = F.R;
	/* every wave encodes a run of consecutive chunks at their final place (records fetched two
	 * chunks ahead, literal bytes one ahead); consecutive chunks are consecutive in the output,
	 * so the wave's staging is drained when it is full, not behind every chunk */
#if CSNAPPY_EMIT_PROF
	unsigned long long ept[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	const unsigned long long ept_begin = __builtin_amdgcn_s_memtime();
#endif
	const uint32_t per = (nchunks + kEmitWaves - 1) / kEmitWaves;
	const uint32_t c_lo = min(wv * per, nchunks), c_hi = min(c_lo + per, nchunks);
	if (c_lo >= c_hi)
		return;
	ChunkIn cur, nxt;
	auto cn = [&](uint32_t ch) { return min(64u, cnt - ch * 64); };
	/* (the loads behind the wave's last chunk are those of its last chunk again: every iteration issues the same number) */
	const uint32_t c_last = c_hi - 1;
	nxt.r = fetch_record(R, c_lo * 64, cn(c_lo), lane);
	uint2 r2 = fetch_record(R, min(c_lo + 1, c_last) * 64, cn(min(c_lo + 1, c_last)), lane);
	fetch_literal(nxt, cn(c_lo), src, avail, lane, R);
	EmitState st = { 0, 0 };
	uint8_t *body = dst + F.base[c_lo];
	/* (two chunks an iteration with the two register sets changing roles instead of being copied -- 21 moves a chunk --
	 * was slower: 1.07 against 0.99 ms per GiB, twice the code) */
	for (uint32_t ch = c_lo; ch < c_hi; ++ch) {
		cur = nxt;
		nxt.r = r2;
		const uint32_t c1 = min(ch + 1, c_last), c2 = min(ch + 2, c_last);
		auto prefetch = [&]() __attribute__((always_inline)) {
			r2 = fetch_record(R, c2 * 64, cn(c2), lane);
			fetch_literal(nxt, cn(c1), src, avail, lane, R);
		};
#if CSNAPPY_EMIT_PROF
		(void)emit_chunk(cur, prefetch, cn(ch), src, avail, body, stage_all[wv], lane, st, ch + 1 == c_hi, ept);
#else
		(void)emit_chunk(cur, prefetch, cn(ch), src, avail, body, stage_all[wv], lane, st, ch + 1 == c_hi);
#endif
	}
#if CSNAPPY_EMIT_PROF
	if (lane == 0) {
		atomicAdd(&g_emit_prof[0], __builtin_amdgcn_s_memtime() - ept_begin);
		for (int k = 0; k < 5; ++k)
			atomicAdd(&g_emit_prof[1 + k], ept[k]);
		atomicAdd(&g_emit_prof[8], ept[7]);
		atomicAdd(&g_emit_prof[9], 1ull);
	}
#endif
}

/* ==========================================================================================
 * DECOMPRESS: one wave per block
 *
 * Per iteration the 64 lanes look at the 64 input bytes at ip..ip+63, each decoding its byte as
 * if it were a tag.  The true tag chain is walked on the scalar unit, a wave prefix sum gives
 * each element its output offset, the reference's checks are evaluated per element in its
 * order, and the elements in front of the first failing one are executed in three passes:
 *   1. literals, one lane per element (they depend on the input only),
 *   2. copies whose source lies entirely in front of this batch's output, one lane per element,
 *   3. the remaining copies (source inside the batch, or overlapping themselves) one after the
 *      other with the whole wave, dst[j] = dst[j mod offset - offset]  (:188-206 semantics).
 * ======================================================================================== */

/* copy exactly len (<= 64) bytes from global memory into the wave's LDS staging, any alignment:
 * the pieces 16,16,16,16 / 8 / 4 / 2 / 1 that make up len are all loaded first (one memory round
 * trip for the lane, whatever its length), then all stored.  (Unconditional stores, with the pieces
 * a lane does not have sent to a scrap slot -- selects instead of exec-mask regions -- change
 * nothing: measured in round 3.) */
DEVINL void copy_exact(uint8_t *d, const uint8_t *s, uint32_t len, bool active, uint64_t &carried)
{
	const uint32_t n16 = active ? len >> 4 : 0; /* 0..4 */
	const uint32_t o8 = len & ~15u, o4 = len & ~7u, o2 = len & ~3u, o1 = len & ~1u;
	const bool b8 = active && (len & 8), b4 = active && (len & 4);
	const bool b2 = active && (len & 2), b1 = active && (len & 1);
	/* (deliberately not initialised: each piece is stored under the condition it was loaded
	 * under, and initialisers would cost 21 moves per call) */
	uint4 c0, c1, c2, c3;
	uint64_t p8;
	uint32_t p4;
	uint16_t p2;
	uint8_t p1;
	if (n16 > 0) {
		__builtin_memcpy(&c0, s, 16);
		if (n16 > 1) {
			__builtin_memcpy(&c1, s + 16, 16);
			if (n16 > 2) {
				__builtin_memcpy(&c2, s + 32, 16);
				if (n16 > 3)
					__builtin_memcpy(&c3, s + 48, 16);
			}
		}
	}
	if (b8)
		__builtin_memcpy(&p8, s + o8, 8);
	if (b4)
		__builtin_memcpy(&p4, s + o4, 4);
	if (b2)
		__builtin_memcpy(&p2, s + o2, 2);
	if (b1)
		p1 = s[o1];
	/* `carried` is a load issued before this call whose value is needed only in the next loop
	 * iteration.  Using it here makes the compiler wait for it now, together with the loads
	 * above; otherwise it waits at the loop top with vmcnt(0) -- gfx9 has one counter for loads
	 * and stores -- and every iteration would sit out the round trip of the stores below. */
	asm volatile("" : "+v"(carried));
	if (n16 > 0) {
		__builtin_memcpy(d, &c0, 16);
		if (n16 > 1) {
			__builtin_memcpy(d + 16, &c1, 16);
			if (n16 > 2) {
				__builtin_memcpy(d + 32, &c2, 16);
				if (n16 > 3)
					__builtin_memcpy(d + 48, &c3, 16);
			}
		}
	}
	if (b8)
		__builtin_memcpy(d + o8, &p8, 8);
	if (b4)
		__builtin_memcpy(d + o4, &p4, 4);
	if (b2)
		__builtin_memcpy(d + o2, &p2, 2);
	if (b1)
		d[o1] = p1;
}

#ifndef CSNAPPY_DEC_WALK4
#define CSNAPPY_DEC_WALK4 1
#endif
#ifndef CSNAPPY_DEC_PROF
#define CSNAPPY_DEC_PROF 0
#endif
#if CSNAPPY_DEC_PROF
/* development builds only (tools/build_variant.sh <name> -DCSNAPPY_DEC_PROF=1, tools/phase_dec.py):
 * s_memtime phase counters of the decompress kernel */
__device__ unsigned long long g_dec_prof[32];
#define DEC_TICK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[k] += now_ - pt_last; pt_last = now_; } while (0)
#define DEC_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DEC_COUNT(k, v) (pc[k] += (v))
#else
#define DEC_TICK(k) do { } while (0)
#define DEC_WAIT_VM() do { } while (0)
#define DEC_COUNT(k, v) do { } while (0)
#endif
[[maybe_unused]] constexpr int kWalkGroup = 4; /* tags per end-of-walk test where the plain walk is used */
constexpr uint32_t kOutStage = 2048; /* bytes of a batch's output assembled in LDS */

/* The reference's char_table (csnappy_decompress.c:152-185) as the kernels use it, computed by
 * the wave (4 entries per lane): bits 0-6 length (0 for a literal whose length follows in extra
 * bytes), 7-9 extra bytes after the tag, 10-12 offset bits 8..10 of a 1-byte-offset copy, 13 literal */
DEVINL void fill_tag_table(uint16_t *ctab, uint32_t lane)
{
	for (uint32_t b = lane; b < 256; b += 64) {
		const uint32_t kd = b & 3, up = b >> 2;
		const uint32_t lx = max(up, 59u) - 59u;          /* literal: extra length bytes 0..4, :351-353 */
		const uint32_t cx = kd + ((kd >> 1) & kd);        /* copy: offset bytes 1, 2, 4 */
		const uint32_t len = kd == 1 ? 4 + (up & 7) : up + 1;
		ctab[b] = (uint16_t)((kd == 0 && lx ? 0 : len) | ((kd == 0 ? lx : cx) << 7) |
				     ((kd == 1 ? b >> 5 : 0u) << 10) | ((kd == 0 ? 1u : 0u) << 13));
	}
	wave_lds_fence();
}

extern "C" __global__ void __launch_bounds__(64) snappy_decompress_blocks(DecompressArgs A)
{
	/* 32-bit cursors: in_len and out_cap are uint32 in the reference API too; the batch API asks
	 * for them to stay below 2^32 - 2^16 so that `cursor + lane + header` cannot wrap. */
	const uint32_t lane = threadIdx.x;
	const uint32_t blk = blockIdx.x;
	const uint8_t *src = A.in + A.in_off[blk];
	const uint32_t n = A.in_len[blk];
	uint8_t *dst = A.out + A.out_off[blk];
	const uint32_t cap = A.out_cap[blk];
	/* the tag table, computed once per wave (4 entries per lane) */
	__shared__ uint16_t ctab[256];
	if (A.skip_if && *A.skip_if)
		return;
	fill_tag_table(ctab, lane);

	uint32_t ip = 0;
	uint32_t limit = cap;
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

#if CSNAPPY_DEC_PROF
	unsigned long long pt[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, pc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	const unsigned long long pt_begin = __builtin_amdgcn_s_memtime();
	unsigned long long pt_last = pt_begin;
#endif
	uint32_t op = 0;        /* bytes produced */
	uint64_t next8 = 0;     /* the 8 input bytes at the next scan iteration's ip + lane */
	bool have_next = false; /* ... valid for every lane */
	/* queue of decoded elements between the two phases: {tag position, length, offset, header
	 * size | literal << 3}, a ring of 128 (a scan iteration adds at most 64) */
	__shared__ uint4 queue[128];
	__shared__ __attribute__((aligned(16))) uint8_t ostage[kOutStage + 16 + 64];
	uint32_t qh = 0, qn = 0;
	const uint64_t lt_mask = (1ull << lane) - 1;
	while (status == CSNAPPY_E_OK) {
		/* ================= phase 1: scan -- find the real tags, queue what they decode to ======
		 * Every lane decodes the byte at ip+lane as if it were a tag (one LDS load of the
		 * reference's char_table entry, csnappy_decompress.c:152-185: bits 0-6 length, 7-9 extra
		 * bytes, 10-12 offset bits 8..10 of a 1-byte-offset copy, 13 literal; then :348-365 as
		 * selects), the true tag chain is walked on the scalar unit (five instructions per tag),
		 * and only the lanes on it write an element.  (Scanning one byte per lane and decoding
		 * only the real tags in phase 2 was tried: fewer instructions, but one more dependent
		 * global round trip per batch -- 2.41 -> 2.85 ms per GiB of text.) */
		while (qn < 64 && ip < n) {
			DEC_TICK(11); /* (whatever ran since the last tick) */
			DEC_COUNT(1, 1);
			DEC_WAIT_VM();
			DEC_TICK(0); /* scan: wait for the window's bytes */
			const uint32_t at = ip + lane;
			uint32_t b0 = 0, tr = 0;
			if (have_next) {
				b0 = (uint32_t)next8 & 0xff;
				tr = (uint32_t)(next8 >> 8);
			} else if (at + 8 <= n) {
				uint64_t v;
				__builtin_memcpy(&v, src + at, 8);
				b0 = (uint32_t)v & 0xff;
				tr = (uint32_t)(v >> 8);
				asm volatile("" : "+v"(b0), "+v"(tr)); /* wait here, not where the paths join */
			} else {
				if (at < n)
					b0 = src[at];
#pragma unroll
				for (int kk = 0; kk < 4; ++kk)
					if (at + 1 + kk < n)
						tr |= (uint32_t)src[at + 1 + kk] << (8 * kk);
				asm volatile("" : "+v"(b0), "+v"(tr));
			}
			const uint32_t e = ctab[b0];
			const uint32_t extra = (e >> 7) & 7u;
			const bool is_lit = (e >> 13) & 1u;
			const uint32_t xmask = 0xffffffffu >> ((32u - 8u * extra) & 31u); /* extra 0 -> all ones (unused) */
			const uint32_t trm = tr & xmask;
			const uint32_t l = (is_lit && extra != 0) ? trm + 1 : (e & 127u);
			const uint32_t off = is_lit ? 0u : trm | (((e >> 10) & 7u) << 8);
			const uint32_t hsz = 1 + extra;
			/* bytes this element takes in the input (a literal length that would wrap 32 bits is
			 * negative as int32 and fails in phase 2 whatever the walk does after it) */
			const uint32_t esz = is_lit ? (l >= 0xfffffff0u ? 0xfffffff8u : hsz + l) : hsz;
			uint64_t tmask = 0;
			uint32_t cur = 0;
			const uint32_t room = n - ip; /* > 0 */
			const uint32_t wlim = min(64u, room);
			const uint32_t nxt = lane + esz < lane ? 0xffffffffu : lane + esz; /* next tag if I am one */
			/* The walk: two instructions per tag (s_bitset1 + v_readlane) in groups of four.  A step
			 * that leaves the window goes back to lane 0, where the walk only marks tags again that
			 * are marked already, so the end is tested once per group: the steps taken no longer
			 * equal the tags marked.  (Marking with v_writelane instead -- no scalar instruction per
			 * tag at all -- was measured in round 3: 2.31 against 2.28 ms, groups of 4 or 8.) */
			const uint32_t nxw = nxt < wlim ? nxt : 0u;
#if CSNAPPY_DEC_PROF
			asm volatile("" : : "v"(nxw), "v"(nxt));
#endif
			DEC_TICK(1); /* scan: decode */
			uint32_t steps = 0;
#if CSNAPPY_DEC_WALK4
			{
				/* Four tags per dependent step.  What a hop costs is not its instructions but the
				 * trip of v_readlane's result to the scalar register file (some 40 cycles before the
				 * next instruction that uses it may issue): the tables of the 2nd, 3rd and 4th
				 * successors (three ds_bpermute per window) let the wave fetch four tags' worth of
				 * lanes back to back and pay that trip once.  A walk that left the window goes on
				 * around the same chain (lane 0 follows the last tag), so the tags are marked in
				 * chain order, none is skipped, and the first tag marked twice ends it as before. */
				const uint32_t n2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nxw << 2), (int)nxw);
				const uint32_t n3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(n2 << 2), (int)nxw);
				const uint32_t n4 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(n2 << 2), (int)n2);
				do {
					uint32_t ta, tb, tc, tf;
					asm volatile("v_readlane_b32 %5, %9, %1\n\t"
						     "v_readlane_b32 %2, %6, %1\n\t"
						     "v_readlane_b32 %3, %7, %1\n\t"
						     "v_readlane_b32 %4, %8, %1\n\t"
						     "s_bitset1_b64 %0, %1\n\t"
						     "s_bitset1_b64 %0, %2\n\t"
						     "s_bitset1_b64 %0, %3\n\t"
						     "s_bitset1_b64 %0, %4\n\t"
						     "v_readlane_b32 %1, %9, %5\n\t"
						     "v_readlane_b32 %2, %6, %5\n\t"
						     "v_readlane_b32 %3, %7, %5\n\t"
						     "v_readlane_b32 %4, %8, %5\n\t"
						     "s_bitset1_b64 %0, %5\n\t"
						     "s_bitset1_b64 %0, %2\n\t"
						     "s_bitset1_b64 %0, %3\n\t"
						     "s_bitset1_b64 %0, %4"
						     : "+s"(tmask), "+s"(cur), "=&s"(ta), "=&s"(tb), "=&s"(tc), "=&s"(tf)
						     : "v"(nxw), "v"(n2), "v"(n3), "v"(n4));
					steps += 8;
				} while ((uint32_t)__builtin_popcountll(tmask) == steps);
			}
#else
			do {
#pragma unroll
				for (int kk = 0; kk < kWalkGroup; ++kk) {
					asm("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(cur));
					cur = rdlane(nxw, cur);
				}
				steps += kWalkGroup;
			} while ((uint32_t)__builtin_popcountll(tmask) == steps);
#endif
			cur = rdlane(nxt, 63u - (uint32_t)__builtin_clzll(tmask)); /* where the last tag's element ends */
			DEC_TICK(2); /* scan: walk */
			/* request the next iteration's bytes now */
			have_next = cur < room && room - cur >= 64 + 8;
			if (have_next)
				__builtin_memcpy(&next8, src + (ip + cur + lane), 8);
			if ((tmask >> lane) & 1) {
				const uint32_t r = (uint32_t)__builtin_popcountll(tmask & lt_mask);
				queue[(qh + qn + r) & 127u] = make_uint4(at, l, off, hsz | (is_lit ? 8u : 0u));
			}
			qn += (uint32_t)__builtin_popcountll(tmask);
			ip = cur >= room ? n : ip + cur;
			DEC_TICK(3); /* scan: next request, queue write */
		}
		if (qn == 0)
			break;
		DEC_COUNT(0, 1);
		wave_lds_fence();
		DEC_TICK(11);
		/* ================= phase 2: execute up to 64 elements, one lane each ===================== */
		const uint32_t m = min(qn, 64u);
		const bool live = lane < m;
		const uint4 q = queue[(qh + lane) & 127u];
		wave_lds_fence();
		const uint32_t at = q.x, l = live ? q.y : 0u, off = q.z, hsz = q.w & 7u;
		const bool is_lit = (q.w >> 3) & 1u;
		/* ---- per-element checks, in the reference's order (Appendix C of SURVEY.md) ---- */
		const bool trunc = at + hsz > n; /* header bytes cut off: reference is undefined, we say -5 */
		const uint32_t avail = trunc ? 0 : n - (at + hsz);
		const bool lit_short = is_lit && (int32_t)l >= 0 && avail < l; /* :374-375 */
		const bool lit_neg = is_lit && (int32_t)l < 0;
		const bool inbad = trunc || lit_short || lit_neg;
		const uint32_t eff = (live && !inbad) ? l : 0;
		const uint32_t excl = wave_incl_scan_dpp(eff) - eff;
		const uint32_t pb = op + excl; /* bytes produced before this element */
		/* literal: short input (-5), then output overrun (-3, :288-289, :274-275), then a length
		 * that is negative as int32 (-5); copy: offset 0 or beyond what was produced (-5,
		 * :301-303), then output overrun (-3, :311-312); cut-off header bytes first of all (-5) */
		const bool overrun = limit - pb < l;
		const bool first5 = trunc || (is_lit ? lit_short : (off == 0 || off > pb));
		const int32_t err_tag = first5 ? CSNAPPY_E_DATA_MALFORMED
				      : overrun ? CSNAPPY_E_OUTPUT_OVERRUN
				      : lit_neg ? CSNAPPY_E_DATA_MALFORMED : 0;
		const int32_t err = live ? err_tag : 0;
		const uint64_t emask = ballot64(err != 0);
		const uint32_t fe = emask ? first_lane(emask) : 64;
		const uint64_t livemask = m < 64 ? (1ull << m) - 1 : ~0ull;
		const uint64_t run = fe < 64 ? (livemask & ((1ull << fe) - 1)) : livemask; /* elements to execute */
		const bool exec_me = (run >> lane) & 1;

		/* The batch's output is assembled in LDS and flushed with aligned 16 B/lane stores: copies
		 * whose source lies inside the batch (most short-distance copies, with 64 elements per
		 * batch) then cost an LDS round trip each instead of an HBM one.  nfit = leading elements
		 * whose output fits the staging; staged byte t is output byte op + t and sits at
		 * ostage[sa + t], sa = (dst + op) & 15, so LDS 16 B chunks line up with global ones. */
		const uint32_t nrun = fe < 64 ? fe : m;
		const uint64_t fitmask = ballot64(exec_me && excl + l <= kOutStage);
		const uint32_t nfit = ~fitmask ? first_lane(~fitmask) : 64u;
		DEC_TICK(4); /* queue read, checks, offsets */
		DEC_COUNT(2, nfit);
		if (nfit == 0 && nrun > 0) {
			/* the first element does not fit: a long literal (copies are <= 64 bytes).  4 x 16 B per
			 * lane per iteration straight to HBM (the four loads are issued before the first store
			 * so that one round trip moves 4 KiB), byte tail */
			const uint32_t L = rdlane(l, 0);
			const uint8_t *ps = src + (rdlane(at, 0) + rdlane(hsz, 0));
			uint8_t *pd = dst + op;
			const uint32_t body = L & ~15u;
			for (uint32_t j0 = lane * 16; j0 < body; j0 += 4096) {
				/* (unconditional loads from a clamped address -- body >= 16 here: the first element
				 * did not fit the staging -- keep the four pieces in registers; loaded under the
				 * condition they are stored under they become an array in scratch, 80 B per lane that
				 * every wave of the kernel pays the set-up of) */
				const uint32_t last16 = body - 16;
				uint4 v0, v1, v2, v3;
				__builtin_memcpy(&v0, ps + min(j0, last16), 16);
				__builtin_memcpy(&v1, ps + min(j0 + 1024u, last16), 16);
				__builtin_memcpy(&v2, ps + min(j0 + 2048u, last16), 16);
				__builtin_memcpy(&v3, ps + min(j0 + 3072u, last16), 16);
				__builtin_memcpy(pd + j0, &v0, 16);
				if (j0 + 1024u < body)
					__builtin_memcpy(pd + j0 + 1024u, &v1, 16);
				if (j0 + 2048u < body)
					__builtin_memcpy(pd + j0 + 2048u, &v2, 16);
				if (j0 + 3072u < body)
					__builtin_memcpy(pd + j0 + 3072u, &v3, 16);
			}
			if (body + lane < L)
				pd[body + lane] = ps[body + lane];
			op += L;
			qh = (qh + 1) & 127u;
			qn -= 1;
			DEC_TICK(10); /* long literal, straight to HBM */
			DEC_COUNT(5, 1);
			continue;
		}
		if (nfit > 0) {
			const uint32_t sa = (uint32_t)(reinterpret_cast<uintptr_t>(dst + op) & 15u);
			const bool mine = lane < nfit;
			uint8_t *o = ostage + sa + excl;
			const bool lit = mine && is_lit;
			const bool cpy = mine && !is_lit;
			const bool indep = cpy && off >= excl + l; /* source ends in front of this batch's output */
			/* ---- literals up to 64 bytes (SAW__Append / SAW__AppendFastPath, :264-293) and copies
			 * that read only what earlier batches produced, one lane per element ---- */
			copy_exact(o, lit ? src + (at + hsz) : dst + pb - off, l, (lit && l <= 64) || indep, next8);
			DEC_TICK(6); /* one-lane copies: round trip, piece stores */
			DEC_COUNT(3, __builtin_popcountll(ballot64(cpy && !indep)));
			DEC_COUNT(4, __builtin_popcountll(ballot64(lit && l > 64)));
			/* ---- longer literals: wave-wide, 16 B per lane ---- */
			for (uint64_t big = ballot64(lit && l > 64); big;) {
				const uint32_t t = first_lane(big);
				asm("s_bitset0_b64 %0, %1" : "+s"(big) : "s"(t));
				const uint32_t L = rdlane(l, t);
				const uint8_t *ps = src + (rdlane(at, t) + rdlane(hsz, t));
				uint8_t *pd = ostage + sa + rdlane(excl, t);
				const uint32_t body = L & ~15u;
				for (uint32_t j0 = lane * 16; j0 < body; j0 += 1024) {
					uint4 v;
					__builtin_memcpy(&v, ps + j0, 16);
					__builtin_memcpy(pd + j0, &v, 16);
				}
				if (body + lane < L)
					pd[body + lane] = ps[body + lane];
			}
			wave_lds_fence();
			DEC_TICK(7); /* literals of more than 64 bytes */
			/* ---- the other copies, in order (SAW__AppendFromSelf, :295-317): source inside this
			 * batch's output or overlapping itself, dst[j] = dst[j mod offset - offset]; bytes from
			 * in front of the batch come from HBM, the rest from the staging ---- */
			for (uint64_t dep = ballot64(cpy && !indep); dep;) {
				const uint32_t t = first_lane(dep);
				asm("s_bitset0_b64 %0, %1" : "+s"(dep) : "s"(t));
				const uint32_t L = rdlane(l, t), OFF = rdlane(off, t), E = rdlane(excl, t);
				if (lane < L) {
					/* lane mod OFF for lane < 64 (a self-overlapping copy only; the generic 32-bit
					 * remainder is two dozen instructions): the quotient by a float reciprocal can
					 * only come out one too small, at exact multiples */
					uint32_t j = lane;
					if (OFF < L) {
						const uint32_t q = (uint32_t)((float)lane * __builtin_amdgcn_rcpf((float)OFF));
						const uint32_t r = lane - q * OFF;
						j = min(r, r - OFF);
					}
					const int32_t s = (int32_t)(E + j) - (int32_t)OFF; /* relative to the batch's start */
					uint32_t byte = ostage[sa + (uint32_t)max(s, 0)];
					asm volatile("" : "+v"(byte)); /* keeps the two loads apart (merged, they become one flat load) */
					if (s < 0)
						byte = dst[(int64_t)op + s];
					ostage[sa + E + lane] = (uint8_t)byte;
				}
				wave_lds_fence();
			}
			DEC_TICK(8); /* dependent copies */
			/* ---- flush ----
			 * ORDERING RELIED ON (and not stated by the ISA manual): later batches of this wave load
			 * bytes these stores write (copy_exact's source `dst + pb - off`, the `dst[op + s]` of a
			 * dependent copy), and no s_waitcnt sits between the stores and those loads -- vmcnt counts
			 * a store as done when it is acknowledged, a wait here would cost every batch a memory
			 * round trip.  What makes the loads see the bytes: all vector-memory instructions of ONE
			 * wave go through the CU's texture addresser and its L1 (TCP) in issue order; the TCP is
			 * write-through and a store updates or invalidates the line it hits before a younger access
			 * of the same wave is looked up, and from the TCP on both take the same path to the same L2
			 * channel (an address maps to one channel), which keeps them in order.  No other wave,
			 * workgroup or agent writes this block's slot.  This is an argument about gfx9 hardware,
			 * not a guarantee of the programming model; it is held by the soaks (tens of millions of
			 * blocks, every one compared with the reference) and would show as wrong bytes there. */
			{
				const uint32_t total = rdlane(excl, nfit - 1) + rdlane(l, nfit - 1);
				uint8_t *gbase = dst + op - sa; /* 16 B aligned */
				const uint32_t end = sa + total;
				uint32_t first_full = 0;
				if (sa > 0) {
					const uint32_t hend = min(16u, end);
					if (lane >= sa && lane < hend)
						gbase[lane] = ostage[lane];
					first_full = 1;
				}
				const uint32_t nfull = end >> 4;
				for (uint32_t cc = first_full + lane; cc < nfull; cc += 64)
					reinterpret_cast<uint4 *>(gbase)[cc] = reinterpret_cast<const uint4 *>(ostage)[cc];
				const uint32_t tail0 = nfull << 4;
				const uint32_t tail = (end > tail0 && (nfull >= 1 || sa == 0)) ? end - tail0 : 0;
				if (lane < tail)
					gbase[tail0 + lane] = ostage[tail0 + lane];
				op += total;
				wave_lds_fence();
			}
			DEC_TICK(9); /* flush */
		}
		if (fe < 64 && nfit == nrun) {
			status = (int32_t)rdlane((uint32_t)err, fe);
			break;
		}
		qh = (qh + nfit) & 127u;
		qn -= nfit;
	}

#if CSNAPPY_DEC_PROF
	if (lane == 0) {
		atomicAdd(&g_dec_prof[0], __builtin_amdgcn_s_memtime() - pt_begin);
		for (int k = 0; k < 12; ++k)
			atomicAdd(&g_dec_prof[1 + k], pt[k]);
		for (int k = 0; k < 8; ++k)
			atomicAdd(&g_dec_prof[16 + k], pc[k]);
		atomicAdd(&g_dec_prof[24], 1ull);
	}
#endif
	if (lane == 0) {
		A.status[blk] = status;
		/* on -3 / -5 the decoded prefix is in dst, as after the reference's write-as-you-go
		 * SnappyArrayWriter (csnappy_decompress.c:258-317): its length is reported too (header
		 * errors -1 / -2 happen before anything is written: op is 0 then) */
		A.produced[blk] = op;
	}
}

/* ==========================================================================================
 * STREAM INDEX: one long stream decoded by many waves  
 *
 * csnappy_decompress (csnappy_decompress.c:390-415) walks a stream of any length tag by tag; the
 * batch kernel above does the same with one wave per stream, which is correct for every stream
 * but leaves a 5 MiB file to a single wave.  What makes a stream divisible is the compressor's
 * fragmenting (csnappy_compress.c:585-616): csnappy_compress restarts its table every 32 KiB of
 * input, so no copy reaches across a multiple of 32 KiB of OUTPUT and an element starts at each of
 * them.  The pre-pass below finds those elements without decoding anything:
 *   index    one wave per 4 KiB segment of the compressed bytes.  (a) It parses the segment from
 *            its first byte, which is usually not a tag: per 64-byte window the tags met and the
 *            bytes they produce, and where this speculative parse leaves the segment.  Parses
 *            that start at different bytes fall into step after a few elements, and from a
 *            common tag on they are one parse.  (b) It settles the matter for EVERY byte of the
 *            segment: each byte, read as a tag, points at the next one; pointer doubling in LDS
 *            (12 sweeps over 4096 u16) turns that into "the last tag inside the segment on the
 *            parse that starts here".  The table goes to HBM, and with it the length of the
 *            prefix of the segment in which every byte ends on the speculative parse's last tag:
 *            a parse entering there leaves with the speculative one, no questions asked.
 *   chain    one wave strings the segments together, exactly, in one pass: an element that
 *            reaches beyond its segment (a long literal) is followed and its successor's header
 *            read on the spot; an entry inside the safe prefix leaves with the speculative parse
 *            (no memory access at all: the common case); any other entry looks its last tag up
 *            in the segment's table and reads two headers (literal-heavy data, where the
 *            speculative parse wanders through literal bytes and never meets the true one).
 *   settle   one wave per segment walks the parse from its entry until it meets the speculative
 *            one and corrects the windows in front of that point.
 *   scan     exclusive sum of the segments' output bytes.
 *   bounds   one wave per segment looks for elements that start at a multiple of 32 KiB of output.
 *   grain    all multiples found: 32 KiB fragments.  Every other one: the stream comes from a
 *            Snappy with 64 KiB blocks, pairs of fragments are decoded as one block.
 *   describe one thread per fragment turns the boundaries into batch descriptors.
 * Then snappy_decompress_blocks decodes the fragments as independent no-header blocks, a verdict
 * kernel accepts the result only if every fragment decoded cleanly to exactly its size, the
 * parse ended at the last input byte and the sizes add up to the expected length -- in that case
 * the one-wave decode would have executed the same elements on the same bytes.  In every other
 * case (a foreign compressor that copies across 32 KiB, a damaged stream, an element larger than
 * a fragment) the one-wave decode runs and its result, error code included, is the answer.
 * ======================================================================================== */
constexpr uint32_t kSegBytes = 4096;       /* 64 windows of 64 bytes: one window per lane */
constexpr uint32_t kNoEntry = 0xffffffffu; /* the true parse does not touch this segment */
constexpr uint32_t kHugeOut = 1u << 24;    /* window outputs saturate here; no fragment holds that much */

struct StreamArgs {
	const uint8_t *in;   /* the body: the stream without its length header */
	uint32_t n, ulength; /* its bytes; the room for what it produces (*dst_len of the reference call) */
	uint32_t nseg, nfrag;
	uint64_t *tagmask;   /* [nseg * 64] tags met in each window */
	uint32_t *winout;    /* [nseg * 64] bytes they produce (saturating) */
	uint64_t *truemask;  /* [nseg * 64] the same of the true parse (settle) */
	uint32_t *trueout;
	uint16_t *last_tag;  /* [nseg * 4096] last tag inside the segment on the parse that starts at each byte */
	/* per segment [nseg]: */
	uint32_t *seg_exit;  /* where the speculative parse left it ... */
	uint32_t *seg_xesz;  /* ... and the input bytes of the element that starts there */
	uint32_t *seg_safe;  /* bytes at its start from which every parse leaves with the speculative one */
	uint32_t *seg_entry; /* where the true parse enters it, or kNoEntry */
	uint32_t *seg_leave; /* ... and leaves it */
	uint32_t *grp_e, *grp_esz; /* [ceil(nseg / 64)] the chain's state behind each group of 64 segments */
	uint64_t *seg_out;   /* [nseg] bytes the segment produces; after the scan, its output offset */
	uint32_t *frag_pos;  /* [nfrag] input position of the element that starts fragment f */
	uint64_t *f_in_off, *f_out_off; /* [nfrag] batch descriptors of the fragments ... */
	uint32_t *f_in_len, *f_out_cap, *f_produced;
	int32_t *f_status;
	uint64_t *one_off;   /* ... and [2] zeros: offsets of the one-block slow path */
	uint32_t *one_len;   /* [2] n, ulength: its in_len and out_cap */
	uint32_t *flags;     /* SF_*: refused, end of the true parse, verdict (1 = fragments stand), grain */
	uint64_t *total;     /* [1] bytes the parse produces */
	int32_t *status;     /* the caller's result */
	uint32_t *produced;
};
enum { SF_REFUSED = 0, SF_END = 1, SF_VERDICT = 2, SF_GRAIN = 3, SF_NFRAG = 4, SF_COUNT = 8 };

struct TagAt {
	uint32_t esz; /* bytes the element takes in the input (saturating) */
	uint32_t l;   /* bytes it produces (saturating at kHugeOut) */
};

/* the byte at `at` read as a tag (csnappy_decompress.c:348-365 as selects; the same arithmetic as
 * the decompressor's scan).  Header bytes beyond the input read as 0: such an element ends beyond
 * the input whatever they were, which is all that matters here. */
DEVINL TagAt tag_at(const uint8_t *src, uint32_t n, uint32_t at, const uint16_t *ctab)
{
	uint32_t b0 = 0, tr = 0;
	const uint32_t room = at < n ? n - at : 0u; /* (`at` may be a saturated 0xffffffff: no at + k here) */
	if (room >= 8) {
		uint64_t v;
		__builtin_memcpy(&v, src + at, 8);
		b0 = (uint32_t)v & 0xff;
		tr = (uint32_t)(v >> 8);
	} else {
		if (room > 0)
			b0 = src[at];
#pragma unroll
		for (uint32_t kk = 0; kk < 4; ++kk)
			if (1 + kk < room)
				tr |= (uint32_t)src[at + 1 + kk] << (8 * kk);
	}
	const uint32_t e = ctab[b0];
	const uint32_t extra = (e >> 7) & 7u;
	const bool is_lit = (e >> 13) & 1u;
	const uint32_t trm = tr & (0xffffffffu >> ((32u - 8u * extra) & 31u));
	const uint32_t l = (is_lit && extra != 0) ? trm + 1 : (e & 127u);
	const uint32_t hsz = 1 + extra;
	TagAt t;
	t.esz = is_lit ? (l >= 0xfffffff0u ? 0xfffffff8u : hsz + l) : hsz;
	t.l = min(l, kHugeOut);
	return t;
}

DEVINL uint64_t rdlane64(uint64_t v, uint32_t l)
{
	return (uint64_t)rdlane((uint32_t)v, l) | ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32);
}

/* The tags of one window on the parse that enters it at lane `start` (< wlim = bytes of the window
 * inside the input): each tag names the next one, nxt = lane + bytes its element takes.  Same walk
 * as the decompressor's: a step out of the window goes back to `start`, where nothing new is
 * marked.  *leave = window-relative position at which the parse leaves (>= wlim). */
DEVINL uint64_t walk_window(uint32_t esz, uint32_t lane, uint32_t start, uint32_t wlim, uint64_t *leave)
{
	const uint32_t nxt = lane + esz < lane ? 0xffffffffu : lane + esz;
	const uint32_t nxw = nxt < wlim ? nxt : start;
	uint64_t tmask = 0;
	uint32_t cur = start, steps = 0;
	do {
#pragma unroll
		for (int kk = 0; kk < 4; ++kk) {
			asm("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(cur));
			cur = rdlane(nxw, cur);
		}
		steps += 4;
	} while ((uint32_t)__builtin_popcountll(tmask) == steps);
	*leave = rdlane(nxt, 63u - (uint32_t)__builtin_clzll(tmask));
	return tmask;
}

/* bytes the tags in `mask` produce, saturating */
DEVINL uint32_t window_out(uint64_t mask, uint32_t l, uint32_t lane)
{
	const uint32_t x = wave_incl_scan_dpp(((mask >> lane) & 1) ? l : 0u);
	return min(rdlane(x, 63), kHugeOut);
}

extern "C" __global__ void __launch_bounds__(64) snappy_stream_index(StreamArgs A)
{
	__shared__ uint16_t ctab[256];
	__shared__ uint16_t last[kSegBytes]; /* first: the next tag from each byte (itself if that lies outside) */
	const uint32_t lane = threadIdx.x, seg = blockIdx.x;
	fill_tag_table(ctab, lane);
	const uint32_t seg_lo = seg * kSegBytes;
	const uint32_t seg_len = min(kSegBytes, A.n - seg_lo);
	uint64_t pos = seg_lo; /* next tag of the speculative parse */
	uint64_t my_mask = 0;
	uint32_t my_out = 0;
	TagAt t = tag_at(A.in, A.n, seg_lo + lane, ctab);
	for (uint32_t w = 0; w < 64; ++w) {
		const uint32_t base = seg_lo + 64 * w;
		const uint32_t p = 64 * w + lane;
		if (base >= A.n) {
			last[p] = (uint16_t)p;
			continue;
		}
		const TagAt cur = t;
		if (w + 1 < 64 && base + 64 < A.n)
			t = tag_at(A.in, A.n, base + 64 + lane, ctab); /* in flight during the walk */
		last[p] = (uint16_t)((uint64_t)p + cur.esz < seg_len ? p + cur.esz : p);
		const uint32_t wlim = min(64u, A.n - base);
		if (pos >= (uint64_t)base + wlim)
			continue;
		uint64_t leave;
		const uint64_t mask = walk_window(cur.esz, lane, (uint32_t)(pos - base), wlim, &leave);
		const uint32_t out = window_out(mask, cur.l, lane);
		pos = base + leave;
		if (lane == w) {
			my_mask = mask;
			my_out = out;
		}
	}
	A.tagmask[(size_t)seg * 64 + lane] = my_mask;
	A.winout[(size_t)seg * 64 + lane] = my_out;
	const uint32_t leave = pos > 0xffffffffull ? 0xffffffffu : (uint32_t)pos;
	const TagAt next = tag_at(A.in, A.n, leave, ctab); /* (every lane the same tag) */
	if (lane == 0) {
		A.seg_exit[seg] = leave;
		A.seg_xesz[seg] = next.esz;
	}
	/* pointer doubling, in place (a pointer only ever moves along its own parse, so reading a
	 * value another lane has just advanced is as good): elements take >= 2 bytes, a parse has at
	 * most 2048 of them in a segment, 11 sweeps would do */
	wave_lds_fence();
	for (uint32_t sweep = 0; sweep < 12; ++sweep) {
		bool moved = false;
		for (uint32_t i = 0; i < 64; ++i) {
			const uint32_t p = 64 * i + lane;
			const uint32_t v = last[p], vv = last[v];
			moved |= vv != v;
			last[p] = (uint16_t)vv;
		}
		wave_lds_fence();
		if (!ballot64(moved))
			break;
	}
	/* the safe prefix: bytes from which the parse ends on the speculative parse's last tag */
	const uint32_t spec_last = last[0];
	uint32_t safe = kSegBytes;
	for (uint32_t i = 0; i < 64; ++i) {
		const uint64_t off = ballot64(last[64 * i + lane] != spec_last);
		if (off) {
			safe = 64 * i + first_lane(off);
			break;
		}
	}
	if (lane == 0)
		A.seg_safe[seg] = safe;
	uint4 *dst = reinterpret_cast<uint4 *>(A.last_tag + (size_t)seg * kSegBytes);
	const uint4 *src = reinterpret_cast<const uint4 *>(last);
	for (uint32_t i = lane; i < kSegBytes * 2 / 16; i += 64)
		dst[i] = src[i];
}

/* The chain through segments [c + j0, c + m): (e, esz) = where the true parse stands and the input
 * bytes of the element that starts there; sx / sz / sf = the lanes' copies of seg_exit / seg_xesz /
 * seg_safe of segments c + lane.  Lane j keeps the entry and the leave of segment c + j. */
DEVINL void chain_segments(const StreamArgs &A, const uint16_t *ctab, uint32_t c, uint32_t m, uint32_t j0, uint32_t sx,
			   uint32_t sz, uint32_t sf, uint32_t &e, uint32_t &esz, uint32_t &ent, uint32_t &lv, uint32_t lane)
{
	for (uint32_t j = j0; j < m; ++j) {
		const uint32_t lo = (c + j) * kSegBytes;
		const uint64_t hi64 = (uint64_t)lo + kSegBytes;
		const uint32_t hi = hi64 < A.n ? (uint32_t)hi64 : A.n;
		if (e >= hi)
			continue; /* inside an element that started earlier */
		if (lane == j)
			ent = e;
		const uint64_t after = (uint64_t)e + esz;
		if (after >= hi) {
			/* the element at the entry reaches beyond the segment: follow it */
			e = after > 0xffffffffull ? 0xffffffffu : (uint32_t)after;
			esz = rdlane(tag_at(A.in, A.n, e, ctab).esz, 0);
		} else if (e - lo < rdlane(sf, j)) {
			e = rdlane(sx, j);
			esz = rdlane(sz, j);
		} else {
			/* anywhere else: the table knows the last tag on this parse; two headers to read */
			const uint32_t lt = lo + A.last_tag[(size_t)(c + j) * kSegBytes + (e - lo)];
			const uint64_t out = (uint64_t)lt + rdlane(tag_at(A.in, A.n, lt, ctab).esz, 0);
			e = out > 0xffffffffull ? 0xffffffffu : (uint32_t)out;
			esz = rdlane(tag_at(A.in, A.n, e, ctab).esz, 0);
		}
		if (lane == j)
			lv = e;
	}
}

/* chain, step 1 of 2: every group of 64 segments at once.  A group other than the first does not
 * know where the parse enters it; it ASSUMES the entry lies in the safe prefix of its first segment
 * and is not an element that reaches beyond that segment -- then the parse leaves the first segment
 * with the speculative one whatever the entry was -- and walks its other 63 segments exactly.
 * (Round 3: one wave used to walk all segments of the stream, 5.2 of the 6.9 ms a 256 MiB stream took.) */
extern "C" __global__ void __launch_bounds__(64) snappy_stream_chain_groups(StreamArgs A)
{
	__shared__ uint16_t ctab[256];
	const uint32_t lane = threadIdx.x;
	fill_tag_table(ctab, lane);
	const uint32_t g = blockIdx.x, c = g * 64;
	const uint32_t m = min(64u, A.nseg - c);
	const uint32_t k = c + lane < A.nseg ? c + lane : 0u;
	const uint32_t sx = A.seg_exit[k], sz = A.seg_xesz[k], sf = A.seg_safe[k];
	uint32_t e, esz, ent = kNoEntry, lv = 0;
	if (g == 0) {
		e = 0; /* the parse starts at the first byte of the body ... */
		esz = rdlane(tag_at(A.in, A.n, 0, ctab).esz, 0); /* ... with an element of this many bytes */
		chain_segments(A, ctab, c, m, 0, sx, sz, sf, e, esz, ent, lv, lane);
	} else {
		e = rdlane(sx, 0);
		esz = rdlane(sz, 0);
		if (lane == 0)
			lv = e; /* (its entry is written by snappy_stream_chain_link once it is known) */
		chain_segments(A, ctab, c, m, 1, sx, sz, sf, e, esz, ent, lv, lane);
	}
	if (c + lane < A.nseg) {
		A.seg_entry[c + lane] = ent;
		A.seg_leave[c + lane] = lv;
	}
	if (lane == 0) {
		A.grp_e[g] = e;
		A.grp_esz[g] = esz;
	}
}

/* chain, step 2 of 2: one wave strings the groups together.  Where the parse really enters a
 * group as the group assumed, the group's result stands and its exit is the state for the next
 * one (three compares per group); where it does not -- a literal that covers the group's first
 * segment, an entry behind the safe prefix: literal-heavy data -- the group is walked again from the
 * true state, as the one-wave chain did for every group. */
extern "C" __global__ void __launch_bounds__(64) snappy_stream_chain_link(StreamArgs A)
{
	__shared__ uint16_t ctab[256];
	const uint32_t lane = threadIdx.x;
	fill_tag_table(ctab, lane);
	const uint32_t ngrp = (A.nseg + 63) / 64;
	uint32_t e = A.grp_e[0], esz = A.grp_esz[0];
	for (uint32_t g0 = 1; g0 < ngrp; g0 += 64) {
		/* the next 64 groups' exits and the safe prefixes of their first segments */
		const uint32_t gk = g0 + lane < ngrp ? g0 + lane : 0u;
		const uint32_t ge = A.grp_e[gk], gz = A.grp_esz[gk], gf = A.seg_safe[(size_t)gk * 64];
		const uint32_t mg = min(64u, ngrp - g0);
		for (uint32_t j = 0; j < mg; ++j) {
			const uint32_t g = g0 + j, c = g * 64;
			const uint32_t lo = c * kSegBytes;
			const uint64_t hi64 = (uint64_t)lo + kSegBytes;
			const uint32_t hi = hi64 < A.n ? (uint32_t)hi64 : A.n;
			if (e >= lo && e < hi && (uint64_t)e + esz < hi && e - lo < rdlane(gf, j)) {
				if (lane == 0)
					A.seg_entry[c] = e;
				e = rdlane(ge, j);
				esz = rdlane(gz, j);
			} else {
				const uint32_t m = min(64u, A.nseg - c);
				const uint32_t k = c + lane < A.nseg ? c + lane : 0u;
				const uint32_t sx = A.seg_exit[k], sz = A.seg_xesz[k], sf = A.seg_safe[k];
				uint32_t ent = kNoEntry, lv = 0;
				chain_segments(A, ctab, c, m, 0, sx, sz, sf, e, esz, ent, lv, lane);
				if (c + lane < A.nseg) {
					A.seg_entry[c + lane] = ent;
					A.seg_leave[c + lane] = lv;
				}
			}
		}
	}
	if (lane == 0)
		A.flags[SF_END] = e;
}

extern "C" __global__ void __launch_bounds__(64) snappy_stream_settle(StreamArgs A)
{
	__shared__ uint16_t ctab[256];
	const uint32_t lane = threadIdx.x, seg = blockIdx.x;
	const uint32_t entry = A.seg_entry[seg];
	uint64_t my_mask = A.tagmask[(size_t)seg * 64 + lane];
	uint32_t my_out = A.winout[(size_t)seg * 64 + lane];
	if (entry == kNoEntry) {
		my_mask = 0;
		my_out = 0;
	} else {
		fill_tag_table(ctab, lane);
		const uint32_t seg_lo = seg * kSegBytes;
		const uint32_t w0 = (entry - seg_lo) >> 6;
		if (lane < w0) {
			my_mask = 0;
			my_out = 0;
		}
		uint64_t pos = entry;
		bool met = false;
		for (uint32_t w = w0; w < 64 && !met; ++w) {
			const uint32_t base = seg_lo + 64 * w;
			if (base >= A.n)
				break;
			const uint32_t wlim = min(64u, A.n - base);
			uint64_t fin = 0;
			uint32_t out = 0;
			if (pos < (uint64_t)base + wlim) {
				const TagAt t = tag_at(A.in, A.n, base + lane, ctab);
				uint64_t leave;
				const uint64_t mine = walk_window(t.esz, lane, (uint32_t)(pos - base), wlim, &leave);
				const uint64_t spec = rdlane64(my_mask, w);
				const uint64_t both = mine & spec;
				fin = mine;
				if (both) {
					/* from their first common tag on, the two parses are one */
					const uint64_t below = (1ull << first_lane(both)) - 1;
					fin = (mine & below) | (spec & ~below);
					met = true;
				}
				out = window_out(fin, t.l, lane);
				pos = base + leave;
			}
			if (lane == w) {
				my_mask = fin;
				my_out = out;
			}
		}
		/* cross-check of the two ways to the segment's exit (chain's and this walk's) */
		const uint32_t leave = met ? A.seg_exit[seg] : pos > 0xffffffffull ? 0xffffffffu : (uint32_t)pos;
		if (lane == 0 && leave != A.seg_leave[seg])
			atomicOr(&A.flags[SF_REFUSED], 1u);
	}
	A.truemask[(size_t)seg * 64 + lane] = my_mask;
	A.trueout[(size_t)seg * 64 + lane] = my_out;
	const uint32_t incl = wave_incl_scan_dpp(min(my_out, kHugeOut));
	if (lane == 63)
		A.seg_out[seg] = incl;
}

/* exclusive sum of seg_out in place: one workgroup, each thread a contiguous run */
extern "C" __global__ void __launch_bounds__(1024) snappy_stream_scan(StreamArgs A)
{
	__shared__ uint64_t part[1024];
	const uint32_t tid = threadIdx.x;
	const uint32_t run = (A.nseg + 1023) / 1024;
	const uint32_t lo = min(tid * run, A.nseg), hi = min(lo + run, A.nseg);
	uint64_t s = 0;
	for (uint32_t k = lo; k < hi; ++k)
		s += A.seg_out[k];
	part[tid] = s;
	__syncthreads();
	if (tid == 0) {
		uint64_t acc = 0;
		for (uint32_t k = 0; k < 1024; ++k) {
			const uint64_t v = part[k];
			part[k] = acc;
			acc += v;
		}
		A.total[0] = acc;
	}
	__syncthreads();
	uint64_t acc = part[tid];
	for (uint32_t k = lo; k < hi; ++k) {
		const uint64_t v = A.seg_out[k];
		A.seg_out[k] = acc;
		acc += v;
	}
}

extern "C" __global__ void __launch_bounds__(64) snappy_stream_bounds(StreamArgs A)
{
	__shared__ uint16_t ctab[256];
	const uint32_t lane = threadIdx.x, seg = blockIdx.x;
	const uint64_t my_mask = A.truemask[(size_t)seg * 64 + lane];
	const uint32_t my_out = min(A.trueout[(size_t)seg * 64 + lane], kHugeOut);
	const uint64_t O = A.seg_out[seg];
	/* an element of 16 MiB is no part of a 32 KiB fragment (and the sums stayed inside 32 bits) */
	if (ballot64(my_out >= kHugeOut) && lane == 0)
		atomicOr(&A.flags[SF_REFUSED], 2u);
	const uint32_t incl = wave_incl_scan_dpp(my_out);
	const uint32_t seg_out = rdlane(incl, 63);
	const uint32_t my_start = incl - my_out; /* output offset of my window inside the segment */
	uint64_t M = (O + (kFragment - 1)) & ~(uint64_t)(kFragment - 1);
	if (M >= O + seg_out)
		return;
	fill_tag_table(ctab, lane);
	for (; M < O + seg_out; M += kFragment) {
		const uint32_t rel = (uint32_t)(M - O);
		const uint64_t holds = ballot64(my_start <= rel && rel - my_start < my_out);
		if (!holds)
			continue; /* cannot happen: the windows tile the segment's output */
		const uint32_t w = first_lane(holds);
		const uint32_t base = seg * kSegBytes + 64 * w;
		const uint64_t mask = rdlane64(my_mask, w);
		const TagAt t = tag_at(A.in, A.n, base + lane, ctab);
		const bool tag = (mask >> lane) & 1;
		const uint32_t mine = tag ? t.l : 0u;
		const uint32_t pre = wave_incl_scan_dpp(mine) - mine;
		const uint64_t hit = ballot64(tag && rdlane(my_start, w) + pre == rel);
		if (hit && lane == 0 && (M >> 15) < A.nfrag)
			A.frag_pos[M >> 15] = base + first_lane(hit);
	}
}

/* Which multiples of 32 KiB of output have an element starting at them: all of them (a csnappy
 * stream, or Snappy 1.0's 32 KiB blocks), or at least every other one (the 64 KiB blocks of later
 * Snappy versions, whose copies stay inside their block just the same)?  SF_GRAIN = 1, 2, or 0. */
extern "C" __global__ void __launch_bounds__(256) snappy_stream_grain(StreamArgs A)
{
	__shared__ uint32_t miss;
	const uint32_t tid = threadIdx.x;
	/* the fragments the parse's output fills (A.nfrag is the host's upper bound).  A parse that
	 * yields nothing, or more than the caller has room for (-3 somewhere), is the slow path's. */
	const uint64_t total = A.total[0];
	const bool fits = total > 0 && total <= A.ulength;
	const uint32_t nfrag = fits ? min((uint32_t)((total + kFragment - 1) / kFragment), A.nfrag) : 0u;
	if (tid == 0)
		miss = 0;
	__syncthreads();
	uint32_t mine = 0;
	for (uint32_t f = tid; f < nfrag; f += 256)
		if (A.frag_pos[f] == kNoEntry)
			mine |= (f & 1) ? 1u : 2u;
	if (mine)
		atomicOr(&miss, mine);
	__syncthreads();
	if (tid == 0) {
		const uint32_t grain = !fits ? 0u : miss == 0 ? 1u : miss == 1 ? 2u : 0u;
		A.flags[SF_GRAIN] = grain;
		A.flags[SF_NFRAG] = nfrag;
		if (!grain)
			atomicOr(&A.flags[SF_REFUSED], 4u); /* an element straddles a block boundary */
	}
}

extern "C" __global__ void __launch_bounds__(256) snappy_stream_describe(StreamArgs A)
{
	const uint32_t f = blockIdx.x * 256 + threadIdx.x;
	if (f >= A.nfrag)
		return;
	const uint32_t grain = A.flags[SF_GRAIN], nfrag = A.flags[SF_NFRAG];
	/* with 64 KiB blocks the even fragments carry two fragments' worth, the odd ones nothing */
	const bool used = grain != 0 && f < nfrag && f % grain == 0;
	const uint32_t pos = used ? A.frag_pos[f] : 0;
	const uint32_t end = !used ? 0 : f + grain < nfrag ? A.frag_pos[f + grain] : A.n;
	const bool ok = used && end >= pos && end <= A.n;
	const uint64_t left = A.total[0] - (uint64_t)f * kFragment; /* (used: f * 32 KiB < total) */
	A.f_in_off[f] = ok ? pos : 0;
	A.f_in_len[f] = ok ? end - pos : 0;
	A.f_out_off[f] = (uint64_t)f * kFragment;
	A.f_out_cap[f] = ok ? (uint32_t)min((uint64_t)grain * kFragment, left) : 0;
	if (used && !ok)
		atomicOr(&A.flags[SF_REFUSED], 4u);
}

extern "C" __global__ void __launch_bounds__(256) snappy_stream_verdict(StreamArgs A)
{
	__shared__ uint32_t bad;
	const uint32_t tid = threadIdx.x;
	if (tid == 0)
		bad = (A.flags[SF_REFUSED] != 0 || A.flags[SF_END] != A.n) ? 1u : 0u;
	__syncthreads();
	uint32_t mine = 0;
	for (uint32_t f = tid; f < A.nfrag; f += 256)
		if (A.f_status[f] != CSNAPPY_E_OK || A.f_produced[f] != A.f_out_cap[f])
			mine = 1;
	if (mine)
		atomicOr(&bad, 1u);
	__syncthreads();
	if (tid == 0) {
		A.flags[SF_VERDICT] = bad ? 0u : 1u;
		if (!bad) {
			A.status[0] = CSNAPPY_E_OK;
			A.produced[0] = (uint32_t)A.total[0];
		}
	}
}

/* descriptors of the one-block slow path; flags; "no boundary found yet" */
extern "C" __global__ void __launch_bounds__(256) snappy_stream_setup(StreamArgs A)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < A.nfrag)
		A.frag_pos[i] = kNoEntry;
	if (i == 0) {
		A.one_off[0] = 0;
		A.one_off[1] = 0;
		A.one_len[0] = A.n;
		A.one_len[1] = A.ulength;
		for (uint32_t k = 0; k < SF_COUNT; ++k)
			A.flags[k] = 0;
		A.total[0] = 0;
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

/* exclusive sum of the compressed lengths (the dense offsets compact_batch wants), in three small
 * launches: per-tile sums (4096 lengths per 256-thread tile), a one-workgroup scan of the tile
 * sums, and the offsets inside every tile */
constexpr uint32_t kScanTile = 4096;

DEVINL uint64_t block_excl_scan64(uint64_t v, uint64_t *wsum, uint64_t *total)
{
	/* 256 threads: wave scans on two 32-bit halves would lose carries; a shared-memory ladder is
	 * plenty for a 256-element scan */
	const uint32_t tid = threadIdx.x;
	wsum[tid] = v;
	__syncthreads();
	for (uint32_t d = 1; d < 256; d <<= 1) {
		const uint64_t add = tid >= d ? wsum[tid - d] : 0;
		__syncthreads();
		wsum[tid] += add;
		__syncthreads();
	}
	*total = wsum[255];
	const uint64_t incl = wsum[tid];
	__syncthreads();
	return incl - v;
}

extern "C" __global__ void __launch_bounds__(256)
snappy_length_tile_sums(const uint32_t *len, uint32_t n, uint64_t *tile_sum)
{
	__shared__ uint64_t wsum[256];
	const uint32_t base = blockIdx.x * kScanTile;
	uint64_t v = 0;
	for (uint32_t k = threadIdx.x; k < kScanTile; k += 256)
		if (base + k < n)
			v += len[base + k];
	uint64_t total;
	(void)block_excl_scan64(v, wsum, &total);
	if (threadIdx.x == 0)
		tile_sum[blockIdx.x] = total;
}

extern "C" __global__ void __launch_bounds__(256)
snappy_length_tile_scan(uint64_t *tile_sum, uint32_t ntiles, uint64_t *total_out)
{
	__shared__ uint64_t wsum[256];
	uint64_t run = 0;
	for (uint32_t b0 = 0; b0 < ntiles; b0 += 256) {
		const uint32_t i = b0 + threadIdx.x;
		const uint64_t v = i < ntiles ? tile_sum[i] : 0;
		uint64_t total;
		const uint64_t ex = block_excl_scan64(v, wsum, &total);
		if (i < ntiles)
			tile_sum[i] = run + ex;
		run += total;
	}
	if (threadIdx.x == 0)
		*total_out = run;
}

extern "C" __global__ void __launch_bounds__(256)
snappy_length_offsets(const uint32_t *len, uint32_t n, const uint64_t *tile_base, uint64_t *off)
{
	__shared__ uint64_t wsum[256];
	const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * (kScanTile / 256); /* 16 consecutive per thread */
	uint64_t v = 0;
	for (uint32_t k = 0; k < kScanTile / 256; ++k)
		if (base + k < n)
			v += len[base + k];
	uint64_t total;
	uint64_t run = tile_base[blockIdx.x] + block_excl_scan64(v, wsum, &total);
	for (uint32_t k = 0; k < kScanTile / 256; ++k)
		if (base + k < n) {
			off[base + k] = run;
			run += len[base + k];
		}
}

/* ==========================================================================================
 * CRC-32C of byte ranges (Snappy framing format, include/csnappy_frame.h): one wave per range.
 * Every lane runs the byte-wise table CRC over its 1/64th of the range; the 64 partial CRCs are
 * combined with crc(A||B) = crc(A) * x^(8|B|) mod P  xor  crc(B)  (the standard combine rule for
 * finalised CRCs; reflected polynomial 0x82F63B78, RFC 3720).
 * ======================================================================================== */
constexpr uint32_t kCrc32cPoly = 0x82F63B78u;

/* a(x) * b(x) mod P, reflected bit order (bit 31 = x^0) */
DEVINL uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
	for (uint32_t m = 0x80000000u; m; m >>= 1) {
		if (a & m)
			p ^= b;
		b = (b & 1u) ? (b >> 1) ^ kCrc32cPoly : b >> 1;
	}
	return p;
}

struct CrcArgs {
	const uint8_t *data;
	const uint64_t *off;
	const uint32_t *len;
	uint32_t *crc;
	uint32_t x2n[32]; /* x^(2^k) mod P */
};

extern "C" __global__ void __launch_bounds__(64) snappy_crc32c_blocks(CrcArgs A)
{
	__shared__ uint32_t T[256];
	const uint32_t lane = threadIdx.x, blk = blockIdx.x;
	for (uint32_t b = lane; b < 256; b += 64) {
		uint32_t c = b;
		for (int k = 0; k < 8; ++k)
			c = (c & 1u) ? (c >> 1) ^ kCrc32cPoly : c >> 1;
		T[b] = c;
	}
	wave_lds_fence();
	const uint8_t *s = A.data + A.off[blk];
	const uint32_t n = A.len[blk];
	const uint32_t seg = (n + 63) / 64;
	const uint32_t lo = min(n, lane * seg), hi = min(n, lo + seg);
	uint32_t c = ~0u;
	for (uint32_t i = lo; i < hi; ++i)
		c = T[(c ^ s[i]) & 0xffu] ^ (c >> 8);
	c = ~c; /* the finalised CRC of my segment (0 for an empty one) */
	/* shift by the bytes behind my segment: x^(8 * after) by square-and-multiply */
	uint32_t after = n - hi, op = 0x80000000u; /* x^0 */
	for (uint32_t k = 3; after; after >>= 1, ++k)
		if (after & 1u)
			op = crc_mulmod(A.x2n[k & 31], op);
	c = crc_mulmod(op, c);
	for (int d = 32; d; d >>= 1)
		c ^= (uint32_t)__shfl_xor((int)c, d, 64);
	if (lane == 0)
		A.crc[blk] = ((c >> 15) | (c << 17)) + 0xa282ead8u; /* masked, framing_format.txt section 3 */
}

/* ==========================================================================================
 * The LDS property the ORD parsers rely on (parse_lean, "TW", and the global-table kernel's exchange):
 * the lanes of ONE LDS instruction that hit one address are served in ascending lane order -- a
 * returning add hands every lane the sum of the LOWER lanes' addends, a returning exchange the value
 * the nearest lower lane put there, and of several stores the highest lane's data stays.  The ISA
 * manual does not promise it; tools/ubench/lds_order.hip measured it on gfx950 (2.5 M lanes in
 * shared slots, no exception), and this probe repeats the measurement on the device in hand before
 * the first parser launch of a process, twice: on an idle device (512 waves), and with the parser's
 * own footprint -- 10 KiB of LDS per workgroup, so sixteen workgroups per CU contend for the LDS,
 * every CU full -- with the parser's own instructions: ds_read_u16, a returning add on a 16-bit
 * half of a dword (other lanes on the other half, carries), its subtraction, ds_write_b16 under a
 * partial exec mask, and ds_wrxchg_rtn on a keyed array; the addresses are drawn like a step's
 * slots (runs: one slot on most lanes; text: a pair now and then; 2-byte strides over the banks).
 * A device that answers differently gets the ORD = false kernels: exact without the property.
 * ======================================================================================== */
constexpr uint32_t kProbeEntries = 5120; /* u16 entries: the dense parser's 10 KiB */
constexpr uint32_t kProbeKeys = 256;     /* the exchange array: the table's last 1 KiB */

extern "C" __global__ void __launch_bounds__(64) snappy_lds_order_probe(uint32_t *bad, uint32_t rounds)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
	uint32_t *tab32 = reinterpret_cast<uint32_t *>(smem);
	constexpr uint32_t kSlots = kProbeEntries - 2 * kProbeKeys;
	uint32_t *keyed = reinterpret_cast<uint32_t *>(smem + 2 * kSlots);
	const uint32_t lane = threadIdx.x;
	uint32_t x = (blockIdx.x * 64 + lane) * 0x9E3779B9u + 0x7f4a7c15u, wrong = 0;
	auto rnd = [&]() {
		x ^= x << 13;
		x ^= x >> 17;
		x ^= x << 5;
		return x;
	};
	for (uint32_t k = lane; k < kSlots / 2; k += 64)
		tab32[k] = rnd() | ((k & 7u) == 0 ? 0xffc0ffc0u : 0u); /* (some halves within 63 of 0xffff: carries) */
	for (uint32_t k = lane; k < kProbeKeys; k += 64)
		keyed[k] = ~0u;
	wave_lds_fence();
	for (uint32_t round = 0; round < rounds; ++round) {
		/* few slots (runs put one slot on most lanes) up to many (text: a pair now and then) */
		const uint32_t sel = (blockIdx.x + round) % 7u;
		const uint32_t range = 1u + sel * sel * sel * 2u; /* 1 .. 433 */
		const uint32_t base = rdlane(rnd(), 0) % (kSlots - 512);
		const uint32_t slot = base + (((rnd() >> 8) * range) >> 24);
		const bool inserted = (rnd() >> 9) & 1u; /* (the commit's exec mask) */
		const uint32_t mine = 0x8000u | (round << 6 & 0x7fc0u) | lane;
		const uint32_t one = 1u << ((slot & 1u) << 4);
		const uint32_t orig32 = tab32[slot >> 1];
		const uint32_t orig16 = tab[slot];
		wave_lds_fence();
		const uint32_t got = atomicAdd(&tab32[slot >> 1], one);
		const uint32_t xgot = atomicExch(&keyed[slot & (kProbeKeys - 1)], (round << 16) | (slot << 6 & 0xffc0u) | lane);
		wave_lds_fence();
		(void)atomicSub(&tab32[slot >> 1], one);
		wave_lds_fence();
		const uint32_t back32 = tab32[slot >> 1];
		wave_lds_fence();
		if (inserted)
			tab[slot] = (uint16_t)mine;
		wave_lds_fence();
		const uint32_t end16 = tab[slot];
		/* what ascending lane order gives */
		uint32_t sum = 0, top = ~0u, near = ~0u, nearslot = 0;
		for (uint32_t j = 0; j < 64; ++j) {
			const uint32_t sj = rdlane(slot, j), ij = rdlane((uint32_t)inserted, j);
			if ((sj >> 1) == (slot >> 1) && j < lane)
				sum += 1u << ((sj & 1u) << 4);
			if (sj == slot && ij)
				top = j;
			if ((sj & (kProbeKeys - 1)) == (slot & (kProbeKeys - 1)) && j < lane) {
				near = j;
				nearslot = sj;
			}
		}
		wrong += got != orig32 + sum;
		wrong += back32 != orig32;
		wrong += end16 != (top == ~0u ? orig16 : (0x8000u | (round << 6 & 0x7fc0u) | top));
		/* the exchange: the nearest lower lane's tag of this round, else something older */
		if (near != ~0u)
			wrong += xgot != ((round << 16) | (nearslot << 6 & 0xffc0u) | near);
		else
			wrong += xgot != ~0u && (xgot >> 16) >= round;
		wave_lds_fence();
	}
	if (wrong)
		atomicAdd(bad, wrong);
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
unsigned long long *g_prof_buf = nullptr;
bool g_timing = false;
Pending g_pending[4096];
int g_npending = 0;
std::mutex g_timing_mu; /* guards g_pending / g_npending (batch calls may come from several threads) */

struct Timer {
	hipStream_t st;
	hipEvent_t a = nullptr, b = nullptr;
	bool on;
	explicit Timer(hipStream_t s) : st(s), on(g_timing) {}
	~Timer()
	{
		/* an error path left between start() and stop(): do not leak the events */
		if (a)
			(void)hipEventDestroy(a);
		if (b)
			(void)hipEventDestroy(b);
	}
	void start()
	{
		if (!on)
			return;
		/* (a stop() that found the pending list full left its pair here) */
		if (a)
			(void)hipEventDestroy(a);
		if (b)
			(void)hipEventDestroy(b);
		a = b = nullptr;
		(void)hipEventCreate(&a);
		(void)hipEventCreate(&b);
		(void)hipEventRecord(a, st);
	}
	void stop(int slot)
	{
		if (!on || !a)
			return;
		(void)hipEventRecord(b, st);
		std::lock_guard<std::mutex> lock(g_timing_mu);
		if (g_npending < 4096) {
			g_pending[g_npending++] = Pending{ slot, a, b };
			a = b = nullptr;
		}
	}
};

constexpr uint32_t kLdsPerCu = 160 * 1024;
constexpr uint32_t kLdsPerWorkgroupMax = kLdsPerCu; /* the most dynamic LDS one workgroup can ask for (gfx950: all of it) */
constexpr uint32_t kChunkFragments = 32768; /* full fragments per GiB of a launch (the unit the workspace is sized in) */
constexpr uint32_t kChunkFragmentsMax = 262144; /* short fragments (pages): as many as make up the same input, at most this per GiB */
constexpr uint32_t kLaunchGibMax = 8;
/* entries of the dense LDS table: 10 KiB = eight of gfx950's 1 280-byte LDS granules, 16 fragments
 * per CU (until round 5, 4 608 entries + 1 KiB of conflict filters).  Fragments with more buckets
 * -- urls.10K's at 4.5-5.5 k -- keep the rest in their HBM spill-over. */
constexpr uint32_t kDenseCapDefault = 5120;
constexpr uint32_t kDenseCap2 = 7168;       /* ... of the second dense launch (14 KiB; 10 per CU) */
constexpr uint32_t kSpillCapDefault = 2048; /* buckets beyond the LDS table kept in HBM (4 KiB per fragment) */
constexpr uint32_t kSampleMinDefault = 700; /* of 2048 sampled positions (text: ~1400, runs: ~300) */
constexpr uint32_t kHashLdsMaxBytes = 8192; /* tables up to this size are indexed by the hash in LDS */

uint32_t frags_per_block(uint32_t max_in_len)
{
	return max_in_len ? (max_in_len + kFragment - 1) / kFragment : 1;
}

/* longest fragment of a batch whose blocks are <= max_in_len */
uint32_t max_fragment(uint32_t max_in_len)
{
	return max_in_len < kFragment ? max_in_len : kFragment;
}

/* dense ids a fragment of <= n bytes can need: a bucket has two or more positions, and the ids count from kFirstBucket */
uint32_t max_ids(uint32_t n)
{
	return (n > 3 ? (n - 3) / 2 : 0) + kFirstBucket;
}

/* records a fragment of <= n bytes can produce: every record but the last holds a copy of >= 4 bytes */
uint32_t record_cap(uint32_t n)
{
	return n / 4 + 8;
}

/* Experiment knobs (environment, read ONCE -- at the first batch call of the process; tests that
 * switch them call the debug entry csnappy_hip_debug_reload_knobs() -- each range-checked; a bad
 * value makes every batch call fail with CSNAPPY_HIP_E_ARG):
 *   CSNAPPY_HIP_TABLE      auto | hash | dense | global   where the hash table lives
 *   CSNAPPY_HIP_DENSE_CAP  256..16384 (multiple of 64)    entries of the dense LDS table
 *   CSNAPPY_HIP_S_ENTRIES  64..4096 (power of two)        global-table kernel: half the keys of its exchange array
 *   CSNAPPY_HIP_WGS_PER_CU 1..32                          cap on fragments in flight per CU
 *   CSNAPPY_HIP_SAMPLE_MIN 0..2048                        dense placement: sampled-distinct threshold
 *                                                         below which a fragment takes the global table
 *   CSNAPPY_HIP_SPILL_CAP  0..8192 (multiple of 64)       dense placement: buckets beyond the LDS table
 *                                                         kept in HBM (0: a second launch with a larger
 *                                                         LDS table takes such fragments instead)
 *   CSNAPPY_HIP_NO_LDS_ORDER 0..1                         1: the parsers that do not rely on the order in which
 *                                                         the LDS serves one instruction's lanes (the ones a
 *                                                         device that fails snappy_lds_order_probe gets)
 *   CSNAPPY_HIP_NO_ISA     0..1                           1: no hand-written step loop: every step takes parse_lean's
 *                                                         compiled C++ (the loops' reference; same bytes, ~10 % slower)
 * (and, read where it is used, once per process, not by the reload entry: CSNAPPY_HIP_DEC_WGS_PER_CU 1..32, a cap on the
 * blocks the decompress kernel keeps in flight per CU -- tools/time_dec.py: 28 / 24 / 20 / 16 cost 14 / 26 / 36 / 44 %) */
struct Knobs {
	int table;      /* -1 auto, else TAB_* */
	uint32_t dense_cap, s_entries, wgs_per_cu, sample_min, spill_cap;
	uint32_t no_lds_order; /* 1: take the ORD = false parsers whatever the probe says */
	uint32_t no_isa;       /* 1: the compiled step loop for every step */
	bool ok;
};

bool knob_u32(const char *name, uint32_t lo, uint32_t hi, uint32_t *out)
{
	const char *e = getenv(name);
	if (!e || !*e)
		return true;
	char *end = nullptr;
	const unsigned long v = strtoul(e, &end, 10);
	if (!end || *end || v < lo || v > hi)
		return false;
	*out = (uint32_t)v;
	return true;
}

Knobs read_knobs()
{
	Knobs k = { -1, 0, 0, 0, kSampleMinDefault, kSpillCapDefault, 0, 0, true };
	if (const char *e = getenv("CSNAPPY_HIP_TABLE")) {
		if (!strcmp(e, "hash"))
			k.table = TAB_LDS_HASH;
		else if (!strcmp(e, "dense"))
			k.table = TAB_LDS_DENSE;
		else if (!strcmp(e, "global"))
			k.table = TAB_GLOBAL;
		else if (strcmp(e, "auto") && *e)
			k.ok = false;
	}
	k.ok = k.ok && knob_u32("CSNAPPY_HIP_DENSE_CAP", 256, 16384, &k.dense_cap) && (k.dense_cap & 63) == 0;
	k.ok = k.ok && knob_u32("CSNAPPY_HIP_SPILL_CAP", 0, 8192, &k.spill_cap) && (k.spill_cap & 63) == 0;
	k.ok = k.ok && knob_u32("CSNAPPY_HIP_S_ENTRIES", 64, 4096, &k.s_entries) &&
	       (k.s_entries & (k.s_entries - 1)) == 0;
	k.ok = k.ok && knob_u32("CSNAPPY_HIP_WGS_PER_CU", 1, 32, &k.wgs_per_cu);
	k.ok = k.ok && knob_u32("CSNAPPY_HIP_SAMPLE_MIN", 0, 2048, &k.sample_min);
	k.ok = k.ok && knob_u32("CSNAPPY_HIP_NO_LDS_ORDER", 0, 1, &k.no_lds_order);
	k.ok = k.ok && knob_u32("CSNAPPY_HIP_NO_ISA", 0, 1, &k.no_isa);
	return k;
}

/* the knobs as the process sees them: read at first use, never again on the launch path */
Knobs g_knobs;
std::once_flag g_knobs_once;

const Knobs &knobs()
{
	std::call_once(g_knobs_once, [] { g_knobs = read_knobs(); });
	return g_knobs;
}

/* Launch geometry of the parser for table power p and fragments of <= maxfrag bytes. */
struct ParsePlan {
	int tab;            /* TAB_* of the first launch */
	uint32_t lds0, lds_bytes, dense_cap, s_entries;
	uint32_t sample_min;
	uint32_t cap2, lds0_2, lds_bytes_2; /* second dense launch (0 = none) */
	uint32_t spill_cap; /* dense: buckets beyond the LDS table, kept in HBM (full fragments only) */
	bool fallback;      /* a TAB_GLOBAL launch follows for fragments the dense table cannot hold */
	uint32_t g_lds0, g_lds_bytes, g_keys; /* global-table kernel: occupancy bitmap, all of its LDS, keys of its exchange array */
};

ParsePlan plan_parse(int p, uint32_t maxfrag, const Knobs &kn)
{
	ParsePlan P;
	memset(&P, 0, sizeof(P));
	const uint32_t slots = 1u << (p - 1);
	/* half the keys of the global-table kernel's exchange array (the LDS placements find slot sharing
	 * through the table itself and have no such array; only a SPILL fragment's lanes in HBM keep a
	 * 128-entry filter, carved out of the table's tail) */
	const uint32_t s_cap = kn.s_entries ? kn.s_entries : 128u;
	/* a fragment of n bytes has at most (n - 3) / 2 buckets of two or more positions (max_ids) */
	uint32_t cap = kn.dense_cap ? kn.dense_cap : kDenseCapDefault;
	uint32_t dense_scratch = 0;
	const uint32_t most = ((maxfrag / 2 + 63) & ~63u) + 64;
	if (!kn.dense_cap && cap > most)
		cap = most;
	if (cap > slots)
		cap = slots;
	/* hash-indexed table in LDS when it is small anyway; the dense one when that is clearly smaller
	 * (its prologue costs ~8 %): 64 KiB blocks at p <= 13 -> hash, 4 KiB pages at p = 13 -> dense
	 * (2 112 entries instead of 4 096: 30 instead of 16 pages per CU, +6 %) */
	{
		const uint32_t scratch0 = (CSNAPPY_PROLOGUE_PAIRS ? 8u : 10u) * (slots >> 5);
		const uint32_t dense_lds0 = 2 * cap > scratch0 ? 2 * cap : scratch0;
		const bool hash_ok = (1u << p) <= kHashLdsMaxBytes && (1u << p) * 10 <= dense_lds0 * 13;
		P.tab = kn.table >= 0 ? kn.table : (hash_ok ? TAB_LDS_HASH : TAB_LDS_DENSE);
	}
	if (P.tab == TAB_LDS_DENSE && cap >= slots)
		P.tab = TAB_LDS_HASH; /* the dense table would be no smaller */
	if (P.tab == TAB_LDS_HASH) {
		P.lds0 = 1u << p; /* (no filters: the LDS placements find slot sharing through the table) */
	} else if (P.tab == TAB_LDS_DENSE) {
		/* small fragments (pages): a table for every fragment there can be (`most`) leaves 25
		 * pages per CU; few pages have more than 0.56 x that many buckets, so the first launch
		 * takes a table of that size (32 pages per CU, the wave limit) and the second one, with
		 * the full table, the pages that overflowed */
		uint32_t cap_full = 0;
		if (!kn.dense_cap && maxfrag < kFragment && cap == most && 2 * most > kLdsPerCu / 32) {
			cap_full = cap;
			cap = ((most * 9 / 16) + 63) & ~63u;
		}
		P.dense_cap = cap;
		/* prologue: two bitmaps + the prefix.  The filters behind the table are set up after the
		 * prologue, so they may lie inside its scratch. */
		const uint32_t scratch = (CSNAPPY_PROLOGUE_PAIRS ? 8u : 10u) * (slots >> 5);
		dense_scratch = scratch;
		P.lds0 = (2 * cap + 15) & ~15u;
		P.fallback = cap < max_ids(maxfrag) || (kn.sample_min && maxfrag == kFragment);
		P.sample_min = kn.sample_min;
		/* fragments with more buckets than the first table get a second try with a larger one
		 * (fewer fragments per CU) before the global table: URL lists sit at 4.5-5.5 k buckets */
		if (cap_full) {
			P.cap2 = cap_full;
			P.fallback = false; /* the second table holds any page */
		} else if (maxfrag == kFragment && kn.spill_cap) {
			/* full fragments: the buckets beyond the LDS table go to HBM in the same launch */
			P.spill_cap = kn.spill_cap;
			P.fallback = cap + P.spill_cap - kSpillFilterSlots < max_ids(maxfrag) || kn.sample_min;
		} else if (!kn.dense_cap && cap == kDenseCapDefault && kDenseCap2 < slots && kDenseCap2 < max_ids(maxfrag)) {
			P.cap2 = kDenseCap2;
		}
		if (P.cap2) {
			P.lds0_2 = (2 * P.cap2 + 15) & ~15u;
			P.lds_bytes_2 = P.lds0_2;
			if (P.lds_bytes_2 < scratch)
				P.lds_bytes_2 = scratch;
		}
	}
	/* global-table geometry (first launch when forced, else the fallback) */
	P.g_lds0 = ((1u << p) >> 4) < 16 ? 16 : (1u << p) >> 4;
	/* (its keyed array: 256 keys = 1 KiB unless the knob says otherwise; powers of two all) */
	P.g_keys = 2 * s_cap < slots ? 2 * s_cap : slots;
	if (P.g_keys < 64)
		P.g_keys = 64;
	P.g_lds_bytes = P.g_lds0 + P.g_keys * 4;
	if (P.tab == TAB_GLOBAL) {
		P.lds0 = P.g_lds0;
		P.s_entries = P.g_keys;
	}
	P.lds_bytes = P.lds0 + P.s_entries * 4;
	if (P.tab == TAB_LDS_DENSE && P.lds_bytes < dense_scratch)
		P.lds_bytes = dense_scratch;
	if (kn.wgs_per_cu) {
		/* experiments: cap the fragments per CU by padding the LDS request */
		const uint32_t pad = (kLdsPerCu / kn.wgs_per_cu) & ~255u;
		if (pad > P.lds_bytes)
			P.lds_bytes = pad;
		if (pad > P.g_lds_bytes)
			P.g_lds_bytes = pad;
		if (P.cap2 && pad > P.lds_bytes_2)
			P.lds_bytes_2 = pad;
	}
	return P;
}

/* bytes per fragment of the `tabs` workspace region: 2-byte ids for every position, or the
 * 2^p-byte global table (p is not known when the workspace is sized: 64 KiB) */
uint32_t tab_stride_for(uint32_t maxfrag, const Knobs &kn)
{
	const uint32_t ids = ((maxfrag + 63) & ~63u) * 2;
	const uint32_t cap = kn.dense_cap ? kn.dense_cap : kDenseCapDefault;
	const bool global_possible = kn.table == TAB_GLOBAL || cap < max_ids(maxfrag) ||
				     (kn.sample_min && maxfrag == kFragment);
	const uint32_t spill = maxfrag == kFragment ? kn.spill_cap * 2 : 0; /* behind the ids / the global table */
	return (global_possible ? 65536u : (ids < 1024 ? 1024u : ids)) + spill;
}

struct Workspace {
	uint64_t cnt_bytes, rec_bytes, tab_bytes, total;
	uint32_t chunk_blocks, chunk_frags, rec_cap, tab_stride;
};

/* launch_gib: GiB of input one parser launch covers (1..kLaunchGibMax); 0: the floor -- launches of
 * kChunkFragments fragments whatever their size (1 GiB of full fragments, 128 MiB of 4 KiB pages: what
 * a caller with a small pooled scratch can afford; csnappy_hip_compress_workspace_size) */
Workspace plan_workspace(uint32_t nblocks, uint32_t max_in_len, const Knobs &kn, uint32_t launch_gib = 1)
{
	Workspace W;
	const uint32_t fpb = frags_per_block(max_in_len);
	/* a launch covers launch_gib GiB of input whatever the fragment size: a launch of 32 768 pages
	 * (128 MiB) is 0.5 ms long and a tenth of that is its ramp and tail, and the ramp and tail of a
	 * 1 GiB launch of full fragments are 5 % (text) to 11 % (runs) of it */
	const uint32_t mf = max_fragment(max_in_len) ? max_fragment(max_in_len) : 1;
	uint64_t cf = (uint64_t)kChunkFragments * kFragment / mf;
	if (cf > kChunkFragmentsMax)
		cf = kChunkFragmentsMax;
	cf = launch_gib ? cf * launch_gib : kChunkFragments;
	uint64_t cb = cf / fpb;
	if (cb < 1)
		cb = 1;
	if (cb > nblocks)
		cb = nblocks;
	W.chunk_blocks = (uint32_t)cb;
	W.chunk_frags = (uint32_t)cb * fpb;
	W.rec_cap = record_cap(max_fragment(max_in_len));
	W.tab_stride = tab_stride_for(max_fragment(max_in_len), kn);
	W.cnt_bytes = ((uint64_t)W.chunk_frags * 4 + 255) & ~255ull;
	W.rec_bytes = ((uint64_t)W.chunk_frags * W.rec_cap * 8 + 255) & ~255ull;
	W.tab_bytes = (uint64_t)W.chunk_frags * W.tab_stride;
	W.total = W.cnt_bytes + W.rec_bytes + W.tab_bytes + 65536;
	return W;
}


/* Does the LDS of the current device serve one instruction's lanes in ascending order
 * (snappy_lds_order_probe)?  Asked once per device and process, on a stream of its own: the first
 * compress call on a device waits for two small launches and a 4-byte copy (about a millisecond; the
 * caller's stream is not touched).  *ordered = 1 / 0; a negative return is a HIP failure, and the
 * question is asked again by the next call. */
int lds_order_probed(int *ordered)
{
	static std::mutex mu;
	static int verdict[64]; /* 0 unknown, 1 ordered, -1 not */
	int dev = 0;
	if (!hip_ok(hipGetDevice(&dev), "hipGetDevice") || dev < 0 || dev >= 64)
		return CSNAPPY_HIP_E_RUNTIME;
	std::lock_guard<std::mutex> lock(mu);
	if (verdict[dev] == 0) {
		hipStream_t ps = nullptr;
		uint32_t *d_bad = nullptr, h_bad = 1;
		hipDeviceProp_t prop;
		if (!hip_ok(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties (LDS order probe)") ||
		    !hip_ok(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking), "hipStreamCreate (LDS order probe)"))
			return CSNAPPY_HIP_E_RUNTIME;
		bool ok = hip_ok(hipMalloc(&d_bad, 4), "hipMalloc (LDS order probe)") &&
			  hip_ok(hipMemsetAsync(d_bad, 0, 4, ps), "hipMemsetAsync (LDS order probe)");
		if (ok) {
			/* idle device: 512 waves; then every CU full of 10 KiB workgroups, twice over */
			const uint32_t cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
			hipLaunchKernelGGL(snappy_lds_order_probe, dim3(512), dim3(64), 2 * kProbeEntries, ps, d_bad, 32u);
			hipLaunchKernelGGL(snappy_lds_order_probe, dim3(cus * 32), dim3(64), 2 * kProbeEntries, ps, d_bad, 48u);
			ok = hip_ok(hipGetLastError(), "launch snappy_lds_order_probe") &&
			     hip_ok(hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, ps), "hipMemcpyAsync (LDS order probe)") &&
			     hip_ok(hipStreamSynchronize(ps), "hipStreamSynchronize (LDS order probe)");
		}
		if (d_bad)
			(void)hipFree(d_bad);
		(void)hipStreamDestroy(ps);
		if (!ok)
			return CSNAPPY_HIP_E_RUNTIME;
		verdict[dev] = h_bad == 0 ? 1 : -1;
	}
	*ordered = verdict[dev] > 0;
	return 0;
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

/* debug only (not in the public header): 24 x u64 device buffer that receives the parser's
 * s_memtime phase counters; NULL switches back to the production kernel */
void csnappy_hip_debug_set_profile_buffer(void *d_buf)
{
	g_prof_buf = static_cast<unsigned long long *>(d_buf);
}

#if CSNAPPY_EMIT_PROF
/* development builds only: read (and clear) the emit kernel's phase counters */
int csnappy_hip_debug_emit_prof(unsigned long long *out16)
{
	static const unsigned long long zero[16] = { 0 };
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_emit_prof), sizeof(zero)) != hipSuccess)
		return -1;
	return hipMemcpyToSymbol(HIP_SYMBOL(g_emit_prof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

#if CSNAPPY_ISA_PROF
/* development builds only: read (and clear) the dense step loop's phase counters */
extern "C" int csnappy_hip_debug_isa_prof(unsigned long long *out16)
{
	static const unsigned long long zero[16] = { 0 };
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_isa_prof), sizeof(zero)) != hipSuccess)
		return -1;
	return hipMemcpyToSymbol(HIP_SYMBOL(g_isa_prof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

#if CSNAPPY_DEC_PROF
/* development builds only: read (and clear) the decompress kernel's phase counters */
int csnappy_hip_debug_dec_prof(unsigned long long *out32)
{
	static const unsigned long long zero[32] = { 0 };
	if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_dec_prof), sizeof(zero)) != hipSuccess)
		return -1;
	return hipMemcpyToSymbol(HIP_SYMBOL(g_dec_prof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

/* debug only (not in the public header): read the CSNAPPY_HIP_* environment knobs again.  For
 * tests that switch table placements inside one process; not to be called while a batch call runs. */
void csnappy_hip_debug_reload_knobs(void)
{
	(void)knobs();
	g_knobs = read_knobs();
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
	std::lock_guard<std::mutex> lock(g_timing_mu);
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
	if (nblocks == 0)
		return 65536;
	return (size_t)plan_workspace(nblocks, max_in_len, knobs(), 0).total;
}

size_t csnappy_hip_compress_workspace_size_for(uint32_t nblocks, uint32_t max_in_len, uint32_t launch_gib)
{
	if (nblocks == 0)
		return 65536;
	/* (0: the floor, csnappy_hip_compress_workspace_size) */
	if (launch_gib > kLaunchGibMax)
		launch_gib = kLaunchGibMax;
	return (size_t)plan_workspace(nblocks, max_in_len, knobs(), launch_gib).total;
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
	const Knobs kn = knobs();
	if (!kn.ok)
		return CSNAPPY_HIP_E_ARG;
	if (nblocks == 0)
		return 0;
	/* the largest launches the caller's workspace has room for (csnappy_hip_compress_workspace_size_for);
	 * csnappy_hip_compress_workspace_size() -- launches of 32 768 fragments whatever their size -- is the
	 * least that is accepted */
	Workspace W = plan_workspace(nblocks, max_in_len, kn, 0);
	if (workspace_bytes < W.total || (reinterpret_cast<uintptr_t>(d_workspace) & 255))
		return CSNAPPY_HIP_E_WORKSPACE;
	for (uint32_t g = kLaunchGibMax; g >= 1; --g) {
		const Workspace Wg = plan_workspace(nblocks, max_in_len, kn, g);
		if (Wg.total <= workspace_bytes) {
			W = Wg;
			break;
		}
	}
	const uint32_t fpb = frags_per_block(max_in_len);
	if ((uint64_t)nblocks * fpb > 0x7fffffffull)
		return CSNAPPY_HIP_E_ARG;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const ParsePlan P = plan_parse(p, max_fragment(max_in_len), kn);
	/* every placement relies on the order in which the LDS serves one instruction's lanes (the tables
	 * in LDS through their returning add and their stores, the global table through its returning
	 * exchange): a device that does not keep it, or CSNAPPY_HIP_NO_LDS_ORDER=1, gets the parsers
	 * that do without -- same bytes, several times slower */
	int ordered = 0;
	if (!kn.no_lds_order) {
		const int rc = lds_order_probed(&ordered);
		if (rc)
			return rc;
	}

	CompressArgs A;
	A.in = static_cast<const uint8_t *>(d_in);
	A.in_off = d_in_off;
	A.in_len = d_in_len;
	A.out = static_cast<uint8_t *>(d_out);
	A.out_off = d_out_off;
	A.out_len = d_out_len;
	uint8_t *ws = static_cast<uint8_t *>(d_workspace);
	A.rec_cnt = reinterpret_cast<uint32_t *>(ws);
	A.recs = reinterpret_cast<uint64_t *>(ws + W.cnt_bytes);
	{
		/* ids / tables: 64 KiB aligned */
		uintptr_t t = reinterpret_cast<uintptr_t>(ws + W.cnt_bytes + W.rec_bytes);
		A.tabs = reinterpret_cast<uint8_t *>((t + 65535) & ~(uintptr_t)65535);
	}
	A.prof = g_prof_buf;
	A.fpb = fpb;
	A.rec_cap = W.rec_cap;
	A.tab_stride = W.tab_stride;
	A.dense_cap = P.dense_cap;
	A.spill_cap = P.spill_cap;
	A.spill_off = W.tab_stride - P.spill_cap * 2;
	A.max_in_len = max_in_len;
	A.sample_min = P.sample_min;
	A.no_isa = kn.no_isa;
	A.p = p;
	A.mode = mode;

	/* (the s_memtime phase counters of tools/phase_lean.py exist for the dense and the global placement) */
	const void *k2 = !ordered ? reinterpret_cast<const void *>(snappy_parse_fragments_gtab_unordered)
			 : g_prof_buf ? reinterpret_cast<const void *>(snappy_parse_fragments_gtab_prof)
				      : reinterpret_cast<const void *>(snappy_parse_fragments_gtab);
	const void *k1 = P.tab == TAB_LDS_DENSE
				 ? (!ordered ? reinterpret_cast<const void *>(snappy_parse_fragments_dense_lean_unordered)
				    : g_prof_buf ? reinterpret_cast<const void *>(snappy_parse_fragments_dense_lean_prof)
						 : reinterpret_cast<const void *>(snappy_parse_fragments_dense_lean))
			 : P.tab == TAB_LDS_HASH
				 ? (!ordered ? reinterpret_cast<const void *>(snappy_parse_fragments_hash_lean_unordered)
					     : reinterpret_cast<const void *>(snappy_parse_fragments_hash_lean))
				 : k2;
	{
		/* the kernels' dynamic-LDS limit is an attribute per kernel and DEVICE: raised once per device
		 * to the most a workgroup of that device can have, never per call (a call with a small table
		 * must not lower it under another thread's launch) */
		static std::mutex mu;
		static int lds_limit[64]; /* per device: 0 not set yet, -1 failed, else the limit in bytes */
		int dev = 0;
		if (!hip_ok(hipGetDevice(&dev), "hipGetDevice") || dev < 0 || dev >= 64)
			return CSNAPPY_HIP_E_RUNTIME;
		int limit;
		{
			std::lock_guard<std::mutex> lock(mu);
			if (lds_limit[dev] == 0) {
				const void *ks[] = { reinterpret_cast<const void *>(snappy_parse_fragments_dense_lean),
						     reinterpret_cast<const void *>(snappy_parse_fragments_dense_lean_prof),
						     reinterpret_cast<const void *>(snappy_parse_fragments_hash_lean),
						     reinterpret_cast<const void *>(snappy_parse_fragments_gtab),
						     reinterpret_cast<const void *>(snappy_parse_fragments_gtab_prof),
						     reinterpret_cast<const void *>(snappy_parse_fragments_dense_lean_unordered),
						     reinterpret_cast<const void *>(snappy_parse_fragments_hash_lean_unordered),
						     reinterpret_cast<const void *>(snappy_parse_fragments_gtab_unordered) };
				int optin = 0;
				bool ok = hip_ok(hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev),
						 "hipDeviceGetAttribute (LDS per workgroup)");
				if (ok && (optin <= 0 || optin > (int)kLdsPerWorkgroupMax))
					optin = (int)kLdsPerWorkgroupMax;
				for (const void *k : ks)
					ok = ok && hip_ok(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, optin),
							  "hipFuncSetAttribute");
				lds_limit[dev] = ok ? optin : -1;
			}
			limit = lds_limit[dev];
			if (limit < 0)
				lds_limit[dev] = 0; /* (a transient failure is tried again by the next call) */
		}
		if (limit < 0)
			return CSNAPPY_HIP_E_RUNTIME;
		if (P.lds_bytes > (uint32_t)limit || P.g_lds_bytes > (uint32_t)limit || P.lds_bytes_2 > (uint32_t)limit)
			return CSNAPPY_HIP_E_ARG;
	}

	for (uint32_t b0 = 0; b0 < nblocks; b0 += W.chunk_blocks) {
		const uint32_t nb = nblocks - b0 < W.chunk_blocks ? nblocks - b0 : W.chunk_blocks;
		A.blk_base = b0;
		void *args[] = { &A };
		Timer t(st);
		t.start();
		A.lds0 = P.lds0;
		A.s_entries = P.s_entries;
		A.only_unparsed = 0;
		if (!hip_ok(hipLaunchKernel(k1, dim3(nb * fpb), dim3(64), args, P.lds_bytes, st),
			    "launch snappy_parse_fragments"))
			return CSNAPPY_HIP_E_RUNTIME;
		if (P.cap2) {
			A.lds0 = P.lds0_2;
			A.dense_cap = P.cap2;
			A.sample_min = 0;
			A.only_unparsed = 1;
			if (!hip_ok(hipLaunchKernel(k1, dim3(nb * fpb), dim3(64), args, P.lds_bytes_2, st),
				    "launch snappy_parse_fragments (second table size)"))
				return CSNAPPY_HIP_E_RUNTIME;
			A.dense_cap = P.dense_cap;
			A.sample_min = P.sample_min;
		}
		if (P.fallback) {
			A.lds0 = P.g_lds0;
			A.s_entries = P.g_keys;
			A.only_unparsed = 1;
			if (!hip_ok(hipLaunchKernel(k2, dim3(nb * fpb), dim3(64), args, P.g_lds_bytes, st),
				    "launch snappy_parse_fragments_gtab"))
				return CSNAPPY_HIP_E_RUNTIME;
		}
		t.stop(0);
		t.start();
		A.emit_wave_per_block = fpb == 1 && max_in_len <= 8192;
		A.emit_blocks = nb;
		if (A.emit_wave_per_block)
			hipLaunchKernelGGL(snappy_emit_pages, dim3((nb + kEmitWaves - 1) / kEmitWaves), dim3(64 * kEmitWaves), 0,
					   st, A);
		else {
			hipLaunchKernelGGL(snappy_emit_sizes, dim3(nb * fpb), dim3(64 * kSizesWaves), 0, st, A);
			hipLaunchKernelGGL(snappy_emit_bases, dim3(nb), dim3(64), 0, st, A);
			hipLaunchKernelGGL(snappy_emit_blocks, dim3(nb * fpb), dim3(64 * kEmitWaves), 0, st, A);
		}
		t.stop(1);
		if (!hip_ok(hipGetLastError(), "launch snappy_emit_blocks"))
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
	A.skip_if = nullptr;
	Timer t(st);
	t.start();
	/* (experiments: CSNAPPY_HIP_DEC_WGS_PER_CU caps the blocks in flight per CU by padding the LDS request) */
	static const uint32_t dec_pad = [] {
		const char *e = getenv("CSNAPPY_HIP_DEC_WGS_PER_CU");
		const unsigned long v = e && *e ? strtoul(e, nullptr, 10) : 0;
		return v >= 1 && v <= 32 ? (uint32_t)((kLdsPerCu / v) & ~255u) : 0u;
	}();
	const uint32_t dec_static = 512 + 2048 + kOutStage + 16 + 64;
	hipLaunchKernelGGL(snappy_decompress_blocks, dim3(nblocks), dim3(64), dec_pad > dec_static ? dec_pad - dec_static : 0, st, A);
	t.stop(2);
	if (!hip_ok(hipGetLastError(), "launch snappy_decompress_blocks"))
		return CSNAPPY_HIP_E_RUNTIME;
	return 0;
}

namespace {
/* workspace of the stream call: index arrays, fragment descriptors, slow-path descriptor, flags */
struct StreamPlan {
	StreamArgs A;
	size_t total;
	bool indexed; /* false: lengths for which the index is not defined, one-wave decode only */
};

StreamPlan plan_stream(uint32_t n, uint32_t ulength, uint8_t *ws)
{
	StreamPlan P;
	memset(&P.A, 0, sizeof(P.A));
	P.indexed = n > 0 && ulength > 0 && n < 0xffff0000u && ulength < 0xffff0000u;
	const size_t nseg = P.indexed ? ((size_t)n + kSegBytes - 1) / kSegBytes : 0;
	/* fragments: as many as the room allows, or as the body can fill (a 3-byte copy yields 64 bytes) */
	const uint64_t can_fill = (uint64_t)n * 22 + 64;
	const size_t nfrag = P.indexed ? (size_t)(((ulength < can_fill ? ulength : can_fill) + kFragment - 1) / kFragment) : 0;
	size_t at = 0;
	auto take = [&](size_t bytes) {
		uint8_t *p = ws + at;
		at += (bytes + 15) & ~(size_t)15;
		return p;
	};
	P.A.n = n;
	P.A.ulength = ulength;
	P.A.nseg = (uint32_t)nseg;
	P.A.nfrag = (uint32_t)nfrag;
	P.A.tagmask = reinterpret_cast<uint64_t *>(take(nseg * 64 * 8));
	P.A.winout = reinterpret_cast<uint32_t *>(take(nseg * 64 * 4));
	P.A.truemask = reinterpret_cast<uint64_t *>(take(nseg * 64 * 8));
	P.A.trueout = reinterpret_cast<uint32_t *>(take(nseg * 64 * 4));
	P.A.seg_out = reinterpret_cast<uint64_t *>(take(nseg * 8));
	P.A.seg_exit = reinterpret_cast<uint32_t *>(take(nseg * 4));
	P.A.seg_xesz = reinterpret_cast<uint32_t *>(take(nseg * 4));
	P.A.seg_safe = reinterpret_cast<uint32_t *>(take(nseg * 4));
	P.A.seg_leave = reinterpret_cast<uint32_t *>(take(nseg * 4));
	P.A.last_tag = reinterpret_cast<uint16_t *>(take(nseg * kSegBytes * 2));
	P.A.seg_entry = reinterpret_cast<uint32_t *>(take(nseg * 4));
	P.A.grp_e = reinterpret_cast<uint32_t *>(take(((nseg + 63) / 64) * 4));
	P.A.grp_esz = reinterpret_cast<uint32_t *>(take(((nseg + 63) / 64) * 4));
	P.A.f_in_off = reinterpret_cast<uint64_t *>(take(nfrag * 8));
	P.A.f_out_off = reinterpret_cast<uint64_t *>(take(nfrag * 8));
	P.A.frag_pos = reinterpret_cast<uint32_t *>(take(nfrag * 4));
	P.A.f_in_len = reinterpret_cast<uint32_t *>(take(nfrag * 4));
	P.A.f_out_cap = reinterpret_cast<uint32_t *>(take(nfrag * 4));
	P.A.f_produced = reinterpret_cast<uint32_t *>(take(nfrag * 4));
	P.A.f_status = reinterpret_cast<int32_t *>(take(nfrag * 4));
	P.A.one_off = reinterpret_cast<uint64_t *>(take(16));
	P.A.total = reinterpret_cast<uint64_t *>(take(8));
	P.A.one_len = reinterpret_cast<uint32_t *>(take(8));
	P.A.flags = reinterpret_cast<uint32_t *>(take(4 * SF_COUNT));
	P.total = at;
	return P;
}
} // namespace

size_t csnappy_hip_decompress_stream_workspace_size(uint32_t in_len, uint32_t ulength)
{
	return plan_stream(in_len, ulength, nullptr).total;
}

int csnappy_hip_decompress_stream(const void *d_in, uint32_t in_len, uint32_t ulength, void *d_out,
				  int32_t *d_status, uint32_t *d_produced, void *d_workspace,
				  size_t workspace_bytes, void *stream)
{
	StreamPlan P = plan_stream(in_len, ulength, static_cast<uint8_t *>(d_workspace));
	if (!d_workspace || workspace_bytes < P.total || (reinterpret_cast<uintptr_t>(d_workspace) & 15))
		return CSNAPPY_HIP_E_ARG;
	hipStream_t st = static_cast<hipStream_t>(stream);
	StreamArgs &S = P.A;
	S.in = static_cast<const uint8_t *>(d_in);
	S.status = d_status;
	S.produced = d_produced;
	DecompressArgs D;
	D.in = S.in;
	D.out = static_cast<uint8_t *>(d_out);
	D.mode = CSNAPPY_HIP_FRAGMENT;
	{
		Timer t(st);
		t.start();
		hipLaunchKernelGGL(snappy_stream_setup, dim3(S.nfrag / 256 + 1), dim3(256), 0, st, S);
		if (P.indexed) {
			hipLaunchKernelGGL(snappy_stream_index, dim3(S.nseg), dim3(64), 0, st, S);
			hipLaunchKernelGGL(snappy_stream_chain_groups, dim3((S.nseg + 63) / 64), dim3(64), 0, st, S);
			hipLaunchKernelGGL(snappy_stream_chain_link, dim3(1), dim3(64), 0, st, S);
			hipLaunchKernelGGL(snappy_stream_settle, dim3(S.nseg), dim3(64), 0, st, S);
			hipLaunchKernelGGL(snappy_stream_scan, dim3(1), dim3(1024), 0, st, S);
			hipLaunchKernelGGL(snappy_stream_bounds, dim3(S.nseg), dim3(64), 0, st, S);
			hipLaunchKernelGGL(snappy_stream_grain, dim3(1), dim3(256), 0, st, S);
			hipLaunchKernelGGL(snappy_stream_describe, dim3((S.nfrag + 255) / 256), dim3(256), 0, st, S);
		}
		t.stop(3);
	}
	if (P.indexed) {
		/* the fragments, as independent no-header blocks */
		D.in_off = S.f_in_off;
		D.in_len = S.f_in_len;
		D.out_off = S.f_out_off;
		D.out_cap = S.f_out_cap;
		D.status = S.f_status;
		D.produced = S.f_produced;
		D.nblocks = S.nfrag;
		D.skip_if = nullptr;
		Timer t(st);
		t.start();
		hipLaunchKernelGGL(snappy_decompress_blocks, dim3(S.nfrag), dim3(64), 0, st, D);
		hipLaunchKernelGGL(snappy_stream_verdict, dim3(1), dim3(256), 0, st, S);
		t.stop(2);
	}
	/* the whole body with one wave, unless the verdict let the fragments stand */
	D.in_off = S.one_off;
	D.in_len = S.one_len;
	D.out_off = S.one_off + 1;
	D.out_cap = S.one_len + 1;
	D.status = d_status;
	D.produced = d_produced;
	D.nblocks = 1;
	D.skip_if = S.flags + SF_VERDICT;
	{
		Timer t(st);
		t.start();
		hipLaunchKernelGGL(snappy_decompress_blocks, dim3(1), dim3(64), 0, st, D);
		t.stop(2);
	}
	if (!hip_ok(hipGetLastError(), "launch stream decompress"))
		return CSNAPPY_HIP_E_RUNTIME;
	return 0;
}

int csnappy_hip_decompress_stream_took_fast_path(const void *d_workspace, uint32_t in_len, uint32_t ulength,
						  void *stream)
{
	StreamPlan P = plan_stream(in_len, ulength, static_cast<uint8_t *>(const_cast<void *>(d_workspace)));
	uint32_t v = 0;
	if (!hip_ok(hipMemcpyAsync(&v, P.A.flags + SF_VERDICT, 4, hipMemcpyDeviceToHost,
				   static_cast<hipStream_t>(stream)), "read verdict") ||
	    !hip_ok(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "sync"))
		return CSNAPPY_HIP_E_RUNTIME;
	return (int)v;
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

size_t csnappy_hip_dense_offsets_workspace_size(uint32_t nblocks)
{
	return ((size_t)(nblocks + kScanTile - 1) / kScanTile + 1) * 8;
}

int csnappy_hip_dense_offsets(const uint32_t *d_out_len, uint32_t nblocks, uint64_t *d_dense_off, uint64_t *d_total,
			      void *d_workspace, size_t workspace_bytes, void *stream)
{
	if (workspace_bytes < csnappy_hip_dense_offsets_workspace_size(nblocks) ||
	    (reinterpret_cast<uintptr_t>(d_workspace) & 7))
		return CSNAPPY_HIP_E_WORKSPACE;
	hipStream_t st = static_cast<hipStream_t>(stream);
	uint64_t *tiles = static_cast<uint64_t *>(d_workspace);
	const uint32_t ntiles = (nblocks + kScanTile - 1) / kScanTile;
	if (ntiles)
		hipLaunchKernelGGL(snappy_length_tile_sums, dim3(ntiles), dim3(256), 0, st, d_out_len, nblocks, tiles);
	hipLaunchKernelGGL(snappy_length_tile_scan, dim3(1), dim3(256), 0, st, tiles, ntiles, d_total);
	if (ntiles)
		hipLaunchKernelGGL(snappy_length_offsets, dim3(ntiles), dim3(256), 0, st, d_out_len, nblocks, tiles,
				   d_dense_off);
	if (!hip_ok(hipGetLastError(), "launch snappy_length_*"))
		return CSNAPPY_HIP_E_RUNTIME;
	return 0;
}

int csnappy_hip_crc32c_batch(const void *d_data, const uint64_t *d_off, const uint32_t *d_len,
			     uint32_t nblocks, uint32_t *d_crc, void *stream)
{
	if (nblocks == 0)
		return 0;
	CrcArgs A;
	A.data = static_cast<const uint8_t *>(d_data);
	A.off = d_off;
	A.len = d_len;
	A.crc = d_crc;
	/* x^(2^k) mod P by repeated squaring on the host (reflected: bit 31 = x^0, bit 30 = x^1) */
	auto mulmod = [](uint32_t a, uint32_t b) {
		uint32_t p = 0;
		for (uint32_t m = 0x80000000u; m; m >>= 1) {
			if (a & m)
				p ^= b;
			b = (b & 1u) ? (b >> 1) ^ 0x82F63B78u : b >> 1;
		}
		return p;
	};
	uint32_t v = 0x40000000u;
	for (int k = 0; k < 32; ++k) {
		A.x2n[k] = v;
		v = mulmod(v, v);
	}
	hipLaunchKernelGGL(snappy_crc32c_blocks, dim3(nblocks), dim3(64), 0, static_cast<hipStream_t>(stream), A);
	if (!hip_ok(hipGetLastError(), "launch snappy_crc32c_blocks"))
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
