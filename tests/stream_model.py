"""Model of the stream index (TEST INFRASTRUCTURE): how `csnappy_hip_decompress_stream`
(csnappy_amd/csrc/csnappy_kernels.hip, kernels snappy_stream_*) finds the fragments of one long
Snappy body without decoding it.

The kernels cut the body into 4 KiB segments and parse every segment speculatively from its first
byte; the claim is that the fragment boundaries they derive are those of the true, sequential
parse of csnappy_decompress_noheader (csnappy_decompress.c:319-387).  This file restates the
kernels' steps in numpy / plain Python, segment by segment, so that the claim can be checked on
the CPU against a straightforward sequential walk (`sequential_parse`), for every kind of stream
the GPU tests use.  It models the ALGORITHM (speculative parse per segment, the per-byte
"last tag" table by pointer doubling, the safe prefix, the exact chain, settling the windows,
boundaries at multiples of 32 KiB, 32 / 64 KiB grain); the HIP code is a SIMT transcription.
"""
import numpy as np

SEG = 4096          # kSegBytes
FRAG = 32768        # kFragment
NO_ENTRY = 0xFFFFFFFF
HUGE = 1 << 24      # kHugeOut


def tags_at(body):
    """For EVERY byte of the body read as a tag: (input bytes the element takes, bytes it yields).
    tag_at() of the kernels: header bytes beyond the input read as 0; a literal length that wraps 32
    bits takes 0xfffffff8 input bytes (it leaves the input whatever follows)."""
    n = len(body)
    b = np.frombuffer(body, dtype=np.uint8).astype(np.uint64)
    pad = np.concatenate([b, np.zeros(8, dtype=np.uint64)])
    tag = pad[:n]
    kind = tag & 3
    up = tag >> 2
    tr = pad[1:n + 1] | (pad[2:n + 2] << 8) | (pad[3:n + 3] << 16) | (pad[4:n + 4] << 24)
    lit_extra = np.where(up >= 60, up - 59, 0)                 # 1..4 length bytes behind the tag
    extra = np.where(kind == 0, lit_extra, np.array([0, 1, 2, 4], dtype=np.uint64)[kind])
    mask = (np.uint64(1) << (np.uint64(8) * extra)) - np.uint64(1)
    trm = tr & mask
    lit_len = np.where(lit_extra > 0, (trm + 1) & 0xFFFFFFFF, up + 1)
    copy_len = np.where(kind == 1, 4 + (up & 7), up + 1)
    l = np.where(kind == 0, lit_len, copy_len)
    hsz = 1 + extra
    esz = np.where(kind == 0, np.where(l >= 0xFFFFFFF0, 0xFFFFFFF8, hsz + l), hsz)
    return esz.astype(np.int64), np.minimum(l, HUGE).astype(np.int64)


def sequential_parse(body):
    """Ground truth: the tag chain from byte 0 -> (element positions, output offset of each, end)."""
    esz, l = tags_at(body)
    n, p, out = len(body), 0, 0
    pos, outs = [], []
    while p < n:
        pos.append(p)
        outs.append(out)
        out += int(l[p])
        p += int(esz[p])
    return pos, outs, p, out


def index_segment(esz, l, n, seg):
    """snappy_stream_index for one segment: the speculative parse from the segment's first byte
    (set of tags, where it leaves), the last-tag table by pointer doubling, the safe prefix."""
    lo = seg * SEG
    seg_len = min(SEG, n - lo)
    tags, p = [], lo
    while p < lo + seg_len:
        tags.append(p)
        p += int(esz[p])
    leave = min(p, 0xFFFFFFFF)
    nxt = np.arange(SEG)
    rel = np.arange(seg_len)
    step = rel + esz[lo:lo + seg_len]
    nxt[:seg_len] = np.where(step < seg_len, step, rel)      # a tag that leaves the segment points at itself
    last = nxt.copy()
    for _ in range(12):                                        # elements take >= 2 bytes: 2^11 hops at most
        last = last[last]
    off = np.flatnonzero(last != last[0])
    safe = int(off[0]) if len(off) else SEG
    return set(tags), leave, last, safe


def build_index(body):
    esz, l = tags_at(body)
    n = len(body)
    nseg = (n + SEG - 1) // SEG
    segs = [index_segment(esz, l, n, k) for k in range(nseg)]
    return esz, l, segs


def chain(body, esz, segs):
    """The chain, sequentially (what snappy_stream_chain_groups / _link must reproduce): where the
    true parse enters and leaves every segment."""
    n = len(body)
    e = 0
    entry, leave = [], []
    for k, (spec, spec_leave, last, safe) in enumerate(segs):
        lo, hi = k * SEG, min((k + 1) * SEG, n)
        if e >= hi:
            entry.append(NO_ENTRY)
            leave.append(0)
            continue
        entry.append(e)
        after = e + int(esz[e]) if e < n else e + 2
        if after >= hi:
            e = min(after, 0xFFFFFFFF)                         # a long element: followed on the spot
        elif e - lo < safe:
            e = spec_leave                                     # leaves with the speculative parse
        else:
            lt = lo + int(last[e - lo])                        # the table knows this parse's last tag
            e = min(lt + int(esz[lt]), 0xFFFFFFFF)
        leave.append(e)
    return entry, leave, e


def chain_step(body, esz, seg, k, e):
    """one segment of the chain: -> (entry or NO_ENTRY, leave, e afterwards)"""
    n = len(body)
    spec, spec_leave, last, safe = seg
    lo, hi = k * SEG, min((k + 1) * SEG, n)
    if e >= hi:
        return NO_ENTRY, 0, e
    ent = e
    after = e + int(esz[e]) if e < n else e + 2
    if after >= hi:
        e = min(after, 0xFFFFFFFF)
    elif e - lo < safe:
        e = spec_leave
    else:
        lt = lo + int(last[e - lo])
        e = min(lt + int(esz[lt]), 0xFFFFFFFF)
    return ent, e, e


def chain_in_groups(body, esz, segs, group=64):
    """snappy_stream_chain_groups + snappy_stream_chain_link: every group of `group` segments is
    walked on the ASSUMPTION that the parse enters its first segment inside the safe prefix with an
    element that stays inside the segment (then it leaves that segment with the speculative parse
    whatever the entry was); one pass then links the groups and walks those again whose assumption
    does not hold.  -> (entry, leave, end, groups walked again): the first three must equal chain()'s."""
    n = len(body)
    nseg = len(segs)
    entry, leave = [NO_ENTRY] * nseg, [0] * nseg
    exits = []
    for g0 in range(0, nseg, group):                           # step 1: all groups "at once"
        if g0 == 0:
            e, first = 0, 0
        else:
            e, first = segs[g0][1], g0 + 1                     # leaves the first segment with the speculative parse
            leave[g0] = e
        for k in range(first, min(g0 + group, nseg)):
            entry[k], leave[k], e = chain_step(body, esz, segs[k], k, e)
        exits.append(e)
    e, redone = exits[0], 0
    for gi, g0 in enumerate(range(group, nseg, group), start=1):  # step 2: the link
        lo, hi = g0 * SEG, min((g0 + 1) * SEG, n)
        after = e + int(esz[e]) if e < n else e + 2
        if lo <= e < hi and after < hi and e - lo < segs[g0][3]:
            entry[g0] = e
            e = exits[gi]
        else:
            redone += 1
            for k in range(g0, min(g0 + group, nseg)):
                entry[k], leave[k], e = chain_step(body, esz, segs[k], k, e)
    return entry, leave, e, redone


def settle(esz, l, n, k, entry, spec):
    """snappy_stream_settle: the true tags of a segment -- the parse from `entry` until it meets the
    speculative one, then the speculative one.  -> (sorted tag positions, where the walk leaves)"""
    lo, hi = k * SEG, min((k + 1) * SEG, n)
    if entry == NO_ENTRY:
        return [], None
    tags, p = [], entry
    while p < hi:
        if p in spec:                                          # from a common tag on the parses are one
            tags += sorted(t for t in spec if t >= p)
            return tags, None
        tags.append(p)
        p += int(esz[p])
    return tags, min(p, 0xFFFFFFFF)


def fragment_boundaries(body):
    """The whole pre-pass.  -> dict(grain, frag_pos, end, total, refused): grain 1 = every multiple
    of 32 KiB of output has an element starting at it, 2 = every other one, 0 = neither."""
    esz, l, segs = build_index(body)
    n = len(body)
    entry, leave, end = chain(body, esz, segs)
    refused = False
    starts = {}                                                # output offset -> first element starting there
    out = 0
    for k, (spec, spec_leave, last, safe) in enumerate(segs):
        tags, own_leave = settle(esz, l, n, k, entry[k], spec)
        if entry[k] != NO_ENTRY:
            got = spec_leave if own_leave is None else own_leave
            refused |= got != leave[k]                         # the kernels' cross-check
        for t in tags:
            if out % FRAG == 0 and out not in starts:
                starts[out] = t
            out += int(l[t])
    return dict(starts=starts, end=end, total=out, refused=refused)


def grain_of(starts, ulength):
    nfrag = (ulength + FRAG - 1) // FRAG
    missing = [f for f in range(nfrag) if f * FRAG not in starts]
    if not missing:
        return 1
    return 2 if all(f & 1 for f in missing) else 0
