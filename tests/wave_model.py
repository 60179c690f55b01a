"""Lane-level model of the compress kernel's step logic (TEST INFRASTRUCTURE).

`snappy_parse_fragments` (csnappy_amd/csrc/csnappy_kernels.hip) evaluates 64 positions of the
reference's sequential probe loop per step and truncates the step at the first lane that shares
a hash slot with an earlier lane.  This file restates that step logic in plain Python, lane by
lane, so the claim "the wave-step algorithm is bit-identical to the sequential loop of
csnappy_compress.c:469-606" can be fuzzed against the oracle on the CPU, where there is no GPU.
It models the ALGORITHM (lane roles, conflict truncation, the chain of copies through a step,
cursor update, EmitCopy chunking); the HIP code is a SIMT transcription of it.

  compress_fragment      v1: one match per step (the first GPU version; kept as the simplest
                         statement of the truncation argument)
  compress_fragment_v2   dense multi-match steps + sparse steps: what the kernel runs today
"""
import struct

FRAG = 32768
MARGIN = 15
KMUL = 0x1E35A7BD
WAVE = 64


def scan_pos(s, i):
    a, b = i >> 5, i & 31
    return s + 16 * a * (a + 1) + b * (a + 1)


def plan_copy(length, off):
    k64 = k60 = 0
    if length >= 68:
        k64 = (length - 68) // 64 + 1
        length -= 64 * k64
    if length > 64:
        k60 = 1
        length -= 60
    return k64, k60, length, 3 * (k64 + k60) + (2 if (length < 12 and off < 2048) else 3)


def encode_records(F, records):
    out = bytearray()
    for lit_start, lit_len, coff, clen in records:
        if lit_len:
            n = lit_len - 1
            if lit_len <= 60:
                out.append(n << 2)
            elif lit_len <= 256:
                out += bytes([60 << 2, n])
            else:
                out += bytes([61 << 2, n & 0xFF, n >> 8])
            out += F[lit_start:lit_start + lit_len]
        if clen:
            k64, k60, last, _ = plan_copy(clen, coff)
            lo, hi = coff & 0xFF, coff >> 8
            out += bytes([0xFE, lo, hi]) * k64
            out += bytes([0xEE, lo, hi]) * k60
            if last < 12 and coff < 2048:
                out += bytes([1 + ((last - 4) << 2) + ((coff >> 8) << 5), lo])
            else:
                out += bytes([2 + ((last - 1) << 2), lo, hi])
    return bytes(out)


def compress_fragment(F, p, s_entries=None, stats=None):
    """Wave-step model. F: bytes (<= 32768). Returns compressed bytes."""
    F = bytes(F)
    n = len(F)
    shift = 33 - p
    pad = F + b"\0" * 32
    rd32 = lambda i: struct.unpack_from("<I", pad, i)[0]
    if s_entries is None:
        s_entries = min(1 << (p - 1), 2048)
    smask = s_entries - 1
    records = []
    next_emit = 0
    if n >= MARGIN:
        tab = [0] * (1 << (p - 1))
        ip_limit = n - MARGIN
        ip, spec, s, qi = 0, 0, 1, 0
        while True:
            if stats is not None:
                stats["steps"] = stats.get("steps", 0) + 1
            pos, valid, probing = [0] * WAVE, [False] * WAVE, [True] * WAVE
            for lane in range(WAVE):
                if lane < spec:
                    k = lane + (2 - spec)
                    pos[lane] = ip - 1 + k
                    valid[lane] = True
                    probing[lane] = k == 1
                else:
                    i = qi + lane - spec
                    pos[lane] = scan_pos(s, i)
                    valid[lane] = scan_pos(s, i + 1) <= ip_limit
                    if not valid[lane]:
                        pos[lane] = 0
            w = [rd32(pos[l]) for l in range(WAVE)]
            h = [((w[l] * KMUL) & 0xFFFFFFFF) >> shift for l in range(WAVE)]
            key = [h[l] & smask for l in range(WAVE)]
            first = {}
            for l in range(WAVE):
                if valid[l] and key[l] not in first:
                    first[key[l]] = l
            cand = [tab[h[l]] for l in range(WAVE)]
            cw = [rd32(cand[l]) for l in range(WAVE)]
            c = next((l for l in range(WAVE) if valid[l] and first[key[l]] < l), 64)
            v = next((l for l in range(WAVE) if not valid[l]), 64)
            ulim = min(c, v)
            m = next((l for l in range(ulim) if probing[l] and cw[l] == w[l]), None)
            if m is None:
                for l in range(ulim):
                    tab[h[l]] = pos[l]
                if ulim == v and v < 64:
                    break
                if ulim < spec:
                    spec -= ulim
                    s, qi = ip + 1, 0
                else:
                    if spec:
                        s, qi = ip + 1, 0
                    qi += ulim - spec
                    spec = 0
                continue
            for l in range(m + 1):
                tab[h[l]] = pos[l]
            base, cnd = pos[m], cand[m]
            ma, mb = cnd + 4, base + 4
            L = n - mb
            k = 0
            while k < L and F[ma + k] == F[mb + k]:
                k += 1
            matched = 4 + k
            records.append((next_emit, base - next_emit, base - cnd, matched))
            ip = base + matched
            next_emit = ip
            if ip >= ip_limit:
                break
            spec, s, qi = 2, ip + 1, 0
    if next_emit < n:
        records.append((next_emit, n - next_emit, 0, 0))
    return encode_records(F, records)


# =================================================================================================
# v2: dense multi-match steps.  A dense step lays 64 CONSECUTIVE positions on the lanes, every
# lane computes its candidate and a lane-local match length (capped at LM), and the true chain of
# matches through the step is then walked (on the GPU: on the scalar unit).  Sparse steps (the v1
# layout) are used only once a scan has made 32 probes without a match (stride >= 2).
# =================================================================================================
LM = 16


def compress_fragment_v2(F, p, s_entries=None, stats=None, lm=LM):
    F = bytes(F)
    n = len(F)
    shift = 33 - p
    pad = F + b"\0" * 64
    rd32 = lambda i: struct.unpack_from("<I", pad, i)[0]
    if s_entries is None:
        s_entries = min(1 << (p - 1), 2048)
    smask = s_entries - 1
    records = []
    next_emit = 0

    def lcp(a, b, start, limit):
        k = start
        while k < limit and F[a + k] == F[b + k]:
            k += 1
        return k

    if n >= MARGIN:
        tab = [0] * (1 << (p - 1))
        ip_limit = n - MARGIN
        ip, spec, s, qi = 0, 0, 1, 0
        done = False
        while not done:
            if stats is not None:
                stats["steps"] = stats.get("steps", 0) + 1
            sparse = spec == 0 and qi >= 32
            if stats is not None and sparse:
                stats["sparse"] = stats.get("sparse", 0) + 1
            # ---- lane positions / validity ------------------------------------------------------
            pos, valid = [0] * WAVE, [False] * WAVE
            if sparse:
                for l in range(WAVE):
                    pos[l] = scan_pos(s, qi + l)
                    valid[l] = scan_pos(s, qi + l + 1) <= ip_limit
            else:
                p0 = ip - 1 if spec == 2 else ip if spec == 1 else s + qi
                for l in range(WAVE):
                    pos[l] = p0 + l
                    valid[l] = pos[l] + 1 <= ip_limit
            for l in range(WAVE):
                if not valid[l]:
                    pos[l] = 0
            w = [rd32(pos[l]) for l in range(WAVE)]
            h = [((w[l] * KMUL) & 0xFFFFFFFF) >> shift for l in range(WAVE)]
            key = [h[l] & smask for l in range(WAVE)]
            first = {}
            for l in range(WAVE):
                if valid[l] and key[l] not in first:
                    first[key[l]] = l
            cand = [tab[h[l]] for l in range(WAVE)]
            c_ = next((l for l in range(WAVE) if valid[l] and first[key[l]] < l), 64)
            v = next((l for l in range(WAVE) if not valid[l]), 64)
            ulim = min(c_, v)

            if sparse:
                cw = [rd32(cand[l]) for l in range(WAVE)]
                m = next((l for l in range(ulim) if cw[l] == w[l]), None)
                if m is None:
                    for l in range(ulim):
                        tab[h[l]] = pos[l]
                    if ulim == v and v < 64:
                        break
                    qi += ulim
                    continue
                for l in range(m + 1):
                    tab[h[l]] = pos[l]
                base, cnd = pos[m], cand[m]
                matched = lcp(cnd, base, 4, n - base)
                records.append((next_emit, base - next_emit, base - cnd, matched))
                ip = base + matched
                next_emit = ip
                if ip >= ip_limit:
                    break
                spec, s, qi = 2, ip + 1, 0
                continue

            # ---- dense step ------------------------------------------------------------------------
            mlen = [lcp(cand[l], pos[l], 0, min(lm, n - pos[l])) if valid[l] else 0 for l in range(WAVE)]
            ins = set()
            if spec == 2:
                a, zlane = 1, 2
                if ulim >= 1:
                    ins.add(0)
            elif spec == 1:
                a, zlane = 0, 1
            else:
                a, zlane = 0, -qi
            seg_s = ip + 1 if spec else s
            while True:
                seg_end = zlane + 31
                e = min(seg_end, ulim - 1, 63)
                i = next((l for l in range(a, e + 1) if mlen[l] >= 4), None)
                if i is None:
                    for l in range(a, e + 1):
                        ins.add(l)
                    if e + 1 == v and v == ulim and v <= min(seg_end, 63):
                        done = True  # the next probe is past ip_limit: emit_remainder
                        break
                    if e < a and a == 1 and zlane == 2 and e == 0:
                        # only the insert lane was usable (rematch lane shares its slot)
                        spec = 1
                        break
                    if e < a:
                        # cannot happen for chained segments (c < ulim is required to chain) nor for
                        # spec 0/1 starts (lane 0 is never conflicted; invalid lane 0 handled above)
                        raise AssertionError((a, e, ulim, v, spec))
                    spec, s, qi = 0, seg_s, e + 1 - zlane
                    break
                for l in range(a, i + 1):
                    ins.add(l)
                base, cnd, L = pos[i], cand[i], mlen[i]
                wide = False
                if L == lm and base + L < n:
                    L = lcp(cnd, base, lm, n - base)
                    wide = True
                records.append((next_emit, base - next_emit, base - cnd, L))
                ip = base + L
                next_emit = ip
                if ip >= ip_limit:
                    done = True
                    break
                c = ip - pos[0] if valid[0] else None
                if wide or c is None or c > 63 or c >= ulim:
                    spec, s, qi = 2, ip + 1, 0
                    break
                ins.add(c - 1)
                a, zlane, seg_s = c, c + 1, ip + 1
                if stats is not None:
                    stats["chained"] = stats.get("chained", 0) + 1
            for l in ins:
                tab[h[l]] = pos[l]
    if next_emit < n:
        records.append((next_emit, n - next_emit, 0, 0))
    return encode_records(F, records)


# =================================================================================================
# v3: v2 + optimistic walks.  A lane that shares a hash slot with an earlier lane only matters if
# BOTH lanes are actually probed / inserted by the chain; lanes inside a copy never are.  So the
# chain is first walked over all valid lanes, then the lanes it really inserted are checked for
# slot sharing among themselves, and only if such a lane x exists is the walk redone with the step
# cut at x.  Repetitive data (runs, small-integer words) otherwise collapses to steps of a few
# positions because neighbouring positions inside a run share their slot.
# =================================================================================================
def compress_fragment_v3(F, p, s_entries=None, stats=None, lm=LM):
    F = bytes(F)
    n = len(F)
    shift = 33 - p
    pad = F + b"\0" * 64
    rd32 = lambda i: struct.unpack_from("<I", pad, i)[0]
    if s_entries is None:
        s_entries = min(1 << (p - 1), 1024)
    smask = s_entries - 1
    records = []
    next_emit = 0

    def lcp(a, b, start, limit):
        k = start
        while k < limit and F[a + k] == F[b + k]:
            k += 1
        return k

    if n >= MARGIN:
        tab = [0] * (1 << (p - 1))
        ip_limit = n - MARGIN
        cur = dict(ip=0, spec=0, s=1, qi=0)
        fin = False
        while not fin:
            if stats is not None:
                stats["steps"] = stats.get("steps", 0) + 1
            ip, spec, s, qi = cur["ip"], cur["spec"], cur["s"], cur["qi"]
            sparse = spec == 0 and qi >= 32
            p0 = ip - 1 if spec == 2 else ip if spec == 1 else s + qi
            pos, valid = [0] * WAVE, [False] * WAVE
            for l in range(WAVE):
                if sparse:
                    pos[l] = scan_pos(s, qi + l)
                    valid[l] = scan_pos(s, qi + l + 1) <= ip_limit
                else:
                    pos[l] = p0 + l
                    valid[l] = pos[l] + 1 <= ip_limit
                if not valid[l]:
                    pos[l] = 0
            w = [rd32(pos[l]) for l in range(WAVE)]
            h = [((w[l] * KMUL) & 0xFFFFFFFF) >> shift for l in range(WAVE)]
            key = [h[l] & smask for l in range(WAVE)]
            cand = [tab[h[l]] for l in range(WAVE)]
            mlen = [lcp(cand[l], pos[l], 0, lm) if valid[l] else 0 for l in range(WAVE)]
            v = next((l for l in range(WAVE) if not valid[l]), 64)
            seen = set()
            any_share = False
            for l in range(v):
                if key[l] in seen:
                    any_share = True
                seen.add(key[l])

            def walk(ulim):
                """-> (new records, new cursor, fin, e_final, inserted lanes); no side effects"""
                recs, c = [], dict(cur)
                ne = next_emit
                match = [l < ulim and mlen[l] >= 4 for l in range(WAVE)]
                hole = set()
                done = False
                if sparse:
                    if not any(match):
                        e_final = ulim - 1
                        if ulim == v and v < 64:
                            done = True
                        else:
                            c["qi"] = qi + ulim
                    else:
                        i = match.index(True)
                        e_final = i
                        base, cnd, L = pos[i], cand[i], mlen[i]
                        if L == lm and base + L < n:
                            L = lcp(cnd, base, lm, n - base)
                        recs.append((ne, base - ne, base - cnd, L))
                        c["ip"] = base + L
                        ne = base + L
                        if base + L >= ip_limit:
                            done = True
                        c["spec"] = 2
                else:
                    if spec == 2:
                        a, zl, seg_s = 1, 2, ip + 1
                    elif spec == 1:
                        a, zl, seg_s = 0, 1, ip + 1
                    else:
                        a, zl, seg_s = 0, -qi, s
                    lim = zl + 31
                    while True:
                        i = next((l for l in range(a, 64) if match[l]), 64)
                        if i > lim or i > 63:
                            e = min(lim, ulim - 1)
                            e_final = e
                            if ulim == v and v <= lim and v < 64:
                                done = True
                            elif e < a:
                                c["spec"] = 1
                                e_final = 0
                            else:
                                c["spec"], c["s"], c["qi"] = 0, seg_s, e + 1 - zl
                            break
                        cnd, L, base, wide = cand[i], mlen[i], p0 + i, False
                        if L == lm and base + L < n:
                            wide, L = True, lcp(cnd, base, lm, n - base)
                        recs.append((ne, base - ne, base - cnd, L))
                        c["ip"] = base + L
                        ne = base + L
                        e_final = i
                        cc = i + L
                        if base + L >= ip_limit:
                            done = True
                            break
                        if wide or cc >= ulim:
                            c["spec"] = 2
                            break
                        for l in range(i + 1, cc - 1):
                            hole.add(l)
                        a, zl, lim, seg_s = cc, cc + 1, cc + 32, base + L + 1
                ins = [l for l in range(0, e_final + 1) if l not in hole]
                return recs, c, done, ne, ins

            ulim = v
            recs, c, done, ne, ins = walk(ulim)
            if any_share:
                seen = set()
                x = None
                for l in ins:
                    if key[l] in seen:
                        x = l
                        break
                    seen.add(key[l])
                if x is not None:
                    if stats is not None:
                        stats["rewalk"] = stats.get("rewalk", 0) + 1
                    recs, c, done, ne, ins = walk(x)
            records += recs
            cur, fin, next_emit = c, done, ne
            for l in ins:
                tab[h[l]] = pos[l]
    if next_emit < n:
        records.append((next_emit, n - next_emit, 0, 0))
    return encode_records(F, records)


# =================================================================================================
# v4: v2 + in-step forwarding.  Instead of cutting the step at the first lane that shares a hash
# slot with an earlier lane, the chain treats such flagged lanes as stops: when it arrives at one
# that is about to be PROBED, the lane's candidate is patched to the latest earlier lane of this
# step that was inserted and has the same hash (its bytes are that lane's own bytes -- no memory
# access), else the table value stands.  Lanes inside copies are never probed and need nothing.
# Inserted lanes that are superseded by a later inserted lane with the same hash do not commit.
# =================================================================================================
def compress_fragment_v4(F, p, s_entries=None, stats=None, lm=LM, fallback_cut=False):
    F = bytes(F)
    n = len(F)
    shift = 33 - p
    pad = F + b"\0" * 64
    rd32 = lambda i: struct.unpack_from("<I", pad, i)[0]
    if s_entries is None:
        s_entries = min(1 << (p - 1), 1024)
    smask = s_entries - 1
    records = []
    next_emit = 0

    def lcp(a, b, start, limit):
        k = start
        while k < limit and F[a + k] == F[b + k]:
            k += 1
        return k

    if n >= MARGIN:
        tab = [0] * (1 << (p - 1))
        ip_limit = n - MARGIN
        ip, spec, s, qi = 0, 0, 1, 0
        fin = False
        while not fin:
            if stats is not None:
                stats["steps"] = stats.get("steps", 0) + 1
            sparse = spec == 0 and qi >= 32
            p0 = ip - 1 if spec == 2 else ip if spec == 1 else s + qi
            pos, valid = [0] * WAVE, [False] * WAVE
            for l in range(WAVE):
                if sparse:
                    pos[l] = scan_pos(s, qi + l)
                    valid[l] = scan_pos(s, qi + l + 1) <= ip_limit
                else:
                    pos[l] = p0 + l
                    valid[l] = pos[l] + 1 <= ip_limit
                if not valid[l]:
                    pos[l] = 0
            w = [rd32(pos[l]) for l in range(WAVE)]
            h = [((w[l] * KMUL) & 0xFFFFFFFF) >> shift for l in range(WAVE)]
            key = [h[l] & smask for l in range(WAVE)]
            cand = [tab[h[l]] for l in range(WAVE)]
            mlen = [lcp(cand[l], pos[l], 0, lm) if valid[l] else 0 for l in range(WAVE)]
            v = next((l for l in range(WAVE) if not valid[l]), 64)
            firstk = {}
            flagged = [False] * WAVE  # has an earlier valid lane with the same filter key
            for l in range(v):
                if key[l] in firstk:
                    flagged[l] = True
                else:
                    firstk[key[l]] = l
            if sparse:
                # sparse steps keep the simple cut
                c_ = next((l for l in range(v) if flagged[l]), 64)
                ulim = min(c_, v)
                m = next((l for l in range(ulim) if mlen[l] >= 4), None)
                if m is None:
                    for l in range(ulim):
                        tab[h[l]] = pos[l]
                    if ulim == v and v < 64:
                        break
                    qi += ulim
                    continue
                for l in range(m + 1):
                    tab[h[l]] = pos[l]
                base, cnd = pos[m], cand[m]
                L = mlen[m]
                if L == lm and base + L < n:
                    L = lcp(cnd, base, lm, n - base)
                records.append((next_emit, base - next_emit, base - cnd, L))
                ip = base + L
                next_emit = ip
                if ip >= ip_limit:
                    break
                spec, s, qi = 2, ip + 1, 0
                continue

            # ---- dense step with forwarding ----
            ulim = v
            ins = []      # inserted lanes in order
            dead = set()  # inserted lanes superseded by a later inserted lane with the same hash

            def insert(l):
                for j in ins:
                    if h[j] == h[l]:
                        dead.add(j)
                ins.append(l)

            def patched(l):
                """(candidate, match length) of lane l given the lanes inserted so far"""
                if flagged[l]:
                    js = [j for j in ins if h[j] == h[l]]
                    if js:
                        j = js[-1]
                        if stats is not None:
                            stats["fwd"] = stats.get("fwd", 0) + 1
                        return pos[j], lcp(pos[j], pos[l], 0, lm)
                    if fallback_cut:
                        return None, None  # the table value was not gathered: cut the step here
                return cand[l], mlen[l]

            if spec == 2:
                a, zl, seg_s = 1, 2, ip + 1
                insert(0)
            elif spec == 1:
                a, zl, seg_s = 0, 1, ip + 1
            else:
                a, zl, seg_s = 0, -qi, s
            lim = zl + 31
            while True:
                # probe lanes a.. in order until a match, the segment limit, or the end of the step
                i, found = a, None
                while i <= min(lim, ulim - 1, 63):
                    cnd, L = patched(i)
                    if cnd is None:
                        ulim = i
                        if stats is not None:
                            stats["cuts"] = stats.get("cuts", 0) + 1
                        break
                    insert(i)
                    if L >= 4:
                        found = (i, cnd, L)
                        break
                    i += 1
                if found is None:
                    e = min(lim, ulim - 1)
                    if ulim == v and v <= lim and v < 64:
                        fin = True
                    elif e < a:
                        # only the ip-1 insert (or nothing) was usable
                        if spec == 2 and a == 1:
                            spec = 1
                        elif a == 0:
                            raise AssertionError("lane 0 is never flagged")
                        else:
                            spec = 2  # the re-match probe of the last copy is the cut lane
                    else:
                        spec, s, qi = 0, seg_s, e + 1 - zl
                    break
                i, cnd, L = found
                base, wide = p0 + i, False
                if L == lm and base + L < n:
                    wide, L = True, lcp(cnd, base, lm, n - base)
                records.append((next_emit, base - next_emit, base - cnd, L))
                ip = base + L
                next_emit = ip
                c = i + L
                if ip >= ip_limit:
                    fin = True
                    break
                if c >= ulim:
                    spec = 2
                    break
                # (a match extended past the lane-local cap that still ends inside the usable lanes
                # is a link of the chain like any other)
                insert(c - 1)
                a, zl, lim, seg_s = c, c + 1, c + 32, ip + 1
            for l in ins:
                if l not in dead:
                    tab[h[l]] = pos[l]
    if next_emit < n:
        records.append((next_emit, n - next_emit, 0, 0))
    return encode_records(F, records)


def compress_fragment_v5(F, p, s_entries=None, stats=None, lm=LM):
    """Round 3's step loop (parse_lean in csnappy_kernels.hip), statement by statement on the
    scalar side: the cursor is (s, q1) -- the scan that starts at s has made q1 - 1 probes, q1 == 0
    means "a copy just ended at s - 1: insert ip - 1, probe ip" -- lane 0 of a dense step is always
    insert-only, the lanes in front of the scan limit are arithmetic, every lane precomputes the
    next stop behind its own match (nx: 64 = the re-match probe falls outside the usable lanes,
    65 = none of the 33 probes behind the copy is a stop, lane | 128 = that stop is special), the
    chain follows nx and visits special lanes (flagged: shares its filter key with a lower lane;
    wide: matches the whole lane-local window), and the step's end is read off the `taken` lanes.
    s_entries small makes the filter raise false alarms, which the visits must survive."""
    F = bytes(F)
    n = len(F)
    shift = 33 - p
    pad = F + b"\0" * 64
    rd32 = lambda i: struct.unpack_from("<I", pad, i)[0]
    if s_entries is None:
        s_entries = min(1 << (p - 1), 1024)
    smask = s_entries - 1
    records = []
    next_emit = 0

    def lcp(a, b, start, limit):
        k = start
        while k < limit and F[a + k] == F[b + k]:
            k += 1
        return k

    if n > MARGIN:
        tab = [0] * (1 << (p - 1))
        ip_limit = n - MARGIN
        s, q1 = 1, 1
        fin = False
        guard = 0
        while not fin:
            guard += 1
            assert guard <= n, "the cursor stopped moving"
            if stats is not None:
                stats["steps"] = stats.get("steps", 0) + 1
            sparse = q1 > 32
            p0 = s + q1 - 2
            pos, valid = [0] * WAVE, [False] * WAVE
            for l in range(WAVE):
                if sparse:
                    pos[l] = scan_pos(s, q1 - 1 + l)
                    valid[l] = scan_pos(s, q1 + l) <= ip_limit
                else:
                    pos[l] = p0 + l
                    valid[l] = pos[l] < ip_limit
                if not valid[l]:
                    pos[l] = 0
            h = [((rd32(pos[l]) * KMUL) & 0xFFFFFFFF) >> shift for l in range(WAVE)]
            tabbed = valid[:]
            cand = [tab[h[l]] if tabbed[l] else 0 for l in range(WAVE)]
            seen = {}
            flagged = [False] * WAVE  # (a superset of the lanes that share their SLOT with a lower lane)
            for l in range(WAVE):
                if tabbed[l]:
                    k = h[l] & smask
                    if k in seen and l != 0:
                        flagged[l] = True
                    seen.setdefault(k, l)
            if sparse:
                v = next((l for l in range(WAVE) if not valid[l]), 64)
                c1 = next((l for l in range(WAVE) if flagged[l]), 64)
                ul = min(c1, v)
                mlen = [lcp(cand[l], pos[l], 0, lm) if l < ul and tabbed[l] else 0 for l in range(WAVE)]
                m = next((l for l in range(ul) if mlen[l] >= 4), None)
                if m is None:
                    e_final = ul - 1
                    if ul == v and v < 64:
                        fin = True
                    else:
                        q1 += ul
                else:
                    e_final = m
                    base, cnd, L = pos[m], cand[m], mlen[m]
                    if L == lm and base + L < n:
                        L = lcp(cnd, base, lm, n - base)
                    records.append((next_emit, base - next_emit, base - cnd, L))
                    ip = base + L
                    next_emit = ip
                    if ip >= ip_limit:
                        fin = True
                    s, q1 = ip + 1, 0
                for l in range(WAVE):
                    if l <= e_final and tabbed[l]:
                        tab[h[l]] = pos[l]
                continue

            # ---- dense step ----
            ulim = min(64, ip_limit - p0)
            mlen = [lcp(cand[l], pos[l], 0, lm) if tabbed[l] else 0 for l in range(WAVE)]
            stopm = [(mlen[l] >= 4 and l != 0) or flagged[l] for l in range(WAVE)]
            special = [(mlen[l] == lm and p0 + l + lm < n) or flagged[l] for l in range(WAVE)]
            cl = [l + mlen[l] for l in range(WAVE)]

            def next_code(cc):
                if cc >= ulim:
                    return 64
                j = next((x for x in range(cc, 64) if stopm[x]), None)
                if j is None or j - cc > 32:
                    return 65
                return j | (128 if special[j] else 0)

            nx = [next_code(cl[l]) for l in range(WAVE)]
            lim0 = 33 - q1
            i0 = next((x for x in range(64) if stopm[x]), 64)
            t = (i0 | (128 if special[i0] else 0)) if (i0 <= lim0 and i0 <= 63) else 65
            taken = []
            while True:
                while t < 64:
                    taken.append(t)
                    t = nx[t]
                if t < 128:
                    break
                i = t & 63
                L = mlen[i]
                if flagged[i]:
                    if stats is not None:
                        stats["visits"] = stats.get("visits", 0) + 1
                    # highest lower lane with the same slot that this step inserts (not strictly inside a taken copy)
                    def inside(x):
                        below = [y for y in taken if y < x]
                        return bool(below) and x + 1 < cl[below[-1]]
                    same = [x for x in range(i) if tabbed[x] and h[x] == h[i] and not inside(x)]
                    if same:
                        j = same[-1]
                        L = lcp(pos[j], pos[i], 0, lm)
                        cand[i] = p0 + j
                        if stats is not None:
                            stats["fwd"] = stats.get("fwd", 0) + 1
                    if L < 4:
                        lim_cur = cl[taken[-1]] + 32 if taken else lim0
                        i2 = next((x for x in range(i + 1, 64) if stopm[x]), 64)
                        t = 65 if (i2 > lim_cur or i2 > 63) else i2 | (128 if special[i2] else 0)
                        continue
                if L == lm and p0 + i + lm < n:
                    L = lcp(cand[i], p0 + i, lm, n - (p0 + i))
                mlen[i], cl[i] = L, i + L
                taken.append(i)
                t = next_code(i + L)
            emit0 = next_emit
            any_ = bool(taken)
            last = taken[-1] if any_ else 0
            c = cl[last]
            ip = p0 + c
            end_a = t == 64
            lim = c + 32 if any_ else lim0
            e = min(lim, ulim - 1)
            e_final = last if end_a else e
            fin = (ip >= ip_limit) if end_a else (ulim <= lim and ulim < 64)
            if any_:
                next_emit = ip
            q1 = 0 if end_a else (e + 1 - c if any_ else q1 + e)
            if any_:
                s = ip + 1
            prev_end = None
            for l in taken:
                records.append((emit0 if prev_end is None else p0 + prev_end, None, p0 + l - cand[l], mlen[l], p0 + l))
                prev_end = cl[l]
            # commit: lanes up to e_final that are not strictly inside a taken copy; the last lane of a slot wins
            def inside_final(x):
                below = [y for y in taken if y < x]
                return bool(below) and x + 1 < cl[below[-1]]
            for l in range(WAVE):
                if l <= e_final and tabbed[l] and not inside_final(l):
                    tab[h[l]] = pos[l]
        # (records of dense steps carry (lit_start, -, offset, length, base): bring them to the common form)
        records = [(r[0], r[4] - r[0], r[2], r[3]) if len(r) == 5 else r for r in records]
    if next_emit < n:
        records.append((next_emit, n - next_emit, 0, 0))
    return encode_records(F, records)


def compress_fragment_v6(F, p, stats=None, lm=LM, late_pos=0x7fc1):
    """Round 5's step loop (parse_lean in csnappy_kernels.hip, the placements with their table in
    LDS), on the scalar side: the lanes that share their SLOT with a lower lane of the step are known
    exactly (the returning add on the table entry counts them); once a step may have inserted a
    position >= late_pos every lane with a bucket is flagged instead (an add could carry out of such
    an entry's half of its dword) -- a flagged lane's visit is exact whatever flagged it; lane 0, which
    is insert-only, holds the chain's FIRST stop in its next-stop entry (the same search with cl = 0
    and lim0 probes) and the walk starts at it; the cursor is (pz, q1): pz = the position lane 0 of the
    next step takes, q1 as in v5, the scan's start pz - q1 + 2 only computed by sparse steps; a dense
    step ends the fragment's scan exactly when pz + 1 reaches the scan limit."""
    F = bytes(F)
    n = len(F)
    shift = 33 - p
    pad = F + b"\0" * 64
    rd32 = lambda i: struct.unpack_from("<I", pad, i)[0]
    records = []
    next_emit = 0

    def lcp(a, b, start, limit):
        k = start
        while k < limit and F[a + k] == F[b + k]:
            k += 1
        return k

    if n > MARGIN:
        tab = [0] * (1 << (p - 1))
        ip_limit = n - MARGIN
        pz, q1 = 0, 1
        fin = False
        late = False
        guard = 0
        while not fin:
            guard += 1
            assert guard <= n, "the cursor stopped moving"
            if stats is not None:
                stats["steps"] = stats.get("steps", 0) + 1
            sparse = q1 > 32
            p0 = pz
            s = pz - q1 + 2
            pos, valid = [0] * WAVE, [False] * WAVE
            for l in range(WAVE):
                if sparse:
                    pos[l] = scan_pos(s, q1 - 1 + l)
                    valid[l] = scan_pos(s, q1 + l) <= ip_limit
                else:
                    pos[l] = p0 + l
                    valid[l] = pos[l] < ip_limit
                if not valid[l]:
                    pos[l] = 0
            h = [((rd32(pos[l]) * KMUL) & 0xFFFFFFFF) >> shift for l in range(WAVE)]
            tabbed = valid[:]
            cand = [tab[h[l]] if tabbed[l] else 0 for l in range(WAVE)]
            seen = set()
            flagged = [False] * WAVE  # the lanes that share their slot with a lower lane (late: every lane with a bucket)
            for l in range(WAVE):
                if tabbed[l]:
                    if (h[l] in seen or late) and l != 0:
                        flagged[l] = True
                    seen.add(h[l])
            if sparse:
                v = next((l for l in range(WAVE) if not valid[l]), 64)
                c1 = next((l for l in range(WAVE) if flagged[l]), 64)
                ul = min(c1, v)
                mlen = [lcp(cand[l], pos[l], 0, lm) if l < ul and tabbed[l] else 0 for l in range(WAVE)]
                m = next((l for l in range(ul) if mlen[l] >= 4), None)
                if m is None:
                    e_final = ul - 1
                    if ul == v and v < 64:
                        fin = True
                    else:
                        q1 += ul
                        pz += ul
                else:
                    e_final = m
                    base, cnd, L = pos[m], cand[m], mlen[m]
                    if L == lm and base + L < n:
                        L = lcp(cnd, base, lm, n - base)
                    records.append((next_emit, base - next_emit, base - cnd, L))
                    ip = base + L
                    next_emit = ip
                    if ip >= ip_limit:
                        fin = True
                    pz, q1 = ip - 1, 0
                if 0 <= e_final < WAVE and pos[e_final] >= late_pos:
                    late = True
                for l in range(WAVE):
                    if l <= e_final and tabbed[l]:
                        tab[h[l]] = pos[l]
                continue

            # ---- dense step ----
            ulim = min(64, ip_limit - p0)
            if p0 + 63 >= late_pos:
                late = True  # (for the steps behind this one)
            mlen = [lcp(cand[l], pos[l], 0, lm) if tabbed[l] and l != 0 else 0 for l in range(WAVE)]
            stopm = [mlen[l] >= 4 or flagged[l] for l in range(WAVE)]
            special = [(mlen[l] == lm and p0 + l + lm < n) or flagged[l] for l in range(WAVE)]
            cl = [l + mlen[l] for l in range(WAVE)]

            def next_code(cc):
                if cc >= ulim:
                    return 64
                j = next((x for x in range(cc, 64) if stopm[x]), None)
                if j is None or j - cc > 32:
                    return 65
                return j | (128 if special[j] else 0)

            lim0 = 33 - q1

            def lane_code(l):
                # lane 0: the first stop among lanes 1 .. lim0 (cl = 0, lim0 probes); the others as in v5
                if cl[l] >= ulim:
                    return 64
                j = next((x for x in range(cl[l], 64) if stopm[x]), None)
                if j is None or j - cl[l] > (lim0 if l == 0 else 32):
                    return 65
                return j | (128 if special[j] else 0)

            nx = [lane_code(l) for l in range(WAVE)]
            taken = []
            t = nx[0]  # (the kernel's walk starts AT lane 0 and clears its mark afterwards)
            while True:
                while t < 64:
                    taken.append(t)
                    t = nx[t]
                if t < 128:
                    break
                i = t & 63
                L = mlen[i]
                if flagged[i]:
                    if stats is not None:
                        stats["visits"] = stats.get("visits", 0) + 1
                    # highest lower lane with the same slot that this step inserts (not strictly inside a taken copy)
                    def inside(x):
                        below = [y for y in taken if y < x]
                        return bool(below) and x + 1 < cl[below[-1]]
                    same = [x for x in range(i) if tabbed[x] and h[x] == h[i] and not inside(x)]
                    if same:
                        j = same[-1]
                        L = lcp(pos[j], pos[i], 0, lm)
                        cand[i] = p0 + j
                        if stats is not None:
                            stats["fwd"] = stats.get("fwd", 0) + 1
                    if L < 4:
                        lim_cur = cl[taken[-1]] + 32 if taken else lim0
                        i2 = next((x for x in range(i + 1, 64) if stopm[x]), 64)
                        t = 65 if (i2 > lim_cur or i2 > 63) else i2 | (128 if special[i2] else 0)
                        continue
                if L == lm and p0 + i + lm < n:
                    L = lcp(cand[i], p0 + i, lm, n - (p0 + i))
                mlen[i], cl[i] = L, i + L
                taken.append(i)
                t = next_code(i + L)
            emit0 = next_emit
            any_ = bool(taken)
            last = taken[-1] if any_ else 0
            c = cl[last]
            ip = p0 + c
            end_a = t == 64
            lim = c + 32 if any_ else lim0
            e = min(lim, ulim - 1)
            e_final = last if end_a else e
            if any_:
                next_emit = ip
            q1 = 0 if end_a else (e + 1 - c if any_ else q1 + e)
            pz = p0 + (c - 1 if end_a else e)
            fin = pz + 1 >= ip_limit
            prev_end = None
            for l in taken:
                records.append((emit0 if prev_end is None else p0 + prev_end, None, p0 + l - cand[l], mlen[l], p0 + l))
                prev_end = cl[l]
            # commit: lanes up to e_final that are not strictly inside a taken copy; the last lane of a slot wins
            def inside_final(x):
                below = [y for y in taken if y < x]
                return bool(below) and x + 1 < cl[below[-1]]
            for l in range(WAVE):
                if l <= e_final and tabbed[l] and not inside_final(l):
                    tab[h[l]] = pos[l]
        # (records of dense steps carry (lit_start, -, offset, length, base): bring them to the common form)
        records = [(r[0], r[4] - r[0], r[2], r[3]) if len(r) == 5 else r for r in records]
    if next_emit < n:
        records.append((next_emit, n - next_emit, 0, 0))
    return encode_records(F, records)
