"""CPU checks of the stream index's algorithm (tests/stream_model.py restates the snappy_stream_*
kernels): the fragment boundaries, the end of the parse and the total output it derives from
speculative per-segment parses must be those of the plain sequential walk -- for sound csnappy
streams, for streams with 64 KiB blocks, for foreign streams and for damaged ones (the index is
exact about the tag chain whatever the stream is worth; validity is the decoder's business)."""
import numpy as np
import pytest

import oracle
import stream_model as sm
from csnappy_amd import api


def _truth(body):
    pos, outs, end, total = sm.sequential_parse(body)
    starts = {}
    for p, o in zip(pos, outs):
        if o % sm.FRAG == 0 and o not in starts:
            starts[o] = p
    return starts, min(end, 0xFFFFFFFF), total


def _mixed(seed, nbytes):
    rng = np.random.default_rng(seed)
    parts, have = [], 0
    while have < nbytes:
        kind, n = int(rng.integers(0, 5)), int(rng.integers(3000, 90000))
        if kind == 0:
            part = api.generate_host(api.WG_TEXT, seed + have, 0, 1, n)
        elif kind == 1:
            part = rng.integers(0, 256, n, dtype=np.uint8)                  # 32 KiB literals
        elif kind == 2:
            part = api.generate_host(api.WG_LOW, seed + have, 0, 1, n)
        elif kind == 3:
            part = np.full(n, 7, dtype=np.uint8)
        else:                                                               # sparse matches: literal-heavy
            part = rng.integers(0, 256, n, dtype=np.uint8)
            for at in range(0, n - 600, int(rng.integers(200, 900))):
                part[at + 300:at + 312] = part[at:at + 12]
        parts.append(part)
        have += n
    return np.concatenate(parts)[:nbytes].tobytes()


def _body(stream):
    hdr, ulen = oracle.Port().get_uncompressed_length(stream)
    return stream[hdr:], ulen


@pytest.mark.parametrize("seed,nbytes", [(1, 3 * 32768), (2, 200001), (3, 333333)])
def test_index_of_a_csnappy_stream_finds_every_fragment(seed, nbytes):
    data = _mixed(seed, nbytes)
    body, ulen = _body(oracle.Port().compress(data, 16))
    got = sm.fragment_boundaries(body)
    starts, end, total = _truth(body)
    assert not got["refused"] and got["end"] == end == len(body) and got["total"] == total == ulen
    assert got["starts"] == starts
    assert sm.grain_of(got["starts"], ulen) == 1
    # the table path is exercised (literal-heavy stretches), not only the speculative one
    esz, l, segs = sm.build_index(body)
    entry, _, _ = sm.chain(body, esz, segs)
    beyond = sum(1 for k, e in enumerate(entry) if e != sm.NO_ENTRY and e - k * sm.SEG >= segs[k][3])
    assert beyond > 0 or nbytes <= 3 * 32768


def test_index_of_a_stream_with_64k_blocks_pairs_the_fragments():
    rng = np.random.default_rng(21)
    body, total = b"", 0
    for blk in range(5):
        size = 65536 if blk < 4 else 20000
        nlit = min(40000, size)
        body += bytes([(59 + 2) << 2]) + (nlit - 1).to_bytes(2, "little") + rng.integers(0, 256, nlit, dtype=np.uint8).tobytes()
        done = nlit
        while done < size:
            ln = min(64, size - done)
            body += bytes([2 | ((ln - 1) << 2)]) + int(rng.integers(33000, 39000)).to_bytes(2, "little")
            done += ln
        total += size
    got = sm.fragment_boundaries(body)
    starts, end, tot = _truth(body)
    assert got["starts"] == starts and got["end"] == end == len(body) and got["total"] == tot == total
    assert sm.grain_of(got["starts"], total) == 2


def test_index_is_exact_on_damaged_and_foreign_bodies():
    rng = np.random.default_rng(5)
    good, _ = _body(oracle.Port().compress(_mixed(9, 150000), 16))
    cases = [good[:len(good) - 17], good + bytes(500), bytes(rng.integers(0, 256, 50000, dtype=np.uint8))]
    for _ in range(12):
        m = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        at = int(rng.integers(0, len(m)))
        m[at:at] = bytes.fromhex(["f4ffff", "fcffffffff", "ff00900000", "fe0090"][int(rng.integers(0, 4))])
        cases.append(bytes(m))
    for body in cases:
        got = sm.fragment_boundaries(body)
        starts, end, total = _truth(body)
        assert not got["refused"]
        assert got["end"] == end and got["total"] == total and got["starts"] == starts


def test_chain_in_groups_equals_the_sequential_chain():
    """snappy_stream_chain_groups / _link: groups walked on an assumed entry, linked, re-walked where
    the assumption fails -- same entries, leaves and end as the one-pass chain; with small groups on
    text (the assumption mostly holds), literal-heavy and damaged bodies (it mostly does not)."""
    rng = np.random.default_rng(31)
    good, _ = _body(oracle.Port().compress(_mixed(4, 400000), 16))
    text, _ = _body(oracle.Port().compress(bytes(api.generate_host(0, 0xC5A90001, 0, 8, 65536)), 16))
    bodies = [good, text, good[:len(good) - 33], bytes(rng.integers(0, 256, 70000, dtype=np.uint8))]
    for _ in range(4):
        m = bytearray(good)
        m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        at = int(rng.integers(0, len(m)))
        m[at:at] = bytes.fromhex(["f4ffff", "fe0090"][int(rng.integers(0, 2))])
        bodies.append(bytes(m))
    kept = redone = 0
    for body in bodies:
        esz, l, segs = sm.build_index(body)
        want = sm.chain(body, esz, segs)
        for group in (2, 5, 64):
            entry, leave, end, again = sm.chain_in_groups(body, esz, segs, group)
            assert (entry, leave, end) == want, (len(body), group)
            ngroups = (len(segs) + group - 1) // group
            redone += again
            kept += max(ngroups - 1, 0) - again
    assert kept > 0 and redone > 0  # both outcomes of the link are exercised


def test_tags_at_matches_the_reference_char_table_semantics():
    """every tag byte with a fixed trailer: element sizes as csnappy_decompress.c:152-185 / :348-365"""
    trailer = bytes([0x11, 0x22, 0x33, 0x44, 0x55, 0, 0, 0])
    for t in range(256):
        esz, l = sm.tags_at(bytes([t]) + trailer)
        k, up = t & 3, t >> 2
        if k == 0:
            if up < 60:
                want = (1 + up + 1, up + 1)
            else:
                x = up - 59
                ln = int.from_bytes(trailer[:x], "little") + 1
                want = (1 + x + ln, min(ln, sm.HUGE))
        elif k == 1:
            want = (2, 4 + (up & 7))
        else:
            want = (3 if k == 2 else 5, up + 1)
        assert (int(esz[0]), int(l[0])) == want, t
    esz, l = sm.tags_at(bytes.fromhex("fcffffffff00"))
    assert int(esz[0]) == 5 and int(l[0]) == 0          # the 4-byte length ffffffff wraps to a zero-length literal
