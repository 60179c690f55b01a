"""CPU tests that PIN THE ORACLE: the restatement in oracle/snappy_oracle.c must agree with
(1) the reference's own fixtures, (2) golden vectors generated from the compiled reference
(tests/golden/make_golden.py), (3) the compiled reference itself when oracle/_ref is present,
and the wave-step model of the compress kernel must agree with the oracle."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from csnappy_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def port():
    return oracle.Port()


def codecs():
    out = [oracle.Port()]
    if oracle.have_ref():
        out.append(oracle.Ref())
    return out


# ---- reference fixtures -----------------------------------------------------------------------
def test_shipped_golden_stream_is_p15(port, urls, golden_dir):
    """reference testdata/urls.10K.snappy == csnappy_compress(urls.10K, p=15)."""
    shipped = open(os.path.join(golden_dir, "urls.10K.snappy"), "rb").read()
    assert sha(urls) == GOLD["urls_sha256"]
    assert port.compress(urls, 15) == shipped
    rc, out = port.decompress(shipped, len(urls))
    assert rc == 0 and out == urls


@pytest.mark.parametrize("p", range(9, 17))
def test_urls_whole_file_every_table_power(port, urls, p):
    c = port.compress(urls, p)
    assert len(c) == GOLD["urls_whole"][str(p)]["size"]
    assert sha(c) == GOLD["urls_whole"][str(p)]["sha256"]
    rc, out = port.decompress(c, len(urls))
    assert rc == 0 and out == urls


@pytest.mark.parametrize("name", sorted(GOLD["urls_blocks"]))
def test_urls_block_modes(port, urls, name):
    g = GOLD["urls_blocks"][name]
    outs = port.compress_blocks(urls, g["block"], g["p"], g["mode"])
    assert [len(o) for o in outs] == g["lens"]
    assert sha(b"".join(outs)) == g["sha256"]


def test_unaligned_uint64_torture_stream(port, golden_dir):
    """reference Makefile:37-55: 2409 self-overlapping copies, 532 with offset < 8."""
    s = gzip.open(os.path.join(golden_dir, "unaligned_uint64_test.snappy.gz")).read()
    b = gzip.open(os.path.join(golden_dir, "unaligned_uint64_test.bin.gz")).read()
    assert sha(s) == GOLD["unaligned"]["snappy_sha256"] and sha(b) == GOLD["unaligned"]["bin_sha256"]
    rc, out = port.decompress(s, len(b))
    assert rc == 0 and out == b


def test_baddata3_fails_cleanly(port, golden_dir):
    bad = open(os.path.join(golden_dir, "baddata3.snappy"), "rb").read()
    rc, n = port.get_uncompressed_length(bad)
    assert [rc, n] == GOLD["baddata3"]["get_len"]
    assert port.decompress(bad, n)[0] == GOLD["baddata3"]["decompress"] == -5
    assert port.decompress_noheader(bad[rc:], n)[0] == GOLD["baddata3"]["noheader"]


# ---- golden vectors from the compiled reference -----------------------------------------------
def test_kats(port):
    from golden.make_golden import kat_inputs
    for name, data in kat_inputs().items():
        g = GOLD["kats"][name]
        assert len(data) == g["n"]
        if g["p15"] is not None:
            assert port.compress(data, 15).hex() == g["p15"], name
        for p in (9, 15, 16):
            assert sha(port.compress(data, p)) == g[f"p{p}_sha256"], (name, p)


def test_max_compressed_length(port):
    for n, want in GOLD["max_compressed_length"].items():
        assert port.max_compressed_length(int(n)) == want
        assert oracle.max_compressed_length(int(n)) == want


@pytest.mark.parametrize("e", GOLD["negative"], ids=lambda e: e["hex"] or "empty")
def test_negative_vectors(port, e):
    raw = bytes.fromhex(e["hex"])
    rc, val = port.get_uncompressed_length(raw)
    assert rc == e["get_len"][0]
    if rc > 0:
        assert val == e["get_len"][1]
    assert port.decompress(raw, e["dst_len"])[0] == e["decompress"]
    if "noheader" in e:
        rc_nh, produced, body = port.decompress_noheader(raw[rc:], e["dst_len"])
        assert rc_nh == e["noheader"][0]
        if rc_nh == 0:
            assert produced == e["noheader"][1] and body.hex() == e["noheader"][2]


@pytest.mark.parametrize("name", sorted(GOLD["workloads"]))
def test_workload_generators_and_reference_outputs(port, name):
    g = GOLD["workloads"][name]
    data = api.generate_host(g["kind"], g["seed"], 0, g["nblocks"], g["block"])
    assert sha(data) == g["input_sha256"], "workload generator changed (recipes are frozen)"
    outs = port.compress_blocks(data, g["block"], g["p"], g["mode"])
    assert [len(o) for o in outs] == g["lens"]
    assert sha(b"".join(outs)) == g["sha256"]
    # any block range is reproducible independently (what the multi-GPU sharding relies on)
    part = api.generate_host(g["kind"], g["seed"], 5, 3, g["block"])
    assert np.array_equal(part, data[5 * g["block"]:8 * g["block"]])


# ---- the port against the compiled reference, fuzzed ------------------------------------------
def _fuzz_inputs(seed, count, max_n):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        n = int(rng.choice([rng.integers(0, 70), rng.integers(0, max_n), rng.integers(32760, 32780),
                            rng.integers(65530, 65545)]))
        n = min(n, max_n)
        alpha = int(rng.choice([1, 2, 3, 4, 16, 256]))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            x = rng.integers(0, alpha, n, dtype=np.uint8)
        elif kind == 1:
            x = np.resize(rng.integers(0, alpha, int(rng.integers(1, 40)), dtype=np.uint8), n)
        else:
            x = rng.integers(0, alpha, n, dtype=np.uint8)
            for _ in range(5):
                if n > 100:
                    s = int(rng.integers(0, n - 50))
                    ln = int(rng.integers(4, min(3000, n - s)))
                    d = int(rng.integers(0, n - ln))
                    x[d:d + ln] = x[s:s + ln].copy()
        yield x, int(rng.integers(9, 17))


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_port_equals_compiled_reference_fuzz():
    P, R = oracle.Port(), oracle.Ref()
    for x, p in _fuzz_inputs(1, 600, 70000):
        a, b = P.compress(x, p), R.compress(x, p)
        assert a == b, (len(x), p)
        if len(x) <= 32768:
            assert P.compress_fragment(x, p) == R.compress_fragment(x, p)
        rc, out = P.decompress(a, len(x))
        assert rc == 0 and out == x.tobytes()
        rc2, out2 = R.decompress(a, len(x))
        assert rc2 == 0 and out2 == out


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_port_equals_compiled_reference_on_corrupted_streams():
    """Mutate valid streams; status codes (and output on success) must agree.  Streams whose tag
    header bytes are cut off by the end of input are excluded: the reference reads stale stack
    bytes there, so its result is undefined."""
    P, R = oracle.Port(), oracle.Ref()
    rng = np.random.default_rng(5)
    checked = 0
    for x, p in _fuzz_inputs(2, 300, 6000):
        c = bytearray(P.compress(x, p))
        if len(c) < 4:
            continue
        for _ in range(4):
            m = bytearray(c)
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            cut = int(rng.integers(0, 3))
            if cut and len(m) > cut:
                m = m[:-cut]
            cap = int(rng.choice([len(x), len(x) + 7, max(len(x) - 3, 0), 2 * len(x) + 64]))
            if _has_truncated_tag(bytes(m)):
                continue
            ra, oa = P.decompress(bytes(m), cap)
            rb, ob = R.decompress(bytes(m), cap)
            assert ra == rb, (bytes(m).hex()[:80], cap)
            if ra == 0:
                n = P.get_uncompressed_length(bytes(m))
                rn, prod, body = P.decompress_noheader(bytes(m)[n[0]:], cap)
                rn2, prod2, body2 = R.decompress_noheader(bytes(m)[n[0]:], cap)
                assert (rn, prod, body) == (rn2, prod2, body2)
            checked += 1
    assert checked > 500


def _body_has_truncated_tag(stream, i=0):
    """True if walking the tags from offset i hits a tag whose extra bytes run past the end of
    input (the reference then reads stale stack bytes: undefined)."""
    n = len(stream)
    while i < n:
        t = stream[i]
        k = t & 3
        if k == 0:
            ln = (t >> 2) + 1
            ex = ln - 60 if ln > 60 else 0
            if i + 1 + ex > n:
                return True
            if ex:
                ln = (int.from_bytes(stream[i + 1:i + 1 + ex], "little") + 1) & 0xffffffff  # uint32, as the reference
            i += 1 + ex + ln
        else:
            ex = (1, 2, 4)[k - 1]
            if i + 1 + ex > n:
                return True
            i += 1 + ex
    return False


def _has_truncated_tag(stream):
    """The same for a stream with its varint header (False if the header itself is bad)."""
    rc, _ = oracle.Port().get_uncompressed_length(stream)
    return rc >= 0 and _body_has_truncated_tag(stream, rc)


# ---- the wave-step algorithm of the HIP compress kernel ---------------------------------------
def test_wave_step_model_is_bit_exact(port, urls):
    import wave_model as wm
    frag = urls[100000:100000 + 32768]
    for p in (16, 13):
        assert wm.compress_fragment(frag, p) == port.compress_fragment(frag, p)
    for x, p in _fuzz_inputs(3, 120, 6000):
        x = x[:32768]
        for s_entries in (4, min(1 << (p - 1), 2048)):
            assert wm.compress_fragment(x.tobytes(), p, s_entries) == port.compress_fragment(x, p), \
                (len(x), p, s_entries)


def test_dense_multi_match_step_model_is_bit_exact(port, urls):
    """v2 of the model = what the kernel runs: dense steps that retire several copies, sparse
    steps after 32 fruitless probes, any lane-local match cap, any (even tiny) conflict filter."""
    import wave_model as wm
    frag = urls[200000:200000 + 32768]
    for p in (16, 15, 12):
        assert wm.compress_fragment_v2(frag, p) == port.compress_fragment(frag, p)
    rng = np.random.default_rng(8)
    for x, p in _fuzz_inputs(4, 160, 6000):
        x = x[:32768]
        s_entries = int(rng.choice([4, 64, min(1 << (p - 1), 1024)]))
        lm = int(rng.choice([4, 8, 16]))
        assert wm.compress_fragment_v2(x.tobytes(), p, s_entries, lm=lm) == port.compress_fragment(x, p), \
            (len(x), p, s_entries, lm)


def test_forwarding_step_model_is_bit_exact(port, urls):
    """v4 of the model = dense steps that do not stop at a lane sharing its slot with an earlier
    lane of the step: the lane's candidate is forwarded from that lane (table value / cut when there
    is none).  Both variants the kernel instantiates (LDS table: table value; global table: cut)."""
    import wave_model as wm
    frag = urls[300000:300000 + 32768]
    for p, fc in ((16, True), (13, False), (9, False)):
        assert wm.compress_fragment_v4(frag, p, fallback_cut=fc) == port.compress_fragment(frag, p)
    rng = np.random.default_rng(9)
    k = 0
    for x, p in _fuzz_inputs(6, 220, 5000):
        x = x[:32768]
        s_entries = int(rng.choice([4, 64, min(1 << (p - 1), 1024)]))
        lm = int(rng.choice([4, 8, 16]))
        k += 1
        assert wm.compress_fragment_v4(x.tobytes(), p, s_entries, lm=lm, fallback_cut=bool(k & 1)) == \
            port.compress_fragment(x, p), (len(x), p, s_entries, lm, k & 1)
    # heap-like pages: most lanes of a step share slots
    pages = api.generate_host(2, 0xC5A90004, 0, 48, 4096)
    for i in range(48):
        pg = pages[i * 4096:(i + 1) * 4096]
        for fc in (False, True):
            assert wm.compress_fragment_v4(bytes(pg), 13, fallback_cut=fc) == port.compress_fragment(pg, 13)


def test_lean_step_model_is_bit_exact(port, urls):
    """v5 of the model = round 3's step loop (parse_lean): the (s, q1) cursor, lane 0 insert-only,
    the per-lane next-stop table, visits of flagged and cap-length lanes, the END_A / END_B cursor
    update and the commit mask, restated on the CPU and held against the oracle."""
    import wave_model as wm
    frag = urls[400000:400000 + 32768]
    for p in (16, 13, 9):
        st = {}
        assert wm.compress_fragment_v5(frag, p, stats=st) == port.compress_fragment(frag, p)
        assert st["steps"] < len(frag) // 8
    rng = np.random.default_rng(10)
    visits = 0
    for x, p in _fuzz_inputs(7, 260, 5000):
        x = x[:32768]
        s_entries = int(rng.choice([4, 64, min(1 << (p - 1), 1024)]))
        lm = int(rng.choice([4, 8, 16]))
        st = {}
        assert wm.compress_fragment_v5(x.tobytes(), p, s_entries, stats=st, lm=lm) == \
            port.compress_fragment(x, p), (len(x), p, s_entries, lm)
        visits += st.get("visits", 0)
    assert visits > 0
    # every length around the margin and the first wave steps
    base = bytes(urls[5000:5400])
    for n in list(range(0, 100)) + [127, 128, 129, 191, 192, 193]:
        assert wm.compress_fragment_v5(base[:n], 12) == port.compress_fragment(np.frombuffer(base[:n], np.uint8), 12), n
    # runs (one slot on every lane) and heap-like pages
    for blob in (b"\0" * 5000, b"ab" * 3000, b"abcdefgh" * 700 + b"x" + b"abcdefgh" * 50):
        assert wm.compress_fragment_v5(blob, 14) == port.compress_fragment(np.frombuffer(blob, np.uint8), 14)
    pages = api.generate_host(2, 0xC5A90004, 0, 32, 4096)
    for i in range(32):
        pg = pages[i * 4096:(i + 1) * 4096]
        assert wm.compress_fragment_v5(bytes(pg), 13) == port.compress_fragment(pg, 13)


def test_round5_step_model_is_bit_exact(port, urls):
    """v6 of the model = round 5's step loop: slot sharing known exactly, every lane with a bucket
    flagged in the steps behind one that may have inserted a position >= the late threshold (moved
    down here so that whole fragments run "late"), the chain's first stop in lane 0's next-stop
    entry, the (pz, q1) cursor and the one-compare end of the scan -- held against the oracle."""
    import wave_model as wm
    frag = urls[400000:400000 + 32768]
    for p in (16, 13, 9):
        st = {}
        assert wm.compress_fragment_v6(frag, p, stats=st) == port.compress_fragment(frag, p)
        assert st["steps"] < len(frag) // 8
    # the end of a full fragment: the steps behind position 0x7fc1 (and a model in which every step is one)
    for off in (0, 17, 4099):
        frag = urls[off:off + 32768]
        assert wm.compress_fragment_v6(frag, 16) == port.compress_fragment(frag, 16)
        assert wm.compress_fragment_v6(frag[:9000], 16, late_pos=64) == port.compress_fragment(frag[:9000], 16)
    rng = np.random.default_rng(11)
    visits = 0
    for x, p in _fuzz_inputs(8, 260, 5000):
        x = x[:32768]
        lm = int(rng.choice([4, 8, 16]))
        late_pos = int(rng.choice([0x7fc1, 1000, 64]))
        st = {}
        assert wm.compress_fragment_v6(x.tobytes(), p, stats=st, lm=lm, late_pos=late_pos) == \
            port.compress_fragment(x, p), (len(x), p, lm, late_pos)
        visits += st.get("visits", 0)
    assert visits > 0
    # every length around the margin and the first wave steps (the one-compare end of the scan)
    base = bytes(urls[5000:5400])
    for n in list(range(0, 100)) + [127, 128, 129, 191, 192, 193]:
        assert wm.compress_fragment_v6(base[:n], 12) == port.compress_fragment(np.frombuffer(base[:n], np.uint8), 12), n
    # scan limits that fall on the 64th lane, the lane behind it, the last stride-1 probe
    big = bytes(urls[20000:21000])
    for n in range(64, 160):
        assert wm.compress_fragment_v6(big[:n], 16) == port.compress_fragment(np.frombuffer(big[:n], np.uint8), 16), n
    # runs (one slot on every lane), incompressible data (sparse steps to the end) and heap-like pages
    noise = np.random.default_rng(5).integers(0, 256, 6000, dtype=np.uint8).tobytes()
    for blob in (b"\0" * 5000, b"ab" * 3000, b"abcdefgh" * 700 + b"x" + b"abcdefgh" * 50, noise, noise[:1000] + b"q" * 300 + noise[:777]):
        assert wm.compress_fragment_v6(blob, 14) == port.compress_fragment(np.frombuffer(blob, np.uint8), 14)
    pages = api.generate_host(2, 0xC5A90004, 0, 32, 4096)
    for i in range(32):
        pg = pages[i * 4096:(i + 1) * 4096]
        assert wm.compress_fragment_v6(bytes(pg), 13) == port.compress_fragment(pg, 13)


# ---- batch drivers used by the GPU parity tests and the CPU baseline --------------------------
def test_batch_drivers_match_single_calls(port, urls):
    b = api.Batch.uniform(len(urls), 65536, device=None)
    out, out_len = oracle.batch_compress(port, urls, b.in_off, b.in_len, b.out_off, b.out_bytes, 16,
                                         oracle.STREAM, threads=4)
    g = GOLD["urls_blocks"]["64k_p16"]
    assert out_len.tolist() == g["lens"]
    cat = b"".join(bytes(out[int(o):int(o) + int(n)]) for o, n in zip(b.out_off, out_len))
    assert sha(cat) == g["sha256"]
    back, status, _ = oracle.batch_decompress(port, out, b.out_off, out_len, b.in_off, b.in_len,
                                              b.in_bytes, oracle.STREAM, threads=4)
    assert (status == 0).all() and bytes(back) == urls


def test_sanitizer_build_of_the_cpu_side(port, urls, tmp_path):
    """the reference's `make check_leaks` runs its tester under valgrind
    (reference Makefile:31-35); valgrind is not in the image, so the CPU restatement and the
    product's two host-arithmetic entry points (csnappy_host.c) are compiled with
    -fsanitize=address,undefined (oracle/Makefile: asan_check) and run over the KATs, the negative
    vectors and a fuzz set, every buffer malloc'ed at its exact size."""
    import struct
    import subprocess
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), os.path.join(root, "oracle", "asan_check")])
    recs = []

    def rec(kind, pc, data, want_rc, want):
        recs.append(struct.pack("<IiIiI", kind, pc, len(data), want_rc, len(want)) + bytes(data) + bytes(want))
    from golden.make_golden import kat_inputs
    for name, data in kat_inputs().items():
        for p in (15, 16, 9):
            rec(1, p, data, 0, port.compress(data, p))
        rec(2, 13, data[:32768], 0, port.compress_fragment(data[:32768], 13))
    for x, p in _fuzz_inputs(4242, 150, 70000):
        x = bytes(x)
        rec(1, p, x, 0, port.compress(x, p))
        rec(2, p, x[:32768], 0, port.compress_fragment(x[:32768], p))
    rec(1, 15, urls[:200000], 0, port.compress(urls[:200000], 15))
    for e in GOLD["negative"]:
        raw = bytes.fromhex(e["hex"])
        # (kind 5 carries the expected value in the p_or_cap field)
        val = e["get_len"][1] if e["get_len"][0] > 0 else 0
        rec(5, val - (1 << 32) if val >= (1 << 31) else val, raw, e["get_len"][0], b"")
        rec(3, e["dst_len"], raw, e["decompress"], b"")
        if "noheader" in e:
            body = raw[e["get_len"][0]:]
            rec(4, e["dst_len"], body, e["noheader"][0], bytes.fromhex(e["noheader"][2] or "") if e["noheader"][0] == 0 else b"")
    bad = open(os.path.join(HERE, "golden", "baddata3.snappy"), "rb").read()
    rec(3, GOLD["baddata3"]["get_len"][1], bad, GOLD["baddata3"]["decompress"], b"")
    path = tmp_path / "vectors.bin"
    path.write_bytes(b"".join(recs))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(root, "oracle", "asan_check"), str(path)], capture_output=True, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert b"asan_check ok" in r.stdout and str(len(recs)).encode() in r.stdout


# ---- framing format: the spec restatement, pinned by published vectors ----------
def test_framing_restatement_crc32c_known_answers_and_spec_streams(port):
    """oracle/frame.py restates google/snappy's framing_format.txt (no framing source exists in the
    reference).  CRC-32C known answers: RFC 3720 B.4 and the classic check value; the mask formula
    and the stream layout are checked on streams assembled here byte by byte from the spec."""
    from oracle import frame
    assert frame.crc32c(b"123456789") == 0xE3069283
    assert frame.crc32c(bytes(32)) == 0x8A9136AA
    assert frame.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert frame.crc32c(bytes(range(32))) == 0x46DD794E
    assert frame.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert frame.crc32c(b"") == 0
    assert frame.mask(0) == 0xA282EAD8 and frame.mask(0xE3069283) == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    comp = lambda x: port.compress(x, 16)
    dec = lambda b, n: port.decompress(b, n)
    ulen = lambda b: port.get_uncompressed_length(b)
    # empty input: the identifier alone
    assert frame.encode(b"", comp) == bytes.fromhex("ff060000734e61507059")
    # "abc" does not compress: one uncompressed chunk = 01 | len 7 | masked crc | abc
    crc = frame.mask(frame.crc32c(b"abc")).to_bytes(4, "little")
    want = bytes.fromhex("ff060000734e61507059") + b"\x01\x07\x00\x00" + crc + b"abc"
    assert frame.encode(b"abc", comp) == want
    assert frame.decode(want, dec, ulen) == (0, b"abc")
    # a compressible chunk: 00 | len | masked crc | snappy block; two chunks for 65537 bytes
    x = b"ab" * 40000
    f = frame.encode(x, comp)
    body = comp(x[:65536])
    assert f[10:14] == b"\x00" + (4 + len(body)).to_bytes(3, "little") and f[18:18 + len(body)] == body
    assert frame.decode(f, dec, ulen) == (0, x)
    # padding, a skippable chunk and a repeated identifier are ignored; unskippable is an error
    g = f[:10] + b"\xfe\x03\x00\x00xyz" + b"\x80\x01\x00\x00q" + f[:10] + f[10:]
    assert frame.decode(g, dec, ulen) == (0, x)
    assert frame.decode(f[:10] + b"\x02\x00\x00\x00" + f[10:], dec, ulen)[0] == frame.E_BAD_CHUNK
    assert frame.decode(f[10:], dec, ulen)[0] == frame.E_NO_IDENTIFIER
    assert frame.decode(f[:-1], dec, ulen)[0] == frame.E_BAD_CHUNK
    bad = bytearray(f)
    bad[14] ^= 1
    assert frame.decode(bytes(bad), dec, ulen)[0] == frame.E_CRC
    assert frame.decode(f, dec, ulen, dst_cap=len(x) - 1)[0] == frame.E_OUTPUT_INSUF


def test_foreign_and_regression_fixtures_decode_with_the_oracle(port, golden_dir):
    """The decoder fixtures round 3 added -- Google-snappy streams (tests/golden/make_foreign.py)
    and the stream on which round 2's soak found the stream index wrong -- against the
    restatement and, when present, the compiled reference (the GPU tests decode the same files)."""
    import hashlib
    from golden.make_foreign import foreign_inputs
    index = json.load(open(os.path.join(golden_dir, "foreign.json")))["streams"]
    inputs = foreign_inputs()
    codecs = [port] + ([oracle.Ref()] if oracle.have_ref() else [])
    for name, meta in index.items():
        stream = open(os.path.join(golden_dir, f"foreign_{name}.snappy"), "rb").read()
        assert hashlib.sha256(stream).hexdigest() == meta["stream_sha256"]
        assert hashlib.sha256(inputs[name]).hexdigest() == meta["input_sha256"]
        for c in codecs:
            assert c.decompress(stream, len(inputs[name])) == (0, inputs[name]), (name, c.kind)
    stream = gzip.open(os.path.join(golden_dir, "stream_index_sparse.snappy.gz")).read()
    outs = {c.kind: c.decompress(stream, 813547) for c in codecs}
    for kind, (rc, out) in outs.items():
        assert rc == 0 and hashlib.sha256(out).hexdigest().startswith("33bc95657da2b30e"), kind
