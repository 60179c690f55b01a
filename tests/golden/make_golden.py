#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the COMPILED REFERENCE (oracle/_ref).

Run in the build container, where /root/reference exists:
    python tests/golden/make_golden.py
Every expected value below comes from calling the reference's own csnappy_* functions
(oracle.Ref = ctypes binding of oracle/_ref/libcsnappy_ref.so, built from the reference
sources where they lie by oracle/Makefile).  The data files next to this script
(urls.10K, urls.10K.snappy, baddata3.snappy, unaligned_uint64_test.*.gz) are the reference's
own test fixtures (reference testdata/), copied verbatim as data.
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from csnappy_amd import api  # noqa: E402  (only the host workload generator is used)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def kat_inputs():
    """name -> bytes"""
    period20 = bytes(range(65, 85))
    return {
        "empty": b"",
        "a": b"a",
        "a*14": b"a" * 14,
        "a*15": b"a" * 15,
        "a*100": b"a" * 100,
        "zero*1000": bytes(1000),
        "abcd*16": b"abcd" * 16,
        "bytes0..255": bytes(range(256)),
        "bytes0..255*2": bytes(range(256)) * 2,
        "period20*15": period20 * 15,
        "zero*32768": bytes(32768),
        "zero*65536": bytes(65536),
    }


NEGATIVE = [  # (hex, dst_len)  SURVEY.md Appendix C
    ("", 64), ("ffffffffff01", 64), ("ffffffff7f", 64), ("80", 64), ("00", 64), ("05", 64),
    ("0308616263", 64), ("0308616263", 2), ("080861626301 00".replace(" ", ""), 64),
    ("08086162630104", 64), ("0b0061fe0100", 20), ("32c4666f6f6f6f6f6f", 50),
    ("0a08616263", 64), ("0208616263", 64),
    # a few more shapes: copy-4 tag, long literal forms, overrun inside a copy
    ("0c0c61626364fe0400", 12), ("0c0c616263640f04000000", 64), ("05f00461626364 65".replace(" ", ""), 64),
    ("03f4020061", 64), ("0400610500", 64),
    # a literal tag whose 4-byte length field is ffffffff: length + 1 wraps to 0 (a zero-length literal,
    # csnappy_decompress.c:368-375) -- alone, between two literals, in front of a copy, as the last tag
    ("00fcffffffff", 64), ("040c61626364fcffffffff", 64), ("0404 6162 fcffffffff 0463 64".replace(" ", ""), 64),
    ("080c61626364fcffffffff0d04", 64), ("0400 61 fcffffffff 0562".replace(" ", ""), 64),
    ("0400 61 fcffffffff 0500".replace(" ", ""), 64), ("04fcffffffff0c61626364", 3),
]


def main():
    R = oracle.Ref()
    G = {}
    urls = open(os.path.join(HERE, "urls.10K"), "rb").read()
    G["urls_sha256"] = sha(urls)

    G["urls_whole"] = {}
    for p in range(9, 17):
        c = R.compress(urls, p)
        G["urls_whole"][str(p)] = {"size": len(c), "sha256": sha(c)}

    G["urls_blocks"] = {}
    for name, block, p, mode in (("64k_p16", 65536, 16, oracle.STREAM), ("64k_p15", 65536, 15, oracle.STREAM),
                                 ("4k_p13", 4096, 13, oracle.FRAGMENT), ("32k_p15", 32768, 15, oracle.FRAGMENT)):
        outs = R.compress_blocks(urls, block, p, mode)
        G["urls_blocks"][name] = {"block": block, "p": p, "mode": mode, "lens": [len(o) for o in outs],
                                  "sha256": sha(b"".join(outs))}

    G["kats"] = {}
    for name, data in kat_inputs().items():
        G["kats"][name] = {"n": len(data), "p15": R.compress(data, 15).hex() if len(data) <= 1000 else None,
                           "p15_sha256": sha(R.compress(data, 15)), "p16_sha256": sha(R.compress(data, 16)),
                           "p9_sha256": sha(R.compress(data, 9))}

    G["max_compressed_length"] = {str(n): R.max_compressed_length(n)
                                  for n in (0, 1, 4096, 32768, 65536, 0xFFFFFFFF)}

    G["negative"] = []
    for hx, dst_len in NEGATIVE:
        raw = bytes.fromhex(hx)
        rc_len, val = R.get_uncompressed_length(raw)
        rc_dec, _ = R.decompress(raw, dst_len)
        e = {"hex": hx, "dst_len": dst_len, "get_len": [rc_len, val if rc_len > 0 else None], "decompress": rc_dec}
        if rc_len > 0:
            rc_nh, produced, body = R.decompress_noheader(raw[rc_len:], dst_len)
            e["noheader"] = [rc_nh, produced if rc_nh == 0 else None, body.hex() if rc_nh == 0 else None]
        G["negative"].append(e)

    bad = open(os.path.join(HERE, "baddata3.snappy"), "rb").read()
    rc, n = R.get_uncompressed_length(bad)
    G["baddata3"] = {"get_len": [rc, n], "decompress": R.decompress(bad, n)[0],
                     "noheader": R.decompress_noheader(bad[rc:], n)[0]}
    un_s = gzip.open(os.path.join(HERE, "unaligned_uint64_test.snappy.gz")).read()
    un_b = gzip.open(os.path.join(HERE, "unaligned_uint64_test.bin.gz")).read()
    rc, out = R.decompress(un_s, len(un_b))
    assert rc == 0 and out == un_b
    G["unaligned"] = {"snappy_sha256": sha(un_s), "bin_sha256": sha(un_b), "bin_len": len(un_b)}

    # synthetic workloads (csnappy_amd/csrc/workload_gen.h): pin inputs and reference outputs
    G["workloads"] = {}
    for name, kind, seed, block, nblocks, p, mode in (
            ("G_text_64k_p16", api.WG_TEXT, 0xC5A90001, 65536, 64, 16, oracle.STREAM),
            ("G_text_64k_p15", api.WG_TEXT, 0xC5A90001, 65536, 64, 15, oracle.STREAM),
            ("G_low_64k_p16", api.WG_LOW, 0xC5A90005, 65536, 64, 16, oracle.STREAM),
            ("G_page_4k_p13", api.WG_PAGE, 0xC5A90004, 4096, 1024, 13, oracle.FRAGMENT)):
        data = api.generate_host(kind, seed, 0, nblocks, block)
        outs = R.compress_blocks(data, block, p, mode)
        G["workloads"][name] = {"kind": kind, "seed": seed, "block": block, "nblocks": nblocks, "p": p,
                                "mode": mode, "input_sha256": sha(data), "lens": [len(o) for o in outs],
                                "sha256": sha(b"".join(outs)),
                                "ratio": round(sum(len(o) for o in outs) / len(data), 6)}

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote golden.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in G.items()})


if __name__ == "__main__":
    main()
