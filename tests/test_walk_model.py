"""The decompress scan's tag walk, four tags per dependent step (snappy_decompress_blocks in
csnappy_kernels.hip), restated in numpy and held against the plain sequential walk.

The kernel walks the table of 4th successors and marks the three tags in between from the tables
of 1st..3rd successors; a successor outside the window is lane 0, so a walk that left the window
goes on around the same chain and the first tag marked twice ends it.  The claim checked here: for
every table of element sizes the marked lanes are exactly the tags of the chain that starts at
lane 0, and `leave` is where that chain leaves the window."""
import numpy as np


def sequential(esz, wlim):
    tags, cur = [], 0
    while cur < wlim:
        tags.append(cur)
        cur += int(esz[cur])
    return tags, cur


def walk4(esz, wlim):
    lane = np.arange(64)
    nxt = lane + esz
    nxw = np.where(nxt < wlim, nxt, 0)
    n2 = nxw[nxw]
    n3 = nxw[n2]
    n4 = n2[n2]
    marks, cur, steps = set(), 0, 0
    while True:
        for _ in range(2):  # the kernel's asm block: two groups of four tags
            marks.update((cur, int(nxw[cur]), int(n2[cur]), int(n3[cur])))
            cur = int(n4[cur])
        steps += 8
        if len(marks) != steps:
            break
    last = max(marks)
    return sorted(marks), int(nxt[last])


def test_four_tag_walk_marks_exactly_the_chain():
    rng = np.random.default_rng(11)
    for trial in range(3000):
        kind = trial % 4
        if kind == 0:
            esz = rng.integers(1, 6, 64)  # copies
        elif kind == 1:
            esz = rng.integers(1, 70, 64)  # literals of any size
        elif kind == 2:
            esz = rng.choice([2, 3, 5, 9, 33, 61, 200], 64)
        else:
            esz = np.full(64, int(rng.integers(1, 65)))  # one size: every length of chain
        wlim = int(rng.integers(1, 65))
        tags, leave = sequential(esz, wlim)
        got, got_leave = walk4(esz, wlim)
        assert got == tags, (trial, wlim, esz.tolist())
        assert got_leave == leave


def test_float_reciprocal_remainder_is_exact_within_one_ulp_of_the_reciprocal():
    """The self-overlapping copy's `lane mod OFF` (the kernel needs lane < 64, OFF < 64; checked
    here far beyond that): quotient by a float reciprocal, one correction.  v_rcp_f32 is accurate
    to 1 ulp: every reciprocal within one ulp of the rounded one must still give the exact
    remainder."""
    j = np.arange(64 * 65, dtype=np.uint32)
    for off in range(1, 64 * 65):
        r0 = np.float32(1.0) / np.float32(off)
        for rcp in (np.nextafter(r0, np.float32(0)), r0, np.nextafter(r0, np.float32(2))):
            q = (j.astype(np.float32) * np.float32(rcp)).astype(np.uint32)  # v_cvt_u32_f32 truncates
            r = (j - q * np.uint32(off)).astype(np.uint32)
            got = np.minimum(r, (r - np.uint32(off)).astype(np.uint32))
            assert np.array_equal(got, j % off), (off, float(rcp))
