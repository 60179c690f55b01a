"""The CPU models DESIGN.md section 4.1 ("Ceiling of this design") argues from: tests/models/parser_models.c
replays the reference's probe loop over fragments of the bench workloads and measures what the three
restructurings of the parser proposed after round 3 would need.  The assertions are the statements the
document makes -- if a workload recipe or the model changes, the document has to change with it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("model") / "parser_models")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Wextra", "-Werror", "-o", exe,
                           os.path.join(ROOT, "tests", "models", "parser_models.c")])

    def run(workload, blocks=32):
        out = subprocess.run([exe, workload, str(blocks)], cwd=ROOT, capture_output=True, text=True, check=True).stdout
        nums = lambda line: [float(x) for x in re.findall(r"(?<![A-Za-z_])\d+(?:\.\d+)?", line)]
        lines = {ln.split()[0]: ln for ln in out.splitlines()}
        per = nums(lines["per_fragment"])
        return {"probes": per[0], "copies": per[1], "buckets": per[2], "useful": per[3],
                "live_avg": nums(lines["a_interval_colouring"])[0], "live_max": nums(lines["a_interval_colouring"])[1],
                "deep": dict(zip((1, 2, 3, 4, 5), nums(lines["b_links"]))),
                "members": dict(zip((2, 3, 4, 5), nums(lines["c_hot_cold"])[1::2]))}
    return run


def test_no_restructuring_of_the_table_reaches_five_kib(model):
    """32 waves per CU need <= 5 120 B of LDS per fragment: a table of <= 2 048 two-byte entries beside
    1 KiB of filters.  (a) the buckets LIVE at a time, (c) the buckets with >= 3 members, and the buckets
    in which a match is possible at all are each well above that on text and on urls.10K."""
    for w in ("text", "urls"):
        m = model(w)
        assert 4300 < m["buckets"] < 5200
        assert m["live_avg"] > 2560, "(a): interval colouring would need more than 5 KiB of table"
        hot = m["members"][3] + m["members"][4] + m["members"][5]
        assert hot > 2560, "(c): the buckets with three or more members alone exceed 5 KiB"
        assert m["useful"] > 3500, "pure-collision buckets are only a sixth of the buckets"


def test_links_leave_a_dependent_fetch_every_other_step(model):
    """(b): with K parse-independent predecessor links per position and only an 'inserted' bitmap in
    LDS, a probe whose candidate is deeper than K needs a dependent fetch.  A fragment is ~512 steps."""
    t, u = model("text"), model("urls")
    assert t["deep"][2] > 300 and t["deep"][3] > 200 and t["deep"][4] > 120
    assert u["deep"][3] > 300
    assert 8500 < t["probes"] < 10500 and 2700 < t["copies"] < 3400
