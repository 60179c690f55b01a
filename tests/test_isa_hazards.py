"""The parser's hand-written ISA loops (CSNAPPY_ISA_LOOP in csnappy_amd/csrc/csnappy_kernels.hip) spell their own
wait states out: the compiler's hazard recogniser does not look into inline asm.  This test assembles the
kernels (hipcc -S, no GPU needed) and checks the rules the blocks rely on, instruction by instruction, in program
order inside every inline-asm block of the parser kernels (gfx940 family):

  a vector instruction that writes an SGPR (v_cmp*, v_readlane, v_readfirstlane) must be followed by
    >= 2 other instructions before a vector instruction reads that SGPR as an operand,
    >= 4 before a v_readlane / v_writelane uses it as its lane select;
  a vector instruction that writes a VGPR must be followed by
    >= 2 other instructions before a DPP instruction reads that VGPR,
    >= 1 before a v_readlane reads it.

The check is a forward data-flow analysis over each block's own branches and labels (ages of vector-written
registers, the smaller age where paths meet)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "csnappy_amd", "csrc", "csnappy_kernels.hip")
HIPCC = "/opt/rocm/bin/hipcc"

KERNELS = ("snappy_parse_fragments_dense_lean", "snappy_parse_fragments_hash_lean", "snappy_parse_fragments_gtab")


def _regs(tok):
    out = []
    for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]", tok):
        out += [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    tok = re.sub(r"\b[vs]\[\d+:\d+\]", "", tok)
    out += re.findall(r"\b[vs]\d+\b", tok)
    if "vcc" in tok:
        out += ["vcc_lo", "vcc_hi"]
    return out


def _asm_blocks(text, kernel):
    body = text.split(f"\n{kernel}:", 1)[1].split(".end_amdhsa_kernel", 1)[0]
    for blk in re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", body, re.S):
        lines = [l.strip() for l in blk.split("\n") if l.strip() and not l.strip().startswith(";")]
        if len(lines) > 40:  # the step loops (the small blocks are single instructions with their own nops)
            yield lines


def _check(lines):
    """-> list of violations.  A forward data-flow analysis over the block's control flow: the state is, for every
    register a VECTOR instruction wrote, how many instructions have issued since (capped at CAP); states meet at
    labels by taking the smaller age."""
    CAP = 8
    ins, labels = [], {}
    for l in lines:
        m = re.match(r"^(\d+):$", l)
        if m:
            labels.setdefault(m.group(1), []).append(len(ins))
        else:
            ins.append(l)

    def target(i, ref):  # "13f" / "1b" seen from instruction i
        num, d = ref[:-1], ref[-1]
        cands = labels.get(num, [])
        if d == "f":
            return min(c for c in cands if c > i)
        return max(c for c in cands if c <= i)

    def meet(a, b):
        if a is None:
            return dict(b)
        out = dict(a)
        for k, v in b.items():
            out[k] = min(out.get(k, CAP), v)
        return out

    state_in = [None] * (len(ins) + 1)
    state_in[0] = {}
    bad = set()
    work = [0]
    while work:
        i = work.pop()
        if i >= len(ins) or state_in[i] is None:
            continue
        st = dict(state_in[i])  # keys: ("s", reg) / ("v", reg) -> age
        l = ins[i]
        op, _, rest = l.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        is_vec = op.startswith("v_")
        is_dpp = any(t in l for t in ("row_shr", "row_bcast", "wave_shr"))
        age = lambda kind, r: st.get((kind, r), CAP)
        if is_vec:
            if op.startswith(("v_readlane_b32", "v_writelane_b32")):
                for r in _regs(ops[2]):
                    if age("s", r) < 4:
                        bad.add(f"lane select {r} written by a vector instruction {age('s', r)} instructions before: {l}")
                if op.startswith("v_readlane_b32"):
                    for r in _regs(ops[1]):
                        if r.startswith("v") and age("v", r) < 1:
                            bad.add(f"v_readlane reads {r} written by the vector instruction in front of it: {l}")
                else:
                    for r in _regs(ops[1]):
                        if not r.startswith("v") or r.startswith("vcc"):
                            if age("s", r) < 2:
                                bad.add(f"{r} written by a vector instruction {age('s', r)} instructions before: {l}")
            else:
                for o in ops[1:]:
                    for r in _regs(o.split(" ")[0]):
                        if (not r.startswith("v") or r.startswith("vcc")) and age("s", r) < 2:
                            bad.add(f"{r} written by a vector instruction {age('s', r)} instructions before: {l}")
                if is_dpp:
                    for o in ops[1:3]:
                        for r in _regs(o.split(" ")[0]):
                            if r.startswith("v") and not r.startswith("vcc") and age("v", r) < 2:
                                bad.add(f"DPP reads {r} written by a vector instruction {age('v', r)} instructions before: {l}")
        # everything ages by the wait states this instruction is worth
        step = 1 + (int(rest.strip() or 0) if op.startswith("s_nop") else 0)
        st = {k: v + step for k, v in st.items() if v + step < CAP}
        if is_vec and ops:
            if op.startswith("v_cmp") and op.endswith("_e32"):
                st[("s", "vcc_lo")] = st[("s", "vcc_hi")] = 0
            else:
                for r in _regs(ops[0]):
                    st[("v" if r.startswith("v") and not r.startswith("vcc") else "s", r)] = 0
        elif op.startswith("s_") and ops and not op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_bitcmp")):
            for r in _regs(ops[0]):
                st.pop(("s", r), None)
        succ = []
        if op.startswith("s_branch"):
            succ = [target(i, ops[0])]
        elif op.startswith("s_cbranch"):
            succ = [target(i, ops[0]), i + 1]
        else:
            succ = [i + 1]
        for j in succ:
            m = meet(state_in[j], st)
            if m != state_in[j]:
                state_in[j] = m
                work.append(j)
    return sorted(bad)


@pytest.mark.skipif(not shutil.which(HIPCC) and not os.path.exists(HIPCC), reason="no hipcc")
def test_wait_states_of_the_hand_written_loops(tmp_path):
    out = tmp_path / "k.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-S", "--cuda-device-only",
                    "-I", os.path.join(ROOT, "include"), SRC, "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    text = out.read_text()
    seen = 0
    for k in KERNELS:
        desc = text.split(f"\n{k}:", 1)[1].split(".end_amdhsa_kernel", 1)[0]
        nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        # no kernel with such a loop may declare 61..64 VGPRs (the top four of a 64-register allocation) ...
        assert not 60 < nv <= 64, (k, nv)
        for lines in _asm_blocks(text, k):
            seen += 1
            bad = _check(lines)
            assert not bad, (k, bad[:8])
            # ... and a loop may use registers above v59 only in a kernel that declares more than 64 (see the
            # comment at the global-table kernel's clobber list)
            used = {int(r[1:]) for l in lines for r in _regs(l) if r.startswith("v") and not r.startswith("vcc")}
            assert max(used) <= 59 or nv > 64, (k, nv, sorted(used)[-4:])
    assert seen >= 4  # dense, dense with a spill-over, hash, global


def test_the_checker_sees_what_it_is_for():
    """the mistakes made while the loops were written, as the checker's own known answers"""
    assert _check(["v_cmp_ne_u32_e64 s[84:85], 0, v53", "v_lshrrev_b32_e32 v62, 22, v62",
                   "v_cndmask_b32_e64 v52, 0, v40, s[84:85]"])  # one instruction between: two are needed
    assert not _check(["v_cmp_ne_u32_e64 s[84:85], 0, v53", "v_lshrrev_b32_e32 v62, 22, v62", "v_mov_b32_e32 v61, v26",
                       "v_cndmask_b32_e64 v52, 0, v40, s[84:85]"])
    assert _check(["v_readlane_b32 s80, v35, 0", "s_nop 0", "s_cmp_lt_u32 s80, 64", "v_readlane_b32 s80, v35, s80"])
    assert not _check(["v_readlane_b32 s80, v35, 0", "s_nop 0", "s_cmp_lt_u32 s80, 64", "s_cbranch_scc0 9f", "1:",
                       "s_bitset1_b64 s[68:69], s80", "v_readlane_b32 s80, v35, s80", "s_nop 0", "s_cmp_lt_u32 s80, 64",
                       "s_cbranch_scc1 1b", "9:"])
    assert _check(["v_mov_b32_e32 v59, v1", "s_nop 0", "v_max_u32_dpp v59, v59, v59 row_shr:1 row_mask:0xf bank_mask:0xf"])
    assert not _check(["v_cmp_lt_u32_e32 vcc, v59, v1", "s_and_b64 s[70:71], vcc, s[60:61]",
                       "v_cndmask_b32_e64 v59, 0, v34, s[70:71]"])  # a scalar instruction takes the mask over
