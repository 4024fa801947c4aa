"""CPU-side check of the gfx950 instruction stream of every kernel that stages a shared LDS tile by LDS-DMA: no `s_barrier`
may be reached with one of the wave's own LDS-DMA loads still in flight unless the pipeline was built for it (tests/_isa_lint.py).

The kernels, their pipeline depth (how many barriers a load may be older than the barrier it is still in flight at), and where
the wait in the source is:

  head_argmax_kernel   -1  nothing in flight at any barrier     csrc/head.hip      `s_waitcnt vmcnt(0)` in front of __syncthreads()
  gemm_glds_kernel     -1  nothing in flight at any barrier     csrc/gemm_f32.hip  the same
  gemm_w64_kernel       0  only the tile requested since the    csrc/gemm_w64.hip  `s_waitcnt vmcnt(PER_WAVE)`
                           previous barrier (kt + 2, third buffer)
  gemm_s64_kernel       1  the two youngest tiles (kt + 2,      csrc/gemm_s64.hip  `s_waitcnt vmcnt(2 * PER_WAVE)`
                           kt + 3 of four buffers)

The race this guards against gave wrong speaker ids (`tal/baseline/reconcile.py:76-85`) for four rounds, see
profiles/r5_head_lds_dma_race.txt; the ablation build that has it (-DHEAD_NO_DMA_WAIT) must fail here.
"""
import os
import re
from concurrent.futures import ThreadPoolExecutor

import pytest

from tests import _isa_lint as L

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tal_asrd_amd", "csrc")

# source file -> [(kernel-name pattern, max_age)]; every kernel with LDS-DMA in these files must match exactly one row
TABLE = {
    "head.hip": [(r"head_argmax_kernel", -1)],
    "gemm_f32.hip": [(r"gemm_glds_kernel", -1)],
    "gemm_w64.hip": [(r"gemm_w64_kernel", 0)],
    "gemm_s64.hip": [(r"gemm_s64_kernel", 1)],
}


@pytest.fixture(scope="module")
def listings():
    if not os.path.exists(L.HIPCC):
        pytest.skip("no hipcc")
    with ThreadPoolExecutor(len(TABLE)) as ex:
        asms = list(ex.map(lambda f: L.compile_to_asm(os.path.join(CSRC, f)), TABLE))
    return dict(zip(TABLE, asms))


def _dma_kernels(asm):
    return {n: b for n, b in L.split_kernels(asm).items() if L.uses_lds_dma(b)}


def _age_of(fname, kernel):
    rows = [a for pat, a in TABLE[fname] if re.search(pat, kernel)]
    assert len(rows) == 1, "kernel %s of %s uses LDS-DMA and has no row in the table" % (kernel, fname)
    return rows[0]


def test_no_other_source_uses_lds_dma():
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(CSRC, f)).read()
        if re.search(r"load_lds|offen lds", text):
            assert f in TABLE, "%s stages LDS tiles by LDS-DMA and is not covered by this lint" % f


def test_every_barrier_of_every_lds_dma_kernel_is_covered_by_a_hand_written_wait(listings):
    seen = 0
    for fname, asm in listings.items():
        ks = _dma_kernels(asm)
        assert ks, "no LDS-DMA kernel found in %s: the lint is not looking at what it should" % fname
        for name, blocks in ks.items():
            nb = sum(1 for _, b in blocks for i in b if i.op == "s_barrier")
            assert nb >= 2, (name, nb)
            v = L.barrier_violations(blocks, _age_of(fname, name), strict=True)
            assert not v, "%s (%s): %s" % (name, fname, v)
            seen += 1
    assert seen >= 40          # 2 head + 38 glds + 12 w64 + 4 s64 instantiations today


def test_asm_written_m0_is_not_touched_by_compiler_code(listings):
    for fname, asm in listings.items():
        for name, blocks in _dma_kernels(asm).items():
            assert not L.m0_leaks(blocks), (name, L.m0_leaks(blocks)[:3])
    # and the check sees something: the asm kernels do write M0 by hand
    ks = _dma_kernels(listings["gemm_w64.hip"])
    assert all(any(i.in_asm and i.text.startswith("s_mov_b32 m0") for _, b in blocks for i in b) for blocks in ks.values())


def test_the_round5_race_build_fails_the_lint():
    """-DHEAD_NO_DMA_WAIT is the kernel as it shipped for four rounds: it must fail even when the compiler's own waits count."""
    if not os.path.exists(L.HIPCC):
        pytest.skip("no hipcc")
    asm = L.compile_to_asm(os.path.join(CSRC, "head.hip"), defines=("HEAD_NO_DMA_WAIT",))
    ks = _dma_kernels(asm)
    assert len(ks) == 2
    for name, blocks in ks.items():
        assert L.barrier_violations(blocks, -1, strict=False), name
        assert L.barrier_violations(blocks, -1, strict=True), name


def _mutate(asm, pattern, repl):
    out, n = re.subn(pattern, repl, asm)
    assert n > 0, pattern
    return out


def test_lint_notices_a_wait_that_is_one_load_short(listings):
    """Counted waits: one more load left in flight than the pipeline allows = a load of the tile about to be read."""
    for fname in ("gemm_w64.hip", "gemm_s64.hip"):
        asm = listings[fname]
        counts = sorted({int(c) for c in re.findall(r";;#ASMSTART\n\ts_waitcnt vmcnt\((\d+)\)", asm)} - {0})
        assert counts, fname
        for c in counts:
            bad = _mutate(asm, r"(;;#ASMSTART\n\ts_waitcnt vmcnt\()%d\)" % c, r"\g<1>%d)" % (c + 1))
            hit = [n for n, b in _dma_kernels(bad).items() if L.barrier_violations(b, _age_of(fname, n), strict=True)]
            assert hit, (fname, c)
    # and without any hand-written wait every kernel fails
    for fname, asm in listings.items():
        bad = _mutate(asm, r";;#ASMSTART\n\ts_waitcnt vmcnt\(\d+\)( lgkmcnt\(0\))?\n", ";;#ASMSTART\n")
        for n, b in _dma_kernels(bad).items():
            assert L.barrier_violations(b, _age_of(fname, n), strict=True), (fname, n)


def test_lint_on_hand_made_listings():
    head = "\t.type\tk,@function\nk:\n"
    tail = "\ts_endpgm\n.Lfunc_end0:\n"
    dma = "\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds\n"
    wait = lambda n: ";;#ASMSTART\n\ts_waitcnt vmcnt(%d)\n;;#ASMEND\n" % n     # noqa: E731
    ks = lambda body: L.split_kernels(head + body + tail)["k"]               # noqa: E731
    # a loop: wait, barrier, request the next tile
    loop = dma + ".LBB0_1:\n" + wait(0) + "\ts_barrier\n" + dma + "\ts_cbranch_scc1 .LBB0_1\n" + wait(0) + "\ts_barrier\n"
    assert L.barrier_violations(ks(loop), -1) == []
    # the same loop with the epilogue barrier unguarded: a path-insensitive reader must flag it
    loop2 = dma + ".LBB0_1:\n" + wait(0) + "\ts_barrier\n" + dma + "\ts_cbranch_scc1 .LBB0_1\n\ts_barrier\n"
    assert len(L.barrier_violations(ks(loop2), -1)) == 1
    # a compiler wait does not count in strict mode, does in loose mode
    cw = dma + "\ts_waitcnt vmcnt(0)\n\ts_barrier\n"
    assert L.barrier_violations(ks(cw), -1, strict=True) and not L.barrier_violations(ks(cw), -1, strict=False)
    # depth-0 pipeline: the load requested since the last barrier may fly, the one before may not
    p0 = dma + "\ts_barrier\n" + dma + wait(1) + "\ts_barrier\n"
    assert L.barrier_violations(ks(p0), 0) == []
    p0bad = dma + "\ts_barrier\n" + dma + wait(2) + "\ts_barrier\n"
    assert len(L.barrier_violations(ks(p0bad), 0)) == 1
    # the state at a mid-block conditional branch travels with the edge, not the state at the end of the block
    mid = dma + "\ts_cbranch_scc1 .LBB0_2\n" + wait(0) + ".LBB0_2:\n\ts_barrier\n"
    assert len(L.barrier_violations(ks(mid), -1)) == 1
    # an ordinary load in front of the LDS-DMA loads is retired first (in-order queue)
    q = "\tglobal_load_dword v2, v[0:1], off\n" + dma + wait(1) + "\ts_barrier\n"
    assert len(L.barrier_violations(ks(q), -1)) == 1
    # M0: compiler code touching it in a kernel whose asm writes it
    m = ";;#ASMSTART\n\ts_mov_b32 m0, s4\n;;#ASMEND\n\ts_mov_b32 m0, s5\n"
    assert len(L.m0_leaks(ks(m))) == 1
