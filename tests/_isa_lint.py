"""ISA-level lint of the kernels that fill shared LDS tiles by LDS-DMA (`buffer_load_* ... lds`, `global_load_lds_*`).

Why it exists: an LDS-DMA load has no destination register, so hipcc orders it only in front of the issuing wave's OWN LDS
reads -- not in front of an `s_barrier` behind which OTHER waves read what it loads.  `csrc/head.hip` had such a barrier for
four rounds (a few wrong speaker ids per ~10 calls on 239 k rows: the ids of `tal/baseline/reconcile.py:76-85`,
profiles/r5_head_lds_dma_race.txt).  This module reads the gfx950 assembly hipcc emits (`-S --cuda-device-only`, no GPU
needed) and proves, per kernel, by a forward dataflow over the kernel's control-flow graph:

  at every `s_barrier`, every LDS-DMA load of this wave that may still be in flight was issued at most `max_age` barriers
  ago (max_age = -1: none may be in flight at all).

The state is the wave's queue of outstanding vector-memory operations (gfx9: loads and stores share `vmcnt` and retire in
issue order; the counter has 6 bits, so at most 63 are outstanding), each entry = (is LDS-DMA, barriers passed since issue).
`s_waitcnt vmcnt(N)` keeps the N youngest entries.  In `strict` mode only the waits the source wrote by hand (inside an
`;;#ASMSTART` ... `;;#ASMEND` region) count: the guarantee then does not lean on a wait the compiler happened to place.
Joins at block entries align the queues at their young end and take the worse entry; ages saturate, so the fixpoint is
reached in a few passes.

Second check (`m0_leaks`): the inline-asm LDS-DMA writes M0 without declaring it (hipcc: "inline asm clobber list contains
reserved registers: m0 ... may lead to undefined behaviour" -- the clobber buys nothing), so in a kernel whose asm writes M0 no
instruction outside the asm regions may touch M0.
"""
import re
import subprocess

HIPCC = "/opt/rocm/bin/hipcc"
CAP = 63          # outstanding vector-memory operations the 6-bit vmcnt can express
AGE_CAP = 3

_VM = re.compile(r"^(buffer_|global_|flat_|scratch_|tbuffer_|image_)")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")


def compile_to_asm(src, defines=()):
    """gfx950 device assembly of one .hip file (seconds; hipcc cross-compiles without a GPU)."""
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src]
    cmd += ["-D" + d for d in defines]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed on %s:\n%s" % (src, r.stderr[-2000:]))
    return r.stdout


class Inst:
    __slots__ = ("text", "op", "in_asm", "line")

    def __init__(self, text, in_asm, line):
        self.text, self.in_asm, self.line = text, in_asm, line
        self.op = text.split()[0]


def split_kernels(asm):
    """{kernel symbol: [(label or None, [Inst...]) basic blocks in layout order]} for every function of the listing."""
    kernels, cur, blocks, insts, label, in_asm = {}, None, None, None, None, False
    types = set(re.findall(r"^\s*\.type\s+([\w$.]+),@function", asm, re.M))
    for ln, raw in enumerate(asm.splitlines(), 1):
        s = raw.strip()
        if cur is None:
            m = _FUNC.match(raw)
            if m and m.group(1) in types:
                cur, blocks, insts, label, in_asm = m.group(1), [], [], None, False
            continue
        if s.startswith(".Lfunc_end"):
            blocks.append((label, insts))
            kernels[cur] = blocks
            cur = None
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = _LABEL.match(s)
        if m:
            blocks.append((label, insts))
            label, insts = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if s:
            insts.append(Inst(s, in_asm, ln))
    return kernels


def is_lds_dma(inst):
    t = inst.text
    return (inst.op.startswith("buffer_load") and re.search(r"\blds\b", t) is not None) or inst.op.startswith("global_load_lds")


def uses_lds_dma(blocks):
    return any(is_lds_dma(i) for _, b in blocks for i in b)


def _join(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if len(a) < len(b):
        a, b = b, a
    pad = len(a) - len(b)          # align at the young end (the end of the tuple)
    return tuple((x[0] or y[0], max(x[1], y[1])) for x, y in zip(a, ((False, 0),) * pad + b))


def barrier_violations(blocks, max_age, strict=True):
    """[(asm line, description)] for every s_barrier that an LDS-DMA load older than `max_age` barriers may still be in flight at."""
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab is not None}
    for _, b in blocks:
        for i in b:
            if i.op in ("s_swappc_b64", "s_setpc_b64", "s_call_b64"):
                return [(i.line, "indirect control flow (%s): the kernel cannot be analysed" % i.op)]
    n = len(blocks)
    state_in = [None] * n
    state_in[0] = ()
    work = [0]
    found = {}
    while work:
        bi = work.pop()
        st = state_in[bi]
        succ, falls = [], True
        for inst in blocks[bi][1]:
            op = inst.op
            if op == "s_waitcnt":
                m = _VMCNT.search(inst.text)
                if m and (inst.in_asm or not strict):
                    keep = int(m.group(1))
                    st = st[len(st) - keep:] if keep < len(st) else st
            elif op == "s_barrier":
                bad = [e for e in st if e[0] and e[1] > max_age]
                if bad:      # (block-entry states only ever grow towards the fixpoint: a violation found on the way stays one)
                    found[inst.line] = ("s_barrier with up to %d LDS-DMA load(s) of this wave possibly in flight that were "
                                        "issued more than %d barrier(s) ago" % (len(bad), max_age))
                st = tuple((d, min(a + 1, AGE_CAP)) for d, a in st)
            elif _VM.match(op):
                st = (st + ((is_lds_dma(inst), 0),))[-CAP:]
            elif op == "s_branch":
                succ.append((index[inst.text.split()[1]], st))
                falls = False
                break
            elif op.startswith("s_cbranch"):      # (a block is cut at labels only: the edge carries the state AT the branch)
                succ.append((index[inst.text.split()[-1]], st))
            elif op == "s_endpgm":
                falls = False
                break
        if falls and bi + 1 < n:
            succ.append((bi + 1, st))
        for s, st in succ:
            j = _join(state_in[s], st)
            if j != state_in[s]:
                state_in[s] = j
                if s not in work:
                    work.append(s)
    return sorted(found.items())


def m0_leaks(blocks):
    """[(asm line, text)]: instructions outside the hand-written asm regions that touch M0, in a kernel whose asm regions write M0."""
    writes = any(i.in_asm and re.match(r"s_mov_b32\s+m0\b", i.text) for _, b in blocks for i in b)
    if not writes:
        return []
    return [(i.line, i.text) for _, b in blocks for i in b if not i.in_asm and re.search(r"\bm0\b", i.text)]
