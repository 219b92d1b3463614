"""Pins the hand-counted `s_waitcnt vmcnt(N)` of the LDS-DMA ring kernels on the GENERATED gfx950 code (no GPU needed:
hipcc cross-compiles here).

k_fused_dma / k_sobel_dma / k_conv3x3_dma bring rows HBM -> LDS with inline-asm `global_load_lds_*` instructions, which hipcc does not
count; the kernels wait with their own `s_waitcnt vmcnt(N)` (stream_core.h).  vmcnt retires in order, so a wait for the
row in ring slot S is correct iff, on EVERY path, at least N vector-memory instructions (LDS-DMA pieces of younger rows,
output stores) are issued between the last DMA into slot S and the wait.  If the compiler ever sank an output store below
the wait, dropped one, or the source miscounted, N would be too large: rare, timing-dependent corrupt rows.

The asm statements print markers into the assembly (`; gh_dma slot=S` on each DMA instruction, `; gh_wait slot=S` on each
wait).  This test compiles the three sources to device assembly with the library's own flags (`make isa`) and, for every
shipped k_*_dma instantiation, walks FORWARD over every feasible path of the generated code keeping the list of the
youngest vector-memory operations: at a `vmcnt(N)` wait for slot S no DMA into S may be among the N youngest.  A negative
control mutates the code (N + 1; an output store sunk below the wait) and must be caught.  Finally the shipped
libgrayhip.so must contain the same instruction skeleton as the analysed compile.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VMEM_RE = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")
BRANCH_RE = re.compile(r"^s_c?branch\w*\s+(\S+)")
VMCNT_RE = re.compile(r"vmcnt\((\d+)\)")


class Ins:
    __slots__ = ("text", "mn", "vmem", "dma_slot", "wait_n", "wait_slot", "target", "uncond", "label")

    def __init__(self, text):
        self.text = text
        code = text.split(";")[0].strip()
        self.mn = code.split()[0] if code else ""
        self.vmem = bool(VMEM_RE.match(self.mn))
        m = re.search(r"gh_dma slot=(\d+)", text)
        self.dma_slot = int(m.group(1)) if m else None
        self.wait_n = self.wait_slot = None
        if self.mn == "s_waitcnt":
            m = VMCNT_RE.search(code)
            if m:
                self.wait_n = int(m.group(1))
            m = re.search(r"gh_wait slot=(\d+)", text)
            if m:
                self.wait_slot = int(m.group(1))
        m = BRANCH_RE.match(code)
        self.target = m.group(1) if m else None
        self.uncond = self.mn in ("s_branch", "s_endpgm", "s_setpc_b64")
        self.label = None


def parse_functions(asm_text, want):
    """{function name: [Ins]} for the functions whose mangled name contains `want`; labels attached to the next instruction."""
    out, cur, name, pending = {}, None, None, []
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur, pending = m.group(1), ([] if want in m.group(1) else None), []
            continue
        if line.startswith(".Lfunc_end"):
            if cur is not None:
                out[name] = cur
            cur = None
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            pending.append(m.group(1))
            continue
        s = line.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        ins = Ins(s)
        if pending:
            ins.label = list(pending)
            pending = []
        cur.append(ins)
    return out


def label_index(prog):
    at = {}
    for i, ins in enumerate(prog):
        for lb in (ins.label or []):
            at[lb] = i
    return at


SREG_RE = re.compile(r"^(s\[\d+:\d+\]|s\d+|vcc|exec|vcc_lo|vcc_hi|m0|scc)$")


def dest_of(code):
    """First operand of a scalar / compare instruction = what it writes (good enough to invalidate tracked flags)."""
    parts = code.split(None, 1)
    if len(parts) < 2:
        return None
    first = parts[1].split(",")[0].strip()
    return first if SREG_RE.match(first) else None


def reg_span(r):
    m = re.match(r"^s\[(\d+):(\d+)\]$", r)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"^s(\d+)$", r)
    if m:
        return int(m.group(1)), int(m.group(1))
    return None


def regs_overlap(a, b):
    if a == b or {a, b} <= {"vcc", "vcc_lo", "vcc_hi"}:
        return True
    sa, sb = reg_span(a), reg_span(b)
    return bool(sa and sb and sa[0] <= sb[1] and sb[0] <= sa[1])


def check_waits(prog, queue_cap=16):
    """Forward walk over every FEASIBLE path.  The abstract state is (program counter, the youngest vector-memory operations
    still possibly outstanding -- tagged with the ring slot for DMA instructions --, the SGPR-pair flags whose value is a
    known constant).  hipcc compiles `if (a) {wait<0>} else {issue; wait<N>}` into flag registers (`s_mov_b64 s[a:b], -1 / 0`,
    `s_andn2_b64 vcc, exec, s[a:b]`, `s_cbranch_vccnz`): following those constants is what separates the real paths from
    combinations that cannot happen (take the `else` of the first test and skip the body guarded by the flag it set).
    At a `vmcnt(n)` wait with a gh_wait marker for slot S no DMA into S may be among the n youngest operations; the wait
    then retires everything older.  Returns (counted waits seen, failures)."""
    at = label_index(prog)
    failures, counted = [], set()
    start = (0, (), ())
    seen, stack = {start}, [start]

    def push(pc, q, flags):
        st = (pc, q, flags)
        if pc < len(prog) and st not in seen:
            seen.add(st)
            stack.append(st)

    while stack:
        pc, q, flags = stack.pop()
        ins = prog[pc]
        code = ins.text.split(";")[0].strip()
        fl = dict(flags)
        if ins.vmem:
            q = (q + (("D%d" % ins.dma_slot) if ins.dma_slot is not None else "X",))[-queue_cap:]
        elif ins.wait_n is not None:
            n = ins.wait_n
            if n and ins.wait_slot is None:
                failures.append(f"counted wait without a gh_wait marker: {ins.text}")
            if ins.wait_slot is not None:
                if n:
                    counted.add(pc)
                young = q[len(q) - n:] if n else ()
                if ("D%d" % ins.wait_slot) in young:
                    failures.append(f"`{ins.text}`: a DMA into slot {ins.wait_slot} is among the {n} youngest operations {list(q)}")
            q = q[len(q) - n:] if n else ()
        elif ins.mn == "s_mov_b64":
            ops = [o.strip() for o in code.split(None, 1)[1].split(",")]
            if ops[1] in ("-1", "0"):
                fl[ops[0]] = int(ops[1])
            elif ops[1] in fl and ops[0] != "exec":
                fl[ops[0]] = fl[ops[1]]
            else:
                fl.pop(ops[0], None)
        elif ins.mn in ("s_andn2_b64", "s_and_b64"):
            ops = [o.strip() for o in code.split(None, 1)[1].split(",")]
            known = None
            if ops[1] == "exec" and ops[2] in fl:
                v = fl[ops[2]]
                known = (0 if v == -1 else "nz") if ins.mn == "s_andn2_b64" else ("nz" if v == -1 else 0)
            fl.pop(ops[0], None)
            if known is not None:
                fl[ops[0]] = known
            fl.pop("scc", None)
        else:
            d = dest_of(code)
            if d is not None:
                for k in [k for k in fl if regs_overlap(k, d)]:
                    fl.pop(k)
            if ins.mn.startswith("v_cmp") or ins.mn.startswith("v_cmpx") or ins.mn.startswith("v_readfirstlane") or ins.mn.startswith("v_readlane"):
                fl.pop("vcc", None)
                m = re.match(r"^\S+\s+(s\d+|s\[\d+:\d+\])", code)          # VALU instructions with a scalar destination
                if m:
                    for k in [k for k in fl if regs_overlap(k, m.group(1))]:
                        fl.pop(k)
        flags2 = tuple(sorted(fl.items(), key=lambda kv: kv[0]))
        if ins.mn == "s_endpgm" or ins.mn == "s_setpc_b64":
            continue
        if ins.target is not None:
            assert ins.target in at, f"branch to unknown label {ins.target}"
            tgt = at[ins.target]
            if ins.mn == "s_branch":
                push(tgt, q, flags2)
                continue
            take = fall = True
            if ins.mn in ("s_cbranch_vccnz", "s_cbranch_vccz") and "vcc" in fl:
                nz = fl["vcc"] != 0
                take = nz if ins.mn == "s_cbranch_vccnz" else not nz
                fall = not take
            if take:
                push(tgt, q, flags2)
            if fall:
                push(pc + 1, q, flags2)
            continue
        push(pc + 1, q, flags2)
    return len(counted), failures


def skeleton(prog):
    toks = []
    for ins in prog:
        if ins.mn.startswith("global_load_lds_dwordx4"):
            toks.append("L4")
        elif ins.mn.startswith("global_load_lds_dword"):
            toks.append("L1")
        elif ins.vmem:
            toks.append("ST" if "store" in ins.mn else ins.mn)
        elif ins.wait_n is not None:
            toks.append(f"W{ins.wait_n}")
    return toks


def template_ints(name):
    return [int(v) for v in re.findall(r"Li(\d+)E", name)]


@pytest.fixture(scope="module")
def device_asm():
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    p = subprocess.run(["make", "isa"], cwd=ROOT, capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return {k: open(os.path.join(ROOT, "build_isa", k + ".s")).read() for k in ("sobel", "fused", "conv3x3")}


def dma_functions(device_asm):
    fns = {}
    fns.update(parse_functions(device_asm["sobel"], "k_sobel_dma"))
    fns.update(parse_functions(device_asm["fused"], "k_fused_dma"))
    fns.update(parse_functions(device_asm["conv3x3"], "k_conv3x3_dma"))
    return fns


def test_every_counted_wait_is_covered_on_every_path(device_asm):
    fns = dma_functions(device_asm)
    assert len(fns) >= 10, sorted(fns)          # 6+ Sobel ring depths / store policies, 12 fused (mode x depth x channels)
    for name, prog in fns.items():
        assert not any(i.mn.startswith("scratch_") for i in prog), f"{name}: register spills"
        checked, failures = check_waits(prog)
        assert checked >= 2, f"{name}: no counted waits found (markers missing?)"
        assert not failures, f"{name}:\n" + "\n".join(failures[:5])
        # the counts are the ones the source derives: A pieces per row (+1 output store per row in steady state), D - 1 rows
        t = template_ints(name)
        stores = 1
        if "k_sobel_dma4" in name:
            d, a, stores = t[0], 5, 4            # four strips per wave: 4 pieces + the halo dword, 4 stores per row
        elif "k_sobel_dma2" in name:
            d, a, stores = t[0], 3, 2            # two strips per wave: 2 pieces + the halo dword, 2 stores per row
        elif "k_sobel_dma" in name or "k_conv3x3_dma" in name:
            d, a = t[0], 2
        else:
            d, a = t[1], t[3] + 1
        allowed = {0, a * (d - 1), (a + stores) * (d - 1)}
        got = {i.wait_n for i in prog if i.wait_n is not None and i.wait_slot is not None}
        assert got <= allowed and (a + stores) * (d - 1) in got, f"{name}: vmcnt values {sorted(got)}, expected {sorted(allowed)}"
        # every DMA instruction carries its marker (none was emitted outside the helpers)
        assert all(i.dma_slot is not None for i in prog if i.mn.startswith("global_load_lds"))


def test_the_checker_catches_a_wrong_count(device_asm):
    """Negative control: the same code with one wait's N raised by one, or with an output store moved below the next wait,
    must be reported."""
    fns = dma_functions(device_asm)
    name = next(n for n in fns if "k_sobel_dmaILi3ELi0ELi2ELi1E" in n)
    prog = fns[name]
    w = next(i for i, ins in enumerate(prog) if ins.wait_n == 6)
    prog[w].wait_n = 7
    assert check_waits(prog)[1], "a too-large count was not detected"
    prog[w].wait_n = 6
    assert not check_waits(prog)[1]
    # sink a store below the following counted wait: find a store followed (linearly) by a counted wait and swap them past each other
    st = next(i for i in range(len(prog)) if prog[i].vmem and "store" in prog[i].mn
              and any(p.wait_n == 6 for p in prog[i:i + 400]))
    nxt = next(i for i in range(st, len(prog)) if prog[i].wait_n == 6)
    moved = prog[st]
    mutated = prog[:st] + prog[st + 1:nxt + 1] + [moved] + prog[nxt + 1:]
    # labels travel with instructions; the store had none of its own in straight-line code
    assert check_waits(mutated)[1], "a store sunk below the wait was not detected"


def test_shipped_library_has_the_same_skeleton(device_asm, tmp_path):
    """The analysis above runs on a fresh compile; the shipped libgrayhip.so must contain, per kernel, the same sequence of
    LDS-DMA / store / vmcnt-wait instructions."""
    lib = os.path.join(ROOT, "openmp-parallel-computing_amd", "lib", "libgrayhip.so")
    if not os.path.exists(lib) or not os.path.exists(OBJDUMP):
        pytest.skip("library or llvm-objdump missing")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp_path, capture_output=True, check=True)
    shipped = {}
    for f in sorted(os.listdir(tmp_path)):
        if not f.endswith("gfx950"):
            continue
        dis = subprocess.run([OBJDUMP, "-d", f], cwd=tmp_path, capture_output=True, text=True, check=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                cur = shipped.setdefault(m.group(1), []) if any(k in m.group(1) for k in ("k_sobel_dma", "k_fused_dma", "k_conv3x3_dma")) else None
                continue
            if cur is not None and line.strip():
                cur.append(Ins(line.split("//")[0].strip()))
    fns = dma_functions(device_asm)
    assert set(fns) == set(shipped), sorted(set(fns) ^ set(shipped))
    for name in fns:
        assert skeleton(fns[name]) == skeleton(shipped[name]), f"{name}: shipped code differs from the analysed compile (rebuild: make lib)"
