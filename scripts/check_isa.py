#!/usr/bin/env python3
"""ISA lint for librdgan_hip.so (gfx950): run by build() on the CPU box, fails the build on a hazard.

Why.  Several kernels of the bf16 slab family issue their weight / gate loads from inline asm with hand-counted
`s_waitcnt vmcnt(n)` (csrc/rdgan_upconv16.hip.h: rd_upc_wload / rd_upc_wait and friends), and every LDS-DMA stage
(`buffer_load ... lds`) is published by a barrier.  hipcc neither sees those loads nor pads hazards for them, and round 3 met
four silent bugs of this class:
  (1) a queue register was handed to another value while its load was still in flight (k_d2_dgrad_slab16: GPU memory fault),
  (2) copies of queue registers were placed in front of the wait that covers them (one phase wrong),
  (3) an asm load read its SGPR base right behind the SALU instruction that wrote it (0.3 % of the outputs wrong, now and then),
  (4) a barrier published an LDS-DMA tile that had not been waited for (k_g9_wgrad_mfma: run-to-run differences).
Each depends on hipcc's register allocation and scheduling, so a compiler update can re-introduce any of them with no source
change.  This script checks the machine code itself:

  rule I    no instruction reads or writes a VGPR that is the destination of a vector-memory load still in flight
            (in-order vmcnt FIFO, as LLVM models gfx9; explored over the kernel's control-flow graph, loops included);
  rule II   in the hand-scheduled kernels, >= 5 wait states between an SALU / VALU write of an SGPR and a global load that uses
            it as its base (everywhere: the architected VALU-writes-SGPR -> VMEM hazard);
  rule III  no LDS-DMA is in flight at an s_barrier (the bytes a barrier publishes have landed), except in kernels listed
            with the reason;
  rule IV   zero scratch / spilled registers in the kernels DESIGN.md claims it for, and <= 256 VGPRs + AGPRs where two
            workgroups per CU are assumed.

Usage:  check_isa.py [path/to/librdgan_hip.so]      exit code 0 = clean
        check_isa.py --selftest                      the four historical bugs re-introduced into HEAD's own disassembly must
                                                     each be flagged (tests/test_isa_lint.py runs this)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("RDGAN_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "pr_disagg_radar_gan_amd", "librdgan_hip.so")

# kernels whose loads are issued from inline asm with hand-counted waits (rule II applies to SALU writers there too)
HAND_SCHEDULED = re.compile(r"k_conv_gemm_f16|k_upconv_slab16|k_upconv2_slab16|k_d2_dgrad_slab16|k_d2_dgrad_slab_t16|k_d2_fwd_slab16|k_upconv_slab_t16|k_upconv2_slab_t16")
# kernels DESIGN.md states run without scratch (sections 4.6-4.11): a spill there is a performance bug that looks like a result
NO_SPILL = re.compile(r"k_conv_gemm_f16|k_upconv_slab16|k_upconv2_slab16|k_d2_dgrad_slab16|k_d2_dgrad_slab_t16|k_d2_fwd_slab16|k_upconv_wgrad_slab16|k_upconv2_wgrad_slab16|"
                      r"k_d2_wgrad_slab16|k_d3_wgrad_slab16|k_d1_fwd_sample16|k_g9_bwd_mfma16|k_d1_dgrad_sample16|k_d1_wgrad16|"
                      r"k_upconv_slab_t16|k_upconv2_slab_t16")
# scratch a NO_SPILL kernel may still use, with the reason (bytes)
SCRATCH_ALLOWED = {
    r"^k_upconv2_slab16$": 16,     # three per-sample addresses stored in the prologue and reloaded once per sample, outside the K loop
}
# rule III exceptions: kernel regex -> (K, why): LDS-DMA loads may be in flight at a barrier if every one of them is among the K
# youngest vector-memory operations of the wave (a stage ring: the barrier publishes an OLDER stage; the counted wait in front of
# it is what the rule then checks, on every path)
DMA_ACROSS_BARRIER = {
    r"k_wgrad_gemm_ws16ILi256ELi128E": (12, "three LDS stages: chunk q + 2's 8 + 4 DMAs stay in flight while the barrier publishes chunk q + 1"),
}


class Insn:
    __slots__ = ("addr", "mn", "ops", "text", "target")

    def __init__(self, addr, mn, ops, text, target=None):
        self.addr, self.mn, self.ops, self.text, self.target = addr, mn, ops, text, target


_REG = re.compile(r"^(v|s|a|ttmp)(\d+)$|^(v|s|a|ttmp)\[(\d+):(\d+)\]$")


def regs_of(op):
    """register operand -> set of names like 'v12', 's4', 'a0', 'vcc'; anything else -> empty"""
    op = op.strip()
    if op.startswith("-") or op.startswith("|"):
        op = op.strip("-|")
    m = _REG.match(op)
    if m:
        if m.group(1):
            return {m.group(1) + m.group(2)}
        return {m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1)}
    if op in ("vcc", "vcc_lo", "vcc_hi", "exec", "exec_lo", "exec_hi", "m0", "scc"):
        return {op.split("_")[0]}
    return set()


def parse_disassembly(text):
    """llvm-objdump -d text -> {kernel symbol: [Insn]}"""
    kernels, cur, base = collections.OrderedDict(), None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            base = int(m.group(1), 16)
            cur = kernels.setdefault(m.group(2), [])
            continue
        if cur is None or "//" not in line:
            continue
        code, _, comment = line.partition("//")
        code = code.strip()
        if not code:
            continue
        ma = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not ma:
            continue
        addr = int(ma.group(1), 16)
        mn, _, rest = code.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
        target = None
        mt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", comment)
        if mn.startswith("s_cbranch") or mn == "s_branch":
            if mt:
                target = base + int(mt.group(1), 16)
            elif re.search(r"<[^>+]+>\s*$", comment):
                target = base
        cur.append(Insn(addr, mn, ops, code, target))
    return kernels


def is_vmem(i):
    return i.mn.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store",
                            "scratch_load", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic"))


def vmem_kind(i):
    """'dma' (buffer_load ... lds: no VGPR destination), 'load', 'store'"""
    if "load" in i.mn and any(o.split()[-1] == "lds" or o == "lds" for o in i.ops):
        return "dma"
    if "load" in i.mn or ("atomic" in i.mn and any("glc" in o or "sc0" in o for o in i.ops)):
        return "load"
    return "store"


def vmem_dest(i):
    return regs_of(i.ops[0]) if vmem_kind(i) == "load" and i.ops else set()


def reads_writes(i):
    """(read registers, written registers) of a non-VMEM instruction, conservatively (a destination that is also a source --
    v_fmac, v_mfma with src C -- shows up in both through the operand list)."""
    ops = [o.split()[0] if o else o for o in i.ops]       # drop modifiers such as 'offset:32', 'row_shr:1'
    rs, ws = set(), set()
    mn = i.mn
    if mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_setprio", "s_sleep", "s_branch", "s_cbranch", "s_setreg")):
        return rs, ws
    nodst = mn.startswith(("ds_write", "ds_store", "s_cmp", "s_bitcmp", "v_cmpx", "s_store", "ds_gws", "s_sendmsg", "s_icache", "s_dcache"))
    two_dst = mn.startswith(("v_swap_b32", "v_permlane32_swap", "v_permlane16_swap"))
    for k, o in enumerate(ops):
        r = regs_of(o)
        if not r:
            continue
        if k == 0 and not nodst:
            ws |= r
            if two_dst:
                rs |= r
        elif k == 1 and two_dst:
            ws |= r; rs |= r
        elif k == 1 and "_co_" in mn:           # carry out
            ws |= r
        else:
            rs |= r
    if mn.startswith(("v_fmac", "v_mac", "v_dot", "v_pk_fmac")) and ops:
        rs |= regs_of(ops[0])
    return rs, ws


def wait_states(i):
    if i.mn == "s_nop":
        try:
            return int(i.ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def check_kernel(name, insns, hand=False, dma_ok=None):
    """-> list of (rule, address, message)"""
    issues = []
    n = len(insns)
    index = {ins.addr: k for k, ins in enumerate(insns)}
    leaders = {0}
    for k, ins in enumerate(insns):
        if ins.target is not None:
            if ins.target in index:
                leaders.add(index[ins.target])
            if k + 1 < n:
                leaders.add(k + 1)
        if ins.mn == "s_endpgm" and k + 1 < n:
            leaders.add(k + 1)

    # ---- rule II (block-local look-back)
    for k, ins in enumerate(insns):
        if not (ins.mn.startswith("global_load") or ins.mn.startswith("global_store")):
            continue
        base = set()
        for o in ins.ops[1:]:
            r = regs_of(o.split()[0])
            if r and all(x.startswith("s") for x in r):
                base |= r
        if not base:
            continue
        ws_seen, j = 0, k - 1
        while j >= 0 and ws_seen < 5:
            p = insns[j]
            if (j + 1) in leaders and j + 1 != k:
                break                                         # (another path joins here; the look-back stays inside the block)
            writer_is_valu = p.mn.startswith("v_")
            writer_is_salu = p.mn.startswith("s_") and not p.mn.startswith(("s_nop", "s_waitcnt", "s_load", "s_buffer_load", "s_barrier",
                                                                            "s_cbranch", "s_branch", "s_setprio", "s_cmp", "s_bitcmp"))
            if writer_is_valu or (hand and writer_is_salu):
                _, w = reads_writes(p)
                if w & base:
                    issues.append(("II", ins.addr, f"{ins.text}: SGPR base {sorted(w & base)} written {ws_seen} wait state(s) earlier by "
                                                   f"'{p.text}' (needs 5)"))
                    break
            ws_seen += wait_states(p)
            if (j in leaders):
                break
            j -= 1

    # ---- rules I and III: vector-memory operations in flight, as a forward dataflow over the CFG.
    # State: {key: age}, key = a VGPR that is the destination of a load in flight, or "dma" (an LDS-DMA in flight), age = the
    # number of vector-memory operations issued after it.  vmcnt counts loads, stores and DMAs alike on gfx9 and they retire in
    # order, so `s_waitcnt vmcnt(N)` retires exactly the entries of age >= N.  At a join the YOUNGEST age of a key wins (the
    # conservative side: it retires last) -- the same merge LLVM's SIInsertWaitcnts performs, so compiler-generated code passes
    # by construction and anything flagged comes from what the compiler cannot see (inline asm, LDS-DMA + barrier protocols).
    AGE_CAP = 64
    blocks = sorted(leaders)
    bstart = {k: True for k in blocks}
    state_in = {0: {}}
    work = [0]
    reported = set()

    def merge_into(k, st):
        cur = state_in.get(k)
        if cur is None:
            state_in[k] = dict(st)
            return True
        changed = False
        for key, age in st.items():
            # the YOUNGEST age of a key wins (it retires last); "dma_old" = age of the oldest DMA in flight: the OLDEST wins
            if key not in cur or (cur[key] < age if key == "dma_old" else cur[key] > age):
                cur[key] = age
                changed = True
        return changed

    iters = 0
    while work:
        k = work.pop()
        iters += 1
        if iters > 200000:
            issues.append(("I", insns[k].addr, "dataflow did not converge"))
            break
        st = dict(state_in[k])
        first = True
        while k < n:
            if not first and k in bstart:
                if merge_into(k, st):
                    work.append(k)
                break
            first = False
            ins = insns[k]
            if ins.mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ins.text)
                if m:
                    keep = int(m.group(1))
                    old = st.get("dma_old")
                    st = {key: age for key, age in st.items() if age < keep}
                    if old is not None and old >= keep and "dma" in st:
                        st["dma_old"] = keep - 1              # (the DMAs still in flight are among the `keep` youngest operations)
            elif is_vmem(ins):
                kind = vmem_kind(ins)
                dest = vmem_dest(ins)
                used = set()
                for o in (ins.ops[1:] if kind == "load" else ins.ops):        # address / data registers (not the destination:
                    used |= regs_of(o.split()[0]) if o else set()             # a load into a register whose load is in flight
                clash = used & set(st)                                        # retires in order, the later one wins)
                if clash and (ins.addr, "v") not in reported:
                    reported.add((ins.addr, "v"))
                    issues.append(("I", ins.addr, f"'{ins.text}' uses {sorted(clash)}: destination of a load still in flight"))
                st = {key: age + 1 for key, age in st.items() if age + 1 < AGE_CAP}
                if kind == "dma":
                    st["dma"] = 0
                    st.setdefault("dma_old", 0)
                for r in dest:
                    st[r] = 0
            elif ins.mn == "s_barrier":
                if "dma" in st and dma_ok and st.get("dma_old", AGE_CAP) >= dma_ok[0] and (ins.addr, "b") not in reported:
                    reported.add((ins.addr, "b"))
                    issues.append(("III", ins.addr, f"s_barrier with an LDS-DMA load in flight that is not among the {dma_ok[0]} youngest vector-memory "
                                                    f"operations (oldest: {st.get('dma_old')} younger than it): the counted wait is missing on some path"))
                if "dma" in st and not dma_ok and (ins.addr, "b") not in reported:
                    reported.add((ins.addr, "b"))
                    issues.append(("III", ins.addr, f"s_barrier with an LDS-DMA load still in flight ({st['dma']} vector-memory operation(s) "
                                                    f"younger than it, no s_waitcnt vmcnt covers it on some path)"))
            elif st:
                rs, ws = reads_writes(ins)
                clash = (rs | ws) & set(st)
                if clash and (ins.addr, "r") not in reported:
                    reported.add((ins.addr, "r"))
                    what = "overwrites" if (ws & set(st)) else "reads"
                    issues.append(("I", ins.addr, f"'{ins.text}' {what} {sorted(clash)}: destination of a load still in flight"))
            if ins.mn == "s_endpgm" or ins.mn.startswith(("s_setpc", "s_swappc")):
                break
            if ins.target is not None:
                if ins.target in index and merge_into(index[ins.target], st):
                    work.append(index[ins.target])
                if ins.mn == "s_branch":
                    break
                if k + 1 < n and merge_into(k + 1, st):
                    work.append(k + 1)
                break
            k += 1
    return issues


def extract(lib):
    """-> (disassembly text, notes text) of the gfx950 code object inside the shared library"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fatbin"), os.path.join(td, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "copy.so")], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    return dis, notes


def parse_notes(notes):
    """kernel symbol -> dict(private_segment_fixed_size, vgpr_spill_count, sgpr_spill_count, vgpr_count, agpr_count, ...) from the
    amdhsa metadata note (llvm-readelf --notes prints it as YAML; a kernel's record starts at '  - .agpr_count:')"""
    out, cur = {}, None
    keys = ("agpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "sgpr_count",
            "group_segment_fixed_size", "max_flat_workgroup_size")
    for line in notes.splitlines():
        if re.match(r"^\s{2}-\s+\.agpr_count:", line):
            cur = {}
        if cur is None:
            continue
        m = re.match(r"^\s{2}(?:-\s|\s\s)\.(\w+):\s*(\S+)\s*$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k in keys:
            try:
                cur[k] = int(v)
            except ValueError:
                pass
        elif k == "symbol" and v.endswith(".kd"):
            out[v[:-3]] = cur          # (the dict keeps filling: .vgpr_count etc. follow .symbol in the record)
    return out


def demangled_name(sym):
    m = re.match(r"_Z(\d+)", sym)
    return sym[len(m.group(0)):len(m.group(0)) + int(m.group(1))] if m else sym


def lint(dis, notes, verbose=False):
    kernels = parse_disassembly(dis)
    meta = parse_notes(notes)
    bad = []
    for sym, insns in kernels.items():
        name = demangled_name(sym)
        hand = bool(HAND_SCHEDULED.search(name))
        dma_ok = next((why for rx, why in DMA_ACROSS_BARRIER.items() if re.search(rx, name) or re.search(rx, sym)), None)
        for rule, addr, msg in check_kernel(sym, insns, hand=hand, dma_ok=dma_ok):
            bad.append(f"[rule {rule}] {name} ({sym}) @0x{addr:x}: {msg}")
        md = meta.get(sym)
        if md is not None and NO_SPILL.search(name):
            allowed = max([v for rx, v in SCRATCH_ALLOWED.items() if re.search(rx, name)] + [0])
            # (an SGPR "spill" goes to the lanes of a VGPR, not to memory: counted by the metadata, free at run time)
            if md.get("private_segment_fixed_size", 0) > allowed:
                bad.append(f"[rule IV] {name} ({sym}): scratch {md.get('private_segment_fixed_size')} B, {md.get('vgpr_spill_count')} VGPR / "
                           f"{md.get('sgpr_spill_count')} SGPR spills (DESIGN.md claims none)")
    if verbose:
        nh = sum(1 for s in kernels if HAND_SCHEDULED.search(demangled_name(s)))
        print(f"check_isa: {len(kernels)} kernels ({nh} hand-scheduled), {sum(len(v) for v in kernels.values())} instructions, "
              f"{len(meta)} metadata records, {len(bad)} issue(s)")
    return bad, kernels, meta


# ---------------------------------------------------------------------------------------------------------------------------
# self test: the four historical bugs, re-introduced into HEAD's own machine code, must each be flagged
# ---------------------------------------------------------------------------------------------------------------------------
def _kernel_text(kernels, rx):
    for sym, insns in kernels.items():
        if re.search(rx, demangled_name(sym)):
            return sym, insns
    raise SystemExit(f"selftest: no kernel matches {rx}")


def _clone(insns):
    return [Insn(i.addr, i.mn, list(i.ops), i.text, i.target) for i in insns]


def selftest(lib):
    dis, notes = extract(lib)
    bad, kernels, meta = lint(dis, notes)
    if bad:
        print("selftest: HEAD is not clean:\n  " + "\n  ".join(bad[:10]))
        return 1
    failures = []

    def expect(label, sym, insns, rule, hand=True, dma_ok=None):
        got = [r for r, _, _ in check_kernel(sym, insns, hand=hand, dma_ok=dma_ok)]
        if rule not in got:
            failures.append(f"{label}: rule {rule} not raised (got {got})")
        else:
            print(f"selftest: {label}: flagged by rule {rule}")

    # (3) the asm load without its pad: drop every 's_nop 4' of the block-3 slab kernel
    sym, insns = _kernel_text(kernels, r"^k_upconv_slab16$")
    mut = [i for i in _clone(insns) if not (i.mn == "s_nop" and i.ops and i.ops[0] == "4")]
    assert len(mut) < len(insns), "selftest: no 's_nop 4' in k_upconv_slab16 (the asm pad is gone?)"
    expect("(3) asm load right behind the SALU write of its base", sym, mut, "II")

    # (1) a queue register overwritten while its load is in flight: a v_mov into the destination of the first counted-wait load,
    #     placed right behind the load
    mut = _clone(insns)
    k = next(k for k, i in enumerate(mut) if i.mn == "global_load_dwordx4" and any(x.mn == "s_waitcnt" and "vmcnt(6)" in x.text for x in mut[k:k + 400]))
    d = sorted(regs_of(mut[k].ops[0]), key=lambda r: int(r[1:]))[0]
    mut.insert(k + 1, Insn(mut[k].addr + 1, "v_mov_b32_e32", [d, "0"], f"v_mov_b32_e32 {d}, 0"))
    expect("(1) queue register re-used while its load is in flight", sym, mut, "I")

    # (2) a copy OF a queue register in front of the wait that covers it
    mut = _clone(insns)
    mut.insert(k + 1, Insn(mut[k].addr + 1, "v_mov_b32_e32", ["v255", d], f"v_mov_b32_e32 v255, {d}"))
    expect("(2) copy of a queue register in front of its wait", sym, mut, "I")

    # (4) a barrier publishing LDS-DMA bytes nobody waited for: drop the vmcnt waits between the last DMA and its barrier
    sym9, insns9 = _kernel_text(kernels, r"^k_g9_wgrad_mfma$")
    mut = _clone(insns9)
    last_dma = max(k for k, i in enumerate(mut) if is_vmem(i) and vmem_kind(i) == "dma")
    bar = next(k for k in range(last_dma, len(mut)) if mut[k].mn == "s_barrier")
    mut = mut[:last_dma + 1] + [i for i in mut[last_dma + 1:bar] if not (i.mn == "s_waitcnt" and "vmcnt" in i.text)] + mut[bar:]
    expect("(4) barrier with an LDS-DMA tile still in flight", sym9, mut, "III", hand=False)

    # round 4's kernels with counted waits:
    # (5) k_conv_gemm_f16: the weight queue's wait weakened by one k-step (vmcnt(14) -> vmcnt(16)): an MFMA reads a fragment in flight
    symf, insnsf = next((s_, i_) for s_, i_ in kernels.items() if "k_conv_gemm_f16ILb0E" in s_)
    mut = _clone(insnsf)
    n14 = 0
    for i in mut:
        if i.mn == "s_waitcnt" and "vmcnt(14)" in i.text:
            i.text = i.text.replace("vmcnt(14)", "vmcnt(16)"); n14 += 1
    assert n14 >= 4, "selftest: k_conv_gemm_f16 has no vmcnt(14) waits (the weight queue changed?)"
    expect("(5) fragment GEMM: weight-queue wait one k-step short", symf, mut, "I")
    # (6) its barrier publishing the next chunk's rows behind vmcnt(12) instead of vmcnt(8): four DMAs may still be in flight
    mut = _clone(insnsf)
    n8 = 0
    for k, i in enumerate(mut):
        if i.mn == "s_barrier" and mut[k - 1].mn == "s_waitcnt" and "vmcnt(8)" in mut[k - 1].text:
            mut[k - 1].text = mut[k - 1].text.replace("vmcnt(8)", "vmcnt(12)"); n8 += 1
    assert n8 >= 2, "selftest: k_conv_gemm_f16 has no vmcnt(8) + s_barrier pair"
    expect("(6) fragment GEMM: barrier with the chunk's DMAs not covered", symf, mut, "III")
    # (7) the three-stage weight-gradient kernel: the counted wait in front of its barrier weakened (vmcnt(12) -> vmcnt(20))
    symw, insnsw = next((s_, i_) for s_, i_ in kernels.items() if "k_wgrad_gemm_ws16ILi256ELi128E" in s_)
    okw = DMA_ACROSS_BARRIER[r"k_wgrad_gemm_ws16ILi256ELi128E"]
    mut = _clone(insnsw)
    n12 = 0
    for k, i in enumerate(mut):
        if i.mn == "s_barrier" and mut[k - 1].mn == "s_waitcnt" and "vmcnt(12)" in mut[k - 1].text:
            mut[k - 1].text = mut[k - 1].text.replace("vmcnt(12)", "vmcnt(20)"); n12 += 1
    assert n12 >= 1, "selftest: k_wgrad_gemm_ws16<256,128> has no vmcnt(12) + s_barrier pair"
    expect("(7) stage ring: an older stage's DMAs not covered at the barrier", symw, mut, "III", hand=False, dma_ok=okw)

    # (IV) scratch in a kernel that claims none: the metadata record of the block-3 slab kernel with 304 bytes of private segment
    marker = f".name:           {sym}\n    .private_segment_fixed_size: 0"
    assert marker in notes, "selftest: metadata record of k_upconv_slab16 not found"
    fake_notes = notes.replace(marker, marker[:-1] + "304", 1)
    if any("rule IV" in b and "k_upconv_slab16" in b for b in lint(dis, fake_notes)[0]):
        print("selftest: (IV) scratch / spilled registers in a kernel that claims none: flagged by rule IV")
    else:
        failures.append("(IV) scratch not flagged")
    if failures:
        print("selftest FAILED:\n  " + "\n  ".join(failures))
        return 1
    print("selftest ok")
    return 0


def main(argv):
    if "--selftest" in argv:
        lib = next((a for a in argv[1:] if not a.startswith("--")), DEFAULT_LIB)
        return selftest(lib)
    lib = next((a for a in argv[1:] if not a.startswith("--")), DEFAULT_LIB)
    dis, notes = extract(lib)
    bad, _, _ = lint(dis, notes, verbose=True)
    for b in bad:
        print(b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
