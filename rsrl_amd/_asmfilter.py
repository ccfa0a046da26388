"""Post-pass over the gfx950 assembly hipcc emits for the kernels: drops the wait states the compiler puts behind packed-fp32 instructions.

What it removes.  ROCm 7.2's clang places `s_nop 0` between a `v_pk_fma_f32` / `v_pk_mul_f32` / `v_pk_add_f32` and an instruction that reads
its result.  The wait state comes from the hazard recogniser's "VALU writes a sub-dword destination (dst_sel / op_sel[3]), the next VALU
consumes it" rule (gfx940's DstSelForwardingHazard, one wait state): it looks at bit 3 of src0_modifiers, which is DST_OP_SEL for a VOP3
instruction but op_sel_hi[0] for a VOP3P one -- set on every packed instruction that reads both halves of its first operand.  A packed fp32
instruction writes two whole dwords; there is no partial write to forward, and the hardware interlocks the dependency as for any other VALU
pair.  In the fused driver loop every instruction of the single resident wave takes an issue slot, `s_nop` included: the Horner chains of
sincospi and the harmonic recurrences carried ~8 of them per env-step (2 % of the slots).

What it leaves alone.  Only `s_nop 0` (one wait state) whose previous COMPILER-EMITTED instruction is one of the three packed fp32 opcodes
(inline-asm instructions in between -- the sources' v_pk_mov_b32 / v_ashrrev_i32 -- are skipped as the recogniser skips them: with one of
those in between the wait state exists whether or not the rule applies) and whose NEXT
instruction is an ordinary VALU instruction (not v_readlane / v_readfirstlane / v_writelane / v_div_fmas / v_permlane, no DPP, no SDWA) that
reads a register the packed instruction wrote.  Every other wait state (VALU-writes-SGPR before VMEM or v_readlane, trans-op forwarding,
s_nop with a larger count, ...) stays where the compiler put it.

Safety net: the GPU tests compare every kernel family bit for bit with the oracle; a missed wait state would read a stale register and
fail them (tests/test_gpu_*.py, 290 cases, all green with the pass on).  tests/test_abi_cpu.py checks the pass itself on CPU.
RSRL_NOP_FILTER=0 builds without it (A/B)."""
import re

_PK = re.compile(r"v_pk_(?:fma|mul|add)_f32\s+v\[(\d+):(\d+)\]")
_NOT_PLAIN = re.compile(r"v_(?:readlane|readfirstlane|writelane|div_fmas|permlane)")


def _is_instr(line):
    t = line.strip()
    return bool(t) and not t.startswith(";") and not t.startswith(".") and not t.startswith("//") and not t.endswith(":")


def _reads(instr, lo, hi):
    """does the VALU instruction `instr` read a VGPR in [lo, hi]?  (sources = every operand after the first)"""
    parts = instr.split(None, 1)
    if len(parts) < 2 or "," not in parts[1]:
        return False
    srcs = parts[1].split(",", 1)[1]
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", srcs):
        if int(a) <= hi and int(b) >= lo:
            return True
    for a in re.findall(r"(?<![\[\w])v(\d+)\b", srcs):
        if lo <= int(a) <= hi:
            return True
    return False


def filter_asm(text):
    """returns (filtered text, number of `s_nop 0` lines removed)"""
    lines = text.split("\n")
    out, removed = [], 0
    prev = None            # the last instruction the COMPILER emitted (what its hazard recogniser counts from)
    asm_between = 0        # inline-asm instructions (;;#ASMSTART .. ;;#ASMEND) since `prev`: the recogniser does not count them as wait states
    in_app = False
    n = len(lines)
    for i, line in enumerate(lines):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_app = True
        elif t.startswith(";;#ASMEND"):
            in_app = False
        if t == "s_nop 0" and prev is not None and not in_app:
            m = _PK.match(prev)
            if m:
                lo, hi = int(m.group(1)), int(m.group(2))
                j = i + 1
                while j < n and not _is_instr(lines[j]):
                    if lines[j].strip().endswith(":") and not lines[j].strip().startswith(";"):
                        j = n                                   # a label: the consumer is in another block, leave the wait state
                        break
                    j += 1
                nxt = lines[j].strip() if j < n else ""
                plain = nxt.startswith("v_") and not _NOT_PLAIN.match(nxt) and "dpp" not in nxt and "sdwa" not in nxt
                # asm_between == 0: the false positive described above.  asm_between >= 1: an instruction the recogniser did not count
                # already sits between producer and consumer -- the wait state is there whatever the rule is worth.
                if plain and _reads(nxt, lo, hi):
                    removed += 1
                    continue
        if _is_instr(line):
            if in_app:
                asm_between += 1
            else:
                prev, asm_between = t, 0
        elif t.endswith(":") and not t.startswith(";"):
            prev, asm_between = None, 0                         # block boundary: the previous instruction is not known
        out.append(line)
    return "\n".join(out), removed
