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

Where the rule lives (recalled from LLVM's AMDGPU backend; the image ships no LLVM sources or ISA manual to quote): GCNHazardRecognizer::
checkVALUHazards, the `ST.hasDstSelForwardingHazard()` block -- "VALU writes vdst with dst_sel / op_sel[3], next VALU reads it: 1 wait state"
-- decides "op_sel[3]" by `src0_modifiers & SISrcMods::DST_OP_SEL`; SIDefines.h has OP_SEL_1 = DST_OP_SEL = 1 << 3, and for a VOP3P instruction
bit 3 of src0_modifiers is op_sel_hi[0].  gfx940/gfx950's "VALU write SGPR/VCC -> VMEM/v_readlane", trans-op and VMEM-store-data rules are
other functions of the same recogniser and their wait states are not touched here.

A wait state can also be the LAST slot of a longer hazard counted from an earlier instruction (a VMEM / FLAT store of more than 64 bits
needs 2 wait states before a VALU overwrites its data registers; a VALU that wrote an SGPR / VCC, a trans op or a readlane a few slots back have
their own counts).  So the `s_nop 0` is kept whenever one of the four compiler-emitted instructions before the packed producer is such an
instruction (_LONG_HAZARD_SOURCE): the pass only removes a wait state that nothing but the forwarding rule can have asked for.

Scope: rsrl_amd/_build.py applies the pass only to the translation units where its gain was measured (NOP_FILTER_SOURCES) and only under the
`hipcc --version` it was validated with (anything else builds in one plain hipcc call).  Evidence that the hardware interlocks the pair:
scripts/ubench/pk_forward.hip (tests/test_gpu_round5.py: 10^6 random producer -> consumer pairs back to back inside one asm statement, with and
without the wait state, bit for bit), the GPU suites (every kernel family bitwise against the oracle with the pass on) and scripts/ab_bits.py.
tests/test_abi_cpu.py checks the pass itself and the per-translation-unit removal counts on CPU.  RSRL_NOP_FILTER=0 builds without it (A/B)."""
import re

_PK = re.compile(r"v_pk_(?:fma|mul|add)_f32\s+v\[(\d+):(\d+)\]")
_NOT_PLAIN = re.compile(r"v_(?:readlane|readfirstlane|writelane|div_fmas|permlane)")
# instructions whose own hazards span more than one slot and could own the wait state: wide stores (data registers live for 2 wait states), VALU
# that writes an SGPR / VCC / EXEC (compares, carries, readlanes, div_scale), transcendental ops, MFMA, any VALU whose destination is an SGPR / VCC / EXEC, scalar writes of exec
_LONG_HAZARD_SOURCE = re.compile(
    r"(?:global|flat|scratch|buffer)_store_(?:dwordx[34]|b96|b128)|buffer_atomic|global_atomic|flat_atomic|ds_(?:write|store)_b(?:96|128)|"
    r"v_cmp|v_cmpx|v_readlane|v_readfirstlane|v_writelane|v_div_scale|v_add_co|v_sub_co|v_subrev_co|v_addc_co|v_subb_co|v_mad_u64_u32|v_mad_i64_i32|"
    r"v_(?:exp|log|rcp|rsq|sqrt|sin|cos)_|v_mfma|v_smfmac|v_permlane|s_\S*saveexec|s_.*\bexec\b|v_\S+\s+(?:s\[?\d|vcc|exec)")
LOOKBACK = 4


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
    hist = []              # the instructions before `prev`, newest last (at most LOOKBACK; emptied at labels)
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
                if plain and _reads(nxt, lo, hi) and not any(_LONG_HAZARD_SOURCE.match(h) for h in hist[-LOOKBACK:]):
                    removed += 1
                    continue
        if _is_instr(line):
            if in_app:
                asm_between += 1
                hist.append(t)                                  # (an inline-asm instruction can be a hazard source like any other)
            else:
                if prev is not None:
                    hist.append(prev)
                prev, asm_between = t, 0
            del hist[:-LOOKBACK - 1]
        elif t.endswith(":") and not t.startswith(";"):
            prev, asm_between = None, 0                         # block boundary: the previous instruction is not known
            hist = []
        out.append(line)
    return "\n".join(out), removed
