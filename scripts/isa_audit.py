#!/usr/bin/env python3
"""Static audit of a gfx950 assembly listing (hipcc -S) for the data hazards the hardware does NOT interlock and that hipcc's hazard
recogniser cannot see across an inline-asm boundary (the recogniser treats an asm statement as one opaque instruction).

Rules (LLVM GCNHazardRecognizer for gfx940-class parts; a "wait state" is one issued instruction, s_nop N = N + 1):
  R1  VALU writes VGPR   -> DPP instruction reads it as its DPP operand (src0)        2 wait states
  R2  VALU writes EXEC   -> DPP instruction                                           5
  R3  VALU writes SGPR   -> VALU reads that SGPR                                      2   (v_readlane / v_readfirstlane / v_cmp_e64 results)
  R4  VALU writes SGPR   -> v_readlane / v_writelane uses it as lane select           4
  R5  VALU writes VGPR   -> v_readlane / v_readfirstlane reads it                     1
  R6  trans op (v_rcp/v_rsq/v_sqrt/...) writes VGPR -> non-trans VALU reads it        1
  R7  VALU writes SGPR   -> VMEM/FLAT/DS/SMEM address uses that SGPR                  5

  P1  (not a hardware hazard: a register-allocator defect of this toolchain)  a per-lane instruction (VALU other than
      v_readlane / v_writelane, or a memory instruction) between a block label and the `s_or_b64 exec, exec, s[..]` that re-opens the lanes at the
      head of that block (the join of an if / else): it executes under the mask of ONE side of the branch -- possibly no lane at all -- so a live-range
      split copy or a spill placed there silently loses the other lanes' values.  Found as `v_accvgpr_write_b32 a0, v106` (the instance index) in
      front of the exec restore in one build of rti_solve_kernel<10, 64, 3>: its status / iteration / cost stores went to wrong addresses.

  P2  (the same kind: a second register-allocator defect of this toolchain)  a TORN TUPLE SPILL: a run of v_writelane_b32 that saves
      consecutive scalar registers into consecutive lanes of one VGPR (the spill of ONE wide scalar value) whose registers were last written by TWO different
      s_load instructions with overlapping destinations, the later one having overwritten part of the earlier one's destination before anything read it.
      Found in rti_solve_kernel<3, 32, 2>: `s_load_dwordx16 s[12:27], .. 0x1c8` (iters_acc, status_acc, ...) followed by a re-materialised
      `s_load_dwordx8 s[8:15], .. 0x248` (ep_min_margin .. trace), then s12..s27 spilled as one value: `iters_acc` became the pointer to the step counter.

The scan is linear over the listing (branches are not followed): a hazard window that straddles a taken backward branch is checked by
treating the loop body as falling through into itself once (labels 1: ... s_cbranch 1b inside an asm block are unrolled once).
Only pairs with at least one instruction INSIDE an asm block (between ;;#ASMSTART and ;;#ASMEND) are reported by default; --all reports
compiler-only pairs as well (there should be none: the compiler pads its own).

usage: isa_hazard_check.py file.s [--all] [--kernel SUBSTR]
"""
import re
import sys

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def regs(tok, kind):
    """register numbers of kind 'v' / 's' / 'a' named by one operand token"""
    out = []
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(kind + r"\[(\d+):(\d+)\]", tok)
    if m:
        out = list(range(int(m.group(1)), int(m.group(2)) + 1))
    else:
        m = re.fullmatch(kind + r"(\d+)", tok)
        if m:
            out = [int(m.group(1))]
    if kind == "s" and tok in ("vcc", "vcc_lo", "vcc_hi"):
        out = [106, 107]
    return out


def split_ops(rest):
    rest = rest.split(";")[0]
    # operands end where modifiers begin
    rest = re.split(r"\s+(?:row_|quad_perm|bank_mask|bound_ctrl|offset|wave_|row_mask|op_sel|neg_|clamp|mul:|div:|glc|slc|sc0|sc1|nt|gds|dst_sel|src0_sel|src1_sel)", rest)[0]
    return [o.strip() for o in rest.split(",") if o.strip()]


class Ins:
    __slots__ = ("op", "ops", "line", "in_asm", "ws", "text")


def parse(path, kernel=None):
    ins, in_asm, active = [], False, kernel is None
    for ln, raw in enumerate(open(path), 1):
        s = raw.strip()
        if kernel and re.match(r"^[_A-Za-z0-9.$]+:", s) and not s.startswith(".L"):
            active = kernel in s
        if not active:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s.startswith(";;#ASMEND"):
            in_asm = False; continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":") or re.match(r"^\d+:$", s):
            continue
        m = re.match(r"^([a-z_0-9]+)\s*(.*)$", s)
        if not m:
            continue
        i = Ins(); i.op = m.group(1); i.ops = split_ops(m.group(2)); i.line = ln; i.in_asm = in_asm; i.text = s
        i.ws = 1
        if i.op == "s_nop":
            try:
                i.ws = int(i.ops[0], 0) + 1
            except Exception:
                i.ws = 1
        ins.append(i)
    return ins


def is_valu(i):
    return i.op.startswith("v_")


def valu_vgpr_writes(i):
    if not is_valu(i) or not i.ops:
        return []
    if i.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return []
    if i.op.startswith("v_accvgpr_write"):
        return []
    return regs(i.ops[0], "v")


def valu_sgpr_writes(i):
    if not is_valu(i) or not i.ops:
        return []
    out = []
    if i.op.startswith(("v_readlane", "v_readfirstlane")):
        out += regs(i.ops[0], "s")
    if i.op.startswith("v_cmp") and not i.op.startswith("v_cmpx"):
        out += regs(i.ops[0], "s") if i.op.endswith("_e64") else [106, 107]
    if i.op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")) and len(i.ops) > 1:
        out += regs(i.ops[1], "s")
    return out


def writes_exec_valu(i):
    return i.op.startswith("v_cmpx") or (is_valu(i) and i.ops and i.ops[0] == "exec")


def prologue_findings(path, kernel=None):
    """P1: per-lane instructions in front of the exec restore at the head of a block that an s_cbranch_execz jumps to (the join of an if / else,
    reached with EXEC = 0 on that edge and with one side's lanes on the fall-through edge)"""
    lines = open(path).read().split("\n")
    label_at, targets, kern_of, cur, active = {}, set(), {}, None, kernel is None
    for ln, raw in enumerate(lines):
        s = raw.strip()
        m = re.match(r"^([_A-Za-z0-9.$]+):", s)
        if m:
            if not s.startswith(".L"):
                cur = m.group(1); active = kernel is None or kernel in cur
            label_at[m.group(1)] = ln
            kern_of[m.group(1)] = cur
        m = re.match(r"^s_cbranch_execz\s+(\S+)", s)
        if m and active:
            targets.add(m.group(1))
    out = []
    for lab in sorted(targets, key=lambda x: label_at.get(x, 0)):
        if lab not in label_at:
            continue
        seen = []
        for ln in range(label_at[lab] + 1, min(label_at[lab] + 40, len(lines))):
            s = lines[ln].strip()
            if not s or s.startswith(";") or s.startswith("."):
                if re.match(r"^\.L\S+:", s):
                    break
                continue
            op = s.split()[0]
            if re.match(r"^s_or_b64\s+exec,\s*exec,", s):
                for l2, t2 in seen:
                    out.append((kern_of[lab], lab, l2 + 1, t2, ln + 1, s))
                break
            if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) or ("exec" in s and op.startswith("s_")):
                break
            if (op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))) or \
                    op.startswith(("ds_", "global_", "flat_", "buffer_", "scratch_")):
                seen.append((ln, s))
    return out


def torn_spill_findings(path, kernel=None):
    """P2: see the module docstring.  Linear scan per kernel; last[s] = index of the s_load that last wrote scalar register s (None: something else did)."""
    lines = open(path).read().split("\n")
    out, cur, active = [], None, kernel is None
    last, loads, run = {}, [], []          # run = pending v_writelane entries (line, vgpr, sreg, lane, last def of sreg)

    def flush():
        nonlocal run
        groups, g = [], []
        for e in sorted(run, key=lambda e: (e[1], e[2] - e[3], e[2])):
            if g and (e[1] != g[-1][1] or e[2] - e[3] != g[-1][2] - g[-1][3] or e[2] != g[-1][2] + 1):
                groups.append(g); g = []
            g.append(e)
        if g:
            groups.append(g)
        for g in groups:
            if len(g) < 4:
                continue
            defs = {e[2]: e[4] for e in g}
            ids = sorted({d for d in defs.values() if d is not None})
            for ia in ids:
                for ib in ids:
                    A, B = loads[ia], loads[ib]
                    torn = B["tore"].get(ia, set())       # registers of A that B overwrote before anything had read them
                    if ib <= ia or not torn:
                        continue
                    keptA = [r for r in defs if defs[r] == ia]
                    tornB = [r for r in defs if defs[r] == ib and r in torn]
                    if keptA and tornB:
                        out.append((cur, g[0][0] + 1, f"s[{g[0][2]}:{g[-1][2]}] -> {g[0][1]} lanes {g[0][3]}..{g[-1][3]}", A["line"] + 1, A["text"], B["line"] + 1, B["text"],
                                    f"s[{min(tornB)}:{max(tornB)}]"))
        run = []

    for ln, raw in enumerate(lines):
        t = raw.strip()
        m = re.match(r"^([_A-Za-z0-9.$]+):", t)
        if m:
            if not t.startswith(".L"):
                flush(); cur = m.group(1); active = kernel is None or kernel in cur; last, loads = {}, []
            continue
        if not active or not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        ops = split_ops(t[len(op):])
        if op == "v_writelane_b32" and len(ops) >= 3 and regs(ops[1], "s") and re.fullmatch(r"\d+", ops[2]):
            sreg = regs(ops[1], "s")[0]
            run.append((ln, ops[0], sreg, int(ops[2]), last.get(sreg)))       # (saving a register is not a use of its value)
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm")) or len(run) > 64:
            flush()
        no_dst = op.startswith(("s_cmp", "s_bitcmp", "s_nop", "s_waitcnt", "s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setprio", "s_sleep", "global_store", "ds_write",
                                "buffer_store", "flat_store", "scratch_store"))
        for o in (ops if no_dst else ops[1:]):
            for r in regs(o, "s"):
                if last.get(r) is not None:
                    loads[last[r]]["read"].add(r)
        if op.startswith("s_load_dword") and ops:
            dst = regs(ops[0], "s")
            rec = dict(line=ln, text=t, regs=dst, read=set(), tore={})
            for ia, A in enumerate(loads):
                unread = (set(A["regs"]) & set(dst)) - A["read"]
                if unread:
                    rec["tore"][ia] = unread
            loads.append(rec)
            for r in dst:
                last[r] = len(loads) - 1
        elif not no_dst and ops and op.startswith(("s_", "v_readlane", "v_readfirstlane", "v_cmp")):
            for r in regs(ops[0], "s"):
                last[r] = None
    flush()
    return out


def main():
    path = sys.argv[1]
    report_all = "--all" in sys.argv
    kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else None
    ins = parse(path, kernel)
    found = 0

    def look_back(idx, need):
        """instructions within `need` wait states before ins[idx] (exclusive), nearest first, with the wait states between"""
        out, ws, k = [], 0, idx - 1
        while k >= 0 and ws < need:
            out.append((ins[k], ws))
            ws += ins[k].ws
            k -= 1
        return out

    for idx, i in enumerate(ins):
        checks = []     # (rule, need, predicate on producer)
        if "_dpp" in i.op and len(i.ops) >= 2:
            src = set(regs(i.ops[1], "v"))
            checks.append(("R1 VALU->DPP src", 2, lambda p, src=src: set(valu_vgpr_writes(p)) & src))
            checks.append(("R2 EXEC->DPP", 5, lambda p: writes_exec_valu(p)))
        if is_valu(i):
            sread = set()
            for o in i.ops[1:]:
                sread |= set(regs(o, "s"))
            if i.op.startswith(("v_readlane", "v_writelane")) and len(i.ops) >= 3:
                sel = set(regs(i.ops[2], "s"))
                checks.append(("R4 SGPR->lane select", 4, lambda p, sel=sel: set(valu_sgpr_writes(p)) & sel))
            if sread:
                checks.append(("R3 VALU SGPR->VALU read", 2, lambda p, sread=sread: set(valu_sgpr_writes(p)) & sread))
            if i.op.startswith(("v_readlane", "v_readfirstlane")) and len(i.ops) >= 2:
                src = set(regs(i.ops[1], "v"))
                checks.append(("R5 VGPR->readlane", 1, lambda p, src=src: set(valu_vgpr_writes(p)) & src))
            if not i.op.startswith(TRANS):
                vread = set()
                for o in i.ops[1:]:
                    vread |= set(regs(o, "v"))
                if "fmac" in i.op or "_dpp" in i.op and "fmac" in i.op:
                    vread |= set(regs(i.ops[0], "v"))
                checks.append(("R6 trans->VALU", 1, lambda p, vread=vread: p.op.startswith(TRANS) and set(valu_vgpr_writes(p)) & vread))
        if i.op.startswith(("global_", "flat_", "buffer_", "scratch_", "s_load", "s_buffer_load", "ds_")):
            sread = set()
            for o in i.ops:
                sread |= set(regs(o, "s"))
            if sread:
                checks.append(("R7 VALU SGPR->mem address", 5, lambda p, sread=sread: set(valu_sgpr_writes(p)) & sread))
        for rule, need, pred in checks:
            for p, ws in look_back(idx, need):
                if pred(p) and (report_all or p.in_asm or i.in_asm):
                    found += 1
                    print(f"{rule}: need {need}, have {ws}\n   producer  L{p.line}{' [asm]' if p.in_asm else ''}: {p.text}\n   consumer  L{i.line}{' [asm]' if i.in_asm else ''}: {i.text}")
    for kern, lab, l2, t2, ln, rest in prologue_findings(path, kernel):
        found += 1
        print(f"P1 per-lane instruction ahead of the exec restore of block {lab} (reached by s_cbranch_execz), in {kern}\n   L{l2}: {t2}\n   L{ln}: {rest}")
    for kern, ln, what, la, ta, lb, tb, torn in torn_spill_findings(path, kernel):
        found += 1
        print(f"P2 torn tuple spill in {kern}: L{ln} saves {what} as one value, but {torn} of it were overwritten before anything read them\n   L{la}: {ta}\n   L{lb}: {tb}")
    print(f"{len(ins)} instructions scanned, {found} finding(s)")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
