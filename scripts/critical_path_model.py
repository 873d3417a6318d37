"""What the sweeps of the C2 kernel cost by their own instructions, against what they were measured to take (no GPU needed).
From the SHIPPED listing (build/mpc_api-gfx950.s) of rti_split_kernel<3, 3>: the factor sweep's loop body (compiled C++ around three asm blocks, two stages per pass) and the
hand-scheduled vector sweep's loop body (one asm block, four stages per pass) are counted by class and priced at what ONE wavefront per SIMD pays (micro-benchmarks
scripts/bin_src/dpp64_test.hip, lds_issue_test.hip: independent FP64 instruction 5.0 cycles, dependent one 8.4, LDS instruction 14 of issue):
    factor sweep   per stage = (5.0 x VALU + 14 x LDS) / 2                    (its DPP multiply-adds are interleaved so that none waits for its predecessor)
    vector sweep   per stage = (8.4 x chained multiply-adds + 5.0 x other VALU + 14 x LDS) / 4      (the 5 multiply-adds of a stage are one dependent chain)
and divided by the measured cycles per stage of the same phases (profiles/r05_c2_phase_timing.txt, -DMPC_PHASE_TIMING build, s_memtime around the phase).
Round 6 adds the DEPENDENCY FLOOR of the factor sweep (`dep_floor`): the loop body's register data-flow graph with every VALU result available 8.4 cycles
after its last operand and every LDS load 64 cycles after its address (unlimited issue width: no instruction waits for an issue slot), unrolled until the time per pass settles --
the recurrence through the cost-to-go matrix P~ that no schedule of THIS mapping can beat -- per stage, over the measured cycles per stage.  sweep_frac.factor says how far the
sweep is above its own ISSUE cost, dep_floor how far above its own DEPENDENCY chain: what lies between the two is what more parallel issue (a second wavefront per SIMD) could recover.
usage (after build()): python scripts/critical_path_model.py [tag]   -> profiles/<tag>_critical_path_c2.json (default tag r06; the timing file is profiles/<tag>_c2_phase_timing.txt
if it exists, else r05's)"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 20
lst = open(os.path.join(ROOT, "build", "mpc_api-gfx950.s")).read().split("\n")
syms = [(m.group(1), i) for i, l in enumerate(lst) for m in [re.match(r"^(_Z\w+):", l)] if m]
names = subprocess.run(["c++filt"] + [s for s, _ in syms], capture_output=True, text=True).stdout.split("\n")
code = lambda s, e: [x.strip() for x in lst[s:e] if x.strip() and not x.strip().startswith((";", "."))]
count = lambda body: dict(instr=len(body), valu=sum(x.startswith("v_") for x in body), dpp=sum("row_newbcast" in x or "_dpp" in x.split()[0] for x in body),
                          dpp_fma=sum(x.startswith("v_fmac_f64_dpp") for x in body), lds=sum(x.startswith("ds_") for x in body), s_nop=sum(x.startswith("s_nop") for x in body))


REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+))")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
        out += [(m.group(1), r) for r in range(lo, hi + 1)]
    return out


def dependency_floor(body, passes=8, lat_valu=8.4, lat_lds=64.0):
    """cycles per pass of the loop body's register recurrence (see the module docstring)"""
    ready, ends = {}, []
    for _ in range(passes):
        for x in body:
            op, _, rest = x.partition(" ")
            if not op.startswith(("v_", "ds_")):
                continue
            toks = [t.strip() for t in rest.split(",")]
            if op.startswith("ds_write") or op.startswith("ds_store"):
                continue                                              # stores end a chain (the next stage's operands come from registers)
            dst, src = regs(toks[0]), [r for t in toks[1:] for r in regs(t)]
            if op.startswith(("v_fmac", "v_mac")) or "dpp" in op:
                src += dst                                            # accumulate in place / DPP keeps the old value of disabled lanes
            if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
                dst = []
            t0 = max((ready.get(r, 0.0) for r in src), default=0.0)
            for r in dst:
                ready[r] = t0 + (lat_lds if op.startswith("ds_") else lat_valu)
        ends.append(max(ready.values()))
    return ends[-1] - ends[-2]


def sweeps(kernel, text=False):
    """loop bodies of the factor sweep (two stages per pass: the innermost loop with ~100 DPP multiply-adds) and of the vector sweeps (four stages per pass, 20 of them)"""
    a = next(i for (s, i), n in zip(syms, names) if kernel in n)
    b = next(i for i in range(a, len(lst)) if lst[i].strip().startswith("s_endpgm"))
    labels = {m.group(1): i for i in range(a, b) for m in [re.match(r"^(\.LBB\w+):", lst[i])] if m}
    bodies = [code(labels[m.group(1)], i + 1) for i in range(a, b) for m in [re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", lst[i])]
              if m and m.group(1) in labels and labels[m.group(1)] < i]
    loops = [count(x) for x in bodies]
    i = a
    while i < b:                                    # loops inside asm blocks: local label 1: ... s_cbranch 1b
        if re.match(r"^1:", lst[i].strip()):
            j = next(k for k in range(i + 1, b) if re.search(r"s_cbranch_\w+\s+1b", lst[k]))
            bodies.append(code(i + 1, j + 1)); loops.append(count(bodies[-1])); i = j
        i += 1
    factor = [c for c in loops if 90 <= c["dpp_fma"] <= 140]
    vec = [c for c in loops if c["dpp_fma"] == 20 and c["instr"] < 80]
    assert len(factor) == 1 and len(vec) == 3 and len({json.dumps(v) for v in vec}) == 1, (factor, vec)
    if text:
        return next(x for x, c in zip(bodies, loops) if c is factor[0])
    return factor[0], vec[0]


def per_iter(timing, name):
    return float(re.search(re.escape(name) + r".*per iter\s+(\d+)", timing).group(1))


price_factor = lambda c: (5.0 * c["valu"] + 14.0 * c["lds"]) / 2
price_vec = lambda c: (8.4 * c["dpp_fma"] + 5.0 * (c["valu"] - c["dpp_fma"]) + 14.0 * c["lds"]) / 4
out = dict(N=N, prices_cycles=dict(valu_independent=5.0, valu_dependent=8.4, lds=14.0), kernels=[])
# (C3's kernel is not priced this way: three instances share a wavefront, which iterates until its slowest instance is done, so the phase timing's "per iteration" -- cycles over the
#  MEAN iteration count of the instances -- is not the duration of one pass of the wavefront; its sweeps are the same asm text: sweeps("rti_solve_kernel<3, 21, 3, false>"))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
TIMING = f"{TAG}_c2_phase_timing.txt" if os.path.exists(os.path.join(ROOT, "profiles", f"{TAG}_c2_phase_timing.txt")) else "r05_c2_phase_timing.txt"
for kernel, timing_file, what in (("rti_split_kernel<3, 3, false, false, false>", TIMING, "C2's kernel: one instance per wavefront"),):
    factor, vec = sweeps(kernel)
    timing = open(os.path.join(ROOT, "profiles", timing_file)).read()
    rec = dict(kernel=kernel, what=what, timing=timing_file)
    m = per_iter(timing, "factor sweep") / N
    dep = dependency_floor(sweeps(kernel, text=True)) / 2
    rec["factor_sweep"] = dict(loop_body_two_stages=factor, model_cycles_per_stage=price_factor(factor), measured_cycles_per_stage=m, model_over_measured=round(price_factor(factor) / m, 3),
                               dependency_chain_cycles_per_stage=round(dep, 1), dep_floor=round(dep / m, 3),
                               dep_floor_note="register recurrence of the loop body (every VALU result 8.4 cycles after its last operand, LDS loads 64, unlimited issue) per stage / measured")
    if "split" in kernel:
        m = per_iter(timing, "(split kernel: the affine forward sweep itself") / N
        rec["vector_sweep"] = dict(loop_body_four_stages=vec, model_cycles_per_stage=price_vec(vec), measured_cycles_per_stage=m, model_over_measured=round(price_vec(vec) / m, 3),
                                   note="measured = the affine forward sweep with its prologue (the leading N mod 4 stages, the first operand requests) and the final drain")
    else:
        rec["vector_sweep"] = dict(loop_body_four_stages=vec, model_cycles_per_stage=price_vec(vec), note="same text as in the split kernel; this kernel's phase marks include the staging around the sweep")
    out["kernels"].append(rec)
out["reading"] = ("C2: the sweeps take 1.06x (factor) and 1.13x (vector) what their own instructions cost a lone wavefront: the s_waitcnt cycles the PMC counters show inside them are the "
                  "issue price of the LDS instructions and the latency of the dependent chain, not slack a deeper prefetch could remove")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{TAG}_critical_path_c2.json"), "w"), indent=1)
for r in out["kernels"]:
    print(r["kernel"], "dep_floor", r["factor_sweep"]["dependency_chain_cycles_per_stage"], r["factor_sweep"]["dep_floor"],
          "factor", r["factor_sweep"]["model_cycles_per_stage"], r["factor_sweep"]["measured_cycles_per_stage"], r["factor_sweep"]["model_over_measured"],
          "vector", r["vector_sweep"]["model_cycles_per_stage"], r["vector_sweep"].get("measured_cycles_per_stage"), r["vector_sweep"].get("model_over_measured"))
