"""What the sweeps of the C2 kernel cost by their own instructions, against what they were measured to take (no GPU needed).
From the SHIPPED listing (build/mpc_api-gfx950.s) of rti_split_kernel<3, 3>: the factor sweep's loop body (compiled C++ around three asm blocks, two stages per pass) and the
hand-scheduled vector sweep's loop body (one asm block, four stages per pass) are counted by class and priced at what ONE wavefront per SIMD pays (micro-benchmarks
scripts/bin_src/dpp64_test.hip, lds_issue_test.hip: independent FP64 instruction 5.0 cycles, dependent one 8.4, LDS instruction 14 of issue):
    factor sweep   per stage = (5.0 x VALU + 14 x LDS) / 2                    (its DPP multiply-adds are interleaved so that none waits for its predecessor)
    vector sweep   per stage = (8.4 x chained multiply-adds + 5.0 x other VALU + 14 x LDS) / 4      (the 5 multiply-adds of a stage are one dependent chain)
and divided by the measured cycles per stage of the same phases (profiles/r05_c2_phase_timing.txt, -DMPC_PHASE_TIMING build, s_memtime around the phase).
usage (after build()): python scripts/critical_path_model.py   -> profiles/r05_critical_path_c2.json"""
import json, os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 20
lst = open(os.path.join(ROOT, "build", "mpc_api-gfx950.s")).read().split("\n")
syms = [(m.group(1), i) for i, l in enumerate(lst) for m in [re.match(r"^(_Z\w+):", l)] if m]
names = subprocess.run(["c++filt"] + [s for s, _ in syms], capture_output=True, text=True).stdout.split("\n")
code = lambda s, e: [x.strip() for x in lst[s:e] if x.strip() and not x.strip().startswith((";", "."))]
count = lambda body: dict(instr=len(body), valu=sum(x.startswith("v_") for x in body), dpp=sum("row_newbcast" in x or "_dpp" in x.split()[0] for x in body),
                          dpp_fma=sum(x.startswith("v_fmac_f64_dpp") for x in body), lds=sum(x.startswith("ds_") for x in body), s_nop=sum(x.startswith("s_nop") for x in body))


def sweeps(kernel):
    """loop bodies of the factor sweep (two stages per pass: the innermost loop with ~100 DPP multiply-adds) and of the vector sweeps (four stages per pass, 20 of them)"""
    a = next(i for (s, i), n in zip(syms, names) if kernel in n)
    b = next(i for i in range(a, len(lst)) if lst[i].strip().startswith("s_endpgm"))
    labels = {m.group(1): i for i in range(a, b) for m in [re.match(r"^(\.LBB\w+):", lst[i])] if m}
    loops = [count(code(labels[m.group(1)], i + 1)) for i in range(a, b) for m in [re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", lst[i])]
             if m and m.group(1) in labels and labels[m.group(1)] < i]
    i = a
    while i < b:                                    # loops inside asm blocks: local label 1: ... s_cbranch 1b
        if re.match(r"^1:", lst[i].strip()):
            j = next(k for k in range(i + 1, b) if re.search(r"s_cbranch_\w+\s+1b", lst[k]))
            loops.append(count(code(i + 1, j + 1))); i = j
        i += 1
    factor = [c for c in loops if 90 <= c["dpp_fma"] <= 140]
    vec = [c for c in loops if c["dpp_fma"] == 20 and c["instr"] < 80]
    assert len(factor) == 1 and len(vec) == 3 and len({json.dumps(v) for v in vec}) == 1, (factor, vec)
    return factor[0], vec[0]


def per_iter(timing, name):
    return float(re.search(re.escape(name) + r".*per iter\s+(\d+)", timing).group(1))


price_factor = lambda c: (5.0 * c["valu"] + 14.0 * c["lds"]) / 2
price_vec = lambda c: (8.4 * c["dpp_fma"] + 5.0 * (c["valu"] - c["dpp_fma"]) + 14.0 * c["lds"]) / 4
out = dict(N=N, prices_cycles=dict(valu_independent=5.0, valu_dependent=8.4, lds=14.0), kernels=[])
# (C3's kernel is not priced this way: three instances share a wavefront, which iterates until its slowest instance is done, so the phase timing's "per iteration" -- cycles over the
#  MEAN iteration count of the instances -- is not the duration of one pass of the wavefront; its sweeps are the same asm text: sweeps("rti_solve_kernel<3, 21, 3, false>"))
for kernel, timing_file, what in (("rti_split_kernel<3, 3, false, false, false>", "r05_c2_phase_timing.txt", "C2's kernel: one instance per wavefront"),):
    factor, vec = sweeps(kernel)
    timing = open(os.path.join(ROOT, "profiles", timing_file)).read()
    rec = dict(kernel=kernel, what=what, timing=timing_file)
    m = per_iter(timing, "factor sweep") / N
    rec["factor_sweep"] = dict(loop_body_two_stages=factor, model_cycles_per_stage=price_factor(factor), measured_cycles_per_stage=m, model_over_measured=round(price_factor(factor) / m, 3))
    if "split" in kernel:
        m = per_iter(timing, "(split kernel: the affine forward sweep itself") / N
        rec["vector_sweep"] = dict(loop_body_four_stages=vec, model_cycles_per_stage=price_vec(vec), measured_cycles_per_stage=m, model_over_measured=round(price_vec(vec) / m, 3),
                                   note="measured = the affine forward sweep with its prologue (the leading N mod 4 stages, the first operand requests) and the final drain")
    else:
        rec["vector_sweep"] = dict(loop_body_four_stages=vec, model_cycles_per_stage=price_vec(vec), note="same text as in the split kernel; this kernel's phase marks include the staging around the sweep")
    out["kernels"].append(rec)
out["reading"] = ("C2: the sweeps take 1.06x (factor) and 1.13x (vector) what their own instructions cost a lone wavefront: the s_waitcnt cycles the PMC counters show inside them are the "
                  "issue price of the LDS instructions and the latency of the dependent chain, not slack a deeper prefetch could remove")
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_critical_path_c2.json"), "w"), indent=1)
for r in out["kernels"]:
    print(r["kernel"], "factor", r["factor_sweep"]["model_cycles_per_stage"], r["factor_sweep"]["measured_cycles_per_stage"], r["factor_sweep"]["model_over_measured"],
          "vector", r["vector_sweep"]["model_cycles_per_stage"], r["vector_sweep"].get("measured_cycles_per_stage"), r["vector_sweep"].get("model_over_measured"))
