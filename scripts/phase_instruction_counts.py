"""Static instruction counts per phase of a solve kernel (no GPU needed): the kernel is compiled alone with -DMPC_PHASE_TIMING, whose clock reads (s_memtime) mark the
phase boundaries in the listing; between two marks the instructions are counted by class.  Loops inside a phase (the stage recursions over a run-time horizon) are counted once:
multiply the sweep phases by the passes they make.  The timing build spills more than the shipped one (304 against 75 scalars at C3), so the v_readlane / v_writelane / v_accvgpr
columns are upper bounds.  usage: python scripts/phase_instruction_counts.py [NOBST G FACT]   -> profiles/r03_<kernel>_phase_instruction_counts.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd", "csrc")
NAMES = ["setup (loads, look-ahead, linearise, row state)", "mu / convergence check", "predictor assemble", "stage operands -> LDS", "factor sweep (loop body, per pass)",
         "rollout, affine (loop body + staging)", "affine step + sigma", "corrector rhs", "corrector sweep (loop body + staging)", "rollout (loop body + staging)",
         "combined step + update", "loop back edge + tail (full step, plant, stores)", "(trace write)"]


def main():
    nobst, g, fact = (sys.argv[1:4] + ["3", "21", "3"][len(sys.argv) - 1:])[:3]
    work = tempfile.mkdtemp(prefix="phasecount_")
    src = os.path.join(work, "one.hip"); lst = os.path.join(work, "one.s")
    open(src, "w").write(f'#include "rti_kernel.hpp"\ntemplate __global__ void mpc::rti_solve_kernel<{nobst}, {g}, {fact}, false>(const mpc::KParams);\n')
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DMPC_PHASE_TIMING", "-I", CSRC, "--cuda-device-only", "-S", "-Wno-unused-value",
                           "-o", lst, src], stderr=subprocess.DEVNULL)
    segs, cur = [], {}
    for l in open(lst):
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        if op.startswith("s_memtime"):
            segs.append(cur); cur = {}; continue
        def bump(k): cur[k] = cur.get(k, 0) + 1
        if op.startswith("v_"):
            bump("VALU")
            if op.startswith("v_accvgpr"): bump("accvgpr")
            elif op.startswith(("v_readlane", "v_writelane")): bump("lane moves")
            elif op in ("v_mov_b32_e32", "v_mov_b64_e32") or op.startswith("v_cndmask"): bump("mov/select")
            elif "dpp" in op or "row_newbcast" in t: bump("DPP")
            elif re.match(r"v_(rcp|rsq|sqrt|div_)", op): bump("rcp/sqrt/div")
        elif op.startswith("ds_"): bump("LDS")
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): bump("VMEM")
        elif op.startswith("s_"): bump("SALU")
    segs.append(cur)
    cols = ["VALU", "DPP", "accvgpr", "lane moves", "mov/select", "rcp/sqrt/div", "LDS", "SALU", "VMEM"]
    out = [f"rti_solve_kernel<{nobst}, {g}, {fact}, false>, -DMPC_PHASE_TIMING build, static counts between clock reads", f"{'phase':52s}" + "".join(f"{c:>13s}" for c in cols)]
    for k, c in enumerate(segs):
        out.append(f"{(NAMES[k] if k < len(NAMES) else str(k)):52s}" + "".join(f"{c.get(x, 0):13d}" for x in cols))
    text = "\n".join(out)
    print(text)
    open(os.path.join(ROOT, "profiles", f"{os.environ.get('MPC_PROFILE_TAG', 'r05')}_solve_{nobst}_{g}_{fact}_phase_instruction_counts.txt"), "w").write(text + "\n")


if __name__ == "__main__":
    main()
