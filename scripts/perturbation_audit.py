"""Static robustness check of the workload kernels against the register-allocator defect of DESIGN.md section 8.5 (no GPU needed).

Semantically neutral perturbations of the convergence check -- the edit class that produced the faulting build of round 2 -- are applied to a
copy of the kernel sources; each of the three workload kernels (C2: rti_split_kernel<3,3,false,false>, C3 / C4: rti_solve_kernel<3,21,3,false>,
C5: rti_solve_kernel<10,64,3,false>) is compiled alone to a listing (3-5 s each) and audited with scripts/isa_audit.py.  The table that comes
out (findings, scalar / vector spill counts per build) goes to profiles/r03_perturbation_audit.json.

usage: python scripts/perturbation_audit.py [--git REV]        (--git: audit the sources of a past revision instead of the working tree)
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd", "csrc")
AUDIT = os.path.join(ROOT, "scripts", "isa_audit.py")
OLD = "(it >= kMuCapSettled && mu > p.mu0)"
PERTURBATIONS = {
    "as committed": OLD,
    "round-2 variant: mu > 0.1 mu0": "(it >= kMuCapSettled && mu > 0.1 * p.mu0)",
    "mu > 0.5 mu0": "(it >= kMuCapSettled && mu > 0.5 * p.mu0)",
    "mu > 2 mu0": "(it >= kMuCapSettled && mu > 2.0 * p.mu0)",
    "10 mu > mu0": "(it >= kMuCapSettled && 10.0 * mu > p.mu0)",
    "mu - mu0 > 0": "(it >= kMuCapSettled && mu - p.mu0 > 0.0)",
    "settled from iteration 15": "(it >= 15 && mu > p.mu0)",
    "mu > 0.1 mu0 or cmax > mu0": "(it >= kMuCapSettled && (mu > 0.1 * p.mu0 || cmax > p.mu0))",
    "mu^2 > mu0^2": "(it >= kMuCapSettled && mu * mu > p.mu0 * p.mu0)",
}
KERNELS = {
    "C2 rti_split_kernel<3,3,false,false>": ('#include "rti_kernel.hpp"\n#include "rti_split_kernel.hpp"\n'
                                             "template __global__ void mpc::rti_split_kernel<3, 3, false, false>(const mpc::KParams);\n"),
    "C3 rti_solve_kernel<3,21,3,false>": ('#include "rti_kernel.hpp"\ntemplate __global__ void mpc::rti_solve_kernel<3, 21, 3, false>(const mpc::KParams);\n'),
    "C5 rti_solve_kernel<10,64,3,false>": ('#include "rti_kernel.hpp"\ntemplate __global__ void mpc::rti_solve_kernel<10, 64, 3, false>(const mpc::KParams);\n'),
}


def main():
    rev = sys.argv[sys.argv.index("--git") + 1] if "--git" in sys.argv else None
    work = tempfile.mkdtemp(prefix="perturb_")
    srcs = {}
    for f in ("rti_kernel.hpp", "rti_split_kernel.hpp"):
        if rev:
            srcs[f] = subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:dynamic-obstacle-avoidance-mpc_amd/csrc/{f}"], text=True)
        else:
            srcs[f] = open(os.path.join(CSRC, f)).read()
        assert OLD in srcs[f], f
    out = {"sources": rev or "working tree", "builds": []}
    for pname, new in PERTURBATIONS.items():
        for f, text in srcs.items():
            open(os.path.join(work, f), "w").write(text.replace(OLD, new))
        for kname, tu in KERNELS.items():
            open(os.path.join(work, "one.hip"), "w").write(tu)
            s_path = os.path.join(work, "one.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", work, "--cuda-device-only", "-S", "-Wno-unused-value",
                                   "-o", s_path, os.path.join(work, "one.hip")], stderr=subprocess.DEVNULL)
            listing = open(s_path).read()
            r = subprocess.run([sys.executable, AUDIT, s_path], capture_output=True, text=True)
            rec = {"perturbation": pname, "kernel": kname, "audit_findings": int(r.stdout.strip().split("\n")[-1].split(",")[-1].split()[0]),
                   "sgpr_spills": int(re.search(r"\.sgpr_spill_count:\s*(\d+)", listing).group(1)),
                   "vgpr_spills_to_agpr": int(re.search(r"\.vgpr_spill_count:\s*(\d+)", listing).group(1)),
                   "agprs": int(re.search(r"\.agpr_count:\s*(\d+)", listing).group(1)),
                   "scratch_bytes": int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", listing).group(1))}
            if rec["audit_findings"]:
                rec["first_finding"] = [l for l in r.stdout.split("\n") if l.strip()][:3]
            out["builds"].append(rec)
            print(json.dumps(rec), flush=True)
    shutil.rmtree(work)
    out["failing_builds"] = sum(1 for b in out["builds"] if b["audit_findings"])
    tag = "" if not rev else "_" + rev[:7]
    path = os.path.join(ROOT, "profiles", f"r03_perturbation_audit{tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print("->", path, "failing builds:", out["failing_builds"])


if __name__ == "__main__":
    main()
