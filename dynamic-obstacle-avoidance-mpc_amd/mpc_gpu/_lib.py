"""ctypes binding of libmpcgpu.so (C ABI declared in include/mpc_gpu.h).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) and has no CPU path: every solve entry
point needs a HIP device.  Loading the library and reading its symbol table works without one.
"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.environ.get("MPC_GPU_LIB") or os.path.join(PKG_ROOT, "libmpcgpu.so")    # MPC_GPU_LIB: a diagnostic build (scripts/*_experiment.py)
REPO_ROOT = os.path.dirname(PKG_ROOT)

MPC_OK, MPC_ERR_ARG, MPC_ERR_HIP, MPC_ERR_NODEVICE = 0, -1, -2, -3
STEP_SHIFT, STEP_PLANT, STEP_OBSTACLES, STEP_RESET_ON_FAIL, STEP_ALIAS_BUG, STEP_METRICS, STEP_INTERP_GUESS = 1, 2, 4, 8, 16, 32, 64
COMM_ID_BYTES = 128      # MPC_COMM_ID_BYTES (RCCL unique id)
ABI_VERSION = 7          # MPC_ABI_VERSION of include/mpc_gpu.h this mirror (MpcConfig, SYMBOLS) was written against

_d = C.c_double
_i32 = C.c_int32


class MpcConfig(C.Structure):
    """Mirror of `struct mpc_config` (include/mpc_gpu.h)."""
    _fields_ = [
        ("N", _i32), ("n_obst", _i32), ("Tf", _d),
        ("W", _d * 6), ("We", _d * 4), ("lm", _d),
        ("bx_lo", _d * 4), ("bx_hi", _d * 4), ("bu_lo", _d * 2), ("bu_hi", _d * 2),
        ("r_safe", _d), ("slack_a", _d), ("slack_b", _d),
        ("qp_iter_max", _i32), ("qp_tol", _d),
        ("cost_scale_dt", _i32), ("slack_scale_dt", _i32), ("lm_scaled", _i32),
        ("bx_terminal", _i32), ("soft_h", _i32),
        ("arena", _d * 4), ("bug_compat_predict", _i32),
        ("mu0", _d), ("thr0", _d),
        ("qp_fail_policy", _i32),
        ("polish_ratio", _d), ("polish_tol", _d), ("polish_step_frac", _d), ("polish_res_g", _d),
    ]


# every symbol include/mpc_gpu.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_cfgp = C.POINTER(MpcConfig)
SYMBOLS = {
    "mpc_abi_version": (C.c_int, []),
    "mpc_last_error": (C.c_char_p, []),
    "mpc_device_count": (C.c_int, []),
    "mpc_default_config": (C.c_int, [_cfgp, C.c_int, C.c_int, _d]),
    "mpc_create": (C.c_int, [_cfgp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "mpc_destroy": (C.c_int, [_vp]),
    "mpc_iterate_ptrs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "mpc_set_warmstart": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mpc_get_traj": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mpc_reset_guess": (C.c_int, [_vp, C.c_int, _vp]),
    "mpc_reset_guess_interp": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mpc_shift": (C.c_int, [_vp, C.c_int]),
    "mpc_solve": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mpc_solve_obst": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mpc_set_slack_schedule": (C.c_int, [_vp, C.c_int, _vp]),
    "mpc_set_slack_schedule_dev": (C.c_int, [_vp, _vp]),
    "mpc_plant_step": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "mpc_predict": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mpc_solve_dev": (C.c_int, [_vp, C.c_int] + [_vp] * 10),
    "mpc_closed_loop_step_dev": (C.c_int, [_vp, C.c_int] + [_vp] * 10 + [_d, _d, C.c_int, _vp, _vp, _vp, _vp]),
    "mpc_predict_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "mpc_shift_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "mpc_reset_guess_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "mpc_reset_guess_interp_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "mpc_plant_step_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "mpc_obstacle_step_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _d, _d, _vp]),
    "mpc_linearize_dev": (C.c_int, [_vp, C.c_int] + [_vp] * 12),
    "mpc_debug_adjoint_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int] + [_vp] * 5),
    "mpc_profile_enable": (C.c_int, [_vp, C.c_int]),
    "mpc_profile_read": (C.c_int, [_vp, C.POINTER(_d), C.POINTER(C.c_int)]),
    "mpc_set_accumulators": (C.c_int, [_vp, _vp, _vp]),
    "mpc_debug_trace": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "mpc_set_lanes_per_instance": (C.c_int, [_vp, C.c_int]),
    "mpc_get_lanes_per_instance": (C.c_int, [_vp, C.c_int]),
    "mpc_set_lanes_per_stage": (C.c_int, [_vp, C.c_int]),
    "mpc_get_lanes_per_stage": (C.c_int, [_vp, C.c_int]),
    "mpc_get_kernel_name": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "mpc_set_instance_scheduling": (C.c_int, [_vp, C.c_int]),
    "mpc_get_instance_order": (C.c_int, [_vp, C.c_int, _vp]),
    "mpc_set_waves_per_simd": (C.c_int, [_vp, C.c_int]),
    "mpc_get_waves_per_simd": (C.c_int, [_vp, C.c_int]),
    "mpc_set_matrix_cores": (C.c_int, [_vp, C.c_int]),
    "mpc_set_row_parallel": (C.c_int, [_vp, C.c_int]),
    "mpc_set_block_riccati": (C.c_int, [_vp, C.c_int]),
    "mpc_noise_state_words": (C.c_int, []),
    "mpc_noise_init_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint, _vp, _vp]),
    "mpc_noise_draw_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "mpc_comm_unique_id": (C.c_int, [_vp]),
    "mpc_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "mpc_comm_world": (C.c_int, [_vp]),
    "mpc_comm_library_path": (C.c_int, [C.c_char_p, C.c_int]),
    "mpc_allgather_cost_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "mpc_allgather_cost": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mpc_comm_destroy": (C.c_int, [_vp]),
    "mpc_generate_scenarios_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint, _vp, _vp, _vp]),
    "mpc_generate_scenarios": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint, _vp, _vp]),
}

_LIB = None


class MpcError(RuntimeError):
    pass


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + \
        [os.path.join(REPO_ROOT, "include", "mpc_gpu.h")]


ISA_PATH = os.path.join(REPO_ROOT, "build", "mpc_api-gfx950.s")


def _audit_isa(path):
    """scripts/isa_audit.py over the device listing of the library: data hazards hipcc cannot see across an inline-asm boundary, and per-lane
    instructions the register allocator placed in front of the exec restore of an if / else join (rule P1: the cause of the build variant
    that stored status / iterations / cost to wrong addresses).  Returns the findings as text ('' = clean)."""
    script = os.path.join(REPO_ROOT, "scripts", "isa_audit.py")
    r = subprocess.run([sys.executable, script, path], capture_output=True, text=True)
    return "" if r.returncode == 0 else (r.stdout + r.stderr)


def audit(listing):
    """Audit entry point for a library that did not come out of build() (MPC_GPU_LIB diagnostic builds, which build() uses as they are): pass the
    device listing of THAT compile -- `hipcc ... -save-temps` leaves it as mpc_api-hip-amdgcn-amd-amdhsa-gfx950.s beside the objects; the rules
    need the compiler's labels and inline-asm markers, which a disassembly of the .so no longer has.  Raises MpcError on any finding."""
    findings = _audit_isa(listing)
    if findings:
        raise MpcError("ISA audit failed (scripts/isa_audit.py):\n" + findings[-4000:])
    return True


def build(force=False, verbose=False, run_audit=True):
    """Compile csrc/*.hip for gfx950 into libmpcgpu.so (in-tree).  hipcc cross-compiles without a GPU.
    ONE compile (-save-temps, in a private directory): the device listing that is audited (kept as build/mpc_api-gfx950.s) is the assembler input of
    the code object that ships, not the output of a second compiler run.  A build that fails the audit is deleted and the call raises -- a library
    with a lost-lane copy in it stores to wrong addresses without any test having to notice (MPC_SKIP_ISA_AUDIT=1 skips the audit, for diagnostic
    builds; mpc_gpu._lib.audit(listing) is the entry point for a library supplied through MPC_GPU_LIB).  Concurrent builders (ranks, pytest-xdist) each work under
    their own temporary names and the last os.replace wins."""
    if os.environ.get("MPC_GPU_LIB"):
        return LIB_PATH                  # a diagnostic build supplied by the caller is used as it is
    srcs = sources()
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value"] + os.environ.get("MPC_EXTRA_HIPCC_FLAGS", "").split()
    src = os.path.join(CSRC, "mpc_api.hip")
    do_audit = run_audit and not os.environ.get("MPC_SKIP_ISA_AUDIT")
    os.makedirs(os.path.dirname(ISA_PATH), exist_ok=True)
    work = tempfile.mkdtemp(prefix=f"tmp.{os.getpid()}.", dir=os.path.dirname(ISA_PATH))
    new = f"{LIB_PATH}.{os.getpid()}.new"
    cmd = [hipcc] + flags + ["-fPIC", "-shared"] + (["-save-temps"] if do_audit else []) + ["-o", new, src]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd, cwd=work)
        if do_audit:
            listing = os.path.join(work, "mpc_api-hip-amdgcn-amd-amdhsa-gfx950.s")
            if not os.path.exists(listing):
                raise MpcError(f"hipcc -save-temps left no device listing in {work}")
            findings = _audit_isa(listing)
            if findings:
                raise MpcError("ISA audit of the new build failed (scripts/isa_audit.py; the previous library, if any, is left in place):\n" + findings[-4000:])
            os.replace(listing, ISA_PATH)
        os.replace(new, LIB_PATH)
    finally:
        if os.path.exists(new):
            os.remove(new)
        shutil.rmtree(work, ignore_errors=True)
    return LIB_PATH


def lib():
    """Load libmpcgpu.so and bind every symbol of the header.  Fails loudly when the library is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MpcError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU or PyTorch fallback for the solve path)")
        # One HIP runtime per process: PyTorch's ROCm wheel bundles its own libamdhip64.so (soname libamdhip64.so.7) and
        # libhsa-runtime64; if libmpcgpu pulled in /opt/rocm's copy first, torch's copy would come up second and see no
        # GPU.  Importing torch first makes the dynamic linker bind libmpcgpu's NEEDED libamdhip64.so.7 to the runtime
        # already in the process.  Without torch installed (or MPC_GPU_NO_TORCH=1) the system ROCm runtime is used.
        if not os.environ.get("MPC_GPU_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.mpc_abi_version.restype = C.c_int
        diagnostic_older = bool(os.environ.get("MPC_GPU_LIB")) and L.mpc_abi_version() <= ABI_VERSION
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)      # AttributeError if the .so does not export a declared symbol
            except AttributeError:
                if diagnostic_older:       # (an A/B run against an earlier build, MPC_GPU_LIB: entry points added since are simply absent there)
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        # (a diagnostic build supplied through MPC_GPU_LIB may be OLDER than the mirror -- A/B runs against last round's library: it reads and writes a prefix of
        # the struct, which is safe; a NEWER library behind an older mirror never is)
        older_ab = bool(os.environ.get("MPC_GPU_LIB")) and L.mpc_abi_version() < ABI_VERSION
        if L.mpc_abi_version() != ABI_VERSION and not older_ab:      # a stale .so behind a newer struct mirror would be written past the end of MpcConfig
            raise MpcError(f"{LIB_PATH} has ABI version {L.mpc_abi_version()}, this binding expects {ABI_VERSION}: rebuild (mpc_gpu.build(force=True))")
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        msg = lib().mpc_last_error()
        raise MpcError(f"libmpcgpu error {rc}: {msg.decode() if msg else '?'}")


def default_config(N=20, n_obst=3, Tf=2.0, **overrides):
    cfg = MpcConfig()
    check(lib().mpc_default_config(C.byref(cfg), N, n_obst, Tf))
    for k, v in overrides.items():
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(cfg, k, v)
    return cfg
