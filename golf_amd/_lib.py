"""ctypes binding of libgolf_hip.so (include/golf_amd.h) — the ONLY compute backend.

There is deliberately no CPU or pure-PyTorch fallback: if the HIP library is missing or the
tensors are not on a ROCm device the ops raise.  (The CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgolf_hip.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
SOURCES = ("abi.hip", "lpc_ss.hip", "lpc_ff.hip", "glottal_osc.hip", "noise_fir.hip", "ctrl.hip", "noise_band.hip", "peer.hip")

_c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_sz = ctypes.c_size_t
_vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/golf_amd.h exactly (checked by tests/test_abi.py)
SIGNATURES = {
    "golf_abi_version": (_int, []),
    "golf_last_error": (ctypes.c_char_p, []),
    "golf_target_arch": (ctypes.c_char_p, []),
    "golf_ltv_allpole_workspace_bytes": (_sz, [_int] * 5),
    "golf_ltv_allpole_workspace_bytes_ex": (_sz, [_int] * 6),
    "golf_ltv_allpole_transitions_f32": (_int, [_c_f32p] + [_int] * 5 + [_vp, _sz, _int, _vp]),
    "golf_ltv_allpole_fwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i64] + [_int] * 5
                                 + [_vp, _sz, _int, _vp, _vp]),
    "golf_ltv_allpole_status_u32": (_int, [_vp, _sz] + [_int] * 6 + [_vp, _vp]),
    "golf_ltv_allpole_bwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i64,
                                        _c_f32p, _c_f32p] + [_int] * 5 + [_vp, _sz, _int, _vp]),
    "golf_ltv_inverse_f32": (_int, [_c_f32p, _i64, _c_f32p, _c_f32p, _i64] + [_int] * 5 + [_vp]),
    "golf_ltv_inverse_bwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _c_f32p, _i64, _c_f32p] + [_int] * 5
                                 + [_vp]),
    "golf_lti_frames_workspace_bytes": (_sz, [_int] * 6),
    "golf_lti_frames_ola_fwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64] + [_int] * 7
                                    + [_vp, _sz, _vp]),
    "golf_lti_frames_bwd_workspace_bytes": (_sz, [_int] * 6),
    "golf_lti_frames_ola_bwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _int,
                                           _c_f32p, _c_f32p] + [_int] * 7 + [_vp, _vp, _sz, _vp]),
    "golf_biquad_frames_ola_fwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64] + [_int] * 9
                                       + [_vp, _sz, _vp]),
    "golf_biquad_frames_bwd_workspace_bytes": (_sz, [_int] * 7),
    "golf_biquad_frames_ola_bwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64,
                                              _c_f32p, _c_f32p, _c_f32p] + [_int] * 9 + [_vp, _sz, _vp]),
    "golf_rc2lpc_fwd_f32": (_int, [_c_f32p, _c_f32p, _i64, _int, ctypes.c_float, _int, _vp]),
    "golf_rc2lpc_bwd_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _int, ctypes.c_float, _int, _vp]),
    "golf_sos2lpc_fwd_f32": (_int, [_c_f32p, _c_f32p, _i64, _int, ctypes.c_float, _int, _vp]),
    "golf_sos2lpc_bwd_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _i64, _int, ctypes.c_float, _int, _vp]),
    "golf_glottal_osc_workspace_bytes": (_sz, [_int] * 7),
    "golf_glottal_osc_fwd_f32": (_int, [_c_f32p, _i64, _int, _int, _c_f32p, _int, _int, _c_f32p, _int, _int, _int, _int,
                                        _c_f32p, _int, _c_f32p, _c_f32p, _i64, _int, _int, _vp, _sz, _vp, _c_f32p, _i64,
                                        _int, _vp]),
    "golf_source_transitions_f32": (_int, [_c_f32p, _i64, _int, _int, _c_f32p, _int, _int, _c_f32p, _int, _int, _int, _int,
                                           _c_f32p, _int, _c_f32p, _i64, _int, _int, _vp, _sz, _c_f32p, _i64, _int, _vp,
                                           _c_f32p, _int, _int, _int, _int, _vp, _sz, _int, _vp]),
    "golf_glottal_osc_tap_fragments_bytes": (_sz, [_int] * 2),
    "golf_glottal_osc_tap_fragments_f32": (_int, [_c_f32p, _int, _int, _vp, _sz, _vp]),
    "golf_glottal_osc_bwd_wsel_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _int, _int, _c_f32p, _int, _int, _c_f32p,
                                             _int, _int, _int, _int, _c_f32p, _int, _c_f32p, _int, _int, _vp, _sz,
                                             _vp, _vp]),
    "golf_wavetable_lookup_fwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _int, _int, _int, _c_f32p, _i64, _int, _int, _vp]),
    "golf_wavetable_lookup_bwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _int, _int, _int, _c_f32p, _i64,
                                             _c_f32p, _int, _int, _vp]),
    "golf_phase_accumulate_workspace_bytes": (_sz, [_int] * 2),
    "golf_phase_accumulate_f32": (_int, [_c_f32p, _i64, _int, _int, _int, _c_f32p, _i64, _c_f32p, _i64, _int, _int, _vp,
                                         _sz, _vp]),
    "golf_decimate_fir_f32": (_int, [_c_f32p, _i64, _int, _c_f32p, _int, _int, _c_f32p, _i64, _int, _int, _vp]),
    "golf_decimate_fir_adj_f32": (_int, [_c_f32p, _i64, _int, _c_f32p, _int, _int, _c_f32p, _int, _int, _vp]),
    "golf_noise_band_workspace_bytes": (_sz, [_int] * 3),
    "golf_noise_band_fwd_f32": (_int, [_c_f32p, _int, _vp, _c_f32p, _int, _int, _c_f32p, _i64, _int, _int, _int, _vp]),
    "golf_noise_band_bwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _int, _vp, _c_f32p, _int, _int, _c_f32p, _int, _int, _int,
                                       _vp, _sz, _vp]),
    "golf_harmonic_osc_workspace_bytes": (_sz, [_int] * 5),
    "golf_harmonic_osc_fwd_f32": (_int, [_c_f32p, _i64, _int, _int, _c_f32p, _int, _int, _c_f32p, _int, _int, _c_f32p,
                                         _int, _c_f32p, _i64, _int, _int, _vp, _sz, _vp, _c_f32p, _int, _int, _c_f32p]),
    "golf_harmonic_osc_dphase_f32": (_int, [_c_f32p, _i64, _int, _int, _c_f32p, _int, _int, _c_f32p, _int, _int, _c_f32p,
                                         _int, _c_f32p, _i64, _int, _int, _vp, _sz, _vp, _c_f32p, _int, _int, _c_f32p]),
    "golf_harmonic_osc_bwd_amp_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _int, _int, _int, _int, _c_f32p, _int, _int,
                                             _c_f32p, _int, _c_f32p, _int, _int, _vp, _sz, _vp, _c_f32p, _int, _int,
                                             _c_f32p]),
    "golf_zero_phase_fir_row_stride": (_int, [_int]),
    "golf_zero_phase_fir_basis_bytes": (_sz, [_int]),
    "golf_zero_phase_fir_basis_f32": (_int, [_int, _vp, _sz, _vp]),
    "golf_zero_phase_fir_kernels_f32": (_int, [_c_f32p, _c_f32p, _vp, _c_f32p, _int, _int, _vp]),
    "golf_zero_phase_fir_kernels_bwd_f32": (_int, [_c_f32p, _c_f32p, _c_f32p, _vp, _c_f32p, _int, _int, _vp]),
    "golf_ltv_fir_frames_length": (_int, [_int] * 4),
    "golf_ltv_fir_frames_fwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _int, _c_f32p, _i64] + [_int] * 6 + [_vp]),
    "golf_ltv_fir_frames_bwd_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _int, _c_f32p, _i64, _c_f32p]
                                    + [_int] * 6 + [_vp]),
    "golf_lti_fir_f32": (_int, [_c_f32p, _i64, _c_f32p, _int, _int, _c_f32p, _i64, _int, _int, _vp]),
    "golf_lti_fir_taps_grad_workspace_bytes": (_sz, [_int] * 3),
    "golf_lti_fir_taps_grad_f32": (_int, [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _int, _int, _int, _int, _vp, _sz,
                                          _vp]),
    "golf_peer_alloc": (_int, [_sz, _vp]),
    "golf_peer_free": (_int, [_vp]),
    "golf_peer_export": (_int, [_vp, _vp]),
    "golf_peer_open": (_int, [_vp, _vp]),
    "golf_peer_close": (_int, [_vp]),
    "golf_peer_store_f32": (_int, [_c_f32p, _i64, _int, _int, _vp, _i64, _int, _vp]),
    "golf_peer_signal_u32": (_int, [_vp, _int, ctypes.c_uint32, _vp]),
    "golf_peer_wait_u32": (_int, [_vp, _int, _int, ctypes.c_uint32, _i64, _vp, _vp]),
}

ABI_VERSION = 6
_lock = threading.Lock()
_lib = None


def _hipcc_version(hipcc: str) -> str:
    try:
        return subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout.decode()
    except Exception as e:   # noqa: BLE001 -- an unknown compiler is its own stamp
        return repr(e)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into golf_amd/lib/libgolf_hip.so (hipcc cross-compiles without a GPU).
    Incremental: a translation unit is recompiled when its source or a header is newer than its object OR when the object was
    built by another command line or another hipcc (a stamp file next to every object holds both -- ADVICE r5: objects from
    before a flag change used to be linked silently).  A build with GOLF_HIPCC_FLAGS (kernel-tuning A/B builds) goes to its
    own objects and its own library, libgolf_hip.flags.so -- load it with GOLF_HIP_LIBRARY -- and never replaces the product
    library."""
    import hashlib

    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "device_common.h"), os.path.join(CSRC, "lpc_p1f.h"),
                   os.path.join(INCLUDE, "golf_amd.h")]
    extra = os.environ.get("GOLF_HIPCC_FLAGS", "").split()
    lib_path = os.path.join(LIB_DIR, "libgolf_hip.flags.so") if extra else LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    version = None
    objs, procs, stamps = [], [], []
    hdr_time = max(os.path.getmtime(d) for d in deps[len(srcs):])
    for s in srcs:
        o = os.path.join(LIB_DIR, os.path.basename(s) + (".flags.o" if extra else ".o"))
        objs.append(o)
        # -falign-loops=64: the transition kernel's unrolled loop is ~1 300 8-byte packed-FMA encodings; when an unrelated edit
        # moved its start to 4 mod 8 bytes the kernel went from 38.8 to 43.0 us with an otherwise identical instruction stream
        # (round 4, tools/ab2.sh ab_place2) -- aligned loop heads take the placement lottery out of every kernel
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-falign-loops=64", "-I" + INCLUDE, "-I" + CSRC,
               "-c", s, "-o", o] + extra   # kernel-tuning A/B builds (-DP1F_CHAINS=2 ...)
        fresh = os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), hdr_time)
        stamp_file = o + ".cmd"
        if fresh and not force:
            if version is None:
                version = _hipcc_version(hipcc)
            stamp = hashlib.sha256((" ".join(cmd) + "\n" + version).encode()).hexdigest()
            try:
                same = open(stamp_file).read().strip() == stamp
            except OSError:
                same = False   # an object without a stamp (built before round 6, or by hand): not trusted
            if same:
                continue       # only the translation units that changed (lpc_ss.hip alone takes five minutes)
        if version is None:
            version = _hipcc_version(hipcc)
        stamp = hashlib.sha256((" ".join(cmd) + "\n" + version).encode()).hexdigest()
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        stamps.append((stamp_file, stamp))
    if not procs and not force and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(o) for o in objs):
        return lib_path
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
    for stamp_file, stamp in stamps:
        with open(stamp_file, "w") as f:
            f.write(stamp + "\n")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    for f in os.listdir(LIB_DIR):   # clang-offload-bundler leaves its temporaries next to the output
        if f.startswith(os.path.basename(lib_path) + ".") and ("hipv4-" in f or "host-" in f):
            os.remove(os.path.join(LIB_DIR, f))
    return lib_path


def load():
    """dlopen the in-tree library and attach the prototypes.  Raises if it is not there."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"golf_amd: {LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback by design)")
        lib = ctypes.CDLL(os.environ.get("GOLF_HIP_LIBRARY", LIB_PATH))   # override: an A/B build of the same ABI
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        ver = lib.golf_abi_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"golf_amd: ABI version {ver} != {ABI_VERSION}")
        _lib = lib
        return lib


class GolfError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = load().golf_last_error().decode(errors="replace")
        kind = "bad argument" if rc < 0 else "hipError"
        raise GolfError(f"{what}: {kind} {rc}: {msg}")


def require_device(*tensors: torch.Tensor):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GolfError("golf_amd kernels need ROCm device tensors (got a %s tensor); there is no CPU path"
                            % t.device.type)
        if t.dtype != torch.float32:
            raise GolfError(f"golf_amd kernels are fp32 (got {t.dtype})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()
