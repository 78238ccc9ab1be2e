"""In-tree build of the HIP library (hipcc, gfx950 only).  No JIT cache: the
resulting audiosignalprocess_amd/lib/libasp_amd.so travels with the tree."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
# ASP_AMD_LIB: load another build of the library (same-box A / B runs against an earlier build)
LIB = os.environ.get("ASP_AMD_LIB") or os.path.join(LIBDIR, "libasp_amd.so")
SOURCES = ["ns_kernels.hip", "ns_kernels1.hip", "ns_kernels2.hip", "ns_kernels_hb.hip", "ns_api.hip", "bt_kernels.hip", "bt_kernels8.hip", "bt_api.hip",
           "aec_kernels.hip", "aec_delay_kernels.hip", "aec_api.hip", "qmf_kernels.hip", "qmf_api.hip", "sinc_kernels.hip", "sinc_api.hip"]
C_SOURCES = ["wav_io.c"]  # host-only C (kept C, as in the reference)
# -ffp-contract=off: parity with the reference depends on unfused mul/add.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


# per-file extras; ASP_HIPCC_EXTRA="file.hip:-flag -flag;other.hip:-flag" adds more (experiments)
# kernarg preload (gfx950): the dispatch arrives with its arguments in SGPRs instead of fetching them
# with dependent scalar loads before the first vector load can issue (-0.5 us per NS step, measured);
# the object carries a prologue for firmware without the feature
_PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=8"]
EXTRA = {"ns_kernels.hip": list(_PRELOAD), "ns_kernels1.hip": list(_PRELOAD), "ns_kernels2.hip": list(_PRELOAD),
         # the echo canceller's block is long straight-line code at 4 waves per SIMD: the compiler's ILP-first
         # scheduling measured 92.7-93.8 us per step against 95.5 us in one session (max-ilp: 97-99 us)
         "aec_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}
# second builds of a source under another object name: (source, object, extra flags)
VARIANTS = []
for _item in filter(None, os.environ.get("ASP_HIPCC_EXTRA", "").split(";")):
    _f, _, _fl = _item.partition(":")
    EXTRA.setdefault(_f.strip(), []).extend(_fl.split())


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip into lib/libasp_amd.so; returns its path."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, "ns_layout.h"), os.path.join(CSRC, "ns_device.h"), os.path.join(CSRC, "ns_pair_fft.h"), os.path.join(CSRC, "bt_layout.h"), os.path.join(CSRC, "pk_f32.h"), os.path.join(CSRC, "bt_sure.h"), os.path.join(CSRC, "aec_layout.h"), os.path.join(CSRC, "aec_binspec.h"), os.path.join(CSRC, "aec_estimator.h"), os.path.join(CSRC, "sinc_layout.h"), os.path.join(CSRC, "device_scope.h"),
                   os.path.join(ROOT, "include", "asp_ns.h"), os.path.join(ROOT, "include", "asp_bt.h"), os.path.join(ROOT, "include", "asp_aec.h"), os.path.join(ROOT, "include", "asp_split.h"), os.path.join(ROOT, "include", "asp_resample.h")]
    objs = []
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc()] + FLAGS + EXTRA.get(os.path.basename(s), []) + inc + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(o)
    for src_name, obj_name, flags in VARIANTS:
        s = os.path.join(CSRC, src_name)
        o = os.path.join(LIBDIR, obj_name)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc()] + FLAGS + EXTRA.get(src_name, []) + flags + inc + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(o)
    for s in C_SOURCES:
        src = os.path.join(CSRC, s)
        o = os.path.join(LIBDIR, s + ".o")
        if force or _stale(o, [src, os.path.join(ROOT, "include", "wav_io.h")]):
            cmd = ["gcc", "-O2", "-std=gnu99", "-fPIC", "-Wall"] + inc + ["-c", src, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_drivers(verbose=False):
    """The C WAV drivers (drivers/*.c) linked against the in-tree library."""
    out_dir = os.path.join(ROOT, "drivers", "bin")
    os.makedirs(out_dir, exist_ok=True)
    built = []
    for name in ["test_ns_module", "ns_batch_wav", "test_aec_module", "bt_main"]:
        src = os.path.join(ROOT, "drivers", name + ".c")
        exe = os.path.join(out_dir, name)
        if _stale(exe, [src, LIB]):
            cmd = ["gcc", "-O2", "-std=gnu99", "-Wall", "-I" + os.path.join(ROOT, "include"), src,
                   "-L" + LIBDIR, "-lasp_amd", "-Wl,-rpath,$ORIGIN/../../audiosignalprocess_amd/lib",
                   "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        built.append(exe)
    # C++ client of the header-only APM_NS class (include/apm_ns.h)
    src = os.path.join(ROOT, "drivers", "apm_ns_raw.cpp")
    exe = os.path.join(out_dir, "apm_ns_raw")
    if _stale(exe, [src, LIB, os.path.join(ROOT, "include", "apm_ns.h")]):
        cmd = ["g++", "-O2", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "include"), src,
               "-L" + LIBDIR, "-lasp_amd", "-Wl,-rpath,$ORIGIN/../../audiosignalprocess_amd/lib",
               "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    built.append(exe)
    return built


if __name__ == "__main__":
    print(build_library(verbose=True))
    print(build_drivers(verbose=True))
