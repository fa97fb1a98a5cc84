"""Build libtrl_hip.so (gfx950) in-tree with hipcc.  No CPU fallback exists."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libtrl_hip.so")
NOISE_LIB = os.path.join(LIB_DIR, "libtrl_noise.so")     # host helper of the reference noise stream; links libtorch (optional)
NOISE_SRC = "trl_noise_ext.cpp"
SOURCES = ["trl_host.cpp", "trl_mtjump.cpp", "k_gae.hip", "k_gather.hip", "k_ppo.hip", "k_ppo_generic.hip", "k_vmpo.hip", "k_trpo.hip", "k_rollout.hip", "k_gemm.hip", "k_mlp3.hip", "k_conv1.hip", "k_conv_dx.hip", "k_sac.hip", "k_conv.hip", "k_dqn.hip", "k_norm.hip", "k_frames.hip", "k_comm.hip", "k_peaks.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != NOISE_SRC]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "trl_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def noise_helper_tag():
    """What a helper built now would be built for: this interpreter's torch version and C++ ABI flag."""
    import torch
    return "%s abi%d" % (torch.__version__, int(torch.compiled_with_cxx11_abi()))


def noise_helper_built_for(path=None):
    """The tag compiled into an existing libtrl_noise.so (read from the file, without loading it: a helper made for another
    torch must not get the chance to pull its libtorch in), or None."""
    import re
    try:
        with open(path or NOISE_LIB, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(rb"(\d+\.\d+[^\x00 ]* abi[01])\x00", blob)
    return m.group(1).decode() if m else None


def build_noise_helper(force=False, verbose=True):
    """libtrl_noise.so: csrc/trl_noise_ext.cpp against this interpreter's libtorch (g++, host only).  Optional -- returns
    None (and says why) when the torch headers or g++ are not there; collector/noise.py then draws from Python threads."""
    src = os.path.join(CSRC, NOISE_SRC)
    if not force and os.path.exists(NOISE_LIB) and os.path.getmtime(NOISE_LIB) >= os.path.getmtime(src) \
            and noise_helper_built_for() == noise_helper_tag():
        return NOISE_LIB
    try:
        import torch
        from torch.utils import cpp_extension as ce
        gxx = shutil.which("g++")
        if gxx is None:
            raise RuntimeError("g++ not found")
        libp = ce.library_paths()
        cmd = [gxx, "-O2", "-std=c++17", "-shared", "-fPIC"] + ["-I" + i for i in ce.include_paths()] + \
              ["-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi()),
               '-DTRL_NOISE_BUILT_FOR="%s"' % noise_helper_tag(), src, "-o", NOISE_LIB] + \
              ["-L" + p for p in libp] + ["-ltorch_cpu", "-lc10"] + ["-Wl,-rpath," + p for p in libp]
        os.makedirs(LIB_DIR, exist_ok=True)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if res.returncode != 0:
            raise RuntimeError(res.stdout.decode()[-2000:])
        if verbose:
            print("built", NOISE_LIB)
        return NOISE_LIB
    except Exception as exc:                                   # noqa: BLE001 -- the helper is an accelerator, not a requirement
        if verbose:
            print("libtrl_noise.so not built (%s): the reference-noise chunks are drawn from Python threads" % (exc,))
        return None


def build(force=False, verbose=True, extra_flags=(), lib=None):
    """`extra_flags` / `lib`: experimental builds for the tools/ scripts (e.g. -DTRL_EXP_CLK)."""
    LIB = lib or globals()["LIB"]
    FLAGS = globals()["FLAGS"] + list(extra_flags)
    if not lib:
        build_noise_helper(force=force, verbose=verbose)
    if not force and not lib and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libtrl_hip.so cannot be built")
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + (".exp.o" if lib else ".o"))
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + \
              ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    # python build.py [--force] [--exp NAME -DFLAG ...]  ->  lib/libtrl_hip_NAME.so
    if "--exp" in sys.argv:
        k = sys.argv.index("--exp")
        build(force=True, extra_flags=sys.argv[k + 2:], lib=os.path.join(LIB_DIR, "libtrl_hip_%s.so" % sys.argv[k + 1]))
    else:
        build(force="--force" in sys.argv)
