"""Build recipe for csrc/libswimmer_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("SWIMMER_HIP_LIB") or os.path.join(CSRC, "libswimmer_hip.so")  # override: experiments
SOURCES = ["swimmer_kernels.hip", "host_rng.cpp", "direct_comm.cpp"]
HOST_ONLY = {"host_rng.cpp"}   # plain C++, no device pass: it picks its vector width from the CPU's features at
                               # run time (x86 builtins the device pass of a HIP compile refuses)
HEADERS = ["rlglue_env.cpp", os.path.join("..", "..", "include", "rlglue_swimmer.h"),
           "swimmer_device.h", "swimmer_quad3.h", "swimmer_oct3.h", "swimmer_row.h", "swimmer_row_fused.h", "swimmer_twin.h", os.path.join("..", "..", "include", "swimmer_hip.h")]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC"] + os.environ.get("SWIMMER_HIPCC_EXTRA", "").split()
LINK_LIBS = ["-ldl"]   # direct_comm.cpp resolves RCCL at run time


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the swimmer HIP library cannot be built")
    return exe


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


RLGLUE_LIB_PATH = os.path.join(CSRC, "librlglue_swimmer_hip.so")


def build_library(force=False, verbose=False):
    """Compile the HIP kernels + C ABI into csrc/libswimmer_hip.so for gfx950, and the
    RL-Glue environment plug-in (csrc/librlglue_swimmer_hip.so) on top of it."""
    if not force and not is_stale() and os.path.exists(RLGLUE_LIB_PATH):
        return LIB_PATH
    objects = []
    for src in sorted(HOST_ONLY):      # host-only sources: one plain C++ compile each, linked in below
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        ccmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", "-x", "c++", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(ccmd))
        subprocess.check_call(ccmd, cwd=CSRC)
        objects.append(obj)
    # (objects first: hipcc puts `-x hip` in front of the first source it sees and everything after it)
    cmd = ([_hipcc()] + HIPCC_FLAGS + objects + [os.path.join(CSRC, s) for s in SOURCES if s not in HOST_ONLY]
           + LINK_LIBS + ["-o", LIB_PATH])
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    cmd = [_hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, "rlglue_env.cpp"), "-L" + CSRC,
                                      "-lswimmer_hip", "-Wl,-rpath,$ORIGIN", "-o", RLGLUE_LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH
