"""Build libfalnet_hip.so (gfx950 code objects + C-ABI host stubs) in-tree with hipcc.

`python -m fal_net_amd._build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU;
the resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libfalnet_hip.so")
SOURCES = ["api.cpp", "replay.cpp", "med_head.hip", "med_head2.hip", "losses.hip", "elementwise.hip", "data.hip", "wgrad_rows.hip", "wgrad_wave.hip", "conv_wave.hip", "conv_dma.hip", "conv.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# Per-file flags.  The MED head kernels are f32 VALU work per pixel and plane: the SLP vectoriser pairs independent scalar
# f32 operations into v_pk_* with extra moves to form the pairs, which is slower here (forward 88 -> 79 us, backward 157 -> 153 us
# at 8x192x640x49, same box: profiles/r05_ab_head_noslp.txt).
FILE_FLAGS = {"med_head.hip": ["-fno-slp-vectorize"], "med_head2.hip": ["-fno-slp-vectorize"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_ab(tag="ab", defines=(), verbose=True, raw=()):
    """Experiment build: the same sources with -DFALNET_AB (the FALNET_* kernel switches of include/falnet_hip.h are honoured) plus
    extra -D flags -> fal_net_amd/libfalnet_hip_<tag>.so, selected at run time with FALNET_LIB=<path>.  Never the product library."""
    obj_dir = os.path.join(CSRC, "_obj_" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    flags = FLAGS + ["-DFALNET_AB"] + ["-D" + d for d in defines]
    lib = os.path.join(HERE, f"libfalnet_hip_{tag}.so")

    def run(src):
        cmd = [HIPCC] + flags + FILE_FLAGS.get(src, []) + list(raw) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", os.path.join(CSRC, src), "-o", os.path.join(obj_dir, src + ".o")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, SOURCES))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [os.path.join(obj_dir, s + ".o") for s in SOURCES], check=True)
    return lib


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_epilogue.h"), os.path.join(HERE, "..", "include", "falnet_hip.h")]
    jobs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src + ".o")
        if force or _stale(obj, [path] + headers):
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    if "--ab" in sys.argv:  # python -m fal_net_amd._build --ab [tag] [-DNAME[=V] ...] [-f<compiler flag> ...]
        rest = [a for a in sys.argv[sys.argv.index("--ab") + 1:]]
        tag = next((a for a in rest if not a.startswith("-")), "ab")
        print(build_ab(tag, [a[2:] for a in rest if a.startswith("-D")], raw=[a for a in rest if a.startswith("-") and not a.startswith("-D")]))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
