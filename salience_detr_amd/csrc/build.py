"""Build libsalience_hip.so for gfx950 (MI355X) with hipcc.  In-tree output, git-ignored.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built library
travels to the GPU box with the repo snapshot.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["abi.hip", "msda_forward.hip", "msda_backward.hip", "msda_backward_tiled.hip", "msda_resident.hip", "topk.hip", "rows.hip", "plumbing.hip", "norm.hip", "salience_head.hip", "encoder_rows.hip", "ffn.hip", "token_linear.hip", "decoder_ops.hip", "proposals.hip", "salience_criterion.hip", "attention.hip", "topk_attention.hip", "gemm_x3.hip", "fused_head_value.hip", "neck.hip", "layer_norm_train.hip", "sampling_prep.hip", "attention_train.hip", "mlp_rows.hip"]
LIB = os.path.join(os.path.dirname(HERE), "libsalience_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-munsafe-fp-atomics"]


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


# benchmark build: msda_resident.hip with its ablated instantiations and phase stamps (wrong results by construction:
# never part of the product library), every other object shared with the product build
# (lives under benchmarks/: the product package ships the two product libraries only; the benchmark scripts that want it
# point salience_detr_amd._hip.LIB_PATH at it themselves before the first operator call)
ABLATE_LIB = os.path.join(ROOT, "benchmarks", "libsalience_hip_ablate.so")
# -DSDETR_AB_SWITCHES (common.h ab_env): the benchmark scripts' environment switches exist in this library only
ABLATE_SOURCES = {"msda_resident.hip": ["-DSDETR_MSDA_ABLATIONS", "-DSDETR_AB_SWITCHES"], "topk.hip": ["-DSDETR_AB_SWITCHES"],
                  "neck.hip": ["-DSDETR_AB_SWITCHES"], "salience_head.hip": ["-DSDETR_AB_SWITCHES"]}


def build_ablations(force: bool = False, verbose: bool = False) -> str:
    build(force=force, verbose=verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(ROOT, "include", "salience_hip.h")]
    objs = []
    for src in SOURCES:
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if src in ABLATE_SOURCES:
            s = os.path.join(HERE, src)
            o = os.path.join(OBJ_DIR, src.replace(".hip", ".ablate.o"))
            if force or _stale(o, [s] + headers):
                cmd = [hipcc, *FLAGS, *ABLATE_SOURCES[src], "-x", "hip", "-c", s, "-o", o]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _stale(ABLATE_LIB, objs):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", ABLATE_LIB], check=True)
    return ABLATE_LIB


# The fp16-activation flavour (round 5): the SAME sources with -DSDETR_ACT_F16 (common.h: every 16-bit activation is IEEE
# half instead of bfloat16, v_mfma_*_f16 instead of *_bf16) -> libsalience_hip_f16.so, same C ABI.  Loaded next to the
# bf16 library by _hip.lib(torch.float16); -Bsymbolic keeps each library's internal calls inside it.
F16_LIB = os.path.join(os.path.dirname(HERE), "libsalience_hip_f16.so")
F16_OBJ_DIR = os.path.join(HERE, "_obj_f16")


def _build_flavour(lib, obj_dir, defines, force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(obj_dir, exist_ok=True)
    headers = sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(ROOT, "include", "salience_hip.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *defines, "-x", "hip", "-c", s, "-o", o])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", *objs, "-o", lib])
    return lib


def build(force: bool = False, verbose: bool = False) -> str:
    _build_flavour(F16_LIB, F16_OBJ_DIR, ["-DSDETR_ACT_F16"], force, verbose)
    return _build_flavour(LIB, OBJ_DIR, [], force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--ablations" in sys.argv:
        print(build_ablations(force="--force" in sys.argv, verbose=True))
