"""Builds csrc/*.hip into libsincformer_hip.so (gfx950 only, in-tree).

    python -m sincformer_metacog_speech_enhancement_amd.build [--force]

hipcc cross-compiles without a GPU; objects are rebuilt only when a source or
header is newer.  The .so is git-ignored but travels with the working tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsincformer_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", CSRC]
# per-source extras: keep the attention accumulators in VGPRs (the softmax works on them with
# VALU instructions; the default AGPR placement costs ~220 v_accvgpr moves per key tile)
# -fno-honor-nans: row maxima of the online softmax compile to bare v_max3_f32 (see attention.hip)
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
         "attention_pipe.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"]}


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _newer(src, obj, hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, log in ex.map(cc, jobs):
                if verbose and log.strip():
                    print(log)
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s\n%s" % (src, log))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
