"""Build liblcr_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python lcr-net_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU.  Objects go to csrc/_obj/, the library to lcr-net_amd/liblcr_hip.so
(git-ignored, but it travels to the GPU box with the working tree).  -ffp-contract=off everywhere: the
subsample / radius-search kernels must reproduce the reference's un-fused fp32 arithmetic bit for bit
(SURVEY §7 "hard parts"); kernels that want FMA call fmaf / MFMA explicitly.  -munsafe-fp-atomics: hardware
fp64/fp32 atomic adds (global_atomic_add_f64 / ds_add_f64) instead of CAS loops for the GroupNorm statistics.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(PKG, "liblcr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-value", "-Wno-unused-result"]
# Per-file additions.  -amdgpu-mfma-vgpr-form: MFMA accumulators in VGPRs instead of AGPRs.  With AGPR accumulators the register allocator
# rotates the four small accumulators of the KPConv aggregation through overlapping AGPR ranges and repairs the rotation with 16
# v_accvgpr_read / mov / write per four-neighbour trip (22 % of the loop's VALU instructions, the unit that kernel is short of), and the
# wider layers hold 120 / 190 registers instead of 75 / 112 (4 -> 6 and 2 -> 4 wavefronts per SIMD).  The attention kernel loses 160
# accumulator moves (99 vs 102-109 us per launch at 8 pairs per call).
FILE_FLAGS = {"kpconv.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
if os.environ.get("LCR_BUILD_EXTRA"):       # experiments: "file.hip:-flag -flag;other.hip:-flag"
    for part in os.environ["LCR_BUILD_EXTRA"].split(";"):
        name, _, fl = part.partition(":")
        FILE_FLAGS.setdefault(name.strip(), [])
        FILE_FLAGS[name.strip()] = FILE_FLAGS[name.strip()] + fl.split()


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    deps = sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(PKG, "..", "include", "lcr_hip.h")]
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _newer(s, o, deps):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return o

    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
