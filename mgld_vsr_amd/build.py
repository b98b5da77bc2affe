"""Build libmgld_hip.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU container too; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmgld_hip.so")
STAMP = os.path.join(HERE, "csrc", ".build_stamp")
SOURCES = ["runtime.hip", "igemm.hip", "conv3q.hip", "ppgemm.hip", "conv3r.hip", "pptconv.hip", "norm.hip", "attention.hip", "elementwise.hip", "raft.hip", "hpenc.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _file_digest(paths):
    h = hashlib.sha256()
    for f in paths:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _common():
    return [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "igemm_common.h"), os.path.join(CSRC, "pp_common.h"), os.path.join(HERE, "..", "include", "mgld_hip.h")]


def _digest():
    return _file_digest([os.path.join(CSRC, s) for s in SOURCES] + _common())


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into libmgld_hip.so (objects whose source, headers and flags are unchanged are kept).
    Returns the library path."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip().split("\n")[0] == dig:
                return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        odig = _file_digest([os.path.join(CSRC, s)] + _common())
        ostamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            continue
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print("[mgld build]", " ".join(cmd), flush=True)
        procs.append((s, ostamp, odig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, ostamp, odig, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {s}")
        elif verbose and out.strip():
            print(out.decode(errors="replace"))
        with open(ostamp, "w") as fh:
            fh.write(odig)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[mgld build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
