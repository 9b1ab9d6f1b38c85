"""Build libalpro_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m alpro_amd.build [--force] [--ablations]

--ablations (or ALPRO_ABLATIONS=1) builds the MEASUREMENT variant lib/libalpro_hip_ablate.so with -DALPRO_ABLATIONS: the
result-corrupting knobs of tools/ (gemm_tune 3/4/10/11/12, tn_kind 1) exist only there; load it with ALPRO_HIP_LIB=<path>.

One object per .hip file (parallel), linked into alpro_amd/lib/libalpro_hip.so.  The .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
ABLATIONS = os.environ.get("ALPRO_ABLATIONS", "0") == "1" or "--ablations" in sys.argv
if ABLATIONS:
    OBJDIR = os.path.join(LIBDIR, "obj_ablate")
LIB = os.path.join(LIBDIR, "libalpro_hip_ablate.so" if ABLATIONS else "libalpro_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Werror=inline-asm"] + (["-DALPRO_ABLATIONS"] if ABLATIONS else [])


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_digest(src):
    h = hashlib.sha256()
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "alpro_hip.h"))
    for p in [src] + hdrs:
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, force):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    stamp = obj + ".sha"
    dig = _deps_digest(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    open(stamp, "w").write(dig)
    return obj, True


def build(force=False, verbose=True, force_files=()):
    """force: recompile everything; force_files: basenames (e.g. 'optim.hip') recompiled even when their digest stamp is current --
    the driver's "does it build" check uses this to exercise hipcc on every run without paying for all nine translation units."""
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force or os.path.basename(s) in force_files), srcs))
    objs = [o for o, _ in res]
    if verbose:
        for s, (_, compiled) in zip(srcs, res):
            print("  %-20s %s" % (os.path.basename(s), "compiled (hipcc --offload-arch=gfx950)" if compiled else "up to date (source digest matches its stamp)"))
    if any(c for _, c in res) or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
